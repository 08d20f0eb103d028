/*
 * vdbhip.h -- C-ABI of libvdbhip.so, the MI355X (gfx950) brute-force / IVF-Flat k-NN backend.
 *
 * This is the drop-in boundary for ONE hot path of Human-Augment-Analytics/vectordb-retrieval:
 * what the reference delegates to `faiss.IndexFlat` / `faiss.IndexIVFFlat` / NumPy on behalf of
 *   src/algorithms/exact_search.py:26-78        ExactSearch.build_index / search / batch_search
 *   src/algorithms/modular.py:121-133, 312-390  BruteForceIndexer.build, LinearSearcher.*
 *   src/algorithms/modular.py:244-309, 418-449, 536-548  FaissFactory/IVFIndexer.build, FaissSearcher.*
 *   src/algorithms/approximate_search.py:28-87  ApproximateSearch (IVFn,Flat keys)
 *   src/benchmark/dataset.py:497-504, 858-964   brute-force ground truth
 * Plain pointers and sizes only; no torch / numpy types.  The reference-side binding (ctypes) is
 * shown in INTEGRATION.md and implemented in vectordb-retrieval_amd/vdbhip/_ffi.py.
 *
 * Conventions (identical to faiss.IndexFlat, i.e. what ExactSearch.batch_search returns,
 * exact_search.py:78):
 *   metric VDB_METRIC_L2 : distances are SQUARED L2, ascending.
 *   metric VDB_METRIC_IP : "distances" are raw inner products, descending.
 *   ids are int64 row numbers (+ id_base of vdb_add); when k > ntotal the tail is padded with
 *   id -1 and distance +FLT_MAX (L2) / -FLT_MAX (IP).
 *   Ties are broken by the smaller id (shard-count invariant).
 * The LinearSearcher / FaissSearcher conventions (sqrt, negated scores, cosine = normalise + IP,
 * +inf padding; modular.py:355-360, 381-385, 545-546) are applied by the Python shim on the (nq,k) output.
 *
 * Exactness contract: the neighbours returned are the exact k nearest under float64 arithmetic
 *   L2: sum_d fma(t,t,.) with t = (double)x[d]-(double)q[d];  IP: sum_d fma((double)q[d],(double)x[d],.)
 * (d ascending), the order key being (value, id).  The fp16 MFMA scan only nominates candidates;
 * every returned neighbour is re-scored in this arithmetic and a rigorous error bound guarantees that
 * no true neighbour was left out (DESIGN.md "exactness guard").
 *
 * Threading: a handle may be used from one host thread at a time.  All calls return a status
 * code; vdb_last_error() gives the message of the last failure on the calling thread.
 */
#ifndef VDBHIP_H
#define VDBHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDB_ABI_VERSION 4

typedef struct vdb_index_s *vdb_handle;

enum vdb_metric { VDB_METRIC_L2 = 0, VDB_METRIC_IP = 1 };

enum vdb_status {
    VDB_OK = 0,
    VDB_ERR_INVALID = 1,     /* bad argument (maps to ValueError at build time, RuntimeError at search time) */
    VDB_ERR_STATE = 2,       /* e.g. search before add: "Index has not been built yet." (exact_search.py:53-54) */
    VDB_ERR_HIP = 3,         /* HIP runtime failure */
    VDB_ERR_NOMEM = 4,
    VDB_ERR_UNSUPPORTED = 5
};

/* which search path served the last call (vdb_stats_t.last_path) */
enum vdb_path { VDB_PATH_NONE = 0, VDB_PATH_EXACT_SCAN = 1, VDB_PATH_MFMA_SCAN = 2, VDB_PATH_IVF = 3 };

typedef struct vdb_stats_s {
    int64_t ntotal;            /* rows indexed */
    int32_t dim;
    int32_t metric;
    int64_t bytes_resident;    /* device bytes owned by the handle (index + workspace) */
    int32_t last_path;         /* enum vdb_path */
    int32_t corpus_fp16_exact; /* 1 if every corpus value is exactly representable in the fp16 scan copy */
    int64_t last_nq;
    int64_t last_candidates;   /* candidate groups (4 consecutive rows each; 8 on the flat int8 scan) re-scored exactly */
    int64_t last_rescan_bins;  /* 256-row bins re-scanned exactly (collision guard) */
    int64_t last_fallback_queries; /* queries whose work list overflowed -> exhaustive exact scan */
    float last_scan_ms;        /* mean HIP-event time of the dominant (scan) kernel over the searches recorded since
                                  timing was switched on; 0 if not timed */
    float last_total_ms;       /* same for the whole device pipeline */
    int32_t nlist;             /* IVF: number of inverted lists (0 = flat index) */
    int32_t nprobe;
    int32_t scan_dtype;        /* arithmetic of the last MFMA scan: 0 = fp16 (f32 accumulate), 1 = int8 (i32 accumulate) */
    int32_t has_i8_copy;       /* 1 if the index holds the int8 scan copy (byte-valued integer corpus, D <= 128) next to the float32
                                  rows and the fp16 copy; 2 if it holds ONLY the int8 copies (option "int8_only") */
    int64_t last_rows_scanned; /* IVF: (query, row) pairs scanned by the last search (rows of the probed lists) */
    int64_t upload_blocks;     /* row blocks the last vdb_add / vdb_ivf_add streamed through the pinned staging buffers */
    int64_t graph_replays;     /* searches served by launching the captured hipGraph (option "graph") since the handle was made */
    int32_t ndevices;          /* shards of the handle: 1, or the ndev of vdb_create_multi (sums / maxima over the shards above) */
    int32_t scan_shape;        /* rows of the MFMA tile the flat scan copies (D <= 128) are laid out for: 16 = layout "x16"
                                  (v_mfma_*_16x16x32_f16 / 16x16x64_i8, the default above 15 360 rows), 32 = 32x32x16 / 32x32x32
                                  (option "flat_shape" = 32, small corpora, quads); 0 = no such copy (D > 128, empty index) */
    int64_t bytes_workspace;   /* the part of bytes_resident that is per-search workspace (bin arrays, work lists, staging) */
    float last_prep_ms;        /* timing on: mean time from the start of the device pipeline to the start of the dominant kernel
                                  (query statistics / operands; IVF: + coarse search, plan) ... */
    float last_tail_ms;        /* ... and from its end to the end of the pipeline (bin select + exact refine): with last_scan_ms
                                  the three stages of a search, so a caller can name the longest one */
} vdb_stats_t;

/* ---- library ---------------------------------------------------------------------------- */
int vdb_abi_version(void);
const char *vdb_last_error(void);
int vdb_device_count(int *count);

/* ---- flat (brute-force) index -- replaces faiss.IndexFlat(d, metric) (exact_search.py:38) -- */
int vdb_create(int dim, int metric, int device, vdb_handle *out);
/* ONE index over several GPUs of the node, driven from ONE process -- the form SURVEY 8b proposed
 * (`vdb_create(dim, metric, const int *devs, int ndev, ...)`), because the reference's harness is a single process that calls
 * build_index once and batch_search per batch (src/experiments/experiment_runner.py:329-331, 428-434) on a plugin made from a
 * `type:` string (src/algorithms/__init__.py:37-47).  The handle is an ordinary vdb_handle:
 *   vdb_add / vdb_add_device / vdb_ivf_add(_assigned)  cut the rows of a call into ndev contiguous blocks, block s -> devices[s]
 *                  (SURVEY 8e); ids stay id_base + insertion order, appends work as on one device;
 *   vdb_search / vdb_ivf_search (+ _device, _partial_device; device pointers = memory of devices[0])  run every shard's partial
 *                  search on its own device, stream and host thread, gather the packed partials into devices[0] by peer copies
 *                  over xGMI and merge them there on (float64 key, global id): the result is bit-identical to the
 *                  single-device index of the same rows, whatever ndev;
 *   vdb_ivf_train  runs k-means once (on devices[0], over the whole training set); every shard files its rows under the same
 *                  centroids and probes the same lists;
 *   vdb_reserve, vdb_stats (sums / maxima over the shards, ndevices), vdb_set_option (forwarded), vdb_reset, vdb_destroy,
 *   vdb_ivf_set_centroids / _get_centroids / _set_nprobe / _get_assignment, vdb_rerank(_device)  work as on one device.
 * Not available on such a handle (VDB_ERR_UNSUPPORTED): option "graph", the debug hooks.
 * Option "multi_stage_all" = 1 (tests) makes shards on devices[0] take the remote-shard path too (own query copy, packed buffer,
 * peer copy), so a one-GPU box exercises the code a multi-GPU node runs.
 * A device may be listed more than once (several shards on one GPU). */
int vdb_create_multi(int dim, int metric, const int *devices, int ndev, vdb_handle *out);
int vdb_destroy(vdb_handle h);

/* replaces index.add(vectors) (exact_search.py:39; modular.py:124-130 keeps the raw matrix):
 * uploads n rows (row-major float32, host memory) and builds the scan copy.  APPENDS, as faiss.Index.add does: row i of
 * the index (in insertion order over all adds) gets id id_base + i; id_base (row-sharded corpora pass their shard offset)
 * belongs to the index -- every add of one index passes the same value (VDB_ERR_INVALID otherwise), vdb_reset empties the
 * index.  An append re-derives the scan copies from the float32 rows on the device (~0.3 s per 12.5M x 768 rows) and
 * needs room for the old and the grown row buffer side by side while it copies.
 * The rows are streamed in blocks (option "upload_block_mb", default 64 MiB) through two pinned staging buffers, so
 * x_host may be a memory-mapped file far larger than host RAM comfortably holds (dataset.py:376-471, 1001-1052). */
int vdb_add(vdb_handle h, const float *x_host, int64_t n, int64_t id_base);
/* same, rows already in device memory of the handle's GPU */
int vdb_add_device(vdb_handle h, const float *x_dev, int64_t n, int64_t id_base, void *stream);

/* replaces index.reset(): drops every row (flat and IVF; an IVF index keeps its centroids).  A search before the next add
 * fails with VDB_ERR_STATE. */
int vdb_reset(vdb_handle h);

/* replaces index.search(queries, k) (exact_search.py:58,78): host buffers, synchronous.
 * D (nq,k) float32, I (nq,k) int64, caller-allocated. */
int vdb_search(vdb_handle h, const float *q_host, int64_t nq, int k, float *D, int64_t *I);
/* device-resident variant: all pointers are device memory on the handle's GPU; work is enqueued on
 * `stream` (a hipStream_t, NULL = default stream) and NOT synchronised. */
int vdb_search_device(vdb_handle h, const float *q_dev, int64_t nq, int k, float *D_dev, int64_t *I_dev,
                      void *stream);

/* ---- row-sharded search (one process per GPU; partials exchanged with an RCCL all-gather) ---- */
/* per-shard partial top-k: float64 order keys (L2: squared distance, IP: -score) and global ids,
 * sorted ascending by (key,id); missing entries have id -1 / key +inf. */
int vdb_search_partial_device(vdb_handle h, const float *q_dev, int64_t nq, int k, double *keys_dev,
                              int64_t *ids_dev, void *stream);
/* merge nparts partial lists laid out (nparts, nq, k) into the final (nq,k) result. */
int vdb_merge_partials_device(int metric, int device, const double *keys_dev, const int64_t *ids_dev, int nparts,
                              int64_t nq, int k, float *D_dev, int64_t *I_dev, void *stream);

/* same merge for ONE packed buffer per part -- what a single RCCL all-gather produces when every rank sends
 * its keys (nq*k doubles) immediately followed by its ids (nq*k int64): layout (nparts, 2, nq, k) 8-byte words. */
int vdb_merge_packed_partials_device(int metric, int device, const void *packed_dev, int nparts, int64_t nq, int k,
                                     float *D_dev, int64_t *I_dev, void *stream);

/* ---- candidate re-scoring -- replaces the per-query NumPy loop of FaissSearcher._batch_search_lsh_rerank
 *      (modular.py:483-532: gather candidate rows by id -> exact L2 / inner product -> top-k) and
 *      LSHSearcher._compute_distances (lsh.py:242-250) ------------------------------------------------ */
/* cand (nq, ncand) int64 row ids (id_base-relative ids as returned by search; -1 = empty slot, ids of one
 * query must be distinct).  Output: the k best candidates of every query, flat conventions and padding. */
int vdb_rerank(vdb_handle h, const float *q_host, int64_t nq, const int64_t *cand_host, int ncand, int k, float *D,
               int64_t *I);
int vdb_rerank_device(vdb_handle h, const float *q_dev, int64_t nq, const int64_t *cand_dev, int ncand, int k,
                      float *D_dev, int64_t *I_dev, void *stream);

/* ---- IVF-Flat -- replaces faiss.index_factory(d, "IVF<nlist>,Flat", metric) + train/add/search
 *      (modular.py:277-286, 437-441, 544; approximate_search.py:39-51, 87) ------------------- */
/* k-means (Lloyd) on at most max_points_per_centroid*nlist rows sampled with `seed`; niter iterations. */
int vdb_ivf_train(vdb_handle h, int nlist, const float *x_host, int64_t n, int niter, uint64_t seed,
                  int max_points_per_centroid);
/* inject centroids (nlist, dim) instead of training -- used by parity tests and index loading */
int vdb_ivf_set_centroids(vdb_handle h, const float *centroids_host, int nlist);
int vdb_ivf_get_centroids(vdb_handle h, float *centroids_host);
/* assign rows to their nearest centroid and build the inverted lists (CSR, vectors grouped by list).  APPENDS like vdb_add
 * (same id rule); the lists are the ones a single add of all rows builds (rows of a list stay in insertion order). */
int vdb_ivf_add(vdb_handle h, const float *x_host, int64_t n, int64_t id_base);
/* same with the list of every row given (int32 (n), as vdb_ivf_get_assignment returned it for this corpus and these
 * centroids): what loading a persisted index does -- no coarse assignment pass (covertree_v2_2.py:184-282 is the
 * reference's load protocol).  A row whose list id is out of range is an error. */
int vdb_ivf_add_assigned(vdb_handle h, const float *x_host, int64_t n, int64_t id_base, const int32_t *list_of_row_host);
int vdb_ivf_set_nprobe(vdb_handle h, int nprobe);
/* list id of each indexed row, int32 (n) -- parity tests compare it with the oracle's assignment */
int vdb_ivf_get_assignment(vdb_handle h, int32_t *list_of_row_host);
int vdb_ivf_search(vdb_handle h, const float *q_host, int64_t nq, int k, float *D, int64_t *I);
int vdb_ivf_search_device(vdb_handle h, const float *q_dev, int64_t nq, int k, float *D_dev, int64_t *I_dev,
                          void *stream);
/* row-sharded IVF (coarse quantizer replicated, the rows of every list split over the ranks): per-shard
 * partial top-k among the probed lists, same layout and merge as vdb_search_partial_device */
int vdb_ivf_search_partial_device(vdb_handle h, const float *q_dev, int64_t nq, int k, double *keys_dev,
                                  int64_t *ids_dev, void *stream);

/* Sizes the search workspace for batches of up to nq queries and top-k NOW instead of inside the first search (works on
 * flat and IVF handles after add): one untimed search whose queries are corpus rows.  The reference times its very first
 * batch_search, allocations included (experiment_runner.py:431-437; metrics_methodology.md:119-121: no warm-up) -- the
 * plugins call this from build_index (`reserve_queries`, default 10 000), as FAISS' GPU resources reserve their scratch
 * memory at construction. */
int vdb_reserve(vdb_handle h, int64_t nq, int k);

/* ---- introspection / tuning ---------------------------------------------------------------- */
int vdb_stats(vdb_handle h, vdb_stats_t *out);
/* Options (vdb_set_option; every setting returns exact results unless it says otherwise):
 *   behaviour
 *     "force_path"      0 auto | 1 exact kernels only | 2 MFMA scan whenever legal | 3 exact kernels, one query per wave
 *     "timing"          1: (re)start recording HIP-event times of every search on its stream, averaged by vdb_stats
 *     "list_cap"        work-list capacity per query (0 = default max(64, 2k + 32))
 *     "panel_layout"    0 auto (16-row-tile panels for D > 128) | 1 32-row tiles for every D | 2 16-row tiles for every D;
 *                       takes effect at the next vdb_add
 *     "panel_dtype"     0 auto: byte-valued integer corpora are ALSO kept as an int8 scan copy, and integer query batches
 *                       in the byte window are scanned with int8 MFMA | 1 fp16 scan only
 *     "int8_only"       flat index, D <= 128, more than 32 768 rows, takes effect at the next vdb_add: 0 (default) | 1: a byte-valued
 *                       corpus (every value an integer in 0..255 or in -128..127) keeps ONLY its int8 copies -- row-major int8 rows
 *                       + int8 MFMA panels + their accumulator inits, 0.55x the float32 bytes instead of 3x (the reference holds
 *                       one copy of the corpus, exact_search.py:34-39).  Same results: integer query batches take the int8 scan
 *                       and the integer refine as always; a batch with a non-integer value is scanned in fp16 over slabs converted
 *                       from the int8 panels per search (option "int8_slab_chunks", default 8 scan chunks = 16 MiB of scratch at
 *                       D = 128) and refined in float64 from x = byte + cx.  The rows stream through in blocks at build time (a
 *                       100M x 128 shard never exists in float32 inside the library).  Built by ONE add: appending is
 *                       VDB_ERR_UNSUPPORTED; a corpus that is not byte-valued silently gets the default layout
 *                       (vdb_stats.has_i8_copy tells: 2 = int8 only)
 *     "stream_panels"   D > 128, takes effect at the next vdb_add: 0 (default) the fp16 scan copy stays resident next to the
 *                       float32 rows | 1 it is NOT kept: every search converts the float32 rows slab by slab into one
 *                       scratch slab and scans that (same results; 1.8x -> ~1.15x the corpus bytes resident for a corpus that
 *                       is not exact in fp16, one extra pass over the rows per query batch)
 *     "stream_slab_rows" rows of that scratch slab (0 = default 1 280 000, rounded down to whole scan chunks whose
 *                       workgroups fill whole rounds of the chip)
 *     "upload_block_mb" staging block of the row-block ingestion (default 64)
 *     "small_batch"     1 (default): batches of <= 512 queries are scanned with finer row chunks and, up to 256 queries,
 *                       1 / 2 / 4-wave workgroups, so that the grid still covers the chip; D > 128: waves without queries
 *                       only stage panels, <= 16 queries keep their query block in LDS | 0 the batch shape for every size
 *     "fused_stats"     1 (default): batches of <= 4096 query values (serving shapes) take their statistics inside the
 *                       query-operand kernel, one dependent dispatch less | 0 a separate statistics dispatch for every size
 *     "ivf_min_batch"   smallest query batch the list-major MFMA scan of the IVF index serves (default 1); smaller ones
 *                       take the exact per-query list scan
 *     "graph"           0 (default) | 1: a vdb_search_device / vdb_search_partial_device / vdb_ivf_search*_device call of
 *                       at most 4096 queries on a non-null stream that repeats with the same buffers, shape and stream
 *                       (a serving loop) is captured into a hipGraph on its second occurrence and replayed afterwards;
 *                       any vdb_set_option / add / train drops the graph; the caller keeps the buffers alive and
 *                       rewrites the queries in place.  (The call that drops a stale graph always runs eagerly: on ROCm 7.x an
 *                       executable graph instantiated right behind the destruction of its predecessor faults on its second
 *                       replay -- a runtime defect in graph packet capture, profiles/r04_graph_fault_cause.txt.)
 *     "graph_recapture_at_once"  diagnostic, 0 (default) | 1: the pre-round-3 ordering, for re-checking that defect
 *                       ($VDBHIP_ALLOC_LOG=<file> logs every allocation / graph event for scripts/graph_fault_analyze.py)
 *   tuning knobs (scripts/sweep_*.py)
 *     "i8_variant"      0..7: tile / stage / wave shapes of the flat int8 scan (6 / 7: variant 3 with a pacing barrier
 *                       per 1 / 2 tiles)
 *     "i8_group"        8 (default) | 4 rows per select group of the flat int8 scan
 *     "flat_shape"      (before vdb_add; alias "i8_shape") MFMA shape of the flat scans for D <= 128 and the layout of their scan
 *                       copies: 0 auto (16) | 16 | 32.  16 = v_mfma_f32_16x16x32_f16 / v_mfma_i32_16x16x64_i8 on layout "x16"
 *                       (octs only; +17 - 19 % on the scans); 32 = the 32x32 kernels (also taken when an option asks for quads
 *                       -- "f16_group" / "i8_group" = 4 -- or for an A/B schedule of "scan_variant")
 *     "scan_pair"       layout "x16", index with an int8 copy: 1 (default) both scans in one launch (the device picks the body), 0 two
 *                       launches (the one not needed returns at once) -- A/B and diagnosis
 *     "f16_stage_tiles" / "f16_wide" / "scan_prio"   tuning of the x16 kernels (tiles per LDS stage of the fp16 batch scan: 0 auto | 4 | 8;
 *                       1024-query workgroup tiles of the fp16 scan at D <= 64: 0 auto | 1 never; issue priority of one half of a
 *                       workgroup: 0 | 1 | 2 -- measured, no gain)
 *     "f16_group"       8 (default) | 4 rows per select group of the fp16 flat scan (D <= 128)
 *     "i8_ring"         0 auto (4) | 2 | 4 | 8 LDS staging stages of the serving-shaped and IVF int8 scans
 *     "i8_nt"           0 (default) / 2: the serving-shaped int8 scan stages its panels with non-temporal loads | 1 off
 *     "ivf_nw"          0 auto | 2 / 4 / 8 waves per IVF work item
 *     "ivf_bt"          0 auto | 4 (64-row bins) | 16 (the largest: one bin per half of a 256-row span)
 *     "ivf_tps"         D > 128, takes effect at the next vdb_ivf_add: 0 auto | 16 (256-row spans, 64-row bins) | 64
 *                       (1024-row spans, 256-row bins) of the p16 panel space
 *     "ivf_group"       D > 128, 64-row bins: rows per candidate group of the list scan, 0 auto | 1 | 2 | 4 (smaller groups cost
 *                       select instructions in the scan and save gathered rows in the exact refine)
 *     "ivf_tile"        D > 128, 256-row spans: workgroup tile of the list scan, 0 auto / 2 = 256 rows x 256 query slots | 1 = 128 x 512
 *     "ivf_part"        0 auto | spans (256 rows) per row part of the IVF list scan: long lists are cut into parts
 *                       scanned by one workgroup each (rounded up to a multiple of 4 bins)
 *     "select_variant"  0..2;  "spans_per_chunk", "kloop_qgroup": grid shaping of the flat scans
 *     "scan_variant"    and the timing-only ablations exist only in -DVDB_ABLATIONS builds (`make ablations`, WRONG
 *                       results by design); the shipped library rejects them. */
int vdb_set_option(vdb_handle h, const char *key, double value);

/* ---- test hooks (used by tests/ to validate the error bound of the fp16 scan) -------------- */
/* raw scan scores (scaled units) for queries x rows [row0,row0+nrows), any D <= 4096: out (nq, nrows) float32,
 * together with the per-query bound eps (nq) and the scale cs so that score/cs ~ (||x||^2 - 2 q.x) or -q.x */
int vdb_debug_scan_scores(vdb_handle h, const float *q_host, int64_t nq, int64_t row0, int64_t nrows,
                          float *scores_host, float *eps_host, double *cscale);

/* per-wave cycle stamps {head, mfma, select, barrier, total, late, stages, 0} left by the diagnostic scan build
 * (option scan_variant = 6; never used for results) -- scripts/stamp_scan.py */
int vdb_debug_fetch_stamps(vdb_handle h, unsigned long long *out_host, int64_t max_words, int64_t *nwords);

#ifdef __cplusplus
}
#endif
#endif /* VDBHIP_H */
