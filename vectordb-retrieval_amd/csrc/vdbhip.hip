// vdbhip.hip -- host side of libvdbhip.so: handle management, path selection, kernel launches,
// and the extern "C" entry points declared in include/vdbhip.h.  gfx950 only.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vdbhip.h"
#include "common.hpp"
#include "prep.hpp"
#include "refine.hpp"
#include "scan.hpp"
#include "scan16.hpp"
#include "scan_i8.hpp"
#include "scan_i8x16.hpp"
#include "scan_x16.hpp"
#include "ivf.hpp"
#include "ivf_mfma.hpp"
#include "ivf_kloop.hpp"
#include "dense.hpp"

using namespace vdb;

namespace {

thread_local std::string g_last_error;

// bumped by every (re)allocation or release of a DevBuf: a captured hipGraph holds raw buffer addresses, so it may only be
// replayed while no buffer of this process has moved since its capture (graph_or_run)
std::atomic<uint64_t> g_alloc_epoch{0};

// Diagnostic allocation log ($VDBHIP_ALLOC_LOG=<file>): one line per device / pinned allocation and release of this library
// ("A <ptr> <bytes>", "F <ptr> <bytes>", "HA"/"HF" for pinned host memory) and per hipGraph event of graph_or_run, flushed as
// it is written -- after a GPU memory fault the faulting address is resolved against it offline (live range, freed range,
// or not ours: profiles/r04_graph_fault_cause.txt).  Off unless the variable is set.
FILE *alloc_log() {
    static FILE *f = [] {
        const char *path = getenv("VDBHIP_ALLOC_LOG");
        return (path && *path) ? fopen(path, "a") : (FILE *)nullptr;
    }();
    return f;
}
void alloc_note(const char *what, const void *p, size_t bytes) {
    if (FILE *f = alloc_log()) {
        fprintf(f, "%s %p %zu\n", what, p, bytes);
        fflush(f);
    }
}

// growable device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool borrowed = false;      // a view into another DevBuf's memory (borrow): never freed here, cannot grow
    void borrow(void *ptr, size_t bytes) {
        if (!borrowed) release();
        p = ptr;
        cap = bytes;
        borrowed = true;
    }
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (borrowed) throw Error(VDB_ERR_INVALID, "internal: a borrowed device buffer cannot grow");
        g_alloc_epoch.fetch_add(1, std::memory_order_relaxed);
        if (p) {
            alloc_note("F", p, cap);
            VDB_HIP(hipFree(p));
        }
        p = nullptr;
        cap = 0;
        const size_t want = bytes + std::min<size_t>(bytes >> 3, (size_t)256 << 20);   // (growth slack: 1/8, at most 256 MiB)
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) throw Error(VDB_ERR_NOMEM, "hipMalloc of " + std::to_string(want) + " bytes failed");
        cap = want;
        alloc_note("A", p, cap);
    }
    // exactly `bytes` (no growth slack): buffers of an index that is built once and never grows (int8-only build)
    void reserve_exact(size_t bytes) {
        if (bytes == cap && p) return;
        if (borrowed) throw Error(VDB_ERR_INVALID, "internal: a borrowed device buffer cannot grow");
        release();
        g_alloc_epoch.fetch_add(1, std::memory_order_relaxed);
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e != hipSuccess) { p = nullptr; throw Error(VDB_ERR_NOMEM, "hipMalloc of " + std::to_string(bytes) + " bytes failed"); }
        cap = bytes ? bytes : 16;
        alloc_note("A", p, cap);
    }
    // reserve that keeps the first `keep` bytes (append): old and new allocation coexist for the copy
    void grow(size_t bytes, size_t keep) {
        if (bytes <= cap) return;
        if (borrowed) throw Error(VDB_ERR_INVALID, "internal: a borrowed device buffer cannot grow");
        void *old = p;
        const size_t old_cap = cap;
        g_alloc_epoch.fetch_add(1, std::memory_order_relaxed);
        const size_t want = bytes + std::min<size_t>(bytes >> 3, (size_t)256 << 20);
        void *fresh = nullptr;
        hipError_t e = hipMalloc(&fresh, want);
        if (e != hipSuccess) {              // the old buffer stays as it was: a failed append leaves the index intact
            (void)hipGetLastError();
            p = old;
            cap = old_cap;
            throw Error(VDB_ERR_NOMEM, "hipMalloc of " + std::to_string(want) + " bytes failed");
        }
        p = fresh;
        cap = want;
        alloc_note("A", p, cap);
        if (old && keep) e = hipMemcpy(p, old, keep, hipMemcpyDeviceToDevice);
        if (old) {
            alloc_note("F", old, old_cap);
            (void)hipFree(old);
        }
        if (e != hipSuccess) throw Error(VDB_ERR_HIP, std::string("device copy failed: ") + hipGetErrorString(e));
    }
    void release() {
        if (p && !borrowed) {
            g_alloc_epoch.fetch_add(1, std::memory_order_relaxed);
            alloc_note("F", p, cap);
            (void)hipFree(p);
        }
        p = nullptr;
        cap = 0;
        borrowed = false;
    }
    template <class T>
    T *as() const { return reinterpret_cast<T *>(p); }
};

// temporary of one API call: freed on every exit path (an Error thrown by a later hipMalloc included)
struct ScopedDevBuf : DevBuf {
    ScopedDevBuf() = default;
    ScopedDevBuf(const ScopedDevBuf &) = delete;
    ScopedDevBuf &operator=(const ScopedDevBuf &) = delete;
    ~ScopedDevBuf() { release(); }
};

struct Workspace;
inline QueryBatchInfo *batch_info(Workspace &ws);

struct Workspace {
    DevBuf qpad, qpanels, qpanels8, qrows8, info, eps, bin_m1, bin_m2, bin_m3, bin_m4, bin_m5, sb_m1, sb_m2, sb_span;
    DevBuf cand, rescan, counts, fallback, fb_list, fb_done /* arrival counters of refine_fallback_body (refine_tail_kernel) */, small;  // small: fb_count (int) + 2 stat counters
    DevBuf dense;            // nq x Npad raw scores of the small-corpus path
    DevBuf pkeys, pids;      // partial lists of the exhaustive / fallback passes
    DevBuf stage_q, stage_d, stage_i;  // host-API staging
    size_t bytes() const {
        const DevBuf *all[] = {&qpad, &qpanels, &qpanels8, &qrows8, &info, &eps, &bin_m1, &bin_m2, &bin_m3, &bin_m4, &bin_m5, &sb_m1, &sb_m2, &sb_span, &cand,
                               &rescan, &counts, &fallback, &fb_list, &fb_done, &small, &pkeys, &pids, &stage_q, &stage_d,
                               &stage_i, &dense};
        size_t s = 0;
        for (auto b : all) s += b->cap;
        return s;
    }
    void release() {
        DevBuf *all[] = {&qpad, &qpanels, &qpanels8, &qrows8, &info, &eps, &bin_m1, &bin_m2, &bin_m3, &bin_m4, &bin_m5, &sb_m1, &sb_m2, &sb_span, &cand,
                         &rescan, &counts, &fallback, &fb_list, &fb_done, &small, &pkeys, &pids, &stage_q, &stage_d, &stage_i,
                         &dense};
        for (auto b : all) b->release();
    }
};

inline QueryBatchInfo *batch_info(Workspace &ws) {        // inside ws.small (common.hpp, kInfoOffset)
    return reinterpret_cast<QueryBatchInfo *>(ws.small.as<char>() + kInfoOffset);
}

}  // namespace

struct vdb_multi_s;           // multi.inc: the shards, streams and host threads of a multi-device handle

struct vdb_index_s {
    vdb_multi_s *multi = nullptr;            // non-null: a multi-device handle (vdb_create_multi); device = the primary device,
                                             // N / id_base / built / ivf_built / nlist / nprobe describe the whole index
    int device = 0;
    int dim = 0, D4 = 0, ksteps = 0, metric = 0;
    int64_t N = 0, Npad = 0, id_base = 0;
    bool built = false;
    // index arrays
    DevBuf x32, xnorm2, panels, bias, stats;
    // int8 scan copy (scan_i8.hpp): byte-valued integer corpora with D <= 128, kept NEXT TO the fp16 panels (a batch of
    // non-integer queries still takes the fp16 scan)
    DevBuf panels8, bias8, rows8, rowstat8;  // rows8 / rowstat8: row-major int8 copy + {sum x^2, sum x} for the list refine
    int rows8_pitch = 0;
    bool i8_ok = false;
    int ivf_bt = 0;                          // option "ivf_bt": tiles per level-1 bin of the IVF scan (0 auto, 4, 16)
    int ivf_part = 0;                        // option "ivf_part": spans per row part of the IVF list scan (0 auto)
    int ivf_min_batch = 1;                   // option "ivf_min_batch": smallest query batch the list-major MFMA scan serves
    int ivf_tile = 0;                        // option "ivf_tile": workgroup tile of the D > 128 list scan on 256-row spans (0 / 2 square, 1 = 128 x 512)
    int ivf_i8_group = 4;                    // option "ivf_i8_group": rows per candidate group of the int8 list scan (4 | 8)
    int ivf_group = 0;                       // option "ivf_group": rows per candidate group of the D > 128 list scan (0 auto, 1, 2, 4)
    int ivf_nw = 0;                          // option "ivf_nw": waves per IVF work item (0 auto, 2 / 4 / 8)
    int i8_group = 8;                        // rows per select group of the int8 scan (option "i8_group": 4 or 8)
    int f16_wide = 0;                        // option "f16_wide" (tuning, x16 fp16 batch scan, D <= 64): 0 auto (1024-query tiles when they fit), 1 never
    int f16_stage_tiles = 0;                 // option "f16_stage_tiles" (tuning, x16 fp16 batch scan): tiles per LDS stage, 0 auto | 4 | 8
    int scan_pair_off = 0;                   // option "scan_pair" = 0: never the paired launch of the two x16 scans (A/B, diagnosis)
    int scan_prio = 0;                       // option "scan_prio" (tuning, x16 kernels): issue priority of one half of the workgroup's waves
    int flat_shape_opt = 0;                  // option "flat_shape" (alias "i8_shape"; before vdb_add): MFMA shape of the flat scans for D <= 128,
                                             // 0 auto (16) | 16 | 32
    bool x16 = false;                        // the flat scan copies (fp16 and int8, D <= 128) are in layout "x16": 16x16x32 f16 / 16x16x64 i8 MFMA,
                                             // octs only (scan_x16.hpp, scan_i8x16.hpp); corpora the dense small-corpus path serves keep 32-row tiles
    int f16_group = 8;                       // ... and of the fp16 flat scan (option "f16_group": 4 or 8)
    // option "int8_only" (takes effect at the next add; flat index, D <= 128, > 32768 rows, byte-valued corpus): only the int8
    // copies are kept -- rows8 + panels8 + their biases, 0.55x the float32 corpus instead of 3x.  Integer query batches run as on
    // the default index; a batch with a non-integer value is scanned in fp16 over slabs converted from the int8 panels per search;
    // the exact kernels read x = byte + cx from the int8 rows (same float64 chains, same keys).  One add builds it (no append).
    int int8_only_opt = 0;
    bool int8_only = false;
    int64_t int8_slab_chunks = 0;            // option "int8_slab_chunks": scan chunks per converted fp16 slab (0 = default 8)
    int64_t int8_block_rows = 0;             // option "int8_block_rows": rows per ingestion block of the int8-only build (0 = 4M; tests)
    int i8_nt = 0;                           // option "i8_nt": non-temporal staging loads of the serving-shaped int8 scan (0 auto, 1 never, 2 always)
    int i8_ring = 0;                         // option "i8_ring": LDS staging stages of the streaming-shaped int8 scans (0 auto, 2, 4, 8)
    // option "graph": a device-resident search that repeats with the same shape and buffers (a serving loop) is captured
    // into a hipGraph on its second call and replayed from the third (graph_or_run)
    int graph_mode = 0;
    int graph_recapture_at_once = 0;         // diagnostic option of the same name: the pre-round-3 ordering (see graph_or_run)
    struct GraphKey {
        const void *q = nullptr, *o1 = nullptr, *o2 = nullptr;
        int64_t nq = 0;
        int k = 0, kind = 0, nprobe = 0;
        hipStream_t st = nullptr;
        bool operator==(const GraphKey &o) const {
            return q == o.q && o1 == o.o1 && o2 == o.o2 && nq == o.nq && k == o.k && kind == o.kind && nprobe == o.nprobe && st == o.st;
        }
    } graph_key, graph_warm;
    hipGraphExec_t graph_exec = nullptr;
    hipEvent_t graph_ev = nullptr;           // recorded behind every launch of graph_exec (graph_drop_exec waits for it)
    uint64_t graph_epoch = 0;                // g_alloc_epoch when the graph was captured
    int64_t graph_replays = 0;
    int small_batch_off = 0;                 // option "small_batch" = 0: batches <= 512 queries keep the batch-shaped grid
    int i8_cx = 0, i8_ks = 0, i8_disable = 0, i8_variant = 3;   // (variant 3: +2 % over 0 on the bench shape, scripts/sweep_i8.py)
    // host copies of the corpus statistics
    float absmax = 0.f, maxnorm2 = 0.f, sx = 1.f;
    bool nonfinite = false, corpus_int_unscaled = false, corpus_fp16_exact = false, scan_ok = false;
    // row-block ingestion: two pinned staging buffers (upload_rows)
    void *pin[2] = {nullptr, nullptr};
    size_t pin_bytes = 0;
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    int upload_block_mb = 0;                 // option "upload_block_mb": staging block size (0 = default 64 MiB)
    int64_t last_upload_blocks = 0;          // blocks of the last host upload (vdb_stats: upload_blocks)
    // options
    int force_path = 0, timing = 0, list_cap = 0, scan_variant = 0, select_variant = 0, spc_override = 0, kloop_qgroup = 0;
    int layout_override = 0;                 // option "panel_layout": 1 = keep 32-row tiles for D > 128 (A/B runs)
    bool small_clear_pending = false;        // search_device_impl left the clearing of ws.small to the first batch (serving-shaped calls)
    bool small_is_clean = false;             // ws.small was cleared for this call and no batch has used it yet
    bool small_preset = false;               // coarse quantizer of an IVF index: the parent has just cleared ws.small (it lives in
                                             // the parent's ivf_zero buffer) -- the next search_device_impl skips its own memset
    int64_t info_valid_nq = -1;              // queries whose statistics the last search_batch left in batch_info(ws) (-1: none)
    bool tile16 = false;                     // panels in the p16 layout (16-row tiles, 1024-row spans, 4 bins per span)
    // option "stream_panels" (D > 128, takes effect at the next add): the fp16 panels are NOT kept -- every search converts
    // the float32 rows slab by slab into one scratch slab and scans it (search_flat.inc).  Halves the footprint of a
    // non-fp16-exact corpus (the float32 rows must stay for the exact refine) at the price of one conversion pass per batch.
    int no_fused_stats = 0;                  // option "fused_stats" = 0 (A/B): small batches keep the separate statistics dispatch
    int stream_panels_opt = 0;
    int64_t stream_slab_rows = 0;            // option "stream_slab_rows" (0 = default)
    bool panels_streamed = false;
    DevBuf slab;                             // the scratch slab of a streamed index
    bool set_only = false;                   // coarse quantizer of an IVF index: callers use the SET of the k nearest rows,
                                             // not their order or distances (dense.hpp, DenseSelectArgs.set_only)
    // per-search
    Workspace ws;
    vdb_stats_t last{};
    // timing mode: HIP-event pairs recorded on the search stream around the dominant kernel and the whole
    // device pipeline of every search since timing was switched on (read back by vdb_stats)
    std::vector<hipEvent_t> ev_scan, ev_total;   // [2*i], [2*i+1] = start, stop
    size_t ev_used = 0;
    // ivf (flat handles leave these empty)
    int nlist = 0, nprobe = 1;
    bool ivf_built = false;
    vdb_index_s *coarse = nullptr;           // flat index over the centroids (same metric)
    std::vector<float> ivf_centroids;        // host copy [nlist][dim]
    std::vector<int64_t> ivf_offsets_host;   // [nlist+1]
    std::vector<int32_t> ivf_list_of_row;    // [N] list of every indexed row (original order)
    DevBuf ivf_offsets, ivf_ids, ivf_probe_d, ivf_probe_i;
    // list-major MFMA scan (D <= 128): panel space = lists padded to whole 512-row spans
    bool ivf_mfma_ok = false, ivf_last_mfma = false;
    size_t dbg_words = 0;                    // scan_variant 6: words of per-wave stamps left in ws.dense
    int64_t ivf_pspans = 0;
    int ivf_max_pspans = 0;
    int ivf_span_rows = kSpanRows;           // rows per panel span: 512 (32-row tiles, D <= 128) or 16 * ivf_tps (p16, D > 128)
    int ivf_tps = 0, ivf_tps_override = 0;   // p16 tiles per span of the IVF panel space (16 / 64); option "ivf_tps"
    DevBuf ivf_list_pspan0, ivf_span_row0, ivf_span_valid;
    // everything an IVF search needs zeroed, in ONE buffer cleared by ONE memset per batch (each memset is a ~4 us dispatch
    // of its own): [coarse quantizer's ws.small | this handle's ws.small | per-list counts | cursors | slot -> query map |
    // arrival counters of the flagged-query pass].  Both ws.small are views into it (DevBuf::borrow).
    DevBuf ivf_zero, ivf_slot_off, ivf_list_item0,
        ivf_item_list, ivf_item_slot0, ivf_item_bin0, ivf_plan, ivf_slot_of;
};

namespace {

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Entry points run on the handle's GPU and leave the caller's current device as they found it.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) VDB_HIP(hipSetDevice(dev));
        else prev = -1;
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
#define set_device(dev) DeviceGuard device_guard__(dev)

constexpr size_t kMaxTimedCalls = 1024;

// returns the slot of this call, or -1 when timing is off / the ring is full
long timing_begin(vdb_index_s *h, hipStream_t st);
void timing_mark(vdb_index_s *h, long slot, int which, hipStream_t st);

int kpl_for(int k) {
    int kpl = 1;
    while (kpl * 64 < k) kpl *= 2;
    return kpl;
}

#define DISPATCH_KPL(kpl, ...)                                    \
    switch (kpl) {                                                \
        case 1: { constexpr int KPL = 1; __VA_ARGS__; } break;           \
        case 2: { constexpr int KPL = 2; __VA_ARGS__; } break;           \
        case 4: { constexpr int KPL = 4; __VA_ARGS__; } break;           \
        case 8: { constexpr int KPL = 8; __VA_ARGS__; } break;           \
        case 16: { constexpr int KPL = 16; __VA_ARGS__; } break;         \
        case 32: { constexpr int KPL = 32; __VA_ARGS__; } break;         \
        default: throw Error(VDB_ERR_UNSUPPORTED, "k too large"); \
    }

void launch_refine_full(const RefineFullArgs &a, int64_t max_units, hipStream_t st) {
    int64_t blocks = (max_units + 3) / 4;
    blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, 8192));
    const int kpl = kpl_for(a.c.k);
    DISPATCH_KPL(kpl, (refine_full_kernel<KPL><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(a)));
    VDB_HIP(hipGetLastError());
}

// query-blocked exhaustive scan (4 queries per wave): max_units = ceil(count / 4) * S
constexpr int kRefineQB = 4;
void launch_refine_full_blocked(const RefineFullArgs &a, int64_t max_units, hipStream_t st) {
    int64_t blocks = (max_units + 3) / 4;
    blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, 8192));
    if (kpl_for(a.c.k) == 1)
        refine_full_blocked_kernel<1, kRefineQB><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(a);
    else
        refine_full_blocked_kernel<2, kRefineQB><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(a);
    VDB_HIP(hipGetLastError());
}

void launch_merge(const MergeArgs &a, int64_t max_slots, hipStream_t st) {
    int64_t blocks = (max_slots + 3) / 4;
    blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, 4096));
    const int kpl = kpl_for(a.k);
    DISPATCH_KPL(kpl, (merge_kernel<KPL><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(a)));
    VDB_HIP(hipGetLastError());
}

constexpr double kBinBudget = 6.0 * 1024.0 * 1024.0 * 1024.0;   // bytes of level-1 bin arrays per search pass
constexpr int64_t kDenseMaxRows = 15360;   // dense small-corpus path: one query's scores fit the default 64 KiB of LDS

// ---- row-block ingestion ---------------------------------------------------------------------------
// Host rows reach the device in blocks through TWO pinned staging buffers: block b is copied into pinned memory by
// host threads (this is where a memory-mapped corpus is paged in) while block b-1 is still in flight on the copy
// engine, so a 38 GB shard never has more than two blocks of host staging behind it -- the reference keeps its corpora
// as np.memmap for the same reason (src/benchmark/dataset.py:376-471, 1001-1052).  dst rows are D4 floats apart (D4 >=
// D, the tail must already be zero).
constexpr size_t kUploadBlockBytes = (size_t)64 << 20;

void parallel_memcpy(void *dst, const void *src, size_t bytes) {
    const unsigned hw = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    const size_t per = (bytes / hw + 4095) & ~(size_t)4095;
    if (bytes < ((size_t)8 << 20) || hw == 1) {
        memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> th;
    for (size_t off = per; off < bytes; off += per)
        th.emplace_back([=] { memcpy((char *)dst + off, (const char *)src + off, std::min(per, bytes - off)); });
    memcpy(dst, src, std::min(per, bytes));
    for (auto &t : th) t.join();
}

void upload_rows(vdb_index_s *h, float *dst, int D4, const float *src, int64_t n, int D, hipStream_t st) {
    h->last_upload_blocks = 0;
    if (n <= 0) return;
    const size_t row_bytes = (size_t)D * 4;
    const size_t block_bytes = h->upload_block_mb > 0 ? (size_t)h->upload_block_mb << 20 : kUploadBlockBytes;
    const int64_t rows_per_block = std::max<int64_t>(1, (int64_t)(block_bytes / row_bytes));
    const size_t need = (size_t)std::min<int64_t>(rows_per_block, n) * row_bytes;
    if (h->pin_bytes < need) {
        for (int i = 0; i < 2; ++i) {
            if (h->pin[i]) {
                alloc_note("HF", h->pin[i], h->pin_bytes);
                (void)hipHostFree(h->pin[i]);
            }
            h->pin[i] = nullptr;
        }
        h->pin_bytes = 0;
        if (hipHostMalloc(&h->pin[0], need, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc(&h->pin[1], need, hipHostMallocDefault) == hipSuccess) {
            h->pin_bytes = need;
            alloc_note("HA", h->pin[0], need);
            alloc_note("HA", h->pin[1], need);
        } else {                         // no pinned memory to be had: one pageable copy (the HIP runtime stages it itself)
            for (int i = 0; i < 2; ++i) {
                if (h->pin[i]) (void)hipHostFree(h->pin[i]);
                h->pin[i] = nullptr;
            }
            (void)hipGetLastError();
            VDB_HIP(hipMemcpy2DAsync(dst, (size_t)D4 * 4, src, row_bytes, row_bytes, (size_t)n, hipMemcpyHostToDevice, st));
            VDB_HIP(hipStreamSynchronize(st));
            h->last_upload_blocks = 1;
            return;
        }
    }
    for (int i = 0; i < 2; ++i)
        if (!h->pin_ev[i]) VDB_HIP(hipEventCreateWithFlags(&h->pin_ev[i], hipEventDisableTiming));
    int64_t b = 0;
    for (int64_t r0 = 0; r0 < n; r0 += rows_per_block, ++b) {
        const int64_t rows = std::min<int64_t>(rows_per_block, n - r0);
        const int buf = (int)(b & 1);
        if (b >= 2) VDB_HIP(hipEventSynchronize(h->pin_ev[buf]));       // the copy that last used this buffer is done
        parallel_memcpy(h->pin[buf], src + (size_t)r0 * D, (size_t)rows * row_bytes);
        VDB_HIP(hipMemcpy2DAsync(dst + (size_t)r0 * D4, (size_t)D4 * 4, h->pin[buf], row_bytes, row_bytes, (size_t)rows,
                                 hipMemcpyHostToDevice, st));
        VDB_HIP(hipEventRecord(h->pin_ev[buf], st));
    }
    VDB_HIP(hipStreamSynchronize(st));
    h->last_upload_blocks = b;
}

// ---- index build ---------------------------------------------------------------------------------
// exact row norms + corpus statistics of h->x32 (N rows) -> scales of the fp16 scan copy
void index_stats(vdb_index_s *h, hipStream_t st) {
    const int64_t n = h->N;
    h->xnorm2.reserve((size_t)n * sizeof(float));
    h->stats.reserve(sizeof(IndexStats));
    VDB_HIP(hipMemsetAsync(h->stats.p, 0, sizeof(IndexStats), st));
    corpus_stats_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(h->x32.as<float>(), n, h->D4,
                                                                                h->xnorm2.as<float>(),
                                                                                h->stats.as<IndexStats>());
    VDB_HIP(hipGetLastError());
    IndexStats hs;
    VDB_HIP(hipMemcpyAsync(&hs, h->stats.p, sizeof(hs), hipMemcpyDeviceToHost, st));
    VDB_HIP(hipStreamSynchronize(st));
    memcpy(&h->absmax, &hs.absmax_bits, 4);
    memcpy(&h->maxnorm2, &hs.maxnorm2_bits, 4);
    h->nonfinite = hs.nonfinite != 0;
    h->corpus_int_unscaled = !hs.not_integer && !h->nonfinite && h->absmax <= 2048.f;
    h->i8_cx = !hs.not_u8 ? 128 : 0;                        // u8 window first (SIFT), else s8
    h->i8_ok = !h->nonfinite && (!hs.not_u8 || !hs.not_s8) && h->dim <= 128;
    h->sx = 1.f;
    if (!h->corpus_int_unscaled && h->absmax > 0.f && !h->nonfinite) {
        int e;
        frexpf(h->absmax, &e);
        h->sx = ldexpf(1.f, 14 - e);  // absmax*sx in [8192, 16384)
    }
}

// row-major int8 copy of h->x32 (byte-valued corpora only) for the list refine
void build_rows_i8(vdb_index_s *h, hipStream_t st) {
    h->rows8_pitch = h->i8_ks * 32;                        // = the pitch of the int8 query rows (64 or 128 bytes)
    h->rows8.reserve((size_t)h->N * h->rows8_pitch);
    const int64_t words = h->N * (h->rows8_pitch / 4);
    build_rows_i8_kernel<<<dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st>>>(
        h->x32.as<float>(), h->N, h->dim, h->D4, h->rows8_pitch, h->i8_cx, h->rows8.as<signed char>());
}

void graph_reset(vdb_index_s *h);

// layout "x16" for the scan copies of a D <= 128 flat index, unless an option asks for what only the 32-row kernels have
// (quads as the candidate group, the A/B schedules and ablations behind scan_variant, panel_layout 1)
bool x16_wanted(const vdb_index_s *h) {
    return h->ksteps <= kMaxKSteps && h->flat_shape_opt != 32 && h->layout_override != 1 && h->f16_group == 8 && h->i8_group == 8 &&
           h->scan_variant == 0;
}

// everything derived from the h->N rows in h->x32: statistics, scan copies, biases
void build_derived(vdb_index_s *h, hipStream_t st) {
    const int D = h->dim, D4 = h->D4;
    const int64_t n = h->N;
    // D > 128 (K-loop scan): p16 panels for v_mfma_f32_16x16x32_f16; D <= 128: 32-row tiles (scan_kernel, dense path)
    // (panel_layout 2 = p16 for D <= 128 too, when the corpus is too large for the dense small-corpus kernel)
    // (D > 128 with at most 2048 rows -- an IVF coarse quantizer over embeddings: 32-row tiles, served by the dense path's K-loop
    //  scores + register select instead of the float64 exhaustive kernel, search_flat.inc)
    constexpr int64_t kDenseRegRows = 2048;
    h->tile16 = h->layout_override != 1 && ((h->ksteps > kMaxKSteps && (n > kDenseRegRows || h->layout_override == 2)) ||
                                            (h->layout_override == 2 && n > kDenseMaxRows));
    const int64_t span_rows = h->tile16 ? kSpanRows16 : kSpanRows;
    h->Npad = (n + span_rows - 1) / span_rows * span_rows;
    h->x16 = x16_wanted(h) && !h->tile16 && h->Npad > kDenseMaxRows;
    h->scan_ok = false;
    if (n == 0) {
        h->built = true;
        return;
    }
    index_stats(h, st);
    const bool dims_ok = D <= 4096;
    if (dims_ok && !h->nonfinite) {
        const int64_t ntiles = h->Npad / (h->tile16 ? kTileRows16 : kTileRows);
        const int ksl = h->tile16 ? h->ksteps / 2 : h->ksteps;          // k-steps of the layout (32 or 16 dims)
        h->panels_streamed = h->tile16 && h->stream_panels_opt != 0;
        if (h->panels_streamed) h->panels.release();
        else h->panels.reserve((size_t)ntiles * ksl * 64 * sizeof(half8));
        const int64_t threads = ntiles * ksl * 64;
        if (h->tile16)    // (streamed: the pass only takes the fp16-exactness flag)
            build_panels16_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st>>>(h->x32.as<float>(), n, D, D4, ksl, ntiles, h->sx, h->panels_streamed ? nullptr : h->panels.as<half8>(), h->stats.as<IndexStats>());
        else
            build_panels_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st>>>(h->x32.as<float>(), n, D, D4, h->ksteps, ntiles, h->sx, h->panels.as<half8>(), h->stats.as<IndexStats>(), h->x16 ? 1 : 0);
        VDB_HIP(hipGetLastError());
        h->bias.reserve((size_t)h->Npad * sizeof(float));
        build_bias_kernel<<<dim3((unsigned)((h->Npad + 255) / 256)), dim3(256), 0, st>>>(h->xnorm2.as<float>(), n, h->Npad, h->metric, h->bias.as<float>());
        VDB_HIP(hipGetLastError());
        IndexStats hs;
        VDB_HIP(hipMemcpyAsync(&hs, h->stats.p, sizeof(hs), hipMemcpyDeviceToHost, st));
        VDB_HIP(hipStreamSynchronize(st));
        h->corpus_fp16_exact = hs.not_fp16_exact == 0;
        h->scan_ok = true;
        h->i8_ok = h->i8_ok && !h->tile16;
        if (h->i8_ok) {
            h->i8_ks = D <= 64 ? 2 : 4;
            h->panels8.reserve((size_t)ntiles * h->i8_ks * 64 * sizeof(int4v));
            const int64_t t8 = ntiles * h->i8_ks * 64;
            build_panels_i8_kernel<<<dim3((unsigned)((t8 + 255) / 256)), dim3(256), 0, st>>>(
                h->x32.as<float>(), n, D, D4, h->i8_ks, ntiles, h->i8_cx, h->panels8.as<int4v>(), 0, h->x16 ? 1 : 0);
            h->bias8.reserve((size_t)2 * h->Npad * sizeof(int32_t));
            h->rowstat8.reserve((size_t)n * 2 * sizeof(int));
            build_bias_i8_kernel<<<dim3((unsigned)((h->Npad + 255) / 256)), dim3(256), 0, st>>>(
                h->x32.as<float>(), n, h->Npad, D, D4, h->metric, h->bias8.as<int32_t>(), h->rowstat8.as<int>());
            build_rows_i8(h, st);
            VDB_HIP(hipGetLastError());
            VDB_HIP(hipStreamSynchronize(st));
        }
    } else {
        h->i8_ok = false;
    }
    h->built = true;
}

// rows [row0, row0 + n) of h->x32 from host or device memory (h->x32 already holds room for them)
void ingest_rows(vdb_index_s *h, int64_t row0, const float *x_dev_or_host, bool on_device, int64_t n, hipStream_t st) {
    const int D = h->dim, D4 = h->D4;
    float *dst = h->x32.as<float>() + (size_t)row0 * D4;
    if (D4 != D) VDB_HIP(hipMemsetAsync(dst, 0, (size_t)n * D4 * sizeof(float), st));
    if (on_device)
        VDB_HIP(hipMemcpy2DAsync(dst, (size_t)D4 * 4, x_dev_or_host, (size_t)D * 4, (size_t)D * 4, (size_t)n,
                                 hipMemcpyDeviceToDevice, st));
    else
        upload_rows(h, dst, D4, x_dev_or_host, n, D, st);
}

// ---- int8-only build (option "int8_only") ----------------------------------------------------------------------------------
// The float32 rows are never resident as a whole: they pass through in blocks (a window of the caller's device array when its
// rows are 16-byte multiples, else a padded / uploaded temporary of <= 4M rows), and every block leaves only its int8 rows, row
// statistics, int8 panels and accumulator inits behind.  The byte window (u8: x - 128, s8: x) is taken from the first block
// and checked on the flags of all of them; if the other window fits the whole corpus the build runs once more with it.
// Returns false when the corpus is not byte-valued (the caller then builds the default index).
constexpr int64_t kInt8OnlyMinRows = 32768;
bool build_int8_only(vdb_index_s *h, const float *x, bool on_device, int64_t n, hipStream_t st) {
    const int D = h->dim, D4 = h->D4;
    DevBuf *gone[] = {&h->x32, &h->panels, &h->slab};
    for (auto b : gone) b->release();
    h->Npad = (n + kSpanRows - 1) / kSpanRows * kSpanRows;
    h->i8_ks = D <= 64 ? 2 : 4;
    h->x16 = x16_wanted(h);                 // (more than 32 768 rows: never the dense path's)
    h->rows8_pitch = h->i8_ks * 32;
    const int64_t ntiles = h->Npad / kTileRows;
    h->rows8.reserve_exact((size_t)n * h->rows8_pitch);
    h->rowstat8.reserve_exact((size_t)n * 2 * sizeof(int));
    h->xnorm2.reserve((size_t)n * sizeof(float));
    h->stats.reserve(sizeof(IndexStats));
    h->bias8.reserve_exact((size_t)2 * h->Npad * sizeof(int32_t));
    h->panels8.reserve_exact((size_t)ntiles * h->i8_ks * 64 * sizeof(int4v));
    h->bias.reserve_exact((size_t)h->Npad * sizeof(float));
    const bool direct = on_device && D4 == D && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const int64_t block_want = h->int8_block_rows > 0 ? (h->int8_block_rows + kSpanRows - 1) / kSpanRows * kSpanRows : (int64_t)4 << 20;
    const int64_t block_rows = std::min<int64_t>((n + kSpanRows - 1) / kSpanRows * kSpanRows, block_want);
    ScopedDevBuf tmp;
    if (!direct) tmp.reserve((size_t)std::min<int64_t>(block_rows, n) * D4 * sizeof(float));
    IndexStats hs{};
    int cx = -1;
    for (int attempt = 0; attempt < 2; ++attempt) {
        VDB_HIP(hipMemsetAsync(h->stats.p, 0, sizeof(IndexStats), st));
        for (int64_t r0 = 0; r0 < n; r0 += block_rows) {
            const int64_t r1 = std::min<int64_t>(n, r0 + block_rows), nb = r1 - r0;
            const bool last = r1 == n;
            const float *blk;
            if (direct) {
                blk = x + (size_t)r0 * D;
            } else {
                float *dst = tmp.as<float>();
                if (D4 != D) VDB_HIP(hipMemsetAsync(dst, 0, (size_t)nb * D4 * sizeof(float), st));
                if (on_device)
                    VDB_HIP(hipMemcpy2DAsync(dst, (size_t)D4 * 4, x + (size_t)r0 * D, (size_t)D * 4, (size_t)D * 4, (size_t)nb,
                                             hipMemcpyDeviceToDevice, st));
                else
                    upload_rows(h, dst, D4, x + (size_t)r0 * D, nb, D, st);
                blk = dst;
            }
            corpus_stats_kernel<<<dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st>>>(blk, nb, D4, h->xnorm2.as<float>() + r0,
                                                                                        h->stats.as<IndexStats>());
            if (cx < 0) {           // the window: from the first block's flags
                VDB_HIP(hipMemcpyAsync(&hs, h->stats.p, sizeof(hs), hipMemcpyDeviceToHost, st));
                VDB_HIP(hipStreamSynchronize(st));
                if (hs.nonfinite || (hs.not_u8 && hs.not_s8)) return false;
                cx = !hs.not_u8 ? 128 : 0;
            }
            const int64_t words = nb * (h->rows8_pitch / 4);
            build_rows_i8_kernel<<<dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st>>>(
                blk, nb, D, D4, h->rows8_pitch, cx, h->rows8.as<signed char>() + (size_t)r0 * h->rows8_pitch);
            // (panels and accumulator inits index rows globally: base pointer moved back by the rows in front of the block)
            const float *fake = blk - (size_t)r0 * D4;
            const int64_t row_end = last ? h->Npad : r1;
            const int64_t t0 = r0 / kTileRows, nt = (row_end - r0) / kTileRows;
            build_panels_i8_kernel<<<dim3((unsigned)((nt * h->i8_ks * 64 + 255) / 256)), dim3(256), 0, st>>>(
                fake, r1, D, D4, h->i8_ks, nt, cx, h->panels8.as<int4v>(), t0, h->x16 ? 1 : 0);
            build_bias_i8_kernel<<<dim3((unsigned)((row_end - r0 + 255) / 256)), dim3(256), 0, st>>>(
                fake, r1, h->Npad, D, D4, h->metric, h->bias8.as<int32_t>(), h->rowstat8.as<int>(), r0, row_end);
            VDB_HIP(hipGetLastError());
            if (!direct) VDB_HIP(hipStreamSynchronize(st));     // (the temporary is refilled by the next block)
        }
        VDB_HIP(hipMemcpyAsync(&hs, h->stats.p, sizeof(hs), hipMemcpyDeviceToHost, st));
        VDB_HIP(hipStreamSynchronize(st));
        if (hs.nonfinite) return false;
        const bool fits = cx == 128 ? !hs.not_u8 : !hs.not_s8;
        if (fits) break;
        const bool other = cx == 128 ? !hs.not_s8 : !hs.not_u8;
        if (!other || attempt == 1) return false;
        cx = cx == 128 ? 0 : 128;       // a later block left the first block's window but the whole corpus fits the other one
    }
    memcpy(&h->absmax, &hs.absmax_bits, 4);
    memcpy(&h->maxnorm2, &hs.maxnorm2_bits, 4);
    h->nonfinite = false;
    h->corpus_int_unscaled = true;      // integers 0..255 / -128..127: stored unscaled, exact in fp16
    h->corpus_fp16_exact = true;
    h->sx = 1.f;
    h->i8_cx = cx;
    h->i8_ok = true;
    h->tile16 = false;
    h->panels_streamed = false;
    build_bias_kernel<<<dim3((unsigned)((h->Npad + 255) / 256)), dim3(256), 0, st>>>(h->xnorm2.as<float>(), n, h->Npad, h->metric,
                                                                                    h->bias.as<float>());
    VDB_HIP(hipGetLastError());
    VDB_HIP(hipStreamSynchronize(st));
    h->xnorm2.release();                // (the float norms only fed `bias`)
    h->int8_only = true;
    h->scan_ok = true;
    h->built = true;
    return true;
}

// the index holds exactly these n rows afterwards (whatever it held before)
void build_index(vdb_index_s *h, const float *x_dev_or_host, bool on_device, int64_t n, int64_t id_base,
                 hipStream_t st) {
    if (n < 0) throw Error(VDB_ERR_INVALID, "negative row count");
    graph_reset(h);
    if (n > 2147483647ll - 1024) throw Error(VDB_ERR_UNSUPPORTED, "more than 2^31 rows per shard");
    h->built = false;
    h->ivf_built = false;              // (rows filed by vdb_ivf_add sat in list order under a CSR that no longer describes x32)
    h->ivf_list_of_row.clear();
    h->N = n;
    h->id_base = id_base;
    h->int8_only = false;
    if (h->int8_only_opt && h->dim <= 128 && n > kInt8OnlyMinRows && !h->coarse) {
        bool ok = false;
        try {
            ok = build_int8_only(h, x_dev_or_host, on_device, n, st);
        } catch (...) {
            h->N = 0;
            h->built = false;
            h->scan_ok = false;
            h->int8_only = false;
            throw;
        }
        if (ok) return;
        h->int8_only = false;          // not byte-valued: the default layout (vdb_stats: has_i8_copy says which one it is)
        DevBuf *i8[] = {&h->rows8, &h->rowstat8, &h->bias8, &h->panels8};
        for (auto b : i8) b->release();
    }
    try {
        if (n > 0) {
            h->x32.reserve((size_t)n * h->D4 * sizeof(float));
            ingest_rows(h, 0, x_dev_or_host, on_device, n, st);
        }
        build_derived(h, st);
    } catch (...) {                    // a failed build leaves an EMPTY index (the next add starts over; a search says "not built")
        h->N = 0;
        h->built = false;
        h->scan_ok = false;
        throw;
    }
}

void require_same_id_base(const vdb_index_s *h, int64_t id_base) {
    if (id_base != h->id_base)
        throw Error(VDB_ERR_INVALID, "add appends to the " + std::to_string(h->N) + " rows of this index, whose id base is " +
                                         std::to_string(h->id_base) + " (row i of the index has id base + i): pass the same "
                                         "id_base, or call vdb_reset first; got " + std::to_string(id_base));
}

// vdb_add / vdb_add_device: APPEND, as faiss.Index.add does (the first add of an empty index is build_index)
void append_rows(vdb_index_s *h, const float *x_dev_or_host, bool on_device, int64_t n, int64_t id_base, hipStream_t st) {
    if (h->N == 0 || h->ivf_built) {      // (rows filed by vdb_ivf_add are replaced: they sit in list order, not insertion order)
        build_index(h, x_dev_or_host, on_device, n, id_base, st);
        return;
    }
    if (n < 0) throw Error(VDB_ERR_INVALID, "negative row count");
    require_same_id_base(h, id_base);
    if (n == 0) return;
    if (h->int8_only)
        throw Error(VDB_ERR_UNSUPPORTED, "an int8_only index keeps no float32 rows to re-derive its scan copies from: it is built by "
                                         "ONE add (vdb_reset, then add everything)");
    if (h->N + n > 2147483647ll - 1024) throw Error(VDB_ERR_UNSUPPORTED, "more than 2^31 rows per shard");
    graph_reset(h);
    VDB_HIP(hipDeviceSynchronize());                        // (searches of the rows about to move may still run)
    const int64_t N0 = h->N;
    // the scan copies are re-derived from the float32 rows anyway: free them first, so that the peak is old rows + grown rows
    // (not that plus the copies), and on failure rebuild them from the old rows -- the append is atomic (N, ids and results
    // as before the call)
    DevBuf *derived[] = {&h->panels, &h->panels8, &h->rows8, &h->rowstat8, &h->bias, &h->bias8, &h->slab};
    for (auto b : derived) b->release();
    h->built = false;
    try {
        h->x32.grow((size_t)(N0 + n) * h->D4 * sizeof(float), (size_t)N0 * h->D4 * sizeof(float));
        ingest_rows(h, N0, x_dev_or_host, on_device, n, st);
    } catch (...) {
        (void)hipGetLastError();
        h->N = N0;
        try {
            build_derived(h, st);
        } catch (...) {                // not even the old copies fit any more: the index is emptied, loudly
            h->N = 0;
            h->built = false;
            h->scan_ok = false;
        }
        throw;
    }
    h->N = N0 + n;
    build_derived(h, st);                                   // (scan copies are rebuilt from the float32 rows: 0.3 s per 12.5M x 768)
}

#include "search_flat.inc"   // scan geometry, launchers, search_batch, graph replay, search_device_impl

template <class F>
int guarded(F &&f) {
    try {
        f();
        return VDB_OK;
    } catch (const Error &e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return VDB_ERR_INVALID;
    } catch (...) {
        g_last_error = "unknown error";
        return VDB_ERR_INVALID;
    }
}

vdb_index_s *check(vdb_handle h) {
    if (!h) throw Error(VDB_ERR_INVALID, "null handle");
    return h;
}

// multi-device handles (multi.inc)
void multi_destroy(vdb_index_s *m);
void multi_reset(vdb_index_s *m);
void multi_add(vdb_index_s *m, const float *x, bool on_device, int64_t n, int64_t id_base, hipStream_t user_stream, bool ivf,
               const int32_t *given);
void multi_search(vdb_index_s *m, const float *q, bool device_api, int64_t nq, int k, float *D, int64_t *I, double *PK,
                  int64_t *PI, hipStream_t user_stream, bool ivf);
void multi_reserve(vdb_index_s *m, int64_t nq, int k);
void multi_stats(vdb_index_s *m, vdb_stats_t *out);
void multi_set_option(vdb_index_s *m, const char *key, double value);
void multi_set_centroids(vdb_index_s *m, const float *c_host, int nlist);
void multi_train(vdb_index_s *m, int nlist, const float *x_host, int64_t n, int niter, uint64_t seed, int mppc);
void multi_get_assignment(vdb_index_s *m, int32_t *out);
void multi_rerank(vdb_index_s *m, const float *q, bool device_api, int64_t nq, const int64_t *cand, int ncand, int k, float *D,
                  int64_t *I, hipStream_t user_stream);
vdb_index_s *multi_first_shard(vdb_index_s *m);
void multi_for_each_shard(vdb_index_s *m, const std::function<void(vdb_index_s *)> &f);
[[noreturn]] inline void multi_unsupported(const char *what) {
    throw Error(VDB_ERR_UNSUPPORTED, std::string(what) + " is not available on a multi-device index (vdb_create_multi)");
}

}  // namespace

// =====================================================================================================
extern "C" {

int vdb_abi_version(void) { return VDB_ABI_VERSION; }

const char *vdb_last_error(void) { return g_last_error.c_str(); }

int vdb_device_count(int *count) {
    return guarded([&] {
        if (!count) throw Error(VDB_ERR_INVALID, "null pointer");
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess) n = 0;
        *count = n;
    });
}

int vdb_create(int dim, int metric, int device, vdb_handle *out) {
    return guarded([&] {
        if (!out) throw Error(VDB_ERR_INVALID, "null output handle");
        if (dim < 1 || dim > 65536) throw Error(VDB_ERR_INVALID, "dimension must be in [1, 65536]");
        if (metric != VDB_METRIC_L2 && metric != VDB_METRIC_IP) throw Error(VDB_ERR_INVALID, "unknown metric");
        int n = 0;
        VDB_HIP(hipGetDeviceCount(&n));
        if (device < 0 || device >= n) throw Error(VDB_ERR_INVALID, "no such GPU: " + std::to_string(device));
        set_device(device);
        hipDeviceProp_t prop;
        VDB_HIP(hipGetDeviceProperties(&prop, device));
        if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
            throw Error(VDB_ERR_UNSUPPORTED, std::string("libvdbhip is built for gfx950 (MI355X); found ") +
                                                 prop.gcnArchName);
        auto *h = new vdb_index_s();
        h->device = device;
        h->dim = dim;
        h->D4 = (dim + 3) / 4 * 4;
        h->ksteps = dim <= 64 ? 4 : (dim <= 128 ? 8 : (dim + 63) / 64 * 4);  // D > 128: K-loop kernel, 64-dim steps
        h->metric = metric;
        *out = h;
    });
}

int vdb_destroy(vdb_handle h) {
    return guarded([&] {
        if (!h) return;
        if (h->multi) {
            multi_destroy(h);
            delete h;
            return;
        }
        set_device(h->device);
        (void)hipDeviceSynchronize();
        DevBuf *all[] = {&h->x32, &h->xnorm2, &h->panels, &h->slab, &h->bias, &h->stats, &h->panels8, &h->bias8, &h->rows8, &h->rowstat8, &h->ivf_offsets, &h->ivf_ids,
                         &h->ivf_probe_d, &h->ivf_probe_i, &h->ivf_list_pspan0, &h->ivf_span_row0, &h->ivf_span_valid,
                         &h->ivf_zero, &h->ivf_slot_off, &h->ivf_list_item0, &h->ivf_item_list,
                         &h->ivf_item_slot0, &h->ivf_item_bin0, &h->ivf_plan, &h->ivf_slot_of};
        for (auto b : all) b->release();
        graph_reset(h);
        if (h->graph_ev) (void)hipEventDestroy(h->graph_ev);
        if (h->coarse) (void)vdb_destroy(h->coarse);
        h->ws.release();
        for (int i = 0; i < 2; ++i) {
            if (h->pin[i]) {
                alloc_note("HF", h->pin[i], h->pin_bytes);
                (void)hipHostFree(h->pin[i]);
            }
            if (h->pin_ev[i]) (void)hipEventDestroy(h->pin_ev[i]);
        }
        for (auto e : h->ev_scan) (void)hipEventDestroy(e);
        for (auto e : h->ev_total) (void)hipEventDestroy(e);
        delete h;
    });
}

int vdb_add(vdb_handle hh, const float *x_host, int64_t n, int64_t id_base) {
    return guarded([&] {
        auto *h = check(hh);
        if (n > 0 && !x_host) throw Error(VDB_ERR_INVALID, "null corpus pointer");
        if (h->multi) return multi_add(h, x_host, false, n, id_base, nullptr, false, nullptr);
        set_device(h->device);
        append_rows(h, x_host, false, n, id_base, nullptr);
    });
}

int vdb_reset(vdb_handle hh) {
    return guarded([&] {
        auto *h = check(hh);
        if (h->multi) return multi_reset(h);
        set_device(h->device);
        VDB_HIP(hipDeviceSynchronize());
        graph_reset(h);
        h->N = 0;
        h->built = false;
        h->scan_ok = false;
        h->ivf_built = false;
        h->int8_only = false;
        h->ivf_list_of_row.clear();
        h->ivf_offsets_host.clear();
        // faiss.Index.reset frees its storage: so do we (rows, scan copies, CSR arrays; the workspace and an IVF index's
        // centroids stay) -- a caller that resets a 38 GB shard to load another corpus gets the memory back
        DevBuf *rows[] = {&h->x32, &h->xnorm2, &h->panels, &h->slab, &h->bias, &h->panels8, &h->bias8, &h->rows8, &h->rowstat8,
                          &h->ivf_offsets, &h->ivf_ids, &h->ivf_list_pspan0, &h->ivf_span_row0, &h->ivf_span_valid};
        for (auto b : rows) b->release();
    });
}

int vdb_add_device(vdb_handle hh, const float *x_dev, int64_t n, int64_t id_base, void *stream) {
    return guarded([&] {
        auto *h = check(hh);
        if (n > 0 && !x_dev) throw Error(VDB_ERR_INVALID, "null corpus pointer");
        if (h->multi) return multi_add(h, x_dev, true, n, id_base, as_stream(stream), false, nullptr);
        set_device(h->device);
        append_rows(h, x_dev, true, n, id_base, as_stream(stream));
    });
}

int vdb_search(vdb_handle hh, const float *q_host, int64_t nq, int k, float *D, int64_t *I) {
    return guarded([&] {
        auto *h = check(hh);
        if (!h->built) throw Error(VDB_ERR_STATE, "Index has not been built yet.");
        if (nq > 0 && (!q_host || !D || !I)) throw Error(VDB_ERR_INVALID, "null pointer");
        if (k < 1 || k > 2048) throw Error(VDB_ERR_INVALID, "k must be in [1, 2048]");
        if (nq <= 0) {
            if (nq < 0) throw Error(VDB_ERR_INVALID, "negative query count");
            return;
        }
        if (h->multi) return multi_search(h, q_host, false, nq, k, D, I, nullptr, nullptr, nullptr, false);
        set_device(h->device);
        Workspace &ws = h->ws;
        ws.stage_q.reserve((size_t)nq * h->dim * sizeof(float));
        ws.stage_d.reserve((size_t)nq * k * sizeof(float));
        ws.stage_i.reserve((size_t)nq * k * sizeof(int64_t));
        hipStream_t st = nullptr;
        VDB_HIP(hipMemcpyAsync(ws.stage_q.p, q_host, (size_t)nq * h->dim * sizeof(float), hipMemcpyHostToDevice, st));
        search_device_impl(h, ws.stage_q.as<float>(), nq, k, ws.stage_d.as<float>(), ws.stage_i.as<int64_t>(), nullptr,
                           nullptr, st);
        VDB_HIP(hipMemcpyAsync(D, ws.stage_d.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, st));
        VDB_HIP(hipMemcpyAsync(I, ws.stage_i.p, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        VDB_HIP(hipStreamSynchronize(st));
    });
}

int vdb_search_device(vdb_handle hh, const float *q_dev, int64_t nq, int k, float *D_dev, int64_t *I_dev,
                      void *stream) {
    return guarded([&] {
        auto *h = check(hh);
        if (nq > 0 && (!D_dev || !I_dev)) throw Error(VDB_ERR_INVALID, "null output pointer");
        if (h->multi) return multi_search(h, q_dev, true, nq, k, D_dev, I_dev, nullptr, nullptr, as_stream(stream), false);
        set_device(h->device);
        vdb_index_s::GraphKey key;
        key.q = q_dev; key.o1 = D_dev; key.o2 = I_dev; key.nq = nq; key.k = k; key.kind = 1; key.st = as_stream(stream);
        graph_or_run(h, key, [&] { search_device_impl(h, q_dev, nq, k, D_dev, I_dev, nullptr, nullptr, as_stream(stream)); });
    });
}

int vdb_search_partial_device(vdb_handle hh, const float *q_dev, int64_t nq, int k, double *keys_dev,
                              int64_t *ids_dev, void *stream) {
    return guarded([&] {
        auto *h = check(hh);
        if (nq > 0 && (!keys_dev || !ids_dev)) throw Error(VDB_ERR_INVALID, "null output pointer");
        if (h->multi) return multi_search(h, q_dev, true, nq, k, nullptr, nullptr, keys_dev, ids_dev, as_stream(stream), false);
        set_device(h->device);
        vdb_index_s::GraphKey key;
        key.q = q_dev; key.o1 = keys_dev; key.o2 = ids_dev; key.nq = nq; key.k = k; key.kind = 2; key.st = as_stream(stream);
        graph_or_run(h, key, [&] { search_device_impl(h, q_dev, nq, k, nullptr, nullptr, keys_dev, ids_dev, as_stream(stream)); });
    });
}

int vdb_merge_partials_device(int metric, int device, const double *keys_dev, const int64_t *ids_dev, int nparts,
                              int64_t nq, int k, float *D_dev, int64_t *I_dev, void *stream) {
    return guarded([&] {
        if (metric != VDB_METRIC_L2 && metric != VDB_METRIC_IP) throw Error(VDB_ERR_INVALID, "unknown metric");
        if (k < 1 || k > 2048) throw Error(VDB_ERR_INVALID, "k must be in [1, 2048]");
        if (nparts < 0 || nq < 0) throw Error(VDB_ERR_INVALID, "negative size");
        if (nq == 0) return;
        if (!D_dev || !I_dev || (nparts > 0 && (!keys_dev || !ids_dev))) throw Error(VDB_ERR_INVALID, "null pointer");
        set_device(device);
        MergeArgs ma{};
        ma.pkeys = keys_dev;
        ma.pids = ids_dev;
        ma.part_stride = nq * k;
        ma.slot_stride = k;
        ma.nparts = nparts;
        ma.k = k;
        ma.metric = metric;
        ma.count = nq;
        ma.D = D_dev;
        ma.I = I_dev;
        launch_merge(ma, nq, as_stream(stream));
    });
}

int vdb_merge_packed_partials_device(int metric, int device, const void *packed_dev, int nparts, int64_t nq, int k,
                                     float *D_dev, int64_t *I_dev, void *stream) {
    return guarded([&] {
        if (metric != VDB_METRIC_L2 && metric != VDB_METRIC_IP) throw Error(VDB_ERR_INVALID, "unknown metric");
        if (k < 1 || k > 2048) throw Error(VDB_ERR_INVALID, "k must be in [1, 2048]");
        if (nparts < 0 || nq < 0) throw Error(VDB_ERR_INVALID, "negative size");
        if (nq == 0) return;
        if (!D_dev || !I_dev || (nparts > 0 && !packed_dev)) throw Error(VDB_ERR_INVALID, "null pointer");
        set_device(device);
        MergeArgs ma{};
        ma.pkeys = reinterpret_cast<const double *>(packed_dev);
        ma.pids = reinterpret_cast<const int64_t *>(packed_dev) + nq * k;
        ma.part_stride = 2 * nq * k;
        ma.slot_stride = k;
        ma.nparts = nparts;
        ma.k = k;
        ma.metric = metric;
        ma.count = nq;
        ma.D = D_dev;
        ma.I = I_dev;
        launch_merge(ma, nq, as_stream(stream));
    });
}

namespace {
// (pk / pi: partial rows -- float64 keys + ids -- instead of (D, I); segs / nseg: a shard of a multi-device index, see RerankArgs)
void rerank_device_impl(vdb_index_s *h, const float *dq, int64_t nq, const int64_t *cand, int ncand, int k, float *D,
                        int64_t *I, hipStream_t st, double *pk = nullptr, int64_t *pi = nullptr, const int64_t *segs = nullptr,
                        int nseg = 0) {
    if (!h->built) throw Error(VDB_ERR_STATE, "Index has not been built yet.");
    if (k < 1 || k > 2048) throw Error(VDB_ERR_INVALID, "k must be in [1, 2048]");
    if (nq < 0 || ncand < 0) throw Error(VDB_ERR_INVALID, "negative size");
    if (nq == 0) return;
    if (!dq || (pk ? !pi : (!D || !I)) || (ncand > 0 && !cand)) throw Error(VDB_ERR_INVALID, "null pointer");
    const float *qpad = dq;
    if (h->D4 != h->dim) {
        h->ws.qpad.reserve((size_t)nq * h->D4 * sizeof(float));
        pad_rows_kernel<<<dim3((unsigned)((nq * h->D4 + 255) / 256)), dim3(256), 0, st>>>(dq, nq, h->dim, h->D4,
                                                                                        h->ws.qpad.as<float>());
        qpad = h->ws.qpad.as<float>();
    }
    RerankArgs a{};
    a.c = RefineCommon{h->int8_only ? nullptr : h->x32.as<float>(), qpad, h->N, h->id_base, h->D4, h->metric, k, nullptr};
    if (h->int8_only) {
        a.c.X8 = h->rows8.as<signed char>();
        a.c.x8_pitch = h->rows8_pitch;
        a.c.cx = h->i8_cx;
    }
    a.nq = nq;
    a.cand = cand;
    a.ncand = ncand;
    a.D = pk ? nullptr : D;
    a.I = pk ? nullptr : I;
    a.pkeys = pk;
    a.pids = pi;
    a.segs = segs;
    a.nseg = nseg;
    const int kpl = kpl_for(k);
    DISPATCH_KPL(kpl, (rerank_kernel<KPL><<<dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st>>>(a)));
    VDB_HIP(hipGetLastError());
}
}  // namespace

int vdb_rerank_device(vdb_handle hh, const float *q_dev, int64_t nq, const int64_t *cand_dev, int ncand, int k,
                      float *D_dev, int64_t *I_dev, void *stream) {
    return guarded([&] {
        auto *h = check(hh);
        if (h->multi) return multi_rerank(h, q_dev, true, nq, cand_dev, ncand, k, D_dev, I_dev, as_stream(stream));
        set_device(h->device);
        rerank_device_impl(h, q_dev, nq, cand_dev, ncand, k, D_dev, I_dev, as_stream(stream));
    });
}

int vdb_rerank(vdb_handle hh, const float *q_host, int64_t nq, const int64_t *cand_host, int ncand, int k, float *D,
               int64_t *I) {
    return guarded([&] {
        auto *h = check(hh);
        if (h->multi) return multi_rerank(h, q_host, false, nq, cand_host, ncand, k, D, I, nullptr);
        if (!h->built) throw Error(VDB_ERR_STATE, "Index has not been built yet.");
        if (nq <= 0) {
            if (nq < 0) throw Error(VDB_ERR_INVALID, "negative query count");
            return;
        }
        if (!q_host || !D || !I || (ncand > 0 && !cand_host) || ncand < 0) throw Error(VDB_ERR_INVALID, "bad argument");
        if (k < 1 || k > 2048) throw Error(VDB_ERR_INVALID, "k must be in [1, 2048]");
        set_device(h->device);
        Workspace &ws = h->ws;
        ScopedDevBuf dc;
        ws.stage_q.reserve((size_t)nq * h->dim * sizeof(float));
        ws.stage_d.reserve((size_t)nq * k * sizeof(float));
        ws.stage_i.reserve((size_t)nq * k * sizeof(int64_t));
        dc.reserve((size_t)nq * std::max(ncand, 1) * sizeof(int64_t));
        hipStream_t st = nullptr;
        VDB_HIP(hipMemcpyAsync(ws.stage_q.p, q_host, (size_t)nq * h->dim * sizeof(float), hipMemcpyHostToDevice, st));
        if (ncand > 0)
            VDB_HIP(hipMemcpyAsync(dc.p, cand_host, (size_t)nq * ncand * sizeof(int64_t), hipMemcpyHostToDevice, st));
        rerank_device_impl(h, ws.stage_q.as<float>(), nq, dc.as<int64_t>(), ncand, k, ws.stage_d.as<float>(),
                           ws.stage_i.as<int64_t>(), st);
        VDB_HIP(hipMemcpyAsync(D, ws.stage_d.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, st));
        VDB_HIP(hipMemcpyAsync(I, ws.stage_i.p, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        VDB_HIP(hipStreamSynchronize(st));
    });
}

int vdb_stats(vdb_handle hh, vdb_stats_t *out) {
    return guarded([&] {
        auto *h = check(hh);
        if (!out) throw Error(VDB_ERR_INVALID, "null pointer");
        if (h->multi) return multi_stats(h, out);
        set_device(h->device);
        vdb_stats_t s = h->last;
        s.ndevices = 1;
        s.scan_shape = (!h->scan_ok || h->ksteps > kMaxKSteps || h->N == 0) ? 0 : h->x16 ? 16 : h->tile16 ? 16 : 32;
        s.ntotal = h->N;
        s.dim = h->dim;
        s.metric = h->metric;
        s.corpus_fp16_exact = h->corpus_fp16_exact ? 1 : 0;
        s.bytes_resident = (int64_t)(h->x32.cap + h->xnorm2.cap + h->panels.cap + h->slab.cap + h->bias.cap + h->stats.cap +
                                     h->panels8.cap + h->bias8.cap + h->rows8.cap + h->rowstat8.cap + h->ws.bytes());
        {   // IVF: the CSR arrays, the per-batch plan buffers and the coarse quantizer's own index and workspace
            const DevBuf *ivf[] = {&h->ivf_offsets, &h->ivf_ids, &h->ivf_probe_d, &h->ivf_probe_i, &h->ivf_list_pspan0,
                                   &h->ivf_span_row0, &h->ivf_span_valid, &h->ivf_zero, &h->ivf_slot_off, &h->ivf_list_item0,
                                   &h->ivf_item_list, &h->ivf_item_slot0, &h->ivf_item_bin0, &h->ivf_plan, &h->ivf_slot_of};
            for (auto b : ivf) s.bytes_resident += (int64_t)b->cap;
            if (h->ivf_zero.cap) s.bytes_resident -= (int64_t)h->ws.small.cap;      // (a view into ivf_zero: counted once)
            if (h->coarse)
                s.bytes_resident += (int64_t)(h->coarse->x32.cap + h->coarse->xnorm2.cap + h->coarse->panels.cap +
                                              h->coarse->bias.cap + h->coarse->ws.bytes() -
                                              (h->coarse->ws.small.borrowed ? h->coarse->ws.small.cap : 0));
        }
        s.has_i8_copy = h->int8_only ? 2 : (h->i8_ok ? 1 : 0);
        s.bytes_workspace = (int64_t)h->ws.bytes();
        s.upload_blocks = h->last_upload_blocks;
        s.graph_replays = h->graph_replays;
        s.last_rows_scanned = 0;
        if (h->last.last_path == VDB_PATH_IVF && h->ivf_last_mfma && h->ivf_plan.p) {   // (of the last batch of the call)
            IvfPlan pl;
            VDB_HIP(hipDeviceSynchronize());
            VDB_HIP(hipMemcpy(&pl, h->ivf_plan.p, sizeof(pl), hipMemcpyDeviceToHost));
            s.last_rows_scanned = (int64_t)pl.rows_scanned;
        }
        s.scan_dtype = 0;
        if (h->i8_ok && h->ws.small.p &&
            (h->last.last_path == VDB_PATH_MFMA_SCAN || (h->last.last_path == VDB_PATH_IVF && h->ivf_last_mfma))) {   // which scan the device chose
            QueryBatchInfo qi;
            VDB_HIP(hipDeviceSynchronize());
            VDB_HIP(hipMemcpy(&qi, batch_info(h->ws), sizeof(qi), hipMemcpyDeviceToHost));
            s.scan_dtype = qi.i8_mode ? 1 : 0;
        }
        s.nlist = h->nlist;
        s.nprobe = h->nprobe;
        s.last_candidates = s.last_rescan_bins = s.last_fallback_queries = 0;
        s.last_scan_ms = s.last_total_ms = s.last_prep_ms = s.last_tail_ms = 0.f;
        if (h->ws.small.p && (h->last.last_path == VDB_PATH_MFMA_SCAN || (h->last.last_path == VDB_PATH_IVF && h->ivf_last_mfma))) {
            std::vector<unsigned char> buf(kSmallBytes);
            VDB_HIP(hipDeviceSynchronize());
            VDB_HIP(hipMemcpy(buf.data(), h->ws.small.p, kSmallBytes, hipMemcpyDeviceToHost));
            int32_t fb;
            unsigned long long c[3] = {0, 0, 0};
            memcpy(&fb, buf.data(), 4);
            for (int sh = 0; sh < kStatShards; ++sh) {          // sharded counters (common.hpp, stat_add)
                unsigned long long v[3];
                memcpy(v, buf.data() + 64 + (size_t)sh * kStatStride * 8, 24);
                c[0] += v[0];
                c[1] += v[1];
                c[2] += v[2];
            }
            fb = (int32_t)c[2];        // (counter 2 accumulates over the batches of a call; fb_count restarts per batch)
            s.last_fallback_queries = fb;
            s.last_candidates = (int64_t)c[0];
            s.last_rescan_bins = (int64_t)c[1];
        }
        if (h->ev_used > 0) {  // averages over every search recorded since timing was switched on
            double scan = 0.0, total = 0.0, prep = 0.0, tail = 0.0;
            for (size_t i = 0; i < h->ev_used; ++i) {
                float ms = 0.f;
                VDB_HIP(hipEventSynchronize(h->ev_total[2 * i + 1]));
                VDB_HIP(hipEventElapsedTime(&ms, h->ev_scan[2 * i], h->ev_scan[2 * i + 1]));
                scan += ms;
                VDB_HIP(hipEventElapsedTime(&ms, h->ev_total[2 * i], h->ev_total[2 * i + 1]));
                total += ms;
                VDB_HIP(hipEventElapsedTime(&ms, h->ev_total[2 * i], h->ev_scan[2 * i]));
                prep += ms;
                VDB_HIP(hipEventElapsedTime(&ms, h->ev_scan[2 * i + 1], h->ev_total[2 * i + 1]));
                tail += ms;
            }
            s.last_scan_ms = (float)(scan / h->ev_used);
            s.last_total_ms = (float)(total / h->ev_used);
            s.last_prep_ms = (float)(prep / h->ev_used);
            s.last_tail_ms = (float)(tail / h->ev_used);
        }
        *out = s;
    });
}

int vdb_debug_fetch_stamps(vdb_handle hh, unsigned long long *out_host, int64_t max_words, int64_t *nwords) {
    return guarded([&] {
        auto *h = check(hh);
        if (!out_host || !nwords) throw Error(VDB_ERR_INVALID, "null pointer");
        if (h->multi) multi_unsupported("vdb_debug_fetch_stamps");
        set_device(h->device);
        const int64_t n = std::min<int64_t>((int64_t)h->dbg_words, max_words);
        *nwords = n;
        if (n > 0) {
            VDB_HIP(hipDeviceSynchronize());
            VDB_HIP(hipMemcpy(out_host, h->ws.dense.p, (size_t)n * 8, hipMemcpyDeviceToHost));
        }
    });
}

int vdb_set_option(vdb_handle hh, const char *key, double value) {
    return guarded([&] {
        auto *h = check(hh);
        if (!key) throw Error(VDB_ERR_INVALID, "null option name");
        if (h->multi) return multi_set_option(h, key, value);
        const std::string k(key);
        graph_reset(h);                        // (a captured search embodies the options it was captured under)
        if (k == "graph") {
            if (value != 0 && value != 1) throw Error(VDB_ERR_INVALID, "graph must be 0 or 1");
            h->graph_mode = (int)value;
        } else if (k == "graph_recapture_at_once") {   // diagnostic: destroy a stale exec and capture its successor in ONE call
            h->graph_recapture_at_once = value != 0;
        } else if (k == "force_path") {
            if (value != 0 && value != 1 && value != 2 && value != 3)
                throw Error(VDB_ERR_INVALID, "force_path must be 0, 1, 2 or 3");
            h->force_path = (int)value;
        } else if (k == "timing") {  // (re)starts the recording window
            h->timing = value != 0;
            h->ev_used = 0;
        } else if (k == "panel_layout") {   // 0 auto, 1 = 32-row tiles for every D, 2 = p16 for every D (next add)
            if (value != 0 && value != 1 && value != 2) throw Error(VDB_ERR_INVALID, "panel_layout must be 0, 1 or 2");
            h->layout_override = (int)value;
        } else if (k == "stream_panels") {  // D > 128, next add: 0 keep the fp16 panels resident | 1 convert them per search
            if (value != 0 && value != 1) throw Error(VDB_ERR_INVALID, "stream_panels must be 0 or 1");
            h->stream_panels_opt = (int)value;
        } else if (k == "fused_stats") {     // 1 (default) | 0: separate query_stats_kernel for every batch size (A/B)
            if (value != 0 && value != 1) throw Error(VDB_ERR_INVALID, "fused_stats must be 0 or 1");
            h->no_fused_stats = value == 0;
        } else if (k == "stream_slab_rows") {  // rows of the scratch slab of a streamed index (0 = default 1 280 000)
            if (value < 0) throw Error(VDB_ERR_INVALID, "stream_slab_rows must be >= 0");
            h->stream_slab_rows = value;
        } else if (k == "panel_dtype") {    // 0 auto (int8 scan copy used when corpus and queries allow), 1 = fp16 scan only
            if (value != 0 && value != 1) throw Error(VDB_ERR_INVALID, "panel_dtype must be 0 or 1");
            h->i8_disable = (int)value;
        } else if (k == "upload_block_mb") {   // staging block of the row-block ingestion (0 = default 64 MiB)
            if (value < 0 || value > 4096) throw Error(VDB_ERR_INVALID, "upload_block_mb out of range");
            h->upload_block_mb = (int)value;
        } else if (k == "ivf_bt") {
            if (value != 0 && value != 4 && value != 16) throw Error(VDB_ERR_INVALID, "ivf_bt must be 0, 4 or 16");
            h->ivf_bt = (int)value;
        } else if (k == "ivf_tps") {           // D > 128, next add: tiles per panel span (0 auto, 16 = 64-row bins, 64 = 256-row bins)
            if (value != 0 && value != 16 && value != 64) throw Error(VDB_ERR_INVALID, "ivf_tps must be 0, 16 or 64");
            h->ivf_tps_override = (int)value;
        } else if (k == "ivf_part") {
            if (value < 0 || value > 1024) throw Error(VDB_ERR_INVALID, "ivf_part must be 0 (auto) or 1..1024 spans");
            h->ivf_part = (int)value;
        } else if (k == "ivf_min_batch") {
            if (value < 1 || value > 1e9) throw Error(VDB_ERR_INVALID, "ivf_min_batch must be >= 1");
            h->ivf_min_batch = (int)value;
        } else if (k == "ivf_tile") {
            if (value != 0 && value != 1 && value != 2) throw Error(VDB_ERR_INVALID, "ivf_tile must be 0, 1 or 2");
            h->ivf_tile = (int)value;
        } else if (k == "ivf_i8_group") {
            if (value != 4 && value != 8) throw Error(VDB_ERR_INVALID, "ivf_i8_group must be 4 or 8");
            h->ivf_i8_group = (int)value;
        } else if (k == "ivf_group") {
            if (value != 0 && value != 1 && value != 2 && value != 4) throw Error(VDB_ERR_INVALID, "ivf_group must be 0, 1, 2 or 4");
            h->ivf_group = (int)value;
        } else if (k == "ivf_nw") {
            if (value != 0 && value != 2 && value != 4 && value != 8) throw Error(VDB_ERR_INVALID, "ivf_nw must be 0, 2, 4 or 8");
            h->ivf_nw = (int)value;
        } else if (k == "small_batch") {       // 1 (default): finer chunks / narrower workgroups for batches <= 512 queries
            if (value != 0 && value != 1) throw Error(VDB_ERR_INVALID, "small_batch must be 0 or 1");
            h->small_batch_off = value == 0;
        } else if (k == "i8_group") {          // rows per select group of the int8 scan: 8 (octs, default) or 4 (quads)
            if (value != 4 && value != 8) throw Error(VDB_ERR_INVALID, "i8_group must be 4 or 8");
            h->i8_group = (int)value;
        } else if (k == "flat_shape" || k == "i8_shape") {   // MFMA shape of the flat scans, D <= 128 (layout of the scan copies: set before vdb_add)
            if (value != 0 && value != 16 && value != 32) throw Error(VDB_ERR_INVALID, "flat_shape must be 0 (auto), 16 or 32");
            h->flat_shape_opt = (int)value;
        } else if (k == "int8_only") {
            if (value != 0 && value != 1) throw Error(VDB_ERR_INVALID, "int8_only must be 0 or 1");
            h->int8_only_opt = (int)value;
        } else if (k == "int8_block_rows") {
            if (value < 0 || value > 2147483647.0) throw Error(VDB_ERR_INVALID, "int8_block_rows out of range");
            h->int8_block_rows = (int64_t)value;
        } else if (k == "int8_slab_chunks") {
            if (value < 0 || value > 1024) throw Error(VDB_ERR_INVALID, "int8_slab_chunks out of range");
            h->int8_slab_chunks = (int64_t)value;
        } else if (k == "f16_group") {         // rows per select group of the fp16 flat scan: 8 (octs, default) or 4 (quads)
            if (value != 4 && value != 8) throw Error(VDB_ERR_INVALID, "f16_group must be 4 or 8");
            h->f16_group = (int)value;
        } else if (k == "i8_nt") {
            if (value != 0 && value != 1 && value != 2) throw Error(VDB_ERR_INVALID, "i8_nt must be 0, 1 or 2");
            h->i8_nt = (int)value;
        } else if (k == "i8_ring") {           // staging ring of the serving-shaped / IVF int8 scans: 0 auto, 2 (double buffer), 4, 8
            if (value != 0 && value != 2 && value != 4 && value != 8) throw Error(VDB_ERR_INVALID, "i8_ring must be 0, 2, 4 or 8");
            h->i8_ring = (int)value;
        } else if (k == "i8_variant") {
#ifdef VDB_ABLATIONS
            if (value < 0 || value > 23) throw Error(VDB_ERR_INVALID, "i8_variant must be 0..7 (+8, +16)");
#else
            if (value < 0 || value > 7) throw Error(VDB_ERR_INVALID, "i8_variant must be 0..7");
#endif
            h->i8_variant = (int)value;
        } else if (k == "kloop_qgroup") {
            if (value < 0 || value > 1024) throw Error(VDB_ERR_INVALID, "kloop_qgroup out of range");
            h->kloop_qgroup = (int)value;
        } else if (k == "f16_wide") {
            if (value != 0 && value != 1) throw Error(VDB_ERR_INVALID, "f16_wide must be 0 or 1");
            h->f16_wide = (int)value;
        } else if (k == "f16_stage_tiles") {
            if (value != 0 && value != 4 && value != 8) throw Error(VDB_ERR_INVALID, "f16_stage_tiles must be 0, 4 or 8");
            h->f16_stage_tiles = (int)value;
        } else if (k == "scan_pair") {         // 1 (default): both x16 scans of an index with an int8 copy in one launch; 0: two launches
            if (value != 0 && value != 1) throw Error(VDB_ERR_INVALID, "scan_pair must be 0 or 1");
            h->scan_pair_off = value == 0;
        } else if (k == "scan_prio") {
            if (value < 0 || value > 2) throw Error(VDB_ERR_INVALID, "scan_prio must be 0, 1 or 2");
            h->scan_prio = (int)value;
        } else if (k == "scan_variant") {
#ifdef VDB_ABLATIONS
            if (value < 0 || value >= kNumScanVariants) throw Error(VDB_ERR_INVALID, "scan_variant out of range");
#else
            // the A/B schedules and the timing-only ablations (some return wrong neighbours) exist only in
            // -DVDB_ABLATIONS builds (`make ablations`, scripts/sweep_*.py); the shipped library has the one schedule
            if (value != 0) throw Error(VDB_ERR_UNSUPPORTED, "scan_variant needs a -DVDB_ABLATIONS build of libvdbhip");
#endif
            h->scan_variant = (int)value;
        } else if (k == "spans_per_chunk") {  // tuning: rows per workgroup chunk = 512 * value (0 = default 16)
            if (value < 0 || value > 4096) throw Error(VDB_ERR_INVALID, "spans_per_chunk out of range");
            h->spc_override = (int)value;
        } else if (k == "select_variant") {
            if (value < 0 || value > 2) throw Error(VDB_ERR_INVALID, "select_variant must be 0, 1 or 2");
            h->select_variant = (int)value;
        } else if (k == "list_cap") {
            if (value < 0 || value > 65536) throw Error(VDB_ERR_INVALID, "list_cap out of range");
            h->list_cap = (int)value;
        } else {
            throw Error(VDB_ERR_INVALID, "unknown option '" + k + "'");
        }
    });
}

}  // extern "C"

#include "debug_ivf.inc"
#include "multi.inc"
