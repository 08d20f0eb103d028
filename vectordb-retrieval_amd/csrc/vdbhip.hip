// vdbhip.hip -- host side of libvdbhip.so: handle management, path selection, kernel launches,
// and the extern "C" entry points declared in include/vdbhip.h.  gfx950 only.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
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
#include "ivf.hpp"
#include "ivf_mfma.hpp"
#include "dense.hpp"

using namespace vdb;

namespace {

thread_local std::string g_last_error;

// bumped by every (re)allocation or release of a DevBuf: a captured hipGraph holds raw buffer addresses, so it may only be
// replayed while no buffer of this process has moved since its capture (graph_or_run)
std::atomic<uint64_t> g_alloc_epoch{0};

// growable device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        g_alloc_epoch.fetch_add(1, std::memory_order_relaxed);
        if (p) VDB_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
        const size_t want = bytes + (bytes >> 3);
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) throw Error(VDB_ERR_NOMEM, "hipMalloc of " + std::to_string(want) + " bytes failed");
        cap = want;
    }
    void release() {
        if (p) {
            g_alloc_epoch.fetch_add(1, std::memory_order_relaxed);
            (void)hipFree(p);
        }
        p = nullptr;
        cap = 0;
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
    DevBuf qpad, qpanels, qpanels8, qrows8, info, eps, bin_m1, bin_m2, bin_m3, sb_m1, sb_m2, sb_span;
    DevBuf cand, rescan, counts, fallback, fb_list, small;  // small: fb_count (int) + 2 stat counters
    DevBuf dense;            // nq x Npad raw scores of the small-corpus path
    DevBuf pkeys, pids;      // partial lists of the exhaustive / fallback passes
    DevBuf stage_q, stage_d, stage_i;  // host-API staging
    size_t bytes() const {
        const DevBuf *all[] = {&qpad, &qpanels, &qpanels8, &qrows8, &info, &eps, &bin_m1, &bin_m2, &bin_m3, &sb_m1, &sb_m2, &sb_span, &cand,
                               &rescan, &counts, &fallback, &fb_list, &small, &pkeys, &pids, &stage_q, &stage_d,
                               &stage_i, &dense};
        size_t s = 0;
        for (auto b : all) s += b->cap;
        return s;
    }
    void release() {
        DevBuf *all[] = {&qpad, &qpanels, &qpanels8, &qrows8, &info, &eps, &bin_m1, &bin_m2, &bin_m3, &sb_m1, &sb_m2, &sb_span, &cand,
                         &rescan, &counts, &fallback, &fb_list, &small, &pkeys, &pids, &stage_q, &stage_d, &stage_i,
                         &dense};
        for (auto b : all) b->release();
    }
};

inline QueryBatchInfo *batch_info(Workspace &ws) {        // inside ws.small (common.hpp, kInfoOffset)
    return reinterpret_cast<QueryBatchInfo *>(ws.small.as<char>() + kInfoOffset);
}

}  // namespace

struct vdb_index_s {
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
    int ivf_nw = 0;                          // option "ivf_nw": waves per IVF work item (0 auto, 2 / 4 / 8)
    int i8_group = 8;                        // rows per select group of the int8 scan (option "i8_group": 4 or 8)
    // option "graph": a device-resident search that repeats with the same shape and buffers (a serving loop) is captured
    // into a hipGraph on its second call and replayed from the third (graph_or_run)
    int graph_mode = 0;
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
    bool small_is_clean = false;             // ws.small was cleared for this call and no batch has used it yet
    int64_t info_valid_nq = -1;              // queries whose statistics the last search_batch left in batch_info(ws) (-1: none)
    bool tile16 = false;                     // panels in the p16 layout (16-row tiles, 1024-row spans, 4 bins per span)
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
    DevBuf ivf_list_pspan0, ivf_span_row0, ivf_span_valid;
    DevBuf ivf_cnt /* per-list counts | cursors | slot -> query map: one buffer, one memset */, ivf_slot_off, ivf_list_item0,
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
            if (h->pin[i]) (void)hipHostFree(h->pin[i]);
            h->pin[i] = nullptr;
        }
        h->pin_bytes = 0;
        if (hipHostMalloc(&h->pin[0], need, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc(&h->pin[1], need, hipHostMallocDefault) == hipSuccess) {
            h->pin_bytes = need;
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

void build_index(vdb_index_s *h, const float *x_dev_or_host, bool on_device, int64_t n, int64_t id_base,
                 hipStream_t st) {
    if (n < 0) throw Error(VDB_ERR_INVALID, "negative row count");
    graph_reset(h);
    if (n > 2147483647ll - 1024) throw Error(VDB_ERR_UNSUPPORTED, "more than 2^31 rows per shard");
    const int D = h->dim, D4 = h->D4;
    h->built = false;
    h->N = n;
    h->id_base = id_base;
    // D > 128 (K-loop scan): p16 panels for v_mfma_f32_16x16x32_f16; D <= 128: 32-row tiles (scan_kernel, dense path)
    // (panel_layout 2 = p16 for D <= 128 too, when the corpus is too large for the dense small-corpus kernel)
    h->tile16 = h->layout_override != 1 && (h->ksteps > kMaxKSteps || (h->layout_override == 2 && n > kDenseMaxRows));
    const int64_t span_rows = h->tile16 ? kSpanRows16 : kSpanRows;
    h->Npad = (n + span_rows - 1) / span_rows * span_rows;
    h->scan_ok = false;
    if (n == 0) {
        h->built = true;
        return;
    }
    h->x32.reserve((size_t)n * D4 * sizeof(float));
    if (D4 != D) VDB_HIP(hipMemsetAsync(h->x32.p, 0, (size_t)n * D4 * sizeof(float), st));
    if (on_device)
        VDB_HIP(hipMemcpy2DAsync(h->x32.p, (size_t)D4 * 4, x_dev_or_host, (size_t)D * 4, (size_t)D * 4, (size_t)n,
                                 hipMemcpyDeviceToDevice, st));
    else
        upload_rows(h, h->x32.as<float>(), D4, x_dev_or_host, n, D, st);
    index_stats(h, st);
    const bool dims_ok = D <= 4096;
    if (dims_ok && !h->nonfinite) {
        const int64_t ntiles = h->Npad / (h->tile16 ? kTileRows16 : kTileRows);
        const int ksl = h->tile16 ? h->ksteps / 2 : h->ksteps;          // k-steps of the layout (32 or 16 dims)
        h->panels.reserve((size_t)ntiles * ksl * 64 * sizeof(half8));
        const int64_t threads = ntiles * ksl * 64;
        if (h->tile16)
            build_panels16_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st>>>(h->x32.as<float>(), n, D, D4, ksl, ntiles, h->sx, h->panels.as<half8>(), h->stats.as<IndexStats>());
        else
            build_panels_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st>>>(h->x32.as<float>(), n, D, D4, h->ksteps, ntiles, h->sx, h->panels.as<half8>(), h->stats.as<IndexStats>());
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
                h->x32.as<float>(), n, D, D4, h->i8_ks, ntiles, h->i8_cx, h->panels8.as<int4v>());
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

// ---- geometry of the scan for a given (N, k) --------------------------------------------------------
struct ScanGeom {
    bool ok = false;
    int64_t nspans = 0;
    int spc = 0, rem = 0, nchunks = 0, vpl = 0;
};

// nq: queries of the batch.  Small batches (serving-shaped: up to one 512-query tile) take FINER chunks, so that the
// grid still covers the chip: at 8192 rows per chunk a 1M-row corpus gives 128 workgroups per query tile, half the CUs.
ScanGeom scan_geometry(const vdb_index_s *h, int k, int64_t nq) {
    ScanGeom g;
    const int G = h->tile16 ? 4 : 2;                        // bins (lane groups) per span
    g.nspans = h->Npad / (G * kBinRows);
    if (g.nspans * G < 16) return g;
    // superbins (G per chunk) must comfortably outnumber k and fit 16 values per lane in the select
    int64_t spc_hi = g.nspans * G / (4 * (int64_t)k);       // nsb >= 4k
    int64_t spc_lo = (g.nspans * G + 1023) / 1024;          // nsb <= 1024
    if (spc_hi < 1 || spc_hi < spc_lo) return g;
    int64_t spc_want = 32 / G;                              // 8192 rows per chunk
    // (one query tile only: from 1024 queries on the batch shape is as fast or faster -- scripts/throughput_vs_batch.py)
    if (nq <= 512 && !h->small_batch_off) spc_want = std::max<int64_t>(1, std::min<int64_t>(spc_want, g.nspans / 512));
    int64_t spc = std::min<int64_t>(h->spc_override > 0 ? h->spc_override : spc_want, spc_hi);
    spc = std::max<int64_t>(spc, spc_lo);
    if (spc < 2 && g.nspans * G >= 128) spc = std::min<int64_t>(2, spc_hi);
    int64_t nchunks = (g.nspans + spc - 1) / spc;
    // deal the spans evenly over a multiple of 8 chunks (one XCD label each, see scan_kernel) when possible
    if (nchunks >= 16) {
        int64_t n8 = (nchunks + 7) / 8 * 8;
        if (n8 * G <= 1024 && g.nspans / n8 >= 1) nchunks = n8;   // (more chunks only add superbins)
    }
    g.nchunks = (int)nchunks;
    g.spc = (int)(g.nspans / nchunks);
    g.rem = (int)(g.nspans - (int64_t)g.spc * nchunks);
    if (g.spc < 1) return g;
    const int nsb = G * g.nchunks;
    g.vpl = 1;
    while (g.vpl * 64 < nsb) g.vpl *= 2;
    if (g.vpl > 16) return g;
    g.ok = true;
    return g;
}

// Direct-bin geometry: when N/256 superbins cannot outnumber k four to one, level-1 bins of 128 or 64 rows serve as
// the superbins themselves (SelectArgs.direct_rows, up to 2048 of them); the chunking then only shapes the grid.
// Returns the rows per bin, 0 if not applicable.
int scan_geometry_direct(const vdb_index_s *h, int k, ScanGeom &g) {
    const int G = h->tile16 ? 4 : 2;
    // kernels with the finer bins: scan_kernel<.., BT> (32-row tiles, D <= 128) and scan16_kloop_kernel<.., FP> (p16, D > 128)
    if (h->tile16 != (h->ksteps > kMaxKSteps)) return 0;
    const int64_t nspans = h->Npad / (G * kBinRows);
    if (nspans * G < 16) return 0;
    for (int rows : {128, 64}) {
        const int64_t nb = nspans * G * (kBinRows / rows);
        if (nb >= 4 * (int64_t)k && nb <= 2048) {
            g = ScanGeom{};
            g.nspans = nspans;
            int64_t nchunks = (nspans * G + 31) / 32;                 // ~8192 rows per chunk
            if (nchunks >= 16) nchunks = (nchunks + 7) / 8 * 8;
            nchunks = std::max<int64_t>(1, std::min<int64_t>(nchunks, nspans));
            g.nchunks = (int)nchunks;
            g.spc = (int)(nspans / nchunks);
            g.rem = (int)(nspans - (int64_t)g.spc * nchunks);
            g.vpl = 1;
            while (g.vpl * 64 < nb) g.vpl *= 2;
            g.ok = true;
            return rows;
        }
    }
    return 0;
}

// scan kernel variants: {waves per workgroup, tiles per LDS stage, waves per SIMD}.  Variant 0 is the
// production one; the others exist for the interleaved A/B in scripts/sweep_scan.py (7..9 are timing-only
// ablations of variant 0 and return wrong results).
struct ScanVariant { int nwaves, st, wps; };
constexpr ScanVariant kScanVariants[] = {{8, 4, 2}, {4, 4, 2}, {8, 4, 2}, {8, 4, 2}, {8, 4, 2}, {4, 4, 2}, {8, 4, 2},
                                         {8, 4, 2}, {8, 4, 2}, {8, 4, 2}};
[[maybe_unused]] constexpr int kNumScanVariants = sizeof(kScanVariants) / sizeof(kScanVariants[0]);

template <int KSTEPS>
void launch_scan_k(int variant, ScanArgs &sa, int nchunks, int64_t Qpad, hipStream_t st, int bt = 16, int nw = 8) {
    if (bt == 16 && variant == 0 && nw < 8) {   // small batch: 64 * nw queries per workgroup, only the tiles that hold queries
        sa.nqtiles = (int)((sa.nq_valid + nw * 64 - 1) / (nw * 64));
        const unsigned grid = 8u * (unsigned)((nchunks + 7) / 8) * (unsigned)sa.nqtiles;
        if (nw == 1) scan_kernel<KSTEPS, 1, 4, 1><<<dim3(grid), dim3(64), 0, st>>>(sa);
        else if (nw == 2) scan_kernel<KSTEPS, 2, 4, 1><<<dim3(grid), dim3(128), 0, st>>>(sa);
        else scan_kernel<KSTEPS, 4, 4, 2><<<dim3(grid), dim3(256), 0, st>>>(sa);
        return;
    }
    if (bt != 16) {            // direct-bin mode: finer level-1 bins (8 or 4 tiles), production schedule only
        sa.nqtiles = (int)(Qpad / 512);
        const unsigned grid = 8u * (unsigned)((nchunks + 7) / 8) * (unsigned)sa.nqtiles;
        if (bt == 8) scan_kernel<KSTEPS, 8, 4, 2, 0, 8><<<dim3(grid), dim3(512), 0, st>>>(sa);
        else scan_kernel<KSTEPS, 8, 4, 2, 0, 4><<<dim3(grid), dim3(512), 0, st>>>(sa);
        return;
    }
    const ScanVariant v = kScanVariants[variant];
    sa.nqtiles = (int)(Qpad / (v.nwaves * 64));
    const unsigned grid = 8u * (unsigned)((nchunks + 7) / 8) * (unsigned)sa.nqtiles;
    switch (variant) {
#ifdef VDB_ABLATIONS   // A/B schedules and timing-only ablations (WRONG RESULTS for 4, 7..9): scripts/sweep_scan.py builds
        case 1: scan_kernel<KSTEPS, 4, 4, 2><<<dim3(grid), dim3(256), 0, st>>>(sa); break;
        case 4: scan_kernel<KSTEPS, 8, 4, 2, 5><<<dim3(grid), dim3(512), 0, st>>>(sa); break;
        case 5: scan_kernel<KSTEPS, 4, 4, 2, 0, 16, false, 3><<<dim3(grid), dim3(256), 0, st>>>(sa); break;
        case 2: scan_kernel<KSTEPS, 8, 4, 2, 0, 16, false, 1><<<dim3(grid), dim3(512), 0, st>>>(sa); break;
        case 3: scan_kernel<KSTEPS, 8, 4, 2, 0, 16, false, 2><<<dim3(grid), dim3(512), 0, st>>>(sa); break;
        case 6: scan_kernel<KSTEPS, 8, 4, 2, 4><<<dim3(grid), dim3(512), 0, st>>>(sa); break;
        case 7: scan_kernel<KSTEPS, 8, 4, 2, 1><<<dim3(grid), dim3(512), 0, st>>>(sa); break;
        case 8: scan_kernel<KSTEPS, 8, 4, 2, 2><<<dim3(grid), dim3(512), 0, st>>>(sa); break;
        case 9: scan_kernel<KSTEPS, 8, 4, 2, 3><<<dim3(grid), dim3(512), 0, st>>>(sa); break;
#endif
        default: scan_kernel<KSTEPS, 8, 4, 2><<<dim3(grid), dim3(512), 0, st>>>(sa); break;
    }
}

void launch_scan(vdb_index_s *h, ScanArgs &sa, int nchunks, int64_t Qpad, hipStream_t st, int direct_rows = 0, int nw = 8) {
    const int bt = direct_rows ? direct_rows / 16 : 16;       // 32-row-tile layout: tiles per level-1 bin
    if (h->ksteps > kMaxKSteps) {  // D > 128
        // scan_variant: 0 = 4 row tiles x 2 query blocks per wave (512-query tiles), 1 = 8 x 1 (256-query tiles);
        // option kloop_qgroup = query tiles per group of the block order (0 -> default)
        const bool wide = h->tile16 || (h->scan_variant != 1 && h->scan_variant != 4);   // 512-query tiles
        sa.nqtiles = (int)(Qpad / (wide ? 512 : 256));
        int qgroup = h->kloop_qgroup > 0 ? h->kloop_qgroup : (wide ? 4 : sa.nqtiles);
        qgroup = std::min(qgroup, sa.nqtiles);
        const unsigned ngroups = (unsigned)((sa.nqtiles + qgroup - 1) / qgroup);
        const unsigned grid = 8u * (unsigned)((nchunks + 7) / 8) * ngroups * (unsigned)qgroup;
        const ScanKloopExtra ex{h->ksteps, qgroup};
        if (h->tile16 && direct_rows) {      // finer level-1 bins (production schedule only)
            if (direct_rows == 128) scan16_kloop_kernel<0, 2, 4><<<dim3(grid), dim3(512), 0, st>>>(sa, ex);
            else scan16_kloop_kernel<0, 2, 2><<<dim3(grid), dim3(512), 0, st>>>(sa, ex);
            VDB_HIP(hipGetLastError());
            return;
        }
        if (h->tile16) {
            switch (h->scan_variant) {
#ifdef VDB_ABLATIONS
                case 2: scan16_kloop_kernel<0, 1><<<dim3(grid), dim3(512), 0, st>>>(sa, ex); break;
                case 7: scan16_kloop_kernel<2, 1><<<dim3(grid), dim3(512), 0, st>>>(sa, ex); break;  // no MFMA
                case 8: scan16_kloop_kernel<3, 1><<<dim3(grid), dim3(512), 0, st>>>(sa, ex); break;  // no traffic
                case 9: scan16_kloop_kernel<4, 1><<<dim3(grid), dim3(512), 0, st>>>(sa, ex); break;  // no epilogue
#endif
                default:
                    if (sa.nq_valid > 0 && sa.nq_valid <= 16 && h->ksteps / 2 <= kNarrowMaxKS && !h->small_batch_off)
                        scan16_kloop_kernel<0, 3, 8, 2><<<dim3(grid), dim3(512), 0, st>>>(sa, ex);
                    else if (sa.nq_valid > 0 && sa.nq_valid < 64 && !h->small_batch_off)
                        scan16_kloop_kernel<0, 2, 8, 1><<<dim3(grid), dim3(512), 0, st>>>(sa, ex);
                    else
                        scan16_kloop_kernel<0, 2><<<dim3(grid), dim3(512), 0, st>>>(sa, ex);
                    break;
            }
            VDB_HIP(hipGetLastError());
            return;
        }
        switch (h->scan_variant) {
#ifdef VDB_ABLATIONS
            case 1: scan_kloop_kernel<0, 8, 1, 1><<<dim3(grid), dim3(512), 0, st>>>(sa, ex); break;
            case 2: scan_kloop_kernel<0, 4, 2, 2><<<dim3(grid), dim3(512), 0, st>>>(sa, ex); break;
            case 3: scan_kloop_kernel<0, 4, 2, 4><<<dim3(grid), dim3(512), 0, st>>>(sa, ex); break;
            case 4: scan_kloop_kernel<0, 4, 2, 1, true, 4><<<dim3(grid), dim3(256), 0, st>>>(sa, ex); break;
            case 9: scan_kloop_kernel<4, 4, 2, 1><<<dim3(grid), dim3(512), 0, st>>>(sa, ex); break;  // no epilogue
            case 7: scan_kloop_kernel<2, 4, 2, 1><<<dim3(grid), dim3(512), 0, st>>>(sa, ex); break;  // no MFMA
            case 8: scan_kloop_kernel<3, 4, 2, 1><<<dim3(grid), dim3(512), 0, st>>>(sa, ex); break;  // no traffic
#endif
            default: scan_kloop_kernel<0, 4, 2, 1><<<dim3(grid), dim3(512), 0, st>>>(sa, ex); break;
        }
    } else if (h->tile16) {            // D <= 128 on p16 panels
        sa.nqtiles = (int)(Qpad / 512);
        const unsigned grid = 8u * (unsigned)((nchunks + 7) / 8) * (unsigned)sa.nqtiles;
        const int v = h->scan_variant;
        (void)v;
        if (h->ksteps == 4) {
#ifdef VDB_ABLATIONS
            if (v == 7) scan16_kernel<2, 8, 1><<<dim3(grid), dim3(512), 0, st>>>(sa);
            else if (v == 8) scan16_kernel<2, 8, 2><<<dim3(grid), dim3(512), 0, st>>>(sa);
            else
#endif
            scan16_kernel<2, 8><<<dim3(grid), dim3(512), 0, st>>>(sa);
        } else {
#ifdef VDB_ABLATIONS
            if (v == 7) scan16_kernel<4, 8, 1><<<dim3(grid), dim3(512), 0, st>>>(sa);
            else if (v == 8) scan16_kernel<4, 8, 2><<<dim3(grid), dim3(512), 0, st>>>(sa);
            else if (v == 1) scan16_kernel<4, 4><<<dim3(grid), dim3(512), 0, st>>>(sa);
            else
#endif
            scan16_kernel<4, 8><<<dim3(grid), dim3(512), 0, st>>>(sa);
        }
    } else if (h->ksteps == 4)
        launch_scan_k<4>(h->scan_variant, sa, nchunks, Qpad, st, bt, nw);
    else
        launch_scan_k<8>(h->scan_variant, sa, nchunks, Qpad, st, bt, nw);
    VDB_HIP(hipGetLastError());
}

template <int VPL>
void launch_select(const SelectArgs &a, hipStream_t st) {
    select_kernel<VPL><<<dim3((unsigned)((a.nq + 3) / 4)), dim3(256), 0, st>>>(a);
}

long timing_begin(vdb_index_s *h, hipStream_t st) {
    if (!h->timing || h->ev_used >= kMaxTimedCalls) return -1;
    const size_t slot = h->ev_used++;
    while (h->ev_scan.size() < 2 * (slot + 1)) {
        hipEvent_t e;
        VDB_HIP(hipEventCreate(&e));
        h->ev_scan.push_back(e);
        VDB_HIP(hipEventCreate(&e));
        h->ev_total.push_back(e);
    }
    VDB_HIP(hipEventRecord(h->ev_total[2 * slot], st));
    return (long)slot;
}

// which: 0 = dominant kernel starts, 1 = dominant kernel done, 2 = pipeline done
void timing_mark(vdb_index_s *h, long slot, int which, hipStream_t st) {
    if (slot < 0) return;
    hipEvent_t e = which == 0 ? h->ev_scan[2 * slot] : which == 1 ? h->ev_scan[2 * slot + 1] : h->ev_total[2 * slot + 1];
    VDB_HIP(hipEventRecord(e, st));
}

// ---- one query batch ---------------------------------------------------------------------------------
// outputs: final (D,I) or partial (pk,pi); all device pointers for rows [0,nq) of this batch
void search_batch(vdb_index_s *h, const float *dq, int64_t nq, int k, float *D, int64_t *I, double *pk, int64_t *pi,
                  hipStream_t st) {
    Workspace &ws = h->ws;
    const int Dm = h->dim, D4 = h->D4;
    // float32 queries padded to D4 for the refine kernels
    const float *qpad = dq;
    if (D4 != Dm) {
        ws.qpad.reserve((size_t)nq * D4 * sizeof(float));
        pad_rows_kernel<<<dim3((unsigned)((nq * D4 + 255) / 256)), dim3(256), 0, st>>>(dq, nq, Dm, D4, ws.qpad.as<float>());
        VDB_HIP(hipGetLastError());
        qpad = ws.qpad.as<float>();
    }
    RefineCommon rc{h->x32.as<float>(), qpad, h->N, h->id_base, D4, h->metric, k, nullptr};
    rc.info = batch_info(ws);                             // (group size of the candidates: 4 rows, 8 on the int8 scan)
    if (h->i8_ok && h->rows8.p && !h->i8_disable) {       // (used only by batches the device puts on the int8 scan)
        rc.X8 = h->rows8.as<signed char>();
        rc.rowstat = h->rowstat8.as<int>();
        rc.x8_pitch = h->rows8_pitch;
        rc.cx = h->i8_cx;
        rc.D = Dm;
        ws.qrows8.reserve((size_t)nq * h->rows8_pitch);
        rc.Q8 = ws.qrows8.as<signed char>();               // (filled next to the B fragments when the int8 scan is offered)
    }

    ScanGeom g;
    const bool exact_only = h->force_path == 1 || h->force_path == 3;   // 3: also without query blocking (A/B runs)
    bool use_scan = h->scan_ok && !exact_only && k <= 1024;
    int direct_rows = 0;
    if (use_scan) {
        g = scan_geometry(h, k, nq);
        if (!g.ok) direct_rows = scan_geometry_direct(h, k, g);
    }
    // (measured on 1M x 128 with host I/O: the MFMA pipeline answers 1..512 queries in 0.13..0.18 ms, the exhaustive float64
    //  kernel needs 0.3 ms for one query -- scripts/latency_small_batches.py -- so the batch size does not gate the path)
    // Between the dense small-corpus path (<= 8192 rows) and 32768 rows the scan pays off once the batch carries
    // enough (query, row) pairs: ~0.3 ms of fixed pipeline cost against ~7e-8 ms per pair in the exhaustive kernel.
    use_scan = use_scan && g.ok &&
               (h->force_path == 2 || h->N >= 32768 || (h->Npad > 8192 && (double)nq * (double)h->N >= 4.0e6));
    // (corpora of 8193..15360 rows whose chunking cannot give 4k superbins still have the dense path below)

    h->info_valid_nq = -1;
    // fb_count restarts with every batch (it indexes this batch's fb_list); the statistics counters behind it were
    // zeroed once for the whole call (search_device_impl) and accumulate over the batches
    if (h->small_is_clean) h->small_is_clean = false;      // the first batch of a call: search_device_impl has just cleared all of it
    else VDB_HIP(hipMemsetAsync(ws.small.p, 0, 64, st));
    int32_t *fb_count = ws.small.as<int32_t>();
    unsigned long long *stat_counters = reinterpret_cast<unsigned long long *>(ws.small.as<char>() + 64);

    const long tslot = timing_begin(h, st);

    // small corpora: dense fp16 scores + per-query guard + exact re-score of the few surviving rows
    const bool use_dense = !use_scan && h->scan_ok && !exact_only && h->ksteps <= kMaxKSteps &&
                           h->Npad <= kDenseMaxRows && nq >= 64 && k <= 1024 && (int64_t)k * 2 <= h->N &&
                           !h->tile16 && (h->Npad + std::max(128, 2 * k + 64)) * 4 <= 65536;   // scores + candidates in LDS
    if (use_dense) {
        const int64_t Qp = (nq + 63) / 64 * 64;
        const int cand_cap = std::max(128, 2 * k + 64);
        ws.qpanels.reserve((size_t)(Qp / 32) * h->ksteps * 64 * sizeof(half8));
        ws.eps.reserve((size_t)nq * sizeof(float));
        ws.dense.reserve((size_t)Qp * h->Npad * sizeof(float));
        ws.fallback.reserve((size_t)nq * sizeof(int32_t));
        ws.fb_list.reserve((size_t)nq * sizeof(int32_t));
        QueryBatchInfo *info = batch_info(ws);          // (zeroed with fb_count above)
        const int64_t total = nq * Dm;
        query_stats_kernel<<<dim3(query_stats_blocks(total)), dim3(256), 0, st>>>(dq, total, info,
            FinalizeArgs{h->sx, h->metric, h->corpus_int_unscaled ? 1 : 0, h->maxnorm2, 0});
        h->info_valid_nq = nq;
        const int64_t threads = (Qp / 32) * h->ksteps * 64;
        build_qpanels_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st>>>(
            dq, nq, Dm, D4, h->ksteps, Qp / 32, info, ws.qpanels.as<half8>());
        EpsArgs ea{dq, nq, Dm, h->ksteps * 16, h->metric, sqrtf(h->maxnorm2) * 1.0000002f,
                   h->corpus_fp16_exact ? 1 : 0, h->corpus_int_unscaled ? 1 : 0, h->sx, info, ws.eps.as<float>()};
        query_eps_kernel<<<dim3((unsigned)((nq * 16 + 255) / 256)), dim3(256), 0, st>>>(ea);
        const int64_t ntiles = h->Npad / kTileRows;
        timing_mark(h, tslot, 0, st);
        {
            const dim3 grid((unsigned)((ntiles + 3) / 4), (unsigned)(Qp / 64));
            if (h->ksteps == 4)
                dense_scores_kernel<4><<<grid, dim3(256), 0, st>>>(h->panels.as<half8>(), h->bias.as<float>(),
                                                                  ws.qpanels.as<half8>(), info, ntiles, h->Npad,
                                                                  ws.dense.as<float>());
            else
                dense_scores_kernel<8><<<grid, dim3(256), 0, st>>>(h->panels.as<half8>(), h->bias.as<float>(),
                                                                  ws.qpanels.as<half8>(), info, ntiles, h->Npad,
                                                                  ws.dense.as<float>());
        }
        timing_mark(h, tslot, 1, st);
        DenseSelectArgs da{};
        da.c = rc;
        da.scores = ws.dense.as<float>();
        da.eps = ws.eps.as<float>();
        da.info = info;
        da.nq = nq;
        da.Npad = h->Npad;
        da.cand_cap = cand_cap;
        da.fallback = ws.fallback.as<int32_t>();
        da.fb_list = ws.fb_list.as<int32_t>();
        da.fb_count = fb_count;
        da.stat_counters = stat_counters;
        da.D = D;
        da.I = I;
        da.pkeys = pk;
        da.pids = pi;
        da.set_only = (h->set_only && D != nullptr) ? 1 : 0;
        {
            const int kpl = kpl_for(k);
            if (h->Npad <= 2048 && kpl <= 4) {       // scores in registers, 4 queries per workgroup (dense.hpp)
                const dim3 g4((unsigned)((nq + 3) / 4));
                const size_t lds4 = (size_t)4 * 2 * cand_cap * 4;       // per wave: candidate rows + their keys
                if (h->Npad <= 1024) {
                    DISPATCH_KPL(kpl, (dense_select_reg_kernel<(KPL <= 4 ? KPL : 4), 16><<<g4, dim3(256), lds4, st>>>(da)));
                } else {
                    DISPATCH_KPL(kpl, (dense_select_reg_kernel<(KPL <= 4 ? KPL : 4), 32><<<g4, dim3(256), lds4, st>>>(da)));
                }
            } else {
                const size_t lds = (size_t)(h->Npad + cand_cap) * 4;
                DISPATCH_KPL(kpl, (dense_select_kernel<KPL><<<dim3((unsigned)nq), dim3(64), lds, st>>>(da)));
            }
            VDB_HIP(hipGetLastError());
        }
        // queries whose candidate list overflowed (or unusable scales): exhaustive exact pass
        int64_t S = std::min<int64_t>(16, std::max<int64_t>(1, h->N / 1024));
        const int64_t cap = std::max<int64_t>(1, (int64_t)(256ll << 20) / (nq * k * 16));
        S = std::max<int64_t>(1, std::min<int64_t>(S, cap));
        ws.pkeys.reserve((size_t)nq * S * k * sizeof(double));
        ws.pids.reserve((size_t)nq * S * k * sizeof(int64_t));
        RefineFullArgs fa{};
        fa.c = rc;
        fa.qlist = da.fb_list;
        fa.count_ptr = fb_count;
        fa.S = (int)S;
        fa.rows_per_split = (h->N + S - 1) / S;
        if (S == 1 && D) {       // one split: the exhaustive pass writes the final rows itself, nothing to merge
            fa.D = D;
            fa.I = I;
            launch_refine_full(fa, 1024, st);
        } else {
            fa.pkeys = ws.pkeys.as<double>();
            fa.pids = ws.pids.as<int64_t>();
            launch_refine_full(fa, 1024, st);
            MergeArgs ma{};
            ma.pkeys = fa.pkeys;
            ma.pids = fa.pids;
            ma.part_stride = k;
            ma.slot_stride = S * k;
            ma.nparts = (int)S;
            ma.k = k;
            ma.metric = h->metric;
            ma.qlist = da.fb_list;
            ma.count_ptr = fb_count;
            ma.D = D;
            ma.I = I;
            ma.okeys = pk;
            ma.oids = pi;
            launch_merge(ma, 256, st);
        }
        timing_mark(h, tslot, 2, st);
        h->last.last_path = VDB_PATH_MFMA_SCAN;
        return;
    }

    if (!use_scan) {
        // exhaustive exact scan, split over S waves per query (per group of 4 queries in the query-blocked form,
        // which fetches every row once for the four: k <= 128, at least 4 queries)
        // (it needs enough (group, split) units to fill the chip: small corpora and tiny batches keep the one-query form)
        bool blocked = kpl_for(k) <= 2 && nq >= 2 * kRefineQB && h->force_path != 3;
        if (blocked) {
            const int64_t g4 = (nq + kRefineQB - 1) / kRefineQB;
            const int64_t s4 = std::min<int64_t>((4096 + g4 - 1) / g4, std::max<int64_t>(1, h->N / 2048));
            blocked = g4 * s4 >= 1024;
        }
        const int64_t ngroups = blocked ? (nq + kRefineQB - 1) / kRefineQB : nq;
        auto launch_full = [&](const RefineFullArgs &fa) {
            if (blocked) launch_refine_full_blocked(fa, ngroups * fa.S, st);
            else launch_refine_full(fa, nq * fa.S, st);
        };
        int64_t S = ((blocked ? 4096 : 8192) + ngroups - 1) / ngroups;
        // (a split is worth >= 1024 rows; the blocked form merges 4x the partial lists per unit, so its splits are larger)
        S = std::min<int64_t>(S, std::max<int64_t>(1, h->N / (blocked ? 2048 : 1024)));
        // a handful of queries on a small corpus (an IVF coarse quantizer asked for 1..63 queries): one wave per query
        // would walk all rows alone (92 us for 8 queries x 1024 centroids) -- up to 64 splits of >= 128 rows instead
        if (!blocked && nq < 64) S = std::max<int64_t>(S, std::min<int64_t>(64, h->N / 128));
        const int64_t cap = std::max<int64_t>(1, (int64_t)(256ll << 20) / (nq * k * 16));
        S = std::max<int64_t>(1, std::min<int64_t>(S, cap));
        RefineFullArgs fa{};
        fa.c = rc;
        fa.count = nq;
        fa.S = (int)S;
        fa.rows_per_split = ((h->N + S - 1) / S + 63) / 64 * 64;     // whole 64-row wave iterations
        if (fa.rows_per_split < 64) fa.rows_per_split = 64;
        timing_mark(h, tslot, 0, st);
        if (S == 1 && D) {
            fa.D = D;
            fa.I = I;
            launch_full(fa);
        } else if (S == 1) {
            fa.pkeys = pk;
            fa.pids = pi;
            launch_full(fa);
        } else {
            ws.pkeys.reserve((size_t)nq * S * k * sizeof(double));
            ws.pids.reserve((size_t)nq * S * k * sizeof(int64_t));
            fa.pkeys = ws.pkeys.as<double>();
            fa.pids = ws.pids.as<int64_t>();
            launch_full(fa);
            MergeArgs ma{};
            ma.pkeys = fa.pkeys;
            ma.pids = fa.pids;
            ma.part_stride = k;
            ma.slot_stride = S * k;
            ma.nparts = (int)S;
            ma.k = k;
            ma.metric = h->metric;
            ma.count = nq;
            ma.D = D;
            ma.I = I;
            ma.okeys = pk;
            ma.oids = pi;
            launch_merge(ma, nq, st);
        }
        h->last.last_path = VDB_PATH_EXACT_SCAN;
        timing_mark(h, tslot, 1, st);
        timing_mark(h, tslot, 2, st);
        return;
    }

    // ---- MFMA scan path ------------------------------------------------------------------------------
    const int64_t Qpad = (nq + 511) / 512 * 512;
    const int G = h->tile16 ? 4 : 2;
    const int64_t nbins = g.nspans * G * (direct_rows ? kBinRows / direct_rows : 1), nsb = (int64_t)g.nchunks * G;
    const int cand_cap = h->list_cap > 0 ? h->list_cap : std::max(64, 2 * k + 32);
    const int rescan_cap = std::max(16, k / 2 + 8);
    // int8 scan for byte-valued corpora: offered to the device-side choice whenever the standard geometry is in use
    const bool use_i8 = h->i8_ok && !h->i8_disable && !direct_rows && !h->tile16 && h->scan_variant == 0;
    ws.qpanels.reserve((size_t)(Qpad / 32) * h->ksteps * 64 * sizeof(half8));
    ws.eps.reserve((size_t)nq * sizeof(float));
    ws.bin_m1.reserve((size_t)nbins * Qpad * sizeof(float));
    ws.bin_m2.reserve((size_t)nbins * Qpad * sizeof(float));
    ws.sb_m1.reserve((size_t)nsb * Qpad * sizeof(float));
    ws.sb_m2.reserve((size_t)nsb * Qpad * sizeof(float));
    ws.sb_span.reserve((size_t)nsb * Qpad * sizeof(int32_t));
    ws.cand.reserve((size_t)nq * cand_cap * sizeof(int32_t));
    ws.rescan.reserve((size_t)nq * rescan_cap * 2 * sizeof(int32_t));
    ws.counts.reserve((size_t)nq * 2 * sizeof(int32_t));
    ws.fallback.reserve((size_t)nq * sizeof(int32_t));
    ws.fb_list.reserve((size_t)nq * sizeof(int32_t));

    QueryBatchInfo *info = batch_info(ws);              // (zeroed with fb_count above)
    {
        const int64_t total = nq * Dm;
        query_stats_kernel<<<dim3(query_stats_blocks(total)), dim3(256), 0, st>>>(dq, total, info,
            FinalizeArgs{h->sx, h->metric, h->corpus_int_unscaled ? 1 : 0, h->maxnorm2,
                         use_i8 ? (1 | (h->i8_group == 8 ? 4 : 0)) : 0});
        h->info_valid_nq = nq;
        EpsArgs ea{dq, nq, Dm, h->ksteps * 16, h->metric, sqrtf(h->maxnorm2) * 1.0000002f,
                   h->corpus_fp16_exact ? 1 : 0, h->corpus_int_unscaled ? 1 : 0, h->sx, info, ws.eps.as<float>()};
        const int64_t threads = (Qpad / 32) * h->ksteps * 64;      // (same element count in both layouts)
        if (h->tile16) {
            build_qpanels16_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st>>>(dq, nq, Dm, h->ksteps / 2, Qpad / 16, info, ws.qpanels.as<half8>());
            query_eps_kernel<<<dim3((unsigned)((nq * 16 + 255) / 256)), dim3(256), 0, st>>>(ea);
        } else {            // one dispatch: fp16 fragments | int8 fragments | int8 query rows | error bounds (scan_i8.hpp)
            QueryPrepArgs qp{};
            qp.Q = dq; qp.nq = nq; qp.nqtiles = Qpad / 32;
            qp.D = Dm; qp.D4 = D4; qp.ksteps = h->ksteps; qp.ks32 = h->i8_ks; qp.pitch8 = h->rows8_pitch;
            qp.info = info;
            qp.qpanels = ws.qpanels.as<half8>();
            qp.eps = ea;
            unsigned nblk = (unsigned)((threads + 255) / 256);
            qp.nA = nblk;
            if (use_i8) {   // (the int8 regions return at once unless the device chose the int8 scan for this batch)
                ws.qpanels8.reserve((size_t)(Qpad / 32) * h->i8_ks * 64 * sizeof(int4v));
                qp.qpanels8 = ws.qpanels8.as<int4v>();
                nblk += (unsigned)(((Qpad / 32) * h->i8_ks * 64 + 255) / 256);
            }
            qp.nB = nblk;
            if (use_i8 && rc.Q8) {
                qp.qrows8 = ws.qrows8.as<signed char>();
                nblk += (unsigned)((nq * h->rows8_pitch + 255) / 256);
            }
            qp.nC = nblk;
            nblk += (unsigned)((nq * 16 + 255) / 256);
            query_prep_kernel<<<dim3(nblk), dim3(256), 0, st>>>(qp);
        }
        VDB_HIP(hipGetLastError());
    }

    ScanArgs sa{};
    sa.panels = h->panels.as<half8>();
    sa.bias = h->bias.as<float>();
    sa.qpanels = ws.qpanels.as<half8>();
    sa.info = info;
    sa.bin_m1 = ws.bin_m1.as<float>();
    sa.bin_m2 = ws.bin_m2.as<float>();
    sa.sb_m1 = ws.sb_m1.as<float>();
    sa.sb_m2 = ws.sb_m2.as<float>();
    sa.sb_span = ws.sb_span.as<int32_t>();
    sa.nspans = g.nspans;
    sa.spans_per_chunk = g.spc;
    sa.chunk_rem = g.rem;
    sa.nchunks = g.nchunks;
    sa.Qpad = Qpad;
    sa.nq_valid = nq;
    if (h->scan_variant == 6) {   // diagnostic build: per-wave cycle sums
        const size_t nblocks = 8 * (size_t)((g.nchunks + 7) / 8) * (size_t)(Qpad / 512);
        ws.dense.reserve(nblocks * 8 * 8 * sizeof(unsigned long long));
        VDB_HIP(hipMemsetAsync(ws.dense.p, 0, nblocks * 8 * 8 * sizeof(unsigned long long), st));
        sa.dbg = ws.dense.as<unsigned long long>();
        h->dbg_words = nblocks * 8 * 8;
    }
    timing_mark(h, tslot, 0, st);
    // small batches: as many waves per workgroup as there are 64-query column groups (1, 2, 4; 8 = the batch shape)
    const int nw_small = h->small_batch_off ? 8 : nq <= 64 ? 1 : nq <= 128 ? 2 : nq <= 256 ? 4 : 8;
    launch_scan(h, sa, g.nchunks, Qpad, st, direct_rows, nw_small);   // (fp16: returns at once when the int8 scan serves the batch)
    if (use_i8) {
        ScanI8Args s8{};
        s8.panels = h->panels8.as<int4v>();
        s8.bias8 = h->bias8.as<int32_t>();
        s8.qpanels = ws.qpanels8.as<int4v>();
        s8.info = info;
        s8.bin_m1 = sa.bin_m1; s8.bin_m2 = sa.bin_m2;
        s8.sb_m1 = sa.sb_m1; s8.sb_m2 = sa.sb_m2; s8.sb_span = sa.sb_span;
        s8.nspans = g.nspans; s8.Npad = h->Npad; s8.Qpad = Qpad; s8.nq_valid = nq;
        s8.spans_per_chunk = g.spc; s8.chunk_rem = g.rem; s8.nchunks = g.nchunks;
        // i8_variant (tuning, every variant exact; scripts/sweep_i8.py): 0 = 512-query tiles (2 column blocks per wave),
        // 4-tile stages; 1 = 8-tile stages; 2 = 1024-query tiles (4 column blocks per wave); 3 = both (default);
        // 4 / 5 = 16 waves per workgroup (4 per SIMD) with 8- / 16-tile stages.  Odd batch sizes keep 512-query tiles.
        int v8 = h->i8_variant;
#ifdef VDB_ABLATIONS
        s8.abl_no_bins = (v8 & 8) ? 1 : 0;      // +8: no level-1 bin stores (timing only, WRONG results)
        v8 &= 7;
#endif
        if (Qpad % 1024 != 0) v8 &= 1;
        if ((int64_t)g.nchunks * (Qpad / 1024) < 256) v8 &= 1;     // too few 1024-query tiles to cover the chip: 512-query tiles
        if (nw_small < 8) v8 = 8 + nw_small;
        const int qtile = (v8 > 8) ? 64 * nw_small : (v8 >= 2) ? 1024 : 512;
        s8.nqtiles = (int)((v8 > 8 ? nq + qtile - 1 : Qpad) / qtile);
        const dim3 grid8(8u * (unsigned)((g.nchunks + 7) / 8) * (unsigned)s8.nqtiles);
#define VDB_I8G(KS_, ST_, CB_, NW_, G_) scan_i8_kernel<KS_, ST_, CB_, NW_, 16, false, G_><<<grid8, dim3(NW_ * 64), 0, st>>>(s8)
#define VDB_I8(KS_, ST_, CB_, NW_) do { if (h->i8_group == 8) VDB_I8G(KS_, ST_, CB_, NW_, 8); else VDB_I8G(KS_, ST_, CB_, NW_, 4); } while (0)
        if (h->i8_ks == 2) {
            switch (v8) { case 1: VDB_I8(2, 8, 2, 8); break; case 2: VDB_I8(2, 4, 4, 8); break; case 3: VDB_I8(2, 8, 4, 8); break;
                          case 4: VDB_I8(2, 8, 2, 16); break; case 5: VDB_I8(2, 16, 2, 16); break;
                          case 9: VDB_I8(2, 4, 2, 1); break; case 10: VDB_I8(2, 4, 2, 2); break; case 12: VDB_I8(2, 4, 2, 4); break;
                          default: VDB_I8(2, 4, 2, 8); }
        } else {
            switch (v8) { case 1: VDB_I8(4, 8, 2, 8); break; case 2: VDB_I8(4, 4, 4, 8); break; case 3: VDB_I8(4, 8, 4, 8); break;
                          case 4: VDB_I8(4, 8, 2, 16); break; case 5: VDB_I8(4, 16, 2, 16); break;
                          case 9: VDB_I8(4, 4, 2, 1); break; case 10: VDB_I8(4, 4, 2, 2); break; case 12: VDB_I8(4, 4, 2, 4); break;
                          default: VDB_I8(4, 4, 2, 8); }
        }
#undef VDB_I8G
#undef VDB_I8
        VDB_HIP(hipGetLastError());
    }
    timing_mark(h, tslot, 1, st);

    SelectArgs se{};
    se.bin_m1 = sa.bin_m1;
    se.bin_m2 = sa.bin_m2;
    se.sb_m1 = sa.sb_m1;
    se.sb_m2 = sa.sb_m2;
    se.sb_span = sa.sb_span;
    se.eps = ws.eps.as<float>();
    se.info = info;
    se.nq = nq;
    se.Qpad = Qpad;
    se.nspans = g.nspans;
    se.N = h->N;
    se.spans_per_chunk = g.spc;
    se.chunk_rem = g.rem;
    se.nchunks = g.nchunks;
    se.k = k;
    se.groups = G;
    se.direct_rows = direct_rows;
    if (direct_rows) {        // the level-1 bins are the superbins
        se.sb_m1 = sa.bin_m1;
        se.sb_m2 = sa.bin_m2;
    }
    se.cand_cap = cand_cap;
    se.rescan_cap = rescan_cap;
    se.cand_rows = ws.cand.as<int32_t>();
    se.rescan_rows = ws.rescan.as<int32_t>();
    se.counts = ws.counts.as<int32_t>();
    se.fallback = ws.fallback.as<int32_t>();
    se.fb_list = ws.fb_list.as<int32_t>();
    se.fb_count = fb_count;
    se.stat_counters = stat_counters;
    const int nsb_i = G * g.nchunks;
    if (nsb_i > 128 && nsb_i <= 256 && h->select_variant == 2 && !direct_rows) {   // 32 lanes per query, 2 queries per wave
        select_kernel_v2<8, 32><<<dim3((unsigned)((nq + 7) / 8)), dim3(256), 0, st>>>(se);
    } else if (nsb_i <= 256 && h->select_variant != 1 && !direct_rows) {  // multi-lane form: 16 lanes per query, 4 queries per wave
        const unsigned sgrid = (unsigned)((nq + 15) / 16);
        if (nsb_i <= 64)
            select_kernel_v2<4, 16><<<dim3(sgrid), dim3(256), 0, st>>>(se);
        else if (nsb_i <= 128)
            select_kernel_v2<8, 16><<<dim3(sgrid), dim3(256), 0, st>>>(se);
        else
            select_kernel_v2<16, 16><<<dim3(sgrid), dim3(256), 0, st>>>(se);
    } else {
        switch (g.vpl) {
            case 1: launch_select<1>(se, st); break;
            case 2: launch_select<2>(se, st); break;
            case 4: launch_select<4>(se, st); break;
            case 8: launch_select<8>(se, st); break;
            case 16: launch_select<16>(se, st); break;
            default: launch_select<32>(se, st); break;
        }
    }
    VDB_HIP(hipGetLastError());

    RefineListArgs la{};
    la.c = rc;
    la.nq = nq;
    la.cand_rows = se.cand_rows;
    la.rescan_rows = se.rescan_rows;
    la.counts = se.counts;
    la.fallback = se.fallback;
    la.cand_cap = cand_cap;
    la.rescan_cap = rescan_cap;
    la.D = D;
    la.I = I;
    la.pkeys = pk;
    la.pids = pi;
    {
        const int kpl = kpl_for(k);
        DISPATCH_KPL(kpl, (refine_list_kernel<KPL><<<dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st>>>(la)));
        VDB_HIP(hipGetLastError());
    }

    // queries whose work list overflowed (or whose scales were unusable): exhaustive exact pass
    {
        int64_t S = std::min<int64_t>(64, std::max<int64_t>(1, h->N / 4096));
        const int64_t cap = std::max<int64_t>(1, (int64_t)(256ll << 20) / (nq * k * 16));
        S = std::max<int64_t>(1, std::min<int64_t>(S, cap));
        ws.pkeys.reserve((size_t)nq * S * k * sizeof(double));
        ws.pids.reserve((size_t)nq * S * k * sizeof(int64_t));
        RefineFullArgs fa{};
        fa.c = rc;
        fa.qlist = se.fb_list;
        fa.count_ptr = fb_count;
        fa.S = (int)S;
        fa.rows_per_split = (h->N + S - 1) / S;
        fa.pkeys = ws.pkeys.as<double>();
        fa.pids = ws.pids.as<int64_t>();
        launch_refine_full(fa, 1024, st);
        MergeArgs ma{};
        ma.pkeys = fa.pkeys;
        ma.pids = fa.pids;
        ma.part_stride = k;
        ma.slot_stride = S * k;
        ma.nparts = (int)S;
        ma.k = k;
        ma.metric = h->metric;
        ma.qlist = se.fb_list;
        ma.count_ptr = fb_count;
        ma.D = D;
        ma.I = I;
        ma.okeys = pk;
        ma.oids = pi;
        launch_merge(ma, 256, st);
    }
    timing_mark(h, tslot, 2, st);
    h->last.last_path = VDB_PATH_MFMA_SCAN;
}

// ---- hipGraph replay of a repeated device-resident search -----------------------------------------------------------
// A small batch is a chain of ~9 (flat) / ~18 (IVF) short dependent dispatches.  With option "graph" = 1 the first call
// of a (buffers, shape, stream) combination runs eagerly (it sizes the workspace: allocation is illegal while capturing),
// the second is captured into a graph and launched, later ones replay the graph.  Anything that changes the index or an
// option drops the graph (graph_reset).
void graph_reset(vdb_index_s *h) {
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    h->graph_exec = nullptr;
    h->graph_key = vdb_index_s::GraphKey{};
    h->graph_warm = vdb_index_s::GraphKey{};
}

constexpr int64_t kGraphMaxQueries = 4096;

template <class F>
void graph_or_run(vdb_index_s *h, const vdb_index_s::GraphKey &key, F &&run) {
    const bool eligible = h->graph_mode && key.st != nullptr && !h->timing && key.nq > 0 && key.nq <= kGraphMaxQueries;
    if (!eligible) {
        run();
        return;
    }
    if (h->graph_exec && key == h->graph_key && h->graph_epoch == g_alloc_epoch.load(std::memory_order_relaxed)) {
        VDB_HIP(hipGraphLaunch(h->graph_exec, key.st));
        ++h->graph_replays;
        return;
    }
    if (h->graph_exec) {      // another shape, or a buffer moved since the capture (the graph holds raw addresses): drop it
        (void)hipGraphExecDestroy(h->graph_exec);
        h->graph_exec = nullptr;
        h->graph_key = vdb_index_s::GraphKey{};
    }
    if (!(key == h->graph_warm)) {
        run();
        h->graph_warm = key;
        return;
    }
    const vdb_index_s::GraphKey warm = h->graph_warm;
    graph_reset(h);
    VDB_HIP(hipStreamBeginCapture(key.st, hipStreamCaptureModeThreadLocal));
    hipGraph_t g = nullptr;
    bool captured = true;
    try {
        run();
    } catch (...) {
        captured = false;
    }
    const hipError_t e_end = hipStreamEndCapture(key.st, &g);
    hipGraphExec_t ex = nullptr;
    if (captured && e_end == hipSuccess && g && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess) {
        (void)hipGraphDestroy(g);
        h->graph_exec = ex;
        h->graph_key = key;
        h->graph_warm = warm;
        h->graph_epoch = g_alloc_epoch.load(std::memory_order_relaxed);
        VDB_HIP(hipGraphLaunch(ex, key.st));
        ++h->graph_replays;
        return;
    }
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    run();          // the capture did not work out (e.g. a workspace had to grow): this call runs eagerly, the next one warms up again
}

void search_device_impl(vdb_index_s *h, const float *dq, int64_t nq, int k, float *D, int64_t *I, double *pk,
                        int64_t *pi, hipStream_t st) {
    if (!h->built) throw Error(VDB_ERR_STATE, "Index has not been built yet.");
    if (k < 1 || k > 2048) throw Error(VDB_ERR_INVALID, "k must be in [1, 2048]");
    if (nq < 0) throw Error(VDB_ERR_INVALID, "negative query count");
    h->last.last_nq = nq;
    if (nq == 0) return;
    if (!dq) throw Error(VDB_ERR_INVALID, "null query pointer");
    if (h->N == 0) {  // empty shard: all padding
        MergeArgs ma{};
        ma.nparts = 0;
        ma.k = k;
        ma.metric = h->metric;
        ma.count = nq;
        ma.D = D;
        ma.I = I;
        ma.okeys = pk;
        ma.oids = pi;
        launch_merge(ma, nq, st);
        h->last.last_path = VDB_PATH_EXACT_SCAN;
        return;
    }
    // queries per pass: the level-1 bin arrays cost 8 bytes per (256-row bin, query); keep them within kBinBudget so
    // that a large batch on a large shard is served in several passes instead of failing with VDB_ERR_NOMEM
    int64_t kBatch = 16384;
    {
        const int64_t nbins = std::max<int64_t>(1, h->Npad / kBinRows);
        const int64_t fit = (int64_t)(kBinBudget / (8.0 * (double)nbins)) / 512 * 512;
        kBatch = std::max<int64_t>(512, std::min<int64_t>(kBatch, fit));
    }
    h->ws.small.reserve(kSmallBytes);
    VDB_HIP(hipMemsetAsync(h->ws.small.p, 0, kSmallBytes, st));
    h->small_is_clean = true;
    const size_t ev_mark = h->ev_used;
    try {
        for (int64_t b0 = 0; b0 < nq; b0 += kBatch) {
            const int64_t nb = std::min<int64_t>(kBatch, nq - b0);
            search_batch(h, dq + (size_t)b0 * h->dim, nb, k, D ? D + (size_t)b0 * k : nullptr,
                         I ? I + (size_t)b0 * k : nullptr, pk ? pk + (size_t)b0 * k : nullptr,
                         pi ? pi + (size_t)b0 * k : nullptr, st);
        }
    } catch (...) {
        h->ev_used = ev_mark;      // events of a failed search were never all recorded: do not leave them to vdb_stats
        h->small_is_clean = false;
        throw;
    }
    h->small_is_clean = false;
}

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
        set_device(h->device);
        (void)hipDeviceSynchronize();
        DevBuf *all[] = {&h->x32, &h->xnorm2, &h->panels, &h->bias, &h->stats, &h->panels8, &h->bias8, &h->rows8, &h->rowstat8, &h->ivf_offsets, &h->ivf_ids,
                         &h->ivf_probe_d, &h->ivf_probe_i, &h->ivf_list_pspan0, &h->ivf_span_row0, &h->ivf_span_valid,
                         &h->ivf_cnt, &h->ivf_slot_off, &h->ivf_list_item0, &h->ivf_item_list,
                         &h->ivf_item_slot0, &h->ivf_item_bin0, &h->ivf_plan, &h->ivf_slot_of};
        for (auto b : all) b->release();
        graph_reset(h);
        if (h->coarse) (void)vdb_destroy(h->coarse);
        h->ws.release();
        for (int i = 0; i < 2; ++i) {
            if (h->pin[i]) (void)hipHostFree(h->pin[i]);
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
        set_device(h->device);
        build_index(h, x_host, false, n, id_base, nullptr);
    });
}

int vdb_add_device(vdb_handle hh, const float *x_dev, int64_t n, int64_t id_base, void *stream) {
    return guarded([&] {
        auto *h = check(hh);
        if (n > 0 && !x_dev) throw Error(VDB_ERR_INVALID, "null corpus pointer");
        set_device(h->device);
        build_index(h, x_dev, true, n, id_base, as_stream(stream));
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
void rerank_device_impl(vdb_index_s *h, const float *dq, int64_t nq, const int64_t *cand, int ncand, int k, float *D,
                        int64_t *I, hipStream_t st) {
    if (!h->built) throw Error(VDB_ERR_STATE, "Index has not been built yet.");
    if (k < 1 || k > 2048) throw Error(VDB_ERR_INVALID, "k must be in [1, 2048]");
    if (nq < 0 || ncand < 0) throw Error(VDB_ERR_INVALID, "negative size");
    if (nq == 0) return;
    if (!dq || !D || !I || (ncand > 0 && !cand)) throw Error(VDB_ERR_INVALID, "null pointer");
    const float *qpad = dq;
    if (h->D4 != h->dim) {
        h->ws.qpad.reserve((size_t)nq * h->D4 * sizeof(float));
        pad_rows_kernel<<<dim3((unsigned)((nq * h->D4 + 255) / 256)), dim3(256), 0, st>>>(dq, nq, h->dim, h->D4,
                                                                                        h->ws.qpad.as<float>());
        qpad = h->ws.qpad.as<float>();
    }
    RerankArgs a{};
    a.c = RefineCommon{h->x32.as<float>(), qpad, h->N, h->id_base, h->D4, h->metric, k, nullptr};
    a.nq = nq;
    a.cand = cand;
    a.ncand = ncand;
    a.D = D;
    a.I = I;
    const int kpl = kpl_for(k);
    DISPATCH_KPL(kpl, (rerank_kernel<KPL><<<dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st>>>(a)));
    VDB_HIP(hipGetLastError());
}
}  // namespace

int vdb_rerank_device(vdb_handle hh, const float *q_dev, int64_t nq, const int64_t *cand_dev, int ncand, int k,
                      float *D_dev, int64_t *I_dev, void *stream) {
    return guarded([&] {
        auto *h = check(hh);
        set_device(h->device);
        rerank_device_impl(h, q_dev, nq, cand_dev, ncand, k, D_dev, I_dev, as_stream(stream));
    });
}

int vdb_rerank(vdb_handle hh, const float *q_host, int64_t nq, const int64_t *cand_host, int ncand, int k, float *D,
               int64_t *I) {
    return guarded([&] {
        auto *h = check(hh);
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
        set_device(h->device);
        vdb_stats_t s = h->last;
        s.ntotal = h->N;
        s.dim = h->dim;
        s.metric = h->metric;
        s.corpus_fp16_exact = h->corpus_fp16_exact ? 1 : 0;
        s.bytes_resident = (int64_t)(h->x32.cap + h->xnorm2.cap + h->panels.cap + h->bias.cap + h->stats.cap +
                                     h->panels8.cap + h->bias8.cap + h->rows8.cap + h->rowstat8.cap + h->ws.bytes());
        s.has_i8_copy = h->i8_ok ? 1 : 0;
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
        s.last_scan_ms = s.last_total_ms = 0.f;
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
            double scan = 0.0, total = 0.0;
            for (size_t i = 0; i < h->ev_used; ++i) {
                float ms = 0.f;
                VDB_HIP(hipEventSynchronize(h->ev_total[2 * i + 1]));
                VDB_HIP(hipEventElapsedTime(&ms, h->ev_scan[2 * i], h->ev_scan[2 * i + 1]));
                scan += ms;
                VDB_HIP(hipEventElapsedTime(&ms, h->ev_total[2 * i], h->ev_total[2 * i + 1]));
                total += ms;
            }
            s.last_scan_ms = (float)(scan / h->ev_used);
            s.last_total_ms = (float)(total / h->ev_used);
        }
        *out = s;
    });
}

int vdb_debug_fetch_stamps(vdb_handle hh, unsigned long long *out_host, int64_t max_words, int64_t *nwords) {
    return guarded([&] {
        auto *h = check(hh);
        if (!out_host || !nwords) throw Error(VDB_ERR_INVALID, "null pointer");
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
        const std::string k(key);
        graph_reset(h);                        // (a captured search embodies the options it was captured under)
        if (k == "graph") {
            if (value != 0 && value != 1) throw Error(VDB_ERR_INVALID, "graph must be 0 or 1");
            h->graph_mode = (int)value;
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
        } else if (k == "panel_dtype") {    // 0 auto (int8 scan copy used when corpus and queries allow), 1 = fp16 scan only
            if (value != 0 && value != 1) throw Error(VDB_ERR_INVALID, "panel_dtype must be 0 or 1");
            h->i8_disable = (int)value;
        } else if (k == "upload_block_mb") {   // staging block of the row-block ingestion (0 = default 64 MiB)
            if (value < 0 || value > 4096) throw Error(VDB_ERR_INVALID, "upload_block_mb out of range");
            h->upload_block_mb = (int)value;
        } else if (k == "ivf_bt") {
            if (value != 0 && value != 4 && value != 16) throw Error(VDB_ERR_INVALID, "ivf_bt must be 0, 4 or 16");
            h->ivf_bt = (int)value;
        } else if (k == "ivf_part") {
            if (value < 0 || value > 1024) throw Error(VDB_ERR_INVALID, "ivf_part must be 0 (auto) or 1..1024 spans");
            h->ivf_part = (int)value;
        } else if (k == "ivf_min_batch") {
            if (value < 1 || value > 1e9) throw Error(VDB_ERR_INVALID, "ivf_min_batch must be >= 1");
            h->ivf_min_batch = (int)value;
        } else if (k == "ivf_nw") {
            if (value != 0 && value != 2 && value != 4 && value != 8) throw Error(VDB_ERR_INVALID, "ivf_nw must be 0, 2, 4 or 8");
            h->ivf_nw = (int)value;
        } else if (k == "small_batch") {       // 1 (default): finer chunks / narrower workgroups for batches <= 512 queries
            if (value != 0 && value != 1) throw Error(VDB_ERR_INVALID, "small_batch must be 0 or 1");
            h->small_batch_off = value == 0;
        } else if (k == "i8_group") {          // rows per select group of the int8 scan: 8 (octs, default) or 4 (quads)
            if (value != 4 && value != 8) throw Error(VDB_ERR_INVALID, "i8_group must be 4 or 8");
            h->i8_group = (int)value;
        } else if (k == "i8_variant") {
#ifdef VDB_ABLATIONS
            if (value < 0 || (((int)value) & 7) > 5 || value > 13) throw Error(VDB_ERR_INVALID, "i8_variant must be 0..5 (+8)");
#else
            if (value < 0 || value > 5) throw Error(VDB_ERR_INVALID, "i8_variant must be 0..5");
#endif
            h->i8_variant = (int)value;
        } else if (k == "kloop_qgroup") {
            if (value < 0 || value > 1024) throw Error(VDB_ERR_INVALID, "kloop_qgroup out of range");
            h->kloop_qgroup = (int)value;
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
