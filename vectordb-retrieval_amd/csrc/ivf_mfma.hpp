// ivf_mfma.hpp -- list-major IVF-Flat scan on the matrix cores.
//
// The query-major exact list scan (ivf.hpp) re-reads every probed list once per query: at nprobe = 128 that is
// 640 GB of L2/HBM traffic per 10k-query batch.  Here the (query -> lists) relation is inverted on the device
// into (list -> query slots), so each list's fp16 panels are streamed ONCE per group of up to 512 queries and
// multiplied against them with the same scan_kernel as the flat index (ITEMS mode, 64-row bins).  A per-query
// select over the bins of its probed lists nominates candidate rows / bins to re-scan, and the exact float64
// refine kernel produces the final answer: the result is bit-identical to the exact list scan.
//
// Panel space: every list is padded to whole spans of kIvfSpanRows = 256 rows (common.hpp); span_row0 / span_valid map a
// panel span back to the permuted row range it covers.
#pragma once
#include "common.hpp"
#include "prep.hpp"
#include "scan.hpp"

namespace vdb {

// tiles per level-1 bin: 4 (64-row bins) when few lists are probed, so that a query still owns several
// times k bins; 16 (256-row bins, 4x fewer entries for the select) when nprobe is large.

// ---- build: panels + bias over the list-padded panel space ------------------------------------------
__global__ __launch_bounds__(256) void ivf_build_panels_kernel(const float *__restrict__ X, int D, int D4, int ksteps,
                                                               int64_t ntiles, float sx,
                                                               const int32_t *__restrict__ span_row0,
                                                               const int32_t *__restrict__ span_valid,
                                                               half8 *__restrict__ panels, IndexStats *st) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(gid & 63);
    const int64_t tk = gid >> 6;
    const int ks = (int)(tk % ksteps);
    const int64_t tile = tk / ksteps;
    int inexact = 0;
    if (tile < ntiles) {
        const int rho = lane & 31, kh = lane >> 5;
        const int r = (rho & 3) | ((rho >> 3) << 2), h = (rho >> 2) & 1;
        const int64_t span = tile / kIvfTilesPerSpan;
        const int t = (int)(tile - span * kIvfTilesPerSpan);
        const int local = h * (kIvfSpanRows / 2) + t * 16 + r;
        const bool valid = local < span_valid[span];
        const int64_t row = (int64_t)span_row0[span] + local;
        const int d0 = ks * 16 + kh * 8;
        half8 out;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = d0 + j;
            float v = 0.f;
            if (valid && d < D) v = X[(size_t)row * D4 + d] * sx;
            const _Float16 hv = (_Float16)v;
            inexact |= ((float)hv != v);
            out[j] = hv;
        }
        panels[gid] = out;
    }
    if (__any(inexact) && (threadIdx.x & 63) == 0) atomic_set_flag(&st->not_fp16_exact);
}

// (bias[i] belongs to local row i % span_rows of panel span i / span_rows: both panel layouts map their MFMA rows to
//  local rows such that the accumulator init of a lane is a run of consecutive floats)
__global__ __launch_bounds__(256) void ivf_build_bias_kernel(const float *__restrict__ xnorm2, int64_t nspans, int span_rows,
                                                             int metric, const int32_t *__restrict__ span_row0,
                                                             const int32_t *__restrict__ span_valid,
                                                             float *__restrict__ bias) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nspans * span_rows) return;
    const int64_t span = i / span_rows;
    const int local = (int)(i - span * span_rows);
    bias[i] = (local < span_valid[span]) ? (metric == 0 ? xnorm2[span_row0[span] + local] : 0.f) : kPadBias;
}

// bins a probed list of `spans` panel spans contributes per query slot (and the stride between slots in the bin arrays)
__host__ __device__ inline int ivf_bins_of_list(int spans, int bins_per_span, int run_groups) {
    if (!run_groups) return spans * bins_per_span;
    return run_groups * ((spans * (bins_per_span / run_groups) + 3) & ~3);
}

// ---- per batch: invert (query, probe) -> (list, slot) ---------------------------------------------------
struct IvfPlan {          // device-resident scalars written by ivf_plan_kernel
    int32_t n_items;
    int32_t n_slots;
    int32_t n_bins;
    int32_t overflow;     // plan did not fit the buffers: the batch takes the exact list scan
    unsigned long long rows_scanned;   // (query, row) pairs of this batch: sum over lists of probes x rows
};

// per-list probe counts.  Workgroup-local LDS histogram first (nlist <= kIvfLdsLists), then one global atomic
// per (workgroup, touched list): 1.28 M probes on 1024 lists would otherwise serialise on 1024 addresses.
constexpr int kIvfLdsLists = 8192;
constexpr int kIvfPairsPerBlock = 4096;

// PPB: (query, probe) pairs per workgroup -- kIvfPairsPerBlock for large batches (fewer global atomics per list), a
// quarter of it when that would leave most CUs idle (80 000 pairs = 20 workgroups).
// (a device function: the per-batch prep kernel below runs it in its last workgroups -- one dispatch instead of two)
__device__ __forceinline__ void ivf_count_body(int block, int PPB, const int64_t *__restrict__ probes, int64_t n, int nlist,
                                               int32_t *__restrict__ cnt, int *ivf_hist) {
    const bool use_lds = nlist <= kIvfLdsLists;
    if (use_lds) {
        for (int l = threadIdx.x; l < nlist; l += 256) ivf_hist[l] = 0;
        __syncthreads();
    }
    const int64_t i0 = (int64_t)block * PPB;
    for (int j = threadIdx.x; j < PPB; j += 256) {
        const int64_t i = i0 + j;
        if (i >= n) break;
        const int64_t l = probes[i];
        if (l >= 0 && l < nlist) atomicAdd(use_lds ? &ivf_hist[l] : &cnt[l], 1);
    }
    if (use_lds) {
        __syncthreads();
        for (int l = threadIdx.x; l < nlist; l += 256) {
            const int c = ivf_hist[l];
            if (c) atomicAdd(&cnt[l], c);
        }
    }
}

// one workgroup: slot ranges (padded to `group`), work items (list x group of slots), output bin blocks.
// Lists are handled 1024 at a time with a block-wide exclusive scan of (groups, bins).
__global__ __launch_bounds__(1024) void ivf_plan_kernel(const int32_t *__restrict__ cnt,
                                                        const int32_t *__restrict__ list_pspan0, int nlist, int group,
                                                        int bins_per_span, int run_groups, int max_items, int max_slots, int max_bins,
                                                        int32_t *__restrict__ slot_off, int32_t *__restrict__ list_item0,
                                                        int32_t *__restrict__ item_list, int32_t *__restrict__ item_slot0,
                                                        int32_t *__restrict__ item_bin0, IvfPlan *plan,
                                                        const int64_t *__restrict__ offsets) {
    __shared__ int s_g[1024], s_b[1024];
    __shared__ int s_items, s_bins, s_overflow;
    __shared__ unsigned long long s_rows;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_items = 0;
        s_bins = 0;
        s_overflow = 0;
        s_rows = 0;
    }
    __syncthreads();
    for (int l0 = 0; l0 < nlist; l0 += 1024) {
        const int l = l0 + tid;
        int g = 0, bins_per_item = 0;
        if (l < nlist) {
            const int c = cnt[l];
            const int spans = list_pspan0[l + 1] - list_pspan0[l];
            if (c > 0) atomicAdd(&s_rows, (unsigned long long)c * (unsigned long long)(offsets[l + 1] - offsets[l]));
            if (c > 0 && spans > 0) {
                g = (c + group - 1) / group;
                bins_per_item = ivf_bins_of_list(spans, bins_per_span, run_groups);
            }
        }
        s_g[tid] = g;
        s_b[tid] = g * bins_per_item;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {   // inclusive Hillis-Steele scan of both arrays
            const int vg = tid >= o ? s_g[tid - o] : 0;
            const int vb = tid >= o ? s_b[tid - o] : 0;
            __syncthreads();
            s_g[tid] += vg;
            s_b[tid] += vb;
            __syncthreads();
        }
        const int item0 = s_items + s_g[tid] - g;               // exclusive prefixes
        const int bin0 = s_bins + s_b[tid] - g * bins_per_item;
        if (l < nlist) {
            slot_off[l] = item0 * group;                         // every item owns `group` slots
            list_item0[l] = item0;
            for (int j = 0; j < g; ++j) {
                if (item0 + j < max_items) {
                    item_list[item0 + j] = l;
                    item_slot0[item0 + j] = (item0 + j) * group;
                    item_bin0[item0 + j] = bin0 + j * bins_per_item;
                } else {
                    s_overflow = 1;
                }
            }
        }
        __syncthreads();
        if (tid == 1023) {
            s_items += s_g[1023];
            s_bins += s_b[1023];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int items = s_items, bins = s_bins, slots = items * group;
        slot_off[nlist] = slots;
        list_item0[nlist] = items;
        const int overflow = (s_overflow || slots > max_slots || bins > max_bins || items > max_items) ? 1 : 0;
        plan->n_items = overflow ? 0 : items;
        plan->n_slots = slots;
        plan->n_bins = bins;
        plan->overflow = overflow;
        plan->rows_scanned = s_rows;
    }
}

// slot of every (query, probe): workgroup-local rank from an LDS histogram + one global cursor bump per
// (workgroup, touched list).  Slot order inside a list is arbitrary (results do not depend on it).
template <int PPB>
__global__ __launch_bounds__(256) void ivf_scatter_kernel(const int64_t *__restrict__ probes, int64_t nq, int nprobe,
                                                          int nlist, const int32_t *__restrict__ slot_off,
                                                          const int32_t *__restrict__ list_pspan0,
                                                          int32_t *__restrict__ cursor, const IvfPlan *plan,
                                                          int32_t *__restrict__ slot_query, int32_t *__restrict__ slot_of) {
    extern __shared__ int ivf_hist[];   // [nlist] local counts, then global bases
    const bool use_lds = nlist <= kIvfLdsLists;
    const int64_t n = nq * nprobe;
    const int64_t i0 = (int64_t)blockIdx.x * PPB;
    const bool dead = plan->overflow != 0;
    if (use_lds) {
        for (int l = threadIdx.x; l < nlist; l += 256) ivf_hist[l] = 0;
        __syncthreads();
    }
    constexpr int PER = PPB / 256;
    int rank[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int64_t i = i0 + threadIdx.x + u * 256;
        rank[u] = -1;
        if (i < n && !dead) {
            const int64_t l = probes[i];
            // (an empty list owns no slots: probing it contributes nothing)
            if (l >= 0 && l < nlist && list_pspan0[l + 1] > list_pspan0[l])
                rank[u] = use_lds ? atomicAdd(&ivf_hist[l], 1) : atomicAdd(&cursor[l], 1);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int l = threadIdx.x; l < nlist; l += 256) {
            const int c = ivf_hist[l];
            ivf_hist[l] = c ? atomicAdd(&cursor[l], c) : 0;   // -> global base of this workgroup's run
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int64_t i = i0 + threadIdx.x + u * 256;
        if (i >= n) continue;
        int slot = -1;
        if (rank[u] >= 0) {
            const int64_t l = probes[i];
            slot = slot_off[l] + rank[u] + (use_lds ? ivf_hist[l] : 0);
            slot_query[slot] = (int32_t)(i / nprobe) + 1;     // (0 = padding slot)
        }
        slot_of[i] = slot;
    }
}

// Plan + scatter in ONE dispatch (nlist <= kIvfLdsLists): the plan is a prefix sum over the per-list counts -- a few
// microseconds of work that used to be a one-workgroup kernel of its own between the count and the scatter, i.e. a whole
// dependent dispatch on the critical path of every search.  Here EVERY scatter workgroup recomputes the slot prefix it needs
// (nlist counts -> groups per list -> exclusive scan, in LDS), and workgroup 0 also writes the plan's tables (work items,
// their bin blocks, the per-list offsets the select reads) and scalars.
template <int PPB>
__global__ __launch_bounds__(256) void ivf_plan_scatter_kernel(const int64_t *__restrict__ probes, int64_t nq, int nprobe,
                                                               int nlist, const int32_t *__restrict__ cnt,
                                                               const int32_t *__restrict__ list_pspan0, int group,
                                                               int bins_per_span, int run_groups, int max_items, int max_slots,
                                                               int max_bins, const int64_t *__restrict__ offsets,
                                                               int32_t *__restrict__ slot_off_g, int32_t *__restrict__ list_item0,
                                                               int32_t *__restrict__ item_list, int32_t *__restrict__ item_slot0,
                                                               int32_t *__restrict__ item_bin0, IvfPlan *plan,
                                                               int32_t *__restrict__ cursor, int32_t *__restrict__ slot_query,
                                                               int32_t *__restrict__ slot_of) {
    extern __shared__ int ivf_hist[];          // [nlist] local counts, then global bases | [nlist] first slot of every list
    int *s_slot_off = ivf_hist + nlist;
    __shared__ int s_part_g[256], s_part_b[256];
    __shared__ unsigned long long s_rows;
    const int tid = threadIdx.x;
    const int per = (nlist + 255) / 256;       // lists per thread: a contiguous run
    const int l0 = tid * per, l1 = min(nlist, l0 + per);
    if (tid == 0) s_rows = 0;
    int sum_g = 0, sum_b = 0;
    unsigned long long rows = 0;
    for (int l = l0; l < l1; ++l) {
        const int c = cnt[l];
        const int spans = list_pspan0[l + 1] - list_pspan0[l];
        const int g = (c > 0 && spans > 0) ? (c + group - 1) / group : 0;
        sum_g += g;
        sum_b += g * ivf_bins_of_list(spans, bins_per_span, run_groups);
        if (blockIdx.x == 0 && c > 0) rows += (unsigned long long)c * (unsigned long long)(offsets[l + 1] - offsets[l]);
        ivf_hist[l] = 0;
    }
    s_part_g[tid] = sum_g;
    s_part_b[tid] = sum_b;
    __syncthreads();
    if (blockIdx.x == 0 && rows) atomicAdd(&s_rows, rows);
    for (int o = 1; o < 256; o <<= 1) {        // inclusive Hillis-Steele scan of the 256 partial sums
        const int vg = tid >= o ? s_part_g[tid - o] : 0;
        const int vb = tid >= o ? s_part_b[tid - o] : 0;
        __syncthreads();
        s_part_g[tid] += vg;
        s_part_b[tid] += vb;
        __syncthreads();
    }
    const int items = s_part_g[255], bins = s_part_b[255], slots = items * group;
    const bool dead = slots > max_slots || bins > max_bins || items > max_items;
    int item0 = s_part_g[tid] - sum_g, bin0 = s_part_b[tid] - sum_b;     // exclusive prefixes of this thread's run
    for (int l = l0; l < l1; ++l) {
        const int c = cnt[l];
        const int spans = list_pspan0[l + 1] - list_pspan0[l];
        const int g = (c > 0 && spans > 0) ? (c + group - 1) / group : 0;
        s_slot_off[l] = item0 * group;
        if (blockIdx.x == 0) {
            slot_off_g[l] = item0 * group;
            list_item0[l] = item0;
            if (!dead)
                for (int j = 0; j < g; ++j) {
                    item_list[item0 + j] = l;
                    item_slot0[item0 + j] = (item0 + j) * group;
                    item_bin0[item0 + j] = bin0 + j * ivf_bins_of_list(spans, bins_per_span, run_groups);
                }
        }
        item0 += g;
        bin0 += g * ivf_bins_of_list(spans, bins_per_span, run_groups);
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) {
        slot_off_g[nlist] = slots;
        list_item0[nlist] = items;
        plan->n_items = dead ? 0 : items;
        plan->n_slots = slots;
        plan->n_bins = bins;
        plan->overflow = dead ? 1 : 0;
        plan->rows_scanned = s_rows;
    }
    // ---- scatter (as ivf_scatter_kernel, with the slot prefix from LDS) ----
    const int64_t n = nq * nprobe;
    const int64_t i0 = (int64_t)blockIdx.x * PPB;
    constexpr int PER = PPB / 256;
    int rank[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int64_t i = i0 + tid + u * 256;
        rank[u] = -1;
        if (i < n && !dead) {
            const int64_t l = probes[i];
            if (l >= 0 && l < nlist && list_pspan0[l + 1] > list_pspan0[l]) rank[u] = atomicAdd(&ivf_hist[l], 1);
        }
    }
    __syncthreads();
    for (int l = tid; l < nlist; l += 256) {
        const int c = ivf_hist[l];
        ivf_hist[l] = c ? atomicAdd(&cursor[l], c) : 0;   // -> global base of this workgroup's run
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int64_t i = i0 + tid + u * 256;
        if (i >= n) continue;
        int slot = -1;
        if (rank[u] >= 0) {
            const int64_t l = probes[i];
            slot = s_slot_off[l] + rank[u] + ivf_hist[l];
            slot_query[slot] = (int32_t)(i / nprobe) + 1;     // (0 = padding slot)
        }
        slot_of[i] = slot;
    }
}

// One dispatch for what the list scan needs from the query batch: the scales (when the coarse search of this batch has
// already taken the statistics of these queries: every workgroup finalises its own copy in LDS, workgroup 0 publishes
// it for the kernels that follow), the per-query error bounds, and the query rows the list scan gathers its B fragments
// from, [nq][Dpad]: scaled fp16, or int8 cq - q when the batch is on the int8 scan and q8 is given.  It replaces a
// one-thread finalize kernel + query_eps_kernel + ivf_qrows_kernel (three dependent ~4.5 us dispatches).
struct IvfPrepArgs {
    // workgroups [0, n_eps_blocks): error bounds; [n_eps_blocks, n_eps_blocks + n_row_blocks): query rows; the rest: the
    // per-list probe counts of the batch (ivf_count_body; dynamic LDS = nlist ints when nlist <= kIvfLdsLists)
    const int64_t *probes;
    int64_t npairs;
    int nlist, ppb;
    int32_t *cnt;
    unsigned n_row_blocks;
    EpsArgs eps;                  // (eps.info is replaced by the workgroup's copy)
    const float *Q;
    int64_t nq;
    int D, Dpad;
    _Float16 *qrows;
    signed char *q8;              // nullptr: no int8 list scan offered
    QueryBatchInfo *info;         // this index's batch info (published when from_src)
    const QueryBatchInfo *src;    // statistics of the same queries taken by the coarse index, or nullptr (info is final)
    FinalizeArgs fin;
    unsigned n_eps_blocks;
};
__global__ __launch_bounds__(256) void ivf_prep_kernel(IvfPrepArgs a) {
    extern __shared__ int ivf_hist[];
    __shared__ QueryBatchInfo s_info;
    if (blockIdx.x >= a.n_eps_blocks + a.n_row_blocks) {      // (uniform per workgroup)
        ivf_count_body((int)(blockIdx.x - a.n_eps_blocks - a.n_row_blocks), a.ppb, a.probes, a.npairs, a.nlist, a.cnt, ivf_hist);
        return;
    }
    const QueryBatchInfo *info = a.info;
    if (a.src) {
        if (threadIdx.x == 0) {
            s_info = *a.src;
            query_finalize_values(&s_info, a.fin, __uint_as_float(s_info.absmax_bits), s_info.not_integer, s_info.nonfinite,
                                  s_info.not_u8, s_info.not_s8);
            if (blockIdx.x == 0) *a.info = s_info;
        }
        __syncthreads();
        info = &s_info;
    }
    if (blockIdx.x < a.n_eps_blocks) {
        EpsArgs e = a.eps;
        e.info = info;
        query_eps_body((int64_t)blockIdx.x * 256 + threadIdx.x, e);
        return;
    }
    // query rows: 8 consecutive dims per thread (one 16-byte fp16 store / one 8-byte int8 store; Dpad is a multiple of 32)
    const int64_t i8 = ((int64_t)(blockIdx.x - a.n_eps_blocks) * 256 + threadIdx.x) * 8;
    if (i8 >= a.nq * a.Dpad) return;
    const int64_t q = i8 / a.Dpad;
    const int d0 = (int)(i8 - q * a.Dpad);
    const int mode = info->i8_mode;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (d0 + j < a.D) ? a.Q[(size_t)q * a.D + d0 + j] : 0.f;
    if (mode && a.q8) {
        const int cq = (mode & 3) == 1 ? 127 : -1;
        union { signed char c[8]; int2 w; } o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.c[j] = (signed char)(d0 + j < a.D ? cq - (int)v[j] : 0);
        *reinterpret_cast<int2 *>(a.q8 + i8) = o.w;
    } else {
        const float bs = info->bscale;
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (_Float16)(v[j] * bs);
        *reinterpret_cast<half8 *>(a.qrows + i8) = o;
    }
}

// ---- select over the bins of a query's probed lists ----------------------------------------------------
constexpr int kIvfMaxMinima = 5;
struct IvfSelectArgs {
    const float *bin_m[kIvfMaxMinima];   // the nm smallest quad minima of every bin (3: scan_kernel / scan_i8_kernel items mode,
    int nm;                              // 5: ivf_kloop_scan_kernel), ascending, each carrying its quad id
    const float *eps;
    const QueryBatchInfo *info;
    const IvfPlan *plan;
    const int64_t *probes;        // [nq][nprobe]
    const int32_t *slot_of;       // [nq][nprobe]
    const int32_t *slot_off, *list_item0, *item_bin0, *list_pspan0, *span_row0, *span_valid;
    int64_t nq;
    int nprobe, group, k, cand_cap, rescan_cap, max_entries;
    int vals_entries;             // LDS entries per wave for the bin minima: max_entries when k > 64, else 0 (see the kernel)
    int probe_cap;                // nprobe rounded up to 64: size of the per-probe LDS arrays
    int group_rows;               // rows per candidate group: 0 = by the scan's arithmetic (4, or 8 on the int8 scan); 1 / 2 / 4 given
    int act_cap;                  // active bins (first minimum <= tau1 + 2 eps) a wave lists in LDS: cand_cap + rescan_cap
    int bins_per_span, bin_rows;  // level-1 bins of a panel span and rows per bin: bin b of a span covers its local rows
                                  // [b bin_rows, (b + 1) bin_rows).  Entry e of a probed list:
    int run_groups;               //   0 (K-loop scan): [span][bin], e = span * bins_per_span + bin
                                  //   2 (32-row-tile scans): one run per lane half, [half][span][bin of the half], every run
                                  //   padded to a multiple of 4 entries (the padding holds +inf): see ivf_bins_of_list
    int32_t *cand_rows, *rescan_rows, *counts, *fallback;
    int32_t *fb_list, *fb_count;  // flagged queries, compacted for ivf_fallback_body (ivf.hpp, ivf_tail_kernel)
    unsigned long long *stat_counters;  // [3] candidates, rescans, fallback queries
};

// one wave per query.  Phase 1: lane p resolves probe p (list, item, slot column, #bins) -- the dependent
// index loads of all probes overlap instead of chaining; a wave prefix sum lays the probes' bins out as one
// flat entry range.  Phase 2: lanes stride over the flat entries (binary search probe-of-entry in LDS) and
// copy the bin minima into LDS with independent, per-probe-contiguous loads.  Phase 3: bitwise bisection for
// the k-th smallest.  Phase 4: entries <= That become candidate rows or bins to re-scan.
constexpr int kIvfMaxProbes = 512;
// 32-bit words of LDS per wave: vals[vals_entries] | p_off[probe_cap + 32] | p_base lo / hi [2 probe_cap] | p_list[probe_cap]
// (+ 32 spare) | act[act_cap] | vals2[act_cap * nm]
__host__ __device__ inline size_t ivf_select_lds_words(int vals_entries, int probe_cap, int act_cap, int nm) {
    return (size_t)vals_entries + 4 * (size_t)probe_cap + 64 + (size_t)act_cap * (1 + nm);
}

__global__ __launch_bounds__(128) void ivf_select_kernel(IvfSelectArgs a) {  // 2 waves per workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char ivf_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 2 + wave));
    if (q >= a.nq) return;
    // per wave: vals[max_entries] | p_off[probe_cap+32] | p_base lo/hi [2*probe_cap] | p_list[probe_cap] -- sized by
    // the launch (what this nprobe and the longest list can need), not by the compile-time maxima: the LDS footprint
    // is what limits how many queries a CU works on at once
    // (vals only for k > 64, whose threshold needs every bin minimum in LDS: a.vals_entries = max_entries, else 0 -- without
    //  it a wave needs ~2 KB instead of ~18 KB and eight times as many queries are in flight per CU; the second pass then
    //  re-reads the minima, L2-resident by now)
    unsigned *vals = reinterpret_cast<unsigned *>(ivf_smem) + (size_t)wave * ivf_select_lds_words(a.vals_entries, a.probe_cap, a.act_cap, a.nm);
    int *p_off = reinterpret_cast<int *>(vals + a.vals_entries);
    const bool keep = a.vals_entries > 0;
    unsigned *p_base_lo = reinterpret_cast<unsigned *>(p_off + a.probe_cap + 32);
    unsigned *p_base_hi = p_base_lo + a.probe_cap;
    int *p_list = reinterpret_cast<int *>(p_base_hi + a.probe_cap);
    unsigned *act = reinterpret_cast<unsigned *>(p_list + a.probe_cap);       // [act_cap] active bins: entry | probe << 16
    unsigned *vals2 = act + a.act_cap;                                          // [act_cap][nm] their minima (sortable keys)
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    bool fb = a.plan->overflow || a.info->force_fallback || a.nprobe > kIvfMaxProbes;
    int E = 0;
    if (!fb) {
        for (int p0 = 0; p0 < a.nprobe; p0 += 64) {
            const int p = p0 + lane;
            int nb = 0, lst = -1;
            size_t base = 0;
            if (p < a.nprobe) {
                const int64_t l = a.probes[(size_t)q * a.nprobe + p];
                const int slot = a.slot_of[(size_t)q * a.nprobe + p];
                if (l >= 0 && slot >= 0) {
                    const int rel = slot - a.slot_off[l];
                    const int item = a.list_item0[l] + rel / a.group, col = rel % a.group;
                    nb = ivf_bins_of_list(a.list_pspan0[l + 1] - a.list_pspan0[l], a.bins_per_span, a.run_groups);
                    base = (size_t)a.item_bin0[item] * a.group + (size_t)col * nb;
                    lst = (int)l;
                }
            }
            int incl = nb;                                  // inclusive wave prefix sum of nb
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            if (p < a.nprobe) {
                p_off[p] = E + incl - nb;
                p_base_lo[p] = (unsigned)base;
                p_base_hi[p] = (unsigned)(base >> 32);
                p_list[p] = lst;
            }
            E += __shfl(incl, 63);
        }
        if (lane == 0) p_off[a.nprobe] = E;
        if (E > a.max_entries) fb = true;
    }
    if (!fb && E < a.k) fb = true;            // not enough bins to bound the k-th neighbour: exact list scan
    auto probe_of = [&](int e) {              // largest p with p_off[p] <= e  (p_off is non-decreasing)
        int lo = 0, hi = a.nprobe;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (p_off[mid] <= e) lo = mid; else hi = mid;
        }
        return lo;
    };
    int ncand = 0, nres = 0;
    if (!fb) {
        unsigned low0 = 0xFFFFFFFFu, low1 = 0xFFFFFFFFu;     // the two smallest keys this lane has seen
        // (four entries per lane and round: their probe look-ups and loads overlap -- one entry per round left ONE global
        //  load in flight per lane, and at nprobe 128 (2000+ entries per query) this loop ran as long as the list scan)
        // (a lane's entries ascend by 64: its probe index only moves forward -- one binary search at the start, then a few
        //  steps along p_off per entry instead of seven LDS reads each; p_off[nprobe] = E ends the walk)
        int pcur = lane < E ? probe_of(lane) : 0;
        auto advance = [&](int e) {
            while (p_off[pcur + 1] <= e) ++pcur;
            return pcur;
        };
        for (int e0 = lane; e0 < E; e0 += 256) {
            float raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + 64 * u;
                raw[u] = __builtin_inff();
                if (e < E) {
                    const int p = advance(e);
                    const size_t base = ((size_t)p_base_hi[p] << 32) | p_base_lo[p];
                    raw[u] = a.bin_m[0][base + (e - p_off[p])];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + 64 * u;
                if (e < E) {
                    const unsigned key = sortable_u32(raw[u]);
                    if (keep) vals[e] = key;
                    low1 = min(low1, max(low0, key));
                    low0 = min(low0, key);
                }
            }
        }
        unsigned ans = 0;
        if (a.k <= 64) {
            // Threshold from the lanes' two smallest keys: the k-th smallest of these <= 128 distinct bins is >= the k-th
            // smallest bin minimum tau (equal unless three of the k best bins share a lane), and any U >= tau is a valid
            // threshold.  32 steps of two compares + ballots instead of 32 passes over all E entries in LDS.
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned trial = ans | ((1u << bit) - 1u);
                const int cnt = __popcll(__ballot(low0 <= trial)) + __popcll(__ballot(low1 <= trial));
                if (cnt < a.k) ans |= (1u << bit);
            }
        } else {
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned trial = ans | ((1u << bit) - 1u);
                int cnt = 0;
                for (int e = lane; e < E; e += 64) cnt += (vals[e] <= trial) ? 1 : 0;
                for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
                if (cnt < a.k) ans |= (1u << bit);
            }
        }
        const int i8_mode = a.info->i8_mode;                 // (read ONCE: in the loop below every use would be a fresh global load)
        const float epsq = a.eps[q];
        const float that1 = select_threshold(unsortable_f32(ans), epsq, i8_mode);
        const int grows = a.group_rows > 0 ? a.group_rows : group_rows_of(i8_mode);   // rows per candidate group
        if (!(that1 < 0.9e38f)) fb = true;
        int32_t *cr = a.cand_rows + (size_t)q * a.cand_cap;
        int32_t *rr = a.rescan_rows + (size_t)q * a.rescan_cap * 2;
        // ---- the ACTIVE bins (first minimum <= tau1 + 2 eps), compacted into LDS: (entry, probe) packed into one word ----
        // Everything below works on this short list (typically k .. 2k bins of the hundreds or thousands a query meets)
        // instead of running the whole emit logic over every entry.
        int nact = 0;
        pcur = lane < E ? probe_of(lane) : 0;
        for (int e4 = 0; e4 < E && !fb; e4 += 256) {
            float m1v[4];
            int pv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {      // the minima of four rounds of entries, requested together
                const int e = e4 + 64 * u + lane;
                m1v[u] = __builtin_inff();
                pv[u] = 0;
                if (e < E) {
                    pv[u] = advance(e);
                    if (keep) {
                        m1v[u] = unsortable_f32(vals[e]);
                    } else {
                        const size_t base = ((size_t)p_base_hi[pv[u]] << 32) | p_base_lo[pv[u]];
                        m1v[u] = a.bin_m[0][base + (e - p_off[pv[u]])];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e4 + 64 * u + lane;
                const bool on = e < E && m1v[u] <= that1;
                const unsigned long long am = __ballot(on);
                if (on) {
                    const int pos = nact + __popcll(am & lt_mask);
                    if (pos < a.act_cap) act[pos] = (unsigned)e | ((unsigned)pv[u] << 16);
                }
                nact += __popcll(am);
            }
        }
        if (nact > a.act_cap) fb = true;       // (more active bins than candidates + re-scans the work lists can take)
        // ---- tau2: the k-th smallest over ALL kept minima of the active bins.  Every minimum kept is the minimum of a
        // different quad, i.e. they are scores of distinct rows: k of them <= tau2 bound the k-th neighbour as the k-th
        // bin minimum does, but tighter -- the near neighbours of a query sit in one or two of its probed lists, several per
        // bin, and the k-th smallest FIRST minimum then lies far beyond the k-th smallest score (msmarco-shaped leg: 36
        // candidate quads and 1.1 re-scanned bins per query with tau1, 22 and 0.04 with tau2 at k = 20).  Values above
        // tau1 + 2 eps cannot be among the k smallest (>= k first minima are <= tau1), so the active list holds them all.
        float that = that1;
        if (!fb && a.nm > 1) {
            const int V = nact * a.nm;
            for (int v = lane; v < V; v += 64) {             // gather: value v = (active bin v / nm, minimum v % nm)
                const int ai = v / a.nm, j = v - ai * a.nm;
                const unsigned w = act[ai];
                const int e = (int)(w & 0xFFFFu), p = (int)(w >> 16);
                const size_t base = ((size_t)p_base_hi[p] << 32) | p_base_lo[p];
                vals2[v] = sortable_u32(a.bin_m[j][base + (e - p_off[p])]);
            }
            unsigned t2 = 0;
            if (a.k <= 32) {      // lanes' two smallest of their strided values (a bin's minima land on different lanes)
                unsigned l0 = 0xFFFFFFFFu, l1 = 0xFFFFFFFFu;
                for (int v = lane; v < V; v += 64) {
                    const unsigned key = vals2[v];
                    l1 = min(l1, max(l0, key));
                    l0 = min(l0, key);
                }
                for (int bit = 31; bit >= 0; --bit) {
                    const unsigned trial = t2 | ((1u << bit) - 1u);
                    const int cnt = __popcll(__ballot(l0 <= trial)) + __popcll(__ballot(l1 <= trial));
                    if (cnt < a.k) t2 |= (1u << bit);
                }
            } else {
                for (int bit = 31; bit >= 0; --bit) {
                    const unsigned trial = t2 | ((1u << bit) - 1u);
                    int cnt = 0;
                    for (int v = lane; v < V; v += 64) cnt += (vals2[v] <= trial) ? 1 : 0;
                    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
                    if (cnt < a.k) t2 |= (1u << bit);
                }
            }
            if (t2 < ans) that = select_threshold(unsortable_f32(t2), epsq, i8_mode);
        }
        // ---- emit: an active bin whose j smallest quad minima are active and whose (j + 1)-th is not yields those j quads;
        // when all nm minima kept are active (more quads may be) the whole bin is re-scanned -- as it is when a quad would run
        // past the end of its list
        for (int a0 = 0; a0 < nact && !fb; a0 += 64) {
            const int ai = a0 + lane;
            int ncq = 0;                       // candidate quads of this lane's bin
            bool resc = false;
            int row0 = 0, row1 = 0, crow[kIvfMaxMinima - 1] = {0, 0, 0, 0};
            if (ai < nact) {
                const unsigned w = act[ai];
                const int e = (int)(w & 0xFFFFu), p = (int)(w >> 16);
                const int ei = e - p_off[p];
                const size_t base = ((size_t)p_base_hi[p] << 32) | p_base_lo[p];
                float mv[kIvfMaxMinima];
#pragma unroll
                for (int i = 0; i < kIvfMaxMinima; ++i)
                    mv[i] = i >= a.nm ? __builtin_inff() : (a.nm > 1 ? unsortable_f32(vals2[ai * a.nm + i]) : a.bin_m[i][base + ei]);
                if (mv[0] <= that) {
                    const int l = p_list[p];
                    int span_local, bin;
                    if (a.run_groups) {      // (an active entry is never run padding: that holds +inf)
                        const int bpg = a.bins_per_span / a.run_groups;            // bins of a span in one run
                        const int run_len = ((a.list_pspan0[l + 1] - a.list_pspan0[l]) * bpg + 3) & ~3;
                        const int grp = ei / run_len, r = ei - grp * run_len;
                        span_local = r / bpg;
                        bin = grp * bpg + (r - span_local * bpg);
                    } else {
                        span_local = ei / a.bins_per_span;
                        bin = ei - span_local * a.bins_per_span;
                    }
                    const int pspan = a.list_pspan0[l] + span_local;
                    row0 = a.span_row0[pspan] + bin * a.bin_rows;
                    row1 = row0 + a.bin_rows;
                    const int end = a.span_row0[pspan] + a.span_valid[pspan];
                    if (row1 > end) row1 = end;
                    int active = 0;            // (the minima ascend: the active ones are a prefix)
#pragma unroll
                    for (int i = 0; i < kIvfMaxMinima; ++i) active += (mv[i] <= that) ? 1 : 0;
                    resc = active >= a.nm;
                    ncq = resc ? 0 : active;
#pragma unroll
                    for (int i = 0; i < kIvfMaxMinima - 1; ++i) {
                        crow[i] = row0 + (int)(__float_as_uint(mv[i]) & 0x3Fu) * grows;
                        if (i < ncq && crow[i] + grows > end) resc = true;
                    }
                    if (resc) {
                        ncq = 0;
                        resc = row1 > row0;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < kIvfMaxMinima - 1; ++i) {
                const bool ci = ncq > i;
                const unsigned long long cm = __ballot(ci);
                if (ci) {
                    const int pos = ncand + __popcll(cm & lt_mask);
                    if (pos < a.cand_cap) cr[pos] = crow[i];
                }
                ncand += __popcll(cm);
            }
            const unsigned long long rm = __ballot(resc);
            if (resc) {
                const int pos = nres + __popcll(rm & lt_mask);
                if (pos < a.rescan_cap) {
                    rr[2 * pos] = row0;
                    rr[2 * pos + 1] = row1;
                }
            }
            nres += __popcll(rm);
        }
        if (ncand > a.cand_cap || nres > a.rescan_cap) fb = true;
    }
    if (lane == 0) {
        a.counts[2 * q] = fb ? 0 : ncand;
        a.counts[2 * q + 1] = fb ? 0 : nres;
        a.fallback[q] = fb ? 1 : 0;
        if (fb) {
            a.fb_list[atomicAdd(a.fb_count, 1)] = (int)q;
            stat_add(a.stat_counters, q, 2, 1ull);
        } else {
            stat_add(a.stat_counters, q, 0, (unsigned long long)ncand);
            stat_add(a.stat_counters, q, 1, (unsigned long long)nres);
        }
    }
}

}  // namespace vdb
