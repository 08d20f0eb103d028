// ivf.hpp -- IVF-Flat device kernels: list scan, row gather (CSR build) and the k-means centroid update.
//
// Index layout in HBM: vectors are PERMUTED so that every inverted list is one contiguous row range
// (CSR): list l = rows [offsets[l], offsets[l+1]) of `xperm`, original ids in `ids` (ascending inside a
// list, so the layout and every result are deterministic).  The coarse quantizer is a flat index over the
// centroids served by the same exact kernels as the brute-force path.
#pragma once
#include "common.hpp"
#include "refine.hpp"
#include "topk.hpp"

namespace vdb {

// ---- list scan: exact float64 scoring of the rows of the probed lists -------------------------------
struct IvfScanArgs {
    RefineCommon c;            // X = permuted rows, idmap = original ids
    const int64_t *offsets;    // [nlist+1]
    const int64_t *probes;     // [nq][nprobe] list ids from the coarse search (-1 = none)
    int64_t nq;
    int nprobe;
    int S;                     // waves per query; wave s takes probes s, s+S, ...
    const int32_t *only_flagged; // optional [nq]: process only queries whose flag is set (fallback pass)
    float *D;                  // S == 1: final rows
    int64_t *I;
    double *pkeys;             // else partials [nq][S][k]
    int64_t *pids;
};

template <int KPL>
__global__ __launch_bounds__(256) void ivf_scan_kernel(IvfScanArgs a) {
    const int64_t u = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (u >= a.nq * a.S) return;
    const int64_t q = u / a.S;
    const int split = (int)(u - q * a.S);
    if (a.only_flagged && !a.only_flagged[q]) return;
    const float *qptr = a.c.Q + (size_t)q * a.c.D4;
    WaveTopK<KPL> tk;
    tk.init(a.c.k);
    for (int p = split; p < a.nprobe; p += a.S) {
        const int64_t l = a.probes[(size_t)q * a.nprobe + p];
        if (l < 0) continue;
        scan_rows<KPL>(tk, a.c, qptr, a.offsets[l], a.offsets[l + 1]);
    }
    if (a.D) {
        const size_t o = (size_t)q * a.c.k;
        write_topk<KPL>(tk, a.c.metric, a.D + o, a.I + o, nullptr, nullptr);
    } else {
        const size_t o = (size_t)u * a.c.k;
        write_topk<KPL>(tk, a.c.metric, nullptr, nullptr, a.pkeys + o, a.pids + o);
    }
}

// ---- flagged queries of the list-major path: exact list scan, split over as many waves as the chip has to spare ------
// The select flags a query whose work lists overflowed (or whose scales were unusable); how many there are is known
// only on the device.  One wave per flagged query (the first version) left a single straggler walking all its probed
// lists alone: 2.8 ms for 32 lists of 384-dim rows behind a 0.6 ms scan.  Here a fixed grid reads the count and cuts
// every flagged query into S = min(waves / count, max_split) splits (wave s takes probes s, s + S, ...); the S partial
// lists of a query are merged by whichever of its waves arrives LAST (agent-scope release / acquire around a per-query
// arrival counter), so there is no merge dispatch, and with S = 1 (many flagged queries) a wave writes its result
// directly.  Zero flagged queries: every wave returns after one load.
struct IvfFallbackArgs {
    RefineCommon c;
    const int64_t *offsets;      // [nlist+1]
    const int64_t *probes;       // [nq][nprobe]
    int nprobe;
    const int32_t *fb_list;      // [count] flagged queries (compacted by ivf_select_kernel)
    const int32_t *fb_count;     // [1]
    int max_split;               // <= nprobe
    int64_t cap_units;           // capacity of the partial buffers in (query, split) units, >= the number of queries
    double *pkeys;               // [cap_units][k]
    int64_t *pids;
    int32_t *done;               // [nq] arrival counters, zeroed per batch
    float *D;                    // final rows, or (D == nullptr) per-shard partial rows
    int64_t *I;
    double *okeys;
    int64_t *oids;
};

template <int KPL>
__device__ __forceinline__ void ivf_fallback_body(const IvfFallbackArgs &a, unsigned block, unsigned nblocks) {
    const int64_t count = *a.fb_count;
    if (count <= 0) return;
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)nblocks * 4;
    const int64_t wave0 = __builtin_amdgcn_readfirstlane((int)(block * 4 + (threadIdx.x >> 6)));
    int64_t S = waves / count;
    if (S > a.max_split) S = a.max_split;
    if (S > a.cap_units / count) S = a.cap_units / count;
    if (S < 1) S = 1;
    for (int64_t u = wave0; u < count * S; u += waves) {
        const int64_t f = u / S;
        const int split = (int)(u - f * S);
        const int64_t q = a.fb_list[f];
        const float *qptr = a.c.Q + (size_t)q * a.c.D4;
        WaveTopK<KPL> tk;
        tk.init(a.c.k);
        for (int p = split; p < a.nprobe; p += (int)S) {
            const int64_t l = a.probes[(size_t)q * a.nprobe + p];
            if (l < 0) continue;
            scan_rows<KPL, 4>(tk, a.c, qptr, a.offsets[l], a.offsets[l + 1]);      // (4 loads in flight: see refine_fallback_body)
        }
        const size_t oq = (size_t)q * a.c.k;
        if (S == 1) {
            if (a.D) write_topk<KPL>(tk, a.c.metric, a.D + oq, a.I + oq, nullptr, nullptr);
            else write_topk<KPL>(tk, a.c.metric, nullptr, nullptr, a.okeys + oq, a.oids + oq);
            continue;
        }
        write_topk<KPL>(tk, a.c.metric, nullptr, nullptr, a.pkeys + (size_t)u * a.c.k, a.pids + (size_t)u * a.c.k);
        __threadfence();                                    // release: this lane's partial rows before the arrival below
        int last = 0;
        if (lane == 0) last = atomicAdd(&a.done[f], 1) == (int)S - 1;
        last = __builtin_amdgcn_readfirstlane(last);
        if (!last) continue;
        __threadfence();                                    // acquire: the other waves' partial rows
        tk.init(a.c.k);
        const int total = (int)S * a.c.k;
        const size_t o0 = (size_t)f * S * a.c.k;            // the S partial rows of query f are consecutive units
        for (int base = 0; base < total; base += 64) {
            const int i = base + lane;
            bool valid = i < total;
            uint64_t key = ~0ull;
            int64_t id = -1;
            if (valid) {
                id = __builtin_nontemporal_load(a.pids + o0 + i);
                key = sortable_u64(__builtin_nontemporal_load(a.pkeys + o0 + i));
                valid = id >= 0;
            }
            tk.offer(key, id, valid);
        }
        if (a.D) write_topk<KPL>(tk, a.c.metric, a.D + oq, a.I + oq, nullptr, nullptr);
        else write_topk<KPL>(tk, a.c.metric, nullptr, nullptr, a.okeys + oq, a.oids + oq);
    }
}

// the tail of an IVF search in one launch (as refine_tail_kernel): fallback workgroups first, then the work lists
template <int KPL>
__global__ __launch_bounds__(256, (KPL == 1 ? 6 : KPL == 2 ? 5 : 1)) void ivf_tail_kernel(RefineListArgs la, IvfFallbackArgs fa, unsigned fb_blocks) {
    if (blockIdx.x < fb_blocks) ivf_fallback_body<KPL>(fa, blockIdx.x, fb_blocks);
    else refine_list_body<KPL>(la, blockIdx.x - fb_blocks);
}

// ---- CSR build: xperm[i] = x[perm[i]], ids[i] = id_base + perm[i] (src_ids: ids[i] = src_ids[perm[i]], the append) ---
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ x, const int32_t *__restrict__ perm,
                                                          int64_t n, int D4, int64_t id_base,
                                                          const int64_t *__restrict__ src_ids, float *__restrict__ xperm,
                                                          int64_t *__restrict__ ids) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = n * (D4 / 4);
    if (i >= total) return;
    const int64_t r = i / (D4 / 4);
    const int c = (int)(i - r * (D4 / 4));
    const int64_t src = perm[r];
    reinterpret_cast<float4 *>(xperm)[i] = reinterpret_cast<const float4 *>(x)[src * (D4 / 4) + c];
    if (c == 0) ids[r] = src_ids ? src_ids[src] : id_base + src;
}

// ---- CSR build on the device: stable counting sort of the rows by list ---------------------------------------------
// perm = rows grouped by list, ascending row number inside a list (the layout every result depends on is deterministic),
// offsets = exclusive prefix of the list sizes.  Rows are cut into chunks of `chunk_rows`:
//   csr_hist      per-chunk histogram of the lists (LDS), written as chunk_hist[chunk][list]
//   csr_colscan   per list: exclusive scan over the chunks (in place) + the list total
//   csr_offsets   exclusive scan of the totals (int64)
//   csr_scatter   one wave per chunk walks its rows in order, 64 at a time: lanes holding the same list get consecutive
//                 positions behind the chunk's base and the rows already placed (LDS counter per list)
// The host loop it replaces (two serial passes over N with random access) dominated IVF builds beyond a few million rows
// and ran once per k-means iteration.  nlist <= kCsrMaxLists (the per-list counters live in LDS).
constexpr int kCsrMaxLists = 8192;

__global__ __launch_bounds__(256) void csr_hist_kernel(const int64_t *__restrict__ assign, int64_t n, int nlist,
                                                       int chunk_rows, int32_t *__restrict__ chunk_hist,
                                                       int32_t *__restrict__ bad) {
    extern __shared__ int csr_lds[];
    for (int l = threadIdx.x; l < nlist; l += 256) csr_lds[l] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * chunk_rows;
    const int64_t r1 = min(n, r0 + chunk_rows);
    for (int64_t i = r0 + threadIdx.x; i < r1; i += 256) {
        const int64_t l = assign[i];
        if (l < 0 || l >= nlist) *bad = 1;
        else atomicAdd(&csr_lds[l], 1);
    }
    __syncthreads();
    for (int l = threadIdx.x; l < nlist; l += 256) chunk_hist[(size_t)blockIdx.x * nlist + l] = csr_lds[l];
}

__global__ __launch_bounds__(256) void csr_colscan_kernel(int32_t *__restrict__ chunk_hist, int nchunks, int nlist,
                                                          int32_t *__restrict__ total) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= nlist) return;
    int run = 0;
    for (int c = 0; c < nchunks; ++c) {
        const int v = chunk_hist[(size_t)c * nlist + l];
        chunk_hist[(size_t)c * nlist + l] = run;
        run += v;
    }
    total[l] = run;
}

__global__ __launch_bounds__(1024) void csr_offsets_kernel(const int32_t *__restrict__ total, int nlist,
                                                           int64_t *__restrict__ offsets) {
    __shared__ long long s_v[1024];
    __shared__ long long s_base;
    const int tid = threadIdx.x;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int l0 = 0; l0 < nlist; l0 += 1024) {
        const int l = l0 + tid;
        const long long v = l < nlist ? total[l] : 0;
        s_v[tid] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const long long t = tid >= o ? s_v[tid - o] : 0;
            __syncthreads();
            s_v[tid] += t;
            __syncthreads();
        }
        if (l < nlist) offsets[l] = s_base + s_v[tid] - v;
        __syncthreads();
        if (tid == 1023) s_base += s_v[1023];
        __syncthreads();
    }
    if (tid == 0) offsets[nlist] = s_base;
}

__global__ __launch_bounds__(64) void csr_scatter_kernel(const int64_t *__restrict__ assign, int64_t n, int nlist,
                                                         int chunk_rows, const int32_t *__restrict__ chunk_base,
                                                         const int64_t *__restrict__ offsets, int32_t *__restrict__ perm) {
    extern __shared__ int csr_lds[];          // rows of this chunk already placed, per list
    const int lane = threadIdx.x;
    for (int l = lane; l < nlist; l += 64) csr_lds[l] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * chunk_rows;
    const int64_t r1 = min(n, r0 + chunk_rows);
    const int32_t *cb = chunk_base + (size_t)blockIdx.x * nlist;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int64_t base = r0; base < r1; base += 64) {
        const int64_t row = base + lane;
        const int64_t lraw = row < r1 ? assign[row] : -1;
        const bool valid = lraw >= 0 && lraw < nlist;           // (rows without a list were flagged by csr_hist: the host throws)
        const int l = valid ? (int)lraw : -1;
        unsigned long long todo = __ballot(valid);
        while (todo) {                                        // one round per distinct list among the 64 rows
            const int leader = __ffsll((long long)todo) - 1;
            const int lv = __shfl(l, leader);
            const unsigned long long m = __ballot(valid && l == lv);
            const int placed = csr_lds[lv];                   // (wave-uniform address: broadcast read)
            if (valid && l == lv) perm[offsets[lv] + cb[lv] + placed + __popcll(m & lt_mask)] = (int32_t)row;
            if (lane == leader) csr_lds[lv] = placed + __popcll(m);
            todo &= ~m;
        }
    }
}

// ---- k-means update: centroid = mean of its points, float64 accumulation in list order (deterministic) ---
// one wave per centroid; lane d handles dims d, d+64, ...
__global__ __launch_bounds__(256) void centroid_update_kernel(const float *__restrict__ x, const int32_t *__restrict__ perm,
                                                              const int64_t *__restrict__ offsets, int nlist, int D,
                                                              int D4, int spherical, float *__restrict__ centroids) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nlist) return;
    const int64_t lo = offsets[c], hi = offsets[c + 1];
    if (hi <= lo) return;  // empty cluster: the host re-seeds it
    double n2 = 0.0;
    for (int d0 = 0; d0 < D; d0 += 64) {
        const int d = d0 + lane;
        double acc = 0.0;
        if (d < D)
            for (int64_t i = lo; i < hi; ++i) acc += (double)x[(size_t)perm[i] * D4 + d];
        const double m = acc / (double)(hi - lo);
        if (d < D) centroids[(size_t)c * D + d] = (float)m;
        double sq = (d < D) ? (double)(float)m * (double)(float)m : 0.0;
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        n2 += sq;
    }
    if (spherical && n2 > 0.0) {
        const float inv = (float)(1.0 / sqrt(n2));
        for (int d = lane; d < D; d += 64) centroids[(size_t)c * D + d] *= inv;
    }
}

}  // namespace vdb
