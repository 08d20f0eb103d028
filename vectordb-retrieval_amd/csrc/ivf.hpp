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

// ---- CSR build: xperm[i] = x[perm[i]], ids[i] = id_base + perm[i] ------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ x, const int32_t *__restrict__ perm,
                                                          int64_t n, int D4, int64_t id_base, float *__restrict__ xperm,
                                                          int64_t *__restrict__ ids) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = n * (D4 / 4);
    if (i >= total) return;
    const int64_t r = i / (D4 / 4);
    const int c = (int)(i - r * (D4 / 4));
    const int64_t src = perm[r];
    reinterpret_cast<float4 *>(xperm)[i] = reinterpret_cast<const float4 *>(x)[src * (D4 / 4) + c];
    if (c == 0) ids[r] = id_base + src;
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
