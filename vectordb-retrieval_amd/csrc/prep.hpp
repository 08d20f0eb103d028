// prep.hpp -- build-time and per-batch preparation kernels (HBM-bound, run once per corpus / batch).
#pragma once
#include <algorithm>

#include "common.hpp"

namespace vdb {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// (every atomic below is guarded by a plain read of the current value: thousands of waves hitting ONE address with
//  atomics that would not change it serialise in L2 -- the flags and maxima settle after the first few waves)
__device__ __forceinline__ void atomic_max_bits(unsigned *addr, float v) {
    const unsigned b = __float_as_uint(fabsf(v));  // |v| >= 0: integer order == float order
    if (b > *reinterpret_cast<volatile unsigned *>(addr)) atomicMax(addr, b);
}
__device__ __forceinline__ void atomic_set_flag(int *addr) {
    if (*reinterpret_cast<volatile int *>(addr) == 0) atomicOr(addr, 1);
}

// ---- corpus statistics + exact row norms -----------------------------------------------------------
// one thread per row: ||x||^2 in float64 (rounded once to float32), max |x|, integer / finite flags.
__global__ __launch_bounds__(256) void corpus_stats_kernel(const float *__restrict__ X, int64_t N, int D4,
                                                           float *__restrict__ xnorm2, IndexStats *st) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float amax = 0.f;
    float n2 = 0.f;
    int nonfinite = 0, notint = 0, notu8 = 0, nots8 = 0;
    if (row < N) {
        const float4 *xv = reinterpret_cast<const float4 *>(X + (size_t)row * D4);
        double acc = 0.0;
        for (int i = 0; i < D4 / 4; ++i) {
            const float4 a = xv[i];
            const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = fma((double)v[j], (double)v[j], acc);
                amax = fmaxf(amax, fabsf(v[j]));
                nonfinite |= !(fabsf(v[j]) <= 3.402823466e+38f);
                notint |= (v[j] != rintf(v[j]));
                notu8 |= !(v[j] >= 0.f && v[j] <= 255.f);
                nots8 |= !(v[j] >= -128.f && v[j] <= 127.f);
            }
        }
        n2 = (float)acc;
        xnorm2[row] = n2;
    }
    // wave reduce, one atomic per wave
    for (int o = 32; o > 0; o >>= 1) {
        amax = fmaxf(amax, __shfl_xor(amax, o));
        n2 = fmaxf(n2, __shfl_xor(n2, o));
    }
    const int anyf = __any(nonfinite), anyi = __any(notint), anyu = __any(notu8 | notint), anys = __any(nots8 | notint);
    if ((threadIdx.x & 63) == 0) {
        atomic_max_bits(&st->absmax_bits, amax);
        atomic_max_bits(&st->maxnorm2_bits, n2);
        if (anyf) atomic_set_flag(&st->nonfinite);
        if (anyi) atomic_set_flag(&st->not_integer);
        if (anyu) atomic_set_flag(&st->not_u8);
        if (anys) atomic_set_flag(&st->not_s8);
    }
}

// ---- corpus panels: fp16 copy in MFMA A-fragment order (see common.hpp) ---------------------------
// one thread per (tile, kstep, lane): converts 8 consecutive dims of one corpus row.
__global__ __launch_bounds__(256) void build_panels_kernel(const float *__restrict__ X, int64_t N, int D, int D4,
                                                           int ksteps, int64_t ntiles, float sx,
                                                           half8 *__restrict__ panels, IndexStats *st, int x16 = 0) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(gid & 63);
    const int64_t tk = gid >> 6;
    const int ks = (int)(tk % ksteps);
    const int64_t tile = tk / ksteps;
    int inexact = 0;
    if (tile < ntiles) {
        const int64_t span = tile / kTilesPerSpan;
        const int t = (int)(tile - span * kTilesPerSpan);
        int64_t row;
        int d0;
        if (x16) {     // layout "x16" (scan_x16.hpp): piece v = 2 ks2 + rb, lane = (MFMA row m of block rb, 8-dim quarter of the 32-dim k-step)
            const int m = lane & 15;
            row = span * kSpanRows + (m >> 2) * 128 + t * 8 + (ks & 1) * 4 + (m & 3);
            d0 = (ks >> 1) * 32 + (lane >> 4) * 8;
        } else {
            const int rho = lane & 31, kh = lane >> 5;
            const int r = (rho & 3) | ((rho >> 3) << 2), h = (rho >> 2) & 1;
            row = span * kSpanRows + (int64_t)h * kBinRows + t * 16 + r;
            d0 = ks * 16 + kh * 8;
        }
        half8 out;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = d0 + j;
            float v = 0.f;
            if (row < N && d < D) v = X[(size_t)row * D4 + d] * sx;
            const _Float16 hv = (_Float16)v;
            inexact |= ((float)hv != v);
            out[j] = hv;
        }
        panels[gid] = out;
    }
    if (__any(inexact) && (threadIdx.x & 63) == 0) atomic_set_flag(&st->not_fp16_exact);
}

// ---- 16-row-tile panels for v_mfma_f32_16x16x32_f16 (layout "p16", see common.hpp) ------------------------
// one thread per (tile, 32-dim k-step, lane): lane l holds A row rho = l&15, dims 32*ks + 8*(l>>4) .. +7.
// Row mapping: span = 1024 rows = 64 tiles; tile t of span s, MFMA row rho = 4g+i  <->  corpus row
// 1024 s + 256 g + 4 t + i, so the lane group g = l>>4 of the C/D layout (col = l&15, rows 4g..4g+3) walks 256
// CONSECUTIVE corpus rows per span, one quad per tile.
// tile0 / panels: tiles [tile0, tile0 + ntiles) are written to panels[0 ...) -- the whole corpus at build time (tile0 = 0), or
// one slab of a streamed scan (option "stream_panels").  panels == nullptr: only the fp16-exactness flag is taken.
__global__ __launch_bounds__(256) void build_panels16_kernel(const float *__restrict__ X, int64_t N, int D, int D4,
                                                             int ks32, int64_t ntiles, float sx,
                                                             half8 *__restrict__ panels, IndexStats *st, int64_t tile0 = 0) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(gid & 63);
    const int64_t tk = gid >> 6;
    const int ks = (int)(tk % ks32);
    const int64_t tile = tile0 + tk / ks32;
    int inexact = 0;
    if (tk / ks32 < ntiles) {
        const int rho = lane & 15, kq = lane >> 4;
        const int g = rho >> 2, i = rho & 3;
        const int64_t span = tile / kTilesPerSpan16;
        const int t = (int)(tile - span * kTilesPerSpan16);
        const int64_t row = span * kSpanRows16 + (int64_t)g * kBinRows + 4 * t + i;
        const int d0 = ks * 32 + kq * 8;
        half8 out;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = d0 + j;
            float v = 0.f;
            if (row < N && d < D) v = X[(size_t)row * D4 + d] * sx;
            const _Float16 hv = (_Float16)v;
            inexact |= ((float)hv != v);
            out[j] = hv;
        }
        if (panels) panels[gid] = out;
    }
    if (st && __any(inexact) && (threadIdx.x & 63) == 0) atomic_set_flag(&st->not_fp16_exact);
}

// Slab conversion of a streamed scan (option "stream_panels"): the same p16 panels as build_panels16_kernel for tiles
// [tile0, tile0 + ntiles), written to panels[0 ...).  One WAVE per tile walking all its k-steps (the build kernel's one
// thread per 16 bytes of output means 4.7 M workgroups for a 12.5M x 768 shard: dispatch-bound at ~1 TB/s, 60 ms per pass --
// longer than a third of the scan it feeds): per k-step a wave reads 16 rows x 128 contiguous bytes as two 16-byte loads per
// lane and writes 1 KiB.
__global__ __launch_bounds__(256) void convert_slab16_kernel(const float *__restrict__ X, int64_t N, int D, int D4, int ks32,
                                                             int64_t tile0, int64_t ntiles, float sx,
                                                             half8 *__restrict__ panels) {
    const int lane = threadIdx.x & 63;
    const int64_t tl = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tl >= ntiles) return;
    const int64_t tile = tile0 + tl;
    const int rho = lane & 15, kq = lane >> 4;
    const int g = rho >> 2, i = rho & 3;
    const int64_t span = tile / kTilesPerSpan16;
    const int t = (int)(tile - span * kTilesPerSpan16);
    const int64_t row = span * kSpanRows16 + (int64_t)g * kBinRows + 4 * t + i;
    const bool live = row < N;
    const float *xr = X + (size_t)(live ? row : 0) * D4;
    half8 *out = panels + (size_t)tl * ks32 * 64 + lane;
#pragma unroll 4
    for (int ks = 0; ks < ks32; ++ks) {
        const int d0 = ks * 32 + kq * 8;
        half8 o;
        if (live && d0 + 8 <= D4) {          // (D4 is D padded to a multiple of 4 with zeros: whole float4s)
            const float4 a = *reinterpret_cast<const float4 *>(xr + d0), b = *reinterpret_cast<const float4 *>(xr + d0 + 4);
            o[0] = (_Float16)(a.x * sx); o[1] = (_Float16)(a.y * sx); o[2] = (_Float16)(a.z * sx); o[3] = (_Float16)(a.w * sx);
            o[4] = (_Float16)(b.x * sx); o[5] = (_Float16)(b.y * sx); o[6] = (_Float16)(b.z * sx); o[7] = (_Float16)(b.w * sx);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (_Float16)((live && d0 + j < D) ? xr[d0 + j] * sx : 0.f);
        }
        out[(size_t)ks * 64] = o;
    }
}

// bias (C-init of the MFMA accumulators): ||x||^2 for L2, 0 for IP, pad marker beyond N.
__global__ __launch_bounds__(256) void build_bias_kernel(const float *__restrict__ xnorm2, int64_t N, int64_t Npad,
                                                         int metric, float *__restrict__ bias) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Npad) bias[i] = (i < N) ? (metric == 0 ? xnorm2[i] : 0.f) : kPadBias;
}

// ---- queries ----------------------------------------------------------------------------------------
// scales of a query batch, chosen by ONE thread once the statistics are complete (the last workgroup of
// query_stats_kernel): power-of-two query scale, scan score scale, fallback flag, fp16 / int8 scan choice.
struct FinalizeArgs {
    float sx;                  // corpus scale of the fp16 copy
    int metric;
    int corpus_int_unscaled;
    float maxnorm2;
    int corpus_i8;             // 0 = fp16 scan only; else the index holds an int8 scan copy this search may use: 1 = select on
                               // quads, 5 = on octs (bit 2 is copied into QueryBatchInfo.i8_mode)
};
// scales from the five statistics already present in *info (any address space: the caller has made them visible)
__device__ inline void query_finalize_values(QueryBatchInfo *info, const FinalizeArgs &f, float amax, int not_integer,
                                             int nonfinite, int not_u8, int not_s8);
__device__ inline void query_finalize(QueryBatchInfo *info, const FinalizeArgs &f) {
    const float amax = __uint_as_float(__hip_atomic_load(&info->absmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const int not_integer = __hip_atomic_load(&info->not_integer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int nonfinite = __hip_atomic_load(&info->nonfinite, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int not_u8 = __hip_atomic_load(&info->not_u8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int not_s8 = __hip_atomic_load(&info->not_s8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    query_finalize_values(info, f, amax, not_integer, nonfinite, not_u8, not_s8);
}
__device__ inline void query_finalize_values(QueryBatchInfo *info, const FinalizeArgs &f, float amax, int not_integer,
                                             int nonfinite, int not_u8, int not_s8) {
    const float fm = (f.metric == 0) ? 2.f : 1.f;
    float sq = 1.f;
    const bool int_ok = f.corpus_int_unscaled && !not_integer && fm * amax <= 2048.f;
    if (!int_ok && amax > 0.f && amax <= 3.0e38f) {
        int e;
        frexpf(fm * amax, &e);          // fm*amax = m * 2^e, m in [0.5,1)
        sq = ldexpf(1.f, 14 - e);       // fm*amax*sq in [8192, 16384)
    }
    info->sq = sq;
    info->cs = sq * f.sx;
    info->bscale = -fm * sq;
    const float top = sq * f.sx * f.maxnorm2;
    info->force_fallback = (nonfinite || !(top < 1.0e30f) || !(sq * f.sx > 1.0e-30f)) ? 1 : 0;
    info->i8_mode = 0;
    if (f.corpus_i8 && !nonfinite) {
        const int window = !not_u8 ? 1 : (!not_s8 ? 2 : 0);
        info->i8_mode = window ? (window | (f.corpus_i8 & 4)) : 0;
    }
}

// Launch with query_stats_blocks(total) workgroups: few enough that the one set of atomics + fence per workgroup
// (same addresses for everybody) stays a few dozen operations -- on a 5 MB batch 1250 workgroups cost 21 us, 313 cost
// 15, 79 cost 10 and 40 cost 12 (then the reads of a thread queue up); the read itself takes 1-2.
inline unsigned query_stats_blocks(int64_t total) {
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>((total + 16383) / 16384, 256));
}
__global__ __launch_bounds__(256) void query_stats_kernel(const float *__restrict__ Q, int64_t total,
                                                          QueryBatchInfo *info, FinalizeArgs fin) {
    __shared__ float s_max[4];
    __shared__ int s_flags[4];
    float amax = 0.f;
    int flags = 0;  // bit0 non-finite, bit1 non-integer, bit2 outside 0..255, bit3 outside -128..127
    auto see = [&](float v) {
        amax = fmaxf(amax, fabsf(v));
        flags |= (!(fabsf(v) <= 3.402823466e+38f)) | ((v != rintf(v)) << 1) | ((!(v >= 0.f && v <= 255.f)) << 2) |
                 ((!(v >= -128.f && v <= 127.f)) << 3);
    };
    const int64_t nvec = ((reinterpret_cast<uintptr_t>(Q) & 15) == 0) ? total / 4 : 0;     // float4 body, scalar tail
    const float4 *Q4 = reinterpret_cast<const float4 *>(Q);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {          // four loads in flight per thread (the kernel is latency bound)
        const float4 v0 = Q4[i], v1 = Q4[i + stride], v2 = Q4[i + 2 * stride], v3 = Q4[i + 3 * stride];
        see(v0.x); see(v0.y); see(v0.z); see(v0.w);
        see(v1.x); see(v1.y); see(v1.z); see(v1.w);
        see(v2.x); see(v2.y); see(v2.z); see(v2.w);
        see(v3.x); see(v3.y); see(v3.z); see(v3.w);
    }
    for (; i < nvec; i += stride) {
        const float4 v = Q4[i];
        see(v.x); see(v.y); see(v.z); see(v.w);
    }
    for (int64_t j = nvec * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += stride) see(Q[j]);
    for (int o = 32; o > 0; o >>= 1) {
        amax = fmaxf(amax, __shfl_xor(amax, o));
        flags |= __shfl_xor(flags, o);
    }
    if ((threadIdx.x & 63) == 0) {
        s_max[threadIdx.x >> 6] = amax;
        s_flags[threadIdx.x >> 6] = flags;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // one set of atomics per workgroup
        amax = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        flags = s_flags[0] | s_flags[1] | s_flags[2] | s_flags[3];
        // (returning forms, results consumed below: a returned atomic has been performed at the coherence point)
        unsigned seen = 0;
        const unsigned ab = __float_as_uint(fabsf(amax));
        // (pre-checks: relaxed agent-scope atomic loads -- a stale 0 only costs the atomic it would have skipped)
        auto peek_u = [](unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        auto peek_i = [](int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        if (ab > peek_u(&info->absmax_bits)) seen += atomicMax(&info->absmax_bits, ab);
        if ((flags & 1) && peek_i(&info->nonfinite) == 0) seen += (unsigned)atomicOr(&info->nonfinite, 1);
        if ((flags & 2) && peek_i(&info->not_integer) == 0) seen += (unsigned)atomicOr(&info->not_integer, 1);
        if ((flags & 6) && peek_i(&info->not_u8) == 0) seen += (unsigned)atomicOr(&info->not_u8, 1);
        if ((flags & 10) && peek_i(&info->not_s8) == 0) seen += (unsigned)atomicOr(&info->not_s8, 1);
        asm volatile("" ::"v"(seen));
        // The workgroup that arrives last fixes the scales: no separate one-thread kernel (a ~5 us dispatch) between the
        // statistics and their consumers.  Every contribution is an agent-scope atomic (performed at the coherence point,
        // never parked in this XCD's L2) and query_finalize reads with agent-scope atomic loads, so waiting for this
        // thread's own atomics (vmcnt) before the counter bump orders them -- a __threadfence() here writes back the whole
        // L2 and cost 3-7 us per workgroup (MI355X_MICROARCH.md, "Valid forms": agent atomics on both sides).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (atomicAdd(&info->done_blocks, 1u) == gridDim.x - 1) {
            // acquire side of the arrival counter: the statistics loads of query_finalize may not be hoisted above the
            // counter bump (an agent-scope acquire fence only invalidates; it does not force the L2 write-back avoided above)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            query_finalize(info, fin);
        }
    }
}

// fp16 B-fragment panels of the query batch + padded float32 copy for the refine kernel.
// thread per (qtile32, kstep, lane)
__device__ __forceinline__ void build_qpanels_body(int64_t gid, const float *__restrict__ Q, int64_t nq, int D, int D4,
                                                   int ksteps, int64_t nqtiles,
                                                   const QueryBatchInfo *__restrict__ info,
                                                   half8 *__restrict__ qpanels, int x16 = 0) {
    const int lane = (int)(gid & 63);
    const int64_t tk = gid >> 6;
    if (tk >= nqtiles * ksteps || info->i8_mode) return;
    const float bs = info->bscale;
    int64_t q;
    int d0;
    if (x16) {     // [q / 16][ks2][lane]: query column lane & 15, dims 32 ks2 + 8 (lane >> 4) .. +7 (same size as the 32-query form)
        const int ks2n = ksteps / 2;
        q = (tk / ks2n) * 16 + (lane & 15);
        d0 = (int)(tk % ks2n) * 32 + (lane >> 4) * 8;
    } else {
        q = (tk / ksteps) * 32 + (lane & 31);
        d0 = (int)(tk % ksteps) * 16 + (lane >> 5) * 8;
    }
    half8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int d = d0 + j;
        float v = 0.f;
        if (q < nq && d < D) v = Q[(size_t)q * D + d] * bs;
        out[j] = (_Float16)v;
    }
    qpanels[gid] = out;
}

__global__ __launch_bounds__(256) void build_qpanels_kernel(const float *__restrict__ Q, int64_t nq, int D, int D4,
                                                            int ksteps, int64_t nqtiles,
                                                            const QueryBatchInfo *__restrict__ info,
                                                            half8 *__restrict__ qpanels) {
    build_qpanels_body((int64_t)blockIdx.x * blockDim.x + threadIdx.x, Q, nq, D, D4, ksteps, nqtiles, info, qpanels);
}

// B fragments for v_mfma_f32_16x16x32_f16: thread per (16-query block, 32-dim k-step, lane); lane l holds query
// column l&15, dims 32*ks + 8*(l>>4) .. +7.
__global__ __launch_bounds__(256) void build_qpanels16_kernel(const float *__restrict__ Q, int64_t nq, int D, int ks32,
                                                              int64_t nqblocks, const QueryBatchInfo *__restrict__ info,
                                                              half8 *__restrict__ qpanels) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(gid & 63);
    const int64_t tk = gid >> 6;
    const int ks = (int)(tk % ks32);
    const int64_t qb = tk / ks32;
    if (qb >= nqblocks) return;
    const float bs = info->bscale;
    const int64_t q = qb * 16 + (lane & 15);
    const int d0 = ks * 32 + (lane >> 4) * 8;
    half8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int d = d0 + j;
        float v = 0.f;
        if (q < nq && d < D) v = Q[(size_t)q * D + d] * bs;
        out[j] = (_Float16)v;
    }
    qpanels[gid] = out;
}

__global__ __launch_bounds__(256) void pad_rows_kernel(const float *__restrict__ src, int64_t n, int D, int D4,
                                                       float *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D4) return;
    const int64_t r = i / D4;
    const int d = (int)(i - r * D4);
    dst[i] = d < D ? src[(size_t)r * D + d] : 0.f;
}

// per-query error bound of the fp16 scan, in scan (scaled) units -- DESIGN.md "exactness guard".
struct EpsArgs {
    const float *Q;
    int64_t nq;
    int D;
    int Dpad;
    int metric;
    float xnorm_max;        // max ||x||
    int corpus_exact;       // fp16 copy of the corpus is exact
    int corpus_int_unscaled;// corpus integer valued and stored unscaled (sx == 1)
    float sx;
    const QueryBatchInfo *info;
    float *eps;             // [nq]
};

// 16 lanes per query (coalesced row reads, shuffle reduction); the bound only needs ||q|| to ~1e-15 relative, the
// 2 % slack below dwarfs the summation order.  Launch with ceil(nq / 16) * 16 lanes, 256 per block.
__device__ __forceinline__ void query_eps_body(int64_t gid, const EpsArgs &a) {
    const int64_t q = gid >> 4;
    const int part = (int)(gid & 15);
    const bool qv = q < a.nq;
    if (a.info->i8_mode) {     // int8 scan: exact integers; L2 drops the parity bit of the bias (1/2 in t units, so
        // 2*eps = 1).  The select adds the BITS of eps[q] to the packed key: (2 eps) << 6.
        if (qv && part == 0) a.eps[q] = __int_as_float(a.metric == 0 ? 64 : 0);
        return;
    }
    const float bs = a.info->bscale;
    const double cs = (double)a.info->cs;
    double n2 = 0.0;
    int inexact = 0, notint = 0;
    auto see = [&](float v) {
        n2 = fma((double)v, (double)v, n2);
        const float s = v * bs;
        inexact |= ((float)(_Float16)s != s);
        notint |= (v != rintf(v));
    };
    if (qv) {
        const float *row = a.Q + (size_t)q * a.D;
        if ((a.D & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Q) & 15) == 0) {
            // 16 bytes per lane and load, four loads in flight: one value per lane and trip was 24 DEPENDENT round trips per
            // query at 384 dims -- 20 us of every prep dispatch on a 10 000-query batch
            const float4 *r4 = reinterpret_cast<const float4 *>(row);
            const int n4 = a.D >> 2;
            int i = part;
            for (; i + 48 < n4; i += 64) {
                const float4 v0 = r4[i], v1 = r4[i + 16], v2 = r4[i + 32], v3 = r4[i + 48];
                see(v0.x); see(v0.y); see(v0.z); see(v0.w);
                see(v1.x); see(v1.y); see(v1.z); see(v1.w);
                see(v2.x); see(v2.y); see(v2.z); see(v2.w);
                see(v3.x); see(v3.y); see(v3.z); see(v3.w);
            }
            for (; i < n4; i += 16) {
                const float4 v = r4[i];
                see(v.x); see(v.y); see(v.z); see(v.w);
            }
        } else {
            for (int d = part; d < a.D; d += 16) see(row[d]);
        }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        n2 += __shfl_xor(n2, o);
        inexact |= __shfl_xor(inexact, o);
        notint |= __shfl_xor(notint, o);
    }
    if (!qv || part != 0) return;
    const double qn = sqrt(n2), Xn = (double)a.xnorm_max;
    const double u = 1.0 / 2048.0;  // fp16 unit roundoff 2^-11
    const double ux = a.corpus_exact ? 0.0 : u, uq = inexact ? u : 0.0;
    const double rel_in = ux + uq + ux * uq;
    const double mag = (a.metric == 0) ? (Xn * Xn + 2.0 * qn * Xn) : (qn * Xn);
    double dot_err = rel_in * qn * Xn;
    if (rel_in > 0.0) dot_err += ldexp(1.0, -36) * sqrt((double)a.D) * qn * Xn;  // fp16 subnormal tail
    const bool int_exact = a.corpus_int_unscaled && !notint && a.info->sq == 1.f && a.sx == 1.f &&
                           mag * 1.000001 < 16777216.0;
    const double acc_err = int_exact ? 0.0 : (double)(a.Dpad + 3) * ldexp(1.0, -23) * mag;
    const double pack_err = ldexp(1.0, -17) * mag;     // 6 mantissa bits carry the quad id
    const double bias_err = (a.metric == 0 && !int_exact) ? ldexp(1.0, -24) * Xn * Xn : 0.0;
    const double f = (a.metric == 0) ? 2.0 : 1.0;
    const double eps_true = f * dot_err + acc_err + pack_err + bias_err;
    double e = cs * eps_true * 1.02;
    if (!(e < 1.0e37)) e = 1.0e37;
    a.eps[q] = (float)e + 1.0e-30f;
}

__global__ __launch_bounds__(256) void query_eps_kernel(EpsArgs a) {
    query_eps_body((int64_t)blockIdx.x * blockDim.x + threadIdx.x, a);
}

}  // namespace vdb
