// scan.hpp -- the dominant kernel: fp16 MFMA distance scan fused with a two-level bin-minimum select.
//
// Work decomposition (MI355X: 256 CUs, 8 XCDs with private L2, 160 KiB LDS, 512 VGPRs/lane):
//   workgroup = 8 waves = one (corpus chunk, 512-query tile); wave w owns 64 queries as TWO 32-query
//   MFMA column blocks whose B fragments (-2*q or -q, fp16) stay in registers for the whole chunk.
//   Corpus panels (A fragments, fp16, fragment-major in HBM) stream through a double-buffered LDS stage
//   of 4 tiles (128 MFMA rows); each A fragment read from LDS feeds two MFMAs.
//   The accumulator is initialised with cs*||x||^2 (L2) or 0 (IP), so the MFMA result IS the score
//   cs*(||x||^2 - 2 q.x)  or  -cs*(q.x): no separate epilogue arithmetic.
// Select: in the 32x32 C/D layout a lane owns ONE query column, its 16 accumulator registers are 16
//   corpus rows.  Per score: pack the row's 8-bit offset into the low mantissa bits, then
//   m2 = med3(m1, m2, v); m1 = min(m1, v)  -- 3 VALU ops per score, branch free.  A bin = the 256 consecutive
//   corpus rows a lane half sees per span (16 tiles); (m1, m2) of each bin go to HBM (level 1), and the
//   chunk-wide (min, second-min, span-of-min) triple (level 2) feeds the select kernel.
// Blocks of one chunk get ids that differ by multiples of 8, i.e. the same XCD under round-robin
//   placement: the chunk is fetched from HBM once into that XCD's L2 and re-read by the query tiles.
#pragma once
#include "common.hpp"
#include "prep.hpp"

namespace vdb {

typedef float float16v __attribute__((ext_vector_type(16)));

struct ScanArgs {
    const half8 *panels;   // [ntiles][KSTEPS][64]
    const float *bias;     // [Npad]
    const half8 *qpanels;  // [Qpad/32][KSTEPS][64]
    const QueryBatchInfo *info;
    float *bin_m1;         // [nspans*2][Qpad]  packed min of each bin
    float *bin_m2;         // [nspans*2][Qpad]  second min of each bin
    float *sb_m1;          // [nchunks*2][Qpad] packed min of each superbin (chunk x lane half)
    float *sb_m2;          // [nchunks*2][Qpad] second-smallest score of the superbin
    int32_t *sb_span;      // [nchunks*2][Qpad] span holding the superbin minimum
    int64_t nspans;        // total spans (Npad / 512)
    int spans_per_chunk;
    int nchunks;
    int nqtiles;           // query tiles of 512
    int64_t Qpad;          // multiple of 512
};

// (score & ~0xFF) | id  -- one v_and_or_b32 when the mask lives in a VGPR (the id is wave-uniform)
__device__ __forceinline__ float pack_score(float v, unsigned mask, unsigned id) {
    return __uint_as_float((__float_as_uint(v) & mask) | id);
}
// min without the sNaN-quieting v_max that fminf() drags in: med3(a, b, -inf) == min(a, b)
__device__ __forceinline__ float fast_min(float a, float b, float neg_inf) {
    return __builtin_amdgcn_fmed3f(a, b, neg_inf);
}

template <int KSTEPS>
__global__ __launch_bounds__(512, 2) void scan_kernel(ScanArgs a) {
    constexpr int kStageVec = kStageTiles * KSTEPS * 64;  // 16-byte vectors per stage
    constexpr int kLoads = kStageVec / 512;               // per thread (KSTEPS is even)
    static_assert(kStageVec % 512 == 0, "stage must divide over 512 threads");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (kStageVec * 16 + kStageTiles * 32 * 4)];
    auto lds_a = [&](int buf) { return reinterpret_cast<half8 *>(smem + buf * (kStageVec * 16)); };
    auto lds_b = [&](int buf) {
        return reinterpret_cast<float *>(smem + 2 * kStageVec * 16 + buf * (kStageTiles * 32 * 4));
    };

    // ---- block -> (chunk, query tile), XCD aware -------------------------------------------------
    const int b = blockIdx.x;
    const int x = b & 7, j = b >> 3;
    const int ci = j / a.nqtiles, qt = j - ci * a.nqtiles;
    const int chunk = x + 8 * ci;
    if (chunk >= a.nchunks) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5;
    const int64_t q0 = (int64_t)qt * 512 + wave * 64;  // first query of this wave
    const float cs = a.info->cs;

    // ---- B fragments: resident for the whole chunk ------------------------------------------------
    half8 bfrag[2][KSTEPS];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
            bfrag[cb][ks] = a.qpanels[((size_t)(q0 / 32 + cb) * KSTEPS + ks) * 64 + lane];

    const int64_t span0 = (int64_t)chunk * a.spans_per_chunk;
    int64_t span1 = span0 + a.spans_per_chunk;
    if (span1 > a.nspans) span1 = a.nspans;
    const int nstages = (int)(span1 - span0) * 4;

    const float INF = __builtin_inff();
    float NEG_INF = -INF;
    asm volatile("" : "+v"(NEG_INF));  // opaque, or LLVM folds med3(a,b,-inf) back into a canonicalising fmin
    unsigned idmask = 0xFFFFFF00u;
    asm volatile("" : "+v"(idmask));  // pin the mask in a VGPR (a literal cannot ride in VOP3 next to an SGPR id)
    float m1[2] = {INF, INF}, m2[2] = {INF, INF};   // level 1 (current bin)
    float M1[2] = {INF, INF}, M2[2] = {INF, INF};   // level 2 (whole chunk)
    int Ms[2] = {0, 0};

    // ---- staging: global -> registers -> LDS, double buffered --------------------------------------
    half8 stage_a[kLoads];
    float stage_b = 0.f;
    auto stage_load = [&](int st) {
        const int64_t span = span0 + (st >> 2);
        const int sq4 = st & 3;
        const half8 *src = a.panels + ((size_t)(span * kTilesPerSpan + sq4 * kStageTiles) * KSTEPS) * 64;
#pragma unroll
        for (int i = 0; i < kLoads; ++i) stage_a[i] = src[tid + i * 512];
        if (tid < kStageTiles * 32) {
            const int t = tid >> 5, hh = (tid >> 4) & 1, r = tid & 15;
            const float bv = a.bias[span * kSpanRows + hh * kBinRows + (sq4 * kStageTiles + t) * 16 + r];
            stage_b = (bv >= 0.9e38f) ? kPadBias : bv * cs;
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < kLoads; ++i) lds_a(buf)[tid + i * 512] = stage_a[i];
        if (tid < kStageTiles * 32) lds_b(buf)[tid] = stage_b;
    };

    stage_load(0);
    stage_store(0);
    __syncthreads();

    for (int st = 0; st < nstages; ++st) {
        const int buf = st & 1;
        if (st + 1 < nstages) stage_load(st + 1);

        const half8 *A = lds_a(buf);
        const float4 *B4 = reinterpret_cast<const float4 *>(lds_b(buf));
#pragma unroll
        for (int t = 0; t < kStageTiles; ++t) {
            float16v acc0, acc1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 c = B4[(t * 2 + h) * 4 + g];
                acc0[4 * g + 0] = c.x; acc0[4 * g + 1] = c.y; acc0[4 * g + 2] = c.z; acc0[4 * g + 3] = c.w;
            }
            acc1 = acc0;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const half8 af = A[(t * KSTEPS + ks) * 64 + lane];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bfrag[0][ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bfrag[1][ks], acc1, 0, 0, 0);
            }
            const unsigned idbase = (unsigned)(((st & 3) * kStageTiles + t) * 16);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v0 = pack_score(acc0[r], idmask, idbase + r);
                m2[0] = __builtin_amdgcn_fmed3f(m1[0], m2[0], v0);
                m1[0] = fast_min(m1[0], v0, NEG_INF);
                const float v1 = pack_score(acc1[r], idmask, idbase + r);
                m2[1] = __builtin_amdgcn_fmed3f(m1[1], m2[1], v1);
                m1[1] = fast_min(m1[1], v1, NEG_INF);
            }
        }

        if ((st & 3) == 3) {  // a span (two bins per lane column) is complete: flush level 1, fold level 2
            const int64_t span = span0 + (st >> 2);
            const size_t o = (size_t)(span * 2 + h) * a.Qpad + q0 + (lane & 31);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                a.bin_m1[o + cb * 32] = m1[cb];
                a.bin_m2[o + cb * 32] = m2[cb];
                M2[cb] = __builtin_fminf(__builtin_amdgcn_fmed3f(M1[cb], M2[cb], m1[cb]), m2[cb]);
                if (m1[cb] < M1[cb]) Ms[cb] = (int)span;
                M1[cb] = __builtin_fminf(M1[cb], m1[cb]);
                m1[cb] = INF;
                m2[cb] = INF;
            }
        }

        if (st + 1 < nstages) stage_store(buf ^ 1);
        __syncthreads();
    }

    const size_t so = (size_t)(chunk * 2 + h) * a.Qpad + q0 + (lane & 31);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        a.sb_m1[so + cb * 32] = M1[cb];
        a.sb_m2[so + cb * 32] = M2[cb];
        a.sb_span[so + cb * 32] = Ms[cb];
    }
}

// ---- select: per query, turn the bin minima into an exact-refine work list --------------------------
// One wave per query.  tau = k-th smallest superbin minimum (bitwise bisection on the sortable key);
// That = tau + 2*eps.  Every score <= That is either a bin minimum (-> candidate row) or lives in a bin
// whose second minimum is <= That (-> the bin is re-scanned exactly).  See DESIGN.md for the proof that
// the true top-k is contained in the union.
struct SelectArgs {
    const float *bin_m1, *bin_m2, *sb_m1, *sb_m2;
    const int32_t *sb_span;
    const float *eps;            // [nq]
    const QueryBatchInfo *info;
    int64_t nq, Qpad, nspans, N;
    int spans_per_chunk, nchunks, k;
    int cand_cap, rescan_cap;
    int32_t *cand_rows;          // [nq][cand_cap]
    int32_t *rescan_rows;        // [nq][rescan_cap]
    int32_t *counts;             // [nq][2]
    int32_t *fallback;           // [nq]
    int32_t *fb_list;            // [nq]
    int32_t *fb_count;           // [1]
    unsigned long long *stat_counters;  // [2] total candidates, total rescans
};

template <int VPL>
__global__ __launch_bounds__(256) void select_kernel(SelectArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t q = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (q >= a.nq) return;
    const int nsb = a.nchunks * 2;
    unsigned v[VPL];
#pragma unroll
    for (int e = 0; e < VPL; ++e) {
        const int s = e * 64 + lane;
        v[e] = (s < nsb) ? sortable_u32(a.sb_m1[(size_t)s * a.Qpad + q]) : 0xFFFFFFFFu;
    }
    // k-th smallest by bisection from the top bit
    unsigned ans = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned trial = ans | ((1u << bit) - 1u);
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < VPL; ++e) cnt += __popcll(__ballot(v[e] <= trial));
        if (cnt < a.k) ans |= (1u << bit);
    }
    const float tau = unsortable_f32(ans);
    const float that = tau + 2.0f * a.eps[q];
    const bool force_fb = a.info->force_fallback || !(that < 0.9e38f) || nsb < a.k;

    int ncand = 0, nres = 0;  // wave-uniform
    int32_t *cr = a.cand_rows + (size_t)q * a.cand_cap;
    int32_t *rr = a.rescan_rows + (size_t)q * a.rescan_cap;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    if (!force_fb) {
#pragma unroll
        for (int e = 0; e < VPL; ++e) {
            const int s = e * 64 + lane;
            const bool in = s < nsb;
            const float m1 = in ? unsortable_f32(v[e]) : __builtin_inff();
            const bool active = in && (m1 <= that);
            float sm2 = __builtin_inff();
            int sspan = 0;
            if (active) {
                sm2 = a.sb_m2[(size_t)s * a.Qpad + q];
                sspan = a.sb_span[(size_t)s * a.Qpad + q];
            }
            const bool single = active && !(sm2 <= that);
            // superbins whose only interesting score is their minimum: one candidate row each
            const unsigned long long smask = __ballot(single);
            if (single) {
                const int pos = ncand + __popcll(smask & lt_mask);
                const int hh = s & 1;
                const int row = sspan * kSpanRows + hh * kBinRows + (int)(__float_as_uint(m1) & 0xFFu);
                if (pos < a.cand_cap) cr[pos] = row;
            }
            ncand += __popcll(smask);
            // superbins with two or more interesting scores: walk their level-1 bins
            unsigned long long dmask = __ballot(active && !single);
            while (dmask) {
                const int src = __ffsll(dmask) - 1;
                dmask &= dmask - 1;
                const int sb = e * 64 + src;
                const int chunk = sb >> 1, hh = sb & 1;
                const int64_t sp0 = (int64_t)chunk * a.spans_per_chunk;
                int64_t sp1 = sp0 + a.spans_per_chunk;
                if (sp1 > a.nspans) sp1 = a.nspans;
                for (int64_t base = sp0; base < sp1; base += 64) {
                    const int64_t sp = base + lane;
                    bool act = false, resc = false;
                    float bm1 = 0.f;
                    if (sp < sp1) {
                        const size_t o = (size_t)(sp * 2 + hh) * a.Qpad + q;
                        bm1 = a.bin_m1[o];
                        if (bm1 <= that) {
                            act = true;
                            resc = a.bin_m2[o] <= that;
                        }
                    }
                    const bool cand = act && !resc;
                    const unsigned long long cm = __ballot(cand), rm = __ballot(resc);
                    if (cand) {
                        const int pos = ncand + __popcll(cm & lt_mask);
                        if (pos < a.cand_cap)
                            cr[pos] = (int)(sp * kSpanRows + hh * kBinRows + (int)(__float_as_uint(bm1) & 0xFFu));
                    }
                    if (resc) {
                        const int pos = nres + __popcll(rm & lt_mask);
                        if (pos < a.rescan_cap) rr[pos] = (int)(sp * kSpanRows + hh * kBinRows);
                    }
                    ncand += __popcll(cm);
                    nres += __popcll(rm);
                }
            }
        }
    }
    const bool fb = force_fb || ncand > a.cand_cap || nres > a.rescan_cap;
    if (lane == 0) {
        a.counts[2 * q] = fb ? 0 : ncand;
        a.counts[2 * q + 1] = fb ? 0 : nres;
        a.fallback[q] = fb ? 1 : 0;
        if (fb) {
            const int pos = atomicAdd(a.fb_count, 1);
            a.fb_list[pos] = (int)q;
        } else {
            atomicAdd(&a.stat_counters[0], (unsigned long long)ncand);
            atomicAdd(&a.stat_counters[1], (unsigned long long)nres);
        }
    }
}

}  // namespace vdb
