// dense.hpp -- small-corpus path (N <= 15360 rows: IVF coarse quantizers, k-means assignment, tiny indexes).
//
// With so few rows the bin-minimum select has nothing to work with (k is a sizeable fraction of N), so the
// fp16 MFMA scores of ALL rows are written out (nq x Npad float32) and a per-query kernel keeps every row whose
// score is within 2*eps of the k-th smallest one, re-scores those few rows in the canonical float64 arithmetic
// and returns the exact top-k.  Same guarantee as the bin path with bins of one row: every true neighbour has
// an approximate score <= tau + 2*eps.  Replaces a float64 pass over every (query, row) pair.
#pragma once
#include "common.hpp"
#include "prep.hpp"
#include "refine.hpp"
#include "scan.hpp"

namespace vdb {

// one wave = one 32-row tile x 64 queries; 4 waves of a workgroup take 4 consecutive tiles
template <int KSTEPS>
__global__ __launch_bounds__(256) void dense_scores_kernel(const half8 *__restrict__ panels, const float *__restrict__ bias,
                                                           const half8 *__restrict__ qpanels,
                                                           const QueryBatchInfo *__restrict__ info, int64_t ntiles,
                                                           int64_t Npad, float *__restrict__ out) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int64_t qt = blockIdx.y;                          // 64-query tile
    const float cs = info->cs;
    const int64_t span = tile / kTilesPerSpan;
    const int t = (int)(tile - span * kTilesPerSpan);
    const int64_t row0 = span * kSpanRows + (int64_t)h * kBinRows + t * 16;
    float16v acc0, acc1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 c = *reinterpret_cast<const float4 *>(bias + row0 + 4 * g);
        acc0[4 * g + 0] = (c.x >= 0.9e38f) ? kPadBias : c.x * cs;
        acc0[4 * g + 1] = (c.y >= 0.9e38f) ? kPadBias : c.y * cs;
        acc0[4 * g + 2] = (c.z >= 0.9e38f) ? kPadBias : c.z * cs;
        acc0[4 * g + 3] = (c.w >= 0.9e38f) ? kPadBias : c.w * cs;
    }
    acc1 = acc0;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        const half8 af = panels[((size_t)tile * KSTEPS + ks) * 64 + lane];
        const half8 bf0 = qpanels[((size_t)(qt * 2 + 0) * KSTEPS + ks) * 64 + lane];
        const half8 bf1 = qpanels[((size_t)(qt * 2 + 1) * KSTEPS + ks) * 64 + lane];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf1, acc1, 0, 0, 0);
    }
    float4 *o0 = reinterpret_cast<float4 *>(out + (size_t)(qt * 64 + (lane & 31)) * Npad + row0);
    float4 *o1 = reinterpret_cast<float4 *>(out + (size_t)(qt * 64 + 32 + (lane & 31)) * Npad + row0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        o0[g] = make_float4(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]);
        o1[g] = make_float4(acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]);
    }
}

// The same scores for D > 128 (IVF coarse quantizers over 384 / 768-dim embeddings: a few hundred centroids): the k-step loop
// is a run-time loop, four k-steps of operand loads in flight per wave (panels and query fragments come from L2 -- the
// centroid panels are a few hundred KiB).  Tiles that hold padding rows only store the padding score without touching the
// operands (100 centroids fill 7 of the 16 tiles of their 512-row span).  Replaces the float64 exhaustive kernel there:
// 192 us -> ~15 us per 10 000 queries x 100 centroids x 384 dims (profiles/r04_msmarco_ivf_breakdown.txt).
__global__ __launch_bounds__(256) void dense_scores_kloop_kernel(const half8 *__restrict__ panels, const float *__restrict__ bias,
                                                                 const half8 *__restrict__ qpanels,
                                                                 const QueryBatchInfo *__restrict__ info, int64_t ntiles,
                                                                 int64_t Npad, int64_t N, int ksteps, float *__restrict__ out) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int64_t qt = blockIdx.y;                          // 64-query tile
    const float cs = info->cs;
    const int64_t span = tile / kTilesPerSpan;
    const int t = (int)(tile - span * kTilesPerSpan);
    const int64_t row0 = span * kSpanRows + (int64_t)h * kBinRows + t * 16;
    float16v acc0, acc1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 c = *reinterpret_cast<const float4 *>(bias + row0 + 4 * g);
        acc0[4 * g + 0] = (c.x >= 0.9e38f) ? kPadBias : c.x * cs;
        acc0[4 * g + 1] = (c.y >= 0.9e38f) ? kPadBias : c.y * cs;
        acc0[4 * g + 2] = (c.z >= 0.9e38f) ? kPadBias : c.z * cs;
        acc0[4 * g + 3] = (c.w >= 0.9e38f) ? kPadBias : c.w * cs;
    }
    acc1 = acc0;
    const bool any_row = span * kSpanRows + t * 16 < N;     // (the first row of the tile's lower half: both halves past N otherwise)
    if (any_row) {
        const half8 *pa = panels + (size_t)tile * ksteps * 64 + lane;
        const half8 *pb0 = qpanels + (size_t)(qt * 2 + 0) * ksteps * 64 + lane;
        const half8 *pb1 = qpanels + (size_t)(qt * 2 + 1) * ksteps * 64 + lane;
        for (int ks = 0; ks < ksteps; ks += 4) {            // (ksteps is a multiple of 4 for D > 128)
            half8 af[4], bf0[4], bf1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                af[u] = pa[(size_t)(ks + u) * 64];
                bf0[u] = pb0[(size_t)(ks + u) * 64];
                bf1[u] = pb1[(size_t)(ks + u) * 64];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[u], bf0[u], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[u], bf1[u], acc1, 0, 0, 0);
            }
        }
    }
    float4 *o0 = reinterpret_cast<float4 *>(out + (size_t)(qt * 64 + (lane & 31)) * Npad + row0);
    float4 *o1 = reinterpret_cast<float4 *>(out + (size_t)(qt * 64 + 32 + (lane & 31)) * Npad + row0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        o0[g] = make_float4(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]);
        o1[g] = make_float4(acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]);
    }
}

struct DenseSelectArgs {
    RefineCommon c;
    const float *scores;       // [Qpad][Npad]
    const float *eps;          // [nq]
    const QueryBatchInfo *info;
    int64_t nq, Npad;
    int64_t ncols;             // register kernel: score columns it reads = the index's rows rounded up to 64 (0 = Npad; the
                               // columns beyond hold padding scores: 100 centroids fill 2 of the 8 column groups of their span)
    int cand_cap;              // candidates kept in LDS per query
    int32_t *fallback;         // [nq]
    int32_t *fb_list;
    int32_t *fb_count;
    unsigned long long *stat_counters;
    float *D;                  // final rows, or
    int64_t *I;
    double *pkeys;             // partial rows [nq][k]
    int64_t *pids;
    int inline_fallback;       // register kernel: a query whose candidate list overflows (or whose scales are unusable) is
                               // served at once by its own wave with the exhaustive exact scan of the <= 2048 rows, instead of
                               // being queued for a separate exhaustive pass (one dependent dispatch less per search)
    int set_only;              // final mode, register kernel: the caller only uses the SET of the k nearest rows (IVF coarse
                               // quantizer: which lists to probe), so candidates that are certainly among them skip the
                               // float64 re-scoring (dense_select_reg_kernel).  I then holds the set in no particular
                               // order and D is not meaningful.
};

// one wave (= one workgroup) per query; LDS: Npad sortable scores + cand_cap row ids (<= ~40 KiB)
template <int KPL>
__global__ __launch_bounds__(64) void dense_select_kernel(DenseSelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dense_smem[];
    const int lane = threadIdx.x & 63;
    const int64_t q = blockIdx.x;
    if (q >= a.nq) return;
    unsigned *vals = reinterpret_cast<unsigned *>(dense_smem);
    int *cands = reinterpret_cast<int *>(vals + a.Npad);
    const int n = (int)a.Npad, k = a.c.k;
    const float *src = a.scores + (size_t)q * a.Npad;
    for (int i = lane * 4; i < n; i += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(src + i);
        vals[i] = sortable_u32(v.x); vals[i + 1] = sortable_u32(v.y);
        vals[i + 2] = sortable_u32(v.z); vals[i + 3] = sortable_u32(v.w);
    }
    unsigned ans = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned trial = ans | ((1u << bit) - 1u);
        int cnt = 0;
        for (int i = lane; i < n; i += 64) cnt += (vals[i] <= trial) ? 1 : 0;
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if (cnt < k) ans |= (1u << bit);
    }
    const float that = unsortable_f32(ans) + 2.0f * a.eps[q];
    bool fb = a.info->force_fallback || !(that < 0.9e38f);
    const unsigned tkey = sortable_u32(that);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    int ncand = 0;
    for (int base = 0; base < n && !fb; base += 64) {
        const int i = base + lane;
        const bool hit = i < n && vals[i] <= tkey;
        const unsigned long long m = __ballot(hit);
        if (hit) {
            const int pos = ncand + __popcll(m & lt_mask);
            if (pos < a.cand_cap) cands[pos] = i;
        }
        ncand += __popcll(m);
    }
    if (ncand > a.cand_cap) fb = true;
    if (fb) {
        if (lane == 0) {
            a.fallback[q] = 1;
            a.fb_list[atomicAdd(a.fb_count, 1)] = (int)q;
            stat_add(a.stat_counters, q, 2, 1ull);
        }
        return;
    }
    const float *qptr = a.c.Q + (size_t)q * a.c.D4;
    WaveTopK<KPL> tk;
    tk.init(k);
    for (int base = 0; base < ncand; base += 64) {
        const int i = base + lane;
        bool valid = i < ncand;
        const int64_t row = valid ? (int64_t)cands[i] : 0;
        valid = valid && row < a.c.N;
        uint64_t key = ~0ull;
        if (valid) key = exact_key(a.c.X + (size_t)row * a.c.D4, qptr, a.c.D4, a.c.metric);
        tk.offer(key, a.c.id_base + row, valid);
    }
    const size_t o = (size_t)q * k;
    write_topk<KPL>(tk, a.c.metric, a.D ? a.D + o : nullptr, a.I ? a.I + o : nullptr, a.pkeys ? a.pkeys + o : nullptr,
                    a.pids ? a.pids + o : nullptr);
    if (lane == 0) {
        a.fallback[q] = 0;
        stat_add(a.stat_counters, q, 0, (unsigned long long)ncand);
    }
}

// Everything behind the candidate list of one query, wave-wide (64 lanes on ONE query): overflow / unusable scales -> exhaustive
// scan; set-only mode -> certain rows written as they are, the band around the k-th key re-scored; else exact re-scoring of all
// candidates.  (Round 4 tried a form of dense_select_reg_kernel with 2 / 4 queries per wave -- 32 / 16 lanes per query, counts
// reduced inside the lane group, this tail only for queries its shortcut did not settle: 13 - 23 us SLOWER per coarse search,
// profiles/r04_sweeps.txt.  The bisection needs LPQ x M >= 2k lane minima, so fewer lanes per query mean proportionally more
// compares and ballots per step: the instruction count per query does not fall.  Removed; the tail stayed a function.)
template <int KPL>
__device__ __forceinline__ void dense_select_tail(const DenseSelectArgs &a, int64_t q, int *cands, unsigned *ckeys, int ncand, bool fb,
                                                  unsigned ans, float e2) {
    const int lane = threadIdx.x & 63;
    const int k = a.c.k;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    if (ncand > a.cand_cap) fb = true;
    if (fb) {
        if (a.inline_fallback) {
            WaveTopK<KPL> tkf;
            tkf.init(k);
            scan_rows<KPL, 4>(tkf, a.c, a.c.Q + (size_t)q * a.c.D4, 0, a.c.N);
            const size_t of = (size_t)q * k;
            write_topk<KPL>(tkf, a.c.metric, a.D ? a.D + of : nullptr, a.I ? a.I + of : nullptr, a.pkeys ? a.pkeys + of : nullptr,
                            a.pids ? a.pids + of : nullptr);
            if (lane == 0) {
                a.fallback[q] = 0;
                stat_add(a.stat_counters, q, 2, 1ull);
            }
            return;
        }
        if (lane == 0) {
            a.fallback[q] = 1;
            a.fb_list[atomicAdd(a.fb_count, 1)] = (int)q;
            stat_add(a.stat_counters, q, 2, 1ull);
        }
        return;
    }
    // Set-only mode (which lists to probe): the candidates hold every row with an approximate score <= tau, so the
    // exact tau is the k-th smallest candidate key (a bisection over ~k values).  tau - eps bounds the exact k-th
    // distance from below, hence a row scoring below tau - 2 eps is among the k nearest for certain and is written out
    // as it is; only the band around tau is re-scored in float64, and its best k - #certain rows complete the set.
    int ncert = 0;
    if (a.set_only && a.D && ncand >= k) {
        unsigned t2 = 0xFFFFFFFFu;
        bool decided = ncand == k;     // exactly k rows within tau + 2 eps: every other row is farther than each of them
        if (!decided) {
            // exact tau = the k-th smallest candidate key.  `ans` (the k-th smallest of the lanes' M smallest) IS that value
            // whenever exactly k candidate keys are <= ans -- one counting pass instead of a 32-step bisection whose every
            // step re-reads the keys from LDS (the usual case; the bisection remains for ties and crowded lanes)
            int c_ans = 0;
            for (int base = 0; base < ncand; base += 64)
                c_ans += __popcll(__ballot(base + lane < ncand && ckeys[base + lane] <= ans));
            if (c_ans == k) {
                t2 = ans;
            } else {
                t2 = 0;
                for (int bit = 31; bit >= 0; --bit) {
                    const unsigned trial = t2 | ((1u << bit) - 1u);
                    int cnt = 0;
                    for (int base = 0; base < ncand; base += 64)
                        cnt += __popcll(__ballot(base + lane < ncand && ckeys[base + lane] <= trial));
                    if (cnt < k) t2 |= (1u << bit);
                }
            }
            // A gap wider than 2 eps between the k-th and the (k+1)-th approximate score settles the SET without any exact
            // re-scoring (exact_i <= tau + eps < next - eps <= exact_j; rows outside the candidate list are farther still) --
            // the usual case for a coarse quantizer, and the float64 keys were a third of this kernel's vector work.
            int cnt_le = 0;
            unsigned next = 0xFFFFFFFFu;
            for (int base = 0; base < ncand; base += 64) {
                const bool valid = base + lane < ncand;
                const unsigned ck = valid ? ckeys[base + lane] : 0xFFFFFFFFu;
                cnt_le += __popcll(__ballot(valid && ck <= t2));
                if (ck > t2) next = min(next, ck);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) next = min(next, (unsigned)__shfl_xor((int)next, o));
            const float tau0 = unsortable_f32(t2);
            decided = cnt_le == k && next != 0xFFFFFFFFu && unsortable_f32(next) - tau0 > e2 + 2.0e-6f * fabsf(tau0);
        }
        if (decided) {
            int nout = 0;
            for (int base = 0; base < ncand; base += 64) {
                const int i = base + lane;
                const bool in = i < ncand && ckeys[i] <= t2;
                const unsigned long long m = __ballot(in);
                if (in) a.I[(size_t)q * k + nout + __popcll(m & lt_mask)] = a.c.id_base + cands[i];
                nout += __popcll(m);
            }
            if (lane == 0) {
                a.fallback[q] = 0;
                stat_add(a.stat_counters, q, 0, (unsigned long long)ncand);
            }
            return;
        }
        const float tau = unsortable_f32(t2);
        const unsigned ckey = sortable_u32(tau - e2 - 1.0e-6f * fabsf(tau));   // (lowered a little more: rounding of the subtraction)
        int nband = 0;
        for (int base = 0; base < ncand; base += 64) {
            const int i = base + lane;
            const bool valid = i < ncand;
            const int row = valid ? cands[i] : 0;
            const bool cert = valid && ckeys[i] < ckey && row < a.c.N;
            const bool band = valid && !cert;
            const unsigned long long mc = __ballot(cert), mb = __ballot(band);
            if (cert) a.I[(size_t)q * k + ncert + __popcll(mc & lt_mask)] = a.c.id_base + row;     // (< k of them: tau is in the band)
            if (band) cands[nband + __popcll(mb & lt_mask)] = row;      // compacted in place: nband <= base
            ncert += __popcll(mc);
            nband += __popcll(mb);
        }
        ncand = nband;
    }
    const float *qptr = a.c.Q + (size_t)q * a.c.D4;
    WaveTopK<KPL> tk;
    tk.init(k - ncert);
    for (int base = 0; base < ncand; base += 64) {
        const int i = base + lane;
        bool valid = i < ncand;
        const int64_t row = valid ? (int64_t)cands[i] : 0;
        valid = valid && row < a.c.N;
        uint64_t key = ~0ull;
        if (valid) key = exact_key<4>(a.c.X + (size_t)row * a.c.D4, qptr, a.c.D4, a.c.metric);   // (4: fewer VGPRs, 8 waves per SIMD)
        tk.offer(key, a.c.id_base + row, valid);
    }
    const size_t o = (size_t)q * k + ncert;
    write_topk<KPL>(tk, a.c.metric, a.D ? a.D + o : nullptr, a.I ? a.I + o : nullptr, a.pkeys ? a.pkeys + o : nullptr,
                    a.pids ? a.pids + o : nullptr);
    if (lane == 0) {
        a.fallback[q] = 0;
        stat_add(a.stat_counters, q, 0, (unsigned long long)ncand);
    }
}

// Same selection for Npad <= 64 * VPL rows (IVF coarse quantizers with nlist <= 2048), with the query's scores held
// in REGISTERS (row e*64 + lane in v[e]) and a cheaper threshold.  The LDS form above bisects for the exact k-th smallest
// score tau: 32 passes over all Npad scores, LDS-bandwidth bound (100 us per 10k queries on 1024 centroids); doing the
// same passes on registers with v_cmp + ballot is bound by the CU's one scalar unit instead (58 us).  Any U >= tau is
// a valid threshold (every true neighbour has an approximate score <= tau + 2 eps <= U + 2 eps), so here every lane
// first keeps the M smallest of its VPL scores and U = the k-th smallest of those 64*M values (they are distinct rows,
// so U >= tau; equal to tau unless more than M of the k best fall on one lane).  The bisection then costs M compares
// per step instead of VPL, and the few extra rows U lets through are re-scored in lanes that would idle anyway.
// 4 queries per workgroup; LDS holds only the candidate lists.
template <int KPL, int VPL>
__global__ __launch_bounds__(256, ((KPL <= 2 && VPL <= 16) ? 6 : 4)) void dense_select_reg_kernel(DenseSelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dense_smem[];
    constexpr int M = KPL <= 1 ? 2 : (2 * KPL <= VPL ? 2 * KPL : VPL);     // 64*M >= 2k: k <= 64*KPL
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave));
    if (q >= a.nq) return;
    int *cands = reinterpret_cast<int *>(dense_smem) + (size_t)wave * 2 * a.cand_cap;      // rows | their approximate keys
    unsigned *ckeys = reinterpret_cast<unsigned *>(cands + a.cand_cap);
    const int n = (int)(a.ncols > 0 ? a.ncols : a.Npad), k = a.c.k;
    const float *src = a.scores + (size_t)q * a.Npad;
    unsigned v[VPL];
    float sv[VPL];                                  // (all loads in flight before the first use: see select_kernel, scan.hpp)
#pragma unroll
    for (int e = 0; e < VPL; ++e) sv[e] = src[min(e * 64 + lane, n - 1)];
#pragma unroll
    for (int e = 0; e < VPL; ++e) asm volatile("" : "+v"(sv[e]));
#pragma unroll
    for (int e = 0; e < VPL; ++e) v[e] = (e * 64 + lane < n) ? sortable_u32(sv[e]) : 0xFFFFFFFFu;
    unsigned low[M];                                // the M smallest of this lane, ascending
#pragma unroll
    for (int j = 0; j < M; ++j) low[j] = 0xFFFFFFFFu;
#pragma unroll
    for (int e = 0; e < VPL; ++e) {
        unsigned x = v[e];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const unsigned t = min(low[j], x);
            x = max(low[j], x);
            low[j] = t;
        }
    }
    unsigned ans = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned trial = ans | ((1u << bit) - 1u);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < M; ++j) cnt += __popcll(__ballot(low[j] <= trial));
        if (cnt < k) ans |= (1u << bit);
    }
    const float e2 = 2.0f * a.eps[q];
    const float that = unsortable_f32(ans) + e2;
    bool fb = a.info->force_fallback || !(that < 0.9e38f);
    const unsigned tkey = sortable_u32(that);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    int ncand = 0;
    if (!fb) {
#pragma unroll
        for (int e = 0; e < VPL; ++e) {
            const int i = e * 64 + lane;
            const bool hit = i < n && v[e] <= tkey;
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const int pos = ncand + __popcll(m & lt_mask);
                if (pos < a.cand_cap) {
                    cands[pos] = i;
                    ckeys[pos] = v[e];
                }
            }
            ncand += __popcll(m);
        }
    }
    dense_select_tail<KPL>(a, q, cands, ckeys, ncand, fb, ans, e2);
}


}  // namespace vdb
