// refine.hpp -- exact (float64) re-scoring kernels and the partial-list merge.
//
// These kernels define the library's canonical arithmetic (see include/vdbhip.h):
//   L2: acc = fma(t, t, acc), t = (double)x[d] - (double)q[d];   IP: acc = fma((double)q[d], (double)x[d], acc)
// for d ascending; order key = (sortable(acc or -acc), id).  One lane scores one row (the chain is
// sequential by definition), one wave owns one (query, split) and keeps its top-k in registers.
#pragma once
#include "common.hpp"
#include "topk.hpp"

namespace vdb {

struct RefineCommon {
    const float *X;     // [N][D4] float32 rows, zero padded to a multiple of 4 dims
    const float *Q;     // [nq][D4]
    int64_t N;
    int64_t id_base;
    int D4;
    int metric;
    int k;
    const int64_t *idmap;  // optional: id of row r is idmap[r] (IVF: rows are grouped by list), else id_base + r
    // byte-valued corpora (scan_i8.hpp): row-major int8 copy x - cx, x8_pitch = 64 or 128 bytes per row (padding bytes
    // hold -cx, i.e. x = 0), rowstat[row] = {sum x^2, sum x}, and Q8 = the batch's int8 query rows cq - q (padding 0).
    // When the batch is integer too (info->i8_mode) the list refine scores from them in exact integer arithmetic with
    // v_dot4c_i32_i8 (query row in SGPRs): the results are the same integers the float64 chain produces, so the keys are
    // bit-identical, from 136 gathered bytes per row instead of 512 (the refine is bound by that gather).
    const signed char *X8 = nullptr;
    const int *rowstat = nullptr;
    const signed char *Q8 = nullptr;
    int x8_pitch = 0, cx = 0, D = 0;
    const QueryBatchInfo *info = nullptr;
};

typedef int refine_int4 __attribute__((ext_vector_type(4)));

// U: float4 pairs in flight per lane (16 for the gather-bound refine kernels; kernels that want occupancy pass 4)
template <int U = 16>
__device__ __forceinline__ uint64_t exact_key(const float *__restrict__ x, const float *__restrict__ q, int D4,
                                              int metric) {
    const float4 *xv = reinterpret_cast<const float4 *>(x);
    const float4 *qv = reinterpret_cast<const float4 *>(q);
    double acc = 0.0;
    if (metric == 0) {
#pragma unroll U
        for (int i = 0; i < D4 / 4; ++i) {
            const float4 a = xv[i];
            const float4 b = qv[i];
            double t;
            t = (double)a.x - (double)b.x; acc = fma(t, t, acc);
            t = (double)a.y - (double)b.y; acc = fma(t, t, acc);
            t = (double)a.z - (double)b.z; acc = fma(t, t, acc);
            t = (double)a.w - (double)b.w; acc = fma(t, t, acc);
        }
        return sortable_u64(acc);
    } else {
#pragma unroll U
        for (int i = 0; i < D4 / 4; ++i) {
            const float4 a = xv[i];
            const float4 b = qv[i];
            acc = fma((double)b.x, (double)a.x, acc);
            acc = fma((double)b.y, (double)a.y, acc);
            acc = fma((double)b.z, (double)a.z, acc);
            acc = fma((double)b.w, (double)a.w, acc);
        }
        return sortable_u64(-acc);
    }
}

// The same key from the int8 row copy of a byte-valued corpus (int8-only indexes keep no float32 rows): x = byte + cx exactly,
// then the identical float64 chain -- bit-identical keys.  x8 has room for D4 bytes (padding bytes hold -cx, i.e. x = 0).
template <int U = 16>
__device__ __forceinline__ uint64_t exact_key_i8(const signed char *__restrict__ x8, int cx, const float *__restrict__ q,
                                                 int D4, int metric) {
    const int *xw = reinterpret_cast<const int *>(x8);
    const float4 *qv = reinterpret_cast<const float4 *>(q);
    double acc = 0.0;
#pragma unroll U
    for (int i = 0; i < D4 / 4; ++i) {
        const int w = xw[i];
        const float4 b = qv[i];
        const double x0 = (double)((int)(signed char)(w & 0xff) + cx), x1 = (double)((int)(signed char)((w >> 8) & 0xff) + cx);
        const double x2 = (double)((int)(signed char)((w >> 16) & 0xff) + cx), x3 = (double)((w >> 24) + cx);
        if (metric == 0) {
            double t;
            t = x0 - (double)b.x; acc = fma(t, t, acc);
            t = x1 - (double)b.y; acc = fma(t, t, acc);
            t = x2 - (double)b.z; acc = fma(t, t, acc);
            t = x3 - (double)b.w; acc = fma(t, t, acc);
        } else {
            acc = fma((double)b.x, x0, acc);
            acc = fma((double)b.y, x1, acc);
            acc = fma((double)b.z, x2, acc);
            acc = fma((double)b.w, x3, acc);
        }
    }
    return sortable_u64(metric == 0 ? acc : -acc);
}

// key of corpus row `row`: from the float32 rows, or (c.X == nullptr: int8-only index) from the int8 row copy
template <int U = 16>
__device__ __forceinline__ uint64_t row_key(const RefineCommon &c, int64_t row, const float *__restrict__ q) {
    if (c.X) return exact_key<U>(c.X + (size_t)row * c.D4, q, c.D4, c.metric);
    return exact_key_i8<U>(c.X8 + (size_t)row * c.x8_pitch, c.cx, q, c.D4, c.metric);
}

// QB keys of ONE row against QB queries: the row is fetched once; every query keeps its own sequential chain, so each
// key is bit-identical to exact_key().
template <int QB>
__device__ __forceinline__ void exact_keys(const float *__restrict__ x, const float *const (&q)[QB], int D4, int metric,
                                           uint64_t (&out)[QB]) {
    const float4 *xv = reinterpret_cast<const float4 *>(x);
    double acc[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) acc[j] = 0.0;
#pragma unroll 4
    for (int i = 0; i < D4 / 4; ++i) {
        const float4 a = xv[i];
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            const float4 b = reinterpret_cast<const float4 *>(q[j])[i];
            if (metric == 0) {
                double t;
                t = (double)a.x - (double)b.x; acc[j] = fma(t, t, acc[j]);
                t = (double)a.y - (double)b.y; acc[j] = fma(t, t, acc[j]);
                t = (double)a.z - (double)b.z; acc[j] = fma(t, t, acc[j]);
                t = (double)a.w - (double)b.w; acc[j] = fma(t, t, acc[j]);
            } else {
                acc[j] = fma((double)b.x, (double)a.x, acc[j]);
                acc[j] = fma((double)b.y, (double)a.y, acc[j]);
                acc[j] = fma((double)b.z, (double)a.z, acc[j]);
                acc[j] = fma((double)b.w, (double)a.w, acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < QB; ++j) out[j] = sortable_u64(metric == 0 ? acc[j] : -acc[j]);
}

template <int KPL, int U = 16>
__device__ __forceinline__ void scan_rows(WaveTopK<KPL> &tk, const RefineCommon &c, const float *qptr, int64_t row0,
                                          int64_t row1) {
    const int lane = threadIdx.x & 63;
    if (row1 > c.N) row1 = c.N;
    for (int64_t base = row0; base < row1; base += 64) {
        const int64_t row = base + lane;
        const bool valid = row < row1;
        uint64_t key = ~0ull;
        if (valid) key = row_key<U>(c, row, qptr);
        tk.offer(key, c.idmap ? (valid ? c.idmap[row] : -1) : c.id_base + row, valid);
    }
}

// final (float32 distance, int64 id) rows or partial (float64 key, id) rows
template <int KPL>
__device__ __forceinline__ void write_topk(const WaveTopK<KPL> &tk, int metric, float *D, int64_t *I, double *pk,
                                           int64_t *pi) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int e = 0; e < KPL; ++e) {
        const int s = e * 64 + lane;
        if (s < tk.k) {
            const bool have = s < tk.n;
            const double kv = have ? unsortable_f64(tk.key[e]) : __builtin_inf();
            if (D) {
                float d;
                if (have)
                    d = (float)(metric == 0 ? kv : -kv);
                else
                    d = (metric == 0) ? 3.402823466e+38f : -3.402823466e+38f;
                D[s] = d;
                I[s] = have ? tk.id[e] : -1;
            } else {
                pk[s] = kv;
                pi[s] = have ? tk.id[e] : -1;
            }
        }
    }
}

// ---- list mode: candidates nominated by the scan + bins to re-scan -------------------------------
struct RefineListArgs {
    RefineCommon c;
    int64_t nq;
    const int32_t *cand_rows;   // [nq][cand_cap] first row of each candidate quad (4 consecutive rows)
    const int32_t *rescan_rows; // [nq][rescan_cap][2] row ranges [row0,row1) of the bins to re-scan
    const int32_t *counts;      // [nq][2] {n_cand, n_rescan}
    const int32_t *fallback;    // [nq] 1 -> handled by the exhaustive pass
    int cand_cap;
    int rescan_cap;
    float *D;                   // final mode (nullptr -> partial mode)
    int64_t *I;
    double *pkeys;              // partial mode: [nq][k]
    int64_t *pids;
    int group_shift = -1;       // log2 of the rows per candidate group; -1 = by the scan's arithmetic (quads, octs on the int8 scan)
    int f16_shift = 2;          // ... of the fp16 scan when group_shift is -1 (3: scan_kernel<.., G8>)
};

// int8 form of the list refine for one query (wave): W = x8_pitch / 4 dwords per row.  The query's int8 row b = cq - q
// sits in SGPRs; per row  dot = sum (x - cx)(cq - q)  by W v_dot4c, and with n2 = sum x^2, s1 = sum x of the row
//   x.q     = -dot + cq s1 + cx sum(q) - D cx cq          ||x - q||^2 = n2 - 2 x.q + sum(q^2)
// -- all exact integers below 2^27, equal to what the canonical float64 chain computes on these integer inputs.
template <int KPL, int W>
__device__ __forceinline__ void refine_list_dot8(const RefineListArgs &a, int64_t q, int ncand, int nres, int gshift,
                                                 WaveTopK<KPL> &tk) {
    const int lane = threadIdx.x & 63;
    const RefineCommon &c = a.c;
    const int mode = c.info->i8_mode;
    const int cq = (mode & 3) == 1 ? 127 : -1;
    int qb[W];
    const int *q8 = reinterpret_cast<const int *>(c.Q8 + (size_t)q * c.x8_pitch);      // wave-uniform: scalar loads
#pragma unroll
    for (int w = 0; w < W; ++w) qb[w] = __builtin_amdgcn_readfirstlane(q8[w]);
    // per query: sum q, sum q^2 (q holds integers)
    int sq = 0, sqq = 0;
    const float *qf = c.Q + (size_t)q * c.D4;
    for (int d = lane; d < c.D4; d += 64) {
        const int v = (int)qf[d];
        sq += v;
        sqq += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sq += __shfl_xor(sq, o);
        sqq += __shfl_xor(sqq, o);
    }
    const long long cxq = (long long)c.cx * sq - (long long)c.D * c.cx * cq;          // cx sum(q) - D cx cq
    auto key_of = [&](int64_t row) -> uint64_t {
        const refine_int4 *xr = reinterpret_cast<const refine_int4 *>(c.X8 + (size_t)row * c.x8_pitch);
        const int2 st = reinterpret_cast<const int2 *>(c.rowstat)[row];
        int dot = 0;
#pragma unroll
        for (int i = 0; i < W / 4; ++i) {
            const refine_int4 w = xr[i];
            dot = __builtin_amdgcn_sdot4(w.x, qb[4 * i], dot, false);
            dot = __builtin_amdgcn_sdot4(w.y, qb[4 * i + 1], dot, false);
            dot = __builtin_amdgcn_sdot4(w.z, qb[4 * i + 2], dot, false);
            dot = __builtin_amdgcn_sdot4(w.w, qb[4 * i + 3], dot, false);
        }
        const long long xq = -(long long)dot + (long long)cq * st.y + cxq;
        return c.metric == 0 ? sortable_u64((double)((long long)st.x - 2 * xq + sqq)) : sortable_u64(-(double)xq);
    };
    const int32_t *cr = a.cand_rows + (size_t)q * a.cand_cap;
    for (int base = 0; base < (ncand << gshift); base += 64) {
        const int i = base + lane;
        bool valid = i < (ncand << gshift);
        const int64_t row = valid ? (int64_t)cr[i >> gshift] + (i & ((1 << gshift) - 1)) : 0;
        valid = valid && row < c.N;
        const uint64_t key = valid ? key_of(row) : ~0ull;
        tk.offer(key, c.idmap ? (valid ? c.idmap[row] : -1) : c.id_base + row, valid);
    }
    const int32_t *rr = a.rescan_rows + (size_t)q * a.rescan_cap * 2;
    for (int r = 0; r < nres; ++r) {
        const int64_t row0 = rr[2 * r];
        int64_t row1 = rr[2 * r + 1];
        if (row1 > c.N) row1 = c.N;
        for (int64_t base = row0; base < row1; base += 64) {
            const int64_t row = base + lane;
            const bool valid = row < row1;
            const uint64_t key = valid ? key_of(row) : ~0ull;
            tk.offer(key, c.idmap ? (valid ? c.idmap[row] : -1) : c.id_base + row, valid);
        }
    }
}

template <int KPL>
__device__ __forceinline__ void refine_list_body(const RefineListArgs &a, unsigned block) {
    const int lane = threadIdx.x & 63;
    const int64_t q = __builtin_amdgcn_readfirstlane((int)(block * 4 + (threadIdx.x >> 6)));
    if (q >= a.nq) return;
    if (a.fallback[q]) return;
    const float *qptr = a.c.Q + (size_t)q * a.c.D4;
    WaveTopK<KPL> tk;
    tk.init(a.c.k);
    int ncand = a.counts[2 * q];
    int nres = a.counts[2 * q + 1];
    ncand = ncand < a.cand_cap ? ncand : a.cand_cap;
    nres = nres < a.rescan_cap ? nres : a.rescan_cap;
    const int32_t *cr = a.cand_rows + (size_t)q * a.cand_cap;
    const int mode = a.c.info != nullptr ? a.c.info->i8_mode : 0;
    const int gshift = a.group_shift >= 0 ? a.group_shift : mode ? ((mode & 4) ? 3 : 2) : a.f16_shift;   // a candidate = 4 or 8 (or 1 / 2) consecutive rows
    if (mode != 0 && a.c.X8 != nullptr) {                        // integer batch on a byte-valued corpus: int8 rows
        if (a.c.x8_pitch == 64) refine_list_dot8<KPL, 16>(a, q, ncand, nres, gshift, tk);
        else refine_list_dot8<KPL, 32>(a, q, ncand, nres, gshift, tk);
    } else {
        for (int base = 0; base < (ncand << gshift); base += 64) {
            const int i = base + lane;
            bool valid = i < (ncand << gshift);
            int64_t row = valid ? (int64_t)cr[i >> gshift] + (i & ((1 << gshift) - 1)) : 0;
            valid = valid && row < a.c.N;
            uint64_t key = ~0ull;
            if (valid) key = row_key(a.c, row, qptr);
            tk.offer(key, a.c.idmap ? (valid ? a.c.idmap[row] : -1) : a.c.id_base + row, valid);
        }
        const int32_t *rr = a.rescan_rows + (size_t)q * a.rescan_cap * 2;
        for (int r = 0; r < nres; ++r) scan_rows<KPL>(tk, a.c, qptr, rr[2 * r], rr[2 * r + 1]);
    }
    const size_t o = (size_t)q * a.c.k;
    write_topk<KPL>(tk, a.c.metric, a.D ? a.D + o : nullptr, a.I ? a.I + o : nullptr, a.pkeys ? a.pkeys + o : nullptr,
                    a.pids ? a.pids + o : nullptr);
}

// ---- flagged queries of the MFMA scan path: exhaustive exact scan, split on the device, merged by the last arriver -----
// The select flags the (rare) query whose work lists overflowed; how many there are is known only on the device.  This
// used to be two dispatches behind every search -- an exhaustive pass over S fixed splits and a merge of its partial lists,
// both empty almost always.  Here ONE fixed grid reads the count (zero: every wave returns after one load), cuts every
// flagged query into S = min(waves / count, max_split) row ranges, and the S partial lists of a query are merged by
// whichever of its waves arrives last (agent-scope release / acquire around a per-query arrival counter, which the merging
// wave resets: the counters are zero between searches).  S = 1 (many flagged queries): a wave writes its result directly.
struct RefineFallbackArgs {
    RefineCommon c;
    const int32_t *fb_list;      // [count] flagged queries
    const int32_t *fb_count;     // [1]
    int max_split;
    int64_t cap_units;           // capacity of the partial buffers in (query, split) units, >= the number of queries
    double *pkeys;               // [cap_units][k]
    int64_t *pids;
    int32_t *done;               // [nq] arrival counters, zero on entry and on exit
    float *D;                    // final rows, or (D == nullptr) per-shard partial rows
    int64_t *I;
    double *okeys;
    int64_t *oids;
};

template <int KPL>
__device__ __forceinline__ void refine_fallback_body(const RefineFallbackArgs &a, unsigned block, unsigned nblocks) {
    const int64_t count = *a.fb_count;
    if (count <= 0) return;
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)nblocks * 4;
    const int64_t wave0 = __builtin_amdgcn_readfirstlane((int)(block * 4 + (threadIdx.x >> 6)));
    int64_t S = waves / count;
    if (S > a.max_split) S = a.max_split;
    if (S > a.cap_units / count) S = a.cap_units / count;
    if (S < 1) S = 1;
    const int64_t rows_per_split = ((a.c.N + S - 1) / S + 63) / 64 * 64;
    for (int64_t u = wave0; u < count * S; u += waves) {
        const int64_t f = u / S;
        const int split = (int)(u - f * S);
        const int64_t q = a.fb_list[f];
        WaveTopK<KPL> tk;
        tk.init(a.c.k);
        // (4 row loads in flight, not 16: this body shares its kernel with the work-list refine, whose occupancy -- 71 VGPRs
        //  alone, 136 with a 16-deep exhaustive scan beside it -- is what its gather-bound time hangs on)
        scan_rows<KPL, 4>(tk, a.c, a.c.Q + (size_t)q * a.c.D4, (int64_t)split * rows_per_split, (int64_t)(split + 1) * rows_per_split);
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
        if (lane == 0) a.done[f] = 0;
        tk.init(a.c.k);
        const int total = (int)S * a.c.k;
        const size_t o0 = (size_t)f * S * a.c.k;
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

// The tail of a scan-path search in ONE launch: both parts only consume what the select left (work lists | flagged
// queries) and write disjoint result rows.  The first `fb_blocks` workgroups are the fallback pass (they start first: a
// flagged query is the long pole; with none flagged they return after one load), the rest re-score the work lists.
template <int KPL>
__global__ __launch_bounds__(256, (KPL == 1 ? 6 : KPL == 2 ? 5 : 1)) void refine_tail_kernel(RefineListArgs la, RefineFallbackArgs fa, unsigned fb_blocks) {
    if (blockIdx.x < fb_blocks) refine_fallback_body<KPL>(fa, blockIdx.x, fb_blocks);
    else refine_list_body<KPL>(la, blockIdx.x - fb_blocks);
}

// ---- rerank mode: explicit candidate ids per query ---------------------------------------------------
struct RerankArgs {
    RefineCommon c;
    int64_t nq;
    const int64_t *cand;   // [nq][ncand] ids (id_base + row), -1 = empty
    int ncand;
    float *D;              // final rows, or (D == nullptr) partial rows: float64 keys + ids
    int64_t *I;
    double *pkeys = nullptr;
    int64_t *pids = nullptr;
    // a shard of a multi-device index (multi.inc): candidate ids are GLOBAL; this shard holds the id ranges
    // [segs[3j], segs[3j] + segs[3j + 2]) as its local rows segs[3j + 1] ..., every other id is not ours (nullptr: row = id - id_base)
    const int64_t *segs = nullptr;
    int nseg = 0;
};

template <int KPL>
__global__ __launch_bounds__(256) void rerank_kernel(RerankArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t q = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (q >= a.nq) return;
    const float *qptr = a.c.Q + (size_t)q * a.c.D4;
    WaveTopK<KPL> tk;
    tk.init(a.c.k);
    const int64_t *cr = a.cand + (size_t)q * a.ncand;
    for (int base = 0; base < a.ncand; base += 64) {
        const int i = base + lane;
        bool valid = i < a.ncand;
        const int64_t id = valid ? cr[i] : -1;
        int64_t row = id - a.c.id_base;
        if (a.segs) {
            row = -1;
            for (int j = 0; j < a.nseg; ++j) {
                const int64_t id0 = a.segs[3 * j];
                if (id >= id0 && id < id0 + a.segs[3 * j + 2]) row = a.segs[3 * j + 1] + (id - id0);
            }
        }
        valid = valid && id >= 0 && row >= 0 && row < a.c.N;
        uint64_t key = ~0ull;
        if (valid) key = row_key(a.c, row, qptr);
        tk.offer(key, id, valid);
    }
    const size_t o = (size_t)q * a.c.k;
    write_topk<KPL>(tk, a.c.metric, a.D ? a.D + o : nullptr, a.I ? a.I + o : nullptr, a.pkeys ? a.pkeys + o : nullptr,
                    a.pids ? a.pids + o : nullptr);
}

// ---- exhaustive mode: every row of the shard, split over S waves per query -----------------------
struct RefineFullArgs {
    RefineCommon c;
    const int32_t *qlist;     // optional: slot -> query
    const int32_t *count_ptr; // optional: number of slots lives on the device
    int64_t count;            // used when count_ptr == nullptr
    int S;                    // splits per query
    int64_t rows_per_split;
    // S == 1 and final != nullptr: write final rows directly; else partials [slot][S][k]
    float *D;
    int64_t *I;
    double *pkeys;
    int64_t *pids;
};

template <int KPL>
__global__ __launch_bounds__(256) void refine_full_kernel(RefineFullArgs a) {
    const int64_t waves_total = (int64_t)gridDim.x * 4;
    const int64_t wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int64_t count = a.count_ptr ? (int64_t)*a.count_ptr : a.count;
    const int64_t units = count * a.S;
    for (int64_t u = wave0; u < units; u += waves_total) {
        const int64_t slot = u / a.S;
        const int split = (int)(u - slot * a.S);
        const int64_t q = a.qlist ? (int64_t)a.qlist[slot] : slot;
        const float *qptr = a.c.Q + (size_t)q * a.c.D4;
        WaveTopK<KPL> tk;
        tk.init(a.c.k);
        const int64_t r0 = (int64_t)split * a.rows_per_split;
        scan_rows<KPL>(tk, a.c, qptr, r0, r0 + a.rows_per_split);
        if (a.D) {
            const size_t o = (size_t)q * a.c.k;
            write_topk<KPL>(tk, a.c.metric, a.D + o, a.I + o, nullptr, nullptr);
        } else {
            const size_t o = (size_t)u * a.c.k;
            write_topk<KPL>(tk, a.c.metric, nullptr, nullptr, a.pkeys + o, a.pids + o);
        }
    }
}

// Query-blocked form of the exhaustive scan: one wave owns QB consecutive slots and one row range, fetches every row
// ONCE and scores it against the QB queries (the single-query form re-reads the whole corpus per query and runs at the
// HBM roofline: 512 MB per query per 1M x 128 rows).  Same outputs and layouts as refine_full_kernel.
template <int KPL, int QB>
__global__ __launch_bounds__(256) void refine_full_blocked_kernel(RefineFullArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t waves_total = (int64_t)gridDim.x * 4;
    const int64_t wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int64_t count = a.count_ptr ? (int64_t)*a.count_ptr : a.count;
    const int64_t groups = (count + QB - 1) / QB;
    const int64_t units = groups * a.S;
    for (int64_t u = wave0; u < units; u += waves_total) {
        const int64_t grp = u / a.S;
        const int split = (int)(u - grp * a.S);
        int64_t slot[QB], q[QB];
        const float *qptr[QB];
        WaveTopK<KPL> tk[QB];
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            slot[j] = grp * QB + j;
            const int64_t sj = slot[j] < count ? slot[j] : count - 1;      // (padding slots re-score the last query)
            q[j] = a.qlist ? (int64_t)a.qlist[sj] : sj;
            qptr[j] = a.c.Q + (size_t)q[j] * a.c.D4;
            tk[j].init(a.c.k);
        }
        const int64_t r0 = (int64_t)split * a.rows_per_split;
        int64_t r1 = r0 + a.rows_per_split;
        if (r1 > a.c.N) r1 = a.c.N;
        for (int64_t base = r0; base < r1; base += 64) {
            const int64_t row = base + lane;
            const bool valid = row < r1;
            uint64_t key[QB];
#pragma unroll
            for (int j = 0; j < QB; ++j) key[j] = ~0ull;
            if (valid) {
                if (a.c.X) {
                    exact_keys<QB>(a.c.X + (size_t)row * a.c.D4, qptr, a.c.D4, a.c.metric, key);
                } else {            // int8-only index: the int8 row, once per query (same chains, same keys)
#pragma unroll
                    for (int j = 0; j < QB; ++j)
                        key[j] = exact_key_i8<4>(a.c.X8 + (size_t)row * a.c.x8_pitch, a.c.cx, qptr[j], a.c.D4, a.c.metric);
                }
            }
            const int64_t id = a.c.idmap ? (valid ? a.c.idmap[row] : -1) : a.c.id_base + row;
#pragma unroll
            for (int j = 0; j < QB; ++j) tk[j].offer(key[j], id, valid);
        }
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            if (slot[j] < count) {
                if (a.D) {
                    const size_t o = (size_t)q[j] * a.c.k;
                    write_topk<KPL>(tk[j], a.c.metric, a.D + o, a.I + o, nullptr, nullptr);
                } else {
                    const size_t o = (size_t)(slot[j] * a.S + split) * a.c.k;
                    write_topk<KPL>(tk[j], a.c.metric, nullptr, nullptr, a.pkeys + o, a.pids + o);
                }
            }
        }
    }
}

// ---- merge of sorted partial lists ----------------------------------------------------------------
struct MergeArgs {
    const double *pkeys;
    const int64_t *pids;
    int64_t part_stride; // elements between consecutive parts of one slot
    int64_t slot_stride; // elements between consecutive slots of one part
    int nparts;
    int k;
    int metric;
    const int32_t *qlist;     // optional: slot -> output row
    const int32_t *count_ptr; // optional device-side slot count
    int64_t count;
    float *D;                 // final output (nq,k) ...
    int64_t *I;
    double *okeys;            // ... or partial output (count,k)
    int64_t *oids;
    // optional (multi-device index, multi.inc): the ids of part p are LOCAL row numbers of shard p; segment table =
    // [nparts + 1 offsets | (local0, id0) pairs grouped by part, ascending local0]: rows from local0 on have ids from id0 on
    const int64_t *seg;
};

template <int KPL>
__global__ __launch_bounds__(256) void merge_kernel(MergeArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t waves_total = (int64_t)gridDim.x * 4;
    const int64_t wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int64_t count = a.count_ptr ? (int64_t)*a.count_ptr : a.count;
    const int total = a.nparts * a.k;
    for (int64_t slot = wave0; slot < count; slot += waves_total) {
        WaveTopK<KPL> tk;
        tk.init(a.k);
        for (int base = 0; base < total; base += 64) {
            const int i = base + lane;
            bool valid = i < total;
            uint64_t key = ~0ull;
            int64_t id = -1;
            if (valid) {
                const int p = i / a.k, j = i - p * a.k;
                const size_t o = (size_t)p * a.part_stride + (size_t)slot * a.slot_stride + j;
                id = a.pids[o];
                key = sortable_u64(a.pkeys[o]);
                valid = id >= 0;
                if (valid && a.seg) {          // local row of shard p -> global id (the map is monotone: ties keep their order)
                    const int64_t *pairs = a.seg + a.nparts + 1;
                    int64_t e = a.seg[p];
                    const int64_t e1 = a.seg[p + 1];
                    while (e + 1 < e1 && pairs[2 * (e + 1)] <= id) ++e;
                    id = pairs[2 * e + 1] + (id - pairs[2 * e]);
                }
            }
            tk.offer(key, id, valid);
        }
        const int64_t q = a.qlist ? (int64_t)a.qlist[slot] : slot;
        const size_t o = (size_t)q * a.k;
        if (a.D)
            write_topk<KPL>(tk, a.metric, a.D + o, a.I + o, nullptr, nullptr);
        else
            write_topk<KPL>(tk, a.metric, nullptr, nullptr, a.okeys + o, a.oids + o);
    }
}

}  // namespace vdb
