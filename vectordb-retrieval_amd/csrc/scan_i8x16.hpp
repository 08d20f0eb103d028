// scan_i8x16.hpp -- the flat int8 scan on v_mfma_i32_16x16x64_i8 (layout "x16" of the int8 panels, round 4).
//
// Why a second shape: beside the scan's select, the 16x16x64 instruction (8 passes, 4 accumulator registers) keeps the matrix
// pipe 0.83 busy where 32x32x32 (16 passes, 16 registers) reaches 0.66 in the SAME loop structure -- barrier per stage, LDS-DMA
// staging, the two waves of a SIMD in anti-phase (scripts/microbench/mfma_i8_shapes.hip, rows "+bar+DMA+stag": 2 720 -> 3 480
// TOP/s, profiles/r04_mfma_i8_shapes.txt): the shorter instruction lets the two waves of a SIMD interleave MFMA and select work at
// half the granularity, and a wave needs 9 instead of 14 vector operations per 16 scores for its oct minima.
//
// Everything OUTSIDE the kernel is unchanged -- bins are the same 256 consecutive corpus rows, a packed key names the same oct
// (8 consecutive rows at 8 x id from the bin's first row), superbins are (chunk, span half): the select and refine kernels, the
// fp16 scan of the same index and the geometry do not know which shape wrote the bins.  What changes:
//   * a 32-row tile is two 16-row MFMA blocks rb = 0, 1.  In the 16x16 C/D layout lane (col = lane & 15, g = lane >> 4) holds
//     rows 4g .. 4g+3 of a block, so per tile and query a lane owns ONE oct: its 4 rows of both blocks.  MFMA row m of block rb
//     of tile t in span s is corpus row  512 s + 128 (m >> 2) + 8 t + 4 rb + (m & 3):  lane group g walks 128 consecutive rows
//     of the span, groups 2h and 2h+1 together the 256 rows of bin (s, h).
//   * panels8[tile][v = 2 ks2 + rb][lane][16 x int8]: lane holds MFMA row (lane & 15), dims 64 ks2 + 16 (lane >> 4) .. +15
//     (same bytes per tile as the 32-row layout: 2 or 4 KiB); qpanels8[q / 16][ks2][lane]: query column (lane & 15), same dims.
//   * a bin is complete after 16 tiles: the two lane groups of a half exchange their minima (ds_swizzle, lane ^ 16), both
//     merge, and each stores every other column block of the wave -- ~40 vector operations per 16 tiles beside ~800 of select.
#pragma once
#include "scan_i8.hpp"

namespace vdb {

// corpus row (offset inside its 512-row span) of MFMA row m (0..15) of block rb of tile t, layout "x16"
__host__ __device__ inline int x16_row_in_span(int t, int rb, int m) { return (m >> 2) * 128 + t * 8 + rb * 4 + (m & 3); }

__device__ __forceinline__ void wait_for_mfma4(const int4v &a, const int4v &b) { asm volatile("s_nop 0" ::"v"(a), "v"(b)); }   // (as wait_for_mfma)
__device__ __forceinline__ int swap16(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x401F); }   // lane ^ 16 (and 0x1f, xor 0x10)

// KS2: 64-dim k-steps (D padded to 64 or 128); ST: 32-row tiles per LDS stage; CB: 16-query column blocks per wave (8 -> 128
// queries per wave, 4 -> 64); NWAVES, RING, AUX, DBG as scan_i8_kernel.  Octs only (QueryBatchInfo.i8_mode bit 2 is set by
// the host for an index in this layout).
template <int KS2, int ST, int RING>
constexpr int scan_i8x16_lds_bytes() { return RING * (ST * 2 * KS2 * 64 * 16 + ST * 32 * 4); }

// (the body is a device function over a caller-provided LDS block so that scan_pair_x16_kernel, scan_x16.hpp, can hold it
//  next to the fp16 scan in ONE launch)
template <int KS2, int ST, int CB, int NWAVES = 8, bool DBG = false, int RING = 2, int AUX = 0>
__device__ __forceinline__ void scan_i8x16_body(const ScanI8Args &a, unsigned char *smem) {
    constexpr int NV = 2 * KS2;                           // 16-byte fragments per lane and tile
    constexpr int NT = NWAVES * 64;
    constexpr int kStageVec = ST * NV * 64;               // 16-byte vectors per stage
    constexpr int kBiasLoads = (ST * 32 + NT - 1) / NT;
    constexpr int TPS = kTilesPerSpan;                    // 16 tiles of 32 rows
    constexpr int SPS = TPS / ST;
    constexpr int HC = CB / 2;                            // column blocks whose bins a lane stores
    // tiles of a stage unrolled together: the serving shapes (one wave per SIMD, nothing to overlap with but the wave's own next
    // tile); the batch shape keeps the tile loop rolled (as scan_i8_kernel)
    constexpr int UNR = (ST * CB <= 16) ? ST : 1;
    static_assert(TPS % ST == 0 && ST >= 2 && CB % 2 == 0 && NWAVES <= 8, "bad geometry");
    static_assert(RING >= 2 && RING <= 8 && (RING == 2 || ST % 2 == 0), "bad ring");
    constexpr bool kDeep = RING > 2;
    static_assert(scan_i8x16_lds_bytes<KS2, ST, RING>() == RING * (kStageVec * 16 + ST * 32 * 4), "LDS size");
    const int mode = a.info->i8_mode;
    if (!mode) return;                                    // this batch is served by the fp16 scan
    auto lds_a = [&](int buf) { return reinterpret_cast<int4v *>(smem + buf * (kStageVec * 16)); };
    auto lds_b = [&](int buf) { return reinterpret_cast<int *>(smem + RING * kStageVec * 16 + buf * (ST * 32 * 4)); };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, h = g >> 1;
    const bool odd = (g & 1) != 0;
    const bool late = (NWAVES >= 2) && (wave >= NWAVES / 2);
    const int b = blockIdx.x;
    const int x = b & 7, jb = b >> 3;
    const int ci = jb / a.nqtiles, qt = jb - ci * a.nqtiles;
    const int chunk = x + 8 * ci;
    if (chunk >= a.nchunks) return;
    const int64_t q0 = (int64_t)qt * (NWAVES * 16 * CB) + wave * (16 * CB);
    const int64_t span0 = chunk_span0(chunk, a.spans_per_chunk, a.chunk_rem);
    int64_t span1 = chunk_span0(chunk + 1, a.spans_per_chunk, a.chunk_rem);
    if (span1 > a.nspans) span1 = a.nspans;
    const int64_t out_pitch = a.Qpad;
    // the two lane groups of a span half share the stores of its bins: the even group takes column blocks 0, 2, 4, .., the odd one
    // 1, 3, 5, ..: one store instruction then writes 32 consecutive queries = a whole 128-byte line per bin
    const int64_t col0 = q0 + (odd ? 16 : 0) + (lane & 15);
    const int32_t *bias = a.bias8 + ((mode & 3) == 1 ? 0 : a.Npad);

    int4v bq[CB][KS2];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) bq[cb][ks] = a.qpanels[((size_t)(q0 / 16 + cb) * KS2 + ks) * 64 + lane];
    const int nstages = (int)(span1 - span0) * SPS;
    const int INF = (int)kI8Inf;
    int m1[CB], m2[CB], M1[HC], M2[HC], Ms[HC];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) m1[cb] = m2[cb] = INF;
#pragma unroll
    for (int j = 0; j < HC; ++j) {
        M1[j] = M2[j] = INF;
        Ms[j] = 0;
    }

    constexpr int kPieces = kStageVec / 64;
    static_assert(kPieces % NWAVES == 0, "pieces must divide over the waves");
    constexpr int kBiasPieces = (ST / 2 + NWAVES - 1) / NWAVES;       // ring mode: bias requests per wave and stage
    int stage_b[kBiasLoads];
    // accumulator inits of a stage in LDS: [tile][lane group g][8 ints] = the 8 consecutive rows of (g, tile) -- bias8 is in corpus order
    auto stage_issue = [&](int st, int buf) {
        const int64_t span = span0 + st / SPS;
        const int sq = st % SPS;
        const int4v *src = a.panels + ((size_t)(span * TPS + sq * ST) * NV) * 64;
        int4v *dst = lds_a(buf);
#pragma unroll
        for (int i = 0; i < kPieces / NWAVES; ++i) {
            const int p = wave + i * NWAVES;
            const int4v *gp = src + p * 64 + lane;
            __builtin_amdgcn_global_load_lds(
                reinterpret_cast<const __attribute__((address_space(1))) void *>(reinterpret_cast<uintptr_t>(gp)),
                reinterpret_cast<__attribute__((address_space(3))) void *>(
                    static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dst + p * 64))),
                16, 0, AUX);
        }
        if (kDeep) {
#pragma unroll
            for (int i = 0; i < kBiasPieces; ++i) {
                const int j = (wave + i * NWAVES) % (ST / 2);
                const int t = 2 * j + (lane >> 5), gg = (lane >> 3) & 3, e = lane & 7;
                const int32_t *gp = bias + span * kSpanRows + gg * 128 + (sq * ST + t) * 8 + e;
                __builtin_amdgcn_global_load_lds(
                    reinterpret_cast<const __attribute__((address_space(1))) void *>(reinterpret_cast<uintptr_t>(gp)),
                    reinterpret_cast<__attribute__((address_space(3))) void *>(
                        static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_b(buf) + j * 64))),
                    4, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < kBiasLoads; ++i) {
            const int e = tid + i * NT;
            if (e < ST * 32) {
                const int t = e >> 5, gg = (e >> 3) & 3, r = e & 7;
                stage_b[i] = bias[span * kSpanRows + gg * 128 + (sq * ST + t) * 8 + r];
            }
        }
    };
    auto stage_bias_store = [&](int buf) {
        if (kDeep) return;
#pragma unroll
        for (int i = 0; i < kBiasLoads; ++i)
            if (tid + i * NT < ST * 32) lds_b(buf)[tid + i * NT] = stage_b[i];
    };
    auto ring_issue = [&](int st) { stage_issue(st < nstages ? st : nstages - 1, st % RING); };
    auto ring_wait = [&]() {
        constexpr int kKeep = (RING - 2) * (kPieces / NWAVES + kBiasPieces);
        static_assert(kKeep < 64, "vmcnt is a 6-bit counter");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kKeep) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // bin (span, h) is complete: lane groups 2h and 2h+1 hold the minima of its two 128-row halves
    auto flush_bin = [&](int64_t span) {
        const size_t o = (size_t)(span * 2 + h) * out_pitch + col0;
#pragma unroll
        for (int j = 0; j < HC; ++j) {
            const int p1 = swap16(odd ? m1[2 * j] : m1[2 * j + 1]), p2 = swap16(odd ? m2[2 * j] : m2[2 * j + 1]);
            const int a1 = odd ? m1[2 * j + 1] : m1[2 * j], a2 = odd ? m2[2 * j + 1] : m2[2 * j];
            const int b1 = imin(a1, p1), b2 = imin(imax(a1, p1), imin(a2, p2));
#ifdef VDB_ABLATIONS
            if (!a.abl_no_bins)
#endif
            {
                __builtin_nontemporal_store(__int_as_float(b1), a.bin_m1 + o + j * 32);
                __builtin_nontemporal_store(__int_as_float(b2), a.bin_m2 + o + j * 32);
            }
            M2[j] = imin(imed3(M1[j], M2[j], b1), b2);
            if (b1 < M1[j]) Ms[j] = (int)span;
            M1[j] = imin(M1[j], b1);
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) m1[cb] = m2[cb] = INF;
    };

    if (kDeep) {
#pragma unroll
        for (int s = 0; s < RING - 1; ++s) ring_issue(s);
    } else {
        stage_issue(0, 0);
    }
    stage_bias_store(0);
    if (kDeep) ring_wait();
    else __syncthreads();

    if (a.nq_valid > 0 && q0 >= a.nq_valid) {   // every query column of this wave is padding: keep staging + barriers going
        for (int st = 0; st < nstages; ++st) {
            if (kDeep) {
                ring_issue(st + RING - 1);
                ring_wait();
                continue;
            }
            if (st + 1 < nstages) {
                stage_issue(st + 1, (st & 1) ^ 1);
                stage_bias_store((st & 1) ^ 1);
            }
            __syncthreads();
        }
        return;
    }

    int4v fr[NV], cin[2], acc[2][CB];
    auto read_phase = [&](const int4v *A_tile, const int4v *c_tile) {
#pragma unroll
        for (int v = 0; v < NV; ++v) fr[v] = A_tile[v * 64 + lane];
        cin[0] = c_tile[0];
        cin[1] = c_tile[1];
    };
    auto mfma_phase = [&]() {
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    acc[rb][cb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fr[2 * ks + rb], bq[cb][ks], ks == 0 ? cin[rb] : acc[rb][cb], 0, 0, 0);
    };
    // one oct per column block: v = (min of the lane's 8 rows << 6) | oct id, id = 16 (g & 1) + tile of the bin
    const unsigned id_hi = odd ? 16u : 0u;
    auto select_phase = [&](int t_bin) {
        const unsigned idv = (unsigned)__builtin_amdgcn_readfirstlane(t_bin) | id_hi;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int t1 = imin(imin(acc[0][cb][0], acc[0][cb][1]), acc[0][cb][2]);
            const int t2 = imin(imin(acc[0][cb][3], acc[1][cb][0]), acc[1][cb][1]);
            const int q = imin(imin(imin(acc[1][cb][2], acc[1][cb][3]), t1), t2);
            const int v = (int)(((unsigned)q << 6) | idv);
            m2[cb] = imed3(m1[cb], m2[cb], v);
            m1[cb] = imin(m1[cb], v);
        }
    };
    unsigned long long c_head = 0, c_mfma = 0, c_sel = 0, c_tail = 0, c_bar = 0, t_last = 0, t_first = 0, r_first = 0;
#define tick(bucket)                                   \
    do {                                               \
        if (DBG) {                                     \
            const unsigned long long t__ = stamp();    \
            bucket += t__ - t_last;                    \
            t_last = t__;                              \
        }                                              \
    } while (0)
#define done()                                                                                                    \
    do {                                                                                                          \
        if (DBG) { _Pragma("unroll") for (int cb__ = 0; cb__ < CB; ++cb__) wait_for_mfma4(acc[0][cb__], acc[1][cb__]); } \
    } while (0)
    if (a.prio == 1 && late) __builtin_amdgcn_s_setprio(1);
    else if (a.prio == 2 && !late) __builtin_amdgcn_s_setprio(1);
    if (DBG) {
        t_first = t_last = stamp();
        r_first = realtime_ticks();
    }
    if (!late) {
        for (int st = 0; st < nstages; ++st) {
            const int buf = kDeep ? st % RING : st & 1;
            if (kDeep) ring_issue(st + RING - 1);
            else if (st + 1 < nstages) stage_issue(st + 1, buf ^ 1);
            const int4v *A = lds_a(buf);
            const int4v *B4 = reinterpret_cast<const int4v *>(lds_b(buf)) + g * 2;
            const int ts0 = (st % SPS) * ST;
            read_phase(A, B4);
            tick(c_head);
#pragma unroll UNR
            for (int t = 0; t < ST; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                mfma_phase();
                __builtin_amdgcn_sched_barrier(0);
                done();
                tick(c_mfma);
                if (t + 1 < ST) read_phase(A + (t + 1) * NV * 64, B4 + (t + 1) * 8);
                select_phase(ts0 + t);
                tick(c_sel);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ts0 + ST == TPS) flush_bin(span0 + st / SPS);
            if (!kDeep && st + 1 < nstages) stage_bias_store(buf ^ 1);
            tick(c_tail);
            if (kDeep) ring_wait();
            else __syncthreads();
            tick(c_bar);
        }
    } else {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[rb][cb][r] = (int)(kI8Inf >> 6);   // dummy "previous tile": (x << 6) == "+inf", never wins
        for (int st = 0; st < nstages; ++st) {
            const int buf = kDeep ? st % RING : st & 1;
            if (kDeep) ring_issue(st + RING - 1);
            else if (st + 1 < nstages) stage_issue(st + 1, buf ^ 1);
            const int4v *A = lds_a(buf);
            const int4v *B4 = reinterpret_cast<const int4v *>(lds_b(buf)) + g * 2;
            const int ts0 = (st % SPS) * ST;
            tick(c_head);
#pragma unroll UNR
            for (int t = 0; t < ST; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                read_phase(A + t * NV * 64, B4 + t * 8);
                select_phase((ts0 + t + TPS - 1) % TPS);
                if (t == 0 && st > 0 && ts0 == 0) flush_bin(span0 + st / SPS - 1);
                __builtin_amdgcn_sched_barrier(0);
                tick(c_sel);
                mfma_phase();
                done();
                tick(c_mfma);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!kDeep && st + 1 < nstages) stage_bias_store(buf ^ 1);
            tick(c_tail);
            if (kDeep) ring_wait();
            else __syncthreads();
            tick(c_bar);
        }
        select_phase(TPS - 1);
        flush_bin(span1 - 1);
    }
    if (DBG && a.dbg && lane == 0) {
        const unsigned long long t_end = stamp(), r_end = realtime_ticks();
        unsigned long long *d = a.dbg + ((size_t)blockIdx.x * NWAVES + wave) * 8;
        d[0] = c_head; d[1] = c_mfma; d[2] = c_sel; d[3] = c_tail; d[4] = c_bar; d[5] = t_end - t_first;
        d[6] = r_end - r_first; d[7] = ((unsigned long long)nstages << 1) | (late ? 1ull : 0ull);
    }
#undef tick
#undef done

    const size_t so = (size_t)(chunk * 2 + h) * a.Qpad + col0;
#pragma unroll
    for (int j = 0; j < HC; ++j) {
        a.sb_m1[so + j * 32] = __int_as_float(M1[j]);
        a.sb_m2[so + j * 32] = __int_as_float(M2[j]);
        a.sb_span[so + j * 32] = Ms[j];
    }
}

template <int KS2, int ST, int CB, int NWAVES = 8, bool DBG = false, int RING = 2, int AUX = 0>
__global__ __launch_bounds__(NWAVES * 64, (NWAVES >= 4 ? 2 : 1)) void scan_i8x16_kernel(ScanI8Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[scan_i8x16_lds_bytes<KS2, ST, RING>()];
    scan_i8x16_body<KS2, ST, CB, NWAVES, DBG, RING, AUX>(a, smem);
}

}  // namespace vdb
