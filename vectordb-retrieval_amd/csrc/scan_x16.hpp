// scan_x16.hpp -- the flat fp16 scan (D <= 128) on v_mfma_f32_16x16x32_f16, panel layout "x16" (round 4).
//
// The fp16 twin of scan_i8x16.hpp (read that header first: same tiles, same row mapping, same bins): in the production loop
// structure -- 4-tile stages by LDS-DMA, one barrier per stage, the two waves of a SIMD in anti-phase -- the 4-pass 16x16x32
// instruction delivers 1.17x (D = 128) / 1.19x (D = 64) the FLOP/s of the 8-pass 32x32x16 one with the same oct select
// (scripts/microbench/mfma_f16_staged.hip, profiles/r04_mfma_f16_staged.txt).  scan16_kernel (scan16.hpp), the earlier
// 16x16x32 kernel for D <= 128, lost to scan_kernel because a 16-row tile gives a lane one QUAD per column block (6 vector
// operations per 4 scores); here a 32-row tile is two 16-row blocks whose rows interleave in fours, so the lane's 4 + 4
// accumulator registers are 8 consecutive corpus rows -- one oct, 7 operations per 8 scores.
//   panels[tile][v = 2 ks2 + rb][lane][8 x fp16]: lane holds MFMA row m = lane & 15 of block rb -- corpus row
//   512 s + 128 (m >> 2) + 8 t + 4 rb + (m & 3) -- dims 32 ks2 + 8 (lane >> 4) .. +7;  qpanels[q / 16][ks2][lane] likewise.
// BR: rows per level-1 bin.  256 (default): lane groups 2h, 2h+1 merge their halves when the bin (span, h) is complete and each
// stores two of the wave's four column blocks.  128 / 64 ("direct" bins of scan_geometry_direct: large k on mid-size
// corpora): a lane group's 128 rows of a span are one or two bins of its own -- no merge, every lane stores its four blocks.
#pragma once
#include "scan16.hpp"
#include "scan_i8x16.hpp"

namespace vdb {

template <int KS2, int ST>
constexpr int scan_x16_lds_bytes() { return 2 * (ST * 2 * KS2 * 64 * 16 + ST * 32 * 4); }

// G: rows per candidate group -- 8 (octs), or 4 (quads: the lane's four rows of each block; direct-bin mode, where k is large and the
// exact refine of the float32 rows outweighs the scan, so the smaller group wins: 10k queries, k = 100 on 50k rows 1.20 -> 0.97 ms)
// CB: 16-query column blocks per wave: 4 (64 queries), or for D <= 64 also 8 (128 queries per wave, 1024-query workgroup tiles as
// scan_i8x16_kernel: every A fragment read from LDS feeds 8 MFMAs; at D = 128 the B fragments of 128 queries would take 128 VGPRs)
template <int KS2, int NWAVES, int ST, int BR = 256, bool DBG = false, int G = 8, int CB = 4>
__device__ __forceinline__ void scan_x16_body(const ScanArgs &a, unsigned char *smem) {
    constexpr int NV = 2 * KS2;
    static_assert(CB == 4 || (CB == 8 && KS2 <= 2), "128 queries per wave: D <= 64");
    constexpr int NT = NWAVES * 64;
    constexpr int kStageVec = ST * NV * 64;
    constexpr int kBiasLoads = (ST * 32 + NT - 1) / NT;
    constexpr int TPS = kTilesPerSpan;
    constexpr int SPS = TPS / ST;
    constexpr int BT = BR == 64 ? 8 : 16;                  // tiles per bin of a lane group
    constexpr bool kMerge = BR == 256;
    constexpr int NS = kMerge ? CB / 2 : CB;               // column blocks whose bins a lane stores
    constexpr int UNR = ST <= 4 ? ST : 1;                  // tiles of a stage unrolled together (8-tile stages keep the loop rolled)
    static_assert(TPS % ST == 0 && ST >= 2 && BT % ST == 0 && (BR == 256 || BR == 128 || BR == 64), "bad geometry");
    static_assert(G == 8 || (G == 4 && BR != 256), "quads: direct-bin mode");
    static_assert(scan_x16_lds_bytes<KS2, ST>() == 2 * (kStageVec * 16 + ST * 32 * 4), "LDS size");
    auto lds_a = [&](int buf) { return reinterpret_cast<half8 *>(smem + buf * (kStageVec * 16)); };
    auto lds_b = [&](int buf) { return reinterpret_cast<float *>(smem + 2 * kStageVec * 16 + buf * (ST * 32 * 4)); };
    if (a.info->i8_mode) return;                            // this batch is served by the int8 scan

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, h = g >> 1;
    const bool odd = (g & 1) != 0;
    const bool late = (NWAVES >= 2) && (wave >= NWAVES / 2);
    const int b = blockIdx.x;
    const int x = b & 7, jb = b >> 3;
    const int ci = jb / a.nqtiles, qt = jb - ci * a.nqtiles;
    int chunk = x + 8 * ci;
    if (chunk >= a.nchunks) return;
    chunk += a.chunk0;                                      // (a slab launch of an int8-only index)
    const int64_t q0 = (int64_t)qt * (NWAVES * 16 * CB) + wave * (16 * CB);
    const int64_t span0 = chunk_span0(chunk, a.spans_per_chunk, a.chunk_rem);
    int64_t span1 = chunk_span0(chunk + 1, a.spans_per_chunk, a.chunk_rem);
    if (span1 > a.nspans) span1 = a.nspans;
    // merged bins: the even lane group of a half stores column blocks 0, 2, the odd one 1, 3 (whole 128-byte lines per store)
    const int64_t col0 = q0 + ((kMerge && odd) ? 16 : 0) + (lane & 15);
    constexpr int CS = kMerge ? 32 : 16;                   // query columns between the blocks a lane stores
    const float cs = a.info->cs;

    half8 bq[CB][KS2];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) bq[cb][ks] = a.qpanels[((size_t)(q0 / 16 + cb) * KS2 + ks) * 64 + lane];
    const int nstages = (int)(span1 - span0) * SPS;
    const float INF = __builtin_inff();
    float NEG_INF = -INF;
    asm volatile("" : "+v"(NEG_INF));  // opaque, or LLVM folds med3(a,b,-inf) back into a canonicalising fmin
    unsigned idmask = kQuadIdMask;
    asm volatile("" : "+v"(idmask));
    float m1[CB], m2[CB], M1[NS], M2[NS];
    int Ms[NS];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) m1[cb] = m2[cb] = INF;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        M1[j] = M2[j] = INF;
        Ms[j] = 0;
    }

    constexpr int kPieces = kStageVec / 64;
    static_assert(kPieces % NWAVES == 0, "pieces must divide over the waves");
    float stage_b[kBiasLoads];
    auto stage_issue = [&](int st, int buf) {
        const int64_t span = span0 + st / SPS;
        const int sq = st % SPS;
        const half8 *src = a.panels + ((size_t)(span * TPS + sq * ST) * NV) * 64;
        half8 *dst = lds_a(buf);
#pragma unroll
        for (int i = 0; i < kPieces / NWAVES; ++i) {
            const int p = wave + i * NWAVES;
            const half8 *gp = src + p * 64 + lane;
            __builtin_amdgcn_global_load_lds(
                reinterpret_cast<const __attribute__((address_space(1))) void *>(reinterpret_cast<uintptr_t>(gp)),
                reinterpret_cast<__attribute__((address_space(3))) void *>(
                    static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dst + p * 64))),
                16, 0, NWAVES <= 2 ? 2 : 0);                  // (serving shapes read every panel byte once: non-temporal)
        }
#pragma unroll
        for (int i = 0; i < kBiasLoads; ++i) {
            const int e = tid + i * NT;
            if (e < ST * 32) {      // [tile][lane group][8 consecutive rows]; raw value only (no use before the DMA wait)
                const int t = e >> 5, gg = (e >> 3) & 3, r = e & 7;
                stage_b[i] = a.bias[span * kSpanRows + gg * 128 + (sq * ST + t) * 8 + r];
            }
        }
    };
    auto stage_bias_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < kBiasLoads; ++i)
            if (tid + i * NT < ST * 32)
                lds_b(buf)[tid + i * NT] = (stage_b[i] >= 0.9e38f) ? kPadBias : stage_b[i] * cs;
    };
    // a bin is complete.  BR = 256: bin (span, h), halves merged across lane ^ 16.  BR = 128 / 64: bin bt of lane group g.
    auto flush_bin = [&](int64_t span, int bt) {
        if (kMerge) {
            const size_t o = (size_t)(span * 2 + h) * a.Qpad + col0;
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const float p1 = __int_as_float(swap16(__float_as_int(odd ? m1[2 * j] : m1[2 * j + 1])));
                const float p2 = __int_as_float(swap16(__float_as_int(odd ? m2[2 * j] : m2[2 * j + 1])));
                const float a1 = odd ? m1[2 * j + 1] : m1[2 * j], a2 = odd ? m2[2 * j + 1] : m2[2 * j];
                const float b1 = __builtin_fminf(a1, p1), b2 = __builtin_fminf(__builtin_fmaxf(a1, p1), __builtin_fminf(a2, p2));
                __builtin_nontemporal_store(b1, a.bin_m1 + o + j * CS);
                __builtin_nontemporal_store(b2, a.bin_m2 + o + j * CS);
                M2[j] = __builtin_fminf(__builtin_amdgcn_fmed3f(M1[j], M2[j], b1), b2);
                if (b1 < M1[j]) Ms[j] = (int)span;
                M1[j] = __builtin_fminf(M1[j], b1);
            }
        } else {
            const size_t o = (size_t)((span * 4 + g) * (128 / BR) + bt) * a.Qpad + col0;
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                __builtin_nontemporal_store(m1[j], a.bin_m1 + o + j * 16);
                __builtin_nontemporal_store(m2[j], a.bin_m2 + o + j * 16);
                M2[j] = __builtin_fminf(__builtin_amdgcn_fmed3f(M1[j], M2[j], m1[j]), m2[j]);
                if (m1[j] < M1[j]) Ms[j] = (int)span;
                M1[j] = __builtin_fminf(M1[j], m1[j]);
            }
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) m1[cb] = m2[cb] = INF;
    };

    stage_issue(0, 0);
    stage_bias_store(0);
    __syncthreads();

    if (a.nq_valid > 0 && q0 >= a.nq_valid) {   // every query column of this wave is padding: keep staging + barriers going
        for (int st = 0; st < nstages; ++st) {
            if (st + 1 < nstages) {
                stage_issue(st + 1, (st & 1) ^ 1);
                stage_bias_store((st & 1) ^ 1);
            }
            __syncthreads();
        }
        return;
    }

    half8 fr[NV];
    float4v cin[2], acc[2][CB];
    auto read_phase = [&](const half8 *A_tile, const float4v *c_tile) {
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
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[2 * ks + rb], bq[cb][ks], ks == 0 ? cin[rb] : acc[rb][cb], 0, 0, 0);
    };
    // one oct per column block; id = its number inside the bin: tile of the bin (+ 16 for the upper half of a merged bin)
    const unsigned id_hi = (kMerge && odd) ? 16u : 0u;
    auto select_phase = [&](int t_span) {
        const unsigned idv = (unsigned)__builtin_amdgcn_readfirstlane(t_span % BT) | id_hi;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            if (G == 4) {       // quad id = 2 x tile of the bin + block
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const float qm = quad_min(acc[rb][cb][0], acc[rb][cb][1], acc[rb][cb][2], acc[rb][cb][3], NEG_INF);
                    const float v = pack_score(qm, idmask, 2u * idv + (unsigned)rb);
                    m2[cb] = __builtin_amdgcn_fmed3f(m1[cb], m2[cb], v);
                    m1[cb] = fast_min(m1[cb], v, NEG_INF);
                }
                continue;
            }
            const float o = oct_min(acc[0][cb][0], acc[0][cb][1], acc[0][cb][2], acc[0][cb][3], acc[1][cb][0], acc[1][cb][1],
                                    acc[1][cb][2], acc[1][cb][3], NEG_INF);
            const float v = pack_score(o, idmask, idv);
            m2[cb] = __builtin_amdgcn_fmed3f(m1[cb], m2[cb], v);
            m1[cb] = fast_min(m1[cb], v, NEG_INF);
        }
    };
    unsigned long long c_head = 0, c_mfma = 0, c_sel = 0, c_bar = 0, t_start = 0, ta = 0, tb = 0;
#define tick(bucket)                    \
    do {                                \
        if (DBG) {                      \
            tb = stamp();               \
            bucket += tb - ta;          \
            ta = tb;                    \
        }                               \
    } while (0)
#define done()                                                                                                        \
    do {                                                                                                              \
        if (DBG) { _Pragma("unroll") for (int cb__ = 0; cb__ < CB; ++cb__) asm volatile("s_nop 0" ::"v"(acc[0][cb__]), "v"(acc[1][cb__])); } \
    } while (0)
    if (a.prio == 1 && late) __builtin_amdgcn_s_setprio(1);
    else if (a.prio == 2 && !late) __builtin_amdgcn_s_setprio(1);
    if (DBG) t_start = stamp();
    if (!late) {
        for (int st = 0; st < nstages; ++st) {
            const int buf = st & 1;
            if (DBG) ta = stamp();
            if (st + 1 < nstages) stage_issue(st + 1, buf ^ 1);
            const half8 *A = lds_a(buf);
            const float4v *B4 = reinterpret_cast<const float4v *>(lds_b(buf)) + g * 2;
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
            if (((ts0 + ST) % BT) == 0) flush_bin(span0 + st / SPS, (ts0 + ST) / BT - 1);
            if (st + 1 < nstages) stage_bias_store(buf ^ 1);
            tick(c_sel);
            __syncthreads();
            tick(c_bar);
        }
    } else {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = float4v{3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};   // dummy previous tile
        for (int st = 0; st < nstages; ++st) {
            const int buf = st & 1;
            if (DBG) ta = stamp();
            if (st + 1 < nstages) stage_issue(st + 1, buf ^ 1);
            const half8 *A = lds_a(buf);
            const float4v *B4 = reinterpret_cast<const float4v *>(lds_b(buf)) + g * 2;
            const int ts0 = (st % SPS) * ST;
            tick(c_head);
#pragma unroll UNR
            for (int t = 0; t < ST; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                read_phase(A + t * NV * 64, B4 + t * 8);
                const int tp = (ts0 + t + TPS - 1) % TPS;        // the previous tile (of the span before this one when ts0 + t == 0)
                select_phase(tp);
                if (t == 0 && st > 0 && (ts0 % BT) == 0) flush_bin(span0 + (st * ST - 1) / TPS, tp / BT);
                __builtin_amdgcn_sched_barrier(0);
                tick(c_sel);
                mfma_phase();
                done();
                tick(c_mfma);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 < nstages) stage_bias_store(buf ^ 1);
            tick(c_sel);
            __syncthreads();
            tick(c_bar);
        }
        select_phase(TPS - 1);
        flush_bin(span1 - 1, TPS / BT - 1);
    }
    if (DBG && a.dbg && lane == 0) {       // (record format of scan_kernel's stamped build: scripts/stamp_scan.py)
        const unsigned long long t_end = stamp();
        unsigned long long *d = a.dbg + ((size_t)blockIdx.x * NWAVES + wave) * 8;
        d[0] = c_head; d[1] = c_mfma; d[2] = c_sel; d[3] = c_bar; d[4] = t_end - t_start; d[5] = late ? 1 : 0;
        d[6] = (unsigned long long)nstages;
    }
#undef tick
#undef done

    // superbins (chunk, h) of the 256-row geometry; in direct-bin mode the level-1 bins are the superbins and these go unread
    if (kMerge) {
        const size_t so = (size_t)(chunk * 2 + h) * a.Qpad + col0;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            a.sb_m1[so + j * CS] = M1[j];
            a.sb_m2[so + j * CS] = M2[j];
            a.sb_span[so + j * CS] = Ms[j];
        }
    }
}

template <int KS2, int NWAVES, int ST, int WPS, int BR = 256, bool DBG = false, int G = 8, int CB = 4>
__global__ __launch_bounds__(NWAVES * 64, WPS) void scan_x16_kernel(ScanArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[scan_x16_lds_bytes<KS2, ST>()];
    scan_x16_body<KS2, NWAVES, ST, BR, DBG, G, CB>(a, smem);
}

// Both scans of an index that holds the int8 copy, in ONE launch: which of them serves a batch is decided on the device
// (QueryBatchInfo.i8_mode), so both used to be enqueued and the one not needed returned at once -- a 4.5 us dispatch plus its gap on
// every search, a tenth of a single-query search.  Here every workgroup reads the flag and runs the body of the scan that serves the
// batch; the grid is the larger of the two (the body's own chunk test retires the workgroups beyond its grid), the LDS block the
// larger of the two, registers the larger of the two (254 / 226: both kernels already run 2 waves per SIMD).
// W128: D in (64, 128] (fp16 KS2 = 4, int8 KS2 = 2), else D <= 64 (2 / 1).  STF: tiles per fp16 stage; STI / CBI / RING / AUX: the int8 shape.
template <bool W128, int NWAVES, int STF, int STI, int CBI, int RING, int AUX>
__global__ __launch_bounds__(NWAVES * 64, (NWAVES >= 4 ? 2 : 1)) void scan_pair_x16_kernel(ScanArgs a16, ScanI8Args a8) {
    constexpr int KF = W128 ? 4 : 2, KI = W128 ? 2 : 1;
    constexpr int kLds16 = scan_x16_lds_bytes<KF, STF>(), kLds8 = scan_i8x16_lds_bytes<KI, STI, RING>();
    __shared__ __attribute__((aligned(16))) unsigned char smem[kLds16 > kLds8 ? kLds16 : kLds8];
    if (a8.info->i8_mode) scan_i8x16_body<KI, STI, CBI, NWAVES, false, RING, AUX>(a8, smem);
    else scan_x16_body<KF, NWAVES, STF, 256, false>(a16, smem);
}

}  // namespace vdb
