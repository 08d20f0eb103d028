// scan16.hpp -- scan kernels on the "p16" panel layout (16-row tiles, v_mfma_f32_16x16x32_f16).
//
// Why a second MFMA shape: at these operand mixes the MI355X sits at its power limit, and the 16x16x32 shape
// delivers 1.07-1.16x the FLOP/s of 32x32x16 for the same flops and the same LDS bytes per flop
// (scripts/microbench/mfma_shape.hip).  The bin / superbin semantics and the packed quad ids are
// those of scan.hpp: in the 16x16 C/D layout (col = lane&15, rows 4g..4g+3, g = lane>>4) a lane owns ONE query
// column and its 4 accumulator registers are 4 consecutive corpus rows = one quad; the p16 row mapping
// (common.hpp) makes lane group g walk the 256 consecutive rows of bin (span, g), one quad per tile, so a packed
// minimum's 6-bit id is the tile number inside the span and bin b covers rows [256 b, 256 b + 256).
#pragma once

#include "scan.hpp"

namespace vdb {

typedef float float4v __attribute__((ext_vector_type(4)));

// ---- flat scan for D <= 128 on p16 panels ---------------------------------------------------------------
// Same workgroup geometry and phase stagger as scan_kernel: 8 waves = (corpus chunk) x (512-query tile), wave w
// owns 64 queries as four 16-query column blocks whose B fragments stay in registers for the whole chunk
// (KS x 4 x 4 VGPRs).  A tile is 16 rows: KS A fragments (ds_read_b128) + one broadcast ds_read_b128 of the 4
// biases of the lane's rows, 4*KS v_mfma_f32_16x16x32_f16, and a select phase of 4 quads (one per column
// block): 3 + 1 + 2 VALU each.  ST tiles per LDS stage (DMA, double buffered, one barrier per stage); the four
// 256-row bins (span, g) of a span complete together after its 64th tile.
// ABL: timing-only builds (wrong results): 1 = no select, 2 = no MFMA.
template <int KS, int ST, int ABL = 0>
__global__ __launch_bounds__(512, 2) void scan16_kernel(ScanArgs a) {
    constexpr int NWAVES = 8, NT = 512, CB = 4;
    constexpr int kStageVec = ST * KS * 64;                  // 16-byte vectors per stage
    constexpr int SPS = kTilesPerSpan16 / ST;                // stages per span
    constexpr int kPieces = ST * KS;                         // 1-KiB pieces per stage
    static_assert(kTilesPerSpan16 % ST == 0 && kPieces % NWAVES == 0 && ST * 16 <= NT, "bad geometry");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (kStageVec * 16 + ST * 16 * 4)];
    auto lds_a = [&](int buf) { return reinterpret_cast<half8 *>(smem + buf * (kStageVec * 16)); };
    auto lds_b = [&](int buf) { return reinterpret_cast<float *>(smem + 2 * kStageVec * 16 + buf * (ST * 16 * 4)); };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const bool late = wave >= NWAVES / 2;
    const int b = blockIdx.x;
    const int x = b & 7, j = b >> 3;
    const int ci = j / a.nqtiles, qt = j - ci * a.nqtiles;
    const int chunk = x + 8 * ci;
    if (chunk >= a.nchunks) return;
    const int64_t q0 = (int64_t)qt * (NWAVES * 64) + wave * 64;
    const int64_t span0 = chunk_span0(chunk, a.spans_per_chunk, a.chunk_rem);
    int64_t span1 = chunk_span0(chunk + 1, a.spans_per_chunk, a.chunk_rem);
    if (span1 > a.nspans) span1 = a.nspans;
    const float cs = a.info->cs;

    half8 bq[CB][KS];                                        // B fragments: resident for the whole chunk
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bq[cb][ks] = a.qpanels[((size_t)(q0 / 16 + cb) * KS + ks) * 64 + lane];

    const int nstages = (int)(span1 - span0) * SPS;
    const float INF = __builtin_inff();
    float NEG_INF = -INF;
    asm volatile("" : "+v"(NEG_INF));  // opaque, or LLVM folds med3(a,b,-inf) back into a canonicalising fmin
    unsigned idmask = kQuadIdMask;
    asm volatile("" : "+v"(idmask));
    float m1[CB], m2[CB], M1[CB], M2[CB];
    int Ms[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        m1[cb] = m2[cb] = M1[cb] = M2[cb] = INF;
        Ms[cb] = 0;
    }

    const int lane16 = lane * 16;
    float stage_bv = 0.f;
    auto stage_issue = [&](int st, int buf) {
        const int64_t span = span0 + st / SPS;
        const int sq = st % SPS;
        const half8 *base = a.panels + ((size_t)(span * kTilesPerSpan16 + sq * ST) * KS) * 64;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half8 *>(base), 0, 0x7fffffff, 0x00020000);
        half8 *dst = lds_a(buf);
#pragma unroll
        for (int i = 0; i < kPieces / NWAVES; ++i) {
            const int p = wave + i * NWAVES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rs, reinterpret_cast<__attribute__((address_space(3))) void *>(
                        static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dst + p * 64))),
                16, lane16, p * 1024, 0, 0);
        }
        if (tid < ST * 16) {   // bias of (tile t, MFMA row rho = 4 gg + i): raw value only (no use before the DMA wait)
            const int t = tid >> 4, gg = (tid >> 2) & 3, i = tid & 3;
            stage_bv = a.bias[span * kSpanRows16 + gg * kBinRows + (sq * ST + t) * 4 + i];
        }
    };
    auto stage_bias_store = [&](int buf) {
        if (tid < ST * 16) lds_b(buf)[tid] = (stage_bv >= 0.9e38f) ? kPadBias : stage_bv * cs;
    };
    auto flush_bins = [&](int64_t span) {     // the four bins (span, g) are complete: write them, fold level 2
        const size_t o = (size_t)(span * 4 + g) * a.Qpad + q0 + (lane & 15);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            a.bin_m1[o + cb * 16] = m1[cb];
            a.bin_m2[o + cb * 16] = m2[cb];
            M2[cb] = __builtin_fminf(__builtin_amdgcn_fmed3f(M1[cb], M2[cb], m1[cb]), m2[cb]);
            if (m1[cb] < M1[cb]) Ms[cb] = (int)span;
            M1[cb] = __builtin_fminf(M1[cb], m1[cb]);
            m1[cb] = INF;
            m2[cb] = INF;
        }
    };

    half8 fr[KS];
    float4v cin, acc[CB];
    auto read_phase = [&](const half8 *A_tile, const float *b_tile) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fr[ks] = A_tile[ks * 64 + lane];
        const float4 c = *reinterpret_cast<const float4 *>(b_tile + 4 * g);
        cin[0] = c.x; cin[1] = c.y; cin[2] = c.z; cin[3] = c.w;
    };
    auto mfma_phase = [&]() {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                if (ABL != 2)
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[ks], bq[cb][ks], ks == 0 ? cin : acc[cb], 0, 0, 0);
                else
                    acc[cb] = (ks == 0 ? cin : acc[cb]) + (float)fr[ks][0] * (float)bq[cb][ks][0];
            }
    };
    auto select_phase = [&](unsigned id) {   // id = tile number inside the span = quad number inside the bin
        if (ABL == 1) {
            asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
            return;
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const float qm = quad_min(acc[cb][0], acc[cb][1], acc[cb][2], acc[cb][3], NEG_INF);
            const float v = pack_score(qm, idmask, id);
            m2[cb] = __builtin_amdgcn_fmed3f(m1[cb], m2[cb], v);
            m1[cb] = fast_min(m1[cb], v, NEG_INF);
        }
    };

    stage_issue(0, 0);
    stage_bias_store(0);
    __syncthreads();  // (drains the DMA: hipcc waits vmcnt(0) in front of the barrier)

    if (!late) {
        // ================= early half: MFMA(t), then select(t) =======================================
#pragma unroll 1
        for (int st = 0; st < nstages; ++st) {
            const int buf = st & 1;
            if (st + 1 < nstages) stage_issue(st + 1, buf ^ 1);  // buf^1 was last read before the previous barrier
            const half8 *A = lds_a(buf);
            const float *Bv = lds_b(buf);
            const int ts0 = (st % SPS) * ST;                     // first tile of this stage inside its span
            read_phase(A, Bv);
#pragma unroll
            for (int t = 0; t < ST; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                mfma_phase();
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < ST) read_phase(A + (t + 1) * KS * 64, Bv + (t + 1) * 16);
                select_phase((unsigned)(ts0 + t));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ts0 + ST == kTilesPerSpan16) flush_bins(span0 + st / SPS);
            if (st + 1 < nstages) stage_bias_store(buf ^ 1);
            __syncthreads();
        }
    } else {
        // ================= late half: select(t-1), then MFMA(t) =====================================
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[cb] = float4v{3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};  // dummy previous tile
#pragma unroll 1
        for (int st = 0; st < nstages; ++st) {
            const int buf = st & 1;
            if (st + 1 < nstages) stage_issue(st + 1, buf ^ 1);
            const half8 *A = lds_a(buf);
            const float *Bv = lds_b(buf);
            const int ts0 = (st % SPS) * ST;
#pragma unroll
            for (int t = 0; t < ST; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                read_phase(A + t * KS * 64, Bv + t * 16);
                // retire the previous tile (the last tile of the previous span when ts0 + t == 0)
                select_phase((unsigned)((ts0 + t + kTilesPerSpan16 - 1) % kTilesPerSpan16));
                if (t == 0 && st > 0 && ts0 == 0) flush_bins(span0 + st / SPS - 1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_phase();
            }
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 < nstages) stage_bias_store(buf ^ 1);
            __syncthreads();
        }
        select_phase((unsigned)(kTilesPerSpan16 - 1));  // drain the last tile
        flush_bins(span1 - 1);
    }

    const size_t so = (size_t)(chunk * 4 + g) * a.Qpad + q0 + (lane & 15);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        a.sb_m1[so + cb * 16] = M1[cb];
        a.sb_m2[so + cb * 16] = M2[cb];
        a.sb_span[so + cb * 16] = Ms[cb];
    }
}

// ---- K-loop scan for D > 128 on p16 panels ---------------------------------------------------------------
// Workgroup = 8 waves = (corpus chunk) x (512-query tile); wave w owns four 16-query column blocks and, per pass,
// eight 16-row tiles (8 x 4 x 4 = 128 accumulator registers: a 128-row x 64-query output tile, as the 32x32 form).
// Per 64-dim K-step (two 32-dim k-steps) the pass's A panels (16 KiB) arrive in LDS by buffer-addressed LDS-DMA
// and are shared by the 8 waves; each wave streams its own B fragments (8 x 1 KiB per K-step) from L2 in place.
// A fragments are software pipelined one group (4 ds_read_b128 -> 16 MFMAs) ahead; 4 groups per K-step:
// (k-step 0, tiles 0-3), (0, 4-7), (1, 0-3), (1, 4-7).  Eight passes complete the four 256-row bins of a span.
// BS / ring / barrier placement / ABL builds: see scan_kloop_kernel.
// FP: passes per level-1 bin (8: the 256-row bins (span, g); 4 / 2: 128- / 64-row bins for the direct-bin select).
// NARROW (batches below 64 queries: one wave per workgroup holds queries, and it alone on its SIMD cannot hide a global
// load per K-step behind 16 MFMAs): 1 = MFMAs and select work only for the 16-query column blocks that hold queries
// (1 ... 4); 2 = at most 16 queries and D <= 1536: additionally the B fragments of that one column block are kept in LDS
// for the whole chunk (kNarrowQVec vectors) instead of being streamed from L2 every K-step.
constexpr int kNarrowMaxKS = 48;                            // 32-dim k-steps of the LDS-resident query block (D <= 1536)
template <int ABL, int BS, int FP = 8, int NARROW = 0>
__global__ __launch_bounds__(512, 2) void scan16_kloop_kernel(ScanArgs a, ScanKloopExtra ex) {
    constexpr int NWAVES = 8, RING = 2 * BS, HT = 8, CB = 4;
    constexpr int PPS = kTilesPerSpan16 / HT;               // passes per span (8)
    static_assert(PPS % FP == 0, "bins must tile the span");
    constexpr int kStageVec = HT * 2 * 64;                  // 16-byte vectors per K-step stage (16 KiB)
    constexpr int kNarrowQVec = NARROW == 2 ? kNarrowMaxKS * 64 : 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem[(RING * kStageVec + kNarrowQVec) * 16];
    auto lds_a = [&](int buf) { return reinterpret_cast<half8 *>(smem + buf * (kStageVec * 16)); };
    half8 *lds_q = reinterpret_cast<half8 *>(smem + RING * kStageVec * 16);

    const int b = blockIdx.x;
    const int x = b & 7, j = b >> 3;
    const int cpx = (a.nchunks + 7) >> 3;                   // chunks per XCD label
    const int per_group = cpx * ex.qgroup;
    const int qg = j / per_group, rem = j - qg * per_group;
    const int ci = rem / ex.qgroup, qt = qg * ex.qgroup + (rem - ci * ex.qgroup);
    if (x + 8 * ci >= a.nchunks || qt >= a.nqtiles) return;
    const int chunk = a.chunk0 + x + 8 * ci;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;                                // lane group = bin of the span
    const int KS = ex.ksteps / 2, nK = KS / 2;              // 32-dim k-steps; 64-dim K-steps
    const int64_t q0 = (int64_t)qt * (NWAVES * 16 * CB) + wave * (16 * CB);
    const float cs = a.info->cs;

    const int64_t span0 = chunk_span0(chunk, a.spans_per_chunk, a.chunk_rem);
    int64_t span1 = chunk_span0(chunk + 1, a.spans_per_chunk, a.chunk_rem);
    if (span1 > a.nspans) span1 = a.nspans;
    const int npass = (int)(span1 - span0) * PPS;
    const int nsteps = npass * nK;

    const float INF = __builtin_inff();
    float NEG_INF = -INF;
    asm volatile("" : "+v"(NEG_INF));
    unsigned idmask = kQuadIdMask;
    asm volatile("" : "+v"(idmask));
    float m1[CB], m2[CB], M1[CB], M2[CB];
    int Ms[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        m1[cb] = m2[cb] = M1[cb] = M2[cb] = INF;
        Ms[cb] = 0;
    }

    const int lane16 = lane * 16;
    auto stage_issue = [&](int step, int buf) {             // A panels of K-step `step` -> LDS slot
        const int pass = step / nK, kk = step - pass * nK;
        const int64_t tile0 = (span0 + pass / PPS) * kTilesPerSpan16 + (pass % PPS) * HT;
        const half8 *base = ABL == 3 ? a.panels : a.panels + ((size_t)tile0 * KS + kk * 2) * 64;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half8 *>(base), 0, 0x7fffffff, 0x00020000);
        half8 *dst = lds_a(buf);
#pragma unroll
        for (int i = 0; i < (HT * 2) / NWAVES; ++i) {
            const int p = wave + i * NWAVES;                // piece = (tile t, k-step ks)
            const int t = p >> 1, ks = p & 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rs, reinterpret_cast<__attribute__((address_space(3))) void *>(
                        static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dst + p * 64))),
                16, lane16, ABL == 3 ? p * 1024 : (t * KS + ks) * 1024, 0, NARROW ? 2 : 0);   // (NARROW: one query tile, panels read once per search -> non-temporal)
        }
    };

    float4v acc[HT][CB];
    half8 bq[CB][2];          // B fragments of the current K-step, reloaded in place (see scan_kloop_kernel)
    const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half8 *>(a.qpanels + (size_t)(q0 / 16) * KS * 64), 0, 0x7fffffff, 0x00020000);
    auto load_b = [&](int kk, int ks) {
        if (NARROW == 2) {
            bq[0][ks] = lds_q[(kk * 2 + ks) * 64 + lane];
            return;
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
            bq[cb][ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(
                                                       rsq, lane16, (cb * KS + (ABL == 3 ? 0 : kk * 2) + ks) * 1024, 0));
    };
    if (NARROW == 2) {   // column block 0 of this query tile: KS contiguous 1-KiB fragments (every wave helps, wave 0 uses them)
        const half8 *src = a.qpanels + (size_t)((int64_t)qt * (NWAVES * CB)) * KS * 64;
        for (int i = tid; i < KS * 64; i += NWAVES * 64) lds_q[i] = src[i];
    }
#pragma unroll
    for (int i = 0; i < BS; ++i)
        if (i < nsteps) stage_issue(i, i);
    // A wave whose 64 query columns are all padding (small batches: 1 ... 448 queries leave 7 ... 1 such waves in the only
    // query tile) keeps its share of the A staging and the barriers going and nothing else: a single query then costs
    // one wave's matrix work per workgroup instead of eight, and the scan runs at the rate the panels stream in.
    if (a.nq_valid > 0 && q0 >= a.nq_valid) {
        __syncthreads();
        for (int step = 0; step < nsteps; ++step) {
            stage_issue(step + BS < nsteps ? step + BS : nsteps - 1, (step + BS) % RING);
            if (BS == 1 || (step % BS) == BS - 1) __syncthreads();
        }
        return;
    }
    // column blocks of this wave that hold queries (wave-uniform)
    const int ncb = !NARROW ? CB : NARROW == 2 ? 1 : __builtin_amdgcn_readfirstlane((int)((a.nq_valid - q0 + 15) / 16 < CB ? (a.nq_valid - q0 + 15) / 16 : CB));
    if (NARROW != 2) load_b(0, 0);
    __syncthreads();
    if (NARROW == 2) load_b(0, 0);      // (the query block in LDS is complete only behind the barrier)

    half8 fr[2][4];
    auto read_group = [&](int buf, int grp, half8(&dst)[4]) {   // group grp = (ks = grp>>1, tiles 4*(grp&1)..+3)
        const half8 *A = lds_a(buf);
#pragma unroll
        for (int t = 0; t < 4; ++t) dst[t] = A[(((grp & 1) * 4 + t) * 2 + (grp >> 1)) * 64 + lane];
    };
    read_group(0, 0, fr[0]);

    int step = 0;
    for (int pass = 0; pass < npass; ++pass) {
        const int64_t span = span0 + pass / PPS;
        const int slice = pass % PPS;
        // accumulators start from the bias of their rows: row = 1024*span + 256*g + 4*(HT*slice + t) + i
#pragma unroll
        for (int t = 0; t < (ABL == 4 ? 0 : HT); ++t) {
            const float4 c = *reinterpret_cast<const float4 *>(a.bias + span * kSpanRows16 + g * kBinRows +
                                                               (slice * HT + t) * 4);
            acc[t][0][0] = (c.x >= 0.9e38f) ? kPadBias : c.x * cs;
            acc[t][0][1] = (c.y >= 0.9e38f) ? kPadBias : c.y * cs;
            acc[t][0][2] = (c.z >= 0.9e38f) ? kPadBias : c.z * cs;
            acc[t][0][3] = (c.w >= 0.9e38f) ? kPadBias : c.w * cs;
#pragma unroll
            for (int cb = 1; cb < CB; ++cb) acc[t][cb] = acc[t][0];
        }
#pragma unroll 1   // (unrolled, hipcc hoists the next K-step's loads across the body and spills)
        for (int kk = 0; kk < nK; ++kk, ++step) {
            const int buf = step % RING;
            load_b(kk, 1);
            stage_issue(step + BS < nsteps ? step + BS : nsteps - 1, (step + BS) % RING);
            const int kn = (kk + 1 == nK) ? 0 : kk + 1;      // B repeats every pass
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                __builtin_amdgcn_sched_barrier(0);
                if (grp < 3) {
                    read_group(buf, grp + 1, fr[(grp + 1) & 1]);
                } else {
                    if (BS == 1 || (step % BS) == BS - 1) __syncthreads();
                    read_group((step + 1) % RING, 0, fr[0]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!NARROW) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb)
                            if (ABL != 2)
                                acc[(grp & 1) * 4 + t][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                    fr[grp & 1][t], bq[cb][grp >> 1], acc[(grp & 1) * 4 + t][cb], 0, 0, 0);
                } else {
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        if (cb < ncb) {
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                acc[(grp & 1) * 4 + t][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                    fr[grp & 1][t], bq[cb][grp >> 1], acc[(grp & 1) * 4 + t][cb], 0, 0, 0);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (grp == 1) load_b(kn, 0);
            }
        }
        if (ABL == 4) {      // keep the accumulators alive with one cheap use
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) m1[cb] = fast_min(m1[cb], acc[t][cb][0], NEG_INF);
        }
#pragma unroll
        for (int t = 0; t < (ABL == 4 ? 0 : HT); ++t) {
            const unsigned id = (unsigned)((slice % FP) * HT + t);   // quad number inside the level-1 bin
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                if (NARROW && cb >= ncb) continue;       // (its minima stay +inf: nothing reads those columns)
                const float qm = quad_min(acc[t][cb][0], acc[t][cb][1], acc[t][cb][2], acc[t][cb][3], NEG_INF);
                const float v = pack_score(qm, idmask, id);
                m2[cb] = __builtin_amdgcn_fmed3f(m1[cb], m2[cb], v);
                m1[cb] = fast_min(m1[cb], v, NEG_INF);
            }
        }
        if (slice % FP == FP - 1) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const size_t o = (size_t)((span * 4 + g) * (PPS / FP) + slice / FP) * a.Qpad + q0 + cb * 16 + (lane & 15);
                a.bin_m1[o] = m1[cb];
                a.bin_m2[o] = m2[cb];
                M2[cb] = __builtin_fminf(__builtin_amdgcn_fmed3f(M1[cb], M2[cb], m1[cb]), m2[cb]);
                if (m1[cb] < M1[cb]) Ms[cb] = (int)span;
                M1[cb] = __builtin_fminf(M1[cb], m1[cb]);
                m1[cb] = INF;
                m2[cb] = INF;
            }
        }
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const size_t so = (size_t)(chunk * 4 + g) * a.Qpad + q0 + cb * 16 + (lane & 15);
        a.sb_m1[so] = M1[cb];
        a.sb_m2[so] = M2[cb];
        a.sb_span[so] = Ms[cb];
    }
}

}  // namespace vdb
