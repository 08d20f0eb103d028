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
    float *bin_m3;         // ITEMS mode only: third-smallest quad minimum of each bin (the re-scan guard, see select_phase)
    float *sb_m1;          // [nchunks*2][Qpad] packed min of each superbin (chunk x lane half)
    float *sb_m2;          // [nchunks*2][Qpad] second-smallest score of the superbin
    int32_t *sb_span;      // [nchunks*2][Qpad] span holding the superbin minimum
    int64_t nspans;        // total spans (Npad / 512)
    int spans_per_chunk;   // base chunk length; the first `chunk_rem` chunks are one span longer
    int chunk_rem;
    int nchunks;
    int chunk0;            // scan16_kloop_kernel, streamed panels: this launch covers chunks [chunk0, chunk0 + nchunks) and
                           // `panels` is rebased so that the absolute tile numbers of those chunks land in the slab
    int nqtiles;           // query tiles (64 * NWAVES queries each)
    int64_t Qpad;          // multiple of 512
    int64_t nq_valid;      // flat mode: queries >= nq_valid are padding (0 = unknown: treat every column as real)
    // ---- IVF ("items") mode: one workgroup = (one inverted list) x (one group of 64*NWAVES query slots) ----
    const int32_t *item_list;    // [items] list id
    const int32_t *item_slot0;   // [items] first query slot of the group
    const int32_t *item_bin0;    // [items] first level-1 bin of the item's output block
    const int32_t *n_items;      // [1]
    const int32_t *list_pspan0;  // [nlist+1] panel spans of every list (lists are padded to whole spans)
    int part_spans;              // items mode: spans per row part (blockIdx.y = part of the list this workgroup scans; 0 = whole list)
    const int32_t *slot_query;   // [slots] query + 1 of every slot (0 = padding)
    const _Float16 *qrows;       // [nq][16*KSTEPS] scaled fp16 query rows (B fragments are gathered from them)
    int prio;                    // x16 kernels (option "scan_prio", tuning): 1 = the late half of a workgroup issues at priority 1, 2 = the early half
    unsigned long long *dbg;     // ABL == 4 (diagnostic build): per-wave cycle sums {head, mfma, select, barrier, total, late}
};

// (score & ~mask) | id  -- one v_and_or_b32 when the mask lives in a VGPR (the id is wave-uniform)
__device__ __forceinline__ float pack_score(float v, unsigned mask, unsigned id) {
    return __uint_as_float((__float_as_uint(v) & mask) | id);
}
// min without the sNaN-quieting v_max that fminf() drags in: med3(a, b, -inf) == min(a, b)
__device__ __forceinline__ float fast_min(float a, float b, float neg_inf) {
    return __builtin_amdgcn_fmed3f(a, b, neg_inf);
}

// Minimum of the four (eight) scores of a quad (oct), straight out of the accumulators, as a LEFT-NESTED fminf chain of odd
// length: hipcc folds fminf(fminf(x, y), z) into one v_min3_f32 on raw operands, but puts a quieting v_max in front of each
// operand of a two-operand v_min_f32 (ISA of round 4's first attempt: 2 v_max + v_min + v_min3 per quad -- worse than the three
// v_med3 it replaced).  The opaque +inf (-neg_inf, a source modifier) pads the chain to an odd count: 2 ops per quad, 4 per oct.
__device__ __forceinline__ float quad_min(float a, float b, float c, float d, float neg_inf) {
    return __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fminf(a, b), c), d), -neg_inf);
}
__device__ __forceinline__ float oct_min(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7,
                                         float neg_inf) {
    float t = __builtin_fminf(__builtin_fminf(a0, a1), a2);
    t = __builtin_fminf(__builtin_fminf(t, a3), a4);
    t = __builtin_fminf(__builtin_fminf(t, a5), a6);
    return __builtin_fminf(__builtin_fminf(t, a7), -neg_inf);
}

// ---- phases of one 32-row tile, for one wave (sched_barrier(0) keeps hipcc from blending them) ----------
// a packed bin minimum names a quad: offset of its first row inside the bin
// chunk c covers spans [chunk_span0(c), chunk_span0(c+1)): nspans are dealt as evenly as possible so that the
// workgroup count is a multiple of 8 x (query tiles) and every XCD gets the same amount of work
__host__ __device__ inline int64_t chunk_span0(int chunk, int spc, int rem) {
    return (int64_t)chunk * spc + (chunk < rem ? chunk : rem);
}

__host__ __device__ inline int quad_row_offset(unsigned packed_bits) {
    const unsigned id = packed_bits & 0x3Fu;
    return (int)(((id >> 2) << 4) | ((id & 3u) << 2));
}
constexpr unsigned kQuadIdMask = 0xFFFFFFC0u;
constexpr int kQuadRows = 4;

// in-kernel stamp for the diagnostic build (cdna guide, 'In-kernel stamps'): shader-cycle counter, with the
// lgkmcnt(0) the s_memtime result needs inside the same statement, fenced against the scheduler
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

template <int KSTEPS, int ABL>
__device__ __forceinline__ void mfma_phase(const half8 (&fr)[KSTEPS], const half8 (&b0)[KSTEPS],
                                           const half8 (&b1)[KSTEPS], const float16v &cin, float16v &acc0,
                                           float16v &acc1) {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        if (ABL != 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[ks], b0[ks], ks == 0 ? cin : acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[ks], b1[ks], ks == 0 ? cin : acc1, 0, 0, 0);
        } else {
            if (ks == 0) { acc0 = cin; acc1 = cin; }
            acc0[ks] += (float)fr[ks][0] * (float)b0[ks][0];
            acc1[ks] += (float)fr[ks][1] * (float)b1[ks][1];
        }
    }
}

// Select phase of one tile.  The 16 scores a lane holds per column block are 4 QUADS of 4 consecutive corpus
// rows (registers 4g..4g+3 <-> rows 16t+4g..16t+4g+3 of the bin).  Only the quad minimum enters the running
// (min, second min) of the bin, tagged with the 6-bit quad id (tile-in-bin*4 + g) in its low mantissa bits:
// 3 + 1 + 2 = 6 VALU ops per 4 scores (1.5 per score instead of 3), which takes the VALU off the critical path
// of the matrix pipe.  A candidate is then a quad (4 rows, re-scored exactly), and the bin's second minimum is
// the second-smallest QUAD minimum: two close scores inside one quad need no re-scan at all.
// M3 (IVF items mode): also keep the THIRD-smallest quad minimum m3 (one more med3 per quad).  Both m1 and m2 carry
// their quad ids, so a bin whose third minimum is above the threshold yields two candidate quads instead of a re-scan
// of the whole bin: the probed lists of a query are dense in near neighbours and "two close quads in one bin" is
// common there (0.6-0.9 re-scanned bins per query with the two-minimum guard, a few per hundred with three).
// OCT (flat index, round 4): the group is an OCT -- 8 consecutive rows, two adjacent quads of the lane's 16 -- as on the int8
// scan: 4 min ops (v_min3 chains on the raw accumulators) + 1 + 2 = 7 VALU ops per 8 scores instead of 5 per 4.  The select is
// what the D <= 64 scan is bound by (stamps, profiles/r04_stamps_scan_fp16.txt: 441 - 472 cycles of select against 256 of MFMA
// per tile and wave) and a sixth of the D = 128 scan; the price is twice the rows per candidate in the exact refine.
template <int ABL, bool M3 = false, bool OCT = false>
__device__ __forceinline__ void select_phase(const float16v &acc0, const float16v &acc1, float (&m1)[2],
                                             float (&m2)[2], unsigned idmask, float neg_inf, unsigned id0,
                                             float *m3 = nullptr) {
    if (ABL == 1) {
        asm volatile("" ::"v"(acc0), "v"(acc1));
        return;
    }
    if (OCT) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float o0 = oct_min(acc0[8 * g], acc0[8 * g + 1], acc0[8 * g + 2], acc0[8 * g + 3], acc0[8 * g + 4],
                                     acc0[8 * g + 5], acc0[8 * g + 6], acc0[8 * g + 7], neg_inf);
            const float v0 = pack_score(o0, idmask, (id0 >> 1) + g);
            m2[0] = __builtin_amdgcn_fmed3f(m1[0], m2[0], v0);
            m1[0] = fast_min(m1[0], v0, neg_inf);
            const float o1 = oct_min(acc1[8 * g], acc1[8 * g + 1], acc1[8 * g + 2], acc1[8 * g + 3], acc1[8 * g + 4],
                                     acc1[8 * g + 5], acc1[8 * g + 6], acc1[8 * g + 7], neg_inf);
            const float v1 = pack_score(o1, idmask, (id0 >> 1) + g);
            m2[1] = __builtin_amdgcn_fmed3f(m1[1], m2[1], v1);
            m1[1] = fast_min(m1[1], v1, neg_inf);
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < (ABL == 5 ? 2 : 4); ++g) {   // ABL 5: timing-only, half the select work
        const float q0 = quad_min(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3], neg_inf);
        const float v0 = pack_score(q0, idmask, id0 + g);
        if (M3) m3[0] = __builtin_amdgcn_fmed3f(m2[0], m3[0], v0);
        m2[0] = __builtin_amdgcn_fmed3f(m1[0], m2[0], v0);
        m1[0] = fast_min(m1[0], v0, neg_inf);
        const float q1 = quad_min(acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3], neg_inf);
        const float v1 = pack_score(q1, idmask, id0 + g);
        if (M3) m3[1] = __builtin_amdgcn_fmed3f(m2[1], m3[1], v1);
        m2[1] = __builtin_amdgcn_fmed3f(m1[1], m2[1], v1);
        m1[1] = fast_min(m1[1], v1, neg_inf);
    }
}

template <int KSTEPS>
__device__ __forceinline__ void read_phase(const half8 *__restrict__ A_tile, const float4 *__restrict__ c_tile,
                                           half8 (&fr)[KSTEPS], float16v &cin, int lane) {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) fr[ks] = A_tile[ks * 64 + lane];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 c = c_tile[g];
        cin[4 * g + 0] = c.x; cin[4 * g + 1] = c.y; cin[4 * g + 2] = c.z; cin[4 * g + 3] = c.w;
    }
}

// KSTEPS: 16-dim MFMA k-steps (D padded to 16*KSTEPS); NWAVES: waves per workgroup (64 queries each);
// ST: tiles per LDS stage (divides 16); WPS: waves per SIMD the register allocator must leave room for.
// ABL: timing-only ablations (wrong results!): 1 = no select epilogue, 2 = no MFMA, 3 = no global loads
//
// Phase stagger: a tile costs a wave one MFMA phase (2*KSTEPS back-to-back MFMAs, matrix pipe) and one
// select phase (96 VALU ops).  The two waves that share a SIMD (wave w and w + NWAVES/2) run them in
// opposite order -- the "early" half does MFMA(t) then select(t), the "late" half select(t-1) then MFMA(t) --
// so that while one wave owns the matrix pipe its partner is on the VALU, instead of both queueing for the
// same pipe (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).  Both orders do the same work per stage,
// so the per-stage barrier keeps the stagger locked.
// BT: tiles per level-1 bin per lane half (16 -> 256-row bins for the flat index, 4 -> 64-row bins for IVF,
// whose per-query row count is small); ITEMS: IVF work-item mode (see ScanArgs).
template <int KSTEPS, int NWAVES, int ST, int WPS, int ABL = 0, int BT = 16, bool ITEMS = false, int PRIO = 0, bool G8 = false>
__global__ __launch_bounds__(NWAVES * 64, WPS) void scan_kernel(ScanArgs a) {
    static_assert(!G8 || (!ITEMS && BT == 16), "octs: flat index, 256-row bins");
    constexpr int NT = NWAVES * 64;
    constexpr int kStageVec = ST * KSTEPS * 64;           // 16-byte vectors per stage
    constexpr int kBiasLoads = (ST * 32 + NT - 1) / NT;
    constexpr int TPS = ITEMS ? kIvfTilesPerSpan : kTilesPerSpan;   // tiles per span (IVF panel space: smaller spans, common.hpp)
    constexpr int kSpanR = TPS * 32, kHalfR = TPS * 16;   // rows per span / per lane half of a span
    constexpr int SPS = TPS / ST;                         // stages per span
    constexpr int BPS = TPS / BT;                         // level-1 bins per (span, lane half)
    static_assert(TPS % ST == 0 && ST >= 2 && TPS % BT == 0 && BT % ST == 0, "bad geometry");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (kStageVec * 16 + ST * 32 * 4)];
    auto lds_a = [&](int buf) { return reinterpret_cast<half8 *>(smem + buf * (kStageVec * 16)); };
    auto lds_b = [&](int buf) { return reinterpret_cast<float *>(smem + 2 * kStageVec * 16 + buf * (ST * 32 * 4)); };
    if (a.info->i8_mode) return;                            // this batch is served by scan_i8_kernel (scan_i8.hpp)

    // ---- block -> (chunk, query tile), XCD aware; or -> IVF work item -----------------------------
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const bool late = (PRIO == 3) || ((NWAVES >= 2) && (wave >= NWAVES / 2));
    int chunk = 0;
    int64_t q0, span0, span1, out_pitch, out_col;
    int64_t lspan0 = 0, lspans = 0;                         // items mode: first span / span count of the whole list (bin indexing)
    size_t bin_base = 0;                                    // first level-1 bin of this block's output
    if (ITEMS) {
        const int it = blockIdx.x;
        if (it >= *a.n_items) return;
        const int l = a.item_list[it];
        span0 = a.list_pspan0[l];
        span1 = a.list_pspan0[l + 1];
        lspan0 = span0;
        lspans = span1 - span0;
        if (a.part_spans > 0) {      // long lists are cut into row parts, one workgroup each: same bins, shorter critical path
            span0 += (int64_t)blockIdx.y * a.part_spans;
            if (span0 >= span1) return;
            if (span0 + a.part_spans < span1) span1 = span0 + a.part_spans;
        }
        q0 = (int64_t)a.item_slot0[it] + wave * 64;        // "queries" are gathered query slots
        out_pitch = NWAVES * 64;
        out_col = wave * 64 + (lane & 31);
        bin_base = (size_t)a.item_bin0[it];
    } else {
        const int b = blockIdx.x;
        const int x = b & 7, j = b >> 3;
        const int ci = j / a.nqtiles, qt = j - ci * a.nqtiles;
        chunk = x + 8 * ci;
        if (chunk >= a.nchunks) return;
        chunk += a.chunk0;                                  // (a slab launch of an int8-only index: chunks [chunk0, chunk0 + nchunks))
        q0 = (int64_t)qt * (NWAVES * 64) + wave * 64;       // first query of this wave
        span0 = chunk_span0(chunk, a.spans_per_chunk, a.chunk_rem);
        span1 = chunk_span0(chunk + 1, a.spans_per_chunk, a.chunk_rem);
        if (span1 > a.nspans) span1 = a.nspans;
        out_pitch = a.Qpad;
        out_col = q0 + (lane & 31);
    }
    const float cs = a.info->cs;

    // ---- B fragments: resident for the whole chunk ------------------------------------------------
    half8 b0[KSTEPS], b1[KSTEPS];
    int qa = -1, qb = -1;
    if (ITEMS) {  // the slots' queries now, their rows AFTER the first stage is on its way (gather_b below): a work item
                  // is short, so every dependent round trip in front of its first MFMA counts
        qa = a.slot_query[q0 + (lane & 31)] - 1;
        qb = a.slot_query[q0 + 32 + (lane & 31)] - 1;
    } else {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            b0[ks] = a.qpanels[((size_t)(q0 / 32 + 0) * KSTEPS + ks) * 64 + lane];
            b1[ks] = a.qpanels[((size_t)(q0 / 32 + 1) * KSTEPS + ks) * 64 + lane];
        }
    }

    const int nstages = (int)(span1 - span0) * SPS;
    const int nb_item = 2 * ((((int)lspans * BPS) + 3) & ~3);    // ITEMS: bins per query slot in this item (two half-runs)

    const float INF = __builtin_inff();
    float NEG_INF = -INF;
    asm volatile("" : "+v"(NEG_INF));  // opaque, or LLVM folds med3(a,b,-inf) back into a canonicalising fmin
    unsigned idmask = kQuadIdMask;
    asm volatile("" : "+v"(idmask));   // pin the mask in a VGPR (a literal cannot ride in VOP3 next to an SGPR id)
    float m1[2] = {INF, INF}, m2[2] = {INF, INF};   // level 1 (current bin)
    float m3[2] = {INF, INF};                       // ITEMS mode: third minimum
    float M1[2] = {INF, INF}, M2[2] = {INF, INF};   // level 2 (whole chunk)
    int Ms[2] = {0, 0};

    // ---- staging: panels go global -> LDS by DMA (global_load_lds, 1 KiB per wave instruction: the panel
    //      layout is already lane-linear), the 32 bias floats per tile through registers (scaled by cs) -------
    constexpr int kPieces = kStageVec / 64;            // 1-KiB pieces per stage
    static_assert(kPieces % NWAVES == 0, "pieces must divide over the waves");
    float stage_b[kBiasLoads];
    auto stage_issue = [&](int st, int buf) {
        const int64_t span = span0 + st / SPS;
        const int sq = st % SPS;
        const half8 *src = a.panels + ((size_t)(span * TPS + sq * ST) * KSTEPS) * 64;
        half8 *dst = lds_a(buf);
#pragma unroll
        for (int i = 0; i < kPieces / NWAVES; ++i) {
            const int p = wave + i * NWAVES;
            const half8 *g = src + (ABL == 3 ? (p & 15) : p) * 64 + lane;
            __builtin_amdgcn_global_load_lds(
                reinterpret_cast<const __attribute__((address_space(1))) void *>(reinterpret_cast<uintptr_t>(g)),
                reinterpret_cast<__attribute__((address_space(3))) void *>(
                    static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dst + p * 64))),
                16, 0, (NWAVES <= 2 && !ITEMS) ? 2 : 0);     // (1- / 2-wave workgroups = serving shapes, one query tile: every panel byte
                                                              //  is read once per search -> non-temporal, as scan_i8_kernel's AUX)
        }
#pragma unroll
        for (int i = 0; i < kBiasLoads; ++i) {
            const int e = tid + i * NT;
            if (e < ST * 32) {
                const int t = e >> 5, hh = (e >> 4) & 1, r = e & 15;
                // raw value only: touching it here would make hipcc drain vmcnt(0), i.e. wait for the DMA
                stage_b[i] = a.bias[span * kSpanR + hh * kHalfR + (sq * ST + t) * 16 + r];
            }
        }
    };
    auto stage_bias_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < kBiasLoads; ++i)
            if (tid + i * NT < ST * 32)
                lds_b(buf)[tid + i * NT] = (stage_b[i] >= 0.9e38f) ? kPadBias : stage_b[i] * cs;
    };
    // ITEMS: the bins a lane produces for its query slot form ONE run per lane half -- [item][slot][half][span][bin of the
    // half], nb_half floats per half (rounded up to 4) -- so they are collected in registers and leave as ONE 16-byte store
    // per array every fourth bin, whatever the bin size.  (In [item][slot][bin] order every lane writes its own line: with a
    // 4-byte store per bin and array those stores kept the CU's address path busy most of the time -- the list scan at
    // nprobe 8 ran as long as at nprobe 32 -- and with [span][half][bin] order only the bins of one span half were adjacent.)
    // Row parts start at multiples of 4 bins (the host sizes them so); the last vector of a part is filled with +inf, which
    // lands on the padding behind the half's last bin.
    float4 pend1[2], pend2[2], pend3[2];
    const int nb_half = ITEMS ? (((int)lspans * BPS + 3) & ~3) : 0;
    auto items_push = [&](int r) {      // bin r of this lane's half-run is complete (its minima are in m1 / m2 / m3)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            pend1[cb] = make_float4(pend1[cb].y, pend1[cb].z, pend1[cb].w, m1[cb]);
            pend2[cb] = make_float4(pend2[cb].y, pend2[cb].z, pend2[cb].w, m2[cb]);
            pend3[cb] = make_float4(pend3[cb].y, pend3[cb].z, pend3[cb].w, m3[cb]);
            m1[cb] = INF;
            m2[cb] = INF;
            m3[cb] = INF;
        }
        if ((r & 3) == 3) {
            const size_t o4 = bin_base * out_pitch + (size_t)out_col * nb_item + (size_t)(h * nb_half + r - 3);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                *reinterpret_cast<float4 *>(a.bin_m1 + o4 + (size_t)cb * 32 * nb_item) = pend1[cb];
                *reinterpret_cast<float4 *>(a.bin_m2 + o4 + (size_t)cb * 32 * nb_item) = pend2[cb];
                *reinterpret_cast<float4 *>(a.bin_m3 + o4 + (size_t)cb * 32 * nb_item) = pend3[cb];
            }
        }
    };
    // a level-1 bin (BT tiles per lane half) is complete: write (min, second min), fold level 2
    auto flush_bin = [&](int64_t span, int bt) {
        if (ITEMS) {
            items_push((int)(span - lspan0) * BPS + bt);
            return;
        }
        // flat: [bin][query] (coalesced over the 32 queries of a lane half)
        const size_t o = (size_t)((span * 2 + h) * BPS + bt) * out_pitch + out_col;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            __builtin_nontemporal_store(m1[cb], a.bin_m1 + o + cb * 32);     // (written once, read sparsely by the select: they
            __builtin_nontemporal_store(m2[cb], a.bin_m2 + o + cb * 32);     //  should not push the panels out of L2 -- scan_i8.hpp)
            M2[cb] = __builtin_fminf(__builtin_amdgcn_fmed3f(M1[cb], M2[cb], m1[cb]), m2[cb]);
            if (m1[cb] < M1[cb]) Ms[cb] = (int)span;
            M1[cb] = __builtin_fminf(M1[cb], m1[cb]);
            m1[cb] = INF;
            m2[cb] = INF;
        }
    };

    stage_issue(0, 0);
    if (ITEMS) {  // gather: lane (col = lane&31, k half = lane>>5) reads 16 bytes of its slot's query row
        const half8 *ra = reinterpret_cast<const half8 *>(a.qrows + (size_t)(qa < 0 ? 0 : qa) * (16 * KSTEPS)) + h;
        const half8 *rb = reinterpret_cast<const half8 *>(a.qrows + (size_t)(qb < 0 ? 0 : qb) * (16 * KSTEPS)) + h;
        half8 zero;
#pragma unroll
        for (int j = 0; j < 8; ++j) zero[j] = (_Float16)0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            b0[ks] = qa < 0 ? zero : ra[ks * 2];
            b1[ks] = qb < 0 ? zero : rb[ks * 2];
        }
    }
    stage_bias_store(0);
    __syncthreads();  // (drains the DMA: hipcc waits vmcnt(0) in front of the barrier)

    // A wave whose 64 query columns are all padding (the last query tile of a batch that is not a multiple of 512)
    // only keeps staging and the barriers going: its partner on the SIMD gets the matrix pipe to itself.
    // Items mode: the same for a wave whose 64 slots are all padding (the last item of a list) -- with it a large slot
    // group costs traffic-free padding only, so the lists are streamed fewer times (ivf.inc, waves per work item).
    bool idle_wave = !ITEMS && a.nq_valid > 0 && q0 >= a.nq_valid;
    if (ITEMS) idle_wave = __ballot(qa >= 0 || qb >= 0) == 0ull;
    if (idle_wave) {
        for (int st = 0; st < nstages; ++st) {
            if (st + 1 < nstages) {
                stage_issue(st + 1, (st & 1) ^ 1);
                stage_bias_store((st & 1) ^ 1);
            }
            __syncthreads();
        }
        return;
    }

    half8 fr[KSTEPS];
    float16v cin, acc0, acc1;
    // static issue priority for one half of the SIMD partners keeps the phase stagger from collapsing into
    // lockstep (equal-priority waves share the matrix pipe evenly, finish together and then both sit on the VALU)
    if (PRIO == 1 && late) __builtin_amdgcn_s_setprio(1);
    if (PRIO == 2 && !late) __builtin_amdgcn_s_setprio(1);

    unsigned long long c_head = 0, c_mfma = 0, c_sel = 0, c_bar = 0, t_start = 0, ta = 0, tb = 0;
    if (ABL == 4) t_start = stamp();
    if (!late) {
        // ================= early half: MFMA(t), then select(t) =======================================
        for (int st = 0; st < nstages; ++st) {
            const int buf = st & 1;
            if (ABL == 4) ta = stamp();
            if (st + 1 < nstages) stage_issue(st + 1, buf ^ 1);  // buf^1 was last read before the previous barrier
            const half8 *A = lds_a(buf);
            const float4 *B4 = reinterpret_cast<const float4 *>(lds_b(buf)) + h * 4;
            const int ts0 = (st % SPS) * ST;                 // first tile of this stage inside its span
            read_phase<KSTEPS>(A, B4, fr, cin, lane);
            if (ABL == 4) { tb = stamp(); c_head += tb - ta; ta = tb; }
#pragma unroll
            for (int t = 0; t < ST; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                mfma_phase<KSTEPS, ABL>(fr, b0, b1, cin, acc0, acc1);
                __builtin_amdgcn_sched_barrier(0);
                if (ABL == 4) { asm volatile("s_nop 0" ::"v"(acc0), "v"(acc1)); tb = stamp(); c_mfma += tb - ta; ta = tb; }
                if (t + 1 < ST) read_phase<KSTEPS>(A + (t + 1) * KSTEPS * 64, B4 + (t + 1) * 8, fr, cin, lane);
                select_phase<ABL, ITEMS, G8>(acc0, acc1, m1, m2, idmask, NEG_INF, (unsigned)(((ts0 + t) % BT) << 2), m3);
                if (ABL == 4) { tb = stamp(); c_sel += tb - ta; ta = tb; }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (((ts0 + ST) % BT) == 0) flush_bin(span0 + st / SPS, (ts0 + ST) / BT - 1);  // BT % ST == 0
            if (st + 1 < nstages) stage_bias_store(buf ^ 1);
            if (ABL == 4) { tb = stamp(); c_sel += tb - ta; ta = tb; }
            __syncthreads();
            if (ABL == 4) { tb = stamp(); c_bar += tb - ta; }
        }
    } else {
        // ================= late half: select(t-1), then MFMA(t) =====================================
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = 3.0e38f;  // dummy "previous tile": can never win
        acc1 = acc0;
        for (int st = 0; st < nstages; ++st) {
            const int buf = st & 1;
            if (ABL == 4) ta = stamp();
            if (st + 1 < nstages) stage_issue(st + 1, buf ^ 1);
            const half8 *A = lds_a(buf);
            const float4 *B4 = reinterpret_cast<const float4 *>(lds_b(buf)) + h * 4;
            const int ts0 = (st % SPS) * ST;
            if (ABL == 4) { tb = stamp(); c_head += tb - ta; ta = tb; }
#pragma unroll
            for (int t = 0; t < ST; ++t) {
                if (PRIO != 3) __builtin_amdgcn_sched_barrier(0);
                read_phase<KSTEPS>(A + t * KSTEPS * 64, B4 + t * 8, fr, cin, lane);
                // retire the previous tile: index tp inside its span (the span before this one when ts0 + t == 0)
                const int tp = (ts0 + t + TPS - 1) % TPS;
                select_phase<ABL, ITEMS, G8>(acc0, acc1, m1, m2, idmask, NEG_INF, (unsigned)((tp % BT) << 2), m3);
                if (t == 0 && st > 0 && (ts0 % BT) == 0)      // (BT % ST == 0: bins only end at stage starts)
                    flush_bin(span0 + (st * ST - 1) / TPS, tp / BT);
                if (PRIO != 3) __builtin_amdgcn_sched_barrier(0);
                if (ABL == 4) { tb = stamp(); c_sel += tb - ta; ta = tb; }
                mfma_phase<KSTEPS, ABL>(fr, b0, b1, cin, acc0, acc1);
                if (ABL == 4) { asm volatile("s_nop 0" ::"v"(acc0), "v"(acc1)); tb = stamp(); c_mfma += tb - ta; ta = tb; }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 < nstages) stage_bias_store(buf ^ 1);
            if (ABL == 4) { tb = stamp(); c_sel += tb - ta; ta = tb; }
            __syncthreads();
            if (ABL == 4) { tb = stamp(); c_bar += tb - ta; }
        }
        select_phase<ABL, ITEMS, G8>(acc0, acc1, m1, m2, idmask, NEG_INF, (unsigned)((BT - 1) << 2), m3);  // drain the last tile
        flush_bin(span1 - 1, BPS - 1);
    }
    if (ITEMS)       // fill the last vector of this part's run
        for (int r = (int)(span1 - lspan0) * BPS; r & 3; ++r) items_push(r);
    if (ABL == 4 && a.dbg && lane == 0) {
        const unsigned long long t_end = stamp();
        unsigned long long *d = a.dbg + ((size_t)blockIdx.x * NWAVES + wave) * 8;
        d[0] = c_head; d[1] = c_mfma; d[2] = c_sel; d[3] = c_bar; d[4] = t_end - t_start; d[5] = late ? 1 : 0;
        d[6] = (unsigned long long)nstages;
    }
    if (ITEMS) return;

    const size_t so = (size_t)(chunk * 2 + h) * a.Qpad + q0 + (lane & 31);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        a.sb_m1[so + cb * 32] = M1[cb];
        a.sb_m2[so + cb * 32] = M2[cb];
        a.sb_span[so + cb * 32] = Ms[cb];
    }
}

// ---- K-loop scan for D > 128 (e.g. 384 / 768-dim embeddings) -------------------------------------------
// Same inputs, outputs and bin semantics as scan_kernel, but the query fragments no longer fit in registers,
// so the contraction is tiled like a GEMM: workgroup = 8 waves = (corpus chunk) x (256*CB-query tile); wave w
// owns CB 32-query column blocks and HT row tiles of the current span slice: HT x CB x 16 accumulator registers
// (HT * CB = 8).  Per 64-dim K-step the slice's A panels (HT x 4 KiB) arrive in LDS by DMA (double buffered)
// and are shared by the 8 waves; each wave streams its own B fragments (CB x 4 x 1 KiB per K-step) straight
// from L2 into registers one K-step ahead.  After the K loop the scores go through the same 3-op quad select;
// 16/HT passes complete a 256-row bin per lane half.
//   HT=8, CB=1: one ds_read_b128 (1 KiB) per MFMA -- the four SIMDs then ask the LDS for its whole 128 B/clk.
//   HT=4, CB=2: every A fragment feeds two MFMAs (LDS traffic halved); B is re-read every 128 rows instead of
//               every 256, which is why the block order below keeps few query tiles live per XCD (B stays in L2).
// Block order (QG = query tiles per group): within an XCD consecutive blocks walk (query group, chunk, tile in
// group), so concurrently resident workgroups share QG B panels and each chunk stream is shared by QG blocks.
struct ScanKloopExtra {
    int ksteps;  // 16-dim k-steps, multiple of 4
    int qgroup;  // query tiles per group (>= 1)
};

// BS: K-steps between workgroup barriers.  The LDS holds a ring of 2*BS stages; K-step s+BS is requested during
// K-step s into the slot that K-step s-BS used, which every wave left before the barrier that closed its group.
// NW: waves per workgroup (8: one workgroup per CU; 4: two independent workgroups per CU, each with its own barrier).
// ABL: timing-only builds -- 2 no MFMA, 3 no L2 traffic, 4 no pass epilogue / bias init.
template <int ABL, int HT, int CB, int BS, bool SB = true, int NW = 8>
__global__ __launch_bounds__(NW * 64, 2) void scan_kloop_kernel(ScanArgs a, ScanKloopExtra ex) {
    constexpr int NWAVES = NW, RING = 2 * BS;
    constexpr int PPS = kTilesPerSpan / HT;                 // passes per span
    constexpr int kStageVec = HT * 4 * 64;                  // 16-byte vectors per K-step stage
    static_assert(HT * CB == 8 && (HT * 4) % NWAVES == 0, "bad blocking");
    __shared__ __attribute__((aligned(16))) unsigned char smem[RING * kStageVec * 16];
    auto lds_a = [&](int buf) { return reinterpret_cast<half8 *>(smem + buf * (kStageVec * 16)); };

    const int b = blockIdx.x;
    const int x = b & 7, j = b >> 3;
    const int cpx = (a.nchunks + 7) >> 3;                   // chunks per XCD label
    const int per_group = cpx * ex.qgroup;
    const int qg = j / per_group, rem = j - qg * per_group;
    const int ci = rem / ex.qgroup, qt = qg * ex.qgroup + (rem - ci * ex.qgroup);
    const int chunk = x + 8 * ci;
    if (chunk >= a.nchunks || qt >= a.nqtiles) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int KS = ex.ksteps, nK = KS / 4;
    const int64_t q0 = (int64_t)qt * (NWAVES * 32 * CB) + wave * (32 * CB);
    const float cs = a.info->cs;

    const int64_t span0 = chunk_span0(chunk, a.spans_per_chunk, a.chunk_rem);
    int64_t span1 = chunk_span0(chunk + 1, a.spans_per_chunk, a.chunk_rem);
    if (span1 > a.nspans) span1 = a.nspans;
    const int npass = (int)(span1 - span0) * PPS;
    const int nsteps = npass * nK;                          // K-steps over the whole chunk

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

    // Global requests go through buffer descriptors: uniform 64-bit base in SGPRs + one 32-bit lane offset, so
    // the inner loop carries no per-lane 64-bit addresses (they cost ~30 VGPRs and pushed hipcc into scratch).
    const int lane16 = lane * 16;
    auto stage_issue = [&](int step, int buf) {             // A panels of K-step `step` -> LDS buffer
        const int pass = step / nK, kk = step - pass * nK;
        const int64_t tile0 = (span0 + pass / PPS) * kTilesPerSpan + (pass % PPS) * HT;
        // ABL 3 (timing only): every request falls on the same few KiB, i.e. no L2 / fabric traffic to speak of
        const half8 *base = ABL == 3 ? a.panels : a.panels + ((size_t)tile0 * KS + kk * 4) * 64;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half8 *>(base), 0, 0x7fffffff, 0x00020000);
        half8 *dst = lds_a(buf);
#pragma unroll
        for (int i = 0; i < (HT * 4) / NWAVES; ++i) {
            const int p = wave + i * NWAVES;                // piece = (tile t, k-step ks)
            const int t = p >> 2, ks = p & 3;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rs, reinterpret_cast<__attribute__((address_space(3))) void *>(
                        static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dst + p * 64))),
                16, lane16, ABL == 3 ? p * 1024 : (t * KS + ks) * 1024, 0, 0);
        }
    };

    float16v acc[HT][CB];
    // B fragments of the current K-step, reloaded in place: k-steps 0,1 of the NEXT K-step as soon as this one
    // has used them, k-steps 2,3 at the top of their own K-step (16 / 24 MFMAs ahead of their first use), so no
    // second register set is needed and nothing freshly requested is in flight at the barrier's vmcnt(0)
    half8 bq[CB][4];
    const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half8 *>(a.qpanels + (size_t)(q0 / 32) * KS * 64), 0, 0x7fffffff, 0x00020000);
    auto load_b = [&](int kk, int ks) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
            bq[cb][ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(
                                                       rsq, lane16, (cb * KS + (ABL == 3 ? 0 : kk * 4) + ks) * 1024, 0));
    };
#pragma unroll
    for (int i = 0; i < BS; ++i)
        if (i < nsteps) stage_issue(i, i);
    load_b(0, 0);
    load_b(0, 1);
    __syncthreads();

    // A fragments are software pipelined one k-step (HT ds_read_b128, HT*CB MFMAs) ahead in two register sets,
    // with sched_barrier(0) pinning "request the next group, then issue this group's MFMAs" -- left to itself
    // hipcc requests a fragment two MFMAs before its use and every group then waits out the LDS round trip.
    half8 fr[2][HT];
    auto read_group = [&](int buf, int ks, half8(&dst)[HT]) {
        const half8 *A = lds_a(buf);
#pragma unroll
        for (int t = 0; t < HT; ++t) dst[t] = A[(t * 4 + ks) * 64 + lane];
    };
    read_group(0, 0, fr[0]);

    int step = 0;
    for (int pass = 0; pass < npass; ++pass) {
        const int64_t span = span0 + pass / PPS;
        const int slice = pass % PPS;
        // accumulators start from the bias of their rows: row = 512*span + 256*h + 16*(HT*slice + t) + r
#pragma unroll
        for (int t = 0; t < (ABL == 4 ? 0 : HT); ++t) {
            const float4 *bp = reinterpret_cast<const float4 *>(a.bias + span * kSpanRows + h * kBinRows +
                                                                (slice * HT + t) * 16);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 c = bp[g];
                acc[t][0][4 * g + 0] = (c.x >= 0.9e38f) ? kPadBias : c.x * cs;
                acc[t][0][4 * g + 1] = (c.y >= 0.9e38f) ? kPadBias : c.y * cs;
                acc[t][0][4 * g + 2] = (c.z >= 0.9e38f) ? kPadBias : c.z * cs;
                acc[t][0][4 * g + 3] = (c.w >= 0.9e38f) ? kPadBias : c.w * cs;
            }
#pragma unroll
            for (int cb = 1; cb < CB; ++cb) acc[t][cb] = acc[t][0];
        }
#pragma unroll 1   // (unrolled, hipcc hoists the next K-step's loads across the body and spills ~80 VGPRs)
        for (int kk = 0; kk < nK; ++kk, ++step) {
            const int buf = step % RING;
            load_b(kk, 2);
            load_b(kk, 3);
            // (past the end of the chunk the last stage is simply requested again into a slot nobody reads, the
            //  stale fragments read below are never used: the inner loop stays free of data-dependent branches)
            stage_issue(step + BS < nsteps ? step + BS : nsteps - 1, (step + BS) % RING);
            const int kn = (kk + 1 == nK) ? 0 : kk + 1;      // B repeats every pass
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (SB) __builtin_amdgcn_sched_barrier(0);
                if (ks < 3) {
                    read_group(buf, ks + 1, fr[(ks + 1) & 1]);
                } else {
                    // the barrier that publishes the next group of stages (vmcnt(0) drained by hipcc) and frees
                    // the group just read sits in front of this K-step's last 2*HT MFMAs, so that a wave leaves
                    // it with matrix work in hand while its first fragments of the next K-step are in flight
                    if (BS == 1 || (step % BS) == BS - 1) __syncthreads();
                    read_group((step + 1) % RING, 0, fr[0]);
                }
                if (SB) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < HT; ++t)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        if (ABL != 2)
                            acc[t][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[ks & 1][t], bq[cb][ks], acc[t][cb], 0, 0, 0);
                if (SB) __builtin_amdgcn_sched_barrier(0);
                if (ks < 2) load_b(kn, ks);
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
            const unsigned id0 = (unsigned)((slice * HT + t) * 4);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {       // quad minima, as in select_phase
                    const float qm = quad_min(acc[t][cb][4 * g], acc[t][cb][4 * g + 1], acc[t][cb][4 * g + 2], acc[t][cb][4 * g + 3], NEG_INF);
                    const float v = pack_score(qm, idmask, id0 + g);
                    m2[cb] = __builtin_amdgcn_fmed3f(m1[cb], m2[cb], v);
                    m1[cb] = fast_min(m1[cb], v, NEG_INF);
                }
        }
        if (slice == PPS - 1) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const size_t o = (size_t)(span * 2 + h) * a.Qpad + q0 + cb * 32 + (lane & 31);
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
        const size_t so = (size_t)(chunk * 2 + h) * a.Qpad + q0 + cb * 32 + (lane & 31);
        a.sb_m1[so] = M1[cb];
        a.sb_m2[so] = M2[cb];
        a.sb_span[so] = Ms[cb];
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
    int spans_per_chunk, chunk_rem, nchunks, k;
    int groups;                  // bins per span: 2 (32-row tiles, 512-row spans) or 4 (p16: 16-row tiles, 1024-row spans)
    int direct_rows;             // 0, or: the level-1 bins (direct_rows = 128 or 64 consecutive rows each) ARE the
                                 // superbins -- sb_m1 / sb_m2 alias bin_m1 / bin_m2, sb_span is unused (large k on
                                 // mid-size corpora, where N/256 superbins would be fewer than 4k)
    int cand_cap, rescan_cap;
    int f16_gmode;               // fp16 scan: 0 = candidate groups are quads, 4 = octs (scan_kernel<.., G8>); the int8 scan says so
                                 // itself (QueryBatchInfo.i8_mode)
    int32_t *cand_rows;          // [nq][cand_cap]
    int32_t *rescan_rows;        // [nq][rescan_cap]
    int32_t *counts;             // [nq][2]
    int32_t *fallback;           // [nq]
    int32_t *fb_list;            // [nq]
    int32_t *fb_count;           // [1]
    unsigned long long *stat_counters;  // [2] total candidates, total rescans
};

// T^ = tau + 2 eps.  fp16 scan: float scores.  int8 scan (scan_i8.hpp): the values are packed integer keys
// (t' << 6 | quad id) stored as float bit patterns and eps[q] carries the BITS of (2 eps) << 6: every key whose t' is
// within the bound compares <= ((tau's t' << 6) | 63) + that increment; integer add on the bit patterns.
__device__ __forceinline__ float select_threshold(float tau, float eps, int i8_mode) {
    return i8_mode ? __int_as_float((__float_as_int(tau) | 63) + __float_as_int(eps)) : tau + 2.0f * eps;
}

template <int VPL>
__global__ __launch_bounds__(256) void select_kernel(SelectArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t q = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (q >= a.nq) return;
    const int dbps = a.direct_rows ? kBinRows / a.direct_rows : 0;       // direct mode: bins per (span, lane group)
    const int nsb = a.direct_rows ? (int)a.nspans * a.groups * dbps : a.nchunks * a.groups;
    unsigned v[VPL];
    // All VPL loads are issued before any is used: a load under `s < nsb ? .. : ..` becomes a branch with its own
    // s_waitcnt vmcnt(0) (hipcc also sinks a clamped-index load back under the condition), i.e. VPL serialised memory
    // round trips per wave.  The opaque asm makes every loaded value "used" unconditionally.
    float sv[VPL];
#pragma unroll
    for (int e = 0; e < VPL; ++e) sv[e] = a.sb_m1[(size_t)min(e * 64 + lane, nsb - 1) * a.Qpad + q];
#pragma unroll
    for (int e = 0; e < VPL; ++e) asm volatile("" : "+v"(sv[e]));
#pragma unroll
    for (int e = 0; e < VPL; ++e) v[e] = (e * 64 + lane < nsb) ? sortable_u32(sv[e]) : 0xFFFFFFFFu;
    // k-th smallest by bisection from the top bit.  Any U >= tau is a valid threshold (it only lets more bins through), so
    // with many values per lane (VPL >= 4) and k <= 64 the bisection runs on each lane's TWO smallest values only: their
    // k-th smallest is >= tau (a subset), and equal to it unless three of the k best superbins share a lane -- 2 ballots per
    // step instead of VPL (a single query's select is one wave walking this chain alone: 15 us of a 67 us search).
    unsigned ans = 0;
    if (VPL >= 4 && a.k <= 64) {
        unsigned lo0 = 0xFFFFFFFFu, lo1 = 0xFFFFFFFFu;
#pragma unroll
        for (int e = 0; e < VPL; ++e) {
            const unsigned x = v[e];
            lo1 = min(lo1, max(lo0, x));
            lo0 = min(lo0, x);
        }
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned trial = ans | ((1u << bit) - 1u);
            const int cnt = __popcll(__ballot(lo0 <= trial)) + __popcll(__ballot(lo1 <= trial));
            if (cnt < a.k) ans |= (1u << bit);
        }
    } else {
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned trial = ans | ((1u << bit) - 1u);
            int cnt = 0;
#pragma unroll
            for (int e = 0; e < VPL; ++e) cnt += __popcll(__ballot(v[e] <= trial));
            if (cnt < a.k) ans |= (1u << bit);
        }
    }
    const float tau = unsortable_f32(ans);
    const int i8_mode = a.info->i8_mode;              // (read ONCE: inside the loops below every use would be a fresh global load)
    const float that = select_threshold(tau, a.eps[q], i8_mode);
    const int gmode = i8_mode ? i8_mode : a.f16_gmode;     // rows per candidate group: quads, or octs (bit 2)
    const bool force_fb = a.info->force_fallback || !(that < 0.9e38f) || nsb < a.k;

    int ncand = 0, nres = 0;  // wave-uniform
    int32_t *cr = a.cand_rows + (size_t)q * a.cand_cap;
    int32_t *rr = a.rescan_rows + (size_t)q * a.rescan_cap * 2;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    if (!force_fb) {
        // second minimum and span of every superbin, requested up front and unconditionally (clamped index): under
        // `if (active)` each of the VPL iterations below paid its own memory round trip
        float sm2v[VPL];
        int spanv[VPL];
#pragma unroll
        for (int e = 0; e < VPL; ++e) sm2v[e] = a.sb_m2[(size_t)min(e * 64 + lane, nsb - 1) * a.Qpad + q];
#pragma unroll
        for (int e = 0; e < VPL; ++e) spanv[e] = a.direct_rows ? 0 : a.sb_span[(size_t)min(e * 64 + lane, nsb - 1) * a.Qpad + q];
#pragma unroll
        for (int e = 0; e < VPL; ++e) asm volatile("" : "+v"(sm2v[e]), "+v"(spanv[e]));
#pragma unroll
        for (int e = 0; e < VPL; ++e) {
            const int s = e * 64 + lane;
            const bool in = s < nsb;
            const float m1 = in ? unsortable_f32(v[e]) : __builtin_inff();
            const bool active = in && (m1 <= that);
            const float sm2 = active ? sm2v[e] : __builtin_inff();
            const int sspan = active ? spanv[e] : 0;
            if (a.direct_rows) {   // the superbin is a bin of direct_rows consecutive rows: candidate quad or re-scan, no walk
                const int row0 = (s / dbps) * kBinRows + (s % dbps) * a.direct_rows;
                const bool single = active && !(sm2 <= that), deep = active && !single;
                const unsigned long long smask = __ballot(single), rmask = __ballot(deep);
                if (single) {
                    const int pos = ncand + __popcll(smask & lt_mask);
                    if (pos < a.cand_cap) cr[pos] = row0 + cand_row_offset(__float_as_uint(m1), gmode);
                }
                if (deep) {
                    const int pos = nres + __popcll(rmask & lt_mask);
                    if (pos < a.rescan_cap) {
                        rr[2 * pos] = row0;
                        rr[2 * pos + 1] = row0 + a.direct_rows;   // (clipped to N by the refine kernel)
                    }
                }
                ncand += __popcll(smask);
                nres += __popcll(rmask);
                continue;
            }
            const bool single = active && !(sm2 <= that);
            // superbins whose only interesting score is their minimum: one candidate row each
            const unsigned long long smask = __ballot(single);
            if (single) {
                const int pos = ncand + __popcll(smask & lt_mask);
                const int hh = s % a.groups;
                const int row = (sspan * a.groups + hh) * kBinRows + cand_row_offset(__float_as_uint(m1), gmode);
                if (pos < a.cand_cap) cr[pos] = row;
            }
            ncand += __popcll(smask);
            // superbins with two or more interesting scores: walk their level-1 bins
            unsigned long long dmask = __ballot(active && !single);
            while (dmask) {
                const int src = __ffsll(dmask) - 1;
                dmask &= dmask - 1;
                const int sb = e * 64 + src;
                const int chunk = sb / a.groups, hh = sb % a.groups;
                const int64_t sp0 = chunk_span0(chunk, a.spans_per_chunk, a.chunk_rem);
                int64_t sp1 = chunk_span0(chunk + 1, a.spans_per_chunk, a.chunk_rem);
                if (sp1 > a.nspans) sp1 = a.nspans;
                for (int64_t base = sp0; base < sp1; base += 64) {
                    const int64_t sp = base + lane;
                    bool act = false, resc = false;
                    float bm1 = 0.f;
                    if (sp < sp1) {
                        const size_t o = (size_t)(sp * a.groups + hh) * a.Qpad + q;
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
                            cr[pos] = (int)((sp * a.groups + hh) * kBinRows + cand_row_offset(__float_as_uint(bm1), gmode));
                    }
                    if (resc) {
                        const int pos = nres + __popcll(rm & lt_mask);
                        if (pos < a.rescan_cap) {
                            rr[2 * pos] = (int)((sp * a.groups + hh) * kBinRows);
                            rr[2 * pos + 1] = rr[2 * pos] + kBinRows;   // (clipped to N by the refine kernel)
                        }
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
            stat_add(a.stat_counters, q, 2, 1ull);
        } else {
            stat_add(a.stat_counters, q, 0, (unsigned long long)ncand);
            stat_add(a.stat_counters, q, 1, (unsigned long long)nres);
        }
    }
}

// ---- select, multi-lane form: LPQ lanes per query, 64/LPQ queries per wave ----------------------------------
// The superbin arrays are [superbin][query]: for one superbin row the 16 queries of a wave are one 64-byte
// segment, so every lane-instruction reads whole sectors (the one-wave-per-query form above strides by
// Qpad*4 bytes and is latency bound).  Lane (qi = lane>>2, part = lane&3) keeps superbins part, part+4, ...
// of query qi in registers; counts are reduced over the 4 lanes with two shuffles.  Work lists are appended
// through per-query LDS counters.
template <int V, int LPQ>
__global__ __launch_bounds__(256, 1) void select_kernel_v2(SelectArgs a) {
    constexpr int QPW = 64 / LPQ;
    constexpr int kDeepCap = 48;                    // queued deep superbins per query (more -> exhaustive fallback)
    __shared__ int s_cnt[4][QPW][3];
    __shared__ int s_deep[4][QPW][kDeepCap];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qi = lane / LPQ, part = lane % LPQ;
    const int64_t q = ((int64_t)blockIdx.x * 4 + wave) * QPW + qi;   // < Qpad by construction of the grid
    const bool qvalid = q < a.nq;
    const int nsb = a.nchunks * a.groups;
    if (part < 3) s_cnt[wave][qi][part] = 0;
    unsigned v[V];
    float sv[V];                                    // (all loads in flight before the first use: see select_kernel)
#pragma unroll
    for (int e = 0; e < V; ++e) sv[e] = a.sb_m1[(size_t)min(e * LPQ + part, nsb - 1) * a.Qpad + q];
#pragma unroll
    for (int e = 0; e < V; ++e) asm volatile("" : "+v"(sv[e]));
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = (e * LPQ + part < nsb) ? sortable_u32(sv[e]) : 0xFFFFFFFFu;
    unsigned ans = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned trial = ans | ((1u << bit) - 1u);
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < V; ++e) cnt += (v[e] <= trial) ? 1 : 0;
#pragma unroll
        for (int o = 1; o < LPQ; o <<= 1) cnt += __shfl_xor(cnt, o);
        if (cnt < a.k) ans |= (1u << bit);
    }
    const float tau = unsortable_f32(ans);
    const int i8_mode = a.info->i8_mode;              // (read ONCE, see select_kernel)
    const float that = select_threshold(tau, qvalid ? a.eps[q] : 0.f, i8_mode);
    const int gmode = i8_mode ? i8_mode : a.f16_gmode;     // rows per candidate group: quads, or octs (bit 2)
    const bool force_fb = a.info->force_fallback || !(that < 0.9e38f) || nsb < a.k;
    __syncthreads();  // counters zeroed
    int *cnt_c = &s_cnt[wave][qi][0], *cnt_r = &s_cnt[wave][qi][1], *cnt_d = &s_cnt[wave][qi][2];
    int *deep = &s_deep[wave][qi][0];
    bool deep_overflow = false;
    if (qvalid && !force_fb) {
        int32_t *cr = a.cand_rows + (size_t)q * a.cand_cap;
        // second minimum and span of every ACTIVE superbin, loaded up front so the latencies overlap
        float sm2v[V];
        int spanv[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int s = e * LPQ + part;
            const bool act = (s < nsb) && (unsortable_f32(v[e]) <= that);
            sm2v[e] = act ? a.sb_m2[(size_t)s * a.Qpad + q] : __builtin_inff();
            spanv[e] = act ? a.sb_span[(size_t)s * a.Qpad + q] : 0;
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int s = e * LPQ + part;
            if (s >= nsb) continue;
            const float m1 = unsortable_f32(v[e]);
            if (!(m1 <= that)) continue;
            const int hh = s % a.groups;
            if (!(sm2v[e] <= that)) {  // only the superbin minimum matters
                const int pos = atomicAdd(cnt_c, 1);
                if (pos < a.cand_cap)
                    cr[pos] = (spanv[e] * a.groups + hh) * kBinRows + cand_row_offset(__float_as_uint(m1), gmode);
            } else {               // two or more interesting scores: queue the superbin for the cooperative walk below
                const int pos = atomicAdd(cnt_d, 1);
                if (pos < kDeepCap) deep[pos] = s;
            }
        }
    }
    __syncthreads();
    // Deep superbins: the LPQ lanes of a query walk the level-1 bins of one queued superbin TOGETHER (lane `part` takes
    // spans part, part + LPQ, ...), so a superbin costs two dependent memory round trips however many spans its chunk
    // has -- walking it inside the loop above serialised a wave on every lane that met one.
    {
        const int nd_q = (qvalid && !force_fb) ? *cnt_d : 0;
        int nd_max = nd_q;                                   // loop count must be wave-uniform
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) nd_max = max(nd_max, __shfl_xor(nd_max, o));
        if (nd_q > kDeepCap) deep_overflow = true;
        int32_t *cr = a.cand_rows + (size_t)q * a.cand_cap;
        int32_t *rr = a.rescan_rows + (size_t)q * a.rescan_cap * 2;
        for (int d = 0; d < nd_max && d < kDeepCap; ++d) {
            if (d >= nd_q) continue;
            const int s = deep[d];
            const int chunk = s / a.groups, hh = s % a.groups;
            const int64_t sp0 = chunk_span0(chunk, a.spans_per_chunk, a.chunk_rem);
            int64_t sp1 = chunk_span0(chunk + 1, a.spans_per_chunk, a.chunk_rem);
            if (sp1 > a.nspans) sp1 = a.nspans;
            for (int64_t sp = sp0 + part; sp < sp1; sp += LPQ) {
                const size_t o = (size_t)(sp * a.groups + hh) * a.Qpad + q;
                const float bm1 = a.bin_m1[o];
                if (!(bm1 <= that)) continue;
                if (a.bin_m2[o] <= that) {
                    const int pos = atomicAdd(cnt_r, 1);
                    if (pos < a.rescan_cap) {
                        rr[2 * pos] = (int)((sp * a.groups + hh) * kBinRows);
                        rr[2 * pos + 1] = rr[2 * pos] + kBinRows;   // (clipped to N by the refine kernel)
                    }
                } else {
                    const int pos = atomicAdd(cnt_c, 1);
                    if (pos < a.cand_cap)
                        cr[pos] = (int)((sp * a.groups + hh) * kBinRows + cand_row_offset(__float_as_uint(bm1), gmode));
                }
            }
        }
    }
    __syncthreads();
    if (qvalid && part == 0) {
        const int ncand = *cnt_c, nres = *cnt_r;
        const bool fb = force_fb || deep_overflow || ncand > a.cand_cap || nres > a.rescan_cap;
        a.counts[2 * q] = fb ? 0 : ncand;
        a.counts[2 * q + 1] = fb ? 0 : nres;
        a.fallback[q] = fb ? 1 : 0;
        if (fb) {
            const int pos = atomicAdd(a.fb_count, 1);
            a.fb_list[pos] = (int)q;
            stat_add(a.stat_counters, q, 2, 1ull);
        } else {
            stat_add(a.stat_counters, q, 0, (unsigned long long)ncand);
            stat_add(a.stat_counters, q, 1, (unsigned long long)nres);
        }
    }
}

}  // namespace vdb
