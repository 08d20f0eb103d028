// ivf_kloop.hpp -- list-major IVF-Flat scan on the matrix cores for D > 128 (384 / 768-dim embeddings).
//
// The D <= 128 list scan (scan_kernel / scan_i8_kernel in ITEMS mode) keeps a work item's query fragments in registers;
// beyond 128 dims they no longer fit, so this is the items-mode form of scan16_kloop_kernel (scan16.hpp): the
// contraction is tiled like a GEMM and the B fragments stream in per 64-dim K-step -- GATHERED, because the 64 query
// slots of a wave are arbitrary queries of the batch: lane (column c = lane & 15, k quarter kq = lane >> 4) reads 16
// bytes of ITS slot's scaled fp16 query row (ivf_prep_kernel writes them, [nq][Dpad]) through a buffer descriptor with
// a per-lane row offset.  No slot-ordered copy of the queries is ever materialised (it would be nprobe x the batch).
//
// Work item = (one inverted list) x (512 query slots), workgroup = 8 waves x 64 slots; blockIdx.y = row part of the
// list (part_spans spans each).  A wave whose 64 slots are all padding only keeps its share of the A staging and the
// barriers going.  Columns of padding slots inside a live wave compute against query 0 -- nobody reads their bins.
//
// Panel layout "p16" over the list-padded panel space, span = TPS tiles of 16 rows: tile t of a span, MFMA row 4g + i
// <-> local row (4 TPS) g + 4 t + i, so lane group g = lane >> 4 walks the 4 TPS CONSECUTIVE rows of bin (span, g), one
// quad per tile, and the packed 6-bit id of a bin minimum is the tile number t (TPS <= 64).
//   TPS = 16: 256-row spans, four 64-row bins  (lists of ~1000 rows: +13 % padding; many small bins per query)
//   TPS = 64: 1024-row spans, four 256-row bins (long lists: 4x fewer select entries per probed row)
// Per pass a wave owns 8 tiles x 4 column blocks (128 rows x 64 slots, 128 accumulator registers), TPS / 8 passes
// complete the four bins of a span; bins keep their five smallest quad minima (kIvfKloopMinima below; the D <= 128 items-mode
// scans keep three, scan.hpp) and leave in [item][slot][bin] order for ivf_select_kernel.  K-step pipeline, LDS ring and barrier placement are those of
// scan16_kloop_kernel.
#pragma once
#include "scan16.hpp"

namespace vdb {

// Quad minima kept per bin.  The D <= 128 scans keep three (one v_med3 each per quad: their select is issue-bound); here the
// select runs once per pass of D / 64 K-steps, so two more cost ~1 % -- and they matter: the near neighbours of a query sit
// in one or two of its probed lists, i.e. several per 64-row bin, and a bin with more close quads than minima kept must be
// re-scored whole (with three minima the msmarco-shaped leg re-scored 5.8 bins = 370 rows of 1.5 KB per query, 70 % of
// its time; with five a bin yields up to four candidate quads before it has to be re-scanned).
constexpr int kIvfKloopMinima = 5;

struct IvfKloopArgs {
    const half8 *panels;         // [pspans * TPS tiles][KS][64]
    const float *bias;           // [pspans * TPS * 16] linear in the local row of the span; kPadBias on padding rows
    const _Float16 *qrows;       // [nq][32 * KS] scaled fp16 query rows
    const QueryBatchInfo *info;
    float *bin_m[kIvfKloopMinima];   // the NM smallest quad minima of every bin, each with its quad id
    const int32_t *item_list, *item_slot0, *item_bin0, *n_items, *list_pspan0, *slot_query;
    int part_spans;              // spans per row part (blockIdx.y), 0 = whole list
    int ksteps;                  // 16-dim k-steps of the index (multiple of 4)
};

constexpr int kIvfKloopGroup = 512;     // query slots per work item

// GROUP = rows per candidate group (the unit a packed 6-bit id names and the refine re-scores): 4 = the quad of a tile (any TPS),
// 2 / 1 = pairs / single rows (TPS = 16 only: 32 / 64 ids per 64-row bin).  Smaller groups cost select instructions here
// (one sorted insertion per group: 9 -> 14 -> 24 vector ops per quad of scores) and save gathered rows in the refine, which
// is bound by exactly that gather: 90 -> 47 -> 26 rows of 1.5 KB per query on the msmarco-shaped leg.
// RH = row halves of the workgroup tile (round 4).  RH = 1 (default): 128 rows x 512 slots per pass, every wave owns 64 slots.
// RH = 2 (TPS = 16 only, option "ivf_tile" = 2): **256 rows x 256 slots** -- waves w and w + 4 share 64 slots and take the lower /
// upper 8 tiles of the span; the upper half hands its five minima over through LDS at the end of the span (one pass = one span)
// and the lower half merges and stores.  Why it was built: the scan is not bound by the matrix pipe (PMC: 21 % busy) but by
// the bytes it pulls through the fabric -- per list A x (slots / tile slots) + B x (rows / tile rows) = 0.77 MB x 6.25 + 2.4 MB x
// 7.8 = 23.6 MB with 128 x 512 tiles, 2.5 GB per msmarco-shaped batch (PMC: 2.53 GB of FETCH_SIZE per launch, 5 TB/s) -- and a
// square tile of the same accumulator budget moves 19 MB with 4 % instead of 12 % slot padding.  What the A/B said
// (profiles/r04_sweeps.txt): 0.516 vs 0.499 ms -- no gain.  The two waves of a slot group both gather the B fragments (the
// second from L2), so the per-CU request rate is unchanged, and that, not the fabric byte count, is the limit: 2.45 GB of
// gathered 64-byte segments + 1 GB of panel rows per 0.5 ms are 27 GB/s per CU against the 33.5 GB/s per CU a pure
// Infinity-Cache gather reaches on this chip (MI355X_MICROARCH.md, "Indexed rows").  The form stays as a tested option.
template <int TPS, int BS, int GROUP = 4, int RH = 1>
__global__ __launch_bounds__(512, 2) void ivf_kloop_scan_kernel(IvfKloopArgs a) {
    static_assert(GROUP == 4 || ((GROUP == 2 || GROUP == 1) && TPS == 16), "pairs / rows need the 64-row bins of TPS = 16");
    static_assert(RH == 1 || (RH == 2 && TPS == 16), "the square tile covers one 256-row span per pass");
    constexpr int NWAVES = 8, RING = 2 * BS, HTW = 8, HT = HTW * RH, CB = 4;
    constexpr int CG = NWAVES / RH;                          // slot groups of 64 per work item
    constexpr int kSlots = 64 * CG;                          // query slots per work item (512 or 256)
    constexpr int PPS = TPS / HT;                            // passes per span
    constexpr int SPANROWS = TPS * 16, BINROWS = TPS * 4;
    static_assert(TPS % HT == 0 && TPS <= 64, "the tile number must fit the 6-bit id");
    constexpr int kStageVec = HT * 2 * 64;                   // 16-byte vectors per K-step stage (16 KiB)
    __shared__ __attribute__((aligned(16))) unsigned char smem[RING * kStageVec * 16];
    auto lds_a = [&](int buf) { return reinterpret_cast<half8 *>(smem + buf * (kStageVec * 16)); };

    const int it = blockIdx.x;
    if (it >= *a.n_items) return;
    const int l = a.item_list[it];
    int64_t span0 = a.list_pspan0[l], span1 = a.list_pspan0[l + 1];
    const int64_t lspan0 = span0;
    const int nb_item = (int)(span1 - span0) * 4;            // bins per query slot in this item
    if (a.part_spans > 0) {
        span0 += (int64_t)blockIdx.y * a.part_spans;
        if (span0 >= span1) return;
        if (span0 + a.part_spans < span1) span1 = span0 + a.part_spans;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const int KS = a.ksteps / 2, nK = KS / 2;                // 32-dim k-steps; 64-dim K-steps
    const int rh = RH == 1 ? 0 : wave / CG, cg = RH == 1 ? wave : wave % CG;      // row half, slot group
    const int slot0 = a.item_slot0[it] + cg * 64;
    const size_t bin_base = (size_t)a.item_bin0[it];
    const float cs = a.info->cs;

    const int npass = (int)(span1 - span0) * PPS;
    const int nsteps = npass * nK;

    // the queries behind this wave's slots: per column block one byte offset into qrows (row pitch 64 KS bytes)
    int qoff[CB];
    bool any_query = false;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int qi = a.slot_query[slot0 + cb * 16 + (lane & 15)] - 1;      // (-1: padding slot)
        any_query |= qi >= 0;
        qoff[cb] = (qi < 0 ? 0 : qi) * (KS * 64) + g * 16;
    }
    const bool live = __builtin_amdgcn_readfirstlane((int)(__ballot(any_query) != 0ull)) != 0;

    const int lane16 = lane * 16;
    auto stage_issue = [&](int step, int buf) {              // A panels of K-step `step` -> LDS slot
        const int pass = step / nK, kk = step - pass * nK;
        const int64_t tile0 = (span0 + pass / PPS) * TPS + (pass % PPS) * HT;
        const half8 *base = a.panels + ((size_t)tile0 * KS + kk * 2) * 64;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half8 *>(base), 0, 0x7fffffff, 0x00020000);
        half8 *dst = lds_a(buf);
#pragma unroll
        for (int i = 0; i < (HT * 2) / NWAVES; ++i) {
            const int p = wave + i * NWAVES;                 // piece = (tile t, k-step ks)
            const int t = p >> 1, ks = p & 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rs, reinterpret_cast<__attribute__((address_space(3))) void *>(
                        static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dst + p * 64))),
                16, lane16, (t * KS + ks) * 1024, 0, 0);
        }
    };
#pragma unroll
    for (int i = 0; i < BS; ++i)
        if (i < nsteps) stage_issue(i, i);
    if (!live) {                                             // all 64 slots are padding: staging + barriers only
        __syncthreads();
        for (int step = 0; step < nsteps; ++step) {
            stage_issue(step + BS < nsteps ? step + BS : nsteps - 1, (step + BS) % RING);
            if (BS == 1 || (step % BS) == BS - 1) __syncthreads();
            if (RH == 2 && (step % nK) == nK - 1) __syncthreads();       // (the hand-over barrier of every pass)
        }
        return;
    }

    const float INF = __builtin_inff();
    float NEG_INF = -INF;
    asm volatile("" : "+v"(NEG_INF));
    unsigned idmask = kQuadIdMask;
    asm volatile("" : "+v"(idmask));
    constexpr int NM = kIvfKloopMinima;
    // The running minima of the current bins live in LDS between passes and are in registers only inside the epilogue: with
    // 128 accumulators, 32 + 32 fragment registers and these 20 the K-loop spilled (256 VGPRs, 20 spilled, 84 bytes of
    // scratch per lane: kernel-resource-usage of round 4's first build); 40 ds_read / ds_write per pass of 384 MFMAs are free.
    // (RH = 2: one pass completes its span, nothing persists -- the array is the hand-over buffer of the upper row half)
    __shared__ float s_min[RH == 1 ? NWAVES : CG][NM * CB][64];
    if (RH == 1) {
#pragma unroll
        for (int i = 0; i < NM * CB; ++i) s_min[wave][i][lane] = INF;
    }

    float4v acc[HTW][CB];
    half8 bq[CB][2];          // B fragments of the current K-step, reloaded in place (see scan_kloop_kernel)
    const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(a.qrows), 0, 0x7fffffff, 0x00020000);
    auto load_b = [&](int kk, int ks) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
            bq[cb][ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsq, qoff[cb], (kk * 2 + ks) * 64, 0));
    };
    load_b(0, 0);
    __syncthreads();

    half8 fr[2][4];
    auto read_group = [&](int buf, int grp, half8(&dst)[4]) {   // group grp = (ks = grp>>1, tiles 4*(grp&1)..+3)
        const half8 *A = lds_a(buf);
#pragma unroll
        for (int t = 0; t < 4; ++t) dst[t] = A[((rh * HTW + (grp & 1) * 4 + t) * 2 + (grp >> 1)) * 64 + lane];
    };
    read_group(0, 0, fr[0]);

    int step = 0;
    for (int pass = 0; pass < npass; ++pass) {
        const int64_t span = span0 + pass / PPS;
        const int slice = pass % PPS;
        // accumulators start from the bias of their rows: local row = BINROWS * g + 4 * (HT * slice + t) + i
#pragma unroll
        for (int t = 0; t < HTW; ++t) {
            const float4 c = *reinterpret_cast<const float4 *>(a.bias + span * SPANROWS + g * BINROWS + (slice * HT + rh * HTW + t) * 4);
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
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        acc[(grp & 1) * 4 + t][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                            fr[grp & 1][t], bq[cb][grp >> 1], acc[(grp & 1) * 4 + t][cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (grp == 1) load_b(kn, 0);
            }
        }
        float m[NM][CB];           // m[0] <= m[1] <= ... : sorted insertion by one v_min + (NM - 1) v_med3 per group
#pragma unroll
        for (int i = 0; i < NM; ++i)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) m[i][cb] = RH == 1 ? s_min[wave][i * CB + cb][lane] : INF;
#pragma unroll
        for (int t = 0; t < HTW; ++t) {
            const unsigned id = (unsigned)(slice * HT + rh * HTW + t);  // tile number inside the span = quad number inside the bin
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                auto insert = [&](float v) {
#pragma unroll
                    for (int i = NM - 1; i > 0; --i) m[i][cb] = __builtin_amdgcn_fmed3f(m[i - 1][cb], m[i][cb], v);
                    m[0][cb] = fast_min(m[0][cb], v, NEG_INF);
                };
                if (GROUP == 4) {
                    const float qm = quad_min(acc[t][cb][0], acc[t][cb][1], acc[t][cb][2], acc[t][cb][3], NEG_INF);
                    insert(pack_score(qm, idmask, id));
                } else if (GROUP == 2) {
                    insert(pack_score(fast_min(acc[t][cb][0], acc[t][cb][1], NEG_INF), idmask, 2 * id));
                    insert(pack_score(fast_min(acc[t][cb][2], acc[t][cb][3], NEG_INF), idmask, 2 * id + 1));
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) insert(pack_score(acc[t][cb][i], idmask, 4 * id + i));
                }
            }
        }
        if (RH == 2) {      // hand-over: the upper row half's minima of the same bins reach the lower half through LDS
            if (rh == 1) {
#pragma unroll
                for (int i = 0; i < NM; ++i)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) s_min[cg][i * CB + cb][lane] = m[i][cb];
            }
            __syncthreads();
            if (rh == 0) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int j = 0; j < NM; ++j) {
                        const float v = s_min[cg][j * CB + cb][lane];
#pragma unroll
                        for (int i = NM - 1; i > 0; --i) m[i][cb] = __builtin_amdgcn_fmed3f(m[i - 1][cb], m[i][cb], v);
                        m[0][cb] = fast_min(m[0][cb], v, NEG_INF);
                    }
            }
        }
        if (slice == PPS - 1 && rh == 0) {                   // the four bins (span, g) are complete: [item][slot][bin]
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const size_t o = bin_base * kSlots + (size_t)(cg * 64 + cb * 16 + (lane & 15)) * nb_item +
                                 (size_t)((span - lspan0) * 4 + g);
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    a.bin_m[i][o] = m[i][cb];
                    m[i][cb] = INF;
                }
            }
        }
        if (RH == 1) {
#pragma unroll
            for (int i = 0; i < NM; ++i)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) s_min[wave][i * CB + cb][lane] = m[i][cb];
        }
    }
}

// ---- build: p16 panels over the list-padded panel space (span = tps tiles; span_row0 / span_valid as ivf_mfma.hpp) ----
__global__ __launch_bounds__(256) void ivf_build_panels16_kernel(const float *__restrict__ X, int D, int D4, int ks32,
                                                                 int tps, int64_t ntiles, float sx,
                                                                 const int32_t *__restrict__ span_row0,
                                                                 const int32_t *__restrict__ span_valid,
                                                                 half8 *__restrict__ panels, IndexStats *st) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(gid & 63);
    const int64_t tk = gid >> 6;
    const int ks = (int)(tk % ks32);
    const int64_t tile = tk / ks32;
    int inexact = 0;
    if (tile < ntiles) {
        const int rho = lane & 15, kq = lane >> 4;
        const int g = rho >> 2, i = rho & 3;
        const int64_t span = tile / tps;
        const int t = (int)(tile - span * tps);
        const int local = 4 * tps * g + 4 * t + i;
        const bool valid = local < span_valid[span];
        const int64_t row = (int64_t)span_row0[span] + local;
        const int d0 = ks * 32 + kq * 8;
        half8 out;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = d0 + j;
            float v = 0.f;
            if (valid && d < D) v = X[(size_t)row * D4 + d] * sx;
            const _Float16 hv = (_Float16)v;
            inexact |= ((float)hv != v);
            out[j] = hv;
        }
        panels[gid] = out;
    }
    if (__any(inexact) && (threadIdx.x & 63) == 0) atomic_set_flag(&st->not_fp16_exact);
}

}  // namespace vdb
