// scan_i8.hpp -- int8 MFMA form of the flat scan for byte-valued integer corpora (SIFT descriptors are uint8).
//
// Same work decomposition, LDS staging, row mapping, bins, superbins and outputs as scan_kernel (scan.hpp); what
// changes is the arithmetic: v_mfma_i32_32x32x32_i8 (32 dims per k-step, twice the fp16 MFMA rate per clock, exact
// int32 accumulation) on A = x - cx, B = cq - q (common.hpp, "int8 scan copy"), accumulator initialised with the
// per-row integer bias, and an integer select: per quad  min3 + min, v_lshl_or (t' << 6 | quad id), med3 + min.
// The packed keys are stored as float bit patterns (normal positive floats ordered like the integers), so the select
// kernels of scan.hpp read them unchanged.  Which of the two scans serves a batch is decided ON THE DEVICE
// (QueryBatchInfo.i8_mode, set by query_finalize (prep.hpp) from the query statistics): both kernels are enqueued and the
// one that is not needed returns at once -- vdb_search_device stays asynchronous.
#pragma once
#include "common.hpp"
#include "prep.hpp"
#include "scan.hpp"

namespace vdb {

typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int16v __attribute__((ext_vector_type(16)));

// ---- build: int8 panels in A-fragment order --------------------------------------------------------------------------
// panels8[tile][ks][lane][16 x int8]: lane l holds MFMA row (l & 31) -- same row mapping as the fp16 panels -- dims
// 32*ks + 16*(l >> 5) .. +15, value x - cx (0 beyond D or N).  One thread per (tile, ks, lane).
// (tile0: tiles [tile0, tile0 + ntiles) are written, to panels[tile0 ...) -- the whole corpus (0), or one row block of an
//  int8-only build, where X is the block's base minus the rows in front of it: only rows of the block are touched)
__global__ __launch_bounds__(256) void build_panels_i8_kernel(const float *__restrict__ X, int64_t N, int D, int D4,
                                                              int ks32, int64_t ntiles, int cx,
                                                              int4v *__restrict__ panels, int64_t tile0 = 0, int x16 = 0) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(gid & 63);
    const int64_t tk = gid >> 6;
    const int ks = (int)(tk % ks32);
    int64_t tile = tk / ks32;
    if (tile >= ntiles) return;
    tile += tile0;
    gid += tile0 * ks32 * 64;
    const int64_t span = tile / kTilesPerSpan;
    const int t = (int)(tile - span * kTilesPerSpan);
    int64_t row;
    int d0;
    if (x16) {       // layout "x16" (scan_i8x16.hpp): piece v = 2 ks2 + rb, lane = (MFMA row m of block rb, 16-dim quarter of the k-step)
        const int m = lane & 15;
        row = span * kSpanRows + (m >> 2) * 128 + t * 8 + (ks & 1) * 4 + (m & 3);
        d0 = (ks >> 1) * 64 + (lane >> 4) * 16;
    } else {
        const int rho = lane & 31, kh = lane >> 5;
        const int r = (rho & 3) | ((rho >> 3) << 2), h = (rho >> 2) & 1;
        row = span * kSpanRows + (int64_t)h * kBinRows + t * 16 + r;
        d0 = ks * 32 + kh * 16;
    }
    int4v out;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        unsigned word = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int d = d0 + 4 * w + b;
            int v = 0;
            if (row < N && d < D) v = (int)X[(size_t)row * D4 + d] - cx;
            word |= ((unsigned)v & 0xffu) << (8 * b);
        }
        out[w] = (int)word;
    }
    panels[gid] = out;
}

// accumulator init per row, for both query windows: bias8[w][row], w = 0 (cq = 127), 1 (cq = -1)
//   L2: floor((||x||^2 - 2 cq sum(x)) / 2) + kI8Offset     IP: -cq sum(x) + kI8Offset     padding rows: kI8PadBias
// (also rowstat[row] = {sum x^2, sum x} for the int8 list refine, refine.hpp)
// (row0 / row1: rows [row0, row1) only -- one row block of an int8-only build; default: all Npad rows)
__global__ __launch_bounds__(256) void build_bias_i8_kernel(const float *__restrict__ X, int64_t N, int64_t Npad, int D,
                                                            int D4, int metric, int32_t *__restrict__ bias8,
                                                            int *__restrict__ rowstat, int64_t row0 = 0, int64_t row1 = -1) {
    const int64_t row = row0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= (row1 < 0 ? Npad : row1)) return;
    if (row >= N) {
        bias8[row] = kI8PadBias;
        bias8[Npad + row] = kI8PadBias;
        return;
    }
    long long n2 = 0, s1 = 0;
    for (int d = 0; d < D; ++d) {
        const long long v = (long long)X[(size_t)row * D4 + d];
        n2 += v * v;
        s1 += v;
    }
    rowstat[2 * row] = (int)n2;
    rowstat[2 * row + 1] = (int)s1;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const long long cq = w == 0 ? 127 : -1;
        const long long b = metric == 0 ? ((n2 - 2 * cq * s1) >> 1) : -cq * s1;    // (>> on a negative: floor)
        bias8[(size_t)w * Npad + row] = (int32_t)(b + kI8Offset);
    }
}

// B fragments: qpanels8[qtile32][ks][lane][16 x int8], lane l holds query column l & 31, dims 32*ks + 16*(l>>5) .. +15,
// value cq - q.  One thread per (qtile32, ks, lane).
__device__ __forceinline__ void build_qpanels_i8_body(int64_t gid, const float *__restrict__ Q, int64_t nq, int D, int ks32,
                                                      int64_t nqtiles, const QueryBatchInfo *__restrict__ info,
                                                      int4v *__restrict__ qpanels, int x16 = 0) {
    const int lane = (int)(gid & 63);
    const int64_t tk = gid >> 6;
    const int mode = info->i8_mode;
    if (tk >= nqtiles * ks32 || !mode) return;
    const int cq = (mode & 3) == 1 ? 127 : -1;
    int64_t q;
    int d0;
    if (x16) {       // [q / 16][ks2][lane]: query column lane & 15, dims 64 ks2 + 16 (lane >> 4) .. +15 (same size as the 32-query form)
        const int ks2n = ks32 / 2;
        q = (tk / ks2n) * 16 + (lane & 15);
        d0 = (int)(tk % ks2n) * 64 + (lane >> 4) * 16;
    } else {
        q = (tk / ks32) * 32 + (lane & 31);
        d0 = (int)(tk % ks32) * 32 + (lane >> 5) * 16;
    }
    int4v out;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        unsigned word = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int d = d0 + 4 * w + b;
            int v = 0;
            if (q < nq && d < D) v = cq - (int)Q[(size_t)q * D + d];
            word |= ((unsigned)v & 0xffu) << (8 * b);
        }
        out[w] = (int)word;
    }
    qpanels[gid] = out;
}

__global__ __launch_bounds__(256) void build_qpanels_i8_kernel(const float *__restrict__ Q, int64_t nq, int D, int ks32,
                                                               int64_t nqtiles, const QueryBatchInfo *__restrict__ info,
                                                               int4v *__restrict__ qpanels, int x16 = 0) {
    build_qpanels_i8_body((int64_t)blockIdx.x * blockDim.x + threadIdx.x, Q, nq, D, ks32, nqtiles, info, qpanels, x16);
}

// row-major int8 copy for the list refine (refine.hpp, RefineCommon.X8): rows[row][pitch] = x - cx, bytes beyond D hold
// -cx (x = 0).  One thread per (row, 4-byte word).
__global__ __launch_bounds__(256) void build_rows_i8_kernel(const float *__restrict__ X, int64_t N, int D, int D4, int pitch,
                                                            int cx, signed char *__restrict__ rows) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int wpr = pitch / 4;
    const int64_t row = gid / wpr;
    if (row >= N) return;
    const int w = (int)(gid - row * wpr);
    unsigned word = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int d = 4 * w + b;
        const int v = (d < D ? (int)X[(size_t)row * D4 + d] : 0) - cx;
        word |= ((unsigned)v & 0xffu) << (8 * b);
    }
    reinterpret_cast<unsigned *>(rows)[gid] = word;
}

// ---- int8-only indexes (option "int8_only"): the fp16 scan copy of a slab of tiles, converted from the int8 panels per search ---
// (x16 = 1: both copies in layout "x16".)
// fp16 piece (tile, 16-dim k-step ks, lane = (row rho, k half kh)) = dims [16 ks + 8 kh, +8) of MFMA row rho = 8 consecutive bytes
// of the int8 piece (tile, ks / 2, lane' = rho + 32 * ((ks & 1) * 16 + 8 kh >= 16)) -- values byte + cx, exact in fp16.  Rows past N and
// dims past D hold A = 0 in the int8 panels (x = cx there), so those positions must come out as 0: `N`, `D` are re-checked here.
// Returns at once when the batch is served by the int8 scan (the usual case: the slab is the price of deciding on the device).
__global__ __launch_bounds__(256) void convert_slab_from_i8_kernel(const int4v *__restrict__ panels8, int ks32, int ksteps, int cx,
                                                                   int64_t tile0, int64_t ntiles, int64_t N, int D,
                                                                   const QueryBatchInfo *__restrict__ info,
                                                                   half8 *__restrict__ out, int x16 = 0) {
    if (info->i8_mode) return;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(gid & 63);
    const int64_t tk = gid >> 6;
    const int ks = (int)(tk % ksteps);
    const int64_t tl = tk / ksteps;
    if (tl >= ntiles) return;
    const int64_t tile = tile0 + tl;
    const int64_t span = tile / kTilesPerSpan;
    const int t = (int)(tile - span * kTilesPerSpan);
    int64_t row;
    int d0;
    int4v src;
    if (x16) {       // both copies in layout "x16": fp16 piece v = 2 ks2 + rb of the tile <- the int8 piece of the same block rb that holds its 8 dims
        const int m = lane & 15, rb = ks & 1;
        row = span * kSpanRows + (m >> 2) * 128 + t * 8 + rb * 4 + (m & 3);
        d0 = (ks >> 1) * 32 + (lane >> 4) * 8;
        src = panels8[((size_t)tile * ks32 + (d0 >> 6) * 2 + rb) * 64 + ((d0 & 63) >> 4) * 16 + m];
    } else {
        const int rho = lane & 31, kh = lane >> 5;
        const int r = (rho & 3) | ((rho >> 3) << 2), h = (rho >> 2) & 1;
        row = span * kSpanRows + (int64_t)h * kBinRows + t * 16 + r;
        d0 = ks * 16 + kh * 8;
        const int o = (ks & 1) * 16 + kh * 8;                   // dim offset inside the 32-dim int8 k-step
        src = panels8[((size_t)tile * ks32 + (ks >> 1)) * 64 + rho + 32 * (o >> 4)];
    }
    const int w0 = src[(d0 & 15) >> 2], w1 = src[((d0 & 15) >> 2) + 1];
    half8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int b = (j < 4 ? (w0 >> (8 * j)) : (w1 >> (8 * (j - 4)))) & 0xff;
        const int x = (int)(signed char)b + cx;
        v[j] = (_Float16)((row < N && d0 + j < D) ? (float)x : 0.f);
    }
    out[gid] = v;
}

// queries for vdb_reserve on an int8-only index: corpus rows as float32 (x = byte + cx)
__global__ __launch_bounds__(256) void rows_i8_to_float_kernel(const signed char *__restrict__ rows8, int pitch, int cx, int64_t n,
                                                               int D, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    const int64_t row = i / D;
    const int d = (int)(i - row * D);
    out[i] = (float)((int)rows8[(size_t)row * pitch + d] + cx);
}

// ---- the scan ------------------------------------------------------------------------------------------------------
struct ScanI8Args {
    const int4v *panels;     // [ntiles][KS][64]
    const int32_t *bias8;    // [2][Npad]
    const int4v *qpanels;    // [Qpad/32][KS][64]
    const QueryBatchInfo *info;
    float *bin_m1, *bin_m2;  // [nspans*2][Qpad]   packed keys as float bit patterns
    float *sb_m1, *sb_m2;    // [nchunks*2][Qpad]
    int32_t *sb_span;
    int64_t nspans, Npad, Qpad, nq_valid;
    int spans_per_chunk, chunk_rem, nchunks, nqtiles;
    // ---- IVF ("items") mode, as ScanArgs: one workgroup = (one inverted list) x (one group of 64*NWAVES query slots) ----
    float *bin_m3;               // third-smallest quad minimum of each bin
    const int32_t *item_list, *item_slot0, *item_bin0, *n_items, *list_pspan0, *slot_query;
    const signed char *qrows;    // [nq][32*KS] int8 query rows cq - q (B fragments are gathered from them)
    int part_spans;              // items mode: spans per row part (blockIdx.y), 0 = whole list (see ScanArgs)
    int prio;                    // x16 kernel (option "scan_prio", tuning): 1 = the late half issues at priority 1, 2 = the early half
    int abl_no_bins;             // -DVDB_ABLATIONS builds only (timing, WRONG results): skip the level-1 bin stores
    unsigned long long *dbg;     // DBG builds (-DVDB_ABLATIONS): per wave {head, mfma, select, tail, barrier, total} shader
                                 // cycles, the s_memrealtime ticks of the same span, (stages << 1) | late
};

// a "use" of the accumulators that makes hipcc wait for the matrix pipe.  In a device-only function: the same asm written
// inside the __global__ body makes the HOST pass drop the kernel's launch stub without a diagnostic (the "v" constraint on
// an int16v is not a valid x86 operand, and in a template the failure is a silent substitution failure)
__device__ __forceinline__ void wait_for_mfma(const int16v &acc) { asm volatile("s_nop 0" ::"v"(acc)); }
__device__ __forceinline__ unsigned long long realtime_ticks() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imed3(int a, int b, int c) { return imax(imin(a, b), imin(imax(a, b), c)); }   // v_med3_i32

// select of one tile: per column block 16/G groups of G consecutive corpus rows (G = 4 quads, 8 octs);
// v = (min of the group << 6) | group id.  The id is forced into an SGPR: in the rolled tile loop hipcc otherwise packs
// with a shift and a three-operand add (2 VALU ops) instead of one v_lshl_or_b32.
template <int CB, bool M3 = false, int G = 4>
__device__ __forceinline__ void select_phase_i8(const int16v (&acc)[CB], int (&m1)[CB], int (&m2)[CB], unsigned id0,
                                                int *m3 = nullptr) {
#pragma unroll
    for (int g = 0; g < 16 / G; ++g) {
        const unsigned idg = (unsigned)__builtin_amdgcn_readfirstlane((int)(id0 + g));
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            int q;
            if (G == 8) {      // three v_min3_i32 + one v_min_i32 (a balanced tree of two-operand minima compiles to 2 + 3)
                const int t1 = imin(imin(acc[cb][G * g], acc[cb][G * g + 1]), acc[cb][G * g + 2]);
                const int t2 = imin(imin(acc[cb][G * g + 3], acc[cb][G * g + 4]), acc[cb][G * g + 5]);
                q = imin(imin(imin(acc[cb][G * g + 6], acc[cb][G * g + 7]), t1), t2);
            } else {
                q = imin(imin(acc[cb][G * g], acc[cb][G * g + 1]), imin(acc[cb][G * g + 2], acc[cb][G * g + 3]));
            }
            const int v = (int)(((unsigned)q << 6) | idg);
            if (M3) m3[cb] = imed3(m2[cb], m3[cb], v);       // (items mode: third minimum, see scan.hpp select_phase)
            m2[cb] = imed3(m1[cb], m2[cb], v);
            m1[cb] = imin(m1[cb], v);
        }
    }
}

template <int KS>
__device__ __forceinline__ void read_phase_i8(const int4v *__restrict__ A_tile, const int4v *__restrict__ c_tile,
                                              int4v (&fr)[KS], int16v &cin, int lane) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fr[ks] = A_tile[ks * 64 + lane];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int4v c = c_tile[g];
        cin[4 * g + 0] = c.x; cin[4 * g + 1] = c.y; cin[4 * g + 2] = c.z; cin[4 * g + 3] = c.w;
    }
}

template <int KS, int CB>
__device__ __forceinline__ void mfma_phase_i8(const int4v (&fr)[KS], const int4v (&bq)[CB][KS], const int16v &cin,
                                              int16v (&acc)[CB]) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
            acc[cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr[ks], bq[cb][ks], ks == 0 ? cin : acc[cb], 0, 0, 0);
}

// KS: 32-dim k-steps (D padded to 32*KS: 2 or 4); ST: tiles per LDS stage; CB: 32-query column blocks per wave (2 ->
// 64 queries per wave as scan_kernel, 4 -> 128: every A fragment read from LDS feeds 4 MFMAs and a stage carries twice
// the matrix work per barrier).  NWAVES waves, 2 per SIMD, phase-staggered halves exactly as scan_kernel: the early half
// runs MFMA(t) then select(t), the late half select(t-1) then MFMA(t).  BT: tiles per level-1 bin; ITEMS: IVF work-item
// mode (one inverted list x one group of query slots, bins laid out [item][slot][bin], third minimum kept); G: rows per
// select group (must match bit 2 of QueryBatchInfo.i8_mode, which the select and refine kernels read).
// DBG (diagnostic build, -DVDB_ABLATIONS only): in-kernel cycle stamps around the phases of every stage (results stay
// exact; the stamped kernel is ~10 % slower) -- scripts/stamp_scan_i8.py.
// TB (pacing barriers, 0 = none): an s_barrier every TB tiles inside a stage.  The stamps of the unpaced kernel
// (profiles/r03_stamps_scan_i8.txt) show the early half of a workgroup running its 8 tiles in ~60 % of the stage and then
// waiting at the stage barrier, while the late half -- which loses the pipe arbitration while both are active -- works
// through alternating MFMA and select phases ALONE, i.e. with the matrix pipe idle during its selects.  A bare s_barrier
// (no memory wait) per tile keeps the two waves of a SIMD in anti-phase: MFMA(t) of one beside select of the other.
// RING: LDS stages of the staging ring.  2 = double buffer, one stage in flight beyond the one being computed: right for
// the batch-shaped launch, where a stage carries thousands of MFMA cycles.  The serving-shaped launches (1 ... 4 waves per
// workgroup, one query tile) and the IVF work items stream their panels once, and with ONE stage in flight per workgroup
// the chip holds ~8 MB of requests: at ~2 us of HBM latency that is the 3.5 - 4 TB/s those scans ran at (PMC: 58 % of
// the wave time parked in s_waitcnt).  RING > 2 keeps RING - 1 stages requested: the per-row accumulator inits travel by
// LDS-DMA as well (no register staging), every wave issues the same number of requests per stage, and a stage is
// awaited with s_waitcnt vmcnt((RING - 2) x requests per stage) + s_barrier instead of the vmcnt(0) of __syncthreads().
template <int KS, int ST, int CB, int NWAVES = 8, int BT = 16, bool ITEMS = false, int G = 8, bool DBG = false, int TB = 0,
          int RING = 2, int AUX = 0>
__global__ __launch_bounds__(NWAVES * 64, (NWAVES > 8 ? NWAVES / 4 : (ITEMS && NWAVES == 4 && RING <= 4) ? 3 : NWAVES >= 4 ? 2 : 1)) void scan_i8_kernel(ScanI8Args a) {
    constexpr int GPT = 16 / G;                           // groups per (tile, column block): quads 4, octs 2
    constexpr int NT = NWAVES * 64;
    constexpr int kStageVec = ST * KS * 64;               // 16-byte vectors per stage
    constexpr int kBiasLoads = (ST * 32 + NT - 1) / NT;
    constexpr int TPS = ITEMS ? kIvfTilesPerSpan : kTilesPerSpan;   // tiles per span (IVF panel space: smaller spans, common.hpp)
    constexpr int kSpanR = TPS * 32, kHalfR = TPS * 16;   // rows per span / per lane half of a span
    constexpr int SPS = TPS / ST;
    constexpr int BPS = TPS / BT;                         // level-1 bins per (span, lane half)
    // tiles of a stage unrolled together: the whole stage for the baseline shape; wider shapes keep the tile loop
    // rolled (fully unrolled, hipcc keeps several tiles' fragments and bias registers alive and spills hundreds of VGPRs)
    constexpr int UNR = (ST * CB <= 8) ? ST : 1;
    static_assert(TPS % ST == 0 && ST >= 2 && TPS % BT == 0 && BT % ST == 0, "bad geometry");
    static_assert(!ITEMS || CB == 2, "items mode: 64 slots per wave");
    static_assert(RING >= 2 && RING <= 8 && (RING == 2 || (TB == 0 && ST % 2 == 0)), "bad ring");
    constexpr bool kDeep = RING > 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[RING * (kStageVec * 16 + ST * 32 * 4)];
    const int mode = a.info->i8_mode;
    if (!mode) return;                                    // this batch is served by the fp16 scan
    auto lds_a = [&](int buf) { return reinterpret_cast<int4v *>(smem + buf * (kStageVec * 16)); };
    auto lds_b = [&](int buf) { return reinterpret_cast<int *>(smem + RING * kStageVec * 16 + buf * (ST * 32 * 4)); };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const bool late = (NWAVES >= 2) && (wave >= NWAVES / 2);
    int chunk = 0;
    int64_t q0, span0, span1, out_pitch, out_col;
    int64_t lspan0 = 0, lspans = 0;                         // items mode: first span / span count of the whole list (bin indexing)
    size_t bin_base = 0;
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
        q0 = (int64_t)a.item_slot0[it] + wave * 64;
        out_pitch = NWAVES * 64;
        out_col = wave * 64 + (lane & 31);
        bin_base = (size_t)a.item_bin0[it];
    } else {
        const int b = blockIdx.x;
        const int x = b & 7, j = b >> 3;
        const int ci = j / a.nqtiles, qt = j - ci * a.nqtiles;
        chunk = x + 8 * ci;
        if (chunk >= a.nchunks) return;
        q0 = (int64_t)qt * (NWAVES * 32 * CB) + wave * (32 * CB);
        span0 = chunk_span0(chunk, a.spans_per_chunk, a.chunk_rem);
        span1 = chunk_span0(chunk + 1, a.spans_per_chunk, a.chunk_rem);
        if (span1 > a.nspans) span1 = a.nspans;
        out_pitch = a.Qpad;
        out_col = q0 + (lane & 31);
    }
    const int32_t *bias = a.bias8 + ((mode & 3) == 1 ? 0 : a.Npad);

    int4v bq[CB][KS];
    int qslot[CB];
    if (ITEMS) {   // the slots' queries now, their rows after the first stage is on its way (as scan_kernel)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) qslot[cb] = a.slot_query[q0 + cb * 32 + (lane & 31)] - 1;      // (0 = padding slot)
    } else {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bq[cb][ks] = a.qpanels[((size_t)(q0 / 32 + cb) * KS + ks) * 64 + lane];
    }
    const int nstages = (int)(span1 - span0) * SPS;
    const int nb_item = 2 * ((((int)lspans * BPS) + 3) & ~3);    // ITEMS: bins per query slot in this item (two half-runs)
    const int INF = (int)kI8Inf;
    int m1[CB], m2[CB], m3[CB], M1[CB], M2[CB], Ms[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        m1[cb] = m2[cb] = m3[cb] = M1[cb] = M2[cb] = INF;
        Ms[cb] = 0;
    }

    constexpr int kPieces = kStageVec / 64;
    static_assert(kPieces % NWAVES == 0, "pieces must divide over the waves");
    constexpr int kBiasPieces = (ST / 2 + NWAVES - 1) / NWAVES;       // ring mode: bias requests per wave and stage
    int stage_b[kBiasLoads];
    auto stage_issue = [&](int st, int buf) {
        const int64_t span = span0 + st / SPS;
        const int sq = st % SPS;
        const int4v *src = a.panels + ((size_t)(span * TPS + sq * ST) * KS) * 64;
        int4v *dst = lds_a(buf);
#pragma unroll
        for (int i = 0; i < kPieces / NWAVES; ++i) {
            const int p = wave + i * NWAVES;
            const int4v *g = src + p * 64 + lane;
            __builtin_amdgcn_global_load_lds(
                reinterpret_cast<const __attribute__((address_space(1))) void *>(reinterpret_cast<uintptr_t>(g)),
                reinterpret_cast<__attribute__((address_space(3))) void *>(
                    static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dst + p * 64))),
                16, 0, AUX);       // (AUX = 2: non-temporal -- scan copies larger than the Infinity Cache, read once per search)
        }
        if (kDeep) {        // accumulator inits of the stage: ST / 2 pieces of 64 ints (two tiles each), by LDS-DMA too; every
                            // wave issues kBiasPieces of them (a piece requested twice lands twice with the same bytes)
#pragma unroll
            for (int i = 0; i < kBiasPieces; ++i) {
                const int j = (wave + i * NWAVES) % (ST / 2);
                const int t = 2 * j + (lane >> 5), hh = (lane >> 4) & 1, r = lane & 15;
                const int32_t *g = bias + span * kSpanR + hh * kHalfR + (sq * ST + t) * 16 + r;
                __builtin_amdgcn_global_load_lds(
                    reinterpret_cast<const __attribute__((address_space(1))) void *>(reinterpret_cast<uintptr_t>(g)),
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
                const int t = e >> 5, hh = (e >> 4) & 1, r = e & 15;
                stage_b[i] = bias[span * kSpanR + hh * kHalfR + (sq * ST + t) * 16 + r];
            }
        }
    };
    auto stage_bias_store = [&](int buf) {
        if (kDeep) return;
#pragma unroll
        for (int i = 0; i < kBiasLoads; ++i)
            if (tid + i * NT < ST * 32) lds_b(buf)[tid + i * NT] = stage_b[i];
    };
    // ring mode: stage `st` (clamped to the last one: past the end the last stage is requested again into a slot nobody
    // reads, so that every stage boundary has the same number of requests behind it) and the wait for the OLDEST stage in
    // flight.  vmcnt counts this wave's requests in issue order; other requests in between (bin stores, the items-mode
    // gathers) only make the wait more conservative.
    auto ring_issue = [&](int st) { stage_issue(st < nstages ? st : nstages - 1, st % RING); };
    auto ring_wait = [&]() {
        constexpr int kKeep = (RING - 2) * (kPieces / NWAVES + kBiasPieces);
        static_assert(kKeep < 64, "vmcnt is a 6-bit counter");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kKeep) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // ITEMS: the bins of a lane's query slot form one run per lane half, stored as ONE 16-byte vector per array every fourth
    // bin (see scan_kernel: [item][slot][half][span][bin of the half], nb_half rounded up to 4)
    int4v pend1[CB], pend2[CB], pend3[CB];
    const int nb_half = ITEMS ? (((int)lspans * BPS + 3) & ~3) : 0;
    auto items_push = [&](int r) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            pend1[cb] = int4v{pend1[cb].y, pend1[cb].z, pend1[cb].w, m1[cb]};
            pend2[cb] = int4v{pend2[cb].y, pend2[cb].z, pend2[cb].w, m2[cb]};
            pend3[cb] = int4v{pend3[cb].y, pend3[cb].z, pend3[cb].w, m3[cb]};
            m1[cb] = INF;
            m2[cb] = INF;
            m3[cb] = INF;
        }
        if ((r & 3) == 3) {
            const size_t o4 = bin_base * out_pitch + (size_t)out_col * nb_item + (size_t)(h * nb_half + r - 3);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                *reinterpret_cast<int4v *>(a.bin_m1 + o4 + (size_t)cb * 32 * nb_item) = pend1[cb];
                *reinterpret_cast<int4v *>(a.bin_m2 + o4 + (size_t)cb * 32 * nb_item) = pend2[cb];
                *reinterpret_cast<int4v *>(a.bin_m3 + o4 + (size_t)cb * 32 * nb_item) = pend3[cb];
            }
        }
    };
    // a level-1 bin (BT tiles per lane half) is complete.  flat: [bin][query]
    auto flush_bin = [&](int64_t span, int bt) {
        if (ITEMS) {
            items_push((int)(span - lspan0) * BPS + bt);
            return;
        }
        const size_t o = (size_t)((span * 2 + h) * BPS + bt) * out_pitch + out_col;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
#ifdef VDB_ABLATIONS
            if (!a.abl_no_bins)
#endif
            {
                // (non-temporal: 339 MB of bins per 10k-query batch on 1M rows are written once and read sparsely by the select,
                //  while the 128 MB of panels are re-read by every query tile from L2 / the Infinity Cache; A/B in alternating
                //  processes on one box: 9.19 - 9.28 -> 9.27 - 9.34 M QPS)
                __builtin_nontemporal_store(__int_as_float(m1[cb]), a.bin_m1 + o + cb * 32);
                __builtin_nontemporal_store(__int_as_float(m2[cb]), a.bin_m2 + o + cb * 32);
            }
            M2[cb] = imin(imed3(M1[cb], M2[cb], m1[cb]), m2[cb]);
            if (m1[cb] < M1[cb]) Ms[cb] = (int)span;
            M1[cb] = imin(M1[cb], m1[cb]);
            m1[cb] = INF;
            m2[cb] = INF;
        }
    };

    if (kDeep) {
#pragma unroll
        for (int s = 0; s < RING - 1; ++s) ring_issue(s);
    } else {
        stage_issue(0, 0);
    }
    if (ITEMS) {   // gather: lane (col = lane&31, k half = lane>>5) reads 16 bytes of its slot's int8 query row
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int qa = qslot[cb];
            const int4v *ra = reinterpret_cast<const int4v *>(a.qrows + (size_t)(qa < 0 ? 0 : qa) * (32 * KS)) + h;
            const int4v zero = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bq[cb][ks] = qa < 0 ? zero : ra[ks * 2];
        }
    }
    stage_bias_store(0);
    if (kDeep) ring_wait();
    else __syncthreads();

    bool idle_wave = !ITEMS && a.nq_valid > 0 && q0 >= a.nq_valid;
    if (ITEMS) {        // items mode: every slot of this wave is padding (the last item of a list)
        bool any = false;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) any = any || qslot[cb] >= 0;
        idle_wave = __ballot(any) == 0ull;
    }
    if (idle_wave) {   // every query column of this wave is padding: keep staging + barriers going
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
            if (TB > 0) {
                constexpr int kTb = TB > 0 ? TB : 1;
                for (int t = 0; t + 1 < ST; ++t)
                    if ((t + 1) % kTb == 0) __builtin_amdgcn_s_barrier();
            }
            __syncthreads();
        }
        return;
    }

    int4v fr[KS];
    int16v cin, acc[CB];
    // DBG: cycle sums per phase; tick(x) charges the cycles since the previous stamp to x.  `done` makes the MFMA results
    // "used" so that the stamp behind an MFMA phase waits for the matrix pipe, not just for the issue.
    unsigned long long c_head = 0, c_mfma = 0, c_sel = 0, c_tail = 0, c_bar = 0, t_last = 0, t_first = 0, r_first = 0;
#define tick(bucket)                                   \
    do {                                               \
        if (DBG) {                                     \
            const unsigned long long t__ = stamp();    \
            bucket += t__ - t_last;                    \
            t_last = t__;                              \
        }                                              \
    } while (0)
#define done()                          \
    do {                                \
        if (DBG) { _Pragma("unroll") for (int cb__ = 0; cb__ < CB; ++cb__) wait_for_mfma(acc[cb__]); } \
    } while (0)
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
            const int4v *B4 = reinterpret_cast<const int4v *>(lds_b(buf)) + h * 4;
            const int ts0 = (st % SPS) * ST;
            read_phase_i8<KS>(A, B4, fr, cin, lane);
            tick(c_head);
#pragma unroll UNR
            for (int t = 0; t < ST; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                mfma_phase_i8<KS, CB>(fr, bq, cin, acc);
                __builtin_amdgcn_sched_barrier(0);
                done();
                tick(c_mfma);
                if (t + 1 < ST) read_phase_i8<KS>(A + (t + 1) * KS * 64, B4 + (t + 1) * 8, fr, cin, lane);
                select_phase_i8<CB, ITEMS, G>(acc, m1, m2, (unsigned)(((ts0 + t) % BT) * GPT), m3);
                tick(c_sel);
                if (TB > 0 && (t + 1) % (TB > 0 ? TB : 1) == 0 && t + 1 < ST) {
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    tick(c_bar);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (((ts0 + ST) % BT) == 0) flush_bin(span0 + st / SPS, (ts0 + ST) / BT - 1);
            if (!kDeep && st + 1 < nstages) stage_bias_store(buf ^ 1);
            tick(c_tail);
            if (kDeep) ring_wait();
            else __syncthreads();
            tick(c_bar);
        }
    } else {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = (int)(kI8Inf >> 6);   // dummy "previous tile": (x << 6) == "+inf", never wins
        for (int st = 0; st < nstages; ++st) {
            const int buf = kDeep ? st % RING : st & 1;
            if (kDeep) ring_issue(st + RING - 1);
            else if (st + 1 < nstages) stage_issue(st + 1, buf ^ 1);
            const int4v *A = lds_a(buf);
            const int4v *B4 = reinterpret_cast<const int4v *>(lds_b(buf)) + h * 4;
            const int ts0 = (st % SPS) * ST;
            tick(c_head);
#pragma unroll UNR
            for (int t = 0; t < ST; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                read_phase_i8<KS>(A + t * KS * 64, B4 + t * 8, fr, cin, lane);
                const int tp = (ts0 + t + TPS - 1) % TPS;
                select_phase_i8<CB, ITEMS, G>(acc, m1, m2, (unsigned)((tp % BT) * GPT), m3);
                if (t == 0 && st > 0 && (ts0 % BT) == 0) flush_bin(span0 + (st * ST - 1) / TPS, tp / BT);
                __builtin_amdgcn_sched_barrier(0);
                tick(c_sel);
                mfma_phase_i8<KS, CB>(fr, bq, cin, acc);
                done();
                tick(c_mfma);
                if (TB > 0 && (t + 1) % (TB > 0 ? TB : 1) == 0 && t + 1 < ST) {
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    tick(c_bar);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!kDeep && st + 1 < nstages) stage_bias_store(buf ^ 1);
            tick(c_tail);
            if (kDeep) ring_wait();
            else __syncthreads();
            tick(c_bar);
        }
        select_phase_i8<CB, ITEMS, G>(acc, m1, m2, (unsigned)((BT - 1) * GPT), m3);
        flush_bin(span1 - 1, BPS - 1);
    }
    if (ITEMS)       // fill the last vector of this part's run
        for (int r = (int)(span1 - lspan0) * BPS; r & 3; ++r) items_push(r);
    if (DBG && a.dbg && lane == 0) {
        const unsigned long long t_end = stamp(), r_end = realtime_ticks();
        unsigned long long *d = a.dbg + ((size_t)blockIdx.x * NWAVES + wave) * 8;
        d[0] = c_head; d[1] = c_mfma; d[2] = c_sel; d[3] = c_tail; d[4] = c_bar; d[5] = t_end - t_first;
        d[6] = r_end - r_first; d[7] = ((unsigned long long)nstages << 1) | (late ? 1ull : 0ull);
    }
#undef tick
#undef done
    if (ITEMS) return;

    const size_t so = (size_t)(chunk * 2 + h) * a.Qpad + q0 + (lane & 31);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        a.sb_m1[so + cb * 32] = __int_as_float(M1[cb]);
        a.sb_m2[so + cb * 32] = __int_as_float(M2[cb]);
        a.sb_span[so + cb * 32] = Ms[cb];
    }
}

// ---- IVF build / per-batch helpers for the int8 copy --------------------------------------------------------------
// int8 panels over the list-padded panel space (ivf_mfma.hpp: span_row0 / span_valid map a panel span to its rows)
__global__ __launch_bounds__(256) void ivf_build_panels_i8_kernel(const float *__restrict__ X, int D, int D4, int ks32,
                                                                  int64_t ntiles, int cx,
                                                                  const int32_t *__restrict__ span_row0,
                                                                  const int32_t *__restrict__ span_valid,
                                                                  int4v *__restrict__ panels) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(gid & 63);
    const int64_t tk = gid >> 6;
    const int ks = (int)(tk % ks32);
    const int64_t tile = tk / ks32;
    if (tile >= ntiles) return;
    const int rho = lane & 31, kh = lane >> 5;
    const int r = (rho & 3) | ((rho >> 3) << 2), h = (rho >> 2) & 1;
    const int64_t span = tile / kIvfTilesPerSpan;
    const int t = (int)(tile - span * kIvfTilesPerSpan);
    const int local = h * (kIvfSpanRows / 2) + t * 16 + r;
    const bool valid = local < span_valid[span];
    const int64_t row = (int64_t)span_row0[span] + local;
    const int d0 = ks * 32 + kh * 16;
    int4v out;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        unsigned word = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int d = d0 + 4 * w + b;
            int v = 0;
            if (valid && d < D) v = (int)X[(size_t)row * D4 + d] - cx;
            word |= ((unsigned)v & 0xffu) << (8 * b);
        }
        out[w] = (int)word;
    }
    panels[gid] = out;
}

// bias8[w][panel row] over the panel space (P spans of kIvfSpanRows rows); padding rows get kI8PadBias
__global__ __launch_bounds__(256) void ivf_build_bias_i8_kernel(const float *__restrict__ X, int64_t nspans, int D, int D4,
                                                                int metric, const int32_t *__restrict__ span_row0,
                                                                const int32_t *__restrict__ span_valid,
                                                                int32_t *__restrict__ bias8, int *__restrict__ rowstat) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nspans * kIvfSpanRows;
    if (i >= total) return;
    const int64_t span = i / kIvfSpanRows;
    const int local = (int)(i - span * kIvfSpanRows);
    if (local >= span_valid[span]) {
        bias8[i] = kI8PadBias;
        bias8[total + i] = kI8PadBias;
        return;
    }
    const int64_t row = (int64_t)span_row0[span] + local;
    long long n2 = 0, s1 = 0;
    for (int d = 0; d < D; ++d) {
        const long long v = (long long)X[(size_t)row * D4 + d];
        n2 += v * v;
        s1 += v;
    }
    rowstat[2 * row] = (int)n2;
    rowstat[2 * row + 1] = (int)s1;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const long long cq = w == 0 ? 127 : -1;
        const long long b = metric == 0 ? ((n2 - 2 * cq * s1) >> 1) : -cq * s1;
        bias8[(size_t)w * total + i] = (int32_t)(b + kI8Offset);
    }
}

// int8 copy of the query rows, [nq][32*ks32] = cq - q; the items-mode scan gathers its B fragments from it
__device__ __forceinline__ void qrows_i8_body(int64_t i, const float *__restrict__ Q, int64_t nq, int D, int Dpad,
                                              const QueryBatchInfo *__restrict__ info, signed char *__restrict__ qrows) {
    const int mode = info->i8_mode;
    if (i >= nq * Dpad || !mode) return;
    const int64_t q = i / Dpad;
    const int d = (int)(i - q * Dpad);
    const int cq = (mode & 3) == 1 ? 127 : -1;
    qrows[i] = (signed char)(d < D ? cq - (int)Q[(size_t)q * D + d] : 0);
}

__global__ __launch_bounds__(256) void ivf_qrows_i8_kernel(const float *__restrict__ Q, int64_t nq, int D, int Dpad,
                                                           const QueryBatchInfo *__restrict__ info,
                                                           signed char *__restrict__ qrows) {
    qrows_i8_body((int64_t)blockIdx.x * blockDim.x + threadIdx.x, Q, nq, D, Dpad, info, qrows);
}

// ONE dispatch for the per-batch query operands of the flat scan (each of the four kernels it replaces is a ~4.5 us
// dispatch of its own): workgroups [0, nA) build the fp16 B fragments, [nA, nB) the int8 B fragments, [nB, nC) the
// row-major int8 query rows for the list refine, [nC, nD) the per-query error bounds.
struct QueryPrepArgs {
    const float *Q;
    int64_t nq, nqtiles;        // nqtiles: 32-query blocks (Qpad / 32)
    int D, D4, ksteps, ks32, pitch8;
    const QueryBatchInfo *info;
    half8 *qpanels;
    int4v *qpanels8;            // nullptr: no int8 scan offered
    int x16;                    // B fragments (fp16 and int8) in the 16-query form of layout "x16" (scan_x16.hpp, scan_i8x16.hpp)
    signed char *qrows8;        // nullptr: no int8 refine rows
    EpsArgs eps;
    unsigned nA, nB, nC;        // region boundaries in workgroups
    // small batches (nq * D <= kFusedStatsMax values: serving-shaped calls, whose query_stats_kernel is ONE workgroup
    // anyway): every workgroup of this kernel takes the batch statistics itself (<= 64 KB from L2) and fixes the
    // scales in LDS; workgroup 0 also writes them to `info_out` for the kernels behind it.  One dependent dispatch less.
    int fused_stats;
    FinalizeArgs fin;
    QueryBatchInfo *info_out;
    unsigned *clear_small;      // fused_stats only: ws.small (kSmallBytes), cleared by workgroup 0 around the QueryBatchInfo at kInfoOffset
                                // -- the kernels that count into it (select, refine) run behind this one
};
constexpr int64_t kFusedStatsMax = 4096;    // (measured on 1M x 128: 1 / 8 queries 74.8 -> 70.7 / 78.0 -> 75.0 us; at 64 queries = 8192 values the redundant pass costs what the dispatch saved)
__global__ __launch_bounds__(256) void query_prep_kernel(QueryPrepArgs a) {
    const unsigned b = blockIdx.x;
    __shared__ QueryBatchInfo s_info;
    __shared__ float s_max[4];
    __shared__ int s_flags[4];
    if (a.clear_small && b == 0)
        for (unsigned i = threadIdx.x; i < kSmallBytes / 4; i += 256)
            if (i < kInfoOffset / 4 || i >= 16) a.clear_small[i] = 0u;          // (words [4, 16) = the QueryBatchInfo written below)
    if (a.fused_stats) {
        const int64_t total = a.nq * a.D;
        float amax = 0.f;
        int flags = 0;  // as query_stats_kernel: bit0 non-finite, bit1 non-integer, bit2 outside 0..255, bit3 outside -128..127
#pragma unroll 4
        for (int64_t i = threadIdx.x; i < total; i += 256) {
            const float v = a.Q[i];
            amax = fmaxf(amax, fabsf(v));
            flags |= (!(fabsf(v) <= 3.402823466e+38f)) | ((v != rintf(v)) << 1) | ((!(v >= 0.f && v <= 255.f)) << 2) |
                     ((!(v >= -128.f && v <= 127.f)) << 3);
        }
        for (int o = 32; o > 0; o >>= 1) {
            amax = fmaxf(amax, __shfl_xor(amax, o));
            flags |= __shfl_xor(flags, o);
        }
        if ((threadIdx.x & 63) == 0) {
            s_max[threadIdx.x >> 6] = amax;
            s_flags[threadIdx.x >> 6] = flags;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            amax = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
            flags = s_flags[0] | s_flags[1] | s_flags[2] | s_flags[3];
            QueryBatchInfo v{};
            v.absmax_bits = __float_as_uint(fabsf(amax));
            v.nonfinite = flags & 1;
            v.not_integer = (flags >> 1) & 1;
            v.not_u8 = (flags & 6) ? 1 : 0;
            v.not_s8 = (flags & 10) ? 1 : 0;
            query_finalize_values(&v, a.fin, __uint_as_float(v.absmax_bits), v.not_integer, v.nonfinite, v.not_u8, v.not_s8);
            s_info = v;
            if (b == 0) *a.info_out = v;
        }
        __syncthreads();
        a.info = &s_info;
        a.eps.info = &s_info;
    }
    if (b < a.nA)
        build_qpanels_body((int64_t)b * 256 + threadIdx.x, a.Q, a.nq, a.D, a.D4, a.ksteps, a.nqtiles, a.info, a.qpanels, a.x16);
    else if (b < a.nB)
        build_qpanels_i8_body((int64_t)(b - a.nA) * 256 + threadIdx.x, a.Q, a.nq, a.D, a.ks32, a.nqtiles, a.info, a.qpanels8, a.x16);
    else if (b < a.nC)
        qrows_i8_body((int64_t)(b - a.nB) * 256 + threadIdx.x, a.Q, a.nq, a.D, a.pitch8, a.info, a.qrows8);
    else
        query_eps_body((int64_t)(b - a.nC) * 256 + threadIdx.x, a.eps);
}

}  // namespace vdb
