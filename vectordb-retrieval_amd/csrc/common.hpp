// common.hpp -- shared declarations of libvdbhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <stdexcept>
#include <string>

namespace vdb {

// ---- geometry of the fp16 scan copy ("panels") ---------------------------------------------
// The corpus copy that feeds v_mfma_f32_32x32x16_f16 is stored in HBM in A-fragment order:
//   panels[tile][kstep][lane][8 x fp16]   (1 KiB per (tile,kstep): one wave-wide 16-B load)
// tile = 32 MFMA rows; kstep = 16 dims; lane l holds MFMA row (l & 31), dims 16*kstep + 8*(l>>5) + 0..7.
// MFMA row rho of tile t in span s maps to corpus row  512*s + 256*h + 16*t + r  with
//   r = (rho & 3) | ((rho >> 3) << 2),  h = (rho >> 2) & 1
// which is exactly the (accumulator register r, lane half h) that sees that row in the C/D layout
//   row = (r & 3) + 8*(r >> 2) + 4*h.  So one lane half walks 256 CONSECUTIVE corpus rows per span:
// a "bin" (the unit of the collision guard) is a contiguous 256-row range and the 8-bit id packed
// into a score is the offset inside the bin.
constexpr int kTileRows = 32;
constexpr int kSpanRows = 512;      // 16 tiles; two bins (h = 0,1) of 256 rows
constexpr int kBinRows = 256;
constexpr int kTilesPerSpan = 16;
// layout "p16" (16-row tiles for v_mfma_f32_16x16x32_f16): span = 64 tiles = 1024 rows = four 256-row bins, one
// per lane group g = lane>>4 of the 16x16 C/D layout
constexpr int kTileRows16 = 16;
constexpr int kSpanRows16 = 1024;
constexpr int kTilesPerSpan16 = 64;
// IVF panel space (D <= 128): every inverted list is padded to whole spans, so its spans are SMALLER -- 8 tiles = 256 rows,
// each lane half walking 128 consecutive rows: k-means lists of ~750 rows then carry 2 % of padding instead of 36 %
// (512-row spans padded the median list of the SIFT1M / nlist 1024 index from 752 to 1024 rows).  Same row mapping inside a
// span as the flat layout with 16 -> kIvfTilesPerSpan tiles.
constexpr int kIvfTilesPerSpan = 8;
constexpr int kIvfSpanRows = kIvfTilesPerSpan * kTileRows;
constexpr int kStageTiles = 4;      // tiles per LDS stage (128 MFMA rows)
constexpr int kMaxKSteps = 8;       // register-resident query fragments: D <= 128
constexpr float kPadBias = 1.0e38f; // accumulator init of padding rows (scan units): never selected

// Search statistics (candidates, re-scans, fallback queries) are counted with global atomics by every query; on ONE
// address 20k adds per batch cost ~80 us (same-line atomics serialise in L2).  The counters are therefore kept in
// kStatShards replicas, each on a 128-byte line of its own, picked by the query number; the host sums them.
constexpr int kStatShards = 32;
constexpr int kStatStride = 16;   // unsigned long long per shard (= 128 bytes)
__device__ __forceinline__ void stat_add(unsigned long long *counters, long long q, int idx, unsigned long long v) {
    atomicAdd(&counters[(size_t)(q & (kStatShards - 1)) * kStatStride + idx], v);
}
constexpr size_t kSmallBytes = 64 + (size_t)kStatShards * kStatStride * 8;   // ws.small: fb_count + sharded counters
// ws.small layout: [0,4) fb_count | [16,64) QueryBatchInfo of the current batch | [64, ...) sharded statistics.  The
// first 64 bytes are cleared with ONE memset per batch (each memset is a ~4 us dispatch of its own).
constexpr size_t kInfoOffset = 16;

struct IndexStats {       // filled on device by the corpus-prep kernels
    unsigned absmax_bits; // bits of max |x|
    unsigned maxnorm2_bits; // bits of max ||x||^2 (float)
    int nonfinite;        // any NaN/Inf
    int not_integer;      // any non-integer value
    int not_fp16_exact;   // any value changed by the scaled fp16 conversion
    int not_u8;           // any value outside the integers 0..255      (int8 scan copy, window "u8": stored as x - 128)
    int not_s8;           // any value outside the integers -128..127   (window "s8": stored as is)
    int pad[1];
};

struct QueryBatchInfo {   // per search call, device resident
    unsigned absmax_bits;
    int not_integer;
    int not_fp16_exact;
    int nonfinite;
    float sq;             // power-of-two query scale
    float cs;             // sq * sx : scale of the scan scores
    float bscale;         // factor applied to the B operand (-2*sq for L2, -sq for IP)
    int force_fallback;   // scales unusable (non-finite input / overflow): every query takes the exhaustive path
    int not_u8, not_s8;   // query values outside the integers 0..255 / -128..127
    int i8_mode;          // 0: fp16 scan.  Else int8 scan (scan_i8.hpp): bits 0-1 = query window (1 u8, 2 s8), bit 2 = the
                          // select works on groups of 8 rows (octs) instead of 4 (quads)
    unsigned done_blocks; // workgroups of query_stats_kernel that have contributed (the last one finalises the scales)
};
static_assert(sizeof(QueryBatchInfo) + 16 <= 64, "QueryBatchInfo must fit the first 64 bytes of ws.small");

// ---- int8 scan copy (scan_i8.hpp) ---------------------------------------------------------------------------------
// Integer corpora whose values fit one byte (SIFT descriptors are uint8) get a second scan copy for
// v_mfma_i32_32x32x32_i8 (twice the fp16 MFMA rate per clock, exact int32 accumulation).  Operands: A = x - cx,
// B = cq - q with (cx, cq) = (128, 127) in the u8 window and (0, -1) in the s8 window, so both fit [-128, 127].
//   sum_d A*B = -(x.q) + cq*sum(x) + cx*sum(q) - D*cx*cq
// L2 order key ||x||^2 - 2 x.q = 2*dot + (||x||^2 - 2*cq*sum(x)) + const(q); the accumulator starts from
// floor(bias/2) + kI8Offset and holds t' = dot + floor(bias/2) + kI8Offset, i.e. half the key up to the dropped
// parity bit (|error| <= 1/2 in t units: the select widens its threshold by one).  IP: -x.q = dot - cq*sum(x) + const.
// The offset makes every t' positive, the select packs v = (t' << 6) | quad id, and v read as a float bit pattern
// is a normal positive float whose order is the integer order -- the bin arrays and the select kernels are shared
// with the fp16 path (only T^ = tau + 2 eps is formed in integer arithmetic).
// A packed key names a GROUP of consecutive rows inside its bin: low 6 bits = group id, first row = id * group rows.
// fp16 scan: quads (4 rows).  int8 scan: octs (8 rows) by default -- the integer select is VALU-issue bound beside the
// double-rate MFMAs and an oct costs 7 ops per 8 scores against 5 per 4, while its exact re-scoring reads the 128-byte
// int8 rows (refine.hpp, X8), so twice the candidate rows still gather half the bytes of quads of float32 rows.
__host__ __device__ inline int group_rows_of(int i8_mode) { return (i8_mode & 4) ? 8 : 4; }
__host__ __device__ inline int cand_row_offset(unsigned packed_bits, int i8_mode) {
    return (int)(packed_bits & 0x3Fu) * group_rows_of(i8_mode);
}
constexpr int kI8Offset = 1 << 24;
constexpr int kI8PadBias = 30000000;          // accumulator init of padding rows: above every real t', (t' << 6) < +inf bits
constexpr unsigned kI8Inf = 0x7f800000u;      // "+inf" of the packed integer keys (bits of float +inf)

// work list entry of the refine kernel: rows [row0, row0+count)
struct Range {
    int64_t row0;
    int32_t count;
    int32_t pad;
};

// ---- order keys --------------------------------------------------------------------------------
__host__ __device__ inline uint64_t sortable_u64(double v) {
    uint64_t u = *reinterpret_cast<uint64_t *>(&v);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__host__ __device__ inline double unsortable_f64(uint64_t k) {
    uint64_t u = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return *reinterpret_cast<double *>(&u);
}
__host__ __device__ inline unsigned sortable_u32(float v) {
    unsigned u = *reinterpret_cast<unsigned *>(&v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float unsortable_f32(unsigned k) {
    unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return *reinterpret_cast<float *>(&u);
}

// ---- host-side error plumbing -------------------------------------------------------------------
struct Error : public std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define VDB_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            throw ::vdb::Error(3, std::string(#expr) + " failed: " + hipGetErrorString(e__));      \
    } while (0)

}  // namespace vdb
