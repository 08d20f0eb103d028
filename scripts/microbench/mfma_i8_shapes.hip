// Microbenchmark (round 3): the two int8 MFMA shapes of gfx950 under the scan's instruction mix, with the IN-KERNEL CLOCK.
//
//   v_mfma_i32_32x32x32_i8  (what scan_i8_kernel uses)   vs   v_mfma_i32_16x16x64_i8  (p16 panels, never built for the scan)
//
// Both loops re-read their A fragments from LDS by ds_read_b128 (same LDS bytes per op), keep the B fragments (queries)
// in registers, D = 128, and carry the integer select epilogue of scan_i8.hpp (quads or octs).  Same output tile per
// wave: 64 or 128 queries.  For every variant the kernel stamps s_memtime / s_memrealtime around its loop
// (MI355X_MICROARCH.md, DVFS give-back item 6): TOP/s = ops per cycle x the clock the chip holds under that loop, and the
// two factors are reported separately -- "pipe" = MFMA cycles / wave cycles per SIMD.
//   hipcc -O3 --offload-arch=gfx950 mfma_i8_shapes.hip -o mfma_i8_shapes
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int kLdsVec = 4096;   // 64 KiB of A fragments resident in LDS

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imed3(int a, int b, int c) { return imax(imin(a, b), imin(imax(a, b), c)); }
__device__ __forceinline__ void fold(int q, unsigned id, int &m1, int &m2) {
    const int v = (int)(((unsigned)q << 6) | id);
    m2 = imed3(m1, m2, v);
    m1 = imin(m1, v);
}

struct Stamp { unsigned long long cyc, real; };
__device__ __forceinline__ Stamp stamp_now() {
    Stamp s;
    __builtin_amdgcn_sched_barrier(0);
    s.cyc = __builtin_amdgcn_s_memtime();
    s.real = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return s;
}

// ---- 32x32x32: CB column blocks of 32 queries; a tile = 32 rows x 128 dims = 4 fragments of 1 KiB -------------------
template <int CB, int WPS, int EPI, int G>
__global__ __launch_bounds__(256 * WPS, WPS) void loop32(const int4v *A, const int4v *B, int *out, unsigned long long *clk, int iters) {
    constexpr int NT = 256 * WPS, kTiles = kLdsVec / 256;
    __shared__ int4v lds[kLdsVec];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kLdsVec; i += NT) lds[i] = A[i];
    int4v b[CB][4];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b[cb][ks] = B[(((wave * CB + cb) * 4 + ks) % 128) * 64 + lane];
    __syncthreads();
    int m1[CB], m2[CB], sum = 0;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) m1[cb] = m2[cb] = 0x7fffffff;
    const Stamp s0 = stamp_now();
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int t = 0; t < kTiles; ++t) {
            int16v acc[CB];
            const int4v *bp = lds + ((t * 37 + (lane >> 5) * 4) & 1023);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int4v c = bp[g];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    acc[cb][4 * g] = c.x; acc[cb][4 * g + 1] = c.y; acc[cb][4 * g + 2] = c.z; acc[cb][4 * g + 3] = c.w;
                }
            }
            const int4v *a = lds + t * 4 * 64 + lane;
            int4v f[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) f[ks] = a[ks * 64];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    acc[cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[ks], b[cb][ks], acc[cb], 0, 0, 0);
            if (EPI == 1) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int g = 0; g < 16 / G; ++g) {
                        int q = imin(imin(acc[cb][G * g], acc[cb][G * g + 1]), imin(acc[cb][G * g + 2], acc[cb][G * g + 3]));
                        if (G >= 8) q = imin(q, imin(imin(acc[cb][G * g + 4], acc[cb][G * g + 5]), imin(acc[cb][G * g + 6], acc[cb][G * g + 7])));
                        if (G == 16)     // round 4: one group per (tile, column block) -- 16 rows per candidate group
                            q = imin(q, imin(imin(imin(acc[cb][8], acc[cb][9]), imin(acc[cb][10], acc[cb][11])),
                                             imin(imin(acc[cb][12], acc[cb][13]), imin(acc[cb][14], acc[cb][15]))));
                        fold(q, (unsigned)((t * (16 / G) + g) & 63), m1[cb], m2[cb]);
                    }
            } else {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) sum += acc[cb][0] + acc[cb][7];
            }
        }
    }
    const Stamp s1 = stamp_now();
    int r = sum;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) r += m1[cb] + m2[cb];
    out[blockIdx.x * NT + tid] = r;
    if (lane == 0) {
        clk[((size_t)blockIdx.x * (NT / 64) + wave) * 2] = s1.cyc - s0.cyc;
        clk[((size_t)blockIdx.x * (NT / 64) + wave) * 2 + 1] = s1.real - s0.real;
    }
}

// ---- 32x32x32 with the scan's STAGING: the A tiles are not resident; every 8 tiles (32 KiB) a stage arrives by LDS-DMA
// (global_load_lds_dwordx4, 4 pieces of 1 KiB per wave, source L2-resident) into the other half of a double buffer,
// one workgroup barrier per stage.  SYNC 1: the barrier alone (tiles stay resident); SYNC 2: barrier + LDS-DMA;
// SYNC 3: as 2, and the two halves of the workgroup run MFMA / select in opposite order with sched_barrier fences
// (the production kernel's phase stagger).
template <int CB, int SYNC, int G>
__global__ __launch_bounds__(512, 2) void loop32s(const int4v *A, const int4v *B, int *out, unsigned long long *clk, int iters) {
    constexpr int NT = 512, ST = 8, kStage = ST * 4 * 64;       // vectors per stage
    __shared__ int4v lds[2 * kStage];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * kStage; i += NT) lds[i] = A[i % kLdsVec];
    int4v b[CB][4];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b[cb][ks] = B[(((wave * CB + cb) * 4 + ks) % 128) * 64 + lane];
    __syncthreads();
    int m1[CB], m2[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) m1[cb] = m2[cb] = 0x7fffffff;
    const bool late = SYNC == 3 && wave >= 4;
    int16v acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0x1fffffff;
    auto select = [&](int t) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int g = 0; g < 16 / G; ++g) {
                int q = imin(imin(acc[cb][G * g], acc[cb][G * g + 1]), imin(acc[cb][G * g + 2], acc[cb][G * g + 3]));
                if (G >= 8) q = imin(q, imin(imin(acc[cb][G * g + 4], acc[cb][G * g + 5]), imin(acc[cb][G * g + 6], acc[cb][G * g + 7])));
                if (G == 16)
                    q = imin(q, imin(imin(imin(acc[cb][8], acc[cb][9]), imin(acc[cb][10], acc[cb][11])),
                                     imin(imin(acc[cb][12], acc[cb][13]), imin(acc[cb][14], acc[cb][15]))));
                fold(q, (unsigned)((t * (16 / G) + g) & 63), m1[cb], m2[cb]);
            }
    };
    const Stamp s0 = stamp_now();
    const int nstage = iters * (kLdsVec / kStage);
    for (int st = 0; st < nstage; ++st) {
        const int buf = st & 1;
        if (SYNC >= 2) {
            const int4v *src = A + ((st + 1) % (kLdsVec / kStage)) * kStage;
            int4v *dst = lds + (buf ^ 1) * kStage;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = wave + i * 8;
                __builtin_amdgcn_global_load_lds(
                    reinterpret_cast<const __attribute__((address_space(1))) void *>(reinterpret_cast<uintptr_t>(src + p * 64 + lane)),
                    reinterpret_cast<__attribute__((address_space(3))) void *>(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dst + p * 64))),
                    16, 0, 0);
            }
        }
        const int4v *base = lds + (SYNC >= 2 ? buf * kStage : 0);
#pragma unroll 1
        for (int t = 0; t < ST; ++t) {
            const int4v *bp = base + ((t * 37 + (lane >> 5) * 4) & 1023);
            int16v cin;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int4v c = bp[g];
                cin[4 * g] = c.x; cin[4 * g + 1] = c.y; cin[4 * g + 2] = c.z; cin[4 * g + 3] = c.w;
            }
            const int4v *a = base + t * 4 * 64 + lane;
            int4v f[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) f[ks] = a[ks * 64];
            if (late) {            // retire the previous tile first, then this tile's MFMAs
                __builtin_amdgcn_sched_barrier(0);
                select(t);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    acc[cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[ks], b[cb][ks], ks == 0 ? cin : acc[cb], 0, 0, 0);
            if (!late) {
                if (SYNC == 3) __builtin_amdgcn_sched_barrier(0);
                select(t);
                if (SYNC == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (SYNC >= 1) __syncthreads();
    }
    const Stamp s1 = stamp_now();
    int r = 0;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) r += m1[cb] + m2[cb] + acc[cb][5];
    out[blockIdx.x * NT + tid] = r;
    if (lane == 0) {
        clk[((size_t)blockIdx.x * (NT / 64) + wave) * 2] = s1.cyc - s0.cyc;
        clk[((size_t)blockIdx.x * (NT / 64) + wave) * 2 + 1] = s1.real - s0.real;
    }
}

// ---- 16x16x64: CB column blocks of 16 queries; a tile = 16 rows x 128 dims = 2 fragments of 1 KiB --------------------
// C/D layout: lane (col = lane & 15, g = lane >> 4) holds rows 4g..4g+3 -> one quad per (tile, column block).
// G = 4: quad select per tile (2 + 3 VALU per 4 scores).  G = 8: octs over two consecutive tiles (7 VALU per 8 scores).
template <int CB, int WPS, int EPI, int G>
__global__ __launch_bounds__(256 * WPS, WPS) void loop16(const int4v *A, const int4v *B, int *out, unsigned long long *clk, int iters) {
    constexpr int NT = 256 * WPS, kTiles = kLdsVec / 128;
    __shared__ int4v lds[kLdsVec];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kLdsVec; i += NT) lds[i] = A[i];
    int4v b[CB][2];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) b[cb][ks] = B[(((wave * CB + cb) * 2 + ks) % 128) * 64 + lane];
    __syncthreads();
    int m1[CB], m2[CB], sum = 0;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) m1[cb] = m2[cb] = 0x7fffffff;
    const Stamp s0 = stamp_now();
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int t2 = 0; t2 < kTiles; t2 += 2) {
            int4v acc[2][CB];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = t2 + u;
                const int4v c = lds[((t * 37 + (lane >> 4)) & 1023)];      // the 4 biases of the lane's rows (broadcast)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc[u][cb] = c;
                const int4v *a = lds + t * 2 * 64 + lane;
                const int4v f0 = a[0], f1 = a[64];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc[u][cb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f0, b[cb][0], acc[u][cb], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc[u][cb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f1, b[cb][1], acc[u][cb], 0, 0, 0);
                if (EPI == 1 && G == 4) {
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) {
                        const int q = imin(imin(acc[u][cb][0], acc[u][cb][1]), imin(acc[u][cb][2], acc[u][cb][3]));
                        fold(q, (unsigned)(t & 63), m1[cb], m2[cb]);
                    }
                }
            }
            if (EPI == 1 && G == 8) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    int q = imin(imin(acc[0][cb][0], acc[0][cb][1]), imin(acc[0][cb][2], acc[0][cb][3]));
                    q = imin(q, imin(imin(acc[1][cb][0], acc[1][cb][1]), imin(acc[1][cb][2], acc[1][cb][3])));
                    fold(q, (unsigned)((t2 >> 1) & 63), m1[cb], m2[cb]);
                }
            }
            if (EPI == 0) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) sum += acc[0][cb][0] + acc[1][cb][3];
            }
        }
    }
    const Stamp s1 = stamp_now();
    int r = sum;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) r += m1[cb] + m2[cb];
    out[blockIdx.x * NT + tid] = r;
    if (lane == 0) {
        clk[((size_t)blockIdx.x * (NT / 64) + wave) * 2] = s1.cyc - s0.cyc;
        clk[((size_t)blockIdx.x * (NT / 64) + wave) * 2 + 1] = s1.real - s0.real;
    }
}

// ---- 16x16x64 with the scan's STAGING (round 4): a tile = 32 rows = two 16-row blocks x two 64-dim fragments, 8 column
// blocks of 16 queries per wave (128 queries, as loop32s<4>); one oct per (tile, column block) = the lane's 4 rows of both
// row blocks.  SYNC as loop32s.
template <int SYNC>
__global__ __launch_bounds__(512, 2) void loop16s(const int4v *A, const int4v *B, int *out, unsigned long long *clk, int iters) {
    constexpr int NT = 512, ST = 8, CB = 8, kStage = ST * 4 * 64;
    __shared__ int4v lds[2 * kStage];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * kStage; i += NT) lds[i] = A[i % kLdsVec];
    int4v b[CB][2];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) b[cb][ks] = B[(((wave * CB + cb) * 2 + ks) % 128) * 64 + lane];
    __syncthreads();
    int m1[CB], m2[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) m1[cb] = m2[cb] = 0x7fffffff;
    const bool late = SYNC == 3 && wave >= 4;
    int4v acc[2][CB];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[u][cb] = int4v{0x1fffffff, 0x1fffffff, 0x1fffffff, 0x1fffffff};
    auto select = [&](int t) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int t1 = imin(imin(acc[0][cb][0], acc[0][cb][1]), acc[0][cb][2]);
            const int t2 = imin(imin(acc[0][cb][3], acc[1][cb][0]), acc[1][cb][1]);
            const int q = imin(imin(imin(acc[1][cb][2], acc[1][cb][3]), t1), t2);
            fold(q, (unsigned)(t & 63), m1[cb], m2[cb]);
        }
    };
    const Stamp s0 = stamp_now();
    const int nstage = iters * (kLdsVec / kStage);
    for (int st = 0; st < nstage; ++st) {
        const int buf = st & 1;
        if (SYNC >= 2) {
            const int4v *src = A + ((st + 1) % (kLdsVec / kStage)) * kStage;
            int4v *dst = lds + (buf ^ 1) * kStage;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = wave + i * 8;
                __builtin_amdgcn_global_load_lds(
                    reinterpret_cast<const __attribute__((address_space(1))) void *>(reinterpret_cast<uintptr_t>(src + p * 64 + lane)),
                    reinterpret_cast<__attribute__((address_space(3))) void *>(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dst + p * 64))),
                    16, 0, 0);
            }
        }
        const int4v *base = lds + (SYNC >= 2 ? buf * kStage : 0);
#pragma unroll 1
        for (int t = 0; t < ST; ++t) {
            const int4v c0 = base[(t * 37 + (lane >> 4)) & 1023], c1 = base[(t * 37 + 4 + (lane >> 4)) & 1023];
            const int4v *a = base + t * 4 * 64 + lane;
            const int4v f00 = a[0], f01 = a[64], f10 = a[128], f11 = a[192];
            if (late) {
                __builtin_amdgcn_sched_barrier(0);
                select(t);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[0][cb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f00, b[cb][0], c0, 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[1][cb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f10, b[cb][0], c1, 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[0][cb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f01, b[cb][1], acc[0][cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[1][cb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f11, b[cb][1], acc[1][cb], 0, 0, 0);
            if (!late) {
                if (SYNC == 3) __builtin_amdgcn_sched_barrier(0);
                select(t);
                if (SYNC == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (SYNC >= 1) __syncthreads();
    }
    const Stamp s1 = stamp_now();
    int r = 0;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) r += m1[cb] + m2[cb] + acc[1][cb][2];
    out[blockIdx.x * NT + tid] = r;
    if (lane == 0) {
        clk[((size_t)blockIdx.x * (NT / 64) + wave) * 2] = s1.cyc - s0.cyc;
        clk[((size_t)blockIdx.x * (NT / 64) + wave) * 2 + 1] = s1.real - s0.real;
    }
}

struct Result { float ms; double cyc, ghz; };

template <class K>
Result run(K kernel, int threads, const int4v *dA, const int4v *dB, int *dO, unsigned long long *dC, int nblk, int iters,
           hipEvent_t e0, hipEvent_t e1) {
    Result best{1e30f, 0, 0};
    const size_t nw = (size_t)nblk * (threads / 64);
    std::vector<unsigned long long> h(nw * 2);
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        kernel<<<nblk, threads>>>(dA, dB, dO, dC, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2 && ms < best.ms) {
            CK(hipMemcpy(h.data(), dC, nw * 16, hipMemcpyDeviceToHost));
            std::vector<double> cyc(nw), ghz(nw);
            for (size_t i = 0; i < nw; ++i) { cyc[i] = (double)h[2 * i]; ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; }
            std::nth_element(cyc.begin(), cyc.begin() + nw / 2, cyc.end());
            std::nth_element(ghz.begin(), ghz.begin() + nw / 2, ghz.end());
            best = Result{ms, cyc[nw / 2], ghz[nw / 2]};
        }
    }
    return best;
}

int main() {
    const int nblk = 256 * 4, iters = 300;
    std::mt19937 rng(1);
    std::gamma_distribution<float> gm(0.6f, 40.f);
    auto val = [&]() -> int { return (int)std::min(218.f, std::floor(gm(rng))) - 128; };
    std::vector<int4v> hA(kLdsVec), hB(128 * 64);
    auto fill = [&](std::vector<int4v> &v) {
        for (auto &x : v) for (int j = 0; j < 4; ++j) {
            unsigned w = 0;
            for (int b = 0; b < 4; ++b) w |= ((unsigned)(val() & 0xff)) << (8 * b);
            x[j] = (int)w;
        }
    };
    fill(hA); fill(hB);
    int4v *dA, *dB; int *dO; unsigned long long *dC;
    CK(hipMalloc(&dA, hA.size() * 16)); CK(hipMalloc(&dB, hB.size() * 16)); CK(hipMalloc(&dO, (size_t)nblk * 512 * 4));
    CK(hipMalloc(&dC, (size_t)nblk * 8 * 16));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // per wave and pass over the 64 KiB of LDS: queries per wave x 512 rows x 128 dims x 2 ops; MFMA cycles per wave for
    // that: (ops / 65536) x 32 [32x32x32]  =  (ops / 32768) x 16 [16x16x64]  -- the same
    printf("%-10s %-8s %-5s %-16s %9s %9s %8s %8s\n", "shape", "q/wave", "w/simd", "epilogue", "ms", "TOP/s", "GHz", "pipe");
    for (int round = 0; round < 2; ++round) {
        auto report = [&](const char *shape, int qpw, int wps, const char *label, Result r) {
            if (!round) return;
            const double ops_wave = (double)iters * qpw * 512.0 * 128.0 * 2.0;
            const double tops = ops_wave * nblk * 4 * wps / r.ms / 1e9;
            const double mfma_cyc = ops_wave / 65536.0 * 32.0;
            printf("%-10s %-8d %-5d %-16s %9.3f %9.1f %8.3f %8.3f\n", shape, qpw, wps, label, r.ms, tops, r.ghz,
                   mfma_cyc * wps / r.cyc);
        };
#define R32(CB, WPS, EPI, G, label) report("32x32x32", 32 * CB, WPS, label, run(loop32<CB, WPS, EPI, G>, 256 * WPS, dA, dB, dO, dC, nblk, iters, e0, e1));
#define R16(CB, WPS, EPI, G, label) report("16x16x64", 16 * CB, WPS, label, run(loop16<CB, WPS, EPI, G>, 256 * WPS, dA, dB, dO, dC, nblk, iters, e0, e1));
        R32(2, 2, 0, 4, "bare")   R16(4, 2, 0, 4, "bare")
        R32(2, 2, 1, 4, "quads")  R16(4, 2, 1, 4, "quads")
        R32(2, 2, 1, 8, "octs")   R16(4, 2, 1, 8, "octs")
        R32(4, 2, 0, 4, "bare")   R16(8, 2, 0, 4, "bare")
        R32(4, 2, 1, 4, "quads")  R16(8, 2, 1, 4, "quads")
        R32(4, 2, 1, 8, "octs")   R16(8, 2, 1, 8, "octs")
        R32(4, 1, 0, 4, "bare")   R16(8, 1, 0, 4, "bare")
        R32(4, 1, 1, 8, "octs")   R16(8, 1, 1, 8, "octs")
        R32(4, 2, 1, 16, "hexadecs")      // round 4: what fewer select instructions per score are worth
#define R32S(SYNC, label) report("32x32x32", 128, 2, label, run(loop32s<4, SYNC, 8>, 512, dA, dB, dO, dC, nblk, iters, e0, e1));
        R32S(0, "octs, loop32s")
        R32S(1, "octs +barrier/8t")
        R32S(2, "octs +bar +DMA")
        R32S(3, "octs +bar+DMA+stag")
#define R32S16(SYNC, label) report("32x32x32", 128, 2, label, run(loop32s<4, SYNC, 16>, 512, dA, dB, dO, dC, nblk, iters, e0, e1));
        R32S16(0, "hexadecs, loop32s")
        R32S16(3, "hexa +bar+DMA+stag")
#define R16S(SYNC, label) report("16x16x64", 128, 2, label, run(loop16s<SYNC>, 512, dA, dB, dO, dC, nblk, iters, e0, e1));
        R16S(0, "octs, loop16s")
        R16S(2, "octs +bar +DMA")
        R16S(3, "octs +bar+DMA+stag")
        R32S(3, "octs +bar+DMA+stag")       // (again, next to the 16x16x64 rows: same thermal state)
    }
    return 0;
}
