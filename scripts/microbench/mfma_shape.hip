// Microbenchmark: LDS-fed fp16 MFMA loop, v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16, same flops, same
// LDS bytes per flop (A fragments re-read from LDS, B fragments resident in registers), 8 waves per workgroup,
// one workgroup per CU slot.  Answers: which shape delivers more FLOP/s under the chip's power management, on
// Gaussian data and on SIFT-like small integers.   hipcc -O3 --offload-arch=gfx950 mfma_shape.hip -o mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int kTiles = 8;   // 32-row units resident in LDS (8 x 8 KiB = 64 KiB)

// the scan's select epilogue: quad minimum, id tag in the low mantissa bits, running (min, second min)
__device__ __forceinline__ void quad(float a, float b, float c, float d, unsigned id, float &m1, float &m2) {
    float ninf = -__builtin_inff();
    asm volatile("" : "+v"(ninf));
    const float q = __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(a, b, ninf), __builtin_amdgcn_fmed3f(c, d, ninf), ninf);
    const float v = __uint_as_float((__float_as_uint(q) & 0xFFFFFFC0u) | id);
    m2 = __builtin_amdgcn_fmed3f(m1, m2, v);
    m1 = __builtin_amdgcn_fmed3f(m1, v, ninf);
}

template <int SHAPE, bool EPI, bool BIAS = false>
__global__ __launch_bounds__(512, 2) void loop_kernel(const half8 *A, const half8 *B, float *out, int iters) {
    __shared__ half8 lds[kTiles * 8 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kTiles * 8 * 64; i += 512) lds[i] = A[i];
    half8 b[16];                                    // 64 queries x 128 dims per wave = 64 VGPRs
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = B[(wave * 16 + i) * 64 + lane];
    __syncthreads();
    float sum = 0.f;
    float m1[4] = {1e30f, 1e30f, 1e30f, 1e30f}, m2[4] = {1e30f, 1e30f, 1e30f, 1e30f};
    if (SHAPE == 32) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll 1
            for (int t = 0; t < kTiles; ++t) {
                float16v acc0 = {0}, acc1 = {0};
                if (BIAS) {   // accumulator init from 16 per-row biases, four broadcast ds_read_b128 (as scan_kernel does)
                    const float4 *bp = reinterpret_cast<const float4 *>(lds) + ((t * 37 + (lane >> 5) * 4) & 1023);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 c = bp[g];
                        acc0[4 * g] = c.x; acc0[4 * g + 1] = c.y; acc0[4 * g + 2] = c.z; acc0[4 * g + 3] = c.w;
                    }
                    acc1 = acc0;
                }
                const half8 *a = lds + t * 8 * 64 + lane;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const half8 f = a[ks * 64];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, b[ks], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, b[8 + ks], acc1, 0, 0, 0);
                }
                if (EPI) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        quad(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3], t * 4 + g, m1[0], m2[0]);
                        quad(acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3], t * 4 + g, m1[1], m2[1]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r += 4) sum += __builtin_fminf(acc0[r], acc1[r]);
                }
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll 1
            for (int t = 0; t < kTiles; ++t) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {     // two 16-row tiles per 32-row unit
                    float4v acc[4] = {{0}, {0}, {0}, {0}};
                    const half8 *a = lds + (t * 8 + half * 4) * 64 + lane;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {       // 4 k-steps of 32 dims
                        const half8 f = a[ks * 64];
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
                            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f, b[cb * 4 + ks], acc[cb], 0, 0, 0);
                    }
                    if (EPI) {
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
                            quad(acc[cb][0], acc[cb][1], acc[cb][2], acc[cb][3], t * 2 + half, m1[cb], m2[cb]);
                    } else {
                        sum += __builtin_fminf(__builtin_fminf(acc[0][0], acc[1][0]), __builtin_fminf(acc[2][0], acc[3][0]));
                    }
                }
            }
        }
    }
    out[blockIdx.x * 512 + tid] = sum + m1[0] + m1[1] + m1[2] + m1[3] + m2[0] + m2[1] + m2[2] + m2[3];
}

// 32x32x16 with the select interleaved INTO the MFMA stream by sched_group_barrier: per tile, phase A issues the 8
// MFMAs of column block 0 while retiring block 1 of the previous tile (3 VALU per MFMA), phase B issues block 1's MFMAs
// while retiring block 0 of this tile and requesting the next tile's fragments (1 ds_read per MFMA).
__global__ __launch_bounds__(512, 2) void fused_kernel(const half8 *A, const half8 *B, float *out, int iters) {
    __shared__ half8 lds[kTiles * 8 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kTiles * 8 * 64; i += 512) lds[i] = A[i];
    half8 b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = B[(wave * 16 + i) * 64 + lane];
    __syncthreads();
    float m1[2] = {1e30f, 1e30f}, m2[2] = {1e30f, 1e30f};
    half8 fr[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) fr[ks] = lds[ks * 64 + lane];
    float16v acc0 = {0}, acc1 = {0};
    const float16v zero = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int t = 0; t < kTiles; ++t) {
            // ---- phase A: MFMA block 0 (tile t)  ||  select block 1 (tile t-1)
            float16v prev1 = acc1;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[ks], b[ks], ks == 0 ? zero : acc0, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) quad(prev1[4 * g], prev1[4 * g + 1], prev1[4 * g + 2], prev1[4 * g + 3], t * 4 + g, m1[1], m2[1]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase B: MFMA block 1 (tile t)  ||  select block 0 (tile t)  ||  fragments of tile t+1
            const half8 *an = lds + ((t + 1) % kTiles) * 8 * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[ks], b[8 + ks], ks == 0 ? zero : acc1, 0, 0, 0);
                fr[ks] = an[ks * 64];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) quad(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3], t * 4 + g, m1[0], m2[0]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    out[blockIdx.x * 512 + tid] = m1[0] + m1[1] + m2[0] + m2[1] + acc1[0];
}

// 32x32x16, FOUR column blocks per wave (128 queries, B = 128 VGPRs), one wave per SIMD (4 waves per workgroup):
// every A fragment read from LDS feeds 4 MFMAs instead of 2 -- how much of the power budget do the LDS reads take?
template <bool EPI>
__global__ __launch_bounds__(256, 1) void wide_kernel(const half8 *A, const half8 *B, float *out, int iters) {
    __shared__ half8 lds[kTiles * 8 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kTiles * 8 * 64; i += 256) lds[i] = A[i];
    half8 b[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) b[i] = B[((wave * 32 + i) % 128) * 64 + lane];
    __syncthreads();
    float sum = 0.f;
    float m1[4] = {1e30f, 1e30f, 1e30f, 1e30f}, m2[4] = {1e30f, 1e30f, 1e30f, 1e30f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int t = 0; t < kTiles; ++t) {
            float16v acc[4] = {{0}, {0}, {0}, {0}};
            const half8 *a = lds + t * 8 * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const half8 f = a[ks * 64];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, b[cb * 8 + ks], acc[cb], 0, 0, 0);
            }
            if (EPI) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        quad(acc[cb][4 * g], acc[cb][4 * g + 1], acc[cb][4 * g + 2], acc[cb][4 * g + 3], t * 4 + g, m1[cb], m2[cb]);
            } else {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) sum += acc[cb][0];
            }
        }
    }
    out[blockIdx.x * 256 + tid] = sum + m1[0] + m1[1] + m1[2] + m1[3] + m2[0] + m2[1] + m2[2] + m2[3];
}

int main(int argc, char **argv) {
    const int integer_data = argc > 1 ? atoi(argv[1]) : 0;
    const int nblk = 256 * 4, iters = 600;
    std::mt19937 rng(1);
    std::normal_distribution<float> g(0.f, 1.f);
    std::gamma_distribution<float> gm(0.6f, 40.f);
    auto val = [&]() -> _Float16 { return integer_data ? (_Float16)std::min(218.f, std::floor(gm(rng))) : (_Float16)g(rng); };
    std::vector<half8> hA(kTiles * 8 * 64), hB(8 * 16 * 64);
    for (auto &v : hA) for (int j = 0; j < 8; ++j) v[j] = val();
    for (auto &v : hB) for (int j = 0; j < 8; ++j) v[j] = (_Float16)(-2.f) * val();
    half8 *dA, *dB; float *dO;
    CK(hipMalloc(&dA, hA.size() * 16)); CK(hipMalloc(&dB, hB.size() * 16)); CK(hipMalloc(&dO, nblk * 512 * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = (double)nblk * 8 * iters * kTiles * 16 * 32768.0;   // per wave and unit: 16 MFMA 32x32x16
    for (int round = 0; round < 3; ++round)
      for (int epi = 0; epi < 4; ++epi)      // 0 bare, 1 select, 2 select + bias init, 3 select interleaved by sched_group_barrier
        for (int shape : {32, 16, 64}) {          // 64 = 32x32x16 with 4 column blocks per wave, 1 wave per SIMD
            if (epi >= 2 && shape != 32) continue;   // (variants 2, 3: 32x32 only)
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                if (shape == 64 && !epi) wide_kernel<false><<<nblk, 256>>>(dA, dB, dO, iters);
                else if (shape == 64) wide_kernel<true><<<nblk, 256>>>(dA, dB, dO, iters);
                else if (shape == 32 && !epi) loop_kernel<32, false><<<nblk, 512>>>(dA, dB, dO, iters);
                else if (shape == 32 && epi == 1) loop_kernel<32, true><<<nblk, 512>>>(dA, dB, dO, iters);
                else if (shape == 32 && epi == 2) loop_kernel<32, true, true><<<nblk, 512>>>(dA, dB, dO, iters);
                else if (shape == 32) fused_kernel<<<nblk, 512>>>(dA, dB, dO, iters);
                else if (!epi) loop_kernel<16, false><<<nblk, 512>>>(dA, dB, dO, iters);
                else loop_kernel<16, true><<<nblk, 512>>>(dA, dB, dO, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2 && ms < best) best = ms;
            }
            if (round > 0) printf("data=%s select=%d shape=%dx%d ms=%.3f TFLOP/s=%.1f\n", integer_data ? "sift-int" : "gauss", epi, shape, shape, best, flops / best / 1e9);
        }
    return 0;
}
