// Microbenchmark (round 4): the two fp16 MFMA shapes of gfx950 in the flat scan's PRODUCTION structure -- 8 waves per workgroup
// (2 per SIMD), 4-tile stages arriving by LDS-DMA into a double buffer, one barrier per stage, the two halves of the workgroup
// running MFMA / select in opposite order, oct select (v_min3 chains, id in the low mantissa bits, running min / second min).
//   v_mfma_f32_32x32x16_f16 (scan_kernel)   vs   v_mfma_f32_16x16x32_f16 (a 32-row tile = two 16-row blocks, one oct per lane)
// D = 128 (KS = 8 / 4) and D = 64 (KS = 4 / 2); 64 queries per wave.  In-kernel clock as mfma_i8_shapes.hip.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 mfma_f16_staged.hip -o mfma_f16_staged
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int kVec = 8192;      // 128 KiB of A fragments in global memory (L2 resident), streamed stage by stage

__device__ __forceinline__ float oct_min(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, float neg_inf) {
    float t = __builtin_fminf(__builtin_fminf(a0, a1), a2);
    t = __builtin_fminf(__builtin_fminf(t, a3), a4);
    t = __builtin_fminf(__builtin_fminf(t, a5), a6);
    return __builtin_fminf(__builtin_fminf(t, a7), -neg_inf);
}
__device__ __forceinline__ void fold(float o, unsigned mask, unsigned id, float neg_inf, float &m1, float &m2) {
    const float v = __uint_as_float((__float_as_uint(o) & mask) | id);
    m2 = __builtin_amdgcn_fmed3f(m1, m2, v);
    m1 = __builtin_amdgcn_fmed3f(m1, v, neg_inf);
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
__device__ __forceinline__ void dma16(const half8 *g, half8 *l) {
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void *>(reinterpret_cast<uintptr_t>(g)),
                                     reinterpret_cast<__attribute__((address_space(3))) void *>(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(l))), 16, 0, 0);
}

// SHAPE 32: KS 16-dim k-steps, 2 column blocks of 32 queries.  SHAPE 16: KS2 = KS / 2 32-dim k-steps, 4 column blocks of 16.
template <int SHAPE, int KS, bool STAG>
__global__ __launch_bounds__(512, 2) void loop(const half8 *A, const half8 *B, float *out, unsigned long long *clk, int iters, unsigned mask_, float ninf_) {
    constexpr int ST = 4, kTile = KS * 64, kStage = ST * kTile;      // half8 vectors per tile / stage (same bytes for both shapes)
    __shared__ half8 lds[2 * kStage + 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned mask = mask_;
    const float neg_inf = ninf_;
    for (int i = tid; i < 2 * kStage + 256; i += 512) lds[i] = A[i % kVec];
    constexpr int NB = SHAPE == 32 ? 2 * KS : 4 * (KS / 2);
    half8 b[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i] = B[((wave * NB + i) % 128) * 64 + lane];
    __syncthreads();
    constexpr int CBS = SHAPE == 32 ? 2 : 4;
    float m1[CBS], m2[CBS];
#pragma unroll
    for (int c = 0; c < CBS; ++c) m1[c] = m2[c] = 3.0e38f;
    const bool late = STAG && wave >= 4;
    float16v acc32[2];
    float4v acc16[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[c][r] = 3.0e38f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc16[u][c] = float4v{3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};
    auto select = [&](int t) {
        if (SHAPE == 32) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    fold(oct_min(acc32[c][8 * g], acc32[c][8 * g + 1], acc32[c][8 * g + 2], acc32[c][8 * g + 3], acc32[c][8 * g + 4],
                                 acc32[c][8 * g + 5], acc32[c][8 * g + 6], acc32[c][8 * g + 7], neg_inf), mask, (unsigned)((2 * t + g) & 63), neg_inf, m1[c], m2[c]);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                fold(oct_min(acc16[0][c][0], acc16[0][c][1], acc16[0][c][2], acc16[0][c][3], acc16[1][c][0], acc16[1][c][1],
                             acc16[1][c][2], acc16[1][c][3], neg_inf), mask, (unsigned)(t & 63) | ((lane >> 4) & 1) << 4, neg_inf, m1[c], m2[c]);
        }
    };
    const Stamp s0 = stamp_now();
    const int nstage = iters * (kVec / kStage);
    for (int st = 0; st < nstage; ++st) {
        const int buf = st & 1;
        {
            const half8 *src = A + ((st + 1) % (kVec / kStage)) * kStage;
            half8 *dst = lds + (buf ^ 1) * kStage;
#pragma unroll
            for (int i = 0; i < kStage / 64 / 8; ++i) dma16(src + (wave + i * 8) * 64 + lane, dst + (wave + i * 8) * 64);
        }
        const half8 *base = lds + buf * kStage;
#pragma unroll 1
        for (int t = 0; t < ST; ++t) {
            const half8 *a = base + t * kTile + lane;
            half8 f[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) f[ks] = a[ks * 64];
            const float4v *bp = reinterpret_cast<const float4v *>(lds + 2 * kStage);
            float16v cin32;
            float4v c0, c1;
            if (SHAPE == 32) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4v c = bp[(t * 8 + (lane >> 5) * 4 + g) & 255];
                    cin32[4 * g] = c[0]; cin32[4 * g + 1] = c[1]; cin32[4 * g + 2] = c[2]; cin32[4 * g + 3] = c[3];
                }
            } else {
                c0 = bp[(t * 8 + (lane >> 4) * 2) & 255];
                c1 = bp[(t * 8 + (lane >> 4) * 2 + 1) & 255];
            }
            if (late) {
                __builtin_amdgcn_sched_barrier(0);
                select(t);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (SHAPE == 32) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    acc32[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[ks], b[ks], ks == 0 ? cin32 : acc32[0], 0, 0, 0);
                    acc32[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[ks], b[KS + ks], ks == 0 ? cin32 : acc32[1], 0, 0, 0);
                }
            } else {
                constexpr int K2 = KS / 2;        // fragment v = 2 ks2 + rb
#pragma unroll
                for (int ks = 0; ks < K2; ++ks)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc16[rb][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[2 * ks + rb], b[c * K2 + ks], ks == 0 ? (rb ? c1 : c0) : acc16[rb][c], 0, 0, 0);
            }
            if (!late) {
                if (STAG) __builtin_amdgcn_sched_barrier(0);
                select(t);
                if (STAG) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    const Stamp s1 = stamp_now();
    float r = 0.f;
#pragma unroll
    for (int c = 0; c < CBS; ++c) r += m1[c] + m2[c];
    r += acc32[1][5] + acc16[1][2][1];
    out[blockIdx.x * 512 + tid] = r;
    if (lane == 0) {
        clk[((size_t)blockIdx.x * 8 + wave) * 2] = s1.cyc - s0.cyc;
        clk[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = s1.real - s0.real;
    }
}

struct Result { float ms; double cyc, ghz; };
template <class K>
Result run(K kernel, const half8 *dA, const half8 *dB, float *dO, unsigned long long *dC, int nblk, int iters, hipEvent_t e0, hipEvent_t e1) {
    Result best{1e30f, 0, 0};
    const size_t nw = (size_t)nblk * 8;
    std::vector<unsigned long long> h(nw * 2);
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        kernel<<<nblk, 512>>>(dA, dB, dO, dC, iters, 0xFFFFFFC0u, -__builtin_inff());
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
    const int nblk = 256 * 4, iters = 150;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<half8> hA(kVec), hB(128 * 64);
    for (auto &x : hA) for (int j = 0; j < 8; ++j) x[j] = (_Float16)nd(rng);
    for (auto &x : hB) for (int j = 0; j < 8; ++j) x[j] = (_Float16)nd(rng);
    half8 *dA, *dB; float *dO; unsigned long long *dC;
    CK(hipMalloc(&dA, hA.size() * 16)); CK(hipMalloc(&dB, hB.size() * 16)); CK(hipMalloc(&dO, (size_t)nblk * 512 * 4));
    CK(hipMalloc(&dC, (size_t)nblk * 8 * 16));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-10s %-5s %-10s %9s %9s %8s %8s\n", "shape", "D", "structure", "ms", "TFLOP/s", "GHz", "pipe");
    for (int round = 0; round < 2; ++round) {
        // per wave and pass over the 128 KiB: kVec / (KS * 64) tiles x 32 rows x 64 queries x (16 KS) dims x 2 flops;
        // MFMA cycles: 32x32x16 = 32768 flops per 32 cycles (8 passes), 16x16x32 = 16384 per 16 -- the same
        auto report = [&](const char *shape, int KS, const char *label, Result r) {
            if (!round) return;
            const double flops_wave = (double)iters * (kVec / (KS * 64)) * 32.0 * 64.0 * (16.0 * KS) * 2.0;
            const double tf = flops_wave * nblk * 8 / r.ms / 1e9;
            const double mfma_cyc = flops_wave / 32768.0 * 32.0;
            printf("%-10s %-5d %-10s %9.3f %9.1f %8.3f %8.3f\n", shape, 16 * KS, label, r.ms, tf, r.ghz, mfma_cyc * 2 / r.cyc);
        };
#define R(SHAPE, KS, STAG, name, label) report(name, KS, label, run(loop<SHAPE, KS, STAG>, dA, dB, dO, dC, nblk, iters, e0, e1));
        R(32, 8, false, "32x32x16", "staged")    R(16, 8, false, "16x16x32", "staged")
        R(32, 8, true, "32x32x16", "+stagger")   R(16, 8, true, "16x16x32", "+stagger")
        R(32, 4, false, "32x32x16", "staged")    R(16, 4, false, "16x16x32", "staged")
        R(32, 4, true, "32x32x16", "+stagger")   R(16, 4, true, "16x16x32", "+stagger")
        R(32, 8, true, "32x32x16", "+stagger")   R(16, 8, true, "16x16x32", "+stagger")
    }
    return 0;
}
