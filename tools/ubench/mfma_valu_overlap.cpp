// Do VALU instructions of one wave execute while ANOTHER wave of the same SIMD runs MFMAs?
// One workgroup of 8 waves per CU = two waves per SIMD (waves w and w + 4 share SIMD w).  Three runs per MFMA kind:
//   M : waves 0-3 loop MFMAs (8 independent accumulators, back to back), waves 4-7 exit at once
//   V : waves 4-7 loop independent v_fma_f32 (8 chains), waves 0-3 exit at once
//   MV: both at the same time
// If the two overlap, t(MV) ~ max(t(M), t(V)); if they share an execution resource, t(MV) ~ t(M) + t(V).
// Kinds: f32-input v_mfma_f32_16x16x4_f32 and v_mfma_f32_16x16x32_f16.
//   hipcc --offload-arch=gfx950 -O3 -o bin/mfma_valu_overlap mfma_valu_overlap.cpp && bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <bool F16, int ROLE>   // ROLE bit 0: MFMA waves run, bit 1: VALU waves run
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    float res = 0.f;
    if (wave < 4) {
        if (!(ROLE & 1)) return;
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, 0.f, 0.f, 0.f};
        const float a = seed + threadIdx.x, b = seed * 0.5f;
        f16x8 ah, bh;
        for (int i = 0; i < 8; ++i) {
            ah[i] = (_Float16)(seed + i);
            bh[i] = (_Float16)(seed - i);
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (F16)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
                else
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            }
        }
        for (int i = 0; i < 8; ++i) res += acc[i][0];
    } else {
        if (!(ROLE & 2)) return;
        float c[8];
        for (int i = 0; i < 8; ++i) c[i] = seed + i;
        const float m = 1.0f + seed * 1e-7f, d = seed * 1e-9f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)          // 64 v_fma_f32 per iteration: 256 issue cycles, as 8 f32 MFMAs
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c[i]) : "v"(m), "v"(d));
        }
        for (int i = 0; i < 8; ++i) res += c[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

// One instruction stream per wave: every MFMA followed by NF independent v_fma_f32 ("fillers"); WAVES = 4 (one wave per
// SIMD) or 8 (two per SIMD, both running the same stream).
template <bool F16, int NF, int WAVES>
__global__ __launch_bounds__(512, 1) void kf(float* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    if (wave >= WAVES) return;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, 0.f, 0.f, 0.f};
    const float a = seed + threadIdx.x, b = seed * 0.5f;
    f16x8 ah, bh;
    for (int i = 0; i < 8; ++i) {
        ah[i] = (_Float16)(seed + i);
        bh[i] = (_Float16)(seed - i);
    }
    float c[8];
    for (int i = 0; i < 8; ++i) c[i] = seed + i;
    const float m = 1.0f + seed * 1e-7f, d = seed * 1e-9f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (F16)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
            else
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NF; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c[(i + f) & 7]) : "v"(m), "v"(d));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float res = 0.f;
    for (int i = 0; i < 8; ++i) res += acc[i][0] + c[i];
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <bool F16, int NF, int WAVES>
static float runf(float* d_out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((kf<F16, NF, WAVES>), dim3(256), dim3(512), 0, 0, d_out, iters, 1.0f);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((kf<F16, NF, WAVES>), dim3(256), dim3(512), 0, 0, d_out, iters, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

template <bool F16, int ROLE>
static float run(float* d_out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<F16, ROLE>), dim3(256), dim3(512), 0, 0, d_out, iters, 1.0f);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<F16, ROLE>), dim3(256), dim3(512), 0, 0, d_out, iters, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    float* d_out;
    hipMalloc(&d_out, 256 * 512 * sizeof(float));
    const int iters = 20000;
    const float m32 = run<false, 1>(d_out, iters), v = run<false, 2>(d_out, iters), mv32 = run<false, 3>(d_out, iters);
    const float m16 = run<true, 1>(d_out, iters), mv16 = run<true, 3>(d_out, iters);
    printf("per iteration of 8 MFMAs / 64 v_fma_f32, one wave of each per SIMD (us per %d iterations):\n", iters);
    printf("  f32 MFMA 16x16x4 : M %.1f  V %.1f  M+V together %.1f  (sum %.1f, max %.1f)\n", m32 * 1e3, v * 1e3, mv32 * 1e3,
           (m32 + v) * 1e3, (m32 > v ? m32 : v) * 1e3);
    printf("  f16 MFMA 16x16x32: M %.1f  V %.1f  M+V together %.1f  (sum %.1f, max %.1f)\n", m16 * 1e3, v * 1e3, mv16 * 1e3,
           (m16 + v) * 1e3, (m16 > v ? m16 : v) * 1e3);
    printf("one stream per wave, every MFMA followed by NF v_fma_f32 (us per %d iterations of 8 MFMAs):\n", iters);
    printf("  f32, 1 wave/SIMD : NF=0 %.1f  NF=2 %.1f  NF=4 %.1f  NF=6 %.1f  NF=8 %.1f\n", runf<false, 0, 4>(d_out, iters) * 1e3,
           runf<false, 2, 4>(d_out, iters) * 1e3, runf<false, 4, 4>(d_out, iters) * 1e3, runf<false, 6, 4>(d_out, iters) * 1e3,
           runf<false, 8, 4>(d_out, iters) * 1e3);
    printf("  f32, 2 waves/SIMD: NF=0 %.1f  NF=2 %.1f  NF=4 %.1f  NF=6 %.1f\n", runf<false, 0, 8>(d_out, iters) * 1e3,
           runf<false, 2, 8>(d_out, iters) * 1e3, runf<false, 4, 8>(d_out, iters) * 1e3, runf<false, 6, 8>(d_out, iters) * 1e3);
    printf("  f16, 1 wave/SIMD : NF=0 %.1f  NF=1 %.1f  NF=2 %.1f  NF=3 %.1f  NF=4 %.1f\n", runf<true, 0, 4>(d_out, iters) * 1e3,
           runf<true, 1, 4>(d_out, iters) * 1e3, runf<true, 2, 4>(d_out, iters) * 1e3, runf<true, 3, 4>(d_out, iters) * 1e3,
           runf<true, 4, 4>(d_out, iters) * 1e3);
    printf("  f16, 2 waves/SIMD: NF=0 %.1f  NF=1 %.1f  NF=2 %.1f  NF=4 %.1f\n", runf<true, 0, 8>(d_out, iters) * 1e3,
           runf<true, 1, 8>(d_out, iters) * 1e3, runf<true, 2, 8>(d_out, iters) * 1e3, runf<true, 4, 8>(d_out, iters) * 1e3);
    return 0;
}
