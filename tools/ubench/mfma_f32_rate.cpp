// Micro-benchmark: sustained rate of the f32-input MFMAs under the operand pattern of
// conv_f32_kernel (LDS fragment reads + MT x NT independent accumulators, 1 or 2 waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE bit0: __syncthreads per iteration; bit1: 7 x 16-byte global loads per thread per iteration
// (prefetch pattern, consumed by LDS writes at the end of the iteration)
template <int MT, int NT, int MODE>
__global__ __launch_bounds__(512) void k16s(float* out, const float4* src, int iters) {
    __shared__ float lds[2 * 15360];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 15360; i += 512) lds[i] = (float)(i & 15) * 0.01f;
    __syncthreads();
    f32x4 acc[MT][NT];
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    float af[MT], bf[NT];
    const float* base = lds + (lane & 15) * 18 + (lane >> 4);
    float4 pre[7];
    const float4* sp = src + (size_t)blockIdx.x * 512 * 7 * 4 + tid;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 2) {
#pragma unroll
            for (int u = 0; u < 7; ++u) pre[u] = sp[(u + 7 * (it & 3)) * 512];
            __builtin_amdgcn_sched_barrier(0);
        }
        const float* b2 = base + (it & 1) * 15360;
#pragma unroll
        for (int st = 0; st < 12; ++st) {
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = b2[i * 16 * 18 + st * 4];
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = b2[4700 + j * 48 * 18 / 4 + st * 4];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (MODE & 2) {
            float* d = lds + ((it & 1) ^ 1) * 15360 + tid * 4;
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                float2* q = reinterpret_cast<float2*>(d + u * 2048 + (u & 1) * 2);
                q[0] = make_float2(pre[u].x, pre[u].y);
                q[1] = make_float2(pre[u].z, pre[u].w);
            }
        }
        if (MODE & 1) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 512 + tid] = s;
}

template <int MT, int NT, bool LDS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k16(float* out, int iters) {
    __shared__ float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += WAVES * 64) lds[i] = (float)(i & 15) * 0.01f;
    __syncthreads();
    f32x4 acc[MT][NT];
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    float af[MT], bf[NT];
    for (int i = 0; i < MT; ++i) af[i] = lane * 0.001f + i;
    for (int j = 0; j < NT; ++j) bf[j] = lane * 0.002f + j;
    const float* base = lds + (lane & 15) * 18 + (lane >> 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 12; ++st) {
            if (LDS) {
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = base[i * 16 * 18 + st * 4];
#pragma unroll
                for (int j = 0; j < NT; ++j) bf[j] = base[2048 + j * 48 * 18 / 4 + st * 4];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * WAVES * 64 + tid] = s;
}

template <int T, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k32(float* out, int iters) {
    f32x16 acc[T];
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < T; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    float a = lane * 0.001f, b = lane * 0.002f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 12; ++st)
#pragma unroll
            for (int i = 0; i < T; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + st, b + i, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < T; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
}

template <class F>
double timeit(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4 * 4);
    float4* src; hipMalloc(&src, (size_t)256 * 512 * 7 * 4 * 16); hipMemset(src, 0, (size_t)256 * 512 * 7 * 4 * 16);
    const int iters = 2000;
    {
        const double fl = 256.0 * 8 * iters * 12 * 24 * 2.0 * 16 * 16 * 4;
        double ms = timeit([&] { hipLaunchKernelGGL((k16s<4, 6, 0>), dim3(256), dim3(512), 0, 0, out, src, iters); });
        printf("staged-variant plain:            %.3f ms %.1f TF\n", ms, fl / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((k16s<4, 6, 1>), dim3(256), dim3(512), 0, 0, out, src, iters); });
        printf("staged-variant + barrier/item:   %.3f ms %.1f TF\n", ms, fl / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((k16s<4, 6, 2>), dim3(256), dim3(512), 0, 0, out, src, iters); });
        printf("staged-variant + loads+ldswrite: %.3f ms %.1f TF\n", ms, fl / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((k16s<4, 6, 3>), dim3(256), dim3(512), 0, 0, out, src, iters); });
        printf("staged-variant + both:           %.3f ms %.1f TF\n", ms, fl / ms / 1e9);
    }
    {
        const double fl = 256.0 * 8 * iters * 12 * 24 * 2.0 * 16 * 16 * 4;
        double ms = timeit([&] { hipLaunchKernelGGL((k16<4, 6, false, 8>), dim3(256), dim3(512), 0, 0, out, iters); });
        printf("16x16x4 regs-only 8 waves 4x6 acc: %.3f ms %.1f TF\n", ms, fl / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((k16<4, 6, true, 8>), dim3(256), dim3(512), 0, 0, out, iters); });
        printf("16x16x4 LDS-frag  8 waves 4x6 acc: %.3f ms %.1f TF\n", ms, fl / ms / 1e9);
    }
    {
        const double fl = 256.0 * 4 * iters * 12 * 24 * 2.0 * 16 * 16 * 4;
        double ms = timeit([&] { hipLaunchKernelGGL((k16<4, 6, false, 4>), dim3(256), dim3(256), 0, 0, out, iters); });
        printf("16x16x4 regs-only 4 waves 4x6 acc: %.3f ms %.1f TF\n", ms, fl / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((k16<4, 6, true, 4>), dim3(256), dim3(256), 0, 0, out, iters); });
        printf("16x16x4 LDS-frag  4 waves 4x6 acc: %.3f ms %.1f TF\n", ms, fl / ms / 1e9);
    }
    {
        const double fl = 256.0 * 8 * iters * 12 * 4 * 2.0 * 16 * 16 * 4;
        double ms = timeit([&] { hipLaunchKernelGGL((k16<2, 2, false, 8>), dim3(256), dim3(512), 0, 0, out, iters); });
        printf("16x16x4 regs-only 8 waves 2x2 acc: %.3f ms %.1f TF\n", ms, fl / ms / 1e9);
    }
    {
        const double fl = 256.0 * 8 * iters * 12 * 6 * 2.0 * 32 * 32 * 2;
        double ms = timeit([&] { hipLaunchKernelGGL((k32<6, 8>), dim3(256), dim3(512), 0, 0, out, iters); });
        printf("32x32x2 regs-only 8 waves 6 acc:   %.3f ms %.1f TF\n", ms, fl / ms / 1e9);
        const double fl4 = 256.0 * 4 * iters * 12 * 6 * 2.0 * 32 * 32 * 2;
        ms = timeit([&] { hipLaunchKernelGGL((k32<6, 4>), dim3(256), dim3(256), 0, 0, out, iters); });
        printf("32x32x2 regs-only 4 waves 6 acc:   %.3f ms %.1f TF\n", ms, fl4 / ms / 1e9);
    }
    return 0;
}
