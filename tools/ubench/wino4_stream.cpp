// Micro-benchmark: MFMA stream of conv_wino4_kernel without any memory traffic - MT x NT sub-tiles x 6
// Winograd components, one component per slot, operands in registers; optional per-k-step VALU transform.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MT, int NT, int XF, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void kstream(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[MT][NT][6];
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int q = 0; q < 6; ++q) acc[i][j][q] = (f32x4){0, 0, 0, 0};
    float uf[NT], v[MT][6], d[MT][6];
    for (int j = 0; j < NT; ++j) uf[j] = lane * 0.002f + j;
    for (int i = 0; i < MT; ++i) for (int q = 0; q < 6; ++q) { d[i][q] = lane * 0.001f + q; v[i][q] = d[i][q]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if (XF) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float d0 = d[i][0], d1 = d[i][1], d2 = d[i][2], d3 = d[i][3], d4 = d[i][4], d5 = d[i][5];
                    const float p = fmaf(-4.0f, d2, d4), q = fmaf(-4.0f, d1, d3);
                    const float s2 = d4 - d2, t2 = d3 - d1;
                    v[i][0] = fmaf(4.0f, d0, fmaf(-5.0f, d2, d4));
                    v[i][1] = p + q;
                    v[i][2] = p - q;
                    v[i][3] = fmaf(2.0f, t2, s2);
                    v[i][4] = fmaf(-2.0f, t2, s2);
                    v[i][5] = fmaf(4.0f, d1, fmaf(-5.0f, d3, d5));
                    for (int q2 = 0; q2 < 6; ++q2) asm volatile("" : "+v"(d[i][q2]));
                }
            }
#pragma unroll
            for (int comp = 0; comp < 6; ++comp) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j][comp] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[j], v[i][comp], acc[i][j][comp], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int q = 0; q < 6; ++q) s += acc[i][j][q][0] + acc[i][j][q][3];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
}

// same MFMAs as kstream<1, NT, 0>, but two components interleaved per slot pair (j0c0, j0c1, j1c0, j1c1, ...)
template <int NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void kstream_il(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[NT][6];
    for (int j = 0; j < NT; ++j) for (int q = 0; q < 6; ++q) acc[j][q] = (f32x4){0, 0, 0, 0};
    float uf[NT], v[6];
    for (int j = 0; j < NT; ++j) uf[j] = lane * 0.002f + j;
    for (int q = 0; q < 6; ++q) v[q] = lane * 0.001f + q;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
#pragma unroll
            for (int cp = 0; cp < 6; cp += 2) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[j][cp] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[j], v[cp], acc[j][cp], 0, 0, 0);
                    acc[j][cp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[j], v[cp + 1], acc[j][cp + 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0;
    for (int j = 0; j < NT; ++j) for (int q = 0; q < 6; ++q) s += acc[j][q][0] + acc[j][q][3];
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

template <int MT, int NT, int XF, int WAVES>
void run(float* out, int iters, const char* what) {
    const double fl = 256.0 * WAVES * iters * 24.0 * MT * NT * 2.0 * 16 * 16 * 4;
    double ms = timeit([&] { hipLaunchKernelGGL((kstream<MT, NT, XF, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, iters); });
    printf("%-44s %.3f ms %.1f TF (%.3f of 157.3)\n", what, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3);
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 1500;
    run<1, 5, 0, 8>(out, iters, "1x5x6 regs-only, 8 waves");
    run<1, 5, 1, 8>(out, iters, "1x5x6 + transform, 8 waves");
    run<2, 2, 0, 8>(out, iters, "2x2x6 regs-only, 8 waves");
    run<2, 2, 1, 8>(out, iters, "2x2x6 + transform, 8 waves");
    run<1, 3, 1, 8>(out, iters, "1x3x6 + transform, 8 waves");
    run<1, 5, 1, 4>(out, iters, "1x5x6 + transform, 4 waves");
    run<2, 2, 1, 4>(out, iters, "2x2x6 + transform, 4 waves");
    run<1, 5, 1, 12>(out, iters, "1x5x6 + transform, 12 waves");
    run<5, 1, 0, 8>(out, iters, "5x1x6 regs-only, 8 waves");
    run<1, 4, 0, 8>(out, iters, "1x4x6 regs-only, 8 waves");
    run<1, 3, 0, 8>(out, iters, "1x3x6 regs-only, 8 waves");
    run<1, 5, 0, 4>(out, iters, "1x5x6 regs-only, 4 waves");
    {
        const double fl = 256.0 * 8 * iters * 24.0 * 5 * 2.0 * 16 * 16 * 4;
        double ms = timeit([&] { hipLaunchKernelGGL((kstream_il<5, 8>), dim3(256), dim3(512), 0, 0, out, iters); });
        printf("%-44s %.3f ms %.1f TF (%.3f of 157.3)\n", "1x5x6 regs-only, components interleaved", ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3);
    }
    return 0;
}
