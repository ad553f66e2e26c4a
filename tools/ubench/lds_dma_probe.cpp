// Probe of the LDS-DMA form the 16-bit ring kernel (conv_ring_h16.hip) stages with:
//   buffer_load_dwordx4 <voffset>, <rsrc>, 0 offen lds     (M0 = LDS byte address of lane 0; lane l lands at M0 + 16 l)
// Checks, against a host model: (1) every lane's 16 bytes land at M0 + 16 * lane; (2) a lane whose
// voffset is out of the descriptor's range writes ZEROS to its LDS slot (the kernel relies on it for row -1, rows past
// the end of the activation buffer and channel slots past the row); (3) issue cost of a piece among MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -o bin/lds_dma_probe lds_dma_probe.cpp && bin/lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void dma16(unsigned voff, const __amdgpu_buffer_rsrc_t rsrc, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                 :: "v"(voff), "s"(lds_addr), "s"(rsrc) : "memory");
}

__global__ __launch_bounds__(256) void probe(const unsigned* src, unsigned src_bytes, unsigned* out, const unsigned* voffs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // poison
    for (int i = tid; i < 4096; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0xDEADBEEFu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, src_bytes, 0x00020000);
    // wave w writes piece w (1 KiB) at LDS byte 1024 * w + 4096 * 0
    const unsigned voff = voffs[wave * 64 + lane];
    dma16(voff, rs, (unsigned)(1024 * wave));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) out[i] = reinterpret_cast<unsigned*>(lds)[i];
}

// issue-cost probe: NP pieces interleaved with 48 MFMAs per iteration (8 waves, 2 per SIMD)
template <int NP>
__global__ __launch_bounds__(512, 2) void cost(const unsigned* src, unsigned src_bytes, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, src_bytes, 0x00020000);
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    unsigned voff = (unsigned)(blockIdx.x * 65536 + wave * 8192 + lane * 16);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (g < NP) dma16(voff + (unsigned)(g * 1024 + (it & 3) * 16384), rs, (unsigned)(wave * 8192 + g * 1024 + (it & 1) * 65536));
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + tid] = s + reinterpret_cast<float*>(lds)[tid];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main() {
    const int N = 1 << 22;                                     // dwords
    std::vector<unsigned> h(N);
    for (int i = 0; i < N; ++i) h[i] = 0x10000000u + i;
    unsigned *d_src, *d_out, *d_voff;
    CK(hipMalloc(&d_src, N * 4));
    CK(hipMemcpy(d_src, h.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, 4096 * 4));
    CK(hipMalloc(&d_voff, 256 * 4));
    const unsigned src_bytes = 65536;                          // descriptor covers only the first 64 KiB
    std::vector<unsigned> vo(256);
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l) {
            unsigned v = (unsigned)(w * 4096 + ((l * 37) & 63) * 48);    // scattered, 16-byte aligned
            if (w == 1 && (l % 5) == 0) v = 0x80000000u;                 // far out of range
            if (w == 2 && (l % 7) == 0) v = src_bytes - 8;               // straddles the end: 8 bytes in, 8 out
            if (w == 3 && (l % 3) == 0) v = (unsigned)-16;               // "row -1"
            vo[w * 64 + l] = v;
        }
    CK(hipMemcpy(d_voff, vo.data(), 256 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 16384, 0, d_src, src_bytes, d_out, d_voff);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> o(1024);
    CK(hipMemcpy(o.data(), d_out, 4096, hipMemcpyDeviceToHost));
    int bad = 0, zero_oob = 0, stale_oob = 0, partial = 0;
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l) {
            const unsigned v = vo[w * 64 + l];
            for (int k = 0; k < 4; ++k) {
                const unsigned got = o[(w * 64 + l) * 4 + k];
                const unsigned long long byte = (unsigned long long)v + 4 * k;
                const bool in = byte + 4 <= src_bytes;
                if (in) {
                    if (got != h[byte / 4]) { if (bad < 8) printf("MISMATCH w%d l%d k%d got %08x want %08x\n", w, l, k, got, h[byte / 4]); ++bad; }
                } else {
                    if (got == 0) ++zero_oob; else if (got == 0xDEADBEEFu) ++stale_oob; else { ++partial; if (partial < 8) printf("OOB dword w%d l%d k%d = %08x\n", w, l, k, got); }
                }
            }
        }
    // in-range dwords of a straddling access: did they arrive?
    int straddle_in_ok = 0, straddle_in_zero = 0;
    for (int l = 0; l < 64; l += 7) for (int k = 0; k < 2; ++k) {
        const unsigned got = o[(2 * 64 + l) * 4 + k];
        if (got == h[(src_bytes - 8) / 4 + k]) ++straddle_in_ok; else if (got == 0) ++straddle_in_zero;
    }
    printf("in-range mismatches: %d; out-of-range dwords: %d zero, %d stale (LDS untouched), %d other\n", bad, zero_oob, stale_oob, partial);
    printf("straddling access, in-range half: %d delivered, %d zeroed\n", straddle_in_ok, straddle_in_zero);

    // ---- issue cost ------------------------------------------------------------------------------
    float* d_f;
    CK(hipMalloc(&d_f, 256 * 512 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](auto kern, const char* name) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        const int iters = 2000;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, d_src, (unsigned)(N * 4), d_f, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double mfma = 256.0 * 8 * iters * 48;             // per launch
        const double tf = mfma * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12;
        printf("%s: %.3f ms, %.0f TF, %.1f us per 1000 iterations\n", name, ms, tf, ms * 1e3 / iters * 1000);
    };
    run(cost<0>, "48 MFMA/iter, 0 pieces");
    run(cost<2>, "48 MFMA/iter, 2 pieces");
    run(cost<4>, "48 MFMA/iter, 4 pieces");
    run(cost<6>, "48 MFMA/iter, 6 pieces");
    return bad ? 1 : 0;
}
