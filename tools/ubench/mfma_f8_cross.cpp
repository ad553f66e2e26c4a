// The cross terms of the split-precision product on the block-scaled 8 / 6-bit matrix instruction.
//   part 1: operand and scale lane maps of v_mfma_scale_f32_16x16x128_f8f6f4 checked with exact integer data
//           (e4m3 / e5m2: 32 bytes per lane = row l & 15, k = 32 (l >> 4) + j; e2m3: 32 x 6 bits packed in 6 dwords; the E8M0
//           scale byte of lane l applies to its row and its 32-k block);
//   part 2: what a (64-channel panel, tap) step costs per 16 x 16 accumulator tile, fragments re-read from LDS every step,
//           eight waves per workgroup (two per SIMD), random data, one workgroup per CU:
//             P0  6 x v_mfma_f32_16x16x32_f16                         (today: hi*hi, lo*hi, hi*lo for two 32-channel panels)
//             P1  2 x f16 + 1 x scaled e4m3 16x16x128                  (cross terms K-concatenated on the 8-bit form)
//             P2  2 x f16 + 1 x scaled e2m3 16x16x128                  (... on the 6-bit form)
//             P3  4 x f16                                             (the RS_EMU_MFMA_FRAC=3 emulation of round 5: 2/3 of the MFMAs)
//             P4  2 x f16                                             (hi*hi alone)
//   hipcc --offload-arch=gfx950 -O3 -o bin/mfma_f8_cross mfma_f8_cross.cpp && bin/mfma_f8_cross
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e__ = (x);                                                          \
        if (e__ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

// ---- part 1 -------------------------------------------------------------------------------------------------------
template <int FMT>   // 0 e4m3, 1 e5m2, 2 e2m3
__global__ void layout_kernel(const unsigned* a, const unsigned* b, const unsigned* sa, const unsigned* sb, float* c) {
    const int l = threadIdx.x;
    i32x8 av, bv;
    for (int i = 0; i < 8; ++i) {
        av[i] = (int)a[l * 8 + i];
        bv[i] = (int)b[l * 8 + i];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, FMT, FMT, 0, (int)sa[l], 0, (int)sb[l]);
    for (int i = 0; i < 4; ++i) c[(4 * (l >> 4) + i) * 16 + (l & 15)] = acc[i];
}

static unsigned enc_int(int v, int fmt) {   // small non-negative integers, exact in every format used
    // e4m3 (bias 7), e5m2 (bias 15), e2m3 (bias 1)
    if (v == 0) return 0;
    int e = 0;
    while ((v >> (e + 1)) != 0) ++e;
    const int M = fmt == 1 ? 2 : 3, bias = fmt == 0 ? 7 : fmt == 1 ? 15 : 1;
    const int mant = ((v << M) >> e) & ((1 << M) - 1);
    if (((mant | (1 << M)) << e) >> M != v) {
        fprintf(stderr, "enc_int: %d not exact\n", v);
        exit(1);
    }
    return (unsigned)(((e + bias) << M) | mant);
}

// element j of lane (row, g) holds logical k = kmap(cand, g, j); the scale byte of lane (row, g) applies to k block g
static int kmap(int cand, int g, int j) {
    if (cand == 0) return 32 * g + j;
    if (cand == 1) return 16 * g + (j & 15) + 64 * (j >> 4);
    return 8 * g + (j & 7) + 32 * (j >> 3);
}
static int check_layout(int fmt, int cand = 0) {
    // A[m][k], B[k][n] integers; scales 2^(sa[m][kb]) and 2^(sb[n][kb]) per 32-k block
    static int A[16][128], B[128][16], SA[16][4], SB[16][4];
    srand(7 + fmt);
    const int top = fmt == 2 ? 7 : fmt == 1 ? 7 : 15;
    for (int m = 0; m < 16; ++m)
        for (int k = 0; k < 128; ++k) A[m][k] = rand() % (top + 1);
    for (int k = 0; k < 128; ++k)
        for (int n = 0; n < 16; ++n) B[k][n] = rand() % (top + 1);
    for (int m = 0; m < 16; ++m)
        for (int kb = 0; kb < 4; ++kb) {
            SA[m][kb] = rand() % 5 - 2;
            SB[m][kb] = rand() % 5 - 2;
        }
    std::vector<unsigned> ha(64 * 8, 0), hb(64 * 8, 0), hsa(64), hsb(64);
    for (int l = 0; l < 64; ++l) {
        const int row = l & 15, kb = l >> 4;
        unsigned char abytes[32], bbytes[32];
        for (int j = 0; j < 32; ++j) {
            abytes[j] = (unsigned char)enc_int(A[row][kmap(cand, kb, j)], fmt);
            bbytes[j] = (unsigned char)enc_int(B[kmap(cand, kb, j)][row], fmt);
        }
        if (fmt < 2) {
            memcpy(&ha[l * 8], abytes, 32);
            memcpy(&hb[l * 8], bbytes, 32);
        } else {
            for (int j = 0; j < 32; ++j) {
                const int bit = 6 * j;
                for (int q = 0; q < 6; ++q) {
                    if ((abytes[j] >> q) & 1) ha[l * 8 + (bit + q) / 32] |= 1u << ((bit + q) % 32);
                    if ((bbytes[j] >> q) & 1) hb[l * 8 + (bit + q) / 32] |= 1u << ((bit + q) % 32);
                }
            }
        }
        hsa[l] = (unsigned)(127 + SA[row][kb]) | 0xAB00u;      // byte 0 is the one op_sel 0 picks
        hsb[l] = (unsigned)(127 + SB[row][kb]) | 0xCD00u;
    }
    unsigned *da, *db, *dsa, *dsb;
    float* dc;
    CK(hipMalloc(&da, 2048));
    CK(hipMalloc(&db, 2048));
    CK(hipMalloc(&dsa, 256));
    CK(hipMalloc(&dsb, 256));
    CK(hipMalloc(&dc, 1024));
    CK(hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice));
    if (fmt == 0) layout_kernel<0><<<1, 64>>>(da, db, dsa, dsb, dc);
    if (fmt == 1) layout_kernel<1><<<1, 64>>>(da, db, dsa, dsb, dc);
    if (fmt == 2) layout_kernel<2><<<1, 64>>>(da, db, dsa, dsb, dc);
    CK(hipDeviceSynchronize());
    float hc[256];
    CK(hipMemcpy(hc, dc, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            double want = 0;
            for (int k = 0; k < 128; ++k) want += (double)A[m][k] * B[k][n] * ldexp(1.0, SA[m][k / 32] + SB[n][k / 32]);
            if (hc[m * 16 + n] != (float)want) {
                if (bad < 4) fprintf(stderr, "  fmt %d C[%d][%d] = %g, want %g\n", fmt, m, n, hc[m * 16 + n], want);
                ++bad;
            }
        }
    printf("layout check, k map %d, format %s: %s (%d of 256 wrong)\n", cand, fmt == 0 ? "e4m3" : fmt == 1 ? "e5m2" : "e2m3", bad ? "MISMATCH" : "ok", bad);
    return bad;
}

// ---- part 2 -------------------------------------------------------------------------------------------------------
constexpr int MT = 4, NT = 6;
// LDS image: fragments lane-linear, 1 KiB each: A16[MT][4], B16[NT][4], A8[MT][2], B8[NT][2] (P0 uses the 16-bit ones only)
constexpr int kA16 = 0, kB16 = kA16 + MT * 4, kA8 = kB16 + NT * 4, kB8 = kA8 + MT * 2, kFrags = kB8 + NT * 2;

template <int P>
__global__ __launch_bounds__(512, 2) void rate_kernel(const unsigned* src, float* out, int iters, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < kFrags * 256; i += 512) reinterpret_cast<unsigned*>(lds)[i] = src[(i + blockIdx.x * 977) % (kFrags * 256)];
    __syncthreads();
    f32x4 acc[MT][NT];
    for (int i = 0; i < MT; ++i)
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        unsigned opaque = 0;                                   // keeps the fragment reads inside the loop
        asm volatile("" : "+s"(opaque));
        auto frag = [&](int f) { return *reinterpret_cast<const u32x4*>(lds + opaque + f * 1024 + lane * 16); };
        u32x4 a16[MT][4], b16[NT][4];
        constexpr int N16 = P == 0 ? 4 : 2;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int h = 0; h < N16; ++h) a16[i][h] = frag(kA16 + i * 4 + h);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int h = 0; h < N16; ++h) b16[j][h] = frag(kB16 + j * 4 + h);
        auto m16 = [&](int i, int j, int ha, int hb) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a16[i][ha]), __builtin_bit_cast(f16x8, b16[j][hb]),
                                                               acc[i][j], 0, 0, 0);
        };
        if constexpr (P == 0) {
#pragma unroll
            for (int pass = 0; pass < 6; ++pass)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int pn = pass / 3, t = pass % 3;       // panel, term
                        m16(i, j, 2 * pn + (t == 1 ? 1 : 0), 2 * pn + (t == 2 ? 1 : 0));
                    }
        } else {
            i32x8 a8[MT], b8[NT];
            if constexpr (P == 1 || P == 2) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const u32x4 x = frag(kA8 + i * 2), y = frag(kA8 + i * 2 + 1);
                    a8[i] = (i32x8){(int)x[0], (int)x[1], (int)x[2], (int)x[3], (int)y[0], (int)y[1], (int)y[2], (int)y[3]};
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const u32x4 x = frag(kB8 + j * 2), y = frag(kB8 + j * 2 + 1);
                    b8[j] = (i32x8){(int)x[0], (int)x[1], (int)x[2], (int)x[3], (int)y[0], (int)y[1], (int)y[2], (int)y[3]};
                }
            }
#pragma unroll
            for (int pass = 0; pass < 2; ++pass)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) m16(i, j, pass, pass);
            if constexpr (P == 3) {
#pragma unroll
                for (int pass = 0; pass < 2; ++pass)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) m16(i, j, pass, 1 - pass);
            }
            if constexpr (P == 1 || P == 2) {
                const int sc = 127 | (120 << 8);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[i], b8[j], acc[i][j], P == 1 ? 0 : 2, P == 1 ? 0 : 2, 0,
                                                                                     sc, 0, sc);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float res = 0.f;
    for (int i = 0; i < MT; ++i)
        for (int j = 0; j < NT; ++j) res += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 512 + tid] = res;
    if (tid == 0) {
        clk[blockIdx.x * 2] = t1 - t0;
        clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

static unsigned short f2h(float f) {
    _Float16 h = (_Float16)f;
    unsigned short u;
    memcpy(&u, &h, 2);
    return u;
}

template <int P>
static void run_rate(const unsigned* dsrc, float* dout, unsigned long long* dclk, int grid, int iters, const char* name) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(rate_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize, kFrags * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) rate_kernel<P><<<grid, 512, kFrags * 1024>>>(dsrc, dout, iters, dclk);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) rate_kernel<P><<<grid, 512, kFrags * 1024>>>(dsrc, dout, iters, dclk);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hc(grid * 2);
    CK(hipMemcpy(hc.data(), dclk, grid * 16, hipMemcpyDeviceToHost));
    double cyc = 0, ghz = 0;
    for (int b = 0; b < grid; ++b) {
        cyc += (double)hc[2 * b];
        ghz += (double)hc[2 * b] / ((double)hc[2 * b + 1] * 10.0);
    }
    cyc /= grid;
    ghz /= grid;
    const double us = ms * 1000.0 / reps;
    // cycles per step and accumulator tile of ONE wave; two waves share a SIMD, so the pipe sees half of it per wave-step
    printf("%-34s %8.1f us/launch  %7.1f wave-cycles per step and accumulator tile (pipe: %5.1f)  in-kernel clock %.2f GHz\n", name, us,
           cyc / iters / (MT * NT), cyc / iters / (MT * NT) / 2.0, ghz);
}

int main() {
    int bad = check_layout(2);
    for (int cand = 0; cand < 3; ++cand) check_layout(0, cand), check_layout(1, cand);
    // random operands: f16 normal(0, 1) values for the 16-bit fragments, random bytes without the NaN / inf codes for the 8-bit ones
    std::vector<unsigned> h(kFrags * 256);
    srand(12345);
    auto nrm = [&]() {
        double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
        return (float)(sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v));
    };
    for (int i = 0; i < kA8 * 256; ++i) h[i] = (unsigned)f2h(nrm()) | ((unsigned)f2h(nrm()) << 16);
    for (int i = kA8 * 256; i < kFrags * 256; ++i) {
        unsigned w = 0;
        for (int b = 0; b < 4; ++b) w |= (unsigned)((rand() & 0xBF)) << (8 * b);   // exponent's top bit clear: finite in e4m3 and e5m2
        h[i] = w;
    }
    unsigned* dsrc;
    float* dout;
    unsigned long long* dclk;
    const int grid = 256;
    CK(hipMalloc(&dsrc, h.size() * 4));
    CK(hipMalloc(&dout, grid * 512 * 4));
    CK(hipMalloc(&dclk, grid * 16));
    CK(hipMemcpy(dsrc, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int iters = 2000;
    for (int round = 0; round < 2; ++round) {
        run_rate<0>(dsrc, dout, dclk, grid, iters, "P0 6 x f16 (today)");
        run_rate<1>(dsrc, dout, dclk, grid, iters, "P1 2 x f16 + scaled e4m3 x128");
        run_rate<2>(dsrc, dout, dclk, grid, iters, "P2 2 x f16 + scaled e2m3 x128");
        run_rate<3>(dsrc, dout, dclk, grid, iters, "P3 4 x f16 (2/3 emulation)");
        run_rate<4>(dsrc, dout, dclk, grid, iters, "P4 2 x f16");
    }
    return bad ? 1 : 0;
}
