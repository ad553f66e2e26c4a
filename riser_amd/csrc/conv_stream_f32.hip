// K3s (fp32): ConvNet layers 0 + 1 of the shipped net as ONE streaming kernel, Winograd F(2,3) on the
// f32-input MFMA, no LDS and no barrier.
//
//   layer 0: Conv1d(1 -> C0 <= 20, k=3, 'same', bias) -> ReLU -> MaxPool1d(2,2)
//   layer 1: Conv1d(C0 -> C1 <= 32, k=3, 'same', bias) -> ReLU -> MaxPool1d(2,2)     (riser/nets/cnn.py:52-65)
//
// Layer 1 has K = 20 input channels: one tile of the tiled Winograd kernel (conv_wino.hip) is 80 MFMAs
// per wave against ~400 VALU / LDS / store instructions of staging, transform and epilogue plus a
// workgroup barrier - its matrix pipe is busy 42 % of the time.  Here every WAVE streams 16 pooled output
// rows per step on its own:
//   * the transformed layer-1 weights (A operand: rows = output channels; 4 components x 5 k-steps x NT
//     sub-tiles = 40 registers) and the lane's layer-0 weights stay in registers for the whole kernel;
//   * lane (r, kq) owns pooled row T = 16 u + r and input channels c = kq, 4 + kq, .., 16 + kq: it
//     computes the four layer-1 input rows 2T-1 .. 2T+2 of those channels straight from ten normalised
//     samples x[4T-3 .. 4T+6] (layer 0: the same fmaf chains as conv0_kernel), applies the Winograd input
//     transform, and feeds the MFMAs (B operand: columns = pooled rows) - the layer-0 activations never
//     exist in memory or in LDS;
//   * epilogue as in conv_wino.hip: output transform + bias + ReLU + MaxPool + length mask, one 16-byte
//     buffer store per lane and sub-tile.
// Rows 2T+1, 2T+2 are also computed by the neighbouring lane (as its 2T'-1, 2T'): a 2x redundancy in
// layer-0 arithmetic that is cheaper than any exchange (the VALU work hides under the MFMAs).
// Bit-identical to conv0_kernel + conv_wino_kernel (same fmaf chains, same MFMA accumulation order).
#include "common.hpp"

#include <stdlib.h>

#include <algorithm>
#include <utility>

#ifndef RS_SF32_BLOCKS
#define RS_SF32_BLOCKS 2
#endif
#ifndef RS_SF32_WAVES
#define RS_SF32_WAVES 3
#endif

namespace rs {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 4;
constexpr unsigned kOob = 0x80000000u;
constexpr int KQ = 5;                      // k-steps: 20 input channels / 4

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

struct StreamF32Args {
    const float* xs;          // normalised signals, flat [B * P0], preceded by 16 zero bytes
    const float* w0;          // layer 0: [cp0][4] = (w0, w1, w2, bias)
    const float* w;           // layer 1, Winograd-transformed, packed [n_alloc][nch = 1][4][20]
    const float* bias;        // [n_alloc]
    float* y;                 // [B * P2][cp_out]
    const int32_t* len;
    unsigned xs_bytes, y_bytes;
    int c0;                   // layer-0 channels (<= 20)
    int P2;                   // output rows per read slot (P0 / 4)
    int rows_out;             // B * P2
    int n_reads;
    int cp_out;
    int n_sub;                // ceil(rows_out / 16)
    int sub_per_wave;
};

template <int NT>
__global__ __launch_bounds__(kWaves * 64, RS_SF32_BLOCKS) void conv_stream_f32_kernel(const StreamF32Args a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int gw = blockIdx.x * kWaves + wave;
    const int u0 = gw * a.sub_per_wave;
    const int u1 = min(u0 + a.sub_per_wave, a.n_sub);
    if (u0 >= u1) return;

    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xs - 4), 0, a.xs_bytes + 16u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    const const_len_ptr clen = as_const_len(a.len);

    // ---- resident operands ---------------------------------------------------------------------------
    float uf[4][KQ][NT];                                        // U[comp][n = 16j + r][c = 4 st + kq]
#pragma unroll
    for (int comp = 0; comp < 4; ++comp)
#pragma unroll
        for (int st = 0; st < KQ; ++st)
#pragma unroll
            for (int j = 0; j < NT; ++j) uf[comp][st][j] = a.w[(size_t)(16 * j + r) * 80 + comp * 20 + 4 * st + kq];
    f32x4 w0r[KQ];                                              // layer 0 of this lane's channels 4 st + kq
#pragma unroll
    for (int st = 0; st < KQ; ++st) {
        const int ch = 4 * st + kq;
        w0r[st] = ch < a.c0 ? *reinterpret_cast<const f32x4*>(a.w0 + 4 * ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 bias[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bias[j] = *reinterpret_cast<const f32x4*>(a.bias + 16 * j + 4 * kq);

    // (read, position, lengths) of the sub-tile, carried in scalars and advanced by 16 rows per step
    struct SubInfo {
        int t0, b, len0, len1, lenm;    // position of pooled row 16 u in its read; raw lengths of reads b, b + 1, b - 1
    };
    auto sub_info = [&](int u) {
        SubInfo si;
        const int g0 = 16 * u;
        si.b = g0 / a.P2;
        si.t0 = g0 - si.b * a.P2;
        si.len0 = si.b < a.n_reads ? clen[si.b] : 0;
        si.len1 = si.b + 1 < a.n_reads ? clen[si.b + 1] : 0;
        si.lenm = (si.b >= 1 && si.b - 1 < a.n_reads) ? clen[si.b - 1] : 0;
        return si;
    };
    auto advance = [&](SubInfo& si) {
        si.t0 += 16;
        if (si.t0 >= a.P2) {
            si.t0 -= a.P2;
            ++si.b;
            si.lenm = si.len0;
            si.len0 = si.len1;
            si.len1 = si.b + 1 < a.n_reads ? clen[si.b + 1] : 0;
        }
    };

    // samples x[4T-3 .. 4T+8] of the lane's pooled row T (flat index: P0 = 4 * P2, so sample 4t of read b is
    // element 4 * (b * P2 + t)); the descriptor starts 4 floats early, hence the + 4
    constexpr int D = 2;
    u32x4 pre[D][3];
    auto issue_load = [&](int u, u32x4 (&dst)[3]) {
        const int T = 16 * u + r;
        const unsigned off = T < a.rows_out ? (unsigned)(4 * T - 3 + 4) * 4u : kOob;
#pragma unroll
        for (int k = 0; k < 3; ++k) dst[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off == kOob ? kOob : off + 16u * k, 0, 0);
    };

    auto step = [&](int u, const u32x4 (&raw)[3], const SubInfo& si) {
        if (si.t0 >= (si.len0 >> 2) && si.t0 + 16 <= a.P2) {     // wave-uniform: every output row beyond its read's length
            const unsigned rowoff = (unsigned)(16 * u + r) * (unsigned)(a.cp_out * 4);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = 16 * j + 4 * kq;
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){0u, 0u, 0u, 0u}, rs_y,
                                                       col < a.cp_out ? rowoff + (unsigned)col * 4u : kOob, 0, 0);
            }
            return;
        }
        // ---- validity of the lane's rows (wave-uniform fast path: the whole sub-tile well inside its read) ----
        const int L1a = si.len0 >> 1;                            // valid layer-1 input rows of read b
        const bool fast = si.t0 >= 1 && 2 * si.t0 + 32 < L1a;    // every input row 2T-1 .. 2T+2 of the sub-tile valid, one read
        const int tt = si.t0 + r;
        const bool hi = tt >= a.P2;                              // the lane's pooled row belongs to read b + 1
        const int t = hi ? tt - a.P2 : tt;
        const int L1 = (hi ? si.len1 : si.len0) >> 1;
        // (bit-cast the WHOLE vectors: hipcc compiles __builtin_bit_cast(float, v.y) of a vector element as
        // a load of element 0)
        const f32x4 xa = __builtin_bit_cast(f32x4, raw[0]), xb = __builtin_bit_cast(f32x4, raw[1]),
                    xc = __builtin_bit_cast(f32x4, raw[2]);
        const float xs_[12] = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1], xb[2], xb[3], xc[0], xc[1], xc[2], xc[3]};
        // input row 2t - 1 + k (k = 0..3) of the lane's slot is valid iff it is below the slot's valid rows L1.  Slots are
        // BLOCKS of a read (common.hpp: BlockPlan): row -1 is the last row of the slot before, valid iff that slot is
        // full (the same read continues into this one), and row P1 is row 0 of the slot behind
        const int P1 = 2 * a.P2;
        const int Lprev = (hi ? si.len0 : si.lenm) >> 1, Lnext = hi ? 0 : si.len1 >> 1;
        bool vk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = 2 * t - 1 + k;
            vk[k] = fast || (row < 0 ? Lprev >= P1 : row >= P1 ? row - P1 < Lnext : row < L1);
        }

        f32x4 acc[NT][4];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[j][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        static_for<KQ>([&](auto ST) {
            constexpr int st = decltype(ST)::value;
            // layer 0 for channel 4 st + kq at rows 2T-1 .. 2T+2: row 2T-1+k uses samples xs_[2k .. 2k+3]
            float d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float e = fmaf(w0r[st][2], xs_[2 * k + 2], fmaf(w0r[st][1], xs_[2 * k + 1], fmaf(w0r[st][0], xs_[2 * k], w0r[st][3])));
                const float f = fmaf(w0r[st][2], xs_[2 * k + 3], fmaf(w0r[st][1], xs_[2 * k + 2], fmaf(w0r[st][0], xs_[2 * k + 1], w0r[st][3])));
                d[k] = vk[k] ? fmaxf(fmaxf(e, f), 0.0f) : 0.0f;
            }
            const float v[4] = {d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]};

#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[j][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[c][st][j], v[c], acc[j][c], 0, 0, 0);
        });

        // ---- epilogue ---------------------------------------------------------------------------------------
        const bool valid = t < (L1 >> 1);                        // pooled row < output length of its read
        const unsigned rowoff = (unsigned)(16 * u + r) * (unsigned)(a.cp_out * 4);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float m1 = acc[j][0][q], m2 = acc[j][1][q], m3 = acc[j][2][q], m4 = acc[j][3][q];
                const float y0 = (m1 + m2) + m3;
                const float y1 = (m2 - m3) - m4;
                o[q] = valid ? fmaxf(fmaxf(y0, y1) + bias[j][q], 0.0f) : 0.0f;
            }
            const int col = 16 * j + 4 * kq;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_y,
                                                   col < a.cp_out ? rowoff + (unsigned)col * 4u : kOob, 0, 0);
        }
    };

    issue_load(u0, pre[0]);
    issue_load(u0 + 1, pre[1]);
    SubInfo si = sub_info(u0);
    for (int u = u0; u < u1; u += D) {
        static_for<D>([&](auto K) {
            constexpr int k = decltype(K)::value;
            if (u + k < u1) {
                u32x4 cur[3] = {pre[k][0], pre[k][1], pre[k][2]};
                issue_load(u + k + D, pre[k]);
                step(u + k, cur, si);
                advance(si);
            }
        });
    }
}

using KernelFn = void (*)(const StreamF32Args);

}  // namespace

// layers 0 + 1 qualify when layer 1 is the shipped shape class: 20 padded input channels in one chunk,
// at most 32 outputs, and an output slot of at least 16 rows
bool conv_stream_f32_ok(const ConvLayerDev& L1, int c0, int P_in1) {
    return L1.cp_in == 20 && c0 <= 20 && L1.plan.kc == 20 && L1.plan.nch == 1 && L1.c_out <= 32 && P_in1 >= 64 &&
           !L1.hooks->no_stream_f32;
}

int launch_conv_stream_f32(const ConvLayerDev& L1, const float* d_xs, const float* d_w0, int c0, float* d_y,
                           const int32_t* d_len, int B, int P_in1, int num_cu, hipStream_t st) {
    const int P2 = P_in1 / 2;
    const int64_t rows_out = (int64_t)B * P2;
    const int64_t sb = rows_out * 4 * 4, yb = rows_out * L1.cp_out * 4;
    if (sb >= 0x80000000LL || yb >= 0x80000000LL) {
        set_error("conv_stream_f32: batch too large for the 2 GiB buffer window, split it");
        return RS_ERR_ARG;
    }
    StreamF32Args a;
    a.xs = d_xs;
    a.w0 = d_w0;
    a.w = static_cast<const float*>(L1.d_w);
    a.bias = L1.d_bias;
    a.y = d_y;
    a.len = d_len;
    a.xs_bytes = (unsigned)sb;
    a.y_bytes = (unsigned)yb;
    a.c0 = c0;
    a.P2 = P2;
    a.rows_out = (int)rows_out;
    a.n_reads = B;
    a.cp_out = L1.cp_out;
    a.n_sub = (int)((rows_out + 15) / 16);
    const int waves = num_cu * RS_SF32_WAVES * kWaves;                       // 3 workgroups of 4 waves per CU (153 VGPRs: 3 waves per SIMD)
    // a wave's run: eight sub-blocks or more once every SIMD has a wave; a thin launch (a read or a few) spreads runs of two
    // over more waves (1 / 8 reads: 21 / 22 -> 13 us); RS_SF32_MIN_RUN forces the floor
    const int simds = num_cu * 4;
    const int floor_run = L1.hooks->sf32_min_run > 0 ? L1.hooks->sf32_min_run : std::min(8, std::max(2, (a.n_sub + simds - 1) / simds));
    a.sub_per_wave = std::max(floor_run, (a.n_sub + waves - 1) / waves);
    const int n_waves = (a.n_sub + a.sub_per_wave - 1) / a.sub_per_wave;
    const int grid = (n_waves + kWaves - 1) / kWaves;
    KernelFn fn = round_up(L1.c_out, 16) / 16 <= 1 ? conv_stream_f32_kernel<1> : conv_stream_f32_kernel<2>;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kWaves * 64), 0, st, a);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

}  // namespace rs
