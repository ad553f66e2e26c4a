// Sequential conv programs: the ResNet variant of the reference (riser/nets/resnet.py:7-131) - a secondary
// architecture that riser/model.py cannot even load (it hard-imports ConvNet) and for which no config or weights are
// shipped - and ConvNet configurations outside the shipped class (depth > 1, kernels other than 3:
// riser/nets/cnn.py:17,52-65).  BatchNorm (eval mode) is folded into the preceding conv on the host, so the device
// executes a list of
//   CONV    y = [relu]( conv1d(x; k, stride, pad) + bias [+ residual] )
//   MAXPOOL y = MaxPool1d(2, stride 2, padding 0 | 1)          (cnn.py:64 / the ResNet stem, resnet.py:83)
// over position-major activations [B][T][C] (exactly C channels per row) with one uniform length per batch, then
// GAP -> FC -> softmax.
// CONV runs on the f32-input MFMA (seq_conv_mfma_kernel): in the position-major layout the im2col row of output
// position t IS a contiguous run of memory - x[b][t*s - pad .. t*s - pad + k) x [0, c_in) - so the GEMM
// (M = B * T_out, N = c_out, K = k * c_in) needs no gather: lane (row r, k-group kq) of a 16x16x4 MFMA reads
// xflat[row_base(r) + 4 * step + kq], zero outside the batch element's own [0, T_in * c_in).  The scalar kernel the
// first round shipped (one thread per position x 4 channels) is kept behind RS_SEQ_SCALAR=1 for the comparison.
// Generic over k / stride / pad / widths, not tuned per shape: the hot path of this repository is the ConvNet in
// conv_wino*.hip / conv_ring_h16.hip.
#include "common.hpp"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <type_traits>
#include <vector>

namespace rs {
namespace {

// one thread = one (b, t_out) position x 4 output channels; weights packed [k][c_in][c_out4*4]
__global__ __launch_bounds__(256) void seq_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const float* __restrict__ add,
                                                       float* __restrict__ y, int B, int T_in, int T_out, int c_in,
                                                       int c_out, int cq, int k, int stride, int pad, int relu) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)B * T_out * cq;
    if (g >= total) return;
    const int q = (int)(g % cq);
    const int64_t bt = g / cq;
    const int t = (int)(bt % T_out);
    const int b = (int)(bt / T_out);
    float4 acc = *reinterpret_cast<const float4*>(bias + 4 * q);
    for (int kk = 0; kk < k; ++kk) {
        const int ti = t * stride - pad + kk;
        if (ti < 0 || ti >= T_in) continue;
        const float* xr = x + ((int64_t)b * T_in + ti) * c_in;
        const float* wr = w + ((int64_t)kk * c_in) * (cq * 4) + 4 * q;
        for (int ci = 0; ci < c_in; ++ci) {
            const float xv = xr[ci];
            const float4 wv = *reinterpret_cast<const float4*>(wr + (int64_t)ci * (cq * 4));
            acc.x = fmaf(xv, wv.x, acc.x);
            acc.y = fmaf(xv, wv.y, acc.y);
            acc.z = fmaf(xv, wv.z, acc.z);
            acc.w = fmaf(xv, wv.w, acc.w);
        }
    }
    float o[4] = {acc.x, acc.y, acc.z, acc.w};
    const int64_t obase = ((int64_t)b * T_out + t) * c_out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = 4 * q + j;
        if (co < c_out) {
            float v = o[j];
            if (add) v += add[obase + co];
            if (relu) v = fmaxf(v, 0.0f);
            y[obase + co] = v;
        }
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// CONV on the f32-input MFMA: a 256-thread workgroup = 4 waves x (16 output rows x 16 * NT output channels)
template <int NT>
__global__ __launch_bounds__(256) void seq_conv_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ add,
                                                            float* __restrict__ y, int B, int T_in, int T_out, int c_in,
                                                            int c_out, int wpitch, int K, int stride, int pad, int relu) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int64_t rows = (int64_t)B * T_out;
    const int64_t g = (int64_t)blockIdx.x * 64 + wave * 16 + r;          // the output row whose im2col row this lane feeds
    const bool row_ok = g < rows;
    const int b = row_ok ? (int)(g / T_out) : 0;
    const int t = row_ok ? (int)(g - (int64_t)b * T_out) : 0;
    const int off0 = (t * stride - pad) * c_in;                            // first element of the im2col row inside the element
    const int lim = T_in * c_in;
    const float* xb = x + (int64_t)b * lim;
    const int n0 = blockIdx.y * (16 * NT);
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int steps = (K + 3) / 4;
    constexpr int U = 4;                                                   // k-steps in flight
    for (int s0 = 0; s0 < steps; s0 += U) {
        float av[U], bv[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kidx = 4 * (s0 + u) + kq;
            const int o = off0 + kidx;
            av[u] = (row_ok && kidx < K && o >= 0 && o < lim) ? xb[o] : 0.0f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = n0 + 16 * j + r;
                bv[u][j] = (kidx < K && col < wpitch) ? w[(int64_t)kidx * wpitch + col] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u][j], acc[j], 0, 0, 0);
    }
    // accumulator element e of lane (col = lane & 15, row group = lane >> 4) is output row 4 * (lane >> 4) + e
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = n0 + 16 * j + r;
        if (col >= c_out) continue;
        const float bcol = bias[col];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t row = (int64_t)blockIdx.x * 64 + wave * 16 + 4 * kq + e;
            if (row < rows) {
                float v = acc[j][e] + bcol;
                if (add) v += add[row * c_out + col];
                if (relu) v = fmaxf(v, 0.0f);
                y[row * c_out + col] = v;
            }
        }
    }
}

// CONV on the f32-input MFMA, weights resident in LDS: the layers of these nets are narrow (K * N * 4 bytes fits LDS
// many times over), so a persistent workgroup loads the whole packed weight matrix once - [K / 4][Npad][4], so that
// lane (column, k-group) reads the B operands of FOUR k-steps with one ds_read_b128 - and walks 128-row tiles of the
// GEMM: a wave owns 32 output rows x all Npad columns and per 16 K elements issues 2 (16-byte) loads of its im2col rows,
// NT ds_read_b128 and 8 * NT MFMAs.  The k index is permuted (lane kq of step 4u + i holds element 16u + 4kq + i) so
// that a lane's four A values of a 16-element chunk are one contiguous 16-byte load of the position-major input.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void seq_conv_mfma_lds_kernel(const float* __restrict__ x, unsigned x_bytes,
                                                                const float* __restrict__ wq /* [K16/4][16 NT][4] */,
                                                                const float* __restrict__ bias, const float* __restrict__ add,
                                                                float* __restrict__ y, int B, int T_in, int T_out, int c_in,
                                                                int c_out, int K, int stride, int pad, int relu, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    constexpr int NP = 16 * NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int K16 = (K + 15) & ~15;
    for (int i = threadIdx.x; i < K16 / 4 * NP; i += 256)
        reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(wq)[i];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    const int64_t rows = (int64_t)B * T_out;
    const int lim = T_in * c_in;
    float bcol[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bcol[j] = 16 * j + r < c_out ? bias[16 * j + r] : 0.0f;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t row0 = (int64_t)tile * 128 + wave * 32;
        int off0[2];
        int64_t base[2];
        bool ok[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int64_t g = row0 + 16 * m + r;
            ok[m] = g < rows;
            const int b = ok[m] ? (int)(g / T_out) : 0;
            const int t = ok[m] ? (int)(g - (int64_t)b * T_out) : 0;
            off0[m] = (t * stride - pad) * c_in;
            base[m] = (int64_t)b * lim;
        }
        f32x4 acc[2][NT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // this lane's four im2col elements of rows m = 0, 1 for the chunk at k0 (the next chunk is loaded ahead of the
        // current chunk's MFMAs: the loop is otherwise bound by the round trip of these loads)
        auto load_a = [&](int k0, f32x4 (&av)[2]) {
            const int kidx = k0 + 4 * kq;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int o = off0[m] + kidx;
                if (ok[m] && o >= 0 && o + 3 < lim && kidx + 3 < K) {
                    av[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)((base[m] + o) * 4), 0, 0));
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        av[m][i] = (ok[m] && kidx + i < K && o + i >= 0 && o + i < lim) ? x[base[m] + o + i] : 0.0f;
                }
            }
        };
        f32x4 av[2], avn[2];
        load_a(0, av);
        for (int k0 = 0; k0 < K16; k0 += 16) {
            f32x4 bv[NT];
            if (k0 + 16 < K16) load_a(k0 + 16, avn);
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = *reinterpret_cast<const f32x4*>(wl + ((k0 / 4 + kq) * NP + 16 * j + r) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][i], bv[j][i], acc[m][j], 0, 0, 0);
            av[0] = avn[0];
            av[1] = avn[1];
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = 16 * j + r;
                if (col >= c_out) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int64_t row = row0 + 16 * m + 4 * kq + e;
                    if (row < rows) {
                        float v = acc[m][j][e] + bcol[j];
                        if (add) v += add[row * c_out + col];
                        if (relu) v = fmaxf(v, 0.0f);
                        y[row * c_out + col] = v;
                    }
                }
            }
    }
}

// ---- fused residual block and fused stem (riser/nets/resnet.py:39-43,54-57,79-84) ------------------------------------
// The launch-per-conv program above moves every intermediate through HBM: a basic block reads its input three times
// (conv, shortcut, residual), writes and re-reads the intermediate.  The two kernels below are what rs_seqnet_create
// substitutes when it recognises the patterns build_program emits:
//
//   STEM   conv1d(1 -> C; k, stride, pad) + BN + ReLU -> MaxPool1d(2, 2, padding 1): the GEMM rows of a tile start at an
//          odd conv position, so a pooling pair is two accumulator registers of one lane and the un-pooled activations
//          never exist in memory.
//   BLOCK  y = relu( conv3(relu(conv3(x; stride) + b1)) + b2 + shortcut(x) ), shortcut = x or conv1(x; stride) + b:
//          a workgroup owns R - 2 output positions of one read; phase 1 computes the R rows of the intermediate they
//          need (one halo row each side) into LDS, phase 2 runs the second conv with its im2col rows read from that LDS
//          tile (a row of the tile is a run of the next conv's K index, exactly as in global memory) and the 1x1 shortcut
//          conv as extra K chunks of the same GEMM read from x; both weight matrices stay in LDS for the whole launch.
//          x is read once (plus the halo), y written once.
// Same MFMA orientation and K order as seq_conv_mfma_lds_kernel (so the intermediate has the bits the unfused program
// computes); fp32 throughout.

template <int NT>
__global__ __launch_bounds__(256) void seq_stem_pool_kernel(const float* __restrict__ x, unsigned x_bytes,
                                                            const float* __restrict__ wq, const float* __restrict__ bias,
                                                            float* __restrict__ y, int B, int L, int T_conv, int TP,
                                                            int c_out, int K, int stride, int pad, int n_tiles,
        const int32_t* __restrict__ rlen /* ragged batches: samples of read b (null: L) */,
        const int32_t* __restrict__ rtconv /* ... and its conv positions (null: T_conv) */) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    constexpr int NP = 16 * NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int K16 = (K + 15) & ~15;
    for (int i = threadIdx.x; i < K16 / 4 * NP; i += 256)
        reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(wq)[i];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    // GEMM row g = b * 2 TP + j holds conv position j - 1 of read b: rows (2p, 2p + 1) are the window of pooled row p
    const int rpr = 2 * TP;
    const int rows = B * rpr;
    float bcol[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bcol[j] = 16 * j + r < c_out ? bias[16 * j + r] : 0.0f;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row0 = tile * 128 + wave * 32;
        // (read, row in it) of a GEMM row: one division per tile, then a step or two (integer division per lane and row
        // cost more VALU time than the tile's MFMAs)
        const int tb0 = __builtin_amdgcn_readfirstlane((tile * 128) / rpr);
        auto locate = [&](int g, int& b, int& j) {
            b = tb0;
            j = g - tb0 * rpr;
            while (j >= rpr) {
                j -= rpr;
                ++b;
            }
        };
        int off0[2], base[2], Lr[2];
        bool ok[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int g = row0 + 16 * m + r;
            int b, j;
            locate(g, b, j);
            const int tc = j - 1;
            const int bq = min(b, B - 1);
            Lr[m] = rlen ? rlen[bq] : L;
            ok[m] = g < rows && tc >= 0 && tc < (rtconv ? rtconv[bq] : T_conv);
            off0[m] = tc * stride - pad;
            base[m] = b * L;
        }
        // a wave whose 32 rows and all their samples lie inside one read takes the loads without bounds tests
        const int jw = row0 - tb0 * rpr;                        // first row of the wave in read tb0 (or beyond: then not interior)
        const int bw = min(tb0, B - 1);                         // (wave-uniform: scalar loads)
        const int Lw = rlen ? as_const_len(rlen)[bw] : L, Tw = rtconv ? as_const_len(rtconv)[bw] : T_conv;
        const bool interior = row0 + 32 <= rows && jw >= 1 && jw + 32 <= rpr - 2 && jw + 31 <= Tw && (jw - 1) * stride - pad >= 0 &&
                              (jw + 31) * stride - pad + K16 + 3 < Lw;
        f32x4 acc[2][NT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto load_a = [&](int k0, f32x4 (&av)[2]) {
            const int kidx = k0 + 4 * kq;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int o = off0[m] + kidx;
                // (elements at K index >= K meet zero weights: inside the read they need no mask)
                if (interior || (ok[m] && o >= 0 && o + 3 < Lr[m])) {
                    av[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)(base[m] + o) * 4u, 0, 0));
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        av[m][i] = (ok[m] && kidx + i < K && o + i >= 0 && o + i < Lr[m]) ? x[(int64_t)base[m] + o + i] : 0.0f;
                }
            }
        };
        f32x4 av[2], avn[2];
        load_a(0, av);
        for (int k0 = 0; k0 < K16; k0 += 16) {
            f32x4 bv[NT];
            if (k0 + 16 < K16) load_a(k0 + 16, avn);
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = *reinterpret_cast<const f32x4*>(wl + ((k0 / 4 + kq) * NP + 16 * j + r) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][i], bv[j][i], acc[m][j], 0, 0, 0);
            av[0] = avn[0];
            av[1] = avn[1];
        }
        // lane (column r, row group kq) holds GEMM rows 4 kq + e: (e = 0, 1) and (2, 3) are pooling windows
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const int g = row0 + 16 * m + 4 * kq + e;              // even
                if (g >= rows) continue;
                int b, j0;
                locate(g, b, j0);
                const int ta = j0 - 1, tb = j0;                        // the window's conv positions (MaxPool pads with -inf)
                const int tcb = rtconv ? rtconv[min(b, B - 1)] : T_conv;
                const bool va = ta >= 0 && ta < tcb, vb = tb < tcb;
                float* yr = y + ((int64_t)b * TP + (j0 >> 1)) * c_out;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int col = 16 * j + r;
                    if (col >= c_out) continue;
                    float v = -INFINITY;
                    if (va) v = fmaxf(v, acc[m][j][e] + bcol[j]);
                    if (vb) v = fmaxf(v, acc[m][j][e + 1] + bcol[j]);
                    yr[col] = fmaxf(v, 0.0f);                          // relu(max) == max(relu)
                }
            }
    }
}

struct BlockArgs {
    const float* x;           // [B][T_in][c_in]
    unsigned x_bytes;
    float* y;                 // [B][T_out][c_out]
    const float* w1q;         // [K1_16 / 4][NPs][4], K index = tap * c_in + c
    const float* b1;          // [16 NT]
    const float* w2q;         // [K2_16 / 4][NPs][4], K index = tap * Cp + c for the 3x3 part, K2a16 + c for the 1x1 shortcut
    const float* b2;          // [16 NT] (the shortcut conv's bias included)
    int NPs;                  // column pitch of the weight matrices: c_out rounded up to 4 (the columns behind it are zeros
                              // that a lane takes from a register, not from LDS)
    int B, T_in, T_out, c_in, c_out, Cp, stride;
    int K1, K2a, Ksc;         // 3 c_in; 3 Cp; c_in if the shortcut is a conv, else 0 (identity: c_in == c_out, stride 1)
    int tiles_per_read, n_tiles;
    // RAGGED batches (rs_seqnet_forward_ragged): rows of read b valid in x / in y, or null = T_in / T_out for every read.  The
    // buffers keep the uniform row pitches T_in / T_out (those of the longest read); a read's tiles behind its own end are skipped
    const int32_t* tin;
    const int32_t* tout;
};

template <int NT, int MTW, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void seq_basic_block_kernel(const BlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int NP = a.NPs;
    constexpr int R = 16 * MTW * WAVES;            // rows of the intermediate tile (WAVES waves x MTW x 16)
    constexpr int kThr = 64 * WAVES;
    constexpr int TO = R - 2;                      // output positions per tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int K1_16 = (a.K1 + 15) & ~15, K2a16 = (a.K2a + 15) & ~15, Ksc16 = (a.Ksc + 15) & ~15;
    float* wl1 = lds;
    float* wl2 = wl1 + K1_16 * NP;
    float* tl = wl2 + (K2a16 + Ksc16) * NP;        // [(R + 4)][Cp]: row j = intermediate position to0 - 1 + j; 4 zero rows behind
    for (int i = threadIdx.x; i < K1_16 / 4 * NP; i += kThr)
        reinterpret_cast<f32x4*>(wl1)[i] = reinterpret_cast<const f32x4*>(a.w1q)[i];
    for (int i = threadIdx.x; i < (K2a16 + Ksc16) / 4 * NP; i += kThr)
        reinterpret_cast<f32x4*>(wl2)[i] = reinterpret_cast<const f32x4*>(a.w2q)[i];
    for (int i = threadIdx.x; i < (R + 4) * a.Cp; i += kThr) tl[i] = 0.0f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const int lim_max = a.T_in * a.c_in;              // row pitch of a read in x (the longest read's)
    float b1c[NT], b2c[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        b1c[j] = a.b1[16 * j + r];
        b2c[j] = a.b2[16 * j + r];
    }
    // EDGE = false: a tile whose every intermediate row, output and x access lies inside its read (all but the first
    // and the last tiles of a read): no per-row masks, no bounds tests in front of the loads - in fp32 every such VALU
    // instruction is issue time next to the MFMAs, not hidden behind them
    auto do_tile = [&](int tile, auto EDGE_) {
        constexpr bool EDGE = decltype(EDGE_)::value;
        const int b = tile / a.tiles_per_read;
        const int to0 = (tile - b * a.tiles_per_read) * TO;
        const int64_t xbase = (int64_t)b * lim_max;
        const int T_in = a.tin ? as_const_len(a.tin)[b] : a.T_in, T_out = a.tout ? as_const_len(a.tout)[b] : a.T_out;
        const int lim = T_in * a.c_in;                     // elements of read b that hold data
        // ---- phase 1: the intermediate rows j = 0 .. R-1 (positions to0 - 1 + j) = relu(conv3(x; stride) + b1) -> LDS ----
        {
            int off0[MTW];
            bool ok[MTW];
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int p = to0 - 1 + (wave * MTW + m) * 16 + r;
                ok[m] = !EDGE || (p >= 0 && p < T_out);
                off0[m] = (p * a.stride - 1) * a.c_in;
            }
            f32x4 acc[MTW][NT];
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            auto load_a = [&](int k0, f32x4 (&av)[MTW]) {
                const int kidx = k0 + 4 * kq;
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const int o = off0[m] + kidx;
                    // (elements at K index >= K1 meet zero weights: inside the read they need no mask)
                    if (!EDGE || (ok[m] && o >= 0 && o + 3 < lim)) {
                        av[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)((xbase + o) * 4), 0, 0));
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            av[m][i] = (ok[m] && kidx + i < a.K1 && o + i >= 0 && o + i < lim) ? a.x[xbase + o + i] : 0.0f;
                    }
                }
            };
            f32x4 av[MTW], avn[MTW];
            load_a(0, av);
            for (int k0 = 0; k0 < K1_16; k0 += 16) {
                f32x4 bv[NT];
                if (k0 + 16 < K1_16) load_a(k0 + 16, avn);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bv[j] = 16 * j + r < NP ? *reinterpret_cast<const f32x4*>(wl1 + ((k0 / 4 + kq) * NP + 16 * j + r) * 4)
                                            : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][i], bv[j][i], acc[m][j], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < MTW; ++m) av[m] = avn[m];
            }
            // rows outside [0, T_out) are the second conv's zero padding; channels >= c_out of a row stay zero
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int jrow = (wave * MTW + m) * 16 + 4 * kq + e;
                    const int p = to0 - 1 + jrow;
                    const bool okp = !EDGE || (p >= 0 && p < T_out);
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int col = 16 * j + r;
                        if (col < a.c_out) tl[jrow * a.Cp + col] = okp ? fmaxf(acc[m][j][e] + b1c[j], 0.0f) : 0.0f;
                    }
                }
        }
        __syncthreads();
        // ---- phase 2: output rows i = 0 .. TO-1 (positions to0 + i): conv3 over LDS rows i .. i+2 (+ the 1x1 shortcut) ----
        {
            f32x4 acc[MTW][NT];
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* trow[MTW];
#pragma unroll
            for (int m = 0; m < MTW; ++m) trow[m] = tl + ((wave * MTW + m) * 16 + r) * a.Cp + 4 * kq;
            for (int k0 = 0; k0 < K2a16; k0 += 16) {
                f32x4 bv[NT], av[MTW];
#pragma unroll
                for (int m = 0; m < MTW; ++m) av[m] = *reinterpret_cast<const f32x4*>(trow[m] + k0);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bv[j] = 16 * j + r < NP ? *reinterpret_cast<const f32x4*>(wl2 + ((k0 / 4 + kq) * NP + 16 * j + r) * 4)
                                            : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][i], bv[j][i], acc[m][j], 0, 0, 0);
            }
            if (a.Ksc) {                                             // 1x1 shortcut conv: x[(to0 + i) * stride][0 .. c_in)
                int off0[MTW];
                bool ok[MTW];
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const int i = (wave * MTW + m) * 16 + r;
                    ok[m] = i < TO && (!EDGE || to0 + i < T_out);
                    off0[m] = (to0 + i) * a.stride * a.c_in;
                }
                auto load_sc = [&](int k0, f32x4 (&av)[MTW]) {
                    const int kidx = k0 + 4 * kq;
#pragma unroll
                    for (int m = 0; m < MTW; ++m) {
                        const int o = off0[m] + kidx;
                        if (ok[m] && (!EDGE || o + 3 < lim)) {
                            av[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)((xbase + o) * 4), 0, 0));
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) av[m][i] = (ok[m] && kidx + i < a.Ksc && o + i < lim) ? a.x[xbase + o + i] : 0.0f;
                        }
                    }
                };
                f32x4 av[MTW], avn[MTW];
                load_sc(0, av);
                for (int k0 = 0; k0 < Ksc16; k0 += 16) {
                    f32x4 bv[NT];
                    if (k0 + 16 < Ksc16) load_sc(k0 + 16, avn);
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        bv[j] = 16 * j + r < NP ? *reinterpret_cast<const f32x4*>(wl2 + (((K2a16 + k0) / 4 + kq) * NP + 16 * j + r) * 4)
                                                : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int m = 0; m < MTW; ++m)
#pragma unroll
                            for (int j = 0; j < NT; ++j)
                                acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][i], bv[j][i], acc[m][j], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MTW; ++m) av[m] = avn[m];
                }
            }
            // ---- output through an fp32 IMAGE of the tile's outputs in LDS (round 5, as seq_basic_block_x3_kernel does): the
            // tile's outputs are ONE contiguous span of y, a lane holds one channel of four rows - direct stores are 4 bytes wide
            // in 64-byte segments.  The image aliases the intermediate tile (read by nobody behind the barrier; what it leaves in
            // the tile's padding channels and rows is finite fp32 that meets zero weights), offset by `mis` floats so that image
            // and span share their 16-byte phase; an identity shortcut's residual is the same span of x, added in the copy-out.
            __syncthreads();
            const int n_out = min(TO, T_out - to0);
            const int64_t s0 = ((int64_t)b * a.T_out + to0) * a.c_out;
            const int mis = (int)(s0 & 3);
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = (wave * MTW + m) * 16 + 4 * kq + e;
                    if (i >= n_out) continue;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int col = 16 * j + r;
                        if (col < a.c_out) tl[mis + i * a.c_out + col] = acc[m][j][e] + b2c[j];
                    }
                }
            __syncthreads();
            {
                const int n_f = n_out * a.c_out;
                const int n_q = (mis + n_f + 3) >> 2;
                const float* xres = a.x + xbase + (int64_t)to0 * a.c_in - mis;
                float* ydst = a.y + (s0 - mis);
                for (int q = threadIdx.x; q < n_q; q += kThr) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(tl + 4 * q);
                    const int lo = 4 * q - mis;
                    if (lo >= 0 && lo + 3 < n_f) {
                        if (!a.Ksc) v += *reinterpret_cast<const f32x4*>(xres + 4 * q);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
                        *reinterpret_cast<f32x4*>(ydst + 4 * q) = v;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (lo + i >= 0 && lo + i < n_f) {
                                float w = v[i];
                                if (!a.Ksc) w += xres[4 * q + i];
                                ydst[4 * q + i] = fmaxf(w, 0.0f);
                            }
                    }
                }
            }
        }
        __syncthreads();                                             // the next tile's phase 1 overwrites the LDS tile
    };
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int tt = tile % a.tiles_per_read;
        const int to0 = tt * TO;
        const int bb = tile / a.tiles_per_read;
        const int T_in = a.tin ? as_const_len(a.tin)[bb] : a.T_in, T_out = a.tout ? as_const_len(a.tout)[bb] : a.T_out;
        if (to0 >= T_out) continue;                            // ragged batch: this read ended before the tile
        const bool interior = to0 >= 2 && to0 + TO <= T_out && (to0 + R - 2) * a.stride + 6 <= T_in;
        if (interior)
            do_tile(tile, std::false_type{});
        else
            do_tile(tile, std::true_type{});
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same residual basic block in SPLIT PRECISION on the bf16 MFMA (rs_seqnet_set_mode(m, RS_BF16X3); BASELINE.json's
// north_star names "the nets/ 1D-ResNet forward pass ... im2col -> MFMA bf16").  Arithmetic of conv_ring_h16.hip: every
// activation and weight is a pair hi = bf16(v), lo = bf16(v - hi); a product is hi*hi + lo*hi + hi*lo on three
// v_mfma_f32_16x16x32_bf16 with fp32 accumulate (~2^-17 per operand).  What changes against the fp32 kernel above:
//   * a k-step is 32 K elements: lane (row r, k-group kq) supplies K elements 8 kq .. 8 kq + 7 - eight CONSECUTIVE floats of the
//     position's im2col run (two 16-byte loads), split into a hi and a lo fragment in registers (24 VALU per fragment, shared by
//     the NT column tiles);
//   * the weights are split on the host and live in LDS as two planes [k-step][kq][n][8 x bf16] (a 16-lane group reads 256
//     contiguous bytes: conflict-free ds_read_b128), the same 4 bytes per (k, n) as the fp32 matrices;
//   * the intermediate tile is stored ALREADY SPLIT (two bf16 planes [row][Cp]): phase 2 reads its fragments with two
//     ds_read_b128 and converts nothing.  Cp = 8 (mod 16) halfwords keeps those reads 16-byte aligned and the 16 rows of a group
//     on distinct banks (row pitch 12 / 20 / 28 / 36 dwords);
//   * activations between blocks stay fp32 in HBM (x in, y out, as before).
// 3 MFMAs of 16 cycles per 32 K elements and accumulator tile against 8 of 32 cycles: 5.3 x less matrix-pipe time; the
// kernel becomes bound by the split's VALU work and its loads.
struct BlockX3Args {
    const float* x;
    unsigned x_bytes;
    float* y;
    const unsigned short* w1;  // planes [hi | lo], each [S1][4][NPs][8]
    const float* b1;
    const unsigned short* w2;  // planes [hi | lo], each [S2a + Ssc][4][NPs][8]; the shortcut's k-steps behind the 3x3 conv's
    const float* b2;
    int NPs;
    int B, T_in, T_out, c_in, c_out, Cp, stride;
    int K1, Ksc;               // 3 c_in; c_in if the shortcut is a conv, else 0
    int S1, S2a, Ssc;          // k-steps of 32: ceil(3 c_in / 32), ceil(3 Cp / 32), ceil(Ksc / 32)
    int tiles_per_read, n_tiles;
    const int32_t* tin;        // ragged batches: see BlockArgs
    const int32_t* tout;
};

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// eight floats -> their bf16 hi parts and the bf16 roundings of the residuals
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
    hi[0] = pack_bf16x2(a[0], a[1]);
    hi[1] = pack_bf16x2(a[2], a[3]);
    hi[2] = pack_bf16x2(b[0], b[1]);
    hi[3] = pack_bf16x2(b[2], b[3]);
    auto lo_of = [](unsigned h, float e0, float e1) {
        return pack_bf16x2(e0 - __builtin_bit_cast(float, h << 16), e1 - __builtin_bit_cast(float, h & 0xffff0000u));
    };
    lo[0] = lo_of(hi[0], a[0], a[1]);
    lo[1] = lo_of(hi[1], a[2], a[3]);
    lo[2] = lo_of(hi[2], b[0], b[1]);
    lo[3] = lo_of(hi[3], b[2], b[3]);
}
__device__ __forceinline__ f32x4 mfma_x3(const u32x4& ah, const u32x4& al, const u32x4& bh, const u32x4& bl, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bh), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, al), __builtin_bit_cast(bf16x8_t, bh), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bl), c, 0, 0, 0);
}

template <int NT, int MTW, int WAVES>
// (launch bound 512 also for the four-wave form: with 256 the compiler keeps MFMA accumulators in AGPRs and copies them in
// and out - 192 extra instructions around the 72 MFMAs of a <2, 2, 4> tile; worth 1 % here, 17 % in conv_wino4.hip's thin shapes)
__global__ __launch_bounds__(512) void seq_basic_block_x3_kernel(const BlockX3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    const int NP = a.NPs;
    constexpr int R = 16 * MTW * WAVES;
    constexpr int kThr = 64 * WAVES;
    constexpr int TO = R - 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int S2 = a.S2a + a.Ssc;
    const int w1_plane = a.S1 * 4 * NP * 8, w2_plane = S2 * 4 * NP * 8;          // halfwords per plane
    unsigned short* wl1 = reinterpret_cast<unsigned short*>(lds8);                // [hi plane | lo plane]
    unsigned short* wl2 = wl1 + 2 * w1_plane;
    unsigned short* tlh = wl2 + 2 * w2_plane;                                     // [(R + 4)][Cp] hi, then the same lo
    unsigned short* tll = tlh + (R + 4) * a.Cp;
    for (int i = threadIdx.x; i < 2 * w1_plane / 8; i += kThr)
        reinterpret_cast<u32x4*>(wl1)[i] = reinterpret_cast<const u32x4*>(a.w1)[i];
    for (int i = threadIdx.x; i < 2 * w2_plane / 8; i += kThr)
        reinterpret_cast<u32x4*>(wl2)[i] = reinterpret_cast<const u32x4*>(a.w2)[i];
    for (int i = threadIdx.x; i < 2 * (R + 4) * a.Cp / 2; i += kThr) reinterpret_cast<unsigned*>(tlh)[i] = 0u;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const int lim_max = a.T_in * a.c_in;              // row pitch of a read in x (the longest read's)
    float b1c[NT], b2c[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        b1c[j] = a.b1[16 * j + r];
        b2c[j] = a.b2[16 * j + r];
    }
    // weight fragments of k-step s: column 16 j + r, k-group kq; columns behind the compact pitch are zeros from a register
    auto load_b = [&](const unsigned short* w, int plane, int s, u32x4 (&bh)[NT], u32x4 (&bl)[NT]) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = 16 * j + r;
            if (n < NP) {
                const unsigned short* q = w + ((s * 4 + kq) * NP + n) * 8;
                bh[j] = *reinterpret_cast<const u32x4*>(q);
                bl[j] = *reinterpret_cast<const u32x4*>(q + plane);
            } else {
                bh[j] = (u32x4){0u, 0u, 0u, 0u};
                bl[j] = (u32x4){0u, 0u, 0u, 0u};
            }
        }
    };
    auto do_tile = [&](int tile, auto EDGE_) {
        constexpr bool EDGE = decltype(EDGE_)::value;
        const int b = tile / a.tiles_per_read;
        const int to0 = (tile - b * a.tiles_per_read) * TO;
        const int64_t xbase = (int64_t)b * lim_max;
        const int T_in = a.tin ? as_const_len(a.tin)[b] : a.T_in, T_out = a.tout ? as_const_len(a.tout)[b] : a.T_out;
        const int lim = T_in * a.c_in;                     // elements of read b that hold data
        // eight consecutive floats of x from element offset o of read b (zeros outside the read; inside it every element is
        // real data - K indices behind the conv's own meet zero weights)
        auto load8 = [&](bool ok, int o, int klim, int kidx, f32x4& lo4, f32x4& hi4) {
            if (!EDGE || (ok && o >= 0 && o + 7 < lim)) {
                lo4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)((xbase + o) * 4), 0, 0));
                hi4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)((xbase + o) * 4 + 16), 0, 0));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lo4[i] = (ok && kidx + i < klim && o + i >= 0 && o + i < lim) ? a.x[xbase + o + i] : 0.0f;
                    hi4[i] = (ok && kidx + 4 + i < klim && o + 4 + i >= 0 && o + 4 + i < lim) ? a.x[xbase + o + 4 + i] : 0.0f;
                }
            }
        };
        // ---- phase 1: the intermediate rows j = 0 .. R-1 (positions to0 - 1 + j) = relu(conv3(x; stride) + b1) -> LDS, split ----
        {
            int off0[MTW];
            bool ok[MTW];
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int p = to0 - 1 + (wave * MTW + m) * 16 + r;
                ok[m] = !EDGE || (p >= 0 && p < T_out);
                off0[m] = (p * a.stride - 1) * a.c_in;
            }
            f32x4 acc[MTW][NT];
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 xa[MTW], xb[MTW], xan[MTW], xbn[MTW];
#pragma unroll
            for (int m = 0; m < MTW; ++m) load8(ok[m], off0[m] + 8 * kq, a.K1, 8 * kq, xa[m], xb[m]);
            for (int s = 0; s < a.S1; ++s) {
                if (s + 1 < a.S1) {
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
                        load8(ok[m], off0[m] + 32 * (s + 1) + 8 * kq, a.K1, 32 * (s + 1) + 8 * kq, xan[m], xbn[m]);
                }
                u32x4 bh[NT], bl[NT];
                load_b(wl1, w1_plane, s, bh, bl);
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    u32x4 ah, al;
                    split8(xa[m], xb[m], ah, al);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[m][j] = mfma_x3(ah, al, bh[j], bl[j], acc[m][j]);
                }
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    xa[m] = xan[m];
                    xb[m] = xbn[m];
                }
            }
            // rows outside [0, T_out) are the second conv's zero padding; channels >= c_out of a row stay zero.  A lane holds
            // one channel of four rows: neighbouring lanes (channels c, c + 1) exchange two values by DPP so that the even lane
            // owns the channel PAIR of rows 0 and 1 and the odd lane that of rows 2 and 3 - a dword per row and plane instead of
            // two halfwords
            const bool odd = r & 1;
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int jrow0 = (wave * MTW + m) * 16 + 4 * kq;
                bool okp[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) okp[e] = !EDGE || (to0 - 1 + jrow0 + e >= 0 && to0 - 1 + jrow0 + e < T_out);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int col = 16 * j + r;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (okp[e] && col < a.c_out) ? fmaxf(acc[m][j][e] + b1c[j], 0.0f) : 0.0f;
                    const float g02 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, odd ? v[0] : v[2]), 0xB1, 0xF, 0xF, true));
                    const float g13 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, odd ? v[1] : v[3]), 0xB1, 0xF, 0xF, true));
                    const int c2 = 16 * j + (r & ~1);
                    if (c2 < a.c_out) {
                        // even lane: rows 0, 1 = (own, neighbour's); odd lane: rows 2, 3 = (neighbour's, own)
                        const float a0 = odd ? g02 : v[0], b0 = odd ? v[2] : g02;
                        const float a1 = odd ? g13 : v[1], b1 = odd ? v[3] : g13;
                        const int at = (jrow0 + (odd ? 2 : 0)) * a.Cp + c2;
                        const unsigned h0 = pack_bf16x2(a0, b0), h1 = pack_bf16x2(a1, b1);
                        *reinterpret_cast<unsigned*>(tlh + at) = h0;
                        *reinterpret_cast<unsigned*>(tlh + at + a.Cp) = h1;
                        *reinterpret_cast<unsigned*>(tll + at) =
                            pack_bf16x2(a0 - __builtin_bit_cast(float, h0 << 16), b0 - __builtin_bit_cast(float, h0 & 0xffff0000u));
                        *reinterpret_cast<unsigned*>(tll + at + a.Cp) =
                            pack_bf16x2(a1 - __builtin_bit_cast(float, h1 << 16), b1 - __builtin_bit_cast(float, h1 & 0xffff0000u));
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase 2: output rows i = 0 .. TO-1 (positions to0 + i): conv3 over LDS rows i .. i+2 (+ the 1x1 shortcut) ----
        {
            f32x4 acc[MTW][NT];
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            int trow[MTW];                                   // halfword index of the lane's fragment in a tile plane
#pragma unroll
            for (int m = 0; m < MTW; ++m) trow[m] = ((wave * MTW + m) * 16 + r) * a.Cp + 8 * kq;
            for (int s = 0; s < a.S2a; ++s) {
                u32x4 bh[NT], bl[NT];
                load_b(wl2, w2_plane, s, bh, bl);
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const u32x4 ah = *reinterpret_cast<const u32x4*>(tlh + trow[m] + 32 * s);
                    const u32x4 al = *reinterpret_cast<const u32x4*>(tll + trow[m] + 32 * s);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[m][j] = mfma_x3(ah, al, bh[j], bl[j], acc[m][j]);
                }
            }
            if (a.Ksc) {                                             // 1x1 shortcut conv: x[(to0 + i) * stride][0 .. c_in)
                int off0[MTW];
                bool ok[MTW];
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const int i = (wave * MTW + m) * 16 + r;
                    ok[m] = i < TO && (!EDGE || to0 + i < T_out);
                    off0[m] = (to0 + i) * a.stride * a.c_in;
                }
                for (int s = 0; s < a.Ssc; ++s) {
                    u32x4 bh[NT], bl[NT];
                    load_b(wl2, w2_plane, a.S2a + s, bh, bl);
#pragma unroll
                    for (int m = 0; m < MTW; ++m) {
                        f32x4 xa, xb;
                        // a row past the tile's outputs (i >= TO) is never stored: it may read anything finite - keep it zero
                        if (ok[m] || !EDGE) {
                            if (ok[m])
                                load8(true, off0[m] + 32 * s + 8 * kq, a.Ksc, 32 * s + 8 * kq, xa, xb);
                            else
                                xa = xb = (f32x4){0.f, 0.f, 0.f, 0.f};
                        } else {
                            xa = xb = (f32x4){0.f, 0.f, 0.f, 0.f};
                        }
                        u32x4 ah, al;
                        split8(xa, xb, ah, al);
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[m][j] = mfma_x3(ah, al, bh[j], bl[j], acc[m][j]);
                    }
                }
            }
            // ---- output: through an fp32 IMAGE of the tile's outputs in LDS, so that the tile leaves in coalesced 16-byte
            // pieces.  y[b][to0 .. to0 + n_out)[0 .. c_out) is ONE contiguous span of memory (rows hold exactly c_out floats), a
            // lane of the accumulator holds one channel of four rows: direct stores are 4 bytes wide in 64-byte segments.  The
            // image aliases the intermediate tile (every wave has finished reading it behind the barrier); it starts `mis` floats
            // in, so that image float 4 q and global float (s0 - mis) + 4 q are both 16-byte aligned.  An identity shortcut's
            // residual is the same span of x (c_in == c_out, stride 1): it is added in the copy-out, from coalesced loads.
            __syncthreads();
            float* img = reinterpret_cast<float*>(tlh);
            const int n_out = min(TO, T_out - to0);                                  // valid output rows of this tile
            const int64_t s0 = ((int64_t)b * a.T_out + to0) * a.c_out;                 // first float of the span in y
            const int mis = (int)(s0 & 3);
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = (wave * MTW + m) * 16 + 4 * kq + e;
                    if (i >= n_out) continue;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int col = 16 * j + r;
                        if (col < a.c_out) img[mis + i * a.c_out + col] = acc[m][j][e] + b2c[j];
                    }
                }
            __syncthreads();
            {
                const int n_f = n_out * a.c_out;                                       // floats of the span
                const int n_q = (mis + n_f + 3) >> 2;                                  // 16-byte pieces that touch it
                const float* xres = a.x + xbase + (int64_t)to0 * a.c_in - mis;          // identity: same span, same phase
                float* ydst = a.y + (s0 - mis);
                for (int q = threadIdx.x; q < n_q; q += kThr) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(img + 4 * q);
                    const int lo = 4 * q - mis;                                        // span index of the piece's first float
                    if (lo >= 0 && lo + 3 < n_f) {
                        if (!a.Ksc) v += *reinterpret_cast<const f32x4*>(xres + 4 * q);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
                        *reinterpret_cast<f32x4*>(ydst + 4 * q) = v;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (lo + i >= 0 && lo + i < n_f) {
                                float w = v[i];
                                if (!a.Ksc) w += xres[4 * q + i];
                                ydst[4 * q + i] = fmaxf(w, 0.0f);
                            }
                    }
                }
            }
            __syncthreads();
            // the image overwrote zeros the tile must keep: its padding channels [c_out, Cp) and the 4 rows behind it (phase 2
            // multiplies them by zero weights - a float's halfword may be a bf16 NaN).  Phase 1 rewrites every (row < R,
            // channel < c_out) itself.
            {
                const int n_pad = a.Cp - a.c_out;
                for (int row = threadIdx.x; row < R + 4; row += kThr)
                    for (int c = a.c_out; c < a.Cp; ++c) {
                        tlh[row * a.Cp + c] = 0;
                        tll[row * a.Cp + c] = 0;
                    }
                for (int t = threadIdx.x; t < 4 * a.c_out; t += kThr) {
                    tlh[R * a.Cp + (t / a.c_out) * a.Cp + t % a.c_out] = 0;
                    tll[R * a.Cp + (t / a.c_out) * a.Cp + t % a.c_out] = 0;
                }
                (void)n_pad;
            }
        }
        // (no barrier here: the next tile's phase 1 writes rows < R x channels < c_out only, which nobody reads before its
        // own barrier, and the copy-out above ended with one)
    };
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int tt = tile % a.tiles_per_read;
        const int to0 = tt * TO;
        const int bb = tile / a.tiles_per_read;
        const int T_in = a.tin ? as_const_len(a.tin)[bb] : a.T_in, T_out = a.tout ? as_const_len(a.tout)[bb] : a.T_out;
        if (to0 >= T_out) continue;                            // ragged batch: this read ended before the tile
        const int lim = T_in * a.c_in;
        // interior: every intermediate row, output and x access (the k-steps' over-read of up to 31 elements included) lies
        // inside the read
        const bool interior = to0 >= 2 && to0 + TO <= T_out && ((to0 + R - 2) * a.stride + 2) * a.c_in + 32 * (a.S1 + 1) <= lim;
        if (interior)
            do_tile(tile, std::false_type{});
        else
            do_tile(tile, std::true_type{});
    }
}

// The stem (conv(1 -> C; k, stride, pad) + BN + ReLU + MaxPool1d(2, 2, padding 1), riser/nets/resnet.py:79-84) in the same split
// precision: the GEMM of seq_stem_pool_kernel with k-steps of 32 samples (a 19-tap stem is ONE k-step of three bf16 MFMAs
// instead of eight f32-input ones), the lane's eight consecutive samples split in registers.  Output through a WAVE-PRIVATE fp32
// image in LDS: a wave's 32 GEMM rows are 16 pooled rows = 16 c_out consecutive floats of y (pooled rows are contiguous across
// reads: g / 2 = b TP + p), always 64-byte aligned, stored 16 bytes per lane; LDS operations of one wave execute in order, so
// the image needs no barrier.
template <int NT>
__global__ __launch_bounds__(512) void seq_stem_pool_x3_kernel(const float* __restrict__ x, unsigned x_bytes,
                                                               const unsigned short* __restrict__ wq /* planes [hi | lo] [S][4][16 NT][8] */,
                                                               const float* __restrict__ bias, float* __restrict__ y, int B, int L,
                                                               int T_conv, int TP, int c_out, int K, int S, int stride, int pad,
                                                               int n_tiles,
        const int32_t* __restrict__ rlen /* ragged batches: samples of read b (null: L) */,
        const int32_t* __restrict__ rtconv /* ... and its conv positions (null: T_conv) */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    constexpr int NP = 16 * NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int plane = S * 4 * NP * 8;
    unsigned short* wl = reinterpret_cast<unsigned short*>(lds8);
    float* img = reinterpret_cast<float*>(lds8 + (size_t)2 * plane * 2) + wave * 16 * c_out;      // 16 pooled rows x c_out
    for (int i = threadIdx.x; i < 2 * plane / 8; i += 256) reinterpret_cast<u32x4*>(wl)[i] = reinterpret_cast<const u32x4*>(wq)[i];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    const int rpr = 2 * TP;
    const int rows = B * rpr;
    const int64_t y_floats = (int64_t)B * TP * c_out;
    float bcol[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bcol[j] = 16 * j + r < c_out ? bias[16 * j + r] : 0.0f;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row0 = tile * 128 + wave * 32;
        const int tb0 = __builtin_amdgcn_readfirstlane((tile * 128) / rpr);
        auto locate = [&](int g, int& b, int& j) {
            b = tb0;
            j = g - tb0 * rpr;
            while (j >= rpr) {
                j -= rpr;
                ++b;
            }
        };
        int off0[2], base[2], Lr[2];
        bool ok[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int g = row0 + 16 * m + r;
            int b, j;
            locate(g, b, j);
            const int tc = j - 1;
            const int bq = min(b, B - 1);
            Lr[m] = rlen ? rlen[bq] : L;
            ok[m] = g < rows && tc >= 0 && tc < (rtconv ? rtconv[bq] : T_conv);
            off0[m] = tc * stride - pad;
            base[m] = b * L;
        }
        const int jw = row0 - tb0 * rpr;
        const int bw = min(tb0, B - 1);                         // (wave-uniform: scalar loads)
        const int Lw = rlen ? as_const_len(rlen)[bw] : L, Tw = rtconv ? as_const_len(rtconv)[bw] : T_conv;
        const bool interior = row0 + 32 <= rows && jw >= 1 && jw + 32 <= rpr - 2 && jw + 31 <= Tw && (jw - 1) * stride - pad >= 0 &&
                              (jw + 31) * stride - pad + 32 * S + 8 < Lw;
        f32x4 acc[2][NT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < S; ++s) {
            u32x4 bh[NT], bl[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const unsigned short* q = wl + ((s * 4 + kq) * NP + 16 * j + r) * 8;
                bh[j] = *reinterpret_cast<const u32x4*>(q);
                bl[j] = *reinterpret_cast<const u32x4*>(q + plane);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int kidx = 32 * s + 8 * kq;
                const int o = off0[m] + kidx;
                f32x4 xa, xb;
                // (samples at K index >= K meet zero weights: inside the read they need no mask)
                if (interior || (ok[m] && o >= 0 && o + 7 < Lr[m])) {
                    xa = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)(base[m] + o) * 4u, 0, 0));
                    xb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)(base[m] + o) * 4u + 16u, 0, 0));
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        xa[i] = (ok[m] && kidx + i < K && o + i >= 0 && o + i < Lr[m]) ? x[(int64_t)base[m] + o + i] : 0.0f;
                        xb[i] = (ok[m] && kidx + 4 + i < K && o + 4 + i >= 0 && o + 4 + i < Lr[m]) ? x[(int64_t)base[m] + o + 4 + i] : 0.0f;
                    }
                }
                u32x4 ah, al;
                split8(xa, xb, ah, al);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[m][j] = mfma_x3(ah, al, bh[j], bl[j], acc[m][j]);
            }
        }
        // lane (column r, row group kq) holds GEMM rows 4 kq + e: (e = 0, 1) and (2, 3) are pooling windows -> the wave's image
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const int g = row0 + 16 * m + 4 * kq + e;              // even
                if (g >= rows) continue;
                int b, j0;
                locate(g, b, j0);
                const int ta = j0 - 1, tb = j0;                        // the window's conv positions (MaxPool pads with -inf)
                const int tcb = rtconv ? rtconv[min(b, B - 1)] : T_conv;
                const bool va = ta >= 0 && ta < tcb, vb = tb < tcb;
                float* ir = img + (8 * m + 2 * kq + (e >> 1)) * c_out;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int col = 16 * j + r;
                    if (col >= c_out) continue;
                    float v = -INFINITY;
                    if (va) v = fmaxf(v, acc[m][j][e] + bcol[j]);
                    if (vb) v = fmaxf(v, acc[m][j][e + 1] + bcol[j]);
                    ir[col] = fmaxf(v, 0.0f);                          // relu(max) == max(relu)
                }
            }
        {
            const int64_t f0 = (int64_t)(row0 >> 1) * c_out;           // first float of the wave's span in y
            const int n_q = 4 * c_out;                                 // 16 c_out floats in 16-byte pieces
            for (int q = lane; q < n_q; q += 64) {
                if (f0 + 4 * q + 3 < y_floats)
                    *reinterpret_cast<f32x4*>(y + f0 + 4 * q) = *reinterpret_cast<const f32x4*>(img + 4 * q);
                else
                    for (int i = 0; i < 4; ++i)
                        if (f0 + 4 * q + i < y_floats) y[f0 + 4 * q + i] = img[4 * q + i];
            }
        }
    }
}

// BOTTLENECK block (riser/nets/resnet.py:60-70): y = relu( conv1(relu(conv3(relu(conv1(x) + b1); stride) + b2)) + b3 +
// shortcut(x) ) in one launch, three GEMM phases with two LDS tiles between them:
//   A  t1 = relu(conv1x1(x) + b1) for the RA = 128 input positions (to0 * stride - 1 ..) the tile's 3x3 conv reads (zero
//      rows outside the read: the 3x3 conv's padding), A operand = rows of x;
//   B  t2 = relu(conv3(t1; stride) + b2) for the tile's R2 = (RA - 3) / stride + 1 outputs (126 / 63): the im2col row of
//      output i is the run of three t1 rows from row i * stride of the LDS tile;
//   C  y = relu(conv1x1(t2) + b3 + shortcut): rows of the t2 tile, the 1x1 shortcut conv as extra K chunks read from x
//      (or the identity residual).
// All three weight matrices stay in LDS; x is read once (plus one halo row each side), y written once.
struct BneckArgs {
    const float* x;
    unsigned x_bytes;
    float* y;
    const float *w1q, *b1, *w2q, *b2, *w3q, *b3;   // [K16 / 4][NP][4] packings, biases padded to 16 NT
    int NPm, NPo;             // column pitches of the mid / out weight matrices
    int B, T_in, T_out, c_in, c_mid, c_out, Cmp, stride;
    int Ksc;                  // c_in if the shortcut is a conv, else 0
    int R2, tiles_per_read, n_tiles;
    const int32_t* tin;       // ragged batches: rows of read b valid in x / in y (null: T_in / T_out); see BlockArgs
    const int32_t* tout;
};

template <int NTM, int NTO>
__global__ __launch_bounds__(256) void seq_bottleneck_block_kernel(const BneckArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int RA = 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int K1_16 = (a.c_in + 15) & ~15, K2_16 = (3 * a.Cmp + 15) & ~15, K3_16 = (a.Cmp + 15) & ~15, Ksc16 = (a.Ksc + 15) & ~15;
    float* wl1 = lds;
    float* wl2 = wl1 + K1_16 * a.NPm;
    float* wl3 = wl2 + K2_16 * a.NPm;
    float* t1 = wl3 + (K3_16 + Ksc16) * a.NPo;     // [RA + 4][Cmp]
    float* t2 = t1 + (RA + 4) * a.Cmp;             // [RA + 4][Cmp]
    for (int i = threadIdx.x; i < K1_16 / 4 * a.NPm; i += 256) reinterpret_cast<f32x4*>(wl1)[i] = reinterpret_cast<const f32x4*>(a.w1q)[i];
    for (int i = threadIdx.x; i < K2_16 / 4 * a.NPm; i += 256) reinterpret_cast<f32x4*>(wl2)[i] = reinterpret_cast<const f32x4*>(a.w2q)[i];
    for (int i = threadIdx.x; i < (K3_16 + Ksc16) / 4 * a.NPo; i += 256)
        reinterpret_cast<f32x4*>(wl3)[i] = reinterpret_cast<const f32x4*>(a.w3q)[i];
    for (int i = threadIdx.x; i < 2 * (RA + 4) * a.Cmp; i += 256) t1[i] = 0.0f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const int lim_max = a.T_in * a.c_in;              // row pitch of a read in x (the longest read's)
    int lim = lim_max, T_in = a.T_in, T_out = a.T_out;   // of the read a tile belongs to (set per tile)
    float b1c[NTM], b2c[NTM], b3c[NTO];
#pragma unroll
    for (int j = 0; j < NTM; ++j) {
        b1c[j] = a.b1[16 * j + r];
        b2c[j] = a.b2[16 * j + r];
    }
#pragma unroll
    for (int j = 0; j < NTO; ++j) b3c[j] = a.b3[16 * j + r];
    const int n_mt = (a.R2 + 15) / 16;             // 16-row tiles of phases B and C (8 or 4)
    auto wfrag = [&](const float* wl, int NP, int k0, int j) -> f32x4 {
        return 16 * j + r < NP ? *reinterpret_cast<const f32x4*>(wl + ((k0 / 4 + kq) * NP + 16 * j + r) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    // a lane's four consecutive x values at element o of the read (zero outside it), K index kidx .. kidx + 3 of Klim
    auto xload = [&](int64_t xbase, bool ok, int o, int kidx, int Klim) -> f32x4 {
        if (ok && o >= 0 && o + 3 < lim)
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)((xbase + o) * 4), 0, 0));
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (ok && kidx + i < Klim && o + i >= 0 && o + i < lim) ? a.x[xbase + o + i] : 0.0f;
        return v;
    };
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int b = tile / a.tiles_per_read;
        const int to0 = (tile - b * a.tiles_per_read) * a.R2;
        const int64_t xbase = (int64_t)b * lim_max;
        if (a.tin) {
            T_in = as_const_len(a.tin)[b];
            T_out = as_const_len(a.tout)[b];
            lim = T_in * a.c_in;
        }
        if (to0 >= T_out) continue;                            // ragged batch: this read ended before the tile
        // ---- phase A: t1 rows j = 0 .. RA-1 <-> input positions q0 + j ------------------------------------------------
        {
            const int q0 = to0 * a.stride - 1;
            f32x4 acc[2][NTM];
            int off0[2];
            bool ok[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int q = q0 + (wave * 2 + m) * 16 + r;
                ok[m] = q >= 0 && q < T_in;
                off0[m] = q * a.c_in;
#pragma unroll
                for (int j = 0; j < NTM; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            for (int k0 = 0; k0 < K1_16; k0 += 16) {
                f32x4 av[2], bv[NTM];
#pragma unroll
                for (int m = 0; m < 2; ++m) av[m] = xload(xbase, ok[m], off0[m] + k0 + 4 * kq, k0 + 4 * kq, a.c_in);
#pragma unroll
                for (int j = 0; j < NTM; ++j) bv[j] = wfrag(wl1, a.NPm, k0, j);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int j = 0; j < NTM; ++j)
                            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][i], bv[j][i], acc[m][j], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int jrow = (wave * 2 + m) * 16 + 4 * kq + e;
                    const int q = q0 + jrow;
                    const bool okq = q >= 0 && q < T_in;
#pragma unroll
                    for (int j = 0; j < NTM; ++j) {
                        const int col = 16 * j + r;
                        if (col < a.c_mid) t1[jrow * a.Cmp + col] = okq ? fmaxf(acc[m][j][e] + b1c[j], 0.0f) : 0.0f;
                    }
                }
        }
        __syncthreads();
        // ---- phase B: t2 row i <-> output position to0 + i: conv3 over t1 rows i * stride .. + 2 -----------------------
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int mt = wave + 4 * m;
            if (mt >= n_mt) break;
            f32x4 acc[NTM];
#pragma unroll
            for (int j = 0; j < NTM; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* trow = t1 + ((mt * 16 + r) * a.stride) * a.Cmp + 4 * kq;
            for (int k0 = 0; k0 < K2_16; k0 += 16) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(trow + k0);
#pragma unroll
                for (int j = 0; j < NTM; ++j) {
                    const f32x4 bv = wfrag(wl2, a.NPm, k0, j);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc[j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = mt * 16 + 4 * kq + e;
#pragma unroll
                for (int j = 0; j < NTM; ++j) {
                    const int col = 16 * j + r;
                    if (col < a.c_mid) t2[i * a.Cmp + col] = fmaxf(acc[j][e] + b2c[j], 0.0f);
                }
            }
        }
        __syncthreads();
        // ---- phase C: y row i = conv1x1(t2 row i) + b3 + shortcut -> ReLU ------------------------------------------------
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int mt = wave + 4 * m;
            if (mt >= n_mt) break;
            f32x4 acc[NTO];
#pragma unroll
            for (int j = 0; j < NTO; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* trow = t2 + (mt * 16 + r) * a.Cmp + 4 * kq;
            for (int k0 = 0; k0 < K3_16; k0 += 16) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(trow + k0);
#pragma unroll
                for (int j = 0; j < NTO; ++j) {
                    const f32x4 bv = wfrag(wl3, a.NPo, k0, j);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc[j], 0, 0, 0);
                }
            }
            if (a.Ksc) {
                const int i_r = mt * 16 + r;
                const bool okr = i_r < a.R2 && to0 + i_r < T_out;
                const int off0 = (to0 + i_r) * a.stride * a.c_in;
                for (int k0 = 0; k0 < Ksc16; k0 += 16) {
                    const f32x4 av = xload(xbase, okr, off0 + k0 + 4 * kq, k0 + 4 * kq, a.Ksc);
#pragma unroll
                    for (int j = 0; j < NTO; ++j) {
                        const f32x4 bv = wfrag(wl3, a.NPo, K3_16 + k0, j);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc[j], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = mt * 16 + 4 * kq + e;
                const int pos = to0 + i;
                if (i >= a.R2 || pos >= T_out) continue;
                const int64_t orow = ((int64_t)b * a.T_out + pos) * a.c_out;
#pragma unroll
                for (int j = 0; j < NTO; ++j) {
                    const int col = 16 * j + r;
                    if (col >= a.c_out) continue;
                    float v = acc[j][e] + b3c[j];
                    if (!a.Ksc) v += a.x[xbase + (int64_t)pos * a.c_in + col];    // identity shortcut
                    a.y[orow + col] = fmaxf(v, 0.0f);
                }
            }
        }
        __syncthreads();
    }
}

// The bottleneck block in split precision on the bf16 MFMA (the arithmetic and operand layouts of seq_basic_block_x3_kernel:
// k-steps of 32, weights as two bf16 planes [k-step][kq][n][8], both LDS tiles stored already split with a row pitch of
// 8 (mod 16) halfwords).  Three GEMM phases as above; the output keeps the fp32 kernel's per-lane stores (c_out = 4 c_mid: an
// image of the tile's outputs does not fit next to the weights).
struct BneckX3Args {
    const float* x;
    unsigned x_bytes;
    float* y;
    const unsigned short *w1, *w2, *w3;            // planes [hi | lo]: [S1][4][NPm][8], [S2][4][NPm][8], [S3 + Ssc][4][NPo][8]
    const float *b1, *b2, *b3;
    int NPm, NPo;
    int B, T_in, T_out, c_in, c_mid, c_out, Cmp, stride;
    int Ksc, S1, S2, S3, Ssc;
    int R2, tiles_per_read, n_tiles;
    const int32_t* tin;       // ragged batches: see BlockArgs
    const int32_t* tout;
};

template <int NTM, int NTO>
__global__ __launch_bounds__(256) void seq_bottleneck_block_x3_kernel(const BneckX3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    constexpr int RA = 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int p1 = a.S1 * 4 * a.NPm * 8, p2 = a.S2 * 4 * a.NPm * 8, p3 = (a.S3 + a.Ssc) * 4 * a.NPo * 8;   // halfwords per plane
    unsigned short* wl1 = reinterpret_cast<unsigned short*>(lds8);
    unsigned short* wl2 = wl1 + 2 * p1;
    unsigned short* wl3 = wl2 + 2 * p2;
    const int tplane = (RA + 4) * a.Cmp;
    unsigned short* t1h = wl3 + 2 * p3;            // t1: [hi plane | lo plane], then t2 the same
    unsigned short* t1l = t1h + tplane;
    unsigned short* t2h = t1l + tplane;
    unsigned short* t2l = t2h + tplane;
    for (int i = threadIdx.x; i < 2 * p1 / 8; i += 256) reinterpret_cast<u32x4*>(wl1)[i] = reinterpret_cast<const u32x4*>(a.w1)[i];
    for (int i = threadIdx.x; i < 2 * p2 / 8; i += 256) reinterpret_cast<u32x4*>(wl2)[i] = reinterpret_cast<const u32x4*>(a.w2)[i];
    for (int i = threadIdx.x; i < 2 * p3 / 8; i += 256) reinterpret_cast<u32x4*>(wl3)[i] = reinterpret_cast<const u32x4*>(a.w3)[i];
    for (int i = threadIdx.x; i < 4 * tplane / 2; i += 256) reinterpret_cast<unsigned*>(t1h)[i] = 0u;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const int lim_max = a.T_in * a.c_in;              // row pitch of a read in x (the longest read's)
    int lim = lim_max, T_in = a.T_in, T_out = a.T_out;   // of the read a tile belongs to (set per tile)
    float b1c[NTM], b2c[NTM], b3c[NTO];
#pragma unroll
    for (int j = 0; j < NTM; ++j) {
        b1c[j] = a.b1[16 * j + r];
        b2c[j] = a.b2[16 * j + r];
    }
#pragma unroll
    for (int j = 0; j < NTO; ++j) b3c[j] = a.b3[16 * j + r];
    const int n_mt = (a.R2 + 15) / 16;
    auto wfrag = [&](const unsigned short* w, int plane, int NP, int s, int j, u32x4& bh, u32x4& bl) {
        const int n = 16 * j + r;
        if (n < NP) {
            const unsigned short* q = w + ((s * 4 + kq) * NP + n) * 8;
            bh = *reinterpret_cast<const u32x4*>(q);
            bl = *reinterpret_cast<const u32x4*>(q + plane);
        } else {
            bh = bl = (u32x4){0u, 0u, 0u, 0u};
        }
    };
    // a lane's eight consecutive x values at element o of the read (zero outside it), K index kidx .. kidx + 7 of Klim
    auto xload8 = [&](int64_t xbase, bool ok, int o, int kidx, int Klim, f32x4& lo4, f32x4& hi4) {
        if (ok && o >= 0 && o + 7 < lim) {
            lo4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)((xbase + o) * 4), 0, 0));
            hi4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)((xbase + o) * 4 + 16), 0, 0));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lo4[i] = (ok && kidx + i < Klim && o + i >= 0 && o + i < lim) ? a.x[xbase + o + i] : 0.0f;
                hi4[i] = (ok && kidx + 4 + i < Klim && o + 4 + i >= 0 && o + 4 + i < lim) ? a.x[xbase + o + 4 + i] : 0.0f;
            }
        }
    };
    // relu(acc + bias) of one accumulator register -> the split tile (one channel of one row)
    auto put_split = [&](unsigned short* th, unsigned short* tl, int at, float v) {
        const __bf16 h = (__bf16)v;
        th[at] = __builtin_bit_cast(unsigned short, h);
        tl[at] = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)h));
    };
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int b = tile / a.tiles_per_read;
        const int to0 = (tile - b * a.tiles_per_read) * a.R2;
        const int64_t xbase = (int64_t)b * lim_max;
        if (a.tin) {
            T_in = as_const_len(a.tin)[b];
            T_out = as_const_len(a.tout)[b];
            lim = T_in * a.c_in;
        }
        if (to0 >= T_out) continue;                            // ragged batch: this read ended before the tile
        // ---- phase A: t1 rows j = 0 .. RA-1 <-> input positions q0 + j ------------------------------------------------
        {
            const int q0 = to0 * a.stride - 1;
            f32x4 acc[2][NTM];
            int off0[2];
            bool ok[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int q = q0 + (wave * 2 + m) * 16 + r;
                ok[m] = q >= 0 && q < T_in;
                off0[m] = q * a.c_in;
#pragma unroll
                for (int j = 0; j < NTM; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            for (int s = 0; s < a.S1; ++s) {
                u32x4 bh[NTM], bl[NTM];
#pragma unroll
                for (int j = 0; j < NTM; ++j) wfrag(wl1, p1, a.NPm, s, j, bh[j], bl[j]);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f32x4 xa, xb;
                    xload8(xbase, ok[m], off0[m] + 32 * s + 8 * kq, 32 * s + 8 * kq, a.c_in, xa, xb);
                    u32x4 ah, al;
                    split8(xa, xb, ah, al);
#pragma unroll
                    for (int j = 0; j < NTM; ++j) acc[m][j] = mfma_x3(ah, al, bh[j], bl[j], acc[m][j]);
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int jrow = (wave * 2 + m) * 16 + 4 * kq + e;
                    const int q = q0 + jrow;
                    const bool okq = q >= 0 && q < T_in;
#pragma unroll
                    for (int j = 0; j < NTM; ++j) {
                        const int col = 16 * j + r;
                        if (col < a.c_mid) put_split(t1h, t1l, jrow * a.Cmp + col, okq ? fmaxf(acc[m][j][e] + b1c[j], 0.0f) : 0.0f);
                    }
                }
        }
        __syncthreads();
        // ---- phase B: t2 row i <-> output position to0 + i: conv3 over t1 rows i * stride .. + 2 -----------------------
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int mt = wave + 4 * m;
            if (mt >= n_mt) break;
            f32x4 acc[NTM];
#pragma unroll
            for (int j = 0; j < NTM; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int trow = ((mt * 16 + r) * a.stride) * a.Cmp + 8 * kq;
            for (int s = 0; s < a.S2; ++s) {
                const u32x4 ah = *reinterpret_cast<const u32x4*>(t1h + trow + 32 * s);
                const u32x4 al = *reinterpret_cast<const u32x4*>(t1l + trow + 32 * s);
#pragma unroll
                for (int j = 0; j < NTM; ++j) {
                    u32x4 bh, bl;
                    wfrag(wl2, p2, a.NPm, s, j, bh, bl);
                    acc[j] = mfma_x3(ah, al, bh, bl, acc[j]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = mt * 16 + 4 * kq + e;
#pragma unroll
                for (int j = 0; j < NTM; ++j) {
                    const int col = 16 * j + r;
                    if (col < a.c_mid) put_split(t2h, t2l, i * a.Cmp + col, fmaxf(acc[j][e] + b2c[j], 0.0f));
                }
            }
        }
        __syncthreads();
        // ---- phase C: y row i = conv1x1(t2 row i) + b3 + shortcut -> ReLU ------------------------------------------------
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int mt = wave + 4 * m;
            if (mt >= n_mt) break;
            f32x4 acc[NTO];
#pragma unroll
            for (int j = 0; j < NTO; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int trow = (mt * 16 + r) * a.Cmp + 8 * kq;
            for (int s = 0; s < a.S3; ++s) {
                const u32x4 ah = *reinterpret_cast<const u32x4*>(t2h + trow + 32 * s);
                const u32x4 al = *reinterpret_cast<const u32x4*>(t2l + trow + 32 * s);
#pragma unroll
                for (int j = 0; j < NTO; ++j) {
                    u32x4 bh, bl;
                    wfrag(wl3, p3, a.NPo, s, j, bh, bl);
                    acc[j] = mfma_x3(ah, al, bh, bl, acc[j]);
                }
            }
            if (a.Ksc) {
                const int i_r = mt * 16 + r;
                const bool okr = i_r < a.R2 && to0 + i_r < T_out;
                const int off0 = (to0 + i_r) * a.stride * a.c_in;
                for (int s = 0; s < a.Ssc; ++s) {
                    f32x4 xa, xb;
                    xload8(xbase, okr, off0 + 32 * s + 8 * kq, 32 * s + 8 * kq, a.Ksc, xa, xb);
                    u32x4 ah, al;
                    split8(xa, xb, ah, al);
#pragma unroll
                    for (int j = 0; j < NTO; ++j) {
                        u32x4 bh, bl;
                        wfrag(wl3, p3, a.NPo, a.S3 + s, j, bh, bl);
                        acc[j] = mfma_x3(ah, al, bh, bl, acc[j]);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = mt * 16 + 4 * kq + e;
                const int pos = to0 + i;
                if (i >= a.R2 || pos >= T_out) continue;
                const int64_t orow = ((int64_t)b * a.T_out + pos) * a.c_out;
#pragma unroll
                for (int j = 0; j < NTO; ++j) {
                    const int col = 16 * j + r;
                    if (col >= a.c_out) continue;
                    float v = acc[j][e] + b3c[j];
                    if (!a.Ksc) v += a.x[xbase + (int64_t)pos * a.c_in + col];    // identity shortcut
                    a.y[orow + col] = fmaxf(v, 0.0f);
                }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void seq_maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                          int T_in, int T_out, int c, int pad) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (int64_t)B * T_out * c) return;
    const int ch = (int)(g % c);
    const int64_t bt = g / c;
    const int t = (int)(bt % T_out);
    const int b = (int)(bt / T_out);
    const int t0 = 2 * t - pad, t1 = 2 * t + 1 - pad;             // window of MaxPool1d(2, 2, padding pad)
    float v = -INFINITY;
    if (t0 >= 0 && t0 < T_in) v = fmaxf(v, x[((int64_t)b * T_in + t0) * c + ch]);
    if (t1 >= 0 && t1 < T_in) v = fmaxf(v, x[((int64_t)b * T_in + t1) * c + ch]);
    y[g] = v;
}

// GAP over T rows -> FC(c, 2) -> softmax; one 256-thread workgroup per read: wave w sums the rows t = w (mod 4) of
// each channel (coalesced 256-byte row segments, four rows in flight per channel group), LDS combines the four partial
// sums in a fixed order, wave 0 finishes
__global__ __launch_bounds__(256) void seq_head_kernel(const float* __restrict__ x, int T_pitch, int c,
                                                       const float* __restrict__ fcw, const float* __restrict__ fcb,
                                                       float* __restrict__ probs, float* __restrict__ logits,
                                                       const int32_t* __restrict__ rt /* ragged batches: rows of read b (null: T_pitch) */) {
    __shared__ float part[4][64];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int T = rt ? as_const_len(rt)[b] : T_pitch;
    float a0 = 0.f, a1 = 0.f;
    for (int c0 = 0; c0 < c; c0 += 64) {
        const int ch = c0 + lane;
        float s = 0.f;
        if (ch < c) {
            // eight rows in flight per lane (the loop is a chain of dependent-looking loads otherwise: T / 4 round trips)
            const float* col = x + (int64_t)b * T_pitch * c + ch;
            int t = wave;
            for (; t + 28 < T; t += 32) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = col[(int64_t)(t + 4 * u) * c];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; t < T; t += 4) s += col[(int64_t)t * c];
        }
        part[wave][lane] = s;
        __syncthreads();
        if (wave == 0 && ch < c) {
            const float m = (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) / (float)T;
            a0 = fmaf(m, fcw[ch], a0);
            a1 = fmaf(m, fcw[c + ch], a1);
        }
        __syncthreads();
    }
    if (wave != 0) return;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a0 += __shfl_xor(a0, d, 64);
        a1 += __shfl_xor(a1, d, 64);
    }
    if (lane == 0) {
        const float l0 = a0 + fcb[0], l1 = a1 + fcb[1];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        probs[2 * b] = e0 / (e0 + e1);
        probs[2 * b + 1] = e1 / (e0 + e1);
        if (logits) {
            logits[2 * b] = l0;
            logits[2 * b + 1] = l1;
        }
    }
}

// Ragged batches (rs_seqnet_forward_ragged): table[k + 1][b] = rows of read b after op k, table[0][b] = its samples.  One thread per
// read walks the program's ops (a conv: (T + 2 pad - k) / stride + 1, or 0 when the kernel does not fit; MaxPool1d(2, 2, pad)).
struct LenRecipe {
    int n_ops;
    signed char kind[64], pad[64];
    short k[64], stride[64], prod[64];        // prod: the op that wrote this op's input buffer, -1 = the program's input
};
__global__ __launch_bounds__(256) void seq_lengths_kernel(const int32_t* __restrict__ len, int B, int ld, const LenRecipe rc,
                                                          int32_t* __restrict__ table) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    table[b] = min(max(len[b], 0), ld);                      // a length beyond the row pitch would read the next read's row
    for (int i = 0; i < rc.n_ops; ++i) {
        const int t = table[(size_t)(rc.prod[i] + 1) * B + b];
        int o;
        if (rc.kind[i] == 0)
            o = t + 2 * rc.pad[i] < rc.k[i] ? 0 : (t + 2 * rc.pad[i] - rc.k[i]) / rc.stride[i] + 1;
        else
            o = t <= 0 ? 0 : (rc.pad[i] ? t / 2 + 1 : t / 2);
        table[(size_t)(i + 1) * B + b] = o;
    }
}

struct OpDev {
    int kind, src, dst, add;
    int c_in, c_out, k, stride, pad, relu;
    float* d_w = nullptr;     // [k][c_in][cq*4]
    float* d_b = nullptr;     // [cq*4]
    float* d_wq = nullptr;    // MFMA packing [K16 / 4][Npad][4] (K = k * c_in in im2col order), or null if too large for LDS
    int nt = 0;               // Npad / 16
    // fusion (rs_seqnet_create, fuse_program): this op starts a fused launch that also covers the next `fuse_skip` ops
    int fuse = 0;             // 0 none, 1 stem conv + max-pool, 2 residual basic block
    int fuse_skip = 0;
    int f_src = -1, f_dst = -1;
    int f_cin = 0, f_cout = 0, f_cp = 0, f_stride = 1, f_ksc = 0;
    int f_nt = 0, f_nps = 0;
    float* d_f_w1 = nullptr;  // block: first conv [K1_16 / 4][Npad][4]
    float* d_f_w2 = nullptr;  // block: second conv (+ shortcut conv) [K2_16 / 4][Npad][4]
    float* d_f_b1 = nullptr;  // [Npad]
    float* d_f_b2 = nullptr;  // [Npad]
    size_t f_wfloats = 0;     // floats of the weight matrices in LDS
    // the same block in split precision (seq_basic_block_x3_kernel): bf16 [hi | lo] planes, own row / column pitches
    unsigned short* d_x_w1 = nullptr;   // (fuse == 1, the stem: its one weight matrix)
    unsigned short* d_x_w2 = nullptr;
    unsigned short* d_x_w3 = nullptr;   // (fuse == 3, the bottleneck block: third conv + shortcut)
    int x_npm = 0, x_s3 = 0;            // bottleneck: mid column pitch, k-steps of the third conv
    int x_cp = 0, x_np = 0, x_s1 = 0, x_s2a = 0, x_ssc = 0;
    size_t x_wbytes = 0;      // bytes of both weight matrices in LDS
    // bottleneck block (fuse == 3): third conv (+ shortcut), mid width
    float* d_f_w3 = nullptr;
    float* d_f_b3 = nullptr;
    int f_cmid = 0, f_ntm = 0, f_npm = 0;
};

}  // namespace
}  // namespace rs

using namespace rs;

struct rs_seqnet {
    int64_t window = 0x7fffffffLL; // bytes a kernel addresses through one 32-bit-offset buffer window (RS_SEQ_WINDOW_BYTES at create:
                                   // tests force the size guards of the fused launches with a small one)
    bool fuse = true;              // RS_SEQ_NOFUSE=1 (read at create): one launch per op, as the program is written
    int mode = 0;                  // rs_seqnet_set_mode: 0 fp32 (f32-input MFMA), 1 split precision on the bf16 MFMA
    bool bneck_x3 = false;         // RS_SEQ_BNECK_X3=1 (read at create): bottleneck blocks in split precision too.  Off: measured
                                   // SLOWER than their fp32 form (0.734 against 0.718 ms on the 32-48-68 net, stem included) - with
                                   // mid widths of 8-17 channels a block's GEMMs are a few k-steps and the split costs more VALU
                                   // time than the matrix pipe saves
    std::vector<std::vector<float>> keep_w;   // host copies of the conv weights until fusion has packed them
    int device = 0;
    int n_buffers = 0;
    int c_last = 0;
    bool scalar_conv = false;      // RS_SEQ_SCALAR=1 (read at create): the scalar FMA conv kernel instead of the MFMA one
    int num_cu = 256;
    std::vector<OpDev> ops;
    float* d_fcw = nullptr;
    float* d_fcb = nullptr;
};

namespace {

// lengths and channel widths of every buffer for an input of L samples (buffer 0 = the input)
// (buffer ids are reused by later ops with other shapes: *max_elems is the largest T*C any buffer
// ever holds, which sizes the workspace regions)
struct OpShape {
    int t_in, t_out, c;
};

bool propagate(const rs_seqnet* m, int L, std::vector<int>& T, std::vector<int>& C, size_t* max_elems = nullptr,
               std::vector<OpShape>* shapes = nullptr) {
    if (max_elems) *max_elems = 0;
    if (shapes) shapes->clear();
    T.assign(m->n_buffers, -1);
    C.assign(m->n_buffers, 0);
    T[0] = L;
    C[0] = 1;
    for (const OpDev& o : m->ops) {
        if (T[o.src] < 0) return false;
        int t_out;
        if (o.kind == 0) {
            if (T[o.src] + 2 * o.pad < o.k) return false;          // torch: kernel larger than padded input
            t_out = (T[o.src] + 2 * o.pad - o.k) / o.stride + 1;
            C[o.dst] = o.c_out;
        } else {
            t_out = o.pad ? T[o.src] / 2 + 1 : T[o.src] / 2;      // MaxPool1d(2, 2, padding 1 | 0)
            C[o.dst] = C[o.src];
        }
        if (t_out < 1) return false;
        if (o.add >= 0 && (T[o.add] != t_out || C[o.add] != C[o.dst])) return false;
        if (shapes) shapes->push_back({T[o.src], t_out, C[o.dst]});
        T[o.dst] = t_out;
        if (max_elems) *max_elems = std::max(*max_elems, (size_t)t_out * C[o.dst]);
    }
    return true;
}

// is buffer `buf` dead after op index `after` (never read again before it is rewritten)?
bool dead_after(const std::vector<OpDev>& ops, size_t after, int buf) {
    for (size_t k = after + 1; k < ops.size(); ++k) {
        if (ops[k].src == buf || ops[k].add == buf) return false;
        if (ops[k].dst == buf) return true;
    }
    return true;
}

// float -> bf16, round to nearest even (what v_cvt_pk_bf16_f32 does for finite values)
unsigned short bf16_rne(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);       // inf / nan: truncate
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
float bf16_widen(unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

hipError_t upload_u16(unsigned short** d, const std::vector<unsigned short>& h) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(d), std::max<size_t>(h.size(), 8) * sizeof(unsigned short));
    if (e == hipSuccess) e = hipMemcpy(*d, h.data(), h.size() * sizeof(unsigned short), hipMemcpyHostToDevice);
    return e;
}

template <class T>
hipError_t upload_vec(float** d, const std::vector<T>& h) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(d), std::max<size_t>(h.size(), 1) * sizeof(T));
    if (e == hipSuccess) e = hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

// Recognise what riser_amd/resnet.py:build_program emits for the reference's stem (resnet.py:79-84) and BasicBlock
// (resnet.py:50-57 + shortcut :21-24,45-47) and mark the first op of each pattern as a fused launch.  hw[k] = host copy
// of op k's [c_out][c_in][k] weights (conv ops), hb[k] its bias.
hipError_t fuse_program(rs_seqnet* m, const std::vector<const float*>& hw, const std::vector<const float*>& hb) {
    std::vector<OpDev>& ops = m->ops;
    const size_t lds_cap = 160 * 1024;
    for (size_t k = 0; k < ops.size(); ++k) {
        OpDev& o = ops[k];
        if (o.kind != 0) continue;
        // ---- stem: conv(1 -> C) + ReLU, then MaxPool1d(2, 2, padding 1) of it -------------------------------------------
        if (o.c_in == 1 && o.relu && o.add < 0 && o.d_wq && k + 1 < ops.size() && ops[k + 1].kind == 1 && ops[k + 1].pad == 1 &&
            ops[k + 1].src == o.dst && dead_after(ops, k + 1, o.dst)) {
            o.fuse = 1;
            o.fuse_skip = 1;
            o.f_src = o.src;
            o.f_dst = ops[k + 1].dst;
            {   // the split-precision packing (seq_stem_pool_x3_kernel): planes [hi | lo] of [k-step][kq][16 nt][8]
                const int K = o.k, S = (K + 31) / 32, NPx = 16 * o.nt;
                const size_t plane = (size_t)S * 4 * NPx * 8;
                std::vector<unsigned short> xw(2 * plane, 0);
                for (int co = 0; co < o.c_out; ++co)
                    for (int kk = 0; kk < K; ++kk) {
                        const float w = hw[k][(size_t)co * K + kk];
                        const size_t at = (((size_t)(kk / 32) * 4 + (kk % 32) / 8) * NPx + co) * 8 + kk % 8;
                        const unsigned short h = bf16_rne(w);
                        xw[at] = h;
                        xw[plane + at] = bf16_rne(w - bf16_widen(h));
                    }
                const hipError_t e = upload_u16(&o.d_x_w1, xw);
                if (e != hipSuccess) return e;
                o.x_s1 = S;
            }
            continue;
        }
        // ---- basic block: [1x1 shortcut conv]  conv3(stride) + ReLU  conv3 + residual + ReLU -----------------------------
        size_t k1 = k, ksc = (size_t)-1;
        if (o.k == 1 && o.pad == 0 && !o.relu && o.add < 0 && k + 2 < ops.size()) {
            ksc = k;
            k1 = k + 1;
        }
        if (k1 + 1 >= ops.size()) continue;
        const OpDev& c1 = ops[k1];
        const OpDev& c2 = ops[k1 + 1];
        const int X = ksc != (size_t)-1 ? ops[ksc].src : c1.src;
        const int res = ksc != (size_t)-1 ? ops[ksc].dst : X;
        const bool shape_ok = c1.kind == 0 && c2.kind == 0 && c1.k == 3 && c1.pad == 1 && c1.relu && c1.add < 0 && c1.src == X &&
                              c2.k == 3 && c2.pad == 1 && c2.stride == 1 && c2.relu && c2.src == c1.dst && c2.add == res &&
                              c2.c_in == c1.c_out && c2.c_out == c1.c_out && c2.dst != X && c1.dst != X &&
                              (ksc != (size_t)-1 ? (ops[ksc].stride == c1.stride && ops[ksc].c_in == c1.c_in && ops[ksc].c_out == c1.c_out)
                                                 : (c1.stride == 1 && c1.c_in == c1.c_out));
        if (!shape_ok) {
            // ---- bottleneck block: [1x1 shortcut conv]  conv1 + ReLU  conv3(stride) + ReLU  conv1 + residual + ReLU -------
            if (k1 + 2 >= ops.size()) continue;
            const OpDev& d1 = ops[k1];
            const OpDev& d2 = ops[k1 + 1];
            const OpDev& d3 = ops[k1 + 2];
            const bool bn_ok = d1.kind == 0 && d2.kind == 0 && d3.kind == 0 && d1.k == 1 && d1.pad == 0 && d1.stride == 1 && d1.relu &&
                               d1.add < 0 && d1.src == X && d2.k == 3 && d2.pad == 1 && d2.relu && d2.add < 0 && d2.src == d1.dst &&
                               d2.c_in == d1.c_out && d2.c_out == d1.c_out && d3.k == 1 && d3.pad == 0 && d3.stride == 1 && d3.relu &&
                               d3.src == d2.dst && d3.add == res && d3.c_in == d2.c_out && d3.dst != X && d1.dst != X && d2.dst != X &&
                               (ksc != (size_t)-1 ? (ops[ksc].stride == d2.stride && ops[ksc].c_in == d1.c_in && ops[ksc].c_out == d3.c_out)
                                                  : (d2.stride == 1 && d1.c_in == d3.c_out));
            if (!bn_ok || (d2.stride != 1 && d2.stride != 2)) continue;
            if (!dead_after(ops, k1 + 2, d1.dst) || !dead_after(ops, k1 + 2, d2.dst) ||
                (ksc != (size_t)-1 && !dead_after(ops, k1 + 2, res)))
                continue;
            const int c_in = d1.c_in, c_mid = d1.c_out, c_out = d3.c_out;
            const int ntm = (c_mid + 15) / 16, nto = (c_out + 15) / 16;
            if (ntm > 2 || nto > 5) continue;
            int Cmp = (c_mid + 3) & ~3;
            if (((Cmp / 4) & 1) == 0) Cmp += 4;
            const int NPm = (c_mid + 3) & ~3, NPo = (c_out + 3) & ~3;
            const int K1_16 = (c_in + 15) & ~15, K2_16 = (3 * Cmp + 15) & ~15, K3_16 = (Cmp + 15) & ~15;
            const int Ksc = ksc != (size_t)-1 ? c_in : 0, Ksc16 = (Ksc + 15) & ~15;
            const size_t w_floats = (size_t)(K1_16 + K2_16) * NPm + (size_t)(K3_16 + Ksc16) * NPo;
            if ((w_floats + (size_t)2 * (128 + 4) * Cmp) * 4 > lds_cap) continue;
            std::vector<float> w1q((size_t)K1_16 * NPm, 0.0f), w2q((size_t)K2_16 * NPm, 0.0f), w3q((size_t)(K3_16 + Ksc16) * NPo, 0.0f);
            std::vector<float> b1(16 * ntm, 0.0f), b2(16 * ntm, 0.0f), b3(16 * nto, 0.0f);
            for (int co = 0; co < c_mid; ++co) {
                b1[co] = hb[k1][co];
                b2[co] = hb[k1 + 1][co];
                for (int ci = 0; ci < c_in; ++ci) w1q[((size_t)(ci / 4) * NPm + co) * 4 + ci % 4] = hw[k1][(size_t)co * c_in + ci];
                for (int ci = 0; ci < c_mid; ++ci)
                    for (int kk = 0; kk < 3; ++kk) {
                        const int kidx = kk * Cmp + ci;
                        w2q[((size_t)(kidx / 4) * NPm + co) * 4 + kidx % 4] = hw[k1 + 1][((size_t)co * c_mid + ci) * 3 + kk];
                    }
            }
            for (int co = 0; co < c_out; ++co) {
                b3[co] = hb[k1 + 2][co] + (ksc != (size_t)-1 ? hb[ksc][co] : 0.0f);
                for (int ci = 0; ci < c_mid; ++ci) w3q[((size_t)(ci / 4) * NPo + co) * 4 + ci % 4] = hw[k1 + 2][(size_t)co * c_mid + ci];
                if (ksc != (size_t)-1)
                    for (int ci = 0; ci < c_in; ++ci) {
                        const int kidx = K3_16 + ci;
                        w3q[((size_t)(kidx / 4) * NPo + co) * 4 + kidx % 4] = hw[ksc][(size_t)co * c_in + ci];
                    }
            }
            hipError_t e = upload_vec(&o.d_f_w1, w1q);
            if (e == hipSuccess) e = upload_vec(&o.d_f_w2, w2q);
            if (e == hipSuccess) e = upload_vec(&o.d_f_w3, w3q);
            if (e == hipSuccess) e = upload_vec(&o.d_f_b1, b1);
            if (e == hipSuccess) e = upload_vec(&o.d_f_b2, b2);
            if (e == hipSuccess) e = upload_vec(&o.d_f_b3, b3);
            if (e != hipSuccess) return e;
            {   // the split-precision packing (seq_bottleneck_block_x3_kernel)
                int Cmx = (c_mid + 7) & ~7;
                while (Cmx % 16 != 8) Cmx += 8;
                const int S1 = (c_in + 31) / 32, S2 = (3 * Cmx + 31) / 32, S3 = (Cmx + 31) / 32, Sscx = (Ksc + 31) / 32;
                const size_t q1 = (size_t)S1 * 4 * NPm * 8, q2 = (size_t)S2 * 4 * NPm * 8, q3 = (size_t)(S3 + Sscx) * 4 * NPo * 8;
                const size_t xbytes = (q1 + q2 + q3) * 2 * 2;
                if (xbytes + (size_t)2 * (128 + 4) * Cmx * 4 <= lds_cap) {
                    std::vector<unsigned short> x1(2 * q1, 0), x2(2 * q2, 0), x3v(2 * q3, 0);
                    auto put = [&](std::vector<unsigned short>& dst, size_t plane, int NP, int kidx, int n, float w) {
                        const size_t at = (((size_t)(kidx / 32) * 4 + (kidx % 32) / 8) * NP + n) * 8 + kidx % 8;
                        const unsigned short h = bf16_rne(w);
                        dst[at] = h;
                        dst[plane + at] = bf16_rne(w - bf16_widen(h));
                    };
                    for (int co = 0; co < c_mid; ++co) {
                        for (int ci = 0; ci < c_in; ++ci) put(x1, q1, NPm, ci, co, hw[k1][(size_t)co * c_in + ci]);
                        for (int ci = 0; ci < c_mid; ++ci)
                            for (int kk = 0; kk < 3; ++kk) put(x2, q2, NPm, kk * Cmx + ci, co, hw[k1 + 1][((size_t)co * c_mid + ci) * 3 + kk]);
                    }
                    for (int co = 0; co < c_out; ++co) {
                        for (int ci = 0; ci < c_mid; ++ci) put(x3v, q3, NPo, ci, co, hw[k1 + 2][(size_t)co * c_mid + ci]);
                        if (ksc != (size_t)-1)
                            for (int ci = 0; ci < c_in; ++ci) put(x3v, q3, NPo, 32 * S3 + ci, co, hw[ksc][(size_t)co * c_in + ci]);
                    }
                    e = upload_u16(&o.d_x_w1, x1);
                    if (e == hipSuccess) e = upload_u16(&o.d_x_w2, x2);
                    if (e == hipSuccess) e = upload_u16(&o.d_x_w3, x3v);
                    if (e != hipSuccess) return e;
                    o.x_cp = Cmx;
                    o.x_s1 = S1;
                    o.x_s2a = S2;
                    o.x_s3 = S3;
                    o.x_ssc = Sscx;
                    o.x_wbytes = xbytes;
                }
            }
            o.fuse = 3;
            o.fuse_skip = (int)(k1 + 2 - k);
            o.f_src = X;
            o.f_dst = d3.dst;
            o.f_cin = c_in;
            o.f_cmid = c_mid;
            o.f_cout = c_out;
            o.f_cp = Cmp;
            o.f_npm = NPm;
            o.f_nps = NPo;
            o.f_stride = d2.stride;
            o.f_ksc = Ksc;
            o.f_ntm = ntm;
            o.f_nt = nto;
            o.f_wfloats = w_floats;
            k = k1 + 2;
            continue;
        }
        if (!dead_after(ops, k1 + 1, c1.dst) || (ksc != (size_t)-1 && !dead_after(ops, k1 + 1, res))) continue;
        const int c_in = c1.c_in, c_out = c1.c_out, nt = (c_out + 15) / 16;
        if (nt > 5) continue;
        // column pitch of the weight matrices in LDS: the compact one (c_out rounded to 4) only where the full 16 * nt would
        // cost the block its second workgroup per CU - it costs a select per weight fragment
        int NP = 16 * nt;
        if ((size_t)(((3 * c_in + 15) & ~15) + ((3 * (c_out + 7) + 15) & ~15) + ((c_in + 15) & ~15)) * NP * 4 > 60 * 1024) NP = (c_out + 3) & ~3;
        int Cp = (c_out + 3) & ~3;
        if (((Cp / 4) & 1) == 0) Cp += 4;                      // row pitch = 4 (mod 8) floats: conflict-free ds_read_b128 over 16 rows
        const int K1 = 3 * c_in, K1_16 = (K1 + 15) & ~15, K2a = 3 * Cp, K2a16 = (K2a + 15) & ~15;
        const int Ksc = ksc != (size_t)-1 ? c_in : 0, Ksc16 = (Ksc + 15) & ~15;
        const size_t w_floats = (size_t)(K1_16 + K2a16 + Ksc16) * NP;
        // 126 outputs per tile (two 16-row tiles per wave: half the halo, half the weight reads per MFMA) unless the 62-output
        // tile is what lets a second workgroup share the CU
        if ((w_floats + (size_t)(64 + 4) * Cp) * 4 > lds_cap) continue;       // not even the smallest tile fits
        // pack: conv 1 as the unfused kernel does; conv 2 over the LDS tile's K index tap * Cp + c, the shortcut behind it
        std::vector<float> w1q((size_t)K1_16 * NP, 0.0f), w2q((size_t)(K2a16 + Ksc16) * NP, 0.0f), b1(16 * nt, 0.0f), b2(16 * nt, 0.0f);
        for (int co = 0; co < c_out; ++co) {
            b1[co] = hb[k1][co];
            b2[co] = hb[k1 + 1][co] + (ksc != (size_t)-1 ? hb[ksc][co] : 0.0f);
            for (int ci = 0; ci < c_in; ++ci)
                for (int kk = 0; kk < 3; ++kk) {
                    const int kidx = kk * c_in + ci;
                    w1q[((size_t)(kidx / 4) * NP + co) * 4 + kidx % 4] = hw[k1][((size_t)co * c_in + ci) * 3 + kk];
                }
            for (int ci = 0; ci < c_out; ++ci)
                for (int kk = 0; kk < 3; ++kk) {
                    const int kidx = kk * Cp + ci;
                    w2q[((size_t)(kidx / 4) * NP + co) * 4 + kidx % 4] = hw[k1 + 1][((size_t)co * c_out + ci) * 3 + kk];
                }
            if (ksc != (size_t)-1)
                for (int ci = 0; ci < c_in; ++ci) {
                    const int kidx = K2a16 + ci;
                    w2q[((size_t)(kidx / 4) * NP + co) * 4 + kidx % 4] = hw[ksc][(size_t)co * c_in + ci];
                }
        }
        hipError_t e = upload_vec(&o.d_f_w1, w1q);
        if (e == hipSuccess) e = upload_vec(&o.d_f_w2, w2q);
        if (e == hipSuccess) e = upload_vec(&o.d_f_b1, b1);
        if (e == hipSuccess) e = upload_vec(&o.d_f_b2, b2);
        if (e != hipSuccess) return e;
        // ---- the split-precision packing of the same block (seq_basic_block_x3_kernel): k-steps of 32, bf16 planes
        // [hi | lo] of [k-step][kq][n][8], tile row pitch Cpx = 8 (mod 16) halfwords ------------------------------------------
        {
            int Cpx = (c_out + 7) & ~7;
            while (Cpx % 16 != 8) Cpx += 8;
            const int S1 = (3 * c_in + 31) / 32, S2a = (3 * Cpx + 31) / 32, Ssc = (Ksc + 31) / 32;
            int NPx = 16 * nt;
            if ((size_t)(S1 + S2a + Ssc) * 128 * NPx > 60 * 1024) NPx = (c_out + 3) & ~3;
            const size_t wbytes = (size_t)(S1 + S2a + Ssc) * 128 * NPx;
            if (wbytes + (size_t)(64 + 4) * Cpx * 4 <= lds_cap) {
                const size_t p1 = (size_t)S1 * 4 * NPx * 8, p2 = (size_t)(S2a + Ssc) * 4 * NPx * 8;
                std::vector<unsigned short> x1(2 * p1, 0), x2(2 * p2, 0);
                auto put = [&](std::vector<unsigned short>& dst, size_t plane, int kidx, int n, float w) {
                    const size_t at = (((size_t)(kidx / 32) * 4 + (kidx % 32) / 8) * NPx + n) * 8 + kidx % 8;
                    const unsigned short h = bf16_rne(w);
                    dst[at] = h;
                    dst[plane + at] = bf16_rne(w - bf16_widen(h));
                };
                for (int co = 0; co < c_out; ++co) {
                    for (int ci = 0; ci < c_in; ++ci)
                        for (int kk = 0; kk < 3; ++kk) put(x1, p1, kk * c_in + ci, co, hw[k1][((size_t)co * c_in + ci) * 3 + kk]);
                    for (int ci = 0; ci < c_out; ++ci)
                        for (int kk = 0; kk < 3; ++kk) put(x2, p2, kk * Cpx + ci, co, hw[k1 + 1][((size_t)co * c_out + ci) * 3 + kk]);
                    if (ksc != (size_t)-1)
                        for (int ci = 0; ci < c_in; ++ci) put(x2, p2, 32 * S2a + ci, co, hw[ksc][(size_t)co * c_in + ci]);
                }
                e = upload_u16(&o.d_x_w1, x1);
                if (e == hipSuccess) e = upload_u16(&o.d_x_w2, x2);
                if (e != hipSuccess) return e;
                o.x_cp = Cpx;
                o.x_np = NPx;
                o.x_s1 = S1;
                o.x_s2a = S2a;
                o.x_ssc = Ssc;
                o.x_wbytes = wbytes;
            }
        }
        o.fuse = 2;
        o.fuse_skip = (int)(k1 + 1 - k);
        o.f_src = X;
        o.f_dst = c2.dst;
        o.f_cin = c_in;
        o.f_cout = c_out;
        o.f_cp = Cp;
        o.f_nps = NP;
        o.f_stride = c1.stride;
        o.f_ksc = Ksc;
        o.f_nt = nt;
        o.f_wfloats = w_floats;
        k = k1 + 1;
    }
    return hipSuccess;
}

size_t buffer_bytes(const rs_seqnet* m, int B, int L) {
    std::vector<int> T, C;
    size_t elems = 0;
    if (!propagate(m, L, T, C, &elems)) return 0;
    return ((size_t)B * elems * sizeof(float) + 255) / 256 * 256;
}

}  // namespace

extern "C" {

int rs_seqnet_create(const rs_seq_op* ops, int n_ops, int n_buffers, const float* fc_w, const float* fc_b, int c_last,
                     int device, rs_seqnet** out) {
    if (!ops || n_ops < 1 || n_buffers < 2 || n_buffers > 16 || !fc_w || !fc_b || !out || c_last < 1) {
        set_error("rs_seqnet_create: bad argument");
        return RS_ERR_ARG;
    }
    *out = nullptr;
    DeviceGuard guard(device);            // the caller's current device is restored on return
    RS_HIP(guard.err);
    rs_seqnet* m = new (std::nothrow) rs_seqnet();
    if (!m) return RS_ERR_OOM;
    m->device = device;
    m->n_buffers = n_buffers;
    m->c_last = c_last;
    m->scalar_conv = getenv("RS_SEQ_SCALAR") != nullptr;
    if (const char* e = getenv("RS_SEQ_WINDOW_BYTES"); e && atoll(e) > 0) m->window = std::min<int64_t>(atoll(e), 0x7fffffffLL);
    m->fuse = getenv("RS_SEQ_NOFUSE") == nullptr && !m->scalar_conv;
    m->bneck_x3 = getenv("RS_SEQ_BNECK_X3") != nullptr;
    std::vector<const float*> hw, hb;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) m->num_cu = cus;
    }
    for (int i = 0; i < n_ops; ++i) {
        const rs_seq_op& s = ops[i];
        OpDev o;
        o.kind = s.kind; o.src = s.src; o.dst = s.dst; o.add = s.add;
        o.c_in = s.c_in; o.c_out = s.c_out; o.k = s.k; o.stride = s.stride; o.pad = s.pad; o.relu = s.relu;
        const bool bad_buf = s.src < 0 || s.src >= n_buffers || s.dst < 1 || s.dst >= n_buffers || s.dst == s.src ||
                             s.add >= n_buffers || s.add == s.dst;
        if (bad_buf || (s.kind != 0 && s.kind != 1) || (s.kind == 1 && s.pad != 0 && s.pad != 1) ||
            (s.kind == 0 && (!s.w || !s.b || s.c_in < 1 || s.c_out < 1 || s.k < 1 || s.stride < 1 || s.pad < 0))) {
            rs_seqnet_destroy(m);
            set_error("rs_seqnet_create: bad op %d", i);
            return RS_ERR_ARG;
        }
        if (s.kind == 0) {
            const int cq = (s.c_out + 3) / 4;
            std::vector<float> wp((size_t)s.k * s.c_in * cq * 4, 0.0f), bp((size_t)cq * 4, 0.0f);
            for (int co = 0; co < s.c_out; ++co) {
                bp[co] = s.b[co];
                for (int ci = 0; ci < s.c_in; ++ci)
                    for (int kk = 0; kk < s.k; ++kk)
                        wp[((size_t)kk * s.c_in + ci) * cq * 4 + co] = s.w[((size_t)co * s.c_in + ci) * s.k + kk];
            }
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&o.d_w), wp.size() * 4);
            if (e == hipSuccess) e = hipMemcpy(o.d_w, wp.data(), wp.size() * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&o.d_b), bp.size() * 4);
            if (e == hipSuccess) e = hipMemcpy(o.d_b, bp.data(), bp.size() * 4, hipMemcpyHostToDevice);
            // MFMA packing: element (kidx, n) of the im2col GEMM at [kidx / 4][n][kidx % 4], K padded to 16, N to 16
            const int K = s.k * s.c_in, K16 = (K + 15) & ~15, nt = (s.c_out + 15) / 16;
            if (e == hipSuccess && nt <= 5 && (size_t)K16 * nt * 16 * 4 <= 96 * 1024) {
                std::vector<float> wq((size_t)K16 * nt * 16, 0.0f);
                for (int co = 0; co < s.c_out; ++co)
                    for (int ci = 0; ci < s.c_in; ++ci)
                        for (int kk = 0; kk < s.k; ++kk) {
                            const int kidx = kk * s.c_in + ci;
                            wq[((size_t)(kidx / 4) * nt * 16 + co) * 4 + kidx % 4] = s.w[((size_t)co * s.c_in + ci) * s.k + kk];
                        }
                e = hipMalloc(reinterpret_cast<void**>(&o.d_wq), wq.size() * 4);
                if (e == hipSuccess) e = hipMemcpy(o.d_wq, wq.data(), wq.size() * 4, hipMemcpyHostToDevice);
                o.nt = nt;
            }
            m->ops.push_back(o);
            hw.push_back(s.w);
            hb.push_back(s.b);
            if (e != hipSuccess) {
                rs_seqnet_destroy(m);
                return hip_fail(e, "rs_seqnet_create upload");
            }
        } else {
            m->ops.push_back(o);
            hw.push_back(nullptr);
            hb.push_back(nullptr);
        }
    }
    if (m->fuse) {
        const hipError_t fe = fuse_program(m, hw, hb);
        if (fe != hipSuccess) {
            rs_seqnet_destroy(m);
            return hip_fail(fe, "rs_seqnet_create fused packing");
        }
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->d_fcw), (size_t)2 * c_last * 4);
    if (e == hipSuccess) e = hipMemcpy(m->d_fcw, fc_w, (size_t)2 * c_last * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&m->d_fcb), 8);
    if (e == hipSuccess) e = hipMemcpy(m->d_fcb, fc_b, 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        rs_seqnet_destroy(m);
        return hip_fail(e, "rs_seqnet_create upload");
    }
    *out = m;
    return RS_OK;
}

int rs_seqnet_destroy(rs_seqnet* m) {
    if (!m) return RS_OK;
    DeviceGuard guard(m->device);
    for (OpDev& o : m->ops) {
        if (o.d_w) (void)hipFree(o.d_w);
        if (o.d_b) (void)hipFree(o.d_b);
        if (o.d_wq) (void)hipFree(o.d_wq);
        if (o.d_f_w3) (void)hipFree(o.d_f_w3);
        if (o.d_f_b3) (void)hipFree(o.d_f_b3);
        if (o.d_x_w1) (void)hipFree(o.d_x_w1);
        if (o.d_x_w2) (void)hipFree(o.d_x_w2);
        if (o.d_x_w3) (void)hipFree(o.d_x_w3);
        if (o.d_f_w1) (void)hipFree(o.d_f_w1);
        if (o.d_f_w2) (void)hipFree(o.d_f_w2);
        if (o.d_f_b1) (void)hipFree(o.d_f_b1);
        if (o.d_f_b2) (void)hipFree(o.d_f_b2);
    }
    if (m->d_fcw) (void)hipFree(m->d_fcw);
    if (m->d_fcb) (void)hipFree(m->d_fcb);
    delete m;
    return RS_OK;
}

int rs_seqnet_set_mode(rs_seqnet* m, int dtype) {
    if (!m) {
        set_error("rs_seqnet_set_mode: null program");
        return RS_ERR_ARG;
    }
    if (dtype == RS_F32 || dtype == RS_F32W) {
        m->mode = 0;
        return RS_OK;
    }
    if (dtype != RS_BF16X3) {
        set_error("rs_seqnet_set_mode: generic conv programs run in RS_F32 or RS_BF16X3 (split precision on the bf16 MFMA)");
        return RS_ERR_ARG;
    }
    // split precision covers the stem and the residual BASIC blocks of a program (where a ResNet's time is), and its bottleneck
    // blocks when RS_SEQ_BNECK_X3 was set at create (measured slower than their fp32 form: off); the head and unfused ops keep
    // fp32.  A program without a single fused residual block has nothing to switch.
    bool any = false;
    for (const OpDev& o : m->ops) any = any || ((o.fuse == 2 || o.fuse == 3) && o.d_x_w1);
    if (!any) {
        set_error("rs_seqnet_set_mode: this program has no residual block that runs in split precision");
        return RS_ERR_ARG;
    }
    m->mode = 1;
    return RS_OK;
}

size_t rs_seqnet_workspace_bytes(const rs_seqnet* m, int B, int L) {
    if (!m || B < 1 || L < 1) return 0;
    const size_t per = buffer_bytes(m, B, L);
    if (!per) return 0;
    // the activation buffers, then the per-read length table of a ragged forward ((ops + 1) x B)
    return per * (size_t)(m->n_buffers - 1) + ((m->ops.size() + 1) * (size_t)B * 4 + 255) / 256 * 256;
}

static int seqnet_forward_impl(rs_seqnet* m, const float* d_x, const int32_t* d_len, int B, int L, void* d_ws, size_t ws_bytes,
                               float* d_probs, float* d_logits, void* stream);

int rs_seqnet_max_batch(const rs_seqnet* m, int L) {
    // every activation buffer of B reads stays inside one buffer window: B x (the largest buffer of one read) bytes
    if (!m || L < 1) return 0;
    const size_t per = buffer_bytes(m, 1, L);
    if (!per) return 0;
    return (int)std::max<int64_t>(1, std::min<int64_t>(1 << 30, (m->window - 4096) / (int64_t)per));
}

int rs_seqnet_forward(rs_seqnet* m, const float* d_x, int B, int L, void* d_ws, size_t ws_bytes, float* d_probs,
                      float* d_logits, void* stream) {
    return seqnet_forward_impl(m, d_x, nullptr, B, L, d_ws, ws_bytes, d_probs, d_logits, stream);
}

int rs_seqnet_forward_ragged(rs_seqnet* m, const float* d_x, const int32_t* d_len, int B, int ld, void* d_ws, size_t ws_bytes,
                             float* d_probs, float* d_logits, void* stream) {
    if (!d_len) {
        set_error("rs_seqnet_forward_ragged: null lengths");
        return RS_ERR_ARG;
    }
    return seqnet_forward_impl(m, d_x, d_len, B, ld, d_ws, ws_bytes, d_probs, d_logits, stream);
}

// 1 when every op of the program runs inside a fused launch (stem, residual blocks): what a ragged forward needs
int rs_seqnet_ragged_ok(const rs_seqnet* m) {
    if (!m) return 0;
    for (size_t k = 0; k < m->ops.size(); ++k) {
        const OpDev& o = m->ops[k];
        if (o.fuse < 1 || o.fuse > 3) return 0;
        k += o.fuse_skip;
    }
    return m->ops.size() <= 63 ? 1 : 0;
}

static int seqnet_forward_impl(rs_seqnet* m, const float* d_x, const int32_t* d_len, int B, int L, void* d_ws, size_t ws_bytes,
                               float* d_probs, float* d_logits, void* stream) {
    if (!m || !d_x || !d_ws || !d_probs || B < 1 || L < 1) {
        set_error("rs_seqnet_forward: bad argument");
        return RS_ERR_ARG;
    }
    if (d_len && !rs_seqnet_ragged_ok(m)) {
        set_error("rs_seqnet_forward_ragged: this program has ops outside its fused launches (stem, residual blocks): group the "
                  "reads by length and call rs_seqnet_forward");
        return RS_ERR_ARG;
    }
    std::vector<int> T, C;
    std::vector<OpShape> shp;                                      // buffer ids are reused: shapes are per op
    if (!propagate(m, L, T, C, nullptr, &shp)) {
        set_error("rs_seqnet_forward: input of %d samples is too short for this network", L);
        return RS_ERR_LENGTH;
    }
    const size_t per = buffer_bytes(m, B, L);
    if (ws_bytes < rs_seqnet_workspace_bytes(m, B, L)) {
        set_error("rs_seqnet_forward: workspace too small");
        return RS_ERR_WORKSPACE;
    }
    DeviceGuard guard(m->device);
    RS_HIP(guard.err);
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto buf = [&](int i) -> float* {
        return i == 0 ? const_cast<float*>(d_x) : reinterpret_cast<float*>(static_cast<char*>(d_ws) + per * (size_t)(i - 1));
    };
    // ragged batch: the rows of every read after every op, computed on the device from its length (no host copy needed);
    // L is then the row pitch of d_x and every buffer keeps the pitches of an L-sample read
    int32_t* table = nullptr;
    std::vector<int> prod(m->ops.size(), -1);
    if (d_len) {
        table = reinterpret_cast<int32_t*>(static_cast<char*>(d_ws) + per * (size_t)(m->n_buffers - 1));
        LenRecipe rc;
        memset(&rc, 0, sizeof(rc));
        rc.n_ops = (int)m->ops.size();
        for (size_t k = 0; k < m->ops.size(); ++k) {
            const OpDev& o = m->ops[k];
            for (size_t j = 0; j < k; ++j)
                if (m->ops[j].dst == o.src) prod[k] = (int)j;
            rc.kind[k] = (signed char)o.kind;
            rc.pad[k] = (signed char)o.pad;
            rc.k[k] = (short)o.k;
            rc.stride[k] = (short)(o.stride > 0 ? o.stride : 1);
            rc.prod[k] = (short)prod[k];
        }
        hipLaunchKernelGGL(seq_lengths_kernel, dim3((B + 255) / 256), dim3(256), 0, st, d_len, B, L, rc, table);
        RS_HIP(hipGetLastError());
    }
    auto rows_after = [&](int op) -> const int32_t* { return table ? table + (size_t)(op + 1) * B : nullptr; };   // op = -1: the input
    int last = 0, last_op = -1;
    for (size_t k = 0; k < m->ops.size(); ++k) {
        const OpDev& o = m->ops[k];
        const OpShape& sh = shp[k];
        if (o.fuse == 1 && (int64_t)B * 2 * shp[k + 1].t_out < m->window && (int64_t)B * sh.t_in * 4 < m->window) {
            // stem conv + ReLU + MaxPool(2, 2, pad 1) in one launch: GEMM rows = 2 * pooled rows
            const OpShape& ps = shp[k + 1];
            const int rows = B * 2 * ps.t_out;
            const int n_tiles = (rows + 127) / 128;
            if (m->mode == 1 && o.d_x_w1) {                        // the stem in split precision on the bf16 MFMA
                const size_t lds = (size_t)o.x_s1 * 4 * o.nt * 16 * 8 * 2 * 2 + (size_t)4 * 16 * o.c_out * 4;
                auto fx = o.nt == 1 ? seq_stem_pool_x3_kernel<1> : o.nt == 2 ? seq_stem_pool_x3_kernel<2> : o.nt == 3 ? seq_stem_pool_x3_kernel<3>
                        : o.nt == 4 ? seq_stem_pool_x3_kernel<4> : seq_stem_pool_x3_kernel<5>;
                RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fx), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
                const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1)));
                const int grid = std::min(n_tiles, m->num_cu * per_cu);
                hipLaunchKernelGGL(fx, dim3(grid), dim3(256), lds, st, buf(o.f_src), (unsigned)((int64_t)B * sh.t_in * 4), o.d_x_w1, o.d_b,
                                   buf(o.f_dst), B, sh.t_in, sh.t_out, ps.t_out, o.c_out, o.k * o.c_in, o.x_s1, o.stride, o.pad, n_tiles,
                                   rows_after(prod[k]), rows_after((int)k));
                RS_HIP(hipGetLastError());
                last = o.f_dst;
                last_op = (int)(k + o.fuse_skip);
                k += o.fuse_skip;
                continue;
            }
            const int K = o.k * o.c_in, K16 = (K + 15) & ~15;
            const size_t lds = (size_t)K16 * o.nt * 16 * 4;
            auto fn = o.nt == 1 ? seq_stem_pool_kernel<1> : o.nt == 2 ? seq_stem_pool_kernel<2> : o.nt == 3 ? seq_stem_pool_kernel<3>
                    : o.nt == 4 ? seq_stem_pool_kernel<4> : seq_stem_pool_kernel<5>;
            RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1)));
            const int grid = std::min(n_tiles, m->num_cu * per_cu);
            hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, st, buf(o.f_src), (unsigned)((int64_t)B * sh.t_in * 4), o.d_wq, o.d_b,
                               buf(o.f_dst), B, sh.t_in, sh.t_out, ps.t_out, o.c_out, K, o.stride, o.pad, n_tiles, rows_after(prod[k]),
                               rows_after((int)k));
            RS_HIP(hipGetLastError());
            last = o.f_dst;
            last_op = (int)(k + o.fuse_skip);
            k += o.fuse_skip;
            continue;
        }
        if (o.fuse == 2 && (int64_t)B * shp[k + o.fuse_skip].t_in * o.f_cout * 4 < m->window) {
            // residual basic block in one launch (shapes of its first 3x3 conv: ops[k + fuse_skip - 1])
            const OpShape& s1 = shp[k + o.fuse_skip - 1];
            const int64_t xb = (int64_t)B * s1.t_in * o.f_cin * 4;
            if (xb < m->window && m->mode == 1 && o.d_x_w1) {
                // the block in split precision on the bf16 MFMA: same tiling rules, its own LDS footprint
                BlockX3Args a;
                a.x = buf(o.f_src);
                a.x_bytes = (unsigned)xb;
                a.y = buf(o.f_dst);
                a.w1 = o.d_x_w1;
                a.b1 = o.d_f_b1;
                a.w2 = o.d_x_w2;
                a.b2 = o.d_f_b2;
                a.NPs = o.x_np;
                a.B = B;
                a.T_in = s1.t_in;
                a.T_out = s1.t_out;
                a.c_in = o.f_cin;
                a.c_out = o.f_cout;
                a.Cp = o.x_cp;
                a.stride = o.f_stride;
                a.K1 = 3 * o.f_cin;
                a.Ksc = o.f_ksc;
                a.S1 = o.x_s1;
                a.S2a = o.x_s2a;
                a.Ssc = o.x_ssc;
                a.tin = rows_after(prod[k + o.fuse_skip - 1]);
                a.tout = rows_after((int)(k + o.fuse_skip - 1));
                const size_t lds_cap = 160 * 1024;
                auto lds_of = [&](int rows) { return o.x_wbytes + (size_t)(rows + 4) * o.x_cp * 4; };
                auto waste = [&](int rows) {
                    const int to = rows - 2, n = (s1.t_out + to - 1) / to;
                    return (double)(n * to - s1.t_out) / (double)(n * to);
                };
                int mtw = 1, waves = 4;
                if (lds_cap / lds_of(64) < 2) {
                    if (lds_of(128) <= lds_cap) waves = 8;
                } else {
                    if (lds_cap / lds_of(128) >= 2 && waste(128) < 0.06) mtw = 2;
                }
                const size_t f_lds = lds_of(16 * mtw * waves);
                const int TO = 16 * mtw * waves - 2;
                a.tiles_per_read = (s1.t_out + TO - 1) / TO;
                a.n_tiles = B * a.tiles_per_read;
                using Fn = void (*)(const BlockX3Args);
                static const Fn table[5][3] = {
                    {seq_basic_block_x3_kernel<1, 1, 4>, seq_basic_block_x3_kernel<1, 2, 4>, seq_basic_block_x3_kernel<1, 1, 8>},
                    {seq_basic_block_x3_kernel<2, 1, 4>, seq_basic_block_x3_kernel<2, 2, 4>, seq_basic_block_x3_kernel<2, 1, 8>},
                    {seq_basic_block_x3_kernel<3, 1, 4>, seq_basic_block_x3_kernel<3, 2, 4>, seq_basic_block_x3_kernel<3, 1, 8>},
                    {seq_basic_block_x3_kernel<4, 1, 4>, seq_basic_block_x3_kernel<4, 2, 4>, seq_basic_block_x3_kernel<4, 1, 8>},
                    {seq_basic_block_x3_kernel<5, 1, 4>, seq_basic_block_x3_kernel<5, 2, 4>, seq_basic_block_x3_kernel<5, 1, 8>}};
                Fn fn = table[o.f_nt - 1][waves == 8 ? 2 : mtw - 1];
                RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, lds_cap / f_lds));
                const int grid = std::min(a.n_tiles, m->num_cu * per_cu);
                hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * waves), f_lds, st, a);
                RS_HIP(hipGetLastError());
                last = o.f_dst;
                last_op = (int)(k + o.fuse_skip);
                k += o.fuse_skip;
                continue;
            }
            if (xb < m->window) {
                BlockArgs a;
                a.x = buf(o.f_src);
                a.x_bytes = (unsigned)xb;
                a.y = buf(o.f_dst);
                a.w1q = o.d_f_w1;
                a.b1 = o.d_f_b1;
                a.w2q = o.d_f_w2;
                a.b2 = o.d_f_b2;
                a.B = B;
                a.T_in = s1.t_in;
                a.T_out = s1.t_out;
                a.c_in = o.f_cin;
                a.c_out = o.f_cout;
                a.Cp = o.f_cp;
                a.NPs = o.f_nps;
                a.stride = o.f_stride;
                a.K1 = 3 * o.f_cin;
                a.K2a = 3 * o.f_cp;
                a.Ksc = o.f_ksc;
                a.tin = rows_after(prod[k + o.fuse_skip - 1]);
                a.tout = rows_after((int)(k + o.fuse_skip - 1));
                // tile = (row tiles per wave, waves): rows R = 16 * mtw * waves of the intermediate, R - 2 outputs.  More rows
                // per wave = fewer halo rows, weight-fragment reads and set-up instructions per MFMA; candidates must fit the
                // LDS next to the weights, should leave room for a second workgroup on the CU, and must not waste more than
                // ~6 % of their rows behind the end of the read; a block so wide that one workgroup owns the CU's LDS runs
                // eight waves (two per SIMD: with one, every LDS / global round trip idles the matrix pipe)
                const size_t lds_cap = 160 * 1024;
                auto lds_of = [&](int rows) { return (o.f_wfloats + (size_t)(rows + 4) * o.f_cp) * 4; };
                auto waste = [&](int rows) {
                    const int to = rows - 2, n = (s1.t_out + to - 1) / to;
                    return (double)(n * to - s1.t_out) / (double)(n * to);
                };
                int mtw = 1, waves = 4;
                if (lds_cap / lds_of(64) < 2) {
                    if (lds_of(128) <= lds_cap) waves = 8;
                } else {
                    if (lds_cap / lds_of(128) >= 2 && waste(128) < 0.06) mtw = 2;
                }
                const size_t f_lds = lds_of(16 * mtw * waves);
                const int TO = 16 * mtw * waves - 2;
                a.tiles_per_read = (s1.t_out + TO - 1) / TO;
                a.n_tiles = B * a.tiles_per_read;
                using Fn = void (*)(const BlockArgs);
                static const Fn table[5][3] = {   // [NT - 1][4 waves x 1 | 4 waves x 2 | 8 waves x 1 row tiles per wave]
                    {seq_basic_block_kernel<1, 1, 4>, seq_basic_block_kernel<1, 2, 4>, seq_basic_block_kernel<1, 1, 8>},
                    {seq_basic_block_kernel<2, 1, 4>, seq_basic_block_kernel<2, 2, 4>, seq_basic_block_kernel<2, 1, 8>},
                    {seq_basic_block_kernel<3, 1, 4>, seq_basic_block_kernel<3, 2, 4>, seq_basic_block_kernel<3, 1, 8>},
                    {seq_basic_block_kernel<4, 1, 4>, seq_basic_block_kernel<4, 2, 4>, seq_basic_block_kernel<4, 1, 8>},
                    {seq_basic_block_kernel<5, 1, 4>, seq_basic_block_kernel<5, 2, 4>, seq_basic_block_kernel<5, 1, 8>}};
                // (256-row tiles - four row tiles per wave - were measured on the 20-channel stage: 168 against 165 us)
                Fn fn = table[o.f_nt - 1][waves == 8 ? 2 : mtw - 1];
                RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, lds_cap / f_lds));
                const int grid = std::min(a.n_tiles, m->num_cu * per_cu);
                hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * waves), f_lds, st, a);
                RS_HIP(hipGetLastError());
                last = o.f_dst;
                last_op = (int)(k + o.fuse_skip);
                k += o.fuse_skip;
                continue;
            }
        }
        if (o.fuse == 3) {
            // residual bottleneck block in one launch; shapes: conv 1 = ops[k + skip - 2], conv 2 (the strided one) = [k + skip - 1]
            const OpShape& s1 = shp[k + o.fuse_skip - 2];
            const OpShape& s2 = shp[k + o.fuse_skip - 1];
            const int64_t xb = (int64_t)B * s1.t_in * o.f_cin * 4;
            if (xb < m->window && (int64_t)B * s2.t_out * o.f_cout * 4 < m->window && m->mode == 1 && m->bneck_x3 && o.d_x_w1) {
                BneckX3Args a;
                a.x = buf(o.f_src);
                a.x_bytes = (unsigned)xb;
                a.y = buf(o.f_dst);
                a.w1 = o.d_x_w1; a.w2 = o.d_x_w2; a.w3 = o.d_x_w3;
                a.b1 = o.d_f_b1; a.b2 = o.d_f_b2; a.b3 = o.d_f_b3;
                a.NPm = o.f_npm;
                a.NPo = o.f_nps;
                a.B = B;
                a.T_in = s1.t_in;
                a.T_out = s2.t_out;
                a.c_in = o.f_cin;
                a.c_mid = o.f_cmid;
                a.c_out = o.f_cout;
                a.Cmp = o.x_cp;
                a.stride = o.f_stride;
                a.Ksc = o.f_ksc;
                a.S1 = o.x_s1; a.S2 = o.x_s2a; a.S3 = o.x_s3; a.Ssc = o.x_ssc;
                a.tin = rows_after(prod[k + o.fuse_skip - 2]);
                a.tout = rows_after((int)(k + o.fuse_skip - 1));
                a.R2 = (128 - 3) / o.f_stride + 1;
                a.tiles_per_read = (s2.t_out + a.R2 - 1) / a.R2;
                a.n_tiles = B * a.tiles_per_read;
                using FnX = void (*)(const BneckX3Args);
                static const FnX tablex[2][5] = {
                    {seq_bottleneck_block_x3_kernel<1, 1>, seq_bottleneck_block_x3_kernel<1, 2>, seq_bottleneck_block_x3_kernel<1, 3>,
                     seq_bottleneck_block_x3_kernel<1, 4>, seq_bottleneck_block_x3_kernel<1, 5>},
                    {seq_bottleneck_block_x3_kernel<2, 1>, seq_bottleneck_block_x3_kernel<2, 2>, seq_bottleneck_block_x3_kernel<2, 3>,
                     seq_bottleneck_block_x3_kernel<2, 4>, seq_bottleneck_block_x3_kernel<2, 5>}};
                FnX fx = tablex[o.f_ntm - 1][o.f_nt - 1];
                const size_t f_lds = o.x_wbytes + (size_t)2 * (128 + 4) * o.x_cp * 4;
                RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fx), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(160 * 1024) / f_lds));
                const int grid = std::min(a.n_tiles, m->num_cu * per_cu);
                hipLaunchKernelGGL(fx, dim3(grid), dim3(256), f_lds, st, a);
                RS_HIP(hipGetLastError());
                last = o.f_dst;
                last_op = (int)(k + o.fuse_skip);
                k += o.fuse_skip;
                continue;
            }
            if (xb < m->window && (int64_t)B * s2.t_out * o.f_cout * 4 < m->window) {
                BneckArgs a;
                a.x = buf(o.f_src);
                a.x_bytes = (unsigned)xb;
                a.y = buf(o.f_dst);
                a.w1q = o.d_f_w1; a.b1 = o.d_f_b1; a.w2q = o.d_f_w2; a.b2 = o.d_f_b2; a.w3q = o.d_f_w3; a.b3 = o.d_f_b3;
                a.NPm = o.f_npm;
                a.NPo = o.f_nps;
                a.B = B;
                a.T_in = s1.t_in;
                a.T_out = s2.t_out;
                a.c_in = o.f_cin;
                a.c_mid = o.f_cmid;
                a.c_out = o.f_cout;
                a.Cmp = o.f_cp;
                a.stride = o.f_stride;
                a.Ksc = o.f_ksc;
                a.tin = rows_after(prod[k + o.fuse_skip - 2]);
                a.tout = rows_after((int)(k + o.fuse_skip - 1));
                a.R2 = (128 - 3) / o.f_stride + 1;
                a.tiles_per_read = (s2.t_out + a.R2 - 1) / a.R2;
                a.n_tiles = B * a.tiles_per_read;
                using Fn = void (*)(const BneckArgs);
                static const Fn table[2][5] = {
                    {seq_bottleneck_block_kernel<1, 1>, seq_bottleneck_block_kernel<1, 2>, seq_bottleneck_block_kernel<1, 3>,
                     seq_bottleneck_block_kernel<1, 4>, seq_bottleneck_block_kernel<1, 5>},
                    {seq_bottleneck_block_kernel<2, 1>, seq_bottleneck_block_kernel<2, 2>, seq_bottleneck_block_kernel<2, 3>,
                     seq_bottleneck_block_kernel<2, 4>, seq_bottleneck_block_kernel<2, 5>}};
                Fn fn = table[o.f_ntm - 1][o.f_nt - 1];
                const size_t f_lds = (o.f_wfloats + (size_t)2 * (128 + 4) * o.f_cp) * 4;
                RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(160 * 1024) / f_lds));
                const int grid = std::min(a.n_tiles, m->num_cu * per_cu);
                hipLaunchKernelGGL(fn, dim3(grid), dim3(256), f_lds, st, a);
                RS_HIP(hipGetLastError());
                last = o.f_dst;
                last_op = (int)(k + o.fuse_skip);
                k += o.fuse_skip;
                continue;
            }
        }
        if (d_len) {
            // a ragged batch runs in fused launches only (they mask by every read's own rows; the unfused kernels below treat
            // every read as `ld` samples long): a fused launch that does not fit its 2 GiB buffer windows is an error, not a
            // silent fall-through to wrong probabilities
            set_error("rs_seqnet_forward_ragged: op %zu of the program cannot run as a fused launch on %d reads of pitch %d (a buffer "
                      "beyond the 2 GiB window): split the batch", k, B, L);
            return RS_ERR_ARG;
        }
        if (o.kind == 0 && m->scalar_conv) {
            const int cq = (o.c_out + 3) / 4;
            const int64_t total = (int64_t)B * sh.t_out * cq;
            hipLaunchKernelGGL(seq_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, buf(o.src),
                               o.d_w, o.d_b, o.add >= 0 ? buf(o.add) : nullptr, buf(o.dst), B, sh.t_in, sh.t_out,
                               o.c_in, o.c_out, cq, o.k, o.stride, o.pad, o.relu);
        } else if (o.kind == 0 && o.d_wq && (int64_t)B * sh.t_in * o.c_in * 4 < m->window) {
            const int64_t rows = (int64_t)B * sh.t_out;
            const int n_tiles = (int)((rows + 127) / 128);
            const int K = o.k * o.c_in, K16 = (K + 15) & ~15;
            const size_t lds = (size_t)K16 * o.nt * 16 * 4;
            auto fn = o.nt == 1 ? seq_conv_mfma_lds_kernel<1> : o.nt == 2 ? seq_conv_mfma_lds_kernel<2>
                    : o.nt == 3 ? seq_conv_mfma_lds_kernel<3> : o.nt == 4 ? seq_conv_mfma_lds_kernel<4> : seq_conv_mfma_lds_kernel<5>;
            RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            // small workgroups (256 threads, <= 96 KB LDS, few registers): as many per CU as the weights in LDS allow
            const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1)));
            const int grid = std::min(n_tiles, m->num_cu * per_cu);
            hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, st, buf(o.src), (unsigned)((int64_t)B * sh.t_in * o.c_in * 4),
                               o.d_wq, o.d_b, o.add >= 0 ? buf(o.add) : nullptr, buf(o.dst), B, sh.t_in, sh.t_out, o.c_in,
                               o.c_out, K, o.stride, o.pad, o.relu, n_tiles);
        } else if (o.kind == 0) {
            const int cq = (o.c_out + 3) / 4;
            const int64_t rows = (int64_t)B * sh.t_out;
            const int nt = o.c_out <= 16 ? 1 : o.c_out <= 32 ? 2 : 4;
            const dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((o.c_out + 16 * nt - 1) / (16 * nt)));
            auto fn = nt == 1 ? seq_conv_mfma_kernel<1> : nt == 2 ? seq_conv_mfma_kernel<2> : seq_conv_mfma_kernel<4>;
            hipLaunchKernelGGL(fn, grid, dim3(256), 0, st, buf(o.src), o.d_w, o.d_b, o.add >= 0 ? buf(o.add) : nullptr,
                               buf(o.dst), B, sh.t_in, sh.t_out, o.c_in, o.c_out, cq * 4, o.k * o.c_in, o.stride, o.pad,
                               o.relu);
        } else {
            const int64_t total = (int64_t)B * sh.t_out * sh.c;
            hipLaunchKernelGGL(seq_maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                               buf(o.src), buf(o.dst), B, sh.t_in, sh.t_out, sh.c, o.pad);
        }
        RS_HIP(hipGetLastError());
        last = o.dst;
        last_op = (int)k;
    }
    if (C[last] != m->c_last) {
        set_error("rs_seqnet_forward: last buffer has %d channels, classifier expects %d", C[last], m->c_last);
        return RS_ERR_ARG;
    }
    hipLaunchKernelGGL(seq_head_kernel, dim3(B), dim3(256), 0, st, buf(last), T[last], m->c_last, m->d_fcw, m->d_fcb,
                       d_probs, d_logits, rows_after(last_op));
    RS_HIP(hipGetLastError());
    return RS_OK;
}

}  // extern "C"
