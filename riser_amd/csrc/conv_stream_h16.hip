// K3s (16-bit, narrow layers): streaming conv block for layers whose whole weight set fits in a wave's
// registers (C_in <= 32 = one MFMA k-step, C_out <= 48): ConvNet layers 1 and 2 of the shipped net,
// layer 1 with ConvNet layer 0 folded in ("fused preprocess + conv", BASELINE config 5).
//
//   Conv1d(C_in -> C_out, k=3, 'same', bias) -> ReLU -> MaxPool1d(2,2)   (riser/nets/cnn.py:52-65)
//
// These layers are HBM-bound on the 16-bit MFMA (AI 5-40 flop/B against a ridge of ~310): the
// tiled kernel of conv_h16.hip spends its time on LDS staging, a workgroup barrier per 512-row tile
// and 2-byte stores.  Here every WAVE streams an independent run of 16-row sub-tiles:
//   * the 3 x NT weight fragments (A operand: rows = output channels) and the bias stay in registers;
//   * the wave PRODUCES one 16-row sub-tile of input rows per step - loaded from HBM (64-byte rows,
//     16 bytes per lane: one fully coalesced 1 KiB wave load) or, for layer 1, COMPUTED from the
//     normalised fp32 signal (layer 0: 8 channels per lane from 4 samples, fp32 FMAs, one rounding to
//     16 bit) - and parks it in a wave-private LDS ring of 256 rows; no other wave ever touches the
//     ring, so there is no barrier anywhere in the kernel;
//   * one step later it CONSUMES the previous sub-tile: the three taps are three ds_read_b128
//     fragments at ring rows r-1, r, r+1 (B operand: columns = positions), 3 x NT MFMAs
//     (v_mfma_f32_16x16x32_{f16,bf16}), and the epilogue in registers: MaxPool = max with the
//     neighbouring lane (DPP quad_perm), + bias, ReLU, length mask, pack to 16 bit; even lanes
//     store channels {0,1}, odd lanes {2,3} of their 4-channel group: 4-byte stores that tile
//     whole output rows.
// Ring rows are 64 bytes, XOR-swizzled at 16-byte granularity exactly as in conv_h16.hip
// (conflict-free ds_read_b128 for all three tap shifts).
//
// SPLIT PRECISION (X3, rs_dtype RS_BF16X3 / RS_F16X3; layout and arithmetic of conv_ring_h16.hip): activations are
// hi + lo pairs stored per 32-channel panel as [hi x 32 | lo x 32] (128-byte rows in HBM and in the ring, swizzle
// slot ^ (row & 7)), weights come from the ring packing [tap][n_alloc][hi x 32 | lo x 32], a product is three MFMAs
// (hi*hi, lo*hi, hi*lo) and the epilogue splits every output once more.  Twice the weight registers: two waves per
// SIMD instead of four.
#include "common.hpp"

#include <stdlib.h>

#include <algorithm>
#include <utility>

namespace rs {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kWaves = 4;                     // per workgroup; each wave is independent
constexpr int kRing = 64;                     // ring rows per wave (64 bytes each): 4 sub-tiles, 3 are live
constexpr unsigned kOob = 0x80000000u;

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

struct StreamArgs {
    const void* x;            // producer = load: 16-bit activations [rows_in][cp_in]; FUSE0: unused
    const float* xs;          // FUSE0: normalised signals, flat [B * P0], preceded by 16 zero bytes
    const float* w0;          // FUSE0: layer 0 (w0, w1, w2, bias) per channel, [cp0][4] fp32
    int c0;                   // FUSE0: layer-0 channels (<= 32)
    const unsigned short* w;  // packed [panel = 1][tap][n_alloc][32]; X3: ring packing [tap][n_alloc][64] = hi x 32 | lo x 32
    const float* bias;        // [n_alloc]
    void* y;                  // [rows_in / 2][cp_out] 16-bit
    const int32_t* len;
    unsigned x_bytes, xs_bytes, y_bytes, w_bytes, bias_bytes;
    int rows_in;              // B * P_in
    int P_in;
    int n_reads;
    int cp_in, cp_out;
    int n_alloc;
    int shift_in;             // valid input rows of read b: len[b] >> shift_in   (output: >> (shift_in + 1))
    int n_sub;                // number of 16-row sub-tiles = ceil(rows_in / 16)
    int sub_per_wave;
};

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c,
                                                      0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                       c, 0, 0, 0);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// two fp32 -> one dword of two 16-bit values, round to nearest even (v_cvt_pk_{f16,bf16}_f32)
template <bool F16>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    if constexpr (F16)
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    else
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

template <bool X3>
__device__ __forceinline__ int swz(int row) {
    return X3 ? (row & 7) : ((row >> 2) & 1) << 1;
}

template <bool F16>
__device__ __forceinline__ float widen16(unsigned short u) {
    if constexpr (F16)
        return (float)__builtin_bit_cast(_Float16, u);
    else
        return __builtin_bit_cast(float, (unsigned)u << 16);
}
// lo dword of a pair of values whose hi dword (two 16-bit roundings) is `hi`
template <bool F16>
__device__ __forceinline__ unsigned pack2_lo(float a, float b, unsigned hi) {
    return pack2<F16>(a - widen16<F16>((unsigned short)(hi & 0xffffu)), b - widen16<F16>((unsigned short)(hi >> 16)));
}

// max of x and the value of the lane that holds the other position of the pooling pair (lane ^ 1)
__device__ __forceinline__ float max_pair(float x) {
    const int o = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
    return fmaxf(x, __builtin_bit_cast(float, o));
}

template <bool FUSE0, int NT, bool F16, bool X3>
__global__ __launch_bounds__(kWaves * 64, X3 ? 2 : 4) void conv_stream_h16_kernel(const StreamArgs a) {
    constexpr int ROWB = X3 ? 128 : 64;                         // ring row / input row of one panel
    constexpr int NH = X3 ? 2 : 1;                              // 16-byte halves a lane handles per row: hi (and lo)
    __shared__ __attribute__((aligned(16))) unsigned char ring_all[kWaves * kRing * ROWB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, kq = lane >> 4;
    unsigned char* ring = ring_all + wave * (kRing * ROWB);

    const int gw = blockIdx.x * kWaves + wave;                 // global wave index
    const int u0 = gw * a.sub_per_wave;                         // first sub-tile of this wave's run
    const int u1 = min(u0 + a.sub_per_wave, a.n_sub);
    if (u0 >= u1) return;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        FUSE0 ? (void*)(a.xs - 4) : const_cast<void*>(a.x), 0, FUSE0 ? a.xs_bytes + 16u : a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);

    // ---- resident operands ---------------------------------------------------------------------------
    u32x4 wf[NH][3][NT];                                        // A fragments: W[n = 16j + r][tap][8kq .. 8kq+7], hi (and lo)
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                wf[h][t][j] = *reinterpret_cast<const u32x4*>(a.w + ((size_t)t * a.n_alloc + 16 * j + r) * (32 * NH) + 32 * h + 8 * kq);
    f32x4 bias[NT];                                             // channels 16j + 4kq + q of this lane's accumulators
#pragma unroll
    for (int j = 0; j < NT; ++j) bias[j] = *reinterpret_cast<const f32x4*>(a.bias + 16 * j + 4 * kq);
    f32x4 w0r[FUSE0 ? 8 : 1];                                   // layer 0: this lane's channels 8kq .. 8kq+7
    if constexpr (FUSE0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int ch = 8 * kq + q;
            w0r[q] = ch < a.c0 ? *reinterpret_cast<const f32x4*>(a.w0 + 4 * ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }

    // per-sub-tile row bookkeeping: read index / position of the sub-tile's first row by one scalar
    // division, lanes add their offset (a sub-tile is 16 rows < P_in, so it spans at most two reads).
    // Two wave-uniform fast paths: every row valid (no masking), every row beyond its read's length
    // (mixed-length batches: no layer-0 arithmetic, no MFMAs, zeros in / zeros out).
    const const_len_ptr clen = as_const_len(a.len);
    struct SubInfo {
        int t0, l0, l1, b;
    };
    // the (read, position, lengths) of the sub-tile being produced is carried in scalar registers and
    // advanced incrementally: a look-up per step (division + two dependent scalar loads, used at once
    // by a branch) put ~2 k cycles of exposed latency into every step
    auto sub_info = [&](int u) {
        const int g0 = 16 * u;
        const int b0 = g0 / a.P_in;
        SubInfo si;
        si.t0 = g0 - b0 * a.P_in;
        si.b = b0;
        si.l0 = b0 < a.n_reads ? clen[b0] >> a.shift_in : 0;
        si.l1 = b0 + 1 < a.n_reads ? clen[b0 + 1] >> a.shift_in : 0;
        return si;
    };
    auto advance = [&](SubInfo& si) {
        si.t0 += 16;
        if (si.t0 >= a.P_in) {
            si.t0 -= a.P_in;
            ++si.b;
            si.l0 = si.l1;
            si.l1 = si.b + 1 < a.n_reads ? clen[si.b + 1] >> a.shift_in : 0;
        }
    };
    auto lane_info = [&](const SubInfo& si, int& t, int& lim) {  // position of row 16u + r in its read; its valid rows
        const int tt = si.t0 + r;
        const bool hi = tt >= a.P_in;
        t = hi ? tt - a.P_in : tt;
        lim = hi ? si.l1 : si.l0;
    };

    // ---- producer: one sub-tile of input rows into the ring ------------------------------------------
    constexpr int D = 4;
    constexpr int NL = (X3 && !FUSE0) ? 2 : 1;                  // raw 16-byte loads per lane and sub-tile
    struct Raw {
        u32x4 v[NL];
    };
    Raw pre[D];                                                 // raw loads, D sub-tiles ahead of the producer
    auto issue_load = [&](int u, Raw& dst) {
        const int g = 16 * u + r;
        unsigned off;
        if constexpr (FUSE0)
            off = (u >= 0 && g < a.rows_in) ? (unsigned)(2 * g - 1 + 4) * 4u : kOob;        // x[2g-1 .. 2g+2]
        else if constexpr (X3)
            off = (u >= 0 && g < a.rows_in) ? ((unsigned)g * a.cp_in + 8 * kq) * 2u : kOob;  // hi piece; lo 64 bytes on
        else
            off = (u >= 0 && g < a.rows_in && 8 * kq < a.cp_in) ? ((unsigned)g * a.cp_in + 8 * kq) * 2u : kOob;
        dst.v[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
        if constexpr (NL == 2) dst.v[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off == kOob ? kOob : off + 64u, 0, 0);
    };
    struct Row {
        u32x4 v[NH];                                            // this lane's 8 channels of its row: hi (and lo)
    };
    auto conv0 = [&](const u32x4& raw, bool valid) {
        const f32x4 xv = __builtin_bit_cast(f32x4, raw);
        float o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {                           // same fmaf chains as conv0_kernel, rounded once
            const float e = fmaf(w0r[q][2], xv[2], fmaf(w0r[q][1], xv[1], fmaf(w0r[q][0], xv[0], w0r[q][3])));
            const float f = fmaf(w0r[q][2], xv[3], fmaf(w0r[q][1], xv[2], fmaf(w0r[q][0], xv[1], w0r[q][3])));
            o[q] = valid ? fmaxf(fmaxf(e, f), 0.0f) : 0.0f;
        }
        Row out;
        out.v[0] = (u32x4){pack2<F16>(o[0], o[1]), pack2<F16>(o[2], o[3]), pack2<F16>(o[4], o[5]), pack2<F16>(o[6], o[7])};
        if constexpr (X3)
            out.v[1] = (u32x4){pack2_lo<F16>(o[0], o[1], out.v[0][0]), pack2_lo<F16>(o[2], o[3], out.v[0][1]),
                               pack2_lo<F16>(o[4], o[5], out.v[0][2]), pack2_lo<F16>(o[6], o[7], out.v[0][3])};
        return out;
    };
    auto produce = [&](int u, const Raw& raw, const SubInfo& si) {
        Row v;
        if constexpr (FUSE0) {
            if (u >= 0 && si.t0 + 16 <= si.l0) {
                v = conv0(raw.v[0], true);
            } else if (u < 0 || (si.t0 >= si.l0 && si.t0 + 16 <= a.P_in)) {
#pragma unroll
                for (int h = 0; h < NH; ++h) v.v[h] = (u32x4){0u, 0u, 0u, 0u};
            } else {
                int t, lim;
                lane_info(si, t, lim);
                v = conv0(raw.v[0], t < lim);
            }
        } else {
#pragma unroll
            for (int h = 0; h < NH; ++h) v.v[h] = raw.v[h];
        }
        const int rr = (16 * u + r) & (kRing - 1);
#pragma unroll
        for (int h = 0; h < NH; ++h)
            *reinterpret_cast<u32x4*>(ring + rr * ROWB + (((4 * h + kq) ^ swz<X3>(rr)) << 4)) = v.v[h];
    };

    // ---- consumer: outputs of sub-tile u from ring rows 16u - 1 .. 16u + 16 ---------------------------
    auto consume = [&](int u, const SubInfo& si) {
        const unsigned rowoff = (unsigned)((16 * u + r) >> 1) * (unsigned)(a.cp_out * 2);
        const bool odd = r & 1;
        const bool dead = (si.t0 >> 1) >= (si.l0 >> 1) && si.t0 + 16 <= a.P_in;       // uniform: all outputs are zero
        // byte offset of logical channel ch inside an output row (X3: 32-channel panels of [hi x 32 | lo x 32]) and
        // the number of logical channel slots of a row
        auto ch_off = [&](int ch) { return (unsigned)(X3 ? ((ch >> 5) << 6) + (ch & 31) : ch) * 2u; };
        const int ch_lim = X3 ? a.cp_out / 2 : a.cp_out;
        // X3: a row holds 32 channel slots per panel but only NT * 16 channels are computed: the slots behind them are
        // written as zeros (the next layer multiplies them by zero weights, so they must be finite), and a lane stores
        // 8 bytes: the even lane of a pooling pair the hi halves of its 4 channels, the odd lane the lo halves
        auto zero_tail = [&]() {
            if constexpr (X3)
                for (int j = NT; 16 * j < ch_lim; ++j)
                    __builtin_amdgcn_raw_buffer_store_b64((u32x2){0u, 0u}, rs_y,
                                                          rowoff + ch_off(16 * j + 4 * kq) + (odd ? 64u : 0u), 0, 0);
        };
        if (dead) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if constexpr (X3) {
                    const int ch = 16 * j + 4 * kq;
                    __builtin_amdgcn_raw_buffer_store_b64((u32x2){0u, 0u}, rs_y,
                                                          ch < ch_lim ? rowoff + ch_off(ch) + (odd ? 64u : 0u) : kOob, 0, 0);
                } else {
                    const int ch = 16 * j + 4 * kq + (odd ? 2 : 0);
                    __builtin_amdgcn_raw_buffer_store_b32(0u, rs_y, ch < ch_lim ? rowoff + ch_off(ch) : kOob, 0, 0);
                }
            }
            zero_tail();
            return;
        }
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int rr = (16 * u + r + t - 1) & (kRing - 1);
            u32x4 xf[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h)
                xf[h] = *reinterpret_cast<const u32x4*>(ring + rr * ROWB + (((4 * h + kq) ^ swz<X3>(rr)) << 4));
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[j] = mfma16<F16>(wf[0][t][j], xf[0], acc[j]);               // hi * hi
                if constexpr (X3) {
                    acc[j] = mfma16<F16>(wf[1][t][j], xf[0], acc[j]);           // w lo * x hi
                    acc[j] = mfma16<F16>(wf[0][t][j], xf[1], acc[j]);           // w hi * x lo
                }
            }
        }
        int t, lim;
        lane_info(si, t, lim);
        const bool all_valid = si.t0 + 16 <= (si.l0 & ~1);      // uniform: no masking needed
        const bool valid = all_valid || (t >> 1) < (lim >> 1);  // pooled position < output length of the read
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float p[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float m = max_pair(acc[j][q]);
                p[q] = valid ? fmaxf(m + bias[j][q], 0.0f) : 0.0f;
            }
            if constexpr (X3) {
                const unsigned h0 = pack2<F16>(p[0], p[1]), h1 = pack2<F16>(p[2], p[3]);
                const u32x2 word = odd ? (u32x2){pack2_lo<F16>(p[0], p[1], h0), pack2_lo<F16>(p[2], p[3], h1)} : (u32x2){h0, h1};
                const int ch = 16 * j + 4 * kq;
                const unsigned off = ch < ch_lim ? rowoff + ch_off(ch) + (odd ? 64u : 0u) : kOob;
                __builtin_amdgcn_raw_buffer_store_b64(word, rs_y, off, 0, 0);
            } else {
                const unsigned word = odd ? pack2<F16>(p[2], p[3]) : pack2<F16>(p[0], p[1]);
                const int ch = 16 * j + 4 * kq + (odd ? 2 : 0);
                const unsigned off = ch < ch_lim ? rowoff + ch_off(ch) : kOob;           // rows past the end: out of range
                __builtin_amdgcn_raw_buffer_store_b32(word, rs_y, off, 0, 0);
            }
        }
        zero_tail();
    };

    // ---- run: step s produces sub-tile v = u0 - 1 + s (the first one only for its last row, the last one
    // only for its first row) and consumes v - 1; the raw loads run D sub-tiles ahead of the producer
    // (HBM latency is several thousand cycles, a step a few hundred) --------------------------------------
    static_for<D>([&](auto K) { issue_load(u0 - 1 + decltype(K)::value, pre[decltype(K)::value]); });
    const int S = u1 - u0 + 2;
    int cur_v = u0 > 0 ? u0 - 1 : 0;                          // sub-tile that `cur` describes
    SubInfo cur = sub_info(cur_v), prev = cur;
    for (int s0 = 0; s0 < S; s0 += D) {
        static_for<D>([&](auto K) {
            constexpr int k = decltype(K)::value;
            const int s = s0 + k;
            if (s < S) {
                const int v = u0 - 1 + s;
                if (cur_v < v) {
                    prev = cur;
                    advance(cur);
                    ++cur_v;
                }
                produce(v, pre[k], cur);
                issue_load(v + D, pre[k]);
                if (s >= 2) consume(v - 1, prev);
            }
        });
    }
}

using KernelFn = void (*)(const StreamArgs);

template <bool FUSE0, bool X3>
KernelFn pick(int nt, bool f16) {
    switch (nt) {
        case 1: return f16 ? conv_stream_h16_kernel<FUSE0, 1, true, X3> : conv_stream_h16_kernel<FUSE0, 1, false, X3>;
        case 2: return f16 ? conv_stream_h16_kernel<FUSE0, 2, true, X3> : conv_stream_h16_kernel<FUSE0, 2, false, X3>;
        default: return f16 ? conv_stream_h16_kernel<FUSE0, 3, true, X3> : conv_stream_h16_kernel<FUSE0, 3, false, X3>;
    }
}

}  // namespace

// a layer qualifies when its input is one MFMA k-step wide and its weights fit the register budget
bool conv_stream_h16_ok(const ConvLayerDev& L, int P_in) {
    return L.c_in <= 32 && L.plan.nch == 1 && L.c_out <= 48 && P_in >= 32 && !L.hooks->no_stream_h16;
}

int launch_conv_stream_h16(const ConvLayerDev& L, const void* d_x, void* d_y, const int32_t* d_len, int B, int P_in,
                           int layer_index, int num_cu, bool f16, hipStream_t st, const float* fuse_xs,
                           const float* fuse_w0, int fuse_c0, bool x3) {
    const int64_t rows64 = (int64_t)B * P_in;
    const int64_t xb = rows64 * L.cp_in * 2, yb = rows64 / 2 * L.cp_out * 2, sb = rows64 * 2 * 4;
    if (rows64 > 0x7fffffff || xb >= 0x80000000LL || yb >= 0x80000000LL || (fuse_xs && sb >= 0x80000000LL)) {
        set_error("conv_stream_h16: batch too large for the 2 GiB buffer window, split it");
        return RS_ERR_ARG;
    }
    StreamArgs a;
    a.x = d_x;
    a.xs = fuse_xs;
    a.w0 = fuse_w0;
    a.c0 = fuse_c0;
    a.w = static_cast<const unsigned short*>(x3 ? L.d_w2 : L.d_w);
    if (!a.w) {
        set_error("conv_stream_h16: layer %d has no packed weights for this mode", layer_index);
        return RS_ERR_ARG;
    }
    a.bias = L.d_bias;
    a.y = d_y;
    a.len = d_len;
    a.x_bytes = (unsigned)xb;
    a.xs_bytes = (unsigned)sb;
    a.y_bytes = (unsigned)yb;
    a.w_bytes = 0;
    a.bias_bytes = 0;
    a.rows_in = (int)rows64;
    a.P_in = P_in;
    a.n_reads = B;
    a.cp_in = L.cp_in;
    a.cp_out = L.cp_out;
    a.n_alloc = L.plan.n_alloc;
    a.shift_in = layer_index;
    a.n_sub = (int)((rows64 + 15) / 16);
    // 4 workgroups (16 waves) per CU - 2 in split precision (register budget); runs of at least 32 sub-tiles so the
    // two warm-up sub-tiles stay cheap
    const int waves = num_cu * (x3 ? 2 : 4) * kWaves;
    a.sub_per_wave = std::max(32, (a.n_sub + waves - 1) / waves);
    const int n_waves = (a.n_sub + a.sub_per_wave - 1) / a.sub_per_wave;
    const int grid = (n_waves + kWaves - 1) / kWaves;
    const int nt = (round_up(L.c_out, 16)) / 16;
    KernelFn fn = x3 ? (fuse_xs ? pick<true, true>(nt, f16) : pick<false, true>(nt, f16))
                     : (fuse_xs ? pick<true, false>(nt, f16) : pick<false, false>(nt, f16));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kWaves * 64), 0, st, a);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

}  // namespace rs
