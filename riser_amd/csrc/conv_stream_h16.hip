// K3s (16-bit, narrow layers): streaming conv block for layers whose whole weight set fits in a wave's
// registers (C_in <= 32 = one MFMA k-step, C_out <= 48): ConvNet layers 1 and 2 of the shipped net,
// layer 1 with ConvNet layer 0 folded in ("fused preprocess + conv", BASELINE config 5).
//
//   Conv1d(C_in -> C_out, k=3, 'same', bias) -> ReLU -> MaxPool1d(2,2)   (riser/nets/cnn.py:52-65)
//
// These layers are HBM-bound on the 16-bit MFMA (AI 5-40 flop/B against a ridge of ~310): the tiled kernel spends its
// time on LDS staging, a workgroup barrier per tile and narrow stores.  Here every WAVE streams an independent run of
// 32-row blocks (16 pooled output rows each):
//   * the 3 x NT weight fragments (A operand: rows = output channels) and the bias stay in registers;
//   * the wave PRODUCES 16-row sub-tiles of input rows - loaded from HBM (one fully coalesced 1 KiB wave load) or, for
//     layer 1, COMPUTED from the normalised fp32 signal: layer 0 runs on the f32-input MFMA (v_mfma_f32_16x16x4_f32,
//     K = 3 taps; exact fmaf chains from the bias in the accumulator input, bit-identical to conv0_kernel), columns =
//     pooled rows, one MFMA for the even and one for the odd conv position, so MaxPool + ReLU is one v_max3_f32 -
//     and parks them in a wave-private LDS ring of 64 rows; no other wave ever touches the ring, so there is no
//     barrier anywhere in the kernel;
//   * it CONSUMES a block when the sub-tile behind it is parked: lane (c, kq) owns pooled row c, with one accumulator
//     set for its even conv position and one for the odd one, so four ds_read_b128 row fragments (ring rows
//     2c-1 .. 2c+2) serve 6 x NT MFMAs (v_mfma_f32_16x16x32_{f16,bf16}) and MaxPool is a max of two registers of
//     the same lane; + bias, ReLU, length mask, pack to 16 bit; v_permlane16_swap_b32 gathers 8 consecutive channels
//     per lane and the wave stores 64 contiguous bytes per output row and instruction.
// Ring rows are 64 (X3: 128) bytes at a pitch of 80 (144): with that pitch the consumer's reads of every second row
// and the producer's writes of consecutive rows are free of bank conflicts without a swizzle.
// Blocks never span reads (P_in % 32 == 0); interior blocks (everything valid) take a path without any masking or
// row bookkeeping, the blocks at a read's end the general one.  Both paths issue the same memory operations in the
// same order and there is no branch around a block, which keeps the compiler's vmcnt bookkeeping exact.
// Measured (MI355X, 512 x 16000, f16): layers 0+1 0.097 -> 0.060 ms, layer 2 0.058 -> 0.048 ms; split precision 0.165
// -> 0.125 and 0.123 -> 0.120 ms.  What is left in layers 0+1 is issue time: the f32-input MFMA shares the FP32 data
// path with the VALU (the two never overlap, SQ counters: MFMA busy + VALU active = kernel time), so layer 0 costs
// the same 256 cycles per block on either; v_pk_fma_f32 / v_pk_add_f32 issue at half rate and buy nothing.
//
// SPLIT PRECISION (X3, rs_dtype RS_BF16X3 / RS_F16X3; layout and arithmetic of conv_ring_h16.hip): activations are
// hi + lo pairs stored per 32-channel panel as [hi x 32 | lo x 32] (128-byte rows in HBM and in the ring), weights
// come from the ring packing [tap][n_alloc][hi x 32 | lo x 32], a product is three MFMAs (hi*hi, lo*hi, hi*lo) and
// the epilogue splits every output once more.  Twice the weight registers: two waves per SIMD instead of four.
#include "common.hpp"

#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <utility>

namespace rs {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifndef RS_STREAM_WGS
#define RS_STREAM_WGS 4
#endif
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
    float unscale;            // 2^-k of the packed weights' power-of-two scale (ConvLayerDev::w_unscale; 1 outside half precision)
    void* y;                  // [rows_in / 2][cp_out] 16-bit
    const int32_t* len;
    unsigned* sat;            // half precision: the model's overflow flag (common.hpp: f16_overflow_bits), else null
    unsigned x_bytes, xs_bytes, y_bytes;
    int rows_in;              // B * P_in
    int P_in;
    int n_reads;
    int cp_in, cp_out;
    int n_alloc;
    int shift_in;             // valid input rows of read b: len[b] >> shift_in   (output: >> (shift_in + 1))
    int n_sub;                // number of 32-row blocks (16 output rows each) = rows_in / 32
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

template <bool FUSE0, int NT, bool F16, bool X3>
__global__ __launch_bounds__(kWaves * 64, X3 ? 2 : NT == 3 ? 3 : RS_STREAM_WGS) void conv_stream_h16_kernel(const StreamArgs a) {
    unsigned sat = 0u;                                          // half precision: a conversion overflowed (raised at the end)
    constexpr int ROWB = X3 ? 128 : 64;                         // ring row / input row of one panel
    constexpr int NH = X3 ? 2 : 1;                              // 16-byte halves a lane handles per row: hi (and lo)
    // ring row pitch: 16 bytes more than a row, so that the consumer's ds_read_b128 of every SECOND row (rows 2c + d of
    // lane group c) and the producer's ds_write_b128 of consecutive rows are both free of bank conflicts without a
    // swizzle (slot = (5 row + kq) mod 16 resp. (9 row + kq) mod 16 over the 16-lane groups of ds_read_b128)
    constexpr int ROWP = ROWB + 16;
    __shared__ __attribute__((aligned(16))) unsigned char ring_all[kWaves * kRing * ROWP];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, kq = lane >> 4;
    unsigned char* ring = ring_all + wave * (kRing * ROWP);

    const int gw = blockIdx.x * kWaves + wave;                 // global wave index
    const int u0 = gw * a.sub_per_wave;                         // first block of this wave's run
    const int u1 = u0 + a.sub_per_wave;                         // a multiple of 4 blocks; blocks >= n_sub are dropped
    if (u0 >= a.n_sub) return;

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
    // layer 0 on the f32-input MFMA (v_mfma_f32_16x16x4_f32, K = 3 taps + one zero column; bit-exact fmaf chains from
    // the accumulator input, which carries the bias: the same chain as conv0_kernel): A = W0[channel 16m + r][tap kq],
    // C = bias of channels 16m + 4kq + i
    float w0t[2] = {0.f, 0.f};
    f32x4 cb[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if constexpr (FUSE0) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            w0t[m] = (kq < 3 && 16 * m + r < a.c0) ? a.w0[4 * (16 * m + r) + kq] : 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = 16 * m + 4 * kq + i;
                cb[m][i] = ch < a.c0 ? a.w0[4 * ch + 3] : 0.0f;
            }
        }
    }
    // ring slot (16 bytes = 8 channels) this lane writes: a loaded row arrives as channels 8kq .., a computed layer-0
    // row as channels 16 (kq & 1) + 8 (kq >> 1) .. (see conv0)
    const int wslot = FUSE0 ? 2 * (kq & 1) + (kq >> 1) : kq;

    // ---- row bookkeeping ------------------------------------------------------------------------------
    // P_in is a multiple of 32, so neither a 16-row sub-tile nor a 32-row block spans two reads.  The wave carries the
    // (read, first position, valid rows) of the block it consumes next in scalar registers and advances them by 32
    // rows per block; a sub-tile is described by the position of its first row in its read and the read's valid rows.
    const const_len_ptr clen = as_const_len(a.len);
    struct SubInfo {
        int t0, l0;                                             // first row's position in its read; valid rows of the read
    };
    auto len_of = [&](int b) { return (b >= 0 && b < a.n_reads) ? clen[b] >> a.shift_in : 0; };

    // ---- producer: one 16-row sub-tile of input rows into the ring -----------------------------------
    // prefetch depth in sub-tiles.  vmcnt counts loads and stores in one queue, in order: with 4 sub-tiles in flight
    // the wait for a prefetched load also waited for the store of the block before
    constexpr int D = (X3 && !FUSE0) ? 4 : 8;
    constexpr int NL = (X3 && !FUSE0) ? 2 : 1;                  // raw 16-byte loads per lane and sub-tile
    struct Raw {
        u32x4 v[NL];                                            // FUSE0: two floats in v[0][0 .. 1]
    };
    Raw pre[D];                                                 // raw loads, D sub-tiles ahead of the producer
    // Loads are bounds-checked by the buffer resource alone: a sub-tile before the first row gives offsets that wrap
    // to >= 2^31, one behind the last row offsets >= num_records, and both read as zeros.  Lanes of channel groups the
    // layer does not have carry an offset of 2^31 from the start.
    //   FUSE0: layer-0 B operands of this lane, x[2g - 1 + kq] (even conv position of pooled row g, tap kq) and
    //          x[2g + kq] (odd position); tap 3 meets a zero weight
    const unsigned lane_off = FUSE0 ? (unsigned)(2 * r - 1 + kq + 4) * 4u
                                    : ((X3 || 8 * kq < a.cp_in) ? (unsigned)(r * a.cp_in + 8 * kq) * 2u : kOob);
    const unsigned sub_bytes = FUSE0 ? 128u : 32u * (unsigned)a.cp_in;   // bytes from one sub-tile to the next
    auto issue_load = [&](int u, Raw& dst) {
        const unsigned off = lane_off + (unsigned)u * sub_bytes;
        if constexpr (FUSE0) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_x, off, 0, 0);
            dst.v[0][0] = v[0];
            dst.v[0][1] = v[1];
        } else {
            dst.v[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
            if constexpr (NL == 2) dst.v[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off + 64u, 0, 0);
        }
    };
    struct Row {
        u32x4 v[NH];                                            // this lane's 8 channels of its row: hi (and lo)
    };
    // Layer 0 of 16 pooled rows on the matrix pipe: columns = pooled rows, one MFMA for their even conv positions and
    // one for the odd ones per 16-channel tile, so MaxPool + ReLU is one v_max3_f32 per value, in-lane.  The lane
    // then holds channels 4kq .. +3 and 16 + 4kq .. +3 of its row; v_permlane16_swap_b32 turns that into 8
    // consecutive channels (16 (kq & 1) + 8 (kq >> 1) ..), which is a 16-byte slot of the ring row in natural order.
    auto conv0 = [&](const u32x4& raw) {
        const unsigned ue = raw[0], uo = raw[1];               // (bit_cast of a vector ELEMENT lvalue reads element 0)
        const float xe = __builtin_bit_cast(float, ue), xo = __builtin_bit_cast(float, uo);
        float o[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#ifdef RS_ABL_NOCONV0
            const f32x4 e = cb[m] * xe, f = cb[m] * xo;
#else
            const f32x4 e = __builtin_amdgcn_mfma_f32_16x16x4f32(w0t[m], xe, cb[m], 0, 0, 0);
            const f32x4 f = __builtin_amdgcn_mfma_f32_16x16x4f32(w0t[m], xo, cb[m], 0, 0, 0);
#endif
#pragma unroll
            for (int i = 0; i < 4; ++i) o[m][i] = fmaxf(fmaxf(e[i], f[i]), 0.0f);
        }
        unsigned w[NH][2][2];                                   // [hi / lo][tile][dword]
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            w[0][m][0] = pack2<F16>(o[m][0], o[m][1]);
            w[0][m][1] = pack2<F16>(o[m][2], o[m][3]);
            if constexpr (F16) sat |= f16_overflow_bits(w[0][m][0]) | f16_overflow_bits(w[0][m][1]);
            if constexpr (X3) {
                w[1][m][0] = pack2_lo<F16>(o[m][0], o[m][1], w[0][m][0]);
                w[1][m][1] = pack2_lo<F16>(o[m][2], o[m][3], w[0][m][1]);
            }
        }
        Row out;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const auto s0 = __builtin_amdgcn_permlane16_swap(w[h][0][0], w[h][1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(w[h][0][1], w[h][1][1], false, false);
            out.v[h] = (u32x4){s0[0], s1[0], s0[1], s1[1]};
        }
        return out;
    };
    const unsigned ring_w = (unsigned)(r * ROWP + (wslot << 4));   // this lane's byte in ring row 0 of a sub-tile
    auto park = [&](int u, const Row& v) {                     // sub-tile u occupies ring rows 16 (u & 3) ..
        unsigned char* dst = ring + ring_w + (unsigned)((u & 3) * 16 * ROWP);
#pragma unroll
        for (int h = 0; h < NH; ++h) *reinterpret_cast<u32x4*>(dst + 64 * h) = v.v[h];
    };
    // INTERIOR: every row of the sub-tile is a valid row of its read
    auto produce_fast = [&](int u, const Raw& raw) {
        Row v;
        if constexpr (FUSE0) {
            v = conv0(raw.v[0]);
        } else {
#pragma unroll
            for (int h = 0; h < NH; ++h) v.v[h] = raw.v[h];
        }
        park(u, v);
    };
    // GENERAL: rows beyond the read's length are zero rows (loaded rows already are; computed ones are masked); a
    // sub-tile wholly beyond it costs no arithmetic
    auto produce = [&](int u, const Raw& raw, const SubInfo& si) {
        Row v;
        if constexpr (FUSE0) {
            if (si.t0 >= si.l0) {
#pragma unroll
                for (int h = 0; h < NH; ++h) v.v[h] = (u32x4){0u, 0u, 0u, 0u};
            } else {
                v = conv0(raw.v[0]);
                const unsigned keep = si.t0 + r < si.l0 ? ~0u : 0u;
#pragma unroll
                for (int h = 0; h < NH; ++h) v.v[h] &= (u32x4){keep, keep, keep, keep};
            }
        } else {
#pragma unroll
            for (int h = 0; h < NH; ++h) v.v[h] = raw.v[h];
        }
        park(u, v);
    };

    // ---- consumer: the 16 pooled outputs of block v (input rows 32v .. 32v+31) from ring rows 32v - 1 .. 32v + 32.
    // Lane (c = lane & 15, kq) owns pooled position c of the block: one accumulator set for its even conv position
    // (input rows 2c-1, 2c, 2c+1), one for the odd one (2c, 2c+1, 2c+2), so MaxPool is a max of two registers of the
    // same lane, the four row fragments serve six operands, and every lane stores whole channel groups.
    // byte offset of logical channel ch inside an output row (X3: 32-channel panels of [hi x 32 | lo x 32]) and the
    // number of logical channel slots of a row
    auto ch_off = [&](int ch) { return (unsigned)(X3 ? ((ch >> 5) << 6) + (ch & 31) : ch) * 2u; };
    const int ch_lim = X3 ? a.cp_out / 2 : a.cp_out;
    // Stores.  A lane holds channels 16j + 4kq .. +3 of its row for every channel tile j (8 bytes of 16-bit values).
    // Two tiles j, j+1 are exchanged between the lane rows with v_permlane16_swap_b32 so that every lane ends up with
    // 8 CONSECUTIVE channels - 16j + 16 (kq & 1) + 8 (kq >> 1) .. +7 - and the wave writes 64 contiguous bytes per
    // output row and instruction (X3: the hi half-row, then the lo half-row 64 bytes on).  X3 rows hold 32 channel
    // slots per panel but only NT * 16 channels are computed: the slots behind them are written as zeros (the next
    // layer multiplies them by zero weights, so they must be finite) - they ride in the pair.
    constexpr int NTP = X3 ? 2 * ((NT + 1) / 2) : NT;           // channel tiles written per row
    unsigned st_off[(NTP + 1) / 2];                             // this lane's byte in output row 0 per store, or out of range
#pragma unroll
    for (int j = 0; j + 1 < NTP; j += 2) {
        const int ch = 16 * j + 16 * (kq & 1) + 8 * (kq >> 1);
        st_off[j / 2] = ch < ch_lim ? (unsigned)(r * a.cp_out * 2) + ch_off(ch) : kOob;
    }
    if constexpr (NTP & 1) {
        const int ch = 16 * (NTP - 1) + 4 * kq;
        st_off[NTP / 2] = ch < ch_lim ? (unsigned)(r * a.cp_out * 2) + ch_off(ch) : kOob;
    }
    const unsigned blk_bytes = 32u * (unsigned)a.cp_out;        // 16 output rows
    auto store_rows = [&](int v, const u32x2 (&hi)[NTP], const u32x2 (&lo)[NTP]) {
#ifdef RS_ABL_NOSTORE
        const unsigned base = kOob;
#else
        const unsigned base = v < a.n_sub ? (unsigned)v * blk_bytes : kOob;   // blocks behind the grid are dropped
#endif
#pragma unroll
        for (int j = 0; j + 1 < NTP; j += 2) {
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const u32x2 wa = h ? lo[j] : hi[j], wb = h ? lo[j + 1] : hi[j + 1];
                const auto s0 = __builtin_amdgcn_permlane16_swap(wa[0], wb[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(wa[1], wb[1], false, false);
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){s0[0], s1[0], s0[1], s1[1]}, rs_y,
                                                       ((base | st_off[j / 2]) & kOob) ? kOob : base + st_off[j / 2] + 64u * h, 0, 0);
            }
        }
        if constexpr (NTP & 1)                                  // plain mode, odd tile count: the last tile alone
            __builtin_amdgcn_raw_buffer_store_b64(hi[NTP - 1], rs_y,
                                                  ((base | st_off[NTP / 2]) & kOob) ? kOob : base + st_off[NTP / 2], 0, 0);
    };
    const unsigned ring_r = (unsigned)(2 * r * ROWP + (kq << 4));   // this lane's byte in ring row 2c
    // MASKED = false: every output of the block is a valid output of its read
    auto consume = [&](int v, const SubInfo& si, auto MASKED) {
        constexpr bool masked = decltype(MASKED)::value;
        const int tp0 = si.t0 >> 1, out_len = si.l0 >> 1;
        u32x2 hi[NTP], lo[NTP];
#pragma unroll
        for (int j = 0; j < NTP; ++j) hi[j] = lo[j] = (u32x2){0u, 0u};
        if (masked && tp0 >= out_len) {                         // uniform: every output of the block is zero
            store_rows(v, hi, lo);
            return;
        }
        u32x4 xf[4][NH];                                        // ring rows 32v + 2c + d - 1, d = 0 .. 3
        // block v = sub-tiles 2v, 2v+1 = ring rows 32 (v & 1) .. +31; row -1 and row 32 wrap inside the 64-row ring
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const unsigned row0 = (unsigned)((32 * v + d - 1) & (kRing - 1));       // ring row of lane group c = 0
            // rows 2c + row0 stay below 64 except for d = 3 of the upper half's last lane group (row 64 -> 0)
            unsigned off = row0 * ROWP + ring_r;
            if (d == 3 || d == 0) off = (unsigned)(((32 * v + 2 * r + d - 1) & (kRing - 1)) * ROWP + (kq << 4));
#pragma unroll
            for (int h = 0; h < NH; ++h) xf[d][h] = *reinterpret_cast<const u32x4*>(ring + off + 64 * h);
        }
        f32x4 acc[2][NT];                                       // [even / odd conv position][channel tile]
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[e][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifndef RS_ABL_NOMFMA
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    acc[e][j] = mfma16<F16>(wf[0][t][j], xf[t + e][0], acc[e][j]);               // hi * hi
                    if constexpr (X3) {
                        acc[e][j] = mfma16<F16>(wf[0][t][j], xf[t + e][1], acc[e][j]);           // w hi * x lo
                        acc[e][j] = mfma16<F16>(wf[1][t][j], xf[t + e][0], acc[e][j]);           // w lo * x hi  (order of conv_ring_h16)
                    }
                }
#else
        for (int j = 0; j < NT; ++j) acc[0][j] = acc[1][j] = __builtin_bit_cast(f32x4, xf[0][0] ^ xf[1][0] ^ xf[2][0] ^ xf[3][0]);
#endif
        const unsigned keep = (!masked || tp0 + r < out_len) ? ~0u : 0u;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            // MaxPool, + bias, ReLU: max(e, o) + b == max(e + b, o + b) bit for bit (rounding is monotonic), which is
            // two v_pk_add_f32 and one v_max3_f32 per channel pair instead of max / add / max per channel
            const f32x2 b01 = {bias[j][0], bias[j][1]}, b23 = {bias[j][2], bias[j][3]};
            // x * unscale + b in one rounding (v_pk_fma_f32); unscale = 1 outside half precision: the plain sum, bit for bit
            const f32x2 us = {a.unscale, a.unscale};
            const f32x2 e01 = __builtin_elementwise_fma((f32x2){acc[0][j][0], acc[0][j][1]}, us, b01),
                        e23 = __builtin_elementwise_fma((f32x2){acc[0][j][2], acc[0][j][3]}, us, b23);
            const f32x2 o01 = __builtin_elementwise_fma((f32x2){acc[1][j][0], acc[1][j][1]}, us, b01),
                        o23 = __builtin_elementwise_fma((f32x2){acc[1][j][2], acc[1][j][3]}, us, b23);
            const float p0 = fmaxf(fmaxf(e01[0], o01[0]), 0.0f), p1 = fmaxf(fmaxf(e01[1], o01[1]), 0.0f);
            const float p2 = fmaxf(fmaxf(e23[0], o23[0]), 0.0f), p3 = fmaxf(fmaxf(e23[1], o23[1]), 0.0f);
            hi[j] = (u32x2){pack2<F16>(p0, p1), pack2<F16>(p2, p3)};
            if constexpr (F16) sat |= f16_overflow_bits(hi[j][0]) | f16_overflow_bits(hi[j][1]);
            if constexpr (X3) lo[j] = (u32x2){pack2_lo<F16>(p0, p1, hi[j][0]), pack2_lo<F16>(p2, p3, hi[j][1])};
            if constexpr (masked) {
                hi[j] &= (u32x2){keep, keep};
                lo[j] &= (u32x2){keep, keep};
            }
        }
        store_rows(v, hi, lo);
    };

    // ---- run: the wave owns blocks u0 .. u1-1, i.e. 16-row sub-tiles 2 u0 .. 2 u1 - 1, and also produces the sub-tile
    // before (for its last row) and the one after (for its first row).  Iteration v produces sub-tiles 2v+1 and 2v+2
    // and consumes block v: exactly the four sub-tiles 2v-1 .. 2v+2 are live then, which is the whole ring.  The raw
    // loads run D sub-tiles ahead of the producer (HBM latency is several thousand cycles, a step a few hundred).
    // There is no branch around a block and both paths of a block issue the same memory operations in the same
    // order, so the compiler's vmcnt bookkeeping stays exact; runs are multiples of D / 2 blocks (host), the blocks
    // behind the last one of the grid are dropped by the store.
    const int w_first = 2 * u0 - 1;
    static_for<D>([&](auto K) { issue_load(w_first + decltype(K)::value, pre[decltype(K)::value]); });
    int rb = (32 * u0) / a.P_in;                              // read of block u0, its first position in it, valid rows
    int t0 = 32 * u0 - rb * a.P_in;
    int l0 = len_of(rb), l1 = len_of(rb + 1);
    {
        // sub-tile 2 u0 - 1: the 16 rows before the block (of the read before when the block starts a read)
        const SubInfo sp = t0 > 0 ? SubInfo{t0 - 16, l0} : SubInfo{a.P_in - 16, len_of(rb - 1)};
        produce(w_first, pre[0], sp);
        issue_load(w_first + D, pre[0]);
        produce(w_first + 1, pre[1], SubInfo{t0, l0});
        issue_load(w_first + 1 + D, pre[1]);
    }
    for (int v0 = u0; v0 < u1; v0 += D / 2) {
        static_for<D / 2>([&](auto K) {
            constexpr int k = decltype(K)::value;
            const int v = v0 + k;
            Raw& sa = pre[(2 + 2 * k) % D];
            Raw& sb = pre[(3 + 2 * k) % D];
            if (t0 + 48 <= l0) {                                // interior: the block and the sub-tile behind it are valid rows
                produce_fast(2 * v + 1, sa);
                issue_load(2 * v + 1 + D, sa);
                produce_fast(2 * v + 2, sb);
                issue_load(2 * v + 2 + D, sb);
                consume(v, SubInfo{t0, l0}, std::false_type{});
            } else {
                const bool wrap = t0 + 32 >= a.P_in;
                produce(2 * v + 1, sa, SubInfo{t0 + 16, l0});
                issue_load(2 * v + 1 + D, sa);
                produce(2 * v + 2, sb, wrap ? SubInfo{0, l1} : SubInfo{t0 + 32, l0});
                issue_load(2 * v + 2 + D, sb);
                consume(v, SubInfo{t0, l0}, std::true_type{});
            }
            t0 += 32;
            if (t0 >= a.P_in) {
                t0 = 0;
                ++rb;
                l0 = l1;
                l1 = len_of(rb + 1);
            }
        });
    }
    if constexpr (F16) raise_saturated(a.sat, sat);
}

// ======================================================================================================
// Layers 0 + 1 + 2 in ONE streaming kernel: the activations of layers 0 and 1 never exist in memory.
//
// The wave-private pipeline of the kernel above with a second stage: a consumed block of layer 1 (16 pooled rows x
// 32 channels) is not stored but parked as a 16-row sub-tile of a SECOND ring, and every second block a block of
// layer 2 is consumed from that ring and stored.  Per layer-2 block w (32 layer-2 input rows -> 16 output rows):
//     layer-0 sub-tiles 4w+3, 4w+4 -> layer-1 block 2w+1 -> ring 2;  sub-tiles 4w+5, 4w+6 -> block 2w+2 -> ring 2;
//     layer-2 block w from ring-2 sub-tiles 2w-1 .. 2w+2 -> HBM.
// HBM traffic of the three layers: the normalised signal in, layer 2's output out (33 + 131 MB instead of 33 + 131 +
// 131 + 131 MB at 512 x 16000 in f16; twice the activation bytes in split precision).
// Layer 1 has two channel tiles (16 < C1 <= 32), layer 2 one to three; layer-2 weights: hi halves in registers, the lo
// halves of split precision in LDS (swizzled 64-byte rows).  Split precision runs one workgroup of 8 waves per CU (two
// 36 KB rings per 4 waves), plain 16-bit three workgroups of 4.  Same arithmetic, same order: bit-identical to
// the separate launches.
struct Stream2Args {
    const float* xs;          // normalised signals, flat [B * P0], preceded by 16 zero bytes
    const float* w0;          // layer 0 (w0, w1, w2, bias) per channel, [cp0][4] fp32
    int c0;
    const unsigned short* w1; // layer 1: [tap][n_alloc1][32]; X3: [tap][n_alloc1][64] = hi x 32 | lo x 32
    const float* bias1;
    float unscale1, unscale2; // as StreamArgs::unscale, layers 1 and 2
    int n_alloc1;
    const unsigned short* w2; // layer 2, same packing
    const float* bias2;
    int n_alloc2;
    void* y;                  // layer 2 output [B * P2 / 2][cp_out] 16-bit
    const int32_t* len;
    unsigned* sat;            // as StreamArgs::sat
    unsigned xs_bytes, y_bytes;
    int P2;                   // layer-2 input rows per read slot (P0 / 4), a multiple of 32
    int n_reads;
    int cp_out;
    int n_sub;                // layer-2 blocks = B * P2 / 32
    int sub_per_wave;         // even
};

template <int NT2, bool F16, bool X3>
__global__ __launch_bounds__((X3 ? 8 : 4) * 64, X3 ? 1 : 3) void conv_stream012_h16_kernel(const Stream2Args a) {
    unsigned sat = 0u;
    constexpr int NW = X3 ? 8 : 4;                              // waves per workgroup
    constexpr int NT1 = 2;
    constexpr int ROWB = X3 ? 128 : 64;
    constexpr int NH = X3 ? 2 : 1;
    constexpr int ROWP = ROWB + 16;
    constexpr int W2ROWS = 3 * 16 * NT2;
    __shared__ __attribute__((aligned(16))) unsigned char ring_all[NW * 2 * kRing * ROWP];
    __shared__ __attribute__((aligned(16))) unsigned char w2lo[X3 ? W2ROWS * 64 : 16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, kq = lane >> 4;
    unsigned char* ring1 = ring_all + wave * (2 * kRing * ROWP);
    unsigned char* ring2 = ring1 + kRing * ROWP;

    if constexpr (X3) {
        // lo halves of the layer-2 weights: row = tap * 16 NT2 + channel, 64 bytes, 16-byte slots swizzled by
        // 2 ((row >> 2) & 1) (conflict-free ds_read_b128 of 16 consecutive rows)
        for (int i = threadIdx.x; i < W2ROWS * 4; i += NW * 64) {
            const int row = i >> 2, slot = i & 3, t = row / (16 * NT2), n = row - t * (16 * NT2);
            const u32x4 v = *reinterpret_cast<const u32x4*>(a.w2 + ((size_t)t * a.n_alloc2 + n) * 64 + 32 + 8 * slot);
            *reinterpret_cast<u32x4*>(w2lo + row * 64 + ((slot ^ (((row >> 2) & 1) << 1)) << 4)) = v;
        }
        __syncthreads();
    }
    const int gw = blockIdx.x * NW + wave;
    const int U0 = gw * a.sub_per_wave, U1 = U0 + a.sub_per_wave;
    if (U0 >= a.n_sub) return;

    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.xs - 4), 0, a.xs_bytes + 16u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);

    // ---- resident operands ---------------------------------------------------------------------------
    u32x4 wf1[NH][3][NT1];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < NT1; ++j)
                wf1[h][t][j] = *reinterpret_cast<const u32x4*>(a.w1 + ((size_t)t * a.n_alloc1 + 16 * j + r) * (32 * NH) + 32 * h + 8 * kq);
    u32x4 wf2[3][NT2];                                          // hi halves (plain: the weights)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < NT2; ++j)
            wf2[t][j] = *reinterpret_cast<const u32x4*>(a.w2 + ((size_t)t * a.n_alloc2 + 16 * j + r) * (32 * NH) + 8 * kq);
    f32x4 bias1[NT1], bias2[NT2];
#pragma unroll
    for (int j = 0; j < NT1; ++j) bias1[j] = *reinterpret_cast<const f32x4*>(a.bias1 + 16 * j + 4 * kq);
#pragma unroll
    for (int j = 0; j < NT2; ++j) bias2[j] = *reinterpret_cast<const f32x4*>(a.bias2 + 16 * j + 4 * kq);
    float w0t[2];
    f32x4 cb[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        w0t[m] = (kq < 3 && 16 * m + r < a.c0) ? a.w0[4 * (16 * m + r) + kq] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = 16 * m + 4 * kq + i;
            cb[m][i] = ch < a.c0 ? a.w0[4 * ch + 3] : 0.0f;
        }
    }
    const int wslot = 2 * (kq & 1) + (kq >> 1);                 // ring slot of the 8 consecutive channels a lane gathers

    const const_len_ptr clen = as_const_len(a.len);
    struct SubInfo {
        int t0, l0;                                             // first input row's position in its read; valid input rows
    };
    auto len_of = [&](int b) { return (b >= 0 && b < a.n_reads) ? clen[b] : 0; };   // raw samples

    // ---- layer 0: sub-tiles of 16 layer-1 input rows into ring 1 ---------------------------------------
    constexpr int D = 8;
    u32x2 pre[D];
    const unsigned lane_off = (unsigned)(2 * r - 1 + kq + 4) * 4u;
    auto issue_load = [&](int u, u32x2& dst) { dst = __builtin_amdgcn_raw_buffer_load_b64(rs_x, lane_off + (unsigned)u * 128u, 0, 0); };
    struct Row {
        u32x4 v[NH];
    };
    auto conv0 = [&](const u32x2& raw) {
        const unsigned ue = raw[0], uo = raw[1];
        const float xe = __builtin_bit_cast(float, ue), xo = __builtin_bit_cast(float, uo);
        float o[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const f32x4 e = __builtin_amdgcn_mfma_f32_16x16x4f32(w0t[m], xe, cb[m], 0, 0, 0);
            const f32x4 f = __builtin_amdgcn_mfma_f32_16x16x4f32(w0t[m], xo, cb[m], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[m][i] = fmaxf(fmaxf(e[i], f[i]), 0.0f);
        }
        unsigned w[NH][2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            w[0][m][0] = pack2<F16>(o[m][0], o[m][1]);
            w[0][m][1] = pack2<F16>(o[m][2], o[m][3]);
            if constexpr (F16) sat |= f16_overflow_bits(w[0][m][0]) | f16_overflow_bits(w[0][m][1]);
            if constexpr (X3) {
                w[1][m][0] = pack2_lo<F16>(o[m][0], o[m][1], w[0][m][0]);
                w[1][m][1] = pack2_lo<F16>(o[m][2], o[m][3], w[0][m][1]);
            }
        }
        Row out;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const auto s0 = __builtin_amdgcn_permlane16_swap(w[h][0][0], w[h][1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(w[h][0][1], w[h][1][1], false, false);
            out.v[h] = (u32x4){s0[0], s1[0], s0[1], s1[1]};
        }
        return out;
    };
    const unsigned ring_w = (unsigned)(r * ROWP + (wslot << 4));
    auto park = [&](unsigned char* ring, int u, const Row& v) {   // sub-tile u occupies ring rows 16 (u & 3) ..
        unsigned char* dst = ring + ring_w + (unsigned)((u & 3) * 16 * ROWP);
#pragma unroll
        for (int h = 0; h < NH; ++h) *reinterpret_cast<u32x4*>(dst + 64 * h) = v.v[h];
    };
    auto produce_fast = [&](int u, const u32x2& raw) { park(ring1, u, conv0(raw)); };
    auto produce = [&](int u, const u32x2& raw, const SubInfo& si) {
        Row v;
        if (si.t0 >= si.l0) {
#pragma unroll
            for (int h = 0; h < NH; ++h) v.v[h] = (u32x4){0u, 0u, 0u, 0u};
        } else {
            v = conv0(raw);
            const unsigned keep = si.t0 + r < si.l0 ? ~0u : 0u;
#pragma unroll
            for (int h = 0; h < NH; ++h) v.v[h] &= (u32x4){keep, keep, keep, keep};
        }
        park(ring1, u, v);
    };

    // ---- a conv block: 16 pooled outputs of block v from the 34 ring rows 32v - 1 .. 32v + 32 ------------
    const unsigned ring_r = (unsigned)(2 * r * ROWP + (kq << 4));
    auto block = [&](const unsigned char* ring, int v, const SubInfo& si, auto MASKED, auto NTc, const auto& whi, const auto& wlo,
                     const auto& bias, const float unscale, auto&& sink) {
        constexpr bool masked = decltype(MASKED)::value;
        constexpr int NT = decltype(NTc)::value;
        constexpr int NTP = X3 ? 2 * ((NT + 1) / 2) : NT;
        const int tp0 = si.t0 >> 1, out_len = si.l0 >> 1;
        u32x2 hi[NTP], lo[NTP];
#pragma unroll
        for (int j = 0; j < NTP; ++j) hi[j] = lo[j] = (u32x2){0u, 0u};
        if (masked && tp0 >= out_len) {
            sink(hi, lo);
            return;
        }
        u32x4 xf[4][NH];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            unsigned off = (unsigned)((32 * v + d - 1) & (kRing - 1)) * ROWP + ring_r;
            if (d == 3 || d == 0) off = (unsigned)(((32 * v + 2 * r + d - 1) & (kRing - 1)) * ROWP + (kq << 4));
#pragma unroll
            for (int h = 0; h < NH; ++h) xf[d][h] = *reinterpret_cast<const u32x4*>(ring + off + 64 * h);
        }
        f32x4 acc[2][NT];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[e][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    acc[e][j] = mfma16<F16>(whi(t, j), xf[t + e][0], acc[e][j]);                 // hi * hi
                    if constexpr (X3) {
                        acc[e][j] = mfma16<F16>(whi(t, j), xf[t + e][1], acc[e][j]);             // w hi * x lo
                        acc[e][j] = mfma16<F16>(wlo(t, j), xf[t + e][0], acc[e][j]);             // w lo * x hi  (order of conv_ring_h16)
                    }
                }
        const unsigned keep = (!masked || tp0 + r < out_len) ? ~0u : 0u;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const f32x2 b01 = {bias[j][0], bias[j][1]}, b23 = {bias[j][2], bias[j][3]};
            const f32x2 us = {unscale, unscale};
            const f32x2 e01 = __builtin_elementwise_fma((f32x2){acc[0][j][0], acc[0][j][1]}, us, b01),
                        e23 = __builtin_elementwise_fma((f32x2){acc[0][j][2], acc[0][j][3]}, us, b23);
            const f32x2 o01 = __builtin_elementwise_fma((f32x2){acc[1][j][0], acc[1][j][1]}, us, b01),
                        o23 = __builtin_elementwise_fma((f32x2){acc[1][j][2], acc[1][j][3]}, us, b23);
            const float p0 = fmaxf(fmaxf(e01[0], o01[0]), 0.0f), p1 = fmaxf(fmaxf(e01[1], o01[1]), 0.0f);
            const float p2 = fmaxf(fmaxf(e23[0], o23[0]), 0.0f), p3 = fmaxf(fmaxf(e23[1], o23[1]), 0.0f);
            hi[j] = (u32x2){pack2<F16>(p0, p1), pack2<F16>(p2, p3)};
            if constexpr (F16) sat |= f16_overflow_bits(hi[j][0]) | f16_overflow_bits(hi[j][1]);
            if constexpr (X3) lo[j] = (u32x2){pack2_lo<F16>(p0, p1, hi[j][0]), pack2_lo<F16>(p2, p3, hi[j][1])};
            if constexpr (masked) {
                hi[j] &= (u32x2){keep, keep};
                lo[j] &= (u32x2){keep, keep};
            }
        }
        sink(hi, lo);
    };
    // layer 1: weights in registers, the block becomes sub-tile v of ring 2 (8 consecutive channels per lane)
    auto w1hi = [&](int t, int j) -> const u32x4& { return wf1[0][t][j]; };
    auto w1lo = [&](int t, int j) -> const u32x4& { return wf1[NH - 1][t][j]; };
    auto block1 = [&](int v, const SubInfo& si, auto MASKED) {
        block(ring1, v, si, MASKED, std::integral_constant<int, NT1>{}, w1hi, w1lo, bias1, a.unscale1, [&](const auto& hi, const auto& lo) {
            Row row;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const u32x2 wa = h ? lo[0] : hi[0], wb = h ? lo[1] : hi[1];
                const auto s0 = __builtin_amdgcn_permlane16_swap(wa[0], wb[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(wa[1], wb[1], false, false);
                row.v[h] = (u32x4){s0[0], s1[0], s0[1], s1[1]};
            }
            park(ring2, v, row);
        });
    };
    // layer 2: hi weights in registers, lo weights from LDS; the block is stored
    const int ch_lim = X3 ? a.cp_out / 2 : a.cp_out;
    auto ch_off = [&](int ch) { return (unsigned)(X3 ? ((ch >> 5) << 6) + (ch & 31) : ch) * 2u; };
    constexpr int NTP2 = X3 ? 2 * ((NT2 + 1) / 2) : NT2;
    unsigned st_off[(NTP2 + 1) / 2];
#pragma unroll
    for (int j = 0; j + 1 < NTP2; j += 2) {
        const int ch = 16 * j + 16 * (kq & 1) + 8 * (kq >> 1);
        st_off[j / 2] = ch < ch_lim ? (unsigned)(r * a.cp_out * 2) + ch_off(ch) : kOob;
    }
    if constexpr (NTP2 & 1) {
        const int ch = 16 * (NTP2 - 1) + 4 * kq;
        st_off[NTP2 / 2] = ch < ch_lim ? (unsigned)(r * a.cp_out * 2) + ch_off(ch) : kOob;
    }
    const unsigned blk_bytes = 32u * (unsigned)a.cp_out;
    auto w2hi = [&](int t, int j) -> const u32x4& { return wf2[t][j]; };
    auto block2 = [&](int v, const SubInfo& si, auto MASKED) {
        u32x4 wl[3][NT2];
        if constexpr (X3) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int j = 0; j < NT2; ++j) {
                    const int row = t * 16 * NT2 + 16 * j + r;
                    wl[t][j] = *reinterpret_cast<const u32x4*>(w2lo + row * 64 + ((kq ^ (((row >> 2) & 1) << 1)) << 4));
                }
        }
        auto w2l = [&](int t, int j) -> const u32x4& { return wl[t][j]; };
        block(ring2, v, si, MASKED, std::integral_constant<int, NT2>{}, w2hi, w2l, bias2, a.unscale2, [&](const auto& hi, const auto& lo) {
            const unsigned base = v < a.n_sub ? (unsigned)v * blk_bytes : kOob;
#pragma unroll
            for (int j = 0; j + 1 < NTP2; j += 2) {
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const u32x2 wa = h ? lo[j] : hi[j], wb = h ? lo[j + 1] : hi[j + 1];
                    const auto s0 = __builtin_amdgcn_permlane16_swap(wa[0], wb[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(wa[1], wb[1], false, false);
                    __builtin_amdgcn_raw_buffer_store_b128((u32x4){s0[0], s1[0], s0[1], s1[1]}, rs_y,
                                                           ((base | st_off[j / 2]) & kOob) ? kOob : base + st_off[j / 2] + 64u * h, 0, 0);
                }
            }
            if constexpr (NTP2 & 1)
                __builtin_amdgcn_raw_buffer_store_b64(hi[NTP2 - 1], rs_y,
                                                      ((base | st_off[NTP2 / 2]) & kOob) ? kOob : base + st_off[NTP2 / 2], 0, 0);
        });
    };

    // ---- run ---------------------------------------------------------------------------------------------
    // positions are tracked in layer-2 input rows (T0, a multiple of 32, in read rb); a layer-1 input row is twice
    // that, valid rows are len >> 1 (layer-1 input), len >> 2 (layer-2 input)
    const int s_first = 4 * U0 - 3;
    static_for<D>([&](auto K) { issue_load(s_first + decltype(K)::value, pre[decltype(K)::value]); });
    int rb = (32 * U0) / a.P2;
    int T0 = 32 * U0 - rb * a.P2;
    int len0 = len_of(rb), len1 = len_of(rb + 1);
    {
        // the six sub-tiles and two layer-1 blocks in front of the run's first layer-2 block
        const int lp = T0 > 0 ? len0 >> 1 : len_of(rb - 1) >> 1;
        const int bp = T0 > 0 ? 2 * T0 : 2 * a.P2;             // position of the row after them in THEIR read
        const int l0 = len0 >> 1;
        produce(s_first, pre[0], SubInfo{bp - 48, lp});
        issue_load(s_first + D, pre[0]);
        produce(s_first + 1, pre[1], SubInfo{bp - 32, lp});
        issue_load(s_first + 1 + D, pre[1]);
        produce(s_first + 2, pre[2], SubInfo{bp - 16, lp});
        issue_load(s_first + 2 + D, pre[2]);
        produce(s_first + 3, pre[3], SubInfo{2 * T0, l0});
        issue_load(s_first + 3 + D, pre[3]);
        block1(2 * U0 - 1, SubInfo{bp - 32, lp}, std::true_type{});
        produce(s_first + 4, pre[4], SubInfo{2 * T0 + 16, l0});
        issue_load(s_first + 4 + D, pre[4]);
        produce(s_first + 5, pre[5], SubInfo{2 * T0 + 32, l0});
        issue_load(s_first + 5 + D, pre[5]);
        block1(2 * U0, SubInfo{2 * T0, l0}, std::true_type{});
    }
    for (int w0 = U0; w0 < U1; w0 += 2) {
        static_for<2>([&](auto K) {
            constexpr int k = decltype(K)::value;
            const int w = w0 + k;
            u32x2& sa = pre[(6 + 4 * k) % D];
            u32x2& sb = pre[(7 + 4 * k) % D];
            u32x2& sc = pre[(8 + 4 * k) % D];
            u32x2& sd = pre[(9 + 4 * k) % D];
            const bool wrap = T0 + 32 >= a.P2;
            if (2 * T0 + 112 <= (len0 >> 1)) {                  // interior: everything this iteration touches is valid
                produce_fast(4 * w + 3, sa);
                issue_load(4 * w + 3 + D, sa);
                produce_fast(4 * w + 4, sb);
                issue_load(4 * w + 4 + D, sb);
                block1(2 * w + 1, SubInfo{0, 0}, std::false_type{});
                produce_fast(4 * w + 5, sc);
                issue_load(4 * w + 5 + D, sc);
                produce_fast(4 * w + 6, sd);
                issue_load(4 * w + 6 + D, sd);
                block1(2 * w + 2, SubInfo{0, 0}, std::false_type{});
                block2(w, SubInfo{0, 0}, std::false_type{});
            } else {
                const int tn = wrap ? 0 : 2 * (T0 + 32);       // layer-1 input position of the next layer-2 block
                const int ln = (wrap ? len1 : len0) >> 1, l0 = len0 >> 1;
                produce(4 * w + 3, sa, SubInfo{2 * T0 + 48, l0});
                issue_load(4 * w + 3 + D, sa);
                produce(4 * w + 4, sb, SubInfo{tn, ln});
                issue_load(4 * w + 4 + D, sb);
                block1(2 * w + 1, SubInfo{2 * T0 + 32, l0}, std::true_type{});
                produce(4 * w + 5, sc, SubInfo{tn + 16, ln});
                issue_load(4 * w + 5 + D, sc);
                produce(4 * w + 6, sd, SubInfo{tn + 32, ln});
                issue_load(4 * w + 6 + D, sd);
                block1(2 * w + 2, SubInfo{tn, ln}, std::true_type{});
                block2(w, SubInfo{T0, len0 >> 2}, std::true_type{});
            }
            T0 += 32;
            if (wrap) {
                T0 = 0;
                ++rb;
                len0 = len1;
                len1 = len_of(rb + 1);
            }
        });
    }
    if constexpr (F16) raise_saturated(a.sat, sat);
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
    return L.c_in <= 32 && L.plan.nch == 1 && L.c_out <= 48 && P_in >= 32 && P_in % 32 == 0 && !L.hooks->no_stream_h16;
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
    a.sat = f16 ? L.d_sat : nullptr;
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
    a.unscale = L.w_unscale;
    a.y = d_y;
    a.len = d_len;
    a.x_bytes = (unsigned)xb;
    a.xs_bytes = (unsigned)sb;
    a.y_bytes = (unsigned)yb;
    a.rows_in = (int)rows64;
    a.P_in = P_in;
    a.n_reads = B;
    a.cp_in = L.cp_in;
    a.cp_out = L.cp_out;
    a.n_alloc = L.plan.n_alloc;
    a.shift_in = layer_index;
    a.n_sub = (int)(rows64 / 32);                               // blocks of 32 input rows = 16 output rows
    // 4 workgroups (16 waves) per CU - 3 with three channel tiles, 2 in split precision (register budget); runs of at
    // least 16 blocks so the two warm-up sub-tiles stay cheap
    const int waves = num_cu * (x3 ? 2 : L.c_out > 32 ? 3 : RS_STREAM_WGS) * kWaves;
    a.sub_per_wave = round_up(std::max(16, (a.n_sub + waves - 1) / waves), 4);   // the kernel walks 4 blocks per iteration
    const int n_waves = (a.n_sub + a.sub_per_wave - 1) / a.sub_per_wave;
    const int grid = (n_waves + kWaves - 1) / kWaves;
    const int nt = (round_up(L.c_out, 16)) / 16;
    KernelFn fn = x3 ? (fuse_xs ? pick<true, true>(nt, f16) : pick<false, true>(nt, f16))
                     : (fuse_xs ? pick<true, false>(nt, f16) : pick<false, false>(nt, f16));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kWaves * 64), 0, st, a);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

// layers 1 and 2 qualify for the three-layer kernel when both would stream on their own, layer 1 has two channel tiles
// and the read slots of layer 2 are whole blocks
bool conv_stream012_h16_ok(const ConvLayerDev& L1, const ConvLayerDev& L2, int c0, int P_in1) {
    return c0 <= 32 && conv_stream_h16_ok(L1, P_in1) && conv_stream_h16_ok(L2, P_in1 / 2) && L1.c_out > 16 && L1.c_out <= 32 &&
           (P_in1 / 2) % 32 == 0 && !L1.hooks->no_stream012;
}

int launch_conv_stream012_h16(const ConvLayerDev& L1, const ConvLayerDev& L2, const float* d_xs, const float* d_w0, int c0,
                              void* d_y, const int32_t* d_len, int B, int P_in1, int num_cu, bool f16, bool x3, hipStream_t st) {
    const int P2 = P_in1 / 2;
    const int64_t rows2 = (int64_t)B * P2;
    const int64_t sb = rows2 * 4 * 4, yb = rows2 / 2 * L2.cp_out * 2;
    if (sb >= 0x80000000LL || yb >= 0x80000000LL) {
        set_error("conv_stream012_h16: batch too large for the 2 GiB buffer window, split it");
        return RS_ERR_ARG;
    }
    Stream2Args a;
    a.sat = f16 ? L1.d_sat : nullptr;
    a.xs = d_xs;
    a.w0 = d_w0;
    a.c0 = c0;
    a.w1 = static_cast<const unsigned short*>(x3 ? L1.d_w2 : L1.d_w);
    a.w2 = static_cast<const unsigned short*>(x3 ? L2.d_w2 : L2.d_w);
    if (!a.w1 || !a.w2) {
        set_error("conv_stream012_h16: no packed weights for this mode");
        return RS_ERR_ARG;
    }
    a.bias1 = L1.d_bias;
    a.unscale1 = L1.w_unscale;
    a.unscale2 = L2.w_unscale;
    a.bias2 = L2.d_bias;
    a.n_alloc1 = L1.plan.n_alloc;
    a.n_alloc2 = L2.plan.n_alloc;
    a.y = d_y;
    a.len = d_len;
    a.xs_bytes = (unsigned)sb;
    a.y_bytes = (unsigned)yb;
    a.P2 = P2;
    a.n_reads = B;
    a.cp_out = L2.cp_out;
    a.n_sub = (int)(rows2 / 32);
    const int nw = x3 ? 8 : 4;
    const int waves = num_cu * (x3 ? 8 : 12);                   // one workgroup of 8 waves, or three of 4, per CU
    // runs of eight blocks or more once every SIMD has a wave; a thin launch (a read or a few) spreads runs of two over more
    // waves (conv_stream_f32.hip; RS_SF32_MIN_RUN forces the floor)
    const int simds = num_cu * 4;
    const int floor_run = L1.hooks->sf32_min_run > 0 ? L1.hooks->sf32_min_run : std::min(8, std::max(2, (a.n_sub + simds - 1) / simds));
    a.sub_per_wave = round_up(std::max(floor_run, (a.n_sub + waves - 1) / waves), 2);
    const int n_waves = (a.n_sub + a.sub_per_wave - 1) / a.sub_per_wave;
    const int grid = (n_waves + nw - 1) / nw;
    const int nt2 = round_up(L2.c_out, 16) / 16;
    using Fn = void (*)(const Stream2Args);
    Fn fn;
    if (x3)
        fn = nt2 == 1 ? (f16 ? conv_stream012_h16_kernel<1, true, true> : conv_stream012_h16_kernel<1, false, true>)
           : nt2 == 2 ? (f16 ? conv_stream012_h16_kernel<2, true, true> : conv_stream012_h16_kernel<2, false, true>)
                      : (f16 ? conv_stream012_h16_kernel<3, true, true> : conv_stream012_h16_kernel<3, false, true>);
    else
        fn = nt2 == 1 ? (f16 ? conv_stream012_h16_kernel<1, true, false> : conv_stream012_h16_kernel<1, false, false>)
           : nt2 == 2 ? (f16 ? conv_stream012_h16_kernel<2, true, false> : conv_stream012_h16_kernel<2, false, false>)
                      : (f16 ? conv_stream012_h16_kernel<3, true, false> : conv_stream012_h16_kernel<3, false, false>);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(nw * 64), 0, st, a);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

}  // namespace rs
