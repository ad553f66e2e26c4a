// K3q (RS_F16XF8, LDS-DMA ring): one wide ConvNet block in split precision with the two CROSS terms of the product on the
// block-scaled 8-bit matrix instruction.
//
//   Conv1d(C_in -> C_out, k=3, 'same', bias) -> ReLU -> MaxPool1d(2,2)   (riser/nets/cnn.py:52-65)
//
// Split precision (conv_ring_h16.hip) writes every activation and weight as v = hi + lo (hi = f16(v), lo = v - hi) and spends
// three v_mfma_f32_16x16x32_f16 per 32 K on hi*hi + lo*hi + hi*lo.  The cross terms are 2^-11 of the product: their operands
// need four significant bits, not eleven (tools/fp8_cross_accuracy.py: max |dp| 2.4e-4 on the golden cases and on all 512 reads
// of the bench batch, no label differs).  Here hi*hi stays on the f16 MFMA and
//     hi_a * lo_w + lo_a * hi_w   =   [q(hi_a) | q(lo_a)] . [q(lo_w) | q(hi_w)]      (q = OCP e4m3, K concatenated)
// is ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 64 input channels and tap: per 64 K two f16 instructions (32 cycles) and one
// 8-bit instruction (32 cycles) where split precision issues six (96 cycles).
//
// Layout ("F8 rows"; api.hip: RS_F16XF8).  A row is a sequence of 128-byte panels, two per 64 channels:
//     H panel   hi16 x 64                                                  (the plain 16-bit panel of conv_ring_h16.hip)
//     F panel   [hi8 c0-31 | lo8 c0-31 | hi8 c32-63 | lo8 c32-63]           activations
//               [lo8 c0-31 | hi8 c0-31 | lo8 c32-63 | hi8 c32-63]           weights (so that unit k of A meets unit k of B)
// and a SCALE PLANE [64-channel panel][row][4] = E8M0 bytes (s0, s0 - 11, s1, s1 - 11): s = the block's exponent, one block = one
// row x 32 channels (the MX block of the instruction: lane (r, g) supplies the scale of k block g = unit g of its row).  The
// producing layer's epilogue holds the fp32 values: hi = f16(v), block maximum by DPP over the eight lanes that hold the
// block, hi8 = e4m3(hi 2^-e), lo8 = e4m3((v - hi) 2^(11 - e)), e = floor(log2 max) - 7 (the block's largest value lands in
// [128, 256): no saturation, 14 binades of full precision below it).  Weights carry one power of two per layer and plane
// (f16 weights are scaled to max |w| in [8192, 16384) already: hi x 2^-6, lo x 2^5), so their scale bytes are constants.
// The 8-bit operand of lane (r, g) is bytes [16 g, 16 g + 16) and [64 + 16 g, 64 + 16 g + 16) of its row
// (tools/ubench/mfma_f8_cross.cpp: k = 16 g + j, 64 + 16 g + j - 16) - the two ds_read_b128 the 16-bit panels already use.
//
// Staging, tile walk, ring of slabs, counted waits, epilogue through LDS: conv_ring_h16.hip, unchanged; H and F panels are
// sub-stages of the same 128-byte rows.  The F panel's scales travel as one or two more DMA pieces in the spare slots of the
// activation slab's first half.  IN_F8 = false reads split-precision rows ([hi x 32 | lo x 32] f16, three MFMAs: the first
// layer of a run), OUT_F8 = false writes them (the last layer: the head reads hi + lo).
#include "common.hpp"
#include "tile_walk.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

// Diagnostic build only (-DRS_RING_STAMPS, tools/ring_stamps.py f16xf8): s_memtime sums of the sub-stage loop per wave:
// [0] sub-stage bodies, [1] stage-end wait + barrier, [2] epilogues, [3] walk bookkeeping, [4] sub-stages, [5] total
#ifdef RS_RING_STAMPS
#define RS_STAMP(k)                                                                  \
    do {                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                           \
        unsigned long long t__;                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");  \
        __builtin_amdgcn_sched_barrier(0);                                           \
        ph[k] += t__ - tl;                                                           \
        tl = t__;                                                                    \
    } while (0)
#else
#define RS_STAMP(k) do { } while (0)
#endif

namespace rs {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int kThreads = 512;
constexpr int kRowB = 128;                      // bytes of an LDS row (one panel)
constexpr int kPieceRows = 1024 / kRowB;        // rows per DMA piece (one wave instruction)
constexpr unsigned kOob = 0x80000000u;
// E8M0 bytes of the weight planes (api.hip packs hi8 = e4m3(hi 2^-6), lo8 = e4m3(lo 2^5))
constexpr int kWScaleHi = 127 + 6, kWScaleLo = 127 - 5;

struct F8Args {
    const unsigned short* x;     // [rows_in][cpx_in]
    const unsigned short* w;     // [panel][tap][n_alloc][64]
    const float* bias;           // [n_alloc]
    float unscale;               // 2^-k of the packed weights' power-of-two scale
    unsigned short* y;           // [rows_in / 2][cpx_out]
    const unsigned char* xs;     // IN_F8: scale plane [panel64][xs_stride][4]
    unsigned char* ys;           // OUT_F8: scale plane [panel64][ys_stride][4]
    const int32_t* len;
    unsigned* sat;               // the model's overflow flag (common.hpp: f16_overflow_bits)
    unsigned x_bytes, w_bytes, y_bytes, xs_bytes, ys_bytes;
    int xs_stride, ys_stride;    // rows per plane (multiples of 4)
    int rows_in;
    int P_out;
    float inv_P_out;
    int cpx_in, cpx_out;         // row pitches in 16-bit elements
    int cols_out;                // logical channel slots of an output row (OUT_F8: 64 x panels; else 32 x panels)
    int cols_tiled;              // columns the tile grid covers (a multiple of BN)
    int n_panels;                // 128-byte K panels of an input row (IN_F8: H, F, H, F, ...)
    int n_alloc;
    int n_reads;
    int shift_out;
    WalkArgs walk;
    unsigned long long* stamps;  // diagnostic builds only
};

__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
#ifdef RS_F8_NO_H                // diagnostic build: no 16-bit MFMAs
    return c;
#endif
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// unit k of A (32 e4m3 values of one row) against unit k of B, k = 0..3, each scaled by its lane's E8M0 byte
__device__ __forceinline__ f32x4 mfma8(const u32x4& a_lo, const u32x4& a_hi, const u32x4& b_lo, const u32x4& b_hi, int sa, int sb,
                                       const f32x4& c) {
#ifdef RS_F8_NO_CROSS            // diagnostic build: hi * hi only
    return c;
#endif
    const i32x8 av = {(int)a_lo[0], (int)a_lo[1], (int)a_lo[2], (int)a_lo[3], (int)a_hi[0], (int)a_hi[1], (int)a_hi[2], (int)a_hi[3]};
    const i32x8 bv = {(int)b_lo[0], (int)b_lo[1], (int)b_lo[2], (int)b_lo[3], (int)b_hi[0], (int)b_hi[1], (int)b_hi[2], (int)b_hi[3]};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, sa, 0, sb);
}

// one LDS-DMA piece: lane l's 16 bytes at rsrc + voff land at LDS byte lds_addr + 16 l (zeros if voff is out of range)
// (TAG keeps the compiler from merging the two arms of a wave-uniform branch over different descriptors into ONE instruction
// with a selected - vector-register - descriptor)
typedef int i32x4 __attribute__((ext_vector_type(4)));
// a buffer descriptor as four plain dwords (base, base >> 32, bytes, flags): an inline-asm "s" operand of the opaque
// descriptor type is handed over in VECTOR registers once the kernel runs short of scalar ones
__device__ __forceinline__ i32x4 make_desc(const void* p, unsigned bytes) {
    const unsigned long long u = (unsigned long long)(uintptr_t)p;
    return (i32x4){(int)(unsigned)u, (int)((unsigned)(u >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
template <int TAG = 0>
__device__ __forceinline__ void dma_piece(unsigned voff, const i32x4 rsrc_, unsigned lds_addr) {
    const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    const i32x4 rsrc = {__builtin_amdgcn_readfirstlane(rsrc_[0]), __builtin_amdgcn_readfirstlane(rsrc_[1]),
                        __builtin_amdgcn_readfirstlane(rsrc_[2]), __builtin_amdgcn_readfirstlane(rsrc_[3])};
    if constexpr (TAG == 0)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                     :: "v"(voff), "s"(m0v), "s"(rsrc) : "memory");
    else
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds ; scales"
                     :: "v"(voff), "s"(m0v), "s"(rsrc) : "memory");
}

// two fp32 -> one dword of two f16 (lo in bits 0-15), round to nearest even
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ float widen16(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }

// value of the lane that holds the neighbouring output column (lane ^ 1)
__device__ __forceinline__ float swap_pair(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1 /* quad_perm [1,0,3,2] */,
                                                              0xF, 0xF, true));
}
// packed f16 maximum with the lane `ctrl` points at (all 16 lanes of a row take part)
template <int CTRL>
__device__ __forceinline__ unsigned pkmax_dpp(unsigned v) {
    const unsigned o = (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(f16x2, v), __builtin_bit_cast(f16x2, o)));
}

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// kinds of K panel
constexpr int kX3 = 0;      // [hi x 32 | lo x 32] f16: hi*hi, lo*hi, hi*lo
constexpr int kH = 1;       // hi16 x 64: (h0, h0), (h1, h1)
constexpr int kF = 2;       // 8-bit units: one scaled instruction
constexpr int kNone = 3;    // (as the previous sub-stage's kind: it left nothing pending)

// scale slab: LDS row R' holds the scales of position m0 - 4 + R' (16-byte aligned source for every lane)
constexpr int scale_pieces_of(int bm) { return ((bm + 11) * 4 + 1023) / 1024; }

template <int WM, int WN, int MT, int NT, bool IN_F8, bool OUT_F8>
__global__ __launch_bounds__(kThreads, 2) void conv_ring_f8_kernel(const F8Args a) {
    static_assert(WM * WN == 8, "8 waves per workgroup");
    static_assert(MT % 2 == 0, "an F sub-stage defers the upper half of its row blocks");
    static_assert(!OUT_F8 || NT % 2 == 0, "a scale block is 32 channels = two 16-column groups of one wave");
    constexpr int BM = WM * 16 * MT;
    constexpr int BN = WN * 16 * NT;
    constexpr int XROWS = BM + 8;                                   // slab rows: positions m0 - 1 .. m0 + BM + 6
    constexpr int XS = XROWS * kRowB;
    constexpr int WS = BN * kRowB;
    constexpr int XP = XROWS / kPieceRows;
    constexpr int XH = (XP + 1) / 2;
    constexpr int WP = BN / kPieceRows;
    constexpr int W_OFF = 2 * XS;
    constexpr int CONST_OFF = W_OFF + 3 * WS;
    constexpr int NSP = IN_F8 ? scale_pieces_of(BM) : 0;            // DMA pieces of the scale slab
    constexpr int SC_OFF = CONST_OFF + 4096;
    static_assert(BN % kPieceRows == 0 && XROWS % kPieceRows == 0, "slabs are whole pieces");
    static_assert(BN <= 256, "one piece holds the tile's bias values");
    static_assert(SC_OFF + NSP * 1024 <= 160 * 1024, "LDS capacity");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int r = lane & 15, g = lane >> 4;

    const i32x4 rs_x = make_desc(a.x, a.x_bytes);
    const i32x4 rs_w = make_desc(a.w, a.w_bytes);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ys = __builtin_amdgcn_make_buffer_rsrc(a.ys, 0, OUT_F8 ? a.ys_bytes : 0u, 0x00020000);

    // ---- DMA source maps (conv_ring_h16.hip): lane l of a piece = row l >> 3, physical slot l & 7 = logical slot ^ (row & 7)
    const int prow = lane >> 3, lslot = (lane & 7) ^ prow;
    const unsigned x_lane = (unsigned)(prow * a.cpx_in + 8 * lslot) * 2u;
    const unsigned w_lane = (unsigned)(prow * 64 + 8 * lslot) * 2u;

    struct Panel {
        int m0, n0, p;
    };
    constexpr int WPW = (WP + 7) / 8;
    constexpr int XPW0 = (XH + 7) / 8, XPW1 = (XP - XH + 7) / 8;
    static_assert(XH + NSP <= 8 * XPW0, "the scale pieces ride in the spare slots of the slab's first half");
    // first half of a slab: pieces [0, XH) of the activation rows and - with_scales: the panel is an F panel - the NSP pieces
    // of its scale slab; a wave whose share has run out re-issues the last piece (same bytes)
    auto issue_x_first = [&](const Panel& q, bool live, int xb, bool with_scales, int idx) {
        const int hi = XH + (with_scales ? NSP : 0);
        const int k = min(wave + 8 * idx, hi - 1);
        if (!IN_F8 || k < XH) {
            const unsigned off = (unsigned)(((q.m0 - 1 + k * kPieceRows) * a.cpx_in + q.p * 64) * 2) + x_lane;
            dma_piece(live ? off : kOob, rs_x, (unsigned)(xb * XS + k * 1024));
        } else {
            const int sp = k - XH;
            const unsigned off = (unsigned)((((q.p >> 1) * a.xs_stride + q.m0 - 4) * 4) + sp * 1024 + lane * 16);
            dma_piece<1>(live ? off : kOob, make_desc(a.xs, a.xs_bytes), (unsigned)(SC_OFF + sp * 1024));
        }
    };
    auto issue_x_second = [&](const Panel& q, bool live, int xb, int idx) {
        const int k = min(XH + wave + 8 * idx, XP - 1);
        const unsigned off = (unsigned)(((q.m0 - 1 + k * kPieceRows) * a.cpx_in + q.p * 64) * 2) + x_lane;
        dma_piece(live ? off : kOob, rs_x, (unsigned)(xb * XS + k * 1024));
    };
    auto issue_w_piece = [&](const Panel& q, bool live, int tap, int idx) {
        const int k = min(wave + 8 * idx, WP - 1);
        const unsigned off = (unsigned)((((q.p * 3 + tap) * a.n_alloc + q.n0 + k * kPieceRows) * 64) * 2) + w_lane;
        dma_piece(live ? off : kOob, rs_w, (unsigned)(W_OFF + tap * WS + k * 1024));
    };
    auto stage_end = [&](auto KEEP_) {
        constexpr int KEEP = decltype(KEEP_)::value;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(KEEP) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- tile walk with dead-tile elimination -----------------------------------------------------------------------
    const int tiles = a.walk.q_total;
    const int P_in_ = 2 * a.P_out;
    auto tile_origin = [&](int q, int& tm0, int& tn0) -> bool {
        int mi, nt_;
        const bool ok = walk_tile(a.walk, q, mi, nt_);
        tm0 = a.walk.m_base + mi * BM;
        tn0 = nt_ * BN;
        return ok;
    };
    TileWalk walk;
    auto next_live = [&]() __attribute__((always_inline)) {
        int q = walk.next_index(a.walk);
        while (q < tiles) {
            int tm0, tn0;
            const bool valid = tile_origin(q, tm0, tn0);
            if (valid) {
                if (!a.walk.check_dead) break;
                const int b = tm0 / P_in_;
                const int t0 = tm0 - b * P_in_;
                if (!(t0 + BM <= P_in_ && t0 >= (as_const_len(a.len)[b] >> (a.shift_out - 1)))) break;
                // an all-padding tile: zeros in every slot of its BM / 2 x BN outputs, 16 bytes at a time
                if constexpr (OUT_F8) {
                    // per 16-column group: two hi16 pieces, one hi8, one lo8
                    constexpr int PPR = (BN / 16) * 4;
                    for (int f = threadIdx.x; f < (BM / 2) * PPR; f += blockDim.x) {
                        const int rr = f / PPR, w = f - rr * PPR;
                        const int jj = w >> 2, part = w & 3;
                        const int orow = (tm0 >> 1) + rr, col = tn0 + 16 * jj;
                        if (2 * orow < a.rows_in && col < a.cols_out) {
                            const int c8 = col + 8 * (part & 1);
                            const int byte = part < 2 ? ((c8 >> 6) * 128 + (c8 & 63)) * 2
                                                      : (col >> 6) * 256 + 128 + ((col >> 5) & 1) * 64 + (col & 31) + (part - 2) * 32;
                            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(a.y) + (int64_t)orow * a.cpx_out * 2 + byte) =
                                make_uint4(0u, 0u, 0u, 0u);
                        }
                    }
                    // ... and a valid scale for every block (an E8M0 byte of 0xff would be a NaN times zero)
                    for (int f = threadIdx.x; f < (BM / 2) * (BN / 32); f += blockDim.x) {
                        const int rr = f / (BN / 32), u = f - rr * (BN / 32);
                        const int orow = (tm0 >> 1) + rr, blk = (tn0 >> 5) + u;
                        if (2 * orow < a.rows_in && 32 * blk < a.cols_out)
                            *reinterpret_cast<unsigned short*>(a.ys + ((int64_t)(blk >> 1) * a.ys_stride + orow) * 4 + (blk & 1) * 2) =
                                (unsigned short)(106 | (95 << 8));
                    }
                } else {
                    const int pieces_per_row = (tn0 + BN == a.cols_tiled ? max(a.cols_out - tn0, BN) : BN) / 8;
                    for (int f = threadIdx.x; f < (BM / 2) * pieces_per_row; f += blockDim.x) {
                        const int rr = f / pieces_per_row, cc = (f - rr * pieces_per_row) * 8;
                        const int orow = (tm0 >> 1) + rr, col = tn0 + cc;
                        if (2 * orow < a.rows_in && col < a.cols_out) {
                            unsigned short* dst = a.y + (int64_t)orow * a.cpx_out + ((col >> 5) << 6) + (col & 31);
                            *reinterpret_cast<uint4*>(dst) = make_uint4(0u, 0u, 0u, 0u);
                            *reinterpret_cast<uint4*>(dst + 32) = make_uint4(0u, 0u, 0u, 0u);
                        }
                    }
                }
            }
            q = walk.next_index(a.walk);
        }
        return q;
    };
    Panel cur;
    {
        const int o = next_live();
        if (o >= tiles) return;
        tile_origin(o, cur.m0, cur.n0);
        cur.p = 0;
    }

    // ---- fragment read addresses (bytes in LDS; + xb * XS for the activation slab in use) ---------------------
    unsigned a_rd[3][2], b_rd[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int R = wm * 16 * MT + r + tap;
            a_rd[tap][h] = (unsigned)(R * kRowB + (((4 * h + g) ^ (R & 7)) << 4));
        }
        b_rd[h] = (unsigned)(W_OFF + (wn * 16 * NT + r) * kRowB + (((4 * h + g) ^ (r & 7)) << 4));
    }
    // scale byte of lane (r, g) for row block i and tap: slab row R' = (wm MT + i) 16 + r + tap + 3
    const unsigned s_rd = (unsigned)(SC_OFF + (wm * 16 * MT + r + 3) * 4 + g);
    const int sb = (g & 1) ? kWScaleHi : kWScaleLo;                  // B units: [lo8 | hi8 | lo8 | hi8]

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // One sub-stage = tap TAP of the panel in slab xb.  As in conv_ring_h16.hip the LAST part of its MFMAs is deferred to the
    // head of the next sub-stage, behind that one's first fragment reads: for the 16-bit kinds the last pass, for an F panel
    // the upper half of the row blocks (operands kept in registers either way).
    // 24 accumulator tiles in SIX column tiles keep nothing of an F sub-stage back: the kept weight fragments (2 x NT x 4
    // registers) do not fit beside them - with the deferral <4, 2, 4, 6> spilled 120 bytes per lane and ran layer 9 in 94 us, without
    // it 36 bytes and 80 us; the other shapes gain 1-4 % from it (tools/shape_sweep.py, round 6)
    constexpr bool kFDefer = !(MT * NT >= 24 && NT >= 6);
    u32x4 keep_a[MT], keep_b[NT];                       // pending 16-bit pass
    u32x4 keep8_a[MT / 2][2], keep8_b[NT][2];           // pending 8-bit half: row blocks MT / 2 .. MT - 1
    int keep8_s[MT / 2];
    auto deferred_pass = [&](auto PREV_) __attribute__((always_inline)) {
        constexpr int PREV = decltype(PREV_)::value;
        if constexpr (PREV == kNone) {
        } else if constexpr (PREV == kF && !kFDefer) {
        } else if constexpr (PREV == kF) {
#pragma unroll
            for (int i = 0; i < MT / 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[MT / 2 + i][j] = mfma8(keep8_a[i][0], keep8_a[i][1], keep8_b[j][0], keep8_b[j][1], keep8_s[i], sb, acc[MT / 2 + i][j]);
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(keep_a[i], keep_b[j], acc[i][j]);
        }
    };
    auto substage = [&](auto KIND_, auto PREV_, auto TAP, int xb, bool have_prev, auto NDMA_, auto&& dma, auto&& between) __attribute__((always_inline)) {
        constexpr int KIND = decltype(KIND_)::value;
        constexpr int tap = decltype(TAP)::value;
        constexpr int NDMA = decltype(NDMA_)::value;
        // tap 2 of an F panel keeps nothing back: the sub-stage behind it may run a tile's epilogue, which needs the registers
        constexpr bool F_DEFER = KIND == kF && tap != 2 && kFDefer;
        constexpr int NM = KIND == kX3 ? 2 * MT * NT : KIND == kH ? MT * NT : F_DEFER ? (MT / 2) * NT : MT * NT;      // MFMAs executed here
        constexpr int GAP = NM / (NDMA + 1) > 0 ? NM / (NDMA + 1) : 1;
        const unsigned ax0 = a_rd[tap][0] + (unsigned)(xb * XS), ax1 = a_rd[tap][1] + (unsigned)(xb * XS);
        u32x4 a0[MT], b0[NT], a1[MT], b1[NT];
        int sa[MT];
#pragma unroll
        for (int j = 0; j < NT; ++j) b0[j] = *reinterpret_cast<const u32x4*>(lds + b_rd[0] + tap * WS + j * 16 * kRowB);
#pragma unroll
        for (int i = 0; i < MT; ++i) a0[i] = *reinterpret_cast<const u32x4*>(lds + ax0 + i * 16 * kRowB);
        if (have_prev) deferred_pass(PREV_);
        between();
#pragma unroll
        for (int i = 0; i < MT; ++i) a1[i] = *reinterpret_cast<const u32x4*>(lds + ax1 + i * 16 * kRowB);
#pragma unroll
        for (int j = 0; j < NT; ++j) b1[j] = *reinterpret_cast<const u32x4*>(lds + b_rd[1] + tap * WS + j * 16 * kRowB);
        if constexpr (KIND == kF) {
#pragma unroll
            for (int i = 0; i < MT; ++i) sa[i] = (int)lds[s_rd + (i * 16 + tap) * 4];
        }
        static_for<NM>([&](auto N_) {
            constexpr int n = decltype(N_)::value;
            if constexpr (KIND == kF) {
                constexpr int i = n / NT, j = n % NT;
                acc[i][j] = mfma8(a0[i], a1[i], b0[j], b1[j], sa[i], sb, acc[i][j]);
            } else {
                constexpr int pass = n / (MT * NT), ij = n % (MT * NT), i = ij / NT, j = ij % NT;
                if constexpr (pass == 0)
                    acc[i][j] = mfma16(a0[i], b0[j], acc[i][j]);           // x3: hi * hi;  H: h0 * h0
                else
                    acc[i][j] = mfma16(a1[i], b0[j], acc[i][j]);           // x3: lo * hi
            }
            if constexpr (NDMA > 0 && n % GAP == GAP - 1 && n / GAP < NDMA) dma(std::integral_constant<int, n / GAP>{});
        });
        // a small tile has fewer MFMAs here than pieces to issue: the rest behind them (the stage-end wait counts every piece)
        constexpr int ISSUED = NM / GAP < NDMA ? NM / GAP : NDMA;
        static_for<NDMA - ISSUED>([&](auto I_) { dma(std::integral_constant<int, ISSUED + decltype(I_)::value>{}); });
        if constexpr (KIND == kF && !F_DEFER) {
        } else if constexpr (KIND == kF) {
#pragma unroll
            for (int i = 0; i < MT / 2; ++i) {
                keep8_a[i][0] = a0[MT / 2 + i];
                keep8_a[i][1] = a1[MT / 2 + i];
                keep8_s[i] = sa[MT / 2 + i];
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                keep8_b[j][0] = b0[j];
                keep8_b[j][1] = b1[j];
            }
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i) keep_a[i] = KIND == kX3 ? a0[i] : a1[i];   // x3: hi * lo;  H: h1 * h1
#pragma unroll
            for (int j = 0; j < NT; ++j) keep_b[j] = b1[j];
        }
    };

    auto issue_tile_consts = [&](const Panel& q, int cb) {
        dma_piece((unsigned)(q.n0 * 4 + lane * 16), make_desc(a.bias, (unsigned)a.n_alloc * 4u), (unsigned)(CONST_OFF + cb * 2048));
        const int b0 = (q.m0 >> 1) / a.P_out;
        dma_piece((unsigned)(b0 * 4 + lane * 16), make_desc(a.len, (unsigned)a.n_reads * 4u), (unsigned)(CONST_OFF + cb * 2048 + 1024));
    };

    // ---- epilogue: bias + ReLU + MaxPool(2,2) in registers, then through a wave-private LDS image so that the tile leaves in
    // 16-byte pieces of whole output-row segments (conv_ring_h16.hip).  Per 16-column group of a pooled row the image holds
    // 64 bytes: OUT_F8 [hi16 x 16 | hi8 x 16 | lo8 x 16], else [hi16 x 16 | lo16 x 16].
    constexpr int PW = 4;
    constexpr int PITCH = NT * PW * 16 + 16;
    constexpr int NPIECE = 8 * NT * PW;
    constexpr bool SCR_IN_X = 8 * (8 * PITCH) <= XS;
    static_assert(SCR_IN_X || (4 * (8 * PITCH) <= XS && 4 * (8 * PITCH) <= WS), "epilogue scratch fits the free slabs");
    auto epilogue = [&](const Panel& q, int cb, int xb) __attribute__((always_inline)) {
        const float* lbias = reinterpret_cast<const float*>(lds + CONST_OFF + cb * 2048);
        const int* llen = reinterpret_cast<const int*>(lds + CONST_OFF + cb * 2048 + 1024);
        unsigned char* scr = (SCR_IN_X || wave < 4) ? lds + xb * XS + wave * (8 * PITCH)
                                                     : lds + W_OFF + 2 * WS + (wave - 4) * (8 * PITCH);
        float bias[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bias[j] = lbias[(wn * NT + j) * 16 + r];
        const int pr0 = q.m0 >> 1;
        const int b0 = pr0 / a.P_out;
        const int p0 = pr0 - b0 * a.P_out;
        const int c0 = q.n0 + wn * NT * 16;
        const bool odd = r & 1;
        const float us = a.unscale;
        unsigned sat = 0u;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int orow0 = (q.m0 + (wm * MT + i) * 16) >> 1;      // first of the block's 8 pooled rows
            const int my_row = orow0 + 2 * g + (odd ? 1 : 0);        // the lane's pooled row after the exchange
            unsigned keep;
            {
                const int t = p0 + (my_row - pr0);
                const int e = (int)(((float)t + 0.5f) * a.inv_P_out);
                keep = t - e * a.P_out < (llen[e] >> a.shift_out) ? ~0u : 0u;
            }
            float ca[NT], cb_[NT];                                   // channels (r & ~1, r | 1) of the lane's row
            unsigned hi[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float v0 = fmaxf(fmaxf(fmaf(acc[i][j][0], us, bias[j]), fmaf(acc[i][j][1], us, bias[j])), 0.0f);
                const float v1 = fmaxf(fmaxf(fmaf(acc[i][j][2], us, bias[j]), fmaf(acc[i][j][3], us, bias[j])), 0.0f);
                const float got = swap_pair(odd ? v0 : v1);
                ca[j] = odd ? got : v0;
                cb_[j] = odd ? v1 : got;
                hi[j] = pack2(ca[j], cb_[j]) & keep;
                sat |= f16_overflow_bits(hi[j]);
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if constexpr (OUT_F8) {
#pragma unroll
                for (int u = 0; u < NT / 2; ++u) {
                    // the block's largest hi over its 32 channels: two column groups of this lane, then the eight lanes of the
                    // lane's parity in its row of sixteen (xor 2, rotate 4, rotate 8)
                    unsigned m = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(f16x2, hi[2 * u]),
                                                                                        __builtin_bit_cast(f16x2, hi[2 * u + 1])));
                    m = pkmax_dpp<0x4E>(m);          // quad_perm [2,3,0,1]
                    m = pkmax_dpp<0x124>(m);         // row_ror:4
                    m = pkmax_dpp<0x128>(m);         // row_ror:8
                    const unsigned top = max(m & 0xffffu, m >> 16);            // activations are >= 0: the bit patterns order like the values
                    const int f = max((int)(top >> 10), 1);                   // biased f16 exponent of the maximum (subnormals: 1)
                    const float mul_hi = __builtin_bit_cast(float, (unsigned)(149 - f) << 23);     // 2^-e, e = f - 22
                    const float mul_lo = __builtin_bit_cast(float, (unsigned)(160 - f) << 23);     // 2^(11 - e)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * u + jj;
                        const float h0 = widen16((unsigned short)(hi[j] & 0xffffu)), h1 = widen16((unsigned short)(hi[j] >> 16));
                        const unsigned q8 = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(h0 * mul_hi, h1 * mul_hi, 0, false) & 0xffffu;
                        const unsigned l8 = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32((ca[j] - h0) * mul_lo, (cb_[j] - h1) * mul_lo, 0, false) & keep & 0xffffu;
                        unsigned char* dst = scr + (2 * g + (odd ? 1 : 0)) * PITCH + j * PW * 16;
                        *reinterpret_cast<unsigned*>(dst + (r & ~1) * 2) = hi[j];
                        *reinterpret_cast<unsigned short*>(dst + 32 + (r & ~1)) = (unsigned short)q8;
                        *reinterpret_cast<unsigned short*>(dst + 48 + (r & ~1)) = (unsigned short)l8;
                    }
                    // lanes r = 0, 1 of every lane group store the block's scale bytes (s, s - 11)
                    const int blk = (c0 >> 5) + u;
                    const bool ok = r < 2 && 2 * my_row < a.rows_in && 32 * blk < a.cols_out;
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)((f + 105) | ((f + 94) << 8)), rs_ys,
                                                          ok ? (unsigned)((((blk >> 1) * a.ys_stride + my_row) * 4) + (blk & 1) * 2) : kOob, 0, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    unsigned char* dst = scr + (2 * g + (odd ? 1 : 0)) * PITCH + j * PW * 16 + (r & ~1) * 2;
                    *reinterpret_cast<unsigned*>(dst) = hi[j];
                    *reinterpret_cast<unsigned*>(dst + 32) = keep &
                        pack2(ca[j] - widen16((unsigned short)(hi[j] & 0xffffu)), cb_[j] - widen16((unsigned short)(hi[j] >> 16)));
                }
            }
#pragma unroll
            for (int u = 0; u < (NPIECE + 63) / 64; ++u) {
                const int qi = lane + 64 * u;
                const int row8 = qi / (NT * PW), w = qi - row8 * (NT * PW);
                const int jj = w / PW, part = w - jj * PW;
                const int orow = orow0 + row8;
                const int col = c0 + 16 * jj;
                int byte;                                                     // of the piece inside its output row
                if constexpr (OUT_F8) {
                    const int c8 = col + 8 * (part & 1);
                    byte = part < 2 ? ((c8 >> 6) * 128 + (c8 & 63)) * 2 : (col >> 6) * 256 + 128 + ((col >> 5) & 1) * 64 + (col & 31) + (part - 2) * 32;
                } else {
                    const int c8 = col + 8 * (part & 1);
                    byte = (((c8 >> 5) << 6) + (c8 & 31) + 32 * (part >> 1)) * 2;
                }
                const bool ok = qi < NPIECE && 2 * orow < a.rows_in && col + 8 * ((part & 1) & (OUT_F8 ? (part < 2) : 1)) < a.cols_out;
                const u32x4 v = *reinterpret_cast<const u32x4*>(scr + row8 * PITCH + w * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_y, ok ? (unsigned)(orow * a.cpx_out * 2 + byte) : kOob, 0, 0);
            }
            if constexpr (!OUT_F8) {
                // the slots between the last computed 16-column group and the end of its 32-slot panel: zeros
                if (wn == WN - 1 && q.n0 + BN == a.cols_tiled && a.cols_tiled < a.cols_out) {
                    const int row8 = lane >> 2, part = lane & 3;
                    const int orow = orow0 + row8;
                    const int col = a.cols_tiled + 8 * (part & 1);
                    const int elem = ((col >> 5) << 6) + (col & 31) + 32 * (part >> 1);
                    const bool ok = lane < 32 && 2 * orow < a.rows_in && col < a.cols_out;
                    __builtin_amdgcn_raw_buffer_store_b128((u32x4){0u, 0u, 0u, 0u}, rs_y,
                                                           ok ? (unsigned)(orow * a.cpx_out + elem) * 2u : kOob, 0, 0);
                }
            }
        }
        raise_saturated(a.sat, sat);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: the first panel's slab and its first two tap slabs ----------------------------------------
#pragma unroll
    for (int idx = 0; idx < XPW0; ++idx) issue_x_first(cur, true, 0, false, idx);
#pragma unroll
    for (int idx = 0; idx < XPW1; ++idx) issue_x_second(cur, true, 0, idx);
#pragma unroll
    for (int idx = 0; idx < WPW; ++idx) issue_w_piece(cur, true, 0, idx);
#pragma unroll
    for (int idx = 0; idx < WPW; ++idx) issue_w_piece(cur, true, 1, idx);
    stage_end(std::integral_constant<int, 0>{});
    int xb = 0, cb = 0;
#ifdef RS_RING_STAMPS
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = tl, rt_begin = __builtin_amdgcn_s_memrealtime();
#endif

    bool have_prev = false, prev_tile_end = false;
    Panel done = cur;
    int done_cb = 0, done_xb = 0;
    bool more = true;
    auto nothing = [&]() {};
    // one K panel of kind KIND (the panel before it was of kind PREV): three sub-stages
    auto run_panel = [&](auto KIND_, auto PREV_) __attribute__((always_inline)) {
        constexpr int KIND = decltype(KIND_)::value;
        Panel nxt = cur;
        bool nxt_live = true;
        ++nxt.p;
        if (nxt.p == a.n_panels) {
            const int o = next_live();
            nxt.p = 0;
            nxt_live = o < tiles;
            if (nxt_live) tile_origin(o, nxt.m0, nxt.n0);
        }
        const bool tile_end = cur.p == a.n_panels - 1;
        const bool nxt_scales = IN_F8 && KIND == kH;                 // the next panel is an F panel: its scale slab travels with it
        RS_STAMP(3);
        // tap 0: this panel's tap-2 weights, first half of the next panel's slab; the previous tile's epilogue, if one is pending
        substage(KIND_, PREV_, std::integral_constant<int, 0>{}, xb, have_prev, std::integral_constant<int, WPW + XPW0>{},
                 [&](auto I_) {
                     constexpr int idx = decltype(I_)::value;
                     if constexpr (idx < WPW)
                         issue_w_piece(cur, true, 2, idx);
                     else
                         issue_x_first(nxt, nxt_live, xb ^ 1, nxt_scales, idx - WPW);
                 },
                 [&]() {
                     if (prev_tile_end) {
                         RS_STAMP(0);
                         epilogue(done, done_cb, done_xb);
                         RS_STAMP(2);
                     }
                 });
        have_prev = true;
        RS_STAMP(0);
        if (tile_end) {
            issue_tile_consts(cur, cb);
            stage_end(std::integral_constant<int, WPW + XPW0 + 2>{});
        } else {
            stage_end(std::integral_constant<int, WPW + XPW0>{});
        }
        RS_STAMP(1);
        // tap 1: the next panel's tap-0 weights, second half of its slab
        substage(KIND_, KIND_, std::integral_constant<int, 1>{}, xb, true, std::integral_constant<int, WPW + XPW1>{},
                 [&](auto I_) {
                     constexpr int idx = decltype(I_)::value;
                     if constexpr (idx < WPW)
                         issue_w_piece(nxt, nxt_live, 0, idx);
                     else
                         issue_x_second(nxt, nxt_live, xb ^ 1, idx - WPW);
                 },
                 nothing);
        RS_STAMP(0);
        stage_end(std::integral_constant<int, WPW + XPW1>{});
        RS_STAMP(1);
        // tap 2: the next panel's tap-1 weights
        substage(KIND_, KIND_, std::integral_constant<int, 2>{}, xb, true, std::integral_constant<int, WPW>{},
                 [&](auto I_) { issue_w_piece(nxt, nxt_live, 1, decltype(I_)::value); }, nothing);
        RS_STAMP(0);
        stage_end(std::integral_constant<int, WPW>{});
        RS_STAMP(1);
#ifdef RS_RING_STAMPS
        ph[4] += 3;
#endif

        prev_tile_end = tile_end;
        if (tile_end) {
            done = cur;
            done_cb = cb;
            done_xb = xb;
            cb ^= 1;
        }
        more = nxt_live;
        cur = nxt;
        xb ^= 1;
    };
    constexpr auto KX3 = std::integral_constant<int, kX3>{};
    constexpr auto KH = std::integral_constant<int, kH>{};
    constexpr auto KF = std::integral_constant<int, kF>{};
    constexpr auto KN = std::integral_constant<int, kNone>{};
    while (true) {
        if constexpr (IN_F8) {
            run_panel(KH, KN);               // n_panels is even: a tile never ends on an H panel
            run_panel(KF, KH);
        } else {
            run_panel(KX3, KX3);
        }
        if (!more) break;
    }
    // the walk's last sub-stage: its deferred part and the last tile's epilogue (`xb` has moved on: the slab is done_xb)
    if constexpr (!IN_F8) deferred_pass(KX3);
    epilogue(done, done_cb, done_xb);
    RS_STAMP(2);
#ifdef RS_RING_STAMPS
    if (a.stamps && lane == 0 && blockIdx.x < 4) {
        unsigned long long* q = a.stamps + (blockIdx.x * 8 + wave) * 8;
        for (int k = 0; k < 5; ++k) q[k] = ph[k];
        q[5] = __builtin_amdgcn_s_memtime() - t_begin;
        q[6] = 0;
        q[7] = __builtin_amdgcn_s_memrealtime() - rt_begin;        // 100 MHz ticks
    }
#endif
}

using KernelFn = void (*)(const F8Args);

struct Shape {
    int wm, wn, mt, nt;
    KernelFn fn[3];     // [x3 -> f8, f8 -> f8, f8 -> x3]
};

constexpr size_t lds_bytes_of(int bm, int bn, bool in_f8) {
    return (size_t)(2 * (bm + 8) + 3 * bn) * kRowB + 4096 + (in_f8 ? scale_pieces_of(bm) * 1024 : 0);
}

#define RS_SHAPE(WM, WN, MT, NT)                                                                                         \
    {WM, WN, MT, NT,                                                                                                     \
     {conv_ring_f8_kernel<WM, WN, MT, NT, false, true>, conv_ring_f8_kernel<WM, WN, MT, NT, true, true>,                \
      conv_ring_f8_kernel<WM, WN, MT, NT, true, false>}}
// even NT only: a scale block is two column groups of one wave
const Shape kShapes[] = {
    RS_SHAPE(8, 1, 2, 2), RS_SHAPE(8, 1, 2, 4), RS_SHAPE(8, 1, 2, 6), RS_SHAPE(4, 2, 4, 4), RS_SHAPE(4, 2, 4, 6),
    RS_SHAPE(4, 2, 2, 4), RS_SHAPE(4, 2, 2, 6), RS_SHAPE(2, 4, 4, 4), RS_SHAPE(2, 4, 2, 4),
    RS_SHAPE(4, 2, 2, 2), RS_SHAPE(2, 4, 2, 2),            // thin launches: 128 x 64, 64 x 128 (conv_ring_h16.hip's reason)
    RS_SHAPE(4, 2, 6, 4), RS_SHAPE(2, 4, 6, 2), RS_SHAPE(2, 4, 6, 4),      // 384- and 192-row tiles (conv_ring_h16.hip: batches off the multiples of 256 reads)
};
#undef RS_SHAPE
constexpr int kNumShapes = sizeof(kShapes) / sizeof(kShapes[0]);

size_t lds_bytes(const Shape& s, bool in_f8) { return lds_bytes_of(s.wm * 16 * s.mt, s.wn * 16 * s.nt, in_f8); }

// cost model in SIMD cycles per tile (conv_ring_h16.hip's, with this kernel's MFMA time per 128-byte panel and tap: an H
// or an F sub-stage is MT x NT x 32 cycles for the two waves of a SIMD, a split-precision one 3 x 16 x 2)
double tile_cost(const Shape& s, int n_panels, bool in_f8) {
    if (lds_bytes(s, in_f8) > 160 * 1024) return -1.0;
    const int bm = s.wm * 16 * s.mt, bnt = s.wn * s.nt;
    const double mfma = (in_f8 ? 2.0 : 3.0) * s.mt * s.nt * 16.0 * 2.0;
    const double dma = ((bm + 8) / 3.0 + bnt * 16.0) * 128.0 / 24.0;
    const double ldsr = 2.0 * 8.0 * (s.mt + s.nt) * 1024.0 / 256.0 * 1.2;
    const double sub = std::max(std::max(mfma, dma), ldsr) + 350.0;
    // <4, 2, 4, 6> on F8 input still spills 36-40 bytes per lane (its F sub-stages keep nothing back, see the kernel): a
    // whole-round launch runs 1.09 x the 384 x 128 tile of the same area (tools/shape_sweep.py, layer 10: 107 against 98 us)
    const double spill = in_f8 && s.mt * s.nt >= 24 && s.nt >= 6 ? 1.08 : 1.0;
    return spill * (3.0 * n_panels * sub + 1500.0 + 60.0 * s.mt * s.nt * 2.0);
}

const Shape* choose_shape(int64_t rows, int cols, int n_panels, int num_cu, bool in_f8, double* cost_out = nullptr) {
    const Shape* best = nullptr;
    double best_cost = 1e300;
    for (int k = 0; k < kNumShapes; ++k) {
        const Shape& s = kShapes[k];
        const double tile = tile_cost(s, n_panels, in_f8);
        if (tile < 0) continue;
        const int bm = s.wm * 16 * s.mt, bn = s.wn * s.nt * 16;
        const int64_t mtiles = (rows + bm - 1) / bm;
        const int64_t ntiles = (cols + bn - 1) / bn;
        const int64_t tiles = mtiles * ntiles;
        const int64_t rounds = (tiles + num_cu - 1) / num_cu;
        const double cost = (double)rounds * tile;
        if (cost < best_cost) {
            best_cost = cost;
            best = &s;
        }
    }
    if (cost_out) *cost_out = best_cost;
    return best;
}

}  // namespace

int conv_ring_f8_num_shapes() { return kNumShapes; }
bool conv_ring_f8_shape_ok(const ConvLayerDev& L, int k) {
    return k >= 0 && k < kNumShapes && lds_bytes(kShapes[k], L.f8_in) <= 160 * 1024;
}

// scale plane of a buffer of `rows` F8 rows of `cp` 16-bit elements: behind the rows, 256-byte aligned; one plane per 64-channel
// panel, f8_scale_stride(rows) rows of 4 bytes each
size_t f8_scale_offset(int64_t rows, int cp) { return ((size_t)rows * cp * 2 + 255) / 256 * 256; }
int f8_scale_stride(int64_t rows) { return (int)((rows + 3) / 4 * 4); }
size_t f8_scale_bytes(int64_t rows, int cp) { return (size_t)(cp / 128) * f8_scale_stride(rows) * 4; }

int launch_conv_ring_f8(const ConvLayerDev& L, const void* d_x, void* d_y, const int32_t* d_len, int B, int P_in,
                        int layer_index, int num_cu, int check_dead, hipStream_t st, int* bm_out, int* bn_out) {
    const int64_t rows64 = (int64_t)B * P_in;
    if (rows64 > 0x7fffffff) {
        set_error("conv_ring_f8: batch too large (%lld rows)", (long long)rows64);
        return RS_ERR_ARG;
    }
    if (!L.d_w2) {
        set_error("conv_ring_f8: layer %d has no ring-packed weights", layer_index);
        return RS_ERR_ARG;
    }
    const bool in_f8 = L.f8_in, out_f8 = L.f8_out;
    if (!in_f8 && !out_f8) {
        set_error("conv_ring_f8: layer %d neither reads nor writes F8 rows", layer_index);
        return RS_ERR_ARG;
    }
    // logical channel slots of an output row: every one of them is covered by a tile when the rows are F8 rows (the slots
    // behind the last channel are products with zero weight rows: exact zeros, a valid scale)
    const int cols_out = L.cp_out / 2;      // channel slots of a row: F8 rows hold 128 halfwords per 64 slots, x3 rows 64 per 32
    const int cols_cover = out_f8 ? cols_out : round_up(L.c_out, 16);
    const int n_panels = L.ring_panels;
    const Shape* s = choose_shape(rows64, cols_cover, n_panels, num_cu, in_f8);
    if (const char* force = L.hooks->force_ring; *force) {          // tuning aid: "layer:wm,wn,mt,nt;..."
        int l, wm, wn, mt, nt;
        for (const char* q = force; q && *q; q = strchr(q, ';') ? strchr(q, ';') + 1 : nullptr)
            if (sscanf(q, "%d:%d,%d,%d,%d", &l, &wm, &wn, &mt, &nt) == 5 && l == layer_index)
                for (int k = 0; k < kNumShapes; ++k)
                    if (kShapes[k].wm == wm && kShapes[k].wn == wn && kShapes[k].mt == mt && kShapes[k].nt == nt &&
                        lds_bytes(kShapes[k], in_f8) <= 160 * 1024)
                        s = &kShapes[k];
    }
    if (const int k = tuned_shape(L, rows64); k >= 0 && conv_ring_f8_shape_ok(L, k)) s = &kShapes[k];
    if (!s) {
        set_error("conv_ring_f8: no tile shape fits");
        return RS_ERR_ARG;
    }
    F8Args a;
    a.x = static_cast<const unsigned short*>(d_x);
    a.w = static_cast<const unsigned short*>(L.d_w2);
    a.bias = L.d_bias;
    a.unscale = L.w_unscale;
    a.y = static_cast<unsigned short*>(d_y);
    a.len = d_len;
    a.sat = L.d_sat;
    const int64_t xb = rows64 * L.cp_in * 2, wb = (int64_t)n_panels * 3 * L.plan.n_alloc * 64 * 2;
    const int64_t yb = rows64 / 2 * L.cp_out * 2;
    if (xb >= 0x80000000LL || wb >= 0x80000000LL || yb >= 0x80000000LL) {
        set_error("conv_ring_f8: a buffer exceeds the 2 GiB buffer-load window, split the batch");
        return RS_ERR_ARG;
    }
    a.x_bytes = (unsigned)xb;
    a.w_bytes = (unsigned)wb;
    a.y_bytes = (unsigned)yb;
    a.xs = in_f8 ? static_cast<const unsigned char*>(d_x) + f8_scale_offset(rows64, L.cp_in) : nullptr;
    a.xs_stride = f8_scale_stride(rows64);
    a.xs_bytes = in_f8 ? (unsigned)f8_scale_bytes(rows64, L.cp_in) : 0u;
    a.ys = out_f8 ? static_cast<unsigned char*>(d_y) + f8_scale_offset(rows64 / 2, L.cp_out) : nullptr;
    a.ys_stride = f8_scale_stride(rows64 / 2);
    a.ys_bytes = out_f8 ? (unsigned)f8_scale_bytes(rows64 / 2, L.cp_out) : 0u;
    a.rows_in = (int)rows64;
    a.P_out = P_in / 2;
    a.inv_P_out = 1.0f / (float)a.P_out;
    a.cpx_in = L.cp_in;
    a.cpx_out = L.cp_out;
    a.cols_out = cols_out;
    a.n_panels = n_panels;
    a.n_alloc = L.plan.n_alloc;
    a.n_reads = B;
    a.shift_out = layer_index + 1;
    const int BM = s->wm * 16 * s->mt, BN = s->wn * 16 * s->nt;
    const int n_mtiles = (int)((rows64 + BM - 1) / BM);
    const int n_ntiles = (cols_cover + BN - 1) / BN;
    a.cols_tiled = n_ntiles * BN;
    if (a.cols_tiled > L.plan.n_alloc) {
        set_error("conv_ring_f8: tiles of layer %d overhang the weight table (%d > %d)", layer_index, a.cols_tiled, L.plan.n_alloc);
        return RS_ERR_ARG;
    }
    if (!out_f8 && a.cols_out - a.cols_tiled > 16) {
        set_error("conv_ring_f8: %d slots behind the tiles of layer %d", a.cols_out - a.cols_tiled, layer_index);
        return RS_ERR_ARG;
    }
    const int64_t tiles = (int64_t)n_mtiles * n_ntiles;
    const unsigned grid = (unsigned)std::min<int64_t>(tiles, num_cu);
    a.walk = plan_walk(n_mtiles, n_ntiles, grid, num_cu, BM, 3.0 * BN, check_dead, !L.hooks->no_rect_order);
    a.walk.m_base = 0;
    KernelFn fn = s->fn[!in_f8 ? 0 : out_f8 ? 1 : 2];
    a.stamps = nullptr;
#ifdef RS_RING_STAMPS
    static unsigned long long* d_stamps = nullptr;
    if (!d_stamps) RS_HIP(hipMalloc(&d_stamps, 4 * 8 * 8 * 8));
    a.stamps = d_stamps;
#endif
    RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kThreads), lds_bytes(*s, in_f8), st, a);
    RS_HIP(hipGetLastError());
#ifdef RS_RING_STAMPS
    {
        RS_HIP(hipStreamSynchronize(st));
        unsigned long long hp[4 * 8 * 8];
        RS_HIP(hipMemcpy(hp, d_stamps, sizeof(hp), hipMemcpyDeviceToHost));
        for (int w = 0; w < 8; w += 4) {
            const unsigned long long* q = &hp[w * 8];
            const double n = (double)q[4];
            fprintf(stderr, "[ring-stamps] layer %d f8 %s->%s tile %dx%d panels %d wave %d: %.0f sub-stages, total %.0f cyc; per sub-stage: "
                    "body %.0f | wait+barrier %.0f | epilogue %.0f | walk %.0f; workgroup loop %.1f us (%.2f GHz)\n",
                    layer_index, in_f8 ? "f8" : "x3", out_f8 ? "f8" : "x3", BM, BN, n_panels, w, n, (double)q[5], q[0] / n, q[1] / n, q[2] / n,
                    q[3] / n, q[7] / 100.0, (double)q[5] / (q[7] * 10.0));
        }
    }
#endif
    if (bm_out) *bm_out = BM;
    if (bn_out) *bn_out = BN;
    return RS_OK;
}

}  // namespace rs
