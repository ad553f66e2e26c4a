// K3r (16-bit, LDS-DMA ring): one ConvNet block i >= 1 on the bf16 / f16 MFMA (fp32 accumulate), plain or SPLIT
// PRECISION ("x3": every activation and weight is a pair hi + lo of 16-bit values, the product is
// hi*hi + lo*hi + hi*lo on three MFMAs - ~2^-17 per operand instead of 2^-9 (bf16) / 2^-12 (f16)).
//
//   Conv1d(C_in -> C_out, k=3, 'same', bias) -> ReLU -> MaxPool1d(2,2)   (riser/nets/cnn.py:52-65)
//
// Same lowering as the round-1 register-staged 16-bit kernel (removed in round 4) (position-major activations, GEMM M = positions, N = output channels, K = 3 * C_in,
// bias + ReLU + MaxPool fused in registers) with the staging rebuilt around LDS-DMA:
//   * an LDS row is 128 bytes = one PANEL of one position / output channel: 64 input channels (plain), or 32 input
//     channels as [hi x 32 | lo x 32] (x3).  The x3 activation buffers and the weights are laid out in HBM in exactly
//     that panel-interleaved form, so both modes stage identical bytes and differ only in how the two 64-byte halves of
//     a row are paired on the matrix pipe: (h0,h0),(h1,h1) resp. (hi,hi),(lo,hi),(hi,lo);
//   * a work item is a SUB-STAGE (tile, panel, tap): 2 (plain) or 3 (x3) x MT x NT MFMAs per wave;
//   * staging is `buffer_load_dwordx4 ... offen lds` from inline asm (hipcc neither tracks nor waits for it): 1 KiB per
//     wave instruction = 8 rows; the 16-byte-slot XOR swizzle (slot ^ (row & 7), conflict-free ds_read_b128 for all
//     three tap shifts) is applied on the per-lane SOURCE address, the LDS image of a piece stays lane-linear.  Rows
//     outside the activation buffer (row -1, rows past the end) resolve to out-of-range offsets, for which the DMA
//     writes zeros (tools/ubench/lds_dma_probe.cpp);
//   * the ring: two activation slabs of (BM + 8) rows (current panel / next panel) and one weight slab per TAP
//     (BN rows each).  Tap t's weights for the next use are issued two sub-stages ahead, the next panel's slab during
//     taps 0 and 1, so every transfer has a whole sub-stage of MFMA work to land; a sub-stage ends with
//     s_waitcnt vmcnt(0) + s_barrier - no counted waits, no register staging, no LDS writes by the waves;
//   * the tile's bias / length look-ups are loaded when the tile starts and consumed right after a sub-stage's barrier,
//     when the vector-memory queue is empty, so the compiler's own wait for them costs nothing.
// The accumulation order over K is panel -> tap -> half, identical to the round-1 kernel with 64-channel panels: plain-mode
// results are bit-identical to that kernel.
#include "common.hpp"
#include "tile_walk.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

// Diagnostic build only (-DRS_RING_STAMPS, tools/ring_stamps.py): s_memtime sums of the sub-stage loop per wave:
// [0] sub-stage bodies (fragment reads + MFMAs + piece issue), [1] stage-end wait + barrier, [2] epilogues,
// [3] walk bookkeeping between sub-stages, [4] sub-stages, [5] total
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
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 512;
constexpr int kRowB = 128;                      // bytes of an LDS row (one panel)
constexpr int kPieceRows = 1024 / kRowB;        // rows per DMA piece (one wave instruction)
constexpr unsigned kOob = 0x80000000u;

struct RingArgs {
    const unsigned short* x;     // [rows_in][cpx_in]
    const unsigned short* w;     // ring packing [panel][tap][n_alloc][64]
    const float* bias;           // [n_alloc]
    float unscale;               // 2^-k of the packed weights' power-of-two scale (ConvLayerDev::w_unscale; 1 outside half precision)
    unsigned short* y;           // [rows_in / 2][cpx_out]
    const int32_t* len;
    unsigned x_bytes, w_bytes, y_bytes;
    int rows_in;
    int P_out;
    float inv_P_out;
    int cpx_in, cpx_out;         // row pitches in 16-bit elements
    int cols_out;                // logical output columns that exist in a row (plain: cpx_out; x3: 32 x panels)
    int cols_tiled;              // columns the tile grid covers (a multiple of BN); x3: the slots from here to cols_out are
                                 // written as zeros by the last tile of a row of tiles, not computed
    int n_panels;
    int n_alloc;
    int n_reads;
    int shift_out;
    unsigned* sat;               // half precision: the model's overflow flag (common.hpp: f16_overflow_bits), else null
    int terms;                   // -DRS_X3_MASK builds: which products of split precision run (1 hi*hi | 2 x lo*w hi | 4 x hi*w lo)
    WalkArgs walk;
    unsigned long long* stamps;  // diagnostic builds only
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

template <bool F16>
__device__ __forceinline__ unsigned short cvt16(float f) {
    if constexpr (F16)
        return __builtin_bit_cast(unsigned short, (_Float16)f);
    else
        return __builtin_bit_cast(unsigned short, (__bf16)f);
}
template <bool F16>
__device__ __forceinline__ float widen16(unsigned short u) {
    if constexpr (F16)
        return (float)__builtin_bit_cast(_Float16, u);
    else
        return __builtin_bit_cast(float, (unsigned)u << 16);
}

// one LDS-DMA piece: lane l's 16 bytes at rsrc + voff land at LDS byte lds_addr + 16 l (zeros if voff is out of range)
__device__ __forceinline__ void dma_piece(unsigned voff, const __amdgpu_buffer_rsrc_t rsrc, unsigned lds_addr) {
    // the LDS address is wave-uniform by construction; readfirstlane makes that provable to the compiler ("s" operand)
    const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                 :: "v"(voff), "s"(m0v), "s"(rsrc) : "memory");
#if defined(RS_EMU_DMA_X) && RS_EMU_DMA_X == 2       // measurement build: every staging piece issued twice (2 x the L2 -> LDS bytes)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                 :: "v"(voff), "s"(m0v), "s"(rsrc) : "memory");
#endif
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// two fp32 -> one dword of two 16-bit values (lo in bits 0-15), round to nearest even (v_cvt_pk_{f16,bf16}_f32)
template <bool F16>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    if constexpr (F16)
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    else
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// value of the lane that holds the neighbouring output column (lane ^ 1)
__device__ __forceinline__ float swap_pair(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1 /* quad_perm [1,0,3,2] */,
                                                              0xF, 0xF, true));
}

// physical element index of logical output column c inside a row
template <bool X3>
__device__ __forceinline__ int phys_col(int c) {
    return X3 ? ((c >> 5) << 6) + (c & 31) : c;
}

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

template <int WM, int WN, int MT, int NT, bool F16, bool X3, bool TAIL = false>
__global__ __launch_bounds__(kThreads, 2) void conv_ring_h16_kernel(const RingArgs a) {
    static_assert(WM * WN == 8, "8 waves per workgroup");
    static_assert(!TAIL || X3, "the merged tail panel exists in split precision only");
    constexpr int BM = WM * 16 * MT;
    constexpr int BN = WN * 16 * NT;
    constexpr int XROWS = BM + 8;                                   // slab rows: positions m0 - 1 .. m0 + BM + 6
    constexpr int XS = XROWS * kRowB;                               // bytes of an activation slab
    constexpr int WS = BN * kRowB;                                  // bytes of a weight slab (one tap)
    constexpr int XP = XROWS / kPieceRows;                          // DMA pieces per activation slab
    constexpr int XH = (XP + 1) / 2;                                // ... issued during tap 0 (the rest during tap 1)
    constexpr int WP = BN / kPieceRows;                             // DMA pieces per weight slab
    constexpr int W_OFF = 2 * XS;                                   // LDS: [X slab 0][X slab 1][W tap 0][W tap 1][W tap 2][bias][len]
    // per-tile constants, double buffered (a fast wave issues the next tile's while a slow one is still in its epilogue):
    // 1 KiB = the tile's BN bias values (fp32), 1 KiB = lengths of the reads the tile's rows belong to
    constexpr int CONST_OFF = W_OFF + 3 * WS;
    static_assert(BN % kPieceRows == 0 && XROWS % kPieceRows == 0, "slabs are whole pieces");
    static_assert(BN <= 256, "one piece holds the tile's bias values");
    static_assert(CONST_OFF + 4096 <= 160 * 1024, "LDS capacity");
    // TAIL (round 6): the layer's LAST panel holds at most 8 channels (67 = 64 + 3, 100 = 96 + 4): instead of three more
    // sub-stages of 32-wide K steps that are 3/32 full, its three taps are ONE K step - k-group g of a lane's fragment = tap g's
    // 8 channel slots (group 3: zero weights) - run once per tile between the last sub-stage's deferred pass and the epilogue.
    // Its operands live in two slabs of their own: [XROWS rows x (hi x 8 | lo x 8) = 32 bytes] of activations and one weight
    // slab [BN x 128 bytes] = [hi: tap 0 | tap 1 | tap 2 | 0][lo: the same], staged behind the tap-2 sub-stage of the tile's
    // FIRST panel (a tile has at least two full panels, so the pieces have landed long before the tile ends); a.n_panels
    // counts the full panels only, the tail is panel index a.n_panels of the activation rows and of the weight packing.
    constexpr int TXP = TAIL ? (XROWS + 31) / 32 : 0;               // DMA pieces of the tail activation slab (32 rows each)
    constexpr int TAIL_X_OFF = CONST_OFF + 4096;
    constexpr int TAIL_W_OFF = TAIL_X_OFF + TXP * 1024;
    constexpr int TPW = TAIL ? (TXP + WP + 7) / 8 : 0;              // tail pieces per wave (behind the first panel's tap-2 sub-stage)
    static_assert(!TAIL || TAIL_W_OFF + WP * 1024 <= 160 * 1024, "LDS capacity with the tail slabs");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

#ifdef RS_RING_STAMPS
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
    const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int r = lane & 15, g = lane >> 4;
#ifdef RS_RING_YOUNG_PRIO
    // static priority for the second-dispatched half of the workgroup (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif

    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w), 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, (unsigned)a.n_alloc * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_l =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.len), 0, (unsigned)a.n_reads * 4u, 0x00020000);

    // ---- DMA source maps: lane l of a piece = row l >> 3 of the piece, physical slot l & 7, which holds the
    // logical slot (l & 7) ^ (row & 7) (piece bases are multiples of 8 rows) --------------------------------
    const int prow = lane >> 3, lslot = (lane & 7) ^ prow;
    const unsigned x_lane = (unsigned)(prow * a.cpx_in + 8 * lslot) * 2u;
    const unsigned w_lane = (unsigned)(prow * 64 + 8 * lslot) * 2u;

    // a panel of the walk: tile origin (m0, n0) and panel index p; plain ints (a struct with a bool member goes through
    // scratch when it is copied in the loop)
    struct Panel {
        int m0, n0, p;
    };
    // ---- DMA pieces.  Every wave issues every 8th piece of a slab: piece k = wave + 8 * idx ---------------------
    constexpr int WPW = (WP + 7) / 8;                               // weight pieces per wave and sub-stage
    constexpr int XPW0 = (XH + 7) / 8, XPW1 = (XP - XH + 7) / 8;    // activation pieces per wave in taps 0 / 1
    // The number of pieces a wave issues per sub-stage is STATIC (the stage-end wait counts them): a wave whose share
    // of a slab has run out re-issues the slab's last piece (same bytes to the same LDS rows as the wave that owns it),
    // and when there is no next panel (`live` false) the pieces are issued out of range (zeros into slabs nobody reads).
    auto issue_x_piece = [&](const Panel& q, bool live, int xb, int lo, int hi, int idx) {
        const int k = min(lo + wave + 8 * idx, hi - 1);
        const unsigned off = (unsigned)(((q.m0 - 1 + k * kPieceRows) * a.cpx_in + q.p * 64) * 2) + x_lane;
        dma_piece(live ? off : kOob, rs_x, (unsigned)(xb * XS + k * 1024));
    };
    auto issue_w_piece = [&](const Panel& q, bool live, int tap, int idx) {
        const int k = min(wave + 8 * idx, WP - 1);
        const unsigned off = (unsigned)((((q.p * 3 + tap) * a.n_alloc + q.n0 + k * kPieceRows) * 64) * 2) + w_lane;
        dma_piece(live ? off : kOob, rs_w, (unsigned)(W_OFF + tap * WS + k * 1024));
    };
    auto issue_x = [&](const Panel& q, int xb, int lo, int hi) {
#pragma unroll
        for (int idx = 0; idx < (XP + 7) / 8; ++idx) issue_x_piece(q, true, xb, lo, hi, idx);
    };
    auto issue_w = [&](const Panel& q, int tap) {
#pragma unroll
        for (int idx = 0; idx < WPW; ++idx) issue_w_piece(q, true, tap, idx);
    };
    // the tail slabs of tile q (TAIL): piece k = wave + 8 idx; k < TXP: activation rows 32 k .. 32 k + 31 (lane l: row l >> 1,
    // hi (l & 1 = 0) or lo half of the tail panel's first eight slots); then the WP pieces of the merged weight slab; a wave
    // whose share has run out re-issues the last piece (same bytes, same place: the count per wave is static)
    auto issue_tail_piece = [&](const Panel& q, int idx) {
        const int k = min(wave + 8 * idx, TXP + WP - 1);
        if (k < TXP) {
            const unsigned off = (unsigned)(((q.m0 - 1 + 32 * k + (lane >> 1)) * a.cpx_in + a.n_panels * 64 + (lane & 1) * 32) * 2);
            dma_piece(off, rs_x, (unsigned)(TAIL_X_OFF + k * 1024));
        } else {
            const unsigned off = (unsigned)((((a.n_panels * 3) * a.n_alloc + q.n0 + (k - TXP) * kPieceRows) * 64) * 2) + w_lane;
            dma_piece(off, rs_w, (unsigned)(TAIL_W_OFF + (k - TXP) * 1024));
        }
    };
    // End of a sub-stage: wait until at most KEEP vector-memory operations of this wave are outstanding - the KEEP
    // pieces it issued during THIS sub-stage stay in flight across the barrier, everything older (the previous
    // sub-stage's pieces, epilogue stores, the tile constants) has landed - then the workgroup barrier that publishes
    // those pieces.  A piece is therefore readable two sub-stages after the one that issued it, which is when the issue
    // schedule below first reads it.  Compiler-issued vector-memory operations all sit between a barrier and the next
    // sub-stage's first piece, so they only ever make the counted wait stricter.
    auto stage_end = [&](auto KEEP_) {
#if defined(RS_EMU_DMA_X) && RS_EMU_DMA_X == 2
        constexpr int KEEP = 2 * decltype(KEEP_)::value;
#else
        constexpr int KEEP = decltype(KEEP_)::value;
#endif
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(KEEP) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- tile walk (tile_walk.hpp) with dead-tile elimination, as in the fp32 tiled kernels -----------------------------
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
    auto next_live = [&]() {                                       // order index of this workgroup's next live tile
        int q = walk.next_index(a.walk);
        while (q < tiles) {
            int tm0, tn0;
            const bool valid = tile_origin(q, tm0, tn0);
            if (valid) {
                if (!a.walk.check_dead) break;
                const int b = tm0 / P_in_;
                const int t0 = tm0 - b * P_in_;
                if (!(t0 + BM <= P_in_ && t0 >= (as_const_len(a.len)[b] >> (a.shift_out - 1)))) break;
                // zero-fill the BM/2 x BN output tile in 16-byte pieces (x3: the hi and the lo half of every piece)
                const int pieces_per_row = (tn0 + BN == a.cols_tiled ? max(a.cols_out - tn0, BN) : BN) / 8;
                for (int f = threadIdx.x; f < (BM / 2) * pieces_per_row; f += blockDim.x) {
                    const int rr = f / pieces_per_row, cc = (f - rr * pieces_per_row) * 8;
                    const int orow = (tm0 >> 1) + rr, col = tn0 + cc;
                    if (2 * orow < a.rows_in && col < a.cols_out) {
                        unsigned short* dst = a.y + (int64_t)orow * a.cpx_out + phys_col<X3>(col);
                        *reinterpret_cast<uint4*>(dst) = make_uint4(0u, 0u, 0u, 0u);
                        if constexpr (X3) *reinterpret_cast<uint4*>(dst + 32) = make_uint4(0u, 0u, 0u, 0u);
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

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // One sub-stage: tap TAP of the panel held in slab xb.  Its MFMAs come in PASSES over the MT x NT accumulators -
    // plain: (h0, h0), (h1, h1); x3: hi * hi, lo * hi, hi * lo - and the LAST pass is DEFERRED to the head of the next
    // sub-stage: right behind the barrier a wave issues the new sub-stage's first fragment reads and then has a pass of
    // MFMAs whose operands are already in registers (`keep_a`, `keep_b`), so the matrix pipe works while the eight
    // waves' read bursts drain through the LDS (stamps of the un-deferred loop: ~450 of ~2800 cycles per sub-stage were
    // this bubble).  Per accumulator the MFMA order is unchanged, results are bit-identical.  `between()` runs after
    // the deferred pass (the epilogue, when the previous sub-stage ended a tile).  `dma(idx)` issues the idx-th of this
    // sub-stage's NDMA staging pieces of this wave, spread over the sub-stage's own passes.
    u32x4 keep_a[MT], keep_b[NT];
#ifdef RS_ABL_NOFRAG
    u32x4 nf_a0[MT], nf_b0[NT], nf_a1[MT], nf_b1[NT];
#endif
    auto deferred_pass = [&]() {
#if defined(RS_X2_DROP) && RS_X2_DROP == 2     // measurement build: split precision without the x hi * w lo term
        if constexpr (X3) return;
#endif
#ifdef RS_X3_MASK                              // measurement build: per-layer run-time choice of the terms (tools/x3_terms_sweep.py)
        if (X3 && !(a.terms & 4)) return;
#endif
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#if defined(RS_EMU_MFMA_FRAC)                  // measurement build (tools/wino_x3_price.py): see substage()
            if (RS_EMU_MFMA_FRAC == 2 && i >= (MT + 1) / 2) continue;
#endif
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<F16>(keep_a[i], keep_b[j], acc[i][j]);
        }
    };
    // the tail K step of the tile that has just run its last sub-stage (TAIL): fragments from the tail slabs - lane (r, g) reads
    // slab row r + min(g, 2) (tap g of its output row; group 3 meets zero weights, any finite row serves) - and the three
    // products in the order of every other K step
    auto tail_pass = [&]() {
        if constexpr (TAIL) {
            u32x4 th[MT], tl_[MT];
            const unsigned ta = (unsigned)(TAIL_X_OFF + (wm * 16 * MT + r + (g < 2 ? g : 2)) * 32);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                th[i] = *reinterpret_cast<const u32x4*>(lds + ta + i * 16 * 32);
                tl_[i] = *reinterpret_cast<const u32x4*>(lds + ta + i * 16 * 32 + 16);
            }
            // column by column (the next column's weight fragments are read while this one's MFMAs run): 2 x MT + 4 fragment
            // registers live instead of 2 x (MT + NT) - the widest shapes sit at the register limit here
            const unsigned tb0 = b_rd[0] - W_OFF + TAIL_W_OFF, tb1 = b_rd[1] - W_OFF + TAIL_W_OFF;
            u32x4 uh = *reinterpret_cast<const u32x4*>(lds + tb0), ul = *reinterpret_cast<const u32x4*>(lds + tb1);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                u32x4 nh = uh, nl = ul;
                if (j + 1 < NT) {
                    nh = *reinterpret_cast<const u32x4*>(lds + tb0 + (j + 1) * 16 * kRowB);
                    nl = *reinterpret_cast<const u32x4*>(lds + tb1 + (j + 1) * 16 * kRowB);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j] = mfma16<F16>(th[i], uh, acc[i][j]);       // hi * hi
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j] = mfma16<F16>(tl_[i], uh, acc[i][j]);      // lo * hi
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j] = mfma16<F16>(th[i], ul, acc[i][j]);       // hi * lo
                uh = nh;
                ul = nl;
            }
        }
    };
    auto substage = [&](auto TAP, int xb, bool have_prev, auto NDMA_, auto&& dma, auto&& between) {
        constexpr int tap = decltype(TAP)::value;
        constexpr int NDMA = decltype(NDMA_)::value;
#if defined(RS_X2_DROP) && RS_X2_DROP == 1     // measurement build: split precision without the x lo * w hi term
        constexpr int NM = MT * NT;
#else
        constexpr int NM = (X3 ? 2 : 1) * MT * NT;                  // MFMAs of the passes executed here
#endif
        constexpr int GAP = NM / (NDMA + 1) > 0 ? NM / (NDMA + 1) : 1;
        const unsigned ax0 = a_rd[tap][0] + (unsigned)(xb * XS), ax1 = a_rd[tap][1] + (unsigned)(xb * XS);
#ifdef RS_ABL_NOFRAG
        // measurement build, results wrong: the fragments of the workgroup's FIRST sub-stage are re-used by every later one
        // (no ds_read in the steady state): the upper bound of anything that reads fewer fragment bytes per MFMA - wider
        // wave tiles, v_mfma_f32_32x32x16
        u32x4(&a0)[MT] = nf_a0, (&b0)[NT] = nf_b0, (&a1)[MT] = nf_a1, (&b1)[NT] = nf_b1;
        if (!have_prev) {
#else
        u32x4 a0[MT], b0[NT], a1[MT], b1[NT];
        {
#endif
#pragma unroll
        for (int j = 0; j < NT; ++j) b0[j] = *reinterpret_cast<const u32x4*>(lds + b_rd[0] + tap * WS + j * 16 * kRowB);
#pragma unroll
        for (int i = 0; i < MT; ++i) a0[i] = *reinterpret_cast<const u32x4*>(lds + ax0 + i * 16 * kRowB);
        }
#if defined(RS_EMU_MFMA_FRAC)
        if (have_prev && !(RS_EMU_MFMA_FRAC == 3 && tap == 0)) deferred_pass();     // tap 0 runs tap 2's deferred pass
#else
        if (have_prev) deferred_pass();
#endif
        between();
#ifdef RS_ABL_NOFRAG
        if (!have_prev)
#endif
        {
#pragma unroll
        for (int i = 0; i < MT; ++i) a1[i] = *reinterpret_cast<const u32x4*>(lds + ax1 + i * 16 * kRowB);
#pragma unroll
        for (int j = 0; j < NT; ++j) b1[j] = *reinterpret_cast<const u32x4*>(lds + b_rd[1] + tap * WS + j * 16 * kRowB);
        }
        static_for<NM>([&](auto N_) {
            constexpr int n = decltype(N_)::value;
            constexpr int pass = n / (MT * NT), ij = n % (MT * NT), i = ij / NT, j = ij % NT;
#if defined(RS_EMU_MFMA_FRAC)
            // measurement build, results wrong, timing only: what a Winograd lowering could gain AT BEST on this kernel's
            // data path - the same staging (every DMA piece, every fragment read, the epilogue) with a fraction of the
            // MFMAs: 2 = the upper half of the row blocks skipped in every pass (1/2: F(4,3)), 3 = tap 2 skipped (2/3: F(2,3))
            constexpr bool emu_skip = (RS_EMU_MFMA_FRAC == 2 && i >= (MT + 1) / 2) || (RS_EMU_MFMA_FRAC == 3 && tap == 2);
#else
            constexpr bool emu_skip = false;
#endif
            if constexpr (emu_skip) {
            } else if constexpr (pass == 0)
                acc[i][j] = mfma16<F16>(a0[i], b0[j], acc[i][j]);       // hi * hi  (plain: h0 * h0)
            else {
#ifdef RS_X3_MASK
                if (a.terms & 2)
#endif
                acc[i][j] = mfma16<F16>(a1[i], b0[j], acc[i][j]);       // lo * hi
            }
            if constexpr (NDMA > 0 && n % GAP == GAP - 1 && n / GAP < NDMA) dma(std::integral_constant<int, n / GAP>{});
        });
        // a thin tile has fewer MFMAs here than pieces to issue: the rest behind them (the stage-end wait counts every piece)
        constexpr int ISSUED = NM / GAP < NDMA ? NM / GAP : NDMA;
        static_for<NDMA - ISSUED>([&](auto I_) { dma(std::integral_constant<int, ISSUED + decltype(I_)::value>{}); });
#pragma unroll
        for (int i = 0; i < MT; ++i) keep_a[i] = X3 ? a0[i] : a1[i];   // deferred: hi * lo  (plain: h1 * h1)
#pragma unroll
        for (int j = 0; j < NT; ++j) keep_b[j] = b1[j];
    };

    // ---- per-tile constants: the tile's BN bias values and the lengths of the reads its rows belong to travel by
    // LDS-DMA too (two pieces, issued by every wave during tap 0 of the tile's last panel, identical bytes), so the
    // steady state holds NO compiler-tracked vector load whose wait could drain the pieces in flight ---------------
    auto issue_tile_consts = [&](const Panel& q, int cb) {
        dma_piece((unsigned)(q.n0 * 4 + lane * 16), rs_b, (unsigned)(CONST_OFF + cb * 2048));
        const int b0 = (q.m0 >> 1) / a.P_out;
        dma_piece((unsigned)(b0 * 4 + lane * 16), rs_l, (unsigned)(CONST_OFF + cb * 2048 + 1024));
    };
    // ---- epilogue: bias + ReLU + MaxPool(2,2) in registers, then THROUGH LDS so that the tile leaves in 16-byte
    // pieces of whole output-row segments (a lane of the 16x16 accumulator holds one channel of two pooled rows:
    // direct stores would be 2 bytes wide, 2 * MT * NT (x3: twice that) of them per lane).  Per 16-row block i a wave
    // holds 8 pooled rows x NT*16 channels: neighbouring lanes (channels c, c+1) exchange one value by DPP so that the
    // even lane owns the channel pair of pooled row 2g and the odd lane that of row 2g+1, packs it to one dword
    // (x3: a hi and a lo dword) and parks it in a wave-private scratch image laid out like the output row segment;
    // the wave then reads the image back 16 bytes per lane and stores it with buffer stores (masked pieces resolve to
    // an out-of-range offset).  The scratch aliases the activation slab the tile has just finished with; the barrier
    // behind the epilogue keeps the next sub-stage's pieces out of it.  LDS operations of one wave execute in order,
    // so the image needs no wait between its writes and its reads.
    constexpr int PW = X3 ? 4 : 2;                                  // 16-byte pieces per 16-channel group of an output row
    constexpr int PITCH = NT * PW * 16 + 16;                        // scratch row pitch: rows 2g of the 4 lane groups on distinct banks
    constexpr int NPIECE = 8 * NT * PW;                             // pieces of one block's 8 pooled rows
    // free at this point: the tile's last activation slab and the tap-2 weight slab (the next panel's is issued after
    // the barrier); short tiles put waves 4-7 into the latter
    constexpr bool SCR_IN_X = 8 * (8 * PITCH) <= XS;
    static_assert(SCR_IN_X || (4 * (8 * PITCH) <= XS && 4 * (8 * PITCH) <= WS), "epilogue scratch fits the free slabs");
    auto epilogue = [&](const Panel& q, int cb, int xb) {
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
        unsigned sat = 0u;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int orow0 = (q.m0 + (wm * MT + i) * 16) >> 1;      // first of the block's 8 pooled rows
            // the lane ends up with channels (r & ~1, r | 1) of pooled row 2g + odd: that row's validity masks the packed
            // words (read and position of the row: float quotient, exact for t < 2^16, as in conv_f32.hip)
            unsigned keep;
            {
                const int t = p0 + (orow0 + 2 * g + (odd ? 1 : 0) - pr0);
                const int e = (int)(((float)t + 0.5f) * a.inv_P_out);
                keep = t - e * a.P_out < (llen[e] >> a.shift_out) ? ~0u : 0u;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                // MaxPool, + bias, ReLU: max(a, b) + c == max(a + c, b + c) bit for bit (rounding is monotonic); the sums
                // are canonical, so the compiler emits one v_max3_f32 instead of two canonicalising v_max + max + max
                const float us = a.unscale;                  // fmaf(x, 1, b) == x + b bit for bit: nothing changes outside f16
                const float v0 = fmaxf(fmaxf(fmaf(acc[i][j][0], us, bias[j]), fmaf(acc[i][j][1], us, bias[j])), 0.0f);
                const float v1 = fmaxf(fmaxf(fmaf(acc[i][j][2], us, bias[j]), fmaf(acc[i][j][3], us, bias[j])), 0.0f);
                const float got = swap_pair(odd ? v0 : v1);
                const float ca = odd ? got : v0, cb_ = odd ? v1 : got;          // channels (r & ~1, r | 1) of row 2g + odd
                const unsigned hi = pack2<F16>(ca, cb_);
                if constexpr (F16) sat |= f16_overflow_bits(hi);
                unsigned char* dst = scr + (2 * g + (odd ? 1 : 0)) * PITCH + j * PW * 16 + (r & ~1) * 2;
                *reinterpret_cast<unsigned*>(dst) = hi & keep;
                if constexpr (X3)
                    *reinterpret_cast<unsigned*>(dst + 32) = keep &
                        pack2<F16>(ca - widen16<F16>((unsigned short)(hi & 0xffffu)), cb_ - widen16<F16>((unsigned short)(hi >> 16)));
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < (NPIECE + 63) / 64; ++u) {
                const int qi = lane + 64 * u;
                const int row8 = qi / (NT * PW), w = qi - row8 * (NT * PW);
                const int jj = w / PW, part = w - jj * PW;
                const int orow = orow0 + row8;
                const int col = c0 + 16 * jj + 8 * (part & 1);
                const int elem = phys_col<X3>(col) + (X3 ? 32 * (part >> 1) : 0);
                const bool ok = qi < NPIECE && 2 * orow < a.rows_in && col < a.cols_out;
                const u32x4 v = *reinterpret_cast<const u32x4*>(scr + row8 * PITCH + w * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_y, ok ? (unsigned)(orow * a.cpx_out + elem) * 2u : kOob, 0, 0);
            }
            if constexpr (X3) {
                // the slots between the last computed 16-column tile and the end of its 32-slot panel: zeros (the next layer
                // multiplies them by zero weights, so they must be finite); 8 rows x 2 pieces x (hi, lo) = one store of 32 lanes
                if (wn == WN - 1 && q.n0 + BN == a.cols_tiled && a.cols_tiled < a.cols_out) {
                    const int row8 = lane >> 2, part = lane & 3;
                    const int orow = orow0 + row8;
                    const int col = a.cols_tiled + 8 * (part & 1);
                    const int elem = phys_col<X3>(col) + 32 * (part >> 1);
                    const bool ok = lane < 32 && 2 * orow < a.rows_in && col < a.cols_out;
                    __builtin_amdgcn_raw_buffer_store_b128((u32x4){0u, 0u, 0u, 0u}, rs_y,
                                                           ok ? (unsigned)(orow * a.cpx_out + elem) * 2u : kOob, 0, 0);
                }
            }
        }
        if constexpr (F16) raise_saturated(a.sat, sat);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: the first panel's slab and its first two tap slabs ----------------------------------------
    issue_x(cur, 0, 0, XP);
    issue_w(cur, 0);
    issue_w(cur, 1);
    stage_end(std::integral_constant<int, 0>{});
    int xb = 0, cb = 0;
#ifdef RS_RING_STAMPS
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = tl;
#endif

    bool have_prev = false, prev_tile_end = false;
    Panel done = cur;                                  // the tile whose epilogue is pending (valid when prev_tile_end)
    int done_cb = 0, done_xb = 0;
    auto nothing = [&]() {};
    while (true) {
        // the panel after this one: the next panel of the tile, or panel 0 of the workgroup's next live tile
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
        RS_STAMP(3);

        // tap 0: this panel's tap-2 weights, first half of the next panel's activation slab; the previous tile's
        // epilogue, if one is pending, runs behind its deferred pass (and before any piece is issued into its scratch)
        substage(std::integral_constant<int, 0>{}, xb, have_prev, std::integral_constant<int, WPW + XPW0>{},
                 [&](auto I_) {
                     constexpr int idx = decltype(I_)::value;
                     if constexpr (idx < WPW)
                         issue_w_piece(cur, true, 2, idx);
                     else
                         issue_x_piece(nxt, nxt_live, xb ^ 1, 0, XH, idx - WPW);
                 },
                 [&]() {
                     if (prev_tile_end) {
                         RS_STAMP(0);
                         tail_pass();
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
        // tap 1: the next panel's tap-0 weights, second half of its activation slab
        substage(std::integral_constant<int, 1>{}, xb, true, std::integral_constant<int, WPW + XPW1>{},
                 [&](auto I_) {
                     constexpr int idx = decltype(I_)::value;
                     if constexpr (idx < WPW)
                         issue_w_piece(nxt, nxt_live, 0, idx);
                     else
                         issue_x_piece(nxt, nxt_live, xb ^ 1, XH, XP, idx - WPW);
                 },
                 nothing);
        RS_STAMP(0);
        stage_end(std::integral_constant<int, WPW + XPW1>{});
        RS_STAMP(1);
        // tap 2: the next panel's tap-1 weights
        substage(std::integral_constant<int, 2>{}, xb, true, std::integral_constant<int, WPW>{},
                 [&](auto I_) { issue_w_piece(nxt, nxt_live, 1, decltype(I_)::value); }, nothing);
        RS_STAMP(0);
        if (TAIL && cur.p == 0) {
            // the tile's tail slabs, behind its first panel (like the tile constants: a burst at the end of the sub-stage that
            // stays in flight across the barrier; the tile has at least two full panels, so it has landed long before the tile ends)
#pragma unroll
            for (int idx = 0; idx < TPW; ++idx) issue_tail_piece(cur, idx);
            stage_end(std::integral_constant<int, WPW + TPW>{});
        } else {
            stage_end(std::integral_constant<int, WPW>{});
        }
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
        if (!nxt_live) break;
        cur = nxt;
        xb ^= 1;
    }
    // the walk's last sub-stage: its deferred pass and the last tile's epilogue
#if !defined(RS_EMU_MFMA_FRAC) || RS_EMU_MFMA_FRAC != 3
    deferred_pass();
#endif
    tail_pass();
    epilogue(done, done_cb, done_xb);
    RS_STAMP(2);
#ifdef RS_RING_STAMPS
    if (a.stamps && lane == 0 && blockIdx.x < 4) {
        unsigned long long* q = a.stamps + (blockIdx.x * 8 + wave) * 8;
        for (int k = 0; k < 5; ++k) q[k] = ph[k];
        q[5] = __builtin_amdgcn_s_memtime() - t_begin;
        q[6] = t_begin - t_entry;                                  // prologue: walk set-up + first slab's round trip
        q[7] = __builtin_amdgcn_s_memrealtime() - rt_entry;        // whole workgroup, 100 MHz ticks
    }
#endif
}

using KernelFn = void (*)(const RingArgs);

struct Shape {
    int wm, wn, mt, nt;
    KernelFn fn[2][2];     // [plain, x3][bf16, f16]
    KernelFn tail[2];      // x3 with the merged tail panel [bf16, f16]; null where the tail slabs do not fit the LDS
};

constexpr size_t lds_bytes_of(int bm, int bn) { return (size_t)(2 * (bm + 8) + 3 * bn) * kRowB + 4096; }
// ... with the tail slabs (kernel: TAIL_X_OFF ...): 32-row pieces of the activation tail and the merged weight slab
constexpr size_t lds_tail_bytes_of(int bm, int bn) { return lds_bytes_of(bm, bn) + (size_t)((bm + 8 + 31) / 32) * 1024 + (size_t)bn * kRowB; }
template <int WM, int WN, int MT, int NT, bool F16>
constexpr KernelFn tail_fn() {
    if constexpr (lds_tail_bytes_of(WM * 16 * MT, WN * 16 * NT) <= 160 * 1024)
        return conv_ring_h16_kernel<WM, WN, MT, NT, F16, true, true>;
    else
        return nullptr;
}

#define RS_SHAPE(WM, WN, MT, NT)                                                                                  \
    {WM, WN, MT, NT,                                                                                              \
     {{conv_ring_h16_kernel<WM, WN, MT, NT, false, false>, conv_ring_h16_kernel<WM, WN, MT, NT, true, false>},   \
      {conv_ring_h16_kernel<WM, WN, MT, NT, false, true>, conv_ring_h16_kernel<WM, WN, MT, NT, true, true>}},     \
     {tail_fn<WM, WN, MT, NT, false>(), tail_fn<WM, WN, MT, NT, true>()}}
const Shape kShapes[] = {
    RS_SHAPE(8, 1, 2, 2), RS_SHAPE(8, 1, 2, 3), RS_SHAPE(8, 1, 2, 5), RS_SHAPE(8, 1, 2, 7), RS_SHAPE(8, 1, 4, 2),
    RS_SHAPE(8, 1, 4, 3), RS_SHAPE(8, 1, 4, 4), RS_SHAPE(4, 2, 4, 3), RS_SHAPE(4, 2, 4, 4), RS_SHAPE(4, 2, 4, 5),
    RS_SHAPE(4, 2, 4, 6), RS_SHAPE(4, 2, 2, 4), RS_SHAPE(4, 2, 2, 6), RS_SHAPE(2, 4, 4, 4),
    RS_SHAPE(2, 4, 2, 4), RS_SHAPE(2, 4, 4, 3),
    // row counts between the powers of two (round 6): a batch that is not a multiple of 256 reads leaves the layers 1.5 or 2.5
    // rounds of the 256- / 512-row tiles; 192- / 320- / 384-row tiles turn those into whole rounds
    RS_SHAPE(8, 1, 3, 3), RS_SHAPE(8, 1, 3, 4), RS_SHAPE(4, 2, 3, 4), RS_SHAPE(4, 2, 3, 5), RS_SHAPE(4, 2, 3, 6),
    RS_SHAPE(4, 2, 5, 4), RS_SHAPE(4, 2, 6, 4),
    // thin launches (round 6): a launch of a few rows pays for every staging piece of its tile's slab whether the rows exist
    // or not (a 256-row slab is 33 pieces per panel: ~650 cycles per sub-stage with 8 live rows): 64- and 32-row tiles
    RS_SHAPE(4, 2, 1, 1), RS_SHAPE(4, 2, 1, 2), RS_SHAPE(2, 4, 1, 1),
};
#undef RS_SHAPE
constexpr int kNumShapes = sizeof(kShapes) / sizeof(kShapes[0]);

size_t lds_bytes(const Shape& s, bool tail = false) {
    return tail ? lds_tail_bytes_of(s.wm * 16 * s.mt, s.wn * 16 * s.nt) : lds_bytes_of(s.wm * 16 * s.mt, s.wn * 16 * s.nt);
}

// cost model in SIMD cycles per tile: a sub-stage is bound by its MFMAs (two waves share a SIMD, 16 cycles per
// 16x16x32) or by its DMA (~24 B/clk/CU from L2), plus a fixed barrier / first-fragment bubble; the epilogue is paid
// per tile.  Rounds over the CUs quantise the whole.  (Constants refitted in round 4 on tools/shape_sweep.py at 357 x 8615,
// 300 x 7000, 128 x 16000 and 512 x 16000: the picks are within 1 % of the measured best of the table at all four.)
// tail: the layer's last panel is the merged tail (n_panels counts it): one more pass of MFMAs per tile instead of three sub-stages
double tile_cost(const Shape& s, int n_panels, bool x3, bool tail = false) {
    if (lds_bytes(s, tail) > 160 * 1024 || (tail && !s.tail[0])) return -1.0;
    const int bm = s.wm * 16 * s.mt, bnt = s.wn * s.nt;
    const double mfma = (x3 ? 3.0 : 2.0) * s.mt * s.nt * 16.0 * 2.0;
    const double dma = ((bm + 8) / 3.0 + bnt * 16.0) * 128.0 / 24.0;
    const double ldsr = 2.0 * 8.0 * (s.mt + s.nt) * 1024.0 / 256.0 * 1.2;
    const double sub = std::max(std::max(mfma, dma), ldsr) + 350.0;
    if (tail) return 3.0 * (n_panels - 1) * sub + mfma + 200.0 + 1500.0 + 60.0 * s.mt * s.nt * 1.5;
    return 3.0 * n_panels * sub + 1500.0 + 60.0 * s.mt * s.nt * (x3 ? 1.5 : 1.0);
}

const Shape* choose_shape(int64_t rows, int n16, int n_panels, int num_cu, bool x3, double* cost_out = nullptr, bool tail = false) {
    const Shape* best = nullptr;
    double best_cost = 1e300;
    for (int k = 0; k < kNumShapes; ++k) {
        const Shape& s = kShapes[k];
        const double tile = tile_cost(s, n_panels, x3, tail);
        if (tile < 0) continue;
        const int bm = s.wm * 16 * s.mt, bnt = s.wn * s.nt;
        const int64_t mtiles = (rows + bm - 1) / bm;
        const int64_t ntiles = (n16 + bnt - 1) / bnt;
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

// the planner's own estimate (its cycles) of a launch over `rows` input rows with the best shape of the table
double conv_ring_plan_cost(const ConvLayerDev& L, int64_t rows, int num_cu, bool x3) {
    double cost = 1e300;
    choose_shape(rows, round_up(L.c_out, 16) / 16, L.ring_panels, num_cu, x3, &cost, x3 && L.ring_tail);
    return cost + 2500.0;
}

int conv_ring_max_bn() { return 256; }
int conv_ring_num_shapes() { return kNumShapes; }
bool conv_ring_shape_ok(const ConvLayerDev& L, int k) {
    return k >= 0 && k < kNumShapes && lds_bytes(kShapes[k], L.ring_tail) <= 160 * 1024 && (!L.ring_tail || kShapes[k].tail[0]);
}

int launch_conv_ring_h16(const ConvLayerDev& L, const void* d_x, void* d_y, const int32_t* d_len, int B, int P_in,
                         int layer_index, int num_cu, bool f16, bool x3, int check_dead, hipStream_t st, int* bm_out,
                         int* bn_out) {
    const int64_t rows64 = (int64_t)B * P_in;
    if (rows64 > 0x7fffffff) {
        set_error("conv_ring_h16: batch too large (%lld rows)", (long long)rows64);
        return RS_ERR_ARG;
    }
    if (!L.d_w2) {
        set_error("conv_ring_h16: layer %d has no ring-packed weights", layer_index);
        return RS_ERR_ARG;
    }
    // split precision: a row holds 32 channel slots per panel and EVERY slot must be written (the next layer multiplies
    // the slots behind the last channel by zero weights: they have to be finite): the tiles cover the 16-column groups
    // that hold channels, the last tile of a row of tiles stores zeros into what is left of the last panel
    const int n16 = round_up(L.c_out, 16) / 16;
    const int n_panels = L.ring_panels;
    const bool tail = x3 && L.ring_tail;                              // the last of them is the merged tail panel
    double single_cost = 0.0;
    const Shape* s = choose_shape(rows64, n16, n_panels, num_cu, x3, &single_cost, tail);
    bool pinned = false;                                            // a forced or tuned shape runs as one launch
    if (const char* force = L.hooks->force_ring; *force) {          // tuning aid: "layer:wm,wn,mt,nt;..."
        int l, wm, wn, mt, nt;
        for (const char* q = force; q && *q; q = strchr(q, ';') ? strchr(q, ';') + 1 : nullptr)
            if (sscanf(q, "%d:%d,%d,%d,%d", &l, &wm, &wn, &mt, &nt) == 5 && l == layer_index)
                for (int k = 0; k < kNumShapes; ++k)
                    if (kShapes[k].wm == wm && kShapes[k].wn == wn && kShapes[k].mt == mt && kShapes[k].nt == nt &&
                        conv_ring_shape_ok(L, k)) {
                        s = &kShapes[k];
                        pinned = true;
                    }
    }
    if (const int k = tuned_shape(L, rows64); k >= 0 && conv_ring_shape_ok(L, k)) {
        s = &kShapes[k];
        pinned = true;
    }
    if (!s) {
        set_error("conv_ring_h16: no tile shape fits");
        return RS_ERR_ARG;
    }
    RingArgs a;
    a.x = static_cast<const unsigned short*>(d_x);
    a.w = static_cast<const unsigned short*>(L.d_w2);
    a.bias = L.d_bias;
    a.unscale = L.w_unscale;
    a.y = static_cast<unsigned short*>(d_y);
    a.len = d_len;
    const int64_t xb = rows64 * L.cp_in * 2, wb = (int64_t)n_panels * 3 * L.plan.n_alloc * 64 * 2;
    if (xb >= 0x80000000LL || wb >= 0x80000000LL) {
        set_error("conv_ring_h16: activation buffer exceeds the 2 GiB buffer-load window, split the batch");
        return RS_ERR_ARG;
    }
    const int64_t yb = rows64 / 2 * L.cp_out * 2;
    if (yb >= 0x80000000LL) {
        set_error("conv_ring_h16: output buffer exceeds the 2 GiB buffer window, split the batch");
        return RS_ERR_ARG;
    }
    a.x_bytes = (unsigned)xb;
    a.w_bytes = (unsigned)wb;
    a.y_bytes = (unsigned)yb;
    a.rows_in = (int)rows64;
    a.P_out = P_in / 2;
    a.inv_P_out = 1.0f / (float)a.P_out;
    a.cpx_in = L.cp_in;
    a.cpx_out = L.cp_out;
    a.cols_out = x3 ? L.cp_out / 2 : L.cp_out;
    a.n_panels = tail ? n_panels - 1 : n_panels;                      // the kernel's loop runs over the full panels
    a.n_alloc = L.plan.n_alloc;
    a.n_reads = B;
    a.shift_out = layer_index + 1;
    a.terms = L.x3_terms;
    a.sat = f16 ? L.d_sat : nullptr;
    a.stamps = nullptr;
#ifdef RS_RING_STAMPS
    static unsigned long long* d_stamps = nullptr;
    if (!d_stamps) RS_HIP(hipMalloc(&d_stamps, 4 * 8 * 8 * 8));
    a.stamps = d_stamps;
#endif
    // one launch over the row tiles [m_base, m_base + n_mtiles x BM) of shape sh (m_base in conv rows)
    auto launch_part = [&](const Shape& sh, int m_base, int n_mtiles) -> int {
        const int BM = sh.wm * 16 * sh.mt, BN = sh.wn * 16 * sh.nt;
        const int n_ntiles = (n16 * 16 + BN - 1) / BN;
        a.cols_tiled = n_ntiles * BN;
        if (a.cols_out - a.cols_tiled > 16) {                          // cannot happen: a panel is 32 slots, a column group 16
            set_error("conv_ring_h16: %d slots behind the tiles of layer %d", a.cols_out - a.cols_tiled, layer_index);
            return RS_ERR_ARG;
        }
        const int64_t tiles = (int64_t)n_mtiles * n_ntiles;
        const unsigned grid = (unsigned)std::min<int64_t>(tiles, num_cu);
        a.walk = plan_walk(n_mtiles, n_ntiles, grid, num_cu, BM, 3.0 * BN, check_dead, !L.hooks->no_rect_order);
        a.walk.m_base = m_base;
        KernelFn fn = tail ? sh.tail[f16 ? 1 : 0] : sh.fn[x3 ? 1 : 0][f16 ? 1 : 0];
        RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
        hipLaunchKernelGGL(fn, dim3(grid), dim3(kThreads), lds_bytes(sh, tail), st, a);
        RS_HIP(hipGetLastError());
#ifdef RS_RING_STAMPS
        {
            RS_HIP(hipStreamSynchronize(st));
            unsigned long long hp[4 * 8 * 8];
            RS_HIP(hipMemcpy(hp, d_stamps, sizeof(hp), hipMemcpyDeviceToHost));
            for (int w = 0; w < 8; w += 4) {
                const unsigned long long* q = &hp[w * 8];
                const double n = (double)q[4];
                fprintf(stderr, "[ring-stamps] layer %d %s tile %dx%d panels %d wave %d: %.0f sub-stages, total %.0f cyc; per sub-stage: "
                        "body %.0f | wait+barrier %.0f | epilogue %.0f | walk %.0f; prologue %.0f cyc; workgroup %.1f us (%.2f GHz)\n",
                        layer_index, x3 ? "x3" : "plain", BM, BN, n_panels, w, n, (double)q[5], q[0] / n, q[1] / n, q[2] / n,
                        q[3] / n, (double)q[6], q[7] / 100.0, ((double)q[5] + q[6]) / (q[7] * 10.0));
            }
        }
#endif
        return RS_OK;
    };
    // Head + tail launches are OFF for this kernel unless RS_RING_TAIL_SPLIT is set: measured over 24 random batch shapes
    // (tools/tail_split_check.py, bf16x3) the split is worth -2.7 ... +2.3 % with a mean of 0.0 - this kernel's small tiles pay a
    // prologue and an epilogue each that the planner's tile model does not price well enough to pick the winners - where the
    // fp32 Winograd kernels gain up to 4 % and never lose.
    TailSplit split;
    if (!pinned && !L.hooks->no_tail_split && L.hooks->ring_tail_split)
        split = plan_tail_split(
            kNumShapes, rows64, num_cu, single_cost, [&](int k) { return tile_cost(kShapes[k], n_panels, x3, tail); },
            [&](int k) { return kShapes[k].wm * 16 * kShapes[k].mt; },
            [&](int k) { return (n16 + kShapes[k].wn * kShapes[k].nt - 1) / (kShapes[k].wn * kShapes[k].nt); },
            [&](int64_t r, double* c) {
                const Shape* t = choose_shape(r, n16, n_panels, num_cu, x3, c, tail);
                return t ? (int)(t - kShapes) : -1;
            },
            L.hooks->tail_margin > 0 ? L.hooks->tail_margin : 0.92);      // this kernel's small tiles cost more than the model says (prologue + epilogue per tile): splits the
                        // model prices within 8 % of one launch measured at -1 % (357 x 8615), the others at +1 ... +4.5 %
    int BM, BN;
    if (split.head_shape >= 0) {
        const Shape &h = kShapes[split.head_shape], &t = kShapes[split.tail_shape];
        if (L.hooks->tail_debug)
            fprintf(stderr, "[tail-split] layer %d (ring): head %dx%dx%dx%d x %d row tiles, tail %dx%dx%dx%d; planned %.0f vs %.0f cycles\n",
                    layer_index, h.wm, h.wn, h.mt, h.nt, split.head_mtiles, t.wm, t.wn, t.mt, t.nt, split.cost, single_cost);
        BM = h.wm * 16 * h.mt;
        BN = h.wn * 16 * h.nt;
        int rc = launch_part(h, 0, split.head_mtiles);
        if (rc != RS_OK) return rc;
        const int m_base = split.head_mtiles * BM, tbm = t.wm * 16 * t.mt;
        rc = launch_part(t, m_base, (a.rows_in - m_base + tbm - 1) / tbm);
        if (rc != RS_OK) return rc;
    } else {
        BM = s->wm * 16 * s->mt;
        BN = s->wn * 16 * s->nt;
        const int rc = launch_part(*s, 0, (a.rows_in + BM - 1) / BM);
        if (rc != RS_OK) return rc;
    }
    if (bm_out) *bm_out = BM;
    if (bn_out) *bn_out = BN;
    return RS_OK;
}

}  // namespace rs
