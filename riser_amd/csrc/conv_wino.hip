// K3 (fp32, Winograd): one ConvNet block i >= 1 with the conv lowered to Winograd F(2,3) on the
// f32-input MFMA.
//
//   Conv1d(C_in -> C_out, k=3, stride 1, zero 'same' padding, bias) -> ReLU -> MaxPool1d(2,2)
//   (riser/nets/cnn.py:52-65, depth 1)
//
// MaxPool(2,2) consumes the conv outputs in pairs (y[2T], y[2T+1]) - exactly the output tile of
// the minimal-filtering algorithm F(2,3), which produces such a pair from the four inputs
// d0..d3 = x[2T-1 .. 2T+2] with 4 multiplications per input channel instead of 6:
//     m1 = (d0 - d2) * g0              m2 = (d1 + d2) * (g0 + g1 + g2) / 2
//     m4 = (d1 - d3) * g2              m3 = (d2 - d1) * (g0 - g1 + g2) / 2
//     y[2T] = m1 + m2 + m3             y[2T+1] = m2 - m3 - m4
// so the layer is FOUR GEMMs  M_j[T][n] = sum_c V_j[T][c] * U_j[n][c]  (j = 0..3) over POOLED
// rows T, i.e. 2/3 of the matrix-pipe work of the direct lowering (conv_f32.hip), and the
// output transform + bias + ReLU + MaxPool is a handful of VALU operations per pooled value in
// the epilogue: out[T][n] = relu(max(m1 + m2 + m3, m2 - m3 - m4) + bias[n]).
// The filter transform U_j is done once on the host in fp64 (rs_model_create); the input
// transform V_j is done on the fly at fragment-read time (4 LDS reads + 4 VALU per 4 fragments).
// fp32 throughout; the result differs from the direct fp32 chain by a few ulp (same error
// against an fp64 evaluation, see tests/test_gpu_parity.py).
//
// Data layout as in conv_f32.hip (position-major activations, P-row slots, zero rows beyond each
// read's length).  GEMM orientation: MFMA "A" operand = weights (rows = output channels), "B"
// operand = transformed activations (columns = pooled positions), so a lane of the 16x16
// accumulator holds 4 CONSECUTIVE CHANNELS of one pooled position: the epilogue stores 16 bytes
// per lane straight into the position-major output.
//
// LDS per item (one K chunk of KC input channels): the (2*BMP + 2) input rows of the tile are
// split by parity into two planes (x[2T+1] -> odd plane, x[2T] -> even plane) so that the four
// inputs of pooled row T are rows T, T+1 of the two planes - unit row stride across lanes, and
// with a row pitch of KC + 2 floats (= 2 mod 4) every ds_read_b32 of a 32-lane half hits 32
// distinct banks.  Weights: [component][n][KC + 2].
// Schedule: persistent 8-wave workgroups, double-buffered LDS, register prefetch of the next
// item distributed over the MFMA slots of the current one (conv_f32.hip has the rationale).
#include "common.hpp"
#include "tile_walk.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <utility>

// thin-launch fit of the planner (thin_tile_cost below): layer 11 at 16 ... 64 reads, every shape (tools/shape_sweep.py;
// profiles/r05_wino_shape_sweep_16_to_512_reads.txt).  Launch, byte and per-tile terms are conv_wino4.hip's.
#ifndef RS_WINO_THIN_LAUNCH
#define RS_WINO_THIN_LAUNCH 23500.0
#define RS_WINO_THIN_SLOT 60.0
#define RS_WINO_THIN_SLOT_MN 4.25
#define RS_WINO_THIN_ITEM (-250.0)
#define RS_WINO_THIN_BYTE 0.0062
#define RS_WINO_THIN_BYTE_FILL 0.0124
#define RS_WINO_THIN_TILE (-467.0)
#define RS_WINO_THIN_TILE_MN 554.0
#endif

namespace rs {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// compile-time loop: the body sees its index as a constant expression, so every register-array
// index in the slot loop is static whatever hipcc's unroll heuristics decide
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

struct WinoArgs {
    const float* x;
    const float* w;        // packed [n_alloc][nch][4][kc] (U0..U3), zero rows beyond c_out
    const float* bias;     // [n_alloc]
    float* y;
    const int32_t* len;
    unsigned x_bytes;      // size of the activation buffer x (rows_in * cp_in floats), < 2^31
    unsigned w_bytes;      // size of the packed weights, < 2^31
    unsigned y_bytes;      // size of the output buffer (rows_out * cp_out floats), < 2^31
    unsigned y_row_bytes;  // cp_out * 4
    unsigned len_bytes;    // B * 4
    unsigned bias_bytes;   // n_alloc * 4
    int rows_in;           // B * P_in
    int rows_out;          // B * P_out
    int P_out;
    float inv_P_out;
    int cp_in, cp_out;
    int nch;
    int shift_out;         // valid output rows of read b: len[b] >> shift_out
    WalkArgs walk;         // tile grid, order and dead-tile flag (tile_walk.hpp)
    // FUSE0 (layer 1 only): the input rows are not read from x but computed on the fly from the
    // normalised signal - ConvNet layer 0 (C_in = 1: 3 FMAs per output) folded into the staging
    const float* xs;       // normalised signals, flat [B * P0] (row pitch == P0, zero beyond each read's length),
                           // preceded by >= 16 readable bytes of zeros
    unsigned xs_bytes;
    const float* w0;       // layer 0: [cp_in][4] = (w0, w1, w2, bias)
    int n_reads;
};

// DEEP (thin launches, see conv_wino4.hip): the staging loads of an item are issued ONE ITEM earlier and stay in registers
// across the barrier.  Four-wave workgroups (one wave per SIMD) for launches of fewer tiles than CUs.  Same MFMA sequence
// per accumulator: same bits.  (The launch bound stays 512 for the four-wave shapes: a bound of 256 makes the compiler
// keep the accumulators in AGPRs, with a copy in and out per item.)
template <int WM, int WN, int MT, int NT, int KCT, bool FUSE0 = false, bool DEEP = false>
__global__ __launch_bounds__(512) void conv_wino_kernel(const WinoArgs a) {
    static_assert(WM * WN == 8 || WM * WN == 4, "8 or 4 waves per workgroup");
    static_assert(KCT % 4 == 0 && KCT >= 8, "channel chunk");
    static_assert(!(FUSE0 && DEEP), "the fused layer-0 staging keeps the default schedule");
    constexpr int kThreads = 64 * WM * WN;
    constexpr int BMP = WM * 16 * MT;                  // pooled rows per tile
    constexpr int BN = WN * 16 * NT;
    constexpr int S = KCT + 2;
    constexpr int KQ = KCT / 4;
    constexpr int PL = (BMP + 1) * S;                  // one parity plane
    constexpr int A_ELEMS = 2 * PL;
    constexpr int BUF = A_ELEMS + 4 * BN * S;
    // staging passes: a pass of the workgroup covers RPT slab rows x KQ 16-byte units of the X slab,
    // or NPP output channels x 4 components x KQ units of the weight slab
    constexpr int RPT = (kThreads / KQ) & ~1;
    constexpr int A_ROWS = 2 * BMP + 2;
    constexpr int A_PER = (A_ROWS + RPT - 1) / RPT;
    constexpr int NPP = kThreads / (4 * KQ);
    constexpr int B_PER = (BN + NPP - 1) / NPP;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int r = lane & 15, kq = lane >> 4;

    // ---- staging map ------------------------------------------------------------------------------
    // Thread t of a pass owns slab row t / KQ (global input row 2*m0p - 1 + row), channels 4*(t % KQ)
    // of the chunk; pass u adds u * RPT rows.  RPT is even, so the parity plane of a thread's rows is
    // fixed: even slab row = odd global row -> odd plane (offset 0), odd slab row -> even plane
    // (offset PL), index row >> 1, and pass u is an immediate LDS offset.  Weights: thread t owns
    // output channel t / (4*KQ), unit t % (4*KQ) = comp * KQ + c4 of the packed [n][nch][4][KC] row.
    // Global loads are BUFFER loads with a 32-bit byte offset: rows before the first / after the last
    // row of the activation buffer, the K padding of the last chunk, idle threads and the prefetch
    // after the last item all resolve to an out-of-range offset, which the hardware answers with
    // zeros - no address arithmetic beyond one add per unit, no branches, no zero page.
    const int a_row = tid / KQ, a_c4 = tid - a_row * KQ;
    const bool a_act = a_row < RPT;
    const int b_n = tid / (4 * KQ), b_rem = tid - b_n * (4 * KQ);
    const bool b_act = b_n < NPP;
    const int a_st = ((a_row & 1) ? PL : 0) + (a_row >> 1) * S + 4 * a_c4;          // + u * (RPT/2) * S
    const int b_st = A_ELEMS + ((b_rem / KQ) * BN + b_n) * S + 4 * (b_rem % KQ);    // + u * NPP * S
    const unsigned a_tb = (unsigned)(a_row * a.cp_in + 4 * a_c4) * 4u;
    const unsigned b_tb = (unsigned)(b_n * a.nch * 4 * KCT + 4 * b_rem) * 4u;
    const unsigned a_step = (unsigned)(RPT * a.cp_in) * 4u;
    const unsigned b_step = (unsigned)(NPP * a.nch * 4 * KCT) * 4u;
    constexpr unsigned kOob = 0x80000000u;             // >= num_records of either buffer (host checks)
    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);

    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_len =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.len), 0, a.len_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_bias =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.bias_bytes, 0x00020000);

    // FUSE0: input row g of this layer = output row g of layer 0 = relu(max(conv(x)[2t], conv(x)[2t+1]) + b)
    // of read b = g / P_in at t = g % P_in, computed from the four samples x[2g-1 .. 2g+2] of the FLAT
    // signal buffer (pitch P0 = 2 * P_in, so sample 2t of read b is element 2g; the zero fill beyond
    // each read's length supplies both 'same' pads).  A thread's four channels are fixed (4 * a_c4 ..),
    // so their (w0, w1, w2, bias) live in registers.  A tile spans at most two reads (host-checked).
    // (the descriptor starts 16 bytes before the first sample - the caller guarantees four zero floats
    // there - so the offset of x[2g-1] is never negative: a load that STARTS out of range returns zeros
    // for all four dwords, including the three in-range ones)
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(FUSE0 ? a.xs - 4 : a.x), 0, FUSE0 ? a.xs_bytes + 16u : 0u, 0x00020000);
    const int P_in = 2 * a.P_out;
    const const_len_ptr clen = as_const_len(a.len);
    f32x4 w0r[4];
    if constexpr (FUSE0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            w0r[q] = *reinterpret_cast<const f32x4*>(a.w0 + (size_t)(4 * (a_act ? a_c4 : 0) + q) * 4);
    }
    int f_g0 = 0, f_base = 0, f_gb = 0, f_lim_lo = 0, f_lim_hi = 0;

    u32x4 ra[A_PER], rb[B_PER];
    unsigned a_ib = kOob, b_ib = kOob;                  // per-item byte offsets of this thread's first units
    auto item_offsets = [&](int m0p, int n0, int c, bool live) {
        if constexpr (FUSE0) {
            const int g0 = 2 * m0p - 1;                 // first slab row (global input row)
            const int b_lo = (g0 < 0 ? 0 : g0) / P_in;
            f_g0 = g0;
            f_base = b_lo * P_in;
            f_gb = f_base + P_in;
            f_lim_lo = clen[b_lo < a.n_reads ? b_lo : a.n_reads - 1] >> 1;
            f_lim_hi = b_lo + 1 < a.n_reads ? clen[b_lo + 1] >> 1 : 0;
            if (b_lo >= a.n_reads) f_lim_lo = 0;
            a_ib = (live && a_act) ? (unsigned)(2 * (g0 + a_row) - 1 + 4) * 4u : kOob;     // g0 + a_row >= -1: offset >= 4
        } else {
            const bool a_ok = live && a_act && c * KCT + 4 * a_c4 < a.cp_in;
            a_ib = a_ok ? a_tb + (unsigned)((2 * m0p - 1) * a.cp_in + c * KCT) * 4u : kOob;
        }
        b_ib = (live && b_act) ? b_tb + (unsigned)((n0 * a.nch + c) * 4 * KCT) * 4u : kOob;
#ifdef RS_ABL_NOLOAD                                    // timing experiment only: every staging load out of range
        a_ib = kOob;
        b_ib = kOob;
#endif
    };
    auto load_unit = [&](auto U) {
        constexpr int u = decltype(U)::value;
        if constexpr (u < A_PER) {
            unsigned off = a_ib + (unsigned)u * (FUSE0 ? (unsigned)(2 * RPT * 4) : a_step);
            if constexpr ((u + 1) * RPT > A_ROWS) off = (a_row + u * RPT < A_ROWS) ? off : kOob;
            ra[u] = __builtin_amdgcn_raw_buffer_load_b128(FUSE0 ? rs_s : rs_x, off, 0, 0);
        } else {
            constexpr int v = u - A_PER;
            unsigned off = b_ib + (unsigned)v * b_step;
            if constexpr ((v + 1) * NPP > BN) off = (b_n + v * NPP < BN) ? off : kOob;
            rb[v] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0);
        }
    };
    auto store_unit = [&](auto U, float* buf) {
        constexpr int u = decltype(U)::value;
        if constexpr (u < A_PER) {
            bool act = a_act;
            if constexpr ((u + 1) * RPT > A_ROWS) act = act && (a_row + u * RPT < A_ROWS);
#ifdef RS_ABL_NOLDSW
            asm volatile("" ::"v"(ra[u].x), "v"(ra[u].y), "v"(ra[u].z), "v"(ra[u].w));
            act = false;
#endif
            if constexpr (FUSE0) {
                const int g = f_g0 + a_row + u * RPT;
                const bool hi = g >= f_gb;
                const bool valid = g >= 0 && (g - (hi ? f_gb : f_base)) < (hi ? f_lim_hi : f_lim_lo);
                const f32x4 xv = __builtin_bit_cast(f32x4, ra[u]);             // x[2g-1], x[2g], x[2g+1], x[2g+2]
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {                                  // same fmaf chains as conv0_kernel
                    const float e = fmaf(w0r[q][2], xv[2], fmaf(w0r[q][1], xv[1], fmaf(w0r[q][0], xv[0], w0r[q][3])));
                    const float f = fmaf(w0r[q][2], xv[3], fmaf(w0r[q][1], xv[2], fmaf(w0r[q][0], xv[1], w0r[q][3])));
                    o[q] = valid ? fmaxf(fmaxf(e, f), 0.0f) : 0.0f;
                }
                ra[u] = __builtin_bit_cast(u32x4, o);
            }
            if (act) {
                uint2* d = reinterpret_cast<uint2*>(buf + a_st + u * (RPT / 2) * S);
                d[0] = make_uint2(ra[u].x, ra[u].y);
                d[1] = make_uint2(ra[u].z, ra[u].w);
            }
        } else {
            constexpr int v = u - A_PER;
            bool act = b_act;
            if constexpr ((v + 1) * NPP > BN) act = act && (b_n + v * NPP < BN);
#ifdef RS_ABL_NOLDSW
            asm volatile("" ::"v"(rb[v].x), "v"(rb[v].y), "v"(rb[v].z), "v"(rb[v].w));
            act = false;
#endif
            if (act) {
                uint2* d = reinterpret_cast<uint2*>(buf + b_st + v * NPP * S);
                d[0] = make_uint2(rb[v].x, rb[v].y);
                d[1] = make_uint2(rb[v].z, rb[v].w);
            }
        }
    };

    // ---- tile walk (tile_walk.hpp): XCD-contiguous blocks per round, rotation and zero-fill of tiles that lie
    // entirely in a shorter read's padding ---------------------------------------------------------------
    const int tiles = a.walk.q_total;
    auto tile_origin = [&](int q, int& tm0, int& tn0) -> bool {
        int mi, nt_;
        const bool ok = walk_tile(a.walk, q, mi, nt_);
        tm0 = a.walk.m_base + mi * BMP;
        tn0 = nt_ * BN;
        return ok;
    };
    TileWalk walk;
    auto order_index = [&]() { return walk.next_index(a.walk); };
    auto next_live = [&]() {
        int q = order_index();
        while (q < tiles) {
            int tm0, tn0;
            if (tile_origin(q, tm0, tn0)) {
                if (!a.walk.check_dead) break;
                const int b = tm0 / a.P_out;
                const int t0 = tm0 - b * a.P_out;
                if (!(t0 + BMP <= a.P_out && t0 >= (clen[b] >> a.shift_out))) break;
                const int pieces_per_row = BN / 4;
                for (int f = threadIdx.x; f < BMP * pieces_per_row; f += blockDim.x) {
                    const int rr = f / pieces_per_row, cc = (f - rr * pieces_per_row) * 4;
                    const int prow = tm0 + rr, col = tn0 + cc;
                    if (prow < a.rows_out && col < a.cp_out)
                        *reinterpret_cast<float4*>(a.y + (int64_t)prow * a.cp_out + col) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            q = order_index();
        }
        return q;
    };
    int o = next_live();
    if (o >= tiles) return;

    f32x4 acc[MT][NT][4];                             // small shapes: written by the first item of every tile
    if constexpr (MT * NT > 6) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[i][j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    int c = 0;
    int m0, n0;
    tile_origin(o, m0, n0);
    item_offsets(m0, n0, 0, true);
    static_for<A_PER + B_PER>([&](auto U) { load_unit(U); });
    static_for<A_PER + B_PER>([&](auto U) { store_unit(U, lds); });
    // the item after (oo, cc) of this workgroup's walk; oo >= tiles: none
    auto advance = [&](int& oo, int& cc, int& mm, int& nn) {
        if (oo >= tiles) return;
        if (++cc == a.nch) {
            cc = 0;
            oo = next_live();
            if (oo < tiles) tile_origin(oo, mm, nn);
        }
    };
    int o1 = o, c1 = 0, m1 = m0, n1 = n0;              // DEEP: the next item, already on its way in ra / rb
    if constexpr (DEEP) {
        advance(o1, c1, m1, n1);
        item_offsets(m1, n1, c1, o1 < tiles);
        static_for<A_PER + B_PER>([&](auto U) { load_unit(U); });
    }
    __syncthreads();
    int buf = 0;

    const int a_rd = (wm * 16 * MT + r) * S + kq;                  // + i*16*S (+ S) + c0, (+ PL for the even plane)
    const int b_rd = A_ELEMS + (wn * 16 * NT + r) * S + kq;        // + (comp*BN + j*16)*S + c0

    while (true) {
        int nc = c + 1, no = o;
        int nm0 = m0, nn0 = n0;
        int o2 = o1, c2 = c1, m2 = m1, n2 = n1;
        if constexpr (DEEP) {
            nc = c1, no = o1, nm0 = m1, nn0 = n1;
            advance(o2, c2, m2, n2);
        } else if (nc == a.nch) {
            nc = 0;
            no = next_live();
        }
        const bool has_next = no < tiles;
        if constexpr (!DEEP)
            if (has_next && nc == 0) tile_origin(no, nm0, nn0);
        const float* Ab = lds + buf * BUF + a_rd;
        const float* Bb = lds + buf * BUF + b_rd;
        float* nbuf = lds + (buf ^ 1) * BUF;
        if constexpr (DEEP)
            item_offsets(m2, n2, c2, o2 < tiles);
        else
            item_offsets(nm0, nn0, nc, has_next);

        constexpr int NSLOTS = 4 * KQ;                 // slot = (k-step, component): MT * NT MFMAs
        constexpr int UNITS = A_PER + B_PER;
        // distributed staging: unit u of the next item is loaded after slot ld(u) and written to the
        // other LDS buffer after slot ld(u) + DIST, so only ~DIST * UNITS / NSLOTS units are in
        // registers at any time (more where the accumulators leave registers free: the early
        // layers stream from HBM and need the longer flight time)
        constexpr int DIST_LONG = NSLOTS - UNITS > 4 ? NSLOTS - UNITS : 4;
        constexpr int DIST = MT * NT <= 4 ? DIST_LONG : MT * NT <= 6 ? (NSLOTS / 2 < DIST_LONG ? NSLOTS / 2 : DIST_LONG)
                                                                  : (NSLOTS >= 20 ? 5 : 4);
        constexpr int SPAN = NSLOTS - DIST;            // load slots 0 .. SPAN-1
        constexpr int EPI_SLOT = NSLOTS - 3;           // where the epilogue's look-ups are issued
        // the 128-accumulator shapes have no registers to spare (and their tiles are long: an exposed L2
        // round trip and 128 v_mov per tile are noise): they look up / zero in the epilogue instead
        constexpr bool HOIST = MT * NT <= 6;
        constexpr bool HOIST_BIAS = HOIST;
        // epilogue look-ups (per-read lengths of the lane's rows, bias of its channels): issued
        // UNCONDITIONALLY near the end of every item so that their L2 latency hides under the last
        // MFMAs; only the tile's last item uses them
        const int b0 = m0 / a.P_out;
        const int p0 = m0 - b0 * a.P_out;
        int pin_[MT];
        unsigned lenv_[MT];
        u32x4 bi_[NT];
        auto run_item = [&](auto FIRST) {
            constexpr bool first = decltype(FIRST)::value;     // first item of a tile: accumulate onto zero
            // a slot of the thin shapes is one or two MFMAs (32 / 64 cycles): an LDS read issued one slot ahead is not back
            // in time - their weight fragments are read four (two) slots ahead, the raw rows right behind the transform
            constexpr bool THIN = MT * NT <= 2;
            constexpr int AH = THIN ? 4 / (MT * NT) : 1;   // slots of look-ahead of the weight-fragment reads
            constexpr int RD = THIN ? 0 : 1;               // component slot that reads the raw rows of the next k-step
            float dr[MT][4];                           // raw inputs d0..d3 of the lane's pooled rows (next k-step)
            float uf[AH + 1][NT];                      // weight fragments, a ring over the slots in flight
            float v[MT][4];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                dr[i][0] = Ab[i * 16 * S];
                dr[i][2] = Ab[i * 16 * S + S];
                dr[i][1] = Ab[PL + i * 16 * S];
                dr[i][3] = Ab[PL + i * 16 * S + S];
            }
            static_for<AH>([&](auto SL) {
                constexpr int sl = decltype(SL)::value;
#pragma unroll
                for (int j = 0; j < NT; ++j) uf[sl][j] = Bb[((sl & 3) * BN + j * 16) * S + 4 * (sl >> 2)];
            });
            static_for<NSLOTS>([&](auto SL) {
                constexpr int sl = decltype(SL)::value;
                constexpr int st = sl >> 2, comp = sl & 3;
                if constexpr (comp == 0) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const float d0 = dr[i][0], d1 = dr[i][1], d2 = dr[i][2], d3 = dr[i][3];
#ifdef RS_ABL_NOXFORM                                      // timing experiment only
                        v[i][0] = d0;
                        v[i][1] = d1;
                        v[i][2] = d2;
                        v[i][3] = d3;
#else
                        v[i][0] = d0 - d2;
                        v[i][1] = d1 + d2;
                        v[i][2] = d2 - d1;
                        v[i][3] = d1 - d3;
#endif
                    }
                }
                if constexpr (comp == RD && st + 1 < KQ) {
                    constexpr int c0 = 4 * (st + 1);
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        dr[i][0] = Ab[i * 16 * S + c0];
                        dr[i][2] = Ab[i * 16 * S + S + c0];
                        dr[i][1] = Ab[PL + i * 16 * S + c0];
                        dr[i][3] = Ab[PL + i * 16 * S + S + c0];
                    }
                }
                if constexpr (sl + AH < NSLOTS) {
                    constexpr int nst = (sl + AH) >> 2, ncomp = (sl + AH) & 3;
#pragma unroll
                    for (int j = 0; j < NT; ++j) uf[(sl + AH) % (AH + 1)][j] = Bb[(ncomp * BN + j * 16) * S + 4 * nst];
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if constexpr (first && st == 0)
                            acc[i][j][comp] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                uf[sl % (AH + 1)][j], v[i][comp], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        else
                            acc[i][j][comp] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[sl % (AH + 1)][j], v[i][comp],
                                                                                   acc[i][j][comp], 0, 0, 0);
                    }
#ifndef RS_WINO_NO_SGB2
                {   // one LDS read in the shadow of each of the first MFMAs, so the next slot's fragments are
                    // in flight early without a read burst ahead of the MFMAs
                    constexpr int n_rd = (sl + AH < NSLOTS ? NT : 0) + ((comp == RD && st + 1 < KQ) ? 2 * MT : 0);
                    constexpr int n_pair = n_rd < MT * NT ? n_rd : MT * NT;
                    static_for<n_pair>([&](auto) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    });
                    if constexpr (MT * NT - n_pair > 0) __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - n_pair, 0);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
                static_for<UNITS>([&](auto U) {
                    constexpr int u = decltype(U)::value;
                    if constexpr (DEEP) {
                        // the unit loaded during the previous item goes to the next item's buffer, and its registers
                        // leave again for the item after that
                        if constexpr ((u * NSLOTS) / UNITS == sl) {
                            if (has_next) store_unit(U, nbuf);
                            load_unit(U);
                        }
                    } else {
                        if constexpr ((u * SPAN) / UNITS == sl) load_unit(U);
                        if constexpr ((u * SPAN) / UNITS + DIST == sl) {
                            if (has_next) store_unit(U, nbuf);
                        }
                    }
                });
                if constexpr (HOIST && sl == EPI_SLOT) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int t = p0 + (wm * MT + i) * 16 + r;
                        const int e = (int)(((float)t + 0.5f) * a.inv_P_out);     // t < P_out + BMP < 2^16: exact
                        pin_[i] = t - e * a.P_out;
                        // rows past the end of the batch look up a read index >= B: out of range -> 0 -> masked
                        lenv_[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_len, (unsigned)(b0 + e) * 4u, 0, 0);
                    }
                    if constexpr (HOIST_BIAS) {
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            bi_[j] = __builtin_amdgcn_raw_buffer_load_b128(
                                rs_bias, (unsigned)(n0 + (wn * NT + j) * 16 + 4 * kq) * 4u, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        if constexpr (HOIST) {
            if (c == 0)
                run_item(std::true_type{});
            else
                run_item(std::false_type{});
        } else {
            run_item(std::false_type{});
        }

        if (c == a.nch - 1) {
            // ---- epilogue: output transform + bias + ReLU + MaxPool, 16-byte buffer stores (rows past the
            // end of the batch and columns past cp_out resolve to out-of-range offsets and are dropped;
            // rows beyond their read's length are written as zeros) -------------------------------------
            if constexpr (!HOIST) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int t = p0 + (wm * MT + i) * 16 + r;
                    const int e = (int)(((float)t + 0.5f) * a.inv_P_out);
                    pin_[i] = t - e * a.P_out;
                    lenv_[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_len, (unsigned)(b0 + e) * 4u, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bi_[j] = __builtin_amdgcn_raw_buffer_load_b128(
                        rs_bias, (unsigned)(n0 + (wn * NT + j) * 16 + 4 * kq) * 4u, 0, 0);
            }
            unsigned rowoff_[MT];
            bool valid_[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                rowoff_[i] = (unsigned)(m0 + (wm * MT + i) * 16 + r) * a.y_row_bytes;
                valid_[i] = pin_[i] < (int)(lenv_[i] >> a.shift_out);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = n0 + (wn * NT + j) * 16 + 4 * kq;
                const unsigned coloff = col < a.cp_out ? (unsigned)col * 4u : kOob;
                const f32x4 bi = __builtin_bit_cast(f32x4, bi_[j]);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    f32x4 o4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float m1 = acc[i][j][0][q], m2 = acc[i][j][1][q], m3 = acc[i][j][2][q],
                                    m4 = acc[i][j][3][q];
                        const float y0 = (m1 + m2) + m3;
                        const float y1 = (m2 - m3) - m4;
                        o4[q] = valid_[i] ? fmaxf(fmaxf(y0, y1) + bi[q], 0.0f) : 0.0f;
                    }
#ifdef RS_ABL_NOSTORE
                    asm volatile("" ::"v"(o4[0]), "v"(o4[1]), "v"(o4[2]), "v"(o4[3]));
#else
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o4), rs_y, rowoff_[i] + coloff, 0, 0);
#endif
                    if constexpr (!HOIST) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[i][j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        }
        if (!has_next) break;
#ifndef RS_ABL_NOBARRIER                                // timing experiment only
        __syncthreads();
#endif
        buf ^= 1;
        o = no;
        c = nc;
        m0 = nm0;
        n0 = nn0;
        if constexpr (DEEP) o1 = o2, c1 = c2, m1 = m2, n1 = n2;
    }
}

using KernelFn = void (*)(const WinoArgs);

struct Shape {
    int wm, wn, mt, nt;
    KernelFn fn[3];        // chunk = 16, 20, 24
    KernelFn deep[3];      // the same tile with staging loads one item ahead (thin launches), or null
};
// layer 1 of the shipped net (20 -> 30 channels) with layer 0 folded into its staging
const KernelFn kFusedL1 = conv_wino_kernel<8, 1, 2, 2, 20, true>;
constexpr int kFusedBMP = 8 * 16 * 2, kFusedBN = 32;

#define RS_SHAPE(WM, WN, MT, NT)                                                                       \
    {WM, WN, MT, NT, {conv_wino_kernel<WM, WN, MT, NT, 16>, conv_wino_kernel<WM, WN, MT, NT, 20>,     \
                      conv_wino_kernel<WM, WN, MT, NT, 24>}, {nullptr, nullptr, nullptr}}
#define RS_SHAPE_D(WM, WN, MT, NT)                                                                                 \
    {WM, WN, MT, NT, {conv_wino_kernel<WM, WN, MT, NT, 16>, conv_wino_kernel<WM, WN, MT, NT, 20>,                 \
                      conv_wino_kernel<WM, WN, MT, NT, 24>},                                                      \
     {conv_wino_kernel<WM, WN, MT, NT, 16, false, true>, conv_wino_kernel<WM, WN, MT, NT, 20, false, true>,       \
      conv_wino_kernel<WM, WN, MT, NT, 24, false, true>}}
// four-wave shapes exist in the one-item-ahead form only
#define RS_SHAPE_4(WM, WN, MT, NT)                                                                                        \
    {WM, WN, MT, NT, {conv_wino_kernel<WM, WN, MT, NT, 16, false, true>, conv_wino_kernel<WM, WN, MT, NT, 20, false, true>, \
                      conv_wino_kernel<WM, WN, MT, NT, 24, false, true>}, {nullptr, nullptr, nullptr}}
const Shape kShapes[] = {
    // all 8 waves stacked along pooled rows
    RS_SHAPE(8, 1, 2, 2), RS_SHAPE(8, 1, 2, 3), RS_SHAPE(8, 1, 2, 4), RS_SHAPE(8, 1, 1, 5), RS_SHAPE(8, 1, 1, 6),
    RS_SHAPE(8, 1, 1, 7), RS_SHAPE(8, 1, 1, 8),
    // 4 x 2 waves
    RS_SHAPE(4, 2, 2, 3), RS_SHAPE(4, 2, 2, 4), RS_SHAPE(4, 2, 1, 5), RS_SHAPE(4, 2, 1, 7), RS_SHAPE(4, 2, 1, 8),
    // 2 x 4 waves (few rows)
    RS_SHAPE(2, 4, 2, 2), RS_SHAPE(2, 4, 2, 3), RS_SHAPE(2, 4, 1, 4),
    // small tiles (round 4): a batch of 32 ... 200 reads leaves the late layers a few dozen tiles of the shapes above
    RS_SHAPE_D(8, 1, 1, 2), RS_SHAPE(8, 1, 1, 3), RS_SHAPE(8, 1, 1, 4), RS_SHAPE_D(4, 2, 1, 2), RS_SHAPE(4, 2, 1, 3),
    RS_SHAPE(4, 2, 1, 4), RS_SHAPE_D(2, 4, 1, 2), RS_SHAPE_D(2, 4, 1, 1),
    // four-wave workgroups (round 5): one wave per SIMD, for launches of fewer tiles than CUs
    RS_SHAPE_4(4, 1, 1, 1), RS_SHAPE_4(2, 2, 1, 1), RS_SHAPE_4(4, 1, 1, 2), RS_SHAPE_4(2, 2, 1, 2), RS_SHAPE_4(1, 4, 1, 2),
    RS_SHAPE_4(4, 1, 1, 3), RS_SHAPE_4(2, 2, 1, 3),
};
#undef RS_SHAPE
#undef RS_SHAPE_D
#undef RS_SHAPE_4
constexpr int kNumShapes = sizeof(kShapes) / sizeof(kShapes[0]);

size_t lds_bytes(const Shape& s, int kc) {
    const int bmp = s.wm * 16 * s.mt, bn = s.wn * 16 * s.nt;
    return 2 * (size_t)(2 * (bmp + 1) + 4 * bn) * (kc + 2) * sizeof(float);
}

// Tile shape for a layer launch.  Model: one persistent workgroup per CU; time = rounds x (MFMA
// issue of a tile + per-item staging and barrier + per-tile epilogue), in SIMD cycles.  Full launches (at least one tile
// per CU), eight-wave shapes; calibrated at B = 512.
double tile_cost(const Shape& s, int kc, int nch) {
    if (lds_bytes(s, kc) > 160 * 1024 || s.wm * s.wn != 8) return -1.0;
    const int bmp = s.wm * 16 * s.mt, bnt = s.wn * s.nt;
    const double slots = kc;                                    // 4 components x kc / 4 k-steps
    const double staged = ((2.0 * bmp + 2) + 4.0 * bnt * 16) * kc * 4.0;   // bytes per item
    const double item = slots * (2.0 * s.mt * s.nt * 32.0 + 4.0 * (s.mt + s.nt)) + 900.0 + 0.06 * staged;
    return nch * item + 1500.0 + 60.0 * s.mt * s.nt;
}

// THIN launches (fewer tiles than CUs): least-squares fit over tools/shape_sweep.py at 16 ... 512 reads, layers 4 and 11,
// every shape (see conv_wino4.hip: the MFMAs of a SIMD's waves add up, the rest of a slot does not depend on the waves per
// SIMD, staged bytes cost more the more CUs stream).  per_cu: workgroups of a four-wave shape resident on one CU.
constexpr double kThinLaunch = RS_WINO_THIN_LAUNCH;
double thin_tile_cost(const Shape& s, int kc, int nch, int per_cu, double fill) {
    if (lds_bytes(s, kc) * per_cu > 160 * 1024) return -1.0;
    const int bmp = s.wm * 16 * s.mt, bnt = s.wn * s.nt;
    const double slots = kc;
    const double staged = ((2.0 * bmp + 2) + 4.0 * bnt * 16) * kc * 4.0 * per_cu;
    const double wps = s.wm * s.wn * per_cu / 4.0;
    const double item = slots * (wps * s.mt * s.nt * 32.0 + RS_WINO_THIN_SLOT + RS_WINO_THIN_SLOT_MN * (s.mt + s.nt)) + RS_WINO_THIN_ITEM +
                        staged * (RS_WINO_THIN_BYTE + RS_WINO_THIN_BYTE_FILL * fill);
    return nch * item + RS_WINO_THIN_TILE + RS_WINO_THIN_TILE_MN * s.mt * s.nt;
}

// best shape for a launch over `rows_out` pooled rows: the B = 512 calibration over the eight-wave shapes; when that launch
// leaves CUs idle (and thin is allowed), the thin-launch fit over every shape, four-wave ones at one or two per CU.
const Shape* choose_shape(int64_t rows_out, int n16, int kc, int nch, int num_cu, double* cost_out, int* per_cu_out = nullptr,
                          bool allow_thin = true, bool* thin_out = nullptr) {
    const Shape* best = nullptr;
    double best_cost = 1e300;
    int best_per_cu = 1;
    int64_t best_tiles = 0;
    for (int k = 0; k < kNumShapes; ++k) {
        const Shape& s = kShapes[k];
        const double tile = tile_cost(s, kc, nch);
        if (tile < 0) continue;
        const int bmp = s.wm * 16 * s.mt, bnt = s.wn * s.nt;
        const int64_t tiles = ((rows_out + bmp - 1) / bmp) * ((n16 + bnt - 1) / bnt);
        const double cost = (double)((tiles + num_cu - 1) / num_cu) * tile;
        if (cost < best_cost) {
            best_cost = cost;
            best = &s;
            best_tiles = tiles;
        }
    }
    const bool thin = allow_thin && best && best_tiles < num_cu;
    if (thin) {
        best_cost = 1e300;
        for (int k = 0; k < kNumShapes; ++k) {
            const Shape& s = kShapes[k];
            const int bmp = s.wm * 16 * s.mt, bnt = s.wn * s.nt;
            const int64_t tiles = ((rows_out + bmp - 1) / bmp) * ((n16 + bnt - 1) / bnt);
            const double fill = std::min(1.0, (double)tiles / num_cu);
            for (int per_cu = 1; per_cu <= (s.wm * s.wn == 4 ? 2 : 1); ++per_cu) {
                const double tile = thin_tile_cost(s, kc, nch, per_cu, fill);
                if (tile < 0) continue;
                const int64_t slots_ = (int64_t)num_cu * per_cu;
                const double cost = (double)((tiles + slots_ - 1) / slots_) * tile;
                if (cost < best_cost) {
                    best_cost = cost;
                    best = &s;
                    best_per_cu = per_cu;
                }
            }
        }
    }
    if (cost_out) *cost_out = best_cost;
    if (per_cu_out) *per_cu_out = best_per_cu;
    if (thin_out) *thin_out = thin;
    return best;
}

}  // namespace

int conv_wino_max_bn() { return 256; }
// planner's estimate (SIMD cycles) of one launch with the best tile shape (compared with the small-batch kernel's in api.hip)
double conv_wino_plan_cost(int64_t rows_out, int n16, int kc, int nch, int num_cu) {
    double cost = 1e300;
    choose_shape(rows_out, n16, kc, nch, num_cu, &cost, nullptr, false);
    return cost;
}
// estimate of one launch INCLUDING its launch cost where the thin-launch fit applies (*thin_out): for the choice between
// this kernel and conv_small_f32 (whose fit includes its launch as well)
double conv_wino_launch_cost(int64_t rows_out, int n16, int kc, int nch, int num_cu, bool* thin_out) {
    double cost = 1e300;
    bool thin = false;
    choose_shape(rows_out, n16, kc, nch, num_cu, &cost, nullptr, true, &thin);
    if (thin_out) *thin_out = thin;
    return thin ? cost + kThinLaunch : cost;
}
int conv_wino_num_shapes() { return kNumShapes; }
bool conv_wino_shape_ok(const ConvLayerDev& L, int k) {
    return k >= 0 && k < kNumShapes && (L.plan.kc == 16 || L.plan.kc == 20 || L.plan.kc == 24) &&
           lds_bytes(kShapes[k], L.plan.kc) <= 160 * 1024;
}

// layer 0 can be folded into this layer's staging when the layer is the shipped net's layer 1
// (20 input channels = one 20-channel chunk, <= 32 outputs) and a tile spans at most two reads
bool conv_wino_can_fuse0(const ConvLayerDev& L, int P_in) {
    return L.cp_in == 20 && L.plan.kc == 20 && L.plan.nch == 1 && L.c_out <= kFusedBN && P_in >= 2 * kFusedBMP + 2 &&
           !L.hooks->no_fuse0;
}

int launch_conv_wino(const ConvLayerDev& L, const float* d_x, float* d_y, const int32_t* d_len, int B, int P_in,
                     int layer_index, int num_cu, const float* d_zero, int check_dead, hipStream_t st, int* bm_out,
                     int* bn_out, const float* fuse_xs, const float* fuse_w0) {
    const ConvPlan& p = L.plan;
    if (p.kc != 16 && p.kc != 20 && p.kc != 24) {
        set_error("conv_wino: unsupported channel chunk %d", p.kc);
        return RS_ERR_ARG;
    }
    const int64_t rows64 = (int64_t)B * P_in;
    if (rows64 > 0x7fffffff) {
        set_error("conv_wino: batch too large (%lld rows)", (long long)rows64);
        return RS_ERR_ARG;
    }
    const int n16 = round_up(L.c_out, 16) / 16;
    double single_cost = 0.0;
    int per_cu = 1;
    const Shape* s = choose_shape(rows64 / 2, n16, p.kc, p.nch, num_cu, &single_cost, &per_cu);
    bool pinned = false;                                          // a forced, fused or tuned shape runs as one launch
    if (const char* force = L.hooks->force_wino; *force) {        // tuning aid: "layer:wm,wn,mt,nt;..."
        int l, wm, wn, mt, nt;
        for (const char* q = force; q && *q; q = strchr(q, ';') ? strchr(q, ';') + 1 : nullptr)
            if (sscanf(q, "%d:%d,%d,%d,%d", &l, &wm, &wn, &mt, &nt) == 5 && l == layer_index)
                for (int k = 0; k < kNumShapes; ++k)
                    if (kShapes[k].wm == wm && kShapes[k].wn == wn && kShapes[k].mt == mt && kShapes[k].nt == nt &&
                        lds_bytes(kShapes[k], p.kc) <= 160 * 1024) {
                        s = &kShapes[k];
                        pinned = true;
                        per_cu = 1;
                    }
    }
    const bool fused = fuse_xs != nullptr;
    if (fused) {
        pinned = true;
        per_cu = 1;
        if (!conv_wino_can_fuse0(L, P_in) || !fuse_w0) {
            set_error("conv_wino: layer cannot take the fused layer-0 path");
            return RS_ERR_ARG;
        }
        for (int k = 0; k < kNumShapes; ++k)
            if (kShapes[k].wm == 8 && kShapes[k].wn == 1 && kShapes[k].mt == 2 && kShapes[k].nt == 2) s = &kShapes[k];
    }
    if (const int k = tuned_shape(L, rows64); k >= 0 && conv_wino_shape_ok(L, k)) {
        s = &kShapes[k];
        pinned = true;
        per_cu = 1;
    }
    if (!s) {
        set_error("conv_wino: no tile shape fits (kc=%d)", p.kc);
        return RS_ERR_ARG;
    }
    WinoArgs a;
    a.xs = fuse_xs;
    a.xs_bytes = (unsigned)std::min<int64_t>(rows64 * 2 * 4, 0x7fffffffLL);      // B * P0 floats (P0 = 2 * P_in)
    a.w0 = fuse_w0;
    a.n_reads = B;
    if (fused && rows64 * 2 * 4 >= 0x80000000LL) {
        set_error("conv_wino: signal buffer exceeds the 2 GiB buffer-load window, split the batch");
        return RS_ERR_ARG;
    }
    a.x = d_x;
    a.w = static_cast<const float*>(L.d_w);
    a.bias = L.d_bias;
    a.y = d_y;
    a.len = d_len;
    const int64_t xb = rows64 * L.cp_in * 4, wb = (int64_t)p.n_alloc * p.nch * 4 * p.kc * 4;
    if ((!fused && xb >= 0x80000000LL) || wb >= 0x80000000LL) {
        set_error("conv_wino: activation buffer of %lld bytes exceeds the 2 GiB buffer-load window, split the batch",
                  (long long)xb);
        return RS_ERR_ARG;
    }
    a.x_bytes = fused ? 0u : (unsigned)xb;
    a.w_bytes = (unsigned)wb;
    a.y_bytes = (unsigned)(rows64 / 2 * L.cp_out * 4);          // < x_bytes * 2 ... checked below
    a.y_row_bytes = (unsigned)L.cp_out * 4u;
    a.len_bytes = (unsigned)B * 4u;
    a.bias_bytes = (unsigned)p.n_alloc * 4u;
    if (rows64 / 2 * L.cp_out * 4 >= 0x80000000LL) {
        set_error("conv_wino: output buffer exceeds the 2 GiB buffer-store window, split the batch");
        return RS_ERR_ARG;
    }
    a.rows_in = (int)rows64;
    a.rows_out = (int)(rows64 / 2);
    a.P_out = P_in / 2;
    a.inv_P_out = 1.0f / (float)a.P_out;
    a.cp_in = L.cp_in;
    a.cp_out = L.cp_out;
    a.nch = p.nch;
    a.shift_out = layer_index + 1;
    // one launch over the row tiles [m_base, m_base + n_mtiles x BMP) of shape sh
    auto launch_part = [&](const Shape& sh, int m_base, int n_mtiles, int wg_per_cu) -> int {
        const int BN_ = sh.wn * 16 * sh.nt;
        const int ki = p.kc == 16 ? 0 : p.kc == 20 ? 1 : 2;
        KernelFn fn = fused ? kFusedL1 : sh.fn[ki];
        if (KernelFn d = sh.deep[ki]; d && !fused && !L.hooks->no_deep_staging) fn = d;
        RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
        const int n_ntiles = (n16 * 16 + BN_ - 1) / BN_;
        const int64_t tiles = (int64_t)n_mtiles * n_ntiles;
        const int slots_ = num_cu * wg_per_cu;
        const unsigned grid = (unsigned)std::min<int64_t>(tiles, slots_);
        a.walk = plan_walk(n_mtiles, n_ntiles, grid, slots_, 2.0 * sh.wm * 16 * sh.mt, 4.0 * BN_, check_dead,
                           !L.hooks->no_rect_order);
        a.walk.m_base = m_base;
        hipLaunchKernelGGL(fn, dim3(grid), dim3(fused ? 512 : 64 * sh.wm * sh.wn), lds_bytes(sh, p.kc), st, a);
        RS_HIP(hipGetLastError());
        return RS_OK;
    };
    TailSplit split;
    if (!pinned && !L.hooks->no_tail_split)
        split = plan_tail_split(
            kNumShapes, (int64_t)a.rows_out, num_cu, single_cost, [&](int k) { return tile_cost(kShapes[k], p.kc, p.nch); },
            [&](int k) { return kShapes[k].wm * 16 * kShapes[k].mt; },
            [&](int k) { return (n16 + kShapes[k].wn * kShapes[k].nt - 1) / (kShapes[k].wn * kShapes[k].nt); },
            [&](int64_t r, double* c) {
                // priced with the full-launch calibration like the head (one scale); the tail itself runs the shape the
                // thin-launch fit picks for its rows
                const Shape* t = choose_shape(r, n16, p.kc, p.nch, num_cu, c, nullptr, false);
                return t ? (int)(t - kShapes) : -1;
            });
    int BMP, BN;
    if (split.head_shape >= 0) {
        const Shape& h = kShapes[split.head_shape];
        BMP = h.wm * 16 * h.mt;
        BN = h.wn * 16 * h.nt;
        const int m_base = split.head_mtiles * BMP;
        int tail_per_cu = 1;
        const Shape* tp = choose_shape(a.rows_out - m_base, n16, p.kc, p.nch, num_cu, nullptr, &tail_per_cu);
        const Shape& t = tp ? *tp : kShapes[split.tail_shape];
        if (L.hooks->tail_debug)
            fprintf(stderr, "[tail-split] layer %d: head %dx%dx%dx%d x %d row tiles, tail %dx%dx%dx%d (%d per CU); planned %.0f vs %.0f cycles\n",
                    layer_index, h.wm, h.wn, h.mt, h.nt, split.head_mtiles, t.wm, t.wn, t.mt, t.nt, tail_per_cu, split.cost, single_cost);
        int rc = launch_part(h, 0, split.head_mtiles, 1);
        if (rc != RS_OK) return rc;
        const int tbm = t.wm * 16 * t.mt;
        rc = launch_part(t, m_base, (a.rows_out - m_base + tbm - 1) / tbm, tail_per_cu);
        if (rc != RS_OK) return rc;
    } else {
        BMP = s->wm * 16 * s->mt;
        BN = s->wn * 16 * s->nt;
        const int rc = launch_part(*s, 0, (a.rows_out + BMP - 1) / BMP, per_cu);
        if (rc != RS_OK) return rc;
    }
    if (bm_out) *bm_out = 2 * BMP;          // reported in conv rows, like the direct kernels
    if (bn_out) *bn_out = BN;
    return RS_OK;
}

}  // namespace rs
