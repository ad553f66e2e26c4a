// The F(2,3) Winograd conv kernel template (see conv_wino.hip for the description): included by conv_wino.hip (the eight-wave
// shapes of full launches, the layer-0 fold, planner and launcher) and conv_wino_thin.hip (the thin-launch forms: four-wave
// workgroups and loads one item ahead) - two translation units, so that the two halves of the instantiations compile side by side.
#pragma once
#include "common.hpp"
#include "tile_walk.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <utility>

namespace rs {

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

}  // namespace
}  // namespace rs
