// K3w4 (fp32, Winograd F(4,3)): ConvNet block i >= 1 for the wide late layers.
//
//   Conv1d(C_in -> C_out, k=3, stride 1, zero 'same' padding, bias) -> ReLU -> MaxPool1d(2,2)
//   (riser/nets/cnn.py:52-65, depth 1)
//
// F(4,3) produces FOUR consecutive conv outputs y[4G .. 4G+3] (two pooling pairs) from the six inputs
// d0..d5 = x[4G-1 .. 4G+4] with 6 multiplications per input channel instead of 12 (F(2,3): 8), i.e. half
// the matrix-pipe work of the direct lowering:
//     V = B^T d    B^T = [ 4  0 -5  0  1  0 ]      U = G g    G = [ 1/4    0     0  ]      y = A^T (U . V)
//                        [ 0 -4 -4  1  1  0 ]                     [-1/6  -1/6  -1/6 ]      A^T = [ 1 1  1 1  1 0 ]
//                        [ 0  4 -4 -1  1  0 ]                     [-1/6   1/6  -1/6 ]            [ 0 1 -1 2 -2 0 ]
//                        [ 0 -2 -1  2  1  0 ]                     [ 1/24  1/12  1/6 ]            [ 0 1  1 4  4 0 ]
//                        [ 0  2 -1 -2  1  0 ]                     [ 1/24 -1/12  1/6 ]            [ 0 1 -1 8 -8 1 ]
//                        [ 0  4  0 -5  0  1 ]                     [  0     0     1  ]
// (Lavin & Gray's transform set).  The layer is SIX GEMMs M_j[G][n] = sum_c V_j[G][c] * U_j[n][c] over
// groups G of four input rows; the epilogue forms y0..y3, the two pooled values max(y0,y1), max(y2,y3),
// + bias, ReLU, length mask, and stores two 16-byte pieces per lane and sub-tile.
// fp32 throughout; measured on the CPU in numpy the probabilities stay within 9e-6 of an fp64 evaluation
// (direct fp32: 6e-6), see tests/test_oracle_golden.py and tests/test_gpu_wino.py.
//
// Structure, data layout, staging (buffer loads, pass-major unit map), tile walk and MFMA orientation are
// those of conv_wino.hip; what differs: the input slab is split into FOUR planes by (row mod 4) so that the
// six inputs of group G are rows G, G+1 of the planes (unit lane stride, pitch KC + 2: conflict-free
// ds_read_b32), the weight slab has six components, a slot = (k-step, component) carries MT * NT MFMAs
// with MT * NT <= 5 (6 x 4 accumulator registers per 16 x 16 sub-tile), and LDS capacity limits the channel
// chunk to 16 or 20.  Used for the layers where the matrix pipe is the bound (chosen in rs_model_create).
#include "common.hpp"
#include "tile_walk.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <utility>

#ifndef RS_UF_AHEAD
#define RS_UF_AHEAD 1
#endif
#ifndef RS_UF_AHEAD_THIN          // slots of look-ahead x MFMAs per slot for the one- and two-MFMA slots of the thin shapes
#define RS_UF_AHEAD_THIN 4
#endif
#ifndef RS_RD_THIN
#define RS_RD_THIN 0
#endif
#ifndef RS_XF_THIN
#define RS_XF_THIN 5
#endif
#ifndef RS_W4_DIST
#define RS_W4_DIST 6
#endif
#ifndef RS_W4_PRIO
#define RS_W4_PRIO 0
#endif
#ifndef RS_W4_STORE_ALWAYS
#define RS_W4_STORE_ALWAYS 1
#endif

namespace rs {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

struct Wino4Args {
    const float* x;
    const float* w;        // packed [n_alloc][nch][6][kc] (U0..U5), zero rows beyond c_out
    const float* bias;     // [n_alloc]
    float* y;
    const int32_t* len;
    unsigned x_bytes, w_bytes, y_bytes, y_row_bytes, len_bytes, bias_bytes;
    int rows_in;           // B * P_in
    int rows_out;          // B * P_out (pooled rows)
    int n_groups;          // ceil(rows_in / 4)
    int P_out;
    float inv_P_out;
    int cp_in, cp_out;
    int nch;
    int shift_out;
    WalkArgs walk;         // tile grid, order and dead-tile flag (tile_walk.hpp)
};

// DEEP: the staging loads of an item are issued ONE ITEM earlier (they stay in registers across the barrier and go to LDS
// during the next item): a thin launch (a few dozen tiles of the smallest shapes) has ~1.5 k cycles of MFMAs per item, far
// less than the trip to L2 / HBM that the six-slot distance of the default schedule covers.  Same MFMA sequence per
// accumulator, so the bits do not change.
template <int WM, int WN, int MT, int NT, int KCT, bool DEEP = false>
__global__ __launch_bounds__(512) void conv_wino4_kernel(const Wino4Args a) {   // 512 for the four-wave shapes too: a bound of 256 makes
                                                                                  // the compiler keep the accumulators in AGPRs, with
                                                                                  // a copy in and out per item
    // 8 waves (2 per SIMD), or 4 (1 per SIMD) for launches with fewer tiles than CUs: the VALU work of a wave (input
    // transform, staging, walk) and the f32-input MFMAs of its SIMD neighbour do not overlap, so a thin launch is faster
    // with its waves spread over twice the CUs
    static_assert(WM * WN == 8 || WM * WN == 4, "8 or 4 waves per workgroup");
    static_assert(KCT == 16 || KCT == 20, "channel chunk");
    constexpr int kThreads = 64 * WM * WN;
    constexpr int NC = 6;                               // Winograd components
    constexpr int BG = WM * 16 * MT;                    // groups (of 4 conv rows = 2 pooled rows) per tile
    constexpr int BN = WN * 16 * NT;
    constexpr int S = KCT + 2;
    constexpr int KQ = KCT / 4;
    constexpr int PL = (BG + 1) * S;                    // one plane (input rows = p mod 4)
    constexpr int A_ELEMS = 4 * PL;
    constexpr int BUF = A_ELEMS + NC * BN * S;
    constexpr int RPT = (kThreads / KQ) & ~3;           // slab rows per staging pass, a multiple of 4
    constexpr int A_ROWS = 4 * BG + 2;
    constexpr int A_PER = (A_ROWS + RPT - 1) / RPT;
    constexpr int NPP = kThreads / (NC * KQ);
    constexpr int B_PER = (BN + NPP - 1) / NPP;
    constexpr unsigned kOob = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int r = lane & 15, kq = lane >> 4;

    // ---- staging map (see conv_wino.hip) -----------------------------------------------------------------
    // slab row s (global input row 4*m0g - 1 + s) lives in plane s & 3 at index s >> 2; a thread's rows
    // a_row + u * RPT keep their plane because RPT is a multiple of 4
    const int a_row = tid / KQ, a_c4 = tid - a_row * KQ;
    const bool a_act = a_row < RPT;
    const int b_n = tid / (NC * KQ), b_rem = tid - b_n * (NC * KQ);
    const bool b_act = b_n < NPP;
    const int a_st = (a_row & 3) * PL + (a_row >> 2) * S + 4 * a_c4;                     // + u * (RPT/4) * S
    const int b_st = A_ELEMS + ((b_rem / KQ) * BN + b_n) * S + 4 * (b_rem % KQ);         // + u * NPP * S
    const unsigned a_tb = (unsigned)(a_row * a.cp_in + 4 * a_c4) * 4u;
    const unsigned b_tb = (unsigned)(b_n * a.nch * NC * KCT + 4 * b_rem) * 4u;
    const unsigned a_step = (unsigned)(RPT * a.cp_in) * 4u;
    const unsigned b_step = (unsigned)(NPP * a.nch * NC * KCT) * 4u;
    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_len =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.len), 0, a.len_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_bias =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.bias_bytes, 0x00020000);
    const const_len_ptr clen = as_const_len(a.len);

    u32x4 ra[A_PER], rb[B_PER];
    unsigned a_ib = kOob, b_ib = kOob;
    auto item_offsets = [&](int m0g, int n0, int c, bool live) {
        const bool a_ok = live && a_act && c * KCT + 4 * a_c4 < a.cp_in;
        a_ib = a_ok ? a_tb + (unsigned)((4 * m0g - 1) * a.cp_in + c * KCT) * 4u : kOob;   // row -1 wraps out of range
        b_ib = (live && b_act) ? b_tb + (unsigned)((n0 * a.nch + c) * NC * KCT) * 4u : kOob;
#ifdef RS_ABL_NOLOAD                                    // timing experiment only: every staging load out of range
        a_ib = kOob;
        b_ib = kOob;
#endif
    };
    auto load_unit = [&](auto U) {
        constexpr int u = decltype(U)::value;
        if constexpr (u < A_PER) {
            unsigned off = a_ib + (unsigned)u * a_step;
            if constexpr ((u + 1) * RPT > A_ROWS) off = (a_row + u * RPT < A_ROWS) ? off : kOob;
            ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
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
            if (act) {
                uint2* d = reinterpret_cast<uint2*>(buf + a_st + u * (RPT / 4) * S);
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

    // ---- tile walk (tile_walk.hpp), tiles counted in groups -----------------------------------------------
    const int tiles = a.walk.q_total;
    auto tile_origin = [&](int q, int& tm0, int& tn0) -> bool {
        int mi, nt_;
        const bool ok = walk_tile(a.walk, q, mi, nt_);
        tm0 = a.walk.m_base + mi * BG;
        tn0 = nt_ * BN;
        return ok;
    };
    TileWalk walk;
    auto order_index = [&]() { return walk.next_index(a.walk); };
    auto next_live = [&]() {
        int q = order_index();
        while (q < tiles) {
            int tm0, tn0;
            const bool valid = tile_origin(q, tm0, tn0);
            if (valid) {
                if (!a.walk.check_dead) break;
                const int pr0 = 2 * tm0;                                  // first pooled row of the tile
                const int b = pr0 / a.P_out;
                const int t0 = pr0 - b * a.P_out;
                if (!(b < a.rows_out / a.P_out && t0 + 2 * BG <= a.P_out && t0 >= (clen[b] >> a.shift_out))) break;
                const int pieces_per_row = BN / 4;
                for (int f = threadIdx.x; f < 2 * BG * pieces_per_row; f += blockDim.x) {
                    const int rr = f / pieces_per_row, cc = (f - rr * pieces_per_row) * 4;
                    const int prow = pr0 + rr, col = tn0 + cc;
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

    f32x4 acc[MT][NT][NC];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < NC; ++q) acc[i][j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

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

    const int a_rd = (wm * 16 * MT + r) * S + kq;                  // + (k & 3) * PL + (i * 16 + (k >> 2)) * S + c0
    const int b_rd = A_ELEMS + (wn * 16 * NT + r) * S + kq;        // + (comp * BN + j * 16) * S + c0

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

        constexpr int NSLOTS = NC * KQ;                // slot = (k-step, component): MT * NT MFMAs
        constexpr int UNITS = A_PER + B_PER;
        constexpr int DIST = RS_W4_DIST;               // slots between a unit's load and its LDS write
        constexpr int SPAN = NSLOTS - DIST;
        float dr[MT][6];                               // raw inputs d0..d5 of the lane's groups (next k-step)
        // a slot of the thin shapes is one or two MFMAs (32 / 64 cycles): an LDS read issued one slot ahead is not back
        // in time, so their weight fragments are read four (two) slots ahead, the raw rows at the first slot of a
        // k-step and transformed behind its last
        constexpr bool THIN = MT * NT <= 2;
        constexpr int AH = THIN ? (RS_UF_AHEAD_THIN + MT * NT - 1) / (MT * NT) : RS_UF_AHEAD;   // slots of look-ahead of the weight-fragment reads
        constexpr int RD = THIN ? RS_RD_THIN : 1, XF0 = THIN ? RS_XF_THIN : 3;
        float uf[AH + 1][NT];                          // weight fragments, ring over the slots in flight
        float v[2][MT][NC];                            // transformed inputs: this k-step / the next one
        auto xform = [&](float (&o)[NC], const float (&d)[6]) {   // V = B^T d
#ifdef RS_ABL_NOXFORM                                    // timing experiment only
            o[0] = d[0], o[1] = d[1], o[2] = d[2], o[3] = d[3], o[4] = d[4], o[5] = d[5];
#else
            const float p = fmaf(-4.0f, d[2], d[4]), q = fmaf(-4.0f, d[1], d[3]);
            const float s2 = d[4] - d[2], t2 = d[3] - d[1];
            o[0] = fmaf(4.0f, d[0], fmaf(-5.0f, d[2], d[4]));
            o[1] = p + q;
            o[2] = p - q;
            o[3] = fmaf(2.0f, t2, s2);
            o[4] = fmaf(-2.0f, t2, s2);
            o[5] = fmaf(4.0f, d[1], fmaf(-5.0f, d[3], d[5]));
#endif
        };
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int k = 0; k < 6; ++k) dr[i][k] = Ab[(k & 3) * PL + (i * 16 + (k >> 2)) * S];
        static_for<AH>([&](auto SL) {
            constexpr int sl = decltype(SL)::value;
#pragma unroll
            for (int j = 0; j < NT; ++j) uf[sl][j] = Bb[((sl % NC) * BN + j * 16) * S + 4 * (sl / NC)];
        });
#pragma unroll
        for (int i = 0; i < MT; ++i) xform(v[0][i], dr[i]);      // first k-step of the item: exposed once
#ifdef RS_ABL_NOFRAG                                    // timing experiment only: no fragment reads after the item's first
#pragma unroll
        for (int j = 0; j < NT; ++j) uf[AH][j] = uf[0][j];
#define RS_FRAG_A(dst, expr) asm volatile("" : "+v"(dst))
#define RS_FRAG_B(dst, expr) asm volatile("" : "+v"(dst))
#else
#define RS_FRAG_A(dst, expr) dst = (expr)
#define RS_FRAG_B(dst, expr) dst = (expr)
#endif
        static_for<NSLOTS>([&](auto SL) {
            constexpr int sl = decltype(SL)::value;
            constexpr int st = sl / NC, comp = sl % NC;
            // the raw rows of the next k-step are read in slot 1 and transformed in slots 3 (+ 4 for MT = 2),
            // one VALU group behind each MFMA, so that the transform hides in the MFMA shadow
            constexpr bool rd_next = comp == RD && st + 1 < KQ;
            constexpr bool xf_next = comp >= XF0 && comp - XF0 < MT && st + 1 < KQ;
            if constexpr (rd_next) {
                constexpr int c0 = 4 * (st + 1);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int k = 0; k < 6; ++k) RS_FRAG_A(dr[i][k], Ab[(k & 3) * PL + (i * 16 + (k >> 2)) * S + c0]);
            }
            if constexpr (xf_next) xform(v[(st + 1) & 1][comp - XF0], dr[comp - XF0]);
            if constexpr (sl + AH < NSLOTS) {
                constexpr int nst = (sl + AH) / NC, ncomp = (sl + AH) % NC;
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    RS_FRAG_B(uf[(sl + AH) % (AH + 1)][j], Bb[(ncomp * BN + j * 16) * S + 4 * nst]);
            }
#if RS_W4_PRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j][comp] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[sl % (AH + 1)][j], v[st & 1][i][comp],
                                                                           acc[i][j][comp], 0, 0, 0);
#if RS_W4_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            {
                constexpr int n_mf = MT * NT;
                constexpr int n_rd = (sl + AH < NSLOTS ? NT : 0) + (rd_next ? 3 * MT : 0);
                constexpr int n_va = xf_next ? 12 : 0;
                static_for<n_mf>([&](auto M) {
                    constexpr int m = decltype(M)::value;
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr (m < n_rd) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    constexpr int va0 = n_va * m / n_mf, va1 = n_va * (m + 1) / n_mf;
                    if constexpr (va1 > va0) __builtin_amdgcn_sched_group_barrier(0x002, va1 - va0, 0);
                });
            }
            if constexpr (xf_next) {                   // pin the transformed values to this slot (they are pure
                float(&o)[NC] = v[(st + 1) & 1][comp - XF0];   // functions of dr: the compiler would sink them to their use)
                asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]));
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<UNITS>([&](auto U) {
                constexpr int u = decltype(U)::value;
                if constexpr (DEEP) {
                    // the unit loaded during the previous item goes to the next item's buffer, and its registers
                    // leave again for the item after that
                    if constexpr ((u * NSLOTS) / UNITS == sl) {
                        store_unit(U, nbuf);
                        load_unit(U);
                    }
                } else if constexpr ((u * SPAN) / UNITS == sl)
                    load_unit(U);
                if constexpr (!DEEP && (u * SPAN) / UNITS + DIST == sl) {
#if RS_W4_STORE_ALWAYS                                  // unconditional: after the last item the writes land in the idle buffer
                                                        // (nobody reads it any more); saves a branch per unit: -1.2 % (clock-normalised A/B)
                    store_unit(U, nbuf);
#else
                    if (has_next) store_unit(U, nbuf);
#endif
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });

        if (c == a.nch - 1) {
            // ---- epilogue: y = A^T m, two pooled rows per group, bias + ReLU + length mask, 16-byte stores -----
            const int pr0 = 2 * m0;
            const int b0 = pr0 / a.P_out;
            const int p0 = pr0 - b0 * a.P_out;
            unsigned rowoff_[MT][2];
            bool valid_[MT][2];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int loc = 2 * ((wm * MT + i) * 16 + r) + h;
                    const int t = p0 + loc;
                    const int e = (int)(((float)t + 0.5f) * a.inv_P_out);      // t < P_out + 2 * BG < 2^16: exact
                    const int pin = t - e * a.P_out;
                    const unsigned lv = __builtin_amdgcn_raw_buffer_load_b32(rs_len, (unsigned)(b0 + e) * 4u, 0, 0);
                    rowoff_[i][h] = (unsigned)(pr0 + loc) * a.y_row_bytes;    // rows past the end: out of range
                    valid_[i][h] = pin < (int)(lv >> a.shift_out);
                }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = n0 + (wn * NT + j) * 16 + 4 * kq;
                const unsigned coloff = col < a.cp_out ? (unsigned)col * 4u : kOob;
                const f32x4 bi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bias, (unsigned)col * 4u, 0, 0));
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    f32x4 o0, o1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float m0_ = acc[i][j][0][q], m1 = acc[i][j][1][q], m2 = acc[i][j][2][q],
                                    m3 = acc[i][j][3][q], m4 = acc[i][j][4][q], m5 = acc[i][j][5][q];
                        const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                        const float y0 = (m0_ + s12) + s34;
                        const float y1 = fmaf(2.0f, d34, d12);
                        const float y2 = fmaf(4.0f, s34, s12);
                        const float y3 = fmaf(8.0f, d34, d12) + m5;
                        o0[q] = valid_[i][0] ? fmaxf(fmaxf(y0, y1) + bi[q], 0.0f) : 0.0f;
                        o1[q] = valid_[i][1] ? fmaxf(fmaxf(y2, y3) + bi[q], 0.0f) : 0.0f;
                    }
#ifdef RS_ABL_NOSTORE
                    asm volatile("" ::"v"(o0[0]), "v"(o0[1]), "v"(o0[2]), "v"(o0[3]), "v"(o1[0]), "v"(o1[1]), "v"(o1[2]), "v"(o1[3]));
#else
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rs_y, rowoff_[i][0] + coloff, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), rs_y, rowoff_[i][1] + coloff, 0, 0);
#endif
#pragma unroll
                    for (int q = 0; q < NC; ++q) acc[i][j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
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

using KernelFn = void (*)(const Wino4Args);

struct Shape {
    int wm, wn, mt, nt;
    KernelFn fn[2];        // chunk = 16, 20
    KernelFn deep[2];      // the same tile with staging loads one item ahead (thin launches), or null
};

#define RS_SHAPE(WM, WN, MT, NT) \
    {WM, WN, MT, NT, {conv_wino4_kernel<WM, WN, MT, NT, 16>, conv_wino4_kernel<WM, WN, MT, NT, 20>}, {nullptr, nullptr}}
#define RS_SHAPE_D(WM, WN, MT, NT)                                                                     \
    {WM, WN, MT, NT, {conv_wino4_kernel<WM, WN, MT, NT, 16>, conv_wino4_kernel<WM, WN, MT, NT, 20>}, \
     {conv_wino4_kernel<WM, WN, MT, NT, 16, true>, conv_wino4_kernel<WM, WN, MT, NT, 20, true>}}
// four-wave shapes exist in the one-item-ahead form only
#define RS_SHAPE_4(WM, WN, MT, NT) \
    {WM, WN, MT, NT, {conv_wino4_kernel<WM, WN, MT, NT, 16, true>, conv_wino4_kernel<WM, WN, MT, NT, 20, true>}, {nullptr, nullptr}}
const Shape kShapes[] = {
    // 512 x 96 (round 6): 36 accumulator tiles = 249 / 253 registers, no scratch; its LDS fits with chunks of 16 only (157 KB), so
    // rs_model_create's chunk choice moves layers 7 and 9 of the shipped net (22 and 48 column groups: 4 and 8 tiles of six) from
    // chunks of 20 to 16 for it: layer 9 becomes ONE round of 256 tiles at 512 reads (-9 %), the fp32 step -1.3 %
    RS_SHAPE_D(8, 1, 1, 2), RS_SHAPE(8, 1, 1, 3), RS_SHAPE(8, 1, 1, 4), RS_SHAPE(8, 1, 1, 5), RS_SHAPE(8, 1, 1, 6),
    RS_SHAPE_D(4, 2, 1, 2), RS_SHAPE(4, 2, 1, 3), RS_SHAPE(4, 2, 1, 4), RS_SHAPE(4, 2, 2, 2),
    RS_SHAPE_D(2, 4, 1, 2), RS_SHAPE(2, 4, 1, 3), RS_SHAPE(2, 4, 1, 4), RS_SHAPE(2, 4, 2, 2),
    // small tiles (round 4): a batch of 32 ... 200 reads leaves the late layers a few dozen tiles of the shapes above
    RS_SHAPE_D(4, 2, 1, 1), RS_SHAPE_D(2, 4, 1, 1),
    // four-wave workgroups (round 5): one wave per SIMD, for launches of fewer tiles than CUs
    RS_SHAPE_4(4, 1, 1, 1), RS_SHAPE_4(2, 2, 1, 1), RS_SHAPE_4(1, 4, 1, 1),
    RS_SHAPE_4(4, 1, 1, 2), RS_SHAPE_4(2, 2, 1, 2), RS_SHAPE_4(1, 4, 1, 2),
    RS_SHAPE_4(4, 1, 1, 3), RS_SHAPE_4(2, 2, 1, 3),
};
#undef RS_SHAPE
#undef RS_SHAPE_D
#undef RS_SHAPE_4
constexpr int kNumShapes = sizeof(kShapes) / sizeof(kShapes[0]);

size_t lds_bytes(const Shape& s, int kc) {
    const int bg = s.wm * 16 * s.mt, bn = s.wn * 16 * s.nt;
    return 2 * (size_t)(4 * (bg + 1) + 6 * bn) * (kc + 2) * sizeof(float);
}

// Full launches (at least one tile per CU): per slot the MFMAs of the two waves of a SIMD, fragment reads, 1/6 of the k-step's
// input transform (12 VALU per MT, not hidden behind the MFMAs); per item: fixed cost, staging, and the activation slab's trip
// from L2 / Infinity Cache (the weight slab is an L2 hit).  Calibrated on tools/shape_sweep.py (B = 512); eight-wave shapes.
double tile_cost(const Shape& s, int kc, int nch) {
    if (lds_bytes(s, kc) > 160 * 1024 || s.wm * s.wn != 8) return -1.0;
    const int bg = s.wm * 16 * s.mt, bnt = s.wn * s.nt;
    const double slots = 6.0 * kc / 4.0;
    const double staged = ((4.0 * bg + 2) + 6.0 * bnt * 16) * kc * 4.0;
    const double xslab = (4.0 * bg + 2) * kc * 4.0;
    const double item = slots * (2.0 * s.mt * s.nt * 32.0 + 6.0 * (s.mt + s.nt) + 12.0 * s.mt) + 900.0 +
                        0.06 * staged + 0.02 * xslab;
    return nch * item + 1500.0 + 90.0 * s.mt * s.nt;
}

// THIN launches (fewer tiles than CUs: 16 ... 100 reads in the late layers): a least-squares fit over tools/shape_sweep.py at
// 16 ... 512 reads, every shape, layers 5 - 10 (4 % rms, 15 % worst).  Per slot the MFMAs of a SIMD's waves add up; the
// other instructions of a slot do not depend on the waves per SIMD; staged bytes cost more the more CUs stream (fill).
// per_cu: workgroups of a four-wave shape resident on one CU (two: their waves share the SIMDs like an eight-wave
// workgroup's).  kThinLaunch: launch + first / last tile effects of the same fit (for the comparison with conv_small_f32).
constexpr double kThinLaunch = 23500.0;
double thin_tile_cost(const Shape& s, int kc, int nch, int per_cu, double fill) {
    if (lds_bytes(s, kc) * per_cu > 160 * 1024) return -1.0;
    const int bg = s.wm * 16 * s.mt, bnt = s.wn * s.nt;
    const double slots = 6.0 * kc / 4.0;
    const double staged = ((4.0 * bg + 2) + 6.0 * bnt * 16) * kc * 4.0 * per_cu;
    const double wps = s.wm * s.wn * per_cu / 4.0;
    const double item = slots * (wps * s.mt * s.nt * 32.0 + 71.6 + 4.25 * (s.mt + s.nt)) - 920.0 + staged * (0.0062 + 0.0124 * fill);
    return nch * item - 467.0 + 554.0 * s.mt * s.nt;
}

// best shape for a launch over `groups` row units: the B = 512 calibration over the eight-wave shapes; when that launch
// leaves CUs idle (and thin is allowed), the thin-launch fit over every shape, four-wave ones at one or two per CU.
// *thin_out: which model *cost_out is in (their scales differ by up to 25 %: compare like with like).
const Shape* choose_shape(int64_t groups, int n16, int kc, int nch, int num_cu, double* cost_out, int* per_cu_out = nullptr,
                          bool allow_thin = true, bool* thin_out = nullptr) {
    const Shape* best = nullptr;
    double best_cost = 1e300;
    int best_per_cu = 1;
    int64_t best_tiles = 0;
    for (int k = 0; k < kNumShapes; ++k) {
        const Shape& s = kShapes[k];
        const double tile = tile_cost(s, kc, nch);
        if (tile < 0) continue;
        const int bg = s.wm * 16 * s.mt, bnt = s.wn * s.nt;
        const int64_t tiles = ((groups + bg - 1) / bg) * ((n16 + bnt - 1) / bnt);
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
            const int bg = s.wm * 16 * s.mt, bnt = s.wn * s.nt;
            const int64_t tiles = ((groups + bg - 1) / bg) * ((n16 + bnt - 1) / bnt);
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

int conv_wino4_max_bn() { return 256; }
int conv_wino4_num_shapes() { return kNumShapes; }
bool conv_wino4_shape_ok(const ConvLayerDev& L, int k) {
    return k >= 0 && k < kNumShapes && (L.plan.kc == 16 || L.plan.kc == 20) && lds_bytes(kShapes[k], L.plan.kc) <= 160 * 1024;
}

// planner's estimate (SIMD cycles) of one launch with the best tile shape for this chunk size: lets
// rs_model_create pick the channel chunk (which fixes the weight packing) with the tile shapes it enables in mind
double conv_wino4_plan_cost(int64_t groups, int n16, int kc, int nch, int num_cu) {
    double cost = 1e300;
    choose_shape(groups, n16, kc, nch, num_cu, &cost, nullptr, false);
    return cost;
}

// estimate of one launch INCLUDING its launch cost where the thin-launch fit applies (*thin_out), for the choice between
// this kernel and conv_small_f32 (whose fit includes its launch as well)
double conv_wino4_launch_cost(int64_t groups, int n16, int kc, int nch, int num_cu, bool* thin_out) {
    double cost = 1e300;
    bool thin = false;
    choose_shape(groups, n16, kc, nch, num_cu, &cost, nullptr, true, &thin);
    if (thin_out) *thin_out = thin;
    return thin ? cost + kThinLaunch : cost;
}

int launch_conv_wino4(const ConvLayerDev& L, const float* d_x, float* d_y, const int32_t* d_len, int B, int P_in,
                      int layer_index, int num_cu, int check_dead, hipStream_t st, int* bm_out, int* bn_out) {
    const ConvPlan& p = L.plan;
    if (p.kc != 16 && p.kc != 20) {
        set_error("conv_wino4: unsupported channel chunk %d", p.kc);
        return RS_ERR_ARG;
    }
    const int64_t rows64 = (int64_t)B * P_in;
    const int64_t xb = rows64 * L.cp_in * 4, wb = (int64_t)p.n_alloc * p.nch * 6 * p.kc * 4,
                  yb = rows64 / 2 * L.cp_out * 4;
    if (rows64 > 0x7fffffff || xb >= 0x80000000LL || wb >= 0x80000000LL || yb >= 0x80000000LL) {
        set_error("conv_wino4: batch too large for the 2 GiB buffer window, split it");
        return RS_ERR_ARG;
    }
    const int n16 = round_up(L.c_out, 16) / 16;
    const int64_t groups = (rows64 + 3) / 4;
    double single_cost = 0.0;
    int per_cu = 1;
    const Shape* s = choose_shape(groups, n16, p.kc, p.nch, num_cu, &single_cost, &per_cu);
    bool pinned = false;                                          // a forced or tuned shape runs as one launch
    if (const char* force = L.hooks->force_wino4; *force) {       // tuning aid: "layer:wm,wn,mt,nt;..."
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
    if (const int k = tuned_shape(L, rows64); k >= 0 && conv_wino4_shape_ok(L, k)) {
        s = &kShapes[k];
        pinned = true;
        per_cu = 1;
    }
    if (!s) {
        set_error("conv_wino4: no tile shape fits (kc=%d)", p.kc);
        return RS_ERR_ARG;
    }
    Wino4Args a;
    a.x = d_x;
    a.w = static_cast<const float*>(L.d_w);
    a.bias = L.d_bias;
    a.y = d_y;
    a.len = d_len;
    a.x_bytes = (unsigned)xb;
    a.w_bytes = (unsigned)wb;
    a.y_bytes = (unsigned)yb;
    a.y_row_bytes = (unsigned)L.cp_out * 4u;
    a.len_bytes = (unsigned)B * 4u;
    a.bias_bytes = (unsigned)p.n_alloc * 4u;
    a.rows_in = (int)rows64;
    a.rows_out = (int)(rows64 / 2);
    a.n_groups = (int)groups;
    a.P_out = P_in / 2;
    a.inv_P_out = 1.0f / (float)a.P_out;
    a.cp_in = L.cp_in;
    a.cp_out = L.cp_out;
    a.nch = p.nch;
    a.shift_out = layer_index + 1;
    // one launch over the row tiles [m_base, m_base + n_mtiles x BG) of shape sh
    auto launch_part = [&](const Shape& sh, int m_base, int n_mtiles, int wg_per_cu) -> int {
        const int BN = sh.wn * 16 * sh.nt;
        const int n_ntiles = (n16 * 16 + BN - 1) / BN;
        const int64_t tiles = (int64_t)n_mtiles * n_ntiles;
        const int slots_ = num_cu * wg_per_cu;
        const unsigned grid = (unsigned)std::min<int64_t>(tiles, slots_);
        a.walk = plan_walk(n_mtiles, n_ntiles, grid, slots_, 4.0 * sh.wm * 16 * sh.mt, 6.0 * BN, check_dead,
                           !L.hooks->no_rect_order);
        a.walk.m_base = m_base;
        KernelFn fn = sh.fn[p.kc == 16 ? 0 : 1];
        if (KernelFn d = sh.deep[p.kc == 16 ? 0 : 1]; d && !L.hooks->no_deep_staging) fn = d;
        RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
        hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * sh.wm * sh.wn), lds_bytes(sh, p.kc), st, a);
        RS_HIP(hipGetLastError());
        return RS_OK;
    };
    TailSplit split;
    if (!pinned && !L.hooks->no_tail_split)
        split = plan_tail_split(
            kNumShapes, groups, num_cu, single_cost, [&](int k) { return tile_cost(kShapes[k], p.kc, p.nch); },
            [&](int k) { return kShapes[k].wm * 16 * kShapes[k].mt; },
            [&](int k) { return (n16 + kShapes[k].wn * kShapes[k].nt - 1) / (kShapes[k].wn * kShapes[k].nt); },
            [&](int64_t g, double* c) {
                // priced with the full-launch calibration like the head (one scale); the tail itself runs the shape the
                // thin-launch fit picks for its rows (never slower than this one)
                const Shape* t = choose_shape(g, n16, p.kc, p.nch, num_cu, c, nullptr, false);
                return t ? (int)(t - kShapes) : -1;
            },
            L.hooks->tail_margin > 0 ? L.hooks->tail_margin : 0.97);
    int BG, BN;
    if (split.head_shape >= 0) {
        const Shape& h = kShapes[split.head_shape];
        BG = h.wm * 16 * h.mt;
        BN = h.wn * 16 * h.nt;
        const int m_base = split.head_mtiles * BG;
        int tail_per_cu = 1;
        const Shape* tp = choose_shape(groups - m_base, n16, p.kc, p.nch, num_cu, nullptr, &tail_per_cu);
        const Shape& t = tp ? *tp : kShapes[split.tail_shape];
        if (L.hooks->tail_debug)
            fprintf(stderr, "[tail-split] layer %d: head %dx%dx%dx%d x %d row tiles, tail %dx%dx%dx%d (%d per CU); planned %.0f vs %.0f cycles\n",
                    layer_index, h.wm, h.wn, h.mt, h.nt, split.head_mtiles, t.wm, t.wn, t.mt, t.nt, tail_per_cu, split.cost, single_cost);
        int rc = launch_part(h, 0, split.head_mtiles, 1);
        if (rc != RS_OK) return rc;
        const int tbg = t.wm * 16 * t.mt;
        rc = launch_part(t, m_base, (int)((groups - m_base + tbg - 1) / tbg), tail_per_cu);
        if (rc != RS_OK) return rc;
    } else {
        BG = s->wm * 16 * s->mt;
        BN = s->wn * 16 * s->nt;
        const int rc = launch_part(*s, 0, (a.n_groups + BG - 1) / BG, per_cu);
        if (rc != RS_OK) return rc;
    }
    if (bm_out) *bm_out = 4 * BG;           // reported in conv rows, like the other kernels
    if (bn_out) *bn_out = BN;
    return RS_OK;
}

}  // namespace rs
