// K3m (fp32, Winograd, SMALL batches): the conv blocks of an RS_F32W model when a launch has only a handful of rows.
//
//   Conv1d(C_in -> C_out, k=3, 'same', bias) -> ReLU -> MaxPool1d(2,2)   (riser/nets/cnn.py:52-65)
//
// Model.classify is called at batch 1 by the reference's own loop (riser/model.py:22-28, riser/control.py:68-69), and
// a ReadUntil batch is whatever arrives.  The tiled kernels (conv_wino.hip, conv_wino4.hip) are built for hundreds of
// reads: a tile is 256-512 rows by 32-80 channels and one workgroup walks ALL K chunks of its tile between barriers.
// At batch 1 the late layers have 8-128 rows: a launch is 20-50 tiles whose rows are almost all padding, each a serial
// chain of 50-70 chunk items of ~3 us - layer 11 alone takes 0.18 ms for 41 MFLOP, the whole forward 0.59 ms.
//
// Here the unit of work is ONE WORKGROUP = one 16 x 16 accumulator tile (16 pooled rows / F(4,3) groups x 16 output
// channels) over the whole reduction, ONE WAVE per Winograd component:
//   * every wave streams its component's weights from L2 / Infinity Cache in 16-byte units, three chunks of the reduction
//     ahead, through registers into a wave-private LDS image and reads its MFMA fragments from there (the weights of a tile
//     are read exactly once); the tile's input rows are staged ONCE per workgroup - every wave loads 1 / NC of them into an
//     image the waves share, three buffers, one barrier per chunk (round 5; round 4 staged them once per WAVE with no barrier
//     in the loop: RS_SMALL_SHARED=0);
//   * a launch is (rows / 16) x (C_out / 16) workgroups - 107 for layer 11 at batch 1 - spread over the chip, each wave a
//     chain of C_in / 4 MFMAs; from ~2 workgroups per CU up a wave takes TWO channel sub-tiles (NW = 2: half the workgroups,
//     two MFMAs per k-step on one transformed input fragment).
// The MFMA sequence per accumulator (chunk order, k-step order, operand roles, the Winograd input and output transforms
// and every rounding in them) is that of the tiled kernels, so the results are BIT-IDENTICAL to theirs: a read classified
// alone equals its row of a 512-read batch (tests/test_gpu_small.py).  The launch planner picks this kernel when the
// launch has at most kSmallMaxWaves tiles (api.hip); layers 0 + 1 keep the streaming kernel.
#include "common.hpp"

#include <algorithm>
#include <utility>

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

constexpr unsigned kOob = 0x80000000u;

struct SmallArgs {
    const float* x;        // [rows_in][cp_in]
    const float* w;        // packed [n_alloc][nch][NC][kc] (the tiled kernels' packing)
    const float* bias;     // [n_alloc]
    float* y;              // [rows_in / 2][cp_out]
    const int32_t* len;    // per block
    unsigned x_bytes, w_bytes, y_bytes, bias_bytes;
    int rows_in, rows_out;
    int units;             // MFMA columns of the launch: pooled rows (F(2,3)) or groups of four input rows (F(4,3))
    int P_out;
    int cp_in, cp_out;
    int nch;
    int shift_out;
    int n_ntiles;
    int n_blocks;          // entries of len[]
};

// NC = 4: Winograd F(2,3) (a unit = pooled row T: inputs x[2T-1 .. 2T+2]); NC = 6: F(4,3) (a unit = group G: inputs
// x[4G-1 .. 4G+4], pooled rows 2G and 2G+1).  KC = the layer's channel chunk (the weight packing).
//
// A workgroup = one 16 x 16 tile, a WAVE = one Winograd COMPONENT of it: the NC component accumulators of a tile are
// independent chains over the reduction (they only meet in the output transform), so NC waves on the CU's four SIMDs walk
// them side by side - a single wave issuing everything (fragment reads, transforms, staging and NC MFMAs per k-step, which
// in fp32 do not overlap: DESIGN.md 5) ran layer 11 in 51 us, 3.4x its MFMA time.
// Per chunk of KC input channels a wave moves ITS operands in 16-byte units - its component's 16 weight rows x KC floats
// and (round 4 form, SH = false) the tile's AR = 16 STRIDE + (R - STRIDE) input rows x KC floats (every wave its own copy:
// 3-6 KB from L2, no barrier to share one; SH = true: its 1 / NC share of them, into the shared image) - global -> registers (kDepth chunks ahead, ~8 loads per chunk: the whole look-ahead fits the 6-bit vmcnt
// counter) -> wave-private LDS -> MFMA fragments (ds_read_b32 at the tiled kernels' conflict-free pitch of KC + 2 floats,
// input rows split into STRIDE planes by row mod STRIDE so that the lanes of a fragment read walk consecutive rows of one
// plane).  The chunk's byte offset is a SCALAR operand of the buffer loads: no per-lane address arithmetic in the loop.
// LDS operations of one wave execute in order and no other wave touches its private image: no barrier in the loop for
// the weights (SH = false: none at all; SH = true: one per chunk for the shared input rows); ONE at the end, where the
// component accumulators meet in LDS for the output transform.
template <int NC, int KC, int NW = 1>
struct SmallGeom {
    static constexpr int R = NC == 4 ? 4 : 6;             // input rows of a unit
    static constexpr int STRIDE = NC == 4 ? 2 : 4;        // input rows between consecutive units
    static constexpr int KQ = KC / 4;                     // k-steps per chunk = 16-byte units per input row
    static constexpr int AR = 16 * STRIDE + (R - STRIDE); // input rows of the tile
    static constexpr int S = KC + 2;                      // LDS pitch of a row (input and weight)
    static constexpr int PLROWS = 17;                     // rows per plane: index (slab row / STRIDE) <= 16
    static constexpr int A_LDS = STRIDE * PLROWS * S;     // floats
    static constexpr int WR = 16 * NW;                    // weight rows of a wave: NW channel sub-tiles
    static constexpr int BUF = A_LDS + WR * S + 8;        // a wave's private image: input rows + weights + a dump slot for the lanes of a pass that hold no unit
    static constexpr int WBUF = WR * S + 8;               // ... weights only, when the input rows are shared
};

// the reduction of ONE component (COMP, a compile-time constant: the row of B^T a wave applies is straight-line code) of
// one tile; returns the wave's accumulator
// SH (round 5, the default): the tile's input rows are staged ONCE per workgroup - every wave
// loads 1 / NC of them into an image the NC waves share (three buffers, one barrier per chunk: a wave is never more than a
// chunk ahead of the slowest) - instead of once per wave: a workgroup pulls (AR + 16 NC) instead of NC (AR + 16) rows per
// chunk from L2, a third of the bytes that bound a 16-read launch of layer 10 (430 MB at 9 TB/s).  Weights stay private.
// NW (round 5): channel sub-tiles per wave - a workgroup covers 16 units x 16 NW channels, a wave issues NW MFMAs per k-step
// on one transformed input fragment (launches of several workgroups per CU: half the workgroups, each 1.3 x as long).
template <int NC, int KC, int COMP, bool SH, int NW>
__device__ __forceinline__ void small_chain(const SmallArgs& a, float* lds, float* lds_sh, int lane, int u0, int n0, f32x4 (&acc)[NW]) {
    constexpr int comp = COMP;
    constexpr int kDepth = 3;                      // chunks in flight per wave
    using G = SmallGeom<NC, KC, NW>;
    constexpr int R = G::R, STRIDE = G::STRIDE, KQ = G::KQ, AR = G::AR, S = G::S, PLROWS = G::PLROWS, A_LDS = G::A_LDS,
                  BUF = SH ? G::WBUF : G::BUF, W0 = SH ? 0 : A_LDS;   // private image and where its weights start
    (void)R;
    constexpr int A_UNITS = AR * KQ, W_UNITS = G::WR * KQ;
    constexpr int A_LANES = SH ? 64 * NC : 64;     // lanes that share the loads of one input image
    constexpr int A_PER = (A_UNITS + A_LANES - 1) / A_LANES, W_PER = (W_UNITS + 63) / 64;
    constexpr int ASH = A_LDS + 8;                 // one shared image (+ a dump slot)
    const int r = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);

    // ---- staging maps: unit q = lane + 64 j.  Input rows before the buffer (row -1 of the first tile) and past its end
    // resolve to out-of-range offsets: zeros, the conv's 'same' padding at the batch edges (inside the batch the producer
    // wrote zero rows behind every read).  The LAST chunk may reach past the row width: those units read zeros too (their
    // weights are zero rows of the packing, but an activation row's neighbour in memory is not).
    const int g0 = STRIDE * u0 - 1;                // first input row of the tile
    const int last_kc = a.cp_in - (a.nch - 1) * KC;          // channels of the last chunk
    unsigned a_off[A_PER], a_off_last[A_PER], w_off[W_PER];
    int a_st[A_PER], w_st[W_PER];                  // LDS float offsets
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
        const int q = (SH ? comp * 64 : 0) + lane + A_LANES * j;
        const int row = q / KQ, c4 = q - row * KQ;
        const int g = g0 + row;
        const bool ok = q < A_UNITS && g >= 0 && g < a.rows_in;
        a_off[j] = ok ? (unsigned)(g * a.cp_in + 4 * c4) * 4u : kOob;
        a_off_last[j] = (ok && 4 * c4 < last_kc) ? a_off[j] : kOob;
        a_st[j] = q < A_UNITS ? ((row % STRIDE) * PLROWS + row / STRIDE) * S + 4 * c4 : (SH ? A_LDS : BUF - 8);
    }
#pragma unroll
    for (int j = 0; j < W_PER; ++j) {
        const int q = lane + 64 * j;
        const int row = q / KQ, c4 = q - row * KQ;
        w_off[j] = q < W_UNITS ? (unsigned)(((n0 + row) * a.nch * NC + comp) * KC + 4 * c4) * 4u : kOob;
        w_st[j] = q < W_UNITS ? W0 + row * S + 4 * c4 : BUF - 8;
    }

    u32x4 ra[kDepth][A_PER], rw[kDepth][W_PER];
    // every load is unconditional (the look-ahead past the last chunk re-reads the last one), so the compiler's vmcnt
    // bookkeeping stays a static count
    auto load_chunk = [&](auto D_, int c) {
        constexpr int dd = decltype(D_)::value;
        const int cl = c < a.nch ? c : a.nch - 1;                      // past the end: the last chunk again (never used)
        const bool last = cl == a.nch - 1;                             // wave-uniform
        const int xc = cl * KC * 4, wc = cl * NC * KC * 4;             // scalar byte offsets of the chunk
#pragma unroll
        for (int j = 0; j < A_PER; ++j)
            ra[dd][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, last ? a_off_last[j] : a_off[j], xc, 0);
#pragma unroll
        for (int j = 0; j < W_PER; ++j) rw[dd][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off[j], wc, 0);
    };
    auto store_chunk = [&](auto D_, float* buf, float* abuf) {  // registers -> LDS images (8-byte stores: the pitch is even)
        constexpr int dd = decltype(D_)::value;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            uint2* p = reinterpret_cast<uint2*>(abuf + a_st[j]);
            p[0] = make_uint2(ra[dd][j].x, ra[dd][j].y);
            p[1] = make_uint2(ra[dd][j].z, ra[dd][j].w);
        }
#pragma unroll
        for (int j = 0; j < W_PER; ++j) {
            uint2* p = reinterpret_cast<uint2*>(buf + w_st[j]);
            p[0] = make_uint2(rw[dd][j].x, rw[dd][j].y);
            p[1] = make_uint2(rw[dd][j].z, rw[dd][j].w);
        }
    };

    // fragment addresses: activation operand (MFMA "B"): column r = unit u0 + r, k = kq -> channel 4 st + kq of the unit's
    // input rows; weight operand (MFMA "A"): row r = output channel n0 + r, k = kq
    const int a_rd = r * S + kq;                   // + ((k % STRIDE) * PLROWS + k / STRIDE) * S + 4 st
    const int w_rd = W0 + r * S + kq;              // + 4 st
    // the input rows d_k this wave's component needs: V = B^T d, one row of B^T per wave.  Which rows, and the expression,
    // depend on the component - wave-uniform, so the selection is a scalar branch around a few VALU instructions; the
    // expressions (and their roundings) are those of conv_wino.hip / conv_wino4.hip: xform()
    struct Frag {
        float e[4], uf[NW];
    };
    auto read_frag = [&](Frag& f, const float* buf, const float* abuf, auto ST_) {
        constexpr int st = decltype(ST_)::value;
        auto row = [&](int k) { return abuf[a_rd + ((k % STRIDE) * PLROWS + k / STRIDE) * S + 4 * st]; };
        if constexpr (NC == 4) {                   // comp 0: d0, d2 | 1: d1, d2 | 2: d2, d1 | 3: d1, d3
            constexpr int k0 = comp == 0 ? 0 : comp == 2 ? 2 : 1, k1 = comp == 0 ? 2 : comp == 1 ? 2 : comp == 2 ? 1 : 3;
            f.e[0] = row(k0);
            f.e[1] = row(k1);
            f.e[2] = f.e[3] = 0.f;
        } else {                                   // comp 0: d0, d2, d4 | 1 .. 4: d1, d2, d3, d4 | 5: d1, d3, d5
            if constexpr (comp == 0 || comp == 5) {
                constexpr int o = comp == 5 ? 1 : 0;
                f.e[0] = row(o);
                f.e[1] = row(o + 2);
                f.e[2] = row(o + 4);
                f.e[3] = 0.f;
            } else {
                f.e[0] = row(1);
                f.e[1] = row(2);
                f.e[2] = row(3);
                f.e[3] = row(4);
            }
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) f.uf[j] = buf[w_rd + j * 16 * S + 4 * st];
    };
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto kstep = [&](const Frag& f) {
        float v;
        if constexpr (NC == 4) {                   // conv_wino.hip: v0 = d0 - d2, v1 = d1 + d2, v2 = d2 - d1, v3 = d1 - d3
            if constexpr (comp == 1) v = f.e[0] + f.e[1]; else v = f.e[0] - f.e[1];
        } else {                                   // conv_wino4.hip: xform()
            if constexpr (comp == 0 || comp == 5) {   // fmaf(4, d0, fmaf(-5, d2, d4)) resp. fmaf(4, d1, fmaf(-5, d3, d5))
                v = fmaf(4.0f, f.e[0], fmaf(-5.0f, f.e[1], f.e[2]));
            } else if constexpr (comp <= 2) {                // p = fmaf(-4, d2, d4), q = fmaf(-4, d1, d3): v1 = p + q, v2 = p - q
                const float p = fmaf(-4.0f, f.e[1], f.e[3]), q = fmaf(-4.0f, f.e[0], f.e[2]);
                if constexpr (comp == 1) v = p + q; else v = p - q;
            } else {                               // s2 = d4 - d2, t2 = d3 - d1: v3 = fmaf(2, t2, s2), v4 = fmaf(-2, t2, s2)
                const float s2 = f.e[3] - f.e[1], t2 = f.e[2] - f.e[0];
                v = fmaf(comp == 3 ? 2.0f : -2.0f, t2, s2);
            }
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.uf[j], v, acc[j], 0, 0, 0);
    };

    // chunk c lives in register set c % kDepth and LDS buffer c & 1.  A wave has nobody to hide its LDS latency behind,
    // so the fragments of k-step s + 1 (the first of the next chunk, whose image is written at the START of this chunk's
    // step, behind nothing but the in-order LDS queue) are read before the MFMA of k-step s.
    // image of chunk c: weights in this wave's buffer c & 1; input rows there too, or (SH) in shared buffer c % 3
    auto a_img = [&](int c) -> float* { return SH ? lds_sh + (c % 3) * ASH : lds + (c & 1) * BUF; };
    static_for<kDepth>([&](auto D_) { load_chunk(D_, decltype(D_)::value); });
    store_chunk(std::integral_constant<int, 0>{}, lds, a_img(0));
    load_chunk(std::integral_constant<int, 0>{}, kDepth);
    if constexpr (SH) __syncthreads();
    // fragments are read kAhead k-steps before their MFMA (a ring of kAhead + 1 register sets, indexed statically: the body
    // unrolled below holds kDepth * KQ k-steps, a multiple of kAhead + 1)
    constexpr int kAhead = 2;
    static_assert((kDepth * KQ) % (kAhead + 1) == 0 && KQ > kAhead, "static fragment ring");
    Frag fr[kAhead + 1];
    read_frag(fr[0], lds, a_img(0), std::integral_constant<int, 0>{});
    read_frag(fr[1], lds, a_img(0), std::integral_constant<int, 1>{});
    for (int c = 0; c < a.nch; c += kDepth) {
        static_for<kDepth>([&](auto D_) {
            constexpr int dd = decltype(D_)::value;
            constexpr int nx = (dd + 1) % kDepth;
            const int cc = c + dd;                                     // chunk computed in this step
            const float* here = lds + (cc & 1) * BUF;
            float* next = lds + ((cc + 1) & 1) * BUF;
            const float* here_a = a_img(cc);
            float* next_a = a_img(cc + 1);
            // chunk cc + 1: its registers (loaded kDepth - 1 steps ago) -> the other LDS buffer, then the set is free for
            // chunk cc + 1 + kDepth.  (Every fragment read of chunk cc - 1, whose image this overwrites, was issued during
            // chunk cc - 1's own k-steps: the look-ahead never reaches back.  SH: the shared image of chunk cc + 1 is the
            // one of chunk cc - 2, which every wave left before the barrier of step cc - 1.)
            store_chunk(std::integral_constant<int, nx>{}, next, next_a);
            load_chunk(std::integral_constant<int, nx>{}, cc + 1 + kDepth);
            if constexpr (SH) __syncthreads();                         // chunk cc + 1's shared image is complete
            if (cc < a.nch) {                                          // wave-uniform, and the same in every wave
                static_for<KQ>([&](auto ST_) {
                    constexpr int st = decltype(ST_)::value;
                    constexpr int i = dd * KQ + st;                    // k-step of the unrolled body
                    if constexpr (st + kAhead < KQ)
                        read_frag(fr[(i + kAhead) % (kAhead + 1)], here, here_a, std::integral_constant<int, st + kAhead>{});
                    else
                        read_frag(fr[(i + kAhead) % (kAhead + 1)], next, next_a, std::integral_constant<int, st + kAhead - KQ>{});
                    kstep(fr[i % (kAhead + 1)]);
                });
            }
        });
    }
}

template <int NC, int KC, bool SH = false, int NW = 1>
__global__ __launch_bounds__(64 * NC) void conv_small_f32_kernel(const SmallArgs a) {
    using G = SmallGeom<NC, KC, NW>;
    constexpr int PB = SH ? G::WBUF : G::BUF;      // a wave's private image (two buffers)
    constexpr int LDS_FLOATS = NC * 2 * PB + (SH ? 3 * (G::A_LDS + 8) : 0);
    static_assert(LDS_FLOATS >= NW * NC * 64 * 4, "the components' meeting place reuses the staging images");
    __shared__ __attribute__((aligned(16))) float lds_all[LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int comp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // this wave's component
    float* lds = lds_all + comp * 2 * PB;
    float* lds_sh = lds_all + NC * 2 * PB;
    const int r = lane & 15, kq = lane >> 4;
    const int tile = blockIdx.x;
    const int mi = tile / a.n_ntiles, ni = tile - mi * a.n_ntiles;
    const int u0 = mi * 16, n0 = ni * 16 * NW;
    f32x4 acc[NW];
    switch (comp) {                                                    // wave-uniform: one scalar branch per wave
        case 0: small_chain<NC, KC, 0, SH, NW>(a, lds, lds_sh, lane, u0, n0, acc); break;
        case 1: small_chain<NC, KC, 1, SH, NW>(a, lds, lds_sh, lane, u0, n0, acc); break;
        case 2: small_chain<NC, KC, 2, SH, NW>(a, lds, lds_sh, lane, u0, n0, acc); break;
        case 3: small_chain<NC, KC, 3, SH, NW>(a, lds, lds_sh, lane, u0, n0, acc); break;
        case 4: small_chain<NC, KC, (NC > 4 ? 4 : 0), SH, NW>(a, lds, lds_sh, lane, u0, n0, acc); break;
        default: small_chain<NC, KC, (NC > 4 ? 5 : 0), SH, NW>(a, lds, lds_sh, lane, u0, n0, acc); break;
    }

    // ---- the components meet: [comp][lane] x 4 floats in LDS (the staging images are dead), one barrier, then wave h of
    // the first PR waves forms pooled row h of every unit: output transform, MaxPool, + bias, ReLU, length mask; 16 bytes
    // per lane ------------------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.bias_bytes, 0x00020000);
    __syncthreads();                                                   // every wave is done with its staging image
    f32x4* meet = reinterpret_cast<f32x4*>(lds_all);
#pragma unroll
    for (int j = 0; j < NW; ++j) meet[(j * NC + comp) * 64 + lane] = acc[j];
    __syncthreads();
    constexpr int PR = NC == 4 ? 1 : 2;            // pooled rows per unit
    if (comp >= PR) return;
    const int h = comp;
    const int pr = PR * (u0 + r) + h;              // pooled row of the launch
    const int b = pr / a.P_out;
    const int pin = pr - b * a.P_out;
    const bool in_range = pr < a.rows_out && b < a.n_blocks;
    const bool valid = in_range && pin < (a.len[in_range ? b : 0] >> a.shift_out);
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        f32x4 m[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) m[q] = meet[(j * NC + q) * 64 + lane];
        const int col = n0 + 16 * j + 4 * kq;
        const f32x4 bi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (unsigned)col * 4u, 0, 0));
        const unsigned coloff = col < a.cp_out ? (unsigned)col * 4u : kOob;
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float y0, y1;
            if constexpr (NC == 4) {
                const float m1 = m[0][q], m2 = m[1][q], m3 = m[2][q], m4 = m[3][q];
                y0 = (m1 + m2) + m3;
                y1 = (m2 - m3) - m4;
            } else {
                const float m0_ = m[0][q], m1 = m[1][q], m2 = m[2][q], m3 = m[3][q], m4 = m[4][q], m5 = m[5][q];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                if (h == 0) {
                    y0 = (m0_ + s12) + s34;
                    y1 = fmaf(2.0f, d34, d12);
                } else {
                    y0 = fmaf(4.0f, s34, s12);
                    y1 = fmaf(8.0f, d34, d12) + m5;
                }
            }
            o[q] = valid ? fmaxf(fmaxf(y0, y1) + bi[q], 0.0f) : 0.0f;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_y,
                                               in_range ? (unsigned)pr * (unsigned)(a.cp_out * 4) + coloff : kOob, 0, 0);
    }
}

using KernelFn = void (*)(const SmallArgs);

KernelFn pick(int nc, int kc, bool shared = false, int nw = 1) {
    if (shared && nw == 2) {
        if (nc == 4) return kc == 16 ? conv_small_f32_kernel<4, 16, true, 2> : kc == 20 ? conv_small_f32_kernel<4, 20, true, 2>
                          : kc == 24 ? conv_small_f32_kernel<4, 24, true, 2> : nullptr;
        return kc == 16 ? conv_small_f32_kernel<6, 16, true, 2> : kc == 20 ? conv_small_f32_kernel<6, 20, true, 2> : nullptr;
    }
    if (shared) {
        if (nc == 4) return kc == 16 ? conv_small_f32_kernel<4, 16, true> : kc == 20 ? conv_small_f32_kernel<4, 20, true>
                          : kc == 24 ? conv_small_f32_kernel<4, 24, true> : nullptr;
        return kc == 16 ? conv_small_f32_kernel<6, 16, true> : kc == 20 ? conv_small_f32_kernel<6, 20, true> : nullptr;
    }
    if (nc == 4) return kc == 16 ? conv_small_f32_kernel<4, 16> : kc == 20 ? conv_small_f32_kernel<4, 20>
                      : kc == 24 ? conv_small_f32_kernel<4, 24> : nullptr;
    return kc == 16 ? conv_small_f32_kernel<6, 16> : kc == 20 ? conv_small_f32_kernel<6, 20> : nullptr;
}

}  // namespace

// tiles (= waves) a launch of this kernel would have for `rows_in` input rows of layer L
int64_t conv_small_f32_waves(const ConvLayerDev& L, int64_t rows_in) {
    const int64_t units = L.wino_m == 4 ? (rows_in + 3) / 4 : rows_in / 2;
    return ((units + 15) / 16) * (int64_t)(round_up(L.c_out, 16) / 16);
}

// Estimate in shader cycles, fitted to tools/layer_times.py at 1 ... 64 reads with every layer forced onto this kernel
// (profiles/r05_small_kernel_shared_input_rows.txt, r05_small_kernel_two_channel_subtiles.txt): a wave's chain is C_in / 4
// k-steps of ~131 cycles (F(4,3): six waves on four SIMDs; 210 with two channel sub-tiles per wave) or ~70 (F(2,3): four waves;
// 121) plus ~2500 of prologue and output transform; several workgroups share a CU (35-50 KB of LDS each) and a launch lasts
// 0.75 + 1.15 W (F(4,3)) resp. 0.35 + 1.1 W (F(2,3)) chains for W workgroups per CU (layer 10, one sub-tile: 21 / 21 / 31 / 46 /
// 76 us at 71 / 142 / 284 / 568 / 1136 workgroups; two: 27 / 28 / 28 / 44 / 65 at half as many); + launch, prologue and the timing
// events' own cost, like the tiled kernels' thin-launch fit it is compared with.  *nw_out: the cheaper number of sub-tiles.
double conv_small_f32_cost(const ConvLayerDev& L, int64_t rows_in, int num_cu, int* nw_out) {
    const double ksteps = (L.cp_in + 3) / 4;
    const bool f43 = L.wino_m == 4;
    const int64_t wgs1 = conv_small_f32_waves(L, rows_in);
    const int64_t n16 = round_up(L.c_out, 16) / 16;
    double best = 1e300;
    for (int nw = 1; nw <= 2; ++nw) {
        const int64_t wgs = wgs1 / n16 * ((n16 + nw - 1) / nw);
        const double per_step = f43 ? (nw == 1 ? 131.0 : 210.0) : (nw == 1 ? 70.0 : 121.0);
        const double w = (double)wgs / num_cu;
        const double chains = std::max(1.0, f43 ? 0.75 + 1.15 * w : 0.35 + 1.1 * w);
        const double cost = (ksteps * per_step + 2500.0) * chains + 23000.0;
        if (cost < best) {
            best = cost;
            if (nw_out) *nw_out = nw;
        }
    }
    return best;
}

bool conv_small_f32_ok(const ConvLayerDev& L) { return pick(L.wino_m == 4 ? 6 : 4, L.plan.kc) != nullptr; }

int launch_conv_small_f32(const ConvLayerDev& L, const float* d_x, float* d_y, const int32_t* d_len, int B, int P_in,
                          int layer_index, int num_cu, hipStream_t st, int* bm_out, int* bn_out) {
    const int nc = L.wino_m == 4 ? 6 : 4;
    // the input rows of a tile are staged once per workgroup (RS_SMALL_SHARED=0: once per wave, the round-4 form, kept as a
    // cross-check: 20-35 % slower at every launch size)
    const bool shared = L.hooks->small_shared != 0;
    // one or two channel sub-tiles per wave: the cost model's choice (RS_SMALL_NW forces 1 / 2)
    int nw = 1;
    conv_small_f32_cost(L, (int64_t)B * P_in, num_cu, &nw);
    if (!shared) nw = 1;
    else if (L.hooks->small_nw > 0) nw = std::min(2, L.hooks->small_nw);
    KernelFn fn = pick(nc, L.plan.kc, shared, nw);
    if (!fn) {
        set_error("conv_small_f32: unsupported channel chunk %d", L.plan.kc);
        return RS_ERR_ARG;
    }
    const int64_t rows64 = (int64_t)B * P_in;
    const int64_t xb = rows64 * L.cp_in * 4, wb = (int64_t)L.plan.n_alloc * L.plan.nch * nc * L.plan.kc * 4,
                  yb = rows64 / 2 * L.cp_out * 4;
    if (xb >= 0x80000000LL || wb >= 0x80000000LL || yb >= 0x80000000LL) {
        set_error("conv_small_f32: a buffer exceeds the 2 GiB buffer window");
        return RS_ERR_ARG;
    }
    SmallArgs a;
    a.x = d_x;
    a.w = static_cast<const float*>(L.d_w);
    a.bias = L.d_bias;
    a.y = d_y;
    a.len = d_len;
    a.x_bytes = (unsigned)xb;
    a.w_bytes = (unsigned)wb;
    a.y_bytes = (unsigned)yb;
    a.bias_bytes = (unsigned)L.plan.n_alloc * 4u;
    a.rows_in = (int)rows64;
    a.rows_out = (int)(rows64 / 2);
    a.units = (int)(nc == 6 ? (rows64 + 3) / 4 : rows64 / 2);
    a.P_out = P_in / 2;
    a.cp_in = L.cp_in;
    a.cp_out = L.cp_out;
    a.nch = L.plan.nch;
    a.shift_out = layer_index + 1;
    a.n_ntiles = (round_up(L.c_out, 16) / 16 + nw - 1) / nw;
    a.n_blocks = B;
    const int n_mtiles = (a.units + 15) / 16;
    hipLaunchKernelGGL(fn, dim3((unsigned)(n_mtiles * a.n_ntiles)), dim3(64 * nc), 0, st, a);
    RS_HIP(hipGetLastError());
    if (bm_out) *bm_out = 16 * (nc == 6 ? 4 : 2);      // conv rows per tile, as the tiled kernels report them
    if (bn_out) *bn_out = 16 * nw;
    return RS_OK;
}

}  // namespace rs
