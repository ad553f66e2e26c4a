// K3 (16-bit): one ConvNet block i >= 1 on the bf16 / f16 MFMA (fp32 accumulate).
//
// Same lowering as conv_f32.hip (position-major activations, GEMM M = positions, N = output
// channels, K = 3 * C_in, bias + ReLU + MaxPool fused in registers), re-tiled for a matrix
// pipe that is 16x faster:
//   * v_mfma_f32_16x16x32_{bf16,f16}: one k-step consumes 32 input channels of one tap; a lane
//     holds 8 consecutive channels (16 bytes) of its A row / B column, so fragments are single
//     ds_read_b128;
//   * K is cut into PANELS of 32 channels; a work item = (tile, panel) stages a (BM+2) x 64 B
//     slab of X and a 3 x BN x 64 B slab of weights; the three taps read the same X slab at
//     row offsets 0 / 1 / 2;
//   * 64-byte LDS rows are XOR-swizzled at 16-byte granularity, phys_slot = slot ^ (2 *
//     ((row >> 2) & 1)): found by exhaustive search to make the ds_read_b128 lane groups of
//     gfx950 ({0-3,12-15,20-27}, ...) conflict-free for all three tap shifts at once;
//   * persistent 8-wave workgroups, double-buffered LDS, register prefetch of item k+1 issued
//     above the MFMA block of item k (unconditional loads, zero page for masked units).
// Channels are padded to 8 in HBM (16-byte rows pieces) and to 32 in K (zero weights).
#include "common.hpp"
#include "tile_walk.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

// Diagnostic build only (-DRS_ITEM_STAMPS, tools/h16_stamps.py): per-phase s_memtime sums of the item loop
#ifdef RS_ITEM_STAMPS
#include <vector>
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

struct ConvHArgs {
    const unsigned short* x;
    const unsigned short* w;   // packed [panel][tap][n_alloc][32]
    const float* bias;         // [n_alloc]
    unsigned short* y;
    const int32_t* len;
    const unsigned short* zero;
    unsigned x_bytes, w_bytes;   // sizes of the activation buffer and of the packed weights (< 2^31)
    int rows_in;
    int P_out;
    float inv_P_out;
    int cp_in, cp_out;
    int n_panels;
    int n_alloc;
    int shift_out;
    WalkArgs walk;         // tile grid, order and dead-tile flag (tile_walk.hpp)
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

// KS = 16-byte k-steps per tap = panel width / 32 channels.  64-byte rows (KS = 1): slot ^ 2 * ((row >> 2) & 1);
// 128-byte rows (KS = 2): slot ^ (row & 7) - both found by search over the ds_read_b128 lane groups of gfx950
// ({0-3,12-15,20-27}, ...) to be conflict-free for all three tap shifts (and both k-steps) at once.
template <int KS>
__device__ __forceinline__ int swz(int row) {
    return KS == 1 ? ((row >> 2) & 1) << 1 : (row & 7);
}

template <int WM, int WN, int MT, int NT, bool F16, int KS>
__global__ __launch_bounds__(kThreads, 2) void conv_h16_kernel(const ConvHArgs a) {
    static_assert(WM * WN == 8, "8 waves per workgroup");
    static_assert(KS == 1 || KS == 2, "panels of 32 or 64 channels");
    constexpr int BM = WM * 16 * MT;
    constexpr int BN = WN * 16 * NT;
    constexpr int ROWB = 64 * KS;                       // LDS row: one panel of one position / output channel
    constexpr int SLOTS = 4 * KS;                       // 16-byte slots per row
    constexpr int RPP = kThreads / SLOTS;               // slab rows per staging pass
    constexpr int PANEL = 32 * KS;
    constexpr int A_BYTES = (BM + 2) * ROWB;
    constexpr int BUF_BYTES = A_BYTES + 3 * BN * ROWB;
    constexpr int A_UNITS = (BM + 2) * SLOTS;
    constexpr int B_UNITS = 3 * BN * SLOTS;
    constexpr int A_PER = (A_UNITS + kThreads - 1) / kThreads;
    constexpr int B_PER = (B_UNITS + kThreads - 1) / kThreads;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int r = lane & 15, g = lane >> 4;

    // ---- per-thread staging map ------------------------------------------------------------------
    // A unit f = tid + u * 512: slab row f >> 2 (= tid / 4 + 128 u), 16-byte slot f & 3.  The swizzle of a
    // row depends on (row >> 2) & 1 only, which is the same for every pass u of a thread, so pass u is an
    // immediate LDS offset.  Global loads are BUFFER loads with 32-bit byte offsets: rows outside the
    // activation buffer, channel slots beyond cp_in, unused units and the prefetch after the last item
    // resolve to an out-of-range offset (hardware returns zeros): one add per unit, no 64-bit address
    // arithmetic, no predicates, no zero page.
    constexpr unsigned kOob = 0x80000000u;
    const int a_row0 = tid / SLOTS, a_c = tid % SLOTS;
    const int a_lds0 = a_row0 * ROWB + ((a_c ^ swz<KS>(a_row0)) << 4);         // + u * RPP * ROWB (RPP % 8 == 0)
    const unsigned a_tb = ((unsigned)a_row0 * a.cp_in + 8 * a_c) * 2u;         // + item base + u * a_step
    const unsigned a_step = (unsigned)(RPP * a.cp_in) * 2u;
    int b_lds[B_PER];
    unsigned b_g[B_PER];                                                        // byte offset inside one panel's weights
#pragma unroll
    for (int u = 0; u < B_PER; ++u) {
        const int f = tid + u * kThreads;
        const int tap = f / (BN * SLOTS), rem = f - tap * (BN * SLOTS);
        const int n = rem / SLOTS, c = rem % SLOTS;
        b_lds[u] = A_BYTES + (tap * BN + n) * ROWB + ((c ^ swz<KS>(n)) << 4);
        // weights stay packed in 32-channel panels [panel][tap][n_alloc][32]: slot c of a 64-channel row is slot c & 3 of
        // panel 2p + (c >> 2); the second half of an odd last panel lies past the buffer (hardware zeros)
        b_g[u] = f < B_UNITS ? (unsigned)((((c >> 2) * 3 + tap) * a.n_alloc + n) * 32 + 8 * (c & 3)) * 2u : kOob;
    }
    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w), 0, a.w_bytes, 0x00020000);

    u32x4 ra[A_PER], rb[B_PER];
    auto load_item = [&](int m0, int n0, int p, bool live) {
        bool a_ok = live && p * PANEL + 8 * a_c < a.cp_in;
#ifdef RS_ABL_NOLOAD
        a_ok = false;
        live = false;
#endif
        // row -1 (m0 == 0, slab row 0) wraps to an offset >= 2^31: out of range, zeros
        const unsigned a_ib = a_ok ? a_tb + (unsigned)((m0 - 1) * a.cp_in + p * PANEL) * 2u : kOob;
#pragma unroll
        for (int u = 0; u < A_PER; ++u) {
            unsigned off = a_ib + (unsigned)u * a_step;
            if ((u + 1) * RPP > BM + 2) off = (a_row0 + u * RPP < BM + 2) ? off : kOob;
            ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
        }
        const unsigned w_ib = live ? (unsigned)((p * KS * 3 * a.n_alloc + n0) * 32) * 2u : kOob;
#pragma unroll
        for (int u = 0; u < B_PER; ++u) rb[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_ib + b_g[u], 0, 0);
    };
    auto store_item = [&](unsigned char* buf) {
#ifdef RS_ABL_NOLDSW
#pragma unroll
        for (int u = 0; u < A_PER; ++u) asm volatile("" ::"v"(ra[u]));
#pragma unroll
        for (int u = 0; u < B_PER; ++u) asm volatile("" ::"v"(rb[u]));
#else
#pragma unroll
        for (int u = 0; u < A_PER; ++u)
            if (a_row0 + u * RPP < BM + 2) *reinterpret_cast<u32x4*>(buf + a_lds0 + u * RPP * ROWB) = ra[u];
#pragma unroll
        for (int u = 0; u < B_PER; ++u)
            if (b_g[u] != kOob) *reinterpret_cast<u32x4*>(buf + b_lds[u]) = rb[u];
#endif
    };

    // ---- tile walk (tile_walk.hpp) with dead-tile elimination -------------------------------------
    // A tile whose rows all lie beyond their read's length has an all-zero output: it is zero-filled in
    // next_live, without loads, MFMAs or pipeline slots, when the walk steps over it.  Such a tile lies
    // inside one read's slot (a tile containing a read start always has valid rows): one uniform look-up.
    const int tiles = a.walk.q_total;
    const int P_in_ = 2 * a.P_out;
    auto tile_origin = [&](int q, int& tm0, int& tn0) -> bool {
        int mi, nt_;
        const bool ok = walk_tile(a.walk, q, mi, nt_);
        tm0 = mi * BM;
        tn0 = nt_ * BN;
        return ok;
    };
    TileWalk walk;
    auto order_index = [&]() { return walk.next_index(a.walk); };
    auto next_live = [&]() {                                       // order index of this workgroup's next live tile
        int q = order_index();
        while (q < tiles) {
            int tm0, tn0;
            const bool valid = tile_origin(q, tm0, tn0);
            if (valid) {
                if (!a.walk.check_dead) break;
                const int b = tm0 / P_in_;
                const int t0 = tm0 - b * P_in_;
                if (!(t0 + BM <= P_in_ && t0 >= (as_const_len(a.len)[b] >> (a.shift_out - 1)))) break;
                // zero-fill the BM/2 x BN output tile (16-byte pieces; rows are cp_out wide)
                const int pieces_per_row = BN / 8;
                for (int f = threadIdx.x; f < (BM / 2) * pieces_per_row; f += blockDim.x) {
                    const int rr = f / pieces_per_row, cc = (f - rr * pieces_per_row) * 8;
                    const int prow = (tm0 >> 1) + rr, col = tn0 + cc;
                    if (2 * prow < a.rows_in && col < a.cp_out)
                        *reinterpret_cast<uint4*>(a.y + (int64_t)prow * a.cp_out + col) = make_uint4(0u, 0u, 0u, 0u);
                }
            }
            q = order_index();
        }
        return q;
    };
    int o = next_live();
    if (o >= tiles) return;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int p = 0;
    int m0, n0;
    tile_origin(o, m0, n0);
    load_item(m0, n0, 0, true);
    store_item(lds);
    __syncthreads();
    int buf = 0;

    // fragment read addresses: slab row of lane = wm*16*MT + i*16 + r + tap (A), wn*16*NT + j*16 + r (B)
    int a_rd[3][KS], b_rd[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int R = wm * 16 * MT + r + tap;
            a_rd[tap][ks] = R * ROWB + (((4 * ks + g) ^ swz<KS>(R)) << 4);
        }
        b_rd[ks] = A_BYTES + (wn * 16 * NT + r) * ROWB + (((4 * ks + g) ^ swz<KS>(r)) << 4);
    }

#ifdef RS_ITEM_STAMPS
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = tl;
    int n_items = 0;
#endif
    while (true) {
        int np = p + 1, no = o;
        if (np == a.n_panels) {
            np = 0;
            no = next_live();
        }
        const bool has_next = no < tiles;
        int nm0 = m0, nn0 = n0;
        if (has_next && np == 0) tile_origin(no, nm0, nn0);
        load_item(nm0, nn0, np, has_next);
        __builtin_amdgcn_sched_barrier(0);
        RS_STAMP(0);                                               // bookkeeping + prefetch issue

        const unsigned char* cur = lds + buf * BUF_BYTES;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                u32x4 af[MT], bf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    af[i] = *reinterpret_cast<const u32x4*>(cur + a_rd[tap][ks] + i * 16 * ROWB);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bf[j] = *reinterpret_cast<const u32x4*>(cur + b_rd[ks] + (tap * BN + j * 16) * ROWB);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<F16>(af[i], bf[j], acc[i][j]);
            }
        }

        RS_STAMP(1);                                               // fragment reads + MFMAs
        if (p == a.n_panels - 1) {
            // ---- epilogue: bias + ReLU + MaxPool(2,2) in registers, masked store ------------------
            float bias[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) bias[j] = a.bias[n0 + (wn * NT + j) * 16 + r];
            const int pr0 = m0 >> 1;
            const int b0 = pr0 / a.P_out;
            const int p0 = pr0 - b0 * a.P_out;
            int prow_[MT][2], lim_[MT][2], pin_[MT][2];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = m0 + (wm * MT + i) * 16 + 4 * g;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int prow = (row >> 1) + h;
                    const bool in = 2 * prow < a.rows_in;
                    const int t = p0 + (prow - pr0);
                    const int e = (int)(((float)t + 0.5f) * a.inv_P_out);
                    const int b = in ? b0 + e : 0;
                    prow_[i][h] = in ? prow : -1;
                    pin_[i][h] = t - e * a.P_out;
                    lim_[i][h] = a.len[b];
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int prow = prow_[i][h];
                    if (prow >= 0) {
                        const bool valid = pin_[i][h] < (lim_[i][h] >> a.shift_out);
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const int col = n0 + (wn * NT + j) * 16 + r;
                            if (col < a.cp_out) {
                                const float v =
                                    fmaxf(fmaxf(acc[i][j][2 * h], acc[i][j][2 * h + 1]) + bias[j], 0.0f);
#ifdef RS_ABL_NOSTORE
                                asm volatile("" ::"v"(v));
#else
                                a.y[(int64_t)prow * a.cp_out + col] = valid ? cvt16<F16>(v) : (unsigned short)0;
#endif
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        RS_STAMP(2);                                               // epilogue (on a tile's last panel)
#ifdef RS_ITEM_STAMPS
        ++n_items;
        if (!has_next && a.stamps && lane == 0 && blockIdx.x < 4) {
            unsigned long long* q = a.stamps + (blockIdx.x * 8 + wave) * 8;
            for (int k = 0; k < 5; ++k) q[k] = ph[k];
            q[5] = n_items;
            q[6] = __builtin_amdgcn_s_memtime() - t_begin;
        }
#endif
        if (!has_next) break;
        store_item(lds + (buf ^ 1) * BUF_BYTES);
        RS_STAMP(3);                                               // vmcnt wait + LDS store
#ifndef RS_ABL_NOBARRIER
        __syncthreads();
#endif
        RS_STAMP(4);                                               // barrier
        buf ^= 1;
        o = no;
        p = np;
        m0 = nm0;
        n0 = nn0;
    }
}

using KernelFn = void (*)(const ConvHArgs);

struct Shape {
    int wm, wn, mt, nt;
    KernelFn fn[2][2];     // [panel of 32 / 64 channels][bf16, f16]; null where the 64-channel slabs do not fit in LDS
};

constexpr size_t lds_bytes_of(int bm, int bn, int ks) { return 2 * (size_t)((bm + 2) + 3 * bn) * 64 * ks; }

template <int WM, int WN, int MT, int NT, bool F16>
constexpr KernelFn wide_panel_kernel() {
    if constexpr (lds_bytes_of(WM * 16 * MT, WN * 16 * NT, 2) <= 160 * 1024)
        return conv_h16_kernel<WM, WN, MT, NT, F16, 2>;
    else
        return nullptr;
}

#define RS_SHAPE(WM, WN, MT, NT)                                                                              \
    {WM, WN, MT, NT,                                                                                          \
     {{conv_h16_kernel<WM, WN, MT, NT, false, 1>, conv_h16_kernel<WM, WN, MT, NT, true, 1>},                 \
      {wide_panel_kernel<WM, WN, MT, NT, false>(), wide_panel_kernel<WM, WN, MT, NT, true>()}}}
const Shape kShapes[] = {
    RS_SHAPE(8, 1, 4, 2), RS_SHAPE(8, 1, 4, 3), RS_SHAPE(8, 1, 2, 5), RS_SHAPE(8, 1, 4, 5), RS_SHAPE(8, 1, 2, 7),
    RS_SHAPE(8, 1, 4, 7), RS_SHAPE(4, 2, 4, 2), RS_SHAPE(4, 2, 4, 3), RS_SHAPE(4, 2, 2, 4), RS_SHAPE(4, 2, 4, 4),
    RS_SHAPE(4, 2, 4, 5), RS_SHAPE(4, 2, 2, 6), RS_SHAPE(4, 2, 4, 6), RS_SHAPE(4, 2, 4, 7), RS_SHAPE(4, 2, 2, 8),
    RS_SHAPE(2, 4, 2, 2), RS_SHAPE(2, 4, 2, 4), RS_SHAPE(2, 4, 1, 4),
    RS_SHAPE(4, 2, 2, 5), RS_SHAPE(2, 4, 4, 2), RS_SHAPE(8, 1, 2, 6), RS_SHAPE(4, 2, 2, 3),   // for the 64-channel panels
};
constexpr int kNumAutoShapes = 18;     // 32-channel panels choose among the first 18
#undef RS_SHAPE
constexpr int kNumShapes = sizeof(kShapes) / sizeof(kShapes[0]);

size_t lds_bytes(const Shape& s, int ks) { return lds_bytes_of(s.wm * 16 * s.mt, s.wn * 16 * s.nt, ks); }

// cost model in SIMD cycles: MFMA issue (2 waves share a SIMD, 16 cycles per 16x16x32), LDS
// fragment traffic (256 B/clk per CU shared by 8 waves), L2->LDS staging per item, fixed
// per-item and per-tile overheads
const Shape* choose_shape(int64_t rows, int n16, int n_panels, int num_cu, int ks) {
    const Shape* best = nullptr;
    double best_cost = 1e300;
    for (int k = 0; k < (ks == 1 ? kNumAutoShapes : kNumShapes); ++k) {
        const Shape& s = kShapes[k];
        if (lds_bytes(s, ks) > 160 * 1024 || !s.fn[ks - 1][0]) continue;
        const int bm = s.wm * 16 * s.mt, bnt = s.wn * s.nt;
        const int64_t mtiles = (rows + bm - 1) / bm;
        const int64_t ntiles = (n16 + bnt - 1) / bnt;
        const int64_t tiles = mtiles * ntiles;
        const int64_t rounds = (tiles + num_cu - 1) / num_cu;
        const double mfma = ks * 3.0 * 2.0 * s.mt * s.nt * 16.0;
        const double ldsr = ks * 3.0 * 8.0 * (s.mt + s.nt) * 1024.0 / 256.0 * 1.3;
        const double stage = ks * ((bm + 2) + 3.0 * bnt * 16) * 64.0 / 24.0;      // ~24 B/clk/CU from L2
        const double item = std::max(std::max(mfma, ldsr), stage) + 500.0;
        const double tile = n_panels * item + 2000.0 + 30.0 * s.mt * s.nt;
        const double cost = (double)rounds * tile;
        if (cost < best_cost) {
            best_cost = cost;
            best = &s;
        }
    }
    return best;
}

// panel width of a layer: RS_H16_PANEL = "64" / "32" forces it for every tiled layer, "l:w;l:w" per layer
int panel_ks(const Hooks& hooks, int layer_index) {
    if (const char* e = hooks.h16_panel; *e) {
        if (!strchr(e, ':')) return atoi(e) == 64 ? 2 : 1;
        int l, w;
        for (const char* q = e; q && *q; q = strchr(q, ';') ? strchr(q, ';') + 1 : nullptr)
            if (sscanf(q, "%d:%d", &l, &w) == 2 && l == layer_index) return w == 64 ? 2 : 1;
    }
    return 1;
}

}  // namespace

int conv_h16_max_bn() { return 256; }
int conv_h16_num_shapes() { return 2 * kNumShapes; }
bool conv_h16_shape_ok(const ConvLayerDev&, int k) {
    if (k < 0 || k >= 2 * kNumShapes) return false;
    const int ks = k / kNumShapes + 1;
    const Shape& s = kShapes[k % kNumShapes];
    return s.fn[ks - 1][0] != nullptr && lds_bytes(s, ks) <= 160 * 1024;
}

int launch_conv_h16(const ConvLayerDev& L, const void* d_x, void* d_y, const int32_t* d_len, int B, int P_in,
                    int layer_index, int num_cu, const void* d_zero, bool f16, int check_dead, hipStream_t st,
                    int* bm_out, int* bn_out) {
    const int64_t rows64 = (int64_t)B * P_in;
    if (rows64 > 0x7fffffff) {
        set_error("conv_h16: batch too large (%lld rows)", (long long)rows64);
        return RS_ERR_ARG;
    }
    const int n16 = round_up(L.c_out, 16) / 16;
    int ks = panel_ks(*L.hooks, layer_index);
    int n_panels = (L.plan.nch + ks - 1) / ks;                      // plan.nch counts 32-channel panels
    const Shape* s = choose_shape(rows64, n16, n_panels, num_cu, ks);
    if (!s && ks == 2) {
        ks = 1;
        n_panels = L.plan.nch;
        s = choose_shape(rows64, n16, n_panels, num_cu, ks);
    }
    if (const char* force = L.hooks->force_h16; *force) {         // tuning aid: "layer:wm,wn,mt,nt;..."
        int l, wm, wn, mt, nt;
        for (const char* q = force; q && *q; q = strchr(q, ';') ? strchr(q, ';') + 1 : nullptr)
            if (sscanf(q, "%d:%d,%d,%d,%d", &l, &wm, &wn, &mt, &nt) == 5 && l == layer_index)
                for (int k = 0; k < kNumShapes; ++k)
                    if (kShapes[k].wm == wm && kShapes[k].wn == wn && kShapes[k].mt == mt && kShapes[k].nt == nt &&
                        lds_bytes(kShapes[k], ks) <= 160 * 1024 && kShapes[k].fn[ks - 1][0])
                        s = &kShapes[k];
    }
    if (const int k = tuned_shape(L, rows64); k >= 0 && conv_h16_shape_ok(L, k)) {
        s = &kShapes[k % kNumShapes];
        ks = k / kNumShapes + 1;
        n_panels = (L.plan.nch + ks - 1) / ks;
    }
    if (!s) {
        set_error("conv_h16: no tile shape fits");
        return RS_ERR_ARG;
    }
    const int BM = s->wm * 16 * s->mt, BN = s->wn * 16 * s->nt;
    ConvHArgs a;
    a.x = static_cast<const unsigned short*>(d_x);
    a.w = static_cast<const unsigned short*>(L.d_w);
    a.bias = L.d_bias;
    a.y = static_cast<unsigned short*>(d_y);
    a.len = d_len;
    a.zero = static_cast<const unsigned short*>(d_zero);
    const int64_t xb = rows64 * L.cp_in * 2, wb = (int64_t)L.plan.nch * 3 * L.plan.n_alloc * 32 * 2;
    if (xb >= 0x80000000LL || wb >= 0x80000000LL) {
        set_error("conv_h16: activation buffer exceeds the 2 GiB buffer-load window, split the batch");
        return RS_ERR_ARG;
    }
    a.x_bytes = (unsigned)xb;
    a.w_bytes = (unsigned)wb;
    a.rows_in = (int)rows64;
    a.P_out = P_in / 2;
    a.inv_P_out = 1.0f / (float)a.P_out;
    a.cp_in = L.cp_in;
    a.cp_out = L.cp_out;
    a.n_panels = n_panels;
    a.n_alloc = L.plan.n_alloc;
    a.shift_out = layer_index + 1;
    const int n_mtiles = (a.rows_in + BM - 1) / BM, n_ntiles = (n16 * 16 + BN - 1) / BN;
    const int64_t tiles = (int64_t)n_mtiles * n_ntiles;
    const unsigned grid = (unsigned)std::min<int64_t>(tiles, num_cu);
    a.walk = plan_walk(n_mtiles, n_ntiles, grid, num_cu, BM, 3.0 * BN, check_dead, !L.hooks->no_rect_order);
    KernelFn fn = s->fn[ks - 1][f16 ? 1 : 0];
    RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
    a.stamps = nullptr;
#ifdef RS_ITEM_STAMPS
    static unsigned long long* d_stamps = nullptr;
    if (!d_stamps) RS_HIP(hipMalloc(&d_stamps, 4 * 8 * 8 * 8));
    a.stamps = d_stamps;
#endif
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kThreads), lds_bytes(*s, ks), st, a);
    RS_HIP(hipGetLastError());
#ifdef RS_ITEM_STAMPS
    {
        RS_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> hp(4 * 8 * 8);
        RS_HIP(hipMemcpy(hp.data(), d_stamps, hp.size() * 8, hipMemcpyDeviceToHost));
        for (int w = 0; w < 8; w += 4) {
            const unsigned long long* q = &hp[w * 8];
            const double n = (double)q[5];
            fprintf(stderr, "[item-stamps] layer %d tile %dx%d wave %d items %.0f total %.0f cyc; per item: bookkeeping+prefetch %.0f | "
                    "frag+mfma %.0f | epilogue %.0f | vmcnt+lds store %.0f | barrier %.0f\n",
                    layer_index, BM, BN, w, n, (double)q[6], q[0] / n, q[1] / n, q[2] / n, q[3] / n, q[4] / n);
        }
    }
#endif
    if (bm_out) *bm_out = BM;
    if (bn_out) *bn_out = BN;
    return RS_OK;
}

}  // namespace rs
