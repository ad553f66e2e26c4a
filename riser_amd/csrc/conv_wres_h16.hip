// K3v (16-bit, weights resident): the narrow tiled layers of the 16-bit modes - few input channels, all output channels in
// one tile - with the WHOLE weight tensor of the layer resident in LDS.
//
//   Conv1d(C_in -> C_out, k=3, 'same', bias) -> ReLU -> MaxPool1d(2,2)   (riser/nets/cnn.py:52-65)
//
// conv_ring_h16.hip streams one weight slab per (panel, tap) sub-stage because the wide layers' weights do not fit; a
// sub-stage of a narrow layer is then 20-30 MFMAs per wave between two barriers, and a tile is mostly barrier, DMA wait and
// epilogue (layer 3 of the shipped net: matrix pipe 32 % busy, profiles/r03_pmc_mfma_busy_bf16x3.json).  Here:
//   * the layer's weights ([panel][tap][n][64 x 16 bit], the ring packing) are DMA'd to LDS ONCE per workgroup; the tile
//     covers every output channel, so all tiles of the launch share them;
//   * a work item is a STAGE = (tile, panel): all three taps, 60-90 MFMAs per wave, between two barriers; the next stage's
//     activation slab ((BM + 8) rows x 128 B, one panel) is in flight by LDS-DMA during the whole stage;
//   * one 8-wave workgroup per CU (the weights, two slabs and the constants take 110-155 KB): 256 registers per lane, so
//     the fragments of the next tap are read while the current tap's MFMAs issue;
//   * the stage ends with a COUNTED wait: the slab pieces were issued before the epilogue's stores, so vmcnt(#stores)
//     publishes the slab without waiting for the stores to be acknowledged.
// Same MFMA sequence per accumulator as conv_ring_h16.hip (panel -> tap -> hi*hi, lo*hi, hi*lo resp. h0*h0, h1*h1), same
// epilogue arithmetic: results are bit-identical to the ring kernel (RS_H16_WRES=0 selects it; tests compare the two).
#include "common.hpp"
#include "tile_walk.hpp"

#include <algorithm>
#include <utility>

namespace rs {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 512;
constexpr int kRowB = 128;                      // bytes of an LDS row (one panel of one position / output channel)
constexpr int kPieceRows = 1024 / kRowB;        // rows per DMA piece (one wave instruction)
constexpr unsigned kOob = 0x80000000u;

struct WresArgs {
    const unsigned short* x;     // [rows_in][cpx_in]
    const unsigned short* w;     // ring packing [panel][tap][n_alloc][64]
    const float* bias;           // [n_alloc]
    float unscale;               // as RingArgs::unscale
    unsigned short* y;           // [rows_in / 2][cpx_out]
    const int32_t* len;
    unsigned* sat;               // half precision: the model's overflow flag (common.hpp: f16_overflow_bits), else null
    unsigned x_bytes, w_bytes, y_bytes;
    int rows_in;
    int P_out;
    float inv_P_out;
    int cpx_in, cpx_out;
    int cols_out;                // logical output columns of a row (plain: cpx_out; x3: 32 x panels)
    int n_panels, n_alloc, n_reads, shift_out;
    WalkArgs walk;               // n_ntiles == 1
};

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ float widen16(unsigned short u) {
    if constexpr (F16)
        return (float)__builtin_bit_cast(_Float16, u);
    else
        return __builtin_bit_cast(float, (unsigned)u << 16);
}
template <bool F16>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    if constexpr (F16)
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    else
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float swap_pair(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}
template <bool X3>
__device__ __forceinline__ int phys_col(int c) {
    return X3 ? ((c >> 5) << 6) + (c & 31) : c;
}
// one LDS-DMA piece: lane l's 16 bytes at rsrc + voff land at LDS byte lds_addr + 16 l (zeros if voff is out of range)
__device__ __forceinline__ void dma_piece(unsigned voff, const __amdgpu_buffer_rsrc_t rsrc, unsigned lds_addr) {
    const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(voff), "s"(m0v), "s"(rsrc) : "memory");
}

template <int MT, int NT, bool F16, bool X3>
__global__ __launch_bounds__(kThreads, 1) void conv_wres_h16_kernel(const WresArgs a) {
    constexpr int BM = 8 * 16 * MT;                                 // 8 waves stacked along the rows
    constexpr int BN = 16 * NT;
    constexpr int XROWS = BM + 8;                                   // slab rows: positions m0 - 1 .. m0 + BM + 6
    constexpr int XS = XROWS * kRowB;
    constexpr int WS = BN * kRowB;                                  // one (panel, tap) weight slab
    constexpr int XP = XROWS / kPieceRows;                          // DMA pieces per activation slab
    constexpr int XPW = (XP + 7) / 8;                               // ... per wave
    constexpr int WP = BN / kPieceRows;
    constexpr int CONST_OFF = 2 * XS;                               // LDS: [X slab 0][X slab 1][bias 1 KiB][len 2 x 1 KiB][weights]
    constexpr int W_OFF = CONST_OFF + 3072;
    static_assert(BN % kPieceRows == 0 && XROWS % kPieceRows == 0 && BN <= 256, "whole pieces; one piece of bias values");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w), 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, (unsigned)a.n_alloc * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.len), 0, (unsigned)a.n_reads * 4u, 0x00020000);

    // DMA source maps (conv_ring_h16.hip): lane l of a piece = row l >> 3, physical 16-byte slot l & 7, which holds the
    // logical slot (l & 7) ^ (row & 7)
    const int prow = lane >> 3, lslot = (lane & 7) ^ prow;
    const unsigned x_lane = (unsigned)(prow * a.cpx_in + 8 * lslot) * 2u;
    const unsigned w_lane = (unsigned)(prow * 64 + 8 * lslot) * 2u;

    // ---- tile walk: row tiles only, dead-tile elimination as in the ring kernel ----------------------------------
    const int tiles = a.walk.q_total;
    const int P_in_ = 2 * a.P_out;
    TileWalk walk;
    auto next_live = [&]() {
        int q = walk.next_index(a.walk);
        while (q < tiles) {
            int mi, nt_;
            if (walk_tile(a.walk, q, mi, nt_)) {
                if (!a.walk.check_dead) break;
                const int tm0 = mi * BM;
                const int b = tm0 / P_in_;
                const int t0 = tm0 - b * P_in_;
                if (!(t0 + BM <= P_in_ && t0 >= (as_const_len(a.len)[b] >> (a.shift_out - 1)))) break;
                const int pieces_per_row = max(a.cols_out, BN) / 8;
                for (int f = threadIdx.x; f < (BM / 2) * pieces_per_row; f += blockDim.x) {
                    const int rr = f / pieces_per_row, cc = (f - rr * pieces_per_row) * 8;
                    const int orow = (tm0 >> 1) + rr;
                    if (2 * orow < a.rows_in && cc < a.cols_out) {
                        unsigned short* dst = a.y + (int64_t)orow * a.cpx_out + phys_col<X3>(cc);
                        *reinterpret_cast<uint4*>(dst) = make_uint4(0u, 0u, 0u, 0u);
                        if constexpr (X3) *reinterpret_cast<uint4*>(dst + 32) = make_uint4(0u, 0u, 0u, 0u);
                    }
                }
            }
            q = walk.next_index(a.walk);
        }
        return q;
    };
    auto tile_m0 = [&](int q) {
        int mi, nt_;
        walk_tile(a.walk, q, mi, nt_);
        return mi * BM;
    };
    int m0;
    {
        const int o = next_live();
        if (o >= tiles) return;
        m0 = tile_m0(o);
    }

    // ---- DMA issue: every wave issues every 8th piece of a slab, a static count per call -----------------------------
    auto issue_x = [&](int mm0, int p, bool live, int xb) {
#pragma unroll
        for (int idx = 0; idx < XPW; ++idx) {
            const int k = min(wave + 8 * idx, XP - 1);
            const unsigned off = (unsigned)(((mm0 - 1 + k * kPieceRows) * a.cpx_in + p * 64) * 2) + x_lane;
            dma_piece(live ? off : kOob, rs_x, (unsigned)(xb * XS + k * 1024));
        }
    };
    auto issue_lens = [&](int mm0, int cb) {
        const int b0 = (mm0 >> 1) / a.P_out;
        dma_piece((unsigned)(b0 * 4 + lane * 16), rs_l, (unsigned)(CONST_OFF + 1024 + cb * 1024));
    };

    // ---- prologue: all weights, the bias values, the first slab and the first tile's lengths ---------------------------
    for (int k = wave; k < a.n_panels * 3 * WP; k += 8) {
        const int slab = k / WP, kk = k - slab * WP;                // slab = panel * 3 + tap
        const unsigned off = (unsigned)(((slab * a.n_alloc + kk * kPieceRows) * 64) * 2) + w_lane;
        dma_piece(off, rs_w, (unsigned)(W_OFF + slab * WS + kk * 1024));
    }
    dma_piece((unsigned)(lane * 16), rs_b, (unsigned)CONST_OFF);
    issue_x(m0, 0, true, 0);
    issue_lens(m0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- fragment read addresses (bytes in LDS) ------------------------------------------------------------------
    unsigned a_rd[3][2], b_rd[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int R = wave * 16 * MT + r + tap;
            a_rd[tap][h] = (unsigned)(R * kRowB + (((4 * h + g) ^ (R & 7)) << 4));
        }
        b_rd[h] = (unsigned)(W_OFF + r * kRowB + (((4 * h + g) ^ (r & 7)) << 4));
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- epilogue (conv_ring_h16.hip's: bias + ReLU + MaxPool in registers, packed rows through a wave-private LDS image,
    // 16-byte stores); the image lives in the slab the tile has just finished with ------------------------------------
    constexpr int PW = X3 ? 4 : 2;
    constexpr int PITCH = NT * PW * 16 + 16;
    constexpr int NPIECE = 8 * NT * PW;
    constexpr int NSTORE = MT * ((NPIECE + 63) / 64) + (X3 ? MT : 0);   // vector stores of one epilogue per wave
    static_assert(8 * (8 * PITCH) <= XS, "epilogue scratch fits the activation slab");
    auto epilogue = [&](int mm0, int cb, int xb) {
        const float* lbias = reinterpret_cast<const float*>(lds + CONST_OFF);
        const int* llen = reinterpret_cast<const int*>(lds + CONST_OFF + 1024 + cb * 1024);
        unsigned char* scr = lds + xb * XS + wave * (8 * PITCH);
        float bias[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bias[j] = lbias[j * 16 + r];
        const int pr0 = mm0 >> 1;
        const int b0 = pr0 / a.P_out;
        const int p0 = pr0 - b0 * a.P_out;
        const bool odd = r & 1;
        unsigned sat = 0u;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int orow0 = (mm0 + (wave * MT + i) * 16) >> 1;
            unsigned keep;
            {
                const int t = p0 + (orow0 + 2 * g + (odd ? 1 : 0) - pr0);
                const int e = (int)(((float)t + 0.5f) * a.inv_P_out);
                keep = t - e * a.P_out < (llen[e] >> a.shift_out) ? ~0u : 0u;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float us = a.unscale;
                const float v0 = fmaxf(fmaxf(fmaf(acc[i][j][0], us, bias[j]), fmaf(acc[i][j][1], us, bias[j])), 0.0f);
                const float v1 = fmaxf(fmaxf(fmaf(acc[i][j][2], us, bias[j]), fmaf(acc[i][j][3], us, bias[j])), 0.0f);
                const float got = swap_pair(odd ? v0 : v1);
                const float ca = odd ? got : v0, cb_ = odd ? v1 : got;
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
                const int col = 16 * jj + 8 * (part & 1);
                const int elem = phys_col<X3>(col) + (X3 ? 32 * (part >> 1) : 0);
                const bool ok = qi < NPIECE && 2 * orow < a.rows_in && col < a.cols_out;
                const u32x4 v = *reinterpret_cast<const u32x4*>(scr + row8 * PITCH + w * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_y, ok ? (unsigned)(orow * a.cpx_out + elem) * 2u : kOob, 0, 0);
            }
            if constexpr (X3) {
                // the slots between the last computed 16-column group and the end of its 32-slot panel: zeros (always
                // issued, out of range when there are none, so the number of stores per epilogue is static)
                const int row8 = lane >> 2, part = lane & 3;
                const int orow = orow0 + row8;
                const int col = BN + 8 * (part & 1);
                const int elem = phys_col<X3>(col) + 32 * (part >> 1);
                const bool ok = lane < 32 && 2 * orow < a.rows_in && col < a.cols_out;
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){0u, 0u, 0u, 0u}, rs_y,
                                                       ok ? (unsigned)(orow * a.cpx_out + elem) * 2u : kOob, 0, 0);
            }
        }
        if constexpr (F16) raise_saturated(a.sat, sat);        // (one more vector-memory operation: the counted waits only get stricter)
    };

    // ---- the stage loop ----------------------------------------------------------------------------------------------
    int p = 0, xb = 0, cb = 0;
    while (true) {
        // the stage after this one: the tile's next panel, or panel 0 of the workgroup's next live tile
        int nm0 = m0, np = p + 1;
        bool nlive = true;
        if (np == a.n_panels) {
            const int o = next_live();
            np = 0;
            nlive = o < tiles;
            if (nlive) nm0 = tile_m0(o);
        }
        const bool tile_end = p == a.n_panels - 1;
        issue_x(nm0, np, nlive, xb ^ 1);                              // in flight for the whole stage
        if (tile_end) issue_lens(nm0, cb ^ 1);                        // (out of range past the last tile: harmless zeros)

        const unsigned xoff = (unsigned)(xb * XS), woff = (unsigned)(p * 3 * WS);
        u32x4 fa[2][MT][2], fb[2][NT][2];                           // [buffer][tile][half]: this tap's / the next tap's fragments
        auto read_frags = [&](int tap, int buf) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    fb[buf][j][h] = *reinterpret_cast<const u32x4*>(lds + b_rd[h] + woff + tap * WS + j * 16 * kRowB);
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    fa[buf][i][h] = *reinterpret_cast<const u32x4*>(lds + a_rd[tap][h] + xoff + i * 16 * kRowB);
            }
        };
        read_frags(0, 0);
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int cur = tap & 1;
            if (tap < 2) read_frags(tap + 1, cur ^ 1);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<F16>(fa[cur][i][0], fb[cur][j][0], acc[i][j]);      // hi * hi (h0 * h0)
            if constexpr (X3) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<F16>(fa[cur][i][1], fb[cur][j][0], acc[i][j]);  // x lo * w hi
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<F16>(fa[cur][i][0], fb[cur][j][1], acc[i][j]);  // x hi * w lo
            } else {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<F16>(fa[cur][i][1], fb[cur][j][1], acc[i][j]);  // h1 * h1
            }
        }
        if (tile_end) {
            // every wave has read its fragments of this slab before any wave parks its output rows in it
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            epilogue(m0, cb, xb);
            cb ^= 1;
            // the next slab's pieces were issued BEFORE the epilogue's NSTORE stores: the counted wait publishes them
            // without waiting for the stores to be acknowledged
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NSTORE) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (!nlive) break;
        m0 = nm0;
        p = np;
        xb ^= 1;
    }
}

using KernelFn = void (*)(const WresArgs);
struct Shape {
    int mt, nt;
    KernelFn fn[2][2];     // [plain, x3][bf16, f16]
};
#define RS_SHAPE(MT, NT)                                                                                       \
    {MT, NT,                                                                                                   \
     {{conv_wres_h16_kernel<MT, NT, false, false>, conv_wres_h16_kernel<MT, NT, true, false>},                \
      {conv_wres_h16_kernel<MT, NT, false, true>, conv_wres_h16_kernel<MT, NT, true, true>}}}
const Shape kShapes[] = {RS_SHAPE(2, 2), RS_SHAPE(2, 3), RS_SHAPE(2, 4), RS_SHAPE(2, 5), RS_SHAPE(2, 6), RS_SHAPE(2, 7)};
#undef RS_SHAPE

size_t lds_bytes(int bm, int bn, int n_panels) {
    return (size_t)2 * (bm + 8) * kRowB + 3072 + (size_t)n_panels * 3 * bn * kRowB;
}

const Shape* pick_shape(const ConvLayerDev& L, bool x3) {
    // split precision: the tile must cover all 32 slots of the last panel but 16 (the ring kernel's rule: one zero-filled
    // 16-column group at most behind the computed ones)
    const int n16 = round_up(L.c_out, 16) / 16;
    for (const Shape& s : kShapes)
        if (s.nt == n16 && lds_bytes(8 * 16 * s.mt, 16 * s.nt, L.ring_panels) <= 160 * 1024) return &s;
    return nullptr;
}

}  // namespace

bool conv_wres_h16_ok(const ConvLayerDev& L, bool x3) {
    return L.d_w2 && L.ring_panels >= 1 && !L.ring_tail && L.hooks->h16_wres && pick_shape(L, x3) != nullptr;   // (a merged tail panel is the ring and thin kernels' packing)
}

int launch_conv_wres_h16(const ConvLayerDev& L, const void* d_x, void* d_y, const int32_t* d_len, int B, int P_in,
                         int layer_index, int num_cu, bool f16, bool x3, int check_dead, hipStream_t st, int* bm_out,
                         int* bn_out) {
    const Shape* s = pick_shape(L, x3);
    if (!s || !L.d_w2) {
        set_error("conv_wres_h16: layer %d does not fit the weights-resident kernel", layer_index);
        return RS_ERR_ARG;
    }
    const int64_t rows64 = (int64_t)B * P_in;
    const int n_panels = L.ring_panels;
    const int64_t xb = rows64 * L.cp_in * 2, wb = (int64_t)n_panels * 3 * L.plan.n_alloc * 64 * 2, yb = rows64 / 2 * L.cp_out * 2;
    if (rows64 > 0x7fffffff || xb >= 0x80000000LL || wb >= 0x80000000LL || yb >= 0x80000000LL) {
        set_error("conv_wres_h16: batch too large for the 2 GiB buffer window, split it");
        return RS_ERR_ARG;
    }
    const int BM = 8 * 16 * s->mt, BN = 16 * s->nt;
    WresArgs a;
    a.x = static_cast<const unsigned short*>(d_x);
    a.w = static_cast<const unsigned short*>(L.d_w2);
    a.bias = L.d_bias;
    a.unscale = L.w_unscale;
    a.y = static_cast<unsigned short*>(d_y);
    a.len = d_len;
    a.sat = f16 ? L.d_sat : nullptr;
    a.x_bytes = (unsigned)xb;
    a.w_bytes = (unsigned)wb;
    a.y_bytes = (unsigned)yb;
    a.rows_in = (int)rows64;
    a.P_out = P_in / 2;
    a.inv_P_out = 1.0f / (float)a.P_out;
    a.cpx_in = L.cp_in;
    a.cpx_out = L.cp_out;
    a.cols_out = x3 ? L.cp_out / 2 : L.cp_out;
    a.n_panels = n_panels;
    a.n_alloc = L.plan.n_alloc;
    a.n_reads = B;
    a.shift_out = layer_index + 1;
    if (a.cols_out - BN > 16 || a.cols_out < BN - 15) {
        set_error("conv_wres_h16: tile of %d columns does not match the %d column slots of layer %d", BN, a.cols_out, layer_index);
        return RS_ERR_ARG;
    }
    const int n_mtiles = (a.rows_in + BM - 1) / BM;
    const unsigned grid = (unsigned)std::min<int64_t>(n_mtiles, num_cu);
    a.walk = plan_walk(n_mtiles, 1, grid, num_cu, BM, 3.0 * BN, check_dead, false);
    KernelFn fn = s->fn[x3 ? 1 : 0][f16 ? 1 : 0];
    const size_t lds = lds_bytes(BM, BN, n_panels);
    RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kThreads), lds, st, a);
    RS_HIP(hipGetLastError());
    if (bm_out) *bm_out = BM;
    if (bn_out) *bn_out = BN;
    return RS_OK;
}

}  // namespace rs
