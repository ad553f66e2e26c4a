// K3t (split precision, THIN launches): the tiled layers of an RS_BF16X3 / RS_F16X3 model when a launch has only a few rows.
//
//   Conv1d(C_in -> C_out, k=3, 'same', bias) -> ReLU -> MaxPool1d(2,2)   (riser/nets/cnn.py:52-65)
//
// Model.classify is batch 1 in the reference's own loop (riser/model.py:22-28, riser/control.py:63-69), a chunk client delivers
// ~20 reads.  The ring kernel (conv_ring_h16.hip) makes one sub-stage - a barrier, a counted wait - out of every (panel, tap):
// layer 11 is 108 of them per tile, and with a handful of live rows each is as long as its staging round trip (~500 cycles with
// the 64-row shapes of round 6) whatever it computes: 32 us at batch 1 for 12 MFLOP.
//
// Same lowering, same LDS image, same MFMA order - three stages of a WHOLE PANEL each instead of a ring of taps:
//   * tile = 64 conv rows x 32 output channels, eight waves (4 x 2), one 16 x 16 accumulator per wave;
//   * a stage holds one 32-channel panel: the activation slab (72 rows x 128 bytes, [hi x 32 | lo x 32]) and the weight rows of
//     all three taps (3 x 32 x 128 bytes) - 21 LDS-DMA pieces, three per wave; panel p + 2 is issued while panel p is computed,
//     so a panel's pieces have two iterations to land; ONE barrier and one counted wait per panel: layer 11 is 36 iterations
//     of ~350 cycles;
//   * per accumulator the MFMAs run panel by panel, tap by tap, hi*hi, lo*hi, hi*lo - the ring kernel's order - and the epilogue
//     is its arithmetic: BIT-IDENTICAL results (tests/test_gpu_small.py), a read alone equals its row of a 512-read batch.
// One workgroup per tile, no tile walk (a thin launch is tens to hundreds of tiles).  The launch planner (api.hip) takes this
// kernel where its estimate beats the ring's.
#include "common.hpp"

#include <algorithm>
#include <cmath>

namespace rs {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 512;
constexpr int kBM = 64, kBN = 32;
constexpr int kXRows = kBM + 8;                 // positions m0 - 1 .. m0 + kBM + 6
constexpr int kXPieces = kXRows / 8;            // 9 DMA pieces of 8 rows
constexpr int kWPieces = 3 * kBN / 8;           // 12: the three taps' weight rows
constexpr int kPieces = kXPieces + kWPieces;    // 21 per panel
constexpr int kPerWave = (kPieces + 7) / 8;     // 3 per wave (the last slots re-issue the last piece)
constexpr int kStage = (kXRows + 3 * kBN) * 128;
constexpr int kStages = 3;
constexpr unsigned kOob = 0x80000000u;

struct ThinArgs {
    const unsigned short* x;     // [rows_in][cpx_in]  panels [hi x 32 | lo x 32]
    const unsigned short* w;     // ring packing [panel][tap][n_alloc][64]
    const float* bias;           // [n_alloc]
    float unscale;
    unsigned short* y;           // [rows_in / 2][cpx_out]
    const int32_t* len;          // per block
    unsigned* sat;               // half precision: the model's overflow flag, else null
    unsigned x_bytes, w_bytes, y_bytes;
    int rows_in, P_out;
    int cpx_in, cpx_out;
    int n_panels, n_alloc, n_blocks, shift_out;
    int n_ntiles;
    int tail;                    // the last panel is the merged tail: its three taps are one K step (conv_ring_h16.hip: TAIL)
};

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
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
__device__ __forceinline__ float swap_pair(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}
// one LDS-DMA piece: lane l's 16 bytes at rsrc + voff land at LDS byte lds_addr + 16 l (zeros if voff is out of range)
__device__ __forceinline__ void dma_piece(unsigned voff, const __amdgpu_buffer_rsrc_t rsrc, unsigned lds_addr) {
    const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                 :: "v"(voff), "s"(m0v), "s"(rsrc) : "memory");
}

template <bool F16>
__global__ __launch_bounds__(kThreads) void conv_thin_h16_kernel(const ThinArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int r = lane & 15, g = lane >> 4;
    const int mt = blockIdx.x / a.n_ntiles, nt = blockIdx.x - mt * a.n_ntiles;
    const int m0 = mt * kBM, n0 = nt * kBN;
    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w), 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);

    // DMA source maps (conv_ring_h16.hip): lane l of a piece = row l >> 3 of the piece, physical 16-byte slot l & 7 = logical
    // slot ^ (row & 7): conflict-free fragment reads for all three tap shifts
    const int prow = lane >> 3, lslot = (lane & 7) ^ prow;
    const unsigned x_lane = (unsigned)(prow * a.cpx_in + 8 * lslot) * 2u;
    const unsigned w_lane = (unsigned)(prow * 64 + 8 * lslot) * 2u;
    // this wave's pieces of a panel: k = wave + 8 j; k < 9: activation rows 8 k ..; else weight rows 8 (k - 9) .. of the
    // [tap][32] block of the panel's weights (tap = (k - 9) / 4)
    auto issue_panel = [&](int p, int stage) {
        const bool live = p < a.n_panels;
#pragma unroll
        for (int j = 0; j < kPerWave; ++j) {
            const int k = min(wave + 8 * j, kPieces - 1);
            if (k < kXPieces) {
                const int row0 = m0 - 1 + 8 * k;
                // rows before the buffer (row -1 of the first tile) and past its end: out of range = zeros ('same' padding at the
                // batch edges; inside the batch the producer wrote a zero row behind every read)
                const unsigned off = (unsigned)((row0 * a.cpx_in + p * 64) * 2) + x_lane;
                const bool ok = live && row0 + prow >= 0 && row0 + prow < a.rows_in;
                dma_piece(ok ? off : kOob, rs_x, (unsigned)(stage * kStage + k * 1024));
            } else {
                const int wk = k - kXPieces, tap = wk >> 2, nrow = (wk & 3) * 8;
                const unsigned off = (unsigned)((((p * 3 + tap) * a.n_alloc + n0 + nrow) * 64) * 2) + w_lane;
                dma_piece(live ? off : kOob, rs_w, (unsigned)(stage * kStage + kXRows * 128 + wk * 1024));
            }
        }
    };
    // fragment read addresses inside a stage
    unsigned a_rd[3][2], b_rd[3][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int R = wm * 16 + r + tap;
            a_rd[tap][h] = (unsigned)(R * 128 + (((4 * h + g) ^ (R & 7)) << 4));
            const int N = wn * 16 + r;
            b_rd[tap][h] = (unsigned)(kXRows * 128 + (tap * kBN + N) * 128 + (((4 * h + g) ^ (N & 7)) << 4));
        }

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    issue_panel(0, 0);
    issue_panel(1, 1);
    int stage = 0;
    for (int p = 0; p < a.n_panels; ++p) {
        // panel p has landed when at most the pieces of panel p + 1 are outstanding; the barrier also says that every wave is
        // done with panel p - 1, whose stage panel p + 2 now overwrites
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kPerWave) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        const int nxt = stage == 0 ? 2 : stage - 1;                 // (stage + 2) % 3
        issue_panel(p + 2, nxt);
        const unsigned so = (unsigned)(stage * kStage);
        if (a.tail && p == a.n_panels - 1) {
            // the merged tail panel (its stage holds the panel's activation slab as usual and the merged weight slab in the
            // tap-0 rows): K group g of the fragment = tap g's first eight channel slots, i.e. slab row r + g, 16-byte slot 0
            // of the hi / lo half (group 3: zero weights, any row); one K step, the three products in the usual order
            const int R = wm * 16 + r + (g < 2 ? g : 2);
            const u32x4 th = *reinterpret_cast<const u32x4*>(lds + so + R * 128 + ((0 ^ (R & 7)) << 4));
            const u32x4 tl = *reinterpret_cast<const u32x4*>(lds + so + R * 128 + ((4 ^ (R & 7)) << 4));
            const u32x4 uh = *reinterpret_cast<const u32x4*>(lds + so + b_rd[0][0]);
            const u32x4 ul = *reinterpret_cast<const u32x4*>(lds + so + b_rd[0][1]);
            acc = mfma16<F16>(th, uh, acc);
            acc = mfma16<F16>(tl, uh, acc);
            acc = mfma16<F16>(th, ul, acc);
        } else {
            u32x4 fa[3][2], fb[3][2];
#pragma unroll
            for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    fa[tap][h] = *reinterpret_cast<const u32x4*>(lds + so + a_rd[tap][h]);
                    fb[tap][h] = *reinterpret_cast<const u32x4*>(lds + so + b_rd[tap][h]);
                }
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                acc = mfma16<F16>(fa[tap][0], fb[tap][0], acc);          // hi * hi
                acc = mfma16<F16>(fa[tap][1], fb[tap][0], acc);          // lo * hi
                acc = mfma16<F16>(fa[tap][0], fb[tap][1], acc);          // hi * lo
            }
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
    // every piece in flight targets a stage nobody reads any more; the epilogue's image lives in stage 0: wait for all of them
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue (conv_ring_h16.hip's, MT = NT = 1): bias + ReLU + MaxPool in registers, a wave-private image of 8 pooled
    // rows x 16 channels x (hi, lo) in LDS, out in 16-byte pieces
    constexpr int PW = 4, PITCH = PW * 16 + 16, NPIECE = 8 * PW;
    unsigned char* scr = lds + wave * (8 * PITCH);
    const float bias = a.bias[n0 + wn * 16 + r];
    const float us = a.unscale;
    const bool odd = r & 1;
    const int orow0 = (m0 + wm * 16) >> 1;                           // first of the wave's 8 pooled rows
    unsigned keep;
    {
        const int orow = orow0 + 2 * g + (odd ? 1 : 0);
        const int b = orow / a.P_out, t = orow - b * a.P_out;
        keep = (b < a.n_blocks && t < (as_const_len(a.len)[min(b, a.n_blocks - 1)] >> a.shift_out)) ? ~0u : 0u;
    }
    const float v0 = fmaxf(fmaxf(fmaf(acc[0], us, bias), fmaf(acc[1], us, bias)), 0.0f);
    const float v1 = fmaxf(fmaxf(fmaf(acc[2], us, bias), fmaf(acc[3], us, bias)), 0.0f);
    const float got = swap_pair(odd ? v0 : v1);
    const float ca = odd ? got : v0, cb_ = odd ? v1 : got;          // channels (r & ~1, r | 1) of row 2 g + odd
    const unsigned hi = pack2<F16>(ca, cb_);
    if constexpr (F16) raise_saturated(a.sat, f16_overflow_bits(hi));
    unsigned char* dst = scr + (2 * g + (odd ? 1 : 0)) * PITCH + (r & ~1) * 2;
    *reinterpret_cast<unsigned*>(dst) = hi & keep;
    *reinterpret_cast<unsigned*>(dst + 32) = keep &
        pack2<F16>(ca - widen16<F16>((unsigned short)(hi & 0xffffu)), cb_ - widen16<F16>((unsigned short)(hi >> 16)));
    // (LDS operations of one wave execute in order: the image needs no wait between its writes and its reads)
    if (lane < NPIECE) {
        const int row8 = lane / PW, part = lane - row8 * PW;
        const int orow = orow0 + row8;
        const int col = n0 + wn * 16 + 8 * (part & 1);
        const int elem = ((col >> 5) << 6) + (col & 31) + 32 * (part >> 1);
        const bool ok = 2 * orow < a.rows_in;
        const u32x4 v = *reinterpret_cast<const u32x4*>(scr + row8 * PITCH + part * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_y, ok ? (unsigned)(orow * a.cpx_out + elem) * 2u : kOob, 0, 0);
    }
}

}  // namespace

bool conv_thin_h16_ok(const ConvLayerDev& L) { return L.d_w2 != nullptr && !L.f8_in && !L.f8_out && L.ring_panels >= 1; }

int64_t conv_thin_h16_tiles(const ConvLayerDev& L, int64_t rows_in) { return ((rows_in + kBM - 1) / kBM) * (L.cp_out / 2 / kBN); }

// estimate in shader cycles.  A panel of a tile is latency-bound (~450 cycles: barrier, three pieces per wave, nine MFMAs) while
// the launch is a few tiles per CU, and bound by the staging path once the chip is full: 21.5 KB per tile and panel at the
// ~24 B/clk/CU the L2 -> LDS path delivers (measured, layer 11: 14.5 / 15.5 / 17.5 / 26.5 / 56 / 103 us at 54 / 108 / 216 / 432 /
// 864 / 1728 tiles)
double conv_thin_h16_cost(const ConvLayerDev& L, int64_t rows_in, int num_cu) {
    const double tiles = (double)conv_thin_h16_tiles(L, rows_in);
    const double waves_of_tiles = std::ceil(tiles / (2.0 * num_cu));
    const double per_panel = std::max(450.0 * waves_of_tiles, tiles * 900.0 / num_cu);
    return L.ring_panels * per_panel + 6000.0;
}

int launch_conv_thin_h16(const ConvLayerDev& L, const void* d_x, void* d_y, const int32_t* d_len, int B, int P_in,
                         int layer_index, int num_cu, bool f16, hipStream_t st, int* bm_out, int* bn_out) {
    const int64_t rows64 = (int64_t)B * P_in;
    const int64_t xb = rows64 * L.cp_in * 2, wb = (int64_t)L.ring_panels * 3 * L.plan.n_alloc * 64 * 2, yb = rows64 / 2 * L.cp_out * 2;
    if (rows64 > 0x7fffffff || xb >= 0x80000000LL || wb >= 0x80000000LL || yb >= 0x80000000LL) {
        set_error("conv_thin_h16: batch too large for the 2 GiB buffer window, split it");
        return RS_ERR_ARG;
    }
    ThinArgs a;
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
    a.cpx_in = L.cp_in;
    a.cpx_out = L.cp_out;
    a.n_panels = L.ring_panels;
    a.n_alloc = L.plan.n_alloc;
    a.n_blocks = B;
    a.shift_out = layer_index + 1;
    a.tail = L.ring_tail ? 1 : 0;
    a.n_ntiles = L.cp_out / 2 / kBN;                                 // 32 slots per output panel: every one is written
    if (a.n_ntiles * kBN > L.plan.n_alloc) {
        set_error("conv_thin_h16: layer %d: %d columns overhang the weight table", layer_index, a.n_ntiles * kBN);
        return RS_ERR_ARG;
    }
    const int64_t tiles = ((rows64 + kBM - 1) / kBM) * a.n_ntiles;
    if (tiles > 0x7fffffff / 2) {
        set_error("conv_thin_h16: launch too large");
        return RS_ERR_ARG;
    }
    const size_t lds = (size_t)kStages * kStage;
    auto fn = f16 ? conv_thin_h16_kernel<true> : conv_thin_h16_kernel<false>;
    RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(fn, dim3((unsigned)tiles), dim3(kThreads), lds, st, a);
    RS_HIP(hipGetLastError());
    if (bm_out) *bm_out = kBM;
    if (bn_out) *bn_out = kBN;
    (void)num_cu;
    return RS_OK;
}

}  // namespace rs
