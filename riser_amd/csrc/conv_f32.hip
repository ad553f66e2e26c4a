// K3 (fp32): one ConvNet block i >= 1 as an implicit-im2col GEMM on the f32-input MFMA.
//
//   Conv1d(C_in -> C_out, k=3, stride 1, zero 'same' padding, bias) -> ReLU -> MaxPool1d(2,2)
//   (riser/nets/cnn.py:52-65, depth 1)
//
// Data layout (chosen for the MFMA, not inherited from torch's NCL):
//   activations are position-major: X[row][c], row = b * P + t, c padded to a multiple of
//   4; each read owns a slot of P rows (P even, P > L) and every row t >= L_i[b] of its slot
//   is zero.  Row -1 and row B*P are treated as zero by the loader.  Hence the conv's zero
//   padding at both ends of every read is already in the data, and the im2col row of
//   position `row` is the three consecutive input rows row-1, row, row+1.
// GEMM: M = positions (rows), N = output channels, K = 3 * C_in:
//   out[row][n] = sum_{kw, c} X[row - 1 + kw][c] * W[n][c][kw]
// so the pooling pair (2p, 2p+1) is two adjacent M rows = two adjacent accumulator
// registers of one lane in the 16x16 C/D layout (row = 4*(lane>>4) + reg, col = lane&15):
// bias + ReLU + MaxPool run in registers and the pooled row is stored straight into the next
// layer's buffer at row/2 (P_out = P/2), masked to zero beyond len >> (i+1).
//
// Schedule: PERSISTENT workgroups (one per CU, 8 waves = 2 per SIMD) walk the tile list in an
// XCD-aware order; a tile is BM x BN = (WM*16*MT) x (WN*16*NT) with the 8 waves arranged
// WM x WN.  K is cut into chunks of KC input channels; a work item = (tile, chunk).  Per item
// the workgroup needs a (BM+2) x KC slab of X (the +2 halo rows serve all three taps from ONE
// copy) and a BN x 3 x KC slab of packed weights in LDS.  Items are software-pipelined across
// tile boundaries: the global loads of item k+1 are issued into registers BEFORE the MFMAs of
// item k and written to the other LDS buffer after them (one barrier per item), so HBM/L2
// latency, the epilogue's stores and the next tile's prologue all hide under MFMA issue.
// LDS rows are KC+2 floats (= 2 mod 4): the 16 rows x 2 k-groups that a 32-lane half reads
// with ds_read_b32 then fall in 32 distinct banks.
#include "common.hpp"
#include "tile_walk.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace rs {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Diagnostic build only (-DRS_ITEM_STAMPS): per-phase s_memtime sums of one item loop, written to
// ConvArgs::stamps[2048 + ...]; never compiled into the shipped library.
#ifdef RS_ITEM_STAMPS
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

constexpr int kKcMax = 24;                  // bounds the per-thread staging registers

struct ConvArgs {
    const float* x;
    const float* w;        // packed [n_alloc][nch][3][kc], zero rows beyond c_out
    const float* bias;     // [n_alloc]
    float* y;
    const int32_t* len;
    const float* zero;     // >= 16 bytes of zeros (target of masked-off staging loads)
    int rows_in;           // B * P_in
    int P_out;             // P_in / 2
    float inv_P_out;
    int cp_in, cp_out;
    int kc, nch;
    int shift_out;         // valid output rows of read b: len[b] >> shift_out
    WalkArgs walk;         // tile grid, order and dead-tile flag (tile_walk.hpp)
    unsigned long long* stamps;   // diagnostic (RS_CONV_STAMPS=1): per block {memtime, memrealtime} at entry / exit
};

// KCT > 0: the channel chunk is a compile-time constant, so every LDS fragment address in the
// MFMA loop is "per-item base register + immediate" and the loop is fully unrolled (no address
// VALU between MFMAs).  KCT == 0: generic fallback with the chunk taken from ConvArgs.
template <int WM, int WN, int MT, int NT, int KCT>
__global__ __launch_bounds__(WM * WN * 64) void conv_f32_kernel(const ConvArgs a) {
    static_assert(WM * WN == 8 || WM * WN == 4, "8 waves (2 per SIMD) or 4 waves (1 per SIMD)");
    constexpr int kThreads = WM * WN * 64;
    constexpr int BM = WM * 16 * MT;
    constexpr int BN = WN * 16 * NT;
    constexpr int KCB = KCT ? KCT : kKcMax;                    // bound for the staging registers
    constexpr int A_PER = ((BM + 2) * (KCB / 4) + kThreads - 1) / kThreads;
    constexpr int B_PER = (BN * 3 * (KCB / 4) + kThreads - 1) / kThreads;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int r = lane & 15, kq = lane >> 4;

    const int KC = KCT ? KCT : a.kc;
    const int S = KC + 2;
    const int kc4 = KC >> 2;
    const int a_elems = (BM + 2) * S;
    const int buf_elems = a_elems + BN * 3 * S;

    // ---- per-thread staging map (computed once; the kernel is persistent) ---------------------
    int a_lds[A_PER], a_key[A_PER];            // key = row * 8 + c4, or -1 if the unit is unused
    int b_lds[B_PER], b_g[B_PER];              // b_g < 0 if unused
    {
        const int a_units = (BM + 2) * kc4;
#pragma unroll
        for (int u = 0; u < A_PER; ++u) {
            const int f = tid + u * kThreads;
            const int row = f / kc4, c4 = f - row * kc4;
            a_lds[u] = row * S + 4 * c4;
            a_key[u] = f < a_units ? row * 8 + c4 : -1;
        }
        const int b_row_units = 3 * kc4;
        const int b_units = BN * b_row_units;
#pragma unroll
        for (int u = 0; u < B_PER; ++u) {
            const int f = tid + u * kThreads;
            const int n = f / b_row_units, rem = f - n * b_row_units;
            const int kw = rem / kc4, c4 = rem - kw * kc4;
            b_lds[u] = a_elems + (n * 3 + kw) * S + 4 * c4;
            b_g[u] = f < b_units ? n * a.nch * 3 * KC + 4 * rem : -1;
        }
    }

    float4 ra[A_PER], rb[B_PER];
    auto load_item = [&](int m0, int n0, int c, bool live) {
        const int cbase = c * KC;
        const float* xb = a.x + (int64_t)(m0 - 1) * a.cp_in + cbase;
#pragma unroll
        for (int u = 0; u < A_PER; ++u) {
            // UNCONDITIONAL load: out-of-range units read a 16-byte zero page instead of being
            // branched around (a conditional load makes hipcc drain vmcnt(0) before the MFMA
            // block, which serialises the prefetch with the compute)
            const int key = a_key[u];
            const int row = key >> 3, c4 = key & 7;
            const int gr = m0 - 1 + row;
            const bool ok = live && key >= 0 && gr >= 0 && gr < a.rows_in && cbase + 4 * c4 < a.cp_in;
            const float* src = ok ? xb + (int64_t)row * a.cp_in + 4 * c4 : a.zero;
            ra[u] = *reinterpret_cast<const float4*>(src);
        }
        const float* wb = a.w + ((int64_t)n0 * a.nch + c) * (3 * KC);
#pragma unroll
        for (int u = 0; u < B_PER; ++u) {
            const float* src = (live && b_g[u] >= 0) ? wb + b_g[u] : a.zero;
            rb[u] = *reinterpret_cast<const float4*>(src);
        }
    };
    auto store_item = [&](float* buf) {
#pragma unroll
        for (int u = 0; u < A_PER; ++u)
            if (a_key[u] >= 0) {
                float2* d = reinterpret_cast<float2*>(buf + a_lds[u]);
                d[0] = make_float2(ra[u].x, ra[u].y);
                d[1] = make_float2(ra[u].z, ra[u].w);
            }
#pragma unroll
        for (int u = 0; u < B_PER; ++u)
            if (b_g[u] >= 0) {
                float2* d = reinterpret_cast<float2*>(buf + b_lds[u]);
                d[0] = make_float2(rb[u].x, rb[u].y);
                d[1] = make_float2(rb[u].z, rb[u].w);
            }
    };

    // per-unit forms (u in [0, A_PER + B_PER)): the unrolled k-step loop issues one or two staging
    // units per k-step instead of the whole prefetch / store as a block, so a wave never leaves its
    // MFMA stream for more than a few instructions
    auto load_unit = [&](int u, int m0, int n0, int c, bool live) {
        const int cbase = c * KC;
        if (u < A_PER) {
            const int key = a_key[u];
            const int row = key >> 3, c4 = key & 7;
            const int gr = m0 - 1 + row;
            const bool ok = live && key >= 0 && gr >= 0 && gr < a.rows_in && cbase + 4 * c4 < a.cp_in;
            const float* src = ok ? a.x + (int64_t)(m0 - 1 + row) * a.cp_in + cbase + 4 * c4 : a.zero;
            ra[u] = *reinterpret_cast<const float4*>(src);
        } else {
            const int v = u - A_PER;
            const float* src = (live && b_g[v] >= 0) ? a.w + ((int64_t)n0 * a.nch + c) * (3 * KC) + b_g[v] : a.zero;
            rb[v] = *reinterpret_cast<const float4*>(src);
        }
    };
    auto store_unit = [&](int u, float* buf) {
        if (u < A_PER) {
            if (a_key[u] >= 0) {
                float2* d = reinterpret_cast<float2*>(buf + a_lds[u]);
                d[0] = make_float2(ra[u].x, ra[u].y);
                d[1] = make_float2(ra[u].z, ra[u].w);
            }
        } else {
            const int v = u - A_PER;
            if (b_g[v] >= 0) {
                float2* d = reinterpret_cast<float2*>(buf + b_lds[v]);
                d[0] = make_float2(rb[v].x, rb[v].y);
                d[1] = make_float2(rb[v].z, rb[v].w);
            }
        }
    };

    // ---- tile walk (tile_walk.hpp) with dead-tile elimination -------------------------------------
    // A tile whose rows all lie beyond their read's length has an all-zero output: it is zero-filled here,
    // without loads, MFMAs or pipeline slots, when the walk steps over it.  Such a tile lies inside one
    // read's slot (a tile containing a read start always has valid rows): one uniform look-up.
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
            if (tile_origin(q, tm0, tn0)) {
                if (!a.walk.check_dead) break;
                const int b = tm0 / P_in_;
                const int t0 = tm0 - b * P_in_;
                if (!(t0 + BM <= P_in_ && t0 >= (as_const_len(a.len)[b] >> (a.shift_out - 1)))) break;
                // zero-fill the BM/2 x BN output tile (16-byte pieces; rows are cp_out wide)
                const int pieces_per_row = BN / 4;
                for (int f = threadIdx.x; f < (BM / 2) * pieces_per_row; f += blockDim.x) {
                    const int rr = f / pieces_per_row, cc = (f - rr * pieces_per_row) * 4;
                    const int prow = (tm0 >> 1) + rr, col = tn0 + cc;
                    if (2 * prow < a.rows_in && col < a.cp_out)
                        *reinterpret_cast<float4*>(a.y + (int64_t)prow * a.cp_out + col) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            q = order_index();
        }
        return q;
    };
    int o = next_live();
    if (o >= tiles) return;
    if (a.stamps && tid == 0) {
        a.stamps[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
        a.stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int c = 0;
    int m0, n0;
    tile_origin(o, m0, n0);
    load_item(m0, n0, 0, true);
    store_item(lds);
    __syncthreads();
    int buf = 0;

    const int a_rd = (wm * 16 * MT + r) * S + kq;              // + i*16*S + kw*S + c0
    const int b_rd = a_elems + (wn * 16 * NT + r) * 3 * S + kq;   // + j*48*S + kw*S + c0

#ifdef RS_ITEM_STAMPS
    unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
    int n_items = 0;
#endif
    while (true) {
        RS_STAMP(0);                                               // barrier exit -> loop top
        int nc = c + 1, no = o;
        if (nc == a.nch) {
            nc = 0;
            no = next_live();
        }
        const bool has_next = no < tiles;
        int nm0 = m0, nn0 = n0;
        if (has_next && nc == 0) tile_origin(no, nm0, nn0);
        const float* Ab = lds + buf * buf_elems + a_rd;
        const float* Bb = lds + buf * buf_elems + b_rd;
        if constexpr (KCT > 0) {
            // Fully unrolled k-steps.  The next item's global loads (address VALU + 16-byte loads)
            // are issued after the first k-step and its LDS writes before the last one, i.e. in
            // the shadow of queued MFMAs rather than in the bubble around the barrier where both
            // waves of a SIMD would do them at the same time.  The loads are unconditional (the
            // last item prefetches the zero page) so nothing makes hipcc drain vmcnt early.
            constexpr int KQ = KCT / 4, NSTEPS = 3 * KQ;
            // the two waves of a SIMD (wave w and w + 4 under the usual 0,2,1,3 SIMD assignment)
            // are staggered: the first group prefetches after k-step 0 and stores at 2/3 of the
            // block, the second group prefetches at 1/3 and stores just before the end, so that
            // one wave's VALU / VMEM / LDS-write phase overlaps the other's MFMA stream
            constexpr int LOAD_A = 0, STORE_A = (2 * NSTEPS) / 3, LOAD_B = NSTEPS / 3, STORE_B = NSTEPS - 2;
            const bool grp_b = __builtin_amdgcn_readfirstlane(wave) >= (WM * WN) / 2;
            // fragments are double-buffered in registers: the ds_reads of k-step st+1 are issued
            // before the MFMAs of k-step st, so no MFMA waits on LDS latency
            float af[2][MT], bf[2][NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[0][i] = Ab[i * 16 * (KCT + 2)];
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[0][j] = Bb[j * 48 * (KCT + 2)];
#pragma unroll
            for (int st = 0; st < NSTEPS; ++st) {
                if (st + 1 < NSTEPS) {
                    const int kw = (st + 1) / KQ, c0 = 4 * ((st + 1) % KQ);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        af[(st + 1) & 1][i] = Ab[i * 16 * (KCT + 2) + kw * (KCT + 2) + c0];
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        bf[(st + 1) & 1][j] = Bb[j * 48 * (KCT + 2) + kw * (KCT + 2) + c0];
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] =
                            __builtin_amdgcn_mfma_f32_16x16x4f32(af[st & 1][i], bf[st & 1][j], acc[i][j], 0, 0, 0);
                {
                    // distributed staging: units are loaded in the first half of the item and written
                    // to the other LDS buffer in the second half, UPS units per k-step
                    constexpr int UNITS = A_PER + B_PER;
                    constexpr int HALF = NSTEPS / 2;
                    constexpr int UPS = (UNITS + HALF - 1) / HALF;
                    __builtin_amdgcn_sched_barrier(0);
                    if (st < HALF) {
#pragma unroll
                        for (int q = 0; q < UPS; ++q)
                            if (st * UPS + q < UNITS) load_unit(st * UPS + q, nm0, nn0, nc, has_next);
                    } else {
#pragma unroll
                        for (int q = 0; q < UPS; ++q)
                            if ((st - HALF) * UPS + q < UNITS && has_next)
                                store_unit((st - HALF) * UPS + q, lds + (buf ^ 1) * buf_elems);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            load_item(nm0, nn0, nc, has_next);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
            for (int kw = 0; kw < 3; ++kw) {
                const float* Ak = Ab + kw * S;
                const float* Bk = Bb + kw * S;
#pragma unroll 1
                for (int c0 = 0; c0 < KC; c0 += 4) {
                    float af[MT], bf[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) af[i] = Ak[i * 16 * S + c0];
#pragma unroll
                    for (int j = 0; j < NT; ++j) bf[j] = Bk[j * 48 * S + c0];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
                }
            }
        }

        if (c == a.nch - 1) {
            // ---- epilogue: bias + ReLU + MaxPool(2,2) in registers, masked store -----------------
            float bias[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) bias[j] = a.bias[n0 + (wn * NT + j) * 16 + r];
            // read index of a pooled row: one exact division per tile (wave-uniform), then a
            // small-numerator float quotient per row (t < 2^16: exact)
            const int pr0 = m0 >> 1;
            const int b0 = pr0 / a.P_out;
            const int p0 = pr0 - b0 * a.P_out;
            // pass 1: pooled-row bookkeeping and the per-read length look-ups, all loads issued
            // together (one L2 round trip instead of 2*MT serialised ones)
            int prow_[MT][2];
            int lim_[MT][2];                                       // valid pooled rows of the read, or 0
            int pin_[MT][2];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = m0 + (wm * MT + i) * 16 + 4 * kq;   // even; rows row..row+3 = regs 0..3
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int prow = (row >> 1) + h;                // pooled output row
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
                                a.y[(int64_t)prow * a.cp_out + col] = valid ? v : 0.0f;
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
#ifdef RS_ITEM_STAMPS
        ++n_items;
#endif
        if (!has_next) {
#ifdef RS_ITEM_STAMPS
            if (a.stamps && lane == 0 && blockIdx.x < 4) {
                for (int k = 0; k < 7; ++k) a.stamps[2048 + (blockIdx.x * 8 + wave) * 8 + k] = ph[k];
                a.stamps[2048 + (blockIdx.x * 8 + wave) * 8 + 7] = n_items;
            }
#endif
            if (a.stamps && tid == 0) {
                a.stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime();
                a.stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
            }
            break;
        }
        if constexpr (KCT == 0) store_item(lds + (buf ^ 1) * buf_elems);
        RS_STAMP(5);                                               // last k-steps (+ epilogue)
        __syncthreads();
        buf ^= 1;
        o = no;
        c = nc;
        m0 = nm0;
        n0 = nn0;
    }
}

using KernelFn = void (*)(const ConvArgs);

struct Shape {
    int wm, wn, mt, nt;
    KernelFn fn[4];        // chunk = run-time, 16, 20, 24
};

#define RS_SHAPE(WM, WN, MT, NT)                                                                        \
    {WM, WN, MT, NT, {conv_f32_kernel<WM, WN, MT, NT, 0>, conv_f32_kernel<WM, WN, MT, NT, 16>,        \
                      conv_f32_kernel<WM, WN, MT, NT, 20>, conv_f32_kernel<WM, WN, MT, NT, 24>}}
const Shape kShapes[] = {
    // narrow outputs: all 8 waves stacked along rows
    RS_SHAPE(8, 1, 4, 2), RS_SHAPE(8, 1, 4, 3), RS_SHAPE(8, 1, 2, 5), RS_SHAPE(8, 1, 4, 5),
    RS_SHAPE(8, 1, 2, 7),
    // wide outputs: 4 x 2 waves
    RS_SHAPE(4, 2, 4, 2), RS_SHAPE(4, 2, 4, 3), RS_SHAPE(4, 2, 2, 4), RS_SHAPE(4, 2, 4, 4),
    RS_SHAPE(4, 2, 4, 5), RS_SHAPE(4, 2, 2, 6), RS_SHAPE(4, 2, 4, 6), RS_SHAPE(4, 2, 2, 7),
    RS_SHAPE(4, 2, 2, 8),
    // short batches (few rows): 2 x 4 waves
    RS_SHAPE(2, 4, 2, 2), RS_SHAPE(2, 4, 2, 4), RS_SHAPE(2, 4, 1, 4),
    // experimental: 4 waves = one per SIMD
    RS_SHAPE(4, 1, 4, 8), RS_SHAPE(2, 2, 8, 6), RS_SHAPE(4, 1, 4, 7), RS_SHAPE(4, 1, 4, 5), RS_SHAPE(2, 2, 8, 4),
};
#undef RS_SHAPE
constexpr int kNumShapes = sizeof(kShapes) / sizeof(kShapes[0]);

size_t lds_bytes(const Shape& s, int kc) {
    const int bm = s.wm * 16 * s.mt, bn = s.wn * 16 * s.nt;
    return 2 * (size_t)((bm + 2) + 3 * bn) * (kc + 2) * sizeof(float);
}

// Pick the tile shape for a layer launch.  Model: one persistent workgroup per CU; time =
// rounds x (MFMA issue of a tile + per-item and per-tile overheads), in SIMD cycles.
const Shape* choose_shape(int64_t rows, int n16 /* couts / 16 */, int kc, int nch, int num_cu, double* cost_out) {
    const Shape* best = nullptr;
    double best_cost = 1e300;
    for (int k = 0; k < kNumShapes; ++k) {
        const Shape& s = kShapes[k];
        if (lds_bytes(s, kc) > 160 * 1024 || s.wm * s.wn != 8) continue;
        const int bm = s.wm * 16 * s.mt, bnt = s.wn * s.nt;
        const int64_t mtiles = (rows + bm - 1) / bm;
        const int64_t ntiles = (n16 + bnt - 1) / bnt;
        const int64_t tiles = mtiles * ntiles;
        const int64_t rounds = (tiles + num_cu - 1) / num_cu;
        const double steps = 3.0 * kc / 4.0;
        // two waves share a SIMD: a k-step issues 2 * mt * nt MFMAs of 32 cycles on it; the
        // (mt + nt) LDS reads per step and the barrier per item are partly exposed
        const double item = steps * (2.0 * s.mt * s.nt * 32.0 + 6.0 * (s.mt + s.nt)) + 900.0;
        const double tile = nch * item + 1500.0 + 40.0 * s.mt * s.nt;
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

// largest BN any shape uses: weight / bias tables are padded by this many zero rows so that
// every shape can be chosen at run time without bounds checks on the weight loads
int conv_f32_max_bn() { return 256; }
int conv_f32_kc_max() { return kKcMax; }

int launch_conv_f32(const ConvLayerDev& L, const float* d_x, float* d_y, const int32_t* d_len, int B, int P_in,
                    int layer_index, int num_cu, const float* d_zero, int check_dead, hipStream_t st, int* bm_out,
                    int* bn_out) {
    const ConvPlan& p = L.plan;
    if (p.kc < 4 || p.kc > kKcMax || (p.kc & 3)) {
        set_error("conv_f32: unsupported channel chunk %d", p.kc);
        return RS_ERR_ARG;
    }
    const int64_t rows64 = (int64_t)B * P_in;
    if (rows64 > 0x7fffffff) {
        set_error("conv_f32: batch too large (%lld rows)", (long long)rows64);
        return RS_ERR_ARG;
    }
    const int n16 = round_up(L.c_out, 16) / 16;
    const Shape* s = choose_shape(rows64, n16, p.kc, p.nch, num_cu, nullptr);
    if (const char* force = L.hooks->force_f32; *force) {         // tuning aid: "layer:wm,wn,mt,nt;..."
        int l, wm, wn, mt, nt;
        for (const char* q = force; q && *q; q = strchr(q, ';') ? strchr(q, ';') + 1 : nullptr)
            if (sscanf(q, "%d:%d,%d,%d,%d", &l, &wm, &wn, &mt, &nt) == 5 && l == layer_index)
                for (int k = 0; k < kNumShapes; ++k)
                    if (kShapes[k].wm == wm && kShapes[k].wn == wn && kShapes[k].mt == mt && kShapes[k].nt == nt &&
                        lds_bytes(kShapes[k], p.kc) <= 160 * 1024)
                        s = &kShapes[k];
    }
    if (!s) {
        set_error("conv_f32: no tile shape fits (kc=%d)", p.kc);
        return RS_ERR_ARG;
    }
    const int BM = s->wm * 16 * s->mt, BN = s->wn * 16 * s->nt;
    ConvArgs a;
    a.x = d_x;
    a.w = static_cast<const float*>(L.d_w);
    a.bias = L.d_bias;
    a.y = d_y;
    a.len = d_len;
    a.zero = d_zero;
    a.rows_in = (int)rows64;
    a.P_out = P_in / 2;
    a.inv_P_out = 1.0f / (float)a.P_out;
    a.cp_in = L.cp_in;
    a.cp_out = L.cp_out;
    a.kc = p.kc;
    a.nch = p.nch;
    a.shift_out = layer_index + 1;
    const size_t lds = lds_bytes(*s, p.kc);
    KernelFn fn = s->fn[p.kc == 16 ? 1 : p.kc == 20 ? 2 : p.kc == 24 ? 3 : 0];
    RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
    const int n_mtiles = (a.rows_in + BM - 1) / BM, n_ntiles = (n16 * 16 + BN - 1) / BN;
    const int64_t tiles = (int64_t)n_mtiles * n_ntiles;
    const unsigned grid = (unsigned)std::min<int64_t>(tiles, num_cu);
    a.walk = plan_walk(n_mtiles, n_ntiles, grid, num_cu, BM, 3.0 * BN, check_dead, !L.hooks->no_rect_order);
    static unsigned long long* d_stamps = nullptr;
    const bool want_stamps = L.hooks->conv_stamps;
    if (want_stamps && !d_stamps) RS_HIP(hipMalloc(reinterpret_cast<void**>(&d_stamps), 4096 * 4 * 8));
    a.stamps = want_stamps ? d_stamps : nullptr;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(s->wm * s->wn * 64), lds, st, a);
    RS_HIP(hipGetLastError());
    if (want_stamps) {                                              // diagnostic only: synchronises
        std::vector<unsigned long long> h(grid * 4);
        RS_HIP(hipStreamSynchronize(st));
        RS_HIP(hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> clk, dur;
        for (unsigned k = 0; k < grid; ++k) {
            const double dt = (double)(h[k * 4 + 2] - h[k * 4 + 0]), dr = (double)(h[k * 4 + 3] - h[k * 4 + 1]);
            if (dr > 0) {
                clk.push_back(dt / dr * 100.0);                     // MHz (memrealtime ticks at 100 MHz)
                dur.push_back(dr / 100.0);                          // us
            }
        }
        std::sort(clk.begin(), clk.end());
        std::sort(dur.begin(), dur.end());
#ifdef RS_ITEM_STAMPS
        {
            std::vector<unsigned long long> hp(4 * 8 * 8);
            RS_HIP(hipMemcpy(hp.data(), d_stamps + 2048, hp.size() * 8, hipMemcpyDeviceToHost));
            for (int w = 0; w < 8; w += 4) {
                const unsigned long long* q = &hp[w * 8];
                const double n = (double)q[7];
                fprintf(stderr, "[item-stamps] layer %d wave %d items %.0f: per item cycles: barrier->top %.0f | step0 %.0f | "
                        "prefetch issue %.0f | mfma body %.0f | vmcnt wait %.0f | lds store %.0f | tail(+epi) %.0f\n",
                        layer_index, w, n, q[0] / n, q[1] / n, q[2] / n, q[3] / n, q[6] / n, q[4] / n, q[5] / n);
            }
        }
#endif
        if (!clk.empty())
            fprintf(stderr, "[stamps] layer %d tile %dx%d: shader clock median %.0f MHz (min %.0f max %.0f), block time median %.1f us max %.1f us\n",
                    layer_index, BM, BN, clk[clk.size() / 2], clk.front(), clk.back(), dur[dur.size() / 2], dur.back());
    }
    if (bm_out) *bm_out = BM;
    if (bn_out) *bn_out = BN;
    return RS_OK;
}

}  // namespace rs
