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
#include "conv_wino_kernel.hpp"




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

// conv_wino_thin.hip: the thin-launch instantiation of a shape for chunk index ki (16 / 20 / 24), or null
KernelFn conv_wino_thin_fn(int wm, int wn, int mt, int nt, int ki);

namespace {


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
// the thin-launch forms are instantiated in conv_wino_thin.hip
#define RS_THIN3(WM, WN, MT, NT) \
    {conv_wino_thin_fn(WM, WN, MT, NT, 0), conv_wino_thin_fn(WM, WN, MT, NT, 1), conv_wino_thin_fn(WM, WN, MT, NT, 2)}
#define RS_SHAPE_D(WM, WN, MT, NT)                                                                     \
    {WM, WN, MT, NT, {conv_wino_kernel<WM, WN, MT, NT, 16>, conv_wino_kernel<WM, WN, MT, NT, 20>,     \
                      conv_wino_kernel<WM, WN, MT, NT, 24>}, RS_THIN3(WM, WN, MT, NT)}
// four-wave shapes exist in the one-item-ahead form only
#define RS_SHAPE_4(WM, WN, MT, NT) {WM, WN, MT, NT, RS_THIN3(WM, WN, MT, NT), {nullptr, nullptr, nullptr}}
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
#undef RS_THIN3
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
            },
            L.hooks->tail_margin > 0 ? L.hooks->tail_margin : 0.97);
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
