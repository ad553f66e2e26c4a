// C ABI of libriser_amd (see include/riser_amd.h): model construction / weight packing,
// workspace layout, tile planning and the launch sequence of one forward pass.
#include "common.hpp"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

namespace rs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return RS_ERR_HIP;
}

Hooks Hooks::from_env() {
    Hooks h;
    auto flag = [](const char* n) { return getenv(n) != nullptr; };
    auto text = [](const char* n, char* dst, size_t cap) {
        if (const char* e = getenv(n)) {
            strncpy(dst, e, cap - 1);
            dst[cap - 1] = 0;
        }
    };
    h.no_rect_order = flag("RS_NO_RECT_ORDER");
    h.no_tail_split = flag("RS_NO_TAIL_SPLIT");
    h.tail_debug = flag("RS_TAIL_DEBUG");
    if (const char* e = getenv("RS_TAIL_MARGIN")) h.tail_margin = atof(e);
    h.no_deep_staging = flag("RS_NO_DEEP_STAGING");
    if (const char* e = getenv("RS_SMALL_SHARED")) h.small_shared = atoi(e);
    if (const char* e = getenv("RS_SMALL_NW")) h.small_nw = atoi(e);
    if (const char* e = getenv("RS_SF32_MIN_RUN")) h.sf32_min_run = atoi(e) > 0 ? atoi(e) : 1;
    h.ring_tail_split = flag("RS_RING_TAIL_SPLIT");
    h.no_fuse0 = flag("RS_NO_FUSE0");
    h.no_stream_f32 = flag("RS_NO_STREAM_F32");
    h.no_stream_h16 = flag("RS_NO_STREAM_H16");
    h.no_stream012 = flag("RS_NO_STREAM012");
    h.ensemble_serial = flag("RS_ENSEMBLE_SERIAL");
    h.one_level = flag("RS_ONE_LEVEL");
    h.conv_stamps = flag("RS_CONV_STAMPS");
    text("RS_FORCE_SHAPE_F32", h.force_f32, sizeof(h.force_f32));
    text("RS_FORCE_SHAPE_WINO", h.force_wino, sizeof(h.force_wino));
    text("RS_FORCE_SHAPE_WINO4", h.force_wino4, sizeof(h.force_wino4));
    text("RS_FORCE_SHAPE_RING", h.force_ring, sizeof(h.force_ring));
    text("RS_EMU_ROWS", h.emu_rows, sizeof(h.emu_rows));
    if (const char* e = getenv("RS_SMALL_F32_WAVES")) h.small_f32_waves = atoi(e);
    if (const char* e = getenv("RS_H16_WRES")) h.h16_wres = atoi(e) != 0;
    if (const char* e = getenv("RS_THIN_H16_ROWS")) h.thin_h16_rows = atoi(e);
    if (const char* e = getenv("RS_F8_MIN_CIN")) h.f8_min_cin = atoi(e);
    if (const char* e = getenv("RS_X3_TAIL")) h.x3_tail = atoi(e) != 0;
    return h;
}

const Hooks& default_hooks() {
    static const Hooks h;
    return h;
}

}  // namespace rs

using namespace rs;

struct rs_model {
    // test hook (rs_debug_capture_layer): copy the output buffer of conv layer dbg_layer to dbg_dst
    void* dbg_dst = nullptr;
    size_t dbg_bytes = 0;
    int dbg_layer = -1;
    Hooks hooks;                          // RS_* switches, read once in rs_model_create
    int device = 0;
    int dtype = RS_F32;
    int n_layers = 0;
    int pad_shift = 0;                    // log2 of the block size of the (late layers') packed layout (>= n_layers)
    // two-level packed layout: conv layers 0 .. split - 1 run on FINE blocks of 1 << fine_shift samples, a
    // re-pack of layer split - 1's (small) output moves the batch to the blocks of 1 << pad_shift samples the late layers and
    // the head need.  split == n_layers / fine_shift == pad_shift: one level.
    int fine_shift = 0;
    int split = 0;
    int n_classes = 2;
    int channels[kMaxLayers] = {0};
    int cp[kMaxLayers] = {0};             // padded row width of layer i's OUTPUT buffer
    float* d_w0 = nullptr;                // layer 0: [cp[0]][4] = (w0, w1, w2, bias)
    ConvLayerDev layers[kMaxLayers];      // i >= 1
    float* d_fcw = nullptr;               // [2][c_last]
    float* d_fcb = nullptr;
    FcHead fc;                            // fc.H > 0: the `fc` classifier replaces the gap_fc head (rs_model_set_fc_classifier)
    float* d_zero = nullptr;              // 256 zero bytes: target of masked-off staging loads
    unsigned* d_sat = nullptr;            // half-precision modes: sticky word, non-zero once an activation overflowed f16 (rs_model_saturated)
    int num_cu = 256;
    int last_bm[kMaxLayers] = {0};
    int last_bn[kMaxLayers] = {0};
    bool last_ring[kMaxLayers] = {false};  // the layer's last launch ran the LDS-DMA ring kernel
    // stage profiling (rs_profile_*): events recorded on the launch stream
    bool prof_on = false;
    bool prof_open = false;      // a profiled call has recorded its opening event (rs_classify opens before normalise)
    int prof_level = 1;          // 1: one event per launch; 2: call start, end of normalise + layer 0, end of the conv stack, head
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_stage;            // stage of event k (-1 = start of a call)
    size_t ev_used = 0;
    int prof_calls = 0;
    bool tuning = false;                  // rs_autotune: time every feasible tile shape of each tiled layer in place
    int tuned_changed = 0;                // layers whose measured best differs from the planner's choice
    // rs_classify_ensemble: the forwards of models 1.. run on library-owned side streams next to model 0's on the
    // caller's stream (created on first use)
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

namespace {

constexpr size_t kAlign = 256;
inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

// storage type of the activations: the Winograd fp32 path shares every non-conv kernel with RS_F32
// (RS_F16XF8 is RS_F16X3 everywhere but in the wide layers' conv kernel and their rows)
int act_dtype(const rs_model* m) { return m->dtype == RS_F32W ? RS_F32 : m->dtype == RS_F16XF8 ? RS_F16X3 : m->dtype; }
int esize(const rs_model* m) { return act_dtype(m) == RS_F32 ? 4 : 2; }
// 16-bit storage type of a mode (RS_BF16 or RS_F16), also for the split-precision modes
int base16(int dtype) { return is_f16_family(dtype) ? RS_F16 : RS_BF16; }

// ---- static part of the plan: the K chunking (fixes the weight packing) -------------------------
// kc minimises nch * (3*kc/4 + 0.75) k-steps (0.75 step ~ the per-item barrier + LDS write);
// the tile shape is chosen per launch from the batch's row count (conv_f32.hip).
ConvPlan plan_static_f32(int cp_in, int c_out) {
    ConvPlan p{};
    double best_cost = -1;
    for (int kc = 4; kc <= conv_f32_kc_max(); kc += 4) {
        // chunks of 16 / 20 / 24 channels have fully unrolled kernels (immediate LDS offsets);
        // other sizes run the generic kernel and are only worth it for very narrow layers
        if (cp_in >= 16 && kc != 16 && kc != 20 && kc != 24) continue;
        const int nch = (cp_in + kc - 1) / kc;
        const double cost = nch * (3.0 * kc / 4.0 + 0.75);
        if (best_cost < 0 || cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && kc > p.kc)) {
            best_cost = cost;
            p.kc = kc;
            p.nch = nch;
        }
    }
    p.n_alloc = round_up(c_out, 16) + conv_f32_max_bn();
    return p;
}

// Winograd F(2,3): a chunk of kc channels is 4 * kc / 4 = kc MFMA slots; ~1.5 slots per item for
// the barrier and the staging writes
ConvPlan plan_static_wino(int cp_in, int c_out) {
    ConvPlan p{};
    double best_cost = -1;
    for (int kc = 16; kc <= 24; kc += 4) {
        const int nch = (cp_in + kc - 1) / kc;
        const double cost = nch * (kc + 1.5);
        if (best_cost < 0 || cost < best_cost - 1e-9) {
            best_cost = cost;
            p.kc = kc;
            p.nch = nch;
        }
    }
    p.n_alloc = round_up(c_out, 16) + conv_wino_max_bn();
    return p;
}

// F(4,3): chunks of 16 or 20 channels (LDS capacity); 6 * kc / 4 slots per chunk
// tuning aid: RS_PLAN_KC = "layer:kc;layer:kc" forces the channel chunk of a layer when the model is created
int forced_kc(int layer) {
    if (const char* e = getenv("RS_PLAN_KC")) {
        int l, kc;
        for (const char* q = e; q && *q; q = strchr(q, ';') ? strchr(q, ';') + 1 : nullptr)
            if (sscanf(q, "%d:%d", &l, &kc) == 2 && l == layer) return kc;
    }
    return 0;
}

// The chunk (16 or 20 channels; LDS capacity) fixes the weight packing, but also which tile shapes fit in LDS
// (80-channel-wide tiles need chunks of 16), so it is chosen with the launch planner's own cost estimate at a
// nominal batch (512 reads of 16000 samples, BASELINE config 2).
ConvPlan plan_static_wino4(int cp_in, int c_out, int layer, int num_cu) {
    ConvPlan p{};
    double best_cost = -1;
    const int64_t groups = (int64_t)512 * (16384 >> layer) / 4;
    for (int kc = 16; kc <= 20; kc += 4) {
        if (forced_kc(layer) && kc != forced_kc(layer)) continue;
        const int nch = (cp_in + kc - 1) / kc;
        const double cost = conv_wino4_plan_cost(groups, round_up(c_out, 16) / 16, kc, nch, num_cu);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            p.kc = kc;
            p.nch = nch;
        }
    }
    p.n_alloc = round_up(c_out, 16) + conv_wino4_max_bn();
    return p;
}

// measurement aid (libraries built with -DRS_X3_MASK only): RS_X3_TERMS = "layer:mask;layer:mask" picks the products a
// split-precision layer of the ring kernel executes (1 = hi*hi, 2 = x lo * w hi, 4 = x hi * w lo; 7 = all, the shipped mode)
int x3_terms_of(int layer) {
    if (const char* e = getenv("RS_X3_TERMS")) {
        int l, t;
        for (const char* q = e; q && *q; q = strchr(q, ';') ? strchr(q, ';') + 1 : nullptr)
            if (sscanf(q, "%d:%d", &l, &t) == 2 && l == layer) return (t & 7) | 1;
    }
    return 7;
}

// which layers of an RS_F32W model run F(4,3) instead of F(2,3): by default the wide ones (>= 96 input
// channels: layers 5-11 of the shipped net, where the matrix pipe is the bound: measured -8 % on layer 5,
// -12 ... -20 % on layers 6-10, -5 % on layer 11; +15 % on layer 4, whose tiles are dominated by staging and
// epilogue), plus narrower layers whose output channels fill the 80-wide F(4,3) tile exactly (16 * 5 | padded
// C_out: layer 3 of the shipped net, 45 -> 67 channels, measured -8 %).  RS_WINO4 = comma list of layer indices
// overrides it when the model is created ("none" = F(2,3) everywhere).
bool use_wino4(int layer, int n_layers, int c_in, int c_out) {
    if (const char* e = getenv("RS_WINO4")) {
        for (const char* q = e; *q;) {
            char* end = nullptr;
            const long v = strtol(q, &end, 10);
            if (end == q) break;
            if (v == layer) return true;
            q = *end ? end + 1 : end;
        }
        return false;
    }
    // F(4,3) groups four input rows: every read must start on a group boundary in such a layer (see pad_shift in
    // rs_model_create), i.e. blocks of 2^(layer + 2) samples.  The last layer would double the block size of the packed
    // layout (2^13 samples for the 12-layer net: an 8615-sample read would occupy 16384) for ~5 % of that layer's time,
    // so it stays on F(2,3) unless RS_WINO4 names it.
    if (layer + 2 > n_layers) return false;
    return c_in >= 96 || (c_in >= 32 && (round_up(c_out, 16) / 16) % 5 == 0);
}

// Workspace: [coarse block table][fine block table][16 zero bytes | normalised signals, fine blocks of Uf floats][activation
// buffer A][B].  Laid out for the upper bounds NB = B * (Lmax / U + 1) blocks of either size, so the offsets depend on
// (B, Lmax) only; a batch of mixed lengths uses a prefix of every region.  With one level the fine table IS the coarse one.
struct WsLayout {
    size_t rbase_off, blen_off, bread_off;          // coarse table
    size_t rbase_f_off, blen_f_off, bread_f_off;    // fine table (== coarse when the model has one level)
    size_t xnorm_off, bufa_off, bufb_off, fc_part_off, total;
    int U, Uf;              // block sizes in samples (1 << pad_shift, 1 << fine_shift)
    int nblk_max, nblk_f_max;   // blocks of a read of Lmax samples
    int64_t nb_max, nb_f_max;   // B * nblk_max
};

inline bool two_level(const rs_model* m) { return m->split < m->n_layers && m->fine_shift < m->pad_shift; }
// block size (log2) of the layout conv layer i READS (i = 0: the normalised signal); its output is in the same layout, except
// that layer split - 1's output is re-packed to the coarse layout before layer `split` reads it
inline int layer_shift(const rs_model* m, int i) { return (two_level(m) && i < m->split) ? m->fine_shift : m->pad_shift; }

WsLayout ws_layout(const rs_model* m, int B, int Lmax) {
    WsLayout w{};
    const bool two = two_level(m);
    w.U = 1 << m->pad_shift;
    w.Uf = two ? 1 << m->fine_shift : w.U;
    w.nblk_max = (Lmax >> m->pad_shift) + 1;
    w.nb_max = (int64_t)B * w.nblk_max;
    w.nblk_f_max = two ? (Lmax >> m->fine_shift) + 1 : w.nblk_max;
    w.nb_f_max = (int64_t)B * w.nblk_f_max;
    size_t buf = 0;
    for (int i = 0; i < m->n_layers; ++i) {                       // layer i's output buffer
        const bool fine = two && i < m->split;
        const size_t rows = fine ? (size_t)w.nb_f_max * (w.Uf >> (i + 1)) : (size_t)w.nb_max * (w.U >> (i + 1));
        // F8 rows carry their scale plane behind them (conv_ring_f8.hip)
        const bool f8 = i + 1 < m->n_layers && m->layers[i + 1].f8_in;
        auto bytes_of = [&](size_t r) { return f8 ? f8_scale_offset((int64_t)r, m->cp[i]) + f8_scale_bytes((int64_t)r, m->cp[i]) : r * m->cp[i] * esize(m); };
        buf = std::max(buf, bytes_of(rows));
        if (two && i == m->split - 1)                             // ... and its re-packed copy in the coarse layout
            buf = std::max(buf, bytes_of((size_t)w.nb_max * (w.U >> (i + 1))));
    }
    buf = align_up(buf + kAlign);
    w.rbase_off = 0;
    w.blen_off = align_up((size_t)(B + 1) * 4);
    w.bread_off = w.blen_off + align_up((size_t)w.nb_max * 4);
    size_t at = w.bread_off + align_up((size_t)w.nb_max * 4);
    if (two) {
        w.rbase_f_off = at;
        w.blen_f_off = w.rbase_f_off + align_up((size_t)(B + 1) * 4);
        w.bread_f_off = w.blen_f_off + align_up((size_t)w.nb_f_max * 4);
        at = w.bread_f_off + align_up((size_t)w.nb_f_max * 4);
    } else {
        w.rbase_f_off = w.rbase_off;
        w.blen_f_off = w.blen_off;
        w.bread_f_off = w.bread_off;
    }
    w.xnorm_off = at + kAlign;                                    // the last 16 bytes before the rows are a zero prefix
    w.bufa_off = align_up(w.xnorm_off + (size_t)w.nb_f_max * w.Uf * sizeof(float));
    w.bufb_off = w.bufa_off + buf;
    w.fc_part_off = w.bufb_off + buf;
    w.total = w.fc_part_off + (m->fc.H ? align_up(fc_head_workspace_bytes(B, m->fc.H)) : 0);
    return w;
}

// What one call runs on: the block table(s) in the workspace and the number of blocks in use (host-known: from the host's
// copy of the lengths, or nblk_max blocks for every read when it has none)
struct Batch {
    BlockPlan plan;         // coarse: late layers, head
    BlockPlan fine;         // early layers, normalised rows (a copy of `plan` when the model has one level)
    int NB = 0, NBf = 0;    // blocks in use
    int Lmin_blk = 0, Lmin_blk_f = 0;   // lower bound of blen over the reads' last blocks (dead-tile hint), 0 = unknown
};

// h_len may be NULL.  Returns RS_OK or RS_ERR_LENGTH (a host length outside [2^n_layers, Lmax]).
int make_batch(const rs_model* m, const WsLayout& w, void* d_ws, const int32_t* h_len, int B, int Lmin, int Lmax, Batch* out) {
    char* ws = static_cast<char*>(d_ws);
    const bool two = two_level(m);
    Batch bt;
    bt.plan.rbase = reinterpret_cast<int32_t*>(ws + w.rbase_off);
    bt.plan.blen = reinterpret_cast<int32_t*>(ws + w.blen_off);
    bt.plan.bread = reinterpret_cast<int32_t*>(ws + w.bread_off);
    bt.plan.shift = m->pad_shift;
    bt.fine.rbase = reinterpret_cast<int32_t*>(ws + w.rbase_f_off);
    bt.fine.blen = reinterpret_cast<int32_t*>(ws + w.blen_f_off);
    bt.fine.bread = reinterpret_cast<int32_t*>(ws + w.bread_f_off);
    bt.fine.shift = two ? m->fine_shift : m->pad_shift;
    if (h_len) {
        int64_t nb = 0, nbf = 0;
        int lmin_blk = w.U, lmin_blk_f = w.Uf;
        for (int b = 0; b < B; ++b) {
            const int n = h_len[b];
            if (n < (1 << m->n_layers) || n > Lmax) {
                set_error("read %d has %d samples, outside [%d, Lmax = %d]", b, n, 1 << m->n_layers, Lmax);
                return RS_ERR_LENGTH;
            }
            nb += (n >> m->pad_shift) + 1;
            nbf += (n >> bt.fine.shift) + 1;
            lmin_blk = std::min(lmin_blk, n & (w.U - 1));            // the read's last block holds len mod U samples
            lmin_blk_f = std::min(lmin_blk_f, n & (w.Uf - 1));
        }
        bt.plan.uniform_nblk = bt.fine.uniform_nblk = 0;
        bt.NB = (int)nb;
        bt.NBf = (int)nbf;
        bt.Lmin_blk = lmin_blk;
        bt.Lmin_blk_f = lmin_blk_f;
    } else {
        bt.plan.uniform_nblk = w.nblk_max;
        bt.fine.uniform_nblk = w.nblk_f_max;
        bt.NB = (int)w.nb_max;
        bt.NBf = (int)w.nb_f_max;
        // every read has nblk_max blocks: the last one of the shortest read holds max(Lmin - (nblk_max - 1) U, 0) samples
        bt.Lmin_blk = Lmin > 0 ? std::max(0, std::min(w.U, Lmin - (w.nblk_max - 1) * w.U)) : 0;
        bt.Lmin_blk_f = Lmin > 0 ? std::max(0, std::min(w.Uf, Lmin - (w.nblk_f_max - 1) * w.Uf)) : 0;
    }
    bt.plan.nb_total = bt.NB;
    bt.fine.nb_total = bt.NBf;
    *out = bt;
    return RS_OK;
}

// record an event tagged `stage` (-1 opens a call) on the stream, if profiling is on
void prof_mark(rs_model* m, int stage, hipStream_t st) {
    if (!m->prof_on) return;
    // coarse level: an event costs ~4.5 us on the stream; only the boundaries of the conv stack are kept, a skipped
    // stage's time is added to the next recorded one (normalise -> stage 1, conv layers 1..n-2 -> stage n-1)
    if (m->prof_level == 2 && (stage == 0 || (stage >= 2 && stage < m->n_layers))) return;
    if (m->ev_used == m->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        m->ev_pool.push_back(e);
        m->ev_stage.push_back(0);
    }
    m->ev_stage[m->ev_used] = stage;
    (void)hipEventRecord(m->ev_pool[m->ev_used], st);
    ++m->ev_used;
    if (stage < 0) {
        ++m->prof_calls;
        m->prof_open = true;
    }
    if (stage == m->n_layers + 1) m->prof_open = false;     // the head closes the call
}

// fp32 -> bf16 / f16 bits, round to nearest even (host side, weight packing)
float from_h16(unsigned short u, int dtype) {
    if (dtype == RS_F16) return (float)__builtin_bit_cast(_Float16, u);
    return __builtin_bit_cast(float, (unsigned)u << 16);
}

unsigned short to_h16(float f, int dtype) {
    if (dtype == RS_F16) return __builtin_bit_cast(unsigned short, (_Float16)f);
    unsigned u = __builtin_bit_cast(unsigned, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// fp32 -> OCP e4m3 (bias 7, largest finite 448, subnormals of 2^-9), round to nearest even, saturating
unsigned char to_e4m3(float f) {
    const unsigned char sign = std::signbit(f) ? 0x80 : 0x00;
    float a = fabsf(f);
    if (!(a == a)) return (unsigned char)(sign | 0x7f);
    if (a >= 448.0f) return (unsigned char)(sign | 0x7e);
    int e = a > 0.0f ? ilogbf(a) : -127;
    if (e < -6) e = -6;                                       // subnormal quantum 2^-9
    const float q = ldexpf(1.0f, e - 3);
    const float n = nearbyintf(a / q);                        // half to even (default rounding mode); n <= 16
    if (n == 0.0f) return sign;
    int m = (int)n, ee = e;
    if (m == 16) {                                            // rounded up into the next binade
        m = 8;
        ++ee;
    }
    if (m < 8) return (unsigned char)(sign | m);              // subnormal: exponent field 0
    return (unsigned char)(sign | ((ee + 7) << 3) | (m - 8));
}

// RS_F16XF8: layer i can take part in a run of F8 rows (conv_ring_f8.hip).  Wide layers only (RS_F8_MIN_CIN input channels,
// default 200: layers 7-11 of the shipped net): with 2/3 of the matrix-pipe time a tile of this kernel is bound three ways at
// once - MFMA, L2 -> LDS staging (~24 B/clk/CU) and LDS fragment reads are each ~1 500 cycles per sub-stage at 256 x 192 - and
// its even-NT tile shapes cover the narrow layers' columns worse than the split-precision kernel's (measured, 512 x 16000:
// layers 4, 5 +20 ... +30 %, layer 6 +-0, layers 7 / 8 / 9 / 11 -5 / -11 / -10 / -18 %)
bool f8_eligible(const Hooks& h, int dtype, int i, int n_layers, const int32_t* channels) {
    return dtype == RS_F16XF8 && i >= 3 && i < n_layers && channels[i - 1] >= std::max(64, h.f8_min_cin);
}

template <class T>
int upload(T** dptr, const std::vector<T>& h) {
    RS_HIP(hipMalloc(reinterpret_cast<void**>(dptr), std::max<size_t>(h.size(), 1) * sizeof(T)));
    RS_HIP(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return RS_OK;
}

}  // namespace

extern "C" {

const char* rs_last_error(void) { return g_err; }

int rs_version(void) { return (2 << 16) | 5; }

int rs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rs_model_create(int n_layers, const int32_t* channels, int n_classes, const float* const* conv_w,
                    const float* const* conv_b, const float* fc_w, const float* fc_b, int dtype, int device,
                    rs_model** out) {
    if (!out || !channels || !conv_w || !conv_b || !fc_w || !fc_b) {
        set_error("rs_model_create: null argument");
        return RS_ERR_ARG;
    }
    *out = nullptr;
    if (n_layers < 2 || n_layers > kMaxLayers) {
        set_error("rs_model_create: n_layers %d outside [2, %d]", n_layers, kMaxLayers);
        return RS_ERR_ARG;
    }
    if (n_classes != 2) {
        set_error("rs_model_create: n_classes must be 2 (got %d)", n_classes);
        return RS_ERR_ARG;
    }
    if (dtype != RS_F32 && dtype != RS_BF16 && dtype != RS_F16 && dtype != RS_F32W && !is_x3(dtype)) {      // is_x3: RS_F16XF8 too
        set_error("rs_model_create: unknown dtype %d", dtype);
        return RS_ERR_ARG;
    }
    for (int i = 0; i < n_layers; ++i)
        if (channels[i] < 1 || !conv_w[i] || !conv_b[i]) {
            set_error("rs_model_create: bad layer %d", i);
            return RS_ERR_ARG;
        }
    DeviceGuard guard(device);            // the caller's current device is restored on return
    RS_HIP(guard.err);
    rs_model* m = new (std::nothrow) rs_model();
    if (!m) {
        set_error("rs_model_create: out of host memory");
        return RS_ERR_OOM;
    }
    m->device = device;
    m->hooks = Hooks::from_env();
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
            m->num_cu = cus;
    }
    m->dtype = dtype;
    m->n_layers = n_layers;
    for (int i = 0; i < n_layers; ++i) {
        m->channels[i] = channels[i];
        // row width of layer i's output buffer: channels padded to 16 bytes; split precision: 32-channel panels laid
        // out as [hi x 32 | lo x 32] (conv_ring_h16.hip)
        m->cp[i] = is_x3(dtype) ? 64 * ((channels[i] + 31) / 32)
                                : round_up(channels[i], (dtype == RS_F32 || dtype == RS_F32W) ? 4 : 8);
        // RS_F16XF8: the rows between two layers of a run are F8 rows: 128 elements (an H and an F panel) per 64 channels
        if (f8_eligible(m->hooks, dtype, i, n_layers, channels) && f8_eligible(m->hooks, dtype, i + 1, n_layers, channels))
            m->cp[i] = 128 * ((channels[i] + 63) / 64);
    }
    int rc = RS_OK;
    if (is_f16_family(dtype)) rc = upload(&m->d_sat, std::vector<unsigned>(1, 0u));
    {   // layer 0: (w0, w1, w2, bias) per output channel
        std::vector<float> w4((size_t)m->cp[0] * 4, 0.0f);
        for (int c = 0; c < channels[0]; ++c) {
            w4[c * 4 + 0] = conv_w[0][c * 3 + 0];
            w4[c * 4 + 1] = conv_w[0][c * 3 + 1];
            w4[c * 4 + 2] = conv_w[0][c * 3 + 2];
            w4[c * 4 + 3] = conv_b[0][c];
        }
        rc = upload(&m->d_w0, w4);
    }
    for (int i = 1; i < n_layers && rc == RS_OK; ++i) {
        ConvLayerDev& L = m->layers[i];
        L.hooks = &m->hooks;
        L.d_sat = m->d_sat;
        L.c_in = channels[i - 1];
        L.c_out = channels[i];
        L.cp_in = m->cp[i - 1];
        L.cp_out = m->cp[i];
        L.x3_terms = x3_terms_of(i);
        L.f8_in = f8_eligible(m->hooks, dtype, i, n_layers, channels) && f8_eligible(m->hooks, dtype, i - 1, n_layers, channels);
        L.f8_out = f8_eligible(m->hooks, dtype, i, n_layers, channels) && f8_eligible(m->hooks, dtype, i + 1, n_layers, channels);
        if (dtype == RS_F32) {
            L.plan = plan_static_f32(L.cp_in, L.c_out);
            const ConvPlan& p = L.plan;
            std::vector<float> wp((size_t)p.n_alloc * p.nch * 3 * p.kc, 0.0f);
            for (int n = 0; n < L.c_out; ++n)
                for (int ci = 0; ci < L.c_in; ++ci) {
                    const int c = ci / p.kc, cc = ci - c * p.kc;
                    for (int kw = 0; kw < 3; ++kw)
                        wp[(((size_t)n * p.nch + c) * 3 + kw) * p.kc + cc] =
                            conv_w[i][((size_t)n * L.c_in + ci) * 3 + kw];
                }
            float* dw = nullptr;
            rc = upload(&dw, wp);
            L.d_w = dw;
        } else if (dtype == RS_F32W && use_wino4(i, n_layers, L.c_in, L.c_out)) {
            // Winograd F(4,3) filter transform U = G g (fp64, rounded once); packed [n_alloc][nch][6][kc]
            L.wino_m = 4;
            L.plan = plan_static_wino4(L.cp_in, L.c_out, i, m->num_cu);
            const ConvPlan& p = L.plan;
            std::vector<float> wp((size_t)p.n_alloc * p.nch * 6 * p.kc, 0.0f);
            for (int n = 0; n < L.c_out; ++n)
                for (int ci = 0; ci < L.c_in; ++ci) {
                    const int c = ci / p.kc, cc = ci - c * p.kc;
                    const float* g = &conv_w[i][((size_t)n * L.c_in + ci) * 3];
                    const double g0 = g[0], g1 = g[1], g2 = g[2];
                    const double u[6] = {g0 / 4.0, -(g0 + g1 + g2) / 6.0, -(g0 - g1 + g2) / 6.0,
                                         g0 / 24.0 + g1 / 12.0 + g2 / 6.0, g0 / 24.0 - g1 / 12.0 + g2 / 6.0, g2};
                    for (int j = 0; j < 6; ++j)
                        wp[(((size_t)n * p.nch + c) * 6 + j) * p.kc + cc] = (float)u[j];
                }
            float* dw = nullptr;
            rc = upload(&dw, wp);
            L.d_w = dw;
        } else if (dtype == RS_F32W) {
            // Winograd F(2,3) filter transform (fp64, rounded once): U0 = g0, U1 = (g0+g1+g2)/2,
            // U2 = (g0-g1+g2)/2, U3 = g2; packed [n_alloc][nch][4][kc]
            L.plan = plan_static_wino(L.cp_in, L.c_out);
            const ConvPlan& p = L.plan;
            std::vector<float> wp((size_t)p.n_alloc * p.nch * 4 * p.kc, 0.0f);
            for (int n = 0; n < L.c_out; ++n)
                for (int ci = 0; ci < L.c_in; ++ci) {
                    const int c = ci / p.kc, cc = ci - c * p.kc;
                    const float* g = &conv_w[i][((size_t)n * L.c_in + ci) * 3];
                    const double g0 = g[0], g1 = g[1], g2 = g[2];
                    const double u[4] = {g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2};
                    for (int j = 0; j < 4; ++j)
                        wp[(((size_t)n * p.nch + c) * 4 + j) * p.kc + cc] = (float)u[j];
                }
            float* dw = nullptr;
            rc = upload(&dw, wp);
            L.d_w = dw;
        } else {
            // 16-bit: panels of 32 input channels, packed [panel][tap][n_alloc][32] (conv_stream_h16.hip: layers 1-2)
            const int st16 = base16(dtype);
            const bool x3 = is_x3(dtype);
            float ws = 1.0f;                                          // power-of-two weight scale (half precision only)
            if (is_f16_family(dtype)) {
                float wmax = 0.0f;
                for (size_t k = 0; k < (size_t)L.c_out * L.c_in * 3; ++k) wmax = std::max(wmax, fabsf(conv_w[i][k]));
                if (wmax > 0.0f && std::isfinite(wmax)) ws = ldexpf(1.0f, std::min(60, std::max(-60, 13 - ilogbf(wmax))));
            }
            L.w_unscale = 1.0f / ws;
            ConvPlan& p = L.plan;
            p.kc = 32;
            p.nch = (L.c_in + 31) / 32;
            // rows of the packed weight / bias tables: the channels (split precision: all 32 slots of the last panel,
            // every one of which a tile covers) plus zero rows for the widest tile's overhang
            p.n_alloc = (L.f8_out ? 64 * ((L.c_out + 63) / 64) : x3 ? 32 * ((L.c_out + 31) / 32) : round_up(L.c_out, 16)) + conv_ring_max_bn();
            if (!x3) {
                p.nch = (L.cp_in + 31) / 32;
                std::vector<unsigned short> wp((size_t)p.nch * 3 * p.n_alloc * 32, 0);
                for (int n = 0; n < L.c_out; ++n)
                    for (int ci = 0; ci < L.c_in; ++ci) {
                        const int pn = ci / 32, cc = ci - pn * 32;
                        for (int kw = 0; kw < 3; ++kw)
                            wp[(((size_t)pn * 3 + kw) * p.n_alloc + n) * 32 + cc] =
                                to_h16(conv_w[i][((size_t)n * L.c_in + ci) * 3 + kw] * ws, st16);
                    }
                unsigned short* dw = nullptr;
                rc = upload(&dw, wp);
                L.d_w = dw;
            }
            // ring packing [panel][tap][n_alloc][64] (conv_ring_h16.hip): a panel is 64 input channels, or 32 input
            // channels as [hi x 32 | lo x 32] with lo = round(w - hi) in split precision
            L.ring_panels = x3 ? (L.c_in + 31) / 32 : (L.cp_in + 63) / 64;
            // split precision, a last panel of at most 8 channels behind 2 ... 4 full ones (67 = 64 + 3, 100 = 96 + 4): its three
            // taps become one K step (conv_ring_h16.hip: TAIL).  Not where another kernel reads the same packing (the 8-bit
            // kernel's split-precision input, the weights-resident kernel): a layer's bits must not depend on who runs it.
            L.ring_tail = x3 && m->hooks.x3_tail && !L.f8_in && !L.f8_out && L.c_in % 32 >= 1 && L.c_in % 32 <= 8 &&
                          L.ring_panels >= 3 && L.ring_panels <= 5;
            if (rc == RS_OK && L.f8_in) {
                // F8 rows (conv_ring_f8.hip): per 64 input channels an H panel (hi16 x 64) and an F panel of e4m3 bytes
                // [lo8 c0-31 | hi8 c0-31 | lo8 c32-63 | hi8 c32-63], hi8 = e4m3(hi 2^-6), lo8 = e4m3((w - hi) 2^5)
                L.ring_panels = 2 * ((L.c_in + 63) / 64);
                std::vector<unsigned short> wr((size_t)L.ring_panels * 3 * p.n_alloc * 64, 0);
                unsigned char* wb = reinterpret_cast<unsigned char*>(wr.data());
                for (int n = 0; n < L.c_out; ++n)
                    for (int ci = 0; ci < L.c_in; ++ci)
                        for (int kw = 0; kw < 3; ++kw) {
                            const float wv = conv_w[i][((size_t)n * L.c_in + ci) * 3 + kw] * ws;
                            const unsigned short hi = to_h16(wv, st16);
                            const float hf = from_h16(hi, st16);
                            const int pn = ci / 64, cc = ci - pn * 64;
                            wr[(((size_t)(2 * pn) * 3 + kw) * p.n_alloc + n) * 64 + cc] = hi;
                            const size_t fb = ((((size_t)(2 * pn + 1) * 3 + kw) * p.n_alloc + n) * 64) * 2 + (cc >> 5) * 64 + (cc & 31);
                            wb[fb] = to_e4m3(ldexpf(wv - hf, 5));
                            wb[fb + 32] = to_e4m3(ldexpf(hf, -6));
                        }
                unsigned short* dw2 = nullptr;
                rc = upload(&dw2, wr);
                L.d_w2 = dw2;
            } else if (rc == RS_OK) {
                std::vector<unsigned short> wr((size_t)L.ring_panels * 3 * p.n_alloc * 64, 0);
                for (int n = 0; n < L.c_out; ++n)
                    for (int ci = 0; ci < L.c_in; ++ci)
                        for (int kw = 0; kw < 3; ++kw) {
                            const float wv = conv_w[i][((size_t)n * L.c_in + ci) * 3 + kw] * ws;
                            const unsigned short hi = to_h16(wv, st16);
                            if (x3) {
                                const int pn = ci / 32, cc = ci - pn * 32;
                                // the merged tail slab sits in the last panel's tap-0 place: K group kw = tap kw's 8 channel slots
                                const size_t at = L.ring_tail && pn == L.ring_panels - 1
                                                      ? (((size_t)pn * 3) * p.n_alloc + n) * 64 + 8 * kw + cc
                                                      : (((size_t)pn * 3 + kw) * p.n_alloc + n) * 64 + cc;
                                wr[at] = hi;
                                wr[at + 32] = to_h16(wv - from_h16(hi, st16), st16);
                            } else {
                                const int pn = ci / 64, cc = ci - pn * 64;
                                wr[(((size_t)pn * 3 + kw) * p.n_alloc + n) * 64 + cc] = hi;
                            }
                        }
                unsigned short* dw2 = nullptr;
                rc = upload(&dw2, wr);
                L.d_w2 = dw2;
            }
        }
        const ConvPlan& p = L.plan;
        std::vector<float> bp((size_t)p.n_alloc, 0.0f);
        for (int n = 0; n < L.c_out; ++n) bp[n] = conv_b[i][n];
        if (rc == RS_OK) rc = upload(&L.d_bias, bp);
    }
    // every read's slot must start on a Winograd group boundary in every F(4,3) layer (P0 >> i divisible by 4):
    // then the grouping of a read's rows - and with it every rounding - is the same wherever the read sits in a
    // batch and whatever the batch's longest read is (results are bit-identical across batch compositions)
    // ... and the packed layout's block is never smaller than 4096 samples (shallow nets: several rows of the last
    // buffer per block), so a read spans a handful of blocks whatever the depth
    m->pad_shift = std::max(n_layers, 12);
    for (int i = 1; i < n_layers; ++i)
        if (m->layers[i].wino_m == 4) m->pad_shift = std::max(m->pad_shift, i + 2);
    // Two-level layout: the last three layers (and the head) need the coarse blocks - their launches are one round of
    // tiles at a ReadUntil batch whatever the row count - everything before them runs on blocks a quarter the size (1024
    // samples for the shipped net: a live 8615-sample read occupies 9216 samples of rows there instead of 12288).  The
    // fine block must give layer split - 1 a whole output row per block, every F(4,3) layer below the split its group
    // alignment, and the streaming kernels of layers 0-2 their 32-row steps.
    m->split = n_layers;
    m->fine_shift = m->pad_shift;
    if (!m->hooks.one_level && n_layers >= 6) {
        const int split = n_layers - 3;
        int fs = std::max(split + 1, 8);
        for (int i = 1; i < split; ++i)
            if (m->layers[i].wino_m == 4) fs = std::max(fs, i + 2);
        if (fs < m->pad_shift) {
            m->split = split;
            m->fine_shift = fs;
        }
    }
    if (rc == RS_OK) rc = upload(&m->d_zero, std::vector<float>(64, 0.0f));
    if (rc == RS_OK) {
        const int cl = channels[n_layers - 1];
        rc = upload(&m->d_fcw, std::vector<float>(fc_w, fc_w + 2 * (size_t)cl));
        if (rc == RS_OK) rc = upload(&m->d_fcb, std::vector<float>(fc_b, fc_b + 2));
    }
    if (rc != RS_OK) {
        char keep[512];
        strncpy(keep, g_err, sizeof(keep));
        keep[sizeof(keep) - 1] = 0;
        rs_model_destroy(m);
        set_error("%s", keep);
        return rc;
    }
    *out = m;
    return RS_OK;
}

int rs_model_set_fc_classifier(rs_model* m, int positions, int hidden, const float* w1, const float* b1, const float* w2,
                               const float* b2) {
    if (!m || !w1 || !b1 || !w2 || !b2) {
        set_error("rs_model_set_fc_classifier: null argument");
        return RS_ERR_ARG;
    }
    if (m->dtype != RS_F32 && m->dtype != RS_F32W) {
        set_error("rs_model_set_fc_classifier: fp32 models only (RS_F32 / RS_F32W)");
        return RS_ERR_ARG;
    }
    if (m->fc.H) {
        set_error("rs_model_set_fc_classifier: already set");
        return RS_ERR_ARG;
    }
    const int C = m->channels[m->n_layers - 1], C4 = round_up(C, 4);
    if (positions < 1 || hidden < 64 || hidden % 64 != 0 || C4 > m->cp[m->n_layers - 1] ||
        (int64_t)positions << m->n_layers > kMaxNormLen) {
        set_error("rs_model_set_fc_classifier: positions %d / hidden %d not supported (hidden: a multiple of 64)", positions, hidden);
        return RS_ERR_ARG;
    }
    DeviceGuard guard(m->device);
    const size_t F = (size_t)C * positions;
    float* d_raw = nullptr;
    RS_HIP(hipMalloc(reinterpret_cast<void**>(&d_raw), F * hidden * sizeof(float)));
    int rc = RS_OK;
    FcHead fc;
    fc.C = C;
    fc.C4 = C4;
    fc.P = positions;
    auto fail = [&](hipError_t e) {
        set_error("rs_model_set_fc_classifier: %s", hipGetErrorString(e));
        rc = RS_ERR_HIP;
    };
    hipError_t e = hipMemcpy(d_raw, w1, F * hidden * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) fail(e);
    if (rc == RS_OK && (e = hipMalloc(reinterpret_cast<void**>(&fc.d_w1p), (size_t)positions * C4 * hidden * sizeof(float))) != hipSuccess)
        fail(e);
    if (rc == RS_OK) rc = launch_fc_pack(d_raw, fc.d_w1p, C, C4, positions, hidden, nullptr);
    if (rc == RS_OK && (e = hipDeviceSynchronize()) != hipSuccess) fail(e);
    (void)hipFree(d_raw);
    if (rc == RS_OK) rc = upload(&fc.d_b1, std::vector<float>(b1, b1 + hidden));
    if (rc == RS_OK) rc = upload(&fc.d_w2, std::vector<float>(w2, w2 + 2 * (size_t)hidden));
    if (rc == RS_OK) rc = upload(&fc.d_b2, std::vector<float>(b2, b2 + 2));
    if (rc != RS_OK) {
        if (fc.d_w1p) (void)hipFree(fc.d_w1p);
        if (fc.d_b1) (void)hipFree(fc.d_b1);
        if (fc.d_w2) (void)hipFree(fc.d_w2);
        if (fc.d_b2) (void)hipFree(fc.d_b2);
        return rc;
    }
    fc.H = hidden;
    m->fc = fc;
    return RS_OK;
}

int rs_model_destroy(rs_model* m) {
    if (!m) return RS_OK;
    DeviceGuard guard(m->device);
    if (m->d_w0) (void)hipFree(m->d_w0);
    for (int i = 1; i < m->n_layers; ++i) {
        if (m->layers[i].d_w) (void)hipFree(m->layers[i].d_w);
        if (m->layers[i].d_w2) (void)hipFree(m->layers[i].d_w2);
        if (m->layers[i].d_bias) (void)hipFree(m->layers[i].d_bias);
    }
    if (m->d_fcw) (void)hipFree(m->d_fcw);
    if (m->d_fcb) (void)hipFree(m->d_fcb);
    if (m->d_zero) (void)hipFree(m->d_zero);
    if (m->d_sat) (void)hipFree(m->d_sat);
    if (m->fc.d_w1p) (void)hipFree(m->fc.d_w1p);
    if (m->fc.d_b1) (void)hipFree(m->fc.d_b1);
    if (m->fc.d_w2) (void)hipFree(m->fc.d_w2);
    if (m->fc.d_b2) (void)hipFree(m->fc.d_b2);
    if (m->side_stream) (void)hipStreamDestroy(m->side_stream);
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    for (hipEvent_t e : m->ev_pool) (void)hipEventDestroy(e);
    delete m;
    return RS_OK;
}

int rs_padded_length(const rs_model* m, int Lmax) {
    if (!m || Lmax < 1) return 0;
    return ((Lmax >> m->pad_shift) + 1) << m->pad_shift;
}

int rs_block_samples(const rs_model* m) { return m ? 1 << m->pad_shift : 0; }

int rs_max_batch(const rs_model* m, int Lmax) {
    if (!m || Lmax < 1) return 0;
    // every activation buffer (and the normalised signals) is addressed through a 2 GiB buffer-resource window
    const WsLayout w1 = ws_layout(m, 1, Lmax);
    const int64_t per_read = std::max<int64_t>((int64_t)(w1.bufb_off - w1.bufa_off), (int64_t)w1.nb_f_max * w1.Uf * 4);
    return (int)std::max<int64_t>(1, ((1LL << 31) - (1 << 20)) / per_read);
}

size_t rs_workspace_bytes(const rs_model* m, int B, int Lmax) {
    if (!m || B < 1 || Lmax < 1) return 0;
    return ws_layout(m, B, Lmax).total;
}

int rs_normalise(const int16_t* d_sig, const int64_t* d_off, const int32_t* d_len, int B, int Lmax, float* d_out32,
                 int64_t ld32, int32_t pad_to, double* d_out64, int64_t ld64, double* d_stats, void* stream) {
    if (B < 0 || (B > 0 && (!d_sig || !d_off || !d_len))) {
        set_error("rs_normalise: null argument");
        return RS_ERR_ARG;
    }
    if (!d_out32 && !d_out64 && !d_stats) {
        set_error("rs_normalise: no output requested");
        return RS_ERR_ARG;
    }
    if ((d_out32 && (ld32 < Lmax || pad_to > ld32)) || (d_out64 && ld64 < Lmax)) {
        set_error("rs_normalise: output pitch smaller than Lmax / pad_to");
        return RS_ERR_ARG;
    }
    return launch_normalise(d_sig, d_off, d_len, B, Lmax, d_out32, ld32, pad_to, d_out64, ld64, d_stats,
                            static_cast<hipStream_t>(stream));
}

int rs_normalise_float(const void* d_sig, int elem_bytes, const int64_t* d_off, const int32_t* d_len, int B, void* d_out,
                       int64_t ld, double* d_stats, void* stream) {
    if (B < 0 || (B > 0 && (!d_sig || !d_off || !d_len || !d_out)) || (elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8)) {
        set_error("rs_normalise_float: null argument or element size %d not 2 / 4 / 8", elem_bytes);
        return RS_ERR_ARG;
    }
    return launch_normalise_float(d_sig, elem_bytes, d_off, d_len, B, d_out, ld, d_stats, static_cast<hipStream_t>(stream));
}

static int forward_impl(rs_model* m, const float* d_x, int64_t ldx, const int32_t* d_len, int B, const Batch& bt,
                        const WsLayout& w, void* d_ws, float* d_probs, float* d_logits, void* stream, bool packed_x);

static int check_call(const char* who, const rs_model* m, int B, int Lmax, const WsLayout& w, size_t ws_bytes) {
    if (ws_bytes < w.total) {
        set_error("%s: workspace %zu < required %zu", who, ws_bytes, w.total);
        return RS_ERR_WORKSPACE;
    }
    if (w.nb_max * (w.U / 2) > 0x7fffffffLL || w.nb_f_max * (w.Uf / 2) > 0x7fffffffLL) {
        set_error("%s: batch too large, split it (%d reads of up to %d samples)", who, B, Lmax);
        return RS_ERR_ARG;
    }
    return RS_OK;
}

int rs_forward(rs_model* m, const float* d_x, int64_t ldx, const int32_t* d_len, const int32_t* h_len, int B, int Lmin,
               int Lmax, void* d_ws, size_t ws_bytes, float* d_probs, float* d_logits, void* stream) {
    if (!m || !d_x || !d_len || !d_ws || !d_probs || B < 1) {
        set_error("rs_forward: null argument or empty batch");
        return RS_ERR_ARG;
    }
    if (Lmax < (1 << m->n_layers)) {
        set_error("rs_forward: Lmax %d shorter than the network minimum %d", Lmax, 1 << m->n_layers);
        return RS_ERR_LENGTH;
    }
    if (ldx < Lmax) {
        set_error("rs_forward: ldx %lld < Lmax %d", (long long)ldx, Lmax);
        return RS_ERR_ARG;
    }
    const WsLayout w = ws_layout(m, B, Lmax);
    int rc = check_call("rs_forward", m, B, Lmax, w, ws_bytes);
    if (rc != RS_OK) return rc;
    Batch bt;
    rc = make_batch(m, w, d_ws, h_len, B, Lmin, Lmax, &bt);
    if (rc != RS_OK) return rc;
    DeviceGuard guard(m->device);         // launches go to the model's device whatever the caller's current one is
    RS_HIP(guard.err);
    if (!m->prof_open) prof_mark(m, -1, static_cast<hipStream_t>(stream));
    rc = launch_plan(d_len, B, Lmax, bt.plan, static_cast<hipStream_t>(stream));
    if (rc == RS_OK && two_level(m)) rc = launch_plan(d_len, B, Lmax, bt.fine, static_cast<hipStream_t>(stream));
    if (rc != RS_OK) return rc;
    return forward_impl(m, d_x, ldx, d_len, B, bt, w, d_ws, d_probs, d_logits, stream, false);
}

// The conv stack + head on a planned batch.  packed_x: d_x is the workspace's own normalised-signal region in the packed
// block layout behind 16 zero bytes (rs_classify); otherwise rows of ldx floats, one per read (rs_forward).
static int forward_impl(rs_model* m, const float* d_x, int64_t ldx, const int32_t* d_len, int B, const Batch& bt,
                        const WsLayout& w, void* d_ws, float* d_probs, float* d_logits, void* stream, bool packed_x) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(d_ws);
    void* buf[2] = {ws + w.bufa_off, ws + w.bufb_off};
    // from here on the kernels see NB blocks of U samples as NB reads in slots of U (common.hpp: BlockPlan)
    // the layout a conv layer reads: fine blocks below the split, coarse ones from it on (one level: the same table)
    const bool two = two_level(m);
    auto fine_layer = [&](int i) { return two && i < m->split; };
    const int U0 = w.Uf;                                          // block size of the normalised rows and of layers 0, 1

    // Winograd fp32 path: ConvNet layer 0 (one input channel) is folded into the staging of layer 1
    // when the signal rows are in the packed layout (always true via rs_classify)
    // (either into the streaming kernel of layers 0 + 1, which only needs 64-row blocks, or - RS_NO_STREAM_F32 - into the tiled
    // kernel's staging, which needs a tile to span at most two blocks)
    const bool stream32_l1 = packed_x && m->dtype == RS_F32W && !m->hooks.no_fuse0 &&
                             conv_stream_f32_ok(m->layers[1], m->channels[0], U0 >> 1);
    const bool fuse0 = stream32_l1 || (packed_x && m->dtype == RS_F32W && conv_wino_can_fuse0(m->layers[1], U0 >> 1));
    // 16-bit paths: the narrow layers 1 and 2 run the per-wave streaming kernel; on the rs_classify path
    // layer 0 is folded into layer 1 there as well ("fused preprocess + conv")
    const bool x3 = is_x3(m->dtype);
    const bool is16 = m->dtype == RS_BF16 || m->dtype == RS_F16;          // plain 16-bit
    const bool f16 = is_f16_family(m->dtype);
    const bool fuse0h = (is16 || x3) && packed_x && m->channels[0] <= 32 && conv_stream_h16_ok(m->layers[1], U0 >> 1);
    int rc = RS_OK;
    // unfused layer 0: ldx < 0 tells the kernel that read b's samples start at block rbase[b] of d_x
    if (!fuse0 && !fuse0h)
        rc = launch_conv0(d_x, packed_x ? -1 : ldx, d_len, bt.fine, bt.NBf, m->d_w0, m->cp[0], buf[0], act_dtype(m), st, m->d_sat);
    if (rc != RS_OK) return rc;
    prof_mark(m, 1, st);
    int cur = 0;
    for (int i = 1; i < m->n_layers; ++i) {
        ConvLayerDev& L = m->layers[i];
        // layer split - 1 wrote its rows on fine blocks; the late layers read coarse ones: a re-pack of that (small) buffer -
        // per read, its rows in order, zero rows up to the end of its coarse blocks - into the other buffer.  Not needed when
        // every read fills its coarse blocks with fine ones (16000, 12000, 8000 samples: 16 / 12 / 8 blocks of 1024 = 4 / 3 / 2
        // of 4096): then the two layouts put every row in the same place.
        if (two && i == m->split && (int64_t)bt.NBf * w.Uf != (int64_t)bt.NB * w.U) {
            const size_t row_bytes = (size_t)m->cp[i - 1] * esize(m);
            rc = launch_repack_rows(buf[cur], buf[cur ^ 1], bt.fine, bt.plan, bt.NB, w.Uf >> i, w.U >> i, row_bytes, st);
            if (rc != RS_OK) return rc;
            if (L.f8_in) {                                            // F8 rows: their scale plane moves with them
                const int64_t rows_f = (int64_t)bt.NBf * (w.Uf >> i), rows_c = (int64_t)bt.NB * (w.U >> i);
                rc = launch_repack_scales(static_cast<const char*>(buf[cur]) + f8_scale_offset(rows_f, L.cp_in),
                                          static_cast<char*>(buf[cur ^ 1]) + f8_scale_offset(rows_c, L.cp_in), bt.fine, bt.plan, bt.NB,
                                          w.Uf >> i, w.U >> i, L.cp_in / 128, f8_scale_stride(rows_f), f8_scale_stride(rows_c), st);
                if (rc != RS_OK) return rc;
            }
            cur ^= 1;
        }
        const bool fine = fine_layer(i);
        const int U = fine ? w.Uf : w.U;
        const int NB_all = fine ? bt.NBf : bt.NB;
        const int32_t* d_blen = fine ? bt.fine.blen : bt.plan.blen;
        const int Lmin = fine ? bt.Lmin_blk_f : bt.Lmin_blk;
        const int P_in = U >> i;
        // RS_EMU_ROWS (timing experiments only, results WRONG): run this layer on a share of the blocks, to price a layout
        // with fewer rows before building it (compact rows)
        int NB = NB_all;
        if (*m->hooks.emu_rows) {
            int l, pm;
            for (const char* q = m->hooks.emu_rows; q && *q; q = strchr(q, ';') ? strchr(q, ';') + 1 : nullptr)
                if (sscanf(q, "%d:%d", &l, &pm) == 2 && l == i) NB = std::max(1, (int)((int64_t)NB_all * pm / 1000));
        }
        if (i == 1 && fuse0h && m->n_layers > 2 && !(m->dbg_dst && m->dbg_layer <= 2) &&
            conv_stream012_h16_ok(L, m->layers[2], m->channels[0], P_in)) {
            // layers 0 + 1 + 2 of the 16-bit modes in one streaming kernel; its output takes the place of layer 2's
            rc = launch_conv_stream012_h16(L, m->layers[2], d_x, m->d_w0, m->channels[0], buf[cur], d_blen, NB, P_in, m->num_cu,
                                           f16, x3, st);
            if (rc != RS_OK) return rc;
            for (int k = 1; k <= 2; ++k) {
                m->last_ring[k] = false;
                m->last_bm[k] = 16;
                m->last_bn[k] = round_up(m->layers[k].c_out, 16);
                prof_mark(m, 1 + k, st);
            }
            i = 2;
            continue;
        }
        // a tile of >= 64 rows can only be all padding if some block leaves >= 64 rows unused at this layer;
        // Lmin == 0 means "unknown": keep the test
        const int check_dead = (Lmin <= 0 || (U >> i) - (Lmin >> i) >= 64) ? 1 : 0;
        // kind of kernel this layer runs: 0 streaming (not tuned), 1 F(4,3), 2 F(2,3), 3 direct fp32, 4 tiled 16-bit
        const bool stream32 = i == 1 && stream32_l1;
        const bool stream16 = (is16 || x3) && i <= 2 && conv_stream_h16_ok(L, P_in);
        // every tiled 16-bit layer runs the LDS-DMA ring kernel (the register-staged kernel of round 1, conv_h16.hip, was its
        // bit-for-bit cross-check through round 3 and has been removed)
        const bool ring = !stream16 && (x3 || is16);
        // RS_F16XF8: the wide layers read and / or write F8 rows (cross terms on the 8-bit MFMA: conv_ring_f8.hip)
        const bool f8 = ring && (L.f8_in || L.f8_out);
        // ... and narrow layers whose whole weight tensor fits LDS next to two activation slabs on the weights-resident kernel
        // split precision on a launch of a few rows (Model.classify at batch 1, a thin ReadUntil batch): 64 x 32 tiles that take a
        // WHOLE PANEL per barrier instead of the ring's (panel, tap) sub-stages, each as long as a staging round trip whatever
        // the tile holds (conv_thin_h16.hip; same bits)
        bool thin16 = ring && x3 && !f8 && !m->tuning && m->hooks.thin_h16_rows != 0 && conv_thin_h16_ok(L);
        if (thin16) {
            const int64_t rows_in = (int64_t)NB * P_in;
            if (m->hooks.thin_h16_rows > 0)
                thin16 = rows_in <= m->hooks.thin_h16_rows;
            else
                thin16 = conv_thin_h16_cost(L, rows_in, m->num_cu) < conv_ring_plan_cost(L, rows_in, m->num_cu, x3);
            if (m->hooks.tail_debug)
                fprintf(stderr, "[thin-or-ring] layer %d: rows %lld, thin %.0f (%lld tiles), ring %.0f -> %s\n", i, (long long)rows_in,
                        conv_thin_h16_cost(L, rows_in, m->num_cu), (long long)conv_thin_h16_tiles(L, rows_in),
                        conv_ring_plan_cost(L, rows_in, m->num_cu, x3), thin16 ? "thin" : "ring");
        }
        const bool wres = ring && !f8 && !thin16 && conv_wres_h16_ok(L, x3);
        // fp32 Winograd layers of a launch with only a handful of rows (Model.classify at batch 1, a thin ReadUntil batch):
        // one wave per 16 x 16 tile instead of 256-row tiles that are mostly padding (conv_small_f32.hip; same bits)
        // (not layer 1 when layer 0 is folded into its staging: nothing has written that layer's input)
        bool small32 = m->dtype == RS_F32W && !stream32 && !(fuse0 && i == 1) && !m->tuning && m->hooks.small_f32_waves != 0 &&
                       conv_small_f32_ok(L);
        if (small32) {
            const int64_t rows_in = (int64_t)NB * P_in;
            if (m->hooks.small_f32_waves > 0)                           // forced limit (tests, A/B runs)
                small32 = conv_small_f32_waves(L, rows_in) <= m->hooks.small_f32_waves;
            else {
                // the launch planner's own estimate of the tiled kernel against the small kernel's (both in cycles, both
                // rough): take the small kernel where it is clearly ahead
                const int n16 = round_up(L.c_out, 16) / 16;
                bool thin_fit = false;                               // the tiled estimate is the thin-launch fit (with its launch cost: like the small kernel's)
                const double tiled = L.wino_m == 4
                    ? conv_wino4_launch_cost((rows_in + 3) / 4, n16, L.plan.kc, L.plan.nch, m->num_cu, &thin_fit)
                    : conv_wino_launch_cost(rows_in / 2, n16, L.plan.kc, L.plan.nch, m->num_cu, &thin_fit);
                small32 = conv_small_f32_waves(L, rows_in) <= 4096 &&
                          conv_small_f32_cost(L, rows_in, m->num_cu) < (thin_fit ? 1.0 : 0.8) * tiled;
                if (m->hooks.tail_debug)
                    fprintf(stderr, "[small-or-tiled] layer %d: rows %lld, small %.0f (%lld workgroups), tiled %.0f (%s) -> %s\n", i,
                            (long long)rows_in, conv_small_f32_cost(L, rows_in, m->num_cu), (long long)conv_small_f32_waves(L, rows_in),
                            tiled, thin_fit ? "thin fit + launch" : "full-launch model", small32 ? "small" : "tiled");
            }
        }
        const int kind = (stream32 || stream16 || wres || small32 || thin16) ? 0 : f8 ? 6 : ring ? 5 : m->dtype == RS_F32W ? (L.wino_m == 4 ? 1 : 2)
                                                                    : m->dtype == RS_F32 ? 3 : 4;
        m->last_ring[i] = ring;
        auto launch_layer = [&]() -> int {
            int rc;
            if (stream32) {
                rc = launch_conv_stream_f32(L, d_x, m->d_w0, m->channels[0], static_cast<float*>(buf[cur ^ 1]), d_blen, NB, P_in,
                                            m->num_cu, st);
                m->last_bm[i] = 32;
                m->last_bn[i] = round_up(L.c_out, 16);
            } else if (small32)
                rc = launch_conv_small_f32(L, static_cast<const float*>(buf[cur]), static_cast<float*>(buf[cur ^ 1]), d_blen, NB, P_in,
                                           i, m->num_cu, st, &m->last_bm[i], &m->last_bn[i]);
            else if (m->dtype == RS_F32W && L.wino_m == 4)
                rc = launch_conv_wino4(L, static_cast<const float*>(buf[cur]), static_cast<float*>(buf[cur ^ 1]), d_blen, NB,
                                       P_in, i, m->num_cu, check_dead, st, &m->last_bm[i], &m->last_bn[i]);
            else if (m->dtype == RS_F32W)
                rc = launch_conv_wino(L, static_cast<const float*>(buf[cur]), static_cast<float*>(buf[cur ^ 1]), d_blen,
                                      NB, P_in, i, m->num_cu, m->d_zero, check_dead, st, &m->last_bm[i], &m->last_bn[i],
                                      (fuse0 && i == 1) ? d_x : nullptr, m->d_w0);
            else if (m->dtype == RS_F32)
                rc = launch_conv_f32(L, static_cast<const float*>(buf[cur]), static_cast<float*>(buf[cur ^ 1]), d_blen,
                                     NB, P_in, i, m->num_cu, m->d_zero, check_dead, st, &m->last_bm[i], &m->last_bn[i]);
            else if (stream16) {
                const bool f0 = fuse0h && i == 1;
                rc = launch_conv_stream_h16(L, buf[cur], buf[cur ^ 1], d_blen, NB, P_in, i, m->num_cu, f16, st,
                                            f0 ? d_x : nullptr, m->d_w0, m->channels[0], x3);
                m->last_bm[i] = 16;
                m->last_bn[i] = round_up(L.c_out, 16);
            } else if (thin16)
                rc = launch_conv_thin_h16(L, buf[cur], buf[cur ^ 1], d_blen, NB, P_in, i, m->num_cu, f16, st, &m->last_bm[i], &m->last_bn[i]);
            else if (f8)
                rc = launch_conv_ring_f8(L, buf[cur], buf[cur ^ 1], d_blen, NB, P_in, i, m->num_cu, check_dead, st, &m->last_bm[i],
                                         &m->last_bn[i]);
            else if (wres)
                rc = launch_conv_wres_h16(L, buf[cur], buf[cur ^ 1], d_blen, NB, P_in, i, m->num_cu, f16, x3, check_dead, st,
                                          &m->last_bm[i], &m->last_bn[i]);
            else if (ring)
                rc = launch_conv_ring_h16(L, buf[cur], buf[cur ^ 1], d_blen, NB, P_in, i, m->num_cu, f16, x3, check_dead, st,
                                          &m->last_bm[i], &m->last_bn[i]);
            else {
                set_error("no kernel for layer %d in this mode", i);
                rc = RS_ERR_ARG;
            }
            return rc;
        };
        if (m->tuning && (kind == 1 || kind == 2 || kind == 5 || kind == 6)) {
            // rs_autotune: every feasible entry of the kernel's shape table on THIS layer's real input (the buffers hold the
            // activations of the batch; re-running a layer rewrites the same output), 1 warm + 3 timed launches each;
            // a shape replaces the planner's choice only if it is > 3 % faster
            const int n = kind == 1 ? conv_wino4_num_shapes() : kind == 2 ? conv_wino_num_shapes()
                        : kind == 6 ? conv_ring_f8_num_shapes() : conv_ring_num_shapes();
            auto ok = [&](int k) {
                return kind == 1 ? conv_wino4_shape_ok(L, k) : kind == 2 ? conv_wino_shape_ok(L, k)
                     : kind == 6 ? conv_ring_f8_shape_ok(L, k) : conv_ring_shape_ok(L, k);
            };
            hipEvent_t e0, e1;
            RS_HIP(hipEventCreate(&e0));
            RS_HIP(hipEventCreate(&e1));
            auto timed = [&](int k, float* ms) -> int {
                L.force_shape = k;
                int r = launch_layer();
                if (r == RS_OK) r = hipEventRecord(e0, st) == hipSuccess ? RS_OK : RS_ERR_HIP;
                for (int rep = 0; rep < 3 && r == RS_OK; ++rep) r = launch_layer();
                if (r == RS_OK) r = hipEventRecord(e1, st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess ? RS_OK : RS_ERR_HIP;
                if (r == RS_OK) r = hipEventElapsedTime(ms, e0, e1) == hipSuccess ? RS_OK : RS_ERR_HIP;
                L.force_shape = -1;
                return r;
            };
            const int64_t rows = (int64_t)NB * P_in;
            for (size_t t = 0; t < L.tuned.size(); ++t)                       // re-tuning a geometry: forget the old entry
                if (L.tuned[t].first == rows) L.tuned.erase(L.tuned.begin() + t--);
            float base_ms = 0.f, best_ms = 1e30f;
            int best_k = -1;
            rc = timed(-1, &base_ms);                                          // the planner's own choice
            for (int k = 0; k < n && rc == RS_OK; ++k) {
                if (!ok(k)) continue;
                float ms = 0.f;
                rc = timed(k, &ms);
                if (rc == RS_OK && ms < best_ms) {
                    best_ms = ms;
                    best_k = k;
                }
            }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            if (rc != RS_OK) return rc;
            if (best_k >= 0 && best_ms < 0.97f * base_ms) {
                L.tuned.emplace_back(rows, best_k);
                ++m->tuned_changed;
            }
        }
        rc = launch_layer();
        if (rc != RS_OK) return rc;
        prof_mark(m, 1 + i, st);
        if (m->dbg_dst && m->dbg_layer == i) {
            const int64_t rows_out = (int64_t)NB * (P_in / 2);          // F8 rows: with the scale plane behind them
            const size_t all = L.f8_out ? f8_scale_offset(rows_out, L.cp_out) + f8_scale_bytes(rows_out, L.cp_out)
                                        : (size_t)rows_out * L.cp_out * esize(m);
            const size_t nb = std::min(m->dbg_bytes, all);
            RS_HIP(hipMemcpyAsync(m->dbg_dst, buf[cur ^ 1], nb, hipMemcpyDeviceToDevice, st));
        }
        cur ^= 1;
    }
    if (m->fc.H)
        rc = launch_fc_head(static_cast<const float*>(buf[cur]), m->cp[m->n_layers - 1], w.U >> m->n_layers, m->n_layers, d_len,
                            B, bt.plan, m->fc, reinterpret_cast<float*>(static_cast<char*>(d_ws) + w.fc_part_off), d_probs,
                            d_logits, st);
    else
        rc = launch_head(buf[cur], act_dtype(m), m->cp[m->n_layers - 1], m->channels[m->n_layers - 1],
                         w.U >> m->n_layers, m->n_layers, d_len, B, bt.plan, two ? &bt.fine : nullptr, m->d_fcw, m->d_fcb,
                         d_probs, d_logits, st);
    if (rc == RS_OK) prof_mark(m, m->n_layers + 1, st);
    return rc;
}

// normalise + block plan of one batch into the workspace (the first launch of rs_classify / rs_classify_ensemble)
static int normalise_packed(rs_model* m, const int16_t* d_sig, const int64_t* d_off, const int32_t* d_len, int B, int Lmax,
                            const WsLayout& w, const Batch& bt, void* d_ws, hipStream_t st) {
    float* xn = reinterpret_cast<float*>(static_cast<char*>(d_ws) + w.xnorm_off);
    return launch_normalise(d_sig, d_off, d_len, B, Lmax, xn, 0, 0, nullptr, 0, nullptr, st, /*zero_prefix=*/1, &bt.fine,
                            two_level(m) ? &bt.plan : nullptr);
}

int rs_classify(rs_model* m, const int16_t* d_sig, const int64_t* d_off, const int32_t* d_len, const int32_t* h_len, int B,
                int Lmin, int Lmax, void* d_ws, size_t ws_bytes, float* d_probs, float* d_logits, void* stream) {
    if (!m || !d_sig || !d_off || !d_len || !d_ws || !d_probs || B < 1) {
        set_error("rs_classify: null argument or empty batch");
        return RS_ERR_ARG;
    }
    if (Lmax < (1 << m->n_layers) || Lmax > kMaxNormLen) {
        set_error("rs_classify: Lmax %d outside [%d, %d]", Lmax, 1 << m->n_layers, kMaxNormLen);
        return RS_ERR_LENGTH;
    }
    const WsLayout w = ws_layout(m, B, Lmax);
    int rc = check_call("rs_classify", m, B, Lmax, w, ws_bytes);
    if (rc != RS_OK) return rc;
    Batch bt;
    rc = make_batch(m, w, d_ws, h_len, B, Lmin, Lmax, &bt);
    if (rc != RS_OK) return rc;
    DeviceGuard guard(m->device);
    RS_HIP(guard.err);
    hipStream_t st = static_cast<hipStream_t>(stream);
    prof_mark(m, -1, st);
    rc = normalise_packed(m, d_sig, d_off, d_len, B, Lmax, w, bt, d_ws, st);
    if (rc != RS_OK) return rc;
    prof_mark(m, 0, st);
    const float* xn = reinterpret_cast<const float*>(static_cast<char*>(d_ws) + w.xnorm_off);
    return forward_impl(m, xn, w.Uf, d_len, B, bt, w, d_ws, d_probs, d_logits, stream, true);
}

int rs_autotune(rs_model* m, const int16_t* d_sig, const int64_t* d_off, const int32_t* d_len, const int32_t* h_len, int B,
                int Lmin, int Lmax, void* d_ws, size_t ws_bytes, float* d_probs, int32_t* n_changed, void* stream) {
    if (!m) {
        set_error("rs_autotune: null model");
        return RS_ERR_ARG;
    }
    const bool prof = m->prof_on;
    m->prof_on = false;
    m->tuning = true;
    m->tuned_changed = 0;
    const int rc = rs_classify(m, d_sig, d_off, d_len, h_len, B, Lmin, Lmax, d_ws, ws_bytes, d_probs, nullptr, stream);
    m->tuning = false;
    m->prof_on = prof;
    if (n_changed) *n_changed = m->tuned_changed;
    return rc;
}

// the layout all models of an ensemble share: block table and normalised signals once, activation buffers sized for the
// widest model
static WsLayout ensemble_layout(rs_model* const* models, int n_models, int B, int Lmax) {
    WsLayout w = ws_layout(models[0], B, Lmax);
    for (int k = 1; k < n_models; ++k) {
        const WsLayout wk = ws_layout(models[k], B, Lmax);
        if (wk.bufb_off - wk.bufa_off > w.bufb_off - w.bufa_off) w = wk;
    }
    return w;
}

static bool ensemble_compatible(rs_model* const* models, int n_models) {
    if (!models || n_models < 1 || !models[0]) return false;
    const rs_model* m0 = models[0];
    for (int k = 0; k < n_models; ++k)
        if (!models[k] || models[k]->n_layers != m0->n_layers || models[k]->device != m0->device ||
            esize(models[k]) != esize(m0) || models[k]->pad_shift != m0->pad_shift || models[k]->fine_shift != m0->fine_shift ||
            models[k]->split != m0->split || models[k]->fc.H != m0->fc.H)
            return false;
    return true;
}

size_t rs_ensemble_workspace_bytes(rs_model* const* models, int n_models, int B, int Lmax) {
    if (!ensemble_compatible(models, n_models) || B < 1 || Lmax < 1 || models[0]->fc.H) return 0;
    const WsLayout w = ensemble_layout(models, n_models, B, Lmax);
    return w.bufa_off + (size_t)n_models * 2 * (w.bufb_off - w.bufa_off);
}

int rs_classify_ensemble(rs_model* const* models, int n_models, const int16_t* d_sig, const int64_t* d_off,
                         const int32_t* d_len, const int32_t* h_len, int B, int Lmin, int Lmax, void* d_ws, size_t ws_bytes,
                         float* d_probs, uint8_t* d_decision, int max_len, float threshold, int mode, void* stream) {
    if (!models || n_models < 1 || !d_sig || !d_off || !d_len || !d_ws || !d_probs || B < 1) {
        set_error("rs_classify_ensemble: null argument or empty batch");
        return RS_ERR_ARG;
    }
    if (!ensemble_compatible(models, n_models)) {
        set_error("rs_classify_ensemble: a model is null or differs in depth / device / element size");
        return RS_ERR_ARG;
    }
    rs_model* m0 = models[0];
    if (d_decision && mode != RS_ENRICH && mode != RS_DEPLETE) {
        set_error("rs_classify_ensemble: bad mode");
        return RS_ERR_ARG;
    }
    if (Lmax < (1 << m0->n_layers) || Lmax > kMaxNormLen) {
        set_error("rs_classify_ensemble: Lmax %d outside [%d, %d]", Lmax, 1 << m0->n_layers, kMaxNormLen);
        return RS_ERR_LENGTH;
    }
    // the models share one plan and one copy of the normalised signals: same block size, same element size (checked
    // above), so one layout serves them all except for the width of the activation buffers
    const WsLayout w = ensemble_layout(models, n_models, B, Lmax);
    int rc = check_call("rs_classify_ensemble", m0, B, Lmax, w, ws_bytes);
    if (rc != RS_OK) return rc;
    Batch bt;
    rc = make_batch(m0, w, d_ws, h_len, B, Lmin, Lmax, &bt);
    if (rc != RS_OK) return rc;
    DeviceGuard guard(m0->device);
    RS_HIP(guard.err);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // With a workspace of rs_ensemble_workspace_bytes() every model has its own pair of activation buffers and the
    // forwards run CONCURRENTLY: model 0 on the caller's stream, model k on its own side stream, forked behind the
    // normalise launch and joined in front of the decision.  The forwards are independent (they read the shared block table
    // and normalised rows, write their own buffers and their own slice of d_probs), so the results are those of the serial
    // order, bit for bit; what overlaps is one model's tile-round tails, prologues and launch gaps with another's tiles.
    // A smaller workspace (rs_workspace_bytes of the widest model), a profiled or a tuning model, or RS_ENSEMBLE_SERIAL
    // run the forwards back to back on the caller's stream.
    const size_t buf_bytes = w.bufb_off - w.bufa_off;
    // ... and so does a batch that fills the chip by itself: measured (tools/ensemble_probe.py, three models, same box,
    // interleaved) concurrent / serial = 2.4x at 32 reads, 1.5x at 128, 1.14x (fp32) / 1.13x (bf16x3) at the live 357 x 8615
    // batch (1071 blocks), 1.01x / 0.99x at 512 x 16000 (2048 blocks): beyond ~1800 blocks of 4096 samples one model's
    // launches leave nothing for another's to use, and the fork / join is pure overhead
    constexpr int64_t kConcurrentBelowSamples = 1800LL * 4096;
    bool concurrent = n_models > 1 && ws_bytes >= w.bufa_off + (size_t)n_models * 2 * buf_bytes &&
                      !m0->hooks.ensemble_serial && !m0->fc.H && (int64_t)bt.NB * w.U < kConcurrentBelowSamples;
    for (int k = 0; k < n_models; ++k)
        if (models[k]->prof_on || models[k]->tuning || models[k]->dbg_dst) concurrent = false;
    for (int k = 1; k < n_models && concurrent; ++k)
        for (int j = 0; j < k; ++j)
            if (models[j] == models[k]) concurrent = false;                  // one handle twice: its side stream is one
    if (concurrent)
        for (int k = 1; k < n_models; ++k) {
            rs_model* mk = models[k];
            if (!mk->side_stream) RS_HIP(hipStreamCreateWithFlags(&mk->side_stream, hipStreamNonBlocking));
            if (!mk->ev_fork) RS_HIP(hipEventCreateWithFlags(&mk->ev_fork, hipEventDisableTiming));
            if (!mk->ev_join) RS_HIP(hipEventCreateWithFlags(&mk->ev_join, hipEventDisableTiming));
        }
    prof_mark(m0, -1, st);
    rc = normalise_packed(m0, d_sig, d_off, d_len, B, Lmax, w, bt, d_ws, st);
    if (rc != RS_OK) return rc;
    prof_mark(m0, 0, st);
    const float* xn = reinterpret_cast<const float*>(static_cast<char*>(d_ws) + w.xnorm_off);
    if (concurrent) {
        // Whatever fails after the first fork, every side stream that has received work is JOINED to the caller's stream
        // before this call returns: the caller owns the workspace and d_probs and may reuse or free them behind `st` the
        // moment it sees the error code.  A join that itself fails falls back to a host-side wait.
        int forked = 0;
        auto join_all = [&]() {
            for (int k = 1; k <= forked; ++k) {
                rs_model* mk = models[k];
                if (hipEventRecord(mk->ev_join, mk->side_stream) != hipSuccess ||
                    hipStreamWaitEvent(st, mk->ev_join, 0) != hipSuccess)
                    (void)hipStreamSynchronize(mk->side_stream);
            }
        };
        for (int k = 1; k < n_models && rc == RS_OK; ++k) {
            rs_model* mk = models[k];
            hipError_t he = hipEventRecord(mk->ev_fork, st);
            if (he == hipSuccess) he = hipStreamWaitEvent(mk->side_stream, mk->ev_fork, 0);
            if (he != hipSuccess) {
                set_error("rs_classify_ensemble: fork of model %d: %s", k, hipGetErrorString(he));
                rc = RS_ERR_HIP;
                break;
            }
            forked = k;
            WsLayout wk = w;
            wk.bufa_off += (size_t)k * 2 * buf_bytes;
            wk.bufb_off += (size_t)k * 2 * buf_bytes;
            rc = forward_impl(mk, xn, w.Uf, d_len, B, bt, wk, d_ws, d_probs + (size_t)k * B * 2, nullptr, mk->side_stream, true);
        }
        if (rc == RS_OK) rc = forward_impl(m0, xn, w.Uf, d_len, B, bt, w, d_ws, d_probs, nullptr, stream, true);
        join_all();
        if (rc != RS_OK) return rc;
    } else {
        for (int k = 0; k < n_models; ++k) {
            // every model keeps the block table and the normalised rows at the head of the workspace and ping-pongs behind them
            rc = forward_impl(models[k], xn, w.Uf, d_len, B, bt, w, d_ws, d_probs + (size_t)k * B * 2, nullptr, stream, true);
            if (rc != RS_OK) return rc;
        }
    }
    if (d_decision) rc = launch_decide(d_probs, n_models, B, d_len, max_len, threshold, mode, d_decision, st);
    return rc;
}

int rs_decide(const float* d_probs, int n_models, int B, const int32_t* d_len, int max_len, float threshold,
              int mode, uint8_t* d_out, void* stream) {
    if (B < 0 || n_models < 1 || (B > 0 && (!d_probs || !d_len || !d_out)) ||
        (mode != RS_ENRICH && mode != RS_DEPLETE)) {
        set_error("rs_decide: bad argument");
        return RS_ERR_ARG;
    }
    return launch_decide(d_probs, n_models, B, d_len, max_len, threshold, mode, d_out,
                         static_cast<hipStream_t>(stream));
}

int rs_polya_end(const int16_t* d_sig, const int64_t* d_off, const int32_t* d_len, int B, int32_t* d_end,
                 void* stream) {
    if (B < 0 || (B > 0 && (!d_sig || !d_off || !d_len || !d_end))) {
        set_error("rs_polya_end: null argument");
        return RS_ERR_ARG;
    }
    return launch_polya(d_sig, d_off, d_len, B, d_end, static_cast<hipStream_t>(stream));
}

int rs_polya_end_resume(const int16_t* d_sig, const int64_t* d_off, const int32_t* d_len, int B, const int32_t* d_state_in,
                        int32_t* d_end, int32_t* d_state_out, void* stream) {
    if (B < 0 || (B > 0 && (!d_sig || !d_off || !d_len || !d_end || !d_state_out))) {
        set_error("rs_polya_end_resume: null argument");
        return RS_ERR_ARG;
    }
    if (B > 0 && d_state_in == d_state_out) {
        set_error("rs_polya_end_resume: d_state_in and d_state_out must not be the same buffer");
        return RS_ERR_ARG;
    }
    return launch_polya(d_sig, d_off, d_len, B, d_end, static_cast<hipStream_t>(stream), d_state_in, d_state_out);
}

int rs_copy_segments(const int16_t* d_src, int16_t* d_dst, const int64_t* d_src_off, const int64_t* d_dst_off,
                     const int32_t* d_len, int n, void* stream) {
    if (n < 0 || (n > 0 && (!d_src || !d_dst || !d_src_off || !d_dst_off || !d_len))) {
        set_error("rs_copy_segments: null argument");
        return RS_ERR_ARG;
    }
    return launch_copy_segments(d_src, d_dst, d_src_off, d_dst_off, d_len, n, static_cast<hipStream_t>(stream));
}

int rs_model_saturated(rs_model* m, int reset, void* stream) {
    if (!m) {
        set_error("rs_model_saturated: null model");
        return RS_ERR_ARG;
    }
    if (!m->d_sat) return 0;                                  // fp32 / bf16 modes carry fp32's exponent range
    DeviceGuard guard(m->device);
    RS_HIP(guard.err);
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned v = 0;
    RS_HIP(hipMemcpyAsync(&v, m->d_sat, sizeof(v), hipMemcpyDeviceToHost, st));
    if (reset) RS_HIP(hipMemsetAsync(m->d_sat, 0, sizeof(v), st));
    RS_HIP(hipStreamSynchronize(st));
    return v ? 1 : 0;
}

int rs_debug_capture_layer(rs_model* m, int layer, void* d_dst, size_t bytes) {
    if (!m) {
        set_error("rs_debug_capture_layer: null model");
        return RS_ERR_ARG;
    }
    m->dbg_layer = layer;
    m->dbg_dst = d_dst;
    m->dbg_bytes = bytes;
    return RS_OK;
}

int rs_profile_enable(rs_model* m, int on) {
    if (!m) {
        set_error("rs_profile_enable: null model");
        return RS_ERR_ARG;
    }
    m->prof_on = on != 0;
    m->prof_level = on == 2 ? 2 : 1;
    m->prof_open = false;
    if (!on) {
        m->ev_used = 0;
        m->prof_calls = 0;
    }
    return RS_OK;
}

int rs_profile_read(rs_model* m, float* stage_ms, int32_t* calls) {
    if (!m || !stage_ms) {
        set_error("rs_profile_read: null argument");
        return RS_ERR_ARG;
    }
    if (m->ev_used) RS_HIP(hipEventSynchronize(m->ev_pool[m->ev_used - 1]));
    for (size_t k = 1; k < m->ev_used; ++k) {
        const int stage = m->ev_stage[k];
        if (stage < 0) continue;
        float ms = 0.f;
        RS_HIP(hipEventElapsedTime(&ms, m->ev_pool[k - 1], m->ev_pool[k]));
        stage_ms[stage] += ms;
    }
    if (calls) *calls = m->prof_calls;
    m->ev_used = 0;
    m->prof_calls = 0;
    return RS_OK;
}

int rs_model_layer_info(const rs_model* m, int layer, rs_layer_info* out) {
    if (!m || !out || layer < 0 || layer >= m->n_layers) {
        set_error("rs_model_layer_info: bad argument");
        return RS_ERR_ARG;
    }
    memset(out, 0, sizeof(*out));
    out->c_out = m->channels[layer];
    out->cp_out = m->cp[layer];
    out->rows_format = (layer + 1 < m->n_layers && m->layers[layer + 1].f8_in) ? 2 : is_x3(m->dtype) ? 1 : 0;
    if (layer == 0) {
        out->c_in = 1;
        out->cp_in = 1;
        out->k_pad = 3;
        out->n_pad = m->cp[0];
        out->gemm_row_div = 1;
        out->block_samples = 1 << layer_shift(m, 0);
        return RS_OK;
    }
    const ConvLayerDev& L = m->layers[layer];
    out->c_in = L.c_in;
    out->cp_in = L.cp_in;
    out->k_pad = (m->dtype == RS_F32W ? (L.wino_m == 4 ? 6 : 4) : 3) * L.plan.kc * L.plan.nch;
    if (L.ring_tail) out->k_pad = (3 * (L.ring_panels - 1) + 1) * 32;                       // the last panel's taps share one K step
    if (m->last_ring[layer] && !is_x3(m->dtype)) out->k_pad = 3 * 64 * L.ring_panels;      // 64-channel panels
    out->n_pad = m->last_bn[layer] ? round_up(round_up(L.c_out, 16), m->last_bn[layer]) : round_up(L.c_out, 16);
    out->bm = m->last_bm[layer];
    out->bn = m->last_bn[layer];
    out->kc = L.plan.kc;
    out->gemm_row_div = m->dtype == RS_F32W ? L.wino_m : 1;
    out->block_samples = 1 << layer_shift(m, layer);
    return RS_OK;
}

}  // extern "C"
