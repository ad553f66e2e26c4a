// Shared host/device declarations for libriser_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include <utility>
#include <vector>

#include "../../include/riser_amd.h"

namespace rs {

// thread-local error text behind rs_last_error()
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define RS_HIP(call)                                                   \
    do {                                                               \
        hipError_t e__ = (call);                                       \
        if (e__ != hipSuccess) return rs::hip_fail(e__, #call);        \
    } while (0)

// Per-read lengths are read-only for every kernel: loads through the CONSTANT address space compile to
// scalar loads (s_load, tracked by lgkmcnt) when the index is wave-uniform.  A plain global load of a
// struct-member pointer becomes a VECTOR load, and its s_waitcnt vmcnt(0) drains every prefetch the
// wave has in flight.
typedef const int32_t __attribute__((address_space(4))) * const_len_ptr;
__device__ __forceinline__ const_len_ptr as_const_len(const int32_t* p) {
    return (const_len_ptr)(uintptr_t)p;
}

// Restore the caller's current HIP device on scope exit: model construction / destruction and the launch entry
// points switch to the model's device without leaking that choice into the host program (torch keeps its own
// notion of the current device; Model.__del__ may run at any point of it).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// Tuning / diagnostic switches.  The RS_* environment variables are read ONCE, when a model is
// created (rs_model_create), never on the launch path.
struct Hooks {
    bool no_rect_order = false;      // RS_NO_RECT_ORDER: n-major tile order instead of XCD rectangles
    int sf32_min_run = 0;            // RS_SF32_MIN_RUN: shortest run of 16-row sub-blocks a wave of the fp32 streaming kernel (layers 0 + 1) takes (0: by launch size)
    int small_nw = 0;                // RS_SMALL_NW: 1 / 2 = channel sub-tiles per wave of conv_small_f32 (0: by launch size)
    int small_shared = 1;            // RS_SMALL_SHARED=0: conv_small_f32 stages the input rows once per wave (round 4) instead of once per workgroup
    bool no_deep_staging = false;    // RS_NO_DEEP_STAGING: the thin fp32 Winograd shapes keep the default staging distance (A/B of the one-item-ahead loads)
    bool tail_debug = false;         // RS_TAIL_DEBUG: print every head / tail decision
    bool ring_tail_split = false;    // RS_RING_TAIL_SPLIT: head + tail launches for the 16-bit ring kernel too (measured: a wash)
    double tail_margin = 0.0;        // RS_TAIL_MARGIN: a head + tail split is taken when priced below this share of ONE launch (0: the kernels' own 0.97 / 0.92; tuning aid)
    bool no_tail_split = false;      // RS_NO_TAIL_SPLIT: tiled conv layers (fp32 Winograd, 16-bit ring) always as ONE launch (tile_walk.hpp: plan_tail_split)
    bool no_fuse0 = false;           // RS_NO_FUSE0: layer 0 as its own launch on the fp32 Winograd path
    bool no_stream_f32 = false;      // RS_NO_STREAM_F32 / _H16: tiled kernels instead of the streaming ones
    bool no_stream_h16 = false;
    bool no_stream012 = false;       // RS_NO_STREAM012: layers 0+1 and 2 of the 16-bit modes as two launches instead of one
    int small_f32_waves = -1;        // RS_SMALL_F32_WAVES: fp32 Winograd launches of at most this many 16 x 16 tiles run the
                                     // small-batch kernel (0 = never; default -1: wherever its cost estimate beats the tiled kernel's)
    bool one_level = false;          // RS_ONE_LEVEL: every layer on the coarse blocks (the round-3 layout; bit-identical results)
    bool ensemble_serial = false;    // RS_ENSEMBLE_SERIAL: rs_classify_ensemble runs its forwards back to back on the caller's stream
    bool conv_stamps = false;        // RS_CONV_STAMPS: in-kernel clock stamps of the direct fp32 kernel
    char force_f32[256] = "";        // RS_FORCE_SHAPE_F32 / _WINO / _WINO4 / _H16: "layer:wm,wn,mt,nt;..."
    char force_wino[256] = "";
    char force_wino4[256] = "";
    char force_ring[256] = "";       // RS_FORCE_SHAPE_RING
    char emu_rows[128] = "";         // RS_EMU_ROWS "layer:permille;...": TIMING ONLY - the layer runs on that share of the batch's blocks
    int thin_h16_rows = -1;          // RS_THIN_H16_ROWS: split-precision layers of a launch with at most this many input rows run the
                                     // thin-launch kernel (conv_thin_h16.hip; 0 = never; default -1: by the cost estimates)
    bool x3_tail = true;             // RS_X3_TAIL=0: split-precision layers whose last panel holds <= 8 channels keep it as three 32-wide K steps instead of ONE merged step (conv_ring_h16.hip: TAIL)
    int f8_min_cin = 200;            // RS_F8_MIN_CIN: RS_F16XF8 puts a layer on the 8-bit kernel from this many input channels on (api.hip: f8_eligible)
    bool h16_wres = true;            // RS_H16_WRES=0: narrow 16-bit layers on the ring kernel instead of the weights-resident one
    static Hooks from_env();
};
const Hooks& default_hooks();

inline bool is_x3(int dtype) { return dtype == RS_BF16X3 || dtype == RS_F16X3 || dtype == RS_F16XF8; }
inline bool is_f16_family(int dtype) { return dtype == RS_F16 || dtype == RS_F16X3 || dtype == RS_F16XF8; }
inline bool is_16bit(int dtype) { return dtype == RS_BF16 || dtype == RS_F16 || is_x3(dtype); }

constexpr int kMaxLayers = 16;
constexpr int kMaxNormLen = 65536;

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Packed block layout of the conv stack.  A batch of B reads is laid out as NB BLOCKS of
// U = 1 << shift samples: read b owns the nblk(b) = len[b] / U + 1 consecutive blocks from rbase[b] on (so at least one
// zero sample follows every read), and the conv kernels run on the blocks as if each were a read of blen[k] samples in
// a slot of U: work is proportional to every read's own length, whatever the longest read of the batch is.  The table
// lives in the caller's workspace and is written on the device (normalise_kernel, or plan_kernel when the signals
// arrive normalised); the host only needs NB, which it gets from its copy of the lengths - or, when it has none, from
// uniform_nblk blocks for every read.
struct BlockPlan {
    int32_t* rbase = nullptr;    // [B + 1] first block of read b; rbase[B] = blocks in use
    int32_t* blen = nullptr;     // [nb_total] valid samples of block k: clamp(len[b] - j * U, 0, U), j = k - rbase[b]
    int32_t* bread = nullptr;    // [nb_total] read index b of block k
    int shift = 12;              // log2 U
    int uniform_nblk = 0;        // > 0: every read takes this many blocks
    int nb_total = 0;            // blocks the workspace was laid out for
};

// ---- kernel launchers (each returns RS_OK or records an error) -------------------------
// plan != nullptr: the fp32 rows go to the packed block layout (ld32 / pad_to ignored) and the block table is written
int launch_normalise(const int16_t* d_sig, const int64_t* d_off, const int32_t* d_len, int B, int Lmax,
                     float* d_out32, int64_t ld32, int32_t pad_to, double* d_out64, int64_t ld64,
                     double* d_stats, hipStream_t st, int zero_prefix = 0, const BlockPlan* plan = nullptr,
                     const BlockPlan* plan2 = nullptr);   // plan2: a second table (the late layers' coarser blocks), no rows
// the block table alone (signals that arrive normalised: rs_forward)
int launch_plan(const int32_t* d_len, int B, int Lmax, const BlockPlan& plan, hipStream_t st);

int launch_normalise_float(const void* d_sig, int elem_bytes, const int64_t* d_off, const int32_t* d_len, int B, void* d_out,
                           int64_t ld, double* d_stats, hipStream_t st);

// layer 0: x fp32 [B, ldx] -> y [NB * U / 2, cp_out] (fp32 or 16-bit rows, packed block layout), fused bias+ReLU+maxpool
int launch_conv0(const float* d_x, int64_t ldx, const int32_t* d_len, const BlockPlan& plan, int NB,
                 const float* d_w4 /* [cp_out][4] = w0,w1,w2,bias */, int cp_out,
                 void* d_y, int dtype, hipStream_t st, unsigned* d_sat = nullptr);

#ifdef __HIPCC__
// The half-precision modes (RS_F16 / RS_F16X3 / RS_F16XF8) store activations as IEEE half: a value beyond 65504 comes out of the
// conversion as +inf and the forward pass goes on with it.  Every epilogue ORs this into a per-lane word and raises the
// model's sticky flag (rs_model_saturated) when it is non-zero: bit 15 / 31 is set iff a half of `packed` has exponent field
// 31 (activations are >= 0 behind the ReLU, so a half's top bit is free: adding 0x0400 carries into it from 0x7c00 on).
__device__ __forceinline__ unsigned f16_overflow_bits(unsigned packed) { return (packed + 0x04000400u) & 0x80008000u; }
__device__ __forceinline__ void raise_saturated(unsigned* flag, unsigned bits) {
    if (bits && flag) atomicOr(flag, 1u);
}
#endif

struct ConvPlan {
    int kc;           // input channels per K chunk (fp32: multiple of 4; 16-bit: 32 = one MFMA k-step)
    int nch;          // number of chunks: kc * nch >= cp_in
    int n_alloc;      // rows of the packed weight / bias tables (couts + zero rows for any tile width)
};

struct ConvLayerDev {
    int wino_m = 2;           // RS_F32W: output pairs per Winograd tile (2: F(2,3), conv_wino.hip; 4: F(4,3), conv_wino4.hip)
    int c_in, c_out, cp_in, cp_out;
    ConvPlan plan;            // packing of d_w follows plan.kc / plan.nch / plan.n_pad
    void* d_w;                // packed weights [n_alloc][nch][3][kc] (f32 or bf16)
    void* d_w2 = nullptr;     // 16-bit modes: ring packing [panel][tap][n_alloc][64] (conv_ring_h16.hip)
    int ring_panels = 0;      // panels of the ring packing: 64 channels each (plain) or 32 channels as hi | lo (x3)
    bool ring_tail = false;   // split precision: the LAST panel holds <= 8 channels and is packed as the merged tail slab
                              // [hi: tap 0 | tap 1 | tap 2 | 0][lo: ...] in its tap-0 place (conv_ring_h16.hip: TAIL; conv_thin_h16.hip)
    unsigned* d_sat = nullptr; // half-precision modes: the model's sticky "an activation overflowed f16" flag (device word)
    bool f8_in = false;       // RS_F16XF8: the layer reads / writes F8 rows (conv_ring_f8.hip); cp_in / cp_out are their pitches
    bool f8_out = false;
    int x3_terms = 7;         // split precision, -DRS_X3_MASK measurement builds only (RS_X3_TERMS): 1 hi*hi | 2 x lo*w hi | 4 x hi*w lo
    float* d_bias;            // [n_alloc] fp32, zero padded
    // f16 / f16x3: the packed 16-bit weights are the layer's weights x 2^k (max |w| 2^k in [8192, 16384): the LOW halves of
    // small weights are then normal half-precision numbers instead of subnormals, and weights far below 6e-5 do not vanish);
    // the kernels' epilogues multiply the fp32 sums by w_unscale = 2^-k - exactly - before the bias.  1 for every other mode.
    float w_unscale = 1.0f;
    // rs_autotune: the measured-best entry of the kernel's tile-shape table per launch geometry (GEMM rows of the
    // launch -> shape index), consulted before the cost model; force_shape >= 0 overrides both while tuning
    int force_shape = -1;
    std::vector<std::pair<int64_t, int>> tuned;
    const Hooks* hooks = &default_hooks();    // the owning model's switches
};

inline int tuned_shape(const ConvLayerDev& L, int64_t rows) {
    if (L.force_shape >= 0) return L.force_shape;
    for (const auto& t : L.tuned)
        if (t.first == rows) return t.second;
    return -1;
}

int launch_conv_f32(const ConvLayerDev& L, const float* d_x, float* d_y, const int32_t* d_len,
                    int B, int P_in, int layer_index, int num_cu, const float* d_zero, int check_dead,
                    hipStream_t st, int* bm_out, int* bn_out);
int launch_conv_wino(const ConvLayerDev& L, const float* d_x, float* d_y, const int32_t* d_len,
                     int B, int P_in, int layer_index, int num_cu, const float* d_zero, int check_dead,
                     hipStream_t st, int* bm_out, int* bn_out, const float* fuse_xs = nullptr,
                     const float* fuse_w0 = nullptr);
int launch_conv_wino4(const ConvLayerDev& L, const float* d_x, float* d_y, const int32_t* d_len, int B, int P_in,
                      int layer_index, int num_cu, int check_dead, hipStream_t st, int* bm_out, int* bn_out);
int conv_wino4_max_bn();
double conv_wino4_plan_cost(int64_t groups, int n16, int kc, int nch, int num_cu);
double conv_wino4_launch_cost(int64_t groups, int n16, int kc, int nch, int num_cu, bool* thin_out);
bool conv_wino_can_fuse0(const ConvLayerDev& L, int P_in);
// fp32 Winograd, layers 0 + 1 of the shipped net as one LDS-free streaming kernel (conv_stream_f32.hip)
bool conv_stream_f32_ok(const ConvLayerDev& L1, int c0, int P_in1);
int launch_conv_stream_f32(const ConvLayerDev& L1, const float* d_xs, const float* d_w0, int c0, float* d_y,
                           const int32_t* d_len, int B, int P_in1, int num_cu, hipStream_t st);
int conv_wino_max_bn();
// fp32 Winograd for launches of a few rows (small batches): one wave per 16 x 16 tile, no LDS (conv_small_f32.hip);
// bit-identical to conv_wino.hip / conv_wino4.hip
bool conv_small_f32_ok(const ConvLayerDev& L);
int64_t conv_small_f32_waves(const ConvLayerDev& L, int64_t rows_in);        // workgroups (16 x 16 tiles) of such a launch
double conv_small_f32_cost(const ConvLayerDev& L, int64_t rows_in, int num_cu, int* nw_out = nullptr);   // estimate, shader cycles; the channel sub-tiles per wave it assumes
double conv_wino_plan_cost(int64_t rows_out, int n16, int kc, int nch, int num_cu);
double conv_wino_launch_cost(int64_t rows_out, int n16, int kc, int nch, int num_cu, bool* thin_out);
int launch_conv_small_f32(const ConvLayerDev& L, const float* d_x, float* d_y, const int32_t* d_len, int B, int P_in,
                          int layer_index, int num_cu, hipStream_t st, int* bm_out, int* bn_out);
// narrow 16-bit layers (C_in <= 32, C_out <= 48): per-wave streaming kernel, optionally with ConvNet
// layer 0 folded in (fuse_xs = normalised signals at the padded pitch behind 16 zero bytes)
bool conv_stream_h16_ok(const ConvLayerDev& L, int P_in);
bool conv_stream012_h16_ok(const ConvLayerDev& L1, const ConvLayerDev& L2, int c0, int P_in1);
int launch_conv_stream012_h16(const ConvLayerDev& L1, const ConvLayerDev& L2, const float* d_xs, const float* d_w0, int c0,
                              void* d_y, const int32_t* d_len, int B, int P_in1, int num_cu, bool f16, bool x3, hipStream_t st);
int launch_conv_stream_h16(const ConvLayerDev& L, const void* d_x, void* d_y, const int32_t* d_len, int B, int P_in,
                           int layer_index, int num_cu, bool f16, hipStream_t st, const float* fuse_xs,
                           const float* fuse_w0, int fuse_c0, bool x3 = false);
// LDS-DMA ring kernel (conv_ring_h16.hip): plain 16-bit and split-precision (x3) modes
int launch_conv_ring_h16(const ConvLayerDev& L, const void* d_x, void* d_y, const int32_t* d_len, int B, int P_in,
                         int layer_index, int num_cu, bool f16, bool x3, int check_dead, hipStream_t st, int* bm_out,
                         int* bn_out);
// weights-resident kernel for the narrow tiled 16-bit layers (conv_wres_h16.hip): all output channels in one tile, the
// layer's whole weight tensor in LDS; bit-identical to the ring kernel
bool conv_wres_h16_ok(const ConvLayerDev& L, bool x3);
int launch_conv_wres_h16(const ConvLayerDev& L, const void* d_x, void* d_y, const int32_t* d_len, int B, int P_in,
                         int layer_index, int num_cu, bool f16, bool x3, int check_dead, hipStream_t st, int* bm_out,
                         int* bn_out);
// split precision on launches of a few rows: 64 x 32 tiles, a whole panel per barrier (conv_thin_h16.hip); the ring kernel's bits
bool conv_thin_h16_ok(const ConvLayerDev& L);
int64_t conv_thin_h16_tiles(const ConvLayerDev& L, int64_t rows_in);
double conv_thin_h16_cost(const ConvLayerDev& L, int64_t rows_in, int num_cu);
double conv_ring_plan_cost(const ConvLayerDev& L, int64_t rows_in, int num_cu, bool x3);     // the ring kernel's own estimate of the launch
int launch_conv_thin_h16(const ConvLayerDev& L, const void* d_x, void* d_y, const int32_t* d_len, int B, int P_in,
                         int layer_index, int num_cu, bool f16, hipStream_t st, int* bm_out, int* bn_out);
// RS_F16XF8 (conv_ring_f8.hip): split precision with the cross terms on the block-scaled 8-bit MFMA; F8 rows carry a scale plane
int launch_conv_ring_f8(const ConvLayerDev& L, const void* d_x, void* d_y, const int32_t* d_len, int B, int P_in,
                        int layer_index, int num_cu, int check_dead, hipStream_t st, int* bm_out, int* bn_out);
int conv_ring_f8_num_shapes();
bool conv_ring_f8_shape_ok(const ConvLayerDev& L, int k);
size_t f8_scale_offset(int64_t rows, int cp);     // byte offset of the scale plane behind `rows` F8 rows of cp 16-bit elements
int f8_scale_stride(int64_t rows);                // rows per 64-channel panel of the plane (4 bytes each)
size_t f8_scale_bytes(int64_t rows, int cp);
int conv_ring_max_bn();
int conv_ring_num_shapes();
bool conv_ring_shape_ok(const ConvLayerDev& L, int k);
int conv_f32_max_bn();
// tile-shape tables of the tiled kernels (rs_autotune): number of entries, and whether entry k can run layer L
int conv_wino_num_shapes();
bool conv_wino_shape_ok(const ConvLayerDev& L, int k);
int conv_wino4_num_shapes();
bool conv_wino4_shape_ok(const ConvLayerDev& L, int k);
int conv_f32_kc_max();

// rows of read b: (rbase[b] * P_last) + t, t < len[b] >> n_layers (P_last = rows per block of the last buffer)
// fine: the early layers' block table of a two-level layout (a read that did not fit EITHER table is reported as NaN), or null
int launch_head(const void* d_y, int dtype, int cp, int c, int P_last, int n_layers,
                const int32_t* d_len, int B, const BlockPlan& plan, const BlockPlan* fine, const float* d_fcw, const float* d_fcb,
                float* d_probs, float* d_logits, hipStream_t st);
// rows of layer `split - 1`'s output from the fine block layout (Pf rows per block) to the coarse one (Pc rows per block)
int launch_repack_rows(const void* d_src, void* d_dst, const BlockPlan& fine, const BlockPlan& coarse, int NB_coarse, int Pf,
                       int Pc, size_t row_bytes, hipStream_t st);

// the scale plane of F8 rows (conv_ring_f8.hip: [64-channel panel][stride][4 bytes]) across the same re-pack
int launch_repack_scales(const void* d_src, void* d_dst, const BlockPlan& fine, const BlockPlan& coarse, int NB_coarse, int Pf,
                         int Pc, int n_planes, int stride_f, int stride_c, hipStream_t st);

// the `fc` classifier (fc_head.hip): weights on the device, first Linear re-ordered to [P][C4][H]
struct FcHead {
    float* d_w1p = nullptr;
    float* d_b1 = nullptr;
    float* d_w2 = nullptr;      // [2][H]
    float* d_b2 = nullptr;
    int C = 0, C4 = 0, P = 0, H = 0;
};
int fc_head_splits(int B, int H);
size_t fc_head_workspace_bytes(int B, int H);
int launch_fc_pack(const float* d_w1, float* d_w1p, int C, int C4, int P, int H, hipStream_t st);
int launch_fc_head(const float* d_act, int cp, int P_last, int n_layers, const int32_t* d_len, int B, const BlockPlan& plan,
                   const FcHead& fc, float* d_part, float* d_probs, float* d_logits, hipStream_t st);

int launch_decide(const float* d_probs, int n_models, int B, const int32_t* d_len, int max_len,
                  float thr, int mode, uint8_t* d_out, hipStream_t st);

int launch_polya(const int16_t* d_sig, const int64_t* d_off, const int32_t* d_len, int B, int32_t* d_end, hipStream_t st,
                 const int32_t* d_state_in = nullptr, int32_t* d_state_out = nullptr);
int launch_copy_segments(const int16_t* d_src, int16_t* d_dst, const int64_t* d_src_off, const int64_t* d_dst_off,
                         const int32_t* d_len, int n, hipStream_t st);

}  // namespace rs
