// K2 (layer-0 direct conv), K4 (GAP + FC + softmax head) and the ensemble decision.
// All three are HBM/latency-bound elementwise-style kernels: no MFMA, coalesced 16-byte
// accesses, one pass over their input.
#include "common.hpp"

namespace rs {
namespace {

// DT: 0 = fp32, 1 = bf16, 2 = f16, 4 = bf16 pairs, 5 = f16 pairs (rs_dtype; 4 / 5 = split precision: a value is
// hi + lo, stored per 32-channel panel as [hi x 32 | lo x 32], see conv_ring_h16.hip)
template <int DT>
constexpr bool kX3 = DT == RS_BF16X3 || DT == RS_F16X3;
template <int DT>
__device__ __forceinline__ unsigned short to16(float f) {
    if constexpr (DT == 2 || DT == RS_F16X3)
        return __builtin_bit_cast(unsigned short, (_Float16)f);
    else
        return __builtin_bit_cast(unsigned short, (__bf16)f);
}
template <int DT>
__device__ __forceinline__ float from16(unsigned short u) {
    if constexpr (DT == 2 || DT == RS_F16X3)
        return (float)__builtin_bit_cast(_Float16, u);
    else
        return __builtin_bit_cast(float, (unsigned)u << 16);
}

// ---- layer 0 ------------------------------------------------------------------------------
// ConvNet layer 0 (riser/nets/cnn.py:52-65 with in_channels = 1): Conv1d(1 -> C, k=3, 'same',
// bias) -> ReLU -> MaxPool1d(2,2).  x is the normalised signal [B, ldx]; y is position-major
// ("NLC") [B * P1, cp] with P1 = P0 / 2: row b*P1 + p holds the C outputs of pooled position
// p, zeros for p >= len[b] / 2 (this is what gives the next layer its 'same' zero padding and
// the per-read halo rows).
// HBM-bound on the store (AI ~ 2.7 flop/B).  A thread owns one 16-byte output piece position
// (q = 4 fp32 / 8 16-bit channels) for the whole block and keeps those channels' (w0,w1,w2,b)
// in registers; per iteration the active threads cover `ppi` consecutive pooled positions, so
// a wave's stores are one contiguous run of 16-byte pieces.  32-bit index arithmetic only.
constexpr int kC0PosPerBlock = 1024;

// Packed block layout: blockIdx.y is a BLOCK k of U = 2 * P1 samples of read b = bread[k]; the block's pooled
// positions are p = j * P1 + (0 .. P1) of the read (j = k - rbase[b]) and land in rows k * P1 + ... of y.
template <int DT>
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ x, int64_t ldx,
                                                    const int32_t* __restrict__ len, const int32_t* __restrict__ rbase,
                                                    const int32_t* __restrict__ bread, const int32_t* __restrict__ blen,
                                                    int P1, int cq, int cp,
                                                    const float4* __restrict__ w4, void* __restrict__ yv, unsigned* sat) {
    constexpr int CH = DT == 0 ? 4 : 8;
    const int ppi = 256 / cq;                                     // positions per iteration
    const int tid = threadIdx.x;
    if (tid >= ppi * cq) return;
    const int slot = tid / cq, q = tid - slot * cq;
    const int kb = blockIdx.y;
    const int b = bread[kb];
    const int n = len[b];
    // pooled positions of THIS block that exist: blen >> 1 (= clamp(n / 2 - p_off, 0, P1) for a table that matches the
    // lengths; 0 for the blocks of a read the plan dropped, whose signal rows were never written)
    const int half_blk = blen[kb] >> 1;
    float4 w[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) w[j] = w4[q * CH + j];           // (w0, w1, w2, bias)
    const float* xr0 = x + (ldx < 0 ? (int64_t)rbase[b] * (2 * P1) : (int64_t)b * ldx);   // ldx < 0: packed signals
    const int p_off = (kb - rbase[b]) * P1;                       // first pooled position of the block in its read
    const int p_begin = blockIdx.x * kC0PosPerBlock;
    const int p_end = min(p_begin + kC0PosPerBlock, P1);
    for (int pl = p_begin + slot; pl < p_end; pl += ppi) {
        const int p = p_off + pl;
        float o[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) o[j] = 0.f;
        if (pl < half_blk) {
            const float* xr = xr0 + 2 * p;
            const float xm = p > 0 ? xr[-1] : 0.0f;
            const float x0 = xr[0], x1 = xr[1];
            const float x2 = (2 * p + 2 < n) ? xr[2] : 0.0f;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const float e = fmaf(w[j].z, x1, fmaf(w[j].y, x0, fmaf(w[j].x, xm, w[j].w)));
                const float f = fmaf(w[j].z, x2, fmaf(w[j].y, x1, fmaf(w[j].x, x0, w[j].w)));
                o[j] = fmaxf(fmaxf(e, f), 0.0f);
            }
        }
        const int64_t piece = ((int64_t)kb * P1 + pl) * cq + q;   // 16-byte piece index
        if constexpr (DT == RS_F16 || DT == RS_F16X3) {            // beyond half precision's range: the model's sticky flag
            float mx = o[0];
#pragma unroll
            for (int j = 1; j < CH; ++j) mx = fmaxf(mx, o[j]);
            if (!(mx <= 65504.0f) && sat) atomicOr(sat, 1u);
        }
        if constexpr (kX3<DT>) {
            // channels 8q .. 8q+7 of panel q >> 2: the hi piece, and the lo piece 64 bytes behind it
            unsigned short hi[8], lo[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                hi[j] = to16<DT>(o[j]);
                lo[j] = to16<DT>(o[j] - from16<DT>(hi[j]));
            }
            const int64_t at = ((int64_t)kb * P1 + pl) * (2 * cq) + (q >> 2) * 8 + (q & 3);
            uint4 v;
            v.x = hi[0] | ((unsigned)hi[1] << 16);
            v.y = hi[2] | ((unsigned)hi[3] << 16);
            v.z = hi[4] | ((unsigned)hi[5] << 16);
            v.w = hi[6] | ((unsigned)hi[7] << 16);
            reinterpret_cast<uint4*>(yv)[at] = v;
            v.x = lo[0] | ((unsigned)lo[1] << 16);
            v.y = lo[2] | ((unsigned)lo[3] << 16);
            v.z = lo[4] | ((unsigned)lo[5] << 16);
            v.w = lo[6] | ((unsigned)lo[7] << 16);
            reinterpret_cast<uint4*>(yv)[at + 4] = v;
        } else if constexpr (DT != 0) {
            uint4 v;
            v.x = to16<DT>(o[0]) | ((unsigned)to16<DT>(o[1]) << 16);
            v.y = to16<DT>(o[2]) | ((unsigned)to16<DT>(o[3]) << 16);
            v.z = to16<DT>(o[4]) | ((unsigned)to16<DT>(o[5]) << 16);
            v.w = to16<DT>(o[6]) | ((unsigned)to16<DT>(o[7]) << 16);
            reinterpret_cast<uint4*>(yv)[piece] = v;
        } else {
            reinterpret_cast<float4*>(yv)[piece] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// ---- head ---------------------------------------------------------------------------------
// AdaptiveAvgPool1d(1) -> Flatten -> Linear(C, 2) (riser/nets/cnn.py:28-33) -> softmax
// (riser/model.py:27).  One 1024-thread workgroup per read: threads stride over channels (coalesced
// 4-byte loads, all rows of a channel in flight together), the mean is over the len >> n_layers
// valid rows of the read's slot in the last activation buffer; wave shuffles + one LDS hop reduce
// the two dot products.  The summation order depends only on the channel count, never on the batch.
constexpr int kHeadThreads = 1024;     // 1702 channels: two per thread, every load of a read in flight at once
constexpr int kHeadRowsUnroll = 4;

template <int DT>
__global__ __launch_bounds__(kHeadThreads) void head_kernel(const void* __restrict__ yv, int cp, int c, int P_last,
                                                           int n_layers, const int32_t* __restrict__ len,
                                                           const int32_t* __restrict__ rbase, int nb_total,
                                                           const int32_t* __restrict__ rbase_f, int nb_total_f,
                                                           const float* __restrict__ fcw, const float* __restrict__ fcb,
                                                           float* __restrict__ probs, float* __restrict__ logits) {
    __shared__ float red[kHeadThreads / 64][2];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int rows = len[b] >> n_layers;
    const float inv = 1.0f / (float)rows;
    const int64_t row0 = (int64_t)rbase[b] * P_last;              // the read's first row (packed block layout)
    // the read was dropped by the plan (see normalise_kernel): its blocks did not fit one of the tables
    if (rbase[b + 1] > nb_total || (rbase_f && rbase_f[b + 1] > nb_total_f)) {
        if (tid < 2) {
            probs[2 * b + tid] = __builtin_nanf("");
            if (logits) logits[2 * b + tid] = __builtin_nanf("");
        }
        return;
    }
    auto ld = [&](int t, int ch) -> float {
        if constexpr (kX3<DT>) {
            const int64_t at = (row0 + t) * cp + ((ch >> 5) << 6) + (ch & 31);
            const unsigned short* y16 = reinterpret_cast<const unsigned short*>(yv);
            return from16<DT>(y16[at]) + from16<DT>(y16[at + 32]);
        }
        const int64_t idx = (row0 + t) * cp + ch;
        if constexpr (DT != 0)
            return from16<DT>(reinterpret_cast<const unsigned short*>(yv)[idx]);
        else
            return reinterpret_cast<const float*>(yv)[idx];
    };
    float a0 = 0.f, a1 = 0.f;
    for (int ch = tid; ch < c; ch += kHeadThreads) {
        const float w0 = fcw[ch], w1 = fcw[c + ch];
        float s = 0.f;
        int t = 0;
        for (; t + kHeadRowsUnroll <= rows; t += kHeadRowsUnroll) {
            float v[kHeadRowsUnroll];
#pragma unroll
            for (int k = 0; k < kHeadRowsUnroll; ++k) v[k] = ld(t + k, ch);
#pragma unroll
            for (int k = 0; k < kHeadRowsUnroll; ++k) s += v[k];
        }
        for (; t < rows; ++t) s += ld(t, ch);
        const float m = s * inv;
        a0 = fmaf(m, w0, a0);
        a1 = fmaf(m, w1, a1);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a0 += __shfl_xor(a0, d, 64);
        a1 += __shfl_xor(a1, d, 64);
    }
    if ((tid & 63) == 0) {
        red[tid >> 6][0] = a0;
        red[tid >> 6][1] = a1;
    }
    __syncthreads();
    if (tid == 0) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int w = 0; w < kHeadThreads / 64; ++w) {
            s0 += red[w][0];
            s1 += red[w][1];
        }
        const float l0 = s0 + fcb[0], l1 = s1 + fcb[1];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        const float s = e0 + e1;
        probs[2 * b + 0] = e0 / s;
        probs[2 * b + 1] = e1 / s;
        if (logits) {
            logits[2 * b + 0] = l0;
            logits[2 * b + 1] = l1;
        }
    }
}

// ---- decision -----------------------------------------------------------------------------
// riser/control.py:75-82.  probs [n_models, B, 2] = (p_off, p_on) per model.
__global__ __launch_bounds__(256) void decide_kernel(const float* __restrict__ probs, int n_models, int B,
                                                     const int32_t* __restrict__ len, int max_len, float thr,
                                                     int mode, uint8_t* __restrict__ out) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    bool any_on = false, all_off = true;
    for (int m = 0; m < n_models; ++m) {
        const float p_off = probs[((int64_t)m * B + b) * 2 + 0];
        const float p_on = probs[((int64_t)m * B + b) * 2 + 1];
        any_on |= p_on > thr;
        all_off &= p_off > thr;
    }
    uint8_t d;
    if (any_on)
        d = mode == RS_ENRICH ? RS_ACCEPT : RS_REJECT;
    else if (all_off)
        d = mode == RS_DEPLETE ? RS_ACCEPT : RS_REJECT;
    else if (len[b] >= max_len)
        d = RS_NO_DECISION;
    else
        d = RS_TRY_AGAIN;
    out[b] = d;
}

}  // namespace

// ---- re-pack between the two levels of the packed layout ---------------------------------------------------
// The early conv layers run on fine blocks (Pf rows of the buffer per block), the late ones on coarse blocks (Pc rows):
// read b's rows are contiguous from row rbase_f[b] * Pf resp. rbase_c[b] * Pc on, valid ones first, zeros behind them.
// One workgroup per COARSE block: its Pc rows of `row_bytes` bytes come from the read's fine rows t = j * Pc + r (j = the
// block's index inside the read) while the read's fine blocks last, zeros beyond.  The buffer is small at this depth (a few
// MB for 512 reads at layer 8): one pass of 16-byte pieces.
namespace {
__global__ __launch_bounds__(256) void repack_rows_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                          const int32_t* __restrict__ rbase_f, const int32_t* __restrict__ rbase_c,
                                                          const int32_t* __restrict__ bread_c, int nb_f_total, int Pf, int Pc,
                                                          int pieces_per_row) {
    const int k = blockIdx.x;                                    // coarse block
    const int b = bread_c[k];
    const int j = k - rbase_c[b];
    const int blocks_f = min(rbase_f[b + 1], nb_f_total) - rbase_f[b];        // the read's fine blocks (<= 0 if it was dropped)
    const int64_t src0 = (int64_t)rbase_f[b] * Pf * pieces_per_row, dst0 = (int64_t)k * Pc * pieces_per_row;
    const int limit = max(blocks_f, 0) * Pf;                     // fine rows the read owns
    for (int f = threadIdx.x; f < Pc * pieces_per_row; f += blockDim.x) {
        const int r = f / pieces_per_row, c = f - r * pieces_per_row;
        const int t = j * Pc + r;
        dst[dst0 + f] = t < limit ? src[src0 + (int64_t)t * pieces_per_row + c] : make_uint4(0u, 0u, 0u, 0u);
    }
}
}  // namespace

int launch_repack_rows(const void* d_src, void* d_dst, const BlockPlan& fine, const BlockPlan& coarse, int NB_coarse, int Pf,
                       int Pc, size_t row_bytes, hipStream_t st) {
    if (NB_coarse <= 0) return RS_OK;
    if (row_bytes % 16 != 0 || Pf < 1 || Pc < 1) {
        set_error("repack_rows: rows of %zu bytes / %d, %d rows per block", row_bytes, Pf, Pc);
        return RS_ERR_ARG;
    }
    hipLaunchKernelGGL(repack_rows_kernel, dim3((unsigned)NB_coarse), dim3(256), 0, st, static_cast<const uint4*>(d_src),
                       static_cast<uint4*>(d_dst), fine.rbase, coarse.rbase, coarse.bread, fine.nb_total, Pf, Pc,
                       (int)(row_bytes / 16));
    RS_HIP(hipGetLastError());
    return RS_OK;
}

// the scale plane of F8 rows across the same re-pack: one dword per row and 64-channel panel; rows behind the read's end get the
// scale of an all-zero block (any valid E8M0 byte would do: their data are zeros)
__global__ __launch_bounds__(256) void repack_scales_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst,
                                                            const int32_t* __restrict__ rbase_f, const int32_t* __restrict__ rbase_c,
                                                            const int32_t* __restrict__ bread_c, int nb_f_total, int Pf, int Pc,
                                                            int stride_f, int stride_c) {
    const int k = blockIdx.x, p = blockIdx.y;
    const int b = bread_c[k];
    const int j = k - rbase_c[b];
    const int blocks_f = min(rbase_f[b + 1], nb_f_total) - rbase_f[b];
    const int64_t src0 = (int64_t)p * stride_f + (int64_t)rbase_f[b] * Pf, dst0 = (int64_t)p * stride_c + (int64_t)k * Pc;
    const int limit = max(blocks_f, 0) * Pf;
    for (int r = threadIdx.x; r < Pc; r += blockDim.x) {
        const int t = j * Pc + r;
        dst[dst0 + r] = t < limit ? src[src0 + t] : 0x5f6a5f6au;
    }
}

int launch_repack_scales(const void* d_src, void* d_dst, const BlockPlan& fine, const BlockPlan& coarse, int NB_coarse, int Pf,
                         int Pc, int n_planes, int stride_f, int stride_c, hipStream_t st) {
    if (NB_coarse <= 0 || n_planes <= 0) return RS_OK;
    hipLaunchKernelGGL(repack_scales_kernel, dim3((unsigned)NB_coarse, (unsigned)n_planes), dim3(64), 0, st,
                       static_cast<const unsigned*>(d_src), static_cast<unsigned*>(d_dst), fine.rbase, coarse.rbase, coarse.bread,
                       fine.nb_total, Pf, Pc, stride_f, stride_c);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

int launch_conv0(const float* d_x, int64_t ldx, const int32_t* d_len, const BlockPlan& plan, int NB, const float* d_w4,
                 int cp_out, void* d_y, int dtype, hipStream_t st, unsigned* d_sat) {
    const int P1 = (1 << plan.shift) / 2;
    // 16-byte pieces of 4 (fp32) / 8 (16-bit) channels per output row; split precision: per row cp_out / 2 logical
    // channel slots, each 8-channel group stored as a hi piece and a lo piece
    const int cq = is_x3(dtype) ? cp_out / 16 : cp_out / (dtype == RS_F32 ? 4 : 8);
    if (cq < 1 || cq > 256) {
        set_error("conv0: unsupported first-layer width %d", cp_out);
        return RS_ERR_ARG;
    }
    if (NB > 65535) {
        set_error("conv0: %d blocks exceed the launch grid, split the batch", NB);
        return RS_ERR_ARG;
    }
    dim3 grid((P1 + kC0PosPerBlock - 1) / kC0PosPerBlock, NB);
    auto fn = dtype == RS_F16 ? conv0_kernel<2> : dtype == RS_BF16 ? conv0_kernel<1>
            : dtype == RS_BF16X3 ? conv0_kernel<RS_BF16X3> : dtype == RS_F16X3 ? conv0_kernel<RS_F16X3> : conv0_kernel<0>;
    hipLaunchKernelGGL(fn, grid, dim3(256), 0, st, d_x, ldx, d_len, plan.rbase, plan.bread, plan.blen, P1, cq, cp_out,
                       reinterpret_cast<const float4*>(d_w4), d_y, d_sat);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

int launch_head(const void* d_y, int dtype, int cp, int c, int P_last, int n_layers, const int32_t* d_len,
                int B, const BlockPlan& plan, const BlockPlan* fine, const float* d_fcw, const float* d_fcb, float* d_probs,
                float* d_logits, hipStream_t st) {
    auto fn = dtype == RS_F16 ? head_kernel<2> : dtype == RS_BF16 ? head_kernel<1>
            : dtype == RS_BF16X3 ? head_kernel<RS_BF16X3> : dtype == RS_F16X3 ? head_kernel<RS_F16X3> : head_kernel<0>;
    hipLaunchKernelGGL(fn, dim3(B), dim3(kHeadThreads), 0, st, d_y, cp, c, P_last, n_layers, d_len, plan.rbase,
                       plan.nb_total, fine ? fine->rbase : nullptr, fine ? fine->nb_total : 0, d_fcw, d_fcb, d_probs, d_logits);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

int launch_decide(const float* d_probs, int n_models, int B, const int32_t* d_len, int max_len, float thr,
                  int mode, uint8_t* d_out, hipStream_t st) {
    if (B <= 0) return RS_OK;
    hipLaunchKernelGGL(decide_kernel, dim3((B + 255) / 256), dim3(256), 0, st, d_probs, n_models, B, d_len,
                       max_len, thr, mode, d_out);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

}  // namespace rs
