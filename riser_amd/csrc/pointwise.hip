// K2 (layer-0 direct conv), K4 (GAP + FC + softmax head) and the ensemble decision.
// All three are HBM/latency-bound elementwise-style kernels: no MFMA, coalesced 16-byte
// accesses, one pass over their input.
#include "common.hpp"

#include <hip/hip_bf16.h>

namespace rs {
namespace {

__device__ __forceinline__ unsigned short f2bf(float f) {
    return __builtin_bit_cast(unsigned short, __float2bfloat16(f));
}
__device__ __forceinline__ float bf2f(unsigned short u) {
    return __builtin_bit_cast(float, (unsigned)u << 16);
}

// ---- layer 0 ------------------------------------------------------------------------------
// ConvNet layer 0 (riser/nets/cnn.py:52-65 with in_channels = 1): Conv1d(1 -> C, k=3, 'same',
// bias) -> ReLU -> MaxPool1d(2,2).  x is the normalised signal [B, ldx] (zeros beyond each
// read's length up to P0); y is position-major ("NLC") [B * P1, cp] with P1 = P0 / 2:
// row b*P1 + p holds the C outputs of pooled position p, zeros for p >= len[b] / 2 (this is
// what gives the next layer its 'same' zero padding and the per-read halo rows).
// One thread = one pooled position x 4 channels -> one 16-byte (fp32) / 8-byte (bf16) store;
// consecutive threads write consecutive addresses.  AI ~ 2.7 flop/B: HBM-bound on the store.
template <bool BF16>
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ x, int64_t ldx,
                                                    const int32_t* __restrict__ len, int P1, int cq /* cp/4 */,
                                                    const float4* __restrict__ w4, void* __restrict__ yv,
                                                    int64_t total) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int q = (int)(g % cq);
    const int64_t prow = g / cq;
    const int b = (int)(prow / P1);
    const int p = (int)(prow - (int64_t)b * P1);
    const int n = len[b];
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (p < (n >> 1)) {
        const float* xr = x + (int64_t)b * ldx + 2 * p;
        const float xm = p > 0 ? xr[-1] : 0.0f;
        const float x0 = xr[0], x1 = xr[1];
        const float x2 = (2 * p + 2 < n) ? xr[2] : 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 w = w4[q * 4 + j];                       // (w0, w1, w2, bias)
            const float e = fmaf(w.z, x1, fmaf(w.y, x0, fmaf(w.x, xm, w.w)));
            const float f = fmaf(w.z, x2, fmaf(w.y, x1, fmaf(w.x, x0, w.w)));
            o[j] = fmaxf(fmaxf(e, f), 0.0f);
        }
    }
    if (BF16) {
        ushort4 v;
        v.x = f2bf(o[0]); v.y = f2bf(o[1]); v.z = f2bf(o[2]); v.w = f2bf(o[3]);
        reinterpret_cast<ushort4*>(yv)[g] = v;
    } else {
        reinterpret_cast<float4*>(yv)[g] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---- head ---------------------------------------------------------------------------------
// AdaptiveAvgPool1d(1) -> Flatten -> Linear(C, 2) (riser/nets/cnn.py:28-33) -> softmax
// (riser/model.py:27).  One wave per read: lanes stride over channels, the mean is over the
// len >> n_layers valid rows of the read's slot in the last activation buffer.
template <bool BF16>
__global__ __launch_bounds__(64) void head_kernel(const void* __restrict__ yv, int cp, int c, int P_last,
                                                  int n_layers, const int32_t* __restrict__ len,
                                                  const float* __restrict__ fcw, const float* __restrict__ fcb,
                                                  float* __restrict__ probs, float* __restrict__ logits) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int rows = len[b] >> n_layers;
    const float inv = 1.0f / (float)rows;
    float a0 = 0.f, a1 = 0.f;
    for (int ch = lane; ch < c; ch += 64) {
        float s = 0.f;
        for (int t = 0; t < rows; ++t) {
            const int64_t idx = ((int64_t)b * P_last + t) * cp + ch;
            s += BF16 ? bf2f(reinterpret_cast<const unsigned short*>(yv)[idx]) : reinterpret_cast<const float*>(yv)[idx];
        }
        const float m = s * inv;
        a0 = fmaf(m, fcw[ch], a0);
        a1 = fmaf(m, fcw[c + ch], a1);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a0 += __shfl_xor(a0, d, 64);
        a1 += __shfl_xor(a1, d, 64);
    }
    if (lane == 0) {
        const float l0 = a0 + fcb[0], l1 = a1 + fcb[1];
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

int launch_conv0(const float* d_x, int64_t ldx, const int32_t* d_len, int B, int P0, const float* d_w4,
                 int cp_out, void* d_y, bool bf16_out, hipStream_t st) {
    const int P1 = P0 / 2, cq = cp_out / 4;
    const int64_t total = (int64_t)B * P1 * cq;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (bf16_out)
        hipLaunchKernelGGL(conv0_kernel<true>, dim3(grid), dim3(256), 0, st, d_x, ldx, d_len, P1, cq,
                           reinterpret_cast<const float4*>(d_w4), d_y, total);
    else
        hipLaunchKernelGGL(conv0_kernel<false>, dim3(grid), dim3(256), 0, st, d_x, ldx, d_len, P1, cq,
                           reinterpret_cast<const float4*>(d_w4), d_y, total);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

int launch_head(const void* d_y, bool bf16_in, int cp, int c, int P_last, int n_layers, const int32_t* d_len,
                int B, const float* d_fcw, const float* d_fcb, float* d_probs, float* d_logits, hipStream_t st) {
    if (bf16_in)
        hipLaunchKernelGGL(head_kernel<true>, dim3(B), dim3(64), 0, st, d_y, cp, c, P_last, n_layers, d_len, d_fcw,
                           d_fcb, d_probs, d_logits);
    else
        hipLaunchKernelGGL(head_kernel<false>, dim3(B), dim3(64), 0, st, d_y, cp, c, P_last, n_layers, d_len, d_fcw,
                           d_fcb, d_probs, d_logits);
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
