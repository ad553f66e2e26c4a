// The `fc` classifier of the reference's ConvNet (riser/nets/cnn.py:22-27):
//     Flatten(1) -> Linear(C * P, H) -> ReLU -> Linear(H, 2)      (softmax: riser/model.py:27)
// on the last conv block's output.  The reference hard-codes C * P = 67 * 753 and H = 4096: one input length (12048 ..
// 12063 samples) of one 4-layer net - an early architecture, no shipped config uses it, and a read of any other length
// fails in its matmul.  Kept for completeness of ConvNet.forward, off the live path: clear, not tuned.
//
// Flatten orders the features channel-major (f = c * P + p); the conv stack keeps activations position-major
// (row = position, C_pad channels per row).  fc_pack_kernel re-orders the first Linear's weights ONCE, at
// rs_model_set_fc_classifier, to W1p[p][c (padded to a multiple of 4, zeros)][o], so that
//     hidden[b][o] = sum_p sum_c act[b][p][c] * W1p[p][c][o]
// is a GEMM with M = reads, N = H, K = P * C4 whose B operand is contiguous in o.  fc1_kernel runs it on the f32-input
// MFMA (v_mfma_f32_16x16x4_f32): a workgroup = 16 reads x 64 outputs (four waves, one 16 x 16 tile each) over a range
// of positions (split-K over positions, so that a batch of a few reads still fills the chip); partial sums go to the
// workspace and fc2_kernel adds them in a fixed order, applies bias + ReLU, the second Linear and the softmax.
// Deterministic: no atomics.
#include "common.hpp"

namespace rs {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void fc_pack_kernel(const float* __restrict__ w1, float* __restrict__ w1p, int C, int C4,
                                                      int P, int H) {
    // one workgroup per (p, c): o runs over the threads; reads stride C * P floats, writes are contiguous (one-time cost)
    const int p = blockIdx.x, c = blockIdx.y;
    float* dst = w1p + ((size_t)p * C4 + c) * H;
    for (int o = threadIdx.x; o < H; o += 256) dst[o] = c < C ? w1[(size_t)o * C * P + (size_t)c * P + p] : 0.0f;
}

struct Fc1Args {
    const float* act;           // last conv layer's output, position-major rows of cp floats
    const float* w1p;           // [P][C4][H]
    float* part;                // [S][Mt * 16][H]
    const int32_t* len;
    const int32_t* rbase;
    int B, cp, C, C4, P, H, P_last, n_layers, S;
};

__global__ __launch_bounds__(256) void fc1_kernel(const Fc1Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int o0 = blockIdx.x * 64 + wave * 16;
    const int b0 = blockIdx.y * 16;
    const int s = blockIdx.z;
    const int p_lo = (int)((int64_t)a.P * s / a.S), p_hi = (int)((int64_t)a.P * (s + 1) / a.S);
    const int r = lane & 15, kk = lane >> 4;
    // A operand: read b0 + r, channel c0 + kk of position p.  Reads beyond the batch, or with another number of rows than
    // the classifier was built for, contribute zeros (fc2 reports NaN for the latter).
    const int b = b0 + r;
    const bool live = b < a.B && (a.len[b < a.B ? b : 0] >> a.n_layers) == a.P;
    const float* arow = a.act + (live ? (int64_t)a.rbase[b] * a.P_last * a.cp : 0);
    const float* wcol = a.w1p + o0 + r;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = p_lo; p < p_hi; ++p) {
        const float* ap = arow + (int64_t)p * a.cp + kk;
        const float* wp = wcol + ((size_t)p * a.C4 + kk) * a.H;
#pragma unroll 4
        for (int c0 = 0; c0 < a.C4; c0 += 4) {
            const float av = (live && c0 + kk < a.C) ? ap[c0] : 0.0f;   // the row's padding channels are not defined
            const float wv = wp[(size_t)c0 * a.H];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wv, acc, 0, 0, 0);
        }
    }
    // D: lane holds rows 4 * (lane >> 4) + i, column lane & 15
    float* out = a.part + ((size_t)s * gridDim.y * 16 + b0) * a.H + o0 + r;
#pragma unroll
    for (int i = 0; i < 4; ++i) out[(size_t)(4 * kk + i) * a.H] = acc[i];
}

__global__ __launch_bounds__(256) void fc2_kernel(const float* __restrict__ part, int S, int rows_pad, int H,
                                                  const float* __restrict__ b1, const float* __restrict__ w2,
                                                  const float* __restrict__ b2, const int32_t* __restrict__ len, int P,
                                                  int n_layers, float* __restrict__ probs, float* __restrict__ logits) {
    __shared__ float red[4][2];
    const int b = blockIdx.x, tid = threadIdx.x;
    if ((len[b] >> n_layers) != P) {                               // the reference's matmul raises for this read
        if (tid < 2) {
            probs[2 * b + tid] = __builtin_nanf("");
            if (logits) logits[2 * b + tid] = __builtin_nanf("");
        }
        return;
    }
    float a0 = 0.f, a1 = 0.f;
    for (int o = tid; o < H; o += 256) {
        float h = b1[o];
        for (int s = 0; s < S; ++s) h += part[((size_t)s * rows_pad + b) * H + o];
        h = fmaxf(h, 0.0f);
        a0 = fmaf(h, w2[o], a0);
        a1 = fmaf(h, w2[H + o], a1);
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
        const float l0 = red[0][0] + red[1][0] + red[2][0] + red[3][0] + b2[0];
        const float l1 = red[0][1] + red[1][1] + red[2][1] + red[3][1] + b2[1];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        const float sum = e0 + e1;
        probs[2 * b + 0] = e0 / sum;
        probs[2 * b + 1] = e1 / sum;
        if (logits) {
            logits[2 * b + 0] = l0;
            logits[2 * b + 1] = l1;
        }
    }
}

}  // namespace

int fc_head_splits(int B, int H) {
    const int mt = (B + 15) / 16, nt = H / 64;
    return std::max(1, std::min(16, 1024 / std::max(1, mt * nt)));
}

size_t fc_head_workspace_bytes(int B, int H) {
    const size_t rows_pad = (size_t)((B + 15) / 16) * 16;
    return (size_t)fc_head_splits(B, H) * rows_pad * H * sizeof(float);
}

int launch_fc_pack(const float* d_w1, float* d_w1p, int C, int C4, int P, int H, hipStream_t st) {
    hipLaunchKernelGGL(fc_pack_kernel, dim3(P, C4), dim3(256), 0, st, d_w1, d_w1p, C, C4, P, H);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

int launch_fc_head(const float* d_act, int cp, int P_last, int n_layers, const int32_t* d_len, int B, const BlockPlan& plan,
                   const FcHead& fc, float* d_part, float* d_probs, float* d_logits, hipStream_t st) {
    if (B <= 0) return RS_OK;
    Fc1Args a;
    a.act = d_act;
    a.w1p = fc.d_w1p;
    a.part = d_part;
    a.len = d_len;
    a.rbase = plan.rbase;
    a.B = B;
    a.cp = cp;
    a.C = fc.C;
    a.C4 = fc.C4;
    a.P = fc.P;
    a.H = fc.H;
    a.P_last = P_last;
    a.n_layers = n_layers;
    a.S = fc_head_splits(B, fc.H);
    const int mt = (B + 15) / 16;
    hipLaunchKernelGGL(fc1_kernel, dim3(fc.H / 64, mt, a.S), dim3(256), 0, st, a);
    RS_HIP(hipGetLastError());
    hipLaunchKernelGGL(fc2_kernel, dim3(B), dim3(256), 0, st, d_part, a.S, mt * 16, fc.H, fc.d_b1, fc.d_w2, fc.d_b2, d_len,
                       fc.P, n_layers, d_probs, d_logits);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

}  // namespace rs
