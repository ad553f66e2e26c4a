// Sequential conv programs: the ResNet variant of the reference (riser/nets/resnet.py:7-131),
// a secondary architecture that riser/model.py cannot even load (it hard-imports ConvNet) and
// for which no config or weights are shipped.  Built for PARITY, not speed: BatchNorm (eval mode)
// is folded into the preceding conv on the host, so the device executes a list of
//   CONV    y = [relu]( conv1d(x; k, stride, pad) + bias [+ residual] )     direct fp32 FMA
//   MAXPOOL y = MaxPool1d(2, stride 2, padding 1)                             (stem, resnet.py:83)
// over position-major activations [B][T][C] with one uniform length per batch, then GAP -> FC ->
// softmax.  The hot path of this repository is the ConvNet in conv_f32.hip / conv_h16.hip.
#include "common.hpp"

#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

namespace rs {
namespace {

// one thread = one (b, t_out) position x 4 output channels; weights packed [k][c_in][c_out4*4]
__global__ __launch_bounds__(256) void seq_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const float* __restrict__ add,
                                                       float* __restrict__ y, int B, int T_in, int T_out, int c_in,
                                                       int c_out, int cq, int k, int stride, int pad, int relu) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)B * T_out * cq;
    if (g >= total) return;
    const int q = (int)(g % cq);
    const int64_t bt = g / cq;
    const int t = (int)(bt % T_out);
    const int b = (int)(bt / T_out);
    float4 acc = *reinterpret_cast<const float4*>(bias + 4 * q);
    for (int kk = 0; kk < k; ++kk) {
        const int ti = t * stride - pad + kk;
        if (ti < 0 || ti >= T_in) continue;
        const float* xr = x + ((int64_t)b * T_in + ti) * c_in;
        const float* wr = w + ((int64_t)kk * c_in) * (cq * 4) + 4 * q;
        for (int ci = 0; ci < c_in; ++ci) {
            const float xv = xr[ci];
            const float4 wv = *reinterpret_cast<const float4*>(wr + (int64_t)ci * (cq * 4));
            acc.x = fmaf(xv, wv.x, acc.x);
            acc.y = fmaf(xv, wv.y, acc.y);
            acc.z = fmaf(xv, wv.z, acc.z);
            acc.w = fmaf(xv, wv.w, acc.w);
        }
    }
    float o[4] = {acc.x, acc.y, acc.z, acc.w};
    const int64_t obase = ((int64_t)b * T_out + t) * c_out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = 4 * q + j;
        if (co < c_out) {
            float v = o[j];
            if (add) v += add[obase + co];
            if (relu) v = fmaxf(v, 0.0f);
            y[obase + co] = v;
        }
    }
}

__global__ __launch_bounds__(256) void seq_maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                          int T_in, int T_out, int c) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (int64_t)B * T_out * c) return;
    const int ch = (int)(g % c);
    const int64_t bt = g / c;
    const int t = (int)(bt % T_out);
    const int b = (int)(bt / T_out);
    const int t0 = 2 * t - 1, t1 = 2 * t;                         // window of MaxPool1d(2, 2, padding 1)
    float v = -INFINITY;
    if (t0 >= 0 && t0 < T_in) v = fmaxf(v, x[((int64_t)b * T_in + t0) * c + ch]);
    if (t1 >= 0 && t1 < T_in) v = fmaxf(v, x[((int64_t)b * T_in + t1) * c + ch]);
    y[g] = v;
}

// GAP over T rows -> FC(c, 2) -> softmax; one wave per read
__global__ __launch_bounds__(64) void seq_head_kernel(const float* __restrict__ x, int T, int c,
                                                      const float* __restrict__ fcw, const float* __restrict__ fcb,
                                                      float* __restrict__ probs, float* __restrict__ logits) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float a0 = 0.f, a1 = 0.f;
    for (int ch = lane; ch < c; ch += 64) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += x[((int64_t)b * T + t) * c + ch];
        const float m = s / (float)T;
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
        probs[2 * b] = e0 / (e0 + e1);
        probs[2 * b + 1] = e1 / (e0 + e1);
        if (logits) {
            logits[2 * b] = l0;
            logits[2 * b + 1] = l1;
        }
    }
}

struct OpDev {
    int kind, src, dst, add;
    int c_in, c_out, k, stride, pad, relu;
    float* d_w = nullptr;     // [k][c_in][cq*4]
    float* d_b = nullptr;     // [cq*4]
};

}  // namespace
}  // namespace rs

using namespace rs;

struct rs_seqnet {
    int device = 0;
    int n_buffers = 0;
    int c_last = 0;
    std::vector<OpDev> ops;
    float* d_fcw = nullptr;
    float* d_fcb = nullptr;
};

namespace {

// lengths and channel widths of every buffer for an input of L samples (buffer 0 = the input)
// (buffer ids are reused by later ops with other shapes: *max_elems is the largest T*C any buffer
// ever holds, which sizes the workspace regions)
struct OpShape {
    int t_in, t_out, c;
};

bool propagate(const rs_seqnet* m, int L, std::vector<int>& T, std::vector<int>& C, size_t* max_elems = nullptr,
               std::vector<OpShape>* shapes = nullptr) {
    if (max_elems) *max_elems = 0;
    if (shapes) shapes->clear();
    T.assign(m->n_buffers, -1);
    C.assign(m->n_buffers, 0);
    T[0] = L;
    C[0] = 1;
    for (const OpDev& o : m->ops) {
        if (T[o.src] < 0) return false;
        int t_out;
        if (o.kind == 0) {
            if (T[o.src] + 2 * o.pad < o.k) return false;          // torch: kernel larger than padded input
            t_out = (T[o.src] + 2 * o.pad - o.k) / o.stride + 1;
            C[o.dst] = o.c_out;
        } else {
            t_out = T[o.src] / 2 + 1;
            C[o.dst] = C[o.src];
        }
        if (t_out < 1) return false;
        if (o.add >= 0 && (T[o.add] != t_out || C[o.add] != C[o.dst])) return false;
        if (shapes) shapes->push_back({T[o.src], t_out, C[o.dst]});
        T[o.dst] = t_out;
        if (max_elems) *max_elems = std::max(*max_elems, (size_t)t_out * C[o.dst]);
    }
    return true;
}

size_t buffer_bytes(const rs_seqnet* m, int B, int L) {
    std::vector<int> T, C;
    size_t elems = 0;
    if (!propagate(m, L, T, C, &elems)) return 0;
    return ((size_t)B * elems * sizeof(float) + 255) / 256 * 256;
}

}  // namespace

extern "C" {

int rs_seqnet_create(const rs_seq_op* ops, int n_ops, int n_buffers, const float* fc_w, const float* fc_b, int c_last,
                     int device, rs_seqnet** out) {
    if (!ops || n_ops < 1 || n_buffers < 2 || n_buffers > 16 || !fc_w || !fc_b || !out || c_last < 1) {
        set_error("rs_seqnet_create: bad argument");
        return RS_ERR_ARG;
    }
    *out = nullptr;
    DeviceGuard guard(device);            // the caller's current device is restored on return
    RS_HIP(guard.err);
    rs_seqnet* m = new (std::nothrow) rs_seqnet();
    if (!m) return RS_ERR_OOM;
    m->device = device;
    m->n_buffers = n_buffers;
    m->c_last = c_last;
    for (int i = 0; i < n_ops; ++i) {
        const rs_seq_op& s = ops[i];
        OpDev o;
        o.kind = s.kind; o.src = s.src; o.dst = s.dst; o.add = s.add;
        o.c_in = s.c_in; o.c_out = s.c_out; o.k = s.k; o.stride = s.stride; o.pad = s.pad; o.relu = s.relu;
        const bool bad_buf = s.src < 0 || s.src >= n_buffers || s.dst < 1 || s.dst >= n_buffers || s.dst == s.src ||
                             s.add >= n_buffers || s.add == s.dst;
        if (bad_buf || (s.kind != 0 && s.kind != 1) ||
            (s.kind == 0 && (!s.w || !s.b || s.c_in < 1 || s.c_out < 1 || s.k < 1 || s.stride < 1 || s.pad < 0))) {
            rs_seqnet_destroy(m);
            set_error("rs_seqnet_create: bad op %d", i);
            return RS_ERR_ARG;
        }
        if (s.kind == 0) {
            const int cq = (s.c_out + 3) / 4;
            std::vector<float> wp((size_t)s.k * s.c_in * cq * 4, 0.0f), bp((size_t)cq * 4, 0.0f);
            for (int co = 0; co < s.c_out; ++co) {
                bp[co] = s.b[co];
                for (int ci = 0; ci < s.c_in; ++ci)
                    for (int kk = 0; kk < s.k; ++kk)
                        wp[((size_t)kk * s.c_in + ci) * cq * 4 + co] = s.w[((size_t)co * s.c_in + ci) * s.k + kk];
            }
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&o.d_w), wp.size() * 4);
            if (e == hipSuccess) e = hipMemcpy(o.d_w, wp.data(), wp.size() * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&o.d_b), bp.size() * 4);
            if (e == hipSuccess) e = hipMemcpy(o.d_b, bp.data(), bp.size() * 4, hipMemcpyHostToDevice);
            m->ops.push_back(o);
            if (e != hipSuccess) {
                rs_seqnet_destroy(m);
                return hip_fail(e, "rs_seqnet_create upload");
            }
        } else {
            m->ops.push_back(o);
        }
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->d_fcw), (size_t)2 * c_last * 4);
    if (e == hipSuccess) e = hipMemcpy(m->d_fcw, fc_w, (size_t)2 * c_last * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&m->d_fcb), 8);
    if (e == hipSuccess) e = hipMemcpy(m->d_fcb, fc_b, 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        rs_seqnet_destroy(m);
        return hip_fail(e, "rs_seqnet_create upload");
    }
    *out = m;
    return RS_OK;
}

int rs_seqnet_destroy(rs_seqnet* m) {
    if (!m) return RS_OK;
    DeviceGuard guard(m->device);
    for (OpDev& o : m->ops) {
        if (o.d_w) (void)hipFree(o.d_w);
        if (o.d_b) (void)hipFree(o.d_b);
    }
    if (m->d_fcw) (void)hipFree(m->d_fcw);
    if (m->d_fcb) (void)hipFree(m->d_fcb);
    delete m;
    return RS_OK;
}

size_t rs_seqnet_workspace_bytes(const rs_seqnet* m, int B, int L) {
    if (!m || B < 1 || L < 1) return 0;
    return buffer_bytes(m, B, L) * (size_t)(m->n_buffers - 1);
}

int rs_seqnet_forward(rs_seqnet* m, const float* d_x, int B, int L, void* d_ws, size_t ws_bytes, float* d_probs,
                      float* d_logits, void* stream) {
    if (!m || !d_x || !d_ws || !d_probs || B < 1 || L < 1) {
        set_error("rs_seqnet_forward: bad argument");
        return RS_ERR_ARG;
    }
    std::vector<int> T, C;
    std::vector<OpShape> shp;                                      // buffer ids are reused: shapes are per op
    if (!propagate(m, L, T, C, nullptr, &shp)) {
        set_error("rs_seqnet_forward: input of %d samples is too short for this network", L);
        return RS_ERR_LENGTH;
    }
    const size_t per = buffer_bytes(m, B, L);
    if (ws_bytes < per * (size_t)(m->n_buffers - 1)) {
        set_error("rs_seqnet_forward: workspace too small");
        return RS_ERR_WORKSPACE;
    }
    DeviceGuard guard(m->device);
    RS_HIP(guard.err);
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto buf = [&](int i) -> float* {
        return i == 0 ? const_cast<float*>(d_x) : reinterpret_cast<float*>(static_cast<char*>(d_ws) + per * (size_t)(i - 1));
    };
    int last = 0;
    for (size_t k = 0; k < m->ops.size(); ++k) {
        const OpDev& o = m->ops[k];
        const OpShape& sh = shp[k];
        if (o.kind == 0) {
            const int cq = (o.c_out + 3) / 4;
            const int64_t total = (int64_t)B * sh.t_out * cq;
            hipLaunchKernelGGL(seq_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, buf(o.src),
                               o.d_w, o.d_b, o.add >= 0 ? buf(o.add) : nullptr, buf(o.dst), B, sh.t_in, sh.t_out,
                               o.c_in, o.c_out, cq, o.k, o.stride, o.pad, o.relu);
        } else {
            const int64_t total = (int64_t)B * sh.t_out * sh.c;
            hipLaunchKernelGGL(seq_maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                               buf(o.src), buf(o.dst), B, sh.t_in, sh.t_out, sh.c);
        }
        RS_HIP(hipGetLastError());
        last = o.dst;
    }
    if (C[last] != m->c_last) {
        set_error("rs_seqnet_forward: last buffer has %d channels, classifier expects %d", C[last], m->c_last);
        return RS_ERR_ARG;
    }
    hipLaunchKernelGGL(seq_head_kernel, dim3(B), dim3(64), 0, st, buf(last), T[last], m->c_last, m->d_fcw, m->d_fcb,
                       d_probs, d_logits);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

}  // extern "C"
