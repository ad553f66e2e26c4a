// Sequential conv programs: the ResNet variant of the reference (riser/nets/resnet.py:7-131) - a secondary
// architecture that riser/model.py cannot even load (it hard-imports ConvNet) and for which no config or weights are
// shipped - and ConvNet configurations outside the shipped class (depth > 1, kernels other than 3:
// riser/nets/cnn.py:17,52-65).  BatchNorm (eval mode) is folded into the preceding conv on the host, so the device
// executes a list of
//   CONV    y = [relu]( conv1d(x; k, stride, pad) + bias [+ residual] )
//   MAXPOOL y = MaxPool1d(2, stride 2, padding 0 | 1)          (cnn.py:64 / the ResNet stem, resnet.py:83)
// over position-major activations [B][T][C] (exactly C channels per row) with one uniform length per batch, then
// GAP -> FC -> softmax.
// CONV runs on the f32-input MFMA (seq_conv_mfma_kernel): in the position-major layout the im2col row of output
// position t IS a contiguous run of memory - x[b][t*s - pad .. t*s - pad + k) x [0, c_in) - so the GEMM
// (M = B * T_out, N = c_out, K = k * c_in) needs no gather: lane (row r, k-group kq) of a 16x16x4 MFMA reads
// xflat[row_base(r) + 4 * step + kq], zero outside the batch element's own [0, T_in * c_in).  The scalar kernel the
// first round shipped (one thread per position x 4 channels) is kept behind RS_SEQ_SCALAR=1 for the comparison.
// Generic over k / stride / pad / widths, not tuned per shape: the hot path of this repository is the ConvNet in
// conv_wino*.hip / conv_ring_h16.hip.
#include "common.hpp"

#include <stdlib.h>
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

typedef float f32x4 __attribute__((ext_vector_type(4)));

// CONV on the f32-input MFMA: a 256-thread workgroup = 4 waves x (16 output rows x 16 * NT output channels)
template <int NT>
__global__ __launch_bounds__(256) void seq_conv_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ add,
                                                            float* __restrict__ y, int B, int T_in, int T_out, int c_in,
                                                            int c_out, int wpitch, int K, int stride, int pad, int relu) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int64_t rows = (int64_t)B * T_out;
    const int64_t g = (int64_t)blockIdx.x * 64 + wave * 16 + r;          // the output row whose im2col row this lane feeds
    const bool row_ok = g < rows;
    const int b = row_ok ? (int)(g / T_out) : 0;
    const int t = row_ok ? (int)(g - (int64_t)b * T_out) : 0;
    const int off0 = (t * stride - pad) * c_in;                            // first element of the im2col row inside the element
    const int lim = T_in * c_in;
    const float* xb = x + (int64_t)b * lim;
    const int n0 = blockIdx.y * (16 * NT);
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int steps = (K + 3) / 4;
    constexpr int U = 4;                                                   // k-steps in flight
    for (int s0 = 0; s0 < steps; s0 += U) {
        float av[U], bv[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kidx = 4 * (s0 + u) + kq;
            const int o = off0 + kidx;
            av[u] = (row_ok && kidx < K && o >= 0 && o < lim) ? xb[o] : 0.0f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = n0 + 16 * j + r;
                bv[u][j] = (kidx < K && col < wpitch) ? w[(int64_t)kidx * wpitch + col] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u][j], acc[j], 0, 0, 0);
    }
    // accumulator element e of lane (col = lane & 15, row group = lane >> 4) is output row 4 * (lane >> 4) + e
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = n0 + 16 * j + r;
        if (col >= c_out) continue;
        const float bcol = bias[col];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t row = (int64_t)blockIdx.x * 64 + wave * 16 + 4 * kq + e;
            if (row < rows) {
                float v = acc[j][e] + bcol;
                if (add) v += add[row * c_out + col];
                if (relu) v = fmaxf(v, 0.0f);
                y[row * c_out + col] = v;
            }
        }
    }
}

// CONV on the f32-input MFMA, weights resident in LDS: the layers of these nets are narrow (K * N * 4 bytes fits LDS
// many times over), so a persistent workgroup loads the whole packed weight matrix once - [K / 4][Npad][4], so that
// lane (column, k-group) reads the B operands of FOUR k-steps with one ds_read_b128 - and walks 128-row tiles of the
// GEMM: a wave owns 32 output rows x all Npad columns and per 16 K elements issues 2 (16-byte) loads of its im2col rows,
// NT ds_read_b128 and 8 * NT MFMAs.  The k index is permuted (lane kq of step 4u + i holds element 16u + 4kq + i) so
// that a lane's four A values of a 16-element chunk are one contiguous 16-byte load of the position-major input.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void seq_conv_mfma_lds_kernel(const float* __restrict__ x, unsigned x_bytes,
                                                                const float* __restrict__ wq /* [K16/4][16 NT][4] */,
                                                                const float* __restrict__ bias, const float* __restrict__ add,
                                                                float* __restrict__ y, int B, int T_in, int T_out, int c_in,
                                                                int c_out, int K, int stride, int pad, int relu, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    constexpr int NP = 16 * NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int K16 = (K + 15) & ~15;
    for (int i = threadIdx.x; i < K16 / 4 * NP; i += 256)
        reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(wq)[i];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    const int64_t rows = (int64_t)B * T_out;
    const int lim = T_in * c_in;
    float bcol[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bcol[j] = 16 * j + r < c_out ? bias[16 * j + r] : 0.0f;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t row0 = (int64_t)tile * 128 + wave * 32;
        int off0[2];
        int64_t base[2];
        bool ok[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int64_t g = row0 + 16 * m + r;
            ok[m] = g < rows;
            const int b = ok[m] ? (int)(g / T_out) : 0;
            const int t = ok[m] ? (int)(g - (int64_t)b * T_out) : 0;
            off0[m] = (t * stride - pad) * c_in;
            base[m] = (int64_t)b * lim;
        }
        f32x4 acc[2][NT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // this lane's four im2col elements of rows m = 0, 1 for the chunk at k0 (the next chunk is loaded ahead of the
        // current chunk's MFMAs: the loop is otherwise bound by the round trip of these loads)
        auto load_a = [&](int k0, f32x4 (&av)[2]) {
            const int kidx = k0 + 4 * kq;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int o = off0[m] + kidx;
                if (ok[m] && o >= 0 && o + 3 < lim && kidx + 3 < K) {
                    av[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)((base[m] + o) * 4), 0, 0));
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        av[m][i] = (ok[m] && kidx + i < K && o + i >= 0 && o + i < lim) ? x[base[m] + o + i] : 0.0f;
                }
            }
        };
        f32x4 av[2], avn[2];
        load_a(0, av);
        for (int k0 = 0; k0 < K16; k0 += 16) {
            f32x4 bv[NT];
            if (k0 + 16 < K16) load_a(k0 + 16, avn);
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = *reinterpret_cast<const f32x4*>(wl + ((k0 / 4 + kq) * NP + 16 * j + r) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][i], bv[j][i], acc[m][j], 0, 0, 0);
            av[0] = avn[0];
            av[1] = avn[1];
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = 16 * j + r;
                if (col >= c_out) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int64_t row = row0 + 16 * m + 4 * kq + e;
                    if (row < rows) {
                        float v = acc[m][j][e] + bcol[j];
                        if (add) v += add[row * c_out + col];
                        if (relu) v = fmaxf(v, 0.0f);
                        y[row * c_out + col] = v;
                    }
                }
            }
    }
}

__global__ __launch_bounds__(256) void seq_maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                          int T_in, int T_out, int c, int pad) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (int64_t)B * T_out * c) return;
    const int ch = (int)(g % c);
    const int64_t bt = g / c;
    const int t = (int)(bt % T_out);
    const int b = (int)(bt / T_out);
    const int t0 = 2 * t - pad, t1 = 2 * t + 1 - pad;             // window of MaxPool1d(2, 2, padding pad)
    float v = -INFINITY;
    if (t0 >= 0 && t0 < T_in) v = fmaxf(v, x[((int64_t)b * T_in + t0) * c + ch]);
    if (t1 >= 0 && t1 < T_in) v = fmaxf(v, x[((int64_t)b * T_in + t1) * c + ch]);
    y[g] = v;
}

// GAP over T rows -> FC(c, 2) -> softmax; one 256-thread workgroup per read: wave w sums the rows t = w (mod 4) of
// each channel (coalesced 256-byte row segments, four rows in flight per channel group), LDS combines the four partial
// sums in a fixed order, wave 0 finishes
__global__ __launch_bounds__(256) void seq_head_kernel(const float* __restrict__ x, int T, int c,
                                                       const float* __restrict__ fcw, const float* __restrict__ fcb,
                                                       float* __restrict__ probs, float* __restrict__ logits) {
    __shared__ float part[4][64];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a0 = 0.f, a1 = 0.f;
    for (int c0 = 0; c0 < c; c0 += 64) {
        const int ch = c0 + lane;
        float s = 0.f;
        if (ch < c)
            for (int t = wave; t < T; t += 4) s += x[((int64_t)b * T + t) * c + ch];
        part[wave][lane] = s;
        __syncthreads();
        if (wave == 0 && ch < c) {
            const float m = (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) / (float)T;
            a0 = fmaf(m, fcw[ch], a0);
            a1 = fmaf(m, fcw[c + ch], a1);
        }
        __syncthreads();
    }
    if (wave != 0) return;
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
    float* d_wq = nullptr;    // MFMA packing [K16 / 4][Npad][4] (K = k * c_in in im2col order), or null if too large for LDS
    int nt = 0;               // Npad / 16
};

}  // namespace
}  // namespace rs

using namespace rs;

struct rs_seqnet {
    int device = 0;
    int n_buffers = 0;
    int c_last = 0;
    bool scalar_conv = false;      // RS_SEQ_SCALAR=1 (read at create): the scalar FMA conv kernel instead of the MFMA one
    int num_cu = 256;
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
            t_out = o.pad ? T[o.src] / 2 + 1 : T[o.src] / 2;      // MaxPool1d(2, 2, padding 1 | 0)
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
    m->scalar_conv = getenv("RS_SEQ_SCALAR") != nullptr;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) m->num_cu = cus;
    }
    for (int i = 0; i < n_ops; ++i) {
        const rs_seq_op& s = ops[i];
        OpDev o;
        o.kind = s.kind; o.src = s.src; o.dst = s.dst; o.add = s.add;
        o.c_in = s.c_in; o.c_out = s.c_out; o.k = s.k; o.stride = s.stride; o.pad = s.pad; o.relu = s.relu;
        const bool bad_buf = s.src < 0 || s.src >= n_buffers || s.dst < 1 || s.dst >= n_buffers || s.dst == s.src ||
                             s.add >= n_buffers || s.add == s.dst;
        if (bad_buf || (s.kind != 0 && s.kind != 1) || (s.kind == 1 && s.pad != 0 && s.pad != 1) ||
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
            // MFMA packing: element (kidx, n) of the im2col GEMM at [kidx / 4][n][kidx % 4], K padded to 16, N to 16
            const int K = s.k * s.c_in, K16 = (K + 15) & ~15, nt = (s.c_out + 15) / 16;
            if (e == hipSuccess && nt <= 5 && (size_t)K16 * nt * 16 * 4 <= 96 * 1024) {
                std::vector<float> wq((size_t)K16 * nt * 16, 0.0f);
                for (int co = 0; co < s.c_out; ++co)
                    for (int ci = 0; ci < s.c_in; ++ci)
                        for (int kk = 0; kk < s.k; ++kk) {
                            const int kidx = kk * s.c_in + ci;
                            wq[((size_t)(kidx / 4) * nt * 16 + co) * 4 + kidx % 4] = s.w[((size_t)co * s.c_in + ci) * s.k + kk];
                        }
                e = hipMalloc(reinterpret_cast<void**>(&o.d_wq), wq.size() * 4);
                if (e == hipSuccess) e = hipMemcpy(o.d_wq, wq.data(), wq.size() * 4, hipMemcpyHostToDevice);
                o.nt = nt;
            }
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
        if (o.d_wq) (void)hipFree(o.d_wq);
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
        if (o.kind == 0 && m->scalar_conv) {
            const int cq = (o.c_out + 3) / 4;
            const int64_t total = (int64_t)B * sh.t_out * cq;
            hipLaunchKernelGGL(seq_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, buf(o.src),
                               o.d_w, o.d_b, o.add >= 0 ? buf(o.add) : nullptr, buf(o.dst), B, sh.t_in, sh.t_out,
                               o.c_in, o.c_out, cq, o.k, o.stride, o.pad, o.relu);
        } else if (o.kind == 0 && o.d_wq && (int64_t)B * sh.t_in * o.c_in * 4 < 0x7fffffffLL) {
            const int64_t rows = (int64_t)B * sh.t_out;
            const int n_tiles = (int)((rows + 127) / 128);
            const int K = o.k * o.c_in, K16 = (K + 15) & ~15;
            const size_t lds = (size_t)K16 * o.nt * 16 * 4;
            auto fn = o.nt == 1 ? seq_conv_mfma_lds_kernel<1> : o.nt == 2 ? seq_conv_mfma_lds_kernel<2>
                    : o.nt == 3 ? seq_conv_mfma_lds_kernel<3> : o.nt == 4 ? seq_conv_mfma_lds_kernel<4> : seq_conv_mfma_lds_kernel<5>;
            RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            // small workgroups (256 threads, <= 96 KB LDS, few registers): as many per CU as the weights in LDS allow
            const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1)));
            const int grid = std::min(n_tiles, m->num_cu * per_cu);
            hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, st, buf(o.src), (unsigned)((int64_t)B * sh.t_in * o.c_in * 4),
                               o.d_wq, o.d_b, o.add >= 0 ? buf(o.add) : nullptr, buf(o.dst), B, sh.t_in, sh.t_out, o.c_in,
                               o.c_out, K, o.stride, o.pad, o.relu, n_tiles);
        } else if (o.kind == 0) {
            const int cq = (o.c_out + 3) / 4;
            const int64_t rows = (int64_t)B * sh.t_out;
            const int nt = o.c_out <= 16 ? 1 : o.c_out <= 32 ? 2 : 4;
            const dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((o.c_out + 16 * nt - 1) / (16 * nt)));
            auto fn = nt == 1 ? seq_conv_mfma_kernel<1> : nt == 2 ? seq_conv_mfma_kernel<2> : seq_conv_mfma_kernel<4>;
            hipLaunchKernelGGL(fn, grid, dim3(256), 0, st, buf(o.src), o.d_w, o.d_b, o.add >= 0 ? buf(o.add) : nullptr,
                               buf(o.dst), B, sh.t_in, sh.t_out, o.c_in, o.c_out, cq * 4, o.k * o.c_in, o.stride, o.pad,
                               o.relu);
        } else {
            const int64_t total = (int64_t)B * sh.t_out * sh.c;
            hipLaunchKernelGGL(seq_maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                               buf(o.src), buf(o.dst), B, sh.t_in, sh.t_out, sh.c, o.pad);
        }
        RS_HIP(hipGetLastError());
        last = o.dst;
    }
    if (C[last] != m->c_last) {
        set_error("rs_seqnet_forward: last buffer has %d channels, classifier expects %d", C[last], m->c_last);
        return RS_ERR_ARG;
    }
    hipLaunchKernelGGL(seq_head_kernel, dim3(B), dim3(256), 0, st, buf(last), T[last], m->c_last, m->d_fcw, m->d_fcb,
                       d_probs, d_logits);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

}  // extern "C"
