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
// Workgroup = 4 waves stacked along M; wave tile = (16*MT) x (16*NT); K is processed in
// chunks of KC input channels: per chunk the workgroup stages a (BM+2) x KC slab of X (the
// +2 halo rows serve all three taps from ONE copy) and a BN x 3 x KC slab of packed weights
// through registers into LDS (single buffer of <= 80 KB, so two workgroups share a CU and one
// stages while the other computes), then issues 3*KC/4 k-steps of v_mfma_f32_16x16x4_f32 per
// (mt, nt) tile.  LDS rows are KC+2 floats (= 2 mod 4) so that
// the 16 rows x 2 k-groups a 32-lane half reads with ds_read_b32 hit 32 distinct banks.
#include "common.hpp"

namespace rs {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float* x;
    const float* w;        // packed [n_pad][nch][3][kc]
    const float* bias;     // [n_pad]
    float* y;
    const int32_t* len;
    int rows_in;           // B * P_in
    int P_out;             // P_in / 2
    int cp_in, cp_out;
    int kc, nch;
    int shift_out;         // valid output rows of read b: len[b] >> shift_out
    int n_mtiles, n_ntiles;
};

template <int MT, int NT>
__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvArgs a) {
    constexpr int BM = 64 * MT;
    constexpr int BN = 16 * NT;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 15, kq = lane >> 4;

    // XCD-aware tile order: the 8 XCDs each take a contiguous range of an n-tile-major order,
    // so workgroups sharing an L2 stream the same weight slab (bijective remap).
    const int nwg = a.n_mtiles * a.n_ntiles;
    int o;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
        o = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    }
    const int ntile = o / a.n_mtiles;
    const int mtile = o - ntile * a.n_mtiles;
    const int m0 = mtile * BM;
    const int n0 = ntile * BN;

    const int KC = a.kc;
    const int S = KC + 2;
    const int kc4 = KC >> 2;
    const int a_elems = (BM + 2) * S;

    // ---- staging: global -> registers -> LDS in float4 units ----------------------------------
    // Single LDS buffer; two workgroups are co-resident per CU (LDS <= 80 KB each) so one
    // stages while the other issues MFMAs.
    const int a_units = (BM + 2) * kc4;
    const int b_row_units = 3 * kc4;
    const int b_units = BN * b_row_units;
    float* const Abuf = lds;
    float* const Bbuf = lds + a_elems;

    auto stage = [&](int c) {
        const int cbase = c * KC;
#pragma unroll 4
        for (int f = tid; f < a_units; f += 256) {
            const int row = f / kc4, c4 = f - row * kc4;
            const int gr = m0 - 1 + row;
            const int ch = cbase + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr >= 0 && gr < a.rows_in && ch < a.cp_in)
                v = *reinterpret_cast<const float4*>(a.x + (int64_t)gr * a.cp_in + ch);
            float2* d = reinterpret_cast<float2*>(Abuf + row * S + 4 * c4);
            d[0] = make_float2(v.x, v.y);
            d[1] = make_float2(v.z, v.w);
        }
#pragma unroll 4
        for (int f = tid; f < b_units; f += 256) {
            const int n = f / b_row_units, rem = f - n * b_row_units;
            const int kw = rem / kc4, c4 = rem - kw * kc4;
            const float4 v =
                *reinterpret_cast<const float4*>(a.w + ((int64_t)(n0 + n) * a.nch + c) * (3 * KC) + 4 * rem);
            float2* d = reinterpret_cast<float2*>(Bbuf + (n * 3 + kw) * S + 4 * c4);
            d[0] = make_float2(v.x, v.y);
            d[1] = make_float2(v.z, v.w);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* As = Abuf + (wave * 16 * MT + r) * S + kq;       // + mt*16*S + kw*S + c0
    const float* Bs = Bbuf + r * 3 * S + kq;                      // + nt*48*S + kw*S + c0

    for (int c = 0; c < a.nch; ++c) {
        if (c) __syncthreads();                                   // everyone done reading the previous chunk
        stage(c);
        __syncthreads();
#pragma unroll 1
        for (int kw = 0; kw < 3; ++kw) {
            const float* Ak = As + kw * S;
            const float* Bk = Bs + kw * S;
#pragma unroll 2
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

    // ---- epilogue: bias + ReLU + MaxPool(2,2) in registers, masked store ---------------------
    float bias[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bias[j] = a.bias[n0 + j * 16 + r];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = m0 + wave * 16 * MT + i * 16 + 4 * kq;   // even; rows row..row+3 in regs 0..3
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int prow = (row >> 1) + h;                      // pooled output row
            if (2 * prow >= a.rows_in) continue;
            const int b = prow / a.P_out;
            const int p = prow - b * a.P_out;
            const bool valid = p < (a.len[b] >> a.shift_out);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = n0 + j * 16 + r;
                if (col < a.cp_out) {
                    const float v = fmaxf(fmaxf(acc[i][j][2 * h], acc[i][j][2 * h + 1]) + bias[j], 0.0f);
                    a.y[(int64_t)prow * a.cp_out + col] = valid ? v : 0.0f;
                }
            }
        }
    }
}

using KernelFn = void (*)(const ConvArgs);

template <int MT, int NT>
KernelFn get_kernel() {
    return conv_f32_kernel<MT, NT>;
}

KernelFn lookup_kernel(int mt, int nt) {
#define RS_CASE(M, N) \
    if (mt == M && nt == N) return get_kernel<M, N>();
    RS_CASE(1, 2) RS_CASE(1, 3) RS_CASE(1, 4) RS_CASE(1, 5) RS_CASE(1, 6) RS_CASE(1, 7) RS_CASE(1, 8)
    RS_CASE(2, 2) RS_CASE(2, 3) RS_CASE(2, 4) RS_CASE(2, 5) RS_CASE(2, 6) RS_CASE(2, 7) RS_CASE(2, 8)
    RS_CASE(4, 2) RS_CASE(4, 3) RS_CASE(4, 4) RS_CASE(4, 5) RS_CASE(4, 6) RS_CASE(4, 7) RS_CASE(4, 8)
#undef RS_CASE
    return nullptr;
}

}  // namespace

size_t conv_f32_lds_bytes(int mt, int nt, int kc, int nch) {
    (void)nch;
    return (size_t)((64 * mt + 2) + 48 * nt) * (kc + 2) * sizeof(float);
}

int launch_conv_f32(const ConvLayerDev& L, const float* d_x, float* d_y, const int32_t* d_len, int B, int P_in,
                    int layer_index, hipStream_t st, int* bm_out, int* bn_out) {
    const ConvPlan& p = L.plan;
    KernelFn fn = lookup_kernel(p.mt, p.nt);
    if (!fn || p.kc < 4 || (p.kc & 3)) {
        set_error("conv_f32: no kernel for tile mt=%d nt=%d kc=%d", p.mt, p.nt, p.kc);
        return RS_ERR_ARG;
    }
    const int64_t rows64 = (int64_t)B * P_in;
    if (rows64 > 0x7fffffff) {
        set_error("conv_f32: batch too large (%lld rows)", (long long)rows64);
        return RS_ERR_ARG;
    }
    ConvArgs a;
    a.x = d_x;
    a.w = static_cast<const float*>(L.d_w);
    a.bias = L.d_bias;
    a.y = d_y;
    a.len = d_len;
    a.rows_in = (int)rows64;
    a.P_out = P_in / 2;
    a.cp_in = L.cp_in;
    a.cp_out = L.cp_out;
    a.kc = p.kc;
    a.nch = p.nch;
    a.shift_out = layer_index + 1;
    const int BM = 64 * p.mt, BN = 16 * p.nt;
    a.n_mtiles = (a.rows_in + BM - 1) / BM;
    a.n_ntiles = p.n_pad / BN;
    const size_t lds = conv_f32_lds_bytes(p.mt, p.nt, p.kc, p.nch);
    if (lds > 160 * 1024) {
        set_error("conv_f32: LDS %zu too large", lds);
        return RS_ERR_ARG;
    }
    RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
    const unsigned grid = (unsigned)(a.n_mtiles * a.n_ntiles);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, st, a);
    RS_HIP(hipGetLastError());
    if (bm_out) *bm_out = BM;
    if (bn_out) *bn_out = BN;
    return RS_OK;
}

}  // namespace rs
