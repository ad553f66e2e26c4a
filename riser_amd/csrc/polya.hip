// poly(A) end detector (riser/preprocess.py:42-79) for a batch of raw reads.
// One 256-thread workgroup per read.  Each wave takes every 4th 500-sample window, sorts
// it (bitonic, 512 int keys in LDS, padded with +inf) to get the exact median, sorts the
// integer deviations |2x - 2med| to get the exact MAD, and records the window sum; one
// lane then replays the reference's sequential start/end rule over the window table in
// fp64 with the reference's operation order.  No length limit: the table is filled in passes.
#include "common.hpp"

namespace rs {
namespace {

constexpr int kWin = 500;                          // _TRIM_RESOLUTION
constexpr int kChunkWin = 128;                     // windows per pass over the table (64000 samples)
constexpr int kPad = 0x7fffffff;

__device__ __forceinline__ void bitonic512(int* keys, int lane) {
    for (int k = 2; k <= 512; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = lane + 64 * u;
                const int i = 2 * p - (p & (j - 1));
                const int q = i + j;
                const int a = keys[i], b = keys[q];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[q] = a;
                }
            }
            __syncthreads();
        }
    }
}

// Reads of ANY length (an AccumulatingCache read grows for as long as the strand is in the pore): the window table is
// filled in passes of kChunkWin windows; after each pass one lane replays the reference's sequential rule over the
// pass, carrying (start, end) and the sums of the last two windows (the 1000-sample rolling mean) into the next one.
// Once an end is found nothing later can change it (riser/preprocess.py:66 sets it only while it is None), so the
// scan stops there.
__global__ __launch_bounds__(256) void polya_kernel(const int16_t* __restrict__ sig, const int64_t* __restrict__ off,
                                                    const int32_t* __restrict__ len, int32_t* __restrict__ out) {
    __shared__ int keys[4][512];
    __shared__ int wsum[kChunkWin + 2];            // [0], [1]: the two windows before the pass
    __shared__ int wmad4[kChunkWin];
    __shared__ int state[2];                       // start, end (-1: none yet)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = len[b];
    const int16_t* src = sig + off[b];
    const int nw = n / kWin;
    int* K = keys[wave];
    if (tid == 0) {
        state[0] = -1;
        state[1] = -1;
    }
    __syncthreads();
    for (int c0 = 0; c0 < nw; c0 += kChunkWin) {
        const int cn = min(kChunkWin, nw - c0);
        for (int w0 = 0; w0 < cn; w0 += 4) {
            const int wl = w0 + wave;                              // window of this wave inside the pass
            const bool act = wl < cn;
            const int16_t* wsrc = src + (int64_t)(c0 + wl) * kWin;
            int s = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = lane + 64 * u;
                int v = kPad;
                if (act && i < kWin) {
                    v = wsrc[i];
                    s += v;
                }
                K[i] = v;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
            __syncthreads();
            bitonic512(K, lane);
            const int sum2 = K[kWin / 2 - 1] + K[kWin / 2];            // 2 * median
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = lane + 64 * u;
                const int v = K[i];
                K[i] = (v == kPad) ? kPad : abs(2 * v - sum2);
            }
            __syncthreads();
            bitonic512(K, lane);
            if (act && lane == 0) {
                wsum[2 + wl] = s;
                wmad4[wl] = K[kWin / 2 - 1] + K[kWin / 2];              // 4 * MAD
            }
            __syncthreads();
        }
        if (tid == 0) {
            int start = state[0], end = state[1];
            for (int wl = 0; wl < cn && end <= 0; ++wl) {
                const int w = c0 + wl;
                const int i = w * kWin;
                const double mad = (double)wmad4[wl] * 0.25;
                const double mean = (double)wsum[2 + wl] / 500.0;
                double rolling = mean;
                if (i > 2 * kWin) rolling = (double)(wsum[wl] + wsum[wl + 1]) / 1000.0;
                const double change = (mean - rolling) / rolling * 100.0;
                // `not polyA_start` is also true for index 0 (riser/preprocess.py:62)
                if (start <= 0 && change > 20.0 && mad <= 20.0) start = i;
                if (start > 0 && end <= 0 && mad > 20.0) end = i;
            }
            state[0] = start;
            state[1] = end;
            if (cn >= 2) {
                wsum[0] = wsum[cn];
                wsum[1] = wsum[cn + 1];
            }
        }
        __syncthreads();
        if (state[1] > 0) break;
    }
    if (tid == 0) out[b] = state[1] > 0 ? state[1] : -1;
}

// segment k: src[src_off[k] .. + len[k]) -> dst[dst_off[k] ..); one workgroup per segment, 2-byte elements (the segments
// of a signal cache start at arbitrary sample positions)
__global__ __launch_bounds__(256) void copy_segments_kernel(const int16_t* __restrict__ src, int16_t* __restrict__ dst,
                                                            const int64_t* __restrict__ src_off,
                                                            const int64_t* __restrict__ dst_off,
                                                            const int32_t* __restrict__ len) {
    const int k = blockIdx.x;
    const int16_t* s = src + src_off[k];
    int16_t* d = dst + dst_off[k];
    const int n = len[k];
    for (int i = threadIdx.x; i < n; i += 256) d[i] = s[i];
}

}  // namespace

int launch_copy_segments(const int16_t* d_src, int16_t* d_dst, const int64_t* d_src_off, const int64_t* d_dst_off,
                         const int32_t* d_len, int n, hipStream_t st) {
    if (n <= 0) return RS_OK;
    hipLaunchKernelGGL(copy_segments_kernel, dim3(n), dim3(256), 0, st, d_src, d_dst, d_src_off, d_dst_off, d_len);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

int launch_polya(const int16_t* d_sig, const int64_t* d_off, const int32_t* d_len, int B, int32_t* d_end,
                 hipStream_t st) {
    if (B <= 0) return RS_OK;
    hipLaunchKernelGGL(polya_kernel, dim3(B), dim3(256), 0, st, d_sig, d_off, d_len, d_end);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

}  // namespace rs
