// poly(A) end detector (riser/preprocess.py:42-79) for a batch of raw reads.
// One 256-thread workgroup per read.  Each wave takes every 4th 500-sample window and needs its exact median and MAD:
//   * the window is sorted IN REGISTERS: 512 keys (500 samples + 12 x +inf), 8 per lane, a bitonic network whose
//     exchanges at distance < 8 stay inside the lane and whose 21 longer ones are lane permutes - no LDS traffic and no
//     barrier inside the sort (round 1-3 sorted 512 keys in LDS with a workgroup barrier per stage, twice per window);
//   * the MAD needs no second sort: the deviations |x - med| along the SORTED window fall to the median and rise again,
//     so "at least k + 1 deviations <= d" holds exactly when some run of k + 1 consecutive sorted samples has both of its
//     END points within d:  D_k = min_i max(dev(K[i]), dev(K[i + k])) - 251 candidates, four per lane, one wave minimum;
//   * the window sum rides along.
// One lane then replays the reference's sequential start/end rule over the window table in fp64 with the reference's
// operation order.  No length limit: the table is filled in passes.
#include "common.hpp"

namespace rs {
namespace {

constexpr int kWin = 500;                          // _TRIM_RESOLUTION
constexpr int kChunkWin = 128;                     // windows per pass over the table (64000 samples)
constexpr int kPad = 0x7fffffff;

__device__ __forceinline__ void cmpx(int& lo, int& hi) {        // ascending compare-exchange of two registers
    const int a = lo, b = hi;
    lo = min(a, b);
    hi = max(a, b);
}

// the sub-steps of a merge that stay inside a lane: partners at element distance J = 4, 2, 1
template <int J>
__device__ __forceinline__ void lane_steps(int (&v)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if ((r & J) == 0) cmpx(v[r], v[r ^ J]);
    if constexpr (J > 1) lane_steps<J / 2>(v);
}

// Ascending bitonic sort of 512 keys held as element e = 8 * lane + r.  Every merge of block size k starts with the
// "flip" (e <-> e ^ (k - 1): the block's second half reversed), then halves the distance; all exchanges are ascending
// (the lower element keeps the minimum), so no per-block direction is needed.
__device__ __forceinline__ void sort512(int (&v)[8], int lane) {
    // k = 2, 4, 8: inside the lane
#pragma unroll
    for (int r = 0; r < 8; r += 2) cmpx(v[r], v[r + 1]);
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if ((r & 2) == 0) cmpx(v[r], v[r ^ 3]);
    lane_steps<1>(v);
#pragma unroll
    for (int r = 0; r < 4; ++r) cmpx(v[r], v[7 - r]);
    lane_steps<2>(v);
    // k = 16 ... 512: d = k / 8 lanes per block
#pragma unroll
    for (int d = 2; d <= 64; d <<= 1) {
        {   // flip: partner lane ^ (d - 1), its register 7 - r
            const bool lower = (lane & (d >> 1)) == 0;
            int p[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) p[r] = __shfl_xor(v[7 - r], d - 1, 64);
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = lower ? min(v[r], p[r]) : max(v[r], p[r]);
        }
#pragma unroll
        for (int dl = d >> 2; dl >= 1; dl >>= 1) {            // partner lane ^ dl, same register
            const bool lower = (lane & dl) == 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int q = __shfl_xor(v[r], dl, 64);
                v[r] = lower ? min(v[r], q) : max(v[r], q);
            }
        }
        lane_steps<4>(v);
    }
}

// Reads of ANY length (an AccumulatingCache read grows for as long as the strand is in the pore): the window table is
// filled in passes of kChunkWin windows; after each pass one lane replays the reference's sequential rule over the
// pass, carrying (start, end) and the sums of the last two windows (the 1000-sample rolling mean) into the next one.
// Once an end is found nothing later can change it (riser/preprocess.py:66 sets it only while it is None), so the
// scan stops there.
//
// RESUMING a scan.  A read that stays in its pore comes back longer with every batch, and as long as no end has been found the
// reference scans it again from its first sample.  But the detector's state after W whole windows - W, `start`, the sums of
// the last two windows - is a function of the read's first 500 W samples alone, so a caller that knows the prefix is
// unchanged hands the state of the previous scan back (st_in, four ints per read: windows done, start, the two sums) and
// only the new windows are sorted.  st_out receives the state after this scan; a scan that FINDS an end hands its input state
// back unchanged (resumed again it finds the same end).  Both NULL: every read from its first sample, nothing kept.
__global__ __launch_bounds__(256) void polya_kernel(const int16_t* __restrict__ sig, const int64_t* __restrict__ off,
                                                    const int32_t* __restrict__ len, int32_t* __restrict__ out,
                                                    const int32_t* __restrict__ st_in, int32_t* __restrict__ st_out) {
    __shared__ int keys[4][512];
    __shared__ int wsum[kChunkWin + 2];            // [0], [1]: the two windows before the pass
    __shared__ int wmad4[kChunkWin];
    __shared__ int state[2];                       // start, end (-1: none yet)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = len[b];
    const int16_t* src = sig + off[b];
    const int nw = n / kWin;
    int* K = keys[wave];
    // windows already scanned (0 < w0 <= nw, else the state is not this read's: start over)
    int w0 = st_in ? st_in[4 * b] : 0;
    if (w0 < 0 || w0 > nw) w0 = 0;
    if (tid == 0) {
        state[0] = w0 > 0 ? st_in[4 * b + 1] : -1;
        state[1] = -1;
        wsum[0] = w0 > 0 ? st_in[4 * b + 2] : 0;
        wsum[1] = w0 > 0 ? st_in[4 * b + 3] : 0;
    }
    __syncthreads();
    for (int c0 = w0; c0 < nw; c0 += kChunkWin) {
        const int cn = min(kChunkWin, nw - c0);
        for (int w0 = 0; w0 < cn; w0 += 4) {
            const int wl = w0 + wave;                              // window of this wave inside the pass
            const bool act = wl < cn;
            const int16_t* wsrc = src + (int64_t)(c0 + wl) * kWin;
            int v[8];
            int s = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = 8 * lane + u;
                v[u] = kPad;
                if (act && i < kWin) {
                    v[u] = wsrc[i];
                    s += v[u];
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
            sort512(v, lane);
            // K is this wave's own array: its LDS accesses execute in program order, no barrier between the store of the
            // sorted keys and the look-ups below (or the previous window's look-ups and this store)
#pragma unroll
            for (int u = 0; u < 8; ++u) K[8 * lane + u] = v[u];
            __builtin_amdgcn_wave_barrier();
            const int sum2 = K[kWin / 2 - 1] + K[kWin / 2];                // 2 * median
            int d_lo = kPad, d_hi = kPad;                                  // order statistics kWin/2 - 1 and kWin/2 of |2x - 2 med|
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = lane + 64 * u;
                if (i + kWin / 2 - 1 < kWin) {
                    const int a = abs(2 * K[i] - sum2);
                    d_lo = min(d_lo, max(a, abs(2 * K[i + kWin / 2 - 1] - sum2)));
                    if (i + kWin / 2 < kWin) d_hi = min(d_hi, max(a, abs(2 * K[i + kWin / 2] - sum2)));
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                d_lo = min(d_lo, __shfl_xor(d_lo, d, 64));
                d_hi = min(d_hi, __shfl_xor(d_hi, d, 64));
            }
            if (act && lane == 0) {
                wsum[2 + wl] = s;
                wmad4[wl] = d_lo + d_hi;                                   // 4 * MAD
            }
        }
        __syncthreads();
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
            wsum[0] = wsum[cn];                                    // the sums of the last two windows seen so far (cn >= 1)
            wsum[1] = wsum[cn + 1];
        }
        __syncthreads();
        if (state[1] > 0) break;
    }
    if (tid == 0) {
        out[b] = state[1] > 0 ? state[1] : -1;
        if (st_out) {
            const bool found = state[1] > 0;
            st_out[4 * b + 0] = found ? w0 : nw;
            st_out[4 * b + 1] = found ? (w0 > 0 ? st_in[4 * b + 1] : -1) : state[0];
            st_out[4 * b + 2] = found ? (w0 > 0 ? st_in[4 * b + 2] : 0) : wsum[0];
            st_out[4 * b + 3] = found ? (w0 > 0 ? st_in[4 * b + 3] : 0) : wsum[1];
        }
    }
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
                 hipStream_t st, const int32_t* d_state_in, int32_t* d_state_out) {
    if (B <= 0) return RS_OK;
    hipLaunchKernelGGL(polya_kernel, dim3(B), dim3(256), 0, st, d_sig, d_off, d_len, d_end, d_state_in, d_state_out);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

}  // namespace rs
