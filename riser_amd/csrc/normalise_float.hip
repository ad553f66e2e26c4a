// K1f: MAD normalisation + outlier smoothing of FLOATING-POINT signals (float16, float32 or float64), one 256-thread workgroup
// per read.  The live path feeds raw int16 ADC counts (normalise.hip); this kernel covers the other input the
// reference's SignalProcessor.mad_normalise accepts (riser/preprocess.py:108-147): the retrain path normalises
// pA-scaled float signals (riser/retrain/preprocess.py:79).  numpy keeps the input's precision end to end (NEP 50:
// the Python float 1.4826 adopts the array's dtype), so everything here is evaluated in T:
//   med = np.median(x)                       exact order statistics (mean of the two middle values for even n, in T)
//   mad = np.median(np.abs(x - med))         |x - med| rounded in T, then the same select
//   y   = (x - med) / (T(1.4826) * mad)      one rounding per operation (-ffp-contract=off, IEEE division)
//   the sequential in-place smoothing of {|y| > 3.5} of normalise.hip, in T.
// float16: numpy evaluates every half-precision operation as float32(a) op float32(b) rounded to half, and np.median's
// mean of the two middle values with a float32 accumulator rounded once - Arith<_Float16> spells exactly that.
// Order statistics of floats: MSB-first radix select (8 bits per pass) on the order-preserving integer image of the
// value, over the read in global memory (it is a few tens of KB: L2-resident after the first pass).
// Off the hot path: clarity over speed.
#include "common.hpp"

namespace rs {
namespace {

constexpr int kThreads = 256;

// one operation of numpy on scalars / arrays of dtype T
template <class T> struct Arith {
    __device__ static T add(T a, T b) { return a + b; }
    __device__ static T sub(T a, T b) { return a - b; }
    __device__ static T mul(T a, T b) { return a * b; }
    __device__ static T div(T a, T b) { return a / b; }
    __device__ static T mean2(T a, T b) { return (T)((T)(a + b) / (T)2); }     // np.mean([a, b]) in T
};
template <> struct Arith<_Float16> {
    typedef _Float16 H;
    __device__ static H add(H a, H b) { return (H)((float)a + (float)b); }
    __device__ static H sub(H a, H b) { return (H)((float)a - (float)b); }
    __device__ static H mul(H a, H b) { return (H)((float)a * (float)b); }
    __device__ static H div(H a, H b) { return (H)((float)a / (float)b); }
    __device__ static H mean2(H a, H b) { return (H)(((float)a + (float)b) / 2.0f); }   // float32 accumulator, one rounding
};

template <class T> struct Bits;
template <> struct Bits<_Float16> {
    typedef unsigned type;
    static constexpr int kPasses = 2;
    __device__ static unsigned key(_Float16 v) {
        const unsigned u = (unsigned)__builtin_bit_cast(unsigned short, v);
        return (u & 0x8000u) ? (~u & 0xffffu) : (u | 0x8000u);
    }
    __device__ static _Float16 value(unsigned k) {
        const unsigned u = (k & 0x8000u) ? (k & 0x7fffu) : (~k & 0xffffu);
        return __builtin_bit_cast(_Float16, (unsigned short)u);
    }
};
template <> struct Bits<float> {
    typedef unsigned type;
    static constexpr int kPasses = 4;
    __device__ static unsigned key(float v) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    __device__ static float value(unsigned k) {
        const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
        return __builtin_bit_cast(float, u);
    }
};
template <> struct Bits<double> {
    typedef unsigned long long type;
    static constexpr int kPasses = 8;
    __device__ static unsigned long long key(double v) {
        const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
        return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    }
    __device__ static double value(unsigned long long k) {
        const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
        return __builtin_bit_cast(double, u);
    }
};

// k-th smallest (0-based) of the n values f(0..n-1); every thread returns it
template <class T, class F>
__device__ T block_select(F f, int n, int k, unsigned* hist, int* sel, int tid) {
    typedef typename Bits<T>::type K;
    K prefix = 0;
    for (int pass = 0; pass < Bits<T>::kPasses; ++pass) {
        const int shift = 8 * (Bits<T>::kPasses - 1 - pass);
        hist[tid] = 0u;
        __syncthreads();
        for (int i = tid; i < n; i += kThreads) {
            const K key = Bits<T>::key(f(i));
            if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int c = 0, b = 0;
            for (; b < 255; ++b) {
                const int h = (int)hist[b];
                if (k < c + h) break;
                c += h;
            }
            sel[0] = b;
            sel[1] = k - c;
        }
        __syncthreads();
        prefix = (prefix << 8) | (K)sel[0];
        k = sel[1];
        __syncthreads();
    }
    return Bits<T>::value(prefix);
}

template <class T>
__global__ __launch_bounds__(kThreads) void normalise_float_kernel(const T* __restrict__ sig, const int64_t* __restrict__ off,
                                                                   const int32_t* __restrict__ len, T* __restrict__ out,
                                                                   int64_t ld, double* __restrict__ stats) {
    __shared__ unsigned hist[256];
    __shared__ int sel[2];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = len[b];
    const T* x = sig + off[b];
    T* o = out + (int64_t)b * ld;
    const int k_lo = (n - 1) >> 1, k_hi = n >> 1;

    auto median_of = [&](auto f) -> T {
        const T lo = block_select<T>(f, n, k_lo, hist, sel, tid);
        const T hi = k_hi == k_lo ? lo : block_select<T>(f, n, k_hi, hist, sel, tid);
        return k_hi == k_lo ? lo : Arith<T>::mean2(lo, hi);         // np.mean of the two middle values
    };
    const T med = median_of([&](int i) { return x[i]; });
    const T mad = median_of([&](int i) {
        const T d = Arith<T>::sub(x[i], med);
        return d < (T)0 ? -d : d;
    });
    if (stats && tid == 0) {
        stats[2 * b + 0] = (double)med;
        stats[2 * b + 1] = (double)mad;
    }
    if (mad == (T)0) {                                               // riser/preprocess.py:123-124
        for (int i = tid; i < n; i += kThreads) o[i] = (T)0;
        return;
    }
    const T denom = Arith<T>::mul((T)1.4826, mad);                   // the Python float adopts the array's dtype
    const T lim = (T)3.5;
    auto yv = [&](int j) -> T { return Arith<T>::div(Arith<T>::sub(x[j], med), denom); };
    auto is_out = [&](int j) {
        const T y = yv(j);
        return (y < (T)0 ? -y : y) > lim;
    };
    for (int i = tid; i < n; i += kThreads) {
        if (!is_out(i)) {
            o[i] = yv(i);
            continue;
        }
        if (i > 0 && is_out(i - 1)) continue;                        // inside a run: its head writes it
        T prev = i > 0 ? yv(i - 1) : (T)0;
        int j = i;
        do {
            T nv;
            if (j == 0) {
                nv = yv(1);                                          // :132 (not clipped)
            } else if (j == n - 1) {
                nv = prev;                                           // :134 (not clipped)
            } else {
                nv = Arith<T>::div(Arith<T>::add(prev, yv(j + 1)), (T)2);    // :136
                nv = nv > lim ? lim : (nv < -lim ? -lim : nv);       // :141-147
            }
            o[j] = nv;
            prev = nv;
            ++j;
        } while (j < n && is_out(j));
    }
}

}  // namespace

int launch_normalise_float(const void* d_sig, int elem_bytes, const int64_t* d_off, const int32_t* d_len, int B, void* d_out,
                           int64_t ld, double* d_stats, hipStream_t st) {
    if (B <= 0) return RS_OK;
    if (elem_bytes == 2)
        hipLaunchKernelGGL(normalise_float_kernel<_Float16>, dim3(B), dim3(kThreads), 0, st,
                           static_cast<const _Float16*>(d_sig), d_off, d_len, static_cast<_Float16*>(d_out), ld, d_stats);
    else if (elem_bytes == 4)
        hipLaunchKernelGGL(normalise_float_kernel<float>, dim3(B), dim3(kThreads), 0, st, static_cast<const float*>(d_sig),
                           d_off, d_len, static_cast<float*>(d_out), ld, d_stats);
    else
        hipLaunchKernelGGL(normalise_float_kernel<double>, dim3(B), dim3(kThreads), 0, st,
                           static_cast<const double*>(d_sig), d_off, d_len, static_cast<double*>(d_out), ld, d_stats);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

}  // namespace rs
