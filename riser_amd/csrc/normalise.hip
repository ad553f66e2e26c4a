// K1: MAD normalisation + outlier smoothing, one 512-thread workgroup per read.
//
// Restates SignalProcessor.mad_normalise (riser/preprocess.py:108-147) so that the
// float64 result is bit-identical to the reference's numpy path:
//   * median and MAD are exact ORDER STATISTICS of integers (np.median on an even-length
//     array is the mean of the two middle values): ONE histogram pass over the read (one bin per
//     value; a read whose value range exceeds the bins - a spike - is histogrammed over a clamped
//     window around a typical sample, checked, with a two-pass select as the fall-back) gives the
//     median, and the MAD is read off the same histogram's prefix sum;
//   * y = (x - med) / (1.4826 * mad) is evaluated in fp64 with IEEE division (non-outliers: each
//     distinct quotient once, into a look-up table);
//   * the outlier set {|y| > 3.5} is fixed before any update (:129) and rewritten in
//     ascending order in place (:130-138): the first sample of each run of consecutive outliers is
//     listed, one lane walks a run sequentially (left neighbour = already smoothed value, right
//     neighbour = original value), and a run longer than 16 samples is handed to a whole wave,
//     which skips stretches that sit at the clip limit 64 samples at a time - any run length is
//     handled exactly, and a 3000-sample plateau no longer costs the launch 0.8 ms.
// The read is staged once into LDS (2 B/sample) and every later pass runs out of LDS; what bounds the
// kernel is LDS issue and scan latency, not HBM ("K1 in phases").
#include "common.hpp"

#include <atomic>
#include <stdio.h>

// Diagnostic build only (-DRS_K1_STAMPS, tools/k1_stamps.py): s_memtime at the phase boundaries of workgroup 0
#ifdef RS_K1_STAMPS
__device__ unsigned long long g_k1_stamps[16];
#define RS_K1_STAMP(k)                                                                   \
    do {                                                                                 \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_k1_stamps[k] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define RS_K1_STAMP(k) do { } while (0)
#endif

namespace rs {
namespace {

constexpr int kMaxDevices = 64;
constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kBins = 4096;            // 8 bins per thread
constexpr int kSubBins = 32;           // refinement pass: keys < 2^17 -> shift <= 5
constexpr int kHeadCap = 1024;         // outlier-run heads listed in LDS per read (more: the walk scans the read instead)
constexpr int kLongCap = 64;           // runs handed from a lane to a whole wave (longer than kLaneSteps samples)
constexpr int kLaneSteps = 16;

struct Scratch {
    int wave_tot[kWaves];
    int wave_tot2[kWaves];
    int red_a[kWaves];
    int red_b[kWaves];
    int sel_bin[2];
    int sel_rank[2];
    int result[2];
    unsigned sub[2][kSubBins];
    int dthr;
    int n_heads;                       // outlier runs found by the output pass
    int n_long;                        // ... of which handed to a wave
    int pad_;
    double denom;
};

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive prefix of v over the workgroup (thread order)
__device__ __forceinline__ int block_excl_scan(int v, int* wave_tot, int tid) {
    const int lane = tid & 63, w = tid >> 6;
    const int incl = wave_incl_scan(v, lane);
    if (lane == 63) wave_tot[w] = incl;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < w; ++i) base += wave_tot[i];
    __syncthreads();
    return base + incl - v;
}

// Order statistics k_lo <= k_hi (0-based) of the n integer keys key(i) in [0, R].
// One histogram pass over (key >> shift) with shift chosen so that at most kBins bins are
// used; when shift > 0 (value range wider than 4096) a second pass resolves the low bits
// inside the (at most two) selected bins.
// KeepPrefix: when the pass was exact (shift == 0: one bin per key value) the histogram is left behind as its INCLUSIVE
// PREFIX SUM (hist[v] = number of keys <= v) and the function returns true; the caller reads rank counts from it.
template <bool KeepPrefix = false, class KeyFn>
__device__ bool block_select2(KeyFn key, int n, int R, int k_lo, int k_hi, unsigned* hist, Scratch* sc,
                              int tid, int& out_lo, int& out_hi) {
    const int bits = 32 - __clz(R | 1);
    const int shift = bits > 12 ? bits - 12 : 0;
    for (int i = tid; i < kBins; i += kThreads) hist[i] = 0u;
    if (tid < 2 * kSubBins) (&sc->sub[0][0])[tid] = 0u;
    __syncthreads();
    // Every thread walks its own contiguous run of the read (odd length: neighbouring lanes' 2-byte LDS reads fall
    // into different banks): the lanes of one ds_add then hold samples ~chunk apart.  Nanopore signal dwells on a
    // level for tens of samples, so with lanes on CONSECUTIVE samples most lanes of an instruction hit the same bin and
    // the LDS atomic unit serialises them.
    const int chunk = ((n + kThreads - 1) / kThreads) | 1;
    // (reading four keys ahead of their four atomics was measured: +40 % on the pass.  Round 4 tried to cut the atomics
    // instead - histogram a SAMPLE (every 8th key), bound a window around the wanted ranks, count exactly in registers,
    // histogram only the ~9 % of keys inside the window, with lane-replicated counters: exact, and TWICE as slow (19-21 k
    // cycles per select against 10-12 k; in-kernel stamps: sample 1.2 k, each of the two block scans 3 k, the counting pass
    // 7 k, the window pass 5.5 k).  A pass over a read is ~5 k cycles of per-element VALU work and barrier / scan latency at
    // this occupancy whatever it does with the key - it is not the atomic unit that bounds the selects, and more passes
    // with fewer atomics lose.  DESIGN.md 5, K1 in phases.)
    for (int k = 0, i = tid * chunk; k < chunk; ++k, ++i)
        if (i < n) atomicAdd(&hist[key(i) >> shift], 1u);
    __syncthreads();
    int loc[8];
    int s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        loc[j] = (int)hist[tid * 8 + j];
        s += loc[j];
    }
    const int excl = block_excl_scan(s, sc->wave_tot, tid);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const int k = which ? k_hi : k_lo;
        if (excl <= k && k < excl + s) {
            int c = excl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (k >= c && k < c + loc[j]) {
                    sc->sel_bin[which] = tid * 8 + j;
                    sc->sel_rank[which] = k - c;
                }
                c += loc[j];
            }
        }
    }
    __syncthreads();
    const int bin_lo = sc->sel_bin[0], bin_hi = sc->sel_bin[1];
    if (shift == 0) {
        out_lo = bin_lo;
        out_hi = bin_hi;
        if constexpr (KeepPrefix) {
            int c = excl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                c += loc[j];
                hist[tid * 8 + j] = (unsigned)c;
            }
            __syncthreads();
        }
        return true;
    }
    const int mask = (1 << shift) - 1;
    for (int k = 0, i = tid * chunk; k < chunk; ++k, ++i) {
        if (i >= n) break;
        const int kk = key(i);
        const int hb = kk >> shift;
        if (hb == bin_lo) atomicAdd(&sc->sub[0][kk & mask], 1u);
        if (hb == bin_hi) atomicAdd(&sc->sub[1][kk & mask], 1u);
    }
    __syncthreads();
    if (tid < 2) {
        const int rank = sc->sel_rank[tid];
        int c = 0, res = 0;
        for (int j = 0; j <= mask; ++j) {
            const int h = (int)sc->sub[tid][j];
            if (rank >= c && rank < c + h) res = j;
            c += h;
        }
        sc->result[tid] = ((tid ? bin_hi : bin_lo) << shift) | res;
    }
    __syncthreads();
    out_lo = sc->result[0];
    out_hi = sc->result[1];
    __syncthreads();
    return false;
}

// Block plan of the packed activation layout (common.hpp: BlockPlan).  Read b takes nblk(b) = len / U + 1 blocks of
// U = 1 << shift samples (at least one zero sample behind the read: the conv's 'same' pad and the halo row of the
// next read), its first block is the sum over the reads before it.  Called by every thread of a workgroup.
__device__ __forceinline__ int plan_nblk(int n, const BlockPlan& pl) {
    return pl.uniform_nblk > 0 ? pl.uniform_nblk : (n >> pl.shift) + 1;
}

// one workgroup of 1024 threads: the plan without a normalise launch (rs_forward: the signals arrive normalised)
__global__ __launch_bounds__(1024) void plan_kernel(const int32_t* __restrict__ len, int B, int lmax, const BlockPlan pl) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += 1024) {
        const int b = b0 + tid;
        const int n = b < B ? min(len[b], lmax) : 0;
        const int nb = b < B ? plan_nblk(n, pl) : 0;
        const int incl = wave_incl_scan(nb, lane);
        if (lane == 63) wave_tot[w] = incl;
        __syncthreads();
        int base = carry_s;
        for (int i = 0; i < w; ++i) base += wave_tot[i];
        base += incl - nb;
        __syncthreads();
        if (tid == 1023) carry_s = base + nb;
        if (b < B) {
            pl.rbase[b] = base;
            if (b == B - 1) pl.rbase[B] = base + nb;
            // a read whose blocks do not fit the table (device length above the host's copy) is DROPPED: its slots inside the
            // table become empty blocks of itself, so that every entry the host's block count covers names a valid read
            const bool fits = base + nb <= pl.nb_total;
            for (int j = 0; j < nb && base + j < pl.nb_total; ++j) {
                pl.blen[base + j] = fits ? max(0, min(n - (j << pl.shift), 1 << pl.shift)) : 0;
                pl.bread[base + j] = b;
            }
            if (b == B - 1)                                       // device lengths BELOW the host's: the unused tail
                for (int k = base + nb; k < pl.nb_total; ++k) {
                    pl.blen[k] = 0;
                    pl.bread[k] = 0;
                }
        }
        __syncthreads();
    }
}

// A read's entries of one block table (every thread of the read's workgroup calls it).  Device lengths BELOW the host's
// copy leave a tail of the table unused - empty blocks of read 0, written by the last read - so that every entry the host's
// block count covers names a valid read (the conv kernels index len[] with it); a read whose blocks do not fit the table
// (device length ABOVE the host's copy, which sized the workspace) is DROPPED: its slots inside the table become empty blocks
// of itself and false is returned.
__device__ __forceinline__ bool write_block_table(const BlockPlan& pl, int b, int n, int base, int nblk, bool last_read, int tid) {
    if (tid == 0) {
        pl.rbase[b] = base;
        if (last_read) pl.rbase[b + 1] = base + nblk;
    }
    if (last_read)
        for (int k = base + nblk + tid; k < pl.nb_total; k += kThreads) {
            pl.blen[k] = 0;
            pl.bread[k] = 0;
        }
    if (base + nblk > pl.nb_total) {
        for (int k = base + tid; k < pl.nb_total; k += kThreads) {
            pl.blen[k] = 0;
            pl.bread[k] = b;
        }
        return false;
    }
    for (int j = tid; j < nblk; j += kThreads) {
        pl.blen[base + j] = max(0, min(n - (j << pl.shift), 1 << pl.shift));
        pl.bread[base + j] = b;
    }
    return true;
}

__global__ __launch_bounds__(kThreads) void normalise_kernel(
    const int16_t* __restrict__ sig, const int64_t* __restrict__ off, const int32_t* __restrict__ len,
    float* __restrict__ out32, int64_t ld32, int32_t pad_to, double* __restrict__ out64, int64_t ld64,
    double* __restrict__ stats, int lmax, int zero_prefix, const BlockPlan pl, const BlockPlan pl2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    RS_K1_STAMP(0);
    // rs_classify lays the fp32 rows out behind 16 zero bytes: the conv kernel that folds layer 0
    // into its staging reads x[-1] of the first read from there
    if (zero_prefix && out32 && blockIdx.x == 0 && threadIdx.x < 4) out32[(int)threadIdx.x - 4] = 0.0f;
    unsigned* hist = reinterpret_cast<unsigned*>(smem);
    Scratch* sc = reinterpret_cast<Scratch*>(smem + kBins * 4);
    int16_t* sx0 = reinterpret_cast<int16_t*>(smem + kBins * 4 + 512);       // 16-byte aligned

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int n = min(len[b], lmax);                  // len > Lmax violates the contract; never overrun LDS
    const int16_t* src = sig + off[b];
    // element i of the read lives at sx[i]; sx is offset so that it has the same 16-byte phase as src
    int16_t* sx = sx0 + (int)((reinterpret_cast<uintptr_t>(src) >> 1) & 7);
    float* o32 = out32 ? out32 + (int64_t)b * ld32 : nullptr;
    double* o64 = out64 ? out64 + (int64_t)b * ld64 : nullptr;
    // packed block layout: the read's first block is the number of blocks of the reads before it.  The length loads are
    // issued here and reduced together with min / max below, so their latency hides behind the staging of the read.
    // (two tables when the conv stack runs its early layers on finer blocks than its late ones: pl = the layout of the
    // normalised rows and of the early layers, pl2 = the late layers'; the two sums travel in one 64-bit reduction)
    long long nblk_before = 0;
    if (pl.rbase && pl.uniform_nblk <= 0)
        for (int i = tid; i < b; i += kThreads) {
            const int li = min(len[i], lmax);
            nblk_before += (long long)((li >> pl.shift) + 1) | ((long long)(pl2.rbase ? (li >> pl2.shift) + 1 : 0) << 32);
        }
    RS_K1_STAMP(1);
    // ---- stage the read into LDS, min / max on the way ------------------------------------
    // 16-byte loads over the 16-byte-aligned body of the read (the LDS copy is shifted by the same
    // phase, so aligned global chunks are aligned LDS chunks), element loads for the ragged ends
    int mn = 32767, mx = -32768;
    {
        const int phase = (int)((reinterpret_cast<uintptr_t>(src) >> 1) & 7);     // elements past a 16-byte boundary
        const int head = min(n, (8 - phase) & 7);                                 // elements before the first aligned chunk
        const int n_vec = (n - head) >> 3;
        for (int i = tid; i < head; i += kThreads) {
            const int v = src[i];
            sx[i] = (int16_t)v;
            mn = min(mn, v);
            mx = max(mx, v);
        }
        const uint4* src4 = reinterpret_cast<const uint4*>(src + head);
        for (int c = tid; c < n_vec; c += kThreads) {
            const uint4 q = src4[c];
            *reinterpret_cast<uint4*>(sx + head + 8 * c) = q;
            const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int lo = (int)(int16_t)(w[k] & 0xffffu), hi = (int)(int16_t)(w[k] >> 16);
                mn = min(mn, min(lo, hi));
                mx = max(mx, max(lo, hi));
            }
        }
        for (int i = head + 8 * n_vec + tid; i < n; i += kThreads) {
            const int v = src[i];
            sx[i] = (int16_t)v;
            mn = min(mn, v);
            mx = max(mx, v);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mn = min(mn, __shfl_xor(mn, d, 64));
        mx = max(mx, __shfl_xor(mx, d, 64));
        nblk_before += __shfl_xor(nblk_before, d, 64);
    }
    if (lane == 0) {
        sc->red_a[w] = mn;
        sc->red_b[w] = mx;
        sc->wave_tot[w] = (int)(nblk_before & 0xffffffffLL);
        sc->wave_tot2[w] = (int)(nblk_before >> 32);
    }
    __syncthreads();
    mn = sc->red_a[0];
    mx = sc->red_b[0];
    int nblk_before1 = sc->wave_tot[0], nblk_before2 = sc->wave_tot2[0];
#pragma unroll
    for (int i = 1; i < kWaves; ++i) {
        mn = min(mn, sc->red_a[i]);
        mx = max(mx, sc->red_b[i]);
        nblk_before1 += sc->wave_tot[i];
        nblk_before2 += sc->wave_tot2[i];
    }
    __syncthreads();
    if (pl.rbase) {
        // this read's fp32 row starts at its first block, is zero-filled to the end of its last one, and the workgroup
        // writes the read's entries of the block table(s) the conv stack runs on
        const bool last_read = b == (int)gridDim.x - 1;
        const int base = pl.uniform_nblk > 0 ? b * pl.uniform_nblk : nblk_before1;
        const int nblk = plan_nblk(n, pl);
        bool fits = write_block_table(pl, b, n, base, nblk, last_read, tid);
        if (pl2.rbase) {
            const int base2 = pl2.uniform_nblk > 0 ? b * pl2.uniform_nblk : nblk_before2;
            fits = write_block_table(pl2, b, n, base2, plan_nblk(n, pl2), last_read, tid) && fits;
        }
        if (!fits) return;                        // dropped (the head kernel reports NaN for it): never write outside the workspace
        o32 = out32 + ((int64_t)base << pl.shift);
        pad_to = nblk << pl.shift;
    }

    RS_K1_STAMP(2);
    // ---- median: the two middle order statistics -------------------------------------------
    const int k_lo = (n - 1) >> 1, k_hi = n >> 1;
    int m_lo, m_hi;
    // One bin per sample value needs a value range <= kBins.  A read with ONE spike beyond that (an open pore, a glitch) would
    // pay the two-pass select twice over (+17 us on the launch, measured) - so such a read is histogrammed over a WINDOW of
    // kBins values around a typical sample (the median of three), values outside clamped to the window's edge bins.  The
    // prefix sums stay exact for every interior value (a clamped-low sample IS <= it, a clamped-high one is not), so the
    // median and the MAD read off them are exact as long as they are found in the interior - checked, with the full-range
    // select as the fall-back.
    int w0 = mn, top = mx - mn;                                  // bin 0's value; the last bin whose prefix sum is exact
    bool windowed = false;
    bool exact_hist;
    if (mx - mn >= kBins) {
        const int s0 = sx[n >> 2], s1 = sx[n >> 1], s2 = sx[n - 1 - (n >> 2)];
        const int mid = max(min(s0, s1), min(max(s0, s1), s2));
        w0 = mid - kBins / 2;
        windowed = true;
        exact_hist = block_select2<true>([&](int i) { return min(max((int)sx[i] - w0, 0), kBins - 1); }, n, kBins - 1, k_lo, k_hi,
                                         hist, sc, tid, m_lo, m_hi);
        top = kBins - 2;
        if (m_lo < 1 || m_hi > top) {                            // the median is not inside the window (block-uniform)
            w0 = mn;
            top = mx - mn;
            windowed = false;
            exact_hist = block_select2<true>([&](int i) { return (int)sx[i] - mn; }, n, mx - mn, k_lo, k_hi, hist, sc, tid,
                                             m_lo, m_hi);
        }
    } else {
        exact_hist = block_select2<true>([&](int i) { return (int)sx[i] - mn; }, n, mx - mn, k_lo, k_hi, hist, sc, tid, m_lo,
                                         m_hi);
    }
    const int sum2 = m_lo + m_hi + 2 * w0;                       // 2 * median, exact

    RS_K1_STAMP(3);
    // ---- MAD: middle order statistics of |2x - 2 med| (integers < 2^17) --------------------
    int d_lo, d_hi;
    bool mad_done = false;
    if (exact_hist) {
        // No second pass over the read: hist[v] = #{x - w0 <= v}, and the samples within d of the median are a RANGE of
        // values, so  #{|2x - 2 med| <= d} = hist[hi(d)] - hist[lo(d) - 1].  All deviations have the parity q of
        // sp = m_lo + m_hi; for d = 2 e + q the range is [b - e, a + e] with a = (sp + q) / 2, b = (sp - q) / 2.  The k-th
        // smallest deviation is the smallest d whose count exceeds k: every thread tests 8 consecutive e.
        // (Windowed: a count that reaches outside the exact prefix sums can only be too LARGE, so the smallest e found is
        // the true one whenever it lies inside; if it does not, the full-range select below takes over.)
        const int sp = m_lo + m_hi, q = sp & 1, a = (sp + q) >> 1, b2 = (sp - q) >> 1;
        const int last = windowed ? kBins - 1 : top;
        auto upto = [&](int v) { return v < 0 ? 0 : (int)hist[min(v, last)]; };
        auto within = [&](int e) { return upto(a + e) - upto(b2 - e - 1); };
        const int e0 = tid * 8;
        int prev = e0 == 0 ? 0 : within(e0 - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = within(e0 + j);
            if (c > k_lo && prev <= k_lo) sc->result[0] = e0 + j;
            if (c > k_hi && prev <= k_hi) sc->result[1] = e0 + j;
            prev = c;
        }
        __syncthreads();
        const int e_lo = sc->result[0], e_hi = sc->result[1];
        __syncthreads();
        d_lo = 2 * e_lo + q;
        d_hi = 2 * e_hi + q;
        mad_done = !windowed || (a + e_hi <= top && b2 - e_hi - 1 >= 0);
    }
    if (!mad_done) {
        const int rd = max(abs(2 * mn - sum2), abs(2 * mx - sum2));
        block_select2([&](int i) { return abs(2 * (int)sx[i] - sum2); }, n, rd, k_lo, k_hi, hist, sc, tid, d_lo, d_hi);
    }
    const int mad4 = d_lo + d_hi;                                // 4 * mad, exact

    RS_K1_STAMP(4);
    if (stats && tid == 0) {
        stats[2 * b + 0] = (double)sum2 * 0.5;
        stats[2 * b + 1] = (double)mad4 * 0.25;
    }

    if (mad4 == 0) {                                             // riser/preprocess.py:123-124
        for (int i = tid; i < n; i += kThreads) {
            if (o32) o32[i] = 0.0f;
            if (o64) o64[i] = 0.0;
        }
    } else {
        // smallest integer deviation d (in half counts) with fp64((d/2)/denom) > 3.5: the
        // outlier predicate of :129 becomes an integer compare (IEEE division by a positive
        // constant is monotonic and sign-symmetric).
        if (tid == 0) {
            const double denom = 1.4826 * ((double)mad4 * 0.25);
            long long d = (long long)floor(7.0 * denom);
            if (d < 0) d = 0;
            while (d > 0 && ((double)(d - 1) * 0.5) / denom > 3.5) --d;
            while (!(((double)d * 0.5) / denom > 3.5)) ++d;
            sc->denom = denom;
            sc->dthr = (int)(d > 0x7fffffffLL ? 0x7fffffffLL : d);
            sc->n_heads = 0;
            sc->n_long = 0;
        }
        __syncthreads();
        const double denom = sc->denom;
        const int dthr = sc->dthr;
        auto dev2 = [&](int j) { return 2 * (int)sx[j] - sum2; };
        auto yv = [&](int j) { return ((double)dev2(j) * 0.5) / denom; };
        auto put = [&](int j, double v) {
            if (o32) o32[j] = (float)v;
            if (o64) o64[j] = v;
        };
        // The non-outlier values are (t / 2) / denom for the integers |t| < dthr.  When only the fp32
        // output is wanted and the threshold is small (it is ~7 * 1.4826 * MAD: a few hundred for
        // nanopore signal), each distinct quotient is computed once - same fp64 division, same rounding
        // to fp32 - into the histogram's LDS, and the per-sample work becomes a look-up.
        float* lut = reinterpret_cast<float*>(hist);
        const bool use_lut = o32 && !o64 && dthr <= kBins / 2;
        RS_K1_STAMP(5);
        if (use_lut) {
            for (int k = tid; k < 2 * dthr; k += kThreads) lut[k] = (float)(((double)(k - dthr) * 0.5) / denom);
            __syncthreads();
        }
        RS_K1_STAMP(6);
        // Pass 1, every sample: non-outliers are written (look-up or division), the FIRST sample of each run of
        // consecutive outliers is listed in LDS.  Pass 2, one lane per listed run: the sequential recurrence of
        // riser/preprocess.py:128-147 along the run.  (Through round 3 the run walk sat inside the sample loop: a
        // wave paid for it on every iteration in which one of its 64 samples opened a run - a quarter of the
        // iterations on nanopore-like data - and the loop could not be pipelined around the divergent walk.)
        int* heads = reinterpret_cast<int*>(sx0 + ((lmax + 15) & ~7));
        auto is_head = [&](int i) { return i == 0 || abs(dev2(i - 1)) < dthr; };
        // one step of riser/preprocess.py:128-147 at sample j of a run, the smoothed value of j - 1 in `prev`
        auto step = [&](int j, double prev) -> double {
            double nv;
            if (j == 0) {
                nv = yv(1);                                      // :132 (not clipped)
            } else if (j == n - 1) {
                nv = prev;                                       // :134 (not clipped)
            } else {
                nv = (prev + yv(j + 1)) * 0.5;                   // :136
                nv = nv > 3.5 ? 3.5 : (nv < -3.5 ? -3.5 : nv);   // :141-147
            }
            return nv;
        };
        // A run LONGER than kLaneSteps (a stall, an open pore, what a fixed trim leaves of the adapter: thousands of
        // consecutive outliers) is handed to a whole wave (below): one lane walking it was the slowest workgroup of the
        // launch by a factor of ten (0.8 ms for a 3000-sample plateau).
        int* long_at = heads + kHeadCap;
        double* long_prev = reinterpret_cast<double*>(long_at + kLongCap);
        auto walk = [&](int i) {
            double prev = i > 0 ? yv(i - 1) : 0.0;
            int j = i;
            int steps = 0;
            do {
                if (++steps > kLaneSteps) {
                    const int slot = atomicAdd(&sc->n_long, 1);
                    if (slot < kLongCap) {
                        long_at[slot] = j;
                        long_prev[slot] = prev;
                        return;
                    }
                    steps = -0x40000000;                         // list full: this lane walks its run to the end
                }
                const double nv = step(j, prev);
                put(j, nv);
                prev = nv;
                ++j;
            } while (j < n && abs(dev2(j)) >= dthr);
        };
        // The same recurrence by a wave, every lane holding the same (j, prev).  Once the smoothed value sits AT the clip
        // limit, it stays there for as long as the NEXT sample is an outlier of the same sign: (3.5 + y) / 2 >= 3.5 for every
        // y > 3.5 in fp64 (the sum is >= 7 after rounding, halving is exact), clipped back to 3.5 - so such a stretch needs no
        // arithmetic: 64 samples are tested and written per step.  Everything else is the sequential step, unchanged.
        auto wave_walk = [&](int j, double prev) {
            while (j < n && abs(dev2(j)) >= dthr) {
                if (prev == 3.5 || prev == -3.5) {
                    const int sgn = prev > 0.0 ? 1 : -1;
                    int took;
                    do {
                        const int idx = j + lane;
                        const bool ok = idx >= 1 && idx < n - 1 && sgn * dev2(min(idx + 1, n - 1)) >= dthr;
                        const unsigned long long m = __ballot(ok);
                        took = m == ~0ull ? 64 : __builtin_ctzll(~m);
                        if (lane < took) put(idx, prev);
                        j += took;
                    } while (took == 64);
                    if (!(j < n && abs(dev2(j)) >= dthr)) break;   // (cannot happen: sample j is still the run's, see above)
                }
                const double nv = step(j, prev);
                if (lane == 0) put(j, nv);
                prev = nv;
                ++j;
            }
        };
        auto list_head = [&](int i) {
            if (is_head(i)) {
                const int slot = atomicAdd(&sc->n_heads, 1);
                if (slot < kHeadCap) heads[slot] = i;
            }
        };
        if (use_lut) {
            const int top = 2 * dthr - 1;
            int i = tid;
            for (; i + 3 * kThreads < n; i += 4 * kThreads) {    // four samples in flight per lane
                int t[4];
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = dev2(i + u * kThreads);
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = lut[min(max(t[u] + dthr, 0), top)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (abs(t[u]) < dthr)
                        o32[i + u * kThreads] = v[u];
                    else
                        list_head(i + u * kThreads);
                }
            }
            for (; i < n; i += kThreads) {
                const int t = dev2(i);
                if (abs(t) < dthr)
                    o32[i] = lut[t + dthr];
                else
                    list_head(i);
            }
        } else {
            for (int i = tid; i < n; i += kThreads) {
                const int t = dev2(i);
                if (abs(t) < dthr)
                    put(i, ((double)t * 0.5) / denom);
                else
                    list_head(i);
            }
        }
        __syncthreads();
        RS_K1_STAMP(9);
        const int nh = sc->n_heads;
        if (nh <= kHeadCap) {
            for (int k = tid; k < nh; k += kThreads) walk(heads[k]);
        } else {                                                 // more runs than the list holds: find them again
            for (int i = tid; i < n; i += kThreads)
                if (abs(dev2(i)) >= dthr && is_head(i)) walk(i);
        }
        __syncthreads();
        const int nl = min(sc->n_long, kLongCap);
        for (int k = w; k < nl; k += kWaves) wave_walk(long_at[k], long_prev[k]);
    }
    RS_K1_STAMP(7);
    if (o32)
        for (int i = n + tid; i < pad_to; i += kThreads) o32[i] = 0.0f;
    RS_K1_STAMP(8);
}

}  // namespace

int launch_plan(const int32_t* d_len, int B, int Lmax, const BlockPlan& plan, hipStream_t st) {
    if (B <= 0) return RS_OK;
    hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(1024), 0, st, d_len, B, Lmax, plan);
    RS_HIP(hipGetLastError());
    return RS_OK;
}

int launch_normalise(const int16_t* d_sig, const int64_t* d_off, const int32_t* d_len, int B, int Lmax,
                     float* d_out32, int64_t ld32, int32_t pad_to, double* d_out64, int64_t ld64,
                     double* d_stats, hipStream_t st, int zero_prefix, const BlockPlan* plan, const BlockPlan* plan2) {
    if (B <= 0) return RS_OK;
    if (Lmax < 1 || Lmax > kMaxNormLen) {
        set_error("rs_normalise: read length %d outside [1, %d]", Lmax, kMaxNormLen);
        return RS_ERR_LENGTH;
    }
    const size_t lds = (size_t)kBins * 4 + 512 + (size_t)round_up(Lmax + 8, 8) * 2 + (size_t)kHeadCap * 4 + (size_t)kLongCap * 12;
    // the > 64 KiB dynamic-LDS limit is a per-DEVICE function attribute: raise it once on every device that
    // launches the kernel (a process may hold models on several GPUs)
    static std::atomic<bool> attr_set[kMaxDevices];
    int dev = 0;
    RS_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices || !attr_set[dev].load(std::memory_order_acquire)) {
        RS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(normalise_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (dev >= 0 && dev < kMaxDevices) attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(normalise_kernel, dim3(B), dim3(kThreads), lds, st, d_sig, d_off, d_len, d_out32, ld32,
                       pad_to, d_out64, ld64, d_stats, Lmax, zero_prefix, plan ? *plan : BlockPlan{}, plan2 ? *plan2 : BlockPlan{});
    RS_HIP(hipGetLastError());
#ifdef RS_K1_STAMPS
    {
        RS_HIP(hipStreamSynchronize(st));
        unsigned long long h[16];
        RS_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_k1_stamps), sizeof(h)));
        fprintf(stderr, "[k1-stamps] B %d Lmax %d cycles: plan %llu | stage+minmax %llu | median %llu | mad %llu | thr %llu | lut %llu | "
                "normalise %llu + run walk %llu | pad %llu | total %llu\n", B, Lmax, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3],
                h[5] - h[4], h[6] - h[5], h[9] - h[6], h[7] - h[9], h[8] - h[7], h[8] - h[0]);
    }
#endif
    return RS_OK;
}

}  // namespace rs
