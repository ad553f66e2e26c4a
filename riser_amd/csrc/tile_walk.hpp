// Persistent tile walk shared by the tiled conv kernels (conv_f32 / conv_wino / conv_wino4 / conv_ring_h16).
//
// A launch has n_mtiles x n_ntiles output tiles and (at most) one workgroup per CU.  Round k of the walk gives
// workgroup w the ORDER INDEX k * nwg + block(w) + slot_k(w), where block(w) is the contiguous range of nwg / 8
// indices of the workgroup's XCD (= blockIdx & 7: the 32 workgroups that share an L2) and the slot inside the block
// is rotated by 5 per round when dead tiles can exist (the all-padding tiles of shorter reads sit at fixed positions
// of every read's slot - a power-of-two period, like nwg - and a fixed stride would hand some workgroups nothing but
// dead tiles).
//
// Order index -> tile: n-major (gm == 0), or, when the grid fills the chip, RECTANGLES: every XCD block of a round
// is gm row tiles x gn channel tiles (gm * gn = nwg / 8), so the workgroups sharing an L2 re-use gm activation slabs
// and gn weight slabs per K chunk instead of streaming nwg / 8 different activation slabs against one weight slab.
// plan_walk picks gm x gn by the bytes an XCD pulls over the fabric per round, never adding a round; rectangles
// that overhang the tile grid contain invalid order indices, which the kernels skip.  Measured (fp32 F(4,3) path,
// B = 512): conv-stack FETCH + WRITE 4.37 -> 3.12 GB per step, -1...3 % time on the late layers.
#pragma once
#include <stdint.h>
#include <stdlib.h>

namespace rs {

struct WalkArgs {
    int n_mtiles, n_ntiles;
    int gm, gn, n_mb;        // rectangle order (gm == 0: n-major)
    int q_total;             // order indices in all (>= n_mtiles * n_ntiles)
    int check_dead;          // 0: no tile can be all padding (skip the test and the rotation)
    int m_base;              // first row unit (the kernel's own: pooled rows / F(4,3) groups) of this launch's row tiles: a
                             // layer may run as a HEAD launch of whole rounds of a large tile shape and a TAIL launch of a
                             // smaller one over the rows behind it (plan_tail_split below)
};

// host: fill a WalkArgs for a launch of `grid` workgroups on `num_cu` CUs.  x_rows / w_rows = rows per K channel
// of one tile's activation / weight slab (their ratio decides the rectangle's aspect).  rect_order = false (RS_NO_RECT_ORDER=1
// when the model was created) keeps the n-major order.
inline WalkArgs plan_walk(int n_mtiles, int n_ntiles, int64_t grid, int num_cu, double x_rows, double w_rows,
                          int check_dead, bool rect_order = true) {
    WalkArgs w{};
    w.n_mtiles = n_mtiles;
    w.n_ntiles = n_ntiles;
    w.check_dead = check_dead;
    const int64_t tiles = (int64_t)n_mtiles * n_ntiles;
    w.q_total = (int)tiles;
    if (!rect_order || grid != num_cu || num_cu % 8 != 0 || n_ntiles <= 1) return w;
    const int rect = num_cu / 8;
    const int64_t rounds = (tiles + num_cu - 1) / num_cu;
    double best = 1e300;
    for (int gn = 1; gn <= rect; ++gn) {
        if (rect % gn) continue;
        const int gm = rect / gn;
        const int64_t n_mb = (n_mtiles + gm - 1) / gm, n_nb = (n_ntiles + gn - 1) / gn;
        const int64_t q_total = n_mb * n_nb * rect;
        if ((q_total + num_cu - 1) / num_cu != rounds) continue;             // never pay an extra round
        const double fetch = (double)(n_mb * n_nb) * (gm * x_rows + gn * w_rows);
        if (fetch < best) {
            best = fetch;
            w.gm = gm;
            w.gn = gn;
            w.n_mb = (int)n_mb;
            w.q_total = (int)q_total;
        }
    }
    return w;
}

// A launch costs (rounds over the CUs) x (one tile): 300 tiles of the best shape on 256 CUs cost two full rounds for 1.17
// rounds of work.  Host-side search shared by the fp32 Winograd kernels: run `head_mtiles` row tiles of shape h - as many
// whole rounds as the grid holds - and leave the rows behind them to a second launch with the shape that suits THAT row
// count best.  Every output keeps its accumulation order (it does not depend on the tile shape: the batch-invariance
// tests), so the bits are those of the single launch.  cost units: the planners' SIMD cycles.
struct TailSplit {
    int head_shape = -1;     // -1: single launch
    int head_mtiles = 0;
    int tail_shape = -1;
    double cost = 0.0;
};

// tile_cost(k) / bm(k) / ntiles(k): per-tile cost, row units per tile and channel tiles of shape k (k < n_shapes, < 0 cost =
// shape unusable); best_single(rows, &cost) -> best shape for a launch over `rows` row units.
template <class TileCost, class Bm, class Ntiles, class BestSingle>
inline TailSplit plan_tail_split(int n_shapes, int64_t rows, int num_cu, double single_cost, TileCost tile_cost, Bm bm,
                                 Ntiles ntiles, BestSingle best_single, double margin = 0.97) {
    constexpr double kLaunch = 6000.0;        // a second launch: boundary + its own prologue
    TailSplit out;
    out.cost = single_cost;
    for (int h = 0; h < n_shapes; ++h) {
        const double tc = tile_cost(h);
        if (tc < 0) continue;
        const int64_t n_m = (rows + bm(h) - 1) / bm(h), n_n = ntiles(h);
        const int64_t tiles = n_m * n_n;
        const int64_t full = tiles / num_cu;                     // whole rounds the launch holds
        if (full < 1 || tiles % num_cu == 0) continue;
        const int64_t m1 = full * num_cu / n_n;                  // row tiles of those rounds
        if (m1 < 1 || m1 >= n_m) continue;
        const int64_t head_rounds = (m1 * n_n + num_cu - 1) / num_cu;
        double tail_cost = 0.0;
        const int t = best_single(rows - m1 * bm(h), &tail_cost);
        if (t < 0) continue;
        const double cost = head_rounds * tc + tail_cost + kLaunch;
        if (cost < margin * single_cost && cost < out.cost) {          // the margin is against ONE launch; the best split wins
            out.cost = cost;
            out.head_shape = h;
            out.head_mtiles = (int)m1;
            out.tail_shape = t;
        }
    }
    return out;
}

#ifdef __HIPCC__
// device: the workgroup's position in the walk
struct TileWalk {
    int nwg, blk, blk_base, slot, round_base;
    __device__ __forceinline__ TileWalk() {
        nwg = gridDim.x;
        const bool xcd = (nwg & 7) == 0;
        blk = xcd ? nwg >> 3 : nwg;
        blk_base = xcd ? (int)(blockIdx.x & 7) * blk : 0;
        slot = xcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
        round_base = 0;
    }
    // this workgroup's order index of the next round
    __device__ __forceinline__ int next_index(const WalkArgs& a) {
        const int q = round_base + blk_base + slot;
        round_base += nwg;
        if (a.check_dead) {                                    // rotate only when dead tiles can exist (costs ~1 %)
            slot += 5 % blk;
            if (slot >= blk) slot -= blk;
        }
        return q;
    }
};

// order index -> (row tile, channel tile); false for the invalid indices of overhanging rectangles
__device__ __forceinline__ bool walk_tile(const WalkArgs& a, int q, int& mi, int& nt) {
    if (a.gm == 0) {
        nt = a.n_ntiles == 1 ? 0 : q / a.n_mtiles;
        mi = q - nt * a.n_mtiles;
    } else {
        const int rect = a.gm * a.gn;
        const int bq = q / rect, w = q - bq * rect;
        const int ln = w / a.gm, lm = w - ln * a.gm;
        const int nb = bq / a.n_mb, mb = bq - nb * a.n_mb;
        mi = mb * a.gm + lm;
        nt = nb * a.gn + ln;
    }
    return mi < a.n_mtiles && nt < a.n_ntiles;
}
#endif

}  // namespace rs
