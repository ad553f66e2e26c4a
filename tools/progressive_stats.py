#!/usr/bin/env python3
"""SURVEY 8(f) N3 'measure first'.  A read is re-classified at growing prefixes (offline: riser/test.py:202-224, +1 s per
step; live: the AccumulatingCache client re-sends a read, longer, until it is decided).  Work of an earlier visit can be
reused BIT-EXACTLY only where its inputs are unchanged:
  * the poly(A) end (a property of the prefix before it): cached per read id, as the reference does;
  * (median, MAD): if BOTH are unchanged the normalised prefix is unchanged, and so are the conv activations of every
    row whose receptive field ended before the old length (all but the last few rows of each layer);
  * the median alone (incremental histogram of the new samples) saves one of K1's two select passes.
This script measures how often that happens on STRUCTURED reads (synth.make_raw_read: adapter, poly(A) plateau, RNA with
level structure, spikes; trimmed at the poly(A) end as the control loop does) and on the stationary bench signals, and
what fraction of the conv FLOPs a perfect reuse scheme would save.

    python tools/progressive_stats.py [n_reads]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from riser_amd import synth
from oracle import riser_oracle as ro

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
CH = synth.CHANNELS


def layer_flops(L):
    out, c_in = [], 1
    for c in CH:
        out.append(2.0 * c_in * c * 3 * L)
        c_in, L = c, L // 2
    return out


def reusable_fraction(l_old, l_new):
    """conv FLOPs of the l_new visit that a cache of the l_old visit's activations covers when the normalised prefix is
    bit-identical.  Input rows < R_i of layer i are unchanged (R_0 = l_old: the old visit saw zero padding at l_old);
    pooled output row t reads input rows 2t-1 .. 2t+2, so R_{i+1} = (R_i - 2) // 2 - an edge of about two rows at every
    depth."""
    tot = sum(layer_flops(l_new))
    c_in, saved, R = 1, 0.0, l_old
    for i, c in enumerate(CH):
        R_out = max(0, (R - 2) // 2)                 # unchanged output rows of this layer
        saved += 2.0 * c_in * c * 3 * (2 * R_out)    # their conv rows need not be recomputed
        c_in, R = c, R_out
    return saved / tot


def study(name, reads, steps):
    same_both = {s: 0 for s in steps}
    same_med = {s: 0 for s in steps}
    n_used = {s: 0 for s in steps}
    for sig in reads:
        st = {}
        for a, b in steps:
            if len(sig) < b:
                continue
            for L in (a, b):
                if L not in st:
                    st[L] = ro.median_mad(sig[:L])
            n_used[(a, b)] += 1
            same_med[(a, b)] += st[a][0] == st[b][0]
            same_both[(a, b)] += st[a] == st[b]
    print(f"--- {name}")
    exp = 0.0
    for s in steps:
        n = max(n_used[s], 1)
        fr = reusable_fraction(*s)
        print(f"  {s[0]:5d} -> {s[1]:5d}: {n_used[s]:4d} reads; (median, MAD) identical {100 * same_both[s] / n:5.1f} %; "
              f"median alone {100 * same_med[s] / n:5.1f} %; reusable conv FLOPs if identical {100 * fr:4.1f} % "
              f"-> expected saving {100 * fr * same_both[s] / n:4.1f} % of that visit")


# (1) the stationary bench signals at the offline schedule (2 s -> 3 s -> 4 s)
sig = synth.make_signals(20260103, N, 16000)
study("stationary bench signals, +1 s steps (riser/test.py:202-224)", list(sig), [(8000, 12000), (12000, 16000)])

# (2) structured raw reads, trimmed at the poly(A) end (riser/control.py:36-60), live schedule: a read is re-seen every
# ReadUntil batch (~0.4 s = 1600 samples at 4 kHz) until it reaches the 8615-sample cap
reads = []
rid = 0
while len(reads) < N:
    raw = synth.make_raw_read(4242, rid, 26000, polya=True)
    rid += 1
    end = ro.polya_end(raw)
    if end:
        reads.append(raw[end + 1:])
live = [(4096 + 1600 * k, min(4096 + 1600 * (k + 1), 8615)) for k in range(3)]
study("structured reads (adapter / poly(A) trimmed), live schedule +0.4 s up to the RNA004 cap", reads, live)
study("structured reads, offline schedule +1 s", reads, [(8000, 12000), (12000, 16000)])
