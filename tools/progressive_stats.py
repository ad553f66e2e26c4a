#!/usr/bin/env python3
"""SURVEY 8(f) N3 'measure first': when a read is re-classified at a longer prefix (2 s -> 3 s -> 4 s,
riser/test.py:202-224), how often do the normalisation statistics (median, MAD) stay exactly the
same, i.e. how often could activations of the unchanged prefix be reused bit-exactly?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from riser_amd import synth
from oracle import riser_oracle as ro
N = 400
sig = synth.make_signals(20260103, N, 16000)
same = {(8000, 12000): 0, (12000, 16000): 0}
dmed, dmad = [], []
for s in sig:
    st = {L: ro.median_mad(s[:L]) for L in (8000, 12000, 16000)}
    for a, b in same:
        if st[a] == st[b]:
            same[(a, b)] += 1
    dmed.append(abs(st[12000][0] - st[16000][0])); dmad.append(abs(st[12000][1] - st[16000][1]))
print({f"{a}->{b}": f"{v}/{N} identical (median, MAD)" for (a, b), v in same.items()})
print("3 s -> 4 s: |d median| mean %.3f, |d MAD| mean %.3f ADC counts" % (np.mean(dmed), np.mean(dmad)))
