#!/usr/bin/env python3
"""The same batch N times through a model: every output must be the first one's bits (a race in a kernel's LDS ring or counted
waits shows up as a differing run).  python tools/determinism_soak.py [dtype ...] [runs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
dts = [a for a in sys.argv[1:] if not a.isdigit()] or ["f32w", "bf16x3", "f16x3", "f16xf8"]
runs = int(next((a for a in sys.argv[1:] if a.isdigit()), 300))
dev = torch.device("cuda", 0)
rng = np.random.default_rng(9)
cases = [("512 x 16000", [16000] * 512), ("357 x 8615", [8615] * 357), ("300 ragged", [int(n) for n in rng.integers(4096, 16001, size=300)]),
         ("40 ragged", [int(n) for n in rng.integers(4096, 16001, size=40)]), ("1 x 16000", [16000])]
bad = 0
for dt in dts:
    m = Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype=dt, device=dev)
    for name, lens in cases:
        pool = synth.make_signals(20260103, min(len(lens), 64), 16000)
        sigs = [pool[i % len(pool)][:n] for i, n in enumerate(lens)]
        sig, off, ln, lh = pack_reads(sigs, dev)
        first = m.classify_raw(sig, off, ln, lh, return_logits=True)
        first = (first[0].clone(), first[1].clone())
        diff = 0
        for _ in range(runs):
            g = m.classify_raw(sig, off, ln, lh, return_logits=True)
            diff += not (torch.equal(g[0], first[0]) and torch.equal(g[1], first[1]))
        bad += diff
        print(f"{dt:7s} {name:12s}: {runs} runs, {diff} differing", flush=True)
    m.close()
print(f"{bad} differing runs")
sys.exit(1 if bad else 0)
