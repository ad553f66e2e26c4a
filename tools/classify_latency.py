#!/usr/bin/env python3
"""End-to-end latency of the reference's own call shape: Model.classify(normalised signal) at batch 1 (riser/model.py:22-28),
host wall time from the numpy signal to the two probabilities on the host, plus the raw-signal form (normalise + forward in
one library call).   python tools/classify_latency.py [f32w f16x3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import riser_oracle as ro
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
dev = torch.device("cuda", 0)
for dt in sys.argv[1:] or ["f32w", "f16x3"]:
    m = Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype=dt, device=dev)
    for L in (16000, 8615, 4096):
        s = synth.make_signals(20260103, 1, L)[0]
        x = ro.mad_normalise(s)
        for _ in range(20): m.classify(x).cpu()
        lat = []
        for _ in range(200):
            t = time.perf_counter(); p = m.classify(x).cpu(); lat.append(time.perf_counter() - t)
        sig, off, ln, lh = pack_reads([s], dev)
        out = torch.empty((1, 2), device=dev)
        for _ in range(20): m.classify_raw(sig, off, ln, lh, out=out)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(200): m.classify_raw(sig, off, ln, lh, out=out)
        torch.cuda.synchronize(); dev_ms = (time.perf_counter() - t) / 200 * 1e3
        lat = np.asarray(lat) * 1e3
        print(f"{dt} L={L}: Model.classify(x).cpu() p50 {np.percentile(lat, 50):.3f} ms p99 {np.percentile(lat, 99):.3f} ms | "
              f"classify_raw resident, back to back {dev_ms:.3f} ms per read", flush=True)
    m.close()
