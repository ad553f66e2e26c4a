#!/usr/bin/env python3
"""Does a kernel sequence run slower after the GPU sat idle?  The same classify call (304 ragged reads), device time by HIP
events, back to back and with host-side pauses between calls (the control loop leaves ~0.7 ms between its batches)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
lens = np.clip(rng.normal(7777, 1200, size=304).astype(np.int64), 4096, 8615)
base = synth.make_signals(20260103, 64, 8615)
sigs = [base[i % 64][: lens[i]] for i in range(304)]
sig, off, ln, lh = pack_reads(sigs, dev)
for dt in (sys.argv[1:] or ["f32w", "bf16x3"]):
    m = Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype=dt, device=dev)
    for _ in range(20):
        m.classify_raw(sig, off, ln, lh)
    torch.cuda.synchronize()
    for pause_ms in (0.0, 0.2, 0.7, 2.0, 10.0):
        ts = []
        for _ in range(60):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            m.classify_raw(sig, off, ln, lh)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
            if pause_ms:
                t = time.perf_counter()
                while (time.perf_counter() - t) * 1e3 < pause_ms:
                    pass
        print(f"{dt}: pause {pause_ms:5.1f} ms -> device time per call median {np.median(ts[10:]):.3f} ms (min {np.min(ts[10:]):.3f})", flush=True)
    from riser_amd.model import classify_raw_ensemble
    dec = torch.empty(304, dtype=torch.uint8, device=dev)
    out = torch.empty((1, 304, 2), dtype=torch.float32, device=dev)
    ts = []
    for _ in range(60):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        classify_raw_ensemble([m], sig, off, ln, lh.astype(np.int32), out=out, decision=dec, max_len=8615)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
        t = time.perf_counter()
        while (time.perf_counter() - t) * 1e3 < 0.7:
            pass
    print(f"{dt}: ensemble entry (1 model, decision), pause 0.7 ms -> {np.median(ts[10:]):.3f} ms", flush=True)
    m.close()
