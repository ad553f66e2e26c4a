#!/usr/bin/env python3
"""Throughput with TWO batches in flight (two HIP streams, two model handles = two workspaces) against one:
    python tools/two_stream_probe.py [dtype ...]      (default f32w f16 bf16x3)
Each call is a full 512 x 16000 rs_classify; nothing is shared between the two streams but the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
B, L = int(os.environ.get("RS_B", 512)), 16000
dev = torch.device("cuda", 0)
sigs = synth.make_signals(20260103, B, L)
sig, off, ln, lens = pack_reads(list(sigs), dev)
for dt in (sys.argv[1:] or ["f32w", "f16", "bf16x3"]):
    ms = [Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype=dt, device=dev) for _ in range(2)]
    outs = [torch.empty((B, 2), device=dev) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    def run(n_streams, steps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for k in range(steps):
            j = k % n_streams
            with torch.cuda.stream(streams[j]):
                ms[j].classify_raw(sig, off, ln, lens, out=outs[j])
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / steps
    for n in (1, 2): run(n, 30)
    t1, t2 = run(1, 100), run(2, 100)
    ref = outs[0].clone()
    print("%-7s one stream %.4f ms/step = %.0f chunks/s; two streams %.4f ms/step = %.0f chunks/s (x%.3f); results equal: %s" % (
        dt, t1 * 1e3, B / t1, t2 * 1e3, B / t2, t1 / t2, bool(torch.equal(outs[0], outs[1]))))
    for m in ms: m.close()
