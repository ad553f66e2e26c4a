#!/usr/bin/env python3
"""One 512-read batch as ONE call vs as TWO half-batch calls on two HIP streams (fork / join per step):
    python tools/fork_probe.py [dtype ...]      RS_B, RS_L, RS_MIXED as in layer_times.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
B, L = int(os.environ.get("RS_B", 512)), int(os.environ.get("RS_L", 16000))
dev = torch.device("cuda", 0)
sigs = synth.make_signals(20260103, B, L)
sig, off, ln, lens = pack_reads(list(sigs), dev)
if os.environ.get("RS_MIXED"):
    lens = np.array([(L // 2, 3 * L // 4, L)[i % 3] for i in range(B)], dtype=np.int32)
    ln = torch.from_numpy(lens).to(dev)
for dt in (sys.argv[1:] or ["f32w"]):
    m = Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype=dt, device=dev)
    out = torch.empty((B, 2), device=dev)
    out2 = torch.empty((B, 2), device=dev)
    side = [torch.cuda.Stream(device=dev) for _ in range(2)]
    cur = torch.cuda.current_stream(dev)
    def one():
        m.classify_raw(sig, off, ln, lens, out=out)
    def forked(parts):
        def f():
            bounds = [B * k // parts for k in range(parts + 1)]
            for k in range(parts):
                s = side[k % 2]
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    lo, hi = bounds[k], bounds[k + 1]
                    m.classify_raw(sig, off[lo:hi], ln[lo:hi], lens[lo:hi], out=out2[lo:hi])
            for s in side: cur.wait_stream(s)
        return f
    def run(fn, steps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / steps
    res = {}
    for name, fn in (("one", one), ("fork2", forked(2)), ("fork4", forked(4))):
        run(fn, 30)
    for rep in range(3):
        for name, fn in (("one", one), ("fork2", forked(2)), ("fork4", forked(4))):
            res.setdefault(name, []).append(run(fn, 100))
    t1 = min(res["one"])
    print(dt, "B", B, " ".join("%s %.4f ms (x%.3f)" % (k, min(v) * 1e3, t1 / min(v)) for k, v in res.items()),
          "equal", bool(torch.equal(out, out2)), flush=True)
    m.close()
