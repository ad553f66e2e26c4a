#!/usr/bin/env python3
"""One batch of B reads as ONE forward against the same reads as two sub-batches run side by side on two streams
(the second half's kernels fill the tails of the first's): ms per batch.
    python tools/split_batch_probe.py [dtype ...]      RS_BS="576 640 704" picks the batch sizes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
L = 16000
dev = torch.device("cuda", 0)
BS = [int(b) for b in os.environ.get("RS_BS", "357 448 512 576 640 704 768 1280").split()]
all_sigs = synth.make_signals(20260103, max(BS), L)
for dt in (sys.argv[1:] or ["f32w", "bf16x3"]):
    ms = [Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype=dt, device=dev) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for B in BS:
        def packed(lo, hi):
            return pack_reads(list(all_sigs[lo:hi]), dev)
        whole = packed(0, B)
        out_w = torch.empty((B, 2), device=dev)
        def run_whole(steps):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(steps): ms[0].classify_raw(*whole, out=out_w)
            torch.cuda.synchronize(); return (time.perf_counter() - t) / steps
        run_whole(10); tw = run_whole(50)
        line = "%-7s B=%4d whole %.4f ms (%.0f/s)" % (dt, B, tw * 1e3, B / tw)
        cuts = sorted({B // 2, (B // 2 + 63) // 64 * 64, 512 if B > 512 else 256, 256, B - 64, B - 128} - {0, B})
        for c in cuts:
            if not 0 < c < B: continue
            parts = [packed(0, c), packed(c, B)]
            outs = [torch.empty((c, 2), device=dev), torch.empty((B - c, 2), device=dev)]
            main = torch.cuda.current_stream(dev)
            def run_split(steps):
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(steps):
                    ev = torch.cuda.Event(); ev.record(main)
                    for j in range(2):
                        streams[j].wait_event(ev)
                        with torch.cuda.stream(streams[j]):
                            ms[j].classify_raw(*parts[j], out=outs[j])
                        e2 = torch.cuda.Event(); e2.record(streams[j]); main.wait_event(e2)
                torch.cuda.synchronize(); return (time.perf_counter() - t) / steps
            run_split(10); ts = run_split(50)
            same = torch.equal(torch.cat(outs), out_w)
            line += " | %d+%d %.4f (x%.3f%s)" % (c, B - c, ts * 1e3, tw / ts, "" if same else " DIFF")
        print(line, flush=True)
    for m in ms: m.close()
