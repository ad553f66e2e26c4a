#!/usr/bin/env python3
"""BASELINE config 4 on one GPU: 18 000 host-resident 4 s chunks streamed through the GPU in
sub-batches (PCIe upload overlapped with compute).  Prints END-TO-END chunks/s from host memory - the
PCIe-inclusive figure DESIGN.md quotes beside `python bench.py --config promethion` (HBM-resident)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.stream import StreamClassifier

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=18000)
ap.add_argument("--len", type=int, default=16000)
ap.add_argument("--sub-batch", type=int, default=1024)
ap.add_argument("--dtype", default="f32w")
ap.add_argument("--pinned", action="store_true")
ap.add_argument("--device", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda", a.device)
base = synth.make_signals(20260103, 256, a.len)
sig = np.tile(base, ((a.reads + 255) // 256, 1))[: a.reads]
if a.pinned:
    sig = torch.from_numpy(sig).pin_memory()
m = Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype=a.dtype, device=dev)
sc = StreamClassifier([m], sub_batch=a.sub_batch, max_len=a.len)
sc.classify(sig[: 2 * a.sub_batch])                       # warm-up
t = time.perf_counter()
p = sc.classify(sig)
dt = time.perf_counter() - t
ref = m.classify_raw(torch.from_numpy(base[:8].reshape(-1)).to(dev), torch.arange(8, device=dev) * a.len,
                     torch.full((8,), a.len, dtype=torch.int32, device=dev), np.full(8, a.len, np.int32)).cpu().numpy()
assert np.array_equal(p[0, :8], ref), "streamed result differs from the direct call"
print(json.dumps({"reads": a.reads, "samples": a.len, "dtype": a.dtype, "sub_batch": a.sub_batch, "pinned_input": a.pinned,
                  "seconds": round(dt, 4), "chunks_per_s_end_to_end": round(a.reads / dt, 1),
                  "h2d_GBps": round(a.reads * a.len * 2 / dt / 1e9, 2)}))
