"""Probe: one 512-read batch as two concurrent half batches on two HIP streams (two Model instances = two workspaces)
against the single-stream step.  python tools/two_stream_probe.py [dtype] [parts]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
dt = sys.argv[1] if len(sys.argv) > 1 else "f32w"
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B, L = 512, 16000
dev = torch.device("cuda", 0)
sigs = synth.make_signals(20260103, B, L)
sig, off, ln, lens = pack_reads(list(sigs), dev)
models = [Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype=dt) for _ in range(parts)]
streams = [torch.cuda.Stream(dev) for _ in range(parts)]
probs = torch.empty((B, 2), dtype=torch.float32, device=dev)
cuts = [B * i // parts for i in range(parts + 1)]
offs = [off[cuts[i]:cuts[i + 1]].contiguous() for i in range(parts)]
lns = [ln[cuts[i]:cuts[i + 1]].contiguous() for i in range(parts)]

def single():
    models[0].classify_raw(sig, off, ln, lens, out=probs)

def split():
    for i in range(parts):
        with torch.cuda.stream(streams[i]):
            models[i].classify_raw(sig, offs[i], lns[i], lens[cuts[i]:cuts[i + 1]], out=probs[cuts[i]:cuts[i + 1]])

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

single(); torch.cuda.synchronize(); ref = probs.cpu().numpy().copy()
split(); torch.cuda.synchronize(); print("bit-identical:", np.array_equal(ref, probs.cpu().numpy()))
for k in range(2):
    print("single %.3f ms   %d-way split on %d streams %.3f ms" % (timeit(single), parts, parts, timeit(split)))
