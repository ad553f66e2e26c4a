"""Wall time of Model.classify (one normalised 16000-sample read per call, the reference's own call shape riser/model.py:22-28):
calls queued back to back, calls with the probability read back each time, and a cProfile of the host side."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from oracle import riser_oracle as ro
dev = torch.device("cuda", 0)
m = Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype="f32w", device=dev)
sig = synth.make_signals(20260103, 4, 16000)
xs = [ro.mad_normalise(s).astype(np.float32) for s in sig]
for _ in range(20): m.classify(xs[0])
torch.cuda.synchronize()
import cProfile, pstats, io
N = 2000
t = time.perf_counter()
for i in range(N): p = m.classify(xs[i & 3])
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / N
print("Model.classify wall per call: %.1f us" % (dt * 1e6), type(p))
t = time.perf_counter()
for i in range(N): q = float(m.classify(xs[i & 3])[1])
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / N
print("with the probability read back per call: %.1f us" % (dt * 1e6))
pr = cProfile.Profile(); pr.enable()
for i in range(500): p = m.classify(xs[i & 3])
torch.cuda.synchronize(); pr.disable()
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(14); print(out.getvalue()[:2500])
