"""First-light check on the GPU box: normalise + forward vs the oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import SignalProcessor, Kit
from oracle import riser_oracle as ro

proc = SignalProcessor(Kit.create_from_version("RNA004"))
g = np.load("tests/golden/normalise.npz")
bad = 0
for name in [str(n) for n in g["names"]]:
    sig, want = g[f"{name}.sig"], g[f"{name}.out"]
    if len(sig) < 2: continue
    got = proc.mad_normalise(sig)
    ok = got.dtype == want.dtype and np.array_equal(got, want)
    if not ok:
        bad += 1
        d = np.flatnonzero(got != want)
        print("NORMALISE MISMATCH", name, got.dtype, want.dtype, d[:5], got[d[:3]], want[d[:3]])
print("normalise cases bad:", bad)

m = Model(synth.make_state_dict(1), synth.Config(), None, "mRNA")
print("layers", m.layer_info())
for L, B in ((4096, 3), (6024, 3), (16000, 4)):
    sigs = synth.make_signals(20260103, B, L, first_read=100)
    xs = [ro.mad_normalise(s) for s in sigs]
    want = np.stack([ro.classify(synth.make_state_dict(1), x) for x in xs])
    got = m.classify_batch(xs).cpu().numpy()
    print(L, "max |dp|", np.abs(got - want).max(), got[:, 1], want[:, 1])
    one = m.classify(xs[0]).cpu().numpy()
    print("   single", one, want[0])
# fused path + timing
B, L = 512, 16000
sigs = synth.make_signals(20260103, B, L)
from riser_amd.preprocess import pack_reads
sig, off, ln, lens = pack_reads(list(sigs), m.device)
p = m.classify_raw(sig, off, ln, lens)
torch.cuda.synchronize()
t = time.time()
for _ in range(5):
    p = m.classify_raw(sig, off, ln, lens)
torch.cuda.synchronize()
dt = (time.time() - t) / 5
print("B=512 L=16000: %.3f ms/batch, %.0f chunks/s" % (dt * 1e3, B / dt))
want = ro.classify_reads(synth.make_state_dict(1), sigs[:4])
print("fused vs oracle", np.abs(p[:4].cpu().numpy() - want).max())
