"""bf16 / f16 first light: parity vs the fp32 oracle and timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
from oracle import riser_oracle as ro

sd = synth.make_state_dict(1)
B, L = 512, 16000
sigs = synth.make_signals(20260103, B, L)
dev = torch.device("cuda", 0)
sig, off, ln, lens = pack_reads(list(sigs), dev)
ref = None
for dt in ("f32", "f16", "bf16"):
    m = Model(sd, synth.Config(), None, "mRNA", dtype=dt)
    p = m.classify_raw(sig, off, ln, lens)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(10):
        p = m.classify_raw(sig, off, ln, lens)
    torch.cuda.synchronize()
    dt_s = (time.time() - t) / 10
    p = p.cpu().numpy()
    if ref is None:
        ref = p
        want = ro.classify_reads(sd, sigs[:6])
        print("f32 vs oracle", np.abs(p[:6] - want).max())
    d = np.abs(p - ref)
    flips = int(((p[:, 1] > 0.9) != (ref[:, 1] > 0.9)).sum())
    m.profile(True)
    for _ in range(5):
        m.classify_raw(sig, off, ln, lens)
    ms, calls = m.profile_read()
    print(f"{dt}: {dt_s*1e3:.3f} ms/batch {B/dt_s:.0f} chunks/s | vs f32: max|dp| {d.max():.2e} mean {d.mean():.2e} label flips {flips}/{B} nan {np.isnan(p).sum()}")
    print("   stage ms:", np.round(ms / calls, 3), [ (li['bm'], li['bn']) for li in m.layer_info()][1:])
