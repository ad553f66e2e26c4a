#!/usr/bin/env python3
"""First light of the LDS-DMA ring kernel (conv_ring_h16.hip): every dtype mode against the oracle on a mixed
batch, and per-layer times at B = 512 x 16000.  (Round 2 also compared plain mode bit for bit with the round-1
register-staged kernel; that kernel was removed in round 4.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import riser_oracle as ro
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads

dev = torch.device("cuda", 0)
SEED = 20260103


def model(dtype, env=None, seed=1):
    env = env or {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return Model(synth.make_state_dict(seed), synth.Config(), None, "m", dtype=dtype, device=dev)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


lens = [4096, 16000, 8615, 5000, 12001, 4097, 16383, 9999, 6024, 16000]
sigs = [synth.make_signals(SEED, 1, n, first_read=300 + i)[0] for i, n in enumerate(lens)]
sig, off, ln, lh = pack_reads(sigs, dev)
want = ro.classify_reads(synth.make_state_dict(1), sigs)
for dt in ("f32w", "bf16x3", "f16x3", "f16", "bf16"):
    m = model(dt)
    got = m.classify_raw(sig, off, ln, lh).cpu().numpy()
    print(f"{dt:7s} mixed batch: max |dp| vs oracle {np.abs(got - want).max():.3e}", flush=True)
    m.close()
B, L = 512, 16000
sigs = synth.make_signals(SEED, B, L)
sig, off, ln, lh = pack_reads(list(sigs), dev)
ref = None
for dt, env in (("f32w", {}), ("bf16x3", {}), ("f16x3", {}), ("f16", {}), ("bf16", {})):
    m = model(dt, env)
    for _ in range(10):
        p = m.classify_raw(sig, off, ln, lh)
    m.profile(True)
    for _ in range(10):
        p = m.classify_raw(sig, off, ln, lh)
    ms, calls = m.profile_read()
    m.profile(False)
    p = p.cpu().numpy()
    if ref is None:
        ref = p
    info = m.layer_info()
    per = ms / calls
    print(f"{dt:7s} {env}: step {per.sum():.3f} ms  conv1-11 {per[2:13].sum():.3f} ms  max|dp| vs f32w {np.abs(p - ref).max():.2e} "
          f"flips {int(((p[:, 1] > 0.9) != (ref[:, 1] > 0.9)).sum())}", flush=True)
    print("        " + " ".join(f"L{i}[{info[i]['bm']}x{info[i]['bn']}]={per[1 + i]:.3f}" for i in range(1, 12)), flush=True)
    m.close()
