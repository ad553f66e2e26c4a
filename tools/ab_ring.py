#!/usr/bin/env python3
"""A/B of two builds of the library on the ring-kernel layers (interleaved rounds in ONE process are not possible with
two .so files of the same symbols, so each arm is a child process; arms alternate, 3 rounds):
    python tools/ab_ring.py <lib_a.so> <lib_b.so> [dtype] [env K=V ...]"""
import json, os, subprocess, sys
child = r'''
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
dt = sys.argv[1]
dev = torch.device("cuda", 0)
B, L = 512, 16000
sigs = synth.make_signals(20260103, 64, L); sigs = np.tile(sigs, (B // 64, 1))
sig, off, ln, lens = pack_reads(list(sigs), dev)
m = Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype=dt, device=dev)
for _ in range(20): m.classify_raw(sig, off, ln, lens)
m.profile(True)
for _ in range(20): m.classify_raw(sig, off, ln, lens)
ms, calls = m.profile_read()
print(json.dumps((ms / calls).tolist()))
'''
a, b = sys.argv[1], sys.argv[2]
dt = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
extra = dict(kv.split("=", 1) for kv in sys.argv[4:])
res = {a: [], b: []}
for rnd in range(3):
    for lib in (a, b):
        env = dict(os.environ, RISER_AMD_LIB=os.path.abspath(lib), **extra)
        out = subprocess.run([sys.executable, "-c", child, dt], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("[")]
        if not line:
            print(out.stderr[-2000:]); sys.exit(1)
        res[lib].append(json.loads(line[-1]))
import numpy as np
A, Bm = np.median(np.array(res[a]), axis=0), np.median(np.array(res[b]), axis=0)
print("layer      A        B      B/A")
for i in range(1, 12):
    print(f"L{i:<2d}   {A[1+i]:.4f}  {Bm[1+i]:.4f}  {Bm[1+i]/A[1+i]:.3f}")
print(f"conv   {A[2:13].sum():.4f}  {Bm[2:13].sum():.4f}  {Bm[2:13].sum()/A[2:13].sum():.3f}")
