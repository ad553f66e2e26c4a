#!/usr/bin/env python3
"""Per-phase cycle shares of the ring kernel's sub-stage loop (diagnostic library built with -DRS_RING_STAMPS:
`python tools/ring_stamps.py build`, then run with RISER_AMD_LIB=riser_amd/lib/libriser_amd_ringstamps.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from riser_amd import build as b
    print(b.build(extra_flags=["-DRS_RING_STAMPS"], lib_name="riser_amd_ringstamps"))
    sys.exit(0)
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
dev = torch.device("cuda", 0)
B, L = 512, 16000
sigs = synth.make_signals(20260103, 64, L); sigs = np.tile(sigs, (B // 64, 1))
sig, off, ln, lens = pack_reads(list(sigs), dev)
m = Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype=dt, device=dev)
for _ in range(3):
    m.classify_raw(sig, off, ln, lens)
torch.cuda.synchronize()
