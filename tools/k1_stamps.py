#!/usr/bin/env python3
"""Phase timing of the normalise kernel (workgroup 0, s_memtime at 100 MHz... cycles of the shader clock counter):
    python tools/ablate_build.py normalise.hip k1stamps=RS_K1_STAMPS
    RISER_AMD_LIB=riser_amd/lib/libabl_k1stamps.so python tools/k1_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
dev = torch.device("cuda", 0)
B, L = int(os.environ.get("RS_B", 512)), int(os.environ.get("RS_L", 16000))
sigs = synth.make_signals(20260103, B, L)
sig, off, ln, lens = pack_reads(list(sigs), dev)
m = Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype=sys.argv[1] if len(sys.argv) > 1 else "f16", device=dev)
for _ in range(6):
    m.classify_raw(sig, off, ln, lens)
torch.cuda.synchronize()
