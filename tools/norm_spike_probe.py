import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
dev = torch.device("cuda", 0)
m = Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype="f16", device=dev)
base = synth.make_signals(20260103, 512, 16000)
print("range of the bench reads: median", np.median(base.max(1).astype(int) - base.min(1)), "max", (base.max(1).astype(int) - base.min(1)).max())
for name, frac in (("no spike", 0.0), ("1 read in 16 has one 9000-count sample", 1 / 16), ("every read has one", 1.0)):
    sigs = base.copy()
    for b in range(512):
        if frac and (b % int(round(1 / frac))) == 0:
            sigs[b, 1000 + b] = 9000
    sig, off, ln, lh = pack_reads(list(sigs), dev)
    for _ in range(3): m.classify_raw(sig, off, ln, lh)
    m.profile(True)
    for _ in range(10): m.classify_raw(sig, off, ln, lh)
    ms, calls = m.profile_read(); m.profile(False)
    print(f"{name}: norm {(ms / calls)[0] * 1e3:.1f} us")
