"""Per-layer conv timings (HIP events) for the default plan or RS_FORCE_SHAPE_F32 overrides."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
B, L = int(os.environ.get("RS_B", 512)), int(os.environ.get("RS_L", 16000))
sigs = synth.make_signals(20260103, B, L)
dev = torch.device("cuda", 0)
sig, off, ln, lens = pack_reads(list(sigs), dev)
m = Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype=dt)
for _ in range(3): m.classify_raw(sig, off, ln, lens)
m.profile(True)
for _ in range(10): m.classify_raw(sig, off, ln, lens)
ms, calls = m.profile_read()
info = m.layer_info()
print(os.environ.get("RS_FORCE_SHAPE_F32", "default"), "total %.3f" % (ms.sum() / calls))
print("  ", " ".join("L%d[%dx%d]=%.3f" % (i, info[i]["bm"], info[i]["bn"], ms[1 + i] / calls) for i in range(1, 12)))
