"""A/B two builds of the library in one process launch sequence (alternating), per-step time of dtype $RS_DT (default f32) at
$RS_B x $RS_L (default 512 x 16000; RS_MIXED=1: lengths L/2, 3L/4, L by read index)."""
import sys, os, subprocess, json
libs = sys.argv[1:]
code = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
B, L = int(os.environ.get("RS_B", 512)), int(os.environ.get("RS_L", 16000))
sigs = synth.make_signals(20260103, B, L)
dev = torch.device("cuda", 0)
sig, off, ln, lens = pack_reads(list(sigs), dev)
if os.environ.get("RS_MIXED"):                       # 2 s / 3 s / 4 s thirds (BASELINE config 5)
    lens = np.array([(L // 2, 3 * L // 4, L)[i % 3] for i in range(B)], dtype=np.int32)
    ln = torch.from_numpy(lens).to(dev)
m = Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype=os.environ.get("RS_DT", "f32"))
out = torch.empty((B, 2), device=dev)
for _ in range(5): m.classify_raw(sig, off, ln, lens, out=out)
torch.cuda.synchronize(); t = time.perf_counter()
N = 30 if B >= 256 else 200
for _ in range(N): m.classify_raw(sig, off, ln, lens, out=out)
torch.cuda.synchronize(); print("%.4f" % ((time.perf_counter() - t) / N * 1e3))
'''
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ); env["RISER_AMD_LIB"] = os.path.abspath(l)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        res[l].append(float(out.stdout.strip().split("\n")[-1]))
for l in libs: print(l, res[l], "median %.4f" % sorted(res[l])[1])
