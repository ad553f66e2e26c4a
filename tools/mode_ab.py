"""Interleaved A/B of arithmetic modes on one box: python tools/mode_ab.py bf16x3 f16x3 f16xf8 [rounds]  (512 x 16000, RS_MIXED=1: 2 / 3 / 4 s thirds)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
dts = [a for a in sys.argv[1:] if not a.isdigit()] or ["bf16x3", "f16x3", "f16xf8"]
rounds = int(next((a for a in sys.argv[1:] if a.isdigit()), 7))
B, L = int(os.environ.get("RS_B", 512)), int(os.environ.get("RS_L", 16000))
dev = torch.device("cuda", 0)
sig, off, ln, lens = pack_reads(list(synth.make_signals(20260103, B, L)), dev)
if os.environ.get("RS_MIXED"):
    lens = np.array([(L // 2, 3 * L // 4, L)[i % 3] for i in range(B)], dtype=np.int32)
    ln = torch.from_numpy(lens).to(dev)
models = {dt: Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype=dt, device=dev) for dt in dts}
out = torch.empty((B, 2), device=dev)
res = {dt: [] for dt in dts}
for dt, m in models.items():
    for _ in range(30): m.classify_raw(sig, off, ln, lens, out=out)
torch.cuda.synchronize()
for r in range(rounds):
    for dt, m in models.items():
        for _ in range(10): m.classify_raw(sig, off, ln, lens, out=out)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(100): m.classify_raw(sig, off, ln, lens, out=out)
        torch.cuda.synchronize(); res[dt].append((time.perf_counter() - t) / 100 * 1e3)
for dt in dts:
    v = np.array(res[dt])
    print("%-8s ms/step median %.4f (min %.4f max %.4f)  %.0f chunks/s" % (dt, np.median(v), v.min(), v.max(), B / np.median(v) * 1e3))
