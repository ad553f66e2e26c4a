"""Per-layer conv timings (HIP events) for one or more dtypes: python tools/layer_times.py f32 f32w"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
dts = sys.argv[1:] or ["f32"]
B, L = int(os.environ.get("RS_B", 512)), int(os.environ.get("RS_L", 16000))
sigs = synth.make_signals(20260103, B, L)
dev = torch.device("cuda", 0)
sig, off, ln, lens = pack_reads(list(sigs), dev)
if os.environ.get("RS_MIXED"):                       # 2 s / 3 s / 4 s thirds (BASELINE config 5)
    lens = np.array([(L // 2, 3 * L // 4, L)[i % 3] for i in range(B)], dtype=np.int32)
    ln = torch.from_numpy(lens).to(dev)
flops = [0.0] * len(synth.CHANNELS)
for n in lens:
    c_in, Li = 1, int(n)
    for i, c in enumerate(synth.CHANNELS):
        flops[i] += 2.0 * c_in * c * 3 * Li
        c_in, Li = c, Li // 2
ref = None
for dt in dts:
    m = Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype=dt)
    for _ in range(3): p = m.classify_raw(sig, off, ln, lens)
    p = p.cpu().numpy()
    if ref is None: ref = p
    m.profile(True)
    for _ in range(10): m.classify_raw(sig, off, ln, lens)
    ms, calls = m.profile_read()
    ms = ms / calls
    info = m.layer_info()
    conv = ms[2:13].sum()
    print("%s total %.3f ms  conv1-11 %.3f ms = %.1f TF  norm %.3f conv0 %.3f head %.3f  max|dp vs first| %.2e" % (
        dt, ms.sum(), conv, sum(flops[1:]) / conv / 1e9, ms[0], ms[1], ms[13], np.abs(p - ref).max()))
    print("   " + " ".join("L%d[%dx%d k%d]=%.3f(%.0fTF)" % (i, info[i]["bm"], info[i]["bn"], info[i]["kc"], ms[1 + i], flops[i] / ms[1 + i] / 1e9) for i in range(1, 12)))
    m.close()
