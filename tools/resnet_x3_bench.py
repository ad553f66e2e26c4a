#!/usr/bin/env python3
"""ResNet basic blocks: fp32 (f32-input MFMA) against split precision on the bf16 MFMA (rs_seqnet_set_mode), 512 x 16000,
interleaved in one process:  python tools/resnet_x3_bench.py [B] [L]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.resnet import ResNetModel

B, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 512), (int(sys.argv[2]) if len(sys.argv) > 2 else 16000)
dev = torch.device("cuda", 0)
cfg = synth.RESNET_BENCH_CFG
if os.environ.get("RS_RESNET") == "bottleneck":       # the wide bottleneck net of tools/resnet_bottleneck_probe.py
    cfg = dict(channels=[32, 48, 68], kernel=19, padding=5, stride=3, block="bottleneck", n_layers=3, blocks=[2, 2, 1], n_classes=2)
sd = synth.make_resnet_state_dict(7, cfg)
x = torch.from_numpy(np.stack([np.clip((s.astype(np.float32) - 500.0) / 60.0, -3.5, 3.5)
                               for s in synth.make_signals(20260103, 64, L)])).to(dev).repeat(B // 64, 1).contiguous()
ms = {dt: ResNetModel(sd, types.SimpleNamespace(resnet=types.SimpleNamespace(**cfg)), None, "x", device=dev, dtype=dt)
      for dt in ("f32", "bf16x3")}
res = {dt: [] for dt in ms}
out = {}
for rnd in range(3):
    for dt, m in ms.items():
        for _ in range(3):
            p = m._net.forward(x)
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(10):
            p = m._net.forward(x)
        torch.cuda.synchronize(dev)
        res[dt].append((time.perf_counter() - t) / 10 * 1e3)
        out[dt] = p.cpu().numpy()
for dt in ms:
    print(f"{dt:7s} ms per {B} x {L} step: {[round(v, 4) for v in res[dt]]}  median {np.median(res[dt]):.4f}  {B / np.median(res[dt]) * 1e3:.0f} chunks/s")
print("max |dp| bf16x3 vs f32: %.2e; labels differing at 0.9: %d" % (np.abs(out["bf16x3"] - out["f32"]).max(),
      int(((out["bf16x3"][:, 1] > 0.9) != (out["f32"][:, 1] > 0.9)).sum())))
