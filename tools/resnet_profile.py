#!/usr/bin/env python3
"""ResNet bench architecture, a few steps at 512 x 16000 (for rocprofv3 --kernel-trace --stats)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.resnet import ResNetModel
dev = torch.device("cuda", 0)
cfg = synth.RESNET_BENCH_CFG
m = ResNetModel(synth.make_resnet_state_dict(7), types.SimpleNamespace(resnet=types.SimpleNamespace(**cfg)), None, "x", device=dev)
x = torch.from_numpy(np.stack([np.clip((s.astype(np.float32) - 500.0) / 60.0, -3.5, 3.5)
                               for s in synth.make_signals(20260103, 64, 16000)])).to(dev).repeat(8, 1).contiguous()
for _ in range(12):
    m._net.forward(x)
torch.cuda.synchronize()
