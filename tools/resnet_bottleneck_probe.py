import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.resnet import ResNetModel, build_program, program_traffic_bytes
dev = torch.device("cuda", 0)
cfg = dict(channels=[32, 48, 68], kernel=19, padding=5, stride=3, block="bottleneck", n_layers=3, blocks=[2, 2, 1], n_classes=2)
sd = synth.make_resnet_state_dict(7, cfg)
x = torch.from_numpy(np.stack([np.clip((s.astype(np.float32) - 500.0) / 60.0, -3.5, 3.5) for s in synth.make_signals(20260103, 64, 16000)])).to(dev).repeat(8, 1).contiguous()
for nofuse in (0, 1):
    if nofuse: os.environ["RS_SEQ_NOFUSE"] = "1"
    m = ResNetModel(sd, types.SimpleNamespace(resnet=types.SimpleNamespace(**cfg)), None, "x", device=dev)
    os.environ.pop("RS_SEQ_NOFUSE", None)
    for _ in range(5): p = m._net.forward(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): p = m._net.forward(x)
    torch.cuda.synchronize(); print("nofuse" if nofuse else "fused", round((time.perf_counter() - t) / 20 * 1e3, 4), "ms", p[0].cpu().numpy())
    m.close()
