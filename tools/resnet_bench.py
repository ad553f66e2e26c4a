#!/usr/bin/env python3
"""ResNet (riser/nets/resnet.py) on the generic conv program: step time of a 512 x 16000 batch with the MFMA conv kernel and
with the scalar kernel of round 1 (RS_SEQ_SCALAR=1), agreement of the two, and the conv FLOP rate against the f32 MFMA peak.
    python tools/resnet_bench.py [B] [L]"""
import json, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.resnet import ResNetModel, build_program, program_flops


def resnet_flops(cfg, sd, L):
    return program_flops(build_program(sd, types.SimpleNamespace(**cfg))[0], L)


def run(B=512, L=16000, steps=10):
    dev = torch.device("cuda", 0)
    cfg = synth.RESNET_BENCH_CFG
    sd = synth.make_resnet_state_dict(7)
    x = torch.from_numpy(np.stack([np.clip((s.astype(np.float32) - 500.0) / 60.0, -3.5, 3.5)
                                   for s in synth.make_signals(20260103, 64, L)])).to(dev)
    x = x.repeat(B // 64, 1).contiguous()
    out = {}
    probs = {}
    for mode in ("mfma", "scalar"):
        if mode == "scalar":
            os.environ["RS_SEQ_SCALAR"] = "1"
        try:
            m = ResNetModel(sd, types.SimpleNamespace(resnet=types.SimpleNamespace(**cfg)), None, "x", device=dev)
        finally:
            os.environ.pop("RS_SEQ_SCALAR", None)
        for _ in range(3):
            p = m._net.forward(x)
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(steps):
            p = m._net.forward(x)
        torch.cuda.synchronize(dev)
        out[mode + "_ms_per_step"] = round((time.perf_counter() - t) / steps * 1e3, 4)
        probs[mode] = p.cpu().numpy()
        m.close()
    fl = resnet_flops(cfg, sd, L) * B
    out.update(batch=B, chunk=L, config=cfg, speedup=round(out["scalar_ms_per_step"] / out["mfma_ms_per_step"], 1),
               chunks_per_s=round(B / (out["mfma_ms_per_step"] * 1e-3), 1), conv_gflop_per_step=round(fl / 1e9, 2),
               conv_tflops=round(fl / (out["mfma_ms_per_step"] * 1e-3) / 1e12, 2),
               frac_of_f32_mfma_peak=round(fl / (out["mfma_ms_per_step"] * 1e-3) / 1e12 / 157.3, 4),
               max_abs_dprob_mfma_vs_scalar=float(np.abs(probs["mfma"] - probs["scalar"]).max()))
    return out


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
    print(json.dumps(run(B, L)))
