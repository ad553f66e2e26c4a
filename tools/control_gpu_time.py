#!/usr/bin/env python3
"""Device time of the classify call inside the control loop (HIP events around rs_classify_ensemble, per batch) against the
host's `device_wait` phase: how much of that wait is kernels, how much is latency around them.
    python tools/control_gpu_time.py [--dtype f32w] [--channels 512] [--batches 120]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
import riser_amd.control as ctl
from riser_amd.replay import run_replay, scripted_batches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32w")
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--batches", type=int, default=120)
    args = ap.parse_args()
    from riser_amd import Model, SignalProcessor, Kit
    dev = torch.device("cuda", 0)
    models = [Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype=args.dtype, device=dev)]
    proc = SignalProcessor(Kit.create_from_version("RNA004"), device=dev)
    inner = ctl.classify_raw_ensemble
    pairs, shapes = [], []

    def timed(models_, sig, off, ln, lens, **kw):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = inner(models_, sig, off, ln, lens, **kw)
        b.record()
        pairs.append((a, b))
        shapes.append((int(lens.shape[0]), float(lens.mean())))
        return out

    ctl.classify_raw_ensemble = timed
    res = run_replay(models, proc, scripted_batches(args.batches, args.channels), mode="enrich")
    torch.cuda.synchronize()
    gpu = np.array([a.elapsed_time(b) for a, b in pairs])[8:]
    sh = np.array(shapes)[8:]
    print(f"{args.dtype} {args.channels} channels: classify call device time median {np.median(gpu):.3f} ms (p90 {np.percentile(gpu, 90):.3f}) "
          f"for {np.median(sh[:, 0]):.0f} reads of {np.median(sh[:, 1]):.0f} samples; host phases {res['phase_ms_median']}; p50 {res['p50_ms']}")


if __name__ == "__main__":
    main()
