#!/usr/bin/env python3
"""cProfile of the batched control loop on scripted traffic (host-side hot spots of a PromethION-sized rank):
    python tools/control_profile.py [--channels 18000] [--batches 24] [--dtype f32w]"""
import argparse, cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from riser_amd import synth
from riser_amd.replay import run_replay, scripted_batches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=18000)
    ap.add_argument("--batches", type=int, default=24)
    ap.add_argument("--dtype", default="f32w")
    ap.add_argument("--top", type=int, default=28)
    args = ap.parse_args()
    from riser_amd import Model, SignalProcessor, Kit
    dev = torch.device("cuda", 0)
    models = [Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype=args.dtype, device=dev)]
    proc = SignalProcessor(Kit.create_from_version("RNA004"), device=dev)
    batches = scripted_batches(args.batches, args.channels)
    pr = cProfile.Profile()
    pr.enable()
    res = run_replay(models, proc, batches, mode="enrich")
    pr.disable()
    print({k: res[k] for k in ("p50_ms", "max_ms", "loop_p50_ms", "assessed_per_s", "phase_ms_median")})
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(args.top)
    print(out.getvalue())


if __name__ == "__main__":
    main()
