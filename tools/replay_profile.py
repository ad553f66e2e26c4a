#!/usr/bin/env python3
"""cProfile of the batched control loop on scripted ReadUntil batches (where the host time of a batch goes):
    python tools/replay_profile.py [--dtype f32w] [--models 1]"""
import argparse, cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from riser_amd import synth
from riser_amd.replay import run_replay, scripted_batches

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f32w")
ap.add_argument("--models", type=int, default=1)
args = ap.parse_args()
from riser_amd import Model, SignalProcessor, Kit
dev = torch.device("cuda", 0)
models = [Model(synth.make_state_dict(s), synth.Config(), None, t, dtype=args.dtype, device=dev)
          for s, t in list(zip((1, 2, 3), ("mRNA", "mtRNA", "globin")))[: args.models]]
proc = SignalProcessor(Kit.create_from_version("RNA004"), device=dev)
run_replay(models, proc, scripted_batches(8, 512), mode="enrich")          # warm-up
batches = scripted_batches(40, 512)
pr = cProfile.Profile()
pr.enable()
res = run_replay(models, proc, batches, mode="enrich")
pr.disable()
print({k: res[k] for k in ("p50_ms", "p99_ms", "assessed_per_batch")})
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
