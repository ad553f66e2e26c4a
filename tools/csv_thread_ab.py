#!/usr/bin/env python3
"""Does the CSV writer thread cost MinION-sized batches their tail latency?  512 channels, rows by the writer thread against rows
inline, alternating in one process, p50 / p99 / max per run:   python tools/csv_thread_ab.py [--rounds 4]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.control import SequencerControl
from riser_amd.replay import run_replay, scripted_batches

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--dtype", default="f32w")
args = ap.parse_args()
from riser_amd import Model, SignalProcessor, Kit
dev = torch.device("cuda", 0)
models = [Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype=args.dtype, device=dev)]
proc = SignalProcessor(Kit.create_from_version("RNA004"), device=dev)
batches = scripted_batches(400, 512)
res = {"thread": [], "inline": []}
for _ in range(args.rounds):
    for name, thr in (("thread", 0), ("inline", 1 << 30)):
        SequencerControl.CSV_THREAD_MIN_READS = thr
        r = run_replay(models, proc, batches)
        res[name].append((r["p50_ms"], r["p99_ms"], r["max_ms"], r["loop_p50_ms"]))
for name, rs in res.items():
    print(name, "p50/p99/max/loop_p50 per run:", rs)
