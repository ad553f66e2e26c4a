#!/usr/bin/env python3
"""A/B of the slice pipeline's host-side knobs on the 18 000-channel replay, variants interleaved in ONE process (box-to-box
and minute-to-minute differences of the host are larger than the effects):
    python tools/control_ab.py [--dtype bf16x3] [--rounds 3]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.control import SequencerControl
from riser_amd.replay import run_replay, scripted_batches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=18000)
    ap.add_argument("--batches", type=int, default=30)
    ap.add_argument("--dtype", default="bf16x3")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", default="", help="comma list of variant indices (0 r4 order, 1 thread, 2 thread+half, 3 half)")
    args = ap.parse_args()
    from riser_amd import Model, SignalProcessor, Kit
    dev = torch.device("cuda", 0)
    models = [Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype=args.dtype, device=dev)]
    proc = SignalProcessor(Kit.create_from_version("RNA004"), device=dev)
    batches = scripted_batches(args.batches, args.channels)
    variants = {"r4 order (one thread, equal slices)": (False, False), "stage thread": (True, False),
                "stage thread + half first slice": (True, True), "half first slice only": (False, True)}
    if args.only:
        keep = {int(v) for v in args.only.split(",")}
        variants = {k: v for i, (k, v) in enumerate(variants.items()) if i in keep}
    res = {k: [] for k in variants}
    for _ in range(args.rounds):
        for name, (thr, half) in variants.items():
            SequencerControl.STAGE_THREAD, SequencerControl.FIRST_SLICE_HALF = thr, half
            r = run_replay(models, proc, batches, mode="enrich")
            res[name].append((r["loop_p50_ms"], r["assessed_per_s"], r["phase_ms_median"]))
    for name, rs in res.items():
        print(f"{args.dtype} {name:42s} loop p50 ms {[x[0] for x in rs]}  median {np.median([x[0] for x in rs]):.2f}  reads/s {np.median([x[1] for x in rs]):.0f}")
        print("      phases of the last round:", rs[-1][2])


if __name__ == "__main__":
    main()
