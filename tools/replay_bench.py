#!/usr/bin/env python3
"""Replay scripted 512-channel ReadUntil batches through the batched SequencerControl and report
per-batch latency (host wall time from get_read_batch() to the reject/finish calls) and reads/s.

    python tools/replay_bench.py [--batches 40] [--models 1|3] [--dtype f32|f16|bf16] [--kit RNA004]
"""
import argparse, json, logging, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from riser_amd import synth
from riser_amd.fake_client import FakeClient, FakeRead


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=40)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--models", type=int, default=1)
    ap.add_argument("--dtype", default="f32w")
    ap.add_argument("--kit", default="RNA004")
    ap.add_argument("--mode", default="enrich")
    args = ap.parse_args()
    from riser_amd import Model, SignalProcessor, Kit, SequencerControl
    rng = np.random.default_rng(7)
    # a pool of raw reads at varied lengths (AccumulatingCache: a read is re-seen, longer, until decided)
    pool = []
    for rid in range(args.channels * 2):
        n = int(rng.integers(5000, 22000))
        pool.append(synth.make_raw_read(4242, rid, n, polya=(rid % 5 != 0)))
    batches = []
    for b in range(args.batches):
        reads = []
        for ch in range(args.channels):
            rid = (b * 37 + ch) % len(pool)
            reads.append((ch + 1, FakeRead(f"read-{b // 4}-{rid}", pool[rid])))
        batches.append(reads)
    dev = torch.device("cuda", 0)
    models = [Model(synth.make_state_dict(s), synth.Config(), None, t, dtype=args.dtype, device=dev)
              for s, t in list(zip((1, 2, 3), ("mRNA", "mtRNA", "globin")))[: args.models]]
    proc = SignalProcessor(Kit.create_from_version(args.kit), device=dev)
    client = FakeClient(batches)
    with tempfile.TemporaryDirectory() as d:
        ctl = SequencerControl(client, models, proc, logging.getLogger("replay"), os.path.join(d, "out"))
        ctl.start()
        t0 = time.perf_counter()
        ctl.target(args.mode, 1.0, 0.9)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ctl.finish()
        rows = sum(1 for _ in open(os.path.join(d, "out.csv"))) - 1
    lat = np.asarray(ctl.batch_latencies[3:]) * 1e3
    print(json.dumps({"batches": args.batches, "channels": args.channels, "models": args.models, "dtype": args.dtype,
                      "reads_assessed": rows, "reads_received": args.batches * args.channels,
                      "wall_s": round(wall, 3), "assessed_per_s": round(rows / wall, 1),
                      "batch_latency_ms": {"p50": round(float(np.percentile(lat, 50)), 2),
                                            "p99": round(float(np.percentile(lat, 99)), 2),
                                            "max": round(float(lat.max()), 2)},
                      "decisions": {"rejected": sum(len(r) for r in client.rejected),
                                    "finished": sum(len(r) for r in client.finished)}}))


if __name__ == "__main__":
    main()
