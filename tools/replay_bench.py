#!/usr/bin/env python3
"""Replay scripted 512-channel ReadUntil batches through the batched SequencerControl and report
per-batch latency (host wall time from get_read_batch() to the reject/finish calls) and reads/s.

    python tools/replay_bench.py [--batches 40] [--models 1|3] [--dtype f32w|f16|bf16] [--kit RNA004]
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from riser_amd import synth
from riser_amd.replay import chunked_batches, run_replay, scripted_batches
from riser_amd.fake_client import FakeClient, PlainFakeClient


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=40)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--models", type=int, default=1)
    ap.add_argument("--dtype", default="f32w")
    ap.add_argument("--kit", default="RNA004")
    ap.add_argument("--mode", default="enrich")
    ap.add_argument("--full-reupload", action="store_true", help="no device-resident signals: every read whole, every batch")
    ap.add_argument("--traffic", default="whole", choices=["whole", "chunks"],
                    help="whole: every batch re-sends the read from its start; chunks: a popping client, disjoint 2 s chunks")
    ap.add_argument("--python-loops", action="store_true", help="eight-method duck type only (no C host loops)")
    args = ap.parse_args()
    from riser_amd import Model, SignalProcessor, Kit
    dev = torch.device("cuda", 0)
    models = [Model(synth.make_state_dict(s), synth.Config(), None, t, dtype=args.dtype, device=dev)
              for s, t in list(zip((1, 2, 3), ("mRNA", "mtRNA", "globin")))[: args.models]]
    proc = SignalProcessor(Kit.create_from_version(args.kit), device=dev)
    batches = (chunked_batches if args.traffic == "chunks" else scripted_batches)(args.batches, args.channels)
    res = run_replay(models, proc, batches, mode=args.mode, signal_cache=not args.full_reupload,
                     client_cls=PlainFakeClient if args.python_loops else FakeClient)
    res.update(channels=args.channels, models=args.models, dtype=args.dtype, kit=args.kit, traffic=args.traffic)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
