#!/usr/bin/env python3
"""Random batch shapes through the thin-launch forms against the round-4 forms (big eight-wave tiles forced on every tiled layer,
small kernel off, runs of eight): logits and probabilities must be the same bits.
    python tools/thin_fuzz.py [--cases 60] [--seed 1]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from riser_amd import synth
from riser_amd.preprocess import pack_reads
from conftest import hooked_model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    sd = synth.make_state_dict(1)
    big4 = ";".join("%d:8,1,1,4" % i for i in range(2, 12))
    big2 = ";".join("%d:8,1,2,2" % i for i in range(2, 12))
    new = hooked_model({}, sd, "f32w", dev)
    old = hooked_model({"RS_SMALL_F32_WAVES": "0", "RS_FORCE_SHAPE_WINO4": big4, "RS_FORCE_SHAPE_WINO": big2, "RS_SF32_MIN_RUN": "8"}, sd, "f32w", dev)
    small = hooked_model({"RS_SMALL_F32_WAVES": "100000000"}, sd, "f32w", dev)
    rng = np.random.default_rng(args.seed)
    pool = synth.make_signals(20260103, 96, 16000, first_read=70000)
    bad = 0
    for k in range(args.cases):
        B = int(rng.choice([1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233]) + rng.integers(0, 3))
        mode = rng.integers(0, 3)
        lens = (np.full(B, int(rng.integers(4096, 16001))) if mode == 0 else rng.integers(4096, 16001, size=B) if mode == 1
                else rng.choice([4096, 6024, 8615, 12048, 16000], size=B))
        sigs = [pool[(k * 7 + i) % len(pool)][: int(n)] for i, n in enumerate(lens)]
        sig, off, ln, lh = pack_reads(sigs, dev)
        a = new.classify_raw(sig, off, ln, lh, return_logits=True)
        b = old.classify_raw(sig, off, ln, lh, return_logits=True)
        c = small.classify_raw(sig, off, ln, lh, return_logits=True)
        ok = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(c[0], b[0])
        bad += not ok
        print(f"case {k:3d}: B={B:4d} lengths {['equal', 'random', 'set'][mode]:6s} tiles {[(i['bm'], i['bn']) for i in new.layer_info()[8:12]]} {'ok' if ok else 'DIFFERENT'}", flush=True)
    print(f"{args.cases} cases, {bad} differing")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
