#!/usr/bin/env python3
"""Random batch shapes (1 ... 235 reads, --big: 250 ... 800; equal / random / set lengths) through the 16-bit modes' thin-launch forms (the
thin-launch kernel and the ring kernel's 64-row shapes, as the planner picks them) against the ring kernel alone with one big
shape forced on every layer: the same bits; and against the fp32 path: within 1e-3, labels at 0.9 the same.
    python tools/thin_fuzz_h16.py [--cases 40] [--seed 1] [dtype ...]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from riser_amd import synth
from riser_amd.preprocess import pack_reads
from conftest import hooked_model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="batches of 250 ... 800 reads (the 192- / 320- / 384-row shapes) instead of 1 ... 235")
    ap.add_argument("dtypes", nargs="*", default=["bf16x3", "f16x3", "f16xf8"])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    sd = synth.make_state_dict(1)
    ref = hooked_model({}, sd, "f32w", dev)
    pool = synth.make_signals(20260103, 96, 16000, first_read=70000)
    bad = 0
    for dt in args.dtypes:
        new = hooked_model({}, sd, dt, dev)
        # one 256 x 128 tile shape on every ring layer, no thin-launch kernel (the 8-bit kernel keeps its own shapes)
        old = hooked_model({"RS_THIN_H16_ROWS": "0", "RS_FORCE_SHAPE_RING": ";".join("%d:4,2,4,4" % i for i in range(3, 12))}, sd, dt, dev)
        rng = np.random.default_rng(args.seed)
        worst = 0.0
        for k in range(args.cases):
            B = int(rng.integers(250, 801)) if args.big else int(rng.choice([1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233]) + rng.integers(0, 3))
            mode = rng.integers(0, 3)
            lens = (np.full(B, int(rng.integers(4096, 16001))) if mode == 0 else rng.integers(4096, 16001, size=B) if mode == 1
                    else rng.choice([4096, 6024, 8615, 12048, 16000], size=B))
            sigs = [pool[(k * 7 + i) % len(pool)][: int(n)] for i, n in enumerate(lens)]
            sig, off, ln, lh = pack_reads(sigs, dev)
            a = new.classify_raw(sig, off, ln, lh, return_logits=True)
            b = old.classify_raw(sig, off, ln, lh, return_logits=True)
            r = ref.classify_raw(sig, off, ln, lh)
            same = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
            dp = float((a[0] - r).abs().max())
            flips = int(((a[0][:, 1] > 0.9) != (r[:, 1] > 0.9)).sum())
            worst = max(worst, dp)
            ok = same and dp < 1e-3 and flips == 0 and not new.saturated()
            bad += not ok
            print(f"{dt:7s} case {k:3d}: B={B:4d} lengths {['equal', 'random', 'set'][mode]:6s} tiles "
                  f"{[(i['bm'], i['bn']) for i in new.layer_info()[3:12]]} max|dp vs fp32| {dp:.1e} {'ok' if ok else 'BAD (same bits: %s, flips %d)' % (same, flips)}", flush=True)
        print(f"{dt}: {args.cases} cases, worst |dp| vs fp32 {worst:.2e}")
        new.close(); old.close()
    print(f"{bad} bad cases")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
