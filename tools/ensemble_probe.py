#!/usr/bin/env python3
"""rs_classify_ensemble: forwards of the three models concurrent (side streams) against back to back (RS_ENSEMBLE_SERIAL),
same box, interleaved repeats.   python tools/ensemble_probe.py [f32w bf16x3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model, classify_raw_ensemble
from riser_amd.preprocess import pack_reads
dev = torch.device("cuda", 0)
specs = ((1, "mRNA"), (2, "mtRNA"), (3, "globin"))


def build(dt, serial):
    if serial:
        os.environ["RS_ENSEMBLE_SERIAL"] = "1"
    try:
        return [Model(synth.make_state_dict(k), synth.Config(), None, t, dtype=dt, device=dev) for k, t in specs]
    finally:
        os.environ.pop("RS_ENSEMBLE_SERIAL", None)


for dt in sys.argv[1:] or ["f32w", "bf16x3"]:
    par, ser = build(dt, False), build(dt, True)
    for B, L in ((512, 16000), (357, 8615), (128, 8615), (32, 8615)):
        sigs = synth.make_signals(20260103, min(B, 128), L)
        sigs = np.tile(sigs, ((B + len(sigs) - 1) // len(sigs), 1))[:B]
        sig, off, ln, lens = pack_reads(list(sigs), dev)
        out = torch.empty((3, B, 2), device=dev)
        dec = torch.empty(B, dtype=torch.uint8, device=dev)
        res = {"par": [], "ser": []}
        for rep in range(3):
            for name, ms in (("par", par), ("ser", ser)):
                for _ in range(15): classify_raw_ensemble(ms, sig, off, ln, lens, out=out, decision=dec, max_len=L)
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(40): classify_raw_ensemble(ms, sig, off, ln, lens, out=out, decision=dec, max_len=L)
                torch.cuda.synchronize(); res[name].append((time.perf_counter() - t) / 40 * 1e3)
        p, s = min(res["par"]), min(res["ser"])
        print(f"{dt} B={B} L={L}: concurrent {p:.3f} ms ({B / p * 1e3:.0f} reads/s)  serial {s:.3f} ms  ratio {s / p:.3f}", flush=True)
    for m in par + ser: m.close()
