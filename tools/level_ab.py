#!/usr/bin/env python3
"""Two-level packed layout against the one-level layout (RS_ONE_LEVEL=1), same box, interleaved: ms per batch.
    python tools/level_ab.py [f32w bf16x3 f16x3 f16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
dev = torch.device("cuda", 0)


def model(dt, one):
    if one:
        os.environ["RS_ONE_LEVEL"] = "1"
    try:
        return Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype=dt, device=dev)
    finally:
        os.environ.pop("RS_ONE_LEVEL", None)


rng = np.random.default_rng(3)
mixed = np.array([(8000, 12000, 16000)[i % 3] for i in range(512)])
cases = {"512 x 16000": np.full(512, 16000), "512 mixed 2/3/4 s": mixed, "357 x 8615": np.full(357, 8615),
         "357 x uniform[4096, 8615]": rng.integers(4096, 8616, size=357), "2048 x 8615": np.full(2048, 8615)}
for dt in sys.argv[1:] or ["f32w", "bf16x3", "f16x3", "f16"]:
    ms = {"two levels": model(dt, False), "one level": model(dt, True)}
    for name, lens in cases.items():
        lens = lens.astype(np.int32)
        base = synth.make_signals(20260103, 64, 16000)
        sigs = [base[i % 64][: int(n)] for i, n in enumerate(lens)]
        sig, off, ln, lh = pack_reads(sigs, dev)
        out = torch.empty((len(lens), 2), device=dev)
        res = {k: [] for k in ms}
        for rep in range(3):
            for k, m in ms.items():
                for _ in range(15): m.classify_raw(sig, off, ln, lh, out=out)
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(40): m.classify_raw(sig, off, ln, lh, out=out)
                torch.cuda.synchronize(); res[k].append((time.perf_counter() - t) / 40 * 1e3)
        a, b = min(res["one level"]), min(res["two levels"])
        print(f"{dt} {name}: one level {a:.4f} ms ({len(lens) / a * 1e3:.0f}/s) | two levels {b:.4f} ms ({len(lens) / b * 1e3:.0f}/s): "
              f"{100 * (b / a - 1):+.1f} %", flush=True)
    for m in ms.values(): m.close()
