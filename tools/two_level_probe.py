#!/usr/bin/env python3
"""Price a TWO-LEVEL packed layout before building it (TIMING ONLY, results wrong): layers 1..8 on blocks of 1024 samples
(a read of 8615 samples occupies 9 x 1024 = 9216 instead of 3 x 4096 = 12288), layers 9..11 on today's 4096-sample blocks
behind a re-pack of layer 8's small output.  RS_EMU_ROWS runs layers 1-8 on the share of blocks the fine layout would use.
    python tools/two_level_probe.py [f32w f16x3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
dev = torch.device("cuda", 0)


def model(dt, emu):
    if emu:
        os.environ["RS_EMU_ROWS"] = ";".join(f"{i}:{pm}" for i, pm in emu.items())
    try:
        return Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype=dt, device=dev)
    finally:
        os.environ.pop("RS_EMU_ROWS", None)


rng = np.random.default_rng(3)
cases = {"357 x 8615": np.full(357, 8615), "2048 x 8615": np.full(2048, 8615),
         "357 x uniform[4096, 8615]": rng.integers(4096, 8616, size=357), "512 x 16000": np.full(512, 16000)}
for dt in sys.argv[1:] or ["f32w", "f16x3"]:
    for name, lens in cases.items():
        lens = lens.astype(np.int32)
        coarse = int((lens // 4096 + 1).sum()) * 4096
        fine = int((lens // 1024 + 1).sum()) * 1024
        pm = min(1000, int(round(1000.0 * fine / coarse)))
        emu = {i: pm for i in range(1, 9)}
        base = synth.make_signals(20260103, 64, 16000)
        sigs = [base[i % 64][: int(n)] for i, n in enumerate(lens)]
        sig, off, ln, lh = pack_reads(sigs, dev)
        out = torch.empty((len(lens), 2), device=dev)
        ms = {"one level": model(dt, None), "two levels": model(dt, emu)}
        res = {k: [] for k in ms}
        for rep in range(3):
            for k, m in ms.items():
                for _ in range(15): m.classify_raw(sig, off, ln, lh, out=out)
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(40): m.classify_raw(sig, off, ln, lh, out=out)
                torch.cuda.synchronize(); res[k].append((time.perf_counter() - t) / 40 * 1e3)
        a, b = min(res["one level"]), min(res["two levels"])
        print(f"{dt} {name}: 4096-sample blocks {a:.4f} ms ({len(lens) / a * 1e3:.0f} reads/s) | layers 1-8 on {pm} permille of the "
              f"blocks {b:.4f} ms ({len(lens) / b * 1e3:.0f} reads/s): step -{100 * (1 - b / a):.1f} %", flush=True)
        for m in ms.values(): m.close()
