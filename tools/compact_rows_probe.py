#!/usr/bin/env python3
"""Price compact rows before building them: TIMING ONLY, results wrong.

The packed layout gives a read len / 4096 + 1 blocks; a live 8615-sample read occupies 3 x 4096 = 12288 samples of rows in
every layer.  A ragged per-layer row base (prefix sum of (len >> i) + 1, rounded to even) would cut the rows of layers 6-10
to what the read's own length needs.  RS_EMU_ROWS runs those layers on the corresponding SHARE of the batch's blocks (the
kernels, tiles and launch planner unchanged), which is what such a layout could at best cost.

    python tools/compact_rows_probe.py [f32w f16x3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
dev = torch.device("cuda", 0)
U = 4096


def shares(L, layers):
    out = {}
    for i in layers:
        have = (L // U + 1) * (U >> i)                       # rows of the layer's INPUT buffer per read, packed blocks
        need = (L >> i) + 1
        need += need & 1                                     # even, so that row / 2 pooling stays a shift
        out[i] = min(1000, int(round(1000.0 * need / have)))
    return out


def model(dt, emu):
    if emu:
        os.environ["RS_EMU_ROWS"] = ";".join(f"{i}:{pm}" for i, pm in emu.items())
    try:
        return Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype=dt, device=dev)
    finally:
        os.environ.pop("RS_EMU_ROWS", None)


for dt in sys.argv[1:] or ["f32w", "f16x3"]:
    for B, L in ((357, 8615), (2048, 8615), (512, 12000)):
        emu = shares(L, range(6, 11))
        sigs = synth.make_signals(20260103, 64, L)
        sigs = np.tile(sigs, ((B + 63) // 64, 1))[:B]
        sig, off, ln, lens = pack_reads(list(sigs), dev)
        out = torch.empty((B, 2), device=dev)
        ms = {"packed": model(dt, None), "compact": model(dt, emu)}
        res = {k: [] for k in ms}
        for rep in range(3):
            for k, m in ms.items():
                for _ in range(15): m.classify_raw(sig, off, ln, lens, out=out)
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(40): m.classify_raw(sig, off, ln, lens, out=out)
                torch.cuda.synchronize(); res[k].append((time.perf_counter() - t) / 40 * 1e3)
        a, b = min(res["packed"]), min(res["compact"])
        print(f"{dt} {B} x {L}: packed {a:.4f} ms ({B / a * 1e3:.0f} reads/s)  compact rows in layers 6-10 {b:.4f} ms "
              f"({B / b * 1e3:.0f} reads/s)  step -{100 * (1 - b / a):.1f} %  shares(permille) {emu}", flush=True)
        for m in ms.values(): m.close()
