#!/usr/bin/env python3
"""f16xf8 with the run of F8 rows starting at layer 7 (RS_F8_MIN_CIN=200, the default) against layer 6 (150): interleaved, one box.
    python tools/f8_min_cin_ab.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from riser_amd import synth
from riser_amd.preprocess import pack_reads
from conftest import hooked_model
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device("cuda", 0)
sd = synth.make_state_dict(1)
ms = {"min_cin 200 (layers 7-11)": hooked_model({}, sd, "f16xf8", dev), "min_cin 150 (layers 6-11)": hooked_model({"RS_F8_MIN_CIN": "150"}, sd, "f16xf8", dev)}
for B, L, mixed in ((512, 16000, False), (512, 16000, True), (357, 8615, False), (576, 16000, False)):
    sig, off, ln, lens = pack_reads(list(synth.make_signals(20260103, B, L)), dev)
    if mixed:
        lens = np.array([(L // 2, 3 * L // 4, L)[i % 3] for i in range(B)], dtype=np.int32)
        ln = torch.from_numpy(lens).to(dev)
    out = torch.empty((B, 2), device=dev)
    res = {k: [] for k in ms}
    for m in ms.values():
        for _ in range(20): m.classify_raw(sig, off, ln, lens, out=out)
    for r in range(rounds):
        for k, m in ms.items():
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(100): m.classify_raw(sig, off, ln, lens, out=out)
            torch.cuda.synchronize(); res[k].append((time.perf_counter() - t) / 100 * 1e3)
    for k, v in res.items():
        print("B=%d L=%d %s %-28s median %.4f ms (min %.4f)" % (B, L, "mixed" if mixed else "full ", k, sorted(v)[len(v) // 2], min(v)), flush=True)
