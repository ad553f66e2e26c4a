#!/usr/bin/env python3
"""Split precision with the merged tail panel (RS_X3_TAIL=1: layers whose last 32-channel panel holds <= 8 channels run its three
taps as one K step) against without, interleaved on one box; max |dp| between the two and against fp32.
    python tools/x3_tail_ab.py [dtype ...] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from riser_amd import synth
from riser_amd.preprocess import pack_reads
from conftest import hooked_model
dts = [a for a in sys.argv[1:] if not a.isdigit()] or ["bf16x3", "f16xf8"]
rounds = int(next((a for a in sys.argv[1:] if a.isdigit()), 5))
dev = torch.device("cuda", 0)
sd = synth.make_state_dict(1)
ref = hooked_model({}, sd, "f32w", dev)
for dt in dts:
    ms = {"tail off": hooked_model({"RS_X3_TAIL": "0"}, sd, dt, dev), "tail on ": hooked_model({"RS_X3_TAIL": "1"}, sd, dt, dev)}
    for B, L, mixed in ((512, 16000, False), (512, 16000, True), (357, 8615, False), (576, 16000, False), (16, 16000, False), (1, 16000, False)):
        sig, off, ln, lens = pack_reads(list(synth.make_signals(20260103, B, L)), dev)
        if mixed:
            lens = np.array([(L // 2, 3 * L // 4, L)[i % 3] for i in range(B)], dtype=np.int32)
            ln = torch.from_numpy(lens).to(dev)
        out = torch.empty((B, 2), device=dev)
        res = {k: [] for k in ms}
        probs = {}
        for k, m in ms.items():
            for _ in range(20): m.classify_raw(sig, off, ln, lens, out=out)
            probs[k] = out.clone()
        r32 = ref.classify_raw(sig, off, ln, lens)
        steps = 100 if B >= 64 else 400
        for r in range(rounds):
            for k, m in ms.items():
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(steps): m.classify_raw(sig, off, ln, lens, out=out)
                torch.cuda.synchronize(); res[k].append((time.perf_counter() - t) / steps * 1e3)
        med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        print("%-7s B=%4d L=%5d %s off %.4f ms  on %.4f ms  (%+.1f %%)  max|dp| on-off %.1e, on-fp32 %.1e, off-fp32 %.1e  tiles %s" % (
            dt, B, L, "mixed" if mixed else "full ", med["tail off"], med["tail on "], (med["tail on "] / med["tail off"] - 1) * 100,
            float((probs["tail on "] - probs["tail off"]).abs().max()), float((probs["tail on "] - r32).abs().max()),
            float((probs["tail off"] - r32).abs().max()), [(i["bm"], i["bn"]) for i in ms["tail on "].layer_info()[4:6]]), flush=True)
    for m in ms.values(): m.close()
