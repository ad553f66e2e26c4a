#!/usr/bin/env python3
"""Head + tail launches against single launches over random batch shapes (uniform and mixed lengths), both arithmetic families,
in one process (two models per dtype: one created with RS_NO_TAIL_SPLIT=1), alternating.  Flags every shape where the split is
slower.    python tools/tail_split_check.py [n_shapes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads

dev = torch.device("cuda", 0)
rng = np.random.default_rng(5)
n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 24
sd = synth.make_state_dict(1)
base = synth.make_signals(20260103, 64, 16000)


def make(dt, nosplit):
    if nosplit:
        os.environ["RS_NO_TAIL_SPLIT"] = "1"
    try:
        return Model(sd, synth.Config(), None, "m", dtype=dt, device=dev)
    finally:
        os.environ.pop("RS_NO_TAIL_SPLIT", None)


def timeit(m, args, out, n):
    for _ in range(4):
        m.classify_raw(*args, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        m.classify_raw(*args, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


shapes = []
for _ in range(n_shapes):
    B = int(rng.choice([int(rng.integers(130, 1500)), 357, 576, 640, 704, 900]))
    kind = rng.integers(0, 3)
    if kind == 0:
        lens = np.full(B, int(rng.choice([16000, 8615, 12000, 6024, 12048])), dtype=np.int32)
    elif kind == 1:
        lens = rng.integers(4096, 8616, size=B).astype(np.int32)
    else:
        lens = np.array([(8000, 12000, 16000)[i % 3] for i in range(B)], dtype=np.int32)
    shapes.append(lens)
worst = {}
for dt in ("f32w", "bf16x3"):
    ms, mn = make(dt, False), make(dt, True)
    for lens in shapes:
        B = lens.shape[0]
        sigs = [base[i % 64][: int(n)] for i, n in enumerate(lens)]
        args = pack_reads(sigs, dev)
        out = torch.empty((B, 2), device=dev)
        n = max(5, min(30, 6000 // B))
        a = [timeit(ms, args, out, n), timeit(mn, args, out, n), timeit(ms, args, out, n), timeit(mn, args, out, n)]
        split, single = min(a[0], a[2]), min(a[1], a[3])
        p1 = ms.classify_raw(*args).cpu().numpy()
        p2 = mn.classify_raw(*args).cpu().numpy()
        same = bool(np.array_equal(p1, p2))
        tag = "uniform %d" % lens[0] if (lens == lens[0]).all() else "mixed %d..%d" % (lens.min(), lens.max())
        ratio = split / single
        flag = "  <-- slower" if ratio > 1.015 else ""
        print(f"{dt:7s} B={B:5d} {tag:18s} split {split:7.3f} ms  single {single:7.3f} ms  ratio {ratio:.3f}  bit-identical {same}{flag}", flush=True)
        worst[dt] = max(worst.get(dt, 0.0), ratio)
        assert same
    ms.close(); mn.close()
print("worst ratio (split / single):", worst)
