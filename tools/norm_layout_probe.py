#!/usr/bin/env python3
"""Normalise-kernel time (the library's own per-launch HIP events) for reads laid out as the control loop's signal store holds
them - one 32768-sample row per channel, a read starting anywhere in its row - against the same reads packed back to back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
dev = torch.device("cuda", 0)
B = int(os.environ.get("RS_B", 2100))
rng = np.random.default_rng(5)
m = Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype=sys.argv[1] if len(sys.argv) > 1 else "bf16x3", device=dev)
base = synth.make_signals(20260103, 64, 16000)


def run(tag, lens, offs, total):
    buf = torch.zeros(total, dtype=torch.int16, device=dev)
    host = np.zeros(total, dtype=np.int16)
    for b in range(B):
        host[offs[b]: offs[b] + lens[b]] = base[b % 64][: lens[b]]
    buf.copy_(torch.from_numpy(host))
    off_d, len_d = torch.from_numpy(offs).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev)
    for _ in range(3):
        m.classify_raw(buf, off_d, len_d, lens.astype(np.int32))
    m.profile(True)
    for _ in range(10):
        m.classify_raw(buf, off_d, len_d, lens.astype(np.int32))
    ms, calls = m.profile_read()
    m.profile(False)
    per = ms / calls
    print(f"{tag:34s} norm {per[0]*1e3:7.1f} us   step {per.sum():.3f} ms", flush=True)


for name, lens in (("uniform 8615", np.full(B, 8615)), ("ragged 4096..8615", rng.integers(4096, 8616, size=B))):
    lens = lens.astype(np.int64)
    packed = np.zeros(B, dtype=np.int64); packed[1:] = np.cumsum(lens[:-1])
    run(name + " packed", lens, packed, int(lens.sum()) + 16)
    rows = rng.permutation(18000)[:B].astype(np.int64)
    run(name + " store rows, start 0", lens, rows * 32768, 18000 * 32768)
    run(name + " store rows, odd starts", lens, rows * 32768 + rng.integers(0, 3000, size=B) * 2 + 1, 18000 * 32768)
    run(name + " store rows, sorted", lens, np.sort(rows) * 32768, 18000 * 32768)
