#!/usr/bin/env python3
"""Split precision with TWO MFMAs per product instead of three (measurement builds of conv_ring_h16.hip, layers 3-11):
    python tools/ablate_build.py conv_ring_h16.hip x2a=RS_X2_DROP=1 x2w=RS_X2_DROP=2
    RISER_AMD_LIB=riser_amd/lib/libabl_x2a.so python tools/x2_probe.py
max |dp| and label flips at 0.9 against the reference's golden probabilities (14 cases) and against the fp32 path on the
512 x 16000 bench batch, plus the step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
dev = torch.device("cuda", 0)
net = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "network.npz"))
B, L = 512, 16000
sigs = synth.make_signals(20260103, B, L)
sig, off, ln, lens = pack_reads(list(sigs), dev)
ref = Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype="f32w", device=dev)
pref = ref.classify_raw(sig, off, ln, lens).cpu().numpy()
ref.close()
for dt in sys.argv[1:] or ["bf16x3", "f16x3"]:
    models, worst, flips = {}, 0.0, 0
    for seed, Lc, Bc, first in net["cases"]:
        seed = int(seed)
        if seed not in models:
            models[seed] = Model(synth.make_state_dict(seed), synth.Config(), None, "m", dtype=dt, device=dev)
        s = synth.make_signals(int(net["sig_seed"][0]), int(Bc), int(Lc), first_read=int(first))
        sg, of, l_, lh = pack_reads(list(s), dev)
        got = models[seed].classify_raw(sg, of, l_, lh).cpu().numpy()
        want = net[f"s{seed}_L{Lc}_B{Bc}_r{first}.probs"]
        worst = max(worst, float(np.abs(got - want).max()))
        flips += int(((got[:, 1] > 0.9) != (want[:, 1] > 0.9)).sum())
    m = models[1]
    out = torch.empty((B, 2), device=dev)
    for _ in range(20): m.classify_raw(sig, off, ln, lens, out=out)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(100): m.classify_raw(sig, off, ln, lens, out=out)
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t) / 100
    p = out.cpu().numpy()
    print("%s lib=%s golden: max|dp| %.2e flips %d | 512x16000 vs fp32: max|dp| %.2e flips %d | %.4f ms/step = %.0f chunks/s" % (
        dt, os.path.basename(os.environ.get("RISER_AMD_LIB", "shipped")), worst, flips, float(np.abs(p - pref).max()),
        int(((p[:, 1] > 0.9) != (pref[:, 1] > 0.9)).sum()), dtm * 1e3, B / dtm), flush=True)
    for mm in models.values(): mm.close()
