import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
from oracle import riser_oracle as ro
dev = torch.device("cuda", 0)
rng = np.random.default_rng(7)
sigs = []
for rid in range(304):
    n = int(rng.integers(16000, 22000))
    s = synth.make_raw_read(4242, rid, n, polya=(rid % 5 != 0))
    end = ro.polya_end(s)
    start = min((end + 1) if end else 2048, n - 8615)
    sigs.append(np.ascontiguousarray(s[start: start + 8615]))
print("value range per read: median", np.median([int(s.max()) - int(s.min()) for s in sigs]), "max", max(int(s.max()) - int(s.min()) for s in sigs))
y = ro.mad_normalise(sigs[0]); 
print("outlier share read 0:", float((np.abs((sigs[0] - np.median(sigs[0])) / (1.4826 * np.median(np.abs(sigs[0] - np.median(sigs[0]))))) > 3.5).mean()))
m = Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype="f32w", device=dev)
for name, batch in (("raw-read squiggles", sigs), ("make_signals", [synth.make_signals(20260103, 1, len(s), first_read=i)[0] for i, s in enumerate(sigs)])):
    sig, off, ln, lh = pack_reads(batch, dev)
    for _ in range(3): m.classify_raw(sig, off, ln, lh)
    m.profile(True)
    for _ in range(10): m.classify_raw(sig, off, ln, lh)
    ms, calls = m.profile_read(); m.profile(False)
    per = ms / calls
    print(f"{name}: norm {per[0]*1e3:.1f} us, step {per.sum():.3f} ms")
