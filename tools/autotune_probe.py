"""rs_autotune on the bench batch: step time before / after, layers changed, bit-identity.  python tools/autotune_probe.py [dtype ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
B, L = int(os.environ.get("RS_B", 512)), int(os.environ.get("RS_L", 16000))
dev = torch.device("cuda", 0)
sigs = synth.make_signals(20260103, B, L)
sig, off, ln, lens = pack_reads(list(sigs), dev)
def timeit(m, n=60):
    for _ in range(30): m.classify_raw(sig, off, ln, lens)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): m.classify_raw(sig, off, ln, lens)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for dt in (sys.argv[1:] or ["f32w", "f16"]):
    m = Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype=dt)
    ref = m.classify_raw(sig, off, ln, lens).cpu().numpy()
    t0 = timeit(m)
    before = m.layer_info()
    t = time.perf_counter(); changed = m.autotune(sig, off, ln, lens); torch.cuda.synchronize(); tt = time.perf_counter() - t
    got = m.classify_raw(sig, off, ln, lens).cpu().numpy()
    t1 = timeit(m); t0b = None
    info = m.layer_info()
    print(f"{dt}: {t0:.4f} ms -> {t1:.4f} ms after autotune ({changed} layers changed, tuning took {tt*1e3:.0f} ms); "
          f"max |dp| vs before {np.abs(got - ref).max():.2e}; tiles " + " ".join(
              f"L{i}[{info[i]['bm']}x{info[i]['bn']}]" + ("" if (before[i]['bm'], before[i]['bn']) == (info[i]['bm'], info[i]['bn'])
                                                         else f"(planner {before[i]['bm']}x{before[i]['bn']})") for i in range(1, 12)))
    m.close()
