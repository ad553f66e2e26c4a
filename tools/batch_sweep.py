"""chunks/s versus batch size and chunk length (HBM-resident), all dtypes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
dev = torch.device("cuda", 0)
sd = synth.make_state_dict(1)
SHAPES = os.environ.get("RS_SWEEP_SHAPES")          # "B:L,B:L,...": only these
for dt in (sys.argv[1:] or ["f32w", "f16"]):
    m = Model(sd, synth.Config(), None, "mRNA", dtype=dt)
    for B, L in ((1, 16000), (2, 16000), (4, 16000), (8, 16000), (16, 16000), (32, 16000), (64, 16000), (96, 16000), (128, 16000), (256, 16000), (320, 16000), (357, 16000), (448, 16000), (512, 16000), (576, 16000), (640, 16000), (704, 16000), (768, 16000), (896, 16000), (1024, 16000), (1280, 16000), (1536, 16000), (2048, 16000), (3072, 16000), (4096, 16000), (357, 8615), (512, 8615), (2048, 8615), (512, 6024)) if not SHAPES else [tuple(int(v) for v in x.split(":")) for x in SHAPES.split(",")]:
        sigs = synth.make_signals(20260103, min(B, 64), L)
        sigs = np.tile(sigs, ((B + len(sigs) - 1) // len(sigs), 1))[:B]
        sig, off, ln, lens = pack_reads(list(sigs), dev)
        out = torch.empty((B, 2), device=dev)
        for _ in range(12): m.classify_raw(sig, off, ln, lens, out=out)
        torch.cuda.synchronize(); n = max(int(os.environ.get("RS_SWEEP_MIN_STEPS", 3)), min(50, int(2000 / max(B, 1)) + 3)); t = time.perf_counter()
        for _ in range(n): m.classify_raw(sig, off, ln, lens, out=out)
        torch.cuda.synchronize(); dtm = (time.perf_counter() - t) / n
        print(f"{dt} B={B:5d} L={L:6d}: {dtm*1e3:8.3f} ms/batch  {B/dtm:10.0f} chunks/s")
    m.close()
