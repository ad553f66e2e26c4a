import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from riser_amd import synth, SignalProcessor, Kit
from riser_amd.preprocess import pack_reads
dev = torch.device("cuda", 0)
proc = SignalProcessor(Kit.create_from_version("RNA004"), device=dev)
for B, L in ((512, 9000), (3600, 9000), (3600, 4500)):
    sigs = synth.make_signals(7, B, L)
    sig, off, ln, lens = pack_reads(list(sigs), dev)
    for _ in range(3): proc.polyA_end_device(sig, off, ln, B)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): out = proc.polyA_end_device(sig, off, ln, B)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(f"polya B={B} L={L}: {dt*1e3:.3f} ms, found {(out > 0).sum().item()}")
