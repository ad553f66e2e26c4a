"""debug: packed layout vs oracle per read for a set of lengths / batch compositions"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
from oracle import torch_path
dev = torch.device("cuda", 0)
sd = synth.make_state_dict(1)
cpu = torch_path.TorchCpuModel(sd)
for dtype in sys.argv[1:] or ["f32w"]:
    m = Model(sd, synth.Config(), None, "t", dtype=dtype, device=dev)
    for lens in ([4096], [6024], [8192], [8615], [12048], [16000], [12048, 5000], [5000, 12048], [4096, 4097, 8191, 8192, 8193, 16000],
                 [12048, 7000, 4096, 9000, 12047]):
        sigs = [synth.make_signals(20260103, 1, n, first_read=50 + i)[0] for i, n in enumerate(lens)]
        sig, off, ln, lh = pack_reads(sigs, dev)
        want = torch_path.classify_per_read(cpu, sigs)
        for packed in (True, False):
            got = m.classify_raw(sig, off, ln, lh, packed=packed).cpu().numpy()
            print(dtype, lens, "packed" if packed else "uniform", np.abs(got - want).max(axis=1), flush=True)
    m.close()
