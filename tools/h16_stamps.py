"""One f16 forward under the -DRS_ITEM_STAMPS build of conv_h16.hip (tools/ablate_build.py conv_h16.hip stamps=RS_ITEM_STAMPS;
RISER_AMD_LIB=riser_amd/lib/libabl_stamps.so): prints the per-phase cycle sums of the tiled kernel's item loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
B, L = 512, 16000
sigs = synth.make_signals(20260103, B, L)
sig, off, ln, lens = pack_reads(list(sigs), torch.device("cuda", 0))
m = Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype=sys.argv[1] if len(sys.argv) > 1 else "f16")
for _ in range(3):
    m.classify_raw(sig, off, ln, lens)
torch.cuda.synchronize()
