"""Time every fp32 conv tile shape on chosen layers (RS_FORCE_SHAPE_F32 / RS_FORCE_SHAPE_WINO), to calibrate the
planner:  python tools/shape_sweep.py 4,7,9,10,11 [f32|f32w]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads
B = int(os.environ.get("RS_B", 512)); L = int(os.environ.get("RS_L", 16000))
layers = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,7,9,10,11").split(",")]
DT = sys.argv[2] if len(sys.argv) > 2 else "f32"
W4 = DT == "f32w4"                       # F(4,3) on the listed layers (RS_WINO4), swept with RS_FORCE_SHAPE_WINO4
if W4:
    os.environ["RS_WINO4"] = ",".join(str(l) for l in layers)
    DT = "f32w"
WINO4 = [(8,1,1,2),(8,1,1,3),(8,1,1,4),(8,1,1,5),(8,1,1,6),(4,2,1,2),(4,2,1,3),(4,2,1,4),(4,2,2,2),(2,4,1,2),(2,4,1,3),(2,4,1,4),(2,4,2,2),(4,2,1,1),(2,4,1,1),(4,1,1,1),(2,2,1,1),(1,4,1,1),(4,1,1,2),(2,2,1,2),(1,4,1,2),(4,1,1,3),(2,2,1,3)]
WINO = [(8,1,2,2),(8,1,2,3),(8,1,2,4),(8,1,1,5),(8,1,1,6),(8,1,1,7),(8,1,1,8),(4,2,2,3),(4,2,2,4),(4,2,1,5),(4,2,1,7),(4,2,1,8),(2,4,2,2),(2,4,2,3),(2,4,1,4),(8,1,1,2),(8,1,1,3),(8,1,1,4),(4,2,1,2),(4,2,1,3),(4,2,1,4),(2,4,1,2),(2,4,1,1),(4,1,1,1),(2,2,1,1),(4,1,1,2),(2,2,1,2),(1,4,1,2),(4,1,1,3),(2,2,1,3)]
shapes = [(8,1,4,2),(8,1,4,3),(8,1,2,5),(8,1,4,5),(8,1,2,7),(8,1,4,7),(4,2,4,2),(4,2,4,3),(4,2,2,4),(4,2,4,4),(4,2,4,5),(4,2,2,6),(4,2,4,6),(4,2,4,7),(4,2,2,8),(2,4,2,2),(2,4,2,4),(2,4,1,4),(4,2,2,5),(2,4,4,2),(8,1,2,6),(4,2,2,3)]
sigs = synth.make_signals(20260103, 64, L); sigs = np.tile(sigs, ((B + 63) // 64, 1))[:B]
dev = torch.device("cuda", 0)
sig, off, ln, lens = pack_reads(list(sigs), dev)
m = Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype=DT)
RING = [(8,1,2,2),(8,1,2,3),(8,1,2,5),(8,1,2,7),(8,1,4,2),(8,1,4,3),(8,1,4,4),(4,2,4,3),(4,2,4,4),(4,2,4,5),(4,2,4,6),(4,2,2,4),(4,2,2,6),(2,4,4,4),(2,4,2,4),(2,4,4,3),
        (8,1,3,3),(8,1,3,4),(4,2,3,4),(4,2,3,5),(4,2,3,6),(4,2,5,4),(4,2,6,4)]
RING_F8 = [(8,1,2,2),(8,1,2,4),(8,1,2,6),(4,2,4,4),(4,2,4,6),(4,2,2,4),(4,2,2,6),(2,4,4,4),(2,4,2,4),(4,2,6,4),(2,4,6,2),(2,4,6,4)]   # conv_ring_f8.hip (layers on F8 rows)
IS_RING = DT in ("bf16x3", "f16x3", "f16", "bf16", "f16xf8")     # every tiled 16-bit layer runs the ring kernel
if DT == "f32w": shapes = WINO4 if W4 else WINO
if IS_RING: shapes = RING_F8 if DT == "f16xf8" else RING
ROWMUL = 4 if W4 else 2 if DT == "f32w" else 1
def run():
    global m
    # the library reads RS_FORCE_SHAPE_* once, when a model is created
    m.close(); m = Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype=DT)
    for _ in range(2): m.classify_raw(sig, off, ln, lens)
    m.profile(True)
    for _ in range(4): m.classify_raw(sig, off, ln, lens)
    ms, calls = m.profile_read(); m.profile(False)
    return ms / calls, m.layer_info()
base, info = run()
print("B", B, "default:", " ".join("L%d[%dx%d]=%.3f" % (i, info[i]["bm"], info[i]["bn"], base[1 + i]) for i in layers))
for sh in shapes:
    os.environ["RS_FORCE_SHAPE_RING" if IS_RING else "RS_FORCE_SHAPE_WINO4" if W4 else "RS_FORCE_SHAPE_WINO" if DT == "f32w" else "RS_FORCE_SHAPE_F32"] = ";".join("%d:%d,%d,%d,%d" % ((l,) + sh) for l in layers)
    ms, info = run()
    bm, bn = sh[0]*16*sh[2]*ROWMUL, sh[1]*16*sh[3]
    print("%-12s %4dx%-4d" % (sh, bm, bn), " ".join(("L%d=%.3f" % (i, ms[1 + i])) if (info[i]["bm"], info[i]["bn"]) == (bm, bn) else ("L%d=  -  " % i) for i in layers))
