import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from riser_amd import synth
from riser_amd.model import Model
from riser_amd.stream import StreamClassifier
print("HSA_ENABLE_SDMA", os.environ.get("HSA_ENABLE_SDMA"))
dev = torch.device("cuda", 0)
B, L, nb = 512, 16000, 48
sigs = synth.make_signals(20260103, B, L)
pop = torch.from_numpy(np.ascontiguousarray(sigs)).pin_memory()
pinned = torch.empty((nb * B, L), dtype=torch.int16).pin_memory()
for k in range(nb): pinned[k*B:(k+1)*B].copy_(pop)
# raw H2D bandwidth
d = torch.empty(B * L, dtype=torch.int16, device=dev)
torch.cuda.synchronize()
t=time.perf_counter()
for k in range(20): d.copy_(pinned[k*B:(k+1)*B].reshape(-1), non_blocking=True)
torch.cuda.synchronize()
print("H2D 16.4 MB: %.3f ms each" % ((time.perf_counter()-t)/20*1e3))
for dt in (sys.argv[1:] or ["f32w", "bf16x3", "f16xf8", "bf16x3", "f32w"]):
    m = Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype=dt, device=dev)
    sc = StreamClassifier([m], sub_batch=B, max_len=L)
    sc.classify(pinned[:4*B])
    res=[]
    for rep in range(4):
        torch.cuda.synchronize(); t=time.perf_counter(); sc.classify(pinned); res.append((time.perf_counter()-t)/nb*1e3)
    # resident
    sig = pinned[:B].reshape(-1).to(dev); off = torch.arange(B, dtype=torch.int64, device=dev)*L; ln = torch.full((B,), L, dtype=torch.int32, device=dev); lh=np.full(B,L,dtype=np.int32)
    for _ in range(10): m.classify_raw(sig, off, ln, lh)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(100): m.classify_raw(sig, off, ln, lh)
    torch.cuda.synchronize(); r=(time.perf_counter()-t)/100*1e3
    print(dt, "dev bufs at", [hex(t.data_ptr()) for t in sc._dev], "ws", {k: hex(v.data_ptr()) for k, v in getattr(m._ws, "_bufs", {}).items()} if hasattr(m._ws, "_bufs") else "", "host-fed ms/batch", ["%.3f"%x for x in res], "resident %.3f" % r)
    m.close()
