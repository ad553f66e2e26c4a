"""Layer by layer: the buffers of an f16xf8 model (F8 rows decoded: hi16 + lo8 x scale, and hi8 x scale against hi16) against
those of an f16x3 model on the same reads.  python tools/f8_debug.py [n_reads] [length]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from riser_amd import _native as nv, synth
from riser_amd.model import Model
from riser_amd.preprocess import pack_reads

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
dev = torch.device("cuda", 0)
sigs = synth.make_signals(20260103, B, L)
sig, off, ln, lens = pack_reads(list(sigs), dev)


def e4m3(b):
    b = b.astype(np.int32)
    s = np.where(b & 0x80, -1.0, 1.0)
    e = (b >> 3) & 0xF
    m = b & 7
    v = np.where(e == 0, m * 2.0 ** -9, (8 + m) * 2.0 ** (e - 10.0))
    return s * v


def decode_x3(raw, rows, cp, C):
    a = raw[:rows * cp * 2].view(np.float16).reshape(rows, cp // 64, 2, 32).astype(np.float64)
    hi = a[:, :, 0, :].reshape(rows, -1)[:, :C]
    lo = a[:, :, 1, :].reshape(rows, -1)[:, :C]
    return hi, lo


def decode_f8(raw, rows, cp, C):
    P = cp // 128
    body = raw[:rows * cp * 2].reshape(rows, P, 256)
    hi = body[:, :, :128].copy().view(np.float16).reshape(rows, P * 64).astype(np.float64)[:, :C]
    f = body[:, :, 128:].reshape(rows, P, 2, 2, 32)                 # [row][panel][half][hi8 / lo8][32]
    so = (rows * cp * 2 + 255) // 256 * 256
    stride = (rows + 3) // 4 * 4
    sc = raw[so:so + P * stride * 4].reshape(P, stride, 4)[:, :rows].astype(np.float64)      # [panel][row][s0, s0-11, s1, s1-11]
    sc = np.transpose(sc, (1, 0, 2)).reshape(rows, P, 2, 2)         # [row][panel][half][hi / lo]
    val = e4m3(f) * 2.0 ** (sc[..., None] - 127.0)
    hi8 = val[:, :, :, 0, :].reshape(rows, P * 64)[:, :C]
    lo8 = val[:, :, :, 1, :].reshape(rows, P * 64)[:, :C]
    return hi, lo8, hi8, sc


models = {dt: Model(synth.make_state_dict(1), synth.Config(), None, "m", dtype=dt, device=dev) for dt in ("f16x3", "f16xf8")}
info = {dt: m.layer_info() for dt, m in models.items()}
n_layers = models["f16x3"].n_layers
for i in range(3, n_layers):
    out = {}
    for dt, m in models.items():
        U, bases = m.block_samples(i), m.block_bases([L] * B, i)
        P_out, cp = U >> (i + 1), info[dt][i]["cp_out"]
        rows = int(bases[-1]) * P_out
        cap = torch.zeros(rows * cp * 2 + 256 + (cp // 64) * (rows + 4) * 4 + 1024, dtype=torch.uint8, device=dev)
        nv.check(nv.lib().rs_debug_capture_layer(m._h, i, cap.data_ptr(), cap.numel()), "capture")
        p = m.classify_raw(sig, off, ln, lens).cpu().numpy()
        nv.check(nv.lib().rs_debug_capture_layer(m._h, -1, None, 0), "capture off")
        out[dt] = (cap.cpu().numpy(), rows, cp, p)
    C = info["f16x3"][i]["c_out"]
    raw, rows, cp, p3 = out["f16x3"]
    hi3, lo3 = decode_x3(raw, rows, cp, C)
    v3 = hi3 + lo3
    raw, rows8, cp8, p8 = out["f16xf8"]
    assert rows8 == rows
    f8rows = cp8 != cp or (i + 1 < n_layers and info["f16xf8"][i]["cp_out"] % 128 == 0 and cp8 == 128 * ((C + 63) // 64) and i >= 4 and i < n_layers - 1)
    scale = np.abs(v3).max()
    if f8rows:
        hi, lo8, hi8, sc = decode_f8(raw, rows, cp8, C)
        v8 = hi + lo8
        Cp = (C + 31) // 32 * 32
        def blkmax(x):
            xp = np.zeros((x.shape[0], Cp)); xp[:, :C] = np.abs(x)
            return np.repeat(xp.reshape(x.shape[0], Cp // 32, 32).max(axis=2), 32, axis=1)[:, :C]
        bm = np.maximum(blkmax(hi), 1e-30)
        print("          hi8 - hi16 over block max: %.3e (expect <= 2^-5 = 3.1e-2)   lo8 - (v3 - hi) over block max 2^-11: %.3e" % (
            (np.abs(hi8 - hi) / bm).max(), (np.abs(lo8 - (v3 - hi)) / (bm * 2.0 ** -11)).max()))
        print("layer %2d F8 rows cp %d: max|v - v(f16x3)| / max %.2e   hi16 vs f16x3 hi %.2e   hi8 vs hi16 rel %.2e   lo8 vs exact lo (of ulp) %.2e   scale bytes %d..%d" % (
            i, cp8, np.abs(v8 - v3).max() / scale, np.abs(hi - hi3).max() / scale, (np.abs(hi8 - hi) / np.maximum(np.abs(hi), 1e-30)).max(),
            (np.abs(lo8 - (v3 - hi)) / np.maximum(np.abs(hi) * 2.0 ** -11, 1e-30)).max(), sc.min(), sc.max()))
    else:
        hi, lo = decode_x3(raw, rows, cp8, C)
        print("layer %2d x3 rows: max|v - v(f16x3)| / max %.2e" % (i, np.abs(hi + lo - v3).max() / scale))
    print("          max |dp| %.2e" % np.abs(p8 - p3).max())
