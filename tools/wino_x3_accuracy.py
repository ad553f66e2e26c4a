#!/usr/bin/env python3
"""CPU emulation: is Winograd F(2,3) in SPLIT PRECISION (bf16x3 / f16x3) inside the accuracy budget?  (VERDICT round 4,
item 1a: "First check accuracy on the CPU ... must stay <= 3e-4, 0 flips".)

    python tools/wino_x3_accuracy.py [--layers 5-11] [--f43]

For every (seed, length, batch) case of tests/golden/network.npz (the reference's own probabilities) the forward pass is
evaluated in float64 with the ROUNDINGS of the 16-bit split-precision path put in by hand:

  * an activation / weight v is the pair hi = round16(v), lo = round16(v - hi); a product is hi*hi + lo*hi + hi*lo
    (csrc/conv_ring_h16.hip); accumulation in float64 (the kernels accumulate in fp32: ~1e-7 relative, far below the
    split's 2^-17);
  * "direct": every layer as the kernels compute it today;
  * "wino":   the layers named by --layers as Winograd F(2,3): input transform B^T d evaluated in fp32 on the (hi + lo)
    values and RE-SPLIT into (hi, lo) - what a producing layer's epilogue would write -, weights G g evaluated in float64
    and split AFTER the transform, four component products of three MFMA terms each, output transform in fp32;
    --f43 uses F(4,3) instead (six components; constants 1/4, 1/6, 1/24 ...).
  * f16: weights carry the per-layer power-of-two scale of ConvLayerDev::w_unscale (max |w| 2^k in [8192, 16384)).

Prints max |dp| against the reference's fixture probabilities and the number of accept/reject labels (0.9) that differ.
Test infrastructure: imports oracle/ for the normalisation only.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import riser_oracle as ro          # noqa: E402
from riser_amd import synth                     # noqa: E402

T16 = {"bf16": torch.bfloat16, "f16": torch.float16}


def r16(x, kind):
    return x.to(T16[kind]).to(torch.float64)


def split(x, kind):
    x32 = x.to(torch.float32).to(torch.float64)          # values live in fp32 before they are split
    hi = r16(x32, kind)
    lo = r16(x32 - hi, kind)
    return hi, lo


def wscale(w, kind):
    if kind != "f16":
        return 1.0
    m = float(w.abs().max())
    k = int(np.floor(np.log2(8192.0 / m))) + (1 if m * 2.0 ** (int(np.floor(np.log2(8192.0 / m)))) < 8192 else 0)
    while m * 2.0 ** k >= 16384:
        k -= 1
    while m * 2.0 ** k < 8192:
        k += 1
    return 2.0 ** k


def conv3(x, w):
    return torch.nn.functional.conv1d(x, w, padding=1)


def x3_conv(xh, xl, w, kind):
    """direct lowering: three terms of the split product, 'same' conv, float64 accumulate"""
    s = wscale(w, kind)
    wh, wl = split(w * s, kind)
    return (conv3(xh, wh) + conv3(xl, wh) + conv3(xh, wl)) / s


def wino23(xh, xl, w, kind):
    """F(2,3) over pooling pairs: y[2T], y[2T+1] from d0..d3 = x[2T-1..2T+2]; transformed inputs re-split to (hi, lo)"""
    B, C, L = xh.shape
    x = (xh + xl).to(torch.float32)                                     # exact: |lo| <= ulp(hi) / 2
    P = L // 2
    xp = torch.zeros((B, C, 2 * P + 3), dtype=torch.float32)
    n = min(L, 2 * P + 2)
    xp[:, :, 1:1 + n] = x[:, :, :n]
    d0, d1, d2, d3 = xp[:, :, 0:2 * P:2], xp[:, :, 1:2 * P + 1:2], xp[:, :, 2:2 * P + 2:2], xp[:, :, 3:2 * P + 3:2]
    t = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]                              # fp32, one rounding each
    g0, g1, g2 = (w[:, :, k].to(torch.float64) for k in range(3))
    u = [g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2]
    s = max(wscale(uk, kind) for uk in u) if kind == "f16" else 1.0
    if kind == "f16":
        s = min(wscale(uk, kind) for uk in u)
    m = []
    for tk, uk in zip(t, u):
        th, tl = split(tk.to(torch.float64), kind)
        uh, ul = split(uk * s, kind)
        m.append((torch.einsum("nc,bcp->bnp", uh, th) + torch.einsum("nc,bcp->bnp", uh, tl) +
                  torch.einsum("nc,bcp->bnp", ul, th)) / s)
    m = [mk.to(torch.float32) for mk in m]                                # accumulators are fp32
    y0 = (m[0] + m[1]) + m[2]
    y1 = (m[1] - m[2]) - m[3]
    return y0.to(torch.float64), y1.to(torch.float64)


# F(4,3): Lavin & Gray's matrices
BT43 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                 [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G43 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                [0, 0, 1]], dtype=np.float64)
AT43 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def wino43(xh, xl, w, kind):
    """F(4,3): four conv outputs (two pooling pairs) from six inputs; returns (even outputs, odd outputs) like wino23"""
    B, C, L = xh.shape
    x = (xh + xl).to(torch.float32)
    G4 = (L // 2 + 1) // 2                                                 # groups of 4 conv outputs
    xp = torch.zeros((B, C, 4 * G4 + 3), dtype=torch.float32)
    n = min(L, 4 * G4 + 2)
    xp[:, :, 1:1 + n] = x[:, :, :n]
    d = [xp[:, :, k:4 * G4 + k:4] for k in range(6)]
    bt = torch.from_numpy(BT43).to(torch.float32)
    t = [sum(float(bt[r, k]) * d[k] for k in range(6) if bt[r, k] != 0) for r in range(6)]     # fp32
    g = w.to(torch.float64)
    u = [sum(G43[r, k] * g[:, :, k] for k in range(3)) for r in range(6)]
    m = []
    for tk, uk in zip(t, u):
        s = wscale(uk, kind)
        th, tl = split(tk.to(torch.float64), kind)
        uh, ul = split(uk * s, kind)
        m.append(((torch.einsum("nc,bcp->bnp", uh, th) + torch.einsum("nc,bcp->bnp", uh, tl) +
                   torch.einsum("nc,bcp->bnp", ul, th)) / s).to(torch.float32))
    at = torch.from_numpy(AT43).to(torch.float32)
    y = [sum(float(at[r, k]) * m[k] for k in range(6) if at[r, k] != 0).to(torch.float64) for r in range(4)]
    P = L // 2
    even = torch.stack([y[0], y[2]], dim=3).reshape(B, -1, 2 * G4)[:, :, :P]
    odd = torch.stack([y[1], y[3]], dim=3).reshape(B, -1, 2 * G4)[:, :, :P]
    return even, odd


def forward(sd, x, kind, wino_layers, f43):
    h = torch.from_numpy(x.astype(np.float64))[:, None, :]
    n_layers = sum(1 for k in sd if k.startswith("layers.") and k.endswith(".0.weight"))
    for i in range(n_layers):
        w = torch.from_numpy(np.asarray(sd[f"layers.{i}.0.weight"], dtype=np.float64))
        b = torch.from_numpy(np.asarray(sd[f"layers.{i}.0.bias"], dtype=np.float64))[None, :, None]
        P = h.shape[2] // 2
        if i == 0:                      # layer 0 runs in fp32 (f32-input MFMA) in every mode
            y = conv3(h, w) + b
            y = torch.relu(y)
            h = torch.maximum(y[:, :, 0:2 * P:2], y[:, :, 1:2 * P:2])
            continue
        hh, hl = split(h, kind)
        if i in wino_layers:
            y0, y1 = (wino43 if f43 else wino23)(hh, hl, w, kind)
            h = torch.relu(torch.maximum(y0, y1) + b)
        else:
            y = torch.relu(x3_conv(hh, hl, w, kind) + b)
            h = torch.maximum(y[:, :, 0:2 * P:2], y[:, :, 1:2 * P:2])
    feat = h.to(torch.float32).to(torch.float64).mean(dim=2)
    wfc = torch.from_numpy(np.asarray(sd["classifier.2.weight"], dtype=np.float64))
    bfc = torch.from_numpy(np.asarray(sd["classifier.2.bias"], dtype=np.float64))
    logits = feat @ wfc.T + bfc
    return torch.softmax(logits, dim=1).numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", default="5-11")
    ap.add_argument("--f43", action="store_true")
    ap.add_argument("--max-batch", type=int, default=64)
    args = ap.parse_args()
    lo, hi = (int(v) for v in args.layers.split("-"))
    wl = set(range(lo, hi + 1))
    net = np.load(os.path.join(ROOT, "tests", "golden", "network.npz"))
    torch.set_num_threads(8)
    res = {}
    for seed, L, B, first in net["cases"]:
        if B > args.max_batch:
            continue
        tag = f"s{seed}_L{L}_B{B}_r{first}"
        sigs = synth.make_signals(int(net["sig_seed"][0]), int(B), int(L), first_read=int(first))
        x = np.stack([ro.mad_normalise(s).astype(np.float32) for s in sigs])
        sd = synth.make_state_dict(int(seed))
        want = net[f"{tag}.probs"]
        for kind in ("bf16", "f16"):
            for name, layers in (("direct", set()), ("wino", wl)):
                p = forward(sd, x, kind, layers, args.f43)
                dp = float(np.abs(p - want).max())
                fl = int(((p[:, 1] > 0.9) != (want[:, 1] > 0.9)).sum())
                a = res.setdefault((kind, name), [0.0, 0, 0])
                a[0], a[1], a[2] = max(a[0], dp), a[1] + fl, a[2] + int(B)
        print(tag, {f"{k[0]}x3 {k[1]}": "%.2e" % v[0] for k, v in res.items()}, flush=True)
    print(f"\nWinograd F({'4' if args.f43 else '2'},3) on layers {args.layers}, {res[('bf16', 'wino')][2]} reads of the reference's golden cases:")
    for (kind, name), (dp, fl, n) in res.items():
        print(f"  {kind}x3 {name:6s}: max |dp| vs reference {dp:.2e}, label flips {fl}")


if __name__ == "__main__":
    main()
