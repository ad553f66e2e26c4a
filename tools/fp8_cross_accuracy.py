#!/usr/bin/env python3
"""CPU emulation: can the two CROSS terms of the split-precision product run on the block-scaled 8-bit (or 6-bit)
matrix instruction?  ("CPU emulation first ... go on only if max |dp| <= 3e-4 and 0 flips".)

    python tools/fp8_cross_accuracy.py [--max-batch 64] [--variants e4m3,e5m2u,...] [--bench512 N]

Split precision today (csrc/conv_ring_h16.hip): v = hi + lo, hi = round16(v), lo = round16(v - hi);
a*w ~= hi_a*hi_w + hi_a*lo_w + lo_a*hi_w on three 16-bit MFMAs per 32 K.  The proposal keeps hi_a*hi_w on
v_mfma_f32_16x16x32_f16 and evaluates (hi_a*lo_w + lo_a*hi_w) as ONE K-concatenated product on
v_mfma_scale_f32_16x16x128_f8f6f4: A = [q(hi_a) | q(lo_a)], B = [q(lo_w) | q(hi_w)] with q = an 8-bit (e4m3 / e5m2) or
6-bit (e2m3 / e3m2) float and an E8M0 scale per 32-element K block (one activation row x one 32-channel panel; one weight
column x panel x tap) - 2 instruction-times per 32 K instead of 3 (e4m3 / e5m2) or 1.5 (e2m3 / e3m2).

Every rounding of that path is put in by hand, float64 elsewhere (the kernels accumulate in fp32: ~1e-7 relative):
  * hi = f16(v) (or bf16), lo = v - hi kept EXACT for the q() of the cross terms (the epilogue holds the fp32 value);
  * block scale 2^e, e = floor(log2(max |block|)) - emax(format) + guard: guard 0 is the OCP MX rule (the block's top binade may
    saturate), guard 1 never saturates; `tied`: lo's scale = hi's scale - 11 (one scale byte per row and panel) instead of its own;
  * `w`: the WEIGHT planes carry one power of two per layer instead of block scales (f16 weights are already scaled to
    max |w| in [8192, 16384): hi x 2^-6, lo x 2^5 fit e4m3 with 14 binades of full precision below the layer's largest weight);
  * `u` variants: no block scale at all (unit scale; lo planes carry the fixed 2^11 / 2^8 of the split);
  * q() rounds to nearest even, saturates at the format's largest finite value, keeps subnormals.
Prints max |dp| against the reference's fixture probabilities (tests/golden/network.npz) and the accept / reject labels
(0.9) that differ.  Test infrastructure: imports oracle/ for the normalisation only.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import riser_oracle as ro          # noqa: E402
from riser_amd import synth                     # noqa: E402

T16 = {"bf16": torch.bfloat16, "f16": torch.float16}
#            exponent bits, mantissa bits, bias, largest finite
FMT = {"e4m3": (4, 3, 7, 448.0), "e5m2": (5, 2, 15, 57344.0), "e2m3": (2, 3, 1, 7.5), "e3m2": (3, 2, 3, 28.0)}


def qmini(x, fmt):
    """round float64 x to the mini-float `fmt` (nearest even, saturating, subnormals kept)"""
    E, M, bias, top = FMT[fmt]
    ax = x.abs()
    e = torch.floor(torch.log2(torch.clamp(ax, min=1e-300)))
    e = torch.clamp(e, min=float(1 - bias))                    # below the smallest normal: the subnormal quantum
    q = torch.exp2(e - M)
    r = torch.round(ax / q) * q                                # torch.round is half-to-even
    r = torch.clamp(r, max=top)
    return torch.sign(x) * r


def block_scale(x, fmt, guard, dim):
    """2^e per 32-element block along `dim` (already reshaped so that `dim` has size 32)"""
    E, M, bias, top = FMT[fmt]
    emax = int(np.floor(np.log2(top)))
    amax = x.abs().amax(dim=dim, keepdim=True)
    e = torch.floor(torch.log2(torch.clamp(amax, min=2.0 ** -126))) - emax + guard
    e = torch.clamp(e, min=-127.0, max=127.0)                  # E8M0
    return torch.exp2(e)


def qblock_act(x, fmt, guard, scale=None):
    """x: (B, C, L) float64 -> de-quantised float64; blocks = 32 consecutive channels at one position"""
    B, C, L = x.shape
    Cp = (C + 31) // 32 * 32
    xp = torch.zeros((B, Cp, L), dtype=torch.float64)
    xp[:, :C] = x
    xb = xp.reshape(B, Cp // 32, 32, L)
    s = block_scale(xb, fmt, guard, 2) if scale is None else scale
    out = (qmini(xb / s, fmt) * s).reshape(B, Cp, L)[:, :C]
    return out, s


def qblock_w(w, fmt, guard):
    """w: (N, C, 3) float64; blocks = 32 consecutive input channels of one output channel and tap"""
    N, C, K = w.shape
    Cp = (C + 31) // 32 * 32
    wp = torch.zeros((N, Cp, K), dtype=torch.float64)
    wp[:, :C] = w
    wb = wp.reshape(N, Cp // 32, 32, K)
    s = block_scale(wb, fmt, guard, 2)
    return (qmini(wb / s, fmt) * s).reshape(N, Cp, K)[:, :C]


def wscale(w, kind):
    if kind != "f16":
        return 1.0
    m = float(w.abs().max())
    k = int(np.floor(np.log2(8192.0 / m)))
    while m * 2.0 ** k >= 16384:
        k -= 1
    while m * 2.0 ** k < 8192:
        k += 1
    return 2.0 ** k


def conv3(x, w):
    return torch.nn.functional.conv1d(x, w, padding=1)


def r16(x, kind):
    return x.to(T16[kind]).to(torch.float64)


class Variant:
    def __init__(self, name):
        # name = <hi kind>:<fmt>[u][t][g0|g1][r]   e.g. f16:e4m3g1, f16:e5m2u, bf16:e4m3t
        self.name = name
        kind, rest = name.split(":")
        self.kind = kind
        self.fmt = None
        for f in FMT:
            if rest.startswith(f):
                self.fmt, rest = f, rest[len(f):]
        self.unit = "u" in rest
        self.tied = "t" in rest
        self.guard = 0 if "g0" in rest else 1
        self.lo16 = "r" in rest                 # q() of the ROUNDED 16-bit lo instead of the exact one
        self.wunit = "w" in rest                # weights: no block scales, one power of two per layer and plane
        self.x3 = self.fmt is None              # plain three-term 16-bit split (today's kernels)


def cross_conv(x32, w, v):
    """one conv layer: hi*hi on the 16-bit MFMA, the cross terms per the variant; float64 accumulate"""
    kind = v.kind
    s = wscale(w, kind)
    ws = w * s
    wh = r16(ws, kind)
    wl_exact = ws - wh
    xh = r16(x32, kind)
    xl_exact = x32 - xh
    if v.x3:
        xl, wl = r16(xl_exact, kind), r16(wl_exact, kind)
        return (conv3(xh, wh) + conv3(xl, wh) + conv3(xh, wl)) / s
    if v.lo16:
        xl_exact, wl_exact = r16(xl_exact, kind), r16(wl_exact, kind)
    shift = 11.0 if kind == "f16" else 8.0
    if v.unit:
        # no block scales: q(hi) as it is, q(lo * 2^shift) (same dynamic range as hi)
        E, M, bias, top = FMT[v.fmt]
        xh8 = qmini(xh, v.fmt)
        xl8 = qmini(xl_exact * 2.0 ** shift, v.fmt) * 2.0 ** -shift
        wh8 = qmini(wh, v.fmt)
        wl8 = qmini(wl_exact * 2.0 ** shift, v.fmt) * 2.0 ** -shift
    else:
        xh8, sh = qblock_act(xh, v.fmt, v.guard)
        xl8, _ = qblock_act(xl_exact, v.fmt, v.guard, scale=(sh * 2.0 ** -shift) if v.tied else None)
        if v.wunit:
            # f16 weights are scaled to max |w| in [8192, 16384): hi plane x 2^-6 (<= 256), lo plane x 2^(shift - 6)
            wh8 = qmini(wh * 2.0 ** -6, v.fmt) * 2.0 ** 6
            wl8 = qmini(wl_exact * 2.0 ** (shift - 6), v.fmt) * 2.0 ** (6 - shift)
        else:
            wh8 = qblock_w(wh, v.fmt, v.guard)
            wl8 = qblock_w(wl_exact, v.fmt, v.guard)
    return (conv3(xh, wh) + conv3(xl8, wh8) + conv3(xh8, wl8)) / s


def forward(sd, x, v):
    h = torch.from_numpy(x.astype(np.float64))[:, None, :]
    n_layers = sum(1 for k in sd if k.startswith("layers.") and k.endswith(".0.weight"))
    for i in range(n_layers):
        w = torch.from_numpy(np.asarray(sd[f"layers.{i}.0.weight"], dtype=np.float64))
        b = torch.from_numpy(np.asarray(sd[f"layers.{i}.0.bias"], dtype=np.float64))[None, :, None]
        P = h.shape[2] // 2
        if i == 0:                      # layer 0 runs in fp32 (f32-input MFMA) in every mode
            y = torch.relu(conv3(h, w) + b)
        else:
            x32 = h.to(torch.float32).to(torch.float64)          # values live in fp32 before they are split
            y = torch.relu(cross_conv(x32, w, v) + b)
        h = torch.maximum(y[:, :, 0:2 * P:2], y[:, :, 1:2 * P:2])
    feat = h.to(torch.float32).to(torch.float64).mean(dim=2)
    wfc = torch.from_numpy(np.asarray(sd["classifier.2.weight"], dtype=np.float64))
    bfc = torch.from_numpy(np.asarray(sd["classifier.2.bias"], dtype=np.float64))
    return torch.softmax(feat @ wfc.T + bfc, dim=1).numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="f16:,bf16:,f16:e4m3g1,f16:e4m3g0,f16:e4m3tg1,f16:e4m3rg1,f16:e5m2u,f16:e5m2g1,f16:e2m3g1,"
                                          "f16:e2m3g0,f16:e3m2g1,bf16:e4m3g1")
    ap.add_argument("--max-batch", type=int, default=64)
    ap.add_argument("--bench512", type=int, default=0, help="also the first N reads of the bench batch (make_signals(20260103, 512, 16000), "
                                                           "seed 1) against the float64 direct evaluation")
    args = ap.parse_args()
    variants = [Variant(n) for n in args.variants.split(",")]
    net = np.load(os.path.join(ROOT, "tests", "golden", "network.npz"))
    torch.set_num_threads(8)
    res = {}
    t0 = time.time()
    for seed, L, B, first in net["cases"]:
        if B > args.max_batch:
            continue
        tag = f"s{seed}_L{L}_B{B}_r{first}"
        sigs = synth.make_signals(int(net["sig_seed"][0]), int(B), int(L), first_read=int(first))
        x = np.stack([ro.mad_normalise(s).astype(np.float32) for s in sigs])
        sd = synth.make_state_dict(int(seed))
        want = net[f"{tag}.probs"]
        for v in variants:
            p = forward(sd, x, v)
            dp = float(np.abs(p - want).max())
            fl = int(((p[:, 1] > 0.9) != (want[:, 1] > 0.9)).sum())
            a = res.setdefault(v.name, [0.0, 0, 0])
            a[0], a[1], a[2] = max(a[0], dp), a[1] + fl, a[2] + int(B)
        print(tag, "%.0fs" % (time.time() - t0), {k: "%.2e" % r[0] for k, r in res.items()}, flush=True)
    print(f"\ncross terms of the split product on the block-scaled 8/6-bit MFMA, {next(iter(res.values()))[2]} reads of the "
          "reference's golden cases:")
    for name, (dp, fl, n) in res.items():
        print(f"  {name:14s}: max |dp| vs reference {dp:.2e}, label flips {fl}")
    if args.bench512:
        n = args.bench512
        sigs = synth.make_signals(20260103, 512, 16000)[:n]
        sd = synth.make_state_dict(1)
        res = {}
        for c0 in range(0, n, 16):
            x = np.stack([ro.mad_normalise(s).astype(np.float32) for s in sigs[c0:c0 + 16]])
            want = forward64(sd, x)
            for v in variants:
                p = forward(sd, x, v)
                a = res.setdefault(v.name, [0.0, 0])
                a[0] = max(a[0], float(np.abs(p - want).max()))
                a[1] += int(((p[:, 1] > 0.9) != (want[:, 1] > 0.9)).sum())
            print("bench reads", c0 + 16, "%.0fs" % (time.time() - t0), {k: "%.2e" % r[0] for k, r in res.items()}, flush=True)
        print(f"\nfirst {n} reads of the bench batch (512 x 16000, state dict seed 1) against float64:")
        for name, (dp, fl) in res.items():
            print(f"  {name:14s}: max |dp| {dp:.2e}, label flips {fl}")


def forward64(sd, x):
    h = torch.from_numpy(x.astype(np.float64))[:, None, :]
    n_layers = sum(1 for k in sd if k.startswith("layers.") and k.endswith(".0.weight"))
    for i in range(n_layers):
        w = torch.from_numpy(np.asarray(sd[f"layers.{i}.0.weight"], dtype=np.float64))
        b = torch.from_numpy(np.asarray(sd[f"layers.{i}.0.bias"], dtype=np.float64))[None, :, None]
        P = h.shape[2] // 2
        y = torch.relu(conv3(h, w) + b)
        h = torch.maximum(y[:, :, 0:2 * P:2], y[:, :, 1:2 * P:2])
    feat = h.mean(dim=2)
    wfc = torch.from_numpy(np.asarray(sd["classifier.2.weight"], dtype=np.float64))
    bfc = torch.from_numpy(np.asarray(sd["classifier.2.bias"], dtype=np.float64))
    return torch.softmax(feat @ wfc.T + bfc, dim=1).numpy()


if __name__ == "__main__":
    main()
