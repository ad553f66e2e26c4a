#!/usr/bin/env python3
"""profiles/README.md's kernel table: every kernel name of a rocprofv3 `--kernel-trace --stats` summary of bench.py mapped to the
ConvNet layers it runs and its SURVEY.md 8(d) roofline fraction, recomputed here from nothing but the CSV and the layer shapes
(so the figure can be re-derived without reading bench.py):

    python tools/profile_table.py profiles/r05_kernel_stats_f32.csv f32 gpurun_out/bench_detail_rna004_b512_f32.json
    python tools/profile_table.py profiles/r05_kernel_stats_bf16x3.csv bf16x3 gpurun_out/bench_detail_rna004_b512_bf16x3.json

frac = un-padded direct-convolution FLOPs of the kernel's layers for 512 x 16000-sample chunks (2 C_in C_out 3 L_i per chunk and
layer) / the kernel's mean duration x its launches per step / the dense MFMA peak of the mode (157.3 TF f32-input, 2 500 TF
16-bit).  The third argument (the bench's detail file of the same mode) supplies the tile shape each layer ran with, which is
what ties a template instantiation to its layers."""
import csv
import json
import re
import sys

CH = [20, 30, 45, 67, 100, 150, 225, 337, 505, 757, 1135, 1702]
B, L0 = 512, 16000
PEAK = {"f32": 157.3, "bf16x3": 2500.0, "f16x3": 2500.0}


def layer_flops():
    out, c_in, L = [], 1, L0
    for c in CH:
        out.append(2.0 * c_in * c * 3 * L * B)
        c_in, L = c, L // 2
    return out


def main():
    stats, mode, detail = sys.argv[1:4]
    fl = layer_flops()
    tiles = {x["layer"]: tuple(x["tile"]) for x in json.load(open(detail))["roofline"]["layers"]}
    rows = list(csv.DictReader(open(stats)))
    steps = max(int(r["Calls"]) for r in rows if "normalise_kernel" in r["Name"])
    print(f"| kernel ({mode}) | launches / step | layers | mean µs | algorithmic TF | frac of {PEAK[mode]:.0f} TF |")
    print("|---|---|---|---|---|---|")
    tot_us = 0.0
    for r in rows:
        name = r["Name"].replace("void rs::(anonymous namespace)::", "").replace("rs::(anonymous namespace)::", "").split("(")[0]
        n = int(r["Calls"]) / steps
        us = float(r["AverageNs"]) / 1e3
        m = re.match(r"(conv_\w+?)_kernel<([^>]*)>", name)
        layers = []
        if m:
            k, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
            if k in ("conv_stream_f32",):
                layers = [0, 1]
            elif k == "conv_stream012_h16":
                layers = [0, 1, 2]
            elif k == "conv_wres_h16":                      # weights-resident kernel <WM-rows factor, NT>: layer 3 in split precision
                layers = [3]
            else:
                wm, wn, mt, nt = (int(a) for a in args[:4])
                if k == "conv_wino4":
                    shape = (4 * wm * 16 * mt, wn * 16 * nt, int(args[4]))
                elif k == "conv_wino":
                    shape = (2 * wm * 16 * mt, wn * 16 * nt, int(args[4]))
                else:                                       # 16-bit tiled kernels: conv rows x channels, 32-channel panels
                    shape = (wm * 16 * mt, wn * 16 * nt, 32)
                layers = [i for i, t in tiles.items() if t == shape and i >= (3 if mode != "f32" else 2)]
        if not name.startswith(("conv", "normalise", "head")):
            continue
        tot_us += us * n
        if layers:
            f = sum(fl[i] for i in layers)
            tf = f / (us * 1e-6 * n) / 1e12
            print(f"| `{name}` | {n:g} | {', '.join(map(str, layers))} | {us:.1f} | {tf:.1f} | {tf / PEAK[mode]:.3f} |")
        else:
            print(f"| `{name}` | {n:g} | - | {us:.1f} | - | - |")
    conv = sum(fl[1:])
    print(f"\nstep: {tot_us:.0f} µs of kernels; layers 1-11: {conv / 1e9:.1f} GFLOP per step (582.95 MFLOP x {B})")


if __name__ == "__main__":
    main()
