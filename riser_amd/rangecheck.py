"""Which arithmetic modes a set of weights can run in: the largest activation of every conv layer on sample reads.

The half-precision modes (`f16x3`, `f16`) store activations as IEEE half: a value above 65504 saturates there, silently
(the probabilities stay finite and are wrong).  Weights of any magnitude are fine (they are packed with a power-of-two
scale); what cannot be fixed ahead of time is the ACTIVATION range, which depends on the weights and on the data.  This
module runs the fp32 path on sample reads with the library's layer-capture hook (`rs_debug_capture_layer`) and reports the
per-layer maxima, so that the choice between `f16x3` (22 mantissa bits) and `bf16x3` (16 bits, fp32's range) is made on
evidence:

    python -m riser_amd.rangecheck --model state.pth [--reads 64] [--samples 8615]

Without real signals at hand it uses the synthetic raw reads of riser_amd.synth (MAD-normalised input is confined to
[-3.5, 3.5] whatever the source, which is what makes a sample meaningful)."""
import argparse
import sys

import numpy as np
import torch

from . import _native as nv
from . import synth
from .model import Model
from .preprocess import pack_reads

HALF_MAX = 65504.0


def activation_range(state, config=None, signals=None, device=None) -> list:
    """-> [max |activation| of conv layer i's output for i = 1 .. n_layers - 1] on `signals` (raw int16 reads), fp32 path"""
    config = config or synth.Config()
    dev = device or torch.device("cuda", torch.cuda.current_device())
    if signals is None:
        signals = [synth.make_raw_read(4242, rid, 12000 + 37 * rid, polya=bool(rid % 5))[2048:] for rid in range(64)]
    m = Model(state, config, None, "range", dtype="f32w", device=dev)
    lens = [len(s) for s in signals]
    sig, off, ln, lh = pack_reads(list(signals), dev)
    info = m.layer_info()
    out = []
    try:
        for i in range(1, m.n_layers):
            U, bases = m.block_samples(i), m.block_bases(lens, i)
            cap = torch.zeros((int(bases[-1]) * (U >> (i + 1)), info[i]["cp_out"]), dtype=torch.float32, device=dev)
            nv.check(nv.lib().rs_debug_capture_layer(m._h, i, cap.data_ptr(), cap.numel() * 4), "rs_debug_capture_layer")
            m.classify_raw(sig, off, ln, lh)
            out.append(float(cap.abs().max().item()))
    finally:
        nv.lib().rs_debug_capture_layer(m._h, -1, None, 0)
        m.close()
    return out


def verdict(maxima: list, margin: float = 4.0) -> str:
    worst = max(maxima)
    if worst * margin < HALF_MAX:
        return f"f16x3 / f16 are safe on this sample: the largest activation is {worst:.4g} (limit 65504, margin x{margin:g} kept)"
    if worst < HALF_MAX:
        return (f"largest activation {worst:.4g}: inside half precision's range on this sample but with less than x{margin:g} to "
                "spare - prefer bf16x3")
    return f"largest activation {worst:.4g} exceeds half precision's 65504: use bf16x3 (or fp32); f16x3 / f16 would saturate"


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model", help="state dict (.pth) of the shipped 12-layer config; default: synthetic weights, seed 1")
    ap.add_argument("--reads", type=int, default=64)
    ap.add_argument("--samples", type=int, default=8615)
    args = ap.parse_args(argv)
    state = args.model or synth.make_state_dict(1)
    sigs = [synth.make_raw_read(4242, rid, 2048 + args.samples + 11 * rid, polya=bool(rid % 5))[2048: 2048 + args.samples]
            for rid in range(args.reads)]
    maxima = activation_range(state, signals=sigs)
    for i, v in enumerate(maxima, start=1):
        print(f"layer {i:2d}: max |activation| {v:.5g}")
    print(verdict(maxima))
    return 0


if __name__ == "__main__":
    sys.exit(main())
