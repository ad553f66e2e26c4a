"""Deterministic synthetic weights and squiggle signals (host side, numpy only).

Nothing here is on the compute path: it only manufactures inputs for tests, the
golden-vector generator (tools/make_golden.py) and bench.py, because the reference's
trained weights are not distributed (/root/reference/.MISSING_LARGE_BLOBS:1-6) and
there is no network for datasets.  Every value is a pure function of integer
(seed, stream, index) counters pushed through a 64-bit mixing hash, so the same
arrays are rebuilt bit-for-bit on the GPU box without shipping 41.8 MB per model and
without depending on the numpy / torch RNG streams.

Shapes follow the reference network (riser/nets/cnn.py:13-33, channel list at
riser/model/mRNA_config_RNA004_RP4.yaml:7-12) and its state_dict keys
(`layers.{i}.0.weight|bias`, `classifier.2.weight|bias`).
"""
from __future__ import annotations

import numpy as np

CHANNELS = (20, 30, 45, 67, 100, 150, 225, 337, 505, 757, 1135, 1702)
KERNELS = (3,) * 12
N_CLASSES = 2

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x += np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def _key(seed: int, stream: int) -> np.uint64:
    k = _mix64(np.array([seed & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))
    with np.errstate(over="ignore"):
        k = _mix64(k ^ (np.uint64(stream) * np.uint64(0xD6E8FEB86659FD93)))
    return k[0]


def hash_u64(seed: int, stream: int, n: int, start: int = 0) -> np.ndarray:
    """n hashed uint64 for counters start..start+n-1 of (seed, stream)."""
    idx = np.arange(start, start + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _mix64(idx * np.uint64(0x2545F4914F6CDD1D) ^ _key(seed, stream))


def uniform_pm1(seed: int, stream: int, n: int) -> np.ndarray:
    """float32 uniform in [-1, 1) with 24 random bits (exact in fp32)."""
    h = hash_u64(seed, stream, n)
    u = (h >> np.uint64(40)).astype(np.int64) - (1 << 23)          # [-2^23, 2^23)
    return (u.astype(np.float32) * np.float32(2.0 ** -23)).astype(np.float32)


class CnnConfig:
    """Duck-type of the `config.cnn` namespace the reference ConvNet reads
    (riser/nets/cnn.py:13-21)."""

    def __init__(self, channels=CHANNELS, kernels=KERNELS, n_classes=N_CLASSES,
                 classifier="gap_fc", depth=1):
        self.n_layers = len(channels)
        self.depth = depth
        self.channels = list(channels)
        self.kernels = list(kernels)
        self.n_classes = n_classes
        self.classifier = classifier


class Config:
    def __init__(self, cnn=None):
        self.cnn = cnn or CnnConfig()


# Calibration constants (see tools/make_golden.py --calibrate): conv weights are
# uniform(-a, a) with a = GAIN * sqrt(6 / fan_in) (He-uniform would be GAIN = 1);
# biases uniform(-0.05, 0.05); the FC layer is scaled so that logit differences
# spread over roughly +-4 and some probabilities land near the 0.9 threshold
# (torch default init gives a constant 0.494/0.506,
# naive He init saturates to exactly 0/1).
CONV_GAIN = 0.92
BIAS_AMP = 0.05
FC_AMP = 0.055
# Per-seed (fc_scale, logit-difference offset), measured once over 64 synthetic
# 16000-sample reads so that p_on has spread on both sides of 0.5 / 0.9 for the three
# stand-in models (seeds 1/2/3 = mRNA/mtRNA/globin).  Other seeds use the default.
FC_CAL = {1: (18.0, 8.69), 2: (7.5, 1.98), 3: (10.0, 12.4), 4: (2.6, -6.68)}
FC_CAL_DEFAULT = (8.0, 0.0)


def make_state_dict(seed: int, channels=CHANNELS, n_classes=N_CLASSES) -> dict:
    """Reference-compatible state dict of float32 numpy arrays."""
    sd = {}
    c_in = 1
    for i, c_out in enumerate(channels):
        fan_in = c_in * 3
        a = np.float32(CONV_GAIN * np.sqrt(6.0 / fan_in))
        w = uniform_pm1(seed, 2 * i, c_out * c_in * 3) * a
        b = uniform_pm1(seed, 2 * i + 1, c_out) * np.float32(BIAS_AMP)
        sd[f"layers.{i}.0.weight"] = w.reshape(c_out, c_in, 3).astype(np.float32)
        sd[f"layers.{i}.0.bias"] = b.astype(np.float32)
        c_in = c_out
    fc_scale, fc_off = FC_CAL.get(seed, FC_CAL_DEFAULT)
    fw = uniform_pm1(seed, 1000, n_classes * c_in) * np.float32(FC_AMP * fc_scale)
    fb = uniform_pm1(seed, 1001, n_classes) * np.float32(0.1)
    fb[n_classes - 1] += np.float32(fc_off)
    sd["classifier.2.weight"] = fw.reshape(n_classes, c_in).astype(np.float32)
    sd["classifier.2.bias"] = fb.astype(np.float32)
    return sd


def make_signals(seed: int, n_reads: int, length: int, first_read: int = 0,
                 spikes: bool = True) -> np.ndarray:
    """int16 [n_reads, length] ADC-like squiggles, integer arithmetic only.

    Each read is a piecewise-constant level sequence (events of 6..37 samples whose
    level is drawn per event, with a per-read level spread so reads differ in texture)
    plus approximately normal noise (sum of four 16-bit uniforms), centred near 500 ADC
    counts; dwell scale (x1/2/4) and noise sigma (~7/15/30) are drawn per read.  0.5 % of samples are replaced by +-(400..1500) spikes, and one read in 16
    gets forced runs of 2..4 consecutive spikes, to exercise the outlier smoothing of
    riser/preprocess.py:127-139.  Values are clipped to [0, 4095].
    """
    out = np.empty((n_reads, length), dtype=np.int16)
    n_ev_max = length // 6 + 2
    for r in range(n_reads):
        rid = first_read + r
        # --- events -----------------------------------------------------------
        he = hash_u64(seed, 3 * rid, n_ev_max)
        hr = int(hash_u64(seed, 3 * rid + 1, 1, start=1 << 40)[0])
        dwell = (6 + (he & np.uint64(31)).astype(np.int64)) << ((hr >> 8) % 3)   # 6..37, x1/2/4
        spread = 20 + hr % 90
        nmul = (13, 26, 52)[(hr >> 16) % 3]                               # noise sigma ~ 7/15/30
        lev = ((he >> np.uint64(8)) & np.uint64(0xFFFF)).astype(np.int64) - 32768
        lev = (lev * spread) >> 15                                        # +-spread
        ends = np.cumsum(dwell)
        ev = np.searchsorted(ends, np.arange(length, dtype=np.int64), side="right")
        base = 500 + lev[ev]
        # --- noise: Irwin-Hall(4) ~ N(0,1) after scaling ------------------------
        hn = hash_u64(seed, 3 * rid + 1, length)
        s = ((hn & np.uint64(0xFFFF)) + ((hn >> np.uint64(16)) & np.uint64(0xFFFF))
             + ((hn >> np.uint64(32)) & np.uint64(0xFFFF)) + (hn >> np.uint64(48))).astype(np.int64)
        noise = ((s - 131070) * nmul) >> 16
        x = base + noise
        if spikes:
            hs = hash_u64(seed, 3 * rid + 2, length)
            is_spk = (hs % np.uint64(200)) == 0
            if rid % 16 == 0:                                             # forced runs
                starts = np.nonzero((hs % np.uint64(4000)) == 1)[0]
                for s0 in starts:
                    run = 2 + int(hs[s0] >> np.uint64(60)) % 3
                    is_spk[s0:s0 + run] = True
            mag = 400 + ((hs >> np.uint64(20)) % np.uint64(1101)).astype(np.int64)
            sign = np.where(((hs >> np.uint64(12)) & np.uint64(1)) == 1, 1, -1)
            x = np.where(is_spk, 500 + sign * mag, x)
        out[r] = np.clip(x, 0, 4095).astype(np.int16)
    return out


def make_raw_read(seed: int, rid: int, total_len: int, polya: bool = True) -> np.ndarray:
    """int16 [total_len] raw read as MinKNOW would stream it: sequencing adapter, then
    (optionally) a quiet, elevated poly(A) plateau, then the RNA squiggle.  Built so the
    window rule of riser/preprocess.py:49-72 fires (plateau mean > 1.2 x the preceding
    1000 samples with window MAD <= 20) when `polya` is set and cannot fire otherwise.
    """
    hr = int(hash_u64(seed, 5 * rid + 4, 1, start=1 << 41)[0])
    a_len = 1200 + hr % 1400
    p_len = (1500 + (hr >> 16) % 3000) if polya else 0
    a_len = min(a_len, total_len)
    p_len = min(p_len, total_len - a_len)
    r_len = total_len - a_len - p_len
    parts = []
    # adapter: textured, mean ~470
    ad = make_signals(seed ^ 0xA5A5, 1, max(a_len, 1), first_read=rid, spikes=False)[0, :a_len].astype(np.int64)
    parts.append(ad - 30)
    if p_len:
        hn = hash_u64(seed, 5 * rid + 3, p_len)
        s = ((hn & np.uint64(0xFFFF)) + ((hn >> np.uint64(16)) & np.uint64(0xFFFF))
             + ((hn >> np.uint64(32)) & np.uint64(0xFFFF)) + (hn >> np.uint64(48))).astype(np.int64)
        parts.append(760 + (((s - 131070) * 10) >> 16))                    # sigma ~ 6
    if r_len:
        parts.append(make_signals(seed, 1, r_len, first_read=rid)[0].astype(np.int64))
    x = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)
    return np.clip(x, 0, 4095).astype(np.int16)


def _quiet(seed: int, stream: int, n: int, level: int, mul: int = 10) -> np.ndarray:
    """n samples of a quiet plateau: `level` + approximately normal noise of sigma ~ 0.6 * mul."""
    hn = hash_u64(seed, stream, n)
    s = ((hn & np.uint64(0xFFFF)) + ((hn >> np.uint64(16)) & np.uint64(0xFFFF))
         + ((hn >> np.uint64(32)) & np.uint64(0xFFFF)) + (hn >> np.uint64(48))).astype(np.int64)
    return level + (((s - 131070) * mul) >> 16)


def polya_edge_cases(seed: int = 77):
    """[(name, int16 signal)]: raw reads that exercise the corners of the poly(A) window rule
    (riser/preprocess.py:42-79) that make_raw_read cannot reach.  Deterministic, so the fixture
    (tests/golden/polya.npz: `edge_names`, `edge_ends`) stores only the reference's answers."""
    def noisy(stream, n, level=470):
        return make_signals(seed ^ 0x5A5A, 1, max(n, 1), first_read=stream, spikes=False)[0, :n].astype(np.int64) - 500 + level

    def i16(parts):
        return np.clip(np.concatenate(parts), 0, 4095).astype(np.int16)

    cases = []
    # a start is found but no later window is noisy: the plateau runs to the end of the read -> None
    cases.append(("plateau_never_ends", i16([noisy(1, 2300), _quiet(seed, 1, 5200, 760)])))
    # shorter than one window (the while loop never runs), exactly one window, one sample short of two
    cases.append(("shorter_than_a_window", i16([_quiet(seed, 2, 499, 700)])))
    cases.append(("exactly_one_window", i16([_quiet(seed, 3, 500, 700)])))
    cases.append(("one_short_of_two_windows", i16([noisy(2, 500), _quiet(seed, 4, 499, 900)])))
    # a > 20 % rise inside the first 1000 samples: for i <= 1000 the rolling mean IS the window mean (change 0), and at
    # i = 1500 the previous 1000 samples are already on the plateau, so the rule never fires on this rise -> None
    cases.append(("rise_inside_first_1000", i16([noisy(3, 500), _quiet(seed, 5, 2500, 760), noisy(4, 3000, 520)])))
    # the same rise one window later (i = 1500 sees 1000 samples of adapter): found, ended by the RNA
    cases.append(("rise_at_1500", i16([noisy(5, 1500), _quiet(seed, 6, 2000, 760), noisy(6, 3000, 520)])))
    # rolling mean of 0: (mean - 0) / 0 is +inf (> 20: a start if the window is quiet) or nan (0 / 0: never a start)
    cases.append(("rolling_mean_zero", i16([np.zeros(2500, dtype=np.int64), _quiet(seed, 7, 1500, 700), noisy(7, 2500, 520)])))
    cases.append(("all_zero", np.zeros(6000, dtype=np.int16)))
    # a noisy plateau (window MAD > 20) is not a start
    cases.append(("noisy_plateau", i16([noisy(8, 2000), _quiet(seed, 8, 2500, 760, mul=60), noisy(9, 2500, 520)])))
    # start and end both beyond 65536 samples (an AccumulatingCache read that sat in the pore for a long time)
    cases.append(("end_beyond_65536", i16([noisy(10, 66200), _quiet(seed, 9, 2300, 760), noisy(11, 3100, 520)])))
    # 128-window table boundary of the kernel (64000 samples): the rise straddles it
    cases.append(("start_at_64000", i16([noisy(12, 64000), _quiet(seed, 10, 1500, 760), noisy(13, 2500, 520)])))
    cases.append(("end_at_64000", i16([noisy(14, 62000), _quiet(seed, 11, 2000, 760), noisy(15, 2500, 520)])))
    return cases


def normalise_float_cases(seed: int = 20260103):
    """[(name, float16 / float32 / float64 signal)]: pA-scaled versions (x * scale + offset, the form the retrain path feeds to
    mad_normalise, riser/retrain/preprocess.py:79) of integer cases with outliers at both ends, runs, half-integer
    medians and MAD = 0; tests/golden/normalise_float.npz holds the reference's outputs for them."""
    base = make_signals(seed, 1, 5000, first_read=3, spikes=False)[0]

    def spiked(pos_val):
        s = base.copy()
        for p, v in pos_val:
            s[p] = v
        return s

    ints = {"synth_6024": make_signals(seed, 1, 6024, first_read=6024 % 97)[0],
            "synth_runs_8000": make_signals(seed, 1, 8000, first_read=16)[0],
            "out_first_two": spiked([(0, 2500), (1, 2400)]),
            "out_last_two": spiked([(4998, 30), (4999, 2500)]),
            "run5_then_gap_run2": spiked([(300 + k, 1500 + 37 * k) for k in range(5)] + [(306, 1700), (307, 20)]),
            "even_half_median": np.array([1, 2, 3, 4, 5, 6, 7, 8, 100, -50] * 410, dtype=np.int16),
            "odd_len": np.array([3, -2, 7, 7, 1, 0, 9, 11, -30000, 30000, 4] * 373, dtype=np.int16),
            "mad0_constant": np.full(4096, 512, dtype=np.int16)}
    out = []
    for name, s in ints.items():
        for dt in (np.float32, np.float64):
            out.append((f"{name}.{np.dtype(dt).name}", (s.astype(np.float64) * 0.17548 + 3.25).astype(dt)))
    for name, s in ints.items():                     # half precision (round 4): 11-bit values, medians between ties
        out.append((f"{name}.float16", (s.astype(np.float64) * 0.17548 + 3.25).astype(np.float16)))
    return out


FC_CHANNELS = CHANNELS[:4]          # the 4-layer net the reference's `fc` classifier is hard-coded for (riser/nets/cnn.py:24)
FC_POSITIONS, FC_HIDDEN = 753, 4096  # Linear(67 * 753, 4096): reads of 12048 .. 12063 samples


def make_fc_state_dict(seed: int, positions: int = FC_POSITIONS, hidden: int = FC_HIDDEN) -> dict:
    """Reference-format state dict of the 4-layer ConvNet with the `fc` classifier (riser/nets/cnn.py:22-27): the first
    four conv layers of make_state_dict(seed) and Flatten -> Linear(67 * positions, hidden) -> ReLU -> Linear(hidden, 2).
    The 206 M weights of the first Linear are a 4 M-entry hash table read through an index pattern (rebuilt in seconds,
    never stored)."""
    full = make_state_dict(seed)
    sd = {k: v for k, v in full.items() if k.startswith("layers.") and int(k.split(".")[1]) < len(FC_CHANNELS)}
    F = FC_CHANNELS[-1] * positions
    n = 1 << 22
    base = uniform_pm1(seed, 7001, n) * np.float32(np.sqrt(3.0 / F) * 1.6)
    w1 = np.empty((hidden, F), dtype=np.float32)
    f7 = np.arange(F, dtype=np.int64) * 7
    for o0 in range(0, hidden, 256):
        o = np.arange(o0, min(hidden, o0 + 256), dtype=np.int64)
        w1[o0: o0 + o.size] = base[(f7[None, :] + o[:, None] * 131071) & (n - 1)]
    sd["classifier.1.weight"] = w1
    sd["classifier.1.bias"] = uniform_pm1(seed, 7002, hidden) * np.float32(0.05)
    sd["classifier.3.weight"] = (uniform_pm1(seed, 7003, N_CLASSES * hidden) * np.float32(np.sqrt(3.0 / hidden) * 2.0)).reshape(N_CLASSES, hidden)
    sd["classifier.3.bias"] = uniform_pm1(seed, 7004, N_CLASSES) * np.float32(0.1)
    return sd


RESNET_BENCH_CFG = dict(channels=[20, 30, 45, 67], kernel=19, padding=5, stride=3, block="basic", n_layers=4,
                        blocks=[2, 2, 2, 2], n_classes=2)


def make_resnet_state_dict(seed: int, cfg: dict = None) -> dict:
    """Reference-format ResNet state dict (riser/nets/resnet.py:72-99) from the integer hash: He-style conv weights,
    BatchNorm with non-trivial running statistics (so that folding is exercised).  The reference ships no ResNet config
    or weights; the default shape is the SquiggleNet-like stack the ConvNet's channel list descends from."""
    cfg = cfg or RESNET_BENCH_CFG
    sd, stream = {}, [0]

    def u(n):
        stream[0] += 1
        return uniform_pm1(seed, 5000 + stream[0], n)

    def conv(name, co, ci, k, bias=False):
        sd[name + ".weight"] = (u(co * ci * k) * np.float32(np.sqrt(3.0 / (ci * k)))).reshape(co, ci, k).astype(np.float32)
        if bias:
            sd[name + ".bias"] = (u(co) * np.float32(0.05)).astype(np.float32)

    def bn(name, c):
        sd[name + ".weight"] = (1.0 + 0.3 * u(c)).astype(np.float32)
        sd[name + ".bias"] = (0.1 * u(c)).astype(np.float32)
        sd[name + ".running_mean"] = (0.2 * u(c)).astype(np.float32)
        sd[name + ".running_var"] = (1.0 + 0.5 * u(c)).astype(np.float32)

    ch = cfg["channels"]
    conv("conv_block.0", ch[0], 1, cfg["kernel"], bias=True)
    bn("conv_block.1", ch[0])
    in_ch = ch[0]
    bottleneck = cfg["block"] == "bottleneck"
    for i in range(cfg["n_layers"]):
        out_ch = ch[i]
        for j in range(cfg["blocks"][i]):
            stride = 2 if (i > 0 and j == 0) else 1
            pre = f"layers.{i}.{j}"
            if in_ch != out_ch or stride != 1:
                conv(pre + ".shortcut.0", out_ch, in_ch, 1)
                bn(pre + ".shortcut.1", out_ch)
            if bottleneck:
                mid = out_ch // 4
                conv(pre + ".blocks.0.0", mid, in_ch, 1); bn(pre + ".blocks.0.1", mid)
                conv(pre + ".blocks.1.0", mid, mid, 3); bn(pre + ".blocks.1.1", mid)
                conv(pre + ".blocks.2.0", out_ch, mid, 1); bn(pre + ".blocks.2.1", out_ch)
            else:
                conv(pre + ".blocks.0.0", out_ch, in_ch, 3); bn(pre + ".blocks.0.1", out_ch)
                conv(pre + ".blocks.1.0", out_ch, out_ch, 3); bn(pre + ".blocks.1.1", out_ch)
            in_ch = out_ch
    sd["decoder.2.weight"] = (u(cfg["n_classes"] * in_ch) * np.float32(0.3)).reshape(cfg["n_classes"], in_ch).astype(np.float32)
    sd["decoder.2.bias"] = (u(cfg["n_classes"]) * np.float32(0.1)).astype(np.float32)
    return sd
