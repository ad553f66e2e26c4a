"""Small batches (round 4): the per-wave fp32 Winograd kernel (csrc/conv_small_f32.hip) that runs the conv layers of
launches with only a handful of rows - Model.classify at batch 1, thin ReadUntil batches - against the tiled kernels it
replaces: bit for bit, at every batch size around the hand-over, and against the oracle."""
import numpy as np
import pytest
import torch

from oracle import riser_oracle as ro
from riser_amd import synth

from conftest import hooked_model

pytestmark = pytest.mark.gpu
SIG_SEED = 20260103


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def models(dev):
    from riser_amd.model import Model
    sd = synth.make_state_dict(1)
    small = Model(sd, synth.Config(), None, "m", dtype="f32w", device=dev)          # default: small kernel where it applies
    tiled = hooked_model({"RS_SMALL_F32_WAVES": "0"}, sd, "f32w", dev)              # never
    always = hooked_model({"RS_SMALL_F32_WAVES": "100000000"}, sd, "f32w", dev)     # on every Winograd layer but the fused 0 + 1
    yield small, tiled, always
    for m in (small, tiled, always):
        m.close()


def _reads(lens, first=4100):
    return [synth.make_signals(SIG_SEED, 1, int(n), first_read=first + i)[0] for i, n in enumerate(lens)]


@pytest.mark.parametrize("lens", [[16000], [4096], [8615], [12000, 4097], [5000, 16000, 8191], [8615] * 8,
                                  [4096, 4097, 6024, 8000, 8192, 8615, 12048, 16000]])
def test_small_batches_equal_the_tiled_kernels_bitwise(dev, models, lens):
    from riser_amd.preprocess import pack_reads
    small, tiled, always = models
    sigs = _reads(lens)
    sig, off, ln, lh = pack_reads(sigs, dev)
    want = tiled.classify_raw(sig, off, ln, lh, return_logits=True)
    got = small.classify_raw(sig, off, ln, lh, return_logits=True)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    # the small kernel really ran: its tile report for the last layer is 16 pooled rows (32 conv rows) tall - the smallest
    # tiled shape is 32 pooled rows
    assert small.layer_info()[11]["bm"] == 32 and tiled.layer_info()[11]["bm"] > 32
    oracle = ro.classify_reads(synth.make_state_dict(1), sigs)
    assert np.abs(got[0].cpu().numpy() - oracle).max() < 1e-4


def test_a_read_alone_equals_its_row_of_a_large_batch(dev, models):
    """Model.classify (batch 1, riser/model.py:22-28) against the same read inside a 300-read batch, which runs the tiled
    kernels on every layer: the same bits"""
    from riser_amd.preprocess import pack_reads
    small, _, _ = models
    rng = np.random.default_rng(2)
    lens = rng.integers(4096, 16001, size=300)
    sigs = _reads(lens, first=9000)
    sig, off, ln, lh = pack_reads(sigs, dev)
    big = small.classify_raw(sig, off, ln, lh).cpu().numpy()
    assert small.layer_info()[11]["bm"] > 32                      # the large batch stayed on the tiled kernels
    for k in (0, 17, 151, 299):
        s1, o1, l1, h1 = pack_reads([sigs[k]], dev)
        assert np.array_equal(small.classify_raw(s1, o1, l1, h1).cpu().numpy()[0], big[k])
    # the reference's own entry point: a normalised signal at batch 1
    x = ro.mad_normalise(sigs[17])
    p = small.classify(x).cpu().numpy()
    pb = small.classify_batch([ro.mad_normalise(s) for s in sigs[15:20]]).cpu().numpy()
    assert np.array_equal(p, pb[2])


@pytest.mark.parametrize("B", [12, 24, 48, 96])
def test_hand_over_between_the_kernels(dev, models, B):
    """batch sizes where some layers are below the wave limit and others above it, and the small kernel forced onto every
    layer (hundreds of row tiles, several reads per tile, ragged ends): always the tiled kernels' bits"""
    from riser_amd.preprocess import pack_reads
    small, tiled, always = models
    rng = np.random.default_rng(B)
    lens = rng.integers(4096, 12049, size=B)
    sigs = _reads(lens, first=20000 + B)
    sig, off, ln, lh = pack_reads(sigs, dev)
    want = tiled.classify_raw(sig, off, ln, lh)
    assert torch.equal(small.classify_raw(sig, off, ln, lh), want)
    assert torch.equal(always.classify_raw(sig, off, ln, lh), want)
    info = always.layer_info()
    assert all(info[i]["bn"] in (16, 32) and info[i]["bm"] in (32, 64) for i in range(2, 12))   # 16 units x 16 or 32 channels


@pytest.mark.parametrize("B", [5, 16, 40, 72, 130])
def test_thin_launch_forms_keep_the_bits(dev, models, B):
    """round 5: launches with fewer tiles than CUs run four-wave workgroups (one or two per CU) with their staging loads one
    item ahead, the small kernel shares a tile's input rows inside the workgroup, the streaming kernel shortens its runs -
    against the round-4 forms (eight-wave shapes of a 512-read batch forced onto every tiled layer, default staging distance,
    private input rows, runs of eight) and against each intermediate form: the same bits"""
    from riser_amd.preprocess import pack_reads
    small, tiled, _ = models
    sd = synth.make_state_dict(1)
    big4 = ";".join("%d:8,1,1,4" % i for i in range(2, 12))            # F(4,3) layers: 512 x 64 tiles
    big2 = ";".join("%d:8,1,2,2" % i for i in range(2, 12))            # F(2,3) layers: 512 x 32 tiles
    variants = {
        "round 4": {"RS_SMALL_F32_WAVES": "0", "RS_FORCE_SHAPE_WINO4": big4, "RS_FORCE_SHAPE_WINO": big2, "RS_SF32_MIN_RUN": "8"},
        "default staging": {"RS_NO_DEEP_STAGING": "1", "RS_SMALL_F32_WAVES": "0"},
        "private rows": {"RS_SMALL_SHARED": "0"},
        "one channel sub-tile": {"RS_SMALL_NW": "1"},
        "two channel sub-tiles": {"RS_SMALL_NW": "2", "RS_SMALL_F32_WAVES": "100000000"},
        "runs of one": {"RS_SF32_MIN_RUN": "1"},
    }
    rng = np.random.default_rng(100 + B)
    lens = rng.integers(4096, 16001, size=B)
    sigs = _reads(lens, first=31000 + B)
    sig, off, ln, lh = pack_reads(sigs, dev)
    want = small.classify_raw(sig, off, ln, lh, return_logits=True)
    assert torch.equal(tiled.classify_raw(sig, off, ln, lh, return_logits=True)[0], want[0])
    for name, env in variants.items():
        m = hooked_model(env, sd, "f32w", dev)
        got = m.classify_raw(sig, off, ln, lh, return_logits=True)
        if name == "round 4":
            info = m.layer_info()
            assert all(info[i]["bm"] == 512 for i in range(2, 12)), info
        m.close()
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), name
    oracle = ro.classify_reads(sd, sigs[:6])
    assert np.abs(want[0].cpu().numpy()[:6] - oracle).max() < 1e-4


@pytest.mark.parametrize("dtype", ["bf16x3", "f16x3", "f16"])
def test_streaming_run_length_keeps_the_bits_16bit(dev, dtype):
    """the 16-bit streaming launch (layers 0-2) spreads a thin batch over runs of two blocks: the same bits as runs of eight"""
    from riser_amd.preprocess import pack_reads
    sd = synth.make_state_dict(1)
    a = hooked_model({}, sd, dtype, dev)
    b = hooked_model({"RS_SF32_MIN_RUN": "8"}, sd, dtype, dev)
    for B in (1, 7, 40):
        lens = np.random.default_rng(B).integers(4096, 16001, size=B)
        sig, off, ln, lh = pack_reads(_reads(lens, first=52000 + B), dev)
        ga, gb = a.classify_raw(sig, off, ln, lh, return_logits=True), b.classify_raw(sig, off, ln, lh, return_logits=True)
        assert torch.equal(ga[0], gb[0]) and torch.equal(ga[1], gb[1]), (dtype, B)
    a.close()
    b.close()


def test_forward_path_and_uniform_layout(dev, models):
    """rs_forward (signals that arrive normalised: layer 1 is a Winograd launch of its own) and the layout without host
    lengths (every read in the blocks of the longest)"""
    from riser_amd import Kit, SignalProcessor
    from riser_amd.preprocess import pack_reads
    small, tiled, always = models
    lens = [6000, 4096, 9000]
    sigs = _reads(lens, first=777)
    sig, off, ln, lh = pack_reads(sigs, dev)
    proc = SignalProcessor(Kit.create_from_version("RNA004"), device=dev)
    x = proc.normalise_device(sig, off, ln, 3, 9000)
    want = tiled.forward_batch(x, lh, lens_dev=ln)
    assert torch.equal(small.forward_batch(x, lh, lens_dev=ln), want)
    assert torch.equal(always.forward_batch(x, lh, lens_dev=ln), want)
    assert torch.equal(small.classify_raw(sig, off, ln, lh, packed=False), tiled.classify_raw(sig, off, ln, lh, packed=False))


def test_normalise_selects_on_hostile_long_reads(dev):
    """K1's exact selects on LONG reads built to stress a histogram select (the short adversarial cases of
    tests/test_gpu_parity.py never leave one bin per key): period-8 patterns, bimodal data with the median in the gap, ramps,
    plateaus with the median at a plateau edge, half-constant reads, int16 extremes (range > 4096 bins: the refinement pass),
    MAD = 0 and near-MAD = 0 - all bit-exact float64 against the oracle, at 4096 ... 65536 samples.  (Written for a sampled,
    windowed select that round 4 built and measured at half the speed of the plain pass: DESIGN.md 5, K1.)"""
    from riser_amd import Kit, SignalProcessor
    proc = SignalProcessor(Kit.create_from_version("RNA004"), device=dev)
    rng = np.random.default_rng(12)
    reads = []
    for n in (4096, 4097, 5000, 8615, 16000, 16001, 33333, 65536):
        t = np.arange(n)
        base = rng.normal(500, 60, n)
        reads += [
            np.round(base),                                                        # nanopore-like
            np.where(t % 8 == 0, 900, np.round(base)),                             # every sampled key is an outlier level
            np.where(t % 8 == 0, np.round(base), 100 + (t % 7)),                   # the sample sees only the minority
            np.where(t % 2 == 0, 300, 700) + (t % 5),                              # bimodal, median in the gap
            np.where(t < n // 2, 400, 401),                                        # median at a plateau edge (even n: 400.5)
            t % 3000,                                                              # ramp
            np.sort(np.round(base)),                                               # sorted: systematic sampling is exact ...
            np.sort(np.round(base))[::-1].copy(),
            np.where(t < n // 2 + 1, 512, np.round(base)),                         # > half identical: MAD = 0
            np.where(t < n // 2 - 3, 512, np.round(base)),                         # just under half identical
            np.where(t % 97 == 0, 32767, np.where(t % 89 == 0, -32768, np.round(base))),   # int16 extremes: range 65535
            rng.integers(-32768, 32768, n),                                        # uniform over all of int16
            np.full(n, -7),                                                        # constant
            np.where(t % 16 < 8, 510, 511),                                        # two values, period 16
        ]
    sigs = [np.clip(np.asarray(r), -32768, 32767).astype(np.int16) for r in reads]
    got, stats = proc.mad_normalise_batch(sigs, return_stats=True)
    for k, (g, s) in enumerate(zip(got, sigs)):
        med, mad = ro.median_mad(s)
        assert (stats[k, 0], stats[k, 1]) == (med, mad), (k, len(s), stats[k], med, mad)
        want = ro.mad_normalise(s)
        if mad == 0:
            assert not g.any()
        else:
            assert np.array_equal(g, want), (k, len(s))


def test_normalise_long_outlier_runs(dev):
    """Runs of hundreds to thousands of consecutive outliers (a stall, an open pore, what a fixed trim leaves of the adapter)
    are walked by a whole wave, which skips stretches that sit at the clip limit 64 samples at a time (normalise.hip:
    wave_walk).  Bit-exact float64 AND the fp32 rows the conv stack reads against the oracle's sequential loop: plateaus of
    17 ... 5000 samples at 4 ... 30 MADs on either side, plateaus that change sign inside, that start the read, that end it,
    that are interrupted by single in-range samples, two plateaus back to back, a plateau longer than half the read (it moves
    the median) and a read with more long runs than the hand-over list holds."""
    from riser_amd import Kit, SignalProcessor
    proc = SignalProcessor(Kit.create_from_version("RNA004"), device=dev)
    rng = np.random.default_rng(31)

    def base(n):
        return np.round(rng.normal(500, 40, n))

    reads = []
    for n in (4096, 8615, 16000):
        for length in (17, 63, 64, 65, 128, 129, 1000, 3000):
            for level in (200, 330, 700, 1700):                       # ~ -7.5, -4.2, +5, +30 MADs of the squiggle
                x = base(n)
                at = int(rng.integers(10, n - length - 10))
                x[at: at + length] = level + rng.integers(-3, 4, length)
                reads.append(x)
        x = base(n); x[:700] = 720; reads.append(x)                   # starts the read
        x = base(n); x[-900:] = 310; reads.append(x)                  # ends it
        x = base(n); x[-1:] = 900; x[-400:-1] = 705; reads.append(x)
        x = base(n); x[1000:1500] = 700; x[1500:2100] = 300; reads.append(x)        # changes sign inside
        x = base(n); x[1000:3000] = np.where(np.arange(2000) % 2 == 0, 700, 300); reads.append(x)   # alternates: no stretch at all
        x = base(n); x[2000:2600] = 690; x[2100] = 500; x[2300:2302] = 505; reads.append(x)         # interrupted
        x = base(n); x[500:900] = 700; x[901:1400] = 710; reads.append(x)           # back to back, one sample between
        x = base(n); x[100: 100 + n // 2 + 50] = 900; reads.append(x)               # the plateau IS the median
        x = base(n)
        for k in range(80):                                                          # 80 runs of 20: more than the list holds
            x[40 * k + 5: 40 * k + 25] = 700 if k % 3 else 290
        reads.append(x)
        x = base(n); x[300:2300] = 660 + (np.arange(2000) % 90); reads.append(x)     # a ramp across the clip region
        # value range > 4096 (the windowed exact histogram of the normalise kernel, and its fall-backs)
        x = base(n); x[777] = 9000; reads.append(x)                                  # one spike
        x = base(n); x[n >> 1] = 9000; reads.append(x)                               # ... at one of the three probe samples
        x = base(n); x[n >> 1] = 9000; x[n >> 2] = 9100; reads.append(x)             # two probes on spikes: window misses the median
        x = base(n); x[5] = -9000; x[9] = 12000; x[1000:1800] = 700; reads.append(x)   # spikes on both sides and a plateau
        x = base(n); x[: n // 2 + 10] = 6000 + (np.arange(n // 2 + 10) % 5); reads.append(x)   # bulk far from the probes' window edge
        x = base(n); x[::2] = 5000 + rng.integers(-40, 41, x[::2].shape[0]); reads.append(x)   # bimodal 4500 apart: median between the modes
    sigs = [np.clip(r, -32768, 32767).astype(np.int16) for r in reads]
    got = proc.mad_normalise_batch(sigs)
    for k, (g, sgn) in enumerate(zip(got, sigs)):
        assert np.array_equal(g, ro.mad_normalise(sgn)), (k, len(sgn))
    from riser_amd.preprocess import pack_reads
    sig, off, ln, lh = pack_reads(sigs, dev)
    lmax = int(lh.max())
    rows = proc.normalise_device(sig, off, ln, len(sigs), lmax).cpu().numpy()
    for k, sgn in enumerate(sigs):
        assert np.array_equal(rows[k, : len(sgn)], ro.mad_normalise(sgn).astype(np.float32)), k
        assert not rows[k, len(sgn):].any()


def test_normalise_fuzz_structured_reads(dev):
    """600 reads with random STRUCTURE - a squiggle plus any mix of plateaus (10 ... 6000 samples, 3 ... 40 MADs off, either side),
    spikes (in and far beyond the 4096-count window), level shifts and constant stretches - bit-exact float64 against the oracle's
    sequential loop.  What the fixed cases of the tests above cannot enumerate: how the pieces of the normalise kernel (windowed
    histogram and its fall-backs, run list, lane walk, wave walk and its hand-over) meet."""
    from riser_amd import Kit, SignalProcessor
    proc = SignalProcessor(Kit.create_from_version("RNA004"), device=dev)
    rng = np.random.default_rng(20260107)
    sigs = []
    for k in range(600):
        n = int(rng.integers(4096, 20001))
        sd = float(rng.choice([8, 25, 40, 90]))
        x = np.round(rng.normal(rng.integers(300, 900), sd, n))
        for _ in range(int(rng.integers(0, 5))):                              # plateaus
            length = int(min(n - 2, rng.choice([10, 17, 40, 64, 65, 300, 1500, 6000])))
            at = int(rng.integers(0, n - length))
            x[at: at + length] = x[at: at + length].mean() + rng.choice([-1, 1]) * rng.uniform(3, 40) * sd + rng.integers(-2, 3, length)
        for _ in range(int(rng.integers(0, 4))):                              # spikes
            x[rng.integers(0, n, size=int(rng.integers(1, 6)))] = rng.choice([-20000, -3000, 2500, 6000, 15000, 32000])
        if rng.random() < 0.2:                                                # a level shift for the rest of the read
            at = int(rng.integers(1, n))
            x[at:] += rng.choice([-1, 1]) * rng.uniform(1, 12) * sd
        if rng.random() < 0.1:                                                # a constant stretch (ties)
            at = int(rng.integers(0, n - 50))
            x[at: at + int(rng.integers(50, n - at))] = np.round(x[at])
        sigs.append(np.clip(x, -32768, 32767).astype(np.int16))
    got, stats = proc.mad_normalise_batch(sigs, return_stats=True)
    for k, (g, sgn) in enumerate(zip(got, sigs)):
        med, mad = ro.median_mad(sgn)
        assert (stats[k, 0], stats[k, 1]) == (med, mad), (k, len(sgn))
        if mad == 0:
            assert not g.any()
        else:
            assert np.array_equal(g, ro.mad_normalise(sgn)), (k, len(sgn))



@pytest.mark.parametrize("dtype", ["bf16x3", "f16x3", "f16xf8"])
def test_thin_launch_kernel_of_split_precision_keeps_the_bits(dev, dtype):
    """round 6: the split-precision layers of a THIN launch - Model.classify at batch 1
    (riser/model.py:22-28), the ~20 reads a chunk client delivers (riser/control.py:63-69) - run 64 x 32 tiles that take a whole
    32-channel panel (all three taps) per barrier (csrc/conv_thin_h16.hip) instead of the ring kernel's (panel, tap) sub-stages.
    Same LDS image, same MFMA sequence per accumulator, same epilogue roundings: identical bits to the ring / weights-resident
    kernels (RS_THIN_H16_ROWS=0) with the kernel chosen by the planner AND forced onto every launch, for 1 ... 130 reads of
    ragged lengths; a read alone equals its row of a 300-read batch; within the mode's tolerance of the oracle."""
    from riser_amd.preprocess import pack_reads
    sd = synth.make_state_dict(1)
    auto = hooked_model({}, sd, dtype, dev)
    ring = hooked_model({"RS_THIN_H16_ROWS": "0"}, sd, dtype, dev)
    forced = hooked_model({"RS_THIN_H16_ROWS": "100000000"}, sd, dtype, dev)
    for B in (1, 5, 16, 40, 130):
        rng = np.random.default_rng(200 + B)
        lens = rng.integers(4096, 16001, size=B)
        sigs = _reads(lens, first=41000 + B)
        sig, off, ln, lh = pack_reads(sigs, dev)
        want = ring.classify_raw(sig, off, ln, lh, return_logits=True)
        for name, m in (("planner", auto), ("forced", forced)):
            got = m.classify_raw(sig, off, ln, lh, return_logits=True)
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), (dtype, B, name)
        if B == 1 and dtype != "f16xf8":
            info = auto.layer_info()
            assert info[11]["bm"] == 64 and info[11]["bn"] == 32 and info[4]["bm"] == 64     # the planner took the thin kernel
        if B == 5:
            oracle = ro.classify_reads(sd, sigs)
            assert np.abs(want[0].cpu().numpy() - oracle).max() < 1e-3
    info = forced.layer_info()
    assert all(info[i]["bm"] == 64 and info[i]["bn"] == 32 for i in range(3, 7))   # layers 3-6 are split-precision rows in all three modes
    # a read alone against its row of a large batch (which stays on the big tiles)
    rng = np.random.default_rng(3)
    lens = rng.integers(4096, 16001, size=300)
    sigs = _reads(lens, first=43000)
    sig, off, ln, lh = pack_reads(sigs, dev)
    big = auto.classify_raw(sig, off, ln, lh).cpu().numpy()
    assert auto.layer_info()[7]["bm"] > 64 and auto.layer_info()[11]["bm"] > 64
    for k in (0, 123, 299):
        s1, o1, l1, h1 = pack_reads([sigs[k]], dev)
        assert np.array_equal(auto.classify_raw(s1, o1, l1, h1).cpu().numpy()[0], big[k])
    for m in (auto, ring, forced):
        m.close()


# conv_ring_h16.hip's and conv_ring_f8.hip's shape tables (WM, WN, MT, NT): a forced shape that a kernel does not hold is ignored
RING_SHAPES = [(8, 1, 2, 2), (8, 1, 2, 3), (8, 1, 2, 5), (8, 1, 2, 7), (8, 1, 4, 2), (8, 1, 4, 3), (8, 1, 4, 4), (4, 2, 4, 3), (4, 2, 4, 4),
               (4, 2, 4, 5), (4, 2, 4, 6), (4, 2, 2, 4), (4, 2, 2, 6), (2, 4, 4, 4), (2, 4, 2, 4), (2, 4, 4, 3), (8, 1, 3, 3), (8, 1, 3, 4),
               (4, 2, 3, 4), (4, 2, 3, 5), (4, 2, 3, 6), (4, 2, 5, 4), (4, 2, 6, 4), (4, 2, 1, 1), (4, 2, 1, 2), (2, 4, 1, 1)]
F8_SHAPES = [(8, 1, 2, 2), (8, 1, 2, 4), (8, 1, 2, 6), (4, 2, 4, 4), (4, 2, 4, 6), (4, 2, 2, 4), (4, 2, 2, 6), (2, 4, 4, 4), (2, 4, 2, 4),
             (4, 2, 2, 2), (2, 4, 2, 2), (4, 2, 6, 4), (2, 4, 6, 2), (2, 4, 6, 4)]


@pytest.mark.parametrize("dtype", ["bf16x3", "f16x3", "f16xf8"])
def test_every_tile_shape_of_the_16bit_ring_kernels_keeps_the_bits(dev, dtype):
    """round 6: the ring kernels' tables hold 192- / 320- / 384-row tiles next to the powers of two (a batch that is not a multiple
    of 256 reads fills whole rounds of the CUs with them) and the 8-bit kernel's widest shape keeps no F sub-stage back.  A tile
    shape only regroups rows and channels: every accumulator sees the same panel -> tap -> (hi hi, lo hi, hi lo | 8-bit)
    sequence, so EVERY shape of both tables, forced onto every ring layer, must reproduce the planner's bits on a ragged batch;
    and the planner's own result is within the mode's tolerance of the oracle."""
    from riser_amd.preprocess import pack_reads
    sd = synth.make_state_dict(1)
    rng = np.random.default_rng(77)
    lens = np.concatenate([rng.integers(4096, 16001, size=70), [16000] * 26])
    sigs = _reads(lens, first=47000)
    sig, off, ln, lh = pack_reads(sigs, dev)
    auto = hooked_model({"RS_THIN_H16_ROWS": "0"}, sd, dtype, dev)
    want = auto.classify_raw(sig, off, ln, lh, return_logits=True)
    oracle = ro.classify_reads(sd, sigs[:12])
    assert np.abs(want[0][:12].cpu().numpy() - oracle).max() < 1e-3
    auto.close()
    taken = 0
    for sh in sorted(set(RING_SHAPES + (F8_SHAPES if dtype == "f16xf8" else []))):
        force = ";".join("%d:%d,%d,%d,%d" % ((layer,) + sh) for layer in range(3, 12))
        m = hooked_model({"RS_THIN_H16_ROWS": "0", "RS_FORCE_SHAPE_RING": force}, sd, dtype, dev)
        got = m.classify_raw(sig, off, ln, lh, return_logits=True)
        bm, bn = sh[0] * 16 * sh[2], sh[1] * 16 * sh[3]
        taken += any(i["bm"] == bm and i["bn"] == bn for i in m.layer_info()[4:12])
        m.close()
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), (dtype, sh)
    assert taken >= 20, taken                    # the forces were honoured (a shape whose LDS does not fit a layer is skipped)
