"""ResNet variant (riser/nets/resnet.py): oracle pinned to the reference's outputs (CPU) and the
GPU sequential-conv program against both."""
import json
import os
import types

import numpy as np
import pytest

from oracle import resnet_oracle as rr
from oracle import riser_oracle as ro
from riser_amd import synth


def _load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "resnet.npz"))
    cfg = json.loads(str(g[f"{name}.cfg"]))
    sd = {k[len(name) + 4:]: g[k] for k in g.files if k.startswith(name + ".sd.")}
    return g, cfg, sd


def _inputs(L):
    sigs = synth.make_signals(20260103, 3, L, first_read=40)
    return np.stack([ro.mad_normalise(s) for s in sigs]).astype(np.float32)


@pytest.mark.parametrize("name", ["basic", "bottleneck"])
def test_resnet_oracle_vs_reference(golden_dir, name):
    g, cfg, sd = _load(golden_dir, name)
    for L in (3000, 4097):
        lg = rr.resnet_forward(sd, cfg, _inputs(L))
        assert np.abs(lg - g[f"{name}.L{L}.logits"]).max() < 1e-5
        assert np.abs(ro.softmax(lg) - g[f"{name}.L{L}.probs"]).max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["basic", "bottleneck"])
def test_resnet_gpu_vs_reference(golden_dir, name):
    import torch
    from riser_amd.resnet import ResNetModel
    g, cfg, sd = _load(golden_dir, name)
    config = types.SimpleNamespace(resnet=types.SimpleNamespace(**cfg))
    m = ResNetModel(sd, config, None, "x", device=torch.device("cuda", 0))
    for L in (3000, 4097):
        x = _inputs(L)
        probs, logits = m.classify_batch(x, return_logits=True)
        assert np.abs(logits.cpu().numpy() - g[f"{name}.L{L}.logits"]).max() < 1e-4
        assert np.abs(probs.cpu().numpy() - g[f"{name}.L{L}.probs"]).max() < 1e-3
        one = m.classify(x[1])
        assert one.shape == (2,) and abs(one[1].item() - g[f"{name}.L{L}.probs"][1, 1]) < 1e-3
    if cfg["kernel"] > 4 + 2 * cfg["padding"]:                 # the stem kernel does not fit: torch raises too
        with pytest.raises(ValueError):
            m.classify_batch(np.zeros((1, 4), dtype=np.float32))
    m.close()


BOTTLENECK_WIDE_CFG = dict(channels=[32, 48, 68], kernel=19, padding=5, stride=3, block="bottleneck", n_layers=3,
                           blocks=[2, 2, 1], n_classes=2)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(synth.RESNET_BENCH_CFG), BOTTLENECK_WIDE_CFG], ids=["basic", "bottleneck"])
def test_resnet_fused_blocks_equal_the_op_by_op_program(cfg):
    """rs_seqnet_create fuses the stem (conv + BN + ReLU + MaxPool) and every residual block - basic (two 3x3 convs) and
    bottleneck (1x1, strided 3x3, 1x1), with the 1x1 shortcut, add and ReLU - into one launch each; RS_SEQ_NOFUSE=1 runs
    the program one op per launch.  Same logits to fp32 round-off (stride-2 stages, identity and conv shortcuts, ragged
    last tiles) at several lengths, and against the oracle."""
    import torch
    from riser_amd.resnet import ResNetModel
    dev = torch.device("cuda", 0)
    sd = synth.make_resnet_state_dict(7, cfg)
    config = types.SimpleNamespace(resnet=types.SimpleNamespace(**cfg))
    fused = ResNetModel(sd, config, None, "x", device=dev)
    os.environ["RS_SEQ_NOFUSE"] = "1"
    try:
        plain = ResNetModel(sd, config, None, "x", device=dev)
    finally:
        del os.environ["RS_SEQ_NOFUSE"]
    for L in (3000, 4097, 6024, 16000):
        x = _inputs(L)
        pf, lf = fused.classify_batch(x, return_logits=True)
        pp, lp = plain.classify_batch(x, return_logits=True)
        assert np.abs(lf.cpu().numpy() - lp.cpu().numpy()).max() < 2e-5, L
        want = rr.resnet_forward(sd, cfg, x)
        assert np.abs(lf.cpu().numpy() - want).max() < 1e-4, L
    # a batch that is not a multiple of anything, single read
    x = _inputs(5000)[:1]
    assert np.abs(fused.classify_batch(x).cpu().numpy() - plain.classify_batch(x).cpu().numpy()).max() < 1e-5
    fused.close()
    plain.close()


@pytest.mark.gpu
def test_resnet_basic_blocks_in_split_precision(golden_dir):
    """rs_seqnet_set_mode(RS_BF16X3): the residual basic blocks on the bf16 MFMA in split precision (hi + lo pairs, three MFMAs
    per product) - the north_star's "1D-ResNet forward pass ... MFMA bf16".  Against the REFERENCE's own logits / probabilities
    for its basic-block net (tests/golden/resnet.npz) within the 1e-3 tolerance, and against the oracle and the fp32 program on the
    bench's SquiggleNet-like net at several lengths (stride-2 stages, identity and conv shortcuts, edge tiles, one read alone)."""
    import torch
    from riser_amd.resnet import ResNetModel
    dev = torch.device("cuda", 0)
    g, cfg, sd = _load(golden_dir, "basic")
    config = types.SimpleNamespace(resnet=types.SimpleNamespace(**cfg))
    m = ResNetModel(sd, config, None, "x", device=dev, dtype="bf16x3")
    assert m.dtype == "bf16x3"
    for L in (3000, 4097):
        probs, logits = m.classify_batch(_inputs(L), return_logits=True)
        assert np.abs(probs.cpu().numpy() - g[f"basic.L{L}.probs"]).max() < 1e-3
        assert np.abs(logits.cpu().numpy() - g[f"basic.L{L}.logits"]).max() < 5e-3
    m.close()
    cfg = dict(synth.RESNET_BENCH_CFG)
    sd = synth.make_resnet_state_dict(7, cfg)
    config = types.SimpleNamespace(resnet=types.SimpleNamespace(**cfg))
    x3 = ResNetModel(sd, config, None, "x", device=dev, dtype="bf16x3")
    f32 = ResNetModel(sd, config, None, "x", device=dev)
    worst = 0.0
    for L in (3000, 4097, 6024, 16000):
        x = _inputs(L)
        p3, l3 = x3.classify_batch(x, return_logits=True)
        pf, lf = f32.classify_batch(x, return_logits=True)
        want = rr.resnet_forward(sd, cfg, x)
        assert np.abs(ro.softmax(want) - p3.cpu().numpy()).max() < 1e-3, L
        assert np.abs(l3.cpu().numpy() - want).max() < 5e-3, L
        worst = max(worst, float(np.abs(l3.cpu().numpy() - lf.cpu().numpy()).max()))
    assert 0 < worst < 5e-3          # really another arithmetic, and a close one
    x = _inputs(5000)[:1]
    assert np.abs(x3.classify_batch(x).cpu().numpy() - f32.classify_batch(x).cpu().numpy()).max() < 1e-3
    x3.close()
    f32.close()
    # a program without a fused residual block has nothing to switch: the library says so
    plain_cfg = dict(channels=[100], kernel=7, padding=3, stride=2, block="basic", n_layers=1, blocks=[1], n_classes=2)   # 100 > 80 columns
    with pytest.raises(Exception, match="residual block"):
        ResNetModel(synth.make_resnet_state_dict(7, plain_cfg), types.SimpleNamespace(resnet=types.SimpleNamespace(**plain_cfg)),
                    None, "x", device=dev, dtype="bf16x3")


@pytest.mark.gpu
def test_resnet_bottleneck_blocks_in_split_precision(golden_dir):
    """the bottleneck block (1x1, strided 3x3, 1x1, shortcut, add, ReLU: riser/nets/resnet.py:60-70) on the bf16 MFMA in split
    precision: the reference's golden bottleneck net within 1e-3, the wide bottleneck net against the oracle and the fp32 program"""
    import torch
    from riser_amd.resnet import ResNetModel
    dev = torch.device("cuda", 0)
    g, cfg, sd = _load(golden_dir, "bottleneck")
    os.environ["RS_SEQ_BNECK_X3"] = "1"         # read when a program is created: the kernel is opt-in (measured slower than fp32)
    try:
        _bottleneck_x3_checks(dev, g, cfg, sd)
    finally:
        del os.environ["RS_SEQ_BNECK_X3"]


def _bottleneck_x3_checks(dev, g, cfg, sd):
    from riser_amd.resnet import ResNetModel
    m = ResNetModel(sd, types.SimpleNamespace(resnet=types.SimpleNamespace(**cfg)), None, "x", device=dev, dtype="bf16x3")
    for L in (3000, 4097):
        probs, logits = m.classify_batch(_inputs(L), return_logits=True)
        assert np.abs(probs.cpu().numpy() - g[f"bottleneck.L{L}.probs"]).max() < 1e-3
        assert np.abs(logits.cpu().numpy() - g[f"bottleneck.L{L}.logits"]).max() < 5e-3
    m.close()
    cfg = BOTTLENECK_WIDE_CFG
    sd = synth.make_resnet_state_dict(7, cfg)
    config = types.SimpleNamespace(resnet=types.SimpleNamespace(**cfg))
    x3 = ResNetModel(sd, config, None, "x", device=dev, dtype="bf16x3")
    f32 = ResNetModel(sd, config, None, "x", device=dev)
    worst = 0.0
    for L in (3000, 4097, 6024, 16000):
        x = _inputs(L)
        p3, l3 = x3.classify_batch(x, return_logits=True)
        pf, lf = f32.classify_batch(x, return_logits=True)
        want = rr.resnet_forward(sd, cfg, x)
        scale = max(1.0, float(np.abs(want).max()))
        assert np.abs(l3.cpu().numpy() - want).max() < 5e-3 * scale, L
        assert np.abs(p3.cpu().numpy() - ro.softmax(want)).max() < 1e-3, L
        worst = max(worst, float(np.abs(l3.cpu().numpy() - lf.cpu().numpy()).max()))
    assert 0 < worst < 5e-3 * scale
    x3.close()
    f32.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32w", "bf16x3"])
def test_model_surface_runs_a_resnet_config(tmp_path, dtype):
    """`Model(state, config, logger, target)` with `config.resnet` (riser/nets/resnet.py:72-99) instead of `config.cnn`: the
    reference's Model hard-wires ConvNet (riser/model.py:13); here the same surface - classify of one normalised read, the
    batched raw-read entry point with ragged lengths, the ReadUntil control loop - runs the ResNet, against the oracle's ResNet
    (pinned to the reference's own outputs above) and the oracle's per-read loop."""
    import logging
    import torch
    from riser_amd import Kit, Model, SequencerControl, SignalProcessor
    from riser_amd.fake_client import FakeClient, FakeRead
    from riser_amd.preprocess import pack_reads
    from test_gpu_more import _oracle_loop
    dev = torch.device("cuda", 0)
    cfg = dict(synth.RESNET_BENCH_CFG)
    sd = synth.make_resnet_state_dict(7, cfg)
    config = types.SimpleNamespace(model="resnet", resnet=types.SimpleNamespace(**cfg))
    path = str(tmp_path / "mRNA_model_RNA004_RP4.pth")                 # riser/model.py:19 reads a file
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, path)
    m = Model(path, config, None, "mRNA", dtype=dtype, device=dev)
    assert m.dtype == ("bf16x3" if dtype == "bf16x3" else "f32")
    # one read at a time (riser/control.py:68-69)
    for L in (4097, 8615):
        x = _inputs(L)[1]
        want = ro.softmax(rr.resnet_forward(sd, cfg, x[None]))[0]
        got = m.classify(x)
        assert got.shape == (2,) and np.abs(got.cpu().numpy() - want).max() < 1e-3
    # raw int16 reads of different lengths in one call: normalised on the device, grouped by length for the program
    lens = [4096, 5000, 8615, 5000, 12000]
    sigs = [synth.make_signals(20260103, 1, n, first_read=300 + i)[0] for i, n in enumerate(lens)]
    sig, off, ln, lh = pack_reads(sigs, dev)
    probs = m.classify_raw(sig, off, ln, lh).cpu().numpy()
    for i, s in enumerate(sigs):
        want = ro.softmax(rr.resnet_forward(sd, cfg, ro.mad_normalise(s).astype(np.float32)[None]))[0]
        assert np.abs(probs[i] - want).max() < 1e-3, i
    with pytest.raises(ValueError):                                     # shorter than the stem + stages accept
        m.classify(np.zeros(m.min_length - 1, dtype=np.float32))
    # the control loop with this model: rows, probabilities and decisions of the oracle's per-read loop
    class _Cpu:
        def classify(self, x):
            return torch.from_numpy(ro.softmax(rr.resnet_forward(sd, cfg, np.asarray(x, dtype=np.float32)[None]))[0])
    rng = np.random.default_rng(11)
    batches = [[(ch, FakeRead(f"id-{b * 5 + ch}", synth.make_raw_read(55, b * 5 + ch, int(rng.integers(3000, 24000)),
                                                                      polya=((b * 5 + ch) % 4 != 0))))
                for ch in range(1, 25)] for b in range(2)]
    want_rows, want_rej, want_fin = _oracle_loop(batches, "RNA004", (1,), "enrich", 0.9, {1: _Cpu()})
    proc = SignalProcessor(Kit.create_from_version("RNA004"), device=dev)
    client = FakeClient(batches)
    out = str(tmp_path / "o")
    ctl = SequencerControl(client, [m], proc, logging.getLogger("c"), out)
    ctl.start(); ctl.target("enrich", 0.5, 0.9); ctl.finish()
    lines = open(out + ".csv").read().strip().split("\n")[1:]
    assert len(lines) == len(want_rows) > 10
    for ln_, w in zip(lines, want_rows):
        p = ln_.split(",")
        assert (p[1], int(p[2]), int(p[3])) == w[:3]
        assert np.allclose([float(v) for v in p[5].split(";")], w[4], atol=1e-3)
        if not any(abs(q - 0.9) < 1e-3 or abs(1 - q - 0.9) < 1e-3 for q in w[4]):
            assert p[8] == w[3], (p, w)
    m.close()


RANDOM_BASIC_CFGS = [
    dict(channels=[16, 24, 40, 64], kernel=7, padding=3, stride=2, block="basic", n_layers=4, blocks=[1, 2, 1, 1], n_classes=2),
    dict(channels=[21, 35, 77], kernel=19, padding=5, stride=3, block="basic", n_layers=3, blocks=[2, 1, 2], n_classes=2),
    dict(channels=[8, 80], kernel=5, padding=2, stride=1, block="basic", n_layers=2, blocks=[1, 1], n_classes=2),
    dict(channels=[30], kernel=33, padding=16, stride=4, block="basic", n_layers=1, blocks=[3], n_classes=2),
]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", RANDOM_BASIC_CFGS, ids=["16-24-40-64", "21-35-77", "8-80", "30x3-k33"])
def test_resnet_split_precision_over_other_shapes(cfg):
    """the split-precision stem and block kernels away from the bench's shape: odd and tiny channel counts (8 -> one MFMA
    column tile with a 8-halfword row pitch, 77 / 80 -> five), channel counts that are not multiples of 4, stems of one and
    two k-steps (kernel 33), strides 1 - 4, one to three blocks per stage, lengths that end tiles mid-way - against the oracle
    and the fp32 program"""
    import torch
    from riser_amd.resnet import ResNetModel
    dev = torch.device("cuda", 0)
    sd = synth.make_resnet_state_dict(11, cfg)
    config = types.SimpleNamespace(resnet=types.SimpleNamespace(**cfg))
    x3 = ResNetModel(sd, config, None, "x", device=dev, dtype="bf16x3")
    f32 = ResNetModel(sd, config, None, "x", device=dev)
    for L in (2500, 4099, 9000):
        x = _inputs(L)
        p3, l3 = x3.classify_batch(x, return_logits=True)
        pf, lf = f32.classify_batch(x, return_logits=True)
        want = rr.resnet_forward(sd, cfg, x)
        scale = max(1.0, float(np.abs(want).max()))
        assert np.abs(lf.cpu().numpy() - want).max() < 1e-4 * scale, L
        assert np.abs(l3.cpu().numpy() - want).max() < 5e-3 * scale, L
        assert np.abs(p3.cpu().numpy() - ro.softmax(want)).max() < 1e-3, L
    x3.close()
    f32.close()


@pytest.mark.gpu
@pytest.mark.parametrize("block,dtype", [("basic", "f32"), ("basic", "bf16x3"), ("bottleneck", "f32"), ("bottleneck", "bf16x3")])
def test_resnet_ragged_batch_equals_every_read_alone(block, dtype):
    """rs_seqnet_forward_ragged: the reads of a ReadUntil batch have their own lengths.  One call over the ragged batch - every
    kernel masks by the read's own rows, computed on the device from its length - gives every read the bits of a uniform call
    on it alone: lengths from the program's minimum up, neighbours that differ by one sample, reads far shorter than the row
    pitch (whole tiles skipped), a read that fills it."""
    import torch
    from riser_amd.resnet import ResNetModel
    dev = torch.device("cuda", 0)
    cfg = dict(synth.RESNET_BENCH_CFG) if block == "basic" else BOTTLENECK_WIDE_CFG
    sd = synth.make_resnet_state_dict(7, cfg)
    if block == "bottleneck" and dtype == "bf16x3":
        os.environ["RS_SEQ_BNECK_X3"] = "1"
    try:
        m = ResNetModel(sd, types.SimpleNamespace(resnet=types.SimpleNamespace(**cfg)), None, "x", device=dev, dtype=dtype)
    finally:
        os.environ.pop("RS_SEQ_BNECK_X3", None)
    net = m._net
    assert net.ragged_ok
    lens = [4096, 4097, 4098, 5000, 8615, 8614, 300, 12000, 6024, 16000, 333, 7999]
    ld = 16000
    base = _inputs(ld)
    x = np.zeros((len(lens), ld), dtype=np.float32)
    for i, n in enumerate(lens):
        x[i, :n] = base[i % 3][:n]
        x[i, n:] = np.nan                       # whatever lies behind a read in its row must never be read
    xd = torch.from_numpy(x).to(dev)
    ld_dev = torch.tensor(lens, dtype=torch.int32, device=dev)
    probs, logits = net.forward_ragged(xd, ld_dev, return_logits=True)
    probs, logits = probs.cpu().numpy(), logits.cpu().numpy()
    assert np.isfinite(logits).all()
    for i, n in enumerate(lens):
        p1, l1 = net.forward(xd[i: i + 1, :n].contiguous(), return_logits=True)
        assert np.array_equal(l1.cpu().numpy()[0], logits[i]), (i, n)
        assert np.array_equal(p1.cpu().numpy()[0], probs[i]), (i, n)
    # and again in another order with another pitch: a read's bits do not depend on its batch-mates
    order = [9, 0, 7, 3]
    x2 = torch.from_numpy(np.ascontiguousarray(x[order][:, :16000])).to(dev)
    l2 = net.forward_ragged(x2, torch.tensor([lens[i] for i in order], dtype=torch.int32, device=dev), return_logits=True)[1].cpu().numpy()
    assert np.array_equal(l2, logits[order])
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("block", ["basic", "bottleneck"])
def test_resnet_ragged_batch_beyond_the_buffer_window_is_split_not_wrong(block):
    """a ragged forward whose fused launches do not fit their 2 GiB buffer windows used to fall through to the
    unfused kernels - which treat every read as `ld` samples long - and returned wrong probabilities with RS_OK.  With the
    window forced small (RS_SEQ_WINDOW_BYTES at create): the library refuses the call (RS_ERR_ARG, "split the batch"),
    rs_seqnet_max_batch says what fits, and SeqNet.forward_ragged / Model split the batch - every read still gets the bits of
    a call on it alone.  Lengths outside [0, ld] are clamped, a read below the network's minimum is NaN, nothing is read out
    of range."""
    import torch
    from riser_amd import _native as nv
    from riser_amd.resnet import ResNetModel
    dev = torch.device("cuda", 0)
    cfg = dict(synth.RESNET_BENCH_CFG) if block == "basic" else BOTTLENECK_WIDE_CFG
    sd = synth.make_resnet_state_dict(7, cfg)
    config = types.SimpleNamespace(resnet=types.SimpleNamespace(**cfg))
    ref = ResNetModel(sd, config, None, "x", device=dev)
    per_read = (0x7fffffff - 4096) // ref._net.max_batch(9000)      # bytes of one 9000-sample read's largest buffer (about)
    os.environ["RS_SEQ_WINDOW_BYTES"] = str(5 * per_read + 8192)   # a window of five such reads
    try:
        small = ResNetModel(sd, config, None, "x", device=dev)
    finally:
        os.environ.pop("RS_SEQ_WINDOW_BYTES", None)
    lens = [4096, 9000, 8615, 300, 6024, 7999, 5000, 4097, 8999, 8000, 4444, 9000, 6000, 7000, 8000, 5555]
    ld = 9000
    base = _inputs(ld)
    x = np.full((len(lens), ld), np.nan, dtype=np.float32)
    for i, n in enumerate(lens):
        x[i, :n] = base[i % 3][:n]
    xd = torch.from_numpy(x).to(dev)
    ln = torch.tensor(lens, dtype=torch.int32, device=dev)
    want = ref._net.forward_ragged(xd, ln, return_logits=True)[1].cpu().numpy()
    mb = small._net.max_batch(ld)
    assert 1 <= mb < len(lens) and ref._net.max_batch(ld) > 1000
    # the raw entry point refuses the whole batch instead of answering wrongly ...
    net = small._net
    need = nv.lib().rs_seqnet_workspace_bytes(net._h, len(lens), ld)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    probs = torch.empty((len(lens), 2), dtype=torch.float32, device=dev)
    rc = nv.lib().rs_seqnet_forward_ragged(net._h, xd.data_ptr(), ln.data_ptr(), len(lens), ld, ws.data_ptr(), ws.numel(),
                                           probs.data_ptr(), None, None)
    assert rc == nv.RS_ERR_ARG and "split the batch" in nv.lib().rs_last_error().decode()
    # ... and the wrappers split it: same bits as the unrestricted model
    got = net.forward_ragged(xd, ln, return_logits=True)[1].cpu().numpy()
    assert np.array_equal(got, want)
    # lengths beyond the pitch are clamped to it, a read below the program's minimum is NaN (defined, nothing out of range)
    odd = torch.tensor([ld + 5000, 3, -7, 9000], dtype=torch.int32, device=dev)
    xo = torch.from_numpy(np.ascontiguousarray(np.stack([base[0], base[1], base[2], base[0]])[:, :ld])).to(dev)
    lo = ref._net.forward_ragged(xo, odd, return_logits=True)[1].cpu().numpy()
    assert np.array_equal(lo[0], lo[3]) and np.isnan(lo[1]).all() and np.isnan(lo[2]).all()
    ref.close()
    small.close()
