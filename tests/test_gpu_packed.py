"""Packed block layout: a read occupies len / U + 1 blocks of U samples, so the conv stack's work follows
every read's own length.  Lengths drawn uniformly from the live range against the oracle; packed == uniform pitch bit for
bit in every arithmetic mode; batch composition and order never change a read's bits."""
import numpy as np
import pytest
import torch

from oracle import riser_oracle as ro
from oracle import torch_path
from riser_amd import synth
from riser_amd.model import Model, classify_raw_ensemble
from riser_amd.preprocess import pack_reads

pytestmark = pytest.mark.gpu
SIG_SEED = 20260103
_models = {}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def get_model(seed, dev, dtype="f32w"):
    key = (seed, dtype)
    if key not in _models:
        _models[key] = Model(synth.make_state_dict(seed), synth.Config(), None, "t", dtype=dtype, device=dev)
    return _models[key]


def _reads(lens, first=3000):
    return [synth.make_signals(SIG_SEED, 1, int(n), first_read=first + i)[0] for i, n in enumerate(lens)]


def test_uniform_random_lengths_vs_oracle(dev):
    """96 reads with lengths uniform in [4096, 16000] (any integer: riser/control.py:55-60) through the fused path, every
    read against the oracle's torch-CPU path; labels at 0.9 identical."""
    rng = np.random.default_rng(31)
    lens = rng.integers(4096, 16001, size=96)
    lens[:6] = (4096, 4097, 8191, 8192, 8193, 16000)              # block edges: len % U == 0 leaves a whole dead block
    sigs = _reads(lens)
    sig, off, ln, lh = pack_reads(sigs, dev)
    sd = synth.make_state_dict(1)
    got = get_model(1, dev).classify_raw(sig, off, ln, lh).cpu().numpy()
    cpu = torch_path.TorchCpuModel(sd)
    want = torch_path.classify_per_read(cpu, sigs)
    assert np.abs(got - want).max() < 1e-4, np.abs(got - want).max()
    assert np.array_equal(got[:, 1] > 0.9, want[:, 1] > 0.9)
    # block bookkeeping the host mirrors: bases are the prefix sum of len // U + 1
    m = get_model(1, dev)
    U = m.block_samples()
    assert U == 4096 and m.block_bases(lh)[-1] == int((lh // U + 1).sum())
    assert m.block_samples(3) == 1024 and m.block_bases(lh, 3)[-1] == int((lh // 1024 + 1).sum())


@pytest.mark.parametrize("dtype", ["f32w", "f32", "f16", "bf16", "bf16x3", "f16x3", "f16xf8"])
def test_packed_equals_uniform_pitch_bitwise(dev, dtype):
    """the same batch with and without the host's copy of the lengths (packed blocks vs every read in the slot of the
    longest): identical bits, in every mode, through classify_raw and the unfused rs_forward path"""
    lens = np.array([16000, 4096, 8615, 12000, 8000, 4100, 16000, 9999, 12288, 8192, 5000, 15999, 7000], dtype=np.int32)
    sigs = _reads(lens, 3200)
    sig, off, ln, lh = pack_reads(sigs, dev)
    m = get_model(2, dev, dtype)
    packed = m.classify_raw(sig, off, ln, lh).cpu().numpy()
    uniform = m.classify_raw(sig, off, ln, lh, packed=False).cpu().numpy()
    assert np.array_equal(packed, uniform), np.abs(packed - uniform).max()
    # one read alone, the batch reversed, a sub-batch: same bits per read
    for idx in ([2], list(range(len(lens)))[::-1], [4, 0, 7]):
        s2, o2, l2, h2 = pack_reads([sigs[i] for i in idx], dev)
        part = m.classify_raw(s2, o2, l2, h2).cpu().numpy()
        assert np.array_equal(part, packed[idx]), (dtype, idx)
    if dtype in ("f32w", "f32"):
        # unfused route: conv0 kernel on per-read rows + the tiled kernels (fp32: same fmaf chains, same bits)
        xs = [ro.mad_normalise(s) for s in sigs]
        unf = m.classify_batch(xs).cpu().numpy()
        assert np.array_equal(unf, packed), np.abs(unf - packed).max()


@pytest.mark.parametrize("dtype", ["f32w", "f32", "f16", "bf16", "bf16x3", "f16x3", "f16xf8"])
def test_two_level_layout_equals_one_level_bitwise(dev, dtype):
    """round 4: conv layers 0-8 run on fine blocks of 1024 samples, layers 9-11 and the head on 4096-sample blocks behind a
    re-pack of layer 8's output (a live 8615-sample read occupies 9216 samples of rows in the early layers instead of
    12288).  RS_ONE_LEVEL=1 keeps every layer on the coarse blocks - the round-3 layout.  Identical bits in every mode, with and
    without the host's lengths, through the fused path, rs_forward and the ensemble entry; lengths at every block edge of
    either size."""
    from conftest import hooked_model
    from riser_amd.model import classify_raw_ensemble
    lens = np.array([8615, 4096, 4097, 5119, 5120, 5121, 8191, 8192, 8193, 9215, 9216, 12287, 12288, 16000, 6024, 7168, 10240,
                     16383, 4607], dtype=np.int32)
    sigs = _reads(lens, 5200)
    sig, off, ln, lh = pack_reads(sigs, dev)
    two = get_model(2, dev, dtype)
    one = hooked_model({"RS_ONE_LEVEL": "1"}, synth.make_state_dict(2), dtype, dev)
    info2, info1 = two.layer_info(), one.layer_info()
    assert [info2[i]["block_samples"] for i in (1, 8, 9, 11)] == [1024, 1024, 4096, 4096]
    assert {info1[i]["block_samples"] for i in range(12)} == {4096}
    want = one.classify_raw(sig, off, ln, lh, return_logits=True)
    got = two.classify_raw(sig, off, ln, lh, return_logits=True)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), (dtype, (got[0] - want[0]).abs().max().item())
    assert torch.equal(two.classify_raw(sig, off, ln, lh, packed=False), want[0])          # no host lengths: uniform slots
    if dtype in ("f32w", "f32"):
        xs = [ro.mad_normalise(s) for s in sigs]
        assert torch.equal(two.classify_batch(xs), one.classify_batch(xs))                 # rs_forward: plan kernel, conv0 kernel
    dec2 = torch.empty(len(lens), dtype=torch.uint8, device=dev)
    dec1 = torch.empty(len(lens), dtype=torch.uint8, device=dev)
    e2 = classify_raw_ensemble([two, two], sig, off, ln, lh, decision=dec2, max_len=16000)
    e1 = classify_raw_ensemble([one, one], sig, off, ln, lh, decision=dec1, max_len=16000)
    assert torch.equal(e2, e1) and torch.equal(dec2, dec1)
    one.close()


def test_ensemble_on_packed_blocks(dev):
    """rs_classify_ensemble shares one block table and one normalised copy between the models"""
    lens = np.array([8615] * 20 + [4096, 12048, 6000], dtype=np.int32)
    sigs = _reads(lens, 3400)
    sig, off, ln, lh = pack_reads(sigs, dev)
    models = [get_model(s, dev) for s in (1, 2, 3)]
    dec = torch.empty(len(lens), dtype=torch.uint8, device=dev)
    pe = classify_raw_ensemble(models, sig, off, ln, lh, decision=dec, max_len=8615, threshold=0.9).cpu().numpy()
    for k, m in enumerate(models):
        assert np.array_equal(pe[k], m.classify_raw(sig, off, ln, lh).cpu().numpy())
    want = ro.classify_reads(synth.make_state_dict(2), sigs)
    assert np.abs(pe[1] - want).max() < 1e-4


def test_length_mismatch_is_contained(dev):
    """host lengths that understate the device's (a caller bug) cost those reads their result - NaN - and nothing else:
    the other reads keep their bits, nothing is written outside the workspace"""
    lens = np.array([5000, 9000, 4500, 16000], dtype=np.int32)
    sigs = _reads(lens, 3500)
    sig, off, ln, lh = pack_reads(sigs, dev)
    m = get_model(1, dev)
    good = m.classify_raw(sig, off, ln, lh).cpu().numpy()
    lied = lh.copy()
    lied[3] = 4096                                                # device says 16000 (4 blocks), host says 1 block
    out = m.classify_raw(sig, off, ln, lied).cpu().numpy()
    assert np.array_equal(out[:3], good[:3])
    assert np.isnan(out[3]).all()
    assert np.array_equal(m.classify_raw(sig, off, ln, lh).cpu().numpy(), good)


@pytest.mark.parametrize("dtype", ["f16", "bf16", "f16x3", "bf16x3"])
def test_weights_resident_kernel_equals_ring_kernel_bitwise(dev, dtype):
    """conv_wres_h16.hip (the narrow tiled 16-bit layers with the layer's whole weight tensor resident in LDS: layer 3 of
    the shipped net in every 16-bit mode, layer 4 in the plain ones) issues the same MFMA sequence per accumulator as the
    LDS-DMA ring kernel: RS_H16_WRES=0 (ring kernel everywhere) gives identical bits, on a mixed-length batch and on the
    full 512 x 16000 one."""
    from conftest import hooked_model
    sd = synth.make_state_dict(1)
    m = get_model(1, dev, dtype)
    ring = hooked_model({"RS_H16_WRES": "0"}, sd, dtype, dev)
    lens = np.array([16000, 4096, 8615, 12000, 8000, 4100, 9999, 12288, 8192, 5000, 15999], dtype=np.int32)
    sig, off, ln, lh = pack_reads(_reads(lens, 3600), dev)
    a, b = m.classify_raw(sig, off, ln, lh).cpu().numpy(), ring.classify_raw(sig, off, ln, lh).cpu().numpy()
    assert np.array_equal(a, b), np.abs(a - b).max()
    sigs = synth.make_signals(SIG_SEED, 512, 16000)
    sig, off, ln, lh = pack_reads(list(sigs), dev)
    a, b = m.classify_raw(sig, off, ln, lh).cpu().numpy(), ring.classify_raw(sig, off, ln, lh).cpu().numpy()
    assert np.array_equal(a, b), np.abs(a - b).max()
    info = m.layer_info()
    assert info[3]["bn"] == 80 and info[3]["bm"] == 256          # layer 3 ran the weights-resident tile
    ring.close()


def test_edge_lengths_and_oversized_batches(dev):
    """the longest read the normalise kernel stages (65536 samples = 17 blocks), the shortest the net accepts (4096), a
    batch larger than one library call may address (Model.max_batch: split transparently) - all against the oracle or
    against the same reads classified alone"""
    sd = synth.make_state_dict(3)
    m = get_model(3, dev)
    cpu = torch_path.TorchCpuModel(sd)
    lens = np.array([65536, 4096, 65535, 32768, 4097], dtype=np.int32)
    sigs = _reads(lens, 3700)
    sig, off, ln, lh = pack_reads(sigs, dev)
    got = m.classify_raw(sig, off, ln, lh).cpu().numpy()
    want = torch_path.classify_per_read(cpu, sigs)
    assert np.abs(got - want).max() < 1e-4, np.abs(got - want).max()
    assert np.array_equal(got, m.classify_raw(sig, off, ln, lh, packed=False).cpu().numpy())
    # 16-bit split precision on the same reads (the 17-block read crosses every tile boundary of every layer)
    got3 = get_model(3, dev, "f16x3").classify_raw(sig, off, ln, lh).cpu().numpy()
    assert np.abs(got3 - want).max() < 1e-3
    # more reads than one call addresses: 64 distinct 16000-sample signals repeated
    L = 16000
    n = m.max_batch(L) + 300
    base = synth.make_signals(SIG_SEED, 64, L, first_read=5000)
    idx = np.arange(n) % 64
    big = torch.from_numpy(np.ascontiguousarray(base[idx].reshape(-1))).to(dev)
    offs = torch.arange(n, dtype=torch.int64, device=dev) * L
    lens_h = np.full(n, L, dtype=np.int32)
    lens_h[::7] = 9000                                             # mixed lengths inside the oversized batch
    lens_d = torch.from_numpy(lens_h).to(dev)
    p = m.classify_raw(big, offs, lens_d, lens_h).cpu().numpy()
    first = {}
    for i in range(n):                                             # a read's bits depend on (signal, length) only
        key = (int(idx[i]), int(lens_h[i]))
        if key in first:
            assert np.array_equal(p[i], p[first[key]]), (i, key)
        else:
            first[key] = i
    pick = [0, 7, 63, n - 1]
    want = torch_path.classify_batched(cpu, [base[idx[i]] for i in pick], [int(lens_h[i]) for i in pick])
    assert np.abs(p[pick] - want).max() < 1e-4


def test_forward_path_block_plan_beyond_one_scan_chunk(dev):
    """rs_forward (signals that arrive normalised) builds the block table with plan_kernel, a single workgroup scanning
    1024 reads per pass: 1500 short reads exercise the carry between passes; every read must match its solo result bit for
    bit, with and without the host's lengths"""
    m = get_model(2, dev)
    base = [ro.mad_normalise(s).astype(np.float32) for s in _reads([4096, 4100, 5000, 8192, 4097], 3900)]
    n = 1500
    lens = np.array([len(base[i % 5]) for i in range(n)], dtype=np.int32)
    x = np.zeros((n, int(lens.max())), dtype=np.float32)
    for i in range(n):
        x[i, : lens[i]] = base[i % 5]
    xd = torch.from_numpy(x).to(dev)
    got = m.forward_batch(xd, lens).cpu().numpy()
    solo = np.stack([m.classify(b).cpu().numpy() for b in base])
    assert np.array_equal(got, solo[np.arange(n) % 5])
    # the uniform-pitch form of the same call (no host lengths)
    import ctypes as C
    from riser_amd import _native as nv
    L = nv.lib()
    ld = torch.from_numpy(lens).to(dev)
    ws = torch.empty(L.rs_workspace_bytes(m._h, n, int(lens.max())), dtype=torch.uint8, device=dev)
    out = torch.empty((n, 2), dtype=torch.float32, device=dev)
    nv.check(L.rs_forward(m._h, xd.data_ptr(), x.shape[1], ld.data_ptr(), None, n, int(lens.min()), int(lens.max()),
                          ws.data_ptr(), ws.numel(), out.data_ptr(), None, None), "rs_forward")
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), got)
