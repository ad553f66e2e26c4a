"""Parity of the Winograd F(2,3) fp32 conv path (rs_dtype RS_F32W, csrc/conv_wino.hip) through the
C ABI: same bars as the direct fp32 path - probabilities within 1e-3 of the reference's torch-CPU
results (golden fixtures) and of the numpy oracle, accept/reject labels identical at 0.9."""
import os

import numpy as np
import pytest
import torch

from oracle import riser_oracle as ro
from riser_amd import synth

pytestmark = pytest.mark.gpu

PROB_TOL = 1e-3
SIG_SEED = 20260103


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


_models = {}


def get_model(seed, dev, dtype="f32w"):
    from riser_amd.model import Model
    key = (seed, dtype)
    if key not in _models:
        _models[key] = Model(synth.make_state_dict(seed), synth.Config(), None, "mRNA", dtype=dtype, device=dev)
    return _models[key]


@pytest.mark.parametrize("dtype", ["f32w", "f32"])
def test_wino_forward_golden(dev, golden_dir, dtype):
    """both fp32 lowerings (Winograd = the default of Model, and the direct one) against the
    reference's own outputs"""
    net = np.load(os.path.join(golden_dir, "network.npz"))
    worst = 0.0
    for seed, L, B, first in net["cases"]:
        tag = f"s{seed}_L{L}_B{B}_r{first}"
        sigs = synth.make_signals(int(net["sig_seed"][0]), int(B), int(L), first_read=int(first))
        m = get_model(int(seed), dev, dtype)
        xs = [ro.mad_normalise(s) for s in sigs]
        probs, logits = m.classify_batch(xs, return_logits=True)
        probs, logits = probs.cpu().numpy(), logits.cpu().numpy()
        want = net[f"{tag}.probs"]
        err = float(np.abs(probs - want).max())
        worst = max(worst, err)
        assert err < PROB_TOL, (tag, err)
        assert np.array_equal(probs[:, 1] > 0.9, want[:, 1] > 0.9), tag
        assert np.allclose(logits, net[f"{tag}.logits"], atol=2e-3), tag
    print(dtype, "worst |dp| vs reference:", worst)
    assert worst < 1e-4        # observed ~1e-5, the same as the direct fp32 path


def test_wino_mixed_lengths_permutation_and_direct(dev):
    m = get_model(2, dev)
    md = get_model(2, dev, "f32")
    lens = [4096, 4097, 4099, 5000, 6023, 6024, 8191, 8192, 8193, 8615, 11999, 12048, 15999, 16000, 16383, 16384, 20001]
    sigs = [synth.make_signals(SIG_SEED, 1, n, first_read=50 + i)[0] for i, n in enumerate(lens)]
    xs = [ro.mad_normalise(s) for s in sigs]
    sd = synth.make_state_dict(2)
    want = np.stack([ro.classify(sd, x) for x in xs])
    got = m.classify_batch(xs).cpu().numpy()
    assert np.abs(got - want).max() < PROB_TOL
    direct = md.classify_batch(xs).cpu().numpy()
    assert np.abs(got - direct).max() < 1e-4
    perm = np.random.default_rng(0).permutation(len(xs))
    got_p = m.classify_batch([xs[i] for i in perm]).cpu().numpy()
    assert np.array_equal(got_p, got[perm]), "per-read result must not depend on batch position"
    for i in (0, 9, 16):
        assert np.array_equal(m.classify_batch([xs[i]]).cpu().numpy()[0], got[i]), "batch of 1 == batched"


def test_wino_small_custom_network(dev):
    from riser_amd.model import Model
    rng = np.random.default_rng(3)
    channels = [8, 13, 21, 40, 70, 17]
    sd, c_in = {}, 1
    for i, c in enumerate(channels):
        sd[f"layers.{i}.0.weight"] = (rng.standard_normal((c, c_in, 3)) * np.sqrt(2.0 / (3 * c_in))).astype(np.float32)
        sd[f"layers.{i}.0.bias"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        c_in = c
    sd["classifier.2.weight"] = rng.standard_normal((2, c_in)).astype(np.float32)
    sd["classifier.2.bias"] = rng.standard_normal(2).astype(np.float32)
    cfg = synth.Config(synth.CnnConfig(channels=channels, kernels=[3] * 6))
    m = Model(sd, cfg, None, "x", dtype="f32w", device=dev)
    xs = [rng.standard_normal(n).astype(np.float64) for n in (64, 65, 100, 777, 2048)]
    got = m.classify_batch(xs).cpu().numpy()
    want = np.stack([ro.classify(sd, x) for x in xs])
    assert np.abs(got - want).max() < 1e-4
    m.close()


def test_wino_fused_raw_and_full_batch(dev):
    """raw int16 -> probabilities on a 512 x 16000 batch: equal to the direct fp32 path within 1e-4,
    labels identical, and a sample of reads checked against the oracle."""
    from riser_amd.preprocess import pack_reads
    B, L = 512, 16000
    sigs = synth.make_signals(SIG_SEED, B, L)
    sig, off, ln, lens = pack_reads(list(sigs), dev)
    pw = get_model(1, dev).classify_raw(sig, off, ln, lens).cpu().numpy()
    pd = get_model(1, dev, "f32").classify_raw(sig, off, ln, lens).cpu().numpy()
    assert np.abs(pw - pd).max() < 1e-4
    assert np.array_equal(pw[:, 1] > 0.9, pd[:, 1] > 0.9)
    idx = [0, 17, 255, 511]
    want = ro.classify_reads(synth.make_state_dict(1), [sigs[i] for i in idx])
    assert np.abs(pw[idx] - want).max() < PROB_TOL


def test_wino_layer01_variants_bit_identical(dev):
    """layers 0 + 1 of the fp32 Winograd path exist in three forms - conv0 kernel + tiled kernel (rs_forward),
    tiled kernel with layer 0 folded into its staging, and the LDS-free streaming kernel (both on the
    rs_classify path) - which must agree bit for bit on full-length and mixed-length batches."""
    from riser_amd.preprocess import pack_reads
    from conftest import hooked_model
    m = get_model(3, dev)
    # the tiled kernel's fold needs a tile to span at most two blocks: blocks of 4096 samples (one-level layout)
    m_folded = hooked_model({"RS_NO_STREAM_F32": "1", "RS_ONE_LEVEL": "1"}, synth.make_state_dict(3), "f32w", dev)
    m_nostream = hooked_model({"RS_NO_STREAM_F32": "1"}, synth.make_state_dict(3), "f32w", dev)   # fine blocks: conv0 + tiled
    for lens in ([16000] * 5, [4096, 16000, 8615, 5000, 12001, 4097, 16383, 9999, 4100]):
        sigs = [synth.make_signals(SIG_SEED, 1, n, first_read=400 + i)[0] for i, n in enumerate(lens)]
        sig, off, ln, lh = pack_reads(sigs, dev)
        xs = [ro.mad_normalise(s) for s in sigs]
        stream = m.classify_raw(sig, off, ln, lh).cpu().numpy()
        unfused = m.classify_batch(xs).cpu().numpy()
        folded = m_folded.classify_raw(sig, off, ln, lh).cpu().numpy()
        assert np.array_equal(stream, unfused), np.abs(stream - unfused).max()
        assert np.array_equal(folded, unfused), np.abs(folded - unfused).max()
        assert np.array_equal(m_nostream.classify_raw(sig, off, ln, lh).cpu().numpy(), unfused)
    m_folded.close()
    m_nostream.close()


@pytest.mark.parametrize("dtype", ["f32w", "f32"])
def test_layerwise_activations_vs_oracle(dev, dtype):
    """every conv block's output buffer (rs_debug_capture_layer) against the oracle's activations: valid rows
    within fp32 round-off, every padding row of every read's slot exactly zero (the next layer's 'same'
    padding depends on it)."""
    from riser_amd import _native as nv
    from riser_amd.preprocess import pack_reads
    m = get_model(2, dev, dtype)
    sd = synth.make_state_dict(2)
    lens = [16000, 5000, 4097]
    sigs = [synth.make_signals(SIG_SEED, 1, n, first_read=600 + i)[0] for i, n in enumerate(lens)]
    sig, off, ln, lh = pack_reads(sigs, dev)
    want = []
    for s in sigs:
        x = ro.mad_normalise(s).astype(np.float32)[None, :]
        _, layers = ro.convnet_forward(sd, x, acc=np.float64, return_layers=True)
        want.append(layers)
    # packed block layout: read b owns the blocks bases[b] .. bases[b + 1] of U samples, U >> (i + 1) rows each; two levels:
    # fine blocks for the early layers, coarse ones for the last three (the buffer captured is the layer's own output)
    info = m.layer_info()
    assert {info[i]["block_samples"] for i in range(1, 9)} == {1024} and {info[i]["block_samples"] for i in (9, 10, 11)} == {4096}
    for i in range(1, m.n_layers):
        U, bases = m.block_samples(i), m.block_bases(lens, i)
        assert list(np.diff(bases)) == [n // U + 1 for n in lens]
        P_out, cp = U >> (i + 1), info[i]["cp_out"]
        cap = torch.full((int(bases[-1]) * P_out, cp), float("nan"), dtype=torch.float32, device=dev)
        nv.check(nv.lib().rs_debug_capture_layer(m._h, i, cap.data_ptr(), cap.numel() * 4), "capture")
        m.classify_raw(sig, off, ln, lh)
        got_all = cap.cpu().numpy()
        for b, n in enumerate(lens):
            got = got_all[bases[b] * P_out: bases[b + 1] * P_out]
            ref = want[b][i][0].T                                   # [L_out, C]
            L_out, C = ref.shape
            assert L_out == n >> (i + 1)
            scale = max(1.0, float(np.abs(ref).max()))
            assert np.abs(got[:L_out, :C] - ref).max() < 2e-4 * scale, (i, b)
            assert not got[L_out:, :].any(), (i, b)                 # padding rows of the read's blocks
            assert not got[:, C:].any(), (i, b)                     # padding channels
    nv.check(nv.lib().rs_debug_capture_layer(m._h, -1, None, 0), "capture off")


def test_winograd_f43_layers(dev, monkeypatch):
    """the F(4,3) lowering of the wide late layers (csrc/conv_wino4.hip; RS_WINO4 selects the layers when the model
    is created): probabilities within the fp32 bar of the oracle and of the all-F(2,3) model, labels identical,
    on a full-length batch, a mixed-length batch and a single read; activations of an F(4,3) layer within
    round-off of the oracle."""
    from riser_amd import _native as nv
    from riser_amd.model import Model
    from riser_amd.preprocess import pack_reads
    sd = synth.make_state_dict(1)
    monkeypatch.setenv("RS_WINO4", "none")                        # reference model: F(2,3) on every layer
    base = Model(sd, synth.Config(), None, "m", dtype="f32w", device=dev)
    monkeypatch.setenv("RS_WINO4", "4,5,6,7,8,9,10,11")          # more layers than the default (6-11)
    m = Model(sd, synth.Config(), None, "m", dtype="f32w", device=dev)
    monkeypatch.delenv("RS_WINO4")
    for lens in ([16000] * 8, [4096, 16000, 8615, 5000, 12001, 4097, 16383, 9999, 4100], [6024]):
        sigs = [synth.make_signals(SIG_SEED, 1, n, first_read=800 + i)[0] for i, n in enumerate(lens)]
        sig, off, ln, lh = pack_reads(sigs, dev)
        got = m.classify_raw(sig, off, ln, lh).cpu().numpy()
        ref = base.classify_raw(sig, off, ln, lh).cpu().numpy()
        want = ro.classify_reads(sd, sigs)
        assert np.abs(got - want).max() < 1e-4, np.abs(got - want).max()
        assert np.abs(got - ref).max() < 1e-4
        assert np.array_equal(got[:, 1] > 0.9, want[:, 1] > 0.9)
    # one captured F(4,3) layer against the oracle's activations (valid rows, zero padding)
    lens = [16000, 5000]
    sigs = [synth.make_signals(SIG_SEED, 1, n, first_read=820 + i)[0] for i, n in enumerate(lens)]
    sig, off, ln, lh = pack_reads(sigs, dev)
    i = 8
    assert m.block_samples() == 8192                              # F(4,3) on the last layer doubles the (coarse) block
    U, bases = m.block_samples(i), m.block_bases(lens, i)
    assert U == 1024                                              # ... the early layers keep their fine blocks
    P_out, cp = U >> (i + 1), m.layer_info()[i]["cp_out"]
    cap = torch.full((int(bases[-1]) * P_out, cp), float("nan"), dtype=torch.float32, device=dev)
    nv.check(nv.lib().rs_debug_capture_layer(m._h, i, cap.data_ptr(), cap.numel() * 4), "capture")
    m.classify_raw(sig, off, ln, lh)
    got_all = cap.cpu().numpy()
    for b, s in enumerate(sigs):
        got = got_all[bases[b] * P_out: bases[b + 1] * P_out]
        x = ro.mad_normalise(s).astype(np.float32)[None, :]
        _, layers = ro.convnet_forward(sd, x, acc=np.float64, return_layers=True)
        refl = layers[i][0].T
        L_out, C = refl.shape
        assert np.abs(got[:L_out, :C] - refl).max() < 2e-4 * max(1.0, float(np.abs(refl).max()))
        assert not got[L_out:, :].any() and not got[:, C:].any()
    m.close()
    base.close()
