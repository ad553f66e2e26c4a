"""Pin the CPU oracle (oracle/) to outputs of the reference itself.

The fixtures in tests/golden/ were produced by tools/make_golden.py, which imports
/root/reference and records what its own SignalProcessor / Model / SequencerControl
returned.  These tests run without a GPU and without the reference.
"""
import json
import os

import numpy as np
import pytest

from oracle import riser_oracle as ro
from oracle import torch_path
from riser_amd import synth
from riser_amd.fake_client import FakeClient, FakeRead


@pytest.fixture(scope="module")
def norm(golden_dir):
    return np.load(os.path.join(golden_dir, "normalise.npz"))


def test_normalise_bit_exact(norm):
    names = [str(n) for n in norm["names"]]
    assert len(names) >= 20
    for name in names:
        sig, want, stats = norm[f"{name}.sig"], norm[f"{name}.out"], norm[f"{name}.stats"]
        med, mad = ro.median_mad(sig)
        assert med == stats[0] and mad == stats[1], name
        got = ro.mad_normalise(sig)
        assert got.dtype == want.dtype, name                      # int64 zeros when mad == 0
        assert np.array_equal(got, want), name                    # bit-exact float64


def test_normalise_float_inputs_bit_exact(golden_dir):
    """float32 / float64 signals (the retrain path's pA-scaled input): numpy keeps the input's precision, and so must
    the restatement - dtype and every bit of the reference's output"""
    g = np.load(os.path.join(golden_dir, "normalise_float.npz"))
    cases = synth.normalise_float_cases()
    assert [n for n, _ in cases] == [str(n) for n in g["names"]]
    for name, x in cases:
        want = g[f"{name}.out"]
        got = ro.mad_normalise(x.copy())
        assert got.dtype == want.dtype, name
        assert np.array_equal(got, want), (name, float(np.abs(got.astype(np.float64) - want).max()))
    assert g["mad0_constant.float32.out"].dtype == np.int64 and g["synth_6024.float32.out"].dtype == np.float32


def test_normalise_empty_raises():
    with pytest.raises(ValueError):
        ro.mad_normalise(np.zeros(0, dtype=np.int16))


def test_kit_constants():
    # SURVEY 8(a) A1 [probe values from the reference]
    assert ro.kit_max_length("RNA002") == 12048 and ro.kit_max_length("RNA004") == 8615
    assert ro.kit_fixed_trim_length("RNA002") == 6480 and ro.kit_fixed_trim_length("RNA004") == 4633


def test_polya_end(golden_dir):
    cases = np.load(os.path.join(golden_dir, "polya.npz"))["cases"]
    found = 0
    for seed, rid, n, polya, want in cases:
        got = ro.polya_end(synth.make_raw_read(int(seed), int(rid), int(n), bool(polya)))
        assert (-1 if got is None else got) == want
        found += want >= 0
    assert found >= 20


def test_polya_edge_cases(golden_dir):
    """corners of the window rule (plateau that never ends, reads shorter than a window, a rise where the rule is inert,
    a rolling mean of 0, reads beyond 65536 samples): the reference's answers for synth.polya_edge_cases()"""
    import warnings
    g = np.load(os.path.join(golden_dir, "polya.npz"))
    want = dict(zip([str(n) for n in g["edge_names"]], g["edge_ends"].tolist()))
    cases = synth.polya_edge_cases()
    assert [n for n, _ in cases] == list(want)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for name, sig in cases:
            got = ro.polya_end(sig)
            assert (-1 if got is None else got) == want[name], name
    assert want["plateau_never_ends"] == -1 and want["rolling_mean_zero"] > 0 and want["end_beyond_65536"] > 65536


def _variant_cases(golden_dir):
    import json
    g = np.load(os.path.join(golden_dir, "convnet_variants.npz"))
    for name in ("depth2_k5373", "depth1_k7", "depth3_k3"):
        cfg = json.loads(str(g[f"{name}.cfg"]))
        sd = {k[len(name) + 4:]: g[k] for k in g.files if k.startswith(name + ".sd.")}
        yield name, cfg, sd, g[f"{name}.lens"], g[f"{name}.probs"]


def test_convnet_variants_oracle_vs_reference(golden_dir):
    """depth > 1 and kernels other than 3 (riser/nets/cnn.py:17,52-65): the general restatement against the reference's
    Model.classify on reads of 2^n_layers .. 4096 samples"""
    for name, cfg, sd, lens, want in _variant_cases(golden_dir):
        for j, L in enumerate(lens):
            sig = synth.make_signals(20260103, 1, int(L), first_read=60 + j)[0]
            x = ro.mad_normalise(sig).astype(np.float32)[None, :]
            got = ro.softmax(ro.convnet_forward_general(sd, x, cfg["depth"]))[0]
            assert np.abs(got - want[j]).max() < 2e-5, (name, int(L))


def _gap_cases(golden_dir):
    import json
    g = np.load(os.path.join(golden_dir, "gap_head.npz"))
    for name in ("shipped_gap", "depth2_gap"):
        cfg = json.loads(str(g[f"{name}.cfg"]))
        if cfg["depth"] == 1:                        # the shipped architecture: seed-1 weights, Linear head as a 1 x 1 conv
            sd = dict(synth.make_state_dict(1))
            sd["classifier.0.weight"] = sd.pop("classifier.2.weight")[:, :, None].copy()
            sd["classifier.0.bias"] = sd.pop("classifier.2.bias")
        else:
            sd = {k[len(name) + 4:]: g[k] for k in g.files if k.startswith(name + ".sd.")}
        yield name, cfg, sd, {int(L): g[f"{name}.L{L}.probs"] for L in cfg["lens"]}, str(g[f"{name}.classify_error"])


def test_gap_classifier_oracle_vs_reference(golden_dir):
    """the `gap` head (riser/nets/cnn.py:34-38) through the reference's ConvNet.forward on batches of 3 reads"""
    for name, cfg, sd, want, err in _gap_cases(golden_dir):
        assert err == "IndexError"                   # the reference's Model.classify cannot run this head at batch 1
        for j, (L, probs) in enumerate(want.items()):
            sigs = synth.make_signals(20260103, 3, L, first_read=90 + 3 * j)
            x = np.stack([ro.mad_normalise(s) for s in sigs]).astype(np.float32)
            logits = ro.convnet_forward(sd, x) if cfg["depth"] == 1 else ro.convnet_forward_general(sd, x, cfg["depth"])
            assert np.abs(ro.softmax(logits) - probs).max() < 2e-5, (name, L)


def test_fc_classifier_oracle_vs_reference(golden_dir):
    """the `fc` head (riser/nets/cnn.py:22-27) on the 4-layer net it is hard-coded for, against the reference's
    Model.classify on reads of 12048 .. 12063 samples; another length fails in the first Linear, as in the reference"""
    import json
    g = np.load(os.path.join(golden_dir, "fc_head.npz"))
    sd = synth.make_fc_state_dict(1)
    for j, L in enumerate(g["lens"]):
        sig = synth.make_signals(20260103, 1, int(L), first_read=120 + j)[0]
        got = ro.softmax(ro.convnet_forward(sd, ro.mad_normalise(sig).astype(np.float32)[None, :]))[0]
        assert np.abs(got - g["probs"][j]).max() < 2e-5, int(L)
    assert set(json.loads(str(g["errors"])).values()) == {"RuntimeError"}
    with pytest.raises(RuntimeError):
        ro.convnet_forward(sd, np.zeros((1, 12064), dtype=np.float32))


@pytest.fixture(scope="module")
def net(golden_dir):
    return np.load(os.path.join(golden_dir, "network.npz"))


def _signals(net, seed, L, B, first):
    sigs = synth.make_signals(int(net["sig_seed"][0]), B, L, first_read=first)
    crc = net[f"s{seed}_L{L}_B{B}_r{first}.sig_crc"]
    s = sigs.astype(np.int64)
    assert int(s.sum()) == crc[0] and int((s ** 2).sum()) == crc[1], "synthetic signal generator drifted"
    return sigs


def test_network_numpy_oracle(net):
    """numpy fp32 oracle vs reference Model.classify: probs within 2e-5, labels equal."""
    for seed, L, B, first in net["cases"]:
        if B > 8:
            continue
        tag = f"s{seed}_L{L}_B{B}_r{first}"
        sigs = _signals(net, seed, L, B, first)
        sd = synth.make_state_dict(int(seed))
        probs = ro.classify_reads(sd, sigs)
        want = net[f"{tag}.probs"]
        assert np.abs(probs - want).max() < 2e-5, tag
        assert np.array_equal(probs[:, 1] > 0.9, want[:, 1] > 0.9), tag
        # per-layer statistics of read 0
        _, layers = ro.convnet_forward(sd, ro.mad_normalise(sigs[0]).astype(np.float32)[None], return_layers=True)
        st = net[f"{tag}.layer_stats"]
        for i, h in enumerate(layers):
            assert h.shape[1] == st[i, 4] and h.shape[2] == st[i, 5]
            assert abs(float(h.astype(np.float64).sum()) - st[i, 0]) <= 1e-4 * max(1.0, st[i, 1]), (tag, i)


def test_network_torch_port(net):
    """the torch-CPU port used as bench.py's cpu_baseline reproduces the reference."""
    for seed, L, B, first in net["cases"]:
        if B < 32:
            continue
        tag = f"s{seed}_L{L}_B{B}_r{first}"
        sigs = _signals(net, seed, L, B, first)
        m = torch_path.TorchCpuModel(synth.make_state_dict(int(seed)))
        probs = torch_path.classify_per_read(m, sigs)
        want = net[f"{tag}.probs"]
        assert np.abs(probs - want).max() < 1e-5, tag
        assert np.array_equal(probs[:, 1] > 0.9, want[:, 1] > 0.9), tag


def test_probs_are_discriminating(net):
    """guard against degenerate fixtures (SURVEY 8(c) last row)."""
    p = net["s1_L16000_B32_r1000.probs"][:, 1]
    assert (p > 0.9).sum() >= 2 and (p < 0.2).sum() >= 2 and ((p > 0.3) & (p < 0.7)).sum() >= 2


def test_control_loop_oracle(golden_dir):
    """oracle normalise + polyA + classify + decide replays the reference control loop
    (riser/control.py:31-97) row for row."""
    with open(os.path.join(golden_dir, "control.json")) as f:
        g = json.load(f)
    max_len = ro.kit_max_length(g["kit"])
    fixed = ro.kit_fixed_trim_length(g["kit"])
    models = {}
    for run in g["runs"]:
        if run["mode"] != "enrich":
            continue                                   # same probabilities; decisions covered below
        for s in run["seeds"]:
            models.setdefault(s, torch_path.TorchCpuModel(synth.make_state_dict(s)))
    prob_cache = {}
    for run in g["runs"]:
        rows = []
        cache = {}
        for batch in g["script"]:
            for ch, read_id, rid, n, polya, number in batch:
                sig = synth.make_raw_read(g["raw_seed"], rid, n, bool(polya))
                sig, trimmed = ro.trim_polya(sig, read_id, cache)
                if not trimmed:
                    if len(sig) > fixed + max_len:
                        sig = sig[fixed:][:max_len]
                    else:
                        continue
                else:
                    if len(sig) < ro.MIN_INPUT_SIGNALS:
                        continue
                    sig = sig[:max_len]
                key = (read_id, n)
                p_on, p_off = [], []
                for s in run["seeds"]:
                    if (key, s) not in prob_cache:
                        prob_cache[(key, s)] = models[s].classify(ro.mad_normalise(sig)).numpy()
                    p = prob_cache[(key, s)]
                    p_off.append(p[0]); p_on.append(p[1])
                rows.append((read_id, ch, len(sig), [float(p) for p in p_on],
                             ro.decide(p_on, p_off, run["threshold"], run["mode"], len(sig), max_len)))
        assert len(rows) == len(run["rows"])
        for got, want in zip(rows, run["rows"]):
            assert got[0] == want["read_id"] and got[1] == want["channel"] and got[2] == want["sig_length"]
            assert np.allclose(got[3], want["prob_targets"], atol=1e-5)
            assert got[4] == want["decision"], (run["mode"], run["seeds"], run["threshold"], want)


def test_fake_client_contract():
    c = FakeClient([[(1, FakeRead("a", np.arange(10, dtype=np.int16), number=3))], []])
    assert c.is_running()
    (ch, rd), = c.get_read_batch()
    assert ch == 1 and rd.id == "a" and rd.number == 3 and np.array_equal(c.get_raw_signal(rd), np.arange(10))
    c.get_read_batch()
    assert not c.is_running()


def test_winograd_f23_identity_matches_direct_block():
    """the algebra the default fp32 kernels rely on (csrc/conv_wino.hip): a conv3 + ReLU + MaxPool(2,2) block
    evaluated as Winograd F(2,3) over pooled positions - 4 products per input channel and output pair
    instead of 6 - equals the oracle's direct block (float64: to round-off; float32: to a few ulp)."""
    rng = np.random.default_rng(11)
    for (B, C, L, Co) in ((2, 5, 37, 7), (1, 20, 64, 30), (3, 3, 9, 4)):
        x = rng.standard_normal((B, C, L))
        w = rng.standard_normal((Co, C, 3)) / np.sqrt(3 * C)
        b = rng.standard_normal(Co) * 0.1
        Lo = L // 2
        xp = np.zeros((B, C, 2 * Lo + 3))
        xp[:, :, 1:L + 1] = x                                    # xp[j] = x[j - 1]; zeros are the 'same' padding
        d0, d1, d2, d3 = (xp[:, :, k:k + 2 * Lo:2] for k in range(4))      # x[2T-1], x[2T], x[2T+1], x[2T+2]
        v = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
        u = (w[:, :, 0], (w[:, :, 0] + w[:, :, 1] + w[:, :, 2]) / 2, (w[:, :, 0] - w[:, :, 1] + w[:, :, 2]) / 2, w[:, :, 2])
        for dt, tol in ((np.float64, 1e-12), (np.float32, 2e-5)):
            m = [np.matmul(u[j].astype(dt), v[j].astype(dt)) for j in range(4)]
            y0, y1 = m[0] + m[1] + m[2], m[1] - m[2] - m[3]
            got = np.maximum(np.maximum(y0, y1) + b.astype(dt)[None, :, None], 0)
            want = ro.conv_block(x.astype(dt), w.astype(dt), b.astype(dt), acc=dt)
            assert got.shape == want.shape
            assert np.abs(got - want).max() < tol * max(1.0, np.abs(want).max())


def test_batched_oracle_equals_per_read_oracle():
    """oracle/torch_path.classify_batched (what the full-size GPU tests compare all 512 reads with) against the per-read
    structure of riser/control.py:63-69 on mixed lengths: grouping by length changes nothing but torch's blocking."""
    from oracle import torch_path
    from riser_amd import synth
    lens = [4096, 6024, 4096, 8615, 6024, 5000]
    sigs = [synth.make_signals(20260103, 1, n, first_read=900 + i)[0] for i, n in enumerate(lens)]
    cpu = torch_path.TorchCpuModel(synth.make_state_dict(2))
    a = torch_path.classify_per_read(cpu, sigs)
    b = torch_path.classify_batched(cpu, sigs, batch=2)
    assert np.abs(a - b).max() < 2e-6
    padded = np.zeros((len(lens), 9000), dtype=np.int16)
    for i, s in enumerate(sigs):
        padded[i, : len(s)] = s
    assert np.array_equal(torch_path.classify_batched(cpu, padded, lens, batch=2), b)
