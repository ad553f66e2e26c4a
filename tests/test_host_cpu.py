"""CPU-side checks of the control loop's host pieces: the C loops of riser_amd/_hostpack against their Python
equivalents, the FakeClient's channel range, the channel partition of the sharded launcher and the launcher itself
(world size 2, stub step: no GPU), and the property the poly(A) cache rests on (oracle)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import riser_oracle as ro
from riser_amd import synth
from riser_amd.fake_client import FakeClient, FakeRead, PlainFakeClient
from riser_amd.launch import rank_channel_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ("try_again", "accept", "reject", "no_decision")


@pytest.fixture(scope="module")
def hp():
    from riser_amd import build
    build.build_hostpack()
    from riser_amd import _hostpack
    return _hostpack


def _reads(n, rng, lo=0, hi=400):
    return [FakeRead(f"id-{i}", rng.integers(-300, 4000, size=int(rng.integers(lo, hi)), dtype=np.int16)) for i in range(n)]


def test_hostpack_lengths_and_gather(hp):
    rng = np.random.default_rng(3)
    reads = _reads(70, rng)
    reads.append(FakeRead("mv", memoryview(np.arange(90, dtype=np.int16).tobytes())[20:120]))    # a slice of a shared buffer
    lens = np.empty(len(reads), dtype=np.int64)
    hp.lengths(reads, lens)
    assert lens.tolist() == [len(r.raw_data) // 2 for r in reads]
    for start in (np.zeros(len(reads), dtype=np.int64), np.minimum(lens, rng.integers(0, 60, size=len(reads)))):
        out = np.full(int((lens - start).sum()) + 7, -1, dtype=np.int16)
        n = hp.gather(reads, start, out)
        want = np.concatenate([np.frombuffer(r.raw_data, np.int16)[s:] for r, s in zip(reads, start.tolist())])
        assert n == want.size and np.array_equal(out[:n], want) and (out[n:] == -1).all()
    with pytest.raises(ValueError):
        hp.gather(reads, np.zeros(len(reads), dtype=np.int64), np.zeros(5, dtype=np.int16))          # staging too small
    with pytest.raises(ValueError):
        hp.gather(reads, lens + 1, np.zeros(10, dtype=np.int16))                                      # start behind the end
    with pytest.raises(TypeError):
        hp.lengths(tuple(reads), lens)
    with pytest.raises(AttributeError):
        hp.lengths([object()], lens)


def test_hostpack_gather_threaded_path(hp):
    """more than 8 MiB in one call takes the multi-threaded copy: same bytes"""
    rng = np.random.default_rng(5)
    pool = rng.integers(0, 4000, size=6_000_000, dtype=np.int16)
    mv = memoryview(pool.tobytes())
    reads = [FakeRead(f"r{i}", mv[2 * i * 9000: 2 * (i * 9000 + 8000 + 37 * (i % 11))]) for i in range(640)]
    lens = np.empty(len(reads), dtype=np.int64)
    hp.lengths(reads, lens)
    start = (np.arange(len(reads)) % 5) * 100
    out = np.empty(int((lens - start).sum()), dtype=np.int16)
    assert out.nbytes > (8 << 20)
    assert hp.gather(reads, start.astype(np.int64), out) == out.size
    want = np.concatenate([np.frombuffer(r.raw_data, np.int16)[s:] for r, s in zip(reads, start.tolist())])
    assert np.array_equal(out, want)


def test_hostpack_csv_rows_match_the_reference_format(hp):
    """riser/control.py:145-153 prints p.item() of an fp32 probability with str(): the repr of the widened double"""
    rng = np.random.default_rng(9)
    reads = _reads(40, rng)
    reads[5].id = 12345                                           # minknow-api <= 5 style numeric ids still print
    sel = np.array([0, 5, 17, 39], dtype=np.int64)
    chan = np.array([1, 512, 3000, 18000], dtype=np.int64)
    ns = np.array([4096, 8615, 12048, 5000], dtype=np.int32)
    p32 = np.array([[0.5, 1e-5, 1.0], [0.9, 0.90000004, 0.0], [1e-30, 0.123456789, 3.0e-7], [0.99999994, 1e-45, 0.25]],
                   dtype=np.float32)
    p = np.ascontiguousarray(p32, dtype=np.float64)
    dec = np.array([0, 2, 3, 1], dtype=np.uint8)
    got = hp.format_rows("1700000000,", reads, sel, chan, ns, ",mRNA;mtRNA;globin,", p, 3, ",0.9,enrich,", dec, NAMES)
    want = "".join(f"1700000000,{reads[i].id},{c},{n},mRNA;mtRNA;globin,{';'.join(str(float(q)) for q in row)},0.9,enrich,{NAMES[d]}\n"
                   for i, c, n, row, d in zip(sel.tolist(), chan.tolist(), ns.tolist(), p32, dec.tolist()))
    assert got == want
    assert hp.format_rows("h,", reads, sel[:0], chan[:0], ns[:0], ",m,", p[:0], 3, ",t,", dec[:0], NAMES) == ""
    with pytest.raises(ValueError):
        hp.format_rows("h,", reads, np.array([40], dtype=np.int64), chan[:1], ns[:1], ",m,", p[:1], 3, ",t,", dec[:1], NAMES)


def test_hostpack_attrs_and_cache_lookup(hp):
    """the read ids of a batch and the poly(A) cache's look-ups as C loops: same values as the comprehensions"""
    rng = np.random.default_rng(11)
    reads = _reads(50, rng)
    assert hp.attrs(reads, "id") == [r.id for r in reads]
    with pytest.raises(AttributeError):
        hp.attrs(reads, "no_such_attribute")
    entries = [(7 + 3 * i, r) for i, r in enumerate(reads)]
    channels = np.empty(len(entries), dtype=np.int64)
    got = hp.unpack(entries, channels)
    assert channels.tolist() == [e[0] for e in entries] and all(a is b for a, b in zip(got, reads))
    assert hp.unpack([[1, reads[0]]], channels[:1])[0] is reads[0]
    with pytest.raises(TypeError):
        hp.unpack([(1, reads[0], 2)], channels)
    with pytest.raises(ValueError):
        hp.unpack(entries, np.empty(3, dtype=np.int64))
    ids = np.empty(len(reads), dtype=object)
    ids[:] = [r.id for r in reads]
    cache = {r.id: int(rng.integers(1, 9000)) for r in reads[::3]}
    want = np.array([cache.get(i, 0) for i in ids], dtype=np.int64)
    for keys in (ids, ids[7:31], list(ids), tuple(ids[:5])):          # object array (read in place), a view, sequences
        got = np.full(len(keys), -1, dtype=np.int64)
        hp.lookup(cache, keys, got)
        lo = 7 if len(keys) == 24 else 0
        assert got.tolist() == want[lo: lo + len(keys)].tolist()
    with pytest.raises(ValueError):
        hp.lookup(cache, ids, np.zeros(3, dtype=np.int64))
    with pytest.raises(TypeError):
        hp.lookup({"id-0": "not a number"}, ids[:1], np.zeros(1, dtype=np.int64))


def test_fake_client_channel_range_and_flags():
    batches = [[(ch, FakeRead(f"r{ch}", np.zeros(10, dtype=np.int16))) for ch in range(1, 13)] for _ in range(2)]
    c = FakeClient(batches, first_channel=5, last_channel=8)
    assert [ch for ch, _ in c.get_read_batch()] == [5, 6, 7, 8]
    assert len(FakeClient(batches).get_read_batch()) == 12
    assert np.dtype(FakeClient.raw_data_dtype) == np.int16 and PlainFakeClient.raw_data_dtype is None


def test_rank_channel_ranges_partition_the_flow_cell():
    for channels, world in ((512, 1), (512, 8), (18000 * 8, 8), (3000, 7), (12, 2), (9, 9)):
        ranges = [rank_channel_range(r, world, channels) for r in range(world)]
        assert ranges[0][0] == 1 and ranges[-1][1] == channels
        assert all(ranges[i][1] + 1 == ranges[i + 1][0] for i in range(world - 1))
        sizes = [b - a + 1 for a, b in ranges]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        rank_channel_range(0, 8, 4)


def _launch(args, cwd):
    r = subprocess.run([sys.executable, "-m", "riser_amd.launch", *args], cwd=cwd, capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")} |
                           {"PYTHONPATH": ROOT})
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_launcher_two_ranks_stub(tmp_path):
    """python -m riser_amd.launch --gpus 2 (stub step): two fresh children, each on its own channel range, write their own
    CSV; the union of their rows is the single-rank run's, the minute counters are merged"""
    script = os.path.join(ROOT, "tests", "golden", "control.json")
    common = ["--channels", "22", "--stub", "--replay-script", script]
    one = _launch(["--gpus", "1", "--out", str(tmp_path / "one"), *common], ROOT)
    two = _launch(["--gpus", "2", "--out", str(tmp_path / "two"), *common], ROOT)
    assert two["ranks"] == 2 and two["channel_ranges"] == [[1, 11], [12, 22]]

    def rows(path):
        return sorted(ln.split(",", 1)[1] for ln in open(path).read().strip().split("\n")[1:])
    single = rows(str(tmp_path / "one.rank0.csv"))
    parts = [rows(str(tmp_path / f"two.rank{r}.csv")) for r in range(2)]
    assert sorted(parts[0] + parts[1]) == single and len(single) == 25 and all(parts)
    assert all(int(ln.split(",")[1]) <= 11 for ln in parts[0]) and all(int(ln.split(",")[1]) >= 12 for ln in parts[1])
    assert two["reads_received"] == one["reads_received"] == 25
    assert two["minutes_merged"]["0"]["assessed"] == one["minutes_merged"]["0"]["assessed"] == 25
    assert json.load(open(str(tmp_path / "two.summary.json")))["ranks"] == 2


def test_launcher_refuses_more_ranks_than_devices(tmp_path):
    r = subprocess.run([sys.executable, "-m", "riser_amd.launch", "--gpus", "64", "--replay-synthetic", "1", "--out",
                        str(tmp_path / "x")], cwd=ROOT, capture_output=True, text=True, timeout=120,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
    assert r.returncode != 0 and "device(s) visible" in r.stderr


def test_polya_end_is_prefix_stable():
    """what lets the loop cache a read's poly(A) end by id whatever the cache's lifetime (riser/control.py:96-97,
    riser/preprocess.py:87-102): the scan of riser/preprocess.py:42-79 is causal - window i looks at samples before
    i + 500 only - so an end found on a prefix of a read is the end found on every extension of it"""
    checked = 0
    for rid in range(40):
        sig = synth.make_raw_read(31, rid, 26000, polya=(rid % 4 != 0))
        full = ro.polya_end(sig)
        for n in (3000, 5200, 7777, 9000, 12500, 18001, 24000):
            part = ro.polya_end(sig[:n])
            if part is not None:
                assert part == full, (rid, n, part, full)
                checked += 1
    assert checked > 60


def _launch_raw(args, timeout=300):
    return subprocess.run([sys.executable, "-m", "riser_amd.launch", *args], cwd=ROOT, capture_output=True, text=True,
                          timeout=timeout,
                          env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")} |
                              {"PYTHONPATH": ROOT})


def _csv_rows(path):
    return sorted(ln.split(",", 1)[1] for ln in open(path).read().strip().split("\n")[1:] if not ln.startswith("batch_start"))


def test_launcher_eight_ranks_stub(tmp_path):
    """the PromethION deployment shape rehearsed on CPU: eight ranks, eight channel ranges, union of rows = one rank's"""
    script = os.path.join(ROOT, "tests", "golden", "control.json")
    common = ["--channels", "24", "--stub", "--replay-script", script]
    one = _launch(["--gpus", "1", "--out", str(tmp_path / "one"), *common], ROOT)
    eight = _launch(["--gpus", "8", "--out", str(tmp_path / "eight"), *common], ROOT)
    assert eight["ranks"] == 8 and eight["channel_ranges"] == [[3 * r + 1, 3 * r + 3] for r in range(8)]
    parts = [_csv_rows(str(tmp_path / f"eight.rank{r}.csv")) for r in range(8)]
    assert sorted(sum(parts, [])) == _csv_rows(str(tmp_path / "one.rank0.csv"))
    assert eight["reads_received"] == one["reads_received"] == 25 and eight["rank_failures"] == []


def test_launcher_client_factory(tmp_path):
    """--client pkg.module:factory(logger, first_channel, last_channel): the path a real deployment takes"""
    script = os.path.join(ROOT, "tests", "golden", "control.json")
    ref = _launch(["--gpus", "2", "--channels", "22", "--stub", "--replay-script", script, "--out", str(tmp_path / "ref")], ROOT)
    fac = _launch(["--gpus", "2", "--channels", "22", "--stub", "--client", "tests.helpers:make_client", "--out",
                   str(tmp_path / "fac")], ROOT)
    for r in range(2):
        assert _csv_rows(str(tmp_path / f"fac.rank{r}.csv")) == _csv_rows(str(tmp_path / f"ref.rank{r}.csv"))
    assert fac["reads_received"] == ref["reads_received"] == 25


def test_launcher_dead_rank_aborts_the_run_at_once(tmp_path):
    """a rank that dies must be reported when it dies, not after every other rank has finished.  Rank 3
    of 8 exits with code 3; the parent terminates the rest and exits non-zero with rank 3's stderr tail."""
    import time
    t0 = time.monotonic()
    r = _launch_raw(["--gpus", "8", "--channels", "24", "--stub", "--replay-synthetic", "2", "--fail-rank", "3", "--out",
                     str(tmp_path / "x")])
    assert r.returncode != 0 and time.monotonic() - t0 < 40
    assert "rank 3 (channels 10-12) exited with code 3" in r.stderr
    assert "simulated failure" in r.stderr and "riser_amd.launch: rank 3 exited with code 3" in r.stderr


def test_launcher_restarts_a_dead_rank_as_a_fresh_process(tmp_path):
    """--on-rank-failure restart: the dead rank's channel range gets a fresh child; the run completes with every row"""
    script = os.path.join(ROOT, "tests", "golden", "control.json")
    marker = str(tmp_path / "died-once")
    r = _launch_raw(["--gpus", "2", "--channels", "22", "--stub", "--replay-script", script, "--fail-rank", "1",
                     "--fail-once", marker, "--on-rank-failure", "restart", "--out", str(tmp_path / "re")])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert [f["rank"] for f in out["rank_failures"]] == [1] and out["rank_failures"][0]["channels"] == [12, 22]
    assert "started again as a fresh process" in r.stderr
    one = _launch(["--gpus", "1", "--channels", "22", "--stub", "--replay-script", script, "--out", str(tmp_path / "one")], ROOT)
    assert sorted(_csv_rows(str(tmp_path / "re.rank0.csv")) + _csv_rows(str(tmp_path / "re.rank1.csv"))) == \
        _csv_rows(str(tmp_path / "one.rank0.csv"))
    assert out["reads_received"] == one["reads_received"]


def test_model_dir_convention(tmp_path):
    """riser/riser.py:21-42: {target}_config_{kit}_{pore}.yaml / {target}_model_{kit}_{pore}.pth, pore by kit; the YAML of
    the reference's shipped configs parses to the same object with and without PyYAML"""
    from riser_amd import modeldir
    assert modeldir.get_pore_version("RNA002") == "R9.4.1" and modeldir.get_pore_version("RNA004") == "RP4"
    with pytest.raises(Exception, match="Invalid kit"):
        modeldir.get_pore_version("RNA003")
    cfg, pth = modeldir.model_files("model", "mtRNA", "RNA004")
    assert cfg == os.path.join("model", "mtRNA_config_RNA004_RP4.yaml") and pth == os.path.join("model", "mtRNA_model_RNA004_RP4.pth")
    text = ("model: cnn\nbatch_size: 32\nn_epochs: 30\nlearning_rate: 0.0001\n\ncnn:\n  n_layers: 12\n  depth: 1\n"
            "  channels: [20,30,45,67,100,150,225,337,505,757,1135,1702]\n  kernels: [3,3,3,3,3,3,3,3,3,3,3,3]\n"
            "  n_classes: 2\n  classifier: gap_fc # fc / gap_fc / gap\n")
    p = tmp_path / "mRNA_config_RNA004_RP4.yaml"
    p.write_text(text)
    c = modeldir.get_config(str(p))
    flat = modeldir._namespace(modeldir._parse_flat_yaml(text))
    for obj in (c, flat):
        assert obj.model == "cnn" and obj.learning_rate == 0.0001 and obj.cnn.n_layers == 12 and obj.cnn.depth == 1
        assert obj.cnn.channels == list(synth.CHANNELS) and obj.cnn.kernels == [3] * 12 and obj.cnn.classifier == "gap_fc"
    with pytest.raises(FileNotFoundError, match="riser/riser.py:35-42"):
        modeldir.get_models(["globin"], None, "RNA004", str(tmp_path))


def test_csv_double_format_is_python_repr(hp):
    """the rows' probabilities are written as str(p.item()) writes them (riser/control.py:152): repr of the double.  The
    writer thread formats them without the interpreter (csrc/hostpack.c:fmt_double_repr): same text, digit for digit"""
    rng = np.random.default_rng(5)
    vals = list(rng.random(60000, dtype=np.float32).astype(np.float64)) + list((rng.random(30000) ** 9).astype(np.float32).astype(np.float64))
    vals += list(rng.random(20000)) + list(np.exp(rng.uniform(-60, 60, 20000))) + list(rng.random(2000) * 1e-310)
    vals += [0.0, -0.0, 1.0, 0.5, 1e-5, 1e-4, 9.999e-5, 1e15, 1e16, 1e17, 1e22, 9999999999999998.0, 1e16 + 2, 5e-324,
             1.7976931348623157e308, 2.0 ** -149, float("nan"), float("inf"), -float("inf"), 0.1, 1 / 3, 2.0 ** 53 + 2]
    bad = [(v, hp.repr_double(float(v)), repr(float(v))) for v in vals if hp.repr_double(float(v)) != repr(float(v))]
    assert not bad, bad[:5]


def test_hostpack_decided_lists(hp):
    """the reject / finish lists of a batch: (channel, read.number if the read HAS the attribute else read.id) per decision
    code, in batch order (riser/control.py:85-90,137-143) - against the Python loop it replaces"""
    from riser_amd.control import SequencerControl
    rng = np.random.default_rng(8)
    reads = []
    for i in range(300):
        r = FakeRead(f"id-{i}", np.zeros(4, dtype=np.int16), number=(i if i % 5 else None))    # None: no `.number` (minknow-api >= v6)
        if i % 7 == 0:
            r.number = None                    # the attribute exists and is None: the reference still takes it
        reads.append(r)
    sel = np.sort(rng.choice(300, size=180, replace=False)).astype(np.int64)
    chan = rng.integers(1, 513, size=180).astype(np.int64)
    dec = rng.integers(0, 4, size=180).astype(np.uint8)
    got = hp.decided(reads, sel, chan, dec, (2, 1, 3))
    key = SequencerControl._client_key
    for lst, code in zip(got, (2, 1, 3)):
        assert lst == [(int(chan[k]), key(reads[int(sel[k])])) for k in np.flatnonzero(dec == code)]
    assert hp.decided(reads, sel[:0], chan[:0], dec[:0], (2,)) == ([],)
    with pytest.raises(ValueError):
        hp.decided(reads, np.array([999], dtype=np.int64), chan[:1], np.array([2], dtype=np.uint8), (2,))


def test_hostpack_store_slice_against_the_numpy_steps(hp):
    """_hostpack.store_slice = the host side of _SignalStore.update for one slice, in one call: re-seen reads, delta candidates
    verified on the overlap with the row's tail, raw[start:] staged back to back, the rows' lengths / tails / ids updated.
    Against the numpy steps it replaces, over random traffic: new reads, extensions (delta path), same id with other samples
    (a chunk client: overlap mismatch), shorter re-sends, reads too long for a row, reads shorter than the tail."""
    rng = np.random.default_rng(12)
    T, n_rows, pitch = 32, 40, 3000
    row_id = np.full(n_rows, None, dtype=object)
    row_have = np.zeros(n_rows, dtype=np.int64)
    row_tail = np.zeros((n_rows, T), dtype=np.int16)
    ref_id, ref_have, ref_tail = row_id.copy(), row_have.copy(), row_tail.copy()
    held = {}                                                    # row -> the samples the row's read was built from
    for step in range(30):
        B = int(rng.integers(1, n_rows))
        rows = rng.choice(n_rows, size=B, replace=False).astype(np.int64)
        reads, ids = [], np.empty(B, dtype=object)
        for i, r in enumerate(rows.tolist()):
            kind = rng.integers(0, 6)
            if r in held and kind <= 2:                          # the same read again, longer: delta path
                rid, sig = held[r]
                sig = np.concatenate([sig, rng.integers(-500, 3000, size=int(rng.integers(0, 400)), dtype=np.int16)])
            elif r in held and kind == 3:                        # the same id, other samples of the same length or more
                rid, old = held[r]
                sig = rng.integers(-500, 3000, size=old.shape[0] + int(rng.integers(0, 50)), dtype=np.int16)
            elif r in held and kind == 4:                        # the same id, shorter
                rid, old = held[r]
                sig = old[: max(1, old.shape[0] // 2)].copy()
            else:
                rid = f"read-{step}-{r}"
                sig = rng.integers(-500, 3000, size=int(rng.choice([5, 20, 31, 32, 33, 700, 2500, 3500])), dtype=np.int16)
            held[r] = (rid, sig)
            ids[i] = "".join(rid)                                # an equal string, not the same object
            reads.append(FakeRead(ids[i], sig))
        lens = np.array([len(r.raw_data) // 2 for r in reads], dtype=np.int64)
        fits = lens <= pitch
        # ---- the numpy steps of _SignalStore.update ----
        have = ref_have[rows]
        reseen = fits & (ref_id[rows] == ids) & (have > 0)
        cand = reseen & (have <= lens) & (have >= T)
        bad = 0
        for i in np.flatnonzero(cand):
            sig = np.frombuffer(reads[i].raw_data, dtype=np.int16)
            if not np.array_equal(sig[have[i] - T: have[i]], ref_tail[rows[i]]):
                cand[i] = False
                bad += 1
        start_want = np.where(cand, have - T, 0)
        stage_want = np.concatenate([np.frombuffer(r.raw_data, dtype=np.int16)[s:] for r, s in zip(reads, start_want.tolist())])
        for i, r in enumerate(rows.tolist()):
            if fits[i]:
                ref_id[r], ref_have[r] = ids[i], lens[i]
                if lens[i] >= T:
                    ref_tail[r] = np.frombuffer(reads[i].raw_data, dtype=np.int16)[-T:]
            else:
                ref_have[r] = 0
        # ---- the C call ----
        stage = np.zeros(int(lens.sum()) + 8, dtype=np.int16)
        start, cand_c, stats = np.empty(B, dtype=np.int64), np.empty(B, dtype=np.uint8), np.zeros(3, dtype=np.int64)
        total = hp.store_slice(reads, ids, row_id, rows, lens, fits.astype(np.uint8), row_have, row_tail, T, stage, start, cand_c, stats)
        assert total == stage_want.shape[0] and np.array_equal(stage[:total], stage_want)
        assert np.array_equal(start, start_want) and np.array_equal(cand_c.astype(bool), cand)
        assert stats.tolist() == [int(reseen.sum()), int(cand.sum()), bad]
        assert np.array_equal(row_have, ref_have) and np.array_equal(row_tail, ref_tail) and list(row_id) == list(ref_id)
        for r in rows[~fits].tolist():
            held.pop(r, None)
    with pytest.raises(ValueError):
        hp.store_slice(reads, ids, row_id, rows, lens, fits.astype(np.uint8), row_have, row_tail, T, np.zeros(4, dtype=np.int16), start,
                       cand_c, stats)


def test_numa_aware_core_slices_and_launch_wide_clock(tmp_path):
    """supervise.cpu_slices prefers the cores of each rank's GPU's NUMA node when sysfs has the map and
    falls back to contiguous slices when it does not; a restarted rank numbers its minutes from the LAUNCH's start and stops
    at the launch's deadline; an operator's smaller OMP_NUM_THREADS stands."""
    from riser_amd import launch, supervise
    # a fake sysfs: 4 GPUs, cards 1 and 3 on node 1; node 0 = cpus 0-7, node 1 = cpus 8-15; card2 is another vendor's
    root = tmp_path / "sys"
    for k, (vendor, node, pci) in enumerate([("0x1002", 0, "0000:05:00.0"), ("0x1002", 1, "0000:85:00.0"), ("0x10de", 0, "0000:06:00.0"),
                                             ("0x1002", 1, "0000:c5:00.0"), ("0x1002", 0, "0000:45:00.0")]):
        d = root / "devices" / "pci" / pci
        d.mkdir(parents=True)
        (d / "vendor").write_text(vendor + "\n")
        (d / "numa_node").write_text(f"{node}\n")
        c = root / "class" / "drm" / f"card{k}"
        c.mkdir(parents=True)
        (c / "device").symlink_to(d)
    for n, cl in ((0, "0-7"), (1, "8-11,12-15")):
        nd = root / "devices" / "system" / "node" / f"node{n}"
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(cl + "\n")
    topo = supervise.read_topology(str(root))
    assert topo == ([0, 0, 1, 1], {0: list(range(8)), 1: list(range(8, 16))})          # PCI order: 05, 45 (node 0), 85, c5 (node 1)
    sl = supervise.cpu_slices(4, cpus=range(16), topology=topo)
    assert sl == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]
    sl = supervise.cpu_slices(3, cpus=[0, 1, 2, 8, 9, 10, 11], topology=topo)           # the affinity mask cuts node 0 to three cores
    assert sl == [[0, 1], [2], [8, 9, 10, 11]]
    # not every rank has a GPU in the map / a node without cores / no map: contiguous slices as before
    assert supervise.cpu_slices(8, cpus=range(16), topology=topo) == [[2 * r, 2 * r + 1] for r in range(8)]
    assert supervise.cpu_slices(2, cpus=range(4), topology=([0, -1], {0: [0, 1, 2, 3]})) == [[0, 1], [2, 3]]
    assert supervise.read_topology(str(tmp_path / "nothing")) is None
    assert supervise.cpu_slices(2, cpus=range(4)) == [[0, 1], [2, 3]]
    # thread caps: an explicit smaller setting stands, a larger one is cut to the slice
    env = supervise.rank_env(0, 8, base_env={"OMP_NUM_THREADS": "4", "MKL_NUM_THREADS": "999"}, cpus=range(256))
    assert env["OMP_NUM_THREADS"] == "4" and env["MKL_NUM_THREADS"] == "32" and env["OPENBLAS_NUM_THREADS"] == "32"
    # minutes: a rank restarted 47.4 minutes into the launch reports minute 47 first, then 48, 49 (never 0 again)
    relay = launch._MinuteRelay(3, t0=1000.0)
    assert [relay.minute_index(now=1000.0 + 60.0 * m + 2.0) for m in (1, 2, 3)] == [0, 1, 2]
    late = launch._MinuteRelay(3, t0=1000.0)
    assert [late.minute_index(now=1000.0 + 47.4 * 60 + 60.0 * m) for m in (1, 2, 2.01)] == [47, 48, 49]
    # duration: the launch's deadline caps a restarted rank's run
    assert launch.remaining_duration_h(48.0, env={}) == 48.0
    assert abs(launch.remaining_duration_h(48.0, now=1000.0 + 47 * 3600, env={"RS_LAUNCH_DEADLINE": repr(1000.0 + 48 * 3600)}) - 1.0) < 1e-9
    assert launch.remaining_duration_h(48.0, now=5e9, env={"RS_LAUNCH_DEADLINE": "1000.0"}) == 0.0
