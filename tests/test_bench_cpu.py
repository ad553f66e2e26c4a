"""bench.py's rank plumbing without a GPU: `--gpus N` with no WORLD_SIZE must start N child ranks itself (created
before any GPU call) and relay rank 0's single JSON line; mismatches must fail loudly instead of silently
benchmarking one GPU.  The hidden --stub flag swaps the GPU step for a fixed-cost host step over gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, *args], env=env, capture_output=True, text=True, timeout=300)


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--stub", "--steps", "4", "--warmup", "1", "--batch", "512"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1 and out["scaling"] == "weak"
    # whole-job aggregate: both ranks' chunks over the max-over-ranks time
    assert abs(out["value"] - 2 * 512 * 4 / (out["ms_per_step"] * 4e-3)) / out["value"] < 1e-3


def test_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "2", "--stub"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    r = _run(["--gpus", "1", "--stub"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_more_gpus_than_devices_is_an_error_not_a_one_gpu_run():
    import torch
    if torch.cuda.device_count() >= 16:
        return
    r = _run(["--gpus", "16", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "ROCm device(s) visible" in (r.stderr + r.stdout)
    assert not r.stdout.strip()


def test_gpus_8_stub_rehearsal():
    """the driver's N = 8 launch shape, rehearsed over gloo: eight fresh ranks, each pinned to its own core slice, one line"""
    r = _run(["--gpus", "8", "--stub", "--steps", "3", "--warmup", "1", "--batch", "512"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 3
    assert abs(out["value"] - 8 * 512 * 3 / (out["ms_per_step"] * 3e-3)) / out["value"] < 1e-3
    # every rank's own figures (what makes a first real 8-GPU run diagnosable): min / max / argmin of the per-rank rates
    pr = out["per_rank"]
    assert len(pr["ms_per_step"]) == 8 and 0 <= pr["argmin"] < 8 and pr["min"] <= pr["max"]
    assert abs(pr["min"] - 512 / (max(pr["ms_per_step"]) * 1e-3)) / pr["min"] < 1e-2
    assert max(pr["ms_per_step"]) <= out["ms_per_step"] * 1.001          # the headline is the slowest rank's time


def test_a_rank_that_dies_before_the_rendezvous_ends_the_run_at_once():
    """rank 3 of 8 exits before init_process_group: the parent must terminate the seven ranks waiting in the rendezvous
    and exit non-zero with rank 3's stderr tail - not sit in a collective until an outer timeout"""
    import time
    t0 = time.monotonic()
    r = _run(["--gpus", "8", "--stub", "--steps", "3", "--warmup", "1", "--stub-fail-rank", "3"])
    took = time.monotonic() - t0
    assert r.returncode != 0
    assert "rank 3 exited with code 3" in r.stderr and "simulated failure before the rendezvous" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert took < 40, took            # interpreter start-up of eight children dominates; the poll + grace is < 6 s


def test_rank_environment_has_disjoint_core_slices_and_thread_caps():
    from riser_amd import supervise
    sl = supervise.cpu_slices(8, cpus=range(256))
    assert [len(x) for x in sl] == [32] * 8 and sorted(c for x in sl for c in x) == list(range(256))
    sl = supervise.cpu_slices(3, cpus=range(8))
    assert [len(x) for x in sl] == [3, 3, 2] and len({c for x in sl for c in x}) == 8
    assert supervise.cpu_slices(8, cpus=[0, 1]) == [[0, 1]] * 8          # fewer cores than ranks: shared, not empty
    env = supervise.rank_env(2, 8, base_env={}, master_port=1234, cpus=range(256))
    assert env["RANK"] == env["LOCAL_RANK"] == "2" and env["WORLD_SIZE"] == "8" and env["MASTER_ADDR"] == "127.0.0.1"
    assert env["RS_CPU_SLICE"].split(",")[0] == "64" and len(env["RS_CPU_SLICE"].split(",")) == 32
    assert env["OMP_NUM_THREADS"] == "32" and env["RS_HOST_THREADS"] == "8" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
