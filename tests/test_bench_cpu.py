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
