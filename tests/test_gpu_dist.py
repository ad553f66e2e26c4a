"""The RCCL (torch "nccl") calls of riser_amd/dist.py on the one GPU of the test box, world size 1: initialisation with a device
id and a bounded timeout, barrier with device ids, float64 MAX / SUM all-reduce, all_gather_object, destroy - what bench.py and
the launcher use at N > 1 (the N = 8 run itself is the driver's)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_calls_of_the_multi_gpu_path_world_1():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29513")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_smoke.py")], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert lines[-1] == "ok" and "max 1.25 sum 2.5" in out.stdout and "gather 6.0" in out.stdout, out.stdout[-1000:]
