import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def hooked_model(env, state, dtype, device, target="m", config=None):
    """A Model created while the RS_* tuning / diagnostic variables of `env` are set: the library reads them
    once, in rs_model_create, never on the launch path."""
    from riser_amd import synth
    from riser_amd.model import Model
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return Model(state, config or synth.Config(), None, target, dtype=dtype, device=device)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


_ORACLE_512 = {}


def oracle_bench_batch(seed: int, pattern: str = "full"):
    """Oracle (oracle/torch_path.py: numpy normalise + torch-CPU conv stack) probabilities of ALL 512 reads of the
    bench batch (synth.make_signals(20260103, 512, 16000)) for weight seed `seed`; pattern "full" = 16000 samples each,
    "mixed" = 8000 / 12000 / 16000 by read index mod 3.  Computed once per session (a few seconds)."""
    key = (seed, pattern)
    if key not in _ORACLE_512:
        import numpy as np
        import torch
        from oracle import torch_path
        from riser_amd import synth
        sigs = synth.make_signals(20260103, 512, 16000)
        lens = [16000] * 512 if pattern == "full" else [(8000, 12000, 16000)[i % 3] for i in range(512)]
        old = torch.get_num_threads()
        torch.set_num_threads(min(32, max(1, (os.cpu_count() or 8))))
        try:
            _ORACLE_512[key] = torch_path.classify_batched(torch_path.TorchCpuModel(synth.make_state_dict(seed)), sigs, lens)
        finally:
            torch.set_num_threads(old)
    return _ORACLE_512[key]
