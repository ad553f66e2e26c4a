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
