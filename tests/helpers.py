"""Importable test doubles: `--client tests.helpers:make_client` is how a deployment names its own client factory
(`factory(logger, first_channel, last_channel)`, riser_amd/launch.py)."""
import json
import os

from riser_amd import synth
from riser_amd.fake_client import FakeClient, FakeRead

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_client(logger, first_channel, last_channel):
    """the scripted traffic of tests/golden/control.json on this rank's channel range, built by the factory itself"""
    with open(os.path.join(ROOT, "tests", "golden", "control.json")) as f:
        g = json.load(f)
    seed = int(g.get("raw_seed", 77))
    batches = [[(ch, FakeRead(rid_s, synth.make_raw_read(seed, rid, n, bool(polya)), number))
                for ch, rid_s, rid, n, polya, number in b] for b in g["script"]]
    logger.info("tests.helpers.make_client: channels %d-%d", first_channel, last_channel)
    return FakeClient(batches, first_channel=first_channel, last_channel=last_channel)
