"""Replay harness: scripted ReadUntil batches through the batched SequencerControl.

The reference can only be exercised end to end against MinKNOW (live, or in playback of a bulk
fast5: README.md:85-113).  This module scripts the same traffic for the FakeClient: `channels`
reads per ReadUntil batch, drawn from a pool of synthetic raw reads (adapter + poly(A) plateau +
RNA squiggle) whose ids repeat over consecutive batches, as an AccumulatingCache client re-sends a
read until it is decided (riser/client.py:29-31).  `run_replay` measures what the operator sees:
host wall time per batch from get_read_batch() to the reject / finish calls
(riser/control.py:31-106), which must stay well inside the 1 s ReadUntil decision window.

Used by tools/replay_bench.py and by bench.py's `control_loop` object.
"""
from __future__ import annotations

import logging
import os
import tempfile
import time

import numpy as np
import torch

from . import synth
from .fake_client import FakeClient, FakeRead


def scripted_batches(n_batches: int, channels: int = 512, seed: int = 4242, min_len: int = 5000,
                     max_len: int = 22000, pool_reads: int | None = None):
    """-> list of n_batches lists of (channel, FakeRead)."""
    rng = np.random.default_rng(7)
    pool_reads = pool_reads or channels * 2
    pool = []
    for rid in range(pool_reads):
        n = int(rng.integers(min_len, max_len))
        pool.append(synth.make_raw_read(seed, rid, n, polya=(rid % 5 != 0)))
    batches = []
    for b in range(n_batches):
        reads = []
        for ch in range(channels):
            rid = (b * 37 + ch) % len(pool)
            reads.append((ch + 1, FakeRead(f"read-{b // 4}-{rid}", pool[rid])))
        batches.append(reads)
    return batches


def run_replay(models, processor, batches, mode: str = "enrich", threshold: float = 0.9, skip: int = 3) -> dict:
    """Drive SequencerControl.target over `batches`; returns counts and per-batch latency percentiles
    (the first `skip` batches are warm-up: workspace allocation, first launches)."""
    from .control import SequencerControl
    client = FakeClient(batches)
    with tempfile.TemporaryDirectory() as d:
        ctl = SequencerControl(client, models, processor, logging.getLogger("riser_amd.replay"),
                               os.path.join(d, "out"))
        ctl.start()
        t0 = time.perf_counter()
        ctl.target(mode, 1.0, threshold)
        torch.cuda.synchronize(processor.device)
        wall = time.perf_counter() - t0
        ctl.finish()
        with open(os.path.join(d, "out.csv")) as f:
            rows = sum(1 for _ in f) - 1
    lat = np.asarray(list(ctl.batch_latencies)[skip:], dtype=np.float64) * 1e3
    if lat.size == 0:
        lat = np.zeros(1)
    received = sum(len(b) for b in batches)
    return {"batches": len(batches), "reads_received": received, "reads_assessed": rows,
            "assessed_per_batch": round(rows / max(len(batches), 1), 1),
            "wall_s": round(wall, 3), "assessed_per_s": round(rows / wall, 1),
            "p50_ms": round(float(np.percentile(lat, 50)), 3), "p99_ms": round(float(np.percentile(lat, 99)), 3),
            "max_ms": round(float(lat.max()), 3),
            "rejected": sum(len(r) for r in client.rejected), "finished": sum(len(r) for r in client.finished)}
