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

import gc
import logging
import os
import tempfile
import time

import numpy as np
import torch

from . import synth
from .fake_client import FakeClient, FakeRead


SAMPLES_PER_BATCH = 1600          # what a pore adds between two ReadUntil batches: 0.4 s at RNA004's 4 kHz
VISITS = 5                         # batches a read stays in its pore before the next one takes the channel


def scripted_batches(n_batches: int, channels: int = 512, seed: int = 4242, min_len: int = 7000,
                     max_len: int = 22000, pool_reads: int | None = None):
    """-> list of n_batches lists of (channel, FakeRead).

    The traffic of an AccumulatingCache client (riser/client.py:29-31, `get_read_chunks(last=True)`): every channel
    holds one read at a time and re-sends it WHOLE with every batch, SAMPLES_PER_BATCH samples longer than the batch
    before, for VISITS batches; channels are staggered so every batch carries reads of every age.  Reads come from a
    pool of synthetic raw reads (adapter + poly(A) plateau + RNA squiggle, synth.make_raw_read); a read's id is unique
    to its (channel, turn), and its raw_data is a memoryview of the pool (no copies, so PromethION-scale batches of
    18 000 channels stay cheap to script)."""
    rng = np.random.default_rng(7)
    pool_reads = pool_reads or min(2 * channels, 1024)
    pool = []
    for rid in range(pool_reads):
        n = int(rng.integers(min_len, max_len))
        sig = synth.make_raw_read(seed, rid, n, polya=(rid % 5 != 0))
        pool.append((memoryview(np.ascontiguousarray(sig, dtype=np.int16).tobytes()), n))
    batches = []
    for b in range(n_batches):
        reads = []
        for ch in range(channels):
            age = (b + ch) % VISITS                              # batches this read has been in the pore
            turn = (b + ch) // VISITS
            raw, n = pool[(turn * channels + ch) % pool_reads]
            seen = min(n, n - (VISITS - 1 - age) * SAMPLES_PER_BATCH)
            seen = max(seen, min(n, 2000))
            reads.append((ch + 1, FakeRead(f"read-{turn}-{ch}", raw[: 2 * seen])))
        batches.append(reads)
    return batches


def run_replay(models, processor, batches, mode: str = "enrich", threshold: float = 0.9, skip: int = 3,
               signal_cache: bool = True) -> dict:
    """Drive SequencerControl.target over `batches`; returns counts and per-batch latency percentiles
    (the first `skip` batches are warm-up: first launches, the signal store's first fill)."""
    from .control import SequencerControl
    client = FakeClient(batches)
    with tempfile.TemporaryDirectory() as d:
        ctl = SequencerControl(client, models, processor, logging.getLogger("riser_amd.replay"),
                               os.path.join(d, "out"), signal_cache=signal_cache)
        ctl.reserve(max(len(b) for b in batches))
        ctl.start()
        # the scripted batches are hundreds of thousands of long-lived Python objects that a live run never holds (its
        # reads are transient): keep the cyclic collector from walking them in the middle of a timed batch
        gc.collect()
        gc.freeze()
        try:
            t0 = time.perf_counter()
            ctl.target(mode, 1.0, threshold)
            torch.cuda.synchronize(processor.device)
            wall = time.perf_counter() - t0
        finally:
            gc.unfreeze()
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
            "max_ms": round(float(lat.max()), 3), "latency_samples": int(lat.size),
            "first_ms": [round(float(v) * 1e3, 2) for v in list(ctl.batch_latencies)[:8]],
            "signal_cache": bool(signal_cache),
            "pcie_samples_uploaded": int(ctl._store.samples_uploaded),
            "pcie_samples_full_reupload": int(ctl._store.samples_presented),
            "rejected": sum(len(r) for r in client.rejected), "finished": sum(len(r) for r in client.finished)}
