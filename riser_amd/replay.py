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
                     max_len: int = 22000, pool_reads: int | None = None, first_channel: int = 1, first_batch: int = 0):
    """-> list of n_batches lists of (channel, FakeRead).

    The traffic of an AccumulatingCache client (riser/client.py:29-31, `get_read_chunks(last=True)`): every channel
    holds one read at a time and re-sends it WHOLE with every batch, SAMPLES_PER_BATCH samples longer than the batch
    before, for VISITS batches; channels are staggered so every batch carries reads of every age.  Reads come from a
    pool of synthetic raw reads (adapter + poly(A) plateau + RNA squiggle, synth.make_raw_read); a read's id is unique
    to its (channel, turn), and its raw_data is a memoryview of the pool (no copies, so PromethION-scale batches of
    18 000 channels stay cheap to script).  `first_channel`: the channel numbers run from there (one rank's range of a
    sharded flow cell, riser_amd/launch.py); `first_batch`: continue a script where an earlier call stopped."""
    rng = np.random.default_rng(7)
    pool_reads = pool_reads or min(2 * channels, 1024)
    pool = []
    for rid in range(pool_reads):
        n = int(rng.integers(min_len, max_len))
        sig = synth.make_raw_read(seed, rid, n, polya=(rid % 5 != 0))
        pool.append((memoryview(np.ascontiguousarray(sig, dtype=np.int16).tobytes()), n))
    batches = []
    for b in range(first_batch, first_batch + n_batches):
        reads = []
        for ch in range(channels):
            age = (b + ch) % VISITS                              # batches this read has been in the pore
            turn = (b + ch) // VISITS
            raw, n = pool[(turn * channels + ch) % pool_reads]
            seen = min(n, n - (VISITS - 1 - age) * SAMPLES_PER_BATCH)
            seen = max(seen, min(n, 2000))
            reads.append((first_channel + ch, FakeRead(f"read-{turn}-{first_channel + ch - 1}", raw[: 2 * seen])))
        batches.append(reads)
    return batches


def chunked_batches(n_batches: int, channels: int = 512, seed: int = 4242, chunk: int = 8000, min_len: int = 9000,
                    max_len: int = 30000, pool_reads: int | None = None):
    """-> list of n_batches lists of (channel, FakeRead): the traffic of a client that POPS its cache with every batch.

    `ReadUntilClient.get_read_chunks(last=True)` (riser/client.py:44) removes what it returns from the AccumulatingCache,
    which only concatenates the chunks that arrive between two pops: a read that stays in its pore comes back under the SAME
    id carrying only the samples since the last batch - here `chunk` samples (2 s at RNA004's 4 kHz: the reference's own
    loop takes seconds per batch, so seconds of signal accumulate between its pops).  Consecutive deliveries of a read are
    DISJOINT: nothing of an earlier delivery is re-sent, and the reference applies a cached poly(A) end of the first
    delivery as an offset into the later ones (riser/preprocess.py:93-100) - a drop-in does the same."""
    rng = np.random.default_rng(11)
    pool_reads = pool_reads or min(2 * channels, 1024)
    pool = []
    for rid in range(pool_reads):
        n = int(rng.integers(min_len, max_len))
        sig = synth.make_raw_read(seed, rid, n, polya=(rid % 5 != 0))
        pool.append((memoryview(np.ascontiguousarray(sig, dtype=np.int16).tobytes()), n))
    visits = -(-max_len // chunk)
    batches = []
    for b in range(n_batches):
        reads = []
        for ch in range(channels):
            age = (b + ch) % visits
            turn = (b + ch) // visits
            raw, n = pool[(turn * channels + ch) % pool_reads]
            lo, hi = age * chunk, min(n, (age + 1) * chunk)
            if hi - lo < 500:                                     # the pore is between reads
                continue
            reads.append((ch + 1, FakeRead(f"read-{turn}-{ch}", raw[2 * lo: 2 * hi])))
        batches.append(reads)
    return batches


def run_replay(models, processor, batches, mode: str = "enrich", threshold: float = 0.9, skip: int = 3,
               signal_cache: bool = True, client_cls=FakeClient) -> dict:
    """Drive SequencerControl.target over `batches`; returns counts and per-batch latency percentiles
    (the first `skip` batches are warm-up: first launches, the signal store's first fill)."""
    from .control import PHASES, SequencerControl
    client = client_cls(batches)
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
    loop = np.asarray(list(ctl.batch_loop_times)[skip:], dtype=np.float64) * 1e3
    if lat.size == 0:
        lat = loop = np.zeros(1)
    phases = np.asarray(list(ctl.batch_phases)[skip:], dtype=np.float64).reshape(-1, len(PHASES)) * 1e3
    worst = int(np.argmax(loop)) if phases.shape[0] == loop.shape[0] and loop.size else 0
    received = sum(len(b) for b in batches)
    store = ctl._store
    return {"batches": len(batches), "reads_received": received, "reads_assessed": rows,
            "assessed_per_batch": round(rows / max(len(batches), 1), 1),
            "wall_s": round(wall, 3), "assessed_per_s": round(rows / wall, 1),
            # decision latency: get_read_batch() -> reject / finish calls sent; loop: the whole iteration, CSV rows included
            "p50_ms": round(float(np.percentile(lat, 50)), 3), "p99_ms": round(float(np.percentile(lat, 99)), 3),
            "max_ms": round(float(lat.max()), 3), "latency_samples": int(lat.size),
            "loop_p50_ms": round(float(np.percentile(loop, 50)), 3), "loop_max_ms": round(float(loop.max()), 3),
            "first_ms": [round(float(v) * 1e3, 2) for v in list(ctl.batch_latencies)[:8]],
            "phase_ms_median": ({n: round(float(v), 3) for n, v in zip(PHASES, np.median(phases, axis=0))}
                                if phases.size else {}),
            "phase_ms_slowest_batch": ({n: round(float(v), 3) for n, v in zip(PHASES, phases[worst])} if phases.size else {}),
            "gc_frozen": True, "host_loops": "native" if getattr(client, "raw_data_dtype", None) is not None else "python",
            "signal_cache": bool(signal_cache), "signal_cache_auto_off": bool(store.auto_off),
            "delta_reads": int(store.delta_reads), "overlap_mismatches": int(store.mismatches),
            "pcie_samples_uploaded": int(store.samples_uploaded),
            "pcie_samples_full_reupload": int(store.samples_presented),
            "rejected": sum(len(r) for r in client.rejected), "finished": sum(len(r) for r in client.finished)}
