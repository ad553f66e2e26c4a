"""world_size-2 gloo test of the N > 1 path: read-id sharding, the scalar reductions bench.py
uses for timing, and result assembly.  No data-path collective exists to test."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from riser_amd import dist as rdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = rdist.init(backend="gloo")
    ids = [f"read-{i:04d}" for i in range(1000)] + list(range(50))
    mine = rdist.shard_indices(ids, r, w)
    # fake per-read result: deterministic function of the read index
    probs = np.stack([np.cos(mine.astype(np.float32)), np.sin(mine.astype(np.float32))], axis=1)
    rdist.barrier()
    t_max = rdist.reduce_scalar(1.0 + r, "max")
    n_sum = rdist.reduce_scalar(len(mine), "sum")
    full = rdist.gather_results(mine, probs, len(ids))
    q.put((r, mine.tolist(), t_max, n_sum, float(np.abs(full[:, 0] - np.cos(np.arange(len(ids), dtype=np.float32))).max())))
    rdist.finalize()


def test_two_rank_sharding_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_idx = sorted(res[0][1] + res[1][1])
    assert all_idx == list(range(1050)), "shards must partition the reads"
    assert 400 < len(res[0][1]) < 650, "crc32 sharding should be roughly balanced"
    for r, _, t_max, n_sum, err in res:
        assert t_max == 2.0 and n_sum == 1050 and err == 0.0


def test_shard_is_stable_and_single_rank_identity():
    assert rdist.shard_of("abc", 8) == rdist.shard_of("abc", 8)
    assert rdist.shard_of(17, 1) == 0
    assert rdist.shard_indices(["a", "b", "c"], 0, 1).tolist() == [0, 1, 2]
