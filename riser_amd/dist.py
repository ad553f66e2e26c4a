"""Multi-GPU plumbing: one process per GPU, reads sharded by read id, NO data-path collective.

Reads are independent (riser/control.py:31-93 keeps no cross-read state except a host-side
cache keyed by read id), so scaling out is pure partitioning: a read id hashes to one rank,
each rank classifies its shard on its own GPU with a replicated model, and only scalars
(elapsed time, counts) are ever reduced - through torch.distributed, whose "nccl" backend is
RCCL on ROCm; "gloo" serves the CPU tests.  xGMI carries nothing on this path.
"""
from __future__ import annotations

import datetime
import os
import zlib

import numpy as np
import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str | None = None, device: torch.device | None = None):
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # RS_DIST_BACKEND=gloo rehearses the multi-rank path with several ranks sharing one GPU
        backend = backend or os.environ.get("RS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        # bounded rendezvous: a sibling that died before it reached the store must not leave this rank waiting for the
        # library's default 10-30 minutes (the launcher's poll ends the run first; this is the second line of defence)
        # (180 s, not 60: with the nccl backend the same figure bounds every later collective, and RCCL's first
        # communicator set-up on an 8-GPU node can itself take tens of seconds; RS_DIST_TIMEOUT_S overrides)
        kw = {"timeout": datetime.timedelta(seconds=float(os.environ.get("RS_DIST_TIMEOUT_S", "180")))}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()


def shard_of(read_id, world_size: int) -> int:
    """Stable owner rank of a read id (str or int): crc32, not Python's salted hash()."""
    if isinstance(read_id, (int, np.integer)):
        key = int(read_id).to_bytes(8, "little", signed=True)
    else:
        key = str(read_id).encode("utf-8")
    return zlib.crc32(key) % world_size


def shard_indices(read_ids, rank: int, world_size: int) -> np.ndarray:
    """Indices of `read_ids` owned by `rank`."""
    if world_size == 1:
        return np.arange(len(read_ids), dtype=np.int64)
    return np.asarray([i for i, r in enumerate(read_ids) if shard_of(r, world_size) == rank], dtype=np.int64)


def barrier(device=None):
    if dist.is_initialized():
        if device is not None and device.type == "cuda" and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def reduce_scalar(value: float, op: str, device=None) -> float:
    """max / sum of a host scalar over ranks (the only collective on this path)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64,
                     device=device if (device is not None and dist.get_backend() == "nccl") else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())


def gather_results(local_idx: np.ndarray, local_probs: np.ndarray, total: int, device=None):
    """Assemble per-read results on every rank from the shards (host-side, small: 2 floats
    per read).  Used by the replay harness, not inside timed regions."""
    if not dist.is_initialized():
        out = np.zeros((total, 2), dtype=np.float32)
        out[local_idx] = local_probs
        return out
    objs = [None] * dist.get_world_size()
    dist.all_gather_object(objs, (np.asarray(local_idx), np.asarray(local_probs, dtype=np.float32)))
    out = np.zeros((total, 2), dtype=np.float32)
    for idx, pr in objs:
        out[idx] = pr
    return out


def classify_population(models, read_ids, load_signals, rank: int | None = None, world: int | None = None,
                        sub_batch: int = 1024, gather: bool = True):
    """BASELINE config 4 / 5 as one call per rank: the population `read_ids` is sharded by read id
    (`shard_indices`), each rank loads ONLY its own reads (`load_signals(indices) -> int16 [n, L]` host array,
    plus optional per-read lengths as a second return value), uploads them once, classifies them in sub-batches
    with its replicated models, and - if `gather` - assembles the [n_models, N, 2] probabilities of the whole
    population on every rank (host side, 8 bytes per read and model; never a device collective).
    Returns (my_indices, my_probs [n_models, n_mine, 2] numpy, full or None)."""
    from .stream import classify_resident
    if rank is None or world is None:
        rank, _, world = env_world()
    models = list(models)
    mine = shard_indices(read_ids, rank, world)
    loaded = load_signals(mine)
    sigs, lens = loaded if isinstance(loaded, tuple) else (loaded, None)
    sigs = np.ascontiguousarray(sigs, dtype=np.int16)
    n, L = sigs.shape
    dev = models[0].device
    probs = np.zeros((len(models), 0, 2), dtype=np.float32)
    if n:
        sig_dev = torch.from_numpy(sigs.reshape(-1)).to(dev)
        probs = classify_resident(models, sig_dev, n, L, lens, sub_batch).cpu().numpy()
    full = None
    if gather:
        full = np.stack([gather_results(mine, probs[m], len(read_ids), dev) for m in range(len(models))])
    return mine, probs, full
