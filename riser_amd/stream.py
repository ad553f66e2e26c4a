"""StreamClassifier: classify a large host-resident population of reads in sub-batches, with the
PCIe upload of sub-batch i+1 overlapped with the kernels of sub-batch i (two HIP streams, double
buffered device and pinned host staging).

This is the shape of BASELINE config 4 (a PromethION-scale 144 k concurrent chunks, 18 k per GPU):
reads are independent, so a GPU simply walks its shard; nothing is exchanged between GPUs
(riser_amd.dist shards by read id).  Per read the wire cost is 2 bytes/sample up and 8 bytes
down per model.
"""
from __future__ import annotations

import numpy as np
import torch


class StreamClassifier:
    def __init__(self, models, sub_batch: int = 1024, max_len: int = 16000):
        self.models = list(models)
        self.device = self.models[0].device
        self.sub_batch = int(sub_batch)
        self.max_len = int(max_len)
        # ONE pair of streams per device for every classifier of the process (round 6): HIP multiplexes user streams onto a
        # handful of hardware queues, and the SECOND classifier created in a process got a copy stream and a compute stream that
        # did not overlap (tools/host_fed_probe.py: 1.25 ms per 512 x 16000 bf16x3 batch against 0.95-1.0 for the first, the third
        # and the fourth, whatever their arithmetic); a model's per-stream workspace is then shared by them as well
        self.copy_stream, self.compute_stream = _pipeline_streams(self.device)
        n = self.sub_batch * self.max_len
        self._pinned = [torch.empty(n, dtype=torch.int16).pin_memory() for _ in range(2)]
        self._dev = [torch.empty(n, dtype=torch.int16, device=self.device) for _ in range(2)]
        self._probs = [torch.empty((len(self.models), self.sub_batch, 2), dtype=torch.float32, device=self.device)
                       for _ in range(2)]
        self._uploaded = [torch.cuda.Event() for _ in range(2)]
        self._consumed = [torch.cuda.Event() for _ in range(2)]

    def classify(self, signals: np.ndarray, lengths: np.ndarray | None = None) -> np.ndarray:
        """signals: int16 [N, L] (row i valid for lengths[i] samples, default L) in host memory
        (numpy, or a pinned torch tensor to skip the staging copy).  Returns float32
        [n_models, N, 2] = (p_off, p_on) on the host."""
        pinned_in = torch.is_tensor(signals) and signals.is_pinned()
        N, L = signals.shape
        if L > self.max_len:
            raise ValueError("rows longer than max_len")
        lengths = np.full(N, L, dtype=np.int32) if lengths is None else np.asarray(lengths, dtype=np.int32)
        out = torch.empty((len(self.models), N, 2), dtype=torch.float32).pin_memory()
        SB = self.sub_batch
        n_sub = (N + SB - 1) // SB
        # nothing inside the loop may block the host or touch pageable memory (a pageable copy waits for the stream it is ordered
        # on: the host then queues batch i's kernels only after batch i - 1's have finished, and the device idles for the
        # length of the host's launch sequence - measured 0.80 of the resident rate in fp32): every read's length and offset goes
        # up ONCE, from pinned memory, before the first sub-batch
        lens_pin = torch.from_numpy(np.ascontiguousarray(lengths)).pin_memory()
        with torch.cuda.stream(self.compute_stream):
            lens_dev = lens_pin.to(self.device, non_blocking=True)
            off_dev = torch.arange(SB, dtype=torch.int64, device=self.device) * L
        for i in range(n_sub):
            k = i & 1
            lo, hi = i * SB, min(N, (i + 1) * SB)
            nb = hi - lo
            # ---- upload on the copy stream (after the kernels that last read this buffer) -------
            with torch.cuda.stream(self.copy_stream):
                if i >= 2:
                    self.copy_stream.wait_event(self._consumed[k])
                if pinned_in:
                    src = signals[lo:hi].reshape(-1)
                else:
                    if i >= 2:
                        self._consumed[k].synchronize()             # the pinned staging buffer is host-written
                    src = self._pinned[k][: nb * L]
                    src.numpy()[:] = np.asarray(signals[lo:hi]).reshape(-1)
                self._dev[k][: nb * L].copy_(src, non_blocking=True)
                self._uploaded[k].record(self.copy_stream)
            # ---- kernels on the compute stream ---------------------------------------------------
            lens_h = lengths[lo:hi]
            with torch.cuda.stream(self.compute_stream):
                self.compute_stream.wait_event(self._uploaded[k])
                for m, model in enumerate(self.models):
                    model.classify_raw(self._dev[k], off_dev[:nb], lens_dev[lo:hi], lens_h, out=self._probs[k][m, :nb])
                out[:, lo:hi].copy_(self._probs[k][:, :nb], non_blocking=True)
                self._consumed[k].record(self.compute_stream)
        self.compute_stream.synchronize()
        return out.numpy()


_SIDE_STREAMS = {}
_PIPE_STREAMS = {}


def _pipeline_streams(dev):
    """(copy stream, compute stream) of StreamClassifier on `dev`, created once per process"""
    if dev.index not in _PIPE_STREAMS:
        _PIPE_STREAMS[dev.index] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
    return _PIPE_STREAMS[dev.index]



def _side_streams(dev, n: int):
    """n side streams per device, created once: every stream that classifies owns a workspace in each model
    (model.Workspace is keyed by stream), so fresh streams per call would pin a new multi-GB workspace per call."""
    key = (dev.index, int(n))
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _SIDE_STREAMS[key]


def classify_resident(models, sig_dev: torch.Tensor, n_reads: int, read_len: int, lengths: np.ndarray | None = None,
                      sub_batch: int = 1024, out: torch.Tensor | None = None, streams: int = 1) -> torch.Tensor:
    """Classify a population that is ALREADY resident in HBM: int16 [n_reads * read_len] (row i valid for
    lengths[i] samples, default read_len), walked in sub-batches of `sub_batch` reads - one fused normalise + forward
    call per model and sub-batch, no host round trip in between.  This is the per-GPU inner loop of BASELINE
    config 4 once a rank's shard has been uploaded (18 000 chunks x 16000 samples = 576 MB of a 288 GB HBM).
    Returns fp32 [n_models, n_reads, 2] on the device; results are bit-identical to direct calls on any
    sub-range (reads are independent and every kernel is batch-composition invariant).
    `streams` > 1 keeps that many sub-batches in flight on separate HIP streams (each with its own workspace): the
    idle CUs of one sub-batch's tile rounds, prologues and launch boundaries run the other's work (fp32: +6 % at two)."""
    models = list(models)
    dev = models[0].device
    lengths = (np.full(n_reads, read_len, dtype=np.int32) if lengths is None
               else np.ascontiguousarray(lengths, dtype=np.int32))
    if out is None:
        out = torch.empty((len(models), n_reads, 2), dtype=torch.float32, device=dev)
    off_all = torch.arange(n_reads, dtype=torch.int64, device=dev) * read_len
    len_all = torch.from_numpy(lengths).to(dev)
    if streams <= 1:
        for lo in range(0, n_reads, sub_batch):
            hi = min(n_reads, lo + sub_batch)
            for m, model in enumerate(models):
                model.classify_raw(sig_dev, off_all[lo:hi], len_all[lo:hi], lengths[lo:hi], out=out[m, lo:hi])
        return out
    caller = torch.cuda.current_stream(dev)
    side = _side_streams(dev, streams)
    for st in side:
        st.wait_stream(caller)                                   # inputs (and `out`) were produced on the caller's stream
    for k, lo in enumerate(range(0, n_reads, sub_batch)):
        hi = min(n_reads, lo + sub_batch)
        with torch.cuda.stream(side[k % streams]):
            for m, model in enumerate(models):
                model.classify_raw(sig_dev, off_all[lo:hi], len_all[lo:hi], lengths[lo:hi], out=out[m, lo:hi])
    for st in side:
        caller.wait_stream(st)
    return out
