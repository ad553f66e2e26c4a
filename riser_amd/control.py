"""SequencerControl: the ReadUntil control loop of riser/control.py:3-153, batched.

Same constructor, `start()`, `target(mode, duration_h, threshold, unblock_duration)`,
`finish()`, same CSV columns, same decision rule and the same client calls in the same
order as the reference.  The difference is the shape of the work: the reference walks the
(<= 512) reads of a ReadUntil batch one at a time - trim, normalise, one forward per model at
batch 1, a device sync per probability (riser/control.py:31-93,152) - whereas this loop

  1. keeps the raw int16 signal of every read in flight RESIDENT ON THE DEVICE, one row per channel: an
     AccumulatingCache client (riser/client.py:29-31) re-sends a read's whole signal with every batch, and only the
     samples that are new since the last batch cross PCIe (one compacted transfer + one scatter launch),
  2. finds the poly(A) ends of all un-cached reads in one kernel launch,
  3. applies the reference's length gating on the host as array arithmetic (a trim is an offset into the resident
     signal, a truncation is a length),
  4. normalises every assessable read once and runs one batched forward per model,
  5. takes the ensemble decision on the device and copies probabilities + decisions back in
     a single synchronisation.

Host work per batch is numpy over arrays of the batch's reads; the only per-read Python is what the client's own
interface forces (one `get_raw_signal` call and one `read.id` per read) and the text of the CSV rows.

The polyA cache only memoises a deterministic prefix property of a read, so its state
never changes results; it is cleared at batch granularity once it holds >= 1000 entries
(riser/control.py:96-97 does so per read).
"""
from __future__ import annotations

import time
from collections import deque

import numpy as np
import torch

from . import _native as nv
from .model import classify_raw_ensemble

_MODE = {"enrich": nv.RS_ENRICH, "deplete": nv.RS_DEPLETE}

# user-visible behaviour of the reference that a drop-in must keep (riser/control.py:100-103, 125-133, 145-148):
# the operator warnings, the log lines and the CSV schema
_WARN_START = ('The sequencing run is being controlled by RISER, reads that are '
               'not in the target class will be ejected from the pore.')
_WARN_STOP = 'RISER has stopped running.'
_CSV_COLUMNS = ("batch_start", "read_id", "channel", "sig_length", "models", "prob_targets", "threshold", "mode",
                "decision")
_CACHE_LIMIT = 1000                  # poly(A) cache entries before it is dropped (riser/control.py:96-97)


class _MinuteTally:
    """Counts of the current minute and the once-a-minute progress line (riser/control.py:116-123): the line is
    written by the first batch that STARTS more than 60 s after the previous report."""

    def __init__(self, logger, now):
        self.logger = logger
        self.due = now + 60
        self.assessed = self.accepted = self.rejected = 0

    def add(self, assessed, accepted, rejected):
        self.assessed += assessed
        self.accepted += accepted
        self.rejected += rejected

    def report_if_due(self, batch_start):
        if batch_start <= self.due:
            return
        self.logger.info(f"In the last minute {self.assessed} signals were assessed, {self.accepted} were "
                         f"accepted and {self.rejected} were rejected")
        self.assessed = self.accepted = self.rejected = 0
        self.due = batch_start + 60


class _Pinned:
    """Grow-only pinned host scratch, handed out in aligned pieces and reset once per batch: the small index arrays of a
    batch go to the device as asynchronous copies from here (a pageable source makes every copy a blocking one)."""

    def __init__(self, device):
        self.device = device
        self.buf = None
        self.used = 0

    def reset(self, need: int):
        if self.buf is None or self.buf.numel() < need:
            self.buf = torch.empty(max(need, 1 << 16), dtype=torch.uint8).pin_memory()
        self.used = 0

    def to_device(self, arr: np.ndarray) -> torch.Tensor:
        """arr (contiguous) -> device tensor of the same dtype, copied asynchronously on the current stream"""
        nb = arr.nbytes
        at = self.used
        self.used = (at + nb + 63) & ~63
        if self.used > self.buf.numel():                         # not reserved for: fall back to a blocking copy
            return torch.from_numpy(arr).to(self.device)
        host = self.buf[at: at + nb].view(torch.from_numpy(arr[:0]).dtype)
        host.numpy()[:] = arr
        return host.to(self.device, non_blocking=True)


class _SignalStore:
    """Raw int16 signals of a batch on the device -> per-read offsets into one device buffer.

    resident=True: one row of `pitch` samples per channel holds the read currently in that pore; a read that was
    already there (same id, not shorter) uploads only the samples behind what the row holds.  The client contract this
    relies on is AccumulatingCache's: the same read id always carries the same signal prefix (riser/client.py:29-31).
    Reads longer than a row are uploaded whole into a spill area behind the rows.
    resident=False: every batch uploads every read whole (the round-2 behaviour; kept for the A/B in bench.py)."""

    def __init__(self, device, resident: bool = True, pitch: int = 32768):
        self.device, self.resident, self.pitch = device, resident, int(pitch)
        self.rowmap = np.full(1024, -1, dtype=np.int64)          # channel -> row
        self.n_rows = 0
        self.row_hash = np.zeros(0, dtype=np.int64)              # hash(read id) of the read a row holds
        self.row_have = np.zeros(0, dtype=np.int64)              # samples of it on the device
        self.cap_rows = 0
        self.spill_cap = 0
        self.buf = None                                          # int16 [cap_rows * pitch + spill_cap]
        self.stage = None                                        # pinned int16 staging of one batch's new samples
        self.stage_dev = None
        self.samples_uploaded = 0                                # statistics: what crossed PCIe / what a full re-upload
        self.samples_presented = 0                               # of every batch would have carried

    def _ensure(self, rows: int, spill: int):
        if self.buf is not None and rows <= self.cap_rows and spill <= self.spill_cap:
            return
        new_rows = max(rows, self.cap_rows * 2 if rows > self.cap_rows else self.cap_rows, 512 if self.resident else 0)
        new_spill = max(spill, self.spill_cap * 2 if spill > self.spill_cap else self.spill_cap, 1 << 20)
        buf = torch.zeros(new_rows * self.pitch + new_spill, dtype=torch.int16, device=self.device)
        if self.buf is not None and self.cap_rows:
            buf[: self.cap_rows * self.pitch].copy_(self.buf[: self.cap_rows * self.pitch])
        self.buf, self.cap_rows, self.spill_cap = buf, new_rows, new_spill

    def _stage(self, n: int):
        if self.stage is None or self.stage.numel() < n:
            cap = max(2 * n, 1 << 22)          # generous: pinning memory is a slow system call, growth must be rare
            self.stage = torch.empty(cap, dtype=torch.int16).pin_memory()
            self.stage_dev = torch.empty(cap, dtype=torch.int16, device=self.device)
        return self.stage.numpy()

    def update(self, channels: np.ndarray, ids, raws, lens: np.ndarray, pinned: _Pinned) -> np.ndarray:
        """-> int64 [B]: offset of every read's first sample in self.buf"""
        B = len(raws)
        self.samples_presented += int(lens.sum())
        # the staging buffers below are re-used: whatever the previous batch still has in flight must have landed
        torch.cuda.current_stream(self.device).synchronize()
        if not self.resident:
            total = int(lens.sum())
            offs = np.zeros(B, dtype=np.int64)
            np.cumsum(lens[:-1], out=offs[1:])
            self._ensure(0, total)
            stage = self._stage(total)
            np.concatenate(raws, out=stage[:total])
            self.buf[:total].copy_(self.stage[:total], non_blocking=True)
            self.samples_uploaded += total
            return offs
        # ---- rows: one per channel, assigned at first sight ---------------------------------------------------------
        if int(channels.min()) < 0:
            raise ValueError("negative channel number")
        cmax = int(channels.max())
        if cmax >= self.rowmap.shape[0]:
            grown = np.full(max(cmax + 1, 2 * self.rowmap.shape[0]), -1, dtype=np.int64)
            grown[: self.rowmap.shape[0]] = self.rowmap
            self.rowmap = grown
        rows = self.rowmap[channels]
        fresh = np.flatnonzero(rows < 0)
        if fresh.size:
            uniq, first = np.unique(channels[fresh], return_index=True)     # a channel appears once per batch, but be safe
            self.rowmap[uniq] = self.n_rows + np.arange(uniq.size)
            self.n_rows += int(uniq.size)
            self.row_hash = np.concatenate([self.row_hash, np.zeros(uniq.size, dtype=np.int64)])
            self.row_have = np.concatenate([self.row_have, np.zeros(uniq.size, dtype=np.int64)])
            rows = self.rowmap[channels]
        hashes = np.fromiter(map(hash, ids), dtype=np.int64, count=B)
        fits = lens <= self.pitch
        if np.unique(rows).size != B:                               # two reads of one channel in one batch: nothing resident
            fits = np.zeros(B, dtype=bool)
        same = fits & (self.row_hash[rows] == hashes) & (self.row_have[rows] <= lens) & (self.row_have[rows] > 0)
        start = np.where(same, self.row_have[rows], 0)
        seg_len = lens - start
        # reads longer than a row: whole, into the spill area, nothing remembered
        spill_len = np.where(fits, 0, lens)
        spill_off = np.zeros(B, dtype=np.int64)
        np.cumsum(spill_len[:-1], out=spill_off[1:])
        self._ensure(self.n_rows, int(spill_len.sum()))
        spill_base = self.cap_rows * self.pitch
        dst = np.where(fits, rows * self.pitch + start, spill_base + spill_off)
        self.row_hash[rows[fits]] = hashes[fits]
        self.row_have[rows[fits]] = lens[fits]
        self.row_have[rows[~fits]] = 0
        # ---- the new samples, compacted: one transfer, one scatter ------------------------------------------------
        total = int(seg_len.sum())
        self.samples_uploaded += total
        if total:
            src = np.zeros(B, dtype=np.int64)
            np.cumsum(seg_len[:-1], out=src[1:])
            stage = self._stage(total)
            st = start.tolist()
            np.concatenate([r[s:] for r, s in zip(raws, st)], out=stage[:total])
            self.stage_dev[:total].copy_(self.stage[:total], non_blocking=True)
            live = np.flatnonzero(seg_len > 0)
            d_src = pinned.to_device(np.ascontiguousarray(src[live]))
            d_dst = pinned.to_device(np.ascontiguousarray(dst[live]))
            d_len = pinned.to_device(np.ascontiguousarray(seg_len[live].astype(np.int32)))
            nv.check(nv.lib().rs_copy_segments(self.stage_dev.data_ptr(), self.buf.data_ptr(), d_src.data_ptr(),
                                               d_dst.data_ptr(), d_len.data_ptr(), int(live.size),
                                               torch.cuda.current_stream(self.device).cuda_stream), "rs_copy_segments")
        return np.where(fits, rows * self.pitch, spill_base + spill_off)


class SequencerControl:
    def __init__(self, client, models, processor, logger, out_file, signal_cache: bool = True):
        """signal_cache=False re-uploads every read whole with every batch (no device-resident signals)."""
        self.client, self.models, self.proc, self.logger = client, models, processor, logger
        self.out_filename = out_file
        # host wall time of the most recent assessed batches (seconds): bounded, a run lasts tens of hours
        self.batch_latencies = deque(maxlen=4096)
        self._store = _SignalStore(processor.device, resident=signal_cache)
        self._pinned = _Pinned(processor.device)
        self._res_probs = self._res_dec = None
        self._channels_seen = 0
        self._reserved_for = 0

    # ------------------------------------------------------------------------------------
    def reserve(self, reads: int):
        """Allocate the models' workspaces for batches of up to `reads` assessable reads now, so that no batch of the run
        pays for a device allocation (a growing workspace is a multi-GB hipMalloc + hipFree inside a 1 s window)."""
        from .model import reserve_ensemble
        reserve_ensemble(self.models, int(reads), self.proc.get_max_length())
        self._reserved_for = max(self._reserved_for, int(reads))

    def assess_batch(self, entries, mode, threshold, polyA_cache):
        """entries: list of (channel, read).  Returns one record per ASSESSED read, in
        batch order: (channel, read, sig_length, [p_on per model], decision str)."""
        if not entries:
            return []
        proc = self.proc
        dev = proc.device
        B = len(entries)
        self._channels_seen = max(self._channels_seen, B)
        if B > self._reserved_for:                 # first batch (a flow cell's channel count), or a larger one than ever seen
            self.reserve(max(512, B))
        reads = [e[1] for e in entries]
        channels = np.fromiter((e[0] for e in entries), dtype=np.int64, count=B)
        ids = [r.id for r in reads]
        get = self.client.get_raw_signal
        raws = [get(r) for r in reads]
        if raws[0].dtype != np.int16:
            from .preprocess import _as_int16
            raws = [_as_int16(s) for s in raws]
        lens = np.fromiter((s.shape[0] for s in raws), dtype=np.int64, count=B)
        self._pinned.reset(64 * B + (1 << 12))
        offs = self._store.update(channels, ids, raws, lens, self._pinned)
        sig = self._store.buf

        # -- poly(A) end for reads not in the cache: one launch -------------------------------
        cget = polyA_cache.get
        end = np.fromiter((cget(i, 0) for i in ids), dtype=np.int64, count=B)
        need = np.flatnonzero(end == 0)
        if need.size:
            d_off = self._pinned.to_device(np.ascontiguousarray(offs[need]))
            d_len = self._pinned.to_device(lens[need].astype(np.int32))
            found = proc.polyA_end_device(sig, d_off, d_len, int(need.size)).cpu().numpy().astype(np.int64)
            hit = np.flatnonzero(found > 0)
            end[need[hit]] = found[hit]
            for j in hit.tolist():
                polyA_cache[ids[need[j]]] = int(found[j])

        # -- gating (riser/control.py:36-60) as offsets / lengths ------------------------------
        max_len, min_len = proc.get_max_length(), proc.get_min_length()
        fixed = proc.get_fixed_trim_length()
        has = end > 0
        start = np.where(has, end + 1, fixed)
        length = lens - start
        ok = np.where(has, length >= min_len, lens > fixed + max_len)       # :53-56 / should_trim_fixed_length :39-50
        sel = np.flatnonzero(ok)
        if sel.size == 0:
            return []
        lens_a = np.minimum(length[sel], max_len).astype(np.int32)

        # -- normalise once, one batched forward per model, decision on the device -------------
        n_sel, n_models = int(sel.size), len(self.models)
        off_d = self._pinned.to_device(np.ascontiguousarray(offs[sel] + start[sel]))
        len_d = self._pinned.to_device(lens_a)
        if self._res_probs is None or self._res_probs.dev.shape[1] < n_sel or self._res_probs.dev.shape[0] != n_models:
            cap = max(n_sel, 512)
            self._res_probs = _Pair(torch.empty((n_models, cap, 2), dtype=torch.float32, device=dev),
                                    torch.empty((n_models, cap, 2), dtype=torch.float32).pin_memory())
            self._res_dec = _Pair(torch.empty(cap, dtype=torch.uint8, device=dev),
                                  torch.empty(cap, dtype=torch.uint8).pin_memory())
        probs_d = self._res_probs.dev.view(-1)[: n_models * n_sel * 2].view(n_models, n_sel, 2)
        dec_d = self._res_dec.dev[:n_sel]
        classify_raw_ensemble(self.models, sig, off_d, len_d, lens_a, out=probs_d, decision=dec_d, max_len=max_len,
                              threshold=threshold, mode=_MODE[mode])
        probs_p = self._res_probs.host.view(-1)[: n_models * n_sel * 2].view(n_models, n_sel, 2)
        dec_p = self._res_dec.host[:n_sel]
        probs_p.copy_(probs_d, non_blocking=True)
        dec_p.copy_(dec_d, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        p_on = probs_p.numpy()[:, :, 1].T.astype(np.float64).tolist()          # [read][model]
        dec_h = dec_p.numpy().tolist()
        names = nv.DECISION_NAMES
        return [(int(channels[i]), reads[i], n, p, names[d])
                for i, n, p, d in zip(sel.tolist(), lens_a.tolist(), p_on, dec_h)]

    # ------------------------------------------------------------------------------------
    def target(self, mode, duration_h, threshold, unblock_duration=0.1):
        """The run loop (riser/control.py:99-133): one ReadUntil batch per iteration until the client stops or
        `duration_h` hours have passed; every assessed read becomes a CSV row, rejects are sent first, then every
        decided read (rejected, accepted, at maximum length) is reported as finished."""
        if mode not in _MODE:
            raise ValueError(f"mode must be 'enrich' or 'deplete', got {mode!r}")
        client = self.client
        client.send_warning(_WARN_START)
        with open(f"{self.out_filename}.csv", "a") as sink:
            sink.write(",".join(_CSV_COLUMNS) + "\n")
            t_begin = time.monotonic()
            t_stop = t_begin + duration_h * 3600
            tally = _MinuteTally(self.logger, t_begin)
            cache = {}
            while client.is_running() and time.monotonic() < t_stop:
                cache = self._run_batch(sink, mode, threshold, unblock_duration, cache, tally)
            client.send_warning(_WARN_STOP)
            if not client.is_running():
                self.logger.info("Client has stopped.")
            if time.monotonic() > t_stop:
                self.logger.info(f"RISER has timed out after {duration_h} hours as requested.")

    def _run_batch(self, sink, mode, threshold, unblock_duration, cache, tally):
        t0 = time.monotonic()
        records = self.assess_batch(list(self.client.get_read_batch()), mode, threshold, cache)
        decided = {"reject": [], "accept": [], "no_decision": []}          # "try_again" reads stay with the client
        if records:
            head = f"{t0:.0f},"
            tail = "," + ";".join(m.target for m in self.models) + ","
            tail2 = f",{threshold},{mode},"
            key = self._client_key
            lines = []
            for channel, read, n_samples, p_on, decision in records:
                if decision in decided:
                    decided[decision].append((channel, key(read)))
                lines.append(f"{head}{read.id},{channel},{n_samples}{tail}{';'.join(map(str, p_on))}{tail2}{decision}")
            sink.write("\n".join(lines) + "\n")
        # riser/control.py:96-97 drops the cache at 1000 entries, two flow cells' worth of reads at 512 channels; the same
        # proportion at any channel count (the cache never changes a result, only how often a read is re-scanned)
        if len(cache) >= max(_CACHE_LIMIT, 2 * self._channels_seen):
            cache = {}
        self.client.reject_reads(decided["reject"], unblock_duration)
        self.client.finish_processing_reads(decided["reject"] + decided["accept"] + decided["no_decision"])
        if records:
            self.batch_latencies.append(time.monotonic() - t0)
        tally.add(len(records), len(decided["accept"]), len(decided["reject"]))
        tally.report_if_due(t0)
        return cache

    def start(self):
        self.client.start_streaming_reads()
        self.logger.info("Live read stream started.")

    def finish(self):
        self.client.reset()
        self.logger.info("Client reset and live read stream ended.")

    @staticmethod
    def _client_key(read):
        """What the ReadUntil client identifies a read by: minknow-api <= v5 has `.number`, >= v6 only `.id`
        (riser/control.py:137-143)."""
        return read.number if hasattr(read, "number") else read.id


class _Pair:
    """a device tensor and its pinned host twin"""

    def __init__(self, dev, host):
        self.dev, self.host = dev, host
