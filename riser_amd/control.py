"""SequencerControl: the ReadUntil control loop of riser/control.py:3-153, batched.

Same constructor, `start()`, `target(mode, duration_h, threshold, unblock_duration)`,
`finish()`, same CSV columns, same decision rule and the same client calls in the same
order as the reference.  The difference is the shape of the work: the reference walks the
(<= 512) reads of a ReadUntil batch one at a time - trim, normalise, one forward per model at
batch 1, a device sync per probability (riser/control.py:31-93,152) - whereas this loop

  1. stages the raw int16 signals of the batch in pinned host memory and uploads them in one transfer.  A client that
     re-sends a read's WHOLE signal with every batch (an accumulating client) gets a device-resident signal store: one
     row per channel, and only the samples behind what the row already holds cross PCIe - after the overlap with the
     held samples has been verified (see _SignalStore); a client that delivers disjoint chunks under one read id (what
     `get_read_chunks(last=True)` of riser/client.py:44 does between two pops) is detected and uploaded whole,
  2. finds the poly(A) ends of all un-cached reads in one kernel launch,
  3. applies the reference's length gating on the host as array arithmetic (a trim is an offset into the staged
     signal, a truncation is a length),
  4. normalises every assessable read once and runs one batched forward per model,
  5. takes the ensemble decision on the device and copies probabilities + decisions back in
     a single synchronisation,
  6. sends the reject / finish calls, THEN writes the CSV rows of the batch (the reference writes each row before the
     calls; the file's content is the same, the pore learns its decision earlier).

Host work per batch is numpy over arrays of the batch's reads.  What is left per read in Python is what the client's own
interface forces - one `get_raw_signal` call and one `read.id` per read; a client that declares `raw_data_dtype = np.int16`
(its get_raw_signal is np.frombuffer(read.raw_data, int16), as riser/client.py:46-47 is) has its reads walked by the C loops
of riser_amd/_hostpack instead.

The polyA cache only memoises a deterministic prefix property of a read (the scan of riser/preprocess.py:42-79 is causal:
an end found on a prefix is the end of every extension, tests/test_gpu_more.py), so its state never changes results; it
is dropped at batch granularity once it holds >= 1000 entries (riser/control.py:96-97 does so per read).
"""
from __future__ import annotations

import queue
import threading
import time
from collections import deque
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _native as nv
from .model import classify_raw_ensemble

try:                                     # optional C loops over the reads of a batch (riser_amd/csrc/hostpack.c)
    from . import _hostpack as _hp
except ImportError:                      # not built: the Python loops below do the same work
    _hp = None

_MODE = {"enrich": nv.RS_ENRICH, "deplete": nv.RS_DEPLETE}

# user-visible behaviour of the reference that a drop-in must keep (riser/control.py:100-103, 125-133, 145-148):
# the operator warnings, the log lines and the CSV schema
_WARN_START = ('The sequencing run is being controlled by RISER, reads that are '
               'not in the target class will be ejected from the pore.')
_WARN_STOP = 'RISER has stopped running.'
_CSV_COLUMNS = ("batch_start", "read_id", "channel", "sig_length", "models", "prob_targets", "threshold", "mode",
                "decision")
_CACHE_LIMIT = 1000                  # poly(A) cache entries before it is dropped (riser/control.py:96-97)
_MINION_CHANNELS = 512               # riser/client.py:11
PHASES = ("client_reads", "stage_upload", "polya_sync", "gate_launch", "device_wait", "client_calls", "csv")


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
        self._copied = {}                      # stream -> event recorded behind the last copy issued from the scratch there
        self._lock = threading.Lock()          # the staging thread of the slice pipeline takes pieces too

    def reset(self, need: int):
        """Start a new batch at offset 0 - once every copy of the previous batch has READ its source: a batch may return
        without a stream synchronise (no assessable read), and its index arrays may still be queued."""
        for ev in self._copied.values():
            ev.synchronize()
        if self.buf is None or self.buf.numel() < need:
            self.buf = torch.empty(max(need, 1 << 16), dtype=torch.uint8).pin_memory()
        self.used = 0

    def _mark(self):
        st = torch.cuda.current_stream(self.device)
        with self._lock:
            ev = self._copied.get(st.cuda_stream)
            if ev is None:
                ev = self._copied[st.cuda_stream] = torch.cuda.Event()
        ev.record(st)

    def _take(self, nbytes_list):
        """-> (start, [(offset, nbytes)], end) of a run of 64-byte-aligned pieces, or None when the scratch is too small"""
        with self._lock:
            at0 = at = self.used
            spans = []
            for nb in nbytes_list:
                spans.append((at, nb))
                at = (at + nb + 63) & ~63
            if at > self.buf.numel():
                return None
            self.used = at
        return at0, spans, at

    def to_device_many(self, arrs) -> list:
        """several small contiguous 1-D arrays -> device tensors of their dtypes, as ONE asynchronous copy (a copy per array
        costs ~10 us of launch time each, and a batch uploads seven of them)"""
        got = self._take([a.nbytes for a in arrs])
        if got is None:                                          # not reserved for: one by one (blocking fall-back inside)
            return [self.to_device(a) for a in arrs]
        at0, spans, at = got
        for a, (o, nb) in zip(arrs, spans):
            self.buf[o: o + nb].view(torch.from_numpy(a[:0]).dtype).numpy()[:] = a
        dev = self.buf[at0: at].to(self.device, non_blocking=True)
        self._mark()
        return [dev[o - at0: o - at0 + nb].view(torch.from_numpy(a[:0]).dtype) for a, (o, nb) in zip(arrs, spans)]

    def to_device(self, arr: np.ndarray) -> torch.Tensor:
        """arr (contiguous) -> device tensor of the same dtype, copied asynchronously on the current stream"""
        nb = arr.nbytes
        got = self._take([nb])
        if got is None:                                          # not reserved for: fall back to a blocking copy
            return torch.from_numpy(arr).to(self.device)
        at = got[0]
        host = self.buf[at: at + nb].view(torch.from_numpy(arr[:0]).dtype)
        host.numpy()[:] = arr
        dev = host.to(self.device, non_blocking=True)
        self._mark()
        return dev


def _object_array(items: list) -> np.ndarray:
    """list -> 1-D object array (np.array would make a 2-D array of a list of equal-length tuples)"""
    out = np.empty(len(items), dtype=object)
    if items and type(items[0]) is str:
        out[:] = items
    else:
        for i, v in enumerate(items):
            out[i] = v
    return out


class _Batch:
    """The reads of one ReadUntil batch, as the store sees them: ids, lengths and a way to stage `raw[start:]` of every
    read back to back.  Two implementations of the per-read walk: the client's own get_raw_signal (the eight-method duck
    type), or the C loops of _hostpack for clients whose raw_data IS the int16 signal."""

    def __init__(self, client, reads):
        self.reads = reads
        B = len(reads)
        self.ids = _object_array(_hp.attrs(reads, "id")) if _hp is not None and type(reads) is list else \
            np.fromiter((r.id for r in reads), dtype=object, count=B)
        declared = getattr(client, "raw_data_dtype", None)
        self.native = _hp is not None and declared is not None and np.dtype(declared) == np.int16
        if self.native:
            self.raws = None
            self.lens = np.empty(B, dtype=np.int64)
            _hp.lengths(reads, self.lens)
        else:
            get = client.get_raw_signal
            raws = [get(r) for r in reads]
            if B and raws[0].dtype != np.int16:
                from .preprocess import _as_int16
                raws = [_as_int16(s) for s in raws]
            self.raws = raws
            self.lens = np.fromiter((s.shape[0] for s in raws), dtype=np.int64, count=B)

    def view(self, lo: int, hi: int) -> "_Batch":
        """reads lo .. hi - 1 of the batch as a batch of their own (a slice of a PromethION-scale batch)"""
        if lo == 0 and hi == len(self.reads):
            return self
        v = object.__new__(_Batch)
        v.reads, v.ids, v.lens, v.native = self.reads[lo:hi], self.ids[lo:hi], self.lens[lo:hi], self.native
        v.raws = None if self.raws is None else self.raws[lo:hi]
        return v

    def stage(self, start: np.ndarray, out: np.ndarray) -> int:
        """out[: total] = concat(raw_i[start_i:]) -> total"""
        if self.native:
            return int(_hp.gather(self.reads, np.ascontiguousarray(start, dtype=np.int64), out))
        total = int((self.lens - start).sum())
        if start.any():
            np.concatenate([r[s:] for r, s in zip(self.raws, start.tolist())], out=out[:total])
        else:
            np.concatenate(self.raws, out=out[:total])
        return total


class _RowWriter:
    """The CSV rows of batch k are formatted and written by this thread while the loop assesses batch k + 1
    (riser/control.py:91-93,145-153 writes them inside the loop: same rows, same order, off the critical path).  The text
    is built by `_hostpack.format_rows` without the interpreter lock; rows reach the file in batch order (one thread, one
    queue) and `close()` - called before `target()` returns - waits for the last of them."""

    MAX_PENDING = 16                      # batches: a writer that cannot keep up slows the loop down instead of growing

    def __init__(self, sink, make_rows):
        self._sink, self._make_rows = sink, make_rows
        self._q = queue.Queue(maxsize=self.MAX_PENDING)
        self._err = None
        self._thread = threading.Thread(target=self._run, name="riser_amd-csv", daemon=True)
        self._thread.start()

    def put(self, *job):
        if self._err is not None:
            raise self._err
        self._q.put(job)

    def write_now(self, *job):
        """the same rows on the caller's thread, behind everything queued before them (row order is batch order)"""
        if self._err is not None:
            raise self._err
        self._q.join()
        self._sink.write(self._make_rows(*job))

    def _run(self):
        while True:
            job = self._q.get()
            if job is None:
                self._q.task_done()
                return
            if self._err is None:
                try:
                    self._sink.write(self._make_rows(*job))
                except BaseException as e:                      # noqa: BLE001 - re-raised on the loop's thread
                    self._err = e
            self._q.task_done()

    def close(self):
        self._q.put(None)
        self._thread.join()
        if self._err is not None:
            raise self._err


class _SignalStore:
    """Raw int16 signals of a batch on the device -> per-read offsets into one device buffer.

    resident=True: one row of `pitch` samples per channel holds the read currently in that pore.  A read that arrives
    again under the same id, not shorter than what the row holds, is a CANDIDATE for the delta path: its samples from
    TAIL before the held length on are staged and uploaded, and the first TAIL of them must equal the last TAIL samples
    the row holds (kept on the host) before the upload is scattered behind the held prefix.  A candidate that fails the
    comparison - a client that delivers disjoint chunks under one id, as ReadUntilClient.get_read_chunks does between two
    pops of its cache (riser/client.py:44) - is uploaded whole, exactly as with resident=False; when most re-seen reads
    of three consecutive batches fail, the store stops trying for the rest of the run (`auto_off`).  What the delta path
    therefore rests on: the client re-sends the same samples for [have - TAIL, have) - for a re-sent prefix always true,
    for a disjoint chunk of a noisy ADC signal never - not on a hash and not on trust in the client's contract.
    Reads longer than a row are uploaded whole into a spill area behind the rows.
    resident=False: every batch uploads every read whole."""

    TAIL = 32

    def __init__(self, device, resident: bool = True, pitch: int = 32768, logger=None):
        self.device, self.resident, self.pitch = device, resident, int(pitch)
        self.logger = logger
        self.rowmap = np.full(1024, -1, dtype=np.int64)          # channel -> row
        self.n_rows = 0
        self.row_id = np.empty(0, dtype=object)                  # id of the read a row holds
        self.row_have = np.zeros(0, dtype=np.int64)              # samples of it on the device
        self.row_tail = np.zeros((0, self.TAIL), dtype=np.int16)  # the last TAIL of them
        self.row_pa = np.zeros((0, 4), dtype=np.int32)           # poly(A) scan state of the read (rs_polya_end_resume)
        self.last_rows = self.last_delta = None                  # of the slice update() saw last: rows, delta-path mask
        self.cap_rows = 0
        self.spill_cap = 0
        self.buf = None                                          # int16 [cap_rows * pitch + spill_cap]
        self.stage = None                                        # pinned int16 staging of one batch's new samples
        self.stage_dev = None
        self.samples_uploaded = 0                                # statistics: what crossed PCIe / what a full re-upload
        self.samples_presented = 0                               # of every batch would have carried
        self.delta_reads = 0                                     # reads that took the delta path
        self.mismatches = 0                                      # candidates whose overlap did not match (uploaded whole)
        self.auto_off = False
        self._bad_streak = 0
        self._uploaded = None                                    # event: the last slice's staging has been read by its copy
        self._stats3 = np.zeros(3, dtype=np.int64)               # store_slice's counters (re-seen, delta, mismatches)

    def reserve(self, rows: int, samples: int):
        """device rows / pinned staging for batches of up to `rows` reads carrying up to `samples` new samples"""
        self._ensure(rows if self.resident else 0, 1 << 20)
        self._stage(samples, exact=True)
        if self.resident and rows > self.rowmap.shape[0]:
            grown = np.full(rows + 1, -1, dtype=np.int64)
            grown[: self.rowmap.shape[0]] = self.rowmap
            self.rowmap = grown

    def _ensure(self, rows: int, spill: int):
        if self.buf is not None and rows <= self.cap_rows and spill <= self.spill_cap:
            return
        new_rows = max(rows, self.cap_rows * 2 if rows > self.cap_rows else self.cap_rows, 512 if self.resident else 0)
        new_spill = max(spill, self.spill_cap * 2 if spill > self.spill_cap else self.spill_cap, 1 << 20)
        buf = torch.zeros(new_rows * self.pitch + new_spill, dtype=torch.int16, device=self.device)
        if self.buf is not None and self.cap_rows:
            buf[: self.cap_rows * self.pitch].copy_(self.buf[: self.cap_rows * self.pitch])
        self.buf, self.cap_rows, self.spill_cap = buf, new_rows, new_spill

    def _stage(self, n: int, exact: bool = False):
        if self.stage is None or self.stage.numel() < n:
            cap = max(n if exact else 2 * n, 1 << 22)   # generous: pinning memory is a slow system call, growth must be rare
            if self.stage_dev is not None:
                # the previous slice's transfer and scatter may still read the buffers being replaced, on either stream of
                # the slice pipeline: growth is rare (reserve() sizes them for the run), so it may wait for the device
                torch.cuda.synchronize(self.device)
            self.stage = torch.empty(cap, dtype=torch.int16).pin_memory()
            self.stage_dev = torch.empty(cap, dtype=torch.int16, device=self.device)
        return self.stage.numpy()

    def _rows_of(self, channels: np.ndarray) -> np.ndarray:
        """one row per channel, assigned at first sight"""
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
            uniq = np.unique(channels[fresh])                           # a channel appears once per batch, but be safe
            self.rowmap[uniq] = self.n_rows + np.arange(uniq.size)
            self.n_rows += int(uniq.size)
            self.row_id = np.concatenate([self.row_id, np.full(uniq.size, None, dtype=object)])
            self.row_have = np.concatenate([self.row_have, np.zeros(uniq.size, dtype=np.int64)])
            self.row_tail = np.concatenate([self.row_tail, np.zeros((uniq.size, self.TAIL), dtype=np.int16)])
            self.row_pa = np.concatenate([self.row_pa, np.zeros((uniq.size, 4), dtype=np.int32)])
            rows = self.rowmap[channels]
        return rows

    def begin_batch(self, channels: np.ndarray, lens: np.ndarray):
        """Once per ReadUntil batch, before its slices are updated: every device allocation the batch can need (a buffer
        that grew between two slices would move under the kernels of the earlier one) and the cursors of the areas the
        slices fill one behind the other."""
        self._dup_channels = False
        if self.resident:
            rows = self._rows_of(channels)
            # two reads of one channel in one batch (no real client does that: its cache is keyed by channel): nothing of
            # this batch goes to the rows - a later slice would overwrite a row an earlier slice's kernels still read
            self._dup_channels = bool(np.bincount(rows, minlength=self.n_rows).max(initial=0) > 1)
            self._ensure(self.n_rows, int(lens.sum()) if self._dup_channels else int(lens[lens > self.pitch].sum()))
        elif self.buf is None or self.buf.numel() < int(lens.sum()):
            self._ensure(0, int(lens.sum()))
        self._spill_at = 0           # samples of the spill area (resident) / of the whole buffer (not resident) in use
        self._slice_bad = self._slice_reseen = 0

    def end_batch(self):
        """a client whose re-seen reads are mostly NOT extensions of what the rows hold gains nothing from the rows"""
        if not self.resident:
            return
        if self._slice_reseen >= 16 and (self._slice_reseen - self._slice_bad) * 2 < self._slice_reseen:
            self._bad_streak += 1
            if self._bad_streak >= 3:
                self.resident, self.auto_off = False, True
                if self.logger is not None:
                    self.logger.info("Signal store: re-seen reads do not extend the signal already on the device (this client "
                                     "delivers chunks, not whole reads): every read is uploaded whole from now on.")
        elif self._slice_reseen:
            self._bad_streak = 0

    def update(self, channels: np.ndarray, batch: _Batch, pinned: _Pinned) -> np.ndarray:
        """One slice of a batch (the whole batch, for all but PromethION-scale ones) -> int64 [B]: offset of every read's
        first sample in self.buf.  Copies and launches go to the CURRENT stream.  The pinned staging is re-used from slice to
        slice: the host waits for the previous slice's transfer (an event, not the stream - the stream may hold that slice's
        poly(A) scan, which the caller wants to overlap with this staging)."""
        lens = batch.lens
        B = lens.shape[0]
        presented = int(lens.sum())
        self.samples_presented += presented
        if self._uploaded is None:
            self._uploaded = torch.cuda.Event()
        self._uploaded.synchronize()
        self.last_rows = self.last_delta = None
        if not self.resident:
            offs = np.zeros(B, dtype=np.int64)
            np.cumsum(lens[:-1], out=offs[1:])
            offs += self._spill_at
            total = batch.stage(np.zeros(B, dtype=np.int64), self._stage(presented))
            self.buf[self._spill_at: self._spill_at + total].copy_(self.stage[:total], non_blocking=True)
            self._uploaded.record(torch.cuda.current_stream(self.device))
            self._spill_at += total
            self.samples_uploaded += total
            return offs
        T = self.TAIL
        rows = self._rows_of(channels)
        fits = lens <= self.pitch
        if self._dup_channels:
            fits = np.zeros(B, dtype=bool)
        stage = self._stage(presented)
        if batch.native and _hp is not None and hasattr(_hp, "store_slice") and self.row_id.flags.c_contiguous:
            # the whole slice as ONE C call (csrc/hostpack.c:store_slice): ids compared as objects, the overlap of every delta
            # candidate checked in the read's own buffer BEFORE anything is staged, raw[start:] of every read copied by the
            # copy-thread pool, the rows' lengths / tails / ids updated - the interpreter lock is held for the id work only
            start = np.empty(B, dtype=np.int64)
            cand_u8 = np.empty(B, dtype=np.uint8)
            stats = self._stats3
            total = int(_hp.store_slice(batch.reads, batch.ids, self.row_id, np.ascontiguousarray(rows),
                                        np.ascontiguousarray(lens, dtype=np.int64), np.ascontiguousarray(fits, dtype=np.uint8),
                                        self.row_have, self.row_tail, T, stage, start, cand_u8, stats))
            cand = cand_u8.view(bool)
            seg_len = lens - start
            src = np.zeros(B, dtype=np.int64)
            np.cumsum(seg_len[:-1], out=src[1:])
            n_reseen, n_delta = int(stats[0]), int(stats[1])
            self.mismatches += int(stats[2])
            rows_updated = True
        else:
            rows_updated = False
            have = self.row_have[rows]
            reseen = fits & (self.row_id[rows] == batch.ids) & (have > 0)
            cand = reseen & (have <= lens) & (have >= T)
            windows = np.lib.stride_tricks.sliding_window_view(stage, T)      # windows[i] = stage[i: i + T], no copy
            while True:
                start = np.where(cand, have - T, 0)
                seg_len = lens - start
                src = np.zeros(B, dtype=np.int64)
                np.cumsum(seg_len[:-1], out=src[1:])
                total = batch.stage(start, stage)
                ci = np.flatnonzero(cand)
                if ci.size == 0:
                    break
                ok = (windows[src[ci]] == self.row_tail[rows[ci]]).all(axis=1)
                if ok.all():
                    break
                cand[ci[~ok]] = False                                   # not the prefix the row holds: whole, and staged again
                self.mismatches += int((~ok).sum())
            n_reseen, n_delta = int(reseen.sum()), int(cand.sum())
        self.delta_reads += n_delta
        if not self._dup_channels:                               # (two reads of one channel share a row: no per-row state then)
            self.last_rows, self.last_delta = rows, cand         # a delta read's prefix is what the row held: scans may resume
        # reads longer than a row: whole, into the spill area, nothing remembered
        spill_len = np.where(fits, 0, lens)
        spill_off = np.zeros(B, dtype=np.int64)
        np.cumsum(spill_len[:-1], out=spill_off[1:])
        spill_off += self._spill_at
        self._spill_at += int(spill_len.sum())
        spill_base = self.cap_rows * self.pitch
        dst = np.where(fits, rows * self.pitch + start, spill_base + spill_off)
        if not rows_updated:
            fi = np.flatnonzero(fits)
            self.row_id[rows[fi]] = batch.ids[fi]
            self.row_have[rows[fi]] = lens[fi]
            self.row_have[rows[~fits]] = 0
            keep = fi[lens[fi] >= T]                                # the new tails: the last T staged samples of each read
            if keep.size:
                self.row_tail[rows[keep]] = windows[src[keep] + seg_len[keep] - T]
        # ---- the new samples, compacted: one transfer, one scatter ------------------------------------------------
        self.samples_uploaded += total
        if total:
            self.stage_dev[:total].copy_(self.stage[:total], non_blocking=True)
            self._uploaded.record(torch.cuda.current_stream(self.device))
            live = np.flatnonzero(seg_len > 0)
            d_src, d_dst, d_len = pinned.to_device_many([np.ascontiguousarray(src[live]), np.ascontiguousarray(dst[live]),
                                                         np.ascontiguousarray(seg_len[live].astype(np.int32))])
            nv.check(nv.lib().rs_copy_segments(self.stage_dev.data_ptr(), self.buf.data_ptr(), d_src.data_ptr(),
                                               d_dst.data_ptr(), d_len.data_ptr(), int(live.size),
                                               torch.cuda.current_stream(self.device).cuda_stream), "rs_copy_segments")
        self._slice_reseen += n_reseen
        self._slice_bad += n_reseen - n_delta
        return np.where(fits, rows * self.pitch, spill_base + spill_off)


class _Assessed:
    """What one batch's assessment produced, as arrays over the ASSESSED reads in batch order."""
    __slots__ = ("reads", "sel", "channels", "n_samples", "p_on", "decision")

    def __init__(self, reads, sel, channels, n_samples, p_on, decision):
        self.reads, self.sel, self.channels, self.n_samples, self.p_on, self.decision = (reads, sel, channels, n_samples,
                                                                                         p_on, decision)

    def __len__(self):
        return int(self.sel.shape[0])

    def records(self):
        """one tuple per assessed read: (channel, read, sig_length, [p_on per model], decision str)"""
        names = nv.DECISION_NAMES
        return [(c, self.reads[i], n, p, names[d]) for i, c, n, p, d in
                zip(self.sel.tolist(), self.channels.tolist(), self.n_samples.tolist(), self.p_on.tolist(),
                    self.decision.tolist())]


class SequencerControl:
    def __init__(self, client, models, processor, logger, out_file, signal_cache: bool = True):
        """signal_cache=False uploads every read whole with every batch (no device-resident signals)."""
        self.client, self.models, self.proc, self.logger = client, models, processor, logger
        self.saturated_batches = 0          # batches in which a half-precision model overflowed (warned about, each)
        self.out_filename = out_file
        # host wall time of the most recent assessed batches (seconds): bounded, a run lasts tens of hours.
        # batch_latencies: get_read_batch() -> reject / finish calls sent (what the pore waits for);
        # batch_loop_times: the whole iteration, CSV rows included; batch_phases: seconds per PHASES entry
        self.batch_latencies = deque(maxlen=4096)
        self.batch_loop_times = deque(maxlen=4096)
        self.batch_phases = deque(maxlen=4096)
        self._store = _SignalStore(processor.device, resident=signal_cache, logger=logger)
        self._pinned = _Pinned(processor.device)
        self._res_probs = self._res_dec = self._polya_host = self._polya_state_host = None
        self._pa_events = []                      # one per slice: the poly(A) scan's result is in pinned memory
        self._side, self._events = None, []       # PromethION-scale batches: the upload / poly(A) stream of the slice pipeline
        self._stage_pool = None                   # ... and the host thread that stages the next slice
        self._channels_seen = 0
        self._reserved_for = 0
        self._ph = np.zeros(len(PHASES))

    # ------------------------------------------------------------------------------------
    def reserve(self, reads: int):
        """Allocate everything a batch of up to `reads` reads needs now - the models' workspaces, the signal store's
        device rows, the pinned staging buffers and the result buffers - so that no batch of the run pays for a device
        allocation or a page-locking system call (a growing workspace is a multi-GB hipMalloc + hipFree inside a 1 s
        window)."""
        from .model import reserve_ensemble
        reads = int(reads)
        # the largest call of a run is one slice (assess_batch), not the flow cell's channel count
        reserve_ensemble(self.models, min(reads, self.SLICE_READS * 3 // 2 + 1), self.proc.get_max_length())
        # a first batch carries every read whole: typically <= 6 s of signal per pore
        self._store.reserve(reads, min(reads * 24576, 1 << 29))
        self._pinned.reset(96 * reads + (1 << 12))
        self._result_buffers(reads)
        self._reserved_for = max(self._reserved_for, reads)

    def _result_buffers(self, n: int):
        n_models = len(self.models)
        if self._res_probs is None or self._res_probs.dev.shape[1] < n or self._res_probs.dev.shape[0] != n_models:
            dev = self.proc.device
            cap = max(n, 512)
            self._res_probs = _Pair(torch.empty((n_models, cap, 2), dtype=torch.float32, device=dev),
                                    torch.empty((n_models, cap, 2), dtype=torch.float32).pin_memory())
            self._res_dec = _Pair(torch.empty(cap, dtype=torch.uint8, device=dev),
                                  torch.empty(cap, dtype=torch.uint8).pin_memory())
            self._polya_host = torch.empty(cap, dtype=torch.int32).pin_memory()
            self._polya_state_host = torch.empty((cap, 4), dtype=torch.int32).pin_memory()

    SLICE_READS = 4096          # a batch of more than 1.5 x this many reads is assessed in slices of about this size
    FIRST_SLICE_HALF = True     # the first slice of a sliced batch is half a slice (the device idles until it is staged)
    # batches with at least this many assessed reads hand their CSV rows to the writer thread, smaller ones write them inline:
    # at 18 000 channels the thread takes 4.4 ms out of the loop; at 512 channels (tools/csv_thread_ab.py, alternating in one
    # process) it trades 0.09 ms of loop time for 0.07 ms of decision latency (the writer shares the interpreter lock with the
    # next batch's staging) with the same p99 / max - so a MinION-sized batch keeps its rows inline
    CSV_THREAD_MIN_READS = 2048
    # slices k >= 1 staged on a second host thread instead of on the loop's thread one slice ahead.  Built and measured in
    # round 5 (tools/control_ab.py, variants interleaved in one process, 18 000 channels): 22.4 against 22.5 ms in bf16x3, 33.0
    # against 34.1 ms in fp32 - inside the run-to-run spread.  What the thread can overlap is the memcpy (it releases the
    # interpreter lock); the array arithmetic around it does not, and the loop's own phases grow by what the thread takes
    # (gate_launch 1.2 -> 2.4 ms, client_calls 1.4 -> 3.2 ms).  Off; kept for the day the staging is one C call.
    STAGE_THREAD = False

    def assess_batch(self, entries, mode, threshold, polyA_cache):
        """entries: list of (channel, read).  -> _Assessed (the assessed reads in batch order) or None.

        A PromethION-scale batch (thousands of reads) is cut into SLICES that are pipelined over two HIP streams: while the
        kernels of slice k classify on the caller's stream, the host stages slice k + 1 and a side stream uploads it,
        scatters it and scans it for poly(A) ends - the host's share of the batch and the device's overlap instead of adding
        up.  A MinION-sized batch is one slice on the caller's stream."""
        if not entries:
            return None
        ph = self._ph
        t = time.perf_counter()
        proc = self.proc
        dev = proc.device
        B = len(entries)
        self._channels_seen = max(self._channels_seen, B)
        if B > self._reserved_for:                 # first batch (a flow cell's channel count), or a larger one than ever seen
            self.reserve(max(512, B))
        if _hp is not None and type(entries) is list:
            channels = np.empty(B, dtype=np.int64)
            reads = _hp.unpack(entries, channels)
        else:
            reads = [e[1] for e in entries]
            channels = np.fromiter((e[0] for e in entries), dtype=np.int64, count=B)
        batch = _Batch(self.client, reads)
        t, ph[0] = self._tick(t, 0)
        self._pinned.reset(96 * B + (1 << 12))
        store = self._store
        store.begin_batch(channels, batch.lens)
        n_slices = 1 if B <= self.SLICE_READS * 3 // 2 else -(-B // self.SLICE_READS)
        # the first slice is half a slice: nothing runs on the device until its samples are staged and uploaded
        if n_slices > 1 and self.FIRST_SLICE_HALF:
            bounds = [0] + [B * (2 * k + 1) // (2 * n_slices - 1) for k in range(n_slices - 1)] + [B]
        else:
            bounds = [B * k // n_slices for k in range(n_slices + 1)]
        caller = torch.cuda.current_stream(dev)
        if n_slices > 1:
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            while len(self._events) < n_slices:
                self._events.append(torch.cuda.Event())
            side = self._side
            side.wait_stream(caller)
        else:
            side = caller
        max_len, min_len = proc.get_max_length(), proc.get_min_length()
        fixed = proc.get_fixed_trim_length()
        n_models = len(self.models)
        self._result_buffers(B)
        flat_p, flat_d = self._res_probs.dev.view(-1), self._res_dec.dev
        parts, n_total = [], 0
        pa_host = self._polya_host
        while len(self._pa_events) < n_slices:
            self._pa_events.append(torch.cuda.Event())

        def upload_and_scan(k):
            """slice k: its new samples to the device rows, the poly(A) scan of its reads not in the cache launched, the
            scan's result on its way to pinned memory - nothing here waits for the device.  Runs on the loop's thread for
            the first slice and on the staging thread for the others (the current device and stream are per thread)."""
            lo, hi = bounds[k], bounds[k + 1]
            part = batch.view(lo, hi)
            with torch.cuda.device(dev), torch.cuda.stream(side):
                offs = store.update(channels[lo:hi], part, self._pinned)
                if side is not caller:
                    self._events[k].record(side)            # the slice's samples are in the rows behind this point
                end = np.zeros(hi - lo, dtype=np.int64)
                if polyA_cache:
                    if _hp is not None and type(polyA_cache) is dict:
                        _hp.lookup(polyA_cache, part.ids, end)
                    else:
                        cget = polyA_cache.get
                        end = np.fromiter((cget(i, 0) for i in part.ids), dtype=np.int64, count=hi - lo)
                need = np.flatnonzero(end == 0)
                found_h = found_d = state_h = rows_need = None
                if need.size:                                   # one launch for every read without a known end
                    ups = [np.ascontiguousarray(offs[need]), part.lens[need].astype(np.int32)]
                    if store.last_rows is not None:
                        # a read that took the delta path is the read the row held, longer: its scan resumes behind the
                        # windows the previous batches scanned (the state lives with the row); everything else starts over
                        rows_need = store.last_rows[need]
                        st_in = np.where(store.last_delta[need][:, None], store.row_pa[rows_need], 0).astype(np.int32)
                        d_off, d_len, d_st = self._pinned.to_device_many(ups + [np.ascontiguousarray(st_in).reshape(-1)])
                        found_d, state_d = proc.polyA_end_device(store.buf, d_off, d_len, int(need.size), state_in=d_st.view(-1, 4))
                        state_h = self._polya_state_host[lo: lo + need.size]
                        state_h.copy_(state_d, non_blocking=True)
                    else:
                        d_off, d_len = self._pinned.to_device_many(ups)
                        found_d = proc.polyA_end_device(store.buf, d_off, d_len, int(need.size))
                    found_h = pa_host[lo: lo + need.size]
                    found_h.copy_(found_d, non_blocking=True)
                    self._pa_events[k].record(side)
            return part, offs, end, need, found_h, (found_d, state_h, rows_need)

        def gate_and_classify(k, staged):
            nonlocal t, n_total
            lo = bounds[k]
            part, offs, end, need, found_h, (_found_d, state_h, rows_need) = staged
            ids, lens = part.ids, part.lens
            if need.size:
                self._pa_events[k].synchronize()
                if state_h is not None:
                    store.row_pa[rows_need] = state_h.numpy()
                found = found_h.numpy().astype(np.int64)
                hit = np.flatnonzero(found > 0)
                end[need[hit]] = found[hit]
                polyA_cache.update(zip(ids[need[hit]].tolist(), found[hit].tolist()))
            t, dt = self._tick(t, 2)
            ph[2] += dt
            # -- gating (riser/control.py:36-60) as offsets / lengths ------------------------------
            has = end > 0
            start = np.where(has, end + 1, fixed)
            length = lens - start
            ok = np.where(has, length >= min_len, lens > fixed + max_len)   # :53-56 / should_trim_fixed_length :39-50
            sel = np.flatnonzero(ok)
            if sel.size == 0:
                return
            lens_a = np.minimum(length[sel], max_len).astype(np.int32)
            off_d, len_d = self._pinned.to_device_many([np.ascontiguousarray(offs[sel] + start[sel]), lens_a])
            if side is not caller:
                caller.wait_event(self._events[k])
            # -- normalise once, one batched forward per model, decision on the device -------------
            n_sel = int(sel.size)
            probs_d = flat_p[n_models * 2 * n_total: n_models * 2 * (n_total + n_sel)].view(n_models, n_sel, 2)
            dec_d = flat_d[n_total: n_total + n_sel]
            classify_raw_ensemble(self.models, store.buf, off_d, len_d, lens_a, out=probs_d, decision=dec_d, max_len=max_len,
                                  threshold=threshold, mode=_MODE[mode])
            parts.append((lo + sel, lens_a, n_total, n_sel))
            n_total += n_sel
            t, dt = self._tick(t, 3)
            ph[3] += dt

        # slice k + 1 is staged, uploaded and scanned BEFORE the host waits for slice k's scan: the wait has the next
        # slice's host work in front of it, and the classification of slice k runs under the staging of slice k + 2.
        # With STAGE_THREAD that staging runs on a second host thread instead (see the class attribute: measured, no gain);
        # what the two threads share is the pinned scratch (locked), the poly(A) cache (slices hold different reads; dict
        # operations are atomic) and the store's row tables (begin_batch assigned every row; slices touch different rows).
        staged = upload_and_scan(0)
        t, ph[1] = self._tick(t, 1)
        for k in range(n_slices):
            ahead = None
            if k + 1 < n_slices:
                if self.STAGE_THREAD:
                    ahead = self._stager().submit(upload_and_scan, k + 1)
                else:                                      # one slice ahead on this thread (the round-4 order)
                    nxt = upload_and_scan(k + 1)
                    t, dt = self._tick(t, 1)
                    ph[1] += dt
            gate_and_classify(k, staged)
            if ahead is not None:
                staged = ahead.result()
                t, dt = self._tick(t, 1)                   # what the loop's thread still had to wait for the staging
                ph[1] += dt
            elif k + 1 < n_slices:
                staged = nxt
        store.end_batch()
        if n_total == 0:
            if side is not caller:
                caller.wait_stream(side)
            return None
        probs_h = self._res_probs.host.view(-1)[: n_models * 2 * n_total]
        dec_h = self._res_dec.host[:n_total]
        probs_h.copy_(flat_p[: n_models * 2 * n_total], non_blocking=True)
        dec_h.copy_(flat_d[:n_total], non_blocking=True)
        caller.synchronize()
        if side is not caller:
            side.synchronize()
        # half-precision models: an activation beyond 65504 leaves the conversion as +inf and the read's probabilities are
        # wrong - the reference's fp32 call (riser/model.py:22-28) cannot do that, so the loop says so (csrc: rs_model_saturated)
        for m in self.models:
            if getattr(m, "dtype", None) in getattr(m, "HALF_MODES", ()) and m.saturated(reset=True):
                self.saturated_batches += 1
                self.logger.warning(f"target {m.target!r} ({m.dtype}): an activation overflowed half precision in this batch of "
                                    f"{n_total} reads - their probabilities are unreliable; load the model as 'bf16x3' or 'f32w'")
        pn, dn = probs_h.numpy(), dec_h.numpy()
        p_on = np.empty((n_total, n_models), dtype=np.float64)                              # [read][model]
        for _, _, at, n in parts:
            p_on[at: at + n] = pn[n_models * 2 * at: n_models * 2 * (at + n)].reshape(n_models, n, 2)[:, :, 1].T
        sel_all = np.concatenate([p[0] for p in parts])
        res = _Assessed(reads, sel_all, channels[sel_all], np.concatenate([p[1] for p in parts]), p_on, dn.copy())
        t, ph[4] = self._tick(t, 4)
        return res

    def _stager(self):
        """the host thread that stages slice k + 1 of a PromethION-scale batch (created at the first such batch)"""
        if self._stage_pool is None:
            self._stage_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="riser_amd-stage")
        return self._stage_pool

    @staticmethod
    def _tick(t_prev, _k):
        now = time.perf_counter()
        return now, now - t_prev

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
            writer = _RowWriter(sink, self._csv_rows)
            try:
                while client.is_running() and time.monotonic() < t_stop:
                    cache = self._run_batch(writer, mode, threshold, unblock_duration, cache, tally)
            finally:
                writer.close()                                   # every row is in the file before target() returns
            client.send_warning(_WARN_STOP)
            if not client.is_running():
                self.logger.info("Client has stopped.")
            if time.monotonic() > t_stop:
                self.logger.info(f"RISER has timed out after {duration_h} hours as requested.")

    def _run_batch(self, writer, mode, threshold, unblock_duration, cache, tally):
        t0 = time.monotonic()
        self._ph[:] = 0.0
        res = self.assess_batch(list(self.client.get_read_batch()), mode, threshold, cache)
        n_acc = n_rej = 0
        t = time.perf_counter()
        if res is None:
            self.client.reject_reads([], unblock_duration)
            self.client.finish_processing_reads([])
        else:
            # the decided reads, in batch order per list: rejects are sent first, then every decided read is finished
            # (riser/control.py:85-90,106-112); "try_again" reads stay with the client
            reads, dec = res.reads, res.decision
            codes = (nv.RS_REJECT, nv.RS_ACCEPT, nv.RS_NO_DECISION)
            if _hp is not None and type(reads) is list:
                rejected, accepted, undecided = _hp.decided(reads, np.ascontiguousarray(res.sel, dtype=np.int64),
                                                            np.ascontiguousarray(res.channels, dtype=np.int64),
                                                            np.ascontiguousarray(dec, dtype=np.uint8), codes)
            else:
                key = self._client_key
                chan, sel = res.channels.tolist(), res.sel.tolist()
                rejected, accepted, undecided = ([(chan[k], key(reads[sel[k]])) for k in np.flatnonzero(dec == code).tolist()]
                                                 for code in codes)
            n_acc, n_rej = len(accepted), len(rejected)
            self.client.reject_reads(rejected, unblock_duration)
            self.client.finish_processing_reads(rejected + accepted + undecided)
            self.batch_latencies.append(time.monotonic() - t0)
        t, self._ph[5] = self._tick(t, 5)
        if res is not None:
            if len(res) >= self.CSV_THREAD_MIN_READS:
                writer.put(res, t0, mode, threshold)             # formatted and written by the writer thread
            else:
                writer.write_now(res, t0, mode, threshold)       # a MinION-sized batch: 0.1 ms inline
            t, self._ph[6] = self._tick(t, 6)
            self.batch_loop_times.append(time.monotonic() - t0)
            self.batch_phases.append(self._ph.copy())
        # riser/control.py:96-97 drops the cache at 1000 entries, two flow cells' worth of reads at MinION's 512 channels;
        # kept at exactly 1000 up to that channel count, the same proportion beyond it (the cache never changes a result,
        # only how often a read is re-scanned)
        limit = _CACHE_LIMIT if self._channels_seen <= _MINION_CHANNELS else 2 * self._channels_seen
        if len(cache) >= limit:
            cache = {}
        tally.add(0 if res is None else len(res), n_acc, n_rej)
        tally.report_if_due(t0)
        return cache

    def _csv_rows(self, res: _Assessed, t0: float, mode: str, threshold) -> str:
        """the rows of riser/control.py:145-153 for one batch, as one string"""
        head = f"{t0:.0f},"
        mid = "," + ";".join(m.target for m in self.models) + ","
        tail = f",{threshold},{mode},"
        if _hp is not None:
            return _hp.format_rows(head, res.reads, np.ascontiguousarray(res.sel, dtype=np.int64),
                                   np.ascontiguousarray(res.channels, dtype=np.int64),
                                   np.ascontiguousarray(res.n_samples, dtype=np.int32), mid, res.p_on, int(res.p_on.shape[1]),
                                   tail, np.ascontiguousarray(res.decision, dtype=np.uint8), tuple(nv.DECISION_NAMES))
        lines = [f"{head}{read.id},{channel},{n}{mid}{';'.join(map(str, p))}{tail}{decision}"
                 for channel, read, n, p, decision in res.records()]
        return "\n".join(lines) + "\n"

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
