"""SequencerControl: the ReadUntil control loop of riser/control.py:3-153, batched.

Same constructor, `start()`, `target(mode, duration_h, threshold, unblock_duration)`,
`finish()`, same CSV columns, same decision rule and the same client calls in the same
order as the reference.  The difference is the shape of the work: the reference walks the
(<= 512) reads of a ReadUntil batch one at a time - trim, normalise, one forward per model at
batch 1, a device sync per probability (riser/control.py:31-93,152) - whereas this loop

  1. uploads the raw int16 signals of the whole batch once,
  2. finds the poly(A) ends of all un-cached reads in one kernel launch,
  3. applies the reference's length gating on the host (pure index arithmetic: a trim is an
     offset into the uploaded buffer, a truncation is a length),
  4. normalises every assessable read once and runs one batched forward per model,
  5. takes the ensemble decision on the device and copies probabilities + decisions back in
     a single transfer.

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
from .preprocess import pack_reads

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


class SequencerControl:
    def __init__(self, client, models, processor, logger, out_file):
        self.client, self.models, self.proc, self.logger = client, models, processor, logger
        self.out_filename = out_file
        # host wall time of the most recent assessed batches (seconds): bounded, a run lasts tens of hours
        self.batch_latencies = deque(maxlen=4096)

    # ------------------------------------------------------------------------------------
    def assess_batch(self, entries, mode, threshold, polyA_cache):
        """entries: list of (channel, read).  Returns one record per ASSESSED read, in
        batch order: (channel, read, sig_length, [p_on per model], decision str)."""
        if not entries:
            return []
        proc = self.proc
        dev = proc.device
        signals = [self.client.get_raw_signal(read) for _, read in entries]
        sig, off, ln, lens = pack_reads(signals, dev)
        offs_host = np.zeros(len(signals), dtype=np.int64)
        if len(signals) > 1:
            offs_host[1:] = np.cumsum(lens[:-1], dtype=np.int64)

        # -- poly(A) end for reads not in the cache: one launch -------------------------------
        need = [i for i, (_, read) in enumerate(entries) if read.id not in polyA_cache]
        ends = {}
        if need:
            idx = torch.from_numpy(np.asarray(need, dtype=np.int64)).to(dev)
            found = proc.polyA_end_device(sig, off[idx].contiguous(), ln[idx].contiguous(), len(need)).cpu().numpy()
            for i, e in zip(need, found):
                ends[i] = int(e)

        # -- gating (riser/control.py:36-60) as offsets / lengths ------------------------------
        max_len, min_len = proc.get_max_length(), proc.get_min_length()
        fixed = proc.get_fixed_trim_length()
        sel, a_off, a_len = [], [], []
        for i, (_, read) in enumerate(entries):
            n = int(lens[i])
            if read.id in polyA_cache:
                end = polyA_cache[read.id]
            else:
                end = ends[i] if ends[i] > 0 else None
                if end:
                    polyA_cache[read.id] = end
            if not end:
                if n > fixed + max_len:                              # should_trim_fixed_length
                    start, length = fixed, min(n - fixed, max_len)
                else:
                    continue
            else:
                start, length = end + 1, n - (end + 1)
                if length < min_len:
                    continue
                length = min(length, max_len)
            sel.append(i)
            a_off.append(int(offs_host[i]) + start)
            a_len.append(length)
        if not sel:
            return []

        # -- normalise once, one batched forward per model, decision on the device -------------
        B = len(sel)
        lens_a = np.asarray(a_len, dtype=np.int32)
        off_d = torch.from_numpy(np.asarray(a_off, dtype=np.int64)).to(dev)
        len_d = torch.from_numpy(lens_a).to(dev)
        # one library call: normalise once, one forward per model, decision on the device
        dec = torch.empty(B, dtype=torch.uint8, device=dev)
        probs = classify_raw_ensemble(self.models, sig, off_d, len_d, lens_a, decision=dec, max_len=max_len,
                                      threshold=threshold, mode=_MODE[mode])
        probs_h = probs.cpu().numpy()
        dec_h = dec.cpu().numpy()
        out = []
        for j, i in enumerate(sel):
            channel, read = entries[i]
            out.append((channel, read, int(lens_a[j]), [probs_h[m, j, 1] for m in range(len(self.models))],
                        nv.DECISION_NAMES[int(dec_h[j])]))
        return out

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
        targets = ";".join(m.target for m in self.models)
        for channel, read, n_samples, p_on, decision in records:
            if decision in decided:
                decided[decision].append((channel, self._client_key(read)))
            probs = ";".join(str(float(p)) for p in p_on)
            sink.write(f"{t0:.0f},{read.id},{channel},{n_samples},{targets},{probs},{threshold},{mode},{decision}\n")
        if len(cache) >= _CACHE_LIMIT:
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
