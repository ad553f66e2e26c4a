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

import numpy as np
import torch

from . import _native as nv
from .model import classify_raw_ensemble
from .preprocess import pack_reads

_MODE = {"enrich": nv.RS_ENRICH, "deplete": nv.RS_DEPLETE}


class SequencerControl():
    def __init__(self, client, models, processor, logger, out_file):
        self.client = client
        self.models = models
        self.proc = processor
        self.logger = logger
        self.out_filename = out_file
        self.batch_latencies = []          # host wall time per assessed batch (seconds)

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
        if mode not in _MODE:
            raise ValueError(f"mode must be 'enrich' or 'deplete', got {mode!r}")
        self.client.send_warning(
            'The sequencing run is being controlled by RISER, reads that are '
            'not in the target class will be ejected from the pore.')

        with open(f'{self.out_filename}.csv', 'a') as out_file:
            self._write_header(out_file)
            run_start = time.monotonic()
            progress_time = run_start + 60
            duration_s = self._hours_to_seconds(duration_h)
            n_assessed = 0
            n_rejected = 0
            n_accepted = 0
            polyA_cache = {}
            while self.client.is_running() and time.monotonic() < run_start + duration_s:
                batch_start = time.monotonic()
                reads_to_reject = []
                reads_to_accept = []
                reads_unclassified = []
                entries = list(self.client.get_read_batch())
                records = self.assess_batch(entries, mode, threshold, polyA_cache)
                for channel, read, sig_len, p_on_targets, decision in records:
                    n_assessed += 1
                    if decision == "accept":
                        reads_to_accept.append((channel, self._get_read_id(read)))
                    elif decision == "reject":
                        reads_to_reject.append((channel, self._get_read_id(read)))
                    elif decision == "no_decision":
                        reads_unclassified.append((channel, self._get_read_id(read)))
                    self._write(out_file, batch_start, channel, read.id, sig_len, self.models,
                                p_on_targets, threshold, mode, decision)
                if len(polyA_cache) >= 1000:
                    polyA_cache = {}

                # Send reject requests
                self.client.reject_reads(reads_to_reject, unblock_duration)
                n_rejected += len(reads_to_reject)

                # Rejected, accepted and max-length reads need no reassessment
                done = reads_to_reject + reads_to_accept + reads_unclassified
                self.client.finish_processing_reads(done)
                n_accepted += len(reads_to_accept)
                if records:
                    self.batch_latencies.append(time.monotonic() - batch_start)

                if batch_start > progress_time:
                    self.logger.info(f"In the last minute {n_assessed} signals "
                                     f"were assessed, {n_accepted} were "
                                     f"accepted and {n_rejected} were rejected")
                    n_assessed = 0
                    n_rejected = 0
                    n_accepted = 0
                    progress_time = batch_start + 60
            else:
                self.client.send_warning('RISER has stopped running.')
                if not self.client.is_running():
                    self.logger.info('Client has stopped.')
                if time.monotonic() > run_start + duration_s:
                    self.logger.info(f'RISER has timed out after {duration_h} '
                                     'hours as requested.')

    def start(self):
        self.client.start_streaming_reads()
        self.logger.info('Live read stream started.')

    def finish(self):
        self.client.reset()
        self.logger.info('Client reset and live read stream ended.')

    def _hours_to_seconds(self, hours):
        return hours * 60 * 60

    def _get_read_id(self, read):
        # minknow-api <= v5 exposes .number, >= v6 only .id (riser/control.py:137-143)
        if hasattr(read, "number"):
            return read.number
        return read.id

    def _write_header(self, csv_file):
        csv_file.write('batch_start,read_id,channel,sig_length,models,prob_targets,threshold,mode,decision\n')

    def _write(self, csv_file, batch_start, channel, read, sig_length,
               models, p_on_targets, threshold, mode, decision):
        csv_file.write(f'{batch_start:.0f},{read},{channel},{sig_length},'
                       f'{";".join([m.target for m in models])},'
                       f'{";".join([str(float(p)) for p in p_on_targets])},'
                       f'{threshold},{mode},{decision}\n')
