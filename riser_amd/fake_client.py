"""Scripted stand-in for the MinKNOW ReadUntil client (riser/client.py:25-69).

The real client needs sequencing hardware and the un-vendored `read_until` package, so
tests and the replay harness drive SequencerControl with this duck-type instead: the
same eight methods the control loop calls (riser/control.py:12,25,31,33,100,106,127,131)
over a pre-scripted list of batches.
"""
from __future__ import annotations

import numpy as np


class FakeRead:
    """Shape of a ReadUntil read chunk: `.id`, `.raw_data` (bytes) and, for
    minknow-api <= 5, `.number` (riser/control.py:137-143)."""

    def __init__(self, read_id: str, signal, number=None):
        """signal: an int16 array (copied into bytes, as MinKNOW delivers them) or a bytes-like object that already
        holds little-endian int16 samples (kept as is: the replay harness hands out memoryview slices of one buffer)"""
        self.id = read_id
        if isinstance(signal, (bytes, bytearray, memoryview)):
            self.raw_data = signal
        else:
            self.raw_data = np.ascontiguousarray(signal, dtype=np.int16).tobytes()
        if number is not None:
            self.number = number


class FakeClient:
    signal_dtype = np.int16

    def __init__(self, batches):
        """batches: list of lists of (channel, FakeRead)."""
        self._batches = list(batches)
        self._next = 0
        self.started = False
        self.was_reset = False
        self.warnings = []
        self.rejected = []          # one list per batch (possibly empty)
        self.finished = []
        self.unblock_durations = []

    # --- the eight methods of riser/client.py ---------------------------------------
    def start_streaming_reads(self):
        self.started = True

    def is_running(self):
        return self._next < len(self._batches)

    def get_read_batch(self):
        b = self._batches[self._next]
        self._next += 1
        return b

    def get_raw_signal(self, read):
        return np.frombuffer(read.raw_data, self.signal_dtype)

    def reject_reads(self, reads, unblock_duration):
        self.rejected.append(list(reads))
        self.unblock_durations.append(unblock_duration)

    def finish_processing_reads(self, reads):
        self.finished.append(list(reads))

    def reset(self):
        self.was_reset = True

    def send_warning(self, message):
        self.warnings.append(message)
