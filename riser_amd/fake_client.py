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
    # get_raw_signal below IS np.frombuffer(read.raw_data, int16), as riser/client.py:46-47 is: declaring it lets the
    # control loop walk the reads of a batch with the C loops of riser_amd/_hostpack instead of one Python call per read
    raw_data_dtype = np.int16

    def __init__(self, batches, first_channel: int | None = None, last_channel: int | None = None):
        """batches: list of lists of (channel, FakeRead).  first_channel / last_channel: the channel range this client
        streams, as `ReadUntilClient.run(first_channel=, last_channel=)` takes it (riser/client.py:33-38): reads of other
        channels are not delivered (one rank of a sharded run owns one range, riser_amd/launch.py)."""
        self._batches = list(batches)
        self.first_channel, self.last_channel = first_channel, last_channel
        self._next = 0
        self.started = False
        self.was_reset = False
        self.warnings = []
        self.rejected = []          # one list per batch (possibly empty)
        self.finished = []
        self.unblock_durations = []

    def extend(self, batches):
        """more scripted batches behind the ones already played (the loop's is_running() turns true again)"""
        self._batches.extend(batches)

    # --- the eight methods of riser/client.py ---------------------------------------
    def start_streaming_reads(self):
        self.started = True

    def is_running(self):
        return self._next < len(self._batches)

    def get_read_batch(self):
        b = self._batches[self._next]
        self._next += 1
        if self.first_channel is not None or self.last_channel is not None:
            lo = self.first_channel if self.first_channel is not None else -(1 << 62)
            hi = self.last_channel if self.last_channel is not None else 1 << 62
            b = [e for e in b if lo <= e[0] <= hi]
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


class PlainFakeClient(FakeClient):
    """The same client WITHOUT the raw_data_dtype declaration: the control loop then uses nothing but the eight methods
    (one get_raw_signal call per read), as it must for a client it knows nothing about."""
    raw_data_dtype = None
