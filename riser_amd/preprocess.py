"""Kit / SignalProcessor: drop-in for riser/preprocess.py:15-147 with the arithmetic on the GPU.

Same constructor, same nine methods, same argument meaning and error behaviour
(`mad_normalise` raises ValueError on an empty signal, riser/preprocess.py:109-110).
The median / MAD selection, the normalisation, the outlier smoothing and the poly(A)
window scan run in HIP kernels (riser_amd/csrc/normalise.hip, polya.hip) through the C
ABI; results are bit-identical to the reference's float64 numpy path.  Batched variants
(`mad_normalise_batch`, `get_polyA_end_batch`) are what the batched control loop uses.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _native as nv

# numbers of the reference (riser/preprocess.py:5-13); the kernels hold their own copies of the first two
OUTLIER_SIGMA = 3.5              # |y| above this is an outlier
MAD_TO_SIGMA = 1.4826            # MAD -> standard deviation of a normal distribution
MIN_SAMPLES = 4096               # 12 MaxPool(2) layers need 2^12 samples for one output position
MAX_NT = 280                     # longest prefix assessed, in nucleotides
FIXED_TRIM_NT = 150.6            # adapter + poly(A) estimate when no poly(A) end is found
MAX_SIGNAL = 65536               # LDS staging limit of the normalise kernel

_KITS = {"RNA002": (3012, 70), "RNA004": (4000, 130)}       # sampling rate [Hz], translocation rate [nt/s]


class Kit:
    """Sequencing chemistry (riser/preprocess.py:15-30): converts nucleotides to samples."""

    def __init__(self, sampling_hz, transloc_rate):
        self.sampling_hz, self.transloc_rate = sampling_hz, transloc_rate

    @classmethod
    def create_from_version(cls, version):
        try:
            return cls(*_KITS[version])
        except KeyError:
            raise Exception(f"Invalid kit version {version}") from None

    def samples(self, nt) -> int:
        """nt nucleotides in samples, truncated as the reference does."""
        return int(nt / self.transloc_rate * self.sampling_hz)


def _as_int16(signal) -> np.ndarray:
    a = np.asarray(signal)
    if a.dtype == np.int16:
        return np.ascontiguousarray(a)
    if np.issubdtype(a.dtype, np.integer):
        if a.size and (a.min() < -32768 or a.max() > 32767):
            raise TypeError("raw signal does not fit int16 ADC counts")
        return np.ascontiguousarray(a.astype(np.int16))
    raise TypeError("riser_amd normalises raw int16 ADC signals (riser/client.py:47); "
                    f"got dtype {a.dtype}")


def pack_reads(signals, device):
    """Concatenate int16 reads into one device buffer -> (sig, off, len tensors, lens_host)."""
    sigs = [_as_int16(s) for s in signals]
    lens = np.array([s.shape[0] for s in sigs], dtype=np.int32)
    offs = np.zeros(len(sigs), dtype=np.int64)
    if len(sigs) > 1:
        offs[1:] = np.cumsum(lens[:-1], dtype=np.int64)
    flat = np.concatenate(sigs) if sigs else np.zeros(0, dtype=np.int16)
    if flat.size == 0:
        flat = np.zeros(1, dtype=np.int16)
    return (torch.from_numpy(flat).to(device), torch.from_numpy(offs).to(device),
            torch.from_numpy(lens).to(device), lens)


class SignalProcessor:
    def __init__(self, kit, device=None):
        self.kit = kit
        nv.require_gpu()
        if not torch.cuda.is_available():
            raise nv.NativeError("torch sees no ROCm device")
        d = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device("cuda", d.index if d.index is not None else torch.cuda.current_device())

    # ---- lengths (riser/preprocess.py:33-40,81-85) ------------------------------------------
    def get_min_length(self):
        return MIN_SAMPLES

    def get_max_length(self):
        return self.kit.samples(MAX_NT)

    def is_max_length(self, signal):
        return self.get_max_length() <= len(signal)

    def get_fixed_trim_length(self):
        return self.kit.samples(FIXED_TRIM_NT)

    def should_trim_fixed_length(self, signal):
        # long enough to leave a full-length window after the fixed trim
        return len(signal) - self.get_fixed_trim_length() > self.get_max_length()

    def trim_polyA_fixed_length(self, signal):
        return signal[self.get_fixed_trim_length():]

    # ---- poly(A) (riser/preprocess.py:42-79,87-102) ---------------------------------------
    def get_polyA_end(self, signal):
        end = self.get_polyA_end_batch([signal])[0]
        return None if end < 0 else int(end)

    def get_polyA_end_batch(self, signals) -> np.ndarray:
        """int32 [B]: end index or -1 where the reference returns None."""
        if len(signals) == 0:
            return np.zeros(0, dtype=np.int32)
        sig, off, ln, lens = pack_reads(signals, self.device)
        return self.polyA_end_device(sig, off, ln, len(signals)).cpu().numpy()

    def polyA_end_device(self, sig, off, ln, B, state_in: torch.Tensor = None):
        """-> int32 [B] ends on the device (-1: none).  With `state_in` (int32 [B, 4] on the device: the state a previous scan
        of the SAME reads returned, zeros for new ones) only the windows not yet scanned are: -> (ends, state_out)."""
        out = torch.empty(B, dtype=torch.int32, device=self.device)
        if state_in is None:
            nv.check(nv.lib().rs_polya_end(sig.data_ptr(), off.data_ptr(), ln.data_ptr(), B, out.data_ptr(),
                                           torch.cuda.current_stream(self.device).cuda_stream), "rs_polya_end")
            return out
        state_out = torch.empty((B, 4), dtype=torch.int32, device=self.device)
        nv.check(nv.lib().rs_polya_end_resume(sig.data_ptr(), off.data_ptr(), ln.data_ptr(), B, state_in.data_ptr(),
                                              out.data_ptr(), state_out.data_ptr(),
                                              torch.cuda.current_stream(self.device).cuda_stream), "rs_polya_end_resume")
        return out, state_out

    def trim_polyA(self, signal, read_id, cache):
        """-> (signal without adapter + poly(A), True) when a poly(A) end is known for the read, else (signal, False)
        (riser/preprocess.py:87-102).  Ends that were found are memoised per read id in `cache`."""
        end = cache.get(read_id)
        if end is None:
            end = self.get_polyA_end(signal)
            if end:
                cache[read_id] = end
        if not end:                      # None, or the reference's "0 counts as not found"
            return signal, False
        return signal[end + 1:], True

    # ---- normalisation (riser/preprocess.py:108-147) --------------------------------------
    def mad_normalise(self, signal):
        if np.asarray(signal).shape[0] == 0:
            raise ValueError("Signal must not be empty")
        if np.issubdtype(np.asarray(signal).dtype, np.floating):
            # pA-scaled float signals (riser/retrain/preprocess.py:79): numpy keeps the input's precision end to end,
            # float32 in -> float32 out (riser/preprocess.py:108-115 under NEP 50), and so does the float kernel
            out, stats = self.mad_normalise_float_batch([signal], return_stats=True)
            return np.zeros(out[0].shape[0], dtype=np.int64) if stats[0, 1] == 0 else out[0]
        out, stats = self.mad_normalise_batch([signal], return_stats=True)
        if stats[0, 1] == 0:
            # np.vectorize over Python ints yields an int64 zero array (riser/preprocess.py:122-125)
            return np.zeros(out[0].shape[0], dtype=np.int64)
        return out[0]

    def mad_normalise_batch(self, signals, return_stats: bool = False):
        """List of raw int16 reads -> list of float64 arrays (bit-exact with the reference),
        optionally with the [B, 2] (median, mad) table."""
        if any(np.asarray(s).shape[0] == 0 for s in signals):
            raise ValueError("Signal must not be empty")
        sig, off, ln, lens = pack_reads(signals, self.device)
        B, lmax = len(signals), int(lens.max())
        if lmax > MAX_SIGNAL:
            raise ValueError(f"signal longer than {MAX_SIGNAL} samples")
        out64 = torch.empty((B, lmax), dtype=torch.float64, device=self.device)
        stats = torch.empty((B, 2), dtype=torch.float64, device=self.device)
        nv.check(nv.lib().rs_normalise(sig.data_ptr(), off.data_ptr(), ln.data_ptr(), B, lmax, None, 0, 0,
                                       out64.data_ptr(), lmax, stats.data_ptr(),
                                       torch.cuda.current_stream(self.device).cuda_stream), "rs_normalise")
        host = out64.cpu().numpy()
        res = [host[i, : lens[i]].copy() for i in range(B)]
        return (res, stats.cpu().numpy()) if return_stats else res

    def mad_normalise_float_batch(self, signals, return_stats: bool = False):
        """List of float16 / float32 / float64 signals of one dtype -> list of arrays of that dtype, bit-identical to the
        reference's mad_normalise on the same input (rs_normalise_float)."""
        arrs = [np.ascontiguousarray(s) for s in signals]
        dt = arrs[0].dtype
        if dt not in (np.float16, np.float32, np.float64) or any(a.dtype != dt for a in arrs):
            raise TypeError("riser_amd normalises float16, float32 or float64 signals (one dtype per batch); "
                            f"got {sorted({str(a.dtype) for a in arrs})}")
        if any(a.shape[0] == 0 for a in arrs):
            raise ValueError("Signal must not be empty")
        lens = np.array([a.shape[0] for a in arrs], dtype=np.int32)
        offs = np.zeros(len(arrs), dtype=np.int64)
        offs[1:] = np.cumsum(lens[:-1], dtype=np.int64)
        tdt = {2: torch.float16, 4: torch.float32, 8: torch.float64}[dt.itemsize]
        dev = self.device
        sig = torch.from_numpy(np.concatenate(arrs)).to(dev)
        B, lmax = len(arrs), int(lens.max())
        out = torch.empty((B, lmax), dtype=tdt, device=dev)
        stats = torch.empty((B, 2), dtype=torch.float64, device=dev)
        off_d, len_d = torch.from_numpy(offs).to(dev), torch.from_numpy(lens).to(dev)     # keep them alive over the call
        nv.check(nv.lib().rs_normalise_float(sig.data_ptr(), dt.itemsize, off_d.data_ptr(), len_d.data_ptr(), B,
                                             out.data_ptr(), lmax,
                                             stats.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                 "rs_normalise_float")
        host = out.cpu().numpy()
        res = [host[i, : lens[i]].copy() for i in range(B)]
        return (res, stats.cpu().numpy()) if return_stats else res

    def normalise_device(self, sig, off, ln, B, lmax, pad_to=None) -> torch.Tensor:
        """Raw reads already on the device -> fp32 [B, pad_to] normalised signals (zero
        padded), i.e. what Model.classify would receive after its fp32 cast."""
        pad_to = int(pad_to or lmax)
        out = torch.empty((B, pad_to), dtype=torch.float32, device=self.device)
        nv.check(nv.lib().rs_normalise(sig.data_ptr(), off.data_ptr(), ln.data_ptr(), B, int(lmax),
                                       out.data_ptr(), pad_to, pad_to, None, 0, None,
                                       torch.cuda.current_stream(self.device).cuda_stream), "rs_normalise")
        return out
