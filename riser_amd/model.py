"""Model: drop-in for riser/model.py:6-32 on MI355X, plus the batched entry points.

`Model(state, config, logger, target)` and `classify(signal) -> Tensor[2]` keep the
reference's signature and meaning (probabilities in the order (p_off_target, p_on_target),
riser/control.py:69).  The forward pass itself is the HIP library (riser_amd/csrc) reached
through the C ABI; torch is used only for device memory and the current stream.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _native as nv

_DTYPES = {"f32": nv.RS_F32, "fp32": nv.RS_F32, "float32": nv.RS_F32,
           "bf16": nv.RS_BF16, "bfloat16": nv.RS_BF16, "f16": nv.RS_F16, "fp16": nv.RS_F16,
           "float16": nv.RS_F16, "f32w": nv.RS_F32W, "f32_winograd": nv.RS_F32W,
           "bf16x3": nv.RS_BF16X3, "f16x3": nv.RS_F16X3, "f16xf8": nv.RS_F16XF8}


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class Workspace:
    """Grow-only device scratch owned by the caller side of the ABI: ONE BUFFER PER STREAM, so that calls issued on
    different HIP streams (two batches in flight: stream.classify_resident) never share activations.  Keyed by the
    stream that is current when the call is made; a library call only ever touches the workspace it is handed."""

    MAX_STREAMS = 8          # workspaces kept alive: a caller cycling through more streams than this re-allocates

    def __init__(self, device):
        self.device = device
        self._bufs = {}

    def get(self, nbytes: int) -> torch.Tensor:
        key = _stream_ptr(self.device)
        buf = self._bufs.pop(key, None)                      # re-inserted below: the dict's order is the order of last use
        if buf is None or buf.numel() < nbytes:
            buf = None                                       # release the smaller one before the new one is allocated
            if len(self._bufs) >= self.MAX_STREAMS:          # drop the workspace of the stream used longest ago
                del self._bufs[next(iter(self._bufs))]
            with torch.cuda.stream(torch.cuda.current_stream(self.device)):
                buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._bufs[key] = buf
        return buf


class Model:
    @staticmethod
    def dtypes():
        """Arithmetic modes this build of the library accepts (canonical names)."""
        return ("f32w", "f32", "bf16", "f16", "bf16x3", "f16x3", "f16xf8")

    HALF_MODES = ("f16", "f16x3", "f16xf8")        # activations stored as IEEE half: a range of 65504
    HALF_MAX = 65504.0

    def __init__(self, state, config, logger, target, dtype: str = "f32w", device=None, range_check: bool = True):
        """dtype: "f32w" (default; fp32 end to end, conv layers as Winograd F(2,3) / F(4,3) on the
        f32-input MFMA), "f32" (fp32, direct lowering: exact fmaf chains), "bf16x3" / "f16x3" (split
        precision on the 16-bit MFMA: hi + lo pairs, three MFMAs per product, fp32 accumulate - the 16-bit mode
        that stays within 1e-3 of the reference), "f16xf8" (f16x3 whose wide layers evaluate the two cross terms of the
        split product as one block-scaled e4m3 product on the 8-bit MFMA: two instruction times per product instead of
        three, within ~3e-4 of the reference), "f16" / "bf16" (plain 16-bit activations and weights, fp32
        accumulate: fast, approximate).
        range_check (half-precision modes only): run a small synthetic sample through the new model and REFUSE (ValueError)
        when an activation comes within a factor of 4 of half precision's 65504 - those modes cannot represent larger
        activations, the reference's fp32 path can (riser/model.py:22-28).  RS_RANGE_CHECK=0 or range_check=False skip it;
        at run time `saturated()` (and a warning from `classify`) report an overflow on the real data."""
        self.target = target
        self.logger = logger
        self.device = self._get_device(device)
        if logger is not None:
            logger.info('Using %s device', self.device)
        if getattr(config, "model", None) == "resnet" or (not hasattr(config, "cnn") and hasattr(config, "resnet")):
            # `config.resnet` (riser/nets/resnet.py:72-99: channels, kernel, padding, stride, block, n_layers, blocks,
            # n_classes) instead of `config.cnn`: the reference's second architecture.  Its own Model hard-wires ConvNet
            # (riser/model.py:13) and ships no ResNet config or weights; here the same Model surface - classify, the batched
            # entry points, SequencerControl - runs it as a generic conv program (csrc/seqnet.hip: one launch for the stem
            # and one per residual block), in fp32 or, dtype "bf16x3", with stem and basic blocks on the bf16 MFMA.
            self._init_resnet(state, config.resnet, dtype)
            return
        cnn = config.cnn
        self.classifier = getattr(cnn, "classifier", "gap_fc")
        if self.classifier not in ("gap_fc", "gap", "fc"):
            raise ValueError(f"classifier {self.classifier!r}: the reference knows `fc`, `gap_fc` and `gap` "
                             "(riser/nets/cnn.py:21-41)")
        self._fc_positions = 0
        if isinstance(state, dict):
            sd = state
        else:
            sd = torch.load(state, map_location="cpu")              # riser/model.py:19
        sd = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in sd.items()}
        if self.classifier == "gap":
            # Conv1d(C, n_classes, 1) then the mean over positions (riser/nets/cnn.py:34-38) = the mean, then the same
            # matrix: W mean_p(x_p) + b.  Runs as the gap_fc head with the 1 x 1 kernel read as a matrix (fp32 summation
            # order differs from the reference's, ~1e-7 on a probability).
            w0 = np.asarray(sd["classifier.0.weight"])
            if w0.ndim != 3 or w0.shape[2] != 1:
                raise ValueError("classifier.0.weight: expected [n_classes, channels, 1] for the `gap` classifier")
            sd["classifier.2.weight"] = np.ascontiguousarray(w0[:, :, 0])
            sd["classifier.2.bias"] = np.asarray(sd["classifier.0.bias"])
        self.channels = [int(c) for c in cnn.channels][: int(cnn.n_layers)]
        self.n_layers = len(self.channels)
        fc = None
        if self.classifier == "fc":
            # Flatten -> Linear(C * P, H) -> ReLU -> Linear(H, n_classes) (riser/nets/cnn.py:22-27; the reference hard-codes
            # C * P = 67 * 753 and H = 4096: one input length of one 4-layer net).  Runs behind the conv stack as two extra
            # kernels (csrc/fc_head.hip); the gap_fc weights the library is created with are placeholders.
            if int(getattr(cnn, "depth", 1)) != 1 or any(int(k) != 3 for k in list(cnn.kernels)[: self.n_layers]) \
                    or dtype not in ("f32w", "f32"):
                raise ValueError("the `fc` classifier runs behind the depth-1, kernel-3 conv stack in fp32 (dtype f32w / f32)")
            w1 = np.ascontiguousarray(sd["classifier.1.weight"], dtype=np.float32)
            c_last = self.channels[-1]
            if w1.ndim != 2 or w1.shape[1] % c_last != 0:
                raise ValueError(f"classifier.1.weight {w1.shape}: in_features is not a multiple of the last layer's "
                                 f"{c_last} channels")
            fc = (w1, np.ascontiguousarray(sd["classifier.1.bias"], dtype=np.float32),
                  np.ascontiguousarray(sd["classifier.3.weight"], dtype=np.float32),
                  np.ascontiguousarray(sd["classifier.3.bias"], dtype=np.float32))
            if fc[1].shape != (w1.shape[0],) or fc[2].shape != (int(cnn.n_classes), w1.shape[0]) or fc[3].shape != (int(cnn.n_classes),):
                raise ValueError("classifier.1 / classifier.3 shapes do not match each other or n_classes")
            sd["classifier.2.weight"] = np.zeros((int(cnn.n_classes), c_last), dtype=np.float32)
            sd["classifier.2.bias"] = np.zeros(int(cnn.n_classes), dtype=np.float32)
        self.min_length = 1 << self.n_layers
        self.dtype = dtype
        self._keep = []
        self._seq = None
        self.model = self            # the reference exposes the nn.Module here; kept as an alias
        if int(getattr(cnn, "depth", 1)) != 1 or any(int(k) != 3 for k in list(cnn.kernels)[: self.n_layers]):
            # outside the shipped class (depth > 1 or kernels other than 3, riser/nets/cnn.py:17,52-65): the generic
            # conv / max-pool program of csrc/seqnet.hip (f32-input MFMA, reads grouped by length)
            from .resnet import SeqNet, build_convnet_program
            if int(cnn.n_classes) != 2:
                raise ValueError("riser_amd supports two-class heads only")
            self._seq = SeqNet(*build_convnet_program(sd, cnn), device=self.device)
            self._h = None
            if dtype not in ("f32w", "f32"):
                raise ValueError(f"dtype {dtype!r}: configs with depth > 1 or kernels other than 3 run the generic fp32 "
                                 "conv program only")
            self.dtype = "f32"
            self._ws = Workspace(self.device)
            return
        conv_w, conv_b = [], []
        c_in = 1
        for i, c_out in enumerate(self.channels):
            w = np.ascontiguousarray(sd[f"layers.{i}.0.weight"], dtype=np.float32)
            b = np.ascontiguousarray(sd[f"layers.{i}.0.bias"], dtype=np.float32)
            if w.shape != (c_out, c_in, 3) or b.shape != (c_out,):
                raise ValueError(f"layers.{i}.0: state dict shape {w.shape} does not match config")
            conv_w.append(w)
            conv_b.append(b)
            c_in = c_out
        fc_w = np.ascontiguousarray(sd["classifier.2.weight"], dtype=np.float32)
        fc_b = np.ascontiguousarray(sd["classifier.2.bias"], dtype=np.float32)
        if fc_w.shape != (int(cnn.n_classes), c_in):
            raise ValueError("classifier.2.weight shape does not match config")
        L = nv.lib()
        wp = (C.c_void_p * self.n_layers)(*[w.ctypes.data for w in conv_w])
        bp = (C.c_void_p * self.n_layers)(*[b.ctypes.data for b in conv_b])
        ch = (C.c_int32 * self.n_layers)(*self.channels)
        h = C.c_void_p()
        nv.check(L.rs_model_create(self.n_layers, ch, int(cnn.n_classes), wp, bp, fc_w.ctypes.data,
                                   fc_b.ctypes.data, _DTYPES[dtype], self.device.index, C.byref(h)),
                 "rs_model_create")
        self._h = h
        self._ws = Workspace(self.device)
        if fc is not None:
            w1, b1, w2, b2 = fc
            positions = w1.shape[1] // self.channels[-1]
            nv.check(L.rs_model_set_fc_classifier(h, positions, int(w1.shape[0]), w1.ctypes.data, b1.ctypes.data,
                                                  w2.ctypes.data, b2.ctypes.data), "rs_model_set_fc_classifier")
            self._fc_positions, self._fc_hidden = positions, int(w1.shape[0])
        if dtype in self.HALF_MODES and range_check and os.environ.get("RS_RANGE_CHECK", "1") != "0":
            try:
                self._half_range_check()
            except Exception:
                self.close()
                raise

    # ---- half precision's range ----------------------------------------------------------
    def half_activation_maxima(self, signals=None) -> list:
        """largest activation (the hi half) of every conv layer i >= 1 of THIS half-precision model on raw int16 `signals`
        (default: 16 synthetic reads of riser_amd.synth; MAD-normalised input is confined to ~[-3.5, 3.5] whatever the
        source), through the library's layer-capture hook.  inf = the layer overflowed."""
        from . import synth
        from .preprocess import pack_reads
        self._need_handle("half_activation_maxima")
        if signals is None:
            signals = [synth.make_raw_read(4242, rid, 2048 + 8615 + 37 * rid, polya=bool(rid % 5))[2048:] for rid in range(16)]
        lens = [len(s) for s in signals]
        sig, off, ln, lh = pack_reads(list(signals), self.device)
        info, out, L = self.layer_info(), [], nv.lib()
        try:
            for i in range(1, self.n_layers):
                U, bases = self.block_samples(i), self.block_bases(lens, i)
                rows, cp, fmt = int(bases[-1]) * (U >> (i + 1)), info[i]["cp_out"], info[i]["rows_format"]
                cap = torch.zeros(rows * cp, dtype=torch.float16, device=self.device)
                nv.check(L.rs_debug_capture_layer(self._h, i, cap.data_ptr(), cap.numel() * 2), "rs_debug_capture_layer")
                self.classify_raw(sig, off, ln, lh)
                v = cap.view(rows, cp)
                if fmt == 1:
                    v = v.view(rows, cp // 64, 2, 32)[:, :, 0, :]            # [hi x 32 | lo x 32]
                elif fmt == 2:
                    v = v.view(rows, cp // 128, 2, 64)[:, :, 0, :]           # [hi16 x 64 | e4m3 bytes]
                v = v.float().abs()
                out.append(float("inf") if not torch.isfinite(v).all() else float(v.max().item()))
        finally:
            L.rs_debug_capture_layer(self._h, -1, None, 0)
            self.saturated(reset=True)
        return out

    def _half_range_check(self, margin: float = 4.0):
        mx = self.half_activation_maxima()
        worst = max(mx)
        if not worst * margin < self.HALF_MAX:
            layer = 1 + mx.index(worst)
            raise ValueError(
                f"dtype {self.dtype!r} stores activations as IEEE half (largest value 65504): conv layer {layer} of target "
                f"{self.target!r} reaches {worst:.4g} on the synthetic sample, less than x{margin:g} inside that range. Use "
                "'bf16x3' (fp32's exponent range, within 1e-3 of the reference) or 'f32w'; python -m riser_amd.rangecheck "
                "prints every layer's maximum; range_check=False / RS_RANGE_CHECK=0 load the model anyway")

    def saturated(self, reset: bool = True) -> bool:
        """True if a half-precision conversion overflowed (an activation beyond 65504) in any call on this model since the
        flag was last reset; waits for the caller's stream.  Always False for the fp32 / bf16 modes and the generic conv
        programs (fp32's exponent range)."""
        if self._h is None or self.dtype not in self.HALF_MODES:
            return False
        rc = nv.lib().rs_model_saturated(self._h, 1 if reset else 0, _stream_ptr(self.device))
        if rc < 0:
            nv.check(rc, "rs_model_saturated")
        return rc == 1

    def warn_if_saturated(self, what: str = "a call") -> bool:
        """`saturated()` turned into a RuntimeWarning (and a logger warning): what classify() does after every call in a
        half-precision mode, and the control loop once per batch"""
        if not self.saturated(reset=True):
            return False
        msg = (f"riser_amd: {what} on target {self.target!r} overflowed half precision (an activation beyond 65504 in dtype "
               f"{self.dtype!r}): the probabilities of the affected reads are WRONG. Use dtype 'bf16x3' or 'f32w' for these weights.")
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)
        if self.logger is not None:
            self.logger.warning(msg)
        return True

    # ------------------------------------------------------------------------------------
    def _get_device(self, device=None):
        nv.require_gpu()
        if not torch.cuda.is_available():
            raise nv.NativeError("torch sees no ROCm device")
        if device is None:
            return torch.device("cuda", torch.cuda.current_device())
        d = torch.device(device)
        return torch.device("cuda", d.index if d.index is not None else torch.cuda.current_device())

    def _init_resnet(self, state, rc, dtype: str):
        from .resnet import SeqNet, build_program
        sd = state if isinstance(state, dict) else torch.load(state, map_location="cpu")
        sd = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in sd.items()}
        if int(rc.n_classes) != 2:
            raise ValueError("riser_amd supports two-class heads only")
        seq_dtype = {"f32w": "f32", "f32": "f32", "bf16x3": "bf16x3"}.get(dtype)
        if seq_dtype is None:
            raise ValueError(f"dtype {dtype!r}: a ResNet runs in 'f32w' / 'f32' (f32-input MFMA) or 'bf16x3' (stem and basic "
                             "blocks in split precision on the bf16 MFMA)")
        self.classifier, self._fc_positions = "gap_fc", 0
        self.channels = [int(c) for c in rc.channels]
        self.n_layers = len(self.channels)
        self._keep, self._h, self.model = [], None, self
        self._seq = SeqNet(*build_program(sd, rc), device=self.device, dtype=seq_dtype)
        self.dtype = seq_dtype
        self._ws = Workspace(self.device)
        # shortest input the program accepts (torch raises below it: kernel larger than the padded input, empty pooling)
        lo, hi = 1, 1 << 16
        lib = nv.lib()
        while lo < hi:
            mid = (lo + hi) // 2
            if lib.rs_seqnet_workspace_bytes(self._seq._h, 1, mid):
                hi = mid
            else:
                lo = mid + 1
        self.min_length = lo

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            nv.lib().rs_model_destroy(h)
        seq, self._seq = getattr(self, "_seq", None), None
        if seq is not None:
            seq.close()

    def _seq_forward(self, x: torch.Tensor, lens_host: np.ndarray, return_logits: bool, out, lens_dev: torch.Tensor = None):
        """generic program.  A ResNet (stem + residual blocks, all fused) takes the batch as it is - ragged lengths, one call
        (rs_seqnet_forward_ragged: every kernel masks by the read's own rows); any other program runs one uniform length per
        launch, so its reads are grouped by length"""
        B = x.shape[0]
        probs = out if out is not None else torch.empty((B, 2), dtype=torch.float32, device=self.device)
        if self._seq.ragged_ok and x.is_contiguous():
            if lens_dev is None:
                lens_dev = torch.from_numpy(np.ascontiguousarray(lens_host, dtype=np.int32)).to(self.device)
            return self._seq.forward_ragged(x, lens_dev.to(torch.int32), return_logits, out=probs)
        logits = torch.empty((B, 2), dtype=torch.float32, device=self.device) if return_logits else None
        for L in np.unique(lens_host):
            idx = torch.from_numpy(np.flatnonzero(lens_host == L)).to(self.device)
            r = self._seq.forward(x[idx, : int(L)].contiguous(), return_logits)
            probs[idx] = r[0] if return_logits else r
            if return_logits:
                logits[idx] = r[1]
        return (probs, logits) if return_logits else probs

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def padded_length(self, lmax: int) -> int:
        if self._seq is not None:
            return int(lmax)
        return nv.lib().rs_padded_length(self._h, int(lmax))

    def block_samples(self, layer: int | None = None) -> int:
        """Block size of the packed activation layout in samples: of the late layers and the head (rs_block_samples), or,
        with `layer`, of the layout conv layer `layer` runs on (the early layers use finer blocks: rs_layer_info)."""
        self._need_handle("block_samples")
        if layer is None:
            return nv.lib().rs_block_samples(self._h)
        return self.layer_info()[layer]["block_samples"]

    def block_bases(self, lens_host, layer: int | None = None) -> np.ndarray:
        """First block of every read of a batch in the packed layout (+ the total as the last entry): read b's rows
        in conv layer i's output start at row bases[b] * (U >> (i + 1)), U = block_samples(i)."""
        U = self.block_samples(layer)
        nblk = np.asarray(lens_host, dtype=np.int64) // U + 1
        return np.concatenate([[0], np.cumsum(nblk)]).astype(np.int64)

    def _need_handle(self, what: str):
        if self._h is None:
            raise NotImplementedError(f"{what}: not available for generic conv programs (depth > 1 / kernels other than 3)")

    def layer_info(self):
        self._need_handle("layer_info")
        out = []
        for i in range(self.n_layers):
            li = nv.LayerInfo()
            nv.check(nv.lib().rs_model_layer_info(self._h, i, C.byref(li)), "rs_model_layer_info")
            out.append({n: getattr(li, n) for n, _ in nv.LayerInfo._fields_})
        return out

    def profile(self, on, coarse: bool = False):
        """HIP-event stage timing on the launch stream; coarse = only the conv stack's boundaries (4 events per call)."""
        self._need_handle("profile")
        nv.check(nv.lib().rs_profile_enable(self._h, (2 if coarse else 1) if on else 0), "rs_profile_enable")

    def profile_read(self):
        """-> (stage_ms float32 [n_layers + 2], calls); see rs_profile_read."""
        self._need_handle("profile_read")
        ms = np.zeros(self.n_layers + 2, dtype=np.float32)
        calls = C.c_int32(0)
        nv.check(nv.lib().rs_profile_read(self._h, ms.ctypes.data, C.byref(calls)), "rs_profile_read")
        return ms, calls.value

    def max_batch(self, lmax: int) -> int:
        """Largest batch one library call accepts for reads of up to lmax samples: the conv kernels
        address every activation buffer through a 2 GiB buffer-resource window (32-bit offsets with
        hardware bounds checking).  Bigger batches are split transparently by the methods below."""
        if self._seq is not None:
            return self._seq.max_batch(int(lmax))
        return max(1, nv.lib().rs_max_batch(self._h, int(lmax)))

    # samples (reads x padded length) per library call beyond which a batch is cut into equal sub-batches: throughput is flat
    # from ~1000 reads of 16000 samples on and falls off once a layer's activations outgrow the 256 MB Infinity Cache
    # (profiles/r04_batch_sweep_*.txt: fp32 4096 reads -4 %, split precision 2048 reads -5 %; two bytes more per element there)
    _CALL_SAMPLES = {"f32w": 2048 << 14, "f32": 2048 << 14, "bf16": 2048 << 14, "f16": 2048 << 14,
                     "bf16x3": 1024 << 14, "f16x3": 1024 << 14, "f16xf8": 1024 << 14}

    def call_batch(self, B: int, lmax: int) -> int:
        """Reads per library call for a batch of B reads of up to lmax samples: B itself, or - beyond the 2 GiB buffer
        window (max_batch) or the cache-friendly size - the size of equal sub-batches.  Reads are independent and their
        bits do not depend on their batch-mates, so the cut changes nothing but the time."""
        hard = self.max_batch(lmax)
        cap = hard
        if self._seq is None:
            cap = min(hard, max(512, self._CALL_SAMPLES.get(self.dtype, 2048 << 14) // ((int(lmax) // 1024 + 1) * 1024)))
        if B <= cap or B <= min(hard, cap + cap // 4):          # the cache-friendly size is a preference: no sliver calls
            return B
        n = -(-B // cap)
        # equal shares, rounded up to whole tile rounds of the late layers at 16000-sample reads (a 640-read share is two
        # rounds of tiles for 1.25 rounds of work: 1280 reads run as 768 + 512, not 640 + 640)
        return min(cap, -(-(-(-B // n)) // 256) * 256)

    def reserve(self, B: int, lmax: int):
        """Allocate the workspace of the current stream for batches of up to B reads of up to lmax samples now (a
        growing workspace is a device allocation inside some later call)."""
        if self._h is not None:
            B = max(1, self.call_batch(int(B), lmax))
            self._ws.get(nv.lib().rs_workspace_bytes(self._h, B, int(lmax)))

    def _check_lengths(self, lens_host: np.ndarray):
        if lens_host.size == 0:
            raise ValueError("empty batch")
        if self._fc_positions:
            rows = np.asarray(lens_host, dtype=np.int64) >> self.n_layers
            bad = np.flatnonzero(rows != self._fc_positions)
            if bad.size and int(np.asarray(lens_host).min()) >= self.min_length:
                # what torch raises in the reference's first Linear (riser/nets/cnn.py:24,47) for a read of another length
                c = self.channels[-1]
                raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied (1x{c * int(rows[bad[0]])} and "
                                   f"{c * self._fc_positions}x{self._fc_hidden})")
        if int(lens_host.min()) < self.min_length:
            # torch raises "max_pool1d() Invalid computed output size: 0" in the reference
            raise ValueError(f"signal of {int(lens_host.min())} samples is shorter than the network "
                             f"minimum {self.min_length}")

    # ------------------------------------------------------------------------------------
    def classify(self, signal):
        """riser/model.py:22-28: normalised signal [L] -> fp32 Tensor[2] on the device."""
        if self.classifier == "gap":
            # the reference cannot classify ONE read with this head: x.squeeze() (riser/nets/cnn.py:48-49) drops the batch
            # dimension and softmax(dim=1) (riser/model.py:27) raises this; the batched entry points below run it
            raise IndexError("Dimension out of range (expected to be in range of [-1, 0], but got 1)")
        sig = np.ascontiguousarray(signal)
        if sig.ndim != 1:
            raise ValueError("classify takes ONE normalised signal [L]; classify_batch takes several")
        n = int(sig.shape[0])
        lens = np.array([n], dtype=np.int32)
        if self.device.type != "cuda" or n == 0:
            x = torch.from_numpy(sig).unsqueeze(0).to(self.device, dtype=torch.float)
            return self.forward_batch(x, lens)[0]
        # the reference's loop calls this once per read (riser/control.py:68-69): the signal and its length go to the device
        # through pinned staging slots as asynchronous copies on the caller's stream (a pageable .to() is two blocking copies
        # of ~24 us each that also serialise the host with the previous call's kernels)
        slot = self._classify_slot(n)
        slot["free"].synchronize()                                     # the slot's previous copies have left the host buffer
        slot["x"][:n].copy_(torch.from_numpy(sig))                     # any float dtype / the int64 zeros of mad == 0 -> fp32
        slot["len"][0] = n
        x = torch.empty((1, n), dtype=torch.float32, device=self.device)
        ln = torch.empty((1,), dtype=torch.int32, device=self.device)
        x.copy_(slot["x"][:n].unsqueeze(0), non_blocking=True)
        ln.copy_(slot["len"], non_blocking=True)
        slot["free"].record()
        out = self.forward_batch(x, lens, lens_dev=ln)[0]
        if self.dtype in self.HALF_MODES:          # the reference's fp32 call cannot overflow: say so when this one did
            self.warn_if_saturated("classify()")
        return out

    def _classify_slot(self, n: int):
        """pinned staging for classify(): four slots used in turn, each with an event that says its last copies are done"""
        slots = self.__dict__.setdefault("_cls_slots", [])
        k = self.__dict__.get("_cls_next", 0)
        self._cls_next = (k + 1) & 3
        if len(slots) <= k:
            slots.append({"x": torch.empty(max(n, 16384), dtype=torch.float32).pin_memory(),
                          "len": torch.empty(1, dtype=torch.int32).pin_memory(), "free": torch.cuda.Event()})
        slot = slots[k]
        if slot["x"].numel() < n:
            slot["free"].synchronize()
            slot["x"] = torch.empty(n, dtype=torch.float32).pin_memory()
        return slot

    def classify_batch(self, signals, lengths=None, return_logits: bool = False):
        """Batched classify of already-normalised signals.
        signals: list of 1-D arrays (any float dtype / the int64 zeros of mad == 0), or a
        2-D array / tensor [B, Lmax] with `lengths`.  Returns fp32 [B, 2] on the device."""
        if isinstance(signals, (list, tuple)):
            lens = np.array([len(s) for s in signals], dtype=np.int32)
            self._check_lengths(lens)
            host = np.zeros((len(signals), int(lens.max())), dtype=np.float32)
            for i, s in enumerate(signals):
                host[i, : lens[i]] = np.asarray(s, dtype=np.float32)
            x = torch.from_numpy(host).to(self.device)
        else:
            x = torch.as_tensor(signals).to(self.device, dtype=torch.float)
            if x.dim() != 2:
                raise ValueError("signals must be [B, L]")
            lens = (np.full(x.shape[0], x.shape[1], dtype=np.int32) if lengths is None
                    else np.asarray(lengths, dtype=np.int32))
        return self.forward_batch(x, lens, return_logits=return_logits)

    def forward_batch(self, x: torch.Tensor, lens_host: np.ndarray, lens_dev: torch.Tensor = None,
                      return_logits: bool = False, out: torch.Tensor = None):
        """x: fp32 device tensor [B, ldx] (row b valid for lens[b] samples)."""
        self._check_lengths(lens_host)
        B, ldx = x.shape
        if not x.is_contiguous() or x.dtype != torch.float32 or x.device != self.device:
            raise ValueError("x must be a contiguous fp32 tensor on the model's device")
        lmax = int(lens_host.max())
        if lmax > ldx:
            raise ValueError("a length exceeds the row pitch")
        if self._seq is not None:
            return self._seq_forward(x, np.asarray(lens_host), return_logits, out, lens_dev)
        mb = self.call_batch(B, lmax)
        if B > mb:                                                     # split: reads are independent
            probs = out if out is not None else torch.empty((B, 2), dtype=torch.float32, device=self.device)
            logits = torch.empty((B, 2), dtype=torch.float32, device=self.device) if return_logits else None
            for s0 in range(0, B, mb):
                s1 = min(B, s0 + mb)
                r = self.forward_batch(x[s0:s1], lens_host[s0:s1], None if lens_dev is None else lens_dev[s0:s1],
                                       return_logits, out=probs[s0:s1])
                if return_logits:
                    logits[s0:s1] = r[1]
            return (probs, logits) if return_logits else probs
        lens_host = np.ascontiguousarray(lens_host, dtype=np.int32)
        if lens_dev is None:
            lens_dev = torch.from_numpy(lens_host).to(self.device)
        L = nv.lib()
        need = L.rs_workspace_bytes(self._h, B, lmax)
        ws = self._ws.get(need)
        probs = out if out is not None else torch.empty((B, 2), dtype=torch.float32, device=self.device)
        logits = torch.empty((B, 2), dtype=torch.float32, device=self.device) if return_logits else None
        nv.check(L.rs_forward(self._h, x.data_ptr(), ldx, lens_dev.data_ptr(), lens_host.ctypes.data, B,
                              int(lens_host.min()), lmax, ws.data_ptr(),
                              ws.numel(), probs.data_ptr(), logits.data_ptr() if return_logits else None,
                              _stream_ptr(self.device)), "rs_forward")
        return (probs, logits) if return_logits else probs

    def classify_raw(self, sig_dev: torch.Tensor, off_dev: torch.Tensor, len_dev: torch.Tensor,
                     lens_host: np.ndarray, out: torch.Tensor = None, return_logits: bool = False, packed: bool = True):
        """Fused normalise + forward on raw int16 reads already resident on the device:
        read b = sig_dev[off[b] : off[b] + len[b]].  Returns fp32 [B, 2] on the device.
        packed=False withholds the host's copy of the lengths from the library (every read then takes the blocks of the
        longest one: the layout a caller without host lengths gets; bit-identical results)."""
        self._check_lengths(lens_host)
        B = int(lens_host.shape[0])
        lmax = int(lens_host.max())
        if self._seq is not None:                                      # generic program: normalise, then group by length
            xn = torch.empty((B, lmax), dtype=torch.float32, device=self.device)
            nv.check(nv.lib().rs_normalise(sig_dev.data_ptr(), off_dev.data_ptr(), len_dev.data_ptr(), B, lmax,
                                           xn.data_ptr(), lmax, lmax, None, 0, None, _stream_ptr(self.device)),
                     "rs_normalise")
            return self._seq_forward(xn, np.asarray(lens_host), return_logits, out, len_dev)
        mb = self.call_batch(B, lmax)
        if B > mb:                                                     # split: reads are independent
            probs = out if out is not None else torch.empty((B, 2), dtype=torch.float32, device=self.device)
            logits = torch.empty((B, 2), dtype=torch.float32, device=self.device) if return_logits else None
            for s0 in range(0, B, mb):
                s1 = min(B, s0 + mb)
                r = self.classify_raw(sig_dev, off_dev[s0:s1], len_dev[s0:s1], lens_host[s0:s1], out=probs[s0:s1],
                                      return_logits=return_logits, packed=packed)
                if return_logits:
                    logits[s0:s1] = r[1]
            return (probs, logits) if return_logits else probs
        L = nv.lib()
        lens_host = np.ascontiguousarray(lens_host, dtype=np.int32)
        need = L.rs_workspace_bytes(self._h, B, lmax)
        ws = self._ws.get(need)
        probs = out if out is not None else torch.empty((B, 2), dtype=torch.float32, device=self.device)
        logits = torch.empty((B, 2), dtype=torch.float32, device=self.device) if return_logits else None
        nv.check(L.rs_classify(self._h, sig_dev.data_ptr(), off_dev.data_ptr(), len_dev.data_ptr(),
                               lens_host.ctypes.data if packed else None, B, int(lens_host.min()), lmax,
                               ws.data_ptr(), ws.numel(), probs.data_ptr(),
                               logits.data_ptr() if return_logits else None, _stream_ptr(self.device)),
                 "rs_classify")
        return (probs, logits) if return_logits else probs


def _autotune(self, sig_dev: torch.Tensor, off_dev: torch.Tensor, len_dev: torch.Tensor, lens_host: np.ndarray) -> int:
    """Time every feasible tile shape of each tiled conv layer on this batch (rs_autotune) and keep the measured best
    for later batches of the same geometry; returns the number of layers whose choice changed.  Optional: call once per
    deployment batch size (e.g. before a run); results are bit-identical with or without it."""
    self._need_handle("autotune")
    self._check_lengths(lens_host)
    B, lmax = int(lens_host.shape[0]), int(lens_host.max())
    if B > self.max_batch(lmax):
        raise ValueError("autotune: batch exceeds one library call (Model.max_batch)")
    lens_host = np.ascontiguousarray(lens_host, dtype=np.int32)
    L = nv.lib()
    ws = self._ws.get(L.rs_workspace_bytes(self._h, B, lmax))
    probs = torch.empty((B, 2), dtype=torch.float32, device=self.device)
    changed = C.c_int32(0)
    nv.check(L.rs_autotune(self._h, sig_dev.data_ptr(), off_dev.data_ptr(), len_dev.data_ptr(), lens_host.ctypes.data, B,
                           int(lens_host.min()), lmax, ws.data_ptr(), ws.numel(), probs.data_ptr(), C.byref(changed), _stream_ptr(self.device)),
             "rs_autotune")
    return changed.value


Model.autotune = _autotune


_CONCURRENT_BELOW_SAMPLES = 1800 * 4096          # csrc/api.hip: kConcurrentBelowSamples


def _ensemble_bytes(models, B: int, lmax: int) -> int:
    """workspace of one rs_classify_ensemble call: a pair of activation buffers per model, so that the forwards of the
    models run concurrently (include/riser_amd.h)"""
    hs = (C.c_void_p * len(models))(*[m._h for m in models])
    L = nv.lib()
    serial = max(L.rs_workspace_bytes(m._h, int(B), int(lmax)) for m in models)
    # the library forks the forwards only below ~1800 blocks of 4096 samples (csrc/api.hip, rs_classify_ensemble: beyond that
    # one model's launches fill the chip): a call that can never take the fork needs - and pins - the serial workspace only
    U = int(L.rs_block_samples(models[0]._h))
    if len(models) < 2 or int(B) * (int(lmax) // U + 1) * U >= _CONCURRENT_BELOW_SAMPLES:
        return serial
    need = L.rs_ensemble_workspace_bytes(hs, len(models), int(B), int(lmax))
    return need if need else serial       # 0: models that cannot run concurrently (an `fc` classifier)


def reserve_ensemble(models, B: int, lmax: int):
    """The workspace classify_raw_ensemble will ask for (it lives with models[0]), allocated ahead of the run."""
    m0 = models[0]
    if any(m._h is None for m in models):
        return
    B = max(1, min(m.call_batch(int(B), lmax) for m in models))
    m0._ws.get(_ensemble_bytes(models, B, lmax))


def classify_raw_ensemble(models, sig_dev: torch.Tensor, off_dev: torch.Tensor, len_dev: torch.Tensor,
                          lens_host: np.ndarray, out: torch.Tensor = None, decision: torch.Tensor = None,
                          max_len: int = 0, threshold: float = 0.9, mode: int = nv.RS_ENRICH):
    """The model loop of riser/control.py:63-82 for a whole batch in ONE library call: raw int16 reads on
    the device are normalised once, every model (same architecture) classifies them, and - when a uint8
    `decision` tensor [B] is given - the accept / reject / try-again decision is taken on the device.
    Returns fp32 probabilities [n_models, B, 2] on the device."""
    m0 = models[0]
    m0._check_lengths(lens_host)
    B, lmax = int(lens_host.shape[0]), int(lens_host.max())
    lens_host = np.ascontiguousarray(lens_host, dtype=np.int32)
    if any(m._h is None for m in models):
        # a generic conv program (depth > 1 / kernels other than 3) among the models: one classify_raw per model and
        # the decision kernel on their probabilities - the same arithmetic, just not one library call
        probs = out if out is not None else torch.empty((len(models), B, 2), dtype=torch.float32, device=m0.device)
        for k, m in enumerate(models):
            m.classify_raw(sig_dev, off_dev, len_dev, lens_host, out=probs[k])
        if decision is not None:
            nv.check(nv.lib().rs_decide(probs.data_ptr(), len(models), B, len_dev.data_ptr(), int(max_len), float(threshold),
                                        int(mode), decision.data_ptr(), _stream_ptr(m0.device)), "rs_decide")
        return probs
    mb = min(m.call_batch(B, lmax) for m in models)
    if B > mb:                                                         # split: reads are independent
        probs = out if out is not None else torch.empty((len(models), B, 2), dtype=torch.float32, device=m0.device)
        for s0 in range(0, B, mb):
            s1 = min(B, s0 + mb)
            part = classify_raw_ensemble(models, sig_dev, off_dev[s0:s1].contiguous(), len_dev[s0:s1].contiguous(),
                                         lens_host[s0:s1], decision=None if decision is None else decision[s0:s1],
                                         max_len=max_len, threshold=threshold, mode=mode)
            probs[:, s0:s1] = part
        return probs
    L = nv.lib()
    ws = m0._ws.get(_ensemble_bytes(models, B, lmax))
    probs = out if out is not None else torch.empty((len(models), B, 2), dtype=torch.float32, device=m0.device)
    hs = (C.c_void_p * len(models))(*[m._h for m in models])
    nv.check(L.rs_classify_ensemble(hs, len(models), sig_dev.data_ptr(), off_dev.data_ptr(), len_dev.data_ptr(),
                                    lens_host.ctypes.data, B, int(lens_host.min()), lmax, ws.data_ptr(), ws.numel(), probs.data_ptr(),
                                    decision.data_ptr() if decision is not None else None, int(max_len),
                                    float(threshold), int(mode), _stream_ptr(m0.device)), "rs_classify_ensemble")
    return probs
