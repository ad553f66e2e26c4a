"""ResNetModel: the reference's secondary architecture (riser/nets/resnet.py:7-131) on the GPU.

The reference's `Model` cannot load a ResNet (it hard-imports ConvNet, riser/model.py:3,18) and
ships neither a `config.resnet` section nor weights, so this class mirrors `Model`'s surface
(`classify(signal) -> Tensor[2]`) for users who train one with riser/train.py:177-178.  Eval-mode
BatchNorm is folded into the preceding conv here on the host; the device runs the resulting
conv / max-pool program (csrc/seqnet.hip) - parity-grade kernels, not the tuned ConvNet path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _native as nv

_BN_EPS = 1e-5


def _fold(w, bn, prefix, conv_bias=None):
    """conv weight [co, ci, k] + BatchNorm1d(eval) -> (w', b') with y = conv(x; w') + b'."""
    g, b = bn[prefix + ".weight"], bn[prefix + ".bias"]
    mu, var = bn[prefix + ".running_mean"], bn[prefix + ".running_var"]
    scale = (g.astype(np.float64) / np.sqrt(var.astype(np.float64) + _BN_EPS))
    wf = (w.astype(np.float64) * scale[:, None, None]).astype(np.float32)
    shift = b.astype(np.float64) - mu.astype(np.float64) * scale
    if conv_bias is not None:
        shift = shift + conv_bias.astype(np.float64) * scale
    return np.ascontiguousarray(wf), np.ascontiguousarray(shift.astype(np.float32))


def build_program(sd, c):
    """(ops, n_buffers, fc_w, fc_b, c_last): the conv / max-pool program of a reference ResNet
    state dict.  ops = list of dicts {kind, src, dst, add, w, b, stride, pad, relu}; buffer 0 is the
    input.  Pure numpy (unit-testable without a GPU)."""
    ops = []

    def conv(src, dst, w, b, stride, pad, relu, add=-1):
        ops.append(dict(kind=0, src=src, dst=dst, add=add, w=np.ascontiguousarray(w, dtype=np.float32),
                        b=np.ascontiguousarray(b, dtype=np.float32), stride=int(stride), pad=int(pad), relu=int(relu)))

    # stem (resnet.py:79-84): conv(bias) + BN + ReLU, MaxPool1d(2, 2, padding 1)
    w, b = _fold(sd["conv_block.0.weight"], sd, "conv_block.1", sd["conv_block.0.bias"])
    conv(0, 1, w, b, int(c.stride), int(c.padding), True)
    ops.append(dict(kind=1, src=1, dst=2, add=-1, pad=1))
    cur, in_ch = 2, int(c.channels[0])
    bottleneck = c.block == "bottleneck"
    n_buffers = 7
    for i in range(int(c.n_layers)):
        out_ch = int(c.channels[i])
        for j in range(int(c.blocks[i])):
            stride = 2 if (i > 0 and j == 0) else 1
            pre = f"layers.{i}.{j}"
            free = [k for k in range(1, n_buffers) if k != cur]
            res = cur
            if in_ch != out_ch or stride != 1:                    # should_apply_shortcut (resnet.py:45-47)
                w, b = _fold(sd[pre + ".shortcut.0.weight"], sd, pre + ".shortcut.1")
                res = free.pop()
                conv(cur, res, w, b, stride, 0, False)
            if bottleneck:                                        # resnet.py:60-70
                w, b = _fold(sd[pre + ".blocks.0.0.weight"], sd, pre + ".blocks.0.1")
                t1 = free.pop(); conv(cur, t1, w, b, 1, 0, True)
                w, b = _fold(sd[pre + ".blocks.1.0.weight"], sd, pre + ".blocks.1.1")
                t2 = free.pop(); conv(t1, t2, w, b, stride, 1, True)
                w, b = _fold(sd[pre + ".blocks.2.0.weight"], sd, pre + ".blocks.2.1")
                t3 = free.pop(); conv(t2, t3, w, b, 1, 0, True, add=res)      # relu(blocks + residual), :42
                cur = t3
            else:                                                 # resnet.py:50-57
                w, b = _fold(sd[pre + ".blocks.0.0.weight"], sd, pre + ".blocks.0.1")
                t1 = free.pop(); conv(cur, t1, w, b, stride, 1, True)
                w, b = _fold(sd[pre + ".blocks.1.0.weight"], sd, pre + ".blocks.1.1")
                t2 = free.pop(); conv(t1, t2, w, b, 1, 1, True, add=res)
                cur = t2
            in_ch = out_ch
    fw = np.ascontiguousarray(sd["decoder.2.weight"], dtype=np.float32)
    fb = np.ascontiguousarray(sd["decoder.2.bias"], dtype=np.float32)
    return ops, n_buffers, fw, fb, in_ch


def program_flops(prog, L: int) -> float:
    """conv FLOPs of one chunk of L samples through a program of build_program / build_convnet_program:
    2 * c_in * c_out * k per output position of every conv."""
    T, fl = {0: L}, 0.0
    for o in prog:
        if o["kind"] == 0:
            co, ci, k = o["w"].shape
            t_out = (T[o["src"]] + 2 * o["pad"] - k) // o["stride"] + 1
            fl += 2.0 * ci * co * k * t_out
        else:
            t_out = T[o["src"]] // 2 + (1 if o.get("pad", 1) else 0)
        T[o["dst"]] = t_out
    return fl


def program_traffic_bytes(prog, L: int, fused: bool = True) -> float:
    """fp32 bytes one chunk of L samples moves through HBM under the launch structure rs_seqnet_create builds: every
    launch reads its input(s) once and writes its output once.  fused=True: the stem conv + max-pool and every residual
    block (optional 1x1 shortcut conv; conv3, conv3 + residual - or conv1, conv3, conv1 + residual) are one launch each
    (csrc/seqnet.hip: fuse_program); fused=False: one launch per op."""
    T, C = {0: L}, {0: 1}
    shapes = []
    for o in prog:
        if o["kind"] == 0:
            co, ci, k = o["w"].shape
            t_out = (T[o["src"]] + 2 * o["pad"] - k) // o["stride"] + 1
            c_out = co
        else:
            t_out = T[o["src"]] // 2 + (1 if o.get("pad", 1) else 0)
            c_out = C[o["src"]]
        shapes.append((T[o["src"]] * C[o["src"]], t_out * c_out))
        T[o["dst"]], C[o["dst"]] = t_out, c_out
    total, k = 0.0, 0
    while k < len(prog):
        o = prog[k]
        if fused and o["kind"] == 0 and o["w"].shape[1] == 1 and k + 1 < len(prog) and prog[k + 1]["kind"] == 1:
            total += shapes[k][0] + shapes[k + 1][1]                       # stem: signal in, pooled rows out
            k += 2
            continue
        if fused and o["kind"] == 0:
            sc = o["w"].shape[2] == 1 and not o["relu"]
            k1 = k + 1 if sc else k
            if (k1 + 1 < len(prog) and prog[k1]["kind"] == 0 and prog[k1 + 1]["kind"] == 0 and prog[k1]["w"].shape[2] == 3
                    and prog[k1 + 1]["w"].shape[2] == 3 and prog[k1 + 1]["add"] == (o["dst"] if sc else prog[k1]["src"])):
                total += shapes[k1][0] + shapes[k1 + 1][1]                   # basic block: x in, y out
                k = k1 + 2
                continue
            if (k1 + 2 < len(prog) and all(prog[k1 + d]["kind"] == 0 for d in range(3)) and prog[k1]["w"].shape[2] == 1
                    and prog[k1 + 1]["w"].shape[2] == 3 and prog[k1 + 2]["w"].shape[2] == 1
                    and prog[k1 + 2]["add"] == (o["dst"] if sc else prog[k1]["src"])):
                total += shapes[k1][0] + shapes[k1 + 2][1]                   # bottleneck block: x in, y out
                k = k1 + 3
                continue
        total += shapes[k][0] + shapes[k][1] + (shapes[k][1] if o.get("add", -1) >= 0 else 0)
        k += 1
    return 4.0 * total


class SeqNet:
    """A conv / max-pool program on the device (rs_seqnet_*): uniform-length batches [B, L] -> probabilities."""

    def __init__(self, prog, n_buffers, fw, fb, c_last, device, dtype: str = "f32"):
        """dtype: "f32" (every conv on the f32-input MFMA) or "bf16x3" (the stem and the residual blocks in split precision on the
        bf16 MFMA, rs_seqnet_set_mode: within 1e-3 of the reference, ~1.6 x the fp32 rate)."""
        nv.require_gpu()
        d = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device("cuda", d.index if d.index is not None else torch.cuda.current_device())
        self._keep = (prog, fw, fb)
        ops = []
        for o in prog:
            if o["kind"] == 0:
                w, b = o["w"], o["b"]
                ops.append(nv.SeqOp(0, o["src"], o["dst"], o["add"], w.shape[1], w.shape[0], w.shape[2], o["stride"],
                                    o["pad"], o["relu"], w.ctypes.data, b.ctypes.data))
            else:
                ops.append(nv.SeqOp(1, o["src"], o["dst"], -1, 0, 0, 0, 0, int(o.get("pad", 1)), 0, None, None))
        arr = (nv.SeqOp * len(ops))(*ops)
        h = C.c_void_p()
        nv.check(nv.lib().rs_seqnet_create(arr, len(ops), n_buffers, fw.ctypes.data, fb.ctypes.data, c_last,
                                           self.device.index, C.byref(h)), "rs_seqnet_create")
        self._h = h
        self._ws = None
        self.dtype = {"f32": "f32", "f32w": "f32", "fp32": "f32", "bf16x3": "bf16x3"}.get(dtype)
        if self.dtype is None:
            self.close()
            raise ValueError(f"dtype {dtype!r}: generic conv programs run in 'f32' or 'bf16x3'")
        if self.dtype == "bf16x3":
            try:
                nv.check(nv.lib().rs_seqnet_set_mode(self._h, nv.RS_BF16X3), "rs_seqnet_set_mode")
            except Exception:
                self.close()
                raise

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            nv.lib().rs_seqnet_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, x: torch.Tensor, return_logits: bool = False):
        """x: fp32 device tensor [B, L] (one common length) -> fp32 [B, 2] on the device."""
        B, L = x.shape
        lib = nv.lib()
        need = lib.rs_seqnet_workspace_bytes(self._h, B, L)
        if need == 0:
            raise ValueError(f"signal of {L} samples is too short for this network")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        probs = torch.empty((B, 2), dtype=torch.float32, device=self.device)
        logits = torch.empty((B, 2), dtype=torch.float32, device=self.device) if return_logits else None
        nv.check(lib.rs_seqnet_forward(self._h, x.data_ptr(), B, L, self._ws.data_ptr(), self._ws.numel(),
                                       probs.data_ptr(), logits.data_ptr() if return_logits else None,
                                       torch.cuda.current_stream(self.device).cuda_stream), "rs_seqnet_forward")
        return (probs, logits) if return_logits else probs

    def max_batch(self, L: int) -> int:
        """largest batch of reads of (pitch) L samples one call can address: every activation buffer stays inside the kernels'
        2 GiB buffer window (rs_seqnet_max_batch); bigger batches are split by the callers below and in riser_amd.Model"""
        return max(1, int(nv.lib().rs_seqnet_max_batch(self._h, int(L))))

    @property
    def ragged_ok(self) -> bool:
        """True when forward_ragged can run this program (all of its ops inside fused launches: a ResNet's stem and blocks)"""
        return bool(nv.lib().rs_seqnet_ragged_ok(self._h))

    def forward_ragged(self, x: torch.Tensor, lens_dev: torch.Tensor, return_logits: bool = False, out: torch.Tensor = None):
        """x: fp32 device tensor [B, ld], read b = x[b, :lens_dev[b]] (int32 on the device) -> fp32 [B, 2] on the device; every
        read's result is that of forward() on it alone."""
        B, ld = x.shape
        lib = nv.lib()
        mb = self.max_batch(ld)
        if B > mb:                                  # reads are independent: equal parts, each inside the buffer window
            probs = out if out is not None else torch.empty((B, 2), dtype=torch.float32, device=self.device)
            logits = torch.empty((B, 2), dtype=torch.float32, device=self.device) if return_logits else None
            parts = -(-B // mb)
            step = -(-B // parts)
            for s0 in range(0, B, step):
                s1 = min(B, s0 + step)
                r = self.forward_ragged(x[s0:s1], lens_dev[s0:s1], return_logits, out=probs[s0:s1])
                if return_logits:
                    logits[s0:s1] = r[1]
            return (probs, logits) if return_logits else probs
        need = lib.rs_seqnet_workspace_bytes(self._h, B, ld)
        if need == 0:
            raise ValueError(f"a row pitch of {ld} samples is too short for this network")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        probs = out if out is not None else torch.empty((B, 2), dtype=torch.float32, device=self.device)
        logits = torch.empty((B, 2), dtype=torch.float32, device=self.device) if return_logits else None
        nv.check(lib.rs_seqnet_forward_ragged(self._h, x.data_ptr(), lens_dev.data_ptr(), B, ld, self._ws.data_ptr(),
                                              self._ws.numel(), probs.data_ptr(), logits.data_ptr() if return_logits else None,
                                              torch.cuda.current_stream(self.device).cuda_stream), "rs_seqnet_forward_ragged")
        return (probs, logits) if return_logits else probs


def build_convnet_program(sd, cnn):
    """(ops, n_buffers, fc_w, fc_b, c_last) of a reference ConvNet (riser/nets/cnn.py:8-65) of ANY depth and odd kernel
    sizes with the `gap_fc` classifier: per layer `depth` x [Conv1d(k, stride 1, 'same') + ReLU], then MaxPool1d(2, 2).
    State-dict keys: layers.{i}.{2 d}.weight / .bias (nn.Sequential indices of cnn.py:52-65), classifier.2.*"""
    ops = []
    n_buffers = 3
    cur = 0
    depth = int(getattr(cnn, "depth", 1))
    for i in range(int(cnn.n_layers)):
        k = int(cnn.kernels[i])
        if k % 2 == 0:
            raise ValueError("riser_amd: even conv kernels ('same' pads them asymmetrically) are not supported")
        for d in range(depth):
            w = np.ascontiguousarray(sd[f"layers.{i}.{2 * d}.weight"], dtype=np.float32)
            b = np.ascontiguousarray(sd[f"layers.{i}.{2 * d}.bias"], dtype=np.float32)
            dst = 1 if cur != 1 else 2
            ops.append(dict(kind=0, src=cur, dst=dst, add=-1, w=w, b=b, stride=1, pad=(k - 1) // 2, relu=1))
            cur = dst
        dst = 1 if cur != 1 else 2
        ops.append(dict(kind=1, src=cur, dst=dst, add=-1, pad=0))
        cur = dst
    fw = np.ascontiguousarray(sd["classifier.2.weight"], dtype=np.float32)
    fb = np.ascontiguousarray(sd["classifier.2.bias"], dtype=np.float32)
    return ops, n_buffers, fw, fb, int(cnn.channels[int(cnn.n_layers) - 1])


class ResNetModel:
    def __init__(self, state, config, logger, target, device=None, dtype: str = "f32"):
        """dtype "bf16x3": the stem and the residual blocks in split precision on the bf16 MFMA (SeqNet)."""
        self.target, self.logger = target, logger
        c = config.resnet
        sd = state if isinstance(state, dict) else torch.load(state, map_location="cpu")
        sd = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in sd.items()}
        if int(c.n_classes) != 2:
            raise ValueError("riser_amd supports two-class heads only")
        self._net = SeqNet(*build_program(sd, c), device=device, dtype=dtype)
        self.device = self._net.device
        self.dtype = self._net.dtype

    def close(self):
        self._net.close()

    def classify(self, signal):
        return self.classify_batch(np.asarray(signal)[None, :])[0]

    def classify_batch(self, signals, return_logits: bool = False):
        """signals: [B, L] normalised (one common length).  Returns fp32 [B, 2] on the device."""
        x = torch.as_tensor(np.ascontiguousarray(signals)).to(self.device, dtype=torch.float).contiguous()
        return self._net.forward(x, return_logits)
