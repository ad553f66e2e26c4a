"""torch-CPU restatement of Model.classify, one read at a time.  TEST INFRASTRUCTURE ONLY.

The reference delegates its arithmetic to torch's stock CPU ops (conv1d / relu /
max_pool1d / adaptive_avg_pool1d / linear / softmax: riser/nets/cnn.py:30-32,55-64,
riser/model.py:24-27) and calls them with batch = 1 per read (riser/control.py:63-69).
This file calls the same ops through torch.nn.functional with weights taken from a
reference-format state dict, so that bench.py can time "the reference's CPU path" on the
GPU box's host cores (the reference's Python cannot travel there) and tests can compare
against the numpy oracle.  Never imported by the product package.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import riser_oracle as ro


class TorchCpuModel:
    def __init__(self, sd: dict):
        self.w = []
        i = 0
        while f"layers.{i}.0.weight" in sd:
            self.w.append((torch.as_tensor(np.asarray(sd[f"layers.{i}.0.weight"])).float(),
                           torch.as_tensor(np.asarray(sd[f"layers.{i}.0.bias"])).float()))
            i += 1
        self.fc_w = torch.as_tensor(np.asarray(sd["classifier.2.weight"])).float()
        self.fc_b = torch.as_tensor(np.asarray(sd["classifier.2.bias"])).float()

    @torch.no_grad()
    def logits(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, L] fp32 -> logits [B, 2] (riser/nets/cnn.py:43-49)."""
        h = x.unsqueeze(1)
        for w, b in self.w:
            h = F.max_pool1d(F.relu(F.conv1d(h, w, b, stride=1, padding=1)), 2, 2)
        h = F.adaptive_avg_pool1d(h, 1).flatten(1)
        return F.linear(h, self.fc_w, self.fc_b)

    @torch.no_grad()
    def classify(self, signal: np.ndarray) -> torch.Tensor:
        """riser/model.py:22-28 on CPU: normalised signal [L] -> probs[2]."""
        x = torch.from_numpy(np.ascontiguousarray(signal)).unsqueeze(0).to(dtype=torch.float)
        return F.softmax(self.logits(x), dim=1)[0]


def classify_per_read(model: TorchCpuModel, raw_signals) -> np.ndarray:
    """The structure of riser/control.py:63-69: for each read normalise, then classify
    at batch 1.  raw_signals: iterable of int16 arrays -> [B, 2] float32."""
    out = []
    for s in raw_signals:
        out.append(model.classify(ro.mad_normalise(s)).numpy())
    return np.stack(out)


def classify_batched(model: TorchCpuModel, raw_signals, lengths=None, batch: int = 64) -> np.ndarray:
    """The same arithmetic for a whole batch in seconds: every read normalised by the numpy restatement, then the
    torch-CPU conv stack over groups of equal length (padding to a common length would change the 'same' padding, the
    MaxPool floor and the GAP divisor: riser/nets/cnn.py:30,55-64).  Read i is raw_signals[i][:lengths[i]].
    -> [B, 2] float32, row order of the input."""
    n = len(raw_signals)
    lengths = [len(s) for s in raw_signals] if lengths is None else [int(v) for v in lengths]
    out = np.empty((n, 2), dtype=np.float32)
    by_len = {}
    for i, ln in enumerate(lengths):
        by_len.setdefault(ln, []).append(i)
    for ln, idx in by_len.items():
        for k in range(0, len(idx), batch):
            part = idx[k:k + batch]
            xs = np.stack([ro.mad_normalise(np.asarray(raw_signals[i])[:ln]) for i in part]).astype(np.float32)
            out[part] = F.softmax(model.logits(torch.from_numpy(xs)), dim=1).numpy()
    return out
