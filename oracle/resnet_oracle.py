"""numpy restatement of the reference ResNet in eval mode (riser/nets/resnet.py:7-131).
TEST INFRASTRUCTURE ONLY; pinned by tests/golden/resnet.npz (outputs of the reference ResNet)."""
from __future__ import annotations

import numpy as np

EPS = 1e-5


def conv1d(x, w, b=None, stride=1, pad=0):
    """x [B, C, T], w [Co, C, K] -> [B, Co, T_out] (torch.nn.Conv1d semantics)."""
    B, C, T = x.shape
    Co, _, K = w.shape
    xp = np.zeros((B, C, T + 2 * pad), dtype=x.dtype)
    xp[:, :, pad:pad + T] = x
    To = (T + 2 * pad - K) // stride + 1
    y = np.zeros((B, Co, To), dtype=x.dtype)
    for k in range(K):
        y += np.matmul(w[:, :, k], xp[:, :, k:k + stride * (To - 1) + 1:stride])
    if b is not None:
        y += b[None, :, None]
    return y


def bn(x, sd, p):
    s = sd[p + ".weight"] / np.sqrt(sd[p + ".running_var"] + EPS)
    return (x - sd[p + ".running_mean"][None, :, None]) * s[None, :, None] + sd[p + ".bias"][None, :, None]


def maxpool_2_2_p1(x):
    B, C, T = x.shape
    xp = np.full((B, C, T + 2), -np.inf, dtype=x.dtype)
    xp[:, :, 1:T + 1] = x
    To = T // 2 + 1
    return np.maximum(xp[:, :, 0:2 * To:2], xp[:, :, 1:2 * To:2])


def resnet_forward(sd, cfg, x, acc=np.float64):
    """x [B, L] -> logits [B, 2].  cfg: channels, kernel, padding, stride, block, n_layers, blocks."""
    sd = {k: np.asarray(v).astype(acc) for k, v in sd.items() if "num_batches" not in k}
    h = np.asarray(x, dtype=acc)[:, None, :]
    h = conv1d(h, sd["conv_block.0.weight"], sd["conv_block.0.bias"], cfg["stride"], cfg["padding"])
    h = np.maximum(bn(h, sd, "conv_block.1"), 0)
    h = maxpool_2_2_p1(h)
    in_ch = cfg["channels"][0]
    for i in range(cfg["n_layers"]):
        out_ch = cfg["channels"][i]
        for j in range(cfg["blocks"][i]):
            stride = 2 if (i > 0 and j == 0) else 1
            p = f"layers.{i}.{j}"
            res = h
            if in_ch != out_ch or stride != 1:
                res = bn(conv1d(h, sd[p + ".shortcut.0.weight"], None, stride, 0), sd, p + ".shortcut.1")
            if cfg["block"] == "bottleneck":
                o = np.maximum(bn(conv1d(h, sd[p + ".blocks.0.0.weight"]), sd, p + ".blocks.0.1"), 0)
                o = np.maximum(bn(conv1d(o, sd[p + ".blocks.1.0.weight"], None, stride, 1), sd, p + ".blocks.1.1"), 0)
                o = bn(conv1d(o, sd[p + ".blocks.2.0.weight"]), sd, p + ".blocks.2.1")
            else:
                o = np.maximum(bn(conv1d(h, sd[p + ".blocks.0.0.weight"], None, stride, 1), sd, p + ".blocks.0.1"), 0)
                o = bn(conv1d(o, sd[p + ".blocks.1.0.weight"], None, 1, 1), sd, p + ".blocks.1.1")
            h = np.maximum(o + res, 0)
            in_ch = out_ch
    feat = h.mean(axis=2)
    return feat @ sd["decoder.2.weight"].T + sd["decoder.2.bias"]
