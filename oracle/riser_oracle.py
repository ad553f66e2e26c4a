"""CPU oracle for the RISER squiggle-classification hot path.  TEST INFRASTRUCTURE ONLY.

This module is a from-scratch numpy restatement of what the reference computes on the
path  SignalProcessor.mad_normalise -> Model.classify -> ConvNet.forward -> softmax
(plus the polyA end detector and the accept/reject decision either side of it).  It is
the checker the HIP kernels are compared against; it is never imported by the product
package `riser_amd` (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may use it).

Parity status: the reference has no tests, no golden vectors and no shipped weights
(SURVEY.md section 0 items 2 and 5), so nothing *in the reference* pins results.  This
oracle is instead pinned against outputs of the reference itself, produced in the build
container by importing /root/reference (tools/make_golden.py) and committed as fixtures
under tests/golden/; tests/test_oracle_golden.py replays them.  The arithmetic itself
lives in numpy / torch (un-pinned in the reference's requirements.txt:5,12); fixtures
were produced with numpy 2.2.6 and torch 2.10.0 CPU.

Each function cites the reference lines it restates.
"""
from __future__ import annotations

import numpy as np

# riser/preprocess.py:6-12
OUTLIER_LIMIT = 3.5
SCALING_FACTOR = 1.4826
MIN_INPUT_SIGNALS = 4096
MAX_INPUT_NT = 280
TRIM_RESOLUTION = 500
TRIM_MAD_THRESHOLD = 20
TRIM_FIXED_LENGTH_NT = 150.6

# riser/preprocess.py:21-27  (sampling_hz, transloc_rate)
KITS = {"RNA002": (3012, 70), "RNA004": (4000, 130)}


def kit_max_length(kit: str) -> int:
    """riser/preprocess.py:36-37."""
    hz, rate = KITS[kit]
    return int(MAX_INPUT_NT / rate * hz)


def kit_fixed_trim_length(kit: str) -> int:
    """riser/preprocess.py:81-82."""
    hz, rate = KITS[kit]
    return int(TRIM_FIXED_LENGTH_NT / rate * hz)


# --------------------------------------------------------------------------------------
# normalisation
# --------------------------------------------------------------------------------------
def median_mad(signal: np.ndarray) -> tuple[float, float]:
    """median and median-absolute-deviation, float64 (riser/preprocess.py:111,117-120).

    np.median on an even-length array is the float64 mean of the two middle order
    statistics; the oracle computes it from a sort so that it does not depend on
    np.median's partition internals.
    """
    def _med(v):
        s = np.sort(np.asarray(v), kind="stable")
        n = s.shape[0]
        if n % 2:
            return np.float64(s[n // 2])
        return (np.float64(s[n // 2 - 1]) + np.float64(s[n // 2])) / np.float64(2.0)

    med = _med(signal)
    dev = np.abs(np.asarray(signal).astype(np.float64) - med)
    return float(med), float(_med(dev))


def mad_normalise(signal: np.ndarray) -> np.ndarray:
    """riser/preprocess.py:108-147 restated for integer / float input.

    y = (x - med) / (1.4826 * mad) in float64, then the outlier smoothing.  Outliers
    are the positions with |y| > 3.5 *before any update* (:129); they are rewritten in
    ascending order in place (:130-138), so a run of consecutive outliers is a
    left-to-right recurrence in which the left neighbour is the already smoothed value
    and the right neighbour is the original one.  Index 0 copies y[1] and index L-1
    copies y[L-2] without clipping; interior values are clipped to +-3.5 (:141-147).
    mad == 0 gives the all-zero int64 array np.vectorize produces (:122-125).
    """
    signal = np.asarray(signal)
    n = signal.shape[0]
    if n == 0:
        raise ValueError("Signal must not be empty")                    # :109-110
    if np.issubdtype(signal.dtype, np.floating):
        return mad_normalise_float(signal)
    med, mad = median_mad(signal)
    if mad == 0:
        return np.zeros(n, dtype=np.int64)
    y = (signal.astype(np.float64) - np.float64(med)) / (np.float64(SCALING_FACTOR) * np.float64(mad))
    is_out = np.abs(y) > OUTLIER_LIMIT
    idx = np.flatnonzero(is_out)
    if idx.size == 0:
        return y
    # split the ascending outlier indices into runs of consecutive positions
    breaks = np.flatnonzero(np.diff(idx) != 1) + 1
    for run in np.split(idx, breaks):
        for i in run:                                                    # sequential
            if i == 0:
                y[0] = y[1]                                              # IndexError if n == 1, as in the reference
            elif i == n - 1:
                y[i] = y[i - 1]
            else:
                v = (y[i - 1] + y[i + 1]) / 2
                y[i] = min(max(v, -OUTLIER_LIMIT), OUTLIER_LIMIT)
    return y


def mad_normalise_float(signal: np.ndarray) -> np.ndarray:
    """riser/preprocess.py:108-147 for float16 / float32 / float64 input (the retrain path's pA-scaled signals,
    riser/retrain/preprocess.py:79).  numpy >= 2 (NEP 50) keeps the array's precision: np.median and np.abs(x - med)
    return the input dtype T, the Python float 1.4826 adopts T in `1.4826 * mad`, so y = (x - med) / (T(1.4826) * mad)
    and the smoothing recurrence run entirely in T.  mad == 0 gives the int64 zero array of the integer case."""
    x = np.asarray(signal)
    T = x.dtype.type
    n = x.shape[0]
    med = np.median(x)
    mad = np.median(np.abs(x - med))
    if mad == 0:
        return np.zeros(n, dtype=np.int64)
    y = ((x - med) / (T(SCALING_FACTOR) * mad)).astype(x.dtype)
    lim = T(OUTLIER_LIMIT)
    for i in np.flatnonzero(np.abs(y) > lim):                            # ascending; the set is fixed up front (:129)
        if i == 0:
            y[0] = y[1]
        elif i == n - 1:
            y[i] = y[i - 1]
        else:
            v = (y[i - 1] + y[i + 1]) / T(2)
            y[i] = min(max(v, -lim), lim)
    return y


# --------------------------------------------------------------------------------------
# polyA end detector
# --------------------------------------------------------------------------------------
def polya_end(signal: np.ndarray):
    """riser/preprocess.py:42-79: scan 500-sample windows; the polyA starts at the first
    window whose mean rose > 20 % over the previous 1000 samples with MAD <= 20 and ends
    at the first later window with MAD > 20.  Returns the end index or None."""
    signal = np.asarray(signal)
    start = None
    end = None
    hist = 2 * TRIM_RESOLUTION
    i = 0
    while i + TRIM_RESOLUTION <= signal.shape[0]:
        win = signal[i:i + TRIM_RESOLUTION]
        _, mad = median_mad(win)
        mean = np.mean(win)
        rolling = np.mean(signal[i - hist:i]) if i > hist else mean
        change = (mean - rolling) / rolling * 100
        if not start and change > 20 and mad <= TRIM_MAD_THRESHOLD:
            start = i
        if start and not end and mad > 20:
            end = i
        i += TRIM_RESOLUTION
    return end


def trim_polya(signal, read_id, cache):
    """riser/preprocess.py:87-102."""
    if read_id in cache:
        end = cache[read_id]
    else:
        end = polya_end(signal)
        if end:
            cache[read_id] = end
    if end:
        return signal[end + 1:], True
    return signal, False


# --------------------------------------------------------------------------------------
# network
# --------------------------------------------------------------------------------------
def conv_block(x: np.ndarray, w: np.ndarray, b: np.ndarray, acc=np.float32) -> np.ndarray:
    """One ConvNet layer (riser/nets/cnn.py:52-65 with depth 1): Conv1d(k=3, stride 1,
    zero 'same' padding, bias) -> ReLU -> MaxPool1d(2, 2) (floor: an odd tail sample is
    dropped).  x [B, C_in, L] -> [B, C_out, L // 2]."""
    B, C, L = x.shape
    xp = np.zeros((B, C, L + 2), dtype=acc)
    xp[:, :, 1:L + 1] = x
    w = w.astype(acc)
    y = (np.matmul(w[:, :, 0], xp[:, :, 0:L]) + np.matmul(w[:, :, 1], xp[:, :, 1:L + 1])
         + np.matmul(w[:, :, 2], xp[:, :, 2:L + 2]) + b.astype(acc)[None, :, None])
    y = np.maximum(y, 0)
    Lo = L // 2
    return np.maximum(y[:, :, 0:2 * Lo:2], y[:, :, 1:2 * Lo:2])


def convnet_forward(sd: dict, x: np.ndarray, acc=np.float32, return_layers: bool = False):
    """ConvNet.forward for the shipped `gap_fc` classifier (riser/nets/cnn.py:43-49,
    28-33): x [B, L] -> 12 conv blocks -> mean over length -> Linear -> logits [B, 2]; a state dict with
    `classifier.0.*` keys is the `gap` classifier (gap_head), one with `classifier.1.*` / `classifier.3.*` the `fc` one (fc_head)."""
    n_layers = sum(1 for k in sd if k.startswith("layers.") and k.endswith(".0.weight"))
    h = np.asarray(x, dtype=acc)[:, None, :]
    layers = []
    for i in range(n_layers):
        h = conv_block(h, np.asarray(sd[f"layers.{i}.0.weight"]), np.asarray(sd[f"layers.{i}.0.bias"]), acc)
        if return_layers:
            layers.append(h)
    if h.shape[2] == 0:
        raise RuntimeError("input shorter than 2**n_layers samples")     # torch raises in max_pool1d
    if "classifier.0.weight" in sd:
        logits = gap_head(h, sd, acc)
    elif "classifier.1.weight" in sd:
        logits = fc_head(h, sd, acc)
    else:
        feat = h.mean(axis=2, dtype=acc)
        logits = feat @ np.asarray(sd["classifier.2.weight"]).astype(acc).T + np.asarray(sd["classifier.2.bias"]).astype(acc)
    if return_layers:
        return logits, layers
    return logits


def gap_head(h: np.ndarray, sd: dict, acc=np.float32) -> np.ndarray:
    """The `gap` classifier (riser/nets/cnn.py:34-38): Conv1d(C, n_classes, 1) at every position, then the mean over
    positions (AdaptiveAvgPool1d(1)) - in that order, as the reference computes it.  h [B, C, P] -> logits [B, n_classes]."""
    w = np.asarray(sd["classifier.0.weight"]).astype(acc)[:, :, 0]
    y = np.matmul(w, h) + np.asarray(sd["classifier.0.bias"]).astype(acc)[None, :, None]
    return y.mean(axis=2, dtype=acc)


def fc_head(h: np.ndarray, sd: dict, acc=np.float32) -> np.ndarray:
    """The `fc` classifier (riser/nets/cnn.py:22-27): Flatten(1) (channel-major: f = c * P + p) -> Linear -> ReLU ->
    Linear.  h [B, C, P] -> logits [B, n_classes]; torch raises in the first Linear when C * P is not its in_features."""
    w1 = np.asarray(sd["classifier.1.weight"])
    flat = h.reshape(h.shape[0], -1).astype(acc)
    if flat.shape[1] != w1.shape[1]:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({flat.shape[0]}x{flat.shape[1]} and "
                           f"{w1.shape[1]}x{w1.shape[0]})")
    hid = np.maximum(flat @ w1.astype(acc).T + np.asarray(sd["classifier.1.bias"]).astype(acc), 0)
    return hid @ np.asarray(sd["classifier.3.weight"]).astype(acc).T + np.asarray(sd["classifier.3.bias"]).astype(acc)


def convnet_forward_general(sd: dict, x: np.ndarray, depth: int, acc=np.float32) -> np.ndarray:
    """ConvNet.forward for configurations outside the shipped class (riser/nets/cnn.py:13-18,52-65): per layer `depth` x
    [Conv1d(odd k, stride 1, zero 'same' padding, bias) -> ReLU] then MaxPool1d(2, 2); `gap_fc` classifier.
    State-dict keys layers.{i}.{2 d}.weight / .bias (nn.Sequential indices)."""
    n_layers = sum(1 for k in sd if k.startswith("layers.") and k.endswith(".0.weight"))
    h = np.asarray(x, dtype=acc)[:, None, :]
    for i in range(n_layers):
        for d in range(depth):
            w = np.asarray(sd[f"layers.{i}.{2 * d}.weight"]).astype(acc)
            b = np.asarray(sd[f"layers.{i}.{2 * d}.bias"]).astype(acc)
            k = w.shape[2]
            p = (k - 1) // 2
            B, C, L = h.shape
            hp = np.zeros((B, C, L + k - 1), dtype=acc)
            hp[:, :, p:p + L] = h
            y = b[None, :, None] + sum(np.matmul(w[:, :, t], hp[:, :, t:t + L]) for t in range(k))
            h = np.maximum(y, 0).astype(acc)
        Lo = h.shape[2] // 2
        h = np.maximum(h[:, :, 0:2 * Lo:2], h[:, :, 1:2 * Lo:2])
    if h.shape[2] == 0:
        raise RuntimeError("input shorter than 2**n_layers samples")
    if "classifier.0.weight" in sd:
        return gap_head(h, sd, acc)
    feat = h.mean(axis=2, dtype=acc)
    return feat @ np.asarray(sd["classifier.2.weight"]).astype(acc).T + np.asarray(sd["classifier.2.bias"]).astype(acc)


def softmax(logits: np.ndarray) -> np.ndarray:
    """torch.nn.functional.softmax(dim=1) (riser/model.py:27)."""
    z = logits - logits.max(axis=1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=1, keepdims=True)


def classify(sd: dict, signal: np.ndarray, acc=np.float32) -> np.ndarray:
    """Model.classify (riser/model.py:22-28): normalised signal [L] (float64 or the
    int64 zeros) -> fp32 cast -> net -> softmax -> (p_off, p_on)."""
    x = np.asarray(signal).astype(np.float32)[None, :]
    return softmax(convnet_forward(sd, x, acc))[0].astype(np.float32)


def classify_reads(sd: dict, signals, acc=np.float32) -> np.ndarray:
    """normalise + classify a list of raw reads one by one -> [B, 2] float32."""
    return np.stack([classify(sd, mad_normalise(s), acc) for s in signals])


# --------------------------------------------------------------------------------------
# decision
# --------------------------------------------------------------------------------------
def decide(p_on, p_off, threshold: float, mode: str, sig_len: int, max_len: int) -> str:
    """riser/control.py:75-82 (strict `>` on fp32 values against a Python float)."""
    if any(float(p) > threshold for p in p_on):
        return "accept" if mode == "enrich" else "reject"
    if all(float(p) > threshold for p in p_off):
        return "accept" if mode == "deplete" else "reject"
    if sig_len >= max_len:
        return "no_decision"
    return "try_again"
