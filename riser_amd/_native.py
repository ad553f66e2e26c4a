"""ctypes binding of libriser_amd.so (the C ABI declared in include/riser_amd.h).

There is deliberately no fallback: if the HIP library is missing or fails to load, every
entry point of the package raises.  Build it with `python -m riser_amd.build`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RISER_AMD_LIB") or os.path.join(_HERE, "lib", "libriser_amd.so")

RS_OK = 0
RS_ERR_ARG, RS_ERR_HIP, RS_ERR_LENGTH, RS_ERR_OOM, RS_ERR_WORKSPACE = -1, -2, -3, -4, -5
RS_F32, RS_BF16, RS_F16, RS_F32W, RS_BF16X3, RS_F16X3, RS_F16XF8 = 0, 1, 2, 3, 4, 5, 6
RS_TRY_AGAIN, RS_ACCEPT, RS_REJECT, RS_NO_DECISION = 0, 1, 2, 3
RS_ENRICH, RS_DEPLETE = 0, 1
DECISION_NAMES = ("try_again", "accept", "reject", "no_decision")

# every symbol include/riser_amd.h declares (tests check the .so exports all of them)
SYMBOLS = (
    "rs_last_error", "rs_version", "rs_device_count", "rs_model_create", "rs_model_destroy", "rs_model_set_fc_classifier",
    "rs_workspace_bytes", "rs_max_batch", "rs_block_samples", "rs_normalise", "rs_normalise_float", "rs_forward", "rs_padded_length", "rs_classify",
    "rs_classify_ensemble", "rs_ensemble_workspace_bytes", "rs_autotune", "rs_decide", "rs_polya_end", "rs_copy_segments", "rs_model_layer_info", "rs_profile_enable", "rs_profile_read",
    "rs_debug_capture_layer", "rs_polya_end_resume", "rs_model_saturated",
    "rs_seqnet_create", "rs_seqnet_destroy", "rs_seqnet_workspace_bytes", "rs_seqnet_forward", "rs_seqnet_set_mode", "rs_seqnet_ragged_ok", "rs_seqnet_forward_ragged", "rs_seqnet_max_batch",
)


class SeqOp(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "src", "dst", "add", "c_in", "c_out", "k", "stride", "pad", "relu")] + \
               [("w", C.c_void_p), ("b", C.c_void_p)]


class LayerInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("c_in", "c_out", "cp_in", "cp_out", "k_pad", "n_pad", "bm", "bn", "kc", "gemm_row_div", "block_samples", "rows_format")]


class NativeError(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and return the shared library; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its bundled HIP runtime must be the one already in the process when libriser_amd.so resolves
    # libamdhip64 (loaded the other way round, the system runtime comes up beside torch's and sees no device)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise NativeError(f"{LIB_PATH} not found: the HIP extension is not built "
                          "(run `python -m riser_amd.build`); riser_amd has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
    L.rs_last_error.restype = C.c_char_p
    L.rs_last_error.argtypes = []
    L.rs_version.restype = i32
    L.rs_device_count.restype = i32
    L.rs_model_create.restype = i32
    L.rs_model_create.argtypes = [i32, C.POINTER(C.c_int32), i32, C.POINTER(vp), C.POINTER(vp), vp, vp,
                                  i32, i32, C.POINTER(vp)]
    L.rs_model_destroy.restype = i32
    L.rs_model_destroy.argtypes = [vp]
    L.rs_model_set_fc_classifier.restype = i32
    L.rs_model_set_fc_classifier.argtypes = [vp, i32, i32, vp, vp, vp, vp]
    L.rs_workspace_bytes.restype = sz
    L.rs_workspace_bytes.argtypes = [vp, i32, i32]
    L.rs_padded_length.restype = i32
    L.rs_padded_length.argtypes = [vp, i32]
    L.rs_max_batch.restype = i32
    L.rs_max_batch.argtypes = [vp, i32]
    L.rs_block_samples.restype = i32
    L.rs_block_samples.argtypes = [vp]
    L.rs_normalise.restype = i32
    L.rs_normalise.argtypes = [vp, vp, vp, i32, i32, vp, i64, C.c_int32, vp, i64, vp, vp]
    L.rs_normalise_float.restype = i32
    L.rs_normalise_float.argtypes = [vp, i32, vp, vp, i32, vp, i64, vp, vp]
    L.rs_forward.restype = i32
    L.rs_forward.argtypes = [vp, vp, i64, vp, vp, i32, i32, i32, vp, sz, vp, vp, vp]
    L.rs_classify.restype = i32
    L.rs_classify.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, sz, vp, vp, vp]
    L.rs_autotune.restype = i32
    L.rs_autotune.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, sz, vp, C.POINTER(C.c_int32), vp]
    L.rs_classify_ensemble.restype = i32
    L.rs_classify_ensemble.argtypes = [C.POINTER(vp), i32, vp, vp, vp, vp, i32, i32, i32, vp, sz, vp, vp, i32, C.c_float,
                                       i32, vp]
    L.rs_ensemble_workspace_bytes.restype = sz
    L.rs_ensemble_workspace_bytes.argtypes = [C.POINTER(vp), i32, i32, i32]
    L.rs_decide.restype = i32
    L.rs_decide.argtypes = [vp, i32, i32, vp, i32, C.c_float, i32, vp, vp]
    L.rs_polya_end.restype = i32
    L.rs_polya_end.argtypes = [vp, vp, vp, i32, vp, vp]
    L.rs_copy_segments.restype = i32
    L.rs_copy_segments.argtypes = [vp, vp, vp, vp, vp, i32, vp]
    L.rs_model_layer_info.restype = i32
    L.rs_model_layer_info.argtypes = [vp, i32, C.POINTER(LayerInfo)]
    L.rs_seqnet_create.restype = i32
    L.rs_seqnet_create.argtypes = [C.POINTER(SeqOp), i32, i32, vp, vp, i32, i32, C.POINTER(vp)]
    L.rs_seqnet_destroy.restype = i32
    L.rs_seqnet_destroy.argtypes = [vp]
    L.rs_seqnet_workspace_bytes.restype = sz
    L.rs_seqnet_workspace_bytes.argtypes = [vp, i32, i32]
    L.rs_seqnet_forward.restype = i32
    L.rs_seqnet_forward.argtypes = [vp, vp, i32, i32, vp, sz, vp, vp, vp]
    L.rs_seqnet_set_mode.restype = i32
    L.rs_seqnet_set_mode.argtypes = [vp, i32]
    L.rs_seqnet_ragged_ok.restype = i32
    L.rs_seqnet_ragged_ok.argtypes = [vp]
    L.rs_seqnet_max_batch.restype = i32
    L.rs_seqnet_max_batch.argtypes = [vp, i32]
    L.rs_seqnet_forward_ragged.restype = i32
    L.rs_seqnet_forward_ragged.argtypes = [vp, vp, vp, i32, i32, vp, sz, vp, vp, vp]
    L.rs_polya_end_resume.restype = i32
    L.rs_polya_end_resume.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp]
    L.rs_debug_capture_layer.restype = i32
    L.rs_debug_capture_layer.argtypes = [vp, i32, vp, sz]
    L.rs_model_saturated.restype = i32
    L.rs_model_saturated.argtypes = [vp, i32, vp]
    L.rs_profile_enable.restype = i32
    L.rs_profile_enable.argtypes = [vp, i32]
    L.rs_profile_read.restype = i32
    L.rs_profile_read.argtypes = [vp, vp, C.POINTER(C.c_int32)]
    _lib = L
    return L


def check(rc: int, what: str = "") -> None:
    """Raise on a negative rs_status: ValueError for length problems (the reference raises
    ValueError for an empty signal, riser/preprocess.py:109-110), RuntimeError otherwise."""
    if rc == RS_OK:
        return
    msg = lib().rs_last_error().decode("utf-8", "replace")
    if rc == -3:
        raise ValueError(f"{what}: {msg}")
    raise NativeError(f"{what}: rs_status {rc}: {msg}")


def require_gpu() -> None:
    if lib().rs_device_count() < 1:
        raise NativeError("no HIP device visible: riser_amd runs on MI355X only and has no CPU path")
