"""riser_amd: MI355X-native squiggle-classification hot path (drop-in for comprna/riser's
SignalProcessor.mad_normalise -> Model.classify path and its ReadUntil batching surface).

Importing the package does not touch the GPU; constructing Model / SignalProcessor does,
and raises if the HIP library (riser_amd/lib/libriser_amd.so) or a device is missing.
"""
__all__ = ["Model", "SignalProcessor", "Kit", "SequencerControl"]


def __getattr__(name):
    if name == "Model":
        from .model import Model
        return Model
    if name in ("SignalProcessor", "Kit"):
        from . import preprocess
        return getattr(preprocess, name)
    if name == "SequencerControl":
        from .control import SequencerControl
        return SequencerControl
    raise AttributeError(name)
