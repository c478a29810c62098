"""neural_svd_amd: MI355X-native NestedLoRA / NeuralSVD PDE training step.

The compute lives in ``libnsvd_hip.so`` (hand-written HIP for gfx950, C ABI in ``include/nsvd.h``);
this package is the host-side mirror of the reference's Python interface for that path.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
