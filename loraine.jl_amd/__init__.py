"""loraine.jl_amd -- MI355X (gfx950) implementation of the Loraine.jl per-iteration hot path
(NT scaling, Schur-complement assembly, Cholesky / PCG solve) behind the reference's own
solver interface.  Native code: csrc/*.hip -> libloraine_hip.so (C ABI, include/loraine_hip.h).
"""
from ._capi import LIB_PATH, LoraineHipError, load_library  # noqa: F401
from .device import Device  # noqa: F401

__all__ = ["Device", "LoraineHipError", "load_library", "LIB_PATH"]
