"""MI355X-native Show-Attend-and-Tell train-step hot path.

Python here is plumbing only (parameters, device memory, streams, torch.distributed);
every contraction, the attention step, the LSTM cell, the losses and the conv stack
run in ``libsat_hip.so`` (hand-written HIP for gfx950, C ABI in include/sat_hip.h).
There is no CPU or PyTorch-eager fallback: importing works anywhere, computing
raises if the library or a GPU is missing.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
