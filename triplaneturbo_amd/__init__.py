"""triplaneturbo_amd -- MI355X-native (gfx950) triplane volume-render hot path of TriplaneTurbo.

Host side: Python on PyTorch-ROCm (memory, streams, torch.distributed).  Compute: hand-written HIP
kernels in libtt_hip.so behind the C ABI of include/tt_abi.h.  No CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
