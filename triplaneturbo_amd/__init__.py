"""triplaneturbo_amd -- MI355X-native (gfx950) triplane volume-render hot path of TriplaneTurbo.

Host side: Python on PyTorch-ROCm (memory, streams, torch.distributed).  Compute: hand-written HIP
kernels in libtt_hip.so behind the C ABI of include/tt_abi.h.  No CPU fallback.

    from triplaneturbo_amd import find, register      # threestudio-style plugin registry
    Renderer = find("generative-space-sdf-volume-renderer")
"""
from . import _lib  # noqa: F401
from .registry import C, find, register  # noqa: F401


def _register_plugins():
    from . import background, geometry, renderer  # noqa: F401  (import = registration)


_register_plugins()
__all__ = ["find", "register", "C"]
