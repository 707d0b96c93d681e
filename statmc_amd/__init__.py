"""statmc_amd -- MI355X (gfx950) implementation of StatMC's per-pixel statistics accumulation
and statistics-gated cross-bilateral filter, behind the C ABI of include/statmc.h.

The compute path is libstatmc_hip.so (hand-written HIP, statmc_amd/csrc).  There is no CPU or
PyTorch fallback: importing `statmc_amd.api` without the built library, or calling it without a
gfx950 device, raises.
"""
from . import build as _build  # noqa: F401

__all__ = ["api", "build", "film", "synthetic"]
