"""feos_torch_amd — MI355X-native batched PC-SAFT phase equilibria.

Drop-in for the Python API of feos-torch (``from feos_torch import PcSaftPure, ...``): same
class names, signatures, units and return conventions; the native half (Rust + feos in the
reference) is ``libpcsaft_hip.so`` — hand-written gfx950 kernels behind the C ABI declared in
``include/pcsaft_hip.h``.
"""
from .native import PcSaft  # noqa: F401  (mirror of the reference's extension class)
from .pcsaft_pure import PcSaftPure  # noqa: F401
from .pcsaft_mix import PcSaftMix  # noqa: F401
from .gc_pcsaft import GcPcSaft, GcPcSaftMix  # noqa: F401

__version__ = "0.1.0"
