"""ctypes loader for ``libpcsaft_hip.so`` — the only native entry into the product path.

There is deliberately NO fallback: if the HIP library is missing or a call fails, the
product raises.  (The CPU restatement under ``oracle/`` is test infrastructure and is never
imported from here.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PCS_HIP_LIB: development aid for A/B runs of variant builds (scripts/dev/mk*.sh -> scratch/ab/lib_<name>.so); still a
# libpcsaft_hip build, never a fallback
LIB_PATH = os.environ.get("PCS_HIP_LIB") or os.path.join(_HERE, "libpcsaft_hip.so")

_lib = None

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int

# name -> (restype, argtypes); mirrors include/pcsaft_hip.h one to one
SIGNATURES = {
    "pcs_abi_version": (_int, []),
    "pcs_last_error": (ctypes.c_char_p, []),
    "pcs_workspace_bytes": (_i64, [_i64]),
    "pcs_pure_vle": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcs_pure_vapor_pressure": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "pcs_compact_workspace_bytes": (_i64, [_i64]),
    "pcs_compact_plan": (_int, [_vp, _i64, _vp, _vp]),
    "pcs_compact_rows": (_int, [_vp, _i64, _vp, _vp, _int, _vp, _vp, _vp]),
    "pcs_expand_rows": (_int, [_vp, _i64, _vp, _vp, _vp, _int, _int, _int, _vp, _vp]),
    "pcs_pure_vle_fast": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcs_pure_vle_fp64": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcs_pure_vle_retry": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcs_pure_liquid_density": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "pcs_pure_derivatives": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "pcs_pure_jacobian": (_int, [_int, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "pcs_pure_jacobian_vjp": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "pcs_mix_workspace_bytes": (_i64, [_i64]),
    "pcs_mix_bubble_dew": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcs_mix_jacobian": (_int, [_int, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "pcs_gc_table_doubles": (_i64, [_int]),
    "pcs_gc_bubble_dew": (_int, [_int, _vp, _int, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcs_gc_derivatives": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "pcs_gc_jacobian": (_int, [_int, _vp, _int, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "pcs_gc_segment_gradient": (_int, [_int, _vp, _int, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "pcs_pure_derivatives_vjp": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcs_mix_derivatives_vjp": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcs_gc_derivatives_vjp": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcs_mixn_derivatives": (_int, [_vp, _vp, _vp, _int, _i64, _vp, _vp, _vp, _vp, _vp]),
    "pcs_mixn_derivatives_vjp": (_int, [_vp, _vp, _vp, _int, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcs_mix_derivatives": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
}


class PcsError(RuntimeError):
    pass


def lib():
    """Load the library (once).  Raises if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PcsError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().pcs_last_error().decode("utf-8", "replace")
        raise PcsError(f"{what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def current_stream_ptr(device):
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
