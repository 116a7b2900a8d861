"""CPU: the C-ABI shared library loads and exports every symbol include/pcsaft_hip.h
declares; the Python binding table covers the header; and the product refuses to run
without a GPU instead of falling back to anything."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "pcsaft_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcs_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = header_symbols()
    assert "pcs_pure_vle" in syms and "pcs_last_error" in syms and len(syms) >= 7


def test_library_exports_every_declared_symbol(hip_lib):
    for s in header_symbols():
        assert hasattr(hip_lib, s), f"{s} declared in include/pcsaft_hip.h but not exported"


def test_binding_table_matches_header(hip_lib):
    from feos_torch_amd import _lib

    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_abi_version_and_error_string(hip_lib):
    assert hip_lib.pcs_abi_version() >= 100
    assert hip_lib.pcs_last_error() == b""
    assert hip_lib.pcs_workspace_bytes(1000) == 4 * (1000 + 64)  # row order / retry list + control block


def test_product_has_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from feos_torch_amd import PcSaftPure, _lib

    eos = PcSaftPure(torch.tensor([[1.5, 3.5, 250.0, 0, 0.03, 1500.0, 1, 1]], dtype=torch.float64))
    with pytest.raises(_lib.PcsError):
        eos.vapor_pressure(torch.tensor([300.0], dtype=torch.float64))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "feos_torch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in src and "liboracle" not in src and "from oracle" not in src, f
