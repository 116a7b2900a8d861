"""Generate the golden fixtures in this directory by running the UNMODIFIED reference Python
(/root/reference/feos_torch) in the build container.  Never runs on the GPU box and is not
imported by any test; the JSON files it writes are what travels.

The reference's Rust extension and the `si_units` package are absent (no Rust toolchain, no
network), so two in-memory stand-in modules are registered before the import (recipe:
SURVEY.md Appendix A):
  * `si_units`: SI base units as plain floats;
  * `feos_torch.feos_torch`: `PcSaft` / `GcPcSaft` whose solver entry points return the
    converged densities computed by this repo's CPU oracle (oracle/, long-double solve).
Everything downstream of the densities — Helmholtz energy, dual-number derivatives, the final
Newton-step formulas, unit conversions and torch autograd gradients — is the reference's own
code, so the fixtures pin all of it.  The densities themselves are pinned by README.md:26-29
(the only stored answers in the reference) through `readme_*` below.

Usage:  python3 -B tests/golden/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import pyoracle as orc  # noqa: E402

si = types.ModuleType("si_units")
si.KELVIN = si.PASCAL = si.MOL = si.METER = si.JOULE = 1.0
si.KILO, si.ANGSTROM, si.KB, si.NAV = 1e3, 1e-10, 1.380649e-23, 6.02214076e23
sys.modules["si_units"] = si


class _PcSaft:
    """Stand-in for the Rust pyclass of src/pcsaft.rs:13-80 backed by the CPU oracle."""

    @staticmethod
    def vapor_pressure(parameters, temperature):
        rv, rl, st, _, _ = orc.pure_vle(parameters, temperature, prec=1)
        rho = np.zeros((int((~st).sum()), 4))
        rho[:, 0] = rv[~st]
        rho[:, 1] = rl[~st]
        return rho, st

    @staticmethod
    def liquid_density(parameters, temperature, pressure):
        rho, st = orc.pure_liquid_density_root(parameters, temperature, pressure)
        return rho[~st], st

    @staticmethod
    def bubble_point(parameters, kij, temperature, x, pressure):
        rho, st = orc.mix_bubble_dew_root(parameters, kij, temperature, x, pressure, dew=False)
        return rho[~st], st

    @staticmethod
    def dew_point(parameters, kij, temperature, y, pressure):
        rho, st = orc.mix_bubble_dew_root(parameters, kij, temperature, y, pressure, dew=True)
        return rho[~st], st


class _GcPcSaft:
    """Stand-in for src/gc_pcsaft.rs:15-99 backed by the CPU oracle."""

    def __init__(self, segment_records, segments, bonds, binary_segment_records, phi):
        self.args = (segment_records, segments, bonds, binary_segment_records, np.array(phi, dtype=np.float64))

    def bubble_point(self, temperature, x, pressure):
        rho, st = orc.gc_bubble_dew_root(*self.args, temperature, x, pressure, dew=False)
        return rho[~st], st

    def dew_point(self, temperature, y, pressure):
        rho, st = orc.gc_bubble_dew_root(*self.args, temperature, y, pressure, dew=True)
        return rho[~st], st


ext = types.ModuleType("feos_torch.feos_torch")
ext.PcSaft = _PcSaft
ext.GcPcSaft = _GcPcSaft
sys.modules["feos_torch.feos_torch"] = ext
sys.path.insert(0, "/root/reference")
from feos_torch import PcSaftPure, PcSaftMix, GcPcSaftMix  # noqa: E402  (unmodified reference code)

f64 = torch.float64


def tl(x):
    return x.detach().cpu().numpy().tolist()


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)
    print("wrote", name)


# ------------------------------------------------------------------------------------------
# pure component
# ------------------------------------------------------------------------------------------
# inputs of tests/test_pcsaft_pure.py:10-21
PURE_TEST_PARAMS = [
    [1.5, 3.2, 350, 0, 0, 0, 0, 0],
    [1.5, 3.2, 150, 2.5, 0.03, 2500, 2, 1],
    [1.5, 3.2, 150, 2.5, 0, 2500, 1, 1],
    [1.5, 3.2, 150, 2.5, 0.03, 0, 1, 1],
    [1.5, 3.2, 150, 2.5, 0, 0, 0, 0],
    [1.5, 3.2, 150, 2.5, 0.03, 2500, 0, 2],
]
README_PARAMS = [1.5, 3.5, 250.0, 0, 0.03, 1500.0, 1, 1]  # README.md:13
README_T = [250.0, 300.0, 350.0, 400.0, 450.0]  # README.md:17


def pure_properties(params, T, p_spec):
    """Run the three reference properties + autograd on (params, T[, p])."""
    out = {}
    n = len(T)
    for name in ("vapor_pressure", "liquid_density", "equilibrium_liquid_density"):
        x = torch.tensor(params, dtype=f64, requires_grad=True)
        Tt = torch.tensor(T, dtype=f64, requires_grad=True)
        eos = PcSaftPure(x)
        if name == "liquid_density":
            pt = torch.tensor(p_spec, dtype=f64, requires_grad=True)
            nans, val = eos.liquid_density(Tt, pt)
        else:
            pt = None
            nans, val = getattr(eos, name)(Tt)
        # per-row gradients: rows are independent, so d(sum)/d(row i inputs) = d(val_i)/d(...)
        val.sum().backward()
        out[name] = {
            "nans": tl(nans),
            "value": tl(val),
            "grad_params": tl(x.grad),
            "grad_T": tl(Tt.grad),
        }
        if pt is not None:
            out[name]["grad_p"] = tl(pt.grad)
    return out


def make_pure():
    g = {}
    # (a, p, dp) on the reference test inputs, tests/test_pcsaft_pure.py:19-29
    x = torch.tensor(PURE_TEST_PARAMS, dtype=f64)
    T = torch.tensor([300.0] * 6, dtype=f64)
    rho = torch.tensor([0.001] * 6, dtype=f64)
    a, p, dp = PcSaftPure(x).derivatives(T, rho)
    g["test_inputs"] = {"params": PURE_TEST_PARAMS, "T": tl(T), "rho": tl(rho), "a": tl(a), "p": tl(p), "dp": tl(dp)}
    g["test_inputs"]["properties"] = pure_properties(PURE_TEST_PARAMS, [300.0] * 6, [1e5] * 6)

    # README fluid
    xr = torch.tensor([README_PARAMS] * 5, dtype=f64)
    Tr = torch.tensor(README_T, dtype=f64)
    rho = torch.tensor([0.001] * 5, dtype=f64)
    a, p, dp = PcSaftPure(xr).derivatives(Tr, rho)
    g["readme"] = {"params": [README_PARAMS] * 5, "T": README_T, "rho": tl(rho), "a": tl(a), "p": tl(p), "dp": tl(dp),
                   "readme_vapor_pressure_printed": [20693.5960, 216164.6184, 1049770.6187, 3281855.9640, 7875531.7021],
                   "readme_grad_printed": [-6.7923e04, -1.7737e04, -7.0413e02, 0.0, -5.7458e05, -6.9122e01, -3.6892e04, -3.6892e04]}
    g["readme"]["properties"] = pure_properties([README_PARAMS] * 5, README_T, [1e5] * 5)

    # seeded random rows of the benchmark distribution at random liquid- and vapour-like densities
    from feos_torch_amd.synthetic import pure_batch, pure_pressures
    P, TT = pure_batch(96, seed=7)
    rng = np.random.default_rng(11)
    eta = np.where(rng.random(96) < 0.5, rng.uniform(0.2, 0.45, 96), 10.0 ** rng.uniform(-8, -1.5, 96))
    d = P[:, 1] * (1 - 0.12 * np.exp(-3 * P[:, 2] / TT))
    rho = eta / (np.pi / 6 * P[:, 0] * d**3)
    a, p, dp = PcSaftPure(torch.tensor(P, dtype=f64)).derivatives(torch.tensor(TT, dtype=f64), torch.tensor(rho, dtype=f64))
    g["random"] = {"params": P.tolist(), "T": TT.tolist(), "rho": rho.tolist(), "a": tl(a), "p": tl(p), "dp": tl(dp)}
    psp = np.maximum(pure_pressures(96, seed=13), 0.0)
    g["random"]["p_spec"] = psp.tolist()
    g["random"]["properties"] = pure_properties(P.tolist(), TT.tolist(), psp.tolist())
    dump("pure.json", g)


if __name__ == "__main__":
    which = sys.argv[1:] or ["pure", "mix", "gc"]
    if "pure" in which:
        make_pure()
    sys.path.insert(0, HERE)
    if "mix" in which:
        from make_golden_mix import make_mix
        make_mix(PcSaftMix, dump, tl)
    if "gc" in which:
        from make_golden_mix import make_gc
        make_gc(GcPcSaftMix, dump, tl)
    if "gc_seggrad" in which or len(sys.argv) == 1:
        from make_golden_mix import make_gc_seggrad
        make_gc_seggrad(GcPcSaftMix, dump, tl)
    if "mixn" in which or len(sys.argv) == 1:
        from make_golden_mix import make_mixn
        make_mixn(PcSaftMix, dump, tl)
    if "deriv_grad" in which or len(sys.argv) == 1:  # needs pure.json and mix.json (inputs are re-used)
        from make_golden_mix import make_deriv_grad
        make_deriv_grad(PcSaftPure, PcSaftMix, GcPcSaftMix, dump, tl, PURE_TEST_PARAMS)
