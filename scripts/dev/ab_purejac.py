# A/B of the pure-component Jacobian kernels: python scripts/dev/ab_purejac.py <variant> ... (scratch/ab/lib_<variant>.so)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd import _lib, native
from feos_torch_amd.synthetic import pure_batch, pure_pressures
d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = 10_000_000
P, T = pure_batch(n); pr = pure_pressures(n)
Pd, Td, prd = d(P), d(T), d(pr)
r = native.pure_vle(Pd, Td, want_rho_vl=True)
rho_vl = r["rho_vl"]
rl = native.pure_liquid_density(Pd, Td, prd)
rho_l = torch.stack([torch.zeros_like(rl["rho_root"]), rl["rho_root"]], dim=1)
ref = {}
for nm in sys.argv[1:]:
    _lib.LIB_PATH = os.path.abspath(f"scratch/ab/lib_{nm}.so"); _lib._lib = None
    out = [nm]
    for which, dens in (("vapor_pressure", rho_vl), ("liquid_density", rho_l), ("equilibrium_liquid_density", rho_vl)):
        ts = []
        for k in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); j = native.pure_jacobian(which, Pd, Td, prd if which == "liquid_density" else None, dens); e1.record(); torch.cuda.synchronize()
            if k: ts.append(e0.elapsed_time(e1))
        if which not in ref: ref[which] = j
        dev = ((j - ref[which]).abs() / (ref[which].abs().max(dim=1, keepdim=True).values + 1e-300)).max().item()
        out.append(f"{which} {np.median(ts):.2f} ms (dev {dev:.1e})")
    print("  ".join(out))
