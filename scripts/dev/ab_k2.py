# A/B of k_pure_liquid_density: python scripts/dev/ab_k2.py <variant> ... (scratch/ab/lib_<variant>.so)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd import _lib, native
from feos_torch_amd.synthetic import pure_batch, pure_pressures
d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = 10_000_000
P, T = pure_batch(n); pr = pure_pressures(n)
Pd, Td, prd = d(P), d(T), d(pr)
for rnd in range(2):
    for nm in sys.argv[1:]:
        _lib.LIB_PATH = os.path.abspath(f"scratch/ab/lib_{nm}.so"); _lib._lib = None
        ts = []
        for k in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = native.pure_liquid_density(Pd, Td, prd); e1.record(); torch.cuda.synchronize()
            if k: ts.append(e0.elapsed_time(e1))
        print(nm, "liquid_density 1e7: %.3f ms" % np.median(ts), "failed", int(r["status"].sum()), "checksum %.15e" % float(r["rho"].sum()))
