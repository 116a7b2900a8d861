"""Diagnostics for single rows of the benchmark batch: p_sat from the pressure-only kernel and from the all-fp64 kernel vs the
long-double oracle, iteration counts and the work-list path.  python tests/tools/pure_rows_diag.py <row> [<row> ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from feos_torch_amd import native
from feos_torch_amd.synthetic import pure_batch
from oracle import pyoracle as orc

rows = [int(a) for a in sys.argv[1:]]
P, T = pure_batch(10_000_000, seed=2026)
P, T = np.ascontiguousarray(P[rows]), np.ascontiguousarray(T[rows])
want, _ = orc.pure_vapor_pressure(P, T, prec=1)
rv, rl, st, it, path = orc.pure_vle(P, T, prec=1)
Pd, Td = torch.from_numpy(P).cuda(), torch.from_numpy(T).cuda()
lite = native.pure_vle(Pd, Td, want_rho_vl=False, want_iters=True)
full = native.pure_vle(Pd, Td, want_rho_vl=True, want_iters=True)
plan = native.PureVlePlan(len(rows), Pd.device)
plan.run_fast(Pd, Td)
torch.cuda.synchronize()
fb = plan.retry_count()
for k, r in enumerate(rows):
    pl, pf = lite["p_sat"][k].item(), full["p_sat"][k].item()
    print(f"row {r}: oracle p {want[k]:.15e} | lite rel {abs(pl/want[k]-1):.2e} iters {int(lite['iters'][k])} | fp64 rel {abs(pf/want[k]-1):.2e} iters {int(full['iters'][k])}"
          f" | rho_l GPU/oracle-1 {full['rho_vl'][k,1].item()/rl[k]-1:.2e} rho_v {full['rho_vl'][k,0].item()/rv[k]-1:.2e}")
print("work list of the lite launch (fallback rows, robust rows):", fb)
