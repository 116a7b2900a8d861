"""scripts/dev/pmc_run_gcgrad.py: launch the gc segment-gradient kernel twice on 1e6 rows (for rocprofv3 --pmc; summarise with
pmc_sum.py <dir> k_gc_segment_gradient)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd import native
from feos_torch_amd.gc_pcsaft import encode_rows, build_table
from feos_torch_amd.synthetic import gc_batch, load_segment_table
d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = 1_000_000
table = load_segment_table(os.path.join(ROOT, "tests", "data", "sauer2014_hetero.json"))
b = gc_batch(n, table); ident = [s for s, _ in table]
rows = d(encode_rows(ident, b["segment_lists"], b["bond_lists"]))
seg = torch.tensor(np.stack([v for _, v in table]), dtype=torch.float64)
kab = torch.zeros((len(ident), len(ident)), dtype=torch.float64)
for s1, s2, k in b["kab_list"]:
    kab[ident.index(s1), ident.index(s2)] = k; kab[ident.index(s2), ident.index(s1)] = k
tab = build_table(seg.cuda(), kab.cuda())
phi, T, x, p0 = d(b["phi"]), d(b["T"]), d(b["x"]), d(b["p_init"])
order = native.gc_class_order(tab, len(ident), rows)
r = native.gc_bubble_dew(tab, len(ident), rows, phi, T, x, p0, False, order=order)
rho4 = r["rho4"].clone(); rho4[r["status"]] = torch.tensor([1e-6, 1e-6, 5e-3, 5e-3], dtype=torch.float64, device="cuda")
for _ in range(2):
    native.gc_segment_gradient(tab, len(ident), rows, phi, T, rho4, False, order=order)
    native.gc_jacobian(tab, len(ident), rows, phi, T, rho4, False, order=order)
torch.cuda.synchronize()
