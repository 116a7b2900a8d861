# A/B of gc bubble / dew / Jacobian kernels: python scripts/dev/ab_gc.py <variant> ... (scratch/ab/lib_<variant>.so)
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd import _lib, native
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
def t(fn, reps=4):
    ts = []
    for k in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
        if k: ts.append(e0.elapsed_time(e1))
    return np.median(ts), r
for nm in sys.argv[1:]:
    _lib.LIB_PATH = os.path.abspath(f"scratch/ab/lib_{nm}.so"); _lib._lib = None
    order = native.gc_class_order(tab, len(ident), rows)
    out = [nm]
    for dew in (False, True):
        ms, r = t(lambda: native.gc_bubble_dew(tab, len(ident), rows, phi, T, x, p0, dew, order=order))
        out.append(f"{'dew' if dew else 'bubble'} {ms:.2f} ms fails {int(r['status'].sum())}")
        if not dew:
            rho4 = r["rho4"].clone(); rho4[r["status"]] = torch.tensor([1e-6, 1e-6, 5e-3, 5e-3], dtype=torch.float64, device="cuda")
            ms0, _ = t(lambda: native.gc_jacobian(tab, len(ident), rows, phi, T, rho4, False)); ms, jj = t(lambda: native.gc_jacobian(tab, len(ident), rows, phi, T, rho4, False, order=order)); j = jj[0] if isinstance(jj, (tuple, list)) else jj
            out.append(f"jacobian {ms0:.2f} -> ordered {ms:.2f} ms sum {float(j[~r['status']].abs().sum()):.12e}")
    print("  ".join(out))
    # segment-parameter gradient at the bubble-point densities of the last variant
    ms, gseg = t(lambda: native.gc_segment_gradient(tab, len(ident), rows, phi, T, rho4, False, order=order))
    print(f"  {nm} segment gradient {ms:.2f} ms  sum {float(gseg.abs().sum()):.12e}")
