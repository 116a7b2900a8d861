# evaluation counts of the gc dew rows per pass (needs a diagnostic build that writes them to `iters`: see DESIGN 4b)
import sys, os
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
_lib.LIB_PATH = os.path.abspath(f"scratch/ab/lib_{sys.argv[1]}.so"); _lib._lib = None
order = native.gc_class_order(tab, len(ident), rows)
for dew in (False, True):
    r = native.gc_bubble_dew(tab, len(ident), rows, phi, T, x, p0, dew, want_iters=True, order=order)
    it = r["iters"].cpu().numpy(); st = r["status"].cpu().numpy()
    retry = it >= 1000000; it = it % 1000000; rob = it >= 100000; ev = it % 100000
    print("dew" if dew else "bubble", "rows", n, "retry rows", int(retry.sum()), "robust", int(rob.sum()), "failed", int(st.sum()))
    print("  fast-pass evals: mean %.2f  q50 %d q90 %d q99 %d max %d" % (ev[~retry].mean(), *np.quantile(ev[~retry], [.5, .9, .99]).astype(int), ev[~retry].max()))
    if retry.any():
        e = ev[retry]
        print("  retry evals: mean %.1f q50 %d q90 %d q99 %d max %d; > 40: %d, > 80: %d" % (e.mean(), *np.quantile(e, [.5, .9, .99]).astype(int), e.max(), int((e > 40).sum()), int((e > 80).sum())))
        print("  retry failed rows evals:", np.sort(ev[retry & st])[-10:])
