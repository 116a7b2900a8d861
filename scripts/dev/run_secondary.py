"""scripts/dev/run_secondary.py [mix|gc|all] [reps]: launch the bubble / dew solves of configs 4-5 (1e6 rows, synthetic) through the
product library a few times -- the command scripts/profile_secondary.sh wraps in rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd import native
what = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = int(os.environ.get("PCS_ROWS", 1_000_000))
d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
if what in ("mix", "all"):
    from feos_torch_amd.synthetic import mix_batch
    a = [d(v) for v in mix_batch(n)]
    for _ in range(reps):
        for dew in (False, True):
            r = native.mix_bubble_dew(*a, dew)
    torch.cuda.synchronize()
    print("mix failed rows (dew):", int(r["status"].sum()))
if what in ("gc", "all"):
    from feos_torch_amd.gc_pcsaft import build_table, encode_rows
    from feos_torch_amd.synthetic import gc_batch, load_segment_table
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
    for _ in range(reps):
        for dew in (False, True):
            r = native.gc_bubble_dew(tab, len(ident), rows, phi, T, x, p0, dew, order=order)
    torch.cuda.synchronize()
    print("gc failed rows (dew):", int(r["status"].sum()))
