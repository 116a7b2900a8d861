# ceiling of a batch-wide class order for the gc kernels: host-sorted rows against the seeded order
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
rows = encode_rows(ident, b["segment_lists"], b["bond_lists"])
segv = np.stack([v for _, v in table])
ids, cnt = rows[:, 0:16].astype(int), rows[:, 16:32].astype(float)
par = segv[ids]  # [n,16,8]
def per_mol(col): return (cnt * par[:, :, col]).reshape(n, 2, 8).sum(axis=2)
ka, eab, na, nb = per_mol(4), per_mol(5), per_mol(6), per_mol(7)
mu2 = (cnt * par[:, :, 3] ** 2).reshape(n, 2, 8).sum(axis=2)
associating = ((ka * eab) != 0).sum(axis=1); self_assoc = ((na * nb) != 0).sum(axis=1)
cls = np.zeros(n, int); cls[(associating == 1) & (self_assoc == 1)] = 1; cls[(associating == 2) & (self_assoc == 1)] = 2; cls[(associating == 2) & (self_assoc == 2)] = 3
key = 2 * cls + (mu2 > 0).any(axis=1)
print("class shares", np.bincount(key, minlength=8) / n)
seg = torch.tensor(segv, dtype=torch.float64)
kab = torch.zeros((len(ident), len(ident)), dtype=torch.float64)
for s1, s2, k in b["kab_list"]:
    kab[ident.index(s1), ident.index(s2)] = k; kab[ident.index(s2), ident.index(s1)] = k
tab = build_table(seg.cuda(), kab.cuda())
def t(fn, reps=4):
    ts = []
    for k in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
        if k: ts.append(e0.elapsed_time(e1))
    return np.median(ts)
nb0 = (rows[:, 64:72] > 0).sum(axis=1); nb1 = (rows[:, 72:80] > 0).sum(axis=1)  # bond-type entries per molecule
ns0 = (rows[:, 16:24] > 0).sum(axis=1); ns1 = (rows[:, 24:32] > 0).sum(axis=1)  # segment-type entries per molecule
for name, order in (("seeded order", np.arange(n)), ("sorted by class", np.argsort(key, kind="stable")),
                    ("class, bond entries", np.lexsort((nb0 + nb1, key))), ("class, max bonds, segs", np.lexsort((ns0 + ns1, np.maximum(nb0, nb1), key))),
                    ("class, T", np.lexsort((b["T"], key)))):
    a = [d(v[order]) for v in (rows, b["phi"], b["T"], b["x"], b["p_init"])]
    out = [f"{name:24s}"]
    for dew in (False, True):
        out.append(f"{'dew' if dew else 'bubble'} {t(lambda: native.gc_bubble_dew(tab, len(ident), a[0], a[1], a[2], a[3], a[4], dew)):.2f} ms")
    print("  ".join(out))
