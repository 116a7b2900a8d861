# evaluation counts per row of the two mixture queue kernels (needs a diagnostic build that packs them into `iters`)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd import _lib, native
from feos_torch_amd.synthetic import mix_batch
_lib.LIB_PATH = os.path.abspath(f"scratch/ab/lib_{sys.argv[1]}.so"); _lib._lib = None
n = 1_000_000
P, K, T, X, PI = mix_batch(n)
d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
a = [d(v) for v in (P, K, T, X, PI)]
na0, nb0, na1, nb1 = P[:, 0, 6], P[:, 0, 7], P[:, 1, 6], P[:, 1, 7]
assoc = ((na0 + nb0) != 0).astype(int) + ((na1 + nb1) != 0); selfa = ((na0 * nb0) != 0).astype(int) + ((na1 * nb1) != 0)
cls = np.zeros(n, int); cls[(assoc == 1) & (selfa == 1)] = 1; cls[(assoc == 2) & (selfa == 1)] = 2; cls[(assoc == 2) & (selfa == 2)] = 3
for dew in (False, True):
    r = native.mix_bubble_dew(*a, dew, want_iters=True)
    it = r["iters"].cpu().numpy().astype(np.int64); st = r["status"].cpu().numpy()
    ok = it >= 0
    ini_raw, newt, nit = it % 1000, (it // 1000) % 1000, it // 1000000
    print("dew" if dew else "bubble", "rows with counts", int(ok.sum()), "failed", int(st.sum()))
    ini = ini_raw  # (a build that also counts the root-stage evaluations packs them as evals + 128 * root_evals and needs its own decoding)
    for name, v in (("init-kernel evals", ini), ("newton-kernel evals", newt), ("newton iterations", nit)):
        v = v[ok]
        print("   %-20s mean %.2f  q50 %d q90 %d q99 %d q99.9 %d max %d" % (name, v.mean(), *np.quantile(v, [.5, .9, .99, .999]).astype(int), v.max()))
    for c in range(4):
        m = ok & (cls == c)
        print("   class %d: init %.2f newton %.2f" % (c, ini[m].mean(), newt[m].mean()))
