import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd import _lib
_lib.LIB_PATH = os.path.abspath("scratch/ab/lib_mixdiag2.so")
from feos_torch_amd import native
from feos_torch_amd.synthetic import mix_batch
d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = 1_000_000
P, K, T, X, PI = mix_batch(n)
a = [d(v) for v in (P, K, T, X, PI)]
cls = np.arange(n) % 6
for dew in (True, False):
    r = native.mix_bubble_dew(*a, dew, want_iters=True)
    it = r["iters"].cpu().numpy(); st = r["status"].cpu().numpy().astype(bool)
    retry = (it >> 30) & 1 == 1
    nph = it & 4095; nli = (it >> 12) & 4095
    print("dew" if dew else "bubble", "retry rows", retry.sum())
    f = ~retry
    for name, v in (("phase evals (T2)", nph), ("line evals (D2)", nli)):
        w = np.where(f, v, 0)[: n // 64 * 64].reshape(-1, 64)
        print(f"   fast pass {name}: lane mean {v[f].mean():.1f} quantiles 50/90/99/max {np.quantile(v[f], [.5,.9,.99,1.0])}  wave-max mean {w.max(1).mean():.1f}")
    for name, v in (("phase evals (T2)", nph), ("line evals (D2)", nli)):
        if retry.any(): print(f"   retry pass {name}: lane mean {v[retry].mean():.1f} quantiles 50/90/99/max {np.quantile(v[retry], [.5,.9,.99,1.0])}")
    for c in range(6):
        m = f & (cls == c); print(f"   class {c}: phase {nph[m].mean():.1f} line {nli[m].mean():.1f}")
