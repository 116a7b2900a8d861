# evaluations per row of the mixture work-queue kernel (diagnostics build scratch/ab/lib_mixdiag2.so, -DPCS_MIX_DIAG=2):
# how much of the kernel time is the tail of long rows
import sys, os
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
    ev = it & 4095; t0 = (it >> 12) & 0xffff  # evaluations, start time in 16384-cycle units
    print("dew" if dew else "bubble", "rows", n, "failed", st.sum(), "total evals %.2fM" % (ev.sum() / 1e6), "per lane of 65536: %.0f" % (ev.sum() / 65536))
    for name, m in (("converged", ~st), ("failed", st)):
        q = np.quantile(ev[m], [.5, .9, .99, .999, 1.0])
        print(f"   {name}: mean {ev[m].mean():.1f} quantiles 50/90/99/99.9/max {q}  share of all evals {ev[m].sum()/ev.sum():.3f}")
    for thr in (50, 100, 150, 200, 300):
        m = ev > thr
        print(f"   rows with > {thr} evals: {m.sum()} (failed among them {(m & st).sum()})")
    for c in range(6):
        m = cls == c
        print(f"   class {c}: mean {ev[m].mean():.1f} max {ev[m].max()} failed {st[m].sum()}  start time median {np.median(t0[m])} max {t0[m].max()}")
    long = ev > 100
    np.save(f"gpurun_out/mix_evals_{'dew' if dew else 'bubble'}.npy", np.stack([ev[:100000], st[:100000].astype(ev.dtype)]))
    print("   start times of rows with > 100 evals: quantiles", np.quantile(t0[long], [0, .5, .9, 1.0]), " all rows max", t0.max())
