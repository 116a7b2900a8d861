import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd import native
from feos_torch_amd.synthetic import mix_batch
d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = 1_000_000
P, K, T, X, PI = mix_batch(n)
a = [d(v) for v in (P, K, T, X, PI)]
r = native.mix_bubble_dew(*a, False)
rho = r["rho4"][:, 2:4].contiguous()  # liquid densities
rho[r["status"]] = 1e-3
cls = np.arange(n) % 6
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return np.median(ts)
from feos_torch_amd import _lib
if len(sys.argv) > 1:  # variant library scratch/ab/lib_<name>.so
    _lib.LIB_PATH = os.path.abspath(f"scratch/ab/lib_{sys.argv[1]}.so"); _lib._lib = None
ms = t(lambda: native.mix_derivatives(a[0], a[1], a[2], rho))
print(f"k_mix_derivatives (1 T2 evaluation per row, liquid densities) 1e6 rows: {ms:.3f} ms -> {ms*1e-3*2.4e9*1024/ (n/64):.0f} SIMD-cycles per wave-evaluation")
for c in range(6):
    idx = torch.from_numpy(np.where(cls == c)[0]).cuda()
    b = [v[idx].contiguous() for v in (a[0], a[1], a[2], rho)]
    ms = t(lambda: native.mix_derivatives(*b))
    print(f"   class {c}: {ms:.3f} ms per {len(idx)} rows")
