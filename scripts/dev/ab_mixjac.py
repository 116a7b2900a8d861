import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd import _lib, native
from feos_torch_amd.synthetic import mix_batch
d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = 1_000_000
P, K, T, X, PI = mix_batch(n)
a = [d(v) for v in (P, K, T, X, PI)]
r = native.mix_bubble_dew(*a, False)
rho4 = r["rho4"].clone(); rho4[r["status"]] = torch.tensor([1e-6, 1e-6, 5e-3, 5e-3], dtype=torch.float64, device="cuda")
ref = None
for nm in sys.argv[1:]:
    _lib.LIB_PATH = os.path.abspath(f"scratch/ab/lib_{nm}.so"); _lib._lib = None
    ts = []
    for k in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); j = native.mix_jacobian(a[0], a[1], a[2], rho4, False); e1.record(); torch.cuda.synchronize()
        if k: ts.append(e0.elapsed_time(e1))
    ok = ~r["status"]
    if ref is None: ref = j
    dd = ((j - ref).abs() / (ref.abs().max(dim=1, keepdim=True).values + 1e-300))[ok]
    print(nm, "%.1f ms" % np.median(ts), "max rel dev vs first", dd.max().item())
