"""Ceiling of bucketing: same lib, unsorted vs rows fully sorted by (class, tau) on the host."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd.synthetic import pure_batch
from feos_torch_amd import native
n = 10_000_000
P, T = pure_batch(n)
cls = (P[:, 3] > 0).astype(int) + 2 * (P[:, 4] > 0).astype(int)
tau = T / (1.28 * P[:, 2] * P[:, 0] ** 0.45)
def timeit(P, T, label):
    Pd, Td = torch.from_numpy(np.ascontiguousarray(P)).cuda(), torch.from_numpy(np.ascontiguousarray(T)).cuda()
    plan = native.PureVlePlan(len(T), Pd.device)
    ts = []
    for r in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run(Pd, Td); e1.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(e0.elapsed_time(e1))
    print(f"{label:28s} n={len(T):9d} median {np.median(ts):.3f} ms  {np.median(ts)/len(T)*1e6:.4f} ns/row")
timeit(P, T, "unsorted")
o = np.argsort(cls, kind="stable")
timeit(P[o], T[o], "sorted by class")
o = np.lexsort((tau, cls))
timeit(P[o], T[o], "sorted by class, tau")
o = np.argsort(tau)
timeit(P[o], T[o], "sorted by tau only")
for c in range(4):
    m = cls == c
    timeit(P[m], T[m], f"class {c} only")
    o = np.argsort(tau[m])
    timeit(P[m][o], T[m][o], f"class {c} only, tau-sorted")
