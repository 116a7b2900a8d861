# per-stage times and list sizes of pcs_pure_vle for the three output sets (pressure only / + densities / all-fp64 twin)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd import native
from feos_torch_amd.synthetic import pure_batch
n = 10_000_000
P, T = pure_batch(n)
Pd, Td = torch.from_numpy(P).cuda(), torch.from_numpy(T).cuda()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for label, kw in (("pressure only", {}), ("with rho_eq", {"want_rho_eq": True}), ("with rho_vl", {"want_rho_vl": True}), ("all fp64 + rho_eq", {"want_rho_eq": True, "all_fp64": True})):
    plan = native.PureVlePlan(n, "cuda", **kw)
    if kw.get("all_fp64"):
        print(label, "total %.3f ms" % t(lambda: plan.run(Pd, Td)))
        continue
    a = t(lambda: plan.run_fast(Pd, Td)); torch.cuda.synchronize(); cnt = plan.retry_count(); b = t(lambda: plan.run_retry(Pd, Td))
    print(label, "fast %.3f ms  retry %.3f ms  (fallback rows, robust rows) =" % (a, b), cnt, " failed", int(plan.status.sum()))
