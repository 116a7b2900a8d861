"""A/B several builds of libpcsaft_hip (scratch/ab/lib_<name>.so) in ONE process, interleaved rounds.
Timing only: accuracy is the business of tests/ (the oracle is not imported outside tests/, smoke() and bench.py)."""
import ctypes, sys, glob, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd.synthetic import pure_batch
names = sys.argv[1:] or sorted(os.path.basename(f)[4:-3] for f in glob.glob("scratch/ab/lib_*.so"))
n = 10_000_000
P, T = pure_batch(n)
Pd, Td = torch.from_numpy(P).cuda(), torch.from_numpy(T).cuda()
vp = ctypes.c_void_p
libs = {}
for nm in names:
    L = ctypes.CDLL(os.path.abspath(f"scratch/ab/lib_{nm}.so"))
    L.pcs_pure_vle_fast.argtypes = [vp, vp, ctypes.c_int64] + [vp] * 7
    L.pcs_pure_vle_retry.argtypes = [vp, vp, ctypes.c_int64] + [vp] * 7
    libs[nm] = L
p = torch.empty(n, dtype=torch.float64, device="cuda"); st = torch.empty(n, dtype=torch.uint8, device="cuda")
ws = torch.empty(n + 64, dtype=torch.int32, device="cuda")
# PCS_AB_FULL=1: the all-fp64 kernel (densities returned) instead of the pressure-only one
rho_vl = torch.empty((n, 2), dtype=torch.float64, device="cuda") if os.environ.get("PCS_AB_FULL") else None
stream = vp(torch.cuda.current_stream().cuda_stream)
def run(L, retry=False):
    args = (vp(Pd.data_ptr()), vp(Td.data_ptr()), n, vp(p.data_ptr()), None, vp(rho_vl.data_ptr()) if rho_vl is not None else None, vp(st.data_ptr()), None, vp(ws.data_ptr()), stream)
    assert L.pcs_pure_vle_fast(*args) == 0
    if retry: assert L.pcs_pure_vle_retry(*args) == 0
times = {nm: [] for nm in names}; rtimes = {nm: [] for nm in names}
for nm in names:
    run(libs[nm], True); torch.cuda.synchronize()
    print(f"{nm:10s} fails {int(st.sum())} list entries {int(ws[0].item())}")
for rnd in range(12):
    for nm in names:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e2 = torch.cuda.Event(enable_timing=True)
        e0.record(); run(libs[nm]); e1.record()
        args = (vp(Pd.data_ptr()), vp(Td.data_ptr()), n, vp(p.data_ptr()), None, vp(rho_vl.data_ptr()) if rho_vl is not None else None, vp(st.data_ptr()), None, vp(ws.data_ptr()), stream)
        libs[nm].pcs_pure_vle_retry(*args); e2.record(); torch.cuda.synchronize()
        if rnd >= 2: times[nm].append(e0.elapsed_time(e1)); rtimes[nm].append(e1.elapsed_time(e2))
base = np.median(times[names[0]])
for nm in names:
    t = np.array(times[nm]); print(f"{nm:10s} k_pure_vle median {np.median(t):.3f} ms  min {t.min():.3f}  -> {n/np.median(t)/1e6*1e3/1e3:.3f} Gsolves/s   x{base/np.median(t):.3f}   retry pass median {np.median(rtimes[nm]):.3f} ms")
