import ctypes, sys, os, glob
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd.synthetic import pure_batch
n = 10_000_000
P, T = pure_batch(n)
Pd, Td = torch.from_numpy(P).cuda(), torch.from_numpy(T).cuda()
vp = ctypes.c_void_p
names = sys.argv[1:]
p = torch.empty(n, dtype=torch.float64, device="cuda"); st = torch.empty(n, dtype=torch.uint8, device="cuda")
ws = torch.empty(n + 64, dtype=torch.int32, device="cuda")
libs = {}
for nm in names:
    L = ctypes.CDLL(os.path.abspath(f"scratch/ab/lib_{nm}.so")); L.pcs_pure_vle_fast.argtypes = [vp, vp, ctypes.c_int64] + [vp] * 7; libs[nm] = L
times = {nm: [] for nm in names}
for rnd in range(8):
    for nm in names:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        libs[nm].pcs_pure_vle_fast(vp(Pd.data_ptr()), vp(Td.data_ptr()), n, vp(p.data_ptr()), None, None, vp(st.data_ptr()), None, vp(ws.data_ptr()), vp(torch.cuda.current_stream().cuda_stream))
        e1.record(); torch.cuda.synchronize()
        if rnd >= 2: times[nm].append(e0.elapsed_time(e1))
prev = 0
for nm in names:
    t = np.median(times[nm]); print(f"{nm:8s} {t:.3f} ms   (+{t-prev:.3f})"); prev = t
