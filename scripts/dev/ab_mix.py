import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd.synthetic import mix_batch
names = sys.argv[1:]
n = 1_000_000
P, K, T, X, PI = mix_batch(n)
d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
Pd, Kd, Td, Xd, PId = d(P), d(K), d(T), d(X), d(PI)
vp = ctypes.c_void_p
p = torch.empty(n, dtype=torch.float64, device="cuda"); rho4 = torch.empty((n,4), dtype=torch.float64, device="cuda")
st = torch.empty(n, dtype=torch.uint8, device="cuda"); ws = torch.empty(n+64, dtype=torch.int32, device="cuda")  # row order + control block (pcs_workspace_bytes)
libs = {}
for nm in names:
    L = ctypes.CDLL(os.path.abspath(f"scratch/ab/lib_{nm}.so"))
    L.pcs_mix_bubble_dew.argtypes = [ctypes.c_int] + [vp]*5 + [ctypes.c_int64] + [vp]*6
    libs[nm] = L
def run(L, dew):
    assert L.pcs_mix_bubble_dew(dew, vp(Pd.data_ptr()), vp(Kd.data_ptr()), vp(Td.data_ptr()), vp(Xd.data_ptr()), vp(PId.data_ptr()), n, vp(p.data_ptr()), vp(rho4.data_ptr()), vp(st.data_ptr()), None, vp(ws.data_ptr()), None) == 0
res = {}
for rnd in range(4):
    for nm in names:
        for dew in (0, 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(libs[nm], dew); e1.record(); torch.cuda.synchronize()
            if rnd: res.setdefault((nm, dew), []).append(e0.elapsed_time(e1))
for nm in names:
    print(nm, "bubble %.2f ms" % np.median(res[(nm,0)]), "dew %.2f ms" % np.median(res[(nm,1)]), "fails", int(st.sum()))
