"""scripts/dev/cmp_mix.py <libA> <libB> [dew]: run pcs_mix_bubble_dew of two variant libraries on the same 1e6 rows and compare status / pressure row by row"""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd.synthetic import mix_batch
n = 1_000_000
dew = int(sys.argv[3]) if len(sys.argv) > 3 else 1
P, K, T, X, PI = mix_batch(n)
d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
Pd, Kd, Td, Xd, PId = d(P), d(K), d(T), d(X), d(PI)
vp = ctypes.c_void_p
out = {}
for nm in sys.argv[1:3]:
    L = ctypes.CDLL(os.path.abspath(f"scratch/ab/lib_{nm}.so"))
    L.pcs_mix_bubble_dew.argtypes = [ctypes.c_int] + [vp] * 5 + [ctypes.c_int64] + [vp] * 6
    p = torch.zeros(n, dtype=torch.float64, device="cuda"); rho4 = torch.zeros((n, 4), dtype=torch.float64, device="cuda")
    st = torch.zeros(n, dtype=torch.uint8, device="cuda"); ws = torch.zeros(n + 64, dtype=torch.int32, device="cuda"); it = torch.zeros(n, dtype=torch.int32, device="cuda")
    res = []
    for rep in range(2):
        assert L.pcs_mix_bubble_dew(dew, vp(Pd.data_ptr()), vp(Kd.data_ptr()), vp(Td.data_ptr()), vp(Xd.data_ptr()), vp(PId.data_ptr()), n, vp(p.data_ptr()), vp(rho4.data_ptr()), vp(st.data_ptr()), vp(it.data_ptr()), vp(ws.data_ptr()), None) == 0
        torch.cuda.synchronize()
        res.append((p.cpu().numpy().copy(), st.cpu().numpy().copy(), it.cpu().numpy().copy()))
    print(nm, "fails run1", int(res[0][1].sum()), "run2", int(res[1][1].sum()), "status differs between runs:", int((res[0][1] != res[1][1]).sum()),
          "p differs between runs:", int((res[0][0] != res[1][0]).sum()))
    out[nm] = res[1]
a, b = (out[k] for k in sys.argv[1:3])
only_a = (a[1] == 1) & (b[1] == 0); only_b = (a[1] == 0) & (b[1] == 1)
print("failed only in", sys.argv[1], int(only_a.sum()), "only in", sys.argv[2], int(only_b.sum()))
both = (a[1] == 0) & (b[1] == 0)
rel = np.abs(a[0][both] - b[0][both]) / np.abs(b[0][both])
print("both ok:", int(both.sum()), "max rel p diff", rel.max(), "rows > 1e-9:", int((rel > 1e-9).sum()))
idx = np.nonzero(only_a | only_b)[0][:12]
for i in idx:
    print(i, "status", a[1][i], b[1][i], "iters", a[2][i], b[2][i], "T", T[i], "z", X[i], "p", a[0][i], b[0][i])
