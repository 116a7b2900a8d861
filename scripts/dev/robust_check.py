import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd.synthetic import pure_batch
from oracle import pyoracle as orc
n = 400_000
P, T = pure_batch(n)
Pd, Td = torch.from_numpy(P).cuda(), torch.from_numpy(T).cuda()
vp = ctypes.c_void_p
L = ctypes.CDLL(os.path.abspath("scratch/ab/lib_force.so"))
L.pcs_pure_vle.argtypes = [vp, vp, ctypes.c_int64] + [vp] * 7
L.pcs_workspace_bytes.restype = ctypes.c_int64; L.pcs_workspace_bytes.argtypes = [ctypes.c_int64]
p = torch.empty(n, dtype=torch.float64, device="cuda"); st = torch.empty(n, dtype=torch.uint8, device="cuda")
rho = torch.empty((n, 2), dtype=torch.float64, device="cuda")
ws = torch.empty(L.pcs_workspace_bytes(n) // 4, dtype=torch.int32, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
assert L.pcs_pure_vle(vp(Pd.data_ptr()), vp(Td.data_ptr()), n, vp(p.data_ptr()), None, vp(rho.data_ptr()), vp(st.data_ptr()), None, vp(ws.data_ptr()), vp(torch.cuda.current_stream().cuda_stream)) == 0
e1.record(); torch.cuda.synchronize()
print("robust pass over all", n, "rows:", e0.elapsed_time(e1), "ms")
ref, sref = orc.pure_vapor_pressure(P, T, prec=1)
got = p.cpu().numpy(); sg = st.cpu().numpy().astype(bool)
ok = ~sg & ~sref
err = np.abs(got[ok] / ref[ok] - 1)
print("fails", sg.sum(), "(oracle", sref.sum(), ") ; rel err quantiles 50/99/99.9/max", np.quantile(err, [.5, .99, .999, 1.0]))
bad = np.where(ok)[0][err > 1e-8]
print("rows with err > 1e-8:", len(bad))
tau = T / (1.28 * P[:, 2] * P[:, 0] ** 0.45)
r = rho.cpu().numpy()
for i in bad[:10]:
    print("  row", i, "tau %.3f" % tau[i], "got", got[i], "ref", ref[i], "rho_vl", r[i], "params", P[i])
f = np.where(sg & ~sref)[0]
print("robust-failed but oracle ok:", len(f), "tau", tau[f][:10])
for i in f[:5]: print("  row", i, "tau %.3f" % tau[i], "ref", ref[i], "params", P[i])
