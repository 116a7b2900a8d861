import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd.synthetic import pure_batch
n = 2_000_000
P, T = pure_batch(n)
cls = (P[:, 3] > 0).astype(int) + 2 * (P[:, 4] > 0).astype(int)
Pd, Td = torch.from_numpy(P).cuda(), torch.from_numpy(T).cuda()
vp = ctypes.c_void_p
L = ctypes.CDLL(os.path.abspath("scratch/ab/lib_diag.so"))
L.pcs_pure_vle_fast.argtypes = [vp, vp, ctypes.c_int64] + [vp] * 7
p = torch.empty(n, dtype=torch.float64, device="cuda"); st = torch.empty(n, dtype=torch.uint8, device="cuda")
it = torch.zeros(n, dtype=torch.int32, device="cuda")
ws = torch.empty(n + 64, dtype=torch.int32, device="cuda")
L.pcs_pure_vle_fast(vp(Pd.data_ptr()), vp(Td.data_ptr()), n, vp(p.data_ptr()), None, None, vp(st.data_ptr()), vp(it.data_ptr()), vp(ws.data_ptr()), vp(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
it = it.cpu().numpy()
f64, liq, cpl = it & 255, (it >> 8) & 255, (it >> 16) & 255
for name, a in (("fp32 liquid evals", liq), ("fp32 coupled its", cpl), ("fp64 its", f64)):
    w = a[: n // 64 * 64].reshape(-1, 64).max(1)
    print(f"{name:18s} lane mean {a.mean():.2f} hist {np.bincount(a, minlength=13)[:13]}  | wave-max mean {w.mean():.2f} hist {np.bincount(w, minlength=13)[:13]}")
    for c in range(4):
        print(f"      class {c}: lane mean {a[cls == c].mean():.2f}  hist {np.bincount(a[cls==c], minlength=13)[:13]}")
tau = T / (P[:, 2] * 1.28 * P[:, 0] ** 0.45)
for lo in np.arange(0.55, 0.9, 0.05):
    m = (tau >= lo) & (tau < lo + 0.05)
    print(f"tau [{lo:.2f},{lo+0.05:.2f}) liq {liq[m].mean():.2f} cpl {cpl[m].mean():.2f} f64 {f64[m].mean():.2f}")
code = (it >> 24) & 255
print("fail codes:", {int(k): int(v) for k, v in zip(*np.unique(code, return_counts=True))})
for cd in np.unique(code):
    if cd == 0: continue
    m = code == cd
    print(f"code {cd}: n {m.sum()} classes {np.bincount(cls[m], minlength=4)} tau mean {tau[m].mean():.3f} min {tau[m].min():.3f} max {tau[m].max():.3f}; liq {liq[m].mean():.2f} cpl {cpl[m].mean():.2f} f64 {f64[m].mean():.2f};  p_sat median {np.median(p[m].cpu().numpy()):.3e}")
