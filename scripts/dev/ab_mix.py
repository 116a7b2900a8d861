# A/B of mixture bubble / dew variants: python scripts/dev/ab_mix.py <variant> ... (scratch/ab/lib_<variant>.so, scripts/dev/mkmix.sh)
# prints kernel times (HIP events around the whole call, 1e6 rows) and, against the first variant, the failure-mask and value differences
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd.synthetic import mix_batch
names = sys.argv[1:]
n = int(os.environ.get("AB_ROWS", 1_000_000))
P, K, T, X, PI = mix_batch(n)
d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
Pd, Kd, Td, Xd, PId = d(P), d(K), d(T), d(X), d(PI)
vp = ctypes.c_void_p
p = torch.empty(n, dtype=torch.float64, device="cuda"); rho4 = torch.empty((n,4), dtype=torch.float64, device="cuda")
st = torch.empty(n, dtype=torch.uint8, device="cuda")
ws = torch.empty(160 * n + 128, dtype=torch.int32, device="cuda")  # generous: covers every variant's pcs_mix_workspace_bytes
libs = {}
for nm in names:
    L = ctypes.CDLL(os.path.abspath(f"scratch/ab/lib_{nm}.so"))
    L.pcs_mix_bubble_dew.argtypes = [ctypes.c_int] + [vp]*5 + [ctypes.c_int64] + [vp]*6
    libs[nm] = L
def run(L, dew):
    assert L.pcs_mix_bubble_dew(dew, vp(Pd.data_ptr()), vp(Kd.data_ptr()), vp(Td.data_ptr()), vp(Xd.data_ptr()), vp(PId.data_ptr()), n, vp(p.data_ptr()), vp(rho4.data_ptr()), vp(st.data_ptr()), None, vp(ws.data_ptr()), None) == 0
res, out = {}, {}
for rnd in range(4):
    for nm in names:
        for dew in (0, 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(libs[nm], dew); e1.record(); torch.cuda.synchronize()
            if rnd: res.setdefault((nm, dew), []).append(e0.elapsed_time(e1))
            out[(nm, dew)] = (p.clone(), st.clone())
for nm in names:
    line = [nm, "bubble %.3f ms" % np.median(res[(nm,0)]), "dew %.3f ms" % np.median(res[(nm,1)])]
    for dew in (0, 1):
        pa, sa = out[(names[0], dew)]; pb, sb = out[(nm, dew)]
        both = (sa == 0) & (sb == 0)
        rel = ((pa - pb).abs() / pa.abs())[both]
        line.append(f"{'dew' if dew else 'bubble'}: fails {int(sb.sum())} mask-diff {int((sa != sb).sum())} max-rel {float(rel.max()):.2e} n>1e-9 {int((rel > 1e-9).sum())}")
    print("  ".join(line), flush=True)
