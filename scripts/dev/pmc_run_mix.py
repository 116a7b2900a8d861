"""scripts/dev/pmc_run_mix.py <variant> [n]: launch pcs_mix_bubble_dew (bubble, dew) of scratch/ab/lib_<variant>.so twice each
(for `rocprofv3 --kernel-trace --pmc ... -- python scripts/dev/pmc_run_mix.py <variant>`; summarise with pmc_sum.py <dir> k_mix_bubble_dew_queue)."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd.synthetic import mix_batch
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
P, K, T, X, PI = mix_batch(n)
d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
Pd, Kd, Td, Xd, PId = d(P), d(K), d(T), d(X), d(PI)
vp = ctypes.c_void_p
p = torch.empty(n, dtype=torch.float64, device="cuda"); rho4 = torch.empty((n, 4), dtype=torch.float64, device="cuda")
st = torch.empty(n, dtype=torch.uint8, device="cuda"); ws = torch.empty(n + 64, dtype=torch.int32, device="cuda")
L = ctypes.CDLL(os.path.abspath(f"scratch/ab/lib_{sys.argv[1]}.so"))
L.pcs_mix_bubble_dew.argtypes = [ctypes.c_int] + [vp] * 5 + [ctypes.c_int64] + [vp] * 6
for _ in range(2):
    for dew in (0, 1):
        assert L.pcs_mix_bubble_dew(dew, vp(Pd.data_ptr()), vp(Kd.data_ptr()), vp(Td.data_ptr()), vp(Xd.data_ptr()), vp(PId.data_ptr()), n, vp(p.data_ptr()), vp(rho4.data_ptr()), vp(st.data_ptr()), None, vp(ws.data_ptr()), None) == 0
torch.cuda.synchronize()
