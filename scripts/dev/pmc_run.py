"""scripts/dev/pmc_run.py <variant> [n]: launch pcs_pure_vle_fast of scratch/ab/lib_<variant>.so 4 times on the bench batch
(for `rocprofv3 --kernel-trace --pmc ... -- python scripts/dev/pmc_run.py <variant>`; summarise with pmc_sum.py)."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from feos_torch_amd.synthetic import pure_batch
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
P, T = pure_batch(n)
Pd, Td = torch.from_numpy(P).cuda(), torch.from_numpy(T).cuda()
vp = ctypes.c_void_p
L = ctypes.CDLL(os.path.abspath(f"scratch/ab/lib_{sys.argv[1]}.so"))
L.pcs_pure_vle_fast.argtypes = [vp, vp, ctypes.c_int64] + [vp] * 7
p = torch.empty(n, dtype=torch.float64, device="cuda"); st = torch.empty(n, dtype=torch.uint8, device="cuda")
ws = torch.empty(n + 64, dtype=torch.int32, device="cuda")
for _ in range(4):
    assert L.pcs_pure_vle_fast(vp(Pd.data_ptr()), vp(Td.data_ptr()), n, vp(p.data_ptr()), None, None, vp(st.data_ptr()), None, vp(ws.data_ptr()), vp(torch.cuda.current_stream().cuda_stream)) == 0
torch.cuda.synchronize()
