"""scripts/dev/run_api.py [rows]: PcSaftPure.vapor_pressure through the Python drop-in, forward and forward + backward, on the
benchmark batch and on a batch with failed rows (near-critical temperatures) -- the command whose rocprofv3 kernel trace shows
what a property call launches (profiles/<tag>_api_kernel_stats.csv: no aten index / nonzero / masked kernels)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd import PcSaftPure
from feos_torch_amd.synthetic import pure_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
P, T = pure_batch(n, seed=2026)
Pd, Td = torch.from_numpy(P).cuda(), torch.from_numpy(T).cuda()
T2 = T.copy(); T2[::3] *= np.random.default_rng(1).uniform(1.0, 1.6, size=T2[::3].shape)  # every third row near / above critical
Td2 = torch.from_numpy(T2).cuda()
for Tq, tag in ((Td, "all rows converge"), (Td2, "rows dropped")):
    for _ in range(3):
        nans, p = PcSaftPure(Pd).vapor_pressure(Tq)
    Pg = Pd.clone().requires_grad_(True)
    for _ in range(3):
        Pg.grad = None
        nans, p = PcSaftPure(Pg).vapor_pressure(Tq)
        p.sum().backward()
    torch.cuda.synchronize()
    print(tag, "failed rows", int(nans.sum()))
