"""Write profiles/<tag>_mix_missed.md: failed / missed / no-solution counts of the bubble / dew kernels on mix_batch(1e6) and of
the gc kernels on gc_batch(2.5e5), judged by the independent continuation solver.  Run on the GPU box:
  python tests/tools/mix_missed_table.py r02 > gpurun_out/r02_mix_missed.md"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from feos_torch_amd import native
from feos_torch_amd.synthetic import mix_batch
from oracle import pyoracle as orc

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000


def d(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


P, K, T, X, PI = mix_batch(n, seed=2026)
print(f"# Rows the mixture kernels give up on, judged by the independent continuation solver (`{tag}`)\n")
print(f"`mix_batch({n}, seed=2026)` (SURVEY 8d config 4), kernel = `pcs_mix_bubble_dew` (work queue + robust second pass); judge = "
      "`oracle/mix_continuation.hpp` in double precision on the failed rows.\n")
print("| problem | rows | kernel failed | of these: continuation finds a solution (MISSED) | curve ends in a critical point | stalled at a stability limit | no pure-fluid VLE | kernel ms |")
print("|---|---|---|---|---|---|---|---|")
for dew in (False, True):
    a = [d(v) for v in (P, K, T, X, PI)]
    native.mix_bubble_dew(*a, dew)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = native.mix_bubble_dew(*a, dew)
    e1.record()
    torch.cuda.synchronize()
    failed = r["status"].cpu().numpy().astype(bool)
    idx = np.nonzero(failed)[0]
    pC, rC, code, info = orc.mix_bubble_dew_continuation(P[idx], K[idx], T[idx], X[idx], dew, prec=0)
    print(f"| {'dew' if dew else 'bubble'} point | {n} | {len(idx)} | {(code == 0).sum()} | {(code == 2).sum()} | {(code == 3).sum()} | {(code == 1).sum()} | {e0.elapsed_time(e1):.2f} |")
print("\nRound 1 (no second pass) failed on 6,775 bubble / 3,074 dew rows of this batch.")
