"""Print the rows of mix_batch(n, seed) the kernel fails on (dew or bubble), with the verdict of the oracle's two-attempt
algorithm A and of the continuation solver.  python tests/tools/mix_failed_rows.py <n> <seed> <dew 0|1>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from feos_torch_amd import native
from feos_torch_amd.synthetic import mix_batch
from oracle import pyoracle as orc

n, seed, dew = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
P, K, T, X, PI = mix_batch(n, seed=seed)
d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
r = native.mix_bubble_dew(d(P), d(K), d(T), d(X), d(PI), dew, want_iters=True)
failed = r["status"].cpu().numpy().astype(bool)
idx = np.nonzero(failed)[0]
pA, rA, sA = orc.mix_bubble_dew(P[idx], K[idx], T[idx], X[idx], PI[idx], dew, prec=0)
pC, rC, code, info = orc.mix_bubble_dew_continuation(P[idx], K[idx], T[idx], X[idx], dew, prec=0)
print("kernel failed", len(idx), "| oracle A (two attempts) also fails", int(sA.sum()), "| continuation solves", int((code == 0).sum()))
print("rows kernel fails, oracle A solves:", idx[~sA].tolist()[:60])
print("rows kernel fails, oracle A fails, continuation solves:", idx[sA & (code == 0)].tolist()[:60])
if len(sys.argv) > 4:  # optional: save the verdicts for an offline look (CPU, oracle only)
    np.savez(sys.argv[4], idx=idx, oracle_failed=sA, cont_code=code, p_cont=pC, rho_cont=rC)
