"""Kernel vs oracle (two-attempt algorithm A) vs continuation on single rows of mix_batch(n, seed):
python tests/tools/mix_rows_diag.py <n> <seed> <dew> <row> [<row> ...]"""
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
rows = [int(a) for a in sys.argv[4:]]
P, K, T, X, PI = mix_batch(n, seed=seed)
P, K, T, X, PI = P[rows], K[rows], T[rows], X[rows], PI[rows]
d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
r = native.mix_bubble_dew(d(P), d(K), d(T), d(X), d(PI), dew, want_iters=True)
pA, rA, sA = orc.mix_bubble_dew(P, K, T, X, PI, dew, prec=1)
pC, rC, code, info = orc.mix_bubble_dew_continuation(P, K, T, X, dew, prec=0)
np.set_printoptions(linewidth=200, precision=6)
for k, row in enumerate(rows):
    g = r["rho4"][k].cpu().numpy()
    print(f"row {row}: kernel p {r['p'][k].item():.6e} st {int(r['status'][k])} it {int(r['iters'][k])} xL {g[2]/max(g[2:].sum(),1e-300):.4e} | oracle p {pA[k]:.6e} st {int(sA[k])} xL {rA[k,2]/max(rA[k,2:].sum(),1e-300):.4e}"
          f" | continuation p {pC[k]:.6e} code {code[k]} xL {rC[k,2]/max(rC[k,2:].sum(),1e-300):.4e}")
