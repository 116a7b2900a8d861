"""Classify the rows the first bubble / dew algorithm (oracle/mix_solver.hpp == the kernels' solver) gives up on with the
SECOND, independent solver (oracle/mix_continuation.hpp): `missed` = the continuation finds a solution, `no solution found`
= it does not either (curve ends in a critical point / stalls at a stability limit / no pure-fluid VLE at T).
  python tests/tools/mix_classify.py [rows] [seed] [--all]     (--all: also run the continuation on the rows A solves)
Runs on the CPU (oracle only); tests/test_mix_missed_gpu.py does the same for the kernel's own failure mask."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from feos_torch_amd.synthetic import mix_batch
from oracle import pyoracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 78
P, K, T, X, PI = mix_batch(n, seed=seed)
for dew in (False, True):
    t = time.time()
    pA, rA, sA = orc.mix_bubble_dew(P, K, T, X, PI, dew, prec=0)
    tA = time.time() - t
    idx = np.nonzero(sA)[0] if "--all" not in sys.argv else np.arange(n)
    t = time.time()
    pC, rC, code, info = orc.mix_bubble_dew_continuation(P[idx], K[idx], T[idx], X[idx], dew, prec=0)
    tC = time.time() - t
    f = sA[idx]
    print(f"{'dew' if dew else 'bubble'}: {n} rows, algorithm A failed on {sA.sum()} ({tA:.1f} s); continuation on {len(idx)} rows ({tC:.1f} s)")
    print(f"   A-failed rows: missed (continuation finds a solution) {(f & (code == 0)).sum()}, no pure-fluid VLE {(f & (code == 1)).sum()}, "
          f"critical end {(f & (code == 2)).sum()}, stalled {(f & (code == 3)).sum()}")
    if "--all" in sys.argv:
        both = ~sA & (code == 0)
        rel = np.abs(pA[both] - pC[both]) / np.abs(pA[both])
        print(f"   both solve {both.sum()}: same solution (1e-8) {(rel < 1e-8).sum()}, different solutions {(rel >= 1e-8).sum()}; A solves, continuation does not: {(~sA & (code != 0)).sum()}")
    if "--rows" in sys.argv:
        print("   missed rows:", idx[f & (code == 0)][:40].tolist())
