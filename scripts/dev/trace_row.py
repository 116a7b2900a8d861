# trace of the oracle's (double mirror of the kernels') bubble / dew solve on single rows of mix_batch:
# ORC_TRACE=1 python scripts/dev/trace_row.py dew <row> [<row> ...]
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as oracle
from feos_torch_amd.synthetic import mix_batch
dew = sys.argv[1] == "dew"
rows = [int(a) for a in sys.argv[2:]]
P, K, T, X, PI = mix_batch(1_000_000)
for i in rows:
    s = slice(i, i + 1)
    print(f"=== row {i} T {T[i]:.2f} z {X[i]:.4f}", flush=True)
    sys.stderr.flush()
    p, rho4, st = oracle.mix_bubble_dew(P[s].copy(), K[s].copy(), T[s].copy(), X[s].copy(), PI[s].copy(), dew, prec=0)
    pC, rC, code, info = oracle.mix_bubble_dew_continuation(P[s].copy(), K[s].copy(), T[s].copy(), X[s].copy(), dew, prec=0)
    print(f"   solver: status {st[0]} p {p[0]:.6e}   continuation: code {code[0]} p {pC[0]:.6e} rho4 {rC[0]}", flush=True)
