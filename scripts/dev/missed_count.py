# failure / missed counts of the oracle's double mirror on the 1e6-row batch (environment switches of oracle/mix_solver.hpp apply)
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as oracle
from feos_torch_amd.synthetic import mix_batch
dew = (sys.argv[1] if len(sys.argv) > 1 else "dew") == "dew"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
P, K, T, X, PI = mix_batch(n)
p, rho4, st = oracle.mix_bubble_dew(P, K, T, X, PI, dew, prec=0)
idx = np.nonzero(st)[0]
pC, rC, code, info = oracle.mix_bubble_dew_continuation(P[idx], K[idx], T[idx], X[idx], dew, prec=0)
tag = os.environ.get("TAG", "run")
np.savez(f"scratch/mc_{tag}.npz", p=p, st=st)
print(tag, "failed", int(st.sum()), "missed", int((code == 0).sum()), flush=True)
if os.path.exists("scratch/mc_base.npz") and tag != "base":
    b = np.load("scratch/mc_base.npz")
    both = ~b["st"] & ~st
    rel = np.abs(p[both] - b["p"][both]) / np.abs(b["p"][both])
    print("   vs base: mask diff", int((b["st"] != st).sum()), "newly failed", int((~b["st"] & st).sum()), "newly solved", int((b["st"] & ~st).sum()), "other root (>1e-8)", int((rel > 1e-8).sum()))
