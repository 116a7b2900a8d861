# the dew / bubble rows the solver gives up on although the continuation solver finds a solution (CPU: the oracle's double
# instantiation mirrors the kernels' decisions): python scripts/dev/missed_rows.py [dew|bubble] [rows]
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as oracle
from feos_torch_amd.synthetic import mix_batch
dew = (sys.argv[1] if len(sys.argv) > 1 else "dew") == "dew"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
P, K, T, X, PI = mix_batch(n)
t0 = time.time()
p, rho4, st = oracle.mix_bubble_dew(P, K, T, X, PI, dew, prec=0)
print("oracle (double mirror): %.1f s, failed %d" % (time.time() - t0, st.sum()))
idx = np.nonzero(st)[0]
pC, rC, code, info = oracle.mix_bubble_dew_continuation(P[idx], K[idx], T[idx], X[idx], dew, prec=0)
missed = idx[code == 0]
print("missed", len(missed))
np.save("scratch/missed_%s.npy" % ("dew" if dew else "bubble"), missed)
for k, i in enumerate(missed):
    j = np.nonzero(idx == i)[0][0]
    r = rC[j]
    na0, nb0, na1, nb1 = P[i, 0, 6], P[i, 0, 7], P[i, 1, 6], P[i, 1, 7]
    xl = r[2] / (r[2] + r[3]); yv = r[0] / (r[0] + r[1])
    print(f"{i:7d} T {T[i]:6.1f} z {X[i]:.3e} p_init {PI[i]:.2e} | cont p {pC[j]:.3e} x_L {xl:.3e} y_V {yv:.3e} rhoL {r[2]+r[3]:.3e} | m {P[i,0,0]:.2f}/{P[i,1,0]:.2f} eps {P[i,0,2]:.0f}/{P[i,1,2]:.0f} mu {P[i,0,3]:.1f}/{P[i,1,3]:.1f} assoc {na0:g}{nb0:g}/{na1:g}{nb1:g} eab {P[i,0,5]:.0f}/{P[i,1,5]:.0f} kij {K[i,0]:.3f}")
