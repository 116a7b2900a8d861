# per-row iteration counts of the fp32 pre-solve of k_pure_vle<true> (needs a diagnostic build that writes them to `iters`) against
# candidate predictors of the count: which row property would a difficulty order have to sort by?
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd import _lib, native
from feos_torch_amd.synthetic import pure_batch
_lib.LIB_PATH = os.path.abspath(f"scratch/ab/lib_{sys.argv[1]}.so"); _lib._lib = None
n = 2_000_000
P, T = pure_batch(n)
r = native.pure_vle(torch.from_numpy(P).cuda(), torch.from_numpy(T).cuda(), want_rho_vl=False, want_iters=True)
it = r["iters"].cpu().numpy()
nliq, ncpl, code, nfin = it & 0xff, (it >> 8) & 0xff, (it >> 16) & 0xff, (it >> 24) & 0xff
print("liquid-root evals: mean %.2f" % nliq.mean(), np.bincount(nliq)[:10])
print("coupled its: mean %.2f" % ncpl.mean(), np.bincount(ncpl)[:10])
print("fp64 finish its: mean %.2f" % nfin.mean(), np.bincount(nfin)[:6])
def wave_max(x, order=None, w=64):
    y = x if order is None else x[order]
    m = len(y) // w * w
    return y[:m].reshape(-1, w).max(axis=1).mean()
m, sig, eps, mu, kap, eab, na, nb = [P[:, k] for k in range(8)]
tstar = T / (eps * (1 + 0.3 * (m - 1) / m))
cls = (mu != 0) * 1 + ((na * nb) != 0) * 2
print("wave max (batch order): liq %.2f cpl %.2f fin %.2f" % (wave_max(nliq), wave_max(ncpl), wave_max(nfin)))
# within 256-row blocks: order by class (today), by class then T*, by the true count (ideal)
def block_order(keyfn):
    idx = np.arange(n).reshape(-1, 256)
    out = np.empty_like(idx)
    for b in range(idx.shape[0]):
        k = keyfn(idx[b]); out[b] = idx[b][np.argsort(k, kind="stable")]
    return out.reshape(-1)
nb_ = 2000 * 256
sub = slice(0, nb_)
for name, key in (("class", lambda i: cls[i]), ("class,T/eps", lambda i: cls[i] * 1e6 + T[i] / eps[i]), ("class,t*", lambda i: cls[i] * 1e6 + tstar[i]),
                  ("T/eps only", lambda i: T[i] / eps[i]), ("ideal (true coupled count)", lambda i: cls[i] * 100 + ncpl[i])):
    idx = np.arange(nb_).reshape(-1, 256); o = np.concatenate([b[np.argsort(key(b), kind="stable")] for b in idx])
    print("%-28s wave max: liq %.2f cpl %.2f" % (name, wave_max(nliq, o), wave_max(ncpl, o)))
# correlation of the coupled count with candidate predictors
for name, v in (("T/eps", T / eps), ("t*", tstar), ("m", m), ("eab/T", eab / T), ("mu", mu)):
    print("corr(ncpl, %s) = %.3f   corr(nliq, %s) = %.3f" % (name, np.corrcoef(ncpl, v)[0, 1], name, np.corrcoef(nliq, v)[0, 1]))
