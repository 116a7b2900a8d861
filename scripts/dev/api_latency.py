# End-to-end latency of the Python API (forward, forward + backward) at several batch sizes, inputs resident on the GPU
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
import feos_torch_amd as amd
from feos_torch_amd.synthetic import pure_batch, mix_batch
d = lambda x, g=False: torch.from_numpy(np.ascontiguousarray(x)).cuda().requires_grad_(g)
def t(fn, reps=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
for n in (1_000, 10_000, 100_000, 1_000_000, 10_000_000):
    P, T = pure_batch(n)
    Pd, Td = d(P), d(T)
    fwd = t(lambda: amd.PcSaftPure(Pd).vapor_pressure(Td))
    Pg = d(P, True)
    def fb():
        nans, p = amd.PcSaftPure(Pg).vapor_pressure(Td)
        p.sum().backward(); Pg.grad = None
    print(f"PcSaftPure.vapor_pressure n={n:>9}: forward {fwd:8.3f} ms   forward+backward {t(fb):8.3f} ms", flush=True)
for n in (1_000, 10_000, 100_000, 1_000_000):
    P, K, T, X, PI = mix_batch(n)
    a = [d(v) for v in (P, K, T, X, PI)]
    for dew in (False, True):
        def f():
            eos = amd.PcSaftMix(a[0], a[1])
            return (eos.dew_point if dew else eos.bubble_point)(a[2], a[3], a[4])
        fwd = t(f)
        Kg = d(K, True); Pg = d(P, True)
        def fb():
            eos = amd.PcSaftMix(Pg, Kg)
            p, nans = (eos.dew_point if dew else eos.bubble_point)(a[2], a[3], a[4])
            p.sum().backward(); Kg.grad = None; Pg.grad = None
        print(f"PcSaftMix.{'dew' if dew else 'bubble'}_point n={n:>9}: forward {fwd:8.3f} ms   forward+backward {t(fb):8.3f} ms", flush=True)
