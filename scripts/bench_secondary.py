#!/usr/bin/env python3
"""Secondary configs of BASELINE.json (not the headline line of bench.py): kernel times measured
with HIP events on the launch stream, inputs resident in HBM.
  config 3: PcSaftPure liquid_density + equilibrium_liquid_density, batch 1e7
  config 4: PcSaftMix bubble / dew point, batch 1e6
  config 5: GcPcSaftMix bubble / dew point, batch 1e6
Prints one JSON object per config."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from feos_torch_amd import native
from feos_torch_amd.gc_pcsaft import build_table, encode_rows
from feos_torch_amd.synthetic import gc_batch, load_segment_table, mix_batch, pure_batch, pure_pressures

d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), out


which = sys.argv[1:] or ["3", "4", "5"]
if "3" in which:
    n = 10_000_000
    P, T = pure_batch(n); pp = pure_pressures(n)
    Pd, Td, pd = d(P), d(T), d(pp)
    ms, r = timed(lambda: native.pure_liquid_density(Pd, Td, pd))
    ms2, r2 = timed(lambda: native.pure_vle(Pd, Td, want_p=False, want_rho_eq=True))
    print(json.dumps({"config": "PcSaftPure liquid_density batch=1e7", "ms": ms, "rows_per_s": n / ms * 1e3, "failed": int(r["status"].sum())}))
    print(json.dumps({"config": "PcSaftPure equilibrium_liquid_density batch=1e7", "ms": ms2, "rows_per_s": n / ms2 * 1e3, "failed": int(r2["status"].sum())}))
if "4" in which:
    n = 1_000_000
    P, K, T, X, PI = mix_batch(n)
    a = [d(v) for v in (P, K, T, X, PI)]
    for dew in (False, True):
        ms, r = timed(lambda: native.mix_bubble_dew(*a, dew), reps=3)
        print(json.dumps({"config": f"PcSaftMix {'dew' if dew else 'bubble'} point batch=1e6", "ms": ms, "rows_per_s": n / ms * 1e3, "failed": int(r["status"].sum())}))
    r = native.mix_bubble_dew(*a, False)
    ms, _ = timed(lambda: native.mix_jacobian(a[0], a[1], a[2], r["rho4"], False), reps=3)
    print(json.dumps({"config": "PcSaftMix bubble Jacobian batch=1e6", "ms": ms, "rows_per_s": n / ms * 1e3}))
if "5" in which:
    n = 1_000_000
    table = load_segment_table(os.path.join(ROOT, "tests", "data", "sauer2014_hetero.json"))
    b = gc_batch(n, table); ident = [s for s, _ in table]
    from feos_torch_amd.gc_pcsaft import encode_rows_device
    encode_rows_device(ident, b["segment_lists"][:1000], b["bond_lists"][:1000], "cuda")  # warm (allocator, kernels)
    torch.cuda.synchronize()
    t0 = time.time(); rows = encode_rows_device(ident, b["segment_lists"], b["bond_lists"], "cuda"); torch.cuda.synchronize()
    t_enc = time.time() - t0  # host encoding of the distinct molecules + row indices, H2D of 8 B per row, row assembly on the GPU
    seg = torch.tensor(np.stack([v for _, v in table]), dtype=torch.float64)
    kab = torch.zeros((len(ident), len(ident)), dtype=torch.float64)
    for s1, s2, k in b["kab_list"]:
        kab[ident.index(s1), ident.index(s2)] = k; kab[ident.index(s2), ident.index(s1)] = k
    tab = build_table(seg.cuda(), kab.cuda())
    phi, T, x, p0 = d(b["phi"]), d(b["T"]), d(b["x"]), d(b["p_init"])
    order = native.gc_class_order(tab, len(ident), rows)  # once per model, as GcPcSaftMix does
    for dew in (False, True):
        ms, r = timed(lambda: native.gc_bubble_dew(tab, len(ident), rows, phi, T, x, p0, dew, order=order), reps=3)
        print(json.dumps({"config": f"GcPcSaftMix {'dew' if dew else 'bubble'} point batch=1e6", "ms": ms, "rows_per_s": n / ms * 1e3, "failed": int(r["status"].sum()), "host_encode_s": t_enc}))
