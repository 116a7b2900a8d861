"""Accuracy / time A/B of the pressure-only vapour-pressure kernel for several library builds (scratch/ab/lib_<name>.so)
against the long-double CPU oracle (test infrastructure, hence under tests/): max and quantile relative error of p_sat on
the first 1e6 rows of the BENCHMARK batch (pure_batch(1e7, seed 2026)) and on pure_batch(1e6, seed 77), plus the kernel time
on the full 1e7-row batch.  Usage on the GPU box:  python tests/tools/lite_accuracy_ab.py <name> [<name> ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import numpy as np
import torch

from feos_torch_amd.synthetic import pure_batch
from oracle import pyoracle as orc

names = sys.argv[1:]
vp = ctypes.c_void_p
n = 10_000_000
P, T = pure_batch(n, seed=2026)
P2, T2 = pure_batch(1_000_000, seed=77)
m = 1_000_000
want, st_o = orc.pure_vapor_pressure(P[:m], T[:m], prec=1)
want2, st_o2 = orc.pure_vapor_pressure(P2, T2, prec=1)
Pd, Td = torch.from_numpy(P).cuda(), torch.from_numpy(T).cuda()
P2d, T2d = torch.from_numpy(P2).cuda(), torch.from_numpy(T2).cuda()
for nm in names:
    L = ctypes.CDLL(os.path.abspath(f"scratch/ab/lib_{nm}.so"))
    for f in (L.pcs_pure_vle, L.pcs_pure_vle_fast):
        f.argtypes = [vp, vp, ctypes.c_int64] + [vp] * 7
    p = torch.empty(n, dtype=torch.float64, device="cuda")
    st = torch.empty(n, dtype=torch.uint8, device="cuda")
    ws = torch.empty(n + 64, dtype=torch.int32, device="cuda")
    stream = vp(torch.cuda.current_stream().cuda_stream)

    def run(Pq, Tq, k, fast=False):
        f = L.pcs_pure_vle_fast if fast else L.pcs_pure_vle
        assert f(vp(Pq.data_ptr()), vp(Tq.data_ptr()), k, vp(p.data_ptr()), None, None, vp(st.data_ptr()), None, vp(ws.data_ptr()), stream) == 0

    out = []
    for Pq, Tq, k, w, so in ((Pd, Td, m, want, st_o), (P2d, T2d, 1_000_000, want2, st_o2)):
        run(Pq, Tq, k)
        torch.cuda.synchronize()
        got, sg = p[:k].cpu().numpy(), st[:k].cpu().numpy().astype(bool)
        both = ~sg & ~so
        rel = np.abs(got[both] - w[both]) / np.abs(w[both])
        out.append(f"max {rel.max():.2e} q99.99 {np.quantile(rel, 0.9999):.2e} >1e-10: {(rel > 1e-10).sum()} gpu-failed {sg.sum()}")
    ts = []
    for r in range(14):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(Pd, Td, n, fast=True); e1.record(); torch.cuda.synchronize()
        if r >= 2:
            ts.append(e0.elapsed_time(e1))
    cnt = int(ws[0].item())
    print(f"{nm:10s} bench-batch[:1e6]: {out[0]} | seed77: {out[1]} | k_pure_vle+fallback 1e7 rows: median {np.median(ts):.3f} ms min {min(ts):.3f} list {cnt}", flush=True)
