"""Mixture / gc parts of make_golden.py (run through that script; see its docstring)."""
import os
import sys

import numpy as np
import torch

f64 = torch.float64

# tests/test_pcsaft_mix.py:17-39
MIX_TEST_PARAMS = [
    [[1.5, 3.2, 150, 0, 0, 0, 0, 0], [2.5, 3.5, 250, 0, 0, 0, 0, 0]],
    [[1.5, 3.2, 150, 2.5, 0, 0, 0, 0], [2.5, 3.5, 250, 0, 0, 0, 0, 0]],
    [[1.5, 3.2, 150, 0, 0, 0, 0, 0], [2.5, 3.5, 250, 2, 0, 0, 0, 0]],
    [[1.5, 3.2, 150, 2.5, 0, 0, 0, 0], [2.5, 3.5, 250, 2, 0, 0, 0, 0]],
    [[1.5, 3.2, 150, 0, 0.03, 2500, 2, 1], [2.5, 3.5, 250, 0, 0, 0, 0, 0]],
    [[1.5, 3.2, 150, 0, 0, 0, 0, 0], [2.5, 3.5, 250, 0, 0.025, 1500, 1, 2]],
    [[1.5, 3.2, 150, 0, 0.03, 2500, 1, 1], [2.5, 3.5, 250, 0, 0.025, 1500, 1, 1]],
    [[1.5, 3.2, 150, 2.5, 0.03, 2500, 1, 1], [2.5, 3.5, 250, 2, 0.025, 1500, 1, 1]],
    [[1.5, 3.2, 150, 0, 0.03, 2500, 1, 1], [2.5, 3.5, 250, 0, 0.025, 1500, 0, 1]],
    [[1.5, 3.2, 150, 0, 0.03, -500, 0, 2], [2.5, 3.5, 250, 0, 0.025, 1500, 1, 1]],
    [[1.5, 3.2, 150, 0, 0, 0, 0, 0], [2.5, 3.5, 250, 0, 0.025, 1500, 0, 1]],
    [[1.5, 3.2, 150, 0, 0.03, 2500, 2, 2], [2.5, 3.5, 250, 0, 0.025, 1500, 1, 1]],
    [[1.5, 3.2, 150, 0, 0.03, 2500, 2, 2], [2.5, 3.5, 250, 0, 0.025, 1500, 1, 1]],
    [[1.5, 3.2, 150, 0, 0.03, 2500, 1, 2], [2.5, 3.5, 250, 0, 0.025, 1500, 2, 1]],
]


def _bubble_dew(PcSaftMix, tl, params, kij, T, z, p, dew):
    x = torch.tensor(params, dtype=f64, requires_grad=True)
    k = torch.tensor(kij, dtype=f64, requires_grad=True)
    Tt = torch.tensor(T, dtype=f64, requires_grad=True)
    zt = torch.tensor(z, dtype=f64)
    pt = torch.tensor(p, dtype=f64)
    eos = PcSaftMix(x, k)
    val, nans = (eos.dew_point if dew else eos.bubble_point)(Tt, zt, pt)
    val.sum().backward()
    return {"nans": tl(nans), "value": tl(val), "grad_params": tl(x.grad), "grad_kij": tl(k.grad), "grad_T": tl(Tt.grad)}


def make_mix(PcSaftMix, dump, tl):
    g = {}
    n = len(MIX_TEST_PARAMS)
    kij = [[-0.05, 0.0] for _ in range(n)]
    kij[12][1] = 3000.0
    T = [300.0] * n
    rho = [[0.001, 0.002]] * n
    eos = PcSaftMix(torch.tensor(MIX_TEST_PARAMS, dtype=f64), torch.tensor(kij, dtype=f64))
    a, p, mu, v = eos.derivatives(torch.tensor(T, dtype=f64), torch.tensor(rho, dtype=f64))
    g["test_inputs"] = {"params": MIX_TEST_PARAMS, "kij": kij, "T": T, "rho": rho, "a": tl(a), "p": tl(p), "mu": tl(mu), "v": tl(v)}

    # tests/test_pcsaft_mix.py:127-192 (bubble) and :195-251 (dew): two rows differing by h in k_ij
    h = 1e-8
    pb = [[[1, 3.5, 150, 0, 0.02, 1500, 1, 1], [1, 3.5, 200, 0, 0.03, 2500, 1, 1]]] * 2
    kb = [[-0.15, 1000.0], [-0.15 + h, 1000.0]]
    g["test_bubble"] = {"params": pb, "kij": kb, "T": [150.0] * 2, "z": [0.5] * 2, "p_init": [1e5] * 2, "h": h}
    g["test_bubble"]["result"] = _bubble_dew(PcSaftMix, tl, pb, kb, [150.0] * 2, [0.5] * 2, [1e5] * 2, False)
    pd = [[[1, 3.5, 150, 0, 0, 0, 0, 0], [1, 3.5, 200, 0, 0, 0, 0, 0]]] * 2
    kd = [[-0.15, 0.0], [-0.15 + h, 0.0]]
    g["test_dew"] = {"params": pd, "kij": kd, "T": [150.0] * 2, "z": [0.5] * 2, "p_init": [1e5] * 2, "h": h}
    g["test_dew"]["result"] = _bubble_dew(PcSaftMix, tl, pd, kd, [150.0] * 2, [0.5] * 2, [1e5] * 2, True)

    # seeded random rows of the config-4 distribution
    from feos_torch_amd.synthetic import mix_batch
    P, K, TT, X, PI = mix_batch(60, seed=17)
    rng = np.random.default_rng(19)
    eta = np.where(rng.random(60) < 0.5, rng.uniform(0.2, 0.42, 60), 10.0 ** rng.uniform(-7, -2, 60))
    d = P[:, :, 1] * (1 - 0.12 * np.exp(-3 * P[:, :, 2] / TT[:, None]))
    xx = np.stack([X, 1 - X], axis=1)
    rho_tot = eta / (np.pi / 6 * (xx * P[:, :, 0] * d**3).sum(axis=1))
    rho = xx * rho_tot[:, None]
    eos = PcSaftMix(torch.tensor(P, dtype=f64), torch.tensor(K, dtype=f64))
    a, p, mu, v = eos.derivatives(torch.tensor(TT, dtype=f64), torch.tensor(rho, dtype=f64))
    g["random"] = {"params": P.tolist(), "kij": K.tolist(), "T": TT.tolist(), "z": X.tolist(), "p_init": PI.tolist(),
                   "rho": rho.tolist(), "a": tl(a), "p": tl(p), "mu": tl(mu), "v": tl(v)}
    g["random"]["bubble"] = _bubble_dew(PcSaftMix, tl, P.tolist(), K.tolist(), TT.tolist(), X.tolist(), PI.tolist(), False)
    g["random"]["dew"] = _bubble_dew(PcSaftMix, tl, P.tolist(), K.tolist(), TT.tolist(), X.tolist(), PI.tolist(), True)
    dump("mix.json", g)


# tests/test_gc_pcsaft.py:17-44
GC_TEST_SEGMENTS = [
    [["CH3", "CH2", "CH2", "CH3"], ["CH3", "CH2", "CH3"]],
    [["CH3", ">CH", "CH3", "CH3"], ["CH3", ">C<", "CH3", "CH3", "CH3"]],
    [["CH3", ">CH", "CH3", "CH=O"], ["CH3", ">C<", "CH3", "CH3", "CH3"]],
    [["CH3", ">CH", "CH3", "CH3"], ["CH3", ">C<", "CH3", "CH3", "HCOO"]],
    [["CH3", ">CH", "CH3", "CH=O"], ["CH3", ">C<", "CH3", "CH3", "HCOO"]],
    [["CH3", ">CH", "CH3", "OH"], ["CH3", ">C<", "CH3", "CH3", "CH3"]],
    [["CH3", ">CH", "CH3", "CH3"], ["CH3", ">C<", "CH3", "CH3", "NH2"]],
    [["CH3", ">CH", "CH3", "OH"], ["CH3", ">C<", "CH3", "CH3", "NH2"]],
    [["CH3", ">CH", "CH=O", "OH"], ["CH3", ">C<", "CH3", "HCOO", "NH2"]],
    [["CH3", ">CH", "CH=O", "OH"], ["CH3", ">C<", "CH3", "CH2", "IA"]],
    [["CH3", ">CH", "CH=O", "IA"], ["CH3", ">C<", "CH3", "CH2", "IA"]],
]
GC_TEST_BONDS = [[[[0, 1], [1, 2], [2, 3]], [[0, 1], [1, 2]]]] + [[[[0, 1], [1, 2], [1, 3]], [[0, 1], [1, 2], [1, 3], [1, 4]]]] * 10
GC_TEST_KAB = [("CH3", "CH=O", 0.03), (">CH", "HCOO", -0.01)]


def _parse_segments(table):
    ident = [s for s, _ in table]
    cols = [torch.tensor([v[k] for _, v in table], dtype=f64) for k in range(8)]
    return ident, tuple(cols)


def _gc_bubble_dew(GcPcSaftMix, tl, table, seg, bon, kab_pairs, kab_vals, phi, T, z, p, dew):
    kab = torch.tensor(kab_vals, dtype=f64, requires_grad=True)
    kab_list = [(s1, s2, k) for (s1, s2), k in zip(kab_pairs, kab)]
    ph = torch.tensor(phi, dtype=f64, requires_grad=True)
    Tt = torch.tensor(T, dtype=f64, requires_grad=True)
    eos = GcPcSaftMix(*_parse_segments(table), seg, bon, kab_list, ph)
    val, nans = (eos.dew_point if dew else eos.bubble_point)(Tt, torch.tensor(z, dtype=f64), torch.tensor(p, dtype=f64))
    val.sum().backward()
    return {"nans": tl(nans), "value": tl(val), "grad_kab": tl(kab.grad), "grad_phi": tl(ph.grad), "grad_T": tl(Tt.grad)}


def make_gc(GcPcSaftMix, dump, tl):
    import os
    from feos_torch_amd.synthetic import gc_batch, load_segment_table
    table = load_segment_table(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "sauer2014_hetero.json"))
    g = {}
    n = len(GC_TEST_SEGMENTS)
    phi = [[1.1, 0.98]] * n
    T = [300.0] * n
    rho = [[0.001, 0.002]] * n
    eos = GcPcSaftMix(*_parse_segments(table), GC_TEST_SEGMENTS, GC_TEST_BONDS, GC_TEST_KAB, torch.tensor(phi, dtype=f64))
    a, p, mu, v = eos.derivatives(torch.tensor(T, dtype=f64), torch.tensor(rho, dtype=f64))
    g["test_inputs"] = {"segment_lists": GC_TEST_SEGMENTS, "bond_lists": GC_TEST_BONDS, "kab_list": [list(k) for k in GC_TEST_KAB],
                        "phi": phi, "T": T, "rho": rho, "a": tl(a), "p": tl(p), "mu": tl(mu), "v": tl(v)}
    # tests/test_gc_pcsaft.py:130-222: n-butane / propane at 150 K, x = 0.5, k_ab(CH3, CH2) = -0.15
    seg1, bon1 = GC_TEST_SEGMENTS[:1], GC_TEST_BONDS[:1]
    for key, dew in (("test_bubble", False), ("test_dew", True)):
        g[key] = {"segment_lists": seg1, "bond_lists": bon1, "kab_pairs": [["CH3", "CH2"]], "kab_vals": [-0.15], "phi": [[1.1, 0.98]],
                  "T": [150.0], "z": [0.5], "p_init": [1e5]}
        g[key]["result"] = _gc_bubble_dew(GcPcSaftMix, tl, table, seg1, bon1, [["CH3", "CH2"]], [-0.15], [[1.1, 0.98]], [150.0], [0.5], [1e5], dew)
        hres = _gc_bubble_dew(GcPcSaftMix, tl, table, seg1, bon1, [["CH3", "CH2"]], [-0.15 + 1e-7], [[1.1, 0.98]], [150.0], [0.5], [1e5], dew)
        g[key]["value_kab_plus_1e-7"] = hres["value"]
    # seeded random rows of the config-5 distribution
    b = gc_batch(48, table, seed=31)
    pairs = [[k[0], k[1]] for k in b["kab_list"]]
    vals = [k[2] for k in b["kab_list"]]
    rng = np.random.default_rng(37)
    eos = GcPcSaftMix(*_parse_segments(table), b["segment_lists"], b["bond_lists"], b["kab_list"], torch.tensor(b["phi"], dtype=f64))
    rho = np.stack([b["x"], 1 - b["x"]], axis=1) * (10.0 ** rng.uniform(-6, -2.2, 48))[:, None]
    a, p, mu, v = eos.derivatives(torch.tensor(b["T"], dtype=f64), torch.tensor(rho, dtype=f64))
    g["random"] = {"seed": 31, "n": 48, "kab_pairs": pairs, "kab_vals": vals, "rho": rho.tolist(), "a": tl(a), "p": tl(p), "mu": tl(mu), "v": tl(v)}
    for name, dew in (("bubble", False), ("dew", True)):
        g["random"][name] = _gc_bubble_dew(GcPcSaftMix, tl, table, b["segment_lists"], b["bond_lists"], pairs, vals, b["phi"].tolist(),
                                            b["T"].tolist(), b["x"].tolist(), b["p_init"].tolist(), dew)
    dump("gc.json", g)


# ------------------------------------------------------------------------------------------
# gradients w.r.t. the segment parameter vectors (reference autograd through feos_torch/gc_pcsaft.py:14-22, :54-86,
# :470-512) -> gc_seggrad.json
# ------------------------------------------------------------------------------------------
def _mol(name):
    from feos_torch_amd.synthetic import gc_molecule_library
    for nm, segs, bonds in gc_molecule_library():
        if nm == name:
            return segs, bonds
    raise KeyError(name)


# one pair per model class: non-polar, polar, self-, cross- and induced association, polar + associating, branched
SEGGRAD_PAIRS = [("C4", "C3"), ("C3CHO", "C5"), ("C2OH", "C6"), ("C3OH", "C2NH2"), ("C4OH", "C3IA"), ("C2CHO", "C3OH"),
                 ("HCOOC3", "acetone"), ("isobutane", "butanone"), ("2-propanol", "C4NH2"), ("C3IA", "C7"),
                 ("isopentane", "C2OH"), ("C10", "C2")]


def _seggrad_case(GcPcSaftMix, tl, table, pairs, dew, weights):
    from feos_torch_amd.synthetic import GC_KAB
    par = {s: np.asarray(v, dtype=np.float64) for s, v in table}
    seg_l, bon_l, T = [], [], []
    for a, b in pairs:
        (sa, ba), (sb, bb) = _mol(a), _mol(b)
        seg_l.append([sa, sb])
        bon_l.append([ba, bb])
        tc = []
        for segs in (sa, sb):
            m = np.array([par[s][0] for s in segs]); e = np.array([par[s][2] for s in segs])
            tc.append((m * e).sum() / m.sum() * 1.28 * m.sum() ** 0.45)
        T.append(0.6 * min(tc))
    n = len(pairs)
    rng = np.random.default_rng(41)
    phi = rng.uniform(0.9, 1.1, size=(n, 2))
    x = rng.uniform(0.2, 0.8, n)
    ident = [s for s, _ in table]
    kab_list = [k for k in GC_KAB if k[0] in ident and k[1] in ident]
    cols = [torch.tensor([v[k] for _, v in table], dtype=f64, requires_grad=True) for k in range(8)]
    Tt = torch.tensor(T, dtype=f64)
    eos = GcPcSaftMix(ident, tuple(cols), seg_l, bon_l, kab_list, torch.tensor(phi, dtype=f64))
    val, nans = (eos.dew_point if dew else eos.bubble_point)(Tt, torch.tensor(x, dtype=f64), torch.full((n,), 1e5, dtype=f64))
    assert not bool(nans.any())
    (val * torch.tensor(weights[:n], dtype=f64)).sum().backward()
    return {"pairs": [list(p) for p in pairs], "segment_lists": seg_l, "bond_lists": bon_l, "kab_list": [list(k) for k in kab_list],
            "phi": phi.tolist(), "T": T, "z": x.tolist(), "p_init": [1e5] * n, "weights": list(weights[:n]),
            "value": tl(val), "grad_segments": [tl(c.grad) for c in cols]}


def make_gc_seggrad(GcPcSaftMix, dump, tl):
    import os
    from feos_torch_amd.synthetic import load_segment_table
    table = load_segment_table(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "sauer2014_hetero.json"))
    no_c = [(s, v) for s, v in table if s != ">C<"]  # every epsilon_k > 0: the reference's gradients are finite
    w = np.random.default_rng(43).uniform(0.5, 1.5, 64).tolist()
    g = {"table_without_C": [s for s, _ in no_c], "table_full": [s for s, _ in table]}
    for name, dew in (("bubble", False), ("dew", True)):
        g[name] = _seggrad_case(GcPcSaftMix, tl, no_c, SEGGRAD_PAIRS, dew, w)
        # the reference's own test case (tests/test_gc_pcsaft.py:130-222) alone, unit weight
        g[name + "_butane_propane"] = _seggrad_case(GcPcSaftMix, tl, no_c, SEGGRAD_PAIRS[:1], dew, [1.0])
        # full table incl. '>C<' (epsilon_k = 0): the reference's epsilon_k column is NaN, m / sigma / ... are finite
        g[name + "_full_table"] = _seggrad_case(GcPcSaftMix, tl, table, SEGGRAD_PAIRS + [("neopentane", "C5")], dew, w)
    dump("gc_seggrad.json", g)


# ------------------------------------------------------------------------------------------
# reference autograd THROUGH derivatives / helmholtz_energy(_density) (plain torch graphs in the reference:
# feos_torch/pcsaft_pure.py:106-182, pcsaft_mix.py:31-154 / :395-420, gc_pcsaft.py:116-253 / :443-468) -> deriv_grad.json
# loss = sum_i (w_a a + w_p p + ...)_i with seeded weights per row and output; rows are independent, so the input
# gradients are per-row quantities
# ------------------------------------------------------------------------------------------
def _pure_deriv_grad(PcSaftPure, tl, params, T, rho, rng):
    n = len(T)
    x = torch.tensor(params, dtype=f64, requires_grad=True)
    Tt = torch.tensor(T, dtype=f64, requires_grad=True)
    rt = torch.tensor(rho, dtype=f64, requires_grad=True)
    w = rng.uniform(-1.0, 1.0, size=(3, n))
    a, p, dp = PcSaftPure(x).derivatives(Tt, rt)
    (a * torch.tensor(w[0]) + p * torch.tensor(w[1]) + dp * torch.tensor(w[2])).sum().backward()
    out = {"params": params, "T": T, "rho": rho, "w": w.tolist(), "grad_params": tl(x.grad), "grad_T": tl(Tt.grad), "grad_rho": tl(rt.grad)}
    # helmholtz_energy alone
    x2 = torch.tensor(params, dtype=f64, requires_grad=True)
    T2 = torch.tensor(T, dtype=f64, requires_grad=True)
    r2 = torch.tensor(rho, dtype=f64, requires_grad=True)
    PcSaftPure(x2).helmholtz_energy(T2, r2).sum().backward()
    out["a_only"] = {"grad_params": tl(x2.grad), "grad_T": tl(T2.grad), "grad_rho": tl(r2.grad)}
    return out


def _mix_deriv_grad(PcSaftMix, tl, params, kij, T, rho, rng):
    n = len(T)
    x = torch.tensor(params, dtype=f64, requires_grad=True)
    k = torch.tensor(kij, dtype=f64, requires_grad=True)
    Tt = torch.tensor(T, dtype=f64, requires_grad=True)
    rt = torch.tensor(rho, dtype=f64, requires_grad=True)
    w = rng.uniform(-1.0, 1.0, size=(6, n))
    a, p, mu, v = PcSaftMix(x, k).derivatives(Tt, rt)
    wt = torch.tensor(w)
    (a * wt[0] + p * wt[1] + mu[:, 0] * wt[2] + mu[:, 1] * wt[3] + v[:, 0] * wt[4] + v[:, 1] * wt[5]).sum().backward()
    return {"params": params, "kij": kij, "T": T, "rho": rho, "w": w.tolist(), "grad_params": tl(x.grad), "grad_kij": tl(k.grad),
            "grad_T": tl(Tt.grad), "grad_rho": tl(rt.grad)}


def _gc_deriv_grad(GcPcSaftMix, tl, table, seg_l, bon_l, kab_list, phi, T, rho, rng):
    n = len(T)
    ident = [s for s, _ in table]
    cols = [torch.tensor([v[k] for _, v in table], dtype=f64, requires_grad=True) for k in range(8)]
    kab = torch.tensor([k[2] for k in kab_list], dtype=f64, requires_grad=True)
    kl = [(k[0], k[1], kv) for k, kv in zip(kab_list, kab)]
    ph = torch.tensor(phi, dtype=f64, requires_grad=True)
    Tt = torch.tensor(T, dtype=f64, requires_grad=True)
    rt = torch.tensor(rho, dtype=f64, requires_grad=True)
    w = rng.uniform(-1.0, 1.0, size=(6, n))
    a, p, mu, v = GcPcSaftMix(ident, tuple(cols), seg_l, bon_l, kl, ph).derivatives(Tt, rt)
    wt = torch.tensor(w)
    (a * wt[0] + p * wt[1] + mu[:, 0] * wt[2] + mu[:, 1] * wt[3] + v[:, 0] * wt[4] + v[:, 1] * wt[5]).sum().backward()
    return {"segment_lists": seg_l, "bond_lists": bon_l, "kab_list": [list(k) for k in kab_list], "phi": phi, "T": T, "rho": rho,
            "w": w.tolist(), "grad_segments": [tl(c.grad) for c in cols], "grad_kab": tl(kab.grad), "grad_phi": tl(ph.grad),
            "grad_T": tl(Tt.grad), "grad_rho": tl(rt.grad)}


def make_deriv_grad(PcSaftPure, PcSaftMix, GcPcSaftMix, dump, tl, pure_test_params):
    import json
    import os
    from feos_torch_amd.synthetic import GC_KAB, load_segment_table
    here = os.path.dirname(os.path.abspath(__file__))
    rng = np.random.default_rng(53)
    g = {"pure": {}, "mix": {}, "gc": {}}
    # pure: the reference's test rows (tests/test_pcsaft_pure.py:10-21) at a vapour-like and a liquid-like density + seeded random rows
    with open(os.path.join(here, "pure.json")) as f:
        gp = json.load(f)
    g["pure"]["test_inputs"] = _pure_deriv_grad(PcSaftPure, tl, pure_test_params * 2, [300.0] * 12, [0.001] * 6 + [0.012] * 6, rng)
    g["pure"]["random"] = _pure_deriv_grad(PcSaftPure, tl, gp["random"]["params"], gp["random"]["T"], gp["random"]["rho"], rng)
    # mixtures: the 14 reference cases (tests/test_pcsaft_mix.py:17-39) + the seeded random rows of mix.json
    with open(os.path.join(here, "mix.json")) as f:
        gm = json.load(f)
    ti = gm["test_inputs"]
    g["mix"]["test_inputs"] = _mix_deriv_grad(PcSaftMix, tl, ti["params"], ti["kij"], ti["T"], ti["rho"], rng)
    rr = gm["random"]
    g["mix"]["random"] = _mix_deriv_grad(PcSaftMix, tl, rr["params"], rr["kij"], rr["T"], rr["rho"], rng)
    # gc: one pair per model class on the table without '>C<' (all reference gradients finite) and the reference's 11 test
    # pairs on the full table (its epsilon_k and phi gradients are NaN there: sqrt(0) under autograd)
    table = load_segment_table(os.path.join(os.path.dirname(here), "data", "sauer2014_hetero.json"))
    no_c = [(s, v) for s, v in table if s != ">C<"]
    seg_l, bon_l = [], []
    for a, b in SEGGRAD_PAIRS:
        (sa, ba), (sb, bb) = _mol(a), _mol(b)
        seg_l.append([sa, sb])
        bon_l.append([ba, bb])
    n = len(seg_l)
    phi = rng.uniform(0.9, 1.1, size=(n, 2)).tolist()
    T = rng.uniform(250.0, 400.0, n).tolist()
    rho = (np.stack([rng.uniform(0.2, 0.8, n), rng.uniform(0.2, 0.8, n)], axis=1) * (10.0 ** rng.uniform(-5, -2.4, n))[:, None]).tolist()
    ident = [s for s, _ in no_c]
    kl = [k for k in GC_KAB if k[0] in ident and k[1] in ident]
    g["gc"]["classes"] = _gc_deriv_grad(GcPcSaftMix, tl, no_c, seg_l, bon_l, kl, phi, T, rho, rng)
    g["gc"]["classes"]["table"] = ident
    n = len(GC_TEST_SEGMENTS)
    g["gc"]["test_inputs"] = _gc_deriv_grad(GcPcSaftMix, tl, table, GC_TEST_SEGMENTS, GC_TEST_BONDS, GC_TEST_KAB, [[1.1, 0.98]] * n,
                                            [300.0] * n, [[0.001, 0.002]] * n, rng)
    g["gc"]["test_inputs"]["table"] = [s for s, _ in table]
    dump("deriv_grad.json", g)


# ------------------------------------------------------------------------------------------
# n-component mixtures -> mixn.json
# The reference's n-component code path cannot actually be run: `kij` must be None for n != 2 (pcsaft_mix.py:75-76) and with
# kij = None helmholtz_energy_density fails at `self.kij[cross_associating, 1]` (:141) for every n.  What CAN be produced with the
# unmodified reference is the n = 2 case with kij = 0, on rows with at most one associating component: the n-component code of
# this repo run at nc = 2 must reproduce it, and for nc > 2 it is pinned by exact properties (tests/test_mixn_gpu.py).
# ------------------------------------------------------------------------------------------
def make_mixn(PcSaftMix, dump, tl):
    rng = np.random.default_rng(61)
    n, nc = 48, 2
    P = np.zeros((n, nc, 8))
    P[:, :, 0] = rng.uniform(1.0, 3.5, (n, nc))
    P[:, :, 1] = rng.uniform(2.8, 4.2, (n, nc))
    P[:, :, 2] = rng.uniform(150.0, 350.0, (n, nc))
    polar = rng.random((n, nc)) < 0.4
    P[:, :, 3] = np.where(polar, rng.uniform(0.5, 3.0, (n, nc)), 0.0)
    for r in range(n):  # at most ONE associating component per row, in a third of the rows
        if r % 3 == 0:
            c = rng.integers(0, nc)
            P[r, c, 4] = rng.uniform(0.001, 0.05)
            P[r, c, 5] = rng.uniform(1000.0, 3000.0)
            P[r, c, 6], P[r, c, 7] = [(1, 1), (2, 1), (1, 2)][rng.integers(0, 3)]
    T = rng.uniform(200.0, 450.0, n)
    d = P[:, :, 1] * (1 - 0.12 * np.exp(-3 * P[:, :, 2] / T[:, None]))
    x = rng.dirichlet(np.ones(nc), n)
    eta = np.where(rng.random(n) < 0.5, rng.uniform(0.2, 0.42, n), 10.0 ** rng.uniform(-7, -2, n))
    rho = x * (eta / (np.pi / 6 * (x * P[:, :, 0] * d**3).sum(axis=1)))[:, None]
    eos = PcSaftMix(torch.tensor(P, dtype=f64), torch.zeros((n, 2), dtype=f64))
    a, p, mu, v = eos.derivatives(torch.tensor(T, dtype=f64), torch.tensor(rho, dtype=f64))
    dump("mixn.json", {"nc2_kij0": {"params": P.tolist(), "T": T.tolist(), "rho": rho.tolist(), "a": tl(a), "p": tl(p), "mu": tl(mu), "v": tl(v)}})
