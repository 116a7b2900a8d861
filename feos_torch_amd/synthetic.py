"""Synthetic workloads for the benchmark and the large-batch parity tests.

The reference publishes no workload; the distributions below are the ones
fixed in SURVEY.md §8(d) so that every run (GPU path, CPU oracle, later rounds) sees the
same rows for the same seed.  numpy PCG64 is platform independent.
"""
import numpy as np


def pure_batch(n, seed=2026):
    """PcSaftPure rows: parameters [n, 8] (m, sigma, epsilon_k, mu, kappa_ab, epsilon_k_ab,
    na, nb — README.md:12 of the reference) and temperatures [n], all float64.

    T = epsilon_k * 1.28 * m**0.45 * tau, tau ~ U[0.55, 0.90]: 1.28 m^0.45 fits the reduced
    critical temperature of non-polar PC-SAFT chains, so every row is sub-critical.
    """
    rng = np.random.default_rng(seed)
    m = rng.uniform(1.0, 4.0, n)
    sigma = rng.uniform(2.8, 4.2, n)
    eps = rng.uniform(150.0, 350.0, n)
    polar = rng.random(n) < 0.5
    mu = np.where(polar, rng.uniform(0.5, 3.0, n), 0.0)
    assoc = rng.random(n) < 0.5
    kappa = np.where(assoc, rng.uniform(0.001, 0.05, n), 0.0)
    eps_ab = np.where(assoc, rng.uniform(1000.0, 3000.0, n), 0.0)
    scheme = rng.integers(0, 3, n)
    na = np.where(assoc, np.array([1.0, 2.0, 1.0])[scheme], 0.0)
    nb = np.where(assoc, np.array([1.0, 1.0, 2.0])[scheme], 0.0)
    tau = rng.uniform(0.55, 0.90, n)
    params = np.stack([m, sigma, eps, mu, kappa, eps_ab, na, nb], axis=1)
    T = eps * 1.28 * m**0.45 * tau
    return np.ascontiguousarray(params), np.ascontiguousarray(T)


def pure_pressures(n, seed=2027):
    """Specified pressures for liquid_density (config 3): 1e5 Pa * 10**U[0, 2]."""
    rng = np.random.default_rng(seed)
    return 1e5 * 10.0 ** rng.uniform(0.0, 2.0, n)


def _component(rng, n, m_hi):
    m = rng.uniform(1.0, m_hi, n)
    sigma = rng.uniform(2.8, 4.2, n)
    eps = rng.uniform(150.0, 350.0, n)
    return m, sigma, eps


def mix_batch(n, seed=2028):
    """PcSaftMix rows (config 4): parameters [n, 2, 8], kij [n, 2], T [n], x [n], p_init [n] Pa.

    Two independent parameter draws per row with m ~ U[1, 3]; association classes in equal
    shares by row index mod 6 (mirrors the case list of the reference's
    tests/test_pcsaft_mix.py:17-32): 0 none, 1 polar only, 2 one self-associating component,
    3 cross-association (mean combining rule), 4 cross-association with explicit eps_AiBj,
    5 induced association (one self-associating + one B-site-only component).
    k_ij ~ U[-0.1, 0.1], x ~ U[0.1, 0.9], T = 0.6 * min_i(eps_i * 1.28 * m_i**0.45),
    p_init = 1e5 Pa (the reference tests' initial pressure).
    """
    rng = np.random.default_rng(seed)
    par = np.zeros((n, 2, 8))
    cls = np.arange(n) % 6
    for c in range(2):
        m, sigma, eps = _component(rng, n, 3.0)
        par[:, c, 0], par[:, c, 1], par[:, c, 2] = m, sigma, eps
        polar = (rng.random(n) < 0.5) & (cls != 0)
        polar |= (cls == 1) & (c == 0)
        par[:, c, 3] = np.where(polar, rng.uniform(0.5, 3.0, n), 0.0)
        kappa = rng.uniform(0.001, 0.05, n)
        eab = rng.uniform(1000.0, 3000.0, n)
        scheme = rng.integers(0, 3, n)
        na = np.array([1.0, 2.0, 1.0])[scheme]
        nb = np.array([1.0, 1.0, 2.0])[scheme]
        which = rng.integers(0, 2, n)  # the distinguished component for classes 2 and 5
        self_assoc = (cls == 3) | (cls == 4) | ((cls == 2) & (which == c)) | ((cls == 5) & (which == c))
        induced = (cls == 5) & (which != c)
        par[:, c, 4] = np.where(self_assoc | induced, kappa, 0.0)
        par[:, c, 5] = np.where(self_assoc | induced, eab, 0.0)
        par[:, c, 6] = np.where(self_assoc, na, 0.0)
        par[:, c, 7] = np.where(self_assoc, nb, np.where(induced, np.array([1.0, 2.0])[scheme % 2], 0.0))
        if c == 0:
            which0 = which
        else:
            # both components must use the SAME draw of `which`
            fix = (cls == 2) | (cls == 5)
            self1 = (cls == 3) | (cls == 4) | ((cls == 2) & (which0 == 1)) | ((cls == 5) & (which0 == 1))
            ind1 = (cls == 5) & (which0 != 1)
            par[:, 1, 4] = np.where(self1 | ind1, kappa, 0.0)
            par[:, 1, 5] = np.where(self1 | ind1, eab, 0.0)
            par[:, 1, 6] = np.where(self1, na, 0.0)
            par[:, 1, 7] = np.where(self1, nb, np.where(ind1, np.array([1.0, 2.0])[scheme % 2], 0.0))
            del fix
    kij = np.zeros((n, 2))
    kij[:, 0] = rng.uniform(-0.1, 0.1, n)
    kij[:, 1] = np.where(cls == 4, rng.uniform(1000.0, 3000.0, n), 0.0)
    x = rng.uniform(0.1, 0.9, n)
    tc = par[:, :, 2] * 1.28 * par[:, :, 0] ** 0.45
    T = 0.6 * tc.min(axis=1)
    p_init = np.full(n, 1e5)
    return (np.ascontiguousarray(par), np.ascontiguousarray(kij), np.ascontiguousarray(T),
            np.ascontiguousarray(x), p_init)


# ---------------------------------------------------------------------------------------------
# heterosegmented gc-PC-SAFT (config 5)
# ---------------------------------------------------------------------------------------------
def _chain(segs):
    return list(segs), [[i, i + 1] for i in range(len(segs) - 1)]


def gc_molecule_library():
    """(name, segments, bonds) built from the 23 segments of the reference's
    tests/sauer2014_hetero.json: n-alkanes C2-C10, branched alkanes, 1-alcohols, aldehydes,
    formates, ketones, primary amines and one molecule carrying the induced-association
    pseudo-segment 'IA' (tests/test_gc_pcsaft.py:17-29 uses the same building blocks)."""
    lib = []
    for n in range(2, 11):
        lib.append((f"C{n}",) + tuple(_chain(["CH3"] + ["CH2"] * (n - 2) + ["CH3"])))
    lib.append(("isobutane", ["CH3", ">CH", "CH3", "CH3"], [[0, 1], [1, 2], [1, 3]]))
    lib.append(("neopentane", ["CH3", ">C<", "CH3", "CH3", "CH3"], [[0, 1], [1, 2], [1, 3], [1, 4]]))
    lib.append(("isopentane", ["CH3", ">CH", "CH3", "CH2", "CH3"], [[0, 1], [1, 2], [1, 3], [3, 4]]))
    for n in range(2, 7):
        lib.append((f"C{n}OH",) + tuple(_chain(["CH3"] + ["CH2"] * (n - 1) + ["OH"])))
        lib.append((f"C{n}NH2",) + tuple(_chain(["CH3"] + ["CH2"] * (n - 1) + ["NH2"])))
    for n in range(2, 6):
        lib.append((f"C{n}CHO",) + tuple(_chain(["CH3"] + ["CH2"] * (n - 2) + ["CH=O"])))
        lib.append((f"HCOOC{n}",) + tuple(_chain(["HCOO"] + ["CH2"] * (n - 1) + ["CH3"])))
    lib.append(("acetone", ["CH3", ">C=O", "CH3"], [[0, 1], [1, 2]]))
    lib.append(("butanone", ["CH3", ">C=O", "CH2", "CH3"], [[0, 1], [1, 2], [2, 3]]))
    lib.append(("2-propanol", ["CH3", ">CH", "CH3", "OH"], [[0, 1], [1, 2], [1, 3]]))
    lib.append(("C3IA", ["CH3", "CH2", "CH2", "IA"], [[0, 1], [1, 2], [2, 3]]))
    return lib


GC_KAB = [("CH3", "CH=O", 0.03), (">CH", "HCOO", -0.01), ("CH3", "CH2", -0.02), ("CH2", "OH", 0.01)]


def gc_batch(n, segment_table, seed=2029):
    """GcPcSaftMix rows (config 5).  segment_table = list of (identifier, array(8)) in the order
    of the parameter file.  -> dict(segment_lists, bond_lists, kab_list, phi [n,2], T, x, p_init).
    Molecule pairs are drawn from gc_molecule_library(); phi ~ U[0.9, 1.1]^2;
    T = 0.6 * min_i(eps_mix_i * 1.28 * m_mix_i**0.45) with the molecule-level averages of
    feos_torch/gc_pcsaft.py:66-70; x ~ U[0.1, 0.9]; p_init = 1e5 Pa."""
    rng = np.random.default_rng(seed)
    lib = gc_molecule_library()
    par = {s: np.asarray(v, dtype=np.float64) for s, v in segment_table}
    tc = []
    for _, segs, _ in lib:
        m = np.array([par[s][0] for s in segs])
        e = np.array([par[s][2] for s in segs])
        mm = m.sum()
        tc.append((m * e).sum() / mm * 1.28 * mm**0.45)
    tc = np.array(tc)
    pick = rng.integers(0, len(lib), size=(n, 2))
    segment_lists = [[lib[a][1], lib[b][1]] for a, b in pick]
    bond_lists = [[lib[a][2], lib[b][2]] for a, b in pick]
    phi = rng.uniform(0.9, 1.1, size=(n, 2))
    x = rng.uniform(0.1, 0.9, n)
    T = 0.6 * tc[pick].min(axis=1)
    return {"segment_lists": segment_lists, "bond_lists": bond_lists, "kab_list": list(GC_KAB), "phi": phi,
            "T": np.ascontiguousarray(T), "x": x, "p_init": np.full(n, 1e5), "pick": pick}


def load_segment_table(path):
    """[(identifier, array(8) = m, sigma, epsilon_k, mu, kappa_ab, epsilon_k_ab, na, nb)] from a
    feos segment-record JSON (format of the reference's tests/sauer2014_hetero.json)."""
    import json

    with open(path) as f:
        recs = json.load(f)
    out = []
    for r in recs:
        m = r["model_record"]
        out.append((r["identifier"], np.array([m["m"], m["sigma"], m["epsilon_k"], m.get("mu", 0.0),
                                               m.get("kappa_ab", 0.0), m.get("epsilon_k_ab", 0.0),
                                               m.get("na", 0.0), m.get("nb", 0.0)], dtype=np.float64)))
    return out
