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
