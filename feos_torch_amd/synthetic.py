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
