"""CPU: the oracle's n-component restatement (oracle/pcsaft_mixn.hpp) against the only outputs the unmodified reference can
produce for it (n = 2 with kij = 0, tests/golden/mixn.json; see tests/test_mixn_gpu.py for why) and against exact properties."""
import numpy as np

from conftest import load_golden


def test_nc2_vs_reference_python(oracle):
    g = load_golden("mixn.json")["nc2_kij0"]
    P, T, rho = (np.array(g[k]) for k in ("params", "T", "rho"))
    a, p, mu, v = oracle.mixn_derivatives(P, T, rho, prec=0)
    assert np.max(np.abs(a - np.array(g["a"]))) < 1e-14
    assert np.max(np.abs(p - np.array(g["p"]))) < 1e-14
    assert np.max(np.abs(mu - np.array(g["mu"]))) < 1e-13
    assert np.max(np.abs(v / np.array(g["v"]) - 1.0)) < 1e-11


def test_splitting_a_component_changes_nothing(oracle):
    g = load_golden("mixn.json")["nc2_kij0"]
    P, T, rho = (np.array(g[k]) for k in ("params", "T", "rho"))
    ok = (P[:, 1, 6] + P[:, 1, 7]) == 0  # the duplicated component must not be the associating one
    a, p, mu, v = oracle.mixn_derivatives(P[ok], T[ok], rho[ok], prec=1)
    P3 = np.concatenate([P, P[:, 1:2, :]], axis=1)[ok]
    rho3 = np.concatenate([rho[:, :1], 0.3 * rho[:, 1:2], 0.7 * rho[:, 1:2]], axis=1)[ok]
    a3, p3, mu3, v3 = oracle.mixn_derivatives(P3, T[ok], rho3, prec=1)
    assert np.max(np.abs(a3 - a)) < 1e-15 and np.max(np.abs(p3 - p)) < 1e-14
    assert np.max(np.abs(mu3[:, :2] - mu)) < 1e-13 and np.max(np.abs(mu3[:, 2] - mu3[:, 1])) < 1e-13
    assert np.max(np.abs(v3[:, :2] / v - 1.0)) < 1e-12
