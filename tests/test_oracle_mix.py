"""CPU: pin the mixture oracle (oracle/pcsaft_mix.hpp, oracle/mix_solver.hpp) against the
UNMODIFIED reference Python (tests/golden/mix.json, written by tests/golden/make_golden.py):
  * (a, p, mu, v) on the 14 binary cases of tests/test_pcsaft_mix.py:17-39 (reference tolerance
    abs 1e-14 on a/mu/p and 1e-11 on v, :119-124) and on 60 seeded random rows;
  * bubble / dew pressure and its torch gradients on the inputs of tests/test_pcsaft_mix.py:
    127-251 (reference tolerance abs 1e-8 Pa; gradient abs 1) and on the random rows.
The reference stores no numeric answers for mixtures (it compares with the `feos` package at
run time), so bubble/dew values are pinned by the model definition + the reference tail:
"parity unpinned by stored values" (SURVEY.md §8c).
"""
import numpy as np
import pytest

from conftest import load_golden


@pytest.fixture(scope="module")
def gm():
    return load_golden("mix.json")


@pytest.mark.parametrize("robust", [False, True])
def test_derivatives_match_reference_python(oracle, gm, robust):
    g = gm["test_inputs"]
    a, p, mu, v = oracle.mix_derivatives(g["params"], g["kij"], g["T"], g["rho"], robust=robust)
    assert np.max(np.abs(a - np.array(g["a"]))) < 1e-14
    assert np.max(np.abs(p - np.array(g["p"]))) < 1e-14
    assert np.max(np.abs(mu - np.array(g["mu"]))) < 1e-14
    assert np.max(np.abs(v - np.array(g["v"]))) < 1e-11
    # row 10 (a lone B-site-only component does not associate) equals row 0
    assert a[10] == a[0]


@pytest.mark.parametrize("robust", [False, True])
def test_derivatives_random_rows(oracle, gm, robust):
    g = gm["random"]
    a, p, mu, v = oracle.mix_derivatives(g["params"], g["kij"], g["T"], g["rho"], robust=robust)
    assert np.max(np.abs(a - np.array(g["a"]))) < 1e-13
    assert np.max(np.abs(p - np.array(g["p"]))) < 1e-13
    assert np.max(np.abs(mu - np.array(g["mu"]))) < 1e-10
    assert np.max(np.abs(v / np.array(g["v"]) - 1)) < 1e-9


@pytest.mark.parametrize("key,dew", [("test_bubble", False), ("test_dew", True)])
def test_bubble_dew_reference_cases(oracle, gm, key, dew):
    g = gm[key]
    ref = g["result"]
    p, rho4, st = oracle.mix_bubble_dew(g["params"], g["kij"], g["T"], g["z"], g["p_init"], dew, prec=1)
    assert st.tolist() == ref["nans"]
    assert np.max(np.abs(p - np.array(ref["value"]))) < 1e-8  # tests/test_pcsaft_mix.py:190-191
    val, grad = oracle.mix_bubble_dew_grad(g["params"], g["kij"], g["T"], rho4, dew)
    gk = np.array(ref["grad_kij"])
    assert abs(grad[0, 16] - gk[0][0]) < 1e-3
    # gradient vs the finite difference of the two rows (k_ij and k_ij + h), :192 / :251
    fd = (p[1] - p[0]) / g["h"]
    assert abs(grad[0, 16] - fd) < 1.0


@pytest.mark.parametrize("name,dew", [("bubble", False), ("dew", True)])
def test_bubble_dew_random_rows(oracle, gm, name, dew):
    g = gm["random"]
    ref = g[name]
    p, rho4, st = oracle.mix_bubble_dew(g["params"], g["kij"], g["T"], g["z"], g["p_init"], dew, prec=1)
    assert st.tolist() == ref["nans"]
    ok = ~st
    want = np.array(ref["value"])
    # Rows where the reference's own association iteration (start 0.2, step-back, <= 50 sweeps,
    # feos_torch/pcsaft_mix.py:363-385) runs away — A-site excess with X_A > 0.2 — carry garbage in
    # the golden file (the reference Python returns garbage there); they are identified by
    # literal != safeguarded evaluation at the converged densities and excluded.
    P, K, T = np.array(g["params"]), np.array(g["kij"]), np.array(g["T"])
    valid = np.ones(len(T), dtype=bool)
    noise = np.zeros(len(T))
    for cols in (slice(0, 2), slice(2, 4)):
        rho = np.where(st[:, None], 1e-3, rho4[:, cols])
        lit = oracle.mix_derivatives(P, K, T, rho, robust=False)
        rob = oracle.mix_derivatives(P, K, T, rho, robust=True)
        exact = oracle.mix_derivatives_exact(P, K, T, rho)
        valid &= np.abs(lit[2] - rob[2]).max(axis=1) < 1e-9
        # the reference's fp64 formulas at these densities against the exact model: the cancellation of its
        # self-association term (xb = 2/(sqrt + aux), feos_torch/pcsaft_mix.py:237-238), measured per row
        noise = np.maximum(noise, np.abs(lit[2] - exact[2]).max(axis=1))
    assert valid[ok].mean() > 0.9
    keep = valid[ok]
    noise = noise[ok]
    # The golden values are the reference's fp64 tail at these densities.  Per-row bound: 1e-10 where the reference's
    # own evaluation is clean, 10 x its measured cancellation error on the (two) strongly associating rows
    clean = noise < 1e-10
    assert (keep & ~clean).sum() <= 3
    tol = 1e-10 + 10.0 * noise
    err = np.abs(p[ok] / want - 1)
    assert np.all(err[keep] <= tol[keep]), (err[keep] / tol[keep]).max()
    p64, _, st64 = oracle.mix_bubble_dew(g["params"], g["kij"], g["T"], g["z"], g["p_init"], dew, prec=0)
    assert np.array_equal(st64, st)
    err64 = np.abs(p64[ok] / want - 1)
    assert np.all(err64[keep] <= tol[keep]), (err64[keep] / tol[keep]).max()
    val, grad = oracle.mix_bubble_dew_grad(P[ok], K[ok], T[ok], rho4[ok], dew)
    wantg = np.concatenate([np.array(ref["grad_params"])[ok].reshape(-1, 16), np.array(ref["grad_kij"])[ok],
                            np.array(ref["grad_T"])[ok, None]], axis=1)
    scale = np.abs(wantg).max(axis=1, keepdims=True)
    gerr = (np.abs(grad - wantg) / scale).max(axis=1)
    gtol = 5e-10 + 30.0 * noise  # the gradient differentiates the cancelling term
    assert np.all(gerr[keep] <= gtol[keep]), (gerr[keep] / gtol[keep]).max()
