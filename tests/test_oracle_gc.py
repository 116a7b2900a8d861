"""CPU: pin the gc-PC-SAFT oracle (oracle/gc_pcsaft.hpp) against the UNMODIFIED reference Python
(tests/golden/gc.json from tests/golden/make_golden.py): (a, p, mu, v) on the 11 molecule pairs
of tests/test_gc_pcsaft.py:17-49 (reference tolerance abs 1e-14 / 1e-11, :122-127), the
n-butane/propane bubble and dew points with dp/dk_ab (:130-222, abs 1e-8 Pa / abs 1) and 48
seeded random rows.  The segment table tests/data/sauer2014_hetero.json is the reference's own
test data file (tests/sauer2014_hetero.json)."""
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden


@pytest.fixture(scope="module")
def gg():
    return load_golden("gc.json")


@pytest.fixture(scope="module")
def table():
    from feos_torch_amd.synthetic import load_segment_table

    return load_segment_table(os.path.join(ROOT, "tests", "data", "sauer2014_hetero.json"))


def test_segment_table_is_the_reference_file(table):
    assert len(table) == 23 and table[0][0] == "CH3" and abs(table[0][1][0] - 0.77247) < 1e-15


@pytest.mark.parametrize("robust", [False, True])
def test_derivatives_match_reference_python(oracle, gg, table, robust):
    g = gg["test_inputs"]
    enc = oracle.gc_encode(table, g["segment_lists"], g["bond_lists"], [tuple(k) for k in g["kab_list"]])
    a, p, mu, v = oracle.gc_derivatives(enc, g["phi"], g["T"], g["rho"], robust=robust)
    assert np.max(np.abs(a - np.array(g["a"]))) < 1e-14
    assert np.max(np.abs(p - np.array(g["p"]))) < 1e-14
    assert np.max(np.abs(mu - np.array(g["mu"]))) < 1e-13
    assert np.max(np.abs(v / np.array(g["v"]) - 1)) < 1e-12


@pytest.mark.parametrize("key,dew", [("test_bubble", False), ("test_dew", True)])
def test_bubble_dew_reference_case(oracle, gg, table, key, dew):
    g = gg[key]
    ref = g["result"]
    kab = [(a, b, k) for (a, b), k in zip(g["kab_pairs"], g["kab_vals"])]
    enc = oracle.gc_encode(table, g["segment_lists"], g["bond_lists"], kab)
    p, rho4, st = oracle.gc_bubble_dew(enc, g["phi"], g["T"], g["z"], g["p_init"], dew, prec=1)
    assert st.tolist() == ref["nans"]
    assert abs(p[0] - ref["value"][0]) < 1e-8  # tests/test_gc_pcsaft.py:173 / :221
    val, grad = oracle.gc_bubble_dew_grad(enc, g["phi"], g["T"], rho4, dew, *g["kab_pairs"][0])
    assert abs(grad[0, 0] - ref["grad_kab"][0]) < 1e-6
    assert abs(grad[0, 3] - ref["grad_T"][0]) < 1e-9
    # finite difference in k_ab through the reference tail (:174 / :222: abs 1)
    fd = (g["value_kab_plus_1e-7"][0] - ref["value"][0]) / 1e-7
    assert abs(grad[0, 0] - fd) < 1.0


@pytest.mark.parametrize("name,dew", [("bubble", False), ("dew", True)])
def test_random_rows(oracle, gg, table, name, dew):
    from feos_torch_amd.synthetic import gc_batch

    g = gg["random"]
    b = gc_batch(g["n"], table, seed=g["seed"])
    enc = oracle.gc_encode(table, b["segment_lists"], b["bond_lists"], b["kab_list"])
    a, p, mu, v = oracle.gc_derivatives(enc, b["phi"], b["T"], g["rho"], robust=False)
    assert np.max(np.abs(a - np.array(g["a"])) / np.maximum(1e-6, np.abs(np.array(g["a"])))) < 1e-10
    assert np.max(np.abs(mu - np.array(g["mu"]))) < 1e-10
    ref = g[name]
    pb, rho4, st = oracle.gc_bubble_dew(enc, b["phi"], b["T"], b["x"], b["p_init"], dew, prec=1)
    assert st.tolist() == ref["nans"]
    assert np.max(np.abs(pb[~st] / np.array(ref["value"]) - 1)) < 1e-9
    # d p / d k_ab(CH3, CH2) and d p / dT against the reference's autograd (d/dphi is NaN in the
    # reference: sqrt(0) of the epsilon_k = 0 segment '>C<' under autograd)
    ik = g["kab_pairs"].index(["CH3", "CH2"])
    val, grad = oracle.gc_bubble_dew_grad(enc, b["phi"], b["T"], rho4, dew, "CH3", "CH2")
    want_k = np.array(ref["grad_kab"])[ik]
    assert abs(grad[:, 0].sum() - want_k) < 1e-6 * max(1.0, abs(want_k))
    assert np.max(np.abs(grad[:, 3] - np.array(ref["grad_T"])) / np.maximum(1e-12, np.abs(np.array(ref["grad_T"])))) < 1e-8
