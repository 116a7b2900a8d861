"""GPU parity of the gc segment-parameter gradients: `pcs_gc_segment_gradient` behind
GcPcSaftMix.bubble_point / dew_point vs
  * the reference's own torch autograd through feos_torch/gc_pcsaft.py:14-22, :54-86, :470-512 (fixtures
    tests/golden/gc_seggrad.json, written by tests/golden/make_golden.py from the unmodified reference Python), and
  * central finite differences of the CPU oracle's forward evaluation (independent of the kernel's analytic chain).
Tolerance: 1e-7 of the largest entry of the same parameter column (gradients of different parameters differ by
orders of magnitude: d/dkappa_ab ~ 1e6, d/depsilon_k ~ 1e2)."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
f64 = torch.float64
TOL = 1e-7


@pytest.fixture(scope="module")
def amd():
    assert torch.cuda.is_available()
    import feos_torch_amd

    return feos_torch_amd


@pytest.fixture(scope="module")
def gs():
    return load_golden("gc_seggrad.json")


@pytest.fixture(scope="module")
def table():
    from feos_torch_amd.synthetic import load_segment_table

    return load_segment_table(os.path.join(ROOT, "tests", "data", "sauer2014_hetero.json"))


def _run(amd, tab, g, dew, weights=None):
    ident = [s for s, _ in tab]
    cols = [torch.tensor([v[k] for _, v in tab], dtype=f64, requires_grad=True) for k in range(8)]
    eos = amd.GcPcSaftMix(ident, tuple(cols), g["segment_lists"], g["bond_lists"], [tuple(k) for k in g["kab_list"]],
                          torch.tensor(g["phi"], dtype=f64))
    p, nans = (eos.dew_point if dew else eos.bubble_point)(torch.tensor(g["T"], dtype=f64), torch.tensor(g["z"], dtype=f64),
                                                           torch.tensor(g["p_init"], dtype=f64))
    assert not bool(nans.any())
    w = torch.tensor(g["weights"] if weights is None else weights, dtype=f64)
    (p * w).sum().backward()
    return p.detach().numpy(), np.stack([c.grad.numpy() for c in cols], axis=0)  # [8, S]


def _close(got, ref, tol=TOL):
    """per parameter column: |got - ref| <= tol * max|ref column|"""
    for k in range(8):
        scale = np.nanmax(np.abs(ref[k]))
        if scale == 0.0:
            assert np.all(got[k] == 0.0), k
            continue
        err = np.nanmax(np.abs(got[k] - ref[k])) / scale
        assert err < tol, (k, err)


@pytest.mark.parametrize("key,dew", [("bubble", False), ("dew", True), ("bubble_butane_propane", False), ("dew_butane_propane", True)])
def test_segment_gradients_vs_reference_autograd(amd, gs, table, key, dew):
    """Table without '>C<' (every epsilon_k > 0): all eight columns of the reference's autograd are finite."""
    g = gs[key]
    tab = [(s, v) for s, v in table if s in gs["table_without_C"]]
    p, grad = _run(amd, tab, g, dew)
    assert np.max(np.abs(p / np.array(g["value"]) - 1.0)) < 1e-9
    ref = np.array(g["grad_segments"])
    assert np.all(np.isfinite(ref))
    _close(grad, ref)


@pytest.mark.parametrize("key,dew", [("bubble_full_table", False), ("dew_full_table", True)])
def test_full_table_is_finite_where_the_reference_is_nan(amd, oracle, gs, table, key, dew):
    """With '>C<' (epsilon_k = 0) in the table the reference's d/depsilon_k is NaN for EVERY segment (sqrt(0) under
    autograd, feos_torch/gc_pcsaft.py:181-186); its other seven columns are finite and must agree.  The epsilon_k column
    here is finite and is checked against finite differences of the oracle's forward evaluation."""
    g = gs[key]
    p, grad = _run(amd, table, g, dew)
    ref = np.array(g["grad_segments"])
    assert np.all(np.isnan(ref[2])) and np.all(np.isfinite(np.delete(ref, 2, axis=0)))
    assert np.all(np.isfinite(grad))
    keep = [0, 1, 3, 4, 5, 6, 7]
    full_ref = ref.copy()
    full_ref[2] = grad[2]  # not comparable (NaN in the reference)
    _close(grad, full_ref)
    # epsilon_k (and everything else once more) vs finite differences at the kernel's own converged densities
    enc = oracle.gc_encode(table, g["segment_lists"], g["bond_lists"], [tuple(k) for k in g["kab_list"]])
    _, rho4, st = oracle.gc_bubble_dew(enc, np.array(g["phi"]), np.array(g["T"]), np.array(g["z"]), np.array(g["p_init"]), dew)
    assert not st.any()
    fd = oracle.gc_segment_grad_fd(enc, np.array(g["phi"]), np.array(g["T"]), rho4, dew, weights=np.array(g["weights"]))
    for k in range(8):
        scale = np.max(np.abs(fd[:, k]))
        mask = fd[:, k] != 0.0  # structurally-zero entries are not perturbed by the finite differences
        if scale > 0:
            assert np.max(np.abs(grad[k][mask] - fd[:, k][mask])) / scale < 2e-6, k


def test_random_batch_vs_oracle_finite_differences(amd, oracle, table):
    """2,000 rows of the config-5 distribution (all model classes, '>C<' included), random upstream weights; failed rows
    are compacted before the gradient kernel runs (the class order is then not used)."""
    from feos_torch_amd.synthetic import gc_batch

    n = 2000
    b = gc_batch(n, table, seed=77)
    ident = [s for s, _ in table]
    rng = np.random.default_rng(5)
    for dew in (False, True):
        cols = [torch.tensor([v[k] for _, v in table], dtype=f64, requires_grad=True) for k in range(8)]
        eos = amd.GcPcSaftMix(ident, tuple(cols), b["segment_lists"], b["bond_lists"], b["kab_list"], torch.tensor(b["phi"], dtype=f64))
        p, nans = (eos.dew_point if dew else eos.bubble_point)(torch.tensor(b["T"], dtype=f64), torch.tensor(b["x"], dtype=f64),
                                                               torch.tensor(b["p_init"], dtype=f64))
        ok = ~nans.numpy()
        w = rng.uniform(0.5, 1.5, int(ok.sum()))
        (p * torch.tensor(w, dtype=f64)).sum().backward()
        grad = np.stack([c.grad.numpy() for c in cols], axis=1)  # [S, 8]
        assert np.all(np.isfinite(grad))
        seg_ok = [s for s, k in zip(b["segment_lists"], ok) if k]
        bon_ok = [s for s, k in zip(b["bond_lists"], ok) if k]
        enc = oracle.gc_encode(table, seg_ok, bon_ok, b["kab_list"])
        _, rho4, st = oracle.gc_bubble_dew(enc, b["phi"][ok], b["T"][ok], b["x"][ok], b["p_init"][ok], dew)
        good = ~st
        # rows the oracle fails on contribute to the kernel's sum but not to the finite differences: give them weight 0
        if not good.all():
            cols2 = [torch.tensor([v[k] for _, v in table], dtype=f64, requires_grad=True) for k in range(8)]
            eos2 = amd.GcPcSaftMix(ident, tuple(cols2), seg_ok, bon_ok, b["kab_list"], torch.tensor(b["phi"][ok], dtype=f64))
            p2, nans2 = (eos2.dew_point if dew else eos2.bubble_point)(torch.tensor(b["T"][ok], dtype=f64), torch.tensor(b["x"][ok], dtype=f64),
                                                                       torch.tensor(b["p_init"][ok], dtype=f64))
            assert not bool(nans2.any())
            w = w * good
            (p2 * torch.tensor(w, dtype=f64)).sum().backward()
            grad = np.stack([c.grad.numpy() for c in cols2], axis=1)
            rho4 = np.where(good[:, None], rho4, 1e-3)
        fd = oracle.gc_segment_grad_fd(enc, b["phi"][ok], b["T"][ok], rho4, dew, weights=w)
        for k in range(8):
            scale = np.max(np.abs(fd[:, k]))
            mask = fd[:, k] != 0.0
            if scale > 0:
                err = np.max(np.abs(grad[:, k][mask] - fd[:, k][mask])) / scale
                assert err < 5e-6, (dew, k, err)


def test_gradient_accumulates_and_validates(amd, table):
    """C-ABI level: grad_seg is accumulated (two calls double it), gout = NULL means unit weights, bad arguments are refused."""
    from feos_torch_amd import native
    from feos_torch_amd.synthetic import gc_batch

    n = 300
    b = gc_batch(n, table, seed=9)
    ident = [s for s, _ in table]
    eos = amd.GcPcSaftMix(ident, tuple(torch.tensor([v[k] for _, v in table], dtype=f64) for k in range(8)), b["segment_lists"],
                          b["bond_lists"], b["kab_list"], torch.tensor(b["phi"], dtype=f64))
    dev = eos.rows.device
    tab = eos._table()
    T = torch.tensor(b["T"], dtype=f64, device=dev)
    r = native.gc_bubble_dew(tab, eos.S, eos.rows, torch.tensor(b["phi"], dtype=f64, device=dev), T, torch.tensor(b["x"], dtype=f64, device=dev),
                             torch.tensor(b["p_init"], dtype=f64, device=dev), False)
    ok = ~r["status"]
    rows, ph, Tk, rho4 = eos.rows[ok], torch.tensor(b["phi"], dtype=f64, device=dev)[ok], T[ok], r["rho4"][ok]
    g1 = native.gc_segment_gradient(tab, eos.S, rows, ph, Tk, rho4, False)
    ones = torch.ones(int(ok.sum()), dtype=f64, device=dev)
    g2 = native.gc_segment_gradient(tab, eos.S, rows, ph, Tk, rho4, False, gout=ones)
    scale = g1.abs().max(dim=0).values.clamp_min(1e-300)
    assert float(((g1 - g2).abs() / scale).max()) < 1e-12  # atomics: the order of the sums differs between launches
    g3 = native.gc_segment_gradient(tab, eos.S, rows, ph, Tk, rho4, False, gout=2.0 * ones)
    assert float(((g3 - 2.0 * g1).abs() / scale).max()) < 1e-12
    with pytest.raises(ValueError):
        native.gc_segment_gradient(tab, eos.S, rows, ph[:-1], Tk, rho4, False)
