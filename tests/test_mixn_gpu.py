"""n-component mixtures: `PcSaftMix(parameters[N, n, 8]).derivatives / helmholtz_energy_density` on the GPU.

What pins it.  The reference's n-component code path cannot be run (with kij = None, which n != 2 demands,
feos_torch/pcsaft_mix.py:75-76, helmholtz_energy_density fails at `self.kij[...]`, :141), so there are no reference outputs for
n > 2: "parity unpinned by stored values".  Instead:
  * n = 2, kij = 0, at most one associating component: outputs of the UNMODIFIED reference (tests/golden/mixn.json), reproduced by
    the n-component kernel at nc = 2 to the reference's tolerances (tests/test_pcsaft_mix.py:119-124) and by the binary kernel;
  * nc = 1 ... 6 on random rows: the long-double oracle restatement of the same lines (oracle/pcsaft_mixn.hpp), itself checked
    against those reference outputs (tests/test_oracle_mixn.py);
  * exact properties of the model for any n: splitting a component into two identical ones with the densities shared out
    changes nothing (a, p, mu_i, v_i); permuting the components permutes mu and v; nc = 1 is PcSaftPure."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
f64 = torch.float64


@pytest.fixture(scope="module")
def amd():
    assert torch.cuda.is_available()
    import feos_torch_amd

    return feos_torch_amd


def _t(x):
    return torch.tensor(np.asarray(x), dtype=f64)


def random_rows(n, nc, seed, assoc=True):
    rng = np.random.default_rng(seed)
    P = np.zeros((n, nc, 8))
    P[:, :, 0] = rng.uniform(1.0, 3.5, (n, nc))
    P[:, :, 1] = rng.uniform(2.8, 4.2, (n, nc))
    P[:, :, 2] = rng.uniform(150.0, 350.0, (n, nc))
    P[:, :, 3] = np.where(rng.random((n, nc)) < 0.4, rng.uniform(0.5, 3.0, (n, nc)), 0.0)
    if assoc:
        for r in range(0, n, 3):
            c = rng.integers(0, nc)
            P[r, c, 4:8] = [rng.uniform(0.001, 0.05), rng.uniform(1000.0, 3000.0), *[(1, 1), (2, 1), (1, 2)][rng.integers(0, 3)]]
    T = rng.uniform(200.0, 450.0, n)
    d = P[:, :, 1] * (1 - 0.12 * np.exp(-3 * P[:, :, 2] / T[:, None]))
    x = rng.dirichlet(np.ones(nc), n)
    eta = np.where(rng.random(n) < 0.5, rng.uniform(0.2, 0.42, n), 10.0 ** rng.uniform(-7, -2, n))
    rho = x * (eta / (np.pi / 6 * (x * P[:, :, 0] * d**3).sum(axis=1)))[:, None]
    return P, T, rho


def test_nc2_vs_reference_python(amd):
    g = load_golden("mixn.json")["nc2_kij0"]
    from feos_torch_amd import native

    P, T, rho = (np.array(g[k]) for k in ("params", "T", "rho"))
    a, p, mu, v = (x.cpu().numpy() for x in native.mixn_derivatives(_t(P).cuda(), _t(T).cuda(), _t(rho).cuda()))
    assert np.max(np.abs(a - np.array(g["a"]))) < 1e-14
    assert np.max(np.abs(p - np.array(g["p"]))) < 1e-14
    assert np.max(np.abs(mu - np.array(g["mu"]))) < 1e-13
    assert np.max(np.abs(v / np.array(g["v"]) - 1.0)) < 1e-11
    # and the binary kernel (k_ij = 0) gives the same
    a2, p2, mu2, v2 = amd.PcSaftMix(_t(P), torch.zeros((len(T), 2), dtype=f64)).derivatives(_t(T), _t(rho))
    assert np.max(np.abs(a2.numpy() - a)) < 1e-14 and np.max(np.abs(mu2.numpy() - mu)) < 1e-13
    assert np.max(np.abs(v2.numpy() / v - 1.0)) < 1e-11


@pytest.mark.parametrize("nc", [1, 2, 3, 4, 5, 6])
def test_random_rows_vs_oracle(amd, oracle, nc):
    P, T, rho = random_rows(3000, nc, seed=100 + nc)
    eos = amd.PcSaftMix(_t(P))
    a, p, mu, v = (x.numpy() for x in eos.derivatives(_t(T), _t(rho)))
    A, Pp, MU, V = oracle.mixn_derivatives(P, T, rho, prec=1)
    assert np.max(np.abs(a - A) / np.maximum(np.abs(A), 1e-6)) < 1e-12
    assert np.max(np.abs(p - Pp) / np.maximum(np.abs(Pp), rho.sum(axis=1))) < 1e-11
    assert np.max(np.abs(mu - MU) / np.maximum(1.0, np.abs(MU))) < 1e-12
    assert np.max(np.abs(v / V - 1.0)) < 1e-9
    h = eos.helmholtz_energy_density(_t(T), _t(rho))
    assert h.shape == (3000, 1) and np.array_equal(h[:, 0].numpy(), a)


def test_splitting_and_permutation(amd):
    P, T, rho = random_rows(2000, 3, seed=7, assoc=False)
    a, p, mu, v = (x.numpy() for x in amd.PcSaftMix(_t(P)).derivatives(_t(T), _t(rho)))
    # split component 2 into two identical components carrying 30 % / 70 % of its density: 4 components
    P4 = np.concatenate([P, P[:, 2:3, :]], axis=1)
    rho4 = np.concatenate([rho[:, :2], 0.3 * rho[:, 2:3], 0.7 * rho[:, 2:3]], axis=1)
    a4, p4, mu4, v4 = (x.numpy() for x in amd.PcSaftMix(_t(P4)).derivatives(_t(T), _t(rho4)))
    assert np.max(np.abs(a4 - a) / np.maximum(np.abs(a), 1e-9)) < 1e-12
    assert np.max(np.abs(p4 - p) / np.maximum(np.abs(p), rho.sum(axis=1))) < 1e-11
    assert np.max(np.abs(mu4[:, :3] - mu)) < 1e-11 and np.max(np.abs(mu4[:, 3] - mu4[:, 2])) < 1e-11
    assert np.max(np.abs(v4[:, :3] / v - 1.0)) < 1e-9
    # permutation of the components
    perm = [2, 0, 1]
    ap, pp, mup, vp = (x.numpy() for x in amd.PcSaftMix(_t(P[:, perm])).derivatives(_t(T), _t(rho[:, perm])))
    assert np.max(np.abs(ap - a) / np.maximum(np.abs(a), 1e-9)) < 1e-12
    assert np.max(np.abs(mup - mu[:, perm])) < 1e-11 and np.max(np.abs(vp / v[:, perm] - 1.0)) < 1e-9


def test_one_component_is_the_pure_model(amd):
    P, T, rho = random_rows(2000, 1, seed=3)
    a, p, mu, v = amd.PcSaftMix(_t(P)).derivatives(_t(T), _t(rho))
    a1, p1, dp1 = amd.PcSaftPure(_t(P[:, 0])).derivatives(_t(T), _t(rho[:, 0]))
    assert np.max(np.abs(a.numpy() - a1.numpy()) / np.maximum(np.abs(a1.numpy()), 1e-9)) < 1e-12
    assert np.max(np.abs(p.numpy() - p1.numpy()) / np.maximum(np.abs(p1.numpy()), rho[:, 0])) < 1e-11
    # one component: v = (1 + rho a'') / (rho + rho^2 a'') = 1 / rho, mu = a'
    assert np.max(np.abs(v[:, 0].numpy() * rho[:, 0] - 1.0)) < 1e-12
    assert np.max(np.abs((p.numpy() - rho[:, 0] + a.numpy()) / rho[:, 0] - mu[:, 0].numpy())) < 1e-8


def test_api_errors(amd):
    P, T, rho = random_rows(4, 3, seed=1, assoc=False)
    with pytest.raises(Exception, match="kij can only be used for binary mixtures"):
        amd.PcSaftMix(_t(P), torch.zeros((4, 2), dtype=f64))
    Pa = P.copy()
    Pa[0, 0, 4:8] = [0.01, 2000.0, 1, 1]
    Pa[0, 1, 4:8] = [0.01, 2000.0, 1, 1]
    with pytest.raises(Exception, match="associating components"):
        amd.PcSaftMix(_t(Pa))
    eos = amd.PcSaftMix(_t(P))
    with pytest.raises(Exception, match="binary"):
        eos.bubble_point(_t(T), _t([0.5] * 4), _t([1e5] * 4))
    with pytest.raises(ValueError):
        eos.derivatives(_t(T[:3]), _t(rho))
    with pytest.raises(ValueError):
        amd.PcSaftMix(_t(np.zeros((2, 7, 8))))


def test_vjp_nc2_equals_the_binary_backward(amd):
    """n = 2, k_ij = 0, at most one associating component: the n-component backward pass (pcs_mixn_derivatives_vjp) against the
    binary one (pcs_mix_derivatives_vjp), which tests/test_deriv_grad_gpu.py pins on the reference's own autograd."""
    from feos_torch_amd import native

    P, T, rho = random_rows(400, 2, seed=31)
    n = len(T)
    rng = np.random.default_rng(2)
    ga, gp, gmu, gv = rng.normal(size=n), rng.normal(size=n), rng.normal(size=(n, 2)), rng.normal(size=(n, 2)) * 1e-3
    d = lambda x: _t(x).cuda()
    g = native.mixn_derivatives_vjp(d(P), d(T), d(rho), d(ga), d(gp), d(gmu), d(gv)).cpu().numpy()
    gb = native.mix_derivatives_vjp(d(P), d(np.zeros((n, 2))), d(T), d(rho), d(ga), d(gp), d(gmu), d(gv)).cpu().numpy()
    want = np.concatenate([gb[:, 0:16], gb[:, 18:21]], axis=1)  # (16 parameters, T, rho_0, rho_1); k_ij columns dropped
    scale = np.abs(want).max(axis=1, keepdims=True)
    assert np.max(np.abs(g - want) / scale) < 1e-9


@pytest.mark.parametrize("nc", [1, 3, 4, 6])
def test_vjp_vs_finite_differences_of_the_oracle(amd, oracle, nc):
    """dL/d(parameters, T, rho) for nc components against central differences of L evaluated with the long-double oracle
    restatement (oracle/pcsaft_mixn.hpp)."""
    from feos_torch_amd import native

    P, T, rho = random_rows(24, nc, seed=40 + nc)
    n = len(T)
    rng = np.random.default_rng(nc)
    ga, gp, gmu, gv = rng.normal(size=n), rng.normal(size=n), rng.normal(size=(n, nc)), rng.normal(size=(n, nc)) * 1e-3
    d = lambda x: _t(x).cuda()
    g = native.mixn_derivatives_vjp(d(P), d(T), d(rho), d(ga), d(gp), d(gmu), d(gv)).cpu().numpy()

    def L(P_, T_, rho_):
        a, p, mu, v = oracle.mixn_derivatives(P_, T_, rho_, prec=1)
        return ga * a + gp * p + (gmu * mu).sum(axis=1) + (gv * v).sum(axis=1)

    fd = np.zeros_like(g)
    for k in range(8 * nc + 1 + nc):
        Pp, Pm, Tp, Tm, rp, rm = P.copy(), P.copy(), T.copy(), T.copy(), rho.copy(), rho.copy()
        if k < 8 * nc:
            c, q = divmod(k, 8)
            if q >= 6 or np.all(P[:, c, q] == 0.0):
                # site counts (class switches) and columns that are zero on every row (the relative step has nothing to scale
                # with; the reference's formula still has a derivative there, e.g. w.r.t. kappa_ab of a site-less component,
                # :211-218): not compared
                fd[:, k] = g[:, k]
                continue
            h = 1e-6 * np.where(P[:, c, q] != 0.0, np.abs(P[:, c, q]), 1.0)
            Pp[:, c, q] += h
            Pm[:, c, q] -= h
            mask = P[:, c, q] != 0.0
        elif k == 8 * nc:
            h = 1e-6 * T
            Tp, Tm = T + h, T - h
            mask = np.ones(n, bool)
        else:
            i = k - 8 * nc - 1
            h = 1e-6 * rho[:, i]
            rp[:, i] += h
            rm[:, i] -= h
            mask = np.ones(n, bool)
        fd[:, k] = np.where(mask, (L(Pp, Tp, rp) - L(Pm, Tm, rm)) / (2 * h), g[:, k])
    cols = [k for k in range(9 * nc + 1) if not (k < 8 * nc and k % 8 >= 6)]
    scale = np.abs(fd[:, cols]).max(axis=1, keepdims=True) + 1e-300
    err = np.abs(g[:, cols] - fd[:, cols]) / np.maximum(np.abs(fd[:, cols]), 1e-6 * scale)
    assert np.nanmax(err) < 2e-4, np.nanmax(err)


def test_shell_is_differentiable_for_three_components(amd):
    from feos_torch_amd import PcSaftMix

    P, T, rho = random_rows(50, 3, seed=77)
    Pt = _t(P).cuda().requires_grad_(True)
    Tt = _t(T).cuda().requires_grad_(True)
    rt = _t(rho).cuda().requires_grad_(True)
    a, p, mu, v = PcSaftMix(Pt).derivatives(Tt, rt)
    (a.sum() + p.sum() + mu.sum()).backward()
    assert Pt.grad.shape == (50, 3, 8) and torch.isfinite(Pt.grad).all() and torch.isfinite(Tt.grad).all() and torch.isfinite(rt.grad).all()
    # d a / d rho_i = mu_i (the residual chemical potential is the density derivative of the Helmholtz energy density)
    Pt2, rt2 = _t(P).cuda(), _t(rho).cuda().requires_grad_(True)
    a2, _, mu2, _ = PcSaftMix(Pt2).derivatives(_t(T).cuda(), rt2)
    a2.sum().backward()
    assert torch.allclose(rt2.grad, mu2.detach(), rtol=1e-10, atol=1e-12)
