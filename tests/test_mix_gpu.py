"""GPU parity tests for the binary-mixture path (config 4 of BASELINE.json), in the shape of
the reference's tests/test_pcsaft_mix.py.  Every call goes through the C ABI."""
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


@pytest.fixture(scope="module")
def gm():
    return load_golden("mix.json")


def _t(x, **kw):
    return torch.tensor(np.asarray(x), dtype=f64, **kw)


def test_derivatives_vs_reference_python(amd, gm):
    """tests/test_pcsaft_mix.py:16-124 — the 14 binary cases; reference tolerances abs 1e-14 / 1e-11."""
    g = gm["test_inputs"]
    eos = amd.PcSaftMix(_t(g["params"]), _t(g["kij"]))
    a, p, mu, v = eos.derivatives(_t(g["T"]), _t(g["rho"]))
    assert np.max(np.abs(a.numpy() - np.array(g["a"]))) < 1e-14
    assert np.max(np.abs(p.numpy() - np.array(g["p"]))) < 1e-14
    assert np.max(np.abs(mu.numpy() - np.array(g["mu"]))) < 1e-13  # |mu| ~ 10: 1e-14 relative
    assert np.max(np.abs(v.numpy() - np.array(g["v"]))) < 1e-11 * 50  # v up to 500
    assert a[10].item() == a[0].item()  # "np/x" == "np/np" (lone B-site component does not associate)
    hd = eos.helmholtz_energy_density(_t(g["T"]), _t(g["rho"]))
    assert hd.shape == (14, 1)


def test_derivatives_random_rows(amd, oracle, gm):
    """60 seeded rows of the config-4 distribution, row by row:
      * against the EXACT values of the model (long double, safeguarded association, cancellation-free site fractions) on
        every row: a, p, mu to 1e-13 of their scale, v to 1e-10;
      * against the reference's own fp64 output (the golden vectors, tests/test_pcsaft_mix.py:119-124 tolerances abs 1e-14 /
        1e-11) on every row where that output is itself good, i.e. where the golden agrees with the exact values to its
        tolerance; on the remaining rows (strongly associating: the X_B formula as written cancels, or the association
        Newton as written runs away) the reference's own error is printed and the kernel must be closer to the exact values
        than the reference is."""
    g = gm["random"]
    P, K, T, rho = (np.array(g[k]) for k in ("params", "kij", "T", "rho"))
    a, p, mu, v = (x.numpy() for x in amd.PcSaftMix(_t(P), _t(K)).derivatives(_t(T), _t(rho)))
    A, Pp, MU, V = oracle.mix_derivatives_exact(P, K, T, rho)
    sa = np.maximum(np.abs(A), 1e-6)
    assert np.max(np.abs(a - A) / sa) < 1e-13
    assert np.max(np.abs(p - Pp) / np.maximum(np.abs(Pp), rho.sum(axis=1))) < 1e-12
    assert np.max(np.abs(mu - MU) / np.maximum(1.0, np.abs(MU))) < 1e-13
    assert np.max(np.abs(v / V - 1.0)) < 1e-10
    ga, gp, gmu, gv = (np.array(g[k]) for k in ("a", "p", "mu", "v"))
    def row_err(xa, xp, xmu, xv):  # a, p, mu absolute (|mu| ~ 10); v relative (v ~ 1/rho spans 12 decades)
        return np.maximum.reduce([np.abs(xa - A), np.abs(xp - Pp), np.abs(xmu - MU).max(axis=1), 1e-3 * np.abs(xv / V - 1.0).max(axis=1)])

    ref_err = row_err(ga, gp, gmu, gv)
    good = np.isfinite(ref_err) & (ref_err < 1e-13)
    print(f"reference fp64 output good on {good.sum()} of {len(T)} rows; elsewhere its error is up to {np.nanmax(np.where(good, 0, ref_err)):.2e} "
          f"({(~np.isfinite(ref_err)).sum()} rows NaN)")
    assert good.sum() >= 55
    assert np.max(np.abs(a[good] - ga[good])) < 1e-14
    assert np.max(np.abs(p[good] - gp[good])) < 1e-14
    assert np.max(np.abs(mu[good] - gmu[good])) < 1e-13
    assert np.max(np.abs(v[good] / gv[good] - 1.0)) < 1e-10
    ours = row_err(a, p, mu, v)
    bad = ~good & np.isfinite(ref_err)
    assert np.all(ours[bad] <= ref_err[bad])


@pytest.mark.parametrize("key,dew", [("test_bubble", False), ("test_dew", True)])
def test_bubble_dew_reference_cases(amd, gm, key, dew):
    """tests/test_pcsaft_mix.py:127-192 / :195-251: value abs 1e-8 Pa, dp/dk_ij vs finite difference abs 1."""
    g = gm[key]
    ref = g["result"]
    kij = _t(g["kij"], requires_grad=True)
    par = _t(g["params"], requires_grad=True)
    T = _t(g["T"], requires_grad=True)
    z = _t(g["z"], requires_grad=True)
    p0 = _t(g["p_init"], requires_grad=True)
    eos = amd.PcSaftMix(par, kij)
    p, nans = (eos.dew_point if dew else eos.bubble_point)(T, z, p0)
    assert nans.tolist() == ref["nans"]
    assert np.max(np.abs(p.detach().numpy() - np.array(ref["value"]))) < 1e-8
    p[0].backward()
    fd = (p[1].item() - p[0].item()) / g["h"]
    assert abs(kij.grad[0, 0].item() - fd) < 1.0
    assert abs(kij.grad[0, 0].item() - ref["grad_kij"][0][0]) < 1e-3
    assert torch.all(kij.grad[1] == 0)
    # full gradient of row 0 against the reference's autograd
    got = np.concatenate([par.grad[0].numpy().ravel(), kij.grad[0].numpy(), [T.grad[0].item()]])
    p2, _ = (amd.PcSaftMix(_t(g["params"]), _t(g["kij"])).dew_point if dew else
             amd.PcSaftMix(_t(g["params"]), _t(g["kij"])).bubble_point)(_t(g["T"]), _t(g["z"]), _t(g["p_init"]))
    assert torch.equal(p2, p.detach())
    # golden gradients are of sum(p) -> per-row
    want = np.concatenate([np.array(ref["grad_params"])[0].ravel(), np.array(ref["grad_kij"])[0], [ref["grad_T"][0]]])
    assert np.max(np.abs(got - want)) / np.abs(want).max() < 1e-7


@pytest.mark.parametrize("dew", [False, True])
def test_random_batch_vs_oracle(amd, oracle, dew):
    from feos_torch_amd.synthetic import mix_batch

    n = 6000
    P, K, T, X, PI = mix_batch(n, seed=17)
    eos = amd.PcSaftMix(_t(P).cuda(), _t(K).cuda())
    p, nans = (eos.dew_point if dew else eos.bubble_point)(_t(T).cuda(), _t(X).cuda(), _t(PI).cuda())
    assert p.is_cuda and nans.shape == (n,)
    want, rho4, st = oracle.mix_bubble_dew(P, K, T, X, PI, dew, prec=1)
    nans = nans.cpu().numpy()
    assert (nans != st).mean() < 0.01, "failure masks differ on more than 1 % of the rows"
    assert nans.mean() < 0.03
    got = np.zeros(n)
    got[~nans] = p.cpu().numpy()
    both = ~nans & ~st
    assert np.max(np.abs(got[both] / want[both] - 1)) < 1e-9  # north_star tolerance
    assert eos.parameters.shape[0] == int((~nans).sum()) and eos.kij.shape[0] == int((~nans).sum())


def test_jacobian_vs_oracle(amd, oracle):
    from feos_torch_amd import native
    from feos_torch_amd.synthetic import mix_batch

    n = 1200
    P, K, T, X, PI = mix_batch(n, seed=23)
    for dew in (False, True):
        r = native.mix_bubble_dew(_t(P).cuda(), _t(K).cuda(), _t(T).cuda(), _t(X).cuda(), _t(PI).cuda(), dew)
        ok = ~r["status"].cpu().numpy()
        rho4 = r["rho4"].cpu().numpy()
        J = native.mix_jacobian(_t(P).cuda(), _t(K).cuda(), _t(T).cuda(), r["rho4"], dew).cpu().numpy()
        # exact gradient of the reference's formula (long double; the formulas as written cancel in fp64 on strongly
        # associating rows, which is what round 1's 1e-4 allowance was for)
        _, want = oracle.mix_bubble_dew_grad(P[ok], K[ok], T[ok], rho4[ok], dew, exact=True)
        scale = np.abs(want).max(axis=1, keepdims=True)
        err = np.abs(J[ok] - want) / scale
        print(f"{'dew' if dew else 'bubble'} Jacobian vs exact: max {err.max():.2e} q99 {np.quantile(err.max(axis=1), 0.99):.2e}")
        assert err.max() < 1e-8


def test_jacobian_class_order_is_only_a_schedule(amd):
    """pcs_mix_jacobian with a workspace takes the rows in batch-wide class order; without it they are bucketed inside
    the workgroup.  Same arithmetic per row: identical results."""
    from feos_torch_amd import _lib, native
    from feos_torch_amd.synthetic import mix_batch

    n = 30_000
    P, K, T, X, PI = mix_batch(n, seed=41)
    a = [_t(v).cuda() for v in (P, K, T, X, PI)]
    r = native.mix_bubble_dew(*a, False)
    rho4 = r["rho4"].clone()
    rho4[r["status"]] = torch.tensor([1e-6, 1e-6, 5e-3, 5e-3], dtype=torch.float64, device="cuda")
    with_ws = native.mix_jacobian(a[0], a[1], a[2], rho4, False)
    plain = torch.empty_like(with_ws)
    L = _lib.lib()
    _lib.check(L.pcs_mix_jacobian(0, _lib.ptr(a[0]), _lib.ptr(a[1]), _lib.ptr(a[2]), _lib.ptr(rho4), n, _lib.ptr(plain), None,
                                  _lib.current_stream_ptr(plain.device)), "pcs_mix_jacobian")
    torch.cuda.synchronize()
    ok = ~r["status"]
    assert torch.equal(with_ws[ok], plain[ok])


@pytest.mark.parametrize("dew", [False, True])
def test_phase_equilibrium_conditions_large_batch(amd, dew):
    """size-independent check at 1e6 rows (config 4): equal chemical potentials and pressures,
    the specified phase keeps the specified composition, vapour lighter than liquid."""
    from feos_torch_amd import native
    from feos_torch_amd.synthetic import mix_batch

    n = 1_000_000
    P, K, T, X, PI = mix_batch(n, seed=29)
    Pd, Kd, Td = _t(P).cuda(), _t(K).cuda(), _t(T).cuda()
    r = native.mix_bubble_dew(Pd, Kd, Td, _t(X).cuda(), _t(PI).cuda(), dew=dew)
    ok = ~r["status"]
    assert ok.float().mean().item() > 0.97
    rv, rl = r["rho4"][:, 0:2], r["rho4"][:, 2:4]
    aV, pV, muV, _ = native.mix_derivatives(Pd, Kd, Td, torch.where(ok[:, None], rv, torch.full_like(rv, 1e-3)))
    aL, pL, muL, vL = native.mix_derivatives(Pd, Kd, Td, torch.where(ok[:, None], rl, torch.full_like(rl, 1e-3)))
    dmu = (torch.log(rv) + muV - torch.log(rl) - muL)[ok]
    # a handful of ill-conditioned rows (near-azeotropic cross-associating liquids at 1e-10 Pa) leave through the
    # stagnation exit (Newton step < 1e-7 but no longer shrinking) with a larger residual
    worst = torch.abs(dmu).max(dim=1).values
    assert torch.max(worst).item() < 1e-3 and (worst > 1e-6).sum().item() <= 10
    assert torch.quantile(worst[:200000], 0.999).item() < 1e-10
    p_red = (r["p"] / (Td * 1.380649e-23 / 1e-30))[ok]
    dp = torch.abs(pV[ok] / p_red - 1)
    assert torch.max(dp).item() < 1e-4 and (dp > 1e-6).sum().item() <= 10  # same handful of stagnation-exit rows
    assert torch.quantile(dp[:200000], 0.999).item() < 1e-9
    spec = rv if dew else rl
    z1 = spec[:, 0] / spec.sum(dim=1)
    assert torch.max(torch.abs(z1[ok] - _t(X).cuda()[ok])).item() < 1e-12
    assert torch.all(rv.sum(dim=1)[ok] < rl.sum(dim=1)[ok])


@pytest.mark.parametrize("n", [1, 63, 64, 65, 129, 1000])
def test_ragged_sizes_are_consistent(amd, n):
    """The work queue hands rows to lanes in a run-dependent order; the results of a row must not depend on it
    nor on the batch size."""
    from feos_torch_amd import native
    from feos_torch_amd.synthetic import mix_batch

    P, K, T, X, PI = mix_batch(1000, seed=31)
    a = [_t(v).cuda() for v in (P, K, T, X, PI)]
    for dew in (False, True):
        full = native.mix_bubble_dew(*a, dew)
        part = native.mix_bubble_dew(*[v[:n].contiguous() for v in a], dew)
        assert torch.equal(part["status"], full["status"][:n])
        assert torch.equal(part["p"], full["p"][:n])
        assert torch.equal(part["rho4"], full["rho4"][:n])
        again = native.mix_bubble_dew(*a, dew)
        assert torch.equal(again["p"], full["p"]) and torch.equal(again["status"], full["status"])


def test_single_pass_schedule_without_workspace(amd):
    """pcs_mix_bubble_dew with workspace = NULL (one row per lane, no work queue) solves the same rows as the queue."""
    import ctypes

    from feos_torch_amd import _lib, native
    from feos_torch_amd.synthetic import mix_batch

    n = 2000
    a = [_t(v).cuda() for v in mix_batch(n, seed=37)]
    L = _lib.lib()
    for dew in (False, True):
        ref = native.mix_bubble_dew(*a, dew)
        p = torch.empty(n, dtype=f64, device="cuda")
        rho4 = torch.empty((n, 4), dtype=f64, device="cuda")
        st = torch.empty(n, dtype=torch.uint8, device="cuda")
        rc = L.pcs_mix_bubble_dew(int(dew), *[_lib.ptr(v) for v in a], n, _lib.ptr(p), _lib.ptr(rho4), _lib.ptr(st),
                                  None, None, _lib.current_stream_ptr(p.device))
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(st.bool(), ref["status"])
        # same solver code inlined into two different kernels: identical up to the compiler's FMA contraction
        assert torch.allclose(p, ref["p"], rtol=1e-11, atol=0.0) and torch.allclose(rho4, ref["rho4"], rtol=1e-10, atol=0.0)


def test_empty_and_ffi_mirror(amd, oracle):
    eos = amd.PcSaftMix(torch.zeros((0, 2, 8), dtype=f64), torch.zeros((0, 2), dtype=f64))
    p, nans = eos.bubble_point(torch.zeros(0, dtype=f64), torch.zeros(0, dtype=f64), torch.zeros(0, dtype=f64))
    assert p.shape == (0,) and nans.shape == (0,)
    # PcSaft.bubble_point keeps the Rust layout rho[n_ok,4] = (rhoV_1, rhoV_2, rhoL_1, rhoL_2) (src/pcsaft.rs:216-231)
    par = np.array([[[1, 3.5, 150, 0, 0, 0, 0, 0], [1, 3.5, 200, 0, 0, 0, 0, 0]]] * 3, dtype=float)
    kij = np.array([[-0.15, 0.0]] * 3)
    T = np.array([150.0, 1000.0, 150.0])  # row 1: far super-critical -> fails
    rho, status = amd.PcSaft.bubble_point(par, kij, T, np.full(3, 0.5), np.full(3, 1e5))
    assert status.tolist() == [False, True, False] and rho.shape == (2, 4)
    want, st = oracle.mix_bubble_dew_root(par, kij, T, np.full(3, 0.5), np.full(3, 1e5), dew=False)
    assert np.max(np.abs(rho / want[~st] - 1)) < 1e-7
    rho_d, status_d = amd.PcSaft.dew_point(par, kij, T, np.full(3, 0.5), np.full(3, 1e5))
    assert rho_d.shape == (2, 4) and status_d.tolist() == [False, True, False]


@pytest.mark.parametrize("dew", [False, True])
def test_bad_rows_fail_cleanly(amd, dew):
    """NaN / inf / non-physical inputs in some rows: the kernels terminate, flag exactly those rows (plus whatever has no
    solution anyway) and leave the results of the other rows untouched (rows are independent)."""
    from feos_torch_amd import native
    from feos_torch_amd.synthetic import mix_batch

    n = 4096
    P, K, T, X, PI = mix_batch(n, seed=61)
    ref = native.mix_bubble_dew(_t(P).cuda(), _t(K).cuda(), _t(T).cuda(), _t(X).cuda(), _t(PI).cuda(), dew)
    P2, K2, T2, X2, PI2 = P.copy(), K.copy(), T.copy(), X.copy(), PI.copy()
    bad = np.arange(0, n, 37)
    kinds = [lambda i: T2.__setitem__(i, np.nan), lambda i: T2.__setitem__(i, -10.0), lambda i: T2.__setitem__(i, np.inf),
             lambda i: X2.__setitem__(i, 0.0), lambda i: X2.__setitem__(i, 1.0), lambda i: X2.__setitem__(i, np.nan),
             lambda i: P2.__setitem__((i, 0, 0), np.nan), lambda i: P2.__setitem__((i, 1, 1), 0.0), lambda i: P2.__setitem__((i, 0, 2), -5.0),
             lambda i: PI2.__setitem__(i, np.inf), lambda i: PI2.__setitem__(i, -1.0), lambda i: K2.__setitem__((i, 0), np.nan),
             lambda i: T2.__setitem__(i, 1e-3), lambda i: T2.__setitem__(i, 1e6)]
    for j, i in enumerate(bad):
        kinds[j % len(kinds)](i)
    r = native.mix_bubble_dew(_t(P2).cuda(), _t(K2).cuda(), _t(T2).cuda(), _t(X2).cuda(), _t(PI2).cuda(), dew)
    torch.cuda.synchronize()
    good = np.ones(n, dtype=bool)
    good[bad] = False
    g = torch.from_numpy(good).cuda()
    assert torch.equal(r["status"][g], ref["status"][g])
    assert torch.equal(r["p"][g], ref["p"][g])
    ok = ~r["status"]
    assert bool(torch.isfinite(r["p"][ok]).all()) and bool((r["p"][ok] > 0).all())
    # the rows with NaN / inf / negative temperature or parameters can never converge
    hopeless = torch.from_numpy(np.isin(np.arange(n), bad[[j for j in range(len(bad)) if j % len(kinds) in (0, 1, 2, 5, 6, 11)]])).cuda()
    assert bool(r["status"][hopeless].all())


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 63, 64, 65, 129, 1000])
def test_jacobian_ragged_sizes_are_consistent(amd, n):
    """pcs_mix_jacobian (coefficient adjoints accumulated lane-strided in LDS, csrc/mix_adjoint.hpp): the gradient of a row
    must depend neither on the batch size nor on the lane / workgroup the class-ordered schedule hands it to."""
    from feos_torch_amd import native
    from feos_torch_amd.synthetic import mix_batch

    P, K, T, X, PI = mix_batch(1000, seed=37)
    a = [_t(v).cuda() for v in (P, K, T, X, PI)]
    for dew in (False, True):
        r = native.mix_bubble_dew(*a, dew)
        rho4 = r["rho4"].clone()
        rho4[r["status"]] = torch.tensor([1e-6, 1e-6, 5e-3, 5e-3], dtype=torch.float64, device="cuda")
        full = native.mix_jacobian(a[0], a[1], a[2], rho4, dew)
        part = native.mix_jacobian(a[0][:n].contiguous(), a[1][:n].contiguous(), a[2][:n].contiguous(), rho4[:n].contiguous(), dew)
        ok = ~r["status"][:n]
        assert torch.equal(part[ok], full[:n][ok])
        assert torch.isfinite(part[ok]).all()


@pytest.mark.gpu
def test_hipgraph_replay_of_solve_and_gradient_equals_eager(amd):
    """pcs_mix_bubble_dew (class sort + work-queue kernel) and pcs_mix_jacobian captured in one hipGraph and replayed behind
    pending work: bit-identical to the eager results (every operation of both calls is a kernel node)."""
    from feos_torch_amd import native
    from feos_torch_amd.synthetic import mix_batch

    n = 40_000
    P, K, T, X, PI = mix_batch(n, seed=43)
    dev = torch.device("cuda:0")
    a = [_t(v).to(dev) for v in (P, K, T, X, PI)]
    ref = native.mix_bubble_dew(*a, True)
    refj = native.mix_jacobian(a[0], a[1], a[2], ref["rho4"], True)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        w = native.mix_bubble_dew(*a, True)
        native.mix_jacobian(a[0], a[1], a[2], w["rho4"], True)
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        got = native.mix_bubble_dew(*a, True)
        gotj = native.mix_jacobian(a[0], a[1], a[2], got["rho4"], True)
    for rep in range(2):
        got["p"].fill_(float("nan"))
        gotj.fill_(float("nan"))
        for _ in range(3):  # pending work ahead of the replay
            native.mix_bubble_dew(*a, False)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(got["status"], ref["status"])
        ok = ~ref["status"]
        assert torch.equal(got["p"][ok], ref["p"][ok]) and torch.equal(got["rho4"][ok], ref["rho4"][ok])
        assert torch.equal(gotj[ok], refj[ok])
