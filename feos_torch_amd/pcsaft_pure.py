"""``PcSaftPure`` — drop-in for the reference class of the same name
(feos_torch/pcsaft_pure.py:89-243) backed by the gfx950 kernels.

Same constructor, method names, argument meaning, return-tuple order ``(nans, value)``, units,
row filtering (values only for converged rows, ``nans`` True = failed) and the reference's
quirk of MUTATING the model on every property call (``reduce``, :235-243).

What differs in mechanism, not in results:
  * the solve (Rust/feos in the reference) and the Python tail (:212-215 etc.) are one kernel;
  * gradients w.r.t. parameters / temperature / pressure come from a forward-mode Jacobian
    kernel evaluated at the converged densities instead of torch reverse mode through the
    tail — the same partial derivatives (the densities are detached in the reference too).
Tensors may live on any device; the computation runs on the current AMD GPU and results come
back on the device of ``parameters``.
"""
import torch

from . import native


class _PureProperty(torch.autograd.Function):
    """value[n_ok], nans[n] = property(parameters[n,8], temperature[n], pressure[n] or None)"""

    @staticmethod
    def forward(ctx, which, parameters, temperature, pressure):
        out_device = parameters.device
        dev = native._dev() if not parameters.is_cuda else parameters.device
        par = native._prep(parameters, dev, (8,))
        T = native._prep(temperature, dev)
        if which == "liquid_density":
            P = native._prep(pressure, dev)
            r = native.pure_liquid_density(par, T, P)
            value = r["rho"]
            rho_vl = torch.stack([torch.zeros_like(r["rho_root"]), r["rho_root"]], dim=1)
        else:
            P = None
            # the converged densities are only needed by the Jacobian kernel
            r = native.pure_vle(par, T, want_p=(which == "vapor_pressure"),
                                want_rho_eq=(which == "equilibrium_liquid_density"),
                                want_rho_vl=any(ctx.needs_input_grad[1:4]))
            value = r["p_sat"] if which == "vapor_pressure" else r["rho_eq"]
            rho_vl = r["rho_vl"]
        nans = r["status"]
        # every row converged (the common case): no compaction, no gathers
        all_ok = not bool(nans.any())
        ok = None if all_ok else ~nans
        if not all_ok:
            value = value[ok]
        needs = list(ctx.needs_input_grad[1:4])
        if any(needs):
            # Jacobian only on converged rows (dense kernel on the compacted inputs)
            if all_ok:
                jac = native.pure_jacobian(which, par, T, P, rho_vl)
                ctx.save_for_backward(jac)
            else:
                jac = native.pure_jacobian(which, par[ok], T[ok], None if P is None else P[ok], rho_vl[ok])
                ctx.save_for_backward(jac, ok)
        ctx.all_ok = all_ok
        ctx.needs = needs
        ctx.n = T.shape[0]
        ctx.out_device = out_device
        ctx.in_devices = (parameters.device, temperature.device, None if pressure is None else pressure.device)
        nans = nans.to(out_device)
        ctx.mark_non_differentiable(nans)
        return value.to(out_device), nans

    @staticmethod
    def backward(ctx, g_value, _g_nans):
        if ctx.all_ok:
            (jac,) = ctx.saved_tensors
            ok = None
        else:
            jac, ok = ctx.saved_tensors
        g = g_value.to(jac.device)
        n = ctx.n

        def scatter(x, tail):
            if ok is None:
                return x
            out = torch.zeros((n,) + tail, dtype=torch.float64, device=jac.device)
            out[ok] = x
            return out

        gp = gt = gpr = None
        if ctx.needs[0]:
            gp = scatter(g[:, None] * jac[:, 0:8], (8,)).to(ctx.in_devices[0])
        if ctx.needs[1]:
            gt = scatter(g * jac[:, 8], ()).to(ctx.in_devices[1])
        if ctx.needs[2]:
            gpr = scatter(g * jac[:, 9], ()).to(ctx.in_devices[2])
        return None, gp, gt, gpr


class _PureDerivatives(torch.autograd.Function):
    """(a, p, dp) = derivatives(parameters[n,8], temperature[n], density[n]) with the reference's autograd behaviour
    (feos_torch/pcsaft_pure.py:106-182 are torch graphs): gradients to parameters, temperature and density."""

    @staticmethod
    def forward(ctx, parameters, temperature, density):
        dev = native._dev() if not parameters.is_cuda else parameters.device
        par = native._prep(parameters, dev, (8,))
        T = native._prep(temperature, dev)
        rho = native._prep(density, dev)
        a, p, dp = native.pure_derivatives(par, T, rho)
        ctx.save_for_backward(par, T, rho)
        ctx.set_materialize_grads(False)
        ctx.in_devices = (parameters.device, temperature.device, density.device)
        out = parameters.device
        return a.to(out), p.to(out), dp.to(out)

    @staticmethod
    def backward(ctx, g_a, g_p, g_dp):
        par, T, rho = ctx.saved_tensors
        if g_a is None and g_p is None and g_dp is None:
            return None, None, None
        gpar, gT, grho = native.pure_derivatives_vjp(par, T, rho, g_a, g_p, g_dp)
        need = ctx.needs_input_grad
        return (gpar.to(ctx.in_devices[0]) if need[0] else None, gT.to(ctx.in_devices[1]) if need[1] else None,
                grho.to(ctx.in_devices[2]) if need[2] else None)


class PcSaftPure:
    def __init__(self, parameters):
        """parameters: [N, 8] float64 — m, sigma, epsilon_k, mu, kappa_ab, epsilon_k_ab, na, nb
        (feos_torch/pcsaft_pure.py:90-104, README.md:12)."""
        if parameters.dim() != 2 or parameters.shape[1] != 8:
            raise ValueError("parameters must have shape [N, 8]")
        self._set(parameters)

    def _set(self, parameters):
        self._par = parameters
        # attribute views kept for source compatibility with the reference (:91-104)
        self.m = parameters[:, 0]
        self.sigma = parameters[:, 1]
        self.epsilon_k = parameters[:, 2]
        self.mu2 = parameters[:, 3] ** 2 / (self.m * self.sigma**3 * self.epsilon_k) * 1e-19 * (1.0 / 1.380649e-23)
        self.kappa_ab = parameters[:, 4]
        self.epsilon_k_ab = parameters[:, 5]
        self.na = parameters[:, 6]
        self.nb = parameters[:, 7]

    @property
    def parameters(self):
        """numpy copy of the (current, possibly reduced) parameter rows (:104)."""
        return self._par.detach().cpu().numpy()

    # -- state functions -------------------------------------------------------------------
    def helmholtz_energy(self, temperature, density):
        """Reduced residual Helmholtz energy density a(T, rho) [A^-3] (:106-178); differentiable w.r.t. parameters,
        temperature and density like the reference's torch graph."""
        return self.derivatives(temperature, density)[0]

    def derivatives(self, temperature, density):
        """(a, p, dp/drho), all reduced (:180-182); differentiable (pcs_pure_derivatives_vjp is the backward pass)."""
        temperature = torch.as_tensor(temperature, dtype=torch.float64)
        density = torch.as_tensor(density, dtype=torch.float64)
        return _PureDerivatives.apply(self._par, temperature, density)

    # -- properties ------------------------------------------------------------------------
    def liquid_density(self, temperature, pressure):
        """(nans, rho [kmol/m3]) at (T [K], p [Pa]) (:184-199)."""
        value, nans = _PureProperty.apply("liquid_density", self._par, temperature, pressure)
        self.reduce(nans)
        return nans, value

    def vapor_pressure(self, temperature):
        """(nans, p_sat [Pa]) at T [K] (:201-215)."""
        value, nans = _PureProperty.apply("vapor_pressure", self._par, temperature, None)
        self.reduce(nans)
        return nans, value

    def equilibrium_liquid_density(self, temperature):
        """(nans, saturated liquid density [kmol/m3]) at T [K] (:217-233)."""
        value, nans = _PureProperty.apply("equilibrium_liquid_density", self._par, temperature, None)
        self.reduce(nans)
        return nans, value

    def reduce(self, nans):
        """Drop the rows flagged in ``nans`` from the model (:235-243)."""
        if bool(nans.any()):
            self._set(self._par[~nans.to(self._par.device)])
