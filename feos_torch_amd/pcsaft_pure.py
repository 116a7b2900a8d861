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
from torch.autograd.function import once_differentiable

from . import native


class _PureProperty(torch.autograd.Function):
    """value[n_ok], nans[n] = property(parameters[n,8], temperature[n], pressure[n] or None)

    One solve (dense outputs + status byte per row), one compaction plan whose 4-byte row count is the call's only host
    synchronisation, then -- only if rows were dropped -- single-kernel gathers (native.Compaction; the reference drops
    the rows inside its native call, src/pcsaft.rs:93-101).  The forward value does not depend on whether a gradient is
    requested: the vapour pressure always comes from the pressure-only kernel (pcs_pure_vapor_pressure)."""

    @staticmethod
    def forward(ctx, which, parameters, temperature, pressure, box):
        out_device = parameters.device
        dev = native._dev() if not parameters.is_cuda else parameters.device
        par = native._prep(parameters, dev, (8,))
        T = native._prep(temperature, dev)
        needs = list(ctx.needs_input_grad[1:4])
        if which == "liquid_density":
            P = native._prep(pressure, dev)
            r = native.pure_liquid_density(par, T, P)
            value = r["rho"]
            rho_vl = None
        elif which == "vapor_pressure":
            P = None
            r = native.pure_vapor_pressure(par, T, want_rho_vl=any(needs))
            value, rho_vl = r["p_sat"], r["rho_vl"]
        else:
            P = None
            r = native.pure_vle(par, T, want_p=False, want_rho_eq=True, want_rho_vl=any(needs))
            value, rho_vl = r["rho_eq"], r["rho_vl"]
        nans = r["status"]
        comp = native.Compaction(nans)
        box.append(comp)  # handed to the model for its `reduce`
        value = comp.gather(value)
        if any(needs):
            # Jacobian only on converged rows (dense kernel on the compacted inputs)
            if which == "liquid_density":
                root = comp.gather(r["rho_root"])
                rho_vl = torch.stack([torch.zeros_like(root), root], dim=1)
            else:
                rho_vl = comp.gather(rho_vl)
            if comp.all_ok:
                # every row converged (the common case): the Jacobian kernel runs in backward, in vector-Jacobian form, and
                # writes g * d value / d (parameters, T, p) straight into the gradient arrays
                ctx.save_for_backward(par, T, rho_vl) if P is None else ctx.save_for_backward(par, T, rho_vl, P)
            else:
                jac = native.pure_jacobian(which, comp.gather(par), comp.gather(T), None if P is None else comp.gather(P), rho_vl,
                                           polish=(which == "vapor_pressure"))
                ctx.save_for_backward(jac)
            ctx.comp = comp
            ctx.which = which
        ctx.needs = needs
        ctx.in_devices = (parameters.device, temperature.device, None if pressure is None else pressure.device)
        nans = nans.to(out_device)
        ctx.mark_non_differentiable(nans)
        return value.to(out_device), nans

    @staticmethod
    @once_differentiable
    def backward(ctx, g_value, _g_nans):
        comp = ctx.comp
        g = g_value.to(comp.device).contiguous()
        gp = gt = gpr = None
        if comp.all_ok:
            par, T, rho_vl, *rest = ctx.saved_tensors
            gp, gt, gpr = native.pure_jacobian_vjp(ctx.which, par, T, rest[0] if rest else None, rho_vl, g, ctx.needs,
                                                   polish=(ctx.which == "vapor_pressure"))
            dv = ctx.in_devices
            return (None, None if gp is None else gp.to(dv[0]), None if gt is None else gt.to(dv[1]),
                    None if gpr is None else gpr.to(dv[2]), None)
        (jac,) = ctx.saved_tensors
        # dense gradient rows in one kernel each: g_j * jac[j, cols] scattered to the row's place, zeros for dropped rows
        if ctx.needs[0]:
            gp = comp.expand(jac, g, 0, 8).to(ctx.in_devices[0])
        if ctx.needs[1]:
            gt = comp.expand(jac, g, 8, 1).view(comp.n).to(ctx.in_devices[1])
        if ctx.needs[2]:
            gpr = comp.expand(jac, g, 9, 1).view(comp.n).to(ctx.in_devices[2])
        return None, gp, gt, gpr, None


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
    @once_differentiable
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
        self._par = parameters

    # attribute views of the reference (:91-104), computed on access: the kernels read the [N,8] array itself, and
    # `mu2` alone is six strided passes over it (0.6 ms per 1e7 rows on every construction and every `reduce` if eager)
    m = property(lambda self: self._par[:, 0])
    sigma = property(lambda self: self._par[:, 1])
    epsilon_k = property(lambda self: self._par[:, 2])
    kappa_ab = property(lambda self: self._par[:, 4])
    epsilon_k_ab = property(lambda self: self._par[:, 5])
    na = property(lambda self: self._par[:, 6])
    nb = property(lambda self: self._par[:, 7])

    @property
    def mu2(self):
        """mu^2 / (m sigma^3 epsilon_k) * 1e-19 / k_B (:94-99)."""
        p = self._par
        return p[:, 3] ** 2 / (p[:, 0] * p[:, 1] ** 3 * p[:, 2]) * 1e-19 * (1.0 / 1.380649e-23)

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
    def _property(self, which, temperature, pressure):
        box = []
        value, nans = _PureProperty.apply(which, self._par, temperature, pressure, box)
        self._reduce(box[0])
        return nans, value

    def liquid_density(self, temperature, pressure):
        """(nans, rho [kmol/m3]) at (T [K], p [Pa]) (:184-199)."""
        return self._property("liquid_density", temperature, pressure)

    def vapor_pressure(self, temperature):
        """(nans, p_sat [Pa]) at T [K] (:201-215)."""
        return self._property("vapor_pressure", temperature, None)

    def equilibrium_liquid_density(self, temperature):
        """(nans, saturated liquid density [kmol/m3]) at T [K] (:217-233)."""
        return self._property("equilibrium_liquid_density", temperature, None)

    def _reduce(self, comp):
        if not comp.all_ok:
            self._par = native.compact_rows(comp, self._par)

    def reduce(self, nans):
        """Drop the rows flagged in ``nans`` from the model (:235-243)."""
        dev = self._par.device if self._par.is_cuda else native._dev()
        self._reduce(native.Compaction(nans.to(dev)))
