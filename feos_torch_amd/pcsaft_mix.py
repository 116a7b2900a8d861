"""``PcSaftMix`` — drop-in for the reference class of the same name
(feos_torch/pcsaft_mix.py:12-479) backed by the gfx950 kernels.

Same constructor ``PcSaftMix(parameters[N,2,8], kij[N,2])``, methods ``bubble_point`` /
``dew_point(temperature, molefracs, pressure)`` returning ``(pressure [Pa], nans)`` — note the
tuple order is the opposite of ``PcSaftPure`` in the reference too (:444, :468) — values only
for converged rows, and the model is reduced (mutated) by every property call (:470-479).
Gradients flow to ``parameters``, ``kij`` and ``temperature`` (the reference's final formula
does not depend on the mole fractions or the initial pressure explicitly, so those receive
zero gradient there; here they receive none).
"""
import torch
from torch.autograd.function import once_differentiable

from . import native


class _BubbleDew(torch.autograd.Function):
    """value[n_ok], nans[n] = bubble / dew pressure.  Dense solve, one compaction plan (its 4-byte row count is the call's only
    host synchronisation), single-kernel gathers only when rows were dropped (native.Compaction; the reference drops them
    inside the native call, src/pcsaft.rs:216-231)."""

    @staticmethod
    def forward(ctx, dew, parameters, kij, temperature, molefracs, pressure, box):
        out_device = parameters.device
        dev = native._dev() if not parameters.is_cuda else parameters.device
        par = native._prep(parameters, dev, (2, 8))
        k = native._prep(kij, dev, (2,))
        T = native._prep(temperature, dev)
        z = native._prep(molefracs, dev)
        p0 = native._prep(pressure, dev)
        r = native.mix_bubble_dew(par, k, T, z, p0, dew)
        nans = r["status"]
        comp = native.Compaction(nans)
        box.append(comp)
        value = comp.gather(r["p"])
        needs = list(ctx.needs_input_grad[1:4])
        if any(needs):
            jac = native.mix_jacobian(comp.gather(par), comp.gather(k), comp.gather(T), comp.gather(r["rho4"]), dew)
            ctx.save_for_backward(jac)
            ctx.comp = comp
        ctx.needs = needs
        ctx.in_devices = (parameters.device, kij.device, temperature.device)
        nans = nans.to(out_device)
        ctx.mark_non_differentiable(nans)
        return value.to(out_device), nans

    @staticmethod
    @once_differentiable
    def backward(ctx, g_value, _g_nans):
        (jac,) = ctx.saved_tensors
        comp = ctx.comp
        g = g_value.to(jac.device).contiguous()
        n = comp.n
        gp = gk = gt = None
        if ctx.needs[0]:
            gp = comp.expand(jac, g, 0, 16).view(n, 2, 8).to(ctx.in_devices[0])
        if ctx.needs[1]:
            gk = comp.expand(jac, g, 16, 2).to(ctx.in_devices[1])
        if ctx.needs[2]:
            gt = comp.expand(jac, g, 18, 1).view(n).to(ctx.in_devices[2])
        return None, gp, gk, gt, None, None, None


class _MixDerivatives(torch.autograd.Function):
    """(a, p, mu, v) = derivatives(parameters[n,2,8], kij[n,2], temperature[n], density[n,2]) with gradients to all four
    inputs (feos_torch/pcsaft_mix.py:31-154, :395-420 are torch graphs in the reference)."""

    @staticmethod
    def forward(ctx, parameters, kij, temperature, density):
        dev = native._dev() if not parameters.is_cuda else parameters.device
        par = native._prep(parameters, dev, (2, 8))
        k = native._prep(kij, dev, (2,))
        T = native._prep(temperature, dev)
        rho = native._prep(density, dev, (2,))
        a, p, mu, v = native.mix_derivatives(par, k, T, rho)
        ctx.save_for_backward(par, k, T, rho)
        ctx.set_materialize_grads(False)
        ctx.in_devices = (parameters.device, kij.device, temperature.device, density.device)
        out = parameters.device
        return a.to(out), p.to(out), mu.to(out), v.to(out)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_a, g_p, g_mu, g_v):
        par, k, T, rho = ctx.saved_tensors
        if g_a is None and g_p is None and g_mu is None and g_v is None:
            return None, None, None, None
        g = native.mix_derivatives_vjp(par, k, T, rho, g_a, g_p, g_mu, g_v)
        need, d = ctx.needs_input_grad, ctx.in_devices
        n = T.shape[0]
        return (g[:, 0:16].reshape(n, 2, 8).to(d[0]) if need[0] else None, g[:, 16:18].contiguous().to(d[1]) if need[1] else None,
                g[:, 18].contiguous().to(d[2]) if need[2] else None, g[:, 19:21].contiguous().to(d[3]) if need[3] else None)


class _MixnDerivatives(torch.autograd.Function):
    """n-component (a, p, mu, v) with gradients to parameters [n,nc,8], temperature and density [n,nc] (the reference's model is
    a torch graph for any number of components, feos_torch/pcsaft_mix.py:31-154, :395-420)."""

    @staticmethod
    def forward(ctx, parameters, temperature, density):
        dev = native._dev() if not parameters.is_cuda else parameters.device
        nc = int(parameters.shape[1])
        par = native._prep(parameters, dev, (nc, 8))
        T = native._prep(temperature, dev)
        rho = native._prep(density, dev, (nc,))
        a, p, mu, v = native.mixn_derivatives(par, T, rho)
        ctx.save_for_backward(par, T, rho)
        ctx.set_materialize_grads(False)
        ctx.in_devices = (parameters.device, temperature.device, density.device)
        out = parameters.device
        return a.to(out), p.to(out), mu.to(out), v.to(out)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_a, g_p, g_mu, g_v):
        par, T, rho = ctx.saved_tensors
        if g_a is None and g_p is None and g_mu is None and g_v is None:
            return None, None, None
        n, nc = rho.shape
        g = native.mixn_derivatives_vjp(par, T, rho, g_a, g_p, g_mu, g_v)
        need, d = ctx.needs_input_grad, ctx.in_devices
        return (g[:, : 8 * nc].reshape(n, nc, 8).to(d[0]) if need[0] else None, g[:, 8 * nc].contiguous().to(d[1]) if need[1] else None,
                g[:, 8 * nc + 1:].contiguous().to(d[2]) if need[2] else None)


class PcSaftMix:
    def __init__(self, parameters, kij=None):
        """parameters: [N, 2, 8] float64 (component rows as for PcSaftPure); kij: [N, 2] with
        kij[:,0] = k_ij and kij[:,1] = explicit cross-association energy eps_AiBj/k or 0
        (feos_torch/pcsaft_mix.py:13-29; effectively mandatory in the reference, :141/:477)."""
        if parameters.dim() != 3 or parameters.shape[2] != 8:
            raise ValueError("parameters must have shape [N, n, 8]")
        self.ncomp = int(parameters.shape[1])
        if self.ncomp != 2:
            # n-component mixtures (the reference's hs / hc / dispersion / dipole / self-association code is general, :31-154):
            # state functions only, no k_ij (":75-76 kij can only be used for binary mixtures!"), bubble / dew points are binary
            if kij is not None:
                raise Exception("kij can only be used for binary mixtures!")
            if not 1 <= self.ncomp <= 6:
                raise ValueError("between 1 and 6 components are supported")
            if bool((((parameters[:, :, 6] + parameters[:, :, 7]) != 0).sum(dim=1) > 1).any()):
                raise Exception("Only up to two associating components are allowed, and two only for binary mixtures!")
        elif kij is None:
            kij = torch.zeros((parameters.shape[0], 2), dtype=parameters.dtype, device=parameters.device)
        self._set(parameters, kij)

    def _set(self, parameters, kij):
        self._par = parameters
        self.kij = kij

    # attribute views of the reference (:14-29), computed on access (the kernels read the [N,2,8] array itself)
    m = property(lambda self: self._par[:, :, 0])
    sigma = property(lambda self: self._par[:, :, 1])
    epsilon_k = property(lambda self: self._par[:, :, 2])
    kappa_ab = property(lambda self: self._par[:, :, 4])
    epsilon_k_ab = property(lambda self: self._par[:, :, 5])
    na = property(lambda self: self._par[:, :, 6])
    nb = property(lambda self: self._par[:, :, 7])

    @property
    def mu2(self):
        p = self._par
        return p[:, :, 3] ** 2 / (p[:, :, 0] * p[:, :, 1] ** 3 * p[:, :, 2]) * 1e-19 * (1.0 / 1.380649e-23)

    @property
    def parameters(self):
        return self._par.detach().cpu().numpy()

    @property
    def kij_np(self):
        return None if self.kij is None else self.kij.detach().cpu().numpy()

    def helmholtz_energy_density(self, temperature, density):
        """a(T, rho_1, rho_2) [A^-3], shape [N, 1] like the reference (:31-154); differentiable."""
        return self.derivatives(temperature, density)[0][:, None]

    def derivatives(self, temperature, density):
        """(a [N], p [N], mu [N,n], v [N,n]) (:395-420); differentiable w.r.t. parameters, kij, temperature and density
        (pcs_mix_derivatives_vjp / pcs_mixn_derivatives_vjp are the backward passes)."""
        temperature = torch.as_tensor(temperature, dtype=torch.float64)
        density = torch.as_tensor(density, dtype=torch.float64)
        if self.ncomp != 2:
            return _MixnDerivatives.apply(self._par, temperature, density)
        return _MixDerivatives.apply(self._par, self.kij, temperature, density)

    def _bubble_dew(self, dew, temperature, molefracs, pressure):
        if self.ncomp != 2:
            raise Exception("bubble and dew points are implemented for binary mixtures (src/pcsaft.rs:43-79 takes [N,2,8])")
        box = []
        # mole fractions and initial pressure do not enter the reference's final formula (:435-444): no gradient flows to them
        value, nans = _BubbleDew.apply(dew, self._par, self.kij, temperature, _detached(molefracs), _detached(pressure), box)
        self._reduce(box[0])
        return value, nans

    def bubble_point(self, temperature, liquid_molefracs, pressure):
        """(p [Pa], nans) at T [K], liquid mole fraction of component 1, initial pressure [Pa] (:422-444)."""
        return self._bubble_dew(False, temperature, liquid_molefracs, pressure)

    def dew_point(self, temperature, vapor_molefracs, pressure):
        """(p [Pa], nans) at T [K], vapour mole fraction of component 1, initial pressure [Pa] (:446-468)."""
        return self._bubble_dew(True, temperature, vapor_molefracs, pressure)

    def _reduce(self, comp):
        if not comp.all_ok:
            self._set(native.compact_rows(comp, self._par), None if self.kij is None else native.compact_rows(comp, self.kij))

    def reduce(self, nans):
        """Drop the rows flagged in ``nans`` (:470-479)."""
        dev = self._par.device if self._par.is_cuda else native._dev()
        self._reduce(native.Compaction(nans.to(dev)))


def _detached(x):
    return x.detach() if isinstance(x, torch.Tensor) else x
