"""Device-level wrappers over the C ABI (``include/pcsaft_hip.h``) and the host-side mirror of
the reference's PyO3 class ``PcSaft`` (src/pcsaft.rs:13-80).

Two layers:
  * ``pure_vle`` / ``pure_liquid_density`` / ... : torch CUDA(=HIP) tensors in, dense torch
    tensors out (status mask instead of dropped rows).  Used by the model classes.
  * ``PcSaft``: numpy in / numpy out with exactly the reference extension's signatures and
    output layout (failed rows dropped, ``status`` True = failed), so code written against
    ``feos_torch.feos_torch.PcSaft`` runs unchanged.
"""
import numpy as np
import torch
from torch.autograd.function import once_differentiable

from . import _lib

_F64 = torch.float64


def _dev(device=None):
    if not torch.cuda.is_available():
        raise _lib.PcsError(
            "feos_torch_amd needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback."
        )
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def _prep(x, device, shape_tail=None):
    """float64, contiguous, on `device`, detached."""
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x, dtype=np.float64))
    x = x.detach()
    if x.dtype != _F64:
        raise TypeError(f"expected float64, got {x.dtype}")  # reference: PyReadonlyArray<f64> type error
    x = x.to(device).contiguous()
    if shape_tail is not None and tuple(x.shape[1:]) != tuple(shape_tail):
        raise ValueError(f"expected trailing shape {shape_tail}, got {tuple(x.shape)}")
    return x


def _same_rows(n, **named):
    """Every row-wise argument must have n rows: the kernels launch n lanes over all of them (the reference raises a
    shape error in the same situation, e.g. a model reduced by an earlier property call used with the original T)."""
    for name, t in named.items():
        if t is not None and t.shape[0] != n:
            raise ValueError(f"{name} has {t.shape[0]} rows, expected {n}")


def pure_vle(params, temperature, want_p=True, want_rho_eq=False, want_iters=False, want_rho_vl=True, all_fp64=False):
    """Pure VLE on the GPU.  -> dict(p_sat [Pa], rho_eq [kmol/m3], rho_vl [n,2] A^-3, status bool, iters).
    want_rho_vl=False with want_rho_eq=False selects the pressure-only kernel (fp64 finish with the fp32
    pre-solve's dp/drho; densities not returned); rho_vl: the all-fp64 kernel; rho_eq: pressure-only kernel + one exact fp64
    Newton update of the densities.
    all_fp64: the validation twin (pcs_pure_vle_fp64: fp64 second derivatives in every iteration)."""
    device = params.device if isinstance(params, torch.Tensor) and params.is_cuda else _dev()
    params = _prep(params, device, (8,))
    temperature = _prep(temperature, device)
    n = temperature.shape[0]
    if params.shape[0] != n:
        raise ValueError("parameters and temperature differ in length")
    L = _lib.lib()
    with torch.cuda.device(device):
        p_sat = torch.empty(n, dtype=_F64, device=device) if want_p else None
        rho_eq = torch.empty(n, dtype=_F64, device=device) if want_rho_eq else None
        rho_vl = torch.empty((n, 2), dtype=_F64, device=device) if want_rho_vl else None
        status = torch.empty(n, dtype=torch.uint8, device=device)
        iters = torch.empty(n, dtype=torch.int32, device=device) if want_iters else None
        ws = torch.empty(max(1, L.pcs_workspace_bytes(n) // 4), dtype=torch.int32, device=device)
        fn = L.pcs_pure_vle_fp64 if all_fp64 else L.pcs_pure_vle
        rc = fn(_lib.ptr(params), _lib.ptr(temperature), n, _lib.ptr(p_sat), _lib.ptr(rho_eq),
                _lib.ptr(rho_vl), _lib.ptr(status), _lib.ptr(iters), _lib.ptr(ws),
                _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_pure_vle")
    return {"p_sat": p_sat, "rho_eq": rho_eq, "rho_vl": rho_vl, "status": status.view(torch.bool), "iters": iters}


def pure_vapor_pressure(params, temperature, want_rho_vl=False):
    """PcSaftPure.vapor_pressure in one call (pcs_pure_vapor_pressure): always the pressure-only kernel, so p_sat has the
    same bits with and without the densities.  -> dict(p_sat [Pa], rho_vl [n,2] A^-3 or None, status bool)."""
    device = params.device if isinstance(params, torch.Tensor) and params.is_cuda else _dev()
    params = _prep(params, device, (8,))
    temperature = _prep(temperature, device)
    n = temperature.shape[0]
    _same_rows(n, parameters=params)
    L = _lib.lib()
    with torch.cuda.device(device):
        p_sat = torch.empty(n, dtype=_F64, device=device)
        rho_vl = torch.empty((n, 2), dtype=_F64, device=device) if want_rho_vl else None
        status = torch.empty(n, dtype=torch.uint8, device=device)
        ws = torch.empty(max(1, L.pcs_workspace_bytes(n) // 4), dtype=torch.int32, device=device)
        rc = L.pcs_pure_vapor_pressure(_lib.ptr(params), _lib.ptr(temperature), n, _lib.ptr(p_sat), _lib.ptr(rho_vl),
                                       _lib.ptr(status), _lib.ptr(ws), _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_pure_vapor_pressure")
    return {"p_sat": p_sat, "rho_eq": None, "rho_vl": rho_vl, "status": status.view(torch.bool), "iters": None}


class Compaction:
    """Plan of one status mask (K8, csrc/compact_kernels.hip): which rows a property call keeps, in order.  Building it costs
    two small kernels and ONE 4-byte read-back (`n_ok`) -- the only host synchronisation of a property call; `gather` /
    `expand` are single kernels (no boolean-index gathers, no nonzero)."""

    def __init__(self, status):
        """status: bool or uint8 [n] on the GPU, True / 1 = dropped."""
        if status.dtype == torch.bool:
            status = status.view(torch.uint8)
        if status.dtype != torch.uint8 or status.dim() != 1 or not status.is_cuda:
            raise ValueError("status must be a bool / uint8 vector on the GPU")
        self.status = status.contiguous()
        self.n = int(status.shape[0])
        self.device = status.device
        L = _lib.lib()
        with torch.cuda.device(self.device):
            self.cws = torch.empty(max(1, L.pcs_compact_workspace_bytes(self.n) // 4), dtype=torch.int32, device=self.device)
            _lib.check(L.pcs_compact_plan(_lib.ptr(self.status), self.n, _lib.ptr(self.cws),
                                          _lib.current_stream_ptr(self.device)), "pcs_compact_plan")
        self.n_ok = int(self.cws[0].item()) if self.n else 0
        self.all_ok = self.n_ok == self.n

    def gather(self, x):
        """Rows of x ([n] or [n, ...] float64, or uint8 rows of a multiple of 8 bytes) that are kept, in order."""
        if self.all_ok:
            return x
        if x.shape[0] != self.n:
            raise ValueError(f"tensor has {x.shape[0]} rows, the mask {self.n}")
        x = x.contiguous()
        tail = tuple(x.shape[1:])
        if x.dtype == torch.uint8:
            flat = x.view(self.n, -1)
            if flat.shape[1] % 8:
                raise ValueError("uint8 rows must be a multiple of 8 bytes")
            return self.gather(flat.view(_F64)).view(torch.uint8).view((self.n_ok,) + tail)
        if x.dtype != _F64:
            raise TypeError(f"expected float64, got {x.dtype}")
        width = 1
        for d in tail:
            width *= int(d)
        out = torch.empty((self.n_ok,) + tail, dtype=_F64, device=self.device)
        if self.n_ok and width:
            L = _lib.lib()
            with torch.cuda.device(self.device):
                _lib.check(L.pcs_compact_rows(_lib.ptr(self.status), self.n, _lib.ptr(self.cws), _lib.ptr(x), width,
                                              _lib.ptr(out), None, _lib.current_stream_ptr(self.device)), "pcs_compact_rows")
        return out

    def index(self):
        """int32 [n_ok]: original row of every kept row."""
        out = torch.empty(self.n_ok, dtype=torch.int32, device=self.device)
        if self.n_ok:
            L = _lib.lib()
            with torch.cuda.device(self.device):
                _lib.check(L.pcs_compact_rows(_lib.ptr(self.status), self.n, _lib.ptr(self.cws), None, 1, None, _lib.ptr(out),
                                              _lib.current_stream_ptr(self.device)), "pcs_compact_rows")
        return out

    def expand(self, src, g=None, col0=0, ncol=None):
        """Dense [n, ncol]: g[j] * src[j, col0:col0+ncol] in the row of the j-th kept entry, 0 in dropped rows (the scatter of
        a backward pass fused with its Jacobian product).  src [n_ok] or [n_ok, k] float64; 1-D src gives a 1-D result."""
        one_d = src.dim() == 1
        stride = 1
        for d in src.shape[1:]:
            stride *= int(d)
        src2 = src.contiguous().view(src.shape[0], stride)
        ncol = stride - col0 if ncol is None else ncol
        if src2.shape[0] != self.n_ok or (g is not None and g.shape[0] != self.n_ok):
            raise ValueError("src / g must have one row per kept row")
        if self.n_ok == 0:  # nothing kept: all zeros (an empty tensor has no data pointer to hand to the kernel)
            out = torch.zeros((self.n, ncol), dtype=_F64, device=self.device)
            return out.view(self.n) if (one_d and ncol == 1) else out
        out = torch.empty((self.n, ncol), dtype=_F64, device=self.device)
        if self.n:
            L = _lib.lib()
            with torch.cuda.device(self.device):
                _lib.check(L.pcs_expand_rows(None if self.all_ok else _lib.ptr(self.status), self.n, _lib.ptr(self.cws),
                                             None if g is None else _lib.ptr(g.contiguous()), _lib.ptr(src2), stride, col0, ncol,
                                             _lib.ptr(out), _lib.current_stream_ptr(self.device)), "pcs_expand_rows")
        return out.view(self.n) if (one_d and ncol == 1) else out


class _CompactRows(torch.autograd.Function):
    """Differentiable row filter of a model (`reduce`): keeps the graph to the caller's parameter tensor like the reference's
    boolean indexing (feos_torch/pcsaft_pure.py:235-243) without its nonzero / index kernels."""

    @staticmethod
    def forward(ctx, comp, x):
        ctx.comp = comp
        ctx.in_device = x.device
        ctx.shape = tuple(x.shape)
        return comp.gather(x.detach().to(comp.device)).to(x.device)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        comp = ctx.comp
        g = g.to(comp.device).contiguous()
        out = comp.expand(g.view(g.shape[0], -1) if g.dim() > 1 else g)
        return None, out.view(ctx.shape).to(ctx.in_device)


def compact_rows(comp, x):
    """x[kept rows] with autograd (identity when every row is kept)."""
    return x if comp.all_ok else _CompactRows.apply(comp, x)


def pure_liquid_density(params, temperature, pressure):
    """-> dict(rho [kmol/m3], rho_root [A^-3], status bool)"""
    device = params.device if isinstance(params, torch.Tensor) and params.is_cuda else _dev()
    params = _prep(params, device, (8,))
    temperature = _prep(temperature, device)
    pressure = _prep(pressure, device)
    n = temperature.shape[0]
    if params.shape[0] != n or pressure.shape[0] != n:
        raise ValueError("parameters, temperature and pressure differ in length")
    L = _lib.lib()
    with torch.cuda.device(device):
        rho = torch.empty(n, dtype=_F64, device=device)
        root = torch.empty(n, dtype=_F64, device=device)
        status = torch.empty(n, dtype=torch.uint8, device=device)
        rc = L.pcs_pure_liquid_density(_lib.ptr(params), _lib.ptr(temperature), _lib.ptr(pressure), n,
                                       _lib.ptr(rho), _lib.ptr(root), _lib.ptr(status),
                                       _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_pure_liquid_density")
    return {"rho": rho, "rho_root": root, "status": status.view(torch.bool)}


def pure_derivatives(params, temperature, density):
    """(a, p, dp) reduced — PcSaftPure.derivatives (feos_torch/pcsaft_pure.py:180-182)."""
    device = params.device if isinstance(params, torch.Tensor) and params.is_cuda else _dev()
    params = _prep(params, device, (8,))
    temperature = _prep(temperature, device)
    density = _prep(density, device)
    n = temperature.shape[0]
    _same_rows(n, parameters=params, density=density)
    L = _lib.lib()
    with torch.cuda.device(device):
        a = torch.empty(n, dtype=_F64, device=device)
        p = torch.empty(n, dtype=_F64, device=device)
        dp = torch.empty(n, dtype=_F64, device=device)
        rc = L.pcs_pure_derivatives(_lib.ptr(params), _lib.ptr(temperature), _lib.ptr(density), n, _lib.ptr(a),
                                    _lib.ptr(p), _lib.ptr(dp), _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_pure_derivatives")
    return a, p, dp


_WHICH = {"vapor_pressure": 0, "liquid_density": 1, "equilibrium_liquid_density": 2}


JAC_POLISH = 0x100  # PCS_JAC_POLISH (include/pcsaft_hip.h)


def pure_jacobian(which, params, temperature, pressure, rho_vl, polish=False):
    """[n,10] Jacobian w.r.t. (8 parameters, T, p) at fixed densities.  polish: rho_vl are the pressure-only kernel's
    densities (pure_vapor_pressure) and take one fp64 Newton step first."""
    device = rho_vl.device
    params = _prep(params, device, (8,))
    temperature = _prep(temperature, device)
    pressure = None if pressure is None else _prep(pressure, device)
    rho_vl = _prep(rho_vl, device, (2,))
    n = temperature.shape[0]
    _same_rows(n, parameters=params, pressure=pressure, rho_vl=rho_vl)
    L = _lib.lib()
    with torch.cuda.device(device):
        jac = torch.empty((n, 10), dtype=_F64, device=device)
        rc = L.pcs_pure_jacobian(_WHICH[which] | (JAC_POLISH if polish else 0), _lib.ptr(params), _lib.ptr(temperature), _lib.ptr(pressure),
                                 _lib.ptr(rho_vl), n, _lib.ptr(jac), _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_pure_jacobian")
    return jac


def pure_jacobian_vjp(which, params, temperature, pressure, rho_vl, gout, need=(True, True, True), polish=False):
    """Backward pass of a pure-component property on rows that all converged: (grad_params [n,8], grad_T [n], grad_p [n]) =
    gout[:, None] * Jacobian, produced by the Jacobian kernel itself (pcs_pure_jacobian_vjp)."""
    device = rho_vl.device
    params = _prep(params, device, (8,))
    temperature = _prep(temperature, device)
    pressure = None if pressure is None else _prep(pressure, device)
    rho_vl = _prep(rho_vl, device, (2,))
    gout = _prep(gout, device)
    n = temperature.shape[0]
    _same_rows(n, parameters=params, pressure=pressure, rho_vl=rho_vl, gout=gout)
    L = _lib.lib()
    with torch.cuda.device(device):
        gp = torch.empty((n, 8), dtype=_F64, device=device) if need[0] else None
        gt = torch.empty(n, dtype=_F64, device=device) if need[1] else None
        gpr = torch.empty(n, dtype=_F64, device=device) if (need[2] and pressure is not None) else None
        rc = L.pcs_pure_jacobian_vjp(_WHICH[which] | (JAC_POLISH if polish else 0), _lib.ptr(params), _lib.ptr(temperature), _lib.ptr(pressure), _lib.ptr(rho_vl),
                                     _lib.ptr(gout), n, _lib.ptr(gp), _lib.ptr(gt), _lib.ptr(gpr), _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_pure_jacobian_vjp")
    return gp, gt, gpr


class PcSaft:
    """Mirror of the reference's Rust pyclass ``PcSaft`` (src/pcsaft.rs:13-80): static methods,
    float64 numpy arrays in, ``(rho, status)`` numpy arrays out, failed rows dropped from ``rho``."""

    @staticmethod
    def vapor_pressure(parameters, temperature):
        """src/pcsaft.rs:18-26 — rho[n_ok, 4]: col 0 = rho_V, col 1 = rho_L, cols 2-3 zero (:94-101)."""
        parameters = _as_f64(parameters, 2)
        temperature = _as_f64(temperature, 1)
        r = pure_vle(torch.from_numpy(parameters), torch.from_numpy(temperature), want_p=False)
        comp = Compaction(r["status"])  # failed rows dropped on the device (:93-101)
        rho = np.zeros((comp.n_ok, 4))
        rho[:, 0:2] = comp.gather(r["rho_vl"]).cpu().numpy()
        return rho, r["status"].cpu().numpy()

    @staticmethod
    def liquid_density(parameters, temperature, pressure):
        """src/pcsaft.rs:28-41 — rho[n_ok]."""
        parameters = _as_f64(parameters, 2)
        temperature = _as_f64(temperature, 1)
        pressure = _as_f64(pressure, 1)
        r = pure_liquid_density(torch.from_numpy(parameters), torch.from_numpy(temperature),
                                torch.from_numpy(pressure))
        return Compaction(r["status"]).gather(r["rho_root"]).cpu().numpy(), r["status"].cpu().numpy()


def _as_f64(x, ndim):
    x = np.asarray(x)
    if x.dtype != np.float64:
        raise TypeError(f"argument must be a float64 array, got {x.dtype}")  # PyReadonlyArray<f64>
    if x.ndim != ndim:
        raise TypeError(f"argument must be {ndim}-dimensional, got {x.ndim}")
    return np.ascontiguousarray(x)


class PureVlePlan:
    """Pre-allocated launch plan for repeated pure-VLE solves on a fixed number of rows: all
    outputs and the retry workspace are allocated once; ``run`` only enqueues kernels on the
    current HIP stream (no allocation, no host synchronisation), so steps can be timed with
    HIP events."""

    def __init__(self, n, device, want_rho_eq=False, want_rho_vl=False, all_fp64=False):
        self.all_fp64 = bool(all_fp64)
        self.n = int(n)
        self.device = torch.device(device)
        self._L = _lib.lib()
        with torch.cuda.device(self.device):
            self.p_sat = torch.empty(self.n, dtype=_F64, device=self.device)
            self.rho_eq = torch.empty(self.n, dtype=_F64, device=self.device) if want_rho_eq else None
            self.rho_vl = torch.empty((self.n, 2), dtype=_F64, device=self.device) if want_rho_vl else None
            self.status = torch.empty(self.n, dtype=torch.uint8, device=self.device)
            self.ws = torch.empty(max(1, self._L.pcs_workspace_bytes(self.n) // 4), dtype=torch.int32,
                                  device=self.device)

    def _args(self, params, temperature):
        return (_lib.ptr(params), _lib.ptr(temperature), self.n, _lib.ptr(self.p_sat), _lib.ptr(self.rho_eq),
                _lib.ptr(self.rho_vl), _lib.ptr(self.status), None, _lib.ptr(self.ws),
                _lib.current_stream_ptr(self.device))

    def run(self, params, temperature):
        fn = self._L.pcs_pure_vle_fp64 if self.all_fp64 else self._L.pcs_pure_vle
        _lib.check(fn(*self._args(params, temperature)), "pcs_pure_vle")

    def run_fast(self, params, temperature):
        _lib.check(self._L.pcs_pure_vle_fast(*self._args(params, temperature)), "pcs_pure_vle_fast")

    def run_retry(self, params, temperature):
        _lib.check(self._L.pcs_pure_vle_retry(*self._args(params, temperature)), "pcs_pure_vle_retry")

    def retry_count(self):
        """Rows of the last run that left the main kernel: (all-fp64 fallback rows, robust-pass rows)."""
        cnt = int(self.ws[0].item())
        entries = self.ws[1:1 + cnt]
        fallback = int((entries < 0).sum().item())  # bit 31 set
        return fallback, cnt - fallback


# ------------------------------------------------------------------------------------------
# binary mixtures
# ------------------------------------------------------------------------------------------
def mix_bubble_dew(params, kij, temperature, molefracs, pressure, dew, want_iters=False):
    """Bubble (dew=False) / dew (dew=True) points.  -> dict(p [Pa], rho4 [n,4] A^-3 = (rhoV_1, rhoV_2,
    rhoL_1, rhoL_2), status bool, iters)."""
    device = params.device if isinstance(params, torch.Tensor) and params.is_cuda else _dev()
    params = _prep(params, device, (2, 8))
    kij = _prep(kij, device, (2,))
    temperature = _prep(temperature, device)
    molefracs = _prep(molefracs, device)
    pressure = _prep(pressure, device)
    n = temperature.shape[0]
    if not (params.shape[0] == kij.shape[0] == molefracs.shape[0] == pressure.shape[0] == n):
        raise ValueError("inputs differ in length")
    L = _lib.lib()
    with torch.cuda.device(device):
        p = torch.empty(n, dtype=_F64, device=device)
        rho4 = torch.empty((n, 4), dtype=_F64, device=device)
        status = torch.empty(n, dtype=torch.uint8, device=device)
        iters = torch.empty(n, dtype=torch.int32, device=device) if want_iters else None
        ws = torch.empty(max(1, L.pcs_mix_workspace_bytes(n) // 4), dtype=torch.int32, device=device)
        rc = L.pcs_mix_bubble_dew(int(bool(dew)), _lib.ptr(params), _lib.ptr(kij), _lib.ptr(temperature),
                                  _lib.ptr(molefracs), _lib.ptr(pressure), n, _lib.ptr(p), _lib.ptr(rho4),
                                  _lib.ptr(status), _lib.ptr(iters), _lib.ptr(ws), _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_mix_bubble_dew")
    return {"p": p, "rho4": rho4, "status": status.view(torch.bool), "iters": iters}


def mix_derivatives(params, kij, temperature, density):
    """(a [n], p [n], mu [n,2], v [n,2]) — PcSaftMix.derivatives (feos_torch/pcsaft_mix.py:395-420)."""
    device = params.device if isinstance(params, torch.Tensor) and params.is_cuda else _dev()
    params = _prep(params, device, (2, 8))
    kij = _prep(kij, device, (2,))
    temperature = _prep(temperature, device)
    density = _prep(density, device, (2,))
    n = temperature.shape[0]
    _same_rows(n, parameters=params, kij=kij, density=density)
    L = _lib.lib()
    with torch.cuda.device(device):
        a = torch.empty(n, dtype=_F64, device=device)
        p = torch.empty(n, dtype=_F64, device=device)
        mu = torch.empty((n, 2), dtype=_F64, device=device)
        v = torch.empty((n, 2), dtype=_F64, device=device)
        rc = L.pcs_mix_derivatives(_lib.ptr(params), _lib.ptr(kij), _lib.ptr(temperature), _lib.ptr(density), n,
                                   _lib.ptr(a), _lib.ptr(p), _lib.ptr(mu), _lib.ptr(v), _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_mix_derivatives")
    return a, p, mu, v


def _pcsaft_bubble_dew(parameters, kij, temperature, molefracs, pressure, dew):
    parameters = _as_f64(parameters, 3)
    kij = _as_f64(kij, 2)
    temperature, molefracs, pressure = _as_f64(temperature, 1), _as_f64(molefracs, 1), _as_f64(pressure, 1)
    r = mix_bubble_dew(torch.from_numpy(parameters), torch.from_numpy(kij), torch.from_numpy(temperature),
                       torch.from_numpy(molefracs), torch.from_numpy(pressure), dew)
    return Compaction(r["status"]).gather(r["rho4"]).cpu().numpy(), r["status"].cpu().numpy()  # filter_binary (:216-231)


def _bubble_point(parameters, kij, temperature, liquid_molefracs, pressure):
    """src/pcsaft.rs:43-60 — rho[n_ok, 4] = (rhoV_1, rhoV_2, rhoL_1, rhoL_2), status[N]."""
    return _pcsaft_bubble_dew(parameters, kij, temperature, liquid_molefracs, pressure, False)


def _dew_point(parameters, kij, temperature, vapor_molefracs, pressure):
    """src/pcsaft.rs:62-79."""
    return _pcsaft_bubble_dew(parameters, kij, temperature, vapor_molefracs, pressure, True)


PcSaft.bubble_point = staticmethod(_bubble_point)
PcSaft.dew_point = staticmethod(_dew_point)


def mix_jacobian(params, kij, temperature, rho4, dew):
    """[n,19] gradient of the bubble/dew pressure w.r.t. (params[0,:], params[1,:], kij[0], kij[1], T)."""
    device = rho4.device
    params = _prep(params, device, (2, 8))
    kij = _prep(kij, device, (2,))
    temperature = _prep(temperature, device)
    rho4 = _prep(rho4, device, (4,))
    n = temperature.shape[0]
    _same_rows(n, parameters=params, kij=kij, rho4=rho4)
    L = _lib.lib()
    with torch.cuda.device(device):
        jac = torch.empty((n, 19), dtype=_F64, device=device)
        ws = torch.empty(max(1, L.pcs_workspace_bytes(n) // 4), dtype=torch.int32, device=device)
        rc = L.pcs_mix_jacobian(int(bool(dew)), _lib.ptr(params), _lib.ptr(kij), _lib.ptr(temperature),
                                _lib.ptr(rho4), n, _lib.ptr(jac), _lib.ptr(ws), _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_mix_jacobian")
    return jac


# ------------------------------------------------------------------------------------------
# heterosegmented gc-PC-SAFT
# ------------------------------------------------------------------------------------------
def _check_gc(table, S, rows, n):
    """table / row encoding of include/pcsaft_hip.h on one device, n rows."""
    S = int(S)
    if table.dtype != _F64 or table.dim() != 1 or table.shape[0] != S * 8 + 3 * S * S or not table.is_contiguous():
        raise ValueError(f"table must be a contiguous float64 tensor of {S * 8 + 3 * S * S} elements for S = {S}")
    if rows.dtype != torch.uint8 or rows.dim() != 2 or rows.shape[1] != 80 or not rows.is_contiguous():
        raise ValueError("rows must be a contiguous uint8 tensor [n, 80]")
    if rows.device != table.device:
        raise ValueError(f"rows live on {rows.device}, the segment table on {table.device}")
    _same_rows(n, rows=rows)


def gc_class_order(table, S, rows):
    """Permutation (position -> row, int32 on the device) that sorts the rows of a gc model by model class — association
    class x polarity, expensive classes first — for the `order` argument of gc_bubble_dew.  The rows of a model are
    fixed, so this is computed once per model (and again after `reduce`)."""
    n = rows.shape[0]
    seg = table[: S * 8].view(S, 8)
    ids = rows[:, 0:16].long()
    cnt = rows[:, 16:32].to(_F64)
    par = seg[ids]  # [n,16,8]

    def per_molecule(v):
        return (cnt * v).view(n, 2, 8).sum(dim=2)

    ka, eab = per_molecule(par[:, :, 4]), per_molecule(par[:, :, 5])
    na, nb = per_molecule(par[:, :, 6]), per_molecule(par[:, :, 7])
    mu2 = per_molecule(par[:, :, 3] ** 2)
    associating = ((ka * eab) != 0).sum(dim=1)
    self_assoc = ((na * nb) != 0).sum(dim=1)
    cls = torch.zeros(n, dtype=torch.int64, device=rows.device)
    cls[(associating == 1) & (self_assoc == 1)] = 1
    cls[(associating == 2) & (self_assoc == 1)] = 2
    cls[(associating == 2) & (self_assoc == 2)] = 3
    key = 2 * cls + (mu2 > 0).any(dim=1).long()
    # inside a class: by the number of bond-type and segment-type entries (the trip counts of the per-row loops);
    # measured with host-sorted rows, 1e6 rows: class only 2.09 / 5.65 ms (bubble / dew), with these 2.01 / 5.33 ms
    bonds = (rows[:, 64:80] > 0).view(n, 2, 8).sum(dim=2).max(dim=1).values.long()
    segs = (rows[:, 16:32] > 0).sum(dim=1).long()
    key = (key * 9 + bonds) * 17 + segs
    return torch.argsort(key, descending=True, stable=True).to(torch.int32)


def gc_bubble_dew(table, S, rows, phi, temperature, molefracs, pressure, dew, want_iters=False, order=None):
    """table [S*8+3*S*S] f64, rows [n,80] u8 (include/pcsaft_hip.h); order: optional class order of the rows
    (gc_class_order).  -> dict(p, rho4, status, iters)."""
    device = table.device
    phi = _prep(phi, device, (2,))
    temperature = _prep(temperature, device)
    molefracs = _prep(molefracs, device)
    pressure = _prep(pressure, device)
    n = temperature.shape[0]
    _check_gc(table, S, rows, n)
    _same_rows(n, phi=phi, molefracs=molefracs, pressure=pressure)
    if order is not None and (order.dtype != torch.int32 or order.shape != (n,) or order.device != device or not order.is_contiguous()):
        raise ValueError("order must be a contiguous int32 tensor [n] on the device of the table")
    L = _lib.lib()
    with torch.cuda.device(device):
        p = torch.empty(n, dtype=_F64, device=device)
        rho4 = torch.empty((n, 4), dtype=_F64, device=device)
        status = torch.empty(n, dtype=torch.uint8, device=device)
        iters = torch.empty(n, dtype=torch.int32, device=device) if want_iters else None
        ws = torch.empty(max(1, L.pcs_workspace_bytes(n) // 4), dtype=torch.int32, device=device)
        rc = L.pcs_gc_bubble_dew(int(bool(dew)), _lib.ptr(table), int(S), _lib.ptr(rows), _lib.ptr(phi),
                                 _lib.ptr(temperature), _lib.ptr(molefracs), _lib.ptr(pressure), n, _lib.ptr(p),
                                 _lib.ptr(rho4), _lib.ptr(status), _lib.ptr(iters), _lib.ptr(order), _lib.ptr(ws),
                                 _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_gc_bubble_dew")
    return {"p": p, "rho4": rho4, "status": status.view(torch.bool), "iters": iters}


def gc_derivatives(table, S, rows, phi, temperature, density):
    device = table.device
    phi = _prep(phi, device, (2,))
    temperature = _prep(temperature, device)
    density = _prep(density, device, (2,))
    n = temperature.shape[0]
    _check_gc(table, S, rows, n)
    _same_rows(n, phi=phi, density=density)
    L = _lib.lib()
    with torch.cuda.device(device):
        a = torch.empty(n, dtype=_F64, device=device)
        p = torch.empty(n, dtype=_F64, device=device)
        mu = torch.empty((n, 2), dtype=_F64, device=device)
        v = torch.empty((n, 2), dtype=_F64, device=device)
        rc = L.pcs_gc_derivatives(_lib.ptr(table), int(S), _lib.ptr(rows), _lib.ptr(phi), _lib.ptr(temperature),
                                  _lib.ptr(density), n, _lib.ptr(a), _lib.ptr(p), _lib.ptr(mu), _lib.ptr(v),
                                  _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_gc_derivatives")
    return a, p, mu, v


def gc_jacobian(table, S, rows, phi, temperature, rho4, dew, order=None):
    """-> jac [n,7] = dp/d(A00, A01, A11, B00, B01, B11, T), agg [n,6].  order: optional class order (gc_class_order)."""
    device = table.device
    phi = _prep(phi, device, (2,))
    temperature = _prep(temperature, device)
    rho4 = _prep(rho4, device, (4,))
    n = temperature.shape[0]
    _check_gc(table, S, rows, n)
    _same_rows(n, phi=phi, rho4=rho4)
    L = _lib.lib()
    with torch.cuda.device(device):
        jac = torch.empty((n, 7), dtype=_F64, device=device)
        agg = torch.empty((n, 6), dtype=_F64, device=device)
        rc = L.pcs_gc_jacobian(int(bool(dew)), _lib.ptr(table), int(S), _lib.ptr(rows), _lib.ptr(phi),
                               _lib.ptr(temperature), _lib.ptr(rho4), n, _lib.ptr(jac), _lib.ptr(agg),
                               _lib.ptr(order) if order is not None and order.shape[0] == n else None,
                               _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_gc_jacobian")
    return jac, agg


def gc_segment_gradient(table, S, rows, phi, temperature, rho4, dew, gout=None, order=None):
    """[S,8] = sum_i gout[i] * d p_i / d (segment parameter table) at the converged densities rho4 (gout None = 1)."""
    device = table.device
    phi = _prep(phi, device, (2,))
    temperature = _prep(temperature, device)
    rho4 = _prep(rho4, device, (4,))
    gout = None if gout is None else _prep(gout, device)
    n = temperature.shape[0]
    _check_gc(table, S, rows, n)
    _same_rows(n, phi=phi, rho4=rho4, gout=gout)
    L = _lib.lib()
    with torch.cuda.device(device):
        grad = torch.zeros((int(S), 8), dtype=_F64, device=device)
        rc = L.pcs_gc_segment_gradient(int(bool(dew)), _lib.ptr(table), int(S), _lib.ptr(rows), _lib.ptr(phi),
                                       _lib.ptr(temperature), _lib.ptr(rho4), n, _lib.ptr(gout), _lib.ptr(grad),
                                       _lib.ptr(order) if order is not None and order.shape[0] == n else None,
                                       _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_gc_segment_gradient")
    return grad


def _opt(x, device, shape_tail=None):
    return None if x is None else _prep(x, device, shape_tail)


def pure_derivatives_vjp(params, temperature, density, g_a=None, g_p=None, g_dp=None):
    """Backward of PcSaftPure.derivatives: -> (grad_params [n,8], grad_T [n], grad_rho [n])."""
    device = params.device if isinstance(params, torch.Tensor) and params.is_cuda else _dev()
    params = _prep(params, device, (8,))
    temperature, density = _prep(temperature, device), _prep(density, device)
    g_a, g_p, g_dp = _opt(g_a, device), _opt(g_p, device), _opt(g_dp, device)
    n = temperature.shape[0]
    _same_rows(n, parameters=params, density=density, g_a=g_a, g_p=g_p, g_dp=g_dp)
    L = _lib.lib()
    with torch.cuda.device(device):
        gpar = torch.empty((n, 8), dtype=_F64, device=device)
        gT = torch.empty(n, dtype=_F64, device=device)
        grho = torch.empty(n, dtype=_F64, device=device)
        rc = L.pcs_pure_derivatives_vjp(_lib.ptr(params), _lib.ptr(temperature), _lib.ptr(density), n, _lib.ptr(g_a),
                                        _lib.ptr(g_p), _lib.ptr(g_dp), _lib.ptr(gpar), _lib.ptr(gT), _lib.ptr(grho),
                                        _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_pure_derivatives_vjp")
    return gpar, gT, grho


def mix_derivatives_vjp(params, kij, temperature, density, g_a=None, g_p=None, g_mu=None, g_v=None):
    """Backward of PcSaftMix.derivatives: -> grad [n,21] = dL/d(16 parameters, kij0, kij1, T, rho_0, rho_1)."""
    device = params.device if isinstance(params, torch.Tensor) and params.is_cuda else _dev()
    params = _prep(params, device, (2, 8))
    kij = _prep(kij, device, (2,))
    temperature, density = _prep(temperature, device), _prep(density, device, (2,))
    g_a, g_p, g_mu, g_v = _opt(g_a, device), _opt(g_p, device), _opt(g_mu, device, (2,)), _opt(g_v, device, (2,))
    n = temperature.shape[0]
    _same_rows(n, parameters=params, kij=kij, density=density, g_a=g_a, g_p=g_p, g_mu=g_mu, g_v=g_v)
    L = _lib.lib()
    with torch.cuda.device(device):
        grad = torch.empty((n, 21), dtype=_F64, device=device)
        ws = torch.empty(max(1, L.pcs_workspace_bytes(n) // 4), dtype=torch.int32, device=device)
        rc = L.pcs_mix_derivatives_vjp(_lib.ptr(params), _lib.ptr(kij), _lib.ptr(temperature), _lib.ptr(density), n,
                                       _lib.ptr(g_a), _lib.ptr(g_p), _lib.ptr(g_mu), _lib.ptr(g_v), _lib.ptr(grad),
                                       _lib.ptr(ws), _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_mix_derivatives_vjp")
    return grad


def gc_derivatives_vjp(table, S, rows, phi, temperature, density, g_a=None, g_p=None, g_mu=None, g_v=None, order=None):
    """Backward of GcPcSaftMix.derivatives: -> (grad_seg [S,8], jac9 [n,9] = dL/d(6 aggregates, T, rho_0, rho_1), agg [n,6])."""
    device = table.device
    phi = _prep(phi, device, (2,))
    temperature, density = _prep(temperature, device), _prep(density, device, (2,))
    g_a, g_p, g_mu, g_v = _opt(g_a, device), _opt(g_p, device), _opt(g_mu, device, (2,)), _opt(g_v, device, (2,))
    n = temperature.shape[0]
    _check_gc(table, S, rows, n)
    _same_rows(n, phi=phi, density=density, g_a=g_a, g_p=g_p, g_mu=g_mu, g_v=g_v)
    L = _lib.lib()
    with torch.cuda.device(device):
        gseg = torch.zeros((int(S), 8), dtype=_F64, device=device)
        jac9 = torch.empty((n, 9), dtype=_F64, device=device)
        agg = torch.empty((n, 6), dtype=_F64, device=device)
        rc = L.pcs_gc_derivatives_vjp(_lib.ptr(table), int(S), _lib.ptr(rows), _lib.ptr(phi), _lib.ptr(temperature),
                                      _lib.ptr(density), n, _lib.ptr(g_a), _lib.ptr(g_p), _lib.ptr(g_mu), _lib.ptr(g_v),
                                      _lib.ptr(gseg), _lib.ptr(jac9), _lib.ptr(agg),
                                      _lib.ptr(order) if order is not None and order.shape[0] == n else None,
                                      _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_gc_derivatives_vjp")
    return gseg, jac9, agg


def mixn_derivatives(params, temperature, density):
    """n-component PcSaftMix.derivatives without k_ij: params [n, nc, 8], density [n, nc] -> (a [n], p [n], mu [n,nc], v [n,nc])."""
    device = params.device if isinstance(params, torch.Tensor) and params.is_cuda else _dev()
    if params.dim() != 3 or params.shape[2] != 8 or not 1 <= params.shape[1] <= 6:
        raise ValueError("parameters must have shape [N, n, 8] with 1 <= n <= 6 components")
    nc = int(params.shape[1])
    params = _prep(params, device, (nc, 8))
    temperature = _prep(temperature, device)
    density = _prep(density, device, (nc,))
    n = temperature.shape[0]
    _same_rows(n, parameters=params, density=density)
    L = _lib.lib()
    with torch.cuda.device(device):
        a = torch.empty(n, dtype=_F64, device=device)
        p = torch.empty(n, dtype=_F64, device=device)
        mu = torch.empty((n, nc), dtype=_F64, device=device)
        v = torch.empty((n, nc), dtype=_F64, device=device)
        rc = L.pcs_mixn_derivatives(_lib.ptr(params), _lib.ptr(temperature), _lib.ptr(density), nc, n, _lib.ptr(a), _lib.ptr(p),
                                    _lib.ptr(mu), _lib.ptr(v), _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_mixn_derivatives")
    return a, p, mu, v


def mixn_derivatives_vjp(params, temperature, density, g_a=None, g_p=None, g_mu=None, g_v=None):
    """Backward of the n-component derivatives: grad [n, 9 nc + 1] = dL/d(parameters [nc][8], T, rho [nc])."""
    device = params.device if isinstance(params, torch.Tensor) and params.is_cuda else _dev()
    nc = int(params.shape[1])
    params = _prep(params, device, (nc, 8))
    temperature = _prep(temperature, device)
    density = _prep(density, device, (nc,))
    g_a, g_p, g_mu, g_v = _opt(g_a, device), _opt(g_p, device), _opt(g_mu, device, (nc,)), _opt(g_v, device, (nc,))
    n = temperature.shape[0]
    _same_rows(n, parameters=params, density=density, g_a=g_a, g_p=g_p, g_mu=g_mu, g_v=g_v)
    L = _lib.lib()
    with torch.cuda.device(device):
        grad = torch.empty((n, 9 * nc + 1), dtype=_F64, device=device)
        rc = L.pcs_mixn_derivatives_vjp(_lib.ptr(params), _lib.ptr(temperature), _lib.ptr(density), nc, n, _lib.ptr(g_a), _lib.ptr(g_p),
                                        _lib.ptr(g_mu), _lib.ptr(g_v), _lib.ptr(grad), _lib.current_stream_ptr(device))
        _lib.check(rc, "pcs_mixn_derivatives_vjp")
    return grad
