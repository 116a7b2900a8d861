"""``GcPcSaftMix`` / ``GcPcSaft`` — drop-ins for the reference's heterosegmented gc-PC-SAFT classes
(feos_torch/gc_pcsaft.py:13-528, src/gc_pcsaft.rs:15-171) backed by the gfx950 kernels.

Constructor arguments, method names, return order ``(pressure [Pa], nans)`` and the mutate-on-call
semantics follow the reference.  Instead of its dense ``[N,2,S]`` / ``[N,2,S,S]`` tensors the
molecule structures are encoded once into 80 bytes per row (see include/pcsaft_hip.h); identical
molecules are encoded once and re-used.

Gradients: the eight segment parameter vectors, ``k_ab`` (the tensors inside ``binary_segment_records``),
``phi`` and ``temperature`` — everything the reference's autograd reaches (feos_torch/gc_pcsaft.py:14-22,
:54-86).  k_ab and phi enter the model only through the six dispersion aggregates of a row (:177-194);
the kernel returns dp/d(aggregates) and this module chains them with two [S,N]x[N,S] products.  The
segment parameters enter through 26 molecule-level sums, the dispersion double sums and the bond
diameters; ``pcs_gc_segment_gradient`` differentiates through all of them on the device and reduces
g_i dp_i/d(table) over the rows into one [S,8] array.  (The reference's own d/dphi and d/depsilon_k are
NaN whenever the table holds a segment with epsilon_k = 0 such as '>C<' — sqrt(0) under autograd; here
that one non-differentiable term is left out and everything else is finite.)
"""
import numpy as np
import torch
from torch.autograd.function import once_differentiable

from . import native

MAXE = 8  # distinct segment types / bond types per molecule in the 80-byte row encoding


def _encode_molecule(segs, bonds, idx):
    """40 bytes: seg_id[8], seg_cnt[8], bond_a[8], bond_b[8], bond_cnt[8] (counts 0 = unused)."""
    ids = [idx[s] for s in segs]
    scount = {}
    for a in ids:
        scount[a] = scount.get(a, 0) + 1
    bcount = {}
    for i, j in bonds:
        a, b = ids[i], ids[j]
        key = (a, b) if a >= b else (b, a)  # larger index first (feos_torch/gc_pcsaft.py:35)
        bcount[key] = bcount.get(key, 0) + 1
    if len(scount) > MAXE or len(bcount) > MAXE:
        raise ValueError(f"a molecule may use at most {MAXE} distinct segment types and {MAXE} distinct bond types")
    if max(list(scount.values()) + list(bcount.values()) + [0]) > 255:
        raise ValueError("segment / bond multiplicity above 255")
    out = np.zeros(40, dtype=np.uint8)
    for k, (a, c) in enumerate(sorted(scount.items())):
        out[k] = a
        out[8 + k] = c
    for k, ((a, b), c) in enumerate(sorted(bcount.items())):
        out[16 + k] = a
        out[24 + k] = b
        out[32 + k] = c
    return out


def encode_rows(segment_identifier, segment_lists, bond_lists):
    """[N, 80] uint8 row encoding (layout of include/pcsaft_hip.h), by the native encoder
    (csrc_host/gc_encode.cpp, built by feos_torch_amd.build.build_host / __graft_entry__.build)."""
    try:
        from . import _gc_encode
    except ImportError as e:  # pragma: no cover
        raise ImportError("feos_torch_amd._gc_encode is not built: run `python -m feos_torch_amd.build`") from e
    rows = np.zeros((len(segment_lists), 80), dtype=np.uint8)
    _gc_encode.encode_rows(list(segment_identifier), segment_lists, bond_lists, rows)
    return rows


def encode_rows_device(segment_identifier, segment_lists, bond_lists, device):
    """[N, 80] uint8 row encoding ON THE DEVICE: the host (csrc_host/gc_encode.cpp::encode_indices) encodes every distinct
    molecule once into a 40-byte table entry and writes two int32 per row; the 80-byte rows are gathered from the table on the
    GPU.  8 instead of 80 bytes per row cross the host memory and the PCIe link (1e6 rows: ~0.03 s of host time)."""
    try:
        from . import _gc_encode
    except ImportError as e:  # pragma: no cover
        raise ImportError("feos_torch_amd._gc_encode is not built: run `python -m feos_torch_amd.build`") from e
    n = len(segment_lists)
    idx = np.empty((n, 2), dtype=np.int32)
    table = np.frombuffer(_gc_encode.encode_indices(list(segment_identifier), segment_lists, bond_lists, idx), dtype=np.int64)
    if n == 0:
        return torch.zeros((0, 80), dtype=torch.uint8, device=device)
    tab = torch.from_numpy(table.reshape(-1, 5).copy()).to(device)  # [U, 5] 8-byte fields: seg_id, seg_cnt, bond_a, bond_b, bond_cnt
    ix = torch.from_numpy(idx).to(device).long()
    # rows[r] = (field f of molecule 0, field f of molecule 1) for f = 0..4  (layout of include/pcsaft_hip.h)
    return tab[ix].permute(0, 2, 1).contiguous().view(torch.uint8).view(n, 80)


def encode_rows_py(segment_identifier, segment_lists, bond_lists):
    """Pure-Python specification of encode_rows (tests compare the native encoder against it)."""
    idx = {s: i for i, s in enumerate(segment_identifier)}
    cache, id_cache = {}, {}
    n = len(segment_lists)
    rows = np.zeros((n, 80), dtype=np.uint8)
    for r in range(n):
        for c in range(2):
            segs, bonds = segment_lists[r][c], bond_lists[r][c]
            # batches usually re-use the same list objects for the same molecule: identity first
            enc = id_cache.get((id(segs), id(bonds)))
            if enc is None:
                key = (tuple(segs), tuple(map(tuple, bonds)))
                enc = cache.get(key)
                if enc is None:
                    enc = _encode_molecule(segs, bonds, idx)
                    cache[key] = enc
                id_cache[(id(segs), id(bonds))] = enc
            rows[r, 8 * c:8 * c + 8] = enc[0:8]
            rows[r, 16 + 8 * c:16 + 8 * c + 8] = enc[8:16]
            rows[r, 32 + 8 * c:32 + 8 * c + 8] = enc[16:24]
            rows[r, 48 + 8 * c:48 + 8 * c + 8] = enc[24:32]
            rows[r, 64 + 8 * c:64 + 8 * c + 8] = enc[32:40]
    return rows


def build_table(seg, kab):
    """seg [S,8], kab [S,S] (detached, any device) -> flat table [S*8 + 3*S*S] (see include/pcsaft_hip.h)."""
    sigma, eps = seg[:, 1], seg[:, 2]
    s3 = (0.5 * (sigma[:, None] + sigma[None, :])) ** 3
    ee = eps[:, None] * eps[None, :]
    return torch.cat([seg.reshape(-1), (ee.sqrt() * s3).reshape(-1), (ee * s3).reshape(-1), (1.0 - kab).reshape(-1)]).contiguous()


def _kab_gradient(table, S, rows, ph, T, dA01, dB01):
    """[S,S] gradient w.r.t. k_ab from the per-row derivatives w.r.t. the mixed aggregates A01, B01 (upstream gradient
    already folded in): d A01 / d K_ab = 2 sqrt(phi0 phi1)/T m0a m1b E1_ab, d B01 / d K_ab = 4 phi0 phi1/T^2 m0a m1b E2_ab K_ab
    with K = 1 - k_ab (feos_torch/gc_pcsaft.py:177-194)."""
    seg_m = table[: S * 8].view(S, 8)[:, 0]
    E1 = table[S * 8: S * 8 + S * S].view(S, S)
    E2 = table[S * 8 + S * S: S * 8 + 2 * S * S].view(S, S)
    K = table[S * 8 + 2 * S * S:].view(S, S)
    M = []
    for c in range(2):  # m-weighted dense counts, built on the device from the row encoding
        ids = rows[:, 8 * c:8 * c + 8].long()
        cnt = rows[:, 16 + 8 * c:16 + 8 * c + 8].to(torch.float64)
        Mc = torch.zeros((rows.shape[0], S), dtype=torch.float64, device=rows.device)
        Mc.scatter_add_(1, ids, cnt * seg_m[ids])
        M.append(Mc)
    pp = ph[:, 0] * ph[:, 1]
    w1 = dA01 * 2.0 * pp.sqrt() / T
    w2 = dB01 * 4.0 * pp / (T * T)
    G1 = M[0].t() @ (w1[:, None] * M[1])
    G2 = M[0].t() @ (w2[:, None] * M[1])
    return -(E1 * G1) - (E2 * K) * G2


def _phi_gradient(dAB, agg, ph):
    """[n,2] gradient w.r.t. phi from the per-row derivatives dAB [n,6] w.r.t. the aggregates (A ~ phi_i, sqrt(phi_0 phi_1),
    B ~ phi_i^2, phi_0 phi_1)."""
    A00, A01, A11, B00, B01, B11 = (agg[:, k] for k in range(6))
    d0 = dAB[:, 0] * A00 + 0.5 * dAB[:, 1] * A01 + 2.0 * dAB[:, 3] * B00 + dAB[:, 4] * B01
    d1 = dAB[:, 2] * A11 + 0.5 * dAB[:, 1] * A01 + 2.0 * dAB[:, 5] * B11 + dAB[:, 4] * B01
    return torch.stack([d0 / ph[:, 0], d1 / ph[:, 1]], dim=1)


class _GcDerivatives(torch.autograd.Function):
    """(a, p, mu, v) = derivatives(...) with gradients to k_ab, phi, temperature, density and the eight segment parameter
    vectors (feos_torch/gc_pcsaft.py:116-253, :443-468 are torch graphs in the reference)."""

    @staticmethod
    def forward(ctx, model, kab, phi, temperature, density, *segment_parameters):
        dev = model.device
        table = build_table(model.seg.to(dev), kab.detach().to(dev, torch.float64))
        ph = native._prep(phi, dev, (2,))
        T = native._prep(temperature, dev)
        rho = native._prep(density, dev, (2,))
        a, p, mu, v = native.gc_derivatives(table, model.S, model.rows, ph, T, rho)
        ctx.save_for_backward(table, model.rows, ph, T, rho)
        ctx.set_materialize_grads(False)
        ctx.S = model.S
        ctx.devs = (kab.device, phi.device, temperature.device, density.device)
        ctx.seg_devs = [q.device for q in segment_parameters]
        out = phi.device
        return a.to(out), p.to(out), mu.to(out), v.to(out)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_a, g_p, g_mu, g_v):
        table, rows, ph, T, rho = ctx.saved_tensors
        nseg = len(ctx.seg_devs)
        if g_a is None and g_p is None and g_mu is None and g_v is None:
            return (None,) * (5 + nseg)
        S = ctx.S
        gseg, jac9, agg = native.gc_derivatives_vjp(table, S, rows, ph, T, rho, g_a, g_p, g_mu, g_v)
        need = ctx.needs_input_grad
        gk = _kab_gradient(table, S, rows, ph, T, jac9[:, 1], jac9[:, 4]).to(ctx.devs[0]) if need[1] else None
        gphi = _phi_gradient(jac9, agg, ph).to(ctx.devs[1]) if need[2] else None
        gT = jac9[:, 6].contiguous().to(ctx.devs[2]) if need[3] else None
        grho = jac9[:, 7:9].contiguous().to(ctx.devs[3]) if need[4] else None
        gs = [gseg[:, k].to(ctx.seg_devs[k]) if need[5 + k] else None for k in range(nseg)]
        return (None, gk, gphi, gT, grho, *gs)


class _GcBubbleDew(torch.autograd.Function):
    """value[n_ok], nans[n].  Dense solve, one compaction plan (its 4-byte row count is the call's only host synchronisation),
    single-kernel gathers only when rows were dropped (native.Compaction; the reference drops them inside the native call,
    src/gc_pcsaft.rs:103-171)."""

    @staticmethod
    def forward(ctx, dew, model, kab, phi, temperature, molefracs, pressure, box, *segment_parameters):
        dev = model.device
        table = build_table(model.seg.to(dev), kab.detach().to(dev, torch.float64))
        ph = native._prep(phi, dev, (2,))
        T = native._prep(temperature, dev)
        r = native.gc_bubble_dew(table, model.S, model.rows, ph, T, native._prep(molefracs, dev),
                                 native._prep(pressure, dev), dew, order=model._class_order(table))
        nans = r["status"]
        comp = native.Compaction(nans)
        box.append(comp)
        value = comp.gather(r["p"])
        needs = [ctx.needs_input_grad[2], ctx.needs_input_grad[3], ctx.needs_input_grad[4]]
        seg_needs = list(ctx.needs_input_grad[8:])
        ctx.saved = False
        if any(needs) or any(seg_needs):
            rows_ok, ph_ok, T_ok, rho4_ok = comp.gather(model.rows), comp.gather(ph), comp.gather(T), comp.gather(r["rho4"])
            # the class order belongs to the uncompacted rows: used when every row converged
            order = model._class_order(table) if comp.all_ok else None
            if any(needs):
                jac, agg = native.gc_jacobian(table, model.S, rows_ok, ph_ok, T_ok, rho4_ok, dew, order=order)
            else:
                jac = agg = T.new_empty(0)
            ctx.save_for_backward(jac, agg, rows_ok, ph_ok, T_ok, table, rho4_ok if any(seg_needs) else T.new_empty(0),
                                  order if (order is not None and any(seg_needs)) else nans.new_empty(0))
            ctx.comp = comp
            ctx.saved = True
        ctx.needs = needs
        ctx.seg_needs = seg_needs
        ctx.dew = bool(dew)
        ctx.seg_devs = [p.device if isinstance(p, torch.Tensor) else None for p in segment_parameters]
        ctx.S = model.S
        ctx.devs = (kab.device, phi.device, temperature.device)
        out_device = phi.device
        nans = nans.to(out_device)
        ctx.mark_non_differentiable(nans)
        return value.to(out_device), nans

    @staticmethod
    @once_differentiable
    def backward(ctx, g_value, _g):
        nseg = len(ctx.seg_needs)
        if not ctx.saved:  # nothing the pressure depends on required a gradient
            return (None,) * (8 + nseg)
        jac, agg, rows, ph, T, table, rho4, order = ctx.saved_tensors
        comp, S = ctx.comp, ctx.S
        g = g_value.to(table.device).contiguous()
        gk = gphi = gT = None
        if ctx.needs[0]:
            gk = _kab_gradient(table, S, rows, ph, T, g * jac[:, 1], g * jac[:, 4]).to(ctx.devs[0])
        if ctx.needs[1]:
            gphi = comp.expand(_phi_gradient(jac, agg, ph), g).to(ctx.devs[1])
        if ctx.needs[2]:
            gT = comp.expand(jac, g, 6, 1).view(comp.n).to(ctx.devs[2])
        gseg = [None] * nseg
        if any(ctx.seg_needs):
            # the whole table in one kernel: sum_i g_i dp_i/d seg[S,8] (the segment-parameter gradient is lazy: it needs
            # the upstream gradient, so unlike the per-row Jacobians above it runs here, not in forward)
            G = native.gc_segment_gradient(table, S, rows, ph, T, rho4, ctx.dew, gout=g,
                                           order=order if order.numel() == rows.shape[0] else None)
            gseg = [G[:, k].to(ctx.seg_devs[k]) if need else None for k, need in enumerate(ctx.seg_needs)]
        return (None, None, gk, gphi, gT, None, None, None, *gseg)


class GcPcSaftMix:
    def __init__(self, segment_identifier, parameter, segment_lists, bond_lists, binary_segment_records, phi=None):
        """Arguments as the reference (feos_torch/gc_pcsaft.py:14-22): segment identifiers [S], the
        8 segment parameter vectors (m, sigma, epsilon_k, mu, kappa_ab, epsilon_k_ab, na, nb), per-row
        [segments of molecule 1, of molecule 2], per-row bond index pairs, [(s1, s2, k_ab)], phi [N,2]."""
        parameter = tuple(parameter)
        if len(parameter) != 8:
            raise ValueError("parameter must hold the 8 segment parameter vectors (m, sigma, epsilon_k, mu, kappa_ab, epsilon_k_ab, na, nb)")
        self.segment_identifier = list(segment_identifier)
        self.S = len(self.segment_identifier)
        if self.S > 32:
            raise ValueError("at most 32 segment types per table")
        self.device = _device_of(phi, *parameter)
        # the caller's tensors stay attached to the model: they are inputs of the autograd Function of every property
        self._segment_parameters = tuple(p if isinstance(p, torch.Tensor) else torch.as_tensor(p, dtype=torch.float64)
                                         for p in parameter)
        for p in self._segment_parameters:
            if p.dtype != torch.float64 or tuple(p.shape) != (self.S,):
                raise ValueError(f"every segment parameter vector must be float64 of shape [{self.S}]")
        self.seg = torch.stack([p.detach().cpu() for p in self._segment_parameters], dim=1).contiguous()
        n = len(segment_lists)
        self.rows = encode_rows_device(self.segment_identifier, segment_lists, bond_lists, self.device)
        # "Only up to one associating segment per component is allowed!" (:76-80)
        is_assoc = ((self.seg[:, 4] * self.seg[:, 5]) != 0).to(self.device)
        for c in range(2):
            cnt = (self.rows[:, 16 + 8 * c:16 + 8 * c + 8] * is_assoc[self.rows[:, 8 * c:8 * c + 8].long()]).sum(dim=1)
            if bool((cnt > 1).any()):
                raise Exception("Only up to one associating segment per component is allowed!")
        self._order = None  # class order of the rows, see _class_order
        idx = {s: i for i, s in enumerate(self.segment_identifier)}
        # symmetric k_ab matrix built exactly as the reference does (:60-63), keeps autograd history
        self.kab = torch.zeros((self.S, self.S), dtype=torch.float64)
        for s1, s2, k in binary_segment_records:
            self.kab[idx[s1], idx[s2]] = k
            self.kab[idx[s2], idx[s1]] = k
        self.phi = torch.ones((n, 2), dtype=torch.float64) if phi is None else phi
        self.gc_pcsaft = GcPcSaft._from_model(self)

    def _table(self):
        return build_table(self.seg.to(self.device), self.kab.detach().to(self.device))

    def helmholtz_energy_density(self, temperature, density, kab=None):
        """a(T, rho_1, rho_2) [A^-3], shape [N, 1] (:116-253); differentiable."""
        return self.derivatives(temperature, density)[0][:, None]

    def derivatives(self, temperature, density):
        """(a, p, mu [N,2], v [N,2]) (:443-468); differentiable w.r.t. the segment parameters, k_ab, phi, temperature and
        density (pcs_gc_derivatives_vjp is the backward pass)."""
        temperature = torch.as_tensor(temperature, dtype=torch.float64)
        density = torch.as_tensor(density, dtype=torch.float64)
        return _GcDerivatives.apply(self, self.kab, self.phi, temperature, density, *self._segment_parameters)

    def bubble_point(self, temperature, liquid_molefracs, pressure):
        """(p [Pa], nans) (:470-490)."""
        return self._bubble_dew(False, temperature, liquid_molefracs, pressure)

    def dew_point(self, temperature, vapor_molefracs, pressure):
        """(p [Pa], nans) (:492-512)."""
        return self._bubble_dew(True, temperature, vapor_molefracs, pressure)

    def _bubble_dew(self, dew, temperature, molefracs, pressure):
        box = []
        # mole fractions and initial pressure do not enter the reference's final formula (:483-490): no gradient flows to them
        value, nans = _GcBubbleDew.apply(dew, self, self.kab, self.phi, temperature, _detached(molefracs), _detached(pressure), box,
                                         *self._segment_parameters)
        self._reduce(box[0])
        return value, nans

    def _class_order(self, table):
        """Class order of the model's rows for the kernels' schedule (native.gc_class_order): computed on first use and
        again after `reduce` — the rows are fixed in between."""
        if self._order is None or self._order.shape[0] != self.rows.shape[0]:
            self._order = native.gc_class_order(table, self.S, self.rows)
        return self._order

    def _reduce(self, comp):
        if comp.all_ok:
            return
        self.rows = comp.gather(self.rows)
        self.phi = native.compact_rows(comp, self.phi)
        self._order = None

    def reduce(self, nans):
        """Drop failed rows from the model (:514-528)."""
        self._reduce(native.Compaction(nans.to(self.device)))


def _device_of(*tensors):
    """The GPU a gc model computes on: the device of the first CUDA tensor among its inputs (phi, segment parameters), else the
    current GPU -- not a device pinned at import or construction time of some other model."""
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    return native._dev()


def _detached(x):
    return x.detach() if isinstance(x, torch.Tensor) else x


class GcPcSaft:
    """Mirror of the reference's Rust pyclass ``GcPcSaft`` (src/gc_pcsaft.rs:15-99): built from
    segment records, per-row molecule structures, binary segment records and phi; ``bubble_point``
    / ``dew_point`` take numpy arrays and return ``(rho[n_ok,4], status[N])``."""

    def __init__(self, segment_records, segments, bonds, binary_segment_records, phi):
        ident = [s for s, _ in segment_records]
        par = np.stack([np.asarray(v, dtype=np.float64) for _, v in segment_records], axis=0)
        self.S = len(ident)
        self.device = _device_of(phi)
        self.seg = torch.from_numpy(par).contiguous()
        self.rows = encode_rows_device(ident, segments, bonds, self.device)
        kab = torch.zeros((self.S, self.S), dtype=torch.float64)
        idx = {s: i for i, s in enumerate(ident)}
        for s1, s2, k in binary_segment_records:
            kab[idx[s1], idx[s2]] = float(k)
            kab[idx[s2], idx[s1]] = float(k)
        self.table = build_table(self.seg.to(self.device), kab.to(self.device))
        self.phi = torch.as_tensor(np.asarray(phi, dtype=np.float64))
        self._order = None

    @classmethod
    def _from_model(cls, model):
        obj = cls.__new__(cls)
        obj.S, obj.device, obj.seg, obj.rows = model.S, model.device, model.seg, model.rows
        obj.table = model._table()
        obj.phi = model.phi.detach()
        obj._order = None
        return obj

    def _solve(self, temperature, molefracs, pressure, dew):
        t, x, p = (native._as_f64(v, 1) for v in (temperature, molefracs, pressure))
        if self._order is None or self._order.shape[0] != self.rows.shape[0]:
            self._order = native.gc_class_order(self.table, self.S, self.rows)
        r = native.gc_bubble_dew(self.table, self.S, self.rows, self.phi, torch.from_numpy(t), torch.from_numpy(x),
                                 torch.from_numpy(p), dew, order=self._order)
        return native.Compaction(r["status"]).gather(r["rho4"]).cpu().numpy(), r["status"].cpu().numpy()

    def bubble_point(self, temperature, liquid_molefracs, pressure):
        return self._solve(temperature, liquid_molefracs, pressure, False)

    def dew_point(self, temperature, vapor_molefracs, pressure):
        return self._solve(temperature, vapor_molefracs, pressure, True)
