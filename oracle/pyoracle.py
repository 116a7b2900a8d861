"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes front-end of the CPU restatement in this directory (``liboracle.so``).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package ``feos_torch_amd`` never does.

Every function takes/returns numpy float64 arrays.  ``prec=1`` runs the solver in x87
``long double`` with a 1e-17 step tolerance (results rounded to double) and is the parity
reference; ``prec=0`` is the plain-double port that the benchmark times as ``cpu_baseline``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i64 = ctypes.c_int64
_int = ctypes.c_int


def build(force=False):
    """Compile the oracle with g++ (``make -C oracle``)."""
    if force or not os.path.exists(_LIB_PATH) or _stale():
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def _stale():
    t = os.path.getmtime(_LIB_PATH)
    for f in os.listdir(_HERE):
        if f.endswith((".hpp", ".cpp")) and os.path.getmtime(os.path.join(_HERE, f)) > t:
            return True
    return False


FAST_FLAGS = "-O3 -march=native (default fp contraction), OpenMP"
PARITY_FLAGS = "-O2 -ffp-contract=off, OpenMP"


def _cpu_key():
    """Short hash of this host's CPU model + ISA flags: a -march=native build must not travel to another machine."""
    import hashlib

    model, flags = "", ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and not model:
                    model = line
                elif line.startswith("flags") and not flags:
                    flags = line
    except OSError:
        pass
    return hashlib.sha1((model + flags).encode()).hexdigest()[:12]


def use_fast_build():
    """bench.py's cpu_baseline leg only: switch this process to the performance build of the same sources
    (oracle/Makefile `fast`: -O3 -march=native), compiled here for this host's CPU.  Must be called before the first
    oracle call of the process; parity tests never call it."""
    global _LIB_PATH
    assert _lib is None, "use_fast_build() must come before the first oracle call"
    out = os.path.join(_HERE, "_build", f"liboracle_fast.{_cpu_key()}.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".hpp", ".cpp"))]
    if not os.path.exists(out) or any(os.path.getmtime(f) > os.path.getmtime(out) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "fast", f"FAST_OUT={os.path.relpath(out, _HERE)}"])
    _LIB_PATH = out
    return out


def lib():
    global _lib
    if _lib is None:
        if _LIB_PATH.endswith("liboracle.so"):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_num_threads.restype = _int
        L.orc_dual3_selftest.restype = _int
        L.orc_pure_derivatives.argtypes = [_f64p, _f64p, _f64p, _i64, _f64p, _f64p, _f64p]
        L.orc_pure_helmholtz.argtypes = [_f64p, _f64p, _f64p, _i64, _f64p]
        L.orc_pure_vle.argtypes = [_f64p, _f64p, _i64, _int, _f64p, _f64p, _u8p, _i32p, _i32p]
        L.orc_pure_vapor_pressure.argtypes = [_f64p, _f64p, _i64, _int, _f64p, _u8p]
        L.orc_pure_vapor_pressure_at.argtypes = [_f64p, _f64p, _f64p, _f64p, _i64, _f64p]
        L.orc_pure_liquid_density.argtypes = [_f64p, _f64p, _f64p, _i64, _int, _f64p, _u8p]
        L.orc_pure_liquid_density_root.argtypes = [_f64p, _f64p, _f64p, _i64, _int, _f64p, _u8p]
        L.orc_pure_equilibrium_liquid_density.argtypes = [_f64p, _f64p, _i64, _int, _f64p, _u8p]
        L.orc_pure_property_grad.argtypes = [_int, _f64p, _f64p, _f64p, _f64p, _f64p, _i64, _f64p, _f64p]
        L.orc_pure_property_grad_ld.argtypes = [_int, _f64p, _f64p, _f64p, _f64p, _f64p, _i64, _f64p, _f64p]
        L.orc_mix_derivatives.argtypes = [_f64p, _f64p, _f64p, _f64p, _i64, _int, _f64p, _f64p, _f64p, _f64p]
        L.orc_mix_helmholtz.argtypes = [_f64p, _f64p, _f64p, _f64p, _i64, _f64p]
        L.orc_mix_bubble_dew.argtypes = [_f64p, _f64p, _f64p, _f64p, _f64p, _i64, _int, _int, _f64p, _f64p, _u8p]
        L.orc_mix_bubble_dew_grad.argtypes = [_f64p, _f64p, _f64p, _f64p, _i64, _int, _f64p, _f64p]
        L.orc_mixn_derivatives.argtypes = [_f64p, _f64p, _f64p, _int, _i64, _int, _f64p, _f64p, _f64p, _f64p]
        L.orc_mixn_derivatives.restype = _int
        L.orc_mix_derivatives_ld.argtypes = [_f64p, _f64p, _f64p, _f64p, _i64, _f64p, _f64p, _f64p, _f64p]
        L.orc_mix_bubble_dew_grad_ld.argtypes = [_f64p, _f64p, _f64p, _f64p, _i64, _int, _f64p, _f64p]
        L.orc_mix_bubble_dew_continuation.argtypes = [_f64p, _f64p, _f64p, _f64p, _i64, _int, _int, _f64p, _f64p, _i32p, _i32p]
        _i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
        L.orc_gc_derivatives.argtypes = [_int, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _i64, _int, _f64p, _f64p, _f64p, _f64p]
        L.orc_gc_derivatives.restype = _int
        L.orc_gc_bubble_dew.argtypes = [_int, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _i64, _int, _int, _f64p, _f64p, _u8p]
        L.orc_gc_bubble_dew.restype = _int
        L.orc_gc_bubble_dew_grad.argtypes = [_int, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _i64, _int, _int, _int, _int, _f64p, _f64p]
        _lib = L
    return _lib


def dual3_selftest():
    return lib().orc_dual3_selftest()


def num_threads():
    return lib().orc_num_threads()


def usable_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return n


def _c(x, shape=None):
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    if shape is not None:
        assert x.shape == shape, (x.shape, shape)
    return x


def pure_derivatives(params, T, rho):
    params, T, rho = _c(params), _c(T), _c(rho)
    n = T.shape[0]
    a, p, dp = np.empty(n), np.empty(n), np.empty(n)
    lib().orc_pure_derivatives(params, T, rho, n, a, p, dp)
    return a, p, dp


def pure_helmholtz(params, T, rho):
    params, T, rho = _c(params), _c(T), _c(rho)
    a = np.empty(T.shape[0])
    lib().orc_pure_helmholtz(params, T, rho, T.shape[0], a)
    return a


def pure_vle(params, T, prec=1):
    """-> rho_v, rho_l [A^-3], status (True = failed), iters, path"""
    params, T = _c(params), _c(T)
    n = T.shape[0]
    rv, rl = np.empty(n), np.empty(n)
    st = np.empty(n, dtype=np.uint8)
    it, path = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
    lib().orc_pure_vle(params, T, n, prec, rv, rl, st, it, path)
    return rv, rl, st.astype(bool), it, path


def pure_vapor_pressure(params, T, prec=1):
    """-> p [Pa] (dense, 0 where failed), status (True = failed)"""
    params, T = _c(params), _c(T)
    n = T.shape[0]
    p = np.empty(n)
    st = np.empty(n, dtype=np.uint8)
    lib().orc_pure_vapor_pressure(params, T, n, prec, p, st)
    return p, st.astype(bool)


def pure_vapor_pressure_at(params, T, rho_v, rho_l):
    params, T, rho_v, rho_l = _c(params), _c(T), _c(rho_v), _c(rho_l)
    p = np.empty(T.shape[0])
    lib().orc_pure_vapor_pressure_at(params, T, rho_v, rho_l, T.shape[0], p)
    return p


def pure_liquid_density(params, T, p_pa, prec=1):
    """-> rho [kmol/m3] (dense), status"""
    params, T, p_pa = _c(params), _c(T), _c(p_pa)
    n = T.shape[0]
    rho = np.empty(n)
    st = np.empty(n, dtype=np.uint8)
    lib().orc_pure_liquid_density(params, T, p_pa, n, prec, rho, st)
    return rho, st.astype(bool)


def pure_liquid_density_root(params, T, p_pa, prec=1):
    """-> converged liquid density [A^-3] (dense), status"""
    params, T, p_pa = _c(params), _c(T), _c(p_pa)
    n = T.shape[0]
    rho = np.empty(n)
    st = np.empty(n, dtype=np.uint8)
    lib().orc_pure_liquid_density_root(params, T, p_pa, n, prec, rho, st)
    return rho, st.astype(bool)


def pure_equilibrium_liquid_density(params, T, prec=1):
    params, T = _c(params), _c(T)
    n = T.shape[0]
    rho = np.empty(n)
    st = np.empty(n, dtype=np.uint8)
    lib().orc_pure_equilibrium_liquid_density(params, T, n, prec, rho, st)
    return rho, st.astype(bool)


def pure_property_grad(which, params, T, p_pa, rho_v, rho_l, exact=False):
    """value[n], grad[n,10] (d/d 8 params, T, p_spec) with densities held fixed.  exact=True: long double (the formulas as
    written cancel in fp64 on strongly associating rows)."""
    params, T = _c(params), _c(T)
    n = T.shape[0]
    p_pa = _c(p_pa if p_pa is not None else np.zeros(n))
    rho_v = _c(rho_v if rho_v is not None else np.zeros(n))
    rho_l = _c(rho_l)
    val, grad = np.empty(n), np.empty((n, 10))
    (lib().orc_pure_property_grad_ld if exact else lib().orc_pure_property_grad)(
        {"vapor_pressure": 0, "liquid_density": 1, "equilibrium_liquid_density": 2}[which], params, T, p_pa, rho_v, rho_l, n, val, grad)
    return val, grad


# ------------------------------------------------------------------------------------------
# binary mixtures
# ------------------------------------------------------------------------------------------
def mix_derivatives(params, kij, T, rho, robust=False):
    """PcSaftMix.derivatives: a[n], p[n], mu[n,2], v[n,2] (reduced).  robust=False runs the
    association sub-iterations literally as the reference does; True uses the safeguarded form."""
    params, kij, T, rho = _c(params), _c(kij), _c(T), _c(rho)
    n = T.shape[0]
    a, p, mu, v = np.empty(n), np.empty(n), np.empty((n, 2)), np.empty((n, 2))
    lib().orc_mix_derivatives(params, kij, T, rho, n, int(bool(robust)), a, p, mu, v)
    return a, p, mu, v


def mix_derivatives_exact(params, kij, T, rho):
    """PcSaftMix.derivatives in long double, safeguarded association iterations, cancellation-free site fractions: the exact
    values of the model (rounded to double)."""
    params, kij, T, rho = _c(params), _c(kij), _c(T), _c(rho)
    n = T.shape[0]
    a, p, mu, v = np.empty(n), np.empty(n), np.empty((n, 2)), np.empty((n, 2))
    lib().orc_mix_derivatives_ld(params, kij, T, rho, n, a, p, mu, v)
    return a, p, mu, v


def mixn_derivatives(params, T, rho, prec=0):
    """n-component PcSaftMix.derivatives (parameters [n, nc, 8], kij = None, feos_torch/pcsaft_mix.py:395-420):
    a [n], p [n], mu [n, nc], v [n, nc].  prec=1: long double."""
    params, T, rho = _c(params), _c(T), _c(rho)
    n, nc = rho.shape
    assert params.shape == (n, nc, 8)
    a, p, mu, v = np.empty(n), np.empty(n), np.empty((n, nc)), np.empty((n, nc))
    if lib().orc_mixn_derivatives(params, T, rho, nc, n, int(prec), a, p, mu, v):
        raise Exception("Only up to two associating components are allowed, and two only for binary mixtures!")
    return a, p, mu, v


def mix_helmholtz(params, kij, T, rho):
    params, kij, T, rho = _c(params), _c(kij), _c(T), _c(rho)
    a = np.empty(T.shape[0])
    lib().orc_mix_helmholtz(params, kij, T, rho, T.shape[0], a)
    return a


def mix_bubble_dew(params, kij, T, z, p_init, dew, prec=1):
    """-> p [Pa] (dense), rho4 [n,4] = (rhoV_1, rhoV_2, rhoL_1, rhoL_2) A^-3 (dense), status"""
    params, kij, T, z, p_init = _c(params), _c(kij), _c(T), _c(z), _c(p_init)
    n = T.shape[0]
    rho4, p = np.empty((n, 4)), np.empty(n)
    st = np.empty(n, dtype=np.uint8)
    lib().orc_mix_bubble_dew(params, kij, T, z, p_init, n, int(bool(dew)), prec, rho4, p, st)
    return p, rho4, st.astype(bool)


CONT_CODES = {0: "solution", 1: "no pure-fluid VLE at either end", 2: "curve ends in a critical point", 3: "stalled (stability limit)"}


def mix_bubble_dew_continuation(params, kij, T, z, dew, prec=0):
    """The SECOND, independent bubble / dew solver (oracle/mix_continuation.hpp: continuation in composition from the
    pure-component ends, bracketed start).  -> p [Pa], rho4 [n,4], code [n] (see CONT_CODES), info [n,3] = (steps, Newton
    iterations, route)."""
    params, kij, T, z = _c(params), _c(kij), _c(T), _c(z)
    n = T.shape[0]
    rho4, p = np.empty((n, 4)), np.empty(n)
    code, info = np.empty(n, dtype=np.int32), np.empty((n, 3), dtype=np.int32)
    lib().orc_mix_bubble_dew_continuation(params, kij, T, z, n, int(bool(dew)), int(prec), rho4, p, code, info)
    return p, rho4, code, info


def mix_bubble_dew_root(params, kij, T, z, p_init, dew, prec=1):
    """-> rho4 (dense), status — the role of src/pcsaft.rs:150-231"""
    _, rho4, st = mix_bubble_dew(params, kij, T, z, p_init, dew, prec)
    return rho4, st


def mix_bubble_dew_grad(params, kij, T, rho4, dew, exact=False):
    """value[n], grad[n,19] = d/d(16 params, kij0, kij1, T) at fixed densities.  exact=True: long double (the formulas as
    written cancel in fp64 on strongly associating rows)."""
    params, kij, T, rho4 = _c(params), _c(kij), _c(T), _c(rho4)
    n = T.shape[0]
    val, grad = np.empty(n), np.empty((n, 19))
    (lib().orc_mix_bubble_dew_grad_ld if exact else lib().orc_mix_bubble_dew_grad)(params, kij, T, rho4, n, int(bool(dew)), val, grad)
    return val, grad


# ------------------------------------------------------------------------------------------
# heterosegmented gc-PC-SAFT
# ------------------------------------------------------------------------------------------
def gc_encode(segment_records, segment_lists, bond_lists, binary_segment_records):
    """Dense encoding of the reference's constructor arguments (feos_torch/gc_pcsaft.py:24-63,
    src/gc_pcsaft.rs:25-31): segment_records = [(identifier, array(8))], segment_lists /
    bond_lists per row and component, binary_segment_records = [(s1, s2, k_ab)].
    -> dict(S, seg [S,8], kab [S,S], counts [n,2,S], bonds [n,2,S,S] lower-triangular, ident)."""
    ident = [s for s, _ in segment_records]
    idx = {s: i for i, s in enumerate(ident)}
    S = len(ident)
    seg = np.array([np.asarray(v, dtype=np.float64) for _, v in segment_records]).reshape(S, 8)
    kab = np.zeros((S, S))
    for s1, s2, k in binary_segment_records:
        kab[idx[s1], idx[s2]] = float(k)
        kab[idx[s2], idx[s1]] = float(k)
    n = len(segment_lists)
    counts = np.zeros((n, 2, S))
    bonds = np.zeros((n, 2, S, S))
    for r in range(n):
        for c in range(2):
            segs = segment_lists[r][c]
            for s in segs:
                counts[r, c, idx[s]] += 1
            for i, j in bond_lists[r][c]:
                a, b = sorted((idx[segs[i]], idx[segs[j]]))[::-1]  # larger index first (:35)
                bonds[r, c, a, b] += 1
    return {"S": S, "seg": seg, "kab": kab, "counts": counts, "bonds": bonds, "ident": ident}


def gc_derivatives(enc, phi, T, rho, robust=False):
    phi, T, rho = _c(phi), _c(T), _c(rho)
    n = T.shape[0]
    a, p, mu, v = np.empty(n), np.empty(n), np.empty((n, 2)), np.empty((n, 2))
    bad = lib().orc_gc_derivatives(enc["S"], _c(enc["seg"]), _c(enc["kab"]), _c(enc["counts"]), _c(enc["bonds"]), phi, T,
                                   rho, n, int(bool(robust)), a, p, mu, v)
    if bad:
        raise Exception("Only up to one associating segment per component is allowed!")
    return a, p, mu, v


def gc_bubble_dew(enc, phi, T, z, p_init, dew, prec=1):
    """-> p [Pa], rho4 [n,4] (rhoV_1, rhoV_2, rhoL_1, rhoL_2), status"""
    phi, T, z, p_init = _c(phi), _c(T), _c(z), _c(p_init)
    n = T.shape[0]
    rho4, p = np.empty((n, 4)), np.empty(n)
    st = np.empty(n, dtype=np.uint8)
    bad = lib().orc_gc_bubble_dew(enc["S"], _c(enc["seg"]), _c(enc["kab"]), _c(enc["counts"]), _c(enc["bonds"]), phi, T,
                                  z, p_init, n, int(bool(dew)), prec, rho4, p, st)
    if bad:
        raise Exception("Only up to one associating segment per component is allowed!")
    return p, rho4, st.astype(bool)


def gc_bubble_dew_root(segment_records, segments, bonds, binary_segment_records, phi, T, z, p_init, dew, prec=1):
    """Same argument list as the reference's Rust class (src/gc_pcsaft.rs:25-31 + :71-99)."""
    enc = gc_encode(segment_records, segments, bonds, binary_segment_records)
    _, rho4, st = gc_bubble_dew(enc, phi, T, z, p_init, dew, prec)
    return rho4, st


def gc_bubble_dew_grad(enc, phi, T, rho4, dew, s1, s2, exact=False):
    """value[n], grad[n,4] = d/d(kab[s1,s2], phi_0, phi_1, T) at fixed densities (exact=True: long double)."""
    phi, T, rho4 = _c(phi), _c(T), _c(rho4)
    n = T.shape[0]
    val, grad = np.empty(n), np.empty((n, 4))
    ka, kb = enc["ident"].index(s1), enc["ident"].index(s2)
    lib().orc_gc_bubble_dew_grad(enc["S"], _c(enc["seg"]), _c(enc["kab"]), _c(enc["counts"]), _c(enc["bonds"]), phi, T, rho4,
                                 n, int(bool(dew)), ka, kb, int(bool(exact)), val, grad)
    return val, grad


def gc_segment_grad_fd(enc, phi, T, rho4, dew, weights=None, rel=1e-6):
    """[S,8] central finite differences of sum_i w_i p_i (reference formula at FIXED densities rho4,
    feos_torch/gc_pcsaft.py:470-512) w.r.t. the segment parameter table.  Entries that are structurally zero (and
    decide the model class: kappa_ab, epsilon_k_ab, na, nb, mu = 0, epsilon_k = 0) and segments no row uses are left at
    0.  Independent of the kernels' analytic chain: only the oracle's forward evaluation is used."""
    phi, T, rho4 = _c(phi), _c(T), _c(rho4)
    n = T.shape[0]
    w = np.ones(n) if weights is None else _c(weights)
    S = enc["S"]
    used = enc["counts"].sum(axis=(0, 1)) > 0
    ident = enc["ident"]

    def f(seg):
        e = dict(enc)
        e["seg"] = seg
        val, _ = gc_bubble_dew_grad(e, phi, T, rho4, dew, ident[0], ident[0])
        return float((w * val).sum())

    grad = np.zeros((S, 8))
    base = _c(enc["seg"]).copy()
    for a in range(S):
        if not used[a]:
            continue
        for k in range(8):
            if base[a, k] == 0.0:
                continue
            h = rel * abs(base[a, k])
            sp, sm = base.copy(), base.copy()
            sp[a, k] += h
            sm[a, k] -= h
            grad[a, k] = (f(sp) - f(sm)) / (2 * h)
    return grad
