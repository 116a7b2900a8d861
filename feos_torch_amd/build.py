"""Build ``libpcsaft_hip.so`` for gfx950 with hipcc (in-tree, so it travels with the repo)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpcsaft_hip.so")
# (source, object, extra flags).  pure_kernels.hip is compiled twice: part 1 with re-association (pressure-only VLE kernel,
# Jacobians, C ABI), part 2 without (all-fp64 VLE kernel, liquid density): see the head of the file
SOURCES = [("pure_kernels.hip", "pure_kernels.o", ["-DPCS_PURE_PART=1"]), ("pure_kernels.hip", "pure_kernels_b.o", ["-DPCS_PURE_PART=2"]),
           ("pure_robust.hip", "pure_robust.o", []), ("compact_kernels.hip", "compact_kernels.o", []),
           ("mix_kernels.hip", "mix_kernels.o", []), ("mixn_kernels.hip", "mixn_kernels.o", []),
           ("gc_kernels.hip", "gc_kernels.o", []), ("gc_gradient.hip", "gc_gradient.o", [])]
# -fno-honor-nans/-infinities/-signed-zeros: lets the compiler fold the structural zeros of the dual
# numbers (0 * x, x + 0); every NaN/inf test in the kernels is a bit test (is_finite_bits), so the
# failure detection does not depend on IEEE comparison semantics.  Measured on k_pure_vle: x1.065,
# x1.158 together with the v_rcp_f64 + Newton reciprocal (scratch A/B, 1e7 rows, MI355X).
# -DPCS_F32_PRESOLVE: fp32 initialiser + first Newton iterations of the VLE (csrc/pure_f32.hpp): x1.16
# on k_pure_vle and the robust-pass list shrinks 82,010 -> 184 rows per 1e7.
# They are applied to the pure-component translation unit only (the headline kernel); the
# mixture / gc solvers keep strict IEEE comparisons.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]
# -fno-slp-vectorize: the SLP vectoriser pairs fp32 operations into v_pk_fma_f32 / v_pk_mul_f32 and pays for it with
# ~25 % v_mov (operand pairs must sit in adjacent registers) and 168 instead of 122 VGPRs on k_pure_vle<true> (3 instead of
# 4 waves per SIMD): 1.184 -> 0.943 ms per 1e7 rows without it (scripts/dev/ab_bench.py, round 2).
# REASSOC = -fassociative-math -freciprocal-math (round 3): the compiler may re-associate sums / products and turn x / y into
# x * (1 / y): k_pure_vle<true> 0.783 -> 0.750 ms per 1e7 rows (x1.044, scripts/dev/ab_bench.py; -ffast-math as a whole gives
# the same), Jacobian kernels x1.13; the all-fp64 VLE kernel and the liquid-density kernel get SLOWER with it (x0.96, x0.75),
# hence the two parts.  The explicit fma chains of the logarithm / reciprocal refinements are untouched, and the parity of
# the kernels against the long-double oracle is unchanged (tests/test_large_parity_gpu.py: 1e-10 on 1e6 rows).
RELAXED = ["-fno-honor-nans", "-fno-honor-infinities", "-fno-signed-zeros", "-fno-slp-vectorize", "-DPCS_FAST_RCP", "-DPCS_FAST_LOG",
           "-DPCS_F32_PRESOLVE"]
REASSOC = ["-fassociative-math", "-freciprocal-math"]  # part 1 of pure_kernels.hip only
RELAXED_SOURCES = {"pure_kernels.hip"}
# mixture / gc solver units: the short logarithm and the refined hardware reciprocal in their guarded forms (=2: IEEE
# results for zero, infinite, NaN and negative arguments, which the solvers' failure detection relies on; dual.hpp).
# 1e6 rows, MI355X: bubble 3.21 -> 2.92 ms, dew 7.07 -> 6.71 ms.
# + re-association (round 3; no reciprocal-math here: it was neutral to slower): bubble 2.415 -> 2.301 ms, gc bubble 1.81 ->
# 1.74 ms, gc dew 4.28 -> 4.17 ms, dew unchanged (scripts/dev/ab_mix.py / ab_gc.py); the status masks are identical and the
# results move by <= 6.4e-13 relative.  NaN / infinity semantics stay IEEE (no -fno-honor-*), which the failure detection needs.
GUARDED = ["-DPCS_FAST_LOG=2", "-DPCS_FAST_RCP=2", "-fassociative-math", "-fno-signed-zeros", "-fno-trapping-math"]
GUARDED_SOURCES = {"mix_kernels.hip", "gc_kernels.hip"}
RESOURCES = os.path.join(HERE, "build", "resources.json")  # per-kernel register / stack report of the last build


def _stale():
    if not os.path.exists(OUT) or not os.path.exists(RESOURCES):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "pcsaft_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def _parse_resources(text):
    """kernel-resource-usage remarks of one hipcc run -> {mangled name: {vgpr, agpr, scratch, occupancy}}"""
    import re

    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r"remark:\s+VGPRs: (\d+)"), ("agpr", r"remark:\s+AGPRs: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return out


def _demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def build(force=False, verbose=False):
    if not force and not _stale():
        return OUT
    # one hipcc per translation unit in parallel, then link
    objs, procs = [], []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s, oname, extra in SOURCES:
        obj = os.path.join(HERE, "build", oname)
        cmd = (["hipcc"] + FLAGS + (RELAXED if s in RELAXED_SOURCES else []) + (REASSOC if "-DPCS_PURE_PART=1" in extra else [])
               + (GUARDED if s in GUARDED_SOURCES else []) + extra + ["-c", "-o", obj, os.path.join(CSRC, s)])
        if verbose:
            print(" ".join(cmd))
        # the compiler's per-kernel register / stack report is kept next to the objects (tests/test_abi.py guards the
        # stack frames: above ~2.5 KB per lane the runtime throttles the resident waves)
        procs.append(subprocess.Popen(cmd + ["-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, text=True))
        objs.append(obj)
    resources = {}
    for p in procs:
        _, err = p.communicate()
        if p.returncode != 0:
            print(err)
            raise subprocess.CalledProcessError(p.returncode, "hipcc")
        resources.update(_parse_resources(err))
        rest = [ln for ln in err.splitlines() if "kernel-resource-usage" not in ln and ("warning" in ln or "error" in ln)]
        if rest:
            print("\n".join(rest))
    names = _demangle(list(resources))
    import json

    with open(RESOURCES, "w") as f:
        json.dump({names[k].replace("(anonymous namespace)::", "").split("(")[0]: v for k, v in resources.items()}, f, indent=1, sort_keys=True)
    link = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return OUT


HOST_SRC = os.path.join(HERE, "csrc_host", "gc_encode.cpp")
HOST_OUT = os.path.join(HERE, "_gc_encode.so")


def build_host(force=False, verbose=False):
    """CPython extension with the native gc row encoder (host code, g++)."""
    if not force and os.path.exists(HOST_OUT) and os.path.getmtime(HOST_OUT) >= os.path.getmtime(HOST_SRC):
        return HOST_OUT
    import sysconfig

    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", sysconfig.get_paths()["include"], "-o", HOST_OUT, HOST_SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HOST_OUT


if __name__ == "__main__":
    build(force=True, verbose=True)
    build_host(force=True, verbose=True)
