#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (rocprofv3 csv output of scripts/profile.sh) into
profiles/<tag>_summary.md + profiles/<tag>_kernel_stats.csv + profiles/<tag>_issue_cost.jsonl, and refresh
profiles/pmc_traffic.json (per k_pure_vle launch: HBM bytes, VALU instruction mix, VALU issue fraction; read by bench.py).

HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB, collected in
separate --pmc passes; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced
streaming read, so it is doubled; WRITE_SIZE is taken as is.

Kernel time: from the per-dispatch kernel trace of the driver's command (bench.py --steps 20 --warmup 5), the
mean over the TIMED launches only (the warm-up launches, the first of which is cold, are dropped).

VALU roofline: valu_issue_frac = sum_class N_class x t_class / (SIMDs x kernel time), N_class = wave-instructions of the
class per launch (PMC, dynamic), t_class = time one SIMD needs per wave-instruction of the class at the kernel's
occupancy (scripts/microbench/issue_cost.hip, 3 waves per SIMD).  A number <= 1 by construction of the costs: the
fraction of the SIMDs' issue time the counted instructions account for.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
N_SIMD = 1024
WARMUP, STEPS = 5, 20


def counters(sub):
    acc = collections.defaultdict(list)
    meta = {}
    files = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    if not files:
        return acc, meta
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        k = r["Kernel_Name"]
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        meta[k] = {x: r[x] for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                     "Accum_VGPR_Count", "SGPR_Count")}
    return acc, meta


def short(k):
    k = k.replace("(anonymous namespace)::", "")
    return k.split("(")[0][:60]


def is_headline(k):
    return "k_pure_vle<true, false>" in k or "k_pure_vle<true>" in k or "k_pure_vle<(bool)1, (bool)0>" in k


stats = max(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)  # newest: gpurun merges runs
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
lines = [f"# rocprofv3 summary `{tag}`", "",
         f"Command (scripts/profile.sh): `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {STEPS} --warmup {WARMUP} "
         "--no-cpu-baseline --no-extra` (1e7 rows per launch; the driver's step counts), PMC counters in separate passes "
         "(`--steps 5 --warmup 1`).", "",
         "## Kernel time (`--kernel-trace --stats`, all launches incl. warm-up)", "", "| kernel | calls | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|"]
for r in rows:
    if float(r["Percentage"]) < 0.05:
        continue
    lines.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs'])/1e6:.4f} | {float(r['MinNs'])/1e6:.4f} | "
                 f"{float(r['MaxNs'])/1e6:.4f} | {r['Percentage']} |")

# per-dispatch trace: timed launches only
kernel_ms = None
traces = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
if traces:
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(max(traces, key=os.path.getmtime))):
        per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    lines += ["", f"## Timed launches only (the last {STEPS} of each kernel; warm-up dropped)", "",
              "| kernel | launches | mean ms | min ms | max ms |", "|---|---|---|---|---|"]
    for k, v in per.items():
        if not ("k_pure" in k):
            continue
        v.sort()
        d = [(e - s) / 1e6 for s, e in v[-STEPS:]]
        lines.append(f"| {short(k)} | {len(d)} | {sum(d)/len(d):.4f} | {min(d):.4f} | {max(d):.4f} |")
        if is_headline(k):
            kernel_ms = sum(d) / len(d)
    # step time as the trace sees it: start of k_pure_vle<true> of timed step i to end of the robust kernel of that step
    hk = [k for k in per if is_headline(k)]
    rk = [k for k in per if "k_pure_vle_robust" in k]
    if hk and rk:
        a, b = sorted(per[hk[0]])[-STEPS:], sorted(per[rk[0]])[-STEPS:]
        span = (b[-1][1] - a[0][0]) / 1e6 / STEPS
        lines += ["", f"Trace span of the {STEPS} timed steps (first main-kernel start to last robust-kernel end) / {STEPS} = "
                  f"**{span:.4f} ms per step** — to be compared with `ms_per_step` of the bench line."]

allc = {}
meta = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_mix", "pmc_mix2"):
    a, m = counters(sub)
    for (k, c), v in a.items():
        allc.setdefault(k, {})[c] = sum(v) / len(v)
    meta.update(m)

# issue costs (ns per wave-instruction per SIMD) from the microbenchmark
cost = {}
ic = os.path.join(src, "issue_cost.jsonl")
if os.path.exists(ic):
    shutil.copy(ic, os.path.join(dst, f"{tag}_issue_cost.jsonl"))
    for ln in open(ic):
        ln = ln.strip()
        if ln.startswith("{"):
            d = json.loads(ln)
            cost[d["instr"]] = d["ns_per_wave_instr_per_simd"]

traffic = valu_instr = valu_busy = None
mix_out = None
for k, c in allc.items():
    if "k_pure" not in k and "k_mix" not in k and "k_gc" not in k:
        continue
    lines += ["", f"## {short(k)}", "", f"launch: {meta.get(k)}", "", "| counter (mean per launch) | value |", "|---|---|"]
    for name in sorted(c):
        lines.append(f"| {name} | {c[name]:.6g} |")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        hbm = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        lines.append(f"| **HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024** | {hbm:.6g} |")
        if is_headline(k):
            traffic = hbm
    if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
        lines.append(f"| VALU instructions per wave (= per state point) | {c['SQ_INSTS_VALU']/c['SQ_WAVES']:.1f} |")
        if is_headline(k):
            valu_instr = c["SQ_INSTS_VALU"]
    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        # SQ_ACTIVE_INST_* count quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over 8 XCDs
        simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * N_SIMD
        # raw ratio, not clamped: it came out at 1.17 in round 2, i.e. the unit assumed for SQ_ACTIVE_INST_VALU (quad-cycles) or
        # the clock behind GRBM_GUI_ACTIVE is off for this kernel -- the figure is printed for the record and NOT used
        lines.append(f"| (unreliable) SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE/8) | {c['SQ_ACTIVE_INST_VALU']*4/simd_cycles:.3f} |")
    if is_headline(k) and "SQ_INSTS_VALU_FMA_F64" in c and cost and kernel_ms:
        total = c.get("SQ_INSTS_VALU", valu_instr)
        f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"]
        f32 = c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_FMA_F32"]
        t64, t32 = c["SQ_INSTS_VALU_TRANS_F64"], c["SQ_INSTS_VALU_TRANS_F32"]
        cvt = c.get("SQ_INSTS_VALU_CVT", 0.0)
        i64 = c.get("SQ_INSTS_VALU_INT64", 0.0)
        other = max(0.0, total - f64 - f32 - t64 - t32 - cvt)  # int32 / logic / moves / selects / compares (32-bit issue rate)
        # every opcode class at ITS measured issue cost (round 2 priced add / mul at the FMA cost: 0.92 instead of 0.89)
        classes = [
            ("fp64 fma", c["SQ_INSTS_VALU_FMA_F64"], cost["v_fma_f64"]),
            ("fp64 mul", c["SQ_INSTS_VALU_MUL_F64"], cost["v_mul_f64"]),
            ("fp64 add", c["SQ_INSTS_VALU_ADD_F64"], cost["v_add_f64"]),
            ("fp64 transcendental (rcp, rsq, sqrt)", t64, cost["v_rcp_f64"]),
            ("fp32 fma", c["SQ_INSTS_VALU_FMA_F32"], cost["v_fma_f32"]),
            ("fp32 mul", c["SQ_INSTS_VALU_MUL_F32"], cost["v_mul_f32"]),
            ("fp32 add", c["SQ_INSTS_VALU_ADD_F32"], cost["v_add_f32"]),
            ("fp32 transcendental (rcp, sqrt, exp, log)", t32, cost["v_rcp_f32"]),
            ("conversions", cvt, max(cost.get("v_cvt_f32_f64", 0.0), cost.get("v_cvt_f64_f32", 0.0))),
            ("other VALU (int, logic, moves, selects, compares)", other, cost["v_mov_b32"]),
        ]
        issue_ns = sum(n * t for _, n, t in classes)
        simd_ns = N_SIMD * kernel_ms * 1e6
        frac = issue_ns / simd_ns
        lines += ["", f"### VALU issue roofline of {short(k)}", "",
                  "| class | wave-instr per launch (PMC) | share | ns per wave-instr per SIMD (issue_cost.hip, 3 waves/SIMD) | SIMD-ns |",
                  "|---|---|---|---|---|"]
        for name, n, t in classes:
            lines.append(f"| {name} | {n:.4g} | {n/total:.3f} | {t:.4f} | {n*t:.4g} |")
        lines += [f"| **sum** | {total:.4g} | 1 | | {issue_ns:.4g} |", "",
                  f"valu.frac = {issue_ns:.4g} SIMD-ns of issue / ({N_SIMD} SIMDs x {kernel_ms:.4f} ms) = **{frac:.3f}**"
                  f"  (64-bit integer VALU inside `other`: {i64:.3g})"]
        mix_out = {"classes": [{"class": a, "wave_instr": n, "ns_per_wave_instr_per_simd": t} for a, n, t in classes],
                   "issue_ns": issue_ns, "simd_ns": simd_ns, "frac": frac}
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
if traffic is not None:
    out = {"tag": tag, "kernel": "k_pure_vle<true, false>", "rows": 10_000_000, "hbm_bytes_per_launch": traffic,
           "valu_wave_instr_per_launch": valu_instr, "kernel_ms_timed_launches": kernel_ms,
           "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes, gfx950 FETCH correction"}
    if mix_out:
        out.update({"valu_issue_frac": mix_out["frac"], "valu_issue_cycles_per_launch": mix_out["issue_ns"],
                    "simd_cycles_per_launch": mix_out["simd_ns"], "valu_mix": mix_out["classes"],
                    "valu_issue_unit": "SIMD-nanoseconds (issue_cost.hip wall-clock costs)"})
    json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print("\n".join(lines))
