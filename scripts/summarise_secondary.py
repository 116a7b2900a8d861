#!/usr/bin/env python3
"""Condense gpurun_out/prof_sec_<tag>/ (scripts/profile_secondary.sh) into profiles/<tag>_secondary_pmc.md +
profiles/<tag>_secondary_kernel_stats.csv: per-kernel time of the mixture / gc bubble-dew launches (1e6 rows) and their PMC
counters (mean per launch), with the derived figures the DESIGN quotes: VALU instructions per row, share of the wave cycles
issuing VALU / scalar instructions / waiting, average active lanes per VALU instruction."""
import collections, csv, glob, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_sec_{tag}")
dst = os.path.join(ROOT, "profiles")
ROWS = int(os.environ.get("PCS_ROWS", 1_000_000))


def short(k):
    return k.replace("(anonymous namespace)::", "").split("(")[0][:70]


stats = max(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(stats, os.path.join(dst, f"{tag}_secondary_kernel_stats.csv"))
lines = [f"# rocprofv3 summary `{tag}`: mixture / gc bubble-dew solvers, {ROWS} rows per launch", "",
         "Command (scripts/profile_secondary.sh): `rocprofv3 --kernel-trace --stats -- python3 scripts/dev/run_secondary.py all 3`, PMC "
         "counters in separate passes of the same command.", "",
         "## Kernel time (`--kernel-trace --stats`)", "", "| kernel | calls | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|"]
for r in csv.DictReader(open(stats)):
    if float(r["Percentage"]) < 0.2:
        continue
    lines.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs'])/1e6:.4f} | {float(r['MinNs'])/1e6:.4f} | "
                 f"{float(r['MaxNs'])/1e6:.4f} | {r['Percentage']} |")
allc = {}
for sub in ("pmc_sq", "pmc_sq2", "pmc_mix"):
    files = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    if not files:
        continue
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        per[(r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    acc = collections.defaultdict(list)
    for (k, d, c), v in per.items():
        acc[(k, c)].append(v)
    for (k, c), v in acc.items():
        allc.setdefault(k, {})[c] = sum(v) / len(v)
for k, c in sorted(allc.items()):
    if not any(s in k for s in ("k_mix_", "k_gc_bubble")) or "class" in k:
        continue
    lines += ["", f"## {short(k)}", "", "| counter (mean per launch) | value |", "|---|---|"]
    for name in sorted(c):
        lines.append(f"| {name} | {c[name]:.6g} |")
    d = []
    if "SQ_INSTS_VALU" in c:
        d.append(f"VALU wave-instructions per row = {c['SQ_INSTS_VALU'] * 64 / ROWS:.0f} lane-slots / 64 = {c['SQ_INSTS_VALU'] / ROWS:.1f} per row (x64 lanes)")
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c:
        # both in quad-cycles summed over waves
        d.append(f"share of the wave cycles issuing VALU = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:.3f}")
    if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
        d.append(f"waiting for an instruction to be issuable = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f}")
    if "SQ_INSTS_SALU" in c and "SQ_INSTS_VALU" in c:
        d.append(f"scalar per vector instruction = {c['SQ_INSTS_SALU'] / c['SQ_INSTS_VALU']:.3f}")
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in allc.get(k, {}):
        d.append(f"average active lanes per VALU cycle = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 4) = {c['SQ_THREAD_CYCLES_VALU'] / (c['SQ_ACTIVE_INST_VALU'] * 4):.1f} of 64")
    f64 = sum(c.get(x, 0.0) for x in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
    if f64 and "SQ_INSTS_VALU" in c:
        d.append(f"fp64 add/mul/fma share of the VALU instructions = {f64 / c['SQ_INSTS_VALU']:.3f}, fp64 transcendental {c.get('SQ_INSTS_VALU_TRANS_F64', 0.0) / c['SQ_INSTS_VALU']:.4f}")
    lines += [""] + [f"* {x}" for x in d]
open(os.path.join(dst, f"{tag}_secondary_pmc.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
