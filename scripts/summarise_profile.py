#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (rocprofv3 csv output of scripts/profile.sh) into
profiles/<tag>_summary.md + profiles/<tag>_kernel_stats.csv, and refresh
profiles/pmc_traffic.json (HBM bytes per k_pure_vle launch, read by bench.py).

HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB, collected in
separate --pmc passes; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced
streaming read, so it is doubled; WRITE_SIZE is taken as is.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


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


stats = max(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)  # newest: gpurun merges runs
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
lines = [f"# rocprofv3 summary `{tag}`", "",
         "Command (scripts/profile.sh): `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 "
         "--no-cpu-baseline` (1e7 rows per launch), PMC counters in separate passes.", "",
         "## Kernel time (`--kernel-trace --stats`)", "", "| kernel | calls | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|"]
for r in rows:
    if float(r["Percentage"]) < 0.05:
        continue
    lines.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs'])/1e6:.4f} | {float(r['MinNs'])/1e6:.4f} | "
                 f"{float(r['MaxNs'])/1e6:.4f} | {r['Percentage']} |")
allc = {}
meta = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    a, m = counters(sub)
    for (k, c), v in a.items():
        allc.setdefault(k, {})[c] = sum(v) / len(v)
    meta.update(m)
traffic = None
valu_instr = None
valu_busy = None
for k, c in allc.items():
    if "k_pure" not in k and "k_mix" not in k and "k_gc" not in k:
        continue
    lines += ["", f"## {short(k)}", "", f"launch: {meta.get(k)}", "", "| counter (mean per launch) | value |", "|---|---|"]
    for name in sorted(c):
        lines.append(f"| {name} | {c[name]:.6g} |")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        hbm = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        lines.append(f"| **HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024** | {hbm:.6g} |")
        if "k_pure_vle(" in k or "k_pure_vle<" in k:
            traffic = hbm
    if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
        lines.append(f"| VALU instructions per wave (= per state point) | {c['SQ_INSTS_VALU']/c['SQ_WAVES']:.1f} |")
        if "k_pure_vle(" in k or "k_pure_vle<" in k:
            valu_instr = c["SQ_INSTS_VALU"]
    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        # SQ_ACTIVE_INST_* count quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over 8 XCDs
        simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        lines.append(f"| VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE/8) | {c['SQ_ACTIVE_INST_VALU']*4/simd_cycles:.3f} |")
        if "k_pure_vle(" in k or "k_pure_vle<" in k:
            # the two counters come from different passes (different launches): the ratio can come out a fraction of a
            # per cent above 1 for a kernel that keeps the VALU busy all the time
            valu_busy = min(1.0, c["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles)
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
if traffic is not None:
    json.dump({"tag": tag, "kernel": "k_pure_vle", "rows": 10_000_000, "hbm_bytes_per_launch": traffic,
               "valu_wave_instr_per_launch": valu_instr, "valu_busy": valu_busy,
               "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes, gfx950 FETCH correction"},
              open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print("\n".join(lines))
