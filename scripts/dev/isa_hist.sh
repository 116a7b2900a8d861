#!/bin/bash
# scripts/dev/isa_hist.sh <unit.hip> <mangled kernel prefix> [flags...]: opcode histogram and per-basic-block VALU / mov / fp64 counts of one kernel
set -e
SRC=$1; KER=$2; shift; shift
mkdir -p scratch/isa
hipcc -O3 --offload-arch=gfx950 -std=c++17 --cuda-device-only -S "$@" -o scratch/isa/unit.s $SRC 2>/dev/null
python3 - scratch/isa/unit.s "$KER" <<'PY'
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and l.rstrip().endswith(tuple(":")) or (l.startswith(sys.argv[2]) and ": " in l)][0]
end = [i for i, l in enumerate(lines) if i > start and l.strip().startswith(".amdhsa_kernel")][0]
body = lines[start:end]
open("scratch/isa/kernel.s", "w").write("\n".join(body))
c = collections.Counter()
for l in body:
    s = l.strip()
    if s[:2] in ("v_", "s_", "ds") or s.startswith(("global_", "scratch_")):
        c[s.split()[0]] += 1
print("valu", sum(v for k, v in c.items() if k.startswith("v_")), "salu", sum(v for k, v in c.items() if k.startswith("s_")))
for k, v in c.most_common(40):
    print(f"{v:6d} {k}")
blocks = []; cur = ["entry", 0, 0, 0]
for l in body:
    s = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        blocks.append(cur); cur = [m.group(1), 0, 0, 0]; continue
    if s.startswith("v_"):
        cur[1] += 1
        if s.startswith(("v_mov", "v_pk_mov")): cur[2] += 1
        if "_f64" in s.split()[0]: cur[3] += 1
blocks.append(cur)
print("blocks >= 60 VALU: [label, valu, mov, f64]")
for b in blocks:
    if b[1] >= 60: print("  ", b)
PY
