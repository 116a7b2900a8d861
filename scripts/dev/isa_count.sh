#!/bin/bash
# scripts/dev/isa_count.sh <unit.hip> [flags...]: static fp64 / VALU instruction counts and register use of every
# device function of a translation unit (a proxy for the cost of the non-inlined evaluation functions).
set -e
SRC=$1; shift
OUT=$(mktemp -d)
hipcc -O3 --offload-arch=gfx950 -std=c++17 --cuda-device-only -S "$@" -o $OUT/unit.s $SRC
python3 - $OUT/unit.s <<'PY'
import re, sys
name, rows, cur = None, [], None
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = {"name": m.group(1), "f64": 0, "valu": 0, "scratch": 0, "acc": 0}
        rows.append(cur)
        continue
    if cur is None: continue
    s = line.strip()
    if s.startswith("v_"):
        cur["valu"] += 1
        if re.match(r"v_(fma|fmac|mul|add)_f64", s): cur["f64"] += 1
        if s.startswith("v_accvgpr"): cur["acc"] += 1
    if s.startswith("scratch_"): cur["scratch"] += 1
    m = re.match(r"; NumVgprs: (\d+)", s)
    if m: cur["vgpr"] = int(m.group(1))
    m = re.match(r"; NumAgprs: (\d+)", s)
    if m: cur["agpr"] = int(m.group(1))
for r in rows:
    print(f'{r["f64"]:6d} f64  {r["valu"]:6d} valu  {r["scratch"]:5d} scratch  {r["acc"]:4d} accmov  vgpr {r.get("vgpr","?")} agpr {r.get("agpr","?")}  {r["name"][:110]}')
PY
rm -rf $OUT
