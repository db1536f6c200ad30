#!/usr/bin/env bash
# Developer tool: same-box A/B of two builds of the library on the headline bench (alternating, so clock drift shows).
#   tools/ab_libs.sh OUT.txt LIB_A LIB_B [bench args...]
set -uo pipefail
out=$1; a=$2; b=$3; shift 3
: > "$out"
for rep in 1 2 3; do
  for lib in "$a" "$b"; do
    echo "== $lib" >> "$out"
    NBODY_AMD_LIB="$PWD/$lib" timeout -k 10 120 python3 bench.py --no-cpu-baseline "$@" >> "$out" 2>&1 || { echo "FAILED $lib" >> "$out"; exit 1; }
  done
done
python3 - "$out" <<'PY'
import json,sys,collections
cur=None; d=collections.defaultdict(list)
for l in open(sys.argv[1]):
    if l.startswith("== "): cur=l[3:].strip()
    elif l.startswith("{"):
        j=json.loads(l); r=j["roofline"]; d[cur].append((r["kernel_ms"], r["frac"], (r.get("clock") or {}).get("gfx_mhz"), r.get("frac_at_measured_clock"), j["ms_per_step"]))
for k,v in d.items():
    print(k, " | ".join(f"{x[0]:.3f} ms frac {x[1]:.4f} clk {x[2]} f@clk {x[3]}" for x in v))
PY
