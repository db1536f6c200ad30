#!/usr/bin/env bash
# Developer tool: rocprofv3 kernel trace + stats of one program; prints the per-kernel summary.
#   tools/ktrace.sh TAG -- python3 program.py args...
set -uo pipefail
tag=$1; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/kt_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL ${KT_TIMEOUT:-200} rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- "$@" > "$out/stdout.txt" 2> "$out/stderr.txt"
rc=$?
if [ $rc -ne 0 ]; then echo "ktrace $tag: rc=$rc"; tail -5 "$out/stderr.txt"; exit $rc; fi
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
python3 - "$f" > "$root/gpurun_out/kt_$tag.txt" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    name = re.sub(r"\(.*$", "", r["Name"]).replace("nbody::", "").replace("void ", "")
    print(f"{name[:90]:90s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:10.2f} us  total {float(r['TotalDurationNs'])/1e3:12.1f} us  {float(r['Percentage']):6.2f} %")
PY
cat "$root/gpurun_out/kt_$tag.txt"
