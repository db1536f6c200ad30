#!/usr/bin/env bash
# Round-3 evidence for the Barnes-Hut configs: bench lines (with the unprobed reference as cpu_baseline) and kernel stats.
set -uo pipefail
export TMPDIR=/tmp
timeout -k 10 200 python3 bench.py --workload bh --bodies 10000 --steps 5 > gpurun_out/r3_bh_n10k_theta1_bench.json 2> gpurun_out/r3_bh10k.err || exit 1
timeout -k 10 200 python3 bench.py --workload bh --bodies 65536 --steps 5 > gpurun_out/r3_bh_n64k_theta1_bench.json 2>> gpurun_out/r3_bh10k.err || exit 1
timeout -k 10 300 python3 bench.py --workload bh --bodies 1000000 --steps 8 > gpurun_out/r3_bh_n1m_theta1_bench.json 2> gpurun_out/r3_bh1m.err || exit 1
timeout -k 10 300 python3 bench.py --workload bh --bodies 1000000 --steps 8 --bh-sorted > gpurun_out/r3_bh_n1m_theta1_sorted_bench.json 2>> gpurun_out/r3_bh1m.err || exit 1
timeout -k 10 300 python3 bench.py --workload bh --bodies 1000000 --steps 8 --theta 0.5 > gpurun_out/r3_bh_n1m_theta05_bench.json 2>> gpurun_out/r3_bh1m.err || exit 1
for n in 10000 1000000; do
  rm -rf gpurun_out/kt_bh$n; mkdir -p gpurun_out/kt_bh$n
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_bh$n -- python3 bench.py --workload bh --bodies $n --no-cpu-baseline > gpurun_out/r3_bh_n${n}_bench_under_rocprof.json 2> gpurun_out/kt_bh$n/stderr.txt || exit 1
  cp "$(find gpurun_out/kt_bh$n -name '*kernel_stats.csv' | head -1)" gpurun_out/r3_bh_n${n}_theta1_kernel_stats.csv
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bh_n*_bench.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; v=d.get("roofline_valu",{}); c=d.get("cpu_baseline") or {}
    print(f.split("/")[-1], "ms/step %.3f"%d["value"], "walk %.3f build %.3f"%(r["kernel_ms"], r["build_ms"]), r["kernel"], "hbm frac %.3f valu frac %.3f"%(r["frac"], v.get("frac",0)), "cpu %s"%c.get("value"))
PY
