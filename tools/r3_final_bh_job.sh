#!/usr/bin/env bash
# End of round 3: the Barnes-Hut bench lines and kernel statistics that the documents quote, after the walk kernel's rewrite.
set -uo pipefail
o=gpurun_out/final; mkdir -p $o
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $o/$name.json 2> $o/$name.err || { echo "FAILED $name"; tail -3 $o/$name.err; exit 1; }; python3 - $o/$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1], "ms/step", round(d["ms_per_step"],4), r["kernel"], round(r["kernel_ms"],4), "build", round(r.get("build_ms",0),4), "valu", round(d.get("roofline_valu",{}).get("frac",0),3), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
}
run r3_bh_n1m_theta1_bench --workload bh --bodies 1000000 --steps 8 --warmup 2 &&
run r3_bh_n1m_theta05_bench --workload bh --bodies 1000000 --theta 0.5 --steps 8 --warmup 2 &&
run r3_bh_n1m_theta1_sorted_bench --workload bh --bodies 1000000 --bh-sorted --steps 8 --warmup 2 &&
run r3_bh_n64k_theta1_bench --workload bh --bodies 65536 --steps 50 --warmup 5 &&
run r3_bh_n10k_theta1_bench --workload bh --bodies 10000 --steps 200 --warmup 20 &&
KT_TIMEOUT=200 tools/ktrace.sh bh1m_final -- python3 bench.py --workload bh --bodies 1000000 --steps 8 --warmup 2 --no-cpu-baseline &&
cp gpurun_out/kt_bh1m_final/stdout.txt $o/r3_bh_n1m_theta1_bench_under_rocprof.json &&
cp "$(find gpurun_out/kt_bh1m_final -name '*kernel_stats.csv' | head -1)" $o/r3_bh_n1m_theta1_kernel_stats.csv
