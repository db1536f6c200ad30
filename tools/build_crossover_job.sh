#!/usr/bin/env bash
# Developer tool: the one-workgroup-per-phase small build (fused) against the large pipeline around their crossover.
set -uo pipefail
out=gpurun_out/build_crossover.txt; : > $out
for n in 8192 12288 16384 20480 24576 32768; do
  for mode in fused large; do
    if [ $mode = large ]; then export NBODY_BH_NO_FUSE=1; else unset NBODY_BH_NO_FUSE; fi
    echo -n "$mode " >> $out
    timeout -k 10 120 python3 tools/bh_steady.py $n 1.0 300 2>&1 | grep warm >> $out || exit 1
  done
done
cat $out
