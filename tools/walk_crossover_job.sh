#!/usr/bin/env bash
# Developer tool: four-wave walk against 8 / 4 / 2 groups per wave of bh_walk_kernel at the sizes around their crossover.
set -uo pipefail
out=gpurun_out/walk_crossover.txt; : > $out
for spec in "16384 1.0" "24576 1.0" "32768 1.0" "49152 1.0" "65536 1.0" "131072 1.0" "262144 1.0" "12288 0.5" "16384 0.5" "24576 0.5" "32768 0.5" "65536 0.5"; do
  set -- $spec
  for mode in quad g8 g4 g2; do
    case $mode in
      quad) export NBODY_BH_QUAD_MAX=100000000; unset NBODY_BH_WALK_GROUPS;;
      g8) export NBODY_BH_QUAD_MAX=0 NBODY_BH_WALK_GROUPS=8;;
      g4) export NBODY_BH_QUAD_MAX=0 NBODY_BH_WALK_GROUPS=4;;
      g2) export NBODY_BH_QUAD_MAX=0 NBODY_BH_WALK_GROUPS=2;;
    esac
    if [ $mode = quad ] && [ $1 -gt 70000 ]; then continue; fi
    echo -n "$mode " >> $out
    timeout -k 10 120 python3 tools/bh_steady.py $1 $2 300 2>&1 | grep warm >> $out || exit 1
  done
done
cat $out
