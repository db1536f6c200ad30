#!/usr/bin/env bash
set -uo pipefail
for cfg in "" "NBODY_SORT_W8=1" "NBODY_SORT_E=16" "NBODY_SORT_E=16 NBODY_SORT_W8=1" "NBODY_SORT_HB=128" "NBODY_SORT_HB=64"; do
  echo "=== 1M $cfg"; env $cfg tools/ktrace.sh e1 -- python3 tools/bh_pmc.py 1000000 1.0 --reps 5 | grep -E "sort_"
done
for cfg in "" "NBODY_SORT_E=8" "NBODY_SORT_W8=1" "NBODY_SORT_E=8 NBODY_SORT_W8=1"; do
  echo "=== 4M $cfg"; env $cfg tools/ktrace.sh e4 -- python3 tools/bh_pmc.py 4000000 1.0 --reps 3 | grep -E "sort_"
done
