#!/usr/bin/env bash
# Developer tool: kernel traces of the Barnes-Hut build with several builds of the library (NBODY_AMD_LIB).
#   tools/sort_w_job.sh TAG:LIB ... -- N [N ...]
set -uo pipefail
libs=(); while [ "$1" != "--" ]; do libs+=("$1"); shift; done; shift
for n in "$@"; do
  for tl in "${libs[@]}"; do
    tag=${tl%%:*}; lib=${tl#*:}
    export NBODY_AMD_LIB="$PWD/$lib"
    if [ "$n" -ge 16000000 ]; then args="$n 0.5 --plummer --world 8 --reps 2"; else args="$n 1.0 --reps 4"; fi
    bash tools/ktrace.sh ${tag}_$n -- python3 tools/bh_pmc.py $args > /dev/null || exit 1
    echo "== $tag N=$n"; grep -E "sort_|bh_cells|bh_nodes|bh_com|bh_tie|bh_merge|bh_keys" gpurun_out/kt_${tag}_$n.txt | cut -c1-40,90-140
  done
done
