#!/usr/bin/env bash
# Developer tool: ONE rocprofv3 counter pass (its own process, its own time limit) and its per-kernel summary.
#   tools/pmc_pass.sh TAG "COUNTER COUNTER ..." -- python3 program.py args...
# Output: gpurun_out/pmc_TAG/ (raw) and gpurun_out/pmc_TAG.txt (summary, tools/pmc_summary.py).
# At most 8 SQ_, 2 GRBM_ counters per pass; FETCH_SIZE and WRITE_SIZE each in a pass of their own.
set -uo pipefail
tag=$1; ctrs=$2; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
# shellcheck disable=SC2086
timeout -s KILL ${PMC_TIMEOUT:-150} rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$out" -- "$@" > "$out/stdout.txt" 2> "$out/stderr.txt"
rc=$?
if [ $rc -ne 0 ]; then echo "pmc_pass $tag: rc=$rc"; tail -5 "$out/stderr.txt"; exit $rc; fi
{ echo "# rocprofv3 --pmc $ctrs --kernel-trace -- $*"; python3 "$root/tools/pmc_summary.py" "$out" ${PMC_KERNELS:-}; } > "$root/gpurun_out/pmc_$tag.txt"
echo "pmc_pass $tag: ok"
