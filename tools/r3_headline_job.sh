#!/usr/bin/env bash
# Round-3 evidence for the headline: bench line, kernel stats of the same command, PMC passes.
set -uo pipefail
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 > gpurun_out/r3_naive_n1m_bench_steps20.json 2> gpurun_out/r3_naive_bench.err || exit 1
tail -c 600 gpurun_out/r3_naive_n1m_bench_steps20.json
export TMPDIR=/tmp
rm -rf gpurun_out/kt_naive; mkdir -p gpurun_out/kt_naive
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_naive -- python3 bench.py --no-cpu-baseline > gpurun_out/r3_naive_n1m_bench_under_rocprof.json 2> gpurun_out/kt_naive/stderr.txt || exit 1
cp "$(find gpurun_out/kt_naive -name '*kernel_stats.csv' | head -1)" gpurun_out/r3_naive_n1m_kernel_stats.csv
cat gpurun_out/r3_naive_n1m_kernel_stats.csv | head -5
PMC_KERNELS="force_naive integrate" tools/pmc_pass.sh naive_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline &&
PMC_KERNELS="force_naive integrate" tools/pmc_pass.sh naive_fetch "FETCH_SIZE" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline &&
PMC_KERNELS="force_naive integrate" tools/pmc_pass.sh naive_write "WRITE_SIZE" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
cat gpurun_out/pmc_naive_sq.txt gpurun_out/pmc_naive_fetch.txt gpurun_out/pmc_naive_write.txt
