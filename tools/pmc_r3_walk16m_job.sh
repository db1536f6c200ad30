#!/usr/bin/env bash
# Round 3, after the walk kernel's rewrite: bh_walk_kernel's instruction counters on config 5's 2^24 Plummer input, one
# rank of 8, theta = 0.5 (the first pass of tools/pmc_r3_walk_job.sh on the other BASELINE size).
set -uo pipefail
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
export PMC_KERNELS="bh_walk_kernel"
PMC_TIMEOUT=400 tools/pmc_pass.sh walk16m_sq "$SQ" -- python3 tools/bh_pmc.py 16777216 0.5 --plummer --world 8 --reps 2
