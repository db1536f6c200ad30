#!/usr/bin/env bash
# Round 3, after the walk kernel's rewrite (decisions as lane masks, selects for the next node, early loads): counter
# passes for bh_walk_kernel at N = 10^6, theta = 1.
set -uo pipefail
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
P1M="python3 tools/bh_pmc.py 1000000 1.0 --reps 3"
export PMC_KERNELS="bh_walk_kernel"
tools/pmc_pass.sh walk1m_sq "$SQ" -- $P1M &&
tools/pmc_pass.sh walk1m_sq2 "$SQ2" -- $P1M &&
tools/pmc_pass.sh walk1m_fetch "FETCH_SIZE" -- $P1M &&
tools/pmc_pass.sh walk1m_write "WRITE_SIZE" -- $P1M
