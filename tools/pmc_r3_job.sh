#!/usr/bin/env bash
# Round-3 counter passes for the Barnes-Hut kernels.  Each pass is its own process with its own limit.
set -uo pipefail
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
P10K="python3 tools/bh_pmc.py 10000 1.0 --reps 5"
tools/pmc_pass.sh bh10k_sq "$SQ" -- $P10K &&
tools/pmc_pass.sh bh10k_sq2 "$SQ2" -- $P10K &&
tools/pmc_pass.sh bh10k_fetch "FETCH_SIZE" -- $P10K &&
tools/pmc_pass.sh bh10k_write "WRITE_SIZE" -- $P10K &&
P16M="python3 tools/bh_pmc.py 16777216 0.5 --plummer --world 8 --reps 2" &&
PMC_TIMEOUT=400 tools/pmc_pass.sh bh16m_sq "$SQ" -- $P16M &&
PMC_TIMEOUT=400 tools/pmc_pass.sh bh16m_sq2 "$SQ2" -- $P16M &&
PMC_TIMEOUT=400 tools/pmc_pass.sh bh16m_fetch "FETCH_SIZE" -- $P16M &&
PMC_TIMEOUT=400 tools/pmc_pass.sh bh16m_write "WRITE_SIZE" -- $P16M
