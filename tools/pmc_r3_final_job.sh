#!/usr/bin/env bash
# End-of-round-3 counter passes for the Barnes-Hut build kernels written or rewritten in round 3 (own sort and scan, node
# kernel with staged stores and batched loads, wide climb), N = 10^6 and config 5's 2^24 Plummer as one rank of 8.
set -uo pipefail
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
P1M="python3 tools/bh_pmc.py 1000000 1.0 --reps 3"
tools/pmc_pass.sh bh1m_sq "$SQ" -- $P1M &&
tools/pmc_pass.sh bh1m_sq2 "$SQ2" -- $P1M &&
tools/pmc_pass.sh bh1m_fetch "FETCH_SIZE" -- $P1M &&
tools/pmc_pass.sh bh1m_write "WRITE_SIZE" -- $P1M &&
P16M="python3 tools/bh_pmc.py 16777216 0.5 --plummer --world 8 --reps 2" &&
PMC_TIMEOUT=400 tools/pmc_pass.sh bh16mf_sq "$SQ" -- $P16M &&
PMC_TIMEOUT=400 tools/pmc_pass.sh bh16mf_sq2 "$SQ2" -- $P16M &&
PMC_TIMEOUT=400 tools/pmc_pass.sh bh16mf_fetch "FETCH_SIZE" -- $P16M &&
PMC_TIMEOUT=400 tools/pmc_pass.sh bh16mf_write "WRITE_SIZE" -- $P16M
