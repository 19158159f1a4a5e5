#!/bin/bash
# vs_search_batch_device on 10 M x 768 bf16, 256 queries (bf16 matrix cores): kernel timing, then counters.
set -e
cd "$(dirname "$0")/.."
P="python3 $PWD/tools/batch_probe.py --docs 10000000 --pass-percents 100 --reps 4"
tools/profile_pass.sh r02_batch_stats stats -- python3 $PWD/tools/batch_probe.py --docs 10000000 --pass-percents 100,10 --reps 5
tools/profile_pass.sh r02_batch_pmc1 "pmc:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" -- $P
tools/profile_pass.sh r02_batch_pmc2 "pmc:SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" -- $P
tools/profile_pass.sh r02_batch_pmc3 "pmc:SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL GRBM_GUI_ACTIVE" -- $P
