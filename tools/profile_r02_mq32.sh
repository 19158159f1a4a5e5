#!/bin/bash
# vs_search_batch on 10 M x 768 f32 (32 queries per pass on the f32 matrix cores): kernel timing, then counters.
set -e
cd "$(dirname "$0")/.."
P="python3 $PWD/tools/multi_probe.py --pass-percents 100 --queries 32"
tools/profile_pass.sh r02_mq32_stats stats -- python3 $PWD/tools/multi_probe.py --pass-percents 100,10,1 --queries 64
tools/profile_pass.sh r02_mq32_pmc1 "pmc:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" -- $P
tools/profile_pass.sh r02_mq32_pmc2 "pmc:SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" -- $P
tools/profile_pass.sh r02_mq32_pmc3 "pmc:SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" -- $P
