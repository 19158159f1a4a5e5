#!/bin/bash
# vs_search_batch_device on 10 M x 768 bf16, 256 queries: per-launch timing and the HBM bytes of one call.
set -e
cd "$(dirname "$0")/.."
tools/profile_pass.sh r03_batch_stats stats -- python3 $PWD/tools/batch_probe.py --docs 10000000 --pass-percents 100,10,1 --reps 6
for p in 100 10 1; do
  tools/profile_pass.sh r03_batch_fetch_p$p "pmc:FETCH_SIZE" -- python3 $PWD/tools/batch_probe.py --docs 10000000 --pass-percents $p --reps 3
done
tools/profile_pass.sh r03_batch_pmc "pmc:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" -- python3 $PWD/tools/batch_probe.py --docs 10000000 --pass-percents 100 --reps 4
