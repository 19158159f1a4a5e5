#!/bin/bash
# the single-query scans with one launch per step of 16 queries. Kernel timing and FETCH_SIZE
# passes (never in one run) of bench.py for the f32 headline (100 / 10 / 1 %), BASELINE config 2 (1 M rows),
# the bf16 headline and the Hamming scan.
set -e
cd "$(dirname "$0")/.."
B="python3 $PWD/bench.py --no-cpu-baseline --sweep= --steps 4 --warmup 1"
tools/profile_pass.sh r03_f32_stats stats -- $B
for p in 100 10 1; do
  tools/profile_pass.sh r03_f32_pmc_p$p pmc:FETCH_SIZE -- $B --steps 2 --pass-percent $p
done
for p in 100 10 1; do
  tools/profile_pass.sh r03_f32_1m_stats_p$p stats -- $B --docs 1000000 --steps 10 --pass-percent $p
done
tools/profile_pass.sh r03_bf16_stats stats -- $B --dtype bf16
for p in 100 10; do
  tools/profile_pass.sh r03_bf16_pmc_p$p pmc:FETCH_SIZE -- $B --dtype bf16 --steps 2 --pass-percent $p
done
for p in 100 10 1; do
  tools/profile_pass.sh r03_ham_stats_p$p stats -- $B --dtype bin1 --pass-percent $p
  tools/profile_pass.sh r03_ham_pmc_p$p pmc:FETCH_SIZE -- $B --dtype bin1 --pass-percent $p --steps 2
done
