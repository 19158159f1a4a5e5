#!/bin/bash
# Headline f32 scan and the binary-quantised Hamming scan: kernel timing and FETCH_SIZE passes.
set -e
cd "$(dirname "$0")/.."
B="python3 $PWD/bench.py --no-cpu-baseline --sweep= --steps 4 --warmup 1"
tools/profile_pass.sh r02_f32_stats stats -- $B
tools/profile_pass.sh r02_f32_pmc pmc:FETCH_SIZE -- $B --steps 2
for p in 100 10 1; do
  tools/profile_pass.sh r02_ham_stats_p$p stats -- $B --dtype bin1 --pass-percent $p
  tools/profile_pass.sh r02_ham_pmc_p$p pmc:FETCH_SIZE -- $B --dtype bin1 --pass-percent $p --steps 2
done
