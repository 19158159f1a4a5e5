#!/bin/bash
# Same-box A/B of two builds of the library on the batched bf16 probe: search_engine_amd/abl_A.so against
# libvs_amd.so, alternating, steady-state launches of the streaming kernel from rocprofv3 traces.
cd "$(dirname "$0")/.."
for round in 1 2; do
  for v in A B; do
    if [ $v = A ]; then export VS_AMD_LIB=$PWD/search_engine_amd/abl_A.so; else unset VS_AMD_LIB; fi
    tools/profile_pass.sh ab_$v$round stats -- python3 $PWD/tools/batch_probe.py --docs 10000000 --pass-percents 100 --reps 14 > /dev/null || exit 1
    echo "== $v round $round"; python3 tools/trace_durations.py gpurun_out/ab_$v$round scan_batch_stream | tail -1
  done
done
