#!/bin/bash
# Same-box A/B of builds of the library on the batched bf16 probe: tools/batch_ab.sh [variant.so ...]
# runs each given build (default: search_engine_amd/abl_A.so) and libvs_amd.so alternately, twice, and
# prints the steady-state launches of the streaming kernel from rocprofv3 traces.
cd "$(dirname "$0")/.."
variants=("$@"); [ ${#variants[@]} -eq 0 ] && variants=(search_engine_amd/abl_A.so)
for round in 1 2; do
  for v in "${variants[@]}" lib; do
    if [ $v = lib ]; then unset VS_AMD_LIB; else export VS_AMD_LIB=$PWD/$v; fi
    name=ab_$(basename $v .so)_$round
    tools/profile_pass.sh $name stats -- python3 $PWD/tools/batch_probe.py --docs 10000000 --pass-percents 100 --reps 14 > /dev/null || exit 1
    echo "== $v round $round"; python3 tools/trace_durations.py gpurun_out/$name scan_batch_stream | tail -1
  done
done
