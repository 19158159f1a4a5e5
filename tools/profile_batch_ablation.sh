#!/bin/bash
# The streaming batch kernel with parts of its K loop removed (separately built libraries): how long
# the loop takes without its loads, without its matrix instructions, without its LDS reads.
set -e
cd "$(dirname "$0")/.."
for v in FULL NOLOAD NOLOAD_NOLDSREAD NOLOAD_NOLDSREAD_NOBARRIER; do
  if [ $v = FULL ]; then unset VS_AMD_LIB; else export VS_AMD_LIB=$PWD/search_engine_amd/abl_$v.so; fi
  tools/profile_pass.sh r02_abl_$v stats -- python3 $PWD/tools/batch_probe.py --docs 10000000 --pass-percents 100 --reps 10
  python3 tools/trace_durations.py gpurun_out/r02_abl_$v scan_batch_stream
done
