#!/bin/bash
# Cycles per 256-row tile of the streaming batch kernel and the clock they ran at, from a
# separately built library with in-kernel stamps:
#   cd search_engine_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DVS_STAMP \
#        -c vs_batch.hip -o build/vs_batch_stamp.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../stamp.so \
#        build/vs_kernels.o build/vs_batch_stamp.o build/vs_runtime.o build/vs_group.o build/vs_roaring.o build/vs_ingest.o -ldl
cd "$(dirname "$0")/.."
VS_AMD_LIB=$PWD/search_engine_amd/stamp.so timeout -k 10 120 python3 tools/batch_probe.py --docs 10000000 --pass-percents 100 --reps 10 2>&1 | grep -E "stream:|^p=" | tail -4
