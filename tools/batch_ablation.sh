#!/bin/bash
# Stamped diagnostic builds of the streaming batch kernel (search_engine_amd/abl_*.so, built with -DVS_STAMP and
# the -DVS_ABL_* switches their names list): cycles per 256-row tile and the clock they ran at.
cd "$(dirname "$0")/.."
for lib in search_engine_amd/abl_STAMP*.so; do
  echo "== $(basename $lib .so)"
  VS_AMD_LIB=$PWD/$lib timeout -k 10 120 python3 tools/batch_probe.py --docs 10000000 --pass-percents 100 --reps 10 2>&1 | grep -E "stream|^p=" | tail -19 || exit 1
done
