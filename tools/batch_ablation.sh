#!/bin/bash
# Stamped diagnostic builds of the streaming batch kernel (search_engine_amd/abl_*.so): cycles per tile and clock.
cd "$(dirname "$0")/.."
for v in STAMP STAMP_NOLOAD_NOHITS STAMP_NOMFMA_NOLDSREAD; do
  echo "== $v"
  VS_AMD_LIB=$PWD/search_engine_amd/abl_$v.so timeout -k 10 120 python3 tools/batch_probe.py --docs 10000000 --pass-percents 100 --reps 10 2>&1 | grep -E "stream:|^p=" | tail -3 || exit 1
done
