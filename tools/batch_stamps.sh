#!/bin/bash
# Cycle stamps of the streaming batch kernel (separately built libraries with VS_STAMP and parts removed).
cd "$(dirname "$0")/.."
for v in STAMP STAMP_NOLOAD STAMP_NOLOAD_NOLDSREAD STAMP_NOMFMA; do
  echo "== $v"
  VS_AMD_LIB=$PWD/search_engine_amd/abl_$v.so timeout -k 10 120 python3 tools/batch_probe.py --docs 10000000 --pass-percents 100 --reps 10 2>&1 | grep -E "stream:|^p=" | tail -4 || exit 1
done
