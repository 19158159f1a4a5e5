#!/bin/bash
# Quick same-box A/B of library builds on the batched bf16 probe (wall time per call, no profiler):
#   tools/batch_ab_quick.sh variant.so [variant2.so ...]   -- each build and libvs_amd.so alternately, three rounds;
# an argument of the form NAME=VALUE is exported instead (a measurement switch of the library), and "lib" names the plain build.
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for v in "$@" lib; do
    (
      case $v in
        lib) ;;
        *=*) export "$v" ;;
        *) export VS_AMD_LIB=$PWD/$v ;;
      esac
      echo -n "$v: "; timeout -k 10 120 python3 tools/batch_probe.py --docs 10000000 --pass-percents 100 --reps 20 2>&1 | grep "^p=" | cut -c1-30
    ) || exit 1
  done
done
