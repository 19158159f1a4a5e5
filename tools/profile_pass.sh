#!/bin/bash
# One rocprofv3 pass of one command, output under gpurun_out/<name>:
#   tools/profile_pass.sh <name> stats|pmc:<COUNTERS...> -- python3 <program> <args>
# Kernel timing and counters never share a run. TMPDIR must be /tmp for rocprofv3 here.
set -e
name=$1; mode=$2; shift 3
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$name
rm -rf "$out"; mkdir -p "$out"
cd /tmp; export TMPDIR=/tmp
export PYTHONPATH=$root
case $mode in
  stats) timeout -s KILL 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- "$@" > "$out.log" 2>&1 ;;
  pmc:*) timeout -s KILL 420 rocprofv3 --pmc ${mode#pmc:} --output-format csv -d "$out" -- "$@" > "$out.log" 2>&1 ;;
  *) echo "mode?"; exit 2 ;;
esac
tail -2 "$out.log"
