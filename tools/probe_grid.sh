# Single-query float scan: the default grid against fixed grids, per row shape.
#   bash tools/probe_grid.sh [out-file]
set -e
cd "$(dirname "$0")/.."
P="timeout -k 10 200 python3 tools/scan_probe.py --pass-percents 100,10 --rounds 3"
out=${1:-gpurun_out/probe_grid.txt}; : > $out
for spec in "f32 128 20000000" "bf16 384 10000000" "f32 384 10000000" "bf16 768 10000000" "f32 512 5000000" "bf16 1024 5000000" "f32 768 10000000" "bf16 1536 5000000"; do
  set -- $spec
  echo "== $1 dim $2 docs $3" >> $out
  $P --variants 0:1,256:1,512:1,1024:1 --dtype $1 --dim $2 --docs $3 >> $out 2>&1
done
grep -v amdgpu.ids $out
