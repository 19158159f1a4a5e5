# Scans in which queries share a pass: the default grid with and without a mask, and Hamming
#   bash tools/multi_grid_probe.sh
set -e
cd "$(dirname "$0")/.."
out=gpurun_out/s5_multi_probe2.txt; : > $out
run() { echo "== $*" >> $out; timeout -k 10 150 python3 tools/multi_probe.py "$@" >> $out 2>&1; }
run --dtype f32 --dim 768 --queries 16 --pass-percents 100 --unmasked
run --dtype f32 --dim 768 --queries 16 --pass-percents 100,10
run --dtype f32 --dim 384 --queries 16 --pass-percents 100 --unmasked
run --dtype bf16 --dim 768 --k 40 --queries 16 --pass-percents 100 --unmasked
run --dtype bin1 --dim 1536 --docs 25000000 --k 100 --queries 16 --pass-percents 100 --unmasked
run --dtype bin1 --dim 1536 --docs 25000000 --k 100 --queries 16 --pass-percents 100,10
echo "== bin1 VS_SCAN_BLOCKS=256" >> $out
VS_SCAN_BLOCKS=256 timeout -k 10 150 python3 tools/multi_probe.py --dtype bin1 --dim 1536 --docs 25000000 --k 100 --queries 16 --pass-percents 100 >> $out 2>&1
grep -v amdgpu.ids $out
