#!/bin/bash
# Counter passes over the streaming batch kernel: LDS queueing and the vector-memory issue path.
#   tools/batch_pmc_lds.sh [library.so]
set -e
cd "$(dirname "$0")/.."
[ -n "$1" ] && export VS_AMD_LIB=$PWD/$1
p1="SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
p2="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"
p3="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL GRBM_GUI_ACTIVE"
i=0
for p in "$p1" "$p2" "$p3"; do
  i=$((i+1))
  tools/profile_pass.sh pmc_lds_$i "pmc:$p" -- python3 $PWD/tools/batch_probe.py --docs 10000000 --pass-percents 100 --reps 4 > /dev/null
  python3 - gpurun_out/pmc_lds_$i <<'PY'
import csv, glob, sys, collections, os
path = max(glob.glob(os.path.join(sys.argv[1], "*", "*_counter_collection.csv")), key=os.path.getmtime)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if "scan_batch_stream_kernel<false" in r["Kernel_Name"] or "scan_batch_stream_kernel<0" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("  %-28s %2d launches, mean %.4g" % (k, len(v), sum(v) / len(v)))
PY
done
