#!/usr/bin/env python3
"""Per-launch durations (microseconds) of the kernels whose name contains a filter, from the newest
rocprofv3 kernel trace under a directory: `python tools/trace_durations.py gpurun_out/x mq32`."""
import collections
import csv
import glob
import os
import sys

files = glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True)
newest = max(files, key=os.path.getmtime)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
by = collections.defaultdict(list)
for r in csv.DictReader(open(newest)):
    if flt in r["Kernel_Name"]:
        by[r["Kernel_Name"].split("(")[0][-60:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(newest)
for k, v in by.items():
    print(k, len(v), "launches:", " ".join(f"{x:.0f}" for x in v[:40]))
