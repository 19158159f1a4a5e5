#!/usr/bin/env python3
"""HBM bytes of one batched bf16 call (all its kernels) from a rocprofv3 --pmc FETCH_SIZE pass over
tools/batch_probe.py with ONE pass rate:
    tools/profile_pass.sh r02_batch_fetch "pmc:FETCH_SIZE" -- python3 tools/batch_probe.py --docs 10000000 --pass-percents 10 --reps 3
    python3 tools/batch_traffic.py gpurun_out/r02_batch_fetch 10000000 768 256 10
Writes profiles/scan_traffic_bf16_batch<queries>[_p<percent>].json. FETCH_SIZE is in KiB, and on gfx950 it
counts a 128-byte request as 64 bytes: x 1024 x 2."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (csrc_sha256)
d, docs, dim, n_q, percent = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
path = max(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
per_dispatch = collections.defaultdict(float)
name_of = {}
for r in csv.DictReader(open(path)):
    name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))
    if name.startswith("vs::") and "norms_kernel" not in name and r["Counter_Name"] == "FETCH_SIZE":
        per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
        name_of[r["Dispatch_Id"]] = name
calls = sum(1 for n in name_of.values() if n == "vs::prep_queries_bf16_kernel")
per_kernel = collections.defaultdict(float)
for i, v in per_dispatch.items():
    per_kernel[name_of[i]] += v * 1024 * 2 / calls
out = {
    "docs": docs, "dim": dim, "dtype": "bf16", "batch_queries": n_q, "pass_percent": percent, "n_gpus": 1,
    "hbm_bytes_per_launch": int(sum(per_kernel.values())),
    "per_kernel_bytes": {k: int(v) for k, v in per_kernel.items()},
    "correction": "x1024 (KiB) x2 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
    "csrc_sha256": bench.csrc_sha256(),  # the kernel sources the counters were collected on
    "source": "rocprofv3 --pmc FETCH_SIZE -- python3 tools/batch_probe.py --docs %d --pass-percents %d --reps 3; "
              "all kernels of one vs_search_batch_device call, mean of %d calls" % (docs, percent, calls),
}
sfx = "" if percent == 100 else "_p%d" % percent
with open(os.path.join(ROOT, "profiles", "scan_traffic_bf16_batch%d%s.json" % (n_q, sfx)), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))
