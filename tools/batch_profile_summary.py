#!/usr/bin/env python3
"""profiles/r02_batch_{kernel_stats.csv,by_launch.txt,pmc.txt} from the passes of tools/profile_r02_batch.sh
(gpurun_out/r02_batch_stats, r02_batch_pmc1..3); the header lines of the two text files are kept."""
import collections
import csv
import glob
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)


def newest(d, suffix):
    return max(glob.glob(os.path.join("gpurun_out", d, "*", "*" + suffix)), key=os.path.getmtime)


def short(n):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", n))


shutil.copy(newest("r02_batch_stats", "_kernel_stats.csv"), "profiles/r02_batch_kernel_stats.csv")
rows = list(csv.DictReader(open(newest("r02_batch_stats", "_kernel_trace.csv"))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.OrderedDict()
for r in rows:
    n = short(r["Kernel_Name"])
    if n.startswith("vs::"):
        by.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
head = [l for l in open("profiles/r02_batch_by_launch.txt").read().split("\n") if l and not l.startswith("vs::")]
with open("profiles/r02_batch_by_launch.txt", "w") as f:
    f.write("\n".join(head) + "\n")
    for n, v in by.items():
        f.write(n + ": " + " ".join("%.0f" % x for x in v) + "\n")
out = open("profiles/r02_batch_pmc.txt").read().split("\n")[:2]
for p in ("r02_batch_pmc1", "r02_batch_pmc2", "r02_batch_pmc3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(newest(p, "_counter_collection.csv"))):
        n = short(r["Kernel_Name"])
        if n.startswith("vs::scan_batch"):
            acc[(n, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for (n, c), d in sorted(acc.items()):
        out.append("%s | %s | %d | %.0f" % (n, c, len(d), sum(d.values()) / len(d)))
open("profiles/r02_batch_pmc.txt", "w").write("\n".join(out) + "\n")
