#!/usr/bin/env python3
"""Condenses rocprofv3 output into the files kept under profiles/.

    tools/profile_summary.py --stats gpurun_out/prof_stats --pmc gpurun_out/prof_pmc \
        --tag r01 --docs 10000000 --dtype f32 --pass-percent 100 --n-gpus 1

Writes profiles/<tag>_kernel_stats.csv (the `--kernel-trace --stats` summary, all
kernels), profiles/<tag>_scan_by_grid.txt (per-launch durations of the library's
own kernels, split by grid size) and, when a `--pmc FETCH_SIZE` pass is given,
profiles/scan_traffic[_<dtype>][_p<percent>].json: HBM bytes per scan launch = FETCH_SIZE (KiB) x 1024
x 2 — on gfx950 FETCH_SIZE reports half the bytes of a 16-byte-per-lane
coalesced stream (TCC_EA0_RDREQ counted at 64 B for 128-B requests).
"""
import argparse
import collections
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (csrc_sha256)


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return max(hits, key=os.path.getmtime)  # (gpurun merges a run's files into what earlier runs left behind)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--pmc")
    ap.add_argument("--tag", required=True)
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--pass-percent", type=int, default=100)
    ap.add_argument("--n-gpus", type=int, default=1)
    ap.add_argument("--kernel", default="scan_angular_kernel")
    ap.add_argument("--queries-per-launch", type=int, default=16,
                    help="queries one scan launch takes (bench.py --queries-per-step)")
    ap.add_argument("--k", type=int, default=None)
    args = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    if args.stats:
        with open(find(args.stats, "_kernel_stats.csv")) as f:
            rows = list(csv.reader(f))
        with open(os.path.join(out, f"{args.tag}_kernel_stats.csv"), "w", newline="") as f:
            csv.writer(f).writerows(rows)
        by = collections.defaultdict(list)
        with open(find(args.stats, "_kernel_trace.csv")) as f:
            for r in csv.DictReader(f):
                if "vs::" in r["Kernel_Name"]:
                    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                    wgs = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
                    lanes = int(r.get("Grid_Size_Y") or 1) // int(r.get("Workgroup_Size_Y") or 1)
                    by[(name, f"{wgs} x {lanes}" if lanes > 1 else str(wgs))].append(
                        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        with open(os.path.join(out, f"{args.tag}_scan_by_grid.txt"), "w") as f:
            f.write("kernel | workgroups (x query lanes) | launches | min us | median us | mean us | max us\n")
            for (name, wgs), v in sorted(by.items()):
                f.write(f"{name} | {wgs} | {len(v)} | {min(v):.1f} | {statistics.median(v):.1f} | "
                        f"{statistics.mean(v):.1f} | {max(v):.1f}\n")
        print(open(os.path.join(out, f"{args.tag}_scan_by_grid.txt")).read())
    if args.pmc:
        vals = []
        with open(find(args.pmc, "_counter_collection.csv")) as f:
            for r in csv.DictReader(f):
                if args.kernel in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                    vals.append(float(r["Counter_Value"]))
        if not vals:
            raise SystemExit("no FETCH_SIZE rows for the scan kernel")
        mean_kib = statistics.mean(vals)
        rec = {
            "docs": args.docs, "dim": args.dim, "dtype": args.dtype, "pass_percent": args.pass_percent,
            "n_gpus": args.n_gpus, "kernel": args.kernel, "launches": len(vals),
            "fetch_size_kib_mean": mean_kib,
            "correction": "x1024 (KiB) x2 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
            "hbm_bytes_per_launch": int(mean_kib * 1024 * 2),
            "queries_per_launch": args.queries_per_launch,
            "k": args.k if args.k is not None else (100 if args.dtype == "bin1" else 10),
            # the kernel sources the counters were collected on (bench.py reports other sources as stale)
            "csrc_sha256": bench.csrc_sha256(),
            "source": f"rocprofv3 --pmc FETCH_SIZE, profiles/{args.tag}_pmc_fetch_size.csv",
        }
        # the headline (p = 100 %) keeps the plain names; other pass rates get a suffix
        sfx = "" if args.pass_percent == 100 else f"_p{args.pass_percent}"
        fsfx = "" if args.tag.endswith(sfx) else sfx
        rec["source"] = f"rocprofv3 --pmc FETCH_SIZE, profiles/{args.tag}_pmc_fetch_size{fsfx}.csv"
        name = "scan_traffic" if args.dtype == "f32" else f"scan_traffic_{args.dtype}"
        with open(os.path.join(out, f"{name}{sfx}.json"), "w") as f:
            json.dump(rec, f, indent=1)
        with open(os.path.join(out, f"{args.tag}_pmc_fetch_size{fsfx}.csv"), "w") as f:
            f.write("kernel,launch,FETCH_SIZE_KiB\n")
            for i, v in enumerate(vals):
                f.write(f"{args.kernel},{i},{v}\n")
        print(json.dumps(rec))


if __name__ == "__main__":
    main()
