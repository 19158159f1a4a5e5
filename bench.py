#!/usr/bin/env python3
"""Headline benchmark: filtered top-10 vector search over 10M x 768 f32 rows.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (N > 1: launched by torch.distributed.run, RCCL). The 10M
documents are sharded over the ranks by contiguous docid ranges (strong
scaling: the total stays 10M). A "step" is one batch of `--queries-per-step`
single-query searches: every rank scans its shard for each query (one scan
launch per query, candidate mask and query resident in HBM), then ONE
all-gather moves the batch's shard-local top-k keys (k * 8 bytes per query and
rank) and one launch merges them. Rank 0 prints one JSON line.

value        queries per second over the whole job (max-over-ranks time)
roofline     for the scan kernel: algorithmic bytes per launch / its average
             duration, measured with events on the stream it is launched on,
             against 8 TB/s HBM
cpu_baseline the CPU oracle (one query per core) on a bounded sample of the
             same rows, scaled to 10M rows (N = 1 only)
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, vendor peak


def hash32_torch(x):
    """hash32 of tests/conftest.py on an int64 tensor."""
    m = 0xFFFFFFFF
    x = ((x ^ (x >> 16)) * 0x7FEB352D) & m
    x = ((x ^ (x >> 15)) * 0x846CA68B) & m
    return x ^ (x >> 16)


def device_mask(first_docid, n_slots, pass_percent, dev):
    """Candidate words for docids [first_docid, first_docid + n_slots): bit set
    iff hash32(docid) % 100 < pass_percent. int64 tensor of n_slots / 64 words."""
    words = torch.zeros(n_slots // 64, dtype=torch.int64, device=dev)
    chunk = 1 << 22
    for lo in range(0, n_slots, chunk):
        hi = min(n_slots, lo + chunk)
        ids = torch.arange(first_docid + lo, first_docid + hi, dtype=torch.int64, device=dev)
        ok = (hash32_torch(ids) % 100) < pass_percent
        bits = ok.view(-1, 64).to(torch.int64) << torch.arange(64, dtype=torch.int64, device=dev)
        words[lo // 64: hi // 64] = bits.sum(1)  # disjoint bits: the sum is the OR (wraps into the sign bit)
    return words


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--pass-percent", type=int, default=100,
                    help="facet filter pass rate of the headline number")
    ap.add_argument("--sweep", type=str, default="50,10,1",
                    help="further pass rates reported in `sweep` (N = 1 only)")
    ap.add_argument("--queries-per-step", type=int, default=16)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--batch-queries", type=int, default=0,
                    help="not the headline: N > 0 times a batch of N <= 256 queries per step instead "
                         "(bf16: one pass on the matrix cores; f32: eight queries share each pass)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=500_000)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # VS_BENCH_FORCE_DIST=1: a one-rank RCCL group, so that the multi-rank code (collectives
    # included) can be rehearsed on a single GPU. Never set by the driver.
    use_dist = world > 1 or os.environ.get("VS_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from search_engine_amd import abi

    ctx = abi.Context(local_rank)
    dim, k, qps_batch = args.dim, args.k, args.queries_per_step
    # shard = contiguous docid range, boundaries on multiples of 64
    per = -(-args.docs // world)
    per = -(-per // 64) * 64
    first = rank * per
    n_local = max(0, min(args.docs, first + per) - first)
    dtype = abi.VS_F32 if args.dtype == "f32" else abi.VS_BF16
    tdtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    store = ctx.create_store(0, dim, max(n_local, 64), dtype=dtype, docid_base=first)
    assert store.row_bytes == dim * (4 if args.dtype == "f32" else 2)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xC0FFEE + rank)
    stream = torch.cuda.current_stream().cuda_stream
    chunk = 1 << 18
    sample = None
    for lo in range(0, n_local, chunk):
        m = min(chunk, n_local - lo)
        rows = torch.randn(m, dim, generator=gen, device=dev, dtype=torch.float32).to(tdtype)
        store.upload_device(first + lo, rows.data_ptr(), m, stream)
        if rank == 0 and lo == 0:
            sample = rows
        torch.cuda.synchronize()
    store.seal()

    qgen = torch.Generator(device=dev)
    qgen.manual_seed(42)
    queries = torch.randn(1024, dim, generator=qgen, device=dev, dtype=torch.float32)
    n_slots = -(-max(n_local, 64) // 64) * 64
    row_bytes = store.row_bytes
    from search_engine_amd import sharded
    if args.batch_queries:
        batched_bench(args, ctx, store, dev, stream, queries, first, n_slots, n_local, rank, world, use_dist)
        store.destroy()
        ctx.close()
        if use_dist:
            dist.destroy_process_group()
        return
    cur = {"step": 0, "mask": None, "events": None}
    searcher = sharded.gpu_sharded_search(
        ctx, store, k, qps_batch, dev,
        query_ptr=lambda j: queries[(cur["step"] * qps_batch + j) % 1024].data_ptr(),
        mask_ptr=lambda: cur["mask"].data_ptr(), stream=lambda: stream, always_gather=use_dist)
    plain_local_search = searcher.local_search

    def timed_local_search(j, out_keys):
        # events on the stream the scan is launched on (torch's current stream)
        if cur["events"] is None:
            return plain_local_search(j, out_keys)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        plain_local_search(j, out_keys)
        e1.record()
        cur["events"].append((e0, e1))

    searcher.local_search = timed_local_search
    out = {}

    def step(i, mask, events=None):
        cur.update(step=i, mask=mask, events=events)
        out["ids"], out["dist"], out["n"] = searcher.run()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def run(pass_percent, steps, warmup):
        mask = device_mask(first, n_slots, pass_percent, dev)
        # popcount of the shard's candidate words (present == all for docids < n_local)
        bits = mask.clone()
        if n_local < n_slots:  # tail slots hold no item
            tail = torch.arange(n_slots - 64, n_slots, device=dev) >= n_local
            bits[-1] = bits[-1] & ~((tail.to(torch.int64) << torch.arange(64, device=dev)).sum())
        b = bits.view(torch.uint8)
        lut = torch.tensor([bin(v).count("1") for v in range(256)], dtype=torch.int64, device=dev)
        n_pass = int(lut[b.long()].sum().item())
        for i in range(warmup):
            step(i, mask)
        barrier()
        events = []
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i, mask, events)
        barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        scan_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
        algo_bytes = n_pass * (row_bytes + 4) + n_slots // 8
        return dict(pass_percent=pass_percent, qps=steps * qps_batch / dt, ms_per_step=1e3 * dt / steps,
                    scan_ms=scan_ms, n_pass=n_pass, algo_bytes=algo_bytes,
                    achieved_gbps=algo_bytes / (scan_ms * 1e-3) / 1e9)

    head = run(args.pass_percent, args.steps, args.warmup)
    sweep = []
    if world == 1 and args.sweep:
        for p in [int(x) for x in args.sweep.split(",") if x]:
            r = run(p, max(2, args.steps // 2), 1)
            sweep.append({"pass_percent": p, "qps": round(r["qps"], 2),
                          "scan_ms": round(r["scan_ms"], 4),
                          "achieved_gbps": round(r["achieved_gbps"], 1),
                          "frac": round(r["achieved_gbps"] / HBM_PEAK_GBPS, 4),
                          "algorithmic_bytes_per_launch": r["algo_bytes"]})

    # sanity: the last batch's results are well-formed (k hits, ascending)
    torch.cuda.synchronize()
    d = out["dist"].view(qps_batch, k).cpu().numpy()
    assert (out["n"].cpu().numpy() == k).all() and (np.diff(d, axis=1) >= 0).all(), "malformed results"

    # what a plain device-to-device copy reaches on this GPU (bytes read + bytes written), for scale
    copy_gbps = None
    if rank == 0:
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 5 * 2 * src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sample, queries, args, dim, k)

    if rank == 0:
        def recorded_traffic(pass_percent):
            """HBM bytes per scan launch from the committed rocprofv3 --pmc FETCH_SIZE pass of this
            workload (profiles/scan_traffic*.json, written by tools/profile_summary.py), or None."""
            sfx = "" if pass_percent == 100 else f"_p{pass_percent}"
            tpath = os.path.join(ROOT, "profiles", f"scan_traffic{sfx}.json")
            if not os.path.exists(tpath):
                return None
            with open(tpath) as f:
                t = json.load(f)
            same = (t.get("docs") == args.docs and t.get("dtype") == args.dtype and t.get("n_gpus") == world
                    and t.get("pass_percent") == pass_percent)
            return t.get("hbm_bytes_per_launch") if same else None

        traffic = recorded_traffic(args.pass_percent)
        for entry in sweep:
            entry["traffic"] = recorded_traffic(entry["pass_percent"])
        line = {
            "metric": "filtered top-10 vector QPS, 10M×768, 1/2/4/8 GPUs; % of HBM roofline",
            "value": round(head["qps"], 3),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(head["ms_per_step"], 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{args.docs} docs x {dim}-dim {args.dtype} cosine (arroy Angular), "
                            f"single-query filtered top-{k}, facet pass rate {args.pass_percent}%",
                "docs": args.docs, "dim": dim, "k": k, "pass_percent": args.pass_percent,
                "queries_per_step": qps_batch,
                "parallelism": f"docid-range shards x{world}, one all-gather of top-k keys per step",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(head["achieved_gbps"], 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(head["achieved_gbps"] / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "kernel": "scan_angular_kernel (+ its merge launch)",
                "algorithmic_bytes_per_launch": head["algo_bytes"],
                "avg_launch_ms": round(head["scan_ms"], 4),
                "measured_copy_gbps": round(copy_gbps, 1),
                "frac_of_measured_copy": round(head["achieved_gbps"] / copy_gbps, 4),
            },
            "cpu_baseline": cpu,
            "sweep": sweep,
        }
        print(json.dumps(line), flush=True)
    store.destroy()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


def batched_bench(args, ctx, store, dev, stream, queries, first, n_slots, n_local, rank, world, use_dist):
    """Not the headline. `--batch-queries` queries per step against a store sharded over the ranks;
    one vs_search_batch_device per rank, one all-gather of the keys, one merge launch. With
    --dtype bf16 this is BASELINE config 3 (the matrix-core scan); with f32 the scan that groups
    of queries share."""
    import torch.distributed as dist
    from search_engine_amd import sharded
    if not 0 < args.batch_queries <= 256:
        raise SystemExit("--batch-queries takes at most 256 queries")
    b, k, dim = args.batch_queries, args.k, args.dim
    mask = device_mask(first, n_slots, args.pass_percent, dev)
    cur = {"step": 0}
    searcher = sharded.gpu_sharded_batch_search(
        ctx, store, k, b, dev, queries_ptr=lambda: queries[(cur["step"] * b) % (1024 - b + 1):].data_ptr(),
        mask_ptr=lambda: mask.data_ptr(), stream=lambda: stream, always_gather=use_dist)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        cur["step"] = i
        searcher.run()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        cur["step"] = args.warmup + i
        searcher.run()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        ms = 1e3 * dt / args.steps
        common = {
            "value": round(args.steps * b / dt, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.docs} docs x {dim}-dim {args.dtype} cosine, {b} queries per step, top-{k}, "
                                   f"facet pass rate {args.pass_percent}%", "docs": args.docs, "dim": dim, "k": k,
                       "parallelism": f"docid-range shards x{world}, one all-gather of top-k keys per step"}}
        if args.dtype == "bf16" and k <= 32:
            flop = 2.0 * args.docs * dim * 256  # a launch always computes 256 query columns
            gbps = args.docs * dim * 2 / (ms * 1e-3) / 1e9
            print(json.dumps({
                "metric": "batched top-10 vector QPS (matrix-core scan), 256 queries per step", **common,
                "roofline": {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS * world,
                             "unit": "GB/s", "frac": round(gbps / (HBM_PEAK_GBPS * world), 4), "traffic": None,
                             "mfma_tflops": round(flop / (ms * 1e-3) / 1e12, 1),
                             "mfma_frac_of_2500_per_gpu": round(flop / (ms * 1e-3) / 1e12 / (2500.0 * world), 4)},
            }), flush=True)
        else:
            # 32 queries per pass on the f32 matrix cores (17 or more f32 queries, dim <= 768, k <= 32);
            # else groups of 8 (f32, dim <= 768) or 4 queries share each pass over the rows
            if args.dtype == "f32" and dim <= 768 and k <= 32 and b >= 17:
                group = 32
                passes = b // 32 + (1 if b % 32 >= 17 else -(-(b % 32) // 8))
            else:
                group = 8 if args.dtype == "f32" and dim <= 768 else 4
                passes = -(-b // group)
            row_bytes = dim * (4 if args.dtype == "f32" else 2) + 4
            lut = torch.tensor([bin(v).count("1") for v in range(256)], dtype=torch.int64, device=dev)
            n_pass = min(int(lut[mask.view(torch.uint8).long()].sum().item()), n_local)  # this rank's share
            gbps = passes * world * n_pass * row_bytes / (ms * 1e-3) / 1e9  # shards are equal up to rounding
            print(json.dumps({
                "metric": f"batched top-{k} vector QPS (scan shared by up to {group} queries per pass)", **common,
                "roofline": {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS * world,
                             "unit": "GB/s", "frac": round(gbps / (HBM_PEAK_GBPS * world), 4), "traffic": None,
                             "passes_per_step": passes},
            }), flush=True)


def cpu_baseline(sample, queries, args, dim, k):
    """Times the CPU oracle (kind "port": the build's restatement of the
    reference's exact path; the reference itself cannot be built here) on the
    first rows of the same synthetic set, one query per thread."""
    from oracle import scan as oracle
    n = min(args.cpu_sample_rows, sample.shape[0])
    rows = sample[:n].to(torch.float32).cpu().numpy()
    threads = min(16, os.cpu_count() or 1)
    q = queries[: threads * 2].cpu().numpy()
    norms = oracle.norms(rows)  # arroy keeps the norm in the leaf header: not part of a query
    oracle.nns_batch_mt(rows, q[:threads], k, n_threads=threads, norms=norms)  # warm
    t0 = time.perf_counter()
    reps = 0
    while True:
        _, _, _, used = oracle.nns_batch_mt(rows, q, k, n_threads=threads, norms=norms)
        reps += 1
        if time.perf_counter() - t0 > 10.0 or reps >= 50:
            break
    dt = time.perf_counter() - t0
    qps_sample = reps * q.shape[0] / dt
    t1 = time.perf_counter()
    oracle.nns_by_vector(rows, q[0], k, norms=norms)
    lat = time.perf_counter() - t1
    scale = n / args.docs
    return {
        "value": round(qps_sample * scale, 4),
        "unit": "queries/s",
        "cores": int(used),
        "kind": "port",
        "sample": f"first {n} of {args.docs} rows, unfiltered, {reps * q.shape[0]} queries in "
                  f"{dt:.1f} s, one query per thread; scaled by {scale:.4f} to {args.docs} rows",
        "single_thread_ms_per_query_scaled": round(1e3 * lat / scale, 1),
    }


if __name__ == "__main__":
    main()
