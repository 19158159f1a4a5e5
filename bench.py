#!/usr/bin/env python3
"""Headline benchmark: filtered top-10 vector search over 10M x 768 f32 rows.

    python bench.py --gpus N --steps K --warmup W

One process per GPU. With N > 1 and no WORLD_SIZE in the environment the command
starts its own ranks (a child `python -m torch.distributed.run`, before this
process has touched the GPU); launched by torch.distributed.run itself it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment. The documents are
sharded over the ranks by contiguous docid ranges (strong scaling: the total stays
the same). A "step" is one batch of `--queries-per-step` single-query searches
through ONE vs_search_sharded_device call: every rank scans its shard once per
query (candidate mask and queries resident in HBM), one launch merges the
per-workgroup lists of all queries, ONE RCCL all-gather (inside the library) moves
the batch's shard-local top-k keys (k * 8 bytes per query and rank) and one launch
merges them. Rank 0 prints one JSON line.

  --dtype f32|bf16         the headline workload (f32 is the reference's arithmetic)
  --dtype bin1             BASELINE config 4: 50M x 1536-bit rows, Hamming top-100
  --batch-queries B        not the headline: B queries share passes over the rows
                           (bf16: BASELINE config 3, the matrix-core scan)
  --dry-launch             CPU rehearsal of the launcher: ranks under gloo report
                           (rank, world, shard range); no GPU is touched

value        queries per second over the whole job (max-over-ranks time)
roofline     for the scan kernel: algorithmic bytes per launch / its average
             duration, measured with events on the stream it is launched on,
             against 8 TB/s HBM (rank 0's GPU)
cpu_baseline the CPU oracle (one query per core) on a bounded sample of the
             same rows, scaled to the full size (N = 1 only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, vendor peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16
METRIC = "filtered top-10 vector QPS, 10M×768, 1/2/4/8 GPUs; % of HBM roofline"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", choices=["f32", "bf16", "bin1"], default="f32")
    ap.add_argument("--docs", type=int, default=None, help="default 10M (f32, bf16), 50M (bin1)")
    ap.add_argument("--dim", type=int, default=None, help="default 768 (f32, bf16), 1536 bits (bin1)")
    ap.add_argument("--k", type=int, default=None, help="default 10 (f32, bf16), 100 (bin1)")
    ap.add_argument("--pass-percent", type=int, default=100,
                    help="facet filter pass rate of the headline number")
    ap.add_argument("--sweep", type=str, default="50,10,1",
                    help="further pass rates reported in `sweep` (N = 1 only)")
    ap.add_argument("--queries-per-step", type=int, default=16)
    ap.add_argument("--batch-queries", type=int, default=0,
                    help="not the headline: N > 0 times a batch of N <= 256 queries per step that share "
                         "passes over the rows (bf16: one pass on the matrix cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=500_000)
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher rehearsal on the CPU (gloo): every rank reports its shard, nothing is searched")
    args = ap.parse_args(argv)
    bin1 = args.dtype == "bin1"
    if args.docs is None:
        args.docs = 50_000_000 if bin1 else 10_000_000
    if args.dim is None:
        args.dim = 1536 if bin1 else 768
    if args.k is None:
        args.k = 100 if bin1 else 10
    return args


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`--gpus N` without a rank environment: start N ranks as a child job and pass its exit code on.
    Nothing in this process has initialised the GPU yet (no torch.cuda or HIP call), so the children
    are fresh processes and no program that holds the GPU is ever replaced."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["VS_BENCH_LAUNCHED_BY"] = "self"
    return subprocess.run(cmd, env=env).returncode


class stdout_to_stderr:
    """RCCL greets on stdout when its first communicator is made; this program's stdout carries one
    JSON line and nothing else, so file descriptor 1 points at stderr while communicators are made."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def shard_of(rank, world, docs):
    """Contiguous docid range of `rank`, boundaries on multiples of 64: (first docid, count)."""
    from search_engine_amd.sharded import shard_range
    return shard_range(rank, world, docs)


def dry_launch(args, rank, world):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = shard_of(rank, world, args.docs)
    seen = [None] * world
    dist.all_gather_object(seen, {"rank": rank, "world": world, "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                                  "first_docid": first, "count": count})
    dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": args.gpus, "world": world, "docs": args.docs,
                          "launched_by": os.environ.get("VS_BENCH_LAUNCHED_BY", "environment"),
                          "ranks": seen}), flush=True)
    dist.destroy_process_group()


def hash32_torch(x):
    """hash32 of tests/conftest.py on an int64 tensor."""
    m = 0xFFFFFFFF
    x = ((x ^ (x >> 16)) * 0x7FEB352D) & m
    x = ((x ^ (x >> 15)) * 0x846CA68B) & m
    return x ^ (x >> 16)


def device_mask(torch, first_docid, n_slots, pass_percent, dev):
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


def popcount_words(torch, words, n_local, dev):
    """Bits set among the first n_local slots of `words`."""
    n_slots = words.numel() * 64
    bits = words.clone()
    if n_local < n_slots:  # tail slots hold no item
        tail = torch.arange(n_slots - 64, n_slots, device=dev) >= n_local
        bits[-1] = bits[-1] & ~((tail.to(torch.int64) << torch.arange(64, device=dev)).sum())
    lut = torch.tensor([bin(v).count("1") for v in range(256)], dtype=torch.int64, device=dev)
    total = 0
    b = bits.view(torch.uint8)
    for lo in range(0, b.numel(), 1 << 26):
        total += int(lut[b[lo: lo + (1 << 26)].long()].sum().item())
    return total


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_launch:
        return dry_launch(args, rank, world)

    import numpy as np
    import torch

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # VS_BENCH_FORCE_DIST=1: a one-rank RCCL group and process group, so that the multi-rank code
    # (collectives included) can be rehearsed on a single GPU. Never set by the driver.
    force = os.environ.get("VS_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force
    dist = None
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        with stdout_to_stderr():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()  # (makes the communicator now, not inside the first timed collective)

    from search_engine_amd import abi

    ctx = abi.Context(local_rank)
    # the library's own RCCL communicator: rank 0 makes the id, the host's channel (here the
    # process group) hands it round
    uid = torch.zeros(abi.VS_GROUP_ID_BYTES, dtype=torch.uint8, device=dev)
    if rank == 0:
        with stdout_to_stderr():
            uid.copy_(torch.frombuffer(bytearray(abi.group_unique_id()), dtype=torch.uint8))
    if use_dist:
        dist.broadcast(uid, 0)
    # If the library's communicator cannot be made on some rank, every rank takes the same exchange
    # through torch.distributed instead (the same kernels; `config.collective` says which one ran).
    group, group_error = None, None
    try:
        if os.environ.get("VS_BENCH_NO_GROUP") == "1":  # rehearsal of the other exchange; never set by the driver
            raise RuntimeError("VS_BENCH_NO_GROUP=1")
        with stdout_to_stderr():
            group = abi.Group(ctx, rank, world, bytes(uid.cpu().numpy().tobytes()))
    except Exception as exc:  # reported below, never silent
        group_error = repr(exc)
    if use_dist:
        ok = torch.tensor([1 if group is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and group is not None:
            group.destroy()
            group = None
    if group is None and not use_dist:
        raise SystemExit(f"vs_group_create failed: {group_error}")
    if group is None and rank == 0:
        print(f"bench: vs_group_create failed ({group_error}); exchanging through torch.distributed", file=sys.stderr)

    dim, k = args.dim, args.k
    bin1 = args.dtype == "bin1"
    first, n_local = shard_of(rank, world, args.docs)
    dtype = {"f32": abi.VS_F32, "bf16": abi.VS_BF16, "bin1": abi.VS_BIN1}[args.dtype]
    metric = abi.VS_HAMMING if bin1 else abi.VS_ANGULAR
    store = ctx.create_store(0, dim, max(n_local, 64), dtype=dtype, metric=metric, docid_base=first)
    row_bytes = store.row_bytes
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xC0FFEE + rank)
    stream = torch.cuda.current_stream().cuda_stream
    chunk = 1 << 18
    sample_parts, sample_rows = [], 0  # rank 0 keeps the first --cpu-sample-rows rows for the CPU baseline
    want_sample = rank == 0 and world == 1 and not args.no_cpu_baseline
    for lo in range(0, n_local, chunk):
        m = min(chunk, n_local - lo)
        if bin1:  # uniform random bits = the sign bits of an N(0,1) draw, packed
            rows = torch.randint(-(1 << 31), 1 << 31, (m, row_bytes // 4), generator=gen, device=dev,
                                 dtype=torch.int64).to(torch.int32)
            if dim % 32:
                raise SystemExit("--dim must be a multiple of 32 for bin1 rows")
            if row_bytes * 8 > dim:  # zero the padding bits
                rows[:, dim // 32:] = 0
        else:
            tdtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
            rows = torch.randn(m, dim, generator=gen, device=dev, dtype=torch.float32).to(tdtype)
        store.upload_device(first + lo, rows.data_ptr(), m, stream)
        if want_sample and sample_rows < args.cpu_sample_rows:
            part = rows[: args.cpu_sample_rows - sample_rows]
            sample_parts.append(part)
            sample_rows += part.shape[0]
        torch.cuda.synchronize()
    store.seal()
    sample = torch.cat(sample_parts) if sample_parts else None
    del sample_parts

    qgen = torch.Generator(device=dev)
    qgen.manual_seed(42)
    if bin1:
        queries = torch.randint(-(1 << 31), 1 << 31, (1024, row_bytes // 4), generator=qgen, device=dev,
                                dtype=torch.int64).to(torch.int32)
        queries[:, dim // 32:] = 0
    else:
        queries = torch.randn(1024, dim, generator=qgen, device=dev, dtype=torch.float32)
    n_slots = -(-max(n_local, 64) // 64) * 64
    batched = args.batch_queries > 0
    n_q = args.batch_queries if batched else args.queries_per_step
    if not 0 < n_q <= 256:
        raise SystemExit("at most 256 queries per step")
    flags = (abi.VS_SHARDED_SHARE_PASSES if batched else 0) | (abi.VS_SHARDED_ALWAYS_GATHER if force else 0)
    out_ids = torch.empty(n_q * k, dtype=torch.int32, device=dev)
    out_dist = torch.empty(n_q * k, dtype=torch.float32, device=dev)
    out_n = torch.empty(n_q, dtype=torch.int32, device=dev)

    if group is None:  # the exchange outside the library: local keys, all_gather_into_tensor, one merge launch
        ws = torch.empty(max(store.workspace_bytes(k), store.batch_workspace_bytes(k) if batched else 0),
                         dtype=torch.uint8, device=dev)
        local_keys = torch.empty(n_q * k, dtype=torch.int64, device=dev)
        all_keys = torch.empty(world * n_q * k, dtype=torch.int64, device=dev)

    def step(i, mask, events=None):
        q = queries[(i * n_q) % (1024 - n_q + 1):]
        if events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if group is not None:
            group.search_sharded_device(store, q.data_ptr(), n_q, dim, k, mask.data_ptr(), flags,
                                        out_ids.data_ptr(), out_dist.data_ptr(), out_n.data_ptr(), stream)
        else:
            if batched:
                store.search_batch_device(q.data_ptr(), n_q, dim, k, mask.data_ptr(), ws.data_ptr(),
                                          local_keys.data_ptr(), stream)
            else:
                for j in range(n_q):
                    store.search_device(q[j].data_ptr(), dim, k, mask.data_ptr(), ws.data_ptr(),
                                        local_keys.data_ptr() + 8 * k * j, stream)
            dist.all_gather_into_tensor(all_keys, local_keys)
            ctx.merge_topk_batch_device(all_keys.data_ptr(), world, n_q * k, n_q, k, k, metric,
                                        out_ids.data_ptr(), out_dist.data_ptr(), out_n.data_ptr(), stream)
        if events is not None:
            e1.record()
            events.append((e0, e1))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    norm_bytes = 0 if bin1 else 4

    def run(pass_percent, steps, warmup):
        mask = device_mask(torch, first, n_slots, pass_percent, dev)
        n_pass = popcount_words(torch, mask, n_local, dev)
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
        step_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))  # this rank's kernels of one step
        return dict(pass_percent=pass_percent, qps=steps * n_q / dt, ms_per_step=1e3 * dt / steps,
                    step_kernel_ms=step_ms, n_pass=n_pass)

    head = run(args.pass_percent, args.steps, args.warmup)
    # sanity: the last batch's results are well-formed (ascending, k hits where the shard set has them)
    torch.cuda.synchronize()
    d = out_dist.view(n_q, k).cpu().numpy()
    cnt = out_n.cpu().numpy()
    assert (cnt <= k).all() and (cnt > 0).all(), "malformed result counts"
    for j in range(n_q):
        assert (np.diff(d[j, : cnt[j]]) >= 0).all(), "results not ascending"

    def single_query_roofline(r):
        """The scan reads every passing row once per query: rows + norms + the mask words. One scan
        launch takes all n_q queries of a step (no row read is shared between them), so a launch's
        algorithmic bytes are n_q times a query's, and its duration is the step's (the one merge launch
        and, with more than one rank, the gather included)."""
        algo = n_q * (r["n_pass"] * (row_bytes + norm_bytes) + n_slots // 8)
        ms = r["step_kernel_ms"]
        return algo, ms, algo / (ms * 1e-3) / 1e9

    sweep = []
    if world == 1 and args.sweep and not batched:
        for p in [int(x) for x in args.sweep.split(",") if x]:
            r = run(p, max(2, args.steps // 2), 1)
            algo, ms, gbps = single_query_roofline(r)
            sweep.append({"pass_percent": p, "qps": round(r["qps"], 2), "launch_ms": round(ms, 4),
                          "ms_per_query": round(ms / n_q, 4),
                          "achieved_gbps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4),
                          "algorithmic_bytes_per_launch": algo})

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
        cpu = cpu_baseline(torch, np, sample, queries, args, dim, k)
        if sweep and not bin1:  # the same baseline beside one filtered sweep point
            p = 10 if any(e["pass_percent"] == 10 for e in sweep) else sweep[0]["pass_percent"]
            cpu["filtered"] = cpu_baseline(torch, np, sample, queries, args, dim, k, pass_percent=p,
                                           mask_words=device_mask(torch, first, n_slots, p, dev), seconds=5.0)

    if rank == 0:
        stale = []  # recorded traffic that was measured on other kernel sources than the ones running now

        def recorded_traffic(pass_percent):
            """HBM bytes per scan launch from the committed rocprofv3 --pmc FETCH_SIZE pass of this
            workload (profiles/scan_traffic*.json, written by tools/profile_summary.py), or None. The
            counters need a run of their own under the profiler, so they are never collected inside a
            benchmark run; the file records a hash of the kernel sources it was measured on, and a
            number from other sources is reported with `traffic_stale`."""
            sfx = "" if pass_percent == 100 else f"_p{pass_percent}"
            name = "scan_traffic" if args.dtype == "f32" else f"scan_traffic_{args.dtype}"
            tpath = os.path.join(ROOT, "profiles", f"{name}{sfx}.json")
            if not os.path.exists(tpath):
                return None
            with open(tpath) as f:
                t = json.load(f)
            same = (t.get("docs") == args.docs and t.get("dtype") == args.dtype and t.get("n_gpus") == world
                    and t.get("pass_percent") == pass_percent and t.get("dim", dim) == dim
                    and t.get("queries_per_launch", 1) == n_q and t.get("k", k) == k)
            if not same:
                return None
            if t.get("csrc_sha256") != csrc_sha256():
                stale.append(pass_percent)
            return t.get("hbm_bytes_per_launch")

        def recorded_batch_traffic():
            """The same for one batched bf16 call (every kernel of it), profiles/scan_traffic_bf16_batch256.json."""
            sfx = "" if args.pass_percent == 100 else f"_p{args.pass_percent}"
            tpath = os.path.join(ROOT, "profiles", f"scan_traffic_bf16_batch{n_q}{sfx}.json")
            if not os.path.exists(tpath):
                return None
            with open(tpath) as f:
                t = json.load(f)
            same = (t.get("docs") == args.docs and t.get("dim") == dim and t.get("n_gpus") == world
                    and t.get("pass_percent") == args.pass_percent and t.get("batch_queries") == n_q)
            if not same:
                return None
            if t.get("csrc_sha256") != csrc_sha256():
                stale.append(args.pass_percent)
            return t.get("hbm_bytes_per_launch")

        what = {"f32": "cosine (arroy Angular)", "bf16": "cosine (arroy Angular)", "bin1": "Hamming"}[args.dtype]
        unit = "bit binary-quantised" if bin1 else f"dim {args.dtype}"
        line = {
            "metric": METRIC, "value": round(head["qps"], 3), "unit": "queries/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(head["ms_per_step"], 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64 popcount" if bin1 else args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{args.docs} docs x {dim}-{unit} {what}, "
                            + (f"{n_q} queries per step sharing passes, " if batched else "single-query ")
                            + f"filtered top-{k}, facet pass rate {args.pass_percent}%",
                "docs": args.docs, "dim": dim, "k": k, "pass_percent": args.pass_percent, "queries_per_step": n_q,
                "parallelism": f"docid-range shards x{world}, one RCCL all-gather of top-k keys per step "
                               f"(vs_search_sharded_device)",
                "ranks_in_group": group.world if group is not None else world,
                "collective": "vs_group (RCCL inside libvs_amd.so)" if group is not None
                              else "torch.distributed all_gather_into_tensor (vs_group_create failed)",
            },
        }
        if not batched:
            algo, ms, gbps = single_query_roofline(head)
            kernel = "scan_hamming_kernel" if bin1 else "scan_angular_kernel"
            line["roofline"] = {
                "bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": recorded_traffic(args.pass_percent),
                "traffic_stale": None,
                "kernel": (f"{kernel}: ONE launch for the step's {n_q} queries (+ the step's one merge launch)"
                           if group is not None else
                           f"{kernel}: {n_q} launches per step (vs_search_device per query; the library's group could not be made)"),
                "queries_per_launch": n_q, "algorithmic_bytes_per_launch": algo, "avg_launch_ms": round(ms, 4),
                "algorithmic_bytes_per_query": algo // n_q, "ms_per_query": round(ms / n_q, 4),
                "measured_copy_gbps": round(copy_gbps, 1), "frac_of_measured_copy": round(gbps / copy_gbps, 4),
            }
            for entry in sweep:
                entry["traffic"] = recorded_traffic(entry["pass_percent"])
                entry["traffic_stale"] = entry["pass_percent"] in stale
            line["roofline"]["traffic_stale"] = args.pass_percent in stale
        else:
            line["metric"] = f"batched top-{k} vector QPS, {n_q} queries per step sharing passes over the rows"
            ms = head["step_kernel_ms"]
            if args.dtype == "bf16" and k <= 32:
                # one pass over the shard on the matrix cores; a launch always computes 256 query columns.
                # Rows of more than 128 elements are fetched where the candidates lie (only they are read and
                # multiplied); shorter rows go through whole 256-row tiles.
                rows_read = head["n_pass"] if dim > 128 else n_local
                algo = rows_read * (row_bytes + 4) + n_slots // 8
                flop = 2.0 * rows_read * dim * 256
                line["roofline"] = {
                    "bound": "hbm", "achieved": round(algo / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    "traffic": recorded_batch_traffic(), "traffic_stale": args.pass_percent in stale,
                    "kernel": "scan_batch_bf16_kernel (both passes, merges and the query prologue included)",
                    "algorithmic_bytes_per_launch": algo, "avg_launch_ms": round(ms, 4),
                    "mfma_tflops": round(flop / (ms * 1e-3) / 1e12, 1),
                    "mfma_frac": round(flop / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                }
            else:
                # passes over the rows per step: 32 queries per pass on the f32 matrix cores (17 or more
                # f32 queries, dim <= 768, k <= 32); else 8 (f32, dim <= 768) or 4 queries share a pass
                if args.dtype == "f32" and dim <= 768 and k <= 32 and n_q >= 17:
                    passes = n_q // 32 + (1 if n_q % 32 >= 17 else -(-(n_q % 32) // 8))
                else:
                    per = 8 if args.dtype == "f32" and dim <= 768 else 4
                    passes = -(-n_q // per)
                algo = head["n_pass"] * (row_bytes + norm_bytes) + n_slots // 8  # per pass
                gbps = passes * algo / (ms * 1e-3) / 1e9
                line["roofline"] = {
                    "bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": None, "passes_per_step": passes,
                    "algorithmic_bytes_per_launch": algo, "avg_launch_ms": round(ms / passes, 4),
                }
        line["cpu_baseline"] = cpu
        line["sweep"] = sweep
        print(json.dumps(line), flush=True)
    if group is not None:
        group.destroy()
    store.destroy()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


def csrc_sha256():
    """Hash of what libvs_amd.so is built from -- search_engine_amd/csrc/vs_* and the Makefile's
    compiler, architecture, flags and source lists (the host mirror and the programs beside them are
    compiled into their own users, not into the library): names a build for the recorded-traffic
    check. (The tree travels to the GPU box without its .git, so a git tree hash cannot be asked for
    there; this is the same information from the files themselves.)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "search_engine_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.startswith("vs_") and name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    with open(os.path.join(d, "Makefile")) as f:
        for line in f:
            if line.split(" ", 1)[0] in ("HIPCC", "ARCH", "CXXFLAGS", "SRCS_HIP", "SRCS_CPP"):
                h.update(line.encode())
    return h.hexdigest()


def cpu_baseline(torch, np, sample, queries, args, dim, k, pass_percent=100, mask_words=None, seconds=10.0):
    """Times the CPU oracle (kind "port": the build's restatement of the reference's exact path;
    the reference itself cannot be built here) on the first rows of the same synthetic set, one
    query per thread on min(16, cores) threads, and scales to the full row count. With mask_words
    (the device mask of a sweep point) the same under that filter."""
    from oracle import scan as oracle
    n = min(args.cpu_sample_rows, sample.shape[0])
    mask = None
    if mask_words is not None:
        mask = mask_words[: -(-n // 64)].cpu().numpy().view(np.uint64).copy()
    host_cores = os.cpu_count() or 1
    threads = min(16, host_cores)
    if args.dtype == "bin1":
        from concurrent.futures import ThreadPoolExecutor
        codes = sample[:n].cpu().numpy().view(np.uint64)
        q = queries[: threads * 2].cpu().numpy().view(np.uint64)
        oracle.hamming_topk(codes, q[0], k, mask=mask)  # warm
        t0 = time.perf_counter()
        reps = 0
        with ThreadPoolExecutor(threads) as pool:  # the C call releases the GIL
            while True:
                list(pool.map(lambda qq: oracle.hamming_topk(codes, qq, k, mask=mask), q))
                reps += 1
                if time.perf_counter() - t0 > seconds or reps >= 200:
                    break
        dt = time.perf_counter() - t0
        used = threads
        t1 = time.perf_counter()
        oracle.hamming_topk(codes, q[0], k, mask=mask)
        lat = time.perf_counter() - t1
    else:
        rows = sample[:n].to(torch.float32).cpu().numpy()
        q = queries[: threads * 2].cpu().numpy()
        norms = oracle.norms(rows)  # arroy keeps the norm in the leaf header: not part of a query
        oracle.nns_batch_mt(rows, q[:threads], k, mask=mask, n_threads=threads, norms=norms)  # warm
        t0 = time.perf_counter()
        reps = 0
        while True:
            _, _, _, used = oracle.nns_batch_mt(rows, q, k, mask=mask, n_threads=threads, norms=norms)
            reps += 1
            if time.perf_counter() - t0 > seconds or reps >= 4000:
                break
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        oracle.nns_by_vector(rows, q[0], k, mask=mask, norms=norms)
        lat = time.perf_counter() - t1
    qps_sample = reps * q.shape[0] / dt
    scale = n / args.docs
    return {
        "value": round(qps_sample * scale, 4),
        "unit": "queries/s",
        "cores": int(used),
        "host_cores": host_cores,
        "kind": "port",
        "pass_percent": pass_percent,
        "sample": f"first {n} of {args.docs} rows, {'unfiltered' if mask is None else f'facet pass rate {pass_percent}%'}, {reps * q.shape[0]} queries in "
                  f"{dt:.1f} s, one query per thread on {int(used)} of {host_cores} host cores; "
                  f"scaled by {scale:.4f} to {args.docs} rows",
        "single_thread_ms_per_query_scaled": round(1e3 * lat / scale, 1),
    }


if __name__ == "__main__":
    main()
