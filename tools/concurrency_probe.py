#!/usr/bin/env python3
"""Throughput of synchronous host-buffer searches from several request threads (each call has a
workspace and a stream of its own): vs_search_mask on 10 M x 768 f32 under a kept filter, and the
same requests gathered into vs_search_batch_masks calls of 16, as a server's batching layer would.
    python3 tools/concurrency_probe.py [--docs 10000000] [--pass-percents 100,10,1]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from search_engine_amd import abi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--pass-percents", default="100,10,1")
ap.add_argument("--seconds", type=float, default=3.0)
args = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = abi.Context(0)
store = ctx.create_store(0, args.dim, args.docs)
gen = torch.Generator(device=dev)
gen.manual_seed(1)
for lo in range(0, args.docs, 1 << 18):
    m = min(1 << 18, args.docs - lo)
    rows = torch.randn(m, args.dim, generator=gen, device=dev)
    store.upload_device(lo, rows.data_ptr(), m, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
store.seal()
queries = np.random.default_rng(2).standard_normal((256, args.dim)).astype(np.float32)
n_slots = -(-args.docs // 64) * 64
for p in [int(x) for x in args.pass_percents.split(",")]:
    words = bench.device_mask(torch, 0, n_slots, p, dev).cpu().numpy().view(np.uint64)
    handle = store.create_mask(words)
    line = {"docs": args.docs, "dim": args.dim, "pass_percent": p}
    for n_threads in (1, 2, 4, 8, 16):
        counts = [0] * n_threads
        stop = time.perf_counter() + args.seconds

        def worker(t):
            i = t
            while time.perf_counter() < stop:
                store.search_mask(queries[i % 256], 10, handle)
                i += n_threads
                counts[t] += 1

        th = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        line[f"qps_{n_threads}_threads"] = round(sum(counts) / (time.perf_counter() - t0), 1)
    # the same requests, sixteen per call (each with its own filter argument: here the same kept mask)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.seconds:
        store.search_batch_masks(queries[(n * 16) % 240: (n * 16) % 240 + 16], 10, [handle] * 16)
        n += 1
    line["qps_batches_of_16_one_thread"] = round(16 * n / (time.perf_counter() - t0), 1)
    print(line, flush=True)
    store.destroy_mask(handle)
