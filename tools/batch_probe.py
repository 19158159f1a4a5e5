#!/usr/bin/env python3
"""Times the batched bf16 matrix-core scan (vs_search_batch_device): N docs x D, B queries.
Prints ms per batch, TFLOP/s (2*N*D*B), GB/s of row bytes and queries/s."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_mask as _device_mask  # noqa: E402


def device_mask(first, n, p, dev):
    return _device_mask(torch, first, n, p, dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_250_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--queries", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--pass-percents", default="100,10")
    ap.add_argument("--host-api", action="store_true",
                    help="time vs_search_batch (host queries, host mask, host results) with the wall clock instead")
    args = ap.parse_args()
    from search_engine_amd import abi
    dev = torch.device("cuda", 0)
    ctx = abi.Context(0)
    n = -(-args.docs // 64) * 64
    store = ctx.create_store(0, args.dim, n, dtype=abi.VS_BF16)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xC0FFEE)
    stream = torch.cuda.current_stream().cuda_stream
    for lo in range(0, args.docs, 1 << 18):
        m = min(1 << 18, args.docs - lo)
        rows = torch.randn(m, args.dim, generator=gen, device=dev).to(torch.bfloat16)
        store.upload_device(lo, rows.data_ptr(), m, stream)
        torch.cuda.synchronize()
    store.seal()
    q = torch.randn(args.queries, args.dim, device=dev)
    ws = torch.empty(store.batch_workspace_bytes(args.k), dtype=torch.uint8, device=dev)
    keys = torch.empty(args.queries * args.k, dtype=torch.int64, device=dev)
    for p in [int(x) for x in args.pass_percents.split(",")]:
        mask = device_mask(0, n, p, dev)
        if args.host_api:
            import time
            hq = q.cpu().numpy()
            hm = None if p >= 100 else mask.cpu().numpy().view(np.uint64)
            times = []
            for r in range(args.reps + 1):
                t0 = time.perf_counter()
                store.search_batch(hq, args.k, hm)
                if r:
                    times.append((time.perf_counter() - t0) * 1e3)
            t = float(np.median(times))
            print(f"p={p:3d}%  {t:8.3f} ms/batch through vs_search_batch (uploads and downloads included)  "
                  f"{args.queries / t * 1e3:10.0f} queries/s", flush=True)
            continue
        times = []
        for r in range(args.reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            store.search_batch_device(q.data_ptr(), args.queries, args.dim, args.k, mask.data_ptr(),
                                      ws.data_ptr(), keys.data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            if r:
                times.append(e0.elapsed_time(e1))
        t = float(np.median(times))
        flop = 2.0 * args.docs * args.dim * 256  # the launch always computes 256 query columns
        print(f"p={p:3d}%  {t:8.3f} ms/batch  {flop / t / 1e9:8.1f} TFLOP/s (padded to 256 queries)  "
              f"{args.docs * args.dim * 2 / t / 1e6:7.1f} GB/s rows  {args.queries / t * 1e3:10.0f} queries/s",
              flush=True)
    store.destroy()
    ctx.close()


if __name__ == "__main__":
    main()
