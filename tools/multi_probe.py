#!/usr/bin/env python3
"""Times vs_search_batch where 32, eight or four queries share each pass over the rows (f32 stores, bf16
beyond the matrix-core path's k, Euclidean / dot stores) against vs_search, one query at a time, on the
same store: queries per second and algorithmic GB/s per pass."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_mask  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--metric", default="angular")
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--pass-percents", default="100,10,1")
    ap.add_argument("--unmasked", action="store_true", help="pass no mask at all (pass rate 100 only)")
    ap.add_argument("--no-check", action="store_true", help="skip the comparison with vs_search (ablated builds)")
    args = ap.parse_args()
    from search_engine_amd import abi
    dev = torch.device("cuda", 0)
    ctx = abi.Context(0)
    n = -(-args.docs // 64) * 64
    binary = args.dtype == "bin1"  # Hamming store: four queries share each pass
    dtype = {"f32": abi.VS_F32, "bf16": abi.VS_BF16, "bin1": abi.VS_BIN1}[args.dtype]
    metric = abi.VS_HAMMING if binary else {"angular": abi.VS_ANGULAR, "euclidean": abi.VS_EUCLIDEAN,
                                            "dot": abi.VS_DOT}[args.metric]
    store = ctx.create_store(0, args.dim, n, dtype=dtype, metric=metric)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xC0FFEE)
    stream = torch.cuda.current_stream().cuda_stream
    tdtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    for lo in range(0, args.docs, 1 << 18):
        m = min(1 << 18, args.docs - lo)
        if binary:
            rows = torch.randint(-2**63, 2**63 - 1, (m, args.dim // 64), generator=gen, device=dev, dtype=torch.int64)
        else:
            rows = torch.randn(m, args.dim, generator=gen, device=dev).to(tdtype)
        store.upload_device(lo, rows.data_ptr(), m, stream)
        torch.cuda.synchronize()
    store.seal()
    queries = torch.randn(args.queries, args.dim, device=dev).cpu().numpy()
    lut = torch.tensor([bin(v).count("1") for v in range(256)], dtype=torch.int64, device=dev)
    for p in [int(x) for x in args.pass_percents.split(",")]:
        dmask = device_mask(torch, 0, n, p, dev)
        n_pass = int(lut[dmask.view(torch.uint8).long()].sum().item())
        algo = n_pass * (store.row_bytes + (4 if metric == abi.VS_ANGULAR else 0)) + n // 8
        mask = None if args.unmasked and p == 100 else dmask.cpu().numpy().view(np.uint64)
        store.search_batch(queries, args.k, mask)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            batch = store.search_batch(queries, args.k, mask)
        t_batch = (time.perf_counter() - t0) / reps
        n_single = min(8, args.queries)
        t0 = time.perf_counter()
        singles = [store.search(q, args.k, mask) for q in queries[:n_single]]
        t_single = (time.perf_counter() - t0) / n_single
        mfma = (args.dtype == "f32" and args.metric != "euclidean" and args.k <= 32 and args.dim <= 768
                and args.queries >= 17)
        for (a, b), (c, d) in zip(batch, singles):
            if args.no_check:
                continue
            if mfma:  # another summation order: equal up to rounding
                assert len(set(a.tolist()) ^ set(c.tolist())) <= 2
            else:
                assert a.tolist() == c.tolist() and b.tolist() == d.tolist()
        group = 32 if mfma else 8 if args.dtype == "f32" and args.dim <= 768 else 4  # (4 also for binary rows)
        passes = -(-args.queries // group)
        print(f"p={p:3d}%  batch of {args.queries}: {1e3 * t_batch:8.3f} ms = {args.queries / t_batch:9.1f} queries/s, "
              f"{1e3 * t_batch / passes:7.3f} ms per pass = {algo / (t_batch / passes) / 1e9:7.1f} GB/s;  "
              f"vs_search: {1e3 * t_single:7.3f} ms = {1 / t_single:8.1f} queries/s", flush=True)
    store.destroy()
    ctx.close()


if __name__ == "__main__":
    main()
