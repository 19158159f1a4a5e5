#!/usr/bin/env python3
"""Times the host-buffer entry points (query, candidate mask and results cross PCIe on
every call): vs_search with a dense mask, vs_search_roaring, vs_search without a mask.
Prints queries/s per variant; the numbers quoted as "PCIe-inclusive" come from here."""
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
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    from search_engine_amd import abi
    dev = torch.device("cuda", 0)
    ctx = abi.Context(0)
    n = -(-args.docs // 64) * 64
    store = ctx.create_store(0, args.dim, n)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xC0FFEE)
    for lo in range(0, args.docs, 1 << 18):
        m = min(1 << 18, args.docs - lo)
        rows = torch.randn(m, args.dim, generator=gen, device=dev)
        store.upload_device(lo, rows.data_ptr(), m, 0)
        torch.cuda.synchronize()
    store.seal()
    queries = np.random.default_rng(42).standard_normal((args.reps, args.dim), dtype=np.float32)
    for p in (100, 10, 1):
        mask = device_mask(torch, 0, n, p, dev).cpu().numpy().view(np.uint64)
        blob = abi.mask_to_roaring(mask)
        variants = {"dense mask": lambda q: store.search(q, args.k, mask),
                    "roaring": lambda q: store.search_roaring(q, args.k, blob)}
        if p == 100:
            variants["no mask"] = lambda q: store.search(q, args.k, None)
        ref = None
        for name, fn in variants.items():
            fn(queries[0])
            t0 = time.perf_counter()
            for q in queries:
                ids, _ = fn(q)
            dt = time.perf_counter() - t0
            if name != "no mask":
                ref = ref or ids.tolist()
                assert ids.tolist() == ref
            print(f"p={p:3d}% {name:10s} {args.reps / dt:9.1f} queries/s  {1e3 * dt / args.reps:7.3f} ms/query"
                  f"  (mask {mask.nbytes / 1e6:.2f} MB, roaring {len(blob) / 1e6:.2f} MB)", flush=True)
    store.destroy()
    ctx.close()


if __name__ == "__main__":
    main()
