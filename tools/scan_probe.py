#!/usr/bin/env python3
"""A/B of scan schedules in ONE process (interleaved rounds): builds one store,
then times vs_search_device under each (map, blocks, nt) at several facet pass
rates and prints algorithmic GB/s (median and best of the rounds)."""
import argparse
import os
import sys

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
    ap.add_argument("--metric", default="", help="angular | euclidean | dot (f32/bf16 rows); default by dtype")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--pass-percents", default="100,10,1")
    ap.add_argument("--variants", default="0:1,0:0,512:1,1536:1,2048:1", help="blocks:nt,...")
    args = ap.parse_args()
    from search_engine_amd import abi
    dev = torch.device("cuda", 0)
    ctx = abi.Context(0)
    n = -(-args.docs // 64) * 64
    dtype = {"f32": abi.VS_F32, "bf16": abi.VS_BF16, "bin1": abi.VS_BIN1}[args.dtype]
    tdtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    binary = args.dtype == "bin1"
    store = ctx.create_store(0, args.dim, n, dtype=dtype,
                             metric=abi.VS_HAMMING if binary else
                             {"": abi.VS_ANGULAR, "angular": abi.VS_ANGULAR, "euclidean": abi.VS_EUCLIDEAN,
                              "dot": abi.VS_DOT}[args.metric])
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xC0FFEE)
    stream = torch.cuda.current_stream().cuda_stream
    for lo in range(0, args.docs, 1 << 18):
        m = min(1 << 18, args.docs - lo)
        if binary:
            assert args.dim % 128 == 0
            rows = torch.randint(-2**63, 2**63 - 1, (m, args.dim // 64), generator=gen, device=dev,
                                 dtype=torch.int64)
        else:
            rows = torch.randn(m, args.dim, generator=gen, device=dev).to(tdtype)
        store.upload_device(lo, rows.data_ptr(), m, stream)
        torch.cuda.synchronize()
    store.seal()
    if binary:
        queries = torch.randint(-2**63, 2**63 - 1, (64, args.dim // 64), device=dev, dtype=torch.int64)
    else:
        queries = torch.randn(64, args.dim, device=dev)
    keys = torch.empty(args.k, dtype=torch.int64, device=dev)
    variants = [tuple(int(x) for x in v.split(":")) for v in args.variants.split(",")]
    lut = torch.tensor([bin(v).count("1") for v in range(256)], dtype=torch.int64, device=dev)
    for p in [int(x) for x in args.pass_percents.split(",")]:
        mask = device_mask(torch, 0, n, p, dev)
        n_pass = int(lut[mask.view(torch.uint8).long()].sum().item())
        algo = n_pass * (store.row_bytes + (4 if args.metric in ('', 'angular') and not binary else 0)) + n // 8
        times = {v: [] for v in variants}
        ref = None
        for r in range(args.rounds + 1):
            for v in variants:
                ctx.set_scan_tuning(*v)
                ws = torch.empty(store.workspace_bytes(args.k), dtype=torch.uint8, device=dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(args.reps):
                    store.search_device(queries[i].data_ptr(), args.dim, args.k, mask.data_ptr(),
                                        ws.data_ptr(), keys.data_ptr(), stream)
                e1.record()
                torch.cuda.synchronize()
                got = keys.cpu().numpy().tolist()
                ref = ref or got
                assert got == ref, "schedules disagree"
                if r:  # the first repetition warms up
                    times[v].append(e0.elapsed_time(e1) / args.reps)
        for v in variants:
            t = np.array(times[v])
            print(f"p={p:3d}% blocks={v[0]:4d} nt={v[1]}  median {np.median(t):8.4f} ms "
                  f"{algo / np.median(t) / 1e6:7.1f} GB/s   best {t.min():8.4f} ms "
                  f"{algo / t.min() / 1e6:7.1f} GB/s", flush=True)
    store.destroy()
    ctx.close()


if __name__ == "__main__":
    main()
