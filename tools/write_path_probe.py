#!/usr/bin/env python3
"""Times the write path of the host mirror over GPU stores: write_vector_points + the build loop
(src/update/index_documents/typed_chunk.rs:562-710, index_documents/mod.rs:544-566) for N synthetic
documents with one generated embedding each (every 16th: two).

    python tools/write_path_probe.py [--docs 100000] [--dim 768]

VS_MIRROR_WRITE_EACH=1 in the environment: every add_item is an upload of its own (no batching),
which is what the writer did before it kept its pending changes until the build."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=100_000)
    ap.add_argument("--dim", type=int, default=768)
    args = ap.parse_args()
    from search_engine_amd import _host, abi
    with abi.Context(0) as ctx:
        out = {}
        for run, category in (("first", 30), ("second", 31)):  # the first run pays the staging buffers
            writers = _host.GpuWriters(ctx.handle.value, category, args.dim, max(args.docs, 64))
            out[run] = _host.time_write_path(writers, args.dim, args.docs)
            del writers
            stores = [ctx.open_store((category << 8) | k) for k in range(2)]
            assert [len(s) for s in stores] == [args.docs, -(-args.docs // 16)]
            for s in stores:
                s.destroy()
        print(json.dumps({"docs": args.docs, "dim": args.dim,
                          "writer": "one upload per item" if os.environ.get("VS_MIRROR_WRITE_EACH") else "batched until the build",
                          "seconds_first_run": round(out["first"], 4), "seconds": round(out["second"], 4),
                          "items_per_second": round((args.docs + -(-args.docs // 16)) / out["second"])}))


if __name__ == "__main__":
    main()
