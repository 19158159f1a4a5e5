#!/usr/bin/env python3
"""Soak run of tests/test_store_fuzz_gpu.py's random writer histories over many more seeds than the
test suite carries: tools/fuzz_soak.py [first_seed] [n_seeds]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from search_engine_amd import abi  # noqa: E402
import test_store_fuzz_gpu as fuzz  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ctx = abi.Context(0)
for seed in range(first, first + count):
    for dtype_name in ("f32", "bf16"):
        fuzz.test_random_writer_histories(ctx, seed, dtype_name)
    print("seed", seed, "ok", flush=True)
ctx.close()
