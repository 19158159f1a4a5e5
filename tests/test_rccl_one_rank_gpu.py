"""The multi-rank code path of bench.py under RCCL, rehearsed with a one-rank group on a single
GPU (VS_BENCH_FORCE_DIST=1): process-group set-up next to the library's own HIP runtime use,
barrier, all_gather_into_tensor of the keys, the batch merge, the all-reduce of the time. The
sharding arithmetic itself is covered for world sizes 2 and 3 in tests/test_sharded_gloo.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(*flags, force_dist):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    env.pop("VS_BENCH_FORCE_DIST", None)
    if force_dist:
        env["VS_BENCH_FORCE_DIST"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--sweep", "", *flags],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_single_query_bench_under_a_one_rank_rccl_group():
    line = bench("--docs", "300000", "--pass-percent", "50", force_dist=True)
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["scaling"] == "strong"
    for key in ("metric", "unit", "ms_per_step", "roofline", "cpu_baseline", "config", "dtype", "data"):
        assert key in line
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1


def test_batched_bench_under_a_one_rank_rccl_group():
    line = bench("--docs", "200000", "--dtype", "bf16", "--batch-queries", "256", force_dist=True)
    assert line["n_gpus"] == 1 and line["value"] > 0
