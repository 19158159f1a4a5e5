"""The multi-rank code path of bench.py under RCCL, rehearsed with a one-rank group on a single
GPU (VS_BENCH_FORCE_DIST=1): process-group set-up next to the library's own HIP runtime use and
its own RCCL communicator, the broadcast of the group id, barrier, the all-gather of the keys
inside vs_search_sharded_device, the batch merge, the all-reduce of the time. The launcher that
starts the ranks is covered on the CPU in tests/test_bench_launch.py, the sharding arithmetic for
world sizes 2 and 3 in tests/test_sharded_gloo.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(*flags, force_dist, no_group=False):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    env.pop("VS_BENCH_FORCE_DIST", None)
    env.pop("VS_BENCH_NO_GROUP", None)
    if force_dist:
        env["VS_BENCH_FORCE_DIST"] = "1"
    if no_group:
        env["VS_BENCH_NO_GROUP"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--sweep", "", *flags],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout  # one JSON line and nothing else on stdout
    return json.loads(lines[0])


def test_single_query_bench_under_a_one_rank_rccl_group():
    line = bench("--docs", "300000", "--pass-percent", "50", force_dist=True)
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["scaling"] == "strong"
    for key in ("metric", "unit", "ms_per_step", "roofline", "cpu_baseline", "config", "dtype", "data"):
        assert key in line
    r = line["roofline"]
    assert r["bound"] == "hbm" and line["config"]["ranks_in_group"] == 1
    # the accounting, not the speed: bytes = passing rows * (row + norm) + mask, rate = bytes / launch time
    # one scan launch takes the step's 16 queries; no row read is shared, so its bytes are 16 queries' bytes
    assert r["queries_per_launch"] == 16 and r["algorithmic_bytes_per_launch"] == 16 * r["algorithmic_bytes_per_query"]
    assert abs(r["ms_per_query"] - r["avg_launch_ms"] / 16) < 1e-3
    n_pass = (r["algorithmic_bytes_per_query"] - 300032 // 8) / (768 * 4 + 4)
    assert 0.45 * 300000 < n_pass < 0.55 * 300000
    # no counter pass was recorded for this workload: no traffic, hence nothing stale
    assert r["traffic"] is None and r["traffic_stale"] is False
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 0.02 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3 and 0.02 < r["frac"] < 1


def test_hamming_bench_line():
    line = bench("--docs", "2000000", "--dtype", "bin1", force_dist=False)
    r = line["roofline"]
    assert line["config"]["k"] == 100 and line["config"]["dim"] == 1536
    assert r["algorithmic_bytes_per_query"] == 2000000 * 192 + 2000000 // 8 and 0.02 < r["frac"] < 1


def test_batched_bench_under_a_one_rank_rccl_group():
    line = bench("--docs", "200000", "--dtype", "bf16", "--batch-queries", "256", force_dist=True)
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert 0 < line["roofline"]["frac"] < 1 and 0 < line["roofline"]["mfma_frac"] < 1


def test_shared_pass_f32_bench_line_stays_below_the_roofline():
    line = bench("--docs", "400000", "--batch-queries", "64", "--pass-percent", "10", force_dist=False)
    assert 0 < line["roofline"]["frac"] < 1 and line["roofline"]["passes_per_step"] == 2


def test_the_exchange_through_torch_distributed_when_the_group_cannot_be_made():
    """bench.py's other exchange (local keys, all_gather_into_tensor, one merge launch), taken by every
    rank when vs_group_create fails on any: the same accounting, and the line says which one ran."""
    for flags in (("--docs", "300000", "--pass-percent", "50"),
                  ("--docs", "200000", "--dtype", "bf16", "--batch-queries", "256")):
        via_group = bench(*flags, force_dist=True)
        via_torch = bench(*flags, force_dist=True, no_group=True)
        assert via_group["config"]["collective"].startswith("vs_group")
        assert via_torch["config"]["collective"].startswith("torch.distributed")
        assert via_torch["value"] > 0 and 0 < via_torch["roofline"]["frac"] < 1
        assert via_torch["roofline"]["algorithmic_bytes_per_launch"] == via_group["roofline"]["algorithmic_bytes_per_launch"]


def test_cpu_baseline_legs_and_the_sample_size():
    """The CPU leg samples --cpu-sample-rows rows across upload chunks (not just the first chunk), and
    one sweep point gets the same baseline under its filter."""
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--docs", "400000", "--cpu-sample-rows", "300000", "--sweep", "10"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][0])
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["pass_percent"] == 100 and "first 300000 of 400000 rows" in cpu["sample"]
    assert cpu["filtered"]["pass_percent"] == 10 and "facet pass rate 10%" in cpu["filtered"]["sample"]
    assert cpu["filtered"]["value"] > cpu["value"] > 0  # a tenth of the rows to score
    assert line["sweep"][0]["pass_percent"] == 10 and line["sweep"][0]["traffic_stale"] is False
