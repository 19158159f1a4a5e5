"""bench.py's launcher, rehearsed on the CPU: `--gpus N` without a rank environment must start N
ranks by itself (fresh child processes), and the same file must run when torch.distributed.run
starts it. `--dry-launch` makes every rank report its shard under gloo instead of searching."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "VS_BENCH_LAUNCHED_BY"):
        env.pop(key, None)
    return env


def _one_line(out):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def _check_ranks(line, world, docs):
    from search_engine_amd.sharded import shard_range
    assert line["dry_launch"] and line["world"] == world and line["n_gpus"] == world
    assert [r["rank"] for r in line["ranks"]] == list(range(world))
    assert [r["local_rank"] for r in line["ranks"]] == list(range(world))
    covered = 0
    for r in line["ranks"]:
        assert r["world"] == world
        assert (r["first_docid"], r["count"]) == shard_range(r["rank"], world, docs)
        assert r["first_docid"] % 64 == 0
        covered += r["count"]
    assert covered == docs


@pytest.mark.parametrize("world,dtype", [(2, "f32"), (4, "bin1")])
def test_gpus_n_starts_its_own_ranks(world, dtype):
    out = subprocess.run([sys.executable, BENCH, "--gpus", str(world), "--dry-launch", "--docs", "100003",
                          "--dtype", dtype], env=_clean_env(), capture_output=True, text=True, timeout=300)
    line = _one_line(out)
    assert line["launched_by"] == "self"
    _check_ranks(line, world, 100003)


@pytest.mark.parametrize("flags,docs", [(["--dtype", "f32"], 10_000_000),
                                        (["--dtype", "bf16", "--batch-queries", "256"], 10_000_000),
                                        (["--dtype", "bin1"], 50_000_000)])
def test_eight_ranks_at_the_benchmark_sizes(flags, docs):
    """The three commands of the 8-GPU runs (the f32 headline, BASELINE config 3, config 4), at their
    default sizes, as the driver starts them: eight ranks come up, agree on the world, and their shards
    tile the documents on multiples of 64."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-launch", *flags], env=_clean_env(),
                         capture_output=True, text=True, timeout=600)
    line = _one_line(out)
    assert line["docs"] == docs
    _check_ranks(line, 8, docs)


def test_started_by_torch_distributed_run():
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", BENCH, "--gpus", "2",
                          "--dry-launch", "--docs", "1000"], env=_clean_env(), capture_output=True, text=True,
                         timeout=300)
    line = _one_line(out)
    assert line["launched_by"] == "environment"
    _check_ranks(line, 2, 1000)


def test_world_size_mismatch_is_refused():
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


def test_recorded_traffic_is_tied_to_the_library_sources():
    """profiles/scan_traffic*.json carry a hash of the sources libvs_amd.so is built from; bench.py
    reports a number recorded on other sources as stale. The committed files must match the tree
    they are committed with, and the hash must not move with files that are not in the library."""
    import glob
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    now = bench.csrc_sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "scan_traffic*.json")))
    assert files
    for f in files:
        with open(f) as fh:
            assert json.load(fh).get("csrc_sha256") == now, f"{os.path.basename(f)} was recorded on other kernel sources"
    mirror = os.path.join(ROOT, "search_engine_amd", "csrc", "host_engine.cpp")
    with open(mirror, "rb") as fh:
        before = fh.read()
    try:
        with open(mirror, "ab") as fh:
            fh.write(b"\n// (test)\n")
        assert bench.csrc_sha256() == now
    finally:
        with open(mirror, "wb") as fh:
            fh.write(before)
