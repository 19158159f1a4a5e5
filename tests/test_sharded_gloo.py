"""The multi-rank path (docid-range shards, one all-gather of top-k keys, merge) under
torch.distributed with the gloo backend, world size 2 and 3, on the CPU. The
shard-local scan is stood in for by the CPU oracle and the final merge by the host
restatement of the merge launch's layout contract (sharded.merge_keys_cpu, which
tests/test_shard_merge_gpu.py pins to the kernel); what is under test is the sharding
arithmetic, the key format and the collective's layout ([rank][query][k])."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import facet_mask, synthetic_vectors


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_docs, dim, k, batch, out_dir, all_at_once=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import scan as oracle
        from search_engine_amd import sharded
        vectors = synthetic_vectors(n_docs, dim, 0xC0FFEE)   # every rank regenerates the same set
        queries = synthetic_vectors(batch, dim, 42)
        mask = facet_mask(n_docs, 50)
        first, count = sharded.shard_range(rank, world, n_docs)
        local = vectors[first:first + count]
        docids = np.arange(first, first + count, dtype=np.uint32)

        def local_search(j, out_keys):
            keys = np.full(k, sharded.KEY_NONE, np.uint64)
            if count:
                ids, d = oracle.nns_by_vector(local, queries[j], k, docids=docids, mask=mask)
                keys[: ids.size] = sharded.pack_keys(d, ids)
            out_keys.copy_(torch.from_numpy(keys.view(np.int64)))

        def merge(gathered, w, n_q, kk):
            # the merge launch's layout contract on the gathered [rank][query][k] buffer
            # (list_stride = batch * k, query_stride = k); tests/test_shard_merge_gpu.py holds
            # vs_merge_topk_batch_device to the same function
            merged = sharded.merge_keys_cpu(gathered.numpy(), w, batch * kk, n_q, kk, kk)
            return [sharded.unpack_keys(merged[q]) for q in range(n_q)]

        def local_search_all(n, keys):  # the batched form: every query of the step in one call
            for j in range(n):
                local_search(j, keys[j * k:(j + 1) * k])

        if all_at_once:
            s = sharded.ShardedSearch(None, merge, k, batch, torch.device("cpu"), local_search_all=local_search_all)
        else:
            s = sharded.ShardedSearch(local_search, merge, k, batch, torch.device("cpu"))
        results = s.run()
        if rank == 0:
            np.savez(os.path.join(out_dir, "result.npz"),
                     ids=np.stack([r[0] for r in results]), dist=np.stack([r[1] for r in results]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_docs,all_at_once", [(2, 1000, False), (3, 1000, False), (2, 70, False),
                                                      (2, 1000, True)])
def test_sharded_search_equals_single_store(tmp_path, world, n_docs, all_at_once):
    from oracle import scan as oracle
    dim, k, batch = 32, 10, 4
    mp.spawn(_worker, args=(world, _free_port(), n_docs, dim, k, batch, str(tmp_path), all_at_once), nprocs=world,
             join=True)
    got = np.load(tmp_path / "result.npz")
    vectors = synthetic_vectors(n_docs, dim, 0xC0FFEE)
    queries = synthetic_vectors(batch, dim, 42)
    mask = facet_mask(n_docs, 50)
    for j in range(batch):
        ids, d = oracle.nns_by_vector(vectors, queries[j], k, mask=mask)
        assert got["ids"][j].tolist() == ids.tolist()
        assert got["dist"][j].tolist() == d.tolist()


def test_shard_ranges_tile_the_docid_space():
    from search_engine_amd import sharded
    for n_docs in (0, 1, 63, 64, 65, 1000, 10_000_000):
        for world in (1, 2, 3, 4, 8):
            covered = 0
            for r in range(world):
                first, count = sharded.shard_range(r, world, n_docs)
                assert first % 64 == 0 or count == 0
                assert first == covered or count == 0
                covered += count
            assert covered == n_docs


def test_key_pack_orders_like_ordered_float():
    from search_engine_amd import sharded
    d = np.array([0.5, 0.25, 0.25, -0.0, 0.0, np.inf, np.nan, -1e-8, 1.0], np.float32)
    ids = np.array([9, 8, 7, 6, 5, 4, 3, 2, 1], np.uint32)
    keys = sharded.pack_keys(d, ids)
    order = np.argsort(keys, kind="stable")
    # -1e-8 < (-0.0 == +0.0, by docid) < 0.25 (by docid) < 0.5 < 1.0 < inf < NaN
    assert ids[order].tolist() == [2, 5, 6, 7, 8, 9, 1, 4, 3]
    back_ids, back_d = sharded.unpack_keys(keys)
    assert back_ids.tolist() == ids.tolist()
    np.testing.assert_array_equal(np.nan_to_num(back_d, nan=-7.0), np.nan_to_num(d + np.float32(0), nan=-7.0))
