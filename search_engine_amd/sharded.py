"""Host-side helpers for one logical store sharded over several ranks.

On the GPU the whole sharded search is one C-ABI call (vs_search_sharded_device: local
scans, the RCCL all-gather, the merge). This module holds what a host needs around it —
the shard arithmetic — plus CPU restatements of the key format and of the merge's layout
contract, with which the tests rehearse the multi-rank data flow under gloo.

Documents are split into contiguous docid ranges, one per rank (one process per
GPU), boundaries on multiples of 64 so that a dense candidate mask slices by
word offset. A batch of queries is searched on every shard; ONE all-gather then
moves the shard-local top-k keys (k * 8 bytes per query and rank) and one launch
merges them. The merge is associative, so the result is identical to searching
one store that holds everything.

The reference has no counterpart (it is a single-process library); this is the
multi-GPU form of VectorSort::fill_buffer's per-store loop
(src/search/ranking/vector_sort.rs:73-78): shards play the role of stores, and
the (distance, docid) order of the merged list is the one arroy returns.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(rank, world, n_docs, align=64):
    """(first_docid, count) of `rank`'s shard; every boundary is a multiple of `align`."""
    per = -(-n_docs // world)
    per = -(-per // align) * align
    first = min(rank * per, n_docs)
    return first, max(0, min(n_docs, first + per) - first)


# ---- the 64-bit sort key: (orderable(distance) << 32) | docid -------------------------------------
# Host restatement of the device key so that CPU code (tests, tools) can build and
# read key lists; the GPU path never calls these.

KEY_NONE = np.uint64(0xFFFFFFFFFFFFFFFF)


def pack_keys(distances, docids):
    d = np.ascontiguousarray(distances, np.float32) + np.float32(0.0)  # -0.0 -> +0.0
    u = d.view(np.uint32).astype(np.uint64)
    neg = (u >> np.uint64(31)) == 1
    o = np.where(neg, (~u) & np.uint64(0xFFFFFFFF), u | np.uint64(0x80000000))
    o = np.where(np.isnan(d), np.uint64(0xFFC00000), o)
    return (o << np.uint64(32)) | np.ascontiguousarray(docids, np.uint32).astype(np.uint64)


def unpack_keys(keys):
    keys = np.ascontiguousarray(keys).view(np.uint64)
    keys = keys[keys != KEY_NONE]
    o = (keys >> np.uint64(32)).astype(np.uint32)
    u = np.where(o & np.uint32(0x80000000), o & np.uint32(0x7FFFFFFF), ~o)
    return (keys & np.uint64(0xFFFFFFFF)).astype(np.uint32), u.astype(np.uint32).view(np.float32)


class ShardedSearch:
    """Batch search over a sharded store.

    local_search(j, out_keys) must leave the shard-local ascending top-k keys of
    the batch's j-th query in out_keys (an int64 tensor view of k entries, padded
    with KEY_NONE); merge(gathered, world, n_queries, k) turns the all-gathered
    keys [world][n_queries][k] into (ids, distances, counts). The GPU library does all of
    this inside vs_search_sharded_device; the tests inject CPU stand-ins here to exercise the
    sharding and the collective's layout under gloo.
    """

    def __init__(self, local_search, merge, k, batch, device, group=None, local_search_all=None,
                 always_gather=False):
        self.local_search = local_search
        # optional: searches queries 0..n-1 in one go, keys[j*k:(j+1)*k] = query j (one pass over
        # the shard for all of them instead of one per query)
        self.local_search_all = local_search_all
        self.merge = merge
        self.k = k
        self.batch = batch
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # always_gather: run the collective even in a one-rank group (a rehearsal of the
        # multi-rank path on a single GPU)
        self.gather = self.world > 1 or (always_gather and dist.is_initialized())
        self.keys = torch.empty(batch * k, dtype=torch.int64, device=device)
        self.gathered = (torch.empty(self.world * batch * k, dtype=torch.int64, device=device)
                         if self.gather else self.keys)

    def run(self, n_queries=None):
        """Searches queries 0..n_queries-1 of the current batch (default: a full batch)."""
        n = self.batch if n_queries is None else n_queries
        if self.local_search_all is not None:
            self.local_search_all(n, self.keys)
        else:
            for j in range(n):
                self.local_search(j, self.keys[j * self.k:(j + 1) * self.k])
        if self.gather:
            dist.all_gather_into_tensor(self.gathered, self.keys, group=self.group)
        return self.merge(self.gathered, self.world, n, self.k)


def merge_keys_cpu(keys, n_lists, list_stride, n_queries, query_stride, k):
    """Host restatement of the merge launch's layout contract (vs_merge_topk_batch_device): list i
    of query q starts at keys[q * query_stride + i * list_stride] and holds k ascending keys padded
    with KEY_NONE; the result per query is the k smallest keys of its lists, ascending (keys are
    unique: a docid lives in one shard), padded with KEY_NONE. For the all-gathered layout
    [rank][query][k]: list_stride = n_queries * k, query_stride = k. Returns u64 [n_queries][k]."""
    keys = np.ascontiguousarray(keys).view(np.uint64).reshape(-1)
    out = np.full((n_queries, k), KEY_NONE, np.uint64)
    for q in range(n_queries):
        lists = [keys[q * query_stride + i * list_stride: q * query_stride + i * list_stride + k]
                 for i in range(n_lists)]
        merged = np.sort(np.concatenate(lists))[:k]
        out[q, : merged.size] = merged
    return out
