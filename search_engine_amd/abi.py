"""ctypes binding of libvs_amd.so — one Python function per C-ABI entry point.

This is the same binding a host in another language would write against
include/vs_abi.h; tests and the benchmark call the library only through it.
There is no fallback: if the shared library is missing, importing this module
raises, and if no GPU is usable ``Context()`` raises ``VsError(VS_E_DEVICE)``.
"""
import ctypes
import os

import numpy as np

VS_OK, VS_E_INVALID_ARG, VS_E_DIM_MISMATCH, VS_E_NO_STORE, VS_E_DEVICE = 0, 1, 2, 3, 4
VS_E_NOT_SEALED, VS_E_OOM, VS_E_CAPACITY, VS_E_FORMAT = 5, 6, 7, 8
VS_F32, VS_BF16, VS_BIN1 = 0, 1, 2
VS_ANGULAR, VS_HAMMING, VS_EUCLIDEAN, VS_DOT = 0, 1, 2, 3
VS_KEY_NONE = 0xFFFFFFFFFFFFFFFF
VS_MAX_K = 1024
VS_GROUP_ID_BYTES = 128
VS_SHARDED_SHARE_PASSES, VS_SHARDED_ALWAYS_GATHER = 1, 2

# VS_AMD_LIB names another build of the same library (measurement variants of the tools)
LIB_PATH = os.environ.get("VS_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvs_amd.so")

_vp = ctypes.c_void_p
_u8p = ctypes.POINTER(ctypes.c_uint8)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_f32p = ctypes.POINTER(ctypes.c_float)
_szp = ctypes.POINTER(ctypes.c_size_t)
_u16, _u32, _i32, _u64, _sz = (ctypes.c_uint16, ctypes.c_uint32, ctypes.c_int32, ctypes.c_uint64,
                               ctypes.c_size_t)

# name -> argtypes; every entry point returns int32_t
SIGNATURES = {
    "vs_init": [_i32, ctypes.POINTER(_vp)],
    "vs_shutdown": [_vp],
    "vs_last_error": [_vp, ctypes.c_char_p, _sz],
    "vs_last_dim_mismatch": [_vp, _u32p, _u32p],
    "vs_set_scan_tuning": [_vp, _i32, _i32],
    "vs_store_create": [_vp, _u16, _u32, _i32, _i32, _u32, _u64, ctypes.POINTER(_vp)],
    "vs_store_open": [_vp, _u16, ctypes.POINTER(_vp)],
    "vs_store_upload": [_vp, _u32p, _f32p, _u64],
    "vs_store_upload_device": [_vp, _u32, _vp, _u64, _vp],
    "vs_store_remove": [_vp, _u32p, _u64],
    "vs_store_reserve": [_vp, _u64],
    "vs_store_seal": [_vp],
    "vs_store_destroy": [_vp],
    "vs_store_len": [_vp, _u64p],
    "vs_store_dim": [_vp, _u32p],
    "vs_store_row_bytes": [_vp, _u64p],
    "vs_store_contains": [_vp, _u32, _i32p],
    "vs_store_item_vector": [_vp, _u32, _f32p],
    "vs_store_writer_len": [_vp, _u64p],
    "vs_ingest_arroy_items": [_vp, ctypes.POINTER(_u8p), _szp, ctypes.POINTER(_u8p), _szp, _u64, _u64p, _u64p, _u64p],
    "vs_arroy_parse_key": [_u8p, _sz, ctypes.POINTER(_u16), _u8p, _u32p],
    "vs_arroy_encode_item": [_u16, _u32, _f32p, _u32, _u8p, _u8p, _sz, _szp],
    "vs_search": [_vp, _f32p, _u32, _u32, _u64p, _u64, _u32p, _f32p, _u32p],
    "vs_search_roaring": [_vp, _f32p, _u32, _u32, _u8p, _sz, _u32p, _f32p, _u32p],
    "vs_search_batch": [_vp, _f32p, _u32, _u32, _u32, _u64p, _u64, _u32p, _f32p, _u32p],
    "vs_search_batch_masks": [_vp, _f32p, _u32, _u32, _u32, ctypes.POINTER(_u64p), _u64p,
                              ctypes.POINTER(_vp), _u32p, _f32p, _u32p],
    "vs_mask_create": [_vp, _u64p, _u64, ctypes.POINTER(_vp)],
    "vs_mask_create_roaring": [_vp, _u8p, _sz, ctypes.POINTER(_vp)],
    "vs_mask_destroy": [_vp],
    "vs_mask_fits": [_vp, _vp],
    "vs_search_mask": [_vp, _f32p, _u32, _u32, _vp, _u32p, _f32p, _u32p],
    "vs_search_stores": [_vp, ctypes.c_uint16, _f32p, _u32, _u32, _u64p, _u64, _u8p, _sz, _vp, _u32p, _f32p, _u8p,
                         _u32, _u32p, _u32p],
    "vs_search_workspace_bytes": [_vp, _u32, _u64p],
    "vs_search_device": [_vp, _vp, _u32, _u32, _vp, _vp, _u64, _vp, _vp],
    "vs_search_device_noalloc": [_vp, _vp, _u32, _u32, _vp, _vp, _vp],
    "vs_search_batch_workspace_bytes": [_vp, _u32, _u64p],
    "vs_search_batch_device": [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _u64, _vp, _vp],
    "vs_merge_topk_device": [_vp, _vp, _u32, _u32, _i32, _vp, _vp, _vp, _vp],
    "vs_merge_topk_batch_device": [_vp, _vp, _u32, _u64, _u32, _u64, _u32, _i32, _vp, _vp, _vp, _vp],
    "vs_group_unique_id": [_u8p],
    "vs_group_create": [_vp, _i32, _i32, _u8p, ctypes.POINTER(_vp)],
    "vs_group_destroy": [_vp],
    "vs_search_sharded_device": [_vp, _vp, _vp, _u32, _u32, _u32, _vp, _u32, _vp, _vp, _vp, _vp],
    "vs_roaring_to_mask": [_u8p, _sz, _u64p, _u64, _u64p],
    "vs_mask_to_roaring": [_u64p, _u64, _u8p, _sz, _szp],
    "vs_cbo_roaring_to_mask": [_u8p, _sz, _u64p, _u64, _u64p],
    "vs_mask_to_cbo_roaring": [_u64p, _u64, _u8p, _sz, _szp],
    "vs_bo_roaring_to_mask": [_u8p, _sz, _u64p, _u64, _u64p],
}


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `make -C search_engine_amd/csrc` "
            "(or __graft_entry__.build()); there is no fallback implementation")
    # A process must run ONE HIP runtime. PyTorch wheels bundle their own
    # libamdhip64 (same SONAME as the system one); if torch is going to be used in
    # this process it has to be loaded first so that libvs_amd.so binds to the same
    # copy — two runtimes in one process leave the second without a GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int32
    return lib


lib = load_library()


class VsError(RuntimeError):
    def __init__(self, code, message, expected=None, received=None):
        super().__init__(f"[vs error {code}] {message}")
        self.code = code
        self.message = message
        self.expected = expected
        self.received = received


def _raise(code, ctx_handle=None):
    buf = ctypes.create_string_buffer(512)
    lib.vs_last_error(ctx_handle, buf, 512)
    exp, rec = ctypes.c_uint32(0), ctypes.c_uint32(0)
    if code == VS_E_DIM_MISMATCH:
        lib.vs_last_dim_mismatch(ctx_handle, ctypes.byref(exp), ctypes.byref(rec))
    raise VsError(code, buf.value.decode(errors="replace"), exp.value, rec.value)


def _check(code, ctx_handle=None):
    if code != VS_OK:
        _raise(code, ctx_handle)


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _bytes_ptr(b):
    arr = np.frombuffer(b, dtype=np.uint8) if len(b) else np.zeros(1, np.uint8)
    return arr, arr.ctypes.data_as(_u8p)


class Context:
    """vs_ctx: binds this process to one GPU."""

    def __init__(self, device=0):
        h = _vp()
        _check(lib.vs_init(device, ctypes.byref(h)))
        self.handle = h
        self.device = device

    def close(self):
        if self.handle:
            lib.vs_shutdown(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_scan_tuning(self, blocks=0, nt=1):
        _check(lib.vs_set_scan_tuning(self.handle, blocks, nt), self.handle)

    def create_store(self, index, dim, capacity, dtype=VS_F32, metric=VS_ANGULAR, docid_base=0):
        h = _vp()
        _check(lib.vs_store_create(self.handle, index, dim, dtype, metric, docid_base, capacity,
                                   ctypes.byref(h)), self.handle)
        return Store(self, h)

    def open_store(self, index):
        """Mirror of Context::vector_reader: raises VsError(VS_E_NO_STORE) if absent."""
        h = _vp()
        _check(lib.vs_store_open(self.handle, index, ctypes.byref(h)), self.handle)
        return Store(self, h)

    def ingest_arroy_items(self, pairs):
        """pairs: iterable of (key bytes, value bytes) of the reference's vector-arroy database.
        Returns (item leaves uploaded, other nodes skipped, leaves without a store)."""
        pairs = list(pairs)
        n = len(pairs)
        keep = [(np.frombuffer(k, np.uint8), np.frombuffer(v, np.uint8) if len(v) else np.zeros(1, np.uint8))
                for k, v in pairs]
        keys = (_u8p * max(n, 1))(*[a.ctypes.data_as(_u8p) for a, _ in keep])
        vals = (_u8p * max(n, 1))(*[b.ctypes.data_as(_u8p) for _, b in keep])
        klen = (_sz * max(n, 1))(*[len(k) for k, _ in pairs])
        vlen = (_sz * max(n, 1))(*[len(v) for _, v in pairs])
        a, b, c = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
        _check(lib.vs_ingest_arroy_items(self.handle, keys, klen, vals, vlen, n, ctypes.byref(a), ctypes.byref(b),
                                         ctypes.byref(c)), self.handle)
        return a.value, b.value, c.value

    def search_stores(self, index_base, query, k, mask=None, roaring_bytes=None, mask_handle=None, cap=None):
        """vs_search_stores: one query against every store index_base | 0, | 1, .. under one filter
        (dense u64 words, a serialised RoaringBitmap, or a handle from Store.create_mask; none = every
        item). Returns (ids, distances, store numbers, n_stores), entries ascending by (distance,
        docid, store). cap: entries of the output buffers (default: room for 256 stores)."""
        query = np.ascontiguousarray(query, np.float32)
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint64)
        cap = 256 * max(k, 1) if cap is None else cap
        ids, dist, src = np.zeros(max(cap, 1), np.uint32), np.zeros(max(cap, 1), np.float32), np.zeros(max(cap, 1), np.uint8)
        n, n_stores = ctypes.c_uint32(0), ctypes.c_uint32(0)
        if roaring_bytes is not None:
            keep, rp = _bytes_ptr(roaring_bytes)
        else:
            keep, rp = None, None
        _check(lib.vs_search_stores(self.handle, index_base, _ptr(query, _f32p), query.size, k, _ptr(mask, _u64p),
                                    0 if mask is None else mask.size, rp, 0 if roaring_bytes is None else len(roaring_bytes),
                                    mask_handle, _ptr(ids, _u32p), _ptr(dist, _f32p), _ptr(src, _u8p), cap,
                                    ctypes.byref(n), ctypes.byref(n_stores)), self.handle)
        return ids[: n.value].copy(), dist[: n.value].copy(), src[: n.value].copy(), n_stores.value

    def merge_topk_device(self, d_keys, n_lists, k, metric, d_ids, d_dist, d_n, stream=0):
        _check(lib.vs_merge_topk_device(self.handle, d_keys, n_lists, k, metric, d_ids, d_dist, d_n,
                                        stream), self.handle)


    def merge_topk_batch_device(self, d_keys, n_lists, list_stride, n_queries, query_stride, k,
                                metric, d_ids, d_dist, d_n, stream=0):
        _check(lib.vs_merge_topk_batch_device(self.handle, d_keys, n_lists, list_stride, n_queries,
                                              query_stride, k, metric, d_ids, d_dist, d_n, stream),
               self.handle)


def group_unique_id():
    """The bytes rank 0 hands to every rank before Group()."""
    out = np.zeros(VS_GROUP_ID_BYTES, np.uint8)
    _check(lib.vs_group_unique_id(_ptr(out, _u8p)))
    return out.tobytes()


class Group:
    """vs_group: the ranks (one process per GPU) that hold the shards of one logical store."""

    def __init__(self, ctx, rank, world, unique_id):
        keep, bp = _bytes_ptr(unique_id)
        assert len(unique_id) == VS_GROUP_ID_BYTES
        h = _vp()
        _check(lib.vs_group_create(ctx.handle, rank, world, bp, ctypes.byref(h)), ctx.handle)
        self.ctx, self.handle, self.rank, self.world = ctx, h, rank, world

    def search_sharded_device(self, store, d_queries, n_q, dim, k, d_mask, flags, d_ids, d_dist, d_n, stream=0):
        _check(lib.vs_search_sharded_device(self.handle, store.handle, d_queries, n_q, dim, k, d_mask, flags,
                                            d_ids, d_dist, d_n, stream), self.ctx.handle)

    def destroy(self):
        if self.handle:
            lib.vs_group_destroy(self.handle)
            self.handle = None


class Store:
    """vs_store: one arroy index (embedder_id << 8 | k) as a row slab in HBM."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.handle = handle

    def _ck(self, code):
        _check(code, self.ctx.handle)

    def upload(self, docids, rows):
        docids = np.ascontiguousarray(docids, np.uint32)
        rows = np.ascontiguousarray(rows, np.float32)
        assert rows.ndim == 2 and rows.shape[0] == docids.size
        self._ck(lib.vs_store_upload(self.handle, _ptr(docids, _u32p), _ptr(rows, _f32p), docids.size))

    def upload_device(self, first_docid, device_ptr, n, stream=0):
        self._ck(lib.vs_store_upload_device(self.handle, first_docid, device_ptr, n, stream))

    def remove(self, docids):
        docids = np.ascontiguousarray(docids, np.uint32)
        self._ck(lib.vs_store_remove(self.handle, _ptr(docids, _u32p), docids.size))

    def reserve(self, capacity):
        self._ck(lib.vs_store_reserve(self.handle, capacity))

    def seal(self):
        self._ck(lib.vs_store_seal(self.handle))

    def destroy(self):
        if self.handle:
            lib.vs_store_destroy(self.handle)
            self.handle = None

    def __len__(self):
        n = ctypes.c_uint64(0)
        self._ck(lib.vs_store_len(self.handle, ctypes.byref(n)))
        return n.value

    def writer_len(self):
        n = ctypes.c_uint64(0)
        self._ck(lib.vs_store_writer_len(self.handle, ctypes.byref(n)))
        return n.value

    @property
    def dim(self):
        d = ctypes.c_uint32(0)
        self._ck(lib.vs_store_dim(self.handle, ctypes.byref(d)))
        return d.value

    @property
    def row_bytes(self):
        b = ctypes.c_uint64(0)
        self._ck(lib.vs_store_row_bytes(self.handle, ctypes.byref(b)))
        return b.value

    def contains(self, docid):
        out = ctypes.c_int32(0)
        self._ck(lib.vs_store_contains(self.handle, docid, ctypes.byref(out)))
        return bool(out.value)

    def item_vector(self, docid):
        out = np.zeros(self.dim, np.float32)
        self._ck(lib.vs_store_item_vector(self.handle, docid, _ptr(out, _f32p)))
        return out

    def search(self, query, k, mask=None):
        """(ids, distances), ascending by (distance, docid); mask = dense u64 words or None."""
        query = np.ascontiguousarray(query, np.float32)
        mask = None if mask is None else np.ascontiguousarray(mask, np.uint64)
        ids = np.zeros(max(k, 1), np.uint32)
        dist = np.zeros(max(k, 1), np.float32)
        n = ctypes.c_uint32(0)
        self._ck(lib.vs_search(self.handle, _ptr(query, _f32p), query.size, k, _ptr(mask, _u64p),
                               0 if mask is None else mask.size, _ptr(ids, _u32p), _ptr(dist, _f32p),
                               ctypes.byref(n)))
        return ids[: n.value].copy(), dist[: n.value].copy()

    def search_roaring(self, query, k, roaring_bytes):
        query = np.ascontiguousarray(query, np.float32)
        keep, bp = _bytes_ptr(roaring_bytes)
        ids = np.zeros(max(k, 1), np.uint32)
        dist = np.zeros(max(k, 1), np.float32)
        n = ctypes.c_uint32(0)
        self._ck(lib.vs_search_roaring(self.handle, _ptr(query, _f32p), query.size, k, bp,
                                       len(roaring_bytes), _ptr(ids, _u32p), _ptr(dist, _f32p),
                                       ctypes.byref(n)))
        return ids[: n.value].copy(), dist[: n.value].copy()

    def create_mask(self, mask=None, roaring_bytes=None):
        """A candidate filter kept on the device (dense words or a serialised RoaringBitmap);
        returns an opaque handle for search_mask / destroy_mask."""
        h = _vp()
        if roaring_bytes is not None:
            keep, bp = _bytes_ptr(roaring_bytes)
            self._ck(lib.vs_mask_create_roaring(self.handle, bp, len(roaring_bytes), ctypes.byref(h)))
        else:
            mask = np.ascontiguousarray(mask, np.uint64)
            self._ck(lib.vs_mask_create(self.handle, _ptr(mask, _u64p), mask.size, ctypes.byref(h)))
        return h

    def destroy_mask(self, handle):
        self._ck(lib.vs_mask_destroy(handle))

    def mask_fits(self, handle):
        """Whether the mask was created for this store at its present size."""
        return bool(lib.vs_mask_fits(self.handle, handle))

    def search_mask(self, query, k, mask_handle):
        query = np.ascontiguousarray(query, np.float32)
        ids = np.zeros(max(k, 1), np.uint32)
        dist = np.zeros(max(k, 1), np.float32)
        n = ctypes.c_uint32(0)
        self._ck(lib.vs_search_mask(self.handle, _ptr(query, _f32p), query.size, k, mask_handle,
                                    _ptr(ids, _u32p), _ptr(dist, _f32p), ctypes.byref(n)))
        return ids[: n.value].copy(), dist[: n.value].copy()

    def search_batch(self, queries, k, mask=None):
        queries = np.ascontiguousarray(queries, np.float32)
        mask = None if mask is None else np.ascontiguousarray(mask, np.uint64)
        n_q, dim = queries.shape
        ids = np.zeros((n_q, max(k, 1)), np.uint32)
        dist = np.zeros((n_q, max(k, 1)), np.float32)
        cnt = np.zeros(n_q, np.uint32)
        self._ck(lib.vs_search_batch(self.handle, _ptr(queries, _f32p), n_q, dim, k,
                                     _ptr(mask, _u64p), 0 if mask is None else mask.size,
                                     _ptr(ids, _u32p), _ptr(dist, _f32p), _ptr(cnt, _u32p)))
        return [(ids[i, : cnt[i]].copy(), dist[i, : cnt[i]].copy()) for i in range(n_q)]

    def search_batch_masks(self, queries, k, masks):
        """One filter per query: masks[i] is None (every item), dense u64 words, or a handle
        from create_mask."""
        queries = np.ascontiguousarray(queries, np.float32)
        n_q, dim = queries.shape
        assert len(masks) == n_q
        words = (_u64p * n_q)()
        lens = np.zeros(n_q, np.uint64)
        handles = (_vp * n_q)()
        keep = []
        for i, m in enumerate(masks):
            if m is None:
                continue
            if isinstance(m, np.ndarray):
                keep.append(m if m.dtype == np.uint64 and m.flags.c_contiguous else np.ascontiguousarray(m, np.uint64))
                words[i] = keep[-1].ctypes.data_as(_u64p)
                lens[i] = keep[-1].size
            else:
                handles[i] = m
        ids = np.zeros((n_q, max(k, 1)), np.uint32)
        dist = np.zeros((n_q, max(k, 1)), np.float32)
        cnt = np.zeros(n_q, np.uint32)
        self._ck(lib.vs_search_batch_masks(self.handle, _ptr(queries, _f32p), n_q, dim, k, words,
                                           _ptr(lens, _u64p), handles, _ptr(ids, _u32p), _ptr(dist, _f32p),
                                           _ptr(cnt, _u32p)))
        return [(ids[i, : cnt[i]].copy(), dist[i, : cnt[i]].copy()) for i in range(n_q)]

    def workspace_bytes(self, k):
        b = ctypes.c_uint64(0)
        self._ck(lib.vs_search_workspace_bytes(self.handle, k, ctypes.byref(b)))
        return b.value

    def batch_workspace_bytes(self, k):
        b = ctypes.c_uint64(0)
        self._ck(lib.vs_search_batch_workspace_bytes(self.handle, k, ctypes.byref(b)))
        return b.value

    def search_batch_device(self, d_queries, n_q, dim, k, d_mask, d_workspace, d_out_keys, stream=0,
                            workspace_bytes=None):
        """vs_search_batch_device: up to 256 queries sharing passes over the rows (bf16 matrix cores,
        8 / 32 f32 queries per pass, 4 binary ones; other stores one scan per query); keys out. Raw
        device addresses, asynchronous on `stream`. workspace_bytes: the size of the buffer behind
        d_workspace (the library checks it); None = the caller sized it with batch_workspace_bytes(k)
        for the store as it is now."""
        if workspace_bytes is None:
            workspace_bytes = self.batch_workspace_bytes(k)
        self._ck(lib.vs_search_batch_device(self.handle, d_queries, n_q, dim, k, d_mask, d_workspace,
                                            workspace_bytes, d_out_keys, stream))

    def search_device(self, d_query, dim, k, d_mask, d_workspace, d_out_keys, stream=0, workspace_bytes=None):
        """All arguments are raw device addresses (ints); asynchronous on `stream`. workspace_bytes as in
        search_batch_device (None = sized with workspace_bytes(k) for the store as it is now)."""
        if workspace_bytes is None:
            workspace_bytes = self.workspace_bytes(k)
        self._ck(lib.vs_search_device(self.handle, d_query, dim, k, d_mask, d_workspace, workspace_bytes,
                                      d_out_keys, stream))

    def search_device_noalloc(self, d_query, dim, k, d_mask, d_out_keys, stream=0):
        """vs_search_device on the library's static scratch: no workspace, no allocation."""
        self._ck(lib.vs_search_device_noalloc(self.handle, d_query, dim, k, d_mask, d_out_keys, stream))


# ---- arroy's on-disk item layout (host only) ---------------------------------------------


def arroy_encode_item(index, item, vector):
    """(key bytes, leaf bytes) of one item as the reference's vector-arroy database holds it."""
    vector = np.ascontiguousarray(vector, np.float32)
    key = np.zeros(7, np.uint8)
    n = ctypes.c_size_t(0)
    _check(lib.vs_arroy_encode_item(index, item, _ptr(vector, _f32p), vector.size, _ptr(key, _u8p), None, 0,
                                    ctypes.byref(n)))
    val = np.zeros(n.value, np.uint8)
    _check(lib.vs_arroy_encode_item(index, item, _ptr(vector, _f32p), vector.size, _ptr(key, _u8p),
                                    _ptr(val, _u8p), val.size, ctypes.byref(n)))
    return key.tobytes(), val.tobytes()


def arroy_parse_key(key):
    keep, kp = _bytes_ptr(key)
    index, mode, item = ctypes.c_uint16(0), ctypes.c_uint8(0), ctypes.c_uint32(0)
    _check(lib.vs_arroy_parse_key(kp, len(key), ctypes.byref(index), ctypes.byref(mode), ctypes.byref(item)))
    return index.value, mode.value, item.value


# ---- candidate bitmap codecs (host only, no GPU needed) --------------------------------


def roaring_to_mask(data, n_words, codec="roaring"):
    fn = {"roaring": lib.vs_roaring_to_mask, "cbo": lib.vs_cbo_roaring_to_mask,
          "bo": lib.vs_bo_roaring_to_mask}[codec]
    keep, bp = _bytes_ptr(data)
    words = np.zeros(max(n_words, 1), np.uint64)
    card = ctypes.c_uint64(0)
    _check(fn(bp, len(data), _ptr(words, _u64p), n_words, ctypes.byref(card)))
    return words[:n_words].copy(), card.value


def mask_to_roaring(words, codec="roaring"):
    fn = {"roaring": lib.vs_mask_to_roaring, "cbo": lib.vs_mask_to_cbo_roaring}[codec]
    words = np.ascontiguousarray(words, np.uint64)
    n = ctypes.c_size_t(0)
    _check(fn(_ptr(words, _u64p), words.size, None, 0, ctypes.byref(n)))
    out = np.zeros(max(n.value, 1), np.uint8)
    _check(fn(_ptr(words, _u64p), words.size, _ptr(out, _u8p), out.size, ctypes.byref(n)))
    return out[: n.value].tobytes()
