"""Concurrent requests through every host-buffer entry point at once.

The reference serves searches from many threads, each under its own read transaction
(src/search/ranking/vector_sort.rs:55-78 runs inside whatever thread executes the request). Behind
the C ABI every call takes a workspace of its own from the context's pool (stream, device scratch,
pinned staging that grows on demand), so calls may overlap freely; what they return must be exactly
what the same calls return one at a time."""
import threading

import numpy as np
import pytest

from conftest import facet_mask, synthetic_vectors

pytestmark = pytest.mark.gpu


def test_mixed_entry_points_from_eight_threads(gpu_ctx):
    from search_engine_amd import abi
    category, n, dim = 33, 30_000, 96
    stores = []
    try:
        for j in range(3):
            have = np.arange(0, n, j + 1, dtype=np.uint32)
            st = gpu_ctx.create_store((category << 8) | j, dim, n)
            st.upload(have, synthetic_vectors(have.size, dim, 900 + j))
            st.seal()
            stores.append(st)
        st0 = stores[0]
        queries = synthetic_vectors(24, dim, 910)
        masks = [facet_mask(n, p) for p in (1, 10, 50)]
        roaring = [abi.mask_to_roaring(m) for m in masks]
        handle = st0.create_mask(masks[1])

        def call(kind, i):
            q = queries[i % len(queries)]
            m = i % 3
            if kind == 0:
                return st0.search(q, 10 + i % 40, masks[m])
            if kind == 1:
                return st0.search_roaring(q, 10, roaring[m])
            if kind == 2:
                return st0.search_mask(q, 150, handle)
            if kind == 3:
                return st0.search_batch(queries[i % 8: i % 8 + 5], 10, masks[m])
            if kind == 4:
                return st0.search_batch_masks(queries[:6], 20, [masks[0], None, handle, masks[2], masks[1], None])
            if kind == 5:
                return gpu_ctx.search_stores(category << 8, q, 10, mask=masks[m])[:3]
            h = st0.create_mask(masks[m])  # masks come and go while others search
            try:
                return st0.search_mask(q, 10, h)
            finally:
                st0.destroy_mask(h)

        def flat(r):
            if isinstance(r, list):
                return [flat(x) for x in r]
            return [np.asarray(a).tobytes() for a in r]

        plan = [(kind, i) for i in range(40) for kind in range(7)]
        want = {p: flat(call(*p)) for p in plan}
        errors = []

        def worker(t):
            order = np.random.default_rng(t).permutation(len(plan))
            try:
                for idx in order:
                    p = plan[idx]
                    if flat(call(*p)) != want[p]:
                        errors.append((t, p))
            except Exception as exc:  # noqa: BLE001
                errors.append((t, repr(exc)))

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errors, errors[:5]
        st0.destroy_mask(handle)
    finally:
        for st in stores:
            st.destroy()


def test_searching_threads_beside_an_indexing_thread_through_the_host_mirror():
    """search_engine_amd/stress_mirror (C++, built with the library): six threads in
    execute_vector_search under ONE shared universe object while an indexing thread adds documents
    with ever larger docids through write_vector_points and the build loop -- the stores grow and
    are sealed under the searchers, and the universe's device copy is replaced while others still
    search with it. Every result must be well-formed (no duplicate, nothing outside the universe,
    nothing that was not written yet), nobody may throw, and after the last build the planted
    document is the first hit."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "search_engine_amd", "stress_mirror")
    if not os.path.exists(exe):
        pytest.skip("search_engine_amd/stress_mirror is not built (make -C search_engine_amd/csrc)")
    out = subprocess.run([exe, "6", "40"], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and out.stdout.startswith("ok"), (out.stdout, out.stderr[-2000:])
