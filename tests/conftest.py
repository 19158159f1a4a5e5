import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def synthetic_vectors(n, dim, seed):
    """i.i.d. N(0,1) f32 rows, not normalised."""
    return np.random.default_rng(seed).standard_normal((n, dim), dtype=np.float32)


def hash32(x):
    """Avalanche hash of u32 docids; the synthetic facet is hash32(docid) % 100."""
    x = np.asarray(x, np.uint64)
    x = (x ^ (x >> np.uint64(16))) * np.uint64(0x7FEB352D) & np.uint64(0xFFFFFFFF)
    x = (x ^ (x >> np.uint64(15))) * np.uint64(0x846CA68B) & np.uint64(0xFFFFFFFF)
    return (x ^ (x >> np.uint64(16))).astype(np.uint32)


def facet_mask(n_docids, pass_percent):
    """Dense u64 candidate mask: docid passes iff hash32(docid) % 100 < pass_percent."""
    from oracle.scan import bool_to_mask
    return bool_to_mask(hash32(np.arange(n_docids)) % 100 < pass_percent)


@pytest.fixture(scope="session")
def gpu_ctx():
    from search_engine_amd import abi
    ctx = abi.Context(0)
    yield ctx
    ctx.close()
