"""Candidate-bitmap codecs of the library (host code, no GPU) against the pure-Python
restatement of the formats, plus a replay of the reference's own codec tests
(src/heed_codec/roaring_bitmap/cbo_roaring_bitmap_codec.rs:167-228)."""
import struct

import numpy as np
import pytest

from oracle import roaring as fmt
from oracle.scan import bool_to_mask, mask_to_bool


def ids_of(words):
    return np.nonzero(mask_to_bool(words, words.size * 64))[0].tolist()


def mask_of(ids, n_words):
    b = np.zeros(n_words * 64, bool)
    b[list(ids)] = True
    return bool_to_mask(b)


def random_sets():
    rng = np.random.default_rng(3)
    yield []
    yield [0]
    yield [65535, 65536, 131071]
    yield list(range(0, 70000, 3))                      # > 4096 per chunk: bitmap containers
    yield list(range(5000))                              # one long run
    yield sorted(rng.choice(300000, 5000, replace=False).tolist())   # array containers, 5 chunks
    yield sorted(set(range(100000, 165536)) | {7, 70000 * 4})         # a full chunk
    yield sorted(rng.choice(200000, 150000, replace=False).tolist())  # dense bitmaps


@pytest.mark.parametrize("ids", list(random_sets()), ids=lambda s: f"n{len(s)}")
def test_roaring_round_trip_against_format_oracle(ids):
    from search_engine_amd import abi
    n_words = 5000
    words = mask_of(ids, n_words)
    # library serialisation == format oracle's bytes (run-free form), and both parse back
    blob = abi.mask_to_roaring(words)
    assert blob == fmt.serialize(ids)
    assert fmt.deserialize(blob) == sorted(ids)
    back, card = abi.roaring_to_mask(blob, n_words)
    assert card == len(ids) and np.array_equal(back, words)
    # the run-container form other roaring implementations write must parse too
    runs = fmt.serialize(ids, with_runs=True)
    assert fmt.deserialize(runs) == sorted(ids)
    back, card = abi.roaring_to_mask(runs, n_words)
    assert card == len(ids) and np.array_equal(back, words)


def test_values_beyond_the_mask_are_dropped():
    from search_engine_amd import abi
    ids = [1, 63, 64, 1000, 70000]
    back, card = abi.roaring_to_mask(fmt.serialize(ids), 2)  # 128 docids
    assert ids_of(back) == [1, 63, 64] and card == 3


@pytest.mark.parametrize("blob", [b"", b"\x00" * 8, struct.pack("<II", 12346, 5),
                                  struct.pack("<IIHH", 12346, 1, 0, 9),  # array cut short
                                  struct.pack("<IIHHHH", 12346, 2, 5, 0, 5, 0) + b"\0" * 12])  # keys not ascending
def test_malformed_roaring_is_rejected(blob):
    from search_engine_amd import abi
    with pytest.raises(abi.VsError) as e:
        abi.roaring_to_mask(blob, 4)
    assert e.value.code in (abi.VS_E_FORMAT, abi.VS_E_INVALID_ARG)


# ---- replay of the reference's CboRoaringBitmapCodec tests ----------------------------------


def test_cbo_verify_encoding_decoding():
    """verify_encoding_decoding: RoaringBitmap::from_iter(0..THRESHOLD) round-trips."""
    from search_engine_amd import abi
    ids = list(range(fmt.CBO_THRESHOLD))
    blob = abi.mask_to_roaring(mask_of(ids, 1), codec="cbo")
    assert blob == fmt.cbo_serialize(ids)
    back, _ = abi.roaring_to_mask(blob, 1, codec="cbo")
    assert ids_of(back) == ids


def test_cbo_verify_threshold():
    """verify_threshold: at THRESHOLD items the raw u32 list is smaller than the roaring form."""
    from search_engine_amd import abi
    words = mask_of(range(fmt.CBO_THRESHOLD), 1)
    roaring_size = len(abi.mask_to_roaring(words))
    bo_size = len(abi.mask_to_roaring(words, codec="cbo"))
    assert bo_size == 4 * fmt.CBO_THRESHOLD and roaring_size > bo_size
    # one more item and the codec switches to the roaring form
    words8 = mask_of(range(fmt.CBO_THRESHOLD + 1), 1)
    assert abi.mask_to_roaring(words8, codec="cbo") == abi.mask_to_roaring(words8)


def test_cbo_merge_scenarios_decode():
    """The bitmaps that the reference's merge_cbo_roaring_bitmaps test feeds in and expects out
    (small ones as raw integer lists, the 0..23 result in roaring form) encode and decode to the same
    sets here. Only the codec's two forms are checked: CboRoaringBitmapCodec::merge_into itself
    (cbo_roaring_bitmap_codec.rs:60-...) is off the search path and is not restated."""
    from search_engine_amd import abi
    for ids in ([1, 2, 3], [2, 3, 4], [4, 5], [1, 2], list(range(7, 23)), list(range(23))):
        blob = fmt.cbo_serialize(ids)
        assert (len(blob) <= 28) == (len(ids) <= 7)
        back, card = abi.roaring_to_mask(blob, 2, codec="cbo")
        assert ids_of(back) == ids and card == len(ids)
        assert fmt.cbo_deserialize(abi.mask_to_roaring(mask_of(ids, 2), codec="cbo")) == ids


def test_cbo_ignores_a_trailing_partial_integer_and_bo_rejects_it():
    from search_engine_amd import abi
    blob = struct.pack("=II", 5, 9) + b"\x01\x02"
    back, _ = abi.roaring_to_mask(blob, 1, codec="cbo")  # while let Ok(..) = read_u32
    assert ids_of(back) == [5, 9]
    with pytest.raises(abi.VsError) as e:
        abi.roaring_to_mask(blob, 1, codec="bo")         # chunk.try_into()? fails
    assert e.value.code == abi.VS_E_FORMAT
    back, card = abi.roaring_to_mask(fmt.bo_serialize([3, 64, 100]), 2, codec="bo")
    assert ids_of(back) == [3, 64, 100] and card == 3
