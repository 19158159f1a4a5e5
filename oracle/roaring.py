"""Pure-Python restatement of the candidate-bitmap wire formats.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The portable RoaringBitmap serialisation is the format of the third-party crate
`roaring = "0.10.2"` (reference Cargo.toml:45), which the reference writes with
`RoaringBitmap::serialize_into` (src/heed_codec/roaring_bitmap/
roaring_bitmap_codec.rs:29-33). The crate is not in the reference tree and no
roaring implementation is installed here, so this restates the published format
specification (RoaringFormatSpec) — PARITY UNPINNED for the byte layout itself;
the reference's own tests pin only the CboRoaringBitmapCodec threshold
(cbo_roaring_bitmap_codec.rs:167-228), which tests/test_roaring_codecs.py replays.

Sets are Python sets / sorted lists of u32 docids.
"""
import struct

SERIAL_COOKIE_NO_RUNCONTAINER = 12346
SERIAL_COOKIE = 12347
NO_OFFSET_THRESHOLD = 4
ARRAY_MAX = 4096
CBO_THRESHOLD = 7  # cbo_roaring_bitmap_codec.rs:15


def serialize(ids, with_runs=False):
    """RoaringBitmap::serialize_into. with_runs=False is what roaring-rs 0.10 writes
    (cookie 12346); with_runs=True writes the run-container form other
    implementations produce, choosing a run container when it is smaller."""
    chunks = {}
    for d in sorted(set(ids)):
        chunks.setdefault(d >> 16, []).append(d & 0xFFFF)
    keys = sorted(chunks)
    bodies, is_run = [], []
    for key in keys:
        vals = chunks[key]
        runs = []
        for v in vals:
            if runs and runs[-1][0] + runs[-1][1] + 1 == v:
                runs[-1][1] += 1
            else:
                runs.append([v, 0])
        plain = 8192 if len(vals) > ARRAY_MAX else 2 * len(vals)
        if with_runs and 2 + 4 * len(runs) < plain:
            body = struct.pack("<H", len(runs)) + b"".join(struct.pack("<HH", s, l) for s, l in runs)
            is_run.append(True)
        elif len(vals) > ARRAY_MAX:
            words = [0] * 1024
            for v in vals:
                words[v >> 6] |= 1 << (v & 63)
            body = struct.pack("<1024Q", *words)
            is_run.append(False)
        else:
            body = struct.pack(f"<{len(vals)}H", *vals)
            is_run.append(False)
        bodies.append(body)
    n = len(keys)
    has_runs = with_runs and any(is_run)
    if has_runs:
        head = struct.pack("<I", SERIAL_COOKIE | ((n - 1) << 16))
        bits = bytearray((n + 7) // 8)
        for i, r in enumerate(is_run):
            if r:
                bits[i >> 3] |= 1 << (i & 7)
        head += bytes(bits)
    else:
        head = struct.pack("<II", SERIAL_COOKIE_NO_RUNCONTAINER, n)
    descr = b"".join(struct.pack("<HH", key, len(chunks[key]) - 1) for key in keys)
    offsets = b""
    if not has_runs or n >= NO_OFFSET_THRESHOLD:
        at = len(head) + len(descr) + 4 * n
        for body in bodies:
            offsets += struct.pack("<I", at)
            at += len(body)
    return head + descr + offsets + b"".join(bodies)


def deserialize(data):
    """RoaringBitmap::deserialize_from -> sorted list of docids."""
    (cookie,) = struct.unpack_from("<I", data, 0)
    pos = 4
    run_bits = None
    if cookie & 0xFFFF == SERIAL_COOKIE:
        n = (cookie >> 16) + 1
        run_bits = data[pos: pos + (n + 7) // 8]
        pos += (n + 7) // 8
    elif cookie == SERIAL_COOKIE_NO_RUNCONTAINER:
        (n,) = struct.unpack_from("<I", data, pos)
        pos += 4
    else:
        raise ValueError("unknown cookie")
    descr = [struct.unpack_from("<HH", data, pos + 4 * i) for i in range(n)]
    pos += 4 * n
    if run_bits is None or n >= NO_OFFSET_THRESHOLD:
        pos += 4 * n
    out = []
    for i, (key, card_m1) in enumerate(descr):
        base, card = key << 16, card_m1 + 1
        if run_bits is not None and (run_bits[i >> 3] >> (i & 7)) & 1:
            (n_runs,) = struct.unpack_from("<H", data, pos)
            pos += 2
            for _ in range(n_runs):
                s, l = struct.unpack_from("<HH", data, pos)
                pos += 4
                out.extend(range(base + s, base + s + l + 1))
        elif card > ARRAY_MAX:
            words = struct.unpack_from("<1024Q", data, pos)
            pos += 8192
            for wi, w in enumerate(words):
                while w:
                    b = (w & -w).bit_length() - 1
                    out.append(base + wi * 64 + b)
                    w &= w - 1
        else:
            out.extend(base + v for v in struct.unpack_from(f"<{card}H", data, pos))
            pos += 2 * card
    return out


def cbo_serialize(ids):
    """CboRoaringBitmapCodec::serialize_into (cbo_roaring_bitmap_codec.rs:30-42)."""
    ids = sorted(set(ids))
    if len(ids) <= CBO_THRESHOLD:
        return b"".join(struct.pack("=I", d) for d in ids)  # native endian
    return serialize(ids)


def cbo_deserialize(data):
    """CboRoaringBitmapCodec::deserialize_from (cbo_roaring_bitmap_codec.rs:44-58)."""
    if len(data) <= CBO_THRESHOLD * 4:
        return sorted({struct.unpack_from("=I", data, i)[0] for i in range(0, len(data) - 3, 4)})
    return deserialize(data)


def bo_serialize(ids):
    """BoRoaringBitmapCodec::serialize_into (bo_roaring_bitmap_codec.rs:13-16)."""
    return b"".join(struct.pack("=I", d) for d in sorted(set(ids)))


def bo_deserialize(data):
    """BoRoaringBitmapCodec::bytes_decode (bo_roaring_bitmap_codec.rs:22-31)."""
    if len(data) % 4:
        raise ValueError("chunk.try_into() fails on a short chunk")
    return sorted({struct.unpack_from("=I", data, i)[0] for i in range(0, len(data), 4)})
