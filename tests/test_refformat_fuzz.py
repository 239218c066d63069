"""Byte-mutation fuzz of the on-disk codec (zebra_amd/csrc/zh_refformat.cpp parses untrusted bytes: the values of the
reference's `trees` partition, lsh.rs:99-105, and the `.zebra` header, core.rs:92-102).  Whatever the bytes are, a decode
either succeeds -- and then re-encodes to a blob that decodes to the same forest -- or fails with a ZhError: never a crash,
an out-of-bounds access or an endless loop.  tests/test_sanitizers.py runs this file again on the codec built with
-fsanitize=address,undefined, where an out-of-bounds read is a hard failure, not luck."""
import struct

import numpy as np
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from zebra_amd import refformat as rf
from zebra_amd._ffi import ZhError

DIM = 3


def _uuids(n, seed=3):
    return np.random.default_rng(seed).integers(0, 256, (n, 16), dtype=np.uint8)


U = _uuids(6)


def _leaf(ids):
    return struct.pack("<IQ", 1, len(ids)) + b"".join(struct.pack("<Q", 16) + bytes(i) for i in ids)


def _inner(w, c, left, right):
    return struct.pack("<I", 0) + np.asarray(w, "<f4").tobytes() + struct.pack("<f", c) + left + right


VALID = [
    _leaf([]),
    _leaf([U[0], U[3], U[5]]),
    _inner([1, 2, 3], 0.5, _leaf([U[1]]), _leaf([U[2], U[4]])),
    _inner([1, -2, 0.5], 0.25, _leaf([U[0], U[2]]), _inner([0, 3, 4], -1.5, _leaf([]), _leaf([U[1]]))),
]

mutation = st.one_of(
    st.tuples(st.just("flip"), st.integers(0, 10**6), st.integers(0, 255)),
    st.tuples(st.just("cut"), st.integers(0, 10**6), st.integers(0, 0)),
    st.tuples(st.just("ins"), st.integers(0, 10**6), st.integers(0, 255)),
    st.tuples(st.just("dup"), st.integers(0, 10**6), st.integers(1, 40)),
    st.tuples(st.just("u64"), st.integers(0, 10**6), st.integers(0, 7)),   # a length field becomes huge
)


def _mutate(blob, muts):
    b = bytearray(blob)
    for kind, pos, val in muts:
        if not b and kind != "ins":
            continue
        i = pos % (len(b) + (1 if kind == "ins" else 0))
        if kind == "flip":
            b[i] = val
        elif kind == "cut":
            del b[i:]
        elif kind == "ins":
            b.insert(i, val)
        elif kind == "dup":
            b[i:i] = b[i:i + val]
        elif kind == "u64":
            b[i:i + 8] = struct.pack("<Q", (1 << (8 * val + 7)) + 1)[:max(0, min(8, len(b) - i))]
    return bytes(b)


def _same_forest(a, b):
    return all(np.array_equal(a[k], b[k]) for k in rf.FOREST_KEYS)


@settings(max_examples=400, deadline=None, suppress_health_check=list(HealthCheck))
@given(st.integers(0, len(VALID) - 1), st.lists(mutation, min_size=0, max_size=4))
def test_mutated_tree_values_decode_or_fail_cleanly(which, muts):
    blob = _mutate(VALID[which], muts)
    try:
        f, unknown = rf.decode_trees([blob], DIM, U)
    except ZhError as e:
        assert e.code in (-1, -5), e  # ZH_EINVAL / ZH_ELIMIT: a diagnosis, not a crash
        return
    # accepted: every id is a known row, every leaf run is in range, and the forest round-trips
    assert unknown >= 0 and (f["leaf_ids"] < len(U)).all()
    for n in range(f["plane"].size):
        if f["plane"][n] < 0:
            assert 0 <= int(np.uint32(f["left"][n])) + int(f["right"][n]) <= f["leaf_ids"].size
        else:
            assert 0 <= f["left"][n] < f["plane"].size and 0 <= f["right"][n] < f["plane"].size
    again = rf.encode_trees(f, DIM, U)
    g, _ = rf.decode_trees(again, DIM, U)
    assert _same_forest(f, g)
    if not muts:
        assert again == [blob] and unknown == 0


@settings(max_examples=300, deadline=None, suppress_health_check=list(HealthCheck))
@given(st.lists(mutation, min_size=0, max_size=3), st.sampled_from(["cosine", "l2", "minkowski", "pnorm"]), st.integers(0, 5))
def test_mutated_headers_decode_or_fail_cleanly(muts, metric, model_len):
    model = bytes(range(model_len))
    blob = _mutate(rf.encode_header(bytes(U[0]), 7, 11, metric, power=3, model=model), muts)
    try:
        h = rf.decode_header(blob, metric, model_len)
    except ZhError as e:
        assert e.code == -1, e
        return
    assert rf.encode_header(h["uuid"], h["max_node_size"], h["num_trees"], metric, h["power"], h["model"]) == blob
    if not muts:
        assert (h["uuid"], h["max_node_size"], h["num_trees"], h["model"]) == (bytes(U[0]), 7, 11, model)
        assert h["power"] == (3 if metric in ("minkowski", "pnorm") else 0)
