"""the values of the reference's on-disk partitions (lsh.rs:91-105).  Host-only codec, so
these run without a GPU.  FORMAT UNVERIFIED against the Rust crates (none is available here): the known-answer bytes
below are assembled by hand from the published bincode-legacy / serde encodings of the reference's types
(lsh.rs:16-25,46-60; lib.rs:15-18) and pin this build's reading of them."""
import struct
import uuid

import numpy as np
import pytest

from oracle import zebra_oracle as zo
from zebra_amd import refformat as rf


def _uuids(n, seed=7):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, 16), dtype=np.uint8)


def _leaf(ids):
    return struct.pack("<IQ", 1, len(ids)) + b"".join(struct.pack("<Q", 16) + bytes(i) for i in ids)


def _inner(w, c, left, right):
    return struct.pack("<I", 0) + np.asarray(w, "<f4").tobytes() + struct.pack("<f", c) + left + right


def test_known_answer_bytes_of_a_two_level_tree():
    """Inner(plane A, left = Leaf[u0, u2], right = Inner(plane B, left = Leaf[], right = Leaf[u1])) at N = 3."""
    u = _uuids(3)
    blob = _inner([1.0, -2.0, 0.5], 0.25, _leaf([u[0], u[2]]), _inner([0.0, 3.0, 4.0], -1.5, _leaf([]), _leaf([u[1]])))
    assert len(blob) == 2 * (4 + 12 + 4) + 3 * (4 + 8) + 3 * 24
    f, unknown = rf.decode_trees([blob], 3, u)
    assert unknown == 0
    assert f["roots"].tolist() == [0]
    assert f["plane"].tolist() == [0, -1, 1, -1, -1]                # pre-order: A, leaf, B, leaf, leaf
    assert f["left"][0] == 1 and f["right"][0] == 2 and f["left"][2] == 3 and f["right"][2] == 4
    assert (f["planes"] == np.array([[1, -2, 0.5], [0, 3, 4]], np.float32)).all() and f["consts"].tolist() == [0.25, -1.5]
    assert (f["left"][1], f["right"][1]) == (0, 2) and f["right"][3] == 0 and f["right"][4] == 1
    assert f["leaf_ids"].tolist() == [0, 2, 1]
    assert rf.encode_trees(f, 3, u) == [blob]


def test_uuid_bytes_are_the_uuid_crates_as_bytes_order():
    """Uuid::as_bytes (lsh.rs:92,101) is the big-endian field order = Python's UUID.bytes."""
    ids = [uuid.UUID(int=(i + 1) * 0x0123456789ABCDEF0123456789ABCDEF % (1 << 128)) for i in range(4)]
    table = [x.bytes for x in ids]
    blob = _leaf([ids[3].bytes, ids[0].bytes])
    f, _ = rf.decode_trees([blob], 8, table)
    assert f["leaf_ids"].tolist() == [3, 0] and f["plane"].tolist() == [-1]


def test_embedding_values_are_raw_le_f32():
    X = zo.synth_rows(5, 12)
    vals = rf.encode_embeddings(X)
    assert all(len(v) == 48 for v in vals) and vals[2] == X[2].astype("<f4").tobytes()
    assert (rf.decode_embeddings(vals, 12) == X).all()
    with pytest.raises(ValueError):
        rf.decode_embeddings([b"\0" * 47], 12)


@pytest.mark.parametrize("n,d,M,T", [(400, 8, 5, 3), (3000, 64, 40, 4), (64, 384, 5, 15)])
def test_round_trip_of_oracle_forests(n, d, M, T):
    X = zo.synth_rows(n, d)
    f = zo.Forest.build(X, M, T)
    g = f.arrays()
    u = _uuids(n)
    blobs = rf.encode_trees(g, d, u)
    assert len(blobs) == T
    n_inner = int((g["plane"] >= 0).sum())
    assert sum(map(len, blobs)) == n_inner * (4 + 4 * d + 4) + (len(g["plane"]) - n_inner) * 12 + 24 * len(g["leaf_ids"])
    back, unknown = rf.decode_trees(blobs, d, u)
    assert unknown == 0
    assert zo.canonical_forest(back, d) == zo.canonical_forest(g, d)
    # the same forest with the rows stored in another order (fjall iterates by key, not by insertion)
    perm = np.random.default_rng(3).permutation(n)
    back2, _ = rf.decode_trees(blobs, d, u[perm])
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    assert (np.sort(back2["leaf_ids"]) == np.sort(inv[back["leaf_ids"]])).all()
    f2 = zo.Forest.from_arrays(X[perm], M, back2)
    Q = zo.synth_queries(5, d, n)
    for b in range(5):
        i1, k1 = f.search(Q[b], 7, zo.L2SQ)
        i2, k2 = f2.search(Q[b], 7, zo.L2SQ)
        assert (k1 == k2).all() and (perm[i2.astype(np.int64)] == i1.astype(np.int64)).all()


def test_removed_vectors_and_malformed_values():
    u = _uuids(4)
    blob = _inner([1.0, 1.0], 0.0, _leaf([u[0], u[1]]), _leaf([u[2], u[3]]))
    f, unknown = rf.decode_trees([blob], 2, u[[0, 3]])      # u1, u2 were removed: the reference's trees keep them
    assert unknown == 2 and f["leaf_ids"].tolist() == [0, 1] and f["right"].tolist()[1:] == [1, 1]
    from zebra_amd import ZhError
    for bad in (blob[:-1], blob + b"\0", struct.pack("<I", 2), blob[:30], struct.pack("<IQ", 1, 1 << 40)):
        with pytest.raises(ZhError):
            rf.decode_trees([bad], 2, u)
    with pytest.raises(ZhError):
        rf.decode_trees([blob], 2, np.concatenate([u, u[:1]]))   # duplicate key
    with pytest.raises(ZhError):
        rf.encode_trees(f, 2, u[:1])                               # a leaf row without a uuid
    # 70 nested Inner nodes: deeper than any tree the index accepts
    deep = _leaf([])
    for _ in range(70):
        deep = _inner([0.0, 0.0], 0.0, deep, _leaf([]))
    with pytest.raises(ZhError):
        rf.decode_trees([deep], 2, u)


# ---- random trees (hypothesis): a reference-side encoder written here from the format description, independent of
# ---- zh_ref_tree_encode, against the decoder; and decoder(encoder(x)) == x for arbitrary shapes
from hypothesis import given, settings  # noqa: E402
from hypothesis import strategies as st  # noqa: E402


def _random_tree(rng, d, n_rows, depth):
    """nested ('L', ids) / ('I', w, c, left, right) with random shape; ids drawn without replacement per tree"""
    pool = list(rng.permutation(n_rows))

    def rec(level):
        if level >= depth or rng.random() < 0.3 or not pool:
            k = int(rng.integers(0, min(4, len(pool)) + 1))
            return ("L", [int(pool.pop()) for _ in range(k)])
        return ("I", rng.standard_normal(d).astype(np.float32), np.float32(rng.standard_normal()), rec(level + 1), rec(level + 1))
    return rec(0)


def _encode_py(t, u):
    if t[0] == "L":
        return _leaf([u[i] for i in t[1]])
    return _inner(t[1], float(t[2]), _encode_py(t[3], u), _encode_py(t[4], u))


def _leaves_py(t):
    return [t[1]] if t[0] == "L" else _leaves_py(t[3]) + _leaves_py(t[4])


@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), d=st.sampled_from([1, 3, 8, 33]), T=st.integers(1, 4), depth=st.integers(0, 9))
def test_random_trees_decode_and_reencode(seed, d, T, depth):
    rng = np.random.default_rng(seed)
    n = 64
    u = _uuids(n, seed=seed % 1000)
    trees = [_random_tree(rng, d, n, depth) for _ in range(T)]
    blobs = [_encode_py(t, u) for t in trees]
    f, unknown = rf.decode_trees(blobs, d, u)
    assert unknown == 0 and len(f["roots"]) == T
    # leaves in pre-order carry the same ids in the same order
    got = []
    for t in range(T):
        stack, out = [int(f["roots"][t])], []
        while stack:
            i = stack.pop()
            if f["plane"][i] < 0:
                out.append(f["leaf_ids"][f["left"][i]: f["left"][i] + f["right"][i]].tolist())
            else:
                stack.append(int(f["right"][i]))
                stack.append(int(f["left"][i]))
        got.append(out)
    assert got == [_leaves_py(t) for t in trees]
    assert rf.encode_trees(f, d, u) == blobs


def test_known_answer_bytes_of_the_database_header():
    """The `.zebra` file (core.rs:19-29 DatabaseInner, save_database core.rs:183-190), assembled by hand: the uuid as a byte
    string (u64 16 + 16 bytes), nothing for the unit-struct model (model/text.rs:11) and the unit-struct metric
    (distance.rs:15-17), LSHIndexOptions { max_node_size: usize, num_trees: usize } as two u64 (lsh.rs:124-129)."""
    u = uuid.UUID("0192f0c5-7b1e-7cc3-9d4a-2f1e0a3b5c7d")
    kat = struct.pack("<Q", 16) + u.bytes + struct.pack("<QQ", 5, 15)          # the defaults of lsh.rs:131-138
    assert len(kat) == 40
    h = rf.decode_header(kat, "cosine")
    assert h == dict(uuid=u.bytes, max_node_size=5, num_trees=15, metric=0, power=0, model=b"")
    assert rf.encode_header(u.bytes) == kat
    # MinkowskiDistance { power: i32 } (distance.rs:160-165) sits between the model and the options
    kat_p = struct.pack("<Q", 16) + u.bytes + struct.pack("<i", 3) + struct.pack("<QQ", 4096, 15)
    h = rf.decode_header(kat_p, "minkowski")
    assert (h["power"], h["max_node_size"], h["num_trees"]) == (3, 4096, 15)
    assert rf.encode_header(u.bytes, 4096, 15, "minkowski", 3) == kat_p
    assert rf.encode_header(u.bytes, 4096, 15, "pnorm", 3) == kat_p              # same layout, another type parameter
    # a model that carries bytes (none of the reference's does): opaque, passed through
    kat_m = struct.pack("<Q", 16) + u.bytes + b"\x01\x02\x03" + struct.pack("<QQ", 5, 15)
    assert rf.decode_header(kat_m, "l2", model_len=3)["model"] == b"\x01\x02\x03"
    assert rf.encode_header(u.bytes, 5, 15, "l2", model=b"\x01\x02\x03") == kat_m
    # stating the wrong metric type is detected by the length
    for bad, metric in ((kat, "minkowski"), (kat_p, "cosine"), (kat[:-1], "cosine"), (kat + b"\0", "cosine"),
                        (struct.pack("<Q", 15) + kat[8:], "cosine")):
        with pytest.raises(Exception) as e:
            rf.decode_header(bad, metric)
        assert getattr(e.value, "code", None) == -1
