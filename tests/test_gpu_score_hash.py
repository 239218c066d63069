"""The row-score hash (zh_set_hash_mode 2; zebra_amd/csrc/zh_score.hip): every sign of a small-leaf forest from N row scores per
query instead of one dot product per plane, the signs inside the rounding bound recomputed with point_is_above's own arithmetic
(lsh.rs:39-43).  Bucket membership, ids and keys must stay bit-identical to the oracle and to the per-plane hash."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import zebra_oracle as zo  # noqa: E402  (the checker)


@pytest.fixture(scope="module")
def za():
    import zebra_amd
    return zebra_amd


@pytest.mark.parametrize("n,d,M,T,k,B,kind", [
    (30000, 384, 5, 6, 10, 32, 0),    # the reference's default leaf size
    (20000, 64, 3, 4, 10, 16, 2),     # clustered rows: planes between near-identical rows, small margins
    (8000, 768, 8, 3, 40, 8, 0),
    (15000, 128, 5, 5, 10, 24, 1),    # integer-valued rows: many exact ties (w.x + c == 0) -> the exact path decides
    (6000, 100, 4, 3, 5, 12, 0),      # d not a multiple of 4: the score GEMM's scalar staging
    (12000, 384, 5, 4, 10, 7, 0),     # a batch that is not a multiple of four: padded with zero queries
    (12000, 64, 5, 4, 10, 1, 0),      # a single query
])
def test_score_hash_equals_oracle_and_dense_hash(za, n, d, M, T, k, B, kind):
    X = zo.synth_rows(n, d, kind=kind)
    Q = zo.synth_queries(B, d, n, kind=kind)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    f = zo.Forest.from_arrays(X, M, ix.get_forest())
    ix.set_dense_levels(100)  # every sign precomputed, as the library chooses by itself for wandering walks
    m, om = za.L2SquaredDistance(), zo.L2SQ
    oi, ok, oc = f.search_batch(Q, k, om, 0)
    n_planes = ix.get_forest()["consts"].size
    for mode in ("scores", "dense"):
        ix.set_hash_mode(mode)
        ids, keys, counts = ix.search_batch(Q, k, m)
        st = ix.stats()
        assert st["hash_from_scores"] == (1 if mode == "scores" else 0), mode
        if mode == "scores":  # the bound is neither vacuous nor empty
            assert 0 < st["hash_exact_fixups"] < 0.2 * B * n_planes, (st["hash_exact_fixups"], B * n_planes)
        assert (counts == oc).all(), mode
        for b in range(B):
            c = int(oc[b])
            assert (ids[b, :c] == oi[b, :c]).all() and (keys[b, :c] == ok[b, :c]).all(), (mode, b)
    ix.close()


def test_score_hash_follows_inserts_and_is_refused_for_injected_forests(za):
    n0, n1, d, M, T, k, B = 12000, 3000, 256, 5, 4, 10, 16
    X = zo.synth_rows(n0 + n1, d)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.set_hash_mode("scores")
    ix.add(X[:n0])
    ix.add(X[n0:])  # incremental inserts: leaves split, planes (and their sample rows) are appended
    f = zo.Forest.build(X[:n0], M, T)
    f.insert(X, n0)
    assert zo.canonical_forest(ix.get_forest(), d) == zo.canonical_forest(f.arrays(), d)
    Q = zo.synth_queries(B, d, n0 + n1)
    ix.set_dense_levels(-1)
    ids, keys, counts = ix.search_batch(Q, k, za.L2Distance())
    assert ix.stats()["hash_from_scores"] == 1  # small leaves: the library hashes every plane, from scores
    oi, ok, oc = f.search_batch(Q, k, zo.L2, 0)
    assert (counts == oc).all()
    for b in range(B):
        assert (ids[b, :oc[b]] == oi[b, :oc[b]]).all() and (keys[b, :oc[b]] == ok[b, :oc[b]]).all()
    # an injected forest has arbitrary planes: no sample rows, the per-plane hash serves it
    ix2 = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix2.append(X)
    ix2.set_forest(f.arrays())
    ix2.set_hash_mode("scores")
    ix2.set_dense_levels(100)
    i2, k2, c2 = ix2.search_batch(Q, k, za.L2Distance())
    assert ix2.stats()["hash_from_scores"] == 0
    assert (c2 == counts).all() and (i2 == ids).all() and (k2 == keys).all()
    ix.close()
    ix2.close()


def test_score_hash_on_adversarial_rows(za):
    """duplicates (w = 0: every point is 'above', the score difference is exactly 0), huge and tiny magnitudes, rows that
    differ in one element: whatever the bound cannot decide goes to the exact path, so results equal the per-plane hash"""
    rng = np.random.default_rng(5)
    n, d, M, T, k, B = 9000, 128, 4, 4, 10, 16
    X = zo.synth_rows(n, d)
    X[1000:1400] = X[0]                                   # 400 copies of one row
    X[2000:2200] *= np.float32(1e18)                      # |r|^2 overflows f32: inf norms -> nothing is "certain"
    X[3000:3200] *= np.float32(1e-30)                     # subnormal squares
    X[4000:4300] = X[4000] + (rng.random((300, d)) < 0.01).astype(np.float32) * np.float32(1e-3)  # near-duplicates
    Q = np.concatenate([zo.synth_queries(B - 4, d, n), X[[0, 2000, 3000, 4000]]])
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    ix.set_dense_levels(100)
    f = zo.Forest.from_arrays(X, M, ix.get_forest())
    oi, ok, oc = f.search_batch(Q, k, zo.L2SQ, 0)
    for mode in ("scores", "dense"):
        ix.set_hash_mode(mode)
        ids, keys, counts = ix.search_batch(Q, k, za.L2SquaredDistance())
        assert ix.stats()["hash_from_scores"] == (1 if mode == "scores" else 0)
        assert (counts == oc).all(), mode
        for b in range(B):
            assert (ids[b, :oc[b]] == oi[b, :oc[b]]).all() and (keys[b, :oc[b]] == ok[b, :oc[b]]).all(), (mode, b)
    ix.close()


@pytest.mark.parametrize("n,d,M,T,B,kind,scale", [
    (60000, 384, 5, 8, 64, 0, 1.0),     # ~210k planes x 64 queries = 13M signs
    (40000, 768, 4, 6, 32, 0, 1.0),
    (50000, 128, 5, 8, 64, 1, 1.0),     # integer rows: exact zeros of w.x + c
    (50000, 64, 3, 8, 128, 2, 1.0),     # clustered rows
    (30000, 256, 5, 6, 64, 0, 1e-3),    # small magnitudes
    (30000, 256, 5, 6, 64, 0, 3e4),     # large magnitudes
    (30000, 384, 5, 6, 13, 0, 1.0),     # odd batch
])
def test_whole_sign_matrix_equals_the_per_plane_hash(za, n, d, M, T, B, kind, scale):
    """every sign of every plane for every query, row-score path vs one dot product per plane (zh_hash_signs honours
    zh_set_hash_mode): identical bit for bit -- this is the check of the rounding bound at scale, visited or not"""
    X = (zo.synth_rows(n, d, kind=kind) * np.float32(scale)).astype(np.float32)
    Q = (zo.synth_queries(B, d, n, kind=kind) * np.float32(scale)).astype(np.float32)
    Q[:4] = X[[0, 1, n // 2, n - 1]]  # queries that ARE stored rows: sample points of many planes
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    ix.set_hash_mode("dense")
    bits_dense = ix.hash_signs(Q)
    ix.set_hash_mode("scores")
    bits_score = ix.hash_signs(Q)
    st = ix.stats()
    assert st["hash_from_scores"] == 1 and 0 < st["hash_exact_fixups"] < 0.25 * bits_dense.size, st["hash_exact_fixups"]
    assert bits_dense.shape == bits_score.shape and bits_dense.size > 0
    assert (bits_dense == bits_score).all(), int((bits_dense != bits_score).sum())
    ix.close()


def test_clear_then_refill_to_the_same_count_recomputes_the_row_norms(za):
    """zh_index_clear left the row-score hash's |r|^2/2 and |r| of the OLD rows behind; a refill to the same row
    count found them 'valid' (the cache was keyed on the count alone) and derived wrong signs without any error.  The whole sign
    matrix of the refilled index must equal the per-plane hash and the oracle's search."""
    n, d, M, T, k, B = 20000, 256, 5, 5, 10, 32
    X1 = zo.synth_rows(n, d)
    X2 = (zo.synth_rows(n, d, seed=0x1234567) * np.float32(3.0)).astype(np.float32)  # same count, other rows, other norms
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.set_hash_mode("scores")
    ix.add(X1)
    ix.set_dense_levels(100)
    ix.search_batch(zo.synth_queries(B, d, n), k, za.L2SquaredDistance())   # takes the norms of X1
    assert ix.stats()["hash_from_scores"] == 1
    ix.clear()
    assert len(ix) == 0
    ix.add(X2)
    Q = (zo.synth_queries(B, d, n) * np.float32(3.0)).astype(np.float32)
    bits_score = ix.hash_signs(Q)
    assert ix.stats()["hash_from_scores"] == 1
    ix.set_hash_mode("dense")
    bits_dense = ix.hash_signs(Q)
    assert (bits_dense == bits_score).all(), int((bits_dense != bits_score).sum())
    ix.set_hash_mode("scores")
    f = zo.Forest.from_arrays(X2, M, ix.get_forest())
    ids, keys, counts = ix.search_batch(Q, k, za.L2SquaredDistance())
    oi, ok, oc = f.search_batch(Q, k, zo.L2SQ, 0)
    assert (counts == oc).all()
    for b in range(B):
        assert (ids[b, :oc[b]] == oi[b, :oc[b]]).all() and (keys[b, :oc[b]] == ok[b, :oc[b]]).all(), b
    ix.close()
