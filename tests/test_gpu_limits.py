"""Limits of one launch are not limits of one call: a batch that would pass the per-batch caps (2^26 pairs, 2^28 - 1
leaf visits, 2^36 scored rows -- reachable with the reference's default options, where every query visits ~10^4..10^5
leaves) is split inside the blocking entry points; any vector length works (the build's hyperplane kernel
tiles the vector through LDS); an add that cannot fit is refused before anything changes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import zebra_oracle as zo  # noqa: E402


@pytest.fixture(scope="module")
def za():
    import zebra_amd
    return zebra_amd


def test_batch_beyond_the_visit_cap_is_split_not_refused(za):
    """reference defaults (max_node_size 5, 15 trees) on 1M rows: ~68k leaf visits per query -> 4200 queries are ~2.9e8
    visits, more than one batch may hold (2^28 - 1).  zh_search_batch answers all of them, identically to the same
    queries asked in small batches, and to the oracle."""
    n, d, B, k = 1_000_000, 64, 4200, 10
    ix = za.LSHIndex(d, za.LSHIndexOptions(5, 15), reserve_rows=n)
    ix.append_synthetic(n)
    ix.build()
    Q = zo.synth_queries(B, d, n)
    m = za.L2SquaredDistance()
    for attempt in range(2):  # first call: no visit statistics yet (split by retry); second: sized from the first
        ids, keys, counts = ix.search_batch(Q, k, m)
        assert (counts == k).all()
        st = ix.stats()
        assert st["batch"] < B, "the batch must have been split"
        if attempt == 0:
            first = (ids.copy(), keys.copy())
    assert (ids == first[0]).all() and (keys == first[1]).all()
    for lo in (0, 2000, 4100):
        i2, k2, _ = ix.search_batch(Q[lo:lo + 100], k, m)
        assert (i2 == ids[lo:lo + 100]).all() and (k2 == keys[lo:lo + 100]).all()
    f = zo.Forest.borrow_synth(n, d, 5, ix.get_forest())
    sel = np.array([0, 1, 2099, 2100, 4199])
    oi, ok, oc = f.search_batch_synth(Q[sel], k, zo.L2SQ)
    assert (oc == k).all() and (ids[sel] == oi).all() and (keys[sel] == ok).all()
    ix.close()


@pytest.mark.parametrize("n,d,M,T", [(1500, 4100, 64, 3), (700, 9000, 40, 2), (600, 8194, 50, 2)])
def test_long_vectors_build_and_search(za, n, d, M, T):
    """dimensions beyond the specialised kernels and beyond 8192 (two LDS tiles in make_planes), one not a multiple of 4"""
    X = zo.synth_rows(n, d)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    f = zo.Forest.build(X, M, T)
    assert zo.canonical_forest(ix.get_forest(), d) == zo.canonical_forest(f.arrays(), d)
    Q = zo.synth_queries(6, d, n)
    for m, om, omode in ((za.L2Distance(), zo.L2, 0), (za.CosineDistance(parity=False), zo.COSINE, zo.CORRECTED)):
        ids, keys, counts = ix.search_batch(Q, 10, m)
        oi, ok, oc = f.search_batch(Q, 10, om, omode)
        assert (counts == oc).all() and (ids == oi).all() and (keys == ok).all()
    ix.close()
    with pytest.raises(za.ZhError) as e:
        za.LSHIndex((1 << 20) + 1)
    assert e.value.code == -5  # ZH_ELIMIT


def test_an_add_that_cannot_fit_changes_nothing(za):
    d = 8
    ix = za.LSHIndex(d, za.LSHIndexOptions(16, 4096))  # 4096 trees: T * rows passes 2^32 - 1 at ~1.05M rows
    X = zo.synth_rows(1000, d)
    ix.add(X)
    before = zo.canonical_forest(ix.get_forest(), d)
    big = np.zeros((1_100_000, d), np.float32)
    with pytest.raises(za.ZhError) as e:
        ix.add(big)
    assert e.value.code == -5 and len(ix) == 1000
    assert zo.canonical_forest(ix.get_forest(), d) == before
    ids, _, counts = ix.search_batch(X[:4], 3, za.L2SquaredDistance())  # still searchable
    assert (counts == 3).all() and (ids[:, 0] == np.arange(4)).all()
    ix.close()
