"""GPU parity tests: the HIP path (through the C ABI, zebra_amd -> libzebra_hip.so) against the CPU
oracle on the same seeded inputs.  Bars (BASELINE.json north_star): hash sign bits, bucket
membership, returned ids and counts BIT-EXACT; distance keys bit-exact as well (the summation order
is fixed on both sides), which is stronger than the 1e-5 relative the north_star asks for."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import zebra_oracle as zo  # noqa: E402  (the checker; tests may use it)


@pytest.fixture(scope="module")
def za():
    import zebra_amd
    return zebra_amd


def metrics(za):
    return [("l2sq", za.L2SquaredDistance(), zo.L2SQ, 0), ("l2", za.L2Distance(), zo.L2, 0),
            ("cos_parity", za.CosineDistance(parity=True), zo.COSINE, zo.PARITY),
            ("cos_corrected", za.CosineDistance(parity=False), zo.COSINE, zo.CORRECTED)]


# --------------------------------------------------------------------------- src/distance.rs
@pytest.mark.parametrize("d", [1, 3, 4, 100, 128, 256, 384, 768, 1000, 1536])
def test_distance_keys_bit_exact(za, d):
    rng = np.random.default_rng(d)
    X = rng.standard_normal((257, d)).astype(np.float32)
    X[3] = 0  # zero-norm row
    q = rng.standard_normal(d).astype(np.float32)
    for name, m, om, omode in metrics(za):
        got = m.distance_batch(X, q)
        want = zo.distance_batch(om, omode, X, q)
        assert (got == want).all(), name
        # 1e-5 relative against float64 numpy (independent of the summation order)
    l2 = ((X.astype(np.float64) - q.astype(np.float64)) ** 2).sum(1)
    np.testing.assert_allclose(za.L2SquaredDistance().distance_batch(X, q).view(np.float64), l2, rtol=1e-5)
    assert za.L2SquaredDistance().distance(X[0], q) == zo.distance(zo.L2SQ, 0, X[0], q)
    # zero query: both zero-norm branches of simsimd's cosine
    z = np.zeros(d, np.float32)
    for name, m, om, omode in metrics(za):
        assert (m.distance_batch(X[:5], z) == zo.distance_batch(om, omode, X[:5], z)).all(), name


def more_metrics(za):
    return [("chebyshev", za.ChebyshevDistance(), zo.CHEBYSHEV, 0), ("canberra", za.CanberraDistance(), zo.CANBERRA, 0),
            ("bray_curtis", za.BrayCurtisDistance(), zo.BRAY_CURTIS, 0), ("manhattan", za.ManhattanDistance(), zo.MANHATTAN, 0),
            ("l3", za.L3Distance(), zo.L3, 0), ("l4", za.L4Distance(), zo.L4, 0), ("hamming", za.HammingDistance(), zo.HAMMING, 0),
            ("minkowski3", za.MinkowskiDistance(3), zo.MINKOWSKI, 3), ("minkowski7", za.MinkowskiDistance(7), zo.MINKOWSKI, 7),
            ("minkowski1", za.MinkowskiDistance(1), zo.MINKOWSKI, 1), ("minkowski2", za.MinkowskiDistance(2), zo.MINKOWSKI, 2),
            ("pnorm2", za.PNormDistance(2), zo.PNORM, 2), ("pnorm5", za.PNormDistance(5), zo.PNORM, 5)]


def any_power_metrics(za):
    """`power` is an i32 and the reference's derived Default is 0 (distance.rs:160-165,176-181; core.rs:115,146)"""
    out = [("minkowski_default", za.MinkowskiDistance(), zo.MINKOWSKI, 0), ("pnorm_default", za.PNormDistance(), zo.PNORM, 0)]
    for p in (-2, -1, -3, 64, 65, 66, 127, 1000, -65, 2**31 - 1, -2**31):
        out += [("minkowski%d" % p, za.MinkowskiDistance(p), zo.MINKOWSKI, p), ("pnorm%d" % p, za.PNormDistance(p), zo.PNORM, p)]
    return out


# ------------------------------------------------------- src/distance.rs:51-98,116-190 (f2)
@pytest.mark.parametrize("d", [3, 100, 128, 384, 768, 1000])
def test_distances_crate_metric_keys_bit_exact(za, d):
    rng = np.random.default_rng(d)
    for scale in (1.0, 1e-6, 1e5):
        X = (rng.standard_normal((130, d)) * scale).astype(np.float32)
        X[3] = 0
        q = (rng.standard_normal(d) * scale).astype(np.float32)
        for name, m, om, p in more_metrics(za):
            got, want = m.distance_batch(X, q), zo.distance_batch(om, p, X, q)
            same = (got == want) | (np.isnan(got.astype(np.uint32).view(np.float32)) & np.isnan(want.astype(np.uint32).view(np.float32)))
            assert same.all(), (name, d, scale, got[~same][:3], want[~same][:3])


def _same_f32_keys(got, want):
    return (got == want) | (np.isnan(got.astype(np.uint32).view(np.float32)) & np.isnan(want.astype(np.uint32).view(np.float32)))


@pytest.mark.parametrize("d", [1, 3, 128, 384, 1000])
def test_power_metrics_any_i32_power_bit_exact(za, d):
    """powers {0 (the reference's Default), negative, past 64, the i32 extremes}: keys bit-equal to the oracle, nothing refused"""
    rng = np.random.default_rng(100 + d)
    for lo, hi in ((0.5, 1.6), (0.3, 0.95), (0.0, 3.0)):
        q = rng.uniform(-1, 1, d).astype(np.float32)
        X = (q + rng.uniform(lo, hi, (70, d)).astype(np.float32) * rng.choice([-1.0, 1.0], (70, d)).astype(np.float32)).astype(np.float32)
        X[3] = q            # every |a - b| = 0: powi(0, negative) = inf
        X[4, 0] = np.nan
        X[5, 0] = np.inf
        for name, m, om, p in any_power_metrics(za):
            got, want = m.distance_batch(X, q), zo.distance_batch(om, p, X, q)
            same = _same_f32_keys(got, want)
            assert same.all(), (name, d, lo, got[~same][:3], want[~same][:3])
    assert za.MinkowskiDistance().power == 0 and za.PNormDistance().power == 0
    assert (za.PNormDistance().distance_batch(X, q).astype(np.uint32).view(np.float32) == d).all()


def test_database_with_the_default_constructed_power_metric_and_top_k_zero(za):
    """Database::new builds the metric with Met::default() (core.rs:115,146) -> MinkowskiDistance { power: 0 }: every key is
    +inf (d > 1), so the (key, id) order decides; LSHIndex::search with top_k = 0 is Ok(vec![]) (lsh.rs:544-565)"""
    n, d, M, T, k, B = 3000, 24, 40, 5, 7, 9
    X = zo.synth_rows(n, d)
    Q = zo.synth_queries(B, d, n)
    f = zo.Forest.build(X, M, T)
    for cls, om in ((za.MinkowskiDistance, zo.MINKOWSKI), (za.PNormDistance, zo.PNORM)):
        db = za.Database(d, cls, za.LSHIndexOptions(M, T))  # the class itself: default-constructed, as Met::default()
        assert db.metric.power == 0
        db.insert_records(X, [b"doc%d" % i for i in range(n)])
        oi, ok, oc = f.search_batch(Q, k, om, 0)
        got = db.query_vectors(Q, k)
        for b in range(B):
            assert sorted(got[b]) == sorted(int(i) for i in oi[b, :oc[b]])
            assert all(got[b][i] == b"doc%d" % i for i in got[b])
        ids, keys, counts = db.index.search_batch(Q, k, db.metric)
        assert (counts == oc).all() and (ids == oi).all() and (keys == ok).all()
        for p in (-2, 65):
            ids, keys, counts = db.index.search_batch(Q, k, cls(p))
            oi, ok, oc = f.search_batch(Q, k, om, p)
            assert (counts == oc).all() and (ids == oi).all() and (keys == ok).all(), p
        # top_k = 0: rc 0, no neighbours, through every entry point
        ids, keys, counts = db.index.search_batch(Q, 0, db.metric)
        assert ids.shape == (B, 0) and keys.shape == (B, 0) and (counts == 0).all()
        assert db.index.search(Q[0], 0, za.L2SquaredDistance()) == []
        assert db.query_vectors(Q, 0) == {b: {} for b in range(B)}
        assert f.search_batch(Q, 0, zo.L2SQ)[2].tolist() == [0] * B
        db.index.close()


@pytest.mark.parametrize("n,d,M,T,k,B", [(6000, 128, 200, 6, 10, 20), (3000, 384, 64, 4, 10, 12), (2500, 50, 40, 3, 7, 9)])
def test_search_with_distances_crate_metrics(za, n, d, M, T, k, B):
    X = zo.synth_rows(n, d)
    Q = zo.synth_queries(B, d, n)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    for name, m, om, p in more_metrics(za):
        ids, keys, counts = ix.search_batch(Q, k, m)
        oi, ok, oc = f.search_batch(Q, k, om, p)
        assert (counts == oc).all() and (ids == oi).all() and (keys == ok).all(), name


def test_distance_special_values(za):
    X = np.array([[np.inf, 1, 2, 3], [np.nan, 0, 0, 0], [1e-30, 1e-30, 0, 0], [3e38, 3e38, 3e38, 3e38],
                  [-0.0, 0, 0, 0], [1, 2, 3, 4]], np.float32)
    q = np.array([1, 2, 3, 4], np.float32)
    for name, m, om, omode in metrics(za):
        got, want = m.distance_batch(X, q), zo.distance_batch(om, omode, X, q)
        gf, wf = got.view(np.float64), want.view(np.float64)
        same = (got == want) | (np.isnan(gf) & np.isnan(wf))
        assert same.all(), (name, got, want)


# ------------------------------------------------------------------ lsh.rs:39-43 (the hash)
@pytest.mark.parametrize("n,d,M,T", [(3000, 32, 64, 4), (2000, 384, 48, 3), (1500, 768, 64, 2), (800, 100, 16, 2)])
def test_hash_signs_and_dots_bit_exact(za, n, d, M, T):
    X = zo.synth_rows(n, d)
    f = zo.Forest.build(X, M, T, seed=9)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.append(X)
    ix.set_forest(f.arrays())
    g = zo.Forest.from_arrays(X, M, ix.get_forest())  # the library renumbers planes level-major
    Q = zo.synth_queries(70, d, n)
    signs, dots = ix.hash_signs(Q, dots=True)
    for b in range(Q.shape[0]):
        s, dt = g.hash_signs(Q[b])
        assert (dots[b].view(np.uint32) == dt.view(np.uint32)).all() or np.array_equal(dots[b], dt)
        assert (signs[b] == s.astype(bool)).all()
    # renumbering did not change the forest
    assert zo.canonical_forest(ix.get_forest(), d) == zo.canonical_forest(f.arrays(), d)


# --------------------------------------------------- lsh.rs:290-348 + 544-565 (walk + search)
CASES = [
    # n, d, M, T, k, batch, kind
    (5000, 64, 5, 15, 10, 24, 0),      # reference defaults: near-exhaustive regime
    (20000, 384, 256, 15, 10, 64, 0),  # one leaf per tree
    (8000, 768, 512, 8, 100, 32, 0),   # k = 100
    (6000, 128, 300, 10, 10, 48, 1),   # SIFT-style integers: exact L2
    (3000, 100, 40, 5, 7, 17, 0),      # d not a multiple of 4 -> generic kernels
    (4000, 256, 12, 6, 10, 9, 0),      # leaves ~ k: backup walks with n - k
    (30000, 64, 9000, 3, 100, 16, 0),  # leaves longer than the select kernel's LDS buffer
    (20000, 32, 20001, 2, 10, 8, 0),   # one 20000-row leaf per tree
]


@pytest.mark.parametrize("n,d,M,T,k,B,kind", CASES)
def test_search_bit_exact_with_injected_forest(za, n, d, M, T, k, B, kind):
    X = zo.synth_rows(n, d, kind=kind)
    Q = zo.synth_queries(B, d, n, kind=kind)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.append(X)
    ix.set_forest(f.arrays())
    for name, m, om, omode in metrics(za):
        ids, keys, counts = ix.search_batch(Q, k, m)
        oi, ok, oc = f.search_batch(Q, k, om, omode)
        assert (counts == oc).all(), name
        for b in range(B):
            c = int(oc[b])
            assert (ids[b, :c] == oi[b, :c]).all(), (name, b)
            assert (keys[b, :c] == ok[b, :c]).all(), (name, b)
            assert (ids[b, c:] == np.uint64(2**64 - 1)).all()
    # single-query entry point
    r = ix.search(Q[0], k, za.L2SquaredDistance())
    oi, ok = f.search(Q[0], k, zo.L2SQ)
    assert r == list(zip(oi.tolist(), ok.tolist()))


def test_ties_across_the_cut(za):
    """hundreds of equal keys straddling top_k: ids must come out in ascending order (tie-break on id)"""
    d = 16
    base = zo.synth_rows(40, d)
    X = np.concatenate([base[:20], np.repeat(base[20:21], 1500, axis=0), base[21:], np.repeat(base[5:6], 700, axis=0)])
    n = X.shape[0]
    Q = np.stack([base[20], base[5], base[3]])
    for M in (4000, 64):
        f = zo.Forest.build(X, M, 3, seed=8)
        ix = za.LSHIndex(d, za.LSHIndexOptions(M, 3), seed=8)
        ix.add(X)
        assert zo.canonical_forest(ix.get_forest(), d) == zo.canonical_forest(f.arrays(), d)
        for k in (10, 100, 1000):
            for name, m, om, omode in metrics(za):
                ids, keys, counts = ix.search_batch(Q, k, m)
                oi, ok, oc = f.search_batch(Q, k, om, omode)
                assert (counts == oc).all(), (M, k, name)
                for b in range(3):
                    assert (ids[b, :oc[b]] == oi[b, :oc[b]]).all() and (keys[b, :oc[b]] == ok[b, :oc[b]]).all(), (M, k, name, b)


@pytest.mark.parametrize("levels", [0, 1, 3, 100])
def test_dense_levels_do_not_change_results(za, levels):
    n, d, M, T, k = 6000, 96, 8, 5, 10
    X = zo.synth_rows(n, d)
    Q = zo.synth_queries(20, d, n)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.append(X)
    ix.set_forest(f.arrays())
    ix.set_dense_levels(levels)
    ids, keys, counts = ix.search_batch(Q, k, za.L2SquaredDistance())
    oi, ok, oc = f.search_batch(Q, k, zo.L2SQ)
    assert (counts == oc).all() and (ids == oi).all() and (keys == ok).all()
    st = ix.stats()
    assert st["planes_total"] == f.arrays()["consts"].size
    if levels == 0:
        assert st["planes_dense"] == 0
    if levels == 100:
        assert st["planes_dense"] == st["planes_total"]


# -------------------------------------------------------- lsh.rs:192-267, 411-429 (the build)
@pytest.mark.parametrize("n,d,M,T", [(4000, 64, 5, 3), (20000, 128, 64, 4), (6000, 384, 100, 2), (3000, 768, 256, 2),
                                     (1000, 30, 7, 2)])
def test_gpu_build_equals_oracle_build(za, n, d, M, T):
    X = zo.synth_rows(n, d)
    f = zo.Forest.build(X, M, T, seed=77)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), seed=77)
    ids = ix.add(X)  # LSHIndex::add on an empty index = build_index
    assert ids.tolist() == list(range(n))
    assert not ix.is_empty() and len(ix) == n
    g = ix.get_forest()
    assert zo.canonical_forest(g, d) == zo.canonical_forest(f.arrays(), d)
    # every id in exactly one leaf per tree; built leaves hold < M ids
    assert g["leaf_ids"].size == n * T
    leaves = g["plane"] < 0
    assert (g["right"][leaves] < M).all()
    for t in range(T):
        assert sorted(g["leaf_ids"][t * n:(t + 1) * n].tolist()) == list(range(n))
    # and searching the GPU-built forest gives the oracle's answers
    Q = zo.synth_queries(16, d, n)
    ids_, keys_, counts_ = ix.search_batch(Q, 10, za.L2SquaredDistance())
    oi, ok, oc = f.search_batch(Q, 10, zo.L2SQ)
    assert (counts_ == oc).all() and (ids_ == oi).all() and (keys_ == ok).all()


# ------------------------------------------------- lsh.rs:350-382, 445-462 (incremental insert)
@pytest.mark.parametrize("d,M,T,steps", [
    (16, 8, 4, [100, 1, 299]),        # single row, then a burst that splits the same leaves several times
    (128, 64, 5, [2000, 500, 37]),
    (32, 5, 15, [300, 300]),          # reference defaults
    (16, 8, 2, [3, 27, 100]),         # roots are leaves at first (fewer rows than max_node_size)
    (768, 300, 3, [1500, 900]),
])
def test_incremental_add_equals_oracle_insert(za, d, M, T, steps):
    total = sum(steps)
    X = zo.synth_rows(total, d)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), seed=31)
    n = steps[0]
    assert ix.add(X[:n]).tolist() == list(range(n))
    f = zo.Forest.build(X[:n], M, T, seed=31)
    for more in steps[1:]:
        ids = ix.add(X[n:n + more])  # trees exist -> insert path
        assert ids.tolist() == list(range(n, n + more))
        f.insert(X[:n + more], n)
        n += more
        g = ix.get_forest()
        assert zo.canonical_forest(g, d) == zo.canonical_forest(f.arrays(), d)
        leaves = g["plane"] < 0
        assert (g["right"][leaves] <= M).all()
    assert len(ix) == total
    Q = zo.synth_queries(12, d, total)
    for name, m, om, omode in metrics(za):
        i_, k_, c_ = ix.search_batch(Q, 10, m)
        oi, ok, oc = f.search_batch(Q, 10, om, omode)
        assert (c_ == oc).all() and (i_ == oi).all() and (k_ == ok).all(), name
    # every inserted row is found at distance 0
    r = ix.search(X[total - 1], 1, za.L2SquaredDistance())
    assert r[0] == (total - 1, 0)


def test_incremental_add_of_duplicates_and_compaction(za):
    d, M = 8, 6
    base = zo.synth_rows(50, d)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, 2), seed=2)
    ix.add(base)
    f = zo.Forest.build(base, M, 2, seed=2)
    X = base
    for r in range(12):  # many small adds of the same vector: unsplittable leaves, relocation garbage, compaction
        X = np.concatenate([X, np.repeat(base[7:8], 9, axis=0)])
        ix.add(X[-9:])
        f.insert(X, X.shape[0] - 9)
        assert zo.canonical_forest(ix.get_forest(), d) == zo.canonical_forest(f.arrays(), d)
    assert ix.get_forest()["leaf_ids"].size <= 2 * 2 * X.shape[0] + 1024
    i_, k_, c_ = ix.search_batch(base[:4], 20, za.L2SquaredDistance())
    oi, ok, oc = f.search_batch(base[:4], 20, zo.L2SQ)
    assert (c_ == oc).all() and (i_ == oi).all() and (k_ == ok).all()


def test_synthetic_generators_bit_exact(za):
    n, d = 3000, 384
    ix = za.LSHIndex(d, za.LSHIndexOptions(64, 2))
    ix.append_synthetic(n, seed=zo.SEED_ROWS, first_row=0, kind=0)
    assert (ix.read_rows(0, n).view(np.uint32) == zo.synth_rows(n, d).view(np.uint32)).all()
    ix2 = za.LSHIndex(128, za.LSHIndexOptions(64, 2))
    ix2.append_synthetic(500, seed=5, first_row=1000, kind=1)
    assert (ix2.read_rows(0, 500) == zo.synth_rows(500, 128, seed=5, row0=1000, kind=1)).all()
    ix3 = za.LSHIndex(64, za.LSHIndexOptions(64, 2))
    ix3.append_synthetic(700, seed=zo.SEED_ROWS, first_row=250, kind=2)
    assert (ix3.read_rows(0, 700).view(np.uint32) == zo.synth_rows(700, 64, row0=250, kind=2).view(np.uint32)).all()
    import torch
    ix4 = za.LSHIndex(64, za.LSHIndexOptions(5, 15))
    ix4.append_synthetic(900, seed=zo.SEED_ROWS, first_row=4_000_000_000, kind=3)
    assert (ix4.read_rows(0, 900).view(np.uint32) == zo.synth_rows(900, 64, row0=4_000_000_000, kind=3).view(np.uint32)).all()
    ix4.close()
    for kind, dd in ((0, 384), (1, 128), (2, 64), (3, 96)):
        q = torch.empty((33, dd), dtype=torch.float32, device="cuda")
        za.synth_queries_device(0, q.data_ptr(), 12345, 33, dd, b0=7, kind=kind)
        want = zo.synth_queries(33, dd, 12345, b0=7, kind=kind)
        assert (q.cpu().numpy().view(np.uint32) == want.view(np.uint32)).all()


# ------------------------------------------------------------------------------ edge cases
def test_edge_cases(za):
    d = 16
    ix = za.LSHIndex(d)
    assert ix.is_empty() and ix.no_vectors() and ix.no_trees()
    Q = zo.synth_queries(3, d, 10)
    ids, keys, counts = ix.search_batch(Q, 5, za.L2SquaredDistance())  # empty index -> empty result
    assert (counts == 0).all() and (ids == np.uint64(2**64 - 1)).all()
    # fewer rows than max_node_size: every tree is one leaf; top_k > rows
    X = zo.synth_rows(3, d)
    ix.add(X)
    f = zo.Forest.build(X, 5, 15)
    ids, keys, counts = ix.search_batch(Q, 10, za.L2SquaredDistance())
    oi, ok, oc = f.search_batch(Q, 10, zo.L2SQ)
    assert (counts == oc).all() and (counts == 3).all()
    assert (ids[:, :3] == oi[:, :3]).all() and (keys[:, :3] == ok[:, :3]).all()
    # duplicates of one vector: unsplittable node -> depth guard on both sides, same forest
    Xd = np.repeat(zo.synth_rows(1, d), 40, axis=0)
    ixd = za.LSHIndex(d, za.LSHIndexOptions(8, 2), seed=3)
    ixd.add(Xd)
    fd = zo.Forest.build(Xd, 8, 2, seed=3)
    assert zo.canonical_forest(ixd.get_forest(), d) == zo.canonical_forest(fd.arrays(), d)
    i2, k2, c2 = ixd.search_batch(Q, 10, za.L2SquaredDistance())
    o2 = fd.search_batch(Q, 10, zo.L2SQ)
    assert (c2 == o2[2]).all() and (i2 == o2[0]).all() and (k2 == o2[1]).all()
    # limits and argument errors
    with pytest.raises(za.ZhError):
        ix.search_batch(Q, za.MAX_TOPK + 1, za.L2SquaredDistance())
    ix.clear()
    assert ix.is_empty()
    # id_base shifts returned ids
    ixb = za.LSHIndex(d, za.LSHIndexOptions(5, 3), id_base=1000)
    assert ixb.add(zo.synth_rows(50, d)).tolist() == list(range(1000, 1050))
    r = ixb.search(zo.synth_rows(50, d)[7], 1, za.L2SquaredDistance())
    assert r[0][0] == 1007 and r[0][1] == 0


def test_database_mirror(za):
    """Database::insert_records / query_vectors (core.rs:245-254, 290-313)"""
    d, n = 32, 500
    X = zo.synth_rows(n, d)
    db = za.Database(d, za.L2SquaredDistance(), za.LSHIndexOptions(16, 6))
    assert db.query_vectors(X[:2], 3) == {}
    db.insert_records(X, [f"doc{i}".encode() for i in range(n)])
    res = db.query_vectors(X[[5, 77]], 3)
    assert 5 in res[0] and res[0][5] == b"doc5" and 77 in res[1]
    assert len(res[0]) == 3
    db.remove([5])  # core.rs:205-214
    assert 5 not in db.query_vectors(X[[5]], 3)[0]
    db.insert_records(X[[77]], [b"dup"])  # an exact duplicate of row 77 ...
    db.deduplicate()                      # ... goes again (core.rs:216-225)
    assert len(db.index) == n - 1
    db.clear_database()
    assert db.query_vectors(X[:1], 3) == {}


# ---------------------------------------------------------------- shard merge
@pytest.mark.parametrize("S,k", [(2, 10), (4, 100), (8, 10), (8, 1024)])
def test_shard_merge_device(za, S, k):
    import torch
    n, d, M, T, B = 4000, 64, 64, 5, 13
    X = zo.synth_rows(n, d)
    Q = zo.synth_queries(B, d, n)
    per = n // S
    kk = min(k, 128)  # per-shard lists of kk valid entries inside k-wide slots
    all_ids, all_keys, all_counts = [], [], []
    for s in range(S):
        Xs = X[s * per:(s + 1) * per]
        ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), id_base=s * per, seed=100 + s)
        ix.add(Xs)
        i, kys, c = ix.search_batch(Q, kk, za.L2SquaredDistance())
        # oracle of the shard: same forest seed, local ids + base
        f = zo.Forest.build(Xs, M, T, seed=100 + s)
        oi, ok, oc = f.search_batch(Q, kk, zo.L2SQ)
        assert (c == oc).all()
        for b in range(B):
            assert (i[b, :oc[b]] == oi[b, :oc[b]] + np.uint64(s * per)).all() and (kys[b, :oc[b]] == ok[b, :oc[b]]).all()
        pad_i = np.full((B, k), 2**64 - 1, np.uint64)
        pad_k = np.full((B, k), 2**64 - 1, np.uint64)
        pad_i[:, :kk], pad_k[:, :kk] = i, kys
        all_ids.append(pad_i), all_keys.append(pad_k), all_counts.append(c)
    ids = np.stack(all_ids)
    keys = np.stack(all_keys)
    counts = np.stack(all_counts)
    want = zo.merge_topk(ids, keys, counts, k)
    t_ids = torch.from_numpy(ids.view(np.int64)).cuda()
    t_keys = torch.from_numpy(keys.view(np.int64)).cuda()
    t_counts = torch.from_numpy(counts.view(np.int32)).cuda()
    o_ids = torch.empty((B, k), dtype=torch.int64, device="cuda")
    o_keys = torch.empty((B, k), dtype=torch.int64, device="cuda")
    o_counts = torch.empty(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    za.merge_topk_device(0, S, B, k, t_ids.data_ptr(), t_keys.data_ptr(), t_counts.data_ptr(), o_ids.data_ptr(),
                         o_keys.data_ptr(), o_counts.data_ptr())
    gi = o_ids.cpu().numpy().view(np.uint64)
    gk = o_keys.cpu().numpy().view(np.uint64)
    gc = o_counts.cpu().numpy().view(np.uint32)
    assert (gc == want[2]).all()
    for b in range(B):
        assert (gi[b, :gc[b]] == want[0][b, :gc[b]]).all() and (gk[b, :gc[b]] == want[1][b, :gc[b]]).all()
    # the packed layout (one buffer per shard = one all-gather per batch) merges to the same result
    from zebra_amd import sharding
    W = za.packed_result_words(B, k)
    assert W == 2 * B * k + (B + 1) // 2
    g_packed = torch.zeros((S, W), dtype=torch.int64, device="cuda")
    for s_ in range(S):
        a, b_, c = sharding.packed_views(torch, g_packed[s_], B, k)
        a.copy_(t_ids[s_]), b_.copy_(t_keys[s_]), c.copy_(t_counts[s_])
    p_ids, p_keys, p_counts = torch.empty_like(o_ids), torch.empty_like(o_keys), torch.empty_like(o_counts)
    torch.cuda.synchronize()
    za.merge_topk_packed_device(0, S, B, k, g_packed.data_ptr(), p_ids.data_ptr(), p_keys.data_ptr(), p_counts.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(p_counts, o_counts)
    for b in range(B):
        assert torch.equal(p_ids[b, :gc[b]], o_ids[b, :gc[b]]) and torch.equal(p_keys[b, :gc[b]], o_keys[b, :gc[b]])


def test_search_device_entry_point(za):
    import torch
    n, d, M, T, k, B = 10000, 384, 128, 15, 10, 40
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.append_synthetic(n)
    ix.build()
    X = zo.synth_rows(n, d)
    f = zo.Forest.build(X, M, T)
    q = torch.empty((B, d), dtype=torch.float32, device="cuda")
    za.synth_queries_device(0, q.data_ptr(), n, B, d)
    ids = torch.empty((B, k), dtype=torch.int64, device="cuda")
    keys = torch.empty((B, k), dtype=torch.int64, device="cuda")
    counts = torch.empty(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ix.set_profiling(2)
    m = za.CosineDistance(parity=True)
    ix.search_batch_device(q.data_ptr(), B, k, m, ids.data_ptr(), keys.data_ptr(), counts.data_ptr(),
                           torch.cuda.current_stream().cuda_stream)
    oi, ok, oc = f.search_batch(zo.synth_queries(B, d, n), k, zo.COSINE, zo.PARITY)
    assert (counts.cpu().numpy().view(np.uint32) == oc).all()
    assert (ids.cpu().numpy().view(np.uint64) == oi).all() and (keys.cpu().numpy().view(np.uint64) == ok).all()
    st = ix.stats()
    assert st["timed_batches"] == 1 and st["ms_sweep"] > 0 and st["rows_scored"] >= st["rows_unique"] > 0
    assert st["visits"] >= B * T


def test_pipelined_contexts_match_blocking_search(za):
    """zh_search_begin / finish / wait with two contexts on two streams == the blocking call"""
    import torch
    n, d, M, T, k, B = 20000, 128, 256, 8, 10, 64
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.append_synthetic(n)
    ix.build()
    f = zo.Forest.from_arrays(zo.synth_rows(n, d), M, ix.get_forest())
    m = za.L2SquaredDistance()
    qs = []
    for i in range(5):
        q = torch.empty((B, d), dtype=torch.float32, device="cuda")
        za.synth_queries_device(0, q.data_ptr(), n, B, d, b0=i * B)
        qs.append(q)
    torch.cuda.synchronize()
    slots = [dict(ctx=ix.search_context(), st=torch.cuda.Stream(), ids=torch.empty((B, k), dtype=torch.int64, device="cuda"),
                  keys=torch.empty((B, k), dtype=torch.int64, device="cuda"), counts=torch.empty(B, dtype=torch.int32, device="cuda"))
             for _ in range(2)]
    got = {}
    heavy = torch.cuda.Stream()
    slots[0]["ctx"].begin(qs[0].data_ptr(), B, k, m, slots[0]["st"].cuda_stream)
    for i in range(5):
        if i + 1 < 5:
            sl = slots[(i + 1) % 2]
            if (i - 1) in got and got[i - 1] is None:
                pass
            # the slot's previous results must be consumed before its buffers are reused
            if i - 1 >= 0:
                sl["ctx"].wait()
                got[i - 1] = (sl["ids"].cpu().numpy().view(np.uint64).copy(), sl["keys"].cpu().numpy().view(np.uint64).copy())
            sl["ctx"].begin(qs[i + 1].data_ptr(), B, k, m, sl["st"].cuda_stream)
        sl = slots[i % 2]
        sl["ctx"].finish(sl["ids"].data_ptr(), sl["keys"].data_ptr(), sl["counts"].data_ptr(),
                         heavy.cuda_stream if i % 2 else None)
    for i in (3, 4):
        sl = slots[i % 2]
        sl["ctx"].wait()
        got[i] = (sl["ids"].cpu().numpy().view(np.uint64).copy(), sl["keys"].cpu().numpy().view(np.uint64).copy())
    for i in range(5):
        oi, ok, _ = f.search_batch(zo.synth_queries(B, d, n, b0=i * B), k, zo.L2SQ)
        assert (got[i][0] == oi).all() and (got[i][1] == ok).all(), i
    # misuse is reported, not crashed on
    c = ix.search_context()
    with pytest.raises(za.ZhError):
        c.finish(0, 0, 0)
    c.begin(qs[0].data_ptr(), B, k, m)
    with pytest.raises(za.ZhError):
        c.begin(qs[0].data_ptr(), B, k, m)
    c.finish(slots[0]["ids"].data_ptr(), slots[0]["keys"].data_ptr(), slots[0]["counts"].data_ptr())
    c.wait()


def test_concurrent_searches_from_threads(za):
    """the reference calls search from rayon workers (core.rs:299-303): concurrent calls on one index are safe"""
    import threading
    n, d, M, T, k = 8000, 64, 128, 6, 10
    X = zo.synth_rows(n, d)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    f = zo.Forest.build(X, M, T)
    Qs = [zo.synth_queries(16, d, n, b0=16 * i) for i in range(8)]
    want = [f.search_batch(q, k, zo.L2SQ) for q in Qs]
    got, errs = [None] * 8, []

    def work(i):
        try:
            for _ in range(5):
                got[i] = ix.search_batch(Qs[i], k, za.L2SquaredDistance())
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs
    for i in range(8):
        assert (got[i][0] == want[i][0]).all() and (got[i][1] == want[i][1]).all() and (got[i][2] == want[i][2]).all()


# ------------------------------------------ lsh.rs:473-503 (remove), 270-288 (deduplicate) -- f4
def test_remove_and_deduplicate(za):
    n, d, M, T = 3000, 32, 24, 5
    X = zo.synth_rows(n, d)
    X[1000], X[2000], X[2999] = X[5], X[5], X[17]  # exact duplicates of earlier rows
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), seed=9)
    ix.add(X)
    f = zo.Forest.build(X, M, T, seed=9)
    want_dups = np.nonzero(zo.find_duplicates(X))[0]
    assert want_dups.tolist() == [1000, 2000, 2999]
    got = ix.deduplicate()
    assert got.tolist() == want_dups.tolist()
    assert f.remove(want_dups).all()
    assert len(ix) == n - 3
    assert zo.canonical_forest(ix.get_forest(), d) == zo.canonical_forest(f.arrays(), d)
    assert ix.deduplicate().size == 0  # idempotent
    # remove: present ids leave every tree; unknown / repeated / already removed ids are reported as absent
    ids = np.array([7, 8, 7, 1000, 10**9, 2500], np.uint64)
    removed = ix.remove(ids)
    assert sorted(removed.tolist()) == [7, 8, 2500]
    f.remove([7, 8, 2500])
    assert len(ix) == n - 6
    g = ix.get_forest()
    assert zo.canonical_forest(g, d) == zo.canonical_forest(f.arrays(), d)
    for t in range(T):  # really gone from every tree
        stack, seen = [int(g["roots"][t])], []
        while stack:
            m = stack.pop()
            if g["plane"][m] < 0:
                seen += g["leaf_ids"][g["left"][m]:g["left"][m] + g["right"][m]].tolist()
            else:
                stack += [int(g["left"][m]), int(g["right"][m])]
        assert len(seen) == n - 6 and not ({7, 8, 2500, 1000, 2000, 2999} & set(seen))
    Q = zo.synth_queries(10, d, n)
    i_, k_, c_ = ix.search_batch(np.concatenate([Q, X[[7, 5]]]), 10, za.L2SquaredDistance())
    oi, ok, oc = f.search_batch(np.concatenate([Q, X[[7, 5]]]), 10, zo.L2SQ)
    assert (c_ == oc).all() and (i_ == oi).all() and (k_ == ok).all()
    assert 7 not in i_[10] and i_[11, 0] == 5
    # inserting after a removal still matches the oracle
    more = zo.synth_rows(200, d, row0=50000)
    ix.add(more)
    Xall = np.concatenate([X, more])
    f.insert(Xall, n)
    assert zo.canonical_forest(ix.get_forest(), d) == zo.canonical_forest(f.arrays(), d)
    # a rebuild keeps removed rows out
    ix.build()
    g = ix.get_forest()
    assert g["leaf_ids"].size == T * (n + 200 - 6) and not ({7, 8, 2500, 1000} & set(g["leaf_ids"].tolist()))


def test_clear_refill_remove_resamples_the_live_rows(za):
    """the hyperplane sampler's list of live rows was cached on (stored rows, removed rows) alone; clear, a
    refill to the same count and the removal of the same NUMBER of different rows reused the old list, so later splits drew
    sample points from removed rows.  The forest after clear / add / remove / add must equal the oracle's."""
    n, d, M, T = 3000, 32, 24, 5
    X1, X2 = zo.synth_rows(n, d), zo.synth_rows(n, d, row0=10**6)
    more = zo.synth_rows(600, d, row0=2 * 10**6)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), seed=11)
    ix.add(X1)
    ix.remove(np.arange(1, 51, dtype=np.uint64))
    ix.add(more)                       # splits sample the live rows of X1: the list is built for (3600 stored, 50 removed)
    ix.clear()
    ix.add(X2)
    ix.remove(np.arange(100, 150, dtype=np.uint64))   # same number of removals, other rows
    ix.add(more)
    f = zo.Forest.build(X2, M, T, seed=11)
    f.remove(np.arange(100, 150, dtype=np.uint64))
    f.insert(np.concatenate([X2, more]), n)
    assert zo.canonical_forest(ix.get_forest(), d) == zo.canonical_forest(f.arrays(), d)
    ix.close()


def test_three_deep_pipeline_matches_blocking_calls(za):
    """30 batches, three in flight, sweeps on the index's shared stream: every batch equals the blocking call"""
    import torch
    n, d, M, T, k, B, NB, NS = 300000, 128, 1024, 10, 20, 512, 30, 3
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.append_synthetic(n)
    ix.build()
    m = za.CosineDistance(parity=False)
    qs = []
    for i in range(NB):
        q = torch.empty((B, d), dtype=torch.float32, device="cuda")
        za.synth_queries_device(0, q.data_ptr(), n, B, d, b0=i * B)
        qs.append(q)
    # blocking reference results (same library path, already checked against the oracle elsewhere)
    ref = []
    ids = torch.empty((B, k), dtype=torch.int64, device="cuda")
    keys = torch.empty((B, k), dtype=torch.int64, device="cuda")
    counts = torch.empty(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for i in range(NB):
        ix.search_batch_device(qs[i].data_ptr(), B, k, m, ids.data_ptr(), keys.data_ptr(), counts.data_ptr())
        ref.append((ids.cpu().clone(), keys.cpu().clone(), counts.cpu().clone()))
    slots = [dict(ctx=ix.search_context(), st=torch.cuda.Stream(priority=-1), ids=torch.empty_like(ids), keys=torch.empty_like(keys),
                  counts=torch.empty_like(counts)) for _ in range(NS)]
    heavy = ix.sweep_stream()
    got = [None] * NB
    for i in range(NB):
        sl = slots[i % NS]
        if i >= NS:  # consume the slot's previous batch before its buffers are reused
            sl["ctx"].wait()
            got[i - NS] = (sl["ids"].cpu().clone(), sl["keys"].cpu().clone(), sl["counts"].cpu().clone())
        sl["ctx"].begin(qs[i].data_ptr(), B, k, m, sl["st"].cuda_stream)
        sl["ctx"].finish(sl["ids"].data_ptr(), sl["keys"].data_ptr(), sl["counts"].data_ptr(), heavy)
    for i in range(NB - NS, NB):
        sl = slots[i % NS]
        sl["ctx"].wait()
        got[i] = (sl["ids"].cpu().clone(), sl["keys"].cpu().clone(), sl["counts"].cpu().clone())
    for i in range(NB):
        assert torch.equal(got[i][2], ref[i][2]) and torch.equal(got[i][0], ref[i][0]) and torch.equal(got[i][1], ref[i][1]), i
    st = ix.stats()
    assert st["rows_scored"] > 0


_LOG_FALLBACK_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import zebra_amd as za
from oracle import zebra_oracle as zo
n, d, B, k = 30000, 96, 48, 10
X = zo.synth_rows(n, d); Q = zo.synth_queries(B, d, n)
ix = za.LSHIndex(d, za.LSHIndexOptions(5, 6)); ix.add(X)
f = zo.Forest.from_arrays(X, 5, ix.get_forest())
for metric, om, omode in ((za.L2SquaredDistance(), zo.L2SQ, 0), (za.CosineDistance(parity=True), zo.COSINE, zo.PARITY)):
    for _ in range(2):
        ids, keys, counts = ix.search_batch(Q, k, metric)
        oi, ok, oc = f.search_batch(Q, k, om, omode)
        assert (counts == oc).all() and (ids == oi).all() and (keys == ok).all()
assert ix.stats()["visits"] > 64 * B * 6, ix.stats()["visits"]
print("OK")
"""


@pytest.mark.parametrize("env", [{"ZH_WALK_LOG_CHUNKS": "3", "ZH_WALK_LOG_FIXED": "1"},   # always overflows: emit walk
                                 {"ZH_WALK_LOG_CHUNKS": "3"},                               # overflows once, then grows
                                 {"ZH_WALK_LOG_CHUNKS": "1000000"}])                        # never overflows
def test_visit_log_overflow_falls_back_to_the_emit_walk(za, env):
    """walk_kernel<false> logs the visits beyond the inline ones in chunks from a bump allocator; when the pool runs
    out the batch is walked a second time (walk_kernel<true>).  Both ways give the oracle's answers."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", _LOG_FALLBACK_SCRIPT, root], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


_LAZY_LOG_SCRIPT = """
import sys
sys.path.insert(0, sys.argv[1])
import zebra_amd as za
from oracle import zebra_oracle as zo
n, d, M, T, k, B = 20000, 128, 5, 5, 10, 16
X = zo.synth_rows(n, d)
Q = zo.synth_queries(B, d, n)
ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
ix.add(X)
ix.set_dense_levels(100)
ix.set_hash_mode("scores")
f = zo.Forest.from_arrays(X, M, ix.get_forest())
oi, ok, oc = f.search_batch(Q, k, zo.L2SQ, 0)
for mode in ("auto", "auto", "auto", "auto", "leaf", "leaf"):  # the blocked view (and with it the lazily fixed signs) arrives with the third batch
    ix.set_sweep_mode(mode)
    ids, keys, counts = ix.search_batch(Q, k, za.L2SquaredDistance())
    assert (counts == oc).all() and (ids == oi).all() and (keys == ok).all(), mode
st = ix.stats()
assert st["hash_from_scores"] == 1 and st["hash_exact_fixups"] > 0, st
print("OK")
"""


@pytest.mark.parametrize("env", [{"ZH_WALK_LOG_CHUNKS": "3", "ZH_WALK_LOG_FIXED": "1"}, {"ZH_WALK_LOG_CHUNKS": "3"}])
def test_visit_log_overflow_with_lazily_fixed_signs(za, env):
    """the row-score hash leaves its uncertain signs flagged for the blocked walk (zh_score.hip); a batch whose visit log runs out is
    walked again by the pointer walk, which reads plain bits: the flagged signs are recomputed first -- both walks must see the
    same forest, or the second one writes past what the first one counted"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", _LAZY_LOG_SCRIPT, root], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_reference_format_round_trip_through_the_gpu_index(za):
    """a forest built on the GPU, written out as the reference's tree values (bincode-legacy Node<N>,
    lsh.rs:99-105), read back against the vectors in ANOTHER row order (fjall iterates by key), serves the same answers."""
    from zebra_amd import refformat as rf
    n, d, M, T, k = 5000, 96, 24, 5, 10
    X = zo.synth_rows(n, d)
    Q = zo.synth_queries(16, d, n)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    uu = np.random.default_rng(11).integers(0, 256, (n, 16), dtype=np.uint8)
    blobs = rf.encode_trees(ix.get_forest(), d, uu)
    vals = rf.encode_embeddings(X)
    order = np.lexsort(uu.T[::-1])                       # the key order a partition scan returns
    ix2 = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix2.append(rf.decode_embeddings([vals[i] for i in order], d))
    forest, unknown = rf.decode_trees(blobs, d, uu[order])
    assert unknown == 0
    ix2.set_forest(forest)
    m = za.L2SquaredDistance()
    i1, k1, c1 = ix.search_batch(Q, k, m)
    i2, k2, c2 = ix2.search_batch(Q, k, m)
    assert (c1 == c2).all() and (k1 == k2).all() and (order[i2.astype(np.int64)] == i1.astype(np.int64)).all()
