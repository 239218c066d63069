"""Seeded fuzz: random small configurations, HIP path vs oracle, bit for bit (forest, ids, keys, counts) --
odd dimensions, leaves of 1, single trees, k larger than the index, every metric, incremental adds and removes,
signs from the dense MFMA kernel or from the walk's on-demand chains (random number of dense levels).
ZH_FUZZ_SEEDS=first:count runs a soak over other seeds."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import zebra_oracle as zo  # noqa: E402


def _metrics(za, rng):
    p = int(rng.integers(-4, 9))  # any i32 is a legal power; 0 is the derived Default (distance.rs:160-165)
    return [(za.L2SquaredDistance(), zo.L2SQ, 0), (za.L2Distance(), zo.L2, 0), (za.CosineDistance(True), zo.COSINE, zo.PARITY),
            (za.CosineDistance(False), zo.COSINE, zo.CORRECTED), (za.ChebyshevDistance(), zo.CHEBYSHEV, 0),
            (za.CanberraDistance(), zo.CANBERRA, 0), (za.BrayCurtisDistance(), zo.BRAY_CURTIS, 0),
            (za.ManhattanDistance(), zo.MANHATTAN, 0), (za.L3Distance(), zo.L3, 0), (za.L4Distance(), zo.L4, 0),
            (za.HammingDistance(), zo.HAMMING, 0), (za.MinkowskiDistance(p), zo.MINKOWSKI, p), (za.PNormDistance(p), zo.PNORM, p)]


_first, _count = (int(x) for x in os.environ.get("ZH_FUZZ_SEEDS", "0:24").split(":"))


@pytest.mark.parametrize("seed", range(_first, _first + _count))
def test_fuzz_config(seed):
    import zebra_amd as za
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.choice([1, 2, 5, 17, 32, 63, 64, 100, 128, 200, 384, 500, 768, 1000, 1540]))
    n = int(rng.integers(1, 2500))
    M = int(rng.choice([1, 2, 5, 9, 33, 100, 400, 5000]))
    T = int(rng.integers(1, 7))
    k = int(rng.choice([1, 3, 10, 37, 100, 500]))
    B = int(rng.integers(1, 20))
    kind = int(rng.choice([0, 0, 1, 2]))
    X = zo.synth_rows(n, d, seed=seed, kind=kind)
    if rng.random() < 0.3 and n > 10:  # some exact duplicates
        X[rng.integers(0, n, 5)] = X[0]
    Q = zo.synth_queries(B, d, n, seed_rows=seed, kind=kind)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), seed=seed)
    ix.add(X)
    ix.set_dense_levels(int(rng.choice([-1, -1, 0, 1, 3, 64])))
    ix.set_sweep_mode(str(rng.choice(["auto", "leaf", "scan", "approx", "approx-valu", "leaf-half"])))  # (scan / approx fall back where they have no kernel: odd d, other metrics)
    f = zo.Forest.build(X, M, T, seed=seed)
    assert zo.canonical_forest(ix.get_forest(), d) == zo.canonical_forest(f.arrays(), d), (d, n, M, T)
    mets = _metrics(za, rng)
    for m, om, omode in [mets[i] for i in rng.choice(len(mets), 4, replace=False)]:
        ids, keys, counts = ix.search_batch(Q, k, m)
        oi, ok, oc = f.search_batch(Q, k, om, omode)
        assert (counts == oc).all(), (d, n, M, T, k, om)
        for b in range(B):
            c = int(oc[b])
            kg, kw = keys[b, :c], ok[b, :c]
            same = (kg == kw) | ((om >= zo.CHEBYSHEV) & np.isnan(kg.astype(np.uint32).view(np.float32)) & np.isnan(kw.astype(np.uint32).view(np.float32)))
            assert same.all() and (ids[b, :c] == oi[b, :c]).all(), (d, n, M, T, k, om, b)
    # grow, shrink, search again
    more = int(rng.integers(1, 300))
    X2 = np.concatenate([X, zo.synth_rows(more, d, seed=seed, row0=10**6, kind=kind)])
    ix.add(X2[n:])
    f.insert(X2, n)
    assert zo.canonical_forest(ix.get_forest(), d) == zo.canonical_forest(f.arrays(), d), ("insert", d, n, M, T)
    gone = rng.choice(n + more, size=min(7, n + more), replace=False).astype(np.uint64)
    assert sorted(ix.remove(gone).tolist()) == sorted(gone[f.remove(gone)].tolist())
    assert zo.canonical_forest(ix.get_forest(), d) == zo.canonical_forest(f.arrays(), d), ("remove", d, n, M, T)
    ids, keys, counts = ix.search_batch(Q, k, za.L2SquaredDistance())
    oi, ok, oc = f.search_batch(Q, k, zo.L2SQ)
    assert (counts == oc).all()
    for b in range(B):
        assert (ids[b, :oc[b]] == oi[b, :oc[b]]).all() and (keys[b, :oc[b]] == ok[b, :oc[b]]).all()


_mfirst, _mcount = (int(x) for x in os.environ.get("ZH_FUZZ_MEDIUM_SEEDS", "0:4").split(":"))


@pytest.mark.parametrize("seed", range(_mfirst, _mfirst + _mcount))
def test_fuzz_medium_wandering_walks(seed):
    """Larger indices with small leaves: hundreds to thousands of leaf visits per (query, tree) pair, so the visit log
    runs through many chunks and buffer flushes, the select kernel packs many tiny visits per block and the final
    kernel streams with its threshold filter; random dense levels switch between on-demand chains and dense signs."""
    import zebra_amd as za
    rng = np.random.default_rng(7000 + seed)
    d = int(rng.choice([8, 24, 64, 100, 384]))
    n = int(rng.integers(20_000, 120_000))
    M = int(rng.choice([2, 3, 5, 8, 17]))
    T = int(rng.integers(1, 6))
    k = int(rng.choice([1, 5, 10, 40, 200]))
    B = int(rng.integers(1, 40))
    kind = int(rng.choice([0, 1, 2]))
    X = zo.synth_rows(n, d, seed=seed, kind=kind)
    Q = zo.synth_queries(B, d, n, seed_rows=seed, kind=kind)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), seed=seed)
    ix.add(X)
    f = zo.Forest.from_arrays(X, M, ix.get_forest())
    for dense in (-1, int(rng.choice([0, 2, 5]))):
        ix.set_dense_levels(dense)
        ix.set_sweep_mode("scan" if dense == -1 else "leaf")
        ix.set_hash_mode("scores" if dense == -1 else "dense")  # (row scores whenever the batch hashes every plane and b % 4 == 0)
        m, om, omode = [(za.L2SquaredDistance(), zo.L2SQ, 0), (za.CosineDistance(True), zo.COSINE, zo.PARITY),
                        (za.ManhattanDistance(), zo.MANHATTAN, 0)][int(rng.integers(0, 3))]
        ids, keys, counts = ix.search_batch(Q, k, m)
        oi, ok, oc = f.search_batch(Q, k, om, omode)
        assert (counts == oc).all(), (d, n, M, T, k, B, dense)
        for b in range(B):
            c = int(oc[b])
            assert (ids[b, :c] == oi[b, :c]).all() and (keys[b, :c] == ok[b, :c]).all(), (d, n, M, T, k, B, dense, b)


_pfirst, _pcount = (int(x) for x in os.environ.get("ZH_FUZZ_PREFILTER_SEEDS", "0:12").split(":"))
_pf_seen = []  # per seed: batches that were prefiltered (checked by the last test of the file)


@pytest.mark.parametrize("seed", range(_pfirst, _pfirst + _pcount))
def test_fuzz_prefilter(seed):
    """small-leaf forests with every sign from row scores: the prefilter (candidates judged on the scores, zh_search.hip) and -- from
    the third batch of a forest on, when the blocked view exists -- the lazily fixed signs, against the oracle and against the sweep;
    ties (integer rows, duplicates), clustered rows, zero rows, a query that is a stored row, inserts and removals in between"""
    import zebra_amd as za
    rng = np.random.default_rng(5000 + seed)
    d = int(rng.choice([4, 17, 32, 64, 100, 128, 384, 768]))
    n = int(rng.integers(200, 6000))
    M = int(rng.choice([1, 2, 3, 5, 5, 8]))
    T = int(rng.integers(1, 7))
    k = int(rng.choice([1, 3, 10, 10, 37, 64]))
    B = int(rng.integers(1, 24))
    kind = int(rng.choice([0, 0, 1, 2]))
    X = zo.synth_rows(n, d, seed=seed, kind=kind)
    if rng.random() < 0.4:
        X[rng.integers(0, n, 6)] = X[0]          # exact duplicates: equal keys inside a leaf
    if rng.random() < 0.2:
        X[rng.integers(0, n, 3)] = 0             # zero rows: simsimd's special cases
    Q = zo.synth_queries(B, d, n, seed_rows=seed, kind=kind)
    if rng.random() < 0.3:
        Q[0] = X[int(rng.integers(0, n))]        # distance 0 / cosine 1 exactly
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), seed=seed)
    ix.add(X)
    ix.set_hash_mode("scores")
    ix.set_dense_levels(100)
    f = zo.Forest.build(X, M, T, seed=seed)
    mets = [(za.L2SquaredDistance(), zo.L2SQ, 0), (za.L2Distance(), zo.L2, 0), (za.CosineDistance(True), zo.COSINE, zo.PARITY),
            (za.CosineDistance(False), zo.COSINE, zo.CORRECTED)]
    seen_pf = 0
    for it in range(5):  # the blocked view (and with it the lazily fixed signs) arrives with the third batch
        m, om, omode = mets[int(rng.integers(0, 4))]
        ix.set_sweep_mode("leaf" if it == 3 else "auto")
        ids, keys, counts = ix.search_batch(Q, k, m)
        st = ix.stats()
        seen_pf += st["prefiltered"]
        oi, ok, oc = f.search_batch(Q, k, om, omode)
        assert (counts == oc).all(), (d, n, M, T, k, om, it)
        for b in range(B):
            c = int(oc[b])
            assert (ids[b, :c] == oi[b, :c]).all() and (keys[b, :c] == ok[b, :c]).all(), (d, n, M, T, k, om, omode, it, b, st["prefiltered"])
        if it == 1:  # grow and shrink between batches: the per-slot norms and the blocked view are rebuilt
            more = int(rng.integers(1, 200))
            X = np.concatenate([X, zo.synth_rows(more, d, seed=seed, row0=10**6, kind=kind)])
            ix.add(X[n:])
            f.insert(X, n)
            n += more
            gone = rng.choice(n, size=min(5, n), replace=False).astype(np.uint64)
            ix.remove(gone)
            f.remove(gone)
            ix.set_dense_levels(100)
    _pf_seen.append(seen_pf)
    ix.close()


def test_fuzz_prefilter_was_exercised():
    """the seeds above must actually reach the prefilter (a sweep would pass them too)"""
    if not _pf_seen:
        pytest.skip("no prefilter seeds ran in this process")
    assert sum(1 for s_ in _pf_seen if s_ >= 2) >= 0.6 * len(_pf_seen), _pf_seen


_afirst, _acount = (int(x) for x in os.environ.get("ZH_FUZZ_APPROX_SEEDS", "0:16").split(":"))


@pytest.mark.parametrize("seed", range(_afirst, _afirst + _acount))
def test_fuzz_half_width_scan(seed):
    """the table scan with half-width queries (zh_approx.hip) forced wherever it is implemented: every dimension it has a kernel for,
    leaves from far below to far above top_k (backup visits -> the exact path), duplicates and integer rows (ties at every cut), rows and
    queries scaled by large powers of two (the per-query fp16 scale), all four simsimd-path keys.  ZH_FUZZ_APPROX_SEEDS=first:count soaks."""
    import zebra_amd as za
    rng = np.random.default_rng(77000 + seed)
    d = int(rng.choice([128, 256, 384, 512, 768, 1024]))
    n = int(rng.integers(300, 7000))
    M = int(rng.choice([12, 40, 150, 600, 3000, 9000]))
    T = int(rng.choice([1, 3, 6, 15, 16]))
    k = int(rng.choice([1, 5, 10, 50, 100, 256]))
    B = int(rng.integers(1, 48))
    kind = int(rng.choice([0, 0, 1, 2]))
    X = zo.synth_rows(n, d, seed=seed, kind=kind)
    Q = zo.synth_queries(B, d, n, seed_rows=seed, kind=kind)
    if rng.random() < 0.4:
        X[rng.integers(0, n, 8)] = X[0]
    if rng.random() < 0.3:  # small integers: exact ties everywhere
        X = np.round(X).astype(np.float32)
        Q = np.round(Q).astype(np.float32)
    if rng.random() < 0.3:  # a large power-of-two scale on everything (norms near the f32 range's ends stay finite at 2^+-40)
        sc = np.float32(2.0 ** int(rng.choice([-40, -20, 20, 40])))
        X, Q = X * sc, Q * sc
    if rng.random() < 0.2:
        Q[0] = X[int(rng.integers(0, n))]
    if rng.random() < 0.2:
        Q[B - 1] = 0.0
    f = zo.Forest.build(X, M, T, seed=seed)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), seed=seed)
    ix.add(X)
    assert zo.canonical_forest(ix.get_forest(), d) == zo.canonical_forest(f.arrays(), d), (d, n, M, T)
    if seed % 2:  # d = 128 leaf by leaf: every other case FUSED (intervals, bounds and lists inside the sweep; the library picks it for long leaves only)
        os.environ["ZH_S128H_FUSED"] = "1"
    else:
        os.environ.pop("ZH_S128H_FUSED", None)
    ix.set_sweep_mode("approx-valu" if seed % 3 == 2 else ("leaf-half" if d == 128 and seed % 3 == 1 else "approx"))  # (every third case: the VALU kernel on the
    # f32 rows instead of the matrix cores; d = 128, every third: leaf by leaf at half width)
    ix.set_hash_mode("dense")
    used = 0
    for m, om, omode in ((za.L2SquaredDistance(), zo.L2SQ, 0), (za.L2Distance(), zo.L2, 0), (za.CosineDistance(True), zo.COSINE, zo.PARITY),
                         (za.CosineDistance(False), zo.COSINE, zo.CORRECTED)):
        ids, keys, counts = ix.search_batch(Q, k, m)
        used += ix.stats()["approx_scan"]
        oi, ok, oc = f.search_batch(Q, k, om, omode)
        assert (counts == oc).all(), (seed, d, n, M, T, k, om, omode)
        for b in range(B):
            c = int(oc[b])
            assert (keys[b, :c] == ok[b, :c]).all() and (ids[b, :c] == oi[b, :c]).all(), (seed, d, n, M, T, k, om, omode, b)
    ix.close()
    os.environ.pop("ZH_S128H_FUSED", None)


_ffirst, _fcount = (int(x) for x in os.environ.get("ZH_FUZZ_FUSED_SEEDS", "0:14").split(":"))


@pytest.mark.parametrize("seed", range(_ffirst, _ffirst + _fcount))
def test_fuzz_fused_leaf_sweep(seed, monkeypatch):
    """d = 128, leaf by leaf at half width with the FUSED sweep forced (ZH_S128H_FUSED=1: intervals, per-chunk bounds and the queries' lists inside
    sweep128h_lean_kernel / sweep128h_boundary_kernel; the library itself picks it for long leaves only): leaves from a handful of rows (every chunk
    a boundary chunk, backup visits -> the exact path) to the whole table (hundreds of chunks per visit, many queries per leaf), top_k up to 64,
    ties, duplicates, all four simsimd-path keys, windows of several queries per leaf.  ZH_FUZZ_FUSED_SEEDS=first:count soaks."""
    import zebra_amd as za
    monkeypatch.setenv("ZH_S128H_FUSED", "1")
    if seed % 3 == 2:  # (kind 1 -- and the rounded tables below when they are non-negative -- are rows of BYTES: sweep128b_lean_kernel unless told otherwise)
        monkeypatch.setenv("ZH_S128H_BYTES", "0")
    rng = np.random.default_rng(99000 + seed)
    d = 128
    n = int(rng.integers(500, 30000))
    M = int(rng.choice([8, 30, 100, 400, 1500, 6000, 40000]))
    T = int(rng.choice([1, 4, 8, 15]))
    k = int(rng.choice([1, 3, 10, 20, 64]))
    B = int(rng.integers(1, 200))
    kind = int(rng.choice([0, 1, 1, 2]))
    X = zo.synth_rows(n, d, seed=seed, kind=kind)
    Q = zo.synth_queries(B, d, n, seed_rows=seed, kind=kind)
    if rng.random() < 0.4:
        X[rng.integers(0, n, 16)] = X[0]
    if rng.random() < 0.3:
        X = np.round(X).astype(np.float32)
        Q = np.round(Q).astype(np.float32)
    if rng.random() < 0.3:
        Q[0] = X[int(rng.integers(0, n))]
    if rng.random() < 0.2:
        Q[B - 1] = 0.0
    f = zo.Forest.build(X, M, T, seed=seed)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), seed=seed)
    ix.add(X)
    ix.set_sweep_mode("leaf-half")
    ix.set_hash_mode("dense")
    fused = 0
    for m, om, omode in ((za.L2SquaredDistance(), zo.L2SQ, 0), (za.L2Distance(), zo.L2, 0), (za.CosineDistance(True), zo.COSINE, zo.PARITY),
                         (za.CosineDistance(False), zo.COSINE, zo.CORRECTED)):
        ids, keys, counts = ix.search_batch(Q, k, m)
        st = ix.stats()
        fused += st["approx_fused"]
        oi, ok, oc = f.search_batch(Q, k, om, omode)
        assert (counts == oc).all(), (seed, n, M, T, k, B, om, omode)
        for b in range(B):
            c = int(oc[b])
            assert (keys[b, :c] == ok[b, :c]).all() and (ids[b, :c] == oi[b, :c]).all(), (seed, n, M, T, k, B, om, omode, b, st)
    assert fused >= 1 or ix.stats()["approx_scan"] != 3, (seed, ix.stats())
    ix.close()
