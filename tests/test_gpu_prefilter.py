"""The prefilter (zh_set_sweep_mode 0 / 3 on a batch hashed from row scores; zebra_amd/csrc/zh_search.hip "Prefilter"): which rows a
visited leaf hands over (lsh.rs:300-330) and which of them can still be among the k nearest is judged on the row scores with a
rigorous rounding bound; only the survivors and the visits the bound cannot decide are scored with the reference's arithmetic.
Ids, keys and counts must stay bit-identical to the oracle and to the sweep."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import zebra_oracle as zo  # noqa: E402  (the checker)


@pytest.fixture(scope="module")
def za():
    import zebra_amd
    return zebra_amd


def _metric(za, name):
    return {"l2sq": (za.L2SquaredDistance(), zo.L2SQ, 0), "l2": (za.L2Distance(), zo.L2, 0),
            "cos_parity": (za.CosineDistance(parity=True), zo.COSINE, zo.PARITY),
            "cos": (za.CosineDistance(parity=False), zo.COSINE, zo.CORRECTED)}[name]


def _same(ids, keys, counts, oi, ok, oc, tag):
    assert (counts == oc).all(), tag
    for b in range(len(oc)):
        c = int(oc[b])
        assert (ids[b, :c] == oi[b, :c]).all() and (keys[b, :c] == ok[b, :c]).all(), (tag, b)


def _index(za, X, M, T):
    ix = za.LSHIndex(X.shape[1], za.LSHIndexOptions(M, T))
    ix.add(X)
    ix.set_dense_levels(100)  # every sign precomputed, as the library chooses by itself for wandering walks
    ix.set_hash_mode("scores")
    return ix


@pytest.mark.parametrize("n,d,M,T,k,B,kind,metric", [
    (30000, 384, 5, 6, 10, 32, 0, "l2sq"),        # the reference's default leaf size
    (30000, 384, 5, 6, 10, 32, 0, "cos_parity"),  # ... with the metric of its image / audio databases
    (20000, 256, 5, 4, 10, 16, 2, "l2"),          # clustered rows: the rows of a leaf are neighbours of each other
    (20000, 256, 5, 4, 10, 16, 2, "cos"),
    (8000, 768, 8, 3, 40, 8, 0, "l2sq"),          # leaves of 8 rows, k = 40
    (15000, 128, 5, 5, 10, 24, 1, "l2sq"),        # integer-valued rows: exact ties between keys
    (15000, 128, 5, 5, 64, 24, 1, "cos_parity"),  # the largest k the prefilter serves
    (6000, 100, 4, 3, 5, 12, 0, "l2"),            # d not a multiple of 4
    (12000, 384, 5, 4, 10, 7, 0, "l2sq"),         # a batch padded to a multiple of four
    (12000, 64, 5, 4, 3, 1, 0, "cos"),            # a single query
])
def test_prefilter_equals_oracle_and_sweep(za, n, d, M, T, k, B, kind, metric):
    X = zo.synth_rows(n, d, kind=kind)
    Q = zo.synth_queries(B, d, n, kind=kind)
    ix = _index(za, X, M, T)
    f = zo.Forest.from_arrays(X, M, ix.get_forest())
    m, om, omode = _metric(za, metric)
    oi, ok, oc = f.search_batch(Q, k, om, omode)
    ix.search_batch(Q, k, m)  # (an index's first wandering batch may outgrow the visit log: that batch walks twice and is swept)
    for mode in ("prefilter", "leaf", "auto"):
        ix.set_sweep_mode(mode)
        ids, keys, counts = ix.search_batch(Q, k, m)
        st = ix.stats()
        assert st["hash_from_scores"] == 1
        assert st["prefiltered"] == (0 if mode == "leaf" else 1), (mode, st["prefiltered"], st["prefilter_last_overflow"], st["prefilter_exact_visits"])
        if mode != "leaf":  # the bound decides nearly everything: a small share of the scored rows takes the exact path
            assert 0 < st["prefilter_exact_rows"] < 0.5 * st["rows_scored"] + 64 * B * T, (st["prefilter_exact_rows"], st["rows_scored"])
        _same(ids, keys, counts, oi, ok, oc, mode)
    assert ix.stats()["prefilter_fallbacks_accum"] == 0
    ix.close()


def test_prefilter_with_an_id_base(za):
    """a shard's ids are id_base + row: the exact-key kernel adds it, as the final kernel of the sweep path does"""
    n, d, M, T, k, B, base = 12000, 96, 5, 4, 10, 9, 5_000_000_000
    X = zo.synth_rows(n, d)
    Q = zo.synth_queries(B, d, n)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), id_base=base)
    ix.add(X)
    ix.set_dense_levels(100)
    ix.set_hash_mode("scores")
    f = zo.Forest.from_arrays(X, M, ix.get_forest())
    oi, ok, oc = f.search_batch(Q, k, zo.L2SQ, 0)
    for mode in ("auto", "auto", "leaf"):
        ix.set_sweep_mode(mode)
        ids, keys, counts = ix.search_batch(Q, k, za.L2SquaredDistance())
        assert ix.stats()["prefiltered"] == (0 if mode == "leaf" else 1)
        assert (counts == oc).all()
        for b in range(B):
            c = int(oc[b])
            assert (ids[b, :c] == oi[b, :c] + np.uint64(base)).all() and (keys[b, :c] == ok[b, :c]).all(), (mode, b)
    ix.close()


def test_a_forest_with_long_leaves(za):
    """clustered rows at d = 64 with max_node_size 3: splits between near-identical rows leave leaves of up to ~40 rows; every visit to
    one takes the exact path (the per-lane selection handles 8 rows), thresholded like the rest"""
    n, d, M, T, k, B = 20000, 64, 3, 4, 10, 16
    X = zo.synth_rows(n, d, kind=2)
    Q = zo.synth_queries(B, d, n, kind=2)
    ix = _index(za, X, M, T)
    fo = ix.get_forest()
    assert fo["right"][fo["plane"] < 0].max() > 8
    f = zo.Forest.from_arrays(X, M, fo)
    for metric in ("l2", "cos"):
        m, om, omode = _metric(za, metric)
        oi, ok, oc = f.search_batch(Q, k, om, omode)
        for _ in range(3):
            ids, keys, counts = ix.search_batch(Q, k, m)
            _same(ids, keys, counts, oi, ok, oc, metric)
        st = ix.stats()
        assert st["prefiltered"] == 1 and st["prefilter_exact_visits"] > 0, st
    ix.close()


def test_prefilter_on_adversarial_rows(za):
    """duplicates (every key of a leaf equal), huge and tiny magnitudes, zero rows (simsimd's special cases), near-duplicates:
    whatever the bound cannot decide goes to the exact path or, when a list runs over, to the sweep"""
    rng = np.random.default_rng(7)
    n, d, M, T, k, B = 9000, 128, 4, 4, 10, 16
    X = zo.synth_rows(n, d)
    X[1000:1004] = X[0]                                   # copies of one row (a leaf of equal keys)
    X[2000:2200] *= np.float32(1e18)                      # |r|^2 overflows f32
    X[3000:3200] *= np.float32(1e-30)                     # subnormal squares
    X[3500:3510] = 0                                      # zero rows
    X[5000:5040] *= np.float32(3e18)                      # dot products with a large query overflow as well
    X[5040:5050, ::2] *= np.float32(-1)                   # ... with cancelling signs: inf - inf in one summation order, not in another
    X[4000:4300] = X[4000] + (rng.random((300, d)) < 0.01).astype(np.float32) * np.float32(1e-3)  # near-duplicates
    Q = np.concatenate([zo.synth_queries(B - 6, d, n), X[[0, 2000, 3000, 3500, 4000]], X[[5045]] * np.float32(1e15)])
    ix = _index(za, X, M, T)
    f = zo.Forest.from_arrays(X, M, ix.get_forest())
    for metric in ("l2sq", "cos_parity", "cos", "l2"):
        m, om, omode = _metric(za, metric)
        oi, ok, oc = f.search_batch(Q, k, om, omode)
        for mode in ("prefilter", "leaf"):
            ix.set_sweep_mode(mode)
            ids, keys, counts = ix.search_batch(Q, k, m)
            _same(ids, keys, counts, oi, ok, oc, (metric, mode))
    ix.close()


def test_a_list_that_runs_over_sends_the_batch_to_the_sweep(za):
    n, d, M, T, k, B = 20000, 128, 5, 5, 10, 16
    X = zo.synth_rows(n, d)
    Q = zo.synth_queries(B, d, n)
    ix = _index(za, X, M, T)
    f = zo.Forest.from_arrays(X, M, ix.get_forest())
    oi, ok, oc = f.search_batch(Q, k, zo.L2SQ, 0)
    m = za.L2SquaredDistance()
    ix.search_batch(Q, k, m)
    ix.stats(reset=True)
    os.environ["ZH_PREFILTER_CAP"] = "3"  # fewer slots than k: every list runs over
    try:
        seen = []
        for _ in range(4):
            ids, keys, counts = ix.search_batch(Q, k, m)
            _same(ids, keys, counts, oi, ok, oc, "fallback")
            st = ix.stats()
            seen.append((st["prefiltered"], st["prefilter_fallbacks_accum"], st["prefilter_last_overflow"]))
    finally:
        del os.environ["ZH_PREFILTER_CAP"]
    # two batches in a row handed back to the sweep: the index stops trying until its trees change (no third attempt)
    assert [s_[:2] for s_ in seen] == [(0, 1), (0, 2), (0, 2), (0, 2)] and seen[0][2] & 1, seen
    ix.add(zo.synth_rows(64, d, row0=n))  # the trees change: the prefilter is tried again, with lists that fit
    X2 = np.concatenate([X, zo.synth_rows(64, d, row0=n)])
    f.insert(X2, n)
    ix.set_dense_levels(-1)
    for _ in range(3):
        ids, keys, counts = ix.search_batch(Q, k, m)
    assert ix.stats()["prefiltered"] == 1, ix.stats()
    _same(ids, keys, counts, *f.search_batch(Q, k, zo.L2SQ, 0), "after the trees changed")
    ix.close()


def test_prefilter_follows_inserts_and_removals(za):
    n0, n1, d, M, T, k, B = 12000, 3000, 256, 5, 4, 10, 16
    X = zo.synth_rows(n0 + n1, d)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.set_hash_mode("scores")
    ix.add(X[:n0])
    f = zo.Forest.build(X[:n0], M, T)
    Q = zo.synth_queries(B, d, n0 + n1)
    m = za.L2SquaredDistance()

    def twice(tag):  # the library hashes every plane once it has seen the walks wander; the second batch is the steady state
        ix.search_batch(Q, k, m)
        ids, keys, counts = ix.search_batch(Q, k, m)
        st = ix.stats()
        assert st["hash_from_scores"] == 1 and st["prefiltered"] == 1, (tag, st)
        _same(ids, keys, counts, *f.search_batch(Q, k, zo.L2SQ, 0), tag)
        return ids

    twice("built")
    ix.add(X[n0:])  # leaves split: the per-slot norms are rebuilt with the trees
    f.insert(X, n0)
    ids = twice("grown")
    gone = np.unique(ids[:, :3][ids[:, :3] != np.uint64(0xFFFFFFFFFFFFFFFF)]).astype(np.uint64)
    ix.remove(gone)
    f.remove(gone)
    ids = twice("after removals")
    assert not np.isin(ids, gone).any()
    ix.close()


def test_prefilter_in_pipelined_windows(za):
    """two contexts in flight, windows of two batches: the prefiltered second half runs on the contexts' own streams; then the same
    windows with lists that run over: zh_search_wait redoes the WINDOW with the sweep and hands out the same per-batch results"""
    import torch
    n, d, M, T, k, B = 30000, 128, 5, 6, 10, 32
    X = zo.synth_rows(n, d)
    ix = _index(za, X, M, T)
    f = zo.Forest.from_arrays(X, M, ix.get_forest())
    dev = torch.device("cuda", 0)
    Qs = [zo.synth_queries(B, d, n, b0=j * B) for j in range(4)]
    want = [f.search_batch(q, k, zo.L2SQ, 0) for q in Qs]
    dq = [torch.from_numpy(q).to(dev) for q in Qs]
    m = za.L2SquaredDistance()
    ctxs = [ix.search_context(), ix.search_context()]

    def run():
        outs = [(torch.zeros((B, k), dtype=torch.int64, device=dev), torch.zeros((B, k), dtype=torch.int64, device=dev),
                 torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(4)]
        torch.cuda.synchronize()
        for w in range(2):
            ctxs[w].begin_window([dq[2 * w].data_ptr(), dq[2 * w + 1].data_ptr()], B, k, m)
        for w in range(2):
            o = outs[2 * w:2 * w + 2]
            ctxs[w].finish_window([x[0].data_ptr() for x in o], [x[1].data_ptr() for x in o], [x[2].data_ptr() for x in o])
        for c in ctxs:
            c.wait()
        for j in range(4):
            _same(outs[j][0].cpu().numpy().view(np.uint64), outs[j][1].cpu().numpy().view(np.uint64), outs[j][2].cpu().numpy().view(np.uint32),
                  *want[j], j)

    run()
    run()  # (a context's first wandering window may outgrow its visit log and be swept)
    assert ix.stats()["prefiltered"] == 1
    ix.stats(reset=True)
    os.environ["ZH_PREFILTER_CAP"] = "2"
    try:
        run()
    finally:
        del os.environ["ZH_PREFILTER_CAP"]
    st = ix.stats()
    assert st["prefilter_fallbacks_accum"] == 2 and st["prefiltered"] == 0, st
    for c in ctxs:
        c.close()
    ix.close()
