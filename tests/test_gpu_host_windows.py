"""zh_search_batch with a LARGE host-resident batch (what Database::query_vectors reaches through the shim's search_batch,
/root/reference/src/database/core.rs:290-313): the call cuts the batch into windows that alternate between two contexts, copies beside the
kernels (zh_api.hip, search_host_windows).  Results must be those of one batch -- the oracle's, bit for bit -- whatever the
window size, including a last window that is shorter, windows of one query, and every sweep the windows may choose."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import zebra_oracle as zo  # noqa: E402  (the checker)


@pytest.fixture(scope="module")
def za():
    import zebra_amd
    return zebra_amd


def same(ids, keys, counts, oi, ok, oc):
    assert (counts == oc).all()
    for b in range(oc.shape[0]):
        c = int(oc[b])
        assert (ids[b, :c] == oi[b, :c]).all() and (keys[b, :c] == ok[b, :c]).all(), b


@pytest.mark.parametrize("ahead", ["0", "1", "2"])
@pytest.mark.parametrize("d,M,T,k,mode", [(768, 300, 15, 100, "auto"), (384, 256, 15, 10, "approx"), (128, 200, 8, 10, "leaf-half"), (96, 64, 6, 7, "auto")])
def test_windows_equal_one_batch(za, monkeypatch, d, M, T, k, mode, ahead):
    # ahead = 1 / 2 (ZH_HOST_LOOKAHEAD; 2 = what the library does by itself where the walk wanders): three / four contexts, windows w + 1 (and w + 2) begun before w is finished
    monkeypatch.setenv("ZH_HOST_LOOKAHEAD", ahead)
    n, B = 9000, 700
    kind = 1 if d == 128 else 0
    X = zo.synth_rows(n, d, kind=kind)
    Q = zo.synth_queries(B, d, n, kind=kind)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    ix.set_sweep_mode(mode)
    cases = [(za.L2Distance(), zo.L2, 0), (za.CosineDistance(parity=True), zo.COSINE, zo.PARITY)]
    oracle = [f.search_batch(Q, k, om, omode) for _, om, omode in cases]
    # the first batch of an index goes the classic way (the windows want the visits per pair of earlier batches)
    ids, keys, counts = ix.search_batch(Q[:40], k, cases[0][0])
    same(ids, keys, counts, *[a[:40] for a in oracle[0]])
    assert ix.stats()["host_window_calls_accum"] == 0
    for wq in (64, 100, 333, 1):
        if wq == 1 and d != 96:
            continue
        monkeypatch.setenv("ZH_HOST_WINDOW", str(wq))
        for (m, om, omode), want in zip(cases, oracle):
            nq = B if wq > 1 else 9  # (nine windows of one query)
            before = ix.stats()["host_window_calls_accum"]
            ids, keys, counts = ix.search_batch(Q[:nq], k, m)
            same(ids, keys, counts, *[a[:nq] for a in want])
            assert ix.stats()["host_window_calls_accum"] == before + 1, (wq, om)
    # a batch below one and a half windows, and the switch: the classic path, same answers
    monkeypatch.setenv("ZH_HOST_WINDOW", "512")
    before = ix.stats()["host_window_calls_accum"]
    ids, keys, counts = ix.search_batch(Q, k, cases[0][0])
    same(ids, keys, counts, *oracle[0])
    monkeypatch.setenv("ZH_HOST_WINDOW", "64")
    monkeypatch.setenv("ZH_NO_HOST_WINDOWS", "1")
    ids, keys, counts = ix.search_batch(Q, k, cases[0][0])
    same(ids, keys, counts, *oracle[0])
    assert ix.stats()["host_window_calls_accum"] == before
    ix.close()


def test_small_leaf_forests_run_as_windows_too(za, monkeypatch):
    """the reference's default options (thousands of leaf visits per pair; row-score hash, blocked walk, prefilter): a large host batch runs as windows
    sized for the score table (round 5, second session: the classic path's one-chunk-at-a-time split reached a quarter of the pipelined rate)"""
    n, d, M, T, k, B = 4000, 64, 5, 15, 10, 300
    X = zo.synth_rows(n, d)
    Q = zo.synth_queries(B, d, n)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    m, om = za.L2SquaredDistance(), zo.L2SQ
    want = f.search_batch(Q, k, om, 0)
    # the FIRST batch of an index whose leaves are smaller than top_k already runs as windows (the visits per pair are guessed from the options)
    monkeypatch.setenv("ZH_HOST_WINDOW", "128")
    ids, keys, counts = ix.search_batch(Q, k, m)
    same(ids, keys, counts, *want)
    assert ix.stats()["host_window_calls_accum"] == 1
    ix.stats(reset=True)
    for wq in (64, 100, 7):
        monkeypatch.setenv("ZH_HOST_WINDOW", str(wq))
        before = ix.stats()["host_window_calls_accum"]
        for _ in range(2):
            ids, keys, counts = ix.search_batch(Q, k, m)
            same(ids, keys, counts, *want)
        assert ix.stats()["host_window_calls_accum"] == before + 2, wq
    monkeypatch.delenv("ZH_HOST_WINDOW")   # by regime: 1024 queries per window at 4000 rows -> 300 queries are one classic batch
    before = ix.stats()["host_window_calls_accum"]
    ids, keys, counts = ix.search_batch(Q, k, m)
    same(ids, keys, counts, *want)
    assert ix.stats()["host_window_calls_accum"] == before
    monkeypatch.setenv("ZH_HOST_WINDOW", "64")
    monkeypatch.setenv("ZH_NO_HOST_WINDOWS", "1")
    ids, keys, counts = ix.search_batch(Q, k, m)
    same(ids, keys, counts, *want)
    assert ix.stats()["host_window_calls_accum"] == before
    ix.close()


def test_concurrent_callers_and_large_batches_mix(za, monkeypatch):
    """single-query callers (combined rounds) and a large batch (windows) on one index at the same time: the lane serves them one after another"""
    import threading
    n, d, M, T, k = 8000, 256, 200, 8, 10
    X = zo.synth_rows(n, d)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    m, om = za.L2Distance(), zo.L2
    Qbig = zo.synth_queries(600, d, n)
    Qs = zo.synth_queries(64, d, n, b0=5000)
    want_big = f.search_batch(Qbig, k, om, 0)
    want_s = f.search_batch(Qs, k, om, 0)
    ix.search_batch(Qbig[:8], k, m)
    monkeypatch.setenv("ZH_HOST_WINDOW", "128")
    errs = []

    def small(i):
        try:
            for _ in range(5):
                ids, keys, counts = ix.search_batch(Qs[i:i + 1], k, m)
                same(ids, keys, counts, *[a[i:i + 1] for a in want_s])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    def big():
        try:
            for _ in range(3):
                ids, keys, counts = ix.search_batch(Qbig, k, m)
                same(ids, keys, counts, *want_big)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=small, args=(i,)) for i in range(16)] + [threading.Thread(target=big)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs[:2]
    assert ix.stats()["host_window_calls_accum"] >= 1
    ix.close()
