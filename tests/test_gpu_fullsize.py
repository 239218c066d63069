"""BASELINE.json's configurations at FULL size on one MI355X, checked through size-independent properties
(the oracle cannot run 10M x 768 in seconds):

  * result shape: counts == k, ids unique, ascending by (key, id);
  * keys are exactly the oracle's Metric::distance of the returned rows (rows regenerated on the CPU from the
    counter generator -- any row can be regenerated anywhere);
  * every returned id belongs to one of the T leaves the query hashes to (bucket membership), checked by
    descending the exported forest with the oracle's point_is_above;
  * idempotence and batch-split invariance: the same queries alone, in two halves or in one batch give the
    same answers;
  * dense-level invariance: hashing 0 levels or every level with the MFMA kernel changes nothing;
  * the planted neighbour (query = stored row + 0.3 noise) comes back for most queries.
cfg4 / cfg5 are 8-GPU configurations: their per-GPU shards (12.5M x 768, and 125M x 128 reduced to what one
test run can build in reasonable time) are exercised here."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import zebra_oracle as zo  # noqa: E402


def _descend(g, q, root):
    n = int(root)
    while g["plane"][n] >= 0:
        p = g["plane"][n]
        n = int(g["right"][n] if zo.point_is_above(g["planes"][p], g["consts"][p], q) else g["left"][n])
    return n


def _check(za, n, d, metric_name, k, B, M, T, kind=0, n_check=6, planted_min=0.5, rows0=0):
    met = {"cos": (za.CosineDistance(parity=False), zo.COSINE, zo.CORRECTED),
           "cos_parity": (za.CosineDistance(parity=True), zo.COSINE, zo.PARITY),
           "l2": (za.L2Distance(), zo.L2, 0), "l2sq": (za.L2SquaredDistance(), zo.L2SQ, 0)}
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), reserve_rows=n)
    ix.append_synthetic(n, first_row=rows0, kind=kind)
    ix.build()
    assert len(ix) == n and not ix.is_empty()
    Q = zo.synth_queries(B, d, n, kind=kind)
    m, om, omode = met[metric_name]
    ids, keys, counts = ix.search_batch(Q, k, m)
    assert (counts == k).all()
    # ascending by (key, id), unique ids
    for b in range(B):
        pairs = list(zip(keys[b].tolist(), ids[b].tolist()))
        assert pairs == sorted(pairs) and len(set(ids[b].tolist())) == k
    # keys == oracle distance of the regenerated rows; ids inside the hashed leaves
    g = ix.get_forest()
    leaf_sets = {}
    for b in range(n_check):
        rows = np.stack([zo.synth_rows(1, d, row0=rows0 + int(i), kind=kind)[0] for i in ids[b]])
        assert (zo.distance_batch(om, omode, rows, Q[b]) == keys[b]).all()
        members = set()
        for t in range(T):
            leaf = _descend(g, Q[b], g["roots"][t])
            off, ln = int(np.uint32(g["left"][leaf])), int(g["right"][leaf])
            assert ln >= k, "test assumes the one-leaf-per-tree regime"
            members |= set(g["leaf_ids"][off:off + ln].tolist())
        assert set(ids[b].tolist()) <= members
        leaf_sets[b] = members
    # the answer is exactly the k best of those leaves (brute force over the union with the oracle's keys)
    b = 0
    mem = np.array(sorted(leaf_sets[b]), dtype=np.int64)
    rows = np.stack([zo.synth_rows(1, d, row0=rows0 + int(i), kind=kind)[0] for i in mem[:3000]]) if len(mem) > 3000 else \
        np.stack([zo.synth_rows(1, d, row0=rows0 + int(i), kind=kind)[0] for i in mem])
    if len(mem) <= 3000:
        kk = zo.distance_batch(om, omode, rows, Q[b])
        want = sorted(zip(kk.tolist(), mem.tolist()))[:k]
        assert want == list(zip(keys[b].tolist(), ids[b].tolist()))
    # idempotence, batch-split invariance
    ids2, keys2, _ = ix.search_batch(Q, k, m)
    assert (ids2 == ids).all() and (keys2 == keys).all()
    h = B // 2
    ia, ka, _ = ix.search_batch(Q[:h], k, m)
    ib, kb, _ = ix.search_batch(Q[h:], k, m)
    assert (np.concatenate([ia, ib]) == ids).all() and (np.concatenate([ka, kb]) == keys).all()
    i1, k1, _ = ix.search_batch(Q[3:4], k, m)
    assert (i1[0] == ids[3]).all() and (k1[0] == keys[3]).all()
    # dense-level invariance
    for lv in (0, 100):
        ix.set_dense_levels(lv)
        i3, k3, _ = ix.search_batch(Q[:64], k, m)
        assert (i3 == ids[:64]).all() and (k3 == keys[:64]).all()
    ix.set_dense_levels(-1)
    # planted neighbours
    planted = np.array([zo.synth_query_row(b, n) for b in range(B)], dtype=np.uint64)
    hit = float((ids == planted[:, None]).any(1).mean())
    assert hit >= planted_min, hit
    st = ix.stats()
    assert st["rows_scored"] >= B * T * k and st["rows_swept"] <= st["rows_scored"]
    ix.close()
    return hit


def test_cfg2_1m_384_cosine_top10_batch256(za=None):
    import zebra_amd as za
    # corrected key finds the planted neighbour; the reference's literal key returns the LEAST similar rows (F4)
    _check(za, 1_000_000, 384, "cos", 10, 256, 1024, 15, planted_min=0.3)
    ix = za.LSHIndex(384, za.LSHIndexOptions(1024, 15), reserve_rows=1_000_000)
    ix.append_synthetic(1_000_000)
    ix.build()
    Q = zo.synth_queries(256, 384, 1_000_000)
    ids, keys, counts = ix.search_batch(Q, 10, za.CosineDistance(parity=True))
    planted = np.array([zo.synth_query_row(b, 1_000_000) for b in range(256)], dtype=np.uint64)
    assert not (ids == planted[:, None]).any()
    assert (counts == 10).all() and (np.diff(keys.astype(np.float64), axis=1) >= 0).all()


def test_cfg3_10m_768_l2_top100_batch1024():
    import zebra_amd as za
    _check(za, 10_000_000, 768, "l2", 100, 1024, 4096, 15, planted_min=0.6, n_check=3)


def test_cfg4_shard_12p5m_768_cosine_top10_batch1024():
    import zebra_amd as za
    # rank 3 of 8: rows [37.5M, 50M) of the 100M set, per-shard options 4096 / 15
    _check(za, 12_500_000, 768, "cos_parity", 10, 1024, 4096, 15, planted_min=0.0, n_check=3, rows0=37_500_000)


def test_cfg5_shard_slice_128d_sift_l2_top10_batch4096():
    import zebra_amd as za
    # a 20M-row slice of one cfg5 shard (125M x 128): integer-valued rows -> exact L2
    _check(za, 20_000_000, 128, "l2", 10, 4096, 8192, 15, kind=1, planted_min=0.3, n_check=3)


def test_reference_default_options_at_batch_size_more_than_2_24_visits():
    """lsh.rs:134-135 defaults (max_node_size 5, 15 trees) on 1M rows with a batch of 256: the walk wanders over a good
    part of every tree -- 17.5M leaf visits in one batch, more than 2^24, past the inline visits and the
    first visit-log pool, through the 16-lane-group selection.  First, middle and last queries against the oracle."""
    import zebra_amd as za
    n, d, B, k = 1_000_000, 64, 256, 10
    X = zo.synth_rows(n, d)
    Q = zo.synth_queries(B, d, n)
    ix = za.LSHIndex(d, za.LSHIndexOptions(5, 15))
    ix.add(X)
    m = za.L2SquaredDistance()
    ids, keys, counts = ix.search_batch(Q, k, m)      # visit log overflows: the emit walk runs
    ids2, keys2, counts2 = ix.search_batch(Q, k, m)   # grown log: the flat expansion runs
    st = ix.stats()
    assert st["visits"] > (1 << 24)
    assert (ids == ids2).all() and (keys == keys2).all() and (counts == counts2).all()
    f = zo.Forest.from_arrays(X, 5, ix.get_forest())
    for b in (0, 1, B // 2, B - 2, B - 1):
        oi, ok = f.search(Q[b], k, zo.L2SQ)
        assert (ids[b] == oi).all() and (keys[b] == ok).all()
