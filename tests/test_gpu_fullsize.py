"""Every BASELINE.json configuration at FULL size on one MI355X, compared EXACTLY with the oracle.

The oracle cannot hold 10M x 768 rows (31 GB) or build their forest in seconds, but the stored rows are a counter
generator: zo_search_batch_synth walks the forest exported by the HIP build (zh_index_get_forest) and regenerates
every row it scores, so ids / keys / counts of whole queries are compared bit for bit at the real N
(lsh.rs:290-348, 544-565).  The forest itself -- built on the GPU, too large for zo_forest_build -- is checked against
the build rules by zo_check_forest_synth: every tree's leaves partition the rows, the max_node_size rule holds at
every node, sampled rows sit in the leaves they hash to, and every hyperplane on their paths is bit-identical to
make_hyperplane of the node's sample pair (lsh.rs:192-267, 411-429).

  cfg1  10k x 384 cosine top-10, ONE query, reference default options  -- whole oracle (its own build + search)
  cfg2  1M x 384 cosine top-10, batch 256
  cfg3  10M x 768 L2 top-100, batch 1024
  cfg4  100M x 768 cosine top-10, batch 1024: the 8 shards one after another on the one GPU (build, search, keep the
        packed result, destroy), then zh_merge_topk_packed_device == the oracle's 8-shard merge
  cfg5  1B x 128 SIFT-style L2 top-10, batch 4096: the 8 shards of 125M rows one after another + merge, as cfg4; and one
        full shard through the whole property list
plus the size-independent properties: result shape, idempotence, batch-split and dense-level invariance, planted
neighbours."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import zebra_oracle as zo  # noqa: E402

SEED_INDEX = zo.SEED_INDEX


def _metrics(za):
    return {"cos": (za.CosineDistance(parity=False), zo.COSINE, zo.CORRECTED),
            "cos_parity": (za.CosineDistance(parity=True), zo.COSINE, zo.PARITY),
            "l2": (za.L2Distance(), zo.L2, 0), "l2sq": (za.L2SquaredDistance(), zo.L2SQ, 0)}


def _exact_vs_oracle(ix, n, d, M, Q, sel, k, om, omode, ids, keys, counts, rows0, kind, index_seed=SEED_INDEX,
                     n_sample=24):
    """forest rules + exact ids / keys / counts of the selected queries, rows regenerated on demand"""
    g = ix.get_forest()
    f = zo.Forest.borrow_synth(n, d, M, g, first_row=rows0, kind=kind)
    rc, planes = f.check_synth(index_seed=index_seed, n_sample=n_sample)
    assert rc == 0, f"forest check failed with code {rc}"
    assert planes >= n_sample * len(g["roots"])
    oi, ok, oc, st = f.search_batch_synth(Q[sel], k, om, omode, stats=True)
    base = np.uint64(ix.id_base)
    for j, b in enumerate(sel):
        c = int(oc[j])
        assert counts[b] == c, (b, counts[b], c)
        assert (ids[b, :c] == oi[j, :c] + base).all(), f"ids of query {b} differ from the oracle's"
        assert (keys[b, :c] == ok[j, :c]).all(), f"keys of query {b} differ from the oracle's"
    return oi + base, ok, oc, st


def _check(za, n, d, metric_name, k, B, M, T, kind=0, n_exact=8, planted_min=0.5, rows0=0, n_total=None):
    """n rows [rows0, rows0 + n) of a set of n_total rows (a shard when n_total > n; ids are global)"""
    n_total = n_total or n
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), id_base=rows0, reserve_rows=n)
    ix.append_synthetic(n, first_row=rows0, kind=kind)
    ix.build()
    assert len(ix) == n and not ix.is_empty()
    Q = zo.synth_queries(B, d, n_total, kind=kind)
    m, om, omode = _metrics(za)[metric_name]
    ids, keys, counts = ix.search_batch(Q, k, m)
    assert (counts == k).all()
    for b in range(B):  # ascending by (key, id), unique ids
        pairs = list(zip(keys[b].tolist(), ids[b].tolist()))
        assert pairs == sorted(pairs) and len(set(ids[b].tolist())) == k
    sel = np.unique(np.linspace(0, B - 1, n_exact).astype(int))
    st = _exact_vs_oracle(ix, n, d, M, Q, sel, k, om, omode, ids, keys, counts, rows0, kind)[3]
    assert st.rows_scored >= len(sel) * T * k
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
    planted = np.array([zo.synth_query_row(b, n_total) for b in range(B)], dtype=np.uint64)
    here = (planted >= rows0) & (planted < rows0 + n)  # queries whose planted row lives in this shard
    assert here.sum() >= B // 16
    hit = float((ids[here] == planted[here, None]).any(1).mean())
    assert hit >= planted_min, hit
    assert not (ids[~here] == planted[~here, None]).any()
    s = ix.stats()
    # leaf-major sweep: rows shared inside a group are loaded once; table scan: every stored row is streamed once
    assert s["rows_scored"] >= B * T * k and (s["rows_swept"] == n if s["table_scan"] else s["rows_swept"] <= s["rows_scored"])
    ix.close()
    return hit


def test_cfg1_10k_384_cosine_top10_single_query_reference_defaults():
    """configs[0]: the reference's own CPU-runnable case, default options (lsh.rs:131-138) -> the wandering walk.
    The WHOLE oracle: its own forest build must equal the GPU's, and every query's answer must be equal."""
    import zebra_amd as za
    n, d, k = 10_000, 384, 10
    X = zo.synth_rows(n, d)
    ix = za.LSHIndex(d, za.LSHIndexOptions())  # max_node_size 5, num_trees 15
    ix.add(X)
    f = zo.Forest.build(X, 5, 15)
    assert zo.canonical_forest(ix.get_forest(), d) == zo.canonical_forest(f.arrays(), d)
    Q = zo.synth_queries(16, d, n)
    for m, om, omode in ((za.CosineDistance(parity=True), zo.COSINE, zo.PARITY), (za.CosineDistance(parity=False), zo.COSINE, zo.CORRECTED)):
        for b in range(16):  # single-query calls, as the config says
            got = ix.search(Q[b], k, m)
            oi, ok = f.search(Q[b], k, om, omode)
            assert [g_[0] for g_ in got] == oi.tolist() and [g_[1] for g_ in got] == ok.tolist()
        ids, keys, counts = ix.search_batch(Q, k, m)
        oi, ok, oc = f.search_batch(Q, k, om, omode)
        assert (counts == oc).all() and (ids == oi).all() and (keys == ok).all()
    # with the corrected key the planted neighbour is the nearest row; with the literal key it never comes back (F4)
    ids, _, _ = ix.search_batch(Q, k, za.CosineDistance(parity=False))
    planted = np.array([zo.synth_query_row(b, n) for b in range(16)], dtype=np.uint64)
    assert (ids[:, 0] == planted).mean() >= 0.9
    ix.close()


def test_cfg2_1m_384_cosine_top10_batch256():
    import zebra_amd as za
    # corrected key finds the planted neighbour; the reference's literal key returns the LEAST similar rows (F4)
    _check(za, 1_000_000, 384, "cos", 10, 256, 1024, 15, planted_min=0.3, n_exact=256)
    ix = za.LSHIndex(384, za.LSHIndexOptions(1024, 15), reserve_rows=1_000_000)
    ix.append_synthetic(1_000_000)
    ix.build()
    Q = zo.synth_queries(256, 384, 1_000_000)
    m, om, omode = _metrics(za)["cos_parity"]
    ids, keys, counts = ix.search_batch(Q, 10, m)
    planted = np.array([zo.synth_query_row(b, 1_000_000) for b in range(256)], dtype=np.uint64)
    assert not (ids == planted[:, None]).any()
    assert (counts == 10).all() and (np.diff(keys.astype(np.float64), axis=1) >= 0).all()
    _exact_vs_oracle(ix, 1_000_000, 384, 1024, Q, np.arange(0, 256, 37), 10, om, omode, ids, keys, counts, 0, 0)
    ix.close()


def test_cfg3_10m_768_l2_top100_batch1024():
    import zebra_amd as za
    _check(za, 10_000_000, 768, "l2", 100, 1024, 4096, 15, planted_min=0.6, n_exact=1024)


def _eight_shards_time_multiplexed(N, d, k, B, M, T, metric_name, kind):
    """An 8-GPU configuration END TO END on one GPU: rank r's shard (rows [N/8 r, N/8 (r+1)), its own forest,
    id_base = first row, seed + r exactly as bench.py builds it) is built, searched with the full batch and destroyed,
    one after another; the eight packed results are laid out as the all-gather would leave them and merged by
    zh_merge_topk_packed_device.  Checked: every shard's answers for 512 queries and its forest against the oracle;
    the merged answer of ALL queries against the oracle's merge (zo_merge_topk) of the eight device results; the
    merged answer of the 512 queries against the all-oracle pipeline (eight synth searches + merge)."""
    import torch
    import zebra_amd as za
    from zebra_amd import sharding
    S = 8
    m, om, omode = _metrics(za)[metric_name]
    Q = zo.synth_queries(B, d, N, kind=kind)
    # round 6: 512 queries of the batch (round 5: 8) against the oracle, shard by shard and merged -- the oracle's search is threaded; every query
    # (sel = np.arange(B)) passes too and takes 80 s (cfg4) / 183 s (cfg5) instead of 52 / 80: the GPU suite is kept near six minutes
    sel = np.unique(np.linspace(0, B - 1, 512).astype(int))
    W = za.packed_result_words(B, k)
    dev = torch.device("cuda", 0)
    g_packed = torch.empty((S, W), dtype=torch.int64, device=dev)
    dq = torch.from_numpy(Q).to(dev)
    o_ids = np.zeros((S, len(sel), k), np.uint64)
    o_keys = np.zeros((S, len(sel), k), np.uint64)
    o_counts = np.zeros((S, len(sel)), np.uint32)
    for r in range(S):
        first, n = sharding.shard_rows(N, S, r)
        ix = za.LSHIndex(d, za.LSHIndexOptions(sharding.per_shard_max_node_size(M * S, S, k), T), seed=SEED_INDEX + r,
                         id_base=first, reserve_rows=n)
        ix.append_synthetic(n, first_row=first, kind=kind)
        ix.build()
        p_ids, p_keys, p_counts = sharding.packed_views(torch, g_packed[r], B, k)
        ix.search_batch_device(dq.data_ptr(), B, k, m, p_ids.data_ptr(), p_keys.data_ptr(), p_counts.data_ptr())
        torch.cuda.synchronize()
        ids = p_ids.cpu().numpy().view(np.uint64)
        keys = p_keys.cpu().numpy().view(np.uint64)
        counts = p_counts.cpu().numpy().view(np.uint32)
        assert (counts == k).all() and ids.min() >= first and ids.max() < first + n
        o_ids[r], o_keys[r], o_counts[r], _ = _exact_vs_oracle(ix, n, d, M, Q, sel, k, om, omode, ids, keys, counts, first, kind,
                                                               index_seed=SEED_INDEX + r, n_sample=8)
        ix.close()
    m_ids = torch.empty((B, k), dtype=torch.int64, device=dev)
    m_keys = torch.empty((B, k), dtype=torch.int64, device=dev)
    m_counts = torch.empty(B, dtype=torch.int32, device=dev)
    za.merge_topk_packed_device(0, S, B, k, g_packed.data_ptr(), m_ids.data_ptr(), m_keys.data_ptr(), m_counts.data_ptr())
    torch.cuda.synchronize()
    got_i, got_k, got_c = (m_ids.cpu().numpy().view(np.uint64), m_keys.cpu().numpy().view(np.uint64),
                           m_counts.cpu().numpy().view(np.uint32))
    # (1) the merge kernel on all queries == the oracle's merge of the eight device results
    hp = g_packed.cpu().numpy()
    s_ids = np.stack([hp[r, :B * k].reshape(B, k) for r in range(S)]).view(np.uint64)
    s_keys = np.stack([hp[r, B * k:2 * B * k].reshape(B, k) for r in range(S)]).view(np.uint64)
    s_counts = np.stack([hp[r, 2 * B * k:].view(np.uint32)[:B] for r in range(S)])
    wi, wk, wc = zo.merge_topk(s_ids, s_keys, s_counts, k)
    assert (got_c == wc).all() and (got_i == wi).all() and (got_k == wk).all()
    # (2) the whole set's answer of the selected queries == the all-oracle pipeline
    ai, ak, ac = zo.merge_topk(o_ids, o_keys, o_counts, k)
    assert (got_c[sel] == ac).all() and (got_i[sel] == ai).all() and (got_k[sel] == ak).all()
    # the merged ids are global rows of the whole set, from more than one shard overall
    assert got_i.max() < N and len(np.unique(got_i // np.uint64(N // S))) > 1
    return got_i, got_k


def test_cfg4_100m_768_cosine_top10_batch1024_eight_shards_time_multiplexed():
    _eight_shards_time_multiplexed(100_000_000, 768, 10, 1024, 4096, 15, "cos_parity", 0)


def test_cfg5_1b_128d_sift_l2_top10_batch4096_eight_shards_time_multiplexed():
    """the 1B-row answer formed and compared -- all eight 125M x 128 shards (64 GB each) one after another,
    merged, against the oracle's eight-shard merge.  Integer-valued rows: every L2^2 is exact in f32 under any order."""
    ids, keys = _eight_shards_time_multiplexed(1_000_000_000, 128, 10, 4096, 8192, 15, "l2", 1)
    assert (np.diff(keys.astype(np.float64), axis=1) >= 0).all()


def test_cfg5_full_shard_125m_128d_sift_l2_top10_batch4096():
    import zebra_amd as za
    # one whole shard of the 1B set (rank 5 of 8: rows [625M, 750M)): 64 GB of integer-valued rows -> exact L2
    _check(za, 125_000_000, 128, "l2", 10, 4096, 8192, 15, kind=1, planted_min=0.3, n_exact=4096, rows0=625_000_000,
           n_total=1_000_000_000)


def _every_sweep_agrees(za, n, d, metric_name, k, B, M, T, modes, kind=0, rows0=0, n_total=None):
    """one full-size index, the SAME batch under every sweep kind that serves it: ids, keys and counts of ALL queries must be equal bit for
    bit (the default kind is compared with the oracle in the tests above; this ties every other kernel to it at full size)"""
    n_total = n_total or n
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), id_base=rows0, reserve_rows=n)
    ix.append_synthetic(n, first_row=rows0, kind=kind)
    ix.build()
    Q = zo.synth_queries(B, d, n_total, kind=kind)
    m = _metrics(za)[metric_name][0]
    ref, seen = None, {}
    for mode, code in modes:
        ix.set_sweep_mode(mode)
        got = ix.search_batch(Q, k, m)
        st = ix.stats()
        assert st["approx_scan"] == code, (mode, st["approx_scan"], st["table_scan"])
        seen[mode] = st["rows_swept"]
        if ref is None:
            ref = got
        else:
            assert all((a == b).all() for a, b in zip(ref, got)), mode
    ix.close()
    return seen


def test_cfg3_full_size_every_scan_kernel_agrees():
    """10M x 768, L2 top-100, batch 1024: the matrix-core scan (fp16 rows and queries), the VALU half-width scan (f32 rows), the f32 scan and the
    leaf-major sweep return the same 1024 x 100 ids and keys"""
    import zebra_amd as za
    seen = _every_sweep_agrees(za, 10_000_000, 768, "l2", 100, 1024, 4096, 15, [("approx", 2), ("approx-valu", 1), ("scan", 0), ("leaf", 0)])
    assert seen["approx"] == seen["scan"] == 10_000_000  # (the table read once per batch)


def test_cfg4_shard_full_size_every_scan_kernel_agrees():
    """one of 8 cfg4 shards, 12.5M x 768, the reference's literal cosine key (the hard case of the intervals: twice the exact rows with rounded rows)"""
    import zebra_amd as za
    _every_sweep_agrees(za, 12_500_000, 768, "cos_parity", 10, 1024, 4096, 15, [("approx", 2), ("approx-valu", 1), ("scan", 0)],
                        rows0=25_000_000, n_total=100_000_000)


def test_cfg5_shard_full_size_every_sweep_agrees():
    """one of 8 cfg5 shards, 125M x 128 SIFT-style: leaf by leaf at half width (the default there), the f32 leaf-major sweep, the d = 128 half-width table scan"""
    import zebra_amd as za
    _every_sweep_agrees(za, 125_000_000, 128, "l2", 10, 4096, 8192, 15, [("auto", 3), ("leaf", 0), ("approx", 1)], kind=1,
                        rows0=250_000_000, n_total=1_000_000_000)


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
