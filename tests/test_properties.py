"""Property tests (hypothesis) of the invariants the design leans on, on the oracle:
  * S-shard merge == global top-k of the union (SURVEY s8e parity definition), any counts / ties;
  * union of per-leaf top-k == top-k of the union of the leaves' rows when every take equals k
    (why the distances need not be recomputed in the rerank);
  * the walk's control flow depends only on signs, leaf lengths and n: two metrics give the same visit list."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import zebra_oracle as zo


@settings(max_examples=60, deadline=None)
@given(S=st.integers(1, 8), b=st.integers(1, 5), k=st.integers(1, 12), seed=st.integers(0, 2**31 - 1))
def test_merge_is_topk_of_union(S, b, k, seed):
    rng = np.random.default_rng(seed)
    keys = rng.integers(0, 20, (S, b, k)).astype(np.uint64)
    ids = rng.permutation(S * b * k).reshape(S, b, k).astype(np.uint64)
    counts = rng.integers(0, k + 1, (S, b)).astype(np.uint32)
    oi, ok, oc = zo.merge_topk(ids, keys, counts, k)
    for q in range(b):
        pool = sorted((int(keys[s, q, i]), int(ids[s, q, i])) for s in range(S) for i in range(counts[s, q]))[:k]
        assert [(int(ok[q, i]), int(oi[q, i])) for i in range(oc[q])] == pool


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), M=st.sampled_from([40, 64, 100]), k=st.integers(1, 10))
def test_search_is_topk_of_the_visited_leaves(seed, M, k):
    X = zo.synth_rows(1500, 24, seed=seed)
    f = zo.Forest.build(X, M, 4, seed=seed)
    q = zo.synth_queries(1, 24, 1500, seed_rows=seed)[0]
    a = f.arrays()
    rows, all_full = set(), True
    for t in range(4):
        _, _, visits = f.tree_result(t, q, k, zo.L2SQ)
        for off, ln, take in visits.tolist():
            rows |= set(a["leaf_ids"][off:off + ln].tolist())
            all_full &= take == k
    ids, keys = f.search(q, k, zo.L2SQ)
    if all_full:  # one-leaf-per-tree regime: top-k of the union of per-leaf top-k == top-k of all visited rows
        rr = sorted(rows)
        kk = zo.distance_batch(zo.L2SQ, 0, X[rr], q)
        assert sorted(zip(kk.tolist(), rr))[:k] == list(zip(keys.tolist(), ids.tolist()))
    assert set(ids.tolist()) <= rows


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(1, 30))
def test_walk_control_flow_is_metric_independent(seed, n):
    X = zo.synth_rows(800, 16, seed=seed)
    f = zo.Forest.build(X, 9, 3, seed=seed)
    q = zo.synth_queries(1, 16, 800, seed_rows=seed)[0]
    for t in range(3):
        r1, _, v1 = f.tree_result(t, q, n, zo.L2SQ)
        r2, _, v2 = f.tree_result(t, q, n, zo.COSINE, zo.PARITY)
        assert r1 == r2 and v1.tolist() == v2.tolist()
