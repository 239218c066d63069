"""The wandering walk of the reference's default options (max_node_size 5 < top_k: thousands of leaf visits per (query, tree) pair,
/root/reference/src/database/index/lsh.rs:290-348 with the defaults of lsh.rs:131-138) over BOTH blocked views of the forest: the round-2 blocks
(<= 64 nodes, leaves hold records: the default) and the round-5 blocks of inner nodes only (ZH_WALK_BLOCKS=inner: leaves live in their parent's
record; measured, not faster -- DESIGN.md s9).  Same visits, same results as the oracle's literal walk, for forests that are one leaf, a few
nodes, and thousands of blocks; with the row-score hash's flagged signs and without."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import zebra_oracle as zo  # noqa: E402  (the checker)


@pytest.fixture(scope="module")
def za():
    import zebra_amd
    return zebra_amd


@pytest.mark.parametrize("view", [None, "inner"])
@pytest.mark.parametrize("n,d,M,T,k,B,hash_mode", [
    (30000, 64, 5, 15, 10, 64, "scores"),    # the reference's defaults: blocks under thousands of upper nodes, flagged signs
    (30000, 64, 5, 15, 10, 61, "dense"),     # every sign from the dense hash: no flags
    (700, 32, 5, 6, 10, 16, "dense"),        # a forest of a few blocks
    (40, 16, 5, 4, 10, 8, "dense"),          # trees of a handful of nodes
    (4, 16, 5, 3, 10, 4, "dense"),           # every tree is ONE leaf (a block of zero inner nodes)
    (5000, 48, 12, 5, 10, 32, "dense"),      # leaves around top_k: backups with small demands
    (9000, 32, 3, 4, 64, 24, "dense"),       # top_k far above the leaves
])
def test_blocked_views_equal_the_literal_walk(za, monkeypatch, view, n, d, M, T, k, B, hash_mode):
    if view:
        monkeypatch.setenv("ZH_WALK_BLOCKS", view)
    else:
        monkeypatch.delenv("ZH_WALK_BLOCKS", raising=False)
    X = zo.synth_rows(n, d)
    Q = zo.synth_queries(B, d, n)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    ix.set_hash_mode(hash_mode)
    ix.set_dense_levels(64)  # every plane from the hash kernels: the blocked walk's precondition
    for m, om, omode in ((za.L2SquaredDistance(), zo.L2SQ, 0), (za.CosineDistance(parity=True), zo.COSINE, zo.PARITY)):
        want = f.search_batch(Q, k, om, omode)
        for _ in range(4):  # (the blocked view is built once the forest has served a few batches unchanged)
            ids, keys, counts = ix.search_batch(Q, k, m)
        assert (counts == want[2]).all()
        for b in range(B):
            c = int(want[2][b])
            assert (ids[b, :c] == want[0][b, :c]).all() and (keys[b, :c] == want[1][b, :c]).all(), (view, b)
    ix.close()
