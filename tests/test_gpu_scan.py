"""The table-scan sweep (zh_set_sweep_mode 2: every stored row streamed once per batch, scored against every query that visits
one of its leaves) against the oracle and against the leaf-major sweep: identical ids, keys and counts, bit for bit --
Metric::distance(stored, query) over the rows of the visited leaves (lsh.rs:311-316, 557-560) is the same set of
(row, query) pairs in both; only the order in which the GPU walks them differs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import zebra_oracle as zo  # noqa: E402  (the checker)


@pytest.fixture(scope="module")
def za():
    import zebra_amd
    return zebra_amd


def all_metrics(za):
    return [(za.L2SquaredDistance(), zo.L2SQ, 0), (za.L2Distance(), zo.L2, 0),
            (za.CosineDistance(parity=True), zo.COSINE, zo.PARITY), (za.CosineDistance(parity=False), zo.COSINE, zo.CORRECTED)]


CASES = [
    # n, d, M, T, k, batch, kind
    (5000, 64, 5, 15, 10, 24, 0),       # reference defaults: thousands of tiny leaves per pair
    (20000, 384, 256, 15, 10, 64, 0),   # one leaf per tree, partial last 1-KiB piece of a row
    (8000, 768, 512, 8, 100, 32, 0),
    (6000, 128, 300, 10, 10, 48, 1),    # half-wave rows
    (9000, 256, 100, 64, 10, 12, 0),    # 64 trees: 4 rows per wave
    (4000, 1536, 64, 3, 10, 7, 0),
    (3001, 512, 3002, 5, 10, 300, 0),   # ONE leaf per tree, visited by every query: the hot-leaf path (no pair list)
    (7000, 1024, 12, 6, 10, 9, 0),      # leaves ~ k: backup walks
]


@pytest.mark.parametrize("n,d,M,T,k,B,kind", CASES)
def test_table_scan_equals_oracle_and_leaf_sweep(za, n, d, M, T, k, B, kind):
    X = zo.synth_rows(n, d, kind=kind)
    Q = zo.synth_queries(B, d, n, kind=kind)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    assert zo.canonical_forest(ix.get_forest(), d) == zo.canonical_forest(f.arrays(), d)
    for m, om, omode in all_metrics(za):
        oi, ok, oc = f.search_batch(Q, k, om, omode)
        for mode in ("scan", "leaf"):
            ix.set_sweep_mode(mode)
            ids, keys, counts = ix.search_batch(Q, k, m)
            assert ix.stats()["table_scan"] == (1 if mode == "scan" else 0)
            assert (counts == oc).all(), (mode, om)
            for b in range(B):
                c = int(oc[b])
                assert (ids[b, :c] == oi[b, :c]).all() and (keys[b, :c] == ok[b, :c]).all(), (mode, om, b)
    ix.close()


def test_table_scan_other_metrics_and_fallback(za):
    """the ten `distances`-crate metrics at a production dimension; a dimension the scan does not specialise falls back"""
    n, d, M, T, k, B = 5000, 384, 64, 6, 10, 20
    X = zo.synth_rows(n, d)
    Q = zo.synth_queries(B, d, n)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    for m in (za.ChebyshevDistance(), za.CanberraDistance(), za.BrayCurtisDistance(), za.ManhattanDistance(), za.L3Distance(),
              za.L4Distance(), za.HammingDistance(), za.MinkowskiDistance(3), za.PNormDistance(2)):
        ix.set_sweep_mode("leaf")
        a = ix.search_batch(Q, k, m)
        ix.set_sweep_mode("scan")
        b = ix.search_batch(Q, k, m)
        assert ix.stats()["table_scan"] == 1
        assert all((x == y).all() for x, y in zip(a, b)), type(m).__name__
    ix.close()
    ix = za.LSHIndex(100, za.LSHIndexOptions(40, 5))  # d = 100: only the leaf-major sweep has a runtime-d kernel
    ix.add(zo.synth_rows(3000, 100))
    ix.set_sweep_mode("scan")
    ix.search_batch(zo.synth_queries(9, 100, 3000), 7, za.L2Distance())
    assert ix.stats()["table_scan"] == 0
    ix.close()


def test_table_scan_after_insert_remove_and_in_windows(za):
    """the row -> leaf table follows the forest: incremental add, remove, and a window of batches through the contexts"""
    n0, n1, d, M, T, k, B = 6000, 1500, 128, 64, 5, 10, 32
    X = zo.synth_rows(n0 + n1, d)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.set_sweep_mode("scan")
    ix.add(X[:n0])
    Q = zo.synth_queries(B, d, n0 + n1)
    m = za.L2SquaredDistance()
    ix.search_batch(Q, k, m)
    ix.add(X[n0:])           # insert into the built forest: leaves grow / split
    ix.remove(np.arange(100, 400, dtype=np.uint64))
    f = zo.Forest.build(X[:n0], M, T)
    f.insert(X, n0)
    f.remove(np.arange(100, 400, dtype=np.uint64))
    assert zo.canonical_forest(ix.get_forest(), d) == zo.canonical_forest(f.arrays(), d)
    ids, keys, counts = ix.search_batch(Q, k, m)
    assert ix.stats()["table_scan"] == 1
    oi, ok, oc = f.search_batch(Q, k, zo.L2SQ, 0)
    assert (counts == oc).all()
    for b in range(B):
        assert (ids[b, :oc[b]] == oi[b, :oc[b]]).all() and (keys[b, :oc[b]] == ok[b, :oc[b]]).all()
    assert not np.isin(ids[ids != np.uint64(2**64 - 1)], np.arange(100, 400, dtype=np.uint64)).any()
    ix.close()


def test_table_scan_with_rows_appended_but_not_yet_in_a_tree(za):
    """zh_index_append after a build: the new rows are stored but in no tree until the next add / build; the scan covers them
    with 'in no leaf' entries and results equal the leaf-major sweep's"""
    n, d, M, T, k, B = 5000, 256, 64, 4, 10, 16
    X = zo.synth_rows(n + 700, d)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X[:n])
    Q = zo.synth_queries(B, d, n)
    m = za.L2Distance()
    ix.set_sweep_mode("scan")
    a = ix.search_batch(Q, k, m)
    ix.append(X[n:])            # staged rows: stored, not yet indexed
    b = ix.search_batch(Q, k, m)
    assert ix.stats()["table_scan"] == 1 and ix.stats()["rows_swept"] == n + 700
    ix.set_sweep_mode("leaf")
    c = ix.search_batch(Q, k, m)
    assert all((x == y).all() for x, y in zip(a, b)) and all((x == y).all() for x, y in zip(a, c))
    ix.close()


def test_injected_forest_with_a_row_listed_twice_in_one_tree_is_swept_leaf_by_leaf(za):
    """rowLeaf keeps ONE {leaf, position} per (row, tree); an injected forest that lists a row twice inside a
    tree (legal input for zh_index_set_forest) would leave the second occurrence's key slot unwritten under the table scan.
    The library detects it and serves such a forest leaf by leaf whatever the sweep mode says: results equal the oracle's and
    do not depend on zh_set_sweep_mode."""
    n, d, M, T, k, B = 6000, 128, 64, 4, 10, 32
    X = zo.synth_rows(n, d)
    f0 = zo.Forest.build(X, M, T)
    arr = {k_: np.array(v, copy=True) for k_, v in f0.arrays().items()}
    plane, left, right = arr["plane"], arr["left"], arr["right"]
    leaves = [i for i in range(plane.size) if plane[i] < 0 and right[i] >= 4]
    for node in leaves[:40]:  # repeat the first id of forty leaves inside the same leaf
        off = int(np.uint32(left[node]))
        arr["leaf_ids"][off + 1] = arr["leaf_ids"][off]
    f = zo.Forest.from_arrays(X, M, arr)
    Q = zo.synth_queries(B, d, n)
    oi, ok, oc = f.search_batch(Q, k, zo.L2SQ, 0)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.append(X)
    ix.set_forest(arr)
    for mode in ("scan", "leaf", "auto"):
        ix.set_sweep_mode(mode)
        for _ in range(4):  # (auto builds the row -> leaf table only for a forest that has served a few batches)
            ids, keys, counts = ix.search_batch(Q, k, za.L2SquaredDistance())
        assert ix.stats()["table_scan"] == 0, mode
        assert (counts == oc).all(), mode
        for b in range(B):
            assert (ids[b, :oc[b]] == oi[b, :oc[b]]).all() and (keys[b, :oc[b]] == ok[b, :oc[b]]).all(), (mode, b)
    # the same forest without the repeats takes the scan when asked to
    ix.set_forest(f0.arrays())
    ix.set_sweep_mode("scan")
    ix.search_batch(Q, k, za.L2SquaredDistance())
    assert ix.stats()["table_scan"] == 1
    ix.close()
