"""The on-demand-row oracle (zo_search_batch_synth, zo_check_forest_synth) that checks the full-size configurations
exactly must itself equal the in-memory oracle: same ids / keys / counts for every metric family, both walk regimes,
shard offsets (first_row) and every synthetic data kind; and the structural forest check accepts the oracle's own
build while catching each kind of corruption."""
import numpy as np
import pytest

from oracle import zebra_oracle as zo


@pytest.mark.parametrize("n,d,M,T,k,kind,row0", [
    (3000, 32, 64, 5, 10, 0, 0),        # one leaf per tree
    (3000, 32, 5, 6, 10, 0, 0),         # reference defaults: the wandering walk (F5)
    (4096, 128, 256, 4, 10, 1, 7_000_000),   # SIFT-style integer rows, a shard that starts at row 7M
    (2500, 48, 100, 3, 100, 2, 123),    # clustered rows, k = 100 = leaves barely >= k
    (2500, 48, 100, 3, 10, 3, 77),      # clustered rows scattered over the table (a row's cluster from a hash of its id)
])
def test_synth_search_equals_in_memory_search(n, d, M, T, k, kind, row0):
    X = zo.synth_rows(n, d, row0=row0, kind=kind)
    f = zo.Forest.build(X, M, T)
    fs = zo.Forest.borrow_synth(n, d, M, f.arrays(), first_row=row0, kind=kind)
    Q = zo.synth_queries(12, d, n, kind=kind)
    for metric, mode in ((zo.L2SQ, 0), (zo.L2, 0), (zo.COSINE, zo.PARITY), (zo.COSINE, zo.CORRECTED), (zo.MANHATTAN, 0)):
        a = f.search_batch(Q, k, metric, mode)
        b = fs.search_batch_synth(Q, k, metric, mode, nthreads=2)
        assert (a[2] == b[2]).all()
        for q in range(Q.shape[0]):
            c = int(a[2][q])
            assert (a[0][q, :c] == b[0][q, :c]).all() and (a[1][q, :c] == b[1][q, :c]).all()


def test_forest_check_accepts_the_oracle_build_and_catches_corruption():
    n, d, M, T = 5000, 24, 50, 4
    X = zo.synth_rows(n, d)
    f = zo.Forest.build(X, M, T)
    arr = f.arrays()
    rc, planes = zo.Forest.borrow_synth(n, d, M, arr).check_synth(n_sample=40)
    assert rc == 0 and planes > 40 * T * 3

    def check(mut):
        a = {k_: v.copy() for k_, v in arr.items()}
        mut(a)
        return zo.Forest.borrow_synth(n, d, M, a).check_synth(n_sample=40)[0]

    def dup_id(a):       # a row twice in one tree, another missing
        a["leaf_ids"][1] = a["leaf_ids"][0]
    assert check(dup_id) == 1

    def wrong_plane(a):  # the root plane of tree 0 no longer the bisector of its sample pair
        a["planes"][a["plane"][a["roots"][0]]][3] += 1.0
    assert check(wrong_plane) in (3, 4)

    def wrong_const(a):
        p = a["plane"][a["roots"][1]]
        a["consts"][p] = np.nextafter(a["consts"][p], np.float32(np.inf))
    assert check(wrong_const) == 4

    def swap_children(a):  # below / above exchanged at a root (lsh.rs:260-264)
        r = a["roots"][2]
        a["left"][r], a["right"][r] = a["right"][r], a["left"][r]
    assert check(swap_children) in (3, 4)  # the subtrees then sit under the wrong heap paths too
    # a different sampling seed: every plane differs
    assert zo.Forest.borrow_synth(n, d, M, arr).check_synth(index_seed=zo.SEED_INDEX + 1, n_sample=8)[0] == 4
    # the size rule: claim a larger max_node_size than the build used -> inner nodes smaller than M
    assert zo.Forest.borrow_synth(n, d, 4 * M, arr).check_synth(n_sample=4)[0] == 2
