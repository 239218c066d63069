"""The walk of `tree_result` (/root/reference/src/database/index/lsh.rs:290-348) as a DYNAMIC PROGRAMME -- a host-side proof of concept for the
reference-default regime (max_node_size 5 < top_k: thousands of leaf visits per (query, tree) pair), where the GPU walk is a serial
pointer chase per pair.

What a subtree RETURNS depends on the tree, the leaf lengths, the query's signs and the demand n it is entered with -- never on a distance: a
leaf returns min(len, n) (lsh.rs:300-329: it scores its rows only to pick WHICH n it hands over), an inner node the main child's count or,
when that is short of n, the backup child's count ALONE (lsh.rs:340-345).  So  R[node][n], n = 0 .. top_k,  can be filled bottom-up for every
node at once (level-parallel, 11 bytes per node at top_k = 10), and a top-down pass then gives every node the demand it is entered with (or
"not entered"): the visited leaves and their `take` -- exactly the visit list the sequential walk produces, as a set.

This test checks that equivalence against the oracle's literal walk on small forests in both regimes; no GPU, no product code."""
import numpy as np
import pytest

from oracle import zebra_oracle as zo


def walk_by_dp(arr, signs, tree, n0):
    """-> (return value of the root, {leaf node: (leaf_off, len, take)}) for demand n0; arr = Forest.arrays(), signs[p] = point_is_above(plane p)"""
    plane, left, right = arr["plane"], arr["left"], arr["right"]
    root = int(arr["roots"][tree])
    # the subtree's nodes, parents before children
    order, stack = [], [root]
    while stack:
        v = stack.pop()
        order.append(v)
        if plane[v] >= 0:
            stack += [int(left[v]), int(right[v])]
    # bottom-up: R[v][n]
    R = {}
    for v in reversed(order):
        if plane[v] < 0:
            ln = int(right[v])
            R[v] = np.minimum(ln, np.arange(n0 + 1))
        else:
            above = bool(signs[plane[v]])
            main, backup = (int(right[v]), int(left[v])) if above else (int(left[v]), int(right[v]))  # lsh.rs:335-338
            k = R[main]
            r = k.copy()
            short = k < np.arange(n0 + 1)
            idx = np.arange(n0 + 1) - k
            r[short] = R[backup][idx[short]]                                                           # lsh.rs:341-343: the backup's count alone
            R[v] = r
    # top-down: the demand every entered node is entered with
    demand = {root: n0}
    visits = {}
    for v in order:
        if v not in demand:
            continue
        n = demand[v]
        if plane[v] < 0:
            off, ln = int(left[v]), int(right[v])
            visits[v] = (off, ln, min(ln, n))
            continue
        above = bool(signs[plane[v]])
        main, backup = (int(right[v]), int(left[v])) if above else (int(left[v]), int(right[v]))
        demand[main] = n
        k = int(R[main][n])
        if k < n:
            demand[backup] = n - k
    return int(R[root][n0]), visits


@pytest.mark.parametrize("n,d,M,T,k", [(3000, 32, 5, 4, 10),      # the reference's defaults: near-exhaustive walks
                                        (3000, 32, 5, 3, 3),
                                        (2000, 16, 8, 3, 64),
                                        (4000, 32, 200, 4, 10),    # leaves >= top_k: one leaf per tree
                                        (1500, 16, 12, 3, 10)])    # leaves around top_k: a few backups
def test_dp_visits_equal_the_literal_walk(n, d, M, T, k):
    X = zo.synth_rows(n, d)
    f = zo.Forest.build(X, M, T)
    arr = f.arrays()
    Q = zo.synth_queries(6, d, n)
    for q in Q:
        signs, _ = f.hash_signs(q)
        for t in range(T):
            ret, cand, vis = f.tree_result(t, q, k, zo.L2, 0, cap_visits=1 << 18)
            r_dp, v_dp = walk_by_dp(arr, signs, t, k)
            assert r_dp == ret
            want = {(int(a), int(b), int(c)) for a, b, c in vis}
            got = set(v_dp.values())
            # (the literal walk logs a leaf it steps on even when it is empty; a leaf is met at most once per walk)
            assert got == want, (t, len(got), len(want))
