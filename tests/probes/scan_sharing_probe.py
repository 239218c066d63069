#!/usr/bin/env python3
"""How many of the table scan's query fetches could a block of consecutive stored rows SHARE?  (VERDICT r2 #2a.)

The scan fetches one query (4*d bytes, from L2) per scored (row, query) pair.  A block-cooperative variant would stage each
DISTINCT query of a block of R consecutive rows once in LDS.  This probe measures, on the bench workload's own index and
queries, pairs and distinct queries per block for R = 16 (one wave), 64 (one work-group), 256, 1024 -- for the bench line's
iid rows and for clustered rows (128 consecutive rows share a centre), and for rows in tree-0 leaf order (a block's rows
then share their tree-0 leaf).  Leaf visits are taken as the leaf each query's descent ends in (top_k <= leaf size at these
shapes: one leaf per (query, tree) with rare exceptions), from the library's own hash signs.

    python tests/probes/scan_sharing_probe.py [rows] [dim] [max_node_size] [queries in the window] [kind]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zebra_amd as za  # noqa: E402
from oracle import zebra_oracle as zo  # noqa: E402  (query generator only)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
M = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
W = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
kind = int(sys.argv[5]) if len(sys.argv) > 5 else 0
T = 15

ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), reserve_rows=n)
ix.append_synthetic(n, kind=kind)
ix.build()
g = ix.get_forest()
plane, left, right, roots, leaf_ids = g["plane"], g["left"], g["right"], g["roots"], g["leaf_ids"]
Q = zo.synth_queries(W, d, n, kind=kind)
bits = ix.hash_signs(Q)  # [W, ceil(P / 32)] uint32
ix.close()

# leaf of every row in every tree, and the queries that end in every leaf
leaf_of_row = np.full((T, n), -1, np.int32)
visitors = {}
for t in range(T):
    stack = [int(roots[t])]
    while stack:
        m = stack.pop()
        if plane[m] < 0:
            off, ln = int(np.uint32(left[m])), int(right[m])
            leaf_of_row[t, leaf_ids[off:off + ln]] = m
        else:
            stack += [int(left[m]), int(right[m])]
    for b in range(W):
        m = int(roots[t])
        while plane[m] >= 0:
            p = int(plane[m])
            above = (int(bits[b, p >> 5]) >> (p & 31)) & 1
            m = int(right[m]) if above else int(left[m])
        visitors.setdefault(m, []).append(b)
vis = {m: np.array(v, np.int32) for m, v in visitors.items()}
pairs_total = sum(len(v) * int(right[m]) for m, v in vis.items())
print(f"{n} x {d}, max_node_size {M}, window of {W} queries, kind {kind}: {pairs_total / 1e6:.1f} M pairs = {pairs_total / n:.2f} per stored row", flush=True)

rng = np.random.default_rng(1)


def measure(order, tag):
    """order: physical position -> row id"""
    for R in (16, 64, 256, 1024):
        starts = rng.integers(0, n // R, 400) * R
        pairs = distinct = 0
        for s in starts:
            rows = order[s:s + R] if order is not None else np.arange(s, s + R)
            seen = set()
            for t in range(T):
                for m in np.unique(leaf_of_row[t, rows]):
                    v = vis.get(int(m))
                    if v is not None:
                        pairs += len(v) * int((leaf_of_row[t, rows] == m).sum())
                        seen.update(v.tolist())
            distinct += len(seen)
        print(f"  {tag:34s} block of {R:5d} rows: {pairs / len(starts):8.1f} pairs, {distinct / len(starts):7.1f} distinct queries "
              f"-> fetches shared away: {100 * (1 - distinct / max(pairs, 1)):5.1f} %", flush=True)


measure(None, "rows as stored (id order)")
t0_order = np.concatenate([leaf_ids[int(np.uint32(left[m])):int(np.uint32(left[m])) + int(right[m])]
                           for m in np.unique(leaf_of_row[0]) if m >= 0])
measure(t0_order, "rows in tree-0 leaf order")
