#!/usr/bin/env python3
"""The SHAPE of the reference-default forest (1M x 384, max_node_size 5, 15 trees; /root/reference/src/database/index/lsh.rs:131-138) as the
blocked walk sees it: nodes, blocks (maximal subtrees of <= 64 nodes), upper nodes, and what a second level of blocks over the upper
nodes would hold -- the numbers behind DESIGN.md s9 "the walk as a dynamic programme" (a DP reads EVERY node for every query; the serial
walk a few percent of them).

    gpurun -- python tests/probes/refdefault_shape.py [rows] [dim]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import zebra_amd as za  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
ix = za.LSHIndex(d, za.LSHIndexOptions(5, 15), seed=0x5EB2A003, device=0)
ix.append_synthetic(n, seed=0x5EB2A001, first_row=0, kind=0)
ix.build()
f = ix.get_forest()
plane, left, right, roots = f["plane"], f["left"], f["right"], f["roots"]
nn = plane.shape[0]
inner = plane >= 0
size = np.zeros(nn, np.int64)
# subtree sizes: children have larger indices than their parents?  not guaranteed -> explicit post-order per tree
order = []
for r in roots:
    st = [int(r)]
    while st:
        v = st.pop()
        order.append(v)
        if inner[v]:
            st.append(int(right[v]))
            st.append(int(left[v]))
order = np.array(order, np.int64)
for v in order[::-1]:
    size[v] = 1 + (size[left[v]] + size[right[v]] if inner[v] else 0)
leaf_len = right[~inner]
print(f"rows {n}, dim {d}: nodes {nn} ({inner.sum()} inner = planes, {(~inner).sum()} leaves: mean {leaf_len.mean():.2f} rows, {(leaf_len == 0).mean() * 100:.1f} % empty)")


def decompose(is_leaf_unit, sz, cap):
    """maximal subtrees with sz <= cap: (#units, sizes), and the nodes left above them"""
    units, above = [], 0
    for r in roots:
        st = [int(r)]
        while st:
            v = st.pop()
            if sz[v] <= cap:
                units.append(int(sz[v]))
            else:
                above += 1
                st.append(int(right[v]))
                st.append(int(left[v]))
    return np.array(units), above


blocks, upper = decompose(None, size, 64)
print(f"blocks (<= 64 nodes): {blocks.shape[0]} (mean {blocks.mean():.1f} nodes, {np.percentile(blocks, 10):.0f}-{np.percentile(blocks, 90):.0f} p10-p90), "
      f"upper nodes {upper} ({upper / len(roots):.0f} per tree)")
# a second level: the upper tree with every block as ONE leaf slot; maximal subtrees of <= 63 slots
usize = np.zeros(nn, np.int64)
for v in order[::-1]:
    usize[v] = 1 if size[v] <= 64 else 1 + usize[left[v]] + usize[right[v]]
sb, upper2, st = [], 0, [int(r) for r in roots]
while st:
    v = st.pop()
    if size[v] <= 64:
        continue              # a block hanging off a high upper node: not a super-block by itself
    if usize[v] <= 63:
        sb.append(int(usize[v]))
    else:
        upper2 += 1
        st.append(int(right[v]))
        st.append(int(left[v]))
sblocks = np.array(sb)
print(f"super-blocks (<= 63 slots of upper nodes + block leaves): {sblocks.shape[0]} (mean {sblocks.mean():.1f} slots), upper nodes left above them {upper2} "
      f"({upper2 / len(roots):.0f} per tree)")
print(f"a DP over the whole forest reads {nn} nodes x 256 queries = {nn * 256 / 1e9:.2f} G node-queries per batch")
ix.close()
