#!/usr/bin/env python3
"""A/B of the 'rows stored in tree-0 leaf order' layout WITHOUT changing the library:
build the index, then build a second one whose physical row p is row perm[p] of the first (perm = tree 0's leaves in
order) with every leaf id remapped -- the exact memory layout the change would produce; ids come back permuted, keys and
timing are what the real thing would give.  Prints the sweep's per-launch time and bytes for both layouts, and how
clustered the leaves of the OTHER trees become (runs of physically adjacent rows per leaf).

    python tests/probes/layout_probe.py [rows] [dim] [max_node_size] [batch] [kind]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import zebra_amd as za  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
M = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
kind = int(sys.argv[5]) if len(sys.argv) > 5 else 1
T, k = 15, 10
dev = torch.device("cuda", 0)
met = za.L2Distance()


def wrap(ptr, shape, typestr):
    class E:
        pass
    e = E()
    e.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 3, "strides": None}
    return torch.as_tensor(e, device=dev)


def measure(ix, tag):
    qs = []
    for i in range(6):
        q = torch.empty((B, d), dtype=torch.float32, device=dev)
        za.synth_queries_device(0, q.data_ptr(), n, B, d, b0=i * B, kind=kind)
        qs.append(q)
    ids = torch.empty((B, k), dtype=torch.int64, device=dev)
    keys = torch.empty_like(ids)
    counts = torch.empty(B, dtype=torch.int32, device=dev)
    ix.search_batch_device(qs[0].data_ptr(), B, k, met, ids.data_ptr(), keys.data_ptr(), counts.data_ptr())
    ix.set_profiling(1)
    ix.stats(reset=True)
    t0 = time.perf_counter()
    for q in qs[1:]:
        ix.search_batch_device(q.data_ptr(), B, k, met, ids.data_ptr(), keys.data_ptr(), counts.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    st = ix.stats()
    nl = st["sweep_launches_accum"]
    per = st["ms_sweep"] / nl
    by = (st["swept_rows_accum"] * (4 * d + 4) + st["sweep_rows_accum"] * 8) / nl
    print(f"{tag}: {dt * 1e3:.2f} ms/batch (blocking), sweep {st['ms_sweep'] / 5:.2f} ms/batch, {per:.3f} ms/launch, "
          f"{by / per / 1e6:.0f} GB/s of loaded bytes", flush=True)
    ix.set_profiling(0)
    return keys.cpu().numpy().copy()


ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), reserve_rows=n)
ix.append_synthetic(n, kind=kind)
ix.build()
k1 = measure(ix, "insertion order ")
g = ix.get_forest()
perm = g["leaf_ids"][:n].copy()            # tree 0's leaves, in order: physical row p <- logical row perm[p]
inv = np.empty(n, np.uint32)
inv[perm] = np.arange(n, dtype=np.uint32)
# how clustered do the other trees' leaves become?  runs of physically adjacent rows per leaf (1.0 = all scattered)
leaves = np.flatnonzero(g["plane"] < 0)
rng = np.random.default_rng(1)
fr = []
for node in rng.choice(leaves, 200):
    off, ln = int(np.uint32(g["left"][node])), int(g["right"][node])
    if off < n or ln < 64:
        continue
    p = np.sort(inv[g["leaf_ids"][off:off + ln]])
    runs = 1 + int((np.diff(p) != 1).sum())
    pages = len(np.unique(p // 8))  # 4 KiB of 512-B rows
    fr.append((runs / ln, pages / ln))
fr = np.array(fr)
print(f"other trees' leaves in tree-0 order: {fr[:, 0].mean():.3f} runs per row, {fr[:, 1].mean():.3f} distinct 4-KiB pages per row "
      f"({len(fr)} leaves sampled)", flush=True)
X = wrap(ix.rows_device_ptr(), (n, d), "<f4")
ix2 = za.LSHIndex(d, za.LSHIndexOptions(M, T), reserve_rows=n)
step = 4_000_000
for s in range(0, n, step):
    idx = torch.from_numpy(perm[s:s + step].astype(np.int64)).to(dev)
    chunk = X.index_select(0, idx)
    za._ffi.check(za._ffi.lib().zh_index_append_device(ix2._h, chunk.data_ptr(), chunk.shape[0]))
    del chunk, idx
del X
ix.close()
g2 = dict(g)
lid = inv[g["leaf_ids"]]
# inside a leaf, ascending physical order (the sweep then walks each leaf front to back in memory)
for node in leaves:
    off, ln = int(np.uint32(g["left"][node])), int(g["right"][node])
    if ln > 1:
        lid[off:off + ln].sort()
g2["leaf_ids"] = lid
ix2.set_forest(g2)
k2 = measure(ix2, "tree-0 leaf order")
# same queries, rows and forest: the keys can only differ where equal keys straddle a cut (ties break on the id, and the
# ids are renumbered here; integer-valued rows tie often)
print("queries with identical key lists in both layouts: %.4f" % float((k1 == k2).all(1).mean()))
