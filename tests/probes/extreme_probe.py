#!/usr/bin/env python3
"""Extreme-regime probe: configurations far from the BASELINE shapes, each checked against the oracle (forest
injected from the GPU build) on a sample of the batch.  Prints one line per case; exits non-zero on a mismatch."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zebra_amd as za  # noqa: E402
from oracle import zebra_oracle as zo  # noqa: E402

CASES = [
    # name, n, d, M, T, B, k, kind, metric
    ("huge batch, k=1024", 200_000, 64, 3000, 8, 8192, 1024, 0, "l2sq"),
    ("d=4096", 20_000, 4096, 64, 4, 64, 10, 0, "cos"),
    ("giant leaves", 300_000, 32, 200_000, 3, 128, 1000, 0, "l2"),
    ("64 trees", 50_000, 128, 100, 64, 256, 10, 2, "l2sq"),
    ("leaves of 1", 20_000, 16, 1, 2, 32, 5, 0, "l2sq"),
    ("degenerate duplicates", 60_000, 24, 50, 3, 64, 50, 0, "l2sq"),
    ("k > n", 300, 8, 5, 15, 16, 1024, 0, "l2sq"),
]
MET = {"l2sq": (za.L2SquaredDistance(), zo.L2SQ, 0), "l2": (za.L2Distance(), zo.L2, 0),
       "cos": (za.CosineDistance(parity=True), zo.COSINE, zo.PARITY)}
bad = 0
for name, n, d, M, T, B, k, kind, met in CASES:
    X = zo.synth_rows(n, d, kind=kind)
    if name.startswith("degenerate"):
        X[: n - 100] = X[0]  # one point repeated: an unsplittable node, leaf at the depth guard
    Q = zo.synth_queries(B, d, n, kind=kind)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    t0 = time.perf_counter()
    ix.add(X)
    tb = time.perf_counter() - t0
    m, om, omode = MET[met]
    ids, keys, counts = ix.search_batch(Q, k, m)
    t0 = time.perf_counter()
    ids, keys, counts = ix.search_batch(Q, k, m)
    ts = time.perf_counter() - t0
    st = ix.stats()
    f = zo.Forest.from_arrays(X, M, ix.get_forest())
    ok = True
    for b in sorted(set([0, 1, B // 2, B - 1])):
        oi, okk = f.search(Q[b], k, om, omode)
        c = len(oi)
        if counts[b] != c or not (ids[b, :c] == oi).all() or not (keys[b, :c] == okk).all():
            ok = False
    bad += not ok
    print(f"{'OK ' if ok else 'BAD'} {name:24s} n={n} d={d} M={M} T={T} B={B} k={k}: build {tb:.2f}s search {ts * 1e3:.1f} ms "
          f"visits {st['visits']} rows {st['rows_scored']} cands {st['candidates']}", flush=True)
sys.exit(1 if bad else 0)
