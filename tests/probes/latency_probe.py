#!/usr/bin/env python3
"""Blocking-call latency of small batches on a cfg3-shaped index (rows x 768, max_node_size 4096, 15 trees, L2 top-100):
what a caller of the reference's one-query `search` (lsh.rs:544) sees."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zebra_amd as za  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d, M, T, k = 768, 4096, 15, 100
ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), reserve_rows=n)
ix.append_synthetic(n)
ix.build()
rng = np.random.default_rng(0)
m = za.L2Distance()
for B in (1, 4, 16, 64):
    Q = rng.standard_normal((B, d)).astype(np.float32)
    for _ in range(5):
        ix.search_batch(Q, k, m)
    ix.set_profiling(1)
    ix.stats(reset=True)
    t = []
    for _ in range(50):
        t0 = time.perf_counter()
        ix.search_batch(Q, k, m)
        t.append(time.perf_counter() - t0)
    st = ix.stats()
    ix.set_profiling(0)
    t = np.sort(np.array(t)) * 1e3
    print(f"B={B}: p50 {t[25]:.3f} ms p90 {t[45]:.3f} ms; kernel stages (ms)",
          {s: round(st['ms_' + s] / st['timed_batches'], 3) for s in ("hash", "walk", "sweep", "select", "final")}, flush=True)
