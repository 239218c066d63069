#!/usr/bin/env python3
"""Incremental `add` (lsh.rs:350-382,445-462) at scale: rows per second into a built index, then a sampled check of
the resulting forest's search results against the oracle (forest injected from the GPU)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zebra_amd as za  # noqa: E402
from oracle import zebra_oracle as zo  # noqa: E402

n0, d, M, T = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, 128, int(sys.argv[2]) if len(sys.argv) > 2 else 64, 15
step, steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000, 10
X = zo.synth_rows(n0 + step * steps, d)
ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
t0 = time.perf_counter()
ix.add(X[:n0])
print(f"build of {n0} rows: {time.perf_counter() - t0:.2f} s", flush=True)
for i in range(steps):
    t0 = time.perf_counter()
    ix.add(X[n0 + i * step: n0 + (i + 1) * step])
    dt = time.perf_counter() - t0
    print(f"add #{i}: {step} rows in {dt * 1e3:.1f} ms = {step / dt:.0f} rows/s, {len(ix)} stored", flush=True)
t0 = time.perf_counter()
gone = np.arange(0, n0, n0 // 5000, dtype=np.uint64)
ix.remove(gone)
print(f"remove of {gone.size} ids: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
Q = zo.synth_queries(8, d, n0)
ids, keys, counts = ix.search_batch(Q, 10, za.L2SquaredDistance())
f = zo.Forest.from_arrays(X, M, ix.get_forest())
for b in range(8):
    oi, ok = f.search(Q[b], 10, zo.L2SQ)
    assert (ids[b, :len(oi)] == oi).all() and (keys[b, :len(oi)] == ok).all()
print("search after the adds and removes matches the oracle on the same forest")
