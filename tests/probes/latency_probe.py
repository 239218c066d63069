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

# the reference's calling pattern: ONE query per call from many threads at once (core.rs:299-303).  zh_search_batch combines the
# callers that arrive while a batch is on the GPU: per-call latency and calls per second by thread count
import ctypes as C  # noqa: E402
import threading  # noqa: E402
from zebra_amd import _ffi  # noqa: E402
lib, h = _ffi.lib(), ix._h
NQ = 4096
Q = rng.standard_normal((NQ, d)).astype(np.float32)
ids, keys, counts = np.zeros((NQ, k), np.uint64), np.zeros((NQ, k), np.uint64), np.zeros(NQ, np.uint32)
for NT in (1, 4, 16, 64):
    per = NQ // NT if NT > 1 else 256
    lat = [[] for _ in range(NT)]
    go = threading.Barrier(NT + 1)

    def worker(t):
        go.wait()
        for j in range(per):
            i = t * per + j
            t0 = time.perf_counter()
            lib.zh_search_batch(h, Q[i].ctypes.data_as(C.c_void_p), 1, k, m.metric, m.mode, ids[i].ctypes.data_as(C.c_void_p),
                                keys[i].ctypes.data_as(C.c_void_p), counts[i:i + 1].ctypes.data_as(C.c_void_p))
            lat[t].append(time.perf_counter() - t0)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(NT)]
    for x in th:
        x.start()
    ix.stats(reset=True)
    go.wait()
    t0 = time.perf_counter()
    for x in th:
        x.join()
    el = time.perf_counter() - t0
    a = np.sort(np.concatenate([np.array(x) for x in lat])) * 1e3
    st = ix.stats()
    print(f"single-query calls from {NT} threads: {NT * per / el:.0f} calls/s; per call p50 {a[len(a) // 2]:.3f} ms p99 {a[int(len(a) * 0.99)]:.3f} ms; "
          f"{st['combined_batches_accum']} combined batches served {st['combined_calls_accum']} of {NT * per} calls", flush=True)
