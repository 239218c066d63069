#!/usr/bin/env python3
"""The reference's default options (max_node_size 5, 15 trees) at batch size: the walk wanders over most of the
forest, so a batch is millions of 2-4-row leaf visits.  Prints per-stage times; results are checked against
the oracle for a few queries."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zebra_amd as za  # noqa: E402
from oracle import zebra_oracle as zo  # noqa: E402

n, d, B, k = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000, int(os.environ.get("PROBE_DIM", "384")), int(sys.argv[2]) if len(sys.argv) > 2 else 64, 10
COS = os.environ.get("PROBE_METRIC") == "cosine"  # the reference's image / audio databases: Database<768, CosineDistance>
X = zo.synth_rows(n, d)
Q = zo.synth_queries(B, d, n)
ix = za.LSHIndex(d, za.LSHIndexOptions(5, 15))
t0 = time.perf_counter()
ix.add(X)
print("build s", round(time.perf_counter() - t0, 2), "planes", ix.get_forest()["consts"].size)
m = za.CosineDistance(parity=True) if COS else za.L2SquaredDistance()
if os.environ.get("DENSE_LEVELS"):
    ix.set_dense_levels(int(os.environ["DENSE_LEVELS"]))
for _ in range(5):  # the first batch learns the visits per pair (on-demand hash + emit walk); once the forest has served a few
    ix.search_batch(Q, k, m)  # batches unchanged the library builds its derived views (blocked forest, row -> leaf table: host time, once)
st0 = ix.stats()
print("warm-up:", {a: st0[a] for a in ("prefiltered", "prefilter_fallbacks_accum", "prefilter_last_overflow", "prefilter_exact_visits", "prefilter_exact_rows")})
ix.set_profiling(1)
ix.stats(reset=True)
t0 = time.perf_counter()
for _ in range(3):
    ids, keys, counts = ix.search_batch(Q, k, m)
dt = (time.perf_counter() - t0) / 3
st = ix.stats()
print("ms/batch", round(dt * 1e3, 2), "qps", round(B / dt), {s: round(st["ms_" + s] / st["timed_batches"], 3) for s in ("hash", "walk", "sweep", "select", "final")},
      "visits", st["visits"], "rows", st["rows_scored"], "cands", st["candidates"],
      {a: st[a] for a in ("hash_from_scores", "hash_exact_fixups", "prefiltered", "prefilter_exact_visits", "prefilter_exact_rows", "prefilter_fallbacks_accum", "prefilter_last_overflow")})
f = zo.Forest.from_arrays(X, 5, ix.get_forest())
for b in sorted({0, min(1, B - 1), B // 2, max(B - 2, 0), B - 1}):
    oi, ok = f.search(Q[b], k, zo.COSINE, zo.PARITY) if COS else f.search(Q[b], k, zo.L2SQ)
    assert (ids[b] == oi).all() and (keys[b] == ok).all()
print("checked against the oracle")
