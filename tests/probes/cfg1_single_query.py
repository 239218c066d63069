import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import zebra_amd as za
from oracle import zebra_oracle as zo
n, d, k = 10000, 384, 10
X = zo.synth_rows(n, d); Q = zo.synth_queries(64, d, n)
ix = za.LSHIndex(d, za.LSHIndexOptions(5, 15)); ix.add(X)
m = za.CosineDistance(parity=True)
for i in range(20): ix.search_batch(Q[i:i+1], k, m)
ts = []
for i in range(200):
    t = time.perf_counter(); ix.search_batch(Q[i % 64:i % 64 + 1], k, m); ts.append(time.perf_counter() - t)
ts = np.array(ts) * 1e3
print("cfg1 single query: p50 %.3f ms  p10 %.3f  p90 %.3f" % (np.percentile(ts, 50), np.percentile(ts, 10), np.percentile(ts, 90)))
st = ix.stats(); print({k_: st[k_] for k_ in st if "ms" in k_ or "visits" in k_})
