"""Blocking host-pointer calls (zh_search_batch) in the reference-default regime (1M x 384, max_node_size 5, 15 trees, L2^2 top-10): queries per
second by queries per call and ZH_HOST_LOOKAHEAD (read per call).   python profiles/probe_host_calls.py  > gpurun_out/probe_host_calls.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zebra_amd as za

n, d, M, T, k = 1_000_000, 384, 5, 15, 10
ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
ix.append_synthetic(n)
ix.build()
m = za.L2SquaredDistance()
dq = torch.empty((8192, d), dtype=torch.float32, device="cuda:0")
za.synth_queries_device(0, dq.data_ptr(), n, 8192, d)
torch.cuda.synchronize()
Q = dq.cpu().numpy()
ix.search_batch(Q[:256], k, m)
for la in ("2", "1", "0"):
    os.environ["ZH_HOST_LOOKAHEAD"] = la
    for per_call in (512, 1024, 2048, 4096, 8192):
        q = Q[:per_call]
        ix.search_batch(q, k, m)
        reps = max(2, 4096 // per_call)
        t0 = time.perf_counter()
        for _ in range(reps):
            ix.search_batch(q, k, m)
        dt = (time.perf_counter() - t0) / reps
        print("lookahead %s  %5d queries per call: %7.2f ms per call, %8.0f queries/s" % (la, per_call, dt * 1e3, per_call / dt), flush=True)
os.environ["ZH_HOST_LOOKAHEAD"] = "2"
for hw in (512, 1024):
    os.environ["ZH_HOST_WINDOW"] = str(hw)
    for per_call in (1024, 2048, 4096, 8192):
        q = Q[:per_call]
        ix.search_batch(q, k, m)
        reps = max(2, 4096 // per_call)
        t0 = time.perf_counter()
        for _ in range(reps):
            ix.search_batch(q, k, m)
        dt = (time.perf_counter() - t0) / reps
        print("lookahead 2, windows of %4d  %5d queries per call: %7.2f ms per call, %8.0f queries/s" % (hw, per_call, dt * 1e3, per_call / dt), flush=True)
os.environ.pop("ZH_HOST_WINDOW")
for nb in (512, 1024):
    ix.set_profiling(1)
    ix.stats(reset=True)
    ix.search_batch(Q[:nb], k, m) if False else None
ix.set_profiling(1)
ix.stats(reset=True)
ix.search_batch(Q[:256], k, m)
st = ix.stats()
print({x: round(st[x], 3) for x in ("ms_hash", "ms_walk", "ms_sweep", "ms_select", "ms_final", "ms_total")})
