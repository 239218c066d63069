#!/usr/bin/env python3
"""hash_dense_kernel at GEMM scale: 2048 queries x every plane of a 200k x 768 index with max_node_size 16
(~20k planes), i.e. the 'hash everything densely' regime of the reference's small default leaves.  Run under
rocprofv3 --kernel-trace --stats to read the kernel time; prints the flop count of one call."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zebra_amd as za  # noqa: E402
import numpy as np  # noqa: E402

d = 768
ix = za.LSHIndex(d, za.LSHIndexOptions(16, 1))
ix.append_synthetic(200000)
ix.build()
P = ix.get_forest()["consts"].size
Q = np.random.default_rng(1).standard_normal((2048, d)).astype(np.float32)
for _ in range(3):
    ix.hash_signs(Q)
print("planes", P, "GFLOP per call", 2 * 2048 * P * d / 1e9)
