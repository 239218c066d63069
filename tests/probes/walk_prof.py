#!/usr/bin/env python3
"""Where a wave of the blocked walk (reference-default options: max_node_size 5, 15 trees) spends its cycles.

Builds the library once more with -DZH_WALK_PROF into tests/probes/_build/ (the shipped library never carries the
counters), runs the refdefault shape and prints, over the batch's (query, tree) pairs, the share of a wave's cycles in
block loads, upper-level steps, visit flushes and the in-register DFS, with the step counts beside them.

    python tests/probes/walk_prof.py build      # here (hipcc cross-compiles), then
    gpurun -- python tests/probes/walk_prof.py [rows] [batch]
"""
import ctypes as C
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "probes", "_build", "libzebra_hip_prof.so")


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    src = sorted(glob.glob(os.path.join(ROOT, "zebra_amd", "csrc", "*.hip"))) + [os.path.join(ROOT, "zebra_amd", "csrc", "zh_refformat.cpp")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-DZH_WALK_PROF"] + os.environ.get("PROBE_DEFINES", "").split() + [ "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                           "-ffp-contract=off", "-fvisibility=hidden", "-Wno-unused-parameter", "-Wno-unused-value", "-shared",
                           "-Wl,--no-undefined", "-o", OUT] + src + ["-L/opt/rocm/lib", "-lrccl"])


if len(sys.argv) > 1 and sys.argv[1] == "build":
    build()
    sys.exit(0)

import numpy as np  # noqa: E402
from zebra_amd import _ffi  # noqa: E402

_ffi.LIB_PATH = OUT
import zebra_amd as za  # noqa: E402
from oracle import zebra_oracle as zo  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
d, k, T = 384, 10, 15
X = zo.synth_rows(n, d)
Q = zo.synth_queries(B, d, n)
ix = za.LSHIndex(d, za.LSHIndexOptions(5, T))
ix.add(X)
m = za.L2SquaredDistance()
for _ in range(6):
    ids, keys, counts = ix.search_batch(Q, k, m)
L = _ffi.lib()
L.zh_debug_walk_prof.restype = C.c_int
L.zh_debug_walk_prof.argtypes = [C.c_void_p, C.c_uint32]
P = min(B * T, 8192)
buf = np.zeros((8192, 16), dtype=np.uint64)
assert L.zh_debug_walk_prof(buf.ctypes.data, buf.size) == 0
p = buf[:P].astype(np.float64)
tot, wall = p[:, 0], p[:, 1]
print("pairs", P, "cycles/wave mean %.0f max %.0f min %.0f; wall(100MHz ticks) mean %.0f max %.0f -> clk/tick %.2f" % (
    tot.mean(), tot.max(), tot.min(), wall.mean(), wall.max(), tot.sum() / wall.sum()))
start = buf[:P, 12].astype(np.int64)
end = start + buf[:P, 1].astype(np.int64)
print("launch span (us): first start -> last end %.1f; starts spread %.1f; median wave %.1f; longest wave %.1f" % (
    (end.max() - start.min()) / 100.0, (start.max() - start.min()) / 100.0, np.median(wall) / 100.0, wall.max() / 100.0))
for name, c in (("block load (records + sign gather)", 2), ("upper steps + pops", 3), ("flushes", 4), ("in-register DFS", 5)):
    print("  %-36s %5.1f %% of cycles" % (name, 100 * p[:, c].sum() / tot.sum()))
names = ["blocks", "upper steps", "inner steps", "in-block pops", "upper pops", "visits"]
for i, nm in enumerate(names):
    c = p[:, 6 + i]
    print("  %-14s per pair mean %9.1f max %9.0f" % (nm, c.mean(), c.max()))
print("  cycles per block load %.0f; per upper step %.0f; per flush(16 visits) %.0f; DFS cycles per (inner+visit+pop) %.1f" % (
    p[:, 2].sum() / p[:, 6].sum(), p[:, 3].sum() / max(p[:, 7].sum(), 1), p[:, 4].sum() / (p[:, 11].sum() / 16),
    p[:, 5].sum() / (p[:, 8].sum() + p[:, 11].sum() + p[:, 9].sum())))
print("  flagged signs resolved at block entry (inner-node blocks): per pair mean %.1f" % p[:, 14].mean())
hw = buf[:P, 13]
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
simd = (hw >> 4) & 0x3
key = (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 6) | (cu.astype(np.int64) << 2) | simd.astype(np.int64)
u, cnts = np.unique(key, return_counts=True)
print("  distinct (se, sh, cu, simd) seen (XCD not in HW_ID): %d; waves per such slot mean %.1f max %d" % (len(u), cnts.mean(), cnts.max()))
f = zo.Forest.from_arrays(X, 5, ix.get_forest())
for b in (0, B - 1):
    oi, ok = f.search(Q[b], k, zo.L2SQ)
    assert (ids[b] == oi).all() and (keys[b] == ok).all()
print("checked against the oracle")
