#!/usr/bin/env python3
"""Where a wave of the matrix-core table scan (scan_mfma_kernel) spends its cycles.

Builds the library once more with -DZH_SCAN_PROF into tests/probes/_build/ (the shipped library never carries the
counters), runs bench.py's own loop for a workload in this process and prints, over every 16th wave of all launches, the
mean cycles in phase 1 (row -> leaf entries, visit records), the pair list, the column pass, the wait for the rows' tiles and
the first chunk of query lines, and the tile loop -- with pairs, columns and tiles per wave beside them.

    python tests/probes/scan_prof.py build      # here (hipcc cross-compiles), then
    gpurun -- python tests/probes/scan_prof.py [bench.py arguments, default: --workload cfg3]
"""
import ctypes as C
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "probes", "_build", "libzebra_hip_scanprof.so")


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    src = sorted(glob.glob(os.path.join(ROOT, "zebra_amd", "csrc", "*.hip"))) + [os.path.join(ROOT, "zebra_amd", "csrc", "zh_refformat.cpp")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-DZH_SCAN_PROF"] + os.environ.get("PROBE_DEFINES", "").split() + ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                           "-ffp-contract=off", "-fvisibility=hidden", "-Wno-unused-parameter", "-Wno-unused-value", "-shared",
                           "-Wl,--no-undefined", "-o", OUT] + src + ["-L/opt/rocm/lib", "-lrccl"])


if len(sys.argv) > 1 and sys.argv[1] == "build":
    build()
    sys.exit(0)

import numpy as np  # noqa: E402
from zebra_amd import _ffi  # noqa: E402

_ffi.LIB_PATH = OUT
import bench  # noqa: E402

args = sys.argv[1:] or ["--workload", "cfg3"]
sys.argv = ["bench.py"] + args + ["--cpu-seconds", "0", "--no-recall", "--no-other-configs"]
try:
    bench.main()
except SystemExit:
    pass
L = _ffi.lib()
L.zh_debug_scan_prof.restype = C.c_int
L.zh_debug_scan_prof.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
buf = np.zeros(16, dtype=np.uint64)
assert L.zh_debug_scan_prof(buf.ctypes.data, buf.size, 0) == 0
w = float(buf[0])
if w == 0:
    print("no sampled wave: the matrix-core scan did not run")
    sys.exit(1)
names = ["phase 1 (entries, bitmaps, visit records)", "pair list", "rows' tile loads issued + column pass", "wait: first query chunk (+ row tiles)", "tile loop"]
tot = float(buf[9]) / w
print("sampled waves %d; per wave: pairs %.1f, columns %.1f, tiles %.2f; cycles %.0f" % (w, buf[7] / w, buf[8] / w, buf[6] / w, tot))
for i, n in enumerate(names):
    c = float(buf[1 + i]) / w
    print("  %-46s %9.0f cycles  %5.1f %%" % (n, c, 100.0 * c / tot))
print("  per tile in the loop: %.0f cycles" % (float(buf[5]) / max(float(buf[6]), 1.0)))
