"""ctypes binding of oracle/zebra_cpu_fast.cpp: the reference algorithm on the CPU with free summation order -- the
`port-fast` leg of bench.py's cpu_baseline.  TEST / MEASUREMENT INFRASTRUCTURE ONLY (never imported by zebra_amd/)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
COSINE, L2SQ, L2 = 0, 1, 2
_lib = None


def _has_avx512():
    try:
        return " avx512f " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        return False


def lib():
    global _lib
    if _lib is None:
        name = "libzebra_cpu_fast_v4.so" if _has_avx512() else "libzebra_cpu_fast_v3.so"
        so = os.path.join(_HERE, name)
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        L = C.CDLL(so)
        vp = C.c_void_p
        L.zf_search_batch.argtypes = [vp, vp, vp, vp, C.c_uint32, vp, vp, vp, vp, C.c_uint32, vp, C.c_uint64, C.c_uint32,
                                      C.c_int, C.c_int, vp, vp, vp, C.c_int, vp]
        L.zf_num_threads.restype = C.c_int
        L.zf_isa.restype = C.c_char_p
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def isa():
    return lib().zf_isa().decode()


def num_threads():
    return int(lib().zf_num_threads())


class FastForest:
    """forest arrays (LSHIndex.get_forest() / zo.Forest.arrays()) + rows in host memory, borrowed"""

    def __init__(self, X, arrays):
        self.X = np.ascontiguousarray(X, np.float32)
        self.d = self.X.shape[1]
        a = arrays
        self.a = dict(plane=np.ascontiguousarray(a["plane"], np.int32), left=np.ascontiguousarray(a["left"], np.int32),
                      right=np.ascontiguousarray(a["right"], np.int32), roots=np.ascontiguousarray(a["roots"], np.uint32),
                      planes=np.ascontiguousarray(a["planes"], np.float32), consts=np.ascontiguousarray(a["consts"], np.float32),
                      leaf_ids=np.ascontiguousarray(a["leaf_ids"], np.uint32))

    def search_batch(self, Q, k, metric, mode=0, nthreads=0):
        Q = np.ascontiguousarray(Q, np.float32)
        b = Q.shape[0]
        ids = np.full((b, k), 2**64 - 1, np.uint64)
        keys = np.full((b, k), 2**64 - 1, np.uint64)
        counts = np.zeros(b, np.uint32)
        rows = C.c_uint64()
        a = self.a
        lib().zf_search_batch(_p(a["plane"]), _p(a["left"]), _p(a["right"]), _p(a["roots"]), a["roots"].size, _p(a["planes"]),
                              _p(a["consts"]), _p(a["leaf_ids"]), _p(self.X), self.d, _p(Q), b, k, metric, mode, _p(ids),
                              _p(keys), _p(counts), nthreads, C.byref(rows))
        return ids, keys, counts, rows.value
