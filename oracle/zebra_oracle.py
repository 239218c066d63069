"""ctypes binding of the CPU oracle (oracle/zebra_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under zebra_amd/ imports this module.  PARITY UNPINNED -- see zebra_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libzebra_oracle.so")

COSINE, L2SQ, L2 = 0, 1, 2
CHEBYSHEV, CANBERRA, BRAY_CURTIS, MANHATTAN, L3, L4, HAMMING, MINKOWSKI, PNORM = 3, 4, 5, 6, 7, 8, 9, 10, 11
PARITY, CORRECTED = 0, 1
SEED_ROWS, SEED_QUERIES, SEED_INDEX = 0x5EB2A001, 0x5EB2A002, 0x5EB2A003


def build(force=False):
    src = os.path.join(_HERE, "zebra_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


class Stats(C.Structure):
    _fields_ = [("rows_scored", C.c_uint64), ("planes_evaluated", C.c_uint64),
                ("leaves_visited", C.c_uint64), ("candidates", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        san = os.environ.get("ZEBRA_ORACLE_SAN_LIB")  # tests/test_sanitizers.py: the same source built with -fsanitize=address,undefined
        if not san:
            build()
        L = C.CDLL(san or _SO)
        vp, u64, u32, i32, f32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_float
        L.zo_synth_rows.argtypes = [u64, u64, u64, u32, C.c_int, vp]
        L.zo_synth_query_row.argtypes = [u64, u64, u64]
        L.zo_synth_query_row.restype = u64
        L.zo_synth_queries.argtypes = [u64, u64, u64, u64, u64, u32, C.c_int, vp]
        L.zo_dot32.argtypes = [vp, vp, u32]
        L.zo_dot32.restype = f32
        L.zo_point_is_above.argtypes = [vp, f32, vp, u32]
        L.zo_point_is_above.restype = C.c_int
        L.zo_distance.argtypes = [C.c_int, C.c_int, vp, vp, u32]
        L.zo_distance.restype = u64
        L.zo_distance_batch.argtypes = [C.c_int, C.c_int, vp, vp, u64, u32, vp]
        L.zo_distance_sums.argtypes = [vp, vp, u32, vp]
        L.zo_sample_pair.argtypes = [u64, u32, u64, u64, vp, vp]
        L.zo_make_hyperplane.argtypes = [vp, vp, u32, vp, vp]
        L.zo_forest_build.argtypes = [vp, u64, u32, u32, u32, u64]
        L.zo_forest_build.restype = vp
        L.zo_forest_from_arrays.argtypes = [u64, u32, u32, u32, u32, vp, vp, vp, vp, u32, vp, vp, u64, vp]
        L.zo_forest_from_arrays.restype = vp
        L.zo_forest_insert.argtypes = [vp, vp, u64, u64]
        L.zo_forest_remove.argtypes = [vp, vp, vp, u64, vp]
        L.zo_forest_remove.restype = u64
        L.zo_find_duplicates.argtypes = [vp, u64, u32, vp, vp]
        L.zo_find_duplicates.restype = u64
        L.zo_forest_free.argtypes = [vp]
        L.zo_forest_sizes.argtypes = [vp, vp, vp, vp]
        L.zo_forest_export.argtypes = [vp] * 8
        L.zo_search.argtypes = [vp, vp, vp, u32, C.c_int, C.c_int, vp, vp, vp]
        L.zo_search.restype = u32
        L.zo_search_batch.argtypes = [vp, vp, vp, u64, u32, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp]
        L.zo_forest_borrow_arrays.argtypes = [u64, u32, u32, u32, u32, vp, vp, vp, vp, u32, vp, vp, u64, vp]
        L.zo_forest_borrow_arrays.restype = vp
        L.zo_search_batch_synth.argtypes = [vp, u64, u64, C.c_int, vp, u64, u32, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp]
        L.zo_check_forest_synth.argtypes = [vp, u64, u64, u64, C.c_int, u64, vp]
        L.zo_check_forest_synth.restype = C.c_int
        L.zo_tree_result.argtypes = [vp, vp, u32, vp, i32, C.c_int, C.c_int, vp, vp, vp, u64, vp]
        L.zo_tree_result.restype = i32
        L.zo_hash_signs.argtypes = [vp, vp, vp, vp]
        L.zo_merge_topk.argtypes = [u32, u64, u32, vp, vp, vp, vp, vp, vp]
        L.zo_brute_force.argtypes = [vp, u64, u32, vp, u32, C.c_int, C.c_int, vp, vp]
        L.zo_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def synth_rows(n, d, seed=SEED_ROWS, row0=0, kind=0):
    out = np.empty((n, d), np.float32)
    lib().zo_synth_rows(seed, row0, n, d, kind, _p(out))
    return out


def synth_queries(b, d, n_rows, seed_rows=SEED_ROWS, seed_q=SEED_QUERIES, b0=0, kind=0):
    out = np.empty((b, d), np.float32)
    lib().zo_synth_queries(seed_rows, seed_q, n_rows, b0, b, d, kind, _p(out))
    return out


def synth_query_row(b, n_rows, seed_q=SEED_QUERIES):
    return int(lib().zo_synth_query_row(seed_q, b, n_rows))


def dot32(w, x):
    w, x = _f32(w), _f32(x)
    return float(lib().zo_dot32(_p(w), _p(x), w.size))


def point_is_above(w, c, x):
    w, x = _f32(w), _f32(x)
    return bool(lib().zo_point_is_above(_p(w), float(c), _p(x), w.size))


def distance(metric, mode, a, b):
    a, b = _f32(a), _f32(b)
    return int(lib().zo_distance(metric, mode, _p(a), _p(b), a.size))


def distance_batch(metric, mode, rows, q):
    rows, q = _f32(rows), _f32(q)
    out = np.empty(rows.shape[0], np.uint64)
    lib().zo_distance_batch(metric, mode, _p(rows), _p(q), rows.shape[0], rows.shape[1], _p(out))
    return out


def distance_sums(a, b):
    a, b = _f32(a), _f32(b)
    out = np.empty(4, np.float32)
    lib().zo_distance_sums(_p(a), _p(b), a.size, _p(out))
    return out  # ab, a2, b2, l2sq


def sample_pair(seed, tree, path, n_rows):
    i, j = C.c_uint64(), C.c_uint64()
    lib().zo_sample_pair(seed, tree, path, n_rows, C.byref(i), C.byref(j))
    return i.value, j.value


def make_hyperplane(a, b):
    a, b = _f32(a), _f32(b)
    w = np.empty_like(a)
    c = C.c_float()
    lib().zo_make_hyperplane(_p(a), _p(b), a.size, _p(w), C.byref(c))
    return w, np.float32(c.value)


def key_to_float(keys):
    return np.asarray(keys, dtype=np.uint64).view(np.float64)


class Forest:
    """Flat forest: node i is inner when plane[i] >= 0 (left = below child, right = above child),
    a leaf when plane[i] == -1 (left = offset into leaf_ids, right = length)."""

    def __init__(self, handle, X, n_rows, d, M, T):
        self._h = handle
        self.X = X
        self.n_rows, self.d, self.M, self.T = n_rows, d, M, T

    @classmethod
    def build(cls, X, M, T, seed=SEED_INDEX):
        X = _f32(X)
        n, d = X.shape
        h = lib().zo_forest_build(_p(X), n, d, M, T, seed)
        return cls(h, X, n, d, M, T)

    @classmethod
    def from_arrays(cls, X, M, arrays):
        X = _f32(X)
        n, d = X.shape
        a = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
        plane, left, right = (a[k].astype(np.int32) for k in ("plane", "left", "right"))
        roots = a["roots"].astype(np.uint32)
        planes, consts = _f32(a["planes"]).reshape(-1, d), _f32(a["consts"])
        leaf_ids = a["leaf_ids"].astype(np.uint32)
        h = lib().zo_forest_from_arrays(n, d, M, roots.size, plane.size, _p(plane), _p(left), _p(right), _p(roots),
                                        consts.size, _p(planes), _p(consts), leaf_ids.size, _p(leaf_ids))
        return cls(h, X, n, d, M, roots.size)

    @classmethod
    def borrow_synth(cls, n_rows, d, M, arrays, seed_rows=SEED_ROWS, first_row=0, kind=0):
        """A forest exported by the HIP build (LSHIndex.get_forest()) over rows that exist only as the counter
        generator (seed_rows, first_row + id, kind): no copy of the arrays (GBs of leaf ids at full size), no X.
        Only search_batch_synth / check_synth may be used on it."""
        a = dict(plane=np.ascontiguousarray(arrays["plane"], np.int32), left=np.ascontiguousarray(arrays["left"], np.int32),
                 right=np.ascontiguousarray(arrays["right"], np.int32), roots=np.ascontiguousarray(arrays["roots"], np.uint32),
                 planes=np.ascontiguousarray(arrays["planes"], np.float32).reshape(-1, d),
                 consts=np.ascontiguousarray(arrays["consts"], np.float32),
                 leaf_ids=np.ascontiguousarray(arrays["leaf_ids"], np.uint32))
        h = lib().zo_forest_borrow_arrays(n_rows, d, M, a["roots"].size, a["plane"].size, _p(a["plane"]), _p(a["left"]),
                                          _p(a["right"]), _p(a["roots"]), a["consts"].size, _p(a["planes"]), _p(a["consts"]),
                                          a["leaf_ids"].size, _p(a["leaf_ids"]))
        f = cls(h, None, n_rows, d, M, a["roots"].size)
        f._keep = a  # the borrowed arrays must outlive the forest
        f._synth = (seed_rows, first_row, kind)
        return f

    def search_batch_synth(self, Q, k, metric, mode=PARITY, nthreads=0, stats=False):
        """LSHIndex::search for every query with the stored rows regenerated on demand -> LOCAL ids, keys, counts"""
        Q = _f32(Q)
        b = Q.shape[0]
        ids, keys = np.zeros((b, k), np.uint64), np.zeros((b, k), np.uint64)
        counts = np.zeros(b, np.uint32)
        st = Stats()
        seed, row0, kind = self._synth
        lib().zo_search_batch_synth(self._h, seed, row0, kind, _p(Q), b, k, metric, mode, _p(ids), _p(keys), _p(counts),
                                    nthreads, C.byref(st))
        if stats:
            return ids, keys, counts, st
        return ids, keys, counts

    def check_synth(self, index_seed=SEED_INDEX, n_sample=64):
        """build rules of lsh.rs:192-267,411-429 on a forest built elsewhere: 0 = holds; (code, planes re-derived)"""
        seed, row0, kind = self._synth
        n = C.c_uint64()
        rc = lib().zo_check_forest_synth(self._h, index_seed, seed, row0, kind, n_sample, C.byref(n))
        return rc, n.value

    def insert(self, X_all, n_prev):
        """LSHIndex::add on an index that already has trees (lsh.rs:445-462): X_all = old rows + new rows"""
        X_all = _f32(X_all)
        lib().zo_forest_insert(self._h, _p(X_all), n_prev, X_all.shape[0] - n_prev)
        self.X = X_all
        self.n_rows = X_all.shape[0]

    def remove(self, ids):
        """LSHIndex::remove as intended (the id leaves every tree) -> bool array: was it present"""
        ids = np.ascontiguousarray(ids, np.uint64)
        found = np.zeros(ids.size, np.uint8)
        lib().zo_forest_remove(self._h, _p(self.X), _p(ids), ids.size, _p(found))
        return found.astype(bool)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().zo_forest_free(self._h)
            self._h = None

    def arrays(self):
        nn, npl, nl = C.c_uint32(), C.c_uint32(), C.c_uint64()
        lib().zo_forest_sizes(self._h, C.byref(nn), C.byref(npl), C.byref(nl))
        out = dict(plane=np.empty(nn.value, np.int32), left=np.empty(nn.value, np.int32),
                   right=np.empty(nn.value, np.int32), roots=np.empty(self.T, np.uint32),
                   planes=np.empty((npl.value, self.d), np.float32), consts=np.empty(npl.value, np.float32),
                   leaf_ids=np.empty(nl.value, np.uint32))
        lib().zo_forest_export(self._h, _p(out["plane"]), _p(out["left"]), _p(out["right"]), _p(out["roots"]),
                               _p(out["planes"]), _p(out["consts"]), _p(out["leaf_ids"]))
        return out

    def search(self, q, k, metric, mode=PARITY, stats=False):
        q = _f32(q)
        ids, keys = np.zeros(k, np.uint64), np.zeros(k, np.uint64)
        st = Stats()
        m = lib().zo_search(self._h, _p(self.X), _p(q), k, metric, mode, _p(ids), _p(keys), C.byref(st))
        if stats:
            return ids[:m], keys[:m], st
        return ids[:m], keys[:m]

    def search_batch(self, Q, k, metric, mode=PARITY, nthreads=0, stats=False):
        Q = _f32(Q)
        b = Q.shape[0]
        ids, keys = np.zeros((b, k), np.uint64), np.zeros((b, k), np.uint64)
        counts = np.zeros(b, np.uint32)
        st = Stats()
        lib().zo_search_batch(self._h, _p(self.X), _p(Q), b, k, metric, mode, _p(ids), _p(keys), _p(counts),
                              nthreads, C.byref(st))
        if stats:
            return ids, keys, counts, st
        return ids, keys, counts

    def tree_result(self, tree, q, n, metric, mode=PARITY, cap_visits=1 << 16):
        q = _f32(q)
        cand = np.empty(max(self.n_rows, 1), np.uint32)
        visits = np.zeros((cap_visits, 3), np.uint64)
        nc, nv = C.c_uint64(), C.c_uint64()
        r = lib().zo_tree_result(self._h, _p(self.X), tree, _p(q), n, metric, mode, _p(cand), C.byref(nc),
                                 _p(visits), cap_visits, C.byref(nv))
        return int(r), cand[:nc.value].copy(), visits[:nv.value].copy()

    def hash_signs(self, q):
        q = _f32(q)
        a = self.arrays()
        signs = np.empty(a["consts"].size, np.uint8)
        dots = np.empty(a["consts"].size, np.float32)
        lib().zo_hash_signs(self._h, _p(q), _p(signs), _p(dots))
        return signs, dots


def merge_topk(ids, keys, counts, k):
    """ids/keys: [S, b, k] uint64, counts: [S, b] uint32 -> merged ([b,k],[b,k],[b])"""
    ids = np.ascontiguousarray(ids, np.uint64)
    keys = np.ascontiguousarray(keys, np.uint64)
    counts = np.ascontiguousarray(counts, np.uint32)
    S, b, _ = ids.shape
    oi, ok, oc = np.zeros((b, k), np.uint64), np.zeros((b, k), np.uint64), np.zeros(b, np.uint32)
    lib().zo_merge_topk(S, b, k, _p(ids), _p(keys), _p(counts), _p(oi), _p(ok), _p(oc))
    return oi, ok, oc


def brute_force(X, q, k, metric, mode=PARITY):
    X, q = _f32(X), _f32(q)
    k = min(k, X.shape[0])
    ids, keys = np.zeros(k, np.uint64), np.zeros(k, np.uint64)
    lib().zo_brute_force(_p(X), X.shape[0], X.shape[1], _p(q), k, metric, mode, _p(ids), _p(keys))
    return ids, keys


def find_duplicates(X, alive=None):
    """LSHIndex::deduplicate's rule: rows bit-identical to an earlier live row -> bool array"""
    X = _f32(X)
    out = np.zeros(X.shape[0], np.uint8)
    al = None if alive is None else np.ascontiguousarray(alive, np.uint8)
    lib().zo_find_duplicates(_p(X), X.shape[0], X.shape[1], _p(al) if al is not None else None, _p(out))
    return out.astype(bool)


def num_threads():
    return int(lib().zo_num_threads())


def canonical_forest(arr, d):
    """Order-independent description of a flat forest: per tree, a nested tuple
    (plane bytes, const, below, above) / sorted leaf ids.  Used to compare a forest built by the
    HIP path with the oracle's, whatever order either stored its nodes in."""
    import hashlib

    plane, left, right = arr["plane"], arr["left"], arr["right"]
    planes, consts, leaf_ids = np.asarray(arr["planes"]).reshape(-1, d), arr["consts"], arr["leaf_ids"]

    def rec(i):
        # iterative post-order to survive deep trees
        stack, out = [(i, 0)], {}
        while stack:
            n, st = stack.pop()
            if plane[n] < 0:
                ids = np.sort(leaf_ids[left[n]:left[n] + right[n]]).astype(np.uint32)
                out[n] = hashlib.sha1(b"L" + ids.tobytes()).digest()
            elif st == 0:
                stack.append((n, 1))
                stack.append((int(left[n]), 0))
                stack.append((int(right[n]), 0))
            else:
                h = hashlib.sha1(b"I" + planes[plane[n]].tobytes() + np.float32(consts[plane[n]]).tobytes()
                                 + out.pop(int(left[n])) + out.pop(int(right[n])))
                out[n] = h.digest()
        return out[i]

    return [rec(int(r)).hex() for r in arr["roots"]]
