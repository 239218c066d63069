"""The VALUES of the reference's on-disk partitions -- host-side mirror of
`KeyValue::{upsert_embedding, upsert_tree, embedding}` (/root/reference/src/database/index/lsh.rs:91-119) over the
codec of zebra_amd/csrc/zh_refformat.cpp.  fjall itself (the LSM files) is the host shim's business; what crosses the
C ABI are the raw key / value bytes.

    embeddings partition   key = 16 uuid bytes, value = N little-endian f32           -> `decode_embeddings`
    trees partition        key = 16 uuid bytes, value = bincode(legacy) Node<N>      -> `decode_trees` / `encode_trees`
    the `.zebra` file      bincode(legacy) DatabaseInner (core.rs:19-29,183-190)     -> `decode_header` / `encode_header`
"""
import ctypes as C
import os

import numpy as np

from . import _ffi
from ._ffi import check

_san = None


def lib():
    """libzebra_hip.so -- or, for the sanitizer run of the CPU test-suite only (tests/test_sanitizers.py sets
    ZEBRA_REFFORMAT_SAN_LIB), this file's codec built stand-alone with gcc -fsanitize=address,undefined."""
    global _san
    path = os.environ.get("ZEBRA_REFFORMAT_SAN_LIB")
    if not path:
        return _ffi.lib()
    if _san is None:
        L = C.CDLL(path)
        for name, res, args in _ffi.SYMBOLS:
            if name.startswith("zh_ref_") or name == "zh_last_error":
                fn = getattr(L, name)
                fn.restype, fn.argtypes = res, args
        _ffi._lib = _ffi._lib or L  # check() reads zh_last_error through _ffi.lib()
        _san = L
    return _san

FOREST_KEYS = ("plane", "left", "right", "roots", "planes", "consts", "leaf_ids")


def _uuid_table(uuids):
    u = np.ascontiguousarray(np.frombuffer(b"".join(uuids), np.uint8) if isinstance(uuids, (list, tuple)) else uuids, np.uint8)
    return u.reshape(-1, 16)


def decode_embeddings(values, dim):
    """Embedding<N> values (lsh.rs:94: bincode legacy of [f32; N] = 4N bytes, no prefix) -> float32 [n, dim]."""
    out = np.empty((len(values), dim), np.float32)
    for i, v in enumerate(values):
        if len(v) != 4 * dim:
            raise ValueError(f"embedding {i}: {len(v)} bytes, expected {4 * dim}")
        out[i] = np.frombuffer(v, "<f4")
    return out


def encode_embeddings(rows):
    rows = np.ascontiguousarray(rows, "<f4")
    return [r.tobytes() for r in rows]


def decode_trees(values, dim, uuids):
    """Node<N> values of the trees partition -> (flat forest arrays for LSHIndex.set_forest, unknown id count).
    `uuids`: the keys of the stored vectors in row order ([n, 16] uint8 or a list of 16-byte strings)."""
    u = _uuid_table(uuids)
    bufs = [np.frombuffer(v, np.uint8) for v in values]
    ptrs = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * len(bufs))(*[b.size for b in bufs])
    h = C.c_void_p()
    unknown = C.c_uint64()
    check(lib().zh_ref_forest_decode(dim, len(bufs), ptrs, lens, u.shape[0], u.ctypes.data, C.byref(h), C.byref(unknown)))
    try:
        v = _ffi.ForestView()
        check(lib().zh_ref_forest_view(h, C.byref(v)))

        def arr(ptr, n, dt):
            if n == 0:
                return np.empty(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), (n,)).copy()
        out = dict(plane=arr(v.plane, v.n_nodes, np.int32), left=arr(v.left, v.n_nodes, np.int32),
                   right=arr(v.right, v.n_nodes, np.int32), roots=arr(v.roots, v.n_trees, np.uint32),
                   planes=arr(v.planes, v.n_planes * dim, np.float32).reshape(-1, dim),
                   consts=arr(v.consts, v.n_planes, np.float32), leaf_ids=arr(v.leaf_ids, v.n_leaf_ids, np.uint32))
    finally:
        lib().zh_ref_forest_free(h)
    return out, unknown.value


def encode_trees(forest, dim, uuids):
    """Flat forest arrays (LSHIndex.get_forest) -> one bincode(legacy) Node<N> value per tree."""
    u = _uuid_table(uuids)
    keep = [np.ascontiguousarray(forest["plane"], np.int32), np.ascontiguousarray(forest["left"], np.int32),
            np.ascontiguousarray(forest["right"], np.int32), np.ascontiguousarray(forest["roots"], np.uint32),
            np.ascontiguousarray(forest["planes"], np.float32).reshape(-1, dim), np.ascontiguousarray(forest["consts"], np.float32),
            np.ascontiguousarray(forest["leaf_ids"], np.uint32)]
    fv = _ffi.ForestView(keep[0].size, keep[5].size, keep[3].size, keep[6].size, *[x.ctypes.data for x in keep])
    out = []
    for t in range(keep[3].size):
        n = C.c_size_t()
        check(lib().zh_ref_tree_encode(C.byref(fv), dim, t, u.ctypes.data, u.shape[0], None, 0, C.byref(n)))
        buf = np.empty(n.value, np.uint8)
        check(lib().zh_ref_tree_encode(C.byref(fv), dim, t, u.ctypes.data, u.shape[0], buf.ctypes.data, buf.size, C.byref(n)))
        out.append(buf.tobytes())
    return out


METRIC_IDS = {"cosine": _ffi.COSINE, "l2sq": _ffi.L2SQ, "l2": _ffi.L2, "chebyshev": _ffi.CHEBYSHEV, "canberra": _ffi.CANBERRA,
              "bray_curtis": _ffi.BRAY_CURTIS, "manhattan": _ffi.MANHATTAN, "l3": _ffi.L3, "l4": _ffi.L4, "hamming": _ffi.HAMMING,
              "minkowski": _ffi.MINKOWSKI, "pnorm": _ffi.PNORM}


def decode_header(blob, metric="cosine", model_len=0):
    """The `.zebra` file (core.rs:92-102 `Database::open`): bincode(legacy) DatabaseInner { uuid, model, metric, index_options }.
    `metric` (a name of METRIC_IDS or a zh_metric id) and `model_len` stand for the crate's type parameters Met / Mod, which
    the file does not record.  -> dict(uuid=16 bytes, max_node_size, num_trees, metric, power, model=bytes)."""
    m = METRIC_IDS[metric] if isinstance(metric, str) else int(metric)
    buf = np.frombuffer(bytes(blob), np.uint8)
    h = _ffi.RefHeader()
    check(lib().zh_ref_header_decode(buf.ctypes.data if buf.size else None, buf.size, m, model_len, C.byref(h)))
    return dict(uuid=bytes(h.uuid), max_node_size=int(h.max_node_size), num_trees=int(h.num_trees), metric=int(h.metric),
                power=int(h.power), model=bytes(blob[h.model_off:h.model_off + h.model_len]))


def encode_header(uuid16, max_node_size=5, num_trees=15, metric="cosine", power=0, model=b""):
    """what `Database::save_database` (core.rs:183-190) writes; defaults = LSHIndexOptions::default (lsh.rs:131-138)"""
    m = METRIC_IDS[metric] if isinstance(metric, str) else int(metric)
    h = _ffi.RefHeader()
    h.uuid[:] = list(bytes(uuid16))
    h.max_node_size, h.num_trees, h.metric, h.power, h.model_off, h.model_len = max_node_size, num_trees, m, power, 24, len(model)
    mb = np.frombuffer(bytes(model), np.uint8)
    n = C.c_size_t()
    check(lib().zh_ref_header_encode(C.byref(h), mb.ctypes.data if mb.size else None, None, 0, C.byref(n)))
    out = np.empty(n.value, np.uint8)
    check(lib().zh_ref_header_encode(C.byref(h), mb.ctypes.data if mb.size else None, out.ctypes.data, out.size, C.byref(n)))
    return out.tobytes()
