"""Host-side mirror of the reference's metric / index / database interface for the hot path.

Names, argument meaning and error behaviour follow the reference (file:line in each docstring);
all arithmetic happens in libzebra_hip.so on the GPU."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _ffi
from ._ffi import check, lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a, d=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if d is not None and (a.ndim != 2 or a.shape[1] != d):
        raise ValueError(f"expected an (n, {d}) float32 array, got {a.shape}")
    return a


# ---------------------------------------------------------------------------- src/distance.rs
class _Metric:
    """space::Metric<Embedding<N>> with Unit = u64 (src/distance.rs:19-21)."""
    metric = None
    mode = _ffi.COSINE_PARITY

    def __init__(self, device=-1):
        self.device = device

    def distance(self, a, b):
        """Metric::distance(a, b) -> DistanceUnit: the f64 bit pattern as u64."""
        a, b = _f32(a).ravel(), _f32(b).ravel()
        if a.size != b.size:
            raise ValueError("vectors differ in length")
        out = C.c_uint64()
        check(lib().zh_distance_pair(self.metric, self.mode, _p(a), _p(b), a.size, C.byref(out), self.device))
        return int(out.value)

    def distance_batch(self, rows, query):
        rows, query = _f32(rows), _f32(query).ravel()
        out = np.empty(rows.shape[0], np.uint64)
        check(lib().zh_distance_batch(self.metric, self.mode, _p(rows), _p(query), rows.shape[0], rows.shape[1],
                                      _p(out), self.device))
        return out


class CosineDistance(_Metric):
    """src/distance.rs:15-32.  `parity=True` (default) keeps the reference's literal key
    bits(1.0 - simsimd_cosine) = bits of the SIMILARITY; `parity=False` keys on the distance."""
    metric = _ffi.COSINE

    def __init__(self, parity=True, device=-1):
        super().__init__(device)
        self.mode = _ffi.COSINE_PARITY if parity else _ffi.COSINE_CORRECTED


class L2SquaredDistance(_Metric):
    """src/distance.rs:34-49"""
    metric = _ffi.L2SQ


class L2Distance(_Metric):
    """src/distance.rs:99-114"""
    metric = _ffi.L2


class ChebyshevDistance(_Metric):
    """src/distance.rs:51-61"""
    metric = _ffi.CHEBYSHEV


class CanberraDistance(_Metric):
    """src/distance.rs:63-73"""
    metric = _ffi.CANBERRA


class BrayCurtisDistance(_Metric):
    """src/distance.rs:75-85"""
    metric = _ffi.BRAY_CURTIS


class ManhattanDistance(_Metric):
    """src/distance.rs:87-97"""
    metric = _ffi.MANHATTAN


class L3Distance(_Metric):
    """src/distance.rs:116-126"""
    metric = _ffi.L3


class L4Distance(_Metric):
    """src/distance.rs:128-138"""
    metric = _ffi.L4


class HammingDistance(_Metric):
    """src/distance.rs:140-158: bitwise Hamming distance of the low bytes of the f32 bit patterns"""
    metric = _ffi.HAMMING


class MinkowskiDistance(_Metric):
    """src/distance.rs:160-174; `power` is the struct's i32 field (#[derive(Default)]: 0, what `Met::default()` of
    core.rs:115,146 constructs; any i32 is legal)"""
    metric = _ffi.MINKOWSKI

    def __init__(self, power=0, device=-1):
        super().__init__(device)
        self.power = self.mode = int(power)


class PNormDistance(_Metric):
    """src/distance.rs:176-190; default power 0 as the reference's derived Default"""
    metric = _ffi.PNORM

    def __init__(self, power=0, device=-1):
        super().__init__(device)
        self.power = self.mode = int(power)


# ------------------------------------------------------------------- src/database/index/lsh.rs
@dataclass
class LSHIndexOptions:
    """lsh.rs:122-139; defaults max_node_size = 5, num_trees = 15."""
    max_node_size: int = 5
    num_trees: int = 15


class LSHIndex:
    """LSHIndex<N> (lsh.rs:144-565) for the hot path: new / add / search / is_empty / no_vectors /
    no_trees / clear, plus search_batch (the loop of core.rs:299-303 as one call)."""

    def __init__(self, dim, options=None, seed=0x5EB2A003, device=-1, id_base=0, reserve_rows=0):
        options = options or LSHIndexOptions()
        o = _ffi.Options()
        lib().zh_options_default(C.byref(o))
        o.dim, o.max_node_size, o.num_trees = dim, options.max_node_size, options.num_trees
        o.seed, o.device, o.id_base, o.reserve_rows = seed, device, id_base, reserve_rows
        self._h = C.c_void_p()
        check(lib().zh_index_create(C.byref(o), C.byref(self._h)))
        self.dim, self.options, self.id_base = dim, options, id_base

    def close(self):
        if getattr(self, "_h", None):
            # contexts (and shard groups) hold a pointer to the index: they go first, whatever order the garbage collector would pick
            for c in list(getattr(self, "_children", ())):
                c.close()
            lib().zh_index_destroy(self._h)
            self._h = None

    __del__ = close

    def _adopt(self, child):
        import weakref
        if not hasattr(self, "_children"):
            self._children = weakref.WeakSet()
        self._children.add(child)

    # lsh.rs:389-409
    def no_vectors(self):
        return lib().zh_index_count(self._h) == 0

    def no_trees(self):
        return lib().zh_index_num_trees(self._h) == 0

    def is_empty(self):
        return self.no_vectors() or self.no_trees()

    def __len__(self):
        return int(lib().zh_index_count(self._h))

    def add(self, embeddings):
        """lsh.rs:440-466: returns the ids of the added vectors (dense row ids, not Uuids)."""
        e = _f32(embeddings, self.dim)
        ids = np.empty(e.shape[0], np.uint64)
        check(lib().zh_index_add(self._h, _p(e), e.shape[0], _p(ids)))
        return ids

    def append(self, embeddings):
        e = _f32(embeddings, self.dim)
        ids = np.empty(e.shape[0], np.uint64)
        check(lib().zh_index_append(self._h, _p(e), e.shape[0], _p(ids)))
        return ids

    def append_synthetic(self, n, seed=0x5EB2A001, first_row=0, kind=0):
        check(lib().zh_index_append_synthetic(self._h, n, seed, first_row, kind))

    def build(self):
        check(lib().zh_index_build(self._h))

    def clear(self):
        check(lib().zh_index_clear(self._h))

    def remove(self, embedding_ids):
        """lsh.rs:473-503 (as intended: the ids leave every tree) -> the ids that were present"""
        ids = np.ascontiguousarray(embedding_ids, np.uint64)
        found = np.zeros(ids.size, np.uint8)
        n = C.c_size_t()
        check(lib().zh_index_remove(self._h, _p(ids), ids.size, _p(found), C.byref(n)))
        return ids[found.astype(bool)]

    def deduplicate(self):
        """lsh.rs:270-288 -> ids removed because an earlier vector has the same bits"""
        out = np.zeros(max(len(self), 1), np.uint64)
        n = C.c_size_t()
        check(lib().zh_index_deduplicate(self._h, _p(out), out.size, C.byref(n)))
        return out[:n.value]

    def set_forest(self, arrays):
        a = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
        keep = [a["plane"].astype(np.int32), a["left"].astype(np.int32), a["right"].astype(np.int32),
                a["roots"].astype(np.uint32), _f32(a["planes"]).reshape(-1, self.dim), _f32(a["consts"]),
                a["leaf_ids"].astype(np.uint32)]
        fv = _ffi.ForestView(keep[0].size, keep[5].size, keep[3].size, keep[6].size, *[_p(x).value for x in keep])
        check(lib().zh_index_set_forest(self._h, C.byref(fv)))

    def get_forest(self):
        s = _ffi.ForestSizes()
        check(lib().zh_index_forest_sizes(self._h, C.byref(s)))
        out = dict(plane=np.empty(s.n_nodes, np.int32), left=np.empty(s.n_nodes, np.int32),
                   right=np.empty(s.n_nodes, np.int32), roots=np.empty(s.n_trees, np.uint32),
                   planes=np.empty((s.n_planes, self.dim), np.float32), consts=np.empty(s.n_planes, np.float32),
                   leaf_ids=np.empty(s.n_leaf_ids, np.uint32))
        check(lib().zh_index_get_forest(self._h, *[_p(out[k]) for k in
                                                  ("plane", "left", "right", "roots", "planes", "consts", "leaf_ids")]))
        return out

    def hash_signs(self, queries, dots=False):
        """point_is_above (lsh.rs:39-43) of every plane for every query -> bool array [b, n_planes]."""
        q = _f32(queries, self.dim)
        s = _ffi.ForestSizes()
        check(lib().zh_index_forest_sizes(self._h, C.byref(s)))
        words = (s.n_planes + 31) // 32
        bits = np.zeros((q.shape[0], max(words, 1)), np.uint32)
        dd = np.zeros((q.shape[0], s.n_planes), np.float32) if dots else None
        check(lib().zh_hash_signs(self._h, _p(q), q.shape[0], _p(bits), _p(dd) if dots else None))
        signs = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :s.n_planes].astype(bool)
        return (signs, dd) if dots else signs

    def search_batch(self, queries, top_k, metric):
        """LSHIndex::search for every query: (ids [b,k] u64, keys [b,k] u64, counts [b] u32);
        entries past counts[i] are 2^64-1."""
        q = _f32(queries, self.dim)
        b = q.shape[0]
        ids = np.empty((b, top_k), np.uint64)
        keys = np.empty((b, top_k), np.uint64)
        counts = np.zeros(b, np.uint32)
        check(lib().zh_search_batch(self._h, _p(q), b, top_k, metric.metric, metric.mode, _p(ids), _p(keys), _p(counts)))
        return ids, keys, counts

    def search(self, query, top_k, metric):
        """lsh.rs:544-565 -> list of (id, distance key), ascending."""
        ids, keys, counts = self.search_batch(_f32(query).reshape(1, -1), top_k, metric)
        n = int(counts[0])
        return list(zip(ids[0, :n].tolist(), keys[0, :n].tolist()))

    def search_batch_device(self, d_q_ptr, b, top_k, metric, d_ids_ptr, d_keys_ptr, d_counts_ptr, stream=None):
        """Queries and results already in device memory (raw pointers, e.g. torch .data_ptr())."""
        check(lib().zh_search_batch_device(self._h, d_q_ptr, b, top_k, metric.metric, metric.mode, d_ids_ptr,
                                           d_keys_ptr, d_counts_ptr, stream))

    def debug_keep_raw(self, on=True):
        """tests: half-width batches keep a copy of the scan's raw pairs (zh_debug_keep_raw)"""
        check(lib().zh_debug_keep_raw(self._h, 1 if on else 0))

    def debug_scan_pairs(self, ctx=None):
        """tests: every scored (row, query) pair of the most recent half-width batch of `ctx` (None: the blocking context, i.e. the last
        search_batch_device call) as the scan made it -> (info dict, structured array of zh_debug_pair, qmeta [queries x 4]); zh_debug_scan_pairs"""
        info = _ffi.DebugScanInfo()
        h = ctx._h if ctx is not None else None
        check(lib().zh_debug_scan_pairs(self._h, h, C.byref(info), None, 0, None))
        dt = np.dtype([("row", "<u4"), ("query", "<u4"), ("lo", "<u4"), ("hi", "<u4"), ("raw_s", "<f4"), ("raw_a2", "<f4"), ("flags", "<u4"), ("visit", "<u4")])
        assert dt.itemsize == C.sizeof(_ffi.DebugPair)
        pairs = np.zeros(int(info.pairs), dt)
        qmeta = np.zeros((int(info.queries), 4), np.float32)
        if info.approx_scan:
            check(lib().zh_debug_scan_pairs(self._h, h, C.byref(info), pairs.ctypes.data_as(C.c_void_p), pairs.shape[0], qmeta.ctypes.data_as(C.c_void_p)))
        return info.as_dict(), pairs, qmeta

    def read_rows(self, first, n):
        """KeyValue::embedding (lsh.rs:107-119) for a run of rows."""
        out = np.empty((n, self.dim), np.float32)
        check(lib().zh_index_read_rows(self._h, first, n, _p(out)))
        return out

    def search_context(self):
        """one in-flight batch (zh_search_begin / finish / wait); create two to software-pipeline batches"""
        return SearchContext(self)

    def sweep_stream(self):
        """the index's lowest-priority stream (raw hipStream_t) for SearchContext.finish(..., sweep_stream=)"""
        return lib().zh_index_sweep_stream(self._h)

    def rows_device_ptr(self):
        return lib().zh_index_rows_device(self._h)

    def set_profiling(self, level):
        check(lib().zh_set_profiling(self._h, level))

    def set_dense_levels(self, levels):
        check(lib().zh_set_dense_levels(self._h, levels))

    def set_sweep_mode(self, mode):
        """0 = chosen per batch (default: a batch hashed from row scores is prefiltered, not swept), 1 = leaf by leaf, 2 = table
        scan with f32 queries, 3 = as 0, 4 = table scan with half-width queries wherever it applies, 5 = as 4 with the VALU kernel only, no fp16 copy of the rows, 6 = leaf by leaf at half width where implemented: dim 128 (zh_set_sweep_mode)"""
        check(lib().zh_set_sweep_mode(self._h, {"auto": 0, "leaf": 1, "scan": 2, "prefilter": 3, "approx": 4, "approx-valu": 5, "leaf-half": 6}.get(mode, mode)))

    def set_hash_mode(self, mode):
        """0 = chosen per batch (default), 1 = one dot product per plane, 2 = from row scores (zh_set_hash_mode)"""
        check(lib().zh_set_hash_mode(self._h, {"auto": 0, "dense": 1, "scores": 2}.get(mode, mode)))

    def stats(self, reset=False):
        s = _ffi.Stats()
        check(lib().zh_stats(self._h, C.byref(s)))
        if reset:
            check(lib().zh_stats_reset(self._h))
        return s.as_dict()


class SearchContext:
    """Pipelined search (include/zebra_hip.h, zh_search_begin/finish/wait) on raw device pointers."""

    def __init__(self, index):
        self._index = index  # keeps the index alive
        self._h = C.c_void_p()
        check(lib().zh_search_ctx_create(index._h, C.byref(self._h)))
        index._adopt(self)

    def begin(self, d_q_ptr, b, top_k, metric, stream=None):
        check(lib().zh_search_begin(self._h, d_q_ptr, b, top_k, metric.metric, metric.mode, stream))

    def finish(self, d_ids_ptr, d_keys_ptr, d_counts_ptr, sweep_stream=None):
        check(lib().zh_search_finish(self._h, d_ids_ptr, d_keys_ptr, d_counts_ptr, sweep_stream))

    def begin_window(self, d_q_ptrs, b, top_k, metric, stream=None):
        """len(d_q_ptrs) batches of b queries handled as one internal batch (zh_search_begin_window)"""
        arr = (C.c_void_p * len(d_q_ptrs))(*d_q_ptrs)
        check(lib().zh_search_begin_window(self._h, arr, len(d_q_ptrs), b, top_k, metric.metric, metric.mode, stream))

    def finish_window(self, d_ids_ptrs, d_keys_ptrs, d_counts_ptrs, sweep_stream=None):
        n = len(d_ids_ptrs)
        a, b_, c = (C.c_void_p * n)(*d_ids_ptrs), (C.c_void_p * n)(*d_keys_ptrs), (C.c_void_p * n)(*d_counts_ptrs)
        check(lib().zh_search_finish_window(self._h, a, b_, c, sweep_stream))

    def wait(self):
        check(lib().zh_search_wait(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib().zh_search_ctx_destroy(self._h)
            self._h = None

    __del__ = close


UNIQUE_ID_BYTES = 128


def shard_unique_id():
    """rank 0: the 128-byte id every rank passes to ShardGroup (ncclGetUniqueId inside the library)"""
    buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
    check(lib().zh_shard_unique_id(buf))
    return bytes(buf)


class ShardGroup:
    """This rank's shard (an LSHIndex over its rows, id_base = its first global row) joined with the other ranks'
    through RCCL inside libzebra_hip.so: search_batch over the WHOLE sharded index is one call per batch on every rank
    (local search -> one all-gather of the packed top-k -> merge), the loop of core.rs:299-303 for sharded rows."""

    def __init__(self, index, unique_id, n_ranks, rank):
        self.index = index  # borrowed by the group: keep it alive
        self._h = C.c_void_p()
        uid = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        check(lib().zh_shard_group_create(index._h, uid, n_ranks, rank, C.byref(self._h)))

    def ranks(self):
        return int(lib().zh_shard_group_ranks(self._h))

    def rank(self):
        return int(lib().zh_shard_group_rank(self._h))

    def search_batch(self, queries, top_k, metric):
        """merged global top-k of every query: (ids [b,k] u64, keys [b,k] u64, counts [b] u32)"""
        q = _f32(queries, self.index.dim)
        b = q.shape[0]
        ids = np.empty((b, top_k), np.uint64)
        keys = np.empty((b, top_k), np.uint64)
        counts = np.zeros(b, np.uint32)
        check(lib().zh_shard_search_batch(self._h, _p(q), b, top_k, metric.metric, metric.mode, _p(ids), _p(keys), _p(counts)))
        return ids, keys, counts

    def search_batch_device(self, d_q_ptr, b, top_k, metric, d_ids_ptr, d_keys_ptr, d_counts_ptr):
        check(lib().zh_shard_search_batch_device(self._h, d_q_ptr, b, top_k, metric.metric, metric.mode, d_ids_ptr,
                                                 d_keys_ptr, d_counts_ptr))

    def search_context(self):
        return ShardContext(self)

    def close(self):
        if getattr(self, "_h", None):
            lib().zh_shard_group_destroy(self._h)
            self._h = None

    __del__ = close


class ShardContext:
    """one sharded batch in flight (zh_shard_search_begin / finish / wait) on raw device pointers"""

    def __init__(self, group):
        self._group = group
        self._h = C.c_void_p()
        check(lib().zh_shard_ctx_create(group._h, C.byref(self._h)))

    def begin(self, d_q_ptr, b, top_k, metric):
        check(lib().zh_shard_search_begin(self._h, d_q_ptr, b, top_k, metric.metric, metric.mode))

    def finish(self, d_ids_ptr, d_keys_ptr, d_counts_ptr):
        check(lib().zh_shard_search_finish(self._h, d_ids_ptr, d_keys_ptr, d_counts_ptr))

    def begin_window(self, d_q_ptrs, b, top_k, metric):
        arr = (C.c_void_p * len(d_q_ptrs))(*d_q_ptrs)
        check(lib().zh_shard_search_begin_window(self._h, arr, len(d_q_ptrs), b, top_k, metric.metric, metric.mode))

    def finish_window(self, d_ids_ptrs, d_keys_ptrs, d_counts_ptrs):
        n = len(d_ids_ptrs)
        a, b_, c = (C.c_void_p * n)(*d_ids_ptrs), (C.c_void_p * n)(*d_keys_ptrs), (C.c_void_p * n)(*d_counts_ptrs)
        check(lib().zh_shard_search_finish_window(self._h, a, b_, c))

    def wait(self):
        check(lib().zh_shard_search_wait(self._h))

    def stream(self):
        """raw hipStream_t the merged results complete on"""
        return lib().zh_shard_ctx_stream(self._h)

    def local_result_ptr(self):
        return lib().zh_shard_ctx_local_result(self._h)

    def close(self):
        if getattr(self, "_h", None):
            lib().zh_shard_ctx_destroy(self._h)
            self._h = None

    __del__ = close


def merge_topk_device(device, n_shards, b, k, d_ids, d_keys, d_counts, d_out_ids, d_out_keys, d_out_counts, stream=None):
    check(lib().zh_merge_topk_device(device, n_shards, b, k, d_ids, d_keys, d_counts, d_out_ids, d_out_keys,
                                     d_out_counts, stream))


def packed_result_words(b, k):
    """u64 words of one rank's packed result: [ids b*k][keys b*k][counts b u32, padded]"""
    return int(lib().zh_packed_result_words(b, k))


def merge_topk_packed_device(device, n_shards, b, k, d_packed, d_out_ids, d_out_keys, d_out_counts, stream=None):
    check(lib().zh_merge_topk_packed_device(device, n_shards, b, k, d_packed, d_out_ids, d_out_keys, d_out_counts, stream))


def synth_queries_device(device, d_out, n_rows, b, dim, b0=0, seed_rows=0x5EB2A001, seed_q=0x5EB2A002, kind=0, stream=None):
    check(lib().zh_synth_queries_device(device, d_out, seed_rows, seed_q, n_rows, b0, b, dim, kind, stream))


# ------------------------------------------------------------------------ src/database/core.rs
def trim_device_memory():
    """zh_trim_device_memory: hand the library's cached device blocks (32 MiB and more, kept when an index or context lets go of them) back to the driver"""
    check(lib().zh_trim_device_memory())


class Database:
    """Database<N, Met, Mod> restricted to the two calls on the hot path: insert_records
    (core.rs:245-254) and query_vectors (core.rs:290-313).  Documents are kept in memory (the lz4
    files and the embedding model are out of scope)."""

    def __init__(self, dim, metric, index_options=None, **index_kwargs):
        # a metric CLASS is default-constructed, as Database::new / open do with `Met::default()` (core.rs:115,146)
        self.metric = metric() if isinstance(metric, type) else metric
        self.index = LSHIndex(dim, index_options, **index_kwargs)  # pub field `index`, core.rs:62
        self._documents = {}

    def insert_records(self, embeddings, documents):
        ids = self.index.add(embeddings)
        for i, doc in zip(ids.tolist(), documents):
            self._documents[i] = doc

    def remove(self, embedding_ids):
        """core.rs:205-214: index.remove, then the removed ids' documents go too"""
        for i in self.index.remove(embedding_ids).tolist():
            self._documents.pop(i, None)

    def deduplicate(self):
        """core.rs:216-225"""
        for i in self.index.deduplicate().tolist():
            self._documents.pop(i, None)

    def clear_database(self):
        """core.rs:194-198"""
        self.index.clear()
        self._documents.clear()

    def query_vectors(self, vectors, number_of_results):
        """-> {query index: {id: document}} ; order and distances are dropped as in core.rs:304-305."""
        if self.index.no_vectors():
            return {}
        ids, _, counts = self.index.search_batch(vectors, number_of_results, self.metric)
        return {b: {int(i): self._documents.get(int(i)) for i in ids[b, :counts[b]]} for b in range(ids.shape[0])}
