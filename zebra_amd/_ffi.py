"""ctypes binding of libzebra_hip.so (include/zebra_hip.h).  No fallback: if the library is missing
or no gfx950 device is usable, calls raise."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libzebra_hip.so")

ZH_OK = 0
ZH_EINVAL, ZH_ENOMEM, ZH_EHIP, ZH_ESTATE, ZH_ELIMIT, ZH_EUNSUPPORTED, ZH_EPEER = -1, -2, -3, -4, -5, -6, -7
COSINE, L2SQ, L2 = 0, 1, 2
CHEBYSHEV, CANBERRA, BRAY_CURTIS, MANHATTAN, L3, L4, HAMMING, MINKOWSKI, PNORM = 3, 4, 5, 6, 7, 8, 9, 10, 11
COSINE_PARITY, COSINE_CORRECTED = 0, 1
MAX_TOPK = 1024


class ZhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"zebra_hip error {code}: {msg}")
        self.code = code


class Options(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("max_node_size", C.c_uint32), ("num_trees", C.c_uint32),
                ("seed", C.c_uint64), ("device", C.c_int32), ("id_base", C.c_uint64),
                ("reserve_rows", C.c_uint64)]


class ForestView(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_planes", C.c_uint32), ("n_trees", C.c_uint32),
                ("n_leaf_ids", C.c_uint64), ("plane", C.c_void_p), ("left", C.c_void_p), ("right", C.c_void_p),
                ("roots", C.c_void_p), ("planes", C.c_void_p), ("consts", C.c_void_p), ("leaf_ids", C.c_void_p)]


class ForestSizes(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_planes", C.c_uint32), ("n_trees", C.c_uint32),
                ("n_leaf_ids", C.c_uint64)]


class RefHeader(C.Structure):
    _fields_ = [("uuid", C.c_uint8 * 16), ("max_node_size", C.c_uint64), ("num_trees", C.c_uint64), ("metric", C.c_int32),
                ("power", C.c_int32), ("model_off", C.c_uint64), ("model_len", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("batch", C.c_uint64), ("visits", C.c_uint64), ("rows_scored", C.c_uint64),
                ("rows_unique", C.c_uint64), ("rows_swept", C.c_uint64), ("candidates", C.c_uint64), ("planes_dense", C.c_uint64),
                ("planes_total", C.c_uint64), ("sweep_bytes", C.c_uint64),
                ("ms_hash", C.c_double), ("ms_walk", C.c_double), ("ms_sweep", C.c_double),
                ("ms_select", C.c_double), ("ms_final", C.c_double), ("ms_total", C.c_double),
                ("timed_batches", C.c_uint64), ("sweep_rows_accum", C.c_uint64), ("swept_rows_accum", C.c_uint64), ("sweep_launches_accum", C.c_uint64),
                ("window_batches", C.c_uint64), ("table_scan", C.c_uint64), ("scan_batches_accum", C.c_uint64),
                ("hash_from_scores", C.c_uint64), ("hash_exact_fixups", C.c_uint64),
                ("prefiltered", C.c_uint64), ("prefilter_exact_visits", C.c_uint64), ("prefilter_exact_rows", C.c_uint64),
                ("prefilter_fallbacks_accum", C.c_uint64), ("prefilter_last_overflow", C.c_uint64),
                ("approx_scan", C.c_uint64), ("approx_exact_visits", C.c_uint64), ("approx_survivors", C.c_uint64),
                ("approx_list_entries", C.c_uint64), ("approx_columns", C.c_uint64), ("approx_column_pairs", C.c_uint64), ("approx_batches_accum", C.c_uint64), ("approx_fallbacks_accum", C.c_uint64),
                ("approx_last_overflow", C.c_uint64), ("combined_batches_accum", C.c_uint64), ("combined_calls_accum", C.c_uint64),
                ("host_window_calls_accum", C.c_uint64), ("scan_order_keys", C.c_uint64), ("scan_order_share_permille", C.c_uint64),
                ("row_copy_bytes", C.c_uint64), ("approx_fused", C.c_uint64), ("approx_byte_rows", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class DebugPair(C.Structure):
    _fields_ = [("row", C.c_uint32), ("query", C.c_uint32), ("lo", C.c_uint32), ("hi", C.c_uint32), ("raw_s", C.c_float), ("raw_a2", C.c_float),
                ("flags", C.c_uint32), ("visit", C.c_uint32)]


class DebugScanInfo(C.Structure):
    _fields_ = [("approx_scan", C.c_uint32), ("queries", C.c_uint32), ("top_k", C.c_uint32), ("metric", C.c_int32), ("cosine_mode", C.c_int32),
                ("raw_kept", C.c_uint32), ("overflow", C.c_uint32), ("bound_const", C.c_float), ("row_rho", C.c_float), ("rho_norm", C.c_float),
                ("pairs", C.c_uint64), ("visits", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/zebra_hip.h declares: (name, restype, argtypes)
_vp, _u64, _u32, _sz, _i = C.c_void_p, C.c_uint64, C.c_uint32, C.c_size_t, C.c_int
SYMBOLS = [
    ("zh_options_default", None, [_vp]),
    ("zh_index_create", _i, [_vp, _vp]),
    ("zh_index_destroy", None, [_vp]),
    ("zh_index_clear", _i, [_vp]),
    ("zh_index_add", _i, [_vp, _vp, _sz, _vp]),
    ("zh_index_append", _i, [_vp, _vp, _sz, _vp]),
    ("zh_index_append_device", _i, [_vp, _vp, _sz]),
    ("zh_index_append_synthetic", _i, [_vp, _sz, _u64, _u64, _i]),
    ("zh_index_build", _i, [_vp]),
    ("zh_index_remove", _i, [_vp, _vp, _sz, _vp, _vp]),
    ("zh_index_deduplicate", _i, [_vp, _vp, _sz, _vp]),
    ("zh_index_set_forest", _i, [_vp, _vp]),
    ("zh_index_forest_sizes", _i, [_vp, _vp]),
    ("zh_index_get_forest", _i, [_vp] * 8),
    ("zh_index_count", _u64, [_vp]),
    ("zh_index_num_trees", _u32, [_vp]),
    ("zh_index_dim", _u32, [_vp]),
    ("zh_index_device", C.c_int32, [_vp]),
    ("zh_index_id_base", _u64, [_vp]),
    ("zh_index_rows_device", _vp, [_vp]),
    ("zh_index_sweep_stream", _vp, [_vp]),
    ("zh_index_read_rows", _i, [_vp, _u64, _sz, _vp]),
    ("zh_hash_signs", _i, [_vp, _vp, _sz, _vp, _vp]),
    ("zh_search_batch", _i, [_vp, _vp, _sz, _sz, _i, _i, _vp, _vp, _vp]),
    ("zh_search_batch_device", _i, [_vp, _vp, _sz, _sz, _i, _i, _vp, _vp, _vp, _vp]),
    ("zh_search_ctx_create", _i, [_vp, _vp]),
    ("zh_search_ctx_destroy", None, [_vp]),
    ("zh_search_begin", _i, [_vp, _vp, _sz, _sz, _i, _i, _vp]),
    ("zh_search_finish", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("zh_search_wait", _i, [_vp]),
    ("zh_search_begin_window", _i, [_vp, _vp, _sz, _sz, _sz, _i, _i, _vp]),
    ("zh_search_finish_window", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("zh_distance_batch", _i, [_i, _i, _vp, _vp, _sz, _sz, _vp, _i]),
    ("zh_distance_pair", _i, [_i, _i, _vp, _vp, _sz, _vp, _i]),
    ("zh_merge_topk_device", _i, [_i, _u32, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("zh_packed_result_words", _sz, [_sz, _sz]),
    ("zh_merge_topk_packed_device", _i, [_i, _u32, _sz, _sz, _vp, _vp, _vp, _vp, _vp]),
    ("zh_shard_unique_id", _i, [_vp]),
    ("zh_shard_group_create", _i, [_vp, _vp, _u32, _u32, _vp]),
    ("zh_shard_group_destroy", None, [_vp]),
    ("zh_shard_group_ranks", _u32, [_vp]),
    ("zh_shard_group_rank", _u32, [_vp]),
    ("zh_shard_search_batch_device", _i, [_vp, _vp, _sz, _sz, _i, _i, _vp, _vp, _vp]),
    ("zh_shard_search_batch", _i, [_vp, _vp, _sz, _sz, _i, _i, _vp, _vp, _vp]),
    ("zh_shard_ctx_create", _i, [_vp, _vp]),
    ("zh_shard_ctx_destroy", None, [_vp]),
    ("zh_shard_search_begin", _i, [_vp, _vp, _sz, _sz, _i, _i]),
    ("zh_shard_search_finish", _i, [_vp, _vp, _vp, _vp]),
    ("zh_shard_search_wait", _i, [_vp]),
    ("zh_shard_search_begin_window", _i, [_vp, _vp, _sz, _sz, _sz, _i, _i]),
    ("zh_shard_search_finish_window", _i, [_vp, _vp, _vp, _vp]),
    ("zh_shard_ctx_stream", _vp, [_vp]),
    ("zh_shard_ctx_local_result", _vp, [_vp]),
    ("zh_shard_exchange_words", _sz, [_sz, _sz]),
    ("zh_shard_status_word", _u64, [_i, _u32]),
    ("zh_shard_verdict", _i, [_vp, _u32, _u32, _vp, _vp, _vp]),
    ("zh_synth_queries_device", _i, [_i, _vp, _u64, _u64, _u64, _u64, _sz, _u32, _i, _vp]),
    ("zh_ref_forest_decode", _i, [_u32, _sz, _vp, _vp, _sz, _vp, _vp, _vp]),
    ("zh_ref_forest_view", _i, [_vp, _vp]),
    ("zh_ref_forest_free", None, [_vp]),
    ("zh_ref_tree_encode", _i, [_vp, _u32, _u32, _vp, _u64, _vp, _sz, _vp]),
    ("zh_ref_header_decode", _i, [_vp, _sz, _i, _sz, _vp]),
    ("zh_ref_header_encode", _i, [_vp, _vp, _vp, _sz, _vp]),
    ("zh_debug_keep_raw", _i, [_vp, _i]),
    ("zh_debug_scan_pairs", _i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    ("zh_set_profiling", _i, [_vp, _i]),
    ("zh_stats", _i, [_vp, _vp]),
    ("zh_stats_reset", _i, [_vp]),
    ("zh_set_dense_levels", _i, [_vp, _i]),
    ("zh_set_sweep_mode", _i, [_vp, _i]),
    ("zh_set_hash_mode", _i, [_vp, _i]),
    ("zh_last_error", C.c_char_p, []),
    ("zh_version", C.c_char_p, []),
    ("zh_trim_device_memory", _i, []),
]

_lib = None


def lib():
    """Load libzebra_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                              f"g.build()'` or `make -C zebra_amd/csrc` (there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != ZH_OK:
        raise ZhError(rc, lib().zh_last_error().decode("utf-8", "replace"))
