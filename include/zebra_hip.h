/*
 * zebra_hip.h -- C ABI of the MI355X (gfx950) implementation of Zebra's LSH bucket-scan + distance
 * hot path.  This is the drop-in boundary: what the reference crate's FFI for this path binds
 * (INTEGRATION.md shows the Rust `extern "C"` block and the shim that keeps `space::Metric`,
 * `LSHIndex<N>` and `Database<N,Met,Mod>` signatures on top of it).
 *
 * Reference interfaces replaced (all paths relative to /root/reference):
 *   zh_index_create / zh_index_destroy   LSHIndex::new            src/database/index/lsh.rs:162-167
 *   zh_index_add                         LSHIndex::add            src/database/index/lsh.rs:440-466
 *                                        (first call = build_index, lsh.rs:411-429)
 *   zh_index_build                       build_a_tree x num_trees src/database/index/lsh.rs:250-267
 *   zh_search_batch[_device]             LSHIndex::search         src/database/index/lsh.rs:544-565
 *                                        driven per batch as Database::query_vectors does,
 *                                        src/database/core.rs:290-313
 *   zh_hash_signs                        Hyperplane::point_is_above  src/database/index/lsh.rs:39-43
 *   zh_distance_batch / zh_distance_pair Metric::distance for CosineDistance, L2SquaredDistance,
 *                                        L2Distance               src/distance.rs:19-49, 103-114
 *   zh_index_count / zh_index_num_trees  LSHIndex::no_vectors / no_trees / is_empty  lsh.rs:389-409
 *   zh_index_clear                       LSHIndex::clear          src/database/index/lsh.rs:506-529
 *   zh_index_remove / zh_index_deduplicate  LSHIndex::remove / deduplicate  src/database/index/lsh.rs:473-503, 270-288
 *   zh_shard_group_* / zh_shard_search_* (new) the loop of Database::query_vectors (src/database/core.rs:299-303) over an index
 *                                        whose rows are sharded across GPUs (README.md:31 "can be sharded"): local search on
 *                                        this rank's shard, ONE RCCL all-gather of the packed top-k, merge on every rank
 *   zh_merge_topk_device                 (new) the shard merge alone
 *
 * Conventions
 *   - Every function returns ZH_OK (0) or a negative zh_status; zh_last_error() gives the message
 *     of the calling thread's last failure.  Nothing throws or aborts across this boundary.
 *   - The caller owns every buffer it passes; the library owns all device memory behind zh_index.
 *   - ids are dense row numbers in insertion order (the Rust shim keeps row -> Uuid, lsh.rs:415);
 *     results carry id_base + row so that shards of one logical index return global ids.
 *   - keys are the reference's DistanceUnit (distance.rs:13): the IEEE-754 bit pattern of the f64
 *     distance, compared as an unsigned integer; ties order by id.
 *   - search/hash/distance calls on one index may come from several host threads (the reference
 *     calls search from rayon workers, core.rs:299-303); they serialise on an internal lock.
 *     add/build/set_forest/clear/destroy need external exclusion, like any &mut in the crate.
 *   - There is NO CPU fallback: every entry point that computes runs gfx950 kernels and fails with
 *     ZH_EHIP when no device is usable.
 */
#ifndef ZEBRA_HIP_H
#define ZEBRA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define ZH_API __attribute__((visibility("default")))
#else
#define ZH_API
#endif

typedef struct zh_index zh_index;

typedef enum zh_status {
    ZH_OK = 0,
    ZH_EINVAL = -1,       /* bad argument */
    ZH_ENOMEM = -2,       /* host or device allocation failed */
    ZH_EHIP = -3,         /* HIP runtime / no usable gfx950 device */
    ZH_ESTATE = -4,       /* call not valid in the index's current state */
    ZH_ELIMIT = -5,       /* a documented limit was exceeded (e.g. top_k > ZH_MAX_TOPK) */
    ZH_EUNSUPPORTED = -6,
    ZH_EPEER = -7         /* sharded search: ANOTHER rank of the group failed its part of the batch (or died); the exchange
                           * completed (or was aborted after a timeout), the batch's results are not valid on any rank */
} zh_status;

/* the 13 metric structs of src/distance.rs.  0-2 are the simsimd path (key = f64 bits); 3-11 the `distances`
 * crate path (key = f32 bits widened, distance.rs:59; Hamming an integer count, distance.rs:144-158) */
typedef enum zh_metric {
    ZH_COSINE = 0,      /* CosineDistance      distance.rs:15-32  */
    ZH_L2SQ = 1,        /* L2SquaredDistance   distance.rs:34-49  */
    ZH_L2 = 2,          /* L2Distance          distance.rs:99-114 */
    ZH_CHEBYSHEV = 3,   /* ChebyshevDistance   distance.rs:51-61  */
    ZH_CANBERRA = 4,    /* CanberraDistance    distance.rs:63-73  */
    ZH_BRAY_CURTIS = 5, /* BrayCurtisDistance  distance.rs:75-85  */
    ZH_MANHATTAN = 6,   /* ManhattanDistance   distance.rs:87-97  */
    ZH_L3 = 7,          /* L3Distance          distance.rs:116-126 */
    ZH_L4 = 8,          /* L4Distance          distance.rs:128-138 */
    ZH_HAMMING = 9,     /* HammingDistance     distance.rs:140-158 (low byte of each f32's bits) */
    ZH_MINKOWSKI = 10,  /* MinkowskiDistance { power }  distance.rs:160-174 */
    ZH_PNORM = 11       /* PNormDistance { power }      distance.rs:176-190 */
} zh_metric;
/* `power` is the struct's i32 field and every value is served, as in the reference: the DEFAULT-constructed metric
 * (`Met::default()`, core.rs:115,146; #[derive(Default)], distance.rs:160-165,176-181) has power 0 -- p-norm(0) = d,
 * Minkowski(0) = d^(1/0) = +inf (1 at d = 1) --, negative powers go through powi's reciprocal and pow's IEEE cases. */

/* distance.rs:23-25 applies `1.0 - c` to simsimd's cosine, which is already a distance, so the
 * reference key is the bit pattern of the cosine SIMILARITY.  PARITY reproduces that literally;
 * CORRECTED keys on the distance.  The `cosine_mode` argument of the calls below is this mode for ZH_COSINE,
 * the `power` field for ZH_MINKOWSKI / ZH_PNORM, and ignored by the other metrics. */
typedef enum zh_cosine_mode { ZH_COSINE_PARITY = 0, ZH_COSINE_CORRECTED = 1 } zh_cosine_mode;

#define ZH_MAX_TOPK 1024u
#define ZH_MAX_DEPTH 60u /* the reference recurses without bound on an unsplittable node */
#define ZH_MAX_DIM (1u << 20) /* zh_options.dim beyond this is refused with ZH_ELIMIT (row offsets stay far below 2^63) */

typedef struct zh_options {
    uint32_t dim;           /* N of Embedding<N>, lib.rs:18 */
    uint32_t max_node_size; /* LSHIndexOptions::max_node_size, default 5  (lsh.rs:126,134) */
    uint32_t num_trees;     /* LSHIndexOptions::num_trees,     default 15 (lsh.rs:128,135) */
    uint64_t seed;          /* hyperplane sampling seed (the reference uses an unseeded RNG, lsh.rs:201) */
    int32_t device;         /* HIP device ordinal; -1 = the calling thread's current device */
    uint64_t id_base;       /* global id of local row 0 (shard offset) */
    uint64_t reserve_rows;  /* capacity hint: avoids a reallocating copy on growth */
} zh_options;

/* Flat forest (host pointers).  Node i is inner when plane[i] >= 0: left[i] is the child holding
 * the rows BELOW the plane, right[i] the rows ABOVE (lsh.rs:260-264).  It is a leaf when
 * plane[i] == -1: (uint32_t)left[i] is an offset into leaf_ids and right[i] the leaf's length.
 * planes is n_planes x dim row-major, consts the matching offsets (Hyperplane, lsh.rs:16-25). */
typedef struct zh_forest_view {
    uint32_t n_nodes, n_planes, n_trees;
    uint64_t n_leaf_ids;
    const int32_t *plane, *left, *right;
    const uint32_t *roots;
    const float *planes, *consts;
    const uint32_t *leaf_ids;
} zh_forest_view;

typedef struct zh_forest_sizes {
    uint32_t n_nodes, n_planes, n_trees;
    uint64_t n_leaf_ids;
} zh_forest_sizes;

/* Counters and timings of the most recent search batch on this index (zh_stats). */
typedef struct zh_stats_t {
    uint64_t batch;            /* queries in the batch */
    uint64_t visits;           /* leaf visits (bucket probes) */
    uint64_t rows_scored;      /* R_total: stored rows whose distance was computed */
    uint64_t rows_unique;      /* R_unique: rows of the distinct leaves touched (0 unless stats level >= 2) */
    uint64_t rows_swept;       /* rows the sweep kernel actually loaded: queries that probe the same leaf share them */
    uint64_t candidates;       /* ids handed to the final top-k (before de-duplication) */
    uint64_t planes_dense;     /* hyperplanes hashed by the dense MFMA kernel, per query */
    uint64_t planes_total;     /* hyperplanes in the forest */
    uint64_t sweep_bytes;      /* bytes the sweep moves: (4*dim + 4) * rows_swept + 8 * rows_scored */
    /* accumulated since zh_stats_reset, in milliseconds, measured with hipEvents on the stream the
     * kernels run on (only when profiling is enabled with zh_set_profiling) */
    double ms_hash, ms_walk, ms_sweep, ms_select, ms_final, ms_total;
    uint64_t timed_batches;
    uint64_t sweep_rows_accum;  /* rows_scored summed over the timed batches */
    uint64_t swept_rows_accum;  /* rows_swept summed over the timed batches */
    uint64_t sweep_launches_accum; /* sweep_kernel launches over the timed batches (a batch is several launches) */
    uint64_t window_batches;    /* API batches handled together in the most recent internal batch (zh_search_begin_window) */
    uint64_t table_scan;        /* 1: the most recent batch was swept by the table scan (every stored row streamed once, scored
                                 * against every query that visits one of its leaves; rows_swept = stored rows), 0: leaf by leaf */
    uint64_t scan_batches_accum; /* timed internal batches swept by the table scan */
    uint64_t hash_from_scores;  /* 1: the most recent batch took every sign of the forest from row scores (zh_set_hash_mode) */
    uint64_t hash_exact_fixups; /* ... and this many of its signs lay inside the rounding bound and were recomputed exactly */
    uint64_t prefiltered;       /* 1: the most recent batch picked its candidates from the row scores (zh_set_sweep_mode): rows_scored
                                 * rows were judged on their scores, rows_swept of them scored with the reference's arithmetic */
    uint64_t prefilter_exact_visits; /* ... leaf visits whose `take` nearest rows the scores could not decide (scored exactly) */
    uint64_t prefilter_exact_rows;   /* ... rows scored exactly in total (= rows_swept) */
    uint64_t prefilter_fallbacks_accum; /* batches redone with the sweep because a candidate list ran over (since zh_stats_reset) */
    uint64_t prefilter_last_overflow;   /* ... what ran over in the most recent of them: 1 | 2 a (query, tree) list, 4 the table of
                                         * visits to score exactly, 8 a leaf longer than 64 rows, 16 a query's lists together hold more than the final sort */
    uint64_t approx_scan;           /* 1: the most recent batch's table scan read HALF-WIDTH (fp16) copies of the queries: every (row, query)
                                     * pair got an interval that contains the reference's key, the intervals picked the candidates, and only
                                     * the rows they could not rule out were scored with the reference's arithmetic (zh_set_sweep_mode);
                                     * 2: the same with the products on the matrix cores, from the index's fp16 copy of the stored rows;
                                     * 3: 128-d rows leaf by leaf from their fp16 copy (table_scan = 0) */
    uint64_t approx_exact_visits;   /* ... leaf visits that take fewer than top_k rows: scored and ranked exactly */
    uint64_t approx_survivors;      /* ... rows (over all queries) that got the reference's key for the final top_k */
    uint64_t approx_list_entries;   /* ... candidates handed to the per-query stage (before de-duplication) */
    uint64_t approx_columns;        /* ... approx_scan 2: a tile column of the matrix-core scan is a DISTINCT query of a wave's pairs; columns and pairs of */
    uint64_t approx_column_pairs;   /*     (a 1-in-64 sample of) the waves of the most recent such batch: what the pairs share (1.0 = nothing) */
    uint64_t approx_batches_accum;  /* timed internal batches scanned this way */
    uint64_t approx_fallbacks_accum; /* batches redone by the f32 scan ON THE DEVICE, in stream order, because a list ran over */
    uint64_t approx_last_overflow;  /* ... what ran over: 1 a query's candidate list, 2 a query's survivors, 4 the table of exact
                                     * visits, 8 their key scratch */
    uint64_t combined_batches_accum; /* zh_search_batch: internal batches that served MORE than one concurrent caller (since reset) */
    uint64_t combined_calls_accum;   /* ... and the calls they served */
    uint64_t host_window_calls_accum; /* zh_search_batch calls whose (large, host-resident) batch ran as windows over two contexts, copies beside
                                      * the kernels (since reset) */
    uint64_t scan_order_keys;        /* the matrix-core scan keeps ITS view of the rows (fp16 tiles, row -> leaf entries) in the order that lets a tile's 16 rows
                                      * share the most leaves: 0 id order, 2 / 3 sorted by the leaves in that many trees -- measured when the copy is made */
    uint64_t scan_order_share_permille; /* ... (adjacent rows, tree) combinations in the same leaf under the kept order, per thousand */
    uint64_t row_copy_bytes;         /* device memory the index holds for fp16 copies of its stored rows (the half-width sweeps: zh_set_sweep_mode);
                                     * 0 until a batch has used one, and with modes 1 / 2 / 5 */
    uint64_t approx_fused;           /* 1: approx_scan 3 and the most recent batch's sweep was FUSED -- intervals, bounds and the queries' candidate
                                     * lists inside the sweep kernel (no raw pairs written, no select pass; round 6) */
    uint64_t approx_byte_rows;       /* 1: approx_scan 3 and the sweep read an EXACT copy of 128 BYTES per stored row: every element of the table is an
                                     * integer in 0 .. 255 (SIFT descriptors), checked when the copy is made; any other row and the copy is of halves */
} zh_stats_t;

/* ---- lifecycle ------------------------------------------------------------------------------ */
ZH_API void zh_options_default(zh_options *opt); /* dim 0, max_node_size 5, num_trees 15 (lsh.rs:131-138) */
ZH_API int zh_index_create(const zh_options *opt, zh_index **out);
ZH_API void zh_index_destroy(zh_index *idx);
ZH_API int zh_index_clear(zh_index *idx); /* drops vectors AND trees (what lsh.rs:506-529 intends) */

/* ---- insert --------------------------------------------------------------------------------- */
/* LSHIndex::add: rows is n x dim row-major host memory.  If the index has no trees the forest is
 * built over everything stored so far plus these rows (build_index); otherwise the rows descend
 * the existing trees and split full leaves (insert, lsh.rs:350-382) -- as the sequential execution
 * "one row after another, each into every tree" of the reference's racy par_iter (lsh.rs:445-462).
 * out_row_ids (may be NULL) receives id_base + row for each new row. */
ZH_API int zh_index_add(zh_index *idx, const float *rows, size_t n, uint64_t *out_row_ids);
/* staged loading for large shards: append without building, then zh_index_build once */
ZH_API int zh_index_append(zh_index *idx, const float *rows, size_t n, uint64_t *out_row_ids);
ZH_API int zh_index_append_device(zh_index *idx, const float *d_rows, size_t n);
/* append n synthetic rows generated on the device (bit-identical to oracle zo_synth_rows);
 * kind 0 = ~N(0,1), kind 1 = integer-valued "SIFT-style" in [0,255], kind 2 = clustered (128 consecutive rows
 * share a centre: centre + 0.25 * noise) */
ZH_API int zh_index_append_synthetic(zh_index *idx, size_t n, uint64_t seed, uint64_t first_row, int kind);
ZH_API int zh_index_build(zh_index *idx); /* (re)build all trees on the GPU */
/* LSHIndex::remove (lsh.rs:473-503) as intended: the ids leave every tree (the reference only edits trees whose root
 * is a leaf) and zh_index_count drops; later splits no longer sample them (lsh.rs:495, 197-201).  out_found (may be
 * NULL): 1 per id that was present.  LSHIndex::deduplicate (lsh.rs:270-288): rows bit-identical to an earlier row are
 * removed; out_ids (may be NULL) receives up to cap removed ids, ascending. */
ZH_API int zh_index_remove(zh_index *idx, const uint64_t *ids, size_t n, uint8_t *out_found, size_t *out_n_removed);
ZH_API int zh_index_deduplicate(zh_index *idx, uint64_t *out_ids, size_t cap, size_t *out_n_removed);

/* ---- forest exchange (parity tests inject / extract the exact same forest) ----------------- */
ZH_API int zh_index_set_forest(zh_index *idx, const zh_forest_view *forest);
ZH_API int zh_index_forest_sizes(zh_index *idx, zh_forest_sizes *out);
/* caller-allocated arrays of the sizes reported above */
ZH_API int zh_index_get_forest(zh_index *idx, int32_t *plane, int32_t *left, int32_t *right, uint32_t *roots,
                        float *planes, float *consts, uint32_t *leaf_ids);

/* ---- queries -------------------------------------------------------------------------------- */
ZH_API uint64_t zh_index_count(const zh_index *idx);     /* stored vectors (0 <=> no_vectors) */
ZH_API uint32_t zh_index_num_trees(const zh_index *idx); /* built trees    (0 <=> no_trees)   */
ZH_API uint32_t zh_index_dim(const zh_index *idx);
ZH_API int32_t zh_index_device(const zh_index *idx);     /* HIP device ordinal the index lives on */
ZH_API uint64_t zh_index_id_base(const zh_index *idx);
ZH_API const float *zh_index_rows_device(const zh_index *idx); /* device pointer to the stored rows (read-only) */
/* copy n stored rows starting at local row `first` back to host memory (KeyValue::embedding, lsh.rs:107-119) */
ZH_API int zh_index_read_rows(zh_index *idx, uint64_t first, size_t n, float *out);

/* point_is_above for every plane of the forest (numbered as zh_index_get_forest returns them) and
 * every query: out_bits is b x ceil(n_planes/32) words, bit p%32 of word p/32 = above.
 * out_dots (may be NULL) is b x n_planes raw f32 dot products.  q is b x dim host memory.  With zh_set_hash_mode(2) and
 * out_dots == NULL the signs are taken through the row-score path where the forest allows (b % 4 == 0): same bits. */
ZH_API int zh_hash_signs(zh_index *idx, const float *q, size_t b, uint32_t *out_bits, float *out_dots);

/* LSHIndex::search for a batch: q is b x dim; out_ids/out_keys are b x k (entries past
 * out_counts[i] are set to UINT64_MAX); ascending by (key, id).  Host pointers.
 * Callable from many threads at once -- the crate calls search with ONE query from every rayon worker (core.rs:299-303) -- and
 * built for that: callers that arrive while a batch is on the GPU queue up, the thread that finds the engine idle leads a round
 * and runs every queued request with the same top_k / metric / mode as ONE internal batch (answers are those of separate calls,
 * bit for bit: the queries of a batch never interact); the others sleep until theirs is done.  No timer: a lone caller is served
 * at once, a crowd is batched by the time the batch before it takes.  zh_stats_t::combined_* count it. */
ZH_API int zh_search_batch(zh_index *idx, const float *q, size_t b, size_t k, int metric, int cosine_mode,
                    uint64_t *out_ids, uint64_t *out_keys, uint32_t *out_counts);
/* The same with queries and results already resident in device memory; kernels are enqueued on
 * `stream` (a hipStream_t; NULL = the index's own stream) and have completed on return. */
ZH_API int zh_search_batch_device(zh_index *idx, const float *d_q, size_t b, size_t k, int metric, int cosine_mode,
                           uint64_t *d_out_ids, uint64_t *d_out_keys, uint32_t *d_out_counts, void *stream);

/* Pipelined form of zh_search_batch_device (new; the reference has one blocking search per query): a context
 * is one in-flight batch with its own scratch.  The context calls do NOT take the index's internal lock (the blocking
 * calls do): contexts of one index may be driven from several threads, one thread per context at a time, concurrently with
 * each other and with blocking searches -- they share only read-only index state and the statistics of zh_stats (guarded
 * separately) -- but never concurrently with add / build / set_forest / remove / clear / destroy on that index.
 * begin enqueues the hash and the walk's counting pass on
 * `stream` and returns; finish waits (host side) only for three totals, then enqueues the distance sweep,
 * the selection and the final top-k and returns; wait blocks until the results are complete.  With two
 * contexts on two streams one host thread keeps the sweep of batch i and the small latency-bound kernels of
 * batches i and i+1 on the GPU together.  Queries and outputs must stay valid until wait, and the outputs are complete only
 * when wait has returned -- a stream sync is not enough: a prefiltered batch (zh_set_sweep_mode) whose candidate lists ran over is
 * redone with the sweep inside wait. */
typedef struct zh_search_ctx zh_search_ctx;
ZH_API int zh_search_ctx_create(zh_index *idx, zh_search_ctx **out);
ZH_API void zh_search_ctx_destroy(zh_search_ctx *ctx);
ZH_API int zh_search_begin(zh_search_ctx *ctx, const float *d_q, size_t b, size_t k, int metric, int cosine_mode,
                    void *stream);
/* A lowest-priority, non-blocking stream owned by the index, meant to be passed as `sweep_stream` below: light kernels
 * (high-priority streams) and collectives (normal priority) then never share a hardware queue with the sweeps. */
ZH_API void *zh_index_sweep_stream(const zh_index *idx);
/* sweep_stream (may be NULL = the begin stream): the stream the HBM-bound distance sweep is enqueued on; sharing
 * one sweep stream between contexts runs the sweeps of successive batches back to back while the other kernels of
 * each batch overlap them on the contexts' own streams (the library inserts the event dependencies). */
ZH_API int zh_search_finish(zh_search_ctx *ctx, uint64_t *d_out_ids, uint64_t *d_out_keys, uint32_t *d_out_counts,
                     void *sweep_stream);
ZH_API int zh_search_wait(zh_search_ctx *ctx);
/* A WINDOW: n_batches batches (same b, k and metric; queries at n_batches device pointers) handled as ONE internal batch.
 * The walk, the leaf groups and the sweep span the whole window, so a stored row crosses HBM once per group of queries of
 * the WINDOW that score it -- with b << leaves per tree, two batches share rows a single batch cannot -- and the light
 * kernels are launched once per window.  Results are delivered per batch (n_batches output pointers each), bit-identical
 * to n_batches separate calls; the price is latency: no batch of the window completes before the whole window has.
 * zh_search_wait as for a single batch.  The counters of zh_stats then describe the window (batch = n_batches * b). */
#define ZH_MAX_WINDOW 64u
ZH_API int zh_search_begin_window(zh_search_ctx *ctx, const float *const *d_q, size_t n_batches, size_t b, size_t k, int metric,
                                  int cosine_mode, void *stream);
ZH_API int zh_search_finish_window(zh_search_ctx *ctx, uint64_t *const *d_out_ids, uint64_t *const *d_out_keys,
                                   uint32_t *const *d_out_counts, void *sweep_stream);

/* Metric::distance(stored=a[i], query=q) for n stored rows against one query (host pointers).  Device buffers are kept
 * per calling thread between calls (a pair costs two small copies in, three small kernels, one copy out).
 * COST of the single-pair forms (n = 1, zh_distance_pair): tens of microseconds -- a host -> device copy, three launches and a
 * blocking copy back -- where the crate's Metric::distance spends ~100 ns in simsimd (distance.rs:21-31).  There is deliberately
 * no host-side arithmetic behind this ABI (every key this library returns comes from the gfx950 kernels: one implementation
 * to hold bit-exact, and nothing that could pass for a CPU fallback): a caller that needs isolated pairs at CPU speed keeps the
 * crate's own metric for them (the shim routes Metric::distance to the crate, batches to this library). */
ZH_API int zh_distance_batch(int metric, int cosine_mode, const float *a, const float *q, size_t n, size_t dim,
                      uint64_t *out_keys, int device);
ZH_API int zh_distance_pair(int metric, int cosine_mode, const float *a, const float *b, size_t dim, uint64_t *out_key,
                     int device);

/* Shard merge: S lists of b x k (ids, keys) with counts S x b, all in device memory (e.g. the
 * output of an RCCL all-gather of every rank's zh_search_batch_device result) -> b x k merged.
 * The kernel is enqueued on `stream` and the call returns; synchronise the stream before reading. */
ZH_API int zh_merge_topk_device(int device, uint32_t n_shards, size_t b, size_t k, const uint64_t *d_ids,
                         const uint64_t *d_keys, const uint32_t *d_counts, uint64_t *d_out_ids,
                         uint64_t *d_out_keys, uint32_t *d_out_counts, void *stream);

/* The same merge over ONE buffer per shard -- [ids b*k u64][keys b*k u64][counts b u32, padded to 8 bytes],
 * zh_packed_result_words(b, k) u64 words -- so that a batch needs a single all-gather: point
 * zh_search_batch_device / zh_search_finish at the three sections of such a buffer. */
ZH_API size_t zh_packed_result_words(size_t b, size_t k);
ZH_API int zh_merge_topk_packed_device(int device, uint32_t n_shards, size_t b, size_t k, const uint64_t *d_packed,
                                uint64_t *d_out_ids, uint64_t *d_out_keys, uint32_t *d_out_counts, void *stream);

/* ---- sharded search: rows partitioned over GPUs, one RCCL all-gather per batch ---------------------------------
 * One process per GPU (the harness' model) -- or several groups in one process, one per device.  Every rank owns an
 * ordinary zh_index over its rows (its own forest, options.id_base = global id of its first row) and joins a group;
 * a search on the group is then ONE call per batch on every rank, with the same queries everywhere:
 *     local zh_search on this rank's shard -> ncclAllGather of the packed [ids | keys | counts] result (in place,
 *     b*k*16 + b*4 bytes per rank) -> merge_wave_kernel on every rank -> every rank holds the global top-k.
 * top-k(union of the shards' top-k) == top-k(union of the shards' candidates), so the result is bit-identical to the
 * reference searching S independent LSHIndex instances and merging by (key, id).
 * The library links librccl itself; the caller only moves the 128-byte unique id from rank 0 to the other ranks
 * (any host channel: a file, MPI, a torch.distributed store).
 * Calls on one group must come from one thread at a time and in the same order on every rank (they are collectives).
 *
 * Errors are a GROUP outcome (the reference swallows a failed query, core.rs:303; a collective cannot): a rank whose local
 * search fails still joins the all-gather, with an empty slot and its code in a status word that travels with the packed
 * result (zh_shard_exchange_words = zh_packed_result_words + 1).  zh_shard_search_wait -- and the blocking calls -- then
 * return the rank's own code, or ZH_EPEER where only other ranks failed: the same verdict everywhere, nobody hangs, the group
 * stays usable.  The blocking calls split a batch that passes a per-launch limit IDENTICALLY on every rank (a ZH_ELIMIT
 * anywhere makes every rank halve the chunk and repeat it; the first size comes from the largest visits-per-query any rank
 * reported).  Pipelined calls: after a begin that returned an error, STILL call finish (it joins the exchange and returns the
 * error) and wait.  A rank that dies or cannot join is bounded by ZH_SHARD_TIMEOUT_MS (environment, default 300000): wait
 * aborts the communicator, returns ZH_EPEER, and the group is dead (later calls fail fast; destroy it). */
typedef struct zh_shard_group zh_shard_group;
#define ZH_UNIQUE_ID_BYTES 128
ZH_API int zh_shard_unique_id(uint8_t out_id[ZH_UNIQUE_ID_BYTES]); /* rank 0: ncclGetUniqueId */
/* collective over the n_ranks callers (ncclCommInitRank on the shard's device).  The group borrows `shard`, which must
 * outlive it; rows may still be added to the shard between searches. */
ZH_API int zh_shard_group_create(zh_index *shard, const uint8_t id[ZH_UNIQUE_ID_BYTES], uint32_t n_ranks, uint32_t rank,
                                 zh_shard_group **out);
ZH_API void zh_shard_group_destroy(zh_shard_group *grp);
ZH_API uint32_t zh_shard_group_ranks(const zh_shard_group *grp); /* ncclCommCount: the ranks RCCL actually connected */
ZH_API uint32_t zh_shard_group_rank(const zh_shard_group *grp);
/* Blocking search over the whole sharded index; queries / results in device memory on every rank (d_out_* receive the
 * MERGED global top-k, same layout as zh_search_batch_device). */
ZH_API int zh_shard_search_batch_device(zh_shard_group *grp, const float *d_q, size_t b, size_t k, int metric, int cosine_mode,
                                        uint64_t *d_out_ids, uint64_t *d_out_keys, uint32_t *d_out_counts);
/* The same from / to host memory (what a Rust Database::query_vectors over a sharded index calls). */
ZH_API int zh_shard_search_batch(zh_shard_group *grp, const float *q, size_t b, size_t k, int metric, int cosine_mode,
                                 uint64_t *out_ids, uint64_t *out_keys, uint32_t *out_counts);
/* Pipelined form, as zh_search_begin / finish / wait: a context is one batch in flight with its own streams (light
 * kernels: high priority; exchange + merge: normal priority, beside the next batch's sweep on the index's sweep stream).
 * finish enqueues sweep, select, final, the all-gather and the merge and returns; wait blocks until the merged results
 * are complete.  zh_shard_ctx_stream is the stream they complete on (enqueue result copies behind it). */
typedef struct zh_shard_ctx zh_shard_ctx;
ZH_API int zh_shard_ctx_create(zh_shard_group *grp, zh_shard_ctx **out);
ZH_API void zh_shard_ctx_destroy(zh_shard_ctx *ctx);
ZH_API int zh_shard_search_begin(zh_shard_ctx *ctx, const float *d_q, size_t b, size_t k, int metric, int cosine_mode);
ZH_API int zh_shard_search_finish(zh_shard_ctx *ctx, uint64_t *d_out_ids, uint64_t *d_out_keys, uint32_t *d_out_counts);
ZH_API int zh_shard_search_wait(zh_shard_ctx *ctx);
/* windows, as zh_search_begin_window / zh_search_finish_window: one all-gather and one merge per window */
ZH_API int zh_shard_search_begin_window(zh_shard_ctx *ctx, const float *const *d_q, size_t n_batches, size_t b, size_t k,
                                        int metric, int cosine_mode);
ZH_API int zh_shard_search_finish_window(zh_shard_ctx *ctx, uint64_t *const *d_out_ids, uint64_t *const *d_out_keys,
                                         uint32_t *const *d_out_counts);
ZH_API void *zh_shard_ctx_stream(const zh_shard_ctx *ctx);
/* this rank's own (unmerged) packed result of the context's last finished batch: zh_packed_result_words(b, k) words */
ZH_API const uint64_t *zh_shard_ctx_local_result(const zh_shard_ctx *ctx);
/* The status protocol, as pure host arithmetic (no GPU needed; tests carry it over gloo).  Words per rank in the exchange;
 * a rank's status word = its local status (low 32 bits, sign-extended zh_status) | leaf visits per query it has seen << 32;
 * the verdict every rank derives from the n_ranks gathered words: ZH_OK, status_words[rank]'s own code, or ZH_EPEER.
 * out_* may be NULL: first failed rank (n_ranks if none), largest visits-per-query reported, 1 if every failure is ZH_ELIMIT. */
ZH_API size_t zh_shard_exchange_words(size_t b, size_t k);
ZH_API uint64_t zh_shard_status_word(int status, uint32_t visits_per_query);
ZH_API int zh_shard_verdict(const uint64_t *status_words, uint32_t n_ranks, uint32_t rank, uint32_t *out_first_failed_rank,
                            uint32_t *out_max_visits_per_query, int *out_all_elimit);

/* synthetic queries on the device (bit-identical to oracle zo_synth_queries) */
ZH_API int zh_synth_queries_device(int device, float *d_out, uint64_t seed_rows, uint64_t seed_q, uint64_t n_rows,
                            uint64_t b0, size_t b, uint32_t dim, int kind, void *stream);

/* ---- the reference's on-disk VALUES ------------------------------------------------
 * The reference stores an index in two fjall partitions (lsh.rs:62-120): "<uuid>-embeddings" maps a vector's 16 uuid
 * bytes to bincode(legacy) Embedding<N> -- exactly the N little-endian f32 that zh_index_append takes, no codec
 * needed (lsh.rs:91-97, lib.rs:15-18) -- and "<uuid>-trees" maps a tree's uuid to bincode(legacy) Node<N>
 * (lsh.rs:46-60,99-105).  The host shim, which links fjall, iterates the partitions; these functions turn the tree
 * values into the flat forest of zh_index_set_forest and back, so an existing database can be served by the GPU
 * path and a GPU-built forest can be saved in the reference's format.  Host code only (no GPU needed).  fjall's own
 * file layout is not read.  Byte layout: zebra_amd/csrc/zh_refformat.cpp (FORMAT UNVERIFIED against the crates).
 *
 * decode: `uuids` holds the keys of the n_vectors stored rows in row order (row i of zh_index_append <-> uuids[16 i]);
 * leaf ids that are not among them (vectors removed by the reference, whose trees keep them, lsh.rs:473-503) are
 * dropped and counted in out_unknown_ids (may be NULL). */
typedef struct zh_ref_forest zh_ref_forest;
ZH_API int zh_ref_forest_decode(uint32_t dim, size_t n_trees, const uint8_t *const *values, const size_t *lens,
                                size_t n_vectors, const uint8_t *uuids, zh_ref_forest **out, uint64_t *out_unknown_ids);
/* borrowed pointers, valid until zh_ref_forest_free */
ZH_API int zh_ref_forest_view(const zh_ref_forest *forest, zh_forest_view *out);
ZH_API void zh_ref_forest_free(zh_ref_forest *forest);
/* one tree of a flat forest as bincode(legacy) Node<N>; uuids = 16 bytes per row.  out == NULL: only *out_len (the
 * size needed) is written. */
ZH_API int zh_ref_tree_encode(const zh_forest_view *forest, uint32_t dim, uint32_t tree, const uint8_t *uuids,
                              uint64_t n_rows, uint8_t *out, size_t cap, size_t *out_len);

/* The database header, the `.zebra` file (core.rs:19-29 DatabaseInner, written by save_database core.rs:183-190, read by open
 * core.rs:92-102): bincode(legacy) { uuid, model: Mod, metric: Met, index_options }.  Met and Mod are type parameters of the
 * crate, so the file does not record them: the caller states the metric (only MinkowskiDistance / PNormDistance carry bytes:
 * their i32 power) and the length of the model's serialisation (0 for the reference's three unit-struct models).  Host code. */
typedef struct zh_ref_header {
    uint8_t uuid[16];         /* DatabaseInner::uuid: the prefix of the two partition names "<uuid>-embeddings" / "<uuid>-trees" */
    uint64_t max_node_size;   /* LSHIndexOptions (lsh.rs:124-129), usize as u64 */
    uint64_t num_trees;
    int32_t metric;           /* zh_metric as stated by the caller */
    int32_t power;            /* MinkowskiDistance / PNormDistance { power } (distance.rs:160-190); 0 for the other metrics */
    uint64_t model_off, model_len; /* where the model's own bytes sit inside the file */
} zh_ref_header;
ZH_API int zh_ref_header_decode(const uint8_t *bytes, size_t len, int metric, size_t model_len, zh_ref_header *out);
/* out == NULL: only *out_len (the size needed) is written */
ZH_API int zh_ref_header_encode(const zh_ref_header *header, const uint8_t *model_bytes, uint8_t *out, size_t cap, size_t *out_len);

/* ---- test / debug access: what the half-width scans computed for EVERY scored pair --------------------------------
 * The half-width scans (zh_set_sweep_mode 4 / 5 / 6) give every member of every visited leaf -- every (stored row, query) pair that
 * tree_result scores with Metric::distance (lsh.rs:310-323; distance.rs:23,41,106) -- an interval that must CONTAIN the reference's key; only
 * then are the returned ids / keys the reference's.  These two calls let a test check that claim pair by pair on the device's own numbers
 * (tests/test_gpu_intervals.py) instead of end to end: zh_debug_keep_raw(idx, 1) makes every later half-width batch keep a copy of the
 * scan's raw output {x^ . h^ / sigma_x, |x|^2} (the intervals overwrite it in place); zh_debug_scan_pairs returns, for the most recent
 * batch of `ctx` (NULL: the index's blocking context, i.e. the last zh_search_batch_device call; the context must be idle), one record per
 * scored pair.  lo / hi are f32 values in the scale the scan ranks in, mapped to order-preserving u32 ("sortable": bits ^ (sign ? ~0 :
 * 0x80000000)): L2 family: the canonical f32 sum of (x_i - q_i)^2; cosine, corrected key: the clipped distance 1 - cos; the reference's
 * literal key (distance.rs:23-25): key > 0 ? key : 2 - key for key = 1 - distance.  (lo, hi) = (0, ~0): nothing is certain about the pair
 * (it takes the exact path).  Not a product path: host-side copies of the whole batch's scratch.  While zh_debug_keep_raw is on, the d = 128
 * half-width sweep is never FUSED (a fused sweep -- zh_stats_t::approx_fused -- writes no per-pair result at all: zh_debug_scan_pairs after one
 * returns ZH_ESTATE). */
typedef struct zh_debug_pair {
    uint32_t row;          /* stored row (local: without id_base) */
    uint32_t query;        /* query of the batch (of the window: batch * b + i) */
    uint32_t lo, hi;       /* the interval, sortable f32 (valid when flags & 1) */
    float raw_s, raw_a2;   /* the scan's raw output for the pair (valid when zh_debug_scan_info::raw_kept) */
    uint32_t flags;        /* 1: the visit's raw pairs were turned into intervals (every visit that hands rows on); 2: a visit that takes FEWER
                            * than top_k rows of a longer leaf -- ranked by the reference's arithmetic, its intervals only gate the query's list */
    uint32_t visit;        /* index of the leaf visit the pair belongs to */
} zh_debug_pair;
typedef struct zh_debug_scan_info {
    uint32_t approx_scan;  /* as zh_stats_t::approx_scan: 1 VALU scan, 2 matrix-core scan, 3 leaf-major at half width; 0: the batch was not half-width */
    uint32_t queries, top_k;
    int32_t metric, cosine_mode;
    uint32_t raw_kept;     /* 1: raw_s / raw_a2 are valid */
    uint32_t overflow;     /* the batch's overflow word (non-zero: it was redone by the f32 path; the intervals are still what the scan made) */
    float bound_const;     /* zh_approx_bound for the batch ("Half-width scan: the bound") */
    float row_rho, rho_norm; /* the measured relative rounding error of the stored rows' fp16 copy (0 with f32 rows) */
    uint64_t pairs, visits;
} zh_debug_scan_info;
ZH_API int zh_debug_keep_raw(zh_index *idx, int on);
/* out (cap records; NULL with cap 0 to size: info->pairs) in key-slot order; qmeta (may be NULL): queries x 4 floats {1 / sigma_q, f32 |q|^2,
 * upper estimate of |q|, upper estimate of |q - h / sigma_q|} (NaN: nothing certain about the query) */
ZH_API int zh_debug_scan_pairs(zh_index *idx, zh_search_ctx *ctx, zh_debug_scan_info *info, zh_debug_pair *out, size_t cap, float *qmeta);

/* ---- instrumentation ------------------------------------------------------------------------ */
ZH_API int zh_set_profiling(zh_index *idx, int level); /* 0 off, 1 per-stage hipEvent timing, 2 + unique-row count */
ZH_API int zh_stats(zh_index *idx, zh_stats_t *out);
ZH_API int zh_stats_reset(zh_index *idx);
/* number of leading tree levels hashed by the dense MFMA kernel; -1 = choose per batch (default) */
ZH_API int zh_set_dense_levels(zh_index *idx, int levels);
/* How the distance sweep of a batch is organised: 1 = leaf by leaf (the rows of every visited leaf are gathered from HBM once
 * per group of <= 4 queries that visit it), 2 = table scan (every stored row is streamed from HBM once per batch window, in
 * address order, and scored against every query that visits one of its num_trees leaves; the queries come from L2), 0 = the
 * library chooses per batch from the counted work (default).  Results are bit-identical in both.
 * A batch whose hash came from row scores (zh_set_hash_mode) holds row . query for every stored row, i.e. every L2 / L2^2 / cosine
 * distance of the batch up to rounding.  In mode 0 (and 3 = the same, stated) such a batch is PREFILTERED instead of swept: per
 * (query, tree) the visited leaves' rows are judged on their scores with a rigorous rounding bound -- which `take` rows a leaf
 * hands over (lsh.rs:300-330), and which of those can still be among the k nearest -- and only the survivors, plus every visit the
 * bound cannot decide, are scored with the reference's arithmetic; ids, keys and counts stay bit-identical.  Modes 1 and 2 always
 * sweep.  (max_node_size <= 8 and no leaf longer than 64 rows, top_k <= 64, forests built by this library.)
 * The table scan of mode 0 reads HALF-WIDTH (fp16) copies of the queries where that pays (L2 / L2^2 / cosine keys, dim 256 / 384 /
 * 512 / 768 / 1024, top_k <= 256, two or more scored (row, query) pairs per stored row): 2 * dim instead of 4 * dim bytes per pair, an
 * INTERVAL per pair that contains the reference's key (the fp16 roundings measured, every f32 rounding bounded), the candidates
 * picked on the intervals and only the rows they cannot rule out scored with the reference's arithmetic -- ids, keys and counts
 * stay bit-identical.  With up to 16 trees the products run on the matrix cores from an fp16 copy of the STORED ROWS that the
 * index makes on first use and extends as rows are appended: + 2 * dim + 8 bytes per stored row of device memory (+50 % of the row
 * table; skipped, and the f32 rows read by a VALU kernel instead, when less than that plus a sixteenth of the device is free).
 * A list that runs over is redone by the f32 scan on the device, in stream order (no host round trip: safe for callers that
 * consume results in stream order).  Mode 2 keeps the f32 scan; 4 = the half-width scan wherever it is implemented (dim 128 ...
 * 1024), whatever the cost model says; 5 = as 4 with the VALU kernel only (no copy of the rows).
 * 128-d tables (SIFT-style shards) whose batches score 4M rows or more leaf by leaf get the same treatment on the leaf-major sweep: a row-major
 * fp16 copy of the rows under ONE power-of-two scale (+ 256 bytes per stored row; exact for integer-valued rows; rows the scale does not serve
 * are scored exactly), 16-row tiles on the matrix cores, the same intervals and exact passes behind; 6 = that sweep wherever it is
 * implemented, 1 keeps the f32 sweep.  zh_stats_t::approx_* report it. */
ZH_API int zh_set_sweep_mode(zh_index *idx, int mode);
/* How a batch that needs EVERY sign of the forest (small leaves: the reference's default max_node_size 5) gets them:
 * 1 = one dot product per (query, plane), 2 * b * planes * dim flop on the matrix cores; 2 = from row scores: a plane is built from
 * two stored rows a, b (build_hyperplane, lsh.rs:192-225), and in exact arithmetic w.x + c = (b.x - |b|^2/2) - (a.x - |a|^2/2),
 * so b * rows dot products decide all planes (planes / rows ~ 6.5x fewer flop at the default options); a sign closer to zero than
 * a rigorous bound on the rounding that separates the two computations is recomputed with point_is_above's own arithmetic, so
 * the bits are identical.  Only for forests built or grown by this library (an injected forest has arbitrary planes);
 * 0 = chosen per batch (default). */
ZH_API int zh_set_hash_mode(zh_index *idx, int mode);

ZH_API const char *zh_last_error(void);
ZH_API const char *zh_version(void);
/* Device memory.  With ZH_POOL=1 in the environment, blocks of 32 MiB and more that an index or a search context lets go of are kept by the library
 * (a process-wide cache keyed by size) and handed to the next buffer of a fitting size instead of going back to the driver ("order
 * effect": worth ~5 % on an index created late in a process; off by default).  hipMalloc failing empties the cache by itself; this call empties it
 * on request (e.g. before another library needs the memory).  Without ZH_POOL=1 it does nothing. */
ZH_API int zh_trim_device_memory(void);

#ifdef __cplusplus
}
#endif
#endif /* ZEBRA_HIP_H */
