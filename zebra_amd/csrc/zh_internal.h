// zh_internal.h -- shared declarations between the kernel translation units and the C ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/zebra_hip.h"

// sets zh_last_error() for the calling thread and returns `code`
int zh_set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// (zh_shard.hip) leaf visits per (query, tree) pair, running mean of the index's batches so far (0 before the first)
double zh_index_visits_per_pair(zh_index *ix);
// (zh_shard.hip) a context whose batch was begun and will never be finished goes back to idle
void zh_search_ctx_abandon(zh_search_ctx *c);
// the caller consumes a batch's outputs in STREAM order, before zh_search_wait has returned (the sharded search enqueues its
// all-gather behind finish): the context then never takes a path that may redo the batch from the host (the prefilter)
void zh_search_ctx_stream_ordered(zh_search_ctx *c);

// ---- one leaf visit of the walk (tree_result, lsh.rs:290-348): score `len` rows of a leaf for
// query `b`, keep the `take` smallest.  row_off / cand_off are the visit's slices of the key
// scratch and of the candidate pool.
struct ZhVisit {
    uint32_t b;
    uint32_t leaf_off;  // offset into leaf_ids
    uint32_t len;
    uint32_t take;      // min(n, len)
    uint64_t row_off;
    uint64_t cand_off;
    uint32_t node;      // the leaf's node index
    uint32_t pad;
};

// Visits of one leaf by different queries of the batch (or window) are swept together, `group` at a time (2 or 4, chosen
// per index by zh_group_size): the leaf's rows cross HBM once per group instead of once per query.
#define ZH_GROUP_MAX 4
struct ZhGroup {
    uint32_t leaf_off, len, gsize;
    uint32_t take4;  // byte j = min(take of member j, 255) (join_group)
    uint32_t b[ZH_GROUP_MAX];
    uint64_t key_off[ZH_GROUP_MAX];  // the member visits' row_off (slice of the key scratch)
};
// 4 queries per group (ZH_GROUP=2|4 forces; A/B in profiles/).
uint32_t zh_group_size(uint32_t dim);

// per (query, tree) pair counts produced by the walk's first pass, then their exclusive scans
struct ZhPairCounts {
    uint32_t visits, rows, takes, pad;
};
struct ZhTotals {
    uint64_t visits, rows, takes, flags;  // flags bit 0: the visit log overflowed (the emit walk must run)
    uint64_t groups, group_rows;  // filled by the leaf scan
    unsigned long long hash_fixups;  // signs the row-score hash (zh_score.hip) recomputed exactly
};

// Visit log of the walk's (single) pass: a pair's visits beyond the ZH_INLINE_VISITS inline ones are appended as
// {leaf node, take} to 512-byte chunks handed out by a bump allocator; entry 0 of a chunk is {next chunk, unused}.
#define ZH_LOG_CHUNK 64
struct ZhLogCtl {
    uint32_t next_chunk, overflow;
};
struct ZhWalkLog {
    uint2 *pool;          // capacity * ZH_LOG_CHUNK entries
    uint32_t capacity;    // chunks
    uint32_t *head;       // first chunk of every pair (valid when the pair has more than ZH_INLINE_VISITS visits)
    ZhLogCtl *ctl;
    uint32_t leaf_entries;  // 1: entries are {offset into leaf_ids, take | length << 16} (a prefiltered batch: its reader needs no node record)
};

// Blocked view of the forest for walks that find every sign precomputed (ZH_BLOCK_NODES, zh_api.hip build_blocks): every
// maximal subtree of at most ZH_BLOCK_NODES nodes is one BLOCK, its nodes stored contiguously in pre-order as 16-byte
// records {plane | -1, inner: left_local | right_local << 16 / leaf: offset into leaf_ids, inner: nodes in the block (root
// record only) / leaf: length, global node id}.  The nodes above the blocks ("upper" nodes, all inner, numbered densely)
// keep two 16-byte records: {plane, left ref, right ref, 0} and {plane of the left child, plane of the right child, 0, 0}
// -- a child's sign can be requested together with its record.  A ref is >= 0 for an upper node and -(offset of the
// block's first record + 1) for a block: no directory lookup between a ref and its data.
#define ZH_BLOCK_NODES 64
// Round 5 (inner_only): a block = a maximal subtree of at most ZH_BLOCK_INNER INNER nodes, one 32-byte record (two int4) per inner node in
// pre-order; its leaves live in their parent's record (zh_api.hip build_blocks_inner); a ref is -(index of the block's first 32-byte record + 1).
#define ZH_BLOCK_INNER 63
struct ZhBlocksDev {
    const int4 *recs;           // all blocks' node records (+ ZH_BLOCK_NODES records of padding: a wave always loads 64)
    const int4 *upper;          // 2 records per upper node
    const int2 *root;           // per tree: {ref, plane of the root when it is an upper node}
    uint32_t n_blocks, n_upper;
    uint32_t inner_only;        // 1: the round-5 records
    const int4 *recs_b;         // ... their second halves (same indices as recs)
};

struct ZhForestDev {
    const int32_t *node_plane, *node_left, *node_right;
    // one 16-byte record per node for the walk: {plane, left, right, bits of the plane's constant} for an internal node,
    // {-1, leaf offset, leaf length, 0} for a leaf -- one dependent load per step instead of four
    const int4 *node_pack;
    const uint32_t *roots;
    const float *planes, *consts;
    const uint32_t *leaf_ids;
    uint32_t n_nodes, n_planes, n_trees;
    uint32_t group;  // queries per sweep group (zh_group_size)
};

#if defined(__HIPCC__)
// point_is_above's dot product: sequential k-ascending fma chain from +0 (the order contract of the hash);
// loads are issued 16 at a time ahead of the chain
__device__ __forceinline__ bool zh_plane_above(const float *__restrict__ w, float c, const float *__restrict__ x,
                                               uint32_t d) {
    float acc = 0.0f;
    if ((d & 3u) == 0) {
        const float4 *w4 = reinterpret_cast<const float4 *>(w);
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        const uint32_t n4 = d / 4;
        uint32_t k = 0;
        for (; k + 8 <= n4; k += 8) {
            float4 a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { a[u] = w4[k + u]; b[u] = x4[k + u]; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                acc = __builtin_fmaf(a[u].x, b[u].x, acc);
                acc = __builtin_fmaf(a[u].y, b[u].y, acc);
                acc = __builtin_fmaf(a[u].z, b[u].z, acc);
                acc = __builtin_fmaf(a[u].w, b[u].w, acc);
            }
        }
        for (; k < n4; k++) {
            float4 a = w4[k], b = x4[k];
            acc = __builtin_fmaf(a.x, b.x, acc);
            acc = __builtin_fmaf(a.y, b.y, acc);
            acc = __builtin_fmaf(a.z, b.z, acc);
            acc = __builtin_fmaf(a.w, b.w, acc);
        }
    } else {
        for (uint32_t k = 0; k < d; k++) acc = __builtin_fmaf(w[k], x[k], acc);
    }
    return ((double)acc + (double)c) >= 0.0;  // lsh.rs:40-42; NaN -> false
}
#endif

#define ZH_SORT_N 4096        // entries of the LDS sort buffer of the select / final kernels
#define ZH_INLINE_VISITS 32    // visits a pair may record in the walk's first pass

// ---- launchers (zh_search.hip) ---------------------------------------------------------------
hipError_t zh_launch_hash_dense(const float *dQ, uint32_t B, const float *dPlanes, const float *dConsts,
                                uint32_t P, uint32_t d, uint32_t *dBits, uint32_t words_per_q, float *dDots,
                                hipStream_t s);
hipError_t zh_launch_qnorm(const float *dQ, uint32_t B, uint32_t d, float *dQQ, hipStream_t s);
hipError_t zh_launch_walk_count(ZhForestDev f, const float *dQ, uint32_t B, uint32_t d, int32_t n,
                                const uint32_t *dBits, uint32_t words_per_q, uint32_t P_dense, ZhPairCounts *dCounts,
                                ZhVisit *dInline, uint32_t *dLeafCount, ZhWalkLog log, hipStream_t s);
// the counting pass for all-dense signs over the blocked forest: one memory round trip per BLOCK instead of per node
hipError_t zh_launch_walk_blocked(ZhForestDev f, ZhBlocksDev blk, uint32_t B, int32_t n, const uint32_t *dBits,
                                  uint32_t words_per_q, ZhPairCounts *dCounts, ZhVisit *dInline, uint32_t *dLeafCount,
                                  ZhWalkLog log, const uint32_t *dUnc /* null, or the row-score hash's flagged signs */, const float *dQ,
                                  uint32_t d, hipStream_t s);
// places every visit recorded by the counting pass (inline + log) and joins the leaf groups: the cheap, flat
// replacement of the emit walk whenever the log did not overflow
hipError_t zh_launch_expand(ZhForestDev f, uint32_t B, const ZhPairCounts *dCounts, const ZhVisit *dInline,
                            const uint64_t *dRowBase, const uint64_t *dCandBase, const uint64_t *dVisitBase,
                            ZhVisit *dVisits, const uint32_t *dLeafCount, uint32_t *dLeafFill,
                            const uint32_t *dGroupBase, const uint64_t *dGroupRowBase, ZhGroup *dGroups,
                            uint64_t *dGroupRowOff, ZhWalkLog log, hipStream_t s);
// zeroes what a batch starts from zero: leaf visit counts / fill cursors, the log allocator, the packed leaf counter
hipError_t zh_launch_batch_init(uint32_t *dLeafCount, uint32_t *dLeafFill, uint32_t n_nodes, ZhLogCtl *dLogCtl,
                                ZhTotals *dTotals, hipStream_t s);
// per leaf node: visits -> groups; exclusive scans of groups and of groups * len over the nodes
hipError_t zh_launch_leaf_scan(ZhForestDev f, const uint32_t *dLeafCount, uint32_t *dGroupBase,
                               uint64_t *dGroupRowBase, ZhTotals *dTotals, hipStream_t s);
// exclusive scans over the pairs; the three base arrays have n_pairs + 1 entries
hipError_t zh_launch_pair_scan(const ZhPairCounts *dCounts, uint32_t n_pairs, uint64_t *dRowBase,
                               uint64_t *dCandBase, uint64_t *dVisitBase, ZhTotals *dTotals, const ZhLogCtl *dLogCtl,
                               hipStream_t s);
hipError_t zh_launch_walk_emit(ZhForestDev f, const float *dQ, uint32_t B, uint32_t d, int32_t n,
                               const uint32_t *dBits, uint32_t words_per_q, uint32_t P_dense,
                               const ZhPairCounts *dCounts, const ZhVisit *dInline, const uint64_t *dRowBase,
                               const uint64_t *dCandBase, const uint64_t *dVisitBase, ZhVisit *dVisits,
                               const uint32_t *dLeafCount, uint32_t *dLeafFill, const uint32_t *dGroupBase,
                               const uint64_t *dGroupRowBase, ZhGroup *dGroups, uint64_t *dGroupRowOff, hipStream_t s);
// dWaveGroup: the wave-start table of zh_launch_wave_groups (ceil(R_grouped / 64) entries), or nullptr
hipError_t zh_launch_sweep(const float *dX, uint32_t d, const float *dQ, const float *dQQ, const ZhGroup *dGroups,
                           const uint64_t *dGroupRowOff, uint64_t n_groups, const uint32_t *dWaveGroup,
                           const uint32_t *dLeafIds, uint64_t R_grouped, int metric, int mode, uint64_t *dKeys,
                           uint32_t group, hipStream_t s, const uint32_t *dRunIf = nullptr);
bool zh_sweep_has_predicate(uint32_t d, int metric);
// waveGroup[w] = group of flat row 64 w, for every wave start of [0, R_grouped)
hipError_t zh_launch_wave_groups(const ZhGroup *dGroups, const uint64_t *dGroupRowOff, uint64_t n_groups, uint32_t *dWaveGroup,
                                 hipStream_t s);
// ---- the table-scan sweep (zh_search.hip): the stored rows streamed once, each scored against every query that visits
// one of its leaves.  rowLeaf is n_rows x T {leaf node, position in the leaf}, built per forest by zh_launch_row_leaf from
// node_pack, a node -> tree map (UINT32_MAX for nodes no root reaches) and leaf_ids.
#define ZH_SCAN_NE 4     // (row, tree) entries per lane of a table-scan wave: RW * T <= 64 * ZH_SCAN_NE
bool zh_scan_sweep_supported(uint32_t d, uint32_t T, int metric);
uint32_t zh_scan_rows_per_wave(uint32_t T);
hipError_t zh_launch_row_leaf(const int4 *dNodePack, const uint32_t *dNodeTree, uint32_t n_nodes, const uint32_t *dLeafIds,
                              uint32_t T, uint64_t n_rows, uint2 *dRowLeaf, hipStream_t s);
// per batch (after the groups are filled): bit n of dBits = leaf node n is visited; dNodeVisit[n] = {visits, first group, query and
// key slice of the first visit} of the visited leaves
hipError_t zh_launch_node_visits(const uint32_t *dLeafCount, const uint32_t *dGroupBase, const ZhGroup *dGroups, uint32_t n_nodes,
                                 uint32_t *dBits, uint4 *dNodeVisit, hipStream_t s);
hipError_t zh_launch_scan_sweep(const float *dX, uint32_t d, uint64_t n_rows, const float *dQ, const float *dQQ,
                                const uint2 *dRowLeaf, uint32_t T, const uint32_t *dVisitBits, const uint4 *dNodeVisit,
                                const ZhGroup *dGroups, uint32_t group, int metric, int param, uint64_t *dKeys,
                                const uint32_t *dRunIf /* null, or: run only when this device word is nonzero */, hipStream_t s);
// flat rows covered by one sweep launch (a batch is issued as ceil(R / this) launches)
uint64_t zh_sweep_rows_per_launch(uint32_t d);
// max_leaf_len: the longest leaf of the forest (picks the LDS footprint of the select blocks)
hipError_t zh_launch_select(const ZhVisit *dVisits, uint64_t n_visits, const uint32_t *dLeafIds,
                            const uint64_t *dKeys, uint64_t *dCandKeys, uint32_t *dCandIds, uint32_t max_leaf_len,
                            const uint32_t *dRunIf, hipStream_t s);
hipError_t zh_launch_final(const uint64_t *dCandBase, uint32_t B, uint32_t T, uint32_t k, const uint64_t *dCandKeys,
                           const uint32_t *dCandIds, uint64_t id_base, uint64_t *dOutIds, uint64_t *dOutKeys,
                           uint32_t *dOutCounts, const uint32_t *dRunIf, hipStream_t s);
// stride64 / stride32: distance between consecutive shards' ids (= keys) in u64 units and counts in u32 units
// (0 = contiguous [S][B][k] / [S][B])
hipError_t zh_launch_merge(uint32_t S, uint32_t B, uint32_t k, const uint64_t *dIds, const uint64_t *dKeys,
                           const uint32_t *dCounts, uint64_t *dOutIds, uint64_t *dOutKeys, uint32_t *dOutCounts,
                           uint64_t stride64, uint64_t stride32, hipStream_t s);
// plain distance of n contiguous rows against one query (zh_distance_batch)
#define ZH_DISTANCE_SCRATCH_BYTES (sizeof(ZhGroup) + 16)
hipError_t zh_launch_distance_rows(const float *dX, uint64_t n, uint32_t d, const float *dq, int metric, int mode,
                                   uint64_t *dKeys, void *dScratch, hipStream_t s);

// ---- the table scan with half-width queries (zh_approx.inl): intervals instead of keys, exact keys for the few rows they cannot
// decide.  ctl[0] = visits sent to the exact path, ctl[1] = overflow bits (1 a query's candidate list, 2 a query's survivors,
// 4 the table of exact visits, 8 the exact visits' key scratch) -- nonzero: the exact scan + select + final enqueued behind
// redo the batch (their predicate), ctl[2] = key scratch handed out, ctl[3] = survivors scored exactly, ctl[4] = list entries,
// ctl[5] / ctl[6] = tile columns / pairs of the matrix-core scan's listed waves (a column per distinct query: how much the pairs share).
#define ZH_APX_CTL_WORDS 8
struct ZhApprox {
    const void *Qh;          // the fp16 copy of the batch's queries (2 * d bytes each, qhalf_kernel's layout)
    const float4 *qmeta;     // per query {1 / sigma, |q|^2, upper estimate of |q|, upper estimate of |q - h / sigma|}
    uint64_t *iv;            // per (row, query) pair, in the pair's key slot: from the scan {x . h, |x|^2} (f32 bits); select_tau_kernel
                             // turns a visit's entries into sortable lo | sortable hi << 32 in place
    uint32_t *list_lo, *list_hi, *list_id;  // per query: capq slots
    uint32_t *qcount;
    uint32_t capq;
    uint32_t *tauv;          // per visit: the take-th smallest hi (sortable) of a visit that takes top_k rows of a longer leaf
    uint32_t *qtau;          // per query: the smallest of them -- top_k candidates of the query have keys at or below it
    uint2 *ex_visits;        // the visits for the exact path: {index, their slice of the key scratch}
    uint32_t ex_cap;
    uint64_t *ex_keys, *ex_ckeys;  // their rows' canonical keys; the `take` chosen
    uint32_t *ex_cids;
    uint32_t ex_rows_cap;
    uint32_t *ctl;
    // the scan on the matrix cores (scan_mfma_kernel): the stored rows are rounded to fp16 as well
    const void *row_half;    // the fp16 copy of the stored rows, tiles of 16 rows in the A operand's order (row_half_kernel)
    const float2 *row_meta;  // per stored row {|x|^2, 1 / sigma_x (NaN: nothing certain about the row)}
    float row_rho;           // |x - xh / sigma_x| <= row_rho |x| for every usable stored row (0 with f32 rows)
    float rho_norm;          // = row_rho where |x|^2 itself comes from the rounded row (sweep128h_kernel), else 0
    uint32_t mfma;           // zh_approx_bound's kind: 0 VALU scans, 1 scan_mfma_kernel, 2 sweep128h_kernel
    uint32_t n_queries;      // queries of the internal batch, entries of iv (diagnostic builds check every index against them: -DZH_SCAN_GUARD)
    uint64_t iv_cap;
    float nx_max;            // > 0: no stored row's norm estimate (approx_interval's nx) exceeds it -- the byte copy: sqrt(128) * 255 (fused sweep's pre-test)
};
uint32_t zh_approx_groups(uint32_t d);
bool zh_approx_pays(uint32_t d);
float zh_approx_bound(int metric, uint32_t d, int mfma);
bool zh_scan_mfma_supported(uint32_t d, uint32_t T);
// the fp16 copy of rows [row0, row0 + n_rows) in scan_mfma_kernel's operand order (2 * d bytes per row, tiles of 16 rows), per row {|x|^2,
// 1 / sigma_x}; *dRhoMax = the largest relative rounding error of a row (f32 bits, atomicMax)
hipError_t zh_launch_row_half(const float *dX, uint64_t row0, uint64_t n_rows, uint32_t d, void *dXh, float2 *dRowMeta, uint32_t *dRhoMax,
                              const uint32_t *dPerm /* null, or: position p < perm_rows holds row dPerm[p] */, uint64_t perm_rows, hipStream_t s);
// the matrix-core scan's row order: the permutation (zh_order.hip) and the row -> leaf table gathered into it
// (zh_order.hip) dPerm[p] = the row at position p of the order (leaf in tree 0, leaf in tree 1[, leaf in tree 2], row id); synchronises the stream
hipError_t zh_launch_scan_order(const uint2 *dRowLeaf, uint64_t n_rows, uint32_t T, uint32_t n_keys, uint32_t *dPerm, hipStream_t s);
// ... and what an order is worth: (adjacent positions, tree) combinations in the same leaf (dPerm null: id order)
hipError_t zh_launch_order_agreement(const uint2 *dRowLeaf, const uint32_t *dPerm, uint64_t n_rows, uint32_t T, unsigned long long *dOut, hipStream_t s);
hipError_t zh_launch_permute_row_leaf(const uint2 *dRowLeaf, const uint32_t *dPerm, uint64_t perm_rows, uint64_t n_rows, uint32_t T, uint2 *dOut, hipStream_t s);
bool zh_scan_approx_supported(uint32_t d, uint32_t T, int metric);
hipError_t zh_launch_qhalf(const float *dQ, uint32_t B, uint32_t d, void *dQh, float4 *dQmeta, int layout, hipStream_t s);
// d = 128, leaf by leaf at half width (sweep128h_kernel): the table's common scale from its largest finite element, the row-major fp16 copy, the sweep
hipError_t zh_launch_absmax(const float *dX, uint64_t n, uint32_t *dOut, hipStream_t s);
hipError_t zh_launch_row_half128(const float *dX, uint64_t row0, uint64_t n_rows, float sigma, void *dXh, uint32_t *dRhoMax, hipStream_t s);
hipError_t zh_launch_sweep128h(const void *dXh, const void *dQh, float inv, const ZhGroup *dGroups, const uint64_t *dGroupRowOff, uint64_t n_groups,
                               const uint32_t *dWaveGroup, const uint32_t *dLeafIds, uint64_t R_grouped, uint64_t *dIv, hipStream_t s,
                               const ZhApprox *fuse = nullptr, int fuse_kinda = 0, uint32_t k_top = 0, float Kc = 0.f, bool byte_rows = false);
// ... and of a table whose every element is an integer in 0 .. 255 (SIFT descriptors): an EXACT copy of 128 bytes per row (*dNotBytes |= 1 when an
// element of rows row0 .. is anything else: the copy is then void); byte_rows above = dXh is that copy (the lean kernels only)
uint64_t zh_sweep128h_rows_per_launch(bool lean, bool byte_rows);  // stored rows one launch of the sweep takes (the host's launch accounting)
hipError_t zh_launch_row_byte128(const float *dX, uint64_t row0, uint64_t n_rows, void *dXb, uint32_t *dNotBytes, hipStream_t s);
// (fused sweep) the exact path's visits, as select_tau_kernel lists them
hipError_t zh_launch_exact_register(const ZhVisit *dVisits, uint64_t n_visits, uint32_t k, ZhApprox ap, hipStream_t s);
hipError_t zh_launch_scan_approx(const float *dX, uint32_t d, uint64_t n_rows, ZhApprox ap, const uint2 *dRowLeaf, uint32_t T,
                                 const uint32_t *dVisitBits, const uint4 *dNodeVisit, const ZhGroup *dGroups, uint32_t group, int metric,
                                 int mode, hipStream_t s);
hipError_t zh_launch_select_interval(const ZhVisit *dVisits, uint64_t n_visits, uint32_t k, const uint32_t *dLeafIds, ZhApprox ap,
                                     int metric, int mode, uint32_t d, hipStream_t s);
// the exact visits, then per query: duplicates out, tau, the survivors' canonical keys, top_k
hipError_t zh_launch_final_interval(const ZhVisit *dVisits, const float *dX, uint32_t d, const float *dQ, const float *dQQ, uint32_t B,
                                    uint32_t k, const uint32_t *dLeafIds, int metric, int mode, uint64_t id_base, ZhApprox ap,
                                    uint64_t *dOutIds, uint64_t *dOutKeys, uint32_t *dOutCounts, uint32_t max_leaf_len, hipStream_t s);

// ---- launchers (zh_score.hip): every sign of a forest built from stored rows, from N row scores per query --------
// Prefilter (zh_search.hip, "Prefilter"): a batch hashed from row scores picks the rows that can be among a pair's k best from
// those scores; only they are scored with the reference's arithmetic.  Lists: one per (tree, query), `cap` slots, list (t, b) at
// (t * B + b) * cap.  ctl[0] = ambiguous visits appended, ctl[1] = overflow bits (1 a list, 2 a list in the exact pass, 4 the
// ambiguous-visit table, 8 a leaf longer than a wave, 16 a query's lists together longer than the final sort), ctl[2] = rows scored exactly.
struct ZhPrefilter {
    const float *S;            // row scores S[row * Bp + b]
    uint32_t Bp;
    const float4 *leaf_meta;   // per slot of leaf_ids: {the row id (its bits), |r|^2 / 2, |r|, 0} of the row stored there
    const float *qnorm;        // |q| per query
    uint32_t *rows, *counts;
    float *tau;                // per list: k of the pair's rows have keys at or below this value (the scale of the prefilter's v); +inf: fewer than k
    uint32_t cap;
    uint4 *amb;                // {pair, leaf_off, len, take}
    uint32_t amb_cap;
    uint32_t *ctl;
};
float zh_prefilter_bound(int metric, uint32_t d);
hipError_t zh_launch_leaf_meta(const uint32_t *dLeafIds, uint64_t n, const float *dHalfN2, const float *dNorm, float4 *dOut, hipStream_t s);
hipError_t zh_launch_prefilter(ZhForestDev f, uint32_t d, uint32_t B, uint32_t k, int metric, int mode, const ZhPairCounts *dCounts,
                               const ZhVisit *dInline, ZhWalkLog log, ZhPrefilter pf, hipStream_t s);
hipError_t zh_launch_prefilter_exact(ZhForestDev f, uint32_t d, const float *dX, const float *dQ, const float *dQQ, uint32_t B, int metric,
                                     int mode, uint64_t id_base, ZhPrefilter pf, uint64_t *dKeys, uint64_t *dIds, hipStream_t s);
hipError_t zh_launch_final_lists(uint32_t T, uint32_t B, uint32_t k, uint32_t cap, const uint64_t *dKeys, const uint64_t *dIds,
                                 const uint32_t *dCounts, uint64_t *dOutIds, uint64_t *dOutKeys, uint32_t *dOutCounts,
                                 uint32_t *dOver /* the prefilter's overflow word: bit 16 = a query's lists hold more than the final sort */, hipStream_t s);
hipError_t zh_launch_row_norms(const float *dX, uint64_t n, uint32_t d, float *dHalfN2 /* may be null */, float *dNorm, hipStream_t s);
// the score table of exactly four queries (dQ4: 4 x d): dS[row][0..3]
hipError_t zh_launch_row_scores4(const float *dX, uint64_t n, uint32_t d, const float *dQ4, float *dS, hipStream_t s);
// dS: scores [n_rows][B] (row . query, from zh_launch_hash_dense with the roles swapped); dSamples: the two sample rows of every
// plane (UINT32_MAX = a default zero vector); writes the sign words of all P planes for the B queries (B % 4 == 0), the signs
// inside the rounding bound recomputed exactly (list of fix_cap entries; *dFixCount must be 0 on entry and receives their number)
// dPlaneHab: per plane {|a|^2/2, |b|^2/2, |a| + |b|, 0} of its sample rows (zh_launch_plane_hab), read in plane order by the signs kernel
hipError_t zh_launch_plane_hab(const uint2 *dSamples, uint32_t P, const float *dHalfN2, const float *dRowNorm, float4 *dOut, hipStream_t s);
hipError_t zh_launch_score_unc_fix(const float *dQ, uint32_t B, uint32_t d, const float *dPlanes, const float *dConsts, uint32_t P,
                                   uint32_t *dBits, const uint32_t *dUnc, uint32_t wpq, hipStream_t s);
hipError_t zh_launch_score_signs(const float *dS, uint32_t B, const uint2 *dSamples, uint32_t P, const float *dHalfN2, const float *dRowNorm,
                                 const float4 *dPlaneHab, const float *dQNorm, const float *dQ, uint32_t d, const float *dPlanes, const float *dConsts,
                                 uint32_t *dBits, uint32_t wpq, uint2 *dFixList, uint32_t fix_cap, unsigned long long *dFixCount, uint32_t *dUnc /* null: list + fix-up kernel */,
                                 hipStream_t s);

// ---- launchers (zh_build.hip) ----------------------------------------------------------------
struct ZhBuildNode {   // an active (to be split) node of the current level
    uint64_t seg_start;  // global position in perm (tree * N + offset)
    uint32_t len;
    uint32_t plane;      // plane index this node's hyperplane is written to
    uint64_t sample_a, sample_b;  // row numbers; UINT64_MAX = the all-zero default vector
    uint32_t first_chunk, n_chunks;
};
struct ZhBuildChunk {
    uint32_t node;       // index into the level's ZhBuildNode array
    uint32_t count;      // <= 256
    uint64_t pos;        // global position of the chunk's first row in perm
};

// incremental insert (lsh.rs:350-382): descend a stored row from `node` to a leaf
struct ZhDescend {
    uint32_t row, node, depth, tree;
    uint64_t path;
};
// node_pack from the three node arrays and the plane constants
hipError_t zh_launch_pack_nodes(const int32_t *dPlane, const int32_t *dLeft, const int32_t *dRight, const float *dConsts,
                                int4 *dPack, uint32_t n_nodes, hipStream_t s);
hipError_t zh_launch_descend(ZhForestDev f, const float *dX, uint32_t d, ZhDescend *dItems, uint32_t n, hipStream_t s);

// 64-bit content hash of every stored row (order-sensitive, exact integer arithmetic): deduplicate
hipError_t zh_launch_row_hash(const float *dX, uint64_t n, uint32_t d, uint64_t *dHash, hipStream_t s);

// out[i] = 1 when rows pairs[2i] and pairs[2i+1] have identical f32 bit patterns
hipError_t zh_launch_rows_equal(const float *dX, uint32_t d, const uint32_t *dPairs, uint32_t n_pairs, uint8_t *dOut,
                                hipStream_t s);

hipError_t zh_launch_synth_rows(float *dX, uint64_t n, uint32_t d, uint64_t seed, uint64_t row0, int kind,
                                hipStream_t s);
hipError_t zh_launch_synth_queries(float *dOut, uint64_t seed_rows, uint64_t seed_q, uint64_t n_rows, uint64_t b0,
                                   uint64_t b, uint32_t d, int kind, hipStream_t s);
hipError_t zh_launch_iota_perm(uint32_t *dPerm, uint64_t N, uint32_t T, hipStream_t s);
hipError_t zh_launch_make_planes(const float *dX, uint32_t d, const ZhBuildNode *dNodes, uint32_t n_nodes,
                                 float *dPlanes, float *dConsts, hipStream_t s);
hipError_t zh_launch_classify(const float *dX, uint32_t d, const uint32_t *dPerm, const ZhBuildNode *dNodes,
                              const ZhBuildChunk *dChunks, uint32_t n_chunks, const float *dPlanes,
                              const float *dConsts, uint8_t *dFlags, uint32_t *dChunkAbove, hipStream_t s);
hipError_t zh_launch_scan_u32(const uint32_t *dIn, uint32_t *dOut /* n+1 */, uint64_t n, uint32_t *dTmp /* >= n/1024+2 */,
                              hipStream_t s);
hipError_t zh_launch_scatter(const uint32_t *dPermIn, uint32_t *dPermOut, const ZhBuildNode *dNodes,
                             const ZhBuildChunk *dChunks, uint32_t n_chunks, const uint8_t *dFlags,
                             const uint32_t *dChunkScan, uint32_t *dNodeAbove, hipStream_t s);
