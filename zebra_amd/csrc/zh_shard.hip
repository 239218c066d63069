// zh_shard.hip -- the multi-GPU exchange behind the C ABI: a shard group is this rank's zh_index plus an RCCL
// communicator; one search over the sharded index = local search -> ONE in-place ncclAllGather of the packed
// [ids | keys | counts] result -> merge kernel on every rank (what replaces the query loop of
// /root/reference/src/database/core.rs:299-303 when the stored rows are sharded, README.md:31).
// Built on the public entry points of zebra_hip.h only (zh_search_begin / finish / wait, zh_merge_topk_packed_device),
// so the single-GPU pipeline is exactly the one the parity tests exercise.  Host code; no kernels here.
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "zh_internal.h"

#define FAIL(code, ...) zh_set_error(code, __VA_ARGS__)
#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return FAIL(e_ == hipErrorOutOfMemory ? ZH_ENOMEM : ZH_EHIP, "%s: %s (%s:%d)", #expr,             \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                           \
    } while (0)
#define NCCLCHK(expr)                                                                                         \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess) return FAIL(ZH_EHIP, "%s: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

struct zh_shard_group {
    zh_index *ix = nullptr;
    ncclComm_t comm = nullptr;
    uint32_t n_ranks = 0, rank = 0;
    int device = 0;
    std::mutex mu;           // collectives of one group are issued one at a time, in call order
    zh_shard_ctx *dctx = nullptr;  // the blocking entry points run on this context
    void *dQ = nullptr, *dOut = nullptr;  // staging of the host-pointer variant
    size_t capQ = 0, capOut = 0;
};

struct zh_shard_ctx {
    zh_shard_group *g = nullptr;
    zh_search_ctx *sc = nullptr;
    hipStream_t light = nullptr;   // hash / walk / select / final of this batch: high priority
    hipStream_t xs = nullptr;      // all-gather + merge: normal priority (never behind a sweep launch)
    hipEvent_t ev_final = nullptr, ev_xdone = nullptr;
    uint64_t *gathered = nullptr;  // [n_ranks][W] packed results; slot `rank` is this rank's own (in-place all-gather)
    size_t cap_words = 0, W = 0, B = 0, k = 0;  // B = every query of the window
    size_t nwin = 1, bwin = 0;
    uint64_t *merged = nullptr;    // a window's merged results before they are handed out per batch
    size_t cap_merged = 0;
    bool xused = false;
    int state = 0;  // 0 idle, 1 begun, 2 finished
};

static int set_dev(const zh_shard_group *g) {
    hipError_t e = hipSetDevice(g->device);
    if (e != hipSuccess) return FAIL(ZH_EHIP, "hipSetDevice(%d): %s", g->device, hipGetErrorString(e));
    return ZH_OK;
}

extern "C" int zh_shard_unique_id(uint8_t out_id[ZH_UNIQUE_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == ZH_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!out_id) return FAIL(ZH_EINVAL, "zh_shard_unique_id: null argument");
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    memcpy(out_id, &id, sizeof id);
    return ZH_OK;
}

extern "C" int zh_shard_group_create(zh_index *shard, const uint8_t id[ZH_UNIQUE_ID_BYTES], uint32_t n_ranks, uint32_t rank,
                                     zh_shard_group **out) {
    if (!shard || !id || !out) return FAIL(ZH_EINVAL, "zh_shard_group_create: null argument");
    if (n_ranks == 0 || n_ranks > 1024 || rank >= n_ranks) return FAIL(ZH_EINVAL, "zh_shard_group_create: rank %u of %u", rank, n_ranks);
    zh_shard_group *g = new (std::nothrow) zh_shard_group();
    if (!g) return FAIL(ZH_ENOMEM, "out of host memory");
    g->ix = shard; g->n_ranks = n_ranks; g->rank = rank; g->device = zh_index_device(shard);
    int rc = set_dev(g);
    if (rc) { delete g; return rc; }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t r = ncclCommInitRank(&g->comm, (int)n_ranks, uid, (int)rank);
    if (r != ncclSuccess) { delete g; return FAIL(ZH_EHIP, "ncclCommInitRank(rank %u of %u): %s", rank, n_ranks, ncclGetErrorString(r)); }
    rc = zh_shard_ctx_create(g, &g->dctx);
    if (rc) { ncclCommDestroy(g->comm); delete g; return rc; }
    *out = g;
    return ZH_OK;
}

extern "C" void zh_shard_group_destroy(zh_shard_group *g) {
    if (!g) return;
    hipSetDevice(g->device);
    zh_shard_ctx_destroy(g->dctx);
    if (g->dQ) hipFree(g->dQ);
    if (g->dOut) hipFree(g->dOut);
    if (g->comm) ncclCommDestroy(g->comm);
    delete g;
}

extern "C" uint32_t zh_shard_group_ranks(const zh_shard_group *g) {
    if (!g) return 0;
    int n = 0;
    return ncclCommCount(g->comm, &n) == ncclSuccess ? (uint32_t)n : 0;
}
extern "C" uint32_t zh_shard_group_rank(const zh_shard_group *g) {
    if (!g) return 0;
    int r = 0;
    return ncclCommUserRank(g->comm, &r) == ncclSuccess ? (uint32_t)r : 0;
}

extern "C" int zh_shard_ctx_create(zh_shard_group *g, zh_shard_ctx **out) {
    if (!g || !out) return FAIL(ZH_EINVAL, "zh_shard_ctx_create: null argument");
    int rc = set_dev(g);
    if (rc) return rc;
    zh_shard_ctx *c = new (std::nothrow) zh_shard_ctx();
    if (!c) return FAIL(ZH_ENOMEM, "out of host memory");
    c->g = g;
    auto bail = [&](int code) { zh_shard_ctx_destroy(c); return code; };
    if ((rc = zh_search_ctx_create(g->ix, &c->sc))) return bail(rc);
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
    hipError_t e = hipStreamCreateWithPriority(&c->light, hipStreamNonBlocking, greatest);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_final, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_xdone, hipEventDisableTiming);
    if (e != hipSuccess) return bail(FAIL(ZH_EHIP, "shard context streams / events: %s", hipGetErrorString(e)));
    *out = c;
    return ZH_OK;
}

extern "C" void zh_shard_ctx_destroy(zh_shard_ctx *c) {
    if (!c) return;
    hipSetDevice(c->g->device);
    if (c->xs) hipStreamSynchronize(c->xs);
    if (c->light) hipStreamSynchronize(c->light);
    zh_search_ctx_destroy(c->sc);
    if (c->gathered) hipFree(c->gathered);
    if (c->merged) hipFree(c->merged);
    if (c->ev_final) hipEventDestroy(c->ev_final);
    if (c->ev_xdone) hipEventDestroy(c->ev_xdone);
    if (c->xs) hipStreamDestroy(c->xs);
    if (c->light) hipStreamDestroy(c->light);
    delete c;
}

extern "C" void *zh_shard_ctx_stream(const zh_shard_ctx *c) { return c ? (void *)c->xs : nullptr; }
extern "C" const uint64_t *zh_shard_ctx_local_result(const zh_shard_ctx *c) {
    return (c && c->gathered) ? c->gathered + (size_t)c->g->rank * c->W : nullptr;
}

static int shard_begin(zh_shard_ctx *c, const float *const *d_q, size_t nwin, size_t b, size_t k, int metric, int mode) {
    if (c->state == 1) return FAIL(ZH_ESTATE, "zh_shard_search_begin: the context already has a batch begun");
    if (k == 0 || k > ZH_MAX_TOPK) return FAIL(ZH_ELIMIT, "top_k must be in 1..%u", ZH_MAX_TOPK);
    if (b == 0) return FAIL(ZH_EINVAL, "zh_shard_search_begin: empty batch");
    int rc = set_dev(c->g);
    if (rc) return rc;
    if (c->state == 2 && (rc = zh_shard_search_wait(c))) return rc;
    if ((rc = zh_search_begin_window(c->sc, d_q, nwin, b, k, metric, mode, c->light))) return rc;
    c->B = nwin * b; c->k = k; c->nwin = nwin; c->bwin = b;
    c->state = 1;
    return ZH_OK;
}
extern "C" int zh_shard_search_begin(zh_shard_ctx *c, const float *d_q, size_t b, size_t k, int metric, int mode) {
    if (!c || !d_q) return FAIL(ZH_EINVAL, "zh_shard_search_begin: null argument");
    return shard_begin(c, &d_q, 1, b, k, metric, mode);
}
extern "C" int zh_shard_search_begin_window(zh_shard_ctx *c, const float *const *d_q, size_t n_batches, size_t b, size_t k,
                                            int metric, int mode) {
    if (!c || !d_q || n_batches == 0 || n_batches > ZH_MAX_WINDOW) return FAIL(ZH_EINVAL, "zh_shard_search_begin_window: bad argument");
    return shard_begin(c, d_q, n_batches, b, k, metric, mode);
}

// local results of the whole window land in this rank's slot of the gather buffer ([ids B*k | keys B*k | counts B], B =
// every query of the window), ONE all-gather and ONE merge per window; a window's merged results are handed out per batch
static int shard_finish(zh_shard_ctx *c, uint64_t *const *out_ids, uint64_t *const *out_keys, uint32_t *const *out_counts) {
    if (c->state != 1) return FAIL(ZH_ESTATE, "zh_shard_search_finish without zh_shard_search_begin");
    zh_shard_group *g = c->g;
    int rc = set_dev(g);
    if (rc) return rc;
    const size_t B = c->B, k = c->k, nwin = c->nwin, b = c->bwin;
    const size_t W = zh_packed_result_words(B, k), need = W * g->n_ranks;
    if (need > c->cap_words) {  // another batch shape: the previous exchange must have read the old buffer
        if (c->xused) HIPCHK(hipEventSynchronize(c->ev_xdone));
        if (c->gathered) hipFree(c->gathered);
        c->gathered = nullptr; c->cap_words = 0;
        HIPCHK(hipMalloc((void **)&c->gathered, need * 8));
        c->cap_words = need;
    } else if (c->xused) {
        // the buffer is free again once this context's previous exchange has read it (long done: a batch ago)
        HIPCHK(hipStreamWaitEvent(c->light, c->ev_xdone, 0));
    }
    if (nwin > 1 && W > c->cap_merged) {
        if (c->xused) HIPCHK(hipEventSynchronize(c->ev_xdone));
        if (c->merged) hipFree(c->merged);
        c->merged = nullptr; c->cap_merged = 0;
        HIPCHK(hipMalloc((void **)&c->merged, W * 8));
        c->cap_merged = W;
    }
    c->W = W;
    uint64_t *mine = c->gathered + (size_t)g->rank * W;
    uint64_t *l_ids[ZH_MAX_WINDOW], *l_keys[ZH_MAX_WINDOW];
    uint32_t *l_counts[ZH_MAX_WINDOW];
    for (size_t j = 0; j < nwin; j++) {
        l_ids[j] = mine + j * b * k;
        l_keys[j] = mine + B * k + j * b * k;
        l_counts[j] = reinterpret_cast<uint32_t *>(mine + 2 * B * k) + j * b;
    }
    c->state = 0;
    if ((rc = zh_search_finish_window(c->sc, l_ids, l_keys, l_counts, zh_index_sweep_stream(g->ix)))) return rc;
    HIPCHK(hipEventRecord(c->ev_final, c->light));
    {
        std::lock_guard<std::mutex> lk(g->mu);
        HIPCHK(hipStreamWaitEvent(c->xs, c->ev_final, 0));
        NCCLCHK(ncclAllGather(mine, c->gathered, W, ncclUint64, g->comm, c->xs));  // in place: send == recv + rank * W
        uint64_t *m_ids = nwin > 1 ? c->merged : out_ids[0], *m_keys = nwin > 1 ? c->merged + B * k : out_keys[0];
        uint32_t *m_counts = nwin > 1 ? reinterpret_cast<uint32_t *>(c->merged + 2 * B * k) : out_counts[0];
        if ((rc = zh_merge_topk_packed_device(g->device, g->n_ranks, B, k, c->gathered, m_ids, m_keys, m_counts, c->xs)))
            return rc;
        for (size_t j = 0; j < nwin && nwin > 1; j++) {
            HIPCHK(hipMemcpyAsync(out_ids[j], m_ids + j * b * k, b * k * 8, hipMemcpyDeviceToDevice, c->xs));
            HIPCHK(hipMemcpyAsync(out_keys[j], m_keys + j * b * k, b * k * 8, hipMemcpyDeviceToDevice, c->xs));
            HIPCHK(hipMemcpyAsync(out_counts[j], m_counts + j * b, b * 4, hipMemcpyDeviceToDevice, c->xs));
        }
        HIPCHK(hipEventRecord(c->ev_xdone, c->xs));
    }
    c->xused = true;
    c->state = 2;
    return ZH_OK;
}
extern "C" int zh_shard_search_finish(zh_shard_ctx *c, uint64_t *d_out_ids, uint64_t *d_out_keys, uint32_t *d_out_counts) {
    if (!c || !d_out_ids || !d_out_keys || !d_out_counts) return FAIL(ZH_EINVAL, "zh_shard_search_finish: null argument");
    if (c->nwin != 1) return FAIL(ZH_ESTATE, "zh_shard_search_finish: the context holds a window; use zh_shard_search_finish_window");
    return shard_finish(c, &d_out_ids, &d_out_keys, &d_out_counts);
}
extern "C" int zh_shard_search_finish_window(zh_shard_ctx *c, uint64_t *const *d_out_ids, uint64_t *const *d_out_keys,
                                             uint32_t *const *d_out_counts) {
    if (!c || !d_out_ids || !d_out_keys || !d_out_counts) return FAIL(ZH_EINVAL, "zh_shard_search_finish_window: null argument");
    for (size_t j = 0; j < c->nwin; j++)
        if (!d_out_ids[j] || !d_out_keys[j] || !d_out_counts[j]) return FAIL(ZH_EINVAL, "zh_shard_search_finish_window: null output");
    return shard_finish(c, d_out_ids, d_out_keys, d_out_counts);
}

extern "C" int zh_shard_search_wait(zh_shard_ctx *c) {
    if (!c) return FAIL(ZH_EINVAL, "zh_shard_search_wait: null context");
    if (c->state == 1) return FAIL(ZH_ESTATE, "zh_shard_search_wait: the batch was begun but not finished");
    if (c->state != 2) return ZH_OK;
    int rc = set_dev(c->g);
    if (rc) return rc;
    c->state = 0;
    if ((rc = zh_search_wait(c->sc))) return rc;
    HIPCHK(hipEventSynchronize(c->ev_xdone));
    ncclResult_t async = ncclSuccess;
    if (ncclCommGetAsyncError(c->g->comm, &async) == ncclSuccess && async != ncclSuccess)
        return FAIL(ZH_EHIP, "RCCL asynchronous error: %s", ncclGetErrorString(async));
    return ZH_OK;
}

extern "C" int zh_shard_search_batch_device(zh_shard_group *g, const float *d_q, size_t b, size_t k, int metric, int mode,
                                            uint64_t *d_out_ids, uint64_t *d_out_keys, uint32_t *d_out_counts) {
    if (!g) return FAIL(ZH_EINVAL, "zh_shard_search_batch_device: null group");
    if (b == 0) return ZH_OK;
    int rc = zh_shard_search_begin(g->dctx, d_q, b, k, metric, mode);
    if (rc) return rc;
    if ((rc = zh_shard_search_finish(g->dctx, d_out_ids, d_out_keys, d_out_counts))) return rc;
    return zh_shard_search_wait(g->dctx);
}

extern "C" int zh_shard_search_batch(zh_shard_group *g, const float *q, size_t b, size_t k, int metric, int mode,
                                     uint64_t *out_ids, uint64_t *out_keys, uint32_t *out_counts) {
    if (!g || (b && (!q || !out_ids || !out_keys || !out_counts))) return FAIL(ZH_EINVAL, "zh_shard_search_batch: null argument");
    if (b == 0) return ZH_OK;
    int rc = set_dev(g);
    if (rc) return rc;
    const size_t d = zh_index_dim(g->ix), qbytes = b * d * 4, obytes = b * k * 16 + b * 4;
    if (qbytes > g->capQ) {
        if (g->dQ) hipFree(g->dQ);
        g->dQ = nullptr; g->capQ = 0;
        HIPCHK(hipMalloc(&g->dQ, qbytes));
        g->capQ = qbytes;
    }
    if (obytes > g->capOut) {
        if (g->dOut) hipFree(g->dOut);
        g->dOut = nullptr; g->capOut = 0;
        HIPCHK(hipMalloc(&g->dOut, obytes));
        g->capOut = obytes;
    }
    uint64_t *dIds = (uint64_t *)g->dOut, *dKeys = dIds + b * k;
    uint32_t *dCounts = (uint32_t *)(dKeys + b * k);
    hipStream_t xs = g->dctx->xs;
    HIPCHK(hipMemcpy(g->dQ, q, qbytes, hipMemcpyHostToDevice));
    if ((rc = zh_shard_search_begin(g->dctx, (const float *)g->dQ, b, k, metric, mode))) return rc;
    if ((rc = zh_shard_search_finish(g->dctx, dIds, dKeys, dCounts))) return rc;
    HIPCHK(hipMemcpyAsync(out_ids, dIds, b * k * 8, hipMemcpyDeviceToHost, xs));
    HIPCHK(hipMemcpyAsync(out_keys, dKeys, b * k * 8, hipMemcpyDeviceToHost, xs));
    HIPCHK(hipMemcpyAsync(out_counts, dCounts, b * 4, hipMemcpyDeviceToHost, xs));
    if ((rc = zh_shard_search_wait(g->dctx))) return rc;
    HIPCHK(hipStreamSynchronize(xs));
    return ZH_OK;
}
