// zh_shard.hip -- the multi-GPU exchange behind the C ABI: a shard group is this rank's zh_index plus an RCCL
// communicator; one search over the sharded index = local search -> ONE in-place ncclAllGather of the packed
// [ids | keys | counts] result -> merge kernel on every rank (what replaces the query loop of
// /root/reference/src/database/core.rs:299-303 when the stored rows are sharded, README.md:31).
// The local search goes through the public entry points of zebra_hip.h (zh_search_begin / finish / wait), so the single-GPU
// pipeline is exactly the one the parity tests exercise.  Host code; no kernels here.
//
// FAILING TOGETHER.  A collective cannot swallow a per-rank error the way core.rs:303 swallows a per-query one: a rank that
// returned before the all-gather would leave its peers inside it for ever.  So a rank whose local search fails (ZH_ELIMIT: its
// shard passed 2^28 visits; ZH_ENOMEM: its scratch) STILL joins the exchange, with an empty slot (every count 0) and its code
// in a STATUS WORD that travels with the packed result (slot = zh_packed_result_words(b, k) + 1 words,
// zh_shard_exchange_words).  After the all-gather every rank holds every status: zh_shard_search_wait returns the rank's own
// code, or ZH_EPEER when only other ranks failed -- the same verdict everywhere, nobody hangs, the communicator stays usable.
// The blocking entry points use the same words to split an over-long batch IDENTICALLY on every rank: if every failure of a
// chunk is ZH_ELIMIT, all ranks halve the chunk and repeat it (they all see the same words); the first chunk size comes from the
// largest visits-per-query any rank has reported so far (the word's upper half).  What a status word cannot cover -- a rank
// that dies, or cannot even allocate its gather buffer -- is bounded by a timeout in zh_shard_search_wait
// (ZH_SHARD_TIMEOUT_MS, default 300000): the waiting rank aborts its communicator (ncclCommAbort), returns ZH_EPEER, and the
// group is dead (every later call fails fast, zh_shard_group_destroy aborts instead of ncclCommDestroy, which would block).
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
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
    std::atomic<bool> dead{false};        // a collective may be outstanding for ever (timeout, RCCL async error, a rank that could not join)
    std::atomic<uint32_t> peer_vpq{0};    // largest leaf-visits-per-query any rank has reported (status words): sizes the blocking calls' chunks
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
    // status words (pinned host memory): [0, n_ranks) every rank's word as gathered, [n_ranks] this rank's outgoing word
    uint64_t *h_status = nullptr;
    int begin_rc = 0;        // a failed local begin: finish still joins the exchange, with this code
    int verdict = 0;         // of the last waited batch: ZH_OK, this rank's own code, or ZH_EPEER
    bool all_elimit = false; // ... and every failing rank said ZH_ELIMIT (the blocking calls retry with a smaller chunk)
};

// ---- status words: pure host arithmetic (callable without a GPU; tests/test_sharding_gloo.py carries them over gloo) ----
extern "C" size_t zh_shard_exchange_words(size_t b, size_t k) { return zh_packed_result_words(b, k) + 1; }
extern "C" uint64_t zh_shard_status_word(int code, uint32_t visits_per_query) {
    return (uint64_t)(uint32_t)(int32_t)code | ((uint64_t)visits_per_query << 32);
}
extern "C" int zh_shard_verdict(const uint64_t *status_words, uint32_t n_ranks, uint32_t rank, uint32_t *out_first_failed_rank,
                                uint32_t *out_max_visits_per_query, int *out_all_elimit) {
    if (!status_words || rank >= n_ranks) return FAIL(ZH_EINVAL, "zh_shard_verdict: bad argument");
    int own = (int)(int32_t)(uint32_t)status_words[rank], first_code = 0;
    uint32_t first = n_ranks, vmax = 0;
    bool all_el = true;
    for (uint32_t r = 0; r < n_ranks; r++) {
        const int code = (int)(int32_t)(uint32_t)status_words[r];
        vmax = std::max(vmax, (uint32_t)(status_words[r] >> 32));
        if (code != ZH_OK) {
            if (first == n_ranks) { first = r; first_code = code; }
            if (code != ZH_ELIMIT) all_el = false;
        }
    }
    if (out_first_failed_rank) *out_first_failed_rank = first;
    if (out_max_visits_per_query) *out_max_visits_per_query = vmax;
    if (out_all_elimit) *out_all_elimit = (first != n_ranks && all_el) ? 1 : 0;
    if (first == n_ranks) return ZH_OK;
    if (own != ZH_OK) return FAIL(own, "this rank's (%u) local search of the sharded batch failed with status %d; every rank of the group was told", rank, own);
    return FAIL(ZH_EPEER, "rank %u of the shard group failed its local search with status %d: the batch's results are not valid on any rank", first, first_code);
}

// test hook (like ZH_WALK_LOG_FIXED): ZH_SHARD_INJECT="code[,count[,skip]]" makes `count` (default 1) local searches of this
// process fail with `code` after letting `skip` (default 0) pass; read on every call, counted from the value's first appearance
static int injected_failure() {
    static std::string last;
    static int seen = 0;
    const char *e = getenv("ZH_SHARD_INJECT");
    if (!e || !*e) { last.clear(); seen = 0; return 0; }
    if (last != e) { last = e; seen = 0; }
    int code = 0, count = 1, skip = 0;
    if (sscanf(e, "%d,%d,%d", &code, &count, &skip) < 1) return 0;
    const int i = seen++;
    return (i >= skip && i < skip + count) ? code : 0;
}

static void kill_group(zh_shard_group *g) {  // a collective of this communicator may never complete: abort it (frees the queued work)
    bool was = g->dead.exchange(true);
    if (!was && g->comm) { ncclCommAbort(g->comm); g->comm = nullptr; }
}

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
    if (g->comm) ncclCommDestroy(g->comm);  // (a dead group's communicator was aborted -- and nulled -- by kill_group: destroy would block)
    delete g;
}

extern "C" uint32_t zh_shard_group_ranks(const zh_shard_group *g) {
    if (!g || !g->comm) return 0;
    int n = 0;
    return ncclCommCount(g->comm, &n) == ncclSuccess ? (uint32_t)n : 0;
}
extern "C" uint32_t zh_shard_group_rank(const zh_shard_group *g) {
    if (!g || !g->comm) return g ? g->rank : 0;
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
    zh_search_ctx_stream_ordered(c->sc);  // the exchange is enqueued behind the local search, before any wait
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
    hipError_t e = hipStreamCreateWithPriority(&c->light, hipStreamNonBlocking, greatest);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_final, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_xdone, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_status, ((size_t)g->n_ranks + 1) * 8, hipHostMallocDefault);
    if (e != hipSuccess) return bail(FAIL(ZH_EHIP, "shard context streams / events: %s", hipGetErrorString(e)));
    memset(c->h_status, 0, ((size_t)g->n_ranks + 1) * 8);
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
    if (c->h_status) hipHostFree(c->h_status);
    if (c->ev_final) hipEventDestroy(c->ev_final);
    if (c->ev_xdone) hipEventDestroy(c->ev_xdone);
    if (c->xs) hipStreamDestroy(c->xs);
    if (c->light) hipStreamDestroy(c->light);
    delete c;
}

extern "C" void *zh_shard_ctx_stream(const zh_shard_ctx *c) { return c ? (void *)c->xs : nullptr; }
extern "C" const uint64_t *zh_shard_ctx_local_result(const zh_shard_ctx *c) {
    return (c && c->gathered) ? c->gathered + (size_t)c->g->rank * (c->W + 1) : nullptr;
}

static int shard_begin(zh_shard_ctx *c, const float *const *d_q, size_t nwin, size_t b, size_t k, int metric, int mode) {
    if (c->state == 1) return FAIL(ZH_ESTATE, "zh_shard_search_begin: the context already has a batch begun");
    if (k == 0 || k > ZH_MAX_TOPK) return FAIL(ZH_ELIMIT, "top_k must be in 1..%u", ZH_MAX_TOPK);
    if (b == 0) return FAIL(ZH_EINVAL, "zh_shard_search_begin: empty batch");
    if (c->g->dead.load()) return FAIL(ZH_EPEER, "the shard group is dead (an earlier exchange timed out or failed): destroy it");
    int rc = set_dev(c->g);
    if (rc) return rc;
    if (c->state == 2) {
        // Retire the batch the context still holds.  Its verdict was the caller's to collect; one that was not OK is reported HERE and
        // the new batch is not begun: the merged outputs of the retired window are invalid on every rank, and every rank sees a
        // failed verdict for it (its own code, or ZH_EPEER), so every rank's begin returns alike and the collective sequence stays aligned.
        const int v = zh_shard_search_wait(c);
        if (v != ZH_OK) {
            const std::string why = zh_last_error();
            return FAIL(v, "zh_shard_search_begin retired the batch this context still held, and it had FAILED (its outputs are invalid on "
                           "every rank; nothing new was begun): %s", why.c_str());
        }
    }
    if (c->g->dead.load()) return FAIL(ZH_EPEER, "the shard group is dead (an earlier exchange timed out or failed): destroy it");
    // From here on the call is part of a collective sequence: whatever happens locally, this rank joins the exchange in
    // finish.  A failed local begin is remembered and travels in the status word.
    c->B = nwin * b; c->k = k; c->nwin = nwin; c->bwin = b;
    c->begin_rc = zh_search_begin_window(c->sc, d_q, nwin, b, k, metric, mode, c->light);
    c->state = 1;
    return c->begin_rc;
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

// local results of the whole window land in this rank's slot of the gather buffer ([ids B*k | keys B*k | counts B | status],
// B = every query of the window), ONE all-gather and ONE merge per window; a window's merged results are handed out per batch.
// A local failure still joins: an empty slot (counts 0) and the code in the status word; the call then returns that code
// (state 2: zh_shard_search_wait completes the exchange and reports the group's verdict).
static int shard_finish(zh_shard_ctx *c, uint64_t *const *out_ids, uint64_t *const *out_keys, uint32_t *const *out_counts) {
    if (c->state != 1) return FAIL(ZH_ESTATE, "zh_shard_search_finish without zh_shard_search_begin");
    zh_shard_group *g = c->g;
    if (g->dead.load()) { c->state = 0; return FAIL(ZH_EPEER, "the shard group is dead (an earlier exchange timed out or failed): destroy it"); }
    int rc = set_dev(g);
    if (rc) {  // this rank cannot join the exchange its peers are about to enter: fail together, fast (not after ZH_SHARD_TIMEOUT_MS)
        c->state = 0;
        kill_group(g);
        return rc;
    }
    const size_t B = c->B, k = c->k, nwin = c->nwin, b = c->bwin;
    const size_t W = zh_packed_result_words(B, k), SW = W + 1, need = SW * g->n_ranks;
    // a rank that cannot even hold the gather buffer cannot join: it kills the group (peers time out in wait -> ZH_EPEER)
    auto cannot_join = [&](int code, const char *what, hipError_t e) {
        c->state = 0;
        kill_group(g);
        return FAIL(code, "%s: %s -- this rank cannot join the exchange; the shard group is dead", what, hipGetErrorString(e));
    };
    if (need > c->cap_words) {  // another batch shape: the previous exchange must have read the old buffer
        if (c->xused) { hipError_t e = hipEventSynchronize(c->ev_xdone); if (e != hipSuccess) return cannot_join(ZH_EHIP, "hipEventSynchronize", e); }
        if (c->gathered) hipFree(c->gathered);
        c->gathered = nullptr; c->cap_words = 0;
        hipError_t e = hipMalloc((void **)&c->gathered, need * 8);
        if (e != hipSuccess) return cannot_join(ZH_ENOMEM, "hipMalloc of the gather buffer", e);
        c->cap_words = need;
    } else if (c->xused) {
        // the buffer is free again once this context's previous exchange has read it (long done: a batch ago)
        hipError_t e = hipStreamWaitEvent(c->light, c->ev_xdone, 0);
        if (e != hipSuccess) return cannot_join(ZH_EHIP, "hipStreamWaitEvent", e);
    }
    if (nwin > 1 && W > c->cap_merged) {
        if (c->xused) { hipError_t e = hipEventSynchronize(c->ev_xdone); if (e != hipSuccess) return cannot_join(ZH_EHIP, "hipEventSynchronize", e); }
        if (c->merged) hipFree(c->merged);
        c->merged = nullptr; c->cap_merged = 0;
        hipError_t e = hipMalloc((void **)&c->merged, W * 8);
        if (e != hipSuccess) return cannot_join(ZH_ENOMEM, "hipMalloc of the merge buffer", e);
        c->cap_merged = W;
    }
    c->W = W;
    uint64_t *mine = c->gathered + (size_t)g->rank * SW;
    uint64_t *l_ids[ZH_MAX_WINDOW], *l_keys[ZH_MAX_WINDOW];
    uint32_t *l_counts[ZH_MAX_WINDOW];
    for (size_t j = 0; j < nwin; j++) {
        l_ids[j] = mine + j * b * k;
        l_keys[j] = mine + B * k + j * b * k;
        l_counts[j] = reinterpret_cast<uint32_t *>(mine + 2 * B * k) + j * b;
    }
    c->state = 0;
    // ---- the local search; its outcome is this rank's status word ----
    int local = c->begin_rc;
    std::string local_msg = local ? zh_last_error() : "";
    if (!local && (local = injected_failure())) {
        // (test hook) the local search is abandoned exactly as a real ZH_ELIMIT abandons it: begun, never finished
        local_msg = "injected failure (ZH_SHARD_INJECT)";
        zh_search_ctx_abandon(c->sc);
    } else if (!local && (local = zh_search_finish_window(c->sc, l_ids, l_keys, l_counts, zh_index_sweep_stream(g->ix))))
        local_msg = zh_last_error();
    const double vpp = zh_index_visits_per_pair(g->ix) * zh_index_num_trees(g->ix);
    c->h_status[g->n_ranks] = zh_shard_status_word(local, (uint32_t)std::min(vpp + 0.999, 4294967295.0));
    hipError_t e = hipSuccess;
    if (local) e = hipMemsetAsync(mine, 0, W * 8, c->light);  // an empty slot: every count 0, nothing for the merge to read
    if (e == hipSuccess) e = hipMemcpyAsync(mine + W, c->h_status + g->n_ranks, 8, hipMemcpyHostToDevice, c->light);
    if (e == hipSuccess) e = hipEventRecord(c->ev_final, c->light);
    if (e != hipSuccess) return cannot_join(ZH_EHIP, "status word / event", e);
    {
        std::lock_guard<std::mutex> lk(g->mu);
        e = hipStreamWaitEvent(c->xs, c->ev_final, 0);
        if (e != hipSuccess) return cannot_join(ZH_EHIP, "hipStreamWaitEvent", e);
        ncclResult_t r = ncclAllGather(mine, c->gathered, SW, ncclUint64, g->comm, c->xs);  // in place: send == recv + rank * SW
        if (r != ncclSuccess) {
            kill_group(g);
            return FAIL(ZH_EHIP, "ncclAllGather: %s; the shard group is dead", ncclGetErrorString(r));
        }
        // from here on the collective is enqueued: failures below only affect this rank's copy of the results
        uint64_t *m_ids = nwin > 1 ? c->merged : out_ids[0], *m_keys = nwin > 1 ? c->merged + B * k : out_keys[0];
        uint32_t *m_counts = nwin > 1 ? reinterpret_cast<uint32_t *>(c->merged + 2 * B * k) : out_counts[0];
        e = zh_launch_merge(g->n_ranks, (uint32_t)B, (uint32_t)k, c->gathered, c->gathered + B * k,
                            reinterpret_cast<const uint32_t *>(c->gathered + 2 * B * k), m_ids, m_keys, m_counts, SW, 2 * SW, c->xs);
        for (size_t j = 0; j < nwin && nwin > 1 && e == hipSuccess; j++) {
            e = hipMemcpyAsync(out_ids[j], m_ids + j * b * k, b * k * 8, hipMemcpyDeviceToDevice, c->xs);
            if (e == hipSuccess) e = hipMemcpyAsync(out_keys[j], m_keys + j * b * k, b * k * 8, hipMemcpyDeviceToDevice, c->xs);
            if (e == hipSuccess) e = hipMemcpyAsync(out_counts[j], m_counts + j * b, b * 4, hipMemcpyDeviceToDevice, c->xs);
        }
        // every rank's status word to the host (n_ranks words, SW apart)
        if (e == hipSuccess) e = hipMemcpy2DAsync(c->h_status, 8, c->gathered + W, SW * 8, 8, g->n_ranks, hipMemcpyDeviceToHost, c->xs);
        hipError_t e2 = hipEventRecord(c->ev_xdone, c->xs);
        if (e == hipSuccess) e = e2;
    }
    c->xused = true;
    c->state = 2;
    if (e != hipSuccess) return FAIL(ZH_EHIP, "merge / result copies: %s", hipGetErrorString(e));
    if (local) return FAIL(local, "%s (this rank still joined the exchange: every rank of the group is told)", local_msg.c_str());
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

// Blocks until the exchange has completed -- or has not within ZH_SHARD_TIMEOUT_MS (a peer died or could not join): then the
// communicator is aborted and the group is dead.  Returns the group's verdict on the batch: ZH_OK, this rank's own failure,
// or ZH_EPEER (the same outcome class on every rank).
extern "C" int zh_shard_search_wait(zh_shard_ctx *c) {
    if (!c) return FAIL(ZH_EINVAL, "zh_shard_search_wait: null context");
    if (c->state == 1) return FAIL(ZH_ESTATE, "zh_shard_search_wait: the batch was begun but not finished");
    if (c->state != 2) return ZH_OK;
    zh_shard_group *g = c->g;
    int rc = set_dev(g);
    if (rc) return rc;
    c->state = 0;
    c->verdict = ZH_OK; c->all_elimit = false;
    const int rc_local = zh_search_wait(c->sc);  // (idle when the local search had failed)
    static const long timeout_ms = [] { const char *e = getenv("ZH_SHARD_TIMEOUT_MS"); return e ? atol(e) : 300000L; }();
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0;; spin++) {
        const hipError_t q = hipEventQuery(c->ev_xdone);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) { kill_group(g); return c->verdict = FAIL(ZH_EHIP, "exchange: %s; the shard group is dead", hipGetErrorString(q)); }
        if (g->dead.load()) return c->verdict = FAIL(ZH_EPEER, "the shard group died while this batch was in flight");
        if ((spin & 63) == 63) {
            ncclResult_t async = ncclSuccess;
            if (ncclCommGetAsyncError(g->comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
                kill_group(g);
                return c->verdict = FAIL(ZH_EPEER, "RCCL asynchronous error: %s; the shard group is dead", ncclGetErrorString(async));
            }
            if (timeout_ms > 0 && std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms) {
                kill_group(g);
                return c->verdict = FAIL(ZH_EPEER, "the exchange did not complete within %ld ms (ZH_SHARD_TIMEOUT_MS): a rank died or could not join; "
                                                   "the communicator was aborted, the shard group is dead", timeout_ms);
            }
        }
        if (spin < 2000) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    uint32_t vmax = 0;
    int all_el = 0;
    const int verdict = zh_shard_verdict(c->h_status, g->n_ranks, g->rank, nullptr, &vmax, &all_el);
    uint32_t cur = g->peer_vpq.load();
    while (vmax > cur && !g->peer_vpq.compare_exchange_weak(cur, vmax)) {}
    c->all_elimit = all_el != 0;
    if (verdict) return c->verdict = verdict;
    if (rc_local) {
        // the exchange said OK everywhere but this rank's own local wait failed AFTER its status word had left (a HIP error): its
        // peers believe the batch and would go on to the next collective without it -- the group cannot continue
        const std::string why = zh_last_error();
        kill_group(g);
        return c->verdict = FAIL(rc_local, "local search failed after the exchange (%s); the shard group is dead", why.c_str());
    }
    return ZH_OK;
}

// The blocking calls take any batch: the chunks are the SAME on every rank -- sized from the largest visits-per-query any rank
// has reported (the status words of earlier exchanges), and when a chunk fails with ZH_ELIMIT on any rank (and with nothing
// else anywhere) every rank halves it and repeats it: all ranks read the same status words, so all take the same decision.
static int search_chunks(zh_shard_group *g, const float *d_q, size_t b, size_t k, int metric, int mode, uint64_t *d_out_ids,
                         uint64_t *d_out_keys, uint32_t *d_out_counts) {
    const size_t T = std::max<uint32_t>(zh_index_num_trees(g->ix), 1), d = zh_index_dim(g->ix);
    zh_shard_ctx *c = g->dctx;
    size_t chunk = std::min<size_t>(b, ((1ull << 26) - 1) / T);
    const uint32_t vpq = g->peer_vpq.load();
    if (vpq) chunk = std::min<size_t>(chunk, std::max<size_t>(1, (size_t)((1ull << 27) / vpq)));  // half the 2^28-visit cap: headroom
    if (chunk == 0) chunk = 1;
    for (size_t b0 = 0; b0 < b;) {
        const size_t nb = std::min(chunk, b - b0);
        int rc = zh_shard_search_begin(c, d_q + b0 * d, nb, k, metric, mode);
        if (c->state != 1) return rc;  // refused before anything collective happened (bad argument, dead group)
        rc = zh_shard_search_finish(c, d_out_ids + b0 * k, d_out_keys + b0 * k, d_out_counts + b0);
        if (c->state != 2) return rc;  // this rank could not join: the group is dead
        rc = zh_shard_search_wait(c);
        if (rc && c->all_elimit && nb > 1) { chunk = (nb + 1) / 2; continue; }  // every rank repeats [b0, b0 + chunk)
        if (rc) return rc;
        b0 += nb;
    }
    return ZH_OK;
}

extern "C" int zh_shard_search_batch_device(zh_shard_group *g, const float *d_q, size_t b, size_t k, int metric, int mode,
                                            uint64_t *d_out_ids, uint64_t *d_out_keys, uint32_t *d_out_counts) {
    if (!g) return FAIL(ZH_EINVAL, "zh_shard_search_batch_device: null group");
    if (b == 0) return ZH_OK;
    if (!d_q || !d_out_ids || !d_out_keys || !d_out_counts) return FAIL(ZH_EINVAL, "zh_shard_search_batch_device: null argument");
    return search_chunks(g, d_q, b, k, metric, mode, d_out_ids, d_out_keys, d_out_counts);
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
    HIPCHK(hipMemcpy(g->dQ, q, qbytes, hipMemcpyHostToDevice));
    if ((rc = search_chunks(g, (const float *)g->dQ, b, k, metric, mode, dIds, dKeys, dCounts))) return rc;
    // every chunk's exchange has completed (search_chunks waits for each): plain copies
    HIPCHK(hipMemcpy(out_ids, dIds, b * k * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out_keys, dKeys, b * k * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out_counts, dCounts, b * 4, hipMemcpyDeviceToHost));
    return ZH_OK;
}
