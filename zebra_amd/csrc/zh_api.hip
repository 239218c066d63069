// zh_api.hip -- the C ABI of include/zebra_hip.h: index lifetime, device memory, forest build
// orchestration and the per-batch search pipeline.  Host code only; kernels live in
// zh_search.hip / zh_build.hip.  There is no CPU compute path in this file: every entry point
// that produces distances, signs or neighbours launches gfx950 kernels.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <deque>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "zh_internal.h"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail(e_ == hipErrorOutOfMemory ? ZH_ENOMEM : ZH_EHIP, "%s: %s (%s:%d)", #expr,          \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                        \
    } while (0)

// the same thread-local message for the host-only translation units
int zh_set_error(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

extern "C" const char *zh_last_error(void) { return g_err.c_str(); }
extern "C" const char *zh_version(void) { return "zebra-hip 0.1 (gfx950)"; }

// ------------------------------------------------------------------------------------------------
// growable device buffer
// ------------------------------------------------------------------------------------------------
// Device memory for the library's buffers.  Default: hipMalloc.  ZH_VMM=1 (read once; an experiment for the "order effect", DESIGN.md s9: an index
// whose buffers are allocated after another library has churned device memory scans 25-40 % slower): allocations of 2 MiB and more are made through
// the virtual-memory API instead -- an address range reserved, physical memory created in chunks of ZH_VMM_CHUNK_MB (default 256) and mapped -- so
// that what backs a buffer is a handful of large physical allocations whatever state hipMalloc's pool is in.
struct VmmRec { size_t size; std::vector<hipMemGenericAllocationHandle_t> handles; std::vector<size_t> sizes; };
static std::mutex g_vmm_mu;
static std::unordered_map<void *, VmmRec> g_vmm;
static int vmm_mode() {
    static const int m = [] { const char *e = getenv("ZH_VMM"); return e ? atoi(e) : 0; }();
    return m;
}
// ---- the block cache (round 6; the "order effect"), OPT-IN: ZH_POOL=1.  How a buffer is MAPPED decides how fast the matrix-core scan
// runs over it: the same cfg3 launch takes 3.2 ms on hipMalloc'd buffers of a fresh process and on ranges of the virtual-memory API aligned to 2 MiB
// and backed by 256-MiB chunks, 3.7 ms on ranges aligned to 4 KiB, 3.4-4.8 on ranges made of 2-MiB chunks (profiles/r06_order_effect.txt; ZH_VMM above
// is that experiment) -- and 3.6-3.8 on the hipMalloc'd buffers of the second / third index a process creates.  With the cache, large blocks are not
// handed back to the driver when an index or a context lets go of them: they wait here, keyed by size, and the next buffer of a fitting size takes
// one (what a framework's caching allocator does); hipMalloc is asked only for what no cached block serves, and when it fails the cache is emptied
// and it is asked again.  Measured: the late index's launch 3.77 -> 3.59 ms (cfg4 shard), 0.99 -> 0.93 (cfg2) -- a twentieth, not the effect: buffers
// that are NEW late in a process are slow whether or not anything was freed before them.  Off by default (held memory is invisible to other
// libraries' allocators); zh_trim_device_memory() empties it; blocks below 32 MiB are not kept.
struct PoolBlock { void *p; size_t cap; int dev; };
static std::mutex g_pool_mu;
static std::vector<PoolBlock> g_pool;
static std::unordered_map<void *, size_t> g_pool_caps;  // every live block of 32 MiB and more this library allocated: its capacity
static size_t g_pool_bytes = 0;
static bool pool_on() {
    static const bool on = [] { const char *e = getenv("ZH_POOL"); return e && e[0] == '1'; }();
    return on;
}
static const size_t POOL_MIN = size_t(32) << 20;
static void zh_dev_free_raw(void *p);
static size_t pool_cached_bytes(int dev) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    size_t n = 0;
    for (const PoolBlock &b : g_pool) if (b.dev == dev) n += b.cap;
    return n;
}
static void pool_trim() {
    std::vector<PoolBlock> gone;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        gone.swap(g_pool);
        g_pool_bytes = 0;
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (const PoolBlock &b : gone) {
        { std::lock_guard<std::mutex> lk(g_pool_mu); g_pool_caps.erase(b.p); }
        (void)hipSetDevice(b.dev);
        zh_dev_free_raw(b.p);
    }
    (void)hipSetDevice(cur);
}
// free device memory as the library's own headroom rules should see it: what the driver reports plus what the block cache would give back
static hipError_t zh_mem_info(size_t *mem_free, size_t *mem_total) {
    hipError_t e = hipMemGetInfo(mem_free, mem_total);
    if (e != hipSuccess) return e;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) *mem_free += pool_cached_bytes(dev);
    return hipSuccess;
}
static hipError_t zh_dev_alloc_raw(void **out, size_t bytes);
static hipError_t zh_dev_alloc(void **out, size_t bytes) {
    if (pool_on() && bytes >= POOL_MIN) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            size_t best = (size_t)-1;
            for (size_t i = 0; i < g_pool.size(); i++)  // best fit, at most half as much again as asked for
                if (g_pool[i].dev == dev && g_pool[i].cap >= bytes && g_pool[i].cap <= bytes + bytes / 2 && (best == (size_t)-1 || g_pool[i].cap < g_pool[best].cap)) best = i;
            if (best != (size_t)-1) {
                *out = g_pool[best].p;
                g_pool_bytes -= g_pool[best].cap;
                g_pool[best] = g_pool.back();
                g_pool.pop_back();
            } else
                *out = nullptr;
        }
        if (*out) {
            // hipFree would have waited for everything queued on the device before the block could be handed out again: so does a hit (rare: an
            // index or a context being set up)
            (void)hipDeviceSynchronize();
            return hipSuccess;
        }
        hipError_t e = zh_dev_alloc_raw(out, bytes);
        if (e != hipSuccess) {  // out of memory with blocks in the cache: give them back and ask again
            (void)hipGetLastError();
            pool_trim();
            e = zh_dev_alloc_raw(out, bytes);
        }
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            g_pool_caps[*out] = bytes;
        }
        return e;
    }
    hipError_t e = zh_dev_alloc_raw(out, bytes);
    if (e != hipSuccess && pool_on() && g_pool_bytes) {
        (void)hipGetLastError();
        pool_trim();
        e = zh_dev_alloc_raw(out, bytes);
    }
    return e;
}
static void zh_dev_free_raw(void *p);
static void zh_dev_free(void *p) {
    if (!p) return;
    if (pool_on()) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool_caps.find(p);
        if (it != g_pool_caps.end()) {  // (a block of this library's, 32 MiB or more: it stays mapped; nothing reads it once a later user's work is queued behind this call in host order -- hipFree's own rule for the caller)
            g_pool.push_back(PoolBlock{p, it->second, dev});
            g_pool_bytes += it->second;
            return;
        }
    }
    zh_dev_free_raw(p);
}
static hipError_t zh_dev_alloc_raw(void **out, size_t bytes) {
    if (vmm_mode() <= 0 || bytes < (size_t(2) << 20)) return hipMalloc(out, bytes);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended)) != hipSuccess || gran == 0) return hipMalloc(out, bytes);
    static const size_t chunk_mb = [] { const char *c = getenv("ZH_VMM_CHUNK_MB"); return (size_t)(c ? atoi(c) : 256); }();
    // the reported granularity is 4 KiB on this platform (profiles/micro/vmm_probe.hip): the range's ALIGNMENT and the chunks' sizes are what let the
    // page tables use large fragments -- ZH_VMM_ALIGN_MB (default 2; 0 = the reported granularity, for the experiment's control)
    static const size_t align_mb = [] { const char *c = getenv("ZH_VMM_ALIGN_MB"); return (size_t)(c ? atoi(c) : 2); }();
    if (align_mb) gran = std::max<size_t>(gran, align_mb << 20);
    const size_t chunk = std::max<size_t>(gran, (chunk_mb << 20) / gran * gran);
    const size_t total = (bytes + gran - 1) / gran * gran;
    void *va = nullptr;
    if ((e = hipMemAddressReserve(&va, total, gran, nullptr, 0)) != hipSuccess) return e;
    VmmRec rec;
    rec.size = total;
    size_t off = 0;
    while (off < total) {
        const size_t sz = std::min(chunk, total - off);
        hipMemGenericAllocationHandle_t h;
        if ((e = hipMemCreate(&h, sz, &prop, 0)) != hipSuccess) break;
        if ((e = hipMemMap((char *)va + off, sz, 0, h, 0)) != hipSuccess) { hipMemRelease(h); break; }
        rec.handles.push_back(h); rec.sizes.push_back(sz);
        off += sz;
    }
    if (e == hipSuccess) {
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(va, total, &acc, 1);
    }
    if (e != hipSuccess) {
        size_t o2 = 0;
        for (size_t i = 0; i < rec.handles.size(); i++) { hipMemUnmap((char *)va + o2, rec.sizes[i]); hipMemRelease(rec.handles[i]); o2 += rec.sizes[i]; }
        hipMemAddressFree(va, total);
        (void)hipGetLastError();
        return e;
    }
    {
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        g_vmm.emplace(va, std::move(rec));
    }
    *out = va;
    return hipSuccess;
}
static void zh_dev_free_raw(void *p) {
    if (!p) return;
    VmmRec rec;
    bool mine = false;
    {
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        auto it = g_vmm.find(p);
        if (it != g_vmm.end()) { rec = std::move(it->second); g_vmm.erase(it); mine = true; }
    }
    if (!mine) { hipFree(p); return; }
    hipDeviceSynchronize();  // (hipFree's guarantee: nothing in flight still uses the range)
    size_t off = 0;
    for (size_t i = 0; i < rec.handles.size(); i++) { hipMemUnmap((char *)p + off, rec.sizes[i]); hipMemRelease(rec.handles[i]); off += rec.sizes[i]; }
    hipMemAddressFree(p, rec.size);
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    // hipFree waits for EVERY stream of the device to drain.  A per-batch scratch buffer that outgrows its capacity while other
    // batches are in flight would stall the host until their sweeps, selects and finals have run -- and the sweep queue runs dry
    // meanwhile (rocprofv3 kernel trace, r03: a 1.3-2.9 ms hole in front of the next window's sweep each time).  Buffers of a
    // search context therefore keep the outgrown allocation (geometric growth: at most twice the final size in total) until
    // the context is released, and ask for 1/8 more than a batch needs so that batch-to-batch variation rarely grows them at all.
    // ... but not without bound: a score table grows to 12 GB per context, and three generations of it would be held until the
    // context is destroyed.  Past 2 GiB of outgrown allocations they are freed on the spot (one drain, rare).
    bool defer = false;
    std::vector<void *> old;
    size_t old_bytes = 0;
    int ensure(size_t bytes, bool keep = false, hipStream_t s = nullptr) {
        if (bytes <= cap) return ZH_OK;
        if (defer) bytes += bytes / 8;
        if (defer && old_bytes + cap > (size_t(2) << 30)) {
            for (void *o : old) zh_dev_free(o);
            old.clear();
            old_bytes = 0;
        }
        size_t ncap = std::max(bytes, cap + cap / 2);
        ncap = (ncap + 255) & ~size_t(255);
        void *np = nullptr;
        hipError_t e = zh_dev_alloc(&np, ncap);
        if (e != hipSuccess) {
            ncap = (bytes + 255) & ~size_t(255);
            e = zh_dev_alloc(&np, ncap);
        }
        if (e != hipSuccess) return fail(ZH_ENOMEM, "hipMalloc(%zu bytes): %s", ncap, hipGetErrorString(e));
        if (keep && p && cap) {
            e = hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { zh_dev_free(np); return fail(ZH_EHIP, "grow copy: %s", hipGetErrorString(e)); }
        }
        if (p) { if (defer) { old.push_back(p); old_bytes += cap; } else zh_dev_free(p); }
        p = np;
        cap = ncap;
        return ZH_OK;
    }
    void release() {
        if (p) zh_dev_free(p);
        for (void *o : old) zh_dev_free(o);
        old.clear();
        old_bytes = 0;
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T *as() const { return reinterpret_cast<T *>(p); }
};

// One in-flight search batch: its own workspace, events and (borrowed) stream, so that several batches can be
// in flight on different streams (zh_search_begin / zh_search_finish / zh_search_wait).
struct zh_search_ctx {
    zh_index *ix = nullptr;
    DevBuf wQQ, wBits, wCounts, wInline, wRowBase, wCandBase, wVisitBase, wTotals, wVisits, wKeys, wCandKeys, wCandIds,
        wLeafCount, wLeafFill, wGroupBase, wGroupRowBase, wGroups, wGroupRowOff, wWaveGroup, wVisitBits, wNodeVisit, wScore, wJunkBits, wZeros, wQnorm, wQpad, wFixList, wLogPool, wLogHead, wLogCtl,
        wPfRows, wPfCounts, wPfKeys, wPfIds, wPfAmb, wPfCtl, wUnc, wQh, wQmeta, wApList, wApCount, wApEx, wApExKeys, wApCtl, wRaw;
    // test / debug (zh_debug_scan_pairs): the half-width batch this context ran last -- its ZhApprox, and (zh_debug_keep_raw) the scan's raw pairs
    ZhApprox dbg_ap{};
    bool dbg_valid = false, dbg_raw = false;
    size_t log_chunks = 0;  // capacity of wLogPool for the batch in flight
    ZhTotals *h_totals = nullptr;  // pinned
    hipEvent_t ev[6] = {};         // stage boundaries: hash | walk | sweep | select | final
    hipEvent_t ev_totals = nullptr, ev_emit = nullptr, ev_sw0 = nullptr, ev_sw1 = nullptr;
    bool ev_ok = false;
    // the batch in flight
    int state = 0;  // 0 idle, 1 begun (first half enqueued), 2 finished (second half enqueued, results pending)
    bool trivial = false;
    const float *dQ = nullptr;
    size_t B = 0, k = 0;       // B = every query of the window
    size_t nwin = 1, bwin = 0;  // a window = nwin API batches of bwin queries handled as ONE internal batch
    DevBuf wQwin, wOutWin;      // the window's queries side by side / its results before they are handed out per batch
    int metric = 0, mode = 0;
    uint32_t P_dense = 0, wpq = 0;
    hipStream_t s = nullptr;
    ZhTotals tot{};
    bool scan = false;  // the batch in flight was swept by the table scan (rows streamed once) instead of leaf by leaf
    // ... with half-width queries (zh_approx.hip): intervals from the scan, the reference's keys for the few rows they cannot
    // decide.  Its control words come back pinned for the statistics only: an overflow is redone on the device, in stream order.
    bool approx_f32rows = false;  // (approx_mfma without an fp16 copy of the table: scan_mfma_kernel<D, true>)
    bool approx_fused = false;  // (approx_leaf, round 6: intervals, bounds and lists inside the sweep kernel)
    bool approx_bytes = false;  // (approx_leaf, round 6: the sweep reads the copy of BYTES -- a table of integers 0 .. 255)
    bool approx = false, approx_mfma = false, approx_leaf = false;  // (approx_leaf: the d = 128 leaf-major sweep at half width, sweep128h_kernel)
    uint32_t *h_ap = nullptr;
    bool score_hash = false;  // its signs came from row scores (zh_score.hip) instead of one dot product per plane
    uint32_t score_Bp = 0;    // ... for this many (padded) queries per stored row
    bool lazy_fix = false;    // ... with the uncertain signs flagged (wUnc) for the blocked walk to recompute when it meets one
    // prefilter (zh_search.hip): the batch's candidates are picked from those row scores and only they are scored exactly.  Its
    // control words come back pinned; a list that ran over makes zh_search_wait redo the batch the classic way, from these:
    bool prefilter = false, prefilter_off_once = false, stream_ordered = false;
    uint32_t pf_cap = 0;
    uint32_t *h_pf = nullptr;
    std::vector<const float *> sv_q;
    std::vector<uint64_t *> sv_ids, sv_keys;
    std::vector<uint32_t *> sv_counts;
    hipStream_t sv_heavy = nullptr;
    std::vector<DevBuf *> all_bufs() {
        return {&wQQ, &wBits, &wCounts, &wInline, &wRowBase, &wCandBase, &wVisitBase, &wTotals, &wVisits, &wKeys,
                &wCandKeys, &wCandIds, &wLeafCount, &wLeafFill, &wGroupBase, &wGroupRowBase, &wGroups, &wGroupRowOff, &wWaveGroup, &wVisitBits, &wNodeVisit, &wScore, &wJunkBits, &wZeros, &wQnorm, &wQpad, &wFixList,
                &wLogPool, &wLogHead, &wLogCtl, &wQwin, &wOutWin, &wPfRows, &wPfCounts, &wPfKeys, &wPfIds, &wPfAmb, &wPfCtl, &wUnc,
                &wQh, &wQmeta, &wApList, &wApCount, &wApEx, &wApExKeys, &wApCtl, &wRaw};
    }
    void release_all() {
        for (DevBuf *b : all_bufs()) b->release();
        if (ev_ok) { for (auto &e : ev) hipEventDestroy(e); hipEventDestroy(ev_totals); hipEventDestroy(ev_emit); hipEventDestroy(ev_sw0); hipEventDestroy(ev_sw1); ev_ok = false; }
        if (h_totals) { hipHostFree(h_totals); h_totals = nullptr; }
        if (h_pf) { hipHostFree(h_pf); h_pf = nullptr; }
        if (h_ap) { hipHostFree(h_ap); h_ap = nullptr; }
    }
};

struct zh_index {
    zh_options opt{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t sweep_stream = nullptr;  // lowest priority: shared by the sweeps of pipelined contexts
    // exclusive: everything that changes the index or uses its one blocking context (dctx); shared: the combining front end's lanes,
    // which search on contexts of their own (searches on different contexts run side by side, as the pipelined calls do)
    std::shared_mutex mu;

    // stored vectors: n_rows x dim row-major f32 (Embedding<N>, lib.rs:18)
    DevBuf X;
    uint64_t n_rows = 0;

    // forest
    DevBuf node_plane, node_left, node_right, node_pack, roots, planes, consts, leaf_ids;
    uint32_t n_nodes = 0, n_planes = 0, n_trees = 0;
    uint64_t n_leaf_ids = 0;
    std::vector<int32_t> h_plane, h_left, h_right;
    std::vector<uint32_t> h_roots;
    std::vector<uint32_t> planes_below_level;  // [L] = planes whose level < L ; planes are stored level-major
    std::vector<uint32_t> h_leaf_ids;          // host mirror of leaf_ids, materialised by the first incremental add
    std::vector<uint8_t> h_dead;               // rows removed by zh_index_remove (their vectors stay in X)
    uint64_t n_dead = 0;
    std::vector<uint32_t> h_live;              // ascending live rows, for hyperplane sampling once rows were removed
    uint64_t h_live_rows = ~0ull;              // n_rows / n_dead / dead_gen h_live was built for
    uint64_t h_live_dead = ~0ull;
    uint64_t h_live_gen = ~0ull;
    // Caches derived from the stored rows are keyed on a generation as well as on a count: clear() followed by a refill to the
    // same count must not find them valid.  rows_gen: bumped whenever the CONTENT of rows [0, n_rows) may have changed (clear);
    // dead_gen: whenever the set of removed rows changed (remove, clear).
    uint64_t rows_gen = 1, dead_gen = 1;
    uint32_t max_leaf_len = 0;
    // blocked view of the forest (ZhBlocksDev) for all-dense walks: built on first use, dropped whenever the trees change
    DevBuf blk_recs, blk_upper, blk_roots;
    uint32_t n_blocks = 0, n_upper = 0;
    bool blocks_valid = false;
    bool blocks_inner = false;  // the view holds blocks of INNER nodes only (round 5; ZhBlocksDev::inner_only)
    size_t blk_recs_b = 0;      // ... whose records' second halves start at this index of blk_recs
    std::mutex blk_mu;
    // row -> (leaf, position) per tree for the table-scan sweep (zh_launch_row_leaf): built on first use, dropped with the trees
    DevBuf row_leaf;
    bool row_leaf_valid = false, row_leaf_failed = false;
    // Both derived views cost host time per build (a traversal of every node: ~0.35 s for the 13M nodes of a 1M-row forest at the
    // default leaf size), so a forest that changes between searches (insert, search, insert, ...) keeps the pointer walk / the
    // leaf-major sweep: the views are built once the forest has served a few batches unchanged.
    std::atomic<uint32_t> batches_since_change{0};
    // the MFMA table scan's view of the stored rows (zh_launch_row_half): their fp16 copy in the A operand's order (2 * d bytes per row), per
    // row {|x|^2, 1 / scale}, and the largest relative rounding error of any row; for rows [0, scale_rows) of generation scale_gen,
    // extended when rows are appended; a failed allocation is remembered until the rows change (the VALU kernel serves the index)
    DevBuf row_half, row_meta, row_rho_dev;
    // (the flags and generations below are read without a lock by the per-batch choices -- mfma_wanted, choose_scan, use_approx_leaf -- while another
    // context may be inside ensure_row_half / ensure_row_half128 under blk_mu: atomics; the buffers themselves are only touched under blk_mu)
    std::atomic<bool> row_half_failed{false};
    // d = 128, leaf by leaf at half width (sweep128h_kernel): a row-major fp16 copy under ONE power-of-two scale (2^(14 - h128_ex)), same life cycle
    DevBuf row_half128;
    std::atomic<uint64_t> h128_rows{0}, h128_gen{0};
    int h128_ex = 0;
    float h128_rho = 0.f;
    std::atomic<bool> h128_failed{false};
    // ... or, for a table of integers 0 .. 255 (SIFT descriptors), an EXACT copy of 128 bytes per row in the same buffer (row_byte128_kernel):
    // h128_bytes = the current copy is that one; h128_not_bytes = some stored row is known not to qualify (until the rows are replaced)
    bool h128_bytes = false, h128_not_bytes = false;
    std::atomic<uint64_t> scale_rows{0}, scale_gen{0};
    float row_rho = 0.f;
    // Round 5 (VERDICT r4 #4b): the matrix-core scan's OWN view of the rows -- the fp16 tiles, their {|x|^2, 1 / scale} and the row -> leaf entries --
    // is kept sorted by (leaf in tree 0, leaf in tree 1, leaf in tree 2, id): position p holds row scan_perm[p] (zh_order.hip).  The scan never needs
    // a row's id (its outputs go to key slots that the row -> leaf entries name), so nothing else changes; a wave's 16 rows then share tree 0's leaf --
    // its visitors are ONE tile column each for all 16 -- and, on data with structure, most other trees' too, whatever order the rows were inserted
    // in.  Made with the copy, from the row -> leaf table; rows appended later keep position = id; the f32 rows and every other kernel stay in id
    // order.  row_leaf_p = the row -> leaf table gathered into that order (for row_leaf generation row_leaf_p_gen).
    DevBuf scan_perm, row_leaf_p;
    bool row_order_off = false;
    uint32_t order_keys = 0;   // trees whose leaves the kept order sorts by (0: id order)
    double order_worth = 0;    // share of (adjacent positions, tree) combinations in the same leaf under the kept order
    uint64_t perm_rows = 0, row_leaf_gen = 0, row_leaf_p_gen = 0, perm_gen = 0, row_leaf_p_perm = 0;
    uint64_t row_leaf_rows = 0;  // stored rows the table was built for (rows appended since are in no tree yet, but must not be scanned past it)

    // the blocking entry points run on this context (under `mu`); staging buffers of the host-pointer variant
    zh_search_ctx dctx;
    DevBuf wQ, wOutIds, wOutKeys, wOutCounts;

    bool broken = false;  // an incremental add failed half way: trees are stale until zh_index_build
    int dense_levels = -1;
    bool debug_keep_raw = false;  // zh_debug_keep_raw
    std::atomic<int> sweep_mode{0};  // zh_set_sweep_mode (atomic: set by one thread while pipelined contexts read it per batch): 0 cost model (prefilter where the batch has row scores), 1 leaf-major, 2 table scan (exact), 3 = 0, 4 table scan with half-width queries wherever it applies, 5 = 4 with the VALU kernel only (no fp16 copy of the rows)
    // a batch whose half-width scan ran over was redone by the f32 scan: both scans paid.  Data whose keys are dense around the cut (the
    // parity cosine key on iid rows in 20k-row leaves: thousands of rows per query inside the bound) would do so batch after batch:
    // after the first such batch -- or one whose lists came close -- the per-query lists get 8192 slots instead of 4096 (a final
    // kernel with twice the LDS); after the second the index keeps the f32 scan until its trees change
    std::atomic<uint32_t> approx_strikes{0};
    // the matrix-core scan on f32 rows (no fp16 copy) rounds the rows as well: wider intervals, longer lists.  Lists that run over at 8192 slots send the
    // index back to the VALU half-width scan (which rounds the queries only) instead of all the way to the f32 scan
    std::atomic<bool> mfma_f32_off{false};
    // ... and a FUSED d = 128 sweep whose lists ran over (it hands on what the bounds known at the time cannot rule out: more than the select pass's exact
    // per-visit bounds would) goes back to the unfused half-width sweep first, not to the f32 sweep
    std::atomic<bool> fused_off{false};
    int hash_mode = 0;   // zh_set_hash_mode: 0 chosen per batch, 1 one dot product per plane, 2 row scores where the forest allows
    // the two sample rows of every plane (build_hyperplane, lsh.rs:197-225), kept for forests this library built or grew: the
    // row-score hash derives signs from them.  An injected forest (zh_index_set_forest) has arbitrary planes: not valid.
    DevBuf plane_samples;
    bool samples_valid = false;
    DevBuf row_hn2, row_norm;  // |r|^2 / 2 and |r| of the stored rows, for the first norm_rows rows of generation norm_gen
    uint64_t norm_rows = 0, norm_gen = 0;
    DevBuf plane_hab;          // the same per PLANE (its two sample rows), for the first hab_planes planes at (hab_rows, hab_gen)
    uint32_t hab_planes = 0;
    uint64_t hab_rows = 0, hab_gen = 0;
    // {|r|^2/2, |r|} of the row in every slot of leaf_ids, for the prefilter: built on first use from the norms of (meta_rows,
    // meta_gen), dropped with the trees
    DevBuf leaf_meta;
    bool leaf_meta_valid = false;
    uint64_t meta_rows = 0, meta_gen = 0;
    // batches in a row that the prefilter handed back to the sweep (a list ran over: rows the bound cannot tell apart, many long
    // leaves): after two the forest is swept until it changes -- such data would pay for both every batch
    std::atomic<uint32_t> prefilter_strikes{0};
    bool scan_unsafe = false;  // an injected forest lists a row twice in one tree: rowLeaf holds one slot per (row, tree) -> leaf-major only
    double visits_per_pair = 0;  // leaf visits per (query, tree) pair, running mean over the batches so far (stats_mu)
    int profiling = 0;
    std::mutex stats_mu;
    zh_stats_t stats{};
    // Combining front end of zh_search_batch (the crate calls LSHIndex::search with ONE query from many rayon workers,
    // core.rs:299-303): callers that arrive while a batch is on the GPU queue their requests; the thread that finds the engine
    // idle leads -- it runs every compatible queued request (same top_k, metric, mode) as ONE internal batch and hands out the
    // results -- and the others sleep on the condition variable until theirs is done.
    struct CombineReq {
        const float *q; size_t b, k; int metric, mode;
        uint64_t *ids, *keys; uint32_t *counts;
        int rc = 0; std::string err;
        // completion and the appointment as next leader travel on the request's OWN mutex / condition variable: a finished round
        // wakes its callers side by side (one shared condition variable made 64 woken threads queue for one mutex)
        std::mutex m; std::condition_variable cv; bool done = false, lead = false;
    };
    // The lane a round runs on: a search context, a stream and staging of its own (not the index's blocking context: zh_search_batch
    // holds the index lock SHARED, zh_search_batch_device / add / remove exclusively).  One lane: a second one, so that one round's
    // host work overlaps the other's kernels, was measured slower (67 k against 80 k calls/s from 64 threads: rounds half the
    // size, and the HIP runtime serialises the two host threads' launches anyway).
    struct Lane {
        zh_search_ctx ctx;
        bool init = false, busy = false;
        hipStream_t s = nullptr;
        // a large host-resident batch is cut into windows that alternate between two contexts (search_host_windows): the second one
        zh_search_ctx ctx2;
        bool init2 = false;
        hipStream_t s2 = nullptr;
        // ... and, where hash + walk of a window outweigh its sweep (the crate's default options: thousands of leaf visits per pair), a third:
        // window w + 1 is BEGUN before window w is finished (its hash runs beside w's walk), as bench.py's look-ahead loop does (round 6)
        zh_search_ctx ctx3;
        bool init3 = false;
        hipStream_t s3 = nullptr;
        // ... and a fourth for TWO windows begun ahead (the finish of window w + 1 -- its walk -- is then queued before window w's results are
        // waited for: w's prefilter and select run beside it)
        zh_search_ctx ctx4;
        bool init4 = false;
        hipStream_t s4 = nullptr;
        hipEvent_t ev_d2h[4] = {nullptr, nullptr, nullptr, nullptr};
        DevBuf wQ, wOutIds, wOutKeys, wOutCounts;
        void *h_stage = nullptr;  // pinned staging of a combined batch: queries in, results out (one H2D, three D2H per round)
        size_t h_stage_cap = 0;
    };
    std::mutex cmu;
    std::deque<CombineReq *> cpend;
    bool cleader = false;   // some thread is leading (running rounds or about to)
    size_t clast = 0;       // callers served by the previous round
    Lane lanes[1];
};

static int set_device(const zh_index *ix) {
    hipError_t e = hipSetDevice(ix->device);
    if (e != hipSuccess) return fail(ZH_EHIP, "hipSetDevice(%d): %s", ix->device, hipGetErrorString(e));
    return ZH_OK;
}

static ZhForestDev forest_dev(const zh_index *ix) {
    ZhForestDev f;
    f.node_plane = ix->node_plane.as<int32_t>();
    f.node_left = ix->node_left.as<int32_t>();
    f.node_right = ix->node_right.as<int32_t>();
    f.node_pack = ix->node_pack.as<int4>();
    f.roots = ix->roots.as<uint32_t>();
    f.planes = ix->planes.as<float>();
    f.consts = ix->consts.as<float>();
    f.leaf_ids = ix->leaf_ids.as<uint32_t>();
    f.n_nodes = ix->n_nodes;
    f.n_planes = ix->n_planes;
    f.n_trees = ix->n_trees;
    f.group = zh_group_size(ix->opt.dim);
    return f;
}

static int ctx_init(zh_search_ctx *c, zh_index *ix) {
    c->ix = ix;
    for (DevBuf *b : c->all_bufs()) b->defer = true;  // never hipFree (a device-wide drain) between batches
    hipError_t e = hipHostMalloc((void **)&c->h_totals, sizeof(ZhTotals), hipHostMallocDefault);
    if (e != hipSuccess) return fail(ZH_ENOMEM, "hipHostMalloc: %s", hipGetErrorString(e));
    e = hipHostMalloc((void **)&c->h_pf, 4 * sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) return fail(ZH_ENOMEM, "hipHostMalloc: %s", hipGetErrorString(e));
    memset(c->h_pf, 0, 4 * sizeof(uint32_t));
    e = hipHostMalloc((void **)&c->h_ap, ZH_APX_CTL_WORDS * sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) return fail(ZH_ENOMEM, "hipHostMalloc: %s", hipGetErrorString(e));
    memset(c->h_ap, 0, ZH_APX_CTL_WORDS * sizeof(uint32_t));
    for (auto &ev : c->ev) HIPCHK(hipEventCreate(&ev));
    HIPCHK(hipEventCreateWithFlags(&c->ev_totals, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_emit, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&c->ev_sw0));
    HIPCHK(hipEventCreate(&c->ev_sw1));
    c->ev_ok = true;
    return ZH_OK;
}

// ------------------------------------------------------------------------------------------------
// lifecycle
// ------------------------------------------------------------------------------------------------
extern "C" void zh_options_default(zh_options *o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->max_node_size = 5;  // lsh.rs:134
    o->num_trees = 15;     // lsh.rs:135
    o->seed = 0x5EB2A003ull;
    o->device = -1;
}

extern "C" int zh_index_create(const zh_options *opt, zh_index **out) {
    if (!opt || !out) return fail(ZH_EINVAL, "zh_index_create: null argument");
    if (opt->dim == 0) return fail(ZH_EINVAL, "zh_index_create: dim must be > 0");
    if (opt->dim > ZH_MAX_DIM) return fail(ZH_ELIMIT, "zh_index_create: dim %u > ZH_MAX_DIM (%u)", opt->dim, ZH_MAX_DIM);
    if (opt->num_trees > 4096) return fail(ZH_ELIMIT, "zh_index_create: num_trees > 4096");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(ZH_EHIP, "no usable HIP device (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    zh_index *ix = new (std::nothrow) zh_index();
    if (!ix) return fail(ZH_ENOMEM, "out of host memory");
    ix->opt = *opt;
    if (opt->device < 0) {
        if (hipGetDevice(&ix->device) != hipSuccess) ix->device = 0;
    } else
        ix->device = opt->device;
    if (ix->device >= ndev) { delete ix; return fail(ZH_EINVAL, "device %d out of range (%d devices)", opt->device, ndev); }
    int rc = set_device(ix);
    if (rc) { delete ix; return rc; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ix->device) == hipSuccess) {
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            delete ix;
            return fail(ZH_EHIP, "device %d is %s; this library is built for gfx950 only", opt->device, prop.gcnArchName);
        }
    }
    e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ix; return fail(ZH_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
        e = hipStreamCreateWithPriority(&ix->sweep_stream, hipStreamNonBlocking, least);
        if (e != hipSuccess) { hipStreamDestroy(ix->stream); delete ix; return fail(ZH_EHIP, "hipStreamCreateWithPriority: %s", hipGetErrorString(e)); }
    }
    rc = ctx_init(&ix->dctx, ix);
    if (rc) { hipStreamDestroy(ix->sweep_stream); hipStreamDestroy(ix->stream); delete ix; return rc; }
    if (opt->reserve_rows) {
        rc = ix->X.ensure((size_t)opt->reserve_rows * opt->dim * sizeof(float));
        if (rc) { zh_index_destroy(ix); return rc; }
    }
    *out = ix;
    return ZH_OK;
}

static void free_forest(zh_index *ix) {
    ix->node_plane.release(); ix->node_left.release(); ix->node_right.release(); ix->node_pack.release(); ix->roots.release();
    ix->planes.release(); ix->consts.release(); ix->leaf_ids.release();
    ix->blk_recs.release(); ix->blk_upper.release(); ix->blk_roots.release();
    ix->n_blocks = 0; ix->blocks_valid = false;
    ix->leaf_meta.release(); ix->leaf_meta_valid = false; ix->prefilter_strikes = 0; ix->approx_strikes = 0; ix->mfma_f32_off = false; ix->fused_off = false;
    ix->row_leaf.release(); ix->row_leaf_valid = false; ix->row_leaf_failed = false; ix->scan_unsafe = false;
    {
        // the matrix-core scan's row order was measured on THIS forest's leaves: the next half-width batch re-makes the copy in the new forest's
        // order, and an order given up for lack of room gets another chance
        std::lock_guard<std::mutex> lb(ix->blk_mu);
        if (ix->perm_rows || ix->order_keys) ix->scale_gen = 0;
        ix->row_order_off = false;
    }
    ix->plane_samples.release(); ix->samples_valid = false;
    ix->plane_hab.release(); ix->hab_planes = 0; ix->hab_rows = 0; ix->hab_gen = 0;
    ix->n_nodes = ix->n_planes = ix->n_trees = 0;
    ix->n_leaf_ids = 0;
    ix->h_plane.clear(); ix->h_left.clear(); ix->h_right.clear(); ix->h_roots.clear();
    ix->planes_below_level.clear();
    ix->h_leaf_ids.clear();
    ix->max_leaf_len = 0;
    {
        std::lock_guard<std::mutex> lk(ix->stats_mu);
        ix->visits_per_pair = 0;
    }
}

extern "C" void zh_index_destroy(zh_index *ix) {
    if (ix)
        for (auto &ln : ix->lanes) {
            if (ln.init) { hipSetDevice(ix->device); ln.ctx.release_all(); }
            if (ln.init2) { hipSetDevice(ix->device); ln.ctx2.release_all(); }
            if (ln.init3) { hipSetDevice(ix->device); ln.ctx3.release_all(); }
            if (ln.init4) { hipSetDevice(ix->device); ln.ctx4.release_all(); }
            if (ln.s2) hipStreamDestroy(ln.s2);
            if (ln.s3) hipStreamDestroy(ln.s3);
            if (ln.s4) hipStreamDestroy(ln.s4);
            for (auto &ev : ln.ev_d2h) if (ev) hipEventDestroy(ev);
            ln.wQ.release(); ln.wOutIds.release(); ln.wOutKeys.release(); ln.wOutCounts.release();
            if (ln.s) hipStreamDestroy(ln.s);
            if (ln.h_stage) hipHostFree(ln.h_stage);
        }
    if (!ix) return;
    hipSetDevice(ix->device);
    if (ix->stream) hipStreamSynchronize(ix->stream);
    if (ix->sweep_stream) hipStreamSynchronize(ix->sweep_stream);
    free_forest(ix);
    ix->X.release();
    ix->row_hn2.release(); ix->row_norm.release(); ix->row_half.release(); ix->row_meta.release(); ix->row_rho_dev.release(); ix->row_half128.release();
    ix->scan_perm.release(); ix->row_leaf_p.release();
    ix->dctx.release_all();
    DevBuf *ws[] = {&ix->wQ, &ix->wOutIds, &ix->wOutKeys, &ix->wOutCounts};
    for (DevBuf *b : ws) b->release();
    if (ix->sweep_stream) hipStreamDestroy(ix->sweep_stream);
    if (ix->stream) hipStreamDestroy(ix->stream);
    delete ix;
}

extern "C" int zh_index_clear(zh_index *ix) {
    if (!ix) return fail(ZH_EINVAL, "null index");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    int rc = set_device(ix);
    if (rc) return rc;
    hipStreamSynchronize(ix->stream);
    free_forest(ix);
    ix->broken = false;
    ix->n_rows = 0;
    ix->h_dead.clear();
    ix->n_dead = 0;
    // everything derived from the rows that were just dropped: the row norms of the row-score hash, the live-row list of
    // the hyperplane sampler (both were keyed on counts alone: a refill to the same count found them "valid")
    ix->rows_gen++; ix->dead_gen++;
    ix->norm_rows = 0; ix->norm_gen = 0;
    ix->row_hn2.release(); ix->row_norm.release();
    std::lock_guard<std::mutex> lb(ix->blk_mu);  // (the fp16 copies' state is blk_mu's: zh_stats reads it under that lock)
    ix->scale_rows = 0; ix->scale_gen = 0; ix->row_rho = 0.f; ix->row_half.release(); ix->row_meta.release(); ix->row_half_failed = false;
    ix->perm_rows = 0; ix->perm_gen++; ix->scan_perm.release(); ix->row_leaf_p.release(); ix->row_order_off = false;  // (a new table gets a new chance at the scan's row order)
    ix->h128_rows = 0; ix->h128_gen = 0; ix->h128_rho = 0.f; ix->row_half128.release(); ix->h128_failed = false;
    ix->h128_bytes = false; ix->h128_not_bytes = false;
    ix->h_live.clear(); ix->h_live.shrink_to_fit();
    ix->h_live_rows = ix->h_live_dead = ix->h_live_gen = ~0ull;
    return ZH_OK;
}

extern "C" uint64_t zh_index_count(const zh_index *ix) { return ix ? ix->n_rows - ix->n_dead : 0; }
extern "C" uint32_t zh_index_num_trees(const zh_index *ix) { return ix ? ix->n_trees : 0; }
extern "C" uint32_t zh_index_dim(const zh_index *ix) { return ix ? ix->opt.dim : 0; }
extern "C" int32_t zh_index_device(const zh_index *ix) { return ix ? ix->device : -1; }
extern "C" uint64_t zh_index_id_base(const zh_index *ix) { return ix ? ix->opt.id_base : 0; }
extern "C" const float *zh_index_rows_device(const zh_index *ix) { return ix ? ix->X.as<float>() : nullptr; }
extern "C" void *zh_index_sweep_stream(const zh_index *ix) { return ix ? (void *)ix->sweep_stream : nullptr; }

// ------------------------------------------------------------------------------------------------
// rows
// ------------------------------------------------------------------------------------------------
static int grow_rows(zh_index *ix, size_t n_more) {
    size_t need = (size_t)(ix->n_rows + n_more) * ix->opt.dim * sizeof(float);
    if (ix->n_rows + n_more > 0xFFFFFFFFull) return fail(ZH_ELIMIT, "more than 2^32-1 rows in one index (shard it)");
    return ix->X.ensure(need, true, ix->stream);
}

static int append_locked(zh_index *ix, const float *rows, size_t n, uint64_t *out_ids) {
    int rc = set_device(ix);
    if (rc) return rc;
    if ((rc = grow_rows(ix, n))) return rc;
    if (n) {
        HIPCHK(hipMemcpyAsync(ix->X.as<float>() + (size_t)ix->n_rows * ix->opt.dim, rows,
                              n * ix->opt.dim * sizeof(float), hipMemcpyHostToDevice, ix->stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
    }
    if (out_ids) for (size_t i = 0; i < n; i++) out_ids[i] = ix->opt.id_base + ix->n_rows + i;
    ix->n_rows += n;
    return ZH_OK;
}
extern "C" int zh_index_append(zh_index *ix, const float *rows, size_t n, uint64_t *out_ids) {
    if (!ix || (!rows && n)) return fail(ZH_EINVAL, "zh_index_append: null argument");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    return append_locked(ix, rows, n, out_ids);
}

extern "C" int zh_index_read_rows(zh_index *ix, uint64_t first, size_t n, float *out) {
    if (!ix || (n && !out)) return fail(ZH_EINVAL, "zh_index_read_rows: null argument");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    int rc = set_device(ix);
    if (rc) return rc;
    if (first + n > ix->n_rows) return fail(ZH_EINVAL, "zh_index_read_rows: rows [%llu, %llu) out of range (%llu stored)",
                                            (unsigned long long)first, (unsigned long long)(first + n), (unsigned long long)ix->n_rows);
    if (n) HIPCHK(hipMemcpy(out, ix->X.as<float>() + (size_t)first * ix->opt.dim, n * ix->opt.dim * sizeof(float), hipMemcpyDeviceToHost));
    return ZH_OK;
}

extern "C" int zh_index_append_device(zh_index *ix, const float *d_rows, size_t n) {
    if (!ix || (!d_rows && n)) return fail(ZH_EINVAL, "zh_index_append_device: null argument");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    int rc = set_device(ix);
    if (rc) return rc;
    if ((rc = grow_rows(ix, n))) return rc;
    if (n) {
        HIPCHK(hipMemcpyAsync(ix->X.as<float>() + (size_t)ix->n_rows * ix->opt.dim, d_rows,
                              n * ix->opt.dim * sizeof(float), hipMemcpyDeviceToDevice, ix->stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
    }
    ix->n_rows += n;
    return ZH_OK;
}

extern "C" int zh_index_append_synthetic(zh_index *ix, size_t n, uint64_t seed, uint64_t first_row, int kind) {
    if (!ix) return fail(ZH_EINVAL, "null index");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    int rc = set_device(ix);
    if (rc) return rc;
    if ((rc = grow_rows(ix, n))) return rc;
    HIPCHK(zh_launch_synth_rows(ix->X.as<float>() + (size_t)ix->n_rows * ix->opt.dim, n, ix->opt.dim, seed, first_row,
                                kind, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    ix->n_rows += n;
    return ZH_OK;
}

// ------------------------------------------------------------------------------------------------
// forest upload (shared by set_forest and build): node arrays on the host, planes level-major
// ------------------------------------------------------------------------------------------------
static int upload_nodes(zh_index *ix) {
    size_t nn = ix->h_plane.size();
    int rc;
    if ((rc = ix->node_plane.ensure(std::max<size_t>(nn, 1) * 4))) return rc;
    if ((rc = ix->node_left.ensure(std::max<size_t>(nn, 1) * 4))) return rc;
    if ((rc = ix->node_right.ensure(std::max<size_t>(nn, 1) * 4))) return rc;
    if ((rc = ix->node_pack.ensure(std::max<size_t>(nn, 1) * sizeof(int4)))) return rc;
    if ((rc = ix->roots.ensure(std::max<size_t>(ix->h_roots.size(), 1) * 4))) return rc;
    if (nn) {
        HIPCHK(hipMemcpyAsync(ix->node_plane.p, ix->h_plane.data(), nn * 4, hipMemcpyHostToDevice, ix->stream));
        HIPCHK(hipMemcpyAsync(ix->node_left.p, ix->h_left.data(), nn * 4, hipMemcpyHostToDevice, ix->stream));
        HIPCHK(hipMemcpyAsync(ix->node_right.p, ix->h_right.data(), nn * 4, hipMemcpyHostToDevice, ix->stream));
        // the plane constants are final on this stream by now (set_forest's copy, make_planes of build / insert)
        HIPCHK(zh_launch_pack_nodes(ix->node_plane.as<int32_t>(), ix->node_left.as<int32_t>(), ix->node_right.as<int32_t>(),
                                    ix->consts.as<float>(), ix->node_pack.as<int4>(), (uint32_t)nn, ix->stream));
    }
    if (!ix->h_roots.empty())
        HIPCHK(hipMemcpyAsync(ix->roots.p, ix->h_roots.data(), ix->h_roots.size() * 4, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    ix->n_nodes = (uint32_t)nn;
    ix->n_trees = (uint32_t)ix->h_roots.size();
    ix->blocks_valid = false;  // the trees changed
    ix->leaf_meta_valid = false; ix->prefilter_strikes = 0; ix->approx_strikes = 0; ix->mfma_f32_off = false; ix->fused_off = false;
    ix->row_leaf_valid = false; ix->row_leaf_failed = false;
    ix->batches_since_change = 0;
    ix->max_leaf_len = 0;
    for (size_t i = 0; i < nn; i++)
        if (ix->h_plane[i] < 0) ix->max_leaf_len = std::max(ix->max_leaf_len, (uint32_t)ix->h_right[i]);
    return ZH_OK;
}

extern "C" int zh_index_set_forest(zh_index *ix, const zh_forest_view *fv) {
    if (!ix || !fv) return fail(ZH_EINVAL, "zh_index_set_forest: null argument");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    int rc = set_device(ix);
    if (rc) return rc;
    const uint32_t nn = fv->n_nodes, np = fv->n_planes, nt = fv->n_trees, d = ix->opt.dim;
    if (nt && (!fv->roots || !fv->plane || !fv->left || !fv->right)) return fail(ZH_EINVAL, "set_forest: null arrays");
    if (np && (!fv->planes || !fv->consts)) return fail(ZH_EINVAL, "set_forest: null plane arrays");
    if (fv->n_leaf_ids && !fv->leaf_ids) return fail(ZH_EINVAL, "set_forest: null leaf_ids");
    if (fv->n_leaf_ids > 0xFFFFFFFFull) return fail(ZH_ELIMIT, "set_forest: more than 2^32-1 leaf entries");
    // validate + level of every node (BFS per tree)
    std::vector<int32_t> level(nn, -1);
    std::vector<uint32_t> order;  // inner nodes in (level, tree, bfs) order
    std::vector<std::vector<uint32_t>> by_level;
    std::vector<std::pair<uint32_t, uint32_t>> tree_leaves;  // (tree, leaf node)
    for (uint32_t t = 0; t < nt; t++) {
        if (fv->roots[t] >= nn) return fail(ZH_EINVAL, "set_forest: root %u out of range", t);
        std::vector<uint32_t> cur{fv->roots[t]}, nxt;
        int32_t lv = 0;
        while (!cur.empty()) {
            if ((size_t)lv >= by_level.size()) by_level.resize(lv + 1);
            for (uint32_t n : cur) {
                if (level[n] >= 0) return fail(ZH_EINVAL, "set_forest: node %u reachable twice", n);
                level[n] = lv;
                if (fv->plane[n] >= 0) {
                    if ((uint32_t)fv->plane[n] >= np) return fail(ZH_EINVAL, "set_forest: plane index out of range at node %u", n);
                    if (fv->left[n] < 0 || (uint32_t)fv->left[n] >= nn || fv->right[n] < 0 || (uint32_t)fv->right[n] >= nn)
                        return fail(ZH_EINVAL, "set_forest: child index out of range at node %u", n);
                    by_level[lv].push_back(n);
                    nxt.push_back((uint32_t)fv->left[n]);
                    nxt.push_back((uint32_t)fv->right[n]);
                } else {
                    uint64_t off = (uint32_t)fv->left[n], len = (uint32_t)fv->right[n];
                    if (fv->right[n] < 0 || off + len > fv->n_leaf_ids) return fail(ZH_EINVAL, "set_forest: leaf range out of bounds at node %u", n);
                    tree_leaves.push_back({t, n});
                }
            }
            cur.swap(nxt);
            nxt.clear();
            lv++;
            if (lv > 63) return fail(ZH_ELIMIT, "set_forest: tree deeper than 63 levels");
        }
    }
    for (uint64_t i = 0; i < fv->n_leaf_ids; i++)
        if (fv->leaf_ids[i] >= ix->n_rows) return fail(ZH_EINVAL, "set_forest: leaf id %u >= stored rows %llu", fv->leaf_ids[i], (unsigned long long)ix->n_rows);
    // A row listed twice inside ONE tree (a repeated id, or two leaves of a tree over the same leaf_ids range) is legal input --
    // the leaf-major sweep scores every listed occurrence -- but the table scan keeps one {leaf, position} per (row, tree): such
    // a forest is served leaf by leaf only, whatever the cost model or zh_set_sweep_mode say, so results never depend on the sweep.
    bool dup_in_tree = false;
    {
        std::vector<uint32_t> stamp(ix->n_rows, 0xFFFFFFFFu);
        for (size_t i = 0; i < tree_leaves.size() && !dup_in_tree; i++) {
            const uint32_t t = tree_leaves[i].first, n = tree_leaves[i].second;
            const uint32_t off = (uint32_t)fv->left[n], len = (uint32_t)fv->right[n];
            for (uint32_t j = 0; j < len; j++) {
                const uint32_t r = fv->leaf_ids[(size_t)off + j];
                if (stamp[r] == t) { dup_in_tree = true; break; }
                stamp[r] = t;
            }
        }
    }
    // renumber planes level-major
    std::vector<int32_t> new_of_old(np, -1);
    std::vector<uint32_t> old_of_new;
    std::vector<uint32_t> below{0};
    for (auto &lvl : by_level) {
        for (uint32_t n : lvl) {
            uint32_t op = (uint32_t)fv->plane[n];
            if (new_of_old[op] < 0) { new_of_old[op] = (int32_t)old_of_new.size(); old_of_new.push_back(op); }
        }
        below.push_back((uint32_t)old_of_new.size());
    }
    hipStreamSynchronize(ix->stream);
    free_forest(ix);
    ix->h_plane.assign(fv->plane, fv->plane + nn);
    ix->h_left.assign(fv->left, fv->left + nn);
    ix->h_right.assign(fv->right, fv->right + nn);
    ix->h_roots.assign(fv->roots, fv->roots + nt);
    for (uint32_t n = 0; n < nn; n++)
        if (ix->h_plane[n] >= 0) ix->h_plane[n] = level[n] >= 0 ? new_of_old[ix->h_plane[n]] : 0;
    const uint32_t np_used = (uint32_t)old_of_new.size();
    std::vector<float> hp((size_t)std::max<uint32_t>(np_used, 1) * d), hc(std::max<uint32_t>(np_used, 1));
    for (uint32_t i = 0; i < np_used; i++) {
        memcpy(&hp[(size_t)i * d], fv->planes + (size_t)old_of_new[i] * d, d * sizeof(float));
        hc[i] = fv->consts[old_of_new[i]];
    }
    if ((rc = ix->planes.ensure(hp.size() * 4))) return rc;
    if ((rc = ix->consts.ensure(hc.size() * 4))) return rc;
    if ((rc = ix->leaf_ids.ensure(std::max<uint64_t>(fv->n_leaf_ids, 1) * 4))) return rc;
    HIPCHK(hipMemcpyAsync(ix->planes.p, hp.data(), hp.size() * 4, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(hipMemcpyAsync(ix->consts.p, hc.data(), hc.size() * 4, hipMemcpyHostToDevice, ix->stream));
    if (fv->n_leaf_ids)
        HIPCHK(hipMemcpyAsync(ix->leaf_ids.p, fv->leaf_ids, fv->n_leaf_ids * 4, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    ix->n_planes = np_used;
    ix->n_leaf_ids = fv->n_leaf_ids;
    ix->planes_below_level = below;
    ix->scan_unsafe = dup_in_tree;
    return upload_nodes(ix);
}

extern "C" int zh_index_forest_sizes(zh_index *ix, zh_forest_sizes *out) {
    if (!ix || !out) return fail(ZH_EINVAL, "null argument");
    out->n_nodes = ix->n_nodes; out->n_planes = ix->n_planes; out->n_trees = ix->n_trees; out->n_leaf_ids = ix->n_leaf_ids;
    return ZH_OK;
}

extern "C" int zh_index_get_forest(zh_index *ix, int32_t *plane, int32_t *left, int32_t *right, uint32_t *roots,
                                   float *planes, float *consts, uint32_t *leaf_ids) {
    if (!ix) return fail(ZH_EINVAL, "null index");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    int rc = set_device(ix);
    if (rc) return rc;
    if (plane) memcpy(plane, ix->h_plane.data(), ix->h_plane.size() * 4);
    if (left) memcpy(left, ix->h_left.data(), ix->h_left.size() * 4);
    if (right) memcpy(right, ix->h_right.data(), ix->h_right.size() * 4);
    if (roots) memcpy(roots, ix->h_roots.data(), ix->h_roots.size() * 4);
    if (planes && ix->n_planes) HIPCHK(hipMemcpy(planes, ix->planes.p, (size_t)ix->n_planes * ix->opt.dim * 4, hipMemcpyDeviceToHost));
    if (consts && ix->n_planes) HIPCHK(hipMemcpy(consts, ix->consts.p, (size_t)ix->n_planes * 4, hipMemcpyDeviceToHost));
    if (leaf_ids && ix->n_leaf_ids) HIPCHK(hipMemcpy(leaf_ids, ix->leaf_ids.p, ix->n_leaf_ids * 4, hipMemcpyDeviceToHost));
    return ZH_OK;
}

// ------------------------------------------------------------------------------------------------
// GPU forest build (build_index, lsh.rs:411-429 -> build_a_tree lsh.rs:250-267), level-synchronous
// ------------------------------------------------------------------------------------------------
static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// two distinct rows uniform over the whole database (lsh.rs:197-201), keyed by (seed, tree, heap path)
static void sample_pair(uint64_t seed, uint32_t tree, uint64_t path, uint64_t n_rows, uint64_t *i, uint64_t *j) {
    uint64_t h = splitmix64(seed ^ splitmix64(0x7EE5ull + tree) ^ splitmix64(path * 0xC2B2AE3D27D4EB4Full));
    uint64_t h1 = splitmix64(h), h2 = splitmix64(h1);
    if (n_rows < 2) { *i = 0; *j = 0; return; }
    *i = h1 % n_rows;
    *j = h2 % (n_rows - 1);
    if (*j >= *i) (*j)++;
}

// The two sample rows of a split come from the LIVE rows among the first n_sample: LSHIndex::remove deletes the embedding from
// the store (lsh.rs:495) and build_hyperplane samples the stored embeddings only (lsh.rs:197-201), so a removed row never
// defines a plane.  Returns the live count (< 2: default zero vectors, lsh.rs:203-220).  Same mapping as the oracle's.
static uint64_t sample_live_pair(zh_index *ix, uint32_t tree, uint64_t path, uint64_t n_sample, uint64_t *si, uint64_t *sj) {
    if (ix->n_dead == 0) { sample_pair(ix->opt.seed, tree, path, n_sample, si, sj); return n_sample; }
    if (ix->h_live_rows != ix->n_rows || ix->h_live_dead != ix->n_dead || ix->h_live_gen != ix->dead_gen) {
        ix->h_live.clear();
        ix->h_live.reserve(ix->n_rows - ix->n_dead);
        for (uint64_t r = 0; r < ix->n_rows; r++)
            if (!(r < ix->h_dead.size() && ix->h_dead[r])) ix->h_live.push_back((uint32_t)r);
        ix->h_live_rows = ix->n_rows; ix->h_live_dead = ix->n_dead; ix->h_live_gen = ix->dead_gen;
    }
    const uint64_t n_live = (uint64_t)(std::lower_bound(ix->h_live.begin(), ix->h_live.end(), n_sample,
                                                        [](uint32_t a, uint64_t b) { return (uint64_t)a < b; }) - ix->h_live.begin());
    sample_pair(ix->opt.seed, tree, path, n_live, si, sj);
    if (n_live) { *si = ix->h_live[*si]; *sj = ix->h_live[*sj]; }
    return n_live;
}

struct ActiveNode {   // a node whose id list (a segment of a perm array) is to be split
    uint32_t node, tree, len, depth;
    uint64_t seg_start, path, n_sample;  // n_sample: size of the database the two sample rows are drawn from
};

// Split every active node, level after level, until all descendants are leaves (build_a_tree,
// lsh.rs:250-267).  d_perm holds the id lists as segments; leaves become runs of d_perm and get
// offset leaf_base + position.  New planes are appended to ix->planes from index n_planes on.
static int grow_segments(zh_index *ix, uint32_t *d_perm, uint64_t perm_len, std::vector<ActiveNode> active,
                         uint64_t leaf_base, uint32_t &n_planes, bool record_levels) {
    const uint32_t M = ix->opt.max_node_size, d = ix->opt.dim;
    hipStream_t s = ix->stream;
    int rc;
    DevBuf tmp, flags, dNodes, dChunks, dChunkAbove, dChunkScan, dScanTmp, dNodeAbove;
    struct Guard {
        std::vector<DevBuf *> v;
        ~Guard() { for (auto *b : v) b->release(); }
    } guard;
    guard.v = {&tmp, &flags, &dNodes, &dChunks, &dChunkAbove, &dChunkScan, &dScanTmp, &dNodeAbove};
    if (active.empty()) return ZH_OK;
    if ((rc = tmp.ensure(std::max<uint64_t>(perm_len, 1) * 4))) return rc;
    if ((rc = flags.ensure(std::max<uint64_t>(perm_len, 1)))) return rc;
    auto new_node = [&]() { ix->h_plane.push_back(-1); ix->h_left.push_back(0); ix->h_right.push_back(0); return (uint32_t)(ix->h_plane.size() - 1); };
    auto make_leaf = [&](uint32_t node, uint64_t start, uint32_t len) {
        ix->h_plane[node] = -1; ix->h_left[node] = (int32_t)(uint32_t)(leaf_base + start); ix->h_right[node] = (int32_t)len;
    };
    std::vector<ActiveNode> next;
    std::vector<ZhBuildNode> hn;
    std::vector<ZhBuildChunk> hc;
    std::vector<uint32_t> h_above;
    for (uint32_t round = 0; !active.empty(); round++) {
        hn.clear(); hc.clear();
        for (size_t i = 0; i < active.size(); i++) {
            const ActiveNode &a = active[i];
            ZhBuildNode bn;
            bn.seg_start = a.seg_start; bn.len = a.len; bn.plane = n_planes + (uint32_t)i;
            uint64_t si, sj;
            const uint64_t n_live = sample_live_pair(ix, a.tree, a.path, a.n_sample, &si, &sj);
            bn.sample_a = n_live >= 1 ? si : ~0ull;   // lsh.rs:203-220: a missing sample decodes to the zero vector
            bn.sample_b = n_live >= 2 ? sj : ~0ull;
            bn.first_chunk = (uint32_t)hc.size();
            bn.n_chunks = (a.len + 255) / 256;
            for (uint32_t c = 0; c < bn.n_chunks; c++) {
                ZhBuildChunk ch;
                ch.node = (uint32_t)i; ch.pos = a.seg_start + (uint64_t)c * 256;
                ch.count = std::min<uint32_t>(256, a.len - c * 256);
                hc.push_back(ch);
            }
            hn.push_back(bn);
        }
        const uint32_t na = (uint32_t)hn.size(), nc = (uint32_t)hc.size();
        if ((rc = ix->planes.ensure((size_t)(n_planes + na) * d * 4, true, s))) return rc;
        if ((rc = ix->consts.ensure((size_t)(n_planes + na) * 4, true, s))) return rc;
        if ((rc = dNodes.ensure(na * sizeof(ZhBuildNode)))) return rc;
        if ((rc = dChunks.ensure(std::max<uint32_t>(nc, 1) * sizeof(ZhBuildChunk)))) return rc;
        if ((rc = dChunkAbove.ensure((size_t)std::max<uint32_t>(nc, 1) * 4))) return rc;
        if ((rc = dChunkScan.ensure((size_t)(nc + 1) * 4))) return rc;
        if ((rc = dScanTmp.ensure(((size_t)nc / 1024 + 4) * 4))) return rc;
        if ((rc = dNodeAbove.ensure((size_t)na * 4))) return rc;
        {   // the planes' sample rows, for the row-score hash
            std::vector<uint2> hs(na);
            for (uint32_t i = 0; i < na; i++)
                hs[i] = make_uint2(hn[i].sample_a == ~0ull ? 0xFFFFFFFFu : (uint32_t)hn[i].sample_a,
                                   hn[i].sample_b == ~0ull ? 0xFFFFFFFFu : (uint32_t)hn[i].sample_b);
            if ((rc = ix->plane_samples.ensure((size_t)(n_planes + na) * sizeof(uint2), true, s))) return rc;
            HIPCHK(hipMemcpy(ix->plane_samples.as<uint2>() + n_planes, hs.data(), na * sizeof(uint2), hipMemcpyHostToDevice));
        }
        hipError_t e = hipMemcpyAsync(dNodes.p, hn.data(), na * sizeof(ZhBuildNode), hipMemcpyHostToDevice, s);
        if (e == hipSuccess && nc) e = hipMemcpyAsync(dChunks.p, hc.data(), nc * sizeof(ZhBuildChunk), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemsetAsync(dNodeAbove.p, 0, (size_t)na * 4, s);
        if (e == hipSuccess) e = zh_launch_make_planes(ix->X.as<float>(), d, dNodes.as<ZhBuildNode>(), na, ix->planes.as<float>(), ix->consts.as<float>(), s);
        if (e == hipSuccess) e = zh_launch_classify(ix->X.as<float>(), d, d_perm, dNodes.as<ZhBuildNode>(), dChunks.as<ZhBuildChunk>(), nc, ix->planes.as<float>(), ix->consts.as<float>(), flags.as<uint8_t>(), dChunkAbove.as<uint32_t>(), s);
        if (e == hipSuccess) e = zh_launch_scan_u32(dChunkAbove.as<uint32_t>(), dChunkScan.as<uint32_t>(), nc, dScanTmp.as<uint32_t>(), s);
        if (e == hipSuccess) e = zh_launch_scatter(d_perm, tmp.as<uint32_t>(), dNodes.as<ZhBuildNode>(), dChunks.as<ZhBuildChunk>(), nc, flags.as<uint8_t>(), dChunkScan.as<uint32_t>(), dNodeAbove.as<uint32_t>(), s);
        h_above.resize(na);
        if (e == hipSuccess) e = hipMemcpyAsync(h_above.data(), dNodeAbove.p, (size_t)na * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail(ZH_EHIP, "forest build round %u: %s", round, hipGetErrorString(e));
        next.clear();
        for (size_t i = 0; i < active.size(); i++) {
            const ActiveNode &a = active[i];
            uint32_t nA = h_above[i], nB = a.len - nA;
            uint32_t l = new_node(), r = new_node();
            ix->h_plane[a.node] = (int32_t)(n_planes + i);
            ix->h_left[a.node] = (int32_t)l;   // below  (lsh.rs:262)
            ix->h_right[a.node] = (int32_t)r;  // above  (lsh.rs:263)
            bool child_can_split = a.depth + 1 < ZH_MAX_DEPTH;
            if (nB >= M && child_can_split) next.push_back({l, a.tree, nB, a.depth + 1, a.seg_start, 2 * a.path, a.n_sample});
            else make_leaf(l, a.seg_start, nB);
            if (nA >= M && child_can_split) next.push_back({r, a.tree, nA, a.depth + 1, a.seg_start + nB, 2 * a.path + 1, a.n_sample});
            else make_leaf(r, a.seg_start + nB, nA);
        }
        n_planes += na;
        if (record_levels) ix->planes_below_level.push_back(n_planes);
        active.swap(next);
    }
    return ZH_OK;
}

static int build_forest_locked(zh_index *ix) {
    const uint64_t N = ix->n_rows;
    const uint32_t T = ix->opt.num_trees, M = ix->opt.max_node_size;
    int rc;
    hipStreamSynchronize(ix->stream);
    free_forest(ix);
    ix->broken = false;
    if ((uint64_t)T * N > 0xFFFFFFFFull) return fail(ZH_ELIMIT, "num_trees * rows exceeds 2^32-1 leaf entries; shard the index");
    const uint64_t NL = N - ix->n_dead;  // removed rows stay out of a rebuild, as leaf members and as hyperplane samples
    const uint64_t total = (uint64_t)T * NL;
    DevBuf perm;
    if ((rc = perm.ensure(std::max<uint64_t>(total, 1) * 4))) return rc;
    if (total && ix->n_dead == 0) {
        hipError_t e = zh_launch_iota_perm(perm.as<uint32_t>(), N, T, ix->stream);
        if (e != hipSuccess) { perm.release(); return fail(ZH_EHIP, "iota_perm: %s", hipGetErrorString(e)); }
    } else if (total) {
        if (ix->h_dead.size() < N) ix->h_dead.resize(N, 0);  // rows appended after the last removal are alive
        std::vector<uint32_t> live;
        live.reserve(NL);
        for (uint64_t i = 0; i < N; i++) if (!ix->h_dead[i]) live.push_back((uint32_t)i);
        for (uint32_t t = 0; t < T; t++) {
            hipError_t e = hipMemcpy(perm.as<uint32_t>() + (size_t)t * NL, live.data(), NL * 4, hipMemcpyHostToDevice);
            if (e != hipSuccess) { perm.release(); return fail(ZH_EHIP, "perm upload: %s", hipGetErrorString(e)); }
        }
    }
    std::vector<ActiveNode> active;
    ix->planes_below_level.assign(1, 0);
    for (uint32_t t = 0; t < T; t++) {
        ix->h_plane.push_back(-1); ix->h_left.push_back(0); ix->h_right.push_back(0);
        uint32_t n = (uint32_t)(ix->h_plane.size() - 1);
        ix->h_roots.push_back(n);
        if (NL < M) { ix->h_left[n] = (int32_t)(uint32_t)((uint64_t)t * NL); ix->h_right[n] = (int32_t)NL; }  // lsh.rs:251-252
        else active.push_back({n, t, (uint32_t)NL, 0, (uint64_t)t * NL, 1, N});
    }
    uint32_t n_planes = 0;
    rc = grow_segments(ix, perm.as<uint32_t>(), total, active, 0, n_planes, true);
    if (rc) { perm.release(); free_forest(ix); return rc; }
    if (n_planes == 0) {
        if ((rc = ix->planes.ensure(4 * (size_t)ix->opt.dim)) || (rc = ix->consts.ensure(4))) { perm.release(); free_forest(ix); return rc; }
    }
    ix->n_planes = n_planes;
    ix->samples_valid = true;  // every plane of this forest was made from two stored rows, recorded in plane_samples
    ix->leaf_ids = perm;  // perm is the concatenation of all leaves
    perm.p = nullptr; perm.cap = 0;
    ix->n_leaf_ids = total;
    rc = upload_nodes(ix);
    if (rc) free_forest(ix);
    return rc;
}

// LSHIndex::add on an index that already has trees (lsh.rs:445-462) for rows [n_prev, n_prev + n_new), as the
// sequential execution "one row after another, each into every tree".  A leaf's fate depends only on its own
// arrivals in id order, so the rows are processed in rounds: descend all pending (row, tree) pairs on the GPU,
// append to leaves with room, split the leaves that overflow (GPU classification through grow_segments) and
// send the arrivals that came after a split down the new subtree in the next round.
static int insert_rows_locked(zh_index *ix, uint64_t n_prev, uint64_t n_new) {
    const uint32_t T = ix->n_trees, M = ix->opt.max_node_size, d = ix->opt.dim;
    hipStream_t s = ix->stream;
    int rc;
    if (!n_new) return ZH_OK;
    if (ix->h_leaf_ids.size() != ix->n_leaf_ids) {  // host mirror of the leaf lists (bookkeeping only)
        ix->h_leaf_ids.resize(ix->n_leaf_ids);
        if (ix->n_leaf_ids) HIPCHK(hipMemcpy(ix->h_leaf_ids.data(), ix->leaf_ids.p, ix->n_leaf_ids * 4, hipMemcpyDeviceToHost));
    }
    std::vector<ZhDescend> pending;
    pending.reserve(n_new * T);
    for (uint64_t r = 0; r < n_new; r++)
        for (uint32_t t = 0; t < T; t++) pending.push_back({(uint32_t)(n_prev + r), ix->h_roots[t], 0, t, 1});
    DevBuf dItems, dTemp;
    struct G { DevBuf *a, *b; ~G() { a->release(); b->release(); } } g{&dItems, &dTemp};
    uint32_t n_planes = ix->n_planes;
    while (!pending.empty()) {
        if (pending.size() > 0x7FFFFFFFull) return fail(ZH_ELIMIT, "too many rows in one add call");
        if ((rc = dItems.ensure(pending.size() * sizeof(ZhDescend)))) return rc;
        HIPCHK(hipMemcpyAsync(dItems.p, pending.data(), pending.size() * sizeof(ZhDescend), hipMemcpyHostToDevice, s));
        HIPCHK(zh_launch_descend(forest_dev(ix), ix->X.as<float>(), d, dItems.as<ZhDescend>(), (uint32_t)pending.size(), s));
        HIPCHK(hipMemcpyAsync(pending.data(), dItems.p, pending.size() * sizeof(ZhDescend), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        std::sort(pending.begin(), pending.end(), [](const ZhDescend &a, const ZhDescend &b) {
            return a.node != b.node ? a.node < b.node : a.row < b.row;
        });
        std::vector<ZhDescend> next;
        std::vector<ActiveNode> splits;
        std::vector<uint32_t> temp;  // id lists of the nodes to split, one segment each
        const size_t old_leaf_size = ix->h_leaf_ids.size();
        for (size_t i = 0; i < pending.size();) {
            size_t j = i;
            while (j < pending.size() && pending[j].node == pending[i].node) j++;
            const ZhDescend &first = pending[i];
            const uint32_t node = first.node;
            const uint32_t off = (uint32_t)ix->h_left[node], len = (uint32_t)ix->h_right[node];
            std::vector<uint32_t> cur(ix->h_leaf_ids.begin() + off, ix->h_leaf_ids.begin() + off + len);
            size_t a = i;
            const bool can_split = first.depth < ZH_MAX_DEPTH;
            while (a < j && (cur.size() + 1 <= M || !can_split)) cur.push_back(pending[a++].row);  // lsh.rs:368-369
            if (a == j) {  // the leaf just grew: relocate its run to the end of leaf_ids
                ix->h_left[node] = (int32_t)(uint32_t)ix->h_leaf_ids.size();
                ix->h_right[node] = (int32_t)cur.size();
                ix->h_leaf_ids.insert(ix->h_leaf_ids.end(), cur.begin(), cur.end());
            } else {       // lsh.rs:370-377: this arrival overflows the leaf -> build_a_tree(leaf ids + id)
                cur.push_back(pending[a].row);
                splits.push_back({node, first.tree, (uint32_t)cur.size(), first.depth, (uint64_t)temp.size(), first.path,
                                  (uint64_t)pending[a].row + 1});
                temp.insert(temp.end(), cur.begin(), cur.end());
                for (size_t r = a + 1; r < j; r++) next.push_back(pending[r]);  // they arrive after the split
            }
            i = j;
        }
        if (ix->h_leaf_ids.size() + temp.size() > 0xFFFFFFFFull) return fail(ZH_ELIMIT, "leaf storage exceeds 2^32-1 entries; rebuild the index");
        if (!splits.empty()) {
            if ((rc = dTemp.ensure(temp.size() * 4))) return rc;
            HIPCHK(hipMemcpyAsync(dTemp.p, temp.data(), temp.size() * 4, hipMemcpyHostToDevice, s));
            const uint64_t leaf_base = ix->h_leaf_ids.size();
            if ((rc = grow_segments(ix, dTemp.as<uint32_t>(), temp.size(), splits, leaf_base, n_planes, false))) return rc;
            HIPCHK(hipMemcpy(temp.data(), dTemp.p, temp.size() * 4, hipMemcpyDeviceToHost));
            ix->h_leaf_ids.insert(ix->h_leaf_ids.end(), temp.begin(), temp.end());
        }
        // mirror -> device (appended tail only), nodes -> device
        if (ix->h_leaf_ids.size() > old_leaf_size) {
            if ((rc = ix->leaf_ids.ensure(ix->h_leaf_ids.size() * 4, true, s))) return rc;
            HIPCHK(hipMemcpyAsync(ix->leaf_ids.as<uint32_t>() + old_leaf_size, ix->h_leaf_ids.data() + old_leaf_size,
                                  (ix->h_leaf_ids.size() - old_leaf_size) * 4, hipMemcpyHostToDevice, s));
        }
        ix->n_leaf_ids = ix->h_leaf_ids.size();
        ix->n_planes = n_planes;
        if ((rc = upload_nodes(ix))) return rc;
        pending.swap(next);
    }
    // compaction: relocated runs leave dead entries behind; rewrite the lists when they dominate
    const uint64_t live = (uint64_t)T * ix->n_rows;
    if (ix->h_leaf_ids.size() > 2 * live + 1024) {
        std::vector<uint32_t> compact;
        compact.reserve(live);
        for (size_t n = 0; n < ix->h_plane.size(); n++)
            if (ix->h_plane[n] < 0) {
                uint32_t off = (uint32_t)ix->h_left[n], len = (uint32_t)ix->h_right[n];
                ix->h_left[n] = (int32_t)(uint32_t)compact.size();
                compact.insert(compact.end(), ix->h_leaf_ids.begin() + off, ix->h_leaf_ids.begin() + off + len);
            }
        ix->h_leaf_ids.swap(compact);
        ix->n_leaf_ids = ix->h_leaf_ids.size();
        if (ix->n_leaf_ids) HIPCHK(hipMemcpyAsync(ix->leaf_ids.p, ix->h_leaf_ids.data(), ix->n_leaf_ids * 4, hipMemcpyHostToDevice, s));
        if ((rc = upload_nodes(ix))) return rc;
    }
    return ZH_OK;
}

extern "C" int zh_index_build(zh_index *ix) {
    if (!ix) return fail(ZH_EINVAL, "null index");
    if (ix->opt.max_node_size == 0) return fail(ZH_EINVAL, "max_node_size must be >= 1");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    int rc = set_device(ix);
    if (rc) return rc;
    return build_forest_locked(ix);
}

extern "C" int zh_index_add(zh_index *ix, const float *rows, size_t n, uint64_t *out_ids) {
    if (!ix || (!rows && n)) return fail(ZH_EINVAL, "zh_index_add: null argument");
    if (ix->opt.max_node_size == 0) return fail(ZH_EINVAL, "max_node_size must be >= 1");
    std::unique_lock<std::shared_mutex> lk(ix->mu);  // one critical section: state is read, rows stored and trees updated under it
    if (ix->broken) return fail(ZH_ESTATE, "an earlier add failed half way: call zh_index_build before adding or searching");
    const bool had_trees = ix->n_trees != 0;  // lsh.rs:441: no_trees() decides between build_index and insert
    const uint64_t n_prev = ix->n_rows;
    // capacity checks BEFORE anything is mutated: the rows, and (incremental path) the leaf entries of the new rows
    if (n_prev + n > 0xFFFFFFFFull) return fail(ZH_ELIMIT, "more than 2^32-1 rows in one index (shard it)");
    if ((uint64_t)ix->opt.num_trees * (n_prev + n) > 0xFFFFFFFFull)
        return fail(ZH_ELIMIT, "num_trees * rows exceeds 2^32-1 leaf entries; shard the index");
    if (had_trees && n * (uint64_t)ix->n_trees > 0x7FFFFFFFull) return fail(ZH_ELIMIT, "too many rows in one add call");
    int rc = append_locked(ix, rows, n, out_ids);
    if (rc) return rc;  // nothing changed
    if (!had_trees) return build_forest_locked(ix);  // lsh.rs:441-443; on failure: rows stored, no trees (a consistent state)
    rc = insert_rows_locked(ix, n_prev, n);          // lsh.rs:445-462
    if (rc) {
        // the host mirrors of the trees may be ahead of the device copy: the rows stay stored (their ids were handed out),
        // the forest is declared stale until zh_index_build rebuilds it; searches say so instead of missing rows silently
        ix->broken = true;
    }
    return rc;
}

// LSHIndex::remove (lsh.rs:473-503) as it is meant: the reference only edits trees whose root is a leaf, so an id
// stays in every inner tree while its embedding is gone.  Here every tree drops the id: the row's
// leaf is found on the GPU by descending with the row's own vector, the leaf's run shrinks in place.
static int remove_rows_locked(zh_index *ix, const std::vector<uint32_t> &rows, std::vector<uint8_t> &found) {
    const uint32_t T = ix->n_trees, d = ix->opt.dim;
    hipStream_t s = ix->stream;
    int rc;
    found.assign(rows.size(), 0);
    if (rows.empty()) return ZH_OK;
    if (ix->h_dead.size() < ix->n_rows) ix->h_dead.resize(ix->n_rows, 0);
    if (T) {
        if (ix->h_leaf_ids.size() != ix->n_leaf_ids) {
            ix->h_leaf_ids.resize(ix->n_leaf_ids);
            if (ix->n_leaf_ids) HIPCHK(hipMemcpy(ix->h_leaf_ids.data(), ix->leaf_ids.p, ix->n_leaf_ids * 4, hipMemcpyDeviceToHost));
        }
        std::vector<ZhDescend> items;
        items.reserve(rows.size() * T);
        for (size_t r = 0; r < rows.size(); r++)
            for (uint32_t t = 0; t < T; t++) items.push_back({rows[r], ix->h_roots[t], (uint32_t)r, t, 1});  // depth field carries r
        DevBuf dItems;
        struct G { DevBuf *a; ~G() { a->release(); } } g{&dItems};
        if ((rc = dItems.ensure(items.size() * sizeof(ZhDescend)))) return rc;
        HIPCHK(hipMemcpyAsync(dItems.p, items.data(), items.size() * sizeof(ZhDescend), hipMemcpyHostToDevice, s));
        std::vector<uint32_t> slot(items.size());
        for (size_t i = 0; i < items.size(); i++) slot[i] = items[i].depth;
        HIPCHK(zh_launch_descend(forest_dev(ix), ix->X.as<float>(), d, dItems.as<ZhDescend>(), (uint32_t)items.size(), s));
        HIPCHK(hipMemcpyAsync(items.data(), dItems.p, items.size() * sizeof(ZhDescend), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        auto drop = [&](uint32_t node, uint32_t id) {  // remove id from a leaf's run (in place), true if it was there
            uint32_t off = (uint32_t)ix->h_left[node], len = (uint32_t)ix->h_right[node];
            uint32_t *run = ix->h_leaf_ids.data() + off;
            for (uint32_t i = 0; i < len; i++)
                if (run[i] == id) {
                    memmove(run + i, run + i + 1, (len - i - 1) * sizeof(uint32_t));
                    ix->h_right[node] = (int32_t)(len - 1);
                    return true;
                }
            return false;
        };
        std::vector<uint32_t> touched;
        for (size_t i = 0; i < items.size(); i++) {
            const uint32_t r = slot[i], id = items[i].row, t = items[i].tree;
            uint32_t node = items[i].node;
            bool hit = drop(node, id);
            if (!hit) {  // an injected forest may keep the id elsewhere: scan this tree's leaves
                std::vector<uint32_t> st{ix->h_roots[t]};
                while (!st.empty() && !hit) {
                    uint32_t m = st.back(); st.pop_back();
                    if (ix->h_plane[m] < 0) { if (drop(m, id)) { hit = true; node = m; } }
                    else { st.push_back((uint32_t)ix->h_left[m]); st.push_back((uint32_t)ix->h_right[m]); }
                }
            }
            if (hit) { found[r] = 1; touched.push_back(node); }
        }
        std::sort(touched.begin(), touched.end());
        touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
        for (uint32_t node : touched) {  // shrunken runs back to the device
            uint32_t off = (uint32_t)ix->h_left[node], len = (uint32_t)ix->h_right[node];
            if (len) HIPCHK(hipMemcpyAsync(ix->leaf_ids.as<uint32_t>() + off, ix->h_leaf_ids.data() + off, (size_t)len * 4, hipMemcpyHostToDevice, s));
        }
        if ((rc = upload_nodes(ix))) return rc;
    } else {
        for (size_t r = 0; r < rows.size(); r++) found[r] = 1;  // vectors without trees: just forget them
    }
    for (size_t r = 0; r < rows.size(); r++)
        if (found[r] && !ix->h_dead[rows[r]]) { ix->h_dead[rows[r]] = 1; ix->n_dead++; ix->dead_gen++; }
    return ZH_OK;
}

extern "C" int zh_index_remove(zh_index *ix, const uint64_t *ids, size_t n, uint8_t *out_found, size_t *out_n_removed) {
    if (!ix || (n && !ids)) return fail(ZH_EINVAL, "zh_index_remove: null argument");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    int rc = set_device(ix);
    if (rc) return rc;
    std::vector<uint32_t> rows;
    std::vector<size_t> where;
    std::vector<uint8_t> seen(ix->n_rows, 0);
    for (size_t i = 0; i < n; i++) {
        if (out_found) out_found[i] = 0;
        if (ids[i] < ix->opt.id_base) continue;
        uint64_t r = ids[i] - ix->opt.id_base;
        if (r >= ix->n_rows || seen[r] || (r < ix->h_dead.size() && ix->h_dead[r])) continue;
        seen[r] = 1;
        rows.push_back((uint32_t)r);
        where.push_back(i);
    }
    std::vector<uint8_t> found;
    if ((rc = remove_rows_locked(ix, rows, found))) return rc;
    size_t cnt = 0;
    for (size_t j = 0; j < rows.size(); j++)
        if (found[j]) { cnt++; if (out_found) out_found[where[j]] = 1; }
    if (out_n_removed) *out_n_removed = cnt;
    return ZH_OK;
}

// LSHIndex::deduplicate (lsh.rs:270-288): every live row whose f32 bit pattern equals an earlier live row's is
// removed.  Rows are hashed on the GPU (one pass over the stored vectors), equal hashes are confirmed byte for byte.
extern "C" int zh_index_deduplicate(zh_index *ix, uint64_t *out_ids, size_t cap, size_t *out_n_removed) {
    if (!ix) return fail(ZH_EINVAL, "null index");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    int rc = set_device(ix);
    if (rc) return rc;
    const uint64_t N = ix->n_rows;
    const uint32_t d = ix->opt.dim;
    if (out_n_removed) *out_n_removed = 0;
    if (N < 2) return ZH_OK;
    DevBuf dh;
    struct G { DevBuf *a; ~G() { a->release(); } } g{&dh};
    if ((rc = dh.ensure(N * 8))) return rc;
    HIPCHK(zh_launch_row_hash(ix->X.as<float>(), N, d, dh.as<uint64_t>(), ix->stream));
    std::vector<uint64_t> hh(N);
    HIPCHK(hipMemcpyAsync(hh.data(), dh.p, N * 8, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    std::vector<std::pair<uint64_t, uint32_t>> hv;
    hv.reserve(N - ix->n_dead);
    for (uint64_t i = 0; i < N; i++)
        if (!(i < ix->h_dead.size() && ix->h_dead[i])) hv.push_back({hh[i], (uint32_t)i});
    std::sort(hv.begin(), hv.end());
    // runs of equal hash: every member is compared (on the GPU, bit for bit) with the run's current representative;
    // members that differ from it -- a hash collision -- form the next round against the next representative
    std::vector<uint32_t> dups;
    std::vector<std::vector<uint32_t>> runs;
    for (size_t i = 0; i < hv.size();) {
        size_t j = i + 1;
        while (j < hv.size() && hv[j].first == hv[i].first) j++;
        if (j - i > 1) {
            std::vector<uint32_t> r;
            for (size_t p = i; p < j; p++) r.push_back(hv[p].second);  // ascending ids (sorted by (hash, id))
            runs.push_back(std::move(r));
        }
        i = j;
    }
    DevBuf dPairs, dEq;
    struct G2 { DevBuf *a, *b; ~G2() { a->release(); b->release(); } } g2{&dPairs, &dEq};
    while (!runs.empty()) {
        std::vector<uint32_t> pairs;
        for (auto &r : runs)
            for (size_t p = 1; p < r.size(); p++) { pairs.push_back(r[0]); pairs.push_back(r[p]); }
        const uint32_t np = (uint32_t)(pairs.size() / 2);
        std::vector<uint8_t> eq(np);
        if ((rc = dPairs.ensure(pairs.size() * 4)) || (rc = dEq.ensure(np))) return rc;
        HIPCHK(hipMemcpyAsync(dPairs.p, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice, ix->stream));
        HIPCHK(zh_launch_rows_equal(ix->X.as<float>(), d, dPairs.as<uint32_t>(), np, dEq.as<uint8_t>(), ix->stream));
        HIPCHK(hipMemcpyAsync(eq.data(), dEq.p, np, hipMemcpyDeviceToHost, ix->stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
        std::vector<std::vector<uint32_t>> next;
        size_t q = 0;
        for (auto &r : runs) {
            std::vector<uint32_t> rest;
            for (size_t p = 1; p < r.size(); p++, q++) {
                if (eq[q]) dups.push_back(r[p]); else rest.push_back(r[p]);
            }
            if (rest.size() > 1) next.push_back(std::move(rest));
        }
        runs.swap(next);
    }
    std::sort(dups.begin(), dups.end());
    std::vector<uint8_t> found;
    if ((rc = remove_rows_locked(ix, dups, found))) return rc;
    size_t cnt = 0;
    for (size_t i = 0; i < dups.size(); i++) {
        if (out_ids && cnt < cap) out_ids[cnt] = ix->opt.id_base + dups[i];
        cnt++;
    }
    if (out_n_removed) *out_n_removed = cnt;
    return ZH_OK;
}

// ------------------------------------------------------------------------------------------------
// search
// ------------------------------------------------------------------------------------------------
extern "C" int zh_set_profiling(zh_index *ix, int level) {
    if (!ix) return fail(ZH_EINVAL, "null index");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    int rc = set_device(ix);
    if (rc) return rc;
    ix->profiling = level;
    return ZH_OK;
}
extern "C" int zh_set_dense_levels(zh_index *ix, int levels) {
    if (!ix) return fail(ZH_EINVAL, "null index");
    ix->dense_levels = levels;
    return ZH_OK;
}
extern "C" int zh_set_hash_mode(zh_index *ix, int mode) {
    if (!ix) return fail(ZH_EINVAL, "null index");
    if (mode < 0 || mode > 2) return fail(ZH_EINVAL, "hash mode %d (0 = choose per batch, 1 = one dot product per plane, 2 = row scores)", mode);
    ix->hash_mode = mode;
    return ZH_OK;
}
extern "C" int zh_set_sweep_mode(zh_index *ix, int mode) {
    if (!ix) return fail(ZH_EINVAL, "null index");
    if (mode < 0 || mode > 6)
        return fail(ZH_EINVAL, "sweep mode %d (0 = choose per batch, 1 = leaf by leaf, 2 = table scan with f32 queries, 3 = as 0: prefilter where row scores exist, "
                               "4 = table scan with half-width queries wherever it applies, 5 = as 4 without the fp16 copy of the rows: the VALU kernel, "
                               "6 = leaf by leaf at half width where that kernel exists (dim 128))", mode);
    ix->sweep_mode = mode;
    return ZH_OK;
}
extern "C" int zh_stats(zh_index *ix, zh_stats_t *out) {
    if (!ix || !out) return fail(ZH_EINVAL, "null argument");
    {
        std::lock_guard<std::mutex> lk(ix->stats_mu);
        *out = ix->stats;
    }
    std::lock_guard<std::mutex> lk(ix->blk_mu);  // (the copies are made and dropped under it)
    out->row_copy_bytes = ix->row_half.cap + ix->row_meta.cap + ix->row_half128.cap + ix->scan_perm.cap + ix->row_leaf_p.cap;  // (+ the scan's row order and its view of row -> leaf)
    out->scan_order_keys = ix->perm_rows ? ix->order_keys : 0;
    out->scan_order_share_permille = (uint64_t)(ix->order_worth * 1000.0 + 0.5);
    return ZH_OK;
}
extern "C" int zh_stats_reset(zh_index *ix) {
    if (!ix) return fail(ZH_EINVAL, "null index");
    std::lock_guard<std::mutex> lk(ix->stats_mu);
    memset(&ix->stats, 0, sizeof ix->stats);
    return ZH_OK;
}

// Blocked view of the forest (zh_internal.h, ZhBlocksDev): every maximal subtree of at most ZH_BLOCK_NODES nodes becomes
// one block with its nodes in pre-order; the nodes above keep pointer records whose child refs say "upper node" or
// "block".  Host pass over the node mirrors (iterative: trees may be 60 levels deep), then four uploads.
// Round 5: blocks of INNER nodes only.  A block of the round-2 view spends half of its 64 records on leaves; here a block is a maximal subtree of at
// most ZH_BLOCK_INNER inner nodes, one 32-byte record per inner node in pre-order -- {plane, code of the left child | code of the right child << 8 |
// inner nodes of the block << 16 (root record), left leaf: offset into leaf_ids, length} {left leaf: node id, right leaf: offset, length, node id} --
// and its leaves live in their parent's record: a child code is the child's record (0 .. 62) or 0x80 | side << 6 | the parent's record for a leaf.
// Twice the tree nodes per block: half the blocks, half the upper nodes, and the walk's dependent round trips with them (1M x 384 at the
// reference's default options: 346k blocks / 346k upper nodes -> see tests/probes/refdefault_shape.py).  A leaf hanging directly off an upper node
// (or a tree that is one leaf) is a block of zero inner nodes: record {-1, 0, offset, length} {node id, 0, 0, 0}.
static int build_blocks_inner(zh_index *ix) {
    const size_t nn = ix->h_plane.size();
    const uint32_t T = (uint32_t)ix->h_roots.size();
    std::vector<uint32_t> isize(nn, 0), order, st;
    order.reserve(nn);
    for (uint32_t t = 0; t < T; t++) {  // pre-order of every tree; inner-node counts of the subtrees in a reverse pass
        const size_t o0 = order.size();
        st.assign(1, ix->h_roots[t]);
        while (!st.empty()) {
            const uint32_t n = st.back(); st.pop_back();
            order.push_back(n);
            if (ix->h_plane[n] >= 0) { st.push_back((uint32_t)ix->h_right[n]); st.push_back((uint32_t)ix->h_left[n]); }
        }
        for (size_t i = order.size(); i-- > o0;) {
            const uint32_t n = order[i];
            isize[n] = ix->h_plane[n] >= 0 ? 1 + isize[(uint32_t)ix->h_left[n]] + isize[(uint32_t)ix->h_right[n]] : 0;
        }
    }
    std::vector<int4> recs, recsB, upper;  // recs / recsB: the first / second int4 of every inner node's record (two arrays: a wave's two loads of a
                                           // block are 1 KiB contiguous each -- interleaved 32-byte records made a block load 3404 cycles against 1795)
    std::vector<int2> roots(std::max<uint32_t>(T, 1), make_int2(0, 0));
    recs.reserve(nn / 2 + ZH_BLOCK_NODES);
    recsB.reserve(nn / 2 + ZH_BLOCK_NODES);
    std::vector<uint32_t> nodes;
    uint32_t n_blocks = 0;
    auto emit_block = [&](uint32_t root) -> int32_t {  // the subtree's INNER nodes in pre-order; a ref is -(first 32-byte record + 1)
        const size_t base = recs.size();
        n_blocks++;
        if (ix->h_plane[root] < 0) {  // a leaf by itself
            recs.push_back(make_int4(-1, 0, ix->h_left[root], ix->h_right[root]));
            recsB.push_back(make_int4((int)root, 0, 0, 0));
            return -(int32_t)base - 1;
        }
        nodes.clear();
        st.assign(1, root);
        while (!st.empty()) {
            const uint32_t n = st.back(); st.pop_back();
            nodes.push_back(n);
            const uint32_t l = (uint32_t)ix->h_left[n], r = (uint32_t)ix->h_right[n];
            if (ix->h_plane[r] >= 0) st.push_back(r);
            if (ix->h_plane[l] >= 0) st.push_back(l);
        }
        // inner pre-order: the left child (when inner) is i + 1, the right child i + 1 + (inner nodes of the left subtree)
        for (uint32_t i = 0; i < nodes.size(); i++) {
            const uint32_t n = nodes[i], l = (uint32_t)ix->h_left[n], r = (uint32_t)ix->h_right[n];
            const bool li = ix->h_plane[l] >= 0, ri = ix->h_plane[r] >= 0;
            const uint32_t cl = li ? i + 1 : (0x80u | i), cr = ri ? i + 1 + isize[l] : (0x80u | 0x40u | i);
            const uint32_t word = cl | (cr << 8) | (i == 0 ? (uint32_t)nodes.size() << 16 : 0u);
            recs.push_back(make_int4(ix->h_plane[n], (int)word, li ? 0 : ix->h_left[l], li ? 0 : ix->h_right[l]));
            recsB.push_back(make_int4(li ? 0 : (int)l, ri ? 0 : ix->h_left[r], ri ? 0 : ix->h_right[r], ri ? 0 : (int)r));
        }
        return -(int32_t)base - 1;
    };
    std::vector<std::pair<uint32_t, uint32_t>> todo;  // (node, its upper id)
    for (uint32_t t = 0; t < T; t++) {
        const uint32_t rt = ix->h_roots[t];
        if (isize[rt] <= ZH_BLOCK_INNER) { roots[t] = make_int2(emit_block(rt), 0); continue; }
        roots[t] = make_int2((int32_t)(upper.size() / 2), ix->h_plane[rt]);
        upper.push_back(make_int4(0, 0, 0, 0)); upper.push_back(make_int4(0, 0, 0, 0));
        todo.assign(1, {rt, (uint32_t)(upper.size() / 2 - 1)});
        while (!todo.empty()) {
            const auto [n, uid] = todo.back(); todo.pop_back();
            const uint32_t ch[2] = {(uint32_t)ix->h_left[n], (uint32_t)ix->h_right[n]};
            int32_t ref[2], cpl[2];
            for (int e = 0; e < 2; e++) {
                if (isize[ch[e]] <= ZH_BLOCK_INNER) { ref[e] = emit_block(ch[e]); cpl[e] = 0; }
                else {
                    ref[e] = (int32_t)(upper.size() / 2); cpl[e] = ix->h_plane[ch[e]];
                    upper.push_back(make_int4(0, 0, 0, 0)); upper.push_back(make_int4(0, 0, 0, 0));
                    todo.push_back({ch[e], (uint32_t)ref[e]});
                }
            }
            upper[2 * (size_t)uid] = make_int4(ix->h_plane[n], ref[0], ref[1], 0);
            upper[2 * (size_t)uid + 1] = make_int4(cpl[0], cpl[1], 0, 0);
        }
    }
    if (recs.size() > 0x7FFFFFF0ull || upper.size() > 0x7FFFFFF0ull) return fail(ZH_ELIMIT, "forest too large for the blocked view");
    recs.resize(recs.size() + ZH_BLOCK_NODES, make_int4(-1, 0, 0, 0));  // a wave always reads 64 records
    recsB.resize(recs.size(), make_int4(0, 0, 0, 0));
    const size_t n_inner_recs = recs.size();
    recs.insert(recs.end(), recsB.begin(), recsB.end());  // one buffer: [first halves][second halves]
    std::vector<int4>().swap(recsB);
    if (upper.empty()) upper.assign(2, make_int4(0, 0, 0, 0));
    int rc;
    if ((rc = ix->blk_recs.ensure(recs.size() * sizeof(int4)))) return rc;
    if ((rc = ix->blk_upper.ensure(upper.size() * sizeof(int4)))) return rc;
    if ((rc = ix->blk_roots.ensure(roots.size() * sizeof(int2)))) return rc;
    HIPCHK(hipMemcpy(ix->blk_recs.p, recs.data(), recs.size() * sizeof(int4), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->blk_upper.p, upper.data(), upper.size() * sizeof(int4), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->blk_roots.p, roots.data(), roots.size() * sizeof(int2), hipMemcpyHostToDevice));
    ix->n_blocks = n_blocks;
    ix->n_upper = (uint32_t)(upper.size() / 2);
    ix->blocks_valid = true;
    ix->blocks_inner = true;
    ix->blk_recs_b = n_inner_recs;
    if (getenv("ZH_DEBUG_BLOCKS")) fprintf(stderr, "zebra_hip: blocked view (inner-node blocks): %u blocks, %u upper nodes, %zu inner records\n", n_blocks, ix->n_upper, n_inner_recs);
    return ZH_OK;
}

static int build_blocks(zh_index *ix) {
    // ZH_WALK_BLOCKS=inner (read when a forest's view is built): the round-5 blocks of inner nodes only.  Measured and NOT the default: half the
    // blocks and upper nodes, 27 % fewer block entries and 31 % fewer upper steps per pair -- and the same 8.5 ms per batch (DESIGN.md s9,
    // profiles/r05_walk_prof.txt): the walk's time is its instruction stream and the exact sign chains, not its round trips.
    const char *bv = getenv("ZH_WALK_BLOCKS");
    if (bv && bv[0] == 'i') return build_blocks_inner(ix);
    ix->blocks_inner = false;
    const size_t nn = ix->h_plane.size();
    const uint32_t T = (uint32_t)ix->h_roots.size();
    std::vector<uint32_t> size(nn, 0), order;
    order.reserve(nn);
    std::vector<uint32_t> st;
    for (uint32_t t = 0; t < T; t++) {  // pre-order of every tree; subtree sizes in a reverse pass
        const size_t o0 = order.size();
        st.assign(1, ix->h_roots[t]);
        while (!st.empty()) {
            const uint32_t n = st.back(); st.pop_back();
            order.push_back(n);
            if (ix->h_plane[n] >= 0) { st.push_back((uint32_t)ix->h_right[n]); st.push_back((uint32_t)ix->h_left[n]); }
        }
        for (size_t i = order.size(); i-- > o0;) {
            const uint32_t n = order[i];
            size[n] = 1 + (ix->h_plane[n] >= 0 ? size[(uint32_t)ix->h_left[n]] + size[(uint32_t)ix->h_right[n]] : 0);
        }
    }
    std::vector<int4> recs, upper;
    std::vector<int2> roots(std::max<uint32_t>(T, 1), make_int2(0, 0));
    recs.reserve(nn + ZH_BLOCK_NODES);
    std::vector<uint32_t> nodes;
    uint32_t n_blocks = 0;
    auto emit_block = [&](uint32_t root) -> int32_t {  // nodes of the subtree in pre-order, children as local indices
        const size_t base = recs.size();
        nodes.clear();
        st.assign(1, root);
        while (!st.empty()) {
            const uint32_t n = st.back(); st.pop_back();
            nodes.push_back(n);
            if (ix->h_plane[n] >= 0) { st.push_back((uint32_t)ix->h_right[n]); st.push_back((uint32_t)ix->h_left[n]); }
        }
        // in pre-order the left child of local i is i + 1 and the right child i + 1 + size(left)
        for (uint32_t i = 0; i < nodes.size(); i++) {
            const uint32_t n = nodes[i];
            if (ix->h_plane[n] >= 0) {
                const uint32_t l = i + 1, r = i + 1 + size[(uint32_t)ix->h_left[n]];
                recs.push_back(make_int4(ix->h_plane[n], (int)(l | (r << 16)), i == 0 ? (int)nodes.size() : 0, (int)n));
            } else
                recs.push_back(make_int4(-1, ix->h_left[n], ix->h_right[n], (int)n));
        }
        n_blocks++;
        return -(int32_t)base - 1;
    };
    // upper nodes get dense ids in the order they are met; a child's ref is final once the child has been classified
    std::vector<uint32_t> up;
    std::vector<std::pair<uint32_t, uint32_t>> todo;  // (node, its upper id)
    for (uint32_t t = 0; t < T; t++) {
        const uint32_t rt = ix->h_roots[t];
        if (size[rt] <= ZH_BLOCK_NODES) { roots[t] = make_int2(emit_block(rt), 0); continue; }
        roots[t] = make_int2((int32_t)(upper.size() / 2), ix->h_plane[rt]);
        upper.push_back(make_int4(0, 0, 0, 0)); upper.push_back(make_int4(0, 0, 0, 0));
        todo.assign(1, {rt, (uint32_t)(upper.size() / 2 - 1)});
        while (!todo.empty()) {
            const auto [n, uid] = todo.back(); todo.pop_back();
            const uint32_t ch[2] = {(uint32_t)ix->h_left[n], (uint32_t)ix->h_right[n]};
            int32_t ref[2], cpl[2];
            for (int e = 0; e < 2; e++) {
                if (size[ch[e]] <= ZH_BLOCK_NODES) { ref[e] = emit_block(ch[e]); cpl[e] = 0; }
                else {
                    ref[e] = (int32_t)(upper.size() / 2); cpl[e] = ix->h_plane[ch[e]];
                    upper.push_back(make_int4(0, 0, 0, 0)); upper.push_back(make_int4(0, 0, 0, 0));
                    todo.push_back({ch[e], (uint32_t)ref[e]});
                }
            }
            upper[2 * (size_t)uid] = make_int4(ix->h_plane[n], ref[0], ref[1], 0);
            upper[2 * (size_t)uid + 1] = make_int4(cpl[0], cpl[1], 0, 0);
        }
    }
    if (recs.size() > 0x7FFFFFF0ull || upper.size() > 0x7FFFFFF0ull) return fail(ZH_ELIMIT, "forest too large for the blocked view");
    const size_t n_recs = recs.size();
    recs.resize(n_recs + ZH_BLOCK_NODES, make_int4(-1, 0, 0, 0));  // a wave always reads 64 records
    if (upper.empty()) upper.assign(2, make_int4(0, 0, 0, 0));
    int rc;
    if ((rc = ix->blk_recs.ensure(recs.size() * sizeof(int4)))) return rc;
    if ((rc = ix->blk_upper.ensure(upper.size() * sizeof(int4)))) return rc;
    if ((rc = ix->blk_roots.ensure(roots.size() * sizeof(int2)))) return rc;
    HIPCHK(hipMemcpy(ix->blk_recs.p, recs.data(), recs.size() * sizeof(int4), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->blk_upper.p, upper.data(), upper.size() * sizeof(int4), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->blk_roots.p, roots.data(), roots.size() * sizeof(int2), hipMemcpyHostToDevice));
    ix->n_blocks = n_blocks;
    ix->n_upper = (uint32_t)(upper.size() / 2);
    ix->blocks_valid = true;
    if (getenv("ZH_DEBUG_BLOCKS")) fprintf(stderr, "zebra_hip: blocked view (round-2 blocks): %u blocks, %u upper nodes\n", n_blocks, ix->n_upper);
    return ZH_OK;
}

// row -> {leaf node, position in the leaf} for every tree (the table-scan sweep's view of the forest): host pass for the
// node -> tree map (iterative, as build_blocks), then one kernel over the leaves.  A failed allocation is remembered: the
// leaf-major sweep serves the index until the trees change.
static int build_row_leaf(zh_index *ix) {
    const size_t nn = ix->h_plane.size();
    const uint32_t T = (uint32_t)ix->h_roots.size();
    if (ix->row_leaf.ensure(std::max<uint64_t>(ix->n_rows * T, 1) * sizeof(uint2)) != ZH_OK) { ix->row_leaf_failed = true; return ZH_OK; }
    std::vector<uint32_t> node_tree(std::max<size_t>(nn, 1), 0xFFFFFFFFu), st;
    for (uint32_t t = 0; t < T; t++) {
        st.assign(1, ix->h_roots[t]);
        while (!st.empty()) {
            const uint32_t n = st.back(); st.pop_back();
            node_tree[n] = t;
            if (ix->h_plane[n] >= 0) { st.push_back((uint32_t)ix->h_right[n]); st.push_back((uint32_t)ix->h_left[n]); }
        }
    }
    DevBuf dTree;
    int rc = dTree.ensure(node_tree.size() * 4);
    if (rc) { ix->row_leaf_failed = true; return ZH_OK; }
    hipError_t e = hipMemcpy(dTree.p, node_tree.data(), node_tree.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = zh_launch_row_leaf(ix->node_pack.as<int4>(), dTree.as<uint32_t>(), (uint32_t)nn, ix->leaf_ids.as<uint32_t>(), T, ix->n_rows,
                               ix->row_leaf.as<uint2>(), ix->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    dTree.release();
    if (e != hipSuccess) return fail(ZH_EHIP, "row -> leaf table: %s", hipGetErrorString(e));
    ix->row_leaf_valid = true;
    ix->row_leaf_rows = ix->n_rows;
    ix->row_leaf_gen++;
    return ZH_OK;
}

// the matrix-core scan's row order (under blk_mu; zh_order.hip), from the row -> leaf table, which must be valid for the stored rows.  Candidates:
// id order (rows inserted cluster by cluster are best left alone), sorted by (leaf in tree 0, leaf in tree 1, id), and by three trees' leaves; each
// is MEASURED -- (adjacent positions, tree) combinations that share a leaf = visitors a tile fetches once -- and the best one kept.
// perm_rows = 0: id order (it won, there is no table, the order is switched off, or there is no room for the sort).  ZH_ROW_ORDER=0|2|3 forces one.
static int build_scan_perm(zh_index *ix) {
    const char *env_o = getenv("ZH_ROW_ORDER");  // (read when a copy is made: tests and A/B switch it)
    const int forced = env_o ? atoi(env_o) : (getenv("ZH_NO_ROW_ORDER") ? 0 : -1);
    ix->perm_rows = 0;
    ix->perm_gen++;
    ix->order_keys = 0;
    if (forced == 0 || ix->row_order_off || !ix->row_leaf_valid || ix->row_leaf_rows != ix->n_rows || ix->n_rows < 32 || ix->n_rows > 0x7FFFFFF0ull) return ZH_OK;
    size_t mem_free = 0, mem_total = 0;  // the sort's scratch: ~28 bytes per row, released before the copy is made
    if (zh_mem_info(&mem_free, &mem_total) != hipSuccess || mem_free < ix->n_rows * 48 + mem_total / 16) return ZH_OK;
    DevBuf cand, dsum;
    if (ix->scan_perm.ensure(ix->n_rows * 4) || cand.ensure(ix->n_rows * 4) || dsum.ensure(8)) { cand.release(); dsum.release(); return ZH_OK; }
    const uint2 *rl = ix->row_leaf.as<uint2>();
    const uint32_t T = ix->n_trees;
    auto worth = [&](const uint32_t *perm, unsigned long long *out) -> hipError_t {
        hipError_t e = zh_launch_order_agreement(rl, perm, ix->n_rows, T, dsum.as<unsigned long long>(), ix->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out, dsum.p, 8, hipMemcpyDeviceToHost, ix->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
        return e;
    };
    unsigned long long best = 0, w = 0;
    hipError_t e = forced > 0 ? hipSuccess : worth(nullptr, &best);
    uint32_t best_keys = 0;
    for (uint32_t keys = 2; keys <= 3 && e == hipSuccess; keys++) {
        if (forced > 0 && (uint32_t)forced != keys) continue;
        if ((e = zh_launch_scan_order(rl, ix->n_rows, T, keys, cand.as<uint32_t>(), ix->stream)) != hipSuccess) break;
        if ((e = worth(cand.as<uint32_t>(), &w)) != hipSuccess) break;
        if (forced > 0 || w > best + best / 50) {  // (an order must be worth 2 % more than what there is)
            best = w; best_keys = keys;
            e = hipMemcpyAsync(ix->scan_perm.p, cand.p, ix->n_rows * 4, hipMemcpyDeviceToDevice, ix->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
        }
    }
    cand.release(); dsum.release();
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return ZH_OK; }
    if (e != hipSuccess) return fail(ZH_EHIP, "row order of the scan: %s", hipGetErrorString(e));
    ix->order_keys = best_keys;
    ix->order_worth = (double)best / ((double)(ix->n_rows - 1) * T);
    if (best_keys) ix->perm_rows = ix->n_rows;
    return ZH_OK;
}
// The scan's inputs (row_half / row_meta, row_leaf_p, row_half128) are rewritten IN PLACE when rows were appended, the forest changed or the row
// order is given up.  blk_mu orders the host threads that do it, not the GPU work of other contexts: a scan another pipelined context queued on the
// shared sweep stream (or its own) before this batch got here would read half-rewritten tiles.  Rewrites are rare (the first batch after an add /
// a rebuild): wait for everything the device has queued, then rewrite.  
static int quiesce_before_rewrite() {
    HIPCHK(hipDeviceSynchronize());
    return ZH_OK;
}
// the row -> leaf table in the scan's row order (under blk_mu; row_leaf must be valid)
static int ensure_row_leaf_p(zh_index *ix) {
    if (!ix->perm_rows) return ZH_OK;
    if (ix->row_leaf_p_gen == ix->row_leaf_gen && ix->row_leaf_p_perm == ix->perm_gen) return ZH_OK;
    const uint32_t T = ix->n_trees;
    // the same headroom rule as the copies themselves: a sixteenth of the device stays free for the batches' scratch
    const uint64_t want_p = std::max<uint64_t>(ix->n_rows * T, 1) * sizeof(uint2);
    size_t mem_free = 0, mem_total = 0;
    if (want_p > ix->row_leaf_p.cap && (zh_mem_info(&mem_free, &mem_total) != hipSuccess || mem_free < want_p + mem_total / 16)) { ix->perm_rows = 0; return ZH_OK; }
    int rc = quiesce_before_rewrite();
    if (rc) return rc;
    rc = ix->row_leaf_p.ensure(want_p);
    if (rc) { ix->perm_rows = 0; return ZH_OK; }  // (no room: the scan reads the table in id order -- but its tiles are in tree-0 order: the copy is remade)
    hipError_t e = zh_launch_permute_row_leaf(ix->row_leaf.as<uint2>(), ix->scan_perm.as<uint32_t>(), ix->perm_rows, ix->n_rows, T, ix->row_leaf_p.as<uint2>(), ix->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    if (e != hipSuccess) return fail(ZH_EHIP, "row -> leaf table in the scan's order: %s", hipGetErrorString(e));
    ix->row_leaf_p_gen = ix->row_leaf_gen;
    ix->row_leaf_p_perm = ix->perm_gen;
    return ZH_OK;
}

// The fp16 copy of the stored rows for the MFMA table scan (under blk_mu): rows appended since the last call only.  false: no room for it.
static int ensure_row_half(zh_index *ix, bool *ok) {
    *ok = false;
    if (ix->row_half_failed && ix->scale_gen == ix->rows_gen && ix->scale_rows == ix->n_rows) return ZH_OK;
    if (!ix->row_half_failed && ix->scale_gen == ix->rows_gen && ix->scale_rows == ix->n_rows) { *ok = true; return ZH_OK; }
    const uint64_t d = ix->opt.dim, tiles = (ix->n_rows + 15) / 16;
    ix->row_half_failed = false;
    // room for it?  (a copy that grows is allocated anew before the old one is freed; a sixteenth of the device stays free for the batches' scratch)
    size_t mem_free = 0, mem_total = 0;
    const uint64_t want = std::max<uint64_t>(tiles, 1) * 16 * (d * 2 + sizeof(float2));
    const bool room = want <= ix->row_half.cap + ix->row_meta.cap ||
                      (zh_mem_info(&mem_free, &mem_total) == hipSuccess && mem_free >= want + want / 2 + mem_total / 16);
    // No room for the copy (64M x 768 on one GPU: 196 GB of rows + 98 GB), or ZH_ROW_HALF_META_ONLY=1 (tests; read per call): the per-row scales and
    // norms alone (8 bytes per row) -- scan_mfma_kernel<D, true> then converts the f32 rows itself, to the bits the copy would hold (round 6: the
    // same matrix-core scan at every N of the series)
    const char *mo_e = getenv("ZH_ROW_HALF_META_ONLY");
    const bool meta_only = !room || (mo_e && mo_e[0] == '1');
    if (meta_only) {
        const uint64_t want_meta = std::max<uint64_t>(tiles, 1) * 16 * sizeof(float2);
        const bool room_meta = want_meta <= ix->row_meta.cap || (zh_mem_info(&mem_free, &mem_total) == hipSuccess && mem_free >= want_meta + mem_total / 16);
        ix->row_half.release();
        if (!room_meta || ix->row_meta.ensure(want_meta, false, ix->stream) != ZH_OK || ix->row_rho_dev.ensure(4) != ZH_OK) {
            ix->row_meta.release();
            ix->row_half_failed = true;
            ix->scale_rows = ix->n_rows; ix->scale_gen = ix->rows_gen;
            return ZH_OK;
        }
        int rcq = quiesce_before_rewrite();
        if (rcq) return rcq;
        ix->perm_rows = 0; ix->perm_gen++; ix->order_keys = 0;  // (the f32 rows are in id order)
        HIPCHK(hipMemsetAsync(ix->row_rho_dev.p, 0, 4, ix->stream));
        HIPCHK(zh_launch_row_half(ix->X.as<float>(), 0, ix->n_rows, (uint32_t)d, nullptr, ix->row_meta.as<float2>(), ix->row_rho_dev.as<uint32_t>(), nullptr, 0, ix->stream));
        float rho_m = 0.f;
        HIPCHK(hipMemcpyAsync(&rho_m, ix->row_rho_dev.p, 4, hipMemcpyDeviceToHost, ix->stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
        ix->row_rho = rho_m;
        ix->scale_rows = ix->n_rows; ix->scale_gen = ix->rows_gen;
        *ok = true;
        return ZH_OK;
    }
    if (ix->row_half.ensure(std::max<uint64_t>(tiles, 1) * 16 * d * 2, true, ix->stream) != ZH_OK ||
        ix->row_meta.ensure(std::max<uint64_t>(tiles, 1) * 16 * sizeof(float2), true, ix->stream) != ZH_OK || ix->row_rho_dev.ensure(4) != ZH_OK) {
        ix->row_half.release(); ix->row_meta.release();
        ix->row_half_failed = true;
        ix->scale_rows = ix->n_rows; ix->scale_gen = ix->rows_gen;
        return ZH_OK;
    }
    uint64_t from = ix->scale_rows;
    {
        const int rcq = quiesce_before_rewrite();
        if (rcq) return rcq;
    }
    // rows appended since the order was made keep position = id and share no tile with their leaf mates: once they are a quarter of the table
    // the copy is made again, in an order measured on all of it
    const bool stale_order = ix->perm_rows && ix->n_rows - ix->perm_rows > ix->perm_rows / 4;
    if (ix->scale_gen != ix->rows_gen || from > ix->n_rows || stale_order) {
        from = 0;
        HIPCHK(hipMemsetAsync(ix->row_rho_dev.p, 0, 4, ix->stream));
    }
    if (from == 0) {  // a copy made from scratch: in tree-0 leaf order where the forest allows (rows appended later keep position = id)
        int rcp = build_scan_perm(ix);
        if (rcp) return rcp;
    }
    HIPCHK(zh_launch_row_half(ix->X.as<float>(), from, ix->n_rows - from, (uint32_t)d, ix->row_half.p, ix->row_meta.as<float2>(),
                              ix->row_rho_dev.as<uint32_t>(), ix->perm_rows ? ix->scan_perm.as<uint32_t>() : nullptr, ix->perm_rows, ix->stream));
    float rho = 0.f;
    HIPCHK(hipMemcpyAsync(&rho, ix->row_rho_dev.p, 4, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    ix->row_rho = rho;
    ix->scale_rows = ix->n_rows; ix->scale_gen = ix->rows_gen;
    *ok = true;
    return ZH_OK;
}

// The row-major fp16 copy of a d = 128 table under one scale (under blk_mu).  Appended rows that fit the scale are added; a larger element
// than the scale allows re-makes the copy.  want_bytes: the caller's kernels can read the 128-byte copy of a table of integers 0 .. 255 -- made
// instead whenever every stored row qualifies (an appended row that does not re-makes the copy in halves); the two kinds share the buffer, and a
// caller that wants the other kind than the one present (tests switching kernels) gets the copy re-made.
static int ensure_row_half128(zh_index *ix, bool *ok, bool want_bytes) {
    *ok = false;
    const bool bytes = want_bytes && !ix->h128_not_bytes;
    const bool current = ix->h128_gen == ix->rows_gen && ix->h128_rows == ix->n_rows;
    if (current && (ix->h128_failed || ix->h128_bytes == bytes)) { *ok = !ix->h128_failed; return ZH_OK; }
    ix->h128_failed = false;
    size_t mem_free = 0, mem_total = 0;
    const uint64_t want = std::max<uint64_t>(ix->n_rows, 1) * (bytes ? 128 : 256);
    const bool room = want <= ix->row_half128.cap || (zh_mem_info(&mem_free, &mem_total) == hipSuccess && mem_free >= want + want / 2 + mem_total / 16);
    if (!room || ix->row_half128.ensure(want, true, ix->stream) != ZH_OK || ix->row_rho_dev.ensure(12) != ZH_OK) {
        ix->row_half128.release();
        ix->h128_failed = true; ix->h128_bytes = false;
        ix->h128_rows = ix->n_rows; ix->h128_gen = ix->rows_gen;
        return ZH_OK;
    }
    uint64_t from = (ix->h128_gen == ix->rows_gen && ix->h128_rows <= ix->n_rows && ix->h128_bytes == bytes) ? ix->h128_rows.load() : 0;
    {
        const int rcq = quiesce_before_rewrite();
        if (rcq) return rcq;
    }
    if (bytes) {
        uint32_t *dflag = ix->row_rho_dev.as<uint32_t>() + 2, flag = 0;
        HIPCHK(hipMemsetAsync(dflag, 0, 4, ix->stream));
        HIPCHK(zh_launch_row_byte128(ix->X.as<float>(), from, ix->n_rows - from, ix->row_half128.p, dflag, ix->stream));
        HIPCHK(hipMemcpyAsync(&flag, dflag, 4, hipMemcpyDeviceToHost, ix->stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
        if (!flag) {
            ix->h128_bytes = true; ix->h128_rho = 0.f; ix->h128_ex = 14;
            ix->h128_rows = ix->n_rows; ix->h128_gen = ix->rows_gen;
            *ok = true;
            return ZH_OK;
        }
        ix->h128_not_bytes = true;  // some row is not a row of bytes: the copy of halves, from the first row, now and from now on
        ix->h128_bytes = false; ix->h128_gen = 0;
        return ensure_row_half128(ix, ok, false);
    }
    uint32_t *dmax = ix->row_rho_dev.as<uint32_t>() + 1;
    for (int pass = 0; pass < 2; pass++) {
        uint32_t mbits = 0;
        HIPCHK(hipMemsetAsync(dmax, 0, 4, ix->stream));
        HIPCHK(zh_launch_absmax(ix->X.as<float>() + from * 128, (ix->n_rows - from) * 128, dmax, ix->stream));
        HIPCHK(hipMemcpyAsync(&mbits, dmax, 4, hipMemcpyDeviceToHost, ix->stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
        float m;
        memcpy(&m, &mbits, 4);
        int ex = 14;
        if (m > 0.f) (void)frexpf(m, &ex);
        ex = std::min(std::max(ex, -100), 100);
        if (from == 0) { ix->h128_ex = ex; break; }
        if (ex <= ix->h128_ex) break;
        from = 0;  // the appended rows do not fit the scale: the whole copy again, under theirs
    }
    if (from == 0) HIPCHK(hipMemsetAsync(ix->row_rho_dev.p, 0, 4, ix->stream));
    HIPCHK(zh_launch_row_half128(ix->X.as<float>(), from, ix->n_rows - from, ldexpf(1.f, 14 - ix->h128_ex), ix->row_half128.p,
                                 ix->row_rho_dev.as<uint32_t>(), ix->stream));
    float rho = 0.f;
    HIPCHK(hipMemcpyAsync(&rho, ix->row_rho_dev.p, 4, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    ix->h128_rho = from == 0 ? rho : std::max(rho, ix->h128_rho);
    ix->h128_bytes = false;
    ix->h128_rows = ix->n_rows; ix->h128_gen = ix->rows_gen;
    *ok = true;
    return ZH_OK;
}

// Leaf by leaf or the whole table once?  The leaf-major sweep gathers group_rows rows from HBM (5.9-6.1 TB/s of 3-KB rows,
// 5.5 TB/s of 512-byte ones); the table scan streams every stored row once (plus 8 bytes per tree of row -> leaf entries) and
// fetches one query from L2 per scored (row, query) pair -- measured 0.075 + 0.00022 d ns per pair chip-wide (4.1 G pairs/s
// at d = 768 = what the L2s deliver; 10 G pairs/s at d = 128 with the paired kernel, where the leaf-major sweep still wins at the BASELINE shapes), slower once the window's queries no
// longer fit beside the stream in the 8 x 4 MB of L2.  zh_set_sweep_mode / ZH_SWEEP_MODE=leaf|scan force one of them.
static bool use_approx(const zh_index *ix, const ZhTotals &tot, size_t B, size_t k, int metric);
// The half-width scan on the matrix cores (scan_mfma_kernel) where it has a kernel and the index can keep an fp16 copy of its rows
// (+50 % of the row table; ensure_row_half): mode 5 / ZH_NO_MFMA=1 keep the VALU kernel, which reads the f32 rows.
static bool mfma_wanted(const zh_index *ix) {
    static const bool off = getenv("ZH_NO_MFMA") != nullptr;
    return !off && ix->sweep_mode != 5 && zh_scan_mfma_supported(ix->opt.dim, ix->n_trees) && !ix->mfma_f32_off &&
           !(ix->row_half_failed && ix->scale_gen == ix->rows_gen && ix->scale_rows == ix->n_rows);
}
static bool choose_scan(const zh_index *ix, const ZhTotals &tot, int metric, size_t B, size_t k) {
    static const int forced = [] { const char *e = getenv("ZH_SWEEP_MODE"); return !e ? 0 : (e[0] == 's' ? 2 : (e[0] == 'a' ? 4 : 1)); }();
    const uint32_t d = ix->opt.dim, T = ix->n_trees;
    const int sm_ = ix->sweep_mode.load();
    int mode = sm_ == 3 ? forced : (sm_ ? sm_ : forced);
    if (mode == 5) mode = 4;
    if (mode == 6) mode = 1;
    if (mode == 1 || !zh_scan_sweep_supported(d, T, metric) || ix->row_leaf_failed || ix->scan_unsafe) return false;
    if (mode == 2 || mode == 4) return true;
    if (!ix->row_leaf_valid && ix->batches_since_change.load() < 3) return false;  // (the row -> leaf table is built for a forest that stays)
    const double row_b = 4.0 * d;
    const double t_leaf = (double)tot.group_rows * row_b / (d >= 256 ? 6.0e12 : 5.5e12);
    // with half-width queries (zh_approx.hip) a pair pulls 2 d bytes through the vector L1 instead of 4 d: measured 0.157 ns per pair at
    // d = 768 (cfg3, window 2, profiles/r04_*), and twice the queries fit beside the stream in L2
    const bool half = use_approx(ix, tot, B, k, metric), mfma = half && mfma_wanted(ix);
    const double q_bytes = (double)B * row_b * (half ? 0.5 : 1.0);
    // (on the matrix cores: 3.4-3.5 ms per 3.33M rows + 28.6M pairs at d = 768, window 2 -- the L2s' rate for 2 d bytes per pair; the rows are fp16 too)
    const double pair_s = mfma ? 0.02e-9 + 0.00011e-9 * d
                        : half ? 0.05e-9 + 0.00014e-9 * d
                               : (d == 128 ? 0.10e-9 : 0.075e-9 + 0.00022e-9 * d);  // (d = 128: the paired kernel, two pairs per step)
    const double t_pairs = (double)tot.rows * pair_s * (q_bytes > 8e6 ? 1.25 : 1.0);
    const double t_scan = std::max((double)ix->n_rows * (row_b * (mfma ? 0.5 : 1.0) + 8.0 * T) / 6.0e12, t_pairs) + 20e-6;
    return t_scan < 0.92 * t_leaf;
}

// The table scan with half-width queries (zh_approx.hip) instead of the f32 one?  Wherever it applies: the metrics whose key a dot
// product determines, the dimensions its groups divide, batches in the few-visits-per-pair regime (long leaves) whose candidate
// lists fit the per-query sort.  ZH_SWEEP_MODE=scan / zh_set_sweep_mode(2) keep the f32 scan; =approx / mode 4 say so explicitly.
static bool use_approx(const zh_index *ix, const ZhTotals &tot, size_t B, size_t k, int metric) {
    static const int forced = [] { const char *e = getenv("ZH_SWEEP_MODE"); return !e ? 0 : (e[0] == 's' ? 2 : (e[0] == 'a' ? 4 : 1)); }();
    static const bool off = getenv("ZH_NO_APPROX") != nullptr;
    const int sm_ = ix->sweep_mode.load();
    int mode = sm_ == 3 ? forced : (sm_ ? sm_ : forced);
    if (mode == 5) mode = 4;
    if (off || mode == 2 || mode == 1 || mode == 6) return false;
    if (!zh_scan_approx_supported(ix->opt.dim, ix->n_trees, metric) || k > 256 || B == 0 || B >= (1u << 24)) return false;  // (24 bits of a packed pair record)
    if (tot.takes > 2048ull * B || tot.visits > 8ull * B * ix->n_trees) return false;
    // by itself only where it pays: per pair it moves half the bytes, but a window of ONE cfg3 batch (4.3 pairs per stored row) gains 4 % on
    // the scan (3.28 against 3.42 ms per launch) and pays more than that for the interval stages: from ~5 pairs per stored row on
    // (the matrix-core kernel reads half the row bytes as well and multiplies for nothing: measured from d = 256 and ~2 pairs per stored row on)
    const bool mfma = mfma_wanted(ix);
    if (mode != 4 && (ix->approx_strikes.load() >= 2 || !(mfma ? ix->opt.dim >= 256 : zh_approx_pays(ix->opt.dim)) ||
                      tot.rows < (mfma ? 2 : 5) * ix->n_rows))
        return false;
    return true;
}

// The leaf-major sweep of a d = 128 table at half width (sweep128h_kernel)?  Where the f32 sweep is what HBM gives random 512-byte rows: long
// leaves, many scored rows; the same list limits as the half-width scan.  zh_set_sweep_mode(6) asks for it wherever it is implemented,
// (1) keeps the f32 sweep; ZH_NO_LEAF_HALF=1 switches it off.
static bool use_approx_leaf(const zh_index *ix, const ZhTotals &tot, size_t B, size_t k, int metric) {
    static const bool off = getenv("ZH_NO_LEAF_HALF") != nullptr || getenv("ZH_NO_APPROX") != nullptr;
    static const bool env_sweep = getenv("ZH_SWEEP_MODE") != nullptr;
    if (off || ix->opt.dim != 128 || !zh_sweep_has_predicate(128, metric)) return false;
    if (ix->sweep_mode != 6 && (env_sweep || (ix->sweep_mode != 0 && ix->sweep_mode != 3))) return false;
    if (k > 256 || B == 0 || B >= (1u << 24) || tot.takes > 2048ull * B || tot.visits > 8ull * B * ix->n_trees) return false;
    if (ix->h128_failed && ix->h128_gen == ix->rows_gen && ix->h128_rows == ix->n_rows) return false;
    if (ix->sweep_mode != 6 && (ix->approx_strikes.load() >= 2 || tot.group_rows < (4u << 20))) return false;
    return true;
}

static bool use_score_hash(const zh_index *ix, size_t B, uint32_t P_dense);
// estimated time of the row-score hash of a batch: the score GEMM (or its row-streaming bound), the gather of two score rows per
// plane (>= one 64-byte sector each), the exact fix-ups
static double score_hash_seconds(const zh_index *ix, size_t B) {
    const size_t Bp = (B + 3) & ~(size_t)3;
    const double d = ix->opt.dim, P = ix->n_planes, N = ix->n_rows;
    return std::max(2.0 * Bp * N * d / 9e13, N * d * 4.0 / 5e12) + P * 2.0 * std::max(64.0, Bp * 4.0) / 4e12 + 0.003 * B * P * 0.55e-9 + 30e-6;
}
// number of leading planes hashed densely (MFMA kernel, before the walk) for a batch of B queries asking for k neighbours
static uint32_t choose_dense_planes(zh_index *ix, size_t B, size_t k) {
    const auto &below = ix->planes_below_level;
    if (below.size() <= 1 || ix->n_planes == 0) return 0;
    if (ix->dense_levels >= 0) return below[std::min<size_t>((size_t)ix->dense_levels, below.size() - 1)];
    // The top levels, shared by every query, always go to the MFMA kernel (a 2 GFLOP budget: tens of microseconds).
    const double d = ix->opt.dim, per_plane = 2.0 * (double)B * d;
    uint32_t top = 0;
    for (size_t L = 1; L < below.size(); L++)
        if ((double)below[L] * per_plane <= 2e9) top = below[L];
    // Below them a pair either hashes the planes it meets on demand (a 16-lane ordered fma chain per plane) or finds every
    // sign precomputed.  With leaves comfortably >= k a pair meets ~depth planes and on demand is all but free; with
    // small leaves (the reference's defaults, max_node_size 5 < top_k) the walk wanders over a good part of every tree
    //  and the dense kernel, ~20x cheaper per sign, wins although it hashes planes nobody asks for.  The
    // visits per pair of the previous batches decide; measured constants (MI355X): 90 TF for the dense kernel, 1.4 us per
    // step of the slowest pair (~5x the mean visit count) or 0.7 us of a SIMD per four steps when waves queue.
    double vpp;
    {
        std::lock_guard<std::mutex> lk(ix->stats_mu);
        vpp = ix->visits_per_pair;
    }
    if (vpp <= 0) vpp = (size_t)ix->opt.max_node_size < 2 * k + 2 ? 1000.0 : 2.0;
    const double pairs = (double)B * ix->n_trees;
    const double t_chain = std::max(1.4e-6 * 5.0 * vpp, pairs / 4.0 * 2.3 * vpp * 0.7e-6 * (d / 384.0) / 1024.0);
    // the dense kernel: MFMA-bound for real batches, a plane-streaming GEMV (HBM-bound) for a handful of queries
    const double t_dense = std::max((double)ix->n_planes * per_plane / 9e13, (double)ix->n_planes * d * 4.0 / 5e12);
    const double bits_bytes = (double)ix->n_planes * (double)B / 8.0;
    // ... or every sign from the row scores (zh_score.hip) -- counted only where use_score_hash would in fact take that path for an
    // all-dense batch (same mode switch, same ZH_HASH_MODE, same limits, same threshold: ONE decision), so that "all-dense because the
    // scores are cheap" is never followed by the per-plane dense hash
    double t_all = t_dense;
    if (use_score_hash(ix, B, ix->n_planes)) t_all = std::min(t_all, score_hash_seconds(ix, B));
    if (t_all < t_chain && bits_bytes < 2e9) return ix->n_planes;  // (every context in flight holds its own sign bits)
    return top;
}

// All signs of the batch from N row scores per query instead of P dot products (zh_score.hip)?  Only for forests whose planes
// are known differences of stored rows, when every plane is hashed anyway, and when the score table fits.
static bool use_score_hash(const zh_index *ix, size_t B, uint32_t P_dense) {
    static const int forced = [] { const char *e = getenv("ZH_HASH_MODE"); return !e ? 0 : (e[0] == 's' ? 2 : 1); }();
    const int mode = ix->hash_mode ? ix->hash_mode : forced;
    if (mode == 1 || !ix->samples_valid || ix->n_planes == 0 || P_dense < ix->n_planes) return false;
    const size_t Bp = (B + 3) & ~(size_t)3;  // (the kernels take queries four at a time: the batch is padded with zero queries)
    if ((uint64_t)ix->n_rows * Bp * 4 > (12ull << 30) || ix->n_rows > 0xFFFFFFF0ull) return false;  // the score table of ONE context (288 GB of HBM)
    if (mode == 2) return true;
    // per-plane hash: MFMA-bound for real batches, a plane-streaming GEMV for a handful of queries; row scores: the same two
    // bounds over the ROWS, + the gather of two score rows per plane (>= one 64-byte sector each) + the exact fix-ups
    const double d = ix->opt.dim, P = ix->n_planes;
    const double t_dense = std::max(2.0 * B * P * d / 9e13, P * d * 4.0 / 5e12);
    return score_hash_seconds(ix, B) < 0.7 * t_dense;
}

static int launch_score_hash(zh_search_ctx *c, const float *dQ, size_t B, bool lazy, hipStream_t s) {
    zh_index *ix = c->ix;
    const uint32_t d = ix->opt.dim;
    int rc;
    if (ix->norm_rows != ix->n_rows || ix->norm_gen != ix->rows_gen) {  // rows were added (or replaced: clear) since the norms were taken
        std::lock_guard<std::mutex> lk(ix->blk_mu);
        if (ix->norm_rows != ix->n_rows || ix->norm_gen != ix->rows_gen) {
            if ((rc = ix->row_hn2.ensure(ix->n_rows * 4)) || (rc = ix->row_norm.ensure(ix->n_rows * 4))) return rc;
            HIPCHK(zh_launch_row_norms(ix->X.as<float>(), ix->n_rows, d, ix->row_hn2.as<float>(), ix->row_norm.as<float>(), ix->stream));
            HIPCHK(hipStreamSynchronize(ix->stream));
            ix->norm_rows = ix->n_rows; ix->norm_gen = ix->rows_gen;
        }
    }
    if (ix->hab_planes != ix->n_planes || ix->hab_rows != ix->norm_rows || ix->hab_gen != ix->norm_gen) {  // planes were added, or the norms retaken
        std::lock_guard<std::mutex> lk(ix->blk_mu);
        if (ix->hab_planes != ix->n_planes || ix->hab_rows != ix->norm_rows || ix->hab_gen != ix->norm_gen) {
            if ((rc = ix->plane_hab.ensure(std::max<size_t>(ix->n_planes, 1) * sizeof(float4)))) return rc;
            HIPCHK(zh_launch_plane_hab(ix->plane_samples.as<uint2>(), ix->n_planes, ix->row_hn2.as<float>(), ix->row_norm.as<float>(),
                                       ix->plane_hab.as<float4>(), ix->stream));
            HIPCHK(hipStreamSynchronize(ix->stream));
            ix->hab_planes = ix->n_planes; ix->hab_rows = ix->norm_rows; ix->hab_gen = ix->norm_gen;
        }
    }
    if (B % 4) {  // pad the batch with zero queries to a multiple of four (their signs are computed and never read)
        const size_t Bp = (B + 3) & ~(size_t)3;
        if ((rc = c->wQpad.ensure(Bp * d * 4)) || (rc = c->wBits.ensure(Bp * c->wpq * 4))) return rc;
        if (lazy && (rc = c->wUnc.ensure(Bp * c->wpq * 4))) return rc;
        HIPCHK(hipMemcpyAsync(c->wQpad.p, dQ, B * d * 4, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemsetAsync(c->wQpad.as<float>() + B * d, 0, (Bp - B) * d * 4, s));
        dQ = c->wQpad.as<float>();
        B = Bp;
    }
    c->score_Bp = (uint32_t)B;
    const uint32_t wq = (uint32_t)((B + 63) / 64 * 2);  // sign words per ROW of the score GEMM's (unused) bit output
    const uint64_t cap64 = (uint64_t)B * ix->n_planes / 64 + (1u << 16);  // ~1.6 % of the signs: 4x the share seen on ~N(0,1) rows
    const uint32_t fix_cap = (uint32_t)std::min<uint64_t>(cap64, 0x7FFFFFFFull);
    if ((rc = c->wScore.ensure((size_t)ix->n_rows * B * 4))) return rc;
    if ((rc = c->wJunkBits.ensure((size_t)ix->n_rows * wq * 4))) return rc;
    if ((rc = c->wQnorm.ensure(B * 4))) return rc;
    if ((rc = c->wFixList.ensure((size_t)fix_cap * sizeof(uint2)))) return rc;
    if (c->wZeros.cap < B * 4) {
        if ((rc = c->wZeros.ensure(B * 4))) return rc;
        HIPCHK(hipMemsetAsync(c->wZeros.p, 0, c->wZeros.cap, s));
    }
    // S[row][q] = row . query: the MFMA hash kernel with the roles swapped (stored rows as "queries", the batch as B "planes");
    // a batch of <= 4 queries: one stream over the stored rows
    if (B == 4)
        HIPCHK(zh_launch_row_scores4(ix->X.as<float>(), ix->n_rows, d, dQ, c->wScore.as<float>(), s));
    else
        HIPCHK(zh_launch_hash_dense(ix->X.as<float>(), (uint32_t)ix->n_rows, dQ, c->wZeros.as<float>(), (uint32_t)B, d, c->wJunkBits.as<uint32_t>(), wq,
                                    c->wScore.as<float>(), s));
    HIPCHK(zh_launch_row_norms(dQ, B, d, nullptr, c->wQnorm.as<float>(), s));
    ZhTotals *tot = c->wTotals.as<ZhTotals>();
    HIPCHK(zh_launch_score_signs(c->wScore.as<float>(), (uint32_t)B, ix->plane_samples.as<uint2>(), ix->n_planes, ix->row_hn2.as<float>(),
                                 ix->row_norm.as<float>(), ix->plane_hab.as<float4>(), c->wQnorm.as<float>(), dQ, d, ix->planes.as<float>(), ix->consts.as<float>(),
                                 c->wBits.as<uint32_t>(), c->wpq, c->wFixList.as<uint2>(), fix_cap, &tot->hash_fixups,
                                 lazy ? c->wUnc.as<uint32_t>() : nullptr, s));
    return ZH_OK;
}

int ctx_wait(zh_search_ctx *c);

// A batch hashed from row scores can pick its candidates from them (zh_search.hip, "Prefilter"): for the metrics the scores
// determine, leaves short enough for the per-lane selection, and forests this library built (a row is in one leaf per tree).
static bool use_prefilter(const zh_index *ix, const zh_search_ctx *c, size_t k, int metric) {
    static const bool off = getenv("ZH_NO_PREFILTER") != nullptr;  // A/B: sweep + select for every batch
    // (mode 3 keeps trying past two strikes -- a caller who knows the overflows are rare -- but not past eight in a row: every one is a batch computed twice)
    const uint32_t strikes = ix->prefilter_strikes.load();
    if (off || !c->score_hash || c->prefilter_off_once || c->stream_ordered || (strikes >= 2 && ix->sweep_mode != 3) || strikes >= 8) return false;
    static const bool env_sweep = getenv("ZH_SWEEP_MODE") != nullptr;
    if ((ix->sweep_mode != 0 && ix->sweep_mode != 3) || (env_sweep && ix->sweep_mode != 3)) return false;  // a sweep was asked for
    if (metric != ZH_L2SQ && metric != ZH_L2 && metric != ZH_COSINE) return false;
    // leaves of <= 8 rows are judged per lane; the odd longer one (rows a split could not separate) is a visit for the exact path
    if (ix->opt.max_node_size > 8 || ix->max_leaf_len == 0 || ix->max_leaf_len > 64 || k > 64 || ix->scan_unsafe || !ix->n_leaf_ids) return false;
    return true;
}

static int build_leaf_meta(zh_index *ix) {  // under blk_mu; the norms are those launch_score_hash just validated
    int rc;
    if ((rc = ix->leaf_meta.ensure(ix->n_leaf_ids * sizeof(float4)))) return rc;
    HIPCHK(zh_launch_leaf_meta(ix->leaf_ids.as<uint32_t>(), ix->n_leaf_ids, ix->row_hn2.as<float>(), ix->row_norm.as<float>(),
                               ix->leaf_meta.as<float4>(), ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    ix->leaf_meta_valid = true; ix->meta_rows = ix->norm_rows; ix->meta_gen = ix->norm_gen;
    return ZH_OK;
}

static ZhWalkLog walk_log(const zh_search_ctx *c) {
    ZhWalkLog l;
    l.pool = c->wLogPool.as<uint2>();
    l.capacity = (uint32_t)std::min<size_t>(c->log_chunks, 0xFFFFFFFEu);
    l.head = c->wLogHead.as<uint32_t>();
    l.ctl = c->wLogCtl.as<ZhLogCtl>();
    // a prefiltered batch with leaves the entry's 16 bits can hold logs {leaf offset, take | length << 16}: prefilter_kernel then
    // reads no node record per visit (1.1 GB of 64-byte sectors and a dependent round trip per 63 visits at the reference's defaults)
    l.leaf_entries = c->prefilter && c->ix->max_leaf_len <= 64 && c->k <= 0xFFFF ? 1u : 0u;
    return l;
}

// first half of a batch: hash, the walk's counting pass, scans; the three totals travel to the host
static int ctx_begin(zh_search_ctx *c, const float *const *dQs, size_t nwin, size_t bwin, size_t k, int metric, int mode,
                     hipStream_t s) {
    zh_index *ix = c->ix;
    const uint32_t d = ix->opt.dim, T = ix->n_trees;
    const size_t B = nwin * bwin;
    const float *dQ = dQs[0];
    int rc;
    if (c->state == 1) return fail(ZH_ESTATE, "zh_search_begin: the context already has a batch begun; finish it first");
    if (ix->broken) return fail(ZH_ESTATE, "an earlier zh_index_add failed half way: call zh_index_build before searching");
    if (c->state == 2 && (rc = ctx_wait(c))) return rc;  // the previous batch was never waited for: retire it
    c->dQ = dQ; c->B = B; c->k = k; c->metric = metric; c->mode = mode; c->s = s;
    c->nwin = nwin; c->bwin = bwin;
    c->trivial = (B == 0 || k == 0 || ix->n_rows == 0 || T == 0);  // core.rs:295-297: empty index -> no neighbours
    c->approx = false; c->approx_leaf = false; c->approx_mfma = false; c->approx_fused = false; c->approx_bytes = false;
    c->state = 1;
    if (c->trivial) return ZH_OK;
    const uint64_t pairs = (uint64_t)B * T;
    if (pairs >= (1ull << 26)) { c->state = 0; return fail(ZH_ELIMIT, "batch * num_trees >= 2^26"); }
    c->P_dense = choose_dense_planes(ix, B, k);
    const uint32_t seen = ix->batches_since_change.fetch_add(1);
    if (c->P_dense >= ix->n_planes && ix->n_planes && !ix->blocks_valid && seen >= 2) {  // the forest has been stable for a few batches
        std::lock_guard<std::mutex> lk(ix->blk_mu);
        if (!ix->blocks_valid && (rc = build_blocks(ix))) { c->state = 0; return rc; }
    }
    c->wpq = (c->P_dense + 255) / 256 * 8;  // sign words per query, a multiple of eight: score_signs_kernel stores whole 32-byte sectors
    const size_t nn = std::max<uint32_t>(ix->n_nodes, 1);
    c->state = 0;  // a failure below leaves the context idle
    if (nwin > 1) {  // the window's batches side by side: the kernels address query b of the window as Q + b * d
        if ((rc = c->wQwin.ensure(B * d * 4))) return rc;
        for (size_t j = 0; j < nwin; j++)
            HIPCHK(hipMemcpyAsync(c->wQwin.as<float>() + j * bwin * d, dQs[j], bwin * d * 4, hipMemcpyDeviceToDevice, s));
        c->dQ = dQ = c->wQwin.as<float>();
    }
    if ((rc = c->wQQ.ensure(B * 4))) return rc;
    if ((rc = c->wBits.ensure(std::max<size_t>((size_t)B * c->wpq * 4, 4)))) return rc;
    if ((rc = c->wCounts.ensure(pairs * sizeof(ZhPairCounts)))) return rc;
    if ((rc = c->wInline.ensure(pairs * ZH_INLINE_VISITS * sizeof(ZhVisit)))) return rc;
    if ((rc = c->wRowBase.ensure((pairs + 1) * 8))) return rc;
    if ((rc = c->wCandBase.ensure((pairs + 1) * 8))) return rc;
    if ((rc = c->wVisitBase.ensure((pairs + 1) * 8))) return rc;
    if ((rc = c->wTotals.ensure(sizeof(ZhTotals)))) return rc;
    if ((rc = c->wLeafCount.ensure(nn * 4))) return rc;
    if ((rc = c->wLeafFill.ensure(nn * 4))) return rc;
    if ((rc = c->wGroupBase.ensure(nn * 4))) return rc;
    if ((rc = c->wGroupRowBase.ensure(nn * 8))) return rc;
    if ((rc = c->wLogHead.ensure(pairs * 4))) return rc;
    if ((rc = c->wLogCtl.ensure(sizeof(ZhLogCtl)))) return rc;
    if (!c->wLogPool.p) {
        static const size_t first_env = [] { const char *e = getenv("ZH_WALK_LOG_CHUNKS"); return e ? (size_t)atoll(e) : (size_t)0; }();
        size_t first = first_env ? first_env : 8192;
        if (!first_env) {  // a context created after the index has served batches starts with a log that fits what those batches logged
            std::lock_guard<std::mutex> lk(ix->stats_mu);
            const double v = ix->visits_per_pair * (double)pairs;
            if (v > 0) first = std::max<size_t>(first, (size_t)(1.5 * (v / (ZH_LOG_CHUNK - 1) + (double)pairs)) + 16);
        }
        if ((rc = c->wLogPool.ensure(std::max<size_t>(first, 1) * ZH_LOG_CHUNK * sizeof(uint2)))) return rc;
        c->log_chunks = std::max<size_t>(first, 1);
    }
    c->score_hash = use_score_hash(ix, B, c->P_dense);
    c->prefilter = use_prefilter(ix, c, k, metric);
    c->sv_q.assign(dQs, dQs + nwin);
    // a prefiltered batch forms no leaf groups: no per-leaf visit counts to zero, to bump from the walk, or to scan afterwards
    uint32_t *leaf_count = c->prefilter ? nullptr : c->wLeafCount.as<uint32_t>();
    HIPCHK(zh_launch_batch_init(c->wLeafCount.as<uint32_t>(), c->wLeafFill.as<uint32_t>(), c->prefilter ? 0u : (uint32_t)nn, c->wLogCtl.as<ZhLogCtl>(),
                                c->wTotals.as<ZhTotals>(), s));
    ZhForestDev f = forest_dev(ix);
    HIPCHK(hipEventRecord(c->ev[0], s));
    if (metric == ZH_COSINE) HIPCHK(zh_launch_qnorm(dQ, (uint32_t)B, d, c->wQQ.as<float>(), s));
    static const bool no_blocks = getenv("ZH_WALK_NO_BLOCKS") != nullptr;  // A/B: the pointer walk for all-dense signs too
    const bool blocked = c->P_dense >= ix->n_planes && ix->n_planes && ix->blocks_valid && !no_blocks;
    static const bool eager_fix = getenv("ZH_EAGER_FIXUPS") != nullptr;  // A/B: every uncertain sign recomputed before the walk
    c->lazy_fix = c->score_hash && blocked && !eager_fix && ((B + 3) & ~(size_t)3) != 4 && d % 4 == 0;  // (plane_above_wave's shape)
    if (c->lazy_fix && (rc = c->wUnc.ensure(std::max<size_t>((size_t)B * c->wpq * 4, 4)))) return rc;
    if (c->score_hash) {
        if ((rc = launch_score_hash(c, dQ, B, c->lazy_fix, s))) return rc;
    } else if (c->P_dense)
        HIPCHK(zh_launch_hash_dense(dQ, (uint32_t)B, f.planes, f.consts, c->P_dense, d, c->wBits.as<uint32_t>(), c->wpq, nullptr, s));
    HIPCHK(hipEventRecord(c->ev[1], s));
    if (blocked) {
        ZhBlocksDev bd;
        bd.recs = ix->blk_recs.as<int4>(); bd.upper = ix->blk_upper.as<int4>(); bd.root = ix->blk_roots.as<int2>();
        bd.n_blocks = ix->n_blocks; bd.n_upper = ix->n_upper; bd.inner_only = ix->blocks_inner ? 1u : 0u;
        bd.recs_b = ix->blocks_inner ? bd.recs + ix->blk_recs_b : nullptr;
        HIPCHK(zh_launch_walk_blocked(f, bd, (uint32_t)B, (int32_t)k, c->wBits.as<uint32_t>(), c->wpq, c->wCounts.as<ZhPairCounts>(),
                                      c->wInline.as<ZhVisit>(), leaf_count, walk_log(c), c->lazy_fix ? c->wUnc.as<uint32_t>() : nullptr, dQ, d, s));
    } else
        HIPCHK(zh_launch_walk_count(f, dQ, (uint32_t)B, d, (int32_t)k, c->wBits.as<uint32_t>(), c->wpq, c->P_dense,
                                    c->wCounts.as<ZhPairCounts>(), c->wInline.as<ZhVisit>(), leaf_count, walk_log(c), s));
    if (!c->prefilter)
        HIPCHK(zh_launch_leaf_scan(f, c->wLeafCount.as<uint32_t>(), c->wGroupBase.as<uint32_t>(),
                                   c->wGroupRowBase.as<uint64_t>(), c->wTotals.as<ZhTotals>(), s));
    HIPCHK(zh_launch_pair_scan(c->wCounts.as<ZhPairCounts>(), (uint32_t)pairs, c->wRowBase.as<uint64_t>(),
                               c->wCandBase.as<uint64_t>(), c->wVisitBase.as<uint64_t>(), c->wTotals.as<ZhTotals>(),
                               c->wLogCtl.as<ZhLogCtl>(), s));
    HIPCHK(hipMemcpyAsync(c->h_totals, c->wTotals.p, sizeof(ZhTotals), hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(c->ev_totals, s));
    c->state = 1;
    return ZH_OK;
}

// second half of a prefiltered batch: candidates from the row scores -> the exact path for what they cannot decide -> exact keys
// of the few survivors -> final top-k over num_trees short lists per query
static int finish_prefilter(zh_search_ctx *c, uint64_t *dOutIds, uint64_t *dOutKeys, uint32_t *dOutCounts, uint64_t *const *outIds,
                            uint64_t *const *outKeys, uint32_t *const *outCounts) {
    zh_index *ix = c->ix;
    const uint32_t d = ix->opt.dim, T = ix->n_trees;
    const size_t B = c->B, k = c->k, nwin = c->nwin, bwin = c->bwin;
    hipStream_t s = c->s;
    int rc;
    if (!ix->leaf_meta_valid || ix->meta_rows != ix->norm_rows || ix->meta_gen != ix->norm_gen) {
        std::lock_guard<std::mutex> lk(ix->blk_mu);
        if ((!ix->leaf_meta_valid || ix->meta_rows != ix->norm_rows || ix->meta_gen != ix->norm_gen) && (rc = build_leaf_meta(ix))) return rc;
    }
    const char *cap_e = getenv("ZH_PREFILTER_CAP");  // tests: lists that run over (read per batch)
    const uint32_t cap_env = cap_e ? (uint32_t)atoi(cap_e) : 0u;
    const uint32_t cap = cap_env ? std::max<uint32_t>(cap_env, 1) : (k <= 16 && c->metric != ZH_COSINE ? 64u : (k <= 32 ? 128u : 256u));
    // (cosine with parity keys: "nearest" is the smallest non-negative cosine, where candidates are densest, and every row within the
    // bound of ZERO is a legitimate top candidate from the exact pass: ~49 rows per list at d = 768, batch 196 -- 64 slots ran over)
    const uint64_t lists = (uint64_t)B * T;
    const uint32_t amb_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(4096, c->tot.visits / 32 + 1024), 1u << 26);
    c->pf_cap = cap;
    if ((rc = c->wPfRows.ensure(lists * cap * 4)) || (rc = c->wPfCounts.ensure(lists * 8)) || (rc = c->wPfKeys.ensure(lists * cap * 8)) ||
        (rc = c->wPfIds.ensure(lists * cap * 8)) || (rc = c->wPfAmb.ensure((size_t)amb_cap * sizeof(uint4))) || (rc = c->wPfCtl.ensure(16)))
        return rc;
    HIPCHK(hipMemsetAsync(c->wPfCtl.p, 0, 16, s));
    ZhPrefilter pf;
    pf.S = c->wScore.as<float>(); pf.Bp = c->score_Bp; pf.leaf_meta = ix->leaf_meta.as<float4>(); pf.qnorm = c->wQnorm.as<float>();
    pf.rows = c->wPfRows.as<uint32_t>(); pf.counts = c->wPfCounts.as<uint32_t>(); pf.tau = reinterpret_cast<float *>(pf.counts + lists); pf.cap = cap;
    pf.amb = c->wPfAmb.as<uint4>(); pf.amb_cap = amb_cap; pf.ctl = c->wPfCtl.as<uint32_t>();
    ZhForestDev f = forest_dev(ix);
    HIPCHK(hipEventRecord(c->ev[2], s));
    HIPCHK(hipEventRecord(c->ev_sw0, s));
    HIPCHK(zh_launch_prefilter(f, d, (uint32_t)B, (uint32_t)k, c->metric, c->mode, c->wCounts.as<ZhPairCounts>(), c->wInline.as<ZhVisit>(),
                               walk_log(c), pf, s));
    HIPCHK(hipEventRecord(c->ev_sw1, s));
    HIPCHK(hipEventRecord(c->ev[3], s));
    HIPCHK(zh_launch_prefilter_exact(f, d, ix->X.as<float>(), c->dQ, c->wQQ.as<float>(), (uint32_t)B, c->metric, c->mode, ix->opt.id_base, pf,
                                     c->wPfKeys.as<uint64_t>(), c->wPfIds.as<uint64_t>(), s));
    HIPCHK(hipEventRecord(c->ev[4], s));
    HIPCHK(zh_launch_final_lists(T, (uint32_t)B, (uint32_t)k, cap, c->wPfKeys.as<uint64_t>(), c->wPfIds.as<uint64_t>(), c->wPfCounts.as<uint32_t>(),
                                 dOutIds, dOutKeys, dOutCounts, pf.ctl + 1, s));
    for (size_t j = 0; j < nwin && nwin > 1; j++) {
        HIPCHK(hipMemcpyAsync(outIds[j], dOutIds + j * bwin * k, bwin * k * 8, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(outKeys[j], dOutKeys + j * bwin * k, bwin * k * 8, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(outCounts[j], dOutCounts + j * bwin, bwin * 4, hipMemcpyDeviceToDevice, s));
    }
    HIPCHK(hipMemcpyAsync(c->h_pf, c->wPfCtl.p, 16, hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(c->ev[5], s));
    c->state = 2;
    return ZH_OK;
}

// second half: waits (host side) for the totals only, sizes the scratch, enqueues emit -> sweep -> select -> final
static int ctx_finish(zh_search_ctx *c, uint64_t *const *outIds, uint64_t *const *outKeys, uint32_t *const *outCounts,
                      hipStream_t heavy) {
    zh_index *ix = c->ix;
    if (c->state != 1) return fail(ZH_ESTATE, "zh_search_finish without zh_search_begin");
    const uint32_t d = ix->opt.dim, T = ix->n_trees;
    const size_t B = c->B, k = c->k, nwin = c->nwin, bwin = c->bwin;
    hipStream_t s = c->s;
    int rc;
    c->state = 0;
    if (c->trivial) {
        for (size_t j = 0; j < nwin && bwin; j++) {
            HIPCHK(hipMemsetAsync(outCounts[j], 0, bwin * 4, s));
            if (k) {
                HIPCHK(hipMemsetAsync(outIds[j], 0xFF, bwin * k * 8, s));
                HIPCHK(hipMemsetAsync(outKeys[j], 0xFF, bwin * k * 8, s));
            }
        }
        HIPCHK(hipEventRecord(c->ev[5], s));
        c->state = 2;
        return ZH_OK;
    }
    // one batch: the final kernel writes straight into the caller's buffers; a window: into the context's own, handed out
    // per batch by small device copies afterwards
    uint64_t *dOutIds = outIds[0], *dOutKeys = outKeys[0];
    uint32_t *dOutCounts = outCounts[0];
    if (nwin > 1) {
        if ((rc = c->wOutWin.ensure(B * k * 16 + B * 4))) { return rc; }
        dOutIds = c->wOutWin.as<uint64_t>();
        dOutKeys = dOutIds + B * k;
        dOutCounts = reinterpret_cast<uint32_t *>(dOutKeys + B * k);
    }
    HIPCHK(hipEventSynchronize(c->ev_totals));
    const ZhTotals tot = c->tot = *c->h_totals;
    {
        std::lock_guard<std::mutex> lk(ix->stats_mu);
        const double v = (double)tot.visits / (double)((uint64_t)B * T);
        ix->visits_per_pair = ix->visits_per_pair > 0 ? 0.5 * (ix->visits_per_pair + v) : v;
    }
    if (tot.visits > 0xFFFFFFFull || tot.rows >= (1ull << 36))
        return fail(ZH_ELIMIT, "more than 2^28 leaf visits or 2^36 scored rows in one batch; use a smaller batch");
    c->sv_ids.assign(outIds, outIds + nwin); c->sv_keys.assign(outKeys, outKeys + nwin); c->sv_counts.assign(outCounts, outCounts + nwin);
    c->sv_heavy = heavy;
    if (c->prefilter && (tot.flags & 1u)) {
        // the visit log ran out (an index's first wandering batches): this batch again the classic way -- its emit walk needs the
        // per-leaf counts a prefiltered first half does not take -- and with a log that fits the next one
        const std::vector<const float *> q = c->sv_q;
        c->prefilter_off_once = true;
        rc = ctx_begin(c, q.data(), nwin, bwin, k, c->metric, c->mode, s);
        c->prefilter_off_once = false;
        if (rc) return rc;
        return ctx_finish(c, outIds, outKeys, outCounts, heavy);
    }
    if (c->prefilter) return finish_prefilter(c, dOutIds, dOutKeys, dOutCounts, outIds, outKeys, outCounts);
    if ((rc = c->wVisits.ensure(std::max<uint64_t>(tot.visits, 1) * sizeof(ZhVisit)))) return rc;
    if ((rc = c->wGroups.ensure(std::max<uint64_t>(tot.groups, 1) * sizeof(ZhGroup)))) return rc;
    if ((rc = c->wGroupRowOff.ensure(std::max<uint64_t>(tot.groups, 1) * 8))) return rc;
    static const bool no_wave_table = getenv("ZH_NO_WAVE_TABLE") != nullptr;  // A/B: per-lane binary search over all groups
    if (!no_wave_table && (rc = c->wWaveGroup.ensure((tot.group_rows / 64 + 2) * 4))) return rc;
    if ((rc = c->wKeys.ensure(std::max<uint64_t>(tot.rows, 1) * 8))) return rc;
    if ((rc = c->wCandKeys.ensure(std::max<uint64_t>(tot.takes, 1) * 8))) return rc;
    if ((rc = c->wCandIds.ensure(std::max<uint64_t>(tot.takes, 1) * 4))) return rc;
    ZhForestDev f = forest_dev(ix);
    if (!(tot.flags & 1u)) {
        // the counting pass logged every visit: place them (no second walk)
        HIPCHK(zh_launch_expand(f, (uint32_t)B, c->wCounts.as<ZhPairCounts>(), c->wInline.as<ZhVisit>(),
                                c->wRowBase.as<uint64_t>(), c->wCandBase.as<uint64_t>(), c->wVisitBase.as<uint64_t>(),
                                c->wVisits.as<ZhVisit>(), c->wLeafCount.as<uint32_t>(), c->wLeafFill.as<uint32_t>(),
                                c->wGroupBase.as<uint32_t>(), c->wGroupRowBase.as<uint64_t>(), c->wGroups.as<ZhGroup>(),
                                c->wGroupRowOff.as<uint64_t>(), walk_log(c), s));
    } else {
        // the visit log ran out of chunks: walk again, emitting this time, and give the next batch a log that fits
        if (c->lazy_fix)  // the pointer walk reads plain sign bits: the flagged ones are recomputed first
            HIPCHK(zh_launch_score_unc_fix(c->dQ, (uint32_t)B, d, f.planes, f.consts, ix->n_planes, c->wBits.as<uint32_t>(), c->wUnc.as<uint32_t>(), c->wpq, s));
        HIPCHK(zh_launch_walk_emit(f, c->dQ, (uint32_t)B, d, (int32_t)k, c->wBits.as<uint32_t>(), c->wpq, c->P_dense,
                                   c->wCounts.as<ZhPairCounts>(), c->wInline.as<ZhVisit>(), c->wRowBase.as<uint64_t>(),
                                   c->wCandBase.as<uint64_t>(), c->wVisitBase.as<uint64_t>(), c->wVisits.as<ZhVisit>(),
                                   c->wLeafCount.as<uint32_t>(), c->wLeafFill.as<uint32_t>(), c->wGroupBase.as<uint32_t>(),
                                   c->wGroupRowBase.as<uint64_t>(), c->wGroups.as<ZhGroup>(), c->wGroupRowOff.as<uint64_t>(), s));
        const uint64_t pairs = (uint64_t)B * T;
        const uint64_t want = tot.visits / (ZH_LOG_CHUNK - 1) + std::min<uint64_t>(pairs, tot.visits / (ZH_INLINE_VISITS + 1)) + 16;
        static const bool fixed = getenv("ZH_WALK_LOG_FIXED") != nullptr;  // tests: keep the log small, always fall back
        if (!fixed && want > c->log_chunks) {
            const size_t chunks = (size_t)(want + want / 2);
            // the pool is idle: the counting pass has completed (its totals are here) and the emit walk does not use it
            if (c->wLogPool.ensure(chunks * ZH_LOG_CHUNK * sizeof(uint2)) == ZH_OK) c->log_chunks = chunks;
        }
    }
    c->scan = choose_scan(ix, tot, c->metric, B, k);
    if (c->scan && (!ix->row_leaf_valid || ix->row_leaf_rows != ix->n_rows)) {  // first table scan since the trees (or the row count) changed
        std::lock_guard<std::mutex> lk(ix->blk_mu);
        if ((!ix->row_leaf_valid || ix->row_leaf_rows != ix->n_rows) && !ix->row_leaf_failed && (rc = build_row_leaf(ix))) return rc;
        if (!ix->row_leaf_valid || ix->row_leaf_rows != ix->n_rows) c->scan = false;
    }
    if (c->scan) {
        const size_t nn = std::max<uint32_t>(ix->n_nodes, 1);
        if ((rc = c->wVisitBits.ensure((nn + 31) / 32 * 4))) return rc;
        if ((rc = c->wNodeVisit.ensure(nn * sizeof(uint4)))) return rc;
        HIPCHK(zh_launch_node_visits(c->wLeafCount.as<uint32_t>(), c->wGroupBase.as<uint32_t>(), c->wGroups.as<ZhGroup>(), ix->n_nodes,
                                     c->wVisitBits.as<uint32_t>(), c->wNodeVisit.as<uint4>(), s));
    }
    if (!no_wave_table && !c->scan)
        HIPCHK(zh_launch_wave_groups(c->wGroups.as<ZhGroup>(), c->wGroupRowOff.as<uint64_t>(), tot.groups, c->wWaveGroup.as<uint32_t>(), s));
    c->approx = c->scan && use_approx(ix, tot, B, k, c->metric);
    c->approx_leaf = false; c->approx_bytes = false;
    if (!c->scan && !no_wave_table && use_approx_leaf(ix, tot, B, k, c->metric)) {
        bool ok = false;
        // rows of bytes (integers 0 .. 255: SIFT descriptors) are copied as bytes -- half the sweep's bytes, no rounding -- for the lean kernels;
        // ZH_S128H_BYTES=0 (read per batch: tests) keeps the copy of halves
        const char *be = getenv("ZH_S128H_BYTES"), *ke = getenv("ZH_S128H_KERNEL"), *de = getenv("ZH_S128H_DMA");
        const bool want_bytes = !(be && be[0] == '0') && !(ke && ke[0] == 'r') && !(de && de[0] == '1');
        std::lock_guard<std::mutex> lk(ix->blk_mu);
        if ((rc = ensure_row_half128(ix, &ok, want_bytes))) return rc;
        c->approx = c->approx_leaf = ok;
        c->approx_bytes = ok && ix->h128_bytes;
    }
    ZhApprox ap{};
    // the scan on the matrix cores, from an fp16 copy of the stored rows (+50 % of the row table, made on first use); no room for it, or mode 5:
    // the VALU kernel on the f32 rows
    bool mfma = c->approx && !c->approx_leaf && mfma_wanted(ix) && B < (1u << 23) - 1;  // (the column pass packs query + 1 into 23 bits of an LDS word)
    const uint2 *scan_row_leaf = ix->row_leaf.as<uint2>();
    if (mfma) {
        std::lock_guard<std::mutex> lk(ix->blk_mu);
        if ((rc = ensure_row_half(ix, &mfma))) return rc;
        if (mfma && ix->perm_rows) {
            if ((rc = ensure_row_leaf_p(ix))) return rc;
            if (!ix->perm_rows) {  // no room for the table in the scan's order: the copy again, in id order, and no further attempt
                ix->row_order_off = true;
                ix->scale_gen = 0;
                if ((rc = ensure_row_half(ix, &mfma))) return rc;
            } else
                scan_row_leaf = ix->row_leaf_p.as<uint2>();
        }
    }
    c->approx_mfma = mfma;
    c->approx_f32rows = mfma && ix->row_half.p == nullptr;
    if (c->approx) {
        // per query: what final_interval_kernel's sort holds (48 KB per query of HBM).  A visit hands on the rows its intervals cannot
        // rule out: a few more than `take` -- or dozens more where the keys are dense around the cut: the parity cosine key on iid
        // rows, whose "nearest" are the similarities just above zero, sends ~100 rows per visit of a 20k-row leaf.  The exact path
        // for visits that take fewer than top_k rows: its table and key scratch sized from the batch
        const char *cap_e = getenv("ZH_APX_CAPS");  // tests: "capq,ex_cap,ex_rows" -- lists and tables that run over (read per batch)
        unsigned e_capq = 0, e_excap = 0, e_exrows = 0;
        if (cap_e) sscanf(cap_e, "%u,%u,%u", &e_capq, &e_excap, &e_exrows);
        const uint32_t capq = e_capq ? std::min(e_capq, 8192u) : (ix->approx_strikes.load() ? 8192u : 4096u);
        const uint32_t ex_cap = e_excap ? e_excap : (uint32_t)std::min<uint64_t>(std::max<uint64_t>(4096, tot.visits / 4 + 1024), 1u << 24);
        const uint32_t ex_rows = e_exrows ? e_exrows : (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1u << 20, tot.rows / 16), 1u << 26);
        if ((rc = c->wQh.ensure(B * d * 2)) || (rc = c->wQmeta.ensure(B * sizeof(float4))) || (rc = c->wApList.ensure((size_t)B * capq * 12)) ||
            (rc = c->wApCount.ensure(B * 8 + std::max<uint64_t>(tot.visits, 1) * 4)) || (rc = c->wApEx.ensure((size_t)ex_cap * 8)) || (rc = c->wApExKeys.ensure((size_t)ex_rows * 20)) ||
            (rc = c->wApCtl.ensure(ZH_APX_CTL_WORDS * 4)))
            return rc;
        ap.Qh = c->wQh.p; ap.qmeta = c->wQmeta.as<float4>(); ap.iv = c->wKeys.as<uint64_t>();
        ap.list_lo = c->wApList.as<uint32_t>(); ap.list_hi = ap.list_lo + (size_t)B * capq; ap.list_id = ap.list_hi + (size_t)B * capq;
        ap.qcount = c->wApCount.as<uint32_t>(); ap.capq = capq; ap.qtau = ap.qcount + B; ap.tauv = ap.qtau + B;
        ap.ex_visits = c->wApEx.as<uint2>(); ap.ex_cap = ex_cap;
        ap.ex_keys = c->wApExKeys.as<uint64_t>(); ap.ex_ckeys = ap.ex_keys + ex_rows; ap.ex_cids = reinterpret_cast<uint32_t *>(ap.ex_ckeys + ex_rows);
        ap.ex_rows_cap = ex_rows; ap.ctl = c->wApCtl.as<uint32_t>();
        ap.n_queries = (uint32_t)B; ap.iv_cap = tot.rows;
        ap.mfma = mfma ? 1u : 0u; ap.row_half = mfma ? ix->row_half.p : nullptr; ap.row_meta = mfma ? ix->row_meta.as<float2>() : nullptr;  // (row_half null with mfma: the scan converts the f32 rows itself)
        ap.row_rho = mfma ? ix->row_rho : 0.f;
        if (c->approx_leaf) { ap.mfma = 2u; ap.row_rho = ap.rho_norm = c->approx_bytes ? 0.f : ix->h128_rho; }
        // (a row of bytes: |x|^2 <= 128 * 255^2 exactly, approx_interval's nx = sqrtf of it * (1 + 1e-5) < 2885.1; ZH_S128H_PRETEST=0, read per batch: tests)
        const char *pte = getenv("ZH_S128H_PRETEST");
        ap.nx_max = c->approx_leaf && c->approx_bytes && !(pte && pte[0] == '0') ? 2885.1f : 0.f;
        HIPCHK(hipMemsetAsync(c->wApCount.p, 0, B * 4, s));
        HIPCHK(hipMemsetAsync(ap.qtau, 0xFF, B * 4, s));
        HIPCHK(hipMemsetAsync(c->wApCtl.p, 0, ZH_APX_CTL_WORDS * 4, s));
        HIPCHK(zh_launch_qhalf(c->dQ, (uint32_t)B, d, c->wQh.p, c->wQmeta.as<float4>(), c->approx_leaf ? 2 : (mfma ? 1 : 0), s));
    }
    HIPCHK(hipEventRecord(c->ev[2], s));
    // the HBM-bound sweep may run on a different ("heavy") stream shared by all contexts, so that sweeps of
    // successive batches execute back to back while everything else overlaps them on the contexts' own streams
    hipStream_t hs = heavy ? heavy : s;
    if (hs != s) { HIPCHK(hipEventRecord(c->ev_emit, s)); HIPCHK(hipStreamWaitEvent(hs, c->ev_emit, 0)); }
    HIPCHK(hipEventRecord(c->ev_sw0, hs));
    // d = 128 leaf by leaf at half width: FUSED (intervals, bounds and the queries' lists inside the sweep: no raw pairs, no select pass) when the
    // leaves are long -- a bound is derived per 64-row chunk that lies inside one leaf group, and the lists must hold what the first waves of a visit
    // hand on before any bound has spread -- and top_k fits a chunk.  ZH_S128H_FUSED=0 / 1 (read per batch: tests) forbids / forces it;
    // zh_debug_keep_raw wants the raw pairs
    c->approx_fused = false;
    if (c->approx_leaf && k <= 64 && !ix->debug_keep_raw) {
        const char *fe = getenv("ZH_S128H_FUSED"), *ke = getenv("ZH_S128H_KERNEL"), *de = getenv("ZH_S128H_DMA");
        const bool lean = !(ke && ke[0] == 'r') && !(de && de[0] == '1');
        c->approx_fused = lean && (fe ? fe[0] == '1' : (!ix->fused_off && tot.rows >= 1024 * std::max<uint64_t>(tot.visits, 1)));
    }
    const int kinda = c->metric != ZH_COSINE ? 0 : (c->mode == ZH_COSINE_PARITY ? 2 : 1);
    if (c->approx_leaf)
        HIPCHK(zh_launch_sweep128h(ix->row_half128.p, c->wQh.p, ldexpf(1.f, ix->h128_ex - 14), c->wGroups.as<ZhGroup>(), c->wGroupRowOff.as<uint64_t>(),
                                   tot.groups, c->wWaveGroup.as<uint32_t>(), f.leaf_ids, tot.group_rows, c->wKeys.as<uint64_t>(), hs,
                                   c->approx_fused ? &ap : nullptr, kinda, (uint32_t)k, zh_approx_bound(c->metric, d, 2), c->approx_bytes));
    else if (c->approx)
        HIPCHK(zh_launch_scan_approx(ix->X.as<float>(), d, ix->n_rows, ap, mfma ? scan_row_leaf : ix->row_leaf.as<uint2>(), T, c->wVisitBits.as<uint32_t>(),
                                     c->wNodeVisit.as<uint4>(), c->wGroups.as<ZhGroup>(), f.group, c->metric, c->mode, hs));
    else if (c->scan)
        HIPCHK(zh_launch_scan_sweep(ix->X.as<float>(), d, ix->n_rows, c->dQ, c->wQQ.as<float>(), ix->row_leaf.as<uint2>(), T,
                                    c->wVisitBits.as<uint32_t>(), c->wNodeVisit.as<uint4>(), c->wGroups.as<ZhGroup>(), f.group,
                                    c->metric, c->mode, c->wKeys.as<uint64_t>(), nullptr, hs));
    else
        HIPCHK(zh_launch_sweep(ix->X.as<float>(), d, c->dQ, c->wQQ.as<float>(), c->wGroups.as<ZhGroup>(),
                               c->wGroupRowOff.as<uint64_t>(), tot.groups, no_wave_table ? nullptr : c->wWaveGroup.as<uint32_t>(), f.leaf_ids,
                               tot.group_rows, c->metric, c->mode, c->wKeys.as<uint64_t>(), f.group, hs));
    HIPCHK(hipEventRecord(c->ev_sw1, hs));
    if (hs != s) HIPCHK(hipStreamWaitEvent(s, c->ev_sw1, 0));
    HIPCHK(hipEventRecord(c->ev[3], s));
    const uint32_t *run_if = nullptr;
    c->dbg_valid = c->approx;
    c->dbg_raw = false;
    if (c->approx) {
        c->dbg_ap = ap;
        if (ix->debug_keep_raw && tot.rows) {  // tests: the scan's raw pairs, before select_tau_kernel turns them into intervals in place
            if ((rc = c->wRaw.ensure(tot.rows * 8))) return rc;
            HIPCHK(hipMemcpyAsync(c->wRaw.p, c->wKeys.p, tot.rows * 8, hipMemcpyDeviceToDevice, s));
            c->dbg_raw = true;
        }
        if (c->approx_fused) HIPCHK(zh_launch_exact_register(c->wVisits.as<ZhVisit>(), tot.visits, (uint32_t)k, ap, s));
        else HIPCHK(zh_launch_select_interval(c->wVisits.as<ZhVisit>(), tot.visits, (uint32_t)k, f.leaf_ids, ap, c->metric, c->mode, d, s));
        HIPCHK(hipEventRecord(c->ev[4], s));
        HIPCHK(zh_launch_final_interval(c->wVisits.as<ZhVisit>(), ix->X.as<float>(), d, c->dQ, c->wQQ.as<float>(), (uint32_t)B, (uint32_t)k,
                                        f.leaf_ids, c->metric, c->mode, ix->opt.id_base, ap, dOutIds, dOutKeys, dOutCounts, ix->max_leaf_len, s));
        HIPCHK(hipMemcpyAsync(c->h_ap, c->wApCtl.p, ZH_APX_CTL_WORDS * 4, hipMemcpyDeviceToHost, s));
        // A list or table ran over (ctl[1] != 0): the f32 scan, select and final, enqueued here with that word as their predicate,
        // redo the batch in stream order -- kernels that return at once otherwise.  On the context's own stream: the shared sweep
        // stream must not wait for this batch's light kernels.
        run_if = ap.ctl + 1;
        if (c->approx_leaf)
            HIPCHK(zh_launch_sweep(ix->X.as<float>(), d, c->dQ, c->wQQ.as<float>(), c->wGroups.as<ZhGroup>(), c->wGroupRowOff.as<uint64_t>(), tot.groups,
                                   c->wWaveGroup.as<uint32_t>(), f.leaf_ids, tot.group_rows, c->metric, c->mode, c->wKeys.as<uint64_t>(), f.group, s, run_if));
        else
            HIPCHK(zh_launch_scan_sweep(ix->X.as<float>(), d, ix->n_rows, c->dQ, c->wQQ.as<float>(), ix->row_leaf.as<uint2>(), T,
                                        c->wVisitBits.as<uint32_t>(), c->wNodeVisit.as<uint4>(), c->wGroups.as<ZhGroup>(), f.group,
                                        c->metric, c->mode, c->wKeys.as<uint64_t>(), run_if, s));
    }
    HIPCHK(zh_launch_select(c->wVisits.as<ZhVisit>(), tot.visits, f.leaf_ids, c->wKeys.as<uint64_t>(),
                            c->wCandKeys.as<uint64_t>(), c->wCandIds.as<uint32_t>(), ix->max_leaf_len, run_if, s));
    if (!c->approx) HIPCHK(hipEventRecord(c->ev[4], s));
    HIPCHK(zh_launch_final(c->wCandBase.as<uint64_t>(), (uint32_t)B, T, (uint32_t)k, c->wCandKeys.as<uint64_t>(),
                           c->wCandIds.as<uint32_t>(), ix->opt.id_base, dOutIds, dOutKeys, dOutCounts, run_if, s));
    for (size_t j = 0; j < nwin && nwin > 1; j++) {
        HIPCHK(hipMemcpyAsync(outIds[j], dOutIds + j * bwin * k, bwin * k * 8, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(outKeys[j], dOutKeys + j * bwin * k, bwin * k * 8, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(outCounts[j], dOutCounts + j * bwin, bwin * 4, hipMemcpyDeviceToDevice, s));
    }
    HIPCHK(hipEventRecord(c->ev[5], s));
    c->state = 2;
    return ZH_OK;
}

// block until the batch's results are complete; fold its counters and stage timings into the index's stats
int ctx_wait(zh_search_ctx *c) {
    zh_index *ix = c->ix;
    if (c->state == 1) return fail(ZH_ESTATE, "zh_search_wait: the batch was begun but not finished");
    if (c->state != 2) return ZH_OK;
    c->state = 0;
    HIPCHK(hipEventSynchronize(c->ev[5]));
    if (c->trivial) return ZH_OK;
    const bool pf = c->prefilter;
    const uint32_t pf_amb = pf ? c->h_pf[0] : 0, pf_flags = pf ? c->h_pf[1] : 0, pf_exact = pf ? c->h_pf[2] : 0;
    if (pf_flags) {  // a candidate list (or the ambiguous-visit table) ran over: this batch again, swept the classic way
        {
            std::lock_guard<std::mutex> lk(ix->stats_mu);
            ix->stats.prefilter_fallbacks_accum++;
            ix->stats.prefilter_last_overflow = pf_flags;
        }
        ix->prefilter_strikes.fetch_add(1);
        const std::vector<const float *> q = c->sv_q;
        const std::vector<uint64_t *> ids = c->sv_ids, keys = c->sv_keys;
        const std::vector<uint32_t *> counts = c->sv_counts;
        c->prefilter_off_once = true;
        int rc = ctx_begin(c, q.data(), c->nwin, c->bwin, c->k, c->metric, c->mode, c->s);
        if (!rc) rc = ctx_finish(c, ids.data(), keys.data(), counts.data(), c->sv_heavy);
        c->prefilter_off_once = false;
        if (rc) return rc;
        return ctx_wait(c);
    }
    if (pf) ix->prefilter_strikes = 0;
    const ZhTotals tot = c->tot;
    uint64_t uniq = 0;
    if (ix->profiling >= 2 && tot.visits && !pf) {  // R_unique: rows of the distinct leaves touched by the batch
        std::vector<ZhVisit> hv(tot.visits);
        HIPCHK(hipMemcpy(hv.data(), c->wVisits.p, tot.visits * sizeof(ZhVisit), hipMemcpyDeviceToHost));
        std::vector<std::pair<uint32_t, uint32_t>> leaves(hv.size());
        for (size_t i = 0; i < hv.size(); i++) leaves[i] = {hv[i].leaf_off, hv[i].len};
        std::sort(leaves.begin(), leaves.end());
        leaves.erase(std::unique(leaves.begin(), leaves.end()), leaves.end());
        for (auto &l : leaves) uniq += l.second;
    }
    float ms[5] = {0, 0, 0, 0, 0};
    if (ix->profiling > 0) {
        for (int i = 0; i < 5; i++) HIPCHK(hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]));
        HIPCHK(hipEventElapsedTime(&ms[2], c->ev_sw0, c->ev_sw1));  // the sweep kernel alone, on the stream it ran on
    }
    std::lock_guard<std::mutex> lk(ix->stats_mu);
    zh_stats_t &st = ix->stats;
    st.batch = c->B; st.window_batches = c->nwin; st.visits = tot.visits; st.rows_scored = tot.rows; st.candidates = tot.takes;
    st.planes_dense = c->P_dense; st.planes_total = ix->n_planes;
    const uint64_t swept = pf ? pf_exact : (c->scan ? ix->n_rows : tot.group_rows);
    st.rows_swept = swept;
    st.sweep_bytes = pf ? tot.rows * 64 + (uint64_t)pf_exact * 4 * ix->opt.dim  // one 64-byte sector of the score table per scored row
                        : (c->scan ? ix->n_rows * ((uint64_t)4 * ix->opt.dim + 8 * ix->n_trees) + tot.rows * 8
                                   : tot.group_rows * ((uint64_t)4 * ix->opt.dim + 4) + tot.rows * 8);
    st.table_scan = c->scan && !pf ? 1 : 0;
    const bool apx = c->approx && !pf;
    st.approx_scan = apx ? (c->approx_leaf ? 3 : (c->approx_mfma ? 2 : 1)) : 0;
    st.approx_fused = apx && c->approx_leaf && c->approx_fused ? 1 : 0;
    st.approx_byte_rows = apx && c->approx_leaf && c->approx_bytes ? 1 : 0;
    st.approx_exact_visits = apx ? c->h_ap[0] : 0;
    st.approx_survivors = apx ? c->h_ap[3] : 0;
    st.approx_list_entries = apx ? c->h_ap[4] : 0;
    if (apx && c->h_ap[7]) fprintf(stderr, "zebra_hip: scan guard tripped, bits %u (diagnostic build)\n", c->h_ap[7]);
    st.approx_columns = apx && c->approx_mfma ? c->h_ap[5] : 0;
    st.approx_column_pairs = apx && c->approx_mfma ? c->h_ap[6] : 0;
    if (apx && c->h_ap[1]) {
        st.approx_fallbacks_accum++;
        st.approx_last_overflow = c->h_ap[1];
        if (c->approx_mfma && c->approx_f32rows && ix->approx_strikes.load() >= 1) ix->mfma_f32_off = true;  // (8192-slot lists ran over: the VALU scan next)
        else if (c->approx_leaf && c->approx_fused && !ix->fused_off) ix->fused_off = true;                     // (the fused sweep's lists ran over: the select pass next)
        else ix->approx_strikes.fetch_add(1);
    } else if (apx && (uint64_t)c->h_ap[4] > (uint64_t)c->B * (ix->approx_strikes.load() ? 5600 : 2800))  // lists 70 % full on average: some query's will run over
        ix->approx_strikes.fetch_add(1);
    st.prefiltered = pf ? 1 : 0;
    st.prefilter_exact_visits = pf_amb;
    st.prefilter_exact_rows = pf_exact;
    st.hash_from_scores = c->score_hash ? 1 : 0;
    st.hash_exact_fixups = c->score_hash ? tot.hash_fixups : 0;
    if (ix->profiling >= 2) st.rows_unique = uniq;
    if (ix->profiling > 0) {
        st.ms_hash += ms[0]; st.ms_walk += ms[1]; st.ms_sweep += ms[2]; st.ms_select += ms[3]; st.ms_final += ms[4];
        st.ms_total += ms[0] + ms[1] + ms[2] + ms[3] + ms[4];
        st.timed_batches++;
        st.sweep_rows_accum += tot.rows;
        st.swept_rows_accum += swept;
        // (the half-width d = 128 sweep's lean kernels take two or four times the rows per launch: 256- or 128-byte rows -- launch_sweep128h_lean)
        const char *ke = getenv("ZH_S128H_KERNEL"), *de = getenv("ZH_S128H_DMA");
        const bool lean128 = apx && c->approx_leaf && !(ke && ke[0] == 'r') && !(de && de[0] == '1');
        const uint64_t rpl = apx && c->approx_leaf ? zh_sweep128h_rows_per_launch(lean128, c->approx_bytes) : zh_sweep_rows_per_launch(ix->opt.dim);
        st.sweep_launches_accum += (swept + rpl - 1) / rpl;
        st.scan_batches_accum += c->scan && !pf ? 1 : 0;
        st.approx_batches_accum += apx ? 1 : 0;
    }
    return ZH_OK;
}

static int search_once(zh_index *ix, const float *dQ, size_t B, size_t k, int metric, int mode, uint64_t *dOutIds,
                       uint64_t *dOutKeys, uint32_t *dOutCounts, hipStream_t s, zh_search_ctx *c) {
    int rc = ctx_begin(c, &dQ, 1, B, k, metric, mode, s);
    if (rc) return rc;
    if ((rc = ctx_finish(c, &dOutIds, &dOutKeys, &dOutCounts, nullptr))) return rc;
    return ctx_wait(c);
}

// The blocking entry points take any batch: a batch that would pass a per-launch limit (batch * num_trees < 2^26 pairs,
// <= 2^28 - 1 leaf visits, < 2^36 scored rows -- reached by a few thousand queries under the reference's default options,
// where a query visits ~10^4..10^5 leaves) is split into sub-batches whose outputs land side by side.  The
// first split is sized from the visits per pair seen so far; a sub-batch that still overflows is halved and retried.
static int search_locked(zh_index *ix, const float *dQ, size_t B, size_t k, int metric, int mode, uint64_t *dOutIds,
                         uint64_t *dOutKeys, uint32_t *dOutCounts, hipStream_t s, zh_search_ctx *c = nullptr) {
    if (!c) c = &ix->dctx;
    const uint32_t T = ix->n_trees, d = ix->opt.dim;
    if (B == 0 || T == 0) return search_once(ix, dQ, B, k, metric, mode, dOutIds, dOutKeys, dOutCounts, s, c);
    size_t chunk = std::min<size_t>(B, ((1ull << 26) - 1) / T);
    {
        std::lock_guard<std::mutex> lk(ix->stats_mu);
        if (ix->visits_per_pair > 0) {
            const double per_query = ix->visits_per_pair * T;
            chunk = std::min<size_t>(chunk, (size_t)std::max(1.0, (double)(1ull << 27) / per_query));  // half the cap: headroom
        }
    }
    if (chunk == 0) chunk = 1;
    for (size_t b0 = 0; b0 < B;) {
        const size_t nb = std::min(chunk, B - b0);
        const int rc = search_once(ix, dQ + b0 * d, nb, k, metric, mode, dOutIds + b0 * k, dOutKeys + b0 * k, dOutCounts + b0, s, c);
        if (rc == ZH_ELIMIT && nb > 1) { chunk = (nb + 1) / 2; continue; }  // the context is idle again: retry smaller
        if (rc) return rc;
        b0 += nb;
    }
    return ZH_OK;
}

static int check_metric(int metric, int mode) {
    if (metric < ZH_COSINE || metric > ZH_PNORM) return fail(ZH_EINVAL, "unknown metric %d", metric);
    if (metric == ZH_COSINE && mode != ZH_COSINE_PARITY && mode != ZH_COSINE_CORRECTED) return fail(ZH_EINVAL, "unknown cosine mode %d", mode);
    // ZH_MINKOWSKI / ZH_PNORM: `mode` is the struct's i32 power and every value is legal (distance.rs:160-190; Default = 0)
    return ZH_OK;
}

extern "C" int zh_search_batch_device(zh_index *ix, const float *d_q, size_t b, size_t k, int metric, int mode,
                                      uint64_t *d_out_ids, uint64_t *d_out_keys, uint32_t *d_out_counts, void *stream) {
    if (!ix || (b && (!d_q || !d_out_counts || (k && (!d_out_ids || !d_out_keys))))) return fail(ZH_EINVAL, "zh_search_batch_device: null argument");
    if (k > ZH_MAX_TOPK) return fail(ZH_ELIMIT, "top_k %zu > ZH_MAX_TOPK (%u)", k, ZH_MAX_TOPK);
    int rc = check_metric(metric, mode);
    if (rc) return rc;
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    if ((rc = set_device(ix))) return rc;
    return search_locked(ix, d_q, b, k, metric, mode, d_out_ids, d_out_keys, d_out_counts,
                         stream ? (hipStream_t)stream : ix->stream);
}

// ---- pipelined form: several batches in flight, each on its own context and stream ----------------------
extern "C" int zh_search_ctx_create(zh_index *ix, zh_search_ctx **out) {
    if (!ix || !out) return fail(ZH_EINVAL, "zh_search_ctx_create: null argument");
    int rc = set_device(ix);
    if (rc) return rc;
    zh_search_ctx *c = new (std::nothrow) zh_search_ctx();
    if (!c) return fail(ZH_ENOMEM, "out of host memory");
    if ((rc = ctx_init(c, ix))) { c->release_all(); delete c; return rc; }
    *out = c;
    return ZH_OK;
}
extern "C" void zh_search_ctx_destroy(zh_search_ctx *c) {
    if (!c) return;
    hipSetDevice(c->ix->device);
    if (c->state == 2) hipEventSynchronize(c->ev[5]);
    else if (c->state == 1 && c->s) hipStreamSynchronize(c->s);
    c->release_all();
    delete c;
}
extern "C" int zh_search_begin(zh_search_ctx *c, const float *d_q, size_t b, size_t k, int metric, int mode, void *stream) {
    if (!c || (b && !d_q)) return fail(ZH_EINVAL, "zh_search_begin: null argument");
    if (k > ZH_MAX_TOPK) return fail(ZH_ELIMIT, "top_k %zu > ZH_MAX_TOPK (%u)", k, ZH_MAX_TOPK);
    int rc = check_metric(metric, mode);
    if (rc) return rc;
    if ((rc = set_device(c->ix))) return rc;
    return ctx_begin(c, &d_q, 1, b, k, metric, mode, stream ? (hipStream_t)stream : c->ix->stream);
}
extern "C" int zh_search_begin_window(zh_search_ctx *c, const float *const *d_q, size_t n_batches, size_t b, size_t k, int metric,
                                      int mode, void *stream) {
    if (!c || !d_q || n_batches == 0 || n_batches > ZH_MAX_WINDOW) return fail(ZH_EINVAL, "zh_search_begin_window: bad argument (1..%u batches)", ZH_MAX_WINDOW);
    for (size_t j = 0; j < n_batches; j++)
        if (b && !d_q[j]) return fail(ZH_EINVAL, "zh_search_begin_window: null query pointer");
    if (k > ZH_MAX_TOPK) return fail(ZH_ELIMIT, "top_k %zu > ZH_MAX_TOPK (%u)", k, ZH_MAX_TOPK);
    int rc = check_metric(metric, mode);
    if (rc) return rc;
    if ((rc = set_device(c->ix))) return rc;
    return ctx_begin(c, d_q, n_batches, b, k, metric, mode, stream ? (hipStream_t)stream : c->ix->stream);
}
extern "C" int zh_search_finish(zh_search_ctx *c, uint64_t *d_out_ids, uint64_t *d_out_keys, uint32_t *d_out_counts,
                                void *sweep_stream) {
    if (!c) return fail(ZH_EINVAL, "zh_search_finish: null context");
    if (c->B && (!d_out_counts || (c->k && (!d_out_ids || !d_out_keys)))) return fail(ZH_EINVAL, "zh_search_finish: null output");
    if (c->nwin != 1) return fail(ZH_ESTATE, "zh_search_finish: the context holds a window; use zh_search_finish_window");
    int rc = set_device(c->ix);
    if (rc) return rc;
    return ctx_finish(c, &d_out_ids, &d_out_keys, &d_out_counts, (hipStream_t)sweep_stream);
}
extern "C" int zh_search_finish_window(zh_search_ctx *c, uint64_t *const *d_out_ids, uint64_t *const *d_out_keys,
                                       uint32_t *const *d_out_counts, void *sweep_stream) {
    if (!c || !d_out_ids || !d_out_keys || !d_out_counts) return fail(ZH_EINVAL, "zh_search_finish_window: null argument");
    for (size_t j = 0; j < c->nwin && c->bwin; j++)
        if (!d_out_counts[j] || (c->k && (!d_out_ids[j] || !d_out_keys[j]))) return fail(ZH_EINVAL, "zh_search_finish_window: null output");
    int rc = set_device(c->ix);
    if (rc) return rc;
    return ctx_finish(c, d_out_ids, d_out_keys, d_out_counts, (hipStream_t)sweep_stream);
}
double zh_index_visits_per_pair(zh_index *ix) {
    std::lock_guard<std::mutex> lk(ix->stats_mu);
    return ix->visits_per_pair;
}
void zh_search_ctx_stream_ordered(zh_search_ctx *c) {
    if (c) c->stream_ordered = true;
}
void zh_search_ctx_abandon(zh_search_ctx *c) {
    if (c && c->state == 1) c->state = 0;  // what was enqueued by begin completes on its stream; its results are never used
}
extern "C" int zh_search_wait(zh_search_ctx *c) {
    if (!c) return fail(ZH_EINVAL, "zh_search_wait: null context");
    int rc = set_device(c->ix);
    if (rc) return rc;
    return ctx_wait(c);
}

// ---- test / debug access (include/zebra_hip.h, "test / debug access") ----
extern "C" int zh_trim_device_memory(void) {
    pool_trim();
    return ZH_OK;
}

extern "C" int zh_debug_keep_raw(zh_index *ix, int on) {
    if (!ix) return fail(ZH_EINVAL, "zh_debug_keep_raw: null index");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    ix->debug_keep_raw = on != 0;
    return ZH_OK;
}
extern "C" int zh_debug_scan_pairs(zh_index *ix, zh_search_ctx *ctx, zh_debug_scan_info *info, zh_debug_pair *out, size_t cap, float *qmeta) {
    if (!ix || !info || (cap && !out)) return fail(ZH_EINVAL, "zh_debug_scan_pairs: null argument");
    std::unique_lock<std::shared_mutex> lk(ix->mu, std::defer_lock);
    if (!ctx) lk.lock();
    zh_search_ctx *c = ctx ? ctx : &ix->dctx;
    if (c->ix != ix) return fail(ZH_EINVAL, "zh_debug_scan_pairs: the context belongs to another index");
    if (c->state != 0) return fail(ZH_ESTATE, "zh_debug_scan_pairs: the context has a batch in flight");
    int rc = set_device(ix);
    if (rc) return rc;
    memset(info, 0, sizeof(*info));
    if (!c->dbg_valid) return ZH_OK;  // (approx_scan 0: the last batch was not a half-width one)
    if (c->approx_leaf && c->approx_fused)  // (a fused sweep leaves no interval per pair anywhere: zh_debug_keep_raw keeps the sweep unfused)
        return fail(ZH_ESTATE, "zh_debug_scan_pairs: the batch's sweep was fused (no per-pair results exist); call zh_debug_keep_raw(idx, 1) first");
    const ZhTotals tot = c->tot;
    const ZhApprox &ap = c->dbg_ap;
    info->approx_scan = c->approx_leaf ? 3 : (c->approx_mfma ? 2 : 1);
    info->queries = (uint32_t)c->B; info->top_k = (uint32_t)c->k; info->metric = c->metric; info->cosine_mode = c->mode;
    info->raw_kept = c->dbg_raw ? 1 : 0; info->overflow = c->h_ap[1];
    info->bound_const = zh_approx_bound(c->metric, ix->opt.dim, (int)ap.mfma);
    info->row_rho = ap.row_rho; info->rho_norm = ap.rho_norm;
    info->pairs = tot.rows; info->visits = tot.visits;
    HIPCHK(hipDeviceSynchronize());
    if (qmeta) HIPCHK(hipMemcpy(qmeta, ap.qmeta, c->B * sizeof(float4), hipMemcpyDeviceToHost));
    if (!cap) return ZH_OK;
    if (cap < tot.rows) return fail(ZH_ELIMIT, "zh_debug_scan_pairs: %llu pairs, room for %zu", (unsigned long long)tot.rows, cap);
    std::vector<ZhVisit> hv(tot.visits);
    std::vector<uint64_t> iv(tot.rows), raw(c->dbg_raw ? tot.rows : 0);
    std::vector<uint32_t> lids(ix->n_leaf_ids);
    if (tot.visits) HIPCHK(hipMemcpy(hv.data(), c->wVisits.p, tot.visits * sizeof(ZhVisit), hipMemcpyDeviceToHost));
    if (tot.rows) HIPCHK(hipMemcpy(iv.data(), c->wKeys.p, tot.rows * 8, hipMemcpyDeviceToHost));
    if (c->dbg_raw && tot.rows) HIPCHK(hipMemcpy(raw.data(), c->wRaw.p, tot.rows * 8, hipMemcpyDeviceToHost));
    if (ix->n_leaf_ids) HIPCHK(hipMemcpy(lids.data(), ix->leaf_ids.p, ix->n_leaf_ids * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tot.rows; i++) { out[i].row = out[i].query = 0xFFFFFFFFu; out[i].flags = 0; out[i].visit = 0xFFFFFFFFu; }
    for (size_t v = 0; v < hv.size(); v++) {
        const ZhVisit &z = hv[v];
        if (z.row_off + z.len > tot.rows || (uint64_t)z.leaf_off + z.len > ix->n_leaf_ids) return fail(ZH_ESTATE, "zh_debug_scan_pairs: visit %zu out of range", v);
        const uint32_t fl = (z.take ? 1u : 0u) | ((z.take && z.take < z.len && z.take < c->k) ? 2u : 0u);
        for (uint32_t i = 0; i < z.len; i++) {
            zh_debug_pair &o = out[z.row_off + i];
            o.row = lids[(size_t)z.leaf_off + i]; o.query = z.b; o.visit = (uint32_t)v; o.flags = fl;
            const uint64_t w = iv[z.row_off + i];
            o.lo = (uint32_t)w; o.hi = (uint32_t)(w >> 32);
            o.raw_s = o.raw_a2 = 0.f;
            if (c->dbg_raw) { const uint64_t r = raw[z.row_off + i]; uint32_t a = (uint32_t)r, b2 = (uint32_t)(r >> 32); memcpy(&o.raw_s, &a, 4); memcpy(&o.raw_a2, &b2, 4); }
        }
    }
    return ZH_OK;
}

// ---- a large HOST-resident batch (zh_search_batch; Database::query_vectors -> the shim's search_batch, core.rs:290-313): windows over two contexts ----
// The device-pointer context API keeps two windows in flight (bench.py's loop: one sweeping, the next in its light phases, the copies beside both);
// a blocking host-pointer call used to run H2D -> one internal batch -> D2H back to back (132 k against 185-192 k QPS at
// cfg3).  Here the call cuts its batch into windows of `wq` queries (the size at which the half-width scan's query halves still fit the L2s beside
// the streamed rows), alternates them between the lane's two contexts -- light kernels on the contexts' own high-priority streams, the sweeps back
// to back on the index's sweep stream -- stages queries and results through pinned memory window by window, and hands out results as windows
// complete.  Same ids / keys / counts as one batch: the queries of a batch never interact.  Only in the few-visits-per-pair regime (long leaves):
// the wandering walk of small-leaf forests has its own limits per internal batch (search_locked splits by visits) and stays on the classic path.
// Queries per window of a large host batch.  Few visits per (query, tree) pair (leaves >= top_k): what keeps the scan's query halves in the L2s, by
// dimension.  The wandering walk of small-leaf forests (the reference's default options: thousands of visits per pair): what keeps the row-score
// hash's score table (n_rows x window x 4 bytes per context) at 1 GiB -- 256 queries at 1M rows; the classic path's own split there is by the visit
// log alone (~2000 queries: a 7.8-GB score table per chunk, one chunk at a time: 7 k QPS where windows over two contexts reach the pipelined rate).
static size_t host_window_queries(const zh_index *ix, bool wander) {
    const char *e = getenv("ZH_HOST_WINDOW");  // tests / A-B, read per call (unset or 0: by regime and dimension)
    const size_t forced = e ? (size_t)atoll(e) : (size_t)0;
    if (forced) return forced;
    const uint32_t d = ix->opt.dim;
    if (wander) {
        const size_t w = ((size_t)1 << 28) / std::max<uint64_t>(ix->n_rows, 1) / 64 * 64;
        return std::min<size_t>(std::max<size_t>(w, 64), 1024);
    }
    return d >= 512 ? 2048 : (d >= 256 ? 1024 : 4096);
}
// the window size for this call, or 0: the classic path (one internal batch, split only by its own limits)
static size_t host_windows_wanted(zh_index *ix, size_t B, size_t k, bool *wander) {
    const bool off = getenv("ZH_NO_HOST_WINDOWS") != nullptr;  // (read per call: tests switch it)
    if (off || ix->n_trees == 0 || ix->n_rows == 0) return 0;
    double vpp;
    {
        std::lock_guard<std::mutex> lk(ix->stats_mu);
        vpp = ix->visits_per_pair;  // (known from earlier batches)
    }
    // the first batch of an index: leaves below top_k wander (the guess choose_dense_planes makes too: lsh.rs:131-138's defaults against any
    // top_k >= 2) -- windows at once, the one-chunk classic path is four times slower there; otherwise the classic way, once
    if (!(vpp > 0)) {
        if ((size_t)ix->opt.max_node_size >= 2 * k + 2) return 0;
        vpp = 1000.0;
    }
    const size_t wq = host_window_queries(ix, vpp > 4.0);
    if (B < 2 * wq && B < wq + wq / 2) return 0;
    *wander = vpp > 4.0;
    return wq;
}
// returns ZH_OK with every result in the caller's buffers, or an error after which BOTH contexts are idle and nothing is in flight (the caller
// then runs the classic path: a window that passes a per-batch limit is not an error of the call)
#ifndef ZH_HOST_AHEAD_WANDER
#define ZH_HOST_AHEAD_WANDER 2   // windows begun ahead in the wandering regime (profiles/r06_host_calls_refdefault.txt: 1M x 384, default options, queries per
                                 // second at 1024 / 4096 / 8192 queries per call: none 17.2 / 17.4 / 17.4 k, one 23.1 / 25.1 / 25.5 k, two 23.7 / 28.4 / 29.1 k)
#endif
static int search_host_windows(zh_index *ix, zh_index::Lane &ln, size_t wq, int ahead, const float *q, size_t B, size_t k, int metric, int mode, uint64_t *out_ids,
                               uint64_t *out_keys, uint32_t *out_counts) {
    const uint32_t d = ix->opt.dim;
    int rc;
    auto init_ctx = [&](zh_search_ctx *c, hipStream_t *st, bool *flag) -> int {
        if (*flag) return ZH_OK;
        int r = ctx_init(c, ix);
        if (r) { c->release_all(); return r; }
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        hipError_t e = hipStreamCreateWithPriority(st, hipStreamNonBlocking, hi);
        for (auto &ev : ln.ev_d2h)
            if (e == hipSuccess && !ev) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) {
            c->release_all();
            if (*st) { hipStreamDestroy(*st); *st = nullptr; }
            return fail(ZH_EHIP, "host windows: %s", hipGetErrorString(e));
        }
        *flag = true;
        return ZH_OK;
    };
    if ((rc = init_ctx(&ln.ctx2, &ln.s2, &ln.init2))) return rc;
    if (ahead >= 1 && init_ctx(&ln.ctx3, &ln.s3, &ln.init3) != ZH_OK) ahead = 0;  // (no room for a third context: two, begin + finish back to back)
    if (ahead >= 2 && init_ctx(&ln.ctx4, &ln.s4, &ln.init4) != ZH_OK) ahead = 1;
    const int NC = 2 + ahead;                     // contexts the windows rotate through
    const size_t A = (size_t)ahead;               // windows begun ahead of the one being finished
    const size_t nw = (B + wq - 1) / wq, per = (B + nw - 1) / nw;
    const size_t off_ids = B * d * 4, off_keys = off_ids + B * k * 8, off_counts = off_keys + B * k * 8, need = off_counts + B * 4;
    if (need > ln.h_stage_cap) {
        if (ln.h_stage) hipHostFree(ln.h_stage);
        ln.h_stage = nullptr; ln.h_stage_cap = 0;
        const size_t cap = need + need / 2;
        hipError_t e = hipHostMalloc(&ln.h_stage, cap, hipHostMallocDefault);
        if (e != hipSuccess) return fail(ZH_ENOMEM, "hipHostMalloc(%zu): %s", cap, hipGetErrorString(e));
        ln.h_stage_cap = cap;
    }
    uint8_t *hs = static_cast<uint8_t *>(ln.h_stage);
    float *dQ = ln.wQ.as<float>();
    uint64_t *dIds = ln.wOutIds.as<uint64_t>(), *dKeys = ln.wOutKeys.as<uint64_t>();
    uint32_t *dCounts = ln.wOutCounts.as<uint32_t>();
    uint64_t *hIds = reinterpret_cast<uint64_t *>(hs + off_ids), *hKeys = reinterpret_cast<uint64_t *>(hs + off_keys);
    uint32_t *hCounts = reinterpret_cast<uint32_t *>(hs + off_counts);
    zh_search_ctx *ctxs[4] = {&ln.ctx, &ln.ctx2, &ln.ctx3, &ln.ctx4};
    hipStream_t str[4] = {ln.s, ln.s2, ln.s3, ln.s4};
    auto win = [&](size_t w, size_t *b0, size_t *nb) { *b0 = w * per; *nb = std::min(per, B - *b0); };
    auto bail = [&](int code) {  // leave nothing in flight: what was begun is abandoned, what was finished is retired, every stream drained
        const std::string why = g_err;
        for (int i = 0; i < NC; i++) {
            zh_search_ctx_abandon(ctxs[i]);
            if (ctxs[i]->state == 2) ctx_wait(ctxs[i]);
            hipStreamSynchronize(str[i]);
        }
        hipStreamSynchronize(ix->sweep_stream);
        g_err = why;
        return code;
    };
    auto retire = [&](size_t w) -> int {  // window w: wait (its outputs are complete only then), results -> pinned staging, behind them an event
        const int i = (int)(w % NC);
        size_t b0, nb;
        win(w, &b0, &nb);
        int r = ctx_wait(ctxs[i]);
        if (r) return r;
        hipError_t e = hipSuccess;
        if (k) {
            e = hipMemcpyAsync(hIds + b0 * k, dIds + b0 * k, nb * k * 8, hipMemcpyDeviceToHost, str[i]);
            if (e == hipSuccess) e = hipMemcpyAsync(hKeys + b0 * k, dKeys + b0 * k, nb * k * 8, hipMemcpyDeviceToHost, str[i]);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(hCounts + b0, dCounts + b0, nb * 4, hipMemcpyDeviceToHost, str[i]);
        if (e == hipSuccess) e = hipEventRecord(ln.ev_d2h[i], str[i]);
        return e == hipSuccess ? ZH_OK : fail(ZH_EHIP, "host windows, D2H: %s", hipGetErrorString(e));
    };
    auto hand_out = [&](size_t w) -> int {  // window w's results, once its copies have landed, into the caller's buffers
        size_t b0, nb;
        win(w, &b0, &nb);
        hipError_t e = hipEventSynchronize(ln.ev_d2h[w % NC]);
        if (e != hipSuccess) return fail(ZH_EHIP, "host windows: %s", hipGetErrorString(e));
        if (k) { memcpy(out_ids + b0 * k, hIds + b0 * k, nb * k * 8); memcpy(out_keys + b0 * k, hKeys + b0 * k, nb * k * 8); }
        memcpy(out_counts + b0, hCounts + b0, nb * 4);
        return ZH_OK;
    };
    // window w is BEGUN at step w (its context: that of window w - NC, retired first; that window's results are handed out one step later, before
    // its pinned event is recorded again) and FINISHED at step w + A
    for (size_t step = 0; step < nw + A; step++) {
        if (step < nw) {
            const size_t w = step;
            const int i = (int)(w % NC);
            size_t b0, nb;
            win(w, &b0, &nb);
            if (w >= (size_t)NC) {
                if (w >= (size_t)NC + 1 && (rc = hand_out(w - NC - 1))) return bail(rc);  // (its copies were queued one step ago: landed long since)
                if ((rc = retire(w - NC))) return bail(rc);
            }
            memcpy(hs + b0 * d * 4, q + b0 * d, nb * d * 4);  // (pageable -> pinned: the copy engine then runs beside the other windows' kernels)
            hipError_t e = hipMemcpyAsync(dQ + b0 * d, hs + b0 * d * 4, nb * d * 4, hipMemcpyHostToDevice, str[i]);
            if (e != hipSuccess) return bail(fail(ZH_EHIP, "host windows, H2D: %s", hipGetErrorString(e)));
            const float *dq = dQ + b0 * d;
            if ((rc = ctx_begin(ctxs[i], &dq, 1, nb, k, metric, mode, str[i]))) return bail(rc);
        }
        if (step >= A) {
            const size_t w = step - A;
            size_t b0, nb;
            win(w, &b0, &nb);
            uint64_t *oi = dIds + b0 * k, *ok = dKeys + b0 * k;
            uint32_t *oc = dCounts + b0;
            if ((rc = ctx_finish(ctxs[w % NC], &oi, &ok, &oc, ix->sweep_stream))) return bail(rc);
        }
    }
    // what is still in flight: the last NC windows were never retired, the one before them not handed out
    const size_t first_unretired = nw > (size_t)NC ? nw - NC : 0;
    if (first_unretired >= 1 && (rc = hand_out(first_unretired - 1))) return bail(rc);
    for (size_t w = first_unretired; w < nw; w++) {
        if ((rc = retire(w))) return bail(rc);
        if ((rc = hand_out(w))) return bail(rc);
    }
    return ZH_OK;
}

// one round of the combining front end: the requests of `grp` (same top_k, metric, mode) as ONE internal batch on lane `ln`
static void run_group(zh_index *ix, zh_index::Lane &ln, const std::vector<zh_index::CombineReq *> &grp) {
    auto all_fail = [&](int rc) {
        const std::string why = g_err;
        for (auto *r : grp) { r->rc = rc; r->err = why; }
    };
    std::shared_lock<std::shared_mutex> lk(ix->mu);
    int rc = set_device(ix);
    if (rc) return all_fail(rc);
    hipError_t e;
    auto hip_fail = [&](hipError_t err, const char *what) { all_fail(fail(ZH_EHIP, "%s: %s", what, hipGetErrorString(err))); };
    if (!ln.init) {  // (`init` only once BOTH the context and its stream exist; a partial lane is taken down again)
        if ((rc = ctx_init(&ln.ctx, ix))) { ln.ctx.release_all(); return all_fail(rc); }
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);  // (hi = the numerically lowest = highest priority: the light kernels' pool, as the pipelined callers')
        if ((e = hipStreamCreateWithPriority(&ln.s, hipStreamNonBlocking, hi)) != hipSuccess) { ln.ctx.release_all(); ln.s = nullptr; return hip_fail(e, "hipStreamCreate"); }
        ln.init = true;
    }
    hipStream_t s = ln.s;
    const uint32_t d = ix->opt.dim;
    const size_t k = grp[0]->k;
    size_t B = 0;
    for (auto *r : grp) B += r->b;
    if ((rc = ln.wQ.ensure(B * d * 4)) || (rc = ln.wOutIds.ensure(std::max<size_t>(B * k, 1) * 8)) ||
        (rc = ln.wOutKeys.ensure(std::max<size_t>(B * k, 1) * 8)) || (rc = ln.wOutCounts.ensure(B * 4)))
        return all_fail(rc);
    bool wander = false;
    const size_t wq = grp.size() == 1 ? host_windows_wanted(ix, B, k, &wander) : 0;
    if (wq) {  // one large batch: windows over two contexts (three, one begun ahead, where the walk wanders), copies beside the kernels
        const char *la_e = getenv("ZH_HOST_LOOKAHEAD");  // tests / A-B, read per call: 0 never, 1 / 2 always, that many windows begun ahead
        const int ahead = la_e ? (la_e[0] == '2' ? 2 : (la_e[0] == '1' ? 1 : 0)) : (wander ? ZH_HOST_AHEAD_WANDER : 0);
        if (search_host_windows(ix, ln, wq, ahead, grp[0]->q, B, k, grp[0]->metric, grp[0]->mode, grp[0]->ids, grp[0]->keys, grp[0]->counts) == ZH_OK) {
            std::lock_guard<std::mutex> ls(ix->stats_mu);
            ix->stats.host_window_calls_accum++;
            return;
        }
        // (a window passed a per-batch limit, or an allocation failed: nothing is in flight; the classic path below decides what the call returns)
    }
    const bool staged = grp.size() > 1;  // a lone caller's buffers are used as they are
    uint8_t *hs = nullptr;
    const size_t off_ids = B * d * 4, off_keys = off_ids + B * k * 8, off_counts = off_keys + B * k * 8, need = off_counts + B * 4;
    if (staged) {
        if (need > ln.h_stage_cap) {
            if (ln.h_stage) hipHostFree(ln.h_stage);
            ln.h_stage = nullptr; ln.h_stage_cap = 0;
            const size_t cap = need + need / 2;
            if ((e = hipHostMalloc(&ln.h_stage, cap, hipHostMallocDefault)) != hipSuccess) { all_fail(fail(ZH_ENOMEM, "hipHostMalloc(%zu): %s", cap, hipGetErrorString(e))); return; }
            ln.h_stage_cap = cap;
        }
        hs = static_cast<uint8_t *>(ln.h_stage);
        size_t o = 0;
        for (auto *r : grp) { memcpy(hs + o, r->q, r->b * d * 4); o += r->b * d * 4; }
        if ((e = hipMemcpyAsync(ln.wQ.p, hs, B * d * 4, hipMemcpyHostToDevice, s)) != hipSuccess) return hip_fail(e, "H2D");
    } else if ((e = hipMemcpyAsync(ln.wQ.p, grp[0]->q, B * d * 4, hipMemcpyHostToDevice, s)) != hipSuccess)
        return hip_fail(e, "H2D");
    rc = search_locked(ix, ln.wQ.as<float>(), B, k, grp[0]->metric, grp[0]->mode, ln.wOutIds.as<uint64_t>(), ln.wOutKeys.as<uint64_t>(),
                       ln.wOutCounts.as<uint32_t>(), s, &ln.ctx);
    if (rc) return all_fail(rc);
    uint64_t *hi = staged ? reinterpret_cast<uint64_t *>(hs + off_ids) : grp[0]->ids;
    uint64_t *hk = staged ? reinterpret_cast<uint64_t *>(hs + off_keys) : grp[0]->keys;
    uint32_t *hc = staged ? reinterpret_cast<uint32_t *>(hs + off_counts) : grp[0]->counts;
    e = hipSuccess;
    if (k) {
        e = hipMemcpyAsync(hi, ln.wOutIds.p, B * k * 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(hk, ln.wOutKeys.p, B * k * 8, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hc, ln.wOutCounts.p, B * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return hip_fail(e, "D2H");
    if (staged) {
        size_t b0 = 0;
        for (auto *r : grp) {
            if (k) { memcpy(r->ids, hi + b0 * k, r->b * k * 8); memcpy(r->keys, hk + b0 * k, r->b * k * 8); }
            memcpy(r->counts, hc + b0, r->b * 4);
            b0 += r->b;
        }
        std::lock_guard<std::mutex> ls(ix->stats_mu);
        ix->stats.combined_batches_accum++;
        ix->stats.combined_calls_accum += grp.size();
    }
}

extern "C" int zh_search_batch(zh_index *ix, const float *q, size_t b, size_t k, int metric, int mode,
                               uint64_t *out_ids, uint64_t *out_keys, uint32_t *out_counts) {
    if (!ix || (b && (!q || !out_counts || (k && (!out_ids || !out_keys))))) return fail(ZH_EINVAL, "zh_search_batch: null argument");
    if (k > ZH_MAX_TOPK) return fail(ZH_ELIMIT, "top_k %zu > ZH_MAX_TOPK (%u)", k, ZH_MAX_TOPK);
    int rc = check_metric(metric, mode);
    if (rc) return rc;
    if (b == 0) return ZH_OK;
    zh_index::CombineReq me;
    me.q = q; me.b = b; me.k = k; me.metric = metric; me.mode = mode; me.ids = out_ids; me.keys = out_keys; me.counts = out_counts;
    bool lead;
    {
        std::lock_guard<std::mutex> lk(ix->cmu);
        ix->cpend.push_back(&me);
        lead = !ix->cleader;  // nobody leads: this caller does, at once
        if (lead) ix->cleader = true;
    }
    for (;;) {
        if (!lead) {  // sleep until my request is done, or until a finishing leader hands me the lead
            std::unique_lock<std::mutex> l(me.m);
            me.cv.wait(l, [&] { return me.done || me.lead; });
            if (me.done) break;
            me.lead = false;
        }
        // ---- lead one round ----
        std::unique_lock<std::mutex> lk(ix->cmu);
        if (ix->clast >= 2) {
            // The round before served a crowd whose threads are on their way back with their next requests.  Taking the queue as it
            // stands would split a crowd of N into two cohorts of N / 2 that take turns (one is served while the other queues), and a
            // round costs about the same for 30 queries as for 60: wait until the arrivals pause (nothing new for ~10 us), 120 us at most.
            // (polled with yields: sleep_for(10 us) takes ~60 us with the default timer slack)
            const auto t0 = std::chrono::steady_clock::now();
            auto tq = t0;
            size_t n0 = ix->cpend.size();
            for (;;) {
                lk.unlock();
                std::this_thread::yield();
                lk.lock();
                const auto now = std::chrono::steady_clock::now();
                if (ix->cpend.size() != n0) { n0 = ix->cpend.size(); tq = now; }
                if (now - tq > std::chrono::microseconds(10) || now - t0 > std::chrono::microseconds(120)) break;
            }
        }
        // the head of the queue and everything behind it that can share its batch (same top_k, metric, mode; arrival order kept)
        // (nothing between taking requests off the queue and handing the lead on may throw across the extern "C" boundary or leave
        // `cleader` set with nobody leading: the list's room is reserved BEFORE anything is taken, and a round that throws fails its requests)
        std::vector<zh_index::CombineReq *> grp;
        try {
            grp.reserve(ix->cpend.size());
        } catch (...) {  // no memory to even list the round: my own request fails, the lead goes to whoever heads the queue without me
            for (auto it = ix->cpend.begin(); it != ix->cpend.end(); ++it)
                if (*it == &me) { ix->cpend.erase(it); break; }
            me.rc = ZH_ENOMEM;
            if (ix->cpend.empty()) ix->cleader = false;
            else {
                zh_index::CombineReq *nx = ix->cpend.front();
                std::lock_guard<std::mutex> l(nx->m);
                nx->lead = true;
                nx->cv.notify_one();
            }
            break;
        }
        const zh_index::CombineReq *head = ix->cpend.front();  // (never empty: the leader's own request is in it until served)
        size_t total = 0;
        for (auto it = ix->cpend.begin(); it != ix->cpend.end();) {
            zh_index::CombineReq *r = *it;
            if (r->k == head->k && r->metric == head->metric && r->mode == head->mode && (grp.empty() || total + r->b <= 8192)) {
                grp.push_back(r);
                total += r->b;
                it = ix->cpend.erase(it);
            } else
                ++it;
        }
        lk.unlock();
        try {
            run_group(ix, ix->lanes[0], grp);
        } catch (...) {  // (bad_alloc in a message copy or a host vector: the round's requests fail, the queue lives on)
            for (auto *r : grp) r->rc = ZH_ENOMEM;
        }
        bool mine = false;
        for (auto *r : grp) {
            if (r == &me) { mine = true; continue; }
            std::lock_guard<std::mutex> l(r->m);  // (notify under the lock: the owner destroys r as soon as it sees done)
            r->done = true;
            r->cv.notify_one();
        }
        lk.lock();
        ix->clast = grp.size();
        if (!mine) { lk.unlock(); lead = true; continue; }  // my own request waits in the queue (it could not share the head's batch): lead on
        // served: hand the lead to the head of the queue, or retire it
        if (ix->cpend.empty()) ix->cleader = false;
        else {
            zh_index::CombineReq *nx = ix->cpend.front();
            std::lock_guard<std::mutex> l(nx->m);
            nx->lead = true;
            nx->cv.notify_one();
        }
        break;
    }
    if (me.rc) return fail(me.rc, "%s", me.err.empty() ? "zh_search_batch: out of host memory" : me.err.c_str());
    return ZH_OK;
}

extern "C" int zh_hash_signs(zh_index *ix, const float *q, size_t b, uint32_t *out_bits, float *out_dots) {
    if (!ix || (b && (!q || !out_bits))) return fail(ZH_EINVAL, "zh_hash_signs: null argument");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    int rc = set_device(ix);
    if (rc) return rc;
    const uint32_t P = ix->n_planes, d = ix->opt.dim;
    if (!b || !P) return ZH_OK;
    hipStream_t s = ix->stream;
    const uint32_t wpq = (P + 255) / 256 * 8, words = (P + 31) / 32;
    DevBuf dq, dbits, ddots;
    struct G { DevBuf *a, *b, *c; ~G() { a->release(); b->release(); c->release(); } } g{&dq, &dbits, &ddots};
    size_t chunk = std::max<size_t>(1, std::min<size_t>(b, (size_t)(256u << 20) / ((size_t)P * 4 + 1)));
    if (chunk >= 4) chunk &= ~(size_t)3;  // (the row-score path takes queries four at a time)
    if ((rc = dq.ensure(chunk * d * 4))) return rc;
    if ((rc = dbits.ensure(chunk * wpq * 4))) return rc;
    if (out_dots && (rc = ddots.ensure(chunk * P * 4))) return rc;
    std::vector<uint32_t> hb(chunk * wpq);
    for (size_t b0 = 0; b0 < b; b0 += chunk) {
        size_t nb = std::min(chunk, b - b0);
        HIPCHK(hipMemcpyAsync(dq.p, q + b0 * d, nb * d * 4, hipMemcpyHostToDevice, s));
        // zh_set_hash_mode(2): the signs (not the dots) through the row-score path where the forest allows -- the whole sign matrix of
        // both paths can then be compared bit for bit (tests/test_gpu_score_hash.py)
        if (!out_dots && ix->hash_mode == 2 && use_score_hash(ix, nb, P)) {
            zh_search_ctx *c = &ix->dctx;
            if (c->state == 2 && (rc = ctx_wait(c))) return rc;
            c->wpq = wpq;
            if ((rc = c->wTotals.ensure(sizeof(ZhTotals))) || (rc = c->wBits.ensure(nb * wpq * 4))) return rc;
            HIPCHK(hipMemsetAsync(c->wTotals.p, 0, sizeof(ZhTotals), s));
            if ((rc = launch_score_hash(c, dq.as<float>(), nb, false, s))) return rc;
            HIPCHK(hipMemcpyAsync(hb.data(), c->wBits.p, nb * wpq * 4, hipMemcpyDeviceToHost, s));
            ZhTotals ht;
            HIPCHK(hipMemcpyAsync(&ht, c->wTotals.p, sizeof ht, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            std::lock_guard<std::mutex> lks(ix->stats_mu);
            ix->stats.hash_from_scores = 1;
            ix->stats.hash_exact_fixups = ht.hash_fixups;
        } else {
            HIPCHK(zh_launch_hash_dense(dq.as<float>(), (uint32_t)nb, ix->planes.as<float>(), ix->consts.as<float>(), P, d,
                                        dbits.as<uint32_t>(), wpq, out_dots ? ddots.as<float>() : nullptr, s));
            HIPCHK(hipMemcpyAsync(hb.data(), dbits.p, nb * wpq * 4, hipMemcpyDeviceToHost, s));
        }
        if (out_dots) HIPCHK(hipMemcpyAsync(out_dots + b0 * P, ddots.p, nb * P * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (size_t i = 0; i < nb; i++) {
            memcpy(out_bits + (b0 + i) * words, &hb[i * wpq], words * 4);
            if (P & 31) out_bits[(b0 + i) * words + words - 1] &= (1u << (P & 31)) - 1;
        }
    }
    return ZH_OK;
}

// ------------------------------------------------------------------------------------------------
// stand-alone metric calls (src/distance.rs), multi-GPU merge, synthetic queries
// ------------------------------------------------------------------------------------------------
static int pick_device(int device) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(ZH_EHIP, "no usable HIP device; this library has no CPU fallback");
    if (device >= ndev) return fail(ZH_EINVAL, "device %d out of range", device);
    if (device >= 0) { e = hipSetDevice(device); if (e != hipSuccess) return fail(ZH_EHIP, "hipSetDevice: %s", hipGetErrorString(e)); }
    return ZH_OK;
}

// Stand-alone metric calls keep their device buffers per calling thread (and device): Metric::distance is called pair by
// pair from the crate's own code paths, and six hipMalloc / hipFree per call cost far more than the kernels.
struct DistScratch {
    int device = -1;
    DevBuf in, keys, small;  // [rows | query] staged together, the keys, the sweep's one-group record
    void drop() { in.release(); keys.release(); small.release(); device = -1; }
    ~DistScratch() { drop(); }
};
static thread_local DistScratch g_dist;

extern "C" int zh_distance_batch(int metric, int mode, const float *a, const float *q, size_t n, size_t dim,
                                 uint64_t *out_keys, int device) {
    if ((n && (!a || !out_keys)) || !q || !dim) return fail(ZH_EINVAL, "zh_distance_batch: null argument");
    if (n > 0xFFFFFFFFull) return fail(ZH_ELIMIT, "zh_distance_batch: n > 2^32-1");
    int rc = check_metric(metric, mode);
    if (rc) return rc;
    if ((rc = pick_device(device))) return rc;
    if (!n) return ZH_OK;
    int cur = 0;
    HIPCHK(hipGetDevice(&cur));
    DistScratch &sc = g_dist;
    if (sc.device != cur) { sc.drop(); sc.device = cur; }
    if ((rc = sc.in.ensure((n + 1) * dim * 4))) return rc;
    if ((rc = sc.keys.ensure(n * 8))) return rc;
    if ((rc = sc.small.ensure(ZH_DISTANCE_SCRATCH_BYTES))) return rc;
    float *da = sc.in.as<float>(), *dq = da + n * dim;
    HIPCHK(hipMemcpyAsync(da, a, n * dim * 4, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(dq, q, dim * 4, hipMemcpyHostToDevice, nullptr));
    HIPCHK(zh_launch_distance_rows(da, n, (uint32_t)dim, dq, metric, mode, sc.keys.as<uint64_t>(), sc.small.p, nullptr));
    HIPCHK(hipMemcpy(out_keys, sc.keys.p, n * 8, hipMemcpyDeviceToHost));  // synchronises the null stream
    return ZH_OK;
}

extern "C" int zh_distance_pair(int metric, int mode, const float *a, const float *b, size_t dim, uint64_t *out_key, int device) {
    return zh_distance_batch(metric, mode, a, b, 1, dim, out_key, device);
}

extern "C" int zh_merge_topk_device(int device, uint32_t n_shards, size_t b, size_t k, const uint64_t *d_ids,
                                    const uint64_t *d_keys, const uint32_t *d_counts, uint64_t *d_out_ids,
                                    uint64_t *d_out_keys, uint32_t *d_out_counts, void *stream) {
    if (b && (!d_ids || !d_keys || !d_counts || !d_out_ids || !d_out_keys || !d_out_counts)) return fail(ZH_EINVAL, "zh_merge_topk_device: null argument");
    if (k == 0 || k > ZH_MAX_TOPK) return fail(ZH_ELIMIT, "top_k must be in 1..%u", ZH_MAX_TOPK);
    if (n_shards == 0 || n_shards > 1024) return fail(ZH_EINVAL, "n_shards must be in 1..1024");
    int rc = pick_device(device);
    if (rc) return rc;
    HIPCHK(zh_launch_merge(n_shards, (uint32_t)b, (uint32_t)k, d_ids, d_keys, d_counts, d_out_ids, d_out_keys, d_out_counts, 0, 0, (hipStream_t)stream));
    return ZH_OK;  // enqueued on `stream`; the caller synchronises
}

// one buffer per shard: [ids b*k u64][keys b*k u64][counts b u32, padded to 8 bytes] -- what ONE all-gather of
// every rank's packed result produces
extern "C" size_t zh_packed_result_words(size_t b, size_t k) { return 2 * b * k + (b + 1) / 2; }
extern "C" int zh_merge_topk_packed_device(int device, uint32_t n_shards, size_t b, size_t k, const uint64_t *d_packed,
                                           uint64_t *d_out_ids, uint64_t *d_out_keys, uint32_t *d_out_counts, void *stream) {
    if (b && (!d_packed || !d_out_ids || !d_out_keys || !d_out_counts)) return fail(ZH_EINVAL, "zh_merge_topk_packed_device: null argument");
    if (k == 0 || k > ZH_MAX_TOPK) return fail(ZH_ELIMIT, "top_k must be in 1..%u", ZH_MAX_TOPK);
    if (n_shards == 0 || n_shards > 1024) return fail(ZH_EINVAL, "n_shards must be in 1..1024");
    int rc = pick_device(device);
    if (rc) return rc;
    const uint64_t words = zh_packed_result_words(b, k);
    HIPCHK(zh_launch_merge(n_shards, (uint32_t)b, (uint32_t)k, d_packed, d_packed + b * k,
                           reinterpret_cast<const uint32_t *>(d_packed + 2 * b * k), d_out_ids, d_out_keys, d_out_counts,
                           words, 2 * words, (hipStream_t)stream));
    return ZH_OK;
}

extern "C" int zh_synth_queries_device(int device, float *d_out, uint64_t seed_rows, uint64_t seed_q, uint64_t n_rows,
                                       uint64_t b0, size_t b, uint32_t dim, int kind, void *stream) {
    if (b && !d_out) return fail(ZH_EINVAL, "null output");
    if (!n_rows) return fail(ZH_EINVAL, "n_rows must be > 0");
    int rc = pick_device(device);
    if (rc) return rc;
    HIPCHK(zh_launch_synth_queries(d_out, seed_rows, seed_q, n_rows, b0, b, dim, kind, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return ZH_OK;
}
