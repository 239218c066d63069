// The reference's calling pattern against the C ABI, without an interpreter lock in the way: NT host threads each issue
// single-query zh_search_batch calls (LSHIndex::search from rayon workers, /root/reference/src/database/core.rs:299-303).
//   test_concurrent <queries.bin> <results.bin> n d M T k NT PER
// queries.bin: f32 Q[NT*PER][d] (written by tests/test_gpu_concurrent.py from the oracle's generator); results.bin: u32 counts,
// u64 ids, u64 keys of every call, compared with the oracle by the caller.  Prints "serial <calls/s> threaded <calls/s> ...".
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "zebra_hip.h"

#define CHECK(x)                                                                              \
    do {                                                                                      \
        int rc_ = (x);                                                                        \
        if (rc_ != ZH_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, zh_last_error()); exit(3); } \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 10) { fprintf(stderr, "usage: %s queries.bin results.bin n d M T k NT PER\n", argv[0]); return 2; }
    const size_t n = strtoull(argv[3], nullptr, 10), d = strtoull(argv[4], nullptr, 10), M = strtoull(argv[5], nullptr, 10),
                 T = strtoull(argv[6], nullptr, 10), k = strtoull(argv[7], nullptr, 10), NT = strtoull(argv[8], nullptr, 10),
                 PER = strtoull(argv[9], nullptr, 10);
    const size_t NQ = NT * PER;
    std::vector<float> Q(NQ * d);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(Q.data(), 4, Q.size(), f) != Q.size()) { fprintf(stderr, "short query file\n"); return 2; }
    fclose(f);
    zh_options opt;
    zh_options_default(&opt);
    opt.dim = (uint32_t)d; opt.max_node_size = (uint32_t)M; opt.num_trees = (uint32_t)T; opt.device = 0; opt.reserve_rows = n;
    opt.seed = 0x5EB2A003ull;
    zh_index *ix = nullptr;
    CHECK(zh_index_create(&opt, &ix));
    CHECK(zh_index_append_synthetic(ix, n, 0x5EB2A001ull, 0, 0));
    CHECK(zh_index_build(ix));
    std::vector<uint64_t> ids(NQ * k), keys(NQ * k);
    std::vector<uint32_t> counts(NQ);
    auto call = [&](size_t i) {
        CHECK(zh_search_batch(ix, Q.data() + i * d, 1, k, ZH_COSINE, ZH_COSINE_PARITY, ids.data() + i * k, keys.data() + i * k, counts.data() + i));
    };
    for (size_t i = 0; i < 64 && i < NQ; i++) call(i);  // warm-up: scratch, the row -> leaf table
    const size_t n_serial = NQ < 512 ? NQ : 512;
    auto t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < n_serial; i++) call(i);
    const double serial = n_serial / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    zh_stats_reset(ix);
    if (getenv("ZH_TEST_PROFILE")) zh_set_profiling(ix, 1);
    std::vector<std::thread> th;
    t0 = std::chrono::steady_clock::now();
    for (size_t t = 0; t < NT; t++)
        th.emplace_back([&, t] { for (size_t j = 0; j < PER; j++) call(t * PER + j); });
    for (auto &x : th) x.join();
    const double threaded = NQ / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    zh_stats_t st;
    CHECK(zh_stats(ix, &st));
    printf("serial %.0f threaded %.0f ratio %.2f combined_batches %llu combined_calls %llu\n", serial, threaded, threaded / serial,
           (unsigned long long)st.combined_batches_accum, (unsigned long long)st.combined_calls_accum);
    if (getenv("ZH_TEST_PROFILE") && st.timed_batches)
        printf("stages per internal batch (ms): hash %.3f walk %.3f sweep %.3f select %.3f final %.3f over %llu batches; table_scan %llu\n",
               st.ms_hash / st.timed_batches, st.ms_walk / st.timed_batches, st.ms_sweep / st.timed_batches, st.ms_select / st.timed_batches,
               st.ms_final / st.timed_batches, (unsigned long long)st.timed_batches, (unsigned long long)st.table_scan);
    f = fopen(argv[2], "wb");
    if (!f) return 2;
    fwrite(counts.data(), 4, NQ, f); fwrite(ids.data(), 8, NQ * k, f); fwrite(keys.data(), 8, NQ * k, f);
    fclose(f);
    zh_index_destroy(ix);
    return 0;
}
