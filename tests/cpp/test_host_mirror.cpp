// Parity test of the C++ host mirror (include/zebra.hpp) written the way a test of the reference crate
// would read: build a Database / LSHIndex, insert records, search, compare with expected neighbours.
// Expected values come from the CPU oracle and are handed over in a binary fixture written by
// tests/test_cpp_mirror.py:  u32 n,d,M,T,k,B ; u64 seed ; f32 X[n*d] ; f32 Q[B*d] ;
//   for metric in {L2SQ, COSINE parity}: u32 counts[B] ; u64 ids[B*k] ; u64 keys[B*k]
//   u64 pair_l2sq, pair_cos_parity, pair_cos_corrected, pair_l2   (distance(X[0], Q[0]))
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "zebra.hpp"

constexpr std::size_t N = 64;
using namespace zebra;

template <class T>
static std::vector<T> rd(FILE *f, std::size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short fixture\n"); exit(2); }
    return v;
}

#define EXPECT(c)                                                                   \
    do {                                                                            \
        if (!(c)) { fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); fails++; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s fixture.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("fixture"); return 2; }
    auto h = rd<std::uint32_t>(f, 6);
    const std::uint32_t n = h[0], d = h[1], M = h[2], T = h[3], k = h[4], B = h[5];
    if (d != N) { fprintf(stderr, "fixture dim %u != %zu\n", d, N); return 2; }
    const std::uint64_t seed = rd<std::uint64_t>(f, 1)[0];
    auto X = rd<float>(f, (std::size_t)n * d), Q = rd<float>(f, (std::size_t)B * d);
    int fails = 0;

    std::vector<Embedding<N>> rows(n), queries(B);
    for (std::uint32_t i = 0; i < n; i++) for (std::size_t c = 0; c < N; c++) rows[i][c] = X[(std::size_t)i * d + c];
    for (std::uint32_t i = 0; i < B; i++) for (std::size_t c = 0; c < N; c++) queries[i][c] = Q[(std::size_t)i * d + c];

    LSHIndexOptions<N> opt;
    opt.max_node_size = M;
    opt.num_trees = T;
    LSHIndex<N> index(opt, seed);
    EXPECT(index.is_empty() && index.no_vectors() && index.no_trees());
    auto ids = index.add(rows);  // build_index
    EXPECT(ids.size() == n && ids.front() == 0 && ids.back() == n - 1);
    EXPECT(!index.is_empty());
    LSHIndex<N> clone = index;  // Clone shares the store
    EXPECT(!clone.no_vectors());

    for (int which = 0; which < 2; which++) {
        auto counts = rd<std::uint32_t>(f, B);
        auto eids = rd<std::uint64_t>(f, (std::size_t)B * k), ekeys = rd<std::uint64_t>(f, (std::size_t)B * k);
        std::vector<std::vector<std::pair<Id, DistanceUnit>>> got;
        if (which == 0) got = clone.search_batch(queries, k, L2SquaredDistance<N>{});
        else got = clone.search_batch(queries, k, CosineDistance<N>{});
        for (std::uint32_t b = 0; b < B; b++) {
            EXPECT(got[b].size() == counts[b]);
            for (std::uint32_t j = 0; j < counts[b] && j < got[b].size(); j++) {
                EXPECT(got[b][j].first == eids[(std::size_t)b * k + j]);
                EXPECT(got[b][j].second == ekeys[(std::size_t)b * k + j]);
            }
        }
        if (which == 0) {  // the single-query entry point, LSHIndex::search
            auto one = index.search(queries[0], k, L2SquaredDistance<N>{});
            EXPECT(one == got[0]);
        }
    }
    auto pairs = rd<std::uint64_t>(f, 4);
    CosineDistance<N> corrected;
    corrected.parity = false;
    EXPECT(L2SquaredDistance<N>{}.distance(rows[0], queries[0]) == pairs[0]);
    EXPECT(CosineDistance<N>{}.distance(rows[0], queries[0]) == pairs[1]);
    EXPECT(corrected.distance(rows[0], queries[0]) == pairs[2]);
    EXPECT(L2Distance<N>{}.distance(rows[0], queries[0]) == pairs[3]);
    fclose(f);

    // Database::insert_records / query_vectors (core.rs:245-254, 290-313)
    Database<N, L2SquaredDistance<N>> db(opt);
    EXPECT(db.query_vectors(queries, 3).empty());
    std::vector<std::string> docs(n);
    for (std::uint32_t i = 0; i < n; i++) docs[i] = "doc" + std::to_string(i);
    db.insert_records(rows, docs);
    auto res = db.query_vectors({rows[5], rows[77]}, 3);
    EXPECT(res.size() == 2 && res[0].count(5) && res[0][5] == "doc5" && res[1].count(77) && res[0].size() == 3);
    db.remove({5});
    auto res2 = db.query_vectors({rows[5]}, 3);
    EXPECT(res2[0].count(5) == 0);
    db.clear_database();
    EXPECT(db.query_vectors({rows[5]}, 3).empty());

    // LSHIndex::remove / deduplicate (lsh.rs:473-503, 270-288)
    {
        LSHIndex<N> ix2(opt, seed);
        std::vector<Embedding<N>> dup_rows(rows.begin(), rows.begin() + 200);
        dup_rows[150] = dup_rows[3];
        dup_rows[199] = dup_rows[3];
        ix2.add(dup_rows);
        auto removed = ix2.deduplicate();
        EXPECT(removed.size() == 2 && removed[0] == 150 && removed[1] == 199);
        auto gone = ix2.remove({7, 150, 100000});
        EXPECT(gone.size() == 1 && gone[0] == 7);
        auto hit = ix2.search(dup_rows[3], 3, L2SquaredDistance<N>{});
        EXPECT(hit.size() == 3 && hit[0].first == 3 && hit[0].second == 0 && hit[1].second != 0);
    }

    // error behaviour: a limit violation surfaces as zebra::Error (anyhow::Error in the crate)
    bool threw = false;
    try { index.search(queries[0], ZH_MAX_TOPK + 1, L2SquaredDistance<N>{}); } catch (const Error &e) { threw = e.code == ZH_ELIMIT; }
    EXPECT(threw);
    index.clear();
    EXPECT(index.is_empty());
    if (fails == 0) printf("cpp host mirror: ok (%u rows, %u queries, k=%u)\n", n, B, k);
    return fails ? 1 : 0;
}
