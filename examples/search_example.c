/* Plain C99 user of the C ABI (include/zebra_hip.h): insert vectors, search a batch, grow the index, remove.
 *   gcc -std=c99 -Iinclude examples/search_example.c -Lzebra_amd/lib -lzebra_hip -Wl,-rpath,$PWD/zebra_amd/lib -o /tmp/ex && /tmp/ex
 * Mirrors what Database::insert_records / query_vectors do in the reference (src/database/core.rs:245-254, 290-313). */
#include <stdio.h>
#include <stdlib.h>

#include "zebra_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != ZH_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, zh_last_error()); return 1; } \
    } while (0)

static float noise(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

int main(void) {
    enum { N = 20000, D = 384, B = 8, K = 10 };
    zh_options opt;
    zh_options_default(&opt);          /* max_node_size 5, num_trees 15: the reference defaults (lsh.rs:131-138) */
    opt.dim = D;
    opt.max_node_size = 256;
    zh_index *idx = NULL;
    CHECK(zh_index_create(&opt, &idx));

    float *rows = malloc(sizeof(float) * N * D), *q = malloc(sizeof(float) * B * D);
    unsigned seed = 1;
    for (size_t i = 0; i < (size_t)N * D; i++) rows[i] = noise(&seed);
    for (int b = 0; b < B; b++)
        for (int c = 0; c < D; c++) q[b * D + c] = rows[(size_t)(b * 997) * D + c] + 0.05f * noise(&seed);

    CHECK(zh_index_add(idx, rows, N, NULL));             /* first add builds the forest on the GPU */
    uint64_t ids[B * K], keys[B * K];
    uint32_t counts[B];
    CHECK(zh_search_batch(idx, q, B, K, ZH_L2SQ, 0, ids, keys, counts));
    int hits = 0;
    for (int b = 0; b < B; b++) hits += counts[b] > 0 && ids[b * K] == (uint64_t)(b * 997);
    printf("planted neighbour first for %d of %d queries; %llu vectors, %u trees\n", hits, B,
           (unsigned long long)zh_index_count(idx), zh_index_num_trees(idx));

    CHECK(zh_index_add(idx, rows, 100, NULL));           /* trees exist: incremental insert (duplicates of rows 0..99) */
    size_t removed = 0;
    CHECK(zh_index_deduplicate(idx, NULL, 0, &removed)); /* ...which deduplicate finds again */
    uint64_t gone[2] = {997, 123456789};
    uint8_t found[2];
    CHECK(zh_index_remove(idx, gone, 2, found, NULL));
    printf("deduplicate removed %zu, remove found [%d %d], %llu vectors left\n", removed, found[0], found[1],
           (unsigned long long)zh_index_count(idx));
    int ok = hits == B && removed == 100 && found[0] == 1 && found[1] == 0 && zh_index_count(idx) == (uint64_t)N - 1;
    zh_index_destroy(idx);
    free(rows);
    free(q);
    printf(ok ? "example ok\n" : "example FAILED\n");
    return ok ? 0 : 1;
}
