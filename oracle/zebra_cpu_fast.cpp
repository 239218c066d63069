/*
 * zebra_cpu_fast.cpp -- the reference algorithm on the CPU written for SPEED, not for bit-exactness: the second
 * `cpu_baseline` leg of bench.py ("port-fast").
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (same rule as zebra_oracle.c: only tests/, smoke() and bench.py's
 * cpu_baseline leg may load it; the product never does).
 *
 * The bit-exact oracle (zebra_oracle.c) emulates the GPU's summation order with 256 scalar accumulators and a 64-way
 * butterfly per row, qsort()s every leaf and re-scores every candidate: a faithful checker, but far slower than what
 * the reference really executes -- simsimd's SIMD kernels over contiguous rows (distance.rs:23,41,106).  This file is
 * the honest CPU figure: the same algorithm (walk lsh.rs:290-348 incl. the return-value quirk, union + top-k
 * lsh.rs:544-565, keys distance.rs:19-49,103-114) with
 *   - free summation order: straight vectorisable loops with several independent accumulators (-O3, AVX2 or AVX-512
 *     build picked at run time), f32 accumulation like simsimd's SIMD kernels;
 *   - std::nth_element instead of a full sort of every leaf;
 *   - no second scoring pass: the rerank reuses the key computed in the leaf (same function, same arguments);
 *   - vectors and trees in RAM (no fjall, no bincode): favourable to the CPU.
 * Results differ from the oracle only by f32 summation order (<= ~1e-6 relative in the keys; the order of candidates
 * whose keys differ by less than that may swap), which tests/test_cpu_fast.py bounds.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ZF_EXPORT extern "C" __attribute__((visibility("default")))

enum { ZF_COSINE = 0, ZF_L2SQ = 1, ZF_L2 = 2 };

namespace {

struct Forest {
    const int32_t *plane, *left, *right;
    const uint32_t *roots;
    const float *planes, *consts;
    const uint32_t *leaf_ids;
    uint32_t d, T;
};

inline float dot(const float *__restrict a, const float *__restrict b, uint32_t d) {
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    uint32_t i = 0;
    for (; i + 64 <= d; i += 64) {
        float t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma omp simd reduction(+ : t0, t1, t2, t3)
        for (uint32_t j = 0; j < 16; j++) {
            t0 += a[i + j] * b[i + j];
            t1 += a[i + 16 + j] * b[i + 16 + j];
            t2 += a[i + 32 + j] * b[i + 32 + j];
            t3 += a[i + 48 + j] * b[i + 48 + j];
        }
        s0 += t0; s1 += t1; s2 += t2; s3 += t3;
    }
    for (; i < d; i++) s0 += a[i] * b[i];
    return (s0 + s1) + (s2 + s3);
}

inline float l2sq(const float *__restrict a, const float *__restrict b, uint32_t d) {
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    uint32_t i = 0;
    for (; i + 64 <= d; i += 64) {
        float t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma omp simd reduction(+ : t0, t1, t2, t3)
        for (uint32_t j = 0; j < 16; j++) {
            float e0 = a[i + j] - b[i + j], e1 = a[i + 16 + j] - b[i + 16 + j];
            float e2 = a[i + 32 + j] - b[i + 32 + j], e3 = a[i + 48 + j] - b[i + 48 + j];
            t0 += e0 * e0; t1 += e1 * e1; t2 += e2 * e2; t3 += e3 * e3;
        }
        s0 += t0; s1 += t1; s2 += t2; s3 += t3;
    }
    for (; i < d; i++) { float e = a[i] - b[i]; s0 += e * e; }
    return (s0 + s1) + (s2 + s3);
}

inline void cos_sums(const float *__restrict a, const float *__restrict b, uint32_t d, float &ab, float &a2) {
    float p0 = 0, p1 = 0, n0 = 0, n1 = 0;
    uint32_t i = 0;
    for (; i + 32 <= d; i += 32) {
        float t0 = 0, t1 = 0, u0 = 0, u1 = 0;
#pragma omp simd reduction(+ : t0, t1, u0, u1)
        for (uint32_t j = 0; j < 16; j++) {
            t0 += a[i + j] * b[i + j]; t1 += a[i + 16 + j] * b[i + 16 + j];
            u0 += a[i + j] * a[i + j]; u1 += a[i + 16 + j] * a[i + 16 + j];
        }
        p0 += t0; p1 += t1; n0 += u0; n1 += u1;
    }
    for (; i < d; i++) { p0 += a[i] * b[i]; n0 += a[i] * a[i]; }
    ab = p0 + p1; a2 = n0 + n1;
}

inline uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }

struct Scorer {
    const float *X, *q;
    uint32_t d;
    int metric, mode;
    float qq;
    inline uint64_t key(uint32_t id) const {
        const float *a = X + (size_t)id * d;
        if (metric == ZF_COSINE) {
            float ab, a2;
            cos_sums(a, q, d, ab, a2);
            double c;  // simsimd cos(): the distance, clipped at 0, two zero-norm cases; then distance.rs:23-25
            if (a2 == 0.0f && qq == 0.0f) c = 0.0;
            else if (ab == 0.0f) c = 1.0;
            else { c = 1.0 - (double)ab / std::sqrt((double)a2 * (double)qq); if (c < 0.0) c = 0.0; }
            return bits(mode == 0 ? 1.0 - c : c);
        }
        const float s = l2sq(a, q, d);
        return bits(metric == ZF_L2SQ ? (double)s : std::sqrt((double)s));
    }
};

struct Cand { uint64_t key; uint32_t id; };
inline bool cand_less(const Cand &x, const Cand &y) { return x.key != y.key ? x.key < y.key : x.id < y.id; }

struct Ctx {
    const Forest *f;
    Scorer sc;
    std::vector<Cand> leaf, cand;
    uint64_t rows_scored = 0;
};

// tree_result, lsh.rs:290-348 (a leaf shorter than n is inserted whole: scored here, once, instead of in the rerank)
int32_t walk(Ctx &c, int32_t node, int32_t n) {
    const Forest &f = *c.f;
    if (f.plane[node] < 0) {
        const uint32_t off = (uint32_t)f.left[node], len = (uint32_t)f.right[node];
        const uint32_t *ids = f.leaf_ids + off;
        if (n <= 0) return 0;
        c.rows_scored += len;
        if ((int64_t)len < (int64_t)n) {
            for (uint32_t i = 0; i < len; i++) c.cand.push_back({c.sc.key(ids[i]), ids[i]});
            return (int32_t)len;
        }
        c.leaf.resize(len);
        for (uint32_t i = 0; i < len; i++) c.leaf[i] = {c.sc.key(ids[i]), ids[i]};
        if ((uint32_t)n < len) std::nth_element(c.leaf.begin(), c.leaf.begin() + n, c.leaf.end(), cand_less);
        c.cand.insert(c.cand.end(), c.leaf.begin(), c.leaf.begin() + n);
        return n;
    }
    const int32_t p = f.plane[node];
    const bool above = ((double)dot(f.planes + (size_t)p * f.d, c.sc.q, f.d) + (double)f.consts[p]) >= 0.0;  // lsh.rs:39-43
    const int32_t main_n = above ? f.right[node] : f.left[node], backup = above ? f.left[node] : f.right[node];
    const int32_t k = walk(c, main_n, n);
    if (k < n) return walk(c, backup, n - k);  // lsh.rs:341-343: the backup's count alone
    return k;
}

}  // namespace

/* LSHIndex::search for b queries on `nthreads` OpenMP threads (rayon's par_iter over queries, core.rs:299).
 * Forest arrays as zh_index_get_forest returns them; X is n_rows x d in host memory.  out_* are b x k. */
ZF_EXPORT void zf_search_batch(const int32_t *plane, const int32_t *left, const int32_t *right, const uint32_t *roots,
                               uint32_t T, const float *planes, const float *consts, const uint32_t *leaf_ids,
                               const float *X, uint32_t d, const float *Q, uint64_t b, uint32_t k, int metric, int mode,
                               uint64_t *out_ids, uint64_t *out_keys, uint32_t *out_counts, int nthreads,
                               uint64_t *out_rows_scored) {
    Forest f{plane, left, right, roots, planes, consts, leaf_ids, d, T};
    uint64_t rows = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel reduction(+ : rows)
#endif
    {
        Ctx c;
        c.f = &f;
        c.sc.X = X; c.sc.d = d; c.sc.metric = metric; c.sc.mode = mode;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (uint64_t i = 0; i < b; i++) {
            const float *q = Q + i * d;
            c.sc.q = q;
            c.sc.qq = metric == ZF_COSINE ? dot(q, q, d) : 0.0f;
            c.cand.clear();
            for (uint32_t t = 0; t < T; t++) walk(c, (int32_t)roots[t], (int32_t)k);
            // union over the trees (lsh.rs:550 DashSet): an id scored in two trees carries the same key twice
            std::sort(c.cand.begin(), c.cand.end(), [](const Cand &x, const Cand &y) { return x.id < y.id; });
            c.cand.erase(std::unique(c.cand.begin(), c.cand.end(), [](const Cand &x, const Cand &y) { return x.id == y.id; }),
                         c.cand.end());
            const uint32_t m = (uint32_t)std::min<size_t>(k, c.cand.size());
            std::partial_sort(c.cand.begin(), c.cand.begin() + m, c.cand.end(), cand_less);
            for (uint32_t j = 0; j < m; j++) { out_ids[i * k + j] = c.cand[j].id; out_keys[i * k + j] = c.cand[j].key; }
            out_counts[i] = m;
        }
        rows += c.rows_scored;
    }
    if (out_rows_scored) *out_rows_scored = rows;
}

ZF_EXPORT int zf_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
ZF_EXPORT const char *zf_isa(void) {
#if defined(__AVX512F__)
    return "avx512";
#elif defined(__AVX2__)
    return "avx2";
#else
    return "generic";
#endif
}
