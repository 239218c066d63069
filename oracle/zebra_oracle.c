/*
 * zebra_oracle.c -- CPU restatement of emmyoh/zebra's LSH bucket-scan + distance hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / the reported CPU baseline.  The product path
 * (zebra_amd/, include/) never links, imports or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference (/root/reference, Rust) has no tests, golden vectors or fixtures
 * (SURVEY.md s4), cannot be compiled here (no cargo/rustc), and its arithmetic lives in the
 * un-vendored crates simsimd = "6.2.3" and distances = "1.8.0" (Cargo.toml:27,38).  This file
 * restates the algorithm from the reference's own call sites; simsimd semantics (f32 accumulation,
 * cos() returning the *distance* 1-cos, clip at 0, the two zero-norm special cases) are restated
 * from that crate's published behaviour.  It is cross-checked against numpy/scipy float64 brute
 * force in tests/test_oracle_kat.py.
 *
 * What follows which reference lines:
 *   zo_dot32 / zo_point_is_above   src/database/index/lsh.rs:39-43
 *   make_hyperplane                src/database/index/lsh.rs:174-190, 192-231
 *   build_node (classification)    src/database/index/lsh.rs:233-267
 *   walk (tree_result)             src/database/index/lsh.rs:290-348
 *   zo_search                      src/database/index/lsh.rs:544-565
 *   zo_distance (keys)             src/distance.rs:13-49, 99-114
 *   zo_search_batch                src/database/core.rs:290-313
 *
 * Choices the reference leaves open (it is non-deterministic: unseeded RNG lsh.rs:201, unstable
 * sorts lsh.rs:318,561, Uuid::now_v7 ids lsh.rs:415) and that this restatement FIXES, identically
 * to the HIP path, so that "bit-exact" is well defined:
 *   - ids are dense row indices in insertion order;
 *   - the two sample rows of a hyperplane come from a counter RNG keyed by (seed, tree, node path);
 *   - the hash dot product is a sequential k-ascending f32 fmaf chain starting from +0
 *     (bitwise what a non-split-K f32 MFMA produces);
 *   - distance accumulations use 256 strided f32 accumulators (element e -> accumulator e mod 256,
 *     ascending e, fmaf), combined as ((a0+a1)+(a2+a3)) per group of four and then a 64-way
 *     xor-butterfly 1,2,4,8,16,32 -- the order a wave64 produces with one float4 per lane;
 *   - ties sort by (key, id) ascending.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ZO_EXPORT __attribute__((visibility("default")))

enum { ZO_COSINE = 0, ZO_L2SQ = 1, ZO_L2 = 2, ZO_CHEBYSHEV = 3, ZO_CANBERRA = 4, ZO_BRAY_CURTIS = 5, ZO_MANHATTAN = 6,
       ZO_L3 = 7, ZO_L4 = 8, ZO_HAMMING = 9, ZO_MINKOWSKI = 10, ZO_PNORM = 11 };
enum { ZO_PARITY = 0, ZO_CORRECTED = 1 };
#define ZO_MAX_DEPTH 60 /* guard: the reference recurses forever on an unsplittable node */

/* ---------------------------------------------------------------- counter RNG / synthetic data */

static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* Irwin-Hall(4) of four 16-bit fields: integer arithmetic + one exact int->float + one f32 multiply,
 * so the CPU and the GPU generator agree bit for bit (no libm). mean 0, variance 1. */
static inline int32_t synth_centered(uint64_t seed, uint64_t idx) {
    uint64_t z = splitmix64(seed ^ (idx * 0xD1342543DE82EF95ull));
    uint32_t s = (uint32_t)(z & 0xFFFF) + (uint32_t)((z >> 16) & 0xFFFF) + (uint32_t)((z >> 32) & 0xFFFF) +
                 (uint32_t)(z >> 48);
    return (int32_t)s - 131070;
}
/* kind 2 ("clustered"): 128 consecutive rows share a centre: x = centre + 0.25 * own noise */
#define ZO_CLUSTER_ROWS 128
static inline float synth_value(uint64_t seed, uint64_t idx, int kind) {
    int32_t t = synth_centered(seed, idx);
    if (kind == 1) { /* "SIFT-style": integer-valued in [0,255] -> L2^2 exact in f32 */
        int32_t v = 30 + (t * 35) / 37837;
        if (v < 0) v = 0;
        if (v > 255) v = 255;
        return (float)v;
    }
    return (float)t * (1.0f / 37837.2f);
}

/* kind 3 ("clustered, shuffled"): a row's cluster is drawn by a hash of its id (65,536 clusters): cluster-mates scattered over the table */
static inline float synth_elem(uint64_t seed, uint64_t row, uint32_t col, uint32_t d, int kind) {
    if (kind == 2 || kind == 3) {
        const uint64_t cluster = kind == 2 ? row / ZO_CLUSTER_ROWS : (splitmix64(seed ^ 0x5C0FF1Eull ^ (row * 0x9E3779B97F4A7C15ull)) & 0xFFFFull);
        float centre = (float)synth_centered(seed ^ 0xC1A57E5ull, cluster * d + col) * (1.0f / 37837.2f);
        float own = (float)synth_centered(seed, row * d + col) * (1.0f / 37837.2f);
        return fmaf(0.25f, own, centre);
    }
    return synth_value(seed, row * d + col, kind);
}
ZO_EXPORT void zo_synth_rows(uint64_t seed, uint64_t row0, uint64_t n, uint32_t d, int kind, float *out) {
    for (uint64_t r = 0; r < n; r++)
        for (uint32_t c = 0; c < d; c++) out[r * d + c] = synth_elem(seed, row0 + r, c, d, kind);
}

/* query b = stored row r_b + 0.3 * noise (planted neighbour); r_b from the query stream */
ZO_EXPORT uint64_t zo_synth_query_row(uint64_t seed_q, uint64_t b, uint64_t n_rows) {
    return splitmix64(seed_q ^ (b * 0xA24BAED4963EE407ull)) % n_rows;
}
ZO_EXPORT void zo_synth_queries(uint64_t seed_rows, uint64_t seed_q, uint64_t n_rows, uint64_t b0, uint64_t b,
                                uint32_t d, int kind, float *out) {
    for (uint64_t i = 0; i < b; i++) {
        uint64_t r = zo_synth_query_row(seed_q, b0 + i, n_rows);
        for (uint32_t c = 0; c < d; c++) {
            float x = synth_elem(seed_rows, r, c, d, kind);
            float g = (float)synth_centered(seed_q + 0x51ED270B5EB2A002ull, (b0 + i) * d + c) * (1.0f / 37837.2f);
            out[i * d + c] = (kind == 1) ? x + (float)((int32_t)(g * 4.0f)) : fmaf(kind >= 2 ? 0.1f : 0.3f, g, x);
        }
    }
}

/* ------------------------------------------------------------------------------- hash (lsh.rs:39-43) */

/* simsimd f32 dot: f32 accumulation, returned widened.  Order fixed: sequential fmaf chain. */
ZO_EXPORT float zo_dot32(const float *w, const float *x, uint32_t d) {
    float acc = 0.0f;
    for (uint32_t k = 0; k < d; k++) acc = fmaf(w[k], x[k], acc);
    return acc;
}

/* lsh.rs:40-42: dot (f64 from f32 accumulator) + constant as f64 >= 0.0 ; NaN -> false */
ZO_EXPORT int zo_point_is_above(const float *w, float c, const float *x, uint32_t d) {
    return ((double)zo_dot32(w, x, d) + (double)c) >= 0.0;
}

/* eight rows at once (independent chains -> hides fma latency); same per-row order as zo_dot32 */
static void dot32_rows8(const float *w, const float *const *rows, int n, uint32_t d, float *out) {
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    const float *r[8];
    for (int i = 0; i < 8; i++) r[i] = rows[i < n ? i : 0];
    for (uint32_t k = 0; k < d; k++) {
        float wk = w[k];
        a0 = fmaf(wk, r[0][k], a0);
        a1 = fmaf(wk, r[1][k], a1);
        a2 = fmaf(wk, r[2][k], a2);
        a3 = fmaf(wk, r[3][k], a3);
        a4 = fmaf(wk, r[4][k], a4);
        a5 = fmaf(wk, r[5][k], a5);
        a6 = fmaf(wk, r[6][k], a6);
        a7 = fmaf(wk, r[7][k], a7);
    }
    float a[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
    for (int i = 0; i < n; i++) out[i] = a[i];
}

/* --------------------------------------------------------------------- keys (distance.rs:13-49,99-114) */

typedef struct {
    float ab, a2, b2, l2;
} zo_sums;

static inline float combine256(const float *acc) {
    float s[64], t[64];
    for (int l = 0; l < 64; l++) s[l] = (acc[4 * l] + acc[4 * l + 1]) + (acc[4 * l + 2] + acc[4 * l + 3]);
    for (int m = 1; m < 64; m <<= 1) {
        for (int l = 0; l < 64; l++) t[l] = s[l] + s[l ^ m];
        memcpy(s, t, sizeof s);
    }
    return s[0];
}

static inline float sum_l2sq(const float *a, const float *b, uint32_t d) {
    float acc[256];
    memset(acc, 0, sizeof acc);
    for (uint32_t base = 0; base < d; base += 256) {
        uint32_t m = d - base < 256 ? d - base : 256;
        for (uint32_t i = 0; i < m; i++) {
            float df = a[base + i] - b[base + i];
            acc[i] = fmaf(df, df, acc[i]);
        }
    }
    return combine256(acc);
}
static inline float sum_prod(const float *a, const float *b, uint32_t d) {
    float acc[256];
    memset(acc, 0, sizeof acc);
    for (uint32_t base = 0; base < d; base += 256) {
        uint32_t m = d - base < 256 ? d - base : 256;
        for (uint32_t i = 0; i < m; i++) acc[i] = fmaf(a[base + i], b[base + i], acc[i]);
    }
    return combine256(acc);
}

static inline uint64_t bits64(double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
}

/* simsimd cos(): the cosine DISTANCE, clipped at 0, with the two zero-norm cases */
static inline double cos_distance(float ab, float a2, float b2) {
    if (a2 == 0.0f && b2 == 0.0f) return 0.0;
    if (ab == 0.0f) return 1.0;
    double r = 1.0 - (double)ab / sqrt((double)a2 * (double)b2);
    return r > 0.0 ? r : 0.0;
}

static inline uint64_t key_from_sums(int metric, int mode, float ab, float a2, float b2, float l2) {
    switch (metric) {
    case ZO_COSINE: {
        double c = cos_distance(ab, a2, b2);
        /* distance.rs:23-25: `.map(|c| 1.0 - c)` applied to what is already a distance */
        return bits64(mode == ZO_PARITY ? 1.0 - c : c);
    }
    case ZO_L2SQ: return bits64((double)l2);        /* distance.rs:41-42 */
    default: return bits64(sqrt((double)l2));       /* distance.rs:106-107 */
    }
}


/* ---- the ten metrics of the `distances` crate path (distance.rs:51-98,116-190): keys are f32::to_bits()
 * widened to u64 (distance.rs:59 etc.), Hamming is an integer count (distance.rs:144-158).  Semantics of
 * distances = "1.8.0" restated from its published API: manhattan sum|a-b|, chebyshev max|a-b|,
 * canberra sum |a-b|/(|a|+|b|), bray_curtis sum|a-b| / sum|a+b|, l3/l4 norms, minkowski(p) and minkowski_p(p)
 * with powi.  Sums use the same 256-accumulator order as the simsimd-path metrics (the crate sums
 * sequentially; the difference is within the 1e-5 bar).  Roots other than sqrt are computed by a fixed
 * Newton iteration in f64 so that CPU and GPU agree bit for bit. */
static inline float powi_f32(float a, int b) { /* compiler-rt __powisf2: what Rust's f32::powi lowers to */
    const int recip = b < 0;
    float r = 1.0f;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0f / r : r;
}

static inline double root_p(double s, int p) { /* s^(1/p), 1 <= p <= 64, deterministic (only + * / on f64) */
    if (!(s > 0.0) || s == (double)INFINITY || p == 1) return s; /* 0, NaN, inf pass through */
    if (p == 2) return sqrt(s);
    uint64_t u;
    memcpy(&u, &s, 8);
    int e = (int)((u >> 52) & 0x7FF) - 1023;            /* s = m * 2^e, m in [1,2): s is a widened f32, never subnormal */
    int fl = e >= 0 ? e / p : -((-e + p - 1) / p);      /* floor(e / p) */
    int rem = e - fl * p;                               /* in [0, p) */
    /* root = 2^fl * 2^(rem/p) * m^(1/p) <= 2^fl * (1 + rem/p) * (1 + 1/p): start just above, Newton from above */
    double y = ldexp((1.0 + (double)rem / (double)p) * (1.0 + 1.0 / (double)p), fl);
    for (int it = 0; it < 16; it++) {
        double yp = 1.0;
        for (int i = 0; i < p - 1; i++) yp *= y;        /* y^(p-1) */
        y = ((double)(p - 1) * y + s / yp) / (double)p;
    }
    return y;
}

/* s^(1/p) for p > 64 (Newton's basin shrinks like 1/p): exp(ln(s) / p) from fixed f64 series -- only + * / and
 * int <-> double conversions, so that CPU and GPU agree bit for bit.  s finite, > 0, a widened f32. */
static inline double root_big(double s, double p) {
    uint64_t u;
    memcpy(&u, &s, 8);
    int e = (int)((u >> 52) & 0x7FF) - 1023;
    u = (u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m;
    memcpy(&m, &u, 8);                                  /* m in [1, 2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double z = (m - 1.0) / (m + 1.0), z2 = z * z; /* ln m = 2 atanh z, |z| <= 0.1716 */
    double t = 0.0;
    for (int k = 25; k >= 1; k -= 2) t = t * z2 + 1.0 / (double)k;
    const double LN2 = 0.6931471805599453;
    const double x = ((double)e * LN2 + 2.0 * z * t) / p;   /* |x| <= 104 / 65 */
    const double xs = x / LN2;
    const int n = (int)(xs < 0.0 ? xs - 0.5 : xs + 0.5);
    const double r = x - (double)n * LN2;               /* |r| <= 0.35 */
    double term = 1.0, sum = 1.0;
    for (int k = 1; k <= 20; k++) { term = term * r / (double)k; sum = sum + term; }
    return ldexp(sum, n);
}

/* `sum.powf(1.0 / p as f32)` of distances::vectors::minkowski for ANY i32 p (MinkowskiDistance derives Default:
 * power 0, distance.rs:160-165; negative powers are legal).  IEEE pow: p = 0 -> exponent +inf: NaN -> NaN, s > 1 -> +inf,
 * s == 1 -> 1, s < 1 -> 0; p < 0: pow(0, neg) = +inf, pow(+inf, neg) = 0, else 1 / s^(1/|p|). */
static inline double root_any(double s, int p) {
    if (s != s) return s;
    if (p == 0) return s > 1.0 ? (double)INFINITY : (s == 1.0 ? 1.0 : 0.0);
    const int64_t ap = p < 0 ? -(int64_t)p : (int64_t)p;
    double r;
    if (!(s > 0.0) || s == (double)INFINITY || ap == 1) r = s;
    else r = ap <= 64 ? root_p(s, (int)ap) : root_big(s, (double)ap);
    if (p > 0) return r;
    if (r == 0.0) return (double)INFINITY;
    if (r == (double)INFINITY) return 0.0;
    return 1.0 / r;
}

static inline uint32_t bits32(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
// memcpy / qsort with a count of zero and a null pointer (an empty leaf, an empty index) are undefined behaviour by the letter
// of the standard, and -fsanitize=undefined says so (tests/test_sanitizers.py): nothing is called for nothing
static inline void copy_n(void *dst, const void *src, size_t bytes) { if (bytes) memcpy(dst, src, bytes); }
static inline void sort_n(void *base, size_t n, size_t size, int (*cmp)(const void *, const void *)) { if (n > 1) qsort(base, n, size, cmp); }

typedef struct { float s0, s1; } zo_pairsum;

static inline float combine256_max(const float *acc) {
    float s[64], t[64];
    for (int l = 0; l < 64; l++) s[l] = fmaxf(fmaxf(acc[4 * l], acc[4 * l + 1]), fmaxf(acc[4 * l + 2], acc[4 * l + 3]));
    for (int m = 1; m < 64; m <<= 1) {
        for (int l = 0; l < 64; l++) t[l] = fmaxf(s[l], s[l ^ m]);
        memcpy(s, t, sizeof s);
    }
    return s[0];
}

static zo_pairsum sums_generic(int metric, int power, const float *a, const float *b, uint32_t d) {
    float acc0[256], acc1[256];
    memset(acc0, 0, sizeof acc0);
    memset(acc1, 0, sizeof acc1);
    for (uint32_t e = 0; e < d; e++) {
        uint32_t i = e & 255;
        float x = a[e], y = b[e], ad = fabsf(x - y);
        switch (metric) {
        case ZO_CHEBYSHEV: acc0[i] = fmaxf(acc0[i], ad); break;
        case ZO_CANBERRA: acc0[i] = acc0[i] + ad / (fabsf(x) + fabsf(y)); break;
        case ZO_BRAY_CURTIS: acc0[i] = acc0[i] + ad; acc1[i] = acc1[i] + fabsf(x + y); break;
        case ZO_MANHATTAN: acc0[i] = acc0[i] + ad; break;
        case ZO_L3: acc0[i] = acc0[i] + ad * (ad * ad); break;
        case ZO_L4: { float t = ad * ad; acc0[i] = acc0[i] + t * t; break; }
        case ZO_HAMMING: acc0[i] = acc0[i] + (float)__builtin_popcount((bits32(x) ^ bits32(y)) & 0xFFu); break;
        default: acc0[i] = acc0[i] + powi_f32(ad, power); break; /* MINKOWSKI, PNORM */
        }
    }
    zo_pairsum r;
    r.s0 = metric == ZO_CHEBYSHEV ? combine256_max(acc0) : combine256(acc0);
    r.s1 = metric == ZO_BRAY_CURTIS ? combine256(acc1) : 0.0f;
    return r;
}

static uint64_t key_generic(int metric, int power, zo_pairsum s) {
    float f;
    switch (metric) {
    case ZO_BRAY_CURTIS: f = s.s0 / s.s1; break;
    case ZO_L3: f = (float)root_p((double)s.s0, 3); break;
    case ZO_L4: f = sqrtf(sqrtf(s.s0)); break;
    case ZO_HAMMING: return (uint64_t)s.s0;
    case ZO_MINKOWSKI: f = (float)root_any((double)s.s0, power); break;
    default: f = s.s0; break; /* CHEBYSHEV, CANBERRA, MANHATTAN, PNORM */
    }
    return (uint64_t)bits32(f);
}

/* Metric::distance(a = stored, b = query) -> DistanceUnit (u64 bit pattern); `mode` is the cosine mode for
 * ZO_COSINE and the power for ZO_MINKOWSKI / ZO_PNORM */
ZO_EXPORT uint64_t zo_distance(int metric, int mode, const float *a, const float *b, uint32_t d) {
    if (metric >= ZO_CHEBYSHEV) return key_generic(metric, mode, sums_generic(metric, mode, a, b, d));
    if (metric == ZO_COSINE)
        return key_from_sums(metric, mode, sum_prod(a, b, d), sum_prod(a, a, d), sum_prod(b, b, d), 0.0f);
    return key_from_sums(metric, mode, 0, 0, 0, sum_l2sq(a, b, d));
}

ZO_EXPORT void zo_distance_batch(int metric, int mode, const float *rows, const float *q, uint64_t n, uint32_t d,
                                 uint64_t *out_keys) {
    float b2 = metric == ZO_COSINE ? sum_prod(q, q, d) : 0.0f;
    for (uint64_t i = 0; i < n; i++) {
        const float *a = rows + i * d;
        if (metric >= ZO_CHEBYSHEV) { out_keys[i] = key_generic(metric, mode, sums_generic(metric, mode, a, q, d)); continue; }
        out_keys[i] = metric == ZO_COSINE
                          ? key_from_sums(metric, mode, sum_prod(a, q, d), sum_prod(a, a, d), b2, 0.0f)
                          : key_from_sums(metric, mode, 0, 0, 0, sum_l2sq(a, q, d));
    }
}

/* raw sums, for tests that want to look under the key */
ZO_EXPORT void zo_distance_sums(const float *a, const float *b, uint32_t d, float *out4) {
    out4[0] = sum_prod(a, b, d);
    out4[1] = sum_prod(a, a, d);
    out4[2] = sum_prod(b, b, d);
    out4[3] = sum_l2sq(a, b, d);
}

/* ------------------------------------------------------------------------------------------- forest */

typedef struct zo_forest {
    uint64_t n_rows;
    uint32_t d, M, T;
    uint64_t seed;
    /* nodes: plane >= 0 -> inner (left = below child, right = above child, lsh.rs:260-264);
     *        plane == -1 -> leaf (left = offset into leaf_ids, right = length) */
    int32_t *plane, *left, *right;
    uint8_t *depth;
    uint32_t n_nodes, cap_nodes;
    uint32_t *roots;
    float *planes, *consts;
    uint32_t n_planes, cap_planes;
    uint32_t *leaf_ids;
    uint64_t n_leaf_ids, cap_leaf_ids;
    int borrowed; /* the arrays belong to the caller (zo_forest_borrow_arrays): never reallocated or freed here */
    /* rows removed by zo_forest_remove: LSHIndex::remove deletes the embedding from the KV store (lsh.rs:495) and
     * build_hyperplane samples from the stored embeddings only (lsh.rs:197-201), so a removed row never defines a plane */
    uint8_t *dead;
    uint64_t dead_cap, n_dead;
} zo_forest;

static uint32_t new_node(zo_forest *f) {
    if (f->n_nodes == f->cap_nodes) {
        f->cap_nodes = f->cap_nodes ? f->cap_nodes * 2 : 1024;
        f->plane = realloc(f->plane, f->cap_nodes * sizeof(int32_t));
        f->left = realloc(f->left, f->cap_nodes * sizeof(int32_t));
        f->right = realloc(f->right, f->cap_nodes * sizeof(int32_t));
        f->depth = realloc(f->depth, f->cap_nodes);
    }
    return f->n_nodes++;
}
static uint32_t new_plane(zo_forest *f) {
    if (f->n_planes == f->cap_planes) {
        f->cap_planes = f->cap_planes ? f->cap_planes * 2 : 256;
        f->planes = realloc(f->planes, (size_t)f->cap_planes * f->d * sizeof(float));
        f->consts = realloc(f->consts, f->cap_planes * sizeof(float));
    }
    return f->n_planes++;
}
static uint64_t push_leaf(zo_forest *f, const uint32_t *ids, uint32_t n) {
    if (f->n_leaf_ids + n > f->cap_leaf_ids) {
        while (f->n_leaf_ids + n > f->cap_leaf_ids) f->cap_leaf_ids = f->cap_leaf_ids ? f->cap_leaf_ids * 2 : 4096;
        f->leaf_ids = realloc(f->leaf_ids, f->cap_leaf_ids * sizeof(uint32_t));
    }
    uint64_t off = f->n_leaf_ids;
    copy_n(f->leaf_ids + off, ids, n * sizeof(uint32_t));
    f->n_leaf_ids += n;
    return off;
}

/* two distinct rows, uniform over the WHOLE database (lsh.rs:197-201), keyed by
 * (seed, tree, heap path of the node: root 1, below child 2p, above child 2p+1) */
ZO_EXPORT void zo_sample_pair(uint64_t seed, uint32_t tree, uint64_t path, uint64_t n_rows, uint64_t *i, uint64_t *j) {
    uint64_t h = splitmix64(seed ^ splitmix64(0x7EE5ull + tree) ^ splitmix64(path * 0xC2B2AE3D27D4EB4Full));
    uint64_t h1 = splitmix64(h), h2 = splitmix64(h1);
    if (n_rows < 2) { *i = 0; *j = 0; return; }
    *i = h1 % n_rows;
    *j = h2 % (n_rows - 1);
    if (*j >= *i) (*j)++;
}

/* The two sample rows of a split, drawn from the LIVE rows among the first n_rows (lsh.rs:197-201 iterates the embeddings
 * partition, which remove() has deleted from): the draw (i, j) over the live count is mapped to the i-th / j-th live row.
 * Returns the live count (< 2: lsh.rs:203-220 falls back to default vectors). */
static uint64_t sample_live_pair(const zo_forest *f, uint32_t tree, uint64_t path, uint64_t n_rows, uint64_t *si, uint64_t *sj) {
    uint64_t n_live = n_rows;
    if (f->n_dead)
        for (uint64_t r = 0; r < n_rows && r < f->dead_cap; r++) n_live -= f->dead[r];
    zo_sample_pair(f->seed, tree, path, n_live, si, sj);
    if (f->n_dead && n_live) {
        uint64_t want[2] = {*si, *sj}, got[2] = {0, 0}, k = 0;
        for (uint64_t r = 0; r < n_rows; r++) {
            if (r < f->dead_cap && f->dead[r]) continue;
            if (k == want[0]) got[0] = r;
            if (k == want[1]) got[1] = r;
            k++;
        }
        *si = got[0]; *sj = got[1];
    }
    return n_live;
}

/* lsh.rs:222-225: w = b - a ; p = (a + b) / 2 ; c = -(dot(w, p)) as f32 */
ZO_EXPORT void zo_make_hyperplane(const float *a, const float *b, uint32_t d, float *w, float *c) {
    float acc = 0.0f;
    for (uint32_t k = 0; k < d; k++) {
        w[k] = b[k] - a[k];
        float p = (a[k] + b[k]) / 2.0f;
        acc = fmaf(w[k], p, acc);
    }
    *c = -acc;
}

static int32_t build_node(zo_forest *f, const float *X, uint32_t tree, uint64_t path, int depth, uint32_t *ids,
                          uint32_t n, uint32_t *scratch);

/* (re)build the subtree of an existing node index `me` over ids (build_a_tree, lsh.rs:250-267) */
static int32_t build_node_into(zo_forest *f, uint32_t me, const float *X, uint32_t tree, uint64_t path, int depth,
                               uint32_t *ids, uint32_t n, uint32_t *scratch) {
    f->depth[me] = (uint8_t)depth;
    if (n < f->M || depth >= ZO_MAX_DEPTH) { /* lsh.rs:251-252 */
        f->plane[me] = -1;
        f->left[me] = (int32_t)push_leaf(f, ids, n);
        f->right[me] = (int32_t)n;
        return (int32_t)me;
    }
    uint32_t d = f->d;
    uint64_t si, sj;
    const uint64_t n_live = sample_live_pair(f, tree, path, f->n_rows, &si, &sj);
    uint32_t p = new_plane(f);
    float *w = f->planes + (size_t)p * d;
    float *zero = NULL;
    const float *a = X + si * d, *b = X + sj * d;
    if (n_live < 2) { /* lsh.rs:203-220: missing samples decode to the all-zero default */
        zero = calloc(d, sizeof(float));
        if (n_live == 0) a = zero;
        b = zero;
    }
    zo_make_hyperplane(a, b, d, w, &f->consts[p]);
    free(zero);
    float c = f->consts[p];
    /* lsh.rs:236-241: classify; keep ascending id order on both sides (stable) */
    uint32_t na = 0, nb = 0;
    uint32_t *above = scratch, *below = scratch + n;
    for (uint32_t i = 0; i < n; i += 8) {
        int m = n - i < 8 ? (int)(n - i) : 8;
        const float *rows[8];
        float dots[8];
        for (int r = 0; r < m; r++) rows[r] = X + (size_t)ids[i + r] * d;
        dot32_rows8(f->planes + (size_t)p * d, rows, m, d, dots);
        for (int r = 0; r < m; r++) {
            if (((double)dots[r] + (double)c) >= 0.0) above[na++] = ids[i + r];
            else below[nb++] = ids[i + r];
        }
    }
    copy_n(ids, below, nb * sizeof(uint32_t));
    copy_n(ids + nb, above, na * sizeof(uint32_t));
    f->plane[me] = (int32_t)p;
    /* lsh.rs:257-264: above first, then below; left = below, right = above */
    int32_t r = build_node(f, X, tree, 2 * path + 1, depth + 1, ids + nb, na, scratch);
    int32_t l = build_node(f, X, tree, 2 * path, depth + 1, ids, nb, scratch);
    f->left[me] = l;
    f->right[me] = r;
    return (int32_t)me;
}

static int32_t build_node(zo_forest *f, const float *X, uint32_t tree, uint64_t path, int depth, uint32_t *ids,
                          uint32_t n, uint32_t *scratch) {
    return build_node_into(f, new_node(f), X, tree, path, depth, ids, n, scratch);
}

/* lsh.rs:350-382 insert: descend by point_is_above; a leaf takes the id while len + 1 <= M, else the node is
 * rebuilt by build_a_tree over (leaf ids + id), sampling its hyperplanes from the database as it is NOW */
static void insert_one(zo_forest *f, const float *X, uint32_t tree, uint32_t id) {
    uint32_t node = f->roots[tree];
    uint64_t path = 1;
    int depth = 0;
    const float *x = X + (size_t)id * f->d;
    while (f->plane[node] >= 0) {
        int32_t p = f->plane[node];
        int above = zo_point_is_above(f->planes + (size_t)p * f->d, f->consts[p], x, f->d);
        node = (uint32_t)(above ? f->right[node] : f->left[node]);
        path = 2 * path + (above ? 1 : 0);
        depth++;
    }
    uint32_t off = (uint32_t)f->left[node], len = (uint32_t)f->right[node];
    uint32_t *ids = malloc((len + 1) * sizeof(uint32_t));
    copy_n(ids, f->leaf_ids + off, len * sizeof(uint32_t));
    ids[len] = id;
    if (len + 1 > f->M) { /* lsh.rs:368-377 */
        uint32_t *scratch = malloc((size_t)(len + 1) * 2 * sizeof(uint32_t));
        build_node_into(f, node, X, tree, path, depth, ids, len + 1, scratch);
        free(scratch);
    } else { /* lsh.rs:369: push */
        f->left[node] = (int32_t)push_leaf(f, ids, len + 1);
        f->right[node] = (int32_t)(len + 1);
    }
    free(ids);
}

/* lsh.rs:445-462, as ONE admissible sequential execution of its racy par_iter: rows n_prev .. n_prev+n_new-1
 * of X are added one after another, each into every tree, the database growing by one row each time */
ZO_EXPORT void zo_forest_insert(zo_forest *f, const float *X, uint64_t n_prev, uint64_t n_new) {
    for (uint64_t r = 0; r < n_new; r++) {
        f->n_rows = n_prev + r + 1;
        for (uint32_t t = 0; t < f->T; t++) insert_one(f, X, t, (uint32_t)(n_prev + r));
    }
}

/* LSHIndex::remove (lsh.rs:473-503), as what it intends: the reference only edits trees whose ROOT is a leaf, so
 * ids stay in inner trees while their embedding is deleted; here the id leaves every tree.  The row's
 * leaf is found by descending with the row's own vector (the path insert/build took); a forest that was injected
 * with other contents falls back to scanning the tree's leaves.  Returns 1 if the id was present in any tree. */
static int remove_from_leaf(zo_forest *f, uint32_t node, uint32_t id) {
    uint32_t off = (uint32_t)f->left[node], len = (uint32_t)f->right[node];
    for (uint32_t i = 0; i < len; i++)
        if (f->leaf_ids[off + i] == id) {
            uint32_t *ids = malloc((len ? len : 1) * sizeof(uint32_t));
            copy_n(ids, f->leaf_ids + off, i * sizeof(uint32_t));
            copy_n(ids + i, f->leaf_ids + off + i + 1, (len - i - 1) * sizeof(uint32_t));
            f->left[node] = (int32_t)push_leaf(f, ids, len - 1);
            f->right[node] = (int32_t)(len - 1);
            free(ids);
            return 1;
        }
    return 0;
}
ZO_EXPORT uint64_t zo_forest_remove(zo_forest *f, const float *X, const uint64_t *ids, uint64_t n, uint8_t *out_found) {
    uint64_t removed = 0;
    for (uint64_t r = 0; r < n; r++) {
        int found = 0;
        if (ids[r] < f->n_rows) {
            uint32_t id = (uint32_t)ids[r];
            const float *x = X + (size_t)id * f->d;
            for (uint32_t t = 0; t < f->T; t++) {
                uint32_t node = f->roots[t];
                while (f->plane[node] >= 0) {
                    int32_t p = f->plane[node];
                    node = (uint32_t)(zo_point_is_above(f->planes + (size_t)p * f->d, f->consts[p], x, f->d) ? f->right[node] : f->left[node]);
                }
                int hit = remove_from_leaf(f, node, id);
                if (!hit) { /* not where it hashes to (injected forest): scan this tree's leaves */
                    uint32_t *stack = malloc((f->n_nodes + 1) * sizeof(uint32_t));
                    uint32_t sp = 0;
                    stack[sp++] = f->roots[t];
                    while (sp && !hit) {
                        uint32_t m = stack[--sp];
                        if (f->plane[m] < 0) hit = remove_from_leaf(f, m, id);
                        else { stack[sp++] = (uint32_t)f->left[m]; stack[sp++] = (uint32_t)f->right[m]; }
                    }
                    free(stack);
                }
                found |= hit;
            }
        }
        if (found) { /* the embedding is gone from the store: it can no longer be sampled */
            if (ids[r] >= f->dead_cap) {
                uint64_t cap = f->n_rows > ids[r] + 1 ? f->n_rows : ids[r] + 1;
                f->dead = realloc(f->dead, cap);
                memset(f->dead + f->dead_cap, 0, cap - f->dead_cap);
                f->dead_cap = cap;
            }
            if (!f->dead[ids[r]]) { f->dead[ids[r]] = 1; f->n_dead++; }
        }
        if (out_found) out_found[r] = (uint8_t)found;
        removed += found;
    }
    return removed;
}

/* LSHIndex::deduplicate (lsh.rs:270-288): rows whose f32 bit patterns equal an EARLIER live row's are removed;
 * `alive` (may be NULL = all alive) marks rows still in the index; out_dup[i] = 1 for the rows to remove */
typedef struct { const uint32_t *bits; uint32_t d; } zo_rowcmp_ctx;
static zo_rowcmp_ctx g_rc;
static int rowcmp(const void *a, const void *b) {
    uint32_t i = *(const uint32_t *)a, j = *(const uint32_t *)b;
    int c = memcmp(g_rc.bits + (size_t)i * g_rc.d, g_rc.bits + (size_t)j * g_rc.d, (size_t)g_rc.d * 4);
    if (c) return c;
    return i < j ? -1 : (i > j);
}
ZO_EXPORT uint64_t zo_find_duplicates(const float *X, uint64_t n, uint32_t d, const uint8_t *alive, uint8_t *out_dup) {
    uint32_t *order = malloc((n ? n : 1) * sizeof(uint32_t));
    uint64_t m = 0, dups = 0;
    for (uint64_t i = 0; i < n; i++) { out_dup[i] = 0; if (!alive || alive[i]) order[m++] = (uint32_t)i; }
    g_rc.bits = (const uint32_t *)X; g_rc.d = d;
    sort_n(order, m, sizeof(uint32_t), rowcmp);
    for (uint64_t i = 1; i < m; i++)
        if (memcmp(X + (size_t)order[i] * d, X + (size_t)order[i - 1] * d, (size_t)d * 4) == 0) { out_dup[order[i]] = 1; dups++; }
    free(order);
    return dups;
}

/* lsh.rs:411-429 build_index: T independent trees over all ids */
ZO_EXPORT zo_forest *zo_forest_build(const float *X, uint64_t n_rows, uint32_t d, uint32_t M, uint32_t T,
                                     uint64_t seed) {
    zo_forest *f = calloc(1, sizeof *f);
    f->n_rows = n_rows; f->d = d; f->M = M; f->T = T; f->seed = seed;
    f->roots = calloc(T ? T : 1, sizeof(uint32_t));
    uint32_t *ids = malloc((n_rows ? n_rows : 1) * sizeof(uint32_t));
    uint32_t *scratch = malloc((n_rows ? n_rows : 1) * 2 * sizeof(uint32_t));
    for (uint32_t t = 0; t < T; t++) {
        for (uint64_t i = 0; i < n_rows; i++) ids[i] = (uint32_t)i;
        f->roots[t] = (uint32_t)build_node(f, X, t, 1, 0, ids, (uint32_t)n_rows, scratch);
    }
    free(ids); free(scratch);
    return f;
}

/* hand-built forests for known-answer tests and for forests exported by the HIP build */
ZO_EXPORT zo_forest *zo_forest_from_arrays(uint64_t n_rows, uint32_t d, uint32_t M, uint32_t T, uint32_t n_nodes,
                                           const int32_t *plane, const int32_t *left, const int32_t *right,
                                           const uint32_t *roots, uint32_t n_planes, const float *planes,
                                           const float *consts, uint64_t n_leaf_ids, const uint32_t *leaf_ids) {
    zo_forest *f = calloc(1, sizeof *f);
    f->n_rows = n_rows; f->d = d; f->M = M; f->T = T;
    f->n_nodes = f->cap_nodes = n_nodes;
    f->plane = malloc((n_nodes + 1) * sizeof(int32_t)); copy_n(f->plane, plane, n_nodes * sizeof(int32_t));
    f->left = malloc((n_nodes + 1) * sizeof(int32_t)); copy_n(f->left, left, n_nodes * sizeof(int32_t));
    f->right = malloc((n_nodes + 1) * sizeof(int32_t)); copy_n(f->right, right, n_nodes * sizeof(int32_t));
    f->depth = calloc(n_nodes + 1, 1);
    f->roots = malloc((T + 1) * sizeof(uint32_t)); copy_n(f->roots, roots, T * sizeof(uint32_t));
    f->n_planes = f->cap_planes = n_planes;
    f->planes = malloc(((size_t)n_planes * d + 1) * sizeof(float)); copy_n(f->planes, planes, (size_t)n_planes * d * sizeof(float));
    f->consts = malloc((n_planes + 1) * sizeof(float)); copy_n(f->consts, consts, n_planes * sizeof(float));
    f->n_leaf_ids = f->cap_leaf_ids = n_leaf_ids;
    f->leaf_ids = malloc((n_leaf_ids + 1) * sizeof(uint32_t)); copy_n(f->leaf_ids, leaf_ids, n_leaf_ids * sizeof(uint32_t));
    return f;
}

/* the same without copying: the forest of a full-size shard (GBs of leaf ids) exported by the HIP build stays in the
 * caller's arrays, which must outlive the forest.  Read-only use (search, checks): insert / remove would realloc. */
ZO_EXPORT zo_forest *zo_forest_borrow_arrays(uint64_t n_rows, uint32_t d, uint32_t M, uint32_t T, uint32_t n_nodes,
                                             const int32_t *plane, const int32_t *left, const int32_t *right,
                                             const uint32_t *roots, uint32_t n_planes, const float *planes,
                                             const float *consts, uint64_t n_leaf_ids, const uint32_t *leaf_ids) {
    zo_forest *f = calloc(1, sizeof *f);
    f->n_rows = n_rows; f->d = d; f->M = M; f->T = T; f->borrowed = 1;
    f->n_nodes = f->cap_nodes = n_nodes;
    f->plane = (int32_t *)plane; f->left = (int32_t *)left; f->right = (int32_t *)right;
    f->roots = (uint32_t *)roots;
    f->n_planes = f->cap_planes = n_planes;
    f->planes = (float *)planes; f->consts = (float *)consts;
    f->n_leaf_ids = f->cap_leaf_ids = n_leaf_ids;
    f->leaf_ids = (uint32_t *)leaf_ids;
    return f;
}

ZO_EXPORT void zo_forest_free(zo_forest *f) {
    if (!f) return;
    free(f->dead);
    if (f->borrowed) { free(f->depth); free(f); return; }
    free(f->plane); free(f->left); free(f->right); free(f->depth); free(f->roots);
    free(f->planes); free(f->consts); free(f->leaf_ids); free(f);
}

ZO_EXPORT void zo_forest_sizes(const zo_forest *f, uint32_t *n_nodes, uint32_t *n_planes, uint64_t *n_leaf_ids) {
    *n_nodes = f->n_nodes; *n_planes = f->n_planes; *n_leaf_ids = f->n_leaf_ids;
}
ZO_EXPORT void zo_forest_export(const zo_forest *f, int32_t *plane, int32_t *left, int32_t *right, uint32_t *roots,
                                float *planes, float *consts, uint32_t *leaf_ids) {
    copy_n(plane, f->plane, f->n_nodes * sizeof(int32_t));
    copy_n(left, f->left, f->n_nodes * sizeof(int32_t));
    copy_n(right, f->right, f->n_nodes * sizeof(int32_t));
    copy_n(roots, f->roots, f->T * sizeof(uint32_t));
    copy_n(planes, f->planes, (size_t)f->n_planes * f->d * sizeof(float));
    copy_n(consts, f->consts, f->n_planes * sizeof(float));
    copy_n(leaf_ids, f->leaf_ids, f->n_leaf_ids * sizeof(uint32_t));
}

/* --------------------------------------------------------------------------- walk (lsh.rs:290-348) */

typedef struct {
    uint32_t id;
    uint64_t key;
} zo_pair;

static int pair_cmp(const void *a, const void *b) {
    const zo_pair *x = a, *y = b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1; /* unsigned order of the bit pattern */
    return x->id < y->id ? -1 : (x->id > y->id);
}

typedef struct {
    const zo_forest *f;
    const float *X, *q;
    int metric, mode;
    float qq; /* sum_prod(q,q): the query-side norm, same value for every stored row */
    /* stored rows that are not in memory: row id -> counter generator (seed, first_row + id, kind), regenerated into
     * `rowbuf` on demand -- what lets the oracle check a 10M..125M-row shard it could never hold (X == NULL) */
    uint64_t synth_seed, synth_row0; int synth_kind; float *rowbuf;
    uint32_t *stamp, epoch; /* candidate set (DashSet, lsh.rs:550); NULL: cand is a list, de-duplicated by sorting */
    uint32_t *cand; uint64_t n_cand, cap_cand;
    zo_pair *buf; uint64_t cap_buf; /* leaf scoring / rerank buffer, grown on demand */
    /* optional visit trace: (leaf offset, leaf length, n taken) triples */
    uint64_t *visits; uint64_t n_visits, cap_visits;
    uint64_t rows_scored, planes_evaluated, leaves_visited;
} zo_ctx;

static inline const float *ctx_row(zo_ctx *c, uint32_t id) {
    if (c->X) return c->X + (size_t)id * c->f->d;
    for (uint32_t col = 0; col < c->f->d; col++) c->rowbuf[col] = synth_elem(c->synth_seed, c->synth_row0 + id, col, c->f->d, c->synth_kind);
    return c->rowbuf;
}
static inline void ctx_reserve_buf(zo_ctx *c, uint64_t n) {
    if (n <= c->cap_buf) return;
    c->cap_buf = n + n / 2 + 64;
    c->buf = realloc(c->buf, c->cap_buf * sizeof(zo_pair));
}
static inline uint64_t ctx_key(zo_ctx *c, uint32_t id) {
    const float *a = ctx_row(c, id);
    if (c->metric >= ZO_CHEBYSHEV) return key_generic(c->metric, c->mode, sums_generic(c->metric, c->mode, a, c->q, c->f->d));
    if (c->metric == ZO_COSINE)
        return key_from_sums(c->metric, c->mode, sum_prod(a, c->q, c->f->d), sum_prod(a, a, c->f->d), c->qq, 0.0f);
    return key_from_sums(c->metric, c->mode, 0, 0, 0, sum_l2sq(a, c->q, c->f->d));
}
static inline void cand_insert(zo_ctx *c, uint32_t id) {
    if (c->stamp) {
        if (c->stamp[id] != c->epoch) { c->stamp[id] = c->epoch; c->cand[c->n_cand++] = id; }
        return;
    }
    if (c->n_cand == c->cap_cand) {
        c->cap_cand = c->cap_cand ? c->cap_cand * 2 : 1024;
        c->cand = realloc(c->cand, c->cap_cand * sizeof(uint32_t));
    }
    c->cand[c->n_cand++] = id;
}
static int u32_cmp(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : (x > y);
}

static int32_t walk(zo_ctx *c, int32_t node, int32_t n) {
    const zo_forest *f = c->f;
    if (f->plane[node] < 0) {
        uint32_t off = (uint32_t)f->left[node], len = (uint32_t)f->right[node];
        const uint32_t *ids = f->leaf_ids + off;
        c->leaves_visited++;
        int32_t ret;
        if ((int64_t)len < (int64_t)n) { /* lsh.rs:300-306: fewer than n -> all of them, unscored */
            for (uint32_t i = 0; i < len; i++) cand_insert(c, ids[i]);
            ret = (int32_t)len;
        } else { /* lsh.rs:308-329: score all, sort ascending by key, take n */
            ctx_reserve_buf(c, len);
            for (uint32_t i = 0; i < len; i++) { c->buf[i].id = ids[i]; c->buf[i].key = ctx_key(c, ids[i]); }
            c->rows_scored += len;
            sort_n(c->buf, len, sizeof(zo_pair), pair_cmp);
            for (int32_t i = 0; i < n; i++) cand_insert(c, c->buf[i].id);
            ret = n;
        }
        if (c->visits && c->n_visits < c->cap_visits) {
            c->visits[3 * c->n_visits] = off; c->visits[3 * c->n_visits + 1] = len;
            c->visits[3 * c->n_visits + 2] = (uint64_t)(ret < 0 ? 0 : ret);
            c->n_visits++;
        }
        return ret;
    }
    int32_t p = f->plane[node];
    c->planes_evaluated++;
    int above = zo_point_is_above(f->planes + (size_t)p * f->d, f->consts[p], c->q, f->d);
    int32_t main_n = above ? f->right[node] : f->left[node];   /* lsh.rs:335-338 */
    int32_t backup = above ? f->left[node] : f->right[node];
    int32_t k = walk(c, main_n, n);
    if (k < n) return walk(c, backup, n - k); /* lsh.rs:341-343: the backup's count ALONE */
    return k;
}

typedef struct {
    uint64_t rows_scored, planes_evaluated, leaves_visited, candidates;
} zo_stats;

static void ctx_init(zo_ctx *c, const zo_forest *f, const float *X, int metric, int mode) {
    memset(c, 0, sizeof *c);
    c->f = f; c->X = X; c->metric = metric; c->mode = mode;
    if (f->n_rows > (1u << 21)) return; /* large stored sets: the candidate set as a list (no n_rows-sized arrays per thread) */
    c->stamp = calloc(f->n_rows ? f->n_rows : 1, sizeof(uint32_t));
    c->cap_cand = f->n_rows ? f->n_rows : 1;
    c->cand = malloc(c->cap_cand * sizeof(uint32_t));
}
/* rows from the counter generator instead of memory; candidate set kept as a list (no n_rows-sized arrays) */
static void ctx_init_synth(zo_ctx *c, const zo_forest *f, uint64_t seed, uint64_t row0, int kind, int metric, int mode) {
    memset(c, 0, sizeof *c);
    c->f = f; c->metric = metric; c->mode = mode;
    c->synth_seed = seed; c->synth_row0 = row0; c->synth_kind = kind;
    c->rowbuf = malloc((f->d ? f->d : 1) * sizeof(float));
}
static void ctx_free(zo_ctx *c) { free(c->stamp); free(c->cand); free(c->buf); free(c->rowbuf); }

/* lsh.rs:544-565 search: every tree with n = top_k; union; RE-score every candidate; sort; take k */
static uint32_t search_one(zo_ctx *c, const float *q, uint32_t k, uint64_t *out_ids, uint64_t *out_keys) {
    const zo_forest *f = c->f;
    c->q = q; c->epoch++; c->n_cand = 0;
    c->qq = c->metric == ZO_COSINE ? sum_prod(q, q, f->d) : 0.0f;
    if (f->n_rows == 0) return 0; /* core.rs:295-297 */
    for (uint32_t t = 0; t < f->T; t++) walk(c, (int32_t)f->roots[t], (int32_t)k);
    if (!c->stamp && c->n_cand > 1) { /* the union of the trees' candidates as a set (lsh.rs:550 DashSet) */
        sort_n(c->cand, c->n_cand, sizeof(uint32_t), u32_cmp);
        uint64_t m = 1;
        for (uint64_t i = 1; i < c->n_cand; i++) if (c->cand[i] != c->cand[m - 1]) c->cand[m++] = c->cand[i];
        c->n_cand = m;
    }
    ctx_reserve_buf(c, c->n_cand);
    zo_pair *r = c->buf;
    for (uint64_t i = 0; i < c->n_cand; i++) { r[i].id = c->cand[i]; r[i].key = ctx_key(c, c->cand[i]); }
    c->rows_scored += c->n_cand;
    sort_n(r, c->n_cand, sizeof(zo_pair), pair_cmp);
    uint32_t m = c->n_cand < k ? (uint32_t)c->n_cand : k;
    for (uint32_t i = 0; i < m; i++) { out_ids[i] = r[i].id; out_keys[i] = r[i].key; }
    return m;
}

ZO_EXPORT uint32_t zo_search(const zo_forest *f, const float *X, const float *q, uint32_t k, int metric, int mode,
                             uint64_t *out_ids, uint64_t *out_keys, zo_stats *st) {
    zo_ctx c; ctx_init(&c, f, X, metric, mode);
    uint32_t m = search_one(&c, q, k, out_ids, out_keys);
    if (st) { st->rows_scored = c.rows_scored; st->planes_evaluated = c.planes_evaluated;
              st->leaves_visited = c.leaves_visited; st->candidates = c.n_cand; }
    ctx_free(&c);
    return m;
}

/* core.rs:290-313 query_vectors: queries are independent; rayon par_iter -> OpenMP threads here.
 * out arrays are b*k (unused tail entries of a row are left untouched), counts[b]. */
ZO_EXPORT void zo_search_batch(const zo_forest *f, const float *X, const float *Q, uint64_t b, uint32_t k, int metric,
                               int mode, uint64_t *out_ids, uint64_t *out_keys, uint32_t *out_counts, int nthreads,
                               zo_stats *st) {
    uint64_t rs = 0, pe = 0, lv = 0, cd = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel reduction(+ : rs, pe, lv, cd)
#endif
    {
        zo_ctx c; ctx_init(&c, f, X, metric, mode);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (uint64_t i = 0; i < b; i++) {
            out_counts[i] = search_one(&c, Q + i * f->d, k, out_ids + i * k, out_keys + i * k);
            cd += c.n_cand;
        }
        rs += c.rows_scored; pe += c.planes_evaluated; lv += c.leaves_visited;
        ctx_free(&c);
    }
    if (st) { st->rows_scored = rs; st->planes_evaluated = pe; st->leaves_visited = lv; st->candidates = cd; }
}

/* The same search with the stored rows REGENERATED from the counter generator (row id -> zo_synth_rows(seed,
 * first_row + id, kind)) instead of read from memory: the full-size configurations (10M x 768 = 31 GB, a 125M x 128
 * shard = 64 GB) are then checked exactly -- ids, keys, counts -- against a forest exported by the HIP build, with
 * no host copy of the rows.  Output ids are LOCAL row ids (add the shard's id_base). */
ZO_EXPORT void zo_search_batch_synth(const zo_forest *f, uint64_t seed_rows, uint64_t first_row, int kind, const float *Q,
                                     uint64_t b, uint32_t k, int metric, int mode, uint64_t *out_ids, uint64_t *out_keys,
                                     uint32_t *out_counts, int nthreads, zo_stats *st) {
    uint64_t rs = 0, pe = 0, lv = 0, cd = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel reduction(+ : rs, pe, lv, cd)
#endif
    {
        zo_ctx c; ctx_init_synth(&c, f, seed_rows, first_row, kind, metric, mode);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (uint64_t i = 0; i < b; i++) {
            out_counts[i] = search_one(&c, Q + i * f->d, k, out_ids + i * k, out_keys + i * k);
            cd += c.n_cand;
        }
        rs += c.rows_scored; pe += c.planes_evaluated; lv += c.leaves_visited;
        ctx_free(&c);
    }
    if (st) { st->rows_scored = rs; st->planes_evaluated = pe; st->leaves_visited = lv; st->candidates = cd; }
}

/* Structural check of a forest built elsewhere (the HIP build at full size, where zo_forest_build would take hours)
 * against the build rules, rows from the counter generator.  Returns 0 when everything holds, else a code:
 *   1  a tree's leaves are not a partition of the live rows [0, n_rows)          (build_index, lsh.rs:411-429)
 *   2  a leaf holds >= M rows above the depth guard, or an inner node < M        (build_a_tree, lsh.rs:251-252)
 *   3  a sampled row does not sit in the leaf it hashes to                        (lsh.rs:233-247 + 39-43)
 *   4  a plane on a sampled row's path is not make_hyperplane of the node's sample pair (lsh.rs:197-231)
 * n_sample rows (evenly spread) are descended through every tree; every plane on their paths is re-derived. */
ZO_EXPORT int zo_check_forest_synth(const zo_forest *f, uint64_t index_seed, uint64_t seed_rows, uint64_t first_row,
                                    int kind, uint64_t n_sample, uint64_t *out_planes_checked) {
    const uint32_t d = f->d;
    int bad = 0;
    uint64_t planes_checked = 0;
    /* 1 + 2: per tree, the leaves partition [0, n_rows); subtree sizes obey the M rule (trees in parallel) */
    int bad1 = 0, bad2 = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) reduction(| : bad1, bad2)
#endif
    for (uint32_t t = 0; t < f->T; t++) {
        uint8_t *seen = calloc(f->n_rows ? f->n_rows : 1, 1);
        /* nodes of this tree in pre-order, with their depth; sizes accumulate in a reverse pass over that order */
        uint32_t cap = 1024, n_order = 0, sp = 0, cap_st = 256;
        uint32_t *order = malloc(cap * sizeof(uint32_t)), *stack = malloc(cap_st * sizeof(uint32_t));
        uint8_t *dep = malloc(cap), *dstack = malloc(cap_st);
        uint64_t total = 0;
        stack[sp] = f->roots[t]; dstack[sp++] = 0;
        while (sp) {
            const uint32_t n = stack[--sp];
            const uint8_t dn = dstack[sp];
            if (n_order == cap) { cap *= 2; order = realloc(order, cap * sizeof(uint32_t)); dep = realloc(dep, cap); }
            order[n_order] = n; dep[n_order++] = dn;
            if (f->plane[n] < 0) {
                const uint32_t off = (uint32_t)f->left[n], len = (uint32_t)f->right[n];
                for (uint32_t i = 0; i < len; i++) {
                    const uint32_t id = f->leaf_ids[(size_t)off + i];
                    if (id >= f->n_rows || seen[id]) { bad1 = 1; break; }
                    seen[id] = 1;
                }
                total += len;
                if (len >= f->M && dn < ZO_MAX_DEPTH) bad2 = 1;
            } else {
                if (sp + 2 > cap_st) { cap_st *= 2; stack = realloc(stack, cap_st * sizeof(uint32_t)); dstack = realloc(dstack, cap_st); }
                stack[sp] = (uint32_t)f->left[n]; dstack[sp++] = (uint8_t)(dn + 1);
                stack[sp] = (uint32_t)f->right[n]; dstack[sp++] = (uint8_t)(dn + 1);
            }
            if (bad1 || n_order > f->n_nodes) { bad1 = 1; break; }
        }
        if (total != f->n_rows) bad1 = 1;
        if (!bad1) { /* subtree sizes: in pre-order every subtree is a contiguous run, children after the parent */
            /* reverse pass with an explicit stack of completed subtree sizes */
            uint64_t *done = malloc(((size_t)n_order + 1) * sizeof(uint64_t));
            uint32_t nd = 0;
            for (uint32_t i = n_order; i-- > 0;) {
                const uint32_t n = order[i];
                if (f->plane[n] < 0) done[nd++] = (uint32_t)f->right[n];
                else { /* its two children's subtrees are the last two completed ones */
                    const uint64_t sz = done[nd - 1] + done[nd - 2];
                    nd -= 2;
                    if (sz < f->M) bad2 = 1;
                    done[nd++] = sz;
                }
            }
            free(done);
        }
        free(seen); free(order); free(stack); free(dep); free(dstack);
    }
    bad = bad1 ? 1 : (bad2 ? 2 : 0);
    if (bad) { if (out_planes_checked) *out_planes_checked = 0; return bad; }
    /* 3 + 4: sampled rows */
    int bad3 = 0, bad4 = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : planes_checked) reduction(| : bad3, bad4)
#endif
    for (uint64_t s = 0; s < n_sample; s++) {
        float *x = malloc(d * sizeof(float)), *a = malloc(d * sizeof(float)), *bb = malloc(d * sizeof(float)), *w = malloc(d * sizeof(float));
        const uint64_t id = n_sample > 1 ? (uint64_t)((double)s * (double)(f->n_rows - 1) / (double)(n_sample - 1)) : 0;
        for (uint32_t c = 0; c < d; c++) x[c] = synth_elem(seed_rows, first_row + id, c, d, kind);
        for (uint32_t t = 0; t < f->T; t++) {
            uint32_t node = f->roots[t];
            uint64_t path = 1;
            while (f->plane[node] >= 0) {
                const int32_t p = f->plane[node];
                uint64_t si, sj;
                float cc;
                zo_sample_pair(index_seed, t, path, f->n_rows, &si, &sj);
                for (uint32_t c = 0; c < d; c++) { a[c] = synth_elem(seed_rows, first_row + si, c, d, kind); bb[c] = synth_elem(seed_rows, first_row + sj, c, d, kind); }
                zo_make_hyperplane(a, bb, d, w, &cc);
                if (memcmp(w, f->planes + (size_t)p * d, d * sizeof(float)) != 0 || memcmp(&cc, &f->consts[p], 4) != 0) bad4 = 1;
                planes_checked++;
                const int above = zo_point_is_above(f->planes + (size_t)p * d, f->consts[p], x, d);
                node = (uint32_t)(above ? f->right[node] : f->left[node]);
                path = 2 * path + (above ? 1 : 0);
            }
            const uint32_t off = (uint32_t)f->left[node], len = (uint32_t)f->right[node];
            int found = 0;
            for (uint32_t i = 0; i < len; i++) if (f->leaf_ids[(size_t)off + i] == id) { found = 1; break; }
            if (!found) bad3 = 1;
        }
        free(x); free(a); free(bb); free(w);
    }
    if (out_planes_checked) *out_planes_checked = planes_checked;
    return bad4 ? 4 : (bad3 ? 3 : 0);
}

/* one tree_result call, exposing the return value, the candidate ids (insertion order) and the
 * visit trace -- for the walk-quirk known-answer tests */
ZO_EXPORT int32_t zo_tree_result(const zo_forest *f, const float *X, uint32_t tree, const float *q, int32_t n,
                                 int metric, int mode, uint32_t *cand_out, uint64_t *n_cand_out, uint64_t *visits_out,
                                 uint64_t cap_visits, uint64_t *n_visits_out) {
    zo_ctx c; ctx_init(&c, f, X, metric, mode);
    c.q = q; c.epoch = 1; c.qq = metric == ZO_COSINE ? sum_prod(q, q, f->d) : 0.0f;
    c.visits = visits_out; c.cap_visits = cap_visits;
    int32_t r = walk(&c, (int32_t)f->roots[tree], n);
    if (cand_out) copy_n(cand_out, c.cand, c.n_cand * sizeof(uint32_t));
    if (n_cand_out) *n_cand_out = c.n_cand;
    if (n_visits_out) *n_visits_out = c.n_visits;
    ctx_free(&c);
    return r;
}

/* sign of every plane of the forest for one query (the dense hash the HIP path computes with MFMA) */
ZO_EXPORT void zo_hash_signs(const zo_forest *f, const float *q, uint8_t *out_signs, float *out_dots) {
    for (uint32_t p = 0; p < f->n_planes; p++) {
        float dt = zo_dot32(f->planes + (size_t)p * f->d, q, f->d);
        if (out_dots) out_dots[p] = dt;
        out_signs[p] = ((double)dt + (double)f->consts[p]) >= 0.0;
    }
}

/* S-shard merge: per query, S lists of <= k (key,id) pairs sorted or not; keep the k
 * smallest by (key,id).  lists are [S][b][k] with counts [S][b]. */
ZO_EXPORT void zo_merge_topk(uint32_t S, uint64_t b, uint32_t k, const uint64_t *ids, const uint64_t *keys,
                             const uint32_t *counts, uint64_t *out_ids, uint64_t *out_keys, uint32_t *out_counts) {
    typedef struct { uint64_t id, key; } mp;
    mp *buf = malloc((size_t)S * k * sizeof(mp) + 16);
    for (uint64_t q = 0; q < b; q++) {
        uint32_t n = 0;
        for (uint32_t s = 0; s < S; s++)
            for (uint32_t i = 0; i < counts[s * b + q]; i++) {
                buf[n].id = ids[((size_t)s * b + q) * k + i];
                buf[n].key = keys[((size_t)s * b + q) * k + i];
                n++;
            }
        for (uint32_t i = 1; i < n; i++) { /* insertion sort: S*k is small */
            mp v = buf[i]; int64_t j = (int64_t)i - 1;
            while (j >= 0 && (buf[j].key > v.key || (buf[j].key == v.key && buf[j].id > v.id))) { buf[j + 1] = buf[j]; j--; }
            buf[j + 1] = v;
        }
        uint32_t m = n < k ? n : k;
        for (uint32_t i = 0; i < m; i++) { out_ids[q * k + i] = buf[i].id; out_keys[q * k + i] = buf[i].key; }
        out_counts[q] = m;
    }
    free(buf);
}

/* exact brute force over all rows (recall ground truth at oracle sizes) */
ZO_EXPORT void zo_brute_force(const float *X, uint64_t n, uint32_t d, const float *q, uint32_t k, int metric, int mode,
                              uint64_t *out_ids, uint64_t *out_keys) {
    zo_pair *r = malloc((n + 1) * sizeof(zo_pair));
    float qq = metric == ZO_COSINE ? sum_prod(q, q, d) : 0.0f;
    for (uint64_t i = 0; i < n; i++) {
        const float *a = X + i * d;
        r[i].id = (uint32_t)i;
        if (metric >= ZO_CHEBYSHEV) { r[i].key = key_generic(metric, mode, sums_generic(metric, mode, a, q, d)); continue; }
        r[i].key = metric == ZO_COSINE ? key_from_sums(metric, mode, sum_prod(a, q, d), sum_prod(a, a, d), qq, 0.0f)
                                       : key_from_sums(metric, mode, 0, 0, 0, sum_l2sq(a, q, d));
    }
    sort_n(r, n, sizeof(zo_pair), pair_cmp);
    for (uint32_t i = 0; i < k && i < n; i++) { out_ids[i] = r[i].id; out_keys[i] = r[i].key; }
    free(r);
}

ZO_EXPORT int zo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
