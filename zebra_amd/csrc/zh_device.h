// zh_device.h -- device helpers shared by the kernel translation units (zh_search.hip, zh_approx.hip): the canonical
// butterflies and sums, key finalisers, row loads, the LDS bitonic sort and the select partition.  Moved verbatim out of
// zh_search.hip in round 4 so that the half-width scan compiles on its own.
#pragma once
#include "zh_internal.h"

#define WAVE 64

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// The canonical 64-lane xor-butterfly (steps 1,2,4,8,16,32; lane l adds its partner group's value).  After a step
// every lane of a 2^k group holds the same value, so the partner may be ANY lane of the partner group: steps 1,2 are
// DPP quad_perm, 4 and 8 DPP row_half_mirror / row_mirror, 16 and 32 gfx950's v_permlane16_swap / v_permlane32_swap (a
// register-to-register exchange of 16- / 32-lane rows: both operands = s gives {my row pair's first value, its second} in
// the two results, and a + b == b + a bit for bit, as is max) -- no LDS round trip (round 2 used ds_swizzle for 16 and two
// v_readlane for 32: an lgkmcnt wait per scored (row, query) pair), no ds_bpermute; bit-identical to s + __shfl_xor(s, m).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
struct OpAdd { __device__ __forceinline__ static float f(float a, float b) { return a + b; } };
struct OpMax { __device__ __forceinline__ static float f(float a, float b) { return fmaxf(a, b); } };
template <class OP>
__device__ __forceinline__ float xor16(float s) {  // every lane: OP(value of the even 16-lane row of its pair, value of the odd one)
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return OP::f(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
template <class OP>
__device__ __forceinline__ float wave_butterfly(float s) {
    s = OP::f(s, dpp_mov<0xB1>(s));   // quad_perm [1,0,3,2]  : xor 1
    s = OP::f(s, dpp_mov<0x4E>(s));   // quad_perm [2,3,0,1]  : xor 2
    s = OP::f(s, dpp_mov<0x141>(s));  // row_half_mirror      : other quad of the 8
    s = OP::f(s, dpp_mov<0x140>(s));  // row_mirror           : other 8 of the 16
    s = xor16<OP>(s);
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);  // {lanes 0..31's value, lanes 32..63's}
    return OP::f(__uint_as_float(r[0]), __uint_as_float(r[1]));  // xor 32
}
__device__ __forceinline__ float wave_sum_canonical(float s) { return wave_butterfly<OpAdd>(s); }
__device__ __forceinline__ uint64_t f64_bits(double x) { return (uint64_t)__double_as_longlong(x); }

// simsimd cos(): cosine DISTANCE clipped at 0 with the two zero-norm cases; then distance.rs:23-25
__device__ __forceinline__ uint64_t key_cosine(float ab, float a2, float b2, int mode) {
    double c;
    if (a2 == 0.0f && b2 == 0.0f) c = 0.0;
    else if (ab == 0.0f) c = 1.0;
    else {
        double r = 1.0 - (double)ab / sqrt((double)a2 * (double)b2);
        c = r > 0.0 ? r : 0.0;
    }
    return f64_bits(mode == ZH_COSINE_PARITY ? 1.0 - c : c);
}
__device__ __forceinline__ uint64_t key_l2(float l2sq, int metric) {
    return f64_bits(metric == ZH_L2SQ ? (double)l2sq : sqrt((double)l2sq));
}

// ---- the `distances`-crate metrics (distance.rs:51-98,116-190): f32 result, key = f32 bits widened ----
enum { K_L2 = 0, K_COS = 1, K_MAX = 2, K_CANB = 3, K_BRAY = 4, K_ABS = 5, K_P3 = 6, K_P4 = 7, K_HAMM = 8, K_PP = 9 };

__host__ __device__ inline int zh_kind_of(int metric) {
    switch (metric) {
    case ZH_COSINE: return K_COS;
    case ZH_CHEBYSHEV: return K_MAX;
    case ZH_CANBERRA: return K_CANB;
    case ZH_BRAY_CURTIS: return K_BRAY;
    case ZH_MANHATTAN: return K_ABS;
    case ZH_L3: return K_P3;
    case ZH_L4: return K_P4;
    case ZH_HAMMING: return K_HAMM;
    case ZH_MINKOWSKI: case ZH_PNORM: return K_PP;
    default: return K_L2;
    }
}

__device__ __forceinline__ float powi_f32(float a, int b) {  // compiler-rt __powisf2 (Rust f32::powi)
    const bool recip = b < 0;
    float r = 1.0f;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0f / r : r;
}

// s^(1/p), 1 <= p <= 64: fixed Newton iteration in f64 (only + * /), bit-identical to the oracle's root_p
__device__ __forceinline__ double root_p(double s, int p) {
    if (!(s > 0.0) || s == (double)INFINITY || p == 1) return s;
    if (p == 2) return sqrt(s);
    uint64_t u = (uint64_t)__double_as_longlong(s);
    int e = (int)((u >> 52) & 0x7FF) - 1023;
    int fl = e >= 0 ? e / p : -((-e + p - 1) / p);
    int rem = e - fl * p;
    double y = ldexp((1.0 + (double)rem / (double)p) * (1.0 + 1.0 / (double)p), fl);
    for (int it = 0; it < 16; it++) {
        double yp = 1.0;
        for (int i = 0; i < p - 1; i++) yp *= y;
        y = ((double)(p - 1) * y + s / yp) / (double)p;
    }
    return y;
}

// s^(1/p), p > 64: exp(ln(s) / p) from fixed f64 series (the oracle's root_big, operation for operation)
__device__ __forceinline__ double root_big(double s, double p) {
    uint64_t u = (uint64_t)__double_as_longlong(s);
    int e = (int)((u >> 52) & 0x7FF) - 1023;
    u = (u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m = __longlong_as_double((long long)u);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double z = (m - 1.0) / (m + 1.0), z2 = z * z;
    double t = 0.0;
    for (int k = 25; k >= 1; k -= 2) t = t * z2 + 1.0 / (double)k;
    const double LN2 = 0.6931471805599453;
    const double x = ((double)e * LN2 + 2.0 * z * t) / p;
    const double xs = x / LN2;
    const int n = (int)(xs < 0.0 ? xs - 0.5 : xs + 0.5);
    const double r = x - (double)n * LN2;
    double term = 1.0, sum = 1.0;
    for (int k = 1; k <= 20; k++) { term = term * r / (double)k; sum = sum + term; }
    return ldexp(sum, n);
}

// distances::vectors::minkowski's `sum.powf(1 / p)` for ANY i32 p (distance.rs:160-174; Default is power 0): the oracle's root_any
__device__ __forceinline__ double root_any(double s, int p) {
    if (s != s) return s;
    if (p == 0) return s > 1.0 ? (double)INFINITY : (s == 1.0 ? 1.0 : 0.0);
    const long long ap = p < 0 ? -(long long)p : (long long)p;
    double r;
    if (!(s > 0.0) || s == (double)INFINITY || ap == 1) r = s;
    else r = ap <= 64 ? root_p(s, (int)ap) : root_big(s, (double)ap);
    if (p > 0) return r;
    if (r == 0.0) return (double)INFINITY;
    if (r == (double)INFINITY) return 0.0;
    return 1.0 / r;
}

// sums -> DistanceUnit for every metric; s0/s1 are the canonical sums, qq the query norm (cosine)
__device__ __forceinline__ uint64_t key_of(int metric, int param, float s0, float s1, float qq) {
    float f;
    switch (metric) {
    case ZH_COSINE: return key_cosine(s0, s1, qq, param);
    case ZH_L2SQ: case ZH_L2: return key_l2(s0, metric);
    case ZH_BRAY_CURTIS: f = s0 / s1; break;
    case ZH_L3: f = (float)root_p((double)s0, 3); break;
    case ZH_L4: f = sqrtf(sqrtf(s0)); break;
    case ZH_HAMMING: return (uint64_t)s0;
    case ZH_MINKOWSKI: f = (float)root_any((double)s0, param); break;
    default: f = s0; break;  // CHEBYSHEV, CANBERRA, MANHATTAN, PNORM
    }
    return (uint64_t)__float_as_uint(f);
}

// one element of the per-lane accumulation, by kind (a = stored, q = query)
template <int KIND>
__device__ __forceinline__ void acc_elem(float a, float q, float &x0, float &x1, int power) {
    if (KIND == K_L2) { float df = a - q; x0 = __builtin_fmaf(df, df, x0); }
    else if (KIND == K_COS) { x0 = __builtin_fmaf(a, q, x0); }
    else {
        float ad = fabsf(a - q);
        if (KIND == K_MAX) x0 = fmaxf(x0, ad);
        else if (KIND == K_CANB) x0 = x0 + ad / (fabsf(a) + fabsf(q));
        else if (KIND == K_BRAY) { x0 = x0 + ad; x1 = x1 + fabsf(a + q); }
        else if (KIND == K_ABS) x0 = x0 + ad;
        else if (KIND == K_P3) x0 = x0 + ad * (ad * ad);
        else if (KIND == K_P4) { float t = ad * ad; x0 = x0 + t * t; }
        else if (KIND == K_HAMM) x0 = x0 + (float)__popc((__float_as_uint(a) ^ __float_as_uint(q)) & 0xFFu);
        else x0 = x0 + powi_f32(ad, power);
    }
}
// Four consecutive elements of a lane (one float4 of the row against one of the query) into the lane's four accumulators.
// The two simsimd kinds use gfx950's PACKED f32 instructions (v_pk_add_f32 / v_pk_fma_f32: two IEEE operations per lane and
// instruction, the only way to the f32 VALU peak): (x, y) and (z, w) are aligned register pairs of the loaded float4, every
// component is the same fused / unfused operation as the scalar form, so the sums are bit-identical -- at half the vector
// instructions (the table scan spends 13-20 % of its time on them: a build without the arithmetic, profiles/r03_ab_scan_*).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__device__ __forceinline__ void acc4(const float4 &v, const float4 &q, float4 &a, float4 &e, int power) {
    if constexpr (KIND == K_L2 || KIND == K_COS) {
        const f32x2 v0 = {v.x, v.y}, v1 = {v.z, v.w}, q0 = {q.x, q.y}, q1 = {q.z, q.w};
        f32x2 a0 = {a.x, a.y}, a1 = {a.z, a.w};
        if constexpr (KIND == K_L2) {
            const f32x2 d0 = v0 - q0, d1 = v1 - q1;
            a0 = __builtin_elementwise_fma(d0, d0, a0);
            a1 = __builtin_elementwise_fma(d1, d1, a1);
        } else {
            a0 = __builtin_elementwise_fma(v0, q0, a0);
            a1 = __builtin_elementwise_fma(v1, q1, a1);
        }
        a.x = a0.x; a.y = a0.y; a.z = a1.x; a.w = a1.y;
    } else {
        acc_elem<KIND>(v.x, q.x, a.x, e.x, power);
        acc_elem<KIND>(v.y, q.y, a.y, e.y, power);
        acc_elem<KIND>(v.z, q.z, a.z, e.z, power);
        acc_elem<KIND>(v.w, q.w, a.w, e.w, power);
    }
}
// c += v * v, component-wise (the stored row's squared norm for cosine), packed
__device__ __forceinline__ void sq4(const float4 &v, float4 &c) {
    const f32x2 v0 = {v.x, v.y}, v1 = {v.z, v.w};
    f32x2 c0 = {c.x, c.y}, c1 = {c.z, c.w};
    c0 = __builtin_elementwise_fma(v0, v0, c0);
    c1 = __builtin_elementwise_fma(v1, v1, c1);
    c.x = c0.x; c.y = c0.y; c.z = c1.x; c.w = c1.y;
}
template <int KIND>
__device__ __forceinline__ float wave_combine(float x, float y, float z, float w) {
    if (KIND == K_MAX) return wave_butterfly<OpMax>(fmaxf(fmaxf(x, y), fmaxf(z, w)));
    return wave_sum_canonical((x + y) + (z + w));
}


// ------------------------------------------------------------------------------------------------
// canonical row sums: one wave per row, lane-strided float4
// ------------------------------------------------------------------------------------------------
// generic (runtime d, any d): element e handled by lane (e mod 256)/4, component e mod 4
template <int KIND>
__device__ __forceinline__ void lane_sums_generic(const float *__restrict__ a, const float *__restrict__ q,
                                                  uint32_t d, uint32_t lane, int power, float &o_s0, float &o_s1) {
    float x0[4] = {0, 0, 0, 0}, x1[4] = {0, 0, 0, 0};
    for (uint32_t base = 0; base < d; base += 256) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            uint32_t e = base + 4 * lane + t;
            if (e < d) {
                float av = a[e];
                acc_elem<KIND>(av, q[e], x0[t], x1[t], power);
                if (KIND == K_COS) x1[t] = __builtin_fmaf(av, av, x1[t]);
            }
        }
    }
    o_s0 = wave_combine<KIND>(x0[0], x0[1], x0[2], x0[3]);
    o_s1 = (KIND == K_COS || KIND == K_BRAY) ? wave_combine<K_L2>(x1[0], x1[1], x1[2], x1[3]) : 0.0f;
}


template <int D>
struct RowVec {
    static constexpr int NJ = D / 256;            // full 1-KiB pieces
    static constexpr int REM4 = (D % 256) / 4;    // lanes holding the last, partial piece
    static constexpr int NV = NJ + (REM4 ? 1 : 0);
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld16(const float4 *p) {
    if (NT) {  // streamed once: non-temporal hint (global_load_dwordx4 ... nt)
        f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
        return make_float4(t.x, t.y, t.z, t.w);
    }
    return *p;
}
template <int D, bool NT = false>
__device__ __forceinline__ void load_row(const float *__restrict__ row, uint32_t lane, float4 *v) {
    const float4 *r4 = reinterpret_cast<const float4 *>(row);
#pragma unroll
    for (int j = 0; j < RowVec<D>::NJ; j++) v[j] = ld16<NT>(r4 + lane + 64 * j);
    if (RowVec<D>::REM4) {
        v[RowVec<D>::NJ] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < (uint32_t)RowVec<D>::REM4) v[RowVec<D>::NJ] = ld16<NT>(r4 + lane + 64 * RowVec<D>::NJ);
    }
}


// one stored row (in the canonical lane layout) against one query: the canonical sums of the pair
template <int D, int KIND>
__device__ __forceinline__ void row_pair_sums(const float4 *v, const float4 *q, uint32_t lane, int power, float &s0, float &s1) {
    constexpr int NV = RowVec<D>::NV;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), e = a;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const bool act = (j < RowVec<D>::NJ) || (lane < (uint32_t)RowVec<D>::REM4);
        if (act) acc4<KIND>(v[j], q[j], a, e, power);
    }
    s0 = wave_combine<KIND>(a.x, a.y, a.z, a.w);
    if (KIND == K_BRAY) s1 = wave_combine<K_L2>(e.x, e.y, e.z, e.w);
}

// ------------------------------------------------------------------------------------------------
// LDS bitonic sort of (key, id) ascending; n is a power of two
// ------------------------------------------------------------------------------------------------
template <typename IdT>
__device__ __forceinline__ void block_bitonic_sort(uint64_t *sk, IdT *si, uint32_t n) {
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    for (uint32_t size = 2; size <= n; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t i = tid; i < (n >> 1); i += nt) {
                uint32_t lo = ((i & ~(stride - 1)) << 1) | (i & (stride - 1));
                uint32_t hi = lo + stride;
                bool asc = (lo & size) == 0;
                uint64_t ka = sk[lo], kb = sk[hi];
                IdT ia = si[lo], ib = si[hi];
                bool gt = ka > kb || (ka == kb && ia > ib);
                if (gt == asc) { sk[lo] = kb; sk[hi] = ka; si[lo] = ib; si[hi] = ia; }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ uint32_t next_pow2(uint32_t x) {
    uint32_t p = 2;
    while (p < x) p <<= 1;
    return p;
}

// per visit: the `take` smallest (key, id) of the leaf (lsh.rs:317-323); take == len copies all.
// The candidates of a query are sorted again by the final kernel, so a visit's slice of the pool
// need not be ordered: select = partition.  Fast path (leaf fits the LDS buffer): histogram
// refinement of the unsigned key range, 8 bits per round, until the bucket that holds the take-th
// key is small; everything below it is emitted as is, the bucket itself is sorted by (key, id).
// Slow path (leaf longer than the buffer, or a large group of equal keys): streaming bitonic sort.
#define SEL_SMALL 512

__device__ __forceinline__ void select_slow(const ZhVisit &v, const uint32_t *__restrict__ leaf_ids,
                                            const uint64_t *__restrict__ keys, uint64_t *__restrict__ cand_keys,
                                            uint32_t *__restrict__ cand_ids, uint64_t *sk, uint32_t *si, uint32_t cap) {
    const uint32_t tid = threadIdx.x;
    uint32_t have = 0, pos = 0;
    while (pos < v.len) {
        uint32_t m = cap - have;
        if (m > v.len - pos) m = v.len - pos;
        for (uint32_t i = tid; i < m; i += 256) {
            sk[have + i] = keys[v.row_off + pos + i];
            si[have + i] = leaf_ids[(size_t)v.leaf_off + pos + i];
        }
        uint32_t total = have + m, np2 = next_pow2(total);
        for (uint32_t i = total + tid; i < np2; i += 256) { sk[i] = ~0ull; si[i] = ~0u; }
        block_bitonic_sort<uint32_t>(sk, si, np2);
        have = total < v.take ? total : v.take;
        pos += m;
    }
    for (uint32_t i = tid; i < have; i += 256) {
        cand_keys[v.cand_off + i] = sk[i];
        cand_ids[v.cand_off + i] = si[i];
    }
}

// KEY(i): the visit's i-th key, from the LDS copy when the leaf fits it, else straight from the
// (L2 / Infinity-Cache resident) key scratch -- a few 8-B passes against the 4*d bytes the sweep read
template <bool IN_LDS>
__device__ __forceinline__ void select_fast(const ZhVisit &v, const uint32_t *__restrict__ leaf_ids,
                                            const uint64_t *__restrict__ gkeys, uint64_t *__restrict__ cand_keys,
                                            uint32_t *__restrict__ cand_ids, uint64_t *sk, uint32_t *si,
                                            uint32_t *s_u32, bool &need_slow) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // carve-up of si[]: tk (SEL_SMALL u64) | ti (SEL_SMALL u32) | hist (256 u32) | wmin/wmax (8 u64)
    uint64_t *tk = reinterpret_cast<uint64_t *>(si);
    uint32_t *ti = si + 2 * SEL_SMALL;
    uint32_t *hist = si + 3 * SEL_SMALL;
    uint64_t *wred = reinterpret_cast<uint64_t *>(si + 3 * SEL_SMALL + 256);
    const uint64_t *kp = gkeys + v.row_off;
#define SEL_KEY(i) (IN_LDS ? sk[i] : kp[i])
    uint64_t kmin = ~0ull, kmax = 0;
    for (uint32_t i = tid; i < v.len; i += 256) {
        uint64_t k = kp[i];
        if (IN_LDS) sk[i] = k;
        kmin = k < kmin ? k : kmin;
        kmax = k > kmax ? k : kmax;
    }
    for (int m = 1; m < 64; m <<= 1) {
        uint64_t a = __shfl_xor(kmin, m), b = __shfl_xor(kmax, m);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    if (lane == 0) { wred[wv] = kmin; wred[4 + wv] = kmax; }
    __syncthreads();
    uint64_t lo = wred[0], hi = wred[4];
    for (int w = 1; w < 4; w++) { lo = wred[w] < lo ? wred[w] : lo; hi = wred[4 + w] > hi ? wred[4 + w] : hi; }
    uint32_t need = v.take;   // how many to take from [lo, hi]; every key < lo is already taken
    uint32_t inb = v.len;     // keys inside [lo, hi]
    while (inb > SEL_SMALL && inb != need) {
        uint64_t range = hi - lo;
        if (range == 0) break;  // > SEL_SMALL equal keys: slow path
        int sh = 64 - __clzll((long long)range) - 8;
        if (sh < 0) sh = 0;
        __syncthreads();
        hist[tid] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < v.len; i += 256) {
            uint64_t k = SEL_KEY(i);
            if (k >= lo && k <= hi) atomicAdd(&hist[(uint32_t)((k - lo) >> sh)], 1u);
        }
        __syncthreads();
        if (wv == 0) {  // find the bucket holding the need-th key
            uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            uint32_t ssum = h0 + h1 + h2 + h3, inc = ssum;
            for (int m = 1; m < 64; m <<= 1) {
                uint32_t t = __shfl_up(inc, m);
                if ((int)lane >= m) inc += t;
            }
            uint32_t exc = inc - ssum;
            bool mine = exc < need && need <= inc;
            if (mine) {
                uint32_t c = exc, j = 4 * lane, hb = h0;
                if (need > c + h0) { c += h0; j++; hb = h1;
                    if (need > c + h1) { c += h1; j++; hb = h2;
                        if (need > c + h2) { c += h2; j++; hb = h3; } } }
                s_u32[0] = j; s_u32[1] = c; s_u32[2] = hb;
            }
        }
        __syncthreads();
        uint32_t j = s_u32[0], before = s_u32[1];
        inb = s_u32[2];
        need -= before;
        uint64_t nlo = lo + ((uint64_t)j << sh);
        uint64_t nhi = nlo + ((1ull << sh) - 1);
        if (nhi < nlo) nhi = hi;
        lo = nlo;
        hi = nhi < hi ? nhi : hi;
    }
    if (inb > SEL_SMALL && inb != need) { need_slow = true; return; }
    const bool take_all_bucket = (inb == need);
    // emit: keys < lo (and the whole bucket when it is taken whole) straight to the pool; otherwise the
    // bucket's members go to the small sort buffer
    __syncthreads();
    if (tid == 0) { s_u32[3] = 0; s_u32[4] = 0; }
    __syncthreads();
    for (uint32_t i = tid; i < v.len; i += 256) {
        uint64_t k = SEL_KEY(i);
        bool below = k < lo || (take_all_bucket && k <= hi);
        bool inside = !take_all_bucket && k >= lo && k <= hi;
        if (below) {
            uint32_t o = atomicAdd(&s_u32[3], 1u);
            cand_keys[v.cand_off + o] = k;
            cand_ids[v.cand_off + o] = leaf_ids[(size_t)v.leaf_off + i];
        } else if (inside) {
            uint32_t o = atomicAdd(&s_u32[4], 1u);
            if (o < SEL_SMALL) { tk[o] = k; ti[o] = leaf_ids[(size_t)v.leaf_off + i]; }
        }
    }
    __syncthreads();
    if (!take_all_bucket && need > 0) {
        uint32_t nb = s_u32[4] < SEL_SMALL ? s_u32[4] : SEL_SMALL, base = s_u32[3];
        uint32_t np2 = next_pow2(nb);
        for (uint32_t i = nb + tid; i < np2; i += 256) { tk[i] = ~0ull; ti[i] = ~0u; }
        block_bitonic_sort<uint32_t>(tk, ti, np2);
        for (uint32_t i = tid; i < need && i < nb; i += 256) {
            cand_keys[v.cand_off + base + i] = tk[i];
            cand_ids[v.cand_off + base + i] = ti[i];
        }
    }
#undef SEL_KEY
}


__device__ __forceinline__ uint32_t f32_sortable(float x) {
    const uint32_t u = __float_as_uint(x);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// ---- the leaf-major sweeps' view of a batch: flat rows = the rows of group 0, then group 1, ... (moved out of zh_search.hip) ----
// lane i of a sweep wave -> (group, stored row, position in the group) of flat row r0 + i.  With the wave-start table the
// search is confined to the <= 64 groups the wave's 64 rows can span (every group has at least one row), and skipped when
// the whole wave lies in one group (the common case with leaves of thousands of rows).
__device__ __forceinline__ void resolve_flat_rows(uint64_t r0, uint32_t cnt, uint32_t lane, const ZhGroup *__restrict__ groups,
                                                  const uint64_t *__restrict__ groupRowOff, uint64_t n_groups,
                                                  const uint32_t *__restrict__ waveGroup, const uint32_t *__restrict__ leaf_ids,
                                                  uint32_t &my_g, uint32_t &my_id, uint32_t &my_within,
                                                  uint32_t *my_leaf_off = nullptr, uint32_t *my_len = nullptr) {
    const uint64_t r = r0 + (lane < cnt ? lane : cnt - 1);
    uint64_t lo = 0, hi = n_groups;  // last group with row offset <= r
    if (waveGroup) {
        lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)waveGroup[r0 >> 6]);
        hi = lo + 64 < n_groups ? lo + 64 : n_groups;
        if (lo + 1 >= n_groups || groupRowOff[lo + 1] > r0 + cnt - 1) hi = lo + 1;  // wave-uniform: one group
    }
    while (hi - lo > 1) {
        uint64_t mid = (lo + hi) >> 1;
        if (groupRowOff[mid] <= r) lo = mid; else hi = mid;
    }
    my_g = (uint32_t)lo;
    my_within = (uint32_t)(r - groupRowOff[lo]);
    const uint32_t lo_off = groups[lo].leaf_off;
    if (my_leaf_off) *my_leaf_off = lo_off;
    if (my_len) *my_len = groups[lo].len;
    my_id = leaf_ids ? leaf_ids[(size_t)lo_off + my_within] : lo_off + my_within;
}
