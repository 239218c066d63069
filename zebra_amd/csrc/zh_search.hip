// zh_search.hip -- gfx950 kernels of the query path:
//   hash_dense   S = Q.W^T for the leading planes of the forest with v_mfma_f32_32x32x2_f32, sign bits
//                (Hyperplane::point_is_above, /root/reference/src/database/index/lsh.rs:39-43)
//   walk         tree_result's control flow (lsh.rs:290-348) -> list of leaf visits (leaf, n)
//   sweep        Metric::distance(stored, query) for every row of every visited leaf, HBM-bound
//                (lsh.rs:311-316 / 557-560, src/distance.rs:19-49,103-114)
//   select       per-leaf ascending sort by key, take n (lsh.rs:317-323)
//   final        union over trees, sort, take top_k (lsh.rs:557-564)
//   merge        shard merge after the all-gather (new)
//
// Arithmetic order contract (identical in oracle/zebra_oracle.c):
//   hash dot      : sequential k-ascending f32 fma chain from +0 (what a non-split-K f32 MFMA computes)
//   distance sums : element e -> accumulator (e mod 256) [lane = (e mod 256)/4, component e mod 4],
//                   ascending e, fma; per lane ((x+y)+(z+w)); wave xor-butterfly 1,2,4,8,16,32
//   ordering      : unsigned (key, id)
#include "zh_internal.h"

#define WAVE 64

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum_canonical(float s) {
    s = s + __shfl_xor(s, 1);
    s = s + __shfl_xor(s, 2);
    s = s + __shfl_xor(s, 4);
    s = s + __shfl_xor(s, 8);
    s = s + __shfl_xor(s, 16);
    s = s + __shfl_xor(s, 32);
    return s;
}

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

// ------------------------------------------------------------------------------------------------
// hash_dense: 64 queries x 64 planes per block, 4 waves each owning a 32x32 tile,
// K staged 32 at a time through LDS in [k][row] order so that a lane's MFMA operand
// (row = lane&31, k = lane>>5) is a conflict-free ds_read_b32.
// ------------------------------------------------------------------------------------------------
#define HD_KT 32
#define HD_PITCH 65

template <bool WRITE_DOTS>
__global__ __launch_bounds__(256) void hash_dense_kernel(const float *__restrict__ Q, uint32_t B,
                                                          const float *__restrict__ W,
                                                          const float *__restrict__ C, uint32_t P, uint32_t d,
                                                          uint32_t *__restrict__ bits, uint32_t wpq,
                                                          float *__restrict__ dots) {
    __shared__ float Qs[HD_KT][HD_PITCH];
    __shared__ float Ws[HD_KT][HD_PITCH];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t wr = wv >> 1, wc = wv & 1;
    const uint32_t p0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    const bool vec4 = (d & 3u) == 0;
    for (uint32_t k0 = 0; k0 < d; k0 += HD_KT) {
#pragma unroll
        for (int it = 0; it < 2; it++) {
            uint32_t i = tid + it * 256;  // 512 float4 per tile
            uint32_t row = i >> 3, c4 = (i & 7) * 4;
            uint32_t k = k0 + c4;
            float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), wvv = qv;
            if (vec4) {
                if (k < d) {
                    if (b0 + row < B) qv = *reinterpret_cast<const float4 *>(Q + (size_t)(b0 + row) * d + k);
                    if (p0 + row < P) wvv = *reinterpret_cast<const float4 *>(W + (size_t)(p0 + row) * d + k);
                }
            } else {
                float t[4] = {0, 0, 0, 0}, u[4] = {0, 0, 0, 0};
                for (int e = 0; e < 4; e++)
                    if (k + e < d) {
                        if (b0 + row < B) t[e] = Q[(size_t)(b0 + row) * d + k + e];
                        if (p0 + row < P) u[e] = W[(size_t)(p0 + row) * d + k + e];
                    }
                qv = make_float4(t[0], t[1], t[2], t[3]);
                wvv = make_float4(u[0], u[1], u[2], u[3]);
            }
            Qs[c4 + 0][row] = qv.x; Qs[c4 + 1][row] = qv.y; Qs[c4 + 2][row] = qv.z; Qs[c4 + 3][row] = qv.w;
            Ws[c4 + 0][row] = wvv.x; Ws[c4 + 1][row] = wvv.y; Ws[c4 + 2][row] = wvv.z; Ws[c4 + 3][row] = wvv.w;
        }
        __syncthreads();
        const uint32_t r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < HD_KT; kk += 2) {
            float a = Qs[kk + h][wr * 32 + r];   // A[i = query r][k = kk + h]
            float bb = Ws[kk + h][wc * 32 + r];  // B[k = kk + h][j = plane r]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D map of 32x32: col (plane) = lane & 31, row (query) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const uint32_t plane = p0 + wc * 32 + (lane & 31);
    const double cval = plane < P ? (double)C[plane] : 0.0;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        uint32_t qrow_lo = b0 + wr * 32 + (reg & 3) + 8 * (reg >> 2);
        uint32_t qrow = qrow_lo + 4 * (lane >> 5);
        float s = acc[reg];
        bool above = ((double)s + cval) >= 0.0;  // lsh.rs:40-42, NaN -> false
        unsigned long long m = __ballot(above);
        uint32_t word = (p0 + wc * 32) >> 5;
        if (lane == 0 && qrow_lo < B) bits[(size_t)qrow_lo * wpq + word] = (uint32_t)m;
        if (lane == 32 && qrow_lo + 4 < B) bits[(size_t)(qrow_lo + 4) * wpq + word] = (uint32_t)(m >> 32);
        if (WRITE_DOTS && qrow < B && plane < P) dots[(size_t)qrow * P + plane] = s;
    }
}

hipError_t zh_launch_hash_dense(const float *dQ, uint32_t B, const float *dPlanes, const float *dConsts,
                                uint32_t P, uint32_t d, uint32_t *dBits, uint32_t words_per_q, float *dDots,
                                hipStream_t s) {
    if (B == 0 || P == 0) return hipSuccess;
    dim3 grid((P + 63) / 64, (B + 63) / 64);
    if (dDots)
        hipLaunchKernelGGL(hash_dense_kernel<true>, grid, dim3(256), 0, s, dQ, B, dPlanes, dConsts, P, d, dBits,
                           words_per_q, dDots);
    else
        hipLaunchKernelGGL(hash_dense_kernel<false>, grid, dim3(256), 0, s, dQ, B, dPlanes, dConsts, P, d, dBits,
                           words_per_q, dDots);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// canonical row sums: one wave per row, lane-strided float4
// ------------------------------------------------------------------------------------------------
// generic (runtime d, any d): element e handled by lane (e mod 256)/4, component e mod 4
__device__ __forceinline__ void lane_sums_generic(const float *__restrict__ a, const float *__restrict__ q,
                                                  uint32_t d, uint32_t lane, bool cosine, float &o_ab, float &o_a2,
                                                  float &o_l2) {
    float ab[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0}, l2[4] = {0, 0, 0, 0};
    for (uint32_t base = 0; base < d; base += 256) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            uint32_t e = base + 4 * lane + t;
            if (e < d) {
                float av = a[e], qv = q[e];
                if (cosine) {
                    ab[t] = __builtin_fmaf(av, qv, ab[t]);
                    a2[t] = __builtin_fmaf(av, av, a2[t]);
                } else {
                    float df = av - qv;
                    l2[t] = __builtin_fmaf(df, df, l2[t]);
                }
            }
        }
    }
    o_ab = wave_sum_canonical((ab[0] + ab[1]) + (ab[2] + ab[3]));
    o_a2 = wave_sum_canonical((a2[0] + a2[1]) + (a2[2] + a2[3]));
    o_l2 = wave_sum_canonical((l2[0] + l2[1]) + (l2[2] + l2[3]));
}

// sum_prod(q, q): the query-side norm of the cosine metric, one wave per query
__global__ __launch_bounds__(256) void qnorm_kernel(const float *__restrict__ Q, uint32_t B, uint32_t d,
                                                     float *__restrict__ QQ) {
    uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (w >= B) return;
    const float *q = Q + (size_t)w * d;
    float acc[4] = {0, 0, 0, 0};
    for (uint32_t base = 0; base < d; base += 256) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            uint32_t e = base + 4 * lane + t;
            if (e < d) acc[t] = __builtin_fmaf(q[e], q[e], acc[t]);
        }
    }
    float s = wave_sum_canonical((acc[0] + acc[1]) + (acc[2] + acc[3]));
    if (lane == 0) QQ[w] = s;
}

hipError_t zh_launch_qnorm(const float *dQ, uint32_t B, uint32_t d, float *dQQ, hipStream_t s) {
    if (!B) return hipSuccess;
    hipLaunchKernelGGL(qnorm_kernel, dim3((B + 3) / 4), dim3(256), 0, s, dQ, B, d, dQQ);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// walk: one lane per (query, tree).  Control flow of tree_result depends only on hash signs, leaf
// lengths and n -- never on distances -- so it runs ahead of the sweep and emits the visit list.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool plane_above_on_demand(const float *__restrict__ w, float c,
                                                      const float *__restrict__ q, uint32_t d) {
    float acc = 0.0f;
    if ((d & 3u) == 0) {
        const float4 *w4 = reinterpret_cast<const float4 *>(w);
        const float4 *q4 = reinterpret_cast<const float4 *>(q);
        for (uint32_t k = 0; k < d / 4; k++) {
            float4 a = w4[k], x = q4[k];
            acc = __builtin_fmaf(a.x, x.x, acc);
            acc = __builtin_fmaf(a.y, x.y, acc);
            acc = __builtin_fmaf(a.z, x.z, acc);
            acc = __builtin_fmaf(a.w, x.w, acc);
        }
    } else {
        for (uint32_t k = 0; k < d; k++) acc = __builtin_fmaf(w[k], q[k], acc);
    }
    return ((double)acc + (double)c) >= 0.0;
}

#define WALK_STACK 64

template <bool EMIT>
__global__ __launch_bounds__(64) void walk_kernel(ZhForestDev f, const float *__restrict__ Q, uint32_t B, uint32_t d,
                                                   int32_t n, const uint32_t *__restrict__ bits, uint32_t wpq,
                                                   uint32_t P_dense, ZhPairCounts *__restrict__ counts,
                                                   ZhVisit *__restrict__ inl, const uint64_t *__restrict__ rowBase,
                                                   const uint64_t *__restrict__ candBase,
                                                   const uint64_t *__restrict__ visitBase,
                                                   ZhVisit *__restrict__ visits, uint64_t *__restrict__ visitRowOff) {
    const uint32_t T = f.n_trees;
    const uint64_t pair = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= (uint64_t)B * T) return;
    const uint32_t b = (uint32_t)(pair / T), t = (uint32_t)(pair % T);
    if (EMIT) {
        uint32_t nv = counts[pair].visits;
        if (nv <= ZH_INLINE_VISITS) {
            uint64_t vb = visitBase[pair], rb = rowBase[pair], cb = candBase[pair];
            for (uint32_t i = 0; i < nv; i++) {
                ZhVisit v = inl[pair * ZH_INLINE_VISITS + i];
                v.row_off += rb;
                v.cand_off += cb;
                visits[vb + i] = v;
                visitRowOff[vb + i] = v.row_off;
            }
            return;
        }
    }
    const float *q = Q + (size_t)b * d;
    int32_t st_node[WALK_STACK], st_n[WALK_STACK];
    int sp = 0;
    int32_t cur = (int32_t)f.roots[t], ncur = n;
    uint32_t nv = 0;
    uint64_t nrows = 0, ntakes = 0;
    uint64_t vb = 0, rb = 0, cb = 0;
    if (EMIT) { vb = visitBase[pair]; rb = rowBase[pair]; cb = candBase[pair]; }
    for (;;) {
        int32_t p;
        while ((p = f.node_plane[cur]) >= 0) {
            bool above;
            if ((uint32_t)p < P_dense) above = (bits[(size_t)b * wpq + ((uint32_t)p >> 5)] >> (p & 31)) & 1u;
            else above = plane_above_on_demand(f.planes + (size_t)p * d, f.consts[p], q, d);
            int32_t l = f.node_left[cur], r = f.node_right[cur];
            if (sp < WALK_STACK) { st_node[sp] = above ? l : r; st_n[sp] = ncur; }
            sp++;
            cur = above ? r : l;  // lsh.rs:335-338: above -> right is main
        }
        uint32_t off = (uint32_t)f.node_left[cur], len = (uint32_t)f.node_right[cur];
        uint32_t take = ncur <= 0 ? 0u : (len < (uint32_t)ncur ? len : (uint32_t)ncur);
        int32_t ret = (int32_t)take;  // lsh.rs:306 / 329
        if (take > 0) {
            ZhVisit v;
            v.b = b; v.leaf_off = off; v.len = len; v.take = take;
            if (EMIT) {
                v.row_off = rb + nrows; v.cand_off = cb + ntakes;
                visits[vb + nv] = v;
                visitRowOff[vb + nv] = v.row_off;
            } else if (nv < ZH_INLINE_VISITS) {
                v.row_off = nrows; v.cand_off = ntakes;
                inl[pair * ZH_INLINE_VISITS + nv] = v;
            }
            nv++; nrows += len; ntakes += take;
        }
        bool down = false;
        while (sp > 0) {
            sp--;
            if (sp < WALK_STACK && ret < st_n[sp]) {  // lsh.rs:341-343: k < n -> the backup's count alone
                cur = st_node[sp];
                ncur = st_n[sp] - ret;
                down = true;
                break;
            }
        }
        if (!down) break;
    }
    if (!EMIT) {
        ZhPairCounts c;
        c.visits = nv; c.rows = (uint32_t)nrows; c.takes = (uint32_t)ntakes; c.pad = 0;
        counts[pair] = c;
    }
}

hipError_t zh_launch_walk_count(ZhForestDev f, const float *dQ, uint32_t B, uint32_t d, int32_t n,
                                const uint32_t *dBits, uint32_t words_per_q, uint32_t P_dense, ZhPairCounts *dCounts,
                                ZhVisit *dInline, hipStream_t s) {
    uint64_t pairs = (uint64_t)B * f.n_trees;
    if (!pairs) return hipSuccess;
    hipLaunchKernelGGL(walk_kernel<false>, dim3((uint32_t)((pairs + 63) / 64)), dim3(64), 0, s, f, dQ, B, d, n, dBits,
                       words_per_q, P_dense, dCounts, dInline, nullptr, nullptr, nullptr, nullptr, nullptr);
    return hipGetLastError();
}
hipError_t zh_launch_walk_emit(ZhForestDev f, const float *dQ, uint32_t B, uint32_t d, int32_t n,
                               const uint32_t *dBits, uint32_t words_per_q, uint32_t P_dense,
                               const ZhPairCounts *dCounts, const ZhVisit *dInline, const uint64_t *dRowBase,
                               const uint64_t *dCandBase, const uint64_t *dVisitBase, ZhVisit *dVisits,
                               uint64_t *dVisitRowOff, hipStream_t s) {
    uint64_t pairs = (uint64_t)B * f.n_trees;
    if (!pairs) return hipSuccess;
    hipLaunchKernelGGL(walk_kernel<true>, dim3((uint32_t)((pairs + 63) / 64)), dim3(64), 0, s, f, dQ, B, d, n, dBits,
                       words_per_q, P_dense, const_cast<ZhPairCounts *>(dCounts), const_cast<ZhVisit *>(dInline),
                       dRowBase, dCandBase, dVisitBase, dVisits, dVisitRowOff);
    return hipGetLastError();
}

// exclusive scans of the three per-pair counts (single block; pairs <= a few 100k)
__global__ __launch_bounds__(1024) void pair_scan_kernel(const ZhPairCounts *__restrict__ counts, uint32_t n,
                                                          uint64_t *__restrict__ rowBase,
                                                          uint64_t *__restrict__ candBase,
                                                          uint64_t *__restrict__ visitBase,
                                                          ZhTotals *__restrict__ totals) {
    __shared__ uint64_t sr[1024], sc[1024], sv[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = tid * per, hi = lo + per < n ? lo + per : n;
    uint64_t r = 0, c = 0, v = 0;
    for (uint32_t i = lo; i < hi; i++) { r += counts[i].rows; c += counts[i].takes; v += counts[i].visits; }
    sr[tid] = r; sc[tid] = c; sv[tid] = v;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint64_t ar = 0, ac = 0, av = 0;
        if (tid >= off) { ar = sr[tid - off]; ac = sc[tid - off]; av = sv[tid - off]; }
        __syncthreads();
        sr[tid] += ar; sc[tid] += ac; sv[tid] += av;
        __syncthreads();
    }
    uint64_t br = sr[tid] - r, bc = sc[tid] - c, bv = sv[tid] - v;
    for (uint32_t i = lo; i < hi; i++) {
        rowBase[i] = br; candBase[i] = bc; visitBase[i] = bv;
        br += counts[i].rows; bc += counts[i].takes; bv += counts[i].visits;
    }
    if (tid == 1023) {
        rowBase[n] = sr[1023]; candBase[n] = sc[1023]; visitBase[n] = sv[1023];
        totals->rows = sr[1023]; totals->takes = sc[1023]; totals->visits = sv[1023]; totals->flags = 0;
    }
}

hipError_t zh_launch_pair_scan(const ZhPairCounts *dCounts, uint32_t n_pairs, uint64_t *dRowBase,
                               uint64_t *dCandBase, uint64_t *dVisitBase, ZhTotals *dTotals, hipStream_t s) {
    hipLaunchKernelGGL(pair_scan_kernel, dim3(1), dim3(1024), 0, s, dCounts, n_pairs, dRowBase, dCandBase, dVisitBase,
                       dTotals);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// sweep: the HBM-bound kernel.  The rows of all visits of the batch form one flat sequence
// [0, R_total); each wave owns 64 consecutive flat rows: lane i resolves flat row r0+i to
// (visit -> query b, stored row id), then the wave streams the 64 rows one after another --
// every lane loading one float4 per 1 KiB of row (coalesced 1 KiB wave loads), RG rows in flight --
// and lane i keeps the canonical sums of row i.  Keys go to the scratch (8 B per 4*D B read).
// ------------------------------------------------------------------------------------------------
template <int D>
struct RowVec {
    static constexpr int NJ = D / 256;            // full 1-KiB pieces
    static constexpr int REM4 = (D % 256) / 4;    // lanes holding the last, partial piece
    static constexpr int NV = NJ + (REM4 ? 1 : 0);
};

template <int D>
__device__ __forceinline__ void load_row(const float *__restrict__ row, uint32_t lane, float4 *v) {
    const float4 *r4 = reinterpret_cast<const float4 *>(row);
#pragma unroll
    for (int j = 0; j < RowVec<D>::NJ; j++) v[j] = r4[lane + 64 * j];
    if (RowVec<D>::REM4) {
        v[RowVec<D>::NJ] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < (uint32_t)RowVec<D>::REM4) v[RowVec<D>::NJ] = r4[lane + 64 * RowVec<D>::NJ];
    }
}

template <int D, bool COSINE>
__device__ __forceinline__ void row_sums(const float4 *v, const float4 *q, uint32_t lane, float &s0, float &s1) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
#pragma unroll
    for (int j = 0; j < RowVec<D>::NV; j++) {
        bool act = (j < RowVec<D>::NJ) || (lane < (uint32_t)RowVec<D>::REM4);
        if (act) {
            if (COSINE) {
                a.x = __builtin_fmaf(v[j].x, q[j].x, a.x); a.y = __builtin_fmaf(v[j].y, q[j].y, a.y);
                a.z = __builtin_fmaf(v[j].z, q[j].z, a.z); a.w = __builtin_fmaf(v[j].w, q[j].w, a.w);
                c.x = __builtin_fmaf(v[j].x, v[j].x, c.x); c.y = __builtin_fmaf(v[j].y, v[j].y, c.y);
                c.z = __builtin_fmaf(v[j].z, v[j].z, c.z); c.w = __builtin_fmaf(v[j].w, v[j].w, c.w);
            } else {
                float dx = v[j].x - q[j].x, dy = v[j].y - q[j].y, dz = v[j].z - q[j].z, dw = v[j].w - q[j].w;
                a.x = __builtin_fmaf(dx, dx, a.x); a.y = __builtin_fmaf(dy, dy, a.y);
                a.z = __builtin_fmaf(dz, dz, a.z); a.w = __builtin_fmaf(dw, dw, a.w);
            }
        }
    }
    s0 = wave_sum_canonical((a.x + a.y) + (a.z + a.w));
    s1 = COSINE ? wave_sum_canonical((c.x + c.y) + (c.z + c.w)) : 0.0f;
}

#define SWEEP_RG 4

// D > 0: compile-time dimension (multiple of 4); D == 0: runtime d, any value (slow path)
template <int D, bool COSINE>
__global__ __launch_bounds__(256) void sweep_kernel(const float *__restrict__ X, uint32_t d,
                                                     const float *__restrict__ Q, const float *__restrict__ QQ,
                                                     const ZhVisit *__restrict__ visits,
                                                     const uint64_t *__restrict__ visitRowOff, uint64_t n_visits,
                                                     const uint32_t *__restrict__ leaf_ids, uint64_t R_total,
                                                     int metric, int mode, uint64_t *__restrict__ keys) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t r0 = wave * 64;
    if (r0 >= R_total) return;
    const uint32_t cnt = (uint32_t)(R_total - r0 < 64 ? R_total - r0 : 64);
    // lane i -> (query, stored row) of flat row r0 + i
    uint32_t my_b = 0, my_id = 0;
    {
        uint64_t r = r0 + (lane < cnt ? lane : cnt - 1);
        uint64_t lo = 0, hi = n_visits;  // last visit with row_off <= r
        while (hi - lo > 1) {
            uint64_t mid = (lo + hi) >> 1;
            if (visitRowOff[mid] <= r) lo = mid; else hi = mid;
        }
        ZhVisit v = visits[lo];
        uint32_t within = (uint32_t)(r - v.row_off);
        my_b = v.b;
        my_id = leaf_ids ? leaf_ids[(size_t)v.leaf_off + within] : v.leaf_off + within;
    }
    float mine0 = 0.f, mine1 = 0.f, mine_qq = 0.f;
    if (D > 0) {
        constexpr int NV = RowVec<(D > 0 ? D : 4)>::NV;
        float4 q[NV];
        uint32_t cur_b = 0xFFFFFFFFu;
        float cur_qq = 0.f;
        for (uint32_t i0 = 0; i0 < cnt; i0 += SWEEP_RG) {
            float4 v[SWEEP_RG][NV];
#pragma unroll
            for (int r = 0; r < SWEEP_RG; r++) {
                uint32_t i = i0 + r < cnt ? i0 + r : cnt - 1;
                uint32_t id = __builtin_amdgcn_readlane(my_id, i);
                load_row<(D > 0 ? D : 4)>(X + (size_t)id * D, lane, v[r]);
            }
#pragma unroll
            for (int r = 0; r < SWEEP_RG; r++) {
                uint32_t i = i0 + r;
                if (i < cnt) {
                    uint32_t bq = __builtin_amdgcn_readlane(my_b, i);
                    if (bq != cur_b) {
                        cur_b = bq;
                        load_row<(D > 0 ? D : 4)>(Q + (size_t)bq * D, lane, q);
                        if (COSINE) cur_qq = QQ[bq];
                    }
                    float s0, s1;
                    row_sums<(D > 0 ? D : 4), COSINE>(v[r], q, lane, s0, s1);
                    if (lane == i) { mine0 = s0; mine1 = s1; mine_qq = cur_qq; }
                }
            }
        }
    } else {
        for (uint32_t i = 0; i < cnt; i++) {
            uint32_t id = __builtin_amdgcn_readlane(my_id, i);
            uint32_t bq = __builtin_amdgcn_readlane(my_b, i);
            float ab, a2, l2;
            lane_sums_generic(X + (size_t)id * d, Q + (size_t)bq * d, d, lane, COSINE, ab, a2, l2);
            if (lane == i) { mine0 = COSINE ? ab : l2; mine1 = a2; mine_qq = COSINE ? QQ[bq] : 0.f; }
        }
    }
    if (lane < cnt) keys[r0 + lane] = COSINE ? key_cosine(mine0, mine1, mine_qq, mode) : key_l2(mine0, metric);
}

template <int D>
static hipError_t launch_sweep_d(const float *dX, uint32_t d, const float *dQ, const float *dQQ,
                                 const ZhVisit *dVisits, const uint64_t *dVisitRowOff, uint64_t n_visits,
                                 const uint32_t *dLeafIds, uint64_t R_total, int metric, int mode, uint64_t *dKeys,
                                 hipStream_t s) {
    uint64_t waves = (R_total + 63) / 64;
    uint64_t blocks = (waves + 3) / 4;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (metric == ZH_COSINE)
        hipLaunchKernelGGL((sweep_kernel<D, true>), dim3((uint32_t)blocks), dim3(256), 0, s, dX, d, dQ, dQQ, dVisits,
                           dVisitRowOff, n_visits, dLeafIds, R_total, metric, mode, dKeys);
    else
        hipLaunchKernelGGL((sweep_kernel<D, false>), dim3((uint32_t)blocks), dim3(256), 0, s, dX, d, dQ, dQQ, dVisits,
                           dVisitRowOff, n_visits, dLeafIds, R_total, metric, mode, dKeys);
    return hipGetLastError();
}

hipError_t zh_launch_sweep(const float *dX, uint32_t d, const float *dQ, const float *dQQ, const ZhVisit *dVisits,
                           const uint64_t *dVisitRowOff, uint64_t n_visits, const uint32_t *dLeafIds,
                           uint64_t R_total, int metric, int mode, uint64_t *dKeys, hipStream_t s) {
    if (R_total == 0 || n_visits == 0) return hipSuccess;
#define ZH_SWEEP_CASE(DD) \
    case DD: return launch_sweep_d<DD>(dX, d, dQ, dQQ, dVisits, dVisitRowOff, n_visits, dLeafIds, R_total, metric, mode, dKeys, s)
    switch (d) {
        ZH_SWEEP_CASE(64);
        ZH_SWEEP_CASE(128);
        ZH_SWEEP_CASE(256);
        ZH_SWEEP_CASE(384);
        ZH_SWEEP_CASE(512);
        ZH_SWEEP_CASE(768);
        ZH_SWEEP_CASE(1024);
        ZH_SWEEP_CASE(1536);
    default: return launch_sweep_d<0>(dX, d, dQ, dQQ, dVisits, dVisitRowOff, n_visits, dLeafIds, R_total, metric, mode, dKeys, s);
    }
#undef ZH_SWEEP_CASE
}

// n contiguous rows against one query: a single synthetic visit, ids = row numbers
__global__ void one_visit_kernel(ZhVisit *v, uint64_t *rowoff, uint64_t n) {
    v->b = 0; v->leaf_off = 0; v->len = (uint32_t)n; v->take = 0; v->row_off = 0; v->cand_off = 0;
    rowoff[0] = 0;
}
hipError_t zh_launch_distance_rows(const float *dX, uint64_t n, uint32_t d, const float *dq, int metric, int mode,
                                   uint64_t *dKeys, hipStream_t s) {
    if (!n) return hipSuccess;
    ZhVisit *dv = nullptr;
    uint64_t *dro = nullptr;
    float *dqq = nullptr;
    hipError_t e;
    if ((e = hipMalloc(&dv, sizeof(ZhVisit))) != hipSuccess) return e;
    if ((e = hipMalloc(&dro, 8)) != hipSuccess) { hipFree(dv); return e; }
    if ((e = hipMalloc(&dqq, 4)) != hipSuccess) { hipFree(dv); hipFree(dro); return e; }
    hipLaunchKernelGGL(one_visit_kernel, dim3(1), dim3(1), 0, s, dv, dro, n);
    e = zh_launch_qnorm(dq, 1, d, dqq, s);
    if (e == hipSuccess) e = zh_launch_sweep(dX, d, dq, dqq, dv, dro, 1, nullptr, n, metric, mode, dKeys, s);
    hipError_t e2 = hipStreamSynchronize(s);
    hipFree(dv); hipFree(dro); hipFree(dqq);
    return e != hipSuccess ? e : e2;
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

// per visit: the `take` smallest (key, id) of the leaf (lsh.rs:317-323); take == len copies all
__global__ __launch_bounds__(256) void select_kernel(const ZhVisit *__restrict__ visits,
                                                      const uint32_t *__restrict__ leaf_ids,
                                                      const uint64_t *__restrict__ keys,
                                                      uint64_t *__restrict__ cand_keys,
                                                      uint32_t *__restrict__ cand_ids) {
    __shared__ uint64_t sk[ZH_SORT_N];
    __shared__ uint32_t si[ZH_SORT_N];
    const ZhVisit v = visits[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    if (v.take == 0) return;
    if (v.take >= v.len) {
        for (uint32_t i = tid; i < v.len; i += 256) {
            cand_keys[v.cand_off + i] = keys[v.row_off + i];
            cand_ids[v.cand_off + i] = leaf_ids[(size_t)v.leaf_off + i];
        }
        return;
    }
    uint32_t have = 0, pos = 0;
    while (pos < v.len) {
        uint32_t m = ZH_SORT_N - have;
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

hipError_t zh_launch_select(const ZhVisit *dVisits, uint64_t n_visits, const uint32_t *dLeafIds,
                            const uint64_t *dKeys, uint64_t *dCandKeys, uint32_t *dCandIds, hipStream_t s) {
    if (!n_visits) return hipSuccess;
    if (n_visits > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(select_kernel, dim3((uint32_t)n_visits), dim3(256), 0, s, dVisits, dLeafIds, dKeys, dCandKeys,
                       dCandIds);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// final / merge: per query, the k smallest DISTINCT (key, id) of a candidate stream
// ------------------------------------------------------------------------------------------------
#define FIN_SORT_N 2048

// keeps the first k distinct ids of the sorted buffer sk/si[0..total) in rk/ri; returns their count
__device__ __forceinline__ uint32_t block_unique_topk(const uint64_t *sk, const uint64_t *si, uint32_t total,
                                                      uint32_t k, uint64_t *rk, uint64_t *ri, uint32_t *scan) {
    const uint32_t tid = threadIdx.x;  // 256 threads, FIN_SORT_N / 256 = 8 entries each
    constexpr uint32_t PER = FIN_SORT_N / 256;
    uint32_t base = tid * PER, cntl = 0;
    for (uint32_t j = 0; j < PER; j++) {
        uint32_t i = base + j;
        if (i < total && (i == 0 || si[i] != si[i - 1])) cntl++;
    }
    scan[tid] = cntl;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        uint32_t a = tid >= off ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += a;
        __syncthreads();
    }
    uint32_t rank = scan[tid] - cntl, uniq = scan[255];
    for (uint32_t j = 0; j < PER; j++) {
        uint32_t i = base + j;
        if (i < total && (i == 0 || si[i] != si[i - 1])) {
            if (rank < k) { rk[rank] = sk[i]; ri[rank] = si[i]; }
            rank++;
        }
    }
    __syncthreads();
    return uniq < k ? uniq : k;
}

// MERGE == false: candidates of query b are cand[cbase[b*T] .. cbase[(b+1)*T]) (u32 local ids)
// MERGE == true : S shard lists of k entries (u64 global ids) with counts
template <bool MERGE>
__global__ __launch_bounds__(256) void final_kernel(const uint64_t *__restrict__ candBase, uint32_t B, uint32_t T,
                                                     uint32_t k, const uint64_t *__restrict__ cand_keys,
                                                     const uint32_t *__restrict__ cand_ids32,
                                                     const uint64_t *__restrict__ cand_ids64,
                                                     const uint32_t *__restrict__ shard_counts, uint64_t id_base,
                                                     uint64_t *__restrict__ out_ids, uint64_t *__restrict__ out_keys,
                                                     uint32_t *__restrict__ out_counts) {
    __shared__ uint64_t sk[FIN_SORT_N];
    __shared__ uint64_t si[FIN_SORT_N];
    __shared__ uint64_t rk[ZH_MAX_TOPK];
    __shared__ uint64_t ri[ZH_MAX_TOPK];
    __shared__ uint32_t scan[256];
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    uint64_t n_src;  // entries in the source stream (MERGE: S*k slots, some invalid)
    uint64_t c0 = 0;
    if (MERGE) n_src = (uint64_t)T * k;  // T carries the shard count here
    else { c0 = candBase[(uint64_t)b * T]; n_src = candBase[(uint64_t)(b + 1) * T] - c0; }
    uint32_t have = 0;
    uint64_t pos = 0;
    bool first = true;
    while (pos < n_src || first) {
        first = false;
        uint32_t m = FIN_SORT_N - have;
        if ((uint64_t)m > n_src - pos) m = (uint32_t)(n_src - pos);
        for (uint32_t i = tid; i < have; i += 256) { sk[i] = rk[i]; si[i] = ri[i]; }
        for (uint32_t i = tid; i < m; i += 256) {
            uint64_t e = pos + i, key, id;
            if (MERGE) {
                uint32_t s = (uint32_t)(e / k), j = (uint32_t)(e % k);
                bool valid = j < shard_counts[(size_t)s * B + b];
                size_t src = ((size_t)s * B + b) * k + j;
                key = valid ? cand_keys[src] : ~0ull;
                id = valid ? cand_ids64[src] : ~0ull;
            } else {
                key = cand_keys[c0 + e];
                id = id_base + cand_ids32[c0 + e];
            }
            sk[have + i] = key; si[have + i] = id;
        }
        uint32_t total = have + m, np2 = next_pow2(total);
        for (uint32_t i = total + tid; i < np2; i += 256) { sk[i] = ~0ull; si[i] = ~0ull; }
        block_bitonic_sort<uint64_t>(sk, si, np2);
        if (MERGE) {  // invalid slots sorted last: drop them from the count
            uint32_t c = 0;
            for (uint32_t i = tid; i < total; i += 256) c += (sk[i] == ~0ull && si[i] == ~0ull) ? 0u : 1u;
            scan[tid] = c;
            __syncthreads();
            if (tid == 0) { uint32_t t2 = 0; for (int i = 0; i < 256; i++) t2 += scan[i]; scan[0] = t2; }
            __syncthreads();
            total = scan[0];
            __syncthreads();
        }
        have = block_unique_topk(sk, si, total, k, rk, ri, scan);
        pos += m;
    }
    for (uint32_t i = tid; i < k; i += 256) {
        out_ids[(size_t)b * k + i] = i < have ? ri[i] : ~0ull;
        out_keys[(size_t)b * k + i] = i < have ? rk[i] : ~0ull;
    }
    if (tid == 0) out_counts[b] = have;
}

hipError_t zh_launch_final(const uint64_t *dCandBase, uint32_t B, uint32_t T, uint32_t k, const uint64_t *dCandKeys,
                           const uint32_t *dCandIds, uint64_t id_base, uint64_t *dOutIds, uint64_t *dOutKeys,
                           uint32_t *dOutCounts, hipStream_t s) {
    if (!B) return hipSuccess;
    hipLaunchKernelGGL(final_kernel<false>, dim3(B), dim3(256), 0, s, dCandBase, B, T, k, dCandKeys, dCandIds,
                       (const uint64_t *)nullptr, (const uint32_t *)nullptr, id_base, dOutIds, dOutKeys, dOutCounts);
    return hipGetLastError();
}

hipError_t zh_launch_merge(uint32_t S, uint32_t B, uint32_t k, const uint64_t *dIds, const uint64_t *dKeys,
                           const uint32_t *dCounts, uint64_t *dOutIds, uint64_t *dOutKeys, uint32_t *dOutCounts,
                           hipStream_t s) {
    if (!B) return hipSuccess;
    hipLaunchKernelGGL(final_kernel<true>, dim3(B), dim3(256), 0, s, (const uint64_t *)nullptr, B, S, k, dKeys,
                       (const uint32_t *)nullptr, dIds, dCounts, (uint64_t)0, dOutIds, dOutKeys, dOutCounts);
    return hipGetLastError();
}
