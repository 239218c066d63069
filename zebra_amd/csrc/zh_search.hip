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
#include <cstdlib>

#include "zh_internal.h"
#include "zh_device.h"

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
    // tile (k0): thread t owns two float4 of the Q tile and two of the W tile: rows (t + 256*it) >> 3, k offset 4*((t + 256*it) & 7)
    auto fetch = [&](uint32_t k0, float4 *qv, float4 *wv) {
#pragma unroll
        for (int it = 0; it < 2; it++) {
            uint32_t i = tid + it * 256;
            uint32_t row = i >> 3, k = k0 + (i & 7) * 4;
            qv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            wv[it] = qv[it];
            if (vec4) {
                if (k < d) {
                    if (b0 + row < B) qv[it] = *reinterpret_cast<const float4 *>(Q + (size_t)(b0 + row) * d + k);
                    if (p0 + row < P) wv[it] = *reinterpret_cast<const float4 *>(W + (size_t)(p0 + row) * d + k);
                }
            } else {
                float t[4] = {0, 0, 0, 0}, u[4] = {0, 0, 0, 0};
                for (int e = 0; e < 4; e++)
                    if (k + e < d) {
                        if (b0 + row < B) t[e] = Q[(size_t)(b0 + row) * d + k + e];
                        if (p0 + row < P) u[e] = W[(size_t)(p0 + row) * d + k + e];
                    }
                qv[it] = make_float4(t[0], t[1], t[2], t[3]);
                wv[it] = make_float4(u[0], u[1], u[2], u[3]);
            }
        }
    };
    float4 qn[2], wn[2];
    fetch(0, qn, wn);
    for (uint32_t k0 = 0; k0 < d; k0 += HD_KT) {
#pragma unroll
        for (int it = 0; it < 2; it++) {
            uint32_t i = tid + it * 256;
            uint32_t row = i >> 3, c4 = (i & 7) * 4;
            Qs[c4 + 0][row] = qn[it].x; Qs[c4 + 1][row] = qn[it].y; Qs[c4 + 2][row] = qn[it].z; Qs[c4 + 3][row] = qn[it].w;
            Ws[c4 + 0][row] = wn[it].x; Ws[c4 + 1][row] = wn[it].y; Ws[c4 + 2][row] = wn[it].z; Ws[c4 + 3][row] = wn[it].w;
        }
        __syncthreads();
        if (k0 + HD_KT < d) fetch(k0 + HD_KT, qn, wn);  // next tile's loads fly during this tile's MFMAs
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

// The same contraction for GEMM-sized launches (every plane of a small-leaf forest: millions of planes x hundreds of
// queries): a 128 x 128 tile per block, each of the four waves owns 64 x 64 of it as 2 x 2 MFMA accumulators, so one
// k-step of four ds_read_b32 feeds four MFMAs (the 64 x 64 kernel: two reads per MFMA) and every plane row crosses
// L2 once per 128 queries.  Same per-output arithmetic: the k-ascending chain of v_mfma_f32_32x32x2_f32, no split-K.
#define HB_KT 16
#define HB_PITCH 129
template <bool WRITE_DOTS>
__global__ __launch_bounds__(256) void hash_dense_big_kernel(const float *__restrict__ Q, uint32_t B,
                                                              const float *__restrict__ W,
                                                              const float *__restrict__ C, uint32_t P, uint32_t d,
                                                              uint32_t *__restrict__ bits, uint32_t wpq,
                                                              float *__restrict__ dots) {
    __shared__ float Qs[HB_KT][HB_PITCH];
    __shared__ float Ws[HB_KT][HB_PITCH];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t wr = wv >> 1, wc = wv & 1;
    const uint32_t p0 = blockIdx.x * 128, b0 = blockIdx.y * 128;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
    // K-tile: 128 rows x 16 k of Q and of W = 512 float4 each; thread t owns float4 number t and t + 256:
    // row = i >> 2, k offset = 4 * (i & 3)   (requires d % 4 == 0: the launcher falls back to the small kernel otherwise)
    auto fetch = [&](uint32_t k0, float4 *qv, float4 *wvv) {
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const uint32_t i = tid + it * 256, row = i >> 2, k = k0 + (i & 3) * 4;
            qv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            wvv[it] = qv[it];
            if (k < d) {
                if (b0 + row < B) qv[it] = *reinterpret_cast<const float4 *>(Q + (size_t)(b0 + row) * d + k);
                if (p0 + row < P) wvv[it] = *reinterpret_cast<const float4 *>(W + (size_t)(p0 + row) * d + k);
            }
        }
    };
    float4 qn[2], wn[2];
    fetch(0, qn, wn);
    const uint32_t r = lane & 31, h = lane >> 5;
    for (uint32_t k0 = 0; k0 < d; k0 += HB_KT) {
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const uint32_t i = tid + it * 256, row = i >> 2, c4 = (i & 3) * 4;
            Qs[c4 + 0][row] = qn[it].x; Qs[c4 + 1][row] = qn[it].y; Qs[c4 + 2][row] = qn[it].z; Qs[c4 + 3][row] = qn[it].w;
            Ws[c4 + 0][row] = wn[it].x; Ws[c4 + 1][row] = wn[it].y; Ws[c4 + 2][row] = wn[it].z; Ws[c4 + 3][row] = wn[it].w;
        }
        __syncthreads();
        if (k0 + HB_KT < d) fetch(k0 + HB_KT, qn, wn);  // next tile's loads fly during this tile's MFMAs
#pragma unroll
        for (int kk = 0; kk < HB_KT; kk += 2) {
            const float a0 = Qs[kk + h][wr * 64 + r], a1 = Qs[kk + h][wr * 64 + 32 + r];
            const float c0 = Ws[kk + h][wc * 64 + r], c1 = Ws[kk + h][wc * 64 + 32 + r];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, c0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, c1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, c0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, c1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint32_t pbase = p0 + wc * 64 + j * 32, plane = pbase + (lane & 31);
            const double cval = plane < P ? (double)C[plane] : 0.0;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const uint32_t qrow_lo = b0 + wr * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2);
                const uint32_t qrow = qrow_lo + 4 * (lane >> 5);
                const float sdot = acc[i][j][reg];
                const bool above = ((double)sdot + cval) >= 0.0;  // lsh.rs:40-42, NaN -> false
                const unsigned long long m = __ballot(above);
                const uint32_t word = pbase >> 5;
                if (pbase < P) {  // a word of planes beyond P does not exist in the bit rows
                    if (lane == 0 && qrow_lo < B) bits[(size_t)qrow_lo * wpq + word] = (uint32_t)m;
                    if (lane == 32 && qrow_lo + 4 < B) bits[(size_t)(qrow_lo + 4) * wpq + word] = (uint32_t)(m >> 32);
                }
                if (WRITE_DOTS && qrow < B && plane < P) dots[(size_t)qrow * P + plane] = sdot;
            }
        }
}

hipError_t zh_launch_hash_dense(const float *dQ, uint32_t B, const float *dPlanes, const float *dConsts,
                                uint32_t P, uint32_t d, uint32_t *dBits, uint32_t words_per_q, float *dDots,
                                hipStream_t s) {
    if (B == 0 || P == 0) return hipSuccess;
    static const int variant = [] { const char *e = getenv("ZH_HASH_VARIANT"); return e ? atoi(e) : 0; }();  // 1: small tiles only
    if ((d & 3u) == 0 && B >= 128 && (uint64_t)P * B >= (1ull << 22) && variant != 1) {
        dim3 grid((P + 127) / 128, (B + 127) / 128);
        if (dDots)
            hipLaunchKernelGGL(hash_dense_big_kernel<true>, grid, dim3(256), 0, s, dQ, B, dPlanes, dConsts, P, d, dBits,
                               words_per_q, dDots);
        else
            hipLaunchKernelGGL(hash_dense_big_kernel<false>, grid, dim3(256), 0, s, dQ, B, dPlanes, dConsts, P, d, dBits,
                               words_per_q, dDots);
        return hipGetLastError();
    }
    dim3 grid((P + 63) / 64, (B + 63) / 64);
    if (dDots)
        hipLaunchKernelGGL(hash_dense_kernel<true>, grid, dim3(256), 0, s, dQ, B, dPlanes, dConsts, P, d, dBits,
                           words_per_q, dDots);
    else
        hipLaunchKernelGGL(hash_dense_kernel<false>, grid, dim3(256), 0, s, dQ, B, dPlanes, dConsts, P, d, dBits,
                           words_per_q, dDots);
    return hipGetLastError();
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
#define WALK_STACK 64
#define WALK_BUF 16

// emit pass: the s-th visit of a leaf joins group s / GRP of that leaf as member s % GRP
__device__ __forceinline__ void join_group(const uint32_t GRP, const ZhVisit &v, const uint32_t *__restrict__ leafCount,
                                           uint32_t *__restrict__ leafFill, const uint32_t *__restrict__ groupBase,
                                           const uint64_t *__restrict__ groupRowBase, ZhGroup *__restrict__ groups,
                                           uint64_t *__restrict__ groupRowOff) {
    const uint32_t s = atomicAdd(&leafFill[v.node], 1u);
    const uint32_t c = leafCount[v.node];
    const uint32_t gl = s / GRP, slot = s % GRP;
    const uint32_t g = groupBase[v.node] + gl;
    const uint32_t rest = c - gl * GRP;
    ZhGroup *G = groups + g;
    G->b[slot] = v.b;
    G->key_off[slot] = v.row_off;
    // what the member visit takes of the leaf (lsh.rs:300-330), capped at 255, a byte per slot (each slot writes its own: the slots of a group
    // arrive in any order): the fused half-width sweep (zh_approx.hip, round 6) tells from it which visits rank top_k rows of a longer leaf
    reinterpret_cast<uint8_t *>(&G->take4)[slot] = (uint8_t)(v.take < 255u ? v.take : 255u);
    if (slot == 0) {
        G->leaf_off = v.leaf_off; G->len = v.len; G->gsize = rest < GRP ? rest : GRP;
        groupRowOff[g] = groupRowBase[v.node] + (uint64_t)gl * v.len;
    }
}

// Four (query, tree) pairs per wave, one per 16-lane ROW; a row's 16 lanes carry identical copies of the pair's DFS
// state.  A plane below the dense levels is hashed on demand inside the row: lane lr holds the contiguous elements
// [lr*E, lr*E+E) of a 16*E-element pass of the plane and of the query, runs its E ordered fmas on the running sum and
// hands the sum to lane lr+1 with DPP row_newbcast -- 16 hops per pass, exactly the k-ascending chain, no LDS, and
// the four rows' chains share every instruction (a wave-wide chain would spend the same instructions on ONE pair).
#define WALK_E 24                      // elements per lane per pass (16 * 24 = 384 elements per pass)
#define WALK_PASS (16 * WALK_E)

template <int HOP>
__device__ __forceinline__ float row_bcast(float a) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x150 + HOP, 0xF, 0xF, false));
}

// this lane's E contiguous elements of src[k0 + lr*E ..), zero beyond d.  Branch-free: an out-of-range element is
// read from a clamped (valid) address and replaced by zero; `full` (d a multiple of the pass) skips the selects.
template <int E>
__device__ __forceinline__ void load_row_block(const float *__restrict__ src, uint32_t k0, uint32_t d, uint32_t lr, bool vec4,
                                               bool full, float *r) {
    const uint32_t base = k0 + lr * E;
    if (vec4) {  // d % 4 == 0: a float4 lies entirely inside or entirely outside the row
        if (full) {
#pragma unroll
            for (int j4 = 0; j4 < E / 4; j4++) {
                const float4 v = *reinterpret_cast<const float4 *>(src + base + 4 * j4);
                r[4 * j4] = v.x; r[4 * j4 + 1] = v.y; r[4 * j4 + 2] = v.z; r[4 * j4 + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int j4 = 0; j4 < E / 4; j4++) {
                const uint32_t e = base + 4 * j4;
                const bool in = e < d;
                const float4 v = *reinterpret_cast<const float4 *>(src + (in ? e : 0u));
                r[4 * j4] = in ? v.x : 0.0f; r[4 * j4 + 1] = in ? v.y : 0.0f;
                r[4 * j4 + 2] = in ? v.z : 0.0f; r[4 * j4 + 3] = in ? v.w : 0.0f;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < E; j++) {
            const uint32_t e = base + j;
            const bool in = e < d;
            const float v = src[in ? e : 0u];
            r[j] = in ? v : 0.0f;
        }
    }
}

// one pass of the chain: the running sum visits the row's lanes 0..15 in order (padded elements are 0*0: a + 0 = a)
template <int E>
__device__ __forceinline__ float chain_row_pass(const float *w, const float *q, float acc) {
#define ZH_HOP(H)                                                          \
    {                                                                      \
        float a = acc;                                                     \
        _Pragma("unroll") for (int j = 0; j < E; j++) a = __builtin_fmaf(w[j], q[j], a); \
        acc = row_bcast<H>(a);                                             \
    }
    ZH_HOP(0) ZH_HOP(1) ZH_HOP(2) ZH_HOP(3) ZH_HOP(4) ZH_HOP(5) ZH_HOP(6) ZH_HOP(7)
    ZH_HOP(8) ZH_HOP(9) ZH_HOP(10) ZH_HOP(11) ZH_HOP(12) ZH_HOP(13) ZH_HOP(14) ZH_HOP(15)
#undef ZH_HOP
    return acc;
}

// component-wise select (a struct-valued ?: goes through scratch memory)
__device__ __forceinline__ int4 sel4(bool c, const int4 &a, const int4 &b) {
    return make_int4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}

// DFS state of a row: the current node's RECORD travels with its id (the records of both children are fetched as
// soon as an internal node's record is known -- they arrive while the sign is worked out -- and the backup child's
// record waits on the LDS stack), so a step's only exposed miss is the plane row of an on-demand hash.
// E = elements per lane per pass of an on-demand chain; NP = passes whose query block stays in registers for the whole
// walk (d <= NP * 16 * E), 0 = the query block is re-read with every plane (very long vectors)
// ALLDENSE: every sign is precomputed -- one pair per wave, no chains; the DFS state is wave-uniform and is kept in
// scalar registers (readfirstlane after every load), so the loop runs on scalar branches instead of exec masks
__device__ __forceinline__ int4 uni4(int4 v) {
    return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y),
                     __builtin_amdgcn_readfirstlane(v.z), __builtin_amdgcn_readfirstlane(v.w));
}
template <bool EMIT, int E, int NP, bool ALLDENSE>
__global__ __launch_bounds__(64) void walk_kernel(ZhForestDev f, const float *__restrict__ Q, uint32_t B, uint32_t d,
                                                   int32_t n, const uint32_t *__restrict__ bits, uint32_t wpq,
                                                   uint32_t P_dense, ZhPairCounts *__restrict__ counts,
                                                   ZhVisit *__restrict__ inl, const uint64_t *__restrict__ rowBase,
                                                   const uint64_t *__restrict__ candBase,
                                                   const uint64_t *__restrict__ visitBase,
                                                   ZhVisit *__restrict__ visits, uint32_t *__restrict__ leafCount,
                                                   uint32_t *__restrict__ leafFill,
                                                   const uint32_t *__restrict__ groupBase,
                                                   const uint64_t *__restrict__ groupRowBase,
                                                   ZhGroup *__restrict__ groups, uint64_t *__restrict__ groupRowOff,
                                                   ZhWalkLog wlog) {
    constexpr uint32_t ppw = ALLDENSE ? 1 : 4;
#define ZH_U(x) (ALLDENSE ? __builtin_amdgcn_readfirstlane(x) : (x))
#define ZH_U4(x) (ALLDENSE ? uni4(x) : (x))
    __shared__ int4 st_rec[4][WALK_STACK];
    __shared__ int2 st_nn[4][WALK_STACK];  // {node, n}
    // a row's latest visits {node, leaf offset, leaf length, take} and their row / candidate offsets, written out
    // WALK_BUF at a time
    __shared__ uint4 vb_a[4][WALK_BUF];
    __shared__ uint64_t vb_r[4][WALK_BUF], vb_c[4][WALK_BUF];
    // ppw = pairs per wave: 4 (one per 16-lane row) normally; 1 when every plane is hashed densely -- no chains to
    // share then, and one pair per wave keeps the leaf-heavy DFS of small-leaf forests free of row divergence
    const uint32_t T = f.n_trees, lane = threadIdx.x, row = ppw == 4 ? lane >> 4 : 0, lr = ppw == 4 ? (lane & 15) : lane;
    const uint64_t n_pairs = (uint64_t)B * T;
    const uint64_t pair = (uint64_t)blockIdx.x * ppw + row;  // every lane of a row works on the same pair
    bool active = pair < n_pairs;
    const uint32_t b = active ? (uint32_t)(pair / T) : 0, t = active ? (uint32_t)(pair % T) : 0;
    if (EMIT && active) {
        uint32_t nv0 = counts[pair].visits;
        if (nv0 <= ZH_INLINE_VISITS) {  // the first pass recorded every visit: place them, one lane each
            for (uint32_t i = lr; i < nv0; i += (ppw == 4 ? 16 : 64)) {
                ZhVisit v = inl[pair * ZH_INLINE_VISITS + i];
                v.row_off += rowBase[pair];
                v.cand_off += candBase[pair];
                visits[visitBase[pair] + i] = v;
                join_group(f.group, v, leafCount, leafFill, groupBase, groupRowBase, groups, groupRowOff);
            }
            active = false;
        }
    }
    const float *q = Q + (size_t)b * d;
    const bool vec4 = (d & 3u) == 0, full = (d % (16u * E)) == 0;
    const uint32_t l16 = lane & 15;
    float qreg[NP > 0 ? NP : 1][E];
    if (NP > 0 && P_dense < f.n_planes) {
#pragma unroll
        for (int p = 0; p < NP; p++)
            if (p == 0 || (uint32_t)p * 16u * E < d) load_row_block<E>(q, (uint32_t)p * 16u * E, d, l16, vec4, full, qreg[p]);
    }
    int sp = 0;
    int32_t cur = ZH_U(active ? (int32_t)f.roots[t] : 0), ncur = n;
    int4 rec = ZH_U4(active ? f.node_pack[cur] : make_int4(-1, 0, 0, 0));
    uint32_t nv = 0;
    uint64_t nrows = 0, ntakes = 0;
    uint64_t vb = 0, rb = 0, cb = 0;
    if (EMIT && active) { vb = visitBase[pair]; rb = rowBase[pair]; cb = candBase[pair]; }
    uint32_t nbuf = 0;
    uint32_t log_chunk = 0xFFFFFFFFu;  // the log chunk being filled and its number in the pair's chain (-1: none yet)
    int32_t log_cn = -1;
    bool log_ok = true;
    // Stores and atomics share the loads' in-order counter (vmcnt): one issued per visit would stall the very next
    // record load for a full write round trip.  Visits therefore collect in LDS and leave WALK_BUF at a time, one
    // lane each: leafCount atomics, inline visits, log entries (count pass) / placed visits and group joins (emit pass).
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t first = nv - nbuf, vi = first + lr;
        const bool mine = lr < nbuf;
        const uint4 va = vb_a[row][mine ? lr : 0];
        ZhVisit v;
        v.b = b; v.leaf_off = va.y; v.len = va.z; v.take = va.w; v.node = va.x; v.pad = 0;
        v.row_off = vb_r[row][mine ? lr : 0]; v.cand_off = vb_c[row][mine ? lr : 0];
        if (EMIT) {
            if (mine) {
                visits[vb + vi] = v;
                join_group(f.group, v, leafCount, leafFill, groupBase, groupRowBase, groups, groupRowOff);
            }
        } else {
            if (mine) {
                if (leafCount) atomicAdd(&leafCount[v.node], 1u);  // (a prefiltered batch forms no leaf groups)
                if (vi < ZH_INLINE_VISITS) inl[pair * ZH_INLINE_VISITS + vi] = v;
            }
            if (nv > ZH_INLINE_VISITS && log_ok) {  // row-uniform: entries [max(first, 32), nv) go to the log
                constexpr uint32_t PER = ZH_LOG_CHUNK - 1;
                const int32_t c1 = (int32_t)((nv - 1 - ZH_INLINE_VISITS) / PER);  // chain number of the last entry's chunk
                uint32_t newc = 0xFFFFFFFFu;
                if (c1 > log_cn) {  // at most one new chunk per flush (WALK_BUF < PER)
                    uint32_t c = 0;
                    if (lr == 0) c = atomicAdd(&wlog.ctl->next_chunk, 1u);
                    c = __shfl(c, (int)(lane - lr));
                    if (c >= wlog.capacity) {
                        if (lr == 0) wlog.ctl->overflow = 1u;
                        log_ok = false;
                    } else {
                        newc = c;
                        if (lr == 0) {
                            if (log_cn < 0) wlog.head[pair] = c;
                            else wlog.pool[(size_t)log_chunk * ZH_LOG_CHUNK].x = c;
                        }
                    }
                }
                if (log_ok) {
                    if (mine && vi >= ZH_INLINE_VISITS) {
                        const uint32_t li = vi - ZH_INLINE_VISITS;
                        const uint32_t id = (int32_t)(li / PER) == log_cn ? log_chunk : newc;
                        wlog.pool[(size_t)id * ZH_LOG_CHUNK + 1 + li % PER] = wlog.leaf_entries ? make_uint2(v.leaf_off, v.take | (v.len << 16)) : make_uint2(v.node, v.take);
                    }
                    if (newc != 0xFFFFFFFFu) { log_chunk = newc; log_cn = c1; }
                }
            }
        }
        nbuf = 0;
        __builtin_amdgcn_wave_barrier();
    };
    bool need = false;   // this row waits for the sign of node `cur` (an on-demand plane)
#define ZH_DESCEND(ABOVE, RL, RR)                                                                        \
    {                                                                                                    \
        const bool ab_ = (ABOVE);                                                                        \
        if (sp < WALK_STACK) { st_rec[row][sp] = sel4(ab_, (RL), (RR)); st_nn[row][sp] = make_int2(ab_ ? rec.y : rec.z, ncur); } \
        sp++;                                                                                            \
        cur = ab_ ? rec.z : rec.y; /* lsh.rs:335-338: above -> right is main */                          \
        rec = sel4(ab_, (RR), (RL));                                                                     \
    }
    for (;;) {
        // ---- phase A: every row runs its DFS until it needs an on-demand sign or is finished ----
        if (active && !need) {
            for (;;) {
                if (rec.x >= 0) {
                    if (!ALLDENSE && (uint32_t)rec.x >= P_dense) { need = true; break; }
                    const int4 rl = ZH_U4(f.node_pack[rec.y]), rr = ZH_U4(f.node_pack[rec.z]);
                    const bool above = (ZH_U((int)bits[(size_t)b * wpq + ((uint32_t)rec.x >> 5)]) >> (rec.x & 31)) & 1;
                    ZH_DESCEND(above, rl, rr)
                    continue;
                }
                uint32_t off = (uint32_t)rec.y, len = (uint32_t)rec.z;
                uint32_t take = ncur <= 0 ? 0u : (len < (uint32_t)ncur ? len : (uint32_t)ncur);
                int32_t ret = (int32_t)take;  // lsh.rs:306 / 329
                if (take > 0) {
                    if (lr == 0) {
                        vb_a[row][nbuf] = make_uint4((uint32_t)cur, off, len, take);
                        vb_r[row][nbuf] = (EMIT ? rb : 0) + nrows;
                        vb_c[row][nbuf] = (EMIT ? cb : 0) + ntakes;
                    }
                    nbuf++; nv++; nrows += len; ntakes += take;
                    if (nbuf == WALK_BUF) flush();
                }
                bool down = false;
                while (sp > 0) {
                    sp--;
                    if (sp < WALK_STACK && ret < ZH_U(st_nn[row][sp].y)) {  // lsh.rs:341-343: k < n -> the backup's count alone
                        cur = ZH_U(st_nn[row][sp].x);
                        ncur = ZH_U(st_nn[row][sp].y) - ret;
                        rec = ZH_U4(st_rec[row][sp]);
                        down = true;
                        break;
                    }
                }
                if (!down) { active = false; break; }
            }
        }
        if (ALLDENSE || !__any(need)) break;  // no row waits for a sign: every row has finished
        // ---- phase B: the rows that need a sign hash their plane, four chains per wave ----
        int4 rl = make_int4(-1, 0, 0, 0), rr = rl;
        if (need) { rl = f.node_pack[rec.y]; rr = f.node_pack[rec.z]; }
        const float *w = f.planes + (size_t)(need ? rec.x : 0) * d;  // a row that needs no sign rides along on plane 0
        float acc = 0.0f;
        if (NP > 0) {
#pragma unroll
            for (int p = 0; p < NP; p++)
                if (p == 0 || (uint32_t)p * 16u * E < d) {  // wave-uniform
                    float wreg[E];
                    load_row_block<E>(w, (uint32_t)p * 16u * E, d, l16, vec4, full, wreg);
                    acc = chain_row_pass<E>(wreg, qreg[p], acc);
                }
        } else {
            for (uint32_t k0 = 0; k0 < d; k0 += 16 * E) {
                float wreg[E], qr[E];
                load_row_block<E>(w, k0, d, l16, vec4, full, wreg);
                load_row_block<E>(q, k0, d, l16, vec4, full, qr);
                acc = chain_row_pass<E>(wreg, qr, acc);
            }
        }
        if (need) {
            const bool above = ((double)acc + (double)__int_as_float(rec.w)) >= 0.0;  // lsh.rs:40-42
            ZH_DESCEND(above, rl, rr)
            need = false;
        }
    }
#undef ZH_DESCEND
#undef ZH_U
#undef ZH_U4
    if (nbuf) flush();
    if (!EMIT && pair < n_pairs && lr == 0) {
        ZhPairCounts c;
        c.visits = nv; c.rows = (uint32_t)nrows; c.takes = (uint32_t)ntakes; c.pad = 0;
        counts[pair] = c;
    }
}

template <bool EMIT>
static void launch_walk(ZhForestDev f, const float *dQ, uint32_t B, uint32_t d, int32_t n, const uint32_t *dBits,
                        uint32_t wpq, uint32_t P_dense, ZhPairCounts *dCounts, ZhVisit *dInline, const uint64_t *dRowBase,
                        const uint64_t *dCandBase, const uint64_t *dVisitBase, ZhVisit *dVisits, uint32_t *dLeafCount,
                        uint32_t *dLeafFill, const uint32_t *dGroupBase, const uint64_t *dGroupRowBase, ZhGroup *dGroups,
                        uint64_t *dGroupRowOff, ZhWalkLog log, hipStream_t s) {
    const uint64_t pairs = (uint64_t)B * f.n_trees;
    const uint32_t ppw = P_dense >= f.n_planes ? 1u : 4u;
    const dim3 grid((uint32_t)((pairs + ppw - 1) / ppw));
#define ZH_WALK_(E_, NP_, AD_)                                                                                              \
    hipLaunchKernelGGL((walk_kernel<EMIT, E_, NP_, AD_>), grid, dim3(64), 0, s, f, dQ, B, d, n, dBits, wpq, P_dense, dCounts, \
                       dInline, dRowBase, dCandBase, dVisitBase, dVisits, dLeafCount, dLeafFill, dGroupBase, dGroupRowBase, \
                       dGroups, dGroupRowOff, log)
#define ZH_WALK(E_, NP_) ZH_WALK_(E_, NP_, false)
    // registers: the query block kept in registers (NP > 0) costs occupancy -- worth it while every wave of the launch
    // is resident anyway (deep walks of few pairs); with more waves than that, or no chains at all, the lean variants
    const uint64_t waves = grid.x;
    if (P_dense >= f.n_planes) ZH_WALK_(4, 0, true);
    else if (waves > 2048) {
        if (d <= 64) ZH_WALK(4, 0);
        else if (d <= 128) ZH_WALK(8, 0);
        else if (d <= 256) ZH_WALK(16, 0);
        else ZH_WALK(WALK_E, 0);
    } else if (d <= 64) ZH_WALK(4, 1);
    else if (d <= 128) ZH_WALK(8, 1);
    else if (d <= 256) ZH_WALK(16, 1);
    else if (d <= 16 * WALK_E) ZH_WALK(WALK_E, 1);
    else if (d <= 32 * WALK_E) ZH_WALK(WALK_E, 2);
    else ZH_WALK(WALK_E, 0);
#undef ZH_WALK
#undef ZH_WALK_
}

hipError_t zh_launch_walk_count(ZhForestDev f, const float *dQ, uint32_t B, uint32_t d, int32_t n,
                                const uint32_t *dBits, uint32_t words_per_q, uint32_t P_dense, ZhPairCounts *dCounts,
                                ZhVisit *dInline, uint32_t *dLeafCount, ZhWalkLog log, hipStream_t s) {
    if (!((uint64_t)B * f.n_trees)) return hipSuccess;
    launch_walk<false>(f, dQ, B, d, n, dBits, words_per_q, P_dense, dCounts, dInline, nullptr, nullptr, nullptr, nullptr,
                       dLeafCount, nullptr, nullptr, nullptr, nullptr, nullptr, log, s);
    return hipGetLastError();
}
hipError_t zh_launch_walk_emit(ZhForestDev f, const float *dQ, uint32_t B, uint32_t d, int32_t n,
                               const uint32_t *dBits, uint32_t words_per_q, uint32_t P_dense,
                               const ZhPairCounts *dCounts, const ZhVisit *dInline, const uint64_t *dRowBase,
                               const uint64_t *dCandBase, const uint64_t *dVisitBase, ZhVisit *dVisits,
                               const uint32_t *dLeafCount, uint32_t *dLeafFill, const uint32_t *dGroupBase,
                               const uint64_t *dGroupRowBase, ZhGroup *dGroups, uint64_t *dGroupRowOff, hipStream_t s) {
    if (!((uint64_t)B * f.n_trees)) return hipSuccess;
    ZhWalkLog nolog{nullptr, 0, nullptr, nullptr, 0};
    launch_walk<true>(f, dQ, B, d, n, dBits, words_per_q, P_dense, const_cast<ZhPairCounts *>(dCounts),
                      const_cast<ZhVisit *>(dInline), dRowBase, dCandBase, dVisitBase, dVisits,
                      const_cast<uint32_t *>(dLeafCount), dLeafFill, dGroupBase, dGroupRowBase, dGroups, dGroupRowOff, nolog, s);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// walk over the BLOCKED forest (every sign precomputed): one (query, tree) pair per wave.  The pointer-based walk above
// pays a memory round trip (~1 us on a busy chip) for every node it steps on; the wandering walk of small-leaf forests
// (thousands of leaf visits per pair, ~3 new inner nodes per visit) is nothing but such steps.  Here the bottom
// of every tree is cut into blocks of <= 64 nodes: lane i loads record i of the block the walk enters (one coalesced
// 1-KiB load) and gathers that node's sign bit, and the DFS inside the block then runs on v_readlane / v_writelane with a
// wave-uniform cursor (the block's own stack is one VGPR, entry j in lane j) -- no memory access at all until the walk
// leaves the block.  Same control flow as tree_result
// (lsh.rs:290-348), same visit order, same logs as walk_kernel<false, ..., ALLDENSE>.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int rl(int v, uint32_t lane) { return __builtin_amdgcn_readlane(v, (int)lane); }

#ifdef ZH_WALK_PROF  // tests/probes/walk_prof.py: where a wave of the blocked walk spends its cycles (never defined in the shipped build)
__device__ uint64_t zh_walk_prof_buf[16 * 8192];
extern "C" __attribute__((visibility("default"))) int zh_debug_walk_prof(uint64_t *out, uint32_t words) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(zh_walk_prof_buf), (size_t)words * 8);
}
#define WP(...) __VA_ARGS__
#else
#define WP(...)
#endif

// a wave-uniform value the compiler must keep in a VGPR (it would otherwise compute it on the scalar unit: the DFS below is bound
// by the SIMD's one scalar issue slot per four cycles -- tests/probes/walk_prof.py -- while its vector slot idles)
__device__ __forceinline__ uint32_t in_vgpr(uint32_t x) {
    uint32_t r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(x));
    return r;
}
__device__ __forceinline__ uint32_t in_vgpr_shl8(uint32_t x) {
    uint32_t r;
    asm volatile("v_lshlrev_b32 %0, 8, %1" : "=v"(r) : "s"(x));
    return r;
}
// point_is_above for ONE plane by a whole wave (d % 4 == 0): the plane's and the query's rows are loaded 256 elements at a time,
// lane i holding float4 i of the pass -- one round trip per 256 elements instead of zh_plane_above's one per 32 -- and the
// k-ascending fma chain (the order contract of the hash) runs on v_readlane broadcasts.  Wave-uniform arguments and result.
__device__ __forceinline__ bool plane_above_wave(const float *__restrict__ w, float c, const float *__restrict__ x, uint32_t d,
                                                 uint32_t lane) {
    const float4 *w4 = reinterpret_cast<const float4 *>(w), *x4 = reinterpret_cast<const float4 *>(x);
    const uint32_t n4 = d / 4;
    float acc = 0.0f;
    for (uint32_t base = 0; base < n4; base += 64) {  // 256 elements per pass, eight registers: the walk's waves stay small enough
        float4 wr = make_float4(0.f, 0.f, 0.f, 0.f), xr = wr;  // for the hash of the next batch to find room beside them
        if (base + lane < n4) { wr = w4[base + lane]; xr = x4[base + lane]; }
        const uint32_t cnt = n4 - base < 64u ? n4 - base : 64u;
        for (uint32_t t = 0; t < cnt; t++) {
#define RLF(v_) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v_), (int)t))
            acc = __builtin_fmaf(RLF(wr.x), RLF(xr.x), acc);
            acc = __builtin_fmaf(RLF(wr.y), RLF(xr.y), acc);
            acc = __builtin_fmaf(RLF(wr.z), RLF(xr.z), acc);
            acc = __builtin_fmaf(RLF(wr.w), RLF(xr.w), acc);
#undef RLF
        }
    }
    // lsh.rs:40-42, as zh_plane_above; every lane ran the same chain: said so, or the compiler treats everything the walk derives from
    // the result (the cursor, the stack pointer) as divergent and moves it to VGPRs
    return __builtin_amdgcn_readfirstlane((int)(((double)acc + (double)c) >= 0.0)) != 0;
}

#define WALK_RING 128u       // visit records waiting to leave the wave: ring of 128 slots (+ one slot every lane without a visit writes to)
#define WALK_FLUSH 32u       // ... leave 32 at a time (< ZH_LOG_CHUNK - 1: at most one new log chunk per flush)
#define WALK_EXIT (0xC0000000u | 63u)

__global__ __launch_bounds__(64) void walk_blocked_kernel(ZhForestDev f, ZhBlocksDev blk, uint32_t B, int32_t n0,
                                                           const uint32_t *__restrict__ bits, uint32_t wpq,
                                                           ZhPairCounts *__restrict__ counts, ZhVisit *__restrict__ inl,
                                                           uint32_t *__restrict__ leafCount, ZhWalkLog wlog,
                                                           const uint32_t *__restrict__ unc, const float *__restrict__ Q, uint32_t d) {
    // unc != null: the row-score hash left its uncertain signs flagged instead of fixing them (zh_score.hip, "lazy"): a flagged sign
    // is recomputed here with point_is_above's own arithmetic (plane_above_wave) when -- and only when -- the walk steps on its node
    __shared__ int4 ust[WALK_STACK];  // upper-level stack {child ref, n, the child's plane when it is an upper node}
    __shared__ uint4 vb_a[WALK_RING + 1];
    __shared__ uint32_t vb_r[WALK_RING + 1], vb_c[WALK_RING + 1];
    const uint32_t T = f.n_trees, lane = threadIdx.x;
    const uint64_t pair = blockIdx.x;
    const uint32_t b = (uint32_t)(pair / T), t = (uint32_t)(pair % T);
    const uint32_t *__restrict__ qbits = bits + (size_t)b * wpq;
    const uint32_t *__restrict__ qunc = unc ? unc + (size_t)b * wpq : nullptr;
    if (n0 <= 0) {  // lsh.rs:306: the first leaf takes nothing and nothing is ever < 0: no visit
        if (lane == 0) { ZhPairCounts c; c.visits = 0; c.rows = 0; c.takes = 0; c.pad = 0; counts[pair] = c; }
        return;
    }
    // the pair's running totals: wave-uniform, kept in VGPRs (see in_vgpr)
    uint32_t v_nv = in_vgpr(0), v_nrows = in_vgpr(0), v_ntakes = in_vgpr(0);
    uint32_t flushed = 0;
    uint32_t log_chunk = 0xFFFFFFFFu;
    int32_t log_cn = -1;
    bool log_ok = true;
    WP(uint64_t p_t0 = clock64(); uint64_t p_w0 = wall_clock64(); uint64_t p_load = 0, p_upper = 0, p_flush = 0, p_dfs = 0;
       uint32_t p_blocks = 0, p_uppers = 0, p_inner = 0, p_pops = 0, p_upops = 0;)
    auto flush = [&](uint32_t cnt) {  // as walk_kernel's count-pass flush: the ring's oldest `cnt` (<= WALK_FLUSH) visits leave, one lane each
        WP(const uint64_t p_f0 = clock64();)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t first = flushed, vi = first + lane, upto = first + cnt;
        const bool mine = lane < cnt;
        const uint32_t slot = mine ? (vi & (WALK_RING - 1)) : WALK_RING;
        const uint4 va = vb_a[slot];
        ZhVisit v;
        v.b = b; v.leaf_off = va.y; v.len = va.z; v.take = va.w; v.node = va.x; v.pad = 0;
        v.row_off = vb_r[slot]; v.cand_off = vb_c[slot];
        if (mine) {
            if (leafCount) atomicAdd(&leafCount[v.node], 1u);  // (a prefiltered batch forms no leaf groups)
            if (vi < ZH_INLINE_VISITS) inl[pair * ZH_INLINE_VISITS + vi] = v;
        }
        if (upto > ZH_INLINE_VISITS && log_ok) {
            constexpr uint32_t PER = ZH_LOG_CHUNK - 1;
            static_assert(WALK_FLUSH < PER, "a flush may open at most one log chunk");
            const int32_t c1 = (int32_t)((upto - 1 - ZH_INLINE_VISITS) / PER);
            uint32_t newc = 0xFFFFFFFFu;
            if (c1 > log_cn) {
                uint32_t c = 0;
                if (lane == 0) c = atomicAdd(&wlog.ctl->next_chunk, 1u);
                c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                if (c >= wlog.capacity) {
                    if (lane == 0) wlog.ctl->overflow = 1u;
                    log_ok = false;
                } else {
                    newc = c;
                    if (lane == 0) {
                        if (log_cn < 0) wlog.head[pair] = c;
                        else wlog.pool[(size_t)log_chunk * ZH_LOG_CHUNK].x = c;
                    }
                }
            }
            if (log_ok) {
                if (mine && vi >= ZH_INLINE_VISITS) {
                    const uint32_t li = vi - ZH_INLINE_VISITS;
                    const uint32_t id = (int32_t)(li / PER) == log_cn ? log_chunk : newc;
                    wlog.pool[(size_t)id * ZH_LOG_CHUNK + 1 + li % PER] = wlog.leaf_entries ? make_uint2(v.leaf_off, v.take | (v.len << 16)) : make_uint2(v.node, v.take);
                }
                if (newc != 0xFFFFFFFFu) { log_chunk = newc; log_cn = c1; }
            }
        }
        flushed = upto;
        __builtin_amdgcn_wave_barrier();
        WP(p_flush += clock64() - p_f0;)
    };
    const int2 root = blk.root[t];
    int32_t ref = __builtin_amdgcn_readfirstlane(root.x), pl = __builtin_amdgcn_readfirstlane(root.y);
    int32_t n = n0;
    int usp = 0;
    for (;;) {
        if (ref >= 0) {  // an upper (inner) node: its records and its sign word are requested together
            WP(const uint64_t p_u0 = clock64(); p_uppers++;)
            // wave-uniform addresses: scalar loads (through the constant address space) -- the records arrive in SGPRs, no vector
            // memory instruction, no v_readfirstlane per word
            typedef const int32_t __attribute__((address_space(4))) *cptr_t;
            const cptr_t up = (cptr_t)(uintptr_t)(blk.upper + 2 * (size_t)ref);
            const int4 a = make_int4(up[0], up[1], up[2], up[3]), c2 = make_int4(up[4], up[5], up[6], up[7]);
            const uint32_t word = (uint32_t)((cptr_t)(uintptr_t)qbits)[(uint32_t)pl >> 5];
            const uint32_t uword = qunc ? (uint32_t)((cptr_t)(uintptr_t)qunc)[(uint32_t)pl >> 5] : 0u;
            bool above = (word >> (pl & 31)) & 1;
            if ((uword >> (pl & 31)) & 1)
                above = plane_above_wave(f.planes + (size_t)pl * d, f.consts[pl], Q + (size_t)b * d, d, lane);
            if (usp < WALK_STACK && lane == 0) ust[usp] = make_int4(above ? a.y : a.z, n, above ? c2.x : c2.y, 0);
            usp++;
            ref = above ? a.z : a.y;  // lsh.rs:335-338: above -> right is main
            pl = above ? c2.y : c2.x;
            WP(p_upper += clock64() - p_u0;)
            continue;
        }
        // ---- a block: records and signs into registers, then the DFS without memory ----
        WP(const uint64_t p_l0 = clock64(); p_blocks++;)
        const uint32_t s0 = (uint32_t)(-ref - 1);
        int4 r = blk.recs[s0 + lane];  // the array is padded: 64 records can always be read
        const int r0x = rl(r.x, 0);
        const uint32_t cnt = r0x >= 0 ? (uint32_t)rl(r.z, 0) : 1u;  // odd (a full binary subtree), so <= 63: lane 63 is never a node
        if (lane >= cnt) r = make_int4(-1, 0, 0, 0);
        int sgn = 0;
        uint32_t ub = 0;
        if (r.x >= 0) {
            sgn = (int)((qbits[(uint32_t)r.x >> 5] >> (r.x & 31)) & 1u);
            if (qunc) ub = (qunc[(uint32_t)r.x >> 5] >> (r.x & 31)) & 1u;
        }
        // One word per lane holds everything a step needs.  Inner node: main child | backup child << 8 (lsh.rs:335-338: above ->
        // right is main) -- the next step's v_readlane takes the word itself as its lane select (the hardware uses bits 5:0).  Leaf:
        // bit 31 | own lane -- unique in the block, so `pk == w` is true in the visited leaf's lane alone and that lane records the
        // visit from its own registers (`pkm`: the same word, or no word at all for an empty leaf, which is stepped on but not a
        // visit).  Lanes past the block read as empty leaves; lane 63 (WALK_EXIT) is where the stack's sentinel leads once the
        // block is exhausted.  A node whose sign is flagged looks like a leaf with bit 30 set: the descent stops on it, the sign is
        // recomputed and the descent goes on (WALK_DESCEND).
        uint32_t pk = lane == 63 ? WALK_EXIT : (0x80000000u | lane);
        const uint32_t pkm = r.x < 0 && r.z > 0 ? pk : 0xFFFFFFFFu;
        if (r.x >= 0) {
            const uint32_t l = (uint32_t)r.y & 0xFFFFu, rr = (uint32_t)r.y >> 16;
            pk = sgn ? (rr | (l << 8)) : (l | (rr << 8));
            if (ub) pk = 0xC0000000u | lane;
        }
        WP(if (__builtin_amdgcn_readfirstlane((int)pk) == 12345) p_blocks++; const uint64_t p_d0 = clock64(); p_load += p_d0 - p_l0;
           const uint64_t p_fl0 = p_flush;)
        // the block's DFS stack: lane j holds entry j = backup node | n << 8; entry 0 is a sentinel no return value ever exhausts
        uint32_t lstk = lane == 0 ? ((0x7FFFFFu << 8) | 63u) : 0u;
        uint32_t lsp = 1;
        uint32_t vidx = 0xFFFFFFFFu, vtk = 0, vrow = 0, vcand = 0;  // the visit of this lane's leaf, if any (a DFS meets a leaf once)
        uint32_t n8 = in_vgpr_shl8((uint32_t)n);
        int32_t ret = 0;
        uint32_t w = (uint32_t)rl((int)pk, 0);
        // down the main children to a leaf: 1 scalar + 5 vector instructions and the branch per inner node.  (Round 2's loop spent
        // 37 instructions here, most of them scalar copies of the walk's other live state.)
#define WALK_DESCEND()                                                                                                       \
        while (!(w >> 31)) {                                                                                                 \
            const uint32_t e_ = __builtin_amdgcn_perm(n8, w, 0x07060501u); /* byte 0 = the word's byte 1 (backup), bytes 1-3 = n */ \
            lstk = lane == lsp ? e_ : lstk;                                                                                  \
            lsp++;                                                                                                           \
            w = (uint32_t)rl((int)pk, w);                                                                                    \
            WP(p_inner++;)                                                                                                   \
        }
        WALK_DESCEND();
        for (;;) {  // (the outer loop turns only for a flagged sign: one node in ~400)
            if (w < 0xC0000000u) {
                do {  // one leaf per turn; WALK_EXIT and the flagged words are the only ones with bit 30: one compare per leaf
                    // the leaf: no branch -- an empty one (lsh.rs:306 / 329: it returns 0) matches no lane and adds nothing
                    const uint32_t len = (uint32_t)rl(r.z, w);
                    const uint32_t take = len < (uint32_t)n ? len : (uint32_t)n;  // n >= 1 here (n0 >= 1; a backup is entered with nn - ret > 0)
                    ret = (int32_t)take;
                    const bool me = pkm == w;
                    vidx = me ? v_nv : vidx; vtk = me ? take : vtk; vrow = me ? v_nrows : vrow; vcand = me ? v_ntakes : vcand;
                    v_nv += in_vgpr(len) != 0u ? 1u : 0u; v_nrows += len; v_ntakes += take;
                    uint32_t e;
                    do {  // lsh.rs:341-343: the nearest pending backup whose n the return value has not used up
                        lsp--;
                        WP(p_pops++;)
                        e = (uint32_t)rl((int)lstk, lsp);
                    } while ((uint32_t)ret >= (e >> 8));
                    n = (int32_t)((e >> 8) - (uint32_t)ret);
                    n8 = in_vgpr_shl8((uint32_t)n);
                    w = (uint32_t)rl((int)pk, e);
                    WALK_DESCEND();
                } while (w < 0xC0000000u);
            }
            if (w == WALK_EXIT) break;  // through the sentinel: the block is done and `ret` is its return value
            // a flagged sign: point_is_above itself, then on down
            const int p_ = rl(r.x, w);
            const uint32_t ry_ = (uint32_t)rl(r.y, w), l_ = ry_ & 0xFFFFu, rr_ = ry_ >> 16;
            w = plane_above_wave(f.planes + (size_t)p_ * d, f.consts[p_], Q + (size_t)b * d, d, lane) ? (rr_ | (l_ << 8)) : (l_ | (rr_ << 8));
            WALK_DESCEND();
        }
#undef WALK_DESCEND
        // the visited leaves' records join the ring (every other lane writes the spare slot: no branch), full groups leave the wave
        {
            const uint32_t slot = vidx != 0xFFFFFFFFu ? (vidx & (WALK_RING - 1)) : WALK_RING;
            vb_a[slot] = make_uint4((uint32_t)r.w, (uint32_t)r.y, (uint32_t)r.z, vtk);
            vb_r[slot] = vrow; vb_c[slot] = vcand;
            const uint32_t nv = (uint32_t)__builtin_amdgcn_readfirstlane((int)v_nv);
            while (nv - flushed >= WALK_FLUSH) flush(WALK_FLUSH);
        }
        // the block returned `ret` to the upper walk
        WP(p_dfs += clock64() - p_d0 - (p_flush - p_fl0); const uint64_t p_p0 = clock64();)
        bool down = false;
        __builtin_amdgcn_wave_barrier();
        while (usp > 0) {
            usp--;
            WP(p_upops++;)
            if (usp < WALK_STACK) {
                const int4 e = ust[usp];
                const int32_t nn = __builtin_amdgcn_readfirstlane(e.y);
                if (ret < nn) {
                    ref = __builtin_amdgcn_readfirstlane(e.x); pl = __builtin_amdgcn_readfirstlane(e.z); n = nn - ret;
                    down = true;
                    break;
                }
            }
        }
        WP(p_upper += clock64() - p_p0;)
        if (!down) break;
    }
    const uint32_t nv = (uint32_t)__builtin_amdgcn_readfirstlane((int)v_nv);
    if (nv > flushed) flush(nv - flushed);
    if (lane == 0) {
        ZhPairCounts c;
        c.visits = nv; c.rows = v_nrows; c.takes = v_ntakes; c.pad = 0;
        counts[pair] = c;
    }
    WP(if (lane == 0 && pair < 8192) {
        uint64_t *o = zh_walk_prof_buf + pair * 16;
        o[0] = clock64() - p_t0; o[1] = wall_clock64() - p_w0; o[2] = p_load; o[3] = p_upper; o[4] = p_flush; o[5] = p_dfs;
        o[6] = p_blocks; o[7] = p_uppers; o[8] = p_inner; o[9] = p_pops; o[10] = p_upops; o[11] = nv; o[12] = p_w0;
        uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); o[13] = hw;
    })
}

// ---- the same walk over blocks of INNER nodes (round 5): twice the tree per block entry, half the upper nodes ----
__global__ __launch_bounds__(64) void walk_blocked_inner_kernel(ZhForestDev f, ZhBlocksDev blk, uint32_t B, int32_t n0,
                                                           const uint32_t *__restrict__ bits, uint32_t wpq,
                                                           ZhPairCounts *__restrict__ counts, ZhVisit *__restrict__ inl,
                                                           uint32_t *__restrict__ leafCount, ZhWalkLog wlog,
                                                           const uint32_t *__restrict__ unc, const float *__restrict__ Q, uint32_t d) {
    // unc != null: the row-score hash left its uncertain signs flagged instead of fixing them (zh_score.hip, "lazy"): a flagged sign
    // is recomputed here with point_is_above's own arithmetic (plane_above_wave) when -- and only when -- the walk steps on its node
    __shared__ int4 ust[WALK_STACK];  // upper-level stack {child ref, n, the child's plane when it is an upper node}
    static_assert(ZH_BLOCK_INNER + 1 + WALK_FLUSH <= WALK_RING, "a block's visits and the pending ones fit the ring");
    __shared__ uint4 vb_a[WALK_RING + 1];
    __shared__ uint32_t vb_r[WALK_RING + 1], vb_c[WALK_RING + 1];
    const uint32_t T = f.n_trees, lane = threadIdx.x;
    const uint64_t pair = blockIdx.x;
    const uint32_t b = (uint32_t)(pair / T), t = (uint32_t)(pair % T);
    const uint32_t *__restrict__ qbits = bits + (size_t)b * wpq;
    const uint32_t *__restrict__ qunc = unc ? unc + (size_t)b * wpq : nullptr;
    if (n0 <= 0) {  // lsh.rs:306: the first leaf takes nothing and nothing is ever < 0: no visit
        if (lane == 0) { ZhPairCounts c; c.visits = 0; c.rows = 0; c.takes = 0; c.pad = 0; counts[pair] = c; }
        return;
    }
    // the pair's running totals: wave-uniform, kept in VGPRs (see in_vgpr)
    uint32_t v_nv = in_vgpr(0), v_nrows = in_vgpr(0), v_ntakes = in_vgpr(0);
    uint32_t flushed = 0;
    uint32_t log_chunk = 0xFFFFFFFFu;
    int32_t log_cn = -1;
    bool log_ok = true;
    WP(uint64_t p_t0 = clock64(); uint64_t p_w0 = wall_clock64(); uint64_t p_load = 0, p_upper = 0, p_flush = 0, p_dfs = 0;
       uint32_t p_blocks = 0, p_uppers = 0, p_inner = 0, p_pops = 0, p_upops = 0, p_flags = 0;)
    auto flush = [&](uint32_t cnt) {  // as walk_kernel's count-pass flush: the ring's oldest `cnt` (<= WALK_FLUSH) visits leave, one lane each
        WP(const uint64_t p_f0 = clock64();)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t first = flushed, vi = first + lane, upto = first + cnt;
        const bool mine = lane < cnt;
        const uint32_t slot = mine ? (vi & (WALK_RING - 1)) : WALK_RING;
        const uint4 va = vb_a[slot];
        ZhVisit v;
        v.b = b; v.leaf_off = va.y; v.len = va.z; v.take = va.w; v.node = va.x; v.pad = 0;
        v.row_off = vb_r[slot]; v.cand_off = vb_c[slot];
        if (mine) {
            if (leafCount) atomicAdd(&leafCount[v.node], 1u);  // (a prefiltered batch forms no leaf groups)
            if (vi < ZH_INLINE_VISITS) inl[pair * ZH_INLINE_VISITS + vi] = v;
        }
        if (upto > ZH_INLINE_VISITS && log_ok) {
            constexpr uint32_t PER = ZH_LOG_CHUNK - 1;
            static_assert(WALK_FLUSH < PER, "a flush may open at most one log chunk");
            const int32_t c1 = (int32_t)((upto - 1 - ZH_INLINE_VISITS) / PER);
            uint32_t newc = 0xFFFFFFFFu;
            if (c1 > log_cn) {
                uint32_t c = 0;
                if (lane == 0) c = atomicAdd(&wlog.ctl->next_chunk, 1u);
                c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                if (c >= wlog.capacity) {
                    if (lane == 0) wlog.ctl->overflow = 1u;
                    log_ok = false;
                } else {
                    newc = c;
                    if (lane == 0) {
                        if (log_cn < 0) wlog.head[pair] = c;
                        else wlog.pool[(size_t)log_chunk * ZH_LOG_CHUNK].x = c;
                    }
                }
            }
            if (log_ok) {
                if (mine && vi >= ZH_INLINE_VISITS) {
                    const uint32_t li = vi - ZH_INLINE_VISITS;
                    const uint32_t id = (int32_t)(li / PER) == log_cn ? log_chunk : newc;
                    wlog.pool[(size_t)id * ZH_LOG_CHUNK + 1 + li % PER] = wlog.leaf_entries ? make_uint2(v.leaf_off, v.take | (v.len << 16)) : make_uint2(v.node, v.take);
                }
                if (newc != 0xFFFFFFFFu) { log_chunk = newc; log_cn = c1; }
            }
        }
        flushed = upto;
        __builtin_amdgcn_wave_barrier();
        WP(p_flush += clock64() - p_f0;)
    };
    const int2 root = blk.root[t];
    int32_t ref = __builtin_amdgcn_readfirstlane(root.x), pl = __builtin_amdgcn_readfirstlane(root.y);
    int32_t n = n0;
    int usp = 0;
    for (;;) {
        if (ref >= 0) {  // an upper (inner) node: its records and its sign word are requested together
            WP(const uint64_t p_u0 = clock64(); p_uppers++;)
            // wave-uniform addresses: scalar loads (through the constant address space) -- the records arrive in SGPRs, no vector
            // memory instruction, no v_readfirstlane per word
            typedef const int32_t __attribute__((address_space(4))) *cptr_t;
            const cptr_t up = (cptr_t)(uintptr_t)(blk.upper + 2 * (size_t)ref);
            const int4 a = make_int4(up[0], up[1], up[2], up[3]), c2 = make_int4(up[4], up[5], up[6], up[7]);
            const uint32_t word = (uint32_t)((cptr_t)(uintptr_t)qbits)[(uint32_t)pl >> 5];
            const uint32_t uword = qunc ? (uint32_t)((cptr_t)(uintptr_t)qunc)[(uint32_t)pl >> 5] : 0u;
            bool above = (word >> (pl & 31)) & 1;
            if ((uword >> (pl & 31)) & 1)
                above = plane_above_wave(f.planes + (size_t)pl * d, f.consts[pl], Q + (size_t)b * d, d, lane);
            if (usp < WALK_STACK && lane == 0) ust[usp] = make_int4(above ? a.y : a.z, n, above ? c2.x : c2.y, 0);
            usp++;
            ref = above ? a.z : a.y;  // lsh.rs:335-338: above -> right is main
            pl = above ? c2.y : c2.x;
            WP(p_upper += clock64() - p_u0;)
            continue;
        }
        // ---- a block of INNER nodes (round 5; ZhBlocksDev::inner_only): 32-byte records and signs into registers, then the DFS without memory ----
        // Lane i holds inner node i of the block: {plane, codes, left leaf: offset, length} {left leaf: node, right leaf: offset, length, node}.  A child
        // code is an inner node's lane (0 .. 62) or, with bit 7, a LEAF: bit 6 = which child, bits 5:0 = the parent's lane -- the leaf's numbers are
        // read from its parent's registers and the parent's lane records the visit (a DFS meets a leaf once: one slot per side).  0xFF (lane 63,
        // never a node) is where the stack's sentinel leads once the block is exhausted.  A block of zero inner nodes is a leaf by itself: its
        // record has the layout of a left leaf of lane 0.
        WP(const uint64_t p_l0 = clock64(); p_blocks++;)
        const uint32_t s0 = (uint32_t)(-ref - 1);
        int4 ra = blk.recs[(size_t)s0 + lane], rb = blk.recs_b[(size_t)s0 + lane];  // (padded: 64 records can always be read; two contiguous 1-KiB loads)
        const uint32_t cnt = rl(ra.x, 0) >= 0 ? ((uint32_t)rl(ra.y, 0) >> 16) & 0x7Fu : 0u;
        int sgn = 0;
        uint32_t ub = 0;
        if (lane < cnt) {
            sgn = (int)((qbits[(uint32_t)ra.x >> 5] >> (ra.x & 31)) & 1u);
            if (qunc) ub = (qunc[(uint32_t)ra.x >> 5] >> (ra.x & 31)) & 1u;
        }
        // A flagged sign (the row-score hash left it uncertain: one node in ~400, 0.16 per block) is recomputed HERE, with point_is_above's own
        // arithmetic, for every flagged node of the block the walk enters -- not when the walk steps on it: the DFS below then has no flag to test
        // (its loop is bound by the SIMD's scalar issue slot: every scalar instruction per step counts; ~4x the exact chains of the lazy scheme,
        // < 1 % of a pair's cycles)
#ifndef ZH_PROBE_NO_EAGER_FLAGS  // (timing probe only: results are wrong without it)
        WP(p_flags += (uint32_t)__builtin_popcountll(__ballot(ub != 0));)
        for (unsigned long long um = __ballot(ub != 0); um; um &= um - 1) {
            const int j = __builtin_ctzll(um);
            const int p_ = rl(ra.x, (uint32_t)j);
            const bool ab = plane_above_wave(f.planes + (size_t)p_ * d, f.consts[p_], Q + (size_t)b * d, d, lane);
            if ((int)lane == j) sgn = ab ? 1 : 0;
        }
#endif
        // one word per inner node: main child's code | backup child's code << 8 (lsh.rs:335-338: above -> right is main)
        uint32_t pk;
        {
            const uint32_t cl = (uint32_t)ra.y & 0xFFu, cr = ((uint32_t)ra.y >> 8) & 0xFFu;
            pk = sgn ? (cr | (cl << 8)) : (cl | (cr << 8));
        }
        const uint32_t kL = lane | 0x80u, kR = lane | 0xC0u;  // the codes of this lane's left / right leaf child
        WP(if (__builtin_amdgcn_readfirstlane((int)pk) == 12345) p_blocks++; const uint64_t p_d0 = clock64(); p_load += p_d0 - p_l0;
           const uint64_t p_fl0 = p_flush;)
        // the block's DFS stack: lane j holds entry j = backup child's code | n << 8; entry 0 is a sentinel no return value ever exhausts
        uint32_t lstk = lane == 0 ? ((0x7FFFFFu << 8) | 0xFFu) : 0u;
        uint32_t lsp = 1;
        uint32_t vidxL = 0xFFFFFFFFu, vtkL = 0, vrowL = 0, vcandL = 0, vidxR = 0xFFFFFFFFu, vtkR = 0, vrowR = 0, vcandR = 0;
        uint32_t n8 = in_vgpr_shl8((uint32_t)n);
        int32_t ret = 0;
        uint32_t c = 0x80u;  // (wave-uniform) the code of the node the walk stands on: a lone leaf has the layout of lane 0's left leaf
        // from inner node `w` (its word) down the main children to a leaf's code: per inner node the perm, the stack write (compare + select), one
        // scalar add, the bit test + branch and the v_readlane whose lane select is the word itself (the hardware uses bits 5:0)
#define WALK_DESCEND()                                                                                                       \
        for (;;) {                                                                                                           \
            const uint32_t e_ = __builtin_amdgcn_perm(n8, w, 0x07060501u); /* byte 0 = the word's byte 1 (backup), bytes 1-3 = n */ \
            lstk = lane == lsp ? e_ : lstk;                                                                                  \
            lsp++;                                                                                                           \
            WP(p_inner++;)                                                                                                   \
            if (w & 0x80u) break;                                                                                            \
            w = (uint32_t)rl((int)pk, w);                                                                                    \
        }                                                                                                                    \
        c = w & 0xFFu;
        if (cnt) {
            uint32_t w = (uint32_t)rl((int)pk, 0);
            WALK_DESCEND();
        }
        while (c != 0xFFu) {  // one leaf per turn; 0xFF: through the sentinel -- the block is done and `ret` is its return value
            // the leaf: child (c >> 6 & 1) of the inner node in lane c & 63; an empty one (lsh.rs:306 / 329: it returns 0) is stepped on, not a visit
            const uint32_t lenl = (uint32_t)rl(ra.w, c), lenr = (uint32_t)rl(rb.z, c);  // (v_readlane takes bits 5:0 of the code)
            const uint32_t len = (c & 0x40u) ? lenr : lenl;
            const uint32_t take = len < (uint32_t)n ? len : (uint32_t)n;  // n >= 1 here (n0 >= 1; a backup is entered with nn - ret > 0)
            ret = (int32_t)take;
            const uint32_t cm = len ? c : 0xFFFFFFFFu;  // (an empty leaf matches no lane)
            const bool meL = kL == cm, meR = kR == cm;
            vidxL = meL ? v_nv : vidxL; vtkL = meL ? take : vtkL; vrowL = meL ? v_nrows : vrowL; vcandL = meL ? v_ntakes : vcandL;
            vidxR = meR ? v_nv : vidxR; vtkR = meR ? take : vtkR; vrowR = meR ? v_nrows : vrowR; vcandR = meR ? v_ntakes : vcandR;
            v_nv += in_vgpr(len) != 0u ? 1u : 0u; v_nrows += len; v_ntakes += take;
            uint32_t e;
            do {  // lsh.rs:341-343: the nearest pending backup whose n the return value has not used up
                lsp--;
                WP(p_pops++;)
                e = (uint32_t)rl((int)lstk, lsp);
            } while ((uint32_t)ret >= (e >> 8));
            n = (int32_t)((e >> 8) - (uint32_t)ret);
            n8 = in_vgpr_shl8((uint32_t)n);
            c = e & 0xFFu;
            if (!(c & 0x80u)) {  // the backup child is an inner node: down its main children
                uint32_t w = (uint32_t)rl((int)pk, c);
                WALK_DESCEND();
            }
        }
#undef WALK_DESCEND
        // the visited leaves' records join the ring (a lane without a visit on that side writes the spare slot: no branch), full groups leave the wave
        {
            const uint32_t sl = vidxL != 0xFFFFFFFFu ? (vidxL & (WALK_RING - 1)) : WALK_RING;
            vb_a[sl] = make_uint4((uint32_t)rb.x, (uint32_t)ra.z, (uint32_t)ra.w, vtkL);
            vb_r[sl] = vrowL; vb_c[sl] = vcandL;
            __builtin_amdgcn_wave_barrier();  // (the spare slot is written by both sides' idle lanes: keep the two rounds apart)
            const uint32_t sr = vidxR != 0xFFFFFFFFu ? (vidxR & (WALK_RING - 1)) : WALK_RING;
            vb_a[sr] = make_uint4((uint32_t)rb.w, (uint32_t)rb.y, (uint32_t)rb.z, vtkR);
            vb_r[sr] = vrowR; vb_c[sr] = vcandR;
            const uint32_t nv = (uint32_t)__builtin_amdgcn_readfirstlane((int)v_nv);
            while (nv - flushed >= WALK_FLUSH) flush(WALK_FLUSH);
        }
        // the block returned `ret` to the upper walk
        WP(p_dfs += clock64() - p_d0 - (p_flush - p_fl0); const uint64_t p_p0 = clock64();)
        bool down = false;
        __builtin_amdgcn_wave_barrier();
        while (usp > 0) {
            usp--;
            WP(p_upops++;)
            if (usp < WALK_STACK) {
                const int4 e = ust[usp];
                const int32_t nn = __builtin_amdgcn_readfirstlane(e.y);
                if (ret < nn) {
                    ref = __builtin_amdgcn_readfirstlane(e.x); pl = __builtin_amdgcn_readfirstlane(e.z); n = nn - ret;
                    down = true;
                    break;
                }
            }
        }
        WP(p_upper += clock64() - p_p0;)
        if (!down) break;
    }
    const uint32_t nv = (uint32_t)__builtin_amdgcn_readfirstlane((int)v_nv);
    if (nv > flushed) flush(nv - flushed);
    if (lane == 0) {
        ZhPairCounts c;
        c.visits = nv; c.rows = v_nrows; c.takes = v_ntakes; c.pad = 0;
        counts[pair] = c;
    }
    WP(if (lane == 0 && pair < 8192) {
        uint64_t *o = zh_walk_prof_buf + pair * 16;
        o[0] = clock64() - p_t0; o[1] = wall_clock64() - p_w0; o[2] = p_load; o[3] = p_upper; o[4] = p_flush; o[5] = p_dfs;
        o[6] = p_blocks; o[7] = p_uppers; o[8] = p_inner; o[9] = p_pops; o[10] = p_upops; o[11] = nv; o[12] = p_w0;
        uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); o[13] = hw; o[14] = p_flags;
    })
}

hipError_t zh_launch_walk_blocked(ZhForestDev f, ZhBlocksDev blk, uint32_t B, int32_t n, const uint32_t *dBits,
                                  uint32_t words_per_q, ZhPairCounts *dCounts, ZhVisit *dInline, uint32_t *dLeafCount,
                                  ZhWalkLog log, const uint32_t *dUnc, const float *dQ, uint32_t d, hipStream_t s) {
    const uint64_t pairs = (uint64_t)B * f.n_trees;
    if (!pairs) return hipSuccess;
    if (blk.inner_only)
        hipLaunchKernelGGL(walk_blocked_inner_kernel, dim3((uint32_t)pairs), dim3(64), 0, s, f, blk, B, n, dBits, words_per_q, dCounts,
                           dInline, dLeafCount, log, dUnc, dQ, d);
    else
        hipLaunchKernelGGL(walk_blocked_kernel, dim3((uint32_t)pairs), dim3(64), 0, s, f, blk, B, n, dBits, words_per_q, dCounts,
                           dInline, dLeafCount, log, dUnc, dQ, d);
    return hipGetLastError();
}

// expand: one wave per (query, tree) pair places the visits the counting pass recorded -- the inline ones, then the
// log's chunks, 63 entries at a time with a wave scan for the row / candidate offsets -- and joins the leaf groups.
__global__ __launch_bounds__(64) void expand_kernel(ZhForestDev f, uint32_t T, const ZhPairCounts *__restrict__ counts,
                                                     const ZhVisit *__restrict__ inl, const uint64_t *__restrict__ rowBase,
                                                     const uint64_t *__restrict__ candBase,
                                                     const uint64_t *__restrict__ visitBase, ZhVisit *__restrict__ visits,
                                                     const uint32_t *__restrict__ leafCount, uint32_t *__restrict__ leafFill,
                                                     const uint32_t *__restrict__ groupBase,
                                                     const uint64_t *__restrict__ groupRowBase, ZhGroup *__restrict__ groups,
                                                     uint64_t *__restrict__ groupRowOff, ZhWalkLog wlog) {
    const uint64_t pair = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t nv = counts[pair].visits;
    if (!nv) return;
    const uint32_t b = (uint32_t)(pair / T);
    const uint64_t vb = visitBase[pair], rb = rowBase[pair], cb = candBase[pair];
    const uint32_t n_inl = nv < ZH_INLINE_VISITS ? nv : ZH_INLINE_VISITS;
    if (lane < n_inl) {
        ZhVisit v = inl[pair * ZH_INLINE_VISITS + lane];
        v.row_off += rb;
        v.cand_off += cb;
        visits[vb + lane] = v;
        join_group(f.group, v, leafCount, leafFill, groupBase, groupRowBase, groups, groupRowOff);
    }
    if (nv <= ZH_INLINE_VISITS) return;
    const ZhVisit last = inl[pair * ZH_INLINE_VISITS + ZH_INLINE_VISITS - 1];
    uint64_t run_rows = last.row_off + last.len, run_takes = last.cand_off + last.take;
    uint32_t chunk = wlog.head[pair], idx = ZH_INLINE_VISITS, remaining = nv - ZH_INLINE_VISITS;
    while (remaining) {  // wave-uniform
        const uint32_t cnt = remaining < ZH_LOG_CHUNK - 1 ? remaining : ZH_LOG_CHUNK - 1;
        const uint2 *C = wlog.pool + (size_t)chunk * ZH_LOG_CHUNK;
        const uint32_t next = C[0].x;
        const bool on = lane >= 1 && lane <= cnt;
        uint2 e = make_uint2(0, 0);
        int4 r = make_int4(-1, 0, 0, 0);
        if (on) { e = C[lane]; r = f.node_pack[e.x]; }
        const uint32_t len = on ? (uint32_t)r.z : 0u, take = on ? e.y : 0u;
        uint32_t sl = len, stk = take;  // inclusive scans over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = __shfl_up(sl, o), c = __shfl_up(stk, o);
            if (lane >= (uint32_t)o) { sl += a; stk += c; }
        }
        if (on) {
            ZhVisit v;
            v.b = b; v.leaf_off = (uint32_t)r.y; v.len = len; v.take = take; v.node = e.x; v.pad = 0;
            v.row_off = rb + run_rows + (sl - len);
            v.cand_off = cb + run_takes + (stk - take);
            visits[vb + idx + lane - 1] = v;
            join_group(f.group, v, leafCount, leafFill, groupBase, groupRowBase, groups, groupRowOff);
        }
        run_rows += __shfl(sl, 63);
        run_takes += __shfl(stk, 63);
        idx += cnt; remaining -= cnt; chunk = next;
    }
}
hipError_t zh_launch_expand(ZhForestDev f, uint32_t B, const ZhPairCounts *dCounts, const ZhVisit *dInline,
                            const uint64_t *dRowBase, const uint64_t *dCandBase, const uint64_t *dVisitBase,
                            ZhVisit *dVisits, const uint32_t *dLeafCount, uint32_t *dLeafFill,
                            const uint32_t *dGroupBase, const uint64_t *dGroupRowBase, ZhGroup *dGroups,
                            uint64_t *dGroupRowOff, ZhWalkLog log, hipStream_t s) {
    uint64_t pairs = (uint64_t)B * f.n_trees;
    if (!pairs) return hipSuccess;
    hipLaunchKernelGGL(expand_kernel, dim3((uint32_t)pairs), dim3(64), 0, s, f, f.n_trees, dCounts, dInline, dRowBase,
                       dCandBase, dVisitBase, dVisits, dLeafCount, dLeafFill, dGroupBase, dGroupRowBase, dGroups,
                       dGroupRowOff, log);
    return hipGetLastError();
}

// leaf allocation: node i with c visits forms ceil(c / group) groups of len(i) rows each.  Group indices
// and flat row offsets are handed out by ONE packed 64-bit atomic (groups << 36 | rows), so both are
// monotone in the same (arbitrary) order -- the sweep's binary search only needs that.
__global__ __launch_bounds__(256) void leaf_alloc_kernel(ZhForestDev f, const uint32_t *__restrict__ leafCount,
                                                          uint32_t *__restrict__ groupBase,
                                                          uint64_t *__restrict__ groupRowBase,
                                                          unsigned long long *__restrict__ packed) {
    // one atomic per WAVE (wave scan of the packed increments, the last lane adds the wave's total): with the reference's
    // default options millions of visited leaves would otherwise queue on one address (2.4 ms for 6.5M atomics)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const uint32_t c = i < f.n_nodes ? leafCount[i] : 0;
    unsigned long long v = 0;
    if (c) {
        const uint64_t ng = (c + f.group - 1) / f.group;
        v = (unsigned long long)((ng << 36) | (ng * (uint32_t)f.node_right[i]));
    }
    unsigned long long incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(incl, o);
        if (lane >= (uint32_t)o) incl += t;
    }
    // ... and one per BLOCK: the four waves' totals meet in LDS (same-address atomics retire ~12 ns apart)
    __shared__ unsigned long long wtot[4], bbase;
    const uint32_t wv = threadIdx.x >> 6;
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        bbase = total ? atomicAdd(packed, total) : 0;
    }
    __syncthreads();
    if (c) {
        unsigned long long mine = bbase + incl - v;
        for (uint32_t w2 = 0; w2 < wv; w2++) mine += wtot[w2];
        groupBase[i] = (uint32_t)(mine >> 36);
        groupRowBase[i] = mine & ((1ull << 36) - 1);
    }
}
hipError_t zh_launch_leaf_scan(ZhForestDev f, const uint32_t *dLeafCount, uint32_t *dGroupBase,
                               uint64_t *dGroupRowBase, ZhTotals *dTotals, hipStream_t s) {
    // the packed counter lives in the totals' `flags` word (zeroed by zh_launch_batch_init, read and rewritten by the
    // pair scan, which runs next)
    unsigned long long *packed = reinterpret_cast<unsigned long long *>(&dTotals->flags);
    hipLaunchKernelGGL(leaf_alloc_kernel, dim3((f.n_nodes + 255) / 256), dim3(256), 0, s, f, dLeafCount, dGroupBase,
                       dGroupRowBase, packed);
    return hipGetLastError();
}

// everything a batch starts from zero, in one launch (separate memsets are separate kernels, each a scheduling
// round trip on a busy GPU): per-leaf visit counts and fill cursors, the visit log's allocator, the packed
// group / row counter of the leaf allocation
__global__ __launch_bounds__(256) void batch_init_kernel(uint32_t *__restrict__ leafCount, uint32_t *__restrict__ leafFill,
                                                          uint32_t n_nodes, ZhLogCtl *__restrict__ logCtl,
                                                          ZhTotals *__restrict__ totals) {
    const uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (uint32_t i = i0; i < n_nodes; i += stride) { leafCount[i] = 0; leafFill[i] = 0; }
    if (i0 == 0) { logCtl->next_chunk = 0; logCtl->overflow = 0; totals->flags = 0; totals->hash_fixups = 0; }
}
hipError_t zh_launch_batch_init(uint32_t *dLeafCount, uint32_t *dLeafFill, uint32_t n_nodes, ZhLogCtl *dLogCtl,
                                ZhTotals *dTotals, hipStream_t s) {
    uint32_t blocks = (n_nodes + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(batch_init_kernel, dim3(blocks), dim3(256), 0, s, dLeafCount, dLeafFill, n_nodes, dLogCtl, dTotals);
    return hipGetLastError();
}

// exclusive scans of the three per-pair counts: ONE 256-thread block walks the pairs 1024 at a time -- four consecutive pairs
// per thread (64 contiguous bytes), a serial scan of the four in registers, wave scans by shuffle + one LDS hop across the 4
// waves, a running total carried between chunks.  256 threads, not 1024: beside a sweep every CU is full of 256-thread sweep
// blocks, and a work-group only starts when ALL its waves fit at once -- a 16-wave group waited 6-8 ms for four sweep blocks of
// one CU to retire together (rocprofv3 kernel trace, r03: 0.1 ms alone, 7.7 ms beside the table scan, and the host's finish()
// waits for exactly this kernel's totals), a 4-wave group takes the first slot any sweep block leaves.
// Also finishes the totals: groups / group rows out of the leaf allocation's packed counter, the log's overflow flag.
__global__ __launch_bounds__(256) void pair_scan_kernel(const ZhPairCounts *__restrict__ counts, uint32_t n,
                                                         uint64_t *__restrict__ rowBase,
                                                         uint64_t *__restrict__ candBase,
                                                         uint64_t *__restrict__ visitBase,
                                                         ZhTotals *__restrict__ totals,
                                                         const ZhLogCtl *__restrict__ logCtl) {
    __shared__ uint64_t wr[4], wc[4], wv[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint64_t run_r = 0, run_c = 0, run_v = 0;
    for (uint32_t base = 0; base < n; base += 1024) {  // block-uniform
        const uint32_t i0 = base + 4 * tid;
        ZhPairCounts pc[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            pc[j].visits = 0; pc[j].rows = 0; pc[j].takes = 0; pc[j].pad = 0;
            if (i0 + j < n) pc[j] = counts[i0 + j];
        }
        uint64_t r = 0, c = 0, v = 0;  // the thread's four pairs, then inclusive scans of the thread totals inside the wave
#pragma unroll
        for (int j = 0; j < 4; j++) { r += pc[j].rows; c += pc[j].takes; v += pc[j].visits; }
        const uint64_t tr4 = r, tc4 = c, tv4 = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t a = __shfl_up(r, o), b2 = __shfl_up(c, o), d2 = __shfl_up(v, o);
            if (lane >= (uint32_t)o) { r += a; c += b2; v += d2; }
        }
        if (lane == 63) { wr[w] = r; wc[w] = c; wv[w] = v; }
        __syncthreads();
        uint64_t pr = 0, pcn = 0, pv = 0, tr = 0, tc = 0, tv = 0;  // totals of the waves before mine / of the chunk
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint64_t a = wr[j], b2 = wc[j], d2 = wv[j];
            if (j < w) { pr += a; pcn += b2; pv += d2; }
            tr += a; tc += b2; tv += d2;
        }
        uint64_t er = run_r + pr + r - tr4, ec = run_c + pcn + c - tc4, evv = run_v + pv + v - tv4;  // exclusive, at the thread's first pair
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (i0 + j < n) { rowBase[i0 + j] = er; candBase[i0 + j] = ec; visitBase[i0 + j] = evv; }
            er += pc[j].rows; ec += pc[j].takes; evv += pc[j].visits;
        }
        run_r += tr; run_c += tc; run_v += tv;
        __syncthreads();
    }
    if (tid == 0) {
        rowBase[n] = run_r; candBase[n] = run_c; visitBase[n] = run_v;
        const unsigned long long packed = *reinterpret_cast<const unsigned long long *>(&totals->flags);
        totals->groups = packed >> 36;
        totals->group_rows = packed & ((1ull << 36) - 1);
        totals->rows = run_r; totals->takes = run_c; totals->visits = run_v;
        totals->flags = logCtl->overflow ? 1u : 0u;
    }
}

hipError_t zh_launch_pair_scan(const ZhPairCounts *dCounts, uint32_t n_pairs, uint64_t *dRowBase,
                               uint64_t *dCandBase, uint64_t *dVisitBase, ZhTotals *dTotals, const ZhLogCtl *dLogCtl,
                               hipStream_t s) {
    hipLaunchKernelGGL(pair_scan_kernel, dim3(1), dim3(256), 0, s, dCounts, n_pairs, dRowBase, dCandBase, dVisitBase,
                       dTotals, dLogCtl);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// sweep: the HBM-bound kernel.  The rows of all visits of the batch form one flat sequence
// [0, R_total); each wave owns 64 consecutive flat rows: lane i resolves flat row r0+i to
// (visit -> query b, stored row id), then the wave streams the 64 rows one after another --
// every lane loading one float4 per 1 KiB of row (coalesced 1 KiB wave loads), RG rows in flight --
// and lane i keeps the canonical sums of row i.  Keys go to the scratch (8 B per 4*D B read).
// ------------------------------------------------------------------------------------------------
// Which group does the wave's first flat row fall in?  One thread per group writes the slots of the wave starts its row range
// covers (waveGroup[w] = group of flat row 64 w), so a sweep wave starts with ONE load instead of a ~17-step binary search
// over all groups -- a prologue that costs a 32-KB wave of 512-byte rows as much time as its rows.
__global__ __launch_bounds__(256) void wave_group_kernel(const ZhGroup *__restrict__ groups, const uint64_t *__restrict__ groupRowOff,
                                                          uint64_t n_groups, uint32_t *__restrict__ waveGroup) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const uint64_t off = groupRowOff[g], end = off + groups[g].len;
    for (uint64_t w = (off + 63) >> 6; (w << 6) < end; w++) waveGroup[w] = (uint32_t)g;
}
hipError_t zh_launch_wave_groups(const ZhGroup *dGroups, const uint64_t *dGroupRowOff, uint64_t n_groups, uint32_t *dWaveGroup,
                                 hipStream_t s) {
    if (!n_groups) return hipSuccess;
    hipLaunchKernelGGL(wave_group_kernel, dim3((uint32_t)((n_groups + 255) / 256)), dim3(256), 0, s, dGroups, dGroupRowOff, n_groups,
                       dWaveGroup);
    return hipGetLastError();
}


// one row against the (up to G) queries of its group.  s0[m] (and s1[m] for Bray-Curtis) are per
// member; for cosine s1[0] carries a2, the stored row's norm, shared by the members
template <int D, int KIND, int G>
__device__ __forceinline__ void row_sums_group(const float4 *v, const float4 (*q)[RowVec<D>::NV], uint32_t gsize,
                                               uint32_t lane, int power, float *s0, float *s1) {
    constexpr int NV = RowVec<D>::NV;
    if (KIND == K_COS) {
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NV; j++) {
            bool act = (j < RowVec<D>::NJ) || (lane < (uint32_t)RowVec<D>::REM4);
            if (act) sq4(v[j], c);
        }
        s1[0] = wave_sum_canonical((c.x + c.y) + (c.z + c.w));
    }
#pragma unroll
    for (int m = 0; m < G; m++) {
        if ((uint32_t)m < gsize) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), e = a;
#pragma unroll
            for (int j = 0; j < NV; j++) {
                bool act = (j < RowVec<D>::NJ) || (lane < (uint32_t)RowVec<D>::REM4);
                if (act) acc4<KIND>(v[j], q[m][j], a, e, power);
            }
            s0[m] = wave_combine<KIND>(a.x, a.y, a.z, a.w);
            if (KIND == K_BRAY) s1[m] = wave_combine<K_L2>(e.x, e.y, e.z, e.w);
        }
    }
}

// D > 0: compile-time dimension (multiple of 4); D == 0: runtime d, any value (slow path)
// QLDS (A/B only, ZH_SWEEP_QLDS=1): the group's queries staged in LDS (one member's float4s in registers at a time) instead of
// all G members in registers -- the north_star's "LDS-staged query tiles" taken literally; measured equal (62.2 k against 61.7 k QPS at cfg3:
// the kernel is HBM-bound either way, profiles/r02_ab_query_staging.txt), registers kept.
template <int D, int KIND, int G, int SWEEP_RG = 4, bool NT = false, bool QLDS = false>
__global__ __launch_bounds__(256) void sweep_kernel(const float *__restrict__ X, uint32_t d,
                                                     const float *__restrict__ Q, const float *__restrict__ QQ,
                                                     const ZhGroup *__restrict__ groups,
                                                     const uint64_t *__restrict__ groupRowOff, uint64_t n_groups,
                                                     const uint32_t *__restrict__ waveGroup,
                                                     const uint32_t *__restrict__ leaf_ids, uint64_t row_begin,
                                                     uint64_t R_grouped, int metric, int param,
                                                     uint64_t *__restrict__ keys) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t r0 = row_begin + wave * 64;  // this launch covers flat rows [row_begin, R_grouped)
    if (r0 >= R_grouped) return;
    const uint32_t cnt = (uint32_t)(R_grouped - r0 < 64 ? R_grouped - r0 : 64);
    // lane i -> (group, stored row) of flat row r0 + i
    uint32_t my_g, my_id, my_within;
    resolve_flat_rows(r0, cnt, lane, groups, groupRowOff, n_groups, waveGroup, leaf_ids, my_g, my_id, my_within);
    float mine0[G], mine1[G];
#pragma unroll
    for (int m = 0; m < G; m++) { mine0[m] = 0.f; mine1[m] = 0.f; }
    if (D > 0) {
        constexpr int DD = (D > 0 ? D : 4);
        constexpr int NV = RowVec<DD>::NV;
        float4 q[QLDS ? 1 : G][NV];
        __shared__ float4 qs[QLDS ? 4 : 1][QLDS ? G : 1][QLDS ? NV : 1][QLDS ? 64 : 1];
        const uint32_t wq = (threadIdx.x >> 6) & 3;
        uint32_t cur_g = 0xFFFFFFFFu, gsize = 0;
        for (uint32_t i0 = 0; i0 < cnt; i0 += SWEEP_RG) {
            float4 v[SWEEP_RG][NV];
#pragma unroll
            for (int r = 0; r < SWEEP_RG; r++) {
                uint32_t i = i0 + r < cnt ? i0 + r : cnt - 1;
                uint32_t id = __builtin_amdgcn_readlane(my_id, i);
                load_row<DD, NT>(X + (size_t)id * DD, lane, v[r]);
            }
#pragma unroll
            for (int r = 0; r < SWEEP_RG; r++) {
                uint32_t i = i0 + r;
                if (i < cnt) {
                    uint32_t g = __builtin_amdgcn_readlane(my_g, i);
                    if (g != cur_g) {
                        cur_g = g;
                        gsize = groups[g].gsize;
#pragma unroll
                        for (int m = 0; m < G; m++)
                            if ((uint32_t)m < gsize) {
                                if constexpr (QLDS) {
                                    load_row<DD>(Q + (size_t)groups[g].b[m] * DD, lane, q[0]);
#pragma unroll
                                    for (int j = 0; j < NV; j++) qs[wq][m][j][lane] = q[0][j];
                                } else
                                    load_row<DD>(Q + (size_t)groups[g].b[m] * DD, lane, q[m]);
                            }
                    }
                    float s0[G], s1[G];
#pragma unroll
                    for (int m = 0; m < G; m++) { s0[m] = 0.f; s1[m] = 0.f; }
                    if constexpr (QLDS) {
#pragma unroll
                        for (int m = 0; m < G; m++)
                            if ((uint32_t)m < gsize) {
#pragma unroll
                                for (int j = 0; j < NV; j++) q[0][j] = qs[wq][m][j][lane];
                                float t0[1] = {0.f}, t1[1] = {0.f};
                                row_sums_group<DD, KIND, 1>(v[r], q, 1, lane, param, t0, t1);
                                s0[m] = t0[0]; s1[m] = t1[0];
                                if (KIND == K_COS) s1[0] = t1[0];
                            }
                    } else
                        row_sums_group<DD, KIND, G>(v[r], q, gsize, lane, param, s0, s1);
                    if (lane == i) {
#pragma unroll
                        for (int m = 0; m < G; m++) { mine0[m] = s0[m]; mine1[m] = KIND == K_COS ? s1[0] : s1[m]; }
                    }
                }
            }
        }
    } else {
        for (uint32_t i = 0; i < cnt; i++) {
            uint32_t id = __builtin_amdgcn_readlane(my_id, i);
            uint32_t g = __builtin_amdgcn_readlane(my_g, i);
            uint32_t gsize = groups[g].gsize;
            for (uint32_t m = 0; m < gsize; m++) {
                float t0, t1;
                lane_sums_generic<KIND>(X + (size_t)id * d, Q + (size_t)groups[g].b[m] * d, d, lane, param, t0, t1);
                if (lane == i) {
#pragma unroll
                    for (int mm = 0; mm < G; mm++)
                        if ((uint32_t)mm == m) { mine0[mm] = t0; mine1[mm] = t1; }
                }
            }
        }
    }
    if (lane < cnt) {
        const ZhGroup grp = groups[my_g];
#pragma unroll
        for (int m = 0; m < G; m++)
            if ((uint32_t)m < grp.gsize)
                keys[grp.key_off[m] + my_within] = key_of(metric, param, mine0[m], mine1[m], KIND == K_COS ? QQ[grp.b[m]] : 0.f);
    }
}

// ---- d = 128 (SIFT-style shards): a 512-byte row is HALF a wave-load, so the wave streams TWO rows per instruction ----
// Lanes 0..31 take flat row j of the wave's 64, lanes 32..63 flat row j + 32: every load instruction is a full 1-KiB wave
// load (two 512-byte segments), the per-row instruction count halves (one butterfly serves two rows) and RG instructions
// keep 2 * RG rows in flight.  Same canonical sums: lane l < 32 of a row's half holds accumulators 4l..4l+3, the butterfly
// steps 1..16 stay inside the half, and step 32 adds the other half's total -- which is +0.0 at d <= 128, added explicitly.
__device__ __forceinline__ float half_sum_canonical(float s) {
    s = s + dpp_mov<0xB1>(s);   // xor 1
    s = s + dpp_mov<0x4E>(s);   // xor 2
    s = s + dpp_mov<0x141>(s);  // other quad of the 8
    s = s + dpp_mov<0x140>(s);  // other 8 of the 16
    s = xor16<OpAdd>(s);        // xor 16 (inside a 32-lane half)
    return s + 0.0f;            // xor 32: lanes 32..63 of a 128-d row hold no elements
}

// A wave takes CH consecutive 64-row chunks.  Resolving a chunk (wave-start table -> group offsets -> group record -> leaf ids)
// is a chain of four dependent loads, ~3 us before the first row load can go out -- a third of the life of a wave that then
// streams 32 KB (profiles/micro/gather512.hip: the bare gather of random 512-byte rows runs at 6.55 TB/s, this kernel with
// one chunk per wave at 5.3-5.5).  Chunks after the first usually continue in the group the previous chunk ended in (leaves
// hold thousands of rows): their ids are then ONE independent load, issued before the current chunk is streamed.
template <int KIND, int G, int RG, bool NT, int CH>
__global__ __launch_bounds__(256) void sweep128_kernel(const float *__restrict__ X, const float *__restrict__ Q,
                                                        const float *__restrict__ QQ, const ZhGroup *__restrict__ groups,
                                                        const uint64_t *__restrict__ groupRowOff, uint64_t n_groups,
                                                        const uint32_t *__restrict__ waveGroup,
                                                        const uint32_t *__restrict__ leaf_ids, uint64_t row_begin,
                                                        uint64_t R_grouped, int metric, int param,
                                                        uint64_t *__restrict__ keys, const uint32_t *__restrict__ run_if) {
    static_assert(KIND == K_L2 || KIND == K_COS, "the half-wave sweep covers the two simsimd-path kinds");
    if (run_if && *run_if == 0) return;  // (the redo behind a half-width sweep whose lists did not run over: zh_approx.hip)
    // A PREDICATED launch (run_if) is a small grid whose blocks stride over the work: when the predicate is zero -- all but never -- a few
    // thousand waves look at it and leave, instead of one wave per 256 rows (0.1 ms per 25M-row launch, 8 % of the half-width sweep it stands
    // behind: profiles/r06_sweep128h_experiments.txt).  An ordinary launch has a block per unit and runs the body once.
    for (uint64_t blk_i = blockIdx.x;; blk_i += gridDim.x) {
    const uint32_t lane = threadIdx.x & 63, hl = lane & 31;
    const bool up = lane >= 32;
    const uint64_t wave = blk_i * (blockDim.x >> 6) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint64_t r0 = row_begin + wave * (64 * CH);
    if (r0 >= R_grouped) return;
    uint32_t cnt = (uint32_t)(R_grouped - r0 < 64 ? R_grouped - r0 : 64);
    uint32_t my_g, my_id, my_within, my_off, my_len;
    resolve_flat_rows(r0, cnt, lane, groups, groupRowOff, n_groups, waveGroup, leaf_ids, my_g, my_id, my_within, &my_off, &my_len);
    const float4 *__restrict__ X4 = reinterpret_cast<const float4 *>(X);
    const float4 *__restrict__ Q4 = reinterpret_cast<const float4 *>(Q);
    float4 q[G];
#pragma unroll
    for (int m = 0; m < G; m++) q[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t cur_g = 0xFFFFFFFFu, gsize = 0;  // per half
    for (int c = 0; c < CH; c++) {
        // the next chunk: does it lie entirely in the group this chunk's last row belongs to?  (wave-uniform)
        const uint64_t r0n = r0 + 64;
        const bool have_next = c + 1 < CH && r0n < R_grouped;
        const uint32_t cntn = have_next ? (uint32_t)(R_grouped - r0n < 64 ? R_grouped - r0n : 64) : 0;
        const uint32_t lw = (uint32_t)__builtin_amdgcn_readlane((int)my_within, (int)cnt - 1) + 1;  // position after this chunk's last row
        const uint32_t loff = (uint32_t)__builtin_amdgcn_readlane((int)my_off, (int)cnt - 1);
        const uint32_t llen = (uint32_t)__builtin_amdgcn_readlane((int)my_len, (int)cnt - 1);
        const uint32_t lg = (uint32_t)__builtin_amdgcn_readlane((int)my_g, (int)cnt - 1);
        const bool fast = have_next && cnt == 64 && (uint64_t)lw + cntn <= llen;
        uint32_t nxt_id = 0;
        if (fast) {
            const uint32_t wn = lw + (lane < cntn ? lane : cntn - 1);
            nxt_id = leaf_ids ? leaf_ids[(size_t)loff + wn] : loff + wn;  // in flight while this chunk is streamed
        }
        float mine0[G], mine1 = 0.f;
#pragma unroll
        for (int m = 0; m < G; m++) mine0[m] = 0.f;
        const uint32_t npair = cnt < 32 ? cnt : 32;  // pair j = flat rows j (lower half) and j + 32 (upper half, if < cnt)
        for (uint32_t j0 = 0; j0 < npair; j0 += RG) {
            float4 v[RG];
#pragma unroll
            for (int r = 0; r < RG; r++) {
                const uint32_t j = j0 + r < npair ? j0 + r : npair - 1;
                const uint32_t jh = j + 32 < cnt ? j + 32 : cnt - 1;
                const uint32_t idl = __builtin_amdgcn_readlane(my_id, j), idh = __builtin_amdgcn_readlane(my_id, jh);
                v[r] = ld16<NT>(X4 + (size_t)(up ? idh : idl) * 32 + hl);
            }
#pragma unroll
            for (int r = 0; r < RG; r++) {
                const uint32_t j = j0 + r;
                if (j < npair) {
                    const uint32_t jh = j + 32 < cnt ? j + 32 : cnt - 1;
                    const uint32_t gl = __builtin_amdgcn_readlane(my_g, j), gh = __builtin_amdgcn_readlane(my_g, jh);
                    const uint32_t g = up ? gh : gl;
                    if (g != cur_g) {  // uniform inside a half
                        cur_g = g;
                        gsize = groups[g].gsize;
#pragma unroll
                        for (int m = 0; m < G; m++)
                            if ((uint32_t)m < gsize) q[m] = Q4[(size_t)groups[g].b[m] * 32 + hl];
                    }
                    float a2 = 0.f;
                    if (KIND == K_COS) {
                        float4 cc;
                        cc.x = __builtin_fmaf(v[r].x, v[r].x, 0.f); cc.y = __builtin_fmaf(v[r].y, v[r].y, 0.f);
                        cc.z = __builtin_fmaf(v[r].z, v[r].z, 0.f); cc.w = __builtin_fmaf(v[r].w, v[r].w, 0.f);
                        a2 = half_sum_canonical((cc.x + cc.y) + (cc.z + cc.w));
                    }
                    float s0[G];
#pragma unroll
                    for (int m = 0; m < G; m++) {
                        s0[m] = 0.f;
                        if ((uint32_t)m < gsize) {
                            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), e = a;
                            acc4<KIND>(v[r], q[m], a, e, param);
                            s0[m] = half_sum_canonical((a.x + a.y) + (a.z + a.w));
                        }
                    }
                    if (hl == j) {  // lane j holds flat row j, lane 32 + j flat row j + 32
#pragma unroll
                        for (int m = 0; m < G; m++) mine0[m] = s0[m];
                        mine1 = a2;
                    }
                }
            }
        }
        if (lane < cnt) {
            // only the fields the keys need (the 64-byte group record per lane was four 16-byte load instructions per chunk
            // on a kernel whose texture addresser is ~80 % busy with the rows)
            const ZhGroup *grp = groups + my_g;
            const uint32_t gs = grp->gsize;
#pragma unroll
            for (int m = 0; m < G; m++)
                if ((uint32_t)m < gs)
                    keys[grp->key_off[m] + my_within] = key_of(metric, param, mine0[m], mine1, KIND == K_COS ? QQ[grp->b[m]] : 0.f);
        }
        if (!have_next) break;
        r0 = r0n; cnt = cntn;
        if (fast) {  // the same group continues: nothing to search
            my_g = lg; my_off = loff; my_len = llen;
            my_within = lw + (lane < cntn ? lane : cntn - 1);
            my_id = nxt_id;
        } else
            resolve_flat_rows(r0, cnt, lane, groups, groupRowOff, n_groups, waveGroup, leaf_ids, my_g, my_id, my_within, &my_off, &my_len);
    }
    if (!run_if) return;
    }
}

uint64_t zh_sweep_rows_per_launch(uint32_t d) {
    static const uint64_t launch_bytes = [] { const char *e = getenv("ZH_SWEEP_LAUNCH_MB"); return (uint64_t)(e ? atoi(e) : 12288) << 20; }();
    uint64_t rows = launch_bytes / ((uint64_t)4 * (d ? d : 1));
    rows = (rows + 255) / 256 * 256;
    return rows < 65536 ? 65536 : rows;
}

struct SweepArgs {
    const float *dX; uint32_t d; const float *dQ, *dQQ;
    const ZhGroup *dGroups; const uint64_t *dGroupRowOff; uint64_t n_groups;
    const uint32_t *dWaveGroup;  // wave-start table (zh_launch_wave_groups) or nullptr: full binary search per lane
    const uint32_t *dLeafIds; uint64_t R_grouped; int metric, param; uint64_t *dKeys; uint32_t group; hipStream_t s;
    const uint32_t *dRunIf;  // a predicate on the device (d = 128 half-wave kernel only): the launch returns at once when it is zero
};

template <int D, int KIND, int G>
static hipError_t launch_sweep_g(const SweepArgs &a) {
    // rows are streamed once per batch: non-temporal loads (+1.3 % measured, profiles/); ZH_SWEEP_VARIANT=1
    // switches them off for A/B runs.  Rows in flight (2/4/8) made no measurable difference at d = 768: 4.
    static const int variant = [] { const char *e = getenv("ZH_SWEEP_VARIANT"); return e ? atoi(e) : 0; }();
    // d = 128: the half-wave kernel, RG load instructions = 2 RG rows in flight (ZH_SWEEP128=0: the generic kernel; A/B)
    static const int v128 = [] { const char *e = getenv("ZH_SWEEP128"); return e ? atoi(e) : 8; }();
    // One batch is issued as several launches of ~ZH_SWEEP_LAUNCH_BYTES each (about 2 ms of HBM time): a single
    // 18-ms dispatch keeps its dispatch pipe busy until its last workgroup is issued, and kernels of other
    // queues that share the pipe (the next batch's hash / walk, RCCL) would wait that long.
    // (a PREDICATED sweep -- the redo behind a half-width sweep, all but never run -- is ONE launch of a small striding grid for the whole batch)
    const uint64_t rows_per_launch = a.dRunIf ? (a.R_grouped + 255) / 256 * 256 : zh_sweep_rows_per_launch(a.d);
    for (uint64_t r = 0; r < a.R_grouped; r += rows_per_launch) {
        uint64_t r_end = r + rows_per_launch < a.R_grouped ? r + rows_per_launch : a.R_grouped;
        uint64_t waves = (r_end - r + 63) / 64;
        uint64_t blocks = (waves + 3) / 4;
        if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
        const dim3 grid((uint32_t)blocks), blk(256);
        if constexpr (D == 128 && (KIND == K_L2 || KIND == K_COS)) {
            if (v128 > 0) {
                // ZH_SWEEP128_CHUNKS (A/B): 64-row chunks per wave, 1 = the round-2 kernel's shape
                static const int ch128 = [] { const char *e = getenv("ZH_SWEEP128_CHUNKS"); return e ? atoi(e) : 4; }();
#define ZH_S128(RG_, CH_) do { const uint64_t w_ = (r_end - r + 64 * CH_ - 1) / (64 * CH_), b_ = (w_ + 3) / 4;                        \
                               hipLaunchKernelGGL((sweep128_kernel<KIND, G, RG_, true, CH_>), dim3((uint32_t)(a.dRunIf && b_ > 2048 ? 2048 : b_)), blk, 0, a.s, a.dX, a.dQ, a.dQQ, \
                                                  a.dGroups, a.dGroupRowOff, a.n_groups, a.dWaveGroup, a.dLeafIds, r, r_end, a.metric, a.param, a.dKeys, a.dRunIf); } while (0)
                if (ch128 == 1) ZH_S128(8, 1);
                else if (ch128 == 8) ZH_S128(8, 8);
                else if (v128 == 4) ZH_S128(4, 4);
                else if (v128 == 16) ZH_S128(16, 4);
                else ZH_S128(8, 4);
#undef ZH_S128
                continue;
            }
        }
        if constexpr (D == 768 && KIND == K_L2 && G == 4) {
            static const bool qlds = getenv("ZH_SWEEP_QLDS") != nullptr;
            if (qlds) {
                hipLaunchKernelGGL((sweep_kernel<D, KIND, G, 4, true, true>), grid, blk, 0, a.s, a.dX, a.d, a.dQ, a.dQQ, a.dGroups,
                                   a.dGroupRowOff, a.n_groups, a.dWaveGroup, a.dLeafIds, r, r_end, a.metric, a.param, a.dKeys);
                continue;
            }
        }
        if (variant == 1)
            hipLaunchKernelGGL((sweep_kernel<D, KIND, G, 4, false>), grid, blk, 0, a.s, a.dX, a.d, a.dQ, a.dQQ, a.dGroups,
                               a.dGroupRowOff, a.n_groups, a.dWaveGroup, a.dLeafIds, r, r_end, a.metric, a.param, a.dKeys);
        else
            hipLaunchKernelGGL((sweep_kernel<D, KIND, G, 4, true>), grid, blk, 0, a.s, a.dX, a.d, a.dQ, a.dQQ, a.dGroups,
                               a.dGroupRowOff, a.n_groups, a.dWaveGroup, a.dLeafIds, r, r_end, a.metric, a.param, a.dKeys);
    }
    return hipGetLastError();
}

template <int D, int KIND>
static hipError_t launch_sweep_k(const SweepArgs &a) {
    if (a.group == 4) return launch_sweep_g<D, KIND, 4>(a);
    return launch_sweep_g<D, KIND, 2>(a);
}

// the two simsimd-path kinds get every specialised dimension; the ten `distances`-path kinds the three
// production dimensions (text 384, image/audio 768, SIFT-style 128) and otherwise the runtime-d kernel
template <int KIND>
static hipError_t launch_sweep_kind(const SweepArgs &a) {
#define ZH_SWEEP_CASE(DD) case DD: return launch_sweep_k<DD, KIND>(a)
    if constexpr (KIND == K_L2 || KIND == K_COS) {
        switch (a.d) {
            ZH_SWEEP_CASE(64);
            ZH_SWEEP_CASE(256);
            ZH_SWEEP_CASE(512);
            ZH_SWEEP_CASE(1024);
            ZH_SWEEP_CASE(1536);
        default: break;
        }
    }
    switch (a.d) {
        ZH_SWEEP_CASE(128);
        ZH_SWEEP_CASE(384);
        ZH_SWEEP_CASE(768);
    default: return launch_sweep_k<0, KIND>(a);
    }
#undef ZH_SWEEP_CASE
}

// the leaf-major sweep that takes a device-side predicate: the d = 128 half-wave kernel of the two simsimd-path kinds (not under ZH_SWEEP128=0)
bool zh_sweep_has_predicate(uint32_t d, int metric) {
    static const int v128 = [] { const char *e = getenv("ZH_SWEEP128"); return e ? atoi(e) : 8; }();
    const int kind = zh_kind_of(metric);
    return d == 128 && v128 > 0 && (kind == K_L2 || kind == K_COS);
}

uint32_t zh_group_size(uint32_t dim) {
    static const int forced = [] { const char *e = getenv("ZH_GROUP"); return e ? atoi(e) : 0; }();
    if (forced == 2 || forced == 4) return (uint32_t)forced;
    return 4u;  // (d = 128 with the half-wave kernel: 4 measures equal or better than 2, profiles/r02_ab_sweep128.txt)
}

hipError_t zh_launch_sweep(const float *dX, uint32_t d, const float *dQ, const float *dQQ, const ZhGroup *dGroups,
                           const uint64_t *dGroupRowOff, uint64_t n_groups, const uint32_t *dWaveGroup,
                           const uint32_t *dLeafIds, uint64_t R_grouped, int metric, int param, uint64_t *dKeys,
                           uint32_t group, hipStream_t s, const uint32_t *dRunIf) {
    if (R_grouped == 0 || n_groups == 0) return hipSuccess;
    if (dRunIf && !zh_sweep_has_predicate(d, metric)) return hipErrorInvalidValue;
    const SweepArgs a{dX, d, dQ, dQQ, dGroups, dGroupRowOff, n_groups, dWaveGroup, dLeafIds, R_grouped, metric, param, dKeys,
                      group == 4 ? 4u : 2u, s, dRunIf};
#define ZH_KIND_CASE(K) case K: return launch_sweep_kind<K>(a)
    switch (zh_kind_of(metric)) {
        ZH_KIND_CASE(K_COS);
        ZH_KIND_CASE(K_MAX);
        ZH_KIND_CASE(K_CANB);
        ZH_KIND_CASE(K_BRAY);
        ZH_KIND_CASE(K_ABS);
        ZH_KIND_CASE(K_P3);
        ZH_KIND_CASE(K_P4);
        ZH_KIND_CASE(K_HAMM);
        ZH_KIND_CASE(K_PP);
    default: return launch_sweep_kind<K_L2>(a);
    }
#undef ZH_KIND_CASE
}

// ------------------------------------------------------------------------------------------------
// TABLE-SCAN sweep.  With the batch sizes of the BASELINE configurations a batch scores several times more (row, query)
// pairs than the index has rows (cfg3: 1024 queries x 15 trees x ~2.8k rows = 43M pairs over 10M rows): every stored row
// is wanted by ~4 queries, through DIFFERENT trees, and the leaf-major sweep above -- whose unit is (leaf, <= 4 queries) --
// fetches it from HBM once per tree that wants it (98 GB per batch against a 31 GB table).  Here the unit is the stored
// ROW: the table is streamed ONCE, in address order (sequential HBM reads, no gather), and each row is scored against
// every query that visits any of its T leaves; the queries (3 MB for 1024 x 768) come from L2.  Per (row, query) pair one
// operand has to reach the CU either way; here it is the query, from L2, instead of the row, from HBM.
//   rowLeaf[row][tree] = {leaf node of `row` in `tree`, position of the row inside that leaf}   (built once per forest)
//   per batch: leafCount[node] visits, groups[groupBase[node] + s / GRP].{b, key_off}[s % GRP] for visit s -- exactly what
//   the walk / expand kernels produce for the leaf-major sweep, so keys land in the same slots and select / final are
//   unchanged; the sums are the same canonical sums, so the keys are bit-identical.
// A wave takes RW consecutive rows (RW * T <= 256): phase 1 resolves their RW * T (row, tree) entries lane-parallel (which
// leaves are visited, by which query, where the key goes); phase 2 walks the rows, loads each once, and runs its pairs;
// the finished sums wait in lane registers (pair p in lane p mod 64) so that key_of and the stores run for 64 pairs at once.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void row_leaf_kernel(const int4 *__restrict__ node_pack, const uint32_t *__restrict__ node_tree,
                                                       const uint32_t *__restrict__ leaf_ids, uint32_t T, uint64_t n_rows,
                                                       uint32_t node0, uint2 *__restrict__ rowLeaf) {
    const uint32_t node = node0 + blockIdx.x;
    const int4 rec = node_pack[node];
    if (rec.x >= 0) return;  // inner node
    const uint32_t t = node_tree[node];
    if (t >= T) return;      // not reachable from a root
    const uint32_t off = (uint32_t)rec.y, len = (uint32_t)rec.z;
    for (uint32_t i = threadIdx.x; i < len; i += 64) {
        const uint32_t r = leaf_ids[(size_t)off + i];
        if (r < n_rows) rowLeaf[(size_t)r * T + t] = make_uint2(node, i);
    }
}
hipError_t zh_launch_row_leaf(const int4 *dNodePack, const uint32_t *dNodeTree, uint32_t n_nodes, const uint32_t *dLeafIds,
                              uint32_t T, uint64_t n_rows, uint2 *dRowLeaf, hipStream_t s) {
    hipError_t e = hipMemsetAsync(dRowLeaf, 0xFF, n_rows * T * sizeof(uint2), s);  // {-1, -1}: the row is not in that tree
    if (e != hipSuccess) return e;
    for (uint32_t n0 = 0; n0 < n_nodes; n0 += (1u << 22)) {  // block-indexed: slices well below 2^32 threads per launch
        const uint32_t nb = n_nodes - n0 < (1u << 22) ? n_nodes - n0 : (1u << 22);
        hipLaunchKernelGGL(row_leaf_kernel, dim3(nb), dim3(64), 0, s, dNodePack, dNodeTree, dLeafIds, T, n_rows, n0, dRowLeaf);
    }
    return hipGetLastError();
}

// Per batch, for the table scan: bit n of `bits` = leaf node n is visited by the batch (a 16-KB bitmap for the 130k nodes of
// cfg3: it stays in L1, and two thirds of a row's T entries are answered by it without an L2 request), and the visited
// leaves' {visits, first group, first visit's query and key slice} in ONE 16-byte record: a leaf with a single visit -- most of
// them -- needs no look-up in the group array.
__global__ __launch_bounds__(256) void node_visit_kernel(const uint32_t *__restrict__ leafCount, const uint32_t *__restrict__ groupBase,
                                                          const ZhGroup *__restrict__ groups, uint32_t n_nodes,
                                                          uint32_t *__restrict__ bits, uint4 *__restrict__ nodeVisit) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;  // one 32-node word per thread
    const uint32_t n0 = w * 32;
    if (n0 >= n_nodes) return;
    uint32_t word = 0;
    for (uint32_t i = 0; i < 32 && n0 + i < n_nodes; i++) {
        const uint32_t c = leafCount[n0 + i];
        if (c) {  // {visits | bits 32..35 of the first visit's key slice << 28, first group, first visit's query, its key slice (low 32)}
            word |= 1u << i;
            const uint32_t gb = groupBase[n0 + i];
            const uint64_t k0 = groups[gb].key_off[0];
            nodeVisit[n0 + i] = make_uint4(c | ((uint32_t)(k0 >> 32) << 28), gb, groups[gb].b[0], (uint32_t)k0);
        }
    }
    bits[w] = word;
}
hipError_t zh_launch_node_visits(const uint32_t *dLeafCount, const uint32_t *dGroupBase, const ZhGroup *dGroups, uint32_t n_nodes,
                                 uint32_t *dBits, uint4 *dNodeVisit, hipStream_t s) {
    const uint32_t words = (n_nodes + 31) / 32;
    if (!words) return hipSuccess;
    hipLaunchKernelGGL(node_visit_kernel, dim3((words + 255) / 256), dim3(256), 0, s, dLeafCount, dGroupBase, dGroups, n_nodes, dBits, dNodeVisit);
    return hipGetLastError();
}

// A stored row of the table scan: streamed exactly once per batch window.  ZH_SCAN_ROWPOL (build-time, A/B) picks the cache
// policy of these loads: 0 = global_load ... nt; n > 0 = buffer_load with aux bits n (1 = sc0, 2 = nt, 16 = sc1).
#ifndef ZH_SCAN_ROWPOL
#define ZH_SCAN_ROWPOL 0
#endif
template <int D>
__device__ __forceinline__ void load_row_stream(const float *__restrict__ row, uint32_t lane, float4 *v) {
#if ZH_SCAN_ROWPOL == 0
    load_row<D, true>(row, lane, v);
#else
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(row), 0, D * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < RowVec<D>::NV; j++) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, (lane + 64 * j) * 16, 0, ZH_SCAN_ROWPOL);  // out of range -> 0
        v[j] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
    }
#endif
}

#define ZH_SCAN_CAP 256  // pair records of a wave's LDS list (16 bytes each); a wave with more pairs takes the slow path
template <int D, int KIND>
__global__ __launch_bounds__(256) void scan_sweep_kernel(const float *__restrict__ X, const float *__restrict__ Q,
                                                          const float *__restrict__ QQ, const uint2 *__restrict__ rowLeaf,
                                                          uint32_t T, uint32_t RW, const uint32_t *__restrict__ visitBits,
                                                          const uint4 *__restrict__ nodeVisit,
                                                          const ZhGroup *__restrict__ groups, uint32_t GRP, uint64_t row_begin,
                                                          uint64_t row_end, int metric, int param,
                                                          uint64_t *__restrict__ keys, const uint32_t *__restrict__ run_if) {
    constexpr int NV = RowVec<D>::NV;
    __shared__ uint4 pair_list[4][ZH_SCAN_CAP];  // {row of the wave's RW, query, key slot lo, hi}
    if (run_if && *run_if == 0) return;  // the exact redo behind a half-width scan (zh_approx.hip): only when something ran over
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wid;
    const uint64_t r0 = row_begin + wave * RW;
    if (r0 >= row_end) return;
    const uint32_t nr = (uint32_t)(row_end - r0 < RW ? row_end - r0 : RW);
    const uint32_t n_ent = nr * T;
    const uint2 *__restrict__ ent = rowLeaf + (size_t)r0 * T;
    // ---- phase 1: the wave's nr * T (row, tree) entries, lane-parallel: visits of the entry's leaf -> pairs; the pairs of
    // the whole wave go to an LDS list in (row, tree, visit) order ----
    uint32_t eGb[ZH_SCAN_NE], eWithin[ZH_SCAN_NE], eC[ZH_SCAN_NE], off[ZH_SCAN_NE], eB0[ZH_SCAN_NE];
    uint64_t eK0[ZH_SCAN_NE];
    uint32_t P = 0;
#pragma unroll
    for (int j = 0; j < ZH_SCAN_NE; j++) {
        const uint32_t e = lane + 64u * j;
        eGb[j] = 0; eWithin[j] = 0; eC[j] = 0; eB0[j] = 0; eK0[j] = 0;
        if (e < n_ent) {
            const unsigned long long rlw = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(ent + e));  // streamed once
            const uint2 rl = make_uint2((uint32_t)rlw, (uint32_t)(rlw >> 32));
            eWithin[j] = rl.y;
            if (rl.x != 0xFFFFFFFFu && ((visitBits[rl.x >> 5] >> (rl.x & 31)) & 1u)) {
                const uint4 nv = nodeVisit[rl.x];
                eC[j] = nv.x & 0x0FFFFFFFu; eGb[j] = nv.y; eB0[j] = nv.z;
                eK0[j] = ((uint64_t)(nv.x >> 28) << 32) | nv.w;
            }
        }
        uint32_t incl = eC[j];  // inclusive scan over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o);
            if (lane >= (uint32_t)o) incl += t;
        }
        off[j] = P + incl - eC[j];
        P += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (P == 0) return;  // nobody wants any of these rows: they are not even loaded
    float my_s0 = 0.f, my_s1 = 0.f;  // finished sums wait here (pair n in lane n mod 64): key_of and the stores run 64 at a time
    uint64_t my_slot = 0;
    uint32_t my_b = 0, npend = 0;
    auto flush = [&]() {
        // keys are written once and read once, much later, by the select kernel: non-temporal, so that 8-byte stores to
        // 64 different lines do not push the queries out of L2
        // (the kind is a template parameter: for the two simsimd kinds only their own finaliser is compiled in, not key_of's switch
        // over every metric)
        if (lane < npend) {
            uint64_t key;
            if constexpr (KIND == K_COS) key = key_cosine(my_s0, my_s1, QQ[my_b], param);
            else if constexpr (KIND == K_L2) key = key_l2(my_s0, metric);
            else key = key_of(metric, param, my_s0, my_s1, 0.f);
            __builtin_nontemporal_store(key, keys + my_slot);
        }
        npend = 0;
    };
    auto row_norm = [&](const float4 *v) {  // cosine: the stored row's a2, once per row
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const bool act = (j < RowVec<D>::NJ) || (lane < (uint32_t)RowVec<D>::REM4);
            if (act) sq4(v[j], c);
        }
        return wave_sum_canonical((c.x + c.y) + (c.z + c.w));
    };
    auto score = [&](const float4 *v, float a2, uint32_t b, uint64_t slot, const float4 *q) {
        float s0 = 0.f, s1 = 0.f;
        row_pair_sums<D, KIND>(v, q, lane, param, s0, s1);
        if (lane == npend) { my_s0 = s0; my_s1 = KIND == K_COS ? a2 : s1; my_slot = slot; my_b = b; }
        if (++npend == 64) flush();
    };
    if (P <= ZH_SCAN_CAP) {
        uint4 *list = pair_list[wid];
        uint32_t mrows = 0;  // rows of the wave that have pairs
#pragma unroll
        for (int j = 0; j < ZH_SCAN_NE; j++) {
            const uint32_t c = eC[j];
            if (c) {
                const uint32_t rl = (lane + 64u * j) / T, gb = eGb[j];
                mrows |= 1u << rl;
                {   // the first visit comes with the node's record; further visits of a hot leaf from the group array
                    const uint64_t slot = eK0[j] + eWithin[j];
                    list[off[j]] = make_uint4(rl, eB0[j], (uint32_t)slot, (uint32_t)(slot >> 32));
                }
                for (uint32_t sidx = 1; sidx < c; sidx++) {
                    const ZhGroup *g = groups + gb + sidx / GRP;
                    const uint64_t slot = g->key_off[sidx % GRP] + eWithin[j];
                    list[off[j] + sidx] = make_uint4(rl, g->b[sidx % GRP], (uint32_t)slot, (uint32_t)(slot >> 32));
                }
            }
        }
        uint32_t rowmask = 0;
#pragma unroll
        for (int bit = 0; bit < 16; bit++) rowmask |= (__ballot((mrows >> bit) & 1u) != 0 ? 1u : 0u) << bit;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        auto rd = [&](uint32_t p) {
            const uint4 r = list[p];
            return make_uint4((uint32_t)__builtin_amdgcn_readfirstlane((int)r.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)r.y),
                              (uint32_t)__builtin_amdgcn_readfirstlane((int)r.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)r.w));
        };
        // ---- phase 2: rows with pairs FOUR at a time (one HBM round trip per four rows, as the generic kernel), and the pairs
        // of a row FOUR at a time: the four queries are requested together and scored as they arrive.  Round 2's loop had ONE
        // query on its way while the current pair was scored and copied it into place afterwards -- a wait for the younger load
        // at the end of every pair.  Measured (cfg3, window 2): 6.93 -> 6.83 ms per launch, +1.5 % -- the latency of the pair
        // loop is NOT what bounds this kernel (profiles/r03_ab_scan_pairs_in_flight.txt).  (A ring that refills a slot right
        // after its pair was scored keeps four queries in flight ALL the time, but its loads sit in conditional blocks -- row
        // switches, tails -- where the compiler's static count of outstanding loads collapses to "drain everything", or it
        // selects between register arrays through temporaries it then waits for: three formulations compiled to a drained
        // pipeline; this one needs nothing from the compiler but vmcnt(0).)
        float4 qa[NV], qb[NV], qc[NV], qd[NV];
        uint32_t p = 0;
        auto segment = [&](const float4 *vr, float a2, uint32_t rowid) {
            for (;;) {
                // the next four list entries (past the end: the last entry again) and which of them belong to this row
                const uint4 ra = rd(p < P ? p : P - 1), rb = rd(p + 1 < P ? p + 1 : P - 1), rc = rd(p + 2 < P ? p + 2 : P - 1),
                            rdd = rd(p + 3 < P ? p + 3 : P - 1);
                const bool ha = p < P && ra.x == rowid, hb = ha && p + 1 < P && rb.x == rowid, hc = hb && p + 2 < P && rc.x == rowid,
                           hd = hc && p + 3 < P && rdd.x == rowid;
                if (!ha) return;
                load_row<D>(Q + (size_t)ra.y * D, lane, qa);
                if (hb) load_row<D>(Q + (size_t)rb.y * D, lane, qb);
                if (hc) load_row<D>(Q + (size_t)rc.y * D, lane, qc);
                if (hd) load_row<D>(Q + (size_t)rdd.y * D, lane, qd);
                score(vr, a2, ra.y, ((uint64_t)ra.w << 32) | ra.z, qa);
                if (hb) score(vr, a2, rb.y, ((uint64_t)rb.w << 32) | rb.z, qb);
                if (hc) score(vr, a2, rc.y, ((uint64_t)rc.w << 32) | rc.z, qc);
                if (hd) score(vr, a2, rdd.y, ((uint64_t)rdd.w << 32) | rdd.z, qd);
                p += hd ? 4u : (hc ? 3u : (hb ? 2u : 1u));
                if (!hd) return;
            }
        };
        while (rowmask) {
            uint32_t rid[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                rid[r] = rowmask ? (uint32_t)__builtin_ctz(rowmask) : 0xFFFFFFFFu;
                rowmask &= rowmask - 1;  // (0 & anything stays 0)
            }
            float4 v0[NV], v1[NV], v2[NV], v3[NV];
            load_row_stream<D>(X + (size_t)(r0 + rid[0]) * D, lane, v0);
            if (rid[1] != 0xFFFFFFFFu) load_row_stream<D>(X + (size_t)(r0 + rid[1]) * D, lane, v1);
            if (rid[2] != 0xFFFFFFFFu) load_row_stream<D>(X + (size_t)(r0 + rid[2]) * D, lane, v2);
            if (rid[3] != 0xFFFFFFFFu) load_row_stream<D>(X + (size_t)(r0 + rid[3]) * D, lane, v3);
            segment(v0, KIND == K_COS ? row_norm(v0) : 0.f, rid[0]);
            if (rid[1] != 0xFFFFFFFFu) segment(v1, KIND == K_COS ? row_norm(v1) : 0.f, rid[1]);
            if (rid[2] != 0xFFFFFFFFu) segment(v2, KIND == K_COS ? row_norm(v2) : 0.f, rid[2]);
            if (rid[3] != 0xFFFFFFFFu) segment(v3, KIND == K_COS ? row_norm(v3) : 0.f, rid[3]);
        }
    } else {
        // more pairs than the list holds (hot leaves): entry after entry, no list
        float4 v[NV];
        float a2 = 0.f;
        uint32_t cur = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < ZH_SCAN_NE; j++) {
            unsigned long long m = __ballot(eC[j] != 0);
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)eC[j], l);
                const uint32_t gb = (uint32_t)__builtin_amdgcn_readlane((int)eGb[j], l);
                const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)eWithin[j], l);
                const uint32_t rl = ((uint32_t)l + 64u * j) / T;
                if (rl != cur) {
                    cur = rl;
                    load_row<D, true>(X + (size_t)(r0 + rl) * D, lane, v);
                    if (KIND == K_COS) a2 = row_norm(v);
                }
                for (uint32_t sidx = 0; sidx < c; sidx++) {
                    const ZhGroup *g = groups + gb + sidx / GRP;
                    const uint32_t b = g->b[sidx % GRP];
                    float4 q[NV];
                    load_row<D>(Q + (size_t)b * D, lane, q);
                    score(v, a2, b, g->key_off[sidx % GRP] + w, q);
                }
            }
        }
    }
    flush();
}

// ---- table scan at d = 128: TWO pairs per step.  A 512-byte row or query fills half a wave, and at 3-4 pairs per stored row
// the generic kernel above spends its time on per-pair instructions (one butterfly, one record fetch, one stash per pair on a
// half-empty wave).  Here lanes 0..31 work on pair p, lanes 32..63 on pair p + 1: pair records are read per LANE (no scalar
// round trip), one load instruction fetches both queries, one half-wave butterfly reduces both, and the rows of a sub-batch
// (4 at a time) wait in LDS so that each half can take the row its own pair names.  Same canonical sums (half_sum_canonical).
#define ZH_SCAN128_CAP 192
template <int KIND>
__global__ __launch_bounds__(256) void scan128_sweep_kernel(const float *__restrict__ X, const float *__restrict__ Q,
                                                             const float *__restrict__ QQ, const uint2 *__restrict__ rowLeaf,
                                                             uint32_t T, uint32_t RW, const uint32_t *__restrict__ visitBits,
                                                             const uint4 *__restrict__ nodeVisit,
                                                             const ZhGroup *__restrict__ groups, uint32_t GRP, uint64_t row_begin,
                                                             uint64_t row_end, int metric, int param,
                                                             uint64_t *__restrict__ keys, const uint32_t *__restrict__ run_if) {
    static_assert(KIND == K_L2 || KIND == K_COS, "the paired scan covers the two simsimd-path kinds");
    if (run_if && *run_if == 0) return;
    __shared__ uint4 pair_list[4][ZH_SCAN128_CAP];
    __shared__ float4 row_lds[4][4][32];
    __shared__ uint32_t row_start[4][20];
    const uint32_t lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wid;
    const uint64_t r0 = row_begin + wave * RW;
    if (r0 >= row_end) return;
    const uint32_t nr = (uint32_t)(row_end - r0 < RW ? row_end - r0 : RW);
    const uint32_t n_ent = nr * T;
    const uint2 *__restrict__ ent = rowLeaf + (size_t)r0 * T;
    const float4 *__restrict__ X4 = reinterpret_cast<const float4 *>(X);
    const float4 *__restrict__ Q4 = reinterpret_cast<const float4 *>(Q);
    // ---- phase 1 (as scan_sweep_kernel): entries -> pairs, plus where every row's pairs start in the list ----
    uint32_t eGb[ZH_SCAN_NE], eWithin[ZH_SCAN_NE], eC[ZH_SCAN_NE], off[ZH_SCAN_NE], eB0[ZH_SCAN_NE];
    uint64_t eK0[ZH_SCAN_NE];
    uint32_t P = 0;
#pragma unroll
    for (int j = 0; j < ZH_SCAN_NE; j++) {
        const uint32_t e = lane + 64u * j;
        eGb[j] = 0; eWithin[j] = 0; eC[j] = 0; eB0[j] = 0; eK0[j] = 0;
        if (e < n_ent) {
            const unsigned long long rlw = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(ent + e));
            const uint2 rl = make_uint2((uint32_t)rlw, (uint32_t)(rlw >> 32));
            eWithin[j] = rl.y;
            if (rl.x != 0xFFFFFFFFu && ((visitBits[rl.x >> 5] >> (rl.x & 31)) & 1u)) {
                const uint4 nv = nodeVisit[rl.x];
                eC[j] = nv.x & 0x0FFFFFFFu; eGb[j] = nv.y; eB0[j] = nv.z;
                eK0[j] = ((uint64_t)(nv.x >> 28) << 32) | nv.w;
            }
        }
        uint32_t incl = eC[j];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o);
            if (lane >= (uint32_t)o) incl += t;
        }
        off[j] = P + incl - eC[j];
        P += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (P == 0) return;
    float my_s0 = 0.f, my_s1 = 0.f;
    uint64_t my_slot = ~0ull;
    uint32_t my_b = 0;
    if (P <= ZH_SCAN128_CAP) {
        uint4 *list = pair_list[wid];
        uint32_t *rstart = row_start[wid];
#pragma unroll
        for (int j = 0; j < ZH_SCAN_NE; j++) {
            const uint32_t e = lane + 64u * j, c = eC[j];
            if (e < n_ent && e % T == 0) rstart[e / T] = off[j];  // the first entry of a row: its pairs start here
            if (c) {
                const uint32_t rl = e / T, gb = eGb[j];
                {   // the first visit comes with the node's record; further visits of a hot leaf from the group array
                    const uint64_t slot = eK0[j] + eWithin[j];
                    list[off[j]] = make_uint4(rl, eB0[j], (uint32_t)slot, (uint32_t)(slot >> 32));
                }
                for (uint32_t sidx = 1; sidx < c; sidx++) {
                    const ZhGroup *g = groups + gb + sidx / GRP;
                    const uint64_t slot = g->key_off[sidx % GRP] + eWithin[j];
                    list[off[j] + sidx] = make_uint4(rl, g->b[sidx % GRP], (uint32_t)slot, (uint32_t)(slot >> 32));
                }
            }
        }
        if (lane == 0) rstart[nr] = P;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t npl = 0;  // pairs waiting in each half (pair n of a half in its lane n)
        auto flush = [&]() {
            if (hl < npl && my_slot != ~0ull)
                __builtin_nontemporal_store(key_of(metric, param, my_s0, my_s1, KIND == K_COS ? QQ[my_b] : 0.f), keys + my_slot);
            npl = 0;
        };
        for (uint32_t rb = 0; rb < nr; rb += 4) {
            const uint32_t re = rb + 4 < nr ? rb + 4 : nr;
            const uint32_t pa = (uint32_t)__builtin_amdgcn_readfirstlane((int)rstart[rb]);
            const uint32_t pb = (uint32_t)__builtin_amdgcn_readfirstlane((int)rstart[re]);
            if (pa == pb) continue;  // nobody wants these four rows
            {   // the sub-batch's rows into LDS: two rows per load instruction
                const uint32_t ra = rb + half, rc = rb + 2 + half;
                float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vc = va;
                if (ra < re) va = ld16<true>(X4 + (size_t)(r0 + ra) * 32 + hl);
                if (rc < re) vc = ld16<true>(X4 + (size_t)(r0 + rc) * 32 + hl);
                row_lds[wid][half][hl] = va;
                row_lds[wid][2 + half][hl] = vc;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // lower half: pairs pa, pa + 2, ...; upper half: pa + 1, pa + 3, ...
            uint32_t pi = pa + half;
            uint4 rec = list[pi < pb ? pi : pb - 1];
            float4 qc = Q4[(size_t)rec.y * 32 + hl];
            for (uint32_t p = pa; p < pb; p += 2) {
                const bool valid = p + half < pb;
                const uint32_t pn = p + 2 + half;
                const uint4 recn = list[pn < pb ? pn : pb - 1];
                const float4 qn = Q4[(size_t)recn.y * 32 + hl];  // the next step's queries fly while this step is scored
                const float4 v = row_lds[wid][rec.x - rb][hl];
                float a2 = 0.f;
                if (KIND == K_COS) {
                    float4 c;
                    c.x = __builtin_fmaf(v.x, v.x, 0.f); c.y = __builtin_fmaf(v.y, v.y, 0.f);
                    c.z = __builtin_fmaf(v.z, v.z, 0.f); c.w = __builtin_fmaf(v.w, v.w, 0.f);
                    a2 = half_sum_canonical((c.x + c.y) + (c.z + c.w));
                }
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f), e1 = a;
                acc4<KIND>(v, qc, a, e1, param);
                const float s0 = half_sum_canonical((a.x + a.y) + (a.z + a.w));
                if (hl == npl) {
                    my_s0 = s0; my_s1 = a2; my_b = rec.y;
                    my_slot = valid ? (((uint64_t)rec.w << 32) | rec.z) : ~0ull;
                }
                if (++npl == 32) flush();
                rec = recn; qc = qn;
            }
            __builtin_amdgcn_wave_barrier();  // the next sub-batch overwrites row_lds
        }
        flush();
        return;
    }
    // more pairs than the list holds (hot leaves): entry after entry on the whole wave, as scan_sweep_kernel's slow path
    uint32_t npend = 0;
    auto flush1 = [&]() {
        if (lane < npend) __builtin_nontemporal_store(key_of(metric, param, my_s0, my_s1, KIND == K_COS ? QQ[my_b] : 0.f), keys + my_slot);
        npend = 0;
    };
    float4 v[1];
    float a2 = 0.f;
    uint32_t cur = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < ZH_SCAN_NE; j++) {
        unsigned long long m = __ballot(eC[j] != 0);
        while (m) {
            const int l = __builtin_ctzll(m);
            m &= m - 1;
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)eC[j], l);
            const uint32_t gb = (uint32_t)__builtin_amdgcn_readlane((int)eGb[j], l);
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)eWithin[j], l);
            const uint32_t rl = ((uint32_t)l + 64u * j) / T;
            if (rl != cur) {
                cur = rl;
                load_row<128, true>(X + (size_t)(r0 + rl) * 128, lane, v);
                if (KIND == K_COS) {
                    float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (lane < 32) {
                        c4.x = __builtin_fmaf(v[0].x, v[0].x, 0.f); c4.y = __builtin_fmaf(v[0].y, v[0].y, 0.f);
                        c4.z = __builtin_fmaf(v[0].z, v[0].z, 0.f); c4.w = __builtin_fmaf(v[0].w, v[0].w, 0.f);
                    }
                    a2 = wave_sum_canonical((c4.x + c4.y) + (c4.z + c4.w));
                }
            }
            for (uint32_t sidx = 0; sidx < c; sidx++) {
                const ZhGroup *g = groups + gb + sidx / GRP;
                const uint32_t b = g->b[sidx % GRP];
                float4 q[1];
                load_row<128>(Q + (size_t)b * 128, lane, q);
                float s0 = 0.f, s1 = 0.f;
                row_pair_sums<128, KIND>(v, q, lane, param, s0, s1);
                if (lane == npend) { my_s0 = s0; my_s1 = KIND == K_COS ? a2 : s1; my_slot = g->key_off[sidx % GRP] + w; my_b = b; }
                if (++npend == 64) flush1();
            }
        }
    }
    flush1();
}

// rows per wave of the table scan for T trees (0: T is beyond what a wave's entry registers hold -> leaf-major sweep)
uint32_t zh_scan_rows_per_wave(uint32_t T) {
    if (T == 0 || T > 64u * ZH_SCAN_NE) return 0;
    const uint32_t rw = 64u * ZH_SCAN_NE / T;
    return rw > 16 ? 16 : rw;
}
// dimensions as the leaf-major sweep specialises them: the two simsimd-path kinds everywhere, the others at 128 / 384 / 768
bool zh_scan_sweep_supported(uint32_t d, uint32_t T, int metric) {
    if (!zh_scan_rows_per_wave(T)) return false;
    const int kind = zh_kind_of(metric);
    if (d == 128 || d == 384 || d == 768) return true;
    if (kind != K_L2 && kind != K_COS) return false;
    switch (d) { case 64: case 256: case 512: case 1024: case 1536: return true; default: return false; }
}

template <int D, int KIND>
static hipError_t launch_scan_dk(const float *dX, uint64_t n_rows, const float *dQ, const float *dQQ, const uint2 *dRowLeaf,
                                 uint32_t T, const uint32_t *dVisitBits, const uint4 *dNodeVisit, const ZhGroup *dGroups,
                                 uint32_t group, int metric, int param, uint64_t *dKeys, const uint32_t *dRunIf, hipStream_t s) {
    const uint32_t RW = zh_scan_rows_per_wave(T);
    uint64_t rows_per_launch = zh_sweep_rows_per_launch(D);
    rows_per_launch = rows_per_launch / (4 * RW) * (4 * RW);
    for (uint64_t r = 0; r < n_rows; r += rows_per_launch) {
        const uint64_t r_end = r + rows_per_launch < n_rows ? r + rows_per_launch : n_rows;
        const uint64_t waves = (r_end - r + RW - 1) / RW, blocks = (waves + 3) / 4;
        if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
        if constexpr (D == 128 && (KIND == K_L2 || KIND == K_COS)) {
            static const bool paired = getenv("ZH_SCAN128_GENERIC") == nullptr;  // A/B: the generic kernel at d = 128
            if (paired) {
                hipLaunchKernelGGL((scan128_sweep_kernel<KIND>), dim3((uint32_t)blocks), dim3(256), 0, s, dX, dQ, dQQ, dRowLeaf, T, RW,
                                   dVisitBits, dNodeVisit, dGroups, group, r, r_end, metric, param, dKeys, dRunIf);
                continue;
            }
        }
        hipLaunchKernelGGL((scan_sweep_kernel<D, KIND>), dim3((uint32_t)blocks), dim3(256), 0, s, dX, dQ, dQQ, dRowLeaf, T, RW,
                           dVisitBits, dNodeVisit, dGroups, group, r, r_end, metric, param, dKeys, dRunIf);
    }
    return hipGetLastError();
}
template <int KIND>
static hipError_t launch_scan_k(const float *dX, uint32_t d, uint64_t n_rows, const float *dQ, const float *dQQ,
                                const uint2 *dRowLeaf, uint32_t T, const uint32_t *dVisitBits, const uint4 *dNodeVisit,
                                const ZhGroup *dGroups, uint32_t group, int metric, int param, uint64_t *dKeys, const uint32_t *dRunIf, hipStream_t s) {
#define ZH_SCAN_CASE(DD) \
    case DD: return launch_scan_dk<DD, KIND>(dX, n_rows, dQ, dQQ, dRowLeaf, T, dVisitBits, dNodeVisit, dGroups, group, metric, param, dKeys, dRunIf, s)
    if constexpr (KIND == K_L2 || KIND == K_COS) {
        switch (d) {
            ZH_SCAN_CASE(64); ZH_SCAN_CASE(256); ZH_SCAN_CASE(512); ZH_SCAN_CASE(1024); ZH_SCAN_CASE(1536);
        default: break;
        }
    }
    switch (d) {
        ZH_SCAN_CASE(128); ZH_SCAN_CASE(384); ZH_SCAN_CASE(768);
    default: return hipErrorInvalidValue;
    }
#undef ZH_SCAN_CASE
}
hipError_t zh_launch_scan_sweep(const float *dX, uint32_t d, uint64_t n_rows, const float *dQ, const float *dQQ,
                                const uint2 *dRowLeaf, uint32_t T, const uint32_t *dVisitBits, const uint4 *dNodeVisit,
                                const ZhGroup *dGroups, uint32_t group, int metric, int param, uint64_t *dKeys, const uint32_t *dRunIf,
                                hipStream_t s) {
    if (!n_rows) return hipSuccess;
#define ZH_KIND_CASE(K) \
    case K: return launch_scan_k<K>(dX, d, n_rows, dQ, dQQ, dRowLeaf, T, dVisitBits, dNodeVisit, dGroups, group, metric, param, dKeys, dRunIf, s)
    switch (zh_kind_of(metric)) {
        ZH_KIND_CASE(K_COS);
        ZH_KIND_CASE(K_MAX);
        ZH_KIND_CASE(K_CANB);
        ZH_KIND_CASE(K_BRAY);
        ZH_KIND_CASE(K_ABS);
        ZH_KIND_CASE(K_P3);
        ZH_KIND_CASE(K_P4);
        ZH_KIND_CASE(K_HAMM);
        ZH_KIND_CASE(K_PP);
    default: return launch_scan_k<K_L2>(dX, d, n_rows, dQ, dQQ, dRowLeaf, T, dVisitBits, dNodeVisit, dGroups, group, metric, param, dKeys, dRunIf, s);
    }
#undef ZH_KIND_CASE
}

// n contiguous rows against one query: a single synthetic group, ids = row numbers
__global__ void one_group_kernel(ZhGroup *g, uint64_t *rowoff, uint64_t n) {
    g->leaf_off = 0; g->len = (uint32_t)n; g->gsize = 1; g->take4 = 0;
    for (int m = 0; m < ZH_GROUP_MAX; m++) { g->b[m] = 0; g->key_off[m] = 0; }
    rowoff[0] = 0;
}
// scratch: one ZhGroup, one u64 row offset and one float (the query's norm), caller-owned device memory
hipError_t zh_launch_distance_rows(const float *dX, uint64_t n, uint32_t d, const float *dq, int metric, int mode,
                                   uint64_t *dKeys, void *dScratch, hipStream_t s) {
    if (!n) return hipSuccess;
    ZhGroup *dg = reinterpret_cast<ZhGroup *>(dScratch);
    uint64_t *dro = reinterpret_cast<uint64_t *>(dg + 1);
    float *dqq = reinterpret_cast<float *>(dro + 1);
    hipLaunchKernelGGL(one_group_kernel, dim3(1), dim3(1), 0, s, dg, dro, n);
    hipError_t e = zh_launch_qnorm(dq, 1, d, dqq, s);
    if (e == hipSuccess) e = zh_launch_sweep(dX, d, dq, dqq, dg, dro, 1, nullptr, nullptr, n, metric, mode, dKeys, 2, s);
    return e;
}

// A block owns `chunk` consecutive visits (1 when visits are few and long, up to 256 when the walk produced millions
// of tiny ones -- small-leaf forests): thread t first looks at visit t; a short leaf taken whole is copied by that
// thread on the spot, everything else queues for the block's partition code.  (One block per visit would also run into
// the 2^32-threads-per-launch limit at 2^24 visits.)
#define SEL_GROUP_LEN 16
// CAP = entries of the LDS key buffer: 4096 in general; 1024 when no leaf of the forest is longer (16 KB instead of
// 48 KB of LDS per block: 8 blocks per CU instead of 3)
template <int CAP>
__global__ __launch_bounds__(256) void select_kernel(const ZhVisit *__restrict__ visits, uint64_t n_visits, uint32_t chunk,
                                                      const uint32_t *__restrict__ leaf_ids,
                                                      const uint64_t *__restrict__ keys,
                                                      uint64_t *__restrict__ cand_keys,
                                                      uint32_t *__restrict__ cand_ids, const uint32_t *__restrict__ run_if) {
    __shared__ uint64_t sk[CAP];
    __shared__ __attribute__((aligned(16))) uint32_t si[CAP < 2048 ? 2048 : CAP];  // slow path ids; fast path: small-sort buffers + histogram (1808 words)
    __shared__ uint32_t s_u32[8];
    __shared__ uint32_t s_list[256], s_small[256], s_nlist, s_nsmall;
    const uint32_t tid = threadIdx.x;
    if (run_if && *run_if == 0) return;
    const uint64_t base = (uint64_t)blockIdx.x * chunk;
    const uint32_t cnt = (uint32_t)(n_visits - base < chunk ? n_visits - base : chunk);
    if (tid == 0) { s_nlist = 0; s_nsmall = 0; }
    __syncthreads();
    if (tid < cnt) {
        const uint32_t len = visits[base + tid].len, take = visits[base + tid].take;
        if (take > 0) {
            if (len <= SEL_GROUP_LEN) s_small[atomicAdd(&s_nsmall, 1u)] = tid;
            else s_list[atomicAdd(&s_nlist, 1u)] = tid;
        }
    }
    __syncthreads();
    // leaves of at most 16 rows: one 16-lane group per visit, a lane per row; a row's rank among the visit's (key, id)
    // pairs says whether it is one of the `take` smallest (and where it goes)
    {
        const uint32_t ns = s_nsmall, g = tid >> 4, l = tid & 15;
        for (uint32_t i = g; i < ns; i += 16) {
            const ZhVisit v = visits[base + s_small[i]];
            const bool on = l < v.len;
            const uint64_t k = on ? keys[v.row_off + l] : ~0ull;
            const uint32_t id = on ? leaf_ids[(size_t)v.leaf_off + l] : ~0u;
            uint32_t rank = l;
            if (v.take < v.len) {
                rank = 0;
#pragma unroll
                for (int j = 0; j < SEL_GROUP_LEN; j++) {
                    const uint64_t kj = __shfl(k, j, SEL_GROUP_LEN);
                    const uint32_t ij = __shfl(id, j, SEL_GROUP_LEN);
                    rank += (kj < k || (kj == k && ij < id)) ? 1u : 0u;
                }
            }
            if (on && rank < v.take) {
                cand_keys[v.cand_off + rank] = k;
                cand_ids[v.cand_off + rank] = id;
            }
        }
    }
    const uint32_t nl = s_nlist;
    for (uint32_t li = 0; li < nl; li++) {  // block-uniform
        const ZhVisit v = visits[base + s_list[li]];
        if (v.take >= v.len) {
            for (uint32_t i = tid; i < v.len; i += 256) {
                cand_keys[v.cand_off + i] = keys[v.row_off + i];
                cand_ids[v.cand_off + i] = leaf_ids[(size_t)v.leaf_off + i];
            }
            continue;
        }
        bool need_slow = false;
        if (v.len <= CAP) select_fast<true>(v, leaf_ids, keys, cand_keys, cand_ids, sk, si, s_u32, need_slow);
        else select_fast<false>(v, leaf_ids, keys, cand_keys, cand_ids, sk, si, s_u32, need_slow);
        if (need_slow) {  // a large group of equal keys straddles the cut: order it by id the slow way
            __syncthreads();
            select_slow(v, leaf_ids, keys, cand_keys, cand_ids, sk, si, CAP);
        }
        __syncthreads();  // the LDS buffers are reused by the next visit
    }
}

hipError_t zh_launch_select(const ZhVisit *dVisits, uint64_t n_visits, const uint32_t *dLeafIds,
                            const uint64_t *dKeys, uint64_t *dCandKeys, uint32_t *dCandIds, uint32_t max_leaf_len,
                            const uint32_t *dRunIf, hipStream_t s) {
    if (!n_visits) return hipSuccess;
    if (n_visits > 0x7FFFFFFFull) return hipErrorInvalidValue;
    uint64_t chunk = (n_visits + 16383) / 16384;  // >= 16k blocks before a block takes a second visit
    if (chunk > 256) chunk = 256;
    const uint64_t blocks = (n_visits + chunk - 1) / chunk;  // <= 2^23: 2^31 threads
    if (max_leaf_len <= 1024)
        hipLaunchKernelGGL(select_kernel<1024>, dim3((uint32_t)blocks), dim3(256), 0, s, dVisits, n_visits, (uint32_t)chunk,
                           dLeafIds, dKeys, dCandKeys, dCandIds, dRunIf);
    else if (getenv("ZH_SELECT_4096") != nullptr)  // A/B: the round-1 LDS footprint (48 KB per block)
        hipLaunchKernelGGL(select_kernel<ZH_SORT_N>, dim3((uint32_t)blocks), dim3(256), 0, s, dVisits, n_visits, (uint32_t)chunk,
                           dLeafIds, dKeys, dCandKeys, dCandIds, dRunIf);
    else  // Leaves longer than the LDS buffer re-read their keys from the L2-resident scratch per histogram round.  Small buffers win:
          // 2048 entries (24 KB per block, six blocks per CU) against 4096: 0.33 / 0.46 ms alone at cfg3, 0.4 / 3.8 ms beside the sweep;
          // a variant that held leaves of up to 8192 rows in 96 KB -- one block per CU -- 4.6 against 2.7 ms per cfg5 batch alone and
          // 142 k against 150 k QPS beside the sweep (profiles/r02_ab_select_lds.txt)
        hipLaunchKernelGGL(select_kernel<2048>, dim3((uint32_t)blocks), dim3(256), 0, s, dVisits, n_visits, (uint32_t)chunk,
                           dLeafIds, dKeys, dCandKeys, dCandIds, dRunIf);
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
                                                     uint32_t *__restrict__ out_counts, uint64_t stride64,
                                                     uint64_t stride32, uint32_t L, const uint32_t *__restrict__ run_if) {
    if (run_if && *run_if == 0) return;
    // MERGE: every source list has L slots per query (L = k for shard results; the prefilter's per-tree lists are longer)
    __shared__ uint64_t sk[FIN_SORT_N];
    __shared__ uint64_t si[FIN_SORT_N];
    __shared__ uint64_t rk[ZH_MAX_TOPK];
    __shared__ uint64_t ri[ZH_MAX_TOPK];
    __shared__ uint32_t scan[256];
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    uint64_t n_src;  // entries in the source stream (MERGE: S*k slots, some invalid)
    uint64_t c0 = 0;
    if (MERGE) n_src = (uint64_t)T * L;  // T carries the shard count here
    else { c0 = candBase[(uint64_t)b * T]; n_src = candBase[(uint64_t)(b + 1) * T] - c0; }
    __shared__ uint32_t s_fill;
    uint32_t have = 0;
    uint64_t pos = 0;
    bool first = true;
    // Streaming top-k: the sort buffer holds the k best so far plus new entries.  Once k distinct entries are held, an
    // entry that does not beat the k-th (key, id) can never be output, so the stream is filtered on the way in and
    // the buffer is sorted only when 1024 survivors have gathered (a wandering walk hands over ~10^5 candidates per
    // query, nearly all of them beyond the k-th best after the first round).
    constexpr uint32_t TILE = 1024;
    while (pos < n_src || first) {
        first = false;
        for (uint32_t i = tid; i < have; i += 256) { sk[i] = rk[i]; si[i] = ri[i]; }
        const bool filt = k > 0 && have == k;
        const uint64_t tk = filt ? rk[k - 1] : 0, ti = filt ? ri[k - 1] : 0;
        if (tid == 0) s_fill = have;
        __syncthreads();
        for (;;) {
            const uint32_t fill = s_fill;  // block-uniform
            if (pos >= n_src || FIN_SORT_N - fill < TILE) break;
            __syncthreads();  // everyone has read s_fill before anyone appends
#pragma unroll
            for (uint32_t u = 0; u < TILE / 256; u++) {
                const uint64_t e = pos + u * 256 + tid;
                if (e < n_src) {
                    uint64_t key, id;
                    if (MERGE) {
                        uint32_t s = (uint32_t)(e / L), j = (uint32_t)(e % L);
                        bool valid = j < shard_counts[(size_t)s * stride32 + b];  // shard s starts stride32 counts further
                        size_t src = (size_t)s * stride64 + (size_t)b * L + j;        // ... and stride64 ids / keys further
                        key = valid ? cand_keys[src] : ~0ull;
                        id = valid ? cand_ids64[src] : ~0ull;
                    } else {
                        key = cand_keys[c0 + e];
                        id = id_base + cand_ids32[c0 + e];
                    }
                    if (!filt || key < tk || (key == tk && id < ti)) {
                        const uint32_t slot = atomicAdd(&s_fill, 1u);
                        sk[slot] = key; si[slot] = id;
                    }
                }
            }
            pos = n_src - pos < TILE ? n_src : pos + TILE;
            __syncthreads();
        }
        if (filt && s_fill == have) continue;  // nothing new beat the k-th best: rk / ri stand
        uint32_t total = s_fill, np2 = next_pow2(total);
        __syncthreads();
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
    }
    for (uint32_t i = tid; i < k; i += 256) {
        out_ids[(size_t)b * k + i] = i < have ? ri[i] : ~0ull;
        out_keys[(size_t)b * k + i] = i < have ? rk[i] : ~0ull;
    }
    if (tid == 0) out_counts[b] = have;
}

hipError_t zh_launch_final(const uint64_t *dCandBase, uint32_t B, uint32_t T, uint32_t k, const uint64_t *dCandKeys,
                           const uint32_t *dCandIds, uint64_t id_base, uint64_t *dOutIds, uint64_t *dOutKeys,
                           uint32_t *dOutCounts, const uint32_t *dRunIf, hipStream_t s) {
    if (!B) return hipSuccess;
    hipLaunchKernelGGL(final_kernel<false>, dim3(B), dim3(256), 0, s, dCandBase, B, T, k, dCandKeys, dCandIds,
                       (const uint64_t *)nullptr, (const uint32_t *)nullptr, id_base, dOutIds, dOutKeys, dOutCounts,
                       (uint64_t)0, (uint64_t)0, k, dRunIf);
    return hipGetLastError();
}

// Shard merge for the usual sizes (S * k <= 1024 entries per query): ONE WAVE per query sorts the S lists in LDS
// (bitonic by (key, id), 16 KB) and keeps the k smallest distinct ids.  The 256-thread final_kernel<MERGE> does the same
// for longer lists, but its 50 KB of LDS per block made a 1024-query merge occupy the whole chip next to the sweep.
#define MERGE_WAVE_N 1024
__global__ __launch_bounds__(64) void merge_wave_kernel(uint32_t B, uint32_t S, uint32_t k, const uint64_t *__restrict__ keys,
                                                         const uint64_t *__restrict__ ids, const uint32_t *__restrict__ counts,
                                                         uint64_t *__restrict__ out_ids, uint64_t *__restrict__ out_keys,
                                                         uint32_t *__restrict__ out_counts, uint64_t stride64, uint64_t stride32,
                                                         uint32_t L, uint32_t *__restrict__ over) {
    // every source list has L slots per query (L = k for shard results; the prefilter's per-tree lists are longer and mostly
    // empty): the valid entries are packed, so the sort is as long as what is there
    __shared__ uint64_t sk[MERGE_WAVE_N], si[MERGE_WAVE_N];
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    uint32_t n = 0;
    for (uint32_t sh = 0; sh < S; sh++) {  // wave-uniform
        const uint32_t c0 = counts[(size_t)sh * stride32 + b], c = c0 < L ? c0 : L;
        if (n + c > MERGE_WAVE_N) {  // only with S * L > MERGE_WAVE_N slots (the prefilter's lists, launched with `over`): reported, the batch is redone
            if (lane == 0 && over) atomicOr(over, 16u);
            break;
        }
        for (uint32_t j = lane; j < c; j += 64) {
            const size_t src = (size_t)sh * stride64 + (size_t)b * L + j;
            sk[n + j] = keys[src]; si[n + j] = ids[src];
        }
        n += c;
    }
    const uint32_t np2 = next_pow2(n);
    for (uint32_t e = n + lane; e < np2; e += 64) { sk[e] = ~0ull; si[e] = ~0ull; }
    __syncthreads();
    for (uint32_t size = 2; size <= np2; size <<= 1)
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = lane; t < np2 / 2; t += 64) {
                const uint32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t ka = sk[lo], kb = sk[hi], ia = si[lo], ib = si[hi];
                const bool gt = ka > kb || (ka == kb && ia > ib);
                if (gt == up) { sk[lo] = kb; sk[hi] = ka; si[lo] = ib; si[hi] = ia; }
            }
            __syncthreads();
        }
    // first k distinct ids of the sorted run (invalid slots, all ones, sort last)
    uint32_t have = 0;
    for (uint32_t base = 0; base < n && have < k; base += 64) {  // wave-uniform
        const uint32_t e = base + lane;
        const bool keep = e < n && !(sk[e] == ~0ull && si[e] == ~0ull) && (e == 0 || si[e] != si[e - 1]);
        const unsigned long long m = __ballot(keep);
        const uint32_t pos = have + (uint32_t)__popcll(m & ((1ull << lane) - 1));
        if (keep && pos < k) { out_ids[(size_t)b * k + pos] = si[e]; out_keys[(size_t)b * k + pos] = sk[e]; }
        have += (uint32_t)__popcll(m);
    }
    if (have > k) have = k;
    for (uint32_t i = have + lane; i < k; i += 64) { out_ids[(size_t)b * k + i] = ~0ull; out_keys[(size_t)b * k + i] = ~0ull; }
    if (lane == 0) out_counts[b] = have;
}

hipError_t zh_launch_merge(uint32_t S, uint32_t B, uint32_t k, const uint64_t *dIds, const uint64_t *dKeys,
                           const uint32_t *dCounts, uint64_t *dOutIds, uint64_t *dOutKeys, uint32_t *dOutCounts,
                           uint64_t stride64, uint64_t stride32, hipStream_t s) {
    if (!B) return hipSuccess;
    if ((uint64_t)S * k <= MERGE_WAVE_N) {
        hipLaunchKernelGGL(merge_wave_kernel, dim3(B), dim3(64), 0, s, B, S, k, dKeys, dIds, dCounts, dOutIds, dOutKeys, dOutCounts,
                           stride64 ? stride64 : (uint64_t)B * k, stride32 ? stride32 : (uint64_t)B, k, (uint32_t *)nullptr);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(final_kernel<true>, dim3(B), dim3(256), 0, s, (const uint64_t *)nullptr, B, S, k, dKeys,
                       (const uint32_t *)nullptr, dIds, dCounts, (uint64_t)0, dOutIds, dOutKeys, dOutCounts,
                       stride64 ? stride64 : (uint64_t)B * k, stride32 ? stride32 : (uint64_t)B, k, (const uint32_t *)nullptr);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Prefilter: which rows can be among a pair's k best is decided from the ROW SCORES the hash already computed.
//
// A batch hashed from row scores (zh_score.hip) holds S[row][q] = row . q for every stored row.  With the stored |r|^2/2 and
// |r| that is every L2^2 / cosine distance of the batch up to rounding -- and the wandering walk of a small-leaf forest asks for
// ~10^5 of them per query only to keep k.  So instead of sweeping the visited leaves (4*d bytes per scored row), a wave per
// (query, tree) pair replays the pair's visits on the scores (one 64-byte sector per scored row):
//   * value v and half-width e per row such that the key the reference computes lies in [v - e, v + e] in the value's scale
//     (derivation at zh_prefilter_bound, rigorous in the same sense as the row-score hash);
//   * a visit takes its `take` nearest rows (lsh.rs:300-330): decided by the values when the take-th and (take+1)-th intervals are
//     disjoint, otherwise the visit is AMBIGUOUS and goes to the exact path (prefilter_amb_kernel: the leaf scored with the
//     reference's arithmetic, as the sweep + select would);
//   * of the rows taken, only those whose lower bound does not exceed the k-th smallest upper bound seen so far can be among
//     the pair's k best -- a few more than k survive.
// The survivors (and the ambiguous visits' rows) are then scored exactly (prefilter_keys_kernel, the canonical sums) and the
// usual final top-k runs over num_trees short lists per query.  Results are bit-identical to the sweep's by construction; a list
// or the ambiguous-visit table running over is reported to the host, which redoes the batch the classic way.
// ------------------------------------------------------------------------------------------------
#define PF_MAXLEN 8      // rows per leaf the per-lane selection handles (longer leaves: not prefiltered, zh_api.hip)
#define PF_BUF 512       // survivors buffered per pair between compactions


// KINDA 0: L2 family, v = |r|^2/2 - r.q (= (distance^2 - |q|^2) / 2: same order per query); 1: cosine distance; 2: 1 - cosine
// distance (ZH_COSINE_PARITY keys) in the order of the key's bits.  false: nothing certain about this row (zero / tiny / infinite norms, NaN scores)
template <int KINDA>
__device__ __forceinline__ bool pf_value(float h, float nx, float s, float nq, float Kb, float &v, float &e) {
    if (!(s - s == 0.f) || !(nx - nx == 0.f) || !(nq - nq == 0.f)) return false;  // an overflowed dot product or norm: the two summation orders need not overflow alike
    if (KINDA == 0) {
        const float nn = nx + nq;
        if (!(nn > 1e-12f)) return false;
        v = h - s;
        e = Kb * nn * nn;
    } else {
        if (!(nx > 1e-12f && nq > 1e-12f)) return false;  // simsimd's zero-norm cases, and norms whose squares left the normal range
        float r = 1.0f - s / (nx * nq);
        r = r > 0.f ? r : 0.f;
        e = Kb;
        if (KINDA == 2) {
            // keys compare as the BITS of the f64 (distance.rs:23-25 as the oracle restates it): 1 - distance < 0 -- an obtuse angle --
            // sorts after every non-negative key, larger magnitudes later.  v follows that order (2 + |key| for a negative key);
            // a key within the bound of zero could be on either side: not certain
            const float key = 1.0f - r;
            if (!(fabsf(key) > e)) return false;
            v = key > 0.f ? key : 2.0f - key;
        } else
            v = r;
    }
    return (v - v == 0.f) && (e - e == 0.f);
}

template <int KINDA>
__global__ __launch_bounds__(64) void prefilter_kernel(ZhForestDev f, uint32_t T, uint32_t B, uint32_t k,
                                                        const ZhPairCounts *__restrict__ counts, const ZhVisit *__restrict__ inl,
                                                        ZhWalkLog wlog, ZhPrefilter pf, float Kb) {
    __shared__ uint64_t bk[PF_BUF];  // sortable upper bound << 32 | sortable lower bound
    __shared__ uint32_t br[PF_BUF];  // the row
    const uint32_t lane = threadIdx.x;
    const uint64_t pair = blockIdx.x;
    const uint32_t b = (uint32_t)(pair / T), t = (uint32_t)(pair % T);
    const size_t lp = (size_t)t * B + b;
    const uint32_t nv = counts[pair].visits;
    if (!nv) {
        if (lane == 0) { pf.counts[lp] = 0; pf.tau[lp] = INFINITY; }
        return;
    }
    const float nq = pf.qnorm[b];
    uint32_t fill = 0, tau = 0xFFFFFFFFu;  // wave-uniform: buffered survivors, the k-th smallest upper bound so far (sortable)
    bool over = false;
    auto compact = [&]() {  // sort by upper bound, take the k-th as the new threshold, drop what it excludes
        const uint32_t np2 = next_pow2(fill);
        for (uint32_t i = fill + lane; i < np2; i += 64) { bk[i] = ~0ull; br[i] = ~0u; }
        block_bitonic_sort<uint32_t>(bk, br, np2);
        if (fill >= k) {
            tau = (uint32_t)(bk[k - 1] >> 32);
            uint32_t nf = 0;
            for (uint32_t base = 0; base < fill; base += 64) {
                const uint32_t i = base + lane;
                const bool in = i < fill;
                const uint64_t key = in ? bk[i] : 0ull;
                const uint32_t row = in ? br[i] : 0u;
                const bool keep = in && (uint32_t)key <= tau;
                const uint64_t m = __ballot(keep);
                __syncthreads();  // every lane holds its entry before any slot of this chunk is overwritten
                if (keep) {
                    const uint32_t slot = nf + (uint32_t)__popcll(m & ((1ull << lane) - 1));
                    bk[slot] = key; br[slot] = row;
                }
                nf += (uint32_t)__popcll(m);
            }
            fill = nf;
        }
        __syncthreads();
    };
    auto process = [&](bool on, uint32_t leaf_off, uint32_t len, uint32_t take) {
        float v[PF_MAXLEN], e[PF_MAXLEN];
        uint32_t id[PF_MAXLEN];
        bool amb = on && len > PF_MAXLEN;
        float4 m[PF_MAXLEN];  // {row id (bits), |r|^2/2, |r|, -}: one 16-byte record per leaf slot, a leaf's records side by side
#pragma unroll
        for (int j = 0; j < PF_MAXLEN; j++) {
            id[j] = 0;
            m[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on && (uint32_t)j < len) { m[j] = pf.leaf_meta[(size_t)leaf_off + j]; id[j] = __float_as_uint(m[j].x); }
        }
#pragma unroll
        for (int j = 0; j < PF_MAXLEN; j++) {
            v[j] = INFINITY; e[j] = 0.f;
            if (on && (uint32_t)j < len) {
                const float s = pf.S[(size_t)id[j] * pf.Bp + b];
                if (!pf_value<KINDA>(m[j].y, m[j].z, s, nq, Kb, v[j], e[j])) { amb = true; v[j] = INFINITY; e[j] = 0.f; }
            }
        }
        uint32_t rank[PF_MAXLEN];
#pragma unroll
        for (int j = 0; j < PF_MAXLEN; j++) {
            rank[j] = 0;
#pragma unroll
            for (int i = 0; i < PF_MAXLEN; i++)
                if (i != j) rank[j] += (v[i] < v[j] || (v[i] == v[j] && i < j)) ? 1u : 0u;
        }
        if (on && !amb && take < len) {  // the take nearest rows are certain when the intervals on both sides of the cut are disjoint
            float maxin = -INFINITY, minout = INFINITY;
#pragma unroll
            for (int j = 0; j < PF_MAXLEN; j++)
                if ((uint32_t)j < len) {
                    if (rank[j] < take) maxin = fmaxf(maxin, v[j] + e[j]);
                    else minout = fminf(minout, v[j] - e[j]);
                }
            if (!(minout > maxin)) amb = true;
        }
        if (on && amb) {
            const uint32_t slot = atomicAdd(&pf.ctl[0], 1u);
            if (slot < pf.amb_cap) pf.amb[slot] = make_uint4((uint32_t)pair, leaf_off, len, take);
            else atomicOr(&pf.ctl[1], 4u);
        }
        const bool live = on && !amb;
#pragma unroll
        for (int j = 0; j < PF_MAXLEN; j++) {
            const uint32_t lo = f32_sortable(v[j] - e[j]), hi = f32_sortable(v[j] + e[j]);
            const bool pass = live && !over && (uint32_t)j < len && rank[j] < take && lo <= tau;
            const uint64_t m = __ballot(pass);
            if (m) {  // wave-uniform
                if (pass) {
                    const uint32_t slot = fill + (uint32_t)__popcll(m & ((1ull << lane) - 1));
                    bk[slot] = ((uint64_t)hi << 32) | lo; br[slot] = id[j];
                }
                fill += (uint32_t)__popcll(m);
                if (fill > PF_BUF - 64) {
                    __syncthreads();
                    compact();
                    if (fill > PF_BUF - 64) over = true;  // more rows within the bound of the k-th best than the buffer holds
                }
            }
        }
    };
    const uint32_t n_inl = nv < ZH_INLINE_VISITS ? nv : ZH_INLINE_VISITS;
    {
        const bool on = lane < n_inl;
        uint32_t lo = 0, ln = 0, tk = 0;
        if (on) { const ZhVisit vv = inl[pair * ZH_INLINE_VISITS + lane]; lo = vv.leaf_off; ln = vv.len; tk = vv.take; }
        process(on, lo, ln, tk);
    }
    if (nv > ZH_INLINE_VISITS) {
        uint32_t chunk = wlog.head[pair], remaining = nv - ZH_INLINE_VISITS;
        while (remaining) {  // wave-uniform
            const uint32_t cnt = remaining < ZH_LOG_CHUNK - 1 ? remaining : ZH_LOG_CHUNK - 1;
            const uint2 *C = wlog.pool + (size_t)chunk * ZH_LOG_CHUNK;
            const uint32_t next = C[0].x;
            const bool on = lane >= 1 && lane <= cnt;
            uint2 en = make_uint2(0, 0);
            uint32_t lo = 0, ln = 0, tk = 0;
            if (on) {
                en = C[lane];
                if (wlog.leaf_entries) { lo = en.x; ln = en.y >> 16; tk = en.y & 0xFFFFu; }  // (wave-uniform choice)
                else { const int4 r = f.node_pack[en.x]; lo = (uint32_t)r.y; ln = (uint32_t)r.z; tk = en.y; }
            }
            process(on, lo, ln, tk);
            remaining -= cnt; chunk = next;
        }
    }
    __syncthreads();
    if (fill) compact();
    uint32_t n = fill;
    if (n > pf.cap) { over = true; n = pf.cap; }
    for (uint32_t i = lane; i < n; i += 64) pf.rows[lp * pf.cap + i] = br[i];
    if (lane == 0) {
        pf.counts[lp] = n;
        // the pair's threshold for the exact pass: k of its rows have keys at or below this value (in the scale of v)
        float tf = INFINITY;
        if (tau != 0xFFFFFFFFu) tf = __uint_as_float((tau & 0x80000000u) ? tau ^ 0x80000000u : ~tau);
        pf.tau[lp] = tf;
        if (over) atomicOr(&pf.ctl[1], 1u);
    }
}

// an ambiguous visit, the reference's way: every row of the leaf scored with the canonical sums, the `take` smallest
// (key, id) are what the visit hands over -- what sweep + select do for every visit of a batch that is not prefiltered -- and
// those of them that can still be among the pair's k best join its list: k rows of the pair have keys at or below pf.tau (in
// the scale of the prefilter's v), so a row whose exact key lies above it is out.  The exact key in that scale: cosine: the
// key itself (parity: in the order of its bits, as pf_value); L2: (d* - |q|^2) / 2 with the computed |q|^2 and its rounding.
template <int KIND>
__global__ __launch_bounds__(256) void prefilter_amb_kernel(const float *__restrict__ X, uint32_t d, const float *__restrict__ Q,
                                                             const float *__restrict__ QQ, uint32_t T, uint32_t B,
                                                             const uint32_t *__restrict__ leaf_ids, int metric, int param,
                                                             ZhPrefilter pf) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6))), nw = gridDim.x * 4;
    const uint32_t total = pf.ctl[0] < pf.amb_cap ? pf.ctl[0] : pf.amb_cap;
    for (uint32_t i = wave; i < total; i += nw) {
        const uint4 a = pf.amb[i];
        const uint32_t b = a.x / T, t = a.x % T, len = a.z, take = a.w;
        const size_t lp = (size_t)t * B + b;
        if (len > 64) {
            if (lane == 0) atomicOr(&pf.ctl[1], 8u);
            continue;
        }
        uint64_t mykey = ~0ull;
        uint32_t myid = ~0u;
        double myv = 0.0;  // the exact key in the scale of v, less its own uncertainty (L2: the rounded |q|^2)
        const double nq = (double)pf.qnorm[b];
        for (uint32_t j = 0; j < len; j++) {
            const uint32_t id = leaf_ids[(size_t)a.y + j];
            float s0, s1;
            lane_sums_generic<KIND>(X + (size_t)id * d, Q + (size_t)b * d, d, lane, param, s0, s1);
            const uint64_t key = key_of(metric, param, s0, s1, KIND == K_COS ? QQ[b] : 0.f);
            if (lane == j) {
                mykey = key; myid = id;
                if (KIND == K_COS) {
                    const double kv = __longlong_as_double((long long)key);
                    myv = (param == ZH_COSINE_PARITY && kv < 0.0) ? 2.0 - kv : kv;
                } else
                    myv = 0.5 * (double)s0 - 0.5 * nq * nq - ((double)d + 10.0) * 5.9604644775390625e-8 * 1.01 * nq * nq;
            }
        }
        uint32_t rank = 0;
        for (uint32_t j = 0; j < len; j++) {
            const uint64_t kj = __shfl(mykey, (int)j);
            const uint32_t ij = __shfl(myid, (int)j);
            rank += (kj < mykey || (kj == mykey && ij < myid)) ? 1u : 0u;
        }
        if (lane < len && rank < take && !(myv > (double)pf.tau[lp])) {  // (NaN: kept)
            const uint32_t slot = atomicAdd(&pf.counts[lp], 1u);
            if (slot < pf.cap) pf.rows[lp * pf.cap + slot] = myid;
            else atomicOr(&pf.ctl[1], 2u);
        }
    }
}

// the lists' rows scored exactly: one wave per (query, tree) list, its rows one after the other with the next row's loads in flight
// (a wave per SLOT was 250k waves of which 4 in 5 found an empty slot: 0.75 ms for 60k rows)
template <int KIND>
__global__ __launch_bounds__(256) void prefilter_keys_kernel(const float *__restrict__ X, uint32_t d, const float *__restrict__ Q,
                                                              const float *__restrict__ QQ, uint32_t B, uint64_t lists, int metric,
                                                              int param, uint64_t id_base, ZhPrefilter pf,
                                                              uint64_t *__restrict__ keys, uint64_t *__restrict__ ids) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t lp = (uint64_t)blockIdx.x * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (lp >= lists) return;
    const uint32_t c0 = pf.counts[lp], c = c0 < pf.cap ? c0 : pf.cap;
    if (!c) return;
    const uint32_t b = (uint32_t)(lp % B);
    const uint32_t myrow = lane < c ? pf.rows[lp * pf.cap + lane] : 0u;  // cap <= 256: four passes at most
    for (uint32_t base = 0; base < c; base += 64) {
        const uint32_t rows64 = base == 0 ? myrow : (base + lane < c ? pf.rows[lp * pf.cap + base + lane] : 0u);
        const uint32_t cnt = c - base < 64u ? c - base : 64u;
        uint64_t mykey = 0;
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)rows64, (int)j);
            float s0, s1;
            lane_sums_generic<KIND>(X + (size_t)row * d, Q + (size_t)b * d, d, lane, param, s0, s1);
            const uint64_t key = key_of(metric, param, s0, s1, KIND == K_COS ? QQ[b] : 0.f);
            if (lane == j) mykey = key;
        }
        if (lane < cnt) {
            keys[lp * pf.cap + base + lane] = mykey;
            ids[lp * pf.cap + base + lane] = id_base + rows64;
        }
    }
    if (lane == 0) atomicAdd(&pf.ctl[2], c);
}

__global__ __launch_bounds__(256) void leaf_meta_kernel(const uint32_t *__restrict__ leaf_ids, uint64_t n, const float *__restrict__ hn2,
                                                         const float *__restrict__ norm, float4 *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = leaf_ids[i];
    out[i] = make_float4(__uint_as_float(r), hn2[r], norm[r], 0.f);
}
hipError_t zh_launch_leaf_meta(const uint32_t *dLeafIds, uint64_t n, const float *dHalfN2, const float *dNorm, float4 *dOut, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(leaf_meta_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, dLeafIds, n, dHalfN2, dNorm, dOut);
    return hipGetLastError();
}

// Half-width of the interval the reference's key lies in around the value computed from a row score (u = 2^-24, c0 = ceil(d/256)
// + 8 = longest chain of the canonical sums).
// L2 family, in the scale of v = |r|^2/2 - r.q:  the canonical d* = sum (x_i - q_i)^2 is within (c0 + 3) u D of the real D =
// |x|^2 + |q|^2 - 2 x.q; the score is within (d + 8) u |x||q| of x.q (the bound the row-score hash rests on), the stored |r|^2/2
// within (d + 8) u |x|^2/2, their difference rounds once more: |d*/2 - (v + |q|^2/2)| <= (d + c0 + 13) u / 2 * (|x| + |q|)^2;
// + 2 u (|x| + |q|)^2 for v +- e themselves, * 1.002 for the rounded norms the kernel squares.
// Cosine: ab, a2, b2 within c0 u (relative to |x||q|, |x|^2, |q|^2) give ab / sqrt(a2 b2) within 2 c0 u of the real cosine; the
// score and the two rounded norms give s / (|x||q|) within (2 (d + 8) + 4) u; + 4 u for the kernel's own roundings.
float zh_prefilter_bound(int metric, uint32_t d) {
    const double u = 5.9604644775390625e-8, c0 = (d + 255) / 256 + 8.0;
    if (metric == ZH_COSINE) return (float)((2.0 * (c0 + 4.0) + 2.0 * (d + 9.0) + 8.0) * u * 1.002);
    return (float)((0.5 * (d + c0 + 13.0) + 2.0) * u * 1.002);
}

hipError_t zh_launch_prefilter(ZhForestDev f, uint32_t d, uint32_t B, uint32_t k, int metric, int mode, const ZhPairCounts *dCounts,
                               const ZhVisit *dInline, ZhWalkLog log, ZhPrefilter pf, hipStream_t s) {
    const uint64_t pairs = (uint64_t)B * f.n_trees;
    if (!pairs) return hipSuccess;
    const float Kb = zh_prefilter_bound(metric, d);
    const dim3 grid((uint32_t)pairs), blk(64);
    if (metric == ZH_COSINE && mode == ZH_COSINE_PARITY)
        hipLaunchKernelGGL(prefilter_kernel<2>, grid, blk, 0, s, f, f.n_trees, B, k, dCounts, dInline, log, pf, Kb);
    else if (metric == ZH_COSINE)
        hipLaunchKernelGGL(prefilter_kernel<1>, grid, blk, 0, s, f, f.n_trees, B, k, dCounts, dInline, log, pf, Kb);
    else
        hipLaunchKernelGGL(prefilter_kernel<0>, grid, blk, 0, s, f, f.n_trees, B, k, dCounts, dInline, log, pf, Kb);
    return hipGetLastError();
}

hipError_t zh_launch_prefilter_exact(ZhForestDev f, uint32_t d, const float *dX, const float *dQ, const float *dQQ, uint32_t B, int metric, int mode,
                                     uint64_t id_base, ZhPrefilter pf, uint64_t *dKeys, uint64_t *dIds, hipStream_t s) {
    const uint64_t lists = (uint64_t)B * f.n_trees;
    if (!lists) return hipSuccess;
    const dim3 gk((uint32_t)((lists + 3) / 4)), blk(256);
    if (metric == ZH_COSINE) {
        hipLaunchKernelGGL(prefilter_amb_kernel<K_COS>, dim3(2048), blk, 0, s, dX, d, dQ, dQQ, f.n_trees, B, f.leaf_ids, metric, mode, pf);
        hipLaunchKernelGGL(prefilter_keys_kernel<K_COS>, gk, blk, 0, s, dX, d, dQ, dQQ, B, lists, metric, mode, id_base, pf, dKeys, dIds);
    } else {
        hipLaunchKernelGGL(prefilter_amb_kernel<K_L2>, dim3(2048), blk, 0, s, dX, d, dQ, dQQ, f.n_trees, B, f.leaf_ids, metric, mode, pf);
        hipLaunchKernelGGL(prefilter_keys_kernel<K_L2>, gk, blk, 0, s, dX, d, dQ, dQQ, B, lists, metric, mode, id_base, pf, dKeys, dIds);
    }
    return hipGetLastError();
}

hipError_t zh_launch_final_lists(uint32_t T, uint32_t B, uint32_t k, uint32_t cap, const uint64_t *dKeys, const uint64_t *dIds,
                                 const uint32_t *dCounts, uint64_t *dOutIds, uint64_t *dOutKeys, uint32_t *dOutCounts, uint32_t *dOver,
                                 hipStream_t s) {
    if (!B) return hipSuccess;
    // one wave per query over the PACKED lists, 16 KB of LDS (the 256-thread kernel's 50 KB wait for room beside the walks).  The
    // lists are mostly empty (~15-50 of 64-128 slots): up to 4x as many slots as the sort holds are tried this way; a query whose
    // lists do hold more than the sort reports it and the batch is redone with the sweep
    // ... but not where the lists CANNOT fit it: every list of a pair that walked >= k rows keeps at least k survivors, so T * k
    // entries are certain (15 trees, k = 64: 960 + extras > 1024 raised the overflow flag on every batch): those shapes go
    // to the streaming block kernel, which takes any length
    if ((uint64_t)T * cap <= 4 * MERGE_WAVE_N && (uint64_t)T * k <= MERGE_WAVE_N * 3 / 4) {
        hipLaunchKernelGGL(merge_wave_kernel, dim3(B), dim3(64), 0, s, B, T, k, dKeys, dIds, dCounts, dOutIds, dOutKeys, dOutCounts,
                           (uint64_t)B * cap, (uint64_t)B, cap, dOver);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(final_kernel<true>, dim3(B), dim3(256), 0, s, (const uint64_t *)nullptr, B, T, k, dKeys,
                       (const uint32_t *)nullptr, dIds, dCounts, (uint64_t)0, dOutIds, dOutKeys, dOutCounts, (uint64_t)B * cap, (uint64_t)B, cap,
                       (const uint32_t *)nullptr);
    return hipGetLastError();
}

