// zh_score.hip -- the hash of a whole small-leaf forest from ROW SCORES instead of one dot product per plane.
//
// With the reference's default options (max_node_size 5, lsh.rs:131-138) the walk wanders over thousands of leaves per
// (query, tree) pair and every sign of the forest is precomputed per batch (zh_api.hip, choose_dense_planes): 2 B P d flop for
// P ~ 0.44 N T planes (6.5M planes for 1M rows and 15 trees).  But a plane is built from two STORED rows a, b
// (build_hyperplane, lsh.rs:192-225: w = b - a, p = (a + b) / 2, c = -w.p), so in exact arithmetic
//     w.x + c = (b.x - |b|^2 / 2) - (a.x - |a|^2 / 2) = s_b(x) - s_a(x),     s_r(x) = r.x - |r|^2 / 2,
// and the signs of ALL planes follow from the N row scores of a query: 2 B N d flop, P / N ~ 6.5x fewer.
//
// Point_is_above (lsh.rs:39-43) is a rounded f32 computation, and results must be bit-identical to it.  So the score
// difference only DECIDES a sign when it is further from zero than a rigorous bound on everything that separates it from the
// reference's value (derivation below); the few signs inside the bound are recomputed exactly with the reference's own
// arithmetic -- the k-ascending fma chain of zh_plane_above -- by the fix-up kernel.
//
// Bound.  u = 2^-24, gamma = d u.  The reference computes w_k = fl(b_k - a_k), p_k = fl(fl(a_k + b_k) / 2),
// c = -chain(w_k p_k), dot = chain(w_k x_k), and tests f64(dot) + f64(c) >= 0.  With T = sum w_k x_k - sum w_k p_k in exact
// arithmetic on those f32 values: |(dot + c) - T| <= gamma (sum|w_k x_k| + sum|w_k p_k|) <= gamma (|w||x| + |w||p|).
// w_k = (b_k - a_k)(1 + d1), p_k = (a_k + b_k)/2 (1 + d2), |d1|, |d2| <= u, so
//     T = s_b - s_a + e,  |e| <= u |b - a||x| + (2u + u^2)(|a|^2 + |b|^2)/2 .
// The scores themselves are computed in f32: |fl(r.x) - r.x| <= gamma |r||x|, the stored |r|^2/2 within gamma |r|^2/2, and the
// three subtractions add u times their magnitudes.  With A = |a| + |b| (so |w| <= A(1+u), |p| <= A/2 (1+u)) every term is
// covered: gamma (2 A|x| + A^2) from the four chains (reference dot and constant, the two score dots, the two stored |r|^2/2)
// plus u (4 A|x| + 2.5 A^2) from the roundings of w, p and the three subtractions, i.e. (d + 2.5)(1 + 1e-4) u (2 A|x| + A^2).
// The kernel tests against (d + 8) u (2 A|x| + A^2) * 1.001 (the norms are themselves rounded, relative error ~ d u / 2) -- at
// d = 384 on ~N(0,1) rows about 0.2 % of the signs fall inside and take the exact path.
#include "zh_internal.h"

// |r|^2 / 2 and |r| of n rows (one wave per row; any summation order: the bound above covers it)
__global__ __launch_bounds__(256) void row_norms_kernel(const float *__restrict__ X, uint64_t n, uint32_t d,
                                                         float *__restrict__ half_n2, float *__restrict__ norm) {
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (r >= n) return;
    const float *row = X + r * d;
    float s = 0.f;
    for (uint32_t k = lane; k < d; k += 64) s = __builtin_fmaf(row[k], row[k], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        if (half_n2) half_n2[r] = 0.5f * s;
        norm[r] = sqrtf(s);
    }
}
hipError_t zh_launch_row_norms(const float *dX, uint64_t n, uint32_t d, float *dHalfN2, float *dNorm, hipStream_t s) {
    if (!n) return hipSuccess;
    for (uint64_t r0 = 0; r0 < n; r0 += (1ull << 24)) {  // block-indexed launches stay far below 2^32 threads
        const uint64_t nr = n - r0 < (1ull << 24) ? n - r0 : (1ull << 24);
        hipLaunchKernelGGL(row_norms_kernel, dim3((uint32_t)((nr + 3) / 4)), dim3(256), 0, s, dX + r0 * d, nr, d,
                           dHalfN2 ? dHalfN2 + r0 : nullptr, dNorm + r0);
    }
    return hipGetLastError();
}

// S[row][0..3] = row . q_j for a batch of at most four queries: a wave per row, the queries in registers -- the table of a small batch
// is a stream over the stored rows (HBM-bound), not a GEMM with 124 idle plane columns.  Any summation order: the bound covers it.
__global__ __launch_bounds__(256) void row_scores4_kernel(const float *__restrict__ X, uint64_t n, uint32_t d, const float *__restrict__ Q,
                                                           float *__restrict__ S) {
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (r >= n) return;
    const float *row = X + r * d;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (uint32_t k = lane; k < d; k += 64) {
        const float x = row[k];
        s0 = __builtin_fmaf(x, Q[k], s0);
        s1 = __builtin_fmaf(x, Q[d + k], s1);
        s2 = __builtin_fmaf(x, Q[2 * (size_t)d + k], s2);
        s3 = __builtin_fmaf(x, Q[3 * (size_t)d + k], s3);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); s3 += __shfl_xor(s3, o);
    }
    if (lane == 0) *reinterpret_cast<float4 *>(S + r * 4) = make_float4(s0, s1, s2, s3);
}
hipError_t zh_launch_row_scores4(const float *dX, uint64_t n, uint32_t d, const float *dQ4, float *dS, hipStream_t s) {
    if (!n) return hipSuccess;
    for (uint64_t r0 = 0; r0 < n; r0 += (1ull << 24)) {
        const uint64_t nr = n - r0 < (1ull << 24) ? n - r0 : (1ull << 24);
        hipLaunchKernelGGL(row_scores4_kernel, dim3((uint32_t)((nr + 3) / 4)), dim3(256), 0, s, dX + r0 * d, nr, d, dQ4, dS + r0 * 4);
    }
    return hipGetLastError();
}

// signs of 32 consecutive planes (one output word) for every query, from the score table S[row][B]: a wave per word, lane l
// takes queries 4l .. 4l+3 of each 256-query chunk (one float4 of both sample rows' score rows per plane).  Signs inside the
// bound are appended to the fix-up list, one atomic per wave, word and chunk.
// A block produces EIGHT consecutive words (each of its four waves two): the sign matrix is [query][word], so one word of 256
// queries is 256 four-byte stores a row pitch (hundreds of KB) apart -- every one a read-modify-write of a 32-byte sector
// (rocprofv3, round 2: 16.7 GB read for 13.3 GB of score rows gathered, 1.2 GB written for 0.2 GB of signs).  The block's
// 256 x 8 words meet in LDS and thread t stores the eight words of query t as one aligned 32-byte sector (wpq is a multiple
// of eight words: zh_score_words_per_query).
__global__ __launch_bounds__(256) void score_signs_kernel(const float *__restrict__ S, uint32_t B, const uint2 *__restrict__ samples,
                                                           uint32_t P, const float4 *__restrict__ plane_hab,
                                                           const float *__restrict__ qnorm,
                                                           float K, uint32_t *__restrict__ bits, uint32_t wpq,
                                                           uint2 *__restrict__ fix_list, uint32_t fix_cap,
                                                           unsigned long long *__restrict__ fix_count, uint32_t *__restrict__ unc) {
    // unc != null ("lazy" fix-ups): the uncertain signs are not listed for the fix-up kernel but flagged in a second bit matrix of
    // the same shape; the blocked walk recomputes the ones it actually steps on (zh_search.hip) -- a wandering walk meets ~5 % of
    // the forest's planes per query, so ~95 % of the listed fix-ups were signs nobody read
    __shared__ uint32_t sbits[256][9];  // [query of the chunk][word of the block] (+1: the eight words of a query start on different banks)
    __shared__ uint32_t subits[256][9];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t Wb = blockIdx.x * 8;
    for (uint32_t q0 = 0; q0 < B; q0 += 256) {
        const uint32_t q = q0 + 4 * lane;
        const bool in = q < B;  // B % 4 == 0: a lane's four queries are in or out together
        float4 xn = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) xn = *reinterpret_cast<const float4 *>(qnorm + q);
        for (uint32_t i = 0; i < 2; i++) {
            const uint32_t W = Wb + wv * 2 + i, p0 = W * 32;
            const uint32_t np = p0 >= P ? 0u : (P - p0 < 32 ? P - p0 : 32);  // (words past the last plane: zeros)
            uint32_t sw[4] = {0, 0, 0, 0}, uw[4] = {0, 0, 0, 0};
            for (uint32_t j = 0; j < np; j++) {
                const uint2 ab = samples[p0 + j];  // wave-uniform
                if (ab.x == 0xFFFFFFFFu || ab.y == 0xFFFFFFFFu) {  // a default (zero) sample vector, lsh.rs:203-220: exact path
#pragma unroll
                    for (int c = 0; c < 4; c++) uw[c] |= 1u << j;
                    continue;
                }
                // {|a|^2/2, |b|^2/2, |a| + |b|} of the plane's two sample rows, in PLANE order (plane_hab_kernel): read in sequence
                // beside the samples -- the per-row norm arrays cost four random sector fetches per plane (3.3 GB per batch)
                const float4 hab = plane_hab[p0 + j];
                const float A = hab.z, ha = hab.x, hb = hab.y;
                float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
                if (in) {
                    sa = *reinterpret_cast<const float4 *>(S + (size_t)ab.x * B + q);
                    sb = *reinterpret_cast<const float4 *>(S + (size_t)ab.y * B + q);
                }
                const float av[4] = {sa.x, sa.y, sa.z, sa.w}, bv[4] = {sb.x, sb.y, sb.z, sb.w}, xv[4] = {xn.x, xn.y, xn.z, xn.w};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float diff = (bv[c] - hb) - (av[c] - ha);
                    const float E = K * (2.0f * A * xv[c] + A * A);
                    sw[c] |= (diff >= 0.0f ? 1u : 0u) << j;
                    uw[c] |= (fabsf(diff) > E ? 0u : 1u) << j;  // NaN / inf anywhere -> not certain
                }
            }
            // the fix-up list: one atomic per wave, word and chunk
            uint32_t mine = in ? __popc(uw[0]) + __popc(uw[1]) + __popc(uw[2]) + __popc(uw[3]) : 0;
            uint32_t incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(incl, o);
                if (lane >= (uint32_t)o) incl += t;
            }
            const uint32_t total = __shfl(incl, 63);
            unsigned long long base = 0;
            if (total) {
                if (lane == 63) base = atomicAdd(fix_count, (unsigned long long)total);
                base = __shfl(base, 63);
            }
            if (in) {
                unsigned long long pos = base + incl - mine;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    sbits[4 * lane + c][wv * 2 + i] = sw[c];
                    if (unc) subits[4 * lane + c][wv * 2 + i] = uw[c];
                    uint32_t u = unc ? 0u : uw[c];
                    while (u) {
                        const uint32_t j = (uint32_t)__builtin_ctz(u);
                        u &= u - 1;
                        if (pos < fix_cap) fix_list[pos] = make_uint2(q + c, p0 + j);
                        pos++;
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t qt = q0 + threadIdx.x;
        if (qt < B) {  // the eight words of one query: one aligned 32-byte sector
            uint4 *dst = reinterpret_cast<uint4 *>(bits + (size_t)qt * wpq + Wb);
            dst[0] = make_uint4(sbits[threadIdx.x][0], sbits[threadIdx.x][1], sbits[threadIdx.x][2], sbits[threadIdx.x][3]);
            dst[1] = make_uint4(sbits[threadIdx.x][4], sbits[threadIdx.x][5], sbits[threadIdx.x][6], sbits[threadIdx.x][7]);
            if (unc) {
                uint4 *du = reinterpret_cast<uint4 *>(unc + (size_t)qt * wpq + Wb);
                du[0] = make_uint4(subits[threadIdx.x][0], subits[threadIdx.x][1], subits[threadIdx.x][2], subits[threadIdx.x][3]);
                du[1] = make_uint4(subits[threadIdx.x][4], subits[threadIdx.x][5], subits[threadIdx.x][6], subits[threadIdx.x][7]);
            }
        }
        __syncthreads();
    }
}

// the same for a batch of exactly four queries (a padded single query, a handful of queries): a LANE per plane -- with a lane per
// query group 63 of 64 lanes would idle and every plane would be a dependent-load chain of its own
__global__ __launch_bounds__(256) void score_signs4_kernel(const float *__restrict__ S, const uint2 *__restrict__ samples, uint32_t P,
                                                            const float *__restrict__ hn2, const float *__restrict__ rnorm,
                                                            const float *__restrict__ qnorm, float K, uint32_t *__restrict__ bits,
                                                            uint32_t wpq, uint2 *__restrict__ fix_list, uint32_t fix_cap,
                                                            unsigned long long *__restrict__ fix_count) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t p0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;  // this wave's 64 planes = two output words per query
    if (p0 >= P) return;
    const uint32_t p = p0 + lane;
    bool sg[4] = {false, false, false, false}, un[4] = {false, false, false, false};
    if (p < P) {
        const uint2 ab = samples[p];
        if (ab.x == 0xFFFFFFFFu || ab.y == 0xFFFFFFFFu) {
            un[0] = un[1] = un[2] = un[3] = true;
        } else {
            const float A = rnorm[ab.x] + rnorm[ab.y], ha = hn2[ab.x], hb = hn2[ab.y];
            const float4 sa = *reinterpret_cast<const float4 *>(S + (size_t)ab.x * 4), sb = *reinterpret_cast<const float4 *>(S + (size_t)ab.y * 4);
            const float4 xn = *reinterpret_cast<const float4 *>(qnorm);
            const float av[4] = {sa.x, sa.y, sa.z, sa.w}, bv[4] = {sb.x, sb.y, sb.z, sb.w}, xv[4] = {xn.x, xn.y, xn.z, xn.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float diff = (bv[c] - hb) - (av[c] - ha);
                sg[c] = diff >= 0.0f;
                un[c] = !(fabsf(diff) > K * (2.0f * A * xv[c] + A * A));
            }
        }
    }
    const uint32_t mine = (un[0] ? 1u : 0u) + (un[1] ? 1u : 0u) + (un[2] ? 1u : 0u) + (un[3] ? 1u : 0u);
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= (uint32_t)o) incl += t;
    }
    const uint32_t total = __shfl(incl, 63);
    unsigned long long base = 0;
    if (total) {
        if (lane == 63) base = atomicAdd(fix_count, (unsigned long long)total);
        base = __shfl(base, 63);
    }
    unsigned long long pos = base + incl - mine;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const unsigned long long m = __ballot(sg[c]);
        if (lane == 0) bits[(size_t)c * wpq + (p0 >> 5)] = (uint32_t)m;
        if (lane == 32 && p0 + 32 < P) bits[(size_t)c * wpq + (p0 >> 5) + 1] = (uint32_t)(m >> 32);
        if (un[c]) {
            if (pos < fix_cap) fix_list[pos] = make_uint2((uint32_t)c, p);
            pos++;
        }
    }
}

// the exact value of every listed sign: the reference's own arithmetic (zh_plane_above: k-ascending fma chain, f64 test)
__global__ __launch_bounds__(256) void score_fixup_kernel(const float *__restrict__ Q, uint32_t d, const float *__restrict__ planes,
                                                           const float *__restrict__ consts, uint32_t *__restrict__ bits,
                                                           uint32_t wpq, const uint2 *__restrict__ fix_list, uint32_t fix_cap,
                                                           const unsigned long long *__restrict__ fix_count) {
    const unsigned long long n = *fix_count < fix_cap ? *fix_count : fix_cap;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint2 e = fix_list[i];
        const bool above = zh_plane_above(planes + (size_t)e.y * d, consts[e.y], Q + (size_t)e.x * d, d);
        uint32_t *w = bits + (size_t)e.x * wpq + (e.y >> 5);
        if (above) atomicOr(w, 1u << (e.y & 31)); else atomicAnd(w, ~(1u << (e.y & 31)));
    }
}
// more uncertain signs than the list holds (never seen; the list is sized for 4x the expected share): the same decision is
// re-derived for every sign and the uncertain ones are recomputed in place, a thread per output word
__global__ __launch_bounds__(256) void score_overflow_kernel(const float *__restrict__ S, uint32_t B, const uint2 *__restrict__ samples,
                                                              uint32_t P, const float *__restrict__ hn2, const float *__restrict__ rnorm,
                                                              const float *__restrict__ qnorm, float K, const float *__restrict__ Q,
                                                              uint32_t d, const float *__restrict__ planes, const float *__restrict__ consts,
                                                              uint32_t *__restrict__ bits, uint32_t wpq, uint32_t fix_cap,
                                                              const unsigned long long *__restrict__ fix_count) {
    if (*fix_count <= fix_cap) return;
    const uint32_t words = (P + 31) / 32;
    const unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (unsigned long long)B * words) return;
    const uint32_t q = (uint32_t)(idx / words), W = (uint32_t)(idx % words), p0 = W * 32;
    uint32_t word = bits[(size_t)q * wpq + W];
    for (uint32_t j = 0; j < 32 && p0 + j < P; j++) {
        const uint2 ab = samples[p0 + j];
        bool certain = false;
        if (ab.x != 0xFFFFFFFFu && ab.y != 0xFFFFFFFFu) {
            const float A = rnorm[ab.x] + rnorm[ab.y];
            const float diff = (S[(size_t)ab.y * B + q] - hn2[ab.y]) - (S[(size_t)ab.x * B + q] - hn2[ab.x]);
            certain = fabsf(diff) > K * (2.0f * A * qnorm[q] + A * A);
        }
        if (!certain) {
            const bool above = zh_plane_above(planes + (size_t)(p0 + j) * d, consts[p0 + j], Q + (size_t)q * d, d);
            word = above ? word | (1u << j) : word & ~(1u << j);
        }
    }
    bits[(size_t)q * wpq + W] = word;
}

// lazy fix-ups after all: a consumer other than the blocked walk needs the batch's signs (the emit walk of a batch whose visit log
// ran out): every flagged sign recomputed in place, a thread per sign word
__global__ __launch_bounds__(256) void score_unc_fix_kernel(const float *__restrict__ Q, uint32_t B, uint32_t d, const float *__restrict__ planes,
                                                             const float *__restrict__ consts, uint32_t P, uint32_t *__restrict__ bits,
                                                             const uint32_t *__restrict__ unc, uint32_t wpq) {
    const uint32_t words = (P + 31) / 32;
    const unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (unsigned long long)B * words) return;
    const uint32_t q = (uint32_t)(idx / words), W = (uint32_t)(idx % words);
    uint32_t u = unc[(size_t)q * wpq + W];
    if (!u) return;
    uint32_t word = bits[(size_t)q * wpq + W];
    while (u) {
        const uint32_t j = (uint32_t)__builtin_ctz(u);
        u &= u - 1;
        const uint32_t p = W * 32 + j;
        if (p >= P) break;
        const bool above = zh_plane_above(planes + (size_t)p * d, consts[p], Q + (size_t)q * d, d);
        word = above ? word | (1u << j) : word & ~(1u << j);
    }
    bits[(size_t)q * wpq + W] = word;
}
hipError_t zh_launch_score_unc_fix(const float *dQ, uint32_t B, uint32_t d, const float *dPlanes, const float *dConsts, uint32_t P,
                                   uint32_t *dBits, const uint32_t *dUnc, uint32_t wpq, hipStream_t s) {
    const unsigned long long n = (unsigned long long)B * ((P + 31) / 32);
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(score_unc_fix_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, dQ, B, d, dPlanes, dConsts, P, dBits, dUnc, wpq);
    return hipGetLastError();
}

// per plane: {|a|^2 / 2, |b|^2 / 2, |a| + |b|, 0} of its two sample rows (the same float values score_signs4 / the overflow path
// read from the per-row arrays)
__global__ __launch_bounds__(256) void plane_hab_kernel(const uint2 *__restrict__ samples, uint32_t P, const float *__restrict__ hn2,
                                                         const float *__restrict__ rnorm, float4 *__restrict__ out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const uint2 ab = samples[p];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ab.x != 0xFFFFFFFFu && ab.y != 0xFFFFFFFFu) v = make_float4(hn2[ab.x], hn2[ab.y], rnorm[ab.x] + rnorm[ab.y], 0.f);
    out[p] = v;
}
hipError_t zh_launch_plane_hab(const uint2 *dSamples, uint32_t P, const float *dHalfN2, const float *dRowNorm, float4 *dOut, hipStream_t s) {
    if (!P) return hipSuccess;
    hipLaunchKernelGGL(plane_hab_kernel, dim3((P + 255) / 256), dim3(256), 0, s, dSamples, P, dHalfN2, dRowNorm, dOut);
    return hipGetLastError();
}

float zh_score_bound_factor(uint32_t d) { return (float)(d + 8) * 5.9604645e-8f * 1.001f; }

hipError_t zh_launch_score_signs(const float *dS, uint32_t B, const uint2 *dSamples, uint32_t P, const float *dHalfN2, const float *dRowNorm,
                                 const float4 *dPlaneHab, const float *dQNorm, const float *dQ, uint32_t d, const float *dPlanes, const float *dConsts,
                                 uint32_t *dBits, uint32_t wpq, uint2 *dFixList, uint32_t fix_cap, unsigned long long *dFixCount,
                                 uint32_t *dUnc, hipStream_t s) {
    if (!B || !P) return hipSuccess;
    const uint32_t words = (P + 31) / 32;
    const float K = zh_score_bound_factor(d);
    if (B == 4)
        hipLaunchKernelGGL(score_signs4_kernel, dim3((P + 255) / 256), dim3(256), 0, s, dS, dSamples, P, dHalfN2, dRowNorm, dQNorm, K, dBits, wpq,
                           dFixList, fix_cap, dFixCount);
    else
        hipLaunchKernelGGL(score_signs_kernel, dim3((words + 7) / 8), dim3(256), 0, s, dS, B, dSamples, P, dPlaneHab, dQNorm, K,
                           dBits, wpq, dFixList, fix_cap, dFixCount, B == 4 ? nullptr : dUnc);
    if (dUnc && B != 4) return hipGetLastError();  // lazy: the walk recomputes the flagged signs it meets
    hipLaunchKernelGGL(score_fixup_kernel, dim3(4096), dim3(256), 0, s, dQ, d, dPlanes, dConsts, dBits, wpq, dFixList, fix_cap, dFixCount);
    const unsigned long long n = (unsigned long long)B * words;
    hipLaunchKernelGGL(score_overflow_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, dS, B, dSamples, P, dHalfN2, dRowNorm,
                       dQNorm, K, dQ, d, dPlanes, dConsts, dBits, wpq, fix_cap, dFixCount);
    return hipGetLastError();
}
