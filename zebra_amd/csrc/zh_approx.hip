// zh_approx.hip -- the table scan with HALF-WIDTH queries (round 4); shares zh_device.h's helpers (butterflies, select_fast /
// select_slow, the canonical sums) with zh_search.hip.
//
// What bounds scan_sweep_kernel is the rate at which (row, query) pairs pull their query through TA / vector L1 / L2: 4*d bytes
// per pair (profiles/r03_pmc_scan_mix.txt).  Here the queries of a batch are copied ONCE to fp16 (power-of-two scaled per query,
// round to nearest even), laid out so that a lane's eight halves are one 16-byte load: 2*d bytes per pair, and every load a full
// dwordx4.  A wave works as G groups of 64 / G lanes; all groups hold the SAME stored row (f32, from HBM, as before) and each
// scores it against a different query of the row's pair list: x . h by v_fma_mix_f32 (f32 += f32 * f16, one rounding), a
// 16- or 32-lane butterfly, one result per group and step.
//
// The f32 sum is NOT the reference's key.  It determines the key up to a rigorous interval [lo, hi] (derivation at
// zh_approx_bound): the fp16 rounding of the query is MEASURED per query (|q - h / sigma|, not assumed 2^-11), the f32 rounding
// of every sum involved is bounded as for the prefilter.  The intervals decide what they can:
//   select_interval  a visit hands over its `take` nearest rows (lsh.rs:317-323).  take == top_k: ANY superset inside the leaf
//                    gives the same final answer (a row outside the visit's top_k is beaten by top_k rows of that visit, all of
//                    them candidates), so every row whose lo does not exceed the take-th smallest hi goes on.  take < top_k (a
//                    backup visit after a short leaf, lsh.rs:340-345): membership matters -- exact_visit_kernel scores the
//                    leaf with the reference's arithmetic and ranks by (key, id), as sweep + select would.
//   final_interval   per query: duplicates out (a row reached through several trees), tau = the top_k-th smallest hi, the rows
//                    with lo <= tau (a few more than top_k) get the reference's key (canonical sums), sort by (key, id), top_k
//                    (lsh.rs:557-564).
// Every key returned is the canonical one and no leaf member is left unranked: results are bit-identical to the exact scan by
// construction.  A list or table that runs over raises a flag ON THE DEVICE, and the exact scan + select + final -- enqueued
// behind with that flag as their predicate -- redo the batch in stream order; no host round trip, so the sharded search (which
// consumes results in stream order) can use it too.

#include <cstdlib>

#include "zh_internal.h"
#include "zh_device.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// groups per wave by dimension (0: not supported -> exact scan): every lane loads NH = d * G / 512 dwordx4 of halves per pair
#ifndef ZH_APX_G768
#define ZH_APX_G768 2   // A/B: 4 = four 16-lane groups at d = 768 (48 row registers per lane)
#endif
#ifndef ZH_APX_G128
#define ZH_APX_G128 8   // A/B: 4 = 16-lane groups at d = 128
#endif
uint32_t zh_approx_groups(uint32_t d) {
    switch (d) {
    case 128: return ZH_APX_G128;
    case 256: case 384: return 4;
    case 768: return ZH_APX_G768;
    case 512: case 1024: return 2;
    default: return 0;
    }
}
// ... and where it pays by itself (the cost model's side): all groups of a wave work on ONE stored row, so a row wanted by fewer
// queries than the wave has groups leaves lanes idle -- d = 256 (four groups) at 3-6 pairs per row: only when asked for
// (zh_set_sweep_mode 4).  d = 128 has a kernel of its own (rows in LDS, every group its own pair), but
// its per-(row, tree) entry work -- 15 entries per row for 5.5 pairs at cfg5 -- keeps it level with the leaf-major sweep there (29.5
// against 27.7 ms per batch at window 2, 22.7 against 24.4 at window 4; profiles/r04_pmc_scan_approx128_cfg5*.txt): also only when asked for.
bool zh_approx_pays(uint32_t d) { return d >= 384; }

// Half-width of the interval around the value the approximate scan computes (u = 2^-24, c0 = ceil(d / 256) + 8 = the longest
// chain of the canonical sums; nx, nq upper estimates of |x|, |q|; dq >= |q - h / sigma|, measured by qhalf_kernel).
// L2 family, in the scale of d* = the canonical f32 sum of (x_i - q_i)^2 (L2's sqrt is monotone): the kernel forms
//   V = (a2 + b2) - 2 s / sigma,   a2, b2 = f32 sums of squares (<= 32 roundings each), s = f32 sum of x_i h_i (<= 32 roundings)
// and |d* - D| <= (c0 + 3) u D, |a2 - |x|^2| <= 32 u |x|^2, |b2 - |q|^2| <= 32 u |q|^2, |s / sigma - x.q| <= |x| dq + 33 u |x||q|,
// two roundings for V itself: |V - d*| <= (c0 + 54) u (|x| + |q|)^2 + 2 |x| dq; (c0 + 100) leaves room for V +- E and E themselves.
// Cosine, in the scale of the clipped distance 1 - cos: the canonical sums give the reference's value within 2 c0 u of the real
// one; s / (sigma nx nq) within dq / |q| + 73 u; + 4 u for the kernel's own roundings -> (2 c0 + 80) u + dq / |q|.
// mfma (scan_mfma_kernel): s is the sum of four MFMA accumulators, ops = 33 d / 128 + 2 operations of at most 2 u each on its longest chain
// instead of 33 of u: |s / sigma - x.q| gains (2 ops - 33) u |x||q| -- in V's scale 2 |x||q| <= (|x| + |q|)^2 / 2, i.e. + ops u; in the cosine's + 2 ops u.
// The rounding of the ROW (|x - xh / sigma_x| <= rho |x|) is approx_interval's `rho` term.
float zh_approx_bound(int metric, uint32_t d, int mfma) {  // 0 the VALU scans, 1 scan_mfma_kernel, 2 sweep128h_kernel (|x|^2 from 16 v_dot2 of the rounded row: + 40)
    const double u = 5.9604644775390625e-8, c0 = (d + 255) / 256 + 8.0, ops = mfma ? 33.0 * (d / 128) + 2.0 + (mfma == 2 ? 40.0 : 0.0) : 0.0;
    if (metric == ZH_COSINE) return (float)(1.01 * (2.0 * c0 + 80.0 + 2.0 * ops) * u);
    return (float)(1.01 * (c0 + 100.0 + ops) * u);
}

// No fp16 copy holds a SUBNORMAL half: whether the matrix cores (or v_dot2) take subnormal f16 inputs at face value or as zeros is then
// no concern of the bound -- the copy itself says zero, and the error that makes is part of the MEASURED rounding error (the query's delta,
// the rows' rho).
__device__ __forceinline__ _Float16 f16_no_subnormal(float v) {
    const _Float16 h = (_Float16)v;  // round to nearest even
    return fabsf((float)h) < 6.103515625e-05f ? (_Float16)0.f : h;
}

// ---- the fp16 copy of a batch's queries ----
// sigma = 2^(14 - e) with max |q_i| in [2^(e-1), 2^e): h_i = rne_f16(q_i sigma), |h_i| < 2^14.  Layout per query (2 d bytes): NH
// blocks of LG * 16 bytes; block i, lane l holds the halves of elements 4 LG (2 i) + 4 l + t (t = 0..3) and 4 LG (2 i + 1) + 4 l + t
// -- the elements of the row registers 2 i and 2 i + 1 of a lane whose group loads the stored row as 4 LG-element pieces.
// qmeta = {1 / sigma, f32 |q|^2, upper estimate of |q|, upper estimate of |q - h / sigma|}; all NaN: nothing is certain about
// this query (non-finite or out-of-range elements) -- its pairs get the interval (-inf, +inf).
template <int G>
__global__ __launch_bounds__(64) void qhalf_kernel(const float *__restrict__ Q, uint32_t B, uint32_t d, _Float16 *__restrict__ Qh,
                                                    float4 *__restrict__ qmeta) {
    constexpr uint32_t LG = 64 / G;
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    const float *q = Q + (size_t)b * d;
    float m = 0.f, s2 = 0.f;
    bool bad = false;
    for (uint32_t e = lane; e < d; e += 64) {
        const float v = q[e];
        bad |= !(v - v == 0.f);
        m = fmaxf(m, fabsf(v));
        s2 = __builtin_fmaf(v, v, s2);
    }
    m = wave_butterfly<OpMax>(m);
    s2 = wave_sum_canonical(s2);
    bad = __ballot(bad) != 0;
    float sigma = 1.f, inv = 1.f;
    if (!bad && m > 0.f) {
        int ex = 0;
        (void)frexpf(m, &ex);
        if (ex < -90 || ex > 90) bad = true;
        else { sigma = ldexpf(1.f, 14 - ex); inv = ldexpf(1.f, ex - 14); }
    }
    float d2 = 0.f;
    for (uint32_t e = lane; e < d; e += 64) {
        const float v = bad ? 0.f : q[e];
        const _Float16 h = f16_no_subnormal(v * sigma);
        const float df = v - (float)h * inv;
        d2 = __builtin_fmaf(df, df, d2);
        const uint32_t j = e / (4 * LG), rem = e % (4 * LG), l = rem / 4, t = rem % 4, i = j / 2, wh = j % 2;
        Qh[(size_t)b * d + ((size_t)(i * LG + l)) * 8 + wh * 4 + t] = h;
    }
    d2 = wave_sum_canonical(d2);
    if (lane == 0) {
        float4 o;
        if (bad || !(s2 - s2 == 0.f)) o = make_float4(NAN, NAN, NAN, NAN);
        else o = make_float4(inv, s2, sqrtf(s2) * (1.0f + 1e-5f), sqrtf(d2) * 1.001f);
        qmeta[b] = o;
    }
}

hipError_t zh_launch_qhalf(const float *dQ, uint32_t B, uint32_t d, void *dQh, float4 *dQmeta, int layout, hipStream_t s) {
    if (!B) return hipSuccess;
    // layout 0: the VALU scan's groups; 1: the MFMA scan's k-groups (four lanes wide); 2: natural order (LG = 1; sweep128h_kernel)
    const uint32_t G = layout == 2 ? 64 : (layout == 1 ? 16 : zh_approx_groups(d));
    if (G == 64) hipLaunchKernelGGL(qhalf_kernel<64>, dim3(B), dim3(64), 0, s, dQ, B, d, (_Float16 *)dQh, dQmeta);
    else if (G == 16) hipLaunchKernelGGL(qhalf_kernel<16>, dim3(B), dim3(64), 0, s, dQ, B, d, (_Float16 *)dQh, dQmeta);
    else if (G == 8) hipLaunchKernelGGL(qhalf_kernel<8>, dim3(B), dim3(64), 0, s, dQ, B, d, (_Float16 *)dQh, dQmeta);
    else if (G == 4) hipLaunchKernelGGL(qhalf_kernel<4>, dim3(B), dim3(64), 0, s, dQ, B, d, (_Float16 *)dQh, dQmeta);
    else if (G == 2) hipLaunchKernelGGL(qhalf_kernel<2>, dim3(B), dim3(64), 0, s, dQ, B, d, (_Float16 *)dQh, dQmeta);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// sum over the LG lanes of a group (LG = 8, 16, 32 or 64): the first steps of the canonical butterfly
template <int LG>
__device__ __forceinline__ float group_sum(float s) {
    s = s + dpp_mov<0xB1>(s);
    s = s + dpp_mov<0x4E>(s);
    s = s + dpp_mov<0x141>(s);
    if (LG >= 16) s = s + dpp_mov<0x140>(s);
    if (LG >= 32) s = xor16<OpAdd>(s);
    if (LG >= 64) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
        s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return s;
}

// The L2 family's interval is V -+ E: V from the pair's sums, E from the norms.  Both as functions of their own, because the fused sweep's pre-test
// (fused_pretest) evaluates V the same way and E at an upper bound of nx: every operation of l2_E is non-decreasing in nx >= 0 (all factors are
// non-negative) and so is its rounded result -- E(nx_max) >= E(nx) in f32, whatever the roundings.
__device__ __forceinline__ float l2_V(float s, float a2, const float4 qm) {
    const float sh = s * qm.x, sum = a2 + qm.y;
    return sum - 2.0f * sh;
}
__device__ __forceinline__ float l2_E(float nx, const float4 qm, float Kc, float rho, float rho_n) {
    const float nn = nx + qm.z;
    // |s / sigma - x.q| <= |x| dq + |x - x'| (|q| + dq), |x - x'| <= rho |x| (rho = 0: the scan multiplied the f32 row)
    return Kc * nn * nn + 2.02f * nx * (qm.w + rho * (qm.z + qm.w)) + 2.2f * rho_n * nx * nx;
}
// the interval of one (row, query) pair from its sums: sortable lo | sortable hi << 32; (0, all ones) = nothing certain
template <int KINDA>
__device__ __forceinline__ uint64_t approx_interval(float s, float a2, const float4 qm, float Kc, float rho, float rho_n) {
    uint32_t lo_s = 0u, hi_s = 0xFFFFFFFFu;
    const float sh = s * qm.x;
    if (KINDA == 0) {
        // rho_n != 0: a2 is the ROUNDED row's |x'|^2 (sweep128h_kernel): |x| <= |x'| (1 + 2 rho_n), ||x'|^2 - |x|^2| <= 2.2 rho_n |x'|^2
        const float nx = sqrtf(a2) * (1.0f + 1e-5f) * (1.0f + 2.0f * rho_n), nn = nx + qm.z, sum = a2 + qm.y;
        const float V = l2_V(s, a2, qm);
        const float E = l2_E(nx, qm, Kc, rho, rho_n);
        if ((V - V == 0.f) && (E - E == 0.f) && nn > 1e-12f && sum < 1e37f) { lo_s = f32_sortable(V - E); hi_s = f32_sortable(V + E); }
    } else {
        const float nx = sqrtf(a2), nq = sqrtf(qm.y);
        if (nx > 1e-12f && nq > 1e-12f && (sh - sh == 0.f) && (nx - nx == 0.f) && (nq - nq == 0.f)) {
            float r = 1.0f - sh / (nx * nq);
            r = r > 0.f ? r : 0.f;
            const float dqr = qm.w / nq;
            const float e = Kc + 1.01f * (dqr + rho * (1.0f + dqr) + rho_n);
            float v = r;
            bool ok = true;
            if (KINDA == 2) {  // ZH_COSINE_PARITY keys compare as the bits of 1 - distance: as pf_value<2>
                const float key = 1.0f - r;
                ok = fabsf(key) > e;
                v = key > 0.f ? key : 2.0f - key;
            }
            if (ok && (v - v == 0.f) && (e - e == 0.f)) { lo_s = f32_sortable(v - e); hi_s = f32_sortable(v + e); }
        }
    }
    return ((uint64_t)hi_s << 32) | lo_s;
}

// Phase 1 of a table-scan wave: its nr * T (row, tree) entries -> which leaves this batch visits, by which query, where the result
// goes, and the exclusive scan of the visit counts (= positions in the wave's pair list).  Three dependent fetches per entry -- the
// entry (streamed), a word of the visited-leaf bitmap, the leaf's visit record -- issued STAGE BY STAGE for all of a lane's
// entries (clamped / dummy addresses instead of branches), so that a wave pays three memory round trips, not three per entry:
// the scan of a 125M x 128 shard spends more line requests here than on its pairs (profiles/r04_pmc_scan_approx128_cfg5*.txt).
__device__ __forceinline__ uint32_t scan_phase1(uint32_t lane, uint32_t n_ent, const uint2 *__restrict__ ent,
                                                const uint32_t *__restrict__ visitBits, const uint4 *__restrict__ nodeVisit,
                                                uint32_t *eGb, uint32_t *eWithin, uint32_t *eC, uint32_t *off, uint32_t *eB0, uint64_t *eK0) {
    unsigned long long rlw[ZH_SCAN_NE];
#pragma unroll
    for (int j = 0; j < ZH_SCAN_NE; j++) {
        const uint32_t e = lane + 64u * j;
        rlw[j] = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(ent + (e < n_ent ? e : 0)));
    }
    uint32_t word[ZH_SCAN_NE];
    bool in[ZH_SCAN_NE];
#pragma unroll
    for (int j = 0; j < ZH_SCAN_NE; j++) {
        const uint32_t node = (uint32_t)rlw[j];
        in[j] = lane + 64u * j < n_ent && node != 0xFFFFFFFFu;
        word[j] = visitBits[in[j] ? node >> 5 : 0];
    }
    uint4 nv[ZH_SCAN_NE];
#pragma unroll
    for (int j = 0; j < ZH_SCAN_NE; j++) {
        const uint32_t node = (uint32_t)rlw[j];
        in[j] = in[j] && ((word[j] >> (node & 31)) & 1u);
        nv[j] = nodeVisit[in[j] ? node : 0];
    }
    uint32_t P = 0;
#pragma unroll
    for (int j = 0; j < ZH_SCAN_NE; j++) {
        eWithin[j] = (uint32_t)(rlw[j] >> 32);
        eC[j] = in[j] ? nv[j].x & 0x0FFFFFFFu : 0u;
        eGb[j] = nv[j].y; eB0[j] = nv[j].z;
        eK0[j] = ((uint64_t)(nv[j].x >> 28) << 32) | nv[j].w;
        uint32_t incl = eC[j];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o);
            if (lane >= (uint32_t)o) incl += t;
        }
        off[j] = P + incl - eC[j];
        P += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    return P;
}

#define ZH_APX_CAP 512   // pair records of a wave's LDS list
#ifndef ZH_APX_WAVES
#define ZH_APX_WAVES 0
#endif
#ifndef ZH_APX_REPL_ROWS
#define ZH_APX_REPL_ROWS 0   // A/B: every group loads the row for itself
#endif
template <int D, int G, int KINDA>
__global__ __launch_bounds__(256)
#if ZH_APX_WAVES
__attribute__((amdgpu_waves_per_eu(ZH_APX_WAVES, 8)))
#endif
void scan_approx_kernel(const float *__restrict__ X, const uint4 *__restrict__ Qh,
                                                           const float4 *__restrict__ qmeta, const uint2 *__restrict__ rowLeaf,
                                                           uint32_t T, uint32_t RW, const uint32_t *__restrict__ visitBits,
                                                           const uint4 *__restrict__ nodeVisit, const ZhGroup *__restrict__ groups,
                                                           uint32_t GRP, uint64_t row_begin, uint64_t row_end, float Kc,
                                                           uint64_t *__restrict__ iv) {
    constexpr int LG = 64 / G, NH = D * G / 512, NR = 2 * NH;  // lanes per group; dwordx4 of halves per pair; float4 registers per row
#ifdef ZH_APX_RB
    constexpr int RB = ZH_APX_RB;
#else
    constexpr int RB = NR <= 6 ? 4 : 2;                         // rows per HBM round trip
#endif
    static_assert(D * G % 512 == 0 && NH >= 1, "a lane's share of a query is whole 16-byte loads");
    __shared__ uint4 pair_list[4][ZH_APX_CAP];  // {row of the wave's RW, query, interval slot lo, hi}
    __shared__ uint32_t row_start[4][20];
    const uint32_t lane = threadIdx.x & 63, l = lane & (LG - 1), g = lane / LG;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wid;
    const uint64_t r0 = row_begin + wave * RW;
    if (r0 >= row_end) return;
    const uint32_t nr = (uint32_t)(row_end - r0 < RW ? row_end - r0 : RW);
    const uint32_t n_ent = nr * T;
    const uint2 *__restrict__ ent = rowLeaf + (size_t)r0 * T;
    // ---- phase 1 (as scan_sweep_kernel): the wave's nr * T (row, tree) entries -> pairs, in (row, tree, visit) order ----
    uint32_t eGb[ZH_SCAN_NE], eWithin[ZH_SCAN_NE], eC[ZH_SCAN_NE], off[ZH_SCAN_NE], eB0[ZH_SCAN_NE];
    uint64_t eK0[ZH_SCAN_NE];
    const uint32_t P = scan_phase1(lane, n_ent, ent, visitBits, nodeVisit, eGb, eWithin, eC, off, eB0, eK0);
    if (P == 0) return;
    const float4 *__restrict__ X4 = reinterpret_cast<const float4 *>(X);
    // finished sums wait in lane registers -- group g's result of step n in lane g * LG + n -- so that the interval arithmetic and
    // the stores run for 64 pairs at once
    // ... and what is stored is the RAW pair {x . h, |x|^2}: the interval arithmetic needs four constants of the pair's QUERY, a 16-byte
    // gather per pair here (one more L1-miss line per pair, 6 % of the kernel's requests) but ONE load per visit in select_tau_kernel,
    // which turns a visit's raw pairs into intervals in place before it ranks them
    float my_s = 0.f, my_a2 = 0.f;
    uint64_t my_slot = 0;
    uint32_t nst = 0;
    bool my_on = false;
    auto flush = [&]() {
        if (my_on) __builtin_nontemporal_store(((uint64_t)__float_as_uint(my_a2) << 32) | __float_as_uint(my_s), iv + my_slot);
        my_on = false;
        nst = 0;
    };
    auto load_q = [&](uint32_t b, uint4 *hq) {
#ifdef ZH_APX_EXP_Q_L1   // timing experiment (results invalid): eight queries only -> the query loads hit the vector L1
        b &= 7u;
#endif
        const uint4 *qp = Qh + (size_t)b * (D / 8) + l;
#pragma unroll
        for (int i = 0; i < NH; i++) hq[i] = qp[i * LG];
    };
    // A stored row, streamed once per window.  Every group needs the whole row as 4 LG-element pieces (register j: elements
    // 4 LG j + 4 l + t).  G = 2: the wave loads each 1-KiB piece ONCE (64 lanes x 16 bytes) and v_permlane32_swap hands both halves
    // to both groups -- the replicated form (each group loading for itself) put every row through the TA twice, 20 of the 64 GB
    // a cfg3 launch pulls through the vector L1s (profiles/r04_pmc_scan_approx.txt).  G = 4: replicated loads.
    auto load_x = [&](uint64_t row, float4 *v) {
        if constexpr (G == 2 && !ZH_APX_REPL_ROWS) {
            const float4 *r4 = X4 + (size_t)row * (D / 4) + lane;
#pragma unroll
            for (int j = 0; j < NR / 2; j++) {
                const float4 t = ld16<true>(r4 + 64 * j);
                const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(t.x), __float_as_uint(t.x), false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(t.y), __float_as_uint(t.y), false, false);
                const auto sz = __builtin_amdgcn_permlane32_swap(__float_as_uint(t.z), __float_as_uint(t.z), false, false);
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(t.w), __float_as_uint(t.w), false, false);
                v[2 * j] = make_float4(__uint_as_float(sx[0]), __uint_as_float(sy[0]), __uint_as_float(sz[0]), __uint_as_float(sw[0]));
                v[2 * j + 1] = make_float4(__uint_as_float(sx[1]), __uint_as_float(sy[1]), __uint_as_float(sz[1]), __uint_as_float(sw[1]));
            }
        } else {
#ifdef ZH_APX_EXP_ROWS_L2   // timing experiment (results invalid): every row from a 3-MB region -> L2 hits instead of HBM reads
            const float4 *r4 = X4 + (size_t)(row & 1023) * (D / 4) + l;
#else
            const float4 *r4 = X4 + (size_t)row * (D / 4) + l;
#endif
#pragma unroll
            for (int j = 0; j < NR; j++) v[j] = ld16<true>(r4 + LG * j);
        }
    };
    auto row_sumsq = [&](const float4 *v) {
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NR; j++) sq4(v[j], c);
        return group_sum<LG>((c.x + c.y) + (c.z + c.w));
    };
    auto dot_h = [&](const float4 *v, const uint4 *hq) {  // x . h over the lane's 8 NH elements, then over the group
        float ax = 0.f, ay = 0.f, az = 0.f, aw = 0.f;
#pragma unroll
        for (int i = 0; i < NH; i++) {
            f16x8 h;
            __builtin_memcpy(&h, &hq[i], 16);
            ax = __builtin_fmaf(v[2 * i].x, (float)h[0], ax);
            ay = __builtin_fmaf(v[2 * i].y, (float)h[1], ay);
            az = __builtin_fmaf(v[2 * i].z, (float)h[2], az);
            aw = __builtin_fmaf(v[2 * i].w, (float)h[3], aw);
            ax = __builtin_fmaf(v[2 * i + 1].x, (float)h[4], ax);
            ay = __builtin_fmaf(v[2 * i + 1].y, (float)h[5], ay);
            az = __builtin_fmaf(v[2 * i + 1].z, (float)h[6], az);
            aw = __builtin_fmaf(v[2 * i + 1].w, (float)h[7], aw);
        }
        return group_sum<LG>((ax + ay) + (az + aw));
    };
    auto stash = [&](float s, float a2, uint32_t, uint64_t slot, bool valid) {
        if (l == nst) { my_s = s; my_a2 = a2; my_slot = slot; my_on = valid; }
        if (++nst == (uint32_t)LG) flush();
    };
    if (P <= ZH_APX_CAP) {
        uint4 *list = pair_list[wid];
        uint32_t *rstart = row_start[wid];
        uint32_t mrows = 0;
#pragma unroll
        for (int j = 0; j < ZH_SCAN_NE; j++) {
            const uint32_t e = lane + 64u * j, c = eC[j];
            if (e < n_ent && e % T == 0) rstart[e / T] = off[j];  // the first entry of a row: its pairs start here
            if (c) {
                const uint32_t rl = e / T, gb = eGb[j];
                mrows |= 1u << rl;
                {
                    const uint64_t slot = eK0[j] + eWithin[j];
                    list[off[j]] = make_uint4(rl, eB0[j], (uint32_t)slot, (uint32_t)(slot >> 32));
                }
                for (uint32_t sidx = 1; sidx < c; sidx++) {
                    const ZhGroup *gp = groups + gb + sidx / GRP;
                    const uint64_t slot = gp->key_off[sidx % GRP] + eWithin[j];
                    list[off[j] + sidx] = make_uint4(rl, gp->b[sidx % GRP], (uint32_t)slot, (uint32_t)(slot >> 32));
                }
            }
        }
        if (lane == 0) rstart[nr] = P;
        uint32_t rowmask = 0;
#pragma unroll
        for (int bit = 0; bit < 16; bit++) rowmask |= (__ballot((mrows >> bit) & 1u) != 0 ? 1u : 0u) << bit;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- phase 2: rows with pairs RB at a time (one HBM round trip), a row's pairs G at a time -- group g takes pair p + g; the
        // pairs of the wave are ONE sequence (the next row's pairs follow the current row's in the list), so the queries of the
        // next step, whichever row it belongs to, are requested before the current step is computed ----
        uint32_t p = 0;
        uint4 rec = list[g < P ? g : P - 1];
        uint4 hq[NH];
        load_q(rec.y, hq);
        while (rowmask) {
            uint32_t rid[RB];
#pragma unroll
            for (int r = 0; r < RB; r++) {
                rid[r] = rowmask ? (uint32_t)__builtin_ctz(rowmask) : 0xFFFFFFFFu;
                rowmask &= rowmask - 1;
            }
            float4 v[RB][NR];
#pragma unroll
            for (int r = 0; r < RB; r++)
                if (rid[r] != 0xFFFFFFFFu) load_x(r0 + rid[r], v[r]);
#pragma unroll
            for (int r = 0; r < RB; r++) {
                if (rid[r] == 0xFFFFFFFFu) continue;
                const uint32_t pb = (uint32_t)__builtin_amdgcn_readfirstlane((int)rstart[rid[r] + 1]);
                const float a2 = row_sumsq(v[r]);
#ifndef ZH_APX_NO_PINGPONG
                // two steps per trip, the two query buffers swapping roles: no register copies except after a row's odd last step
                uint4 recn, hqn[NH];
                auto one = [&](const uint4 &ru, const uint4 *hu, uint4 &rp, uint4 *hp) {
                    const uint32_t pn = p + G < pb ? p + G : pb;
                    rp = list[pn + g < P ? pn + g : P - 1];
                    load_q(rp.y, hp);
                    const float s = dot_h(v[r], hu);
                    stash(s, a2, ru.y, ((uint64_t)ru.w << 32) | ru.z, p + g < pb);
                    p = pn;
                };
                while (p < pb) {
                    one(rec, hq, recn, hqn);
                    if (p < pb) one(recn, hqn, rec, hq);
                    else {
                        rec = recn;
#pragma unroll
                        for (int i = 0; i < NH; i++) hq[i] = hqn[i];
                    }
                }
#else
                while (p < pb) {
                    const uint32_t pn = p + G < pb ? p + G : pb;
                    const uint4 recn = list[pn + g < P ? pn + g : P - 1];
                    uint4 hqn[NH];
                    load_q(recn.y, hqn);
                    const float s = dot_h(v[r], hq);
                    stash(s, a2, rec.y, ((uint64_t)rec.w << 32) | rec.z, p + g < pb);
                    rec = recn;
#pragma unroll
                    for (int i = 0; i < NH; i++) hq[i] = hqn[i];
                    p = pn;
                }
#endif
            }
        }
    } else {
        // more pairs than the list holds (hot leaves): entry after entry, a leaf's visits G at a time, records from the group array
        float4 v[NR];
        float a2 = 0.f;
        uint32_t cur = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < ZH_SCAN_NE; j++) {
            unsigned long long m = __ballot(eC[j] != 0);
            while (m) {
                const int ll = __builtin_ctzll(m);
                m &= m - 1;
                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)eC[j], ll);
                const uint32_t gb = (uint32_t)__builtin_amdgcn_readlane((int)eGb[j], ll);
                const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)eWithin[j], ll);
                const uint32_t rl = ((uint32_t)ll + 64u * j) / T;
                if (rl != cur) {
                    cur = rl;
                    load_x(r0 + rl, v);
                    a2 = row_sumsq(v);
                }
                for (uint32_t s0 = 0; s0 < c; s0 += G) {
                    const uint32_t sidx = s0 + g < c ? s0 + g : c - 1;
                    const ZhGroup *gp = groups + gb + sidx / GRP;
                    const uint32_t b = gp->b[sidx % GRP];
                    uint4 hq[NH];
                    load_q(b, hq);
                    const float s = dot_h(v, hq);
                    stash(s, a2, b, gp->key_off[sidx % GRP] + w, s0 + g < c);
                }
            }
        }
    }
    flush();
}

// ---- the table scan on the matrix cores (round 4, second half; reworked in round 5) -- THE DEFAULT half-width scan with up to 16 trees at
// d = 256 / 384 / 512 / 768 / 1024 (zh_api.hip mfma_wanted; zh_set_sweep_mode(5) / ZH_NO_MFMA keep the VALU kernel above, which needs no copy of
// the rows) ----
// The VALU scan spends 36 vector instructions per pair (profiles/r04_pmc_scan_mix.txt): a 32-lane fma column, its reduce, its record and its
// stash.  Here a wave's 16 stored rows are the A operand of v_mfma_f32_16x16x32_f16, loaded from the index's fp16 copy of the rows
// (row_half_kernel: power-of-two scaled per row, tiles of 16 rows in the operand's own order -- + 2 d + 8 bytes of device memory per stored
// row, +50 % of the f32 table) and held in registers for all of the wave's pairs; the columns of B are queries.  Round 4: sixteen PAIRS of the
// wave's list were the 16 columns, 15 of 16 products waste, every pair its own d / 64 lines of query from L2 -- 3.4-3.5 ms per cfg3 launch
// against 4.4-4.55 for the VALU kernel (profiles/r04_ab_scan_mfma.txt), at the rate the L2s answer a CU's line requests (16.7 TB/s of
// requests).  Round 5: a column is a DISTINCT QUERY of the wave's pairs (the column pass below) and the scan's view of the rows is kept in the
// order that lets 16 neighbours share the most leaves (zh_order.hip): 0.89 columns per pair on iid rows, 0.61 on clustered rows whose ids are
// scattered, 0.40 on rows inserted cluster by cluster (profiles/r05_ab_row_order.txt): 3.4 / 3.0-3.1 / 2.5 ms per launch.
// Rounding the ROWS as well widens the intervals (the literal cosine key scores twice the rows exactly: 879 against 395 per query on a cfg4
// shard) -- paid back by the scan.
// k order: both operands use the SAME map (step s, k-group h, element j -> stored element 32 s + 4 h + (j & 3) + 16 (j >> 2)), which is
// all a dot product needs: a lane's eight query halves are ONE 16-byte load of qhalf_kernel<16>'s layout.
// What changes for the intervals: the row is rounded too.  |x - xh / sigma_x| <= rho |x| with rho MEASURED over the stored rows
// (row_half_kernel: the largest relative rounding error of any usable row, ~0.3 * 2^-11 on generic data, never above 2^-11), and the
// accumulation happens inside the MFMA in an order the ISA does not specify: four accumulators, each the target of D / 128 MFMAs of
// 32 products -- any order, any rounding of at most 2 u per operation gives |acc - sum| <= (33 D / 128 + 2) 2 u sum |xh_i h_i|
// (zh_approx_bound's `ops`; MEASURED on the device, tests/test_gpu_intervals.py: <= 9.2 u where 400-532 u are assumed).  Rows whose scale is
// out of range (or not finite) hand every pair the interval (-inf, +inf).
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

// The fp16 copy of the stored rows, written once (rows appended later: their own slots), in the A operand's own order: per tile of 16
// rows and MFMA step s one KiB = lane (c, h)'s eight halves {row 16 tile + c, elements 32 s + 4 h + t and 32 s + 16 + 4 h + t}, so that a
// scan wave's whole operand is D / 32 fully coalesced 1-KiB loads -- half the HBM bytes of the f32 rows, an eighth of the tag look-ups of
// loading them fragment-shaped, nothing to convert in the scan.  rowMeta[row] = {f32 |x|^2, 1 / sigma_x (NaN: scale out of range or a
// non-finite element: nothing is certain about this row)}; *rhoMax = the largest |x - xh / sigma_x| / |x| (f32 bits; atomicMax).
// perm (may be null): the scan's row order -- POSITION p < perm_rows holds stored row perm[p] (tree-0 leaf order, zh_api.hip build_scan_perm); tiles and
// rowMeta are indexed by position, which is all the scan ever uses.
__global__ __launch_bounds__(256) void row_half_kernel(const float *__restrict__ X, uint64_t row0, uint64_t n_rows, uint32_t d,
                                                       _Float16 *__restrict__ Xh, float2 *__restrict__ rowMeta, uint32_t *__restrict__ rhoMax,
                                                       const uint32_t *__restrict__ perm, uint64_t perm_rows) {
    const uint32_t lane = threadIdx.x & 63, NS = d / 32;
    const uint64_t nw = (uint64_t)gridDim.x * 4;
    float rho_w = 0.f;
    for (uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n_rows; i += nw) {
        const uint64_t row = row0 + i;  // the POSITION written
        const float *x = X + (size_t)(perm && row < perm_rows ? perm[row] : row) * d;
        float m = 0.f, a2 = 0.f;
        bool bad = false;
        for (uint32_t e = lane; e < d; e += 64) {
            const float v = x[e];
            bad |= !(v - v == 0.f);
            m = fmaxf(m, fabsf(v));
            a2 = __builtin_fmaf(v, v, a2);
        }
        m = wave_butterfly<OpMax>(m);
        a2 = wave_sum_canonical(a2);  // (d / 64 + 6 <= 32 roundings: what zh_approx_bound assumes of |x|^2)
        bad = __ballot(bad) != 0;
        bool usable = false;
        float sigma = 1.f, inv = __uint_as_float(0x7FC00000u);
        if (!bad) {
            int ex = 14;  // (a zero row: sigma = 1)
            if (m > 0.f) (void)frexpf(m, &ex);
            if (ex >= -100 && ex <= 100) { usable = true; sigma = ldexpf(1.f, 14 - ex); inv = ldexpf(1.f, ex - 14); }
        }
        // the rounding error in the row's scaled units (max in [2^13, 2^14)): nothing under- or overflows that matters at the 1.001
        float s2 = 0.f, d2 = 0.f;
        _Float16 *tile = Xh + ((size_t)(row >> 4) * NS * 64 + (row & 15)) * 8;
        for (uint32_t e = lane; e < d; e += 64) {
            const float v = x[e] * sigma;
            const _Float16 hv = usable ? f16_no_subnormal(v) : (_Float16)0.f;
            const float df = v - (float)hv;
            s2 = __builtin_fmaf(v, v, s2);
            d2 = __builtin_fmaf(df, df, d2);
            const uint32_t st = e >> 5, w = e & 31, hh = (w & 15) >> 2, j = (w & 3) + ((w >> 4) << 2);
            if (Xh) tile[((size_t)st * 64 + hh * 16) * 8 + j] = hv;  // (Xh null: the per-row scales, norms and the rounding error only -- the scan converts the f32 rows itself)
        }
        s2 = wave_sum_canonical(s2);
        d2 = wave_sum_canonical(d2);
        if (usable && s2 > 0.f) rho_w = fmaxf(rho_w, sqrtf(d2) * 1.001f / (sqrtf(s2) * 0.9999f));
        if (lane == 0) rowMeta[row] = make_float2(a2, inv);
    }
    if (lane == 0 && rho_w > 0.f) atomicMax(rhoMax, __float_as_uint(rho_w));
}

hipError_t zh_launch_row_half(const float *dX, uint64_t row0, uint64_t n_rows, uint32_t d, void *dXh, float2 *dRowMeta, uint32_t *dRhoMax,
                              const uint32_t *dPerm, uint64_t perm_rows, hipStream_t s) {
    if (!n_rows) return hipSuccess;
    const uint64_t blocks = std::min<uint64_t>((n_rows + 3) / 4, 256 * 32);
    hipLaunchKernelGGL(row_half_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, dX, row0, n_rows, d, (_Float16 *)dXh, dRowMeta, dRhoMax, dPerm, perm_rows);
    return hipGetLastError();
}

// out[p][t] = rowLeaf[perm[p]][t] for p < perm_rows, rowLeaf[p][t] beyond (rows appended after the order was made)
__global__ __launch_bounds__(256) void permute_row_leaf_kernel(const uint2 *__restrict__ rowLeaf, const uint32_t *__restrict__ perm, uint64_t perm_rows,
                                                               uint64_t n_entries, uint32_t T, uint2 *__restrict__ out) {
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = e / T, t = e % T;
        out[e] = rowLeaf[(size_t)(p < perm_rows ? perm[p] : p) * T + t];
    }
}
hipError_t zh_launch_permute_row_leaf(const uint2 *dRowLeaf, const uint32_t *dPerm, uint64_t perm_rows, uint64_t n_rows, uint32_t T, uint2 *dOut, hipStream_t s) {
    const uint64_t n = n_rows * T;
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(permute_row_leaf_kernel, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 256 * 64)), dim3(256), 0, s, dRowLeaf, dPerm, perm_rows, n, T, dOut);
    return hipGetLastError();
}

#ifndef ZH_MFMA_EXP
#define ZH_MFMA_EXP 0   // timing experiments of scan_mfma_kernel (diagnostic builds, results invalid): 1 no MFMA, 2 queries from L1, 3 no LDS staging writes, 4 no result stores, 5 no tile loop, 6 result stores into an L2-resident region
#endif
#ifndef ZH_MFMA_CL
#define ZH_MFMA_CL 0   // A/B: 128-byte query lines per column and chunk (0: by dimension)
#endif
// -DZH_SCAN_GUARD (diagnostic builds, never shipped): every index the matrix-core scan derives from its inputs is checked before it is used as an address
// -- global (bits 1, 2: a query past the batch, a key slot past the scratch) and, round 6, LDS (4 / 8: a regrouped record past the list / a column
// past the wave's columns; 16: a probe that does not end; 32: the pair list; 64: the column table; 128: a pair whose query pass 1 did not number);
// a violation sets a bit of ctl[7] (printed by zh_search_wait) instead of faulting: 1 a query id >= the batch, 2 a key slot >= the scratch, 4 a list
// position past the wave's list, 8 a column >= the wave's columns, 16 a probe sequence that does not end
#ifdef ZH_SCAN_GUARD
#define ZH_GUARD(cond, bit) ((cond) ? true : (atomicOr(guard, (bit)), false))
#else
#define ZH_GUARD(cond, bit) true
#endif
#ifdef ZH_SCAN_PROF  // tests/probes/scan_prof.py: where a wave of the matrix-core scan spends its cycles (never defined in the shipped build)
__device__ unsigned long long zh_scan_prof_buf[16];  // sums over every 16th wave: [0] waves [1] phase 1 [2] list [3] column pass [4] first chunk's wait [5] tile loop [6] tiles [7] pairs [8] columns [9] total
extern "C" __attribute__((visibility("default"))) int zh_debug_scan_prof(uint64_t *out, uint32_t words, int reset) {
    static const unsigned long long zero[16] = {};
    if (reset) return (int)hipMemcpyToSymbol(HIP_SYMBOL(zh_scan_prof_buf), zero, sizeof(zero));
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(zh_scan_prof_buf), (size_t)words * 8);
}
#define SP(...) __VA_ARGS__
#else
#define SP(...)
#endif
// Waves per SIMD: the compiler's own choice at d = 768 is 224 registers (every query line of a tile requested up front) = two waves; held to 168 it
// hoists a third of them and three waves fit (with `stage` in pair_list's words three blocks' LDS fits too) -- the third wave's preamble hides behind the
// others' tiles: 3.46 -> 3.35 ms per cfg3 launch, 192 -> 198 k QPS (same box, alternating).  d = 1024 spills at 168 (196 bytes of scratch): two waves.
#ifndef ZH_MFMA_WAVES
#define ZH_MFMA_WAVES 3   // A/B: 2 = the compiler's choice at d = 768
#endif
template <int D, bool F32ROWS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(D > 768 ? 2 : ZH_MFMA_WAVES, D > 768 ? 2 : ZH_MFMA_WAVES)))
void scan_mfma_kernel(const u32x4v *__restrict__ Xh, const float *__restrict__ Xf, const float2 *__restrict__ rowMeta,
                                                         const uint4 *__restrict__ Qh, const uint2 *__restrict__ rowLeaf, uint32_t T,
                                                         uint32_t RW, const uint32_t *__restrict__ visitBits,
                                                         const uint4 *__restrict__ nodeVisit, const ZhGroup *__restrict__ groups,
                                                         uint32_t GRP, uint64_t row_begin, uint64_t row_end, uint64_t *__restrict__ iv,
                                                         uint32_t *__restrict__ colStats /* {columns, pairs} of (a sample of) the listed waves */,
                                                         uint32_t nq, uint64_t iv_cap, uint32_t *__restrict__ guard) {
    constexpr int NS = D / 32;   // MFMA steps of a tile
    constexpr int NL = D / 64;   // 128-byte lines of a query = two steps each
    constexpr int CL = ZH_MFMA_CL ? ZH_MFMA_CL : (NL % 4 == 0 ? 2 : 3);  // lines per chunk: 2 CL load instructions, 2 CL KB of the wave's LDS
    constexpr int NCH = NL / CL;
    static_assert(D % 128 == 0 && NL % CL == 0 && NCH % 2 == 0, "four accumulators; the chunk registers alternate with a static parity");
    // pair records, 8 bytes: interval slot (36 bits) | query (24 bits; after the column pass: column (9) | position in its tile (9)) | row of the wave (4)
    __shared__ uint64_t pair_list[4][ZH_APX_CAP > CL * 256 ? ZH_APX_CAP : CL * 256];  // (... and, once the records are regrouped, the wave's chunk of query lines: `stage`)
    __shared__ uint64_t tile_list[4][ZH_APX_CAP];  // the same records grouped by tile of 16 columns; before that: the open-addressing table of the column pass (1024 words)
    __shared__ uint32_t col_query[4][ZH_APX_CAP];  // the query of every column (= distinct query of the wave's pairs)
    __shared__ uint32_t tile_start[4][36];         // pairs per tile -> exclusive scan
    __shared__ f32x4v acc_lds[4][64];              // a tile's 16 x 16 products: [column][row]
    // stage: a chunk of the tile's query halves, [line][column / 8][column % 8][16-byte piece, swizzled]: CL * 2 KiB per wave -- in pair_list's words, which are
    // done with when the tile loop starts (45 KB of LDS per block instead of 62: three blocks per CU fit)
    __shared__ float2 row_meta[4][16];          // {|x|^2, 1 / sigma_x (NaN: nothing is certain about this row)}
    const uint32_t lane = threadIdx.x & 63, c16 = lane & 15, h = lane >> 4;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wid;
    const uint64_t r0 = row_begin + wave * RW;
    if (r0 >= row_end) return;
    SP(const uint64_t sp0 = clock64();)
    const uint32_t nr = (uint32_t)(row_end - r0 < RW ? row_end - r0 : RW);
    const uint32_t n_ent = nr * T;
    const uint2 *__restrict__ ent = rowLeaf + (size_t)r0 * T;
    uint32_t eGb[ZH_SCAN_NE], eWithin[ZH_SCAN_NE], eC[ZH_SCAN_NE], off[ZH_SCAN_NE], eB0[ZH_SCAN_NE];
    uint64_t eK0[ZH_SCAN_NE];
    const uint32_t P = scan_phase1(lane, n_ent, ent, visitBits, nodeVisit, eGb, eWithin, eC, off, eB0, eK0);
    if (P == 0) return;
    SP(const uint64_t sp1 = clock64();)
    uint64_t *list = pair_list[wid];
    const bool listed = P <= ZH_APX_CAP;
    auto pack = [](uint32_t rl, uint32_t b, uint64_t slot) { return slot | ((uint64_t)b << 36) | ((uint64_t)rl << 60); };
#pragma unroll
    for (int j = 0; j < ZH_SCAN_NE; j++) {
        const uint32_t e = lane + 64u * j, c = eC[j];
        if (c && listed && ZH_GUARD(off[j] + c <= (uint32_t)ZH_APX_CAP && e / T < 16u, 32u)) {  // (LDS: the wave's pair list)
            const uint32_t rl = e / T, gb = eGb[j];
            list[off[j]] = pack(rl, eB0[j], eK0[j] + eWithin[j]);
            for (uint32_t sidx = 1; sidx < c; sidx++) {
                const ZhGroup *gp = groups + gb + sidx / GRP;
                list[off[j] + sidx] = pack(rl, gp->b[sidx % GRP], gp->key_off[sidx % GRP] + eWithin[j]);
            }
        }
    }
    SP(__builtin_amdgcn_s_waitcnt(0); const uint64_t sp2 = clock64();)
    // ---- the wave's 16 rows: their fp16 copy, already in the A operand's order (row_half_kernel): D / 32 coalesced 1-KiB loads ----
    f16x8 A[NS];
    if constexpr (!F32ROWS) {
        const u32x4v *tp = Xh + (size_t)(r0 >> 4) * (NS * 64) + lane;
#pragma unroll
        for (int st = 0; st < NS; st++) A[st] = __builtin_bit_cast(f16x8, __builtin_nontemporal_load(tp + 64 * st));
        if (lane < 16) row_meta[wid][lane] = rowMeta[r0 + (lane < nr ? lane : 0u)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        // ---- no fp16 copy of the table (it does not fit beside the rows: 64M x 768 on one GPU): the wave converts its 16 f32 rows itself, with the
        // per-row scale row_half_kernel would use (rowMeta, made without the copy) and the same rounding -- the operand is bit for bit the copy's tile.
        // 64 elements of every row at a time: four load instructions (lane (g, p): 16 bytes of row 4 i + g -- four 256-byte runs each), one 4-KiB
        // LDS image (the column pass's table, idle until then; pieces XOR-swizzled by row as in sweep128h_kernel), read back as the A operand's
        // k order: lane (c, h), step s: elements 32 s + 4 h .. + 3 and 32 s + 16 + 4 h .. + 3 of row c.
        if (lane < 16) row_meta[wid][lane] = rowMeta[r0 + (lane < nr ? lane : 0u)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float inv_c = row_meta[wid][c16].y;            // 1 / sigma of row c16 (NaN: unusable -> zeros, as the copy)
        const bool usable_c = inv_c - inv_c == 0.f;
        const float sigma_c = usable_c ? 1.0f / inv_c : 0.f; // (a power of two: exact)
        f32x4v *xt = reinterpret_cast<f32x4v *>(tile_list[wid]);
        const f32x4v *X4 = reinterpret_cast<const f32x4v *>(Xf);
        const uint32_t g4 = lane >> 4, p16 = lane & 15;
        auto cvt8 = [&](const f32x4v lo, const f32x4v hi) {
            f16x8 o;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                o[t] = usable_c ? f16_no_subnormal(lo[t] * sigma_c) : (_Float16)0.f;
                o[4 + t] = usable_c ? f16_no_subnormal(hi[t] * sigma_c) : (_Float16)0.f;
            }
            return o;
        };
        f32x4v Rr[4];
        auto issue_rows = [&](int kb) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t rr = 4u * i + g4;
                Rr[i] = __builtin_nontemporal_load(X4 + (size_t)(r0 + (rr < nr ? rr : nr - 1)) * (D / 4) + kb * 16 + p16);
            }
        };
        issue_rows(0);
#pragma unroll
        for (int kb = 0; kb < D / 64; kb++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t rr = 4u * i + g4;
                xt[rr * 16 + (p16 ^ rr)] = Rr[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (kb + 1 < D / 64) issue_rows(kb + 1);
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const f32x4v lo = xt[c16 * 16 + ((8u * s2 + h) ^ c16)], hi = xt[c16 * 16 + ((8u * s2 + 4u + h) ^ c16)];
                A[2 * kb + s2] = cvt8(lo, hi);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    const float2 *rmeta = row_meta[wid];
    // The query halves of a tile come in FULL 128-byte lines -- lane (g, p) = (lane >> 3, lane & 7) loads 16 bytes of column g's (then
    // column 8 + g's) query, eight lanes a line: eight tag look-ups per load instruction.  Loaded as the MFMA wants them (lane (c, h):
    // 16 bytes of column c, every lane of a quarter-wave another query) the same bytes cost 64 look-ups per instruction and the texture
    // path, not L2, set the pace: 6.9 ms per launch (profiles/r04_pmc_scan_mfma.txt, v1).  The lines pass
    // through the wave's own LDS chunk (written as loaded, pieces XOR-swizzled by column so that the fragment reads -- ds_read_b128, lane
    // (c, h) piece 4 (s & 1) + h of column c -- meet no bank twice within their 16-lane groups).
    u32x4v *stg = reinterpret_cast<u32x4v *>(pair_list[wid]);  // (native vectors: HIP's uint4 struct copies global -> private -> LDS stayed memcpys in scratch memory)
    const u32x4v *__restrict__ Qv = reinterpret_cast<const u32x4v *>(Qh);
    const uint32_t g8 = lane >> 3, pc = lane & 7;
    const uint32_t swA = (g8 >> 1) & 7u, swB = (4u + (g8 >> 1)) & 7u;          // columns g8 and 8 + g8
    const uint32_t rd0 = (c16 >> 3) * 64u + (c16 & 7u) * 8u + ((h ^ (c16 >> 1)) & 7u);   // even steps: piece h
    const uint32_t rd1 = (c16 >> 3) * 64u + (c16 & 7u) * 8u + (((4u + h) ^ (c16 >> 1)) & 7u);
    auto issue = [&](uint32_t bA, uint32_t bB, int chunk, u32x4v *dst) {
        if (!ZH_GUARD(bA < nq && bB < nq, 1u)) { bA = 0; bB = 0; }
#if ZH_MFMA_EXP == 2   // timing experiment (results invalid): eight queries only -> the query lines hit the vector L1
        bA &= 7u; bB &= 7u;
#endif
        const u32x4v *qa = Qv + (size_t)bA * (D / 8) + (size_t)(chunk * CL * 8) + (pc ^ swA);
        const u32x4v *qb = Qv + (size_t)bB * (D / 8) + (size_t)(chunk * CL * 8) + (pc ^ swB);
#pragma unroll
        for (int i = 0; i < CL; i++) { dst[2 * i] = qa[8 * i]; dst[2 * i + 1] = qb[8 * i]; }
    };
    auto to_lds = [&](const u32x4v *src) {
#if ZH_MFMA_EXP == 3   // timing experiment (results invalid): the staged lines are not written to LDS (one word keeps the loads alive)
        u32x4v f = src[0];
#pragma unroll
        for (int i = 1; i < 2 * CL; i++) f ^= src[i];
        if (f[0] == 0x12345678u) stg[lane] = f;
#else
#pragma unroll
        for (int i = 0; i < 2 * CL; i++) stg[64 * i + lane] = src[i];
#endif
    };
    auto mfma_chunk = [&](int chunk, f32x4v *acc) {
#pragma unroll
        for (int i = 0; i < 2 * CL; i++) {  // step 2 CL chunk + i: line i / 2 of the chunk
            const u32x4v v = stg[128 * (i / 2) + ((i & 1) ? rd1 : rd0)];
            const int st = chunk * 2 * CL + i;
#if ZH_MFMA_EXP == 1   // timing experiment (results invalid): no MFMA
            acc[st & 3] += __builtin_bit_cast(f32x4v, v) + __builtin_bit_cast(f32x4v, __builtin_shufflevector(A[st], A[st], 0, 1, 2, 3, 4, 5, 6, 7));
#else
            acc[st & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[st], __builtin_bit_cast(f16x8, v), acc[st & 3], 0, 0, 0);
#endif
        }
    };
    auto emit = [&](const f32x4v *acc, uint32_t rl, uint64_t slot, bool valid) {  // column c16 wants row rl: lane (c16, rl >> 2), register rl & 3
        const f32x4v t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        if (valid && (rl >> 2) == h) {
            const uint32_t i = rl & 3u;
            const float sv = i == 0 ? t[0] : (i == 1 ? t[1] : (i == 2 ? t[2] : t[3]));
            const float2 rm = rmeta[rl];
            if (ZH_GUARD(slot < iv_cap, 2u))
                __builtin_nontemporal_store(((uint64_t)__float_as_uint(rm.x) << 32) | __float_as_uint(sv * rm.y), iv + slot);
        }
    };
    if (listed) {
        // ---- a tile column is a distinct QUERY, not a pair (round 5; VERDICT r4 #4a).  The MFMA gives all 16 x 16 products of a tile's 16 rows and 16
        // columns; with a column per PAIR one of a column's 16 products was wanted and a query that visits several of the wave's rows -- every
        // cluster-mate of a planted neighbour on clustered data, two thirds of the pairs there -- was fetched (d / 64 lines) and multiplied once
        // per pair.  Column pass: the pairs' queries go through an open-addressing table in LDS (one ds_cmpst per pair and probe: the first
        // pair to claim a query leads and numbers the column), every pair learns its column, the records are regrouped by tile (a counting
        // sort over <= 32 tiles), and after a tile's MFMAs its accumulators pass through 1 KiB of LDS so that every pair of the tile -- any
        // (row, column) -- picks its product.  ~250 more LDS / VALU instructions per wave; one tile per 16 distinct queries instead of per 16 pairs.
        uint32_t *tab = reinterpret_cast<uint32_t *>(tile_list[wid]);  // 1024 words: (query + 1) << 9 | column; 0 = free
        uint32_t *colq = col_query[wid], *tstart = tile_start[wid];
#pragma unroll
        for (int i = 0; i < 4; i++) reinterpret_cast<uint4 *>(tab)[lane + 64 * i] = make_uint4(0u, 0u, 0u, 0u);
        if (lane < 36) tstart[lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        auto hash_of = [](uint32_t b) { return (b * 2654435761u) >> 22; };
        uint32_t nd = 0;  // distinct queries so far (wave-uniform)
        for (uint32_t p0 = 0; p0 < P; p0 += 64) {
            const uint32_t pi = p0 + lane;
            const bool valid = pi < P;
            const uint32_t b = valid ? (uint32_t)(list[pi] >> 36) & 0xFFFFFFu : 0u, key = b + 1u;
            uint32_t hs = hash_of(b);
            bool leader = false, open = valid;
            while (__ballot(open)) {
                if (open) {
                    const uint32_t old = atomicCAS(&tab[hs], 0u, (key << 9) | 0x1FFu);
                    if (old == 0u) { leader = true; open = false; }
                    else if ((old >> 9) == key) open = false;
                    else hs = (hs + 1u) & 1023u;
                }
            }
            const unsigned long long lm = __ballot(leader);
            if (leader) {
                const uint32_t col = nd + (uint32_t)__builtin_popcountll(lm & ((1ull << lane) - 1ull));
                if (ZH_GUARD(col < (uint32_t)ZH_APX_CAP && hs < 1024u, 64u)) {  // (LDS: the column table, the columns' queries)
                    tab[hs] = (key << 9) | col;
                    colq[col] = b;
                }
            }
            nd += (uint32_t)__builtin_popcountll(lm);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // every pair: its column (a read-only probe), its position among its tile's pairs
        for (uint32_t p0 = 0; p0 < P; p0 += 64) {
            const uint32_t pi = p0 + lane;
            if (pi < P) {
                const uint64_t rec = list[pi];
                const uint32_t b = (uint32_t)(rec >> 36) & 0xFFFFFFu, key = b + 1u;
                uint32_t hs = hash_of(b), w = tab[hs];
                for (uint32_t tries = 0; (w >> 9) != key && ZH_GUARD(tries < 1024u, 16u); tries++) { (void)tries; hs = (hs + 1u) & 1023u; w = tab[hs]; }
                const uint32_t col = w & 0x1FFu;
                if (!ZH_GUARD((w >> 9) == key && col < nd, 128u)) continue;  // (a pair whose query is not in the table: pass 1 numbers every query)
                const uint32_t pos = atomicAdd(&tstart[1 + (col >> 4)], 1u);
                list[pi] = (rec & ~(0xFFFFFFull << 36)) | ((uint64_t)(col | (pos << 9)) << 36);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {   // tstart[1 + t] = pairs of tile t -> tstart[t] = first record of tile t, tstart[32] = P
            uint32_t v = lane < 33 ? tstart[lane] : 0u;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t u = __shfl_up(v, o);
                if (lane >= (uint32_t)o) v += u;
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < 33) tstart[lane] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint64_t *tl = tile_list[wid];  // (the table is done with: its words become the regrouped records)
        {
            uint64_t recs[ZH_APX_CAP / 64];
#pragma unroll
            for (int i = 0; i < ZH_APX_CAP / 64; i++) recs[i] = (uint32_t)(64 * i) + lane < P ? list[64 * i + lane] : ~0ull;
            __builtin_amdgcn_wave_barrier();  // (every probe of the table has been answered: pass 2 is complete)
#pragma unroll
            for (int i = 0; i < ZH_APX_CAP / 64; i++)
                if ((uint32_t)(64 * i) + lane < P) {
                    const uint32_t cp = (uint32_t)(recs[i] >> 36) & 0xFFFFFFu, col = cp & 0x1FFu, pos = (cp >> 9) & 0x1FFu;
                    if (ZH_GUARD(tstart[col >> 4] + pos < (uint32_t)ZH_APX_CAP && col < nd, 4u | (col < nd ? 0u : 8u))) tl[tstart[col >> 4] + pos] = recs[i];
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the wave's distinct queries sixteen at a time; the next chunk of query lines is requested before the current one is multiplied,
        // across tile boundaries
#if ZH_MFMA_EXP == 5   // timing experiment (results invalid): no tile loop -- phase 1, pair list, column pass and the rows' tile loads only
        const uint32_t nt = (nd + 15) / 16 > 4096u ? 1u : 0u;
        if (lane == 0 && __builtin_bit_cast(u32x4v, A[NS - 1])[0] == 0x12345678u) iv[0] = 0;
#else
        const uint32_t nt = (nd + 15) / 16;
#endif
        SP(const uint64_t sp3 = clock64();)
        // what the sharing is worth (zh_stats_t::approx_columns / approx_column_pairs): every 64th wave reports -- same-address atomics from all
        // 200k waves of a launch serialise in one L2 channel (measured: 3.6 -> 5.5 ms per launch)
        if (colStats && lane == 0 && (wave & 63u) == 0u) { atomicAdd(&colStats[0], nd); atomicAdd(&colStats[1], P); }
        auto col_b = [&](uint32_t t, uint32_t col) { const uint32_t ci = 16 * t + col; return colq[ci < nd ? ci : nd - 1]; };
        uint32_t bA = col_b(0, g8), bB = col_b(0, 8 + g8);
        u32x4v ra[2 * CL], rb[2 * CL];
        issue(bA, bB, 0, ra);
        SP(__builtin_amdgcn_s_waitcnt(0); const uint64_t sp4 = clock64();)
        f32x4v *al = acc_lds[wid];
        for (uint32_t t = 0; t < nt; t++) {
            f32x4v acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int c = 0; c < NCH; c += 2) {
                issue(bA, bB, c + 1, rb);
                to_lds(ra);
                mfma_chunk(c, acc);
                if (c + 2 < NCH) issue(bA, bB, c + 2, ra);
                else {
                    bA = col_b(t + 1, g8); bB = col_b(t + 1, 8 + g8);
                    issue(bA, bB, 0, ra);
                }
                to_lds(rb);
                mfma_chunk(c + 1, acc);
            }
            // lane (column c16, h): the products of rows 4 h .. 4 h + 3 -> acc_lds[column][row]; then every pair of the tile picks its own
            al[c16 * 4 + h] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t pe = tstart[t + 1];
            for (uint32_t pi = tstart[t] + lane; pi < pe; pi += 64) {
                const uint64_t rec = tl[pi];
                const uint32_t rl = (uint32_t)(rec >> 60), col = (uint32_t)(rec >> 36) & 15u;
                const float sv = reinterpret_cast<const float *>(al)[col * 16 + rl];
                const float2 rm = rmeta[rl];
#if ZH_MFMA_EXP == 4   // timing experiment (results invalid): one pair in 64 stores its result
                if ((rec & 63u) == 0u)
#endif
#if ZH_MFMA_EXP == 6   // timing experiment (results invalid): every result store lands in 2 MiB that stay in the L2s (the stores' issue and acknowledgement without their HBM side)
                __builtin_nontemporal_store(((uint64_t)__float_as_uint(rm.x) << 32) | __float_as_uint(sv * rm.y), iv + ((wave & 4095u) * 64u + lane));
#else
                if (ZH_GUARD((rec & 0xFFFFFFFFFull) < iv_cap, 2u))
                    __builtin_nontemporal_store(((uint64_t)__float_as_uint(rm.x) << 32) | __float_as_uint(sv * rm.y), iv + (rec & 0xFFFFFFFFFull));
#endif
            }
            __builtin_amdgcn_wave_barrier();
        }
        SP(if ((wave & 15u) == 0u && lane == 0) {
            __builtin_amdgcn_s_waitcnt(0);
            const uint64_t sp5 = clock64();
            atomicAdd(&zh_scan_prof_buf[0], 1ull); atomicAdd(&zh_scan_prof_buf[1], sp1 - sp0); atomicAdd(&zh_scan_prof_buf[2], sp2 - sp1);
            atomicAdd(&zh_scan_prof_buf[3], sp3 - sp2); atomicAdd(&zh_scan_prof_buf[4], sp4 - sp3); atomicAdd(&zh_scan_prof_buf[5], sp5 - sp4);
            atomicAdd(&zh_scan_prof_buf[6], (unsigned long long)nt); atomicAdd(&zh_scan_prof_buf[7], (unsigned long long)P);
            atomicAdd(&zh_scan_prof_buf[8], (unsigned long long)nd); atomicAdd(&zh_scan_prof_buf[9], sp5 - sp0);
        })
    } else {
        // more pairs than the list holds (hot leaves): entry after entry, a leaf's visits sixteen at a time, records from the group array
#pragma unroll
        for (int j = 0; j < ZH_SCAN_NE; j++) {
            unsigned long long m = __ballot(eC[j] != 0);
            while (m) {
                const int ll = __builtin_ctzll(m);
                m &= m - 1;
                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)eC[j], ll);
                const uint32_t gb = (uint32_t)__builtin_amdgcn_readlane((int)eGb[j], ll);
                const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)eWithin[j], ll);
                const uint32_t rl = ((uint32_t)ll + 64u * j) / T;
                for (uint32_t s0 = 0; s0 < c; s0 += 16) {
                    auto visit_b = [&](uint32_t col) { const uint32_t si = s0 + col < c ? s0 + col : c - 1; return groups[gb + si / GRP].b[si % GRP]; };
                    const uint32_t bA = visit_b(g8), bB = visit_b(8 + g8);
                    const uint32_t sidx = s0 + c16 < c ? s0 + c16 : c - 1;
                    const ZhGroup *gp = groups + gb + sidx / GRP;
                    f32x4v acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                    u32x4v rq[2 * CL];
#pragma unroll
                    for (int cc = 0; cc < NCH; cc++) {
                        issue(bA, bB, cc, rq);
                        to_lds(rq);
                        mfma_chunk(cc, acc);
                        __builtin_amdgcn_sched_barrier(0);  // (the rare path must not set the kernel's register count: all chunks' loads hoisted)
                    }
                    emit(acc, rl, gp->key_off[sidx % GRP] + w, s0 + c16 < c);
                }
            }
        }
    }
}

template <int D>
static hipError_t launch_scan_mfma_d(const float *dX, uint64_t n_rows, const ZhApprox &ap, const uint2 *dRowLeaf, uint32_t T,
                                     const uint32_t *dVisitBits, const uint4 *dNodeVisit, const ZhGroup *dGroups, uint32_t group, hipStream_t s) {
    const uint32_t RW = zh_scan_rows_per_wave(T);
    uint64_t rows_per_launch = zh_sweep_rows_per_launch(D);
    rows_per_launch = rows_per_launch / (4 * RW) * (4 * RW);
    for (uint64_t r = 0; r < n_rows; r += rows_per_launch) {
        const uint64_t r_end = r + rows_per_launch < n_rows ? r + rows_per_launch : n_rows;
        const uint64_t waves = (r_end - r + RW - 1) / RW, blocks = (waves + 3) / 4;
        if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
        if (ap.row_half)
            hipLaunchKernelGGL((scan_mfma_kernel<D, false>), dim3((uint32_t)blocks), dim3(256), 0, s, (const u32x4v *)ap.row_half, dX, ap.row_meta, (const uint4 *)ap.Qh, dRowLeaf, T, RW,
                               dVisitBits, dNodeVisit, dGroups, group, r, r_end, ap.iv, ap.ctl + 5, ap.n_queries, ap.iv_cap, ap.ctl + 7);
        else  // no fp16 copy (no room for it): the scan converts the f32 rows itself (rows in id order)
            hipLaunchKernelGGL((scan_mfma_kernel<D, true>), dim3((uint32_t)blocks), dim3(256), 0, s, (const u32x4v *)nullptr, dX, ap.row_meta, (const uint4 *)ap.Qh, dRowLeaf, T, RW,
                               dVisitBits, dNodeVisit, dGroups, group, r, r_end, ap.iv, ap.ctl + 5, ap.n_queries, ap.iv_cap, ap.ctl + 7);
    }
    return hipGetLastError();
}

// ---- d = 128, leaf by leaf, at half width: the sweep of SIFT-style shards (cfg5) on the matrix cores ----
// The leaf-major sweep of 512-byte rows runs at what HBM gives random rows (0.82-0.88 of the measured gather ceiling, round 3), and a random
// 256-byte row costs HBM half of a 512-byte one (profiles/micro/gather512.hip: 6.6 TB/s for both): the lever is the ROW's bytes.  The index
// keeps an fp16 copy of its rows (row-major, ONE power-of-two scale for the table: integer-valued SIFT rows are exact; a row that does not
// survive the common scale within 2^-10 of its norm -- or holds a non-finite element -- is stored as NaNs and lands on the exact path), the
// batch's queries get qhalf_kernel's fp16 copy in natural order, and a wave takes its 64 flat rows as four MFMA tiles: the queries of up to
// four groups are the 16 rows of A (group g of the set in rows 4 g .. 4 g + 3), 16 stored rows the columns of B, 4 MFMAs (K = 128) give
// every (query, row) product of the tile, and lane (c, h) -- stored row c, group h of the set -- stores its group's results where the f32
// sweep stores keys: the raw {x^ . h^ / sigma_X, |x^|^2 / sigma_X^2} that select_tau_kernel turns into intervals.  Rows are loaded whole
// (four 256-byte rows per instruction), pass through the wave's LDS tile (XOR-swizzled 16-byte pieces) and come back as B fragments.
// |x|^2 comes from the ROUNDED row: approx_interval's rho_n term.
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ X, uint64_t n, uint32_t *__restrict__ out) {
    float m = 0.f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float v = fabsf(X[i]);
        if (v - v == 0.f) m = fmaxf(m, v);  // (finite elements only)
    }
    m = wave_butterfly<OpMax>(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}

__global__ __launch_bounds__(256) void row_half128_kernel(const float *__restrict__ X, uint64_t row0, uint64_t n_rows, float sigma,
                                                          _Float16 *__restrict__ Xh, uint32_t *__restrict__ rhoMax) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t nw = (uint64_t)gridDim.x * 4;
    float rho_w = 0.f;
    for (uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n_rows; i += nw) {
        const uint64_t row = row0 + i;
        const float2 x = reinterpret_cast<const float2 *>(X + (size_t)row * 128)[lane];
        const float v0 = x.x * sigma, v1 = x.y * sigma;
        bool bad = !(v0 - v0 == 0.f) || !(v1 - v1 == 0.f) || fabsf(v0) > 60000.f || fabsf(v1) > 60000.f;
        const _Float16 h0 = f16_no_subnormal(v0), h1 = f16_no_subnormal(v1);
        const float d0 = v0 - (float)h0, d1 = v1 - (float)h1;
        const float s2 = wave_sum_canonical(__builtin_fmaf(v0, v0, v1 * v1));
        const float d2 = wave_sum_canonical(__builtin_fmaf(d0, d0, d1 * d1));
        const float m = wave_butterfly<OpMax>(fmaxf(fabsf(x.x), fabsf(x.y)));
        bad = __ballot(bad) != 0 || !(s2 - s2 == 0.f) || (s2 == 0.f && m > 0.f);
        const float rho = s2 > 0.f ? sqrtf(d2) * 1.001f / (sqrtf(s2) * 0.9999f) : 0.f;
        bad = bad || !(rho <= 9.765625e-4f);
        if (!bad) rho_w = fmaxf(rho_w, rho);
        const _Float16 nanh = (_Float16)__uint_as_float(0x7FC00000u);
        typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
        reinterpret_cast<f16x2v *>(Xh + (size_t)row * 128)[lane] = bad ? f16x2v{nanh, nanh} : f16x2v{h0, h1};
    }
    if (lane == 0 && rho_w > 0.f) atomicMax(rhoMax, __float_as_uint(rho_w));
}

hipError_t zh_launch_absmax(const float *dX, uint64_t n, uint32_t *dOut, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(absmax_kernel, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 256 * 16)), dim3(256), 0, s, dX, n, dOut);
    return hipGetLastError();
}
hipError_t zh_launch_row_half128(const float *dX, uint64_t row0, uint64_t n_rows, float sigma, void *dXh, uint32_t *dRhoMax, hipStream_t s) {
    if (!n_rows) return hipSuccess;
    hipLaunchKernelGGL(row_half128_kernel, dim3((uint32_t)std::min<uint64_t>((n_rows + 3) / 4, 256 * 32)), dim3(256), 0, s, dX, row0, n_rows, sigma,
                       (_Float16 *)dXh, dRhoMax);
    return hipGetLastError();
}

// ---- BYTE rows (round 6).  A table whose every element is an integer in 0 .. 255 -- SIFT descriptors, BASELINE's "SIFT-style" cfg5 -- has an EXACT
// copy of 128 bytes per row: half the bytes of the fp16 copy for a sweep that is bound by the bytes it gathers, and no rounding of the row at all
// (rho = 0 in approx_interval; |x|^2 <= 128 * 255^2 < 2^24 is exact in the Gram diagonal's f32).  A thread per four elements; any element that is
// not such an integer (a fraction, a negative, > 255, NaN) raises *notBytes and the caller falls back to the fp16 copy.
__global__ __launch_bounds__(256) void row_byte128_kernel(const float4 *__restrict__ X4, uint64_t n4, uint32_t *__restrict__ Xb,
                                                          uint32_t *__restrict__ notBytes) {
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) {
        const float4 v = X4[i];
        const float e[4] = {v.x, v.y, v.z, v.w};
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t b = (uint32_t)fminf(fmaxf(e[j], 0.f), 255.f);  // (NaN -> 0, then caught by the comparison)
            bad |= !((float)b == e[j]);
            w |= b << (8 * j);
        }
        Xb[i] = w;
    }
    if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(notBytes, 1u);
}
hipError_t zh_launch_row_byte128(const float *dX, uint64_t row0, uint64_t n_rows, void *dXb, uint32_t *dNotBytes, hipStream_t s) {
    if (!n_rows) return hipSuccess;
    const uint64_t n4 = n_rows * 32;
    hipLaunchKernelGGL(row_byte128_kernel, dim3((uint32_t)std::min<uint64_t>((n4 + 255) / 256, 256 * 32)), dim3(256), 0, s,
                       reinterpret_cast<const float4 *>(dX + (size_t)row0 * 128), n4, reinterpret_cast<uint32_t *>(dXb) + (size_t)row0 * 32, dNotBytes);
    return hipGetLastError();
}
// eight stored bytes (two words) -> the eight halves of an MFMA fragment, exactly: 0x6400 | b is the half 1024 + b (one ulp is 1 from 1024 to
// 2048), and (1024 + b) - 1024 is exact: a v_perm_b32 and a v_pk_add_f16 per two elements
__device__ __forceinline__ f16x8 bytes8_to_f16(uint32_t w0, uint32_t w1) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 off = {(_Float16)-1024.f, (_Float16)-1024.f};
    const h2 a = __builtin_bit_cast(h2, __builtin_amdgcn_perm(0x64646464u, w0, 0x04010400u)) + off;
    const h2 b = __builtin_bit_cast(h2, __builtin_amdgcn_perm(0x64646464u, w0, 0x04030402u)) + off;
    const h2 c = __builtin_bit_cast(h2, __builtin_amdgcn_perm(0x64646464u, w1, 0x04010400u)) + off;
    const h2 d = __builtin_bit_cast(h2, __builtin_amdgcn_perm(0x64646464u, w1, 0x04030402u)) + off;
    return f16x8{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}

typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot8_self(const f16x8 v, float c) {
    c = __builtin_amdgcn_fdot2(__builtin_shufflevector(v, v, 0, 1), __builtin_shufflevector(v, v, 0, 1), c, false);
    c = __builtin_amdgcn_fdot2(__builtin_shufflevector(v, v, 2, 3), __builtin_shufflevector(v, v, 2, 3), c, false);
    c = __builtin_amdgcn_fdot2(__builtin_shufflevector(v, v, 4, 5), __builtin_shufflevector(v, v, 4, 5), c, false);
    return __builtin_amdgcn_fdot2(__builtin_shufflevector(v, v, 6, 7), __builtin_shufflevector(v, v, 6, 7), c, false);
}

#ifndef ZH_S128H_WAVES
#define ZH_S128H_WAVES 0   // A/B: force this many waves per SIMD (registers spill past 4)
#endif
template <int CH>
__global__ __launch_bounds__(256)
#if ZH_S128H_WAVES
__attribute__((amdgpu_waves_per_eu(ZH_S128H_WAVES, ZH_S128H_WAVES)))
#endif
void sweep128h_kernel(const u32x4v *__restrict__ Xh, const u32x4v *__restrict__ Qh, float inv,
                                                         const ZhGroup *__restrict__ groups, const uint64_t *__restrict__ groupRowOff,
                                                         uint64_t n_groups, const uint32_t *__restrict__ waveGroup,
                                                         const uint32_t *__restrict__ leaf_ids, uint64_t row_begin, uint64_t R_grouped,
                                                         uint64_t *__restrict__ iv) {
    __shared__ u32x4v rows_lds[4][16 * 16];  // per wave: ONE tile = 16 rows x 16 pieces of 16 bytes, piece p of row R at R * 16 + (p ^ R)
    const uint32_t lane = threadIdx.x & 63, c16 = lane & 15, h = lane >> 4;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wid;
    uint64_t r0 = row_begin + wave * (64 * CH);
    if (r0 >= R_grouped) return;
    uint32_t cnt = (uint32_t)(R_grouped - r0 < 64 ? R_grouped - r0 : 64);
    uint32_t my_g, my_id, my_within, my_off, my_len;
    resolve_flat_rows(r0, cnt, lane, groups, groupRowOff, n_groups, waveGroup, leaf_ids, my_g, my_id, my_within, &my_off, &my_len);
    u32x4v *tl = rows_lds[wid];
    f16x8 Aq[4];
#pragma unroll
    for (int st = 0; st < 4; st++) Aq[st] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t a_g0 = 0xFFFFFFFFu;  // the groups whose queries A holds: a_g0 .. a_g0 + 3 (wave-uniform)
    // a tile's rows, whole: instruction i = the tile's rows 4 i .. 4 i + 3, lane (h, c16) the 16-byte piece c16 of row 4 i + h.  ONE register
    // set: once a tile is written to LDS its registers take the next tile's loads (across chunk boundaries, where the next chunk's rows
    // are known), which travel while this tile is multiplied.
    u32x4v R[4];
    uint32_t c_rg = 0xFFFFFFFFu, c_hi = 0, c_lo[4] = {0, 0, 0, 0};  // the group record this lane's column last stored through: group, key_off[0..3] (low words; high nibbles | gsize << 16)
    auto issue_tile = [&](uint32_t ids, uint32_t n, uint32_t t) {  // flat rows 16 t .. 16 t + 15 of a chunk of n rows whose ids the lanes hold
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t fr = 16u * t + 4u * i + h;
            const uint32_t id = (uint32_t)__shfl((int)ids, (int)(fr < n ? fr : n - 1));
            R[i] = __builtin_nontemporal_load(Xh + (size_t)id * 16 + c16);
        }
    };
    issue_tile(my_id, cnt, 0);
    for (int c = 0; c < CH; c++) {
        // the next chunk: does it lie entirely in the group this chunk's last row belongs to?  (wave-uniform; as sweep128_kernel)
        const uint64_t r0n = r0 + 64;
        const bool have_next = c + 1 < CH && r0n < R_grouped;
        const uint32_t cntn = have_next ? (uint32_t)(R_grouped - r0n < 64 ? R_grouped - r0n : 64) : 0;
        const uint32_t lw = (uint32_t)__builtin_amdgcn_readlane((int)my_within, (int)cnt - 1) + 1;
        const uint32_t loff = (uint32_t)__builtin_amdgcn_readlane((int)my_off, (int)cnt - 1);
        const uint32_t llen = (uint32_t)__builtin_amdgcn_readlane((int)my_len, (int)cnt - 1);
        const uint32_t lg = (uint32_t)__builtin_amdgcn_readlane((int)my_g, (int)cnt - 1);
        const bool fast = have_next && cnt == 64 && (uint64_t)lw + cntn <= llen;
        uint32_t nxt_id = 0;
        if (fast) {
            const uint32_t wn = lw + (lane < cntn ? lane : cntn - 1);
            nxt_id = leaf_ids ? leaf_ids[(size_t)loff + wn] : loff + wn;
        }
        const uint32_t ntile = (cnt + 15) / 16;
#pragma unroll 1
        for (uint32_t t = 0; t < ntile; t++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t rr = 4u * i + h;
                tl[rr * 16 + (c16 ^ rr)] = R[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // this lane's column: flat row fr of the chunk, its group and where its results go.  The group's record is kept in registers and
            // fetched again only when the group changes (a leaf is thousands of rows) -- BEFORE the next tile's rows are requested: loads return in
            // order, and a fetch behind them (round 4: gsize and every key_off in the epilogue, each with its own wait) held every tile's stores
            // until the NEXT tile's rows had arrived from HBM, then paid four more round trips
            const uint32_t fr = 16 * t + c16;
            const uint32_t rg = (uint32_t)__shfl((int)my_g, (int)(fr < cnt ? fr : cnt - 1));
            const uint32_t rw = (uint32_t)__shfl((int)my_within, (int)(fr < cnt ? fr : cnt - 1));
            if (rg != c_rg) {
                const uint4 *gq = reinterpret_cast<const uint4 *>(groups + rg);
                const uint4 g0 = gq[0], k01 = gq[2], k23 = gq[3];
                c_rg = rg;
                c_lo[0] = k01.x; c_lo[1] = k01.z; c_lo[2] = k23.x; c_lo[3] = k23.z;  // (key slices are < 2^36: the four high nibbles and gsize share a word)
                c_hi = (k01.y & 15u) | ((k01.w & 15u) << 4) | ((k23.y & 15u) << 8) | ((k23.w & 15u) << 12) | (g0.z << 16);
            }
            if (t + 1 < ntile) issue_tile(my_id, cnt, t + 1);
            else if (fast) issue_tile(nxt_id, cntn, 0);
            f16x8 Bf[4];
#pragma unroll
            for (int st = 0; st < 4; st++) Bf[st] = __builtin_bit_cast(f16x8, tl[c16 * 16 + ((4u * st + h) ^ c16)]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t last = 16 * t + 15 < cnt ? 16 * t + 15 : cnt - 1;
            const uint32_t g_first = (uint32_t)__builtin_amdgcn_readlane((int)my_g, (int)(16 * t));
            const uint32_t g_last = (uint32_t)__builtin_amdgcn_readlane((int)my_g, (int)last);
            float a2 = 0.f;
#pragma unroll
            for (int st = 0; st < 4; st++) a2 = dot8_self(Bf[st], a2);
            a2 = xor16<OpAdd>(a2);
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a2), __float_as_uint(a2), false, false);
                a2 = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
            for (uint32_t gp = g_first; gp <= g_last; gp += 4) {
                if (gp != a_g0) {  // (wave-uniform) the queries of groups gp .. gp + 3: A's row m = slot m & 3 of group gp + (m >> 2)
                    a_g0 = gp;
                    const uint32_t grp = gp + (c16 >> 2);
                    const bool on = grp < n_groups && (c16 & 3u) < groups[grp < n_groups ? grp : 0].gsize;
                    const uint32_t b = on ? groups[grp].b[c16 & 3u] : 0u;
#pragma unroll
                    for (int st = 0; st < 4; st++) {
                        const u32x4v v = Qh[(size_t)b * 16 + 4 * st + h];
                        Aq[st] = on ? __builtin_bit_cast(f16x8, v) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    }
                }
                f32x4v acc[4];
#pragma unroll
                for (int st = 0; st < 4; st++) acc[st] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aq[st], Bf[st], f32x4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const f32x4v dsum = (acc[0] + acc[1]) + (acc[2] + acc[3]);  // [i] = query slot i of group gp + h against stored row c16 of the tile
                if (fr < cnt && rg == gp + h) {
                    const uint32_t a2b = __float_as_uint(a2 * inv * inv);
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if ((uint32_t)i < (c_hi >> 16))
                            __builtin_nontemporal_store(((uint64_t)a2b << 32) | __float_as_uint(dsum[i] * inv),
                                                        iv + ((((uint64_t)((c_hi >> (4 * i)) & 15u)) << 32) | c_lo[i]) + rw);
                }
            }
        }
        if (!have_next) break;
        r0 = r0n; cnt = cntn;
        if (fast) {
            my_g = lg; my_off = loff; my_len = llen;
            my_within = lw + (lane < cntn ? lane : cntn - 1);
            my_id = nxt_id;
        } else {
            resolve_flat_rows(r0, cnt, lane, groups, groupRowOff, n_groups, waveGroup, leaf_ids, my_g, my_id, my_within, &my_off, &my_len);
            issue_tile(my_id, cnt, 0);
        }
    }
}

// ---- the same sweep with the rows' tiles requested STRAIGHT INTO LDS (global_load_lds_dwordx4) and TWO tiles in flight per wave.
// The kernel above keeps one tile travelling in 16 registers while the previous one is multiplied: 16 waves x 4 KiB = 64 KiB in flight per CU, and at
// the latency a loaded HBM answers random 256-byte rows with that is 4.2-4.3 TB/s (a wave spends ~8000 cycles per tile, three quarters of them
// waiting).  Here a tile needs no registers while it travels: a wave owns two 4-KiB tile buffers, tile t + 2 is requested (into the buffer tile t
// was just read from) as soon as tile t's results are stored, so two tiles -- 128 KiB per CU at the same four waves per SIMD -- are on their way
// while one is multiplied.  The XOR swizzle of the LDS image moves to the SOURCE side (lane (h, c) asks for piece c ^ row of its row: the DMA writes
// lane after lane), the fragment reads are unchanged.  The compiler does not order LDS reads against LDS-DMA (it hoisted them above its own
// waits in a probe): every wait of the steady state is explicit -- loads and stores of a wave complete in issue order, so "at most four VMEM
// instructions outstanding" behind a younger tile's four, or none, says the tile has landed -- and anything else a wave loads meanwhile (the next
// chunk's ids, a new group's record: rare) only makes those waits stricter.
#define ZH_DMA16(src, dst) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src), (__attribute__((address_space(3))) void *)(dst), 16, 0, 0)
template <int CH>
__global__ __launch_bounds__(256) void sweep128h_dma_kernel(const u32x4v *__restrict__ Xh, const u32x4v *__restrict__ Qh, float inv,
                                                             const ZhGroup *__restrict__ groups, const uint64_t *__restrict__ groupRowOff,
                                                             uint64_t n_groups, const uint32_t *__restrict__ waveGroup,
                                                             const uint32_t *__restrict__ leaf_ids, uint64_t row_begin, uint64_t R_grouped,
                                                             uint64_t *__restrict__ iv) {
    __shared__ u32x4v rows_lds[4][2][16 * 16];  // per wave: TWO tiles = 16 rows x 16 pieces of 16 bytes, piece p of row R at R * 16 + (p ^ R)
    __shared__ uint32_t ids_lds[4][64];         // per wave: the next chunk's row ids (requested the same way: no wait of the compiler's for them)
    const uint32_t lane = threadIdx.x & 63, c16 = lane & 15, h = lane >> 4;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wid;
    uint64_t r0 = row_begin + wave * (64 * CH);
    if (r0 >= R_grouped) return;
    uint32_t cnt = (uint32_t)(R_grouped - r0 < 64 ? R_grouped - r0 : 64);
    uint32_t my_g, my_id, my_within, my_off, my_len;
    resolve_flat_rows(r0, cnt, lane, groups, groupRowOff, n_groups, waveGroup, leaf_ids, my_g, my_id, my_within, &my_off, &my_len);
    f16x8 Aq[4];
#pragma unroll
    for (int st = 0; st < 4; st++) Aq[st] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t a_g0 = 0xFFFFFFFFu;  // the groups whose queries A holds: a_g0 .. a_g0 + 3 (wave-uniform)
    uint32_t c_rg = 0xFFFFFFFFu, c_hi = 0, c_lo[4] = {0, 0, 0, 0};  // the group record this lane's column last stored through (as above)
    // the issuing side runs up to two tiles ahead of the consuming side: its chunk's ids / row count, the next tile of it, and the chunk after it
    // (known when the same leaf group continues)
    uint32_t I_ids = my_id, I_cnt = cnt, I_t = 0, N_ids = 0, N_cnt = 0;
    bool N_known = false;
    uint32_t issued = 0, consumed = 0;  // tiles (wave-uniform); tile n lives in buffer n & 1
    // LDS byte addresses: the wave's tile buffers, its id words, the lane's four fragment pieces of a tile ((c, h): piece 4 st + h of row c)
    const uint32_t tile_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)&rows_lds[wid][0][0];
    const uint32_t ids_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)&ids_lds[wid][0] + 4u * lane;
    uint32_t frag_off[4];
#pragma unroll
    for (int st = 0; st < 4; st++) frag_off[st] = (c16 * 16u + ((4u * st + h) ^ c16)) * 16u;
    auto issue_next = [&]() {
        if (16u * I_t >= I_cnt) {
            if (!N_known) return;
            I_ids = N_ids; I_cnt = N_cnt; I_t = 0; N_known = false;
        }
        u32x4v *dst = rows_lds[wid][issued & 1u];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t rr = 4u * i + h, fr = 16u * I_t + rr;
            const uint32_t id = (uint32_t)__shfl((int)I_ids, (int)(fr < I_cnt ? fr : I_cnt - 1));
            ZH_DMA16(Xh + (size_t)id * 16 + (c16 ^ rr), dst + 64 * i);
        }
        I_t++; issued++;
    };
    issue_next();
    issue_next();
    for (int c = 0; c < CH; c++) {
        // the next chunk: does it lie entirely in the group this chunk's last row belongs to?  (wave-uniform; as sweep128_kernel)
        const uint64_t r0n = r0 + 64;
        const bool have_next = c + 1 < CH && r0n < R_grouped;
        const uint32_t cntn = have_next ? (uint32_t)(R_grouped - r0n < 64 ? R_grouped - r0n : 64) : 0;
        const uint32_t lw = (uint32_t)__builtin_amdgcn_readlane((int)my_within, (int)cnt - 1) + 1;
        const uint32_t loff = (uint32_t)__builtin_amdgcn_readlane((int)my_off, (int)cnt - 1);
        const uint32_t llen = (uint32_t)__builtin_amdgcn_readlane((int)my_len, (int)cnt - 1);
        const uint32_t lg = (uint32_t)__builtin_amdgcn_readlane((int)my_g, (int)cnt - 1);
        const bool fast = have_next && cnt == 64 && (uint64_t)lw + cntn <= llen;
        if (fast) {  // (cnt == 64: four tiles follow)
            const uint32_t wn = lw + (lane < cntn ? lane : cntn - 1);
            if (leaf_ids) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(leaf_ids + (size_t)loff + wn),
                                                           (__attribute__((address_space(3))) void *)ids_lds[wid], 4, 0, 0);
            else { N_ids = loff + wn; N_cnt = cntn; N_known = true; }
        }
        const uint32_t ntile = (cnt + 15) / 16;
#pragma unroll 1
        for (uint32_t t = 0; t < ntile; t++) {
            // tile `consumed` has landed once at most the younger tile's four requests are outstanding
            if (issued - consumed >= 2u) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // the next chunk's ids were requested before this chunk's first tile was waited for -- older than the tile requested behind that tile:
            // the second tile's wait covers them
            // (LDS reads of what the DMA wrote are inline asm: where the compiler SEES such a read it waits for every outstanding request)
            if (t == 1 && fast && leaf_ids) {
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(N_ids) : "v"(ids_addr) : "memory");
                N_cnt = cntn; N_known = true;
            }
            f16x8 Bf[4];
            {
                const uint32_t ta = tile_addr + (consumed & 1u) * 4096u;
                u32x4v f0, f1, f2, f3;
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3)
                             : "v"(ta + frag_off[0]), "v"(ta + frag_off[1]), "v"(ta + frag_off[2]), "v"(ta + frag_off[3]) : "memory");
                Bf[0] = __builtin_bit_cast(f16x8, f0); Bf[1] = __builtin_bit_cast(f16x8, f1);
                Bf[2] = __builtin_bit_cast(f16x8, f2); Bf[3] = __builtin_bit_cast(f16x8, f3);
            }
            const uint32_t fr = 16 * t + c16;                       // this lane's column: flat row fr of the chunk
            const uint32_t rg = (uint32_t)__shfl((int)my_g, (int)(fr < cnt ? fr : cnt - 1));
            const uint32_t rw = (uint32_t)__shfl((int)my_within, (int)(fr < cnt ? fr : cnt - 1));
            if (rg != c_rg) {
                const uint4 *gq = reinterpret_cast<const uint4 *>(groups + rg);
                const uint4 g0 = gq[0], k01 = gq[2], k23 = gq[3];
                c_rg = rg;
                c_lo[0] = k01.x; c_lo[1] = k01.z; c_lo[2] = k23.x; c_lo[3] = k23.z;
                c_hi = (k01.y & 15u) | ((k01.w & 15u) << 4) | ((k23.y & 15u) << 8) | ((k23.w & 15u) << 12) | (g0.z << 16);
            }
            const uint32_t last = 16 * t + 15 < cnt ? 16 * t + 15 : cnt - 1;
            const uint32_t g_first = (uint32_t)__builtin_amdgcn_readlane((int)my_g, (int)(16 * t));
            const uint32_t g_last = (uint32_t)__builtin_amdgcn_readlane((int)my_g, (int)last);
            float a2 = 0.f;
#pragma unroll
            for (int st = 0; st < 4; st++) a2 = dot8_self(Bf[st], a2);
            a2 = xor16<OpAdd>(a2);
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a2), __float_as_uint(a2), false, false);
                a2 = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
            for (uint32_t gp = g_first; gp <= g_last; gp += 4) {
                if (gp != a_g0) {  // (wave-uniform) the queries of groups gp .. gp + 3: A's row m = slot m & 3 of group gp + (m >> 2)
                    a_g0 = gp;
                    const uint32_t grp = gp + (c16 >> 2);
                    const bool on = grp < n_groups && (c16 & 3u) < groups[grp < n_groups ? grp : 0].gsize;
                    const uint32_t b = on ? groups[grp].b[c16 & 3u] : 0u;
#pragma unroll
                    for (int st = 0; st < 4; st++) {
                        const u32x4v v = Qh[(size_t)b * 16 + 4 * st + h];
                        Aq[st] = on ? __builtin_bit_cast(f16x8, v) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    }
                }
                f32x4v acc[4];
#pragma unroll
                for (int st = 0; st < 4; st++) acc[st] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aq[st], Bf[st], f32x4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const f32x4v dsum = (acc[0] + acc[1]) + (acc[2] + acc[3]);  // [i] = query slot i of group gp + h against stored row c16 of the tile
                if (fr < cnt && rg == gp + h) {
                    const uint32_t a2b = __float_as_uint(a2 * inv * inv);
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if ((uint32_t)i < (c_hi >> 16))
                            __builtin_nontemporal_store(((uint64_t)a2b << 32) | __float_as_uint(dsum[i] * inv),
                                                        iv + ((((uint64_t)((c_hi >> (4 * i)) & 15u)) << 32) | c_lo[i]) + rw);
                }
            }
            consumed++;
            // the buffer this tile was read from takes the tile after next (every fragment read has been consumed by the MFMAs above)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            issue_next();
        }
        if (!have_next) break;
        r0 = r0n; cnt = cntn;
        if (fast) {
            my_g = lg; my_off = loff; my_len = llen;
            my_within = lw + (lane < cntn ? lane : cntn - 1);
            my_id = N_ids;
        } else {
            // (nothing of the next chunk was known: the issuing side has run dry and every tile issued is consumed)
            resolve_flat_rows(r0, cnt, lane, groups, groupRowOff, n_groups, waveGroup, leaf_ids, my_g, my_id, my_within, &my_off, &my_len);
            I_ids = my_id; I_cnt = cnt; I_t = 0; N_known = false;
            issue_next();
            issue_next();
        }
    }
}

// ---- round 6: the same sweep with a LEAN instruction stream for the common chunk.  The two kernels above spend ~270 instructions of a wave on
// a 16-row tile (ids -> ds_bpermute -> 64-bit addresses, sixteen dependent v_dot2 + a butterfly for |x^|^2, per-lane group look-ups through
// __shfl, up to four predicated 64-bit-address stores), which round 5 took for what holds them at 0.54 of 8 TB/s.  It is not: this kernel spends ~55
// and, storing as they do, ran no faster -- what it bought is room for the fused form below.  A leaf is thousands of rows and a chunk
// is 64: all but ~1 chunk in 70 lie inside ONE (leaf, <= 4 queries) group, and for those everything that is per-row bookkeeping above is wave-uniform:
//   * the group's record sits in scalar registers; A's row m is query slot m & 3 of THE group, so every lane (c, h) holds stored row c against all
//     four slots (accumulator j = slot j); the lanes of 16-lane row h keep tile h's, so that after a chunk's four tiles lane l holds flat row l:
//     unfused, ONE 512-byte run of results per slot and chunk (a 128-byte run per slot and tile cost 8 % more: scattered writes among the row
//     reads are dearer than their bytes);
//   * the chunk's ids are loaded in the order lane (h, c) = flat row 4 c + h: instruction i of tile t wants row 4 (4 t + i) + h -- the same
//     16-lane row, lane 4 t + i: a DPP row_newbcast, no LDS crossbar;
//   * |x^|^2 is the DIAGONAL of the tile's Gram matrix: four more MFMAs with the B fragments as both operands (the A and B layouts of
//     v_mfma_f32_16x16x32_f16 coincide) on a matrix pipe that has nothing else to do, three selects and one ds_bpermute;
//   * the four K-steps chain through one accumulator.
// A chunk with ONE group boundary in it (and the batch's last, short one) is scored here too, as two SEGMENTS with the lanes past each segment's count
// switched off (chunk_segments(), last session of round 6); a chunk of three or more groups -- leaves of a handful of rows -- is
// sweep128h_boundary_kernel's: the general per-lane form of the kernel above, one chunk at a time, one launch per batch.  Same raw pairs up to the summation
// order inside the matrix pipe (covered by zh_approx_bound's MFMA term, tests/test_gpu_intervals.py) -- the results behind them are bit-identical
// by construction as before.
// ---- the fused sweep (round 6).  The stores of the raw pairs -- under 4 % of the sweep's bytes -- cost the lean kernel a fifth of its time (0.99 ms per
// 25M-row launch without them = the bare gather's 6.5 TB/s, 1.23-1.30 with; profiles/r06_sweep128h_experiments.txt), and select_tau_kernel /
// select_emit_kernel then read them all back (7 ms per window of a cfg5 shard) to keep ~60 per query.  Fused, a chunk's 64 results never leave the
// wave: per slot (= a visit of the leaf by one query, lsh.rs:290-330)
//   * the interval of every row (approx_interval: the arithmetic select_tau_kernel applies to the raw pair, same operands, same bits);
//   * a visit that takes top_k rows of a longer leaf: the wave keeps the top_k SMALLEST hi of the visit's rows it has seen so far (sorted, lane i
//     the i-th; merged chunk by chunk: topk_merge).  The top_k-th of them is a bound: top_k distinct rows of one leaf have keys at or below it,
//     hence (each is a candidate or beaten by top_k candidates of its leaf) so have top_k candidates of the query.  atomicMin into the query's
//     qtau, which every visit of the query reads.  (A bound from one 64-row chunk -- the first form of this kernel -- is the chunk's 16 % quantile
//     at top_k = 10, and the minimum over chunks tightens slowly: 3900 list entries per query on a cfg5 shard.  The bound only ever certifies what
//     select_tau_kernel's per-visit tau certifies, from a subset of the visit's rows);
//   * rows whose lo exceeds the bound are beaten by top_k candidates and can be in no final top_k: everything else joins the query's list
//     (one atomicAdd per slot and chunk).  Which superset of the survivors the list holds depends on the order the waves ran in; the answer does
//     not: final_survivors / final_exact / final_topk rank the list's rows by their canonical keys, and a row that is not among its leaf's top_k is
//     beaten in the list by top_k rows that are;
//   * a visit that takes the whole leaf (lsh.rs:300-306): rows at or below the query's bound; a visit that takes FEWER than top_k rows of a longer
//     leaf: membership matters -- its intervals go to iv, where exact_visit_kernel expects them (exact_register_kernel lists the visit).
// `take` is the group record's byte (join_group): top_k <= 64 < 255 in this form.

// topv: lane i < kk holds the i-th smallest value seen so far (ascending; 0xFFFFFFFF = none yet), lanes >= kk 0xFFFFFFFF.  Merges the 64 values `v`
// (one per lane) into it; `scratch`: 64 words of wave-private LDS.  Ranks in the union (old before new on ties, lower lane first): only values below
// the current kk-th can enter, so the loops are as long as the chunk has such values (a handful once the wave has seen a few hundred rows).
__device__ __forceinline__ void topk_merge(uint32_t &topv, uint32_t v, uint32_t kk, uint32_t lane, uint32_t *scratch) {
    const uint32_t bound = (uint32_t)__builtin_amdgcn_readlane((int)topv, (int)(kk - 1));
    const bool isnew = v < bound;
    const uint64_t nm = __ballot(isnew);
    if (!nm) return;  // (wave-uniform)
    uint32_t r_old = lane, r_new = 0;
    for (uint64_t m = nm; m; m &= m - 1) {
        const uint32_t l = (uint32_t)__builtin_ctzll(m);
        const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
        r_old += x < topv ? 1u : 0u;
        r_new += (x < v || (x == v && l < lane)) ? 1u : 0u;
    }
    for (uint32_t i = 0; i < kk; i++) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)topv, (int)i);
        r_new += o <= v ? 1u : 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < kk && r_old < kk) scratch[r_old] = topv;
    if (isnew && r_new < kk) scratch[r_new] = v;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    topv = lane < kk ? scratch[lane] : 0xFFFFFFFFu;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int KINDA>
__device__ __forceinline__ void fused_slot(float s, float a2, uint32_t id, uint32_t lane, uint32_t bq, uint32_t take, uint32_t glen, uint32_t k_top,
                                           float Kc, const ZhApprox &ap, uint64_t *__restrict__ iv_slot, uint32_t &topv, uint32_t *scratch, uint32_t tau,
                                           bool valid) {
    // valid: the lane holds a row of the segment (a short segment's other lanes hold copies of its last row: no interval stored, none offered, none listed)
    // tau: the query's bound as the caller read it BEFORE the chunk's tiles (fused_tau) -- any value qtau ever held is a certified bound, and a read
    // here would sit behind the row loads in flight (vmcnt counts in order): every slot a memory round trip with nothing else to do
    if (take == 0) return;  // (wave-uniform)
    const float4 qm = ap.qmeta[bq];
    const uint64_t w = approx_interval<KINDA>(s, a2, qm, Kc, ap.row_rho, ap.rho_norm);
    const uint32_t lo = (uint32_t)w, hi = valid ? (uint32_t)(w >> 32) : 0xFFFFFFFFu;
    if (take < glen && take < k_top) {  // the exact path's visit
        if (valid) __builtin_nontemporal_store(w, iv_slot);
        return;
    }
    tau = (uint32_t)__builtin_amdgcn_readfirstlane((int)tau);
    if (take < glen) {  // top_k rows of a longer leaf
        const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)topv, (int)(take - 1));
        topk_merge(topv, hi, take, lane, scratch);
        const uint32_t mine = (uint32_t)__builtin_amdgcn_readlane((int)topv, (int)(take - 1));
        if (mine < before && mine < tau && lane == 0) atomicMin(&ap.qtau[bq], mine);
        if (mine < tau) tau = mine;
    }
    const uint64_t m = __ballot(valid && lo <= tau);
    if (!m) return;
    const uint32_t M = (uint32_t)__builtin_popcountll(m);
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&ap.qcount[bq], M);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (base + M > ap.capq) {
        if (lane == 0) atomicOr(&ap.ctl[1], 1u);
        return;
    }
    if (valid && lo <= tau) {
        const size_t o = (size_t)bq * ap.capq + base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        ap.list_lo[o] = lo; ap.list_hi[o] = hi; ap.list_id[o] = id;
    }
}

// the bounds of a group's queries, requested before the chunk's tiles are waited for (slots past gsize: none)
__device__ __forceinline__ void fused_tau(const ZhGroup *gr, uint32_t gsize, const ZhApprox &ap, uint32_t (&tau)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++)
        tau[j] = (uint32_t)j < gsize ? __hip_atomic_load(&ap.qtau[gr->b[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
}
// The fused sweep's PRE-TEST (L2 family, a table whose rows' norm estimates are bounded: ap.nx_max > 0 -- the byte copy).  Once a query has a bound,
// nearly every (row, query) pair of a chunk is far above it, and telling so does not take the pair's interval: with Emax = l2_E(nx_max) >= E,
//     V - Emax > tau   =>   lo = V - E >= V - Emax > tau:   the row joins no list, and its hi > tau could only enter the wave's running top-k as a
// value that never brings the k-th below tau -- i.e. never changes what the sweep certifies.  A chunk whose 64 rows all pass for a slot skips
// fused_slot for it (five VALU operations per pair instead of ~45); anything else -- a row near or under the bound, no bound yet, a query or sum
// that approx_interval would call uncertain -- takes fused_slot as before, for the whole chunk.  pre[j] = {1 / sigma, |q|^2, Emax} of slot j, Emax < 0: no pre-test.
__device__ __forceinline__ void fused_pretest_setup(const ZhGroup *gr, uint32_t gsize, const ZhApprox &ap, float Kc, float (&pre)[4][3]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        pre[j][0] = 0.f; pre[j][1] = 0.f; pre[j][2] = -1.f;
        if ((uint32_t)j < gsize && ap.nx_max > 0.f) {
            const float4 qm = ap.qmeta[gr->b[j]];
            const float E = l2_E(ap.nx_max, qm, Kc, ap.row_rho, ap.rho_norm);
            // what approx_interval asks of a pair before it is certain of anything: nn > 1e-12 (here from the query alone), sum < 1e37 (the rows' share is < 1e36)
            const bool ok = qm.z > 1e-12f && qm.y < 1e36f && E < 3.0e38f && ap.nx_max < 1e18f;
            // (wave-uniform, and said so: twelve scalar registers, not twelve vector ones)
            pre[j][0] = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(qm.x)));
            pre[j][1] = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(qm.y)));
            pre[j][2] = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(ok ? E : -1.f)));
        }
    }
}
__device__ __forceinline__ float f32_unsortable(uint32_t s) {  // (0xFFFFFFFF, "no bound yet", is a NaN: every comparison below fails)
    return __uint_as_float(s ^ ((s >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
// true: no row of the chunk can matter to the slot's query (wave-uniform)
__device__ __forceinline__ bool fused_pretest(float s, float a2, const float (&pre)[3], uint32_t tau, bool valid) {
    if (!(pre[2] >= 0.f)) return false;
    const float V = l2_V(s, a2, make_float4(pre[0], pre[1], 0.f, 0.f));
    const bool drop = !valid || ((V - pre[2] > f32_unsortable(tau)) && V < 3.0e38f);
    return __ballot(!drop) == 0;
}
template <int N>
__device__ __forceinline__ uint32_t row_newbcast(uint32_t v) {  // every lane: the value of lane N of its own 16-lane row (DPP, no LDS)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + N, 0xF, 0xF, true);  // (bound_ctrl: no `old` operand to initialise)
}
// the four row-load instructions of tile T of a chunk whose ids the lanes hold as lane (h, c) = flat row 4 c + h: instruction i = rows 4 (4 T + i) + h
template <int T>
__device__ __forceinline__ void issue_lean_tile(const u32x4v *__restrict__ Xh, uint32_t ids, uint32_t c16, u32x4v (&R)[4]) {
#if defined(ZH_S128L_EXP) && ZH_S128L_EXP == 3  // plain (temporal) row loads
    R[0] = Xh[(size_t)row_newbcast<4 * T + 0>(ids) * 16 + c16];
    R[1] = Xh[(size_t)row_newbcast<4 * T + 1>(ids) * 16 + c16];
    R[2] = Xh[(size_t)row_newbcast<4 * T + 2>(ids) * 16 + c16];
    R[3] = Xh[(size_t)row_newbcast<4 * T + 3>(ids) * 16 + c16];
#else
    R[0] = __builtin_nontemporal_load(Xh + (size_t)row_newbcast<4 * T + 0>(ids) * 16 + c16);
    R[1] = __builtin_nontemporal_load(Xh + (size_t)row_newbcast<4 * T + 1>(ids) * 16 + c16);
    R[2] = __builtin_nontemporal_load(Xh + (size_t)row_newbcast<4 * T + 2>(ids) * 16 + c16);
    R[3] = __builtin_nontemporal_load(Xh + (size_t)row_newbcast<4 * T + 3>(ids) * 16 + c16);
#endif
}
// The chunk of flat rows r0 .. r0 + cnt - 1 (r0 a multiple of 64, cnt <= 64: the batch's last chunk is short) as at most TWO SEGMENTS, each inside one
// group: cntA rows of group g from its row within0 and -- where a group ends inside the chunk -- cntB rows from the first row of group g + 1.
// THE predicate that splits a launch's chunks between the lean kernels (true: with leaves of thousands of rows every chunk; a segment is scored
// like a chunk whose lanes past its count are switched off) and sweep128h_boundary_kernel (false: three or more groups in 64 rows): every chunk
// is scored by exactly one of them.  (Until the last session of round 6 a chunk with a group boundary in it was the boundary kernel's: ~1 chunk in 70
// at cfg5, scored one chunk per wave at a time -- a serial 1.2-1.3 ms behind every window's sweep launches, 5 % of the stage.)
__device__ __forceinline__ bool chunk_segments(uint64_t r0, uint32_t cnt, const uint64_t *__restrict__ groupRowOff, uint64_t n_groups,
                                               const uint32_t *__restrict__ waveGroup, uint32_t &g, uint32_t &within0, uint32_t &cntA, uint32_t &cntB) {
    if (!waveGroup || cnt == 0) return false;
    g = waveGroup[r0 >> 6];
    const uint64_t off = groupRowOff[g], end = r0 + cnt;
    const uint64_t nxt = (uint64_t)g + 1 < n_groups ? groupRowOff[g + 1] : ~0ull;
    within0 = (uint32_t)(r0 - off);
    cntA = cnt; cntB = 0;
    if (nxt >= end) return true;
    const uint64_t nxt2 = (uint64_t)g + 2 < n_groups ? groupRowOff[g + 2] : ~0ull;
    if (nxt2 < end) return false;
    cntA = (uint32_t)(nxt - r0); cntB = (uint32_t)(end - nxt);
    return true;
}
#ifndef ZH_S128L_EXP
#define ZH_S128L_EXP 0     // timing experiments on the lean kernel (profiles/r06_sweep128h_experiments.txt); 0 = the shipped form
#endif
#ifndef ZH_S128L_WAVES
#define ZH_S128L_WAVES 5   // waves per SIMD the register allocation is held to (A/B)
#endif
// FUSE >= 0 (= approx_interval's KINDA): the FUSED form -- see "the fused sweep" below; FUSE < 0: raw pairs to iv for select_tau_kernel
template <int CH, int FUSE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FUSE >= 0 ? 4 : ZH_S128L_WAVES, FUSE >= 0 ? 4 : ZH_S128L_WAVES)))
void sweep128h_lean_kernel(const u32x4v *__restrict__ Xh, const u32x4v *__restrict__ Qh, float inv,
                           const ZhGroup *__restrict__ groups, const uint64_t *__restrict__ groupRowOff,
                           uint64_t n_groups, const uint32_t *__restrict__ waveGroup,
                           const uint32_t *__restrict__ leaf_ids, uint64_t row_begin, uint64_t R_grouped,
                           uint64_t *__restrict__ iv, ZhApprox ap, uint32_t k_top, float Kc) {
    __shared__ u32x4v rows_lds[4][16 * 16];  // per wave: ONE tile = 16 rows x 16 pieces of 16 bytes, piece p of row R at R * 16 + (p ^ R)
    const uint32_t lane = threadIdx.x & 63, c16 = lane & 15, h = lane >> 4;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wid;
    const uint64_t r_first = row_begin + wave * (64 * CH);
    if (r_first >= R_grouped) return;
    u32x4v *tl = rows_lds[wid];
    f16x8 Aq[4];
#pragma unroll
    for (int st = 0; st < 4; st++) Aq[st] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t a_g = 0xFFFFFFFFu;   // the group whose queries A holds: row m = slot m & 3, zero past gsize
    const float inv2 = inv * inv;
    const uint32_t diag_src = (((c16 >> 2) << 4) | c16) << 2;  // ds_bpermute address of lane (c16, c16 >> 2): where G[c16][c16] lives
    const bool c_odd = (c16 & 1u) != 0, c_up = (c16 & 2u) != 0;  // which of a lane's four Gram entries sits on the diagonal: register c16 & 3
    u32x4v R[4];
    // fused form: per slot, the top_k smallest hi of the visit's rows this wave has seen (lane i the i-th; reset when the group changes), and where
    // flat row `lane` of a chunk finds its id (lane (h, c16) holds row 4 c16 + h)
    uint32_t topv[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    float pre[4][3] = {{0.f, 0.f, -1.f}, {0.f, 0.f, -1.f}, {0.f, 0.f, -1.f}, {0.f, 0.f, -1.f}};  // (fused_pretest_setup, per group)
    const uint32_t nat_src = (((lane & 3u) << 4) | (lane >> 2)) << 2;
    bool carried = false;          // the previous chunk left THIS chunk's first tile travelling in R and its ids in `ids`
    uint32_t ids = 0;              // lane (h, c16) holds the id of flat row 4 c16 + h of the chunk
    uint32_t g = 0, within0 = 0;   // the chunk's group and its first row's position in the group (wave-uniform)
    for (int c = 0; c < CH; c++) {
        const uint64_t r0 = r_first + 64ull * c;
        if (r0 >= R_grouped) break;
        uint32_t cntA = 64, cntB = 0;  // (a carried chunk is a full one inside its group)
        if (!carried) {
            uint32_t g_v, w_v, a_v, b_v;
            if (!chunk_segments(r0, (uint32_t)(R_grouped - r0 < 64 ? R_grouped - r0 : 64), groupRowOff, n_groups, waveGroup, g_v, w_v, a_v, b_v))
                continue;  // (wave-uniform: sweep128h_boundary_kernel's chunk)
            g = (uint32_t)__builtin_amdgcn_readfirstlane((int)g_v);
            within0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)w_v);
            cntA = (uint32_t)__builtin_amdgcn_readfirstlane((int)a_v);
            cntB = (uint32_t)__builtin_amdgcn_readfirstlane((int)b_v);
        }
      for (uint32_t seg = 0; seg < (cntB ? 2u : 1u); seg++) {  // the chunk's segments: one, or two where a group ends inside it
        const uint32_t cnt = seg ? cntB : cntA;
        if (seg) { g += 1; within0 = 0; }
        const ZhGroup *gr = groups + g;
        const uint32_t leaf_off = gr->leaf_off, glen = gr->len, gsize = gr->gsize;
        if (!carried) {
            const uint32_t fr = 4u * c16 + h, w = within0 + (fr < cnt ? fr : cnt - 1);  // (a short segment: the lanes past it hold its last row again)
            ids = leaf_ids ? leaf_ids[(size_t)leaf_off + w] : leaf_off + w;
            issue_lean_tile<0>(Xh, ids, c16, R);
        }
        if (g != a_g) {  // (wave-uniform) A's row m = query slot m & 3 of group g, zero past gsize
            a_g = g;
            topv[0] = topv[1] = topv[2] = topv[3] = 0xFFFFFFFFu;
            if constexpr (FUSE == 0) fused_pretest_setup(gr, gsize, ap, Kc, pre);
            const bool on = (c16 & 3u) < gsize;
            const uint32_t b = on ? gr->b[c16 & 3u] : 0u;
#pragma unroll
            for (int st = 0; st < 4; st++) {
                const u32x4v v = Qh[(size_t)b * 16 + 4 * st + h];
                Aq[st] = on ? __builtin_bit_cast(f16x8, v) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        // the next chunk: entirely inside the same group?  Then its ids are requested now and its first tile behind this chunk's last one
        const bool next_fast = cnt == 64 && cntB == 0 && c + 1 < CH && (uint64_t)within0 + 128 <= glen;
        uint32_t nxt_ids = 0;
        if (next_fast) {
            const uint32_t w = within0 + 64u + 4u * c16 + h;
            nxt_ids = leaf_ids ? leaf_ids[(size_t)leaf_off + w] : leaf_off + w;
        }
        // Every lane (c16, h) of a tile holds stored row c16 against ALL four slots (accumulator j = slot j, whatever h): the lanes of 16-lane row
        // h keep tile h's, so that after the chunk's four tiles lane l holds flat row l of the chunk -- ONE 512-byte run per slot and chunk
        // instead of a 128-byte run per slot and tile (small scattered writes among the row reads cost HBM more than their bytes:
        // profiles/r06_sweep128h_experiments.txt)
        uint32_t tau_pre[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        if constexpr (FUSE >= 0) fused_tau(gr, gsize, ap, tau_pre);
        float res[4] = {0.f, 0.f, 0.f, 0.f}, res_a2 = 0.f;
        auto tile = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t rr = 4u * i + h;
                tl[rr * 16 + (c16 ^ rr)] = R[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if constexpr (t < 3) issue_lean_tile<(t + 1) & 3>(Xh, ids, c16, R);
            else if (next_fast) issue_lean_tile<0>(Xh, nxt_ids, c16, R);
            f16x8 Bf[4];
#pragma unroll
            for (int st = 0; st < 4; st++) Bf[st] = __builtin_bit_cast(f16x8, tl[c16 * 16 + ((4u * st + h) ^ c16)]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            f32x4v dsum = {0.f, 0.f, 0.f, 0.f}, gram = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 4; st++) {
                dsum = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aq[st], Bf[st], dsum, 0, 0, 0);
                gram = __builtin_amdgcn_mfma_f32_16x16x32_f16(Bf[st], Bf[st], gram, 0, 0, 0);
            }
            const float dg_lo = c_odd ? gram[1] : gram[0], dg_hi = c_odd ? gram[3] : gram[2], dg = c_up ? dg_hi : dg_lo;
            const float a2 = __int_as_float(__builtin_amdgcn_ds_bpermute((int)diag_src, __float_as_int(dg)));
            const bool keep = h == (uint32_t)t;
#pragma unroll
            for (int j = 0; j < 4; j++) res[j] = keep ? dsum[j] : res[j];
            res_a2 = keep ? a2 : res_a2;
        };
        tile(std::integral_constant<int, 0>{});
        tile(std::integral_constant<int, 1>{});
        tile(std::integral_constant<int, 2>{});
        tile(std::integral_constant<int, 3>{});
        if constexpr (FUSE < 0) {
            const uint32_t a2b = __float_as_uint(res_a2 * inv2);
            uint64_t *const dst = iv + within0 + lane;  // slot j's results of the chunk: its key slice, positions within0 .. within0 + 63
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((uint32_t)j < gsize && lane < cnt) {
                    const uint64_t v = ((uint64_t)a2b << 32) | __float_as_uint(res[j] * inv);
#if ZH_S128L_EXP == 1    // timing experiment (results invalid): no result stores
                    if (res_a2 == 123456.789f) dst[gr->key_off[j]] = v;
#elif ZH_S128L_EXP == 2  // plain (write-back) stores
                    dst[gr->key_off[j]] = v;
#elif ZH_S128L_EXP == 7  // timing experiment (results invalid): every run starts on a 128-byte line
                    __builtin_nontemporal_store(v, iv + ((gr->key_off[j] + within0) & ~15ull) + lane);
#elif ZH_S128L_EXP == 8  // timing experiment (results invalid): the runs land in 512 KiB that stay in the L2s
                    __builtin_nontemporal_store(v, iv + (wave & 1023u) * 64 + lane);
#else
                    __builtin_nontemporal_store(v, dst + gr->key_off[j]);
#endif
                }
        } else {
            // ---- the fused sweep: lane l holds flat row l of the chunk against every slot.  Per slot: the interval (select_tau_kernel's arithmetic), the
            // visit's bound, the rows it cannot rule out straight into the query's list -- nothing else leaves the wave
            const uint32_t idn = (uint32_t)__builtin_amdgcn_ds_bpermute((int)nat_src, (int)ids);
            const float a2s = res_a2 * inv2;
            const uint32_t take4 = gr->take4;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((uint32_t)j < gsize) {
                    const uint32_t take = (take4 >> (8 * j)) & 255u;
                    if constexpr (FUSE == 0) {
                        if (!(take < glen && take < k_top)) {  // (the exact path's visits keep every interval)
                            const uint32_t t_w = take < glen ? (uint32_t)__builtin_amdgcn_readlane((int)topv[j], (int)(take - 1)) : 0xFFFFFFFFu;
                            const uint32_t t_q = (uint32_t)__builtin_amdgcn_readfirstlane((int)tau_pre[j]);
                            if (fused_pretest(res[j] * inv, a2s, pre[j], t_w < t_q ? t_w : t_q, lane < cnt)) continue;
                        }
                    }
                    fused_slot<FUSE>(res[j] * inv, a2s, idn, lane, gr->b[j], take, glen, k_top, Kc, ap,
                                     iv + gr->key_off[j] + within0 + lane, topv[j], reinterpret_cast<uint32_t *>(tl), tau_pre[j], lane < cnt);  // (the tile buffer is idle here: the next tile waits in R)
                }
        }
        carried = next_fast;
        if (next_fast) { ids = nxt_ids; within0 += 64; }
      }
    }
}

// ---- the lean kernel over BYTE rows (row_byte128_kernel's copy): the same chunk, the same epilogues, half the bytes.  A stored row is eight 16-byte
// pieces, so ONE wave instruction loads eight rows and the unit staged through LDS is a DOUBLE tile -- 32 rows, the same four instructions and 4 KB
// in flight per wave as a 16-row tile of halves -- scored by two sets of MFMAs:
//   * instruction i of double tile T: the 16-lane row h loads the rows whose ids its lanes 8 T + 2 i and 8 T + 2 i + 1 hold (flat rows
//     32 T + 8 i + 4 sub + h: two row_newbcasts and a select), lanes 0-7 the pieces of one, lanes 8-15 of the other; they land in the LDS image as
//     image rows J = 8 i + 2 h + sub (neighbours: the sixteen lanes fill 256 consecutive bytes), piece p of image row J at J * 8 + (p ^ ((J >> 1) & 7));
//   * set m scores image rows 16 m .. 16 m + 15: lane (c16, h) reads pieces 2 h and 2 h + 1 of image row 16 m + c16 -- elements 32 h .. 32 h + 31 --
//     converts them (bytes8_to_f16: exact) and step st multiplies elements 32 h + 8 st ..: the K index of the matrix instruction is only a
//     summation order, so A holds the queries' pieces 4 h + st where the kernel of halves holds 4 st + h;
//   * the lanes of 16-lane row h keep set h's results (h = 2 T + m), i.e. lane l ends up with flat row pi(l) = 16 h + 8 (c16 >> 3) + 4 (c16 & 1)
//     + ((c16 >> 1) & 3) of the chunk: a fixed permutation inside each 16 -- ids, key positions and raw runs are addressed through it.
// inv = 1: the rows are their own values (no table scale); |x|^2 is exact.
template <int T>
__device__ __forceinline__ void issue_byte_tile(const u32x4v *__restrict__ Xb, uint32_t ids, uint32_t p8, bool sub, u32x4v (&R)[4]) {
    // (every broadcast with ALL lanes active -- a DPP read of a lane that is switched off returns zero -- and only then the choice)
    const uint32_t e0 = row_newbcast<8 * T + 0>(ids), o0 = row_newbcast<8 * T + 1>(ids), e1 = row_newbcast<8 * T + 2>(ids), o1 = row_newbcast<8 * T + 3>(ids);
    const uint32_t e2 = row_newbcast<8 * T + 4>(ids), o2 = row_newbcast<8 * T + 5>(ids), e3 = row_newbcast<8 * T + 6>(ids), o3 = row_newbcast<8 * T + 7>(ids);
    const uint32_t i0 = sub ? o0 : e0, i1 = sub ? o1 : e1, i2 = sub ? o2 : e2, i3 = sub ? o3 : e3;
    R[0] = __builtin_nontemporal_load(Xb + (size_t)i0 * 8 + p8);
    R[1] = __builtin_nontemporal_load(Xb + (size_t)i1 * 8 + p8);
    R[2] = __builtin_nontemporal_load(Xb + (size_t)i2 * 8 + p8);
    R[3] = __builtin_nontemporal_load(Xb + (size_t)i3 * 8 + p8);
}
#ifndef ZH_S128B_WAVES
#define ZH_S128B_WAVES 4   // waves per SIMD the byte kernel's register allocation is held to (A/B)
#endif
#ifndef ZH_S128B_EXP
#define ZH_S128B_EXP 0     // timing experiments on the byte kernel (profiles/r06_sweep128b_experiments.txt), results INVALID: 1 no conversion, 2 no epilogue, 3 no Gram; valid: 4 no pre-test
#endif
template <int CH, int FUSE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ZH_S128B_WAVES, ZH_S128B_WAVES)))
void sweep128b_lean_kernel(const u32x4v *__restrict__ Xb, const u32x4v *__restrict__ Qh,
                           const ZhGroup *__restrict__ groups, const uint64_t *__restrict__ groupRowOff,
                           uint64_t n_groups, const uint32_t *__restrict__ waveGroup,
                           const uint32_t *__restrict__ leaf_ids, uint64_t row_begin, uint64_t R_grouped,
                           uint64_t *__restrict__ iv, ZhApprox ap, uint32_t k_top, float Kc) {
    __shared__ u32x4v rows_lds[4][32 * 8];  // per wave: ONE double tile = 32 image rows x 8 pieces of 16 bytes
    const uint32_t lane = threadIdx.x & 63, c16 = lane & 15, h = lane >> 4, p8 = c16 & 7u;
    const bool sub = (c16 & 8u) != 0;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wid;
    const uint64_t r_first = row_begin + wave * (64 * CH);
    if (r_first >= R_grouped) return;
    u32x4v *tl = rows_lds[wid];
    f16x8 Aq[4];
#pragma unroll
    for (int st = 0; st < 4; st++) Aq[st] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t a_g = 0xFFFFFFFFu;
    const uint32_t diag_src = (((c16 >> 2) << 4) | c16) << 2;
    const bool c_odd = (c16 & 1u) != 0, c_up = (c16 & 2u) != 0;
    const uint32_t pi = 16u * h + 8u * (c16 >> 3) + 4u * (c16 & 1u) + ((c16 >> 1) & 3u);  // the flat row of the chunk this lane ends up with
    const uint32_t nat_src = (((pi & 3u) << 4) | (pi >> 2)) << 2;                          // ... and the lane that holds its id
    // where this lane's loads land in the image, and where its fragments come from (16-byte units; + 128 * m for set m)
    uint32_t wr_at[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t J = 8u * i + 2u * h + (sub ? 1u : 0u);
        wr_at[i] = J * 8 + (p8 ^ ((J >> 1) & 7u));
    }
    const uint32_t rd_sw = (c16 >> 1) & 7u, rd0 = c16 * 8 + ((2u * h) ^ rd_sw), rd1 = c16 * 8 + ((2u * h + 1u) ^ rd_sw);
    u32x4v R[4];
    uint32_t topv[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    float pre[4][3] = {{0.f, 0.f, -1.f}, {0.f, 0.f, -1.f}, {0.f, 0.f, -1.f}, {0.f, 0.f, -1.f}};  // (fused_pretest_setup, per group)
    bool carried = false;
    uint32_t ids = 0;              // lane (h, c16) holds the id of flat row 4 c16 + h of the chunk
    uint32_t g = 0, within0 = 0;
    for (int c = 0; c < CH; c++) {
        const uint64_t r0 = r_first + 64ull * c;
        if (r0 >= R_grouped) break;
        uint32_t cntA = 64, cntB = 0;  // (a carried chunk is a full one inside its group)
        if (!carried) {
            uint32_t g_v, w_v, a_v, b_v;
            if (!chunk_segments(r0, (uint32_t)(R_grouped - r0 < 64 ? R_grouped - r0 : 64), groupRowOff, n_groups, waveGroup, g_v, w_v, a_v, b_v))
                continue;  // (wave-uniform: sweep128h_boundary_kernel's chunk)
            g = (uint32_t)__builtin_amdgcn_readfirstlane((int)g_v);
            within0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)w_v);
            cntA = (uint32_t)__builtin_amdgcn_readfirstlane((int)a_v);
            cntB = (uint32_t)__builtin_amdgcn_readfirstlane((int)b_v);
        }
      for (uint32_t seg = 0; seg < (cntB ? 2u : 1u); seg++) {  // the chunk's segments: one, or two where a group ends inside it
        const uint32_t cnt = seg ? cntB : cntA;
        if (seg) { g += 1; within0 = 0; }
        const ZhGroup *gr = groups + g;
        const uint32_t leaf_off = gr->leaf_off, glen = gr->len, gsize = gr->gsize;
        if (!carried) {
            const uint32_t fr = 4u * c16 + h, w = within0 + (fr < cnt ? fr : cnt - 1);  // (a short segment: the lanes past it hold its last row again)
            ids = leaf_ids ? leaf_ids[(size_t)leaf_off + w] : leaf_off + w;
            issue_byte_tile<0>(Xb, ids, p8, sub, R);
        }
        if (g != a_g) {  // (wave-uniform) A's row m = query slot m & 3 of group g, zero past gsize; step st = the query's piece 4 h + st
            a_g = g;
            topv[0] = topv[1] = topv[2] = topv[3] = 0xFFFFFFFFu;
            if constexpr (FUSE == 0) fused_pretest_setup(gr, gsize, ap, Kc, pre);
            const bool on = (c16 & 3u) < gsize;
            const uint32_t b = on ? gr->b[c16 & 3u] : 0u;
#pragma unroll
            for (int st = 0; st < 4; st++) {
                const u32x4v v = Qh[(size_t)b * 16 + 4 * h + st];
                Aq[st] = on ? __builtin_bit_cast(f16x8, v) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        const bool next_fast = cnt == 64 && cntB == 0 && c + 1 < CH && (uint64_t)within0 + 128 <= glen;
        uint32_t nxt_ids = 0;
        if (next_fast) {
            const uint32_t w = within0 + 64u + 4u * c16 + h;
            nxt_ids = leaf_ids ? leaf_ids[(size_t)leaf_off + w] : leaf_off + w;
        }
        uint32_t tau_pre[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        if constexpr (FUSE >= 0) fused_tau(gr, gsize, ap, tau_pre);
        float res[4] = {0.f, 0.f, 0.f, 0.f}, res_a2 = 0.f;
        auto dtile = [&](auto tc) {
            constexpr int T = decltype(tc)::value;
#pragma unroll
            for (int i = 0; i < 4; i++) tl[wr_at[i]] = R[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if constexpr (T == 0) issue_byte_tile<1>(Xb, ids, p8, sub, R);
            else if (next_fast) issue_byte_tile<0>(Xb, nxt_ids, p8, sub, R);
            const u32x4v v00 = tl[rd0], v01 = tl[rd1], v10 = tl[128 + rd0], v11 = tl[128 + rd1];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const u32x4v v0 = m ? v10 : v00, v1 = m ? v11 : v01;
#if ZH_S128B_EXP == 1
                const f16x8 Bf[4] = {__builtin_bit_cast(f16x8, v0), __builtin_bit_cast(f16x8, v1), __builtin_bit_cast(f16x8, v0), __builtin_bit_cast(f16x8, v1)};
#else
                const f16x8 Bf[4] = {bytes8_to_f16(v0[0], v0[1]), bytes8_to_f16(v0[2], v0[3]), bytes8_to_f16(v1[0], v1[1]), bytes8_to_f16(v1[2], v1[3])};
#endif
                f32x4v dsum = {0.f, 0.f, 0.f, 0.f}, gram = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 4; st++) {
                    dsum = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aq[st], Bf[st], dsum, 0, 0, 0);
#if ZH_S128B_EXP != 3
                    gram = __builtin_amdgcn_mfma_f32_16x16x32_f16(Bf[st], Bf[st], gram, 0, 0, 0);
#else
                    gram = dsum;
#endif
                }
                const float dg_lo = c_odd ? gram[1] : gram[0], dg_hi = c_odd ? gram[3] : gram[2], dg = c_up ? dg_hi : dg_lo;
                const float a2 = __int_as_float(__builtin_amdgcn_ds_bpermute((int)diag_src, __float_as_int(dg)));
                const bool keep = h == (uint32_t)(2 * T + m);
#pragma unroll
                for (int j = 0; j < 4; j++) res[j] = keep ? dsum[j] : res[j];
                res_a2 = keep ? a2 : res_a2;
            }
        };
        dtile(std::integral_constant<int, 0>{});
        dtile(std::integral_constant<int, 1>{});
        if constexpr (FUSE < 0) {
            const uint32_t a2b = __float_as_uint(res_a2);
            uint64_t *const dst = iv + within0 + pi;  // slot j's results of the chunk: its key slice, positions within0 .. within0 + 63 (a permutation of them)
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((uint32_t)j < gsize && pi < cnt) __builtin_nontemporal_store(((uint64_t)a2b << 32) | __float_as_uint(res[j]), dst + gr->key_off[j]);
        } else {
            const uint32_t idn = (uint32_t)__builtin_amdgcn_ds_bpermute((int)nat_src, (int)ids);
            const uint32_t take4 = gr->take4;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((uint32_t)j < gsize && (ZH_S128B_EXP != 2 || res[j] == 123456.789f)) {
                    const uint32_t take = (take4 >> (8 * j)) & 255u;
                    if constexpr (FUSE == 0 && ZH_S128B_EXP != 4) {
                        if (!(take < glen && take < k_top)) {  // (the exact path's visits keep every interval)
                            const uint32_t t_w = take < glen ? (uint32_t)__builtin_amdgcn_readlane((int)topv[j], (int)(take - 1)) : 0xFFFFFFFFu;
                            const uint32_t t_q = (uint32_t)__builtin_amdgcn_readfirstlane((int)tau_pre[j]);
                            if (fused_pretest(res[j], res_a2, pre[j], t_w < t_q ? t_w : t_q, pi < cnt)) continue;
                        }
                    }
                    fused_slot<FUSE>(res[j], res_a2, idn, lane, gr->b[j], take, glen, k_top, Kc, ap,
                                     iv + gr->key_off[j] + within0 + pi, topv[j], reinterpret_cast<uint32_t *>(tl), tau_pre[j], pi < cnt);  // (the image is idle here: the next double tile waits in R)
                }
        }
        carried = next_fast;
        if (next_fast) { ids = nxt_ids; within0 += 64; }
      }
    }
}

// The chunks chunk_segments() turns down -- three or more groups inside the 64 rows: none with leaves of thousands of rows, most with leaves of a handful --
// in the general per-lane form of sweep128h_kernel, one chunk at a time.  A wave looks at 64 chunks (a lane each: two loads) and
// works through the ones that are its business; |x^|^2 from the Gram diagonal here too.  BYTES: Xh is row_byte128_kernel's copy (a 16-row tile
// is two load instructions: image row J = 8 i + 2 h + sub = the tile's row J, the image and the K order as in sweep128b_lean_kernel; inv = 1).
template <int FUSE, bool BYTES>
__global__ __launch_bounds__(256) void sweep128h_boundary_kernel(const u32x4v *__restrict__ Xh, const u32x4v *__restrict__ Qh, float inv,
                                                                  const ZhGroup *__restrict__ groups, const uint64_t *__restrict__ groupRowOff,
                                                                  uint64_t n_groups, const uint32_t *__restrict__ waveGroup,
                                                                  const uint32_t *__restrict__ leaf_ids, uint64_t row_begin, uint64_t R_grouped,
                                                                  uint64_t *__restrict__ iv, ZhApprox ap, uint32_t k_top, float Kc) {
    __shared__ u32x4v rows_lds[4][16 * 16];
    const uint32_t lane = threadIdx.x & 63, c16 = lane & 15, h = lane >> 4;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wid;
    const uint64_t n_chunks = (R_grouped - row_begin + 63) >> 6;
    if (wave * 64 >= n_chunks) return;
    uint64_t todo;
    {
        const uint64_t ci = wave * 64 + lane;
        bool mine = false;
        if (ci < n_chunks) {
            const uint64_t r0 = row_begin + (ci << 6);
            uint32_t g_v, w_v, a_v, b_v;
            mine = !chunk_segments(r0, (uint32_t)(R_grouped - r0 < 64 ? R_grouped - r0 : 64), groupRowOff, n_groups, waveGroup, g_v, w_v, a_v, b_v);
        }
        todo = __ballot(mine);
    }
    if (!todo) return;
    u32x4v *tl = rows_lds[wid];
    const float inv2 = inv * inv;
    const uint32_t diag_src = (((c16 >> 2) << 4) | c16) << 2;
    const bool c_odd = (c16 & 1u) != 0, c_up = (c16 & 2u) != 0;
    f16x8 Aq[4];
#pragma unroll
    for (int st = 0; st < 4; st++) Aq[st] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t a_g0 = 0xFFFFFFFFu;  // A holds groups a_g0 .. a_g0 + 3: row m = slot m & 3 of group a_g0 + (m >> 2)
    u32x4v R[4];
    while (todo) {
        const uint32_t bit = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1;
        const uint64_t r0 = row_begin + ((wave * 64 + bit) << 6);
        const uint32_t cnt = (uint32_t)(R_grouped - r0 < 64 ? R_grouped - r0 : 64);
        uint32_t my_g, my_id, my_within;
        resolve_flat_rows(r0, cnt, lane, groups, groupRowOff, n_groups, waveGroup, leaf_ids, my_g, my_id, my_within);
        uint32_t c_rg = 0xFFFFFFFFu, c_hi = 0, c_lo[4] = {0, 0, 0, 0};
        uint32_t c_b[4] = {0, 0, 0, 0}, c_take4 = 0, c_len = 0;  // fused form: the rest of the lane's group record
        auto issue_tile = [&](uint32_t t) {
            if constexpr (BYTES) {
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const uint32_t fr = 16u * t + 8u * i + 2u * h + (c16 >> 3);
                    const uint32_t id = (uint32_t)__shfl((int)my_id, (int)(fr < cnt ? fr : cnt - 1));
                    R[i] = __builtin_nontemporal_load(Xh + (size_t)id * 8 + (c16 & 7u));
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t fr = 16u * t + 4u * i + h;
                    const uint32_t id = (uint32_t)__shfl((int)my_id, (int)(fr < cnt ? fr : cnt - 1));
                    R[i] = __builtin_nontemporal_load(Xh + (size_t)id * 16 + c16);
                }
            }
        };
        issue_tile(0);
        const uint32_t ntile = (cnt + 15) / 16;
#pragma unroll 1
        for (uint32_t t = 0; t < ntile; t++) {
            if constexpr (BYTES) {
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const uint32_t J = 8u * i + 2u * h + (c16 >> 3);
                    tl[J * 8 + ((c16 & 7u) ^ ((J >> 1) & 7u))] = R[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t rr = 4u * i + h;
                    tl[rr * 16 + (c16 ^ rr)] = R[i];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t fr = 16 * t + c16;
            const uint32_t rg = (uint32_t)__shfl((int)my_g, (int)(fr < cnt ? fr : cnt - 1));
            const uint32_t rw = (uint32_t)__shfl((int)my_within, (int)(fr < cnt ? fr : cnt - 1));
            const uint32_t rid = (uint32_t)__shfl((int)my_id, (int)(fr < cnt ? fr : cnt - 1));
            if (rg != c_rg) {
                const uint4 *gq = reinterpret_cast<const uint4 *>(groups + rg);
                const uint4 g0 = gq[0], k01 = gq[2], k23 = gq[3];
                c_rg = rg;
                c_lo[0] = k01.x; c_lo[1] = k01.z; c_lo[2] = k23.x; c_lo[3] = k23.z;  // (key slices are < 2^36: the four high nibbles and gsize share a word)
                c_hi = (k01.y & 15u) | ((k01.w & 15u) << 4) | ((k23.y & 15u) << 8) | ((k23.w & 15u) << 12) | (g0.z << 16);
                if constexpr (FUSE >= 0) {
                    const uint4 gb = gq[1];
                    c_b[0] = gb.x; c_b[1] = gb.y; c_b[2] = gb.z; c_b[3] = gb.w;
                    c_take4 = g0.w; c_len = g0.y;
                }
            }
            if (t + 1 < ntile) issue_tile(t + 1);
            f16x8 Bf[4];
            if constexpr (BYTES) {
                const uint32_t sw = (c16 >> 1) & 7u;
                const u32x4v v0 = tl[c16 * 8 + ((2u * h) ^ sw)], v1 = tl[c16 * 8 + ((2u * h + 1u) ^ sw)];
                Bf[0] = bytes8_to_f16(v0[0], v0[1]); Bf[1] = bytes8_to_f16(v0[2], v0[3]);
                Bf[2] = bytes8_to_f16(v1[0], v1[1]); Bf[3] = bytes8_to_f16(v1[2], v1[3]);
            } else {
#pragma unroll
                for (int st = 0; st < 4; st++) Bf[st] = __builtin_bit_cast(f16x8, tl[c16 * 16 + ((4u * st + h) ^ c16)]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t last = 16 * t + 15 < cnt ? 16 * t + 15 : cnt - 1;
            const uint32_t g_first = (uint32_t)__builtin_amdgcn_readlane((int)my_g, (int)(16 * t));
            const uint32_t g_last = (uint32_t)__builtin_amdgcn_readlane((int)my_g, (int)last);
            f32x4v gram = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 4; st++) gram = __builtin_amdgcn_mfma_f32_16x16x32_f16(Bf[st], Bf[st], gram, 0, 0, 0);
            const float dg_lo = c_odd ? gram[1] : gram[0], dg_hi = c_odd ? gram[3] : gram[2], dg = c_up ? dg_hi : dg_lo;
            const float a2 = __int_as_float(__builtin_amdgcn_ds_bpermute((int)diag_src, __float_as_int(dg)));
            for (uint32_t gp = g_first; gp <= g_last; gp += 4) {
                if (gp != a_g0) {
                    a_g0 = gp;
                    const uint32_t grp = gp + (c16 >> 2);
                    const bool on = grp < n_groups && (c16 & 3u) < groups[grp < n_groups ? grp : 0].gsize;
                    const uint32_t b = on ? groups[grp].b[c16 & 3u] : 0u;
#pragma unroll
                    for (int st = 0; st < 4; st++) {
                        const u32x4v v = Qh[(size_t)b * 16 + (BYTES ? 4 * h + st : 4 * st + h)];
                        Aq[st] = on ? __builtin_bit_cast(f16x8, v) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    }
                }
                f32x4v dsum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 4; st++) dsum = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aq[st], Bf[st], dsum, 0, 0, 0);
                if (fr < cnt && rg == gp + h) {
                    const uint32_t a2b = __float_as_uint(a2 * inv2);
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if ((uint32_t)i < (c_hi >> 16)) {
                            uint64_t *const slot = iv + ((((uint64_t)((c_hi >> (4 * i)) & 15u)) << 32) | c_lo[i]) + rw;
                            if constexpr (FUSE < 0) {
                                __builtin_nontemporal_store(((uint64_t)a2b << 32) | __float_as_uint(dsum[i] * inv), slot);
                            } else {
                                // fused form, a lane at a time (these chunks are ~1 in 70): the interval; the exact path's visits keep theirs in iv;
                                // everything else joins its query's list unless the query's bound rules it out (no bound is derived here)
                                const uint32_t take = (c_take4 >> (8 * i)) & 255u, bq = c_b[i];
                                if (take) {
                                    const uint64_t w = approx_interval<FUSE>(dsum[i] * inv, a2 * inv2, ap.qmeta[bq], Kc, ap.row_rho, ap.rho_norm);
                                    if (take < c_len && take < k_top) __builtin_nontemporal_store(w, slot);
                                    else if ((uint32_t)w <= __hip_atomic_load(&ap.qtau[bq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                                        const uint32_t pos = atomicAdd(&ap.qcount[bq], 1u);
                                        if (pos >= ap.capq) atomicOr(&ap.ctl[1], 1u);
                                        else {
                                            const size_t o = (size_t)bq * ap.capq + pos;
                                            ap.list_lo[o] = (uint32_t)w; ap.list_hi[o] = (uint32_t)(w >> 32); ap.list_id[o] = rid;
                                        }
                                    }
                                }
                            }
                        }
                }
            }
        }
    }
}

// the visits of the exact path -- fewer than top_k rows of a longer leaf -- as select_tau_kernel lists them (the fused sweep has no pass over the visits)
__global__ __launch_bounds__(256) void exact_register_kernel(const ZhVisit *__restrict__ visits, uint64_t n_visits, uint32_t k_top, ZhApprox ap) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_visits) return;
    const ZhVisit v = visits[i];
    if (!(v.take && v.take < v.len && v.take < k_top)) return;
    const uint32_t slot = atomicAdd(&ap.ctl[0], 1u);
    const uint32_t eb = atomicAdd(&ap.ctl[2], v.len);
    if (slot >= ap.ex_cap) atomicOr(&ap.ctl[1], 4u);
    else if ((uint64_t)eb + v.len > ap.ex_rows_cap) { atomicOr(&ap.ctl[1], 8u); ap.ex_visits[slot] = make_uint2(0xFFFFFFFFu, 0u); }
    else ap.ex_visits[slot] = make_uint2((uint32_t)i, eb);
    ap.tauv[i] = 0u;
}
hipError_t zh_launch_exact_register(const ZhVisit *dVisits, uint64_t n_visits, uint32_t k, ZhApprox ap, hipStream_t s) {
    if (!n_visits) return hipSuccess;
    if (n_visits > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(exact_register_kernel, dim3((uint32_t)((n_visits + 255) / 256)), dim3(256), 0, s, dVisits, n_visits, k, ap);
    return hipGetLastError();
}

#ifndef ZH_S128L_CH
#define ZH_S128L_CH 4   // 64-row chunks per wave of the lean kernel (A/B)
#endif
#ifndef ZH_S128F_CH
#define ZH_S128F_CH 16  // ... of its fused form: a wave's bound comes from the rows IT has seen (64 chunks = the top_k / 1024 quantile at best)
#endif
#ifndef ZH_S128H_LAUNCH_X
#define ZH_S128H_LAUNCH_X 4   // rows per launch of the lean kernels over the copy of halves, in units of zh_sweep_rows_per_launch(128): 100.8M rows, ~4.1 ms
                              // (cfg5 shard on halves, two passes: 2: 377 k QPS, 4: 393-400 k; with 32 / 64 chunks per wave of the fused form: 382-390 / 376-385 k)
#endif
#ifndef ZH_S128B_LAUNCH_X
#define ZH_S128B_LAUNCH_X 8   // rows per launch of the byte kernel, in units of zh_sweep_rows_per_launch(128): 201.6M rows, ~4.4 ms (A/B on a cfg5 shard,
#endif                        // profiles/r06_sweep128b_experiments.txt: 2 / 4 / 8 with 64 chunks per wave: 643-650 / 713-715 / 731 k QPS -- the tail of a launch
                              // and the first waves of the next, which start without bounds, are paid half as often)
#ifndef ZH_S128B_CH
#define ZH_S128B_CH 64        // 64-row chunks per wave of the fused byte kernel (8: -11 %, 16: -3.5 %, 32: -1 % against 64)
#endif
uint64_t zh_sweep128h_rows_per_launch(bool lean, bool byte_rows) {
    return (lean ? (byte_rows ? ZH_S128B_LAUNCH_X : ZH_S128H_LAUNCH_X) : 1) * zh_sweep_rows_per_launch(128);
}
template <int FUSE, bool BYTES>
static void launch_sweep128h_lean(const void *dXh, const void *dQh, float inv, const ZhGroup *dGroups, const uint64_t *dGroupRowOff, uint64_t n_groups,
                                  const uint32_t *dWaveGroup, const uint32_t *dLeafIds, uint64_t R_grouped, uint64_t *dIv, const ZhApprox &ap,
                                  uint32_t k_top, float Kc, hipStream_t s) {
    // zh_sweep_rows_per_launch sizes a launch as ~12 GB of F32 rows (~2 ms of HBM time: other queues get dispatch slots between launches); this sweep
    // reads 256-byte rows -- the same row count was a 1.1-ms launch whose tail (24.6k waves over 4096 wave slots: six rounds, the last one partly
    // empty) cost 6-7 %: twice the rows, the same ~2.2 ms (cfg5 shard, window 4: 359-367 -> 383-384 k QPS; four times: 390-392 k).  Last session:
    // four times (ZH_S128H_LAUNCH_X) for the halves, eight (ZH_S128B_LAUNCH_X) for the byte rows -- ~4.2-ms launches; the pipelined contexts' small
    // kernels did not suffer in the measurement that decided it (the pipelined loop's own QPS)
    const uint64_t rows_per_launch = zh_sweep128h_rows_per_launch(true, BYTES);
    for (uint64_t r = 0; r < R_grouped; r += rows_per_launch) {
        const uint64_t r_end = r + rows_per_launch < R_grouped ? r + rows_per_launch : R_grouped;
        constexpr int CHL = FUSE >= 0 ? (BYTES ? ZH_S128B_CH : ZH_S128F_CH) : ZH_S128L_CH;
        const uint64_t wl = (r_end - r + 64 * CHL - 1) / (64 * CHL);
        if constexpr (BYTES)
            hipLaunchKernelGGL((sweep128b_lean_kernel<CHL, FUSE>), dim3((uint32_t)((wl + 3) / 4)), dim3(256), 0, s, (const u32x4v *)dXh,
                               (const u32x4v *)dQh, dGroups, dGroupRowOff, n_groups, dWaveGroup, dLeafIds, r, r_end, dIv, ap, k_top, Kc);
        else
            hipLaunchKernelGGL((sweep128h_lean_kernel<CHL, FUSE>), dim3((uint32_t)((wl + 3) / 4)), dim3(256), 0, s, (const u32x4v *)dXh,
                               (const u32x4v *)dQh, inv, dGroups, dGroupRowOff, n_groups, dWaveGroup, dLeafIds, r, r_end, dIv, ap, k_top, Kc);
    }
#if ZH_S128L_EXP != 5    // (5: timing experiment without the boundary kernel, results invalid)
    // the chunks of three or more groups, of the WHOLE batch in one launch (a launch of its own per 25M rows was 0.04-0.06 ms each): a wave
    // looks at 64 chunks.  Chunk boundaries are absolute (every launch above starts on a multiple of 256 rows), so both kernels see the same chunks
    const uint64_t bw = ((R_grouped + 63) / 64 + 63) / 64, bb = (bw + 3) / 4;
    hipLaunchKernelGGL((sweep128h_boundary_kernel<FUSE, BYTES>), dim3((uint32_t)bb), dim3(256), 0, s, (const u32x4v *)dXh, (const u32x4v *)dQh, inv, dGroups,
                       dGroupRowOff, n_groups, dWaveGroup, dLeafIds, (uint64_t)0, R_grouped, dIv, ap, k_top, Kc);
#endif
}
// fuse: null = the raw pairs go to dIv for select_tau_kernel / select_emit_kernel; else the FUSED sweep (intervals, bounds and the queries' lists
// inside the sweep; exact_register_kernel instead of the select pass): `fuse_kinda` = approx_interval's kind, k_top <= 64
hipError_t zh_launch_sweep128h(const void *dXh, const void *dQh, float inv, const ZhGroup *dGroups, const uint64_t *dGroupRowOff, uint64_t n_groups,
                               const uint32_t *dWaveGroup, const uint32_t *dLeafIds, uint64_t R_grouped, uint64_t *dIv, hipStream_t s,
                               const ZhApprox *fuse, int fuse_kinda, uint32_t k_top, float Kc, bool byte_rows) {
    if (R_grouped == 0 || n_groups == 0) return hipSuccess;
    constexpr int CH = 4;
    const uint64_t rows_per_launch = zh_sweep_rows_per_launch(128);
    if ((R_grouped + 16383) / 16384 > 0x7FFFFFFFull) return hipErrorInvalidValue;
    // ZH_S128H_KERNEL (read per call: tests switch it): unset / "lean" = the round-6 kernels; "r5" = the round-5 register-staged kernel;
    // ZH_S128H_DMA=1: the LDS-DMA kernel, two tiles in flight per wave -- measured EQUAL to the register-staged one (1.528-1.533 against
    // 1.504-1.523 ms per launch on one box, profiles/r05_ab_sweep128h_dma.txt)
    const char *dma_e = getenv("ZH_S128H_DMA"), *kern_e = getenv("ZH_S128H_KERNEL");
    const bool dma = dma_e && dma_e[0] == '1', r5 = kern_e && kern_e[0] == 'r';
    if (fuse) {
        if (!dWaveGroup || k_top == 0 || k_top > 64) return hipErrorInvalidValue;
#define ZH_LEAN(F)                                                                                                                                  \
    do {                                                                                                                                            \
        if (byte_rows) launch_sweep128h_lean<F, true>(dXh, dQh, 1.f, dGroups, dGroupRowOff, n_groups, dWaveGroup, dLeafIds, R_grouped, dIv, *fuse, k_top, Kc, s); \
        else launch_sweep128h_lean<F, false>(dXh, dQh, inv, dGroups, dGroupRowOff, n_groups, dWaveGroup, dLeafIds, R_grouped, dIv, *fuse, k_top, Kc, s);           \
    } while (0)
        if (fuse_kinda == 0) ZH_LEAN(0);
        else if (fuse_kinda == 1) ZH_LEAN(1);
        else ZH_LEAN(2);
#undef ZH_LEAN
        return hipGetLastError();
    }
    if (!dma && !r5) {
        const ZhApprox none{};
        if (byte_rows) launch_sweep128h_lean<-1, true>(dXh, dQh, 1.f, dGroups, dGroupRowOff, n_groups, dWaveGroup, dLeafIds, R_grouped, dIv, none, 0u, 0.f, s);
        else launch_sweep128h_lean<-1, false>(dXh, dQh, inv, dGroups, dGroupRowOff, n_groups, dWaveGroup, dLeafIds, R_grouped, dIv, none, 0u, 0.f, s);
        return hipGetLastError();
    }
    if (byte_rows) return hipErrorInvalidValue;  // (the round-5 kernels read the copy of halves)
    for (uint64_t r = 0; r < R_grouped; r += rows_per_launch) {
        const uint64_t r_end = r + rows_per_launch < R_grouped ? r + rows_per_launch : R_grouped;
        const uint64_t w = (r_end - r + 64 * CH - 1) / (64 * CH), blocks = (w + 3) / 4;
        if (!dma)
            hipLaunchKernelGGL((sweep128h_kernel<CH>), dim3((uint32_t)blocks), dim3(256), 0, s, (const u32x4v *)dXh, (const u32x4v *)dQh, inv, dGroups,
                               dGroupRowOff, n_groups, dWaveGroup, dLeafIds, r, r_end, dIv);
        else
            hipLaunchKernelGGL((sweep128h_dma_kernel<CH>), dim3((uint32_t)blocks), dim3(256), 0, s, (const u32x4v *)dXh, (const u32x4v *)dQh, inv, dGroups,
                               dGroupRowOff, n_groups, dWaveGroup, dLeafIds, r, r_end, dIv);
    }
    return hipGetLastError();
}

// (Measured and not kept, round 4: phase 1 as a kernel of its own -- scan_pairs_kernel wrote every 16-row unit's pair records (8 bytes
// each, compacted through one atomic per unit) and a "listed" scan kernel started from them with independent loads only.  cfg3, window
// 2: the pairs kernel ~4 ms per batch, the listed scan 6.5 against 4.7 ms per launch beside it, 100 k against 142 k QPS.  Phase 1 is
// gather WORK -- one visit record per (row, visited tree), a third as many line requests as the rows themselves -- not a latency the
// fused kernel fails to hide: other waves' pair loops already run beside it.)

// ---- d = 128 (SIFT-style shards, cfg5): a row or a query is ONE 16-lane group's worth (8 elements per lane), and a stored row is
// wanted by only 3-6 queries of a window -- four groups on the same row (the kernel above) would idle half the wave.  Here the
// wave's 16 consecutive rows (8 KB, contiguous in memory: eight coalesced 1-KiB loads) wait in LDS and every group takes ANY pair
// of the wave's list: per step G pairs, each with its own row (ds_read_b128 from the wave's LDS copy) and its own query (dwordx4
// loads of halves), v_fma_mix_f32, a short DPP reduce.  G = 8 (8-lane groups, 16 elements per lane: two query loads, four row reads,
// sixteen fma, three DPP adds per EIGHT pairs) halves the per-pair share of everything that is not arithmetic -- record, addresses,
// reduce, stash -- which is what this kernel's time is (21 VALU per pair at G = 4 for 2 of arithmetic).  Same intervals, same stages behind it.
template <int G, int KINDA>
__global__ __launch_bounds__(256) void scan_approx128_kernel(const float *__restrict__ X, const uint4 *__restrict__ Qh,
                                                              const float4 *__restrict__ qmeta, const uint2 *__restrict__ rowLeaf,
                                                              uint32_t T, uint32_t RW, const uint32_t *__restrict__ visitBits,
                                                              const uint4 *__restrict__ nodeVisit, const ZhGroup *__restrict__ groups,
                                                              uint32_t GRP, uint64_t row_begin, uint64_t row_end, float Kc,
                                                              uint64_t *__restrict__ iv) {
    constexpr int LG = 64 / G, NH = G / 4, NR = 2 * NH;  // G = 4: 16-lane groups, 8 elements per lane; G = 8: 8-lane groups, 16 per lane
    // pair records packed into 8 bytes (interval slot: 36 bits | query: 24 bits | row of the wave: 4 bits) and a list of 240: 40 KB of
    // LDS per block with the rows, four blocks per CU
    constexpr uint32_t CAP = 240;
    __shared__ uint64_t pair_list[4][CAP];
    __shared__ float4 row_lds[4][16 * 32];  // the wave's rows, as in memory
    __shared__ float a2_lds[4][16];
    const uint32_t lane = threadIdx.x & 63, l = lane & (LG - 1), g = lane / LG;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wid;
    const uint64_t r0 = row_begin + wave * RW;
    if (r0 >= row_end) return;
    const uint32_t nr = (uint32_t)(row_end - r0 < RW ? row_end - r0 : RW);
    const uint32_t n_ent = nr * T;
    const uint2 *__restrict__ ent = rowLeaf + (size_t)r0 * T;
    uint32_t eGb[ZH_SCAN_NE], eWithin[ZH_SCAN_NE], eC[ZH_SCAN_NE], off[ZH_SCAN_NE], eB0[ZH_SCAN_NE];
    uint64_t eK0[ZH_SCAN_NE];
    const uint32_t P = scan_phase1(lane, n_ent, ent, visitBits, nodeVisit, eGb, eWithin, eC, off, eB0, eK0);
    if (P == 0) return;
    // the wave's rows into LDS (rows r0 .. r0 + nr - 1 are one contiguous piece of the table), and their squared norms
    {
        const float4 *src = reinterpret_cast<const float4 *>(X) + (size_t)r0 * 32;
        float4 t[8];
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = (lane + 64u * i < nr * 32u) ? ld16<true>(src + lane + 64 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; i++) row_lds[wid][lane + 64 * i] = t[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int rr = 0; rr < 16 / G; rr++) {  // group g: rows g, g + G, ...
        const uint32_t row = g + (uint32_t)G * rr;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NR; j++) sq4(row_lds[wid][row * 32 + LG * j + l], c);
        const float a2 = group_sum<LG>((c.x + c.y) + (c.z + c.w));
        if (l == 0) a2_lds[wid][row] = a2;
    }
    float my_s = 0.f;
    uint64_t my_slot = 0;
    uint32_t my_row = 0, nst = 0;
    bool my_on = false;
    auto flush = [&]() {  // the raw pair {x . h, |x|^2}: select_tau_kernel makes the interval (see scan_approx_kernel)
        if (my_on) __builtin_nontemporal_store(((uint64_t)__float_as_uint(a2_lds[wid][my_row]) << 32) | __float_as_uint(my_s), iv + my_slot);
        my_on = false;
        nst = 0;
    };
    struct Ops { float4 v[NR]; uint4 hq[NH]; };  // a pair's operands in a lane: its slice of the row (from LDS) and of the query's halves
    auto fetch = [&](uint32_t row, uint32_t b, Ops &o) {
#pragma unroll
        for (int i = 0; i < NH; i++) o.hq[i] = Qh[(size_t)b * 16 + i * LG + l];
#pragma unroll
        for (int j = 0; j < NR; j++) o.v[j] = row_lds[wid][row * 32 + LG * j + l];
    };
    auto dot8 = [&](const Ops &o) {
        float ax = 0.f, ay = 0.f, az = 0.f, aw = 0.f;
#pragma unroll
        for (int i = 0; i < NH; i++) {
            f16x8 h;
            __builtin_memcpy(&h, &o.hq[i], 16);
            ax = __builtin_fmaf(o.v[2 * i].x, (float)h[0], ax); ay = __builtin_fmaf(o.v[2 * i].y, (float)h[1], ay);
            az = __builtin_fmaf(o.v[2 * i].z, (float)h[2], az); aw = __builtin_fmaf(o.v[2 * i].w, (float)h[3], aw);
            ax = __builtin_fmaf(o.v[2 * i + 1].x, (float)h[4], ax); ay = __builtin_fmaf(o.v[2 * i + 1].y, (float)h[5], ay);
            az = __builtin_fmaf(o.v[2 * i + 1].z, (float)h[6], az); aw = __builtin_fmaf(o.v[2 * i + 1].w, (float)h[7], aw);
        }
        return group_sum<LG>((ax + ay) + (az + aw));
    };
    auto stash = [&](float s, uint32_t row, uint32_t, uint64_t slot, bool valid) {
        if (l == nst) { my_s = s; my_row = row; my_slot = slot; my_on = valid; }
        if (++nst == (uint32_t)LG) flush();
    };
    auto pack = [](uint32_t rl, uint32_t b, uint64_t slot) { return slot | ((uint64_t)b << 36) | ((uint64_t)rl << 60); };
    if (P <= CAP) {
        uint64_t *list = pair_list[wid];
#pragma unroll
        for (int j = 0; j < ZH_SCAN_NE; j++) {
            const uint32_t e = lane + 64u * j, c = eC[j];
            if (c) {
                const uint32_t rl = e / T, gb = eGb[j];
                {
                    const uint64_t slot = eK0[j] + eWithin[j];
                    list[off[j]] = pack(rl, eB0[j], slot);
                }
                for (uint32_t sidx = 1; sidx < c; sidx++) {
                    const ZhGroup *gp = groups + gb + sidx / GRP;
                    const uint64_t slot = gp->key_off[sidx % GRP] + eWithin[j];
                    list[off[j] + sidx] = pack(rl, gp->b[sidx % GRP], slot);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // G pairs per step, the next step's records, queries and rows requested before the current step is computed
        auto r_row = [](uint64_t r) { return (uint32_t)(r >> 60); };
        auto r_b = [](uint64_t r) { return (uint32_t)(r >> 36) & 0xFFFFFFu; };
        uint64_t rec = list[g < P ? g : P - 1];
        Ops cur;
        fetch(r_row(rec), r_b(rec), cur);
        for (uint32_t p = 0; p < P; p += G) {
            const uint32_t pn = p + G + g;
            const uint64_t recn = list[pn < P ? pn : P - 1];
            Ops nxt;
            fetch(r_row(recn), r_b(recn), nxt);
            const float s = dot8(cur);
            stash(s, r_row(rec), r_b(rec), rec & 0xFFFFFFFFFull, p + g < P);
            rec = recn; cur = nxt;
        }
    } else {
        // more pairs than the list holds (hot leaves): entry after entry, a leaf's visits four at a time
#pragma unroll
        for (int j = 0; j < ZH_SCAN_NE; j++) {
            unsigned long long m = __ballot(eC[j] != 0);
            while (m) {
                const int ll = __builtin_ctzll(m);
                m &= m - 1;
                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)eC[j], ll);
                const uint32_t gb = (uint32_t)__builtin_amdgcn_readlane((int)eGb[j], ll);
                const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)eWithin[j], ll);
                const uint32_t rl = ((uint32_t)ll + 64u * j) / T;
                for (uint32_t s0 = 0; s0 < c; s0 += G) {
                    const uint32_t sidx = s0 + g < c ? s0 + g : c - 1;
                    const ZhGroup *gp = groups + gb + sidx / GRP;
                    const uint32_t b = gp->b[sidx % GRP];
                    Ops o;
                    fetch(rl, b, o);
                    const float s = dot8(o);
                    stash(s, rl, b, gp->key_off[sidx % GRP] + w, s0 + g < c);
                }
            }
        }
    }
    flush();
}

template <int D, int G>
static hipError_t launch_scan_approx_d(const float *dX, uint64_t n_rows, const ZhApprox &ap, const uint2 *dRowLeaf, uint32_t T,
                                       const uint32_t *dVisitBits, const uint4 *dNodeVisit, const ZhGroup *dGroups, uint32_t group,
                                       int metric, int mode, hipStream_t s) {
    const uint32_t RW = zh_scan_rows_per_wave(T);
    uint64_t rows_per_launch = zh_sweep_rows_per_launch(D);
    rows_per_launch = rows_per_launch / (4 * RW) * (4 * RW);
    const float Kc = zh_approx_bound(metric, D, 0);
    const int kinda = metric == ZH_COSINE ? (mode == ZH_COSINE_PARITY ? 2 : 1) : 0;
    for (uint64_t r = 0; r < n_rows; r += rows_per_launch) {
        const uint64_t r_end = r + rows_per_launch < n_rows ? r + rows_per_launch : n_rows;
        const uint64_t waves = (r_end - r + RW - 1) / RW, blocks = (waves + 3) / 4;
        if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
#define ZH_APX_LAUNCH(KA) \
        hipLaunchKernelGGL((scan_approx_kernel<D, G, KA>), dim3((uint32_t)blocks), dim3(256), 0, s, dX, (const uint4 *)ap.Qh, ap.qmeta, dRowLeaf, T, \
                           RW, dVisitBits, dNodeVisit, dGroups, group, r, r_end, Kc, ap.iv)
        if (kinda == 0) ZH_APX_LAUNCH(0);
        else if (kinda == 1) ZH_APX_LAUNCH(1);
        else ZH_APX_LAUNCH(2);
#undef ZH_APX_LAUNCH
    }
    return hipGetLastError();
}

// (16 rows per wave = one tile of the fp16 row copy: up to 16 trees)
bool zh_scan_mfma_supported(uint32_t d, uint32_t T) { return (d == 256 || d == 384 || d == 512 || d == 768 || d == 1024) && zh_scan_rows_per_wave(T) == 16; }
bool zh_scan_approx_supported(uint32_t d, uint32_t T, int metric) {
    if (metric != ZH_L2 && metric != ZH_L2SQ && metric != ZH_COSINE) return false;
    return zh_approx_groups(d) != 0 && zh_scan_rows_per_wave(T) != 0;
}

hipError_t zh_launch_scan_approx(const float *dX, uint32_t d, uint64_t n_rows, ZhApprox ap, const uint2 *dRowLeaf, uint32_t T,
                                 const uint32_t *dVisitBits, const uint4 *dNodeVisit, const ZhGroup *dGroups, uint32_t group, int metric,
                                 int mode, hipStream_t s) {
    if (!n_rows) return hipSuccess;
    if (ap.mfma) switch (d) {
    case 256: return launch_scan_mfma_d<256>(dX, n_rows, ap, dRowLeaf, T, dVisitBits, dNodeVisit, dGroups, group, s);
    case 384: return launch_scan_mfma_d<384>(dX, n_rows, ap, dRowLeaf, T, dVisitBits, dNodeVisit, dGroups, group, s);
    case 512: return launch_scan_mfma_d<512>(dX, n_rows, ap, dRowLeaf, T, dVisitBits, dNodeVisit, dGroups, group, s);
    case 768: return launch_scan_mfma_d<768>(dX, n_rows, ap, dRowLeaf, T, dVisitBits, dNodeVisit, dGroups, group, s);
    case 1024: return launch_scan_mfma_d<1024>(dX, n_rows, ap, dRowLeaf, T, dVisitBits, dNodeVisit, dGroups, group, s);
    default: return hipErrorInvalidValue;
    }
    switch (d) {
    case 128: {
        const uint32_t RW = zh_scan_rows_per_wave(T);
        uint64_t rows_per_launch = zh_sweep_rows_per_launch(128);
        rows_per_launch = rows_per_launch / (4 * RW) * (4 * RW);
        const float Kc = zh_approx_bound(metric, 128, 0);
        const int kinda = metric == ZH_COSINE ? (mode == ZH_COSINE_PARITY ? 2 : 1) : 0;
        for (uint64_t r = 0; r < n_rows; r += rows_per_launch) {
            const uint64_t r_end = r + rows_per_launch < n_rows ? r + rows_per_launch : n_rows;
            const uint64_t waves = (r_end - r + RW - 1) / RW, blocks = (waves + 3) / 4;
            if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
#define ZH_APX_LAUNCH(KA) \
            hipLaunchKernelGGL((scan_approx128_kernel<ZH_APX_G128, KA>), dim3((uint32_t)blocks), dim3(256), 0, s, dX, (const uint4 *)ap.Qh, ap.qmeta, dRowLeaf, T, RW, \
                               dVisitBits, dNodeVisit, dGroups, group, r, r_end, Kc, ap.iv)
            if (kinda == 0) ZH_APX_LAUNCH(0);
            else if (kinda == 1) ZH_APX_LAUNCH(1);
            else ZH_APX_LAUNCH(2);
#undef ZH_APX_LAUNCH
        }
        return hipGetLastError();
    }
    case 256: return launch_scan_approx_d<256, 4>(dX, n_rows, ap, dRowLeaf, T, dVisitBits, dNodeVisit, dGroups, group, metric, mode, s);
    case 384: return launch_scan_approx_d<384, 4>(dX, n_rows, ap, dRowLeaf, T, dVisitBits, dNodeVisit, dGroups, group, metric, mode, s);
    case 512: return launch_scan_approx_d<512, 2>(dX, n_rows, ap, dRowLeaf, T, dVisitBits, dNodeVisit, dGroups, group, metric, mode, s);
    case 768: return launch_scan_approx_d<768, ZH_APX_G768>(dX, n_rows, ap, dRowLeaf, T, dVisitBits, dNodeVisit, dGroups, group, metric, mode, s);
    case 1024: return launch_scan_approx_d<1024, 2>(dX, n_rows, ap, dRowLeaf, T, dVisitBits, dNodeVisit, dGroups, group, metric, mode, s);
    default: return hipErrorInvalidValue;
    }
}

// ---- select on intervals: what a visit hands on ----
// block-wide radix select: the `need`-th smallest (1-based) of the values VAL(i), i < n, 8 bits per round from the top
#define ZH_APX_RADIX_SELECT(n, need_in, VAL, result)                                                        \
    {                                                                                                       \
        uint32_t prefix_ = 0, need_ = (need_in);                                                            \
        for (int sh_ = 24; sh_ >= 0; sh_ -= 8) {                                                            \
            __syncthreads();                                                                                \
            hist[tid] = 0;                                                                                  \
            __syncthreads();                                                                                \
            const uint32_t himask_ = sh_ == 24 ? 0u : (0xFFFFFFFFu << (sh_ + 8));                           \
            for (uint32_t i_ = tid; i_ < (n); i_ += 256) {                                                  \
                const uint32_t x_ = (VAL(i_));                                                              \
                if ((x_ & himask_) == prefix_) atomicAdd(&hist[(x_ >> sh_) & 255u], 1u);                    \
            }                                                                                               \
            __syncthreads();                                                                                \
            if (tid < 64) { /* wave 0: the bucket that holds the need-th value */                           \
                const uint32_t h0_ = hist[4 * tid], h1_ = hist[4 * tid + 1], h2_ = hist[4 * tid + 2], h3_ = hist[4 * tid + 3]; \
                const uint32_t ss_ = h0_ + h1_ + h2_ + h3_;                                                 \
                uint32_t inc_ = ss_;                                                                        \
                for (int m_ = 1; m_ < 64; m_ <<= 1) {                                                       \
                    const uint32_t t_ = __shfl_up(inc_, m_);                                                \
                    if ((int)tid >= m_) inc_ += t_;                                                         \
                }                                                                                           \
                const uint32_t exc_ = inc_ - ss_;                                                           \
                if (exc_ < need_ && need_ <= inc_) {                                                        \
                    uint32_t c_ = exc_, j_ = 4 * tid;                                                       \
                    if (need_ > c_ + h0_) { c_ += h0_; j_++;                                                \
                        if (need_ > c_ + h1_) { c_ += h1_; j_++;                                            \
                            if (need_ > c_ + h2_) { c_ += h2_; j_++; } } }                                  \
                    s_u[6] = j_; s_u[7] = c_;                                                               \
                }                                                                                           \
            }                                                                                               \
            __syncthreads();                                                                                \
            prefix_ |= s_u[6] << sh_;                                                                       \
            need_ -= s_u[7];                                                                                \
        }                                                                                                   \
        (result) = prefix_;                                                                                 \
    }

// pass 1: per visit that takes top_k of a longer leaf, tau = the take-th smallest hi (top_k rows of the visit have keys at or
// below it); the smallest tau of a query's visits bounds the query's top_k-th key as well (those rows are candidates)
template <int KINDA>
__global__ __launch_bounds__(256) void select_tau_kernel(const ZhVisit *__restrict__ visits, uint64_t n_visits, uint32_t chunk,
                                                          uint32_t k_top, float Kc, ZhApprox ap) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_u[8];
    const uint32_t tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * chunk;
    const uint32_t cnt = (uint32_t)(n_visits - base < chunk ? n_visits - base : chunk);
    for (uint32_t c = 0; c < cnt; c++) {  // block-uniform
        const ZhVisit v = visits[base + c];
        if (v.take) {  // the scan's raw pairs {x . h, |x|^2} of this visit -> intervals, in place (one load of the query's constants per visit)
            const float4 qm = ap.qmeta[v.b];
            uint64_t *__restrict__ raw = ap.iv + v.row_off;
            for (uint32_t i = tid; i < v.len; i += 256) {
                const uint64_t w = raw[i];
                raw[i] = approx_interval<KINDA>(__uint_as_float((uint32_t)w), __uint_as_float((uint32_t)(w >> 32)), qm, Kc, ap.row_rho, ap.rho_norm);
            }
            __syncthreads();
        }
        uint32_t tau = 0xFFFFFFFFu;
        if (v.take && v.take < v.len) {
            if (v.take < k_top) {  // membership matters (see the header of this file): the exact path
                if (tid == 0) {
                    const uint32_t slot = atomicAdd(&ap.ctl[0], 1u);
                    const uint32_t eb = atomicAdd(&ap.ctl[2], v.len);  // its slice of the exact-key scratch
                    if (slot >= ap.ex_cap) atomicOr(&ap.ctl[1], 4u);
                    else if ((uint64_t)eb + v.len > ap.ex_rows_cap) { atomicOr(&ap.ctl[1], 8u); ap.ex_visits[slot] = make_uint2(0xFFFFFFFFu, 0u); }
                    else ap.ex_visits[slot] = make_uint2((uint32_t)(base + c), eb);
                    ap.tauv[base + c] = 0u;  // (its rows join the list from exact_visit_kernel, not from the emit pass)
                }
                continue;
            }
            const uint64_t *__restrict__ kp = ap.iv + v.row_off;
#define ZH_APX_HI(i) ((uint32_t)(kp[i] >> 32))
            ZH_APX_RADIX_SELECT(v.len, v.take, ZH_APX_HI, tau);
#undef ZH_APX_HI
            if (tid == 0) atomicMin(&ap.qtau[v.b], tau);
        }
        if (tid == 0) ap.tauv[base + c] = tau;
    }
}

// pass 2: a visit hands on the rows whose lo exceeds neither its own tau nor the query's
__global__ __launch_bounds__(256) void select_emit_kernel(const ZhVisit *__restrict__ visits, uint64_t n_visits, uint32_t chunk,
                                                           uint32_t k_top, const uint32_t *__restrict__ leaf_ids, ZhApprox ap) {
    __shared__ uint32_t s_u[8];
    const uint32_t tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * chunk;
    const uint32_t cnt = (uint32_t)(n_visits - base < chunk ? n_visits - base : chunk);
    for (uint32_t c = 0; c < cnt; c++) {  // block-uniform
        const ZhVisit v = visits[base + c];
        if (!v.take || (v.take < v.len && v.take < k_top)) continue;
        const uint64_t *__restrict__ kp = ap.iv + v.row_off;
        const uint32_t tv = ap.tauv[base + c], tq = ap.qtau[v.b];
        const uint32_t tau = tv < tq ? tv : tq;
        __syncthreads();
        if (tid == 0) { s_u[0] = 0; s_u[2] = 0; }
        __syncthreads();
        uint32_t mine = 0;
        for (uint32_t i = tid; i < v.len; i += 256) mine += ((uint32_t)kp[i] <= tau) ? 1u : 0u;
        if (mine) atomicAdd(&s_u[0], mine);
        __syncthreads();
        const uint32_t M = s_u[0];
        if (!M) continue;
        if (tid == 0) {
            const uint32_t b0 = atomicAdd(&ap.qcount[v.b], M);
            s_u[1] = b0;
            if (b0 + M > ap.capq) atomicOr(&ap.ctl[1], 1u);
        }
        __syncthreads();
        const uint32_t b0 = s_u[1];
        if (b0 + M <= ap.capq) {
            const size_t ob = (size_t)v.b * ap.capq + b0;
            for (uint32_t i = tid; i < v.len; i += 256) {
                const uint64_t w = kp[i];
                if ((uint32_t)w <= tau) {
                    const uint32_t pos = atomicAdd(&s_u[2], 1u);
                    ap.list_lo[ob + pos] = (uint32_t)w;
                    ap.list_hi[ob + pos] = (uint32_t)(w >> 32);
                    ap.list_id[ob + pos] = leaf_ids[(size_t)v.leaf_off + i];
                }
            }
        }
    }
}

// a visit that hands over fewer than top_k rows: the leaf scored with the canonical sums (exact_keys_kernel: a block per 256 rows of
// a visit, so that a handful of long leaves does not serialise on a handful of blocks), the `take` smallest (key, id) -- what
// sweep + select do -- join the query's list with their intervals (their keys are computed again with the other survivors')
template <int KIND>
__global__ __launch_bounds__(256) void exact_keys_kernel(const ZhVisit *__restrict__ visits, const float *__restrict__ X, uint32_t d,
                                                          const float *__restrict__ Q, const float *__restrict__ QQ,
                                                          const uint32_t *__restrict__ leaf_ids, int metric, int param, ZhApprox ap) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (ap.ctl[0] > ap.ex_cap) return;  // the table ran over (flagged by select_tau): the batch is redone
    const uint32_t total = ap.ctl[0];
    for (uint32_t e = blockIdx.x; e < total; e += gridDim.x) {
        const uint2 ev = ap.ex_visits[e];
        if (ev.x == 0xFFFFFFFFu) continue;
        const ZhVisit v = visits[ev.x];
        const uint32_t i0 = blockIdx.y * 256u + wv * 64u;
        for (uint32_t i = i0; i < v.len && i < i0 + 64u; i++) {
            const uint32_t id = leaf_ids[(size_t)v.leaf_off + i];
            float s0, s1;
            lane_sums_generic<KIND>(X + (size_t)id * d, Q + (size_t)v.b * d, d, lane, param, s0, s1);
            if (lane == 0) ap.ex_keys[(size_t)ev.y + i] = key_of(metric, param, s0, s1, KIND == K_COS ? QQ[v.b] : 0.f);
        }
    }
}

__global__ __launch_bounds__(256) void exact_visit_kernel(const ZhVisit *__restrict__ visits, const uint32_t *__restrict__ leaf_ids, ZhApprox ap) {
    __shared__ uint64_t sk[2048];
    __shared__ __attribute__((aligned(16))) uint32_t si[2048];
    __shared__ uint32_t s_u32[8];
    __shared__ uint64_t s_red[8];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (ap.ctl[0] > ap.ex_cap) return;
    const uint32_t total = ap.ctl[0];
    for (uint32_t e = blockIdx.x; e < total; e += gridDim.x) {  // block-uniform
        const uint2 ev = ap.ex_visits[e];
        if (ev.x == 0xFFFFFFFFu) continue;
        const ZhVisit v = visits[ev.x];
        const uint32_t eb = ev.y;
        __syncthreads();
        ZhVisit v2 = v;
        v2.row_off = eb; v2.cand_off = eb;
        bool need_slow = false;
        select_fast<false>(v2, leaf_ids, ap.ex_keys, ap.ex_ckeys, ap.ex_cids, sk, si, s_u32, need_slow);
        if (need_slow) {
            __syncthreads();
            select_slow(v2, leaf_ids, ap.ex_keys, ap.ex_ckeys, ap.ex_cids, sk, si, 2048);
        }
        __syncthreads();
        // the largest (key, id) of the take chosen = the visit's threshold
        uint64_t tk = 0, ti = 0;
        for (uint32_t i = tid; i < v.take; i += 256) {
            const uint64_t kk = ap.ex_ckeys[(size_t)eb + i], ii = ap.ex_cids[(size_t)eb + i];
            if (kk > tk || (kk == tk && ii > ti)) { tk = kk; ti = ii; }
        }
        for (int m = 1; m < 64; m <<= 1) {
            const uint64_t ok = __shfl_xor(tk, m), oi = __shfl_xor(ti, m);
            if (ok > tk || (ok == tk && oi > ti)) { tk = ok; ti = oi; }
        }
        if (lane == 0) { s_red[wv] = tk; s_red[4 + wv] = ti; }
        __syncthreads();
        tk = s_red[0]; ti = s_red[4];
        for (int w = 1; w < 4; w++)
            if (s_red[w] > tk || (s_red[w] == tk && s_red[4 + w] > ti)) { tk = s_red[w]; ti = s_red[4 + w]; }
        // of the `take` chosen, those the query's tau does not rule out join the list
        if (tid == 0) s_u32[7] = 0;
        __syncthreads();
        const uint32_t tq = ap.qtau[v.b];
        uint32_t mine = 0;
        for (uint32_t i = tid; i < v.len; i += 256) {
            const uint64_t kk = ap.ex_keys[(size_t)eb + i];
            const uint32_t id = leaf_ids[(size_t)v.leaf_off + i];
            if ((kk < tk || (kk == tk && id <= ti)) && (uint32_t)ap.iv[v.row_off + i] <= tq) mine++;
        }
        if (mine) atomicAdd(&s_u32[7], mine);
        __syncthreads();
        const uint32_t M = s_u32[7];
        __syncthreads();
        if (tid == 0) {
            const uint32_t b0 = M ? atomicAdd(&ap.qcount[v.b], M) : 0u;
            s_u32[6] = b0; s_u32[7] = 0;
            if (M && b0 + M > ap.capq) atomicOr(&ap.ctl[1], 1u);
        }
        __syncthreads();
        const uint32_t b0 = s_u32[6];
        if (M && b0 + M <= ap.capq) {
            const size_t ob = (size_t)v.b * ap.capq + b0;
            for (uint32_t i = tid; i < v.len; i += 256) {
                const uint64_t kk = ap.ex_keys[(size_t)eb + i];
                const uint32_t id = leaf_ids[(size_t)v.leaf_off + i];
                const uint64_t w = ap.iv[v.row_off + i];
                if ((kk < tk || (kk == tk && id <= ti)) && (uint32_t)w <= tq) {
                    const uint32_t pos = atomicAdd(&s_u32[7], 1u);
                    ap.list_lo[ob + pos] = (uint32_t)w;
                    ap.list_hi[ob + pos] = (uint32_t)(w >> 32);
                    ap.list_id[ob + pos] = id;
                }
            }
        }
        __syncthreads();
    }
}

// ---- per query: duplicates out, tau, the survivors' canonical keys, top_k (lsh.rs:557-564) ----
// LCAP = entries of a query's list the sort holds (and, after it, its survivors): 4096 (48 KB of LDS per block), or 8192 (96 KB: one
// block per CU) for an index whose lists ran over 4096 once -- keys that are dense around the cut (the parity cosine key on iid rows in
// 20k-row leaves: ~3000 rows per query inside the bound)
// the reference's key of stored row `id` against the block's query: canonical sums (D > 0: the specialised row loads, the query's
// float4s in registers; D == 0: any d)
template <int D, int KIND>
__device__ __forceinline__ uint64_t exact_key(const float *__restrict__ X, uint32_t d, uint32_t id, const float *__restrict__ q,
                                              const float4 *qreg, float qq, uint32_t lane, int metric, int param) {
    float s0 = 0.f, s1 = 0.f;
    if constexpr (D > 0) {
        constexpr int NV = RowVec<D>::NV;
        float4 v[NV];
        load_row<D>(X + (size_t)id * D, lane, v);
        row_pair_sums<D, KIND>(v, qreg, lane, param, s0, s1);
        if (KIND == K_COS) {
            float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < NV; j++) {
                const bool act = (j < RowVec<D>::NJ) || (lane < (uint32_t)RowVec<D>::REM4);
                if (act) sq4(v[j], c);
            }
            s1 = wave_sum_canonical((c.x + c.y) + (c.z + c.w));
        }
    } else
        lane_sums_generic<KIND>(X + (size_t)id * d, q, d, lane, param, s0, s1);
    return key_of(metric, param, s0, s1, qq);
}

// Round 5: three kernels instead of one block per query doing everything.  The survivors' canonical keys were computed INSIDE the
// query's block -- four waves, two rows in flight each -- so a query with thousands of survivors (the literal cosine key on 20k-row leaves: 3120 per
// query at 64M rows, 879 on a cfg4 shard) kept its block, and 96 KB of LDS, for 7.6-23 ms.  Now: (1) final_survivors_kernel, a block per query:
// duplicates out, tau, the survivors' ids back to the query's list slots; (2) final_exact_kernel, waves over ALL queries' survivors (grid.y = query,
// a wave strides over the query's survivors with the query in registers): the reference's arithmetic at the rate HBM gives random rows; (3)
// final_topk_kernel, a block per query: sort by (key, id), top_k (lsh.rs:557-564).  The survivors' keys live where the lists' lo / hi words were
// (both arrays are dead once (1) has run; list_lo and list_hi are contiguous: B * capq u64 slots).
// `only` (an index whose lists have 8192 slots runs BOTH instantiations over all queries): 1 = the queries whose list holds <= 4096 entries (this
// instantiation sorts in 48 KB of LDS: three blocks per CU), 2 = the longer ones (96 KB: one block per CU), 0 = every query
template <int LCAP>
__global__ __launch_bounds__(256) void final_survivors_kernel(uint32_t B, uint32_t k, ZhApprox ap, uint32_t only) {
    __shared__ uint64_t sk[LCAP];  // id << 32 | sortable hi: equal ids end up side by side
    __shared__ uint32_t sl[LCAP];  // sortable lo; later: the survivors' ids
    __shared__ uint32_t hist[256], scan[256], s_u[8];
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    // something ran over in select_interval / exact_visit (complete by now: same stream): a list whose count ran past its
    // capacity has slots nobody wrote -- nothing here may be dereferenced, and the f32 scan behind redoes the batch anyway
    if (ap.ctl[1] & (1u | 4u | 8u)) return;
    uint32_t n = ap.qcount[b];
    if (n > ap.capq) n = ap.capq;
    if ((only == 1 && n > 4096u) || (only == 2 && n <= 4096u)) return;  // (the other instantiation's query; block-uniform)
    if (n > (uint32_t)LCAP) n = LCAP;  // capq <= LCAP
    const size_t ob = (size_t)b * ap.capq;
    for (uint32_t i = tid; i < n; i += 256) {
        sk[i] = ((uint64_t)ap.list_id[ob + i] << 32) | ap.list_hi[ob + i];
        sl[i] = ap.list_lo[ob + i];
    }
    const uint32_t np2 = next_pow2(n);
    for (uint32_t i = n + tid; i < np2; i += 256) { sk[i] = ~0ull; sl[i] = ~0u; }
    block_bitonic_sort<uint32_t>(sk, sl, np2);
    // first entry of every id (the smallest hi of its copies -- the copies are identical in fact: same row registers, same
    // group arithmetic -- ), compacted to the front
    constexpr uint32_t PER = LCAP / 256;
    uint64_t ek[PER];
    uint32_t el[PER], cntl = 0;
    bool keep[PER];
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) {
        const uint32_t i = tid * PER + j;
        keep[j] = i < n && (i == 0 || (uint32_t)(sk[i] >> 32) != (uint32_t)(sk[i - 1] >> 32));
        ek[j] = i < n ? sk[i] : 0ull;
        el[j] = i < n ? sl[i] : 0u;
        cntl += keep[j] ? 1u : 0u;
    }
    auto block_scan = [&](uint32_t mine) {  // exclusive prefix of `mine` over the threads; scan[255] = the total
        scan[tid] = mine;
        __syncthreads();
        for (uint32_t o = 1; o < 256; o <<= 1) {
            const uint32_t a = tid >= o ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += a;
            __syncthreads();
        }
        return scan[tid] - mine;
    };
    uint32_t rank = block_scan(cntl);
    const uint32_t nu = scan[255];
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < PER; j++)
        if (keep[j]) { sk[rank] = ek[j]; sl[rank] = el[j]; rank++; }
    __syncthreads();
    // tau = the k-th smallest hi among the distinct rows; fewer than k: everything stays
    uint32_t tau = 0xFFFFFFFFu;
    if (nu > k && k > 0) {
#define ZH_APX_HI(i) ((uint32_t)sk[i])
        ZH_APX_RADIX_SELECT(nu, k, ZH_APX_HI, tau);
#undef ZH_APX_HI
    }
    __syncthreads();
    // the survivors (lo <= tau): their ids go back to the front of the query's list slots (everything of the list is in LDS by now)
    cntl = 0;
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) {
        const uint32_t i = tid * PER + j;
        keep[j] = i < nu && sl[i] <= tau;
        el[j] = i < nu ? (uint32_t)(sk[i] >> 32) : 0u;
        cntl += keep[j] ? 1u : 0u;
    }
    rank = block_scan(cntl);
    const uint32_t ns = scan[255];
#pragma unroll
    for (uint32_t j = 0; j < PER; j++)
        if (keep[j]) ap.list_id[ob + rank++] = el[j];
    if (tid == 0) {
        ap.qcount[b] = ns;  // (from here on: the query's survivors)
        atomicAdd(&ap.ctl[3], ns);
        atomicAdd(&ap.ctl[4], n);
    }
}

// the survivors' keys, the reference's arithmetic (canonical sums): grid (SX, B); the block's four waves stride over query blockIdx.y's
// survivors, two rows in flight per wave, the query's float4s in registers (D > 0) -- a wave per ROW over all queries' survivors
template <int D, int KIND>
__global__ __launch_bounds__(256) void final_exact_kernel(const float *__restrict__ X, uint32_t d, const float *__restrict__ Q,
                                                           const float *__restrict__ QQ, int metric, int param, uint32_t b_first, ZhApprox ap) {
    if (ap.ctl[1] & (1u | 4u | 8u)) return;
    const uint32_t b = b_first + blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t ns = ap.qcount[b], stride = gridDim.x * 4u, w0 = blockIdx.x * 4u + wv;
    if (w0 >= ns) return;
    const size_t ob = (size_t)b * ap.capq;
    uint64_t *__restrict__ skeys = reinterpret_cast<uint64_t *>(ap.list_lo) + ob;
    const float *q = Q + (size_t)b * d;
    const float qq = KIND == K_COS ? QQ[b] : 0.f;
    float4 qreg[D > 0 ? RowVec<(D > 0 ? D : 4)>::NV : 1];
    if constexpr (D > 0) load_row<D>(q, lane, qreg);
    for (uint32_t i = w0; i < ns; i += 2 * stride) {
        const uint32_t i2 = i + stride;
        const uint64_t k0 = exact_key<D, KIND>(X, d, ap.list_id[ob + i], q, qreg, qq, lane, metric, param);
        uint64_t k1 = 0;
        if (i2 < ns) k1 = exact_key<D, KIND>(X, d, ap.list_id[ob + i2], q, qreg, qq, lane, metric, param);
        if (lane == 0) {
            skeys[i] = k0;
            if (i2 < ns) skeys[i2] = k1;
        }
    }
}

template <int LCAP>
__global__ __launch_bounds__(256) void final_topk_kernel(uint32_t B, uint32_t k, uint64_t id_base, ZhApprox ap, uint64_t *__restrict__ out_ids,
                                                          uint64_t *__restrict__ out_keys, uint32_t *__restrict__ out_counts, uint32_t only) {
    __shared__ uint64_t sk[LCAP];
    __shared__ uint32_t sl[LCAP];
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    if (ap.ctl[1] & (1u | 4u | 8u)) return;
    uint32_t ns = ap.qcount[b];
    if ((only == 1 && ns > 4096u) || (only == 2 && ns <= 4096u)) return;  // (survivors: as final_survivors_kernel's `only`)
    if (ns > (uint32_t)LCAP) ns = LCAP;
    const size_t ob = (size_t)b * ap.capq;
    const uint64_t *__restrict__ skeys = reinterpret_cast<const uint64_t *>(ap.list_lo) + ob;
    const uint32_t sp2 = next_pow2(ns);
    for (uint32_t i = tid; i < sp2; i += 256) {
        sk[i] = i < ns ? skeys[i] : ~0ull;
        sl[i] = i < ns ? ap.list_id[ob + i] : ~0u;
    }
    block_bitonic_sort<uint32_t>(sk, sl, sp2);
    const uint32_t have = ns < k ? ns : k;
    for (uint32_t i = tid; i < k; i += 256) {
        out_ids[(size_t)b * k + i] = i < have ? id_base + sl[i] : ~0ull;
        out_keys[(size_t)b * k + i] = i < have ? sk[i] : ~0ull;
    }
    if (tid == 0) out_counts[b] = have;
}

hipError_t zh_launch_select_interval(const ZhVisit *dVisits, uint64_t n_visits, uint32_t k, const uint32_t *dLeafIds, ZhApprox ap,
                                     int metric, int mode, uint32_t d, hipStream_t s) {
    if (!n_visits) return hipSuccess;
    if (n_visits > 0x7FFFFFFFull) return hipErrorInvalidValue;
    uint64_t chunk = (n_visits + 16383) / 16384;
    if (chunk > 256) chunk = 256;
    const uint64_t blocks = (n_visits + chunk - 1) / chunk;
    const float Kc = zh_approx_bound(metric, d, (int)ap.mfma);
    if (metric != ZH_COSINE)
        hipLaunchKernelGGL(select_tau_kernel<0>, dim3((uint32_t)blocks), dim3(256), 0, s, dVisits, n_visits, (uint32_t)chunk, k, Kc, ap);
    else if (mode == ZH_COSINE_PARITY)
        hipLaunchKernelGGL(select_tau_kernel<2>, dim3((uint32_t)blocks), dim3(256), 0, s, dVisits, n_visits, (uint32_t)chunk, k, Kc, ap);
    else
        hipLaunchKernelGGL(select_tau_kernel<1>, dim3((uint32_t)blocks), dim3(256), 0, s, dVisits, n_visits, (uint32_t)chunk, k, Kc, ap);
    hipLaunchKernelGGL(select_emit_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, dVisits, n_visits, (uint32_t)chunk, k, dLeafIds, ap);
    return hipGetLastError();
}

template <int KIND>
static void launch_final_interval_k(const ZhVisit *dVisits, const float *dX, uint32_t d, const float *dQ, const float *dQQ, uint32_t B,
                                    uint32_t k, const uint32_t *dLeafIds, int metric, int mode, uint64_t id_base, const ZhApprox &ap,
                                    uint64_t *dOutIds, uint64_t *dOutKeys, uint32_t *dOutCounts, uint32_t max_leaf_len, hipStream_t s) {
    hipLaunchKernelGGL(exact_keys_kernel<KIND>, dim3(512, (max_leaf_len + 255) / 256), dim3(256), 0, s, dVisits, dX, d, dQ, dQQ, dLeafIds, metric, mode, ap);
    hipLaunchKernelGGL(exact_visit_kernel, dim3(1024), dim3(256), 0, s, dVisits, dLeafIds, ap);
    // (1) survivors per query, (2) their keys by waves over all queries' survivors, (3) top_k per query
    const uint32_t SX = ap.capq > 4096 ? 16u : 8u;  // blocks of four waves per query in (2): a wave with no survivor left returns at once
#define ZH_APX_EXACT(DD) \
    for (uint32_t b0 = 0; b0 < B; b0 += 65535u) /* (gridDim.y <= 65535) */ \
        hipLaunchKernelGGL((final_exact_kernel<DD, KIND>), dim3(SX, std::min(B - b0, 65535u)), dim3(256), 0, s, dX, d, dQ, dQQ, metric, mode, b0, ap)
    // 8192-slot lists (an index whose lists ran over once): most queries' lists still hold <= 4096 entries -- those sort in the 48-KB instantiation
    // (three blocks per CU), only the longer ones in the 96-KB one (r05_scale64m_kernel_stats.csv: 4.4 ms per window with every query in the latter)
    if (ap.capq > 4096) {
        hipLaunchKernelGGL((final_survivors_kernel<4096>), dim3(B), dim3(256), 0, s, B, k, ap, 1u);
        hipLaunchKernelGGL((final_survivors_kernel<8192>), dim3(B), dim3(256), 0, s, B, k, ap, 2u);
    } else hipLaunchKernelGGL((final_survivors_kernel<4096>), dim3(B), dim3(256), 0, s, B, k, ap, 0u);
    switch (d) {
    case 128: ZH_APX_EXACT(128); break;
    case 384: ZH_APX_EXACT(384); break;
    case 768: ZH_APX_EXACT(768); break;
    default: ZH_APX_EXACT(0); break;
    }
#undef ZH_APX_EXACT
    if (ap.capq > 4096) {
        hipLaunchKernelGGL((final_topk_kernel<4096>), dim3(B), dim3(256), 0, s, B, k, id_base, ap, dOutIds, dOutKeys, dOutCounts, 1u);
        hipLaunchKernelGGL((final_topk_kernel<8192>), dim3(B), dim3(256), 0, s, B, k, id_base, ap, dOutIds, dOutKeys, dOutCounts, 2u);
    } else hipLaunchKernelGGL((final_topk_kernel<4096>), dim3(B), dim3(256), 0, s, B, k, id_base, ap, dOutIds, dOutKeys, dOutCounts, 0u);
}

hipError_t zh_launch_final_interval(const ZhVisit *dVisits, const float *dX, uint32_t d, const float *dQ, const float *dQQ, uint32_t B,
                                    uint32_t k, const uint32_t *dLeafIds, int metric, int mode, uint64_t id_base, ZhApprox ap,
                                    uint64_t *dOutIds, uint64_t *dOutKeys, uint32_t *dOutCounts, uint32_t max_leaf_len, hipStream_t s) {
    if (!B) return hipSuccess;
    if (max_leaf_len == 0) max_leaf_len = 1;
    if (metric == ZH_COSINE)
        launch_final_interval_k<K_COS>(dVisits, dX, d, dQ, dQQ, B, k, dLeafIds, metric, mode, id_base, ap, dOutIds, dOutKeys, dOutCounts, max_leaf_len, s);
    else
        launch_final_interval_k<K_L2>(dVisits, dX, d, dQ, dQQ, B, k, dLeafIds, metric, mode, id_base, ap, dOutIds, dOutKeys, dOutCounts, max_leaf_len, s);
    return hipGetLastError();
}
