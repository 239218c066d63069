// What would the BOTTOM-UP half of "tree_result's walk as a dynamic programme" (tests/test_walk_dp.py) cost on MI355X at the
// reference-default shape (1M rows, max_node_size 5, 15 trees: ~95k blocks of <= 63 nodes, batch 256)?  This microbenchmark runs exactly that
// pass on synthetic blocks and checks it against a host restatement:
//   * a block = a random binary subtree of <= 63 nodes in PRE-order records {plane or -1, left record, right record, leaf length};
//   * R[node][n], n = 0 .. 10, eleven BYTES in three dwords: a leaf's table is min(len, n); an inner node's
//         r[n] = m[n] >= n ? m[n] : bk[n - m[n]]        (m = the main child's table, bk = the backup's; lsh.rs:335-345)
//     in its packed form: idx = n - m[n] (one v_sub_u32 per four demands: R <= n, no borrow), bk[idx] by v_perm_b32, and because R[.][0] = 0
//     and m[n] >= n <=> idx = 0:   r = perm(bk, idx) | (m & perm({0xFF, 0, ...}, idx));
//   * lanes = queries, a wave = one block for 64 queries: the block's shape is shared, so the traversal (reverse pre-order, a stack of
//     tables in LDS) is uniform control flow; only main / backup differ per lane (the query's sign of the node's plane, read from a
//     plane-major bit matrix: one or two dwords per wave and node);
//   * output: the block root's table per (block, query), 12 bytes.
//   hipcc --offload-arch=gfx950 -O3 walk_dp_blocks.hip -o walk_dp_blocks && ./walk_dp_blocks [blocks] [queries]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <vector>

#define TOPK 10
struct Rec { int32_t plane, l, r, len; };  // plane < 0: a leaf of `len` rows; l, r: record indices inside the block
struct Tab { uint32_t w[3]; };             // bytes 0 .. 10 = R[.][0 .. 10]

static Tab leaf_tab(uint32_t len) {
    Tab t{{0, 0, 0}};
    for (uint32_t n = 0; n <= TOPK; n++) t.w[n >> 2] |= (len < n ? len : n) << (8 * (n & 3));
    return t;
}
static inline uint32_t tb(const Tab &t, uint32_t n) { return (t.w[n >> 2] >> (8 * (n & 3))) & 255u; }
static Tab combine_host(const Tab &m, const Tab &bk) {
    Tab r{{0, 0, 0}};
    for (uint32_t n = 0; n <= TOPK; n++) {
        const uint32_t k = tb(m, n), v = k >= n ? k : tb(bk, n - k);
        r.w[n >> 2] |= v << (8 * (n & 3));
    }
    return r;
}

__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// r = perm(bk, idx) | (m & perm(mask, idx)) per group of four demands; group g holds demands 4 g .. 4 g + 3
__device__ __forceinline__ void combine_dev(const uint32_t m[3], const uint32_t bk[3], uint32_t r[3]) {
    const uint32_t n0 = 0x03020100u, n1 = 0x07060504u, n2 = 0x0B0A0908u;
    const uint32_t i0 = n0 - m[0], i1 = n1 - m[1], i2 = n2 - m[2];  // idx bytes (no borrow: R <= n); unused bytes of group 2 stay harmless
    // v_perm_b32 selects bytes 0 .. 7 of {hi, lo}; idx of group 0 is 0 .. 3, of group 1 0 .. 7, of group 2 0 .. 10
    const uint32_t z0 = 0x000000FFu;  // mask table: byte 0 = 0xFF
    r[0] = perm(bk[1], bk[0], i0) | (m[0] & perm(0u, z0, i0));
    r[1] = perm(bk[1], bk[0], i1) | (m[1] & perm(0u, z0, i1));
    // group 2: idx in 4 .. 10 for m <= 4 ... in general 0 .. 10: two perms (bytes 0 .. 7, bytes 8 .. 10) chosen per byte by idx >= 8
    const uint32_t lo8 = perm(bk[1], bk[0], i2 & 0x07070707u), hi8 = perm(0u, bk[2], i2 & 0x03030303u);
    const uint32_t ge8 = ((i2 >> 3) & 0x01010101u) * 0xFFu;  // 0xFF in the bytes whose idx >= 8
    r[2] = ((lo8 & ~ge8) | (hi8 & ge8)) | (m[2] & perm(0u, z0, i2 & 0x07070707u) & ~ge8);
    r[2] &= 0x00FFFFFFu;
}

// grid: (blocks, queries / 64); block of 64 threads = one wave; recs: 64 records per block (pre-order, padded), bitsT[plane][queries / 32]
__global__ __launch_bounds__(64) void block_dp_kernel(const Rec *__restrict__ recs, const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ bitsT,
                                                      uint32_t qwords, uint32_t *__restrict__ out) {
    __shared__ uint32_t stk[8][3][64];  // the stack of tables: depth <= 7 for blocks of <= 63 nodes built as here (<= 6 levels)
    const uint32_t lane = threadIdx.x, blk = blockIdx.x, q = blockIdx.y * 64 + lane;
    const Rec *R = recs + (size_t)blk * 64;
    const uint32_t n = cnt[blk];
    int sp = 0;
    for (int j = (int)n - 1; j >= 0; j--) {  // reverse pre-order: right subtree, left subtree, parent
        const Rec rc = R[j];                 // (wave-uniform: scalar loads)
        uint32_t t[3];
        if (rc.plane < 0) {
            const uint32_t len = (uint32_t)rc.len;
            t[0] = t[1] = t[2] = 0;
#pragma unroll
            for (uint32_t nn = 0; nn <= TOPK; nn++) t[nn >> 2] |= (len < nn ? len : nn) << (8 * (nn & 3));
        } else {
            const uint32_t word = bitsT[(size_t)rc.plane * qwords + (q >> 5)];
            const bool above = (word >> (q & 31)) & 1u;
            // stack top = the LEFT child's table, below it the RIGHT child's
            uint32_t L[3], Rr[3];
            sp--;
#pragma unroll
            for (int w = 0; w < 3; w++) L[w] = stk[sp][w][lane];
            sp--;
#pragma unroll
            for (int w = 0; w < 3; w++) Rr[w] = stk[sp][w][lane];
            uint32_t m[3], bk[3];
#pragma unroll
            for (int w = 0; w < 3; w++) { m[w] = above ? Rr[w] : L[w]; bk[w] = above ? L[w] : Rr[w]; }  // lsh.rs:335-338: above -> right is main
            combine_dev(m, bk, t);
        }
#pragma unroll
        for (int w = 0; w < 3; w++) stk[sp][w][lane] = t[w];
        sp++;
    }
    uint32_t *o = out + ((size_t)blk * gridDim.y * 64 + q) * 3;
    o[0] = stk[0][0][lane]; o[1] = stk[0][1][lane]; o[2] = stk[0][2][lane];
}

int main(int argc, char **argv) {
    const uint32_t NB = argc > 1 ? (uint32_t)atoi(argv[1]) : 95000u, NQ = argc > 2 ? (uint32_t)atoi(argv[2]) : 256u;
    std::mt19937_64 rng(11);
    std::vector<Rec> recs((size_t)NB * 64);
    std::vector<uint32_t> cnt(NB);
    uint32_t planes = 0;
    // random binary subtrees of <= 63 nodes and <= 6 levels, in pre-order; leaf lengths 1 .. 4 (max_node_size 5)
    for (uint32_t b = 0; b < NB; b++) {
        Rec *R = &recs[(size_t)b * 64];
        uint32_t n = 1;
        // grown in pre-order: the node at `slot` becomes inner with probability 0.85 while room and depth allow
        std::function<void(uint32_t, uint32_t)> grow = [&](uint32_t slot, uint32_t depth) {
            const bool inner = depth < 5 && n + 2 <= 63 && (rng() % 100) < 85;
            if (!inner) { R[slot] = Rec{-1, 0, 0, (int32_t)(1 + rng() % 4)}; return; }
            R[slot].plane = (int32_t)planes++;
            const uint32_t l = n++;
            R[slot].l = (int32_t)l;
            grow(l, depth + 1);
            const uint32_t r = n++;
            R[slot].r = (int32_t)r;
            grow(r, depth + 1);
            R[slot].len = 0;
        };
        grow(0, 0);
        cnt[b] = n;
    }
    const uint32_t qwords = NQ / 32;
    std::vector<uint32_t> bits((size_t)planes * qwords);
    for (auto &w : bits) w = (uint32_t)rng();
    // host restatement for a sample of blocks and all queries
    Rec *dR; uint32_t *dC, *dB, *dO;
    hipMalloc(&dR, recs.size() * sizeof(Rec)); hipMalloc(&dC, NB * 4); hipMalloc(&dB, bits.size() * 4); hipMalloc(&dO, (size_t)NB * NQ * 12);
    hipMemcpy(dR, recs.data(), recs.size() * sizeof(Rec), hipMemcpyHostToDevice);
    hipMemcpy(dC, cnt.data(), NB * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, bits.data(), bits.size() * 4, hipMemcpyHostToDevice);
    const dim3 grid(NB, NQ / 64);
    hipLaunchKernelGGL(block_dp_kernel, grid, dim3(64), 0, 0, dR, dC, dB, qwords, dO);
    hipEvent_t a, e;
    hipEventCreate(&a); hipEventCreate(&e);
    hipEventRecord(a);
    const int reps = 5;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(block_dp_kernel, grid, dim3(64), 0, 0, dR, dC, dB, qwords, dO);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, a, e);
    ms /= reps;
    std::vector<uint32_t> out((size_t)NB * NQ * 3);
    hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost);
    uint64_t bad = 0, checked = 0, nodes = 0;
    for (uint32_t b = 0; b < NB; b++) nodes += cnt[b];
    for (uint32_t b = 0; b < NB; b += NB / 500 + 1) {
        const Rec *R = &recs[(size_t)b * 64];
        for (uint32_t q = 0; q < NQ; q += 7) {
            std::function<Tab(uint32_t)> ev = [&](uint32_t j) -> Tab {
                if (R[j].plane < 0) return leaf_tab((uint32_t)R[j].len);
                const Tab L = ev((uint32_t)R[j].l), Rt = ev((uint32_t)R[j].r);
                const bool above = (bits[(size_t)R[j].plane * qwords + (q >> 5)] >> (q & 31)) & 1u;
                return above ? combine_host(Rt, L) : combine_host(L, Rt);
            };
            const Tab t = ev(0);
            const uint32_t *o = &out[((size_t)b * NQ + q) * 3];
            checked++;
            if (o[0] != t.w[0] || o[1] != t.w[1] || o[2] != t.w[2]) bad++;
        }
    }
    printf("blocks %u (%.1f nodes per block, %u planes), queries %u: %.3f ms per pass over all blocks x queries; %llu of %llu sampled root tables differ from the host's\n",
           NB, (double)nodes / NB, planes, NQ, ms, (unsigned long long)bad, (unsigned long long)checked);
    return bad ? 1 : 0;
}
