// zh_build.hip -- gfx950 kernels of the insert path: level-synchronous forest build
// (LSHIndex::build_hyperplane / build_a_tree, /root/reference/src/database/index/lsh.rs:192-267)
// and the counter-based synthetic data generators used by the bench and the parity tests.
//
// Every node's id list is a contiguous segment of perm[tree]; a level splits all of its active
// segments at once: make_planes (one hyperplane per node from two sampled rows) -> classify
// (point_is_above for every member, the hash's sequential fma chain) -> stable partition
// (below | above) by a scan over per-chunk counts.  Leaves end up as contiguous runs of perm,
// which is then used directly as leaf_ids.
#include <cstdlib>

#include "zh_internal.h"

// ------------------------------------------------------------------------------------------------
// synthetic data: Irwin-Hall(4) of four 16-bit fields of splitmix64 -- integer arithmetic plus one
// exact int->float conversion and one f32 multiply, bit-identical to oracle zo_synth_rows.
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline uint64_t zh_splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ int32_t synth_centered(uint64_t seed, uint64_t idx) {
    uint64_t z = zh_splitmix64(seed ^ (idx * 0xD1342543DE82EF95ull));
    uint32_t s = (uint32_t)(z & 0xFFFF) + (uint32_t)((z >> 16) & 0xFFFF) + (uint32_t)((z >> 32) & 0xFFFF) +
                 (uint32_t)(z >> 48);
    return (int32_t)s - 131070;
}
__device__ __forceinline__ float synth_value(uint64_t seed, uint64_t idx, int kind) {
    int32_t t = synth_centered(seed, idx);
    if (kind == 1) {
        int32_t v = 30 + (t * 35) / 37837;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        return (float)v;
    }
    return (float)t * (1.0f / 37837.2f);
}

// kind 2 ("clustered"): 128 consecutive rows share a centre: x = centre + 0.25 * own noise
// kind 3 ("clustered, shuffled"): the same with a row's cluster drawn by a hash of its id (65,536 clusters): cluster-mates are scattered
// over the table, as rows inserted in arbitrary order are -- what the scan's tree-0 row order is for
__device__ __forceinline__ float synth_elem(uint64_t seed, uint64_t row, uint32_t col, uint32_t d, int kind) {
    if (kind == 2 || kind == 3) {
        const uint64_t cluster = kind == 2 ? row / 128 : (zh_splitmix64(seed ^ 0x5C0FF1Eull ^ (row * 0x9E3779B97F4A7C15ull)) & 0xFFFFull);
        float centre = (float)synth_centered(seed ^ 0xC1A57E5ull, cluster * d + col) * (1.0f / 37837.2f);
        float own = (float)synth_centered(seed, row * d + col) * (1.0f / 37837.2f);
        return __builtin_fmaf(0.25f, own, centre);
    }
    return synth_value(seed, row * d + col, kind);
}
__global__ __launch_bounds__(256) void synth_rows_kernel(float *__restrict__ X, uint64_t n_elems, uint32_t d,
                                                          uint64_t seed, uint64_t row0, int kind) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n_elems; i += stride) X[i] = synth_elem(seed, row0 + i / d, (uint32_t)(i % d), d, kind);
}

hipError_t zh_launch_synth_rows(float *dX, uint64_t n, uint32_t d, uint64_t seed, uint64_t row0, int kind,
                                hipStream_t s) {
    uint64_t n_elems = n * d;
    if (!n_elems) return hipSuccess;
    uint64_t blocks = (n_elems + 255) / 256;
    if (blocks > 65536 * 8) blocks = 65536 * 8;
    hipLaunchKernelGGL(synth_rows_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, dX, n_elems, d, seed, row0, kind);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void synth_queries_kernel(float *__restrict__ out, uint64_t seed_rows,
                                                             uint64_t seed_q, uint64_t n_rows, uint64_t b0,
                                                             uint64_t b, uint32_t d, int kind) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b * d) return;
    uint64_t q = b0 + i / d;
    uint32_t c = (uint32_t)(i % d);
    uint64_t r = zh_splitmix64(seed_q ^ (q * 0xA24BAED4963EE407ull)) % n_rows;
    float x = synth_elem(seed_rows, r, c, d, kind);
    float g = (float)synth_centered(seed_q + 0x51ED270B5EB2A002ull, q * d + c) * (1.0f / 37837.2f);
    out[i] = (kind == 1) ? x + (float)((int32_t)(g * 4.0f)) : __builtin_fmaf(kind >= 2 ? 0.1f : 0.3f, g, x);
}

hipError_t zh_launch_synth_queries(float *dOut, uint64_t seed_rows, uint64_t seed_q, uint64_t n_rows, uint64_t b0,
                                   uint64_t b, uint32_t d, int kind, hipStream_t s) {
    uint64_t n = b * d;
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(synth_queries_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, dOut, seed_rows,
                       seed_q, n_rows, b0, b, d, kind);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// build
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iota_perm_kernel(uint32_t *__restrict__ perm, uint64_t N, uint64_t total) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) perm[i] = (uint32_t)(i % N);
}
hipError_t zh_launch_iota_perm(uint32_t *dPerm, uint64_t N, uint32_t T, hipStream_t s) {
    uint64_t total = N * T;
    if (!total) return hipSuccess;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 65536 * 8) blocks = 65536 * 8;
    hipLaunchKernelGGL(iota_perm_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, dPerm, N, total);
    return hipGetLastError();
}

// A launch holds fewer than 2^32 threads: block-indexed kernels over many million nodes / chunks / rows are issued
// in slices of ZH_MAX_BLOCKS blocks, each told the index of its first block.
#define ZH_MAX_BLOCKS (1u << 23)
#define ZH_SLICED(KERNEL, N_BLOCKS, THREADS, SHMEM, STREAM, ...)                                              \
    for (uint64_t b0_ = 0; b0_ < (uint64_t)(N_BLOCKS); b0_ += ZH_MAX_BLOCKS) {                                \
        const uint64_t nb_ = (uint64_t)(N_BLOCKS) - b0_ < ZH_MAX_BLOCKS ? (uint64_t)(N_BLOCKS) - b0_ : ZH_MAX_BLOCKS; \
        hipLaunchKernelGGL(KERNEL, dim3((uint32_t)nb_), dim3(THREADS), SHMEM, STREAM, __VA_ARGS__, (uint32_t)b0_); \
    }

// lsh.rs:222-225: w = b - a ; p = (a + b) / 2 ; c = -(dot(w, p)) as f32 (sequential fma chain).  The vector is
// staged through LDS MP_TILE elements at a time (a fixed 32 KB, any dimension): the block fills a tile, thread 0 runs
// the next MP_TILE links of the chain.
#define MP_TILE 4096
__global__ __launch_bounds__(256) void make_planes_kernel(const float *__restrict__ X, uint32_t d,
                                                           const ZhBuildNode *__restrict__ nodes,
                                                           float *__restrict__ planes, float *__restrict__ consts,
                                                           uint32_t block0) {
    __shared__ float w[MP_TILE], p[MP_TILE];
    const ZhBuildNode nd = nodes[blockIdx.x + block0];
    const float *a = nd.sample_a == ~0ull ? nullptr : X + (size_t)nd.sample_a * d;
    const float *b = nd.sample_b == ~0ull ? nullptr : X + (size_t)nd.sample_b * d;
    float acc = 0.0f;
    for (uint32_t k0 = 0; k0 < d; k0 += MP_TILE) {  // block-uniform
        const uint32_t kc = d - k0 < MP_TILE ? d - k0 : MP_TILE;
        for (uint32_t k = threadIdx.x; k < kc; k += blockDim.x) {
            float av = a ? a[k0 + k] : 0.0f, bv = b ? b[k0 + k] : 0.0f;
            float wk = bv - av;
            w[k] = wk;
            p[k] = (av + bv) / 2.0f;
            planes[(size_t)nd.plane * d + k0 + k] = wk;
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (uint32_t k = 0; k < kc; k++) acc = __builtin_fmaf(w[k], p[k], acc);
        __syncthreads();
    }
    if (threadIdx.x == 0) consts[nd.plane] = -acc;
}
hipError_t zh_launch_make_planes(const float *dX, uint32_t d, const ZhBuildNode *dNodes, uint32_t n_nodes,
                                 float *dPlanes, float *dConsts, hipStream_t s) {
    if (!n_nodes) return hipSuccess;
    ZH_SLICED(make_planes_kernel, n_nodes, 256, 0, s, dX, d, dNodes, dPlanes, dConsts)
    return hipGetLastError();
}

// classify one chunk (<= 256 consecutive members of one node): thread i owns member i and runs the
// hash's sequential chain over its row.  The plane is block-uniform (served by broadcast loads).
__global__ __launch_bounds__(256) void classify_kernel(const float *__restrict__ X, uint32_t d,
                                                        const uint32_t *__restrict__ perm,
                                                        const ZhBuildNode *__restrict__ nodes,
                                                        const ZhBuildChunk *__restrict__ chunks,
                                                        const float *__restrict__ planes,
                                                        const float *__restrict__ consts,
                                                        uint8_t *__restrict__ flags,
                                                        uint32_t *__restrict__ chunk_above, uint32_t block0) {
    __shared__ uint32_t wsum[4];
    const uint32_t bid = blockIdx.x + block0;
    const ZhBuildChunk ch = chunks[bid];
    const uint32_t plane = nodes[ch.node].plane;
    const float *__restrict__ w = planes + (size_t)plane * d;
    const float c = consts[plane];
    const uint32_t tid = threadIdx.x;
    bool above = false;
    if (tid < ch.count) {
        uint32_t id = perm[ch.pos + tid];
        const float *__restrict__ x = X + (size_t)id * d;
        float acc = 0.0f;
        if ((d & 3u) == 0) {
            const float4 *x4 = reinterpret_cast<const float4 *>(x);
            const float4 *w4 = reinterpret_cast<const float4 *>(w);
            for (uint32_t k = 0; k < d / 4; k++) {
                float4 xv = x4[k], wv = w4[k];
                acc = __builtin_fmaf(wv.x, xv.x, acc);
                acc = __builtin_fmaf(wv.y, xv.y, acc);
                acc = __builtin_fmaf(wv.z, xv.z, acc);
                acc = __builtin_fmaf(wv.w, xv.w, acc);
            }
        } else {
            for (uint32_t k = 0; k < d; k++) acc = __builtin_fmaf(w[k], x[k], acc);
        }
        above = ((double)acc + (double)c) >= 0.0;  // lsh.rs:40-42
        flags[ch.pos + tid] = above ? 1 : 0;
    }
    unsigned long long m = __ballot(above);
    if ((tid & 63) == 0) wsum[tid >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (tid == 0) chunk_above[bid] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// The same classification with the rows staged through LDS: the block loads its 256 rows in coalesced 256-B
// pieces (16 lanes per row piece, 4 rows per wave instruction) into a [row][64 + 4] tile -- every fetched line is
// used whole, where the thread-per-row loads above re-fetch each line eight times -- and thread r then runs the
// ordered chain over row r of the tile (ds_read_b128, pitch 68 floats: conflict-free) against the block-uniform
// plane.  Requires d % 4 == 0.
#define CL_KC 64
#define CL_PITCH (CL_KC + 4)
__global__ __launch_bounds__(256) void classify_lds_kernel(const float *__restrict__ X, uint32_t d,
                                                            const uint32_t *__restrict__ perm,
                                                            const ZhBuildNode *__restrict__ nodes,
                                                            const ZhBuildChunk *__restrict__ chunks,
                                                            const float *__restrict__ planes,
                                                            const float *__restrict__ consts,
                                                            uint8_t *__restrict__ flags,
                                                            uint32_t *__restrict__ chunk_above, uint32_t block0) {
    __shared__ __attribute__((aligned(16))) float tile[256 * CL_PITCH];
    __shared__ uint32_t ids_s[256];
    __shared__ uint32_t wsum[4];
    const uint32_t bid = blockIdx.x + block0;
    const ZhBuildChunk ch = chunks[bid];
    const uint32_t plane = nodes[ch.node].plane;
    const float4 *__restrict__ w4 = reinterpret_cast<const float4 *>(planes + (size_t)plane * d);
    const float c = consts[plane];
    const uint32_t tid = threadIdx.x;
    ids_s[tid] = tid < ch.count ? perm[ch.pos + tid] : 0xFFFFFFFFu;
    __syncthreads();
    float acc = 0.0f;
    for (uint32_t k0 = 0; k0 < d; k0 += CL_KC) {
        const uint32_t kc = d - k0 < CL_KC ? d - k0 : CL_KC;  // multiple of 4
#pragma unroll 4
        for (int i = 0; i < 16; i++) {
            uint32_t fidx = tid + 256 * i;
            uint32_t row = fidx >> 4, c4 = fidx & 15;
            uint32_t id = ids_s[row];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (id != 0xFFFFFFFFu && 4 * c4 < kc) v = *reinterpret_cast<const float4 *>(X + (size_t)id * d + k0 + 4 * c4);
            *reinterpret_cast<float4 *>(&tile[row * CL_PITCH + 4 * c4]) = v;
        }
        __syncthreads();
        const float4 *mine = reinterpret_cast<const float4 *>(&tile[tid * CL_PITCH]);
        for (uint32_t j = 0; j < kc / 4; j++) {
            float4 x = mine[j], wv = w4[(k0 >> 2) + j];
            acc = __builtin_fmaf(wv.x, x.x, acc);
            acc = __builtin_fmaf(wv.y, x.y, acc);
            acc = __builtin_fmaf(wv.z, x.z, acc);
            acc = __builtin_fmaf(wv.w, x.w, acc);
        }
        __syncthreads();
    }
    bool above = false;
    if (tid < ch.count) {
        above = ((double)acc + (double)c) >= 0.0;  // lsh.rs:40-42
        flags[ch.pos + tid] = above ? 1 : 0;
    }
    unsigned long long m = __ballot(above);
    if ((tid & 63) == 0) wsum[tid >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (tid == 0) chunk_above[bid] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

hipError_t zh_launch_classify(const float *dX, uint32_t d, const uint32_t *dPerm, const ZhBuildNode *dNodes,
                              const ZhBuildChunk *dChunks, uint32_t n_chunks, const float *dPlanes,
                              const float *dConsts, uint8_t *dFlags, uint32_t *dChunkAbove, hipStream_t s) {
    if (!n_chunks) return hipSuccess;
    static const int variant = [] { const char *e = getenv("ZH_CLASSIFY_VARIANT"); return e ? atoi(e) : 0; }();
    if ((d & 3u) == 0 && d >= 64 && variant != 1) {
        ZH_SLICED(classify_lds_kernel, n_chunks, 256, 0, s, dX, d, dPerm, dNodes, dChunks, dPlanes, dConsts, dFlags, dChunkAbove)
        return hipGetLastError();
    }
    ZH_SLICED(classify_kernel, n_chunks, 256, 0, s, dX, d, dPerm, dNodes, dChunks, dPlanes, dConsts, dFlags, dChunkAbove)
    return hipGetLastError();
}

// ---- exclusive scan of n u32 into n+1 entries (out[n] = total) ----------------------------------
__global__ __launch_bounds__(256) void scan_block_sums_kernel(const uint32_t *__restrict__ in, uint64_t n,
                                                               uint32_t *__restrict__ sums) {
    __shared__ uint32_t sm[256];
    uint64_t base = (uint64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    uint32_t v = 0;
    for (int j = 0; j < 4; j++)
        if (base + j < n) v += in[base + j];
    sm[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = sm[0];
}
// in-place exclusive scan of m values by one block
__global__ __launch_bounds__(1024) void scan_single_kernel(uint32_t *__restrict__ a, uint64_t m) {
    __shared__ uint32_t sm[1024];
    const uint32_t tid = threadIdx.x;
    uint64_t per = (m + 1023) / 1024, lo = tid * per, hi = lo + per < m ? lo + per : m;
    uint32_t s = 0;
    for (uint64_t i = lo; i < hi; i++) s += a[i];
    sm[tid] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t x = tid >= off ? sm[tid - off] : 0;
        __syncthreads();
        sm[tid] += x;
        __syncthreads();
    }
    uint32_t run = sm[tid] - s;
    for (uint64_t i = lo; i < hi; i++) { uint32_t v = a[i]; a[i] = run; run += v; }
    if (tid == 1023) a[m] = sm[1023];
}
__global__ __launch_bounds__(256) void scan_apply_kernel(const uint32_t *__restrict__ in, uint64_t n,
                                                          const uint32_t *__restrict__ block_excl,
                                                          uint32_t *__restrict__ out) {
    __shared__ uint32_t sm[256];
    uint64_t base = (uint64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    uint32_t v[4], s = 0;
    for (int j = 0; j < 4; j++) { v[j] = base + j < n ? in[base + j] : 0; s += v[j]; }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        uint32_t x = threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
        __syncthreads();
        sm[threadIdx.x] += x;
        __syncthreads();
    }
    uint32_t run = block_excl[blockIdx.x] + sm[threadIdx.x] - s;
    for (int j = 0; j < 4; j++)
        if (base + j < n) { out[base + j] = run; run += v[j]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) out[n] = block_excl[gridDim.x];
}
hipError_t zh_launch_scan_u32(const uint32_t *dIn, uint32_t *dOut, uint64_t n, uint32_t *dTmp, hipStream_t s) {
    if (!n) return hipMemsetAsync(dOut, 0, sizeof(uint32_t), s);
    uint32_t nb = (uint32_t)((n + 1023) / 1024);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nb), dim3(256), 0, s, dIn, n, dTmp);
    hipLaunchKernelGGL(scan_single_kernel, dim3(1), dim3(1024), 0, s, dTmp, (uint64_t)nb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(256), 0, s, dIn, n, dTmp, dOut);
    return hipGetLastError();
}

// stable partition of every active segment: below | above; perm_out receives the new order for the
// positions of active segments only (copied back by copyback_kernel)
__global__ __launch_bounds__(256) void scatter_kernel(const uint32_t *__restrict__ perm_in,
                                                       uint32_t *__restrict__ perm_out,
                                                       const ZhBuildNode *__restrict__ nodes,
                                                       const ZhBuildChunk *__restrict__ chunks,
                                                       const uint8_t *__restrict__ flags,
                                                       const uint32_t *__restrict__ chunk_scan,
                                                       uint32_t *__restrict__ node_above, uint32_t block0) {
    __shared__ uint32_t sm[256];
    const uint32_t bid = blockIdx.x + block0;
    const ZhBuildChunk ch = chunks[bid];
    const ZhBuildNode nd = nodes[ch.node];
    const uint32_t tid = threadIdx.x;
    uint32_t f = 0, id = 0;
    if (tid < ch.count) { f = flags[ch.pos + tid]; id = perm_in[ch.pos + tid]; }
    sm[tid] = f;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        uint32_t x = tid >= off ? sm[tid - off] : 0;
        __syncthreads();
        sm[tid] += x;
        __syncthreads();
    }
    const uint32_t local_above = sm[tid] - f;  // exclusive
    const uint32_t base_scan = chunk_scan[nd.first_chunk];
    const uint32_t above_before = chunk_scan[bid] - base_scan;
    const uint32_t n_above = chunk_scan[nd.first_chunk + nd.n_chunks] - base_scan;
    const uint32_t n_below = nd.len - n_above;
    if (tid < ch.count) {
        uint32_t pos_in_node = (uint32_t)(ch.pos - nd.seg_start) + tid;
        uint32_t a_excl = above_before + local_above;
        uint64_t dest = f ? nd.seg_start + n_below + a_excl : nd.seg_start + (pos_in_node - a_excl);
        perm_out[dest] = id;
    }
    if (bid == nd.first_chunk && tid == 0) node_above[ch.node] = n_above;
}
__global__ __launch_bounds__(256) void copyback_kernel(uint32_t *__restrict__ perm, const uint32_t *__restrict__ tmp,
                                                        const ZhBuildChunk *__restrict__ chunks, uint32_t block0) {
    const ZhBuildChunk ch = chunks[blockIdx.x + block0];
    if (threadIdx.x < ch.count) perm[ch.pos + threadIdx.x] = tmp[ch.pos + threadIdx.x];
}
hipError_t zh_launch_scatter(const uint32_t *dPermIn, uint32_t *dPermOut, const ZhBuildNode *dNodes,
                             const ZhBuildChunk *dChunks, uint32_t n_chunks, const uint8_t *dFlags,
                             const uint32_t *dChunkScan, uint32_t *dNodeAbove, hipStream_t s) {
    if (!n_chunks) return hipSuccess;
    ZH_SLICED(scatter_kernel, n_chunks, 256, 0, s, dPermIn, dPermOut, dNodes, dChunks, dFlags, dChunkScan, dNodeAbove)
    ZH_SLICED(copyback_kernel, n_chunks, 256, 0, s, const_cast<uint32_t *>(dPermIn), dPermOut, dChunks)
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// incremental insert: one lane per (new row, tree) descends from its current node to a leaf
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void descend_kernel(ZhForestDev f, const float *__restrict__ X, uint32_t d,
                                                      ZhDescend *__restrict__ items, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ZhDescend it = items[i];
    const float *x = X + (size_t)it.row * d;
    int32_t p;
    while ((p = f.node_plane[it.node]) >= 0) {
        bool above = zh_plane_above(f.planes + (size_t)p * d, f.consts[p], x, d);
        it.node = (uint32_t)(above ? f.node_right[it.node] : f.node_left[it.node]);
        it.path = 2 * it.path + (above ? 1 : 0);
        it.depth++;
    }
    items[i] = it;
}
hipError_t zh_launch_descend(ZhForestDev f, const float *dX, uint32_t d, ZhDescend *dItems, uint32_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(descend_kernel, dim3((n + 63) / 64), dim3(64), 0, s, f, dX, d, dItems, n);
    return hipGetLastError();
}

// the walk's 16-byte node records (ZhForestDev::node_pack)
__global__ __launch_bounds__(256) void pack_nodes_kernel(const int32_t *__restrict__ plane, const int32_t *__restrict__ left,
                                                          const int32_t *__restrict__ right, const float *__restrict__ consts,
                                                          int4 *__restrict__ pack, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t p = plane[i];
    pack[i] = make_int4(p, left[i], right[i], p >= 0 ? __float_as_int(consts[p]) : 0);
}
hipError_t zh_launch_pack_nodes(const int32_t *dPlane, const int32_t *dLeft, const int32_t *dRight, const float *dConsts,
                                int4 *dPack, uint32_t n_nodes, hipStream_t s) {
    if (!n_nodes) return hipSuccess;
    hipLaunchKernelGGL(pack_nodes_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, s, dPlane, dLeft, dRight, dConsts, dPack,
                       n_nodes);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// deduplicate (lsh.rs:270-288 compares the f32 bit patterns): one wave per row, h = sum over words of
// mix(word, position) in wrapping u64 -- order-independent to reduce, position-sensitive by construction
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix_word(uint32_t w, uint32_t pos) {
    uint64_t x = ((uint64_t)pos << 32 | w) + 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void row_hash_kernel(const float *__restrict__ X, uint64_t n, uint32_t d,
                                                        uint64_t *__restrict__ out, uint32_t block0) {
    const uint64_t row = (((uint64_t)blockIdx.x + block0) * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (row >= n) return;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(X + (size_t)row * d);
    uint64_t h = 0;
    for (uint32_t e = lane; e < d; e += 64) h += mix_word(w[e], e);
    for (int m = 1; m < 64; m <<= 1) h += __shfl_xor(h, m);
    if (lane == 0) out[row] = h;
}
hipError_t zh_launch_row_hash(const float *dX, uint64_t n, uint32_t d, uint64_t *dHash, hipStream_t s) {
    if (!n) return hipSuccess;
    uint64_t blocks = (n + 3) / 4;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    ZH_SLICED(row_hash_kernel, blocks, 256, 0, s, dX, n, d, dHash)
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void rows_equal_kernel(const float *__restrict__ X, uint32_t d,
                                                          const uint32_t *__restrict__ pairs, uint32_t n_pairs,
                                                          uint8_t *__restrict__ out) {
    const uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (p >= n_pairs) return;
    const uint32_t *a = reinterpret_cast<const uint32_t *>(X + (size_t)pairs[2 * p] * d);
    const uint32_t *b = reinterpret_cast<const uint32_t *>(X + (size_t)pairs[2 * p + 1] * d);
    bool same = true;
    for (uint32_t e = lane; e < d; e += 64) same = same && (a[e] == b[e]);
    unsigned long long m = __ballot(same);
    if (lane == 0) out[p] = (m == ~0ull) ? 1 : 0;
}
hipError_t zh_launch_rows_equal(const float *dX, uint32_t d, const uint32_t *dPairs, uint32_t n_pairs, uint8_t *dOut,
                                hipStream_t s) {
    if (!n_pairs) return hipSuccess;
    hipLaunchKernelGGL(rows_equal_kernel, dim3((n_pairs + 3) / 4), dim3(256), 0, s, dX, d, dPairs, n_pairs, dOut);
    return hipGetLastError();
}
