// What can MI355X deliver for RANDOM whole rows gathered into registers, with no arithmetic at all?  The ceiling of the
// leaf-major distance sweep at d = 128 (cfg5: 125M x 128 f32 per shard = 64 GB; a leaf's rows are scattered over the table, so
// every scored row is a random 512-byte read).  The loop is the sweep's access shape and nothing else: a wave owns 64 row ids;
// lanes 0..31 load 16 bytes each of row j, lanes 32..63 of row j + 32 (one 1-KiB wave instruction = two 512-byte rows), RG
// instructions in flight; the data is folded into one register so that the loads cannot be dropped.
//   hipcc --offload-arch=gfx950 -O3 gather512.hip -o gather512 && ./gather512 [table_GB] [t] > gather512.csv
// A table that fits the L2s (./gather512 0.003 t, 0.006 t, 0.012 t: the 3 / 6 / 12 MB of a window's 768-d queries; `t` = plain
// temporal loads, as the table scan's query fetches) measures the L2 -> CU operand ceiling of the scan's access shape instead.
// CSV columns: row_bytes, table_GB, order, rows_in_flight_per_wave, waves_per_block, rows, ms, TB_per_s
// order: random = every id uniform over the table (what a leaf's ids are); sorted = the same ids of one launch sorted ascending
// (what sorting a launch's groups by leaf address could at best approach); run64 = random runs of 64 consecutive rows (rows
// stored in leaf order); seq = the table in order (the streaming ceiling of this loop).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
__constant__ int g_temporal;
__device__ __forceinline__ float4 ldnt(const float4 *p) {  // global_load_dwordx4 ... nt, as the sweep's row loads (or plain)
    if (g_temporal) return *p;
    f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}

template <int ROW16, int RG>  // ROW16 = 16-byte pieces per row (32 = 512 B, 64 = 1 KiB, 192 = 3 KiB)
__global__ __launch_bounds__(256) void gather(const float4 *__restrict__ X, const uint32_t *__restrict__ ids, uint64_t n_ids,
                                              float *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t r0 = wave * 64;
    if (r0 >= n_ids) return;
    const uint32_t my_id = ids[r0 + lane < n_ids ? r0 + lane : n_ids - 1];
    float acc = 0.f;
    if (ROW16 == 8) {  // EIGHT 128-byte rows per wave instruction (a uint8 copy of a d = 128 row): lane's eighth q = lane >> 3
        const uint32_t ql = lane & 7, q = lane >> 3;
        for (uint32_t j0 = 0; j0 < 8; j0 += RG) {
            float4 v[RG];
#pragma unroll
            for (int r = 0; r < RG; r++) {
                const uint32_t id = (uint32_t)__shfl((int)my_id, (int)(8 * (j0 + r) + q));
                v[r] = ldnt(X + (size_t)id * 8 + ql);
            }
#pragma unroll
            for (int r = 0; r < RG; r++) acc += (v[r].x + v[r].y) + (v[r].z + v[r].w);
        }
    } else if (ROW16 == 16) {  // four 256-byte rows per wave instruction (the fp16 copy of a d = 128 row): lane's quarter q = lane >> 4
        const uint32_t ql = lane & 15, q = lane >> 4;
        for (uint32_t j0 = 0; j0 < 16; j0 += RG) {
            float4 v[RG];
#pragma unroll
            for (int r = 0; r < RG; r++) {
                const uint32_t i0 = __builtin_amdgcn_readlane(my_id, j0 + r), i1 = __builtin_amdgcn_readlane(my_id, j0 + r + 16),
                               i2 = __builtin_amdgcn_readlane(my_id, j0 + r + 32), i3 = __builtin_amdgcn_readlane(my_id, j0 + r + 48);
                v[r] = ldnt(X + (size_t)(q == 0 ? i0 : (q == 1 ? i1 : (q == 2 ? i2 : i3))) * 16 + ql);
            }
#pragma unroll
            for (int r = 0; r < RG; r++) acc += (v[r].x + v[r].y) + (v[r].z + v[r].w);
        }
    } else if (ROW16 == 32) {  // two rows per wave instruction
        const uint32_t hl = lane & 31;
        const bool up = lane >= 32;
        for (uint32_t j0 = 0; j0 < 32; j0 += RG) {
            float4 v[RG];
#pragma unroll
            for (int r = 0; r < RG; r++) {
                const uint32_t idl = __builtin_amdgcn_readlane(my_id, j0 + r), idh = __builtin_amdgcn_readlane(my_id, j0 + r + 32);
                v[r] = ldnt(X + (size_t)(up ? idh : idl) * 32 + hl);
            }
#pragma unroll
            for (int r = 0; r < RG; r++) acc += (v[r].x + v[r].y) + (v[r].z + v[r].w);
        }
    } else {            // one row per ROW16 / 64 wave instructions
        constexpr int NV = ROW16 / 64;
        for (uint32_t j0 = 0; j0 < 64; j0 += RG) {
            float4 v[RG][NV];
#pragma unroll
            for (int r = 0; r < RG; r++) {
                const uint32_t id = __builtin_amdgcn_readlane(my_id, j0 + r);
#pragma unroll
                for (int k = 0; k < NV; k++) v[r][k] = ldnt(X + (size_t)id * ROW16 + lane + 64 * k);
            }
#pragma unroll
            for (int r = 0; r < RG; r++)
#pragma unroll
                for (int k = 0; k < NV; k++) acc += (v[r][k].x + v[r][k].y) + (v[r][k].z + v[r][k].w);
        }
    }
    if (acc == 123456.789f) out[wave] = acc;  // (never true for this data; keeps the loads alive)
}

template <int ROW16, int RG>
static void run(const float4 *dX, size_t table_rows, double table_gb, const char *order, const std::vector<uint32_t> &ids, uint32_t *dIds, float *dOut) {
    const uint64_t n = ids.size();
    hipMemcpy(dIds, ids.data(), n * 4, hipMemcpyHostToDevice);
    const uint32_t blocks = (uint32_t)((n / 64 + 3) / 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((gather<ROW16, RG>), dim3(blocks), dim3(256), 0, 0, dX, dIds, n, dOut);  // warm
    hipEventRecord(a);
    const int reps = 5;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((gather<ROW16, RG>), dim3(blocks), dim3(256), 0, 0, dX, dIds, n, dOut);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    const int in_flight = ROW16 == 8 ? 8 * RG : (ROW16 == 16 ? 4 * RG : (ROW16 == 32 ? 2 * RG : RG));
    printf("%d,%.4f,%s,%d,4,%llu,%.3f,%.3f\n", ROW16 * 16, table_gb, order, in_flight, (unsigned long long)n, ms, (double)n * ROW16 * 16 / (ms * 1e-3) / 1e12);
    fflush(stdout);
    hipEventDestroy(a); hipEventDestroy(b);
}

int main(int argc, char **argv) {
    const double table_gb = argc > 1 ? atof(argv[1]) : 64.0;
    const int temporal = argc > 2 && argv[2][0] == 't';
    hipMemcpyToSymbol(HIP_SYMBOL(g_temporal), &temporal, sizeof(int));
    const size_t bytes = (size_t)(table_gb * 1e9) / 3072 * 3072;
    float4 *dX;
    if (hipMalloc(&dX, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc(%zu) failed\n", bytes); return 1; }
    hipMemset(dX, 0x3c, bytes);  // (any non-zero pattern: nothing compresses or short-cuts a zero page)
    const uint64_t n_ids = 24u << 20;  // rows per launch, as one ~12-GB launch of the sweep at d = 128
    uint32_t *dIds; float *dOut;
    hipMalloc(&dIds, n_ids * 4); hipMalloc(&dOut, (n_ids / 64 + 8) * 4);
    printf("row_bytes,table_GB,order,rows_in_flight_per_wave,waves_per_block,rows,ms,TB_per_s\n");
    std::mt19937_64 rng(7);
    const bool only256 = argc > 3 && argv[3][0] == 'h';  // ./gather512 32 n h: 256-byte rows only (a 125M x 128 fp16 table is 32 GB)
    const bool only128 = argc > 3 && argv[3][0] == 'b';  // ./gather512 16 n b: 128-byte rows only (a 125M x 128 uint8 table is 16 GB)
    for (int row16 : {8, 16, 32, 64, 192}) {
        if (only128 != (row16 == 8)) continue;
        if (!only128 && only256 != (row16 == 16)) continue;
        const size_t rows = bytes / ((size_t)row16 * 16);
        const uint64_t n = row16 == 192 ? n_ids / 6 : (row16 == 64 ? n_ids / 2 : n_ids);
        std::vector<uint32_t> rnd(n), srt, run64(n), seq(n);
        for (auto &x : rnd) x = (uint32_t)(rng() % rows);
        srt = rnd;
        std::sort(srt.begin(), srt.end());
        for (uint64_t i = 0; i < n; i += 64) {
            const uint32_t base = (uint32_t)(rng() % (rows - 64));
            for (uint64_t j = 0; j < 64 && i + j < n; j++) run64[i + j] = base + (uint32_t)j;
        }
        for (uint64_t i = 0; i < n; i++) seq[i] = (uint32_t)(i % rows);
        const std::pair<const char *, const std::vector<uint32_t> *> orders[] = {{"random", &rnd}, {"sorted", &srt}, {"run64", &run64}, {"seq", &seq}};
        for (auto &o : orders) {
            if (row16 == 8) {
                run<8, 1>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);
                run<8, 2>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);
                run<8, 4>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);
                run<8, 8>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);
            } else if (row16 == 16) {
                run<16, 2>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);
                run<16, 4>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);
                run<16, 8>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);
                run<16, 16>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);
            } else if (row16 == 32) {
                run<32, 2>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);
                run<32, 4>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);
                run<32, 8>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);   // the sweep's shape: 16 rows in flight
                run<32, 16>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);
            } else if (row16 == 64) {
                run<64, 4>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);
                run<64, 8>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);
                run<64, 16>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);
            } else {
                run<192, 2>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);
                run<192, 4>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);  // the d = 768 sweep's shape
                run<192, 8>(dX, rows, table_gb, o.first, *o.second, dIds, dOut);
            }
        }
    }
    return 0;
}
