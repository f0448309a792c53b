// Developer microbenchmark (not part of the product), follow-up to membw4: the partition kernels' shape (one 1024-thread work-group per CU,
// U 16-byte loads per thread prefetched for the next tile, U stores, barriers) reached 5.2 TB/s at U = 9 but 5.8 at U = 4.  Is it the size of
// the load burst, or all CUs bursting in step?  Variants: staggered start, loads paced between the stores, both; 8-byte stores (the kernels'
// store width); scattered stores are not modelled here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// STAGGER: work-group b sleeps ((b / 8) % 4) quarter-tiles before it starts.  PACE: G loads are issued after every (U / groups) stores.
// ST8: stores are 8-byte (two per loaded word), like the tuple stores of k_part1_narrow.
template <int U, int STAGGER, int GROUPS, bool ST8>
__global__ __launch_bounds__(1024) void k_tiles(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n_tiles, int sleep_units) {
    __shared__ u32x4 s_x[1024];
    u32x4 v[U], nx[U];
    size_t t = blockIdx.x;
    if (t >= n_tiles) return;
    if (STAGGER) for (int i = 0; i < (int)((blockIdx.x / 8) % STAGGER) * sleep_units; ++i) __builtin_amdgcn_s_sleep(127);
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[t * 1024 * U + (size_t)u * 1024 + threadIdx.x];
    for (; t < n_tiles; t += gridDim.x) {
        const size_t tn = t + gridDim.x < n_tiles ? t + gridDim.x : t;
        s_x[threadIdx.x] = v[0]; __syncthreads(); v[0] = s_x[threadIdx.x ^ 1];
        constexpr int PER = (U + GROUPS - 1) / GROUPS;
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
            for (int u = g * PER; u < (g + 1) * PER && u < U; ++u) nx[u] = in[tn * 1024 * U + (size_t)u * 1024 + threadIdx.x];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = g * PER; u < (g + 1) * PER && u < U; ++u) {
                if (ST8) {
                    u32x2* o = reinterpret_cast<u32x2*>(out) + 2 * (t * 1024 * U + (size_t)u * 1024);
                    o[threadIdx.x] = u32x2{v[u].x, v[u].y};
                    o[1024 + threadIdx.x] = u32x2{v[u].z, v[u].w};
                } else out[t * 1024 * U + (size_t)u * 1024 + threadIdx.x] = v[u];
            }
            asm volatile("" ::: "memory");
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = nx[u];
    }
}

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t bytes = 8ull << 30;
    void *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    const size_t n16 = bytes / 16;
    const double GB = (double)bytes / 1e9;
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        float best = 1e9f, sum = 0;
        for (int i = 0; i < 7; ++i) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; sum += ms; }
        printf("%-64s best %8.3f ms %7.1f GB/s   mean %7.1f GB/s\n", name, best, 2 * GB / best * 1e3, 2 * GB / (sum / 7) * 1e3);
        fflush(stdout);
    };
#define T(U, STG, GRP, ST8, GRID, SLP, NAME) timeit(NAME, [&] { hipLaunchKernelGGL((k_tiles<U, STG, GRP, ST8>), dim3(GRID), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16 / (1024 * U), SLP); })
    T(9, 0, 1, false, 256, 0, "x9 baseline");
    T(9, 4, 1, false, 256, 1, "x9 staggered 4 x 1 sleep unit (~4 us)");
    T(9, 4, 1, false, 256, 2, "x9 staggered 4 x 2 sleep units");
    T(9, 8, 1, false, 256, 1, "x9 staggered 8 x 1 sleep unit");
    T(9, 0, 3, false, 256, 0, "x9 paced: 3 groups of 3 loads + 3 stores");
    T(9, 0, 9, false, 256, 0, "x9 paced: load, store, load, store ...");
    T(9, 4, 3, false, 256, 1, "x9 staggered + paced 3 groups");
    T(9, 0, 1, true, 256, 0, "x9 baseline, 8-byte stores");
    T(9, 0, 3, true, 256, 0, "x9 paced 3 groups, 8-byte stores");
    T(9, 4, 3, true, 256, 1, "x9 staggered + paced 3 groups, 8-byte stores");
    T(4, 0, 1, false, 256, 0, "x4 baseline");
    T(4, 0, 1, true, 256, 0, "x4 baseline, 8-byte stores");
    T(4, 4, 1, false, 256, 1, "x4 staggered");
    T(4, 0, 2, false, 256, 0, "x4 paced 2 groups");
    T(6, 0, 1, false, 256, 0, "x6 baseline");
    T(6, 0, 2, false, 256, 0, "x6 paced 2 groups");
    T(6, 0, 3, false, 256, 0, "x6 paced 3 groups");
    T(9, 0, 1, false, 2048, 0, "x9 baseline grid 2048 (8 rounds, like the kernels)");
    T(9, 0, 3, false, 2048, 0, "x9 paced 3 groups grid 2048");
    T(4, 0, 1, false, 512, 0, "x4 baseline grid 512 (2 rounds)");
    return 0;
}
