// Developer microbenchmark (not part of the product), follow-up to membw3: WHY does the one-load-per-thread copy reach 6.25 TB/s while
// every form with more loads in flight per thread, or a persistent loop, stays at 5.4-5.9?  Variants: wave-contiguous accesses, capped
// occupancy, a persistent loop with one load per thread, loads and stores of different waves decoupled.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// (a) block-strided (membw3): thread t reads words base + u * NT + t
template <int NT, int U>
__global__ __launch_bounds__(NT) void k_strided(const u32x4* __restrict__ in, u32x4* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * NT * U + threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[base + (size_t)u * NT];
#pragma unroll
    for (int u = 0; u < U; ++u) out[base + (size_t)u * NT] = v[u];
}
// (b) wave-contiguous: a wave reads U consecutive 1 KiB segments
template <int NT, int U>
__global__ __launch_bounds__(NT) void k_wavecontig(const u32x4* __restrict__ in, u32x4* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * NT * U + (size_t)(threadIdx.x >> 6) * 64 * U + (threadIdx.x & 63);
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[base + (size_t)u * 64];
#pragma unroll
    for (int u = 0; u < U; ++u) out[base + (size_t)u * 64] = v[u];
}
// (c) occupancy capped through dynamic LDS (blocks per CU = 160 KiB / lds)
template <int NT, int U>
__global__ __launch_bounds__(NT) void k_capped(const u32x4* __restrict__ in, u32x4* __restrict__ out) {
    extern __shared__ unsigned char s_dummy[];
    const size_t base = (size_t)blockIdx.x * NT * U + threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[base + (size_t)u * NT];
    if (v[0].x == 0xdeadbeefu && v[0].y == 0x1234567u) s_dummy[threadIdx.x] = 1;
#pragma unroll
    for (int u = 0; u < U; ++u) out[base + (size_t)u * NT] = v[u];
}
// (d) persistent loop, ONE load per thread per iteration, interleaved chunks; PF: software-pipelined (load k+1 before store k)
template <int NT, bool PF>
__global__ __launch_bounds__(NT) void k_persist1(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n_chunks) {
    size_t c = blockIdx.x;
    if (c >= n_chunks) return;
    u32x4 v = in[c * NT + threadIdx.x];
    for (; c < n_chunks; c += gridDim.x) {
        u32x4 nx = v;
        const size_t cn = c + gridDim.x;
        if (PF && cn < n_chunks) nx = in[cn * NT + threadIdx.x];
        out[c * NT + threadIdx.x] = v;
        if (!PF && cn < n_chunks) nx = in[cn * NT + threadIdx.x];
        v = nx;
    }
}
// (e) the partition kernels' shape with a barrier per tile: 1024 threads, U 16-byte loads, barrier, U stores, next tile prefetched
template <int U, bool BARRIER>
__global__ __launch_bounds__(1024) void k_tile_barrier(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n_tiles) {
    __shared__ u32x4 s_x[1024];
    u32x4 v[U], nx[U];
    size_t t = blockIdx.x;
    if (t >= n_tiles) return;
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[t * 1024 * U + (size_t)u * 1024 + threadIdx.x];
    for (; t < n_tiles; t += gridDim.x) {
        const size_t tn = t + gridDim.x < n_tiles ? t + gridDim.x : t;
        if (BARRIER) { s_x[threadIdx.x] = v[0]; __syncthreads(); v[0] = s_x[threadIdx.x ^ 1]; }
#pragma unroll
        for (int u = 0; u < U; ++u) nx[u] = in[tn * 1024 * U + (size_t)u * 1024 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < U; ++u) out[t * 1024 * U + (size_t)u * 1024 + threadIdx.x] = v[u];
        if (BARRIER) __syncthreads();
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
        hipError_t e = hipGetLastError();
        printf("%-58s best %8.3f ms %7.1f GB/s   mean %7.1f GB/s %s\n", name, best, 2 * GB / best * 1e3, 2 * GB / (sum / 7) * 1e3, e == hipSuccess ? "" : hipGetErrorString(e));
        fflush(stdout);
    };
#define L(K, NT, U, ...) hipLaunchKernelGGL((K<NT, U>), dim3((unsigned)(n16 / ((size_t)NT * U))), dim3(NT), __VA_ARGS__, 0, (const u32x4*)a, (u32x4*)b)
    timeit("strided 256 x1", [&] { L(k_strided, 256, 1, 0); });
    timeit("strided 64 x1", [&] { L(k_strided, 64, 1, 0); });
    timeit("strided 128 x1", [&] { L(k_strided, 128, 1, 0); });
    timeit("strided 512 x1", [&] { L(k_strided, 512, 1, 0); });
    timeit("strided 1024 x1", [&] { L(k_strided, 1024, 1, 0); });
    timeit("strided 256 x4", [&] { L(k_strided, 256, 4, 0); });
    timeit("wave-contiguous 256 x2", [&] { L(k_wavecontig, 256, 2, 0); });
    timeit("wave-contiguous 256 x4", [&] { L(k_wavecontig, 256, 4, 0); });
    timeit("wave-contiguous 256 x8", [&] { L(k_wavecontig, 256, 8, 0); });
    timeit("wave-contiguous 1024 x8", [&] { L(k_wavecontig, 1024, 8, 0); });
    for (int lds : {20 << 10, 40 << 10, 80 << 10}) {
        char nm[96];
        CK(hipFuncSetAttribute((const void*)k_capped<256, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CK(hipFuncSetAttribute((const void*)k_capped<256, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CK(hipFuncSetAttribute((const void*)k_capped<256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        snprintf(nm, sizeof nm, "capped 256 x1, %d blocks/CU", (160 << 10) / lds); timeit(nm, [&] { L(k_capped, 256, 1, lds); });
        snprintf(nm, sizeof nm, "capped 256 x4, %d blocks/CU", (160 << 10) / lds); timeit(nm, [&] { L(k_capped, 256, 4, lds); });
        snprintf(nm, sizeof nm, "capped 256 x8, %d blocks/CU", (160 << 10) / lds); timeit(nm, [&] { L(k_capped, 256, 8, lds); });
    }
    for (int g : {2048, 4096, 8192}) {
        char nm[96];
        snprintf(nm, sizeof nm, "persistent 256 x1 no prefetch grid %d", g);
        timeit(nm, [&] { hipLaunchKernelGGL((k_persist1<256, false>), dim3(g), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n16 / 256); });
        snprintf(nm, sizeof nm, "persistent 256 x1 prefetch grid %d", g);
        timeit(nm, [&] { hipLaunchKernelGGL((k_persist1<256, true>), dim3(g), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n16 / 256); });
    }
    timeit("persistent 1024 x1 prefetch grid 512", [&] { hipLaunchKernelGGL((k_persist1<1024, true>), dim3(512), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16 / 1024); });
    timeit("tiles 1024 x9 prefetch, no barrier, grid 256", [&] { hipLaunchKernelGGL((k_tile_barrier<9, false>), dim3(256), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16 / (1024 * 9)); });
    timeit("tiles 1024 x9 prefetch, barriers, grid 256", [&] { hipLaunchKernelGGL((k_tile_barrier<9, true>), dim3(256), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16 / (1024 * 9)); });
    timeit("tiles 1024 x4 prefetch, barriers, grid 256", [&] { hipLaunchKernelGGL((k_tile_barrier<4, true>), dim3(256), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16 / (1024 * 4)); });
    timeit("tiles 1024 x2 prefetch, barriers, grid 256", [&] { hipLaunchKernelGGL((k_tile_barrier<2, true>), dim3(256), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16 / (1024 * 2)); });
    timeit("tiles 1024 x2 prefetch, barriers, grid 512", [&] { hipLaunchKernelGGL((k_tile_barrier<2, true>), dim3(512), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16 / (1024 * 2)); });
    return 0;
}
