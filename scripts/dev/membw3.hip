// Developer microbenchmark (not part of the product): the HBM copy ceiling of this box in the shape MI355X_MICROARCH.md quotes
// 6.29 TB/s for -- 256-thread blocks, many blocks per CU, 16 bytes per lane, every block one contiguous chunk, no grid-stride loop --
// next to scripts/dev/membw.hip's 1024-thread grid-stride shape (5.0-5.2 TB/s in round 2).
// hipcc --offload-arch=gfx950 -O3 scripts/dev/membw3.hip -o scripts/dev/membw3 && scripts/dev/membw3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// block b copies 16-byte words [b * NT * U, (b + 1) * NT * U): U loads in flight per lane, then U stores
template <int NT, int U, int MODE>   // MODE 0 copy, 1 read, 2 write
__global__ __launch_bounds__(NT) void k_chunk(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n16) {
    const size_t base = (size_t)blockIdx.x * NT * U + threadIdx.x;
    u32x4 v[U];
    if (MODE != 2) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[base + (size_t)u * NT];
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = u32x4{(unsigned)base, (unsigned)u, 2u, 3u};
    }
    if (MODE != 1) {
#pragma unroll
        for (int u = 0; u < U; ++u) out[base + (size_t)u * NT] = v[u];
    } else {
        u32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
        if (acc.x == 0x12345u && acc.y == 0x777u) out[0] = acc;
    }
}
// persistent form: G blocks, block b walks chunks b, b + G, ... (what a partition kernel's work-group does with its tiles)
template <int NT, int U>
__global__ __launch_bounds__(NT) void k_persist(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n_chunks) {
    for (size_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const size_t base = c * NT * U + threadIdx.x;
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[base + (size_t)u * NT];
#pragma unroll
        for (int u = 0; u < U; ++u) out[base + (size_t)u * NT] = v[u];
    }
}
// the same with each block owning ONE contiguous range of the buffer (k_part1_narrow's chunking)
template <int NT, int U>
__global__ __launch_bounds__(NT) void k_persist_range(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n_chunks) {
    const size_t per = (n_chunks + gridDim.x - 1) / gridDim.x;
    const size_t c0 = blockIdx.x * per, c1 = c0 + per < n_chunks ? c0 + per : n_chunks;
    for (size_t c = c0; c < c1; ++c) {
        const size_t base = c * NT * U + threadIdx.x;
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[base + (size_t)u * NT];
#pragma unroll
        for (int u = 0; u < U; ++u) out[base + (size_t)u * NT] = v[u];
    }
}

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t gib : {2, 8}) {
        const size_t bytes = gib << 30;
        void *a, *b;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
        const size_t n16 = bytes / 16;
        const double GB = (double)bytes / 1e9;
        auto timeit = [&](const char* name, double gb, auto launch) {
            for (int i = 0; i < 2; ++i) launch();
            float best = 1e9f, sum = 0;
            for (int i = 0; i < 7; ++i) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; sum += ms; }
            printf("%zu GiB  %-46s best %8.3f ms %7.1f GB/s   mean %7.1f GB/s\n", gib, name, best, gb / best * 1e3, gb / (sum / 7) * 1e3);
            fflush(stdout);
        };
#define CHUNK(NT, U, MODE, NAME, GBS) timeit(NAME, GBS, [&] { hipLaunchKernelGGL((k_chunk<NT, U, MODE>), dim3((unsigned)(n16 / ((size_t)NT * U))), dim3(NT), 0, 0, (const u32x4*)a, (u32x4*)b, n16); })
        CHUNK(256, 1, 0, "copy 256thr x1", 2 * GB);
        CHUNK(256, 2, 0, "copy 256thr x2", 2 * GB);
        CHUNK(256, 4, 0, "copy 256thr x4", 2 * GB);
        CHUNK(256, 8, 0, "copy 256thr x8", 2 * GB);
        CHUNK(512, 4, 0, "copy 512thr x4", 2 * GB);
        CHUNK(1024, 4, 0, "copy 1024thr x4", 2 * GB);
        CHUNK(1024, 9, 0, "copy 1024thr x9", 2 * GB);
        CHUNK(256, 4, 1, "read 256thr x4", GB);
        CHUNK(256, 8, 1, "read 256thr x8", GB);
        CHUNK(256, 4, 2, "write 256thr x4", GB);
        CHUNK(256, 8, 2, "write 256thr x8", GB);
        CHUNK(1024, 4, 2, "write 1024thr x4", GB);
        for (int g : {256, 512, 1024, 2048}) {
            char nm[96];
            snprintf(nm, sizeof nm, "persist 1024thr x9 interleaved grid %d", g);
            timeit(nm, 2 * GB, [&] { hipLaunchKernelGGL((k_persist<1024, 9>), dim3(g), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16 / (1024 * 9)); });
            snprintf(nm, sizeof nm, "persist 1024thr x9 ranges grid %d", g);
            timeit(nm, 2 * GB, [&] { hipLaunchKernelGGL((k_persist_range<1024, 9>), dim3(g), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16 / (1024 * 9)); });
        }
        for (int g : {2048, 4096, 8192}) {
            char nm[96];
            snprintf(nm, sizeof nm, "persist 256thr x8 interleaved grid %d", g);
            timeit(nm, 2 * GB, [&] { hipLaunchKernelGGL((k_persist<256, 8>), dim3(g), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n16 / (256 * 8)); });
            snprintf(nm, sizeof nm, "persist 256thr x8 ranges grid %d", g);
            timeit(nm, 2 * GB, [&] { hipLaunchKernelGGL((k_persist_range<256, 8>), dim3(g), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n16 / (256 * 8)); });
        }
        timeit("hipMemcpyDtoD", 2 * GB, [&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
        timeit("hipMemset", GB, [&] { (void)hipMemsetAsync(b, 3, bytes, 0); });
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
