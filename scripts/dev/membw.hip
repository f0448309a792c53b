// Developer microbenchmark (not part of the product): HBM copy / read / write ceilings of this box for the access shapes the
// partition kernels use.  hipcc --offload-arch=gfx950 -O3 scripts/dev/membw.hip -o /tmp/membw && /tmp/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(1024) void k_copy16(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(&in[i + u * stride]) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { if (NT) __builtin_nontemporal_store(v[u], &out[i + u * stride]); else out[i + u * stride] = v[u]; }
    }
    for (; i < n16; i += stride) out[i] = in[i];
}
// block-contiguous tiles: each 1024-thread block copies chunks of TILE 16-byte words (the partition kernels' shape)
template <int RPT>
__global__ __launch_bounds__(1024) void k_copy_tiles(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n16) {
    const size_t tile = (size_t)1024 * RPT;
    for (size_t t = (size_t)blockIdx.x * tile; t < n16; t += (size_t)gridDim.x * tile) {
        u32x4 v[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) { size_t i = t + (size_t)r * 1024 + threadIdx.x; v[r] = in[i < n16 ? i : n16 - 1]; }
#pragma unroll
        for (int r = 0; r < RPT; ++r) { size_t i = t + (size_t)r * 1024 + threadIdx.x; if (i < n16) out[i] = v[r]; }
    }
}
template <int UNROLL>
__global__ __launch_bounds__(1024) void k_read16(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= in[i + u * stride];
    }
    if (acc.x == 0x12345u) out[0] = acc;
}
__global__ __launch_bounds__(1024) void k_write16(u32x4* __restrict__ out, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) out[i] = u32x4{(unsigned)i, 1, 2, 3};
}
// 8-byte stores in runs of RUN rows scattered over F streams (the first pass's write shape), reads linear 8 B/row as 16-byte loads
template <int RUN>
__global__ __launch_bounds__(1024) void k_scatter_runs(const u32x2* __restrict__ in, u32x2* __restrict__ out, size_t n8, unsigned F, size_t cap) {
    // block b, iteration k: rows [pos, pos + 1024) go to stream (lane group) ...: thread t writes stream s = (t / RUN + k * 7 + b) % F at offset cursor
    const size_t stride = (size_t)gridDim.x * 1024;
    size_t iter = 0;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n8; i += stride, ++iter) {
        const u32x2 v = in[i];
        const unsigned s = (unsigned)((threadIdx.x / RUN + iter * 37 + blockIdx.x * 11) % F);
        const size_t slot = ((size_t)blockIdx.x + iter * gridDim.x) * (1024 / F ? 1 : 1);   // position of this (block, iter) inside the stream
        const size_t o = (size_t)s * cap + (slot * RUN + threadIdx.x % RUN) % cap;
        out[o] = v;
    }
}

int main() {
    const size_t bytes = 8ull << 30;
    void *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes + (64 << 20)));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double gb, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        float best = 1e9f;
        for (int i = 0; i < 5; ++i) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-44s %8.3f ms  %7.1f GB/s\n", name, best, gb / best * 1e3);
        fflush(stdout);
        return 0;
    };
    const size_t n16 = bytes / 16, n8 = bytes / 8;
    const double GB = (double)bytes / 1e9;
    for (int g : {256, 512, 1024, 2048, 4096}) {
        char nm[96];
        snprintf(nm, sizeof nm, "copy16 unroll4 grid %d", g); timeit(nm, 2 * GB, [&] { hipLaunchKernelGGL((k_copy16<4, false>), dim3(g), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16); });
    }
    timeit("copy16 unroll8 grid 512", 2 * GB, [&] { hipLaunchKernelGGL((k_copy16<8, false>), dim3(512), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16); });
    timeit("copy16 unroll8 grid 256", 2 * GB, [&] { hipLaunchKernelGGL((k_copy16<8, false>), dim3(256), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16); });
    timeit("copy16 unroll4 nt grid 1024", 2 * GB, [&] { hipLaunchKernelGGL((k_copy16<4, true>), dim3(1024), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16); });
    timeit("copy tiles 9x16B grid 256 (1 WG/CU)", 2 * GB, [&] { hipLaunchKernelGGL((k_copy_tiles<9>), dim3(256), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16); });
    timeit("copy tiles 9x16B grid 2048", 2 * GB, [&] { hipLaunchKernelGGL((k_copy_tiles<9>), dim3(2048), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16); });
    timeit("copy tiles 2x16B grid 512", 2 * GB, [&] { hipLaunchKernelGGL((k_copy_tiles<2>), dim3(512), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16); });
    timeit("read16 unroll8 grid 1024", GB, [&] { hipLaunchKernelGGL((k_read16<8>), dim3(1024), dim3(1024), 0, 0, (const u32x4*)a, (u32x4*)b, n16); });
    timeit("write16 grid 1024", GB, [&] { hipLaunchKernelGGL(k_write16, dim3(1024), dim3(1024), 0, 0, (u32x4*)b, n16); });
    const unsigned F = 512; const size_t cap = n8 / F;
    timeit("scatter 8B rows, runs of 16 (128 B), F=512", 2 * GB, [&] { hipLaunchKernelGGL((k_scatter_runs<16>), dim3(2048), dim3(1024), 0, 0, (const u32x2*)a, (u32x2*)b, n8, F, cap); });
    timeit("scatter 8B rows, runs of 32 (256 B), F=512", 2 * GB, [&] { hipLaunchKernelGGL((k_scatter_runs<32>), dim3(2048), dim3(1024), 0, 0, (const u32x2*)a, (u32x2*)b, n8, F, cap); });
    timeit("scatter 8B rows, runs of 64 (512 B), F=512", 2 * GB, [&] { hipLaunchKernelGGL((k_scatter_runs<64>), dim3(2048), dim3(1024), 0, 0, (const u32x2*)a, (u32x2*)b, n8, F, cap); });
    timeit("scatter 8B rows, runs of 1024 (8 KB), F=512", 2 * GB, [&] { hipLaunchKernelGGL((k_scatter_runs<1024>), dim3(2048), dim3(1024), 0, 0, (const u32x2*)a, (u32x2*)b, n8, F, cap); });
    return 0;
}
