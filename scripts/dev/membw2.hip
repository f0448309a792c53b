// Developer microbenchmark (not part of the product): the last partition pass's write shape and the join's read shape under two
// layouts of a group's output --
//   A (current): bucket-major slots: line i of bucket b at ((g*F + b)*L + i) lines
//   B          : line-interleaved:   line i of bucket b at ((g*L + i)*F + b) lines  (the write frontier of a group is one narrow band)
// hipcc --offload-arch=gfx950 -O3 scripts/dev/membw2.hip -o /tmp/membw2 && /tmp/membw2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned F = 512, L = 269;   // buckets per group, 128-byte lines per bucket slot (4304 rows)

template <int LAYOUT, bool RAGGED>
__global__ __launch_bounds__(1024) void k_w(const u32x2* __restrict__ in, u32x2* __restrict__ out, unsigned n_groups) {
    const unsigned q = threadIdx.x >> 4, l = threadIdx.x & 15;
    for (unsigned g = blockIdx.x; g < n_groups; g += gridDim.x) {
        for (unsigned k = 0; k < L * 8; ++k) {
            const u32x2 v = in[((size_t)g * L * 8 + k) * 1024 + threadIdx.x];
            const unsigned b = q + 64u * (k & 7u);
            unsigned i = k >> 3;
            if (RAGGED) { i = (i + ((b * 2654435761u) >> 29)) % L; }   // buckets run up to 7 lines ahead of each other
            const size_t line = LAYOUT == 0 ? ((size_t)g * F + b) * L + i : ((size_t)g * L + i) * F + b;
            out[line * 16 + l] = v;
        }
    }
}
template <int LAYOUT>
__global__ __launch_bounds__(1024) void k_r(const u32x2* __restrict__ in, u32x2* __restrict__ out, unsigned n_groups) {
    const unsigned q = threadIdx.x >> 4, l = threadIdx.x & 15;
    u32x2 acc = {0, 0};
    for (unsigned sp = blockIdx.x; sp < n_groups * F; sp += gridDim.x) {
        const unsigned g = sp / F, b = sp % F;
        for (unsigned i0 = 0; i0 < L; i0 += 64) {
            const unsigned i = i0 + q < L ? i0 + q : L - 1;
            const size_t line = LAYOUT == 0 ? ((size_t)g * F + b) * L + i : ((size_t)g * L + i) * F + b;
            acc ^= in[line * 16 + l];
        }
    }
    if (acc.x == 0x12345u) out[0] = acc;
}
int main() {
    const unsigned G = 512;
    const size_t bytes = (size_t)G * F * L * 128;
    void *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double gb, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        float best = 1e9f;
        for (int i = 0; i < 5; ++i) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-60s %8.3f ms  %7.1f GB/s\n", name, best, gb / best * 1e3);
        fflush(stdout);
    };
    const double GB = (double)bytes / 1e9;
    printf("%.2f GB per direction\n", GB);
    timeit("pass-2 shape, layout A (bucket-major), in step", 2 * GB, [&] { hipLaunchKernelGGL((k_w<0, false>), dim3(256), dim3(1024), 0, 0, (const u32x2*)a, (u32x2*)b, G); });
    timeit("pass-2 shape, layout A (bucket-major), ragged", 2 * GB, [&] { hipLaunchKernelGGL((k_w<0, true>), dim3(256), dim3(1024), 0, 0, (const u32x2*)a, (u32x2*)b, G); });
    timeit("pass-2 shape, layout B (line-interleaved), in step", 2 * GB, [&] { hipLaunchKernelGGL((k_w<1, false>), dim3(256), dim3(1024), 0, 0, (const u32x2*)a, (u32x2*)b, G); });
    timeit("pass-2 shape, layout B (line-interleaved), ragged", 2 * GB, [&] { hipLaunchKernelGGL((k_w<1, true>), dim3(256), dim3(1024), 0, 0, (const u32x2*)a, (u32x2*)b, G); });
    timeit("join read shape, layout A, 768 work-groups", GB, [&] { hipLaunchKernelGGL((k_r<0>), dim3(768), dim3(1024), 0, 0, (const u32x2*)b, (u32x2*)a, G); });
    timeit("join read shape, layout B, 768 work-groups", GB, [&] { hipLaunchKernelGGL((k_r<1>), dim3(768), dim3(1024), 0, 0, (const u32x2*)b, (u32x2*)a, G); });
    timeit("join read shape, layout B, 4096 work-groups", GB, [&] { hipLaunchKernelGGL((k_r<1>), dim3(4096), dim3(1024), 0, 0, (const u32x2*)b, (u32x2*)a, G); });
    return 0;
}
