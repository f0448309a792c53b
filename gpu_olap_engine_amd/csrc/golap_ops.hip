// golap_ops.hip -- GpuFilter, GpuAggregate, arrow-interop widening, synthetic
// generators and the pair verifier, as ahead-of-time gfx950 kernels.
//
// Reference semantics restated (paths relative to the reference root):
//   filter     gpu-olap-core/src/executor.rs:443-476 (mask), :478-507 (compaction/take);
//              gpu-kernel-compiler/src/filter_kernel.rs:3-5 (u8 mask, then prefix-sum + scatter);
//              codegen.rs:11-28 (CmpOp / FilterPredicate) -- the PTX emitter (:36-47, body `ret;`)
//              is replaced by kernels that take `op` as a runtime argument.
//   aggregate  executor.rs:297-355 + :509-527 (f64 accumulators, Int64 `as f64`);
//              aggregate_kernel.rs:17-58 (AggFunc, identities, [agg][group] output);
//              join_kernels.cuh:216-240 (slot = key % num_groups).
//   widening   arrow-interop/src/record_batch_convert.rs:35-110.
#include "golap_device.h"
#include "golap_hidden.h"
#include "golap_internal.h"

#include <algorithm>

namespace golap {

constexpr int OP_THREADS = 256;
constexpr int OP_WAVES = OP_THREADS / 64;

__device__ __forceinline__ bool cmp_i64(int op, long long l, long long r) {
    switch (op) {                       // executor.rs:457-465
        case GOLAP_CMP_EQ: return l == r;
        case GOLAP_CMP_NE: return l != r;
        case GOLAP_CMP_LT: return l < r;
        case GOLAP_CMP_LE: return l <= r;
        case GOLAP_CMP_GT: return l > r;
        case GOLAP_CMP_GE: return l >= r;
        default: return false;          // :464 unknown operator -> no row
    }
}

// ---------------------------------------------------------------------------
// filter mask: 4 rows per thread, 2 x 16-byte loads, one packed 4-byte mask store
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(OP_THREADS) void k_filter_mask4(const longlong2* __restrict__ col2, unsigned long long n4, int op, long long lit,
                                                             uint32_t* __restrict__ mask4) {
    // 4 quads per thread per round, all eight 16-byte loads issued before the first compare (clamped, unconditional)
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long q0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; q0 < n4; q0 += 4 * stride) {
        longlong2 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long q = q0 + u * stride, qc = q < n4 ? q : n4 - 1;
            a[u] = col2[2 * qc];
            b[u] = col2[2 * qc + 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long q = q0 + u * stride;
            const uint32_t m = (cmp_i64(op, a[u].x, lit) ? 1u : 0u) | (cmp_i64(op, a[u].y, lit) ? 1u << 8 : 0u) |
                               (cmp_i64(op, b[u].x, lit) ? 1u << 16 : 0u) | (cmp_i64(op, b[u].y, lit) ? 1u << 24 : 0u);
            if (q < n4) mask4[q] = m;
        }
    }
}

// valid (may be null): byte per row, 0 = NULL (column_buffer.rs:59-60).  A comparison with NULL is not true: the row does not pass.
__global__ __launch_bounds__(OP_THREADS) void k_filter_mask1(const long long* __restrict__ col, unsigned long long first, unsigned long long n, int op,
                                                             long long lit, uint8_t* __restrict__ mask, const uint8_t* __restrict__ valid = nullptr) {
    for (unsigned long long i = first + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        mask[i] = (cmp_i64(op, col[i], lit) && (!valid || valid[i])) ? 1 : 0;
}

// ---------------------------------------------------------------------------
// fused predicate + ordered compaction (selection vector of ascending row ids).
// block b owns rows [b*4096, (b+1)*4096): count -> flat scan -> write with
// wavefront ballot + mbcnt prefix (no atomics, order preserved: executor.rs:479-484).
// ---------------------------------------------------------------------------
constexpr int FC_ROWS = 4096;
constexpr int FC_RPT = FC_ROWS / OP_THREADS;   // 16 rows per thread, interleaved by wave-sized strips

// `bits` (may be null): the pass/fail ballot of every 64-row strip, [block][wave][strip] -- 1 bit per row, so that the
// write pass reads n/8 bytes instead of the 8n-byte column again (1B rows: 4.03 -> see DESIGN.md 5).
__global__ __launch_bounds__(OP_THREADS) void k_filter_count(const long long* __restrict__ col, unsigned long long n, int op, long long lit,
                                                             uint32_t* __restrict__ counts, unsigned long long* __restrict__ bits,
                                                             const uint8_t* __restrict__ valid = nullptr) {
    __shared__ uint32_t s_w[OP_WAVES];
    const unsigned long long base = (unsigned long long)blockIdx.x * FC_ROWS;
    uint32_t c = 0;
    long long v[FC_RPT];
#pragma unroll
    for (int r = 0; r < FC_RPT; ++r) {   // unconditional (clamped) loads first, so that all 16 are in flight together
        const unsigned long long i = base + r * OP_THREADS + threadIdx.x;
        v[r] = col[i < n ? i : n - 1];
    }
    unsigned long long mine = 0;   // lane r of a wave keeps strip r's ballot: 16 words leave as one 128-byte store per wave
#pragma unroll
    for (int r = 0; r < FC_RPT; ++r) {
        const unsigned long long i = base + r * OP_THREADS + threadIdx.x;
        const bool keep = i < n && cmp_i64(op, v[r], lit) && (!valid || valid[i < n ? i : n - 1]);
        c += keep ? 1u : 0u;
        const unsigned long long m = __ballot(keep);
        if ((threadIdx.x & 63) == (uint32_t)r) mine = m;
    }
    if (bits && (threadIdx.x & 63) < FC_RPT)
        bits[((size_t)blockIdx.x * OP_WAVES + (threadIdx.x >> 6)) * FC_RPT + (threadIdx.x & 63)] = mine;   // [block][wave][strip]
    c = wave_reduce_add(c);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// the same for a conjunction of comparisons, each on its own column: one ballot per strip after all columns have been looked at
struct FilterAnd {
    const long long* col[GOLAP_FILTER_AND_MAX];
    const uint8_t* valid[GOLAP_FILTER_AND_MAX];
    long long lit[GOLAP_FILTER_AND_MAX];
    int op[GOLAP_FILTER_AND_MAX];
    uint32_t n;
};
__global__ __launch_bounds__(OP_THREADS) void k_filter_count_and(FilterAnd fa, unsigned long long n, uint32_t* __restrict__ counts,
                                                                 unsigned long long* __restrict__ bits) {
    __shared__ uint32_t s_w[OP_WAVES];
    const unsigned long long base = (unsigned long long)blockIdx.x * FC_ROWS;
    uint32_t keep = 0;   // bit r: row r of this thread passes so far
#pragma unroll
    for (int r = 0; r < FC_RPT; ++r) keep |= (base + r * OP_THREADS + threadIdx.x < n ? 1u : 0u) << r;
    for (uint32_t p = 0; p < fa.n; ++p) {   // (work-group-uniform)
        long long v[FC_RPT];
#pragma unroll
        for (int r = 0; r < FC_RPT; ++r) {   // unconditional (clamped) loads first: all 16 in flight together
            const unsigned long long i = base + r * OP_THREADS + threadIdx.x;
            v[r] = fa.col[p][i < n ? i : n - 1];
        }
#pragma unroll
        for (int r = 0; r < FC_RPT; ++r) {
            const unsigned long long i = base + r * OP_THREADS + threadIdx.x;
            const bool ok = cmp_i64(fa.op[p], v[r], fa.lit[p]) && (!fa.valid[p] || fa.valid[p][i < n ? i : n - 1]);
            if (!ok) keep &= ~(1u << r);
        }
    }
    uint32_t c = 0;
    unsigned long long mine = 0;
#pragma unroll
    for (int r = 0; r < FC_RPT; ++r) {
        const bool k = (keep >> r) & 1u;
        c += k ? 1u : 0u;
        const unsigned long long m = __ballot(k);
        if ((threadIdx.x & 63) == (uint32_t)r) mine = m;
    }
    if ((threadIdx.x & 63) < FC_RPT) bits[((size_t)blockIdx.x * OP_WAVES + (threadIdx.x >> 6)) * FC_RPT + (threadIdx.x & 63)] = mine;
    c = wave_reduce_add(c);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(OP_THREADS) void k_filter_write(const unsigned long long* __restrict__ bits, const uint32_t* __restrict__ scanned,
                                                             uint32_t n_blocks, uint32_t* __restrict__ out_idx,
                                                             unsigned long long* __restrict__ out_count) {
    __shared__ uint32_t s_w[FC_RPT * OP_WAVES];   // strip-major (r, wave) totals -> exclusive offsets
    const unsigned long long base = (unsigned long long)blockIdx.x * FC_ROWS;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bool keep[FC_RPT];
    uint32_t lp[FC_RPT];
#pragma unroll
    for (int r = 0; r < FC_RPT; ++r) {
        const unsigned long long m = bits[((size_t)blockIdx.x * OP_WAVES + wv) * FC_RPT + r];   // one word per wave: a broadcast load
        keep[r] = (m >> lane) & 1ull;
        lp[r] = mbcnt(m);
        if (lane == 0) s_w[r * OP_WAVES + wv] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (wv == 0) {   // 64 (strip, wave) totals, row order = strip-major then wave
        const uint32_t v = s_w[lane];
        const uint32_t inc = wave_inclusive_scan(v);
        s_w[lane] = inc - v;
    }
    __syncthreads();
    const uint32_t gbase = scanned[blockIdx.x];
#pragma unroll
    for (int r = 0; r < FC_RPT; ++r) {
        if (keep[r]) out_idx[gbase + s_w[r * OP_WAVES + wv] + lp[r]] = (uint32_t)(base + r * OP_THREADS + threadIdx.x);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = scanned[n_blocks];
}
static_assert(FC_RPT * OP_WAVES == 64, "strip totals are scanned by one wave");

// take_int64 (executor.rs:496-507) for any 8-byte column
__global__ __launch_bounds__(OP_THREADS) void k_take8(const unsigned long long* __restrict__ col, const uint32_t* __restrict__ idx,
                                                      const unsigned long long* __restrict__ count_ptr, unsigned long long n_max,
                                                      unsigned long long* __restrict__ out) {
    unsigned long long n = count_ptr ? *count_ptr : n_max;
    if (n > n_max) n = n_max;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
    {
        const uint32_t j = idx[i];
        out[i] = j == 0xFFFFFFFFu ? 0ull : col[j];   // NULL row id (outer joins) -> 0, the caller marks the slot invalid
    }
}

// ---------------------------------------------------------------------------
// reductions: per-block partials in a fixed order, then one block folds them in
// index order -> bitwise reproducible run to run (no float atomics).
// partial layout: [block][4] = sum, count, min, max
// ---------------------------------------------------------------------------
struct Acc4 { double sum, cnt, mn, mx; };

__device__ __forceinline__ void block_reduce_acc(Acc4& a, double* s /*[4*OP_WAVES]*/) {
    a.sum = wave_reduce_addf(a.sum);
    a.cnt = wave_reduce_addf(a.cnt);
    a.mn = wave_reduce_minf(a.mn);
    a.mx = wave_reduce_maxf(a.mx);
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { s[wv * 4 + 0] = a.sum; s[wv * 4 + 1] = a.cnt; s[wv * 4 + 2] = a.mn; s[wv * 4 + 3] = a.mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < OP_WAVES; ++w) {
            a.sum += s[w * 4 + 0]; a.cnt += s[w * 4 + 1];
            a.mn = s[w * 4 + 2] < a.mn ? s[w * 4 + 2] : a.mn;
            a.mx = s[w * 4 + 3] > a.mx ? s[w * 4 + 3] : a.mx;
        }
    }
}

#define GOLAP_F64_MAX 1.7976931348623157e308

// key_valid / val_valid (may be null): byte per row, 0 = NULL.  A row whose predicate column is NULL does not pass; a NULL value
// does not take part in SUM / COUNT / MIN / MAX / AVG (SQL aggregates skip NULLs).
template <bool VAL_F64, bool HAS_PRED>
__global__ __launch_bounds__(OP_THREADS) void k_reduce_partials(const long long* __restrict__ key, const void* __restrict__ val, unsigned long long n, int op,
                                                                long long lit, double* __restrict__ partials,
                                                                const uint8_t* __restrict__ key_valid = nullptr, const uint8_t* __restrict__ val_valid = nullptr) {
    __shared__ double s[4 * OP_WAVES];
    Acc4 a{0.0, 0.0, GOLAP_F64_MAX, -GOLAP_F64_MAX};   // executor.rs:308-311 seeds
    const long long* vi = (const long long*)val;
    const double* vf = (const double*)val;
    // 8 rows per thread per round, every load issued before the first use (clamped, unconditional); a thread still folds
    // its rows in increasing row order, so the partials are bit-identical to a one-row-per-round loop
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 8 * stride) {
        long long k[8];
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned long long i = i0 + u * stride, ic = i < n ? i : n - 1;
            if (HAS_PRED) k[u] = key[ic];
            x[u] = VAL_F64 ? vf[ic] : (double)vi[ic];   // executor.rs:521 `as f64`
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned long long i = i0 + u * stride;
            bool keep = i < n;
            if (HAS_PRED) keep = keep && cmp_i64(op, k[u], lit);
            if (keep && key_valid) keep = key_valid[i] != 0;
            if (keep && val_valid) keep = val_valid[i] != 0;
            if (keep) {
                a.sum += x[u]; a.cnt += 1.0;                       // :317-318
                a.mn = x[u] < a.mn ? x[u] : a.mn;                  // :319-321
                a.mx = x[u] > a.mx ? x[u] : a.mx;                  // :322-324
            }
        }
    }
    block_reduce_acc(a, s);
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 4 + 0] = a.sum; partials[blockIdx.x * 4 + 1] = a.cnt;
        partials[blockIdx.x * 4 + 2] = a.mn;  partials[blockIdx.x * 4 + 3] = a.mx;
    }
}

// folds partials[0..n_part) in index order (strided per thread, then a fixed tree)
__global__ __launch_bounds__(OP_THREADS) void k_reduce_final(const double* __restrict__ partials, uint32_t n_part, double* __restrict__ out4,
                                                             uint32_t n_out = 4 /*how many of sum, count, min, max the caller's buffer takes*/) {
    __shared__ double s[4 * OP_WAVES];
    Acc4 a{0.0, 0.0, GOLAP_F64_MAX, -GOLAP_F64_MAX};
    for (uint32_t i = threadIdx.x; i < n_part; i += OP_THREADS) {
        a.sum += partials[i * 4 + 0]; a.cnt += partials[i * 4 + 1];
        a.mn = partials[i * 4 + 2] < a.mn ? partials[i * 4 + 2] : a.mn;
        a.mx = partials[i * 4 + 3] > a.mx ? partials[i * 4 + 3] : a.mx;
    }
    block_reduce_acc(a, s);
    if (threadIdx.x == 0) {
        out4[0] = a.sum;
        if (n_out > 1) out4[1] = a.cnt;
        if (n_out > 2) { out4[2] = a.mn; out4[3] = a.mx; }
    }
}

// out[spec plane * n_groups + g] from stats4[g] = sum,count,min,max (executor.rs:334-346)
struct SpecDev { int32_t func; uint32_t plane; };
__global__ void k_finalize_specs(const double* __restrict__ stats4, uint32_t n_groups, SpecDev sp, double* __restrict__ out) {
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += gridDim.x * blockDim.x) {
        const double sum = stats4[g * 4 + 0], cnt = stats4[g * 4 + 1], mn = stats4[g * 4 + 2], mx = stats4[g * 4 + 3];
        double v;
        switch (sp.func) {
            case GOLAP_AGG_SUM: v = sum; break;
            case GOLAP_AGG_COUNT: v = cnt; break;                 // count as f64, executor.rs:336
            case GOLAP_AGG_MIN: v = mn; break;
            case GOLAP_AGG_MAX: v = mx; break;
            default: v = cnt > 0.0 ? sum / cnt : 0.0; break;      // AVG, :339-345
        }
        out[(size_t)sp.plane * n_groups + g] = v;
    }
}

__global__ void k_counts_from_stats(const double* __restrict__ stats4, uint32_t n_groups, unsigned long long* __restrict__ out_count) {
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += gridDim.x * blockDim.x)
        out_count[g] = (unsigned long long)stats4[g * 4 + 1];
}

// slot = (u64)key % n_groups (join_kernels.cuh:230; negative keys wrap like the reference's unsigned arithmetic).
// Keys below 2^32 -- every non-negative int32 key -- take a 32-bit remainder: the 64-bit one is a ~100-instruction
// software routine and made the grouped kernels VALU-bound.
template <typename KeyT>
__device__ __forceinline__ uint32_t slot_of(KeyT key, uint32_t n_groups) {
    const unsigned long long u = (unsigned long long)key;
    if ((n_groups & (n_groups - 1u)) == 0) return (uint32_t)u & (n_groups - 1u);   // work-group-uniform: a power of two is a mask
    if ((u >> 32) == 0) return (uint32_t)u % n_groups;
    return (uint32_t)(u % n_groups);
}

// ---------------------------------------------------------------------------
// grouped aggregate, LDS pre-aggregation (aggregate_kernel.rs:3-6 "phase 1" /
// "phase 2"): every work-group folds its rows into an LDS table of n_groups
// slots with ds_add_f64 / ds_min_f64 / ds_max_f64, then writes its table as one
// row of a [work-group][group] partial slab; a second kernel folds the slab per
// group in work-group order (reproducible; no global float atomics).
// slot = key % n_groups (join_kernels.cuh:230).
// ---------------------------------------------------------------------------
template <typename KeyT, bool VAL_F64, bool NEED_MM>
__global__ __launch_bounds__(512) void k_group_lds(const KeyT* __restrict__ keys, const void* __restrict__ val, unsigned long long n, uint32_t n_groups,
                                                    unsigned long long rows_per_block, double* __restrict__ slab /*[grid][4][n_groups]*/,
                                                    const uint8_t* __restrict__ vv = nullptr /*validity of the value column, 0 = NULL: skipped*/,
                                                    int vec = 0 /*16-byte loads: see the loop*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    double* s_sum = reinterpret_cast<double*>(s_raw);
    double* s_mn = s_sum + n_groups;   // NEED_MM only: without MIN / MAX an entry is 12 bytes (sum + count) and 12.9k groups fit one table
    double* s_mx = s_mn + n_groups;
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(NEED_MM ? s_mx + n_groups : s_sum + n_groups);
    for (uint32_t g = threadIdx.x; g < n_groups; g += blockDim.x) {
        s_sum[g] = 0.0;
        s_cnt[g] = 0;
        if (NEED_MM) { s_mn[g] = GOLAP_F64_MAX; s_mx[g] = -GOLAP_F64_MAX; }
    }
    __syncthreads();
    const long long* vi = (const long long*)val;
    const double* vf = (const double*)val;
    const unsigned long long lo = (unsigned long long)blockIdx.x * rows_per_block;
    unsigned long long hi = lo + rows_per_block;
    if (hi > n) hi = n;
    // 8 rows per thread per round, loads unconditional on clamped indices so that all of them are in flight together
    // (a conditional load compiles to branch + s_waitcnt vmcnt(0) and serialises them)
    constexpr int GR = 8;
    auto fold = [&](KeyT key, double xv) {
        const uint32_t g = slot_of(key, n_groups);
        atomicAdd(&s_sum[g], xv);
        atomicAdd(&s_cnt[g], 1u);
        if (NEED_MM) {   // MIN / MAX requested by some spec: two more LDS atomics per row
            __hip_atomic_fetch_min(&s_mn[g], xv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(&s_mx[g], xv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    if (vec) {
        // 16-byte loads (round 3): a thread takes 4 consecutive rows -- one load for four int32 keys (two for int64 keys), two for
        // their values -- instead of eight 4- and 8-byte loads per 8 rows.  Needs 16-byte aligned columns, no validity bytes and a
        // block range that starts on a multiple of 4 (the host checks all three); the last < 4 rows of the range go one by one.
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const unsigned long long hi4 = lo + ((hi - lo) & ~3ull);
        for (unsigned long long base = lo; base < hi4; base += (unsigned long long)blockDim.x * GR) {
            u4 kq[GR / 4 * (sizeof(KeyT) / 4)], xq[GR / 4 * 2];
#pragma unroll
            for (int u = 0; u < GR / 4; ++u) {
                unsigned long long i = base + ((unsigned long long)u * blockDim.x + threadIdx.x) * 4;
                if (i >= hi4) i = hi4 - 4;
                if (sizeof(KeyT) == 4) kq[u] = *reinterpret_cast<const u4*>(keys + i);
                else { kq[2 * u] = *reinterpret_cast<const u4*>(keys + i); kq[2 * u + 1] = *reinterpret_cast<const u4*>(keys + i + 2); }
                xq[2 * u] = *reinterpret_cast<const u4*>(vi + i);
                xq[2 * u + 1] = *reinterpret_cast<const u4*>(vi + i + 2);
            }
#pragma unroll
            for (int u = 0; u < GR / 4; ++u) {
                const unsigned long long i = base + ((unsigned long long)u * blockDim.x + threadIdx.x) * 4;
                if (i < hi4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        KeyT key;
                        if (sizeof(KeyT) == 4) key = (KeyT)(int)kq[u][j];
                        else { const u4 q = kq[2 * u + (j >> 1)]; key = (KeyT)(((unsigned long long)q[2 * (j & 1) + 1] << 32) | q[2 * (j & 1)]); }
                        const u4 q = xq[2 * u + (j >> 1)];
                        const unsigned long long bits = ((unsigned long long)q[2 * (j & 1) + 1] << 32) | q[2 * (j & 1)];
                        fold(key, VAL_F64 ? __longlong_as_double((long long)bits) : (double)(long long)bits);
                    }
                }
            }
        }
        if (threadIdx.x < (uint32_t)(hi - hi4)) {
            const unsigned long long i = hi4 + threadIdx.x;
            fold(keys[i], VAL_F64 ? vf[i] : (double)vi[i]);
        }
    } else
    for (unsigned long long base = lo; base < hi; base += (unsigned long long)blockDim.x * GR) {
        KeyT k[GR];
        double x[GR];
        uint8_t ok[GR];
#pragma unroll
        for (int r = 0; r < GR; ++r) {
            unsigned long long i = base + (unsigned long long)r * blockDim.x + threadIdx.x;
            if (i >= hi) i = hi - 1;
            k[r] = keys[i];
            x[r] = VAL_F64 ? vf[i] : (double)vi[i];
            ok[r] = vv ? vv[i] : (uint8_t)1;
        }
#pragma unroll
        for (int r = 0; r < GR; ++r) {
            const unsigned long long i = base + (unsigned long long)r * blockDim.x + threadIdx.x;
            if (i < hi && ok[r]) fold(k[r], x[r]);   // SQL aggregates skip NULL values (COUNT counts the non-NULL ones)
        }
    }
    __syncthreads();
    double* row = slab + (size_t)blockIdx.x * 4 * n_groups;
    for (uint32_t g = threadIdx.x; g < n_groups; g += blockDim.x) {
        row[g] = s_sum[g];
        row[n_groups + g] = (double)s_cnt[g];
        row[2 * n_groups + g] = NEED_MM ? s_mn[g] : GOLAP_F64_MAX;    // (the fold reads all four planes: identities when nobody asked)
        row[3 * n_groups + g] = NEED_MM ? s_mx[g] : -GOLAP_F64_MAX;
    }
}

// Fold of the per-work-group partial tables, in a fixed order (reproducible): a 1024-thread block owns 64 groups; wave w
// folds slabs w, w+16, w+32, ... (independent coalesced loads), then the 16 wave partials are combined in wave order.
// (One thread per group walking all 512 slabs serially took 0.24 ms of the 2.5 ms C4 run.)
constexpr int GF_THREADS = 1024;
constexpr int GF_PARTS = GF_THREADS / 64;
__global__ __launch_bounds__(GF_THREADS) void k_group_fold(const double* __restrict__ slab, uint32_t n_rows_slab, uint32_t n_groups,
                                                           double* __restrict__ stats4) {
    __shared__ double s_part[GF_PARTS][4][64];
    const uint32_t gl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const uint32_t g = blockIdx.x * 64 + gl;
    double sum = 0.0, cnt = 0.0, mn = GOLAP_F64_MAX, mx = -GOLAP_F64_MAX;
    if (g < n_groups) {
#pragma unroll 4
        for (uint32_t b = part; b < n_rows_slab; b += GF_PARTS) {
            const double* row = slab + (size_t)b * 4 * n_groups;
            const double a0 = row[g], a1 = row[n_groups + g], a2 = row[2 * (size_t)n_groups + g], a3 = row[3 * (size_t)n_groups + g];
            sum += a0; cnt += a1;
            mn = a2 < mn ? a2 : mn;
            mx = a3 > mx ? a3 : mx;
        }
    }
    s_part[part][0][gl] = sum; s_part[part][1][gl] = cnt; s_part[part][2][gl] = mn; s_part[part][3][gl] = mx;
    __syncthreads();
    if (part == 0 && g < n_groups) {
        sum = 0.0; cnt = 0.0; mn = GOLAP_F64_MAX; mx = -GOLAP_F64_MAX;
#pragma unroll
        for (int q = 0; q < GF_PARTS; ++q) {
            sum += s_part[q][0][gl]; cnt += s_part[q][1][gl];
            mn = s_part[q][2][gl] < mn ? s_part[q][2][gl] : mn;
            mx = s_part[q][3][gl] > mx ? s_part[q][3][gl] : mx;
        }
        stats4[g * 4 + 0] = sum; stats4[g * 4 + 1] = cnt; stats4[g * 4 + 2] = mn; stats4[g * 4 + 3] = mx;
    }
}

// ---------------------------------------------------------------------------
// grouped aggregate for group counts beyond one LDS table: partition, then LDS-aggregate.
// Global float atomics run at ~24 G adds/s on this chip whatever the schedule (measured: 42 ms per 1B rows), so
// rows are first range-partitioned by slot / AG_GB (slot = key % n_groups, join_kernels.cuh:230) with the same
// histogram -> flat scan -> LDS-staged tile scatter scheme as the join (12 B/row: u32 slot + f64 value), after
// which every bucket holds <= AG_GB consecutive slots and is folded with the LDS pre-aggregation kernel
// (several work-groups per bucket, partial tables folded in a fixed order: reproducible, no global atomics).
// ---------------------------------------------------------------------------
constexpr int AG_THREADS = 512;
constexpr int AG_WAVES = AG_THREADS / 64;
constexpr int AG_RPT = 8;
constexpr int AG_TILE = AG_THREADS * AG_RPT;   // 4096 rows staged in LDS
constexpr int AG_GB = 2048;                    // slots per bucket (one 56 KiB LDS table)
constexpr int AG_GB_SUM = 8192;                // slots per bucket when only sums and counts are kept (12 bytes per slot: one 96 KiB table)
constexpr int AG_MAX_B = 1024;                 // buckets (fan-out of the single partition pass)
constexpr uint32_t AG_NULL_MARK = 0x80000000u;   // partition path: top bit of the slot word = the row's VALUE is NULL (it travels, but is not aggregated)

template <typename KeyT>
__global__ __launch_bounds__(AG_THREADS) void k_agg_hist(const KeyT* __restrict__ keys, unsigned long long n, uint32_t n_groups, uint32_t chunk_rows,
                                                         uint32_t S, uint32_t B, uint32_t* __restrict__ hist /*[B][S]*/, uint32_t gb_shift,
                                                         const uint32_t* __restrict__ run_if = nullptr /*non-null: a no-op unless *run_if != 0*/) {
    __shared__ uint32_t s_cnt[AG_MAX_B];
    if (run_if && *run_if == 0) return;
    const unsigned long long lo = (unsigned long long)blockIdx.x * chunk_rows;
    if (lo >= n) return;
    const unsigned long long hi = lo + chunk_rows < n ? lo + chunk_rows : n;
    for (uint32_t b = threadIdx.x; b < B; b += AG_THREADS) s_cnt[b] = 0;
    __syncthreads();
    for (unsigned long long base = lo; base < hi; base += (unsigned long long)AG_THREADS * 8) {
        KeyT k[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            unsigned long long i = base + (unsigned long long)r * AG_THREADS + threadIdx.x;
            k[r] = keys[i < hi ? i : hi - 1];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const unsigned long long i = base + (unsigned long long)r * AG_THREADS + threadIdx.x;
            if (i < hi) atomicAdd(&s_cnt[slot_of(k[r], n_groups) >> gb_shift], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < B; b += AG_THREADS) hist[(size_t)b * S + blockIdx.x] = s_cnt[b];
}

template <typename KeyT, bool VAL_F64>
__global__ __launch_bounds__(AG_THREADS) void k_agg_scatter(const KeyT* __restrict__ keys, const void* __restrict__ val, unsigned long long n,
                                                            uint32_t n_groups, uint32_t chunk_rows, uint32_t S, uint32_t B,
                                                            const uint32_t* __restrict__ scanned, uint32_t* __restrict__ out_slot,
                                                            double* __restrict__ out_val, const uint8_t* __restrict__ vv = nullptr) {
    __shared__ double s_val[AG_TILE];
    __shared__ uint32_t s_slot[AG_TILE];
    __shared__ uint32_t s_cnt[AG_MAX_B], s_start[AG_MAX_B], s_gpos[AG_MAX_B], s_w[AG_WAVES];
    const unsigned long long lo = (unsigned long long)blockIdx.x * chunk_rows;
    if (lo >= n) return;
    const unsigned long long hi = lo + chunk_rows < n ? lo + chunk_rows : n;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long* vi = (const long long*)val;
    const double* vf = (const double*)val;
    for (uint32_t b = tid; b < B; b += AG_THREADS) s_gpos[b] = scanned[(size_t)b * S + blockIdx.x];
    for (unsigned long long t0 = lo; t0 < hi; t0 += AG_TILE) {
        const uint32_t tn = hi - t0 < (unsigned long long)AG_TILE ? (uint32_t)(hi - t0) : (uint32_t)AG_TILE;
        for (uint32_t b = tid; b < B; b += AG_THREADS) s_cnt[b] = 0;
        uint32_t slot[AG_RPT], br[AG_RPT];
        double x[AG_RPT];
#pragma unroll
        for (int r = 0; r < AG_RPT; ++r) {   // unconditional loads on clamped indices: all in flight together
            const uint32_t i = r * AG_THREADS + tid;
            const unsigned long long gi = t0 + (i < tn ? i : tn - 1);
            slot[r] = slot_of(keys[gi], n_groups) | ((vv && !vv[gi]) ? AG_NULL_MARK : 0u);
            x[r] = VAL_F64 ? vf[gi] : (double)vi[gi];   // executor.rs:521 `as f64`
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < AG_RPT; ++r) {
            const uint32_t i = r * AG_THREADS + tid;
            if (i < tn) {
                const uint32_t b = (slot[r] & ~AG_NULL_MARK) / AG_GB;
                br[r] = (b << 16) | atomicAdd(&s_cnt[b], 1u);
            }
        }
        __syncthreads();
        {
            const uint32_t f0 = 2 * tid, f1 = 2 * tid + 1;
            const uint32_t c0 = f0 < B ? s_cnt[f0] : 0u, c1 = f1 < B ? s_cnt[f1] : 0u;
            const uint32_t inc = wave_inclusive_scan(c0 + c1);
            if (lane == 63) s_w[wv] = inc;
            __syncthreads();
            uint32_t base = 0;
#pragma unroll
            for (int i = 0; i < AG_WAVES; ++i) base += (uint32_t)i < wv ? s_w[i] : 0u;
            const uint32_t e0 = base + inc - (c0 + c1);
            if (f0 < B) s_start[f0] = e0;
            if (f1 < B) s_start[f1] = e0 + c0;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < AG_RPT; ++r) {
            const uint32_t i = r * AG_THREADS + tid;
            if (i < tn) {
                const uint32_t pos = s_start[br[r] >> 16] + (br[r] & 0xFFFFu);
                s_slot[pos] = slot[r];
                s_val[pos] = x[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < AG_RPT; ++r) {
            const uint32_t i = r * AG_THREADS + tid;
            if (i < tn) {
                const uint32_t sl = s_slot[i], b = (sl & ~AG_NULL_MARK) / AG_GB;
                const uint32_t gp = s_gpos[b] + (i - s_start[b]);
                out_slot[gp] = sl;
                out_val[gp] = s_val[i];
            }
        }
        __syncthreads();
        for (uint32_t b = tid; b < B; b += AG_THREADS) s_gpos[b] += s_cnt[b];
    }
}

// Register-resident big tile (same idea as the join's k_scatter_big): 1024 threads x 16 rows = 16384-row tile held in
// registers, ranked once, pushed through an 8192-row LDS window in two rounds.  At 512 buckets a bucket's run per tile is
// 32 rows (256-byte value run, 128-byte slot run) instead of 8 rows with the 4096-row tile above.
constexpr int AGB_THREADS = 1024;
constexpr int AGB_RPT = 16;
constexpr int AGB_TILE = AGB_THREADS * AGB_RPT;
constexpr int AGB_WIN = 8192;
constexpr size_t AGB_LDS = (size_t)AGB_WIN * 12 + (size_t)AG_MAX_B * 12 + (AGB_THREADS / 64) * 4 + 16;

template <typename KeyT, bool VAL_F64>
__global__ __launch_bounds__(AGB_THREADS) void k_agg_scatter_big(const KeyT* __restrict__ keys, const void* __restrict__ val, unsigned long long n,
                                                                 uint32_t n_groups, uint32_t chunk_rows, uint32_t S, uint32_t B,
                                                                 const uint32_t* __restrict__ scanned, uint32_t* __restrict__ out_slot,
                                                                 double* __restrict__ out_val, const uint8_t* __restrict__ vv = nullptr,
                                                                 const uint32_t* __restrict__ run_if = nullptr /*non-null: a no-op unless *run_if != 0*/) {
    if (run_if && *run_if == 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    double* s_val = reinterpret_cast<double*>(s_raw);                          // [AGB_WIN]
    uint32_t* s_slot = reinterpret_cast<uint32_t*>(s_raw + AGB_WIN * 8);       // [AGB_WIN]
    uint32_t* s_cnt = s_slot + AGB_WIN;
    uint32_t* s_start = s_cnt + AG_MAX_B;
    uint32_t* s_gpos = s_start + AG_MAX_B;
    uint32_t* s_w = s_gpos + AG_MAX_B;
    const unsigned long long lo = (unsigned long long)blockIdx.x * chunk_rows;
    if (lo >= n) return;
    const unsigned long long hi = lo + chunk_rows < n ? lo + chunk_rows : n;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long* vi = (const long long*)val;
    const double* vf = (const double*)val;
    for (uint32_t b = tid; b < B; b += AGB_THREADS) s_gpos[b] = scanned[(size_t)b * S + blockIdx.x];
    // The tile's raw rows live in registers from their load to their staging; the NEXT tile's rows are requested as soon as the
    // last window has been staged (the registers are dead from then on), so its HBM reads run under this tile's write-out instead
    // of after it (the SQ counters had this kernel parked at waits 77 % of the time: one work-group per CU, nothing else to overlap).
    KeyT kraw[AGB_RPT];
    double x[AGB_RPT];
    uint32_t nullbits = 0;   // bit r: the value of row r is NULL
    auto load_tile = [&](unsigned long long t0) {
        const uint32_t tn = hi - t0 < (unsigned long long)AGB_TILE ? (uint32_t)(hi - t0) : (uint32_t)AGB_TILE;
        nullbits = 0;
#pragma unroll
        for (int r = 0; r < AGB_RPT; ++r) {   // unconditional loads on clamped indices: all in flight together
            const uint32_t i = r * AGB_THREADS + tid;
            const unsigned long long gi = t0 + (i < tn ? i : tn - 1);
            kraw[r] = keys[gi];
            x[r] = VAL_F64 ? vf[gi] : (double)vi[gi];   // executor.rs:521 `as f64`
            if (vv) nullbits |= (vv[gi] ? 0u : 1u) << r;
        }
    };
    load_tile(lo);
    for (unsigned long long t0 = lo; t0 < hi; t0 += AGB_TILE) {
        const uint32_t tn = hi - t0 < (unsigned long long)AGB_TILE ? (uint32_t)(hi - t0) : (uint32_t)AGB_TILE;
        for (uint32_t b = tid; b < B; b += AGB_THREADS) s_cnt[b] = 0;
        uint32_t slot[AGB_RPT], br[AGB_RPT];
#pragma unroll
        for (int r = 0; r < AGB_RPT; ++r) slot[r] = slot_of(kraw[r], n_groups) | (((nullbits >> r) & 1u) ? AG_NULL_MARK : 0u);
        lds_barrier();
#pragma unroll
        for (int r = 0; r < AGB_RPT; ++r) {
            const uint32_t i = r * AGB_THREADS + tid;
            br[r] = 0xFFFFFFFFu;
            if (i < tn) {
                const uint32_t b = (slot[r] & ~AG_NULL_MARK) / AG_GB;
                br[r] = (b << 16) | atomicAdd(&s_cnt[b], 1u);   // b < 1024, rank < 16384
            }
        }
        lds_barrier();
        {
            const uint32_t c = tid < B ? s_cnt[tid] : 0u;
            const uint32_t inc = wave_inclusive_scan(c);
            if (lane == 63) s_w[wv] = inc;
            lds_barrier();
            uint32_t base = 0;
#pragma unroll
            for (int i = 0; i < AGB_THREADS / 64; ++i) base += (uint32_t)i < wv ? s_w[i] : 0u;
            if (tid < B) s_start[tid] = base + inc - c;
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < AGB_RPT; ++r)
            if (br[r] != 0xFFFFFFFFu) br[r] = s_start[br[r] >> 16] + (br[r] & 0xFFFFu);
        // windows: stage, (last window: request the next tile -- slot / br / x are dead from here on, which hipcc only sees when the
        // last window is not an iteration of the loop), write out
        auto window = [&](uint32_t w0, auto last) {
#pragma unroll
            for (int r = 0; r < AGB_RPT; ++r) {
                const uint32_t pos = br[r] - w0;   // invalid rows and rows of the other window fall outside
                if (pos < (uint32_t)AGB_WIN) {
                    s_slot[pos] = slot[r];
                    s_val[pos] = x[r];
                }
            }
            lds_barrier();
            if constexpr (decltype(last)::value) {
                if (t0 + AGB_TILE < hi) load_tile(t0 + AGB_TILE);   // work-group-uniform
            }
            const uint32_t wn = tn - w0 < (uint32_t)AGB_WIN ? tn - w0 : (uint32_t)AGB_WIN;
#pragma unroll 4
            for (int j = 0; j < AGB_WIN / AGB_THREADS; ++j) {
                const uint32_t i = j * AGB_THREADS + tid;
                if (i < wn) {
                    const uint32_t sl = s_slot[i], b = (sl & ~AG_NULL_MARK) / AG_GB;
                    const uint32_t gp = s_gpos[b] + (w0 + i - s_start[b]);
                    out_slot[gp] = sl;
                    out_val[gp] = s_val[i];
                }
            }
            lds_barrier();
        };
        uint32_t w0 = 0;
        for (; w0 + AGB_WIN < tn; w0 += AGB_WIN) window(w0, std::false_type{});
        window(w0, std::true_type{});
        for (uint32_t b = tid; b < B; b += AGB_THREADS) s_gpos[b] += s_cnt[b];
    }
}

// ---------------------------------------------------------------------------
// Partition scatter with line-aligned write combining (fan-outs <= 128: SUM / COUNT / AVG over up to 2^20 groups, whose 12-byte
// LDS table entries allow 8192 slots per bucket).  Round 2's profile of k_agg_scatter_big: 3.1 TB/s, 77 % of the wave cycles
// parked at waits, 1.14 x write amplification -- every (tile, bucket) run ends inside a 128-byte line that the same work-group
// completes one tile later, after the L2 has evicted it.  The histogram gives every chunk (= work-group) a PRIVATE contiguous range
// in every bucket, so here a bucket's unfinished line stays in LDS: per tile a bucket emits rows only up to the last 16-row boundary
// of its cursor (a whole 128-byte value line, a 64-byte half line of slot words) and carries the < 16 remaining rows into the next
// tile, where they are staged in front of the bucket's new rows -- each line leaves the chip once, whole.  Only the first and the
// last line of a (chunk, bucket) range can be partial.  Same register-resident 16384-row tile as k_agg_scatter_big; the window
// (9216 rows) holds the carried rows too, so a tile still takes two rounds.
// ---------------------------------------------------------------------------
constexpr int AGW_RPT = 16;
constexpr int AGW_MAX_B = 128;
constexpr int AGW_CL = 16;       // rows per carried line
// two shapes: one 1024-thread work-group per CU (16384-row tile, two window rounds: the default), or two 512-thread work-groups per
// CU (8192-row tile, three rounds of a 3584-row window; slower, kept as an A/B reference -- see the launch site)
static constexpr size_t agw_lds(int win) { return (size_t)win * 12 + (size_t)AGW_MAX_B * AGW_CL * 12 + (size_t)AGW_MAX_B * 5 * 4 + 20 * 4 + 16; }

template <typename KeyT, bool VAL_F64, int AGW_THREADS, int AGW_WIN>
__global__ __launch_bounds__(AGW_THREADS) void k_agg_scatter_wc(const KeyT* __restrict__ keys, const void* __restrict__ val, unsigned long long n,
                                                                uint32_t n_groups, uint32_t gb_shift, uint32_t chunk_rows, uint32_t S, uint32_t B,
                                                                const uint32_t* __restrict__ scanned, uint32_t* __restrict__ out_slot,
                                                                double* __restrict__ out_val, const uint8_t* __restrict__ vv = nullptr,
                                                                const uint32_t* __restrict__ run_if = nullptr /*non-null: a no-op unless *run_if != 0*/) {
    constexpr int AGW_TILE = AGW_THREADS * AGW_RPT;
    static_assert(AGW_WIN % AGW_THREADS == 0, "window rows per thread");
    static_assert(3 * AGW_WIN >= AGW_TILE + AGW_MAX_B * (AGW_CL - 1), "a tile and its carried rows pass the window in at most three rounds");
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    double* s_val = reinterpret_cast<double*>(s_raw);                     // [WIN]      window: values
    double* c_val = s_val + AGW_WIN;                                      // [B][CL]    carried rows: values
    uint32_t* s_slot = reinterpret_cast<uint32_t*>(c_val + AGW_MAX_B * AGW_CL);   // [WIN] window: slot words
    uint32_t* c_slot = s_slot + AGW_WIN;                                  // [B][CL]
    uint32_t* s_cnt = c_slot + AGW_MAX_B * AGW_CL;                        // [B] new rows of the bucket in this tile
    uint32_t* s_start = s_cnt + AGW_MAX_B;                                // [B] tile-local position of the bucket's first (carried) row
    uint32_t* s_gpos = s_start + AGW_MAX_B;                               // [B] output row of the bucket's first staged row
    uint32_t* s_carry = s_gpos + AGW_MAX_B;                               // [B] rows carried into this tile
    uint32_t* s_emit = s_carry + AGW_MAX_B;                               // [B] rows that leave in this tile (the rest is carried on)
    uint32_t* s_w = s_emit + AGW_MAX_B;                                   // [<= 16] wave totals, [16] rows staged in this tile
    if (run_if && *run_if == 0) return;
    const unsigned long long lo = (unsigned long long)blockIdx.x * chunk_rows;
    if (lo >= n) return;
    const unsigned long long hi = lo + chunk_rows < n ? lo + chunk_rows : n;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long* vi = (const long long*)val;
    const double* vf = (const double*)val;
    if (tid < B) { s_gpos[tid] = scanned[(size_t)tid * S + blockIdx.x]; s_carry[tid] = 0; s_cnt[tid] = 0; }
    KeyT kraw[AGW_RPT];
    double x[AGW_RPT];
    uint32_t nullbits = 0;   // bit r: the value of row r is NULL
    auto load_tile = [&](unsigned long long t0) {
        const uint32_t tn = hi - t0 < (unsigned long long)AGW_TILE ? (uint32_t)(hi - t0) : (uint32_t)AGW_TILE;
        nullbits = 0;
#pragma unroll
        for (int r = 0; r < AGW_RPT; ++r) {   // unconditional loads on clamped indices: all in flight together
            const uint32_t i = r * AGW_THREADS + tid;
            const unsigned long long gi = t0 + (i < tn ? i : tn - 1);
            kraw[r] = keys[gi];
            x[r] = VAL_F64 ? vf[gi] : (double)vi[gi];   // executor.rs:521 `as f64`
            if (vv) nullbits |= (vv[gi] ? 0u : 1u) << r;
        }
    };
    load_tile(lo);
    for (unsigned long long t0 = lo; t0 < hi; t0 += AGW_TILE) {
        const uint32_t tn = hi - t0 < (unsigned long long)AGW_TILE ? (uint32_t)(hi - t0) : (uint32_t)AGW_TILE;
        const bool last_tile = t0 + AGW_TILE >= hi;
        uint32_t slot[AGW_RPT], br[AGW_RPT];
#pragma unroll
        for (int r = 0; r < AGW_RPT; ++r) slot[r] = slot_of(kraw[r], n_groups) | (((nullbits >> r) & 1u) ? AG_NULL_MARK : 0u);
        lds_barrier();   // s_cnt zeroed, cursors and carry counts of the previous tile updated
#pragma unroll
        for (int r = 0; r < AGW_RPT; ++r) {
            const uint32_t i = r * AGW_THREADS + tid;
            br[r] = 0xFFFFFFFFu;
            if (i < tn) {
                const uint32_t b = (slot[r] & ~AG_NULL_MARK) >> gb_shift;
                br[r] = (b << 16) | atomicAdd(&s_cnt[b], 1u);   // b < 128, rank < 16384
            }
        }
        lds_barrier();
        {
            const uint32_t tot = tid < B ? s_cnt[tid] + s_carry[tid] : 0u;   // rows of the bucket staged in this tile: carried + new
            const uint32_t inc = wave_inclusive_scan(tot);
            if (lane == 63) s_w[wv] = inc;
            lds_barrier();
            uint32_t base = 0;
#pragma unroll
            for (int i = 0; i < AGW_THREADS / 64; ++i) base += (uint32_t)i < wv ? s_w[i] : 0u;
            if (tid < B) {
                s_start[tid] = base + inc - tot;
                const uint32_t g = s_gpos[tid];
                uint32_t e = tot;   // the chunk's last tile: everything leaves
                if (!last_tile) {
                    const uint32_t end = (g + tot) & ~(uint32_t)(AGW_CL - 1);
                    e = end > g ? end - g : 0u;
                }
                s_emit[tid] = e;
                if (tid == B - 1) s_w[16] = base + inc;
            }
        }
        lds_barrier();
        const uint32_t total = s_w[16];
#pragma unroll
        for (int r = 0; r < AGW_RPT; ++r)
            if (br[r] != 0xFFFFFFFFu) { const uint32_t b = br[r] >> 16; br[r] = s_start[b] + s_carry[b] + (br[r] & 0xFFFFu); }
        // windows: stage, (last window: request the next tile -- slot / br / x are dead from there on, which hipcc only sees when the
        // last window is not an iteration of the loop: as one loop the kernel spilled 111 registers), write out
        auto window = [&](uint32_t w0, auto last) {
#pragma unroll
            for (int r = 0; r < AGW_RPT; ++r) {
                const uint32_t pos = br[r] - w0;   // invalid rows and rows of the other window fall outside
                if (pos < (uint32_t)AGW_WIN) {
                    s_slot[pos] = slot[r];
                    s_val[pos] = x[r];
                }
            }
            // the carried rows go in front of their bucket's new rows (a bucket's new carry is written by the write-out below, in
            // the round that holds its LAST rows -- never before the round that stages its first ones)
            for (uint32_t idx = tid; idx < B * (uint32_t)AGW_CL; idx += AGW_THREADS) {
                const uint32_t b = idx / AGW_CL, j = idx % AGW_CL;
                if (j < s_carry[b]) {
                    const uint32_t pos = s_start[b] + j - w0;
                    if (pos < (uint32_t)AGW_WIN) { s_slot[pos] = c_slot[idx]; s_val[pos] = c_val[idx]; }
                }
            }
            lds_barrier();
            if constexpr (decltype(last)::value) {
                if (!last_tile) load_tile(t0 + AGW_TILE);   // work-group-uniform: the registers are free, the reads run under the write-out
            }
            const uint32_t wn = total - w0 < (uint32_t)AGW_WIN ? total - w0 : (uint32_t)AGW_WIN;
#pragma unroll 3
            for (int j = 0; j < AGW_WIN / AGW_THREADS; ++j) {
                const uint32_t i = j * AGW_THREADS + tid;
                if (i < wn) {
                    const uint32_t sl = s_slot[i], b = (sl & ~AG_NULL_MARK) >> gb_shift;
                    const double v = s_val[i];
                    const uint32_t k = w0 + i - s_start[b], e = s_emit[b];
                    if (k < e) {
                        const uint32_t gp = s_gpos[b] + k;
                        out_slot[gp] = sl;
                        out_val[gp] = v;
                    } else {
                        c_slot[b * AGW_CL + (k - e)] = sl;
                        c_val[b * AGW_CL + (k - e)] = v;
                    }
                }
            }
            lds_barrier();
        };
        uint32_t w0 = 0;
        for (; w0 + AGW_WIN < total; w0 += AGW_WIN) window(w0, std::false_type{});
        window(w0, std::true_type{});
        if (tid < B) {
            const uint32_t tot = s_cnt[tid] + s_carry[tid], e = s_emit[tid];
            s_gpos[tid] += e;
            s_carry[tid] = tot - e;
            s_cnt[tid] = 0;
        }
    }
}

// ---------------------------------------------------------------------------
// Partition scatter WITHOUT a histogram pass (round 3): the shape of the join's first pass (k_part1_narrow, WIDE form) on
// (slot32, value64) rows.  Bucket b owns AGR_NSUB sub-ranges of `cap` rows (one per XCD: work-groups b and b + 8 share an L2, so the
// line at the boundary of two neighbouring runs is completed inside one L2); a tile reserves its run in every bucket with one
// returning atomic per (tile, bucket) on cursor[b * AGR_NSUB + blockIdx.x % AGR_NSUB].  One 1024-thread work-group per CU, a
// register-resident 12288-row tile that passes the LDS window in ONE round (12 bytes per row), the next tile's loads requested as soon
// as the tile is staged, so that HBM reads run under the write-out; ASM: hidden 16-byte loads, hidden stores issued at wave-uniform
// control flow and a counted vmcnt (golap_hidden.h) -- the compiler-visible form waits vmcnt(0), i.e. for every store of the
// write-out, before it touches a prefetched key.
// Against histogram + k_agg_scatter_wc: no second read of the key column (4 GB per 1B int32 keys), no scan, 3 barriers per tile
// instead of 8.  Price: the ranges are provisioned (cap = mean * (1 + 1/8) + a tile), not exact -- a slot range that receives more
// than that (skewed groups) raises flags[0]; every work-group that starts afterwards returns at once, and the histogram path,
// enqueued behind with run_if = flags, redoes the pass exactly.  k_agg_sample raises the same word beforehand when a sample of the
// key column already shows the skew, so that such inputs do not pay for the attempt.  Row order inside a bucket is arrival order:
// as with every LDS-atomic fold here, only sums of exactly representable terms are order-independent.
// Rows handled: a multiple of 4 (a thread takes 4 consecutive rows: one 16-byte load of int32 keys, two of int64 keys, two of
// values); the column's last n % 4 rows are appended by k_agg_scatter_tail.
// ---------------------------------------------------------------------------
constexpr int AGR_THREADS = 1024;
constexpr int AGR_RPT = 12;
constexpr int AGR_TILE = AGR_THREADS * AGR_RPT;   // 12288 rows = the LDS window
constexpr int AGR_NSUB = 8;
constexpr int AGR_MAX_B = 512;                    // buckets: a tile brings >= 24 rows per bucket (beyond that the histogram path's line-aligned runs win)
static constexpr size_t agr_lds(uint32_t B) { return (size_t)AGR_TILE * 12 + (size_t)B * 12 + (AGR_THREADS / 64 + 2) * 4 + 16; }

template <typename KeyT, bool VAL_F64, bool ASM>
__global__ __launch_bounds__(AGR_THREADS) void k_agg_scatter_rsv(const KeyT* __restrict__ keys, const void* __restrict__ val, unsigned long long n4 /*rows, a multiple of 4*/,
                                                                 uint32_t n_groups, uint32_t gb_shift, uint32_t chunk_rows /*a multiple of 4*/, uint32_t B,
                                                                 uint32_t* __restrict__ cursor /*[B * AGR_NSUB], zeroed*/, uint32_t cap,
                                                                 uint32_t* __restrict__ flags /*[0]: a range overflowed (or will)*/,
                                                                 uint32_t* __restrict__ out_slot, double* __restrict__ out_val) {
    constexpr int NT = AGR_THREADS, NW = NT / 64, RPT = AGR_RPT, TILE = AGR_TILE, UNITS = RPT / 4, KQ = sizeof(KeyT) / 4;
    constexpr int STORES = 2 * RPT;   // store instructions per wave behind the next tile's loads (ASM: all of them are issued)
    static_assert(STORES <= 60, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    u32x2* s_val = reinterpret_cast<u32x2*>(s_raw);                          // [TILE] value bits
    uint32_t* s_slot = reinterpret_cast<uint32_t*>(s_raw + (size_t)TILE * 8);   // [TILE] slot words
    uint32_t* s_cnt = s_slot + TILE;                                          // [B] rows of the bucket in this tile
    uint32_t* s_start = s_cnt + B;                                            // [B] tile-local position of the bucket's first row
    uint32_t* s_delta = s_start + B;                                          // [B] output row - tile-local position
    uint32_t* s_w = s_delta + B;                                              // [NW] wave totals
    if (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;   // work-group-uniform: the histogram path does the work
    const unsigned long long lo = (unsigned long long)blockIdx.x * chunk_rows;
    if (lo >= n4) return;
    const uint32_t len = (uint32_t)((lo + chunk_rows < n4 ? lo + chunk_rows : n4) - lo);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t sub = blockIdx.x % AGR_NSUB;
    const unsigned long long* vals = (const unsigned long long*)val;
    if (tid < B) s_cnt[tid] = 0;

    u32x4 kq[UNITS * KQ], xq[UNITS * 2];
    // rows of unit u: 4 * (u * NT + tid) + {0, 1, 2, 3}; units past the tile's end re-read its last one (unconditional loads)
    auto load_tile = [&](uint32_t t0) {
        const uint32_t tn = len - t0 < (uint32_t)TILE ? len - t0 : (uint32_t)TILE;   // a multiple of 4, >= 4
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            uint32_t ub = 4u * ((uint32_t)u * NT + tid);
            if (ub >= tn) ub = tn - 4u;
            const KeyT* kp = keys + lo + t0 + ub;
            const unsigned long long* xp = vals + lo + t0 + ub;
            if (ASM) {
                ld16_hidden(kq[u * KQ], kp);
                if (KQ == 2) ld16_hidden(kq[u * KQ + KQ - 1], kp + 2);
                ld16_hidden(xq[2 * u], xp);
                ld16_hidden(xq[2 * u + 1], xp + 2);
            } else {
                kq[u * KQ] = *reinterpret_cast<const u32x4*>(kp);
                if (KQ == 2) kq[u * KQ + KQ - 1] = *reinterpret_cast<const u32x4*>(kp + 2);
                xq[2 * u] = *reinterpret_cast<const u32x4*>(xp);
                xq[2 * u + 1] = *reinterpret_cast<const u32x4*>(xp + 2);
            }
        }
    };
    load_tile(0);
    for (uint32_t t0 = 0; t0 < len; t0 += TILE) {
        const uint32_t tn = len - t0 < (uint32_t)TILE ? len - t0 : (uint32_t)TILE;
        if (t0 == 0) lds_barrier();   // s_cnt zeroed (later tiles: re-zeroed by the scan below, two barriers ago)
        if (ASM) {
            // the tile was requested before the STORES stores of the previous write-out: it has landed once at most STORES operations
            // are outstanding (first tile: nothing is younger)
            if (t0 == 0) wait_vmcnt<0>(); else wait_vmcnt<STORES>();
#pragma unroll
            for (int q = 0; q < UNITS * KQ; ++q) pin(kq[q]);
#pragma unroll
            for (int q = 0; q < UNITS * 2; ++q) pin(xq[q]);
        }
        uint32_t slot[RPT], br[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int u = r >> 2, j = r & 3;
            KeyT key;
            if (KQ == 1) key = (KeyT)(int)kq[u][j];
            else key = (KeyT)(((unsigned long long)kq[u * KQ + (j >> 1)][2 * (j & 1) + 1] << 32) | kq[u * KQ + (j >> 1)][2 * (j & 1)]);
            slot[r] = slot_of(key, n_groups);
            br[r] = 0xFFFFFFFFu;
            if (4u * ((uint32_t)u * NT + tid) < tn) {
                const uint32_t b = slot[r] >> gb_shift;
                br[r] = (b << 16) | atomicAdd(&s_cnt[b], 1u);   // b < 512, rank < 12288
            }
        }
        lds_barrier();
        uint32_t c = 0, rsv = 0, start;
        if (tid < B) {
            c = s_cnt[tid];
            s_cnt[tid] = 0;   // for the next tile
            if (c) {
                if (ASM) atomic_add_rtn_hidden(rsv, cursor + tid * AGR_NSUB + sub, c);
                else rsv = atomicAdd(&cursor[tid * AGR_NSUB + sub], c);
            }
        }
        {
            const uint32_t inc = wave_inclusive_scan(c);
            if (lane == 63) s_w[wv] = inc;
            lds_barrier();
            uint32_t base = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) base += (uint32_t)i < wv ? s_w[i] : 0u;
            start = base + inc - c;
            if (tid < B) s_start[tid] = start;
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            if (br[r] != 0xFFFFFFFFu) {
                const int u = r >> 2, j = r & 3;
                const uint32_t pos = s_start[br[r] >> 16] + (br[r] & 0xFFFFu);
                const u32x4 q = xq[2 * u + (j >> 1)];
                u32x2 bits{q[2 * (j & 1)], q[2 * (j & 1) + 1]};
                if (!VAL_F64) {   // executor.rs:521 `as f64`
                    const double d = (double)(long long)(((unsigned long long)bits.y << 32) | bits.x);
                    const unsigned long long db = (unsigned long long)__double_as_longlong(d);
                    bits = u32x2{(uint32_t)db, (uint32_t)(db >> 32)};
                }
                s_val[pos] = bits;
                s_slot[pos] = slot[r];
            }
        }
        if (ASM) {   // nothing hidden is younger than the reservation here (the next tile's loads follow below)
            wait_vmcnt<0>();
            asm volatile("" : "+v"(rsv));
        }
        if (tid < B) {   // the reservations have had the scan and the staging to come back
            s_delta[tid] = (tid * AGR_NSUB + sub) * cap + rsv - start;
            if (rsv + c > cap) atomicOr(flags, 1u);   // (rsv + c cannot wrap: both < 2^31)
        }
        if (t0 + TILE < len) load_tile(t0 + TILE);   // key and value registers are dead: the next tile streams in during the write-out
        lds_barrier();
#pragma unroll 4
        for (int j = 0; j < RPT; ++j) {   // ASM: every store instruction is issued, whatever the data
            const uint32_t i = (uint32_t)j * NT + tid;
            const uint32_t sl = s_slot[i];
            const u32x2 v = s_val[i];
            uint32_t b = sl >> gb_shift;
            b = b < B ? b : B - 1u;   // (rows beyond tn hold stale words)
            const uint32_t gp = s_delta[b] + i;
            const uint32_t ok = (i < tn && gp < (b * AGR_NSUB + sub + 1u) * cap) ? 1u : 0u;   // rows past a range are dropped (flags[0] raised above)
            if (ASM) {
                st8_hidden_if(out_val + gp, v, ok);
                st4_hidden_if(out_slot + gp, sl, ok);
            } else if (ok) {
                reinterpret_cast<u32x2*>(out_val)[gp] = v;
                out_slot[gp] = sl;
            }
        }
        // (the next LDS writes that could collide -- s_start, s_delta, the window -- all sit behind the next tile's barriers)
    }
    if (ASM) wait_vmcnt<0>();
}

// the last n % 4 rows of the column (k_agg_scatter_rsv takes whole 16-byte units): appended to sub-range 0 of their buckets
template <typename KeyT, bool VAL_F64>
__global__ void k_agg_scatter_tail(const KeyT* __restrict__ keys, const void* __restrict__ val, unsigned long long first, unsigned long long n,
                                   uint32_t n_groups, uint32_t gb_shift, uint32_t* __restrict__ cursor, uint32_t cap, uint32_t* __restrict__ flags,
                                   uint32_t* __restrict__ out_slot, double* __restrict__ out_val) {
    const unsigned long long i = first + threadIdx.x;
    if (i >= n || *flags) return;
    const uint32_t sl = slot_of(keys[i], n_groups), b = sl >> gb_shift;
    const uint32_t pos = atomicAdd(&cursor[b * AGR_NSUB], 1u);
    if (pos >= cap) { atomicOr(flags, 1u); return; }
    out_slot[(size_t)b * AGR_NSUB * cap + pos] = sl;
    out_val[(size_t)b * AGR_NSUB * cap + pos] = VAL_F64 ? ((const double*)val)[i] : (double)((const long long*)val)[i];
}

// Will the provisioned ranges hold?  A strided sample of the key column (16-row runs, `stride` rows apart: 2^20 rows at most) is
// counted per bucket -- per work-group in LDS, then into tally[0 .. B) -- and the work-group that finishes last compares: a bucket
// whose share of the sample exceeds what a full bucket gets from uniformly spread keys (bucket_slots / n_groups: the provisioning's
// basis) by more than 1/16 (half the provisioning's slack; the sample's own sigma is ~1 % at 2^20 rows and 128 buckets) raises
// flags[0], and the histogram path runs instead at no extra cost.
constexpr int AGS_THREADS = 256;
template <typename KeyT>
__global__ __launch_bounds__(AGS_THREADS) void k_agg_sample(const KeyT* __restrict__ keys, unsigned long long n, uint32_t n_groups, uint32_t gb_shift, uint32_t B,
                                                            uint32_t runs, unsigned long long stride, uint32_t bucket_slots /*min(slots per bucket, n_groups)*/,
                                                            uint32_t* __restrict__ tally /*[B] counts, [B] rows sampled, [B + 1] work-groups done; zeroed*/,
                                                            uint32_t* __restrict__ flags) {
    __shared__ uint32_t s_cnt[AGR_MAX_B];
    __shared__ uint32_t s_last;
    for (uint32_t b = threadIdx.x; b < B; b += AGS_THREADS) s_cnt[b] = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t r = blockIdx.x * (AGS_THREADS / 16) + (threadIdx.x >> 4); r < runs; r += gridDim.x * (AGS_THREADS / 16)) {
        const unsigned long long i = (unsigned long long)r * stride + (threadIdx.x & 15);
        if (i < n) { atomicAdd(&s_cnt[slot_of(keys[i], n_groups) >> gb_shift], 1u); ++mine; }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < B; b += AGS_THREADS)
        if (s_cnt[b]) atomicAdd(&tally[b], s_cnt[b]);
    if (mine) atomicAdd(&tally[B], mine);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&tally[B + 1], 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    for (uint32_t b = threadIdx.x; b < B; b += AGS_THREADS) {
        const unsigned long long c = __hip_atomic_load(&tally[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long tot = __hip_atomic_load(&tally[B], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // c > tot * (bucket_slots / n_groups) * 17/16 + 64: more than a full bucket's share of uniformly spread keys, plus 1/16
        if (c * n_groups * 16ull > tot * bucket_slots * 17ull + 64ull * n_groups * 16ull) atomicOr(flags, 1u);
    }
}

// bucket b = slots [b*GB, (b+1)*GB); work-group (b, w) folds its share of the bucket's rows into an LDS table.  GB = AG_GB (2048
// slots x 28 bytes: all four statistics) or, when no MIN / MAX is asked for, AG_GB_SUM (8192 slots x 12 bytes: sum + count).
template <bool NEED_MM, int GB = AG_GB, int NT = AG_THREADS>
__global__ __launch_bounds__(NT) void k_agg_bucket(const uint32_t* __restrict__ slot, const double* __restrict__ val,
                                                   const uint32_t* __restrict__ scanned, uint32_t S, uint32_t W,
                                                   double* __restrict__ slab /*[B*W][4][GB]*/,
                                                   const uint32_t* __restrict__ rsv_cursor = nullptr /*k_agg_scatter_rsv's layout (W sub-ranges of rsv_cap rows per bucket) ...*/,
                                                   uint32_t rsv_cap = 0, const uint32_t* __restrict__ rsv_flags = nullptr /*... unless *rsv_flags != 0*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    double* s_sum = reinterpret_cast<double*>(s_raw);
    double* s_mn = s_sum + GB;                               // NEED_MM only
    double* s_mx = s_mn + GB;
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(NEED_MM ? s_mx + GB : s_sum + GB);
    const uint32_t b = blockIdx.x / W, w = blockIdx.x % W;
    for (uint32_t g = threadIdx.x; g < (uint32_t)GB; g += NT) {
        s_sum[g] = 0.0;
        s_cnt[g] = 0;
        if (NEED_MM) { s_mn[g] = GOLAP_F64_MAX; s_mx[g] = -GOLAP_F64_MAX; }
    }
    __syncthreads();
    // reserved layout: W < AGR_NSUB -- work-group (b, w) takes the bucket's sub-ranges w, w + W, ...; W = P * AGR_NSUB -- it takes slice
    // w / AGR_NSUB (of P) of sub-range w % AGR_NSUB
    const bool rsv = rsv_cursor && *rsv_flags == 0;
    const uint32_t kbase = b * GB;
    const uint32_t sub_step = W < (uint32_t)AGR_NSUB ? W : (uint32_t)AGR_NSUB, parts = W < (uint32_t)AGR_NSUB ? 1u : W / AGR_NSUB;
    for (uint32_t sub = w % AGR_NSUB; sub < (rsv ? (uint32_t)AGR_NSUB : w % AGR_NSUB + 1u); sub += sub_step) {
    uint32_t lo, hi;
    if (rsv) {
        const uint32_t c = rsv_cursor[b * AGR_NSUB + sub], len = c < rsv_cap ? c : rsv_cap;
        const uint32_t share = (len + parts - 1) / parts, first = (w / AGR_NSUB) * share;
        lo = (b * AGR_NSUB + sub) * rsv_cap + (first < len ? first : len);
        hi = (b * AGR_NSUB + sub) * rsv_cap + (first + share < len ? first + share : len);
    } else {
        const uint32_t o0 = scanned[(size_t)b * S], o1 = scanned[(size_t)(b + 1) * S];
        const uint32_t share = (o1 - o0 + W - 1) / W;
        lo = o0 + w * share < o1 ? o0 + w * share : o1;
        hi = lo + share < o1 ? lo + share : o1;
    }
    for (uint32_t base = lo; base < hi; base += NT * 8) {
        uint32_t k[8];
        double x[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            uint32_t i = base + r * NT + threadIdx.x;
            if (i >= hi) i = hi - 1;
            k[r] = slot[i];
            x[r] = val[i];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t i = base + r * NT + threadIdx.x;
            if (i < hi && !(k[r] & AG_NULL_MARK)) {
                const uint32_t g = k[r] - kbase;
                atomicAdd(&s_sum[g], x[r]);
                atomicAdd(&s_cnt[g], 1u);
                if (NEED_MM) {
                    __hip_atomic_fetch_min(&s_mn[g], x[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_max(&s_mx[g], x[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    }
    __syncthreads();
    double* row = slab + (size_t)blockIdx.x * 4 * GB;
    for (uint32_t g = threadIdx.x; g < (uint32_t)GB; g += NT) {
        row[g] = s_sum[g];
        row[GB + g] = (double)s_cnt[g];
        if (NEED_MM) {
            row[2 * GB + g] = s_mn[g];
            row[3 * GB + g] = s_mx[g];
        }
    }
}

__global__ __launch_bounds__(OP_THREADS) void k_agg_bucket_fold(const double* __restrict__ slab, uint32_t W, uint32_t n_groups, uint32_t GB, int need_mm,
                                                                double* __restrict__ stats4) {
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += gridDim.x * blockDim.x) {
        const uint32_t b = g / GB, l = g % GB;
        double sum = 0.0, cnt = 0.0, mn = GOLAP_F64_MAX, mx = -GOLAP_F64_MAX;
        for (uint32_t w = 0; w < W; ++w) {   // fixed order: reproducible
            const double* row = slab + (size_t)(b * W + w) * 4 * GB;
            sum += row[l]; cnt += row[GB + l];
            if (need_mm) {   // (the sum-only tables leave these two planes unwritten)
                mn = row[2 * GB + l] < mn ? row[2 * GB + l] : mn;
                mx = row[3 * GB + l] > mx ? row[3 * GB + l] : mx;
            }
        }
        stats4[g * 4 + 0] = sum; stats4[g * 4 + 1] = cnt; stats4[g * 4 + 2] = mn; stats4[g * 4 + 3] = mx;
    }
}

// large group counts: global table, agent-scope atomics per row ("global merge",
// aggregate_kernel.rs:6, :84-86).  Sums of integer-valued data are exact; for
// general doubles the order of the adds is not fixed (documented in DESIGN.md).
__global__ void k_group_init(double* __restrict__ stats4, uint32_t n_groups) {
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += gridDim.x * blockDim.x) {
        stats4[g * 4 + 0] = 0.0; stats4[g * 4 + 1] = 0.0; stats4[g * 4 + 2] = GOLAP_F64_MAX; stats4[g * 4 + 3] = -GOLAP_F64_MAX;
    }
}

template <typename KeyT, bool VAL_F64>
__global__ __launch_bounds__(OP_THREADS) void k_group_global(const KeyT* __restrict__ keys, const void* __restrict__ val, unsigned long long n,
                                                             uint32_t n_groups, double* __restrict__ stats4, int need_minmax, int need_cnt,
                                                             const uint8_t* __restrict__ vv = nullptr) {
    const long long* vi = (const long long*)val;
    const double* vf = (const double*)val;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        if (vv && !vv[i]) continue;
        const uint32_t g = slot_of(keys[i], n_groups);
        const double x = VAL_F64 ? vf[i] : (double)vi[i];
        unsafeAtomicAdd(&stats4[g * 4 + 0], x);
        if (need_cnt) unsafeAtomicAdd(&stats4[g * 4 + 1], 1.0);
        if (need_minmax) {
            __hip_atomic_fetch_min(&stats4[g * 4 + 2], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(&stats4[g * 4 + 3], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------
// widening (record_batch_convert.rs:102-110): null -> 0, 8-byte LE
// ---------------------------------------------------------------------------
template <typename In, typename Out>
__global__ __launch_bounds__(OP_THREADS) void k_widen(const In* __restrict__ in, const uint8_t* __restrict__ validity, unsigned long long n,
                                                      Out* __restrict__ out) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        out[i] = (validity && !validity[i]) ? (Out)0 : (Out)in[i];
}

// Utf8 -> FNV-1a 64 (record_batch_convert.rs:123-130), one value per thread; NULL -> 0 (:112-121)
__global__ __launch_bounds__(OP_THREADS) void k_hash_utf8(const long long* __restrict__ off, const uint8_t* __restrict__ data, const uint8_t* __restrict__ valid,
                                                          unsigned long long n, long long* __restrict__ out) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long h = 14695981039346656037ull;
        if (valid && !valid[i]) { out[i] = 0; continue; }
        for (long long j = off[i], e = off[i + 1]; j < e; ++j) { h ^= (unsigned long long)data[j]; h *= 1099511628211ull; }
        out[i] = (long long)h;
    }
}

// ---------------------------------------------------------------------------
// generators
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(OP_THREADS) void k_generate(int kind, unsigned long long seed, unsigned long long first, unsigned long long count,
                                                         unsigned long long modulus, void* __restrict__ out) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long g = first + i;
        switch (kind) {
            case GOLAP_GEN_BUILD_KEYS: ((long long*)out)[i] = (long long)((2654435761ull * g + 12345ull) % modulus); break;
            case GOLAP_GEN_SM64_MOD_I64: ((long long*)out)[i] = (long long)(sm64(seed, g) % modulus); break;
            case GOLAP_GEN_SM64_MOD_I32: ((int*)out)[i] = (int)(sm64(seed, g) % modulus); break;
            case GOLAP_GEN_SM64_MOD_F64: ((double*)out)[i] = (double)(sm64(seed, g) % modulus); break;
            case GOLAP_GEN_SHIFT32_I64: ((long long*)out)[i] = (long long)(g << 32); break;
        }
    }
}

// ---------------------------------------------------------------------------
// pair verifier
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(OP_THREADS) void k_verify_pairs(const uint2* __restrict__ pairs, unsigned long long n_pairs, const long long* __restrict__ bkeys,
                                                             const long long* __restrict__ pkeys, uint32_t brb, uint32_t prb,
                                                             unsigned long long* __restrict__ out4) {
    unsigned long long eq = 0, cs = 0, ps = 0, px = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint2 p = pairs[i];
        eq += bkeys[p.x - brb] == pkeys[p.y - prb] ? 1ull : 0ull;
        cs += pair_mix(p.x, p.y);
        ps += p.y;
        px ^= sm64(0x1234, p.y);
    }
    eq = wave_reduce_add64(eq);
    cs = wave_reduce_add64(cs);
    ps = wave_reduce_add64(ps);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) px ^= __shfl_xor(px, d, 64);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out4[0], eq);
        atomicAdd(&out4[1], cs);
        atomicAdd(&out4[2], ps);
        atomicXor(&out4[3], px);
    }
}

static int32_t aggregate_dense(golap_ctx* ctx, const golap_aggregate_spec* specs, uint32_t n_specs, const golap_aggregate_params* p, hipStream_t st);
static int32_t aggregate_hashed(golap_ctx* ctx, const golap_aggregate_spec* specs, uint32_t n_specs, const golap_aggregate_params* p, hipStream_t st);
__global__ void k_fill_i64(long long* __restrict__ a, unsigned long long n, long long v) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) a[i] = v;
}

// SEQ_WGS: work-groups per CU for the purely sequential scans (filter mask, filter + sum, ungrouped reduction).  The more a CU keeps in
// flight the less this chip's HBM delivers (DESIGN.md 4.1 "HBM ceiling"): at 1B rows two 256-thread work-groups per CU beat eight --
// filter + sum 2.82 -> 2.54 ms (6.3 TB/s), mask 1.89 -> 1.74, ungrouped SUM / MAX 1.49 -> 1.41 (scripts/dev/stream_ab.py); one per CU
// starves it (3.48 ms).  Gathers and generators keep eight.
constexpr uint32_t SEQ_WGS = 2;
static uint32_t stream_grid(golap_ctx* ctx, uint64_t n, uint32_t per_block, uint64_t per_cu = 8) {
    uint64_t want = ceil_div(std::max<uint64_t>(1, n), per_block);
    // guideline 11: cap the grid (8 per CU = ~2048 blocks), grid-stride the rest
    if (const char* e = knob("GOLAP_STREAM_WGS")) per_cu = (uint64_t)std::max(1, std::min(32, atoi(e)));   // A/B hook
    uint64_t cap = (uint64_t)ctx->prop.multiProcessorCount * per_cu;
    return (uint32_t)std::max<uint64_t>(1, std::min(want, cap));
}

}  // namespace golap

using namespace golap;

extern "C" {

int32_t golap_filter_launch(golap_ctx* ctx, const golap_filter_predicate* pred, const golap_filter_params* p, void* stream) {
    if (!ctx || !pred || !p) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    if (p->n_rows && (!p->column_ptr || !p->out_mask_ptr)) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null column/mask pointer");
    if (pred->op < 0 || pred->op > GOLAP_CMP_NEVER) return fail(ctx, GOLAP_ERR_INVALID_ARG, "bad cmp op %d", pred->op);
    if (p->n_rows == 0) return GOLAP_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick_stream(ctx, stream);
    KernelTimer t(ctx, "filter_mask", st, 9 * p->n_rows);
    const bool aligned = (p->column_ptr % 16 == 0) && (p->out_mask_ptr % 4 == 0);
    uint64_t done = 0;
    if (p->validity_ptr) {   // NULL-aware form: one row per lane
        hipLaunchKernelGGL(k_filter_mask1, dim3(stream_grid(ctx, p->n_rows, OP_THREADS, SEQ_WGS)), dim3(OP_THREADS), 0, st, (const long long*)p->column_ptr,
                           0ull, (unsigned long long)p->n_rows, pred->op, (long long)pred->literal, (uint8_t*)p->out_mask_ptr, (const uint8_t*)p->validity_ptr);
        done = p->n_rows;
    } else if (aligned && p->n_rows >= 4) {
        const uint64_t n4 = p->n_rows / 4;
        hipLaunchKernelGGL(k_filter_mask4, dim3(stream_grid(ctx, n4, OP_THREADS, SEQ_WGS)), dim3(OP_THREADS), 0, st, (const longlong2*)p->column_ptr,
                           (unsigned long long)n4, pred->op, (long long)pred->literal, (uint32_t*)p->out_mask_ptr);
        done = n4 * 4;
    }
    if (done < p->n_rows)
        hipLaunchKernelGGL(k_filter_mask1, dim3(stream_grid(ctx, p->n_rows - done, OP_THREADS, SEQ_WGS)), dim3(OP_THREADS), 0, st, (const long long*)p->column_ptr,
                           (unsigned long long)done, (unsigned long long)p->n_rows, pred->op, (long long)pred->literal, (uint8_t*)p->out_mask_ptr);
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

int32_t golap_filter_compact(golap_ctx* ctx, const golap_filter_predicate* pred, uint64_t column_ptr, uint64_t n_rows, uint64_t out_indices_ptr,
                             uint64_t out_count_ptr, void* stream) {
    return golap_filter_compact_nullable(ctx, pred, column_ptr, 0, n_rows, out_indices_ptr, out_count_ptr, stream);
}

int32_t golap_filter_compact_nullable(golap_ctx* ctx, const golap_filter_predicate* pred, uint64_t column_ptr, uint64_t validity_ptr, uint64_t n_rows,
                                      uint64_t out_indices_ptr, uint64_t out_count_ptr, void* stream) {
    if (!ctx || !pred) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    if (!out_count_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "out_count_ptr is null");
    if (n_rows && (!column_ptr || !out_indices_ptr)) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null column/indices pointer");
    if (pred->op < 0 || pred->op > GOLAP_CMP_NEVER) return fail(ctx, GOLAP_ERR_INVALID_ARG, "bad cmp op %d", pred->op);
    if (n_rows >= 0xFFFFFFF0ull) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "row ids are u32: at most 2^32-16 rows per batch");
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick_stream(ctx, stream);
    if (n_rows == 0) { GOLAP_TRY(zero_async(ctx, st, (void*)((void*)out_count_ptr), 8)); return GOLAP_OK; }
    const uint32_t n_blocks = (uint32_t)ceil_div(n_rows, FC_ROWS);
    const uint64_t n_words = (uint64_t)n_blocks * FC_RPT * OP_WAVES;   // one ballot word per 64-row strip
    GOLAP_TRY(ws_begin(ctx, (uint64_t)(n_blocks + 1) * 4 + ceil_div(n_blocks + 1, FLAT_SCAN_BLOCK) * 4 + n_words * 8 + 8192, st));
    uint32_t* counts = (uint32_t*)ws_take(ctx, (uint64_t)(n_blocks + 1) * 4);
    uint32_t* bsums = (uint32_t*)ws_take(ctx, ceil_div(n_blocks + 1, FLAT_SCAN_BLOCK) * 4 + 4);
    unsigned long long* bits = (unsigned long long*)ws_take(ctx, n_words * 8);
    if (!counts || !bsums || !bits) return fail(ctx, GOLAP_ERR_OOM, "filter workspace exhausted");
    GOLAP_TRY(zero_async(ctx, st, (void*)(counts + n_blocks), 4));
    {
        KernelTimer t(ctx, "filter_count", st, 8 * n_rows);
        hipLaunchKernelGGL(k_filter_count, dim3(n_blocks), dim3(OP_THREADS), 0, st, (const long long*)column_ptr, (unsigned long long)n_rows, pred->op,
                           (long long)pred->literal, counts, bits, (const uint8_t*)validity_ptr);
    }
    GOLAP_TRY(flat_scan(ctx, st, counts, n_blocks + 1, bsums));
    {
        KernelTimer t(ctx, "filter_compact", st, n_rows / 8);
        hipLaunchKernelGGL(k_filter_write, dim3(n_blocks), dim3(OP_THREADS), 0, st, (const unsigned long long*)bits, (const uint32_t*)counts, n_blocks,
                           (uint32_t*)out_indices_ptr, (unsigned long long*)out_count_ptr);
    }
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    ws_end(ctx, st);
    return GOLAP_OK;
}

int32_t golap_filter_compact_and(golap_ctx* ctx, const golap_filter_predicate* preds, const uint64_t* column_ptrs, const uint64_t* validity_ptrs,
                                 uint32_t n_preds, uint64_t n_rows, uint64_t out_indices_ptr, uint64_t out_count_ptr, void* stream) {
    if (!ctx || !preds || !column_ptrs) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    if (n_preds == 0 || n_preds > GOLAP_FILTER_AND_MAX) return fail(ctx, GOLAP_ERR_INVALID_ARG, "1 .. %d comparisons per conjunction", GOLAP_FILTER_AND_MAX);
    if (!out_count_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "out_count_ptr is null");
    if (n_rows && !out_indices_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null indices pointer");
    if (n_rows >= 0xFFFFFFF0ull) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "row ids are u32: at most 2^32-16 rows per batch");
    FilterAnd fa;
    memset(&fa, 0, sizeof fa);
    fa.n = n_preds;
    for (uint32_t k = 0; k < n_preds; ++k) {
        if (preds[k].op < 0 || preds[k].op > GOLAP_CMP_NEVER) return fail(ctx, GOLAP_ERR_INVALID_ARG, "bad cmp op %d", preds[k].op);
        if (n_rows && !column_ptrs[k]) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null column pointer");
        fa.col[k] = (const long long*)column_ptrs[k];
        fa.valid[k] = validity_ptrs ? (const uint8_t*)validity_ptrs[k] : nullptr;
        fa.lit[k] = (long long)preds[k].literal;
        fa.op[k] = preds[k].op;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick_stream(ctx, stream);
    if (n_rows == 0) { GOLAP_TRY(zero_async(ctx, st, (void*)out_count_ptr, 8)); return GOLAP_OK; }
    const uint32_t n_blocks = (uint32_t)ceil_div(n_rows, FC_ROWS);
    const uint64_t n_words = (uint64_t)n_blocks * FC_RPT * OP_WAVES;
    GOLAP_TRY(ws_begin(ctx, (uint64_t)(n_blocks + 1) * 4 + ceil_div(n_blocks + 1, FLAT_SCAN_BLOCK) * 4 + n_words * 8 + 8192, st));
    uint32_t* counts = (uint32_t*)ws_take(ctx, (uint64_t)(n_blocks + 1) * 4);
    uint32_t* bsums = (uint32_t*)ws_take(ctx, ceil_div(n_blocks + 1, FLAT_SCAN_BLOCK) * 4 + 4);
    unsigned long long* bits = (unsigned long long*)ws_take(ctx, n_words * 8);
    if (!counts || !bsums || !bits) return fail(ctx, GOLAP_ERR_OOM, "filter workspace exhausted");
    GOLAP_TRY(zero_async(ctx, st, (void*)(counts + n_blocks), 4));
    {
        KernelTimer t(ctx, "filter_count_and", st, 8ull * n_preds * n_rows);
        hipLaunchKernelGGL(k_filter_count_and, dim3(n_blocks), dim3(OP_THREADS), 0, st, fa, (unsigned long long)n_rows, counts, bits);
    }
    GOLAP_TRY(flat_scan(ctx, st, counts, n_blocks + 1, bsums));
    {
        KernelTimer t(ctx, "filter_compact", st, n_rows / 8);
        hipLaunchKernelGGL(k_filter_write, dim3(n_blocks), dim3(OP_THREADS), 0, st, (const unsigned long long*)bits, (const uint32_t*)counts, n_blocks,
                           (uint32_t*)out_indices_ptr, (unsigned long long*)out_count_ptr);
    }
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    ws_end(ctx, st);
    return GOLAP_OK;
}

int32_t golap_take_8byte(golap_ctx* ctx, uint64_t column_ptr, uint64_t indices_ptr, uint64_t count_ptr, uint64_t n_indices_max, uint64_t out_ptr,
                         void* stream) {
    if (!ctx) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null ctx");
    if (n_indices_max == 0) return GOLAP_OK;
    if (!column_ptr || !indices_ptr || !out_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick_stream(ctx, stream);
    KernelTimer t(ctx, "take_8byte", st, 20 * n_indices_max);
    hipLaunchKernelGGL(k_take8, dim3(stream_grid(ctx, n_indices_max, OP_THREADS)), dim3(OP_THREADS), 0, st, (const unsigned long long*)column_ptr,
                       (const uint32_t*)indices_ptr, (const unsigned long long*)count_ptr, (unsigned long long)n_indices_max,
                       (unsigned long long*)out_ptr);
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

int32_t golap_filter_sum(golap_ctx* ctx, const golap_filter_predicate* pred, uint64_t key_column_ptr, uint64_t value_column_ptr, int32_t value_is_f64,
                         uint64_t n_rows, uint64_t out_ptr, void* stream) {
    return golap_filter_sum_nullable(ctx, pred, key_column_ptr, 0, value_column_ptr, 0, value_is_f64, n_rows, out_ptr, stream);
}

int32_t golap_filter_sum_nullable(golap_ctx* ctx, const golap_filter_predicate* pred, uint64_t key_column_ptr, uint64_t key_validity_ptr,
                                  uint64_t value_column_ptr, uint64_t value_validity_ptr, int32_t value_is_f64, uint64_t n_rows, uint64_t out_ptr,
                                  void* stream) {
    if (!ctx || !pred || !out_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    if (n_rows && (!key_column_ptr || !value_column_ptr)) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null column pointer");
    if (pred->op < 0 || pred->op > GOLAP_CMP_NEVER) return fail(ctx, GOLAP_ERR_INVALID_ARG, "bad cmp op %d", pred->op);
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick_stream(ctx, stream);
    const uint32_t grid = stream_grid(ctx, n_rows, OP_THREADS * 8, SEQ_WGS);
    GOLAP_TRY(ws_begin(ctx, (uint64_t)grid * 32 + 4096, st));
    double* partials = (double*)ws_take(ctx, (uint64_t)grid * 32);
    if (!partials) return fail(ctx, GOLAP_ERR_OOM, "workspace exhausted");
    {
        KernelTimer t(ctx, "filter_sum", st, 16 * n_rows);
        if (value_is_f64)
            hipLaunchKernelGGL((k_reduce_partials<true, true>), dim3(grid), dim3(OP_THREADS), 0, st, (const long long*)key_column_ptr,
                               (const void*)value_column_ptr, (unsigned long long)n_rows, pred->op, (long long)pred->literal, partials,
                               (const uint8_t*)key_validity_ptr, (const uint8_t*)value_validity_ptr);
        else
            hipLaunchKernelGGL((k_reduce_partials<false, true>), dim3(grid), dim3(OP_THREADS), 0, st, (const long long*)key_column_ptr,
                               (const void*)value_column_ptr, (unsigned long long)n_rows, pred->op, (long long)pred->literal, partials,
                               (const uint8_t*)key_validity_ptr, (const uint8_t*)value_validity_ptr);
        hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(OP_THREADS), 0, st, partials, grid, (double*)out_ptr, 2u);   // sum, kept-count: straight into the caller's buffer
    }
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    ws_end(ctx, st);
    return GOLAP_OK;
}

int32_t golap_aggregate(golap_ctx* ctx, const golap_aggregate_spec* specs, uint32_t n_specs, const golap_aggregate_params* p, void* stream) {
    if (!ctx || !p || (n_specs && !specs)) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    if (n_specs == 0) return GOLAP_OK;
    if (!p->out_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "out_ptr is null");
    if (p->n_groups == 0) return fail(ctx, GOLAP_ERR_INVALID_ARG, "n_groups must be >= 1");
    const bool hashed_empty = p->key_mode == GOLAP_KEYS_HASHED && p->n_rows == 0;   // (an empty key column may come without a pointer)
    if (!p->group_key_ptr && p->n_groups != 1 && !hashed_empty) return fail(ctx, GOLAP_ERR_INVALID_ARG, "ungrouped aggregate needs n_groups == 1");
    if (p->key_dtype != GOLAP_DT_INT64 && p->key_dtype != GOLAP_DT_INT32) return fail(ctx, GOLAP_ERR_INVALID_ARG, "key_dtype must be INT64 or INT32");
    for (uint32_t i = 0; i < n_specs; ++i) {
        if (specs[i].func < GOLAP_AGG_SUM || specs[i].func > GOLAP_AGG_AVG) return fail(ctx, GOLAP_ERR_INVALID_ARG, "spec %u: bad func %d", i, specs[i].func);
        if (specs[i].value_col_index >= p->n_value_cols) return fail(ctx, GOLAP_ERR_INVALID_ARG, "spec %u: value_col_index out of range", i);
        if (specs[i].output_col_index >= n_specs) return fail(ctx, GOLAP_ERR_INVALID_ARG, "spec %u: output_col_index out of range", i);
    }
    if (p->n_rows && !p->value_ptrs) return fail(ctx, GOLAP_ERR_INVALID_ARG, "value_ptrs is null");
    if (p->key_mode != GOLAP_KEYS_DENSE && p->key_mode != GOLAP_KEYS_HASHED) return fail(ctx, GOLAP_ERR_INVALID_ARG, "bad key_mode %d", p->key_mode);
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick_stream(ctx, stream);
    if (p->key_mode == GOLAP_KEYS_HASHED) return aggregate_hashed(ctx, specs, n_specs, p, st);
    return aggregate_dense(ctx, specs, n_specs, p, st);
}

}  // extern "C"

namespace golap {

// the dense path: slot = key % n_groups (join_kernels.cuh:230); call with ctx->mu held
static int32_t aggregate_dense(golap_ctx* ctx, const golap_aggregate_spec* specs, uint32_t n_specs, const golap_aggregate_params* p, hipStream_t st) {
    const uint32_t G = p->n_groups;
    const bool grouped = p->group_key_ptr != 0;
    ctx->prof.cur_gate_host = nullptr;
    // LDS path while ONE table per CU fits: 28 bytes per group with MIN / MAX (5558 groups), 12 without (12 969 groups) in 152 KiB of the
    // CU's 160 KiB.  (Round 2 stopped at 64 KiB for two work-groups per CU; one per CU is the faster shape anyway, see per_cu below, and
    // group counts up to there are 3.5 x faster here than through the partition path.)
    bool any_mm_ = false;
    for (uint32_t i = 0; i < n_specs; ++i) any_mm_ = any_mm_ || specs[i].func == GOLAP_AGG_MIN || specs[i].func == GOLAP_AGG_MAX;
    // The budget comes from the device, not from the gfx950 figure: what one work-group may ask for (sharedMemPerBlockOptin,
    // falling back to sharedMemPerBlock) minus 8 KiB for the kernel's static LDS, capped at the 152 KiB the kernel was tuned with; a part
    // with a smaller LDS takes fewer groups on this path and the partition path beyond, instead of failing in hipFuncSetAttribute.
    const uint64_t lds_dev = ctx->prop.sharedMemPerBlockOptin ? ctx->prop.sharedMemPerBlockOptin : ctx->prop.sharedMemPerBlock;
    const uint64_t LDS_TABLE_BYTES = std::min<uint64_t>(152 * 1024, lds_dev > 16 * 1024 ? lds_dev - 8 * 1024 : 0);
    const bool lds_path = grouped && (uint64_t)G * (any_mm_ ? 28 : 12) + 16 <= LDS_TABLE_BYTES;
    // beyond one LDS table: partition by slot range, then LDS-aggregate (big inputs only; GOLAP_AGG_PART=0 -> global atomics)
    const char* part_env = knob("GOLAP_AGG_PART");
    const bool part_path = grouped && !lds_path && G <= (uint32_t)AG_GB * AG_MAX_B && p->n_rows >= (1ull << 22) && p->n_rows < 0xFFFFFFF0ull &&
                           !(part_env && !strcmp(part_env, "0"));
    // big-tile scatter once every chunk holds >= 4 of its 16384-row tiles and there are >= 512 chunks
    const bool p_big = part_path && p->n_rows / (4 * AGB_TILE) >= 512;
    // no MIN / MAX anywhere: 12-byte table entries, 8192 slots per bucket -> a fan-out <= 128, whose unfinished lines fit in LDS next
    // to the tile (k_agg_scatter_wc).  GOLAP_AGG_WC=0: A/B hook.
    bool any_mm = false;
    for (uint32_t i = 0; i < n_specs; ++i) any_mm = any_mm || specs[i].func == GOLAP_AGG_MIN || specs[i].func == GOLAP_AGG_MAX;
    const char* wc_env = knob("GOLAP_AGG_WC");
    const bool p_wc = p_big && !any_mm && ceil_div(G, AG_GB_SUM) <= (uint64_t)AGW_MAX_B && !(wc_env && !strcmp(wc_env, "0"));
    const uint32_t pGB = p_wc ? (uint32_t)AG_GB_SUM : (uint32_t)AG_GB;
    const uint32_t gb_shift = p_wc ? 13u : 11u;
    static_assert(AG_GB == 1 << 11 && AG_GB_SUM == 1 << 13, "bucket = slot >> gb_shift");
    const uint32_t pB = part_path ? (uint32_t)ceil_div(G, pGB) : 0;
    const uint32_t pS = part_path ? (uint32_t)std::min<uint64_t>(2048, p_big ? p->n_rows / (4 * AGB_TILE) : ceil_div(p->n_rows, AG_TILE)) : 0;
    const uint32_t p_chunk = part_path ? (uint32_t)round_up(ceil_div(p->n_rows, pS), AG_TILE) : 0;
    const uint32_t pS2 = part_path ? (uint32_t)ceil_div(p->n_rows, p_chunk) : 0;   // chunks actually used
    // work-groups per bucket in the fold: >= 512 (sums only: one 96 KiB table per CU) / 1024 (all four statistics: 56 KiB) work-groups
    // in all, 8 .. 256 per bucket in multiples of 8 (the reserved layout's sub-ranges); fan-outs beyond 128: min(8, 2048 / B)
    const uint32_t pW = !part_path ? 0 : pB > 128 ? std::max<uint32_t>(1, std::min<uint32_t>(8, 2048u / pB))
                                                  : 8u * (uint32_t)std::min<uint64_t>(32, std::max<uint64_t>(1, ceil_div(p_wc ? 512 : 1024, 8ull * pB)));
    // no histogram pass: provisioned sub-ranges + reservations (k_agg_scatter_rsv), the histogram path standing by behind a device
    // word.  Needs 16-byte aligned columns without validity bytes (checked per value column).  GOLAP_AGG_RSV=0: A/B hook; =plain: the
    // compiler-visible form of the kernel.
    const char* rsv_env = knob("GOLAP_AGG_RSV");
    const uint64_t n4 = p->n_rows & ~3ull;
    // (provisioned for a FULL bucket's share of uniformly spread keys -- the last bucket may be narrower -- plus 1/8 and a tile)
    const uint32_t bucket_slots = std::min<uint32_t>(pGB, G);
    const uint64_t rsv_cap64 = p_big ? round_up(ceil_div(ceil_div(p->n_rows * bucket_slots, G), AGR_NSUB) * 9 / 8 + AGR_TILE, 32) : 0;
    const bool p_rsv = p_big && pB <= (uint32_t)AGR_MAX_B && (AGR_NSUB % pW == 0 || pW % AGR_NSUB == 0) && (uint64_t)pB * AGR_NSUB * rsv_cap64 < 0xFFFFFFF0ull &&
                       (p->group_key_ptr & 15) == 0 && !(rsv_env && !strcmp(rsv_env, "0"));
    const uint32_t rsv_cap = (uint32_t)rsv_cap64;
    const uint64_t part_rows = p_rsv ? (uint64_t)pB * AGR_NSUB * rsv_cap : p->n_rows;   // rows of the partition buffers
    const uint32_t rsv_chunk = p_rsv ? (uint32_t)round_up(ceil_div(n4, 2048), AGR_TILE) : 0;
    const uint32_t rsv_grid = p_rsv ? (uint32_t)ceil_div(n4, rsv_chunk) : 0;
    const uint32_t cus = (uint32_t)ctx->prop.multiProcessorCount;
    uint32_t lds_grid = 0;
    uint64_t rows_per_block = 0;
    if (lds_path) {
        // ONE 512-thread work-group per CU: with the 16-byte loads a second one only adds bytes in flight (96 KB per CU and round), and
        // this chip's HBM delivers less the more a CU keeps in flight (DESIGN.md 4.1 "HBM ceiling"): 1B rows, 1024 groups 2.05 -> 1.93 ms,
        // 2340 groups 2.10 -> 1.90 ms, the same with MIN / MAX (scripts/ab_agg.py).  Half as many partial tables to fold, too.
        uint32_t per_cu = 1;
        if (const char* e = knob("GOLAP_AGG_WGS_PER_CU")) per_cu = (uint32_t)std::max(1, std::min(8, atoi(e)));   // A/B hook
        lds_grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)cus * per_cu, ceil_div(std::max<uint64_t>(1, p->n_rows), 4096)));
        rows_per_block = round_up(ceil_div(std::max<uint64_t>(1, p->n_rows), lds_grid), 4);   // (a multiple of 4: the 16-byte-load form)
    }
    const uint32_t red_grid = stream_grid(ctx, p->n_rows, OP_THREADS * 8, SEQ_WGS);
    uint64_t ws = (uint64_t)G * 32 + 8192;
    ws += lds_path ? (uint64_t)lds_grid * 4 * G * 8 : 0;
    if (part_path)
        ws += round_up(part_rows * 4, 256) + round_up(part_rows * 8, 256) + round_up(((uint64_t)pB * pS2 + 1) * 4, 256) +
              round_up(ceil_div((uint64_t)pB * pS2 + 1, FLAT_SCAN_BLOCK) * 4 + 4, 256) + round_up((uint64_t)pB * pW * 4 * pGB * 8, 256) + 4096 +
              (p_rsv ? round_up(((uint64_t)pB * (AGR_NSUB + 1) + 48) * 4, 256) : 0);
    ws += grouped ? 0 : (uint64_t)red_grid * 32;
    GOLAP_TRY(ws_begin(ctx, ws + 4096, st));
    double* stats4 = (double*)ws_take(ctx, (uint64_t)G * 32);
    double* slab = lds_path ? (double*)ws_take(ctx, (uint64_t)lds_grid * 4 * G * 8) : nullptr;
    double* partials = grouped ? nullptr : (double*)ws_take(ctx, (uint64_t)red_grid * 32);
    if (!stats4 || (lds_path && !slab) || (!grouped && !partials)) return fail(ctx, GOLAP_ERR_OOM, "aggregate workspace exhausted");
    uint32_t *p_slot = nullptr, *p_hist = nullptr, *p_bsums = nullptr, *p_rsvw = nullptr;
    double *p_val = nullptr, *p_slab = nullptr;
    if (part_path) {
        p_slot = (uint32_t*)ws_take(ctx, part_rows * 4);
        p_val = (double*)ws_take(ctx, part_rows * 8);
        if (p_rsv) {   // [0] flag, [1 .. 32) spare, [32 ..) cursors [B * NSUB], then the sample's tally [B + 2]
            p_rsvw = (uint32_t*)ws_take(ctx, ((uint64_t)pB * (AGR_NSUB + 1) + 48) * 4);
            if (!p_rsvw) return fail(ctx, GOLAP_ERR_OOM, "aggregate workspace exhausted (partition buffers)");
        }
        p_hist = (uint32_t*)ws_take(ctx, ((uint64_t)pB * pS2 + 1) * 4);
        p_bsums = (uint32_t*)ws_take(ctx, ceil_div((uint64_t)pB * pS2 + 1, FLAT_SCAN_BLOCK) * 4 + 4);
        p_slab = (double*)ws_take(ctx, (uint64_t)pB * pW * 4 * pGB * 8);
        if (!p_slot || !p_val || !p_hist || !p_bsums || !p_slab) return fail(ctx, GOLAP_ERR_OOM, "aggregate workspace exhausted (partition buffers)");
        if (ctx->attr_done.size() < 3) ctx->attr_done.resize(3, false);   // per context = per device (golap_internal.h)
        if (!ctx->attr_done[1]) {
            GOLAP_HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_agg_bucket<true>, hipFuncAttributeMaxDynamicSharedMemorySize, AG_GB * 28 + 16));
            GOLAP_HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_agg_bucket<false>, hipFuncAttributeMaxDynamicSharedMemorySize, AG_GB * 28 + 16));
            const void* big[] = {(const void*)k_agg_scatter_big<int, true>, (const void*)k_agg_scatter_big<int, false>,
                                 (const void*)k_agg_scatter_big<long long, true>, (const void*)k_agg_scatter_big<long long, false>};
            for (const void* f : big) GOLAP_HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AGB_LDS));
            const void* wcs[] = {(const void*)k_agg_scatter_wc<int, true, 1024, 9216>, (const void*)k_agg_scatter_wc<int, false, 1024, 9216>,
                                 (const void*)k_agg_scatter_wc<long long, true, 1024, 9216>, (const void*)k_agg_scatter_wc<long long, false, 1024, 9216>};
            for (const void* f : wcs) GOLAP_HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)agw_lds(9216)));
            const void* wcs2[] = {(const void*)k_agg_scatter_wc<int, true, 512, 3584>, (const void*)k_agg_scatter_wc<int, false, 512, 3584>,
                                  (const void*)k_agg_scatter_wc<long long, true, 512, 3584>, (const void*)k_agg_scatter_wc<long long, false, 512, 3584>};
            for (const void* f : wcs2) GOLAP_HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)agw_lds(3584)));
            const void* rsvs[] = {(const void*)k_agg_scatter_rsv<int, true, true>,        (const void*)k_agg_scatter_rsv<int, false, true>,
                                  (const void*)k_agg_scatter_rsv<long long, true, true>,  (const void*)k_agg_scatter_rsv<long long, false, true>,
                                  (const void*)k_agg_scatter_rsv<int, true, false>,       (const void*)k_agg_scatter_rsv<int, false, false>,
                                  (const void*)k_agg_scatter_rsv<long long, true, false>, (const void*)k_agg_scatter_rsv<long long, false, false>};
            for (const void* f : rsvs) GOLAP_HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)agr_lds(AGR_MAX_B)));
            GOLAP_HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_agg_bucket<false, AG_GB_SUM, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, AG_GB_SUM * 12 + 16));
            ctx->attr_done[1] = true;
        }
    }
    if (lds_path) {
        if (ctx->attr_done.size() < 3) ctx->attr_done.resize(3, false);
        if (!ctx->attr_done[2]) {
            const void* fns[] = {(const void*)k_group_lds<int, true, true>,        (const void*)k_group_lds<int, true, false>,
                                 (const void*)k_group_lds<int, false, true>,       (const void*)k_group_lds<int, false, false>,
                                 (const void*)k_group_lds<long long, true, true>,  (const void*)k_group_lds<long long, true, false>,
                                 (const void*)k_group_lds<long long, false, true>, (const void*)k_group_lds<long long, false, false>};
            for (const void* f : fns) GOLAP_HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TABLE_BYTES));
            ctx->attr_done[2] = true;
        }
    }
    // one pass per distinct value column; every spec on that column is finalised from the same stats
    for (uint32_t col = 0; col < p->n_value_cols; ++col) {
        bool used = false, need_minmax = false, need_cnt = p->out_count_ptr != 0;
        for (uint32_t i = 0; i < n_specs; ++i)
            if (specs[i].value_col_index == col) {
                used = true;
                need_minmax = need_minmax || specs[i].func == GOLAP_AGG_MIN || specs[i].func == GOLAP_AGG_MAX;
                need_cnt = need_cnt || specs[i].func == GOLAP_AGG_COUNT || specs[i].func == GOLAP_AGG_AVG;
            }
        if (!used) continue;
        const int32_t vdt = p->value_dtypes ? p->value_dtypes[col] : GOLAP_DT_INT64;
        if (vdt != GOLAP_DT_INT64 && vdt != GOLAP_DT_FLOAT64) return fail(ctx, GOLAP_ERR_INVALID_ARG, "value column %u: dtype must be INT64 or FLOAT64", col);
        const bool vf = vdt == GOLAP_DT_FLOAT64;
        const void* val = p->n_rows ? (const void*)p->value_ptrs[col] : nullptr;
        const uint8_t* vv = (p->value_validity_ptrs && p->n_rows) ? (const uint8_t*)p->value_validity_ptrs[col] : nullptr;   // 0 = NULL row
        if (p->n_rows && !val) return fail(ctx, GOLAP_ERR_INVALID_ARG, "value column %u is null", col);
        const uint64_t key_bytes = grouped ? (p->key_dtype == GOLAP_DT_INT32 ? 4 : 8) : 0;
        if (!grouped) {
            KernelTimer t(ctx, "aggregate_ungrouped", st, 8 * p->n_rows);
            if (vf) hipLaunchKernelGGL((k_reduce_partials<true, false>), dim3(red_grid), dim3(OP_THREADS), 0, st, (const long long*)nullptr, val,
                                       (unsigned long long)p->n_rows, 0, 0ll, partials, (const uint8_t*)nullptr, vv);
            else hipLaunchKernelGGL((k_reduce_partials<false, false>), dim3(red_grid), dim3(OP_THREADS), 0, st, (const long long*)nullptr, val,
                                    (unsigned long long)p->n_rows, 0, 0ll, partials, (const uint8_t*)nullptr, vv);
            hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(OP_THREADS), 0, st, partials, red_grid, stats4);
        } else if (lds_path) {
            const size_t lds = (size_t)G * (need_minmax ? 28 : 12) + 16;
            const char* vec_env = knob("GOLAP_AGG_VEC");   // "0": 4- / 8-byte loads (A/B hook)
            const int lds_vec = !vv && ((p->group_key_ptr | (uint64_t)val) & 15) == 0 && !(vec_env && !strcmp(vec_env, "0")) ? 1 : 0;
            {
                KernelTimer t(ctx, "hash_aggregate_lds", st, (key_bytes + 8) * p->n_rows + 12ull * G);
#define GOLAP_LAUNCH_LDS(KT, VF, MM)                                                                                              \
    hipLaunchKernelGGL((k_group_lds<KT, VF, MM>), dim3(lds_grid), dim3(512), lds, st, (const KT*)p->group_key_ptr, val,          \
                       (unsigned long long)p->n_rows, G, (unsigned long long)rows_per_block, slab, vv, lds_vec)
                if (p->key_dtype == GOLAP_DT_INT32) {
                    if (vf) { if (need_minmax) GOLAP_LAUNCH_LDS(int, true, true); else GOLAP_LAUNCH_LDS(int, true, false); }
                    else { if (need_minmax) GOLAP_LAUNCH_LDS(int, false, true); else GOLAP_LAUNCH_LDS(int, false, false); }
                } else {
                    if (vf) { if (need_minmax) GOLAP_LAUNCH_LDS(long long, true, true); else GOLAP_LAUNCH_LDS(long long, true, false); }
                    else { if (need_minmax) GOLAP_LAUNCH_LDS(long long, false, true); else GOLAP_LAUNCH_LDS(long long, false, false); }
                }
#undef GOLAP_LAUNCH_LDS
            }
            {
                KernelTimer t(ctx, "hash_aggregate_fold", st, 0);
                hipLaunchKernelGGL(k_group_fold, dim3((uint32_t)ceil_div(G, 64)), dim3(GF_THREADS), 0, st, slab, lds_grid, G, stats4);
            }
        } else if (part_path) {
            const uint64_t L = (uint64_t)pB * pS2;
            // the reserving scatter first (when this column allows it); everything of the histogram path below then runs only if the
            // device word says so
            const bool rsv = p_rsv && !vv && ((uint64_t)val & 15) == 0;
            uint32_t *rsv_flags = nullptr, *rsv_cursor = nullptr;
            if (rsv) {
                rsv_flags = p_rsvw; rsv_cursor = p_rsvw + 32;   // (the gate word in a 128-byte line of its own, away from the reservation atomics)
                uint32_t* tally = rsv_cursor + (size_t)pB * AGR_NSUB;
                GOLAP_TRY(zero_async(ctx, st, p_rsvw, ((uint64_t)pB * (AGR_NSUB + 1) + 48) * 4));
                if (ctx->prof.enabled) {   // which of the two scatters did the work is read back (asynchronously) for the byte credits
                    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                    if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
                    ctx->prof.cur_gate_host = cs == hipStreamCaptureStatusNone ? ctx->prof.take_gate_slot() : nullptr;
                }
                const bool plain = rsv_env ? !strcmp(rsv_env, "plain") : !kernel_forms_now().agg_rsv_hidden;   // (golap_variants.h)
                const uint32_t runs = (uint32_t)std::min<uint64_t>(65536, p->n_rows / 16);
                const uint64_t stride = p->n_rows / runs;
                KernelTimer t(ctx, "agg_partition_scatter", st, 0, (key_bytes + 8 + 12) * p->n_rows, false);
#define GOLAP_AGG_RSV(KT, VF)                                                                                                         \
    do {                                                                                                                              \
        hipLaunchKernelGGL(k_agg_sample<KT>, dim3(256), dim3(AGS_THREADS), 0, st, (const KT*)p->group_key_ptr, (unsigned long long)p->n_rows, G, \
                           gb_shift, pB, runs, (unsigned long long)stride, bucket_slots, tally, rsv_flags);                                          \
        if (plain)                                                                                                                    \
            hipLaunchKernelGGL((k_agg_scatter_rsv<KT, VF, false>), dim3(rsv_grid), dim3(AGR_THREADS), agr_lds(pB), st, (const KT*)p->group_key_ptr, \
                               val, (unsigned long long)n4, G, gb_shift, rsv_chunk, pB, rsv_cursor, rsv_cap, rsv_flags, p_slot, p_val); \
        else                                                                                                                          \
            hipLaunchKernelGGL((k_agg_scatter_rsv<KT, VF, true>), dim3(rsv_grid), dim3(AGR_THREADS), agr_lds(pB), st, (const KT*)p->group_key_ptr, \
                               val, (unsigned long long)n4, G, gb_shift, rsv_chunk, pB, rsv_cursor, rsv_cap, rsv_flags, p_slot, p_val); \
        if (n4 != p->n_rows)                                                                                                          \
            hipLaunchKernelGGL((k_agg_scatter_tail<KT, VF>), dim3(1), dim3(64), 0, st, (const KT*)p->group_key_ptr, val, (unsigned long long)n4, \
                               (unsigned long long)p->n_rows, G, gb_shift, rsv_cursor, rsv_cap, rsv_flags, p_slot, p_val);             \
    } while (0)
                if (p->key_dtype == GOLAP_DT_INT32) { if (vf) GOLAP_AGG_RSV(int, true); else GOLAP_AGG_RSV(int, false); }
                else { if (vf) GOLAP_AGG_RSV(long long, true); else GOLAP_AGG_RSV(long long, false); }
#undef GOLAP_AGG_RSV
            }
            GOLAP_TRY(zero_async(ctx, st, p_hist, (L + 1) * 4));
            {
                KernelTimer t(ctx, rsv ? "agg_partition_hist_standby" : "agg_partition_hist", st, rsv ? 0 : key_bytes * p->n_rows, rsv ? key_bytes * p->n_rows : 0, true);
                if (p->key_dtype == GOLAP_DT_INT32)
                    hipLaunchKernelGGL(k_agg_hist<int>, dim3(pS2), dim3(AG_THREADS), 0, st, (const int*)p->group_key_ptr, (unsigned long long)p->n_rows, G,
                                       p_chunk, pS2, pB, p_hist, gb_shift, (const uint32_t*)rsv_flags);
                else
                    hipLaunchKernelGGL(k_agg_hist<long long>, dim3(pS2), dim3(AG_THREADS), 0, st, (const long long*)p->group_key_ptr,
                                       (unsigned long long)p->n_rows, G, p_chunk, pS2, pB, p_hist, gb_shift, (const uint32_t*)rsv_flags);
            }
            GOLAP_TRY(flat_scan(ctx, st, p_hist, (uint32_t)(L + 1), p_bsums));   // (of zeros when the histogram did not run)
            {
                KernelTimer t(ctx, rsv ? "agg_partition_scatter_standby" : "agg_partition_scatter", st, rsv ? 0 : (key_bytes + 8 + 12) * p->n_rows,
                              rsv ? (key_bytes + 8 + 12) * p->n_rows : 0, true);
#define GOLAP_AGG_SC(KT, VF)                                                                                                          \
    hipLaunchKernelGGL((k_agg_scatter<KT, VF>), dim3(pS2), dim3(AG_THREADS), 0, st, (const KT*)p->group_key_ptr, val,                 \
                       (unsigned long long)p->n_rows, G, p_chunk, pS2, pB, p_hist, p_slot, p_val, vv)
#define GOLAP_AGG_SCB(KT, VF)                                                                                                         \
    hipLaunchKernelGGL((k_agg_scatter_big<KT, VF>), dim3(pS2), dim3(AGB_THREADS), AGB_LDS, st, (const KT*)p->group_key_ptr, val,      \
                       (unsigned long long)p->n_rows, G, p_chunk, pS2, pB, p_hist, p_slot, p_val, vv, (const uint32_t*)rsv_flags)
#define GOLAP_AGG_SCW(KT, VF)                                                                                                         \
    do {                                                                                                                              \
        if (wc_two)                                                                                                                   \
            hipLaunchKernelGGL((k_agg_scatter_wc<KT, VF, 512, 3584>), dim3(pS2), dim3(512), agw_lds(3584), st, (const KT*)p->group_key_ptr, val, \
                               (unsigned long long)p->n_rows, G, gb_shift, p_chunk, pS2, pB, p_hist, p_slot, p_val, vv, (const uint32_t*)rsv_flags); \
        else                                                                                                                          \
            hipLaunchKernelGGL((k_agg_scatter_wc<KT, VF, 1024, 9216>), dim3(pS2), dim3(1024), agw_lds(9216), st, (const KT*)p->group_key_ptr, val, \
                               (unsigned long long)p->n_rows, G, gb_shift, p_chunk, pS2, pB, p_hist, p_slot, p_val, vv, (const uint32_t*)rsv_flags); \
    } while (0)
                // GOLAP_AGG_WC=512: the two-work-groups-per-CU shape (A/B hook).  Measured at 1B rows, 2^20 groups: 8.7 ms against 6.2 ms for
                // the 1024-thread shape -- the per-tile fixed work (bucket scan, carry staging per window round) is paid twice as often
                // and outweighs what the second work-group overlaps.
                const bool wc_two = wc_env && !strcmp(wc_env, "512");
                if (p_wc) {
                    if (p->key_dtype == GOLAP_DT_INT32) { if (vf) GOLAP_AGG_SCW(int, true); else GOLAP_AGG_SCW(int, false); }
                    else { if (vf) GOLAP_AGG_SCW(long long, true); else GOLAP_AGG_SCW(long long, false); }
                } else
#undef GOLAP_AGG_SCW
                if (p_big) {
                    if (p->key_dtype == GOLAP_DT_INT32) { if (vf) GOLAP_AGG_SCB(int, true); else GOLAP_AGG_SCB(int, false); }
                    else { if (vf) GOLAP_AGG_SCB(long long, true); else GOLAP_AGG_SCB(long long, false); }
                } else
                if (p->key_dtype == GOLAP_DT_INT32) { if (vf) GOLAP_AGG_SC(int, true); else GOLAP_AGG_SC(int, false); }
                else { if (vf) GOLAP_AGG_SC(long long, true); else GOLAP_AGG_SC(long long, false); }
#undef GOLAP_AGG_SC
#undef GOLAP_AGG_SCB
            }
            {
                KernelTimer t(ctx, "hash_aggregate_lds_buckets", st, 12 * p->n_rows + 12ull * G);
                if (p_wc)
                    hipLaunchKernelGGL((k_agg_bucket<false, AG_GB_SUM, 1024>), dim3(pB * pW), dim3(1024), AG_GB_SUM * 12 + 16, st, p_slot, p_val, p_hist, pS2, pW, p_slab,
                                       (const uint32_t*)rsv_cursor, rsv_cap, (const uint32_t*)rsv_flags);
                else if (need_minmax)
                    hipLaunchKernelGGL(k_agg_bucket<true>, dim3(pB * pW), dim3(AG_THREADS), AG_GB * 28 + 16, st, p_slot, p_val, p_hist, pS2, pW, p_slab,
                                       (const uint32_t*)rsv_cursor, rsv_cap, (const uint32_t*)rsv_flags);
                else
                    hipLaunchKernelGGL(k_agg_bucket<false>, dim3(pB * pW), dim3(AG_THREADS), AG_GB * 28 + 16, st, p_slot, p_val, p_hist, pS2, pW, p_slab,
                                       (const uint32_t*)rsv_cursor, rsv_cap, (const uint32_t*)rsv_flags);
                hipLaunchKernelGGL(k_agg_bucket_fold, dim3((uint32_t)std::min<uint64_t>(cus * 8, ceil_div(G, OP_THREADS))), dim3(OP_THREADS), 0, st, p_slab,
                                   pW, G, pGB, (int)(need_minmax && !p_wc), stats4);
            }
            if (rsv && ctx->prof.cur_gate_host) {
                (void)hipMemcpyAsync((void*)ctx->prof.cur_gate_host, rsv_flags, 4, hipMemcpyDeviceToHost, st);
                ctx->prof.cur_gate_host = nullptr;
            }
        } else {
            hipLaunchKernelGGL(k_group_init, dim3((uint32_t)std::min<uint64_t>(cus * 8, ceil_div(G, OP_THREADS))), dim3(OP_THREADS), 0, st, stats4, G);
            KernelTimer t(ctx, "hash_aggregate_global", st, (key_bytes + 8) * p->n_rows + 12ull * G);
            const uint32_t grid = stream_grid(ctx, p->n_rows, OP_THREADS * 4);
            if (p->key_dtype == GOLAP_DT_INT32) {
                if (vf) hipLaunchKernelGGL((k_group_global<int, true>), dim3(grid), dim3(OP_THREADS), 0, st, (const int*)p->group_key_ptr, val,
                                           (unsigned long long)p->n_rows, G, stats4, (int)need_minmax, (int)need_cnt, vv);
                else hipLaunchKernelGGL((k_group_global<int, false>), dim3(grid), dim3(OP_THREADS), 0, st, (const int*)p->group_key_ptr, val,
                                        (unsigned long long)p->n_rows, G, stats4, (int)need_minmax, (int)need_cnt, vv);
            } else {
                if (vf) hipLaunchKernelGGL((k_group_global<long long, true>), dim3(grid), dim3(OP_THREADS), 0, st, (const long long*)p->group_key_ptr, val,
                                           (unsigned long long)p->n_rows, G, stats4, (int)need_minmax, (int)need_cnt, vv);
                else hipLaunchKernelGGL((k_group_global<long long, false>), dim3(grid), dim3(OP_THREADS), 0, st, (const long long*)p->group_key_ptr, val,
                                        (unsigned long long)p->n_rows, G, stats4, (int)need_minmax, (int)need_cnt, vv);
            }
        }
        const uint32_t fg = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(cus * 8, ceil_div(G, OP_THREADS)));
        for (uint32_t i = 0; i < n_specs; ++i) {
            if (specs[i].value_col_index != col) continue;
            SpecDev sd{specs[i].func, (uint32_t)specs[i].output_col_index};
            hipLaunchKernelGGL(k_finalize_specs, dim3(fg), dim3(OP_THREADS), 0, st, stats4, G, sd, (double*)p->out_ptr);
        }
        if (p->out_count_ptr) hipLaunchKernelGGL(k_counts_from_stats, dim3(fg), dim3(OP_THREADS), 0, st, stats4, G, (unsigned long long*)p->out_count_ptr);
    }
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    ws_end(ctx, st);
    return GOLAP_OK;
}

// ---------------------------------------------------------------------------
// GOLAP_KEYS_HASHED: GROUP BY over arbitrary int64 / int32 keys.
// aggregate_kernel.rs:76-77 plans "a global-memory hash table with open addressing"; the kernel the reference ships
// (join_kernels.cuh:216-240) has none -- it adds into aggregates[key % num_groups] and silently merges congruent keys.
// Here a key -> slot table in device memory gives every distinct key a dense slot, and the dense machinery above runs on
// the slots:
//   k_hk_insert   every row's key into an open-addressing table (linear probing, 64-bit CAS on the key word only; no row
//                 waits for another -- slots get their ids in a later kernel, so there is no spin inside a wave).  A
//                 direct-mapped per-work-group LDS cache of "keys already in the table" keeps repeated keys (the common
//                 case: few groups) away from global memory;
//   flat scan     occupied entries -> exclusive prefix = slot ids in TABLE order (reproducible: independent of arrival order);
//   k_hk_keys     the key of every slot (out_keys), the number of slots (out_n_groups);
//   k_hk_lookup   key -> slot id per row (read-only probing, LDS cache of (key, id)), written as a u32 column;
//   aggregate_dense on the u32 column (every slot id < n_groups, so its `% n_groups` is the identity).
// The empty-entry marker is INT64_MIN; a real key INT64_MIN gets the slot after the table's (a flag records its presence).
// ---------------------------------------------------------------------------
constexpr long long HK_EMPTY = (long long)0x8000000000000000ull;
constexpr int HK_CACHE = 2048;   // direct-mapped LDS cache entries per work-group

__device__ __forceinline__ uint32_t hk_hash(long long key, uint32_t mask) {
    unsigned long long z = (unsigned long long)key * 0x9E3779B97F4A7C15ull;
    z ^= z >> 32;
    return (uint32_t)z & mask;
}

template <typename KeyT>
__global__ __launch_bounds__(OP_THREADS) void k_hk_insert(const KeyT* __restrict__ keys, unsigned long long n, long long* __restrict__ table, uint32_t mask,
                                                          uint32_t* __restrict__ flags /*[0] INT64_MIN seen, [1] table (nearly) full, [2] keys in the table*/) {
    __shared__ long long s_seen[HK_CACHE];
    for (uint32_t i = threadIdx.x; i < HK_CACHE; i += OP_THREADS) s_seen[i] = HK_EMPTY;
    __syncthreads();
    // The table has >= 2 * n_groups entries; once 7/8 of them are taken the caller's bound is wrong by a wide margin and probe chains
    // grow without limit (a FULL table made every further row scan all of it: minutes for a 2^23-entry table and 40M rows).  The
    // kernel counts the keys it adds (one atomic per wave and row round), raises flags[1] at 7/8 and every thread gives up soon after:
    // out_n_groups then reads 0xFFFFFFFF, as before.  Below 7/8 the count of distinct keys stays exact, also above n_groups.
    const uint32_t limit = mask - (mask >> 3);
    uint32_t round = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * OP_THREADS + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * OP_THREADS) {
        const long long key = (long long)keys[i];
        bool added = false;
        if (key == HK_EMPTY) {
            flags[0] = 1u;   // (same value from every writer)
        } else {
            const uint32_t h = hk_hash(key, mask);
            const uint32_t c = h & (HK_CACHE - 1);
            if (s_seen[c] != key) {                              // (else: this work-group already made sure the key is in the table)
                uint32_t s = h, probes = 0;
                bool there = false;
                for (;;) {
                    const long long cur = __hip_atomic_load(&table[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cur == key) { there = true; break; }
                    if (cur == HK_EMPTY) {
                        long long expected = HK_EMPTY;
                        if (__hip_atomic_compare_exchange_strong(&table[s], &expected, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            there = added = true;
                            break;
                        }
                        if (expected == key) { there = true; break; }
                    }
                    s = (s + 1) & mask;
                    if (++probes > mask) { flags[1] = 1u; break; }   // every entry holds another key
                    if ((probes & 63u) == 0u && __hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                }
                if (there) s_seen[c] = key;   // (an aligned 8-byte LDS store: readers see the old or the new key, never a mix)
            }
        }
        const unsigned long long m = __ballot(added);
        if (m && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)m) - 1u) {
            const uint32_t k = (uint32_t)__popcll(m);
            if (atomicAdd(&flags[2], k) + k > limit) flags[1] = 1u;
        }
        if ((++round & 7u) == 0u && __hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
}

__global__ __launch_bounds__(OP_THREADS) void k_hk_mark(const long long* __restrict__ table, uint32_t cap, uint32_t* __restrict__ mark /*[cap + 1]*/) {
    for (uint32_t s = blockIdx.x * OP_THREADS + threadIdx.x; s <= cap; s += gridDim.x * OP_THREADS) mark[s] = (s < cap && table[s] != HK_EMPTY) ? 1u : 0u;
}

// after the scan: ids[s] = slot of table entry s, ids[cap] = number of occupied entries
__global__ __launch_bounds__(OP_THREADS) void k_hk_keys(const long long* __restrict__ table, const uint32_t* __restrict__ ids, uint32_t cap, uint32_t n_groups,
                                                        const uint32_t* __restrict__ flags, long long* __restrict__ out_keys, uint32_t* __restrict__ out_n) {
    const uint32_t used = ids[cap];
    for (uint32_t s = blockIdx.x * OP_THREADS + threadIdx.x; s < cap; s += gridDim.x * OP_THREADS) {
        const long long k = table[s];
        if (k != HK_EMPTY && ids[s] < n_groups && out_keys) out_keys[ids[s]] = k;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t total = used + (flags[0] ? 1u : 0u);
        if (flags[0] && used < n_groups && out_keys) out_keys[used] = HK_EMPTY;   // the key INT64_MIN owns the slot behind the table's
        if (out_n) *out_n = flags[1] ? 0xFFFFFFFFu : total;
    }
}

template <typename KeyT>
__global__ __launch_bounds__(OP_THREADS) void k_hk_lookup(const KeyT* __restrict__ keys, unsigned long long n, const long long* __restrict__ table,
                                                          const uint32_t* __restrict__ ids, uint32_t mask, uint32_t cap, uint32_t n_groups,
                                                          uint32_t* __restrict__ slot_out, const uint32_t* __restrict__ flags) {
    // direct-mapped cache of (key, id) in LDS.  An entry is two words, so it carries a version (a seqlock): a writer takes the
    // entry by moving the version from even to odd with ONE compare-and-swap (lanes of one wave that want the same entry are
    // serialised by the LDS atomic: one wins, the others go without caching), stores key and id, and releases it with version + 2;
    // a reader uses the entry only if the version is even and unchanged around its two reads.  (The first form published
    // "key = EMPTY, id, key" with plain stores from every lane: two lanes of one wave with different keys for the same entry
    // write in the same instructions, and the 8-byte key store and the 4-byte id store need not pick the same winner.)
    __shared__ volatile long long s_key[HK_CACHE];
    __shared__ volatile uint32_t s_id[HK_CACHE];
    __shared__ uint32_t s_ver[HK_CACHE];
    for (uint32_t i = threadIdx.x; i < HK_CACHE; i += OP_THREADS) { s_key[i] = HK_EMPTY; s_ver[i] = 0; }
    __syncthreads();
    if (flags[1]) {   // the table overflowed (k_hk_insert): keys may be missing from it, the call's outputs are undefined -- no probing
        for (unsigned long long i = (unsigned long long)blockIdx.x * OP_THREADS + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * OP_THREADS) slot_out[i] = 0;
        return;
    }
    const uint32_t used = ids[cap];
    for (unsigned long long i = (unsigned long long)blockIdx.x * OP_THREADS + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * OP_THREADS) {
        const long long key = (long long)keys[i];
        uint32_t id;
        if (key == HK_EMPTY) {
            id = used;
        } else {
            const uint32_t h = hk_hash(key, mask);
            const uint32_t c = h & (HK_CACHE - 1);
            const uint32_t v1 = __hip_atomic_load(&s_ver[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const long long ck = s_key[c];
            const uint32_t cid = s_id[c];
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t v2 = __hip_atomic_load(&s_ver[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (v1 == v2 && !(v1 & 1u) && ck == key) {
                id = cid;
            } else {
                uint32_t s = h, probes = 0;
                while (table[s] != key && probes <= mask) { s = (s + 1) & mask; ++probes; }   // (the key is there unless the table overflowed)
                id = ids[s];
                if (!(v2 & 1u) && atomicCAS(&s_ver[c], v2, v2 + 1u) == v2) {
                    s_key[c] = key;
                    s_id[c] = id;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __hip_atomic_store(&s_ver[c], v2 + 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        slot_out[i] = id < n_groups ? id : id % n_groups;   // (ids beyond the bound only occur when the bound was too small: results are void, stay in range)
    }
}

static int32_t aggregate_hashed(golap_ctx* ctx, const golap_aggregate_spec* specs, uint32_t n_specs, const golap_aggregate_params* p, hipStream_t st) {
    if (!p->group_key_ptr && p->n_rows) return fail(ctx, GOLAP_ERR_INVALID_ARG, "GOLAP_KEYS_HASHED needs a group key column");
    // (round 3: the scan over the key table nests, so the bound is the table's memory -- 2^28 groups = a 4 GiB table -- not 2^22)
    if (p->n_groups > (1u << 28)) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "GOLAP_KEYS_HASHED: at most 2^28 groups (key table of 2^29 entries)");
    if (p->n_rows >= 0xFFFFFFF0ull) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "GOLAP_KEYS_HASHED: at most 2^32-16 rows");
    uint32_t cap = 1024;
    while (cap < 2 * p->n_groups) cap <<= 1;   // load factor <= 0.5 while the bound holds
    const uint32_t mask = cap - 1;
    const uint64_t n = p->n_rows;
    GOLAP_TRY(aux_begin(ctx, round_up((uint64_t)cap * 8, 256) + round_up(((uint64_t)cap + 1) * 4, 256) + round_up(flat_scan_scratch_words((uint64_t)cap + 1) * 4, 256) +
                                 round_up(std::max<uint64_t>(n, 1) * 4, 256) + 8192, st));
    long long* table = (long long*)aux_take(ctx, (uint64_t)cap * 8);
    uint32_t* ids = (uint32_t*)aux_take(ctx, ((uint64_t)cap + 1) * 4);
    uint32_t* bsums = (uint32_t*)aux_take(ctx, flat_scan_scratch_words((uint64_t)cap + 1) * 4);
    // (the slot column goes straight into the caller's buffer when it wants it: out_slot_ids_ptr)
    uint32_t* slots = p->out_slot_ids_ptr ? (uint32_t*)p->out_slot_ids_ptr : (uint32_t*)aux_take(ctx, std::max<uint64_t>(n, 1) * 4);
    uint32_t* flags = (uint32_t*)aux_take(ctx, 16);
    if (!table || !ids || !bsums || !slots || !flags) return fail(ctx, GOLAP_ERR_OOM, "aggregate scratch exhausted (key table)");
    const uint32_t cus = (uint32_t)ctx->prop.multiProcessorCount;
    const uint32_t tgrid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)cus * 8, ceil_div((uint64_t)cap + 1, OP_THREADS)));
    const uint32_t rgrid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)cus * 8, ceil_div(std::max<uint64_t>(n, 1), OP_THREADS * 8)));
    const uint64_t key_bytes = p->key_dtype == GOLAP_DT_INT32 ? 4 : 8;
    GOLAP_HIP_TRY(ctx, hipMemsetAsync(flags, 0, 16, st));
    {   // every entry = HK_EMPTY (0x80 in the top byte only: a kernel, not a byte memset)
        KernelTimer t(ctx, "hash_keys_insert", st, key_bytes * n);
        hipLaunchKernelGGL(k_fill_i64, dim3(tgrid), dim3(OP_THREADS), 0, st, table, (unsigned long long)cap, HK_EMPTY);
        if (n) {
            if (p->key_dtype == GOLAP_DT_INT32)
                hipLaunchKernelGGL(k_hk_insert<int>, dim3(rgrid), dim3(OP_THREADS), 0, st, (const int*)p->group_key_ptr, (unsigned long long)n, table, mask, flags);
            else
                hipLaunchKernelGGL(k_hk_insert<long long>, dim3(rgrid), dim3(OP_THREADS), 0, st, (const long long*)p->group_key_ptr, (unsigned long long)n, table, mask, flags);
        }
    }
    hipLaunchKernelGGL(k_hk_mark, dim3(tgrid), dim3(OP_THREADS), 0, st, (const long long*)table, cap, ids);
    GOLAP_TRY(flat_scan(ctx, st, ids, cap + 1, bsums));
    hipLaunchKernelGGL(k_hk_keys, dim3(tgrid), dim3(OP_THREADS), 0, st, (const long long*)table, (const uint32_t*)ids, cap, p->n_groups, (const uint32_t*)flags,
                       (long long*)p->out_keys_ptr, (uint32_t*)p->out_n_groups_ptr);
    if (n) {
        KernelTimer t(ctx, "hash_keys_lookup", st, (key_bytes + 4) * n);
        if (p->key_dtype == GOLAP_DT_INT32)
            hipLaunchKernelGGL(k_hk_lookup<int>, dim3(rgrid), dim3(OP_THREADS), 0, st, (const int*)p->group_key_ptr, (unsigned long long)n, (const long long*)table,
                               (const uint32_t*)ids, mask, cap, p->n_groups, slots, (const uint32_t*)flags);
        else
            hipLaunchKernelGGL(k_hk_lookup<long long>, dim3(rgrid), dim3(OP_THREADS), 0, st, (const long long*)p->group_key_ptr, (unsigned long long)n,
                               (const long long*)table, (const uint32_t*)ids, mask, cap, p->n_groups, slots, (const uint32_t*)flags);
    }
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    golap_aggregate_params q = *p;
    q.group_key_ptr = (uint64_t)slots;
    q.key_dtype = GOLAP_DT_INT32;
    q.key_mode = GOLAP_KEYS_DENSE;
    const int32_t rc = aggregate_dense(ctx, specs, n_specs, &q, st);
    aux_end(ctx, st);
    return rc;
}

}  // namespace golap

extern "C" {

int32_t golap_widen_to_8byte(golap_ctx* ctx, uint64_t in_ptr, int32_t in_dtype, uint64_t validity_ptr, uint64_t n_rows, uint64_t out_ptr, void* stream) {
    if (!ctx) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null ctx");
    if (n_rows == 0) return GOLAP_OK;
    if (!in_ptr || !out_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick_stream(ctx, stream);
    // (one element per thread and round: this kernel wants MANY work-groups -- 1B int32 -> int64: 2.55 ms with 8 per CU, 2.37 with 32, 5.2 with 2)
    const dim3 g(stream_grid(ctx, n_rows, OP_THREADS * 4, 32)), b(OP_THREADS);
    const uint8_t* v = (const uint8_t*)validity_ptr;
    const unsigned long long n = n_rows;
    KernelTimer t(ctx, "widen_to_8byte", st, 9 * n_rows);
    switch (in_dtype) {
        case GOLAP_WIDEN_I8: hipLaunchKernelGGL((k_widen<int8_t, long long>), g, b, 0, st, (const int8_t*)in_ptr, v, n, (long long*)out_ptr); break;
        case GOLAP_WIDEN_I16: hipLaunchKernelGGL((k_widen<int16_t, long long>), g, b, 0, st, (const int16_t*)in_ptr, v, n, (long long*)out_ptr); break;
        case GOLAP_WIDEN_I32: hipLaunchKernelGGL((k_widen<int32_t, long long>), g, b, 0, st, (const int32_t*)in_ptr, v, n, (long long*)out_ptr); break;
        case GOLAP_WIDEN_I64: hipLaunchKernelGGL((k_widen<long long, long long>), g, b, 0, st, (const long long*)in_ptr, v, n, (long long*)out_ptr); break;
        case GOLAP_WIDEN_U8: hipLaunchKernelGGL((k_widen<uint8_t, long long>), g, b, 0, st, (const uint8_t*)in_ptr, v, n, (long long*)out_ptr); break;
        case GOLAP_WIDEN_U16: hipLaunchKernelGGL((k_widen<uint16_t, long long>), g, b, 0, st, (const uint16_t*)in_ptr, v, n, (long long*)out_ptr); break;
        case GOLAP_WIDEN_U32: hipLaunchKernelGGL((k_widen<uint32_t, long long>), g, b, 0, st, (const uint32_t*)in_ptr, v, n, (long long*)out_ptr); break;
        case GOLAP_WIDEN_U64: hipLaunchKernelGGL((k_widen<unsigned long long, long long>), g, b, 0, st, (const unsigned long long*)in_ptr, v, n, (long long*)out_ptr); break;
        case GOLAP_WIDEN_F32: hipLaunchKernelGGL((k_widen<float, double>), g, b, 0, st, (const float*)in_ptr, v, n, (double*)out_ptr); break;
        case GOLAP_WIDEN_F64: hipLaunchKernelGGL((k_widen<double, double>), g, b, 0, st, (const double*)in_ptr, v, n, (double*)out_ptr); break;
        default: return fail(ctx, GOLAP_ERR_INVALID_ARG, "unsupported column type for GPU conversion: %d", in_dtype);   // record_batch_convert.rs:98
    }
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

int32_t golap_hash_utf8(golap_ctx* ctx, uint64_t offsets_ptr, uint64_t data_ptr, uint64_t validity_ptr, uint64_t n_rows, uint64_t out_ptr, void* stream) {
    if (!ctx) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null ctx");
    if (n_rows == 0) return GOLAP_OK;
    if (!offsets_ptr || !out_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick_stream(ctx, stream);
    KernelTimer t(ctx, "hash_utf8", st, 16 * n_rows);
    hipLaunchKernelGGL(k_hash_utf8, dim3(stream_grid(ctx, n_rows, OP_THREADS)), dim3(OP_THREADS), 0, st, (const long long*)offsets_ptr, (const uint8_t*)data_ptr,
                       (const uint8_t*)validity_ptr, (unsigned long long)n_rows, (long long*)out_ptr);
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

int32_t golap_generate(golap_ctx* ctx, int32_t kind, uint64_t seed, uint64_t first, uint64_t count, uint64_t modulus, uint64_t out_ptr, void* stream) {
    if (!ctx) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null ctx");
    if (count == 0) return GOLAP_OK;
    if (!out_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null out_ptr");
    if (kind < GOLAP_GEN_BUILD_KEYS || kind > GOLAP_GEN_SHIFT32_I64) return fail(ctx, GOLAP_ERR_INVALID_ARG, "bad generator kind %d", kind);
    if (kind != GOLAP_GEN_SHIFT32_I64 && modulus == 0) return fail(ctx, GOLAP_ERR_INVALID_ARG, "modulus must be > 0");
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick_stream(ctx, stream);
    hipLaunchKernelGGL(k_generate, dim3(stream_grid(ctx, count, OP_THREADS * 4)), dim3(OP_THREADS), 0, st, kind, (unsigned long long)seed,
                       (unsigned long long)first, (unsigned long long)count, (unsigned long long)modulus, (void*)out_ptr);
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

int32_t golap_verify_pairs(golap_ctx* ctx, uint64_t pairs_ptr, uint64_t n_pairs, uint64_t build_key_ptr, uint64_t probe_key_ptr, uint32_t build_rid_base,
                           uint32_t probe_rid_base, uint64_t out_ptr, void* stream) {
    if (!ctx || !out_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    if (n_pairs && (!pairs_ptr || !build_key_ptr || !probe_key_ptr)) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick_stream(ctx, stream);
    GOLAP_TRY(zero_async(ctx, st, (void*)((void*)out_ptr), 32));
    if (n_pairs == 0) return GOLAP_OK;
    hipLaunchKernelGGL(k_verify_pairs, dim3(stream_grid(ctx, n_pairs, OP_THREADS * 4)), dim3(OP_THREADS), 0, st, (const uint2*)pairs_ptr,
                       (unsigned long long)n_pairs, (const long long*)build_key_ptr, (const long long*)probe_key_ptr, build_rid_base, probe_rid_base,
                       (unsigned long long*)out_ptr);
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

}  // extern "C"
