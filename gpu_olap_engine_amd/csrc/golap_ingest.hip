// golap_ingest.hip -- the device half of the operator layer's plumbing: what stands between Arrow buffers in HBM and the
// operators' column layout (8-byte values + one validity byte per row), and between the operators' outputs and Arrow again.
//
//   arrow-interop/src/record_batch_convert.rs:35-110   column_to_buffer: validity = is_valid(i) per row (:36-40), widening, null -> 0
//   arrow-interop/src/record_batch_convert.rs:146-178  buffer_to_array: append_value / append_null by the validity byte
//   gpu-olap-core/src/executor.rs:478-507               filter_record_batch / take_int64: gathers of values (and, here, validity)
//
// The reference does all of it in per-value host loops; here the raw Arrow buffers are uploaded as they are (values at their own
// width, the validity BITMAP at one bit per row) and everything else happens where the data is.  All kernels are plain HBM streams or
// gathers; none is on the join's critical path.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "golap_internal.h"

namespace golap {

constexpr int IG_THREADS = 256;

static uint32_t ig_grid(golap_ctx* ctx, uint64_t items, uint64_t per_cu = 16) {
    const uint64_t want = ceil_div(std::max<uint64_t>(1, items), IG_THREADS);
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)ctx->prop.multiProcessorCount * per_cu));
}

// Arrow validity bitmap (bit (bit_offset + i) of `bitmap`, LSB first) -> one byte per row (column_buffer.rs:59-60: 1 valid, 0 NULL).
// A thread expands 8 consecutive rows from (at most) two bitmap bytes into one 8-byte store.
__global__ __launch_bounds__(IG_THREADS) void k_expand_validity(const uint8_t* __restrict__ bitmap, unsigned long long bit_offset, unsigned long long n,
                                                                uint8_t* __restrict__ out) {
    const unsigned long long n8 = (n + 7) / 8, last_byte = (bit_offset + n - 1) >> 3;
    for (unsigned long long t = (unsigned long long)blockIdx.x * IG_THREADS + threadIdx.x; t < n8; t += (unsigned long long)gridDim.x * IG_THREADS) {
        const unsigned long long r = t * 8, o = bit_offset + r, b = o >> 3;
        uint32_t bits = bitmap[b];
        if (b < last_byte) bits |= (uint32_t)bitmap[b + 1] << 8;
        bits >>= (uint32_t)(o & 7);
        if (r + 8 <= n && (((uintptr_t)out) & 7) == 0) {
            unsigned long long w = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) w |= (unsigned long long)((bits >> k) & 1u) << (8 * k);
            *(unsigned long long*)(out + r) = w;
        } else {
            for (unsigned long long k = 0; r + k < n && k < 8; ++k) out[r + k] = (uint8_t)((bits >> k) & 1u);
        }
    }
}

// one byte per row -> Arrow validity bitmap (bit i of out, LSB first; the last byte's unused bits are 0) + the NULL count
__global__ __launch_bounds__(IG_THREADS) void k_pack_validity(const uint8_t* __restrict__ valid, unsigned long long n, uint8_t* __restrict__ out,
                                                              unsigned long long* __restrict__ null_count) {
    const unsigned long long n8 = (n + 7) / 8;
    unsigned long long nulls = 0;
    for (unsigned long long t = (unsigned long long)blockIdx.x * IG_THREADS + threadIdx.x; t < n8; t += (unsigned long long)gridDim.x * IG_THREADS) {
        const unsigned long long r = t * 8;
        uint32_t bits = 0;
        if (r + 8 <= n && (((uintptr_t)valid) & 7) == 0) {
            const unsigned long long w = *(const unsigned long long*)(valid + r);
#pragma unroll
            for (int k = 0; k < 8; ++k) bits |= (uint32_t)(((w >> (8 * k)) & 0xFFull) != 0) << k;
            nulls += 8u - (uint32_t)__popc(bits);
        } else {
            for (unsigned long long k = 0; r + k < n && k < 8; ++k) {
                const uint32_t v = valid[r + k] != 0;
                bits |= v << k;
                nulls += 1u - v;
            }
        }
        out[t] = (uint8_t)bits;
    }
    for (int d = 32; d >= 1; d >>= 1) nulls += __shfl_xor(nulls, d, 64);
    if ((threadIdx.x & 63) == 0 && nulls) atomicAdd(null_count, nulls);
}

// validity of gathered rows: out[i] = 0 for the NULL row id (the missing side of an outer-join row), else src[idx[i]] (src == null:
// the source column has no NULLs -> 1).  The 1-byte companion of take_int64 (executor.rs:496-507).
__global__ __launch_bounds__(IG_THREADS) void k_take1(const uint8_t* __restrict__ src, const uint32_t* __restrict__ idx, const unsigned long long* __restrict__ count_ptr,
                                                      unsigned long long n_max, uint8_t* __restrict__ out) {
    unsigned long long n = count_ptr ? *count_ptr : n_max;
    if (n > n_max) n = n_max;
    for (unsigned long long i = (unsigned long long)blockIdx.x * IG_THREADS + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * IG_THREADS) {
        const uint32_t j = idx[i];
        out[i] = j == 0xFFFFFFFFu ? (uint8_t)0 : (src ? (uint8_t)(src[j] != 0) : (uint8_t)1);
    }
}

// (build_rid, probe_rid) pairs -> one index vector per side (left_matches / right_matches, join_kernels.cuh:125-126)
__global__ __launch_bounds__(IG_THREADS) void k_unzip_pairs(const uint2* __restrict__ pairs, unsigned long long n, uint32_t* __restrict__ b, uint32_t* __restrict__ p) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * IG_THREADS + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * IG_THREADS) {
        const uint2 v = pairs[i];
        b[i] = v.x;
        p[i] = v.y;
    }
}

__global__ void k_minmax_init(long long* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = INT64_MAX; out[1] = INT64_MIN; out[2] = 0; }
}
// exact int64 minimum / maximum / count of the valid rows of a key column (a double would lose keys beyond 2^53)
__global__ __launch_bounds__(IG_THREADS) void k_col_minmax(const long long* __restrict__ col, const uint8_t* __restrict__ valid, unsigned long long n,
                                                           long long* __restrict__ out) {
    long long mn = INT64_MAX, mx = INT64_MIN;
    unsigned long long cnt = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * IG_THREADS + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * IG_THREADS)
        if (!valid || valid[i]) { const long long v = col[i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; ++cnt; }
    for (int d = 32; d >= 1; d >>= 1) {
        const long long a = __shfl_xor(mn, d, 64), b = __shfl_xor(mx, d, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
        cnt += __shfl_xor(cnt, d, 64);
    }
    if ((threadIdx.x & 63) == 0 && cnt) {
        __hip_atomic_fetch_min(&out[0], mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(&out[1], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd((unsigned long long*)&out[2], cnt);
    }
}

// mixed-radix composite GROUP BY key: sum over the columns of digit * weight, digit = value - min (valid rows) or the column's NULL digit
__global__ __launch_bounds__(IG_THREADS) void k_pack_group_keys(golap_pack_keys_params p, long long* __restrict__ out) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * IG_THREADS + threadIdx.x; i < p.n_rows; i += (unsigned long long)gridDim.x * IG_THREADS) {
        unsigned long long key = 0;
        for (uint32_t c = 0; c < p.n_cols; ++c) {
            const uint8_t* v = (const uint8_t*)p.validity_ptrs[c];
            const unsigned long long dg = (v && !v[i]) ? p.null_digit[c] : (unsigned long long)(((const long long*)p.col_ptrs[c])[i] - p.min_value[c]);
            key += dg * p.weight[c];
        }
        out[i] = (long long)key;
    }
}

__device__ __forceinline__ unsigned long long ig_mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// 64-bit surrogate of a composite GROUP BY key: a seeded chain of mixes over the columns; a NULL is its own symbol per column
// keep_mask: all ones; the test knob GOLAP_SURROGATE_BITS narrows the surrogates until two tuples do share one
__global__ __launch_bounds__(IG_THREADS) void k_hash_key_columns(golap_pack_keys_params p, unsigned long long seed, unsigned long long keep_mask, long long* __restrict__ out) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * IG_THREADS + threadIdx.x; i < p.n_rows; i += (unsigned long long)gridDim.x * IG_THREADS) {
        unsigned long long h = ig_mix64(seed ^ 0x243F6A8885A308D3ull);
        for (uint32_t c = 0; c < p.n_cols; ++c) {
            const uint8_t* v = (const uint8_t*)p.validity_ptrs[c];
            const bool null = v && !v[i];
            const unsigned long long x = null ? 0x9E3779B97F4A7C15ull * (c + 1) : (unsigned long long)((const long long*)p.col_ptrs[c])[i];
            h = ig_mix64(h ^ x) + (null ? 0xD1B54A32D192ED03ull : 0x8CB92BA72F3D8DD7ull) * (c + 1);
        }
        out[i] = (long long)(h & keep_mask);
    }
}
__global__ __launch_bounds__(IG_THREADS) void k_fill_u32(uint32_t* __restrict__ p, uint32_t v, unsigned long long n) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * IG_THREADS + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * IG_THREADS) p[i] = v;
}
// representative row of a slot = the smallest row index aggregated into it
__global__ __launch_bounds__(IG_THREADS) void k_slot_rep(const uint32_t* __restrict__ slot, unsigned long long n, unsigned long long n_slots, uint32_t* __restrict__ rep) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * IG_THREADS + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * IG_THREADS) {
        const uint32_t s = slot[i];
        if (s < n_slots && rep[s] > (uint32_t)i) atomicMin(&rep[s], (uint32_t)i);
    }
}
// rows whose key tuple differs from their slot's representative (NULL == NULL here: GROUP BY puts NULLs of a column together)
__global__ __launch_bounds__(IG_THREADS) void k_group_keys_verify(golap_pack_keys_params p, const uint32_t* __restrict__ slot, unsigned long long n_slots,
                                                                  const uint32_t* __restrict__ rep, unsigned long long* __restrict__ mismatches) {
    unsigned long long bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * IG_THREADS + threadIdx.x; i < p.n_rows; i += (unsigned long long)gridDim.x * IG_THREADS) {
        const uint32_t s = slot[i];
        if (s >= n_slots) { ++bad; continue; }
        const uint32_t r = rep[s];
        bool same = true;
        for (uint32_t c = 0; c < p.n_cols; ++c) {
            const uint8_t* v = (const uint8_t*)p.validity_ptrs[c];
            const bool ni = v && !v[i], nr = v && !v[r];
            const long long* col = (const long long*)p.col_ptrs[c];
            same = same && ni == nr && (ni || col[i] == col[r]);
        }
        bad += same ? 0 : 1;
    }
    for (int d = 32; d >= 1; d >>= 1) bad += __shfl_xor(bad, d, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(mismatches, bad);
}

__global__ __launch_bounds__(IG_THREADS) void k_fill_u8(uint8_t* __restrict__ p, uint8_t v, unsigned long long n) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * IG_THREADS + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * IG_THREADS) p[i] = v;
}

}  // namespace golap

using namespace golap;

#define IG_ENTER(ctx, stream)                                              \
    if (!(ctx)) return fail((ctx), GOLAP_ERR_INVALID_ARG, "null ctx");     \
    std::lock_guard<std::mutex> lk((ctx)->mu);                             \
    KnobScope knobs_(ctx);                                                 \
    GOLAP_HIP_TRY((ctx), hipSetDevice((ctx)->device));                     \
    hipStream_t st = pick_stream((ctx), (stream))

extern "C" {

int32_t golap_expand_validity(golap_ctx* ctx, uint64_t bitmap_ptr, uint64_t bit_offset, uint64_t n_rows, uint64_t out_ptr, void* stream) {
    IG_ENTER(ctx, stream);
    if (n_rows == 0) return GOLAP_OK;
    if (!bitmap_ptr || !out_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null pointer");
    KernelTimer t(ctx, "expand_validity", st, n_rows + n_rows / 8);
    hipLaunchKernelGGL(k_expand_validity, dim3(ig_grid(ctx, (n_rows + 7) / 8)), dim3(IG_THREADS), 0, st, (const uint8_t*)bitmap_ptr, (unsigned long long)bit_offset,
                       (unsigned long long)n_rows, (uint8_t*)out_ptr);
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

int32_t golap_pack_validity(golap_ctx* ctx, uint64_t validity_ptr, uint64_t n_rows, uint64_t out_bitmap_ptr, uint64_t out_null_count_ptr, void* stream) {
    IG_ENTER(ctx, stream);
    if (!out_null_count_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "out_null_count_ptr is null");
    GOLAP_TRY(zero_async(ctx, st, (void*)out_null_count_ptr, 8));
    if (n_rows == 0) return GOLAP_OK;
    if (!validity_ptr || !out_bitmap_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null pointer");
    KernelTimer t(ctx, "pack_validity", st, n_rows + n_rows / 8);
    hipLaunchKernelGGL(k_pack_validity, dim3(ig_grid(ctx, (n_rows + 7) / 8)), dim3(IG_THREADS), 0, st, (const uint8_t*)validity_ptr, (unsigned long long)n_rows,
                       (uint8_t*)out_bitmap_ptr, (unsigned long long*)out_null_count_ptr);
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

int32_t golap_take_1byte(golap_ctx* ctx, uint64_t src_ptr, uint64_t indices_ptr, uint64_t count_ptr, uint64_t n_indices_max, uint64_t out_ptr, void* stream) {
    IG_ENTER(ctx, stream);
    if (n_indices_max == 0) return GOLAP_OK;
    if (!indices_ptr || !out_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null pointer");
    KernelTimer t(ctx, "take_1byte", st, 6 * n_indices_max);
    hipLaunchKernelGGL(k_take1, dim3(ig_grid(ctx, n_indices_max)), dim3(IG_THREADS), 0, st, (const uint8_t*)src_ptr, (const uint32_t*)indices_ptr,
                       (const unsigned long long*)count_ptr, (unsigned long long)n_indices_max, (uint8_t*)out_ptr);
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

int32_t golap_pairs_unzip(golap_ctx* ctx, uint64_t pairs_ptr, uint64_t n_pairs, uint64_t out_build_idx_ptr, uint64_t out_probe_idx_ptr, void* stream) {
    IG_ENTER(ctx, stream);
    if (n_pairs == 0) return GOLAP_OK;
    if (!pairs_ptr || !out_build_idx_ptr || !out_probe_idx_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null pointer");
    KernelTimer t(ctx, "pairs_unzip", st, 16 * n_pairs);
    hipLaunchKernelGGL(k_unzip_pairs, dim3(ig_grid(ctx, n_pairs)), dim3(IG_THREADS), 0, st, (const uint2*)pairs_ptr, (unsigned long long)n_pairs,
                       (uint32_t*)out_build_idx_ptr, (uint32_t*)out_probe_idx_ptr);
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

// Ascending in-place sort of n 64-bit words.  A pair (build_rid, probe_rid) read as ONE little-endian 64-bit word is probe_rid << 32 |
// build_rid, so this IS the canonical (probe_rid, build_rid) order of SURVEY 8(c) -- the NULL row id 0xFFFFFFFF last.  Presentation
// only (the kernels emit pairs in atomic order, join_kernels.cuh:155): rocPRIM's radix sort, scratch from the context's aux slab.
int32_t golap_sort_u64(golap_ctx* ctx, uint64_t ptr, uint64_t n, void* stream) {
    IG_ENTER(ctx, stream);
    if (n < 2) return GOLAP_OK;
    if (!ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null pointer");
    if (n >= (1ull << 32)) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "golap_sort_u64: at most 2^32 - 1 words");
    unsigned long long* keys = (unsigned long long*)ptr;
    size_t tmp_bytes = 0;
    rocprim::double_buffer<unsigned long long> probe(keys, keys);
    if (rocprim::radix_sort_keys(nullptr, tmp_bytes, probe, (size_t)n, 0, 64, st) != hipSuccess) return fail(ctx, GOLAP_ERR_HIP, "rocprim::radix_sort_keys (size query) failed");
    GOLAP_TRY(aux_begin(ctx, round_up(n * 8, 256) + round_up(tmp_bytes, 256) + 4096, st));
    unsigned long long* alt = (unsigned long long*)aux_take(ctx, n * 8);
    void* tmp = (void*)aux_take(ctx, std::max<size_t>(tmp_bytes, 8));
    if (!alt || !tmp) { aux_end(ctx, st); return fail(ctx, GOLAP_ERR_OOM, "golap_sort_u64: scratch"); }
    rocprim::double_buffer<unsigned long long> db(keys, alt);
    {
        KernelTimer t(ctx, "sort_u64", st, 8 * 16 * n);
        const hipError_t e = rocprim::radix_sort_keys(tmp, tmp_bytes, db, (size_t)n, 0, 64, st);
        if (e != hipSuccess) { aux_end(ctx, st); return fail(ctx, GOLAP_ERR_HIP, "rocprim::radix_sort_keys failed: %s", hipGetErrorString(e)); }
        if (db.current() != keys) {
            const hipError_t c = hipMemcpyAsync(keys, db.current(), n * 8, hipMemcpyDeviceToDevice, st);
            if (c != hipSuccess) { aux_end(ctx, st); return fail(ctx, GOLAP_ERR_HIP, "copy back failed: %s", hipGetErrorString(c)); }
        }
    }
    aux_end(ctx, st);
    return GOLAP_OK;
}

int32_t golap_key_minmax(golap_ctx* ctx, uint64_t column_ptr, uint64_t validity_ptr, uint64_t n_rows, uint64_t out_ptr, void* stream) {
    IG_ENTER(ctx, stream);
    if (!out_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "out_ptr is null");
    if (n_rows && !column_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null column");
    hipLaunchKernelGGL(k_minmax_init, dim3(1), dim3(64), 0, st, (long long*)out_ptr);
    if (n_rows) {
        KernelTimer t(ctx, "key_minmax", st, (validity_ptr ? 9 : 8) * n_rows);
        hipLaunchKernelGGL(k_col_minmax, dim3(ig_grid(ctx, ceil_div(n_rows, 8), 8)), dim3(IG_THREADS), 0, st, (const long long*)column_ptr, (const uint8_t*)validity_ptr,
                           (unsigned long long)n_rows, (long long*)out_ptr);
    }
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

int32_t golap_pack_group_keys(golap_ctx* ctx, const golap_pack_keys_params* p, void* stream) {
    IG_ENTER(ctx, stream);
    if (!p) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null params");
    if (p->n_cols < 1 || p->n_cols > 4) return fail(ctx, GOLAP_ERR_INVALID_ARG, "1 to 4 key columns");
    if (p->n_rows == 0) return GOLAP_OK;
    if (!p->out_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "out_ptr is null");
    for (uint32_t c = 0; c < p->n_cols; ++c)
        if (!p->col_ptrs[c]) return fail(ctx, GOLAP_ERR_INVALID_ARG, "key column %u is null", c);
    KernelTimer t(ctx, "pack_group_keys", st, (8ull * p->n_cols + 8) * p->n_rows);
    hipLaunchKernelGGL(k_pack_group_keys, dim3(ig_grid(ctx, p->n_rows)), dim3(IG_THREADS), 0, st, *p, (long long*)p->out_ptr);
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

int32_t golap_hash_key_columns(golap_ctx* ctx, const golap_pack_keys_params* p, uint64_t seed, void* stream) {
    IG_ENTER(ctx, stream);
    if (!p) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null params");
    if (p->n_cols < 1 || p->n_cols > 4) return fail(ctx, GOLAP_ERR_INVALID_ARG, "1 to 4 key columns");
    if (p->n_rows == 0) return GOLAP_OK;
    if (!p->out_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "out_ptr is null");
    for (uint32_t c = 0; c < p->n_cols; ++c)
        if (!p->col_ptrs[c]) return fail(ctx, GOLAP_ERR_INVALID_ARG, "key column %u is null", c);
    KernelTimer t(ctx, "hash_key_columns", st, (8ull * p->n_cols + 8) * p->n_rows);
    unsigned long long keep = ~0ull;
    if (const char* e = knob("GOLAP_SURROGATE_BITS")) { const int b = atoi(e); if (b >= 1 && b < 64) keep = (1ull << b) - 1; }
    hipLaunchKernelGGL(k_hash_key_columns, dim3(ig_grid(ctx, p->n_rows)), dim3(IG_THREADS), 0, st, *p, (unsigned long long)seed, keep, (long long*)p->out_ptr);
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

int32_t golap_group_keys_verify(golap_ctx* ctx, const golap_pack_keys_params* p, uint64_t slot_ids_ptr, uint64_t n_slots, uint64_t out_rep_rows_ptr,
                                uint64_t out_mismatch_ptr, void* stream) {
    IG_ENTER(ctx, stream);
    if (!p || !out_mismatch_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    if (p->n_cols < 1 || p->n_cols > 4) return fail(ctx, GOLAP_ERR_INVALID_ARG, "1 to 4 key columns");
    if (p->n_rows >= 0xFFFFFFFFull) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "row indices are u32");
    GOLAP_TRY(zero_async(ctx, st, (void*)out_mismatch_ptr, 8));
    if (p->n_rows == 0 || n_slots == 0) return GOLAP_OK;
    if (!slot_ids_ptr || !out_rep_rows_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null pointer");
    for (uint32_t c = 0; c < p->n_cols; ++c)
        if (!p->col_ptrs[c]) return fail(ctx, GOLAP_ERR_INVALID_ARG, "key column %u is null", c);
    KernelTimer t(ctx, "group_keys_verify", st, (16ull * p->n_cols + 8) * p->n_rows);
    hipLaunchKernelGGL(k_fill_u32, dim3(ig_grid(ctx, n_slots)), dim3(IG_THREADS), 0, st, (uint32_t*)out_rep_rows_ptr, 0xFFFFFFFFu, (unsigned long long)n_slots);
    hipLaunchKernelGGL(k_slot_rep, dim3(ig_grid(ctx, p->n_rows)), dim3(IG_THREADS), 0, st, (const uint32_t*)slot_ids_ptr, (unsigned long long)p->n_rows,
                       (unsigned long long)n_slots, (uint32_t*)out_rep_rows_ptr);
    hipLaunchKernelGGL(k_group_keys_verify, dim3(ig_grid(ctx, p->n_rows)), dim3(IG_THREADS), 0, st, *p, (const uint32_t*)slot_ids_ptr, (unsigned long long)n_slots,
                       (const uint32_t*)out_rep_rows_ptr, (unsigned long long*)out_mismatch_ptr);
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

int32_t golap_device_copy(golap_ctx* ctx, uint64_t src_ptr, uint64_t dst_ptr, uint64_t bytes, void* stream) {
    IG_ENTER(ctx, stream);
    if (bytes == 0) return GOLAP_OK;
    if (!src_ptr || !dst_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null pointer");
    GOLAP_HIP_TRY(ctx, hipMemcpyAsync((void*)dst_ptr, (const void*)src_ptr, bytes, hipMemcpyDeviceToDevice, st));
    return GOLAP_OK;
}

int32_t golap_device_fill_u8(golap_ctx* ctx, uint64_t ptr, uint32_t value, uint64_t bytes, void* stream) {
    IG_ENTER(ctx, stream);
    if (bytes == 0) return GOLAP_OK;
    if (!ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null pointer");
    hipLaunchKernelGGL(k_fill_u8, dim3(ig_grid(ctx, bytes)), dim3(IG_THREADS), 0, st, (uint8_t*)ptr, (uint8_t)value, (unsigned long long)bytes);
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

}  // extern "C"
