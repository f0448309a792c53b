// golap_device.h -- device-side helpers shared by the gfx950 kernels (wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace golap {

// join_kernels.cuh:26-35  MurmurHash3 32-bit finalizer.
__device__ __forceinline__ uint32_t hash_int32(int32_t key) {
    uint32_t h = (uint32_t)key;
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

// join_kernels.cuh:37-41  fold hi^lo then fmix32.
__device__ __forceinline__ uint32_t hash_int64(long long key) {
    uint32_t h = (uint32_t)(key ^ (key >> 32));
    return hash_int32((int32_t)h);
}

// Table-slot hash for the LDS / broadcast tables.  Deliberately independent of
// hash_int64: the reference's fold makes every key i<<32 collide,
// and all rows of one sub-partition share their low hash bits anyway.  Results
// do not depend on it (probe compares full keys, join_kernels.cuh:152-153).
__device__ __forceinline__ uint32_t slot_hash(long long key, uint32_t log2_slots) {
    return (uint32_t)(((unsigned long long)key * 0x9E3779B97F4A7C15ull) >> (64u - log2_slots));
}

// splitmix64 finalizer of seed + i*golden  (SURVEY 8(d) generators)
__device__ __host__ __forceinline__ unsigned long long sm64(unsigned long long seed, unsigned long long i) {
    unsigned long long z = seed + i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __host__ __forceinline__ unsigned long long pair_mix(uint32_t b, uint32_t p) {
    unsigned long long z = (((unsigned long long)p << 32) | b) + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// ---- wave64 primitives ------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ uint32_t mbcnt(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    const uint32_t lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += t;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_reduce_add(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ unsigned long long wave_reduce_add64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ double wave_reduce_addf(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ double wave_reduce_minf(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { double o = __shfl_xor(v, d, 64); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ double wave_reduce_maxf(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { double o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
}

// Exclusive prefix of `v` over the lanes of a wave + wave total.  Fast path for
// the 0/1 case (match flags): wavefront ballot + mbcnt, no shuffles.
__device__ __forceinline__ uint32_t wave_exclusive_count(uint32_t v, uint32_t& total) {
    unsigned long long big = __ballot(v > 1u);
    if (big == 0ull) {
        unsigned long long m = __ballot(v != 0u);
        total = (uint32_t)__popcll(m);
        return mbcnt(m);
    }
    uint32_t inc = wave_inclusive_scan(v);
    total = __shfl(inc, 63, 64);
    return inc - v;
}

// Work-group barrier for LDS-only hazards.  __syncthreads() also drains vmcnt(0): every in-flight global
// store and prefetched load would have to complete at each barrier, although the kernels below only hand
// data over through LDS between barriers (their global stores are write-once, their inputs read-only).
// Waiting for the wave's own LDS operations is enough (cdna_hip_programming.md: counted waits, not vmcnt(0)).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- partition pass configuration ------------------------------------------
struct PassCfg {
    uint32_t shift;        // hash bits consumed by earlier passes
    uint32_t F;            // fan-out of this pass (power of two, <= 1024)
    uint32_t S;            // chunk slots per group
    uint32_t mode;         // 0: bucket = (h >> shift) & (F-1)
                           // 1: first pass: reference partition p = h & 255 (join_kernels.cuh:69) laid out
                           //    owner-major, refined by e1 further hash bits
    uint32_t e1;           // extra bits of the first pass (F = 256 << e1)
    uint32_t owners_log2;  // log2(n_owners)
    uint32_t piece_rot_off;   // k_part2_ring: 1 = visit a group's pieces from piece 0 on in every work-group (A/B hook GOLAP_P2R_ROT=0; default: rotated by the group number)
    uint32_t pre_hashed;   // k_part1_narrow / k_part2_ring: the tuples' first word is hash_int32(key32 ^ key_hi) instead of key32 -- a
                           // bijection of the key, so equality, partitions and the join are the same, and later passes need not hash
    const uint32_t* key_hi;   // tuple kernels only, may be null (= 0): the upper 32 bits every key of this join shares.  The tuples carry
                              // the LOWER 32 bits; hash_int64(key) = hash_int32(lo ^ hi) (join_kernels.cuh:37-41), so the partitions stay the
                              // reference's for any common upper half -- not just 0.
                              // Layout: key_hi[0 .. 3] = the DICTIONARY of upper halves (k_key_dict; unused entries repeat entry 0),
                              // key_hi[4] = D, the number of distinct ones found (1 .. 4).  D == 1: as above.  D > 1 (e.g. small negative AND
                              // small positive keys: 0xFFFFFFFF and 0): the tuples' first word is (hash >> b1) | (dictionary index << (32 - b1)),
                              // b1 = hash bits the first pass consumed -- those bits are implied by the bucket from then on, so equality of
                              // the word inside a sub-partition is still equality of the 64-bit key, and the rows stay 8 bytes.
                              // More than 4 upper halves inside a window of F consecutive values (ids with a tenant / shard / epoch prefix:
                              // 0 .. 299, or -1 .. 40): key_hi[4] = D > 4, key_hi[KEY_DICT_BASE_AT] = the window's first value; the index is
                              // (upper half - base), no dictionary is needed, and the hash is mixed per index AFTER fmix32
                              // (k_part1_narrow, DICT = 2).
};
constexpr uint32_t KEY_DICT_MAX = 4;
constexpr uint32_t KEY_DICT_BASE_AT = 8;     // key_hi[8]: first value of the window of upper halves (offset form)
constexpr uint32_t KEY_DICT_WORDS = 12;
__device__ __forceinline__ uint32_t key_hi_of(const PassCfg& c) { return c.key_hi ? *c.key_hi : 0u; }
// 0: one upper half (or none given), 1: 2 .. 4 of them (dictionary, scalar compares), 2: more, inside a window of F values (offset)
__device__ __forceinline__ int key_dict_mode(const PassCfg& c) {
    const uint32_t d = c.key_hi ? c.key_hi[KEY_DICT_MAX] : 1u;
    return d <= 1u ? 0 : d <= KEY_DICT_MAX ? 1 : 2;
}

__device__ __forceinline__ uint32_t bucket_of(uint32_t h, const PassCfg& c) {
    if (c.mode == 0) return (h >> c.shift) & (c.F - 1u);
    const uint32_t p = h & 255u;                                   // join_kernels.cuh:69, radix_shift = 0
    const uint32_t sub = (h >> 8) & ((1u << c.e1) - 1u);
    const uint32_t slot = ((p & ((1u << c.owners_log2) - 1u)) << (8u - c.owners_log2)) | (p >> c.owners_log2);
    return (slot << c.e1) | sub;
}

}  // namespace golap
