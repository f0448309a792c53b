// golap_join.hip -- GpuHashJoin on gfx950: multi-pass LDS-staged radix partitioning
// followed by per-sub-partition chained hash tables held in LDS.
//
// Reference arithmetic restated (paths relative to the reference root):
//   hash            gpu-kernel-compiler/kernels/join_kernels.cuh:26-41
//   radix partition join_kernels.cuh:45-76   (partition = (hash >> shift) & 255)
//   build           join_kernels.cuh:80-111  (chained table, head swap by atomic exchange)
//   probe           join_kernels.cuh:115-166 (walk chain, full-key compare, emit (build_rid, probe_rid))
// What is new (MI355X-first, none of it is in the reference):
//   * the per-row global atomicAdd of :72 is replaced by histogram -> flat exclusive scan -> LDS-staged
//     scatter; three scatter kernels, chosen per pass:
//       k_scatter      4096-row LDS tile, rows regrouped by bucket, contiguous runs       (small fan-outs)
//       k_scatter_big  register-resident 12288/24576-row tile pushed through the LDS window (global first pass)
//       k_scatter_wc   line-aligned write combining: per-bucket carries in LDS, every     (local passes)
//                      128-byte key line written once, whole, by one 16-lane group
//   * partitioning is applied repeatedly (radix_shift, :53, is the hook the reference leaves for that)
//     until a sub-partition's build side fits a 4096-row LDS table; large two-pass joins read no histogram
//     at all: the first pass reserves its bucket ranges with per-tile atomics (RESERVE), the second writes
//     into slot-provisioned sub-partitions (FIXED), each with a device-side overflow flag and the exact
//     histogram path standing by behind it (run_if);
//   * optimistic narrow keys (NARROW): those joins first run on 8-byte (key32, row id) tuples; the first key
//     outside [0, 2^32) flips a device word and the 64-bit pipeline, enqueued behind, does the join instead;
//   * build/probe run on LDS tables (ds atomics), matches are counted first, the output range is
//     reserved with ONE global atomic per work-group batch and the pairs are written with wavefront
//     ballot/prefix compaction -- not one global atomic per match (:155); Left/Right/Full variants;
//   * tile loads are unconditional on clamped indices (a conditional load costs a branch + vmcnt(0)) and
//     barriers wait for LDS only (s_waitcnt lgkmcnt(0); s_barrier), so stores and prefetches stay in flight.
#include "golap_device.h"
#include "golap_internal.h"

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdlib>

namespace golap {

// ---------------------------------------------------------------------------
// constants
// ---------------------------------------------------------------------------
constexpr int PT_THREADS = 512;            // 8 waves
constexpr int PT_RPT = 8;                  // rows per thread per tile
constexpr int PT_TILE = PT_THREADS * PT_RPT;   // 4096 rows staged in LDS (48 KiB)
constexpr int PT_MAX_F = 1024;             // max fan-out per pass (10 bits)
constexpr int PT_WAVES = PT_THREADS / 64;

constexpr int JN_THREADS = 512;
constexpr int JN_WAVES = JN_THREADS / 64;
constexpr int JN_CAP = 4096;               // build rows per LDS table
constexpr int JN_LOG2_SLOTS = 12;          // 4096 chain heads
constexpr int JN_R = 8;                    // probe rows per thread per batch
constexpr uint32_t JN_EMPTY = 0xFFFFFFFFu; // join_kernels.cuh:146 sentinel
static_assert(JN_R * JN_WAVES == 64, "slice x wave totals are scanned by one wave");

// streamed-once inputs: non-temporal loads keep them from displacing the partially written output lines in L2
// (GOLAP_NT=0 at build time restores plain loads for A/B runs)
#ifndef GOLAP_NT
#define GOLAP_NT 0   // measured on MI355X: non-temporal input loads were 0-8 % slower in both scatter kernels
#endif
#if GOLAP_NT
#define NT_LOAD(p) __builtin_nontemporal_load(p)
#else
#define NT_LOAD(p) (*(p))
#endif

constexpr int SC_THREADS = 256;
constexpr int SC_EPT = 16;
constexpr int SC_ELEMS = SC_THREADS * SC_EPT;   // 4096 elements per scan block

// ---------------------------------------------------------------------------
// chunk descriptors
// ---------------------------------------------------------------------------
// zero_ptr / zero_words (stand-by chains only, may be null): the histogram array the chain's next kernel accumulates into -- cleared here
// (grid-stride) instead of by a launch of its own; like everything in the chain only when *run_if says the chain runs at all.
__global__ void k_setup_uniform(uint32_t n, uint32_t chunk_rows, uint32_t n_chunks, uint32_t* __restrict__ chunk_off,
                                uint32_t* __restrict__ chunk_len, uint32_t* __restrict__ zero_ptr = nullptr, uint32_t zero_words = 0,
                                const uint32_t* __restrict__ run_if = nullptr) {
    if (run_if && *run_if == 0) return;
    if (zero_ptr)
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < zero_words; i += gridDim.x * blockDim.x) zero_ptr[i] = 0u;
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    unsigned long long o = (unsigned long long)c * chunk_rows;
    uint32_t off = o < n ? (uint32_t)o : n;
    uint32_t len = n - off < chunk_rows ? n - off : chunk_rows;
    chunk_off[c] = off;
    chunk_len[c] = len;
}

// Groups of a pass = buckets of the previous pass.  Two layouts exist for them:
//   exact    group g spans [scanned[g*stride], scanned[(g+1)*stride]) of the previous output (histogram + scan path);
//   reserved group g spans [g*cap, g*cap + min(cursor[g], cap)): the previous pass reserved its output ranges with
//            per-tile atomics on `cursor` instead of a histogram (k_scatter_big<.., RESERVE>); valid while *flag == 0.
struct GroupSrc {
    const uint32_t* scanned;
    uint32_t stride;
    const uint32_t* cursor;   // null: exact layout only
    uint32_t cap;
    const uint32_t* flag;     // set when a reserved range overflowed: the exact path has redone the pass
    uint32_t nsub;            // reserved layout only, 0/1 = one range per group; n: group g is the n sub-ranges
                              // [(g*n + s) * cap, + min(cursor[g*n + s], cap)) (one per XCD, see ReserveArgs::nsub)
};
__device__ __forceinline__ uint32_t group_pieces(const GroupSrc& gs) { return (gs.cursor && *gs.flag == 0 && gs.nsub > 1) ? gs.nsub : 1u; }
__device__ __forceinline__ void group_piece(const GroupSrc& gs, uint32_t g, uint32_t s, uint32_t& start, uint32_t& len);

__device__ __forceinline__ void group_extent(const GroupSrc& gs, uint32_t g, uint32_t& start, uint32_t& len) {
    if (gs.cursor && *gs.flag == 0) {
        const uint32_t c = gs.cursor[g];
        start = g * gs.cap;
        len = c < gs.cap ? c : gs.cap;
    } else {
        start = gs.scanned[(size_t)g * gs.stride];
        len = gs.scanned[(size_t)(g + 1) * gs.stride] - start;
    }
}

__device__ __forceinline__ void group_piece(const GroupSrc& gs, uint32_t g, uint32_t s, uint32_t& start, uint32_t& len) {
    if (group_pieces(gs) == 1) { group_extent(gs, g, start, len); return; }
    const uint32_t i = g * gs.nsub + s, c = gs.cursor[i];
    start = i * gs.cap;
    len = c < gs.cap ? c : gs.cap;
}

// `run_if` (here and on the other exact-path kernels): when non-null the launch is a no-op unless *run_if != 0 --
// the device-side switch that lets the histogram path stand by behind the no-histogram fast path.
__global__ void k_setup_from_scan(GroupSrc gs, uint32_t n_groups, uint32_t S, uint32_t* __restrict__ chunk_off, uint32_t* __restrict__ chunk_len,
                                  const uint32_t* __restrict__ run_if = nullptr, uint32_t* __restrict__ zero_ptr = nullptr, uint32_t zero_words = 0) {
    if (run_if && *run_if == 0) return;
    if (zero_ptr)   // (as in k_setup_uniform)
        for (uint32_t z = blockIdx.x * blockDim.x + threadIdx.x; z < zero_words; z += gridDim.x * blockDim.x) zero_ptr[z] = 0u;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_groups * S) return;
    uint32_t g = i / S, s = i % S;
    uint32_t start, len;
    uint32_t per = S;            // chunks per extent
    const uint32_t np = group_pieces(gs);
    if (np > 1) {                // a group in per-XCD sub-ranges (the host made S a multiple of their number): chunks per piece
        per = S / np;
        group_piece(gs, g, s / per, start, len);
        s %= per;
    } else {
        group_extent(gs, g, start, len);
    }
    uint32_t cr = (len + per - 1) / per;
    cr = (cr + PT_TILE - 1) / PT_TILE * PT_TILE;
    unsigned long long o = (unsigned long long)s * cr;
    uint32_t off = o < len ? (uint32_t)o : len;
    uint32_t l = len - off < cr ? len - off : cr;
    chunk_off[i] = start + off;
    chunk_len[i] = l;
}

// groups = local partition slots, each fed by n_src source segments (after the
// partition exchange).  slot s = src * cps + c.
__global__ void k_setup_from_segments(const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ seg_len, uint32_t n_groups,
                                      uint32_t n_src, uint32_t cps, uint32_t* __restrict__ chunk_off, uint32_t* __restrict__ chunk_len) {
    uint32_t S = n_src * cps;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_groups * S) return;
    uint32_t g = i / S, s = i % S, src = s / cps, c = s % cps;
    uint32_t start = seg_off[g * n_src + src], len = seg_len[g * n_src + src];
    uint32_t cr = (len + cps - 1) / cps;
    cr = (cr + PT_TILE - 1) / PT_TILE * PT_TILE;
    unsigned long long o = (unsigned long long)c * cr;
    uint32_t off = o < len ? (uint32_t)o : len;
    uint32_t l = len - off < cr ? len - off : cr;
    chunk_off[i] = start + off;
    chunk_len[i] = l;
}

// ---------------------------------------------------------------------------
// histogram: one work-group per chunk, LDS-staged counters (ds_add_u32), written
// bucket-major so that ONE flat exclusive scan yields every output position.
//   hist[(g*F + f)*S + s]
// ---------------------------------------------------------------------------
// NARROW_IN: the input is an array of (key32, row id) tuples (see ReserveArgs).  wide_flag (int64 input only, may be
// null): set to 1 when a key lies outside [0, 2^32) -- the caller then must not use a narrow scatter of these rows.
template <int PASS_TAG, bool NARROW_IN = false>   // PASS_TAG only gives each pass its own kernel name in profiles
__global__ __launch_bounds__(PT_THREADS) void k_hist(const long long* __restrict__ in_keys, const uint32_t* __restrict__ chunk_off,
                                                     const uint32_t* __restrict__ chunk_len, uint32_t* __restrict__ hist, PassCfg cfg,
                                                     const uint32_t* __restrict__ run_if = nullptr, uint32_t* __restrict__ wide_flag = nullptr) {
    __shared__ uint32_t s_cnt[PT_MAX_F];
    if (run_if && *run_if == 0) return;
    const uint32_t chunk = blockIdx.x;
    const uint32_t len = chunk_len[chunk];
    if (len == 0) return;   // hist is pre-zeroed
    const uint32_t off = chunk_off[chunk];
    const uint32_t tid = threadIdx.x;
    for (uint32_t f = tid; f < cfg.F; f += PT_THREADS) s_cnt[f] = 0;
    lds_barrier();
    uint32_t i = tid;
    if (NARROW_IN) {
        const uint2* __restrict__ t = reinterpret_cast<const uint2*>(in_keys) + off;
        for (; i + 7 * PT_THREADS < len; i += 8 * PT_THREADS) {
            uint32_t v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = t[i + r * PT_THREADS].x;
#pragma unroll
            for (int r = 0; r < 8; ++r) atomicAdd(&s_cnt[bucket_of(hash_int32((int32_t)v[r]), cfg)], 1u);
        }
        for (; i < len; i += PT_THREADS) atomicAdd(&s_cnt[bucket_of(hash_int32((int32_t)t[i].x), cfg)], 1u);
    } else {
        const long long* __restrict__ k = in_keys + off;
        unsigned long long hi = 0;
        // 8 independent loads in flight per thread
        for (; i + 7 * PT_THREADS < len; i += 8 * PT_THREADS) {
            long long v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = k[i + r * PT_THREADS];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                hi |= (unsigned long long)v[r];
                atomicAdd(&s_cnt[bucket_of(hash_int64(v[r]), cfg)], 1u);
            }
        }
        for (; i < len; i += PT_THREADS) {
            hi |= (unsigned long long)k[i];
            atomicAdd(&s_cnt[bucket_of(hash_int64(k[i]), cfg)], 1u);
        }
        if (wide_flag && (hi >> 32)) atomicOr(wide_flag, 1u);
    }
    lds_barrier();
    const uint32_t g = chunk / cfg.S, s = chunk % cfg.S;
    for (uint32_t f = tid; f < cfg.F; f += PT_THREADS) hist[((size_t)g * cfg.F + f) * cfg.S + s] = s_cnt[f];
}

// ---------------------------------------------------------------------------
// flat exclusive scan (u32), 3 kernels, in place.  n_blocks <= SC_ELEMS.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* s_w /*[4]*/, uint32_t& total) {
    // 256 threads = 4 waves
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    uint32_t base = 0, t = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t x = s_w[i];
        if ((uint32_t)i < w) base += x;
        t += x;
    }
    total = t;
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(SC_THREADS) void k_scan_reduce(const uint32_t* __restrict__ a, uint32_t n, uint32_t* __restrict__ block_sums,
                                                            const uint32_t* __restrict__ run_if = nullptr) {
    __shared__ uint32_t s_w[4];
    if (run_if && *run_if == 0) return;
    const uint32_t base = blockIdx.x * SC_ELEMS;
    uint32_t v = 0;
#pragma unroll
    for (int e = 0; e < SC_EPT; ++e) {
        uint32_t i = base + e * SC_THREADS + threadIdx.x;
        v += i < n ? a[i] : 0u;
    }
    v = wave_reduce_add(v);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(SC_THREADS) void k_scan_top(uint32_t* __restrict__ block_sums, uint32_t n_blocks,
                                                         const uint32_t* __restrict__ run_if = nullptr) {
    __shared__ uint32_t s_w[4];
    if (run_if && *run_if == 0) return;
    // each thread owns SC_EPT consecutive block sums
    uint32_t v[SC_EPT], sum = 0;
#pragma unroll
    for (int e = 0; e < SC_EPT; ++e) {
        uint32_t i = threadIdx.x * SC_EPT + e;
        v[e] = i < n_blocks ? block_sums[i] : 0u;
        sum += v[e];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan_256(sum, s_w, total);
#pragma unroll
    for (int e = 0; e < SC_EPT; ++e) {
        uint32_t i = threadIdx.x * SC_EPT + e;
        if (i < n_blocks) block_sums[i] = run;
        run += v[e];
    }
}

__global__ __launch_bounds__(SC_THREADS) void k_scan_down(uint32_t* __restrict__ a, uint32_t n, const uint32_t* __restrict__ block_sums,
                                                          const uint32_t* __restrict__ run_if = nullptr) {
    __shared__ uint32_t s_w[4];
    if (run_if && *run_if == 0) return;
    const uint32_t base = blockIdx.x * SC_ELEMS + threadIdx.x * SC_EPT;   // thread owns SC_EPT consecutive elements
    uint32_t v[SC_EPT], sum = 0;
#pragma unroll
    for (int e = 0; e < SC_EPT; ++e) {
        uint32_t i = base + e;
        v[e] = i < n ? a[i] : 0u;
        sum += v[e];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan_256(sum, s_w, total) + block_sums[blockIdx.x];
#pragma unroll
    for (int e = 0; e < SC_EPT; ++e) {
        uint32_t i = base + e;
        if (i < n) a[i] = run;
        run += v[e];
    }
}

// The same scan as ONE launch: a single work-group walks the array in SC_ELEMS-element steps with a running carry.  For the scans of
// stand-by chains (run_if given): they almost never run -- then this is one launch that returns at once instead of three (a 100M x 100M
// join enqueues four such scans: 8 launches of ~5 us less) -- and when the chain does run (a skewed column on the exact path) the walk
// costs ~1.5 us per 4096 elements, at most ~1 ms next to a pass over the rows.
__global__ __launch_bounds__(SC_THREADS) void k_scan_single(uint32_t* __restrict__ a, uint32_t n, const uint32_t* __restrict__ run_if) {
    __shared__ uint32_t s_w[4];
    if (run_if && *run_if == 0) return;
    uint32_t carry = 0;
    for (uint32_t blk = 0; blk < n; blk += SC_ELEMS) {
        const uint32_t base = blk + threadIdx.x * SC_EPT;   // thread owns SC_EPT consecutive elements
        uint32_t v[SC_EPT], sum = 0;
#pragma unroll
        for (int e = 0; e < SC_EPT; ++e) {
            const uint32_t i = base + e;
            v[e] = i < n ? a[i] : 0u;
            sum += v[e];
        }
        uint32_t total;
        uint32_t run = block_exclusive_scan_256(sum, s_w, total) + carry;
#pragma unroll
        for (int e = 0; e < SC_EPT; ++e) {
            const uint32_t i = base + e;
            if (i < n) a[i] = run;
            run += v[e];
        }
        carry += total;
    }
}

// ---------------------------------------------------------------------------
// scatter: one work-group per chunk.  Per 4096-row tile:
//   load keys (+row ids) -> bucket + rank by returning LDS atomic -> exclusive
//   scan of the tile's bucket counts -> regroup rows by bucket in LDS ->
//   write each bucket's run contiguously at its running global cursor.
// Replaces join_kernels.cuh:72-75 (per-row global atomicAdd + scattered 8/4-byte stores).
// ---------------------------------------------------------------------------
template <int PASS_TAG>
__global__ __launch_bounds__(PT_THREADS) void k_scatter(const long long* __restrict__ in_keys, const uint32_t* __restrict__ in_rids,
                                                        uint32_t rid_base, const uint32_t* __restrict__ chunk_off,
                                                        const uint32_t* __restrict__ chunk_len, const uint32_t* __restrict__ scanned,
                                                        long long* __restrict__ out_keys, uint32_t* __restrict__ out_rids, PassCfg cfg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    long long* s_keys = reinterpret_cast<long long*>(s_raw);                       // [PT_TILE]
    uint32_t* s_rids = reinterpret_cast<uint32_t*>(s_raw + PT_TILE * 8);           // [PT_TILE]
    uint32_t* s_cnt = s_rids + PT_TILE;                                            // [F]
    uint32_t* s_start = s_cnt + cfg.F;                                             // [F]
    uint32_t* s_gpos = s_start + cfg.F;                                            // [F]
    uint32_t* s_w = s_gpos + cfg.F;                                                // [PT_WAVES]

    const uint32_t chunk = blockIdx.x;
    const uint32_t len = chunk_len[chunk];
    if (len == 0) return;
    const uint32_t off = chunk_off[chunk];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t F = cfg.F;
    {
        const uint32_t g = chunk / cfg.S, s = chunk % cfg.S;
        for (uint32_t f = tid; f < F; f += PT_THREADS) s_gpos[f] = scanned[((size_t)g * F + f) * cfg.S + s];
    }
    for (uint32_t t0 = 0; t0 < len; t0 += PT_TILE) {
        const uint32_t tn = len - t0 < (uint32_t)PT_TILE ? len - t0 : (uint32_t)PT_TILE;
        const uint32_t rbase = off + t0;
        for (uint32_t f = tid; f < F; f += PT_THREADS) s_cnt[f] = 0;
        long long key[PT_RPT];
        uint32_t rid[PT_RPT], br[PT_RPT];
        // unconditional loads on clamped indices: a conditional load compiles to branch + s_waitcnt vmcnt(0),
        // which serialises the tile's loads; clamping keeps all of them in flight together
#pragma unroll
        for (int r = 0; r < PT_RPT; ++r) {
            const uint32_t i = r * PT_THREADS + tid;
            key[r] = in_keys[rbase + (i < tn ? i : tn - 1)];
        }
        if (in_rids) {
#pragma unroll
            for (int r = 0; r < PT_RPT; ++r) {
                const uint32_t i = r * PT_THREADS + tid;
                rid[r] = in_rids[rbase + (i < tn ? i : tn - 1)];
            }
        } else {
#pragma unroll
            for (int r = 0; r < PT_RPT; ++r) rid[r] = rid_base + rbase + r * PT_THREADS + tid;   // join_kernels.cuh:75
        }
        lds_barrier();   // s_cnt zeroed, s_gpos visible
#pragma unroll
        for (int r = 0; r < PT_RPT; ++r) {
            const uint32_t i = r * PT_THREADS + tid;
            if (i < tn) {
                const uint32_t b = bucket_of(hash_int64(key[r]), cfg);
                const uint32_t rank = atomicAdd(&s_cnt[b], 1u);
                br[r] = (b << 16) | rank;   // b < 1024, rank < 4096
            }
        }
        lds_barrier();
        // exclusive scan of s_cnt[0..F) -> s_start ; F <= 1024 = 2 entries per thread
        {
            const uint32_t f0 = 2 * tid, f1 = 2 * tid + 1;
            const uint32_t c0 = f0 < F ? s_cnt[f0] : 0u, c1 = f1 < F ? s_cnt[f1] : 0u;
            const uint32_t inc = wave_inclusive_scan(c0 + c1);
            if (lane == 63) s_w[wv] = inc;
            lds_barrier();
            uint32_t base = 0;
#pragma unroll
            for (int i = 0; i < PT_WAVES; ++i) base += (uint32_t)i < wv ? s_w[i] : 0u;
            const uint32_t e0 = base + inc - (c0 + c1);
            if (f0 < F) s_start[f0] = e0;
            if (f1 < F) s_start[f1] = e0 + c0;
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < PT_RPT; ++r) {
            const uint32_t i = r * PT_THREADS + tid;
            if (i < tn) {
                const uint32_t pos = s_start[br[r] >> 16] + (br[r] & 0xFFFFu);
                s_keys[pos] = key[r];
                s_rids[pos] = rid[r];
            }
        }
        lds_barrier();
        // contiguous runs out: consecutive threads -> consecutive LDS slots -> (mostly) consecutive addresses
#pragma unroll
        for (int r = 0; r < PT_RPT; ++r) {
            const uint32_t i = r * PT_THREADS + tid;
            if (i < tn) {
                const long long k = s_keys[i];
                const uint32_t b = bucket_of(hash_int64(k), cfg);
                const uint32_t gp = s_gpos[b] + (i - s_start[b]);
                out_keys[gp] = k;
                out_rids[gp] = s_rids[i];
            }
        }
        lds_barrier();
        for (uint32_t f = tid; f < F; f += PT_THREADS) s_gpos[f] += s_cnt[f];
        // the zeroing of s_cnt at the top of the next tile is ordered by the barrier there;
        // s_cnt is read here by the same thread that zeroes it (same f stride), so no hazard.
    }
}



// ---------------------------------------------------------------------------
// scatter v3 (global first pass, implicit row ids): register-resident big tile.
// A tile is 512 threads x BT_RPT keys held in registers; buckets are ranked once for the
// whole tile, then the tile is pushed through a 4096-row LDS window in 3 rounds, so a
// bucket's run per tile is 3 x longer than in v1 for the same LDS (fan-out 512: 24 rows =
// 1.5 key lines instead of 8 rows = half a line).  24 keys/thread keep the kernel at 128 VGPRs
// (two work-groups per CU).  Measured at 1B rows, fan-out 512: 18.0 -> 13.8 ms.  No row-id array is read (join_kernels.cuh:75:
// row id = position), so the registers go to keys.
// ---------------------------------------------------------------------------
constexpr int BT_WIN = 4096;                   // rows staged in LDS at a time (default instantiation)

// NT threads x BT_RPT keys per tile, WIN-row LDS window, WPE waves per SIMD (register budget 512 / WPE);
// PF: the next tile's keys are loaded into a second register set while the current tile drains through LDS.
//
// RESERVE (no first-pass histogram): instead of reading cursors from a scanned histogram, every tile reserves its
// bucket runs with one returning global atomic per (tile, bucket) on `rs.cursor` -- F atomics per 24576 rows, issued
// together before the tile's LDS scan and consumed after it.  Bucket f owns rows [f*cap, (f+1)*cap) of the output;
// hash partitioning keeps bucket sizes within n/F + 6 sigma, and a bucket that does not fit raises rs.overflow (the
// histogram path, standing by with run_if = overflow, then redoes the pass exactly).  Rows within a bucket land in
// tile-completion order, which the join does not depend on.
struct ReserveArgs {
    uint32_t* cursor;        // [F] rows reserved so far per bucket (zeroed)
    uint32_t cap;            // rows provisioned per bucket
    uint32_t* overflow;      // set to 1 when a bucket outgrew its range
    uint32_t* wide;          // NARROW: set to 1 when a key does not fit 32 bits (and polled: the work-group stops at once)
    const uint32_t* run_if;  // when non-null the launch is a no-op unless *run_if != 0
    uint32_t nsub;           // 0/1: one cursor per bucket.  n (= 8, the XCDs): bucket f owns n sub-ranges of `cap` rows with cursors
                             // [f*n + s]; work-group b reserves in sub-range b % n.  Work-groups b and b + 8 share an XCD (observed
                             // round-robin dispatch; speed only), so neighbouring runs -- which share the 128-byte line at their common
                             // boundary -- are written through the SAME L2 and leave it as one full line instead of two partial ones
                             // (each of those costs a read-modify-write at the memory side; measured in DESIGN.md 4.1).
};
__device__ __forceinline__ uint32_t reserve_slot(const ReserveArgs& rs, uint32_t f) {
    const uint32_t n = rs.nsub > 1 ? rs.nsub : 1u;
    return f * n + blockIdx.x % n;
}

// NARROW (optimistic 32-bit keys): the partitioned rows are written as ONE array of 8-byte (key32, row id) tuples instead
// of an 8-byte key array + a 4-byte row-id array: 8 instead of 12 bytes per row through both partition passes and the
// join, half as many write streams, and a 16-row segment is exactly one 128-byte line.  The kernel checks every key
// it reads; the first key outside [0, 2^32) raises rs.wide, every work-group stops at its next tile, the later narrow
// kernels return at once and the 64-bit pipeline -- launched behind with run_if = wide -- does the join instead.
// hash_int64(k) == hash_int32((u32)k) for such keys, so both forms partition identically.

template <int PASS_TAG, int BT_RPT, int NT = PT_THREADS, int WIN = BT_WIN, int WPE = 4, bool PF = false, int WU = 8, bool RESERVE = false,
          bool NARROW = false>
__global__ __launch_bounds__(NT, WPE) void k_scatter_big(const long long* __restrict__ in_keys, uint32_t rid_base,
                                                            const uint32_t* __restrict__ chunk_off, const uint32_t* __restrict__ chunk_len,
                                                            const uint32_t* __restrict__ scanned, long long* __restrict__ out_keys,
                                                            uint32_t* __restrict__ out_rids, PassCfg cfg, ReserveArgs rs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    long long* s_keys = reinterpret_cast<long long*>(s_raw);                 // [WIN]
    uint2* s_rows = reinterpret_cast<uint2*>(s_raw);                         // [WIN] (NARROW: tuples instead of s_keys)
    uint32_t* s_rids = reinterpret_cast<uint32_t*>(s_raw + WIN * 8);         // [WIN]
    uint32_t* s_cnt = s_rids + WIN;
    uint32_t* s_start = s_cnt + cfg.F;
    uint32_t* s_gpos = s_start + cfg.F;
    uint32_t* s_w = s_gpos + cfg.F;
    constexpr int BT_TILE = NT * BT_RPT;   // rows per register-resident tile
    constexpr int NW = NT / 64;
    static_assert(BT_TILE <= 65536, "rank is packed into 16 bits");
    uint2* out_rows = reinterpret_cast<uint2*>(out_keys);                    // NARROW

    if (RESERVE && rs.run_if && *rs.run_if == 0) return;
    const uint32_t chunk = blockIdx.x;
    const uint32_t len = chunk_len[chunk];
    if (len == 0) return;
    const uint32_t off = chunk_off[chunk];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t F = cfg.F;
    if (!RESERVE) {
        const uint32_t g = chunk / cfg.S, s = chunk % cfg.S;
        for (uint32_t f = tid; f < F; f += NT) s_gpos[f] = scanned[((size_t)g * F + f) * cfg.S + s];
    }
    long long key[BT_RPT];
    long long nxt[PF ? BT_RPT : 1];
    if (PF) {
#pragma unroll
        for (int r = 0; r < BT_RPT; ++r) {
            const uint32_t i = r * NT + tid;
            key[r] = NT_LOAD(&in_keys[off + (i < len ? i : len - 1)]);
        }
    }
    for (uint32_t t0 = 0; t0 < len; t0 += BT_TILE) {
        const uint32_t tn = len - t0 < (uint32_t)BT_TILE ? len - t0 : (uint32_t)BT_TILE;
        const uint32_t rbase = off + t0;
        for (uint32_t f = tid; f < F; f += NT) s_cnt[f] = 0;
        uint32_t br[BT_RPT];
        if (!PF) {
#pragma unroll
            for (int r = 0; r < BT_RPT; ++r) {   // unconditional, clamped: all loads in flight together
                const uint32_t i = r * NT + tid;
                key[r] = NT_LOAD(&in_keys[rbase + (i < tn ? i : tn - 1)]);
            }
        }
        if (NARROW && RESERVE && tid == 0) s_w[NW] = __hip_atomic_load(rs.wide, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_barrier();
        if (NARROW && RESERVE && s_w[NW]) return;   // work-group-uniform: some key somewhere did not fit, the 64-bit pipeline takes over
        bool wide = false;
#pragma unroll
        for (int r = 0; r < BT_RPT; ++r) {
            const uint32_t i = r * NT + tid;
            br[r] = 0xFFFFFFFFu;
            if (NARROW && RESERVE) wide |= ((unsigned long long)key[r] >> 32) != 0ull;   // clamped duplicates are real keys too
            if (i < tn) {
                const uint32_t b = bucket_of(hash_int64(key[r]), cfg);
                br[r] = (b << 16) | atomicAdd(&s_cnt[b], 1u);   // b < 1024, rank < 65536
            }
        }
        if (NARROW && RESERVE && wide) atomicOr(rs.wide, 1u);   // (the histogram form was checked by its k_hist)
        if (PF) {   // next tile (clamped into the chunk: the last tile re-reads valid rows it then ignores)
#pragma unroll
            for (int r = 0; r < BT_RPT; ++r) {
                const uint32_t i = t0 + BT_TILE + r * NT + tid;
                nxt[r] = NT_LOAD(&in_keys[off + (i < len ? i : len - 1)]);
            }
        }
        lds_barrier();
        uint32_t rsv0 = 0, rsv1 = 0, rsv_over = 0;
        {
            const uint32_t f0 = 2 * tid, f1 = 2 * tid + 1;
            const uint32_t c0 = f0 < F ? s_cnt[f0] : 0u, c1 = f1 < F ? s_cnt[f1] : 0u;
            uint32_t r0 = 0, r1 = 0;
            if (RESERVE) {   // in flight during the scan below
                if (c0) r0 = atomicAdd(&rs.cursor[reserve_slot(rs, f0)], c0);
                if (c1) r1 = atomicAdd(&rs.cursor[reserve_slot(rs, f1)], c1);
            }
            const uint32_t inc = wave_inclusive_scan(c0 + c1);
            if (lane == 63) s_w[wv] = inc;
            lds_barrier();
            uint32_t base = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) base += (uint32_t)i < wv ? s_w[i] : 0u;
            const uint32_t e0 = base + inc - (c0 + c1);
            if (f0 < F) s_start[f0] = e0;
            if (f1 < F) s_start[f1] = e0 + c0;
            if (RESERVE) {   // consumed as late as possible: after the first window is staged (below)
                rsv0 = r0;
                rsv1 = r1;
                if (r0 + c0 > rs.cap || r1 + c1 > rs.cap) rsv_over = 1u;   // r + c cannot wrap: both < 2^31
            }
        }
        lds_barrier();
        // tile-local slot of every row, computed once (the window rounds below only compare it)
#pragma unroll
        for (int r = 0; r < BT_RPT; ++r)
            if (br[r] != 0xFFFFFFFFu) br[r] = s_start[br[r] >> 16] + (br[r] & 0xFFFFu);
        for (uint32_t w0 = 0; w0 < tn; w0 += WIN) {
#pragma unroll
            for (int r = 0; r < BT_RPT; ++r) {
                {
                    const uint32_t pos = br[r] - w0;   // invalid rows (0xFFFFFFFF) and rows of other windows fall outside
                    if (pos < (uint32_t)WIN) {
                        if (NARROW) {
                            s_rows[pos] = make_uint2((uint32_t)key[r], rid_base + rbase + r * NT + tid);
                        } else {
                            s_keys[pos] = key[r];
                            s_rids[pos] = rid_base + rbase + r * NT + tid;          // join_kernels.cuh:75
                        }
                    }
                }
            }
            if (RESERVE && w0 == 0) {   // the reservations have had the scan and the staging above to come back
                if (2 * tid < F) s_gpos[2 * tid] = reserve_slot(rs, 2 * tid) * rs.cap + rsv0;
                if (2 * tid + 1 < F) s_gpos[2 * tid + 1] = reserve_slot(rs, 2 * tid + 1) * rs.cap + rsv1;
                if (rsv_over) atomicOr(rs.overflow, 1u);
            }
            lds_barrier();
            const uint32_t wn = tn - w0 < (uint32_t)WIN ? tn - w0 : (uint32_t)WIN;
#pragma unroll WU
            for (int j = 0; j < WIN / NT; ++j) {
                const uint32_t i = j * NT + tid;
                if (i < wn) {
                    if (NARROW) {
                        const uint2 row = s_rows[i];
                        const uint32_t b = bucket_of(hash_int32((int32_t)row.x), cfg);   // == hash_int64 of the zero-extended key
                        const uint32_t gp = s_gpos[b] + (w0 + i - s_start[b]);
                        if (!RESERVE || gp < (reserve_slot(rs, b) + 1u) * rs.cap) out_rows[gp] = row;
                    } else {
                        const long long k = s_keys[i];
                        const uint32_t b = bucket_of(hash_int64(k), cfg);
                        const uint32_t gp = s_gpos[b] + (w0 + i - s_start[b]);
                        if (!RESERVE || gp < (reserve_slot(rs, b) + 1u) * rs.cap) {   // rows past a bucket's range are dropped (overflow raised above)
                            out_keys[gp] = k;
                            out_rids[gp] = s_rids[i];
                        }
                    }
                }
            }
            lds_barrier();
        }
        if (!RESERVE)
            for (uint32_t f = tid; f < F; f += NT) s_gpos[f] += s_cnt[f];
        if (PF) {
#pragma unroll
            for (int r = 0; r < BT_RPT; ++r) key[r] = nxt[r];
        }
    }
}

// ---------------------------------------------------------------------------
// No-histogram last pass.  With ONE work-group per group (S = 1) the write cursors of a group's buckets live in
// that work-group's LDS, so the pass needs no histogram at all -- provided every (group, bucket) sub-partition
// owns a pre-provisioned slot.  Hash partitioning makes sub-partition sizes concentrate around n_g / F, so a slot of
// avg + 6 sqrt(avg) + 48 rows (16-row aligned, i.e. every slot starts on a 128-byte key line) overflows with
// probability < 1e-8 per slot.  If a slot does overflow (skewed keys), the work-group raises a device flag, the
// histogram path -- launched right behind with run_if = flag -- redoes the pass exactly, and the join kernel picks
// the layout by the same flag: nothing is read back to the host, the call stays asynchronous.
// ---------------------------------------------------------------------------
struct FixedArgs {
    GroupSrc groups;                // extents of this pass's groups in the previous pass's output
    const uint32_t* slot_base;      // [G+1] first row of group g's slot block
    const uint32_t* slot_cap;       // [G]   rows per slot of group g
    uint32_t* sub_start;            // [G*F] out: first row of sub-partition (g, f)
    uint32_t* sub_len;              // [G*F] out: rows of sub-partition (g, f)
    uint32_t* overflow;             // out: set to 1 when any slot overflowed
    const uint32_t* skip_if;        // when non-null the launch is a no-op if *skip_if != 0 (narrow pipeline abandoned)
};

__device__ __forceinline__ uint32_t isqrt_u32(uint32_t x) {
    uint32_t r = (uint32_t)sqrtf((float)x);
    while ((unsigned long long)r * r > x) --r;
    while ((unsigned long long)(r + 1) * (r + 1) <= x) ++r;
    return r;
}

// limit_rows / overflow: rows the output buffer holds; a layout that would not fit raises *overflow instead of being used (the
// pass behind returns at once through its skip_if, the exact path redoes it) -- the host sizes the buffer from the same
// (sigmas, pad), so this is a guard, not a path that is expected to run.
__global__ __launch_bounds__(1024) void k_slots_setup(GroupSrc gs, uint32_t G, uint32_t F, uint32_t sigmas,
                                                      uint32_t pad, uint32_t* __restrict__ slot_base, uint32_t* __restrict__ slot_cap,
                                                      unsigned long long limit_rows, uint32_t* __restrict__ overflow,
                                                      const uint32_t* __restrict__ key_dict = nullptr, uint32_t dict_mul = 1,
                                                      uint4* __restrict__ zero_ptr = nullptr, uint32_t zero_vec = 0, const uint32_t* __restrict__ zero_if = nullptr) {
    __shared__ uint32_t s_w[16];
    // the cursors of a reserving last pass (k_part2_wide: fx.sub_len), cleared here instead of by a launch of their own -- only when
    // that pass is going to run (zero_if: its gate word)
    if (zero_ptr && (!zero_if || *zero_if != 0))
        for (uint32_t z = threadIdx.x; z < zero_vec; z += blockDim.x) zero_ptr[z] = make_uint4(0u, 0u, 0u, 0u);
    if (key_dict && key_dict[KEY_DICT_MAX] > 1u) sigmas *= dict_mul;   // several upper key halves: wider slots (reserve_cap's comment)
    const uint32_t g = threadIdx.x, lane = g & 63, wv = g >> 6;
    uint32_t total = 0;
    if (g < G) {
        uint32_t n_g = 0;
        for (uint32_t s = 0, np = group_pieces(gs); s < np; ++s) {
            uint32_t st, l;
            group_piece(gs, g, s, st, l);
            n_g += l;
        }
        const uint32_t avg = (n_g + F - 1) / F;
        const uint32_t cap = (avg + sigmas * isqrt_u32(avg) + pad + 15u) & ~15u;
        slot_cap[g] = cap;
        total = cap * F;
    }
    const uint32_t inc = wave_inclusive_scan(total);
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t i = 0; i < wv; ++i) base += s_w[i];
    if (g < G) slot_base[g] = base + inc - total;
    if (g == G - 1) {
        slot_base[G] = base + inc;
        if (overflow && (unsigned long long)(base + inc) > limit_rows) atomicOr(overflow, 1u);
    }
}

// ---------------------------------------------------------------------------
// scatter v2: line-aligned write combining.
// rocprofv3 on v1 (profiles/r01a): WRITE_SIZE = 1.54 x the bytes written, because a
// tile's run for one bucket (8-16 rows at fan-out 256-512) starts and ends inside a
// 128-byte line that the NEXT tile completes ~10 us later, after the L2 has already
// evicted it.  Here every bucket keeps its unfinished line in LDS: a tile only emits
// rows up to the last 16-row (128-byte key line / 64-byte row-id segment) boundary of
// the bucket's global cursor and carries the remaining < 16 rows into the next tile, so
// each line is written exactly once, whole, by one 16-lane group.  Only the first and
// last line of a (chunk, bucket) run can be partial.
//   LDS: staging for carry + new rows (12288 x 12 B) + 6 per-bucket words  ~ 156 KiB,
//   one 1024-thread work-group per CU; the next tile's keys are prefetched into
//   registers while the current tile is written out.
// ---------------------------------------------------------------------------
constexpr int WC_THREADS = 1024;
constexpr int WC_WAVES = WC_THREADS / 64;
constexpr int WC_RPT = 8;
constexpr int WC_CAP = 12288;                 // staging rows (carry + new): 144 KiB
constexpr int WC_MAX_F = 512;                 // carry <= 15 * F = 7680 rows, leaves >= 512 new rows per tile
constexpr int WC_CARRY_PT = 16 * WC_MAX_F / WC_THREADS;   // carry slots inspected per thread
constexpr int WCN_RPT = 12;                   // narrow tuples (8 B/row): 12288 new rows per tile ...
constexpr int WCN_CAP = 18432;                // ... in 144 KiB of staging

// LINE = rows per write-combined segment.  16 (a whole 128-byte key line) is the only instantiation: 8-row segments
// (64-byte key / 32-byte row-id writes) halve the carry but measured 12.4 -> 14.8 ms at 1B rows -- HBM wants whole lines;
// 32-row segments (256-byte key / 128-byte row-id writes) double the carry, which halves the tile that fits next to it
// in LDS and spills the carry registers: 12.4 -> 27.7 ms.
// RPT / CAP: rows per thread per tile and staging rows (the 8-byte narrow tuples leave room for 12 x 1024-row tiles).
template <bool NARROW> struct JoinKeyFwd { using type = long long; };
template <> struct JoinKeyFwd<true> { using type = uint32_t; };

template <int PASS_TAG, bool FIXED, int LINE = 16, bool NARROW = false, int RPT = WC_RPT, int CAP = WC_CAP>
__global__ __launch_bounds__(WC_THREADS) void k_scatter_wc(const long long* __restrict__ in_keys, const uint32_t* __restrict__ in_rids,
                                                           uint32_t rid_base, const uint32_t* __restrict__ chunk_off,
                                                           const uint32_t* __restrict__ chunk_len, const uint32_t* __restrict__ scanned,
                                                           long long* __restrict__ out_keys, uint32_t* __restrict__ out_rids, PassCfg cfg,
                                                           FixedArgs fx, const uint32_t* __restrict__ run_if) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    long long* s_keys = reinterpret_cast<long long*>(s_raw);                 // [CAP]
    uint2* s_rows = reinterpret_cast<uint2*>(s_raw);                         // [CAP] (NARROW: (key32, row id) tuples, see ReserveArgs)
    uint32_t* s_rids = reinterpret_cast<uint32_t*>(s_raw + CAP * 8);      // [CAP]
    const uint2* in_rows = reinterpret_cast<const uint2*>(in_keys);
    uint2* out_rows = reinterpret_cast<uint2*>(out_keys);
    const uint32_t F = cfg.F;
    uint32_t* s_cnt = NARROW ? s_rids : s_rids + CAP;   // rows of bucket in staging this tile (carry + new); no row-id plane when NARROW
    uint32_t* s_start = s_cnt + F;         // staging offset of the bucket
    uint32_t* s_gpos = s_start + F;        // global cursor of the bucket (index of its first staged row)
    uint32_t* s_carry = s_gpos + F;        // rows carried into this tile
    uint32_t* s_csrc = s_carry + F;        // staging offset where those carried rows sit (previous layout)
    uint32_t* s_emit = s_csrc + F;         // rows written out this tile
    uint32_t* s_w = s_emit + F;            // [WC_WAVES]
    uint32_t* s_misc = s_w + WC_WAVES;     // [0] carry total of the next tile

    static_assert(CAP - 15 * WC_MAX_F >= 512, "a tile must always have room for new rows");
    if (run_if && *run_if == 0) return;
    if (FIXED && fx.skip_if && *fx.skip_if != 0) return;
    const uint32_t chunk = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t len, off, slot0 = 0, cap = 0;
    uint32_t piece = 0, n_pieces = 1;   // FIXED: the group may come as several pieces (per-XCD sub-ranges of the first pass); they are
                                        // consumed one after the other, the carried lines do not care where rows come from
    if (FIXED) {   // one work-group = one whole group; buckets go to their pre-provisioned slots
        n_pieces = group_pieces(fx.groups);
        len = 0; off = 0;
        for (; piece < n_pieces; ++piece) {   // the first piece with rows
            group_piece(fx.groups, chunk, piece, off, len);
            if (len) break;
        }
        slot0 = fx.slot_base[chunk];
        cap = fx.slot_cap[chunk];
        if (len == 0) {
            if (tid < F) { fx.sub_start[(size_t)chunk * F + tid] = slot0 + tid * cap; fx.sub_len[(size_t)chunk * F + tid] = 0; }
            return;
        }
        if (tid < F) { s_gpos[tid] = slot0 + tid * cap; s_carry[tid] = 0; s_csrc[tid] = 0; }
    } else {
        len = chunk_len[chunk];
        if (len == 0) return;
        off = chunk_off[chunk];
        const uint32_t g = chunk / cfg.S, s = chunk % cfg.S;
        if (tid < F) {
            s_gpos[tid] = scanned[((size_t)g * F + tid) * cfg.S + s];
            s_carry[tid] = 0;
            s_csrc[tid] = 0;
        }
    }
    uint32_t consumed = 0;
    uint32_t t_eff = len < (uint32_t)(RPT * WC_THREADS) ? len : (uint32_t)(RPT * WC_THREADS);
    typename JoinKeyFwd<NARROW>::type key[RPT];   // 32-bit registers for narrow tuples
    uint32_t rid[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {   // unconditional loads on clamped indices (t_eff >= 1): all in flight together
        const uint32_t i = r * WC_THREADS + tid;
        if (NARROW) {
            const uint2 t = in_rows[off + (i < t_eff ? i : t_eff - 1)];
            key[r] = t.x;
            rid[r] = t.y;
        } else {
            key[r] = (typename JoinKeyFwd<NARROW>::type)NT_LOAD(&in_keys[off + (i < t_eff ? i : t_eff - 1)]);
        }
    }
    if (NARROW) {
    } else if (in_rids) {
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const uint32_t i = r * WC_THREADS + tid;
            rid[r] = NT_LOAD(&in_rids[off + (i < t_eff ? i : t_eff - 1)]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < RPT; ++r) rid[r] = rid_base + off + r * WC_THREADS + tid;   // join_kernels.cuh:75
    }
    for (;;) {
        consumed += t_eff;
        uint32_t n_off = off, n_len = len, n_piece = piece;   // where the NEXT tile comes from
        if (FIXED && consumed >= len) {
            n_len = 0;
            for (++n_piece; n_piece < n_pieces; ++n_piece) {   // work-group-uniform
                group_piece(fx.groups, chunk, n_piece, n_off, n_len);
                if (n_len) break;
            }
        }
        const bool last = consumed >= len && (!FIXED || n_len == 0);
        if (tid < F) s_cnt[tid] = s_carry[tid];
        if (tid == 0) s_misc[0] = 0;
        lds_barrier();
        uint32_t br[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const uint32_t i = r * WC_THREADS + tid;
            if (i < t_eff) {
                const uint32_t b = bucket_of(NARROW ? hash_int32((int32_t)key[r]) : hash_int64((long long)key[r]), cfg);
                br[r] = (b << 16) | atomicAdd(&s_cnt[b], 1u);   // rank continues after the carried rows
            }
        }
        lds_barrier();
        {   // staging offsets + how much of every bucket can leave as whole lines
            const uint32_t c = tid < F ? s_cnt[tid] : 0u;
            const uint32_t inc = wave_inclusive_scan(c);
            if (lane == 63) s_w[wv] = inc;
            lds_barrier();
            uint32_t base = 0;
#pragma unroll
            for (int i = 0; i < WC_WAVES; ++i) base += (uint32_t)i < wv ? s_w[i] : 0u;
            uint32_t nc = 0;
            if (tid < F) {
                s_start[tid] = base + inc - c;
                const uint32_t gp = s_gpos[tid], end = gp + c;
                const uint32_t ae = last ? end : (end & ~(uint32_t)(LINE - 1));
                const uint32_t e = ae > gp ? ae - gp : 0u;
                s_emit[tid] = e;
                nc = c - e;
            }
            nc = wave_reduce_add(nc);
            if (lane == 0 && nc) atomicAdd(&s_misc[0], nc);
        }
        lds_barrier();
        // carried rows: old staging position -> registers -> new staging position
        constexpr int CARRY_PT = LINE * WC_MAX_F / WC_THREADS;   // carry slots inspected per thread
        long long ck[CARRY_PT];   // NARROW: the tuple's 8 bytes
        uint32_t cr[NARROW ? 1 : CARRY_PT], cdst[CARRY_PT];
#pragma unroll
        for (int c8 = 0; c8 < CARRY_PT; ++c8) {
            const uint32_t j = c8 * WC_THREADS + tid, b = j / LINE, k = j % LINE;
            cdst[c8] = 0xFFFFFFFFu;
            if (b < F && k < s_carry[b]) {
                const uint32_t src = s_csrc[b] + k;
                ck[c8] = s_keys[src];
                if (!NARROW) cr[c8] = s_rids[src];
                cdst[c8] = s_start[b] + k;
            }
        }
        lds_barrier();
#pragma unroll
        for (int c8 = 0; c8 < CARRY_PT; ++c8) {
            if (cdst[c8] != 0xFFFFFFFFu) {
                s_keys[cdst[c8]] = ck[c8];
                if (!NARROW) s_rids[cdst[c8]] = cr[c8];
            }
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const uint32_t i = r * WC_THREADS + tid;
            if (i < t_eff) {
                const uint32_t pos = s_start[br[r] >> 16] + (br[r] & 0xFFFFu);
                if (NARROW) {
                    s_rows[pos] = make_uint2((uint32_t)key[r], rid[r]);
                } else {
                    s_keys[pos] = (long long)key[r];
                    s_rids[pos] = rid[r];
                }
            }
        }
        lds_barrier();
        // prefetch the next tile while this one is written out
        const uint32_t room = (uint32_t)CAP - s_misc[0];
        uint32_t t_next = 0;
        if (!last) {
            if (FIXED && consumed >= len) { off = n_off; len = n_len; piece = n_piece; consumed = 0; }   // step into the next piece
            t_next = len - consumed;
            if (t_next > (uint32_t)(RPT * WC_THREADS)) t_next = RPT * WC_THREADS;
            if (t_next > room) t_next = room;
        }
        if (t_next) {   // wave-uniform
            const uint32_t nb = off + consumed;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const uint32_t i = r * WC_THREADS + tid;
                if (NARROW) {
                    const uint2 t = in_rows[nb + (i < t_next ? i : t_next - 1)];
                    key[r] = t.x;
                    rid[r] = t.y;
                } else {
                    key[r] = (typename JoinKeyFwd<NARROW>::type)NT_LOAD(&in_keys[nb + (i < t_next ? i : t_next - 1)]);
                }
            }
            if (NARROW) {
            } else if (in_rids) {
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    const uint32_t i = r * WC_THREADS + tid;
                    rid[r] = NT_LOAD(&in_rids[nb + (i < t_next ? i : t_next - 1)]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < RPT; ++r) rid[r] = rid_base + nb + r * WC_THREADS + tid;
            }
        }
        // write out: one 16-lane group per 128-byte key line
        {
            constexpr uint32_t NG = WC_THREADS / LINE;      // lane groups per work-group
            const uint32_t q = tid / LINE, l = tid % LINE;
            const uint32_t gpb = F >= NG ? 1u : NG / F;     // groups per bucket when the fan-out is small
            for (uint32_t b = q / gpb; b < F; b += NG / gpb) {
                const uint32_t e = s_emit[b];
                if (e == 0) continue;
                const uint32_t sb = s_start[b], gp = s_gpos[b];
                const uint32_t lim = FIXED ? slot0 + (b + 1u) * cap : 0xFFFFFFFFu;   // rows past a slot's end are dropped (flag raised below)
                for (uint32_t k = ((l - gp) & (uint32_t)(LINE - 1)) + LINE * (q % gpb); k < e; k += LINE * gpb) {   // lane l owns global index == l (mod LINE)
                    if (!FIXED || gp + k < lim) {
                        if (NARROW) {
                            out_rows[gp + k] = s_rows[sb + k];   // 16 lanes x 8 B = one 128-byte line
                        } else {
                            out_keys[gp + k] = s_keys[sb + k];
                            out_rids[gp + k] = s_rids[sb + k];
                        }
                    }
                }
            }
        }
        lds_barrier();
        if (tid < F) {
            const uint32_t e = s_emit[tid];
            s_csrc[tid] = s_start[tid] + e;
            s_carry[tid] = s_cnt[tid] - e;
            s_gpos[tid] += e;
        }
        if (last) break;
        t_eff = t_next;
    }
    if (FIXED) {
        lds_barrier();
        if (tid < F) {
            const uint32_t start = slot0 + tid * cap, cnt = s_gpos[tid] - start;
            fx.sub_start[(size_t)chunk * F + tid] = start;
            fx.sub_len[(size_t)chunk * F + tid] = cnt < cap ? cnt : cap;
            if (cnt > cap) atomicOr(fx.overflow, 1u);
        }
    }
}
static_assert(WC_CAP - 15 * WC_MAX_F >= 512, "a tile must always have room for new rows");
static_assert(WC_CARRY_PT * WC_THREADS == 16 * WC_MAX_F, "carry slots per thread");
static_assert((8 * WC_MAX_F) % WC_THREADS == 0, "carry slots per thread (8-row segments)");

}  // namespace golap
#include "golap_part_narrow.h"
#include "golap_part_wide.h"
namespace golap {

// ---------------------------------------------------------------------------
// join of one sub-partition per work-group, table in LDS.
//   table: t_keys[CAP] (8 B), t_rids[CAP] (4 B), t_next[CAP] (u16), heads[4096] (u32)  = 72 KiB
// build  = join_kernels.cuh:89-110 with LDS ds_wrxchg instead of a global atomicExch
// probe  = join_kernels.cuh:129-165; emission: count -> one global atomic per batch -> compacted stores
// ---------------------------------------------------------------------------
struct JoinSide {
    const long long* keys;
    const uint32_t* rids;
    const uint32_t* off;   // sub-partition sp spans [off[sp*stride], off[(sp+1)*stride])
    uint32_t stride;
    // no-histogram layout (valid while *flag == 0): sub-partition sp = [alt_start[sp], alt_start[sp] + alt_len[sp])
    const uint32_t* alt_start;
    const uint32_t* alt_len;
    const uint32_t* flag;
};

__device__ __forceinline__ void side_bounds(const JoinSide& S, uint32_t sp, uint32_t& lo, uint32_t& hi) {
    if (S.alt_start && *S.flag == 0) {
        lo = S.alt_start[sp];
        hi = lo + S.alt_len[sp];
    } else {
        lo = S.off[(size_t)sp * S.stride];
        hi = S.off[(size_t)(sp + 1) * S.stride];
    }
}

constexpr uint32_t JN_NULL = 0xFFFFFFFFu;   // row id of the missing side of an unmatched row (outer joins)

// row access for the two partitioned layouts: key array + row-id array, or (NARROW) one array of (key32, row id) tuples
template <bool NARROW> struct JoinKey { using type = long long; };
template <> struct JoinKey<true> { using type = uint32_t; };
template <bool NARROW>
__device__ __forceinline__ void jn_load(const JoinSide& S, uint32_t i, typename JoinKey<NARROW>::type& key, uint32_t& rid) {
    if (NARROW) {
        const uint2 t = reinterpret_cast<const uint2*>(S.keys)[i];
        key = (typename JoinKey<NARROW>::type)t.x;
        rid = t.y;
    } else {
        key = (typename JoinKey<NARROW>::type)S.keys[i];
        rid = S.rids[i];
    }
}
template <bool NARROW>
__device__ __forceinline__ uint32_t jn_rid(const JoinSide& S, uint32_t i) {
    return NARROW ? reinterpret_cast<const uint2*>(S.keys)[i].y : S.rids[i];
}

// work-group-wide reservation of the output range for one batch: per-slice wave totals are already in
// s_wt[slice * JN_WAVES + wave]; after the call s_wt holds their exclusive offsets and the return value is the
// batch's base position (ONE global atomic per batch).  All threads must call it.
__device__ __forceinline__ unsigned long long jn_reserve(uint32_t* s_wt, unsigned long long* s_base, unsigned long long* out_count) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    lds_barrier();
    if (wv == 0) {
        // 64 lanes <-> JN_R*JN_WAVES = 64 (slice, wave) totals, slice-major
        const uint32_t v = s_wt[lane];
        const uint32_t inc = wave_inclusive_scan(v);
        const uint32_t total = __shfl(inc, 63, 64);
        s_wt[lane] = inc - v;
        if (lane == 0) *s_base = total ? atomicAdd(out_count, (unsigned long long)total) : 0ull;
    }
    lds_barrier();
    return *s_base;
}

// JT: golap_join_type (0 Inner, 1 Left = keep unmatched build rows, 2 Right = keep unmatched probe rows, 3 Full).
// probe_hit: global bitmap over probe positions, only touched when a sub-partition's build side needs more
// than one table chunk (then "unmatched" is only known after the last chunk).
// NARROW: both sides are (key32, row id) tuple arrays (see ReserveArgs); gate/gate_skip: run_if / skip_if on the
// device word that says whether the narrow pipeline was abandoned.
// K16 (round 4, tuple pipeline only): the table keeps the UPPER 19 bits of a tuple's first word, packed with the chain link into ONE 32-bit
// word.  The word is hash_int32(key32 ^ key_hi), a bijection of the key, and every row of a sub-partition shares the hash bits the
// partition passes consumed -- with >= 13 of them (always, from 16M rows on; 18 at 1B) the upper 19 decide equality.  16 KiB for keys +
// links instead of 24 and 1920 chain heads instead of 2048: 39.8 KiB per work-group, FOUR per CU instead of three (the kernel is bound
// by per-work-group latency chains).
constexpr int JN_HEADS16 = 1920;
template <int JT, bool NARROW, bool K16 = false>
__device__ __forceinline__ void join_sub_partition(const uint32_t sp, const JoinSide& B, const JoinSide& P, uint2* __restrict__ out_pairs,
                                                   unsigned long long* __restrict__ out_count, unsigned long long max_pairs,
                                                   uint32_t* __restrict__ probe_hit) {
    static_assert(!K16 || NARROW, "16-bit table keys are a form of the tuple pipeline");
    constexpr bool KEEP_B = (JT & 1) != 0, KEEP_P = (JT & 2) != 0;
    using KeyT = typename JoinKey<NARROW>::type;
    using TabT = typename std::conditional<K16, uint32_t, KeyT>::type;
    // chain heads: 4096 for the 72 KiB wide table (two work-groups per CU either way); 2048 for the narrow one, whose
    // 48 KiB then fit three work-groups per CU -- the kernel is bound by per-work-group latency chains, not by LDS reads
    constexpr int LOG2S = NARROW ? JN_LOG2_SLOTS - 1 : JN_LOG2_SLOTS;
    constexpr int NHEADS = K16 ? JN_HEADS16 : (1 << LOG2S);
    // K16: ONE 32-bit LDS word per table row = the key's upper 19 bits | (index of the next row in the chain + 1) in the lower 13 -- a chain
    // step is one LDS read instead of two (the lower 13 key bits are among those the partition passes have already compared)
    constexpr uint32_t KN_NEXT = 0x1FFFu;
    auto tab_key = [](KeyT k) -> TabT { return (TabT)k; };
    auto slot_of_key = [](KeyT k) -> uint32_t {
        if (K16) return (uint32_t)(((unsigned long long)((uint32_t)k * 0x9E3779B1u) * (uint32_t)JN_HEADS16) >> 32);   // (32-bit multiply + mul-hi range reduction)
        return slot_hash((long long)k, LOG2S);
    };
    __shared__ TabT t_keys[JN_CAP];          // K16: the packed (key | next) words
    __shared__ uint32_t t_rids[JN_CAP];
    __shared__ uint32_t heads[NHEADS];
    __shared__ uint16_t t_next[K16 ? 2 : JN_CAP];
    // one chain step: does row e hold probe key k, and which row follows it (0xFFFF = none)
    auto chain_step = [&](uint32_t e, KeyT k, bool& hit, uint32_t& nxt) {
        if (K16) {
            const uint32_t w = (uint32_t)t_keys[e];
            hit = ((w ^ (uint32_t)k) & ~KN_NEXT) == 0u;
            const uint32_t n = w & KN_NEXT;
            nxt = n ? n - 1u : 0xFFFFu;
        } else {
            hit = t_keys[e] == tab_key(k);
            nxt = t_next[e];
        }
    };
    __shared__ uint8_t t_hit[KEEP_B ? JN_CAP : 4];
    __shared__ uint32_t s_wt[JN_R * JN_WAVES];
    __shared__ unsigned long long s_base;

    uint32_t b0, b1, p0, p1;
    side_bounds(B, sp, b0, b1);
    side_bounds(P, sp, p0, p1);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (b1 == b0 || p1 == p0) {
        // one side empty: nothing matches; an outer join still keeps the other side's rows
        const bool keep = (b1 == b0) ? (KEEP_P && p1 != p0) : KEEP_B;
        if (!keep) return;
        const JoinSide& S = (b1 == b0) ? P : B;
        const uint32_t s0 = (b1 == b0) ? p0 : b0, s1 = (b1 == b0) ? p1 : b1;
        for (uint32_t c = s0; c < s1; c += JN_THREADS * JN_R) {
            uint32_t v[JN_R];
#pragma unroll
            for (int r = 0; r < JN_R; ++r) {
                const uint32_t i = c + r * JN_THREADS + tid;
                v[r] = i < s1 ? 1u : 0u;
                const unsigned long long m = __ballot(v[r]);
                if (lane == 0) s_wt[r * JN_WAVES + wv] = (uint32_t)__popcll(m);
            }
            const unsigned long long base = jn_reserve(s_wt, &s_base, out_count);
#pragma unroll
            for (int r = 0; r < JN_R; ++r) {
                const uint32_t i = c + r * JN_THREADS + tid;
                if (v[r]) {
                    const unsigned long long pos = base + s_wt[r * JN_WAVES + wv] + lane;   // full waves except the last: lanes are dense
                    const uint32_t rid = jn_rid<NARROW>(S, i);
                    if (pos < max_pairs) out_pairs[pos] = (b1 == b0) ? make_uint2(JN_NULL, rid) : make_uint2(rid, JN_NULL);
                }
            }
            lds_barrier();
        }
        return;
    }
    const bool multi_chunk = b1 - b0 > (uint32_t)JN_CAP;

    // the first probe batch is fetched before the table is built: its HBM latency hides behind the build
    KeyT pk[JN_R];
    uint32_t prid[JN_R];
#pragma unroll
    for (int r = 0; r < JN_R; ++r) {   // unconditional loads on clamped indices (p1 > p0): all in flight together
        const uint32_t i = p0 + r * JN_THREADS + tid;
        const uint32_t ic = i < p1 ? i : p1 - 1;
        jn_load<NARROW>(P, ic, pk[r], prid[r]);
    }
    for (uint32_t bc = b0; bc < b1; bc += JN_CAP) {
        const uint32_t nb = b1 - bc < (uint32_t)JN_CAP ? b1 - bc : (uint32_t)JN_CAP;
        for (uint32_t h = tid; h < (uint32_t)NHEADS; h += JN_THREADS) heads[h] = JN_EMPTY;
        if (KEEP_B)
            for (uint32_t h = tid; h < (uint32_t)JN_CAP; h += JN_THREADS) t_hit[h] = 0;
        lds_barrier();
        {
            KeyT bk[JN_CAP / JN_THREADS];
            uint32_t brd[JN_CAP / JN_THREADS];
#pragma unroll
            for (int r = 0; r < JN_CAP / JN_THREADS; ++r) {   // the whole table chunk is requested before the first insert
                const uint32_t i = r * JN_THREADS + tid;
                const uint32_t ic = bc + (i < nb ? i : nb - 1);
                jn_load<NARROW>(B, ic, bk[r], brd[r]);
            }
#pragma unroll
            for (int r = 0; r < JN_CAP / JN_THREADS; ++r) {
                const uint32_t i = r * JN_THREADS + tid;
                if (i < nb) {
                    t_rids[i] = brd[r];
                    const uint32_t old = atomicExch(&heads[slot_of_key(bk[r])], i);   // join_kernels.cuh:109
                    if (K16) {
                        t_keys[i] = (TabT)(((uint32_t)bk[r] & ~KN_NEXT) | (old == JN_EMPTY ? 0u : old + 1u));
                    } else {
                        t_keys[i] = tab_key(bk[r]);
                        t_next[i] = (uint16_t)old;                                                 // :110 (0xFFFFFFFF -> 0xFFFF)
                    }
                }
            }
        }
        lds_barrier();

        for (uint32_t pc = p0; pc < p1; pc += JN_THREADS * JN_R) {
            uint32_t cnt[JN_R], first[JN_R], lp[JN_R];
            const bool reload = pc != p0 || (bc != b0 && p1 - p0 > (uint32_t)(JN_THREADS * JN_R));   // registers still hold batch 0 otherwise
#pragma unroll
            for (int r = 0; r < JN_R; ++r) {
                const uint32_t i = pc + r * JN_THREADS + tid;
                const bool ok = i < p1;
                if (reload) {   // wave-uniform; clamped like the preload
                    const uint32_t ic = ok ? i : p1 - 1;
                    jn_load<NARROW>(P, ic, pk[r], prid[r]);
                }
                first[r] = ok ? 0u : JN_EMPTY;   // marks invalid rows until the head is read
            }
            // phase A: count matches per probe row (walk the chain, join_kernels.cuh:146-165)
#pragma unroll
            for (int r = 0; r < JN_R; ++r) {
                uint32_t c = 0, f = 0xFFFFu;
                const bool valid = first[r] != JN_EMPTY;
                if (valid) {
                    uint32_t e = heads[slot_of_key(pk[r])] & 0xFFFFu;
                    while (e != 0xFFFFu) {
                        bool hit;
                        uint32_t nxt;
                        chain_step(e, pk[r], hit, nxt);
                        if (hit) {   // full-key compare (:152-153; K16: the bits the partition passes have not already compared)
                            if (c == 0) f = e;
                            ++c;
                            if (KEEP_B) t_hit[e] = 1;   // same value from every writer: a plain store is enough
                        }
                        e = nxt;
                    }
                }
                if (KEEP_P && valid) {
                    if (!multi_chunk) {
                        if (c == 0) { c = 1; f = 0xFFFEu; }   // unmatched probe row: one (NULL, probe) pair, marked by f
                    } else if (c) {
                        const uint32_t i = pc + r * JN_THREADS + tid;
                        atomicOr(&probe_hit[i >> 5], 1u << (i & 31));
                    }
                }
                cnt[r] = c;
                first[r] = f;
            }
            // wave-level exclusive prefix (ballot + mbcnt when every count is 0/1)
#pragma unroll
            for (int r = 0; r < JN_R; ++r) {
                uint32_t tot;
                lp[r] = wave_exclusive_count(cnt[r], tot);
                if (lane == 0) s_wt[r * JN_WAVES + wv] = tot;
            }
            const unsigned long long base = jn_reserve(s_wt, &s_base, out_count);
            // phase B: emit.  Slice r of the batch is contiguous across the whole work-group.
#pragma unroll
            for (int r = 0; r < JN_R; ++r) {
                if (cnt[r]) {
                    unsigned long long pos = base + s_wt[r * JN_WAVES + wv] + lp[r];
                    if (KEEP_P && first[r] == 0xFFFEu) {
                        if (pos < max_pairs) out_pairs[pos] = make_uint2(JN_NULL, prid[r]);
                        continue;
                    }
                    uint32_t e = first[r];
                    uint32_t left = cnt[r];
                    for (;;) {
                        bool hit;
                        uint32_t nxt;
                        chain_step(e, pk[r], hit, nxt);
                        if (hit) {
                            if (pos < max_pairs) out_pairs[pos] = make_uint2(t_rids[e], prid[r]);   // join_kernels.cuh:157-160
                            ++pos;
                            if (--left == 0) break;
                        }
                        e = nxt;
                    }
                }
            }
            lds_barrier();   // s_wt / s_base reused by the next batch
        }
        if (KEEP_B) {
            // build rows of this table chunk that no probe row matched: (build, NULL)
            uint32_t v[JN_CAP / JN_THREADS], lpb[JN_CAP / JN_THREADS];
#pragma unroll
            for (int r = 0; r < JN_CAP / JN_THREADS; ++r) {
                const uint32_t i = r * JN_THREADS + tid;
                v[r] = (i < nb && !t_hit[i]) ? 1u : 0u;
                const unsigned long long m = __ballot(v[r]);
                lpb[r] = mbcnt(m);
                if (lane == 0) s_wt[r * JN_WAVES + wv] = (uint32_t)__popcll(m);
            }
            const unsigned long long base = jn_reserve(s_wt, &s_base, out_count);
#pragma unroll
            for (int r = 0; r < JN_CAP / JN_THREADS; ++r) {
                if (v[r]) {
                    const unsigned long long pos = base + s_wt[r * JN_WAVES + wv] + lpb[r];
                    if (pos < max_pairs) out_pairs[pos] = make_uint2(t_rids[r * JN_THREADS + tid], JN_NULL);
                }
            }
            lds_barrier();
        }
    }
    if (KEEP_P && multi_chunk) {
        // probe rows that matched in none of the table chunks: (NULL, probe).  __syncthreads() drains vmcnt(0):
        // every atomicOr this work-group issued has been performed before the bits are read back (from L2)
        __syncthreads();
        for (uint32_t pc = p0; pc < p1; pc += JN_THREADS * JN_R) {
            uint32_t v[JN_R], lpb[JN_R];
#pragma unroll
            for (int r = 0; r < JN_R; ++r) {
                const uint32_t i = pc + r * JN_THREADS + tid;
                v[r] = (i < p1 && !((__hip_atomic_load(&probe_hit[i >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (i & 31)) & 1u)) ? 1u : 0u;
                const unsigned long long m = __ballot(v[r]);
                lpb[r] = mbcnt(m);
                if (lane == 0) s_wt[r * JN_WAVES + wv] = (uint32_t)__popcll(m);
            }
            const unsigned long long base = jn_reserve(s_wt, &s_base, out_count);
#pragma unroll
            for (int r = 0; r < JN_R; ++r) {
                const uint32_t i = pc + r * JN_THREADS + tid;
                if (v[r]) {
                    const unsigned long long pos = base + s_wt[r * JN_WAVES + wv] + lpb[r];
                    if (pos < max_pairs) out_pairs[pos] = make_uint2(JN_NULL, jn_rid<NARROW>(P, i));
                }
            }
            lds_barrier();
        }
    }
}

// One sub-partition per work-group in the normal launch (grid = number of sub-partitions: the hardware dispatcher
// balances them); the gated 64-bit stand-by launch uses a small grid that strides, so that standing by costs a few
// microseconds instead of dispatching 262144 empty work-groups (0.11 ms).
template <int JT, bool NARROW = false, bool STRIDE = false>
__global__ __launch_bounds__(JN_THREADS) void k_join_lds(JoinSide B, JoinSide P, uint2* __restrict__ out_pairs,
                                                         unsigned long long* __restrict__ out_count, unsigned long long max_pairs,
                                                         uint32_t* __restrict__ probe_hit, uint32_t n_sub, const uint32_t* __restrict__ run_if,
                                                         const uint32_t* __restrict__ skip_if, const uint32_t* __restrict__ only_if_dict
#ifdef GOLAP_DIAG   // timing-only experiment (WRONG results): n_spread output counters instead of one -- what does the single counter cost?
                                                         , unsigned long long* __restrict__ spread, uint32_t n_spread
#endif
                                                         ) {
    if (run_if && *run_if == 0) return;
    if (skip_if && *skip_if != 0) return;
    if (only_if_dict && *only_if_dict <= 1u) return;   // (k_join_lds16 in front has done the join)
#ifdef GOLAP_DIAG
    if (n_spread) { out_count = spread + (size_t)(blockIdx.x % n_spread) * 16; max_pairs = max_pairs > (1u << 20) ? max_pairs - (1u << 20) : 0; }
#endif
    if (!STRIDE) {
        join_sub_partition<JT, NARROW>(blockIdx.x, B, P, out_pairs, out_count, max_pairs, probe_hit);
    } else {
        for (uint32_t sp = blockIdx.x; sp < n_sub; sp += gridDim.x) {
            join_sub_partition<JT, NARROW>(sp, B, P, out_pairs, out_count, max_pairs, probe_hit);
            if (sp + gridDim.x < n_sub) lds_barrier();   // the LDS table is reused by the next sub-partition (no wait for the stores in flight)
        }
    }
}
static_assert(JN_CAP / JN_THREADS == JN_R, "the unmatched-build pass reuses the (slice, wave) scan layout");

// The tuple join with 16-bit table keys (join_sub_partition<.., K16>): four work-groups per CU = 8 waves per SIMD = 64 registers.
// dict_n (device word, PassCfg::key_hi[KEY_DICT_MAX]): with more than one upper key half the tuples' first word is not the plain hash
// (dictionary / window forms) and its upper 16 bits do not decide equality -- this kernel then returns at once and k_join_lds,
// enqueued behind it with the same word, does the join (and returns at once otherwise).
template <int JT>
__global__ __launch_bounds__(JN_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_join_lds16(JoinSide B, JoinSide P, uint2* __restrict__ out_pairs,
                                                         unsigned long long* __restrict__ out_count, unsigned long long max_pairs,
                                                         uint32_t* __restrict__ probe_hit, const uint32_t* __restrict__ skip_if,
                                                         const uint32_t* __restrict__ dict_n) {
    if (skip_if && *skip_if != 0) return;
    if (dict_n && *dict_n > 1u) return;
    join_sub_partition<JT, true, true>(blockIdx.x, B, P, out_pairs, out_count, max_pairs, probe_hit);
}

// ---------------------------------------------------------------------------
// BroadcastHashJoin (join_kernel.rs:62-68): one global chained table over the whole
// build side, exactly the reference's layout idea (buckets[capacity] heads seeded
// 0xFFFFFFFF, bucket = hash % capacity, capacity = 1.5 n, README.md:114); the entry
// index IS the build row index, so `next[]` replaces HashEntry and the key compare
// reads build_keys[entry] directly (join_kernels.cuh:152).
// ---------------------------------------------------------------------------
__global__ void k_fill_u32(uint32_t* __restrict__ a, uint32_t n, uint32_t v) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) a[i] = v;
}

__global__ __launch_bounds__(256) void k_bcast_build(const long long* __restrict__ keys, uint32_t n, uint32_t* __restrict__ buckets,
                                                      uint32_t capacity, uint32_t* __restrict__ next) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t h = hash_int64(keys[i]);
        next[i] = atomicExch(&buckets[h % capacity], i);   // join_kernels.cuh:102, :109-110
    }
}

constexpr int BC_THREADS = 512;
constexpr int BC_WAVES = BC_THREADS / 64;
constexpr int BC_R = 8;

template <int JT>
__global__ __launch_bounds__(BC_THREADS) void k_bcast_probe(const long long* __restrict__ build_keys, const uint32_t* __restrict__ buckets,
                                                            uint32_t capacity, const uint32_t* __restrict__ next,
                                                            const long long* __restrict__ probe_keys, uint32_t n_probe,
                                                            uint32_t build_rid_base, uint32_t probe_rid_base, uint2* __restrict__ out_pairs,
                                                            unsigned long long* __restrict__ out_count, unsigned long long max_pairs,
                                                            uint8_t* __restrict__ build_hit) {
    constexpr bool KEEP_B = (JT & 1) != 0, KEEP_P = (JT & 2) != 0;
    __shared__ uint32_t s_wt[BC_R * BC_WAVES];
    __shared__ unsigned long long s_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t batch = BC_THREADS * BC_R;
    for (uint32_t pc = blockIdx.x * batch; pc < n_probe; pc += gridDim.x * batch) {   // uniform per work-group
        long long pk[BC_R];
        uint32_t cnt[BC_R], first[BC_R], lp[BC_R];
#pragma unroll
        for (int r = 0; r < BC_R; ++r) {
            const uint32_t i = pc + r * BC_THREADS + tid;
            pk[r] = probe_keys[i < n_probe ? i : n_probe - 1];
        }
#pragma unroll
        for (int r = 0; r < BC_R; ++r) {
            const uint32_t i = pc + r * BC_THREADS + tid;
            uint32_t c = 0, f = JN_EMPTY;
            if (i < n_probe) {
                uint32_t e = buckets[hash_int64(pk[r]) % capacity];   // join_kernels.cuh:142-143
                while (e != JN_EMPTY) {                                // :146
                    if (build_keys[e] == pk[r]) {                      // :152-153
                        if (c == 0) f = e;
                        ++c;
                        if (KEEP_B) build_hit[e] = 1;
                    }
                    e = next[e];                                       // :164
                }
                if (KEEP_P && c == 0) { c = 1; f = JN_EMPTY - 1; }     // unmatched probe row -> (NULL, probe)
            }
            cnt[r] = c;
            first[r] = f;
        }
#pragma unroll
        for (int r = 0; r < BC_R; ++r) {
            uint32_t tot;
            lp[r] = wave_exclusive_count(cnt[r], tot);
            if (lane == 0) s_wt[r * BC_WAVES + wv] = tot;
        }
        const unsigned long long base = jn_reserve(s_wt, &s_base, out_count);
#pragma unroll
        for (int r = 0; r < BC_R; ++r) {
            if (cnt[r]) {
                const uint32_t i = pc + r * BC_THREADS + tid;
                unsigned long long pos = base + s_wt[r * BC_WAVES + wv] + lp[r];
                if (KEEP_P && first[r] == JN_EMPTY - 1) {
                    if (pos < max_pairs) out_pairs[pos] = make_uint2(JN_NULL, probe_rid_base + i);
                    continue;
                }
                uint32_t e = first[r], left = cnt[r];
                for (;;) {
                    if (build_keys[e] == pk[r]) {
                        if (pos < max_pairs) out_pairs[pos] = make_uint2(build_rid_base + e, probe_rid_base + i);
                        ++pos;
                        if (--left == 0) break;
                    }
                    e = next[e];
                }
            }
        }
        lds_barrier();
    }
}
static_assert(BC_R * BC_WAVES == JN_R * JN_WAVES && BC_THREADS == JN_THREADS, "jn_reserve layout");

// Left / Full: build rows no probe row matched -> (build, NULL)
__global__ __launch_bounds__(BC_THREADS) void k_bcast_unmatched_build(const uint8_t* __restrict__ build_hit, uint32_t n_build, uint32_t build_rid_base,
                                                                      uint2* __restrict__ out_pairs, unsigned long long* __restrict__ out_count,
                                                                      unsigned long long max_pairs) {
    __shared__ uint32_t s_wt[BC_R * BC_WAVES];
    __shared__ unsigned long long s_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t batch = BC_THREADS * BC_R;
    for (uint32_t c0 = blockIdx.x * batch; c0 < n_build; c0 += gridDim.x * batch) {
        uint32_t v[BC_R], lp[BC_R];
#pragma unroll
        for (int r = 0; r < BC_R; ++r) {
            const uint32_t i = c0 + r * BC_THREADS + tid;
            v[r] = (i < n_build && !build_hit[i]) ? 1u : 0u;
            const unsigned long long m = __ballot(v[r]);
            lp[r] = mbcnt(m);
            if (lane == 0) s_wt[r * BC_WAVES + wv] = (uint32_t)__popcll(m);
        }
        const unsigned long long base = jn_reserve(s_wt, &s_base, out_count);
#pragma unroll
        for (int r = 0; r < BC_R; ++r) {
            const uint32_t i = c0 + r * BC_THREADS + tid;
            if (v[r]) {
                const unsigned long long pos = base + s_wt[r * BC_WAVES + wv] + lp[r];
                if (pos < max_pairs) out_pairs[pos] = make_uint2(build_rid_base + i, JN_NULL);
            }
        }
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------
// host: plan
// ---------------------------------------------------------------------------
struct PassPlan {
    PassCfg cfg;
    uint32_t G;   // groups entering the pass
    uint32_t bits;
};

struct JoinPlan {
    std::vector<PassPlan> passes;
    uint32_t total_bits = 8;
    uint64_t n_sub = 256;
    bool allow_slots = true;   // false: the slot-padded (no-histogram) layouts would not fit u32 positions for this input
};

static uint32_t ilog2(uint32_t x) { uint32_t l = 0; while ((1u << l) < x) ++l; return l; }

// smallest number of hash bits so that a sub-partition's build side fits the LDS table
// with ~4.5 sigma of slack (hash partitioning is ~Poisson)
static uint32_t bits_needed(uint64_t n_build) {
    uint32_t b = 8;
    for (; b < 30; ++b) {
        double avg = (double)n_build / (double)(1ull << b);
        if (avg + 4.5 * std::sqrt(avg) <= (double)JN_CAP) break;
    }
    // 18 bits are two passes of 9, 19 would be three: just above 1.0003B rows the whole join would fall off a cliff (2B x 2B:
    // 91 ms with three passes of 64-bit rows).  The LDS join loops over table chunks when a sub-partition's build side exceeds one
    // table (it does so for skew anyway), so up to two tables' worth per sub-partition stays with two passes: the join kernel
    // reads the probe rows of such a sub-partition twice, everything else is unchanged.
    if (b > 18) {
        const double avg18 = (double)n_build / (double)(1ull << 18);
        if (avg18 + 4.5 * std::sqrt(avg18) <= 2.0 * (double)JN_CAP) b = 18;
    }
    return b;
}

// pass structure for `extra` = bits beyond the first 8, given a max of `maxb` bits per pass
static void split_bits(uint32_t total_bits, uint32_t max_per_pass, std::vector<uint32_t>& out) {
    out.clear();
    const char* env = knob("GOLAP_PASS_BITS");   // e.g. "8,5,5" -- experiment hook
    if (env && *env) {
        uint32_t sum = 0;
        const char* p = env;
        while (*p) {
            uint32_t v = (uint32_t)strtoul(p, (char**)&p, 10);
            if (v) { out.push_back(v); sum += v; }
            if (*p == ',') ++p; else if (*p) break;
        }
        if (!out.empty() && sum >= total_bits) {   // first entry < 8: experiment only (plain low-bit digits, see make_plan)
            bool ok = true;
            for (uint32_t v : out) ok = ok && v <= 10;
            if (ok) return;
        }
        out.clear();
    }
    uint32_t n_pass = (total_bits + max_per_pass - 1) / max_per_pass;
    if (n_pass == 0) n_pass = 1;
    uint32_t first = std::max<uint32_t>(8, (total_bits + n_pass - 1) / n_pass);
    out.push_back(first);
    uint32_t rest = total_bits > first ? total_bits - first : 0;
    uint32_t left = n_pass - 1;
    while (rest > 0) {
        if (left == 0) left = 1;
        uint32_t b = (rest + left - 1) / left;
        if (b > 10) b = 10;
        out.push_back(b);
        rest -= b;
        --left;
    }
}

static uint32_t slots_for(uint32_t G, uint64_t n_rows) {
    // aim at ~4096 work-groups, but keep chunks >= 128K rows: every chunk ends with partial lines per bucket
    // (the write-combining scatter flushes its carries there), so short chunks waste write bandwidth
    uint64_t by_wg = std::max<uint64_t>(1, 4096 / std::max<uint32_t>(1, G));
    uint64_t avg_group = n_rows / std::max<uint32_t>(1, G);
    uint64_t by_rows = std::max<uint64_t>(1, avg_group / (32 * PT_TILE));
    return (uint32_t)std::max<uint64_t>(1, std::min(by_wg, by_rows));
}

// uniform chunks of the global first pass: <= 2048 work-groups; once the input is large enough, chunks of at least four
// 24576-row tiles so that the 1024-thread big-tile scatter applies (launch_hist_scatter)
// `tile`: chunks are whole tiles of the kernel that will read them (only a side's very last chunk then ends in a partial tile)
// tiles (of 24576 rows) every work-group of the big-tile first pass gets at least: the size from which that form -- and with it the
// no-histogram tuple pipeline -- is used.  2 since round 2 (13M rows per side; 4 = 25M before): the rewritten passes win from there
// (scripts/size_sweep.py: 16M x 16M 0.75 -> 0.63 ms, 24M x 24M 0.97 -> 0.74 ms).  GOLAP_MIN_TILES: A/B hook.
static uint32_t big_tile_min_tiles() {
    const char* e = knob("GOLAP_MIN_TILES");
    const int v = e ? atoi(e) : 2;
    return (uint32_t)(v >= 1 && v <= 16 ? v : 2);
}

static uint32_t first_pass_chunks(uint64_t n_rows, uint32_t* chunk_rows_out, uint32_t tile = PT_TILE, uint32_t min_tiles = 4) {
    const uint64_t n = std::max<uint64_t>(1, n_rows);
    uint64_t c1 = std::max<uint64_t>(1, std::min<uint64_t>(2048, ceil_div(n, PT_TILE)));
    const uint64_t big = n / ((uint64_t)min_tiles * 24576);
    if (big >= 256) c1 = std::min<uint64_t>(2048, big / 256 * 256);   // whole rounds of one 1024-thread work-group per CU (256 CUs)
    uint32_t chunk_rows = (uint32_t)round_up(round_up(ceil_div(n, c1), PT_TILE), big >= 256 ? tile : PT_TILE);
    if (big >= 256 && tile > (uint32_t)PT_TILE && !(knob("GOLAP_CHUNK_ROUNDS") && !strcmp(knob("GOLAP_CHUNK_ROUNDS"), "0"))) {
        // Chunks are whole tiles, so the rule above leaves the last round of work-groups partly empty (1B rows, 18432-row tiles: 2010
        // chunks of 27 tiles = 7.85 rounds of 256).  Pick the tiles per chunk so that the chunks fill whole rounds.  Round 3 searched
        // 4 .. 8 rounds for the least idle tile slots (1B rows: 1024 chunks of 53 tiles, 0.03 % idle instead of 1.9 %) -- which at 100M
        // rows (5426 tiles) picked 1809 chunks of THREE tiles: every chunk pays one tile load that nothing overlaps and one write-out
        // that overlaps nothing, a third of its work.  Round 4: 1 .. 8 rounds, cost = rounds x (tiles per chunk + RAMP), RAMP = the
        // un-overlapped head and tail of a chunk in tile times (GOLAP_CHUNK_ROUNDS = "lo:hi[:ramp_percent]" restricts / tunes it).
        const uint64_t tiles = ceil_div(n, tile);
        uint64_t lo = 1, hi = 8;
        double ramp = 1.0;
        if (const char* e = knob("GOLAP_CHUNK_ROUNDS")) {
            unsigned a = 0, b = 0, r = 100;
            const int got = sscanf(e, "%u:%u:%u", &a, &b, &r);
            if (got >= 2 && a >= 1 && b >= a && b <= 16) { lo = a; hi = b; }
            if (got >= 3) ramp = r / 100.0;
        }
        double best = -1.0;
        for (uint64_t rounds = lo; rounds <= hi; ++rounds) {
            const uint64_t T = ceil_div(tiles, 256 * rounds);
            if (T < min_tiles) continue;
            const uint64_t nc = ceil_div(tiles, T);
            const double cost = (double)ceil_div(nc, 256) * ((double)T + ramp);
            if (best < 0 || cost < best) { best = cost; chunk_rows = (uint32_t)(T * tile); }
        }
    }
    if (chunk_rows_out) *chunk_rows_out = chunk_rows;
    return (uint32_t)std::max<uint64_t>(1, ceil_div(n, chunk_rows));
}

static void make_plan(uint64_t n_build, uint32_t owners_log2, bool first_pass_is_reference_partition, JoinPlan& plan) {
    plan.passes.clear();
    uint32_t need = bits_needed(n_build);
    std::vector<uint32_t> bits;
    split_bits(need, 9, bits);
    uint32_t shift = 0, G = 1;
    for (size_t k = 0; k < bits.size(); ++k) {
        PassPlan pp;
        memset(&pp, 0, sizeof pp);
        pp.bits = bits[k];
        pp.cfg.F = 1u << bits[k];
        pp.cfg.shift = shift;
        if (k == 0 && first_pass_is_reference_partition && bits[k] >= 8) {
            pp.cfg.mode = 1;
            pp.cfg.e1 = bits[k] - 8;
            pp.cfg.owners_log2 = owners_log2;
        }
        pp.G = G;
        plan.passes.push_back(pp);
        shift += bits[k];
        G *= pp.cfg.F;
    }
    plan.total_bits = shift;
    plan.n_sub = G;
}

// ---------------------------------------------------------------------------
// host: one side through the passes
// ---------------------------------------------------------------------------
struct SideBuffers {
    // ping-pong targets; final result of the side lands in `final_*`
    uint64_t tmp_keys[2] = {0, 0}, tmp_rids[2] = {0, 0};
    uint64_t final_keys = 0, final_rids = 0;
};

struct SideResult {
    const uint32_t* off = nullptr;   // scanned hist of the last pass
    uint32_t stride = 1;
    const long long* keys = nullptr;
    const uint32_t* rids = nullptr;
    const uint32_t* alt_start = nullptr;   // no-histogram layout of the last pass (see FixedArgs)
    const uint32_t* alt_len = nullptr;
    const uint32_t* flag = nullptr;
    uint64_t positions = 0;                // rows the final buffers span (padded when the slots layout is used)
};

// the last pass can skip its histogram when it is the second of two passes over contiguous groups
static bool use_fast_last_pass(const JoinPlan& plan, bool has_segs) {
    const char* env = knob("GOLAP_NOHIST");   // "0" disables (A/B hook)
    if (env && !strcmp(env, "0")) return false;
    if (!plan.allow_slots || has_segs || plan.passes.size() != 2) return false;
    const PassPlan& pp = plan.passes[1];
    return pp.G <= 1024 && pp.cfg.F >= 32 && pp.cfg.F <= (uint32_t)WC_MAX_F;
}

static bool use_big_scatter(uint32_t F, bool local_pass, bool has_rids);

static uint32_t isqrt_host(uint64_t x) {
    uint64_t r = (uint64_t)std::sqrt((double)x);
    while (r * r > x) --r;
    while ((r + 1) * (r + 1) <= x) ++r;
    return (uint32_t)r;
}

// rows provisioned per first-pass bucket when the first pass reserves its output by atomics (k_scatter_big<.., RESERVE>)
// sig_mul: DICT_SIGMA_MUL for the tuple pipeline's layouts.  hash_int64 folds the key halves (join_kernels.cuh:37-41), so in a column
// with D upper halves up to D keys share EVERY hash bit (k and -k-1 in a column of small negative and positive ids): bucket sizes
// are then sums of weight-D items and their standard deviation grows by sqrt(D) <= 2 (KEY_DICT_MAX = 4).  The layouts that may
// hold such a column are provisioned for it: +1 % on the first pass's ranges, +9 % on the final buffers.
constexpr uint32_t DICT_SIGMA_MUL = 2;
static uint64_t reserve_cap(uint64_t n_rows, uint32_t F, uint32_t sig_mul = 1) {
    uint32_t sigmas, pad;
    provision_of("GOLAP_RESERVE_SIGMAS", sigmas, pad);   // (test knob: 0 forces overflows)
    const uint64_t avg = ceil_div(std::max<uint64_t>(n_rows, 1), F);
    return round_up(avg + (uint64_t)sigmas * sig_mul * isqrt_host(avg) + pad, 16);
}

// Sub-ranges per first-pass bucket in the narrow pipeline (ReserveArgs::nsub): one per XCD.  GOLAP_NSUB: A/B hook.
static uint32_t narrow_nsub() {
    const char* e = knob("GOLAP_NSUB");
    const uint32_t v = e ? (uint32_t)atoi(e) : 8u;
    return v >= 1 && v <= 8 ? v : 8u;
}
// the same for the 64-bit pipeline's reserving first pass (its second pass and the stand-by setup read pieces since round 2)
static uint32_t wide_nsub() {
    const char* e = knob("GOLAP_NSUB_WIDE");
    const uint32_t v = e ? (uint32_t)atoi(e) : 8u;
    return v >= 1 && v <= 8 ? v : 8u;
}
// rows provisioned per (bucket, sub-range): the sub-range of work-groups b = s (mod nsub) receives at most ceil(chunks / nsub) chunks
static uint64_t reserve_cap_sub(uint64_t n_rows, uint32_t F, uint32_t nsub, uint32_t n_chunks, uint32_t chunk_rows, uint32_t sig_mul = 1) {
    if (nsub <= 1) return reserve_cap(n_rows, F, sig_mul);
    const uint64_t n_s = std::min<uint64_t>(std::max<uint64_t>(n_rows, 1), ceil_div(n_chunks, nsub) * (uint64_t)chunk_rows);
    uint64_t cap = reserve_cap(n_s, F, sig_mul);
    if (const char* e = knob("GOLAP_CAP_PAD")) cap += (uint64_t)(atoi(e) & ~15);   // experiment hook: shifts every sub-range's base address
    return cap;
}

// No-histogram FIRST pass: single-source global pass of a two-pass plan whose second pass is the no-histogram one,
// large enough for the 1024-thread big-tile kernel.
static bool narrow_size_ok(uint64_t n_rows);
static bool use_reserve_first_pass(const JoinPlan& plan, uint64_t n_rows, bool has_segs, bool has_rids) {
    const char* env = knob("GOLAP_RESERVE");   // "0" disables (A/B hook), "1" drops the size condition (test hook)
    if (env && !strcmp(env, "0")) return false;
    if (!use_fast_last_pass(plan, has_segs) || has_rids) return false;
    const uint32_t F = plan.passes[0].cfg.F;
    if (!use_big_scatter(F, false, false) || F > (uint32_t)WC_MAX_F) return false;   // the stand-by pass uses the write-combining scatter
    if (!(env && !strcmp(env, "1")) && !narrow_size_ok(n_rows)) return false;
    return reserve_cap(n_rows, F) * F < 0xFFFFFFF0ull;
}

static uint32_t part1_narrow_tile();
// the 64-bit pipeline's reserving first pass: k_part1_narrow<.., WIDE> (12288-row tiles) when the column is 16-byte aligned, else
// round 1's k_scatter_big.  GOLAP_P1W=0: the old kernel (A/B hook), 2: the new one with compiler-visible loads and stores.
static int p1w_variant() { const char* e = knob("GOLAP_P1W"); return e ? atoi(e) : kernel_forms_now().p1w; }
// the 64-bit pipeline's no-histogram LAST pass: k_part2_wide (golap_part_wide.h; 1 = hidden loads and stores, 2 = compiler-visible ones) when
// both planes of the first pass's output are 16-byte aligned, else -- and with GOLAP_P2W=0 (A/B hook) -- round 1's k_scatter_wc<2, FIXED>
static int p2w_variant() { const char* e = knob("GOLAP_P2W"); return e ? atoi(e) : kernel_forms_now().p2w; }
constexpr int P2W_RPT = 12;   // 12288-row tiles: 144 KiB of LDS next to 8 KiB of per-bucket words at F = 512
static uint32_t wide_chunks(uint64_t n_rows, uint32_t* chunk_rows_out) {
    return p1w_variant() != 0 ? first_pass_chunks(n_rows, chunk_rows_out, 12288, 4) : first_pass_chunks(n_rows, chunk_rows_out);
}
// chunking of the tuple pipeline's first pass (k_part1_narrow: one 1024-thread work-group per chunk, whole tiles per chunk) and the
// size rule for using it: (almost) every CU gets a chunk of at least big_tile_min_tiles() tiles
static uint32_t narrow_chunks(uint64_t n_rows, uint32_t* chunk_rows_out) { return first_pass_chunks(n_rows, chunk_rows_out, part1_narrow_tile(), big_tile_min_tiles()); }
static bool narrow_size_ok(uint64_t n_rows) {
    uint32_t chunk_rows = 0;
    const uint32_t chunks = narrow_chunks(n_rows, &chunk_rows);
    return chunks >= 192 && chunk_rows >= big_tile_min_tiles() * part1_narrow_tile();   // (whole chunks of at least that many tiles; only the last one may be shorter)
}
// rows the first-pass output buffer must hold
static uint64_t pass1_rows(const JoinPlan& plan, uint64_t n, bool has_segs) {
    if (!use_reserve_first_pass(plan, n, has_segs, false)) return n;
    const uint32_t F = plan.passes[0].cfg.F, nsub = narrow_nsub();
    uint32_t chunk_rows = 0;
    const uint32_t nc = narrow_chunks(n, &chunk_rows);
    uint32_t wchunk = 0;
    const uint32_t wnc = wide_chunks(n, &wchunk);
    const uint64_t wide_rows = reserve_cap_sub(n, F, wide_nsub(), wnc, wchunk) * F * wide_nsub();   // the 64-bit pipeline's own chunking
    return std::max<uint64_t>(std::max<uint64_t>(std::max<uint64_t>(n, reserve_cap(n, F) * F), wide_rows),
                              reserve_cap_sub(n, F, nsub, nc, chunk_rows, DICT_SIGMA_MUL) * F * nsub);   // (narrow tuples: 8 B per row as well)
}

// rows the final buffers of one side must hold when the last pass writes into provisioned slots (k_slots_setup: every slot is
// avg_g + sigmas * sqrt(avg_g) + pad rows, 16-row aligned, avg_g = ceil(n_g / F)).  Upper bound over any split of n rows into
// the G groups (sqrt is concave): n + sigmas * (sqrt(FG * n) + FG) + (pad + 24) * FG.  The SAME (sigmas, pad) the device layout
// uses -- golap_engine_config::provision_sigmas / GOLAP_SLOT_SIGMAS -- so a larger head-room setting grows the buffer with it.
static uint64_t slot_rows_bound(uint64_t n, double FG) {
    uint32_t sigmas, pad;
    provision_of("GOLAP_SLOT_SIGMAS", sigmas, pad);
    sigmas *= DICT_SIGMA_MUL;   // (k_slots_setup widens the slots by this factor when the key column has several upper halves: see reserve_cap)
    return n + (uint64_t)((double)sigmas * (std::sqrt(FG * (double)std::max<uint64_t>(n, 1)) + FG)) + (uint64_t)(((double)pad + 24.0) * FG) + 4096;
}
static uint64_t final_rows(const JoinPlan& plan, uint64_t n, bool has_segs) {
    if (!use_fast_last_pass(plan, has_segs)) return n;
    return slot_rows_bound(n, (double)plan.passes[1].G * (double)plan.passes[1].cfg.F);
}

struct SegTable {
    uint64_t dev_off = 0, dev_len = 0;   // device u32 [G*n_src]
    uint32_t n_src = 0;
};

static size_t scatter_lds_bytes(uint32_t F) { return (size_t)PT_TILE * 12 + (size_t)F * 12 + PT_WAVES * 4 + 16; }
static size_t scatter_wc_lds_bytes(uint32_t F, uint32_t cap = WC_CAP, uint32_t row_bytes = 12) { return (size_t)cap * row_bytes + (size_t)F * 24 + WC_WAVES * 4 + 64; }

// write-combining scatter for fan-outs where a tile's runs are short; the plain tile scatter otherwise
// Measured on MI355X (1B rows, fan-out 512): the write-combining scatter wins when a work-group's
// bucket streams stay inside one group's output window (passes >= 2: 21.3 -> 15.3 ms), and loses
// on the global first pass whose 2 x F streams per work-group span the whole output (18.1 -> 21.5 ms).
static bool use_wc_scatter(uint32_t F, bool local_pass) {
    const char* env = knob("GOLAP_SCATTER");   // "v1" / "wc": A/B hook
    if (env && !strcmp(env, "v1")) return false;
    if (F > (uint32_t)WC_MAX_F) return false;
    if (env && !strcmp(env, "wc")) return true;
    return local_pass && F >= 32;
}

static size_t scatter_big_lds_bytes(uint32_t F, uint32_t win = BT_WIN, uint32_t nt = PT_THREADS) { return (size_t)win * 12 + (size_t)F * 12 + (nt / 64) * 4 + 16; }

// big-tile scatter: global first pass with implicit row ids and a fan-out whose v1 runs would be < 1 line
static bool use_big_scatter(uint32_t F, bool local_pass, bool has_rids) {
    const char* env = knob("GOLAP_SCATTER");
    if (env && !strcmp(env, "v1")) return false;
    if (has_rids || local_pass) return false;
    if (env && !strcmp(env, "big")) return true;
    if (env && !strcmp(env, "wc")) return false;
    return F >= 128;
}

template <int TAG>
static void launch_hist_scatter(bool hist, bool wc, uint32_t n_chunks, hipStream_t st, const long long* keys, const uint32_t* rids, uint32_t rid_base,
                                const uint32_t* c_off, const uint32_t* c_len, uint32_t* hist_arr, long long* out_keys, uint32_t* out_rids,
                                const PassCfg& cfg, const uint32_t* run_if, uint64_t n_rows, int narrow, uint32_t* wide_flag) {
    // narrow: 0 = 64-bit rows; 1 = int64 keys in, (key32, row id) tuples out (exchange-time partition: big-tile scatter);
    //         2 = tuples in and out (segment-fed local passes: write-combining scatter).  Callers check *_narrow_ok first.
    if (narrow && hist) {
        if (narrow == 1)
            hipLaunchKernelGGL((k_hist<TAG, false>), dim3(n_chunks), dim3(PT_THREADS), 0, st, keys, c_off, c_len, hist_arr, cfg, run_if, wide_flag);
        else
            hipLaunchKernelGGL((k_hist<TAG, true>), dim3(n_chunks), dim3(PT_THREADS), 0, st, keys, c_off, c_len, hist_arr, cfg, run_if, (uint32_t*)nullptr);
    } else if (narrow == 1) {
        if (n_chunks >= 224 && n_rows / n_chunks >= 4 * 24576)
            hipLaunchKernelGGL((k_scatter_big<1, 24, 1024, 12288, 4, false, 3, false, true>), dim3(n_chunks), dim3(1024), scatter_big_lds_bytes(cfg.F, 12288, 1024),
                               st, keys, rid_base, c_off, c_len, hist_arr, out_keys, out_rids, cfg, ReserveArgs{});
        else
            hipLaunchKernelGGL((k_scatter_big<1, 24, 512, 4096, 4, false, 8, false, true>), dim3(n_chunks), dim3(512), scatter_big_lds_bytes(cfg.F, 4096, 512), st,
                               keys, rid_base, c_off, c_len, hist_arr, out_keys, out_rids, cfg, ReserveArgs{});
    } else if (narrow == 2) {
        hipLaunchKernelGGL((k_scatter_wc<TAG, false, 16, true>), dim3(n_chunks), dim3(WC_THREADS), scatter_wc_lds_bytes(cfg.F, WC_CAP, 8), st, keys,
                           (const uint32_t*)nullptr, rid_base, c_off, c_len, hist_arr, out_keys, (uint32_t*)nullptr, cfg, FixedArgs{}, run_if);
    } else if (hist) {
        hipLaunchKernelGGL(k_hist<TAG>, dim3(n_chunks), dim3(PT_THREADS), 0, st, keys, c_off, c_len, hist_arr, cfg, run_if);
    } else if (!wc && use_big_scatter(cfg.F, false, rids != nullptr) && TAG == 1) {
        const char* e = knob("GOLAP_BIG_RPT");   // A/B hook: "<variant>"
        // default: 1024 threads x 24 keys (24576-row tile, one work-group per CU) once there are enough chunks of several
        // tiles to give every CU a few; the 512-thread tile (two work-groups per CU) for mid-sized inputs
        const int v = e ? atoi(e) : (n_chunks >= 224 && n_rows / n_chunks >= 4 * 24576 ? 1224 : 24);
#define GOLAP_BIG(RPT, NT, WIN, WPE, PF, ...)                                                                                             \
    hipLaunchKernelGGL((k_scatter_big<1, RPT, NT, WIN, WPE, PF, ##__VA_ARGS__>), dim3(n_chunks), dim3(NT), scatter_big_lds_bytes(cfg.F, WIN, NT), st, keys, \
                       rid_base, c_off, c_len, hist_arr, out_keys, out_rids, cfg, ReserveArgs{})
        switch (v) {
            case 20: GOLAP_BIG(20, 512, 4096, 4, false); break;
            case 16: GOLAP_BIG(16, 512, 4096, 4, false); break;
            case 1116: GOLAP_BIG(16, 1024, 8192, 4, false); break;
            case 1224: GOLAP_BIG(24, 1024, 12288, 4, false, 3); break;
            default: GOLAP_BIG(24, 512, 4096, 4, false); break;
        }
#undef GOLAP_BIG
    } else if (wc) {
        hipLaunchKernelGGL((k_scatter_wc<TAG, false>), dim3(n_chunks), dim3(WC_THREADS), scatter_wc_lds_bytes(cfg.F), st, keys, rids, rid_base, c_off, c_len,
                           hist_arr, out_keys, out_rids, cfg, FixedArgs{}, run_if);
    } else {
        hipLaunchKernelGGL(k_scatter<TAG>, dim3(n_chunks), dim3(PT_THREADS), scatter_lds_bytes(cfg.F), st, keys, rids, rid_base, c_off, c_len, hist_arr,
                           out_keys, out_rids, cfg);
    }
}

static void launch_pass_kernel(int tag, bool hist, bool wc, uint32_t n_chunks, hipStream_t st, const long long* keys, const uint32_t* rids,
                               uint32_t rid_base, const uint32_t* c_off, const uint32_t* c_len, uint32_t* hist_arr, long long* out_keys,
                               uint32_t* out_rids, const PassCfg& cfg, const uint32_t* run_if = nullptr, uint64_t n_rows = 0, int narrow = 0,
                               uint32_t* wide_flag = nullptr) {
    switch (tag) {
        case 0: launch_hist_scatter<1>(hist, wc, n_chunks, st, keys, rids, rid_base, c_off, c_len, hist_arr, out_keys, out_rids, cfg, run_if, n_rows, narrow, wide_flag); break;
        case 1: launch_hist_scatter<2>(hist, wc, n_chunks, st, keys, rids, rid_base, c_off, c_len, hist_arr, out_keys, out_rids, cfg, run_if, n_rows, narrow, wide_flag); break;
        default: launch_hist_scatter<3>(hist, wc, n_chunks, st, keys, rids, rid_base, c_off, c_len, hist_arr, out_keys, out_rids, cfg, run_if, n_rows, narrow, wide_flag); break;
    }
}

// block_sums: ceil(n / 4096) + 1 words while n <= 4096 * 4096 (every caller but the hashed GROUP BY); beyond that the block sums are
// scanned by a nested call whose own block sums sit behind them: flat_scan_scratch_words(n) words in all.
uint64_t flat_scan_scratch_words(uint64_t n) {
    uint64_t words = 0;
    for (uint64_t m = n; ; ) {
        const uint64_t nb = ceil_div(std::max<uint64_t>(m, 1), SC_ELEMS);
        words += nb + 1;
        if (nb <= (uint64_t)SC_ELEMS) break;
        m = nb;
    }
    return words;
}
static int32_t flat_scan_level(golap_ctx* ctx, hipStream_t st, uint32_t* a, uint32_t n, uint32_t* block_sums, const uint32_t* run_if) {
    const uint32_t n_blocks = (uint32_t)ceil_div(n, SC_ELEMS);
    hipLaunchKernelGGL(k_scan_reduce, dim3(n_blocks), dim3(SC_THREADS), 0, st, a, n, block_sums, run_if);
    if (n_blocks <= (uint32_t)SC_ELEMS) hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(SC_THREADS), 0, st, block_sums, n_blocks, run_if);
    else GOLAP_TRY(flat_scan_level(ctx, st, block_sums, n_blocks, block_sums + n_blocks + 1, run_if));   // (exclusive, in place: what k_scan_top does)
    hipLaunchKernelGGL(k_scan_down, dim3(n_blocks), dim3(SC_THREADS), 0, st, a, n, block_sums, run_if);
    return GOLAP_OK;
}
int32_t flat_scan(golap_ctx* ctx, hipStream_t st, uint32_t* a, uint32_t n, uint32_t* block_sums, const uint32_t* run_if) {
    KernelTimer t(ctx, "scan_offsets", st);
    const char* e = knob("GOLAP_SCAN_SINGLE");   // "0": three launches for stand-by scans too (A/B and test hook); "1": one launch for every scan
    const bool single = e ? !strcmp(e, "1") || (run_if && strcmp(e, "0")) : (run_if != nullptr && n <= (1u << 23));
    if (single) {
        hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(SC_THREADS), 0, st, a, n, run_if);
        return GOLAP_OK;
    }
    return flat_scan_level(ctx, st, a, n, block_sums, run_if);
}

// bytes of small scratch one side needs for `plan` (hist/scan arrays + chunk descriptors)
static uint64_t side_small_bytes(const JoinPlan& plan, uint64_t n_rows, uint32_t n_src, bool has_segs) {
    uint64_t total = 0;
    for (size_t k = 0; k < plan.passes.size(); ++k) {
        const PassPlan& pp = plan.passes[k];
        uint64_t S;
        if (k == 0 && !has_segs) S = std::max<uint64_t>(1, std::min<uint64_t>(2048, ceil_div(std::max<uint64_t>(1, n_rows), PT_TILE)));
        else if (k == 0) S = (uint64_t)n_src * std::max<uint32_t>(1, slots_for(pp.G * n_src, n_rows));
        else S = round_up(std::max<uint64_t>(slots_for(pp.G, n_rows), 8), 8);   // (a group in up to 8 per-XCD pieces: chunks per piece)
        uint64_t L = (uint64_t)pp.G * pp.cfg.F * S + 1;
        total += round_up(L * 4, 256) + round_up(ceil_div(L, SC_ELEMS) * 4 + 4, 256) + 2 * round_up((uint64_t)pp.G * S * 4, 256) + 1024;
    }
    return total + 96 * 1024;   // reservation cursors (fan-out x 8 sub-ranges: 16 KiB per side and pipeline), control words and flags of both pipelines
}

// Runs all passes for one side.  `in_keys`/`in_rids` describe pass 0's input; when
// `segs` is non-null pass 0's groups are the local partition slots fed by segments.
// `gate` (device word, may be null): when non-null this whole side only does work if *gate != 0 -- the 64-bit pipeline
// standing by behind the optimistic narrow one (run_side_narrow); only valid for the reserving + no-histogram flow, whose
// every non-stand-by kernel takes a run_if.
static int32_t run_side(golap_ctx* ctx, hipStream_t st, const char* side, JoinPlan& plan, uint64_t in_keys, uint64_t in_rids,
                        uint32_t rid_base, uint64_t n_rows, const SegTable* segs, const SideBuffers& buf, SideResult& res,
                        const uint32_t* gate = nullptr, int narrow = 0, uint32_t* wide_flag = nullptr) {
    const size_t np = plan.passes.size();
    uint64_t cur_keys = in_keys, cur_rids = in_rids;
    const uint32_t* prev_scanned = nullptr;
    uint32_t prev_S = 1;
    char name[48];
    // profile names: the stand-alone partition step (exchange-time, multi-GPU) and the segment-fed local passes behind
    // it are different kernels/bytes from the single-GPU join's passes
    const char* pfx = !strcmp(side, "part") ? "partition" : (segs ? "local" : "radix");
    const bool fast_last = !narrow && use_fast_last_pass(plan, segs != nullptr);
    const bool reserve_first = !narrow && use_reserve_first_pass(plan, n_rows, segs != nullptr, in_rids != 0);
    if (gate && !(fast_last && reserve_first)) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "gated side without the no-histogram flow");
    const char* sfx = gate ? "_wide" : "";
    const uint32_t* run_if = nullptr;
    uint32_t* side_flag2 = nullptr;   // the no-histogram last pass's overflow word when the reserving first pass has allocated (and cleared) it
    GroupSrc groups{};   // how the current pass finds its groups in the previous pass's output
    for (size_t k = 0; k < np; ++k) {
        PassPlan& pp = plan.passes[k];
        uint32_t S;
        uint32_t n_chunks;
        uint32_t chunk_rows = 0;
        if (reserve_first && k == 0) {
            // ---- no-histogram first pass: tiles reserve their bucket runs by atomics (k_scatter_big<.., RESERVE>) ----
            const uint32_t F = pp.cfg.F;
            const uint32_t nc = wide_chunks(n_rows, &chunk_rows);
            const uint32_t nsub = wide_nsub();   // per-XCD sub-ranges (ReserveArgs::nsub), as in the tuple pipeline
            const uint32_t cap = (uint32_t)reserve_cap_sub(n_rows, F, nsub, nc, chunk_rows);
            // cursors, this pass's overflow word and the next pass's (fast_last below) in ONE block, cleared by one launch: each word in a
            // 256-byte line of its own behind the cursors (a polled word must not share a line with cursors that take atomics)
            const uint64_t cur_bytes = round_up((uint64_t)F * nsub * 4, 256);
            uint32_t* cursor = (uint32_t*)ws_take(ctx, cur_bytes + 512);
            uint32_t* flag1 = cursor ? cursor + cur_bytes / 4 : nullptr;
            side_flag2 = cursor ? cursor + cur_bytes / 4 + 64 : nullptr;
            uint32_t* c_off = (uint32_t*)ws_take(ctx, (uint64_t)nc * 4);
            uint32_t* c_len = (uint32_t*)ws_take(ctx, (uint64_t)nc * 4);
            if (!cursor || !flag1 || !c_off || !c_len) return fail(ctx, GOLAP_ERR_OOM, "join workspace exhausted (reservation arrays)");
            GOLAP_TRY(zero_async(ctx, st, (void*)cursor, cur_bytes + 512));
            hipLaunchKernelGGL(k_setup_uniform, dim3((uint32_t)ceil_div(nc, 256)), dim3(256), 0, st, (uint32_t)n_rows, chunk_rows, nc, c_off, c_len);
            PassCfg cfg = pp.cfg;
            cfg.S = nc;
            {
                snprintf(name, sizeof name, "radix_scatter_p1%s", sfx);
                KernelTimer t(ctx, name, st, gate ? 0 : 20 * n_rows, gate ? 20 * n_rows : 0, true);
#define GOLAP_RSV(RPT, WIN, WU)                                                                                                            \
    hipLaunchKernelGGL((k_scatter_big<1, RPT, 1024, WIN, 4, false, WU, true>), dim3(nc), dim3(1024), scatter_big_lds_bytes(F, WIN, 1024), st,      \
                       (const long long*)cur_keys, rid_base, (const uint32_t*)c_off, (const uint32_t*)c_len, (const uint32_t*)nullptr,         \
                       (long long*)buf.tmp_keys[0], (uint32_t*)buf.tmp_rids[0], cfg, ReserveArgs{cursor, cap, flag1, nullptr, gate, nsub})
                const int pw = p1w_variant();
                if (pw != 0 && (cur_keys & 15) == 0 && F <= (uint32_t)WC_MAX_F) {
                    const ReserveArgs rsa{cursor, cap, flag1, nullptr, gate, nsub};
                    if (pw == 2)
                        hipLaunchKernelGGL((k_part1_narrow<12, 12288, true, false, 0, 0, 1>), dim3(nc), dim3(P1N_THREADS), part1_narrow_lds_bytes(F, 12288, true), st,
                                           (const long long*)cur_keys, rid_base, (const uint32_t*)c_off, (const uint32_t*)c_len, (uint2*)buf.tmp_keys[0], cfg, rsa,
                                           (uint32_t*)buf.tmp_rids[0]);
                    else
                        hipLaunchKernelGGL((k_part1_narrow<12, 12288, true, true, 0, 0, 1>), dim3(nc), dim3(P1N_THREADS), part1_narrow_lds_bytes(F, 12288, true), st,
                                           (const long long*)cur_keys, rid_base, (const uint32_t*)c_off, (const uint32_t*)c_len, (uint2*)buf.tmp_keys[0], cfg, rsa,
                                           (uint32_t*)buf.tmp_rids[0]);
                } else {
                    GOLAP_RSV(24, 12288, 3);
                }
#undef GOLAP_RSV
            }
            GOLAP_HIP_TRY(ctx, hipGetLastError());
            groups.cursor = cursor;
            groups.cap = cap;
            groups.flag = flag1;
            groups.nsub = nsub;
            run_if = flag1;   // the histogram path below stands by
        }
        if (fast_last && k == 1) {
            // ---- no-histogram pass: one work-group per group writes into pre-provisioned slots ----
            const uint32_t G = pp.G, F = pp.cfg.F;
            uint32_t* slot_base = (uint32_t*)ws_take(ctx, (uint64_t)(G + 1) * 4);
            uint32_t* slot_cap = (uint32_t*)ws_take(ctx, (uint64_t)G * 4);
            uint32_t* sub_start = (uint32_t*)ws_take(ctx, (uint64_t)G * F * 4);
            uint32_t* sub_len = (uint32_t*)ws_take(ctx, (uint64_t)G * F * 4);
            uint32_t* flag = side_flag2 ? side_flag2 : (uint32_t*)ws_take(ctx, 4);   // (side_flag2: already cleared with the first pass's cursors)
            if (!slot_base || !slot_cap || !sub_start || !sub_len || !flag) return fail(ctx, GOLAP_ERR_OOM, "join workspace exhausted (slot arrays)");
            if (!side_flag2) GOLAP_TRY(zero_async(ctx, st, (void*)(flag), 4));
            uint32_t sigmas, pad;
            provision_of("GOLAP_SLOT_SIGMAS", sigmas, pad);   // (test knob: 0 forces overflows)
            const int p2w = (cur_rids && ((cur_keys | cur_rids) & 15) == 0 && F <= (uint32_t)WC_MAX_F && (((uint64_t)G * F) & 3) == 0) ? p2w_variant() : 0;
            hipLaunchKernelGGL(k_slots_setup, dim3(1), dim3(1024), 0, st, groups, G, F, sigmas, pad, slot_base, slot_cap,
                               (unsigned long long)final_rows(plan, n_rows, false), flag, (const uint32_t*)nullptr, 1u,
                               p2w ? (uint4*)sub_len : (uint4*)nullptr, p2w ? (uint32_t)((uint64_t)G * F / 4) : 0u, gate);   // (ws_take hands out 256-byte aligned blocks)
            FixedArgs fx{groups, slot_base, slot_cap, sub_start, sub_len, flag, flag};   // (skip_if = its own overflow word: set before the launch only by the guard above)
            PassCfg cfg = pp.cfg;
            cfg.S = 1;
            {
                snprintf(name, sizeof name, "%s_scatter_p%zu%s", pfx, k + 1, sfx);
                KernelTimer t(ctx, name, st, gate ? 0 : 24 * n_rows, gate ? 24 * n_rows : 0, true);
                if (p2w != 0) {
                    // chunks per group: its per-XCD pieces x GOLAP_P2W_SPLIT parts each (8 x that when the group is one extent).  Measured at
                    // 1B rows, F = 512, both sides (profiles/r04_ab/p2w_split.log): split 1 -> 12.1 ms, 2 -> 11.26, 4 -> 11.33, 8 -> 11.48
                    // (k_scatter_wc<2, FIXED>: 12.55): 16 work-groups of ~10 tiles per group keep two groups in flight per XCD -- runs of one
                    // bucket arrive ~1 us apart and meet in the L2 -- without paying a work-group's start and drain every 5 tiles
                    const char* se = knob("GOLAP_P2W_SPLIT");
                    const uint32_t split = se && atoi(se) >= 1 && atoi(se) <= 8 ? (uint32_t)atoi(se) : 2u;
                    const uint32_t C = (groups.cursor && groups.nsub > 1 ? groups.nsub : 8u) * split;
                    if (p2w == 2)
                        hipLaunchKernelGGL((k_part2_wide<P2W_RPT, false>), dim3(G * C), dim3(P2W_THREADS), part2_wide_lds_bytes(F, P2W_RPT * P2W_THREADS), st,
                                           (const long long*)cur_keys, (const uint32_t*)cur_rids, (long long*)buf.final_keys, (uint32_t*)buf.final_rids, cfg, fx, G, C, gate);
                    else
                        hipLaunchKernelGGL((k_part2_wide<P2W_RPT, true>), dim3(G * C), dim3(P2W_THREADS), part2_wide_lds_bytes(F, P2W_RPT * P2W_THREADS), st,
                                           (const long long*)cur_keys, (const uint32_t*)cur_rids, (long long*)buf.final_keys, (uint32_t*)buf.final_rids, cfg, fx, G, C, gate);
                } else {
                    hipLaunchKernelGGL((k_scatter_wc<2, true>), dim3(G), dim3(WC_THREADS), scatter_wc_lds_bytes(F), st, (const long long*)cur_keys,
                                       (const uint32_t*)cur_rids, rid_base, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr,
                                       (long long*)buf.final_keys, (uint32_t*)buf.final_rids, cfg, fx, gate);
                }
            }
            GOLAP_HIP_TRY(ctx, hipGetLastError());
            res.alt_start = sub_start;
            res.alt_len = sub_len;
            res.flag = flag;
            run_if = flag;   // the exact pass below stands by: its kernels return at once unless a slot overflowed
        }
        if (k == 0 && !segs) {
            // single group: uniform chunks
            S = first_pass_chunks(n_rows, &chunk_rows);
        } else if (k == 0) {
            uint32_t cps = std::max<uint32_t>(1, slots_for(pp.G * segs->n_src, n_rows));
            S = segs->n_src * cps;
        } else {
            S = slots_for(pp.G, n_rows);
            if (groups.nsub > 1) S = (uint32_t)round_up(std::max<uint32_t>(S, groups.nsub), groups.nsub);   // chunks per piece (k_setup_from_scan)
        }
        pp.cfg.S = S;
        n_chunks = pp.G * S;
        const uint64_t L = (uint64_t)pp.G * pp.cfg.F * S;
        if (L + 1 > 0xFFFFFFF0ull) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "partition plan too large");
        uint32_t* hist = (uint32_t*)ws_take(ctx, (L + 1) * 4);
        uint32_t* bsums = (uint32_t*)ws_take(ctx, ceil_div(L + 1, SC_ELEMS) * 4 + 4);
        uint32_t* c_off = (uint32_t*)ws_take(ctx, (uint64_t)n_chunks * 4);
        uint32_t* c_len = (uint32_t*)ws_take(ctx, (uint64_t)n_chunks * 4);
        if (!hist || !bsums || !c_off || !c_len) return fail(ctx, GOLAP_ERR_OOM, "join workspace exhausted (small arrays)");
        // a stand-by chain's histogram is cleared by its own setup kernel (one launch less per pass and side that returns at once)
        const bool fold_zero = run_if != nullptr && !(k == 0 && segs) && !(knob("GOLAP_FOLD_ZERO") && !strcmp(knob("GOLAP_FOLD_ZERO"), "0"));
        if (!fold_zero) GOLAP_TRY(zero_async(ctx, st, (void*)(hist), (L + 1) * 4));
        // descriptors
        {
            const uint32_t tb = 256, gb = (uint32_t)ceil_div(n_chunks, tb);
            uint32_t* zp = fold_zero ? hist : nullptr;
            const uint32_t zw = fold_zero ? (uint32_t)(L + 1) : 0u;
            if (k == 0 && !segs)
                hipLaunchKernelGGL(k_setup_uniform, dim3(fold_zero ? std::max<uint32_t>(gb, 64) : gb), dim3(tb), 0, st, (uint32_t)n_rows, chunk_rows, n_chunks, c_off, c_len,
                                   zp, zw, fold_zero ? run_if : (const uint32_t*)nullptr);
            else if (k == 0)
                hipLaunchKernelGGL(k_setup_from_segments, dim3(gb), dim3(tb), 0, st, (const uint32_t*)segs->dev_off,
                                   (const uint32_t*)segs->dev_len, pp.G, segs->n_src, S / segs->n_src, c_off, c_len);
            else
                hipLaunchKernelGGL(k_setup_from_scan, dim3(fold_zero ? std::max<uint32_t>(gb, 64) : gb), dim3(tb), 0, st, groups, pp.G, S, c_off, c_len, run_if, zp, zw);
        }
        {
            snprintf(name, sizeof name, run_if ? "%s_hist_p%zu_standby" : "%s_hist_p%zu", pfx, k + 1);
            KernelTimer t(ctx, name, st, run_if ? 0 : 8 * n_rows);
            launch_pass_kernel((int)k, true, false, n_chunks, st, (const long long*)cur_keys, nullptr, 0, c_off, c_len, hist, nullptr, nullptr, pp.cfg, run_if, 0,
                               narrow ? (narrow == 1 && k == 0 ? 1 : 2) : 0, wide_flag);
        }
        GOLAP_TRY(flat_scan(ctx, st, hist, (uint32_t)(L + 1), bsums, run_if));
        const bool last = k + 1 == np;
        const uint64_t out_keys = last ? buf.final_keys : buf.tmp_keys[k & 1];
        const uint64_t out_rids = last ? buf.final_rids : buf.tmp_rids[k & 1];
        {
            snprintf(name, sizeof name, run_if ? "%s_scatter_p%zu_standby" : "%s_scatter_p%zu", pfx, k + 1);
            const int nm = narrow ? (narrow == 1 && k == 0 ? 1 : 2) : 0;
            KernelTimer t(ctx, name, st, run_if ? 0 : (nm ? 16 : (cur_rids ? 12 : 8) + 12) * n_rows);
            launch_pass_kernel((int)k, false, run_if ? true : use_wc_scatter(pp.cfg.F, pp.G > 1), n_chunks, st, (const long long*)cur_keys,
                               (const uint32_t*)cur_rids, rid_base, c_off, c_len, hist, (long long*)out_keys, (uint32_t*)out_rids, pp.cfg, run_if, n_rows, nm,
                               wide_flag);
        }
        GOLAP_HIP_TRY(ctx, hipGetLastError());
        cur_keys = out_keys;
        cur_rids = out_rids;
        prev_scanned = hist;
        prev_S = S;
        groups.scanned = hist;
        groups.stride = S;
        if (k > 0) groups.cursor = nullptr;   // only the first pass has a reserved layout
    }
    res.off = prev_scanned;
    res.stride = prev_S;
    res.keys = (const long long*)cur_keys;
    res.rids = (const uint32_t*)cur_rids;
    res.positions = final_rows(plan, n_rows, segs != nullptr);
    return GOLAP_OK;
}

// kernels that ask for more than 64 KiB of dynamic LDS need the attribute once per DEVICE: tracked per context (one context =
// one device), under ctx->mu like every caller
static bool attr_flag(golap_ctx* ctx, size_t which, bool set = false) {
    if (ctx->attr_done.size() <= which) ctx->attr_done.resize(which + 1, false);
    if (set) ctx->attr_done[which] = true;
    return ctx->attr_done[which];
}
static int32_t ensure_scatter_lds(golap_ctx* ctx) {
    if (attr_flag(ctx, 0)) return GOLAP_OK;
    GOLAP_HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_scatter<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_lds_bytes(PT_MAX_F)));
    GOLAP_HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_scatter<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_lds_bytes(PT_MAX_F)));
    GOLAP_HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_scatter<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_lds_bytes(PT_MAX_F)));
#define GOLAP_BIG_ATTR(RPT, NT, WIN, WPE, PF, ...)                                                                              \
    GOLAP_HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_scatter_big<1, RPT, NT, WIN, WPE, PF, ##__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           (int)scatter_big_lds_bytes(PT_MAX_F, WIN, NT)))
    GOLAP_BIG_ATTR(24, 512, 4096, 4, false);
    GOLAP_BIG_ATTR(20, 512, 4096, 4, false);
    GOLAP_BIG_ATTR(16, 512, 4096, 4, false);
    GOLAP_BIG_ATTR(16, 1024, 8192, 4, false);
    GOLAP_BIG_ATTR(24, 1024, 12288, 4, false, 3, true);
    GOLAP_BIG_ATTR(24, 1024, 12288, 4, false, 3, true, true);
    GOLAP_BIG_ATTR(24, 1024, 12288, 4, false, 3, false, true);
    GOLAP_BIG_ATTR(24, 512, 4096, 4, false, 8, false, true);
    GOLAP_BIG_ATTR(24, 1024, 12288, 4, false, 3);
#undef GOLAP_BIG_ATTR
    {
        const void* wcs[] = {(const void*)k_scatter_wc<1, false>, (const void*)k_scatter_wc<2, false>, (const void*)k_scatter_wc<3, false>,
                             (const void*)k_scatter_wc<2, true>};
        GOLAP_HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_scatter_wc<2, true, 16, true, WCN_RPT, WCN_CAP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)scatter_wc_lds_bytes(WC_MAX_F, WCN_CAP, 8)));
        for (const void* f : wcs) GOLAP_HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_wc_lds_bytes(WC_MAX_F)));
        {
            const void* p1[] = {(const void*)k_part1_narrow<18, 18432, true, true>, (const void*)k_part1_narrow<18, 18432, true, true, 0, 1>,
                                (const void*)k_part1_narrow<14, 14336, true, true, 0, 2>,
                                (const void*)k_part1_narrow<18, 18432, true, false>,
                                (const void*)k_part1_narrow<18, 18432, false, false>, (const void*)k_part1_narrow<16, 16384, true, true>,
                                (const void*)k_part1_narrow<16, 16384, true, false>, (const void*)k_part1_narrow<16, 16384, false, false>,
                                (const void*)k_part1_narrow<24, 12288, false, false>,
#ifdef GOLAP_DIAG   // timing-only diagnostic builds (WRONG results): developer builds only, never in the shipped library
                                (const void*)k_part1_narrow<18, 18432, true, false, 1>, (const void*)k_part1_narrow<18, 18432, true, false, 2>,
                                (const void*)k_part1_narrow<18, 18432, true, false, 3>,
#endif
            };
            for (const void* f : p1) GOLAP_HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)part1_narrow_lds_bytes(PT_MAX_F, 18432)));
            const void* p1w[] = {(const void*)k_part1_narrow<12, 12288, true, true, 0, 0, 1>, (const void*)k_part1_narrow<12, 12288, true, false, 0, 0, 1>};
            for (const void* f : p1w) GOLAP_HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)part1_narrow_lds_bytes(WC_MAX_F, 12288, true)));
            const void* p2w[] = {(const void*)k_part2_wide<P2W_RPT, true>, (const void*)k_part2_wide<P2W_RPT, false>};
            for (const void* f : p2w) GOLAP_HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)part2_wide_lds_bytes(WC_MAX_F, P2W_RPT * P2W_THREADS)));
            const void* p2[] = {(const void*)k_part2_ring<4, 32, true>, (const void*)k_part2_ring<4, 64, true>, (const void*)k_part2_ring<4, 128, true>,
                                (const void*)k_part2_ring<4, 256, true>, (const void*)k_part2_ring<4, 32, false>, (const void*)k_part2_ring<4, 64, false>,
                                (const void*)k_part2_ring<4, 128, false>, (const void*)k_part2_ring<4, 256, false>,
#ifdef GOLAP_DIAG
                                (const void*)k_part2_ring<4, 32, false, 2>,
#endif
            };
            for (const void* f : p2) GOLAP_HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)part2_ring_lds_bytes(P2R_MAX_F, 32)));
        }
        const void* wcn[] = {(const void*)k_scatter_wc<1, false, 16, true>, (const void*)k_scatter_wc<2, false, 16, true>,
                             (const void*)k_scatter_wc<3, false, 16, true>};
        for (const void* f : wcn)
            GOLAP_HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_wc_lds_bytes(WC_MAX_F, WC_CAP, 8)));
    }
    attr_flag(ctx, 0, true);
    return GOLAP_OK;
}

static uint64_t radix_join_ws_bytes(const JoinPlan& plan, uint64_t n_build, uint64_t n_probe, uint32_t n_src, bool has_segs) {
    const size_t np = plan.passes.size();
    uint64_t bytes = 0;
    const uint64_t fb = final_rows(plan, n_build, has_segs), fp = final_rows(plan, n_probe, has_segs);
    bytes += round_up(fb * 8, 256) + round_up(fb * 4, 256);   // build final (slot-padded when the last pass skips its histogram)
    bytes += round_up(fp * 8, 256) + round_up(fp * 4, 256);   // probe final
    if (use_fast_last_pass(plan, has_segs))
        bytes += 2 * (2 * round_up((uint64_t)plan.passes[1].G * plan.passes[1].cfg.F * 4, 256) + 2 * round_up((uint64_t)(plan.passes[1].G + 1) * 4, 256) + 1024);
    const size_t n_tmp = np >= 3 ? 2 : (np == 2 ? 1 : 0);
    const uint64_t tmax = std::max(pass1_rows(plan, n_build, has_segs), pass1_rows(plan, n_probe, has_segs));
    bytes += n_tmp * (round_up(tmax * 8, 256) + round_up(tmax * 4, 256));
    bytes += 2 * (round_up((uint64_t)PT_MAX_F * 4, 256) + 512);   // first-pass reservation cursors + flags
    if (np == 2)                                                   // narrow pipeline: cursors, chunk descriptors, slot arrays (both sides)
        bytes += 2 * (round_up((uint64_t)PT_MAX_F * 8 * 4, 256) + 2 * round_up(2048 * 4, 256) + 2 * round_up((uint64_t)(plan.passes[1].G + 1) * 4, 256) +
                      2 * round_up((uint64_t)plan.passes[1].G * plan.passes[1].cfg.F * 4, 256)) + 4096;
    bytes += side_small_bytes(plan, n_build, n_src, has_segs) + side_small_bytes(plan, n_probe, n_src, has_segs);
    bytes += 2 * round_up((uint64_t)4096 * 4 * 2, 256);   // segment tables
    return bytes + (1 << 20);
}

// shared tail: both sides partitioned -> LDS join.  narrow: both sides are (key32, row id) tuple arrays; run_if / skip_if:
// device gates of the narrow-then-wide scheme (see ReserveArgs); probe_hit_shared: bitmap allocated by the caller.
// hashed_key_hi (tuple joins only, may be null): PassCfg::key_hi of tuples whose first word is hash_int32(key32 ^ key_hi) -- with >= 16
// hash bits consumed by the partition passes the join runs in its 16-bit-key form (k_join_lds16) unless the column turned out to
// have several upper halves.
static int32_t launch_lds_join(golap_ctx* ctx, hipStream_t st, const JoinPlan& plan, const SideResult& b, const SideResult& p, uint64_t n_build,
                               uint64_t n_probe, uint64_t out_pairs, uint64_t out_count, uint64_t max_pairs, int32_t join_type = 0,
                               bool narrow = false, const uint32_t* run_if = nullptr, const uint32_t* skip_if = nullptr,
                               uint32_t* probe_hit_shared = nullptr, const uint32_t* hashed_key_hi = nullptr) {
    JoinSide B{b.keys, b.rids, b.off, b.stride, b.alt_start, b.alt_len, b.flag}, P{p.keys, p.rids, p.off, p.stride, p.alt_start, p.alt_len, p.flag};
    uint32_t* probe_hit = probe_hit_shared;
    if ((join_type & 2) && !probe_hit) {   // Right / Full: bitmap over probe positions for the (rare) multi-chunk sub-partitions
        const uint64_t words = ceil_div(std::max<uint64_t>(1, std::max(n_probe, p.positions)), 32) + 1;
        probe_hit = (uint32_t*)ws_take(ctx, words * 4);
        if (!probe_hit) return fail(ctx, GOLAP_ERR_OOM, "join workspace exhausted (probe bitmap)");
        GOLAP_TRY(zero_async(ctx, st, (void*)(probe_hit), words * 4));
    }
    const uint64_t row_bytes = narrow ? 8 : 12;
    // (gated launches -- the 64-bit pipeline standing by, or the tuple join in front of it -- are credited once the gate is known)
    const uint64_t jbytes = row_bytes * (n_build + n_probe);
    KernelTimer t(ctx, run_if ? "join_build_probe_lds_wide" : "join_build_probe_lds", st, run_if || skip_if ? 0 : jbytes, run_if || skip_if ? jbytes : 0,
                  run_if != nullptr);
    const uint32_t n_sub = (uint32_t)plan.n_sub;
    uint32_t stride_grid = 4096u;
    if (const char* e = knob("GOLAP_JOIN_STRIDE_GRID")) stride_grid = (uint32_t)std::max(1, atoi(e));   // debug hook
    const dim3 grid(run_if ? std::min<uint32_t>(n_sub, stride_grid) : n_sub), block(JN_THREADS);   // stand-by launch: small striding grid
    unsigned long long* oc = (unsigned long long*)out_count;
    // packed table rows (19 key bits + 13 link bits): tuples carry the plain hash (hashed_key_hi given) and the passes have consumed >= 13 of its bits
    uint32_t consumed = 0;
    if (!plan.passes.empty()) consumed = plan.passes.back().cfg.shift + plan.passes.back().bits;
    // ... and the sub-partitions are large enough for the fourth work-group per CU to pay (measured, join kernel alone, 32-bit -> 16-bit
    // table: 1B x 1B, 3815 build rows per sub-partition, 4.68 -> 4.27 ms; 800M, 3052 rows, 4.06 -> 3.87; 600M, 2289 rows, 3.56 -> 3.68;
    // 300M, [9,8], 2289 rows, 1.79 -> 1.82; 1.5B, two table chunks per sub-partition, 14.10 -> 14.03)
    const char* k16_env = knob("GOLAP_JOIN_K16");   // "0": the 32-bit table for every tuple join, "1": the 16-bit one whenever it is valid (A/B hooks)
    const bool k16_pays = (k16_env && !strcmp(k16_env, "1")) || n_build / std::max<uint32_t>(1, n_sub) >= 2700;
    const bool k16 = narrow && hashed_key_hi && !run_if && consumed >= 13 && k16_pays && !(k16_env && !strcmp(k16_env, "0"));
    const uint32_t* dict_n = k16 ? hashed_key_hi + KEY_DICT_MAX : nullptr;
    if (k16) {
        switch (join_type) {
            case 0: hipLaunchKernelGGL((k_join_lds16<0>), dim3(n_sub), block, 0, st, B, P, (uint2*)out_pairs, oc, (unsigned long long)max_pairs, probe_hit, skip_if, dict_n); break;
            case 1: hipLaunchKernelGGL((k_join_lds16<1>), dim3(n_sub), block, 0, st, B, P, (uint2*)out_pairs, oc, (unsigned long long)max_pairs, probe_hit, skip_if, dict_n); break;
            case 2: hipLaunchKernelGGL((k_join_lds16<2>), dim3(n_sub), block, 0, st, B, P, (uint2*)out_pairs, oc, (unsigned long long)max_pairs, probe_hit, skip_if, dict_n); break;
            default: hipLaunchKernelGGL((k_join_lds16<3>), dim3(n_sub), block, 0, st, B, P, (uint2*)out_pairs, oc, (unsigned long long)max_pairs, probe_hit, skip_if, dict_n); break;
        }
    }
#ifdef GOLAP_DIAG
    uint32_t n_spread = 0;
    unsigned long long* spread = nullptr;
    if (const char* e = knob("GOLAP_JOIN_SPREAD")) {
        n_spread = (uint32_t)std::max(0, atoi(e));
        if (n_spread) {
            spread = (unsigned long long*)ws_take(ctx, (uint64_t)n_spread * 128);
            if (!spread) return fail(ctx, GOLAP_ERR_OOM, "diag");
            GOLAP_TRY(zero_async(ctx, st, spread, (uint64_t)n_spread * 128));
        }
    }
#define GOLAP_JN_ARGS B, P, (uint2*)out_pairs, oc, (unsigned long long)max_pairs, probe_hit, n_sub, run_if, skip_if, dict_n, spread, n_spread
#else
#define GOLAP_JN_ARGS B, P, (uint2*)out_pairs, oc, (unsigned long long)max_pairs, probe_hit, n_sub, run_if, skip_if, dict_n
#endif
#define GOLAP_JN(JT)                                                                                          \
    if (narrow)                                                                                               \
        hipLaunchKernelGGL((k_join_lds<JT, true, false>), grid, block, 0, st, GOLAP_JN_ARGS);                 \
    else if (run_if)                                                                                          \
        hipLaunchKernelGGL((k_join_lds<JT, false, true>), grid, block, 0, st, GOLAP_JN_ARGS);                 \
    else                                                                                                      \
        hipLaunchKernelGGL((k_join_lds<JT, false, false>), grid, block, 0, st, GOLAP_JN_ARGS)
    switch (join_type) {
        case 0: GOLAP_JN(0); break;
        case 1: GOLAP_JN(1); break;
        case 2: GOLAP_JN(2); break;
        default: GOLAP_JN(3); break;
    }
#undef GOLAP_JN
#undef GOLAP_JN_ARGS
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

// Optimistic narrow pipeline for one side (see ReserveArgs): reserving first pass + no-histogram second pass over
// (key32, row id) tuples.  Anything that does not fit -- a key outside [0, 2^32), a bucket or slot overflow -- raises
// *wide and the 64-bit pipeline behind takes over; there is no stand-by path of its own.
// The upper key half the tuple pipeline stands on = that of the first build key.  While at it, 8192 evenly spaced keys of
// each side are compared with it: columns that really need 64 bits (two or more upper halves) are then known before the narrow
// attempt starts, and its work-groups return at their first tile instead of partitioning ~5 % of the input in vain (1.1 ms at
// 1B x 1B).  A lone odd key is still found by the pass itself.
// out[0 .. 3] = the dictionary of upper halves (unused entries repeat entry 0), out[4] = how many distinct ones the samples hold; more
// than KEY_DICT_MAX raise *wide.  (Round 3: a column with 2 .. 4 upper halves -- small negative next to small positive keys, two id
// ranges -- keeps its 8-byte tuples, PassCfg::key_hi; before, a second upper half meant the 64-bit pipeline: 31 instead of 18 ms.)
// spt: samples per thread and side (1 .. 8: 1024 .. 8192 evenly spaced keys of each side).  Every sample is a page of its own in a large
// column: the 16384 loads of round 3's 8192 + 8192 samples took 43 us -- of a 2.3 ms join at 100M rows -- through the one CU's address
// translation; 2 per thread (2048 per side, ~12 us) still meet any upper half that 0.2 % of the keys carry with probability 98 %, and
// a key the sample misses is found by the first pass itself as ever.
__global__ __launch_bounds__(1024) void k_key_dict(const long long* __restrict__ bkeys, unsigned long long nb, const long long* __restrict__ pkeys,
                                                  unsigned long long np, uint32_t* __restrict__ out, uint32_t* __restrict__ wide, uint32_t max_d,
                                                  uint32_t window, uint32_t spt) {
    // max_d: entries the dictionary form of the first pass can use (1 or KEY_DICT_MAX); window: upper halves its offset form can tell
    // apart (the first pass's fan-out; 0: form not available).  The sampled upper halves go into a small LDS hash set until more than
    // KEY_DICT_MAX distinct ones are known (from then on only their signed minimum and maximum matter).
    constexpr uint32_t SLOTS = 2048;              // (every thread may insert once before it sees that enough values are known)
    __shared__ unsigned long long s_set[SLOTS];   // 1 << 32 | value, 0 = empty
    __shared__ uint32_t s_vals[KEY_DICT_MAX + 1], s_n;
    __shared__ int s_min, s_max;
    const uint32_t tid = threadIdx.x;
    const uint32_t hi0 = (uint32_t)((unsigned long long)bkeys[0] >> 32);
    uint32_t smp[16];
    const unsigned long long n_smp = (unsigned long long)spt * 1024;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
        const unsigned long long s = (unsigned long long)(j < spt ? j : 0) * 1024 + tid;   // sample s of n_smp (slots beyond spt repeat slot 0: no further load)
        smp[2 * j] = nb ? (uint32_t)((unsigned long long)bkeys[(s * nb) / n_smp] >> 32) : hi0;
        smp[2 * j + 1] = np ? (uint32_t)((unsigned long long)pkeys[(s * np) / n_smp] >> 32) : hi0;
    }
    for (uint32_t i = tid; i < SLOTS; i += 1024) s_set[i] = 0ull;
    if (tid == 0) { s_n = 0; s_min = (int)hi0; s_max = (int)hi0; }
    __syncthreads();
    int mn = (int)smp[0], mx = (int)smp[0];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        mn = (int)smp[j] < mn ? (int)smp[j] : mn;
        mx = (int)smp[j] > mx ? (int)smp[j] : mx;
        if (j && smp[j] == smp[j - 1]) continue;   // (the common case: one value everywhere)
        const unsigned long long e = (1ull << 32) | smp[j];
        uint32_t s = (smp[j] * 0x9E3779B1u) >> 21;
        for (uint32_t probes = 0; probes < SLOTS && __hip_atomic_load(&s_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= KEY_DICT_MAX; ++probes) {
            // (more than KEY_DICT_MAX values known: the set is no longer needed)
            const unsigned long long old = atomicCAS(&s_set[s], 0ull, e);
            if (old == e) break;
            if (old == 0ull) {   // a new value
                const uint32_t at = atomicAdd(&s_n, 1u);
                if (at <= KEY_DICT_MAX) s_vals[at] = smp[j];
                break;
            }
            s = (s + 1u) & (SLOTS - 1u);
        }
    }
    if (mn != (int)hi0) atomicMin(&s_min, mn);
    if (mx != (int)hi0) atomicMax(&s_max, mx);
    __syncthreads();
    const uint32_t n = s_n <= KEY_DICT_MAX ? s_n : KEY_DICT_MAX + 1;   // distinct sampled upper halves, or "more than the dictionary holds"
    // several halves inside a window of `window` consecutive values (signed: -1 and 0 are neighbours): the offset form, which also
    // takes halves the sample has missed; else up to max_d arbitrary values: the dictionary form; else 64-bit rows
    const bool offs = n > 1 && max_d > 1 && window && (uint32_t)(s_max - s_min) < window;
    const bool small = !offs && n <= max_d;
    if (small && tid < n) {   // the dictionary: entry 0 = the first build key's upper half, the others ascending behind it
        const uint32_t v = s_vals[tid];
        uint32_t rank = 0, below0 = 0;   // (hi0 is among the samples: sample 0 of the build side, or every sample when that side is empty)
        for (uint32_t i = 0; i < n; ++i) { rank += s_vals[i] < v ? 1u : 0u; below0 += s_vals[i] < hi0 ? 1u : 0u; }
        out[v == hi0 ? 0u : (rank < below0 ? rank + 1u : rank)] = v;
    }
    __syncthreads();
    if (tid < KEY_DICT_MAX && (!small || tid >= n)) out[tid] = small ? out[0] : hi0;   // unused entries repeat entry 0
    if (tid == 0) {
        out[KEY_DICT_MAX] = small ? n : (offs ? KEY_DICT_MAX + 1 : 1u);
        // the window starts below the smallest value seen by as much as the values seen leave free on both sides: halves the sample
        // missed are still inside if they are near
        out[KEY_DICT_BASE_AT] = offs ? (uint32_t)s_min - (window - 1u - (uint32_t)(s_max - s_min)) / 2u : 0u;
        if (!small && !offs) atomicOr(wide, 1u);
    }
}

static uint32_t key_dict_samples() {   // samples per thread and side of k_key_dict (GOLAP_DICT_SAMPLES: 1 .. 8; 8 = round 3's 8192 per side)
    const char* e = knob("GOLAP_DICT_SAMPLES");
    const int v = e ? atoi(e) : 2;
    return (uint32_t)(v >= 1 && v <= 8 ? v : 2);
}

// A/B hooks for the round-2 narrow kernels (scripts/ab_env.py): GOLAP_P1N = 0 (k_scatter_big) | 24 | 18 | 16, GOLAP_P2R = 0 (k_scatter_wc) | 4 | 2
// The defaults -- hidden-memory forms (118 / 104) or plain ones (18 / 4) -- follow the compiler that built the library: golap_variants.h.
static int p1n_variant() { const char* e = knob("GOLAP_P1N"); return e ? atoi(e) : kernel_forms_now().p1n; }
static int p2r_variant() { const char* e = knob("GOLAP_P2R"); return e ? atoi(e) : kernel_forms_now().p2r; }

// rows per tile of the first-pass kernel launch_part1_narrow will choose (chunks are cut in whole tiles of it)
static uint32_t part1_narrow_tile() {
    const int v1 = p1n_variant();
    if (v1 == 16 || v1 == 116) return 16 * P1N_THREADS;
    if (v1 == 24 || v1 == 124 || v1 == 0) return 24 * P1N_THREADS;
    return 18 * P1N_THREADS;
}

// the offset form (k_part1_narrow<.., DICT = 2>: more than 4 upper halves inside a window of F values) exists for the default kernel choice on
// 16-byte aligned columns.  GOLAP_DICT_OFFSET=0: A/B and test hook (such columns then take the 64-bit pipeline, as before).
static bool part1_offset_form_ok(uint64_t key_ptrs_ored) {
    const char* e = knob("GOLAP_DICT_OFFSET");
    return p1n_variant() == 118 && (key_ptrs_ored & 15) == 0 && !(e && !strcmp(e, "0"));
}

// first pass of the narrow pipeline (also the exchange-time partition of the multi-GPU fast path)
static void launch_part1_narrow(hipStream_t st, uint64_t in_keys, uint32_t rid_base, uint32_t nc, const uint32_t* c_off, const uint32_t* c_len,
                                uint64_t tmp_rows, const PassCfg& cfg, const ReserveArgs& rsa) {
    const int v1 = p1n_variant();
    const bool al16 = (in_keys & 15) == 0;
    const uint32_t F1 = cfg.F;
#define GOLAP_P1N(RPT, WIN, VEC, ASM, ...)                                                                                                            \
    hipLaunchKernelGGL((k_part1_narrow<RPT, WIN, VEC, ASM, ##__VA_ARGS__>), dim3(nc), dim3(P1N_THREADS), part1_narrow_lds_bytes(F1, WIN), st, (const long long*)in_keys, \
                       rid_base, c_off, c_len, (uint2*)tmp_rows, cfg, rsa, (uint32_t*)nullptr)
#ifdef GOLAP_DIAG
    if (v1 == 901) { GOLAP_P1N(18, 18432, true, false, 1); return; }   // 9xx: timing-only diagnostic builds (scripts/ab_p12.py; make DIAG=1)
    if (v1 == 902) { GOLAP_P1N(18, 18432, true, false, 2); return; }
    if (v1 == 903) { GOLAP_P1N(18, 18432, true, false, 3); return; }
#endif
    if (v1 == 118 && al16) {
        GOLAP_P1N(18, 18432, true, true);
        if (cfg.key_hi && cfg.pre_hashed) GOLAP_P1N(18, 18432, true, true, 0, 1);   // the dictionary form: runs instead when the column has 2 .. 4 upper halves
        if (cfg.key_hi && cfg.pre_hashed && part1_offset_form_ok(in_keys)) GOLAP_P1N(14, 14336, true, true, 0, 2);   // ... and the offset form: more of them, inside a window of F values
    } else if (v1 == 116 && al16) GOLAP_P1N(16, 16384, true, true);
    else if (v1 == 24 || v1 == 124) GOLAP_P1N(24, 12288, false, false);
    else if ((v1 == 18 || v1 == 118) && al16) GOLAP_P1N(18, 18432, true, false);
    else if (v1 == 18 || v1 == 118) GOLAP_P1N(18, 18432, false, false);
    else if ((v1 == 16 || v1 == 116) && al16) GOLAP_P1N(16, 16384, true, false);
    else if (v1 == 16 || v1 == 116) GOLAP_P1N(16, 16384, false, false);
    else
        hipLaunchKernelGGL((k_scatter_big<1, 24, 1024, 12288, 4, false, 3, true, true>), dim3(nc), dim3(1024), scatter_big_lds_bytes(F1, 12288, 1024), st,
                           (const long long*)in_keys, rid_base, c_off, c_len, (const uint32_t*)nullptr, (long long*)tmp_rows, (uint32_t*)nullptr, cfg, rsa);
#undef GOLAP_P1N
}

// last pass of the narrow pipeline: per-bucket LDS rings (k_part2_ring); F <= P2R_MAX_F
static void launch_part2_ring(hipStream_t st, uint32_t G, uint64_t in_rows, uint64_t out_rows, const PassCfg& cfg_in, const FixedArgs& fx, const SegSrc& seg) {
    PassCfg cfg = cfg_in;
    { const char* e = knob("GOLAP_P2R_ROT"); cfg.piece_rot_off = e && !strcmp(e, "0") ? 1u : 0u; }
    const int v2 = p2r_variant();
    // (ring rows = 16384 / F.  Shorter rings were tried once more in round 4 at the mid sizes -- F = 128: 128 rows 0.746 ms, 64 rows 0.790,
    //  32 rows 1.152 per 2 x 100M rows; F = 256: 64 rows 2.177, 32 rows 2.907 per 2 x 300M -- the longest ring that fits is the fastest)
    const uint32_t R = part2_ring_rows(cfg.F);
#define GOLAP_P2R(RR, ASM, ...)                                                                                                           \
    hipLaunchKernelGGL((k_part2_ring<4, RR, ASM, ##__VA_ARGS__>), dim3(G), dim3(P2R_THREADS), part2_ring_lds_bytes(cfg.F, RR), st, (const uint2*)in_rows, \
                       (uint2*)out_rows, cfg, fx, seg)
#define GOLAP_P2R_BY_R(ASM, ...)                                                                                                          \
    do {                                                                                                                                  \
        if (R == 32) GOLAP_P2R(32, ASM, ##__VA_ARGS__);                                                                                   \
        else if (R == 64) GOLAP_P2R(64, ASM, ##__VA_ARGS__);                                                                              \
        else if (R == 128) GOLAP_P2R(128, ASM, ##__VA_ARGS__);                                                                            \
        else GOLAP_P2R(256, ASM, ##__VA_ARGS__);                                                                                          \
    } while (0)
#ifdef GOLAP_DIAG
    if (v2 == 902) { GOLAP_P2R(32, false, 2); return; }   // timing-only diagnostic (scripts/ab_p12.py; make DIAG=1)
#endif
    if (v2 == 4) GOLAP_P2R_BY_R(false);      // plain loads and stores, hipcc's own waits (A/B reference)
    else GOLAP_P2R_BY_R(true);
#undef GOLAP_P2R_BY_R
#undef GOLAP_P2R
}

// cursor_zeroed (may be null): F1 * narrow cursors' worth of words the caller has already cleared (one launch for both sides' cursors and
// the call's control words instead of three)
static int32_t run_side_narrow(golap_ctx* ctx, hipStream_t st, const JoinPlan& plan, uint64_t in_keys, uint32_t rid_base, uint64_t n_rows,
                               uint64_t tmp_rows, uint64_t final_rows_ptr, uint32_t* wide, const uint32_t* zero_word, SideResult& res,
                               const uint32_t* key_hi = nullptr, uint32_t* cursor_zeroed = nullptr) {
    const PassPlan& p1 = plan.passes[0];
    const PassPlan& p2 = plan.passes[1];
    const uint32_t F1 = p1.cfg.F, G = p2.G, F2 = p2.cfg.F;
    uint32_t chunk_rows = 0;
    const uint32_t nc = narrow_chunks(n_rows, &chunk_rows);
    const uint32_t nsub = (p2r_variant() == 0 || F2 > (uint32_t)P2R_MAX_F) ? 1u : narrow_nsub();   // (the write-combining second pass reads one extent per group)
    // k_part1_narrow + k_part2_ring hand the HASH of the key on instead of the key (a bijection of the 32 key bits): the ring pass
    // and the join need not hash again, equality is the same.  (The A/B knobs that pair an old kernel with a new one keep keys.)
    const bool hashed_tuples = p1n_variant() != 0 && p2r_variant() != 0 && F2 <= (uint32_t)P2R_MAX_F;
    const uint32_t cap1 = (uint32_t)reserve_cap_sub(n_rows, F1, nsub, nc, chunk_rows, key_hi ? DICT_SIGMA_MUL : 1u);
    if ((uint64_t)cap1 * F1 * nsub >= 0xFFFFFFF0ull) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "narrow pipeline: first-pass positions exceed u32");
    uint32_t* cursor = cursor_zeroed ? cursor_zeroed : (uint32_t*)ws_take(ctx, (uint64_t)F1 * nsub * 4);
    uint32_t* c_off = (uint32_t*)ws_take(ctx, (uint64_t)nc * 4);
    uint32_t* c_len = (uint32_t*)ws_take(ctx, (uint64_t)nc * 4);
    uint32_t* slot_base = (uint32_t*)ws_take(ctx, (uint64_t)(G + 1) * 4);
    uint32_t* slot_cap = (uint32_t*)ws_take(ctx, (uint64_t)G * 4);
    uint32_t* sub_start = (uint32_t*)ws_take(ctx, (uint64_t)G * F2 * 4);
    uint32_t* sub_len = (uint32_t*)ws_take(ctx, (uint64_t)G * F2 * 4);
    if (!cursor || !c_off || !c_len || !slot_base || !slot_cap || !sub_start || !sub_len)
        return fail(ctx, GOLAP_ERR_OOM, "join workspace exhausted (narrow pipeline arrays)");
    if (!cursor_zeroed) GOLAP_TRY(zero_async(ctx, st, (void*)cursor, (uint64_t)F1 * nsub * 4));
    hipLaunchKernelGGL(k_setup_uniform, dim3((uint32_t)ceil_div(nc, 256)), dim3(256), 0, st, (uint32_t)n_rows, chunk_rows, nc, c_off, c_len);
    {
        PassCfg cfg = p1.cfg;
        cfg.S = nc;
        cfg.key_hi = key_hi;
        cfg.pre_hashed = hashed_tuples ? 1u : 0u;
        KernelTimer t(ctx, "radix_scatter_p1", st, 0, 16 * n_rows, false);   // credited unless the attempt was abandoned
        const ReserveArgs rsa{cursor, cap1, wide, wide, nullptr, nsub};
        launch_part1_narrow(st, in_keys, rid_base, nc, c_off, c_len, tmp_rows, cfg, rsa);
    }
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    GroupSrc groups{nullptr, 0, cursor, cap1, zero_word, nsub};
    uint32_t sigmas, pad;
    provision_of("GOLAP_SLOT_SIGMAS", sigmas, pad);   // (test knob: 0 forces overflows)
    hipLaunchKernelGGL(k_slots_setup, dim3(1), dim3(1024), 0, st, groups, G, F2, sigmas, pad, slot_base, slot_cap,
                       (unsigned long long)final_rows(plan, n_rows, false), wide, key_hi, DICT_SIGMA_MUL);
    {
        FixedArgs fx{groups, slot_base, slot_cap, sub_start, sub_len, wide, wide};
        PassCfg cfg = p2.cfg;
        cfg.S = 1;
        cfg.key_hi = key_hi;
        cfg.pre_hashed = hashed_tuples ? 1u : 0u;
        KernelTimer t(ctx, "radix_scatter_p2", st, 0, 16 * n_rows, false);
        if (p2r_variant() != 0 && F2 <= (uint32_t)P2R_MAX_F)
            launch_part2_ring(st, G, tmp_rows, final_rows_ptr, cfg, fx, SegSrc{nullptr, nullptr, 0});
        else
            hipLaunchKernelGGL((k_scatter_wc<2, true, 16, true, WCN_RPT, WCN_CAP>), dim3(G), dim3(WC_THREADS), scatter_wc_lds_bytes(F2, WCN_CAP, 8), st, (const long long*)tmp_rows,
                               (const uint32_t*)nullptr, rid_base, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr,
                               (long long*)final_rows_ptr, (uint32_t*)nullptr, cfg, fx, (const uint32_t*)nullptr);
    }
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    res = SideResult();
    res.keys = (const long long*)final_rows_ptr;
    res.alt_start = sub_start;
    res.alt_len = sub_len;
    res.flag = zero_word;
    res.positions = final_rows(plan, n_rows, false);
    return GOLAP_OK;
}

// narrow tuples through the exchange-time partition / the segment-fed local passes need the kernels that have a tuple form
static bool partition_narrow_ok(uint32_t F, bool has_rids) { return use_big_scatter(F, false, has_rids); }
static bool local_narrow_ok(const JoinPlan& plan) {
    for (const PassPlan& pp : plan.passes)
        if (pp.cfg.F < 32 || pp.cfg.F > (uint32_t)WC_MAX_F) return false;
    const char* env = knob("GOLAP_SCATTER");
    return !(env && !strcmp(env, "v1"));
}

// Narrow keys are tried when both sides take the reserving + no-histogram flow (large two-pass joins).
static bool use_narrow_keys(const JoinPlan& plan, uint64_t n_build, uint64_t n_probe) {
    const char* env = knob("GOLAP_NARROW");   // "0" disables (A/B hook)
    if (env && !strcmp(env, "0")) return false;
    return use_reserve_first_pass(plan, n_build, false, false) && use_reserve_first_pass(plan, n_probe, false, false);
}

static int32_t radix_hash_join(golap_ctx* ctx, const golap_join_params* p, hipStream_t st) {
    if (p->n_build >= 0xFFFFFFF0ull || p->n_probe >= 0xFFFFFFF0ull)
        return fail(ctx, GOLAP_ERR_UNSUPPORTED, "row ids are u32 (join_kernels.cuh:48): at most 2^32-16 rows per side");
    if ((uint64_t)p->build_rid_base + p->n_build > 0xFFFFFFFFull || (uint64_t)p->probe_rid_base + p->n_probe > 0xFFFFFFFFull)
        return fail(ctx, GOLAP_ERR_UNSUPPORTED, "rid_base + rows exceeds the u32 row-id range (0xFFFFFFFF is the NULL row id)");
    GOLAP_TRY(ensure_scatter_lds(ctx));
    ctx->prof.cur_gate_host = nullptr;
    JoinPlan plan;
    make_plan(p->n_build, 0, true, plan);
    // positions inside the slot-padded layouts are u32 as well: near 2^32 rows fall back to the dense (histogram) layouts
    if (final_rows(plan, std::max(p->n_build, p->n_probe), false) >= 0xFFFFFFF0ull) plan.allow_slots = false;
    const size_t np = plan.passes.size();
    GOLAP_TRY(ws_begin(ctx, radix_join_ws_bytes(plan, p->n_build, p->n_probe, 1, false) + final_rows(plan, p->n_probe, false) / 8 + 4096, st));
    SideBuffers bb, pb;
    const uint64_t fb = final_rows(plan, p->n_build, false), fp = final_rows(plan, p->n_probe, false);
    bb.final_keys = ws_take(ctx, fb * 8);
    bb.final_rids = ws_take(ctx, fb * 4);
    pb.final_keys = ws_take(ctx, fp * 8);
    pb.final_rids = ws_take(ctx, fp * 4);
    const size_t n_tmp = np >= 3 ? 2 : (np == 2 ? 1 : 0);
    const uint64_t tmax = std::max(pass1_rows(plan, p->n_build, false), pass1_rows(plan, p->n_probe, false));
    for (size_t i = 0; i < n_tmp; ++i) {
        bb.tmp_keys[i] = pb.tmp_keys[i] = ws_take(ctx, tmax * 8);
        bb.tmp_rids[i] = pb.tmp_rids[i] = ws_take(ctx, tmax * 4);
        if (!bb.tmp_keys[i] || !bb.tmp_rids[i]) return fail(ctx, GOLAP_ERR_OOM, "join workspace exhausted");
    }
    if (!bb.final_keys || !bb.final_rids || !pb.final_keys || !pb.final_rids) return fail(ctx, GOLAP_ERR_OOM, "join workspace exhausted");
    // With >= 3 passes the ping-pong alternates tmp[0], tmp[1], tmp[0]...; with an odd
    // number of non-final passes beyond 2 the k&1 indexing below stays within the 2 temps.
    JoinPlan plan_b = plan, plan_p = plan;
    SideResult rb, rp;
    const uint32_t* gate = nullptr;
    uint32_t* probe_hit = nullptr;
    if (use_narrow_keys(plan, p->n_build, p->n_probe)) {
        // optimistic 32-bit pipeline first (its big buffers alias the 64-bit pipeline's: only one of the two does any work)
        // [0] wide flag, [1] constant 0, [2 ..) PassCfg::key_hi: the dictionary of upper key halves ([2 .. 5] the scalar form's, [6] its size, [10 ..) the big one)
        // (the control words in a 256-byte line of their own, then both sides' reservation cursors: ONE block, cleared by one launch)
        const uint64_t words_bytes = round_up((2 + (uint64_t)KEY_DICT_WORDS) * 4, 256);
        const uint64_t cur_bytes = round_up((uint64_t)plan.passes[0].cfg.F * narrow_nsub() * 4, 256);
        uint32_t* words = (uint32_t*)ws_take(ctx, words_bytes + 2 * cur_bytes);
        if (!words) return fail(ctx, GOLAP_ERR_OOM, "join workspace exhausted");
        GOLAP_TRY(zero_async(ctx, st, (void*)words, words_bytes + 2 * cur_bytes));
        uint32_t* const cursors[2] = {words + words_bytes / 4, words + (words_bytes + cur_bytes) / 4};
        if (ctx->prof.enabled) {   // profiling: which of the two pipelines did the work is read back (asynchronously) for the byte credits
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
            ctx->prof.cur_gate_host = cs == hipStreamCaptureStatusNone ? ctx->prof.take_gate_slot() : nullptr;
        }
        // Frame of reference: the tuples carry the lower 32 key bits, valid while every key of BOTH sides shares its upper 32 bits
        // with the first build key (0 for the usual small non-negative ids, 0xFFFFFFFF for small negative ones, anything for keys
        // offset into a common 4 Gi window).  Only the new tuple kernels know about it (knobs that select the old ones keep 0).
        const uint32_t* key_hi = nullptr;
        if (p1n_variant() != 0 && p2r_variant() != 0 && plan.passes[1].cfg.F <= (uint32_t)P2R_MAX_F) {
            hipLaunchKernelGGL(k_key_dict, dim3(1), dim3(1024), 0, st, (const long long*)p->build_key_ptr, (unsigned long long)p->n_build,
                               (const long long*)p->probe_key_ptr, (unsigned long long)p->n_probe, words + 2, words,
                               (p1n_variant() == 118 && ((p->build_key_ptr | p->probe_key_ptr) & 15) == 0) ? KEY_DICT_MAX : 1u,   // (launch_part1_narrow's dictionary forms)
                               part1_offset_form_ok(p->build_key_ptr | p->probe_key_ptr) ? plan.passes[0].cfg.F : 0u, key_dict_samples());
            key_hi = words + 2;
        }
        if (p->join_type & 2) {
            const uint64_t nw = ceil_div(std::max<uint64_t>(1, std::max(p->n_probe, fp)), 32) + 1;
            probe_hit = (uint32_t*)ws_take(ctx, nw * 4);
            if (!probe_hit) return fail(ctx, GOLAP_ERR_OOM, "join workspace exhausted (probe bitmap)");
            GOLAP_TRY(zero_async(ctx, st, (void*)probe_hit, nw * 4));
        }
        SideResult nb_, np_;
        GOLAP_TRY(run_side_narrow(ctx, st, plan, p->build_key_ptr, p->build_rid_base, p->n_build, bb.tmp_keys[0], bb.final_keys, words, words + 1, nb_, key_hi, cursors[0]));
        GOLAP_TRY(run_side_narrow(ctx, st, plan, p->probe_key_ptr, p->probe_rid_base, p->n_probe, pb.tmp_keys[0], pb.final_keys, words, words + 1, np_, key_hi, cursors[1]));
        GOLAP_TRY(launch_lds_join(ctx, st, plan, nb_, np_, p->n_build, p->n_probe, p->out_pairs_ptr, p->out_count_ptr, p->max_pairs, p->join_type,
                                  true, nullptr, words, probe_hit, key_hi));   // (key_hi is only set when both passes are the hashed-tuple kernels)
        gate = words;
    }
    // (profiling contexts) the 64-bit pipeline standing by is not timed while recent joins did not need it: see Profiler::mute
    struct MuteScope { Profiler& pr; bool prev; ~MuteScope() { pr.mute = prev; } } mute_scope{ctx->prof, ctx->prof.mute};
    {
        if (gate && ctx->prof.enabled && ctx->prof.light && ctx->prof.cur_gate_host && ctx->prof.recent_gate() == 0) ctx->prof.mute = true;
    }
    GOLAP_TRY(run_side(ctx, st, "build", plan_b, p->build_key_ptr, 0, p->build_rid_base, p->n_build, nullptr, bb, rb, gate));
    GOLAP_TRY(run_side(ctx, st, "probe", plan_p, p->probe_key_ptr, 0, p->probe_rid_base, p->n_probe, nullptr, pb, rp, gate));
    GOLAP_TRY(launch_lds_join(ctx, st, plan, rb, rp, p->n_build, p->n_probe, p->out_pairs_ptr, p->out_count_ptr, p->max_pairs, p->join_type, false, gate,
                              nullptr, probe_hit));
    ctx->prof.mute = mute_scope.prev;
    if (gate && ctx->prof.cur_gate_host) {
        (void)hipMemcpyAsync((void*)ctx->prof.cur_gate_host, gate, 4, hipMemcpyDeviceToHost, st);
        ctx->prof.note_gate_slot(ctx->prof.cur_gate_host);
        ctx->prof.cur_gate_host = nullptr;
    }
    ws_end(ctx, st);
    return GOLAP_OK;
}

static int32_t broadcast_hash_join(golap_ctx* ctx, const golap_join_params* p, hipStream_t st) {
    if (p->n_build >= 0xFFFFFFF0ull / 2 || p->n_probe >= 0xFFFFFFF0ull)
        return fail(ctx, GOLAP_ERR_UNSUPPORTED, "BroadcastHashJoin: side too large for u32 row ids");
    const uint32_t nb = (uint32_t)p->n_build, npr = (uint32_t)p->n_probe;
    const uint32_t capacity = std::max<uint32_t>(1, nb + nb / 2);   // README.md:114
    const bool keep_b = (p->join_type & 1) != 0;
    GOLAP_TRY(ws_begin(ctx, (uint64_t)capacity * 4 + (uint64_t)nb * 4 + (keep_b ? nb : 0) + 8192, st));
    uint32_t* buckets = (uint32_t*)ws_take(ctx, (uint64_t)capacity * 4);
    uint32_t* next = (uint32_t*)ws_take(ctx, (uint64_t)std::max<uint32_t>(nb, 1) * 4);
    uint8_t* hit = keep_b ? (uint8_t*)ws_take(ctx, std::max<uint32_t>(nb, 1)) : nullptr;
    if (!buckets || !next || (keep_b && !hit)) return fail(ctx, GOLAP_ERR_OOM, "join workspace exhausted");
    if (keep_b) GOLAP_TRY(zero_async(ctx, st, (void*)(hit), std::max<uint32_t>(nb, 1)));
    const uint32_t cus = (uint32_t)ctx->prop.multiProcessorCount;
    {
        KernelTimer t(ctx, "bcast_build_table", st, 8ull * nb);
        hipLaunchKernelGGL(k_fill_u32, dim3(std::min<uint32_t>(cus * 8, (uint32_t)ceil_div(capacity, 256))), dim3(256), 0, st, buckets, capacity,
                           JN_EMPTY);
        if (nb)
            hipLaunchKernelGGL(k_bcast_build, dim3(std::min<uint32_t>(cus * 8, (uint32_t)ceil_div(nb, 256))), dim3(256), 0, st,
                               (const long long*)p->build_key_ptr, nb, buckets, capacity, next);
    }
    if (npr) {
        KernelTimer t(ctx, "bcast_probe_table", st, 8ull * npr);
        const uint32_t batch = BC_THREADS * BC_R;
        const dim3 grid(std::min<uint32_t>(cus * 4, (uint32_t)ceil_div(npr, batch))), block(BC_THREADS);
#define GOLAP_BC(JT)                                                                                                                       \
    hipLaunchKernelGGL(k_bcast_probe<JT>, grid, block, 0, st, (const long long*)p->build_key_ptr, buckets, capacity, next,                 \
                       (const long long*)p->probe_key_ptr, npr, p->build_rid_base, p->probe_rid_base, (uint2*)p->out_pairs_ptr,            \
                       (unsigned long long*)p->out_count_ptr, (unsigned long long)p->max_pairs, hit)
        switch (p->join_type) {
            case 0: GOLAP_BC(0); break;
            case 1: GOLAP_BC(1); break;
            case 2: GOLAP_BC(2); break;
            default: GOLAP_BC(3); break;
        }
#undef GOLAP_BC
    }
    if (keep_b && nb) {
        KernelTimer t(ctx, "bcast_unmatched_build", st, (uint64_t)nb);
        const uint32_t batch = BC_THREADS * BC_R;
        hipLaunchKernelGGL(k_bcast_unmatched_build, dim3(std::min<uint32_t>(cus * 4, (uint32_t)ceil_div(nb, batch))), dim3(BC_THREADS), 0, st, hit, nb,
                           p->build_rid_base, (uint2*)p->out_pairs_ptr, (unsigned long long*)p->out_count_ptr, (unsigned long long)p->max_pairs);
    }
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    ws_end(ctx, st);
    return GOLAP_OK;
}

// ---------------------------------------------------------------------------
// Composite keys and NULLs.  physical_plan.rs:35-36 carries Vec<PhysicalExpr> keys and
// column_buffer.rs:59-60 a byte-per-row validity; the reference's kernels and executor use neither (one Int64 key,
// executor.rs:245; nulls -> 0, record_batch_convert.rs:102-110).  Here, without touching the partition / LDS-join kernels:
//   k_surrogate   one 64-bit SURROGATE key per row: the key itself (one column, row valid), a 64-bit mix of the columns
//                 (composite key), or -- for a row with a NULL in any key column -- a mix of (side, row index): NULL rows get
//                 pairwise different keys that are also different per side, so they match nothing, they do not pile up in one
//                 partition, and an outer join emits them as unmatched rows by itself;
//   the ordinary join runs on the surrogates (row ids stay positions: nothing is compacted, nothing synchronises);
//   k_verify      every emitted pair is checked against the REAL columns and validity bytes.  A surrogate is 64 bits: two
//                 different composite keys (or a NULL row and a real key) collide with probability 2^-64 per candidate pair,
//                 i.e. ~0.05 expected false pairs in a 1B x 1B join -- rare, not never.  False pairs are recorded (<= 64);
//   k_repair_scan / k_repair_fix (no-ops unless a false pair was found): the false pair is removed; in an outer join a row
//                 that thereby lost its only match gets its unmatched pair.  More than 64 false pairs (probability < 1e-60 with
//                 64-bit surrogates; reachable with the test knob): the count word is set to 2^64-1, an impossible count.
// ---------------------------------------------------------------------------
struct KeyCols {
    const long long* col[4];
    const uint8_t* valid[4];
    uint32_t n_cols;
};

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// keep_mask: all ones; the test knob GOLAP_GJ_HASH_BITS narrows the surrogates so that false pairs actually occur
__global__ __launch_bounds__(256) void k_surrogate(KeyCols k, unsigned long long n, unsigned long long side_seed, unsigned long long keep_mask,
                                                   long long* __restrict__ out) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        bool null = false;
        for (uint32_t c = 0; c < k.n_cols; ++c) null = null || (k.valid[c] && !k.valid[c][i]);
        unsigned long long s;
        if (null) {
            s = mix64(side_seed + i * 0x9E3779B97F4A7C15ull);
        } else if (k.n_cols == 1) {
            s = (unsigned long long)k.col[0][i];
        } else {
            s = 0x243F6A8885A308D3ull;
            for (uint32_t c = 0; c < k.n_cols; ++c) s = mix64(s ^ (unsigned long long)k.col[c][i]) + 0x9E3779B97F4A7C15ull * (c + 1);
        }
        out[i] = (long long)(s & keep_mask);
    }
}

constexpr uint32_t GJ_MAX_HOLES = 64;
struct GjCtl {                 // device control block of one generalised join
    uint32_t n_holes;          // false pairs found (may exceed GJ_MAX_HOLES: then only the first ones are listed and `overflow` is set)
    uint32_t overflow;
    uint32_t hole_pos[GJ_MAX_HOLES];     // position of the false pair in out_pairs
    uint32_t hole_b[GJ_MAX_HOLES], hole_p[GJ_MAX_HOLES];
    uint32_t b_has_other[GJ_MAX_HOLES], p_has_other[GJ_MAX_HOLES];   // a surviving pair with the same build / probe row exists
};

__device__ __forceinline__ bool gj_real_match(const KeyCols& B, const KeyCols& P, uint32_t bi, uint32_t pi) {
    for (uint32_t c = 0; c < B.n_cols; ++c) {
        if (B.valid[c] && !B.valid[c][bi]) return false;
        if (P.valid[c] && !P.valid[c][pi]) return false;
        if (B.col[c][bi] != P.col[c][pi]) return false;
    }
    return true;
}

__global__ __launch_bounds__(256) void k_gj_verify(const uint2* __restrict__ pairs, const unsigned long long* __restrict__ count, unsigned long long max_pairs,
                                                   KeyCols B, KeyCols P, uint32_t b_base, uint32_t p_base, GjCtl* __restrict__ ctl) {
    const unsigned long long n = *count < max_pairs ? *count : max_pairs;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        const uint2 pr = pairs[i];
        if (pr.x == JN_NULL || pr.y == JN_NULL) continue;
        if (!gj_real_match(B, P, pr.x - b_base, pr.y - p_base)) {
            const uint32_t h = atomicAdd(&ctl->n_holes, 1u);
            if (h < GJ_MAX_HOLES) { ctl->hole_pos[h] = (uint32_t)i; ctl->hole_b[h] = pr.x; ctl->hole_p[h] = pr.y; }
            else ctl->overflow = 1u;
        }
    }
}

// only when false pairs exist: which of their rows still have a true partner among the surviving pairs?
__global__ __launch_bounds__(256) void k_gj_repair_scan(const uint2* __restrict__ pairs, const unsigned long long* __restrict__ count, unsigned long long max_pairs,
                                                        KeyCols B, KeyCols P, uint32_t b_base, uint32_t p_base, GjCtl* __restrict__ ctl) {
    const uint32_t nh = ctl->n_holes < GJ_MAX_HOLES ? ctl->n_holes : GJ_MAX_HOLES;
    if (nh == 0) return;
    const unsigned long long n = *count < max_pairs ? *count : max_pairs;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        const uint2 pr = pairs[i];
        if (pr.x == JN_NULL || pr.y == JN_NULL) continue;
        for (uint32_t h = 0; h < nh; ++h) {
            if (pr.x != ctl->hole_b[h] && pr.y != ctl->hole_p[h]) continue;
            if (!gj_real_match(B, P, pr.x - b_base, pr.y - p_base)) continue;   // (another false pair)
            if (pr.x == ctl->hole_b[h]) ctl->b_has_other[h] = 1u;
            if (pr.y == ctl->hole_p[h]) ctl->p_has_other[h] = 1u;
        }
    }
}

// one thread: false pairs leave the list; in an outer join a row that lost its only partner gets its unmatched pair
__global__ void k_gj_repair_fix(uint2* __restrict__ pairs, unsigned long long* __restrict__ count, unsigned long long max_pairs, int join_type,
                                GjCtl* __restrict__ ctl) {
    if (blockIdx.x || threadIdx.x) return;
    if (ctl->overflow) { *count = ~0ull; return; }   // more false pairs than can be repaired: an impossible count tells the caller (golap.h)
    const uint32_t nh = ctl->n_holes < GJ_MAX_HOLES ? ctl->n_holes : GJ_MAX_HOLES;
    if (nh == 0) return;
    unsigned long long cnt = *count;
    const bool keep_b = (join_type & 1) != 0, keep_p = (join_type & 2) != 0;
    // replacements first (they reuse the holes' places)
    uint32_t hole_free[GJ_MAX_HOLES];
    for (uint32_t h = 0; h < nh; ++h) hole_free[h] = 1u;
    auto put = [&](uint2 v) {
        for (uint32_t h = 0; h < nh; ++h)
            if (hole_free[h]) { pairs[ctl->hole_pos[h]] = v; hole_free[h] = 0u; return; }
        if (cnt < max_pairs) pairs[cnt] = v;
        ++cnt;
    };
    for (uint32_t h = 0; h < nh; ++h) {
        bool first_b = true, first_p = true;   // a row may sit in several false pairs: one unmatched pair at most
        for (uint32_t g = 0; g < h; ++g) {
            first_b = first_b && ctl->hole_b[g] != ctl->hole_b[h];
            first_p = first_p && ctl->hole_p[g] != ctl->hole_p[h];
        }
        if (keep_b && first_b && !ctl->b_has_other[h]) put(make_uint2(ctl->hole_b[h], JN_NULL));
        if (keep_p && first_p && !ctl->p_has_other[h]) put(make_uint2(JN_NULL, ctl->hole_p[h]));
    }
    // remaining holes: filled from the end of the list (highest hole first, so a moved pair is never itself a hole)
    for (;;) {
        uint32_t best = 0xFFFFFFFFu;
        for (uint32_t h = 0; h < nh; ++h)
            if (hole_free[h] && (best == 0xFFFFFFFFu || ctl->hole_pos[h] > ctl->hole_pos[best])) best = h;
        if (best == 0xFFFFFFFFu) break;
        hole_free[best] = 0u;
        --cnt;
        if (cnt < max_pairs && ctl->hole_pos[best] != cnt) pairs[ctl->hole_pos[best]] = pairs[cnt];
    }
    *count = cnt;
}

static int32_t radix_hash_join(golap_ctx* ctx, const golap_join_params* p, hipStream_t st);
static int32_t broadcast_hash_join(golap_ctx* ctx, const golap_join_params* p, hipStream_t st);

static bool join_is_general(const golap_join_params* p) {
    if (p->n_key_cols > 1) return true;
    for (int c = 0; c < 4; ++c)
        if (p->build_validity[c] || p->probe_validity[c]) return true;
    return false;
}

static int32_t general_join(golap_ctx* ctx, golap_join_strategy strategy, const golap_join_params* p, hipStream_t st) {
    const uint32_t nc = p->n_key_cols ? p->n_key_cols : 1;
    KeyCols B{}, P{};
    B.n_cols = P.n_cols = nc;
    for (uint32_t c = 0; c < nc; ++c) {
        B.col[c] = (const long long*)(c ? p->build_key_extra[c - 1] : p->build_key_ptr);
        P.col[c] = (const long long*)(c ? p->probe_key_extra[c - 1] : p->probe_key_ptr);
        B.valid[c] = (const uint8_t*)p->build_validity[c];
        P.valid[c] = (const uint8_t*)p->probe_validity[c];
        if ((p->n_build && !B.col[c]) || (p->n_probe && !P.col[c])) return fail(ctx, GOLAP_ERR_INVALID_ARG, "key column %u is null", c);
    }
    const uint64_t nb = p->n_build, np = p->n_probe;
    GOLAP_TRY(aux_begin(ctx, round_up(std::max<uint64_t>(nb, 1) * 8, 256) + round_up(std::max<uint64_t>(np, 1) * 8, 256) + 8192, st));
    long long* sb = (long long*)aux_take(ctx, std::max<uint64_t>(nb, 1) * 8);
    long long* sp = (long long*)aux_take(ctx, std::max<uint64_t>(np, 1) * 8);
    GjCtl* ctl = (GjCtl*)aux_take(ctx, sizeof(GjCtl));
    if (!sb || !sp || !ctl) return fail(ctx, GOLAP_ERR_OOM, "join scratch exhausted (surrogate keys)");
    GOLAP_TRY(zero_async(ctx, st, ctl, sizeof(GjCtl)));
    const uint32_t cus = (uint32_t)ctx->prop.multiProcessorCount;
    auto grid_for = [&](uint64_t n) { return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)cus * 8, ceil_div(std::max<uint64_t>(n, 1), 1024))); };
    {
        KernelTimer t(ctx, "join_surrogate_keys", st, (8ull * nc + 8) * (nb + np));
        unsigned long long keep = ~0ull;
        if (const char* e = knob("GOLAP_GJ_HASH_BITS")) { const int b = atoi(e); if (b >= 1 && b < 64) keep = (1ull << b) - 1; }   // test knob: provoke false pairs
        if (nb) hipLaunchKernelGGL(k_surrogate, dim3(grid_for(nb)), dim3(256), 0, st, B, (unsigned long long)nb, 0x6A09E667F3BCC908ull, keep, sb);
        if (np) hipLaunchKernelGGL(k_surrogate, dim3(grid_for(np)), dim3(256), 0, st, P, (unsigned long long)np, 0xBB67AE8584CAA73Bull, keep, sp);
    }
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    golap_join_params q = *p;
    q.build_key_ptr = (uint64_t)sb;
    q.probe_key_ptr = (uint64_t)sp;
    q.n_key_cols = 1;
    for (int c = 0; c < 4; ++c) q.build_validity[c] = q.probe_validity[c] = 0;
    int32_t rc;
    if (nb == 0 || np == 0) {
        const bool keep = nb ? (p->join_type & 1) != 0 : (np && (p->join_type & 2) != 0);
        rc = keep ? broadcast_hash_join(ctx, &q, st) : GOLAP_OK;
    } else {
        rc = strategy == GOLAP_JOIN_BROADCAST_HASH ? broadcast_hash_join(ctx, &q, st) : radix_hash_join(ctx, &q, st);
    }
    if (rc == GOLAP_OK && nb && np) {
        KernelTimer t(ctx, "join_verify_pairs", st, 0);
        const uint32_t g = grid_for(std::min<uint64_t>(p->max_pairs, (uint64_t)cus * 8 * 1024 * 64));
        hipLaunchKernelGGL(k_gj_verify, dim3(g), dim3(256), 0, st, (const uint2*)p->out_pairs_ptr, (const unsigned long long*)p->out_count_ptr,
                           (unsigned long long)p->max_pairs, B, P, p->build_rid_base, p->probe_rid_base, ctl);
        hipLaunchKernelGGL(k_gj_repair_scan, dim3(g), dim3(256), 0, st, (const uint2*)p->out_pairs_ptr, (const unsigned long long*)p->out_count_ptr,
                           (unsigned long long)p->max_pairs, B, P, p->build_rid_base, p->probe_rid_base, ctl);
        hipLaunchKernelGGL(k_gj_repair_fix, dim3(1), dim3(64), 0, st, (uint2*)p->out_pairs_ptr, (unsigned long long*)p->out_count_ptr,
                           (unsigned long long)p->max_pairs, p->join_type, ctl);
        GOLAP_HIP_TRY(ctx, hipGetLastError());
    }
    aux_end(ctx, st);
    return rc;
}

}  // namespace golap

using namespace golap;

extern "C" {

int32_t golap_join(golap_ctx* ctx, golap_join_strategy strategy, const golap_join_params* p, void* stream) {
    if (!ctx || !p) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    if (!p->out_count_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "out_count_ptr is null");
    if (p->max_pairs && !p->out_pairs_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "out_pairs_ptr is null but max_pairs > 0");
    if ((p->n_build && !p->build_key_ptr) || (p->n_probe && !p->probe_key_ptr)) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null key column");
    if (strategy == GOLAP_JOIN_SORT_MERGE)
        return fail(ctx, GOLAP_ERR_UNSUPPORTED, "SortMergeJoin is out of scope (the planner never emits it, physical_plan.rs:144)");
    if (strategy != GOLAP_JOIN_RADIX_HASH && strategy != GOLAP_JOIN_BROADCAST_HASH)
        return fail(ctx, GOLAP_ERR_INVALID_ARG, "unknown join strategy %d", (int)strategy);
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick_stream(ctx, stream);
    GOLAP_TRY(zero_async(ctx, st, (void*)((void*)p->out_count_ptr), 8));
    if (p->join_type < 0 || p->join_type > 3) return fail(ctx, GOLAP_ERR_INVALID_ARG, "bad join_type %d", p->join_type);
    if (p->n_key_cols > 4) return fail(ctx, GOLAP_ERR_INVALID_ARG, "at most 4 key columns");
    if (join_is_general(p)) return general_join(ctx, strategy, p, st);
    if (p->n_build == 0 || p->n_probe == 0) {
        // executor.rs:231-233: nothing to join.  An outer join still keeps the non-empty preserved side.
        const bool keep = p->n_build ? (p->join_type & 1) != 0 : (p->n_probe && (p->join_type & 2) != 0);
        if (!keep) return GOLAP_OK;
        return broadcast_hash_join(ctx, p, st);   // one table pass handles an empty side
    }
    if (strategy == GOLAP_JOIN_BROADCAST_HASH) return broadcast_hash_join(ctx, p, st);
    return radix_hash_join(ctx, p, st);
}

uint64_t golap_join_scratch_bytes(golap_ctx* ctx, uint64_t n_build, uint64_t n_probe) {
    (void)ctx;
    JoinPlan plan;
    make_plan(n_build, 0, true, plan);
    return radix_join_ws_bytes(plan, n_build, n_probe, 1, false);
}

int32_t golap_join_plan_get(golap_ctx* ctx, uint64_t n_build, uint32_t n_owners, golap_join_plan* out) {
    if (!out) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    if (n_owners == 0 || (n_owners & (n_owners - 1)) || n_owners > 256) return fail(ctx, GOLAP_ERR_INVALID_ARG, "n_owners must be a power of two <= 256");
    JoinPlan plan;
    make_plan(n_build, ilog2(n_owners), true, plan);
    if (final_rows(plan, n_build, false) >= 0xFFFFFFF0ull) plan.allow_slots = false;
    memset(out, 0, sizeof *out);
    out->bits1 = plan.passes[0].bits;
    out->bits2 = plan.total_bits - plan.passes[0].bits;
    out->chunks1 = first_pass_chunks(n_build, nullptr);
    // first-pass tile: big-tile kernel = 24 keys per thread, 1024 threads for large inputs (launch_hist_scatter)
    out->tile_rows = !use_big_scatter(plan.passes[0].cfg.F, false, false) ? PT_TILE
                     : (out->chunks1 >= 224 && n_build / out->chunks1 >= 4 * 24576 ? 1024 * 24 : PT_THREADS * 24);
    out->slots2 = plan.passes.size() < 2 ? 0 : (use_fast_last_pass(plan, false) ? 1 : slots_for(plan.passes[1].G, n_build));
    out->no_histogram_last_pass = use_fast_last_pass(plan, false) ? 1 : 0;
    out->no_histogram_first_pass = use_reserve_first_pass(plan, n_build, false, false) ? 1 : 0;
    out->optimistic_narrow_keys = use_narrow_keys(plan, n_build, n_build) ? 1 : 0;   // (for a probe side of the same size class)
    out->table_capacity = JN_CAP;
    out->n_subpartitions = plan.n_sub;
    return GOLAP_OK;
}

// radix_partition_kernel as a stand-alone step: the reference's 256 partitions (join_kernels.cuh:22-23, :69),
// optionally refined by `extra_bits` more hash bits so the receiving rank needs one local pass less.
static int32_t radix_partition_impl(golap_ctx* ctx, const golap_partition_params* p, hipStream_t st);
int32_t golap_radix_partition(golap_ctx* ctx, const golap_partition_params* p, void* stream) {
    if (!ctx || !p) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return radix_partition_impl(ctx, p, pick_stream(ctx, stream));
}
// (call with ctx->mu held)
static int32_t radix_partition_impl(golap_ctx* ctx, const golap_partition_params* p, hipStream_t st) {
    if (!p->out_offsets_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "out_offsets_ptr is null");
    if (p->n_rows && (!p->key_ptr || !p->out_key_ptr || (!p->out_row_id_ptr && !p->narrow))) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null column pointer");
    if (p->narrow && (p->row_id_ptr || !p->wide_flag_ptr)) return fail(ctx, GOLAP_ERR_INVALID_ARG, "narrow partition needs implicit row ids and a wide_flag_ptr");
    if (p->n_owners == 0 || (p->n_owners & (p->n_owners - 1)) || p->n_owners > 256)
        return fail(ctx, GOLAP_ERR_INVALID_ARG, "n_owners must be a power of two <= 256");
    if (p->extra_bits > 2) return fail(ctx, GOLAP_ERR_INVALID_ARG, "extra_bits must be 0..2");
    if (p->n_rows >= 0xFFFFFFF0ull) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "row ids are u32: at most 2^32-16 rows");
    GOLAP_TRY(ensure_scatter_lds(ctx));
    const uint32_t F = 256u << p->extra_bits;
    JoinPlan plan;
    PassPlan pp;
    memset(&pp, 0, sizeof pp);
    pp.bits = 8 + p->extra_bits;
    pp.cfg.F = F;
    pp.cfg.mode = 1;
    pp.cfg.e1 = p->extra_bits;
    pp.cfg.owners_log2 = ilog2(p->n_owners);
    pp.G = 1;
    plan.passes.push_back(pp);
    plan.total_bits = pp.bits;
    plan.n_sub = F;
    if (p->n_rows == 0) {
        GOLAP_TRY(zero_async(ctx, st, (void*)((void*)p->out_offsets_ptr), (size_t)(F + 1) * 4));
        return GOLAP_OK;
    }
    GOLAP_TRY(ws_begin(ctx, side_small_bytes(plan, p->n_rows, 1, false) + 4096, st));
    SideBuffers buf;
    buf.final_keys = p->out_key_ptr;
    buf.final_rids = p->out_row_id_ptr;
    SideResult res;
    if (p->narrow && !partition_narrow_ok(F, false)) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "no tuple form of this partition pass");
    GOLAP_TRY(run_side(ctx, st, "part", plan, p->key_ptr, p->row_id_ptr, p->rid_base, p->n_rows, nullptr, buf, res, nullptr, p->narrow ? 1 : 0,
                       (uint32_t*)p->wide_flag_ptr));
    // offsets[s] = scanned[s * S] for s in 0..F (the scan carries the total in its last slot)
    GOLAP_HIP_TRY(ctx, hipMemcpy2DAsync((void*)p->out_offsets_ptr, 4, res.off, (size_t)res.stride * 4, 4, F + 1, hipMemcpyDeviceToDevice, st));
    ws_end(ctx, st);
    return GOLAP_OK;
}

}  // extern "C"

namespace golap {

// State of a partitioned (post-exchange) join between begin / side / finish.
struct PartJoinState {
    bool active = false;
    JoinPlan plan;
    uint32_t n_groups = 0, n_sources = 0;
    uint64_t rows[2] = {0, 0};
    SideBuffers buf[2];
    SideResult res[2];
    bool done[2] = {false, false};
    bool narrow = false;
};

static PartJoinState& pjs_of(golap_ctx* ctx) {
    if (!ctx->pjs) ctx->pjs = std::shared_ptr<void>(new PartJoinState(), [](void* q) { delete (PartJoinState*)q; });
    return *(PartJoinState*)ctx->pjs.get();
}

static void make_local_plan(const golap_join_partitioned_plan* p, JoinPlan& plan) {
    plan = JoinPlan();
    // Sub-partition fan-out: the rows of one group already share pre_bits hash bits (the exchange-time
    // radix partition), so the local passes start at shift pre_bits.
    const uint64_t hint = p->total_build_rows_hint ? p->total_build_rows_hint : std::max<uint64_t>(1, p->build_rows);
    const uint64_t per_group = ceil_div(hint, p->n_groups);
    uint32_t need = 0;
    for (; need < 22; ++need) {
        double avg = (double)per_group / (double)(1ull << need);
        if (avg + 4.5 * std::sqrt(avg) <= (double)JN_CAP) break;
    }
    if (need == 0) need = 1;   // always run at least one local pass: it also merges the source segments
    if (const char* env = knob("GOLAP_LOCAL_BITS")) {   // experiment hook: total local bits
        uint32_t v = (uint32_t)strtoul(env, nullptr, 10);
        if (v >= 1 && v <= 20) need = v;
    }
    {
        uint32_t n_pass = (need + 8) / 9;
        uint32_t shift = p->pre_bits, G = p->n_groups, rest = need;
        for (uint32_t k = 0; k < n_pass; ++k) {
            uint32_t b = (rest + (n_pass - k) - 1) / (n_pass - k);
            PassPlan pp;
            memset(&pp, 0, sizeof pp);
            pp.bits = b;
            pp.cfg.F = 1u << b;
            pp.cfg.shift = shift;
            pp.cfg.mode = 0;
            pp.G = G;
            plan.passes.push_back(pp);
            shift += b;
            G *= pp.cfg.F;
            rest -= b;
        }
        plan.total_bits = shift;
        plan.n_sub = G;
    }
}

static int32_t part_begin(golap_ctx* ctx, const golap_join_partitioned_plan* p, hipStream_t st) {
    if (p->n_groups == 0 || p->n_groups > 1024 || (p->n_groups & (p->n_groups - 1))) return fail(ctx, GOLAP_ERR_INVALID_ARG, "n_groups must be a power of two <= 1024");
    if (p->n_sources == 0 || p->n_sources > 64) return fail(ctx, GOLAP_ERR_INVALID_ARG, "n_sources must be in 1..64");
    if (p->pre_bits < 8 || p->pre_bits > 12) return fail(ctx, GOLAP_ERR_INVALID_ARG, "pre_bits must be in 8..12");
    if (p->build_rows >= 0xFFFFFFF0ull || p->probe_rows >= 0xFFFFFFF0ull) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "too many rows for u32 offsets");
    PartJoinState& S = pjs_of(ctx);
    if (S.active) return fail(ctx, GOLAP_ERR_INVALID_ARG, "golap_join_partitioned_begin: a partitioned join is already open on this context (finish or abort it)");
    S = PartJoinState();
    S.n_groups = p->n_groups;
    S.n_sources = p->n_sources;
    S.rows[0] = p->build_rows;
    S.rows[1] = p->probe_rows;
    GOLAP_TRY(ensure_scatter_lds(ctx));
    make_local_plan(p, S.plan);
    S.narrow = p->narrow != 0;
    if (S.narrow && !local_narrow_ok(S.plan)) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "no tuple form of the local passes (golap_join_partitioned_narrow_ok)");
    const size_t np = S.plan.passes.size();
    GOLAP_TRY(ws_begin(ctx, radix_join_ws_bytes(S.plan, std::max<uint64_t>(1, p->build_rows), std::max<uint64_t>(1, p->probe_rows), p->n_sources, true) +
                                (uint64_t)p->n_groups * p->n_sources * 32 + 4096, st));
    const uint64_t nmax = std::max<uint64_t>(1, std::max(p->build_rows, p->probe_rows));
    for (int side = 0; side < 2; ++side) {
        S.buf[side].final_keys = ws_take(ctx, std::max<uint64_t>(1, S.rows[side]) * 8);
        S.buf[side].final_rids = ws_take(ctx, std::max<uint64_t>(1, S.rows[side]) * 4);
        if (!S.buf[side].final_keys || !S.buf[side].final_rids) return fail(ctx, GOLAP_ERR_OOM, "join workspace exhausted");
    }
    // both sides run on the caller's stream one after the other, so they share the ping-pong temporaries
    const size_t n_tmp = np >= 3 ? 2 : (np == 2 ? 1 : 0);
    for (size_t i = 0; i < n_tmp; ++i) {
        const uint64_t tk = ws_take(ctx, nmax * 8), tr = ws_take(ctx, nmax * 4);
        if (!tk || !tr) return fail(ctx, GOLAP_ERR_OOM, "join workspace exhausted");
        S.buf[0].tmp_keys[i] = S.buf[1].tmp_keys[i] = tk;
        S.buf[0].tmp_rids[i] = S.buf[1].tmp_rids[i] = tr;
    }
    S.active = true;
    ctx->ws_owned = true;   // until part_finish / part_abort: ws_begin refuses every other caller (their bump allocation would overwrite this join's rows)
    return GOLAP_OK;
}

static void part_abort(golap_ctx* ctx) {
    pjs_of(ctx).active = false;
    ctx->ws_owned = false;
}

static int32_t part_side(golap_ctx* ctx, int32_t which, const golap_partitioned_side* side, hipStream_t st) {
    PartJoinState& S = pjs_of(ctx);
    if (!S.active) return fail(ctx, GOLAP_ERR_INVALID_ARG, "golap_join_partitioned_side without golap_join_partitioned_begin");
    if (which != 0 && which != 1) return fail(ctx, GOLAP_ERR_INVALID_ARG, "which must be 0 (build) or 1 (probe)");
    if (!side->seg_off || !side->seg_len) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null segment table");
    if (side->n_rows != S.rows[which]) return fail(ctx, GOLAP_ERR_INVALID_ARG, "side row count differs from the plan given to begin");
    const uint32_t nseg = S.n_groups * S.n_sources;
    for (uint32_t i = 0; i < nseg; ++i)
        if (side->seg_off[i] + side->seg_len[i] > side->n_rows) return fail(ctx, GOLAP_ERR_INVALID_ARG, "segment %u exceeds its buffer", i);
    S.done[which] = true;
    if (side->n_rows == 0) return GOLAP_OK;
    std::vector<uint32_t> h_stage((size_t)nseg * 2);   // pageable: hipMemcpyAsync stages it before returning
    for (uint32_t i = 0; i < nseg; ++i) {
        h_stage[i] = (uint32_t)side->seg_off[i];
        h_stage[nseg + i] = (uint32_t)side->seg_len[i];
    }
    uint64_t d_seg = ws_take(ctx, (uint64_t)nseg * 8);
    if (!d_seg) return fail(ctx, GOLAP_ERR_OOM, "join workspace exhausted");
    GOLAP_HIP_TRY(ctx, hipMemcpyAsync((void*)d_seg, h_stage.data(), (size_t)nseg * 8, hipMemcpyHostToDevice, st));
    SegTable seg{d_seg, d_seg + (uint64_t)nseg * 4, S.n_sources};
    JoinPlan plan_side = S.plan;
    GOLAP_TRY(run_side(ctx, st, which ? "probe" : "build", plan_side, side->key_ptr, S.narrow ? 0 : side->row_id_ptr, 0, side->n_rows, &seg, S.buf[which],
                       S.res[which], nullptr, S.narrow ? 2 : 0, nullptr));
    return GOLAP_OK;
}

static int32_t part_finish(golap_ctx* ctx, uint64_t out_pairs, uint64_t out_count, uint64_t max_pairs, hipStream_t st, int32_t join_type = 0) {
    PartJoinState& S = pjs_of(ctx);
    if (!S.active || !S.done[0] || !S.done[1]) return fail(ctx, GOLAP_ERR_INVALID_ARG, "golap_join_partitioned_finish needs begin + both sides");
    GOLAP_TRY(zero_async(ctx, st, (void*)((void*)out_count), 8));
    S.active = false;
    ctx->ws_owned = false;
    if ((S.rows[0] == 0 && !(join_type & 2)) || (S.rows[1] == 0 && !(join_type & 1)) || (S.rows[0] == 0 && S.rows[1] == 0)) { ws_end(ctx, st); return GOLAP_OK; }
    if (S.rows[0] == 0 || S.rows[1] == 0) {
        // an outer join keeps the rows of the non-empty side: the empty side has no partitioned form (run_side was skipped), so
        // describe it as sub-partitions without rows
        SideResult& e = S.res[S.rows[0] == 0 ? 0 : 1];
        const uint64_t n_off = S.plan.n_sub + 1;
        uint32_t* zeros = (uint32_t*)ws_take(ctx, n_off * 4);
        if (!zeros) return fail(ctx, GOLAP_ERR_OOM, "join workspace exhausted");
        GOLAP_TRY(zero_async(ctx, st, zeros, n_off * 4));
        e = SideResult();
        e.off = zeros;
        e.stride = 1;
        e.keys = (const long long*)S.buf[0].final_keys;
        e.rids = (const uint32_t*)S.buf[0].final_rids;
    }
    GOLAP_TRY(launch_lds_join(ctx, st, S.plan, S.res[0], S.res[1], S.rows[0], S.rows[1], out_pairs, out_count, max_pairs, join_type, S.narrow));
    ws_end(ctx, st);
    return GOLAP_OK;
}

}  // namespace golap

extern "C" {

int32_t golap_join_partitioned_narrow_ok(const golap_join_partitioned_plan* p) {
    if (!p || p->n_groups == 0 || p->pre_bits < 8 || p->pre_bits > 12) return 0;
    const char* env = knob("GOLAP_NARROW");
    if (env && !strcmp(env, "0")) return 0;
    JoinPlan plan;
    make_local_plan(p, plan);
    return local_narrow_ok(plan) && partition_narrow_ok(1u << p->pre_bits, false) ? 1 : 0;
}

int32_t golap_join_partitioned_begin(golap_ctx* ctx, const golap_join_partitioned_plan* p, void* stream) {
    if (!ctx || !p) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return part_begin(ctx, p, pick_stream(ctx, stream));
}

int32_t golap_join_partitioned_side(golap_ctx* ctx, int32_t which, const golap_partitioned_side* side, void* stream) {
    if (!ctx || !side) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return part_side(ctx, which, side, pick_stream(ctx, stream));
}

int32_t golap_join_partitioned_finish(golap_ctx* ctx, uint64_t out_pairs_ptr, uint64_t out_count_ptr, uint64_t max_pairs, void* stream) {
    if (!ctx) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null ctx");
    if (!out_count_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "out_count_ptr is null");
    if (max_pairs && !out_pairs_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "out_pairs_ptr is null but max_pairs > 0");
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return part_finish(ctx, out_pairs_ptr, out_count_ptr, max_pairs, pick_stream(ctx, stream));
}

// Join of radix-partitioned inputs in one call: begin + build side + probe side + finish.
int32_t golap_join_partitioned(golap_ctx* ctx, const golap_join_partitioned_params* p, void* stream) {
    if (!ctx || !p) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    if (!p->out_count_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "out_count_ptr is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick_stream(ctx, stream);
    golap_join_partitioned_plan pl;
    memset(&pl, 0, sizeof pl);
    pl.n_groups = p->n_slots;
    pl.n_sources = p->n_sources;
    pl.pre_bits = p->pre_bits ? p->pre_bits : 8;
    pl.total_build_rows_hint = p->total_build_rows_hint;
    pl.build_rows = p->build.n_rows;
    pl.probe_rows = p->probe.n_rows;
    GOLAP_TRY(part_begin(ctx, &pl, st));
    int32_t rc = part_side(ctx, 0, &p->build, st);
    if (rc == GOLAP_OK) rc = part_side(ctx, 1, &p->probe, st);
    if (rc == GOLAP_OK) rc = part_finish(ctx, p->out_pairs_ptr, p->out_count_ptr, p->max_pairs, st);
    if (rc != GOLAP_OK) part_abort(ctx);
    return rc;
}

int32_t golap_join_partitioned_abort(golap_ctx* ctx) {
    if (!ctx) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null ctx");
    std::lock_guard<std::mutex> lk(ctx->mu);
    part_abort(ctx);
    return GOLAP_OK;
}

}  // extern "C"

#include "golap_shard.inc.h"
#include "golap_shard_agg.inc.h"
