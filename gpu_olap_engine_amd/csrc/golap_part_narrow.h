// golap_part_narrow.h -- the two partition passes of the optimistic narrow-key pipeline, rewritten
// in round 2 around the SQ counter profile of their predecessors (profiles/r02_sq0: ~60 vector + 8-13 LDS
// instructions per row, 7 barrier-separated phases per tile, one 1024-thread work-group per CU so nothing overlaps):
//
//   k_part1_narrow  global first pass, int64 keys in -> (key32, row id) tuples out, bucket ranges reserved by per-tile
//                   atomics (ReserveArgs).  Same register-resident big tile as k_scatter_big, but
//                     * the LDS window row carries (key32, bucket | tile position) instead of (key32, row id): the
//                       write-out loop reads the bucket back instead of hashing the key a second time, and one
//                       `delta[bucket]` lookup replaces the s_gpos / s_start pair; the row id is rebuilt from the
//                       tile position (row ids are implicit in the first pass, join_kernels.cuh:75);
//                     * the next tile's keys are requested as soon as the last window has been staged (the key
//                       registers are dead from then on), so HBM reads overlap the write-out instead of starting
//                       after it; 16-byte loads when the column is 16-byte aligned;
//                     * s_cnt is re-zeroed by the thread that consumed it: one barrier less per tile.
//   k_part2_ring    local last pass over one group per work-group, tuples in -> tuples out into slot-provisioned
//                   sub-partitions (FixedArgs).  No tile sort and no carry copy: every bucket owns a RING of R rows in
//                   LDS; a row takes the next ring position of its bucket with one returning LDS atomic and is stored
//                   there directly; after each tile every bucket with a complete 16-row line (one 128-byte HBM line)
//                   writes it out, whole, by one 16-lane group.  Two barriers per tile instead of seven, ~4 LDS
//                   operations per row instead of ~13.  A row that finds its ring full (bucket received more than R
//                   minus its fill in one tile: 0.02 % of (tile, bucket) pairs at 8 rows per bucket and tile) keeps its
//                   reserved position and is placed after the flush (extra rounds until nothing is pending).
//
// ASM forms: vmcnt counts a wave's loads AND stores in issue order, and hipcc, which cannot count stores issued under
// data-dependent control flow, waits vmcnt(0) at the first use of a prefetched key -- i.e. for every store of the
// write-out just issued (p1) or even for the prefetch issued behind it (p2).  The ASM forms hide these loads and stores
// from hipcc (cdna_hip_programming.md 5.7 item 1, form (ii)) and issue every store at wave-uniform control flow
// (predicated through EXEC inside the statement), so the number of stores younger than a load is a compile-time
// constant and the wait is a counted vmcnt(N): N is never larger than the true number of younger operations, extra
// stores (retry rounds) only make the wait stricter.
// Both keep the device-side contracts of the kernels they replace: rs.wide / fx.overflow raise the same device words,
// the layouts (bucket ranges of `cap` rows; sub_start / sub_len per sub-partition) are identical, so the 64-bit
// pipeline standing by behind them and the LDS join are unchanged.
#pragma once
#include "golap_hidden.h"

namespace golap {

constexpr int P1N_THREADS = 1024;
constexpr int P1N_WAVES = P1N_THREADS / 64;

static inline size_t part1_narrow_lds_bytes(uint32_t F, uint32_t win, bool wide = false) { return (size_t)win * (wide ? 12 : 8) + (size_t)F * 12 + (P1N_WAVES + 2) * 4 + 16; }

// VEC2: 16-byte loads (needs a 16-byte aligned column); ASM: hidden loads/stores + counted waits (implies VEC2).
// DIAG != 0: timing-only diagnostic builds for scripts/ab_p12.py (WRONG results, never launched by the product path):
//   1 = write-out goes to the tile's own input positions (linear, no scatter), 2 = no stores at all, 3 = no loads (keys made up).
// DICT != 0: the form for key columns with 2 .. 4 distinct upper halves (PassCfg::key_hi): every key's upper half is looked up in the
// dictionary (scalar registers), the tuple's first word is (hash >> b1) | (index << (32 - b1)).  Both forms are enqueued; each
// returns at once unless the dictionary size on the device says it is the one to run.
// DICT == 2: more than 4 upper halves, all inside a window of F consecutive values (ids with a tenant / shard / epoch prefix).  The
// index is (upper half - key_hi[KEY_DICT_BASE_AT]) -- no dictionary, and halves the sample has not seen are fine as long as they fall
// into the window -- and the hash is fmix32(lower half ^ index * 0x9E3779B1).  With the reference's fold (lower ^ upper half) the rows
// (t1, l) and (t2, l ^ t1 ^ t2) share every hash bit, and a table of D tenants with dense local ids makes D-fold collisions the rule
// (bucket sizes with sqrt(D) times the spread): small upper halves flip LOW bits of the lower half, which lands on another valid id.
// A full-width constant per index flips high bits too: the partner l ^ c1 ^ c2 is a valid local id with probability 2^(k - 32) for
// k-bit local ids, i.e. a row expects (rows / 2^32) such partners -- the rate of a random hash.  Still a bijection of the lower half
// per index, which is all the tuple (hash >> b1 | index << (32 - b1)) needs.
// WIDE != 0 (round 3): the first pass of the 64-bit-row pipeline in the same shape -- whole 64-bit keys and row ids out as two arrays
// (out_rows = the key array, out_rids), 12 bytes per row through the LDS window, so a 12288-row tile; no key check and no abandon
// word (the launch is gated by rs.run_if instead), two hidden stores per row.  Replaces k_scatter_big<.., RESERVE>, which loaded a
// tile only after the previous one had been written out.
template <int RPT, int WIN, bool VEC2, bool ASM, int DIAG = 0, int DICT = 0, int WIDE = 0>
__global__ __launch_bounds__(P1N_THREADS) void k_part1_narrow(const long long* __restrict__ in_keys, uint32_t rid_base,
                                                              const uint32_t* __restrict__ chunk_off, const uint32_t* __restrict__ chunk_len,
                                                              uint2* __restrict__ out_rows, PassCfg cfg, ReserveArgs rs, uint32_t* __restrict__ out_rids) {
    constexpr int NT = P1N_THREADS, NW = P1N_WAVES;
    constexpr int TILE = NT * RPT;
    constexpr int ROUNDS = (TILE + WIN - 1) / WIN;
    constexpr int STORES = (WIDE ? 2 : 1) * (WIN / NT);   // store instructions per wave behind the next tile's loads: those of the last window (ASM: all issued at uniform control flow)
    static_assert(!WIDE || (DICT == 0 && DIAG == 0), "the 64-bit form has no dictionary and no diagnostic builds");
    static_assert(TILE <= 32768, "tile position is packed into 15 bits");
    static_assert(!VEC2 || RPT % 2 == 0, "16-byte loads take two keys");
    static_assert(!ASM || VEC2, "the hidden loads are the 16-byte ones");
    static_assert(WIN % NT == 0 && STORES <= 60, "window rows per thread; vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    uint2* s_rows = reinterpret_cast<uint2*>(s_raw);                       // [WIN] (key32, bucket << 15 | tile position); WIDE: the 64-bit key
    uint32_t* s_meta = reinterpret_cast<uint32_t*>(s_raw + (size_t)WIN * 8);  // WIDE: [WIN] bucket << 15 | tile position
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_raw + (size_t)WIN * (WIDE ? 12 : 8));   // [F] rows of the bucket in this tile
    uint32_t* s_start = s_cnt + cfg.F;                                     // [F] tile-local slot of the bucket's first row
    uint32_t* s_delta = s_start + cfg.F;                                   // [F] global position - tile-local slot
    uint32_t* s_w = s_delta + cfg.F;                                       // [NW] wave totals, [NW] abandon flag

    if (rs.run_if && *rs.run_if == 0) return;
    // the attempt may be over before it starts (k_key_hi's sample found two upper key halves): ~2000 work-groups that each load and
    // rank one tile before they notice cost 0.45 ms per side; asking first costs the good case one L2 round trip per work-group
    if (!WIDE && __hip_atomic_load(rs.wide, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    if (!WIDE && key_dict_mode(cfg) != DICT) return;   // (work-group-uniform: the other form of this pass does the work)
    const uint32_t chunk = blockIdx.x;
    const uint32_t len = chunk_len[chunk];
    if (len == 0) return;
    const uint32_t off = chunk_off[chunk];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t F = cfg.F;   // <= NT: one bucket per thread in the scan
    if (tid < F) s_cnt[tid] = 0;
    if (WIDE && tid == 0) s_w[NW] = 0;   // (no abandon word in the 64-bit form)
    const uint32_t khi = WIDE ? 0u : __builtin_amdgcn_readfirstlane(key_hi_of(cfg));   // the upper key half this attempt stands on (see PassCfg::key_hi)
    const bool emit_hash = cfg.pre_hashed != 0;
    // DICT: the dictionary's other entries and the position of the index in the tuple's first word
    const uint32_t kd1 = DICT == 1 ? __builtin_amdgcn_readfirstlane(cfg.key_hi[1]) : 0u, kd2 = DICT == 1 ? __builtin_amdgcn_readfirstlane(cfg.key_hi[2]) : 0u,
                   kd3 = DICT == 1 ? __builtin_amdgcn_readfirstlane(cfg.key_hi[3]) : 0u;
    const uint32_t b1 = 31u - (uint32_t)__builtin_clz(F);   // hash bits this pass consumes (F = 256 << e1: the low 8 + e1 bits, bucket_of)
    const uint32_t kbase = DICT == 2 ? __builtin_amdgcn_readfirstlane(cfg.key_hi[KEY_DICT_BASE_AT]) : 0u;   // first upper half of the window (offset form)

    // keys of the tile: kv[h] = (lo, hi) of key 2h, (lo, hi) of key 2h + 1
    u32x4 kv[(RPT + 1) / 2];
    // tile position of register r: consecutive lanes read consecutive 8-byte (16-byte) words
    auto tile_pos = [&](int r) -> uint32_t { return VEC2 ? 2u * ((uint32_t)(r >> 1) * NT + tid) + (uint32_t)(r & 1) : (uint32_t)r * NT + tid; };
    auto load_tile = [&](uint32_t t0) {
        const long long* __restrict__ src = in_keys + off + t0;
        const uint32_t tn = len - t0 < (uint32_t)TILE ? len - t0 : (uint32_t)TILE;
        if (DIAG == 3) {
#pragma unroll
            for (int h = 0; h < RPT / 2; ++h) kv[h] = u32x4{(off + t0 + tile_pos(2 * h)) * 2654435761u, 0u, (off + t0 + tile_pos(2 * h + 1)) * 2654435761u, 0u};
        } else if (VEC2 && tn == (uint32_t)TILE) {   // work-group-uniform: whole tile, unconditional 16-byte loads
#pragma unroll
            for (int h = 0; h < RPT / 2; ++h) {
                if (ASM) ld16_hidden(kv[h], src + tile_pos(2 * h));
                else kv[h] = *reinterpret_cast<const u32x4*>(src + tile_pos(2 * h));
            }
        } else if (ASM) {   // partial tile (the last one of a chunk): hidden 8-byte loads, waited for at once
            u32x2 t[RPT];
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const uint32_t i = tile_pos(r);
                ld8_hidden(t[r], src + (i < tn ? i : tn - 1));
            }
            wait_vmcnt<0>();
#pragma unroll
            for (int r = 0; r < RPT; ++r) pin(t[r]);
#pragma unroll
            for (int h = 0; h < RPT / 2; ++h) kv[h] = u32x4{t[2 * h].x, t[2 * h].y, t[2 * h + 1].x, t[2 * h + 1].y};
        } else {
#pragma unroll
            for (int r = 0; r < RPT; ++r) {   // clamped, unconditional (a conditional load costs a branch + vmcnt(0))
                const uint32_t i = tile_pos(r);
                const u32x2 k = *reinterpret_cast<const u32x2*>(src + (i < tn ? i : tn - 1));
                if (r & 1) { kv[r >> 1].z = k.x; kv[r >> 1].w = k.y; } else { kv[r >> 1].x = k.x; kv[r >> 1].y = k.y; }
            }
        }
    };
    uint32_t polled = 0;
    if (ASM && !WIDE && wv == 0) ld4_hidden_sc1(polled, rs.wide);
    load_tile(0);
    for (uint32_t t0 = 0; t0 < len; t0 += TILE) {
        const uint32_t tn = len - t0 < (uint32_t)TILE ? len - t0 : (uint32_t)TILE;
        const uint32_t rbase = off + t0;
        // has the narrow attempt been abandoned?  (one thread asks, everybody learns it behind the next barrier)
        if (!ASM && !WIDE && tid == 0) s_w[NW] = __hip_atomic_load(rs.wide, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t0 == 0) lds_barrier();   // s_cnt zeroed (later tiles: re-zeroed two barriers ago by the scan below)
        if (ASM) {
            // the keys (and, in wave 0, the abandon word in front of them) were requested before the STORES stores of the
            // previous write-out: everything older than those stores has landed once at most STORES operations are
            // outstanding (first tile: nothing is younger)
            if (t0 == 0) wait_vmcnt<0>(); else wait_vmcnt<STORES>();
#pragma unroll
            for (int h = 0; h < RPT / 2; ++h) pin(kv[h]);
            if (!WIDE && wv == 0) {
                asm volatile("" : "+v"(polled));
                if (tid == 0) s_w[NW] = polled;
            }
        }
        uint32_t k32[RPT], br[RPT];
        uint32_t hi = 0;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            k32[r] = (r & 1) ? kv[r >> 1].z : kv[r >> 1].x;
            const uint32_t khi_r = (r & 1) ? kv[r >> 1].w : kv[r >> 1].y;
            br[r] = 0xFFFFFFFFu;
            uint32_t h;
            if (WIDE) {
                h = hash_int32((int32_t)(k32[r] ^ khi_r));                                // == hash_int64 of the whole key (join_kernels.cuh:37-41)
            } else if (DICT == 2) {
                const uint32_t id = khi_r - kbase;
                hi = id > hi ? id : hi;                                                   // (the largest index of the tile: checked against the window below)
                h = hash_int32((int32_t)(k32[r] ^ (id * 0x9E3779B1u)));                   // (see the template's comment: a full-width constant per index)
                k32[r] = (h >> b1) | (id << (32u - b1));
            } else if (DICT) {
                const uint32_t m0 = khi_r ^ khi, m1 = khi_r ^ kd1, m2 = khi_r ^ kd2, m3 = khi_r ^ kd3;
                const uint32_t ma = m0 < m1 ? m0 : m1, mb = m2 < m3 ? m2 : m3;
                hi |= ma < mb ? ma : mb;                                                  // 0 iff the upper half is one of the dictionary's
                const uint32_t id = m3 == 0u ? 3u : (m2 == 0u ? 2u : (m1 == 0u ? 1u : 0u));
                h = hash_int32((int32_t)(k32[r] ^ khi_r));                                // == hash_int64 of the whole key
                k32[r] = (h >> b1) | (id << (32u - b1));                                  // (DICT implies pre_hashed)
            } else {
                hi |= khi_r ^ khi;   // clamped duplicates are real keys too
                h = hash_int32((int32_t)(k32[r] ^ khi));                                  // == hash_int64 of the whole key
                if (emit_hash) k32[r] = h;       // the tuple carries the hash from here on (PassCfg::pre_hashed); work-group-uniform
            }
            if (tile_pos(r) < tn) {
                const uint32_t b = bucket_of(h, cfg);
                br[r] = (b << 16) | atomicAdd(&s_cnt[b], 1u);                    // b < 1024, rank < 32768
            }
        }
        if (DICT == 2) hi >>= b1;   // an upper half outside the window of F values: the attempt is over
        if (!WIDE && hi) atomicOr(rs.wide, 1u);
        lds_barrier();
        if (s_w[NW]) return;   // work-group-uniform: some key somewhere does not fit 32 bits, the 64-bit pipeline takes over
        // exclusive scan of the bucket counts; the bucket's run in the output is reserved meanwhile (one returning
        // global atomic per (tile, bucket), consumed after the first window has been staged)
        uint32_t c = 0, rsv = 0, start;
        if (tid < F) {
            c = s_cnt[tid];
            s_cnt[tid] = 0;   // for the next tile
            if (c) {
                if (ASM) atomic_add_rtn_hidden(rsv, rs.cursor + reserve_slot(rs, tid), c);   // (a returning atomic hipcc tracks costs a vmcnt(0) at every reuse of its register)
                else rsv = atomicAdd(&rs.cursor[reserve_slot(rs, tid)], c);
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
            if (tid < F) s_start[tid] = start;
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < RPT; ++r)
            if (br[r] != 0xFFFFFFFFu) br[r] = (br[r] & 0xFFFF0000u) | (s_start[br[r] >> 16] + (br[r] & 0xFFFFu));   // bucket | tile-local slot
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const uint32_t w0 = (uint32_t)rd * WIN;
            if (w0 < tn) {
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    const uint32_t pos = (br[r] & 0xFFFFu) - w0;   // rows of other windows fall outside
                    if (br[r] != 0xFFFFFFFFu && pos < (uint32_t)WIN) {
                        if (WIDE) {
                            s_rows[pos] = make_uint2(k32[r], (r & 1) ? kv[r >> 1].w : kv[r >> 1].y);
                            s_meta[pos] = ((br[r] >> 16) << 15) | tile_pos(r);
                        } else s_rows[pos] = make_uint2(k32[r], ((br[r] >> 16) << 15) | tile_pos(r));
                    }
                }
            }
            if (ASM && rd == 0) {   // nothing hidden is younger than the reservation here (the next tile's loads follow below)
                wait_vmcnt<0>();
                asm volatile("" : "+v"(rsv));
            }
            if (rd == 0 && tid < F) {   // the reservations have had the scan and the staging above to come back
                s_delta[tid] = reserve_slot(rs, tid) * rs.cap + rsv - start;
                if (rsv + c > rs.cap) atomicOr(rs.overflow, 1u);   // rsv + c cannot wrap: both < 2^31
            }
            if (rd == ROUNDS - 1 && t0 + TILE < len) {   // key registers are dead: the next tile streams in during the write-out
                if (ASM && !WIDE && wv == 0) ld4_hidden_sc1(polled, rs.wide);   // (one tile stale when it is read: abandoning is an optimisation)
                load_tile(t0 + TILE);
            }
            lds_barrier();
            const uint32_t wn = w0 < tn ? (tn - w0 < (uint32_t)WIN ? tn - w0 : (uint32_t)WIN) : 0u;
            if (ASM) {
#pragma unroll 2
                for (int j = 0; j < WIN / NT; ++j) {   // every store instruction is issued, whatever the data (see the file header)
                    const uint32_t i = (uint32_t)j * NT + tid;
                    const uint2 row = s_rows[i];
                    const uint32_t meta = WIDE ? s_meta[i] : row.y;
                    const uint32_t b = (meta >> 15) & (F - 1u);   // (masked: rows beyond wn hold stale words)
                    const uint32_t gp = s_delta[b] + w0 + i;
                    const uint32_t ok = (i < wn && gp < (reserve_slot(rs, b) + 1u) * rs.cap) ? 1u : 0u;
                    st8_hidden_if(out_rows + gp, u32x2{row.x, WIDE ? row.y : rid_base + rbase + (meta & 0x7FFFu)}, ok);
                    if (WIDE) st4_hidden_if(out_rids + gp, rid_base + rbase + (meta & 0x7FFFu), ok);
                }
            } else if (wn) {
#pragma unroll 4
                for (int j = 0; j < WIN / NT; ++j) {
                    const uint32_t i = (uint32_t)j * NT + tid;
                    if (i < wn) {
                        const uint2 row = s_rows[i];
                        const uint32_t meta = WIDE ? s_meta[i] : row.y;
                        const uint32_t b = meta >> 15;
                        const uint32_t gp = DIAG == 1 ? rbase + w0 + i : s_delta[b] + w0 + i;
                        if (DIAG != 2 && (DIAG == 1 || gp < (reserve_slot(rs, b) + 1u) * rs.cap)) {   // rows past a bucket's range are dropped (overflow raised above)
                            out_rows[gp] = make_uint2(row.x, WIDE ? row.y : rid_base + rbase + (meta & 0x7FFFu));   // join_kernels.cuh:75
                            if (WIDE) out_rids[gp] = rid_base + rbase + (meta & 0x7FFFu);
                        }
                    }
                }
            }
            if (rd + 1 < ROUNDS) lds_barrier();   // the window is restaged; (after the last round the next LDS writes that could
                                                  // collide -- s_start, s_delta, s_rows -- all sit behind later barriers)
        }
    }
}

// ---------------------------------------------------------------------------
// last pass over (key32, row id) tuples: per-bucket rings in LDS (see the file header)
// ---------------------------------------------------------------------------
constexpr int P2R_THREADS = 1024;
constexpr int P2R_LINE = 16;     // rows per 128-byte line
constexpr int P2R_MAX_F = 512;

static inline size_t part2_ring_lds_bytes(uint32_t F, uint32_t R) { return (size_t)F * R * 8 + (size_t)F * 16 + 64 + 3 * 128 * 4; }   // (3 piece arrays of P2R_MAX_SRC words)
// ring length for a fan-out: F * R = 16384 rows of LDS (128 KiB) whatever the fan-out -- a tile brings TILE / F rows per bucket,
// and a ring must hold that on top of an unfinished line (R = 32 at F = 512 ... 256 at F <= 64)
static inline uint32_t part2_ring_rows(uint32_t F) { return F >= 512 ? 32u : F >= 256 ? 64u : F >= 128 ? 128u : 256u; }

// Rows of a group: one extent of the previous pass's output (fx.groups), or -- after a partition exchange -- one segment
// per source rank (SegSrc: row offsets relative to in_rows, [group][source]); the rings do not care where rows come from.
struct SegSrc {
    const uint32_t* off;
    const uint32_t* len;
    uint32_t n_src;     // 0: use fx.groups
};
constexpr int P2R_MAX_SRC = 128;   // pieces of a group: per-XCD sub-ranges, or (source rank x row chunk x sub-range) after a partition exchange

template <int RPT, int R, bool ASM, int DIAG = 0>   // DIAG 2: timing-only build without the flush stores (wrong results; scripts/ab_p12.py)
__global__ __launch_bounds__(P2R_THREADS) __attribute__((amdgpu_waves_per_eu(5))) void k_part2_ring(const uint2* __restrict__ in_rows, uint2* __restrict__ out_rows, PassCfg cfg, FixedArgs fx,
                                                            SegSrc seg) {
    constexpr int NT = P2R_THREADS, TILE = NT * RPT, LINE = P2R_LINE;
    constexpr int HL = LINE / 2;                           // lanes per line in the regular flush: 16 bytes (two rows) each
    constexpr int FLUSH_ITERS = P2R_MAX_F / (NT / HL);     // line items per 8-lane group (ASM: always this many store instructions per flush)
    constexpr int LPB = R / (2 * LINE);                    // lines a bucket gets per regular flush (half a ring); F * LPB <= 512 line items
    constexpr int FB = LPB == 1 ? FLUSH_ITERS : 2;         // line items whose LDS reads are in flight together (register budget: 96)
    static_assert((R & (R - 1)) == 0 && R >= 2 * LINE && R <= 256, "ring geometry");
    static_assert(!ASM || 2 * RPT <= P2R_FIXED_PAIRS, "two tiles of hidden loads in the fixed registers");
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const uint32_t F = cfg.F;
    uint2* s_ring = reinterpret_cast<uint2*>(s_raw);                          // [F][R]
    // [F] per bucket ONE 8-byte word: low = rows that have taken a position so far (wr), high = rows written out so far (rd, a
    // multiple of LINE).  A row takes its position with one 64-bit returning LDS atomic that also tells it how far the ring
    // has been drained.
    unsigned long long* s_wrrd = reinterpret_cast<unsigned long long*>(s_raw + (size_t)F * R * 8);
    uint32_t* s_snap = reinterpret_cast<uint32_t*>(s_wrrd + F);               // [2][F], rings longer than two lines only: the rd a flush works
                                                                              // from (several 16-lane groups serve one bucket; one copy is read,
                                                                              // the other written)
    uint32_t* s_flag = s_snap + 2 * F;                                        // [2] "rows are pending" (alternating)
    uint32_t* s_poff = s_flag + 2;                                            // [n_pieces] first row of the piece
    uint32_t* s_plen = s_poff + P2R_MAX_SRC;                                  // [n_pieces] rows of the piece
    uint32_t* s_pend = s_plen + P2R_MAX_SRC;                                  // [n_pieces] tiles up to and including the piece

    if (fx.skip_if && *fx.skip_if != 0) return;
    const uint32_t g = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t n_pieces = seg.n_src ? seg.n_src : group_pieces(fx.groups);
    if (tid == 0) {   // the group's pieces and their tile counts (a handful of entries: one thread)
        // The pieces are visited in an order ROTATED by the group number: all work-groups of a launch advance at the same pace, and with
        // every one of them in its piece 0 (then 1, ...) the addresses they stream from differ by whole multiples of one stride -- at
        // some strides that lands the 256 streams on few HBM channels (100M rows: the pass ran 0.745 or 0.785 ms depending on 32 rows
        // more or less in the sub-range capacity).  Rows of a group may be taken in any order.
        uint32_t tiles = 0;
        const uint32_t rot = cfg.piece_rot_off ? 0u : g % n_pieces;
        for (uint32_t s = 0; s < n_pieces; ++s) {
            uint32_t o, l;
            const uint32_t ps = s + rot < n_pieces ? s + rot : s + rot - n_pieces;
            if (seg.n_src) { o = seg.off[(size_t)g * seg.n_src + ps]; l = seg.len[(size_t)g * seg.n_src + ps]; }
            else group_piece(fx.groups, g, ps, o, l);
            tiles += (l + TILE - 1) / TILE;
            s_poff[s] = o; s_plen[s] = l; s_pend[s] = tiles;
        }
    }
    if (tid < F) { s_wrrd[tid] = 0ull; s_snap[tid] = 0; s_snap[F + tid] = 0; }
    if (tid < 2) s_flag[tid] = 0;
    lds_barrier();
    const uint32_t n_tiles = s_pend[n_pieces - 1];
    // (readfirstlane: work-group-uniform values in scalar registers, and -- because it consumes the loads here -- hipcc's wait for
    // them sits in front of the tile loop; left to their first use inside the loop it becomes a vmcnt(0) per tile, which would
    // also drain the hidden prefetch)
    const uint32_t slot0 = __builtin_amdgcn_readfirstlane(fx.slot_base[g]), cap = __builtin_amdgcn_readfirstlane(fx.slot_cap[g]);
    const uint32_t khi = __builtin_amdgcn_readfirstlane(key_hi_of(cfg));
    const bool hashed = cfg.pre_hashed != 0;   // the tuples carry hash_int32(key) instead of the key (k_part1_narrow wrote them so)
    // dictionary mode (PassCfg::key_hi): the first word is (hash >> b1) | (index << (32 - b1)) with b1 = cfg.shift, the bits the first
    // pass consumed -- this pass's digit sits at bit 0
    if (hashed && key_dict_mode(cfg)) cfg.shift = 0;
    uint2* const out_group = out_rows + slot0;   // work-group-uniform base of every store of this group (byte offsets fit 32 bits)
    if (n_tiles == 0) {
        if (tid < F) { fx.sub_start[(size_t)g * F + tid] = slot0 + tid * cap; fx.sub_len[(size_t)g * F + tid] = 0; }
        return;
    }
    // tile k of the group -> first row (relative to in_rows) and number of rows; k < n_tiles.  Tiles are visited in order, so
    // each of the two streams (loads run two tiles ahead of the processing) keeps a cursor over the pieces: the piece table
    // is only read when a stream steps into the next piece (a scan per call cost ~20 % of a tile with 8 pieces).
    struct PieceCursor { uint32_t s, before, end, off, len; };
    auto cursor_init = [&]() -> PieceCursor { return PieceCursor{0u, 0u, s_pend[0], s_poff[0], s_plen[0]}; };
    auto locate = [&](PieceCursor& c, uint32_t k, uint32_t& row0, uint32_t& tn) {
        while (k >= c.end) {   // work-group-uniform; pieces without rows are skipped (their end == before)
            c.before = c.end;
            ++c.s;
            c.end = s_pend[c.s];
            c.off = s_poff[c.s];
            c.len = s_plen[c.s];
        }
        const uint32_t t0 = (k - c.before) * (uint32_t)TILE;
        row0 = c.off + t0;
        tn = c.len - t0 < (uint32_t)TILE ? c.len - t0 : (uint32_t)TILE;
    };
    PieceCursor cur_load = cursor_init(), cur_proc = cursor_init();
    // Row r * NT + tid of a tile travels in 8-byte load r of thread tid.  Always RPT load instructions, indices clamped into the
    // tile: short tiles (the last one of a piece) need no other code, and for a tile beyond the group's last one every lane
    // re-reads one row (a single line per instruction) -- the number of memory operations between a tile's loads and its
    // counted wait never depends on the data.
    auto load_tile = [&](auto set, u32x2 (&t)[RPT], uint32_t k) {
        uint32_t row0, tn;
        locate(cur_load, k < n_tiles ? k : n_tiles - 1, row0, tn);
        if (k >= n_tiles) tn = 1;
        const uint2* __restrict__ src = in_rows + __builtin_amdgcn_readfirstlane(row0);   // scalar base, 32-bit lane offsets
        if constexpr (ASM) ld8_fixed_tile<decltype(set)::value, RPT>(src, tid, __builtin_amdgcn_readfirstlane(tn), (uint32_t)NT);
        else {
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const uint32_t i = (uint32_t)r * NT + tid;
                const uint2 a = src[i < tn ? i : tn - 1];
                t[r] = u32x2{a.x, a.y};
            }
        }
    };
    // bank rotation of a bucket's ring: neighbouring buckets (flushed by one 32-lane group) start on opposite halves
    auto ring_at = [&](uint32_t b, uint32_t p) -> uint32_t { return b * R + ((p + ((b & 1u) << 4)) & (uint32_t)(R - 1)); };
    uint32_t it = 0;
    uint32_t cur = 0;                                    // LPB > 1: which copy of s_snap is current (flips behind every regular flush)
    const uint32_t lgF = 31u - (uint32_t)__builtin_clz(F);
    // complete lines of every bucket leave the rings; `fin`: also the unfinished last line (no cursor update: nothing follows)
    auto flush = [&](bool fin) {
        const uint32_t q = tid >> 4, l = tid & 15;
        const uint32_t* snap = s_snap + cur * F;
        uint32_t* snap_next = s_snap + (cur ^ 1u) * F;
        if (ASM && !fin) {
            // FLUSH_ITERS line items per 8-lane group (a lane moves two rows = 16 bytes: half as many items per lane as with one
            // row each, and the per-item cursor arithmetic is what this kernel spends its vector instructions on); all cursor
            // reads are issued together, then all ring reads, then the stores -- two LDS round trips per flush.
            // Item q8 + 128 * j = line (item >> lgF) of bucket (item & (F - 1)).
            const uint32_t q8 = tid >> 3, l2 = (tid & 7u) * 2u;
#pragma unroll 1
            for (int j0 = 0; j0 < FLUSH_ITERS; j0 += FB) {
            uint32_t wr[FB], rd[FB], nl[FB];
            u32x4 row[FB];
#pragma unroll
            for (int j = 0; j < FB; ++j) {
                const uint32_t b = (q8 + (uint32_t)(j0 + j) * (NT / HL)) & (F - 1u);
                const unsigned long long w = s_wrrd[b];
                wr[j] = (uint32_t)w;
                rd[j] = LPB == 1 ? (uint32_t)(w >> 32) : snap[b];
            }
#pragma unroll
            for (int j = 0; j < FB; ++j) {
                const uint32_t b = (q8 + (uint32_t)(j0 + j) * (NT / HL)) & (F - 1u);
                const uint32_t li = LPB == 1 ? 0u : (q8 + (uint32_t)(j0 + j) * (NT / HL)) >> lgF;
                uint32_t end = wr[j] < cap ? wr[j] : cap;
                if (end > rd[j] + R) end = rd[j] + R;
                nl[j] = li < (uint32_t)LPB ? (end - rd[j]) / (uint32_t)LINE : 0u;   // (items beyond F * LPB: no lines)
                row[j] = *reinterpret_cast<const u32x4*>(&s_ring[ring_at(b, rd[j] + li * (uint32_t)LINE + l2)]);
            }
            bool extra = false;
#pragma unroll
            for (int j = 0; j < FB; ++j) {   // one store instruction per item at uniform control flow
                const uint32_t b = (q8 + (uint32_t)(j0 + j) * (NT / HL)) & (F - 1u);
                const uint32_t li = LPB == 1 ? 0u : (q8 + (uint32_t)(j0 + j) * (NT / HL)) >> lgF;
                st16_hidden_mask_s(out_group, (b * cap + rd[j] + li * (uint32_t)LINE + l2) * 8u, row[j], __ballot(li < nl[j]));
                extra |= li + (uint32_t)LPB < nl[j];
                if (l2 == 0 && li == 0) {   // (every bucket has its item li = 0)
                    reinterpret_cast<uint32_t*>(s_wrrd + b)[1] = rd[j] + nl[j] * (uint32_t)LINE;
                    if (LPB > 1) snap_next[b] = rd[j] + nl[j] * (uint32_t)LINE;
                }
            }
            if (__ballot(extra) != 0ull) {   // rare: more than half a ring of complete lines (extra stores only make the counted wait stricter)
#pragma unroll
                for (int j = 0; j < FB; ++j) {
                    const uint32_t b = (q8 + (uint32_t)(j0 + j) * (NT / HL)) & (F - 1u);
                    const uint32_t li = LPB == 1 ? 0u : (q8 + (uint32_t)(j0 + j) * (NT / HL)) >> lgF;
                    const uint32_t p1 = rd[j] + (li + (uint32_t)LPB) * (uint32_t)LINE + l2;
                    const u32x4 row2 = *reinterpret_cast<const u32x4*>(&s_ring[ring_at(b, p1)]);
                    st16_hidden_mask_s(out_group, (b * cap + p1) * 8u, row2, __ballot(li + (uint32_t)LPB < nl[j]));
                }
            }
            }
            return;
        }
        for (uint32_t b = q; b < F; b += NT / LINE) {   // one 16-lane group per bucket: all its lines
            const unsigned long long w = s_wrrd[b];
            const uint32_t wr = (uint32_t)w, rd = (uint32_t)(w >> 32);
            uint32_t end = wr < cap ? wr : cap;                 // rows past the slot were dropped
            if (end > rd + R) end = rd + R;                     // positions beyond the ring are still pending
            if (!fin) end = rd + ((end - rd) & ~(uint32_t)(LINE - 1));
            uint2* __restrict__ dst = out_group + b * cap;
            if (DIAG != 2)
                for (uint32_t p = rd + l; p < end; p += LINE) dst[p] = s_ring[ring_at(b, p)];   // 16 lanes x 8 B = one 128-byte line
            if (l == 0 && !fin) {
                reinterpret_cast<uint32_t*>(s_wrrd + b)[1] = end;
                if (LPB > 1) snap_next[b] = end;
            }
        }
    };
    uint32_t tile_no = 0;
    auto process = [&](auto set, u32x2 (&t)[RPT], uint32_t k) {   // k may be n_tiles (odd tile counts): a tile without rows
        uint32_t row0_unused, tn;
        locate(cur_proc, k < n_tiles ? k : n_tiles - 1, row0_unused, tn);
        if (k >= n_tiles) tn = 0;
        if constexpr (ASM) {
            // this tile's rows were requested two tiles ago, before two flushes of FLUSH_ITERS stores each (the RPT loads of the
            // tile in between only add to the younger operations); the first two tiles follow the prologue's full wait
            if (tile_no >= 2) wait_vmcnt<2 * FLUSH_ITERS>(); else wait_vmcnt<0>();
            ld8_take_tile<decltype(set)::value, RPT>(t);
        }
        ++tile_no;
        uint32_t key[RPT], rid[RPT], bk[RPT], pos[RPT];
        uint32_t pend = 0;
        // one returning 64-bit LDS atomic per row: its position in the bucket (low word) and how far the ring is drained (high)
        auto take = [&](int r, uint32_t& rdv) {
            const uint32_t h = hashed ? key[r] : hash_int32((int32_t)(key[r] ^ khi));
            bk[r] = bucket_of(h, cfg);
            const unsigned long long w = atomicAdd(&s_wrrd[bk[r]], 1ull);
            pos[r] = (uint32_t)w;
            rdv = (uint32_t)(w >> 32);
        };
        uint32_t rdv[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) { key[r] = t[r].x; rid[r] = t[r].y; }
        if (tn == (uint32_t)TILE) {   // work-group-uniform: the common tile needs no per-row bounds
#pragma unroll
            for (int r = 0; r < RPT; ++r) take(r, rdv[r]);
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                if (pos[r] >= cap) continue;                                  // slot overflow: dropped, flag raised at the end
                if (pos[r] - rdv[r] >= (uint32_t)R) pend |= 1u << r;          // ring full: placed after the flush
                else s_ring[ring_at(bk[r], pos[r])] = make_uint2(key[r], rid[r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                bk[r] = 0xFFFFFFFFu;
                if ((uint32_t)r * NT + tid < tn) take(r, rdv[r]);
            }
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                if (bk[r] == 0xFFFFFFFFu || pos[r] >= cap) continue;
                if (pos[r] - rdv[r] >= (uint32_t)R) pend |= 1u << r;
                else s_ring[ring_at(bk[r], pos[r])] = make_uint2(key[r], rid[r]);
            }
        }
        if (__ballot(pend != 0) != 0ull && lane == 0) s_flag[it & 1] = 1;
        lds_barrier();                       // ring rows + pending flag visible
        load_tile(set, t, k + 2);            // this tile's registers are free: the tile after next streams in behind two flushes
        for (;;) {
            flush(false);
            const bool more = s_flag[it & 1] != 0;
            if (tid == 0) s_flag[(it + 1) & 1] = 0;
            ++it;
            cur ^= 1u;
            lds_barrier();                       // rings drained (cursors advanced), the other flag cleared
            if (!more) break;                    // work-group-uniform
#pragma unroll
            for (int r = 0; r < RPT; ++r)
                if ((pend >> r) & 1u) {
                    const uint32_t rd_now = reinterpret_cast<const uint32_t*>(s_wrrd + bk[r])[1];
                    if (pos[r] - rd_now < (uint32_t)R) {
                        s_ring[ring_at(bk[r], pos[r])] = make_uint2(key[r], rid[r]);
                        pend &= ~(1u << r);
                    }
                }
            if (__ballot(pend != 0) != 0ull && lane == 0) s_flag[it & 1] = 1;
            lds_barrier();                       // ring rows + pending flag visible
        }
    };
    u32x2 ta[RPT], tb[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) { ta[r] = u32x2{0u, 0u}; tb[r] = u32x2{0u, 0u}; }
    constexpr std::integral_constant<int, 0> set_a{};
    constexpr std::integral_constant<int, 1> set_b{};
    load_tile(set_a, ta, 0);
    load_tile(set_b, tb, 1);
    for (uint32_t k = 0; k < n_tiles; k += 2) {
        process(set_a, ta, k);
        process(set_b, tb, k + 1);
    }
    if (ASM) wait_vmcnt<0>();
    flush(true);   // (the barrier at the end of the last process() ordered every ring write before this)
    if (tid < F) {
        const uint32_t cnt = (uint32_t)s_wrrd[tid];
        fx.sub_start[(size_t)g * F + tid] = slot0 + tid * cap;
        fx.sub_len[(size_t)g * F + tid] = cnt < cap ? cnt : cap;
        if (cnt > cap) atomicOr(fx.overflow, 1u);
    }
}

}  // namespace golap
