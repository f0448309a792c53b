// golap_part_wide.h -- the LAST partition pass of the 64-bit-row pipeline in the shape of its first pass (round 4).
//
// k_scatter_wc<2, FIXED> (round 1) keeps every bucket's unfinished 128-byte line in LDS and so writes each line once, whole: its
// traffic is the algorithmic 24 bytes per row, but a tile costs seven barriers, a carry of up to 15 rows x 512 buckets moves through
// registers on every tile, loads wait for the carry to be known, and the pass runs at 3.8 TB/s (48 % of peak: the kernel furthest
// from its roofline on the join path).  This kernel takes k_part1_narrow's shape instead: a 12288-row tile is ranked
// with LDS atomics, laid out bucket by bucket in ONE LDS window (8-byte key plane + 4-byte row-id plane), and written out as runs
// of ~24 rows per bucket (192-byte key runs, 96-byte row-id runs) while the next tile's rows stream into registers; three barriers
// per tile.
// Unaligned runs only work when the neighbouring run -- which shares the 128-byte line at the common boundary -- arrives while that
// line is still in the L2 (otherwise each boundary costs two partial writes, i.e. read-modify-writes at the memory side: the first
// form of this kernel, ONE work-group per group with its cursors in LDS, wrote a bucket's consecutive runs a whole tile period apart
// and ran at 7.9 ms per 1B rows against the 6.3 ms of the kernel it was meant to replace).  So, as in the first pass, SEVERAL
// work-groups of one XCD feed the same cursors at the same time: a group is split into C chunks (its per-XCD pieces, or parts of
// them), work-group (g, c) reserves its bucket runs with one returning global atomic per (tile, bucket) on the group's cursors --
// fx.sub_len itself, zeroed by k_slots_setup -- and blockIdx is laid out so that the C work-groups of a group are dispatched
// back to back on ONE XCD (blockIdx % 8 = g % 8: observed round-robin dispatch; speed only).
// The device-side contract is k_scatter_wc<2, true>'s: slots of `cap` rows from k_slots_setup, rows past a slot's end dropped and
// fx.overflow raised (the histogram path standing by redoes the pass), run_if / skip_if gates, sub_start / sub_len per sub-partition
// (sub_len = the cursors themselves: below `cap` whenever fx.overflow is still 0, which is the only time the layout is read).
// ASM: hidden loads, stores and atomics with counted waits (golap_hidden.h; audited by tests/test_isa_audit.py).
#pragma once
#include "golap_hidden.h"

namespace golap {

constexpr int P2W_THREADS = 1024;
constexpr int P2W_WAVES = P2W_THREADS / 64;

static inline size_t part2_wide_lds_bytes(uint32_t F, uint32_t tile) { return (size_t)tile * 12 + (size_t)F * 12 + (P2W_WAVES + 4) * 4 + 16; }

// work-group -> (group, chunk): the C chunks of a group on consecutive blocks of ONE XCD (blockIdx % 8) when the group count allows it
__device__ __forceinline__ void p2w_group_chunk(uint32_t b, uint32_t G, uint32_t C, uint32_t& g, uint32_t& c) {
    if ((G & 7u) == 0u) {
        const uint32_t x = b & 7u, q = b >> 3;
        c = q % C;
        g = (q / C) * 8u + x;
    } else {
        g = b / C;
        c = b % C;
    }
}

template <int RPT, bool ASM>
__global__ __launch_bounds__(P2W_THREADS) void k_part2_wide(const long long* __restrict__ in_keys, const uint32_t* __restrict__ in_rids,
                                                            long long* __restrict__ out_keys, uint32_t* __restrict__ out_rids, PassCfg cfg, FixedArgs fx,
                                                            uint32_t G, uint32_t C, const uint32_t* __restrict__ run_if) {
    constexpr int NT = P2W_THREADS, NW = P2W_WAVES;
    constexpr int TILE = NT * RPT;
    constexpr int STORES = 2 * RPT;   // store instructions per wave behind the next tile's loads (all issued at uniform control flow)
    static_assert(RPT % 2 == 0 && STORES <= 60 && TILE <= 65536, "16-byte loads take two keys; vmcnt is a 6-bit counter; ranks are packed into 16 bits");
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    uint2* s_keys = reinterpret_cast<uint2*>(s_raw);                               // [TILE] keys, bucket by bucket
    uint32_t* s_rids = reinterpret_cast<uint32_t*>(s_raw + (size_t)TILE * 8);       // [TILE] their row ids
    uint32_t* s_cnt = s_rids + TILE;                                               // [F] rows of the bucket in this tile
    uint32_t* s_start = s_cnt + cfg.F;                                             // [F] tile-local position of the bucket's first row
    uint32_t* s_delta = s_start + cfg.F;                                           // [F] global position - tile-local position
    uint32_t* s_w = s_delta + cfg.F;                                               // [NW] wave totals

    if (run_if && *run_if == 0) return;
    if (fx.skip_if && *fx.skip_if != 0) return;
    uint32_t g, c;
    p2w_group_chunk(blockIdx.x, G, C, g, c);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t F = cfg.F;   // <= NT: one bucket per thread in the scan
    const uint32_t slot0 = fx.slot_base[g], cap = fx.slot_cap[g];
    // rows reserved so far per bucket of this group (zeroed by k_slots_setup).  The cursors ARE the pass's sub_len output: a cursor beyond
    // its slot raises fx.overflow, and the layout is only read while that word is 0 -- so no clamping pass is needed behind this kernel
    uint32_t* __restrict__ cursor = fx.sub_len + (size_t)g * F;
    if (c == 0 && tid < F) fx.sub_start[(size_t)g * F + tid] = slot0 + tid * cap;   // (chunk 0 exists for every group, rows or not)
    // this work-group's rows: chunk c of the group = a part of one of its pieces (per-XCD sub-ranges of the first pass), or of its one extent
    uint32_t off, len;
    {
        const uint32_t np = group_pieces(fx.groups);
        const uint32_t parts = (np > 1 && C % np == 0) ? C / np : (np > 1 ? 0u : C);   // parts per piece (0: C does not fit the pieces, chunk 0 .. np - 1 take whole pieces)
        uint32_t po, pl, k = 0;
        if (np > 1 && parts == 0) { po = 0; pl = 0; if (c < np) group_piece(fx.groups, g, c, po, pl); }
        else if (np > 1) { group_piece(fx.groups, g, c / parts, po, pl); k = c % parts; }
        else { group_extent(fx.groups, g, po, pl); k = c; }
        const uint32_t pp = parts ? parts : 1u;
        const uint32_t per = ((pl + pp - 1) / pp + (uint32_t)TILE - 1u) / (uint32_t)TILE * (uint32_t)TILE;   // whole tiles per part: only a piece's last part ends in a partial tile
        const uint32_t lo = k * per < pl ? k * per : pl, hi = (k + 1u) * per < pl ? (k + 1u) * per : pl;
        off = po + lo;
        len = hi - lo;
    }
    if (len == 0) return;
    if (tid < F) s_cnt[tid] = 0;

    // rows of the tile: kv[h] = (lo, hi) of key 2h, (lo, hi) of key 2h + 1; rv[h] = their row ids
    u32x4 kv[RPT / 2];
    u32x2 rv[RPT / 2];
    auto tile_pos = [&](int r) -> uint32_t { return 2u * ((uint32_t)(r >> 1) * NT + tid) + (uint32_t)(r & 1); };
    auto load_tile = [&](uint32_t t0) {
        const uint32_t src = off + t0;
        const uint32_t tn = len - t0 < (uint32_t)TILE ? len - t0 : (uint32_t)TILE;
        const long long* __restrict__ kp = in_keys + src;
        const uint32_t* __restrict__ rp = in_rids + src;
        if (tn == (uint32_t)TILE && (src & 1u) == 0u) {   // work-group-uniform: whole tile on a 16-byte boundary, unconditional wide loads
#pragma unroll
            for (int h = 0; h < RPT / 2; ++h) {
                if (ASM) { ld16_hidden(kv[h], kp + tile_pos(2 * h)); ld8_hidden(rv[h], rp + tile_pos(2 * h)); }
                else { kv[h] = *reinterpret_cast<const u32x4*>(kp + tile_pos(2 * h)); rv[h] = *reinterpret_cast<const u32x2*>(rp + tile_pos(2 * h)); }
            }
        } else if (ASM) {   // the last tile of a chunk (or an odd start): hidden 8- and 4-byte loads on clamped indices, waited for at once
            u32x2 t[RPT];
            uint32_t q[RPT];
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const uint32_t i = tile_pos(r), ci = i < tn ? i : tn - 1;
                ld8_hidden(t[r], kp + ci);
                ld4_hidden(q[r], rp + ci);
            }
            wait_vmcnt<0>();
#pragma unroll
            for (int r = 0; r < RPT; ++r) { pin(t[r]); pin(q[r]); }
#pragma unroll
            for (int h = 0; h < RPT / 2; ++h) {
                kv[h] = u32x4{t[2 * h].x, t[2 * h].y, t[2 * h + 1].x, t[2 * h + 1].y};
                rv[h] = u32x2{q[2 * h], q[2 * h + 1]};
            }
        } else {
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const uint32_t i = tile_pos(r), ci = i < tn ? i : tn - 1;
                const u32x2 k = *reinterpret_cast<const u32x2*>(kp + ci);
                const uint32_t q = rp[ci];
                if (r & 1) { kv[r >> 1].z = k.x; kv[r >> 1].w = k.y; rv[r >> 1].y = q; } else { kv[r >> 1].x = k.x; kv[r >> 1].y = k.y; rv[r >> 1].x = q; }
            }
        }
    };
    load_tile(0);
    lds_barrier();   // s_cnt zeroed (later tiles: re-zeroed two barriers ago by the scan below)
    if (ASM) {
        wait_vmcnt<0>();
#pragma unroll
        for (int h = 0; h < RPT / 2; ++h) { pin(kv[h]); pin(rv[h]); }
    }
    for (uint32_t t0 = 0;; t0 += TILE) {
        const uint32_t tn = len - t0 < (uint32_t)TILE ? len - t0 : (uint32_t)TILE;
        const bool has_next = t0 + TILE < len;
        uint32_t br[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const uint32_t lo = (r & 1) ? kv[r >> 1].z : kv[r >> 1].x, hi = (r & 1) ? kv[r >> 1].w : kv[r >> 1].y;
            br[r] = 0xFFFFFFFFu;
            if (tile_pos(r) < tn) {
                const uint32_t b = bucket_of(hash_int32((int32_t)(lo ^ hi)), cfg);   // == hash_int64 of the whole key (join_kernels.cuh:37-41)
                br[r] = (b << 16) | atomicAdd(&s_cnt[b], 1u);                        // b < 1024, rank < 65536
            }
        }
        lds_barrier();
        // exclusive scan of the bucket counts; the bucket's run in its slot is reserved meanwhile (one returning global atomic per
        // (tile, bucket), consumed after the tile has been staged)
        uint32_t cnt = 0, rsv = 0, start;
        if (tid < F) {
            cnt = s_cnt[tid];
            s_cnt[tid] = 0;   // for the next tile
            if (cnt) {
                if (ASM) atomic_add_rtn_hidden(rsv, cursor + tid, cnt);   // (a returning atomic hipcc tracks costs a vmcnt(0) at every reuse of its register)
                else rsv = atomicAdd(&cursor[tid], cnt);
            }
        }
        {
            const uint32_t inc = wave_inclusive_scan(cnt);
            if (lane == 63) s_w[wv] = inc;
            lds_barrier();
            uint32_t base = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) base += (uint32_t)i < wv ? s_w[i] : 0u;
            start = base + inc - cnt;
            if (tid < F) s_start[tid] = start;
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            if (br[r] != 0xFFFFFFFFu) {
                const uint32_t pos = s_start[br[r] >> 16] + (br[r] & 0xFFFFu);
                s_keys[pos] = make_uint2((r & 1) ? kv[r >> 1].z : kv[r >> 1].x, (r & 1) ? kv[r >> 1].w : kv[r >> 1].y);
                s_rids[pos] = (r & 1) ? rv[r >> 1].y : rv[r >> 1].x;
            }
        }
        if (ASM) {   // nothing hidden is younger than the reservation here (the next tile's loads follow below)
            wait_vmcnt<0>();
            asm volatile("" : "+v"(rsv));
        }
        if (tid < F) {   // the reservations have had the scan and the staging above to come back
            s_delta[tid] = slot0 + tid * cap + rsv - start;
            if (rsv + cnt > cap) atomicOr(fx.overflow, 1u);   // rsv + cnt cannot wrap: both < 2^31
        }
        if (has_next) load_tile(t0 + TILE);   // the row registers are dead: the next tile streams in during the write-out
        lds_barrier();
        if (ASM) {
#pragma unroll 2
            for (int j = 0; j < RPT; ++j) {   // every store instruction is issued, whatever the data (golap_hidden.h)
                const uint32_t i = (uint32_t)j * NT + tid;
                const uint2 key = s_keys[i];
                const uint32_t rid = s_rids[i];
                const uint32_t b = bucket_of(hash_int32((int32_t)(key.x ^ key.y)), cfg);   // (rows beyond tn hold stale words: any bucket < F, store predicated off)
                const uint32_t gp = s_delta[b] + i;
                const uint32_t ok = (i < tn && gp < slot0 + (b + 1u) * cap) ? 1u : 0u;      // rows past a slot's end are dropped (overflow raised above)
                st8_hidden_if(out_keys + gp, u32x2{key.x, key.y}, ok);
                st4_hidden_if(out_rids + gp, rid, ok);
            }
        } else {
#pragma unroll 4
            for (int j = 0; j < RPT; ++j) {
                const uint32_t i = (uint32_t)j * NT + tid;
                if (i < tn) {
                    const uint2 key = s_keys[i];
                    const uint32_t b = bucket_of(hash_int32((int32_t)(key.x ^ key.y)), cfg);
                    const uint32_t gp = s_delta[b] + i;
                    if (gp < slot0 + (b + 1u) * cap) {
                        reinterpret_cast<uint2*>(out_keys)[gp] = key;
                        out_rids[gp] = s_rids[i];
                    }
                }
            }
        }
        // (the next LDS writes that could collide with this write-out -- s_start, s_delta, s_keys, s_rids -- all sit behind later barriers)
        if (ASM) {
            // the next tile's rows were requested before the STORES stores of the write-out above: everything older than those stores has
            // landed once at most STORES operations are outstanding.  (In front of the loop's exit test, not behind it: every path back
            // to the loop header passes this wait in the generated code too, which is what tests/test_isa_audit.py checks.)
            wait_vmcnt<STORES>();
#pragma unroll
            for (int h = 0; h < RPT / 2; ++h) { pin(kv[h]); pin(rv[h]); }
        }
        if (!has_next) break;
    }
}

}  // namespace golap
