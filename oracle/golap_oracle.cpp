// golap_oracle.cpp -- CPU ORACLE. TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE.
//
// A CPU restatement of the reference's GpuHashJoin / GpuAggregate / GpuFilter
// arithmetic (codebasecomprehension987/gpu-olap-engine), written from the
// algorithm its sources state.  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load this library; the shipped HIP path in
// gpu_olap_engine_amd/ never links, imports or calls it.
//
// PINNING STATUS
//   * filter mask            : pinned by the reference's own known-answer test
//                              gpu-olap-core/src/executor.rs:553-570
//                              ([5,15,25] > 10 -> [F,T,T]).
//   * strategy threshold     : pinned by gpu-kernel-compiler/src/join_kernel.rs:84-88.
//   * launch params          : pinned by gpu-kernel-compiler/src/filter_kernel.rs:86-95.
//   * aggregate identities   : pinned by gpu-kernel-compiler/src/aggregate_kernel.rs:138-143.
//   * 8-byte widening layout : pinned by arrow-interop/src/record_batch_convert.rs:201-217.
//   * hash values            : the reference's tests hold no vector, but join_kernels.cuh:26-34 names
//                              its hash: "MurmurHash3 32-bit finalizer", a PUBLISHED algorithm (fmix32
//                              of A. Appleby's MurmurHash3).  MurmurHash3_x86_32("", seed) == fmix32(seed),
//                              so tests/test_oracle.py::test_hashes_against_an_independent_murmurhash3 pins
//                              orc_hash_int32 / orc_hash_int64 / the first pass's partition ids against an
//                              independent implementation (scikit-learn's port of the reference C++) and
//                              the published empty-message vectors; the 64-bit fold is :36-41 verbatim.
//   * partition ids beyond the first pass, join row-id pairs, grouped sums:
//                              PARITY UNPINNED BY THE REFERENCE -- its tests hold no
//                              vector for them, its GPU kernels are never launched and
//                              its CPU join returns no rows (executor.rs:255-264).  The
//                              reference cannot be built here (Rust + CUDA toolchains
//                              absent; join_kernels.cuh needs <cuda_runtime.h>).  These
//                              are pinned instead by hand-evaluated vectors of
//                              join_kernels.cuh:26-41 committed in tests/golden/, by an
//                              independent numpy twin (tests/np_twin.py) and by independent
//                              implementations of the same relational semantics: Apache Arrow C++
//                              (Acero hash join / hash aggregate, pyarrow.compute) and pandas merge
//                              (tests/test_oracle_vs_arrow.py: joins of all four types incl. composite
//                              and NULL keys, filters with NULLs, grouped and ungrouped aggregates).
//
// Every function cites the reference file:line it follows (paths relative to
// the reference root).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <map>
#include <unordered_map>
#include <vector>

extern "C" {

// ---------------------------------------------------------------------------
// Hashes -- gpu-kernel-compiler/kernels/join_kernels.cuh:26-41
// ---------------------------------------------------------------------------

// join_kernels.cuh:26-35  MurmurHash3 32-bit finalizer.
uint32_t orc_hash_int32(int32_t key) {
    uint32_t h = (uint32_t)key;
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

// join_kernels.cuh:37-41  fold hi^lo (arithmetic shift on the signed key), then fmix32.
uint32_t orc_hash_int64(int64_t key) {
    uint32_t h = (uint32_t)(key ^ (key >> 32));
    return orc_hash_int32((int32_t)h);
}

// join_kernels.cuh:69  partition = (hash >> radix_shift) & (NUM_RADIX_PARTITIONS-1), RADIX_BITS=8 (:22-23).
uint32_t orc_partition_id(uint32_t hash, int radix_shift) {
    return (hash >> radix_shift) & 255u;
}

void orc_hash_int64_array(const int64_t* keys, uint64_t n, uint32_t* out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = orc_hash_int64(keys[i]);
}

// ---------------------------------------------------------------------------
// Radix partition -- join_kernels.cuh:45-76.
// The reference scatters with one atomicAdd per row on pre-seeded offsets, so
// the order inside a partition is unspecified; the oracle uses the stable
// (input) order, which is one of the orders the reference admits.
// offsets[257] receives the exclusive prefix sums the reference expects its
// caller to have seeded partition_offsets with.
// ---------------------------------------------------------------------------
void orc_radix_partition(const int64_t* keys, const uint32_t* row_ids, uint64_t n,
                         int radix_shift, int64_t* out_keys, uint32_t* out_rids,
                         uint64_t* offsets /*[257]*/) {
    uint64_t hist[256];
    memset(hist, 0, sizeof hist);
    for (uint64_t i = 0; i < n; ++i)
        hist[orc_partition_id(orc_hash_int64(keys[i]), radix_shift)]++;
    uint64_t run = 0;
    for (int p = 0; p < 256; ++p) { offsets[p] = run; run += hist[p]; }
    offsets[256] = run;
    uint64_t cur[256];
    memcpy(cur, offsets, sizeof cur);
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t p = orc_partition_id(orc_hash_int64(keys[i]), radix_shift);
        uint64_t pos = cur[p]++;                       // join_kernels.cuh:72
        out_keys[pos] = keys[i];                       // :74
        out_rids[pos] = row_ids ? row_ids[i] : (uint32_t)i;   // :75
    }
}

// ---------------------------------------------------------------------------
// Hash join, kernel semantics -- join_kernels.cuh:80-111 (build) + :115-166 (probe).
// Chained table: entries[n] = {hash,row_id,next}, buckets[capacity] seeded with
// 0xFFFFFFFF (:146), capacity = 1.5 x build size (README.md:114).
// Emits every (build_row_id, probe_row_id) with equal hash AND equal full key
// (:150-153), inner join.  Output is then put in the canonical order
// (probe_row_id, build_row_id) ascending, because the reference's order is
// unspecified (atomicAdd at :155).
// Returns the true match count (keeps counting past max_pairs, as :155-160);
// writes at most max_pairs pairs (interleaved u32 build_rid, probe_rid).
// ---------------------------------------------------------------------------
struct OrcEntry { uint32_t hash, row_id, next; };   // join_kernels.cuh:8-12

static inline bool pair_less(const uint64_t a, const uint64_t b) { return a < b; }

uint64_t orc_hash_join_i64(const int64_t* build_keys, uint64_t n_build,
                           const int64_t* probe_keys, uint64_t n_probe,
                           uint32_t build_rid_base, uint32_t probe_rid_base,
                           uint32_t* out_pairs, uint64_t max_pairs) {
    if (n_build == 0 || n_probe == 0) return 0;     // executor.rs:231-233
    uint32_t capacity = (uint32_t)std::max<uint64_t>(1, n_build + n_build / 2);
    std::vector<OrcEntry> entries(n_build);
    std::vector<uint32_t> buckets(capacity, 0xFFFFFFFFu);
    for (uint64_t i = 0; i < n_build; ++i) {        // build, :89-110
        uint32_t h = orc_hash_int64(build_keys[i]);
        uint32_t b = h % capacity;                  // :102
        entries[i].hash = h;
        entries[i].row_id = (uint32_t)i;
        entries[i].next = buckets[b];               // atomicExch, :109-110
        buckets[b] = (uint32_t)i;
    }
    // canonical key = probe_rid << 32 | build_rid  -> sort gives (probe, build) order
    std::vector<uint64_t> pairs;
    pairs.reserve(n_probe);
    for (uint64_t j = 0; j < n_probe; ++j) {        // probe, :129-165
        int64_t pk = probe_keys[j];
        uint32_t h = orc_hash_int64(pk);
        uint32_t e = buckets[h % capacity];         // :142-143
        while (e != 0xFFFFFFFFu) {                  // :146
            const OrcEntry& en = entries[e];
            if (en.hash == h && build_keys[en.row_id] == pk)   // :150-153
                pairs.push_back(((uint64_t)(probe_rid_base + (uint32_t)j) << 32) |
                                (uint64_t)(build_rid_base + en.row_id));
            e = en.next;                            // :164
        }
    }
    std::sort(pairs.begin(), pairs.end());
    uint64_t m = pairs.size();
    uint64_t w = std::min(m, max_pairs);
    if (out_pairs)
        for (uint64_t i = 0; i < w; ++i) {
            out_pairs[2 * i]     = (uint32_t)(pairs[i] & 0xFFFFFFFFu);   // build (left) row id
            out_pairs[2 * i + 1] = (uint32_t)(pairs[i] >> 32);           // probe (right) row id
        }
    return m;
}

// Outer variants of the same join (JoinType, gpu-olap-core/src/logical_plan.rs:53-58; the reference's executor
// and kernels only ever produce the inner result).  Left = build side preserved, Right = probe side preserved,
// Full = both; the missing side of an unmatched row is row id 0xFFFFFFFF (NULL).  Canonical order as above
// (NULL sorts last).  join_type: 0 Inner, 1 Left, 2 Right, 3 Full.
uint64_t orc_hash_join_outer_i64(const int64_t* build_keys, uint64_t n_build, const int64_t* probe_keys, uint64_t n_probe,
                                 int join_type, uint32_t* out_pairs, uint64_t max_pairs) {
    std::unordered_map<int64_t, std::vector<uint32_t>> m;
    for (uint64_t i = 0; i < n_build; ++i) m[build_keys[i]].push_back((uint32_t)i);
    std::vector<uint8_t> hit(n_build, 0);
    std::vector<uint64_t> pairs;
    const bool keep_build = join_type == 1 || join_type == 3, keep_probe = join_type == 2 || join_type == 3;
    for (uint64_t j = 0; j < n_probe; ++j) {
        auto it = m.find(probe_keys[j]);
        if (it == m.end()) {
            if (keep_probe) pairs.push_back(((uint64_t)(uint32_t)j << 32) | 0xFFFFFFFFull);
            continue;
        }
        for (uint32_t b : it->second) { hit[b] = 1; pairs.push_back(((uint64_t)(uint32_t)j << 32) | b); }
    }
    if (keep_build)
        for (uint64_t i = 0; i < n_build; ++i)
            if (!hit[i]) pairs.push_back((0xFFFFFFFFull << 32) | (uint32_t)i);
    std::sort(pairs.begin(), pairs.end());
    uint64_t mm = pairs.size(), w = std::min(mm, max_pairs);
    if (out_pairs)
        for (uint64_t i = 0; i < w; ++i) {
            out_pairs[2 * i] = (uint32_t)(pairs[i] & 0xFFFFFFFFu);
            out_pairs[2 * i + 1] = (uint32_t)(pairs[i] >> 32);
        }
    return mm;
}

// Equi-join on SEVERAL key columns with NULLs (physical_plan.rs:35-36 declares
// left_keys / right_keys as Vec<PhysicalExpr>, column_buffer.rs:59-60 a byte-per-row validity).  SQL semantics, which the
// reference's roadmap (README.md:411) names and none of its code reaches: two rows match iff EVERY key column is equal
// and none of them is NULL; a row with a NULL key matches nothing and, on a preserved side of an outer join, survives
// as an unmatched row.  keys: n_cols column pointers per side; valid: n_cols pointers per side, each null (all valid)
// or a byte per row (0 = NULL).  Canonical (probe, build) order like the joins above.
uint64_t orc_hash_join_multi(const int64_t* const* build_cols, const uint8_t* const* build_valid, uint64_t n_build,
                             const int64_t* const* probe_cols, const uint8_t* const* probe_valid, uint64_t n_probe, uint32_t n_cols,
                             int join_type, uint32_t* out_pairs, uint64_t max_pairs) {
    auto is_null = [&](const uint8_t* const* v, uint64_t i) {
        for (uint32_t c = 0; c < n_cols; ++c)
            if (v && v[c] && !v[c][i]) return true;
        return false;
    };
    std::map<std::vector<int64_t>, std::vector<uint32_t>> m;
    std::vector<int64_t> k(n_cols);
    for (uint64_t i = 0; i < n_build; ++i) {
        if (is_null(build_valid, i)) continue;
        for (uint32_t c = 0; c < n_cols; ++c) k[c] = build_cols[c][i];
        m[k].push_back((uint32_t)i);
    }
    std::vector<uint8_t> hit(n_build, 0);
    std::vector<uint64_t> pairs;
    const bool keep_build = join_type == 1 || join_type == 3, keep_probe = join_type == 2 || join_type == 3;
    for (uint64_t j = 0; j < n_probe; ++j) {
        bool matched = false;
        if (!is_null(probe_valid, j)) {
            for (uint32_t c = 0; c < n_cols; ++c) k[c] = probe_cols[c][j];
            auto it = m.find(k);
            if (it != m.end()) {
                matched = true;
                for (uint32_t b : it->second) { hit[b] = 1; pairs.push_back(((uint64_t)(uint32_t)j << 32) | b); }
            }
        }
        if (!matched && keep_probe) pairs.push_back(((uint64_t)(uint32_t)j << 32) | 0xFFFFFFFFull);
    }
    if (keep_build)
        for (uint64_t i = 0; i < n_build; ++i)
            if (!hit[i]) pairs.push_back((0xFFFFFFFFull << 32) | (uint32_t)i);
    std::sort(pairs.begin(), pairs.end());
    const uint64_t mm = pairs.size(), w = std::min(mm, max_pairs);
    if (out_pairs)
        for (uint64_t i = 0; i < w; ++i) {
            out_pairs[2 * i] = (uint32_t)(pairs[i] & 0xFFFFFFFFu);
            out_pairs[2 * i + 1] = (uint32_t)(pairs[i] >> 32);
        }
    return mm;
}

// ---------------------------------------------------------------------------
// Hash join, executor semantics -- gpu-olap-core/src/executor.rs:220-265.
// Build side = left, FIRST key column only, Int64, HashMap<i64, Vec<usize>>
// whose values are global row indices = running offset over the left batches
// (:238-253).  The reference stops there and returns an empty batch (:255-264);
// the probe it announces at :236 ("for each right row, look up in hash map")
// is restated here.  This is the single-threaded "faithful" CPU baseline.
// batch_lens describe how the left/right rows are split into RecordBatches.
// ---------------------------------------------------------------------------
uint64_t orc_execute_hash_join(const int64_t* left_keys, const uint64_t* left_batch_lens, uint32_t n_left_batches,
                               const int64_t* right_keys, const uint64_t* right_batch_lens, uint32_t n_right_batches,
                               uint32_t* out_pairs, uint64_t max_pairs) {
    if (n_left_batches == 0 || n_right_batches == 0) return 0;   // :231-233
    std::unordered_map<int64_t, std::vector<size_t>> build_map;   // :238
    size_t left_row_offset = 0;                                   // :239
    const int64_t* p = left_keys;
    for (uint32_t b = 0; b < n_left_batches; ++b) {               // :241
        for (uint64_t i = 0; i < left_batch_lens[b]; ++i)
            build_map[p[i]].push_back(left_row_offset + i);       // :249
        p += left_batch_lens[b];
        left_row_offset += left_batch_lens[b];                    // :252
    }
    std::vector<uint64_t> pairs;
    size_t right_row_offset = 0;
    p = right_keys;
    for (uint32_t b = 0; b < n_right_batches; ++b) {
        for (uint64_t j = 0; j < right_batch_lens[b]; ++j) {
            auto it = build_map.find(p[j]);
            if (it == build_map.end()) continue;
            for (size_t l : it->second)
                pairs.push_back(((uint64_t)(uint32_t)(right_row_offset + j) << 32) | (uint64_t)(uint32_t)l);
        }
        p += right_batch_lens[b];
        right_row_offset += right_batch_lens[b];
    }
    std::sort(pairs.begin(), pairs.end());
    uint64_t m = pairs.size(), w = std::min(m, max_pairs);
    if (out_pairs)
        for (uint64_t i = 0; i < w; ++i) {
            out_pairs[2 * i] = (uint32_t)(pairs[i] & 0xFFFFFFFFu);
            out_pairs[2 * i + 1] = (uint32_t)(pairs[i] >> 32);
        }
    return m;
}

// Same join without materialising/sorting pairs: returns the match count and an
// order-independent checksum.  Used as the timed single-thread CPU baseline
// (build + probe only, like the GPU timed region, no canonical sort).
//   checksum = sum over matches of mix(build_rid, probe_rid)  (mod 2^64)
static inline uint64_t pair_mix(uint32_t b, uint32_t p) {
    uint64_t z = ((uint64_t)p << 32 | b) + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
uint64_t orc_pair_mix(uint32_t b, uint32_t p) { return pair_mix(b, p); }

uint64_t orc_pairs_checksum(const uint32_t* pairs, uint64_t m) {
    uint64_t s = 0;
    for (uint64_t i = 0; i < m; ++i) s += pair_mix(pairs[2 * i], pairs[2 * i + 1]);
    return s;
}

uint64_t orc_execute_hash_join_count(const int64_t* left_keys, uint64_t n_left,
                                     const int64_t* right_keys, uint64_t n_right,
                                     uint64_t* checksum_out) {
    std::unordered_map<int64_t, std::vector<size_t>> build_map;
    for (uint64_t i = 0; i < n_left; ++i) build_map[left_keys[i]].push_back(i);
    uint64_t m = 0, cs = 0;
    for (uint64_t j = 0; j < n_right; ++j) {
        auto it = build_map.find(right_keys[j]);
        if (it == build_map.end()) continue;
        for (size_t l : it->second) { cs += pair_mix((uint32_t)l, (uint32_t)j); ++m; }
    }
    if (checksum_out) *checksum_out = cs;
    return m;
}

// "Fair" CPU baseline: same semantics, radix-partitioned
// on orc_hash_int64 low 8 bits (join_kernels.cuh:69) and multi-threaded over
// n_threads host threads with std::thread; each partition is joined with the
// kernel-semantics chained table.  Returns match count + checksum.
uint64_t orc_radix_join_mt(const int64_t* build_keys, uint64_t n_build,
                           const int64_t* probe_keys, uint64_t n_probe,
                           uint32_t n_threads, uint64_t* checksum_out) {
    if (n_threads == 0) n_threads = 1;
    std::vector<int64_t> bk(n_build), pk(n_probe);
    std::vector<uint32_t> br(n_build), pr(n_probe);
    uint64_t boff[257], poff[257];
    std::thread t1([&] { orc_radix_partition(build_keys, nullptr, n_build, 0, bk.data(), br.data(), boff); });
    orc_radix_partition(probe_keys, nullptr, n_probe, 0, pk.data(), pr.data(), poff);
    t1.join();
    std::atomic<uint32_t> next_p(0);
    std::vector<uint64_t> ms(n_threads, 0), css(n_threads, 0);
    auto work = [&](uint32_t tid) {
        uint64_t m = 0, cs = 0;
        std::vector<OrcEntry> entries;
        std::vector<uint32_t> buckets;
        for (;;) {
            uint32_t p = next_p.fetch_add(1);
            if (p >= 256) break;
            uint64_t nb = boff[p + 1] - boff[p], np = poff[p + 1] - poff[p];
            if (!nb || !np) continue;
            uint32_t cap = (uint32_t)(nb + nb / 2);
            entries.resize(nb);
            buckets.assign(cap, 0xFFFFFFFFu);
            const int64_t* k = bk.data() + boff[p];
            const uint32_t* r = br.data() + boff[p];
            for (uint64_t i = 0; i < nb; ++i) {
                uint32_t h = orc_hash_int64(k[i]);
                uint32_t b = h % cap;
                entries[i] = {h, (uint32_t)i, buckets[b]};
                buckets[b] = (uint32_t)i;
            }
            const int64_t* q = pk.data() + poff[p];
            const uint32_t* qr = pr.data() + poff[p];
            for (uint64_t j = 0; j < np; ++j) {
                uint32_t h = orc_hash_int64(q[j]);
                uint32_t e = buckets[h % cap];
                while (e != 0xFFFFFFFFu) {
                    const OrcEntry& en = entries[e];
                    if (en.hash == h && k[en.row_id] == q[j]) { cs += pair_mix(r[en.row_id], qr[j]); ++m; }
                    e = en.next;
                }
            }
        }
        ms[tid] = m; css[tid] = cs;
    };
    std::vector<std::thread> ths;
    for (uint32_t t = 1; t < n_threads; ++t) ths.emplace_back(work, t);
    work(0);
    for (auto& t : ths) t.join();
    uint64_t m = 0, cs = 0;
    for (uint32_t t = 0; t < n_threads; ++t) { m += ms[t]; cs += css[t]; }
    if (checksum_out) *checksum_out = cs;
    return m;
}

// ---------------------------------------------------------------------------
// Filter -- gpu-olap-core/src/executor.rs:443-476 (eval_filter_mask),
//           :478-507 (filter_record_batch / take_int64).
// op strings exactly as executor.rs:457-465; an unknown op selects no row (:464).
// ---------------------------------------------------------------------------
enum { ORC_OP_EQ = 0, ORC_OP_NE, ORC_OP_LT, ORC_OP_LE, ORC_OP_GT, ORC_OP_GE, ORC_OP_UNKNOWN };

int orc_parse_cmp_op(const char* op) {
    if (!strcmp(op, "=") || !strcmp(op, "==")) return ORC_OP_EQ;    // executor.rs:458
    if (!strcmp(op, "!=") || !strcmp(op, "<>")) return ORC_OP_NE;   // :459
    if (!strcmp(op, ">")) return ORC_OP_GT;                         // :460
    if (!strcmp(op, ">=")) return ORC_OP_GE;                        // :461
    if (!strcmp(op, "<")) return ORC_OP_LT;                         // :462
    if (!strcmp(op, "<=")) return ORC_OP_LE;                        // :463
    return ORC_OP_UNKNOWN;                                          // :464
}

static inline bool cmp_i64(int op, int64_t l, int64_t r) {
    switch (op) {
        case ORC_OP_EQ: return l == r;
        case ORC_OP_NE: return l != r;
        case ORC_OP_LT: return l < r;
        case ORC_OP_LE: return l <= r;
        case ORC_OP_GT: return l > r;
        case ORC_OP_GE: return l >= r;
        default: return false;
    }
}

// col[i] <op> literal  ->  mask[i] in {0,1}   (u8 per row, filter_kernel.rs:3-5)
void orc_eval_filter_mask_i64(const int64_t* col, uint64_t n, const char* op, int64_t literal, uint8_t* mask) {
    int o = orc_parse_cmp_op(op);
    for (uint64_t i = 0; i < n; ++i) mask[i] = cmp_i64(o, col[i], literal) ? 1 : 0;
}

// executor.rs:479-484: indices of kept rows, original order.  Returns count.
uint64_t orc_mask_to_indices(const uint8_t* mask, uint64_t n, uint32_t* indices) {
    uint64_t c = 0;
    for (uint64_t i = 0; i < n; ++i) if (mask[i]) indices[c++] = (uint32_t)i;
    return c;
}

// executor.rs:496-507 take (Int64 and Float64 are both plain 8-byte moves).
void orc_take_8byte(const uint64_t* col, const uint32_t* indices, uint64_t m, uint64_t* out) {
    for (uint64_t i = 0; i < m; ++i) out[i] = col[indices[i]];
}

// ---------------------------------------------------------------------------
// Aggregate, executor semantics -- executor.rs:297-355 + :509-527.
// Ungrouped; every input is converted to f64 (Int64 `as f64`, :521) and folded
// sequentially in row order: count u64, sum f64 (0.0), min seeded f64::MAX,
// max seeded f64::MIN (= -f64::MAX) (:308-311, :316-325).
// out[5] = {sum, count as f64, min, max, avg (0.0 when count==0)} (:334-346).
// ---------------------------------------------------------------------------
static const double F64_MAX = 1.7976931348623157e308;

void orc_execute_aggregate_f64(const double* v, uint64_t n, double out[5]) {
    double sum = 0.0, mn = F64_MAX, mx = -F64_MAX;
    uint64_t cnt = 0;
    for (uint64_t i = 0; i < n; ++i) {
        cnt += 1; sum += v[i];
        if (v[i] < mn) mn = v[i];
        if (v[i] > mx) mx = v[i];
    }
    out[0] = sum; out[1] = (double)cnt; out[2] = mn; out[3] = mx;
    out[4] = cnt > 0 ? sum / (double)cnt : 0.0;
}

void orc_execute_aggregate_i64(const int64_t* v, uint64_t n, double out[5]) {
    double sum = 0.0, mn = F64_MAX, mx = -F64_MAX;
    uint64_t cnt = 0;
    for (uint64_t i = 0; i < n; ++i) {
        double x = (double)v[i];                    // executor.rs:521
        cnt += 1; sum += x;
        if (x < mn) mn = x;
        if (x > mx) mx = x;
    }
    out[0] = sum; out[1] = (double)cnt; out[2] = mn; out[3] = mx;
    out[4] = cnt > 0 ? sum / (double)cnt : 0.0;
}

// BASELINE config 1: SELECT sum(v) FROM t WHERE k < c  ==  execute_filter
// (executor.rs:161-178) feeding execute_aggregate (:297-355): mask, gather v,
// then sequential f64 sum of the kept rows.
double orc_filter_sum_i64(const int64_t* k, const int64_t* v, uint64_t n, const char* op, int64_t literal,
                          uint64_t* kept_out) {
    int o = orc_parse_cmp_op(op);
    double sum = 0.0; uint64_t kept = 0;
    for (uint64_t i = 0; i < n; ++i)
        if (cmp_i64(o, k[i], literal)) { sum += (double)v[i]; ++kept; }
    if (kept_out) *kept_out = kept;
    return sum;
}

// aggregate_kernel.rs:27-33 identity values.
int64_t orc_agg_identity_value(int func /*0 Sum,1 Count,2 Min,3 Max,4 Avg*/) {
    switch (func) {
        case 2: return INT64_MAX;
        case 3: return INT64_MIN;
        default: return 0;
    }
}

// ---------------------------------------------------------------------------
// Grouped aggregate.  The reference's executor ignores group_by
// (executor.rs:307) and its kernel folds key % num_groups (join_kernels.cuh:230,
// a defect: distinct keys are merged) -- so the grouped semantics follow the
// operator's stated contract (aggregate_kernel.rs:80-92, :17-34): one slot per
// DISTINCT key, per-group {sum,count,min,max} with executor.rs:313-327
// arithmetic applied in row order within each group; rows sorted by key.
// Returns number of groups; out_keys[g], out[g*5 + {sum,count,min,max,avg}].
// ---------------------------------------------------------------------------
uint64_t orc_group_aggregate(const int64_t* keys, const double* vals, uint64_t n,
                             int64_t* out_keys, double* out, uint64_t max_groups) {
    struct Acc { double sum; uint64_t cnt; double mn, mx; };
    std::unordered_map<int64_t, Acc> m;
    for (uint64_t i = 0; i < n; ++i) {
        auto it = m.find(keys[i]);
        if (it == m.end()) it = m.emplace(keys[i], Acc{0.0, 0, F64_MAX, -F64_MAX}).first;
        Acc& a = it->second;
        a.cnt += 1; a.sum += vals[i];
        if (vals[i] < a.mn) a.mn = vals[i];
        if (vals[i] > a.mx) a.mx = vals[i];
    }
    std::vector<int64_t> ks;
    ks.reserve(m.size());
    for (auto& kv : m) ks.push_back(kv.first);
    std::sort(ks.begin(), ks.end());
    uint64_t g = 0;
    for (int64_t k : ks) {
        if (g >= max_groups) break;
        const Acc& a = m[k];
        out_keys[g] = k;
        out[g * 5 + 0] = a.sum; out[g * 5 + 1] = (double)a.cnt; out[g * 5 + 2] = a.mn; out[g * 5 + 3] = a.mx;
        out[g * 5 + 4] = a.cnt ? a.sum / (double)a.cnt : 0.0;
        ++g;
    }
    return ks.size();
}

// join_kernels.cuh:216-240 as written (direct-mapped SUM by key % num_groups,
// integer add for 8-byte values).  Kept so tests can show where the new path
// deliberately departs from the reference kernel's defect.
void orc_hash_aggregate_kernel_i64(const uint32_t* group_keys, const int64_t* values, uint64_t n,
                                   int64_t* aggregates, uint32_t num_groups) {
    for (uint64_t i = 0; i < n; ++i)
        aggregates[group_keys[i] % num_groups] += values[i];     // :230, :237
}

// ---------------------------------------------------------------------------
// API constants the reference's unit tests pin.
// ---------------------------------------------------------------------------
// join_kernel.rs:71-77: <= 1_000_000 build rows -> Broadcast(2) else Radix(0).
int orc_choose_strategy(uint64_t n_build) { return n_build <= 1000000ull ? 2 : 0; }

// filter_kernel.rs:75-79 / aggregate_kernel.rs:127-131: block 256, grid ceil(n/256) >= 1
// (the u32 truncation of n_rows is the reference's own).
void orc_compute_launch_params(uint64_t n_rows, uint32_t* grid, uint32_t* block) {
    const uint32_t B = 256;
    uint32_t g = ((uint32_t)n_rows + B - 1) / B;
    *grid = g > 1 ? g : 1; *block = B;
}

// ---------------------------------------------------------------------------
// arrow-interop widening -- record_batch_convert.rs:102-110 (null -> 0, 8-byte LE),
// :123-130 FNV-1a 64.
// ---------------------------------------------------------------------------
void orc_widen_i32_to_i64(const int32_t* in, const uint8_t* validity /*nullable, byte per row*/, uint64_t n, int64_t* out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = (validity && !validity[i]) ? 0 : (int64_t)in[i];
}
void orc_widen_f32_to_f64(const float* in, const uint8_t* validity, uint64_t n, double* out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = (validity && !validity[i]) ? 0.0 : (double)in[i];
}
uint64_t orc_fnv1a_hash(const uint8_t* s, uint64_t len) {
    uint64_t h = 14695981039346656037ull;
    for (uint64_t i = 0; i < len; ++i) { h ^= s[i]; h *= 1099511628211ull; }
    return h;
}

// ---------------------------------------------------------------------------
// Synthetic generators -- SURVEY.md section 8(d) / BASELINE.md section 3.
// sm64(seed,i) = splitmix64 finalizer of seed + i*0x9E3779B97F4A7C15.
// ---------------------------------------------------------------------------
static inline uint64_t sm64(uint64_t seed, uint64_t i) {
    uint64_t z = seed + i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
uint64_t orc_sm64(uint64_t seed, uint64_t i) { return sm64(seed, i); }

// build key[i] = (A*i + B) mod N, A = 2654435761, B = 12345 (unique keys when gcd(A,N)=1)
void orc_gen_build_keys(int64_t* out, uint64_t first, uint64_t count, uint64_t N) {
    for (uint64_t i = 0; i < count; ++i) out[i] = (int64_t)((2654435761ull * (first + i) + 12345ull) % N);
}
// probe key[j] = sm64(43, j) mod N
void orc_gen_probe_keys(int64_t* out, uint64_t first, uint64_t count, uint64_t N) {
    for (uint64_t i = 0; i < count; ++i) out[i] = (int64_t)(sm64(43, first + i) % N);
}
// generic: out[i] = sm64(seed, first+i) mod modulus  (dup variant uses modulus N/2)
void orc_gen_sm64_mod(int64_t* out, uint64_t seed, uint64_t first, uint64_t count, uint64_t modulus) {
    for (uint64_t i = 0; i < count; ++i) out[i] = (int64_t)(sm64(seed, first + i) % modulus);
}
void orc_gen_sm64_mod_i32(int32_t* out, uint64_t seed, uint64_t first, uint64_t count, uint64_t modulus) {
    for (uint64_t i = 0; i < count; ++i) out[i] = (int32_t)(sm64(seed, first + i) % modulus);
}
void orc_gen_sm64_mod_f64(double* out, uint64_t seed, uint64_t first, uint64_t count, uint64_t modulus) {
    for (uint64_t i = 0; i < count; ++i) out[i] = (double)(sm64(seed, first + i) % modulus);
}

// wall clock helper for bench.py's cpu_baseline leg
double orc_now_seconds(void) {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

}  // extern "C"
