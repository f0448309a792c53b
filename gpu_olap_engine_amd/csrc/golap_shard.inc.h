// golap_shard.inc.h -- multi-GPU sharded radix hash join; included at the end of golap_join.hip (it uses
// that file's planning helpers and kernels).  One process per GPU; the only data-path collective is the partition
// exchange (grouped ncclSend/ncclRecv over xGMI, all links of a GPU busy at once), plus tiny all-gathers.
//
// Fast path (every key in [0, 2^32), large inputs), per rank, all on the caller's stream + the communicator's stream:
//   partition(build) -> [all-gather bucket counts, exchange(build)] || partition(probe) -> [all-gather, exchange(probe)]
//   || ring pass(build) -> ring pass(probe) -> LDS join -> status all-gather -> ONE host read of the status words.
//   * the exchange-time partition is the single-GPU first pass itself (k_part1_narrow: no histogram, bucket ranges
//     reserved by atomics, 8-byte (key32, row id) tuples), laid out owner-major;
//   * because every bucket owns a range of `cap` rows whose size depends only on the source rank's row count, both
//     ends of a message know its size without any size exchange: the slice of rank r's buffer that holds the buckets
//     owned by rank d is sent whole (slack included: +0.4 % at 125M rows per rank); the row counts inside
//     (the reservation cursors) are all-gathered device-to-device and only ever read by kernels;
//   * the receiving side's last pass is the single-GPU last pass (k_part2_ring) fed by one segment per source.
// Exact path (anything else): histogram partition (64-bit rows), bucket sizes all-gathered and read back by the host,
// variable-size exchange, segment-fed local passes (run_side) -- the round-1 Python orchestration, now in C++.
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace golap {

// ---------------------------------------------------------------------------
// RCCL, bound at run time (no link-time dependency: the library loads on hosts without RCCL; torch ships its own copy
// and an already loaded one is reused)
// ---------------------------------------------------------------------------
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi* rccl_api(golap_ctx* ctx) {
    static std::mutex mu;
    static RcclApi api;
    static bool tried = false;
    std::lock_guard<std::mutex> lk(mu);
    if (tried) {
        if (!api.handle) fail(ctx, GOLAP_ERR_UNSUPPORTED, "librccl is not available (an earlier attempt to load it failed)");
        return api.handle ? &api : nullptr;
    }
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;   // a copy some other library (torch) has already loaded
    for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fail(ctx, GOLAP_ERR_UNSUPPORTED, "librccl not found (%s)", dlerror()); return nullptr; }
#define GOLAP_SYM(field, name)                                                                      \
    *(void**)&api.field = dlsym(h, name);                                                           \
    if (!api.field) { fail(ctx, GOLAP_ERR_UNSUPPORTED, "librccl lacks %s", name); return nullptr; }
    GOLAP_SYM(GetUniqueId, "ncclGetUniqueId");
    GOLAP_SYM(CommInitRank, "ncclCommInitRank");
    GOLAP_SYM(CommDestroy, "ncclCommDestroy");
    GOLAP_SYM(AllGather, "ncclAllGather");
    GOLAP_SYM(Send, "ncclSend");
    GOLAP_SYM(Recv, "ncclRecv");
    GOLAP_SYM(GroupStart, "ncclGroupStart");
    GOLAP_SYM(GroupEnd, "ncclGroupEnd");
    GOLAP_SYM(GetErrorString, "ncclGetErrorString");
#undef GOLAP_SYM
    api.handle = h;
    return &api;
}

}  // namespace golap

struct golap_comm {
    golap_ctx* ctx = nullptr;
    int32_t world = 1, rank = 0;
    bool builtin = false;
    golap_transport tr{};
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr;          // the exchange runs here, beside the caller's compute stream
    static constexpr int MAX_CHUNKS = 8;
    hipEvent_t ev_part[2][MAX_CHUNKS] = {}, ev_xchg[2] = {nullptr, nullptr}, ev_half[2] = {nullptr, nullptr}, ev_hdr = nullptr, ev_misc = nullptr;   // ordering only (no timing)
    hipEvent_t ev_keys = nullptr, ev_agree = nullptr;   // payload of the agreement round written / agreement words (and payload) gathered
    hipEvent_t ev_t[2 * GOLAP_SHARD_STAGES + 1] = {};   // stage begin / end + call start (timing events; recorded only with profiling on)
    uint32_t* h_status = nullptr;          // pinned: status words / size matrices read back by the host
    uint64_t h_status_bytes = 0;
    // agreement round: one 16-word unit per rank -- [0] the agreement word, [1 .. 16) payload written on the device (the fast path's summary
    // of its upper key halves).  d_agree: [0 .. 16) this rank's unit, [16 + 16 r ..) rank r's (all-gathered); never in the workspace.
    // h_agree: [0] word to send, [16 + 16 r] words received, [AGREE_SCRATCH ..) small host-to-device staging of the callers.
    static constexpr uint32_t AGREE_UNIT = 16, AGREE_MAX_WORLD = 128, AGREE_SCRATCH = 16 + 16 * 128, AGREE_WORDS = 16 + 16 * 128 + 256;
    uint32_t* d_agree = nullptr;
    uint32_t* h_agree = nullptr;
    // Bit 31 of the agreement word is not an error: it says "this rank runs the call in its GENERAL form" (golap_sharded_join: key columns
    // packed first because it was handed validity arrays).  Whether a shard carries NULLs is a property of the DATA, so ranks may differ;
    // the first agreement round of a call therefore also settles the form: agree_mode = what this rank posts, agree_seen_* = what the
    // last round showed over ALL ranks (shard_agree_wait).
    static constexpr uint32_t AGREE_GENERAL = 0x80000000u;
    uint32_t agree_mode = 0;
    bool agree_seen_general = false, agree_seen_plain = false;
    golap_sharded_join_info info{};
};

namespace golap {

#define GOLAP_NCCL_TRY(ctx, api, expr)                                                                                     \
    do {                                                                                                                   \
        ncclResult_t _r = (expr);                                                                                          \
        if (_r != ncclSuccess) return fail((ctx), GOLAP_ERR_HIP, "%s failed: %s", #expr, (api)->GetErrorString(_r));        \
    } while (0)

static int32_t rccl_all_gather(void* user, uint64_t send_ptr, uint64_t recv_ptr, uint64_t bytes, void* stream) {
    golap_comm* cm = (golap_comm*)user;
    RcclApi* api = rccl_api(cm->ctx);
    if (!api) return GOLAP_ERR_UNSUPPORTED;
    GOLAP_NCCL_TRY(cm->ctx, api, api->AllGather((const void*)send_ptr, (void*)recv_ptr, (size_t)bytes, ncclUint8, cm->nccl, (hipStream_t)stream));
    return GOLAP_OK;
}

static int32_t rccl_exchange(void* user, const golap_p2p_op* ops, uint32_t n_ops, void* stream) {
    golap_comm* cm = (golap_comm*)user;
    RcclApi* api = rccl_api(cm->ctx);
    if (!api) return GOLAP_ERR_UNSUPPORTED;
    if (n_ops == 0) return GOLAP_OK;
    GOLAP_NCCL_TRY(cm->ctx, api, api->GroupStart());
    for (uint32_t i = 0; i < n_ops; ++i) {
        const golap_p2p_op& o = ops[i];
        // byte counts as size_t of ncclUint8; callers keep every message <= max_msg_bytes (see split_message)
        ncclResult_t r = o.is_send ? api->Send((const void*)o.ptr, (size_t)o.bytes, ncclUint8, o.peer, cm->nccl, (hipStream_t)stream)
                                   : api->Recv((void*)o.ptr, (size_t)o.bytes, ncclUint8, o.peer, cm->nccl, (hipStream_t)stream);
        if (r != ncclSuccess) {
            (void)api->GroupEnd();
            return fail(cm->ctx, GOLAP_ERR_HIP, "ncclSend/ncclRecv failed: %s", api->GetErrorString(r));
        }
    }
    GOLAP_NCCL_TRY(cm->ctx, api, api->GroupEnd());
    return GOLAP_OK;
}

// A message of `bytes` bytes between two ranks travels as ceil(bytes / max_msg) pieces; both ends derive the same list.
// (One 8 GB ncclSend/ncclRecv pair was observed to deliver wrong data on this stack -- round 1, scripts/dbg_sharded.py
// CHECK_BIG -- and tests/test_sharded_gpu.py::test_rccl_message_size_boundary pins where that starts; bounded pieces
// also let a transport pipeline them.)
// default bound of one message: half of the largest size verified intact (1 GiB; see golap_comm_selftest)
constexpr uint64_t SHARD_DEFAULT_MAX_MSG = 1ull << 29;
constexpr uint64_t SHARD_RCCL_MAX_MSG = 1ull << 30;   // the built-in transport never sends more in one message, whatever the caller asks for:
                                                      // beyond it this RCCL build delivers only the first half (golap_comm_selftest)
static uint64_t shard_max_msg(const golap_comm* cm, uint64_t asked) {
    uint64_t m = asked ? asked : SHARD_DEFAULT_MAX_MSG;
    if (cm->builtin && m > SHARD_RCCL_MAX_MSG) m = SHARD_RCCL_MAX_MSG;
    return m;
}

static void split_message(int32_t peer, bool is_send, uint64_t ptr, uint64_t bytes, uint64_t max_msg, std::vector<golap_p2p_op>& ops) {
    for (uint64_t o = 0; o < bytes; o += max_msg) {
        golap_p2p_op op;
        op.peer = peer;
        op.is_send = is_send ? 1 : 0;
        op.ptr = ptr + o;
        op.bytes = std::min(max_msg, bytes - o);
        ops.push_back(op);
    }
}

static void rows_per_rank(uint64_t n_total, uint32_t world, const uint64_t* given, std::vector<uint64_t>& out) {
    out.resize(world);
    for (uint32_t r = 0; r < world; ++r) {
        if (given) out[r] = given[r];
        else {
            uint64_t lo, hi;
            golap_shard_range(n_total, r, world, &lo, &hi);
            out[r] = hi - lo;
        }
    }
}

// ---------------------------------------------------------------------------
// fast-path plan (host only; every rank computes the same numbers)
// ---------------------------------------------------------------------------
struct ShardPlan {
    bool eligible = false;
    uint32_t world = 1, e1 = 0, F1 = 256, per = 256, local_bits = 0, F2 = 1;
    uint32_t nsub = 1;                         // sub-ranges per bucket of the exchange-time partition (ReserveArgs::nsub): a bucket is nsub * cap rows
    uint32_t K = 1;                            // row chunks per side: chunk k is shipped while chunk k + 1 is partitioned
    std::vector<uint64_t> rows[2];             // [side][rank]
    std::vector<uint64_t> chunk_rows[2];       // [side][rank] rows of chunks 0 .. K - 2 (whole first-pass tiles); the last chunk takes the rest
    std::vector<uint64_t> cap[2];              // [side][rank * K + chunk] rows provisioned per (bucket, sub-range)
    JoinPlan local;                            // the one local pass
    uint64_t rows_of(int side, uint32_t r, uint32_t c) const {
        const uint64_t n = rows[side][r], cr = chunk_rows[side][r], lo = std::min<uint64_t>(n, (uint64_t)c * cr);
        return c + 1 == K ? n - lo : std::min<uint64_t>(n, lo + cr) - lo;
    }
};

static uint32_t exchange_extra_bits(uint64_t n_build_total) {
    JoinPlan p;
    make_plan(std::max<uint64_t>(1, n_build_total), 0, true, p);
    const int32_t over = (int32_t)p.total_bits - 8 - 9;   // hash bits one local pass (<= 9) cannot resolve
    return (uint32_t)std::max(0, std::min(2, over));
}

// Row chunks per side.  With peers, a rank's first message can leave once its first chunk is partitioned instead of after the whole
// side (at 8 GPUs the partition pass of a side, ~0.4 ms, is a seventh of the join); alone there is nothing to overlap.  Every chunk
// must still be a launch the first-pass kernel is efficient at.  GOLAP_SHARD_CHUNKS: A/B and test hook.
static uint32_t shard_chunks(uint32_t world, uint64_t min_rows_per_rank, bool forced_small) {
    uint32_t k = world > 1 ? 4u : 1u;
    if (const char* e = knob("GOLAP_SHARD_CHUNKS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= golap_comm::MAX_CHUNKS) return (uint32_t)v;
    }
    while (k > 1 && !forced_small && !narrow_size_ok(min_rows_per_rank / k)) k >>= 1;
    return k;
}

static void make_shard_plan(uint32_t world, uint64_t n_build_total, uint64_t n_probe_total, const uint64_t* b_rows, const uint64_t* p_rows,
                            ShardPlan& sp) {
    sp = ShardPlan();
    sp.world = world;
    rows_per_rank(n_build_total, world, b_rows, sp.rows[0]);
    rows_per_rank(n_probe_total, world, p_rows, sp.rows[1]);
    sp.e1 = exchange_extra_bits(n_build_total);
    sp.F1 = 256u << sp.e1;
    sp.per = sp.F1 / world;
    golap_join_partitioned_plan pl;
    memset(&pl, 0, sizeof pl);
    pl.n_groups = sp.per;
    pl.n_sources = world;
    pl.pre_bits = 8 + sp.e1;
    pl.total_build_rows_hint = std::max<uint64_t>(1, n_build_total / world);
    make_local_plan(&pl, sp.local);
    sp.local_bits = sp.local.total_bits - pl.pre_bits;
    sp.F2 = sp.local.passes.empty() ? 1 : sp.local.passes[0].cfg.F;
    const char* env = knob("GOLAP_SHARD_FAST");   // test hook: "1" drops the size conditions, "0" disables the fast path
    const bool forced = env && !strcmp(env, "1");
    uint64_t min_rows = ~0ull;
    for (int side = 0; side < 2; ++side)
        for (uint32_t r = 0; r < world; ++r) min_rows = std::min(min_rows, sp.rows[side][r]);
    sp.K = shard_chunks(world, min_rows, forced);
    // world * K * nsub pieces per group in the ring pass
    sp.nsub = std::max<uint32_t>(1, std::min<uint32_t>(narrow_nsub(), (uint32_t)P2R_MAX_SRC / (world * sp.K)));
    const uint32_t tile = part1_narrow_tile();
    for (int side = 0; side < 2; ++side) {
        sp.cap[side].resize((size_t)world * sp.K);
        sp.chunk_rows[side].resize(world);
        for (uint32_t r = 0; r < world; ++r) {
            sp.chunk_rows[side][r] = round_up(std::max<uint64_t>(1, ceil_div(sp.rows[side][r], sp.K)), tile);
            for (uint32_t c = 0; c < sp.K; ++c) {
                const uint64_t n_c = sp.rows_of(side, r, c);
                uint32_t cr = 0;
                const uint32_t chunks = narrow_chunks(n_c, &cr);
                sp.cap[side][(size_t)r * sp.K + c] = reserve_cap_sub(n_c, sp.F1, sp.nsub, chunks, cr);
            }
        }
    }
    // the fast path is the two-pass tuple pipeline: one local pass with a ring-sized fan-out, every first-pass launch large
    // enough for the big-tile kernel, positions within u32
    bool ok = world * sp.K <= (uint32_t)P2R_MAX_SRC && sp.local.passes.size() == 1 && sp.F2 >= 32 && sp.F2 <= (uint32_t)P2R_MAX_F && sp.per <= 1024;
    for (int side = 0; side < 2 && ok && !forced; ++side)
        for (uint32_t r = 0; r < world && ok; ++r)
            if (!narrow_size_ok(sp.rows[side][r] / sp.K)) ok = false;
    for (int side = 0; side < 2 && ok; ++side)
        for (uint32_t r = 0; r < world && ok; ++r) {
            uint64_t send = 0;
            for (uint32_t c = 0; c < sp.K; ++c) send += sp.cap[side][(size_t)r * sp.K + c] * sp.F1 * sp.nsub;
            if (send >= 0xFFFFFFF0ull) ok = false;
        }
    if (env && !strcmp(env, "0")) ok = false;
    if (const char* e2 = knob("GOLAP_NARROW")) if (!strcmp(e2, "0")) ok = false;
    sp.eligible = ok;
}

// ---------------------------------------------------------------------------
// device helpers of the fast path
// ---------------------------------------------------------------------------
// Upper key halves across ranks.  The tuple pipeline needs ONE frame of reference on every rank (PassCfg::key_hi): each rank summarises
// a sample of its shard (k_key_summary -> the payload of its agreement unit), the units are all-gathered with the agreement words, and
// every rank derives the same words from the same P summaries (k_key_combine): one upper half everywhere -- D = 1, any value -- or
// several inside a window as wide as the partition's fan-out -- the offset form of the first pass (golap_part_narrow.h, DICT = 2), also
// when every rank holds ONE half of its own (a table sharded by tenant) -- or up to four halves far apart: the dictionary form.  Anything
// else raises the build side's failure word: the
// partition kernels return at once and the call takes the exact path, as it did for every key beyond 2^32 before.
//   summary (agreement unit words 1 .. 7): [1] distinct sampled halves (1 .. 4; 5 = more), [2 .. 5] the first four, [6] signed min, [7] signed max
__global__ __launch_bounds__(1024) void k_key_summary(const long long* __restrict__ bkeys, unsigned long long nb, const long long* __restrict__ pkeys,
                                                     unsigned long long np, uint32_t* __restrict__ unit) {
    constexpr uint32_t SLOTS = 2048;
    __shared__ unsigned long long s_set[SLOTS];
    __shared__ uint32_t s_vals[KEY_DICT_MAX + 1], s_n;
    __shared__ int s_min, s_max;
    const uint32_t tid = threadIdx.x;
    if (nb == 0 && np == 0) {   // an empty shard has nothing to say
        if (tid == 0) { unit[1] = 0; unit[6] = 0; unit[7] = 0; }
        return;
    }
    const uint32_t hi0 = (uint32_t)((unsigned long long)(nb ? bkeys[0] : pkeys[0]) >> 32);
    uint32_t smp[16];
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
        const unsigned long long s = (unsigned long long)j * 1024 + tid;   // sample s of 8192 per side
        smp[2 * j] = nb ? (uint32_t)((unsigned long long)bkeys[(s * nb) >> 13] >> 32) : hi0;
        smp[2 * j + 1] = np ? (uint32_t)((unsigned long long)pkeys[(s * np) >> 13] >> 32) : hi0;
    }
    for (uint32_t i = tid; i < SLOTS; i += 1024) s_set[i] = 0ull;
    if (tid == 0) { s_n = 0; s_min = (int)hi0; s_max = (int)hi0; }
    __syncthreads();
    int mn = (int)smp[0], mx = (int)smp[0];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        mn = (int)smp[j] < mn ? (int)smp[j] : mn;
        mx = (int)smp[j] > mx ? (int)smp[j] : mx;
        if (j && smp[j] == smp[j - 1]) continue;
        const unsigned long long e = (1ull << 32) | smp[j];
        uint32_t s = (smp[j] * 0x9E3779B1u) >> 21;
        for (uint32_t probes = 0; probes < SLOTS && __hip_atomic_load(&s_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= KEY_DICT_MAX; ++probes) {
            const unsigned long long old = atomicCAS(&s_set[s], 0ull, e);
            if (old == e) break;
            if (old == 0ull) {
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
    if (tid == 0) {
        const uint32_t n = s_n <= KEY_DICT_MAX ? s_n : KEY_DICT_MAX + 1;
        unit[1] = n;
        for (uint32_t i = 0; i < KEY_DICT_MAX; ++i) unit[2 + i] = i < n ? s_vals[i] : s_vals[0];
        unit[6] = (uint32_t)s_min;
        unit[7] = (uint32_t)s_max;
    }
}

// gathered: [world][AGREE_UNIT] (words as above); out = PassCfg::key_hi words (KEY_DICT_WORDS); window: the partition's fan-out, 0 = the offset
// form is not available (then only one common upper half is accepted)
__global__ void k_key_combine(const uint32_t* __restrict__ gathered, uint32_t world, uint32_t unit_words, uint32_t window, uint32_t max_d,
                              uint32_t* __restrict__ out, uint32_t* __restrict__ fail_word) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t vals[KEY_DICT_MAX + 1], n = 0;
    int mn = 0, mx = 0;
    bool any = false, many = false;
    for (uint32_t r = 0; r < world; ++r) {
        const uint32_t* u = gathered + (size_t)r * unit_words;
        const uint32_t k = u[1];
        if (k == 0) continue;   // (an empty shard)
        if (!any) { mn = (int)u[6]; mx = (int)u[7]; any = true; }
        mn = (int)u[6] < mn ? (int)u[6] : mn;
        mx = (int)u[7] > mx ? (int)u[7] : mx;
        if (k > KEY_DICT_MAX) many = true;
        for (uint32_t i = 0; i < (k < KEY_DICT_MAX ? k : KEY_DICT_MAX); ++i) {
            bool known = false;
            for (uint32_t j = 0; j < n; ++j) known = known || vals[j] == u[2 + i];
            if (!known) { if (n <= KEY_DICT_MAX) vals[n] = u[2 + i]; n = n < KEY_DICT_MAX + 1 ? n + 1 : n; }
        }
    }
    if (!any) { n = 1; vals[0] = 0; }
    const bool one = n == 1 && !many;
    const bool offs = !one && window && (uint32_t)(mx - mn) < window;
    const bool dict = !one && !offs && !many && n <= max_d;   // 2 .. 4 values far apart: the dictionary form (the order of the gathered units: the same on every rank)
    for (uint32_t i = 0; i < KEY_DICT_MAX; ++i) out[i] = dict && i < n ? vals[i] : vals[0];
    out[KEY_DICT_MAX] = dict ? n : (offs ? KEY_DICT_MAX + 1 : 1u);
    out[KEY_DICT_BASE_AT] = offs ? (uint32_t)mn - (window - 1u - (uint32_t)(mx - mn)) / 2u : 0u;
    if (!one && !offs && !dict) atomicOr(fail_word, 1u);
}

struct ShardSrcTable {
    uint32_t cap[P2R_MAX_SRC];    // [source r * K + chunk c] rows provisioned per (bucket, sub-range)
    uint32_t base[P2R_MAX_SRC];   // row offset (relative to the tuple base pointer) of that (source, chunk)'s group 0
};

// Segment table of the ring pass + the slot sizes of k_slots_setup, from the all-gathered reservation cursors:
//   hdr[r * hdr_stride + c * F1 * nsub + b * nsub + s] = rows source r put into sub-range s of bucket b in its chunk c (b owner-major:
//   my buckets are [me * per, (me + 1) * per)),  hdr[r * hdr_stride + K * F1 * nsub] = source r's "narrow attempt failed" word.
__global__ __launch_bounds__(1024) void k_shard_segments(const uint32_t* __restrict__ hdr, uint32_t hdr_stride, uint32_t F1, uint32_t world, uint32_t K,
                                                        uint32_t me, uint32_t per, uint32_t nsub, ShardSrcTable src, uint32_t F2, uint32_t sigmas,
                                                        uint32_t pad, const uint32_t* __restrict__ key_dict /*PassCfg::key_hi words, may be null*/, uint32_t dict_mul,
                                                        uint32_t* __restrict__ seg_off, uint32_t* __restrict__ seg_len,
                                                        uint32_t* __restrict__ slot_base, uint32_t* __restrict__ slot_cap, uint32_t* __restrict__ status,
                                                        unsigned long long limit_rows) {
    __shared__ uint32_t s_w[16];
    const uint32_t g = threadIdx.x, lane = g & 63, wv = g >> 6;
    const uint32_t n_src = world * K * nsub;
    uint32_t total = 0;
    if (g < per) {
        uint32_t n_g = 0;
        for (uint32_t r = 0; r < world; ++r)
            for (uint32_t c = 0; c < K; ++c) {
                const uint32_t cp = src.cap[r * K + c];
                for (uint32_t sr = 0; sr < nsub; ++sr) {
                    const uint32_t cnt = hdr[(size_t)r * hdr_stride + (size_t)c * F1 * nsub + (me * per + g) * nsub + sr];
                    const uint32_t len = cnt < cp ? cnt : cp;   // (an overflowing range raised source r's word; the result is discarded)
                    const uint32_t piece = (r * K + c) * nsub + sr;
                    seg_off[g * n_src + piece] = src.base[r * K + c] + (g * nsub + sr) * cp;
                    seg_len[g * n_src + piece] = len;
                    n_g += len;
                }
            }
        const uint32_t avg = (n_g + F2 - 1) / F2;
        const uint32_t sg = key_dict && key_dict[KEY_DICT_MAX] > 1u ? sigmas * dict_mul : sigmas;   // several upper key halves: wider slots (reserve_cap's comment)
        const uint32_t cap = (avg + sg * isqrt_u32(avg) + pad + 15u) & ~15u;
        slot_cap[g] = cap;
        total = cap * F2;
    }
    if (g < world && hdr[(size_t)g * hdr_stride + (size_t)K * F1 * nsub]) atomicOr(status, 1u);   // some rank met a wide key / an overflowing bucket
    const uint32_t inc = wave_inclusive_scan(total);
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t i = 0; i < wv; ++i) base += s_w[i];
    if (g < per) slot_base[g] = base + inc - total;
    if (g == per - 1) {
        slot_base[per] = base + inc;
        if ((unsigned long long)(base + inc) > limit_rows) atomicOr(status, 1u);   // guard: the slot layout must fit the final buffer
    }
}

static int32_t comm_all_gather(golap_comm* cm, uint64_t send, uint64_t recv, uint64_t bytes, hipStream_t s) {
    int32_t st = cm->tr.all_gather(cm->tr.user, send, recv, bytes, (void*)s);
    if (st != GOLAP_OK && cm->ctx->err.empty()) return fail(cm->ctx, st, "transport all_gather failed (%d)", st);
    return st;
}
static int32_t comm_exchange(golap_comm* cm, const std::vector<golap_p2p_op>& ops, hipStream_t s) {
    int32_t st = cm->tr.exchange(cm->tr.user, ops.data(), (uint32_t)ops.size(), (void*)s);
    if (st != GOLAP_OK && cm->ctx->err.empty()) return fail(cm->ctx, st, "transport exchange failed (%d)", st);
    return st;
}

static int32_t ensure_host_words(golap_comm* cm, uint64_t bytes) {
    if (cm->h_status_bytes >= bytes) return GOLAP_OK;
    if (cm->h_status) (void)hipHostFree(cm->h_status);
    cm->h_status = nullptr;
    cm->h_status_bytes = 0;
    GOLAP_HIP_TRY(cm->ctx, hipHostMalloc((void**)&cm->h_status, bytes, hipHostMallocDefault));
    cm->h_status_bytes = bytes;
    return GOLAP_OK;
}

// ---------------------------------------------------------------------------
// agreement: every rank contributes one word (0 = ready, else its error code) and learns everybody's -- the only way a collective
// call can fail on ONE rank without the others waiting forever in the next ncclSend/ncclRecv.  Runs on the communicator's stream
// (which waits for nothing at that point) and ends in a host synchronisation; the fast path hides it behind its first partition
// kernel.  Returns GOLAP_OK when every rank is ready, this rank's own error if it has one, GOLAP_ERR_PEER otherwise.
// ---------------------------------------------------------------------------
// Two halves, so that a caller can enqueue work that depends on the gathered payload between them (the host blocks in the second):
//   shard_agree_post   h_agree[0] -> d_agree[0], all-gather of the 16-word units, ev_agree, read-back; `payload_ready` (may be null): the
//                      communicator's stream first waits for it (the event behind the kernel that wrote d_agree[1 .. 16))
//   shard_agree_wait   host synchronisation + verdict
constexpr int32_t SHARD_SWITCH_TO_GENERAL = 1000;   // internal status, never returned across the ABI
static int32_t shard_agree_post(golap_ctx* ctx, golap_comm* cm, int32_t my_err, hipEvent_t payload_ready) {
    const uint32_t P = (uint32_t)cm->world;
    cm->h_agree[0] = (uint32_t)my_err | cm->agree_mode;
    hipError_t e = hipSuccess;
    if (payload_ready) e = hipStreamWaitEvent(cm->stream, payload_ready, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(cm->d_agree, cm->h_agree, 4, hipMemcpyHostToDevice, cm->stream);
    int32_t st = e == hipSuccess ? comm_all_gather(cm, (uint64_t)cm->d_agree, (uint64_t)(cm->d_agree + 16), golap_comm::AGREE_UNIT * 4, cm->stream) : GOLAP_ERR_HIP;
    if (st == GOLAP_OK) e = hipEventRecord(cm->ev_agree, cm->stream);
    if (st == GOLAP_OK && e == hipSuccess)
        e = hipMemcpyAsync(cm->h_agree + 16, cm->d_agree + 16, (size_t)P * golap_comm::AGREE_UNIT * 4, hipMemcpyDeviceToHost, cm->stream);
    if (st == GOLAP_OK && e != hipSuccess) st = fail(ctx, GOLAP_ERR_HIP, "agreement round failed: %s", hipGetErrorString(e));
    return st;
}
static int32_t shard_agree_wait(golap_ctx* ctx, golap_comm* cm, int32_t my_err, int32_t post_status) {
    const uint32_t P = (uint32_t)cm->world;
    std::string my_msg = my_err != GOLAP_OK ? ctx->err : std::string();
    hipError_t e = post_status == GOLAP_OK ? hipStreamSynchronize(cm->stream) : hipSuccess;
    cm->info.host_syncs += 1;
    if (my_err != GOLAP_OK) { ctx->err = my_msg; set_thread_error(my_msg.c_str()); return my_err; }
    if (post_status != GOLAP_OK) return post_status;
    if (e != hipSuccess) return fail(ctx, GOLAP_ERR_HIP, "agreement round failed: %s", hipGetErrorString(e));
    cm->agree_seen_general = cm->agree_seen_plain = false;
    for (uint32_t r = 0; r < P; ++r) {
        const uint32_t w = cm->h_agree[16 + golap_comm::AGREE_UNIT * r];
        if ((w & ~golap_comm::AGREE_GENERAL) != 0u)
            return fail(ctx, GOLAP_ERR_PEER, "golap_sharded_join: rank %u failed before the exchange (its error code %u); no rank exchanged anything", r,
                        w & ~golap_comm::AGREE_GENERAL);
        if (w & golap_comm::AGREE_GENERAL) cm->agree_seen_general = true; else cm->agree_seen_plain = true;
    }
    // a rank in the plain form among ranks in the general one: not an error, but it cannot go on as it is (golap_sharded_join starts over)
    if (cm->agree_seen_general && !cm->agree_mode) return SHARD_SWITCH_TO_GENERAL;
    return GOLAP_OK;
}
static int32_t shard_agree(golap_ctx* ctx, golap_comm* cm, int32_t my_err) {
    std::string my_msg = my_err != GOLAP_OK ? ctx->err : std::string();
    const int32_t posted = shard_agree_post(ctx, cm, my_err, nullptr);
    if (my_err != GOLAP_OK) ctx->err = my_msg;   // (a failing post must not replace this rank's own message)
    return shard_agree_wait(ctx, cm, my_err, posted);
}

// Leaves a collective call on an error path with nothing in flight: kernels and transfers still queued on either stream may
// read or write the workspace and the held slabs, which the caller releases right after.
struct ShardDrain {
    golap_ctx* ctx;
    golap_comm* cm;
    hipStream_t st;
    bool armed = true;
    bool ws_open = false;
    ShardDrain(golap_ctx* c, golap_comm* m, hipStream_t s) : ctx(c), cm(m), st(s) {}
    ~ShardDrain() {
        if (!armed) return;
        (void)hipStreamSynchronize(cm->stream);
        (void)hipStreamSynchronize(st);
        if (ws_open) ws_end(ctx, st);
    }
};

// ---------------------------------------------------------------------------
// fast path
// ---------------------------------------------------------------------------
struct ShardFastLayout {   // everything the fast path derives from the plan alone (host only, cannot fail after it has been checked)
    uint64_t send_rows[2], recv_rows[2], final_rows_[2];
    uint64_t send_base[2][golap_comm::MAX_CHUNKS];          // [side][chunk] first row of my chunk's partitioned region
    std::vector<uint64_t> recv_off[2];                       // [side][r * K + c] first row of (source r, chunk c)'s rows bound for me
    uint32_t hdr_words, hdr_stride;                          // per side: K * F1 * nsub cursors + one line (32 words); both sides gathered at once
    uint64_t need;
};

static int32_t shard_fast_layout(golap_ctx* ctx, const golap_sharded_join_params* p, const ShardPlan& sp, uint32_t me, ShardFastLayout& L) {
    const uint32_t P = sp.world, F1 = sp.F1, per = sp.per, F2 = sp.F2, nsub = sp.nsub, K = sp.K;
    for (int side = 0; side < 2; ++side) {
        uint64_t so = 0;
        for (uint32_t c = 0; c < K; ++c) {
            L.send_base[side][c] = so;
            so += (uint64_t)F1 * nsub * sp.cap[side][(size_t)me * K + c];
        }
        L.send_rows[side] = so;
        L.recv_off[side].assign((size_t)P * K, 0);
        uint64_t ro = so, n_g_max = 0;
        for (uint32_t r = 0; r < P; ++r)
            for (uint32_t c = 0; c < K; ++c) {
                const uint64_t cp = sp.cap[side][(size_t)r * K + c];
                n_g_max += (uint64_t)nsub * cp;
                if (r == me) { L.recv_off[side][(size_t)r * K + c] = L.send_base[side][c] + (uint64_t)me * per * nsub * cp; continue; }   // own buckets are read in place
                L.recv_off[side][(size_t)r * K + c] = ro;
                ro += (uint64_t)per * nsub * cp;
            }
        L.recv_rows[side] = ro - so;
        const uint64_t avg = ceil_div(n_g_max, F2);
        uint32_t sg, pd;
        provision_of("GOLAP_SLOT_SIGMAS", sg, pd);   // the same head-room k_shard_segments lays the slots out with
        sg *= DICT_SIGMA_MUL;   // (a column with several upper halves gets wider slots, as in golap_join: slot_rows_bound)
        L.final_rows_[side] = (uint64_t)per * F2 * round_up(avg + (uint64_t)sg * isqrt_host(avg) + pd + 15, 16);
        if (L.send_rows[side] + L.recv_rows[side] >= 0xFFFFFFF0ull || L.final_rows_[side] >= 0xFFFFFFF0ull)
            return fail(ctx, GOLAP_ERR_UNSUPPORTED, "sharded join: positions exceed u32");
    }
    // (+ one whole 128-byte line for the failure word: k_part1_narrow polls it every tile, and in a line of its own it does not wait
    //  behind the reservation atomics on neighbouring cursors -- with "+ 4" the probe side's word shared its line with the last four
    //  cursors of that side and the probe partition ran 8 % slower than the build partition)
    L.hdr_words = K * F1 * nsub + 32;
    L.hdr_stride = 2 * L.hdr_words;
    uint64_t small = 0;
    small += round_up((uint64_t)L.hdr_stride * 4, 256) + round_up((uint64_t)P * L.hdr_stride * 4, 256);                                   // cursors + gathered
    small += 2 * (2 * round_up((uint64_t)per * P * K * nsub * 4, 256) + 2 * round_up((uint64_t)(per + 1) * 4, 256) + 2 * round_up((uint64_t)per * F2 * 4, 256));
    small += 2 * K * 2 * round_up(2048 * 4, 256) + 4096 + round_up((uint64_t)P * 4, 256);
    L.need = small + (1 << 20);
    for (int side = 0; side < 2; ++side) L.need += round_up((L.send_rows[side] + L.recv_rows[side]) * 8, 256) + round_up(L.final_rows_[side] * 8, 256);
    if (p->join_type & 2) L.need += round_up(ceil_div(L.final_rows_[1], 32) * 4 + 8, 256);
    return GOLAP_OK;
}

// host timeline of one call (GOLAP_SHARD_TRACE=1: printed to stderr at the end of the fast path; developer aid for the gap between calls)
struct HostTrace {
    bool on;
    std::chrono::steady_clock::time_point t0;
    std::vector<std::pair<const char*, double>> pts;
    HostTrace() : on(knob("GOLAP_SHARD_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what) { if (on) pts.emplace_back(what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count()); }
    void print() {
        if (!on) return;
        std::string line = "[golap_sharded_join host us]";
        char buf[96];
        for (auto& q : pts) { snprintf(buf, sizeof buf, " %s=%.0f", q.first, q.second); line += buf; }
        fprintf(stderr, "%s\n", line.c_str());
    }
};

static int32_t sharded_join_fast(golap_ctx* ctx, golap_comm* cm, const golap_sharded_join_params* p, const ShardPlan& sp, hipStream_t st, bool* failed,
                                 int32_t arg_err) {
    HostTrace ht;
    const uint32_t P = sp.world, me = (uint32_t)cm->rank, F1 = sp.F1, per = sp.per, F2 = sp.F2, nsub = sp.nsub, K = sp.K;
    const uint64_t max_msg = shard_max_msg(cm, p->max_msg_bytes);
    const uint64_t n_local[2] = {p->n_build_local, p->n_probe_local};
    const uint64_t keys_in[2] = {p->build_key_ptr, p->probe_key_ptr};
    const uint32_t rid_base[2] = {p->build_rid_base, p->probe_rid_base};
    const bool timing = ctx->prof.enabled;
    auto stamp = [&](int which, hipStream_t s) { if (timing) (void)hipEventRecord(cm->ev_t[which], s); };   // which: 2 * stage + {0, 1}; 16 = call start
    ShardDrain drain(ctx, cm, st);
    // ---- everything that can fail on the host, BEFORE anything is exchanged: arguments (checked by the caller: arg_err), layout,
    //      kernel attributes, workspace, small arrays.  Its outcome is this rank's agreement word. ----
    ShardFastLayout L;
    struct SideArrays { uint32_t *c_off[golap_comm::MAX_CHUNKS], *c_len[golap_comm::MAX_CHUNKS], *seg_off, *seg_len, *slot_base, *slot_cap, *sub_start, *sub_len; };
    SideArrays A[2];
    uint64_t tuples[2] = {0, 0}, finals[2] = {0, 0};
    uint32_t *words = nullptr, *key_words = nullptr, *status_all = nullptr, *probe_hit = nullptr, *hdr = nullptr, *hdr_all = nullptr;
    uint32_t nc[2][golap_comm::MAX_CHUNKS], ccr[2][golap_comm::MAX_CHUNKS];
    auto prepare = [&]() -> int32_t {
        if (arg_err != GOLAP_OK) return arg_err;
        for (int side = 0; side < 2; ++side)
            if (n_local[side] != sp.rows[side][me]) return fail(ctx, GOLAP_ERR_INVALID_ARG, "local row count differs from the sharding every rank assumes");
        GOLAP_TRY(shard_fast_layout(ctx, p, sp, me, L));
        GOLAP_TRY(ensure_scatter_lds(ctx));
        GOLAP_TRY(ensure_host_words(cm, (uint64_t)P * 4));
        GOLAP_TRY(ws_begin(ctx, L.need, st));
        drain.ws_open = true;
        const char* oom = "sharded join workspace exhausted";
        for (int side = 0; side < 2; ++side) {
            tuples[side] = ws_take(ctx, (L.send_rows[side] + L.recv_rows[side]) * 8);
            finals[side] = ws_take(ctx, L.final_rows_[side] * 8);
            if (!tuples[side] || !finals[side]) return fail(ctx, GOLAP_ERR_OOM, oom);
        }
        words = (uint32_t*)ws_take(ctx, 16);            // [0] local status (any rank's failure ORed in), [1] constant 0
        key_words = (uint32_t*)ws_take(ctx, (uint64_t)KEY_DICT_WORDS * 4);   // PassCfg::key_hi, the same on every rank (k_key_combine)
        status_all = (uint32_t*)ws_take(ctx, (uint64_t)P * 4);
        hdr = (uint32_t*)ws_take(ctx, (uint64_t)L.hdr_stride * 4);          // [side][K][F1 * nsub] reservation cursors, [side] failure word
        hdr_all = (uint32_t*)ws_take(ctx, (uint64_t)P * L.hdr_stride * 4);
        if (!words || !key_words || !status_all || !hdr || !hdr_all) return fail(ctx, GOLAP_ERR_OOM, oom);
        if (p->join_type & 2) {
            probe_hit = (uint32_t*)ws_take(ctx, (ceil_div(L.final_rows_[1], 32) + 1) * 4);
            if (!probe_hit) return fail(ctx, GOLAP_ERR_OOM, "sharded join workspace exhausted (probe bitmap)");
        }
        for (int side = 0; side < 2; ++side) {
            for (uint32_t c = 0; c < K; ++c) {
                nc[side][c] = narrow_chunks(sp.rows_of(side, me, c), &ccr[side][c]);
                A[side].c_off[c] = (uint32_t*)ws_take(ctx, (uint64_t)nc[side][c] * 4);
                A[side].c_len[c] = (uint32_t*)ws_take(ctx, (uint64_t)nc[side][c] * 4);
                if (!A[side].c_off[c] || !A[side].c_len[c]) return fail(ctx, GOLAP_ERR_OOM, oom);
            }
            A[side].seg_off = (uint32_t*)ws_take(ctx, (uint64_t)per * P * K * nsub * 4);
            A[side].seg_len = (uint32_t*)ws_take(ctx, (uint64_t)per * P * K * nsub * 4);
            A[side].slot_base = (uint32_t*)ws_take(ctx, (uint64_t)(per + 1) * 4);
            A[side].slot_cap = (uint32_t*)ws_take(ctx, (uint64_t)per * 4);
            A[side].sub_start = (uint32_t*)ws_take(ctx, (uint64_t)per * F2 * 4);
            A[side].sub_len = (uint32_t*)ws_take(ctx, (uint64_t)per * F2 * 4);
            if (!A[side].seg_off || !A[side].seg_len || !A[side].slot_base || !A[side].slot_cap || !A[side].sub_start || !A[side].sub_len)
                return fail(ctx, GOLAP_ERR_OOM, oom);
        }
        return GOLAP_OK;
    };
    const int32_t prep = prepare();
    ht.mark("prepared");
    // ---- partition: `K` chunks per side on the compute stream.  The first chunk is enqueued BEFORE the agreement round so that
    //      the round's host synchronisation is hidden behind it; nothing is exchanged until every rank is known to be ready. ----
    PassCfg cfg1;
    memset(&cfg1, 0, sizeof cfg1);
    cfg1.F = F1;
    cfg1.mode = 1;
    cfg1.e1 = sp.e1;
    cfg1.owners_log2 = ilog2(P);
    cfg1.pre_hashed = p1n_variant() != 0 ? 1u : 0u;   // the tuples carry hash_int32(key) from here on (PassCfg::pre_hashed): what travels is (hash, row id)
    auto partition_chunk = [&](int side, uint32_t c) -> int32_t {
        const uint64_t lo = std::min<uint64_t>(n_local[side], (uint64_t)c * sp.chunk_rows[side][me]), n_c = sp.rows_of(side, me, c);
        uint32_t* cur = hdr + (size_t)side * L.hdr_words + (size_t)c * F1 * nsub;
        uint32_t* failw = hdr + (size_t)side * L.hdr_words + (size_t)K * F1 * nsub;
        hipLaunchKernelGGL(k_setup_uniform, dim3((uint32_t)ceil_div(nc[side][c], 256)), dim3(256), 0, st, (uint32_t)n_c, ccr[side][c], nc[side][c], A[side].c_off[c],
                           A[side].c_len[c]);
        PassCfg cfg = cfg1;
        cfg.S = nc[side][c];
        {
            KernelTimer t(ctx, side ? "partition_scatter_probe" : "partition_scatter_build", st, 16 * n_c);
            launch_part1_narrow(st, keys_in[side] + lo * 8, rid_base[side] + (uint32_t)lo, nc[side][c], A[side].c_off[c], A[side].c_len[c],
                                tuples[side] + L.send_base[side][c] * 8, cfg, ReserveArgs{cur, (uint32_t)sp.cap[side][(size_t)me * K + c], failw, failw, nullptr, nsub});
        }
        GOLAP_HIP_TRY(ctx, hipGetLastError());
        GOLAP_HIP_TRY(ctx, hipEventRecord(cm->ev_part[side][c], st));
        return GOLAP_OK;
    };
    int32_t launch_err = prep;
    const bool key_frames = prep == GOLAP_OK && p1n_variant() != 0;   // (the tuple kernels know about PassCfg::key_hi; GOLAP_P1N=0 selects round 1's)
    if (prep == GOLAP_OK) {
        stamp(16, st);
        launch_err = zero_async(ctx, st, words, 16);
        if (launch_err == GOLAP_OK) launch_err = zero_async(ctx, st, hdr, (uint64_t)L.hdr_stride * 4);
        if (launch_err == GOLAP_OK) launch_err = zero_async(ctx, st, key_words, (uint64_t)KEY_DICT_WORDS * 4);
        if (launch_err == GOLAP_OK && probe_hit) launch_err = zero_async(ctx, st, probe_hit, (ceil_div(L.final_rows_[1], 32) + 1) * 4);
        if (launch_err == GOLAP_OK && key_frames) {   // this rank's upper key halves: the payload of its agreement unit
            hipLaunchKernelGGL(k_key_summary, dim3(1), dim3(1024), 0, st, (const long long*)keys_in[0], (unsigned long long)n_local[0], (const long long*)keys_in[1],
                               (unsigned long long)n_local[1], cm->d_agree);
            if (hipEventRecord(cm->ev_keys, st) != hipSuccess) launch_err = fail(ctx, GOLAP_ERR_HIP, "event record failed");
        }
    }
    // The agreement round is posted -- its all-gather also carries every rank's summary --, the first partition chunk is enqueued behind
    // it, and only then the host waits for the verdict: the synchronisation hides behind that kernel.  Nothing is exchanged until
    // every rank is known to be ready.
    const int32_t my_word = launch_err;   // what this rank tells the others (and what it returns itself when it is an error)
    const int32_t posted = shard_agree_post(ctx, cm, my_word, my_word == GOLAP_OK && key_frames ? cm->ev_keys : nullptr);
    if (launch_err == GOLAP_OK && posted == GOLAP_OK) {
        if (key_frames) {
            GOLAP_HIP_TRY(ctx, hipStreamWaitEvent(st, cm->ev_agree, 0));
            hipLaunchKernelGGL(k_key_combine, dim3(1), dim3(64), 0, st, (const uint32_t*)(cm->d_agree + 16), P, golap_comm::AGREE_UNIT,
                               part1_offset_form_ok(keys_in[0] | keys_in[1]) ? F1 : 0u,
                               (p1n_variant() == 118 && ((keys_in[0] | keys_in[1]) & 15) == 0) ? KEY_DICT_MAX : 1u, key_words, hdr + (size_t)K * F1 * nsub);
            cfg1.key_hi = key_words;
        }
        stamp(2 * GOLAP_SHARD_ST_PART_BUILD, st);
        launch_err = partition_chunk(0, 0);
    }
    ht.mark("first_chunk_enqueued");
    {
        const int32_t agreed = shard_agree_wait(ctx, cm, my_word, posted);   // (on failure `drain` empties both streams and closes the workspace)
        if (agreed != GOLAP_OK) return agreed;
        if (launch_err != GOLAP_OK) return launch_err;   // (a launch failure behind the agreement: as fatal as a HIP error in the middle of the exchange)
    }
    ht.mark("agreed");
    for (uint32_t c = 1; c < K; ++c) GOLAP_TRY(partition_chunk(0, c));
    stamp(2 * GOLAP_SHARD_ST_PART_BUILD + 1, st);
    stamp(2 * GOLAP_SHARD_ST_PART_PROBE, st);
    for (uint32_t c = 0; c < K; ++c) GOLAP_TRY(partition_chunk(1, c));
    stamp(2 * GOLAP_SHARD_ST_PART_PROBE + 1, st);
    // ---- exchange (communicator's stream): build chunks as they are produced, then ONE all-gather of both sides' cursors and
    //      failure words (the probe side is partitioned long before the build side's last message has left), then the probe chunks ----
    // The probe side's rows travel in H group halves (with peers: 2): the ring pass and the join of the first half of this rank's
    // groups run while the second half is still on the wire -- the join of a group needs nothing from any other group, so the
    // tail behind the last message is half a ring pass + half a join instead of whole ones.  (The build side is shipped by row
    // chunks only: its ring pass hides behind the probe side's exchange anyway.)
    uint32_t H = P > 1 && per % 2 == 0 ? 2u : 1u;
    if (const char* e = knob("GOLAP_SHARD_HALVES")) { const int v = atoi(e); if ((v == 1 || v == 2) && per % (uint32_t)v == 0) H = (uint32_t)v; }
    auto exchange_chunk = [&](int side, uint32_t c, uint32_t h, uint32_t nh) -> int32_t {
        GOLAP_HIP_TRY(ctx, hipStreamWaitEvent(cm->stream, cm->ev_part[side][c], 0));
        if (c == 0 && h == 0) stamp(2 * (side ? GOLAP_SHARD_ST_XCHG_PROBE : GOLAP_SHARD_ST_XCHG_BUILD), cm->stream);
        std::vector<golap_p2p_op> ops;
        const uint64_t my_cap = sp.cap[side][(size_t)me * K + c], g0 = (uint64_t)h * (per / nh), gc = per / nh;
        for (uint32_t r = 0; r < P; ++r) {
            if (r == me) continue;
            const uint64_t r_cap = sp.cap[side][(size_t)r * K + c];
            const uint64_t sbytes = gc * nsub * my_cap * 8, rbytes = gc * nsub * r_cap * 8;
            split_message((int32_t)r, true, tuples[side] + (L.send_base[side][c] + ((uint64_t)r * per + g0) * nsub * my_cap) * 8, sbytes, max_msg, ops);
            split_message((int32_t)r, false, tuples[side] + (L.recv_off[side][(size_t)r * K + c] + g0 * nsub * r_cap) * 8, rbytes, max_msg, ops);
            cm->info.bytes_sent[side] += sbytes;
        }
        GOLAP_TRY(comm_exchange(cm, ops, cm->stream));
        return GOLAP_OK;
    };
    for (uint32_t c = 0; c < K; ++c) GOLAP_TRY(exchange_chunk(0, c, 0, 1));
    stamp(2 * GOLAP_SHARD_ST_XCHG_BUILD + 1, cm->stream);
    GOLAP_HIP_TRY(ctx, hipEventRecord(cm->ev_xchg[0], cm->stream));
    GOLAP_HIP_TRY(ctx, hipStreamWaitEvent(cm->stream, cm->ev_part[1][K - 1], 0));
    GOLAP_TRY(comm_all_gather(cm, (uint64_t)hdr, (uint64_t)hdr_all, (uint64_t)L.hdr_stride * 4, cm->stream));
    GOLAP_HIP_TRY(ctx, hipEventRecord(cm->ev_hdr, cm->stream));
    for (uint32_t h = 0; h < H; ++h) {
        for (uint32_t c = 0; c < K; ++c) GOLAP_TRY(exchange_chunk(1, c, h, H));
        GOLAP_HIP_TRY(ctx, hipEventRecord(cm->ev_half[h], cm->stream));
    }
    stamp(2 * GOLAP_SHARD_ST_XCHG_PROBE + 1, cm->stream);
    // ---- ring pass: the build side as soon as its rows (and the cursors) are in, the probe side half by half; each half joined at once ----
    SideResult res[2];
    const uint32_t n_src = P * K * nsub;
    PassCfg cfg2 = sp.local.passes[0].cfg;
    cfg2.S = 1;
    cfg2.pre_hashed = p1n_variant() != 0 ? 1u : 0u;
    cfg2.key_hi = cfg1.key_hi;
    GOLAP_HIP_TRY(ctx, hipStreamWaitEvent(st, cm->ev_hdr, 0));   // (recorded behind ev_xchg[0]: the build rows and every rank's cursors)
    for (int side = 0; side < 2; ++side) {
        cm->info.rows_received[side] = L.recv_rows[side];
        for (uint32_t c = 0; c < K; ++c) cm->info.rows_received[side] += (uint64_t)per * nsub * sp.cap[side][(size_t)me * K + c];
        ShardSrcTable tab;
        memset(&tab, 0, sizeof tab);
        for (uint32_t r = 0; r < P; ++r)
            for (uint32_t c = 0; c < K; ++c) {
                tab.cap[r * K + c] = (uint32_t)sp.cap[side][(size_t)r * K + c];
                tab.base[r * K + c] = (uint32_t)L.recv_off[side][(size_t)r * K + c];
            }
        uint32_t sigmas, pad;
        provision_of("GOLAP_SLOT_SIGMAS", sigmas, pad);   // (test knob: 0 forces overflows)
        if (side == 0) stamp(2 * GOLAP_SHARD_ST_RING_BUILD, st);
        hipLaunchKernelGGL(k_shard_segments, dim3(1), dim3(1024), 0, st, (const uint32_t*)(hdr_all + (size_t)side * L.hdr_words), L.hdr_stride, F1, P, K, me, per,
                           nsub, tab, F2, sigmas, pad, cfg1.key_hi, DICT_SIGMA_MUL, A[side].seg_off, A[side].seg_len, A[side].slot_base, A[side].slot_cap, words,
                           (unsigned long long)L.final_rows_[side]);
        res[side] = SideResult();
        res[side].keys = (const long long*)finals[side];
        res[side].alt_start = A[side].sub_start;
        res[side].alt_len = A[side].sub_len;
        res[side].flag = words + 1;
        res[side].positions = L.final_rows_[side];
    }
    auto ring = [&](int side, uint32_t g0, uint32_t gc) -> int32_t {   // groups [g0, g0 + gc) of one side
        FixedArgs fx{GroupSrc{}, A[side].slot_base + g0, A[side].slot_cap + g0, A[side].sub_start + (size_t)g0 * F2, A[side].sub_len + (size_t)g0 * F2, words, words};
        KernelTimer t(ctx, side ? "local_scatter_probe" : "local_scatter_build", st, 16 * cm->info.rows_received[side] * gc / per);
        launch_part2_ring(st, gc, tuples[side], finals[side], cfg2, fx, SegSrc{A[side].seg_off + (size_t)g0 * n_src, A[side].seg_len + (size_t)g0 * n_src, n_src});
        GOLAP_HIP_TRY(ctx, hipGetLastError());
        return GOLAP_OK;
    };
    GOLAP_TRY(ring(0, 0, per));
    stamp(2 * GOLAP_SHARD_ST_RING_BUILD + 1, st);
    for (uint32_t h = 0; h < H; ++h) {
        const uint32_t g0 = h * (per / H), gc = per / H;
        GOLAP_HIP_TRY(ctx, hipStreamWaitEvent(st, cm->ev_half[h], 0));
        if (h == 0) stamp(2 * GOLAP_SHARD_ST_RING_PROBE, st);
        GOLAP_TRY(ring(1, g0, gc));
        if (h + 1 == H) stamp(2 * GOLAP_SHARD_ST_RING_PROBE + 1, st);
        SideResult hb = res[0], hp = res[1];
        hb.alt_start += (size_t)g0 * F2; hb.alt_len += (size_t)g0 * F2;
        hp.alt_start += (size_t)g0 * F2; hp.alt_len += (size_t)g0 * F2;
        JoinPlan jp = sp.local;
        jp.n_sub = (uint64_t)gc * F2;
        if (h == 0) stamp(2 * GOLAP_SHARD_ST_JOIN, st);
        GOLAP_TRY(launch_lds_join(ctx, st, jp, hb, hp, cm->info.rows_received[0] / H, cm->info.rows_received[1] / H, p->out_pairs_ptr, p->out_count_ptr,
                                  p->max_pairs, p->join_type, true, nullptr, words, probe_hit, cfg1.pre_hashed ? (const uint32_t*)cfg1.key_hi : nullptr));
    }
    stamp(2 * GOLAP_SHARD_ST_JOIN + 1, st);
    // ---- did the narrow attempt hold on EVERY rank?  (all ranks must take the same path: the words are gathered) ----
    GOLAP_HIP_TRY(ctx, hipEventRecord(cm->ev_misc, st));
    GOLAP_HIP_TRY(ctx, hipStreamWaitEvent(cm->stream, cm->ev_misc, 0));
    stamp(2 * GOLAP_SHARD_ST_STATUS, cm->stream);
    GOLAP_TRY(comm_all_gather(cm, (uint64_t)words, (uint64_t)status_all, 4, cm->stream));
    GOLAP_HIP_TRY(ctx, hipMemcpyAsync(cm->h_status, status_all, (size_t)P * 4, hipMemcpyDeviceToHost, cm->stream));
    stamp(2 * GOLAP_SHARD_ST_STATUS + 1, cm->stream);
    GOLAP_HIP_TRY(ctx, hipEventRecord(cm->ev_misc, cm->stream));
    drain.ws_open = false;
    ws_end(ctx, st);
    GOLAP_HIP_TRY(ctx, hipStreamWaitEvent(st, cm->ev_misc, 0));   // the caller's stream also orders behind the communicator's
    ht.mark("all_enqueued");
    GOLAP_HIP_TRY(ctx, hipEventSynchronize(cm->ev_misc));         // the host synchronisation at the end of the fast path
    ht.mark("synchronised");
    drain.armed = false;
    cm->info.host_syncs += 1;
    cm->info.chunks = K;
    cm->info.sub_ranges = nsub;
    cm->info.halves = H;
    if (timing) {
        for (int sg = 0; sg < GOLAP_SHARD_STAGES; ++sg)
            for (int e = 0; e < 2; ++e) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, cm->ev_t[16], cm->ev_t[2 * sg + e]) != hipSuccess) { (void)hipGetLastError(); ms = 0.f; }
                cm->info.stage_ms[sg][e] = ms;
            }
        cm->info.total_ms = cm->info.stage_ms[GOLAP_SHARD_ST_STATUS][1];
    }
    uint32_t any = 0;
    for (uint32_t r = 0; r < P; ++r) any |= cm->h_status[r];
    *failed = any != 0;
    ht.mark("done");
    ht.print();
    return GOLAP_OK;
}

// ---------------------------------------------------------------------------
// exact path: histogram partition (64-bit rows), sizes read back, variable-size exchange, segment-fed local passes
// ---------------------------------------------------------------------------
struct SlabHold {   // slabs that must outlive the workspace (the workspace is re-begun by the partition and by the local passes)
    golap_ctx* ctx;
    std::vector<std::pair<uint64_t, uint32_t>> slabs;
    explicit SlabHold(golap_ctx* c) : ctx(c) {}
    uint64_t take(uint64_t bytes) {
        uint64_t p = 0;
        uint32_t cls = 0;
        if (ctx->slabs.allocate(ctx, std::max<uint64_t>(bytes, 256), &p, &cls) != GOLAP_OK) return 0;
        slabs.push_back({p, cls});
        return p;
    }
    ~SlabHold() {
        for (auto& s : slabs) (void)ctx->slabs.release(ctx, s.first, s.second);
    }
};

// ---------------------------------------------------------------------------
// key tuples that do not fit one exact int64 (composites beyond 62 bits; a full-range column with NULLs): the exact path joins
// on a 64-bit SURROGATE of the tuple (k_surrogate: the single-GPU join's; a row with a NULL gets a key of its own) and ships the
// real key columns and a "no NULL in this row" byte along with the rows -- same partition order, same messages, one more array
// each.  The receiving rank joins on positions instead of row ids, checks every pair against the shipped columns (k_gj_verify /
// k_gj_repair_*: two tuples share a surrogate with probability 2^-64 per candidate pair -- rare, not never), and only then turns
// positions into the global row ids.
// ---------------------------------------------------------------------------
struct ShardWideKeys {
    KeyCols cols[2];        // [side] this rank's original key columns and validity arrays
};
__global__ __launch_bounds__(256) void k_gather8_by_rid(const long long* __restrict__ col, const uint32_t* __restrict__ rids, uint32_t rid_base, unsigned long long n,
                                                        long long* __restrict__ out) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) out[i] = col[rids[i] - rid_base];
}
__global__ __launch_bounds__(256) void k_rowvalid_by_rid(KeyCols k, const uint32_t* __restrict__ rids, uint32_t rid_base, unsigned long long n, uint8_t* __restrict__ out) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        const uint32_t r = rids[i] - rid_base;
        bool ok = true;
        for (uint32_t c = 0; c < k.n_cols; ++c) ok = ok && !(k.valid[c] && !k.valid[c][r]);
        out[i] = ok ? 1 : 0;
    }
}
__global__ __launch_bounds__(256) void k_iota_u32(uint32_t* __restrict__ out, unsigned long long n) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) out[i] = (uint32_t)i;
}
// (position in the received build rows, position in the received probe rows) -> global row ids; the NULL id of an outer join stays
__global__ __launch_bounds__(256) void k_translate_pairs(uint2* __restrict__ pairs, const unsigned long long* __restrict__ count, unsigned long long max_pairs,
                                                         const uint32_t* __restrict__ rid_b, const uint32_t* __restrict__ rid_p) {
    const unsigned long long n = *count < max_pairs ? *count : max_pairs;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        uint2 v = pairs[i];
        if (v.x != JN_NULL) v.x = rid_b[v.x];
        if (v.y != JN_NULL) v.y = rid_p[v.y];
        pairs[i] = v;
    }
}

static int32_t sharded_join_exact(golap_ctx* ctx, golap_comm* cm, const golap_sharded_join_params* p, const ShardPlan& sp, hipStream_t st, int32_t arg_err,
                                  const ShardWideKeys* wide = nullptr) {
    const uint32_t P = sp.world, me = (uint32_t)cm->rank, e1 = sp.e1, F1 = sp.F1, per = sp.per;
    const uint64_t max_msg = shard_max_msg(cm, p->max_msg_bytes);
    const uint64_t n_local[2] = {p->n_build_local, p->n_probe_local};
    const uint64_t keys_in[2] = {p->build_key_ptr, p->probe_key_ptr};
    const uint32_t rid_base[2] = {p->build_rid_base, p->probe_rid_base};
    SlabHold hold(ctx);                  // (declared before `drain`: on an error path the streams are emptied BEFORE the slabs go back to the pool)
    ShardDrain drain(ctx, cm, st);
    uint64_t recv_keys[2] = {0, 0}, recv_rids[2] = {0, 0}, n_recv[2] = {0, 0};
    std::vector<uint64_t> seg_off[2], seg_len[2];
    const uint64_t off_words = (uint64_t)F1 + 1;
    // what depends on the arguments alone is allocated first; every rank says so before anything is exchanged (shard_agree).  The
    // receive buffers depend on the data: they get their own agreement round per side, next to the read-back that sizes them.
    uint64_t pk_[2] = {0, 0}, pr_[2] = {0, 0}, offs_[2] = {0, 0}, offs_all_[2] = {0, 0};
    // wide key tuples (ShardWideKeys): the key columns and the row-valid byte in partition order (sent), and as received
    const uint32_t WK = wide ? wide->cols[0].n_cols : 0;
    uint64_t wsend[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, wsend_v[2] = {0, 0}, wrecv[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, wrecv_v[2] = {0, 0}, wiota[2] = {0, 0}, wctl = 0;
    auto prepare = [&]() -> int32_t {
        if (arg_err != GOLAP_OK) return arg_err;
        GOLAP_TRY(ensure_host_words(cm, 2 * (uint64_t)P * off_words * 4));
        for (int side = 0; side < 2; ++side) {
            pk_[side] = hold.take(std::max<uint64_t>(1, n_local[side]) * 8);
            pr_[side] = hold.take(std::max<uint64_t>(1, n_local[side]) * 4);
            offs_[side] = hold.take(off_words * 4);
            offs_all_[side] = hold.take((uint64_t)P * off_words * 4);
            if (!pk_[side] || !pr_[side] || !offs_[side] || !offs_all_[side]) return fail(ctx, GOLAP_ERR_OOM, "sharded join: partition buffers");
            for (uint32_t c = 0; c < WK; ++c)
                if (!(wsend[side][c] = hold.take(std::max<uint64_t>(1, n_local[side]) * 8))) return fail(ctx, GOLAP_ERR_OOM, "sharded join: key columns in partition order");
            if (WK && !(wsend_v[side] = hold.take(std::max<uint64_t>(1, n_local[side])))) return fail(ctx, GOLAP_ERR_OOM, "sharded join: key columns in partition order");
        }
        if (WK && !(wctl = hold.take(sizeof(GjCtl)))) return fail(ctx, GOLAP_ERR_OOM, "sharded join: verification block");
        return GOLAP_OK;
    };
    GOLAP_TRY(shard_agree(ctx, cm, prepare()));
    for (int side = 0; side < 2; ++side) {
        // partition this rank's rows (exact layout: contiguous buckets, owner-major)
        const uint64_t pk = pk_[side], pr = pr_[side], offs = offs_[side], offs_all = offs_all_[side];
        golap_partition_params pp;
        memset(&pp, 0, sizeof pp);
        pp.key_ptr = keys_in[side];
        pp.n_rows = n_local[side];
        pp.rid_base = rid_base[side];
        pp.n_owners = P;
        pp.extra_bits = e1;
        pp.out_key_ptr = pk;
        pp.out_row_id_ptr = pr;
        pp.out_offsets_ptr = offs;
        GOLAP_TRY(radix_partition_impl(ctx, &pp, st));
        if (WK && n_local[side]) {   // the real key columns follow their rows: gathered by the partitioned row ids
            const uint32_t g = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)ctx->prop.multiProcessorCount * 8, ceil_div(n_local[side], 1024)));
            for (uint32_t c = 0; c < WK; ++c)
                hipLaunchKernelGGL(k_gather8_by_rid, dim3(g), dim3(256), 0, st, wide->cols[side].col[c], (const uint32_t*)pr, rid_base[side], (unsigned long long)n_local[side],
                                   (long long*)wsend[side][c]);
            hipLaunchKernelGGL(k_rowvalid_by_rid, dim3(g), dim3(256), 0, st, wide->cols[side], (const uint32_t*)pr, rid_base[side], (unsigned long long)n_local[side],
                               (uint8_t*)wsend_v[side]);
            GOLAP_HIP_TRY(ctx, hipGetLastError());
        }
        // bucket sizes of every rank: device all-gather, then the one read-back this side needs
        GOLAP_HIP_TRY(ctx, hipEventRecord(cm->ev_part[side][0], st));
        GOLAP_HIP_TRY(ctx, hipStreamWaitEvent(cm->stream, cm->ev_part[side][0], 0));
        GOLAP_TRY(comm_all_gather(cm, offs, offs_all, off_words * 4, cm->stream));
        uint32_t* h = cm->h_status + (uint64_t)side * P * off_words;
        GOLAP_HIP_TRY(ctx, hipMemcpyAsync(h, (const void*)offs_all, (size_t)P * off_words * 4, hipMemcpyDeviceToHost, cm->stream));
        GOLAP_HIP_TRY(ctx, hipStreamSynchronize(cm->stream));
        cm->info.host_syncs += 1;
        // what I receive: from every source r its buckets [me * per, (me + 1) * per)
        std::vector<uint64_t> from(P), base(P);
        for (uint32_t r = 0; r < P; ++r) {
            const uint32_t* o = h + (uint64_t)r * off_words;
            from[r] = (uint64_t)o[(me + 1) * per] - o[me * per];
            base[r] = n_recv[side];
            n_recv[side] += from[r];
        }
        recv_keys[side] = hold.take(std::max<uint64_t>(1, n_recv[side]) * 8);
        recv_rids[side] = hold.take(std::max<uint64_t>(1, n_recv[side]) * 4);
        bool wide_ok = true;
        for (uint32_t c = 0; c < WK; ++c) wide_ok = wide_ok && (wrecv[side][c] = hold.take(std::max<uint64_t>(1, n_recv[side]) * 8)) != 0;
        if (WK) wide_ok = wide_ok && (wrecv_v[side] = hold.take(std::max<uint64_t>(1, n_recv[side]))) != 0 && (wiota[side] = hold.take(std::max<uint64_t>(1, n_recv[side]) * 4)) != 0;
        GOLAP_TRY(shard_agree(ctx, cm, recv_keys[side] && recv_rids[side] && wide_ok ? GOLAP_OK : fail(ctx, GOLAP_ERR_OOM, "sharded join: receive buffers")));
        seg_off[side].assign((size_t)per * P, 0);
        seg_len[side].assign((size_t)per * P, 0);
        for (uint32_t r = 0; r < P; ++r) {
            const uint32_t* o = h + (uint64_t)r * off_words;
            for (uint32_t g = 0; g < per; ++g) {
                seg_off[side][(size_t)g * P + r] = base[r] + ((uint64_t)o[me * per + g] - o[me * per]);
                seg_len[side][(size_t)g * P + r] = (uint64_t)o[me * per + g + 1] - o[me * per + g];
            }
        }
        const uint32_t* mine = h + (uint64_t)me * off_words;
        std::vector<golap_p2p_op> ops;
        for (uint32_t r = 0; r < P; ++r) {
            const uint64_t s0 = mine[r * per], s1 = mine[(r + 1) * per];   // my rows bound for r: one contiguous slice, nothing is packed
            if (r == me) {
                if (s1 > s0) {   // own partitions never touch the transport
                    GOLAP_HIP_TRY(ctx, hipMemcpyAsync((void*)(recv_keys[side] + base[r] * 8), (const void*)(pk + s0 * 8), (s1 - s0) * 8, hipMemcpyDeviceToDevice, cm->stream));
                    GOLAP_HIP_TRY(ctx, hipMemcpyAsync((void*)(recv_rids[side] + base[r] * 4), (const void*)(pr + s0 * 4), (s1 - s0) * 4, hipMemcpyDeviceToDevice, cm->stream));
                    for (uint32_t c = 0; c < WK; ++c)
                        GOLAP_HIP_TRY(ctx, hipMemcpyAsync((void*)(wrecv[side][c] + base[r] * 8), (const void*)(wsend[side][c] + s0 * 8), (s1 - s0) * 8, hipMemcpyDeviceToDevice, cm->stream));
                    if (WK) GOLAP_HIP_TRY(ctx, hipMemcpyAsync((void*)(wrecv_v[side] + base[r]), (const void*)(wsend_v[side] + s0), s1 - s0, hipMemcpyDeviceToDevice, cm->stream));
                }
                continue;
            }
            split_message((int32_t)r, true, pk + s0 * 8, (s1 - s0) * 8, max_msg, ops);
            split_message((int32_t)r, false, recv_keys[side] + base[r] * 8, from[r] * 8, max_msg, ops);
            split_message((int32_t)r, true, pr + s0 * 4, (s1 - s0) * 4, max_msg, ops);
            split_message((int32_t)r, false, recv_rids[side] + base[r] * 4, from[r] * 4, max_msg, ops);
            for (uint32_t c = 0; c < WK; ++c) {
                split_message((int32_t)r, true, wsend[side][c] + s0 * 8, (s1 - s0) * 8, max_msg, ops);
                split_message((int32_t)r, false, wrecv[side][c] + base[r] * 8, from[r] * 8, max_msg, ops);
            }
            if (WK) {
                split_message((int32_t)r, true, wsend_v[side] + s0, s1 - s0, max_msg, ops);
                split_message((int32_t)r, false, wrecv_v[side] + base[r], from[r], max_msg, ops);
            }
            cm->info.bytes_sent[side] += (s1 - s0) * (12 + 8ull * WK + (WK ? 1 : 0));
        }
        GOLAP_TRY(comm_exchange(cm, ops, cm->stream));   // in flight while the other side is partitioned
        GOLAP_HIP_TRY(ctx, hipEventRecord(cm->ev_xchg[side], cm->stream));
        cm->info.rows_received[side] = n_recv[side];
    }
    golap_join_partitioned_plan pl;
    memset(&pl, 0, sizeof pl);
    pl.n_groups = per;
    pl.n_sources = P;
    pl.pre_bits = 8 + e1;
    pl.total_build_rows_hint = std::max<uint64_t>(1, p->n_build_total / P);
    pl.build_rows = n_recv[0];
    pl.probe_rows = n_recv[1];
    GOLAP_TRY(part_begin(ctx, &pl, st));
    int32_t rc = GOLAP_OK;
    for (int side = 0; side < 2 && rc == GOLAP_OK; ++side) {
        if (hipStreamWaitEvent(st, cm->ev_xchg[side], 0) != hipSuccess) { rc = fail(ctx, GOLAP_ERR_HIP, "hipStreamWaitEvent failed"); break; }
        golap_partitioned_side sd;
        sd.key_ptr = recv_keys[side];
        sd.row_id_ptr = recv_rids[side];
        if (WK) {   // wide key tuples: the join emits POSITIONS in the received rows (checked against the shipped columns below)
            if (n_recv[side]) hipLaunchKernelGGL(k_iota_u32, dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(2048, ceil_div(n_recv[side], 1024)))), dim3(256), 0, st,
                                                 (uint32_t*)wiota[side], (unsigned long long)n_recv[side]);
            sd.row_id_ptr = wiota[side];
        }
        sd.n_rows = n_recv[side];
        sd.seg_off = seg_off[side].data();
        sd.seg_len = seg_len[side].data();
        rc = part_side(ctx, side, &sd, st);
    }
    if (rc == GOLAP_OK) rc = part_finish(ctx, p->out_pairs_ptr, p->out_count_ptr, p->max_pairs, st, p->join_type);
    if (rc != GOLAP_OK) { part_abort(ctx); return rc; }   // (`drain` empties both streams)
    if (WK) {
        KeyCols B{}, Pc{};
        B.n_cols = Pc.n_cols = WK;
        for (uint32_t c = 0; c < WK; ++c) { B.col[c] = (const long long*)wrecv[0][c]; Pc.col[c] = (const long long*)wrecv[1][c]; }
        B.valid[0] = (const uint8_t*)wrecv_v[0];     // one byte per ROW ("no key column of this row is NULL"): a row with a NULL matches nothing
        Pc.valid[0] = (const uint8_t*)wrecv_v[1];
        GOLAP_TRY(zero_async(ctx, st, (void*)wctl, sizeof(GjCtl)));
        const uint32_t g = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)ctx->prop.multiProcessorCount * 8, ceil_div(std::max<uint64_t>(p->max_pairs, 1), 1024)));
        if (n_recv[0] && n_recv[1]) {
            hipLaunchKernelGGL(k_gj_verify, dim3(g), dim3(256), 0, st, (const uint2*)p->out_pairs_ptr, (const unsigned long long*)p->out_count_ptr, (unsigned long long)p->max_pairs, B, Pc,
                               0u, 0u, (GjCtl*)wctl);
            hipLaunchKernelGGL(k_gj_repair_scan, dim3(g), dim3(256), 0, st, (const uint2*)p->out_pairs_ptr, (const unsigned long long*)p->out_count_ptr, (unsigned long long)p->max_pairs,
                               B, Pc, 0u, 0u, (GjCtl*)wctl);
            hipLaunchKernelGGL(k_gj_repair_fix, dim3(1), dim3(64), 0, st, (uint2*)p->out_pairs_ptr, (unsigned long long*)p->out_count_ptr, (unsigned long long)p->max_pairs, p->join_type,
                               (GjCtl*)wctl);
        }
        hipLaunchKernelGGL(k_translate_pairs, dim3(g), dim3(256), 0, st, (uint2*)p->out_pairs_ptr, (const unsigned long long*)p->out_count_ptr, (unsigned long long)p->max_pairs,
                           (const uint32_t*)recv_rids[0], (const uint32_t*)recv_rids[1]);
        GOLAP_HIP_TRY(ctx, hipGetLastError());
    }
    // the slabs above go back to the pool at scope exit: every kernel that reads them must have finished
    GOLAP_HIP_TRY(ctx, hipStreamSynchronize(st));
    drain.armed = false;
    cm->info.host_syncs += 1;
    return GOLAP_OK;
}

}  // namespace golap

extern "C" {

void golap_shard_range(uint64_t n_total, uint32_t rank, uint32_t world, uint64_t* lo, uint64_t* hi) {
    if (world == 0) world = 1;
    const uint64_t per = n_total / world;
    if (lo) *lo = (uint64_t)rank * per;
    if (hi) *hi = rank + 1 == world ? n_total : ((uint64_t)rank + 1) * per;
}

int32_t golap_comm_unique_id(uint8_t out_id[GOLAP_COMM_ID_BYTES]) {
    if (!out_id) return fail(nullptr, GOLAP_ERR_INVALID_ARG, "null argument");
    RcclApi* api = rccl_api(nullptr);
    if (!api) return GOLAP_ERR_UNSUPPORTED;
    ncclUniqueId id;
    GOLAP_NCCL_TRY(nullptr, api, api->GetUniqueId(&id));
    static_assert(sizeof(id) == GOLAP_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(out_id, &id, sizeof id);
    return GOLAP_OK;
}

static int32_t comm_common(golap_ctx* ctx, int32_t world, int32_t rank, golap_comm** out, golap_comm** made) {
    if (!ctx || !out) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (world < 1 || world > P2R_MAX_SRC || (world & (world - 1)) || 256 % world) return fail(ctx, GOLAP_ERR_INVALID_ARG, "world size must be a power of two <= 64");
    if (rank < 0 || rank >= world) return fail(ctx, GOLAP_ERR_INVALID_ARG, "rank %d out of range", rank);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    golap_comm* cm = new golap_comm();
    cm->ctx = ctx;
    cm->world = world;
    cm->rank = rank;
    hipError_t e = hipStreamCreateWithFlags(&cm->stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        for (int c = 0; c < golap_comm::MAX_CHUNKS && e == hipSuccess; ++c) e = hipEventCreateWithFlags(&cm->ev_part[i][c], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&cm->ev_xchg[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&cm->ev_half[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&cm->ev_misc, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&cm->ev_hdr, hipEventDisableTiming);
    for (int i = 0; i < 2 * GOLAP_SHARD_STAGES + 1 && e == hipSuccess; ++i) e = hipEventCreate(&cm->ev_t[i]);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&cm->ev_keys, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&cm->ev_agree, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&cm->d_agree, golap_comm::AGREE_WORDS * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(cm->d_agree, 0, golap_comm::AGREE_WORDS * sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc((void**)&cm->h_agree, golap_comm::AGREE_WORDS * sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) { golap_comm_destroy(cm); return fail(ctx, GOLAP_ERR_HIP, "communicator setup failed: %s", hipGetErrorString(e)); }
    *made = cm;
    return GOLAP_OK;
}

int32_t golap_comm_create_rccl(golap_ctx* ctx, const uint8_t id[GOLAP_COMM_ID_BYTES], int32_t world, int32_t rank, golap_comm** out) {
    if (!id) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    RcclApi* api = rccl_api(ctx);
    if (!api) return GOLAP_ERR_UNSUPPORTED;
    golap_comm* cm = nullptr;
    GOLAP_TRY(comm_common(ctx, world, rank, out, &cm));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t r = api->CommInitRank(&cm->nccl, world, uid, rank);
    if (r != ncclSuccess) { golap_comm_destroy(cm); return fail(ctx, GOLAP_ERR_HIP, "ncclCommInitRank failed: %s", api->GetErrorString(r)); }
    cm->builtin = true;
    cm->tr.user = cm;
    cm->tr.all_gather = rccl_all_gather;
    cm->tr.exchange = rccl_exchange;
    *out = cm;
    return GOLAP_OK;
}

int32_t golap_comm_create_custom(golap_ctx* ctx, const golap_transport* transport, int32_t world, int32_t rank, golap_comm** out) {
    if (!transport || !transport->all_gather || !transport->exchange) return fail(ctx, GOLAP_ERR_INVALID_ARG, "transport callbacks missing");
    golap_comm* cm = nullptr;
    GOLAP_TRY(comm_common(ctx, world, rank, out, &cm));
    cm->tr = *transport;
    *out = cm;
    return GOLAP_OK;
}

void golap_comm_destroy(golap_comm* cm) {
    if (!cm) return;
    if (cm->ctx) (void)hipSetDevice(cm->ctx->device);
    if (cm->stream) (void)hipStreamSynchronize(cm->stream);
    if (cm->nccl) {
        RcclApi* api = rccl_api(cm->ctx);
        if (api) (void)api->CommDestroy(cm->nccl);
    }
    for (int i = 0; i < 2; ++i) {
        for (int c = 0; c < golap_comm::MAX_CHUNKS; ++c)
            if (cm->ev_part[i][c]) (void)hipEventDestroy(cm->ev_part[i][c]);
        if (cm->ev_xchg[i]) (void)hipEventDestroy(cm->ev_xchg[i]);
        if (cm->ev_half[i]) (void)hipEventDestroy(cm->ev_half[i]);
    }
    if (cm->ev_misc) (void)hipEventDestroy(cm->ev_misc);
    if (cm->ev_hdr) (void)hipEventDestroy(cm->ev_hdr);
    if (cm->ev_keys) (void)hipEventDestroy(cm->ev_keys);
    if (cm->ev_agree) (void)hipEventDestroy(cm->ev_agree);
    for (int i = 0; i < 2 * GOLAP_SHARD_STAGES + 1; ++i)
        if (cm->ev_t[i]) (void)hipEventDestroy(cm->ev_t[i]);
    if (cm->d_agree) (void)hipFree(cm->d_agree);
    if (cm->h_agree) (void)hipHostFree(cm->h_agree);
    if (cm->h_status) (void)hipHostFree(cm->h_status);
    if (cm->stream) (void)hipStreamDestroy(cm->stream);
    delete cm;
}

int32_t golap_sharded_plan_get(uint32_t world, uint64_t n_build_total, uint64_t n_probe_total, const uint64_t* build_rows_per_rank,
                               const uint64_t* probe_rows_per_rank, uint64_t max_msg_bytes, golap_sharded_plan* out) {
    if (!out) return fail(nullptr, GOLAP_ERR_INVALID_ARG, "null argument");
    if (world < 1 || world > (uint32_t)P2R_MAX_SRC || (world & (world - 1))) return fail(nullptr, GOLAP_ERR_INVALID_ARG, "world size must be a power of two <= 64");
    ShardPlan sp;
    make_shard_plan(world, n_build_total, n_probe_total, build_rows_per_rank, probe_rows_per_rank, sp);
    memset(out, 0, sizeof *out);
    out->eligible = sp.eligible ? 1 : 0;
    out->extra_bits = sp.e1;
    out->n_buckets = sp.F1;
    out->buckets_per_rank = sp.per;
    out->local_bits = sp.local_bits;
    out->sub_ranges = sp.nsub;
    out->chunks = sp.K;
    const uint64_t mm = max_msg_bytes ? max_msg_bytes : SHARD_DEFAULT_MAX_MSG;
    for (int side = 0; side < 2; ++side)
        for (uint32_t r = 0; r < world; ++r) {
            out->cap_rows[side][r] = sp.cap[side][(size_t)r * sp.K];
            out->cap_rows_last[side][r] = sp.cap[side][(size_t)r * sp.K + sp.K - 1];
            for (uint32_t c = 0; c < sp.K; ++c) {
                const uint64_t b = (uint64_t)sp.per * sp.nsub * sp.cap[side][(size_t)r * sp.K + c] * 8;
                out->msg_bytes[side][r] += b;
                out->msg_pieces[side][r] += (uint32_t)ceil_div(b, mm);
            }
        }
    return GOLAP_OK;
}

// Diagnostic: one UNSPLIT message of `bytes` bytes from this rank to itself through the transport (ncclSend + ncclRecv in one
// group), checked word by word on the device.  Pins where large single messages stop being delivered correctly on a stack
// (tests/test_sharded_gpu.py::test_rccl_message_size_boundary); the join itself never sends more than max_msg_bytes at once.
__global__ void k_selftest_fill(uint32_t* __restrict__ a, unsigned long long n) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        a[i] = (uint32_t)(i * 2654435761ull) ^ (uint32_t)(i >> 32);
}
__global__ void k_selftest_check(const uint32_t* __restrict__ a, unsigned long long n, unsigned long long* __restrict__ bad, unsigned long long* __restrict__ first_bad) {
    unsigned long long local = 0, first = ~0ull;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        if (a[i] != ((uint32_t)(i * 2654435761ull) ^ (uint32_t)(i >> 32))) { ++local; if (i < first) first = i; }
    if (local) { atomicAdd(bad, local); atomicMin(first_bad, first); }
}

int32_t golap_comm_selftest(golap_ctx* ctx, golap_comm* cm, uint64_t bytes, uint64_t* out_bad_words, uint64_t* out_first_bad_word) {
    if (!ctx || !cm || !out_bad_words) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    if (bytes == 0 || (bytes & 3)) return fail(ctx, GOLAP_ERR_INVALID_ARG, "bytes must be a positive multiple of 4");
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    SlabHold hold(ctx);
    const uint64_t src = hold.take(bytes), dst = hold.take(bytes), res = hold.take(256);
    if (!src || !dst || !res) return GOLAP_ERR_OOM;
    const unsigned long long n = bytes / 4;
    hipStream_t s = cm->stream;
    hipLaunchKernelGGL(k_selftest_fill, dim3(2048), dim3(256), 0, s, (uint32_t*)src, n);
    GOLAP_TRY(zero_async(ctx, s, (void*)dst, bytes));
    GOLAP_HIP_TRY(ctx, hipMemsetAsync((void*)res, 0, 8, s));
    GOLAP_HIP_TRY(ctx, hipMemsetAsync((void*)(res + 8), 0xFF, 8, s));
    std::vector<golap_p2p_op> ops(2);
    ops[0].peer = cm->rank; ops[0].is_send = 1; ops[0].ptr = src; ops[0].bytes = bytes;
    ops[1].peer = cm->rank; ops[1].is_send = 0; ops[1].ptr = dst; ops[1].bytes = bytes;
    GOLAP_TRY(comm_exchange(cm, ops, s));
    hipLaunchKernelGGL(k_selftest_check, dim3(2048), dim3(256), 0, s, (const uint32_t*)dst, n, (unsigned long long*)res, (unsigned long long*)(res + 8));
    uint64_t h[2] = {0, 0};
    GOLAP_HIP_TRY(ctx, hipMemcpyAsync(h, (const void*)res, 16, hipMemcpyDeviceToHost, s));
    GOLAP_HIP_TRY(ctx, hipStreamSynchronize(s));
    *out_bad_words = h[0];
    if (out_first_bad_word) *out_first_bad_word = h[1];
    return GOLAP_OK;
}

int32_t golap_sharded_join_info_get(golap_comm* cm, golap_sharded_join_info* out) {
    if (!cm || !out) return fail(cm ? cm->ctx : nullptr, GOLAP_ERR_INVALID_ARG, "null argument");
    *out = cm->info;
    return GOLAP_OK;
}

}  // extern "C"

namespace golap {

// ---------------------------------------------------------------------------
// composite keys and NULLs on the sharded join: the ranks agree on every key column's value range and join on the exact
// mixed-radix number of the digits (value - min); a row with a NULL in any key column gets a key no other row has
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_key_minmax(const long long* __restrict__ col, const uint8_t* __restrict__ valid, unsigned long long n,
                                                    long long* __restrict__ out /*[0] min, [1] max (of the valid rows)*/) {
    long long mn = INT64_MAX, mx = INT64_MIN;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        if (!valid || valid[i]) { const long long v = col[i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    for (int d = 32; d >= 1; d >>= 1) {
        const long long a = __shfl_xor(mn, d, 64), b = __shfl_xor(mx, d, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if ((threadIdx.x & 63) == 0) {
        if (mn != INT64_MAX) __hip_atomic_fetch_min(&out[0], mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mx != INT64_MIN) __hip_atomic_fetch_max(&out[1], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
struct PackCols {
    const long long* col[4];
    const uint8_t* valid[4];
    long long mn[4];
    unsigned long long weight[4];
    uint32_t n;
};
__global__ __launch_bounds__(256) void k_pack_keys(PackCols pc, unsigned long long n, unsigned long long null_base, long long* __restrict__ out) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long key = 0;
        bool null = false;
        for (uint32_t c = 0; c < pc.n; ++c) {
            null = null || (pc.valid[c] && !pc.valid[c][i]);
            key += (unsigned long long)(pc.col[c][i] - pc.mn[c]) * pc.weight[c];
        }
        out[i] = (long long)(null ? null_base + i : key);   // NULL rows: 2^62 + side * 2^61 + global row id -- above every packed key, unique
    }
}

// -> *packed_b / *packed_p: device int64 columns (slabs in `hold`) that replace the key columns; collective (one all-gather + host read)
// `switched`: this rank came here from the plain form after its first agreement round showed general peers (that round is the
// one the general ranks ran below); both kinds then meet in a second round, which carries the late-comers' prepare() outcome.
static int32_t shard_pack_keys(golap_ctx* ctx, golap_comm* cm, const golap_sharded_join_params* p, hipStream_t st, SlabHold& hold, uint64_t* packed_b,
                               uint64_t* packed_p, bool switched, ShardWideKeys* wide_out, bool* is_wide) {
    const uint32_t P = (uint32_t)cm->world, K = std::max<uint32_t>(1, p->n_key_cols);
    const uint64_t n[2] = {p->n_build_local, p->n_probe_local};
    uint64_t cols[2][4], valids[2][4];
    for (uint32_t c = 0; c < K; ++c) {
        cols[0][c] = c ? p->build_key_extra[c - 1] : p->build_key_ptr;
        cols[1][c] = c ? p->probe_key_extra[c - 1] : p->probe_key_ptr;
        valids[0][c] = p->build_validity[c];
        valids[1][c] = p->probe_validity[c];
    }
    uint64_t d_mm = 0, d_all = 0;
    auto prepare = [&]() -> int32_t {
        if (K > 4) return fail(ctx, GOLAP_ERR_INVALID_ARG, "at most 4 key columns");
        for (int side = 0; side < 2; ++side)
            for (uint32_t c = 0; c < K; ++c)
                if (n[side] && !cols[side][c]) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null key column %u", c);
        *packed_b = hold.take(std::max<uint64_t>(1, n[0]) * 8);
        *packed_p = hold.take(std::max<uint64_t>(1, n[1]) * 8);
        d_mm = hold.take(256);
        d_all = hold.take((uint64_t)P * 64 + 256);
        if (!*packed_b || !*packed_p || !d_mm || !d_all) return fail(ctx, GOLAP_ERR_OOM, "sharded join: packed key columns");
        GOLAP_TRY(ensure_host_words(cm, (uint64_t)P * 64 + 64));
        return GOLAP_OK;
    };
    if (switched) {
        GOLAP_TRY(shard_agree(ctx, cm, prepare()));
    } else {
        GOLAP_TRY(shard_agree(ctx, cm, prepare()));
        if (cm->agree_seen_plain) GOLAP_TRY(shard_agree(ctx, cm, GOLAP_OK));   // (every rank saw the same words: all take this round or none)
    }
    // value range of every key column over BOTH sides of this rank, then over all ranks
    long long init[8];
    for (uint32_t c = 0; c < 4; ++c) { init[2 * c] = INT64_MAX; init[2 * c + 1] = INT64_MIN; }
    GOLAP_HIP_TRY(ctx, hipMemcpyAsync((void*)d_mm, init, 64, hipMemcpyHostToDevice, st));
    const uint32_t cus = (uint32_t)ctx->prop.multiProcessorCount;
    for (int side = 0; side < 2; ++side)
        for (uint32_t c = 0; c < K && n[side]; ++c)
            hipLaunchKernelGGL(k_key_minmax, dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(cus * 8, ceil_div(n[side], 2048)))), dim3(256), 0, st,
                               (const long long*)cols[side][c], (const uint8_t*)valids[side][c], (unsigned long long)n[side], (long long*)d_mm + 2 * c);
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    GOLAP_HIP_TRY(ctx, hipEventRecord(cm->ev_misc, st));
    GOLAP_HIP_TRY(ctx, hipStreamWaitEvent(cm->stream, cm->ev_misc, 0));
    GOLAP_TRY(comm_all_gather(cm, d_mm, d_all, 64, cm->stream));
    long long* h = (long long*)cm->h_status;
    GOLAP_HIP_TRY(ctx, hipMemcpyAsync(h, (const void*)d_all, (size_t)P * 64, hipMemcpyDeviceToHost, cm->stream));
    GOLAP_HIP_TRY(ctx, hipStreamSynchronize(cm->stream));
    cm->info.host_syncs += 1;
    PackCols pc[2];
    memset(pc, 0, sizeof pc);
    unsigned __int128 total = 1;
    long long mn[4], mx[4];
    for (uint32_t c = 0; c < K; ++c) {
        mn[c] = INT64_MAX; mx[c] = INT64_MIN;
        for (uint32_t r = 0; r < P; ++r) { mn[c] = std::min(mn[c], h[r * 8 + 2 * c]); mx[c] = std::max(mx[c], h[r * 8 + 2 * c + 1]); }
        if (mn[c] > mx[c]) mn[c] = mx[c] = 0;   // no valid value anywhere
    }
    unsigned long long weight[4] = {1, 1, 1, 1};
    bool fits = true;
    for (uint32_t c = K; c-- > 0 && fits;) {
        weight[c] = (unsigned long long)total;
        total *= (unsigned __int128)((__int128)mx[c] - (__int128)mn[c]) + 1;
        if (total > ((unsigned __int128)1 << 62)) fits = false;   // (the same numbers on every rank: every rank takes the same branch)
    }
    if (const char* e = knob("GOLAP_SHARD_WIDE_KEYS")) if (!strcmp(e, "1")) fits = false;   // test hook: the surrogate route for tuples that would pack
    *is_wide = !fits;
    if (!fits) {
        // the tuple does not fit one exact int64: join on a 64-bit surrogate (NULL rows get keys of their own) and let the exact path ship
        // the real columns for the check (ShardWideKeys, sharded_join_exact)
        for (int side = 0; side < 2; ++side) {
            KeyCols& kc = wide_out->cols[side];
            memset(&kc, 0, sizeof kc);
            kc.n_cols = K;
            for (uint32_t c = 0; c < K; ++c) { kc.col[c] = (const long long*)cols[side][c]; kc.valid[c] = (const uint8_t*)valids[side][c]; }
            if (!n[side]) continue;
            unsigned long long keep = ~0ull;
            if (const char* e = knob("GOLAP_GJ_HASH_BITS")) { const int b = atoi(e); if (b >= 1 && b < 64) keep = (1ull << b) - 1; }   // test knob: provoke false pairs
            // (NULL rows: a mix of (side seed, GLOBAL row id) -- distinct across ranks too)
            const unsigned long long seed = (side ? 0xBB67AE8584CAA73Bull : 0x6A09E667F3BCC908ull) + (unsigned long long)(side ? p->probe_rid_base : p->build_rid_base) * 0x9E3779B97F4A7C15ull;
            hipLaunchKernelGGL(k_surrogate, dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(cus * 8, ceil_div(n[side], 1024)))), dim3(256), 0, st, kc,
                               (unsigned long long)n[side], seed, keep, (long long*)(side ? *packed_p : *packed_b));
        }
        GOLAP_HIP_TRY(ctx, hipGetLastError());
        return GOLAP_OK;
    }
    for (int side = 0; side < 2; ++side) {
        pc[side].n = K;
        for (uint32_t c = 0; c < K; ++c) {
            pc[side].col[c] = (const long long*)cols[side][c];
            pc[side].valid[c] = (const uint8_t*)valids[side][c];
            pc[side].mn[c] = mn[c];
            pc[side].weight[c] = weight[c];
        }
        if (!n[side]) continue;
        const unsigned long long null_base = (1ull << 62) + ((unsigned long long)side << 61) + (side ? p->probe_rid_base : p->build_rid_base);
        hipLaunchKernelGGL(k_pack_keys, dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(cus * 8, ceil_div(n[side], 1024)))), dim3(256), 0, st, pc[side],
                           (unsigned long long)n[side], null_base, (long long*)(side ? *packed_p : *packed_b));
    }
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

}  // namespace golap

namespace golap {

static int32_t sharded_join_run(golap_ctx* ctx, golap_comm* cm, const golap_sharded_join_params* p_in, golap_sharded_join_params* p_local_, hipStream_t st,
                                bool general, bool switched, SlabHold& packed_hold, bool& packed_on) {
    golap_sharded_join_params& p_local = *p_local_;
    const golap_sharded_join_params* p = p_local_;
    ShardWideKeys wide_keys;
    bool is_wide = false;
    if (general) {
        uint64_t pb = 0, pp = 0;
        packed_on = true;
        GOLAP_TRY(shard_pack_keys(ctx, cm, p_in, st, packed_hold, &pb, &pp, switched, &wide_keys, &is_wide));
        p_local.build_key_ptr = pb;
        p_local.probe_key_ptr = pp;
        p_local.n_key_cols = 1;
        memset(p_local.build_validity, 0, sizeof p_local.build_validity);
        memset(p_local.probe_validity, 0, sizeof p_local.probe_validity);
    }

    // The call is collective: an argument error on ONE rank must not leave the others waiting in the exchange.  It becomes this
    // rank's agreement word (shard_agree): every rank leaves the call together, this one with its error, the others with
    // GOLAP_ERR_PEER.  (The plan below only reads the totals and row-count lists, which every rank passes identically.)
    auto check_args = [&]() -> int32_t {
        if (!p->out_count_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "out_count_ptr is null");
        if (p->max_pairs && !p->out_pairs_ptr) return fail(ctx, GOLAP_ERR_INVALID_ARG, "out_pairs_ptr is null but max_pairs > 0");
        if ((p->n_build_local && !p->build_key_ptr) || (p->n_probe_local && !p->probe_key_ptr)) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null key column");
        if (p->join_type < 0 || p->join_type > 3) return fail(ctx, GOLAP_ERR_INVALID_ARG, "bad join_type %d", p->join_type);
        if (p->n_build_total >= 0xFFFFFFF0ull || p->n_probe_total >= 0xFFFFFFF0ull) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "global row ids are u32");
        if ((uint64_t)p->build_rid_base + p->n_build_local > 0xFFFFFFFFull || (uint64_t)p->probe_rid_base + p->n_probe_local > 0xFFFFFFFFull)
            return fail(ctx, GOLAP_ERR_UNSUPPORTED, "rid_base + rows exceeds the u32 row-id range");
        return GOLAP_OK;
    };
    const int32_t arg_err = check_args();
    ShardPlan sp;
    make_shard_plan((uint32_t)cm->world, p->n_build_total, p->n_probe_total, p->build_rows_per_rank, p->probe_rows_per_rank, sp);
    cm->info.extra_bits = sp.e1;
    cm->info.n_groups = sp.per;
    cm->info.local_bits = sp.local_bits;
    if (arg_err == GOLAP_OK) GOLAP_TRY(zero_async(ctx, st, (void*)p->out_count_ptr, 8));
    if (sp.eligible && !p->force_exact && !is_wide) {   // (surrogate keys are 64-bit hashes and need the shipped columns: the exact path)
        bool failed = false;
        GOLAP_TRY(sharded_join_fast(ctx, cm, p, sp, st, &failed, arg_err));
        if (!failed) { cm->info.fast_path = 1; return GOLAP_OK; }
        const uint32_t syncs = cm->info.host_syncs;
        memset(&cm->info, 0, sizeof cm->info);
        cm->info.extra_bits = sp.e1;
        cm->info.n_groups = sp.per;
        cm->info.local_bits = sp.local_bits;
        cm->info.host_syncs = syncs;
        cm->info.fast_path_abandoned = 1;
        GOLAP_TRY(zero_async(ctx, st, (void*)p->out_count_ptr, 8));
    }
    return sharded_join_exact(ctx, cm, p, sp, st, arg_err, is_wide ? &wide_keys : nullptr);
}

}  // namespace golap

extern "C" {

int32_t golap_sharded_join(golap_ctx* ctx, golap_comm* cm, const golap_sharded_join_params* p_in, void* stream) {
    if (!ctx || !cm || !p_in) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    if (cm->ctx != ctx) return fail(ctx, GOLAP_ERR_INVALID_ARG, "communicator belongs to another context");
    std::lock_guard<std::mutex> lk(ctx->mu);
    KnobScope knobs_(ctx);
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick_stream(ctx, stream);
    memset(&cm->info, 0, sizeof cm->info);
    // composite keys / NULLs: the key columns are replaced by ONE exact int64 column per side first (sharded_join_run's general form)
    golap_sharded_join_params p_local = *p_in;
    SlabHold packed_hold(ctx);
    struct PackedSync {   // the packed columns are read by kernels of this call: they go back to the pool only when the stream is done
        hipStream_t st; bool on = false;
        ~PackedSync() { if (on) (void)hipStreamSynchronize(st); }
    } packed_sync{st};
    bool general = p_in->n_key_cols > 1;
    for (int c = 0; c < 4; ++c) general = general || p_in->build_validity[c] || p_in->probe_validity[c];
    // n_key_cols is part of the query, but whether a shard comes with validity arrays depends on its DATA (a shard without NULLs is
    // handed none; the Arrow import drops them when null_count is 0): the form is settled in the call's first agreement round --
    // a plain rank that finds a general peer there (SHARD_SWITCH_TO_GENERAL) abandons what it had enqueued and starts over below.
    cm->agree_mode = general ? golap_comm::AGREE_GENERAL : 0u;
    struct ModeReset { golap_comm* cm; ~ModeReset() { cm->agree_mode = 0; } } mode_reset{cm};
    const int32_t run = sharded_join_run(ctx, cm, p_in, &p_local, st, general, false, packed_hold, packed_sync.on);
    if (run != SHARD_SWITCH_TO_GENERAL) return run;
    cm->agree_mode = golap_comm::AGREE_GENERAL;
    return sharded_join_run(ctx, cm, p_in, &p_local, st, true, true, packed_hold, packed_sync.on);
}

}  // extern "C"
