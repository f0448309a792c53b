// golap_shard_agg.inc.h -- GROUP BY over row-sharded inputs, one process per GPU (SURVEY 8(e) "Aggregate"); included behind
// golap_shard.inc.h (same communicator, transports, agreement round).  The reference has no distributed layer (README.md:414).
//
//   dense keys   every rank pre-aggregates its rows into the [statistic][group] table (golap_aggregate: LDS pre-aggregation /
//                partition-then-aggregate), the tables are all-gathered device to device and folded IN RANK ORDER by one
//                kernel -- sums, counts, minima, maxima; AVG is finalised from the folded sum and count.  A fixed fold
//                order: the result is the same on every rank and from run to run, whatever the transport does.  The result is
//                replicated.
//   hashed keys  every rank pre-aggregates its rows by key (GOLAP_KEYS_HASHED: one partial group per distinct local key), the
//                partial groups are radix-partitioned by the join's partition function (hash & 255 -> rank p mod P, owner-major,
//                golap_radix_partition), exchanged (sizes all-gathered and read by the host), and every rank merges the partial
//                groups it owns with a second hashed aggregate (sum of sums, sum of counts, min of minima, max of maxima).  The
//                result stays sharded: a key is reported by exactly one rank.
// Orchestration over the public C ABI (each step takes the context's lock itself): the call must not run concurrently with other
// calls on the same context.
namespace golap {

struct ShardAggDesc {
    uint32_t n_planes, n_specs;
    int32_t count_plane;          // plane folded into out_count (-1: none)
    uint8_t plane_func[64];       // statistic of each plane: GOLAP_AGG_SUM / COUNT / MIN / MAX
    uint8_t spec_func[64];
    uint8_t spec_a[64], spec_b[64];   // plane of the spec's statistic; AVG: a = sum plane, b = count plane
    uint32_t spec_out[64];        // output plane
};

// gathered: the flat [n_planes * G] tables of P ranks, all-gathered in chunks of `chunk` elements ([chunk index][rank][element])
__device__ __forceinline__ double shard_fold(const double* __restrict__ gathered, uint32_t P, unsigned long long E, unsigned long long chunk, uint32_t plane,
                                             unsigned long long G, unsigned long long g, uint32_t func) {
    const unsigned long long e = (unsigned long long)plane * G + g, c = e / chunk, w = e % chunk;
    const unsigned long long len = E - c * chunk < chunk ? E - c * chunk : chunk;
    const double* base = gathered + c * chunk * P + w;
    double acc = base[0];
    for (uint32_t r = 1; r < P; ++r) {   // rank order: reproducible
        const double v = base[(unsigned long long)r * len];
        acc = func == GOLAP_AGG_MIN ? (v < acc ? v : acc) : func == GOLAP_AGG_MAX ? (v > acc ? v : acc) : acc + v;
    }
    return acc;
}
__global__ __launch_bounds__(256) void k_shard_agg_fold(const double* __restrict__ gathered, uint32_t P, unsigned long long G, unsigned long long chunk,
                                                        ShardAggDesc d, double* __restrict__ out, unsigned long long out_stride,
                                                        unsigned long long* __restrict__ out_count, const uint32_t* __restrict__ n_live) {
    const unsigned long long E = (unsigned long long)d.n_planes * G;
    const unsigned long long live = n_live ? (*n_live < G ? *n_live : G) : G;   // hashed: only the slots that hold a group
    for (unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; g < live; g += (unsigned long long)gridDim.x * blockDim.x) {
        for (uint32_t s = 0; s < d.n_specs; ++s) {
            const uint32_t f = d.spec_func[s];
            const double a = shard_fold(gathered, P, E, chunk, d.spec_a[s], G, g, f == GOLAP_AGG_AVG ? (uint32_t)GOLAP_AGG_SUM : d.plane_func[d.spec_a[s]]);
            double v = a;
            if (f == GOLAP_AGG_AVG) {
                const double c = shard_fold(gathered, P, E, chunk, d.spec_b[s], G, g, GOLAP_AGG_SUM);
                v = c > 0.0 ? a / c : 0.0;   // executor.rs:339-345
            }
            out[(unsigned long long)d.spec_out[s] * out_stride + g] = v;
        }
        if (out_count && d.count_plane >= 0) out_count[g] = (unsigned long long)shard_fold(gathered, P, E, chunk, (uint32_t)d.count_plane, G, g, GOLAP_AGG_SUM);
    }
}

static int32_t shard_agg_planes(golap_ctx* ctx, const golap_aggregate_spec* specs, uint32_t n_specs, const golap_aggregate_params* p, ShardAggDesc& d,
                                std::vector<golap_aggregate_spec>& derived) {
    memset(&d, 0, sizeof d);
    d.count_plane = -1;
    if (n_specs > 64) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "golap_sharded_aggregate: at most 64 aggregates");
    derived.clear();
    auto plane_of = [&](int32_t func, uint64_t col) -> int32_t {
        for (size_t i = 0; i < derived.size(); ++i)
            if (derived[i].func == func && derived[i].value_col_index == col) return (int32_t)i;
        if (derived.size() >= 64) return -1;
        derived.push_back(golap_aggregate_spec{func, col, (uint64_t)derived.size()});
        d.plane_func[derived.size() - 1] = (uint8_t)func;
        return (int32_t)derived.size() - 1;
    };
    uint64_t last_col = 0;
    for (uint32_t i = 0; i < n_specs; ++i) {
        const golap_aggregate_spec& s = specs[i];
        if (s.func < GOLAP_AGG_SUM || s.func > GOLAP_AGG_AVG) return fail(ctx, GOLAP_ERR_INVALID_ARG, "bad aggregate function %d", s.func);
        if (s.value_col_index >= p->n_value_cols) return fail(ctx, GOLAP_ERR_INVALID_ARG, "aggregate %u reads value column %llu of %u", i,
                                                              (unsigned long long)s.value_col_index, p->n_value_cols);
        const int32_t a = plane_of(s.func == GOLAP_AGG_AVG ? (int32_t)GOLAP_AGG_SUM : s.func, s.value_col_index);
        const int32_t b = s.func == GOLAP_AGG_AVG ? plane_of(GOLAP_AGG_COUNT, s.value_col_index) : 0;
        if (a < 0 || b < 0) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "golap_sharded_aggregate: more than 64 distinct statistics");
        d.spec_func[i] = (uint8_t)s.func;
        d.spec_a[i] = (uint8_t)a;
        d.spec_b[i] = (uint8_t)b;
        d.spec_out[i] = (uint32_t)s.output_col_index;
        last_col = std::max<uint64_t>(last_col, s.value_col_index);
    }
    if (p->out_count_ptr && n_specs) {   // rows per group = the non-NULL rows of the LAST aggregated column (golap_aggregate_params::out_count_ptr)
        d.count_plane = plane_of(GOLAP_AGG_COUNT, last_col);
        if (d.count_plane < 0) return fail(ctx, GOLAP_ERR_UNSUPPORTED, "golap_sharded_aggregate: more than 64 distinct statistics");
    }
    d.n_planes = (uint32_t)derived.size();
    d.n_specs = n_specs;
    return GOLAP_OK;
}

}  // namespace golap

extern "C" int32_t golap_sharded_aggregate(golap_ctx* ctx, golap_comm* cm, const golap_aggregate_spec* specs, uint32_t n_specs,
                                           const golap_aggregate_params* p, void* stream) {
    using namespace golap;
    if (!ctx || !cm || !p || (n_specs && !specs)) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    if (cm->ctx != ctx) return fail(ctx, GOLAP_ERR_INVALID_ARG, "communicator belongs to another context");
    GOLAP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick_stream(ctx, stream);
    const uint32_t P = (uint32_t)cm->world, me = (uint32_t)cm->rank;
    const bool hashed = p->key_mode == GOLAP_KEYS_HASHED;
    const uint64_t G = std::max<uint32_t>(1, p->n_groups);
    ShardAggDesc d;
    std::vector<golap_aggregate_spec> derived;
    SlabHold hold(ctx);             // (before `drain`: on an error path the streams are emptied before the slabs go back to the pool)
    ShardDrain drain(ctx, cm, st);
    uint64_t t_local = 0, t_all = 0, k_local = 0, d_n = 0;
    auto prepare = [&]() -> int32_t {
        if (!p->out_ptr && n_specs) return fail(ctx, GOLAP_ERR_INVALID_ARG, "out_ptr is null");
        if (hashed && (!p->out_keys_ptr || !p->out_n_groups_ptr)) return fail(ctx, GOLAP_ERR_INVALID_ARG, "GOLAP_KEYS_HASHED needs out_keys_ptr and out_n_groups_ptr");
        if (n_specs == 0) return fail(ctx, GOLAP_ERR_INVALID_ARG, "golap_sharded_aggregate needs at least one aggregate");
        GOLAP_TRY(shard_agg_planes(ctx, specs, n_specs, p, d, derived));
        GOLAP_TRY(ensure_host_words(cm, (uint64_t)P * (P + 2) * 4 + 4096));
        t_local = hold.take((uint64_t)d.n_planes * G * 8);
        if (!hashed) t_all = hold.take((uint64_t)P * d.n_planes * G * 8);
        else { k_local = hold.take(G * 8); d_n = hold.take(256); }
        if (!t_local || (!hashed && !t_all) || (hashed && (!k_local || !d_n))) return fail(ctx, GOLAP_ERR_OOM, "sharded aggregate: table buffers");
        return GOLAP_OK;
    };
    GOLAP_TRY(shard_agree(ctx, cm, prepare()));
    // ---- local pre-aggregation: the statistics every requested aggregate is made of ----
    golap_aggregate_params q = *p;
    q.out_ptr = t_local;
    q.out_count_ptr = 0;
    if (hashed) { q.out_keys_ptr = k_local; q.out_n_groups_ptr = d_n; }
    // (its argument checks and scratch allocations can fail on ONE rank -- a larger shard running out of memory, a null pointer of
    // this rank only: the outcome is agreed on before the next collective, like prepare()'s)
    int32_t local_err = golap_aggregate(ctx, derived.data(), d.n_planes, &q, st);
    const uint32_t fold_grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)ctx->prop.multiProcessorCount * 8, ceil_div(G, 256)));
    if (!hashed) {
        // ---- all-gather the tables (in messages of at most max_msg bytes), fold in rank order ----
        GOLAP_TRY(shard_agree(ctx, cm, local_err));   // (on the communicator's stream: the host waits beside the local aggregate's kernels, not behind them)
        const uint64_t E = (uint64_t)d.n_planes * G, chunk = std::max<uint64_t>(1, std::min<uint64_t>(E, shard_max_msg(cm, 0) / 8));
        GOLAP_HIP_TRY(ctx, hipEventRecord(cm->ev_misc, st));
        GOLAP_HIP_TRY(ctx, hipStreamWaitEvent(cm->stream, cm->ev_misc, 0));
        for (uint64_t c = 0; c * chunk < E; ++c) {
            const uint64_t len = std::min(chunk, E - c * chunk);
            GOLAP_TRY(comm_all_gather(cm, t_local + c * chunk * 8, t_all + c * chunk * P * 8, len * 8, cm->stream));
        }
        GOLAP_HIP_TRY(ctx, hipEventRecord(cm->ev_hdr, cm->stream));
        GOLAP_HIP_TRY(ctx, hipStreamWaitEvent(st, cm->ev_hdr, 0));
        hipLaunchKernelGGL(k_shard_agg_fold, dim3(fold_grid), dim3(256), 0, st, (const double*)t_all, P, (unsigned long long)G, (unsigned long long)chunk, d,
                           (double*)p->out_ptr, (unsigned long long)G, (unsigned long long*)p->out_count_ptr, (const uint32_t*)nullptr);
        GOLAP_HIP_TRY(ctx, hipGetLastError());
        GOLAP_HIP_TRY(ctx, hipStreamSynchronize(st));   // the tables go back to the pool at scope exit
        drain.armed = false;
        return GOLAP_OK;
    }
    // ---- hashed keys: partition the partial groups by owner, exchange, merge ----
    uint32_t nl = 0;
    if (local_err == GOLAP_OK) {
        GOLAP_HIP_TRY(ctx, hipMemcpyAsync(cm->h_status, (const void*)d_n, 4, hipMemcpyDeviceToHost, st));
        GOLAP_HIP_TRY(ctx, hipStreamSynchronize(st));
        nl = cm->h_status[0];
    }
    if (local_err == GOLAP_OK && nl > G) local_err = fail(ctx, GOLAP_ERR_CAPACITY, "golap_sharded_aggregate: %u distinct local keys exceed the bound n_groups = %llu", nl, (unsigned long long)G);
    const uint64_t nl1 = std::max<uint32_t>(nl, 1);
    const uint64_t pk = hold.take(nl1 * 8), pr = hold.take(nl1 * 4), offs = hold.take(257 * 4), cnt_all = hold.take((uint64_t)P * (P + 1) * 4 + 512);
    uint64_t planes_sorted = hold.take((uint64_t)d.n_planes * nl1 * 8);
    if (local_err == GOLAP_OK && (!pk || !pr || !offs || !cnt_all || !planes_sorted)) local_err = fail(ctx, GOLAP_ERR_OOM, "sharded aggregate: partition buffers");
    GOLAP_TRY(shard_agree(ctx, cm, local_err));
    uint32_t send_cnt[64] = {0}, send_off[65] = {0};
    // partition + gather run only on ranks that hold partial groups and have argument checks and scratch of their own: what
    // they return rides as word P of this rank's unit in the counts all-gather below -- every rank takes part in it whatever
    // happened here, and every rank learns of a failure from the same read-back
    auto partition_local = [&]() -> int32_t {
        golap_partition_params pp;
        memset(&pp, 0, sizeof pp);
        pp.key_ptr = k_local;
        pp.n_rows = nl;
        pp.n_owners = P;
        pp.out_key_ptr = pk;
        pp.out_row_id_ptr = pr;
        pp.out_offsets_ptr = offs;
        GOLAP_TRY(golap_radix_partition(ctx, &pp, st));
        for (uint32_t pl = 0; pl < d.n_planes; ++pl)
            GOLAP_TRY(golap_take_8byte(ctx, t_local + (uint64_t)pl * G * 8, pr, 0, nl, planes_sorted + (uint64_t)pl * nl1 * 8, st));
        GOLAP_HIP_TRY(ctx, hipMemcpyAsync(cm->h_status, (const void*)offs, 257 * 4, hipMemcpyDeviceToHost, st));
        GOLAP_HIP_TRY(ctx, hipStreamSynchronize(st));
        const uint32_t per = 256 / P;
        for (uint32_t r = 0; r < P; ++r) { send_off[r] = cm->h_status[r * per]; send_cnt[r] = cm->h_status[(r + 1) * per] - cm->h_status[r * per]; }
        return GOLAP_OK;
    };
    int32_t part_err = GOLAP_OK;
    if (nl) part_err = partition_local();
    else GOLAP_HIP_TRY(ctx, hipStreamSynchronize(st));
    const std::string part_msg = part_err != GOLAP_OK ? ctx->err : std::string();
    if (part_err != GOLAP_OK) for (uint32_t r = 0; r < P; ++r) send_cnt[r] = 0;
    // sizes: every rank's per-owner counts (+ its status word), all-gathered and read back
    const uint32_t U = P + 1;
    for (uint32_t r = 0; r < P; ++r) cm->h_agree[golap_comm::AGREE_SCRATCH + r] = send_cnt[r];
    cm->h_agree[golap_comm::AGREE_SCRATCH + P] = (uint32_t)part_err;
    GOLAP_HIP_TRY(ctx, hipMemcpyAsync((void*)cnt_all, cm->h_agree + golap_comm::AGREE_SCRATCH, (size_t)U * 4, hipMemcpyHostToDevice, cm->stream));
    GOLAP_TRY(comm_all_gather(cm, cnt_all, cnt_all + 512, (uint64_t)U * 4, cm->stream));
    uint32_t* h_all = cm->h_status + 512;
    GOLAP_HIP_TRY(ctx, hipMemcpyAsync(h_all, (const void*)(cnt_all + 512), (size_t)P * U * 4, hipMemcpyDeviceToHost, cm->stream));
    GOLAP_HIP_TRY(ctx, hipStreamSynchronize(cm->stream));
    if (part_err != GOLAP_OK) { ctx->err = part_msg; set_thread_error(part_msg.c_str()); return part_err; }
    for (uint32_t r = 0; r < P; ++r)
        if (h_all[r * U + P] != 0u)
            return fail(ctx, GOLAP_ERR_PEER, "golap_sharded_aggregate: rank %u failed before the exchange (its error code %u); no rank exchanged anything", r,
                        h_all[r * U + P]);
    std::vector<uint32_t> h_cnt_v((size_t)P * P);
    for (uint32_t r = 0; r < P; ++r)
        for (uint32_t c = 0; c < P; ++c) h_cnt_v[(size_t)r * P + c] = h_all[r * U + c];
    const uint32_t* h_cnt = h_cnt_v.data();
    uint64_t nr = 0, recv_off[65] = {0};
    for (uint32_t r = 0; r < P; ++r) { recv_off[r] = nr; nr += h_cnt[r * P + me]; }
    recv_off[P] = nr;
    const uint64_t nr1 = std::max<uint64_t>(nr, 1);
    const uint64_t rk = hold.take(nr1 * 8), rplanes = hold.take((uint64_t)d.n_planes * nr1 * 8), merged = hold.take((uint64_t)d.n_planes * G * 8);
    local_err = (rk && rplanes && merged) ? GOLAP_OK : fail(ctx, GOLAP_ERR_OOM, "sharded aggregate: receive buffers");
    if (local_err == GOLAP_OK && nr > 0xFFFFFFF0ull) local_err = fail(ctx, GOLAP_ERR_UNSUPPORTED, "sharded aggregate: more than 2^32 partial groups on one rank");
    GOLAP_TRY(shard_agree(ctx, cm, local_err));
    {
        const uint64_t max_msg = shard_max_msg(cm, 0);
        std::vector<golap_p2p_op> ops;
        for (uint32_t r = 0; r < P; ++r) {
            const uint64_t sc = send_cnt[r], rc = h_cnt[r * P + me];
            if (r == me) {
                if (sc) {
                    GOLAP_HIP_TRY(ctx, hipMemcpyAsync((void*)(rk + recv_off[r] * 8), (const void*)(pk + (uint64_t)send_off[r] * 8), sc * 8, hipMemcpyDeviceToDevice, cm->stream));
                    for (uint32_t pl = 0; pl < d.n_planes; ++pl)
                        GOLAP_HIP_TRY(ctx, hipMemcpyAsync((void*)(rplanes + ((uint64_t)pl * nr1 + recv_off[r]) * 8),
                                                          (const void*)(planes_sorted + ((uint64_t)pl * nl1 + send_off[r]) * 8), sc * 8, hipMemcpyDeviceToDevice, cm->stream));
                }
                continue;
            }
            split_message((int32_t)r, true, pk + (uint64_t)send_off[r] * 8, sc * 8, max_msg, ops);
            split_message((int32_t)r, false, rk + recv_off[r] * 8, rc * 8, max_msg, ops);
            for (uint32_t pl = 0; pl < d.n_planes; ++pl) {
                split_message((int32_t)r, true, planes_sorted + ((uint64_t)pl * nl1 + send_off[r]) * 8, sc * 8, max_msg, ops);
                split_message((int32_t)r, false, rplanes + ((uint64_t)pl * nr1 + recv_off[r]) * 8, rc * 8, max_msg, ops);
            }
        }
        GOLAP_TRY(comm_exchange(cm, ops, cm->stream));
        GOLAP_HIP_TRY(ctx, hipEventRecord(cm->ev_hdr, cm->stream));
        GOLAP_HIP_TRY(ctx, hipStreamWaitEvent(st, cm->ev_hdr, 0));
    }
    // ---- merge the partial groups this rank owns: a hashed aggregate over the received (key, statistic...) rows ----
    {
        std::vector<golap_aggregate_spec> ms(d.n_planes);
        std::vector<uint64_t> vptrs(d.n_planes);
        std::vector<int32_t> vdt(d.n_planes, GOLAP_DT_FLOAT64);
        for (uint32_t pl = 0; pl < d.n_planes; ++pl) {
            const int32_t f = d.plane_func[pl];
            ms[pl] = golap_aggregate_spec{f == GOLAP_AGG_COUNT ? (int32_t)GOLAP_AGG_SUM : f, pl, pl};   // counts add up
            vptrs[pl] = rplanes + (uint64_t)pl * nr1 * 8;
        }
        golap_aggregate_params m;
        memset(&m, 0, sizeof m);
        m.group_key_ptr = rk;
        m.value_ptrs = vptrs.data();
        m.n_value_cols = d.n_planes;
        m.n_rows = nr;
        m.n_groups = (uint32_t)G;
        m.out_ptr = merged;
        m.key_dtype = GOLAP_DT_INT64;
        m.value_dtypes = vdt.data();
        m.key_mode = GOLAP_KEYS_HASHED;
        m.out_keys_ptr = p->out_keys_ptr;
        m.out_n_groups_ptr = p->out_n_groups_ptr;
        GOLAP_TRY(golap_aggregate(ctx, ms.data(), d.n_planes, &m, st));
        // (a partial group with no non-NULL value carries MIN = +max / MAX = -max, the identities: they survive the merge unchanged)
        hipLaunchKernelGGL(k_shard_agg_fold, dim3(fold_grid), dim3(256), 0, st, (const double*)merged, 1u, (unsigned long long)G, (unsigned long long)((uint64_t)d.n_planes * G), d,
                           (double*)p->out_ptr, (unsigned long long)G, (unsigned long long*)p->out_count_ptr, (const uint32_t*)p->out_n_groups_ptr);
        GOLAP_HIP_TRY(ctx, hipGetLastError());
    }
    GOLAP_HIP_TRY(ctx, hipStreamSynchronize(st));
    drain.armed = false;
    return GOLAP_OK;
}
