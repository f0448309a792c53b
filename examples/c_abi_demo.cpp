// c_abi_demo.cpp -- a host that is neither Python nor torch drives the C ABI directly:
// context -> SlabAllocator -> TransferQueue (H2D) -> GpuFilter / GpuHashJoin / GpuAggregate -> D2H.
// Self-checking with plain loops (the CPU oracle is test infrastructure and is not used here).
//   build:  see tests/test_c_abi_demo.py (g++ -Iinclude ... -lgolap_hip, rpath to gpu_olap_engine_amd/lib and /opt/rocm/lib)
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "golap.h"

#define CK(expr)                                                                            \
    do {                                                                                    \
        int32_t _s = (expr);                                                                \
        if (_s != GOLAP_OK) { fprintf(stderr, "%s -> %d: %s\n", #expr, _s, golap_last_error(ctx)); return 1; } \
    } while (0)

struct Slab { uint64_t ptr = 0; uint32_t cls = 0; };

int main() {
    golap_engine_config cfg;
    golap_engine_config_default(&cfg);          // EngineConfig::default(), lib.rs:33-43
    cfg.max_gpu_memory = 0;
    golap_ctx* ctx = nullptr;
    int32_t st = golap_ctx_create(&cfg, &ctx);
    if (st != GOLAP_OK) { fprintf(stderr, "ctx_create -> %d: %s\n", st, golap_last_error(nullptr)); return st == GOLAP_ERR_NO_DEVICE ? 77 : 1; }
    golap_device_info info;
    CK(golap_device_info_get(ctx, &info));
    printf("device: %s (%s), %u CUs, wave %u; library built with %s, hidden-memory kernel forms %s (audited with %s)\n", info.name, info.gcn_arch, info.compute_units,
           info.wavefront_size, info.built_with, info.hidden_memory_forms ? "on" : "off", info.audited_with);

    const uint64_t nb = 200000, np = 300000;
    std::vector<int64_t> bk(nb), pk(np), pv(np);
    for (uint64_t i = 0; i < nb; ++i) bk[i] = (int64_t)((2654435761ull * i + 12345ull) % nb);    // unique build keys
    for (uint64_t j = 0; j < np; ++j) { pk[j] = (int64_t)((j * 7919ull) % (nb + nb / 2)); pv[j] = (int64_t)(j % 1024); }

    auto alloc = [&](uint64_t bytes, Slab& s) { return golap_slab_alloc(ctx, bytes, &s.ptr, &s.cls); };
    Slab dbk, dpk, dpv, dpairs, dcnt, dmask, dagg, dfs;
    CK(alloc(nb * 8, dbk)); CK(alloc(np * 8, dpk)); CK(alloc(np * 8, dpv)); CK(alloc(np * 8, dpairs)); CK(alloc(8, dcnt));
    CK(alloc(np, dmask)); CK(alloc(5 * 8, dagg)); CK(alloc(16, dfs));
    CK(golap_xfer_enqueue(ctx, GOLAP_HOST_TO_DEVICE, (uint64_t)bk.data(), dbk.ptr, nb * 8, 0));
    CK(golap_xfer_enqueue(ctx, GOLAP_HOST_TO_DEVICE, (uint64_t)pk.data(), dpk.ptr, np * 8, 1));
    CK(golap_xfer_enqueue(ctx, GOLAP_HOST_TO_DEVICE, (uint64_t)pv.data(), dpv.ptr, np * 8, 2));
    CK(golap_xfer_synchronize(ctx));

    // ---- GpuHashJoin (JoinKernel::join with choose_strategy, join_kernel.rs:39-77) ----
    golap_join_params jp{};
    jp.build_key_ptr = dbk.ptr; jp.probe_key_ptr = dpk.ptr; jp.n_build = nb; jp.n_probe = np;
    jp.out_pairs_ptr = dpairs.ptr; jp.out_count_ptr = dcnt.ptr; jp.max_pairs = np;
    for (golap_join_strategy strat : {GOLAP_JOIN_RADIX_HASH, golap_choose_strategy(nb)}) {
        CK(golap_join(ctx, strat, &jp, nullptr));
        CK(golap_ctx_synchronize(ctx, nullptr));
        uint64_t m = 0;
        CK(golap_xfer_enqueue(ctx, GOLAP_DEVICE_TO_HOST, dcnt.ptr, (uint64_t)&m, 8, 0));
        CK(golap_xfer_synchronize(ctx));
        std::vector<uint32_t> pairs(2 * m);
        CK(golap_xfer_enqueue(ctx, GOLAP_DEVICE_TO_HOST, dpairs.ptr, (uint64_t)pairs.data(), m * 8, 0));
        CK(golap_xfer_synchronize(ctx));
        std::map<int64_t, uint32_t> where;
        for (uint64_t i = 0; i < nb; ++i) where[bk[i]] = (uint32_t)i;
        uint64_t expect = 0;
        for (uint64_t j = 0; j < np; ++j) expect += where.count(pk[j]);
        uint64_t ok = 0;
        std::vector<uint8_t> seen(np, 0);
        for (uint64_t i = 0; i < m; ++i) {
            const uint32_t b = pairs[2 * i], p = pairs[2 * i + 1];
            if (b < nb && p < np && bk[b] == pk[p] && !seen[p]) { ++ok; seen[p] = 1; }
        }
        printf("join strategy %d: %llu pairs (expected %llu), %llu verified\n", (int)strat, (unsigned long long)m, (unsigned long long)expect,
               (unsigned long long)ok);
        if (m != expect || ok != m) return 2;
    }

    // ---- GpuFilter (FilterKernel::launch, filter_kernel.rs:42) + fused filter -> Sum (config 1 shape) ----
    golap_filter_predicate pred{0, golap_parse_cmp_op("<"), 100000};
    uint32_t grid = 0, block = 0;
    golap_compute_launch_params(np, &grid, &block);
    golap_filter_params fp{dpk.ptr, np, dmask.ptr, grid, block};
    CK(golap_filter_launch(ctx, &pred, &fp, nullptr));
    CK(golap_filter_sum(ctx, &pred, dpk.ptr, dpv.ptr, 0, np, dfs.ptr, nullptr));
    CK(golap_ctx_synchronize(ctx, nullptr));
    std::vector<uint8_t> mask(np);
    double fs[2];
    CK(golap_xfer_enqueue(ctx, GOLAP_DEVICE_TO_HOST, dmask.ptr, (uint64_t)mask.data(), np, 0));
    CK(golap_xfer_enqueue(ctx, GOLAP_DEVICE_TO_HOST, dfs.ptr, (uint64_t)fs, 16, 1));
    CK(golap_xfer_synchronize(ctx));
    double esum = 0; uint64_t ekept = 0, mask_bad = 0;
    for (uint64_t j = 0; j < np; ++j) { const bool k = pk[j] < 100000; mask_bad += mask[j] != (k ? 1 : 0); if (k) { esum += (double)pv[j]; ++ekept; } }
    printf("filter: %llu mask mismatches; filter_sum = %.1f over %.0f rows (expected %.1f over %llu)\n", (unsigned long long)mask_bad, fs[0], fs[1], esum,
           (unsigned long long)ekept);
    if (mask_bad || fs[0] != esum || fs[1] != (double)ekept) return 3;

    // ---- GpuAggregate (AggregateKernel::aggregate, aggregate_kernel.rs:80): ungrouped SUM/COUNT/MIN/MAX/AVG ----
    golap_aggregate_spec specs[5] = {{GOLAP_AGG_SUM, 0, 0}, {GOLAP_AGG_COUNT, 0, 1}, {GOLAP_AGG_MIN, 0, 2}, {GOLAP_AGG_MAX, 0, 3}, {GOLAP_AGG_AVG, 0, 4}};
    uint64_t vptr = dpv.ptr; int32_t vdt = GOLAP_DT_INT64;
    golap_aggregate_params ap{};
    ap.value_ptrs = &vptr; ap.n_value_cols = 1; ap.n_rows = np; ap.n_groups = 1; ap.out_ptr = dagg.ptr; ap.value_dtypes = &vdt;
    CK(golap_aggregate(ctx, specs, 5, &ap, nullptr));
    CK(golap_ctx_synchronize(ctx, nullptr));
    double agg[5];
    CK(golap_xfer_enqueue(ctx, GOLAP_DEVICE_TO_HOST, dagg.ptr, (uint64_t)agg, 40, 0));
    CK(golap_xfer_synchronize(ctx));
    double s = 0, mn = 1e300, mx = -1e300;
    for (uint64_t j = 0; j < np; ++j) { s += (double)pv[j]; mn = std::min(mn, (double)pv[j]); mx = std::max(mx, (double)pv[j]); }
    printf("aggregate: sum %.1f count %.0f min %.0f max %.0f avg %.6f\n", agg[0], agg[1], agg[2], agg[3], agg[4]);
    if (agg[0] != s || agg[1] != (double)np || agg[2] != mn || agg[3] != mx || agg[4] != s / (double)np) return 4;

    // ---- device-side ingest and materialisation (ABI 4; record_batch_convert.rs:35-110 and :146-178 without a host loop): an Int32 column
    //      with an Arrow validity BITMAP goes up as it is (4 bytes per row + 1 bit), is expanded / widened / zero-filled on the device,
    //      filtered (a NULL never passes), gathered with its validity, and comes back as values + bitmap + NULL count ----
    {
        const uint64_t n = 100003;
        std::vector<int32_t> col(n);
        std::vector<uint8_t> bitmap((n + 7) / 8, 0);
        for (uint64_t i = 0; i < n; ++i) {
            col[i] = (int32_t)((i * 2654435761ull) % 2000) - 1000;
            if (i % 7 != 3) bitmap[i >> 3] |= (uint8_t)(1u << (i & 7));       // every 7th row NULL
        }
        Slab draw, dbits, dval, dwide, didx, dn, dtake, dtv, dpackv, dnulls;
        CK(alloc(n * 4, draw)); CK(alloc(bitmap.size(), dbits)); CK(alloc(n, dval)); CK(alloc(n * 8, dwide)); CK(alloc(n * 4, didx)); CK(alloc(8, dn));
        CK(alloc(n * 8, dtake)); CK(alloc(n, dtv)); CK(alloc((n + 7) / 8, dpackv)); CK(alloc(8, dnulls));
        CK(golap_xfer_enqueue(ctx, GOLAP_HOST_TO_DEVICE, (uint64_t)col.data(), draw.ptr, n * 4, 0));
        CK(golap_xfer_enqueue(ctx, GOLAP_HOST_TO_DEVICE, (uint64_t)bitmap.data(), dbits.ptr, bitmap.size(), 1));
        CK(golap_xfer_synchronize(ctx));
        CK(golap_expand_validity(ctx, dbits.ptr, 0, n, dval.ptr, nullptr));                          // is_valid(i) per row, :36-40
        CK(golap_widen_to_8byte(ctx, draw.ptr, GOLAP_WIDEN_I32, dval.ptr, n, dwide.ptr, nullptr));   // as i64, NULL -> 0, :102-110
        golap_filter_predicate neg{0, golap_parse_cmp_op("<"), 0};
        CK(golap_filter_compact_nullable(ctx, &neg, dwide.ptr, dval.ptr, n, didx.ptr, dn.ptr, nullptr));
        CK(golap_ctx_synchronize(ctx, nullptr));
        uint64_t kept = 0;
        CK(golap_xfer_enqueue(ctx, GOLAP_DEVICE_TO_HOST, dn.ptr, (uint64_t)&kept, 8, 0));
        CK(golap_xfer_synchronize(ctx));
        CK(golap_take_8byte(ctx, dwide.ptr, didx.ptr, 0, kept, dtake.ptr, nullptr));                 // take_int64, executor.rs:496-507
        CK(golap_take_1byte(ctx, dval.ptr, didx.ptr, 0, kept, dtv.ptr, nullptr));                    // ... and the rows' validity
        CK(golap_pack_validity(ctx, dval.ptr, n, dpackv.ptr, dnulls.ptr, nullptr));                  // buffer_to_array's bitmap, :146-178
        CK(golap_ctx_synchronize(ctx, nullptr));
        std::vector<int64_t> got(kept);
        std::vector<uint8_t> gotv(kept), back(bitmap.size());
        uint64_t nulls = 0;
        CK(golap_xfer_enqueue(ctx, GOLAP_DEVICE_TO_HOST, dtake.ptr, (uint64_t)got.data(), kept * 8, 0));
        CK(golap_xfer_enqueue(ctx, GOLAP_DEVICE_TO_HOST, dtv.ptr, (uint64_t)gotv.data(), kept, 1));
        CK(golap_xfer_enqueue(ctx, GOLAP_DEVICE_TO_HOST, dpackv.ptr, (uint64_t)back.data(), back.size(), 2));
        CK(golap_xfer_enqueue(ctx, GOLAP_DEVICE_TO_HOST, dnulls.ptr, (uint64_t)&nulls, 8, 3));
        CK(golap_xfer_synchronize(ctx));
        uint64_t e = 0, bad = 0, enulls = 0;
        for (uint64_t i = 0; i < n; ++i) {
            const bool valid = i % 7 != 3;
            enulls += !valid;
            if (valid && col[i] < 0) { bad += e >= kept || got[e] != (int64_t)col[i] || gotv[e] != 1; ++e; }
        }
        printf("ingest: %llu negative non-NULL rows kept (expected %llu), %llu mismatches, %llu NULLs (expected %llu), bitmap %s\n", (unsigned long long)kept,
               (unsigned long long)e, (unsigned long long)bad, (unsigned long long)nulls, (unsigned long long)enulls, back == bitmap ? "round-trips" : "DIFFERS");
        if (kept != e || bad || nulls != enulls || back != bitmap) return 5;
        for (Slab* sl : {&draw, &dbits, &dval, &dwide, &didx, &dn, &dtake, &dtv, &dpackv, &dnulls}) CK(golap_slab_free(ctx, sl->ptr, sl->cls));
    }

    for (Slab* sl : {&dbk, &dpk, &dpv, &dpairs, &dcnt, &dmask, &dagg, &dfs}) CK(golap_slab_free(ctx, sl->ptr, sl->cls));
    golap_ctx_destroy(ctx);
    printf("c_abi_demo ok\n");
    return 0;
}
