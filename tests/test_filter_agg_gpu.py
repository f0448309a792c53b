"""GPU parity tests for GpuFilter / GpuAggregate / widening vs the CPU oracle (bit-exact:
integer work exactly; f64 sums on integer-valued data are order-independent and exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import oracle  # noqa: E402
from gpu_olap_engine_amd import _native as N  # noqa: E402

OPS = ["=", "==", "!=", "<>", ">", ">=", "<", "<=", "AND", "like"]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_reference_known_answer_mask(ctx):
    """executor.rs:553-570: [5,15,25] > 10 -> [false,true,true]."""
    m = ctx.filter_mask(dev(np.array([5, 15, 25], np.int64)), ">", 10).cpu().numpy()
    assert m.tolist() == [0, 1, 1]


@pytest.mark.parametrize("op", OPS)
@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 255, 4096, 4097, 100_003])
def test_filter_mask_and_compact(ctx, op, n):
    rng = np.random.default_rng(n + len(op))
    col = rng.integers(-50, 50, n).astype(np.int64)
    lit = 7
    exp = oracle.eval_filter_mask(col, op, lit)
    d = dev(col) if n else torch.empty(0, dtype=torch.int64, device="cuda")
    got = ctx.filter_mask(d, op, lit).cpu().numpy()
    assert np.array_equal(got, exp)
    idx, cnt = ctx.filter_compact(d, op, lit)
    torch.cuda.synchronize()
    c = int(cnt.item())
    eidx = oracle.mask_to_indices(exp)
    assert c == eidx.size
    assert np.array_equal(idx[:c].cpu().numpy().view(np.uint32), eidx)   # original row order (executor.rs:479-484)
    if c:
        f = rng.random(n)
        taken = ctx.take(dev(f), idx, cnt, n_max=n)[:c].cpu().numpy()
        assert np.array_equal(taken, oracle.take_8byte(f, eidx))
        taken_i = ctx.take(d, idx, cnt, n_max=n)[:c].cpu().numpy()
        assert np.array_equal(taken_i, col[eidx.astype(np.int64)])


def test_filter_mask_unaligned_views(ctx):
    col = np.arange(-500, 1500, dtype=np.int64)
    d = dev(col)
    for off in (1, 2, 3):
        got = ctx.filter_mask(d[off:], "<", 100).cpu().numpy()
        assert np.array_equal(got, oracle.eval_filter_mask(col[off:], "<", 100))


def test_extreme_literals(ctx):
    col = np.array([-(2**63), -1, 0, 1, 2**63 - 1], np.int64)
    for op in ("<", "<=", ">", ">=", "=", "!="):
        for lit in (-(2**63), 0, 2**63 - 1):
            assert np.array_equal(ctx.filter_mask(dev(col), op, lit).cpu().numpy(), oracle.eval_filter_mask(col, op, lit))


def test_config1_filter_sum(ctx):
    """BASELINE config 1: SELECT sum(v) FROM t WHERE k<c on a 1M-row batch."""
    n = 1_000_000
    k = oracle.gen_sm64_mod(1, 0, n, 1000)
    v = oracle.gen_sm64_mod(2, 0, n, 1024)
    es, ek = oracle.filter_sum(k, v, "<", 500)
    got = ctx.filter_sum(dev(k), dev(v), "<", 500).cpu().numpy()
    assert got[0] == es and got[1] == ek
    # the unfused route: compact -> take -> ungrouped Sum gives the same scalar
    idx, cnt = ctx.filter_compact(dev(k), "<", 500)
    taken = ctx.take(dev(v), idx, cnt, n_max=n)
    c = int(cnt.item())
    s = ctx.aggregate([(N.AGG_SUM, 0)], [taken[:c].contiguous()]).cpu().numpy()
    assert s[0, 0] == es
    gf = ctx.filter_sum(dev(k), dev(v.astype(np.float64)), "<", 500).cpu().numpy()
    assert gf[0] == es and gf[1] == ek


@pytest.mark.parametrize("op", ["=", "!=", "<", ">=", "like"])
@pytest.mark.parametrize("n", [1, 5, 4097, 300_007])
def test_filters_with_validity(ctx, op, n):
    """a NULL key (validity byte 0, column_buffer.rs:59-60) never passes a predicate -- also `!=` and also
    when the zero-filled value under it would; a NULL value is left out of the fused sum.  Against the oracle's mask on the
    valid rows (eval_filter_mask, executor.rs:443-476)."""
    rng = np.random.default_rng(n * 7 + len(op))
    col = rng.integers(-20, 20, n).astype(np.int64)
    val = rng.integers(-1000, 1000, n).astype(np.int64)
    kv = (rng.random(n) > 0.3).astype(np.uint8)
    vv = (rng.random(n) > 0.2).astype(np.uint8)
    col[kv == 0] = 0                                             # what ingest leaves under a NULL (record_batch_convert.rs:102-110)
    exp = oracle.eval_filter_mask(col, op, 0).astype(bool) & (kv != 0)
    m = ctx.filter_mask(dev(col), op, 0, valid=dev(kv)).cpu().numpy()
    assert np.array_equal(m.astype(bool), exp)
    idx, cnt = ctx.filter_compact(dev(col), op, 0, valid=dev(kv))
    c = int(cnt.item())
    assert np.array_equal(idx.cpu().numpy()[:c].view(np.uint32), np.flatnonzero(exp).astype(np.uint32))
    got = ctx.filter_sum(dev(col), dev(val), op, 0, key_valid=dev(kv), value_valid=dev(vv)).cpu().numpy()
    both = exp & (vv != 0)                                       # NULL values are skipped by the sum AND the count (SQL aggregates)
    assert got[1] == float(both.sum()) and got[0] == float(val[both].sum())
    # no validity at all == all valid
    m0 = ctx.filter_mask(dev(col), op, 0).cpu().numpy()
    assert np.array_equal(m0, ctx.filter_mask(dev(col), op, 0, valid=dev(np.ones(n, np.uint8))).cpu().numpy())


@pytest.mark.parametrize("dtype", [np.int64, np.float64])
@pytest.mark.parametrize("n", [0, 1, 1000, 1_000_003])
def test_ungrouped_aggregate_all_funcs(ctx, dtype, n):
    """execute_aggregate (executor.rs:297-355): SUM, COUNT, MIN, MAX, AVG as f64; empty input keeps the seeds."""
    rng = np.random.default_rng(n)
    v = rng.integers(-1000, 1000, n).astype(dtype)
    exp = oracle.execute_aggregate(v)
    specs = [(N.AGG_SUM, 0), (N.AGG_COUNT, 0), (N.AGG_MIN, 0), (N.AGG_MAX, 0), (N.AGG_AVG, 0)]
    d = dev(v) if n else torch.empty(0, dtype=torch.from_numpy(v).dtype, device="cuda")
    got = ctx.aggregate(specs, [d]).cpu().numpy()[:, 0]
    assert got.tolist() == [exp["sum"], exp["count"], exp["min"], exp["max"], exp["avg"]]


@pytest.mark.parametrize("key_dtype", [np.int32, np.int64])
@pytest.mark.parametrize("G", [1, 7, 1024, 2048, 5000, 100_000])
def test_grouped_aggregate(ctx, key_dtype, G):
    """GROUP BY dense key -> SUM/COUNT/MIN/MAX/AVG per group, sorted by key (= slot index)."""
    n = 400_000
    keys = oracle.gen_sm64_mod(7, 0, n, G, key_dtype)
    vals = oracle.gen_sm64_mod(8, 0, n, 1024, np.float64)
    ok, ost = oracle.group_aggregate(keys.astype(np.int64), vals)
    specs = [(N.AGG_SUM, 0), (N.AGG_COUNT, 0), (N.AGG_MIN, 0), (N.AGG_MAX, 0), (N.AGG_AVG, 0)]
    out, counts = ctx.aggregate(specs, [dev(vals)], group_keys=dev(keys), n_groups=G, want_counts=True)
    out, counts = out.cpu().numpy(), counts.cpu().numpy()
    present = counts > 0
    assert np.array_equal(np.nonzero(present)[0], ok)
    assert np.array_equal(out[:, present].T, ost)
    # absent groups keep the identities: sum 0, count 0, min f64::MAX, max f64::MIN, avg 0 (executor.rs:308-311, :339-345)
    if (~present).any():
        a = out[:, ~present]
        assert (a[0] == 0).all() and (a[1] == 0).all() and (a[2] == np.finfo(np.float64).max).all() and (a[3] == -np.finfo(np.float64).max).all() and (a[4] == 0).all()


@pytest.mark.parametrize("G,with_min_max", [(5558, True), (5559, True), (12_969, False), (12_970, False)])
def test_grouped_aggregate_one_full_cu_table(ctx, G, with_min_max):
    """The LDS pre-aggregation takes ONE table per CU of up to 152 KiB: 5558 groups of 28 bytes with MIN / MAX, 12 969 groups of 12 bytes
    without; one group more goes through the partition path (big inputs) -- the same result on both sides of each boundary."""
    n = 5_000_000
    keys = oracle.gen_sm64_mod(91, 0, n, G, np.int32)
    vals = oracle.gen_sm64_mod(92, 0, n, 4096, np.float64) - 2048.0
    ok, ost = oracle.group_aggregate(keys.astype(np.int64), vals)
    specs = [(N.AGG_SUM, 0), (N.AGG_COUNT, 0), (N.AGG_AVG, 0)] + ([(N.AGG_MIN, 0), (N.AGG_MAX, 0)] if with_min_max else [])
    ctx.stats_reset()
    out, counts = ctx.aggregate(specs, [dev(vals)], group_keys=dev(keys), n_groups=G, want_counts=True)
    out, counts = out.cpu().numpy(), counts.cpu().numpy()
    present = counts > 0
    assert np.array_equal(np.nonzero(present)[0], ok)
    assert np.array_equal(out[0, present], ost[:, 0]) and np.array_equal(out[1, present], ost[:, 1]) and np.array_equal(out[2, present], ost[:, 4])
    if with_min_max:
        assert np.array_equal(out[3, present], ost[:, 2]) and np.array_equal(out[4, present], ost[:, 3])
    st = ctx.stats()
    assert ("hash_aggregate_lds" in st) == (G in (5558, 12_969)) and ("agg_partition_scatter" in st) == (G in (5559, 12_970))


@pytest.mark.parametrize("key_dtype", [np.int32, np.int64])
@pytest.mark.parametrize("G", [6000, 100_000, 1_000_003])
def test_grouped_aggregate_partitioned_path(ctx, key_dtype, G):
    """group counts beyond one LDS table on a big input: partition by slot range, then LDS-aggregate (no global atomics);
    must equal the oracle and the global-atomic path (GOLAP_AGG_PART=0)."""
    import os
    n = 5_000_000
    keys = oracle.gen_sm64_mod(71, 0, n, G, key_dtype)
    vals = oracle.gen_sm64_mod(72, 0, n, 4096, np.float64) - 2048.0
    ok, ost = oracle.group_aggregate(keys.astype(np.int64), vals)
    specs = [(N.AGG_SUM, 0), (N.AGG_COUNT, 0), (N.AGG_MIN, 0), (N.AGG_MAX, 0), (N.AGG_AVG, 0)]
    dk, dv = dev(keys), dev(vals)
    ctx.stats_reset()
    out, counts = ctx.aggregate(specs, [dv], group_keys=dk, n_groups=G, want_counts=True)
    out, counts = out.cpu().numpy(), counts.cpu().numpy()
    assert "agg_partition_scatter" in ctx.stats()
    present = counts > 0
    assert np.array_equal(np.nonzero(present)[0], ok) and np.array_equal(out[:, present].T, ost)
    with ctx.knobs(GOLAP_AGG_PART="0"):
        out2 = ctx.aggregate(specs, [dv], group_keys=dk, n_groups=G).cpu().numpy()
    assert np.array_equal(out2, out)


@pytest.mark.parametrize("shape", ["uniform", "skewed"])
def test_grouped_aggregate_big_tile_partition(ctx, shape):
    """>= 33.5M rows: the partition pass uses the 1024-thread register-resident tile (k_agg_scatter_big); uniform keys, and a
    skewed mix (one hot key, a sequential run, a short last chunk) that leaves most buckets of a tile empty."""
    n, G = 40_000_123, 1_000_003
    if shape == "uniform":
        keys = oracle.gen_sm64_mod(81, 0, n, G, np.int32)
    else:
        keys = np.concatenate([np.full(n // 2, 777_777, np.int32), (np.arange(n - n // 2, dtype=np.int64) % G).astype(np.int32)])
    vals = oracle.gen_sm64_mod(82, 0, n, 4096, np.float64) - 2048.0
    ok, ost = oracle.group_aggregate(keys.astype(np.int64), vals)
    specs = [(N.AGG_SUM, 0), (N.AGG_COUNT, 0), (N.AGG_MIN, 0), (N.AGG_MAX, 0), (N.AGG_AVG, 0)]
    ctx.stats_reset()
    out, counts = ctx.aggregate(specs, [dev(vals)], group_keys=dev(keys), n_groups=G, want_counts=True)
    out, counts = out.cpu().numpy(), counts.cpu().numpy()
    assert "agg_partition_scatter" in ctx.stats()
    present = counts > 0
    assert np.array_equal(np.nonzero(present)[0], ok) and np.array_equal(out[:, present].T, ost)


RSV_CASES = ["uniform_i32_f64", "uniform_i64_i64", "hot_key", "hot_key_the_sample_cannot_see", "sequential", "unaligned", "plain_form",
             "min_max_tables", "three_buckets", "narrow_last_bucket"]


@pytest.mark.parametrize("case", RSV_CASES)
def test_grouped_aggregate_reserving_scatter(ctx, case):
    """SUM / COUNT / AVG over > one LDS table of groups on >= 33.5M rows: the partition scatter without a histogram pass
    (k_agg_scatter_rsv: provisioned sub-ranges + reservations) with the histogram path standing by.  Which of the two did the work is
    read from the byte credits; the result must be the oracle's either way.
      hot_key: half the rows in one group -- the key sample sees it, the attempt is not even started;
      hot_key_the_sample_cannot_see: the same skew everywhere EXCEPT at the sampled positions -- a range overflows mid-way, the
        histogram path redoes the pass;
      sequential: sorted keys (every chunk feeds one bucket: the per-tile reservation is one long run);
      unaligned: a key column that is not 16-byte aligned takes the histogram path;  plain_form: the compiler-visible kernel form;
      min_max_tables: MIN / MAX asked for too -- 28-byte table entries, 2048 slots per bucket, 489 buckets;
      three_buckets / narrow_last_bucket: small fan-outs (the fold then runs 40+ work-groups per bucket: slices of the sub-ranges), the
        last bucket much narrower than the others (provisioning and sample go by a FULL bucket's share)."""
    n, G = 40_000_123, 1_000_003   # n % 4 == 3: the last rows go through k_agg_scatter_tail
    G = {"three_buckets": 20_000, "narrow_last_bucket": 100_000}.get(case, G)
    kd, vd = (np.int64, np.int64) if case == "uniform_i64_i64" else (np.int32, np.float64)
    if case in ("hot_key", "hot_key_the_sample_cannot_see"):
        keys = oracle.gen_sm64_mod(83, 0, n, G, kd)
        hot = np.arange(n) % 2 == 0
        if case == "hot_key_the_sample_cannot_see":   # k_agg_sample reads 16-row runs every n // 65536 rows
            stride = n // 65536
            hot &= (np.arange(n) % stride) >= 16
        keys[hot] = 777_777
    elif case == "sequential":
        keys = ((np.arange(n, dtype=np.int64) * G) // n).astype(kd)
    else:
        keys = oracle.gen_sm64_mod(83, 0, n, G, kd)
    vals = (oracle.gen_sm64_mod(84, 0, n, 4096, np.int64) - 2048).astype(vd)
    ok, ost = oracle.group_aggregate(keys.astype(np.int64), vals.astype(np.float64))
    specs = [(N.AGG_SUM, 0), (N.AGG_COUNT, 0), (N.AGG_AVG, 0)] + ([(N.AGG_MIN, 0), (N.AGG_MAX, 0)] if case == "min_max_tables" else [])
    dk, dv = dev(keys), dev(vals)
    if case == "unaligned":
        dk = dev(np.concatenate([keys[:1], keys]))[1:]
    ctx.stats_reset()
    with ctx.knobs(GOLAP_AGG_RSV="plain" if case == "plain_form" else None):
        out, counts = ctx.aggregate(specs, [dv], group_keys=dk, n_groups=G, want_counts=True)
    out, counts = out.cpu().numpy(), counts.cpu().numpy()
    st = ctx.stats()
    present = counts > 0
    assert np.array_equal(np.nonzero(present)[0], ok)
    assert np.array_equal(out[0, present], ost[:, 0]) and np.array_equal(out[1, present], ost[:, 1]) and np.array_equal(out[2, present], ost[:, 4])
    if case == "min_max_tables":
        assert np.array_equal(out[3, present], ost[:, 2]) and np.array_equal(out[4, present], ost[:, 3])
    if case == "unaligned":
        assert "agg_partition_scatter_standby" not in st and st["agg_partition_hist"]["bytes"] > 0
    elif case in ("hot_key", "hot_key_the_sample_cannot_see"):
        assert st["agg_partition_scatter"]["bytes"] == 0 and st["agg_partition_scatter_standby"]["bytes"] > 0
        # (seen by the sample: the attempt returns at once; not seen: it runs into the overflow)
        if case == "hot_key":
            assert st["agg_partition_scatter"]["total_ms"] < 0.5 * st["agg_partition_scatter_standby"]["total_ms"]
    elif case != "sequential":   # (sorted keys: whether every sub-range holds depends on how chunks fall on the XCDs -- either path is right)
        assert st["agg_partition_scatter"]["bytes"] > 0 and st["agg_partition_scatter_standby"]["bytes"] == 0


def test_grouped_two_value_columns_int_and_float(ctx):
    n, G = 200_000, 300
    keys = oracle.gen_sm64_mod(3, 0, n, G, np.int64)
    vi = oracle.gen_sm64_mod(4, 0, n, 5000, np.int64) - 2500
    vf = oracle.gen_sm64_mod(5, 0, n, 64, np.float64)
    out = ctx.aggregate([(N.AGG_SUM, 0), (N.AGG_MAX, 1), (N.AGG_MIN, 0), (N.AGG_AVG, 1)], [dev(vi), dev(vf)], group_keys=dev(keys), n_groups=G).cpu().numpy()
    _, si = oracle.group_aggregate(keys, vi.astype(np.float64))
    _, sf = oracle.group_aggregate(keys, vf)
    assert np.array_equal(out[0], si[:, 0]) and np.array_equal(out[1], sf[:, 3]) and np.array_equal(out[2], si[:, 2]) and np.array_equal(out[3], sf[:, 4])


def test_reference_kernel_mod_semantics(ctx):
    """keys outside [0, n_groups) fold with key % n_groups exactly like join_kernels.cuh:230 (documented defect)."""
    n, G = 50_000, 97
    keys = oracle.gen_sm64_mod(9, 0, n, 10_000, np.int32)
    vi = oracle.gen_sm64_mod(10, 0, n, 100, np.int64)
    exp = np.zeros(G, np.int64)
    oracle.lib().orc_hash_aggregate_kernel_i64(keys.view(np.uint32).ctypes.data_as(oracle.C.POINTER(oracle.C.c_uint32)),
                                               vi.ctypes.data_as(oracle.C.POINTER(oracle.C.c_int64)), n,
                                               exp.ctypes.data_as(oracle.C.POINTER(oracle.C.c_int64)), G)
    out = ctx.aggregate([(N.AGG_SUM, 0)], [dev(vi)], group_keys=dev(keys), n_groups=G).cpu().numpy()[0]
    assert np.array_equal(out, exp.astype(np.float64))


def test_widen_matches_arrow_interop(ctx):
    """record_batch_convert.rs:211-217: Int32 1 widens to i64 1; nulls -> 0 (:102-110)."""
    a = np.array([1, 2, 3, 4, 5], np.int32)
    assert ctx.widen(dev(a)).cpu().numpy().tolist() == [1, 2, 3, 4, 5]
    rng = np.random.default_rng(0)
    n = 100_001
    valid = (rng.random(n) > 0.3).astype(np.uint8)
    for dt in (np.int8, np.int16, np.int32, np.int64, np.uint8):
        x = rng.integers(np.iinfo(dt).min, np.iinfo(dt).max, n, dtype=dt)
        got = ctx.widen(dev(x), validity=dev(valid)).cpu().numpy()
        assert np.array_equal(got, np.where(valid == 1, x.astype(np.int64), 0))
    f = rng.random(n).astype(np.float32)
    got = ctx.widen(dev(f), validity=dev(valid)).cpu().numpy()
    assert np.array_equal(got, np.where(valid == 1, f.astype(np.float64), 0.0))
    for code, dt in ((N.WIDEN_U16, np.uint16), (N.WIDEN_U32, np.uint32)):
        x = rng.integers(0, np.iinfo(dt).max, n, dtype=dt)
        raw = torch.from_numpy(x.view(np.int16 if dt == np.uint16 else np.int32)).cuda()
        got = ctx.widen(raw, in_dtype=code).cpu().numpy()
        assert np.array_equal(got, x.astype(np.int64))


def test_api_constants(ctx):
    from gpu_olap_engine_amd import engine as E
    assert E.choose_strategy(500_000) == N.JOIN_BROADCAST_HASH and E.choose_strategy(5_000_000) == N.JOIN_RADIX_HASH   # join_kernel.rs:85-87
    assert E.compute_launch_params(1024) == (4, 256) and E.compute_launch_params(1) == (1, 256)                     # filter_kernel.rs:87-94
    assert E.agg_identity_value(N.AGG_SUM) == 0 and E.agg_identity_value(N.AGG_MIN) == 2**63 - 1 and E.agg_identity_value(N.AGG_MAX) == -2**63


def _hashed_check(ctx, keys, vals, bound):
    ok, ost = oracle.group_aggregate(keys.astype(np.int64), vals.astype(np.float64))
    specs = [(N.AGG_SUM, 0), (N.AGG_COUNT, 0), (N.AGG_MIN, 0), (N.AGG_MAX, 0), (N.AGG_AVG, 0)]
    out, counts, skeys, nd = ctx.aggregate(specs, [dev(vals)], group_keys=dev(keys), n_groups=bound, want_counts=True, hashed=True)
    nd = int(nd.item())
    assert nd == ok.size
    out, counts, skeys = out.cpu().numpy()[:, :nd], counts.cpu().numpy()[:nd], skeys.cpu().numpy()[:nd]
    order = np.argsort(skeys, kind="stable")
    assert np.array_equal(skeys[order], ok)                      # every distinct key owns exactly one slot
    np.testing.assert_array_equal(out[:, order].T, ost)          # sum, count, min, max, avg per key: bit-exact
    assert np.array_equal(counts[order].astype(np.float64), ost[:, 1])
    return skeys


@pytest.mark.parametrize("key_dtype", [np.int32, np.int64])
@pytest.mark.parametrize("n_distinct", [1, 37, 3000, 150_000])
def test_hashed_group_by_any_keys(ctx, key_dtype, n_distinct):
    """GOLAP_KEYS_HASHED (aggregate_kernel.rs:76-77, the open-addressing table the reference's kernel lacks): negative, sparse and
    congruent keys each get their own group -- what slot = key % n_groups (join_kernels.cuh:230) silently merges."""
    n = 500_000
    rng = np.random.default_rng(n_distinct)
    info = np.iinfo(key_dtype)
    pool = np.unique(np.concatenate([rng.integers(info.min, info.max, n_distinct, dtype=key_dtype),
                                     np.array([0, -1, 1, info.min, info.max, 1024, 2048, -1024], key_dtype)[: max(1, min(8, n_distinct))]]))[:n_distinct]
    keys = pool[rng.integers(0, pool.size, n)]
    vals = rng.integers(-2048, 2048, n).astype(np.float64)
    s1 = _hashed_check(ctx, keys, vals, bound=max(8, 2 * pool.size))
    s2 = _hashed_check(ctx, keys[::-1].copy(), vals[::-1].copy(), bound=max(8, 2 * pool.size))
    # (slot numbering is the key table's order: which of two colliding keys gets the earlier entry depends on who inserts first, so
    #  it is NOT a function of the key set alone -- like SQL, GROUP BY promises no order; the slot_keys column says who is where)
    assert np.array_equal(np.sort(s1), np.sort(s2))


def test_hashed_group_by_reports_a_bound_that_is_too_small(ctx):
    n = 100_000
    keys = np.arange(n, dtype=np.int64) * 7919 - 12345
    vals = np.ones(n)
    out, skeys, nd = ctx.aggregate([(N.AGG_SUM, 0)], [dev(vals)], group_keys=dev(keys), n_groups=1000, hashed=True)
    assert (int(nd.item()) & 0xFFFFFFFF) > 1000                  # (2^32-1 when even the table overflowed): the caller must retry with a larger bound
    _hashed_check(ctx, keys, vals, bound=n)


def test_hashed_group_by_with_a_hopeless_bound_returns_quickly(ctx):
    """12M distinct keys in 40M rows against a bound of 1M (a 2M-entry key table): the table fills up.  Every further row used to scan
    the WHOLE table before giving up -- minutes of GPU time; now the insert pass counts its keys, stops at 7/8 full, and the call
    reports 0xFFFFFFFF at once.  Just above the bound (1.5 x) the distinct count is still exact."""
    import time
    n = 40_000_000
    keys = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=12_000_000, seed=5)
    vals = torch.ones(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t0 = time.time()
    out, skeys, nd = ctx.aggregate([(N.AGG_SUM, 0)], [vals], group_keys=keys, n_groups=1_000_000, hashed=True)
    got = int(nd.item()) & 0xFFFFFFFF
    assert got == 0xFFFFFFFF and time.time() - t0 < 5.0
    keys = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=1_500_000, seed=5)
    out, skeys, nd = ctx.aggregate([(N.AGG_SUM, 0)], [vals], group_keys=keys, n_groups=1_000_000, hashed=True)
    assert (int(nd.item()) & 0xFFFFFFFF) == 1_500_000


def test_hashed_group_by_empty_and_int64_min(ctx):
    keys = np.array([np.iinfo(np.int64).min, 5, np.iinfo(np.int64).min, 5, 9], np.int64)
    _hashed_check(ctx, keys, np.array([1.0, 2.0, 3.0, 4.0, 5.0]), bound=16)
    out, skeys, nd = ctx.aggregate([(N.AGG_SUM, 0)], [torch.empty(0, dtype=torch.float64, device="cuda")],
                                   group_keys=torch.empty(0, dtype=torch.int64, device="cuda"), n_groups=8, hashed=True)
    assert int(nd.item()) == 0


def test_sum_of_general_doubles_is_within_the_stated_bound(ctx):
    """executor.rs:313-327 sums sequentially in row order; the device sums per-block partials folded in a fixed tree order.  For
    doubles that are not integer-valued the two orders round differently: the difference is bounded by n * eps * sum|x| (a loose
    forward bound for any summation order), and in practice far smaller.  Integer-valued data is exact in every order (all other tests)."""
    n = 3_000_017
    rng = np.random.default_rng(3)
    x = rng.random(n) * 1e6 - 3e5
    exp = oracle.execute_aggregate(x)                            # sequential, like the reference
    got = ctx.aggregate([(N.AGG_SUM, 0), (N.AGG_AVG, 0), (N.AGG_MIN, 0), (N.AGG_MAX, 0)], [dev(x)]).cpu().numpy()[:, 0]
    eps = np.finfo(np.float64).eps
    bound = n * eps * float(np.abs(x).sum())
    assert abs(got[0] - exp["sum"]) <= bound and abs(got[0] - exp["sum"]) <= 1e-9 * abs(exp["sum"])
    assert abs(got[1] - exp["avg"]) <= bound / n + eps * abs(exp["avg"])
    assert got[2] == exp["min"] and got[3] == exp["max"]         # order-independent
    again = ctx.aggregate([(N.AGG_SUM, 0)], [dev(x)]).cpu().numpy()[0, 0]
    assert again == got[0]                                       # fixed fold order: reproducible run to run
    # grouped (LDS pre-aggregation + fixed-order fold): same bound per group
    G = 64
    k = rng.integers(0, G, n).astype(np.int32)
    gs = ctx.aggregate([(N.AGG_SUM, 0)], [dev(x)], group_keys=dev(k), n_groups=G).cpu().numpy()[0]
    for g in range(G):
        xs = x[k == g]
        seq = 0.0
        seq = float(np.add.reduce(xs))   # (pairwise in numpy; the sequential sum lies within the same bound)
        assert abs(gs[g] - seq) <= xs.size * eps * float(np.abs(xs).sum())


@pytest.mark.parametrize("shape", ["ungrouped", "lds", "partition", "global", "hashed"])
def test_aggregates_skip_null_values(ctx, shape):
    """SURVEY 8(f) rank 2 for aggregates: a NULL value (validity byte 0, column_buffer.rs:59-60) takes no part in SUM / MIN / MAX /
    AVG and is not counted by COUNT -- on every grouped path (LDS pre-aggregation, partition-then-aggregate, global atomics, hashed
    keys) and ungrouped.  Expected = the oracle on the non-NULL rows; a group whose values are all NULL keeps the identities."""
    n, G = {"ungrouped": (300_007, 1), "lds": (400_003, 100), "partition": (4_300_000, 150_000), "global": (200_000, 150_000),
            "hashed": (500_000, 3000)}[shape]
    rng = np.random.default_rng(len(shape))
    vals = rng.integers(-2048, 2048, n).astype(np.float64)
    ivals = rng.integers(-2048, 2048, n).astype(np.int64)
    vv = (rng.random(n) > 0.3).astype(np.uint8)
    iv = (rng.random(n) > 0.1).astype(np.uint8)
    specs = [(N.AGG_SUM, 0), (N.AGG_COUNT, 0), (N.AGG_MIN, 0), (N.AGG_MAX, 0), (N.AGG_AVG, 0), (N.AGG_SUM, 1), (N.AGG_COUNT, 1)]
    if shape == "ungrouped":
        got = ctx.aggregate(specs, [dev(vals), dev(ivals)], value_valid=[dev(vv), dev(iv)]).cpu().numpy()[:, 0]
        e0, e1 = oracle.execute_aggregate(vals[vv != 0]), oracle.execute_aggregate(ivals[iv != 0].astype(np.float64))
        assert got.tolist() == [e0["sum"], e0["count"], e0["min"], e0["max"], e0["avg"], e1["sum"], e1["count"]]
        allnull = ctx.aggregate(specs[:5], [dev(vals)], value_valid=[dev(np.zeros(n, np.uint8))]).cpu().numpy()[:, 0]
        e = oracle.execute_aggregate(np.empty(0))
        assert allnull.tolist() == [e["sum"], e["count"], e["min"], e["max"], e["avg"]]
        return
    keys = rng.integers(0, G, n).astype(np.int64)
    if shape == "hashed":
        keys = keys * 7919 - 5_000_000
    vv[keys == keys[0]] = 0                                       # one group whose first value column is NULL in every row
    if shape == "hashed":
        out, skeys, nd = ctx.aggregate(specs, [dev(vals), dev(ivals)], group_keys=dev(keys), n_groups=2 * G, hashed=True, value_valid=[dev(vv), dev(iv)])
        nd = int(nd.item())
        sk = skeys.cpu().numpy()[:nd]
        order = np.argsort(sk)
        out = out.cpu().numpy()[:, :nd][:, order]
        present = sk[order]
    else:
        out = ctx.aggregate(specs, [dev(vals), dev(ivals)], group_keys=dev(keys), n_groups=G, value_valid=[dev(vv), dev(iv)]).cpu().numpy()
        present = np.arange(G)
    for col, (x, valid, planes) in enumerate([(vals, vv, [0, 1, 2, 3, 4]), (ivals.astype(np.float64), iv, [5, 6])]):
        ek, est = oracle.group_aggregate(keys[valid != 0], x[valid != 0])
        pos = np.searchsorted(present, ek)
        assert np.array_equal(present[pos], ek)
        exp = np.zeros((present.size, 5))
        exp[:, 2], exp[:, 3] = np.finfo(np.float64).max, -np.finfo(np.float64).max      # identities of groups without a non-NULL value
        exp[pos] = est
        names = ["sum", "count", "min", "max", "avg"]
        for pl, st in zip(planes, [0, 1, 2, 3, 4] if col == 0 else [0, 1]):
            np.testing.assert_array_equal(out[pl], exp[:, st], err_msg=f"{shape}: column {col} {names[st]}")
    g0 = int(np.searchsorted(present, keys[0]))
    assert out[1, g0] == 0.0 and out[0, g0] == 0.0                 # the all-NULL group: COUNT 0, SUM 0


@pytest.mark.parametrize("n", [1, 4095, 4097, 300_001])
def test_filter_compact_and_matches_the_oracle(ctx, n):
    """golap_filter_compact_and through the C ABI: 2 to 4 comparisons on 2 to 3 columns (one column twice: a range), with and without
    validity, every operator incl. an unknown one (selects nothing), against the AND of the oracle's per-comparison masks."""
    rng = np.random.default_rng(n)
    cols = [rng.integers(-50, 50, n).astype(np.int64) for _ in range(3)]
    valid = [None, (rng.random(n) > 0.2).astype(np.uint8), None]
    tc = [torch.from_numpy(c).cuda() for c in cols]
    tv = [None if v is None else torch.from_numpy(v).cuda() for v in valid]
    cases = [([0, 0], [">=", "<"], [-10, 20]), ([0, 1, 2], ["<", "!=", ">"], [30, 0, -40]), ([2, 1, 0, 2], ["<=", ">", "=", "<>"], [45, -45, 3, 7]),
             ([0, 1], ["<", "~"], [10, 10])]
    for which, ops, lits in cases:
        for use_valid in (False, True):
            idx, cnt = ctx.filter_compact_and([tc[w] for w in which], ops, lits, valids=[tv[w] for w in which] if use_valid else None)
            torch.cuda.synchronize()
            m = oracle.eval_filter_mask_and([cols[w] for w in which], ops, lits, [valid[w] for w in which] if use_valid else None)
            k = int(cnt.item())
            assert k == int(m.sum())
            assert np.array_equal(idx[:k].cpu().numpy().view(np.uint32), np.nonzero(m)[0].astype(np.uint32))
