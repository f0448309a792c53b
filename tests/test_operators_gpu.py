"""Operator-level parity: the C++ GpuExecutor (libgolap_host.so) driven through the PhysicalPlan API over
pyarrow RecordBatches, read like the reference's own executor tests (executor.rs:543-570)."""
import numpy as np
import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import oracle  # noqa: E402
from gpu_olap_engine_amd import operators as O  # noqa: E402


def rb(**cols):
    return pa.RecordBatch.from_arrays([pa.array(v) for v in cols.values()], names=list(cols.keys()))


def test_filter_reference_known_answer(ctx):
    """executor.rs:553-570: a=[5,15,25], a > 10 keeps rows 1,2."""
    batch = rb(a=np.array([5, 15, 25], np.int64))
    pred = O.BinaryExpr(O.Column(0, "a"), ">", O.Literal("10"))
    out = O.execute(O.GpuFilter(O.Batches([batch]), pred), ctx)
    assert len(out) == 1 and out[0].column(0).to_pylist() == [15, 25]


def test_filter_multi_batch_multi_column(ctx):
    rng = np.random.default_rng(0)
    batches, exp_k, exp_v, exp_f = [], [], [], []
    for n in (1000, 0, 70_001, 5):
        k = rng.integers(0, 1000, n).astype(np.int64)
        v = rng.integers(-5, 5, n).astype(np.int64)
        f = rng.random(n)
        batches.append(rb(k=k, v=v, f=f))
        m = oracle.eval_filter_mask(k, "<", 500).astype(bool)
        exp_k.append(k[m]); exp_v.append(v[m]); exp_f.append(f[m])
    # column resolved by NAME first even when the index is wrong (executor.rs:379-388)
    out = O.execute(O.GpuFilter(O.Batches(batches), O.BinaryExpr(O.Column(7, "k"), "<", O.Literal("500"))), ctx)
    kept = [(a, b, c) for a, b, c in zip(exp_k, exp_v, exp_f) if len(a)]        # empty results are dropped (:173)
    assert len(out) == len(kept)
    for o, (a, b, c) in zip(out, kept):
        assert o.schema.names == ["k", "v", "f"]
        assert np.array_equal(o.column(0).to_numpy(), a) and np.array_equal(o.column(1).to_numpy(), b) and np.array_equal(o.column(2).to_numpy(), c)


def test_filter_edge_semantics(ctx):
    batch = rb(a=np.arange(10, dtype=np.int64), f=np.arange(10, dtype=np.float64))
    # unknown operator selects no row (executor.rs:464) -> batch dropped
    assert O.execute(O.GpuFilter(O.Batches([batch]), O.BinaryExpr(O.Column(0, "a"), "AND", O.Literal("3"))), ctx) == []
    # non-binary predicate -> all rows pass (:474)
    out = O.execute(O.GpuFilter(O.Batches([batch]), O.Column(0, "a")), ctx)
    assert out[0].num_rows == 10
    # Float64 operand -> not (Int64, Int64) -> all rows pass (:471-472)
    out = O.execute(O.GpuFilter(O.Batches([batch]), O.BinaryExpr(O.Column(1, "f"), ">", O.Literal("3"))), ctx)
    assert out[0].num_rows == 10
    # literal on the left: 3 < a  ==  a > 3
    out = O.execute(O.GpuFilter(O.Batches([batch]), O.BinaryExpr(O.Literal("3"), "<", O.Column(0, "a"))), ctx)
    assert out[0].column(0).to_pylist() == [4, 5, 6, 7, 8, 9]
    with pytest.raises(O.OperatorError):   # unknown column (:386)
        O.execute(O.GpuFilter(O.Batches([batch]), O.BinaryExpr(O.Column(9, "zz"), ">", O.Literal("3"))), ctx)


def test_config1_filter_then_sum_plan(ctx):
    """BASELINE config 1 as the hand-built plan of SURVEY 3.1: GpuAggregate{ GpuFilter{ batch, k < 500 }, [], [Sum(v)] }."""
    n = 1_000_000
    k = oracle.gen_sm64_mod(1, 0, n, 1000)
    v = oracle.gen_sm64_mod(2, 0, n, 1024)
    plan = O.GpuAggregate(O.GpuFilter(O.Batches([rb(k=k, v=v)]), O.BinaryExpr(O.Column(0, "k"), "<", O.Literal("500"))), [], [O.Sum(O.Column(0, "v"))])
    out = O.execute(plan, ctx)
    es, _ = oracle.filter_sum(k, v, "<", 500)
    assert len(out) == 1 and out[0].num_rows == 1 and out[0].column(0).to_pylist() == [es]
    assert out[0].schema.field(0).type == pa.float64()         # executor.rs:347


def test_hash_join_materialises_pairs(ctx):
    """Build side = left, first key only, global row index = running offset over batches (executor.rs:238-253);
    output rows in canonical (probe_rid, build_rid) order, left columns then right columns."""
    rng = np.random.default_rng(1)
    lk = [rng.integers(0, 300, 700).astype(np.int64), rng.integers(0, 300, 301).astype(np.int64)]
    lp = [rng.random(700), rng.random(301)]
    rk = [rng.integers(0, 400, 500).astype(np.int64), rng.integers(0, 400, 99).astype(np.int64), np.empty(0, np.int64)]
    rp = [rng.integers(0, 10**9, 500).astype(np.int64), rng.integers(0, 10**9, 99).astype(np.int64), np.empty(0, np.int64)]
    left = O.Batches([rb(id=a, lpay=b) for a, b in zip(lk, lp)])
    right = O.Batches([rb(fk=a, rpay=b) for a, b in zip(rk, rp)])
    out = O.execute(O.GpuHashJoin(left, right, [O.Column(0, "id")], [O.Column(0, "fk")]), ctx)
    pairs, m = oracle.execute_hash_join(lk, rk)
    L, LP, R, RP = np.concatenate(lk), np.concatenate(lp), np.concatenate(rk), np.concatenate(rp)
    assert len(out) == 1 and out[0].num_rows == m and out[0].schema.names == ["id", "lpay", "fk", "rpay"]
    bi, pi = pairs[:, 0].astype(np.int64), pairs[:, 1].astype(np.int64)
    assert np.array_equal(out[0].column(0).to_numpy(), L[bi]) and np.array_equal(out[0].column(1).to_numpy(), LP[bi])
    assert np.array_equal(out[0].column(2).to_numpy(), R[pi]) and np.array_equal(out[0].column(3).to_numpy(), RP[pi])
    assert np.array_equal(out[0].column(0).to_numpy(), out[0].column(2).to_numpy())


def test_hash_join_edge_cases(ctx):
    one = rb(id=np.array([1, 2, 3], np.int64))
    assert O.execute(O.GpuHashJoin(O.Batches([]), O.Batches([one]), [O.Column(0, "id")], [O.Column(0, "id")]), ctx) == []   # executor.rs:231-233
    assert O.execute(O.GpuHashJoin(O.Batches([one]), O.Batches([]), [O.Column(0, "id")], [O.Column(0, "id")]), ctx) == []
    out = O.execute(O.GpuHashJoin(O.Batches([one]), O.Batches([rb(id=np.array([9], np.int64))]), [O.Column(0, "id")], [O.Column(0, "id")]), ctx)
    assert len(out) == 1 and out[0].num_rows == 0 and out[0].num_columns == 2
    # many-to-many: more pairs than probe rows -> the executor re-runs with the true count (join_kernels.cuh:155-160)
    l = rb(id=np.zeros(50, np.int64), x=np.arange(50, dtype=np.int64))
    r = rb(id=np.zeros(40, np.int64), y=np.arange(40, dtype=np.float64))
    out = O.execute(O.GpuHashJoin(O.Batches([l]), O.Batches([r]), [O.Column(0, "id")], [O.Column(0, "id")]), ctx)
    assert out[0].num_rows == 2000
    assert np.array_equal(out[0].column(1).to_numpy(), np.tile(np.arange(50), 40)) and np.array_equal(out[0].column(3).to_numpy(), np.repeat(np.arange(40.0), 50))
    with pytest.raises(KeyError):
        O.execute(O.GpuHashJoin(O.Batches([l]), O.Batches([r]), [O.Column(0, "id")], [O.Column(0, "id")], join_type="Cross"), ctx)


@pytest.mark.parametrize("jt,code", [("Inner", 0), ("Full", 3)])
def test_hash_join_with_a_build_side_beyond_the_l2_takes_the_radix_join(ctx, jt, code):
    """The operator refines JoinKernel::choose_strategy (join_kernel.rs:71-77, build side only): a build side of more than 65 536 rows goes
    through the radix join even below the reference's 1M-row threshold (its global table would no longer probe out of an XCD's L2).  The
    same pairs in the same canonical order as the reference's restatement, payload and NULLs of the missing side included."""
    rng = np.random.default_rng(31)
    nb, n_p = 150_000, 400_001
    lk = rng.integers(0, 120_000, nb).astype(np.int64)             # duplicates on the build side
    rk = rng.integers(0, 140_000, n_p).astype(np.int64)            # some probe keys match nobody
    lp, rp = rng.random(nb), rng.integers(0, 10**9, n_p).astype(np.int64)
    cut = 90_001
    left = O.Batches([rb(id=lk[:cut], lpay=lp[:cut]), rb(id=lk[cut:], lpay=lp[cut:])])
    right = O.Batches([rb(fk=rk, rpay=rp)])
    out = O.execute(O.GpuHashJoin(left, right, [O.Column(0, "id")], [O.Column(0, "fk")], join_type=jt), ctx)[0]
    pairs, _ = oracle.hash_join_outer(lk, rk, code)
    NULL = 0xFFFFFFFF
    bi, pi = pairs[:, 0].astype(np.int64), pairs[:, 1].astype(np.int64)
    bnull, pnull = bi == NULL, pi == NULL
    assert out.num_rows == len(pairs)
    for col, src, idx, null in ((0, lk, bi, bnull), (1, lp, bi, bnull), (2, rk, pi, pnull), (3, rp, pi, pnull)):
        got = out.column(col)
        assert got.null_count == int(null.sum()) and np.array_equal(np.array(got.is_null()), null)
        assert np.array_equal(got.to_numpy(zero_copy_only=False)[~null], src[idx[~null]])


def test_group_by_over_a_join_with_either_strategy(ctx):
    """GpuAggregate over GpuHashJoin: the groups and their exact sums are numpy's whichever strategy the operator picks for the build side
    -- the radix join (150 000 build rows) or the broadcast join (3 000)."""
    rng = np.random.default_rng(37)
    for n_c in (150_000, 3_000):
        n_o = 600_000
        c_id = rng.permutation(n_c).astype(np.int64)
        c_seg = rng.integers(0, 50, n_c).astype(np.int64)
        o_cust = rng.integers(0, n_c + n_c // 5, n_o).astype(np.int64)
        o_amt = rng.integers(1, 1000, n_o).astype(np.float64)
        plan = O.GpuAggregate(O.GpuHashJoin(O.Batches([rb(id=c_id, seg=c_seg)]), O.Batches([rb(cust=o_cust[:250_000], amt=o_amt[:250_000]), rb(cust=o_cust[250_000:], amt=o_amt[250_000:])]),
                                            [O.Column(0, "id")], [O.Column(0, "cust")]),
                              [O.Column(1, "seg")], [O.Sum(O.Column(3, "amt")), O.Count(O.Column(3, "amt")), O.Min(O.Column(3, "amt")), O.Max(O.Column(3, "amt"))])
        out = O.execute(plan, ctx)[0]
        keep = o_cust < n_c
        seg_of = np.empty(n_c, np.int64)
        seg_of[c_id] = c_seg
        seg = seg_of[o_cust[keep]]
        assert out.column(0).to_pylist() == list(range(50))
        assert np.array_equal(out.column(1).to_numpy(), np.bincount(seg, weights=o_amt[keep], minlength=50))
        assert np.array_equal(out.column(2).to_numpy(), np.bincount(seg, minlength=50).astype(np.float64))
        for g in (0, 17, 49):
            assert out.column(3)[g].as_py() == o_amt[keep][seg == g].min() and out.column(4)[g].as_py() == o_amt[keep][seg == g].max()


def test_hash_join_reports_unrepairable_surrogate_collisions(ctx):
    """A composite-key join whose surrogate keys collide more often than the pair verification can repair (only reachable with the test
    hook that narrows them to a few bits) is an error of the operator, not a result: golap_join's count word reads 2^64 - 1 (golap.h)."""
    rng = np.random.default_rng(3)
    n = 3000
    left = rb(a=np.arange(n, dtype=np.int64) * 2, b=rng.integers(0, 5, n).astype(np.int64))
    right = rb(c=np.arange(n, dtype=np.int64) * 2 + 1, d=rng.integers(0, 5, n).astype(np.int64))
    plan = lambda: O.GpuHashJoin(O.Batches([left]), O.Batches([right]), [O.Column(0, "a"), O.Column(1, "b")], [O.Column(0, "c"), O.Column(1, "d")])
    with ctx.knobs(GOLAP_GJ_HASH_BITS="8"):
        with pytest.raises(O.OperatorError, match="colliding surrogate keys"):
            O.execute(plan(), ctx)
    out = O.execute(plan(), ctx)                                     # without the hook: nothing matches, an empty batch with the joined schema
    assert len(out) == 1 and out[0].num_rows == 0 and out[0].schema.names == ["a", "b", "c", "d"]


@pytest.mark.parametrize("jt,code", [("Left", 1), ("Right", 2), ("Full", 3)])
def test_outer_joins_produce_arrow_nulls(ctx, jt, code):
    """JoinType::{Left,Right,Full} (logical_plan.rs:53-58): the missing side's columns are Arrow nulls."""
    rng = np.random.default_rng(code)
    lk = rng.integers(0, 200, 500).astype(np.int64)
    rk = rng.integers(100, 300, 400).astype(np.int64)
    lp, rp = rng.random(500), rng.integers(0, 10**6, 400).astype(np.int64)
    out = O.execute(O.GpuHashJoin(O.Batches([rb(id=lk, lpay=lp)]), O.Batches([rb(fk=rk, rpay=rp)]), [O.Column(0, "id")], [O.Column(0, "fk")], join_type=jt), ctx)
    pairs, m = oracle.hash_join_outer(lk, rk, code)
    assert out[0].num_rows == m
    bi, pi = pairs[:, 0].astype(np.int64), pairs[:, 1].astype(np.int64)
    bnull, pnull = pairs[:, 0] == 0xFFFFFFFF, pairs[:, 1] == 0xFFFFFFFF
    exp_id = [None if n else int(lk[i]) for i, n in zip(bi, bnull)]
    exp_lp = [None if n else float(lp[i]) for i, n in zip(bi, bnull)]
    exp_fk = [None if n else int(rk[i]) for i, n in zip(pi, pnull)]
    exp_rp = [None if n else int(rp[i]) for i, n in zip(pi, pnull)]
    assert out[0].column(0).to_pylist() == exp_id and out[0].column(1).to_pylist() == exp_lp
    assert out[0].column(2).to_pylist() == exp_fk and out[0].column(3).to_pylist() == exp_rp
    assert out[0].column(0).null_count == int(bnull.sum()) and out[0].column(2).null_count == int(pnull.sum())


def test_aggregate_ungrouped_all_funcs(ctx):
    """execute_aggregate (executor.rs:297-355): one Float64 scalar per AggregateExpr, Int64 `as f64`, over all batches."""
    rng = np.random.default_rng(2)
    vi = [rng.integers(-100, 100, 1000).astype(np.int64), rng.integers(-100, 100, 3).astype(np.int64)]
    vf = [rng.integers(0, 1 << 20, 1000) / 64.0, rng.integers(0, 1 << 20, 3) / 64.0]
    inp = O.Batches([rb(i=a, f=b) for a, b in zip(vi, vf)])
    aggs = [O.Sum(O.Column(0, "i")), O.Count(O.Column(0, "i")), O.Min(O.Column(1, "f")), O.Max(O.Column(1, "f")), O.Avg(O.Column(0, "i"))]
    out = O.execute(O.GpuAggregate(inp, [], aggs), ctx)
    ei, ef = oracle.execute_aggregate(np.concatenate(vi)), oracle.execute_aggregate(np.concatenate(vf))
    assert [c.to_pylist()[0] for c in out[0].columns] == [ei["sum"], ei["count"], ef["min"], ef["max"], ei["avg"]]
    # empty input: sum 0, count 0, min f64::MAX, max f64::MIN, avg 0 (:308-311, :339-345)
    out = O.execute(O.GpuAggregate(O.Batches([rb(i=np.empty(0, np.int64))]), [], [O.Sum(O.Column(0, "i")), O.Min(O.Column(0, "i")), O.Avg(O.Column(0, "i"))]), ctx)
    assert [c.to_pylist()[0] for c in out[0].columns] == [0.0, np.finfo(np.float64).max, 0.0]


def test_aggregate_group_by(ctx):
    """GROUP BY (ignored by the reference executor, executor.rs:307): key column + one Float64 column per aggregate, sorted by key."""
    n = 300_000
    k32 = oracle.gen_sm64_mod(7, 0, n, 1024, np.int32) + 5
    v = oracle.gen_sm64_mod(8, 0, n, 1024, np.float64)
    batches = [rb(g=k32[:100_000], v=v[:100_000]), rb(g=k32[100_000:], v=v[100_000:])]
    out = O.execute(O.GpuAggregate(O.Batches(batches), [O.Column(0, "g")], [O.Sum(O.Column(1, "v")), O.Count(O.Column(1, "v")), O.Max(O.Column(1, "v"))]), ctx)
    ks, st = oracle.group_aggregate(k32.astype(np.int64), v)
    assert out[0].schema.names == ["g", "sum(v)", "count(v)", "max(v)"]
    assert np.array_equal(out[0].column(0).to_numpy(), ks)
    assert np.array_equal(out[0].column(1).to_numpy(), st[:, 0]) and np.array_equal(out[0].column(2).to_numpy(), st[:, 1]) and np.array_equal(out[0].column(3).to_numpy(), st[:, 3])


def test_scan_sort_projection_passthrough(ctx):
    sch = pa.schema([("a", pa.int64()), ("b", pa.float32()), ("c", pa.float64())])
    out = O.execute(O.GpuTableScan("t", sch), ctx)          # executor.rs:118-154: one empty batch
    assert len(out) == 1 and out[0].num_rows == 0 and out[0].schema.names == ["a", "b", "c"]
    assert [f.type for f in out[0].schema] == [pa.int64(), pa.float64(), pa.float64()]
    batch = rb(a=np.array([3, 1, 2], np.int64), b=np.array([1.0, 2.0, 3.0]))
    out = O.execute(O.GpuSort(O.Batches([batch]), [O.Column(0, "a")]), ctx)   # pass-through (:361-370)
    assert out[0].column(0).to_pylist() == [3, 1, 2]
    out = O.execute(O.GpuProjection(O.Batches([batch]), [O.Column(1, "b"), O.Literal("7")]), ctx)
    assert out[0].schema.names == ["b", "lit(7)"] and out[0].column(1).to_pylist() == [7, 7, 7]


def test_filter_null_is_not_true(ctx):
    """SQL semantics on the roadmap of the reference (README.md:411): a comparison with NULL is not true.  The reference's ingest
    turns NULL into 0 (record_batch_convert.rs:102-110), which WOULD pass `a < 10`; here the byte-per-row validity
    (column_buffer.rs:59-60) reaches the kernel and the NULL row is dropped.  NULLs in other columns travel through as Arrow nulls."""
    batch = pa.RecordBatch.from_arrays([pa.array([5, None, 25, 7], pa.int64()), pa.array([1.5, 2.5, None, None], pa.float64())], names=["a", "b"])
    out = O.execute(O.GpuFilter(O.Batches([batch]), O.BinaryExpr(O.Column(0, "a"), "<", O.Literal("10"))), ctx)
    assert out[0].column(0).to_pylist() == [5, 7] and out[0].column(1).to_pylist() == [1.5, None]
    out = O.execute(O.GpuFilter(O.Batches([batch]), O.BinaryExpr(O.Column(0, "a"), "!=", O.Literal("5"))), ctx)
    assert out[0].column(0).to_pylist() == [25, 7]                     # NULL != 5 is not true either
    out = O.execute(O.GpuFilter(O.Batches([batch]), O.BinaryExpr(O.Column(0, "a"), ">=", O.Literal("0"))), ctx)
    assert out[0].column(0).to_pylist() == [5, 25, 7] and out[0].column(1).null_count == 2


def _materialise(pairs, cols_left, cols_right):
    """expected output rows of a join from canonical (build, probe) pairs; None where the side is missing or the source is NULL"""
    rows = []
    for b, p in pairs:
        rows.append(tuple((None if b == 0xFFFFFFFF else c[int(b)]) for c in cols_left) + tuple((None if p == 0xFFFFFFFF else c[int(p)]) for c in cols_right))
    return rows


@pytest.mark.parametrize("jt,code", [("Inner", 0), ("Left", 1), ("Right", 2), ("Full", 3)])
def test_hash_join_null_keys_match_nothing(ctx, jt, code):
    """A NULL key matches nothing -- not key 0 (what the reference's ingest would have made of it), not another NULL -- and an
    outer join keeps the row as unmatched."""
    lk = [1, None, 0, 3, None, 7]
    lp = [10.0, 11.0, 12.0, None, 14.0, 15.0]
    rk = [0, None, 3, 3, 9, None, 1]
    rp = [100, 101, 102, 103, 104, 105, None]
    left = pa.RecordBatch.from_arrays([pa.array(lk, pa.int64()), pa.array(lp, pa.float64())], names=["id", "lpay"])
    right = pa.RecordBatch.from_arrays([pa.array(rk, pa.int64()), pa.array(rp, pa.int64())], names=["fk", "rpay"])
    out = O.execute(O.GpuHashJoin(O.Batches([left]), O.Batches([right]), [O.Column(0, "id")], [O.Column(0, "fk")], join_type=jt), ctx)
    bv, pv = np.array([x is not None for x in lk], np.uint8), np.array([x is not None for x in rk], np.uint8)
    pairs, m = oracle.hash_join_multi([np.array([x or 0 for x in lk])], [np.array([x or 0 for x in rk])], code, [bv], [pv])
    exp = _materialise(pairs, [lk, lp], [rk, rp])
    got = list(zip(*[out[0].column(i).to_pylist() for i in range(4)])) if out and out[0].num_rows else []
    assert got == exp


def test_hash_join_composite_key(ctx):
    """left_keys / right_keys are vectors (physical_plan.rs:35-36); the reference executor reads [0] only (executor.rs:245).  Two key
    columns over two batches per side, with a NULL in the second key column: every column must be equal and non-NULL."""
    rng = np.random.default_rng(4)
    n, m = 3000, 4000
    l0, l1 = rng.integers(0, 40, n), rng.integers(0, 25, n)
    r0, r1 = rng.integers(0, 40, m), rng.integers(0, 25, m)
    l1m = [None if i % 97 == 0 else int(v) for i, v in enumerate(l1)]
    lpay, rpay = rng.random(n), rng.integers(0, 10**6, m)
    mk = lambda a, b, c, names, lo, hi: pa.RecordBatch.from_arrays([pa.array(a[lo:hi], pa.int64()), pa.array(b[lo:hi], pa.int64()), pa.array(c[lo:hi])], names=names)
    left = O.Batches([mk(l0, l1m, lpay, ["a", "b", "lp"], 0, 1000), mk(l0, l1m, lpay, ["a", "b", "lp"], 1000, n)])
    right = O.Batches([mk(r0, list(map(int, r1)), rpay, ["x", "y", "rp"], 0, 2500), mk(r0, list(map(int, r1)), rpay, ["x", "y", "rp"], 2500, m)])
    out = O.execute(O.GpuHashJoin(left, right, [O.Column(0, "a"), O.Column(1, "b")], [O.Column(0, "x"), O.Column(1, "y")]), ctx)
    bv = np.array([x is not None for x in l1m], np.uint8)
    pairs, cnt = oracle.hash_join_multi([l0, np.array([x or 0 for x in l1m])], [r0, r1], 0, [None, bv], None)
    assert out[0].num_rows == cnt
    bi, pi = pairs[:, 0].astype(np.int64), pairs[:, 1].astype(np.int64)
    assert np.array_equal(out[0].column(0).to_numpy(), l0[bi]) and np.array_equal(out[0].column(1).to_numpy(), l1[bi])
    assert np.array_equal(out[0].column(2).to_numpy(), lpay[bi]) and np.array_equal(out[0].column(5).to_numpy(), rpay[pi])
    assert np.array_equal(out[0].column(0).to_numpy(), out[0].column(3).to_numpy()) and np.array_equal(out[0].column(1).to_numpy(), out[0].column(4).to_numpy())


def test_aggregate_group_by_sparse_and_negative_keys(ctx):
    """GROUP BY over keys no dense table can index (negative, 2^40 apart): the operator switches to the key -> slot hash table."""
    n = 200_000
    rng = np.random.default_rng(11)
    pool = np.array([-(1 << 40), -7, 0, 3, 1 << 33, (1 << 62) + 1, np.iinfo(np.int64).min], np.int64)
    k = pool[rng.integers(0, pool.size, n)]
    v = rng.integers(-1000, 1000, n).astype(np.float64)
    out = O.execute(O.GpuAggregate(O.Batches([rb(g=k[:70_000], v=v[:70_000]), rb(g=k[70_000:], v=v[70_000:])]), [O.Column(0, "g")],
                                   [O.Sum(O.Column(1, "v")), O.Count(O.Column(1, "v")), O.Min(O.Column(1, "v")), O.Avg(O.Column(1, "v"))]), ctx)
    ks, st = oracle.group_aggregate(k, v)
    assert np.array_equal(out[0].column(0).to_numpy(), ks)                      # sorted by key
    assert np.array_equal(out[0].column(1).to_numpy(), st[:, 0]) and np.array_equal(out[0].column(2).to_numpy(), st[:, 1])
    assert np.array_equal(out[0].column(3).to_numpy(), st[:, 2]) and np.array_equal(out[0].column(4).to_numpy(), st[:, 4])


def test_aggregates_skip_null_values(ctx):
    """NULLs in an aggregated column (Arrow validity): SUM / MIN / AVG skip them, COUNT counts the others -- ungrouped, with dense
    group keys and with sparse ones; a group whose values are all NULL is still a group (COUNT 0).  The reference zero-fills
    NULLs on ingest (record_batch_convert.rs:102-110) and so counts them and sums zeros; SQL semantics are what its roadmap
    (README.md:411) asks for."""
    n = 50_000
    rng = np.random.default_rng(21)
    v = rng.integers(-500, 500, n).astype(np.float64)
    mask = rng.random(n) < 0.25
    g_dense = rng.integers(0, 40, n).astype(np.int64)
    mask[g_dense == 7] = True                                                  # group 7: every value NULL
    varr = pa.array([None if m else float(x) for x, m in zip(v, mask)], pa.float64())
    for keys in (g_dense, g_dense * (1 << 35) - 9):
        b1 = pa.RecordBatch.from_arrays([pa.array(keys[:20_000]), varr[:20_000]], names=["g", "v"])
        b2 = pa.RecordBatch.from_arrays([pa.array(keys[20_000:]), varr[20_000:]], names=["g", "v"])
        out = O.execute(O.GpuAggregate(O.Batches([b1, b2]), [O.Column(0, "g")],
                                       [O.Sum(O.Column(1, "v")), O.Count(O.Column(1, "v")), O.Min(O.Column(1, "v")), O.Avg(O.Column(1, "v"))]), ctx)
        groups = np.unique(keys)
        assert np.array_equal(out[0].column(0).to_numpy(), groups)              # all 40 groups, also the all-NULL one
        ks, st = oracle.group_aggregate(keys[~mask], v[~mask])
        pos = np.searchsorted(groups, ks)
        exp = np.zeros((groups.size, 5))
        exp[:, 2] = np.finfo(np.float64).max
        exp[pos] = st
        assert np.array_equal(out[0].column(1).to_numpy(), exp[:, 0]) and np.array_equal(out[0].column(2).to_numpy(), exp[:, 1])
        assert np.array_equal(out[0].column(3).to_numpy(), exp[:, 2]) and np.array_equal(out[0].column(4).to_numpy(), exp[:, 4])
    b = pa.RecordBatch.from_arrays([varr], names=["v"])
    out = O.execute(O.GpuAggregate(O.Batches([b]), [], [O.Sum(O.Column(0, "v")), O.Count(O.Column(0, "v")), O.Max(O.Column(0, "v"))]), ctx)
    e = oracle.execute_aggregate(v[~mask])
    assert [out[0].column(i)[0].as_py() for i in range(3)] == [e["sum"], e["count"], e["max"]]


def test_group_by_null_keys_form_one_group(ctx):
    """NULL group keys: one group of their own (SQL), reported last with a NULL key; the other groups are what they would be without
    those rows -- in particular key 0 does not absorb them (the reference's ingest turns a NULL key into 0)."""
    n = 30_000
    rng = np.random.default_rng(33)
    g = rng.integers(0, 12, n).astype(np.int64)
    v = rng.integers(-100, 100, n).astype(np.float64)
    knull = rng.random(n) < 0.2
    vnull = rng.random(n) < 0.1
    garr = pa.array([None if m else int(x) for x, m in zip(g, knull)], pa.int64())
    varr = pa.array([None if m else float(x) for x, m in zip(v, vnull)], pa.float64())
    b1 = pa.RecordBatch.from_arrays([garr[:10_000], varr[:10_000]], names=["g", "v"])
    b2 = pa.RecordBatch.from_arrays([garr[10_000:], varr[10_000:]], names=["g", "v"])
    out = O.execute(O.GpuAggregate(O.Batches([b1, b2]), [O.Column(0, "g")], [O.Sum(O.Column(1, "v")), O.Count(O.Column(1, "v"))]), ctx)
    keys = out[0].column(0).to_pylist()
    assert keys[:-1] == sorted(set(g[~knull].tolist())) and keys[-1] is None
    live = ~knull & ~vnull
    ks, st = oracle.group_aggregate(g[live], v[live])
    assert np.array_equal(np.array(keys[:-1]), ks)
    assert np.array_equal(out[0].column(1).to_numpy()[:-1], st[:, 0]) and np.array_equal(out[0].column(2).to_numpy()[:-1], st[:, 1])
    e = oracle.execute_aggregate(v[knull & ~vnull])
    assert out[0].column(1)[-1].as_py() == e["sum"] and out[0].column(2)[-1].as_py() == e["count"]


def test_group_by_many_aggregates_on_a_fresh_context():
    """A fresh context, a small table, six aggregates over six distinct columns: more than eight same-class slabs are alive at once
    (one per column + keys + outputs); the slab pool's per-class limit is a config field (slab_allocator.rs:25-28), not 8."""
    from gpu_olap_engine_amd.engine import Context
    import torch
    torch.cuda.set_device(0)
    with Context(max_gpu_memory=0) as c:
        n = 1000
        rng = np.random.default_rng(5)
        cols = {f"c{i}": rng.integers(0, 100, n).astype(np.float64) for i in range(6)}
        g = rng.integers(0, 10, n).astype(np.int64)
        aggs = [O.Sum(O.Column(i + 1, f"c{i}")) for i in range(6)]
        out = O.execute(O.GpuAggregate(O.Batches([rb(g=g, **cols)]), [O.Column(0, "g")], aggs), c)
        for i in range(6):
            ks, st = oracle.group_aggregate(g, cols[f"c{i}"])
            assert np.array_equal(out[0].column(0).to_numpy(), ks) and np.array_equal(out[0].column(i + 1).to_numpy(), st[:, 0])
        out = O.execute(O.GpuAggregate(O.Batches([rb(**cols, c6=cols["c0"])]), [], aggs + [O.Sum(O.Column(6, "c6"))]), c)   # 7 ungrouped value columns
        assert out[0].column(6).to_pylist()[0] == float(cols["c0"].sum())


def _group_by_expected(keys_cols, key_nulls, v, vnull):
    """pure-Python GROUP BY over tuples (None = NULL), NULLS LAST per column: (sorted key tuples, sum, count, min, max) over non-NULL values"""
    groups = {}
    n = len(v)
    for i in range(n):
        k = tuple(None if key_nulls[c][i] else int(keys_cols[c][i]) for c in range(len(keys_cols)))
        g = groups.setdefault(k, [0.0, 0, np.inf, -np.inf])
        if not vnull[i]:
            g[0] += v[i]; g[1] += 1; g[2] = min(g[2], v[i]); g[3] = max(g[3], v[i])
    order = sorted(groups, key=lambda t: tuple((x is None, 0 if x is None else x) for x in t))
    return order, [groups[k] for k in order]


@pytest.mark.parametrize("with_nulls", [False, True])
def test_group_by_several_columns(ctx, with_nulls):
    """GpuAggregate.group_by is a Vec<PhysicalExpr> (physical_plan.rs:52-54): several key columns -- Int64 and Int32, negative values,
    NULLs in the keys ((NULL, 1) is its own group, NULLS LAST per column) and in the values -- against a pure-Python GROUP BY over
    key tuples; rows come out sorted by (k0, k1, k2)."""
    n = 40_000
    rng = np.random.default_rng(77)
    a = rng.integers(-3, 4, n).astype(np.int64)
    b = rng.integers(100, 106, n).astype(np.int32)
    c = (rng.integers(0, 3, n) * 1_000_000_007 - 5).astype(np.int64)          # three far-apart values: a wide range, few groups
    v = rng.integers(-50, 50, n).astype(np.float64)
    kn = [rng.random(n) < (0.1 if with_nulls else 0.0) for _ in range(3)]
    vn = rng.random(n) < (0.1 if with_nulls else 0.0)
    arr = lambda x, m, t: pa.array([None if mm else xx.item() for xx, mm in zip(x, m)], t)
    cols = [arr(a, kn[0], pa.int64()), arr(b, kn[1], pa.int32()), arr(c, kn[2], pa.int64()), arr(v, vn, pa.float64())]
    b1 = pa.RecordBatch.from_arrays([x[:15_000] for x in cols], names=["a", "b", "c", "v"])
    b2 = pa.RecordBatch.from_arrays([x[15_000:] for x in cols], names=["a", "b", "c", "v"])
    plan = O.GpuAggregate(O.Batches([b1, b2]), [O.Column(0, "a"), O.Column(1, "b"), O.Column(2, "c")],
                          [O.Sum(O.Column(3, "v")), O.Count(O.Column(3, "v")), O.Min(O.Column(3, "v")), O.Max(O.Column(3, "v"))])
    out = O.execute(plan, ctx)[0]
    order, stats = _group_by_expected([a, b, c], kn, v, vn)
    got_keys = list(zip(out.column(0).to_pylist(), out.column(1).to_pylist(), out.column(2).to_pylist()))
    assert got_keys == order
    assert out.schema.names[:3] == ["a", "b", "c"]
    live = [i for i, s in enumerate(stats) if s[1] > 0]
    assert out.column(3).to_pylist() == [s[0] for s in stats] and out.column(4).to_pylist() == [float(s[1]) for s in stats]
    assert [out.column(5)[i].as_py() for i in live] == [stats[i][2] for i in live] and [out.column(6)[i].as_py() for i in live] == [stats[i][3] for i in live]


_MASK = (1 << 64) - 1


def _mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


def _surrogate(tup, seed):
    """csrc/golap_ingest.hip k_hash_key_columns restated: the seeded 64-bit surrogate of a key tuple (None = NULL)"""
    h = _mix64(seed ^ 0x243F6A8885A308D3)
    for c, v in enumerate(tup):
        x = (0x9E3779B97F4A7C15 * (c + 1)) & _MASK if v is None else v & _MASK
        h = (_mix64(h ^ x) + (0xD1B54A32D192ED03 if v is None else 0x8CB92BA72F3D8DD7) * (c + 1)) & _MASK
    return h


_SEEDS = [0x6A09E667F3BCC908, 0xBB67AE8584CAA73B, 0x3C6EF372FE94F82B]      # the executor's, in the order it tries them


def _wide_batches(rng, tuples, n, with_nulls):
    """n rows drawn from `tuples` (two int64 key columns, None = NULL) + a value column with some NULLs, as two batches"""
    pick = rng.integers(0, len(tuples), n)
    a = [tuples[i][0] for i in pick]
    b = [tuples[i][1] for i in pick]
    v = rng.integers(-50, 50, n).astype(np.float64)
    vn = rng.random(n) < (0.1 if with_nulls else 0.0)
    cols = [pa.array(a, pa.int64()), pa.array(b, pa.int64()), pa.array([None if m else float(x) for x, m in zip(v, vn)], pa.float64())]
    cut = n // 3
    bs = [pa.RecordBatch.from_arrays([x[:cut] for x in cols], names=["a", "b", "v"]), pa.RecordBatch.from_arrays([x[cut:] for x in cols], names=["a", "b", "v"])]
    kn = [np.array([x is None for x in a]), np.array([x is None for x in b])]
    return bs, [np.array([0 if x is None else x for x in a], dtype=object), np.array([0 if x is None else x for x in b], dtype=object)], kn, v, vn


def _check_grouped(out, keys, kn, v, vn):
    order, stats = _group_by_expected(keys, kn, v, vn)
    assert list(zip(out.column(0).to_pylist(), out.column(1).to_pylist())) == order
    assert out.column(2).to_pylist() == [s[0] for s in stats] and out.column(3).to_pylist() == [float(s[1]) for s in stats]


@pytest.mark.parametrize("with_nulls", [False, True])
def test_group_by_composite_keys_beyond_62_bits(ctx, with_nulls):
    """physical_plan.rs:52-54 puts no bound on the group-by columns: two hash-like 64-bit columns (the mixed-radix key would need ~126
    bits; round 3 refused them) are grouped on a 64-bit surrogate of the tuple and every row is then checked against its group's
    representative row on the device -- exact or an error, never a merged group.  Against a pure-Python GROUP BY over the tuples."""
    rng = np.random.default_rng(5 + with_nulls)
    tuples = [(int(x), int(y)) for x, y in zip(rng.integers(-2**62, 2**62, 700), rng.integers(-2**62, 2**62, 700))]
    if with_nulls:
        tuples += [(None, tuples[3][1]), (tuples[3][0], None), (None, None), (None, tuples[9][1])]
    bs, keys, kn, v, vn = _wide_batches(rng, tuples, 60_000, with_nulls)
    out = O.execute(O.GpuAggregate(O.Batches(bs), [O.Column(0, "a"), O.Column(1, "b")], [O.Sum(O.Column(2, "v")), O.Count(O.Column(2, "v"))]), ctx)[0]
    _check_grouped(out, keys, kn, v, vn)


def test_group_by_one_full_range_column_with_nulls(ctx):
    """ONE key column whose values span (almost) all of int64, with NULLs: value - min + a NULL digit does not fit 62 bits either --
    the same surrogate route (the column without NULLs would take the plain hashed GROUP BY)."""
    rng = np.random.default_rng(8)
    pool = [np.iinfo(np.int64).min, np.iinfo(np.int64).max, -1, 0, 1, 1 << 62, -(1 << 62), None] + [int(x) for x in rng.integers(-2**63, 2**63 - 1, 50)]
    n = 30_000
    pick = rng.integers(0, len(pool), n)
    g = [pool[i] for i in pick]
    v = rng.integers(0, 100, n).astype(np.float64)
    rb = pa.RecordBatch.from_arrays([pa.array(g, pa.int64()), pa.array(v)], names=["g", "v"])
    out = O.execute(O.GpuAggregate(O.Batches([rb]), [O.Column(0, "g")], [O.Sum(O.Column(1, "v")), O.Count(O.Column(1, "v"))]), ctx)[0]
    order, stats = _group_by_expected([np.array([0 if x is None else x for x in g], dtype=object)], [np.array([x is None for x in g])], v, np.zeros(n, bool))
    assert out.column(0).to_pylist() == [k[0] for k in order] and out.column(0).to_pylist()[-1] is None
    assert out.column(1).to_pylist() == [s[0] for s in stats] and out.column(2).to_pylist() == [float(s[1]) for s in stats]


def test_group_by_surrogate_collisions_are_caught_and_reseeded(ctx):
    """With the surrogates narrowed (GOLAP_SURROGATE_BITS) two different tuples DO share one.  A set of tuples is searched (on the CPU,
    with the kernel's hash restated above) for which the executor's first seed collides and its second does not: the verify pass must
    see the first, the second attempt must deliver the exact groups.  With 3 bits every seed collides: an error, not a merged group."""
    bits = 11
    for trial in range(200):
        rng = np.random.default_rng(1000 + trial)
        tuples = [(int(x), int(y)) for x, y in zip(rng.integers(-2**62, 2**62, 60), rng.integers(-2**62, 2**62, 60))]
        dup = [len({_surrogate(t, s) & ((1 << bits) - 1) for t in tuples}) < len(tuples) for s in _SEEDS]
        if dup[0] and not dup[1]:
            break
    else:
        pytest.fail("no tuple set found whose first seed collides and second does not")
    bs, keys, kn, v, vn = _wide_batches(rng, tuples, 20_000, False)
    plan = lambda: O.GpuAggregate(O.Batches(bs), [O.Column(0, "a"), O.Column(1, "b")], [O.Sum(O.Column(2, "v")), O.Count(O.Column(2, "v"))])
    with ctx.knobs(GOLAP_SURROGATE_BITS=str(bits)):
        out = O.execute(plan(), ctx)[0]
    _check_grouped(out, keys, kn, v, vn)
    with ctx.knobs(GOLAP_SURROGATE_BITS="3"):
        with pytest.raises(O.OperatorError) as e:
            O.execute(plan(), ctx)
    assert "collided" in str(e.value)


def test_group_by_all_null_keys_is_one_null_group(ctx):
    """every key NULL: one group with a NULL key (not an error)"""
    g = pa.array([None] * 5, pa.int64())
    v = pa.array([1.0, 2.0, 3.0, 4.0, 5.0])
    out = O.execute(O.GpuAggregate(O.Batches([pa.RecordBatch.from_arrays([g, v], names=["g", "v"])]), [O.Column(0, "g")], [O.Sum(O.Column(1, "v"))]), ctx)[0]
    assert out.num_rows == 1 and out.column(0).to_pylist() == [None] and out.column(1).to_pylist() == [15.0]


def test_group_by_ten_million_distinct_sparse_keys(ctx):
    """the hashed GROUP BY beyond round 2's 2^22-group bound (aggregate_kernel.rs:76-77 promises a global table without one): 10M
    distinct sparse keys over 30M rows through the C ABI, checked against torch.unique + index_add on the device."""
    import torch
    from gpu_olap_engine_amd import _native as N
    n, d = 30_000_000, 10_000_000
    ids = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=d, seed=91)
    keys = ids * 1_000_003 - 7_000_000_000_000                                   # sparse, negative and positive
    vals = ctx.generate(N.GEN_SM64_MOD_F64, n, modulus=1024, seed=92)
    out, out_keys, n_groups = ctx.aggregate([(N.AGG_SUM, 0), (N.AGG_COUNT, 0)], [vals], group_keys=keys, n_groups=d + 16, hashed=True)
    torch.cuda.synchronize()
    g = int(n_groups.item())
    uk, inv = torch.unique(keys, return_inverse=True)
    assert g == uk.numel()
    exp_sum = torch.zeros(g, dtype=torch.float64, device="cuda").index_add_(0, inv, vals)
    exp_cnt = torch.zeros(g, dtype=torch.float64, device="cuda").index_add_(0, inv, torch.ones_like(vals))
    order = torch.argsort(out_keys[:g])
    assert torch.equal(out_keys[:g][order], uk)
    assert torch.equal(out[0][:g][order], exp_sum) and torch.equal(out[1][:g][order], exp_cnt)


def test_filter_conjunction(ctx):
    """`a AND b (AND c)`: what optimizer.rs:155-159 builds from stacked filters.  SQL semantics on the device in one pass (the reference's
    CPU mask has no AND and would select nothing, executor.rs:457-465 -- an extension): every leaf keeps the reference's edge cases,
    NULL operands make a comparison not true, a range on one column reads it once."""
    n = 50_000
    rng = np.random.default_rng(8)
    a = rng.integers(-100, 100, n).astype(np.int64)
    b = rng.integers(0, 1000, n).astype(np.int64)
    f = rng.random(n)
    an = rng.random(n) < 0.1
    aarr = pa.array([None if m else int(x) for x, m in zip(a, an)], pa.int64())
    rb = pa.RecordBatch.from_arrays([aarr, pa.array(b), pa.array(f)], names=["a", "b", "f"])
    A, B, F = O.Column(0, "a"), O.Column(1, "b"), O.Column(2, "f")
    And = lambda l, r: O.BinaryExpr(l, "AND", r)
    # range on one column AND a comparison on another
    out = O.execute(O.GpuFilter(O.Batches([rb]), And(And(O.BinaryExpr(A, ">=", O.Literal("-10")), O.BinaryExpr(A, "<", O.Literal("25"))),
                                                     O.BinaryExpr(O.Literal("500"), ">", B))), ctx)
    keep = ~an & (a >= -10) & (a < 25) & (b < 500)
    assert out[0].num_rows == int(keep.sum())
    assert out[0].column(0).to_pylist() == a[keep].tolist() and out[0].column(1).to_pylist() == b[keep].tolist()
    assert np.array_equal(out[0].column(2).to_numpy(), f[keep])
    # a leaf whose operands are not both Int64 passes every row (executor.rs:449-452): the conjunction is the other leaf alone
    out = O.execute(O.GpuFilter(O.Batches([rb]), And(O.BinaryExpr(F, ">", O.Literal("0")), O.BinaryExpr(B, "=", O.Literal("7")))), ctx)
    assert out[0].num_rows == int((b == 7).sum())
    # a leaf with an unknown comparison selects nothing (executor.rs:464), and so does the conjunction
    out = O.execute(O.GpuFilter(O.Batches([rb]), And(O.BinaryExpr(A, "~", O.Literal("1")), O.BinaryExpr(B, "<", O.Literal("500")))), ctx)
    assert sum(x.num_rows for x in out) == 0


@pytest.mark.parametrize("pipelined", [0, 1])
def test_filter_batches_pipelined_and_in_one_phase(ctx, pipelined):
    """GpuFilter over several batches: by default all batches go up in one transfer phase first; with golap_host_set_pipelining(1) batch k
    is filtered while batch k + 1 is uploaded and every kept-row count comes back in ONE download behind the last batch.  The same rows either
    way: batches that keep nothing or are empty are dropped (executor.rs:173), NULL predicate operands do not pass, narrow types, strings
    (as their hashes) and validity bitmaps that start mid-byte travel with their rows."""
    rng = np.random.default_rng(23)
    L = O.load_host()
    hashes = {f"s{i}": oracle.fnv1a_hash(f"s{i}".encode()) for i in range(37)}
    batches, exp = [], []
    for bi, n in enumerate((50_000, 0, 1, 131_073, 4_097, 20_000)):
        k = rng.integers(0, 1000, n).astype(np.int64)
        if bi == 4:
            k += 5000                                                   # this batch keeps no row
        knull = rng.random(n) < 0.1
        v32 = rng.integers(-1000, 1000, n).astype(np.int32)
        f = rng.random(n)
        fnull = rng.random(n) < 0.2
        s = [f"s{int(x) % 37}" for x in k]
        cols = [pa.array(k, mask=knull), pa.array(v32), pa.array(f, mask=fnull), pa.array(s, pa.string())]
        b = pa.RecordBatch.from_arrays(cols, names=["k", "v", "f", "s"])
        if n > 100:
            b = b.slice(3, n - 7)                                        # bitmaps and values from an offset that is not a multiple of 8
            k, knull, v32, f, fnull, s = k[3:n - 4], knull[3:n - 4], v32[3:n - 4], f[3:n - 4], fnull[3:n - 4], s[3:n - 4]
        batches.append(b)
        m = ~knull & (k >= 100) & (k < 700)
        if m.any():
            exp.append((k[m], v32[m].astype(np.int64), f[m], fnull[m], [x for x, keep in zip(s, m) if keep]))
    pred = O.BinaryExpr(O.BinaryExpr(O.Column(0, "k"), ">=", O.Literal("100")), "AND", O.BinaryExpr(O.Column(0, "k"), "<", O.Literal("700")))
    L.golap_host_set_pipelining(pipelined)
    try:
        prof = {}
        out = O.execute(O.GpuFilter(O.Batches(batches), pred), ctx, profile=prof)
    finally:
        L.golap_host_set_pipelining(0)
    assert len(out) == len(exp)
    for o, (k, v, f, fnull, s) in zip(out, exp):
        assert o.schema.names == ["k", "v", "f", "s"] and o.num_rows == len(k)
        assert o.column(0).null_count == 0 and np.array_equal(o.column(0).to_numpy(), k)
        assert np.array_equal(o.column(1).to_numpy(), v)
        got_f = o.column(2)
        assert got_f.null_count == int(fnull.sum())
        assert np.array_equal(np.array(got_f.is_null()), fnull) and np.array_equal(got_f.to_numpy(zero_copy_only=False)[~fnull], f[~fnull])
        assert np.array_equal(o.column(3).to_numpy().view(np.uint64), np.array([hashes[x] for x in s], np.uint64))
    up = sum(b.nbytes for b in batches if b.num_rows)
    assert 0.9 * up < prof["h2d_bytes"] <= 1.01 * up + 65536, prof     # every buffer went up once, whichever way


def test_utf8_hash_known_answers(ctx):
    """golap_hash_utf8 = the reference's fnv1a_hash (record_batch_convert.rs:123-130) on the device: the published FNV-1a 64 test vectors,
    the oracle's restatement on random strings (incl. empty and multi-byte ones), NULL -> 0 (:112-121)."""
    import torch
    strs = ["", "a", "foobar", "hello world", "ünïcödé", "x" * 1000, None, "b"]
    rng = np.random.default_rng(2)
    strs += ["".join(chr(int(c)) for c in rng.integers(32, 127, int(l))) for l in rng.integers(0, 40, 500)]
    enc = [b"" if s is None else s.encode() for s in strs]
    off = np.zeros(len(strs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(e) for e in enc])
    data = np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8)
    valid = np.array([0 if s is None else 1 for s in strs], dtype=np.uint8)
    got = ctx.hash_utf8(torch.from_numpy(off).cuda(), torch.from_numpy(data.copy()).cuda(), torch.from_numpy(valid).cuda()).cpu().numpy().view(np.uint64)
    assert got[0] == 0xcbf29ce484222325 and got[1] == 0xaf63dc4c8601ec8c and got[2] == 0x85944171f73967e8     # FNV-1a 64 test vectors
    exp = np.array([0 if s is None else oracle.fnv1a_hash(e) for s, e in zip(strs, enc)], dtype=np.uint64)
    assert np.array_equal(got, exp)


def test_string_keys_join_and_group_by(ctx):
    """Utf8 key columns (record_batch_convert.rs:93-97: a string column reaches the GPU as its FNV-1a 64 hashes): a join and a GROUP BY
    on string keys against plain Python over the strings -- equal unless two different strings share a 64-bit hash (none here)."""
    rng = np.random.default_rng(12)
    names = [f"customer-{i:05d}" for i in range(300)]
    lk = [names[i] for i in rng.integers(0, 300, 2000)]
    rk = [names[i] for i in rng.integers(0, 350, 1500) if i < 300] + [None, "nobody"]
    lv = rng.integers(0, 100, len(lk)).astype(np.int64)
    rv = rng.integers(0, 100, len(rk)).astype(np.int64)
    left = pa.RecordBatch.from_arrays([pa.array(lk, pa.string()), pa.array(lv)], names=["name", "lv"])
    right = pa.RecordBatch.from_arrays([pa.array(rk, pa.large_string()), pa.array(rv)], names=["rname", "rv"])
    out = O.execute(O.GpuHashJoin(O.Batches([left]), O.Batches([right]), [O.Column(0, "name")], [O.Column(0, "rname")], "Inner"), ctx)
    got = sorted(zip(out[0].column("lv").to_pylist(), out[0].column("rv").to_pylist(), out[0].column("name").to_pylist()))
    h = lambda s: int(np.uint64(oracle.fnv1a_hash(s.encode())).astype(np.int64))
    exp = sorted((int(a), int(b), h(x)) for x, a in zip(lk, lv) for y, b in zip(rk, rv) if y is not None and x == y)
    assert got == exp and len(exp) > 1000
    g = O.execute(O.GpuAggregate(O.Batches([left]), [O.Column(0, "name")], [O.Sum(O.Column(1, "lv")), O.Count(O.Column(1, "lv"))]), ctx)[0]
    sums = {}
    for x, a in zip(lk, lv):
        s = sums.setdefault(h(x), [0.0, 0.0]); s[0] += a; s[1] += 1
    assert dict(zip(g.column(0).to_pylist(), zip(g.column(1).to_pylist(), g.column(2).to_pylist()))) == {k: tuple(v) for k, v in sums.items()}


def test_narrow_types_offsets_and_bitmaps_are_ingested_on_the_device(ctx):
    """record_batch_convert.rs:35-110 on the device: Int8 ... UInt32 / Float32 columns are uploaded at their own width and widened there
    (int -> Int64, Float32 -> Float64), validity travels as the Arrow BITMAP (sliced batches: a bit offset that is not a multiple of 8)
    and is expanded there, NULL slots hold 0 (:102-110).  Filter keeps rows with k < 50; every payload column comes back widened."""
    rng = np.random.default_rng(41)
    n, off = 10_007, 13
    k = rng.integers(0, 100, n).astype(np.int64)
    cols = {"i8": (rng.integers(-128, 128, n).astype(np.int8), pa.int8()), "i16": (rng.integers(-30000, 30000, n).astype(np.int16), pa.int16()),
            "i32": (rng.integers(-2**31, 2**31, n).astype(np.int32), pa.int32()), "u8": (rng.integers(0, 256, n).astype(np.uint8), pa.uint8()),
            "u16": (rng.integers(0, 65536, n).astype(np.uint16), pa.uint16()), "u32": (rng.integers(0, 2**32, n).astype(np.uint32), pa.uint32()),
            "f32": (rng.random(n).astype(np.float32), pa.float32()), "f64": (rng.random(n), pa.float64())}
    nulls = {nm: rng.random(n) < 0.15 for nm in cols}
    arrays = [pa.array(k)] + [pa.array([None if m else x.item() for x, m in zip(v, nulls[nm])], t) for nm, (v, t) in cols.items()]
    full = pa.RecordBatch.from_arrays(arrays, names=["k"] + list(cols))
    batch = full.slice(off, n - off - 3)                                       # offset 13: bitmaps start in the middle of a byte
    out = O.execute(O.GpuFilter(O.Batches([batch]), O.BinaryExpr(O.Column(0, "k"), "<", O.Literal("50"))), ctx)[0]
    sel = np.nonzero(k[off:n - 3] < 50)[0] + off
    assert out.num_rows == sel.size and out.column(0).to_pylist() == k[sel].tolist()
    for j, (nm, (v, t)) in enumerate(cols.items(), start=1):
        want_type = pa.float64() if nm in ("f32", "f64") else pa.int64()
        assert out.schema.field(j).type == want_type, nm
        exp = [None if nulls[nm][i] else (float(v[i]) if nm in ("f32", "f64") else int(v[i])) for i in sel]
        assert out.column(j).to_pylist() == exp, nm
        raw = np.frombuffer(out.column(j).buffers()[1], dtype=np.float64 if nm in ("f32", "f64") else np.int64, count=sel.size)
        assert not raw[nulls[nm][sel]].any(), nm                                 # null -> 0 in the value buffer


def test_filter_join_aggregate_plan_stays_on_the_device(ctx):
    """GpuAggregate{ GpuHashJoin{ GpuFilter{orders}, customers } } over multi-batch inputs with NULLs: every input column is uploaded
    ONCE (the profile's H2D bytes are the Arrow buffers' bytes), nothing but counts and the result comes back, the answer is numpy's."""
    rng = np.random.default_rng(17)
    n_o, n_c = 400_000, 5_000
    o_cust = rng.integers(0, n_c + 500, n_o).astype(np.int64)                   # some orders reference nobody
    o_amt = rng.integers(1, 1000, n_o).astype(np.float64)
    o_flag = rng.integers(0, 10, n_o).astype(np.int64)
    amt_null = rng.random(n_o) < 0.05
    c_id = rng.permutation(n_c).astype(np.int64)
    c_seg = rng.integers(0, 7, n_c).astype(np.int32)
    amt = pa.array([None if m else x for x, m in zip(o_amt.tolist(), amt_null)], pa.float64())
    cut = 150_000
    orders = [pa.RecordBatch.from_arrays([pa.array(o_cust[:cut]), amt[:cut], pa.array(o_flag[:cut])], names=["cust", "amt", "flag"]),
              pa.RecordBatch.from_arrays([pa.array(o_cust[cut:]), amt[cut:], pa.array(o_flag[cut:])], names=["cust", "amt", "flag"])]
    customers = [pa.RecordBatch.from_arrays([pa.array(c_id), pa.array(c_seg)], names=["id", "seg"])]
    plan = O.GpuAggregate(O.GpuHashJoin(O.Batches(customers), O.GpuFilter(O.Batches(orders), O.BinaryExpr(O.Column(2, "flag"), "<", O.Literal("5"))),
                                        [O.Column(0, "id")], [O.Column(0, "cust")]),
                          [O.Column(1, "seg")], [O.Sum(O.Column(3, "amt")), O.Count(O.Column(3, "amt"))])
    prof = {}
    out = O.execute(plan, ctx, profile=prof)[0]
    keep = (o_flag < 5) & (o_cust < n_c)
    seg_of = np.empty(n_c, np.int64)
    seg_of[c_id] = c_seg
    seg = seg_of[o_cust[keep]]
    live = ~amt_null[keep]
    exp_sum = np.bincount(seg[live], weights=o_amt[keep][live], minlength=7)
    exp_cnt = np.bincount(seg[live], minlength=7).astype(np.float64)
    assert out.column(0).to_pylist() == list(range(7))
    assert np.array_equal(out.column(1).to_numpy(), exp_sum) and np.array_equal(out.column(2).to_numpy(), exp_cnt)
    arrow_bytes = n_o * 8 * 3 + (n_o + 7) // 8 + n_c * 8 + n_c * 4
    assert arrow_bytes <= prof["h2d_bytes"] <= arrow_bytes + 4096, prof       # each column once, at its Arrow width, validity as a bitmap
    assert prof["d2h_bytes"] < 64 * 1024, prof                                 # counts, key ranges, a 7-row group table -- no row-id list, no column
    assert prof["total_ms"] > 0 and prof["h2d_ms"] > 0


def test_bench_operator_plan_line():
    """bench.py --config OP: the Arrow in -> Arrow out leg at a small size -- the line verifies its own result against numpy and says where
    the wall time went (each input column uploaded exactly once: h2d_bytes == the Arrow buffers' bytes)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "OP", "--rows", "3000000", "--steps", "2", "--warmup", "1", "--cpu-sample", "200000"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout + out.stderr)[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["verify"]["groups"] > 0 and d["h2d_bytes"] == d["input_bytes"] and d["d2h_bytes"] < 1 << 20
    assert d["end_to_end_ms"] >= d["h2d_ms"] > 0 and d["cpu_baseline"]["kind"] == "port"
