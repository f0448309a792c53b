"""The operator layer on the GPU against Apache Arrow C++ on the CPU, with no oracle in between: the same pyarrow RecordBatches go
through golap_plan_* / golap_execute (HIP kernels) and through pyarrow.compute / Acero (Arrow's own CPU filter, hash join and hash
aggregate), and the results must be the same tables.  The reference's CPU path is Arrow-based (arrow-rs, executor.rs) but returns no
join rows and ignores group_by (:255-264, :307) -- Arrow C++ is the closest runnable "CPU Arrow path" there is.  Integer-valued doubles:
every sum is exact in any order, so equality is bit for bit."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from gpu_olap_engine_amd import operators as O  # noqa: E402

ARROW_JOIN = {"Inner": "inner", "Left": "left outer", "Right": "right outer", "Full": "full outer"}


def nullable(values, rate, rng, typ):
    mask = rng.random(len(values)) < rate
    return pa.array(values, type=typ, mask=mask)


def batches_of(table, cuts):
    """the table as RecordBatches cut at `cuts`: slices (non-zero offsets into the same buffers), an empty batch where two cuts coincide"""
    edges = [0] + list(cuts) + [table.num_rows]
    out = []
    for a, b in zip(edges[:-1], edges[1:]):
        bs = table.slice(a, b - a).to_batches()
        out.append(bs[0] if bs else pa.RecordBatch.from_pylist([], schema=table.schema))
    return out


def sort_all(t):
    return t.sort_by([(c, "ascending") for c in t.column_names])


def same_tables(a, b):
    """the same column names, types, values and NULLs (field flags such as `nullable` are not part of the comparison)"""
    return a.column_names == b.column_names and a.num_rows == b.num_rows and all(a[c].equals(b[c]) for c in a.column_names)


def test_filter_equals_arrow_compute(ctx):
    rng = np.random.default_rng(41)
    n = 300_000
    t = pa.table({"k": nullable(rng.integers(0, 1000, n), 0.1, rng, pa.int64()), "w": pa.array(rng.integers(-9, 9, n).astype(np.int32)),
                  "f": nullable(rng.integers(0, 50, n).astype(np.float64), 0.2, rng, pa.float64()), "j": pa.array(rng.integers(0, 100, n), pa.int64())})
    bs = batches_of(t, [5, 5, 100_003, 250_000])                    # an empty batch, slices that start mid-byte of the bitmaps
    pred = O.BinaryExpr(O.BinaryExpr(O.BinaryExpr(O.Column(0, "k"), ">=", O.Literal("100")), "AND", O.BinaryExpr(O.Column(0, "k"), "<", O.Literal("600"))),
                        "AND", O.BinaryExpr(O.Column(3, "j"), "!=", O.Literal("7")))
    out = pa.Table.from_batches(O.execute(O.GpuFilter(O.Batches(bs), pred), ctx))
    m = pc.and_(pc.and_(pc.greater_equal(t["k"], 100), pc.less(t["k"], 600)), pc.not_equal(t["j"], 7))
    exp = t.filter(m, null_selection_behavior="drop")               # a comparison with NULL is not true
    exp = exp.set_column(1, "w", pc.cast(exp["w"], pa.int64()))      # what has been on the device is Int64 / Float64 (record_batch_convert.rs:19)
    assert same_tables(out, exp)                                     # original row order (executor.rs:479-484)


@pytest.mark.parametrize("jt", ["Inner", "Left", "Right", "Full"])
@pytest.mark.parametrize("nk", [1, 2])
def test_hash_joins_equal_acero(ctx, jt, nk):
    rng = np.random.default_rng(43 + nk)
    nl, nr = 60_000, 90_000
    span = 40_000 if nk == 1 else 200
    lt = pa.table({"id": nullable(rng.integers(0, span, nl), 0.05, rng, pa.int64()), "id2": nullable(rng.integers(0, 150, nl), 0.05, rng, pa.int64()),
                   "lpay": nullable(rng.integers(0, 1000, nl).astype(np.float64), 0.1, rng, pa.float64())})
    rt = pa.table({"fk": nullable(rng.integers(0, span + span // 4, nr), 0.05, rng, pa.int64()), "fk2": nullable(rng.integers(0, 150, nr), 0.05, rng, pa.int64()),
                   "rpay": pa.array(rng.integers(0, 10**9, nr), pa.int64())})
    lk, rk = (["id"], ["fk"]) if nk == 1 else (["id", "id2"], ["fk", "fk2"])
    plan = O.GpuHashJoin(O.Batches(batches_of(lt, [20_001])), O.Batches(batches_of(rt, [1, 50_000])),
                         [O.Column(lt.column_names.index(c), c) for c in lk], [O.Column(rt.column_names.index(c), c) for c in rk], join_type=jt)
    out = pa.Table.from_batches(O.execute(plan, ctx))
    exp = lt.join(rt, keys=lk, right_keys=rk, join_type=ARROW_JOIN[jt], coalesce_keys=False, use_threads=False).select(out.schema.names)
    assert out.num_rows == exp.num_rows
    assert same_tables(sort_all(out), sort_all(exp))               # NULL keys match nothing; the missing side of an outer row is NULL


def test_group_by_equals_acero(ctx):
    rng = np.random.default_rng(47)
    n = 400_000
    for keys in (rng.integers(0, 300, n), rng.integers(0, 300, n) * 1_000_003 - (1 << 41)):      # dense keys; sparse and negative ones (hashed path)
        t = pa.table({"g": nullable(keys, 0.05, rng, pa.int64()), "v": nullable(rng.integers(0, 1024, n).astype(np.float64), 0.1, rng, pa.float64())})
        plan = O.GpuAggregate(O.Batches(batches_of(t, [123_457, 123_460])), [O.Column(0, "g")],
                              [O.Sum(O.Column(1, "v")), O.Count(O.Column(1, "v")), O.Min(O.Column(1, "v")), O.Max(O.Column(1, "v")), O.Avg(O.Column(1, "v"))])
        out = O.execute(plan, ctx)[0]
        exp = t.group_by("g", use_threads=False).aggregate([("v", "sum"), ("v", "count"), ("v", "min"), ("v", "max"), ("v", "mean")]).sort_by("g")   # NULL key last
        assert out.num_rows == exp.num_rows == 301
        assert out.column(0).to_pylist() == exp.column("g").to_pylist()          # the NULL keys form one group, reported last
        for i, name in enumerate(("v_sum", "v_count", "v_min", "v_max", "v_mean")):
            assert np.array_equal(out.column(1 + i).to_numpy(zero_copy_only=False), exp.column(name).to_numpy().astype(np.float64)), name


def test_filter_join_group_by_plan_equals_arrow(ctx):
    """The OP shape end to end: GpuAggregate{ GpuHashJoin{ customers, GpuFilter{ orders } } } against filter -> join -> group_by in Arrow C++."""
    rng = np.random.default_rng(53)
    n_o, n_c = 500_000, 100_000                                       # a build side beyond 65 536 rows: the operator's radix join
    orders = pa.table({"cust": pa.array(rng.integers(0, n_c + n_c // 10, n_o), pa.int64()), "amt": nullable(rng.integers(1, 1000, n_o).astype(np.float64), 0.05, rng, pa.float64()),
                       "flag": pa.array(rng.integers(0, 10, n_o), pa.int64())})
    customers = pa.table({"id": pa.array(rng.permutation(n_c), pa.int64()), "seg": pa.array(rng.integers(0, 64, n_c).astype(np.int32))})
    plan = O.GpuAggregate(O.GpuHashJoin(O.Batches(batches_of(customers, [])), O.GpuFilter(O.Batches(batches_of(orders, [100_000, 300_001])), O.BinaryExpr(O.Column(2, "flag"), "<", O.Literal("5"))),
                                        [O.Column(0, "id")], [O.Column(0, "cust")]),
                          [O.Column(1, "seg")], [O.Sum(O.Column(3, "amt")), O.Count(O.Column(3, "amt")), O.Max(O.Column(3, "amt"))])
    out = O.execute(plan, ctx)[0]
    f = orders.filter(pc.less(orders["flag"], 5))
    j = customers.join(f, keys="id", right_keys="cust", join_type="inner", coalesce_keys=False, use_threads=False)
    exp = j.group_by("seg", use_threads=False).aggregate([("amt", "sum"), ("amt", "count"), ("amt", "max")]).sort_by("seg")
    assert out.column(0).to_pylist() == exp.column("seg").to_pylist()
    for i, name in enumerate(("amt_sum", "amt_count", "amt_max")):
        assert np.array_equal(out.column(1 + i).to_numpy(), exp.column(name).to_numpy().astype(np.float64)), name
