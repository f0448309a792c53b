"""The oracle against INDEPENDENT implementations of the same relational semantics: Apache Arrow C++ (pyarrow.compute and Acero's hash
join / hash aggregate) and pandas.  The reference's CPU path is Arrow-based (arrow-rs; gpu-olap-core/src/executor.rs) but cannot be built
here, and its own join returns no rows (executor.rs:255-264) -- so for join row-id pairs, NULL handling and grouped aggregates the
reference's tests pin nothing.  What CAN be pinned is that the oracle's reading of "equi-join", "comparison with NULL is not true" and
"SUM / COUNT / MIN / MAX / AVG per group" is the one every Arrow implementation shares: these tests compare bit for bit (integer-valued
doubles: sums are exact in any order).  CPU only; the GPU path is compared with the oracle in the -m gpu tests."""
import numpy as np
import pytest

import oracle

pa = pytest.importorskip("pyarrow")
pc = pytest.importorskip("pyarrow.compute")

NULL = 0xFFFFFFFF
ARROW_JOIN = {0: "inner", 1: "left outer", 2: "right outer", 3: "full outer"}   # build = left (executor.rs:238-253); 1 keeps unmatched build rows


def canonical(pairs):
    """(M,2) [build_rid, probe_rid] sorted by (probe, build) -- SURVEY 8(c)'s canonical form, NULL = 0xFFFFFFFF last."""
    p = np.asarray(pairs, dtype=np.uint64).reshape(-1, 2)
    return p[np.lexsort((p[:, 0], p[:, 1]))].astype(np.uint32)


def masked(col, valid):
    return pa.array(col, mask=None if valid is None else ~valid.astype(bool))


def arrow_join(bcols, pcols, jt, bvalid=None, pvalid=None):
    nk = len(bcols)
    lt = pa.table({**{f"b{k}": masked(bcols[k], None if bvalid is None else bvalid[k]) for k in range(nk)}, "b_rid": np.arange(len(bcols[0]), dtype=np.int64)})
    rt = pa.table({**{f"p{k}": masked(pcols[k], None if pvalid is None else pvalid[k]) for k in range(nk)}, "p_rid": np.arange(len(pcols[0]), dtype=np.int64)})
    j = lt.join(rt, keys=[f"b{k}" for k in range(nk)], right_keys=[f"p{k}" for k in range(nk)], join_type=ARROW_JOIN[jt], coalesce_keys=False, use_threads=False)
    b = j.column("b_rid").combine_chunks().fill_null(NULL).to_numpy()
    p = j.column("p_rid").combine_chunks().fill_null(NULL).to_numpy()
    return canonical(np.stack([b, p], axis=1))


@pytest.mark.parametrize("jt", [0, 1, 2, 3])
@pytest.mark.parametrize("seed", [1, 2])
def test_single_key_joins_match_acero_and_pandas(seed, jt):
    pd = pytest.importorskip("pandas")
    rng = np.random.default_rng(100 * seed + jt)
    nb, npr = int(rng.integers(1, 4000)), int(rng.integers(1, 6000))
    span = int(rng.integers(1, 3000))
    b = rng.integers(-span, span, nb).astype(np.int64)            # duplicates on both sides, keys that match nobody
    p = rng.integers(-span, span + span // 3, npr).astype(np.int64)
    got, m = oracle.hash_join_outer(b, p, jt)
    assert m == len(got)
    assert np.array_equal(canonical(got), got)                      # the oracle hands out the canonical order itself
    assert np.array_equal(got, arrow_join([b], [p], jt))
    how = {0: "inner", 1: "left", 2: "right", 3: "outer"}[jt]
    mg = pd.DataFrame({"k": b, "b_rid": np.arange(nb)}).merge(pd.DataFrame({"k": p, "p_rid": np.arange(npr)}), on="k", how=how)
    pairs = np.stack([mg["b_rid"].fillna(NULL).to_numpy(np.int64), mg["p_rid"].fillna(NULL).to_numpy(np.int64)], axis=1)
    assert np.array_equal(got, canonical(pairs))
    if jt == 0:                                                     # the plain inner join entry point says the same
        inner, mi = oracle.hash_join(b, p)
        assert mi == m and np.array_equal(canonical(inner), got)


@pytest.mark.parametrize("jt", [0, 1, 2, 3])
@pytest.mark.parametrize("nk", [1, 2, 4])
def test_composite_and_null_keys_match_acero(nk, jt):
    """SQL semantics of the multi-column NULL-aware join (an extension over the reference, which reads the first key only and has turned
    NULLs into 0 by then): a NULL in any key column matches nothing; outer joins keep such rows as unmatched."""
    rng = np.random.default_rng(7 * nk + jt)
    nb, npr = 3000, 4500
    bcols = [rng.integers(0, 12, nb).astype(np.int64) for _ in range(nk)]
    pcols = [rng.integers(0, 13, npr).astype(np.int64) for _ in range(nk)]
    if nk == 1:
        bcols, pcols = [rng.integers(0, 900, nb).astype(np.int64)], [rng.integers(0, 1000, npr).astype(np.int64)]
    bvalid = [(rng.random(nb) > 0.1).astype(np.uint8) if k % 2 == 0 else None for k in range(nk)]
    pvalid = [(rng.random(npr) > 0.15).astype(np.uint8) for _ in range(nk)]
    got = oracle.hash_join_multi(bcols, pcols, jt, bvalid, pvalid)
    got = got[0] if isinstance(got, tuple) else got
    assert np.array_equal(canonical(got), arrow_join(bcols, pcols, jt, bvalid, pvalid))


@pytest.mark.parametrize("jt", [0, 1, 2, 3])
def test_join_edge_cases_match_acero(jt):
    """empty sides, one row, every key NULL on one side, every row the same key (a full cross product of the two groups)"""
    e = np.empty(0, np.int64)
    one = np.array([7], np.int64)
    same_b, same_p = np.full(37, 5, np.int64), np.full(53, 5, np.int64)
    cases = [([e], [e], None, None), ([e], [one], None, None), ([one], [e], None, None), ([one], [one], None, None), ([one], [one + 1], None, None),
             ([same_b], [same_p], None, None),
             ([same_b], [same_p], [np.zeros(37, np.uint8)], None),                    # every build key NULL: nothing matches
             ([same_b, same_b], [same_p, same_p], [None, (np.arange(37) % 2).astype(np.uint8)], [(np.arange(53) % 3 > 0).astype(np.uint8), None])]
    for bc, pc_, bv, pv in cases:
        got = oracle.hash_join_multi(bc, pc_, jt, bv, pv)
        got = got[0] if isinstance(got, tuple) else got
        assert np.array_equal(canonical(got), arrow_join(bc, pc_, jt, bv, pv)), (jt, [len(c) for c in bc], [len(c) for c in pc_])


@pytest.mark.parametrize("op,fn", [("=", "equal"), ("!=", "not_equal"), ("<", "less"), ("<=", "less_equal"), (">", "greater"), (">=", "greater_equal")])
def test_filter_masks_match_arrow_compute(op, fn):
    """eval_filter_mask (executor.rs:443-476) against pyarrow.compute; with a validity array: Arrow's Kleene-free comparison yields NULL for a
    NULL operand and `filter` drops it -- "a comparison with NULL is not true"."""
    rng = np.random.default_rng(5)
    col = rng.integers(-50, 50, 10_000).astype(np.int64)
    valid = (rng.random(col.size) > 0.2).astype(np.uint8)
    lit = 7
    m = oracle.eval_filter_mask(col, op, lit).astype(bool)
    assert np.array_equal(m, getattr(pc, fn)(pa.array(col), lit).to_numpy(zero_copy_only=False))
    mn = oracle.eval_filter_mask_and([col], [op], [lit], [valid])
    rows = pa.array(np.arange(col.size))
    kept = pc.filter(rows, getattr(pc, fn)(masked(col, valid), lit), null_selection_behavior="drop").to_numpy()
    assert np.array_equal(oracle.mask_to_indices(mn.astype(np.uint8)), kept.astype(np.uint32))
    # order-preserving gather (filter_record_batch / take_int64, executor.rs:478-507) = arrow's take
    payload = rng.random(col.size)
    assert np.array_equal(oracle.take_8byte(payload, kept.astype(np.uint32)), pc.take(pa.array(payload), pa.array(kept)).to_numpy())


def test_ungrouped_aggregates_match_arrow_compute():
    """execute_aggregate (executor.rs:297-355): SUM / COUNT / MIN / MAX / AVG in f64, Int64 inputs cast `as f64`."""
    rng = np.random.default_rng(9)
    for vals in (rng.integers(-1000, 1000, 50_000).astype(np.float64), rng.integers(-10**6, 10**6, 30_000).astype(np.int64)):
        a = oracle.execute_aggregate(vals)
        arr = pa.array(vals)
        mm = pc.min_max(arr)
        assert a["sum"] == float(pc.sum(arr).as_py()) and a["count"] == float(pc.count(arr).as_py())
        assert a["min"] == float(mm["min"].as_py()) and a["max"] == float(mm["max"].as_py())
        assert a["avg"] == a["sum"] / a["count"] == pc.mean(arr).as_py()
    e = oracle.execute_aggregate(np.empty(0, np.float64))
    assert e["count"] == 0.0 and e["sum"] == 0.0 and e["avg"] == 0.0        # AVG of nothing is 0.0 in the reference (:349), NULL in Arrow / SQL


@pytest.mark.parametrize("n,g,sparse", [(50_000, 17, False), (80_000, 5000, False), (20_000, 3000, True)])
def test_grouped_aggregates_match_acero(n, g, sparse):
    """group_aggregate (the GROUP BY the reference's executor skips, executor.rs:307; SUM per key as hash_aggregate_kernel computes it)
    against Acero's hash aggregate: the same groups, and per group the same sum / count / min / max / mean."""
    rng = np.random.default_rng(n + g)
    keys = rng.integers(0, g, n).astype(np.int64)
    if sparse:
        keys = keys * 1_000_003 - 2**40                              # negative and far apart
    vals = rng.integers(0, 1024, n).astype(np.float64)
    ok, st = oracle.group_aggregate(keys, vals)
    t = pa.table({"k": keys, "v": vals}).group_by("k", use_threads=False).aggregate([("v", "sum"), ("v", "count"), ("v", "min"), ("v", "max"), ("v", "mean")]).sort_by("k")
    assert np.array_equal(ok, t.column("k").to_numpy())
    for j, name in enumerate(("v_sum", "v_count", "v_min", "v_max", "v_mean")):
        assert np.array_equal(st[:, j], t.column(name).to_numpy().astype(np.float64)), name
