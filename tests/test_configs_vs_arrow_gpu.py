"""BASELINE.json's configurations through the C ABI on the GPU against Apache Arrow C++ on the CPU, no oracle in between: the bench's own
counter-based generators, Arrow's filter / hash join / hash aggregate as the CPU side.  north_star: "join output (row-id
pairs) and group-by results must be bit-exact against the reference's CPU Arrow path on the same synthetic tables" -- that path (arrow-rs)
cannot be built here and returns no join rows (executor.rs:255-264); Arrow C++ is the runnable Arrow implementation of the same
operators.  Sizes are what Acero finishes in seconds on one thread; the full-size runs are checked in closed form (test_fullsize_gpu.py)."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from gpu_olap_engine_amd import _native as N  # noqa: E402


def host(t):
    return t.cpu().numpy()


def canonical(pairs):
    """[build_rid, probe_rid] pairs as sorted 64-bit words probe << 32 | build: SURVEY 8(c)'s canonical (probe, build) order"""
    p = np.asarray(pairs).reshape(-1, 2).astype(np.uint64)
    w = (p[:, 1] << np.uint64(32)) | p[:, 0]
    w.sort()
    return w


def acero_pairs(b, p):
    lt = pa.table({"bk": b, "b_rid": np.arange(len(b), dtype=np.int64)})
    rt = pa.table({"pk": p, "p_rid": np.arange(len(p), dtype=np.int64)})
    j = lt.join(rt, keys="bk", right_keys="pk", join_type="inner", coalesce_keys=False, use_threads=False)
    return canonical(np.stack([j["b_rid"].to_numpy(), j["p_rid"].to_numpy()], axis=1))


def test_config1_filter_sum_equals_arrow(ctx):
    """config 1: SELECT sum(v) FROM t WHERE k < 500, one 1M-row batch."""
    n = 1_000_000
    k = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=1000, seed=1)
    v = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=1024, seed=2)
    got = host(ctx.filter_sum(k, v, "<", 500))
    ka, va = pa.array(host(k)), pa.array(host(v))
    kept = pc.filter(va, pc.less(ka, 500))
    assert got[0] == float(pc.sum(kept).as_py()) and got[1] == float(len(kept))
    idx, cnt = ctx.filter_compact(k, "<", 500)                       # the unfused operator: row ids in the original order (executor.rs:479-484)
    c = int(cnt.item())
    assert c == len(kept) and np.array_equal(host(idx[:c]).astype(np.int64), pc.indices_nonzero(pc.less(ka, 500)).to_numpy().astype(np.int64))
    assert np.array_equal(host(ctx.take(v, idx, cnt, n_max=n)[:c]), kept.to_numpy())


@pytest.mark.parametrize("strategy", [N.JOIN_RADIX_HASH, N.JOIN_BROADCAST_HASH])
def test_config2_1m_join_pairs_equal_acero(ctx, strategy):
    """config 2: 1M x 1M, unique build keys, one match per probe row -- every row-id pair, both strategies."""
    n = 1_000_000
    b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n)
    p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
    pairs, cnt = ctx.join(b, p, max_pairs=n, strategy=strategy)
    m = int(cnt.item())
    assert m == n
    assert np.array_equal(canonical(host(pairs[:m]).view(np.uint32)), acero_pairs(host(b), host(p)))


@pytest.mark.parametrize("kind", ["unique", "duplicates", "wide"])
def test_config3_shape_join_pairs_equal_acero(ctx, kind):
    """config 3's pipeline (256 radix partitions, two passes, LDS join) at 6M x 6M rows: unique keys, the duplicate variant (both sides
    sm64 mod N/2, E[M] ~ 2N: examples/python_usage.py:301-309) and keys with scattered upper halves (the 64-bit-row pipeline)."""
    n = 6_000_000
    if kind == "duplicates":
        b = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n // 2, seed=11)
        p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n // 2, seed=12)
    else:
        b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n)
        p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
        if kind == "wide":
            for t in (b, p):
                t += ((t * 0x9E3779B1) & 0xFFFF) << 44
    cap = 3 * n
    pairs, cnt = ctx.join(b, p, max_pairs=cap, strategy=N.JOIN_RADIX_HASH)
    m = int(cnt.item())
    exp = acero_pairs(host(b), host(p))
    assert m == len(exp)
    assert np.array_equal(canonical(host(pairs[:m]).view(np.uint32)), exp)


@pytest.mark.parametrize("groups", [1024, 1 << 20])
def test_config4_shape_group_by_equals_acero(ctx, groups):
    """config 4: GROUP BY int32 key SUM(float64) -- the LDS path (G = 1024) and the partition-then-aggregate path (G = 2^20) at 30M rows,
    plus COUNT / MIN / MAX / AVG per group."""
    n = 30_000_000
    keys = ctx.generate(N.GEN_SM64_MOD_I32, n, modulus=groups, seed=7)
    vals = ctx.generate(N.GEN_SM64_MOD_F64, n, modulus=1024, seed=8)
    out = host(ctx.aggregate([(N.AGG_SUM, 0), (N.AGG_COUNT, 0), (N.AGG_MIN, 0), (N.AGG_MAX, 0), (N.AGG_AVG, 0)], [vals], group_keys=keys, n_groups=groups))
    t = pa.table({"k": host(keys), "v": host(vals)}).group_by("k", use_threads=False).aggregate([("v", "sum"), ("v", "count"), ("v", "min"), ("v", "max"), ("v", "mean")]).sort_by("k")
    present = t["k"].to_numpy().astype(np.int64)
    assert present.size == groups or groups > 1024                  # (30M rows over 2^20 groups: every group is hit with overwhelming probability, not by construction)
    for i, name in enumerate(("v_sum", "v_count", "v_min", "v_max", "v_mean")):
        assert np.array_equal(out[i][present], t[name].to_numpy().astype(np.float64)), name
    absent = np.setdiff1d(np.arange(groups), present)
    assert np.all(out[0][absent] == 0) and np.all(out[1][absent] == 0)
