"""BASELINE.json's configurations at their FULL sizes, checked on the device (the oracle finishes small cases in
seconds; at full size the results are checked through properties that do not depend on the size, computed with plain torch ops
that share no code with the library).

join (C3 100M x 100M, C5 1B x 1B): build keys are a permutation of 0..N-1 (GEN_BUILD_KEYS multiplies by a constant coprime to N),
so the exact answer is known in closed form: probe row j pairs with build row inv[probe_key[j]] and with nothing else.  The pair
list is scattered by probe row id and compared with that -- every pair, bit-exact -- plus the match count.

GROUP BY (C4 1B rows, int32 keys, G = 1024 and 1,048,576): integer-valued doubles, whose sums are exact in any order (< 2^53),
against torch.bincount with weights; counts against torch.bincount.
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _check_join(ctx, n, wide_offset=0):
    from gpu_olap_engine_amd import _native as N
    b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n)
    p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
    inv = torch.empty(n, dtype=torch.int32, device="cuda")
    inv[b] = torch.arange(n, dtype=torch.int32, device="cuda")             # build row of every key
    want = inv[p]                                                          # the one build row each probe row must meet
    del inv
    if wide_offset:
        b += wide_offset
        p += wide_offset
    out = torch.empty((n, 2), dtype=torch.int32, device="cuda")
    pairs, cnt = ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH)
    torch.cuda.synchronize()
    assert int(cnt.item()) == n
    got = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    got[pairs[:, 1].to(torch.int64) & 0xFFFFFFFF] = pairs[:, 0]            # probe ids are unique iff every probe row appears once
    assert bool((got == want).all()), "a probe row is missing, duplicated or paired with the wrong build row"
    plan = ctx.join_plan(n)
    return plan


def test_c3_100m_join_every_pair():
    """BASELINE config 3: 100M x 100M int64, [8,7] radix plan, no-histogram narrow pipeline."""
    from gpu_olap_engine_amd.engine import Context
    c = Context(max_gpu_memory=0)
    try:
        plan = _check_join(c, 100_000_000)
        assert plan["bits1"] > 0 and plan["bits2"] > 0 and plan["no_histogram_first_pass"] and plan["optimistic_narrow_keys"]
        _check_join(c, 100_000_000, wide_offset=(1 << 40) + 5)              # the same join on keys outside [0, 2^32)
    finally:
        c.close()
        torch.cuda.empty_cache()


def test_c5_1b_join_every_pair():
    """BASELINE config 5 (the bench workload): 1B x 1B int64."""
    from gpu_olap_engine_amd.engine import Context
    c = Context(max_gpu_memory=0)
    try:
        plan = _check_join(c, 1_000_000_000)
        assert plan["bits1"] > 0 and plan["bits2"] > 0 and plan["no_histogram_first_pass"] and plan["optimistic_narrow_keys"]
    finally:
        c.close()
        torch.cuda.empty_cache()


def test_join_above_one_lds_table_per_subpartition():
    """1.1B x 1.1B: 18 hash bits no longer fit one 4096-row LDS table per sub-partition (mean 4196 rows).  The plan stays at two
    passes -- a third one would more than double the time -- and the join loops over two table chunks where needed; every pair
    is checked like in the 1B case."""
    from gpu_olap_engine_amd.engine import Context
    c = Context(max_gpu_memory=0)
    try:
        plan = _check_join(c, 1_100_000_000)
        assert plan["bits1"] == 9 and plan["bits2"] == 9
    finally:
        c.close()
        torch.cuda.empty_cache()


@pytest.mark.parametrize("n_groups", [1024, 1_048_576])
def test_c4_1b_group_by_sum_and_count(n_groups):
    """BASELINE config 4: 1B rows, int32 group key, SUM(f64) (+ COUNT): LDS pre-aggregation at G = 1024, partition-then-aggregate
    at G = 2^20."""
    from gpu_olap_engine_amd import _native as N
    from gpu_olap_engine_amd.engine import Context
    n = 1_000_000_000
    c = Context(max_gpu_memory=0, profiling=True)
    try:
        keys = c.generate(N.GEN_SM64_MOD_I32, n, modulus=n_groups, seed=7)
        vals = c.generate(N.GEN_SM64_MOD_F64, n, modulus=1000, seed=8)
        out, counts = c.aggregate([(N.AGG_SUM, 0), (N.AGG_COUNT, 0), (N.AGG_MAX, 0)], [vals], group_keys=keys, n_groups=n_groups, want_counts=True)
        torch.cuda.synchronize()
        k64 = keys.to(torch.int64)
        del keys
        want_cnt = torch.bincount(k64, minlength=n_groups)
        want_sum = torch.bincount(k64, weights=vals, minlength=n_groups)
        assert int(want_cnt.sum().item()) == n
        assert torch.equal(counts, want_cnt)
        assert torch.equal(out[1], want_cnt.to(torch.float64))             # COUNT as f64 (executor.rs:346)
        assert torch.equal(out[0], want_sum)                                # exact: integer-valued doubles
        want_max = torch.full((n_groups,), -1.0, dtype=torch.float64, device="cuda").scatter_reduce_(0, k64, vals, "amax")
        assert torch.equal(out[2], want_max)
        del want_max, k64
        # the bench's own call (SUM only: at G = 2^20 the partition scatter without a histogram pass, k_agg_scatter_rsv)
        keys = c.generate(N.GEN_SM64_MOD_I32, n, modulus=n_groups, seed=7)
        c.stats_reset()
        out2 = c.aggregate([(N.AGG_SUM, 0)], [vals], group_keys=keys, n_groups=n_groups)
        torch.cuda.synchronize()
        assert torch.equal(out2[0], want_sum)
        if n_groups > 4096:
            st = c.stats()
            assert st["agg_partition_scatter"]["bytes"] > 0 and st["agg_partition_scatter_standby"]["bytes"] == 0
    finally:
        c.close()
        torch.cuda.empty_cache()


def test_c1_shape_filter_sum_1b():
    """BASELINE config 1's shape (filter feeding an ungrouped SUM) at 1B rows, in one pass: against torch's masked sum."""
    from gpu_olap_engine_amd import _native as N
    from gpu_olap_engine_amd.engine import Context
    n = 1_000_000_000
    c = Context(max_gpu_memory=0)
    try:
        col = c.generate(N.GEN_SM64_MOD_I64, n, modulus=1000, seed=3)
        val = c.generate(N.GEN_SM64_MOD_F64, n, modulus=1000, seed=4)
        r = c.filter_sum(col, val, ">", 500)
        torch.cuda.synchronize()
        m = col > 500
        assert float(r[1].item()) == float(m.sum().item())
        assert float(r[0].item()) == float(val[m].sum().item())
    finally:
        c.close()
        torch.cuda.empty_cache()
