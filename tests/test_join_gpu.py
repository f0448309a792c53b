"""GPU parity tests for GpuHashJoin: the HIP path (through the C ABI) vs the CPU oracle,
bit-exact on canonicalised row-id pairs (sorted by (probe_rid, build_rid))."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import oracle  # noqa: E402
from gpu_olap_engine_amd import _native as N  # noqa: E402
from np_twin import canonical  # noqa: E402

STRATS = [N.JOIN_RADIX_HASH, N.JOIN_BROADCAST_HASH]


def run_join(ctx, b, p, strategy, max_pairs=None, **kw):
    bt = torch.from_numpy(np.ascontiguousarray(b, dtype=np.int64)).cuda()
    pt = torch.from_numpy(np.ascontiguousarray(p, dtype=np.int64)).cuda()
    cap = max(1, (max_pairs if max_pairs is not None else 4 * max(len(b), len(p)) + 16))
    pairs, cnt = ctx.join(bt, pt, max_pairs=cap, strategy=strategy, **kw)
    torch.cuda.synchronize()
    m = int(cnt.item())
    got = pairs[:min(m, cap)].cpu().numpy().view(np.uint32)
    return canonical(got), m


def check(ctx, b, p, strategy, max_pairs=None, **kw):
    got, m = run_join(ctx, b, p, strategy, max_pairs=max_pairs, **kw)
    exp, me = oracle.hash_join(b, p, kw.get("build_rid_base", 0), kw.get("probe_rid_base", 0))
    assert m == me, f"match count {m} != oracle {me}"
    assert np.array_equal(got, exp)
    return m


@pytest.mark.parametrize("strategy", STRATS)
def test_tiny_known_answer(ctx, strategy):
    # hand-checkable: build [10,20,30,20], probe [20,5,30,20] -> (1,0),(3,0),(2,2),(1,3),(3,3)
    got, m = run_join(ctx, [10, 20, 30, 20], [20, 5, 30, 20], strategy)
    assert m == 5
    assert got.tolist() == [[1, 0], [3, 0], [2, 2], [1, 3], [3, 3]]


@pytest.mark.parametrize("strategy", STRATS)
@pytest.mark.parametrize("nb,npr", [(0, 0), (0, 5), (5, 0), (1, 1), (63, 65), (4097, 4095), (1000, 100000), (100000, 1000)])
def test_empty_and_ragged(ctx, strategy, nb, npr):
    rng = np.random.default_rng(nb * 7 + npr)
    b = rng.integers(0, max(1, nb // 2 + 1), nb)
    p = rng.integers(0, max(1, nb // 2 + 2), npr)
    check(ctx, b, p, strategy)


@pytest.mark.parametrize("strategy", STRATS)
def test_config2_1m_unique(ctx, strategy):
    """BASELINE config 2: 1M join 1M, unique build keys, every probe row matches exactly once."""
    n = 1_000_000
    b = oracle.gen_build_keys(0, n, n)
    p = oracle.gen_probe_keys(0, n, n)
    assert check(ctx, b, p, strategy) == n


@pytest.mark.parametrize("strategy", STRATS)
def test_duplicates_both_sides(ctx, strategy):
    """dup variant (examples/python_usage.py:301-309): both sides sm64 mod N/2 -> E[M] ~ 2N."""
    n = 300_000
    b = oracle.gen_sm64_mod(11, 0, n, n // 2)
    p = oracle.gen_sm64_mod(12, 0, n, n // 2)
    m = check(ctx, b, p, strategy)
    assert 1.8 * n < m < 2.2 * n


@pytest.mark.parametrize("strategy", STRATS)
def test_negative_and_extreme_keys(ctx, strategy):
    rng = np.random.default_rng(5)
    base = np.array([0, -1, 1, -2, 2**32, 2**32 + 1, -(2**63), 2**63 - 1, 42, -42], dtype=np.int64)
    b = np.concatenate([base, rng.integers(-2**62, 2**62, 5000)])
    p = np.concatenate([base[::-1], b[rng.integers(0, b.size, 5000)], rng.integers(-2**62, 2**62, 100)])
    check(ctx, b, p, strategy)


@pytest.mark.parametrize("strategy", STRATS)
def test_fold_collision_keys(ctx, strategy):
    """hash_int64 folds hi^lo (join_kernels.cuh:39): key i<<32 hashes like key i.  Every probe key
    i meets a build key i<<32 with the SAME hash but a different value: only the full-key compare
    (join_kernels.cuh:152-153) keeps them apart."""
    n = 20_000
    b = np.concatenate([np.arange(n, dtype=np.int64) << 32, np.arange(0, n, 2, dtype=np.int64)])
    p = np.arange(n, dtype=np.int64)
    assert np.array_equal(oracle.hash_int64_array(b[:n]), oracle.hash_int64_array(p))
    m = check(ctx, b, p, strategy)
    assert m == n // 2 + 1   # key 0 is both 0<<32 and 0


@pytest.mark.parametrize("strategy", STRATS)
def test_same_hash_distinct_keys(ctx, strategy):
    """keys (i<<32)|(i^c) all fold to c: ONE hash value, so one sub-partition holds the whole build side
    (20000 distinct keys > the 4096-row LDS table -> chunked build) and chains are told apart by key only."""
    n = 20_000
    i = np.arange(n, dtype=np.int64)
    b = (i << 32) | (i ^ 0x5A5A)
    assert len(set(oracle.hash_int64_array(b).tolist())) == 1
    p = np.concatenate([b[::3], (i[:500] << 32) | (i[:500] ^ 0x5A5B)])
    m = check(ctx, b, p, strategy)
    assert m == len(b[::3])


@pytest.mark.parametrize("strategy", STRATS)
def test_heavy_skew_single_key(ctx, strategy):
    """one hot key on both sides: M = nb_hot * np_hot, overflows the LDS table chunk and the probe batch."""
    b = np.concatenate([np.full(5000, 7, np.int64), np.arange(100, 3100, dtype=np.int64)])
    p = np.concatenate([np.full(300, 7, np.int64), np.arange(0, 2000, dtype=np.int64)])
    m = check(ctx, b, p, strategy, max_pairs=2_000_000)
    assert m >= 5000 * 300


@pytest.mark.parametrize("strategy", STRATS)
def test_rid_bases(ctx, strategy):
    rng = np.random.default_rng(9)
    b = rng.integers(0, 5000, 10000)
    p = rng.integers(0, 5000, 12000)
    check(ctx, b, p, strategy, build_rid_base=1_000_000, probe_rid_base=4_000_000_000)
    if strategy == N.JOIN_RADIX_HASH:
        from gpu_olap_engine_amd.engine import GolapError
        with pytest.raises(GolapError):      # row ids are u32 and 0xFFFFFFFF is the NULL id: base + rows must stay below it
            run_join(ctx, b, p, strategy, probe_rid_base=0xFFFFFFFF - 5000)


@pytest.mark.parametrize("strategy", STRATS)
def test_capacity_overflow_keeps_counting(ctx, strategy):
    """join_kernels.cuh:155-160: count is the TRUE total; nothing is written past max_pairs."""
    rng = np.random.default_rng(3)
    b = rng.integers(0, 1000, 20000)
    p = rng.integers(0, 1000, 20000)
    exp, me = oracle.hash_join(b, p)
    bt, pt = torch.from_numpy(b).cuda(), torch.from_numpy(p).cuda()
    cap = 1000
    buf = torch.full((cap + 64, 2), -1, dtype=torch.int32, device="cuda")
    pairs, cnt = ctx.join(bt, pt, out_pairs=buf, max_pairs=cap, strategy=strategy)
    torch.cuda.synchronize()
    assert int(cnt.item()) == me
    host = buf.cpu().numpy()
    assert (host[cap:] == -1).all(), "wrote past max_pairs"
    got = host[:cap].view(np.uint32)
    expset = set(map(tuple, exp.tolist()))
    assert all(tuple(r) in expset for r in got.tolist())
    assert len(set(map(tuple, got.tolist()))) == cap


def outer_check(ctx, b, p, strategy, jt):
    bt = torch.from_numpy(np.ascontiguousarray(b, dtype=np.int64)).cuda() if len(b) else torch.empty(0, dtype=torch.int64, device="cuda")
    pt = torch.from_numpy(np.ascontiguousarray(p, dtype=np.int64)).cuda() if len(p) else torch.empty(0, dtype=torch.int64, device="cuda")
    exp, me = oracle.hash_join_outer(b, p, jt)
    cap = max(1, me + 64)
    pairs, cnt = ctx.join(bt, pt, max_pairs=cap, strategy=strategy, join_type=jt)
    torch.cuda.synchronize()
    m = int(cnt.item())
    assert m == me, f"jt={jt}: count {m} != oracle {me}"
    got = canonical(pairs[:m].cpu().numpy().view(np.uint32))
    assert np.array_equal(got, exp), f"jt={jt}"


@pytest.mark.parametrize("strategy", STRATS)
@pytest.mark.parametrize("jt", [N.JT_INNER, N.JT_LEFT, N.JT_RIGHT, N.JT_FULL])
def test_outer_joins(ctx, strategy, jt):
    """JoinType (logical_plan.rs:53-58): Left keeps every build row, Right every probe row, Full both; NULL side = 0xFFFFFFFF."""
    rng = np.random.default_rng(jt)
    # hand-checkable
    outer_check(ctx, [1, 2, 2, 3, 9], [2, 5, 3, 2], strategy, jt)
    # random with misses and duplicates on both sides
    outer_check(ctx, rng.integers(0, 3000, 20_000), rng.integers(1500, 4500, 30_000), strategy, jt)
    # empty sides
    outer_check(ctx, [], [4, 5, 6], strategy, jt)
    outer_check(ctx, [4, 5, 6], [], strategy, jt)
    # disjoint: nothing matches
    outer_check(ctx, np.arange(0, 5000), np.arange(10_000, 17_000), strategy, jt)
    # skew: one sub-partition needs several LDS table chunks (probe "matched" state must survive across chunks)
    i = np.arange(12_000, dtype=np.int64)
    b = (i << 32) | (i ^ 0x5A5A)                      # one hash value, 12000 distinct keys -> 3 table chunks
    p = np.concatenate([b[::2], (i[:3000] << 32) | (i[:3000] ^ 0x1111), b[:10]])
    outer_check(ctx, b, p, strategy, jt)


def test_outer_join_large_radix(ctx):
    n = 3_000_000
    b = oracle.gen_sm64_mod(51, 0, n, 2 * n)
    p = oracle.gen_sm64_mod(52, 0, n, 2 * n)
    outer_check(ctx, b, p, N.JOIN_RADIX_HASH, N.JT_FULL)


def test_radix_multi_pass_16m(ctx):
    """16M join 16M forces two partition passes (sub-partitions must fit the LDS table)."""
    n = 16_000_000
    plan = ctx.join_plan(n)
    assert plan["bits2"] > 0
    bt = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n)
    pt = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
    pairs, cnt = ctx.join(bt, pt, max_pairs=n, strategy=N.JOIN_RADIX_HASH)
    torch.cuda.synchronize()
    m = int(cnt.item())
    assert m == n
    exp, me = oracle.hash_join(bt.cpu().numpy(), pt.cpu().numpy())
    got = canonical(pairs[:m].cpu().numpy().view(np.uint32))
    assert me == n and np.array_equal(got, exp)


@pytest.mark.parametrize("env", [{"GOLAP_NOHIST": "0"}, {"GOLAP_SLOT_SIGMAS": "0"}, {"GOLAP_SLOT_SIGMAS": "1"}, {}])
@pytest.mark.parametrize("jt", [0, 3])
def test_no_histogram_last_pass_and_its_overflow_fallback(ctx, env, jt):
    """Two-pass joins skip the second histogram (slots of avg + 6 sigma); GOLAP_SLOT_SIGMAS=0/1 force slot overflows,
    so the device-side flag must hand over to the stand-by exact pass; GOLAP_NOHIST=0 is the plain exact path."""
    n = 6_000_000
    b = oracle.gen_sm64_mod(61, 0, n, n)              # duplicates + misses on both sides
    p = oracle.gen_sm64_mod(62, 0, n, n)
    env = dict(env, GOLAP_PASS_BITS="8,6")            # second pass with fan-out 64: the no-histogram form applies (it needs F >= 32)
    with ctx.knobs(**env):
        outer_check(ctx, b, p, N.JOIN_RADIX_HASH, jt)


@pytest.mark.parametrize("env", [{}, {"GOLAP_RESERVE_SIGMAS": "0"}, {"GOLAP_RESERVE_SIGMAS": "1"}, {"GOLAP_SLOT_SIGMAS": "0"},
                                 {"GOLAP_RESERVE_SIGMAS": "0", "GOLAP_SLOT_SIGMAS": "0"}])
@pytest.mark.parametrize("jt", [0, 3])
def test_no_histogram_first_pass_and_its_overflow_fallback(ctx, env, jt):
    """Large two-pass joins also skip the FIRST histogram: tiles reserve their bucket runs with atomics inside ranges of
    n/F + 6 sigma rows.  GOLAP_RESERVE=1 applies that form at a size the oracle checks in seconds; GOLAP_RESERVE_SIGMAS=0/1
    force range overflows (hand-over to the stand-by histogram pass), alone and together with second-pass slot overflows."""
    n = 6_000_000
    b = oracle.gen_sm64_mod(71, 0, n, n)
    p = oracle.gen_sm64_mod(72, 0, n, n)
    env = dict(env, GOLAP_PASS_BITS="8,6", GOLAP_RESERVE="1")
    with ctx.knobs(**env):
        ctx.stats_reset()
        outer_check(ctx, b, p, N.JOIN_RADIX_HASH, jt)
        st = ctx.stats()
        assert "radix_hist_p1" not in st and "radix_hist_p1_standby" in st     # the reserving form ran, the histogram stood by


def test_no_histogram_first_pass_with_one_hot_key(ctx):
    """a bucket that outgrows its reserved range for real (one key holds 10 % of the build side)."""
    n = 4_000_000
    rng = np.random.default_rng(9)
    b = np.concatenate([oracle.gen_build_keys(0, n, n), np.full(400_000, 777, np.int64)])
    p = np.concatenate([oracle.gen_probe_keys(0, n, n), np.full(3, 777, np.int64), rng.integers(0, n, 1000)])
    env = dict(GOLAP_PASS_BITS="8,6", GOLAP_RESERVE="1")
    with ctx.knobs(**env):
        check(ctx, b, p, N.JOIN_RADIX_HASH, max_pairs=8_000_000)


def test_no_histogram_pass_with_skewed_groups(ctx):
    """a few heavy keys overflow their slots for real (no test hook): the fallback must kick in on its own."""
    n = 5_000_000
    rng = np.random.default_rng(8)
    b = np.concatenate([oracle.gen_build_keys(0, n, n), np.full(30_000, 123456, np.int64)])
    p = np.concatenate([oracle.gen_probe_keys(0, n, n), np.full(40, 123456, np.int64), rng.integers(0, n, 1000)])
    with ctx.knobs(GOLAP_PASS_BITS="8,6"):
        check(ctx, b, p, N.JOIN_RADIX_HASH, max_pairs=8_000_000)


def test_no_histogram_pass_is_taken_and_flag_stays_clear(ctx):
    """stats show the no-histogram form ran (stand-by kernels present, no radix_hist_p2) and verify_pairs agrees at 60M rows."""
    n = 60_000_000
    assert ctx.join_plan(n)["bits2"] >= 5
    bt = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n)
    pt = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
    out = torch.empty((n, 2), dtype=torch.int32, device="cuda")
    ctx.stats_reset()
    pairs, cnt = ctx.join(bt, pt, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH)
    v = ctx.verify_pairs(pairs, n, bt, pt).cpu().numpy().view(np.uint64)
    torch.cuda.synchronize()
    st = ctx.stats()
    assert "radix_scatter_p2_standby" in st and "radix_hist_p2" not in st
    assert "radix_scatter_p1_standby" in st and "radix_hist_p1" not in st      # first pass reserved its output, no histogram either
    assert st["radix_scatter_p2_standby"]["total_ms"] < 0.5          # stand-by launches return at once: the flags stayed clear
    assert st["radix_scatter_p1_standby"]["total_ms"] < 0.5
    assert int(cnt.item()) == n and int(v[0]) == n and int(v[2]) == n * (n - 1) // 2


@pytest.mark.parametrize("nb,npr", [(60_000_000, 1_000_000), (52_000_000, 70_000_000), (60_000_000, 0), (0, 60_000_000), (60_000_000, 1),
                                    (30_000_000, 27_000_000), (26_000_000, 24_000_000)])
def test_asymmetric_sides_mix_first_pass_forms(ctx, nb, npr):
    """each side picks its first-pass form by its own size: a 60M-row build side reserves (no histogram) while a 1M-row probe
    side takes the histogram path, and vice versa; checked on the device (every pair key-equal, probe ids a permutation)."""
    bt = ctx.generate(N.GEN_BUILD_KEYS, nb, modulus=max(nb, 1)) if nb else torch.empty(0, dtype=torch.int64, device="cuda")
    pt = ctx.generate(N.GEN_SM64_MOD_I64, npr, modulus=max(nb, 1), seed=47) if npr else torch.empty(0, dtype=torch.int64, device="cuda")
    m_exp = npr if nb else 0                                           # an empty side joins to nothing
    out = torch.empty((max(npr, 1), 2), dtype=torch.int32, device="cuda")
    pairs, cnt = ctx.join(bt, pt, out_pairs=out, max_pairs=max(npr, 1), strategy=N.JOIN_RADIX_HASH)
    assert int(cnt.item()) == m_exp
    if m_exp:
        v = ctx.verify_pairs(pairs, npr, bt, pt).cpu().numpy().view(np.uint64)
        assert int(v[0]) == npr and int(v[2]) == npr * (npr - 1) // 2


@pytest.mark.parametrize("shape", ["narrow", "common_upper_half", "negative_small", "two_upper_halves", "negative_and_positive", "four_upper_halves",
                                   "five_upper_halves", "many_upper_halves", "one_wide_key_last", "negative_key_first", "two_halves_and_one_odd_key_last",
                                   "hundred_consecutive_halves", "halves_minus_three_to_forty", "hundred_consecutive_halves_and_one_far_key_last"])
def test_narrow_key_pipeline_and_its_64bit_fallback(ctx, shape):
    """Large joins first try 8-byte (key32, row id) tuples, which stand while the keys of both sides have at most four different
    upper halves (one: 0 for small ids, but also keys offset by 2^40, or small negative keys; two to four: a dictionary index rides
    in the tuple, e.g. small negative NEXT TO small positive keys; any number inside a window of consecutive values as wide as the
    first pass's fan-out -- ids with a tenant / shard prefix: the offset (upper half - window start) rides instead); a fifth upper half
    far from the others -- in the sample, in the first tile or in the very last row -- must hand the whole join over to the 64-bit
    pipeline on the device, with identical results (checked on the device: every pair key-equal, probe ids a permutation)."""
    n = 60_000_000
    bt = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n)
    pt = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
    orphans = 0
    if shape == "common_upper_half":
        bt += 1 << 40
        pt += 1 << 40
    elif shape == "negative_small":
        bt -= n + 5
        pt -= n + 5
    elif shape == "two_upper_halves":
        bt += (bt & 1) << 40
        pt += (pt & 1) << 40
    elif shape == "negative_and_positive":
        bt -= n // 2
        pt -= n // 2
    elif shape in ("four_upper_halves", "five_upper_halves"):
        k = 4 if shape.startswith("four") else 5
        bt += (bt % k) << 40
        pt += (pt % k) << 40
    elif shape == "many_upper_halves":
        bt += ((bt * 0x9E3779B1) & 0xFFFF) << 32
        pt += ((pt * 0x9E3779B1) & 0xFFFF) << 32
    elif shape.startswith("hundred_consecutive_halves"):
        bt += (bt % 100) << 32
        pt += (pt % 100) << 32
        if shape.endswith("far_key_last"):
            old = int(bt[n - 1].item())
            bt[n - 1] = (70_000 << 32) + 11
            pt[n - 1] = (70_000 << 32) + 11                             # far outside the window of upper halves
            orphans = int((pt == old).sum().item())
    elif shape == "halves_minus_three_to_forty":
        bt += ((bt % 44) - 3) << 32
        pt += ((pt % 44) - 3) << 32
    elif shape == "two_halves_and_one_odd_key_last":
        bt += (bt & 1) << 40
        pt += (pt & 1) << 40
        old = int(bt[n - 1].item())
        bt[n - 1] = (5 << 40) + 11
        pt[n - 1] = (5 << 40) + 11                                     # a third upper half the sample cannot have seen
        orphans = int((pt == old).sum().item())
    elif shape != "narrow":
        idx, key = (n - 1, (1 << 35) + 7) if shape == "one_wide_key_last" else (0, -5)
        old = int(bt[idx].item())
        bt[idx] = key
        pt[idx] = key                                                  # the planted pair
        orphans = int((pt == old).sum().item())                        # probe rows whose build key is gone now
    out = torch.empty((n, 2), dtype=torch.int32, device="cuda")
    ctx.stats_reset()
    pairs, cnt = ctx.join(bt, pt, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH)
    m = int(cnt.item())
    v = ctx.verify_pairs(pairs, m, bt, pt).cpu().numpy().view(np.uint64)
    st = ctx.stats()
    assert int(v[0]) == m and m == n - orphans                         # every emitted pair joins equal keys; none missing
    if orphans == 0:
        assert int(v[2]) == n * (n - 1) // 2                           # every probe row exactly once
    wide_ms, narrow_ms = st["join_build_probe_lds_wide"]["total_ms"], st["join_build_probe_lds"]["total_ms"]
    # (negative_key_first: the first build key is the dictionary's first entry, every other key's upper half its second -- tuples;
    #  one_wide_key_last: no sample sees the odd key, the first pass meets it in its last tile and hands over)
    tuples = shape in ("narrow", "common_upper_half", "negative_small", "two_upper_halves", "negative_and_positive", "four_upper_halves", "negative_key_first",
                       "hundred_consecutive_halves", "halves_minus_three_to_forty")
    assert (wide_ms < narrow_ms) if tuples else (wide_ms > narrow_ms)   # which pipeline did the work (the other returns at once)


@pytest.mark.parametrize("bits", ["8,4,4", "10,6", "8,7"])
def test_forced_pass_structures(ctx, bits):
    """every pass structure must give the same pairs (GOLAP_PASS_BITS experiment hook)."""
    n = 2_000_000
    b = oracle.gen_sm64_mod(21, 0, n, n)
    p = oracle.gen_sm64_mod(22, 0, n, n)
    with ctx.knobs(GOLAP_PASS_BITS=bits):
        check(ctx, b, p, N.JOIN_RADIX_HASH)


def test_generators_match_oracle(ctx):
    n = 100_000
    assert np.array_equal(ctx.generate(N.GEN_BUILD_KEYS, n, modulus=1_000_003, first=5).cpu().numpy(), oracle.gen_build_keys(5, n, 1_000_003))
    assert np.array_equal(ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=999, seed=43, first=17).cpu().numpy(), oracle.gen_probe_keys(17, n, 999))
    assert np.array_equal(ctx.generate(N.GEN_SM64_MOD_I32, n, modulus=1024, seed=7).cpu().numpy(), oracle.gen_sm64_mod(7, 0, n, 1024, np.int32))
    assert np.array_equal(ctx.generate(N.GEN_SM64_MOD_F64, n, modulus=1024, seed=8).cpu().numpy(), oracle.gen_sm64_mod(8, 0, n, 1024, np.float64))


def test_radix_partition_standalone(ctx):
    """radix_partition_kernel (join_kernels.cuh:45-76): same partition contents as the oracle, any order inside."""
    n = 500_000
    keys = oracle.gen_sm64_mod(31, 0, n, 10**12) - 5 * 10**11
    kt = torch.from_numpy(keys).cuda()
    ok, orr, off = ctx.radix_partition(kt, rid_base=100)
    torch.cuda.synchronize()
    off = off.cpu().numpy().astype(np.int64)
    ek, er, eoff = oracle.radix_partition(keys)
    assert np.array_equal(off, eoff.astype(np.int64))
    gk, gr = ok.cpu().numpy(), orr.cpu().numpy().view(np.uint32)
    assert np.array_equal(keys[(gr - 100).astype(np.int64)], gk), "row ids do not point at their keys"
    pid = oracle.hash_int64_array(gk) & 255
    for p in (0, 1, 100, 255):
        assert (pid[off[p]:off[p + 1]] == p).all()
        assert np.array_equal(np.sort(gr[off[p]:off[p + 1]] - 100), np.sort(er[eoff[p]:eoff[p + 1]]))
    assert np.array_equal(np.sort(gr), np.arange(100, n + 100, dtype=np.uint32))


@pytest.mark.parametrize("n_owners", [2, 8])
def test_radix_partition_owner_major(ctx, n_owners):
    n = 200_000
    keys = oracle.gen_sm64_mod(32, 0, n, 10**9)
    ok, orr, off = ctx.radix_partition(torch.from_numpy(keys).cuda(), n_owners=n_owners)
    torch.cuda.synchronize()
    off = off.cpu().numpy().astype(np.int64)
    pid = (oracle.hash_int64_array(ok.cpu().numpy()) & 255).astype(np.int64)
    per = 256 // n_owners
    slot = (pid % n_owners) * per + pid // n_owners
    assert (np.diff(slot) >= 0).all(), "slots not contiguous / owner-major"
    counts = np.bincount(slot, minlength=256)
    assert np.array_equal(np.diff(off), counts)


@pytest.mark.parametrize("extra_bits", [1, 2])
def test_radix_partition_extra_bits(ctx, extra_bits):
    """exchange-time partition refined by further hash bits: bucket = slot(p) << e | next e hash bits."""
    n, P = 300_000, 4
    keys = oracle.gen_sm64_mod(33, 0, n, 10**10)
    ok, orr, off = ctx.radix_partition(torch.from_numpy(keys).cuda(), n_owners=P, extra_bits=extra_bits, rid_base=7)
    torch.cuda.synchronize()
    off = off.cpu().numpy().astype(np.int64)
    assert off.size == (256 << extra_bits) + 1 and off[-1] == n
    gk = ok.cpu().numpy()
    h = oracle.hash_int64_array(gk).astype(np.int64)
    p = h & 255
    bucket = (((p % P) * (256 // P) + p // P) << extra_bits) | ((h >> 8) & ((1 << extra_bits) - 1))
    assert (np.diff(bucket) >= 0).all()
    assert np.array_equal(np.diff(off), np.bincount(bucket, minlength=256 << extra_bits))
    assert np.array_equal(keys[(orr.cpu().numpy().view(np.uint32) - 7).astype(np.int64)], gk)


def test_radix_partition_narrow_tuples(ctx):
    """narrow = 1: the same partition written as ONE array of (key32 | row id << 32) tuples; wide_flag reports keys outside [0, 2^32)."""
    n, P, e1 = 400_000, 4, 1
    keys = oracle.gen_sm64_mod(35, 0, n, 3_000_000_000)              # needs all 32 bits, but no more
    kt = torch.from_numpy(keys).cuda()
    ok, orr, off = ctx.radix_partition(kt, n_owners=P, extra_bits=e1, rid_base=11)
    tup, none, off2, flag = ctx.radix_partition(kt, n_owners=P, extra_bits=e1, rid_base=11, narrow=True)
    torch.cuda.synchronize()
    assert none is None and int(flag.item()) == 0 and torch.equal(off, off2)
    t = tup.cpu().numpy().view(np.uint64)
    w = ok.cpu().numpy().astype(np.uint64) | (orr.cpu().numpy().view(np.uint32).astype(np.uint64) << np.uint64(32))
    bid = np.repeat(np.arange(off.numel() - 1), np.diff(off.cpu().numpy().astype(np.int64)))
    # the same rows in the same buckets (the order inside a bucket is that of the tile's LDS ranks: not fixed)
    assert np.array_equal(t[np.lexsort((t, bid))], w[np.lexsort((w, bid))])
    for bad in (-1, 1 << 32):
        k2 = kt.clone()
        k2[n // 2] = bad
        assert int(ctx.radix_partition(k2, n_owners=P, extra_bits=e1, narrow=True)[3].item()) == 1


@pytest.mark.parametrize("narrow", [False, True])
def test_join_partitioned_split_api_with_extra_bits(ctx, narrow):
    """begin / side / side / finish over buckets of 8+1 bits, 2 sources, equals the plain join; also with the rows as tuples
    (the tuple form exists for local fan-outs >= 32, which real inputs reach from ~25M rows per rank: forced here)."""
    with ctx.knobs(**({"GOLAP_LOCAL_BITS": "6"} if narrow else {})):
        _split_api_case(ctx, narrow)


def _split_api_case(ctx, narrow):
    n, P, e1 = 500_000, 2, 1
    b = oracle.gen_build_keys(0, n, n)
    p = oracle.gen_probe_keys(0, n, n)
    exp, me = oracle.hash_join(b, p)
    per = (256 // P) << e1
    h = n // P
    parts = {}
    for name, keys in (("b", b), ("p", p)):
        parts[name] = [ctx.radix_partition(torch.from_numpy(keys[r * h:(r + 1) * h]).cuda(), rid_base=r * h, n_owners=P, extra_bits=e1, narrow=narrow)
                       for r in range(P)]
    torch.cuda.synchronize()
    # (golap_join_partitioned_narrow_ok is a context-free query: it answers for the production rule -- tuple local passes exist
    #  from fan-out 32, i.e. from ~25M rows per rank -- and cannot see the per-context knob that forces them in this small case)
    assert not ctx.join_partitioned_narrow_ok(((256 // P) << e1), P, 8 + e1, n // P)
    assert ctx.join_partitioned_narrow_ok(((256 // P) << e1), P, 8 + e1, 500_000_000)
    allpairs = []
    for owner in range(P):
        sides = {}
        for name in ("b", "p"):
            offs = [x[2].cpu().numpy().astype(np.int64) for x in parts[name]]
            ks, rs, so, sl, pos, srcpos = [], [], [], [], 0, []
            for r in range(P):
                lo, hi = offs[r][owner * per], offs[r][(owner + 1) * per]
                ks.append(parts[name][r][0][lo:hi]); srcpos.append(pos); pos += hi - lo
                if not narrow:
                    rs.append(parts[name][r][1][lo:hi])
            for s_ in range(per):
                for r in range(P):
                    so.append(int(srcpos[r] + offs[r][owner * per + s_] - offs[r][owner * per]))
                    sl.append(int(offs[r][owner * per + s_ + 1] - offs[r][owner * per + s_]))
            sides[name] = (torch.cat(ks).contiguous(), None if narrow else torch.cat(rs).contiguous(), so, sl)
        nb_, np_ = sides["b"][0].numel(), sides["p"][0].numel()
        ctx.join_partitioned_begin(per, P, 8 + e1, n // P, nb_, np_, narrow=narrow)
        ctx.join_partitioned_side(1, *sides["p"], rows=np_)     # probe side first: order must not matter
        ctx.join_partitioned_side(0, *sides["b"], rows=nb_)
        out = torch.empty((np_ + 16, 2), dtype=torch.int32, device="cuda")
        pairs, cnt = ctx.join_partitioned_finish(out)
        torch.cuda.synchronize()
        m = int(cnt.item())
        allpairs.append(pairs[:m].cpu().numpy().view(np.uint32))
    got = canonical(np.concatenate(allpairs))
    assert got.shape[0] == me and np.array_equal(got, exp)
    from gpu_olap_engine_amd.engine import GolapError
    with pytest.raises(GolapError):
        ctx.join_partitioned_finish(out)          # finish without begin


def test_bench_sharded_path_single_rank():
    """bench.py --sharded with one rank: the whole partition -> all_to_all (RCCL, self) -> local join path at 20M rows."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rows", "20000000", "--steps", "2", "--warmup", "1", "--sharded", "--cpu-sample", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["verify"]["matches"] == 20_000_000 and d["verify"]["probe_id_sum_ok"] and d["verify"]["equal_key_pairs"] == 20_000_000


def test_bench_sharded_group_by_two_ranks_rehearsed():
    """bench.py --config C4L --gpus 2 --rehearse: the row-sharded GROUP BY through golap_sharded_aggregate, two ranks on this GPU over
    the staged transport (the launch line the driver would use with RCCL, minus --rehearse); the line must verify its own result."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29543",
                          os.path.join(root, "bench.py"), "--gpus", "2", "--config", "C4L", "--rows", "50000000", "--rehearse", "--steps", "2", "--warmup", "1",
                          "--cpu-sample", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["verify"]["groups_equal_index_add"] and d["scaling"] == "strong"


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher on the command line and no WORLD_SIZE: bench.py starts its two ranks itself
    (a child torch.distributed.run), the join runs through golap_sharded_join on both (rehearsed on this one GPU) and this process
    prints rank 0's line -- with every rank's stage timestamps and what its exchange moved."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse", "--rows", "40000000", "--steps", "2", "--warmup", "1",
                          "--cpu-sample", "0"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # ONE JSON line, nothing else on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["verify"]["matches"] == 40_000_000 and d["verify"]["probe_id_sum_ok"]
    sj = d["sharded_join"]
    assert len(sj["per_rank"]) == 2 and all(len(r["bytes_sent"]) == 2 and "stage_ms" in r for r in sj["per_rank"])
    if sj["fast_path"]:
        assert sj["exchange"]["build_per_link_GBps"] > 0 and sj["exchange"]["probe_per_link_GBps"] > 0


def test_bench_draws_workspaces_and_keeps_the_fastest():
    """bench.py --workspace-draws K (single-GPU joins): K contexts -- K physical placements of the workspace slab -- are tried before the
    warm-up, the fastest is kept, every draw is in the line (DESIGN 9 'Run-to-run spread'); --workspace-draws 1 takes the first as
    it comes.  The line still verifies its own result."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for k in (3, 1):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "C3", "--rows", "30000000", "--steps", "2", "--warmup", "1", "--cpu-sample", "0",
                              "--workspace-draws", str(k)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert d["verify"]["matches"] == 30_000_000 and d["verify"]["probe_id_sum_ok"]
        if k == 1:
            assert "workspace_draws" not in d
        else:
            w = d["workspace_draws"]
            assert w["draws"] == k and len(w["ms_per_join"]) == k and w["ms_per_join"][w["kept"]] == min(w["ms_per_join"]) and "stopped_by" not in w


def test_join_partitioned_two_sources(ctx):
    """post-exchange path: 2 sources x 128 slots on one device must equal the plain join."""
    n = 600_000
    b = oracle.gen_build_keys(0, n, n)
    p = oracle.gen_probe_keys(0, n, n)
    exp, me = oracle.hash_join(b, p)
    P = 2
    per = 256 // P

    def shard_and_partition(keys):
        h = n // P
        parts = []
        for r in range(P):
            k = torch.from_numpy(keys[r * h:(r + 1) * h]).cuda()
            parts.append(ctx.radix_partition(k, rid_base=r * h, n_owners=P))
        torch.cuda.synchronize()
        return parts

    bp, pp = shard_and_partition(b), shard_and_partition(p)
    total = 0
    allpairs = []
    for owner in range(P):
        def gather(parts):
            ks, rs, so, sl = [], [], [], []
            pos = 0
            offs = [x[2].cpu().numpy().astype(np.int64) for x in parts]
            srcpos = []
            for r in range(P):
                lo, hi = offs[r][owner * per], offs[r][(owner + 1) * per]
                ks.append(parts[r][0][lo:hi]); rs.append(parts[r][1][lo:hi])
                srcpos.append(pos); pos += hi - lo
            for s in range(per):
                for r in range(P):
                    a = offs[r][owner * per + s] - offs[r][owner * per]
                    so.append(srcpos[r] + a)
                    sl.append(offs[r][owner * per + s + 1] - offs[r][owner * per + s])
            return torch.cat(ks).contiguous(), torch.cat(rs).contiguous(), [int(x) for x in so], [int(x) for x in sl]
        bk, br, bso, bsl = gather(bp)
        pk, pr, pso, psl = gather(pp)
        out = torch.empty((max(1, pk.numel()) * 2, 2), dtype=torch.int32, device="cuda")
        pairs, cnt = ctx.join_partitioned(bk, br, bso, bsl, pk, pr, pso, psl, per, P, out, total_build_rows_hint=n // P)
        torch.cuda.synchronize()
        m = int(cnt.item())
        total += m
        allpairs.append(pairs[:m].cpu().numpy().view(np.uint32))
    assert total == me
    assert np.array_equal(canonical(np.concatenate(allpairs)), exp)


def test_verify_pairs_and_determinism(ctx):
    """size-independent properties used for the 1B-row runs + set-determinism across runs."""
    n = 3_000_000
    bt = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n)
    pt = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
    sums = []
    for _ in range(2):
        pairs, cnt = ctx.join(bt, pt, max_pairs=n, strategy=N.JOIN_RADIX_HASH)
        v = ctx.verify_pairs(pairs, n, bt, pt).cpu().numpy().view(np.uint64)
        torch.cuda.synchronize()
        assert int(cnt.item()) == n
        assert int(v[0]) == n                        # every emitted pair has equal keys
        assert int(v[2]) == n * (n - 1) // 2         # every probe row exactly once (with M == N)
        sums.append((int(v[1]), int(v[3])))
    assert sums[0] == sums[1]
    exp, _ = oracle.hash_join(bt.cpu().numpy(), pt.cpu().numpy())
    assert sums[0][0] == oracle.pairs_checksum(exp)


@pytest.mark.parametrize("strategy", STRATS)
def test_join_is_hip_graph_capturable(ctx, strategy):
    """No allocation / synchronisation on the warm launch path: a join can be captured into a HIP graph and replayed."""
    n = 300_000
    bt = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n)
    pt = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
    out = torch.empty((n, 2), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx.join(bt, pt, out_pairs=out, max_pairs=n, strategy=strategy, out_count=cnt)   # warms the workspace
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        ctx.join(bt, pt, out_pairs=out, max_pairs=n, strategy=strategy, out_count=cnt)
    torch.cuda.synchronize()
    exp, me = oracle.hash_join(bt.cpu().numpy(), pt.cpu().numpy())
    for _ in range(2):
        out.zero_(); cnt.zero_()
        g.replay()
        torch.cuda.synchronize()
        m = int(cnt.item())
        assert m == me
        assert np.array_equal(canonical(out[:m].cpu().numpy().view(np.uint32)), exp)


def test_large_join_graph_replays_pick_the_pipeline_on_the_device(ctx):
    """The narrow-then-64-bit scheme decides on the device, so ONE captured graph serves both: replayed over narrow keys the
    tuple pipeline does the join, replayed after the same buffers were refilled with wide keys the 64-bit pipeline does."""
    n = 60_000_000
    bt = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n)
    pt = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
    out = torch.empty((n, 2), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx.join(bt, pt, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH, out_count=cnt)   # warms the workspace
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        ctx.join(bt, pt, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH, out_count=cnt)
    torch.cuda.synchronize()
    for shift in (0, 1 << 40, 0):
        bt += shift
        pt += shift
        out.zero_(); cnt.zero_()
        g.replay()
        torch.cuda.synchronize()
        v = ctx.verify_pairs(out, n, bt, pt).cpu().numpy().view(np.uint64)
        assert int(cnt.item()) == n and int(v[0]) == n and int(v[2]) == n * (n - 1) // 2, shift
        bt -= shift
        pt -= shift


EDGE_SIZES = [1, 63, 64, 65, 511, 512, 4095, 4096, 4097, 8191, 8192, 8193, 12287, 12288, 12289, 24576, 49153, 131071, 262145]


@pytest.mark.parametrize("bits", [None, "8,5", "9,6", "10,5", "8,4,5"])
def test_tile_and_chunk_boundaries(ctx, bits):
    """sizes straddling every tile / window / staging boundary of the scatter kernels (4096, 8192, 12288 rows, 16-row lines),
    under forced pass structures (plain, no-histogram second pass, three passes), inner and full outer, vs the oracle."""
    rng = np.random.default_rng(sum(map(ord, bits or "default")))   # (a fixed seed per case: str hashes change per process)
    with ctx.knobs(**({"GOLAP_PASS_BITS": bits} if bits else {})):
        for trial in range(len(EDGE_SIZES)):
            nb = EDGE_SIZES[trial]
            npr = EDGE_SIZES[(trial * 7 + 3) % len(EDGE_SIZES)]
            span = max(4, (nb + npr) // 3)
            b = rng.integers(-span, span, nb).astype(np.int64)
            p = rng.integers(-span, span, npr).astype(np.int64)
            outer_check(ctx, b, p, N.JOIN_RADIX_HASH, N.JT_FULL if trial % 2 else N.JT_INNER)


def test_sort_merge_unsupported(ctx):
    from gpu_olap_engine_amd.engine import GolapError
    t = torch.zeros(4, dtype=torch.int64, device="cuda")
    with pytest.raises(GolapError) as e:
        ctx.join(t, t, strategy=N.JOIN_SORT_MERGE)
    assert e.value.code == N.ERR_UNSUPPORTED


# ---- composite keys and NULLs ------------------------------------------------------------------
def _multi_check(ctx, bcols, pcols, jt, strategy, bvalid=None, pvalid=None, knobs=None):
    exp, me = oracle.hash_join_multi(bcols, pcols, jt, bvalid, pvalid)
    tb = [torch.from_numpy(np.ascontiguousarray(c, dtype=np.int64)).cuda() for c in bcols]
    tp = [torch.from_numpy(np.ascontiguousarray(c, dtype=np.int64)).cuda() for c in pcols]
    tv = lambda v: None if v is None else [None if x is None else torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint8)).cuda() for x in v]
    with ctx.knobs(**(knobs or {})):
        pairs, cnt = ctx.join(tb, tp, max_pairs=max(1, me + 64), strategy=strategy, join_type=jt, build_valid=tv(bvalid), probe_valid=tv(pvalid))
        torch.cuda.synchronize()
    m = int(cnt.item())
    assert m == me, f"jt={jt}: count {m} != oracle {me}"
    assert np.array_equal(canonical(pairs[:m].cpu().numpy().view(np.uint32)), exp), f"jt={jt}"


@pytest.mark.parametrize("strategy", STRATS)
@pytest.mark.parametrize("jt", [N.JT_INNER, N.JT_LEFT, N.JT_RIGHT, N.JT_FULL])
def test_composite_key_join(ctx, strategy, jt):
    """physical_plan.rs:35-36: left_keys / right_keys are vectors; rows match when EVERY key column is equal (the reference
    executor only reads [0], executor.rs:245).  Two and three Int64 columns, duplicates, values of either sign and width."""
    rng = np.random.default_rng(41 + jt)
    nb, npr = 60_000, 90_000
    b = [rng.integers(-40, 40, nb), rng.integers(0, 30, nb) << 35]
    p = [rng.integers(-40, 40, npr), rng.integers(0, 30, npr) << 35]
    _multi_check(ctx, b, p, jt, strategy)
    b3 = b + [rng.integers(0, 3, nb)]
    p3 = p + [rng.integers(0, 3, npr)]
    _multi_check(ctx, b3, p3, jt, strategy)
    # the same first column everywhere: only the second one decides
    _multi_check(ctx, [np.full(1000, 7), np.arange(1000)], [np.full(500, 7), np.arange(250, 750)], jt, strategy)


@pytest.mark.parametrize("strategy", STRATS)
@pytest.mark.parametrize("jt", [N.JT_INNER, N.JT_LEFT, N.JT_RIGHT, N.JT_FULL])
def test_null_keys_match_nothing(ctx, strategy, jt):
    """column_buffer.rs:59-60 validity (a byte per row): a NULL key matches nothing -- not even another NULL -- and on the preserved
    side of an outer join the row survives as an unmatched row (SQL semantics; the reference turns nulls into key 0)."""
    rng = np.random.default_rng(17 + jt)
    nb, npr = 50_000, 70_000
    b, p = rng.integers(0, 20_000, nb), rng.integers(0, 20_000, npr)
    bv, pv = (rng.random(nb) > 0.1).astype(np.uint8), (rng.random(npr) > 0.2).astype(np.uint8)
    b[bv == 0] = 0                                  # what ingest leaves in a NULL slot (record_batch_convert.rs:102-110): must NOT join with key 0
    p[pv == 0] = 0
    _multi_check(ctx, [b], [p], jt, strategy, [bv], [pv])
    _multi_check(ctx, [b], [p], jt, strategy, [bv], None)          # NULLs on one side only
    _multi_check(ctx, [b], [p], jt, strategy, None, [pv])
    _multi_check(ctx, [b, b // 3], [p, p // 3], jt, strategy, [None, bv], [pv, None])   # composite key, NULLs in different columns
    _multi_check(ctx, [b], [p], jt, strategy, [np.zeros(nb, np.uint8)], [pv])           # a side that is all NULL


@pytest.mark.parametrize("jt", [N.JT_INNER, N.JT_LEFT, N.JT_FULL])
def test_surrogate_collisions_are_repaired(ctx, jt):
    """The composite key travels as a 64-bit surrogate; every emitted pair is verified against the real columns and a false pair
    (probability 2^-64 per candidate pair) is removed -- in an outer join its rows get their unmatched pairs back.  The knob
    narrows the surrogate so that false pairs actually happen (a handful, below the repair's capacity of 64)."""
    rng = np.random.default_rng(23)
    nb, npr = 3000, 3000
    b = [np.arange(nb) * 2, rng.integers(0, 5, nb)]                        # build keys even, probe keys odd in column 0: NOTHING really matches ...
    p = [np.arange(npr) * 2 + 1, rng.integers(0, 5, npr)]
    _multi_check(ctx, b, p, jt, N.JOIN_RADIX_HASH, knobs={"GOLAP_GJ_HASH_BITS": "19"})      # ... but ~17 surrogates collide
    b[0][:100] = p[0][:100]                                                # a few true matches beside the false ones
    b[1][:100] = p[1][:100]
    _multi_check(ctx, b, p, jt, N.JOIN_BROADCAST_HASH, knobs={"GOLAP_GJ_HASH_BITS": "20"})


@pytest.mark.parametrize("strategy", STRATS)
def test_more_false_pairs_than_can_be_repaired_give_the_impossible_count(ctx, strategy):
    """include/golap.h: when the verification of a surrogate-key join finds more false pairs than its repair list holds (64; with 64-bit
    surrogates that has probability < 1e-60), the count word is set to 2^64 - 1 -- a count no join can have -- instead of a list that
    silently keeps false pairs.  The knob narrows the surrogates to 8 bits: nothing really matches, ~35 000 candidate pairs do."""
    rng = np.random.default_rng(29)
    nb, npr = 3000, 3000
    b = [np.arange(nb) * 2, rng.integers(0, 5, nb)]
    p = [np.arange(npr) * 2 + 1, rng.integers(0, 5, npr)]
    tb = [torch.from_numpy(np.ascontiguousarray(c, dtype=np.int64)).cuda() for c in b]
    tp = [torch.from_numpy(np.ascontiguousarray(c, dtype=np.int64)).cuda() for c in p]
    with ctx.knobs(GOLAP_GJ_HASH_BITS="8"):
        pairs, cnt = ctx.join(tb, tp, max_pairs=200_000, strategy=strategy, join_type=N.JT_INNER)
        torch.cuda.synchronize()
    assert int(cnt.item()) == -1                                          # 0xFFFFFFFFFFFFFFFF read as int64
    pairs, cnt = ctx.join(tb, tp, max_pairs=200_000, strategy=strategy, join_type=N.JT_INNER)   # the knob gone: no candidate, no pair
    assert int(cnt.item()) == 0


@pytest.mark.parametrize("jt", [N.JT_INNER, N.JT_FULL])
def test_composite_key_join_four_full_range_columns(ctx, jt):
    """four Int64 key columns whose values span all of int64 (hash-like ids; a mixed-radix packing would need ~250 bits), NULLs in two of
    them: surrogate keys + verification, against the oracle's multi-column NULL-aware join."""
    rng = np.random.default_rng(77 + jt)
    pool = [rng.integers(-2**63, 2**63 - 1, 40) for _ in range(4)]
    nb, npr = 30_000, 45_000
    tb = rng.integers(0, 40, (4, nb))
    tp = rng.integers(0, 40, (4, npr))
    tb[2:] = tb[:2] % 7                                   # (correlated columns: ~280 distinct tuples, so that tuples do meet)
    tp[2:] = tp[:2] % 7
    b = [pool[c][tb[c]] for c in range(4)]
    p = [pool[c][tp[c]] for c in range(4)]
    bv = [None, (rng.random(nb) > 0.05).astype(np.uint8), None, None]
    pv = [None, None, (rng.random(npr) > 0.05).astype(np.uint8), None]
    _multi_check(ctx, b, p, jt, N.JOIN_RADIX_HASH, bv, pv)
    _multi_check(ctx, b, p, jt, N.JOIN_BROADCAST_HASH, bv, pv)


def test_composite_key_join_large_radix(ctx):
    """4M x 6M rows, two key columns, through the radix path (surrogates are 64-bit: the 64-bit-row pipeline) -- device-checked."""
    n, m = 4_000_000, 6_000_000
    rng = np.random.default_rng(5)
    b = [np.arange(n) % 100_000, np.arange(n) // 100_000]                  # unique (c0, c1) pairs
    pi = rng.integers(0, n, m)
    p = [b[0][pi], b[1][pi]]
    _multi_check(ctx, b, p, N.JT_INNER, N.JOIN_RADIX_HASH)


def test_key_columns_that_are_only_8_byte_aligned(ctx):
    """The first pass reads its keys in 16-byte words when the column allows it; a column that starts 8 bytes into an allocation (a
    sliced Arrow buffer) takes the 8-byte form of the same kernel -- both sides independently, at a size that runs the tuple
    pipeline (checked on the device: count, key equality of every pair, probe ids a permutation)."""
    n = 20_000_003
    bfull = ctx.generate(N.GEN_BUILD_KEYS, n + 1, modulus=n)
    pfull = ctx.generate(N.GEN_SM64_MOD_I64, n + 3, modulus=n, seed=43)
    for b, p in ((bfull[1:], pfull[:n]), (bfull[:n], pfull[3:]), (bfull[1:], pfull[1:n + 1])):
        assert b.data_ptr() % 16 in (0, 8) and p.numel() == n
        bk = b.clone()                                   # (the generator's key of row i is a function of i: re-derive the expected pairs from the actual columns)
        out = torch.empty((2 * n, 2), dtype=torch.int32, device="cuda")
        pairs, cnt = ctx.join(b, p, out_pairs=out, max_pairs=2 * n, strategy=N.JOIN_RADIX_HASH)
        torch.cuda.synchronize()
        m = int(cnt.item())
        # expected match count: build keys are distinct except possibly the one row that wrapped; count via sort
        ub, cb = torch.unique(bk, return_counts=True)
        pos = torch.searchsorted(ub, p)
        pos = pos.clamp(max=ub.numel() - 1)
        hit = ub[pos] == p
        want = int(cb[pos][hit].sum().item())
        assert m == want
        v = ctx.verify_pairs(pairs, m, b, p).cpu()
        assert int(v[0]) == m


def _pairs_by_probe(ctx, bt, pt, n, **knobs):
    """build row id of every probe row (unique build keys: exactly one partner each), -1 where missing, -2 where seen twice"""
    out = torch.empty((n, 2), dtype=torch.int32, device="cuda")
    with ctx.knobs(**knobs):
        pairs, cnt = ctx.join(bt, pt, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH)
        torch.cuda.synchronize()
    m = int(cnt.item())
    got = torch.full((n,), -1, dtype=torch.int64, device="cuda")
    seen = torch.zeros(n, dtype=torch.int32, device="cuda")
    pr = pairs[:min(m, n), 1].to(torch.int64) & 0xFFFFFFFF
    seen.index_add_(0, pr, torch.ones_like(pr, dtype=torch.int32))
    got[pr] = pairs[:min(m, n), 0].to(torch.int64) & 0xFFFFFFFF
    got[seen > 1] = -2
    return m, got


@pytest.mark.parametrize("bits,ring", [("9,9", 32), ("9,8", 64), ("9,7", 128), ("9,6", 256)])
def test_hidden_forms_match_plain_forms(ctx, bits, ring):
    """The partition kernels that hide their loads and stores from hipcc (hand-counted vmcnt waits, fixed registers: the ASM forms of
    k_part1_narrow and k_part2_ring) against their plain forms, row by row, at every ring length (the second fan-out picks it:
    512 -> 32 ... 64 -> 256 rows) and with a partial last tile in every chunk and group (n is not a multiple of any tile).  With
    unique build keys the partner of every probe row is known in closed form, so a row that either form loses, duplicates, garbles
    or files under the wrong sub-partition shows up as a probe row without exactly its partner."""
    n = 24_000_011
    bt = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n)
    pt = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
    inv = torch.empty(n, dtype=torch.int64, device="cuda")
    inv[bt] = torch.arange(n, dtype=torch.int64, device="cuda")
    exp = inv[pt]
    del inv
    base = {"GOLAP_PASS_BITS": bits, "GOLAP_RESERVE": "1"}
    forms = {"hidden p1 + hidden ring": {"GOLAP_P1N": "118", "GOLAP_P2R": "104"}, "plain p1 + hidden ring": {"GOLAP_P1N": "18", "GOLAP_P2R": "104"},
             "hidden p1 + plain ring": {"GOLAP_P1N": "118", "GOLAP_P2R": "4"}, "plain p1 + plain ring": {"GOLAP_P1N": "18", "GOLAP_P2R": "4"}}
    for name, kn in forms.items():
        ctx.stats_reset()
        m, got = _pairs_by_probe(ctx, bt, pt, n, **base, **kn)
        st = ctx.stats()
        assert st["join_build_probe_lds"]["total_ms"] > st["join_build_probe_lds_wide"]["total_ms"], f"{name}: the tuple pipeline did not run"
        bad = torch.nonzero(got != exp).flatten()
        assert m == n and bad.numel() == 0, (f"{name} (ring {ring}): {m} pairs for {n} probe rows; {bad.numel()} probe rows without exactly their partner, "
                                             f"first {bad[:8].tolist()} -> {got[bad[:8]].tolist()}")


@pytest.mark.parametrize("bits", ["9,9", "9,7", "8,6"])
def test_wide_last_pass_forms_match(ctx, bits):
    """The 64-bit-row pipeline's no-histogram last pass: k_part2_wide (csrc/golap_part_wide.h) with hidden loads and stores (GOLAP_P2W=1,
    the default of an audited build), with compiler-visible ones (2) and round 1's write-combining k_scatter_wc<2, FIXED> (0), row by
    row -- behind the reserving first pass (groups = per-XCD pieces, every piece ends in a partial tile) and behind the exact first
    pass (GOLAP_RESERVE=0: groups start at odd rows, so every tile takes the clamped 8- and 4-byte loads).  Keys with 256 scattered
    upper halves: no 8-byte tuple can carry them."""
    n = 24_000_011
    bt = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n)
    pt = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
    inv = torch.empty(n, dtype=torch.int64, device="cuda")
    inv[bt] = torch.arange(n, dtype=torch.int64, device="cuda")
    exp = inv[pt]
    del inv
    bt += ((bt * 0x9E3779B1) & 0xFF) << 44
    pt += ((pt * 0x9E3779B1) & 0xFF) << 44
    for reserve in ("1", "0"):
        for form in ("1", "2", "0"):
            ctx.stats_reset()
            m, got = _pairs_by_probe(ctx, bt, pt, n, GOLAP_PASS_BITS=bits, GOLAP_RESERVE=reserve, GOLAP_P2W=form)
            st = ctx.stats()
            p2 = [k for k, v in st.items() if "scatter_p2" in k and "standby" not in k and v["total_ms"] > 0.05]
            assert p2, f"no last-pass scatter in the profile: {sorted(st)}"
            bad = torch.nonzero(got != exp).flatten()
            assert m == n and bad.numel() == 0, (f"bits {bits} reserve {reserve} GOLAP_P2W={form}: {m} pairs for {n} probe rows; {bad.numel()} probe rows without "
                                                 f"exactly their partner, first {bad[:8].tolist()} -> {got[bad[:8]].tolist()}")


def test_a_build_by_an_unaudited_compiler_runs_the_plain_forms(ctx):
    """GOLAP_HIDDEN_FORMS=0 flips the one switch a compiler mismatch flips (csrc/golap_variants.h: a library built by a hipcc other
    than the audited one): golap_device_info says so, every one of the five hand-scheduled kernels takes its plain form -- tuple
    pipeline, 64-bit-row pipeline, dictionary columns (which then take the 64-bit rows: the dictionary forms exist hidden only) and
    the GROUP BY scatter -- and every result is still exact."""
    info = ctx.device_info()
    assert info["hidden_memory_forms"] == 1 and info["built_with"] == info["audited_with"] and info["built_with"].startswith("hip ")
    n = 24_000_011
    bt = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n)
    pt = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
    inv = torch.empty(n, dtype=torch.int64, device="cuda")
    inv[bt] = torch.arange(n, dtype=torch.int64, device="cuda")
    exp = inv[pt]
    del inv
    with ctx.knobs(GOLAP_HIDDEN_FORMS="0"):
        assert ctx.device_info()["hidden_memory_forms"] == 0
        for shape in ("narrow", "two_upper_halves", "many_upper_halves"):
            b, p = bt.clone(), pt.clone()
            if shape == "two_upper_halves":
                b += (b & 1) << 40
                p += (p & 1) << 40
            if shape == "many_upper_halves":
                b += ((b * 0x9E3779B1) & 0xFF) << 44
                p += ((p * 0x9E3779B1) & 0xFF) << 44
            ctx.stats_reset()
            m, got = _pairs_by_probe(ctx, b, p, n)
            st = ctx.stats()
            bad = torch.nonzero(got != exp).flatten()
            assert m == n and bad.numel() == 0, f"{shape}: {m} pairs, {bad.numel()} probe rows without exactly their partner"
            tuple_ms, wide_ms = st["join_build_probe_lds"]["total_ms"], st["join_build_probe_lds_wide"]["total_ms"]
            assert (tuple_ms > wide_ms) == (shape == "narrow"), (shape, tuple_ms, wide_ms)
        # the GROUP BY partition scatter: 2^20 groups over 40M rows (k_agg_scatter_rsv's plain form), against torch
        g, nr = 1 << 20, 40_000_000
        keys = ctx.generate(N.GEN_SM64_MOD_I32, nr, modulus=g, seed=7)
        vals = ctx.generate(N.GEN_SM64_MOD_F64, nr, modulus=1024, seed=8)
        out = ctx.aggregate([(N.AGG_SUM, 0)], [vals], group_keys=keys, n_groups=g)
        ref = torch.zeros(g, dtype=torch.float64, device="cuda")
        ref.index_add_(0, keys.to(torch.int64), vals)
        assert torch.equal(out[0], ref)
    assert ctx.device_info()["hidden_memory_forms"] == 1


@pytest.mark.parametrize("sigmas", [8, 12, 40])
def test_large_provision_sigmas_size_the_buffers_too(sigmas):
    """golap_engine_config.provision_sigmas widens the slots of the no-histogram layouts; the buffers behind them must grow with it
    (round 2 sized them for 6 sigmas whatever the setting: with >= 7 the last pass wrote past the final buffers).  Values above
    the library's cap (12) are clamped.  Both pipelines: the tuple one and, with two upper key halves, the 64-bit one."""
    from gpu_olap_engine_amd.engine import Context
    n = 40_000_000
    with Context(max_gpu_memory=0, provision_sigmas=sigmas) as c:
        bt = c.generate(N.GEN_BUILD_KEYS, n, modulus=n)
        pt = c.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
        for wide in (False, True):
            if wide:
                bt += (bt & 1) << 40
                pt += (pt & 1) << 40
            out = torch.empty((n, 2), dtype=torch.int32, device="cuda")
            pairs, cnt = c.join(bt, pt, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH)
            v = c.verify_pairs(pairs, n, bt, pt).cpu().numpy().view(np.uint64)
            assert int(cnt.item()) == n and int(v[0]) == n and int(v[2]) == n * (n - 1) // 2


def test_wide_first_pass_forms_agree(ctx):
    """The 64-bit pipeline's first pass in its three forms -- round 1's k_scatter_big (GOLAP_P1W=0), k_part1_narrow<.., WIDE> with hidden
    loads / stores and counted waits (1, the default) and with compiler-visible ones (2) -- on a key column with 256 upper halves whose
    fold lo ^ hi is still a bijection (no hash collisions: the no-histogram layouts hold), n not a multiple of any tile: every probe row
    must come back with exactly its partner."""
    n = 30_000_011
    bt = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n)
    pt = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
    inv = torch.empty(n, dtype=torch.int64, device="cuda")
    inv[bt] = torch.arange(n, dtype=torch.int64, device="cuda")
    exp = inv[pt]
    del inv
    bt = bt + ((bt & 0xFF) << 36)
    pt = pt + ((pt & 0xFF) << 36)
    for form in ("1", "2", "0"):
        ctx.stats_reset()
        m, got = _pairs_by_probe(ctx, bt, pt, n, GOLAP_P1W=form)
        st = ctx.stats()
        assert st["join_build_probe_lds_wide"]["total_ms"] > st["join_build_probe_lds"]["total_ms"], f"GOLAP_P1W={form}: the 64-bit pipeline did not run"
        assert st["radix_scatter_p1_wide"]["total_ms"] > 0.05 and "radix_hist_p2" not in st, f"GOLAP_P1W={form}: not the no-histogram flow"
        bad = torch.nonzero(got != exp).flatten()
        assert m == n and bad.numel() == 0, f"GOLAP_P1W={form}: {m} pairs for {n} probe rows; {bad.numel()} probe rows without exactly their partner"


def test_light_profiling_times_the_stand_by_pipeline_only_when_it_is_needed():
    """golap_engine_config.enable_profiling = 2 (what bench.py uses): the kernels that do work are always timed; the 64-bit pipeline standing
    by behind the tuple pipeline is timed only while the most recent join whose gate word has landed needed it -- one transitional join
    runs it untimed, from then on its kernels are in the stats again.  Results are the same either way."""
    from gpu_olap_engine_amd.engine import Context
    n = 20_000_000
    with Context(profiling=2, max_gpu_memory=0) as c:
        bt = c.generate(N.GEN_BUILD_KEYS, n, modulus=n)
        pt = c.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
        out = torch.empty((n, 2), dtype=torch.int32, device="cuda")

        def join():
            pairs, cnt = c.join(bt, pt, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH)
            torch.cuda.synchronize()
            v = c.verify_pairs(pairs, n, bt, pt).cpu().numpy().view(np.uint64)
            assert int(cnt.item()) == n and int(v[0]) == n and int(v[2]) == n * (n - 1) // 2
        join()                                   # nothing known yet: everything is timed
        st = c.stats()
        assert st["join_build_probe_lds"]["launches"] == 1 and st["join_build_probe_lds_wide"]["launches"] == 1
        c.stats_reset()
        join(); join()                           # the last join's gate word (0) is known: the stand-by pipeline is not timed
        st = c.stats()
        assert st["join_build_probe_lds"]["launches"] == 2 and st["radix_scatter_p1"]["launches"] == 4
        assert st.get("join_build_probe_lds_wide", {"launches": 0})["launches"] == 0 and not any(k.endswith("_standby") and v["launches"] for k, v in st.items())
        bt += ((bt * 0x9E3779B1) & 0xFF) << 44    # many upper key halves: the 64-bit pipeline does the joins from here on
        pt += ((pt * 0x9E3779B1) & 0xFF) << 44
        c.stats_reset()
        join()                                   # transitional: ran wide, untimed
        join()                                   # its gate word (1) is known now
        st = c.stats()
        assert st["join_build_probe_lds_wide"]["launches"] == 1 and st["join_build_probe_lds_wide"]["total_ms"] > st["join_build_probe_lds"]["total_ms"]
        assert st["radix_scatter_p1_wide"]["launches"] == 2 and st["radix_scatter_p1_wide"]["total_ms"] > 0.05
