"""golap_sharded_join (the C++ multi-GPU join, csrc/golap_shard.inc.h) through the C ABI.

The GPU box has ONE device, so more than one rank can only be rehearsed by letting several processes share it: the library's
transport callbacks are served by gloo with host staging (distributed.StagedTransport) -- the SAME C++ orchestration, kernels,
message sizes and segment tables as with the built-in RCCL transport; only the bytes move differently.  RCCL itself is exercised
at world size 1 (communicator setup, all-gather, the status protocol) and by the message-size probe.  Results are compared
bit-exactly with the oracle's canonical pair list."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


def _free_port():
    """the ranks' rendezvous: a fresh FILE (torch FileStore), not a TCP port -- `bind(0)`, close, hand the number to another process is a
    race (seen once in ~450 GPU tests: EADDRINUSE in rank 0, the other ranks waiting for it until the test's timeout); gloo's own
    pair sockets are bound by gloo itself.  (The name and the `port` arguments stay: it is what identifies a rendezvous.)"""
    import atexit
    import tempfile
    import uuid
    path = os.path.join(tempfile.gettempdir(), f"golap_rdzv_{os.getpid()}_{uuid.uuid4().hex}")
    atexit.register(lambda: os.path.exists(path) and os.remove(path))
    return path


def _keys(variant, n, lo, cnt):
    import oracle
    if variant == "unique":
        return oracle.gen_build_keys(lo, cnt, n), oracle.gen_probe_keys(lo, cnt, n)
    return oracle.gen_sm64_mod(11, lo, cnt, n // 2), oracle.gen_sm64_mod(12, lo, cnt, n // 2)


def _shape_keys(k, shape):
    """upper key halves as a function of the key VALUE (equal keys stay equal on both sides and on every rank)"""
    if shape == "common_offset":
        return k + (1 << 40)
    if shape == "five_tenants":
        return k + ((k % 5) << 32)
    if shape == "tenants_minus_two_to_sixty":
        return k + (((k % 63) - 2) << 32)
    if shape == "far_halves":
        return k + ((k % 5) << 44)
    if shape == "three_far_halves":
        return k + ((k % 3) << 44)
    return k


def _worker(rank, world, port, n, variant, mode, join_type, wide_rank, max_msg, q, shape=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if mode == "fast":
        os.environ["GOLAP_DEBUG_ENV"] = "1"        # (the library imports GOLAP_* knobs from the environment only when asked to)
        os.environ["GOLAP_SHARD_FAST"] = "1"       # the fast path at test sizes (its size condition is a speed rule, not a correctness one)
        os.environ["GOLAP_LOCAL_BITS"] = "6"       # ... with a ring-sized local fan-out (64) although the tables would fit LDS with less
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        from gpu_olap_engine_amd.distributed import NativeShardedJoin
        from gpu_olap_engine_amd.engine import Context, shard_range
        torch.cuda.set_device(0)
        c = Context(max_gpu_memory=0)
        lo, hi = shard_range(n, rank, world)
        b, p = _keys(variant, n, lo, hi - lo)
        b, p = _shape_keys(b, shape), _shape_keys(p, shape)
        if wide_rank == rank:
            b[-1] = -3                                  # ONE key outside [0, 2^32) on ONE rank: every rank must leave the fast path together
        sj = NativeShardedJoin(c, transport="staged", max_msg_bytes=max_msg)
        out = torch.empty((4 * n + 16, 2), dtype=torch.int32, device="cuda")
        pairs, cnt, info = sj.join(torch.from_numpy(b).cuda(), torch.from_numpy(p).cuda(), lo, lo, n, out_pairs=out, join_type=join_type,
                                   force_exact=(mode == "exact"))
        torch.cuda.synchronize()
        m = int(cnt.item())
        q.put((rank, pairs[:m].cpu().numpy().view(np.uint32).copy(), info))
        dist.barrier()
        sj.close()
        c.close()
    finally:
        dist.destroy_process_group()


def _run(world, n, variant, mode, join_type=0, wide_rank=-1, max_msg=0, shape=None):
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, world, port, n, variant, mode, join_type, wide_rank, max_msg, q, shape)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    return sorted(res, key=lambda r: r[0])


def _expected(n, variant, join_type, world, wide_rank, shape=None):
    import oracle
    b, p = _keys(variant, n, 0, n)
    b, p = _shape_keys(b, shape), _shape_keys(p, shape)
    if wide_rank >= 0:
        from gpu_olap_engine_amd.engine import shard_range
        b[shard_range(n, wide_rank, world)[1] - 1] = -3
    if join_type == 0:
        return oracle.hash_join(b, p)
    return oracle.hash_join_outer(b, p, join_type)


@pytest.mark.parametrize("world,variant,mode", [(2, "dups", "fast"), (4, "unique", "fast"), (2, "unique", "exact"), (4, "dups", "exact")])
def test_sharded_join_ranks_share_one_gpu(world, variant, mode):
    """2 and 4 ranks, both paths: owner-major slices, per-source segments, global row ids -- bit-exact against the oracle."""
    from np_twin import canonical
    n = 3_000_000
    res = _run(world, n, variant, mode)
    exp, me = _expected(n, variant, 0, world, -1)
    got = canonical(np.concatenate([r[1] for r in res]))
    assert got.shape[0] == me and np.array_equal(got, exp)
    for _, _, info in res:
        assert info["fast_path"] == (mode == "fast") and not info["fast_path_abandoned"]
        assert info["host_syncs"] == (2 if mode == "fast" else 6)   # fast: agreement word + status words; exact: 3 agreement rounds, 2 size read-backs, the end


def _worker_big(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        from gpu_olap_engine_amd import _native as N
        from gpu_olap_engine_amd.distributed import NativeShardedJoin
        from gpu_olap_engine_amd.engine import Context, shard_range
        torch.cuda.set_device(0)
        c = Context(max_gpu_memory=0)
        lo, hi = shard_range(n, rank, world)
        b = c.generate(N.GEN_BUILD_KEYS, hi - lo, modulus=n, first=lo)            # a permutation of 0..n-1 over all ranks
        p = c.generate(N.GEN_SM64_MOD_I64, hi - lo, modulus=n, seed=43, first=lo)
        sj = NativeShardedJoin(c, transport="staged")
        cap = int((hi - lo) * 1.3) + 65536
        out = torch.empty((cap, 2), dtype=torch.int32, device="cuda")
        pairs, cnt, info = sj.join(b, p, lo, lo, n, out_pairs=out, max_pairs=cap)
        torch.cuda.synchronize()
        m = int(cnt.item())
        pr = pairs[:m].to(torch.int64) & 0xFFFFFFFF
        # every probe row id at most once on this rank; the pair's keys are equal: recompute both from the generators' formulas
        bk = (2654435761 * pr[:, 0] + 12345) % n
        pk = c.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)[pr[:, 1]]     # (the whole probe column: n is small enough)
        ok = bool((bk == pk).all())
        q.put((rank, m, int(pr[:, 1].sum().item()), ok, info))
        dist.barrier()
        sj.close()
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 64_000_000), (4, 120_000_000)])
def test_sharded_fast_path_at_its_natural_size(world, n):
    """No knobs: well above the fast path's size rule (>= 16M rows per rank), so the plan is the production one (extra bits, 8 sub-ranges per bucket, world x 8 pieces per
    group in the ring pass, 512 MiB message pieces); the ranks share the one GPU and the bytes travel through gloo.  Checked without
    the oracle: unique build keys, so every probe row matches exactly once over all ranks (count and id sum) and every pair's keys
    are equal."""
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker_big, args=(r, world, port, n, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    assert sum(r[1] for r in res) == n
    assert sum(r[2] for r in res) == n * (n - 1) // 2
    assert all(r[3] for r in res)
    for r in res:
        assert r[4]["fast_path"] and r[4]["host_syncs"] == 2 and not r[4]["fast_path_abandoned"]


def test_sharded_join_one_wide_key_sends_every_rank_to_the_exact_path():
    """The fast path's status word is gathered: ONE key outside [0, 2^32) on ONE rank and all ranks redo the join exactly."""
    from np_twin import canonical
    n, world = 2_000_000, 2
    res = _run(world, n, "dups", "fast", wide_rank=1)
    exp, me = _expected(n, "dups", 0, world, 1)
    got = canonical(np.concatenate([r[1] for r in res]))
    assert got.shape[0] == me and np.array_equal(got, exp)
    assert all(r[2]["fast_path_abandoned"] and not r[2]["fast_path"] for r in res)


@pytest.mark.parametrize("world,shape", [(2, "common_offset"), (4, "five_tenants"), (2, "tenants_minus_two_to_sixty"), (2, "three_far_halves"), (2, "far_halves")])
def test_sharded_join_upper_key_halves(world, shape):
    """Keys beyond 2^32 on the partition-exchange path: every rank summarises the upper halves of its shard, the summaries travel with
    the agreement words, and all ranks derive the same frame of reference -- one common upper half, or a window of consecutive ones
    (ids with a tenant prefix: the offset form of the first pass), or up to four far-apart ones (the dictionary form) -- and stay on the
    fast path; five halves too far apart for a window send every rank to the exact path together.  Bit-exact against the oracle either way."""
    from np_twin import canonical
    n = 3_000_000
    res = _run(world, n, "dups", "fast", shape=shape)
    exp, me = _expected(n, "dups", 0, world, -1, shape=shape)
    got = canonical(np.concatenate([r[1] for r in res]))
    assert got.shape[0] == me and np.array_equal(got, exp)
    for _, _, info in res:
        if shape == "far_halves":
            assert info["fast_path_abandoned"] and not info["fast_path"]
        else:
            assert info["fast_path"] and not info["fast_path_abandoned"] and info["host_syncs"] == 2


@pytest.mark.parametrize("mode,join_type,world", [("fast", 0, 2), ("fast", 1, 2), ("fast", 2, 2), ("fast", 3, 2), ("fast", 1, 4), ("fast", 2, 4), ("exact", 1, 2), ("exact", 2, 2),
                                                  ("exact", 3, 4)])
def test_sharded_outer_joins(mode, join_type, world):
    """Inner / Left / Right / Full (logical_plan.rs:53-58) on the sharded path, fast and exact: all rows of a key meet on one rank, so
    unmatched rows of either side are found locally -- by the LDS join's own outer forms, in both group halves of the fast path (a group
    lies in exactly one half: an unmatched row is emitted once).  The fast path keeps its two host synchronisations for every type."""
    from np_twin import canonical
    n = 1_500_000
    res = _run(world, n, "dups", mode, join_type=join_type)
    exp, me = _expected(n, "dups", join_type, world, -1)
    got = canonical(np.concatenate([r[1] for r in res]))
    assert got.shape[0] == me and np.array_equal(got, exp)
    for _, _, info in res:
        if mode == "fast":
            assert info["fast_path"] and not info["fast_path_abandoned"] and info["host_syncs"] == 2, info
        else:
            assert not info["fast_path"]


def _worker_uneven(rank, world, port, n, cuts, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if mode == "fast":
        os.environ["GOLAP_DEBUG_ENV"] = "1"
        os.environ["GOLAP_SHARD_FAST"] = "1"
        os.environ["GOLAP_LOCAL_BITS"] = "6"
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        import oracle
        from gpu_olap_engine_amd.distributed import NativeShardedJoin
        from gpu_olap_engine_amd.engine import Context
        torch.cuda.set_device(0)
        c = Context(max_gpu_memory=0)
        lo_b, hi_b = cuts[0][rank], cuts[0][rank + 1]
        lo_p, hi_p = cuts[1][rank], cuts[1][rank + 1]
        b = oracle.gen_sm64_mod(11, lo_b, hi_b - lo_b, n // 2)
        p = oracle.gen_sm64_mod(12, lo_p, hi_p - lo_p, n // 2)
        sj = NativeShardedJoin(c, transport="staged")
        out = torch.empty((4 * n + 16, 2), dtype=torch.int32, device="cuda")
        pairs, cnt, info = sj.join(torch.from_numpy(b).cuda(), torch.from_numpy(p).cuda(), lo_b, lo_p, n, n, out_pairs=out,
                                   force_exact=(mode == "exact"),
                                   build_rows_per_rank=[cuts[0][r + 1] - cuts[0][r] for r in range(world)],
                                   probe_rows_per_rank=[cuts[1][r + 1] - cuts[1][r] for r in range(world)])
        torch.cuda.synchronize()
        m = int(cnt.item())
        q.put((rank, pairs[:m].cpu().numpy().view(np.uint32).copy(), info))
        dist.barrier()
        sj.close()
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_sharded_join_uneven_shards(mode):
    """Shards of different sizes (and different splits of the two tables): the fast path sizes every message from the SOURCE rank's row
    count, which every rank takes from the rows-per-rank lists -- the receiving side must provision exactly what the sender sends."""
    import oracle
    import torch.multiprocessing as mp
    from np_twin import canonical
    n, world = 2_400_000, 4
    cuts = ([0, 300_000, 1_900_000, 1_900_000, n], [0, 1_100_000, 1_150_000, 2_000_000, n])       # (one build shard is empty)
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker_uneven, args=(r, world, port, n, cuts, mode, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    exp, me = oracle.hash_join(oracle.gen_sm64_mod(11, 0, n, n // 2), oracle.gen_sm64_mod(12, 0, n, n // 2))
    got = canonical(np.concatenate([r[1] for r in res]))
    assert got.shape[0] == me and np.array_equal(got, exp)
    assert all(r[2]["fast_path"] == (mode == "fast") for r in res)


def test_sharded_join_small_messages_are_split():
    """max_msg_bytes far below a message: every (source, destination) message travels in many pieces, both ends agree on them."""
    from np_twin import canonical
    n, world = 1_000_000, 2
    for mode in ("fast", "exact"):
        res = _run(world, n, "unique", mode, max_msg=100_000)
        exp, me = _expected(n, "unique", 0, world, -1)
        got = canonical(np.concatenate([r[1] for r in res]))
        assert got.shape[0] == me and np.array_equal(got, exp)


def _rccl_world1(fn):
    """run fn(comm, ctx) with a world-size-1 built-in RCCL communicator on this GPU"""
    from gpu_olap_engine_amd.engine import Context, comm_unique_id
    torch.cuda.set_device(0)
    c = Context(max_gpu_memory=0)
    comm = c.comm_create_rccl(comm_unique_id(), 1, 0)
    try:
        return fn(comm, c)
    finally:
        comm.close()
        c.close()


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_sharded_join_rccl_world1(mode):
    """Built-in RCCL transport, one rank: communicator from a unique id, ncclAllGather of the bucket counts and of the status
    word, the whole partition -> local pass -> join flow at 25M rows; checked on the device."""
    from gpu_olap_engine_amd import _native as N
    n = 25_000_000

    def body(comm, c):
        if mode == "fast":
            c.debug_set("GOLAP_SHARD_FAST", "1")
        b = c.generate(N.GEN_BUILD_KEYS, n, modulus=n)
        p = c.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
        out = torch.empty((n + 16, 2), dtype=torch.int32, device="cuda")
        pairs, cnt = comm.sharded_join(b, p, 0, 0, n, out_pairs=out, force_exact=(mode == "exact"))
        torch.cuda.synchronize()
        m = int(cnt.item())
        v = c.verify_pairs(pairs, m, b, p).cpu()
        assert m == n and int(v[0]) == n and (int(v[2]) & (2**64 - 1)) == (n * (n - 1) // 2) & (2**64 - 1)
        return comm.info()

    info = _rccl_world1(body)
    assert info["fast_path"] == (mode == "fast")


def test_rccl_world1_chunks_halves_composite_and_aggregate():
    """The round-3 pieces over the built-in RCCL transport at world size 1 (the only RCCL a one-GPU box has): a fast-path join with row
    chunks and group halves forced on (self-messages: every bucket is this rank's own), the composite / NULL-key pre-pass (its
    all-gather of the value ranges), and golap_sharded_aggregate in both key modes (table all-gather, partition + self-exchange)."""
    import oracle
    from gpu_olap_engine_amd import _native as N
    from np_twin import canonical
    n = 25_000_000

    def body(comm, c):
        c.debug_set("GOLAP_SHARD_FAST", "1")
        c.debug_set("GOLAP_SHARD_CHUNKS", "4")
        c.debug_set("GOLAP_SHARD_HALVES", "2")
        b = c.generate(N.GEN_BUILD_KEYS, n, modulus=n)
        p = c.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
        out = torch.empty((n + 16, 2), dtype=torch.int32, device="cuda")
        pairs, cnt = comm.sharded_join(b, p, 0, 0, n, out_pairs=out)
        torch.cuda.synchronize()
        m = int(cnt.item())
        v = c.verify_pairs(pairs, m, b, p).cpu()
        info = comm.info()
        assert m == n and int(v[0]) == n and (int(v[2]) & (2**64 - 1)) == (n * (n - 1) // 2) & (2**64 - 1)
        assert info["fast_path"] and info["chunks"] == 4 and info["halves"] == 2
        # composite + NULL keys
        bc, pc, bv, pv = _composite_rows(30_000)
        cu = lambda a: None if a is None else torch.from_numpy(a).cuda()
        out2 = torch.empty((3_000_000, 2), dtype=torch.int32, device="cuda")
        pairs2, cnt2 = comm.sharded_join([cu(x) for x in bc], [cu(x) for x in pc], 0, 0, 30_000, out_pairs=out2, join_type=3,
                                         build_valid=[cu(x) for x in bv], probe_valid=[cu(x) for x in pv])
        torch.cuda.synchronize()
        exp, me = oracle.hash_join_multi(bc, pc, 3, bv, pv)
        got = canonical(pairs2[:int(cnt2.item())].cpu().numpy().view(np.uint32))
        assert got.shape[0] == me and np.array_equal(got, exp)
        # aggregates
        for hashed in (False, True):
            k, vals, vn = _agg_rows(400_000, 3000, hashed)
            kt, vt = torch.from_numpy(k).cuda(), torch.from_numpy(vals).cuda()
            res = comm.sharded_aggregate([(N.AGG_SUM, 0), (N.AGG_COUNT, 0)], [vt], group_keys=kt, n_groups=3000 + (8 if hashed else 0), hashed=hashed)
            torch.cuda.synchronize()
            uk, inv = np.unique(k, return_inverse=True)
            s = np.zeros(uk.size); np.add.at(s, inv, vals)
            cexp = np.bincount(inv, minlength=uk.size).astype(np.float64)
            if hashed:
                outp, okeys, ong = res
                g = int(ong.item())
                order = np.argsort(okeys[:g].cpu().numpy())
                assert g == uk.size and np.array_equal(okeys[:g].cpu().numpy()[order], uk)
                assert np.array_equal(outp[0, :g].cpu().numpy()[order], s) and np.array_equal(outp[1, :g].cpu().numpy()[order], cexp)
            else:
                assert np.array_equal(res[0].cpu().numpy()[uk], s) and np.array_equal(res[1].cpu().numpy()[uk], cexp)
        return info

    _rccl_world1(body)


def test_rccl_message_size_boundary():
    """Where do large single RCCL messages stop arriving intact on this stack?  Round 1 saw an 8 GB self-message deliver wrong
    data (scripts/dbg_sharded.py CHECK_BIG).  golap_sharded_join never sends more than max_msg_bytes (default 512 MiB) in one
    message; this pins that the default is on the safe side and records the first failing size (if any) in the log."""
    sizes = [(1 << 29), (1 << 30), (1 << 31) - 64, (1 << 31) + 64, (1 << 32) - 64, (1 << 32) + 64, (1 << 33)]

    def body(comm, c):
        out = []
        for s in sizes:
            bad, first = comm.selftest(s)
            out.append((s, bad, first))
        return out

    res = _rccl_world1(body)
    print("rccl self-message check (bytes, wrong words, first wrong word):", res)
    with open(os.path.join(ROOT, "gpurun_out", "rccl_selftest.log"), "w") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else open(os.devnull, "w") as f:
        f.write(repr(res) + "\n")
    assert res[0][1] == 0, "a 512 MiB message (the default max_msg_bytes) must arrive intact"
    assert res[1][1] == 0, "1 GiB has arrived intact on every box so far: the default keeps a factor of two below it"


def _worker_err(rank, world, port, n, mode, bad_rank, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if mode == "fast":
        os.environ["GOLAP_DEBUG_ENV"] = "1"
        os.environ["GOLAP_SHARD_FAST"] = "1"
        os.environ["GOLAP_LOCAL_BITS"] = "6"
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        import oracle
        from gpu_olap_engine_amd import _native as N
        from gpu_olap_engine_amd.distributed import NativeShardedJoin
        from gpu_olap_engine_amd.engine import Context, shard_range
        torch.cuda.set_device(0)
        c = Context(max_gpu_memory=0)
        lo, hi = shard_range(n, rank, world)
        b, p = _keys("unique", n, lo, hi - lo)
        sj = NativeShardedJoin(c, transport="staged")
        out = torch.empty((2 * n, 2), dtype=torch.int32, device="cuda")
        bt, pt = torch.from_numpy(b).cuda(), torch.from_numpy(p).cuda()
        code, msg = 0, ""
        try:   # ONE rank passes a bad argument; the call is collective
            sj.join(bt, pt, lo, lo, n, out_pairs=out, join_type=(9 if rank == bad_rank else 0), force_exact=(mode == "exact"))
        except N.GolapError as e:
            code, msg = e.code, e.message
        # ... and the communicator is still usable: the same join with good arguments on every rank
        pairs, cnt, info = sj.join(bt, pt, lo, lo, n, out_pairs=out, force_exact=(mode == "exact"))
        torch.cuda.synchronize()
        m = int(cnt.item())
        q.put((rank, code, msg, pairs[:m].cpu().numpy().view(np.uint32).copy()))
        dist.barrier()
        sj.close()
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_a_failing_rank_takes_every_rank_out_of_the_call(mode):
    """golap_sharded_join is collective: a rank that fails on the host before the exchange (here: a bad argument on rank 1 of 2) must not
    leave its peers waiting in ncclSend/ncclRecv.  Every rank returns -- the failing one with its own error, the other with
    GOLAP_ERR_PEER naming the rank -- and the next call on the same communicator works."""
    import torch.multiprocessing as mp
    from gpu_olap_engine_amd import _native as N
    from np_twin import canonical
    world, n = 2, 2_000_000
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker_err, args=(r, world, port, n, mode, 1, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert res[1][1] == N.ERR_INVALID_ARG and "join_type" in res[1][2]
    assert res[0][1] == N.ERR_PEER and "rank 1" in res[0][2]
    exp, me = _expected(n, "unique", 0, world, -1)
    got = canonical(np.concatenate([r[3] for r in res]))
    assert got.shape[0] == me and np.array_equal(got, exp)


def _worker_agg(rank, world, port, n, groups, hashed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        import oracle
        from gpu_olap_engine_amd import _native as N
        from gpu_olap_engine_amd.distributed import NativeShardedJoin
        from gpu_olap_engine_amd.engine import Context, shard_range
        torch.cuda.set_device(0)
        c = Context(max_gpu_memory=0)
        lo, hi = shard_range(n, rank, world)
        k, v, vn = _agg_rows(n, groups, hashed)
        kt = torch.from_numpy(k[lo:hi]).cuda()
        vt = torch.from_numpy(v[lo:hi]).cuda()
        vv = torch.from_numpy((~vn[lo:hi]).astype(np.uint8)).cuda()
        sj = NativeShardedJoin(c, transport="staged")
        specs = [(N.AGG_SUM, 0), (N.AGG_COUNT, 0), (N.AGG_MIN, 0), (N.AGG_MAX, 0), (N.AGG_AVG, 0)]
        if hashed:
            out, cnt, okeys, ong = sj.comm.sharded_aggregate(specs, [vt], group_keys=kt, n_groups=groups + 8, want_counts=True, hashed=True, value_valid=[vv])
            torch.cuda.synchronize()
            g = int(ong.item())
            q.put((rank, okeys[:g].cpu().numpy(), out[:, :g].cpu().numpy(), cnt[:g].cpu().numpy()))
        else:
            out, cnt = sj.comm.sharded_aggregate(specs, [vt], group_keys=kt, n_groups=groups, want_counts=True, value_valid=[vv])
            torch.cuda.synchronize()
            q.put((rank, None, out.cpu().numpy(), cnt.cpu().numpy()))
        dist.barrier()
        sj.close()
        c.close()
    finally:
        dist.destroy_process_group()


def _agg_rows(n, groups, hashed):
    rng = np.random.default_rng(groups * 7 + n)
    ids = rng.integers(0, groups, n)
    k = (ids * 1_000_003 - 5_000_000_000).astype(np.int64) if hashed else ids.astype(np.int64)   # hashed: sparse, negative and positive keys
    v = rng.integers(-1000, 1000, n).astype(np.float64)
    vn = rng.random(n) < 0.05                                                                   # NULL values
    return k, v, vn


@pytest.mark.parametrize("world,groups,hashed", [(2, 1000, False), (4, 200_000, False), (2, 5000, True), (4, 300_000, True)])
def test_sharded_aggregate_ranks_share_one_gpu(world, groups, hashed):
    """golap_sharded_aggregate (C++, the join's communicator) with 2 and 4 ranks on one GPU through the staged transport: SUM / COUNT /
    MIN / MAX / AVG with NULL values, dense keys (replicated result, identical on every rank) and hashed keys (sharded result: every
    key on exactly one rank), against a numpy GROUP BY of the whole table."""
    import torch.multiprocessing as mp
    n = 1_500_000
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker_agg, args=(r, world, port, n, groups, hashed, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    k, v, vn = _agg_rows(n, groups, hashed)
    uk, inv = np.unique(k, return_inverse=True)
    live = ~vn
    G = uk.size
    s = np.zeros(G); np.add.at(s, inv[live], v[live])
    c = np.zeros(G); np.add.at(c, inv[live], 1.0)
    mn = np.full(G, np.finfo(np.float64).max); np.minimum.at(mn, inv[live], v[live])
    mx = np.full(G, -np.finfo(np.float64).max); np.maximum.at(mx, inv[live], v[live])
    avg = np.where(c > 0, s / np.maximum(c, 1), 0.0)
    exp = np.stack([s, c, mn, mx, avg])
    if hashed:
        keys = np.concatenate([r[1] for r in res])
        outs = np.concatenate([r[2] for r in res], axis=1)
        cnts = np.concatenate([r[3] for r in res])
        assert keys.size == G and np.unique(keys).size == G                 # every key on exactly one rank
        order = np.argsort(keys)
        assert np.array_equal(keys[order], uk)
        assert np.array_equal(outs[:, order], exp) and np.array_equal(cnts[order].astype(np.float64), c)
        assert all(r[1].size > 0 for r in res)                               # and the result is spread over the ranks
    else:
        full = np.zeros((5, groups)); full[2] = np.finfo(np.float64).max; full[3] = -np.finfo(np.float64).max
        full[:, uk] = exp
        for r in res:                                                        # replicated, bit-identical on every rank
            assert np.array_equal(r[2], full) and np.array_equal(r[3].astype(np.float64), full[1])


def _composite_rows(n, clean=None, cols=2, wide=False):
    """clean = (lo, hi): rows [lo, hi) of every column are valid -- the shard of a rank that is handed NO validity arrays.
    wide: the two columns' values are drawn from pools that span all of int64 (a mixed-radix packing would need ~128 bits)"""
    rng = np.random.default_rng(n + 3)
    if wide:
        pool0, pool1 = rng.integers(-2**63, 2**63 - 1, 3000), rng.integers(-2**63, 2**63 - 1, 30)
        mk = lambda: (pool0[rng.integers(0, 3000, n)], pool1[rng.integers(0, 30, n)])
    else:
        mk = lambda: (rng.integers(-40, 40, n).astype(np.int64), rng.integers(1000, 1030, n).astype(np.int64))
    (b0, b1), (p0, p1) = mk(), mk()
    bv = [(rng.random(n) > 0.05).astype(np.uint8), None]
    pv = [(rng.random(n) > 0.05).astype(np.uint8), (rng.random(n) > 0.05).astype(np.uint8)]
    if clean:
        for v in (bv[0], pv[0], pv[1]):
            v[clean[0]:clean[1]] = 1
    if cols == 1:    # one key column with a few hundred distinct values (both sides), NULLs on both
        if wide:
            return [b0], [p0], bv[:1], pv[:1]
        return [b0 * 7 + b1], [p0 * 7 + p1], bv[:1], pv[:1]
    return [b0, b1], [p0, p1], bv, pv


def _worker_composite(rank, world, port, n, jt, q, plain_rank=-1, cols=2, mode=None, wide=False, knobs=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        from gpu_olap_engine_amd.distributed import NativeShardedJoin
        from gpu_olap_engine_amd.engine import Context, shard_range
        torch.cuda.set_device(0)
        c = Context(max_gpu_memory=0)
        lo, hi = shard_range(n, rank, world)
        bc, pc, bv, pv = _composite_rows(n, shard_range(n, plain_rank, world) if plain_rank >= 0 else None, cols, wide)
        cu = lambda a: None if a is None else torch.from_numpy(a[lo:hi]).cuda()
        for k, v in (knobs or {}).items():
            c.debug_set(k, v)
        if mode == "fast":
            c.debug_set("GOLAP_SHARD_FAST", "1")
            c.debug_set("GOLAP_LOCAL_BITS", "6")
        sj = NativeShardedJoin(c, transport="staged")
        out = torch.empty((60 * n + 16, 2), dtype=torch.int32, device="cuda")
        no_nulls_here = rank == plain_rank     # this rank's shard has no NULLs and it is handed no validity arrays at all
        pairs, cnt, info = sj.join([cu(x) for x in bc], [cu(x) for x in pc], lo, lo, n, out_pairs=out, join_type=jt,
                                   build_valid=None if no_nulls_here else [cu(x) for x in bv], probe_valid=None if no_nulls_here else [cu(x) for x in pv],
                                   force_exact=(mode == "exact"))
        torch.cuda.synchronize()
        m = int(cnt.item())
        q.put((rank, pairs[:m].cpu().numpy().view(np.uint32).copy()))
        dist.barrier()
        sj.close()
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,jt", [(2, 0), (4, 3), (2, 1)])
def test_sharded_join_composite_and_null_keys(world, jt):
    """golap_sharded_join with a two-column key and NULLs in three of the four key columns: the ranks agree on the columns' value ranges and
    join on the exact mixed-radix key; a row with a NULL in any key column matches nothing and an outer join keeps it -- against the
    oracle's multi-column NULL-aware join of the whole tables."""
    import oracle
    import torch.multiprocessing as mp
    from np_twin import canonical
    n = 40_000
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker_composite, args=(r, world, port, n, jt, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    bc, pc, bv, pv = _composite_rows(n)
    exp, me = oracle.hash_join_multi(bc, pc, jt, bv, pv)
    got = canonical(np.concatenate([r[1] for r in res]))
    assert got.shape[0] == me and np.array_equal(got, exp)


@pytest.mark.parametrize("world,jt,cols,plain_rank,mode", [(2, 0, 2, 1, None), (2, 3, 1, 0, "exact"), (4, 1, 1, 2, "fast"), (2, 0, 1, 1, "fast")])
def test_sharded_join_ranks_disagree_about_validity_arrays(world, jt, cols, plain_rank, mode):
    """Whether a shard comes with validity arrays depends on its data: ONE rank whose shard has no NULLs passes none (what the Arrow
    import hands over when null_count is 0) while its peers pass theirs.  With one key column that rank would take the plain path
    and its peers the packed-key path -- collectives that do not match.  The call's first agreement round settles the form: the
    plain rank starts over in the general one, every rank returns, and the result is the oracle's NULL-aware join of the whole tables."""
    import oracle
    import torch.multiprocessing as mp
    from gpu_olap_engine_amd.engine import shard_range
    from np_twin import canonical
    n = 40_000
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker_composite, args=(r, world, port, n, jt, q, plain_rank, cols, mode)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    bc, pc, bv, pv = _composite_rows(n, shard_range(n, plain_rank, world), cols)
    exp, me = oracle.hash_join_multi(bc, pc, jt, bv, pv)
    got = canonical(np.concatenate([r[1] for r in res]))
    assert got.shape[0] == me and np.array_equal(got, exp)



@pytest.mark.parametrize("world,jt,cols,knobs", [(2, 0, 2, None), (4, 3, 2, None), (2, 1, 1, None), (2, 2, 2, {"GOLAP_GJ_HASH_BITS": "27"}),
                                                 (2, 3, 2, {"GOLAP_GJ_HASH_BITS": "27"}), (2, 3, 2, {"GOLAP_SHARD_WIDE_KEYS": "1"})])
def test_sharded_join_key_tuples_beyond_62_bits(world, jt, cols, knobs):
    """physical_plan.rs:35-36 puts no bound on the key columns.  Two columns of hash-like 64-bit values (or ONE such column with NULLs)
    do not fit the exact mixed-radix key: the ranks join on a 64-bit surrogate of the tuple and ship the real columns with the rows;
    the owner of a partition joins on positions, checks every pair against the shipped columns, and only then emits global row ids.
    With the surrogates narrowed to 27 bits (GOLAP_GJ_HASH_BITS) a handful of different tuples DO share one (~12 false pairs expected over
    the 40K x 40K rows): they must be found and removed, and an outer join must give a row that thereby lost its only partner its
    unmatched pair.  GOLAP_SHARD_WIDE_KEYS=1 sends small-range tuples (many duplicates per tuple) down the same route.  Against the
    oracle's multi-column join."""
    import oracle
    import torch.multiprocessing as mp
    from np_twin import canonical
    n = 40_000
    wide = not (knobs and "GOLAP_SHARD_WIDE_KEYS" in knobs)      # (last case: small value ranges forced onto the same route)
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker_composite, args=(r, world, port, n, jt, q, -1, cols, None, wide, knobs)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    bc, pc, bv, pv = _composite_rows(n, None, cols, wide)
    exp, me = oracle.hash_join_multi(bc, pc, jt, bv, pv)
    got = canonical(np.concatenate([r[1] for r in res]))
    assert got.shape[0] == me and np.array_equal(got, exp)
