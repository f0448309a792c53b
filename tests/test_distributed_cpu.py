"""world_size-2 (and 4) gloo tests of the N>1 path on CPU: the SAME ShardedJoin class the GPU bench uses
(sharding, owner-major slots, split sizes, all_to_all_single exchange, segment tables, global row ids),
with an oracle-backed stand-in for the two device compute steps (tests may use the oracle; the product
backend is HipBackend)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    """CPU stand-in with the semantics of golap_radix_partition / golap_join_partitioned_{begin,side,finish}."""

    def total_bits(self, n_build_total):
        return 17

    def narrow_ok(self, n_groups, n_sources, pre_bits, hint):
        return True

    def radix_partition(self, keys, rid_base, n_owners, extra_bits, narrow=False):
        import oracle
        k = keys.numpy()
        h = oracle.hash_int64_array(k).astype(np.int64)
        p = h & 255
        sub = (h >> 8) & ((1 << extra_bits) - 1)
        per = 256 // n_owners
        bucket = (((p % n_owners) * per + p // n_owners) << extra_bits) | sub        # owner-major slot, refined by extra bits
        order = np.argsort(bucket, kind="stable")
        nb = 256 << extra_bits
        offs = np.concatenate([[0], np.cumsum(np.bincount(bucket, minlength=nb))])
        rr = (order.astype(np.int64) + rid_base).astype(np.uint32).view(np.int32)
        if narrow:   # (key32 | row id << 32) tuples + "a key outside [0, 2^32) was seen" (golap_partition_params.narrow)
            wide = bool(((k < 0) | (k >= (1 << 32))).any())
            tup = (k[order].astype(np.uint64) & np.uint64(0xFFFFFFFF)) | (rr.view(np.uint32).astype(np.uint64) << np.uint64(32))
            return torch.from_numpy(tup.view(np.int64).copy()), None, torch.tensor(offs, dtype=torch.int32), torch.tensor([int(wide)], dtype=torch.int32)
        return torch.from_numpy(k[order].copy()), torch.from_numpy(rr.copy()), torch.tensor(offs, dtype=torch.int32)

    def join_begin(self, n_groups, n_sources, pre_bits, hint, build_rows, probe_rows, narrow=False):
        self.cfg = (n_groups, n_sources, pre_bits)
        self.sides = {}
        self.narrow = narrow
        self.narrow_joins = getattr(self, "narrow_joins", 0) + int(narrow)

    def join_side(self, which, keys, rids, seg_off, seg_len, rows):
        if self.narrow:
            assert rids is None
            t = keys[:rows].numpy().view(np.uint64)
            k, r = (t & np.uint64(0xFFFFFFFF)).astype(np.int64), (t >> np.uint64(32)).astype(np.uint32)
        else:
            k, r = keys[:rows].numpy().copy(), rids[:rows].numpy().view(np.uint32).copy()
        self.sides[which] = (k, r, list(seg_off), list(seg_len))

    def join_finish(self, out_pairs):
        import oracle
        n_groups, n_sources, pre_bits = self.cfg
        bk, br, bso, bsl = self.sides[0]
        pk, pr, pso, psl = self.sides[1]
        allp = []
        for s in range(n_groups):
            bi = np.concatenate([np.arange(bso[s * n_sources + r], bso[s * n_sources + r] + bsl[s * n_sources + r]) for r in range(n_sources)]).astype(np.int64)
            pi = np.concatenate([np.arange(pso[s * n_sources + r], pso[s * n_sources + r] + psl[s * n_sources + r]) for r in range(n_sources)]).astype(np.int64)
            if bi.size == 0 or pi.size == 0:
                continue
            # every row of a group shares its low pre_bits hash bits
            hb = oracle.hash_int64_array(bk[bi]) & ((1 << pre_bits) - 1)
            assert (hb == hb[0]).all()
            pairs, m = oracle.hash_join(bk[bi], pk[pi])
            if m:
                allp.append(np.stack([br[bi][pairs[:, 0]], pr[pi][pairs[:, 1]]], axis=1))
        got = np.concatenate(allp) if allp else np.empty((0, 2), np.uint32)
        m = got.shape[0]
        out_pairs[:min(m, out_pairs.shape[0])] = torch.from_numpy(got[:out_pairs.shape[0]].view(np.int32).copy())
        return out_pairs, torch.tensor([m], dtype=torch.int64)

    def empty(self, n, dtype, like):
        return torch.empty(n, dtype=dtype)


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


def _widen_some(b, p, lo, hi, n, wide):
    """wide = "" (all keys narrow), "build_last_rank" / "probe_rank0": ONE rank holds ONE key outside [0, 2^32) on one side
    (the other ranks only learn it through the exchange), "all": every key shifted by 2^40."""
    b, p = b.copy(), p.copy()
    if wide == "all":
        b += 1 << 40
        p += 1 << 40
    elif wide == "build_last_rank" and hi == n:
        b[-1] = -7                                   # an unmatched negative key in the last rank's build shard
    elif wide == "probe_rank0" and lo == 0:
        p[0] = (1 << 33) + 5
    return b, p


def _worker(rank, world, port, n, variant, q, max_msg=1 << 30, e1=0, wide=""):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        import oracle
        from gpu_olap_engine_amd.distributed import ShardedJoin, shard_range
        lo, hi = shard_range(n, rank, world)
        if variant == "unique":
            b = oracle.gen_build_keys(lo, hi - lo, n)
            p = oracle.gen_probe_keys(lo, hi - lo, n)
        else:   # duplicates on both sides + rank 1 holds an empty-ish tail
            b = oracle.gen_sm64_mod(11, lo, hi - lo, n // 2)
            p = oracle.gen_sm64_mod(12, lo, hi - lo, n // 2)
        b, p = _widen_some(b, p, lo, hi, n, wide)
        be = OracleBackend()
        sj = ShardedJoin(be, max_msg_bytes=max_msg, extra_bits=e1)
        out = torch.empty((4 * n, 2), dtype=torch.int32)
        pairs, cnt, nb, npr = sj.join(torch.from_numpy(b), torch.from_numpy(p), lo, lo, n, out_pairs=out)
        m = int(cnt.item())
        # tuples travelled iff no rank held a key outside [0, 2^32); otherwise every rank fell back together
        assert be.narrow_joins == (0 if wide else 1)
        # every build/probe row this rank received belongs to a partition it owns (p % world == rank)
        q.put((rank, pairs[:m].numpy().view(np.uint32).copy(), nb, npr))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,variant,max_msg,e1,wide", [(2, "unique", 1 << 30, 0, ""), (2, "dups", 1 << 30, 1, ""), (4, "unique", 1 << 30, 0, ""),
                                                             (2, "unique", 20_000, 1, ""), (4, "dups", 7_000, 2, ""),
                                                             (2, "unique", 1 << 30, 1, "all"), (4, "dups", 1 << 30, 0, "build_last_rank"),
                                                             (2, "dups", 9_000, 1, "probe_rank0")])
def test_sharded_join_gloo(world, variant, max_msg, e1, wide):
    """max_msg small -> the exchange runs in several bounded rounds over slot ranges.  Rows travel as 8-byte (key32, row id)
    tuples unless some rank holds a key outside [0, 2^32): then every rank learns it with the size exchange and all fall
    back to 64-bit rows together (`wide` cases: the offending key sits on one rank only, build or probe side)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from np_twin import canonical
    n = 40_000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, variant, q, max_msg, e1, wide)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if variant == "unique":
        b, pr = oracle.gen_build_keys(0, n, n), oracle.gen_probe_keys(0, n, n)
    else:
        b, pr = oracle.gen_sm64_mod(11, 0, n, n // 2), oracle.gen_sm64_mod(12, 0, n, n // 2)
    b, pr = _widen_some(b, pr, 0, n, n, wide)
    exp, me = oracle.hash_join(b, pr)
    got = canonical(np.concatenate([r[1] for r in res]))
    assert got.shape[0] == me and np.array_equal(got, exp)
    # ownership: each pair's key hashes to a partition owned by the rank that produced it
    for rank, pairs, nb, npr in res:
        if pairs.shape[0]:
            pid = oracle.hash_int64_array(b[pairs[:, 0].astype(np.int64)]) & 255
            assert ((pid % world) == rank).all()
    assert sum(r[2] for r in res) == n and sum(r[3] for r in res) == n     # the exchange neither lost nor duplicated rows


def test_slot_mapping_is_a_bijection():
    from gpu_olap_engine_amd.distributed import owner_of_partition, shard_range, slot_of_partition
    for world in (1, 2, 4, 8):
        slots = [slot_of_partition(p, world) for p in range(256)]
        assert sorted(slots) == list(range(256))
        per = 256 // world
        for p in range(256):
            assert slots[p] // per == owner_of_partition(p, world) == p % world
    assert shard_range(10, 0, 4) == (0, 2) and shard_range(10, 3, 4) == (6, 10)


def _agg_worker(rank, world, port, n, G, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        import oracle
        from gpu_olap_engine_amd.distributed import ShardedAggregate, shard_range
        lo, hi = shard_range(n, rank, world)
        keys = oracle.gen_sm64_mod(7, lo, hi - lo, G, np.int64)
        vals = oracle.gen_sm64_mod(8, lo, hi - lo, 1024, np.float64)

        def local_stats(values, group_keys, n_groups):          # oracle-backed stand-in for golap_aggregate
            ks, st = oracle.group_aggregate(group_keys.numpy(), values.numpy())
            t = np.zeros((4, n_groups)); t[2] = np.finfo(np.float64).max; t[3] = -np.finfo(np.float64).max
            t[0, ks], t[1, ks], t[2, ks], t[3, ks] = st[:, 0], st[:, 1], st[:, 2], st[:, 3]
            return torch.from_numpy(t)

        sa = ShardedAggregate(local_stats)
        out = sa.aggregate(torch.from_numpy(vals), torch.from_numpy(keys), G, [sa.SUM, sa.COUNT, sa.MIN, sa.MAX, sa.AVG])
        q.put((rank, out.numpy().copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_aggregate_gloo():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    n, G, world = 60_000, 500, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agg_worker, args=(r, world, port, n, G, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ks, st = oracle.group_aggregate(oracle.gen_sm64_mod(7, 0, n, G, np.int64), oracle.gen_sm64_mod(8, 0, n, 1024, np.float64))
    assert np.array_equal(ks, np.arange(G))
    for rank, out in res:                                   # every rank holds the full reduced table
        assert np.array_equal(out[0], st[:, 0]) and np.array_equal(out[1], st[:, 1]) and np.array_equal(out[2], st[:, 2])
        assert np.array_equal(out[3], st[:, 3]) and np.array_equal(out[4], st[:, 4])


# ---- golap_sharded_join's host-side plan (C ABI, no device needed): every rank must derive the same messages ---------------
def _plan_worker(rank, world, port, n, max_msg, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        from gpu_olap_engine_amd.engine import shard_range, sharded_plan
        pl = sharded_plan(world, n, max_msg_bytes=max_msg)
        pl["rows"] = shard_range(n, rank, world)[1] - shard_range(n, rank, world)[0]
        box = [None] * world
        dist.all_gather_object(box, pl)
        q.put((rank, box))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,max_msg", [(2, 1_000_000_000, 0), (4, 1_000_000_000, 1 << 28), (2, 999_999_937, 0)])
def test_sharded_plan_is_the_same_on_every_rank(world, n, max_msg):
    """The fast path has no size exchange: the bytes rank s sends to rank d are a function of s's row count alone, so both ends
    must compute them identically (here: every rank computes the whole table through the C ABI and the tables are compared
    under gloo), pieces must cover the message, and every bucket's range must hold its expected rows with 6 sigma to spare."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, world, port, n, max_msg, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mm = max_msg or (1 << 29)      # the library's default bound of one message
    for rank, box in res:
        ref = {k: v for k, v in box[0].items() if k != "rows"}
        for r in range(world):
            assert {k: v for k, v in box[r].items() if k != "rows"} == ref       # identical on every rank
        assert ref["eligible"] and ref["n_buckets"] == 256 << ref["extra_bits"] and ref["buckets_per_rank"] * world == ref["n_buckets"]
        for side in range(2):
            for s in range(world):
                rows = box[s]["rows"]
                cap, cap_last, nb, ns, K = ref["cap_rows"][side][s], ref["cap_rows_last"][side][s], ref["n_buckets"], ref["sub_ranges"], ref["chunks"]
                assert K == 4 and world * K * ns <= 128                              # with peers a side is partitioned and shipped in 4 row chunks
                tile = 18 * 1024                                                      # chunks 0 .. K-2 are whole first-pass tiles, the last takes the rest
                chunk = -(-(-(-rows // K)) // tile) * tile
                last = rows - (K - 1) * chunk
                assert 0 < last <= chunk
                for c_rows, c_cap in ((chunk, cap), (last, cap_last)):
                    avg = -(-c_rows // (nb * ns))
                    assert c_cap % 16 == 0 and c_cap >= avg + 6 * int(avg ** 0.5)   # provisioned sub-range: mean + 6 sigma (Poisson), line aligned
                assert ref["msg_bytes"][side][s] == ref["buckets_per_rank"] * ns * ((K - 1) * cap + cap_last) * 8
                per_chunk = [ref["buckets_per_rank"] * ns * c * 8 for c in [cap] * (K - 1) + [cap_last]]
                assert ref["msg_pieces"][side][s] == sum(-(-b // mm) for b in per_chunk)   # every chunk's message is split on its own


def test_sharded_plan_small_inputs_take_the_exact_path():
    from gpu_olap_engine_amd.engine import sharded_plan
    assert not sharded_plan(2, 3_000_000)["eligible"]          # below the big-tile first pass: histogram partition + size exchange
    assert sharded_plan(8, 1_000_000_000)["eligible"]
    assert sharded_plan(8, 1_000_000_000)["extra_bits"] == 1 and sharded_plan(8, 1_000_000_000)["local_bits"] == 9
