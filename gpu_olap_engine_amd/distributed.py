"""One-process-per-GPU sharded radix hash join.

PRODUCT PATH: `NativeShardedJoin` -- a thin wrapper over the C ABI's golap_sharded_join (csrc/golap_shard.inc.h): partition,
exchange (RCCL over xGMI, called from C++), local passes and join are all driven by the library; Python only hands over
tensors and, once, the communicator's unique id.  `StagedTransport` plugs torch.distributed (gloo) into the same C++ code
through the transport callbacks, so that several ranks can rehearse the real kernels on ONE GPU (tests).

`ShardedJoin` below is the round-1 orchestration of the exact path in Python, kept as the CPU-testable model of the
protocol (tests/test_distributed_cpu.py drives it under gloo with an oracle-backed stand-in for the device steps).

Rows of both tables start range-sharded; every rank radix-partitions its local rows into the
reference's 256 partitions (join_kernels.cuh:69), laid out owner-major so that the rows bound
for each peer are one contiguous range; partition p is owned by rank p % P.  The only data-path
collective is the partition exchange: an all-to-all issued as grouped send/recv on slices
(torch.distributed.batch_isend_irecv; backend "nccl" = RCCL over xGMI, all 7 links of a GPU carry traffic at once).  After the exchange each rank joins its
own 256/P partitions locally (further LDS-sized sub-partitioning + LDS hash tables); output pairs
stay sharded and carry GLOBAL u32 row ids.

The exchange logic is backend-agnostic: `backend` provides the two compute steps.  On GPUs it is
HipBackend (the C-ABI library); the world_size-2 gloo tests on CPU plug an oracle-backed stand-in
into the same class to exercise sharding, ownership mapping, split sizes and the all-to-all.
"""
import torch
import torch.distributed as dist

N_PARTITIONS = 256   # join_kernels.cuh:22-23


def owner_of_partition(p, world):
    """partition p lives on GPU p mod P."""
    return p % world


def slot_of_partition(p, world):
    """owner-major layout used by golap_radix_partition: slot = (p % P) * (256/P) + p // P."""
    return (p % world) * (N_PARTITIONS // world) + p // world


def shard_range(n_total, rank, world):
    """Range sharding: rank g holds rows [g*N/P, (g+1)*N/P) (last rank takes the remainder)."""
    per = n_total // world
    lo = rank * per
    hi = n_total if rank == world - 1 else lo + per
    return lo, hi


class HipBackend:
    """Compute steps on the local GPU through the C ABI."""

    def __init__(self, ctx):
        self.ctx = ctx

    def total_bits(self, n_build_total):
        pl = self.ctx.join_plan(max(1, n_build_total))
        return pl["bits1"] + pl["bits2"]

    def radix_partition(self, keys, rid_base, n_owners, extra_bits, narrow=False):
        return self.ctx.radix_partition(keys, rid_base=rid_base, n_owners=n_owners, extra_bits=extra_bits, narrow=narrow)

    def narrow_ok(self, n_groups, n_sources, pre_bits, hint):
        """can the whole partition -> exchange -> local passes -> join flow run on 8-byte (key32, row id) tuples?"""
        return self.ctx.join_partitioned_narrow_ok(n_groups, n_sources, pre_bits, hint)

    def join_begin(self, n_groups, n_sources, pre_bits, hint, build_rows, probe_rows, narrow=False):
        self.ctx.join_partitioned_begin(n_groups, n_sources, pre_bits, hint, build_rows, probe_rows, narrow=narrow)

    def join_side(self, which, keys, rids, seg_off, seg_len, rows):
        self.ctx.join_partitioned_side(which, keys, rids, seg_off, seg_len, rows)

    def join_finish(self, out_pairs):
        return self.ctx.join_partitioned_finish(out_pairs)

    def empty(self, n, dtype, like):
        return torch.empty(n, dtype=dtype, device=like.device)


class _Exchange:
    """An in-flight partition exchange of one side (async all-to-all rounds + the resulting segment table)."""

    def __init__(self):
        self.works, self.keep = [], []
        self.staged = []           # (host buffer, device slice): rehearsal transport, see ShardedJoin._exchange_start
        self.keys = self.rids = None
        self.n_recv = 0
        self.seg_off, self.seg_len = [], []
        self.any_wide = False      # narrow exchange only: some rank saw a key outside [0, 2^32) -> nothing was sent

    def wait(self):
        for w in self.works:
            w.wait()          # NCCL: makes the current stream wait; gloo: blocks the host
        for host_buf, dst in self.staged:
            dst.copy_(host_buf)
        self.works, self.keep, self.staged = [], [], []


class ShardedJoin:
    def __init__(self, backend, group=None, max_msg_bytes=1 << 30, extra_bits=None, narrow=True):
        self.backend = backend
        self.group = group
        self.max_msg_bytes = max_msg_bytes
        self.forced_extra_bits = extra_bits
        self.narrow = narrow          # try 8-byte (key32, row id) tuples through the exchange (a third less over xGMI)
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if N_PARTITIONS % self.world or self.world & (self.world - 1):
            raise ValueError("world size must be a power of two dividing 256")

    def extra_bits(self, n_build_total):
        """Hash bits beyond the reference's 8 that the exchange-time partition already resolves, chosen so that
        ONE local pass (<= 9 bits) reaches LDS-sized sub-partitions: 1B build rows -> 18 bits -> 8 + 1 | 9."""
        if self.forced_extra_bits is not None:
            return self.forced_extra_bits
        return max(0, min(2, self.backend.total_bits(n_build_total) - 8 - 9))

    def _exchange_start(self, keys, rids, offs_host, e1, rows_hint, wide_flag=None):
        """keys/rids are partitioned owner-major into (256 << e1) buckets; offs_host are the bucket offsets.
        Starts the all-to-all (async) and returns an _Exchange whose segment table is indexed [group][source],
        group = local bucket index in [0, (256/P) << e1).

        The all-to-all is issued as grouped point-to-point operations on SLICES of the partitioned buffers
        (dist.batch_isend_irecv = one ncclGroup of ncclSend/ncclRecv over xGMI): the owner-major layout makes
        every peer's rows one contiguous slice, so nothing is packed; the rank's own partitions are copied on
        the device.  It runs in `rounds` over consecutive bucket ranges so that no single message exceeds
        max_msg_bytes (one 8 GB RCCL message was observed to deliver wrong data on this stack; bounded rounds
        are also what lets rounds pipeline).

        Narrow exchange (rids is None): `keys` holds (key32, row id) tuples and is the only array sent.  wide_flag (0/1,
        this rank's partition saw a key outside [0, 2^32)) rides along with the size exchange, so every rank learns the
        OR over all ranks without another collective; if it is set nothing is sent and x.any_wide tells the caller to
        start over with 64-bit rows."""
        P = self.world
        per = (N_PARTITIONS // P) << e1
        nb = N_PARTITIONS << e1
        dev = keys.device
        offs = [int(v) for v in offs_host]
        x = _Exchange()
        narrow = rids is None
        sizes = torch.tensor([[offs[r * per + s + 1] - offs[r * per + s] for s in range(per)] + [int(bool(wide_flag))] for r in range(P)],
                             dtype=torch.int64)
        # Rehearsal transport: under the gloo backend device tensors are staged through host buffers, so that several
        # ranks can drive the real kernels on ONE GPU (tests); on RCCL ("nccl") nothing is staged.
        staged = dev.type == "cuda" and dist.get_backend(self.group) == "gloo"
        sz_dev = sizes.reshape(-1) if staged else sizes.to(dev).reshape(-1)
        rsz_dev = torch.empty_like(sz_dev)
        dist.all_to_all_single(rsz_dev, sz_dev, group=self.group)           # sizes[src][bucket] of what I receive (+ src's flag)
        rsz_all = rsz_dev.cpu().reshape(P, per + 1)
        x.any_wide = bool(rsz_all[:, per].any())
        if x.any_wide:
            return x
        rsz = rsz_all[:, :per]
        x.n_recv = int(rsz.sum())
        x.keys = self.backend.empty(max(x.n_recv, 1), torch.int64, keys)
        x.rids = None if narrow else self.backend.empty(max(x.n_recv, 1), torch.int32, keys)
        # rounds: every rank must agree WITHOUT another collective -> a bound every rank can compute: no message
        # can carry more rows than one rank holds (rows_hint = global rows / P, identical everywhere)
        rounds = max(1, min(per, -(-(int(rows_hint) + P) * 8 // self.max_msg_bytes)))   # (tuples are 8 bytes as well)
        bounds = [per * j // rounds for j in range(rounds + 1)]
        x.seg_off = [0] * (per * P)
        x.seg_len = [0] * (per * P)
        x.keep += [keys] if narrow else [keys, rids]
        pos = 0
        me = self.rank
        for j in range(rounds):
            s0, s1 = bounds[j], bounds[j + 1]
            if s1 == s0:
                continue
            recv_counts = rsz[:, s0:s1].sum(dim=1).tolist()
            ops = []
            src_base = pos
            for r in range(P):
                a, b = offs[r * per + s0], offs[r * per + s1]          # what I send to r: one contiguous slice, no packing
                n_from_r = int(recv_counts[r])
                if r == me:
                    # own partitions never touch the collective: a plain device copy on the compute stream
                    if n_from_r:
                        x.keys[src_base:src_base + n_from_r].copy_(keys[a:b])
                        if not narrow:
                            x.rids[src_base:src_base + n_from_r].copy_(rids[a:b])
                else:
                    peer = r if self.group is None else dist.get_global_rank(self.group, r)

                    def out_buf(t):
                        if not staged:
                            return t
                        h = t.cpu()
                        x.keep.append(h)
                        return h

                    def in_buf(t):
                        if not staged:
                            return t
                        h = torch.empty(t.shape, dtype=t.dtype)
                        x.staged.append((h, t))
                        return h
                    if b > a:
                        ops.append(dist.P2POp(dist.isend, out_buf(keys[a:b]), peer, self.group))
                        if not narrow:
                            ops.append(dist.P2POp(dist.isend, out_buf(rids[a:b]), peer, self.group))
                    if n_from_r:
                        ops.append(dist.P2POp(dist.irecv, in_buf(x.keys[src_base:src_base + n_from_r]), peer, self.group))
                        if not narrow:
                            ops.append(dist.P2POp(dist.irecv, in_buf(x.rids[src_base:src_base + n_from_r]), peer, self.group))
                within = 0
                for s_ in range(s0, s1):
                    x.seg_off[s_ * P + r] = src_base + within
                    x.seg_len[s_ * P + r] = int(rsz[r, s_])
                    within += int(rsz[r, s_])
                src_base += n_from_r
            if ops:
                x.works += dist.batch_isend_irecv(ops)                 # one grouped ncclSend/ncclRecv batch per round
            pos += int(sum(recv_counts))
        return x

    def join(self, build_keys, probe_keys, build_rid_base, probe_rid_base, n_build_total, out_pairs=None, max_pairs=None,
             n_probe_total=None, _narrow=None):
        """Local shard of the global join -> (pairs [m,2] int32 view of u32 GLOBAL row ids, count tensor, rows received per side).

        Schedule (one compute stream + the collective's own stream):
          partition(build) -> exchange(build) starts || partition(probe) -> exchange(probe) starts ||
          local sub-partitioning(build) -> local sub-partitioning(probe) -> LDS join."""
        P = self.world
        if n_probe_total is None:
            n_probe_total = n_build_total
        e1 = self.extra_bits(n_build_total)
        n_groups = (N_PARTITIONS // P) << e1
        hint = max(1, n_build_total // P)
        narrow = self.narrow if _narrow is None else _narrow
        narrow = bool(narrow) and hasattr(self.backend, "narrow_ok") and self.backend.narrow_ok(n_groups, P, 8 + e1, hint)
        if narrow:
            # 8-byte (key32, row id) tuples: valid only if no rank holds a key outside [0, 2^32).  Every rank learns the OR of
            # all ranks' flags with the size exchange; if it is set, all ranks start over with 64-bit rows (same decision
            # everywhere, so the collectives stay matched).
            bk, br, boff, bflag = self.backend.radix_partition(build_keys, build_rid_base, P, e1, narrow=True)
            boff_h = boff.cpu().numpy()                      # (the flag is final once the offsets are: same stream)
            xb = self._exchange_start(bk, None, boff_h, e1, -(-n_build_total // P), int(bflag.cpu().item()))
            if xb.any_wide:
                return self.join(build_keys, probe_keys, build_rid_base, probe_rid_base, n_build_total, out_pairs, max_pairs, n_probe_total, _narrow=False)
            pk, pr, poff, pflag = self.backend.radix_partition(probe_keys, probe_rid_base, P, e1, narrow=True)
            poff_h = poff.cpu().numpy()
            xp = self._exchange_start(pk, None, poff_h, e1, -(-n_probe_total // P), int(pflag.cpu().item()))
            if xp.any_wide:
                xb.wait()                                    # the build-side transfers in flight must land before their buffers go
                return self.join(build_keys, probe_keys, build_rid_base, probe_rid_base, n_build_total, out_pairs, max_pairs, n_probe_total, _narrow=False)
        else:
            bk, br, boff = self.backend.radix_partition(build_keys, build_rid_base, P, e1)
            xb = self._exchange_start(bk, br, boff.cpu().numpy(), e1, -(-n_build_total // P))
            pk, pr, poff = self.backend.radix_partition(probe_keys, probe_rid_base, P, e1)
            xp = self._exchange_start(pk, pr, poff.cpu().numpy(), e1, -(-n_probe_total // P))
        if narrow:
            self.backend.join_begin(n_groups, P, 8 + e1, hint, xb.n_recv, xp.n_recv, narrow=True)
        else:
            self.backend.join_begin(n_groups, P, 8 + e1, hint, xb.n_recv, xp.n_recv)
        xb.wait()
        self.backend.join_side(0, xb.keys, xb.rids, xb.seg_off, xb.seg_len, xb.n_recv)
        xp.wait()
        self.backend.join_side(1, xp.keys, xp.rids, xp.seg_off, xp.seg_len, xp.n_recv)
        if out_pairs is None:
            cap = max(1, xp.n_recv if max_pairs is None else max_pairs)
            out_pairs = torch.empty((cap, 2), dtype=torch.int32, device=build_keys.device)
        pairs, cnt = self.backend.join_finish(out_pairs)
        # the received buffers may be released now: every kernel that reads them is already enqueued on this
        # stream, and torch's caching allocator reuses memory in stream order
        return pairs, cnt, xb.n_recv, xp.n_recv


class ShardedAggregate:
    """GROUP BY over row-sharded inputs (SURVEY 8(e) "Aggregate"): every rank pre-aggregates its own rows into the dense
    [stat][group] table on its GPU (golap_aggregate: LDS pre-aggregation / partition-then-aggregate), then the tables are
    combined with all-reduce -- SUM for sums and counts, MIN / MAX for extrema; AVG is finalised from the reduced sum and
    count.  Sums of integer-valued data stay bit-exact under any reduction order."""

    SUM, COUNT, MIN, MAX, AVG = 0, 1, 2, 3, 4

    def __init__(self, local_stats, group=None):
        """local_stats(values, group_keys, n_groups) -> float64 tensor [4, n_groups] = (sum, count, min, max) of the LOCAL rows."""
        self.local_stats = local_stats
        self.group = group

    @staticmethod
    def hip_local_stats(ctx):
        def f(values, group_keys, n_groups):
            return ctx.aggregate([(0, 0), (1, 0), (2, 0), (3, 0)], [values], group_keys=group_keys, n_groups=n_groups)
        return f

    def aggregate(self, values, group_keys, n_groups, funcs):
        st = self.local_stats(values, group_keys, n_groups)                 # [4, G] on the local device
        sc = st[0:2].contiguous()
        mn = st[2].contiguous()
        mx = st[3].contiguous()
        dist.all_reduce(sc, op=dist.ReduceOp.SUM, group=self.group)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.group)
        out = []
        for f in funcs:
            if f == self.SUM:
                out.append(sc[0])
            elif f == self.COUNT:
                out.append(sc[1])
            elif f == self.MIN:
                out.append(mn)
            elif f == self.MAX:
                out.append(mx)
            else:                                                            # AVG = sum / count, 0.0 for an empty group (executor.rs:339-345)
                out.append(torch.where(sc[1] > 0, sc[0] / sc[1].clamp(min=1.0), torch.zeros_like(sc[0])))
        return torch.stack(out)



# ---------------------------------------------------------------------------------------------------------------------
# product path: golap_sharded_join through the C ABI
# ---------------------------------------------------------------------------------------------------------------------
class StagedTransport:
    """golap_transport callbacks over torch.distributed with host staging (rehearsal / non-RCCL fabrics): device buffers
    are copied to host memory with the library's TransferQueue, moved with the process group (gloo), and copied back."""

    def __init__(self, ctx, group=None):
        import ctypes as C

        import numpy as np

        from . import _native as N
        self.ctx, self.group = ctx, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        lib = ctx._lib

        def d2h(ptr, nbytes):
            h = np.empty(max(nbytes, 1), dtype=np.uint8)
            ctx._ck(lib.golap_xfer_enqueue(ctx._h, N.DEVICE_TO_HOST, ptr, h.ctypes.data, nbytes, 0))
            return h

        def h2d(h, ptr, nbytes):
            ctx._ck(lib.golap_xfer_enqueue(ctx._h, N.HOST_TO_DEVICE, h.ctypes.data, ptr, nbytes, 0))

        def all_gather(user, send_ptr, recv_ptr, nbytes, stream):
            try:
                ctx._ck(lib.golap_ctx_synchronize(ctx._h, C.c_void_p(stream)))
                h = d2h(send_ptr, nbytes)
                ctx.xfer_synchronize()
                mine = torch.from_numpy(h[:nbytes].copy())
                parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
                dist.all_gather(parts, mine, group=self.group)
                allb = torch.cat(parts).numpy()
                h2d(allb, recv_ptr, nbytes * self.world)
                ctx.xfer_synchronize()
                return 0
            except Exception as e:   # an exception must not unwind through the C frames
                print(f"StagedTransport.all_gather failed: {e!r}", flush=True)
                return N.ERR_HIP

        def exchange(user, ops, n_ops, stream):
            try:
                ctx._ck(lib.golap_ctx_synchronize(ctx._h, C.c_void_p(stream)))
                sends, recvs, works = [], [], []
                for i in range(n_ops):
                    o = ops[i]
                    if o.is_send:
                        sends.append((o.peer, d2h(o.ptr, o.bytes), o.bytes))
                    else:
                        recvs.append((o.peer, torch.empty(max(o.bytes, 1), dtype=torch.uint8), o.ptr, o.bytes))
                ctx.xfer_synchronize()
                for peer, h, nb in sends:
                    g = peer if self.group is None else dist.get_global_rank(self.group, peer)
                    works.append(dist.isend(torch.from_numpy(h[:nb]), g, group=self.group))
                for peer, t, ptr, nb in recvs:
                    g = peer if self.group is None else dist.get_global_rank(self.group, peer)
                    works.append(dist.irecv(t[:nb], g, group=self.group))
                for w in works:
                    w.wait()
                for peer, t, ptr, nb in recvs:
                    h2d(t.numpy(), ptr, nb)
                ctx.xfer_synchronize()
                return 0
            except Exception as e:
                print(f"StagedTransport.exchange failed: {e!r}", flush=True)
                return N.ERR_HIP

        self.all_gather, self.exchange = all_gather, exchange


class NativeShardedJoin:
    """The multi-GPU join as the library runs it (golap_sharded_join).  transport: "rccl" (built-in, ncclSend/ncclRecv from
    C++; the 128-byte unique id is broadcast once through torch.distributed) or "staged" (StagedTransport)."""

    def __init__(self, ctx, group=None, transport="rccl", max_msg_bytes=0):
        from .engine import comm_unique_id
        self.ctx, self.group = ctx, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.max_msg_bytes = max_msg_bytes
        if transport == "rccl":
            box = [comm_unique_id() if self.rank == 0 else None]
            if self.world > 1:
                src = 0 if group is None else dist.get_global_rank(group, 0)
                dist.broadcast_object_list(box, src=src, group=group)
            self.comm = ctx.comm_create_rccl(box[0], self.world, self.rank)
        elif transport == "staged":
            self._tr = StagedTransport(ctx, group)
            self.comm = ctx.comm_create_custom(self._tr.all_gather, self._tr.exchange, self.world, self.rank)
        else:
            raise ValueError("transport must be 'rccl' or 'staged'")

    def join(self, build_keys, probe_keys, build_rid_base, probe_rid_base, n_build_total, n_probe_total=None, out_pairs=None,
             max_pairs=None, join_type=0, force_exact=False, build_rows_per_rank=None, probe_rows_per_rank=None, build_valid=None,
             probe_valid=None):
        """-> (pairs int32[.,2] with GLOBAL u32 row ids, count int64[1] device, info dict)."""
        pairs, cnt = self.comm.sharded_join(build_keys, probe_keys, build_rid_base, probe_rid_base, n_build_total, n_probe_total,
                                            out_pairs=out_pairs, max_pairs=max_pairs, join_type=join_type, force_exact=force_exact,
                                            max_msg_bytes=self.max_msg_bytes, build_rows_per_rank=build_rows_per_rank,
                                            probe_rows_per_rank=probe_rows_per_rank, build_valid=build_valid, probe_valid=probe_valid)
        return pairs, cnt, self.comm.info()

    def close(self):
        self.comm.close()
