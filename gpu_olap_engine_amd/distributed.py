"""One-process-per-GPU sharded radix hash join.

Rows of both tables start range-sharded; every rank radix-partitions its local rows into the
reference's 256 partitions (join_kernels.cuh:69), laid out owner-major so that the rows bound
for each peer are one contiguous range; partition p is owned by rank p % P.  The only data-path
collective is the partition exchange: torch.distributed.all_to_all_single (backend "nccl" = RCCL
over xGMI; all 7 links of a GPU carry traffic at once).  After the exchange each rank joins its
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

    def radix_partition(self, keys, rid_base, n_owners):
        return self.ctx.radix_partition(keys, rid_base=rid_base, n_owners=n_owners)

    def join_partitioned(self, bk, br, bso, bsl, pk, pr, pso, psl, n_slots, n_sources, out_pairs, hint):
        return self.ctx.join_partitioned(bk, br, bso, bsl, pk, pr, pso, psl, n_slots, n_sources, out_pairs,
                                         total_build_rows_hint=hint)

    def empty(self, n, dtype, like):
        return torch.empty(n, dtype=dtype, device=like.device)


class ShardedJoin:
    def __init__(self, backend, group=None, max_msg_bytes=1 << 30):
        self.backend = backend
        self.group = group
        self.max_msg_bytes = max_msg_bytes
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if N_PARTITIONS % self.world or self.world & (self.world - 1):
            raise ValueError("world size must be a power of two dividing 256")
        self.per = N_PARTITIONS // self.world

    def _exchange_side(self, keys, rids, offs_host):
        """keys/rids are partitioned owner-major; offs_host[257] are the slot offsets.
        Returns (recv_keys, recv_rids, n_recv, seg_off, seg_len) with segments indexed [slot][source].

        The exchange runs in `rounds` all-to-all rounds over consecutive slot ranges so that no single
        peer message exceeds max_msg_bytes (RCCL send/recv of > 4 GiB messages was observed to deliver
        garbage on this stack; bounded rounds are also what a pipelined exchange needs)."""
        P, per = self.world, self.per
        dev = keys.device
        offs = [int(v) for v in offs_host]
        # per-slot sizes of what this rank sends to each peer: [peer][slot]
        sizes = torch.tensor([[offs[r * per + s + 1] - offs[r * per + s] for s in range(per)] for r in range(P)], dtype=torch.int64)
        sz_dev = sizes.to(dev).reshape(-1)
        rsz_dev = torch.empty_like(sz_dev)
        dist.all_to_all_single(rsz_dev, sz_dev, group=self.group)           # sizes[src][slot] of what I receive
        rsz = rsz_dev.cpu().reshape(P, per)
        n_recv = int(rsz.sum())
        rk = self.backend.empty(max(n_recv, 1), torch.int64, keys)
        rr = self.backend.empty(max(n_recv, 1), torch.int32, keys)
        # rounds: every rank must agree -> derive from the global maximum message (rows)
        biggest = torch.tensor([max(int(sizes.sum(dim=1).max()), int(rsz.sum(dim=1).max()))], dtype=torch.int64, device=dev)
        dist.all_reduce(biggest, op=dist.ReduceOp.MAX, group=self.group)
        rounds = max(1, min(per, -(-int(biggest.item()) * 8 // self.max_msg_bytes)))
        bounds = [per * j // rounds for j in range(rounds + 1)]
        seg_off = [0] * (per * P)
        seg_len = [0] * (per * P)
        pos = 0
        for j in range(rounds):
            s0, s1 = bounds[j], bounds[j + 1]
            if s1 == s0:
                continue
            send_counts = sizes[:, s0:s1].sum(dim=1).tolist()
            recv_counts = rsz[:, s0:s1].sum(dim=1).tolist()
            n_r = int(sum(recv_counts))
            if rounds == 1:
                sk, sr = keys[:offs[N_PARTITIONS]], rids[:offs[N_PARTITIONS]]
            else:   # gather this round's slot range of every owner into one contiguous send buffer
                idx = [(offs[r * per + s0], offs[r * per + s1]) for r in range(P)]
                sk = torch.cat([keys[a:b] for a, b in idx])
                sr = torch.cat([rids[a:b] for a, b in idx])
            dist.all_to_all_single(rk[pos:pos + n_r], sk, recv_counts, send_counts, group=self.group)
            dist.all_to_all_single(rr[pos:pos + n_r], sr, recv_counts, send_counts, group=self.group)
            src_base = pos
            for r in range(P):
                within = 0
                for s in range(s0, s1):
                    seg_off[s * P + r] = src_base + within
                    seg_len[s * P + r] = int(rsz[r, s])
                    within += int(rsz[r, s])
                src_base += recv_counts[r]
            pos += n_r
        return rk, rr, n_recv, seg_off, seg_len

    def join(self, build_keys, probe_keys, build_rid_base, probe_rid_base, n_build_total, out_pairs=None, max_pairs=None):
        """-> (pairs [m,2] int32 view of u32 global row ids, count tensor).  Local shard of the global join."""
        P = self.world
        bk, br, boff = self.backend.radix_partition(build_keys, build_rid_base, P)
        pk, pr, poff = self.backend.radix_partition(probe_keys, probe_rid_base, P)
        offs = torch.stack([boff, poff]).cpu().numpy()                      # one small D2H for both sides
        rbk, rbr, nb, bso, bsl = self._exchange_side(bk, br, offs[0])
        rpk, rpr, npr, pso, psl = self._exchange_side(pk, pr, offs[1])
        if out_pairs is None:
            cap = max(1, npr if max_pairs is None else max_pairs)
            out_pairs = torch.empty((cap, 2), dtype=torch.int32, device=build_keys.device)
        return self.backend.join_partitioned(rbk[:nb], rbr[:nb], bso, bsl, rpk[:npr], rpr[:npr], pso, psl,
                                             self.per, P, out_pairs, max(1, n_build_total // P)) + (nb, npr)
