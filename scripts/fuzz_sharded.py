"""Randomised parity fuzz of golap_sharded_join against the CPU oracle (developer tool; test infrastructure, not product).
usage (GPU box): python scripts/fuzz_sharded.py SECONDS [SEED] [WORLD]

WORLD ranks (2 or 4) share the one GPU; the library's transport callbacks are served by gloo through host buffers (the same C++
orchestration, kernels, message sizes and segment tables as with RCCL).  Every round all ranks derive the same recipe from the
shared seed: table sizes, an UNEVEN split of both tables over the ranks (empty shards included), a key distribution, a join type,
the fast or the exact path (the fast one forced at sizes far below its production threshold, with a small local fan-out), message
pieces; rank 0 gathers all pairs and compares the canonical list bit for bit with the oracle."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch


def draw(rng, n, kind, domain):
    if n == 0:
        return np.empty(0, np.int64)
    if kind == "uniform":
        return rng.integers(0, domain, n)
    if kind == "unique":
        return rng.permutation(max(domain, n))[:n].astype(np.int64)
    if kind == "hot":
        k = rng.integers(0, domain, n)
        m = rng.random(n) < 0.3
        k[m] = rng.integers(0, domain, 4)[rng.integers(0, 4, int(m.sum()))]
        return k
    if kind == "offset":                                   # one common upper half: the tuple pipeline on the low halves
        return rng.integers(0, domain, n) + (5 << 32)
    if kind == "tenants":                                  # consecutive upper halves (a tenant prefix) from a random, also negative, start:
        k = rng.integers(0, domain, n)                     # the window form on every rank -- or the exact path when the window is too small
        t, lo = int(rng.integers(2, 200)), int(rng.integers(-50, 300))
        return k + (((k % t) + lo) << 32)
    if kind == "wide":
        return rng.integers(-2**62, 2**62, n)
    raise ValueError(kind)


def cuts(rng, n, world):
    c = np.sort(rng.integers(0, n + 1, world - 1)) if rng.random() < 0.7 else np.array([n * (r + 1) // world for r in range(world - 1)])
    if rng.random() < 0.15 and world > 2:
        c[0] = 0                                           # an empty first shard
    return [0] + [int(x) for x in c] + [n]


def worker(rank, world, port, budget, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    import oracle
    from gpu_olap_engine_amd.distributed import NativeShardedJoin
    from gpu_olap_engine_amd.engine import Context
    from np_twin import canonical
    torch.cuda.set_device(0)
    ctx = Context(max_gpu_memory=0)
    sj = NativeShardedJoin(ctx, transport="staged")
    rng = np.random.default_rng(seed)                      # the SAME stream on every rank
    t_end = time.time() + budget
    rnd = 0
    ok_all = True
    while True:
        go = torch.tensor([1 if time.time() < t_end else 0])
        dist.broadcast(go, 0)                              # rank 0 decides when to stop
        if not int(go.item()):
            break
        rnd += 1
        big = rng.random() < 0.3
        nb = int(rng.integers(0, 4_000_000 if big else 100_000))
        npr = int(rng.integers(0, 4_000_000 if big else 100_000))
        kind = str(rng.choice(["uniform", "unique", "hot", "offset", "tenants", "wide"]))
        domain = int(max(1, rng.choice([nb // 4 + 1, nb + 1, 4 * nb + 1, 97])))
        jt = int(rng.choice([0, 0, 1, 2, 3]))
        mode = str(rng.choice(["fast", "exact"]))
        max_msg = int(rng.choice([0, 0, 100_000, 1 << 20]))
        b = draw(rng, nb, kind, domain).astype(np.int64)
        p = draw(rng, npr, kind if rng.random() < 0.7 else "uniform", domain).astype(np.int64)
        if nb and npr and rng.random() < 0.5:
            take = rng.integers(0, nb, min(npr, 1000))
            p[:take.size] = b[take]
        cb, cp = cuts(rng, nb, world), cuts(rng, npr, world)
        if nb and npr:                                     # no quadratic blow-ups
            ub, nbc = np.unique(b, return_counts=True)
            up, npc = np.unique(p, return_counts=True)
            _, ib, ip = np.intersect1d(ub, up, assume_unique=True, return_indices=True)
            if int((nbc[ib].astype(np.int64) * npc[ip]).sum()) > 20_000_000:
                continue
        knobs = {"GOLAP_SHARD_FAST": "1", "GOLAP_LOCAL_BITS": str(rng.choice(["5", "6", "7"]))} if mode == "fast" else {}
        recipe = f"seed={seed} round={rnd} world={world} nb={nb} np={npr} kind={kind} domain={domain} jt={jt} mode={mode} max_msg={max_msg} cuts={cb}/{cp} knobs={knobs}"
        for k in ("GOLAP_SHARD_FAST", "GOLAP_LOCAL_BITS"):
            ctx.debug_set(k, knobs.get(k))
        sj.max_msg_bytes = max_msg
        lb, hb, lp, hp = cb[rank], cb[rank + 1], cp[rank], cp[rank + 1]
        tb = torch.from_numpy(b[lb:hb].copy()).cuda() if hb > lb else torch.empty(0, dtype=torch.int64, device="cuda")
        tp = torch.from_numpy(p[lp:hp].copy()).cuda() if hp > lp else torch.empty(0, dtype=torch.int64, device="cuda")
        exp, me = oracle.hash_join_outer(b, p, jt) if rank == 0 else (None, 0)
        mt = torch.tensor([me])
        dist.broadcast(mt, 0)
        cap = int(mt.item()) + 64
        out = torch.empty((cap, 2), dtype=torch.int32, device="cuda")
        pairs, cnt, info = sj.join(tb, tp, lb, lp, nb, npr, out_pairs=out, max_pairs=cap, join_type=jt, force_exact=(mode == "exact"),
                                   build_rows_per_rank=[cb[r + 1] - cb[r] for r in range(world)],
                                   probe_rows_per_rank=[cp[r + 1] - cp[r] for r in range(world)])
        torch.cuda.synchronize()
        m = int(cnt.item())
        mine = pairs[:min(m, cap)].cpu().numpy().view(np.uint32).copy()
        gathered = [None] * world
        dist.gather_object(mine, gathered if rank == 0 else None, 0)
        if rank == 0:
            got = canonical(np.concatenate(gathered)) if sum(g.shape[0] for g in gathered) else np.empty((0, 2), np.uint32)
            ok = got.shape[0] == me and np.array_equal(got, exp)
            print(("ok   " if ok else "FAIL ") + recipe + f" matches={me} fast={info['fast_path']} abandoned={info['fast_path_abandoned']}", flush=True)
            ok_all = ok_all and ok
        flag = torch.tensor([1 if ok_all else 0])
        dist.broadcast(flag, 0)
        if not int(flag.item()):
            break
    if rank == 0:
        print(f"fuzz_sharded: {rnd} rounds, " + ("all equal to the oracle" if ok_all else "MISMATCH"), flush=True)
        q.put(ok_all)
    dist.barrier()
    sj.close()
    ctx.close()
    dist.destroy_process_group()


def main():
    import socket
    import torch.multiprocessing as mp
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    import tempfile, uuid
    port = os.path.join(tempfile.gettempdir(), f"golap_rdzv_{os.getpid()}_{uuid.uuid4().hex}")   # a FileStore rendezvous: no port to race for
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=worker, args=(r, world, port, budget, seed, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    ok = q.get(timeout=budget + 600)
    for pr in procs:
        pr.join(timeout=120)
    sys.exit(0 if ok and all(pr.exitcode == 0 for pr in procs) else 1)


if __name__ == "__main__":
    main()
