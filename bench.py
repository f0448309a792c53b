#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric: join rows/s + achieved HBM GB/s on the
1B x 1B int64 radix hash join (config C5), at 1/2/4/8 MI355X.

    python bench.py --gpus 1 --steps K --warmup W          (single GPU: the whole 1B x 1B join on one device)
    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts `python -m torch.distributed.run
                                                            --nproc-per-node N bench.py <same arguments>` as a CHILD process before it
                                                            touches the GPU, relays rank 0's JSON line and the child's exit code)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (the same under an external launcher: one rank per GPU, RCCL)
    python bench.py --config C3|C4|C4L|C1                  (the other BASELINE configs, same JSON shape, one GPU)

A "step" is one complete pass of the operator over the synthetic, device-resident input (join: radix passes, [partition
exchange over xGMI when N > 1,] LDS build/probe, pair output).  value = rows of the WHOLE job / max-over-ranks time per
step; join scaling is "strong" (the 1B x 1B table pair is fixed, N GPUs share it).  Rank 0 prints ONE JSON line.

The default C5 run also times the same join with keys outside [0, 2^32) (`wide_keys` in the line): both key columns + 2^40
(all keys share one upper half: still 8-byte tuples, frame of reference), with two different upper halves (8-byte tuples with a
dictionary index) and with 2^17 upper halves (the 64-bit-row pipeline).  BASELINE's own generators emit keys below 2^32.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C5", choices=["C5", "C3", "C4", "C4L", "C1", "OP"],
                    help="C5 1Bx1B join (default) | C3 100Mx100M join | C4 1B-row GROUP BY int32 -> SUM(f64), 1024 groups | C4L the same "
                         "with 1048576 groups | C1 SELECT sum(v) WHERE k<c on a 1M-row batch | OP a filter -> join -> GROUP BY plan over HOST-resident "
                         "Arrow batches through the operator layer (golap_plan_* / golap_execute): H2D / D2H reported separately")
    ap.add_argument("--rows", type=int, default=0, help="rows per side / per column of the WHOLE job (0 = the config's size)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="rows of the bounded CPU-baseline sample (-1 = the config's default, 0 = skip)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-wide", action="store_true", help="C5: skip the second timed leg with 64-bit keys")
    ap.add_argument("--rehearse", action="store_true",
                    help="developer: run the N > 1 code path on ONE GPU (all ranks on device 0, gloo + host-staged transport instead of RCCL); "
                         "the numbers mean nothing, the line is marked")
    ap.add_argument("--keys", default="narrow", choices=["narrow", "offset", "two", "window", "random"],
                    help="C5/C3 main leg: key columns as generated (below 2^32) | + 2^40 | two upper halves | 2^16 key-dependent upper halves (the 64-bit-row "
                         "pipeline): what the profiles of the wide kernels are taken on")
    ap.add_argument("--no-others", action="store_true", help="C5 default run: skip the other_configs block (C3, C4, C4L, C1)")
    ap.add_argument("--pipelining", action="store_true",
                    help="OP: a filter's batch k is filtered while batch k + 1 is uploaded (golap_host_set_pipelining(1)) -- the A/B of the default, "
                         "one transfer phase for all batches; measured slower (DESIGN.md 8)")
    ap.add_argument("--sharded", action="store_true", help="force the partition-exchange path (golap_sharded_join) even with 1 rank")
    ap.add_argument("--workspace-draws", type=int, default=6,
                    help="single-GPU joins: contexts (= workspace slabs, i.e. physical placements of the partition buffers) tried before the warm-up; "
                         "the fastest is kept and every draw is reported (1 = take the first, DESIGN 9 'Run-to-run spread')")
    return ap.parse_args()


def self_launch_command(argv, n_ranks, port):
    """The command `python bench.py --gpus N` (N > 1, no launcher) starts: one rank per GPU under torch.distributed.run on this node,
    rendezvous on 127.0.0.1 (the container's hostname may not resolve), the same bench arguments."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """--gpus N > 1 without WORLD_SIZE in the environment: nobody has launched the ranks, so this process does -- as a CHILD
    (subprocess; this process has made no GPU call and imports no torch, and it never replaces itself: an exec from a process that
    holds the GPU takes the box down).  The child's stderr passes through; of its stdout the JSON line of rank 0 is relayed as this
    process's one line, anything else goes to stderr.  Returns the child's exit code."""
    import subprocess
    cmd = self_launch_command(argv, args.gpus, int(os.environ.get("MASTER_PORT", 0)) or free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    print("bench.py: starting " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, cwd=ROOT)
    line = None
    for ln in proc.stdout:
        t = ln.strip()
        if t.startswith("{") and t.endswith("}") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("bench.py: the ranks exited 0 without printing a line", file=sys.stderr, flush=True)
        rc = 1
    return rc


def kernel_source_hash():
    """sha256 over the kernel sources: ties profiles/pmc_traffic.json to the code it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gpu_olap_engine_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, rows, world):
    """HBM bytes per launch from the committed rocprofv3 PMC passes -- only when they were taken on exactly these kernel sources."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        tj = json.load(open(path))
        if tj.get("kernel_source_sha") != kernel_source_hash():
            return None
        e = tj.get(f"{kernel}@{rows}x{world}")
        return e["hbm_bytes_per_launch"] if e else None
    except Exception:
        return None


def step_traffic(stats, steps, rows, world):
    """HBM bytes ONE step moved according to the committed PMC passes: sum over the step's kernels of (bytes per launch x launches per
    step); None unless every kernel that takes more than 2 % of the step's device time has an entry for these sources and this size."""
    kern = {k: v for k, v in stats.items() if v["launches"]}
    tot_ms = sum(v["total_ms"] for v in kern.values()) or 1.0
    total = 0.0
    for k, v in kern.items():
        t = pmc_traffic(k, rows, world)
        if t is None:
            if v["total_ms"] / tot_ms > 0.02:
                return None
            continue
        total += t * v["launches"] / steps
    return int(total) if total else None


def roofline_of(stats, rows, world, direct, matches_per_launch=0):
    """the dominant kernel (by accumulated device time): algorithmic bytes per launch / mean launch duration (hipEvents on the
    launch stream, recorded inside the library)"""
    kern = {k: v for k, v in stats.items() if v["launches"] and k != "scan_offsets"}
    if not kern:
        return None
    dom = max(kern, key=lambda k: kern[k]["total_ms"])
    s = kern[dom]
    per_launch = s["bytes"] / s["launches"]
    if dom.startswith("join_build_probe_lds"):
        per_launch += 8 * matches_per_launch      # the library credits the rows read at launch; the pairs written are known here
    avg_ms = s["total_ms"] / s["launches"]
    ach = per_launch / (avg_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
            "traffic": pmc_traffic(dom, rows, world) if direct else None, "alg_bytes_per_launch": int(per_launch),
            "avg_launch_ms": round(avg_ms, 4), "launches": s["launches"]}


def cpu_join_baseline(sample, rank):
    """Oracle timed on the host cores: single-thread faithful port (executor.rs:238-253 + probe) on a bounded sample of the
    same generators, plus the multi-threaded radix variant for context."""
    if sample <= 0 or rank != 0:
        return None
    import oracle
    b = oracle.gen_build_keys(0, sample, sample)
    p = oracle.gen_probe_keys(0, sample, sample)
    t0 = time.perf_counter()
    m, _ = oracle.execute_hash_join_count(b, p)
    t1 = time.perf_counter()
    cores = os.cpu_count() or 1
    m2, _ = oracle.radix_join_mt(b, p, cores)
    t2 = time.perf_counter()
    assert m == m2 == sample
    return {"value": 2 * sample / (t1 - t0), "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": f"{sample} x {sample} rows, same generators as the GPU run (N={sample}), single-thread "
                      f"HashMap<i64,Vec<usize>> build + probe as executor.rs:238-253; {t1 - t0:.2f} s",
            "fair_value": 2 * sample / (t2 - t1), "fair_cores": cores,
            "fair_note": f"radix-partitioned, std::thread over all {cores} host cores; {t2 - t1:.2f} s"}


def exchange_rates(join_info, world):
    """What this rank's partition exchange moved, from the last call's device timestamps (golap_sharded_join_info): bytes this rank
    SENT to its world - 1 peers (the library counts them; its own partitions never leave the device) / the exchange stage's
    duration, per side.  per_link = that / (world - 1): every peer is one xGMI hop away and all are served at once, so this is the
    rate ONE link carried in ONE direction while the others were busy too -- a lower bound of the link's capability (the stage also
    waits for the partition chunks it ships)."""
    if world < 2 or not join_info.get("fast_path"):
        return None
    out = {}
    for side, (nm, stage) in enumerate((("build", "exchange_build"), ("probe", "exchange_probe"))):
        t0, t1 = join_info["stage_ms"][stage]
        if t1 > t0:
            gbps = join_info["bytes_sent"][side] / ((t1 - t0) * 1e-3) / 1e9
            out[f"{nm}_out_GBps"] = round(gbps, 1)
            out[f"{nm}_per_link_GBps"] = round(gbps / (world - 1), 1)
    return out or None


def device_block(info):
    return {"name": info["name"], "arch": info["gcn_arch"], "cus": info["compute_units"], "hbm_bus_bits": info["memory_bus_bits"],
            "mem_clock_mhz": info["memory_clock_mhz"]}


def timed(step, steps, warmup, barrier, ctx):
    for _ in range(warmup):
        res = step()
    barrier()
    ctx.stats_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    barrier()
    return res, time.perf_counter() - t0


# ---------------------------------------------------------------------------------------------------------------------
# joins: C5 (default, 1/2/4/8 GPUs) and C3
# ---------------------------------------------------------------------------------------------------------------------
def bench_join(args, n_total, label):
    import torch
    import torch.distributed as dist

    from gpu_olap_engine_amd import _native as N
    from gpu_olap_engine_amd.engine import Context, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start `python bench.py --gpus N` (it launches its ranks itself) or "
                         f"torch.distributed.run --nproc-per-node N with the same N")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.sharded
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)   # barrier / max-over-ranks only; the join's own
                                                                                           # exchange is the library's RCCL communicator
    # profiling = 2: every kernel that does work is bracketed by hipEvents on its launch stream inside the library (the roofline's durations
    # come from them), the ~17 stand-by launches of a join only while a recent join needed them -- a pair of events costs as much stream
    # time as the empty launch it brackets (100M x 100M: 2.02 ms per join without any timing, 2.22 ms with every launch timed)
    ctx = Context(profiling=2, max_gpu_memory=0)
    info = ctx.device_info()

    # ---- synthetic, device-resident inputs (counter-based: every rank generates its own shard) ----
    lo, hi = shard_range(n_total, rank, world)
    n_local = hi - lo
    bkeys = ctx.generate(N.GEN_BUILD_KEYS, n_local, modulus=n_total, first=lo)
    pkeys = ctx.generate(N.GEN_SM64_MOD_I64, n_local, modulus=n_total, seed=43, first=lo)
    if args.keys != "narrow":   # (functions of the key value: equal keys stay equal on both sides)
        for t in (bkeys, pkeys):
            if args.keys in ("offset", "two"):
                t += 1 << 40
            if args.keys == "two":
                t += (t & 1) << 41
            if args.keys == "window":
                t += (1 << 40) + (((t & 511) % 300) << 32)
            if args.keys == "random":
                t += ((t * 0x9E3779B1) & 0xFFFF) << 44
    torch.cuda.synchronize()
    join_info = None
    if not sharded:
        cap = n_total
        out_pairs = torch.empty((cap, 2), dtype=torch.int32, device=dev)
        out_count = torch.zeros(1, dtype=torch.int64, device=dev)

        def step():
            ctx.join(bkeys, pkeys, out_pairs=out_pairs, max_pairs=cap, strategy=N.JOIN_RADIX_HASH, out_count=out_count)
            return out_pairs, out_count
    else:
        if world > 1:
            from gpu_olap_engine_amd.distributed import NativeShardedJoin
            sj = NativeShardedJoin(ctx, transport="staged" if args.rehearse else "rccl")
            comm = sj.comm
        else:
            from gpu_olap_engine_amd.engine import comm_unique_id
            comm = ctx.comm_create_rccl(comm_unique_id(), 1, 0)
        cap = int(n_local * 1.25) + 65536
        out_pairs = torch.empty((cap, 2), dtype=torch.int32, device=dev)
        out_count = torch.zeros(1, dtype=torch.int64, device=dev)

        def step():
            comm.sharded_join(bkeys, pkeys, lo, lo, n_total, out_pairs=out_pairs, max_pairs=cap, out_count=out_count)
            return out_pairs, out_count

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return dt

    # ---- where the driver puts the workspace slab decides 3-4 % of the partition passes (DESIGN 9 "Run-to-run spread": same box, same
    # code, same virtual addresses -- 17.2 or 17.8 ms).  A long-lived context keeps whatever it drew; so before the warm-up this process
    # draws a few (a context = a slab), measures two joins on each and keeps the fastest.  Untimed set-up, every draw in the line.
    draws = None
    if not sharded and world == 1 and args.workspace_draws > 1:
        def two_joins_ms():
            step(); step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step(); step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / 2 * 1e3
        draws = {"draws": 1, "ms_per_join": [round(two_joins_ms(), 3)], "kept": 0,
                 "what": "contexts created before the warm-up, two untimed joins measured on each, the fastest kept: the physical placement of the "
                         "workspace slab decides 3-4 % (DESIGN.md 9, 'Run-to-run spread'); ms_per_join[0] is what a run without this would have drawn"}
        best, best_ms = ctx, draws["ms_per_join"][0]
        try:
            for d in range(1, args.workspace_draws):
                ctx = Context(profiling=2, max_gpu_memory=0)   # (`step` reads this variable)
                ms = two_joins_ms()
                draws["draws"] += 1
                draws["ms_per_join"].append(round(ms, 3))
                if ms < best_ms:
                    best.close()
                    best, best_ms, draws["kept"] = ctx, ms, d
                else:
                    ctx.close()
        except Exception as e:   # (out of memory with two slabs alive, ...): the draws so far stand
            draws["stopped_by"] = repr(e)[:200]
        ctx = best

    (pairs, count), dt = timed(step, args.steps, args.warmup, barrier, ctx)
    dt = max_over_ranks(dt)
    ms_per_step = dt / args.steps * 1e3
    stats = ctx.stats()
    if sharded:
        join_info = comm.info()
    m_local = int(count.item())

    # ---- size-independent correctness properties (every run, untimed) -------------------------
    verify = {}
    if not args.no_verify:
        m_tot = torch.tensor([m_local], dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_reduce(m_tot)
        verify["matches"] = int(m_tot.item())
        assert verify["matches"] == n_total, f"expected exactly one match per probe row: {verify['matches']} != {n_total}"
        if world == 1:   # key equality needs both key columns on one device
            v = ctx.verify_pairs(pairs, m_local, bkeys, pkeys)
            assert int(v[0].item()) == m_local, "emitted pairs with unequal keys"
            verify["equal_key_pairs"] = int(v[0].item())
            psum = int(v[2].item()) & (2**64 - 1)
        else:
            ids = pairs[:m_local, 1].to(torch.int64) & 0xFFFFFFFF
            s = ids.sum().reshape(1)
            dist.all_reduce(s)
            psum = int(s.item()) & (2**64 - 1)
        assert psum == (n_total * (n_total - 1) // 2) & (2**64 - 1), "probe row ids are not a permutation of 0..N-1"
        verify["probe_id_sum_ok"] = True

    alg_join = 40 * 2 * n_total + 8 * n_total          # 40*(Nb+Np) + 8*M
    join_gbps = alg_join / (ms_per_step * 1e-3) / 1e9
    line = {
        "metric": "join rows/sec, 1Bx1B int64 radix hash join" if label == "C5" else "join rows/sec, 100Mx100M int64 radix hash join",
        "value": round(2 * n_total / (ms_per_step * 1e-3), 1),
        "unit": "rows/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, host-staged transport -- not a measurement)" if args.rehearse else ""),
        "config": {"workload": f"{label} radix hash join {n_total} x {n_total} int64 keys (all below 2^32: the 8-byte tuple pipeline), 256 radix "
                               f"partitions, unique build keys, M = N_probe",
                   "n_build": n_total, "n_probe": n_total, "rows_per_gpu_per_side": n_local, "narrow_keys": args.keys == "narrow", "keys": args.keys,
                   "parallelism": f"partition-sharded x{world}" + (" + RCCL grouped send/recv (golap_sharded_join)" if sharded else ""),
                   "plan": ctx.join_plan(n_total if world == 1 else max(1, n_total // world))},
        "probe_rows_per_s": round(n_total / (ms_per_step * 1e-3), 1),
        "join_alg_GBps": round(join_gbps, 1),
        "join_roofline_frac": round(join_gbps / (HBM_PEAK_GBPS * world), 4),
        # SURVEY 8(d)'s contract figure for the WHOLE operator (not one kernel): B_join = 40 (N_b + N_p) + 8 M over the step time
        "alg_bytes": alg_join, "alg_roofline_frac": round(join_gbps / (HBM_PEAK_GBPS * world), 4),
        "traffic_bytes_per_step": step_traffic(stats, args.steps, n_total, world) if not sharded else None,
        "roofline": roofline_of(stats, n_total, world, not sharded, m_local),
        "kernels_ms_per_step": {k: round(v["total_ms"] / args.steps, 4) for k, v in stats.items()},
        "verify": verify,
        "device": device_block(info),
    }
    if draws:
        line["workspace_draws"] = draws
    if join_info:
        # where the time went on every rank (device timestamps of the fast path's stages, ms since that rank's first kernel)
        if world > 1:
            infos = [None] * world
            dist.all_gather_object(infos, {"stage_ms": join_info["stage_ms"], "total_ms": join_info["total_ms"], "bytes_sent": join_info["bytes_sent"],
                                           "fast_path": join_info["fast_path"], "exchange": exchange_rates(join_info, world)})
            join_info["per_rank"] = infos
            ex = [i["exchange"] for i in infos if i["exchange"]]
            if ex:   # the slowest rank's figures: what one xGMI link of this node moved per direction, measured (DESIGN 4.3's 153 vs 76.8 GB/s question)
                join_info["exchange"] = {k: min(e[k] for e in ex) for k in ex[0]}
        line["sharded_join"] = join_info

    # ---- further legs: the same join with keys outside [0, 2^32) -------------------------------
    # (a) both key columns + 2^40: every key is a genuine 64-bit value, all share one upper half -> still 8-byte tuples (frame of
    #     reference on the first build key); (b) two different upper halves -> the 64-bit-row pipeline (12-byte rows)
    if label == "C5" and not args.no_wide and not sharded and args.keys == "narrow":
        def leg(name, what):
            torch.cuda.synchronize()
            (pr, cn), dtw = timed(step, args.steps, args.warmup, barrier, ctx)
            msw = max_over_ranks(dtw) / args.steps * 1e3
            statsw = ctx.stats()
            mw = int(cn.item())
            assert mw == n_total, f"{name}: {mw} matches"
            if not args.no_verify:
                v = ctx.verify_pairs(pr, mw, bkeys, pkeys)
                assert int(v[0].item()) == mw, f"{name} emitted pairs with unequal keys"
            gb = alg_join / (msw * 1e-3) / 1e9
            return {"ms_per_step": round(msw, 3), "value": round(2 * n_total / (msw * 1e-3), 1), "unit": "rows/s", "join_alg_GBps": round(gb, 1),
                    "join_roofline_frac": round(gb / HBM_PEAK_GBPS, 4), "roofline": roofline_of(statsw, n_total, 1, name == "random_upper_halves", mw),   # (PMC traffic only for the 64-bit-row kernels: profiles/r04_bench_1b_wide)
                    "kernels_ms_per_step": {k: round(v["total_ms"] / args.steps, 4) for k, v in statsw.items() if v["total_ms"] / args.steps > 0.02},
                    "workload": what}
        bkeys += 1 << 40
        pkeys += 1 << 40
        wk = leg("offset_2p40", "the same generators + 2^40 on both key columns: every key outside [0, 2^32), one common upper half")
        for t in (bkeys, pkeys):
            t += ((t & 511) % 300) << 32
        wk["window_of_300_upper_halves"] = leg("window_of_300_upper_halves", "as above + (low 9 key bits mod 300) in the upper half: 300 consecutive upper halves "
                                               "(ids with a tenant / shard prefix) -> still 8-byte tuples, the offset of the upper half inside a window as wide as "
                                               "the first pass's fan-out rides in the tuple (before: the 64-bit-row pipeline)")
        for t in (bkeys, pkeys):
            t -= ((t & 511) % 300) << 32
        bkeys += (bkeys & 1) << 41
        pkeys += (pkeys & 1) << 41
        wk["two_upper_halves"] = leg("two_upper_halves", "as above, odd keys moved by another 2^41: two upper halves -> still 8-byte tuples, a dictionary "
                                     "index of the upper half rides in the tuple (up to 4 upper halves; round 2 sent these to the 64-bit-row pipeline)")
        bkeys += ((bkeys * 0x9E3779B1) & 0xFFFF) << 44
        pkeys += ((pkeys * 0x9E3779B1) & 0xFFFF) << 44
        wk["random_upper_halves"] = leg("random_upper_halves", "as above + a key-dependent 16-bit field at bit 44: 2^17 upper halves (hashed / snowflake-like "
                                        "ids) -> the 64-bit-row pipeline (12-byte rows, 120 GB of traffic for the 88 GB credited)")
        line["wide_keys"] = wk
    # C3 is timed at its full size on the CPU too (the reference's own CPU-runnable scale); C5 on a bounded 16M x 16M sample
    sample = args.cpu_sample if args.cpu_sample >= 0 else (n_total if label == "C3" else 16_000_000)
    cb = cpu_join_baseline(min(sample, n_total), rank) if world == 1 else None
    if cb:
        line["cpu_baseline"] = cb
    if sharded:
        comm.close()       # (the communicator before the context it was made on)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return line if rank == 0 else None


# ---------------------------------------------------------------------------------------------------------------------
# C4 / C4L: 1B-row GROUP BY int32 key SUM(float64)
# ---------------------------------------------------------------------------------------------------------------------
def bench_group_by(args, n, groups, label):
    import numpy as np
    import torch

    from gpu_olap_engine_amd import _native as N
    from gpu_olap_engine_amd.engine import Context, shard_range
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or args.sharded:
        return bench_group_by_sharded(args, n, groups, label)
    torch.cuda.set_device(0)
    ctx = Context(profiling=True, max_gpu_memory=0)
    info = ctx.device_info()
    keys = ctx.generate(N.GEN_SM64_MOD_I32, n, modulus=groups, seed=7)
    vals = ctx.generate(N.GEN_SM64_MOD_F64, n, modulus=1024, seed=8)
    torch.cuda.synchronize()

    def step():
        return ctx.aggregate([(N.AGG_SUM, 0)], [vals], group_keys=keys, n_groups=groups)

    out, dt = timed(step, args.steps, args.warmup, torch.cuda.synchronize, ctx)
    ms = dt / args.steps * 1e3
    stats = ctx.stats()
    verify = {}
    if not args.no_verify:   # exact: integer-valued doubles, every partial sum < 2^53; checked on the device
        exp = torch.zeros(groups, dtype=torch.float64, device="cuda")
        exp.index_add_(0, keys.to(torch.int64), vals)
        assert torch.equal(out[0], exp), "group sums differ from torch.index_add_"
        verify = {"groups_equal_index_add": True, "sum_of_sums": float(out[0].sum().item())}
    alg = n * (4 + 8) + groups * 12
    line = {"metric": f"group-by rows/sec, 1B-row GROUP BY int32 key SUM(float64), {groups} groups", "value": round(n / (ms * 1e-3), 1), "unit": "rows/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{label} GROUP BY int32 key SUM(float64), {n} rows, {groups} groups (dense keys), hash-aggregate with LDS pre-aggregation"},
            "agg_alg_GBps": round(alg / (ms * 1e-3) / 1e9, 1), "agg_roofline_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            # B_agg = N (4 + 8) + 12 G over the step time -- the contract fraction of the whole GROUP BY, whatever the
            # implementation moves (the partition-then-aggregate path at 2^20 groups moves three times B_agg)
            "alg_bytes": alg, "alg_roofline_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "traffic_bytes_per_step": step_traffic(stats, args.steps, n, 1),
            "roofline": roofline_of(stats, n, 1, True), "kernels_ms_per_step": {k: round(v["total_ms"] / args.steps, 4) for k, v in stats.items()},
            "verify": verify, "device": device_block(info)}
    sample = args.cpu_sample if args.cpu_sample >= 0 else 200_000_000
    if sample:
        import oracle
        s = min(sample, n)
        k = oracle.gen_sm64_mod(7, 0, s, groups, dtype=np.int32).astype(np.int64)
        v = oracle.gen_sm64_mod(8, 0, s, 1024).astype(np.float64)
        t0 = time.perf_counter()
        oracle.group_aggregate(k, v)
        t1 = time.perf_counter()
        line["cpu_baseline"] = {"value": s / (t1 - t0), "unit": "rows/s", "cores": 1, "kind": "port",
                                "sample": f"{s} rows of the same generators, single-thread sequential f64 accumulation per group "
                                          f"(executor.rs:313-327 extended by the group key); {t1 - t0:.2f} s"}
    ctx.close()
    return line


def bench_group_by_sharded(args, n, groups, label):
    """the same GROUP BY over a ROW-SHARDED table (rank r holds rows [r n / P, (r + 1) n / P)) through golap_sharded_aggregate: every
    rank aggregates its rows, the dense tables are all-gathered in chunks and folded in rank order, the result is replicated.  Strong
    scaling: the table's size is fixed.  One rank per GPU under torch.distributed.run; --sharded alone = the same entry point at world
    size 1; --rehearse = all ranks on GPU 0 over the staged (gloo) transport (functional rehearsal, not a measurement)."""
    import torch
    import torch.distributed as dist

    from gpu_olap_engine_amd import _native as N
    from gpu_olap_engine_amd.engine import Context, comm_unique_id, shard_range
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    ctx = Context(profiling=True, max_gpu_memory=0)
    info = ctx.device_info()
    if world > 1:
        from gpu_olap_engine_amd.distributed import NativeShardedJoin
        comm = NativeShardedJoin(ctx, transport="staged" if args.rehearse else "rccl").comm
    else:
        comm = ctx.comm_create_rccl(comm_unique_id(), 1, 0)
    lo, hi = shard_range(n, rank, world)
    keys = ctx.generate(N.GEN_SM64_MOD_I32, hi - lo, modulus=groups, seed=7, first=lo)
    vals = ctx.generate(N.GEN_SM64_MOD_F64, hi - lo, modulus=1024, seed=8, first=lo)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        return comm.sharded_aggregate([(N.AGG_SUM, 0)], [vals], group_keys=keys, n_groups=groups)

    out, dt = timed(step, args.steps, args.warmup, barrier, ctx)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    stats = ctx.stats()
    verify = {}
    if not args.no_verify:   # the replicated result against this rank's own rows summed over ranks (exact: integer-valued doubles)
        exp = torch.zeros(groups, dtype=torch.float64, device=dev)
        exp.index_add_(0, keys.to(torch.int64), vals)
        if world > 1:
            dist.all_reduce(exp, op=dist.ReduceOp.SUM)
        assert torch.equal(out[0], exp), "group sums differ from torch.index_add_ (+ all-reduce)"
        verify = {"groups_equal_index_add": True, "sum_of_sums": float(out[0].sum().item())}
    alg = n * (4 + 8) + groups * 12 * world
    line = {"metric": f"group-by rows/sec, 1B-row GROUP BY int32 key SUM(float64), {groups} groups", "value": round(n / (ms * 1e-3), 1), "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{label} GROUP BY int32 key SUM(float64), {n} rows row-sharded over {world} rank(s), {groups} groups (dense keys)",
                       "parallelism": f"row-sharded x{world}: local aggregate + all-gather of the dense tables + rank-order fold (golap_sharded_aggregate)"},
            "agg_alg_GBps": round(alg / (ms * 1e-3) / 1e9, 1), "agg_roofline_frac": round(alg / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBPS * world), 4),
            "roofline": roofline_of(stats, n // world, world, False), "kernels_ms_per_step": {k: round(v["total_ms"] / args.steps, 4) for k, v in stats.items()},
            "verify": verify, "device": device_block(info)}
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return line if rank == 0 else None


# ---------------------------------------------------------------------------------------------------------------------
# C1: SELECT sum(v) FROM t WHERE k < c on a 1M-row batch
# ---------------------------------------------------------------------------------------------------------------------
def bench_filter_sum(args, n):
    import torch

    from gpu_olap_engine_amd import _native as N
    from gpu_olap_engine_amd.engine import Context
    torch.cuda.set_device(0)
    ctx = Context(profiling=True, max_gpu_memory=0)
    info = ctx.device_info()
    k = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=1000, seed=1)
    v = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=1024, seed=2)
    torch.cuda.synchronize()

    def step():
        return ctx.filter_sum(k, v, "<", 500)

    out, dt = timed(step, args.steps, args.warmup, torch.cuda.synchronize, ctx)
    ms = dt / args.steps * 1e3
    stats = ctx.stats()
    verify = {}
    if not args.no_verify:
        sel = k < 500
        assert float(out[0].item()) == float(v[sel].sum().item()) and float(out[1].item()) == float(sel.sum().item())
        verify = {"sum": float(out[0].item()), "kept_rows": float(out[1].item())}
    alg = 16 * n
    line = {"metric": "filter+sum rows/sec, SELECT sum(v) FROM t WHERE k<c", "value": round(n / (ms * 1e-3), 1), "unit": "rows/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64", "data": "synthetic",
            "config": {"workload": f"C1 SELECT sum(v) FROM t WHERE k<500, {n}-row batch (k, v int64), fused predicate + sum in one pass"},
            "alg_GBps": round(alg / (ms * 1e-3) / 1e9, 1), "alg_bytes": alg, "alg_roofline_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6),
            "traffic_bytes_per_step": step_traffic(stats, args.steps, n, 1), "roofline": roofline_of(stats, n, 1, True),
            "kernels_ms_per_step": {kk: round(vv["total_ms"] / args.steps, 5) for kk, vv in stats.items()}, "verify": verify, "device": device_block(info)}
    sample = args.cpu_sample if args.cpu_sample >= 0 else n
    if sample:
        import oracle
        s = min(sample, n)
        kk, vv = oracle.gen_sm64_mod(1, 0, s, 1000), oracle.gen_sm64_mod(2, 0, s, 1024)
        reps = max(1, 200_000_000 // s)
        t0 = time.perf_counter()
        for _ in range(reps):
            oracle.filter_sum(kk, vv, "<", 500)
        t1 = time.perf_counter()
        line["cpu_baseline"] = {"value": s * reps / (t1 - t0), "unit": "rows/s", "cores": 1, "kind": "port",
                                "sample": f"{reps} x {s} rows, the reference's 4-pass CPU path (executor.rs:161-178 + :297-355) restated; {t1 - t0:.2f} s"}
    ctx.close()
    return line


# ---------------------------------------------------------------------------------------------------------------------
# OP: Arrow in -> Arrow out through the operator layer ("H2D excluded and reported separately" -- here it is reported)
# ---------------------------------------------------------------------------------------------------------------------
def bench_operator_plan(args, n):
    """GpuAggregate{ GpuHashJoin{ customers, GpuFilter{ orders, flag < 5 } }, [seg], [Sum(amt), Count(amt)] } over pyarrow batches in
    (pageable) HOST memory: what the reference's GpuExecutor::execute is handed.  Every step is one golap_execute call: Arrow C Data in,
    upload (each column once, at its Arrow width), device-resident filter / join / aggregate, the group table back as Arrow.  The line
    says where the wall time went: upload and download phases with their bytes (PCIe), the rest is kernels + host bookkeeping."""
    import numpy as np
    import pyarrow as pa
    import torch

    from gpu_olap_engine_amd import operators as O
    from gpu_olap_engine_amd.engine import Context
    torch.cuda.set_device(0)
    ctx = Context(profiling=False, max_gpu_memory=0)
    info = ctx.device_info()
    O.load_host().golap_host_set_pipelining(1 if args.pipelining else 0)
    n_c = max(1000, n // 100)
    rng = np.random.default_rng(5)
    o_cust = rng.integers(0, n_c + n_c // 10, n, dtype=np.int64)            # 1 in 11 orders references nobody
    o_amt = rng.integers(1, 1000, n).astype(np.float64)                     # integer-valued: sums are exact in any order
    o_flag = rng.integers(0, 10, n, dtype=np.int64)
    c_id = rng.permutation(n_c).astype(np.int64)
    c_seg = rng.integers(0, 1000, n_c).astype(np.int32)
    nb = 4                                                                   # orders arrive as 4 RecordBatches
    cuts = [n * i // nb for i in range(nb + 1)]
    orders = [pa.RecordBatch.from_arrays([pa.array(o_cust[a:b]), pa.array(o_amt[a:b]), pa.array(o_flag[a:b])], names=["cust", "amt", "flag"])
              for a, b in zip(cuts[:-1], cuts[1:])]
    customers = [pa.RecordBatch.from_arrays([pa.array(c_id), pa.array(c_seg)], names=["id", "seg"])]

    def plan():
        return O.GpuAggregate(O.GpuHashJoin(O.Batches(customers), O.GpuFilter(O.Batches(orders), O.BinaryExpr(O.Column(2, "flag"), "<", O.Literal("5"))),
                                            [O.Column(0, "id")], [O.Column(0, "cust")]),
                              [O.Column(1, "seg")], [O.Sum(O.Column(3, "amt")), O.Count(O.Column(3, "amt"))])
    profs = []
    for _ in range(args.warmup):
        out = O.execute(plan(), ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pr = {}
        out = O.execute(plan(), ctx, profile=pr)
        profs.append(pr)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    avg = {k: sum(p[k] for p in profs) / len(profs) for k in profs[0]}
    verify = {}
    if not args.no_verify:
        keep = (o_flag < 5) & (o_cust < n_c)
        seg_of = np.empty(n_c, np.int64)
        seg_of[c_id] = c_seg
        seg = seg_of[o_cust[keep]]
        exp_sum = np.bincount(seg, weights=o_amt[keep], minlength=1000)
        exp_cnt = np.bincount(seg, minlength=1000)
        present = np.nonzero(exp_cnt)[0]
        assert out[0].column(0).to_pylist() == present.tolist(), "group keys differ from numpy"
        assert np.array_equal(out[0].column(1).to_numpy(), exp_sum[present]) and np.array_equal(out[0].column(2).to_numpy(), exp_cnt[present].astype(np.float64))
        verify = {"groups": int(present.size), "joined_rows": int(keep.sum()), "sum_of_sums": float(exp_sum.sum())}
    in_bytes = n * 24 + n_c * 12
    line = {"metric": "operator-plan input rows/sec, filter -> hash join -> GROUP BY over host-resident Arrow batches", "value": round(n / (ms * 1e-3), 1), "unit": "rows/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64", "data": "synthetic",
            "config": {"workload": f"OP GpuAggregate{{GpuHashJoin{{customers {n_c} rows, GpuFilter{{orders {n} rows in {nb} batches, flag < 5}}}}, GROUP BY seg, SUM + COUNT}} "
                                   f"through golap_plan_* / golap_execute, inputs in pageable host memory (pyarrow), result back as Arrow"},
            "end_to_end_ms": round(ms, 3), "input_bytes": in_bytes,
            "pipelined_filter": bool(args.pipelining),   # batch k filtered while batch k + 1 is uploaded (h2d_ms then holds those kernels too); off by default
            "h2d_bytes": int(avg["h2d_bytes"]), "h2d_ms": round(avg["h2d_ms"], 3), "h2d_GBps": round(avg["h2d_bytes"] / max(avg["h2d_ms"], 1e-9) / 1e6, 2),
            "d2h_bytes": int(avg["d2h_bytes"]), "d2h_ms": round(avg["d2h_ms"], 3), "d2h_GBps": round(avg["d2h_bytes"] / max(avg["d2h_ms"], 1e-9) / 1e6, 4),
            "device_and_host_ms": round(avg["total_ms"] - avg["h2d_ms"] - avg["d2h_ms"], 3), "kernel_wait_ms": round(avg["kernel_wait_ms"], 3),
            "alg_bytes": in_bytes, "alg_roofline_frac": round(in_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5), "traffic_bytes_per_step": None,
            "roofline": None, "kernels_ms_per_step": {}, "verify": verify, "device": device_block(info)}
    sample = args.cpu_sample if args.cpu_sample >= 0 else min(n, 4_000_000)
    if sample:   # the same plan on the host cores, single thread, with the oracle's pieces (4-pass filter, HashMap join, sequential GROUP BY)
        import oracle
        s = min(sample, n)
        t0 = time.perf_counter()
        mask = oracle.eval_filter_mask(o_flag[:s], "<", 5).astype(bool)
        idx = np.nonzero(mask)[0]
        pairs, m = oracle.hash_join(c_id, o_cust[:s][idx])
        seg_rows = c_seg[pairs[:, 0].astype(np.int64)].astype(np.int64)
        oracle.group_aggregate(seg_rows, o_amt[:s][idx][pairs[:, 1].astype(np.int64)])
        t1 = time.perf_counter()
        line["cpu_baseline"] = {"value": s / (t1 - t0), "unit": "rows/s", "cores": 1, "kind": "port",
                                "sample": f"the first {s} orders against all {n_c} customers: the reference's CPU pieces restated (filter mask executor.rs:443-476, "
                                          f"HashMap build + probe :238-253, sequential f64 GROUP BY :313-327), single thread; {t1 - t0:.2f} s"}
    ctx.close()
    return line


def run_config(args, config):
    if config == "C5":
        return bench_join(args, args.rows or 1_000_000_000, "C5")
    if config == "C3":
        return bench_join(args, args.rows or 100_000_000, "C3")
    if config == "C4":
        return bench_group_by(args, args.rows or 1_000_000_000, 1024, "C4")
    if config == "C4L":
        return bench_group_by(args, args.rows or 1_000_000_000, 1_048_576, "C4L")
    if config == "OP":
        return bench_operator_plan(args, args.rows or 100_000_000)
    return bench_filter_sum(args, args.rows or 1_000_000)


def other_configs(args):
    """The other BASELINE configurations at their full sizes, a few steps each: every entry is verified like its stand-alone run and
    carries the contract figure of SURVEY 8(d) for the whole operator (alg_bytes, alg_roofline_frac = alg_bytes / step time / 8 TB/s)
    next to its dominant kernel's own roofline, the step's PMC traffic where profiles/pmc_traffic.json has its kernels, and a short
    CPU baseline (a bounded sample, about a second each; `--config X` alone times the larger default samples)."""
    import copy
    import gc

    import torch
    out = {}
    cpu_rows = {"C3": 4_000_000, "C4": 50_000_000, "C4L": 25_000_000, "C1": 1_000_000, "OP": 2_000_000}
    for cfg in ("C3", "C4", "C4L", "C1", "OP"):
        a = copy.copy(args)
        a.config, a.rows, a.no_wide, a.sharded = cfg, 0, True, False
        a.cpu_sample = 0 if args.cpu_sample == 0 else cpu_rows[cfg]
        a.steps, a.warmup = (50, 5) if cfg == "C1" else (3, 1) if cfg == "OP" else (5, 2)
        gc.collect()
        torch.cuda.empty_cache()
        ln = run_config(a, cfg)
        r = ln.get("roofline") or {}
        out[cfg] = {"workload": ln["config"]["workload"], "ms_per_step": ln["ms_per_step"], "value": ln["value"], "unit": ln["unit"], "steps": a.steps,
                    "alg_bytes": ln["alg_bytes"], "alg_roofline_frac": ln["alg_roofline_frac"], "traffic_bytes_per_step": ln["traffic_bytes_per_step"],
                    "roofline": {"kernel": r.get("kernel"), "achieved": r.get("achieved"), "frac": r.get("frac"), "unit": r.get("unit"),
                                 "traffic": r.get("traffic"), "alg_bytes_per_launch": r.get("alg_bytes_per_launch")},
                    "kernels_ms_per_step": ln["kernels_ms_per_step"], "verify": ln["verify"]}
        if cfg == "OP":   # Arrow in -> Arrow out: where the wall time went (PCIe phases with their bytes; the rest is kernels + host bookkeeping)
            out[cfg].update({k: ln[k] for k in ("end_to_end_ms", "h2d_bytes", "h2d_ms", "h2d_GBps", "d2h_bytes", "d2h_ms", "d2h_GBps", "device_and_host_ms", "kernel_wait_ms")})
        cb = ln.get("cpu_baseline")
        if cb:
            out[cfg]["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample")}
        if "workspace_draws" in ln:
            out[cfg]["workspace_draws"] = {k: ln["workspace_draws"][k] for k in ("draws", "ms_per_join", "kept")}
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    line = run_config(args, args.config)
    # the driver runs the default line only: it carries the other BASELINE configurations too (one GPU, < 2 s of GPU time)
    if line is not None and args.config == "C5" and args.gpus == 1 and not args.sharded and not args.rehearse and not args.rows and not args.no_others:
        line["other_configs"] = other_configs(args)
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
