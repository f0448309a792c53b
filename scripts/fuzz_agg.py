"""Randomised parity fuzz of golap_aggregate / golap_filter_* against the CPU oracle (developer tool, not product).
usage (GPU box): python scripts/fuzz_agg.py SECONDS [SEED]

Grouped aggregates over the three device paths (LDS pre-aggregation, partition-then-aggregate with both tile kernels,
global atomics), int32/int64 keys (dense, sparse, skewed, out-of-range keys that fold by key % n_groups), int64/float64
integer-valued inputs (so every summation order is bit-identical); filters with every comparison operator."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import oracle
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context

SPECS = [(N.AGG_SUM, 0), (N.AGG_COUNT, 0), (N.AGG_MIN, 0), (N.AGG_MAX, 0), (N.AGG_AVG, 0)]
SPECS_SUM = [(N.AGG_SUM, 0), (N.AGG_COUNT, 0), (N.AGG_AVG, 0)]   # no MIN / MAX: 12-byte table entries, the write-combining / reserving partition scatters


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    torch.cuda.set_device(0)
    ctx = Context(profiling=False, max_gpu_memory=0)
    t_end = time.time() + budget
    rnd = 0
    while time.time() < t_end:
        rnd += 1
        why = None
        size = rng.random()
        n = int(rng.integers(1, 50_000)) if size < 0.4 else int(rng.integers(50_000, 8_000_000)) if size < 0.9 else int(rng.integers(34_000_000, 45_000_000))
        G = int(rng.choice([1, 7, 300, 1024, 2340, 2341, 5000, 5558, 5559, 12_969, 12_970, 100_000, 1_000_003, 2_097_152, 3_000_000]))   # (5558 / 12969: the last group counts of ONE LDS table with / without MIN / MAX)
        kd = rng.choice([np.int32, np.int64])
        shape = str(rng.choice(["dense", "skewed", "sparse", "fold"]))
        if shape == "dense":
            keys = rng.integers(0, G, n)
        elif shape == "skewed":
            keys = rng.integers(0, G, n)
            keys[rng.random(n) < 0.5] = int(rng.integers(0, G))
        elif shape == "sparse":
            keys = rng.integers(0, max(1, G // 50), n) * 50 % G
        else:                                            # keys outside [0, G): merged by key % n_groups (join_kernels.cuh:230)
            keys = rng.integers(0, 4 * G + 5, n)
        keys = keys.astype(kd)
        vf = rng.random() < 0.5
        vals = (rng.integers(-2048, 2048, n)).astype(np.float64 if vf else np.int64)
        env = {}
        if rng.random() < 0.15:
            env["GOLAP_AGG_PART"] = "0"
        ctx.debug_set("GOLAP_AGG_PART", env.get("GOLAP_AGG_PART"))
        slots = (keys.astype(np.int64) % G)
        vvalid = (rng.random(n) > 0.2).astype(np.uint8) if rng.random() < 0.3 else None      # NULL values: skipped by every aggregate
        live = np.ones(n, bool) if vvalid is None else vvalid != 0
        ok, ost = oracle.group_aggregate(slots[live], vals.astype(np.float64)[live])
        sums_only = rng.random() < 0.5
        if sums_only and rng.random() < 0.3:
            env["GOLAP_AGG_RSV"] = str(rng.choice(["0", "plain"]))
        ctx.debug_set("GOLAP_AGG_RSV", env.get("GOLAP_AGG_RSV"))
        out, counts = ctx.aggregate(SPECS_SUM if sums_only else SPECS, [torch.from_numpy(vals).cuda()], group_keys=torch.from_numpy(keys).cuda(), n_groups=G,
                                    want_counts=True, value_valid=None if vvalid is None else [torch.from_numpy(vvalid).cuda()])
        out, counts = out.cpu().numpy(), counts.cpu().numpy()
        present = counts > 0
        good = np.array_equal(np.nonzero(present)[0], ok) and np.array_equal(out[:, present].T, ost[:, [0, 1, 4]] if sums_only else ost)
        why = why or (None if good else "dense")
        # the same rows through the key table (GOLAP_KEYS_HASHED): arbitrary keys, one slot per distinct key
        if rng.random() < 0.4 and G <= 3_000_000:
            hk = keys.astype(np.int64)
            mode = rng.random()
            if mode < 0.4:
                hk = hk * int(rng.integers(1, 1 << 20)) - int(rng.integers(0, 1 << 40))      # sparse, negative
            elif mode < 0.6 and kd is np.int64:
                hk = hk ^ np.int64(-9223372036854775808)                                      # around INT64_MIN (the table's empty word)
            if kd is np.int32:
                hk = hk.astype(np.int32)
            hk64 = hk.astype(np.int64)
            ek, est = oracle.group_aggregate(hk64, vals.astype(np.float64))
            bound = min(int(ek.size * rng.choice([1, 2, 5])) + int(rng.integers(0, 3)), 1 << 22)   # (the key table limit)
            ho, hc, hkeys, hn = ctx.aggregate(SPECS, [torch.from_numpy(vals).cuda()], group_keys=torch.from_numpy(hk).cuda(), n_groups=max(1, bound),
                                              want_counts=True, hashed=True)
            nd = int(hn.item()) & 0xFFFFFFFF
            if bound >= ek.size:
                sk = hkeys.cpu().numpy()[:nd]
                order = np.argsort(sk, kind="stable")
                good = good and nd == ek.size and np.array_equal(sk[order], ek) and np.array_equal(ho.cpu().numpy()[:, :nd][:, order].T, est)
                why = why or (None if good else "hashed")
            else:
                good = good and nd > bound
                why = why or (None if good else "hashed-bound")
        # a filter on the same column (sometimes with NULLs)
        op = int(rng.integers(0, 6))
        lit = int(rng.integers(-2100, 2100))
        col = vals.astype(np.int64)
        mask = ctx.filter_mask(torch.from_numpy(col).cuda(), op, lit).cpu().numpy().astype(bool)
        exp_mask = [col == lit, col != lit, col < lit, col <= lit, col > lit, col >= lit][op]
        idx, cnt = ctx.filter_compact(torch.from_numpy(col).cuda(), op, lit)
        k = int(cnt.item())
        good = good and np.array_equal(mask, exp_mask) and k == int(exp_mask.sum()) and np.array_equal(idx[:k].cpu().numpy().view(np.uint32), np.nonzero(exp_mask)[0].astype(np.uint32))
        why = why or (None if good else "filter")
        if rng.random() < 0.4:
            kv = (rng.random(n) > 0.25).astype(np.uint8)
            vv = (rng.random(n) > 0.25).astype(np.uint8)
            em = exp_mask & (kv != 0)
            tkv, tvv = torch.from_numpy(kv).cuda(), torch.from_numpy(vv).cuda()
            m2 = ctx.filter_mask(torch.from_numpy(col).cuda(), op, lit, valid=tkv).cpu().numpy().astype(bool)
            idx2, cnt2 = ctx.filter_compact(torch.from_numpy(col).cuda(), op, lit, valid=tkv)
            k2 = int(cnt2.item())
            fs = ctx.filter_sum(torch.from_numpy(col).cuda(), torch.from_numpy(vals).cuda(), op, lit, key_valid=tkv, value_valid=tvv).cpu().numpy()
            good = good and np.array_equal(m2, em) and k2 == int(em.sum()) and np.array_equal(idx2[:k2].cpu().numpy().view(np.uint32), np.nonzero(em)[0].astype(np.uint32))
            why = why or (None if good else "filter-null")
            both = em & (vv != 0)                            # NULL values are skipped by the sum AND the count (SQL aggregates, golap.h)
            good = good and fs[1] == float(both.sum()) and fs[0] == float(vals[both].astype(np.float64).sum())
            why = why or (None if good else "filter-sum-null")
        print(("ok   " if good else "FAIL ") + f"seed={seed} round={rnd} n={n} G={G} key={np.dtype(kd).name} shape={shape} val={'f64' if vf else 'i64'} nulls={vvalid is not None} env={env} op={op} lit={lit}" + (f" FAILED AT {why}" if why else ""), flush=True)
        if not good:
            sys.exit(1)
    print(f"fuzz: {rnd} rounds, all equal to the oracle", flush=True)


if __name__ == "__main__":
    main()
