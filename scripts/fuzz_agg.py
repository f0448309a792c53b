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
        size = rng.random()
        n = int(rng.integers(1, 50_000)) if size < 0.4 else int(rng.integers(50_000, 8_000_000)) if size < 0.9 else int(rng.integers(34_000_000, 45_000_000))
        G = int(rng.choice([1, 7, 300, 1024, 2340, 2341, 5000, 100_000, 1_000_003, 2_097_152, 3_000_000]))
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
        ok, ost = oracle.group_aggregate(slots, vals.astype(np.float64))
        out, counts = ctx.aggregate(SPECS, [torch.from_numpy(vals).cuda()], group_keys=torch.from_numpy(keys).cuda(), n_groups=G, want_counts=True)
        out, counts = out.cpu().numpy(), counts.cpu().numpy()
        present = counts > 0
        good = np.array_equal(np.nonzero(present)[0], ok) and np.array_equal(out[:, present].T, ost)
        # a filter on the same column
        op = int(rng.integers(0, 6))
        lit = int(rng.integers(-2100, 2100))
        col = vals.astype(np.int64)
        mask = ctx.filter_mask(torch.from_numpy(col).cuda(), op, lit).cpu().numpy().astype(bool)
        exp_mask = [col == lit, col != lit, col < lit, col <= lit, col > lit, col >= lit][op]
        idx, cnt = ctx.filter_compact(torch.from_numpy(col).cuda(), op, lit)
        k = int(cnt.item())
        good = good and np.array_equal(mask, exp_mask) and k == int(exp_mask.sum()) and np.array_equal(idx[:k].cpu().numpy().view(np.uint32), np.nonzero(exp_mask)[0].astype(np.uint32))
        print(("ok   " if good else "FAIL ") + f"seed={seed} round={rnd} n={n} G={G} key={np.dtype(kd).name} shape={shape} val={'f64' if vf else 'i64'} env={env} op={op} lit={lit}", flush=True)
        if not good:
            sys.exit(1)
    print(f"fuzz: {rnd} rounds, all equal to the oracle", flush=True)


if __name__ == "__main__":
    main()
