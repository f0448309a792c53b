"""Randomised parity fuzz of golap_join against the CPU oracle (developer tool; test infrastructure, not product).
usage (GPU box): python scripts/fuzz_join.py SECONDS [SEED]

Every round draws sizes, a key distribution, a join type, a strategy and (sometimes) experiment hooks that force the
rarely-taken device paths (pass structures, no-histogram passes and their overflow fallbacks), runs the HIP join through the
C ABI and compares the canonicalised pairs bit for bit with oracle.hash_join_outer (a third of the rounds: composite keys and
NULLs against oracle.hash_join_multi).  Prints one line per round; exits 1 on the
first mismatch with the recipe that reproduces it."""
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
from np_twin import canonical

HOOKS = ["GOLAP_PASS_BITS", "GOLAP_RESERVE", "GOLAP_RESERVE_SIGMAS", "GOLAP_SLOT_SIGMAS", "GOLAP_NOHIST", "GOLAP_SCATTER", "GOLAP_GJ_HASH_BITS"]


def draw_keys(rng, n, kind, domain):
    if n == 0:
        return np.empty(0, np.int64)
    if kind == "uniform":
        return rng.integers(0, domain, n)
    if kind == "unique":
        return rng.permutation(max(domain, n))[:n].astype(np.int64)
    if kind == "hot":                                   # a few keys carry most rows
        hot = rng.integers(0, domain, 4)
        k = rng.integers(0, domain, n)
        m = rng.random(n) < 0.3
        k[m] = hot[rng.integers(0, 4, int(m.sum()))]
        return k
    if kind == "fold":                                  # keys whose hi^lo fold collides (same hash, distinct keys)
        i = rng.integers(0, max(2, domain // 64), n).astype(np.int64)
        return (i << 32) | (i ^ rng.integers(0, 4, n))
    if kind == "wide":
        return rng.integers(-2**62, 2**62, n)
    raise ValueError(kind)


GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def _mix64(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def surrogates(cols, valids, side_seed, keep):
    """k_surrogate (csrc/golap_join.hip) restated: the 64-bit surrogate key of every row, narrowed by the GOLAP_GJ_HASH_BITS test knob"""
    n = cols[0].size
    null = np.zeros(n, bool)
    for c, v in enumerate(valids if valids is not None else []):
        if v is not None:
            null |= v == 0
    with np.errstate(over="ignore"):
        if len(cols) == 1:
            s = np.ascontiguousarray(cols[0], dtype=np.int64).view(np.uint64).copy()
        else:
            s = np.full(n, 0x243F6A8885A308D3, np.uint64)
            for c, col in enumerate(cols):
                s = _mix64(s ^ np.ascontiguousarray(col, dtype=np.int64).view(np.uint64)) + GOLDEN * np.uint64(c + 1)
        s_null = _mix64(np.uint64(side_seed) + np.arange(n, dtype=np.uint64) * GOLDEN)
    return np.where(null, s_null, s) & np.uint64(keep)


def false_pairs(bc, pc, bv, pv, bits, true_inner):
    """candidate pairs of the surrogate join that the real columns refute: what k_gj_verify will find (<= 64 are repaired, more set the
    count word to 2^64 - 1, include/golap.h)"""
    keep = (1 << bits) - 1
    sb = surrogates(bc, bv, 0x6A09E667F3BCC908, keep)
    sp = surrogates(pc, pv, 0xBB67AE8584CAA73B, keep)
    ub, cb = np.unique(sb, return_counts=True)
    up, cp = np.unique(sp, return_counts=True)
    _, ib, ip = np.intersect1d(ub, up, assume_unique=True, return_indices=True)
    return int((cb[ib].astype(np.int64) * cp[ip].astype(np.int64)).sum()) - int(true_inner)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    dry = os.environ.get("FUZZ_DRY") == "1"           # CPU side only: the recipes and their false-pair counts, no device
    if not dry:
        torch.cuda.set_device(0)
        ctx = Context(profiling=False, max_gpu_memory=0)
    t_end = time.time() + budget
    rnd = 0
    while time.time() < t_end:
        rnd += 1
        big = rng.random() < 0.35
        nb = int(rng.integers(0, 6_000_000 if big else 200_000))
        npr = int(rng.integers(0, 6_000_000 if big else 200_000))
        kind = str(rng.choice(["uniform", "unique", "hot", "fold", "wide"]))
        domain = int(max(1, rng.choice([nb // 4 + 1, nb + 1, 4 * nb + 1, 97])))
        jt = int(rng.integers(0, 4))
        strategy = N.JOIN_RADIX_HASH if (rng.random() < 0.8 or nb > 2_000_000) else N.JOIN_BROADCAST_HASH
        env = {}
        if strategy == N.JOIN_RADIX_HASH and rng.random() < 0.7:
            env["GOLAP_PASS_BITS"] = str(rng.choice(["8,6", "8,5", "9,5", "7,7", "8,4,4", "10,6", "8"]))
            if rng.random() < 0.6:
                env["GOLAP_RESERVE"] = "1"
            if rng.random() < 0.3:
                env["GOLAP_RESERVE_SIGMAS"] = str(rng.choice(["0", "1"]))
            if rng.random() < 0.3:
                env["GOLAP_SLOT_SIGMAS"] = str(rng.choice(["0", "1"]))
            if rng.random() < 0.1:
                env["GOLAP_NOHIST"] = "0"
            if rng.random() < 0.1:
                env["GOLAP_SCATTER"] = str(rng.choice(["v1", "wc", "big"]))
        b = draw_keys(rng, nb, kind, domain)
        p = draw_keys(rng, npr, kind if rng.random() < 0.7 else "uniform", domain)
        if nb and npr and rng.random() < 0.5:           # make sure something matches
            take = rng.integers(0, nb, min(npr, 1000))
            p[:take.size] = b[take]
        if nb and npr:                                  # quadratic blow-ups are not what this is about: count matches first
            ub, cb = np.unique(b, return_counts=True)
            up, cp = np.unique(p, return_counts=True)
            _, ib, ip = np.intersect1d(ub, up, assume_unique=True, return_indices=True)
            if int((cb[ib].astype(np.int64) * cp[ip]).sum()) > 30_000_000:
                continue
        # a third of the rounds: composite keys (the first column split into 2..4) and / or NULLs
        ncols, with_nulls = 1, False
        if rng.random() < 0.33:
            ncols = int(rng.integers(1, 5))
            with_nulls = rng.random() < 0.6 or ncols == 1
            if rng.random() < 0.25 and nb and npr:   # narrow surrogates so that a handful of false pairs occur (<= 64 can be repaired)
                env["GOLAP_GJ_HASH_BITS"] = str(int(min(63, max(8, np.ceil(np.log2(nb * npr / 4.0))))))
        dt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda() if a.size else torch.empty(0, dtype=torch.from_numpy(a[:0]).dtype, device="cuda")
        recipe = (f"seed={seed} round={rnd} nb={nb} np={npr} kind={kind} domain={domain} jt={jt} strategy={strategy} ncols={ncols} "
                  f"nulls={with_nulls} env={env}")
        if ncols > 1 or with_nulls:
            def split(k):     # an injective split of one key into ncols columns (so the match structure is the single-column one)
                k = np.ascontiguousarray(k, dtype=np.int64)
                if ncols == 1:
                    return [k]
                cols = [(k >> (16 * i)) & 0xFFFF for i in range(ncols - 1)]
                return [c - 300 for c in cols] + [k >> (16 * (ncols - 1))]
            bc, pc = split(b), split(p)
            bv = pv = None
            if with_nulls:
                mk = lambda n: [None if rng.random() < 0.3 else (rng.random(n) > 0.15).astype(np.uint8) for _ in range(ncols)]
                bv, pv = mk(nb), mk(npr)
            exp, me = oracle.hash_join_multi(bc, pc, jt, bv, pv)
            # The knob narrows the surrogates so that the verify / repair route runs.  How many false pairs that makes depends on the KEYS (one
            # column: the surrogate is the key itself, so structured keys collide in bulk): counted here, the buffer sized for them, and
            # beyond 64 the library's answer must be the impossible count 2^64 - 1 instead of a repaired list.
            n_false, general = 0, ncols > 1 or any(v is not None for v in (bv or []) + (pv or []))
            if "GOLAP_GJ_HASH_BITS" in env and general and nb and npr:
                n_false = false_pairs(bc, pc, bv, pv, int(env["GOLAP_GJ_HASH_BITS"]), oracle.hash_join_multi(bc, pc, 0, bv, pv)[1])
                recipe += f" false_pairs={n_false}"
                if n_false > 5_000_000:
                    continue
            if dry:
                print("dry  " + recipe + f" matches={me}", flush=True)
                continue
            for k in HOOKS:
                ctx.debug_set(k, env.get(k))
            tv = lambda v: None if v is None else [None if x is None else dt(x) for x in v]
            pairs, cnt = ctx.join([dt(c) for c in bc], [dt(c) for c in pc], max_pairs=max(1, me + n_false + 72), strategy=strategy, join_type=jt,
                                  build_valid=tv(bv), probe_valid=tv(pv))
            if n_false > 64:
                torch.cuda.synchronize()
                ok = int(cnt.item()) == -1                       # 2^64 - 1 read as int64
                print(("ok   " if ok else "FAIL ") + recipe + f" matches={me} (more false pairs than can be repaired: the impossible count)", flush=True)
                if not ok:
                    sys.exit(1)
                continue
        else:
            exp, me = oracle.hash_join_outer(b, p, jt)
            if dry:
                print("dry  " + recipe + f" matches={me}", flush=True)
                continue
            for k in HOOKS:
                ctx.debug_set(k, env.get(k))
            pairs, cnt = ctx.join(dt(np.ascontiguousarray(b, dtype=np.int64)), dt(np.ascontiguousarray(p, dtype=np.int64)), max_pairs=max(1, me + 8),
                                  strategy=strategy, join_type=jt)
        torch.cuda.synchronize()
        m = int(cnt.item())
        ok = m == me and np.array_equal(canonical(pairs[:m].cpu().numpy().view(np.uint32)), exp)
        print(("ok   " if ok else "FAIL ") + recipe + f" matches={me}", flush=True)
        if not ok:
            sys.exit(1)
    print(f"fuzz: {rnd} rounds, all equal to the oracle", flush=True)


if __name__ == "__main__":
    main()
