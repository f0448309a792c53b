"""Developer tool: the radix hash join over the whole size range of one MI355X -- 1M x 1M (BASELINE config 2) ... 2B x 2B rows -- with the
bench's generators (unique build keys, one match per probe row), every size verified on the device (match count, key equality of every
pair, probe ids a permutation) and priced against SURVEY 8(d)'s contract bytes (40 (Nb + Np) + 8 M).  One JSON object per size on
stdout; `--keys random` runs the 64-bit-row pipeline (2^16 upper key halves).
usage: python scripts/size_curve.py [--keys narrow|random] [N ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context

args = sys.argv[1:]
keys = "narrow"
if args and args[0] == "--keys":
    keys = args[1]
    args = args[2:]
sizes = [int(x) for x in args] or [1_000_000, 4_000_000, 16_000_000, 32_000_000, 64_000_000, 100_000_000, 200_000_000, 400_000_000,
                                   700_000_000, 1_000_000_000, 1_500_000_000, 2_000_000_000]
torch.cuda.set_device(0)
ctx = Context(profiling=2, max_gpu_memory=0)
for n in sizes:
    rec = {"n_build": n, "n_probe": n, "keys": keys}
    try:
        b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n)
        p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
        if keys == "random":
            for t in (b, p):
                t += ((t * 0x9E3779B1) & 0xFFFF) << 44
        out = torch.empty((n, 2), dtype=torch.int32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        strategy = N.JOIN_RADIX_HASH

        def step():
            ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=strategy, out_count=cnt)

        iters = max(3, min(50, int(2_000_000_000 // n)))
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ctx.stats_reset()
        t0 = time.perf_counter()
        for _ in range(iters):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        stats = ctx.stats()
        assert int(cnt.item()) == n, f"{int(cnt.item())} matches for {n} probe rows"
        v = ctx.verify_pairs(out, n, b, p)
        assert int(v[0].item()) == n, "pairs with unequal keys"
        assert (int(v[2].item()) & (2**64 - 1)) == (n * (n - 1) // 2) & (2**64 - 1), "probe ids are not a permutation"
        alg = 40 * 2 * n + 8 * n
        plan = ctx.join_plan(n)
        rec.update(ms_per_join=round(ms, 4), rows_per_s=round(2 * n / (ms * 1e-3), 1), alg_bytes=alg,
                   alg_roofline_frac=round(alg / (ms * 1e-3) / 8e12, 4), iters=iters, verified=True,
                   bits=[plan["bits1"], plan["bits2"]], n_subpartitions=plan["n_subpartitions"],
                   kernels_ms_per_join={k: round(s["total_ms"] / iters, 4) for k, s in stats.items() if s["launches"] and s["total_ms"] / iters > 0.004 * ms})
        del out, b, p
    except Exception as e:   # (a size this box cannot hold, a refused plan ...): recorded, the sweep goes on
        rec["error"] = repr(e)[:300]
    torch.cuda.empty_cache()
    print(json.dumps(rec), flush=True)
