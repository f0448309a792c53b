"""Developer tool: join time per size, default rule vs the no-histogram tuple pipeline forced below its threshold (GOLAP_RESERVE=1).
usage: python scripts/size_sweep.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context
sizes = [int(x) for x in sys.argv[1:]] or [2_000_000, 4_000_000, 8_000_000, 12_000_000, 16_000_000, 20_000_000, 24_000_000, 26_000_000, 32_000_000, 48_000_000]
torch.cuda.set_device(0)
ctx = Context(max_gpu_memory=0)
for n in sizes:
    b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n); p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
    out = torch.empty((n, 2), dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    res = []
    for forced in (None, "1"):
        ctx.debug_set("GOLAP_RESERVE", forced)
        for _ in range(3):
            ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH, out_count=cnt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH, out_count=cnt)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
        assert int(cnt.item()) == n
        v = ctx.verify_pairs(out, n, b, p).cpu()
        assert int(v[0]) == n
        res.append(dt)
    plan = ctx.join_plan(n)
    print(f"n={n:>11,d}  default {res[0]:7.3f} ms ({2*n/res[0]/1e6:6.1f} G rows/s)   forced {res[1]:7.3f} ms   bits {plan['bits1']},{plan['bits2']}", flush=True)
