import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context
torch.cuda.set_device(0)
ctx = Context(max_gpu_memory=0, profiling=True)
for n in (100_000_000, 200_000_000, 400_000_000):
    b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n); p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
    out = torch.empty((n, 2), dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    for bits in (None, "9,6", "9,7", "8,8", "9,8"):
        ctx.debug_set("GOLAP_PASS_BITS", bits)
        for _ in range(2):
            ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH, out_count=cnt)
        torch.cuda.synchronize(); ctx.stats_reset(); t0 = time.perf_counter()
        for _ in range(5):
            ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH, out_count=cnt)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
        assert int(cnt.item()) == n
        st = ctx.stats()
        print(n, bits, round(dt, 3), {k: round(v["total_ms"] / 5, 3) for k, v in st.items() if v["total_ms"] / 5 > 0.05}, ctx.join_plan(n)["bits1"], ctx.join_plan(n)["bits2"], flush=True)
    del b, p, out
