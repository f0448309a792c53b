import os, sys
sys.path.insert(0, "/root/repo")
import torch
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context
n = 1_000_000_000
torch.cuda.set_device(0)
ctx = Context(profiling=True, max_gpu_memory=0)
b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n); p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
out = torch.empty((n + (1 << 21), 2), dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
for rnd in range(3):
    for v in ("0", "8", "64", "1024"):
        ctx.debug_set("GOLAP_JOIN_SPREAD", v)
        ctx.stats_reset()
        ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH, out_count=cnt)
        torch.cuda.synchronize()
        st = ctx.stats()
        if rnd: print(f"spread={v}: join {st['join_build_probe_lds']['total_ms']:.3f} ms  p1 {st['radix_scatter_p1']['total_ms']:.3f}  p2 {st['radix_scatter_p2']['total_ms']:.3f}", flush=True)
