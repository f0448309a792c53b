"""A/B of GOLAP_STREAM_WGS for golap_widen_to_8byte (int32 -> int64) and golap_take_8byte at N rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
ctx = Context(profiling=True, max_gpu_memory=0)
k = ctx.generate(N.GEN_SM64_MOD_I32, n, modulus=1000, seed=1)
for x in sys.argv[2:] or ["default"]:
    ctx.debug_set("GOLAP_STREAM_WGS", None if x == "default" else x)
    ctx.widen(k); torch.cuda.synchronize(); ctx.stats_reset()
    for _ in range(5): o = ctx.widen(k)
    torch.cuda.synchronize()
    print("GOLAP_STREAM_WGS", x, {kk: round(v["total_ms"] / 5, 3) for kk, v in ctx.stats().items()}, flush=True)
    del o
