"""A/B of the streaming kernels' work-groups per CU (GOLAP_STREAM_WGS) at N rows: filter+sum, filter mask, filter compact, ungrouped aggregate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
vals = sys.argv[2:] or ["8", "4", "2", "1"]
ctx = Context(profiling=True, max_gpu_memory=0)
kk = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=1000, seed=1)
vv = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=1024, seed=2)
ops = {"filter_sum": lambda: ctx.filter_sum(kk, vv, "<", 500), "filter_mask": lambda: ctx.filter_mask(kk, "<", 500),
       "filter_compact": lambda: ctx.filter_compact(kk, "<", 500), "ungrouped": lambda: ctx.aggregate([(N.AGG_SUM, 0), (N.AGG_MAX, 0)], [vv])}
for x in vals:
    ctx.debug_set("GOLAP_STREAM_WGS", None if x == "default" else x)
    line = f"GOLAP_STREAM_WGS={x}:"
    for name, fn in ops.items():
        fn(); torch.cuda.synchronize(); ctx.stats_reset()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        st = ctx.stats()
        line += f"  {name} " + " + ".join(f"{v['total_ms'] / 5:.3f}" for v in st.values())
    print(line, flush=True)
