"""A/B of env-selected variants of the grouped aggregate inside ONE process.  usage: [AB_AGG_MM=1] python scripts/ab_agg.py N G ENVVAR v1 v2 ...
(AB_AGG_MM=1: SUM + MAX, i.e. the 28-byte table entries and the 2048-slot buckets)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context
n, G, var, vals = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4:]
ctx = Context(profiling=True, max_gpu_memory=0)
k = ctx.generate(N.GEN_SM64_MOD_I32, n, modulus=G, seed=7)
v = ctx.generate(N.GEN_SM64_MOD_F64, n, modulus=1024, seed=8)
res = {x: [] for x in vals}
for rnd in range(6):
    for x in vals:
        ctx.debug_set(var, x)
        ctx.stats_reset()
        ctx.aggregate([(N.AGG_SUM, 0), (N.AGG_MAX, 0)] if os.environ.get("AB_AGG_MM") else [(N.AGG_SUM, 0)], [v], group_keys=k, n_groups=G)
        torch.cuda.synchronize()
        if rnd:
            res[x].append({kk: s["total_ms"] for kk, s in ctx.stats().items()})
for x in vals:
    keys = res[x][0].keys()
    med = {kk: sorted(r[kk] for r in res[x])[len(res[x]) // 2] for kk in keys}
    print(f"{var}={x}: total {sum(med.values()):.3f} ms | " + "  ".join(f"{kk}={vv:.3f}" for kk, vv in med.items()))
