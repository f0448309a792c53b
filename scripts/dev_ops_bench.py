"""Developer perf probe for GpuAggregate / GpuFilter at BASELINE sizes (C4: 1B-row GROUP BY int32 -> SUM(f64))."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
torch.cuda.set_device(0)
ctx = Context(profiling=True, max_gpu_memory=0)
def timed(name, fn, bytes_, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    print(f"{name:46s} {dt*1e3:9.3f} ms  {bytes_/dt/1e9:8.1f} GB/s  {n/dt/1e9:7.2f} Grows/s", flush=True)
for G in (1024, 1 << 20):
    k = ctx.generate(N.GEN_SM64_MOD_I32, n, modulus=G, seed=7)
    v = ctx.generate(N.GEN_SM64_MOD_F64, n, modulus=1024, seed=8)
    timed(f"C4 group-by int32 G={G} SUM(f64)", lambda: ctx.aggregate([(N.AGG_SUM, 0)], [v], group_keys=k, n_groups=G), 12 * n)
    if G == 1024:
        timed(f"   same, SUM+MIN+MAX+AVG", lambda: ctx.aggregate([(N.AGG_SUM, 0), (N.AGG_MIN, 0), (N.AGG_MAX, 0), (N.AGG_AVG, 0)], [v], group_keys=k, n_groups=G), 12 * n)
    out = ctx.aggregate([(N.AGG_SUM, 0)], [v], group_keys=k, n_groups=G)
    tot = float(out.sum().item()); exp = float(v.sum().item())
    print(f"   check: sum of group sums {tot} vs column sum {exp} -> {'OK' if tot == exp else 'MISMATCH'}", flush=True)
    del k, v
kk = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=1000, seed=1)
vv = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=1024, seed=2)
timed("filter_sum k<500 (C1 shape at n rows)", lambda: ctx.filter_sum(kk, vv, "<", 500), 16 * n)
timed("filter_mask", lambda: ctx.filter_mask(kk, "<", 500), 9 * n)
timed("filter_compact", lambda: ctx.filter_compact(kk, "<", 500), 8 * n + 2 * n)   # column read once (the write pass reads the 1-bit/row ballot map) + ~n/2 u32 indices
timed("ungrouped SUM/MIN/MAX (i64)", lambda: ctx.aggregate([(N.AGG_SUM, 0), (N.AGG_MIN, 0), (N.AGG_MAX, 0)], [vv]), 8 * n)
print({k: (round(v["total_ms"] / max(1, v["launches"]), 3), v["launches"]) for k, v in ctx.stats().items()})
