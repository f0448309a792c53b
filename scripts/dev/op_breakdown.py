"""Developer tool: where the device time of bench.py's OP plan goes -- the same plan (filter -> join -> GROUP BY over host-resident Arrow
batches) on a context that times every launch (profiling = 1), per-kernel totals per golap_execute call next to the call's own profile.
usage: python scripts/dev/op_breakdown.py [ROWS]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pyarrow as pa
import torch

from gpu_olap_engine_amd import operators as O
from gpu_olap_engine_amd.engine import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
torch.cuda.set_device(0)
ctx = Context(profiling=1, max_gpu_memory=0)
n_c = max(1000, n // 100)
rng = np.random.default_rng(5)
o_cust = rng.integers(0, n_c + n_c // 10, n, dtype=np.int64)
o_amt = rng.integers(1, 1000, n).astype(np.float64)
o_flag = rng.integers(0, 10, n, dtype=np.int64)
c_id = rng.permutation(n_c).astype(np.int64)
c_seg = rng.integers(0, 1000, n_c).astype(np.int32)
cuts = [n * i // 4 for i in range(5)]
orders = [pa.RecordBatch.from_arrays([pa.array(o_cust[a:b]), pa.array(o_amt[a:b]), pa.array(o_flag[a:b])], names=["cust", "amt", "flag"]) for a, b in zip(cuts[:-1], cuts[1:])]
customers = [pa.RecordBatch.from_arrays([pa.array(c_id), pa.array(c_seg)], names=["id", "seg"])]


def plan():
    return O.GpuAggregate(O.GpuHashJoin(O.Batches(customers), O.GpuFilter(O.Batches(orders), O.BinaryExpr(O.Column(2, "flag"), "<", O.Literal("5"))),
                                        [O.Column(0, "id")], [O.Column(0, "cust")]),
                          [O.Column(1, "seg")], [O.Sum(O.Column(3, "amt")), O.Count(O.Column(3, "amt"))])


for _ in range(2):
    O.execute(plan(), ctx)
steps = 3
ctx.stats_reset()
t0 = time.perf_counter()
profs = []
for _ in range(steps):
    pr = {}
    O.execute(plan(), ctx, profile=pr)
    profs.append(pr)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
st = ctx.stats()
print(f"end to end {ms:.2f} ms per call;", {k: round(sum(p[k] for p in profs) / steps, 3) for k in profs[0]})
tot = 0.0
for k, v in sorted(st.items(), key=lambda kv: -kv[1]["total_ms"]):
    if v["launches"]:
        print(f"  {k:34s} {v['total_ms'] / steps:8.3f} ms per call  {v['launches'] / steps:6.1f} launches")
        tot += v["total_ms"] / steps
print(f"  kernels in all {tot:.3f} ms per call")
