"""Key columns and the pair buffer allocated ONCE; only the context (i.e. the workspace slab) is torn down and set up again: does the
first pass's mode follow the workspace alone?  (developer tool)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context

n = 1_000_000_000
torch.cuda.set_device(0)
c0 = Context(profiling=0, max_gpu_memory=0)
b = c0.generate(N.GEN_BUILD_KEYS, n, modulus=n)
p = c0.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
c0.close()
pairs = torch.empty((n, 2), dtype=torch.int32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
hold = []
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    ctx = Context(profiling=1, max_gpu_memory=0)
    for _ in range(2):
        ctx.join(b, p, out_pairs=pairs, out_count=cnt)
    torch.cuda.synchronize()
    ctx.stats_reset()
    t0 = time.perf_counter()
    for _ in range(4):
        cnt.zero_()
        ctx.join(b, p, out_pairs=pairs, out_count=cnt)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 4 * 1e3
    st = ctx.stats()
    print(f"pid {os.getpid()} workspace {rnd}: {ms:.2f} ms  p1 {st['radix_scatter_p1']['total_ms'] / 4:.3f}  p2 {st['radix_scatter_p2']['total_ms'] / 4:.3f}  "
          f"join {st['join_build_probe_lds']['total_ms'] / 4:.3f}", flush=True)
    ctx.close()
    if os.environ.get("HOLD_GB"):   # keep a block between two workspaces so that the next one cannot land where the last one was
        hold.append(torch.empty(int(os.environ["HOLD_GB"]) << 30, dtype=torch.uint8, device="cuda"))
