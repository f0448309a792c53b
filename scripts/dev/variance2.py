"""Is the 3-5 % spread of the partition passes a property of the PROCESS (where its buffers landed)?  Two rounds per process: set up the
1B x 1B join, time it, print the device addresses of the key columns and the pairs, tear everything down, do it again.
usage: python scripts/dev/variance2.py [ROUNDS]   (developer tool)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context

n = 1_000_000_000
torch.cuda.set_device(0)
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    ctx = Context(profiling=1, max_gpu_memory=0)
    pre = os.environ.get("PREALLOC_GB")
    if pre:   # take and give back a large block first: does a coalesced free list change where the buffers land?
        t = torch.empty(int(pre) << 30, dtype=torch.uint8, device="cuda")
        del t
        torch.cuda.empty_cache()
    b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n)
    p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
    pairs = torch.empty((n, 2), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    for _ in range(2):
        ctx.join(b, p, out_pairs=pairs, out_count=cnt)
    torch.cuda.synchronize()
    ctx.stats_reset()
    t0 = time.perf_counter()
    for _ in range(6):
        cnt.zero_()
        ctx.join(b, p, out_pairs=pairs, out_count=cnt)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 6 * 1e3
    st = ctx.stats()
    print(f"pid {os.getpid()} round {rnd}: {ms:.2f} ms  p1 {st['radix_scatter_p1']['total_ms'] / 6:.3f}  p2 {st['radix_scatter_p2']['total_ms'] / 6:.3f}  "
          f"join {st['join_build_probe_lds']['total_ms'] / 6:.3f}  build @{b.data_ptr():#x} probe @{p.data_ptr():#x} pairs @{pairs.data_ptr():#x}", flush=True)
    del b, p, pairs, cnt
    ctx.close()
    torch.cuda.empty_cache()
