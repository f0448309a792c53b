"""Does replaying a join as a HIP graph shorten the gaps between its ~43 launches?  100M x 100M, direct launches vs graph replays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
torch.cuda.set_device(0)
ctx = Context(max_gpu_memory=0)
b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n); p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
out = torch.empty((n, 2), dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
s = torch.cuda.Stream()
def direct(k):
    with torch.cuda.stream(s):
        for _ in range(k):
            ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH, out_count=cnt)
direct(3); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH, out_count=cnt)
torch.cuda.synchronize()
for rnd in range(3):
    t0 = time.perf_counter(); direct(30); torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(30): g.replay()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    assert int(cnt.item()) == n
    print(f"direct {1e3 * (t1 - t0) / 30:.3f} ms per join, graph replay {1e3 * (t2 - t1) / 30:.3f} ms", flush=True)
