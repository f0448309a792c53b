"""What does a key column cost whose only odd upper half sits where no sample sees it?  (k_key_dict samples 8192 keys per side; the first
pass checks every key it reads.)  1B x 1B join: all keys below 2^32 / the one odd key in the build side's LAST row (the whole tuple attempt
is wasted, then the 64-bit pipeline runs) / in its first tile (the attempt stops at once).  usage: python scripts/time_rare_wide_key.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
torch.cuda.set_device(0)
ctx = Context(profiling=True, max_gpu_memory=0)
b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n); p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
out = torch.empty((n, 2), dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
def run(name):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for i in range(4):
        ctx.stats_reset()
        e0.record()
        ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH, out_count=cnt)
        e1.record(); torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    st = {k: round(v["total_ms"], 2) for k, v in ctx.stats().items() if v["total_ms"] > 0.2}
    print(f"{name}: {sorted(ts)[len(ts) // 2]:.2f} ms  matches {int(cnt.item())}  {st}", flush=True)
run("all keys below 2^32")
old = int(b[n - 1].item()); b[n - 1] = (7 << 40) + 3
run("one odd key in the build side's last row")
b[n - 1] = old; old = int(b[5].item()); b[5] = (7 << 40) + 3
run("one odd key in the build side's first tile")
