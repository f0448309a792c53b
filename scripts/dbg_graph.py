"""Is golap_join capturable into a HIP graph (launch-bound small joins, config C2)?  Replays and times it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
torch.cuda.set_device(0)
ctx = Context(profiling=False, max_gpu_memory=0)
b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n); p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
out = torch.empty((n, 2), dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
for strat, name in ((N.JOIN_RADIX_HASH, "radix"), (N.JOIN_BROADCAST_HASH, "broadcast")):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=strat, out_count=cnt)   # warm the workspace on this stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for _ in range(50):
            ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=strat, out_count=cnt)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50 * 1e3
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=strat, out_count=cnt)
        torch.cuda.synchronize()
        cnt.zero_(); out.zero_()
        g.replay(); torch.cuda.synchronize()
        m = int(cnt.item())
        v = ctx.verify_pairs(out, m, b, p).cpu().numpy()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        rep = (time.perf_counter() - t0) / 50 * 1e3
        print(f"{name}: n={n} eager {eager:.3f} ms/join, graph replay {rep:.3f} ms/join, matches {m}, key-equal pairs {int(v[0])}", flush=True)
    except Exception as e:
        print(f"{name}: capture failed: {e!r}", flush=True)
