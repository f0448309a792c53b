"""A/B over TWO knobs inside one process, interleaved rounds.  usage: python scripts/dev/ab2.py N VAR1 a,b,.. VAR2 x,y,.. [--wide]"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context
wide = "--wide" in sys.argv
if wide: sys.argv.remove("--wide")
n = int(sys.argv[1]); v1, a1, v2, a2 = sys.argv[2], sys.argv[3].split(","), sys.argv[4], sys.argv[5].split(",")
torch.cuda.set_device(0)
ctx = Context(profiling=True, max_gpu_memory=0)
b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n); p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
if wide:
    b += ((b * 0x9E3779B1) & 0xFFFF) << 32; p += ((p * 0x9E3779B1) & 0xFFFF) << 32
out = torch.empty((n, 2), dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
combos = list(itertools.product(a1, a2))
res = {c: {} for c in combos}
for rnd in range(6):
    for c in combos:
        ctx.debug_set(v1, None if c[0] == "-" else c[0]); ctx.debug_set(v2, None if c[1] == "-" else c[1])
        ctx.stats_reset()
        ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH, out_count=cnt)
        torch.cuda.synchronize()
        assert int(cnt.item()) == n
        if rnd:
            for k, s in ctx.stats().items():
                res[c].setdefault(k, []).append(s["total_ms"])
for c in combos:
    tot = sorted(sum(x) for x in zip(*res[c].values()))
    print(f"{v1}={c[0]} {v2}={c[1]}: kernels {tot[len(tot)//2]:.3f} ms | " + "  ".join(f"{k}={sorted(x)[len(x)//2]:.3f}" for k, x in res[c].items() if sorted(x)[len(x)//2] > 0.05))
