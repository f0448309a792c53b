"""A/B of env-selected kernel variants inside ONE process, interleaved rounds (methodology rule 24).
usage: python scripts/ab_env.py N ENVVAR v1 v2 ... ; reports per-kernel ms (median of rounds) per variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context
wide = "--wide" in sys.argv      # keys with 2^16 different upper halves: the 64-bit-row pipeline
if wide: sys.argv.remove("--wide")
n = int(sys.argv[1]); var = sys.argv[2]; vals = sys.argv[3:]
torch.cuda.set_device(0)
ctx = Context(profiling=True, max_gpu_memory=0)
b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n); p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
if wide:
    b += ((b * 0x9E3779B1) & 0xFFFF) << 32; p += ((p * 0x9E3779B1) & 0xFFFF) << 32
out = torch.empty((n, 2), dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
res = {v: {} for v in vals}
for rnd in range(5):
    for v in vals:
        ctx.debug_set(var, v)
        ctx.stats_reset()
        ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH, out_count=cnt)
        torch.cuda.synchronize()
        assert int(cnt.item()) == n
        if rnd:
            for k, s in ctx.stats().items():
                res[v].setdefault(k, []).append(s["total_ms"])
for v in vals:
    tot = sorted(sum(x) for x in zip(*res[v].values()))
    print(f"{var}={v}: total kernels median {tot[len(tot)//2]:.3f} ms | " + "  ".join(f"{k}={sorted(x)[len(x)//2]:.3f}" for k, x in res[v].items() if sorted(x)[len(x)//2] > 0.05))
