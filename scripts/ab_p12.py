"""A/B of the narrow partition-pass variants (GOLAP_P1N x GOLAP_P2R) in ONE process, interleaved rounds; every run is
verified on the device (match count, key equality of every pair, probe-id sum).
usage: python scripts/ab_p12.py N "p1,p2[,nsub]" ...   e.g. 1000000000 0,0 118,0 0,104,8 118,104,8"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context
n = int(sys.argv[1]); combos = [tuple(c.split(",")) for c in sys.argv[2:]]
rounds = int(os.environ.get("ROUNDS", "4"))
torch.cuda.set_device(0)
ctx = Context(profiling=True, max_gpu_memory=0)
b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n); p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
out = torch.empty((n, 2), dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
res = {c: {} for c in combos}
want_sum = (n * (n - 1) // 2) & (2**64 - 1)
for rnd in range(rounds):
    for c in combos:
        ctx.debug_set("GOLAP_P1N", c[0]); ctx.debug_set("GOLAP_P2R", c[1])
        ctx.debug_set("GOLAP_NSUB", c[2] if len(c) > 2 else "1")
        ctx.stats_reset()
        out.zero_()
        ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH, out_count=cnt)
        torch.cuda.synchronize()
        m = int(cnt.item())
        v = ctx.verify_pairs(out, min(m, n), b, p).cpu()
        ok = m == n and int(v[0]) == n and (int(v[2]) & (2**64 - 1)) == want_sum
        if not ok and not any(x.startswith("9") for x in c[:2]):   # 9xx = timing-only diagnostic builds
            print(f"!! P1N,P2R={c}: WRONG RESULT matches={m} equal={int(v[0])} idsum_ok={(int(v[2]) & (2**64 - 1)) == want_sum}", flush=True)
        if rnd:
            for k, s in ctx.stats().items():
                res[c].setdefault(k, []).append(s["total_ms"])
for c in combos:
    tot = sorted(sum(x) for x in zip(*res[c].values()))
    med = lambda x: sorted(x)[len(x) // 2]
    print(f"P1N,P2R,NSUB={','.join(c):10s}: kernels median {tot[len(tot)//2]:7.3f} ms | " + "  ".join(f"{k}={med(x):.3f}" for k, x in res[c].items() if med(x) > 0.05), flush=True)
