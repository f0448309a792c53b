"""Developer tool: BroadcastHashJoin against RadixHashJoin for small build sides and probe sides of growing size (the reference's rule,
join_kernel.rs:71-77, looks at the build side only: <= 1M rows -> broadcast).  Every run verified (matches, key equality).
usage: python scripts/dev/strategy_ab.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context

torch.cuda.set_device(0)
ctx = Context(max_gpu_memory=0)
for nb in (1_000, 30_000, 1_000_000):
    for np_ in (100_000, 1_000_000, 4_000_000, 16_000_000, 50_000_000):
        b = ctx.generate(N.GEN_BUILD_KEYS, nb, modulus=nb)
        p = ctx.generate(N.GEN_SM64_MOD_I64, np_, modulus=nb + nb // 10, seed=43)     # 1 in 11 probe rows matches nobody
        out = torch.empty((np_, 2), dtype=torch.int32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        exp = int((p < nb).sum().item())
        res = {}
        for name, strat in (("broadcast", N.JOIN_BROADCAST_HASH), ("radix", N.JOIN_RADIX_HASH)):
            for _ in range(3):
                ctx.join(b, p, out_pairs=out, max_pairs=np_, strategy=strat, out_count=cnt)
            torch.cuda.synchronize()
            it = 20 if np_ <= 4_000_000 else 8
            t0 = time.perf_counter()
            for _ in range(it):
                ctx.join(b, p, out_pairs=out, max_pairs=np_, strategy=strat, out_count=cnt)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / it * 1e3
            m = int(cnt.item())
            assert m == exp, (name, m, exp)
            v = ctx.verify_pairs(out, m, b, p)
            assert int(v[0].item()) == m, name
        print(f"n_build={nb:>9,d} n_probe={np_:>11,d}  broadcast {res['broadcast']:7.3f} ms   radix {res['radix']:7.3f} ms   radix/broadcast {res['radix'] / res['broadcast']:.2f}", flush=True)
