"""Developer tool: which rows does a k_part2_ring variant lose?  Runs the same join with GOLAP_P2R=104 (hidden-load form) and 4 (plain
form) and reports the probe rows missing from the first, with their hash digits.  usage: python scripts/dbg_ring.py N [PASS_BITS]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context
n = int(sys.argv[1]); bits = sys.argv[2] if len(sys.argv) > 2 else None
torch.cuda.set_device(0)
ctx = Context(max_gpu_memory=0)
b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n); p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
res = {}
for v in ("4", "104"):
    ctx.debug_set("GOLAP_P2R", v)
    if bits:
        ctx.debug_set("GOLAP_PASS_BITS", bits); ctx.debug_set("GOLAP_RESERVE", "1")
    out = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
    pairs, cnt = ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH)
    torch.cuda.synchronize()
    m = int(cnt.item())
    got = torch.zeros(n, dtype=torch.int32, device="cuda")
    got[pairs[:m, 1].long()] += 1
    res[v] = (m, got)
    print("variant", v, "matches", m, "probe rows seen once", int((got == 1).sum()), "never", int((got == 0).sum()), "twice+", int((got > 1).sum()), flush=True)
miss = torch.nonzero(res["104"][1] == 0).flatten().cpu().numpy()
print("missing probe rows:", miss.size, miss[:40])
if miss.size:
    keys = p.cpu().numpy()[miss]
    h = np.array([oracle.hash_int64(int(k)) for k in keys[:4000]], dtype=np.uint64)
    print("keys", keys[:16]); print("hash & 255", (h & 255)[:32]); print("(hash>>8)&63", ((h >> 8) & 63)[:32]); print("(hash>>9)&511", ((h >> 9) & 511)[:32])
    d = np.diff(miss[:4000]); print("gaps between missing rows (first 40):", d[:40])
    # are the build partners of the missing rows alive?  (a missing build row loses all its probe partners)
    inv = torch.empty(n, dtype=torch.int64, device="cuda"); inv[b] = torch.arange(n, device="cuda")
    bp = inv[p[torch.from_numpy(miss).cuda()]].cpu().numpy()
    u, c = np.unique(bp, return_counts=True)
    print("distinct build partners of the missing probe rows:", u.size, "max probe rows per partner", int(c.max()))
