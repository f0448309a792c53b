import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, oracle
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context
ctx = Context(max_gpu_memory=0)
for n, G in ((1000, 4), (5000, 1024), (200000, 1024)):
    gk = oracle.gen_sm64_mod(7, 0, n, G, np.int32); gv = oracle.gen_sm64_mod(8, 0, n, 1024, np.float64)
    specs = [(N.AGG_SUM, 0), (N.AGG_COUNT, 0), (N.AGG_MIN, 0), (N.AGG_MAX, 0)]
    out = ctx.aggregate(specs, [torch.from_numpy(gv).cuda()], group_keys=torch.from_numpy(gk).cuda(), n_groups=G).cpu().numpy()
    exp = np.bincount(gk, weights=gv, minlength=G); ec = np.bincount(gk, minlength=G)
    print(n, G, "sum ok", np.array_equal(out[0], exp), "cnt ok", np.array_equal(out[1], ec))
    bad = np.nonzero(out[0] != exp)[0][:5]
    print("  bad idx", bad, out[0][bad], exp[bad], out[1][bad], ec[bad], "totals", out[0].sum(), exp.sum(), out[1].sum())
