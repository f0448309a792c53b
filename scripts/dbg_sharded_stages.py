import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context
from gpu_olap_engine_amd.distributed import HipBackend, ShardedJoin
n = int(sys.argv[1])
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
ctx = Context(profiling=False, max_gpu_memory=0)
b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n); p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
sj = ShardedJoin(HipBackend(ctx)); be = sj.backend
out = torch.empty((n + 1024, 2), dtype=torch.int32, device="cuda")
for it in range(3):
    T = []
    def mark(name):
        torch.cuda.synchronize(); T.append((name, time.time()))
    mark("start")
    e1 = sj.extra_bits(n); mark("extra_bits")
    bk, br, boff = be.radix_partition(b, 0, 1, e1); mark("partition b")
    o = boff.cpu().numpy(); mark("offs cpu")
    xb = sj._exchange_start(bk, br, o, e1, n); mark("exchange_start b")
    pk, pr, poff = be.radix_partition(p, 0, 1, e1); mark("partition p")
    xp = sj._exchange_start(pk, pr, poff.cpu().numpy(), e1, n); mark("exchange_start p")
    be.join_begin(256 << e1, 1, 8 + e1, n, xb.n_recv, xp.n_recv); mark("join_begin")
    xb.wait(); be.join_side(0, xb.keys, xb.rids, xb.seg_off, xb.seg_len, xb.n_recv); mark("side b")
    xp.wait(); be.join_side(1, xp.keys, xp.rids, xp.seg_off, xp.seg_len, xp.n_recv); mark("side p")
    pairs, cnt = be.join_finish(out); mark("finish")
    print(f"iter {it} matches {int(cnt.item())}: " + "  ".join(f"{T[i][0]}={1e3*(T[i][1]-T[i-1][1]):.1f}" for i in range(1, len(T))), flush=True)
dist.destroy_process_group()
