import faulthandler, os, sys, time
faulthandler.dump_traceback_later(int(os.environ.get("DUMP_AFTER", "60")), exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context
from gpu_olap_engine_amd.distributed import HipBackend, ShardedJoin
n = int(sys.argv[1])
t0 = time.time()
def log(*a):
    print(f"[{time.time()-t0:7.2f}]", *a, flush=True)
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
log("pg up")
ctx = Context(profiling=True, max_gpu_memory=0)
b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n); p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
torch.cuda.synchronize(); log("generated")
sj = ShardedJoin(HipBackend(ctx))
bk, br, boff = ctx.radix_partition(b, rid_base=0, n_owners=1); torch.cuda.synchronize(); log("partition b done")
pk, pr, poff = ctx.radix_partition(p, rid_base=0, n_owners=1); torch.cuda.synchronize(); log("partition p done")
offs = torch.stack([boff, poff]).cpu().numpy(); log("offs on host", offs[0][:3], offs[0][-1])
if os.environ.get("CHECK_BIG"):
    sj.max_msg_bytes = 1 << 40
    x = sj._exchange_side(bk, br, offs[0]); torch.cuda.synchronize()
    log("single-round exchange equal to source:", bool(torch.equal(x[0][:n], bk[:n])), bool(torch.equal(x[1][:n], br[:n])))
    sj.max_msg_bytes = 1 << 30
x = sj._exchange_side(bk, br, offs[0]); torch.cuda.synchronize(); log("exchange b done", x[2], "equal:", bool(torch.equal(x[0][:n], bk[:n])))
y = sj._exchange_side(pk, pr, offs[1]); torch.cuda.synchronize(); log("exchange p done", y[2])
out = torch.empty((n + 1024, 2), dtype=torch.int32, device="cuda")
pairs, cnt = ctx.join_partitioned(x[0][:x[2]], x[1][:x[2]], x[3], x[4], y[0][:y[2]], y[1][:y[2]], y[3], y[4], 256, 1, out, total_build_rows_hint=n)
torch.cuda.synchronize(); log("join_partitioned done", int(cnt.item()))
log({k: round(v["total_ms"], 3) for k, v in ctx.stats().items()})
dist.destroy_process_group(); log("done")
