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
if os.environ.get("CHECK_BIG"):   # demonstrates the RCCL > 4 GiB single-message problem (world size 1, 8 GB self send)
    bk, br, boff = ctx.radix_partition(b, rid_base=0, n_owners=1)
    sj.max_msg_bytes = 1 << 40
    x = sj._exchange_start(bk, br, boff.cpu().numpy(), 0, n); x.wait(); torch.cuda.synchronize()
    log("single-round exchange equal to source:", bool(torch.equal(x.keys[:n], bk[:n])), bool(torch.equal(x.rids[:n], br[:n])))
    sj.max_msg_bytes = 1 << 30
    x = sj._exchange_start(bk, br, boff.cpu().numpy(), 0, n); x.wait(); torch.cuda.synchronize()
    log("bounded-round exchange equal to source:", bool(torch.equal(x.keys[:n], bk[:n])), bool(torch.equal(x.rids[:n], br[:n])))
    del x, bk, br
out = torch.empty((n + 1024, 2), dtype=torch.int32, device="cuda")
for it in range(2):
    torch.cuda.synchronize(); t1 = time.time()
    pairs, cnt, nb, npr = sj.join(b, p, 0, 0, n, out_pairs=out)
    torch.cuda.synchronize()
    log(f"sharded join iter {it}: matches {int(cnt.item())} in {(time.time()-t1)*1e3:.2f} ms (extra_bits={sj.extra_bits(n)})")
log({k: round(v["total_ms"], 3) for k, v in ctx.stats().items()})
dist.destroy_process_group(); log("done")
