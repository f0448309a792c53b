"""Upload rate of the operator layer vs host threads / transfer streams (bench --config OP's plan at 100M rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pyarrow as pa, torch
from gpu_olap_engine_amd import operators as O
from gpu_olap_engine_amd.engine import Context, default_config
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rng = np.random.default_rng(5)
cols = [pa.array(rng.integers(0, 1 << 40, n, dtype=np.int64)) for _ in range(3)]
batch = pa.RecordBatch.from_arrays(cols, names=["a", "b", "c"])
L = O.load_host()
for streams, threads in ((8, 4), (8, 8), (16, 12), (16, 16), (32, 24), (32, 32)):
    cfg = default_config(); cfg.num_streams = streams; cfg.max_gpu_memory = 0
    torch.cuda.set_device(0)
    with Context(config=cfg) as ctx:
        L.golap_host_set_transfer_threads(threads)
        best = 0
        for _ in range(4):
            pr = {}
            O.execute(O.GpuFilter(O.Batches([batch]), O.BinaryExpr(O.Column(0, "a"), "<", O.Literal("1000"))), ctx, profile=pr)
            best = max(best, pr["h2d_bytes"] / pr["h2d_ms"] / 1e6)
        print(f"streams {streams} threads {threads}: upload {best:.1f} GB/s", flush=True)
