"""Developer perf probe (not part of the product): per-kernel device times of the radix join
at a few sizes / pass structures, plus a copy-kernel bandwidth reference.
usage: python scripts/dev_bench.py N [bits ...]   e.g. 100000000 default 8,7 8,4,3"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    variants = sys.argv[2:] or ["default"]
    torch.cuda.set_device(0)
    ctx = Context(profiling=True, max_gpu_memory=0)
    print(json.dumps(ctx.device_info()), flush=True)
    b = ctx.generate(N.GEN_BUILD_KEYS, n, modulus=n)
    p = ctx.generate(N.GEN_SM64_MOD_I64, n, modulus=n, seed=43)
    out = torch.empty((n, 2), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    # copy bandwidth reference (torch copy of 8 GB-ish)
    x = torch.empty(min(n, 500_000_000), dtype=torch.int64, device="cuda")
    for _ in range(2):
        x.copy_(b[: x.numel()])
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        x.copy_(b[: x.numel()])
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    print(f"copy_bw_GBps {2 * 8 * x.numel() / dt / 1e9:.1f}", flush=True)
    del x
    for v in variants:
        mode, _, bits = v.partition(":") if ":" in v else ("", "", v)
        if mode:
            ctx.debug_set("GOLAP_SCATTER", mode)
        else:
            ctx.debug_set("GOLAP_SCATTER", None)
        if bits == "default" or not bits:
            ctx.debug_set("GOLAP_PASS_BITS", None)
        else:
            ctx.debug_set("GOLAP_PASS_BITS", bits)
        for it in range(3):
            ctx.stats_reset()
            torch.cuda.synchronize()
            t0 = time.time()
            ctx.join(b, p, out_pairs=out, max_pairs=n, strategy=N.JOIN_RADIX_HASH, out_count=cnt)
            torch.cuda.synchronize()
            dt = time.time() - t0
            m = int(cnt.item())
        st = ctx.stats()
        tot = sum(s["total_ms"] for s in st.values())
        alg = 40 * 2 * n + 8 * m
        print(f"== n={n} bits={v} matches={m} wall_ms={dt*1e3:.2f} kernels_ms={tot:.2f} alg_GBps={alg/dt/1e9:.1f} rows_per_s={2*n/dt/1e9:.2f}G", flush=True)
        for k, s in st.items():
            gb = s["bytes"] / 1e9
            print(f"   {k:28s} n={s['launches']:3d} ms={s['total_ms']:9.3f} alg_GB={gb:8.3f} GBps={gb/max(s['total_ms'],1e-9)*1e3:9.1f}", flush=True)
        v4 = ctx.verify_pairs(out, min(m, n), b, p).cpu().numpy()
        print(f"   verify: equal={int(v4[0])} of {min(m,n)}", flush=True)


if __name__ == "__main__":
    main()
