"""Randomised checks of golap_join at 20M..70M rows per side -- the size range of the no-histogram and narrow-tuple flows --
without the CPU oracle (too slow there): the expected match count and the expected sum of probe row ids come from a
sort + searchsorted on the device (torch), and golap_verify_pairs checks that every emitted pair joins equal keys.
usage (GPU box): python scripts/fuzz_join_big.py SECONDS [SEED]      (developer tool, not product)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gpu_olap_engine_amd import _native as N
from gpu_olap_engine_amd.engine import Context


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    torch.cuda.set_device(0)
    ctx = Context(profiling=False, max_gpu_memory=0)
    t_end = time.time() + budget
    rnd = 0
    while time.time() < t_end:
        rnd += 1
        nb = int(rng.integers(20_000_000, 70_000_000))
        npr = int(rng.integers(1_000_000, 70_000_000))
        kind = str(rng.choice(["unique", "dups", "wide", "one_wide", "skew", "neg_pos", "halves", "many_halves", "halves_one_odd", "window_of_halves", "window_one_far"]))
        dom = nb if kind != "dups" else nb // 3
        bt = ctx.generate(N.GEN_BUILD_KEYS if kind != "dups" else N.GEN_SM64_MOD_I64, nb, modulus=dom, seed=int(rng.integers(1, 1000)))
        pt = ctx.generate(N.GEN_SM64_MOD_I64, npr, modulus=int(dom * rng.choice([1.0, 1.3])), seed=int(rng.integers(1, 1000)))
        if kind == "wide":
            bt += 1 << 41
            pt += 1 << 41
        elif kind == "one_wide":
            i = int(rng.integers(0, nb)); j = int(rng.integers(0, npr))
            bt[i] = -11; pt[j] = -11
        elif kind == "neg_pos":                                       # upper halves FFFFFFFF and 0: the dictionary form of the tuple pipeline
            bt -= dom // 2; pt -= dom // 2
        elif kind in ("halves", "halves_one_odd"):                    # 2 .. 4 upper halves; one_odd: plus a key whose upper half no sample can have seen
            k = int(rng.integers(2, 5))
            bt += (bt % k) << 40; pt += (pt % k) << 40
            if kind == "halves_one_odd":
                i = int(rng.integers(0, nb)); j = int(rng.integers(0, npr))
                bt[i] = (9 << 40) + 5; pt[j] = (9 << 40) + 5
        elif kind in ("window_of_halves", "window_one_far"):          # 5 .. 120 consecutive upper halves from a random (also negative) start: the offset form
            k = int(rng.integers(5, 121)); lo = int(rng.integers(-60, 1000))
            bt += ((bt % k) + lo) << 32; pt += ((pt % k) + lo) << 32
            if kind == "window_one_far":                              # ... and one key far outside the window, in a row no sample sees
                i = int(rng.integers(0, nb)); j = int(rng.integers(0, npr))
                bt[i] = (123_456 << 32) + 5; pt[j] = (123_456 << 32) + 5
        elif kind == "many_halves":                                   # the 64-bit-row pipeline
            bt += ((bt * 0x9E3779B1) & 0xFF) << 36; pt += ((pt * 0x9E3779B1) & 0xFF) << 36
        elif kind == "skew":                                          # a hot key: 200k extra build rows, 50 probe rows
            bt[:200_000] = 4242
            pt[:50] = 4242
        # expected count and probe-id sum on the device
        sb, _ = torch.sort(bt)
        mult = torch.searchsorted(sb, pt, right=True) - torch.searchsorted(sb, pt, right=False)
        m_exp = int(mult.sum().item())
        psum_exp = int((mult * torch.arange(npr, device="cuda", dtype=torch.int64)).sum().item()) & (2**64 - 1)
        del sb
        if m_exp > 200_000_000:
            continue
        out = torch.empty((max(m_exp, 1), 2), dtype=torch.int32, device="cuda")
        pairs, cnt = ctx.join(bt, pt, out_pairs=out, max_pairs=max(m_exp, 1), strategy=N.JOIN_RADIX_HASH)
        m = int(cnt.item())
        v = ctx.verify_pairs(pairs, min(m, m_exp), bt, pt).cpu().numpy().view(np.uint64)
        ok = m == m_exp and int(v[0]) == m_exp and int(v[2]) == psum_exp
        print(("ok   " if ok else "FAIL ") + f"seed={seed} round={rnd} nb={nb} np={npr} kind={kind} matches={m_exp} got={m} equal={int(v[0])}", flush=True)
        if not ok:
            sys.exit(1)
        del out, pairs, bt, pt, mult
    print(f"fuzz_big: {rnd} rounds, all as expected", flush=True)


if __name__ == "__main__":
    main()
