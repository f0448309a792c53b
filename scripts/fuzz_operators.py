"""Randomised parity fuzz of the operator layer (libgolap_host.so: golap_plan_* / golap_execute over pyarrow RecordBatches) against
numpy / pure Python (developer tool, not product).
usage (GPU box): python scripts/fuzz_operators.py SECONDS [SEED]

Every round builds a random table -- 1 to 4 batches (some empty, some sliced so that validity bitmaps start in the middle of a byte),
columns of every numeric Arrow type with and without NULLs -- and one random plan over it:
  filter        <int64 column> <op> <literal>, or a conjunction of up to three such comparisons; every payload column comes back
                widened (int -> Int64, Float32 -> Float64), NULLs preserved, rows in order
  join          inner / left / right / full on one or two Int64 key columns with NULLs, payload columns of any numeric type, against
                the oracle's multi-column NULL-aware join
  group by      one to three key columns (dense, sparse, negative, full-range / hash-like, with NULLs), SUM / COUNT / MIN / MAX / AVG of
                an integer-valued Float64 or an Int64 column with NULLs, against a pure-Python GROUP BY over key tuples
  filter -> join -> group by   the three chained (device-resident between the operators)
Integer-valued doubles keep every summation order bit-identical."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import pyarrow as pa
import torch

import oracle
from gpu_olap_engine_amd import operators as O
from gpu_olap_engine_amd.engine import Context

TYPES = [("i8", pa.int8(), np.int8, -100, 100), ("u8", pa.uint8(), np.uint8, 0, 200), ("i16", pa.int16(), np.int16, -30000, 30000),
         ("u16", pa.uint16(), np.uint16, 0, 60000), ("i32", pa.int32(), np.int32, -2**31, 2**31 - 1), ("u32", pa.uint32(), np.uint32, 0, 2**32 - 1),
         ("i64", pa.int64(), np.int64, -2**62, 2**62), ("f32", pa.float32(), np.float32, -1000, 1000), ("f64", pa.float64(), np.float64, -1000, 1000)]
OPS = {"<": np.less, "<=": np.less_equal, ">": np.greater, ">=": np.greater_equal, "=": np.equal, "!=": np.not_equal}


def col_values(rng, n, spec):
    nm, t, dt, lo, hi = spec
    v = rng.integers(lo, hi, n, endpoint=False) if nm[0] in "iu" else rng.integers(lo, hi, n)
    return v.astype(dt)


def make_table(rng, n, cols, null_p):
    """cols: list of (name, arrow type, numpy values, has_nulls) -> (list of RecordBatch, dict name -> python list with None)"""
    py, arrays = {}, []
    for nm, t, v, nulls in cols:
        m = (rng.random(n) < null_p) if nulls else np.zeros(n, bool)
        lst = [None if mm else x.item() for x, mm in zip(v, m)]
        py[nm] = lst
        arrays.append(pa.array(lst, t))
    full = pa.RecordBatch.from_arrays(arrays, names=[c[0] for c in cols])
    cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, int(rng.integers(0, 4)))]))
    batches = [full.slice(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])]       # slices: bitmaps with bit offsets, shared buffers
    if rng.random() < 0.3:
        batches.insert(int(rng.integers(0, len(batches) + 1)), full.slice(0, 0))
    return batches, py


def widened(x):
    return None if x is None else (float(x) if isinstance(x, float) else int(x))


def table_rows(out):
    rows = []
    for b in out:
        cols = [b.column(i).to_pylist() for i in range(b.num_columns)]
        rows += list(zip(*cols)) if cols else []
    return rows


def round_filter(rng, ctx):
    n = int(rng.integers(1, 200_000))
    k0 = rng.integers(-50, 50, n).astype(np.int64)
    k1 = rng.integers(0, 1000, n).astype(np.int64)
    specs = [TYPES[i] for i in rng.choice(len(TYPES), int(rng.integers(1, 5)), replace=False)]
    cols = [("k0", pa.int64(), k0, rng.random() < 0.5), ("k1", pa.int64(), k1, rng.random() < 0.3)] + \
           [(s[0], s[1], col_values(rng, n, s), rng.random() < 0.5) for s in specs]
    batches, py = make_table(rng, n, cols, 0.15)
    leaves = []
    for _ in range(int(rng.integers(1, 4))):
        c = str(rng.choice(["k0", "k1"]))
        op = str(rng.choice(list(OPS)))
        lit = int(rng.integers(-60, 60)) if c == "k0" else int(rng.integers(-10, 1010))
        leaves.append((c, op, lit))
    pred = None
    for c, op, lit in leaves:
        e = O.BinaryExpr(O.Column(0, c), op, O.Literal(str(lit))) if rng.random() < 0.7 else \
            O.BinaryExpr(O.Literal(str(lit)), {"<": ">", "<=": ">=", ">": "<", ">=": "<=", "=": "=", "!=": "!="}[op], O.Column(0, c))
        pred = e if pred is None else O.BinaryExpr(pred, "AND", e)
    out = O.execute(O.GpuFilter(O.Batches(batches), pred), ctx)
    keep = np.ones(n, bool)
    for c, op, lit in leaves:
        v = np.array([0 if x is None else x for x in py[c]], dtype=np.int64)
        keep &= OPS[op](v, lit) & np.array([x is not None for x in py[c]])
    names = [c[0] for c in cols]
    exp = [tuple(widened(py[nm][i]) for nm in names) for i in np.nonzero(keep)[0]]
    got = table_rows(out)
    assert got == exp, f"filter {leaves}: {len(got)} rows, expected {len(exp)}"
    return f"filter n={n} leaves={len(leaves)} kept={len(exp)}"


def round_join(rng, ctx):
    nb, np_ = int(rng.integers(1, 60_000)), int(rng.integers(1, 80_000))
    nk = int(rng.integers(1, 3))
    dom = int(rng.integers(5, 5000))
    wide = rng.random() < 0.3
    pool = rng.integers(-2**63, 2**63 - 1, dom) if wide else np.arange(dom) - dom // 3
    bk = [pool[rng.integers(0, dom, nb)].astype(np.int64) for _ in range(nk)]
    pk = [pool[rng.integers(0, dom, np_)].astype(np.int64) for _ in range(nk)]
    if nk == 2:    # (keep the pair space small enough for matches)
        bk[1], pk[1] = bk[1] % 7, pk[1] % 7
    sb, sp = TYPES[int(rng.integers(0, len(TYPES)))], TYPES[int(rng.integers(0, len(TYPES)))]
    bcols = [(f"bk{i}", pa.int64(), bk[i], rng.random() < 0.5) for i in range(nk)] + [("bpay", sb[1], col_values(rng, nb, sb), rng.random() < 0.5)]
    pcols = [(f"pk{i}", pa.int64(), pk[i], rng.random() < 0.5) for i in range(nk)] + [("ppay", sp[1], col_values(rng, np_, sp), rng.random() < 0.5)]
    bb, bpy = make_table(rng, nb, bcols, 0.1)
    pb, ppy = make_table(rng, np_, pcols, 0.1)
    jt = int(rng.integers(0, 4))
    plan = O.GpuHashJoin(O.Batches(bb), O.Batches(pb), [O.Column(0, f"bk{i}") for i in range(nk)], [O.Column(0, f"pk{i}") for i in range(nk)],
                         join_type=["Inner", "Left", "Right", "Full"][jt])
    out = O.execute(plan, ctx)
    val = lambda py, nm: np.array([0 if x is None else x for x in py[nm]], dtype=np.int64)
    vld = lambda py, nm: np.array([x is not None for x in py[nm]], dtype=np.uint8)
    pairs, m = oracle.hash_join_multi([val(bpy, f"bk{i}") for i in range(nk)], [val(ppy, f"pk{i}") for i in range(nk)], jt,
                                      [vld(bpy, f"bk{i}") for i in range(nk)], [vld(ppy, f"pk{i}") for i in range(nk)])
    bn, pn = [c[0] for c in bcols], [c[0] for c in pcols]
    exp = [tuple(None if b == 0xFFFFFFFF else widened(bpy[nm][int(b)]) for nm in bn) + tuple(None if p == 0xFFFFFFFF else widened(ppy[nm][int(p)]) for nm in pn)
           for b, p in pairs]
    got = table_rows(out)
    assert got == exp, f"join jt={jt} nk={nk}: {len(got)} rows, expected {len(exp)}"
    return f"join jt={jt} nk={nk} wide={wide} nb={nb} np={np_} pairs={m}"


def py_group_by(keys, vals, funcs):
    groups = {}
    for i in range(len(vals)):
        k = tuple(c[i] for c in keys)
        g = groups.setdefault(k, [0.0, 0, np.inf, -np.inf])
        if vals[i] is not None:
            x = float(vals[i])
            g[0] += x; g[1] += 1; g[2] = min(g[2], x); g[3] = max(g[3], x)
    order = sorted(groups, key=lambda t: tuple((x is None, 0 if x is None else x) for x in t))
    ident = {"min": np.finfo(np.float64).max, "max": -np.finfo(np.float64).max}
    rows = []
    for k in order:
        s, c, mn, mx = groups[k]
        agg = {"sum": s, "count": float(c), "min": mn if c else ident["min"], "max": mx if c else ident["max"], "avg": (s / c if c else 0.0)}
        rows.append(tuple(k) + tuple(agg[f] for f in funcs))
    return rows


def round_group_by(rng, ctx):
    n = int(rng.integers(1, 150_000))
    nk = int(rng.integers(1, 4))
    shape = str(rng.choice(["dense", "sparse", "negative", "wide"]))
    keys = []
    for c in range(nk):
        d = int(rng.integers(2, 400 if nk == 1 else 12))
        base = rng.integers(0, d, n)
        if shape == "sparse":
            base = base * 1_000_003 - 17
        elif shape == "negative":
            base = base - d // 2
        elif shape == "wide":
            base = rng.integers(-2**63, 2**63 - 1, d)[base]
        kt = pa.int32() if (shape in ("dense", "negative") and rng.random() < 0.4) else pa.int64()
        keys.append((f"g{c}", kt, base.astype(np.int32 if kt == pa.int32() else np.int64), rng.random() < 0.4))
    vt = str(rng.choice(["f64", "i64"]))
    v = rng.integers(-500, 500, n)
    vcol = ("v", pa.float64() if vt == "f64" else pa.int64(), v.astype(np.float64 if vt == "f64" else np.int64), rng.random() < 0.5)
    batches, py = make_table(rng, n, keys + [vcol], 0.1)
    funcs = [str(f) for f in rng.choice(["sum", "count", "min", "max", "avg"], int(rng.integers(1, 4)), replace=False)]
    mk = {"sum": O.Sum, "count": O.Count, "min": O.Min, "max": O.Max, "avg": O.Avg}
    plan = O.GpuAggregate(O.Batches(batches), [O.Column(0, k[0]) for k in keys], [mk[f](O.Column(0, "v")) for f in funcs])
    out = O.execute(plan, ctx)
    exp = py_group_by([py[k[0]] for k in keys], py["v"], funcs)
    got = table_rows(out)
    assert got == exp, f"group by {shape} nk={nk} {funcs}: {len(got)} rows, expected {len(exp)}; first diff {next(((a, b) for a, b in zip(got, exp) if a != b), None)}"
    return f"group-by {shape} nk={nk} n={n} groups={len(exp)} {funcs}"


def round_chain(rng, ctx):
    n_o, n_c = int(rng.integers(100, 300_000)), int(rng.integers(10, 3000))
    o_cust = rng.integers(0, n_c + n_c // 5 + 1, n_o).astype(np.int64)
    o_amt = rng.integers(1, 1000, n_o).astype(np.float64)
    o_flag = rng.integers(0, 10, n_o).astype(np.int64)
    c_id = rng.permutation(n_c).astype(np.int64)
    c_seg = rng.integers(0, 9, n_c).astype(np.int32)
    ob, opy = make_table(rng, n_o, [("cust", pa.int64(), o_cust, True), ("amt", pa.float64(), o_amt, True), ("flag", pa.int64(), o_flag, False)], 0.05)
    cb, cpy = make_table(rng, n_c, [("id", pa.int64(), c_id, False), ("seg", pa.int32(), c_seg, True)], 0.05)
    thr = int(rng.integers(1, 10))
    plan = O.GpuAggregate(O.GpuHashJoin(O.Batches(cb), O.GpuFilter(O.Batches(ob), O.BinaryExpr(O.Column(0, "flag"), "<", O.Literal(str(thr)))),
                                        [O.Column(0, "id")], [O.Column(0, "cust")]),
                          [O.Column(0, "seg")], [O.Sum(O.Column(0, "amt")), O.Count(O.Column(0, "amt"))])
    out = O.execute(plan, ctx)
    seg_of = {int(i): s for i, s in zip(c_id, cpy["seg"])}
    ks, vs = [], []
    for cu, am, fl in zip(opy["cust"], opy["amt"], opy["flag"]):
        if fl < thr and cu is not None and cu in seg_of:
            ks.append(seg_of[cu]); vs.append(am)
    exp = py_group_by([ks], vs, ["sum", "count"]) if ks else []
    got = table_rows(out)
    assert got == exp, f"chain: {got[:5]} vs {exp[:5]}"
    return f"chain n_o={n_o} n_c={n_c} joined={len(ks)} groups={len(exp)}"


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    torch.cuda.set_device(0)
    ctx = Context(max_gpu_memory=0)
    kinds = [round_filter, round_join, round_group_by, round_chain]
    counts = {k.__name__: 0 for k in kinds}
    t_end = time.time() + budget
    rnd = 0
    while time.time() < t_end:
        rnd += 1
        k = kinds[int(rng.integers(0, len(kinds)))]
        O.load_host().golap_host_set_pipelining(int(rng.integers(0, 2)))   # a filter's batches: one transfer phase (the default) or pipelined
        try:
            what = k(rng, ctx)
        except Exception as e:
            print(f"round {rnd} seed {seed} {k.__name__} FAILED: {e!r}", flush=True)
            raise
        counts[k.__name__] += 1
        if rnd % 25 == 0:
            print(f"round {rnd}: {what}", flush=True)
    print(f"fuzz_operators: {rnd} rounds, all equal: {counts}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
