"""CPU tests: the C++ oracle against (1) the reference's own unit-test vectors, (2) the hand-evaluated
KATs of join_kernels.cuh, (3) the committed golden fixtures and (4) the independent numpy twin."""
import json
import os

import numpy as np
import pytest

import np_twin as T
import oracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KAT = json.load(open(os.path.join(G, "kat.json")))


def test_hash_kats():
    for k, v in KAT["hash_int64"].items():
        assert oracle.hash_int64(int(k)) == v, k
    for k, v in KAT["hash_int32"].items():
        assert oracle.hash_int32(int(k)) == v, k
    for k, (p0, p24) in KAT["partition_shift0_shift24"].items():
        h = oracle.hash_int64(int(k))
        assert (oracle.partition_id(h, 0), oracle.partition_id(h, 24)) == (p0, p24)


def test_hashes_against_an_independent_murmurhash3():
    """join_kernels.cuh:26-34 calls hash_int32 "MurmurHash3 32-bit finalizer": a PUBLISHED third-party algorithm (A. Appleby, MurmurHash3.cpp,
    fmix32).  MurmurHash3_x86_32 of the EMPTY message with seed s is exactly fmix32(s) (no block, no tail, h1 ^= len is a no-op), so any
    independent MurmurHash3_x86_32 pins the oracle's hash: here scikit-learn's Cython port of Appleby's reference code, plus the three
    empty-message vectors that circulate with every MurmurHash3 test suite.  hash_int64 is then the reference's own fold,
    hash_int32(lo ^ hi) (:36-41), and the partition id (hash >> shift) & 255 (:69)."""
    sk = pytest.importorskip("sklearn.utils")
    mm3 = lambda seed: int(sk.murmurhash3_32(b"", seed=int(seed), positive=True))
    for seed, h in ((0, 0x00000000), (1, 0x514E28B7), (0xFFFFFFFF, 0x81F16F39)):          # published MurmurHash3_x86_32("", seed)
        assert mm3(seed) == h and oracle.hash_int32(int(np.uint32(seed).astype(np.int32))) == h
    rng = np.random.default_rng(2024)
    k32 = np.concatenate([np.array([0, 1, 2, 42, 255, 256, 65535, 65536, 2**31 - 1, 2**31, 2**32 - 1], np.uint64),
                          rng.integers(0, 2**32, 20000, dtype=np.uint64), np.arange(1 << 12, dtype=np.uint64) << np.uint64(20)])
    exp = np.array([mm3(k) for k in k32], np.uint32)
    # keys in [0, 2^32): the upper half is 0, hash_int64(k) = hash_int32(k)
    assert np.array_equal(oracle.hash_int64_array(k32.astype(np.int64)), exp)
    assert np.array_equal(T.hash_int64(k32.astype(np.int64)), exp)
    for k, h in zip(k32[:200].tolist(), exp[:200].tolist()):
        assert oracle.hash_int32(int(np.uint32(k).astype(np.int32))) == h
    # full-range keys: the fold, then the same finalizer
    k64 = np.concatenate([np.array([-1, -2, 2**63 - 1, -2**63, 0x0123456789ABCDEF, 2**32, 2**32 + 1], np.int64), rng.integers(-2**63, 2**63 - 1, 20000, dtype=np.int64)])
    u = k64.view(np.uint64)
    fold = (u ^ (u >> np.uint64(32))) & np.uint64(0xFFFFFFFF)
    assert np.array_equal(oracle.hash_int64_array(k64), np.array([mm3(f) for f in fold], np.uint32))
    # and the 256-way partition id of the first pass on top of it (:69)
    _, _, off = oracle.radix_partition(k64)
    assert np.array_equal(np.diff(off.astype(np.int64)), np.bincount(np.array([mm3(f) & 255 for f in fold]), minlength=256))


def test_dense_partition_sizes():
    _, _, off = oracle.radix_partition(np.arange(1 << 20, dtype=np.int64))
    s = np.diff(off.astype(np.int64))
    d = KAT["dense_2pow20_partition_sizes"]
    assert (s.min(), s.max(), s.mean()) == (d["min"], d["max"], d["mean"])


def test_reference_unit_test_vectors():
    r = KAT["reference_unit_tests"]
    v = r["executor.rs:553-570 test_eval_filter_mask_gt"]
    assert oracle.eval_filter_mask(np.array(v["col"]), v["op"], v["literal"]).tolist() == v["mask"]
    L = oracle.lib()
    assert L.orc_choose_strategy(500_000) == 2 and L.orc_choose_strategy(5_000_000) == 0          # join_kernel.rs:85-87
    assert L.orc_choose_strategy(1_000_000) == 2 and L.orc_choose_strategy(1_000_001) == 0        # threshold <= 1M (:72)
    import ctypes as C
    g, b = C.c_uint32(), C.c_uint32()
    for n, (eg, eb) in r["filter_kernel.rs:86-95 launch_params"].items():
        L.orc_compute_launch_params(int(n), C.byref(g), C.byref(b))
        assert (g.value, b.value) == (eg, eb)
    ids = r["aggregate_kernel.rs:138-143 identity_values"]
    assert L.orc_agg_identity_value(0) == ids["Sum"] and L.orc_agg_identity_value(2) == ids["Min"] and L.orc_agg_identity_value(3) == ids["Max"]
    w = r["record_batch_convert.rs:211-217 int32_widens_to_i64"]
    a = np.array(w["int32"], np.int32)
    out = np.empty(a.size, np.int64)
    L.orc_widen_i32_to_i64(a.ctypes.data_as(C.POINTER(C.c_int32)), None, a.size, out.ctypes.data_as(C.POINTER(C.c_int64)))
    assert out.tobytes()[:8] == bytes(w["first_i64_le_bytes"]) and len(out.tobytes()) == 5 * 8


def test_hash_and_partition_match_twin():
    rng = np.random.default_rng(1)
    keys = np.concatenate([rng.integers(-2**63, 2**63 - 1, 100_000), np.arange(-1000, 1000)]).astype(np.int64)
    h = oracle.hash_int64_array(keys)
    assert np.array_equal(h, T.hash_int64(keys))
    ok, orr, off = oracle.radix_partition(keys)
    pid = T.partition_id(h)
    assert np.array_equal(np.diff(off.astype(np.int64)), np.bincount(pid, minlength=256))
    assert np.array_equal(keys[orr], ok)
    # stable inside a partition
    for p in (0, 17, 255):
        assert np.array_equal(orr[off[p]:off[p + 1]], np.nonzero(pid == p)[0])
    ok2, orr2, off2 = oracle.radix_partition(keys, radix_shift=24)
    assert np.array_equal(np.diff(off2.astype(np.int64)), np.bincount(T.partition_id(h, 24), minlength=256))


def test_golden_join_small():
    z = np.load(os.path.join(G, "join_small.npz"))
    got, m = oracle.hash_join(z["build"], z["probe"])
    assert m == z["pairs"].shape[0] and np.array_equal(got, z["pairs"])
    assert np.array_equal(oracle.hash_int64_array(z["build"]), z["hash_build"])
    # executor-semantics route (HashMap build with running batch offsets, executor.rs:238-253) gives the same pairs
    bl, pl = z["build_batch_lens"], z["probe_batch_lens"]
    bb = [z["build"][:int(bl[0])], z["build"][int(bl[0]):]]
    pb = [z["probe"][:int(pl[0])], z["probe"][int(pl[0]):]]
    got2, m2 = oracle.execute_hash_join(bb, pb)
    assert m2 == m and np.array_equal(got2, z["pairs"])
    assert oracle.execute_hash_join([], pb)[1] == 0 and oracle.execute_hash_join(bb, [])[1] == 0     # executor.rs:231-233
    mc, cs = oracle.execute_hash_join_count(z["build"], z["probe"])
    assert mc == m and cs == oracle.pairs_checksum(z["pairs"])
    mt, cst = oracle.radix_join_mt(z["build"], z["probe"], 4)
    assert (mt, cst) == (mc, cs)


@pytest.mark.parametrize("seed,nb,npr,span", [(1, 0, 10, 5), (2, 10, 0, 5), (3, 1, 1, 1), (4, 5000, 7000, 300), (5, 20000, 100, 10**12), (6, 300, 30000, 50)])
def test_join_random_vs_twin(seed, nb, npr, span):
    rng = np.random.default_rng(seed)
    b = rng.integers(-span, span + 1, nb).astype(np.int64)
    p = rng.integers(-span, span + 1, npr).astype(np.int64)
    got, m = oracle.hash_join(b, p, 10, 20)
    exp = T.hash_join(b, p, 10, 20)
    assert m == exp.shape[0] and np.array_equal(got, exp)


def test_join_max_pairs_truncates_but_counts():
    b = np.zeros(100, np.int64)
    p = np.zeros(50, np.int64)
    got, m = oracle.hash_join(b, p, max_pairs=123)
    assert m == 5000 and got.shape == (123, 2)


def test_golden_filter_and_aggregate():
    z = np.load(os.path.join(G, "filter_agg_small.npz"))
    names = {"=": "eq1", "==": "eq2", "!=": "ne1", "<>": "ne2", ">": "gt", ">=": "ge", "<": "lt", "<=": "le", "AND": "unknown"}
    for op, nm in names.items():
        assert np.array_equal(oracle.eval_filter_mask(z["col"], op, int(z["literal"])), z["mask_" + nm]), op
    assert z["mask_unknown"].sum() == 0                     # executor.rs:464
    s, kept = oracle.filter_sum(z["col"], z["val_i"], "<", int(z["literal"]))
    assert s == float(z["sum_i_lt"]) and kept == int(z["kept_lt"])
    idx = oracle.mask_to_indices(z["mask_lt"])
    assert np.array_equal(idx, np.nonzero(z["col"] < 13)[0]) and np.array_equal(oracle.take_8byte(z["val_f"], idx), z["val_f"][z["col"] < 13])
    a = oracle.execute_aggregate(z["val_i"])
    assert [a["sum"], a["count"], a["min"], a["max"], a["avg"]] == z["agg_i"].tolist()
    ks, st = oracle.group_aggregate(z["gk"], z["val_f"])
    assert np.array_equal(ks, np.arange(37))
    assert np.array_equal(st[:, 0], z["gsum"]) and np.array_equal(st[:, 1], z["gcnt"]) and np.array_equal(st[:, 2], z["gmin"]) and np.array_equal(st[:, 3], z["gmax"])
    assert np.array_equal(st[:, 4], z["gsum"] / z["gcnt"])


def test_aggregate_empty_conventions():
    a = oracle.execute_aggregate(np.empty(0, np.float64))   # executor.rs:308-311, :339-345
    assert a == {"sum": 0.0, "count": 0.0, "min": np.finfo(np.float64).max, "max": -np.finfo(np.float64).max, "avg": 0.0}


def test_golden_generators():
    z = np.load(os.path.join(G, "generators.npz"))
    i = np.arange(64, dtype=np.uint64)
    got = np.array([oracle.lib().orc_sm64(43, int(x)) for x in i], np.uint64)
    assert np.array_equal(got, z["sm64_seed43"])
    assert np.array_equal(oracle.gen_build_keys(0, 64, 1000003), z["build_keys_N1000003"])
    assert np.array_equal(oracle.gen_sm64_mod(7, 1000, 64, 2**63).astype(np.uint64), z["sm64_seed7_first1000"] % np.uint64(2**63))
    # C2-shaped property: unique build keys, every probe row matches exactly once
    n = 100_000
    b, p = oracle.gen_build_keys(0, n, n), oracle.gen_probe_keys(0, n, n)
    assert np.unique(b).size == n
    _, m = oracle.hash_join(b, p)
    assert m == n


def test_fnv1a_and_widen():
    import ctypes as C
    s = b"hello"
    h = oracle.lib().orc_fnv1a_hash((C.c_uint8 * len(s))(*s), len(s))
    assert h == 0xA430D84680AABD0B                           # FNV-1a 64 of "hello" (published test vector)
    assert oracle.lib().orc_fnv1a_hash(None, 0) == 14695981039346656037   # offset basis, record_batch_convert.rs:124


@pytest.mark.parametrize("jt", [0, 1, 2, 3])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_multi_column_null_aware_join_vs_brute_force(seed, jt):
    """orc_hash_join_multi (composite keys, SQL NULLs, outer joins) has no counterpart in the reference: pinned here by a pure-Python
    nested loop on small tables -- rows match when EVERY key column is equal and none is NULL; Left / Right / Full keep unmatched
    rows of the preserved side with the other id = 0xFFFFFFFF."""
    rng = np.random.default_rng(seed)
    nb, npr, nc = 57, 83, int(rng.integers(1, 4))
    bc = [rng.integers(-3, 4, nb).astype(np.int64) for _ in range(nc)]
    pc = [rng.integers(-3, 4, npr).astype(np.int64) for _ in range(nc)]
    bv = [None if rng.random() < 0.3 else (rng.random(nb) > 0.2).astype(np.uint8) for _ in range(nc)]
    pv = [None if rng.random() < 0.3 else (rng.random(npr) > 0.2).astype(np.uint8) for _ in range(nc)]
    ok_b = [all(bv[c] is None or bv[c][i] for c in range(nc)) for i in range(nb)]
    ok_p = [all(pv[c] is None or pv[c][j] for c in range(nc)) for j in range(npr)]
    pairs, hit_b, hit_p = [], [False] * nb, [False] * npr
    for j in range(npr):
        for i in range(nb):
            if ok_b[i] and ok_p[j] and all(bc[c][i] == pc[c][j] for c in range(nc)):
                pairs.append((i, j)); hit_b[i] = True; hit_p[j] = True
    NULL = 0xFFFFFFFF
    if jt & 1:
        pairs += [(i, NULL) for i in range(nb) if not hit_b[i]]
    if jt & 2:
        pairs += [(NULL, j) for j in range(npr) if not hit_p[j]]
    exp = T.canonical(np.array(pairs, np.uint32).reshape(-1, 2))
    got, m = oracle.hash_join_multi(bc, pc, jt, bv, pv)
    assert m == exp.shape[0] and np.array_equal(got, exp)


def test_filter_conjunction_is_the_and_of_the_reference_masks():
    """eval_filter_mask_and (an extension: the reference's mask has no AND, executor.rs:457-465) = AND of the per-comparison reference masks;
    checked against plain numpy comparisons, with a NULL operand making a comparison not true."""
    rng = np.random.default_rng(3)
    a = rng.integers(-20, 20, 5000).astype(np.int64)
    b = rng.integers(0, 9, 5000).astype(np.int64)
    va = (rng.random(5000) > 0.3).astype(np.uint8)
    m = oracle.eval_filter_mask_and([a, a, b], [">=", "<", "!="], [-5, 12, 4], [va, va, None])
    assert np.array_equal(m, (a >= -5) & (a < 12) & (b != 4) & va.astype(bool))
    assert not oracle.eval_filter_mask_and([a, b], ["<", "~"], [100, 1]).any()     # an unknown operator selects nothing (executor.rs:464)
