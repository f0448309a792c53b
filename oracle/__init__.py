"""ctypes loader for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (gpu_olap_engine_amd) never does.
See golap_oracle.cpp for the pinning status and the reference file:line cites.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgolap_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "golap_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        i64p, i32p, f64p, f32p = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_float)
        sig = {
            "orc_hash_int32": (C.c_uint32, [C.c_int32]),
            "orc_hash_int64": (C.c_uint32, [C.c_int64]),
            "orc_partition_id": (C.c_uint32, [C.c_uint32, C.c_int]),
            "orc_hash_int64_array": (None, [i64p, C.c_uint64, u32p]),
            "orc_radix_partition": (None, [i64p, u32p, C.c_uint64, C.c_int, i64p, u32p, u64p]),
            "orc_hash_join_i64": (C.c_uint64, [i64p, C.c_uint64, i64p, C.c_uint64, C.c_uint32, C.c_uint32, u32p, C.c_uint64]),
            "orc_hash_join_outer_i64": (C.c_uint64, [i64p, C.c_uint64, i64p, C.c_uint64, C.c_int, u32p, C.c_uint64]),
            "orc_execute_hash_join": (C.c_uint64, [i64p, u64p, C.c_uint32, i64p, u64p, C.c_uint32, u32p, C.c_uint64]),
            "orc_pair_mix": (C.c_uint64, [C.c_uint32, C.c_uint32]),
            "orc_pairs_checksum": (C.c_uint64, [u32p, C.c_uint64]),
            "orc_execute_hash_join_count": (C.c_uint64, [i64p, C.c_uint64, i64p, C.c_uint64, u64p]),
            "orc_radix_join_mt": (C.c_uint64, [i64p, C.c_uint64, i64p, C.c_uint64, C.c_uint32, u64p]),
            "orc_parse_cmp_op": (C.c_int, [C.c_char_p]),
            "orc_eval_filter_mask_i64": (None, [i64p, C.c_uint64, C.c_char_p, C.c_int64, u8p]),
            "orc_mask_to_indices": (C.c_uint64, [u8p, C.c_uint64, u32p]),
            "orc_take_8byte": (None, [u64p, u32p, C.c_uint64, u64p]),
            "orc_execute_aggregate_f64": (None, [f64p, C.c_uint64, f64p]),
            "orc_execute_aggregate_i64": (None, [i64p, C.c_uint64, f64p]),
            "orc_filter_sum_i64": (C.c_double, [i64p, i64p, C.c_uint64, C.c_char_p, C.c_int64, u64p]),
            "orc_agg_identity_value": (C.c_int64, [C.c_int]),
            "orc_group_aggregate": (C.c_uint64, [i64p, f64p, C.c_uint64, i64p, f64p, C.c_uint64]),
            "orc_hash_aggregate_kernel_i64": (None, [u32p, i64p, C.c_uint64, i64p, C.c_uint32]),
            "orc_choose_strategy": (C.c_int, [C.c_uint64]),
            "orc_compute_launch_params": (None, [C.c_uint64, u32p, u32p]),
            "orc_widen_i32_to_i64": (None, [i32p, u8p, C.c_uint64, i64p]),
            "orc_widen_f32_to_f64": (None, [f32p, u8p, C.c_uint64, f64p]),
            "orc_fnv1a_hash": (C.c_uint64, [u8p, C.c_uint64]),
            "orc_sm64": (C.c_uint64, [C.c_uint64, C.c_uint64]),
            "orc_gen_build_keys": (None, [i64p, C.c_uint64, C.c_uint64, C.c_uint64]),
            "orc_gen_probe_keys": (None, [i64p, C.c_uint64, C.c_uint64, C.c_uint64]),
            "orc_gen_sm64_mod": (None, [i64p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]),
            "orc_gen_sm64_mod_i32": (None, [i32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]),
            "orc_gen_sm64_mod_f64": (None, [f64p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]),
            "orc_now_seconds": (C.c_double, []),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ---- numpy-friendly wrappers ------------------------------------------------
def hash_int32(k):
    return int(lib().orc_hash_int32(int(np.int32(k))))


def hash_int64(k):
    return int(lib().orc_hash_int64(int(np.int64(k))))


def partition_id(h, shift):
    return int(lib().orc_partition_id(h, shift))


def hash_int64_array(keys):
    keys = _c(keys, np.int64)
    out = np.empty(keys.size, np.uint32)
    lib().orc_hash_int64_array(_p(keys, C.c_int64), keys.size, _p(out, C.c_uint32))
    return out


def radix_partition(keys, row_ids=None, radix_shift=0):
    keys = _c(keys, np.int64)
    rids = None if row_ids is None else _c(row_ids, np.uint32)
    ok = np.empty_like(keys)
    orr = np.empty(keys.size, np.uint32)
    off = np.zeros(257, np.uint64)
    lib().orc_radix_partition(_p(keys, C.c_int64), _p(rids, C.c_uint32), keys.size, radix_shift,
                              _p(ok, C.c_int64), _p(orr, C.c_uint32), _p(off, C.c_uint64))
    return ok, orr, off


def hash_join(build_keys, probe_keys, build_rid_base=0, probe_rid_base=0, max_pairs=None):
    """Canonical (probe_rid, build_rid)-sorted pairs as an (M,2) uint32 array [build_rid, probe_rid]."""
    b, p = _c(build_keys, np.int64), _c(probe_keys, np.int64)
    m = lib().orc_hash_join_i64(_p(b, C.c_int64), b.size, _p(p, C.c_int64), p.size, build_rid_base, probe_rid_base, None, 0)
    cap = m if max_pairs is None else min(m, max_pairs)
    out = np.empty((cap, 2), np.uint32)
    lib().orc_hash_join_i64(_p(b, C.c_int64), b.size, _p(p, C.c_int64), p.size, build_rid_base, probe_rid_base,
                            _p(out, C.c_uint32), cap)
    return out, m


def hash_join_outer(build_keys, probe_keys, join_type):
    """join_type 0 Inner / 1 Left / 2 Right / 3 Full -> canonical (M,2) uint32 [build_rid, probe_rid], NULL = 0xFFFFFFFF."""
    b, p = _c(build_keys, np.int64), _c(probe_keys, np.int64)
    m = lib().orc_hash_join_outer_i64(_p(b, C.c_int64), b.size, _p(p, C.c_int64), p.size, join_type, None, 0)
    out = np.empty((m, 2), np.uint32)
    lib().orc_hash_join_outer_i64(_p(b, C.c_int64), b.size, _p(p, C.c_int64), p.size, join_type, _p(out, C.c_uint32), m)
    return out, m


def hash_join_multi(build_cols, probe_cols, join_type=0, build_valid=None, probe_valid=None):
    """Equi-join on several Int64 key columns with NULLs (SQL semantics: NULL keys match nothing; outer joins keep such rows as
    unmatched).  *_cols: lists of int64 arrays; *_valid: None or lists of (None | uint8 array, 0 = NULL).
    -> canonical (M,2) uint32 [build_rid, probe_rid], NULL = 0xFFFFFFFF."""
    nc = len(build_cols)
    assert nc == len(probe_cols) and nc >= 1
    b = [_c(x, np.int64) for x in build_cols]
    p = [_c(x, np.int64) for x in probe_cols]
    nb, npr = b[0].size, p[0].size

    def vlist(v, n):
        if v is None:
            return None, [None] * nc
        keep = [None if x is None else _c(x, np.uint8) for x in v]
        arr = (C.POINTER(C.c_uint8) * nc)(*[C.cast(None, C.POINTER(C.c_uint8)) if x is None else _p(x, C.c_uint8) for x in keep])
        return arr, keep
    bv, _kb = vlist(build_valid, nb)
    pv, _kp = vlist(probe_valid, npr)
    bc = (C.POINTER(C.c_int64) * nc)(*[_p(x, C.c_int64) for x in b])
    pc = (C.POINTER(C.c_int64) * nc)(*[_p(x, C.c_int64) for x in p])
    f = lib().orc_hash_join_multi
    f.restype = C.c_uint64
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_uint64]
    m = f(bc, bv, nb, pc, pv, npr, nc, join_type, None, 0)
    out = np.empty((m, 2), np.uint32)
    f(bc, bv, nb, pc, pv, npr, nc, join_type, out.ctypes.data if m else None, m)
    return out, m


def execute_hash_join(left_batches, right_batches):
    """executor.rs:220-265 semantics over lists of int64 key arrays (one per RecordBatch)."""
    if len(left_batches) == 0 or len(right_batches) == 0:
        return np.empty((0, 2), np.uint32), 0
    l = _c(np.concatenate(left_batches) if left_batches else np.empty(0), np.int64)
    r = _c(np.concatenate(right_batches) if right_batches else np.empty(0), np.int64)
    ll = np.array([len(x) for x in left_batches], np.uint64)
    rl = np.array([len(x) for x in right_batches], np.uint64)
    m = lib().orc_execute_hash_join(_p(l, C.c_int64), _p(ll, C.c_uint64), ll.size, _p(r, C.c_int64), _p(rl, C.c_uint64), rl.size, None, 0)
    out = np.empty((m, 2), np.uint32)
    lib().orc_execute_hash_join(_p(l, C.c_int64), _p(ll, C.c_uint64), ll.size, _p(r, C.c_int64), _p(rl, C.c_uint64), rl.size,
                                _p(out, C.c_uint32), m)
    return out, m


def pairs_checksum(pairs):
    pairs = _c(pairs, np.uint32).reshape(-1)
    return int(lib().orc_pairs_checksum(_p(pairs, C.c_uint32), pairs.size // 2))


def execute_hash_join_count(left, right):
    l, r = _c(left, np.int64), _c(right, np.int64)
    cs = C.c_uint64(0)
    m = lib().orc_execute_hash_join_count(_p(l, C.c_int64), l.size, _p(r, C.c_int64), r.size, C.byref(cs))
    return int(m), int(cs.value)


def radix_join_mt(build, probe, n_threads):
    b, p = _c(build, np.int64), _c(probe, np.int64)
    cs = C.c_uint64(0)
    m = lib().orc_radix_join_mt(_p(b, C.c_int64), b.size, _p(p, C.c_int64), p.size, n_threads, C.byref(cs))
    return int(m), int(cs.value)


def fnv1a_hash(data: bytes):
    """record_batch_convert.rs:123-130 (FNV-1a 64) -> python int (u64)"""
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data if data else b"\0")
    return int(lib().orc_fnv1a_hash(buf, len(data)))


def eval_filter_mask(col, op, literal):
    col = _c(col, np.int64)
    mask = np.empty(col.size, np.uint8)
    lib().orc_eval_filter_mask_i64(_p(col, C.c_int64), col.size, op.encode(), int(literal), _p(mask, C.c_uint8))
    return mask


def eval_filter_mask_and(cols, ops, literals, valids=None):
    """Conjunction with SQL semantics (an EXTENSION: the reference's mask has no AND, executor.rs:457-465): the elementwise AND of the
    reference's own per-comparison masks (eval_filter_mask), a NULL operand making its comparison not true."""
    m = np.ones(len(cols[0]), dtype=bool)
    for k, (c, o, l) in enumerate(zip(cols, ops, literals)):
        m &= eval_filter_mask(c, o, l).astype(bool)
        if valids is not None and valids[k] is not None:
            m &= np.asarray(valids[k]).astype(bool)
    return m


def mask_to_indices(mask):
    mask = _c(mask, np.uint8)
    idx = np.empty(mask.size, np.uint32)
    c = lib().orc_mask_to_indices(_p(mask, C.c_uint8), mask.size, _p(idx, C.c_uint32))
    return idx[:c].copy()


def take_8byte(col, indices):
    col = np.ascontiguousarray(col)
    assert col.dtype.itemsize == 8
    idx = _c(indices, np.uint32)
    out = np.empty(idx.size, col.dtype)
    lib().orc_take_8byte(col.ctypes.data_as(C.POINTER(C.c_uint64)), _p(idx, C.c_uint32), idx.size,
                         out.ctypes.data_as(C.POINTER(C.c_uint64)))
    return out


def execute_aggregate(values):
    """-> dict(sum,count,min,max,avg) with executor.rs:297-355 arithmetic."""
    out = np.zeros(5, np.float64)
    v = np.ascontiguousarray(values)
    if v.dtype == np.float64:
        lib().orc_execute_aggregate_f64(_p(v, C.c_double), v.size, _p(out, C.c_double))
    else:
        v = _c(v, np.int64)
        lib().orc_execute_aggregate_i64(_p(v, C.c_int64), v.size, _p(out, C.c_double))
    return dict(zip(("sum", "count", "min", "max", "avg"), out.tolist()))


def filter_sum(k, v, op, literal):
    k, v = _c(k, np.int64), _c(v, np.int64)
    kept = C.c_uint64(0)
    s = lib().orc_filter_sum_i64(_p(k, C.c_int64), _p(v, C.c_int64), k.size, op.encode(), int(literal), C.byref(kept))
    return float(s), int(kept.value)


def group_aggregate(keys, vals):
    """-> (sorted distinct keys int64[G], stats float64[G,5] = sum,count,min,max,avg)."""
    keys, vals = _c(keys, np.int64), _c(vals, np.float64)
    g = lib().orc_group_aggregate(_p(keys, C.c_int64), _p(vals, C.c_double), keys.size, None, None, 0) if False else None
    cap = int(np.unique(keys).size)
    ok = np.empty(cap, np.int64)
    out = np.empty((cap, 5), np.float64)
    g = lib().orc_group_aggregate(_p(keys, C.c_int64), _p(vals, C.c_double), keys.size, _p(ok, C.c_int64), _p(out, C.c_double), cap)
    assert g == cap
    return ok, out


def gen_build_keys(first, count, N):
    out = np.empty(count, np.int64)
    lib().orc_gen_build_keys(_p(out, C.c_int64), first, count, N)
    return out


def gen_probe_keys(first, count, N):
    out = np.empty(count, np.int64)
    lib().orc_gen_probe_keys(_p(out, C.c_int64), first, count, N)
    return out


def gen_sm64_mod(seed, first, count, modulus, dtype=np.int64):
    if dtype == np.int64:
        out = np.empty(count, np.int64)
        lib().orc_gen_sm64_mod(_p(out, C.c_int64), seed, first, count, modulus)
    elif dtype == np.int32:
        out = np.empty(count, np.int32)
        lib().orc_gen_sm64_mod_i32(_p(out, C.c_int32), seed, first, count, modulus)
    else:
        out = np.empty(count, np.float64)
        lib().orc_gen_sm64_mod_f64(_p(out, C.c_double), seed, first, count, modulus)
    return out
