"""Independent numpy restatement of the reference arithmetic, used ONLY to cross-check
the C++ oracle (two independent restatements of join_kernels.cuh / executor.rs must agree)."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def fmix32(h):
    """join_kernels.cuh:26-35."""
    h = np.asarray(h, dtype=np.uint64) & M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M32
    h ^= h >> np.uint64(16)
    return h.astype(np.uint32)


def hash_int64(keys):
    """join_kernels.cuh:37-41."""
    k = np.asarray(keys, dtype=np.int64)
    folded = (k ^ (k >> np.int64(32))).astype(np.uint64) & M32
    return fmix32(folded)


def partition_id(h, shift=0):
    """join_kernels.cuh:69."""
    return (np.asarray(h, dtype=np.uint32) >> np.uint32(shift)) & np.uint32(255)


def hash_join(build, probe, build_rid_base=0, probe_rid_base=0):
    """Inner equi-join by sorting: canonical (probe_rid, build_rid) order, (M,2) uint32 [build_rid, probe_rid]."""
    build = np.asarray(build, dtype=np.int64)
    probe = np.asarray(probe, dtype=np.int64)
    order = np.argsort(build, kind="stable")
    sb = build[order]
    lo = np.searchsorted(sb, probe, side="left")
    hi = np.searchsorted(sb, probe, side="right")
    cnt = hi - lo
    m = int(cnt.sum())
    prid = np.repeat(np.arange(probe.size, dtype=np.int64), cnt)
    starts = np.repeat(lo, cnt)
    within = np.arange(m, dtype=np.int64) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    brid = order[starts + within]
    out = np.stack([brid + build_rid_base, prid + probe_rid_base], axis=1).astype(np.uint32)
    return out[np.lexsort((out[:, 0], out[:, 1]))]


def canonical(pairs):
    """sort an (M,2) [build_rid, probe_rid] array by (probe_rid, build_rid)."""
    pairs = np.asarray(pairs).reshape(-1, 2).view(np.uint32) if pairs.dtype != np.uint32 else pairs.reshape(-1, 2)
    return pairs[np.lexsort((pairs[:, 0], pairs[:, 1]))]


def sm64(seed, i):
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.asarray(i, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def filter_mask(col, op, lit):
    col = np.asarray(col, dtype=np.int64)
    f = {"=": np.equal, "==": np.equal, "!=": np.not_equal, "<>": np.not_equal, ">": np.greater, ">=": np.greater_equal,
         "<": np.less, "<=": np.less_equal}.get(op)
    if f is None:
        return np.zeros(col.size, np.uint8)     # executor.rs:464
    return f(col, np.int64(lit)).astype(np.uint8)
