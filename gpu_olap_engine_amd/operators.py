"""PhysicalPlan operator API over pyarrow RecordBatches (harness for include/golap_host.h).

Mirrors gpu-olap-core/src/physical_plan.rs:11-84 (PhysicalPlan / PhysicalExpr / AggregateExpr) and
executor.rs:15-24 (execute).  Everything is executed by the C++ GpuExecutor in libgolap_host.so; this
module only builds the plan through the C ABI and moves batches through the Arrow C Data Interface.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional

import pyarrow as pa

from . import _native as N

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.environ.get("GOLAP_HOST_DEV_LIB") or os.path.join(_HERE, "lib", "libgolap_host.so")   # (GOLAP_HOST_DEV_LIB: developer builds, e.g. a sanitizer build)


class ArrowSchema(C.Structure):
    _fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
                ("n_children", C.c_int64), ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p),
                ("private_data", C.c_void_p)]


class ArrowArray(C.Structure):
    _fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
                ("n_children", C.c_int64), ("buffers", C.c_void_p), ("children", C.c_void_p), ("dictionary", C.c_void_p),
                ("release", C.c_void_p), ("private_data", C.c_void_p)]


class ExecProfile(C.Structure):
    """golap_exec_profile: where the wall-clock time of one golap_execute call went"""
    _fields_ = [("h2d_bytes", C.c_uint64), ("d2h_bytes", C.c_uint64), ("h2d_ms", C.c_double), ("d2h_ms", C.c_double),
                ("kernel_wait_ms", C.c_double), ("total_ms", C.c_double)]


_VP = C.c_void_p
HOST_SIGNATURES = {
    "golap_expr_column": (_VP, [C.c_uint64, C.c_char_p]),
    "golap_expr_literal": (_VP, [C.c_char_p]),
    "golap_expr_binary": (_VP, [_VP, C.c_char_p, _VP]),
    "golap_expr_free": (None, [_VP]),
    "golap_plan_batches": (_VP, [C.POINTER(C.POINTER(ArrowArray)), C.POINTER(C.POINTER(ArrowSchema)), C.c_uint32]),
    "golap_plan_table_scan": (_VP, [C.c_char_p, C.POINTER(ArrowSchema)]),
    "golap_plan_filter": (_VP, [_VP, _VP]),
    "golap_plan_projection": (_VP, [_VP, C.POINTER(_VP), C.c_uint32]),
    "golap_plan_hash_join": (_VP, [_VP, _VP, C.POINTER(_VP), C.c_uint32, C.POINTER(_VP), C.c_uint32, C.c_int]),
    "golap_plan_aggregate": (_VP, [_VP, C.POINTER(_VP), C.c_uint32, C.POINTER(C.c_int32), C.POINTER(_VP), C.c_uint32]),
    "golap_plan_sort": (_VP, [_VP, C.POINTER(_VP), C.c_uint32]),
    "golap_plan_free": (None, [_VP]),
    "golap_execute": (C.c_int32, [_VP, _VP, C.POINTER(_VP), C.POINTER(C.c_uint32)]),
    "golap_result_export": (C.c_int32, [_VP, C.c_uint32, C.POINTER(ArrowArray), C.POINTER(ArrowSchema)]),
    "golap_result_profile": (C.c_int32, [_VP, C.c_void_p]),
    "golap_host_set_transfer_threads": (None, [C.c_uint32]),
    "golap_host_set_pipelining": (None, [C.c_uint32]),
    "golap_result_free": (None, [_VP]),
    "golap_host_last_error": (C.c_char_p, []),
}

_hlib = None


def load_host():
    global _hlib
    if _hlib is None:
        N.load()   # libgolap_hip.so first (and torch's HIP runtime before it)
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError(f"{HOST_LIB_PATH} is missing; run `make -C gpu_olap_engine_amd/csrc`")
        lib = C.CDLL(HOST_LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in HOST_SIGNATURES.items():
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        _hlib = lib
    return _hlib


# ---- PhysicalExpr (physical_plan.rs:66-75) ------------------------------------------------------
@dataclass
class Column:
    index: int
    name: str


@dataclass
class Literal:
    value: str


@dataclass
class BinaryExpr:
    left: object
    op: str
    right: object


# ---- AggregateExpr (physical_plan.rs:77-84) -------------------------------------------------------
@dataclass
class Sum:
    input: object


@dataclass
class Count:
    input: object


@dataclass
class Min:
    input: object


@dataclass
class Max:
    input: object


@dataclass
class Avg:
    input: object


_AGG = {Sum: N.AGG_SUM, Count: N.AGG_COUNT, Min: N.AGG_MIN, Max: N.AGG_MAX, Avg: N.AGG_AVG}


# ---- PhysicalPlan (physical_plan.rs:11-64) --------------------------------------------------------
@dataclass
class Batches:
    """a materialised child: the Vec<RecordBatch> an executed subtree returned"""
    batches: List[pa.RecordBatch]


@dataclass
class GpuTableScan:
    table_name: str
    schema: pa.Schema
    projection: Optional[List[int]] = None


@dataclass
class GpuFilter:
    input: object
    predicate: object


@dataclass
class GpuProjection:
    input: object
    exprs: list


@dataclass
class GpuHashJoin:
    left: object
    right: object
    left_keys: list
    right_keys: list
    join_type: str = "Inner"          # JoinType, logical_plan.rs:53-58
    schema: Optional[pa.Schema] = None


@dataclass
class GpuAggregate:
    input: object
    group_by: list
    aggr_exprs: list
    schema: Optional[pa.Schema] = None


@dataclass
class GpuSort:
    input: object
    exprs: list = field(default_factory=list)


_JT = {"Inner": 0, "Left": 1, "Right": 2, "Full": 3}


class OperatorError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"golap operator error {code}: {msg}")
        self.code = code


def _expr(L, e):
    if isinstance(e, Column):
        return L.golap_expr_column(e.index, e.name.encode())
    if isinstance(e, Literal):
        return L.golap_expr_literal(str(e.value).encode())
    if isinstance(e, BinaryExpr):
        return L.golap_expr_binary(_expr(L, e.left), e.op.encode(), _expr(L, e.right))
    raise TypeError(f"not a PhysicalExpr: {e!r}")


def _expr_array(L, exprs):
    arr = (_VP * max(1, len(exprs)))(*[_expr(L, e) for e in exprs])
    return arr, len(exprs)


def _plan(L, p):
    if isinstance(p, Batches):
        n = len(p.batches)
        arrays = (ArrowArray * max(n, 1))()
        schemas = (ArrowSchema * max(n, 1))()
        for i, b in enumerate(p.batches):
            b._export_to_c(C.addressof(arrays[i]), C.addressof(schemas[i]))
        ap = (C.POINTER(ArrowArray) * max(n, 1))(*[C.pointer(arrays[i]) for i in range(max(n, 1))])
        sp = (C.POINTER(ArrowSchema) * max(n, 1))(*[C.pointer(schemas[i]) for i in range(max(n, 1))])
        h = L.golap_plan_batches(ap, sp, n)
        if not h:
            raise OperatorError(N.ERR_INVALID_ARG, L.golap_host_last_error().decode())
        return h
    if isinstance(p, GpuTableScan):
        sch = p.schema if p.projection is None else pa.schema([p.schema.field(i) for i in p.projection])
        cs = ArrowSchema()
        sch._export_to_c(C.addressof(cs))
        return L.golap_plan_table_scan(p.table_name.encode(), C.byref(cs))
    if isinstance(p, GpuFilter):
        return L.golap_plan_filter(_plan(L, p.input), _expr(L, p.predicate))
    if isinstance(p, GpuProjection):
        a, n = _expr_array(L, p.exprs)
        return L.golap_plan_projection(_plan(L, p.input), a, n)
    if isinstance(p, GpuHashJoin):
        lk, nl = _expr_array(L, p.left_keys)
        rk, nr = _expr_array(L, p.right_keys)
        return L.golap_plan_hash_join(_plan(L, p.left), _plan(L, p.right), lk, nl, rk, nr, _JT[p.join_type])
    if isinstance(p, GpuAggregate):
        gb, ng = _expr_array(L, p.group_by)
        ins, na = _expr_array(L, [a.input for a in p.aggr_exprs])
        funcs = (C.c_int32 * max(1, na))(*[_AGG[type(a)] for a in p.aggr_exprs])
        return L.golap_plan_aggregate(_plan(L, p.input), gb, ng, funcs, ins, na)
    if isinstance(p, GpuSort):
        a, n = _expr_array(L, p.exprs)
        return L.golap_plan_sort(_plan(L, p.input), a, n)
    raise TypeError(f"not a PhysicalPlan: {p!r}")


def execute(plan, ctx, profile: Optional[dict] = None) -> List[pa.RecordBatch]:
    """executor::execute (executor.rs:15-24): PhysicalPlan -> Vec<RecordBatch>, on the GPU of `ctx`.
    profile: a dict that receives golap_exec_profile of the call (bytes and wall time of the upload / download phases, total)."""
    L = load_host()
    h = _plan(L, plan)
    res, n = _VP(), C.c_uint32(0)
    st = L.golap_execute(ctx._h, h, C.byref(res), C.byref(n))
    if st != 0:
        raise OperatorError(st, L.golap_host_last_error().decode(errors="replace"))
    out = []
    try:
        if profile is not None:
            pr = ExecProfile()
            L.golap_result_profile(res, C.byref(pr))
            profile.update({f: getattr(pr, f) for f, _ in pr._fields_})
        for i in range(n.value):
            a, s = ArrowArray(), ArrowSchema()
            st = L.golap_result_export(res, i, C.byref(a), C.byref(s))
            if st != 0:
                raise OperatorError(st, L.golap_host_last_error().decode(errors="replace"))
            out.append(pa.RecordBatch._import_from_c(C.addressof(a), C.addressof(s)))
    finally:
        L.golap_result_free(res)
    return out
