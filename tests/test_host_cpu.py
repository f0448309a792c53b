"""CPU tests of the operator-level host layer (libgolap_host.so): exported symbols, Arrow C Data Interface
import/export, and the device-free PhysicalPlan arms (scan / sort / projection, executor.rs:110-155, :184-214, :361-370)."""
import os
import re
import subprocess

import numpy as np
import pyarrow as pa
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NoCtx:
    _h = None


@pytest.fixture(scope="module")
def O():
    import __graft_entry__ as ge
    ge.build()
    from gpu_olap_engine_amd import operators
    operators.load_host()
    return operators


def test_host_header_symbols_exported(O):
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "golap_host.h")).read(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(golap_[a-z0-9_]+)\s*\(", src)))
    out = subprocess.check_output(["nm", "-D", "--defined-only", O.HOST_LIB_PATH], text=True)
    exported = set(l.split()[-1] for l in out.splitlines() if l.strip())
    assert len(syms) >= 15 and not [s for s in syms if s not in exported]
    assert not [s for s in syms if s not in O.HOST_SIGNATURES]


def test_host_header_compiles_as_plain_c(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "golap_host.h"\nint main(void){ struct ArrowArray a; struct ArrowSchema s; (void)a; (void)s; return 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(c), "-o", str(tmp_path / "t.o")])


def test_arrow_roundtrip_offsets_and_nulls(O):
    a = pa.array([1, None, 3, 4, 5, None, 7], pa.int64())
    f = pa.array([0.5, 1.5, None, 3.5, 4.5, 5.5, 6.5], pa.float64())
    i = pa.array([10, 20, 30, 40, 50, 60, 70], pa.int32())
    full = pa.RecordBatch.from_arrays([a, f, i], names=["a", "f", "i"])
    sl = full.slice(1, 5)                                     # non-zero offset + validity bitmap
    out = O.execute(O.GpuSort(O.Batches([full, sl])), NoCtx())
    assert [b.num_rows for b in out] == [7, 5]
    # the value buffer holds 0 under a null (record_batch_convert.rs:102-110) and the validity bitmap travels with the column
    # (nulls survive the operator layer instead of turning into zeros)
    assert out[0].column(0).to_pylist() == [1, None, 3, 4, 5, None, 7]
    assert out[0].column(0).fill_null(-1).to_pylist() == [1, -1, 3, 4, 5, -1, 7]
    assert np.frombuffer(out[0].column(0).buffers()[1], dtype=np.int64, count=7).tolist() == [1, 0, 3, 4, 5, 0, 7]
    assert out[0].column(1).to_pylist() == [0.5, 1.5, None, 3.5, 4.5, 5.5, 6.5]
    assert out[1].column(0).to_pylist() == [None, 3, 4, 5, None] and out[1].column(2).to_pylist() == [20, 30, 40, 50, 60]
    assert out[1].schema.names == ["a", "f", "i"] and out[1].schema.field(2).type == pa.int32()


def test_scan_projection_without_device(O):
    sch = pa.schema([("x", pa.int32()), ("y", pa.float32())])
    out = O.execute(O.GpuTableScan("t", sch, projection=[1]), NoCtx())       # executor.rs:121-128 projected schema
    assert len(out) == 1 and out[0].num_rows == 0 and out[0].schema.names == ["y"] and out[0].schema.field(0).type == pa.float64()
    b = pa.RecordBatch.from_arrays([pa.array(np.arange(4, dtype=np.int64)), pa.array(np.arange(4.0))], names=["p", "q"])
    out = O.execute(O.GpuProjection(O.Batches([b]), [O.Column(5, "q"), O.Column(0, "nope"), O.Literal("-3")]), NoCtx())
    assert out[0].schema.names == ["q", "nope", "lit(-3)"]                    # name first, then index (executor.rs:379-388)
    assert out[0].column(1).to_pylist() == [0, 1, 2, 3] and out[0].column(2).to_pylist() == [-3] * 4
    with pytest.raises(O.OperatorError):
        O.execute(O.GpuProjection(O.Batches([b]), [O.BinaryExpr(O.Column(0, "p"), "+", O.Literal("1"))]), NoCtx())


def test_device_arms_fail_loudly_without_ctx(O):
    b = pa.RecordBatch.from_arrays([pa.array(np.arange(4, dtype=np.int64))], names=["a"])
    with pytest.raises(O.OperatorError):     # no CPU fallback: the filter needs the device
        O.execute(O.GpuFilter(O.Batches([b]), O.BinaryExpr(O.Column(0, "a"), ">", O.Literal("1"))), NoCtx())


def test_narrow_types_pass_through_with_their_own_types(O):
    """Columns no operator touches leave as they came -- type, offset, nulls -- for every numeric Arrow type the importer knows (they are
    only widened when an operator takes the batch to the device: record_batch_convert.rs:35-110)."""
    n = 37
    rng = np.random.default_rng(3)
    specs = [("i8", pa.int8(), np.int8), ("u8", pa.uint8(), np.uint8), ("i16", pa.int16(), np.int16), ("u16", pa.uint16(), np.uint16),
             ("i32", pa.int32(), np.int32), ("u32", pa.uint32(), np.uint32), ("i64", pa.int64(), np.int64), ("u64", pa.uint64(), np.uint64),
             ("f32", pa.float32(), np.float32), ("f64", pa.float64(), np.float64)]
    arrays, expect = [], []
    for nm, t, dt in specs:
        v = (rng.random(n) * 100).astype(dt)
        nulls = rng.random(n) < 0.3
        arrays.append(pa.array([None if m else x.item() for x, m in zip(v, nulls)], t))
        expect.append([None if m else x.item() for x, m in zip(v, nulls)])
    full = pa.RecordBatch.from_arrays(arrays, names=[s[0] for s in specs])
    sl = full.slice(5, 29)
    out = O.execute(O.GpuSort(O.Batches([full, sl])), NoCtx())
    for j, (nm, t, _) in enumerate(specs):
        assert out[0].schema.field(j).type == t and out[1].schema.field(j).type == t, nm
        assert out[0].column(j).to_pylist() == expect[j], nm
        assert out[1].column(j).to_pylist() == expect[j][5:34], nm
