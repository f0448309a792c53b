"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/golap.h
declares, its pure-host entry points answer like the reference's unit tests, and compute entry points
fail loudly (no CPU fallback) when no HIP device is present."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "golap.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from gpu_olap_engine_amd import _native
    return _native.load()


def declared_symbols():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(golap_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported(lib):
    from gpu_olap_engine_amd import _native
    syms = declared_symbols()
    assert len(syms) >= 30
    out = subprocess.check_output(["nm", "-D", "--defined-only", _native.LIB_PATH], text=True)
    exported = set(l.split()[-1] for l in out.splitlines() if l.strip())
    missing = [s for s in syms if s not in exported]
    assert not missing, f"declared in golap.h but not exported: {missing}"
    # and the ctypes binding covers every declared function
    unbound = [s for s in syms if s not in _native.SIGNATURES]
    assert not unbound, f"not bound in _native.py: {unbound}"


def test_header_compiles_as_plain_c(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "golap.h"\nint main(void){ golap_engine_config c; golap_join_params p; (void)c; (void)p; return sizeof(golap_stats) > 0 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(c), "-o", str(tmp_path / "t.o")])


def test_struct_layouts_match_ctypes(tmp_path, lib):
    """sizeof of every POD struct as the C compiler sees it == the ctypes mirror."""
    from gpu_olap_engine_amd import _native as N
    names = {"golap_engine_config": N.EngineConfig, "golap_device_info": N.DeviceInfo, "golap_slab_stats": N.SlabStats,
             "golap_filter_predicate": N.FilterPredicate, "golap_filter_params": N.FilterParams, "golap_join_params": N.JoinParams,
             "golap_partition_params": N.PartitionParams, "golap_partitioned_side": N.PartitionedSide,
             "golap_join_partitioned_params": N.JoinPartitionedParams, "golap_aggregate_spec": N.AggregateSpec,
             "golap_aggregate_params": N.AggregateParams, "golap_kernel_stat": N.KernelStat, "golap_stats": N.Stats, "golap_join_plan": N.JoinPlan,
             "golap_join_partitioned_plan": N.JoinPartitionedPlan, "golap_pack_keys_params": N.PackKeysParams}
    body = "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in names)
    c = tmp_path / "s.c"
    c.write_text(f'#include <stdio.h>\n#include "golap.h"\nint main(void){{{body} return 0;}}\n')
    exe = tmp_path / "s"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        n, sz = line.split()
        assert C.sizeof(names[n]) == int(sz), n


def test_host_only_entry_points(lib):
    from gpu_olap_engine_amd import _native as N
    assert lib.golap_abi_version() == 4      # (4: golap_device_info says which kernel forms run and what built the library; device-side ingest entry points)
    assert lib.golap_choose_strategy(500_000) == N.JOIN_BROADCAST_HASH and lib.golap_choose_strategy(5_000_000) == N.JOIN_RADIX_HASH   # join_kernel.rs:85-87
    g, b = C.c_uint32(), C.c_uint32()
    lib.golap_compute_launch_params(1024, C.byref(g), C.byref(b))
    assert (g.value, b.value) == (4, 256)                                                     # filter_kernel.rs:87-90
    lib.golap_compute_launch_params(1, C.byref(g), C.byref(b))
    assert (g.value, b.value) == (1, 256)                                                     # filter_kernel.rs:92-94
    assert lib.golap_agg_identity_value(N.AGG_SUM) == 0                                        # aggregate_kernel.rs:139-143
    assert lib.golap_agg_identity_value(N.AGG_MIN) == 2**63 - 1 and lib.golap_agg_identity_value(N.AGG_MAX) == -2**63
    ops = {"=": N.CMP_EQ, "==": N.CMP_EQ, "!=": N.CMP_NE, "<>": N.CMP_NE, ">": N.CMP_GT, ">=": N.CMP_GE, "<": N.CMP_LT, "<=": N.CMP_LE,
           "AND": N.CMP_NEVER, "": N.CMP_NEVER}
    for s, code in ops.items():
        assert lib.golap_parse_cmp_op(s.encode()) == code                                      # executor.rs:457-465
    cfg = N.EngineConfig()
    lib.golap_engine_config_default(C.byref(cfg))                                              # lib.rs:33-43
    assert (cfg.max_gpu_memory, cfg.num_streams, cfg.use_unified_memory, cfg.batch_size, cfg.enable_cache, cfg.radix_bits) == (8 << 30, 8, 1, 1 << 20, 1, 8)


def test_join_plan_is_host_only(lib):
    from gpu_olap_engine_amd import _native as N
    p = N.JoinPlan()
    for n, (b1, b2) in {500_000: (8, 0), 1_000_000: (9, 0), 100_000_000: (8, 7), 1_000_000_000: (9, 9)}.items():
        assert lib.golap_join_plan_get(None, n, 1, C.byref(p)) == 0
        assert (p.bits1, p.bits2) == (b1, b2), n
        assert p.n_subpartitions == 1 << (b1 + b2) and p.table_capacity == 4096 and p.tile_rows in (4096, 12288, 24576)
        assert p.no_histogram_last_pass == (1 if b2 >= 5 else 0) and p.chunks1 >= 1
        assert p.no_histogram_first_pass == (1 if n >= 100_000_000 else 0) and p.optimistic_narrow_keys == p.no_histogram_first_pass
        assert p.tile_rows == (24576 if n >= 100_000_000 else 12288 if b1 >= 7 else 4096)


def test_no_device_fails_loudly(lib):
    """There is no CPU fallback: without a GPU, context creation reports GOLAP_ERR_NO_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gpu_olap_engine_amd import _native as N
    h = C.c_void_p()
    st = lib.golap_ctx_create(None, C.byref(h))
    assert st == N.ERR_NO_DEVICE and not h.value
    assert b"no CPU fallback" in lib.golap_last_error(None)
    from gpu_olap_engine_amd.engine import Context, GolapError
    with pytest.raises(GolapError):
        Context()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under gpu_olap_engine_amd/ or include/ may reference it."""
    bad = []
    for base in ("gpu_olap_engine_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"\bimport oracle\b|from oracle\b|golap_oracle|liboracle|orc_[a-z]", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
