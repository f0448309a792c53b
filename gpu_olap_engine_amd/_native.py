"""ctypes binding of include/golap.h (the C-ABI drop-in boundary).

The HIP library is REQUIRED: there is no CPU fallback anywhere in this package.
`import torch` must happen before the library is loaded so that both resolve
the same libamdhip64.so.7 (torch bundles its own copy with that SONAME).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GOLAP_DEV_LIB") or os.path.join(_HERE, "lib", "libgolap_hip.so")   # (GOLAP_DEV_LIB: developer builds, e.g. `make DIAG=1 OUT=...`)

GOLAP_OK = 0
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_UNSUPPORTED, ERR_CAPACITY, ERR_PEER = 1, 2, 3, 4, 5, 6, 7

JOIN_RADIX_HASH, JOIN_SORT_MERGE, JOIN_BROADCAST_HASH = 0, 1, 2
JT_INNER, JT_LEFT, JT_RIGHT, JT_FULL = 0, 1, 2, 3
NULL_ROW_ID = 0xFFFFFFFF
CMP_EQ, CMP_NE, CMP_LT, CMP_LE, CMP_GT, CMP_GE, CMP_NEVER = range(7)
AGG_SUM, AGG_COUNT, AGG_MIN, AGG_MAX, AGG_AVG = range(5)
DT_INT64, DT_FLOAT64, DT_INT32 = 0, 1, 2
KEYS_DENSE, KEYS_HASHED = 0, 1
(WIDEN_I8, WIDEN_I16, WIDEN_I32, WIDEN_I64, WIDEN_U8, WIDEN_U16, WIDEN_U32, WIDEN_U64, WIDEN_F32, WIDEN_F64) = range(10)
GEN_BUILD_KEYS, GEN_SM64_MOD_I64, GEN_SM64_MOD_I32, GEN_SM64_MOD_F64, GEN_SHIFT32_I64 = range(5)
HOST_TO_DEVICE, DEVICE_TO_HOST = 0, 1
MAX_KERNEL_STATS = 64
PROVISION_NONE = 0xFFFFFFFF


class GolapError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"golap error {code}: {msg}")
        self.code = code
        self.message = msg


class EngineConfig(C.Structure):
    """golap_engine_config == EngineConfig, gpu-olap-core/src/lib.rs:20-31."""
    _fields_ = [("max_gpu_memory", C.c_uint64), ("num_streams", C.c_uint32), ("use_unified_memory", C.c_uint32),
                ("batch_size", C.c_uint64), ("enable_cache", C.c_uint32), ("device_id", C.c_int32),
                ("radix_bits", C.c_uint32), ("enable_profiling", C.c_uint32), ("max_slabs_per_class", C.c_uint32),
                ("provision_sigmas", C.c_uint32)]


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("gcn_arch", C.c_char * 64), ("compute_units", C.c_uint32),
                ("lds_bytes_per_block", C.c_uint32), ("wavefront_size", C.c_uint32), ("clock_mhz", C.c_uint32),
                ("memory_clock_mhz", C.c_uint32), ("memory_bus_bits", C.c_uint32), ("total_mem_bytes", C.c_uint64),
                ("free_mem_bytes", C.c_uint64), ("l2_bytes", C.c_uint32), ("hidden_memory_forms", C.c_uint32), ("built_with", C.c_char * 32),
                ("audited_with", C.c_char * 32)]


class SlabStats(C.Structure):
    _fields_ = [("n_classes", C.c_uint32), ("class_size", C.c_uint64 * 16), ("allocated", C.c_uint32 * 16),
                ("pooled", C.c_uint32 * 16), ("bytes_reserved", C.c_uint64)]


class FilterPredicate(C.Structure):
    """codegen.rs:24-28."""
    _fields_ = [("col_index", C.c_uint64), ("op", C.c_int32), ("literal", C.c_int64)]


class FilterParams(C.Structure):
    """filter_kernel.rs:17-28."""
    _fields_ = [("column_ptr", C.c_uint64), ("n_rows", C.c_uint64), ("out_mask_ptr", C.c_uint64),
                ("grid_dim", C.c_uint32), ("block_dim", C.c_uint32), ("validity_ptr", C.c_uint64)]


class JoinParams(C.Structure):
    """join_kernel.rs:20-27 + extensions."""
    _fields_ = [("build_key_ptr", C.c_uint64), ("probe_key_ptr", C.c_uint64), ("n_build", C.c_uint64),
                ("n_probe", C.c_uint64), ("out_pairs_ptr", C.c_uint64), ("out_count_ptr", C.c_uint64),
                ("max_pairs", C.c_uint64), ("build_rid_base", C.c_uint32), ("probe_rid_base", C.c_uint32), ("join_type", C.c_int32),
                ("n_key_cols", C.c_uint32), ("build_key_extra", C.c_uint64 * 3), ("probe_key_extra", C.c_uint64 * 3),
                ("build_validity", C.c_uint64 * 4), ("probe_validity", C.c_uint64 * 4)]


class PartitionParams(C.Structure):
    _fields_ = [("key_ptr", C.c_uint64), ("row_id_ptr", C.c_uint64), ("n_rows", C.c_uint64), ("rid_base", C.c_uint32),
                ("n_owners", C.c_uint32), ("extra_bits", C.c_uint32), ("out_key_ptr", C.c_uint64), ("out_row_id_ptr", C.c_uint64),
                ("out_offsets_ptr", C.c_uint64), ("narrow", C.c_uint32), ("reserved0", C.c_uint32), ("wide_flag_ptr", C.c_uint64)]


class PartitionedSide(C.Structure):
    _fields_ = [("key_ptr", C.c_uint64), ("row_id_ptr", C.c_uint64), ("n_rows", C.c_uint64),
                ("seg_off", C.POINTER(C.c_uint64)), ("seg_len", C.POINTER(C.c_uint64))]


class JoinPartitionedParams(C.Structure):
    _fields_ = [("build", PartitionedSide), ("probe", PartitionedSide), ("n_slots", C.c_uint32), ("n_sources", C.c_uint32),
                ("pre_bits", C.c_uint32), ("total_build_rows_hint", C.c_uint64), ("out_pairs_ptr", C.c_uint64),
                ("out_count_ptr", C.c_uint64), ("max_pairs", C.c_uint64)]


class JoinPartitionedPlan(C.Structure):
    _fields_ = [("n_groups", C.c_uint32), ("n_sources", C.c_uint32), ("pre_bits", C.c_uint32), ("total_build_rows_hint", C.c_uint64),
                ("build_rows", C.c_uint64), ("probe_rows", C.c_uint64), ("narrow", C.c_uint32), ("reserved0", C.c_uint32)]


class P2POp(C.Structure):
    _fields_ = [("peer", C.c_int32), ("is_send", C.c_int32), ("ptr", C.c_uint64), ("bytes", C.c_uint64)]


ALL_GATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(P2POp), C.c_uint32, C.c_void_p)


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather", ALL_GATHER_FN), ("exchange", EXCHANGE_FN)]


COMM_ID_BYTES = 128


class ShardedJoinParams(C.Structure):
    _fields_ = [("build_key_ptr", C.c_uint64), ("probe_key_ptr", C.c_uint64), ("n_build_local", C.c_uint64), ("n_probe_local", C.c_uint64),
                ("build_rid_base", C.c_uint32), ("probe_rid_base", C.c_uint32), ("n_build_total", C.c_uint64), ("n_probe_total", C.c_uint64),
                ("build_rows_per_rank", C.POINTER(C.c_uint64)), ("probe_rows_per_rank", C.POINTER(C.c_uint64)),
                ("out_pairs_ptr", C.c_uint64), ("out_count_ptr", C.c_uint64), ("max_pairs", C.c_uint64),
                ("join_type", C.c_int32), ("force_exact", C.c_int32), ("max_msg_bytes", C.c_uint64),
                ("n_key_cols", C.c_uint32), ("reserved0", C.c_uint32), ("build_key_extra", C.c_uint64 * 3), ("probe_key_extra", C.c_uint64 * 3),
                ("build_validity", C.c_uint64 * 4), ("probe_validity", C.c_uint64 * 4)]


class ShardedJoinInfo(C.Structure):
    _fields_ = [("fast_path", C.c_uint32), ("fast_path_abandoned", C.c_uint32), ("extra_bits", C.c_uint32), ("n_groups", C.c_uint32),
                ("local_bits", C.c_uint32), ("host_syncs", C.c_uint32), ("rows_received", C.c_uint64 * 2), ("bytes_sent", C.c_uint64 * 2),
                ("chunks", C.c_uint32), ("sub_ranges", C.c_uint32), ("stage_ms", (C.c_float * 2) * 8), ("total_ms", C.c_float), ("halves", C.c_uint32)]


SHARD_STAGES = ["partition_build", "partition_probe", "exchange_build", "exchange_probe", "ring_build", "ring_probe", "join", "status"]


class ShardedPlan(C.Structure):
    _fields_ = [("eligible", C.c_uint32), ("extra_bits", C.c_uint32), ("n_buckets", C.c_uint32), ("buckets_per_rank", C.c_uint32),
                ("local_bits", C.c_uint32), ("sub_ranges", C.c_uint32), ("chunks", C.c_uint32), ("cap_rows", (C.c_uint64 * 64) * 2),
                ("cap_rows_last", (C.c_uint64 * 64) * 2), ("msg_bytes", (C.c_uint64 * 64) * 2), ("msg_pieces", (C.c_uint32 * 64) * 2)]


class AggregateSpec(C.Structure):
    """aggregate_kernel.rs:38-44."""
    _fields_ = [("func", C.c_int32), ("value_col_index", C.c_uint64), ("output_col_index", C.c_uint64)]


class AggregateParams(C.Structure):
    """aggregate_kernel.rs:47-58 + extensions."""
    _fields_ = [("group_key_ptr", C.c_uint64), ("value_ptrs", C.POINTER(C.c_uint64)), ("n_value_cols", C.c_uint32),
                ("n_rows", C.c_uint64), ("n_groups", C.c_uint32), ("out_ptr", C.c_uint64), ("key_dtype", C.c_int32),
                ("value_dtypes", C.POINTER(C.c_int32)), ("out_count_ptr", C.c_uint64), ("key_mode", C.c_int32), ("reserved0", C.c_int32),
                ("out_keys_ptr", C.c_uint64), ("out_n_groups_ptr", C.c_uint64), ("value_validity_ptrs", C.POINTER(C.c_uint64)),
                ("out_slot_ids_ptr", C.c_uint64)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double), ("bytes", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("n", C.c_uint32), ("k", KernelStat * MAX_KERNEL_STATS)]


class JoinPlan(C.Structure):
    _fields_ = [("bits1", C.c_uint32), ("bits2", C.c_uint32), ("tile_rows", C.c_uint32), ("chunks1", C.c_uint32),
                ("slots2", C.c_uint32), ("table_capacity", C.c_uint32), ("n_subpartitions", C.c_uint64),
                ("no_histogram_last_pass", C.c_uint32), ("no_histogram_first_pass", C.c_uint32),
                ("optimistic_narrow_keys", C.c_uint32), ("reserved0", C.c_uint32)]


class PackKeysParams(C.Structure):
    _fields_ = [("col_ptrs", C.c_uint64 * 4), ("validity_ptrs", C.c_uint64 * 4), ("min_value", C.c_int64 * 4), ("weight", C.c_uint64 * 4),
                ("null_digit", C.c_uint64 * 4), ("n_cols", C.c_uint32), ("reserved0", C.c_uint32), ("n_rows", C.c_uint64), ("out_ptr", C.c_uint64)]


# every symbol include/golap.h declares: name -> (restype, argtypes)
_P = C.POINTER
_VP = C.c_void_p
SIGNATURES = {
    "golap_abi_version": (C.c_uint32, []),
    "golap_engine_config_default": (None, [_P(EngineConfig)]),
    "golap_ctx_create": (C.c_int32, [_P(EngineConfig), _P(_VP)]),
    "golap_ctx_destroy": (None, [_VP]),
    "golap_last_error": (C.c_char_p, [_VP]),
    "golap_debug_set": (C.c_int32, [_VP, C.c_char_p, C.c_char_p]),
    "golap_device_info_get": (C.c_int32, [_VP, _P(DeviceInfo)]),
    "golap_ctx_synchronize": (C.c_int32, [_VP, _VP]),
    "golap_slab_alloc": (C.c_int32, [_VP, C.c_uint64, _P(C.c_uint64), _P(C.c_uint32)]),
    "golap_slab_free": (C.c_int32, [_VP, C.c_uint64, C.c_uint32]),
    "golap_slab_get_stats": (C.c_int32, [_VP, _P(SlabStats)]),
    "golap_xfer_enqueue": (C.c_int32, [_VP, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]),
    "golap_xfer_synchronize": (C.c_int32, [_VP]),
    "golap_parse_cmp_op": (C.c_int32, [C.c_char_p]),
    "golap_compute_launch_params": (None, [C.c_uint64, _P(C.c_uint32), _P(C.c_uint32)]),
    "golap_filter_launch": (C.c_int32, [_VP, _P(FilterPredicate), _P(FilterParams), _VP]),
    "golap_filter_compact": (C.c_int32, [_VP, _P(FilterPredicate), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _VP]),
    "golap_filter_compact_nullable": (C.c_int32, [_VP, _P(FilterPredicate), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _VP]),
    "golap_filter_compact_and": (C.c_int32, [_VP, _P(FilterPredicate), _P(C.c_uint64), _P(C.c_uint64), C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, _VP]),
    "golap_filter_sum_nullable": (C.c_int32, [_VP, _P(FilterPredicate), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_uint64, C.c_uint64, _VP]),
    "golap_take_8byte": (C.c_int32, [_VP, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _VP]),
    "golap_filter_sum": (C.c_int32, [_VP, _P(FilterPredicate), C.c_uint64, C.c_uint64, C.c_int32, C.c_uint64, C.c_uint64, _VP]),
    "golap_choose_strategy": (C.c_int32, [C.c_uint64]),
    "golap_join": (C.c_int32, [_VP, C.c_int32, _P(JoinParams), _VP]),
    "golap_join_scratch_bytes": (C.c_uint64, [_VP, C.c_uint64, C.c_uint64]),
    "golap_radix_partition": (C.c_int32, [_VP, _P(PartitionParams), _VP]),
    "golap_join_partitioned": (C.c_int32, [_VP, _P(JoinPartitionedParams), _VP]),
    "golap_join_partitioned_begin": (C.c_int32, [_VP, _P(JoinPartitionedPlan), _VP]),
    "golap_join_partitioned_narrow_ok": (C.c_int32, [_P(JoinPartitionedPlan)]),
    "golap_join_partitioned_side": (C.c_int32, [_VP, C.c_int32, _P(PartitionedSide), _VP]),
    "golap_join_partitioned_finish": (C.c_int32, [_VP, C.c_uint64, C.c_uint64, C.c_uint64, _VP]),
    "golap_join_partitioned_abort": (C.c_int32, [_VP]),
    "golap_comm_unique_id": (C.c_int32, [_P(C.c_uint8)]),
    "golap_comm_create_rccl": (C.c_int32, [_VP, _P(C.c_uint8), C.c_int32, C.c_int32, _P(_VP)]),
    "golap_comm_create_custom": (C.c_int32, [_VP, _P(Transport), C.c_int32, C.c_int32, _P(_VP)]),
    "golap_comm_destroy": (None, [_VP]),
    "golap_sharded_join": (C.c_int32, [_VP, _VP, _P(ShardedJoinParams), _VP]),
    "golap_sharded_join_info_get": (C.c_int32, [_VP, _P(ShardedJoinInfo)]),
    "golap_comm_selftest": (C.c_int32, [_VP, _VP, C.c_uint64, _P(C.c_uint64), _P(C.c_uint64)]),
    "golap_shard_range": (None, [C.c_uint64, C.c_uint32, C.c_uint32, _P(C.c_uint64), _P(C.c_uint64)]),
    "golap_sharded_plan_get": (C.c_int32, [C.c_uint32, C.c_uint64, C.c_uint64, _P(C.c_uint64), _P(C.c_uint64), C.c_uint64, _P(ShardedPlan)]),
    "golap_agg_identity_value": (C.c_int64, [C.c_int32]),
    "golap_aggregate": (C.c_int32, [_VP, _P(AggregateSpec), C.c_uint32, _P(AggregateParams), _VP]),
    "golap_sharded_aggregate": (C.c_int32, [_VP, _VP, _P(AggregateSpec), C.c_uint32, _P(AggregateParams), _VP]),
    "golap_widen_to_8byte": (C.c_int32, [_VP, C.c_uint64, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, _VP]),
    "golap_hash_utf8": (C.c_int32, [_VP, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _VP]),
    "golap_expand_validity": (C.c_int32, [_VP, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _VP]),
    "golap_pack_validity": (C.c_int32, [_VP, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _VP]),
    "golap_take_1byte": (C.c_int32, [_VP, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _VP]),
    "golap_pairs_unzip": (C.c_int32, [_VP, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _VP]),
    "golap_sort_u64": (C.c_int32, [_VP, C.c_uint64, C.c_uint64, _VP]),
    "golap_key_minmax": (C.c_int32, [_VP, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _VP]),
    "golap_pack_group_keys": (C.c_int32, [_VP, _P(PackKeysParams), _VP]),
    "golap_hash_key_columns": (C.c_int32, [_VP, _P(PackKeysParams), C.c_uint64, _VP]),
    "golap_group_keys_verify": (C.c_int32, [_VP, _P(PackKeysParams), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _VP]),
    "golap_device_copy": (C.c_int32, [_VP, C.c_uint64, C.c_uint64, C.c_uint64, _VP]),
    "golap_device_fill_u8": (C.c_int32, [_VP, C.c_uint64, C.c_uint32, C.c_uint64, _VP]),
    "golap_generate": (C.c_int32, [_VP, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _VP]),
    "golap_verify_pairs": (C.c_int32, [_VP, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, _VP]),
    "golap_stats_get": (C.c_int32, [_VP, _P(Stats)]),
    "golap_stats_reset": (C.c_int32, [_VP]),
    "golap_join_plan_get": (C.c_int32, [_VP, C.c_uint64, C.c_uint32, _P(JoinPlan)]),
}

_lib = None


def load():
    """Load libgolap_hip.so (fails loudly when it is missing: no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C gpu_olap_engine_amd/csrc`. There is no CPU fallback.")
    import torch  # noqa: F401  (see module docstring: load torch's HIP runtime first)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(lib, name)   # AttributeError if a declared symbol is not exported
        f.restype, f.argtypes = res, args
    _lib = lib
    return lib


def check(ctx_handle, status):
    if status != GOLAP_OK:
        msg = load().golap_last_error(ctx_handle)
        raise GolapError(status, msg.decode(errors="replace") if msg else "")
