//! Raw FFI of `include/golap.h` (ABI version 4), one declaration per C declaration, same names and field order.
//!
//! NOT COMPILED in the authoring environment (no Rust toolchain): layout-checked only.  `tests/test_rust_binding.py` parses this
//! file and the header and compares EVERY `#[repr(C)]` struct field by field and every bound function's arity; the `unsafe`
//! wrapper bodies and `build.rs` are unverified until a maintainer builds the crate.  Reference interfaces each item replaces are cited as in the header (paths relative to the reference root).
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_void};

pub const GOLAP_ABI_VERSION: u32 = 4;
pub const GOLAP_OK: i32 = 0;
pub const GOLAP_ERR_INVALID_ARG: i32 = 1;
pub const GOLAP_ERR_NO_DEVICE: i32 = 2;
pub const GOLAP_ERR_HIP: i32 = 3;
pub const GOLAP_ERR_OOM: i32 = 4;
pub const GOLAP_ERR_UNSUPPORTED: i32 = 5;
pub const GOLAP_ERR_CAPACITY: i32 = 6;
/// collective call: ANOTHER rank failed before the exchange; every rank has left the call together
pub const GOLAP_ERR_PEER: i32 = 7;
pub const GOLAP_SHARD_STAGES: usize = 8;

#[repr(C)]
pub struct golap_ctx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct golap_comm {
    _p: [u8; 0],
}

/// == EngineConfig, gpu-olap-core/src/lib.rs:20-31
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct golap_engine_config {
    pub max_gpu_memory: u64,
    pub num_streams: u32,
    pub use_unified_memory: u32,
    pub batch_size: u64,
    pub enable_cache: u32,
    pub device_id: i32,
    pub radix_bits: u32,
    pub enable_profiling: u32,
    pub max_slabs_per_class: u32,
    pub provision_sigmas: u32,
}

/// == FilterPredicate, gpu-kernel-compiler/src/codegen.rs:24-28 (CmpOp order Eq,Ne,Lt,Le,Gt,Ge = 0..5, :11-21)
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct golap_filter_predicate {
    pub col_index: u64,
    pub op: i32,
    pub literal: i64,
}

/// == FilterParams, filter_kernel.rs:17-28 (+ validity)
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct golap_filter_params {
    pub column_ptr: u64,
    pub n_rows: u64,
    pub out_mask_ptr: u64,
    pub grid_dim: u32,
    pub block_dim: u32,
    pub validity_ptr: u64,
}

/// == JoinParams, join_kernel.rs:20-27 (first six fields), + extensions
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct golap_join_params {
    pub build_key_ptr: u64,
    pub probe_key_ptr: u64,
    pub n_build: u64,
    pub n_probe: u64,
    pub out_pairs_ptr: u64,
    pub out_count_ptr: u64,
    pub max_pairs: u64,
    pub build_rid_base: u32,
    pub probe_rid_base: u32,
    pub join_type: i32,
    pub n_key_cols: u32,
    pub build_key_extra: [u64; 3],
    pub probe_key_extra: [u64; 3],
    pub build_validity: [u64; 4],
    pub probe_validity: [u64; 4],
}

/// == AggregateSpec, aggregate_kernel.rs:38-44 (AggFunc order Sum,Count,Min,Max,Avg = 0..4, :17-24)
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct golap_aggregate_spec {
    pub func: i32,
    pub value_col_index: u64,
    pub output_col_index: u64,
}

/// == AggregateParams, aggregate_kernel.rs:47-58 (+ dtypes, counts, hashed keys)
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct golap_aggregate_params {
    pub group_key_ptr: u64,
    pub value_ptrs: *const u64,
    pub n_value_cols: u32,
    pub n_rows: u64,
    pub n_groups: u32,
    pub out_ptr: u64,
    pub key_dtype: i32,
    pub value_dtypes: *const i32,
    pub out_count_ptr: u64,
    pub key_mode: i32,
    pub reserved0: i32,
    pub out_keys_ptr: u64,
    pub out_n_groups_ptr: u64,
    pub value_validity_ptrs: *const u64,
    pub out_slot_ids_ptr: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct golap_p2p_op {
    pub peer: i32,
    pub is_send: i32,
    pub ptr: u64,
    pub bytes: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct golap_transport {
    pub user: *mut c_void,
    pub all_gather: Option<unsafe extern "C" fn(user: *mut c_void, send_ptr: u64, recv_ptr: u64, bytes_per_rank: u64, stream: *mut c_void) -> i32>,
    pub exchange: Option<unsafe extern "C" fn(user: *mut c_void, ops: *const golap_p2p_op, n_ops: u32, stream: *mut c_void) -> i32>,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct golap_sharded_join_params {
    pub build_key_ptr: u64,
    pub probe_key_ptr: u64,
    pub n_build_local: u64,
    pub n_probe_local: u64,
    pub build_rid_base: u32,
    pub probe_rid_base: u32,
    pub n_build_total: u64,
    pub n_probe_total: u64,
    pub build_rows_per_rank: *const u64,
    pub probe_rows_per_rank: *const u64,
    pub out_pairs_ptr: u64,
    pub out_count_ptr: u64,
    pub max_pairs: u64,
    pub join_type: i32,
    pub force_exact: i32,
    pub max_msg_bytes: u64,
    pub n_key_cols: u32,
    pub reserved0: u32,
    pub build_key_extra: [u64; 3],
    pub probe_key_extra: [u64; 3],
    pub build_validity: [u64; 4],
    pub probe_validity: [u64; 4],
}

/// what the last `golap_sharded_join` on a communicator did; `stage_ms[stage][begin, end]` in ms since the call's first kernel
/// (stages: partition build/probe, exchange build/probe, ring build/probe, join, status)
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct golap_sharded_join_info {
    pub fast_path: u32,
    pub fast_path_abandoned: u32,
    pub extra_bits: u32,
    pub n_groups: u32,
    pub local_bits: u32,
    pub host_syncs: u32,
    pub rows_received: [u64; 2],
    pub bytes_sent: [u64; 2],
    pub chunks: u32,
    pub sub_ranges: u32,
    pub stage_ms: [[f32; 2]; 8],
    pub total_ms: f32,
    pub halves: u32,
}

/// host-only description of the fast path's messages (every rank computes the same numbers)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct golap_sharded_plan {
    pub eligible: u32,
    pub extra_bits: u32,
    pub n_buckets: u32,
    pub buckets_per_rank: u32,
    pub local_bits: u32,
    pub sub_ranges: u32,
    pub chunks: u32,
    pub cap_rows: [[u64; 64]; 2],
    pub cap_rows_last: [[u64; 64]; 2],
    pub msg_bytes: [[u64; 64]; 2],
    pub msg_pieces: [[u32; 64]; 2],
}

/// stand-alone radix partition (join_kernels.cuh:45-76), owner-major for the partition exchange
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct golap_partition_params {
    pub key_ptr: u64,
    pub row_id_ptr: u64,
    pub n_rows: u64,
    pub rid_base: u32,
    pub n_owners: u32,
    pub extra_bits: u32,
    pub out_key_ptr: u64,
    pub out_row_id_ptr: u64,
    pub out_offsets_ptr: u64,
    pub narrow: u32,
    pub reserved0: u32,
    pub wide_flag_ptr: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct golap_partitioned_side {
    pub key_ptr: u64,
    pub row_id_ptr: u64,
    pub n_rows: u64,
    pub seg_off: *const u64,
    pub seg_len: *const u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct golap_join_partitioned_plan {
    pub n_groups: u32,
    pub n_sources: u32,
    pub pre_bits: u32,
    pub total_build_rows_hint: u64,
    pub build_rows: u64,
    pub probe_rows: u64,
    pub narrow: u32,
    pub reserved0: u32,
}

/// Composite GROUP BY key as one exact int64 (golap_pack_group_keys, ABI 4).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct golap_pack_keys_params {
    pub col_ptrs: [u64; 4],
    pub validity_ptrs: [u64; 4],
    pub min_value: [i64; 4],
    pub weight: [u64; 4],
    pub null_digit: [u64; 4],
    pub n_cols: u32,
    pub reserved0: u32,
    pub n_rows: u64,
    pub out_ptr: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct golap_join_plan {
    pub bits1: u32,
    pub bits2: u32,
    pub tile_rows: u32,
    pub chunks1: u32,
    pub slots2: u32,
    pub table_capacity: u32,
    pub n_subpartitions: u64,
    pub no_histogram_last_pass: u32,
    pub no_histogram_first_pass: u32,
    pub optimistic_narrow_keys: u32,
    pub reserved0: u32,
}

extern "C" {
    // ---- context (EngineConfig, lib.rs:20-43) ----
    pub fn golap_abi_version() -> u32;
    pub fn golap_engine_config_default(cfg: *mut golap_engine_config);
    pub fn golap_ctx_create(cfg: *const golap_engine_config, out_ctx: *mut *mut golap_ctx) -> i32;
    pub fn golap_ctx_destroy(ctx: *mut golap_ctx);
    pub fn golap_last_error(ctx: *const golap_ctx) -> *const c_char;
    pub fn golap_ctx_synchronize(ctx: *mut golap_ctx, stream: *mut c_void) -> i32;
    // ---- SlabAllocator::{allocate, free} (slab_allocator.rs:50, :71) ----
    pub fn golap_slab_alloc(ctx: *mut golap_ctx, size: u64, out_ptr: *mut u64, out_class_idx: *mut u32) -> i32;
    pub fn golap_slab_free(ctx: *mut golap_ctx, ptr: u64, class_idx: u32) -> i32;
    // ---- TransferQueue::{enqueue_transfer, synchronize} (transfer_queue.rs:55, :133) ----
    pub fn golap_xfer_enqueue(ctx: *mut golap_ctx, direction: i32, src: u64, dst: u64, size: u64, stream_id: u32) -> i32;
    pub fn golap_xfer_synchronize(ctx: *mut golap_ctx) -> i32;
    // ---- FilterKernel (filter_kernel.rs:42-79) ----
    pub fn golap_parse_cmp_op(op: *const c_char) -> i32;
    pub fn golap_compute_launch_params(n_rows: u64, grid: *mut u32, block: *mut u32);
    pub fn golap_filter_launch(ctx: *mut golap_ctx, pred: *const golap_filter_predicate, params: *const golap_filter_params, stream: *mut c_void) -> i32;
    pub fn golap_filter_compact(ctx: *mut golap_ctx, pred: *const golap_filter_predicate, column_ptr: u64, n_rows: u64, out_indices_ptr: u64,
                                out_count_ptr: u64, stream: *mut c_void) -> i32;
    pub fn golap_filter_compact_nullable(ctx: *mut golap_ctx, pred: *const golap_filter_predicate, column_ptr: u64, validity_ptr: u64, n_rows: u64,
                                         out_indices_ptr: u64, out_count_ptr: u64, stream: *mut c_void) -> i32;
    pub fn golap_filter_compact_and(ctx: *mut golap_ctx, preds: *const golap_filter_predicate, column_ptrs: *const u64, validity_ptrs: *const u64,
                                    n_preds: u32, n_rows: u64, out_indices_ptr: u64, out_count_ptr: u64, stream: *mut c_void) -> i32;
    pub fn golap_take_8byte(ctx: *mut golap_ctx, column_ptr: u64, indices_ptr: u64, count_ptr: u64, n_indices_max: u64, out_ptr: u64,
                            stream: *mut c_void) -> i32;
    pub fn golap_filter_sum(ctx: *mut golap_ctx, pred: *const golap_filter_predicate, key_column_ptr: u64, value_column_ptr: u64, value_is_f64: i32,
                            n_rows: u64, out_ptr: u64, stream: *mut c_void) -> i32;
    // ---- JoinKernel (join_kernel.rs:39-77) ----
    pub fn golap_choose_strategy(n_build: u64) -> i32;
    pub fn golap_join(ctx: *mut golap_ctx, strategy: i32, params: *const golap_join_params, stream: *mut c_void) -> i32;
    pub fn golap_join_scratch_bytes(ctx: *mut golap_ctx, n_build: u64, n_probe: u64) -> u64;
    // ---- AggregateKernel (aggregate_kernel.rs:80-131) ----
    pub fn golap_agg_identity_value(func: i32) -> i64;
    pub fn golap_aggregate(ctx: *mut golap_ctx, specs: *const golap_aggregate_spec, n_specs: u32, params: *const golap_aggregate_params,
                           stream: *mut c_void) -> i32;
    // ---- arrow-interop widening (record_batch_convert.rs:35-110) ----
    pub fn golap_widen_to_8byte(ctx: *mut golap_ctx, in_ptr: u64, in_dtype: i32, validity_ptr: u64, n_rows: u64, out_ptr: u64, stream: *mut c_void) -> i32;
    pub fn golap_hash_utf8(ctx: *mut golap_ctx, offsets_ptr: u64, data_ptr: u64, validity_ptr: u64, n_rows: u64, out_ptr: u64, stream: *mut c_void) -> i32;
    // ---- device-side ingest and materialisation (ABI 4; record_batch_convert.rs:35-110, :146-178 on the device) ----
    pub fn golap_expand_validity(ctx: *mut golap_ctx, bitmap_ptr: u64, bit_offset: u64, n_rows: u64, out_ptr: u64, stream: *mut c_void) -> i32;
    pub fn golap_pack_validity(ctx: *mut golap_ctx, validity_ptr: u64, n_rows: u64, out_bitmap_ptr: u64, out_null_count_ptr: u64, stream: *mut c_void) -> i32;
    pub fn golap_take_1byte(ctx: *mut golap_ctx, src_ptr: u64, indices_ptr: u64, count_ptr: u64, n_indices_max: u64, out_ptr: u64, stream: *mut c_void) -> i32;
    pub fn golap_pairs_unzip(ctx: *mut golap_ctx, pairs_ptr: u64, n_pairs: u64, out_build_idx_ptr: u64, out_probe_idx_ptr: u64, stream: *mut c_void) -> i32;
    pub fn golap_sort_u64(ctx: *mut golap_ctx, ptr: u64, n: u64, stream: *mut c_void) -> i32;
    pub fn golap_key_minmax(ctx: *mut golap_ctx, column_ptr: u64, validity_ptr: u64, n_rows: u64, out_ptr: u64, stream: *mut c_void) -> i32;
    pub fn golap_pack_group_keys(ctx: *mut golap_ctx, params: *const golap_pack_keys_params, stream: *mut c_void) -> i32;
    pub fn golap_hash_key_columns(ctx: *mut golap_ctx, params: *const golap_pack_keys_params, seed: u64, stream: *mut c_void) -> i32;
    pub fn golap_group_keys_verify(ctx: *mut golap_ctx, params: *const golap_pack_keys_params, slot_ids_ptr: u64, n_slots: u64, out_rep_rows_ptr: u64, out_mismatch_ptr: u64, stream: *mut c_void) -> i32;
    pub fn golap_device_copy(ctx: *mut golap_ctx, src_ptr: u64, dst_ptr: u64, bytes: u64, stream: *mut c_void) -> i32;
    pub fn golap_device_fill_u8(ctx: *mut golap_ctx, ptr: u64, value: u32, bytes: u64, stream: *mut c_void) -> i32;
    // ---- multi-GPU (no counterpart in the reference, README.md:414) ----
    pub fn golap_comm_unique_id(out_id: *mut u8) -> i32;
    pub fn golap_comm_create_rccl(ctx: *mut golap_ctx, id: *const u8, world: i32, rank: i32, out: *mut *mut golap_comm) -> i32;
    pub fn golap_comm_create_custom(ctx: *mut golap_ctx, transport: *const golap_transport, world: i32, rank: i32, out: *mut *mut golap_comm) -> i32;
    pub fn golap_comm_destroy(comm: *mut golap_comm);
    pub fn golap_sharded_join(ctx: *mut golap_ctx, comm: *mut golap_comm, params: *const golap_sharded_join_params, stream: *mut c_void) -> i32;
    pub fn golap_sharded_aggregate(ctx: *mut golap_ctx, comm: *mut golap_comm, specs: *const golap_aggregate_spec, n_specs: u32,
                                   params: *const golap_aggregate_params, stream: *mut c_void) -> i32;
    pub fn golap_sharded_join_info_get(comm: *mut golap_comm, out: *mut golap_sharded_join_info) -> i32;
    pub fn golap_sharded_plan_get(world: u32, n_build_total: u64, n_probe_total: u64, build_rows_per_rank: *const u64, probe_rows_per_rank: *const u64,
                                  max_msg_bytes: u64, out: *mut golap_sharded_plan) -> i32;
    pub fn golap_shard_range(n_total: u64, rank: u32, world: u32, lo: *mut u64, hi: *mut u64);
    pub fn golap_radix_partition(ctx: *mut golap_ctx, params: *const golap_partition_params, stream: *mut c_void) -> i32;
    pub fn golap_join_partitioned_begin(ctx: *mut golap_ctx, plan: *const golap_join_partitioned_plan, stream: *mut c_void) -> i32;
    pub fn golap_join_partitioned_side(ctx: *mut golap_ctx, which: i32, side: *const golap_partitioned_side, stream: *mut c_void) -> i32;
    pub fn golap_join_partitioned_finish(ctx: *mut golap_ctx, out_pairs_ptr: u64, out_count_ptr: u64, max_pairs: u64, stream: *mut c_void) -> i32;
    pub fn golap_join_partitioned_abort(ctx: *mut golap_ctx) -> i32;
    pub fn golap_join_plan_get(ctx: *mut golap_ctx, n_build: u64, n_owners: u32, out: *mut golap_join_plan) -> i32;
}

/// Maps a status code to the `anyhow::Result` the reference uses everywhere (nothing unwinds across the boundary).
pub fn check(ctx: *const golap_ctx, status: i32) -> anyhow::Result<()> {
    if status == GOLAP_OK {
        return Ok(());
    }
    let msg = unsafe { std::ffi::CStr::from_ptr(golap_last_error(ctx)) }.to_string_lossy().into_owned();
    anyhow::bail!("golap error {status}: {msg}")
}

pub mod wrappers;
