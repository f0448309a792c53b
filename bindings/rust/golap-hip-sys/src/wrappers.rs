//! The three wrapper bodies of gpu-kernel-compiler, re-written over the FFI: each replaces a stub that today logs and returns
//! `Ok(())` (join_kernel.rs:47-68, aggregate_kernel.rs:80-125, filter_kernel.rs:42-72).  Public names, argument structs and
//! `anyhow::Result` returns are the reference's; only `device: Arc<CudaDevice>` becomes the context handle.
//!
//! Not compiled in the authoring environment (no Rust toolchain).
use crate::*;
use std::ptr::null_mut;

/// join_kernel.rs:14-18 -- discriminants equal golap_join_strategy
#[repr(i32)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum JoinStrategy {
    RadixHashJoin = 0,
    SortMergeJoin = 1,
    BroadcastHashJoin = 2,
}

/// join_kernel.rs:20-27, field for field
#[derive(Clone, Copy, Debug)]
pub struct JoinParams {
    pub build_key_ptr: u64,
    pub probe_key_ptr: u64,
    pub n_build: u64,
    pub n_probe: u64,
    pub out_pairs_ptr: u64,
    pub out_count_ptr: u64,
}

pub struct JoinKernel {
    pub ctx: *mut golap_ctx,
}

impl JoinKernel {
    /// join_kernel.rs:39-45.  `max_pairs`: capacity of `out_pairs_ptr` in pairs (the count keeps counting past it,
    /// join_kernels.cuh:155-160); `join_type`: JoinType as i32 (logical_plan.rs:53-58).
    pub fn join(&self, strategy: JoinStrategy, params: JoinParams, max_pairs: u64, join_type: i32) -> anyhow::Result<()> {
        let p = golap_join_params {
            build_key_ptr: params.build_key_ptr,
            probe_key_ptr: params.probe_key_ptr,
            n_build: params.n_build,
            n_probe: params.n_probe,
            out_pairs_ptr: params.out_pairs_ptr,
            out_count_ptr: params.out_count_ptr,
            max_pairs,
            build_rid_base: 0,
            probe_rid_base: 0,
            join_type,
            n_key_cols: 1,
            build_key_extra: [0; 3],
            probe_key_extra: [0; 3],
            build_validity: [0; 4],
            probe_validity: [0; 4],
        };
        check(self.ctx, unsafe { golap_join(self.ctx, strategy as i32, &p, null_mut()) })
    }

    /// join_kernel.rs:71-77 (same threshold: <= 1M build rows -> Broadcast)
    pub fn choose_strategy(n_build: u64) -> JoinStrategy {
        match unsafe { golap_choose_strategy(n_build) } {
            2 => JoinStrategy::BroadcastHashJoin,
            1 => JoinStrategy::SortMergeJoin,
            _ => JoinStrategy::RadixHashJoin,
        }
    }
}

/// aggregate_kernel.rs:17-24 -- discriminants equal golap_agg_func
#[repr(i32)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum AggFunc {
    Sum = 0,
    Count = 1,
    Min = 2,
    Max = 3,
    Avg = 4,
}

/// aggregate_kernel.rs:38-44
#[derive(Clone, Copy, Debug)]
pub struct AggregateSpec {
    pub func: AggFunc,
    pub value_col_index: usize,
    pub output_col_index: usize,
}

/// aggregate_kernel.rs:47-58
#[derive(Clone, Debug)]
pub struct AggregateParams {
    pub group_key_ptr: u64,
    pub value_ptrs: Vec<u64>,
    pub n_rows: u64,
    pub n_groups: u32,
    pub out_ptr: u64,
}

pub struct AggregateKernel {
    pub ctx: *mut golap_ctx,
}

impl AggregateKernel {
    /// aggregate_kernel.rs:80-125.  `value_is_f64[i]`: dtype of value column i (the reference's params do not carry it).
    pub fn aggregate(&self, specs: &[AggregateSpec], params: AggregateParams, value_is_f64: &[bool]) -> anyhow::Result<()> {
        let s: Vec<golap_aggregate_spec> = specs
            .iter()
            .map(|a| golap_aggregate_spec { func: a.func as i32, value_col_index: a.value_col_index as u64, output_col_index: a.output_col_index as u64 })
            .collect();
        let dt: Vec<i32> = value_is_f64.iter().map(|&f| if f { 1 } else { 0 }).collect();
        let p = golap_aggregate_params {
            group_key_ptr: params.group_key_ptr,
            value_ptrs: params.value_ptrs.as_ptr(),
            n_value_cols: params.value_ptrs.len() as u32,
            n_rows: params.n_rows,
            n_groups: params.n_groups,
            out_ptr: params.out_ptr,
            key_dtype: 0,
            value_dtypes: dt.as_ptr(),
            out_count_ptr: 0,
            key_mode: 0,
            reserved0: 0,
            out_keys_ptr: 0,
            out_n_groups_ptr: 0,
            value_validity_ptrs: std::ptr::null(),
            out_slot_ids_ptr: 0,
        };
        check(self.ctx, unsafe { golap_aggregate(self.ctx, s.as_ptr(), s.len() as u32, &p, null_mut()) })
    }
}

/// codegen.rs:11-21 -- discriminants equal golap_cmp_op
#[repr(i32)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum CmpOp {
    Eq = 0,
    Ne = 1,
    Lt = 2,
    Le = 3,
    Gt = 4,
    Ge = 5,
}

/// codegen.rs:24-28
#[derive(Clone, Copy, Debug)]
pub struct FilterPredicate {
    pub col_index: usize,
    pub op: CmpOp,
    pub literal: i64,
}

/// filter_kernel.rs:17-28
#[derive(Clone, Copy, Debug)]
pub struct FilterParams {
    pub column_ptr: u64,
    pub n_rows: u64,
    pub out_mask_ptr: u64,
    pub grid_dim: u32,
    pub block_dim: u32,
}

pub struct FilterKernel {
    pub ctx: *mut golap_ctx,
}

impl FilterKernel {
    /// filter_kernel.rs:42-72.  The `ptx: &str` argument of the reference (KernelCodegen::emit_filter_ptx, codegen.rs:36-47, whose body is
    /// `ret;`) is gone: kernels are ahead-of-time gfx950 code objects and the predicate travels as data.
    pub fn launch(&self, pred: &FilterPredicate, params: &FilterParams) -> anyhow::Result<()> {
        let fp = golap_filter_predicate { col_index: pred.col_index as u64, op: pred.op as i32, literal: pred.literal };
        let p = golap_filter_params {
            column_ptr: params.column_ptr,
            n_rows: params.n_rows,
            out_mask_ptr: params.out_mask_ptr,
            grid_dim: params.grid_dim,
            block_dim: params.block_dim,
            validity_ptr: 0,
        };
        check(self.ctx, unsafe { golap_filter_launch(self.ctx, &fp, &p, null_mut()) })
    }

    /// filter_kernel.rs:75-79
    pub fn compute_launch_params(n_rows: u64) -> (u32, u32) {
        let (mut g, mut b) = (0u32, 0u32);
        unsafe { golap_compute_launch_params(n_rows, &mut g, &mut b) };
        (g, b)
    }
}
