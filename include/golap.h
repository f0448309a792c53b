/* golap.h -- C ABI of the MI355X-native GpuHashJoin / GpuAggregate / GpuFilter path.
 *
 * Drop-in boundary for codebasecomprehension987/gpu-olap-engine: every entry
 * point below replaces one interface of the reference's (never-launched) kernel
 * wrapper / memory-manager crates; the reference file:line it replaces is cited
 * on each declaration (paths relative to the reference root).  A Rust host binds
 * this header with a plain `extern "C"` block.
 *
 * Conventions
 *   - plain pointers / sizes only; device addresses travel as uint64_t exactly
 *     like the reference's param structs (join_kernel.rs:20-27).
 *   - every call returns int32 status (GOLAP_OK == 0); golap_last_error() gives
 *     the message.  Nothing throws or aborts across the ABI (the reference uses
 *     anyhow::Result everywhere).
 *   - the caller owns input and output buffers; the library owns only its
 *     context, slabs and scratch.
 *   - `stream` is a hipStream_t passed as void*; NULL is the HIP null stream, as in
 *     every HIP API.  Calls are asynchronous on that stream unless stated otherwise.
 *   - there is NO CPU fallback: without a HIP device every compute call fails
 *     with GOLAP_ERR_NO_DEVICE.
 */
#ifndef GOLAP_H
#define GOLAP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GOLAP_ABI_VERSION 4

/* ---- status codes ------------------------------------------------------- */
enum {
    GOLAP_OK = 0,
    GOLAP_ERR_INVALID_ARG = 1,
    GOLAP_ERR_NO_DEVICE = 2,
    GOLAP_ERR_HIP = 3,
    GOLAP_ERR_OOM = 4,          /* slab class exhausted / size above largest class */
    GOLAP_ERR_UNSUPPORTED = 5,  /* e.g. SortMergeJoin: out of scope */
    GOLAP_ERR_CAPACITY = 6,     /* informational: join produced more pairs than max_pairs */
    GOLAP_ERR_PEER = 7          /* collective call: ANOTHER rank failed before the exchange; every rank has left the call together */
};

typedef struct golap_ctx golap_ctx;

/* ---- context / config ---------------------------------------------------
 * Mirrors EngineConfig, gpu-olap-core/src/lib.rs:20-31 (defaults :33-43),
 * plus the device/radix fields the survey adds (SURVEY 8(b) "Config"). */
typedef struct golap_engine_config {
    uint64_t max_gpu_memory;     /* lib.rs:22  default 8 GiB (0 = no cap)       */
    uint32_t num_streams;        /* lib.rs:24  default 8                        */
    uint32_t use_unified_memory; /* lib.rs:26  default 1 (accepted, unused)     */
    uint64_t batch_size;         /* lib.rs:28  default 1 Mi rows                */
    uint32_t enable_cache;       /* lib.rs:30  default 1 (accepted, unused)     */
    int32_t  device_id;          /* HIP ordinal; -1 = current device            */
    uint32_t radix_bits;         /* join_kernels.cuh:22  RADIX_BITS = 8         */
    uint32_t enable_profiling;   /* 1 = record per-kernel hipEvents (golap_stats); 2 = the same, but the stand-by launches of a join (the 64-bit
                                    pipeline behind the tuple pipeline: ~17 launches that return at once) are only timed while a recent join
                                    needed them -- a pair of timing events costs as much stream time as the launch it brackets */
    uint32_t max_slabs_per_class;/* slab_allocator.rs:25-28 (constructor argument); default 4096: in practice only max_gpu_memory bounds the pool */
    uint32_t provision_sigmas;   /* standard deviations of head-room in the provisioned (no-histogram) layouts; 0 = default 6.
                                    Tests set 1 / GOLAP_PROVISION_NONE to force the overflow fall-backs; results never depend on it. */
} golap_engine_config;
#define GOLAP_PROVISION_NONE 0xFFFFFFFFu   /* provision_sigmas: no head-room at all */

void    golap_engine_config_default(golap_engine_config* cfg);
int32_t golap_ctx_create(const golap_engine_config* cfg, golap_ctx** out_ctx);
void    golap_ctx_destroy(golap_ctx* ctx);
/* ctx may be NULL: returns the calling thread's last error. */
const char* golap_last_error(const golap_ctx* ctx);
uint32_t golap_abi_version(void);

/* Developer / test knobs: named strings that select experiment variants or force rarely taken device paths (the names are
 * listed in scripts/README.md; results never depend on them).  value NULL removes a knob.  The library does not read the
 * process environment on any compute path; GOLAP_* variables are imported once at golap_ctx_create when GOLAP_DEBUG_ENV is set. */
int32_t golap_debug_set(golap_ctx* ctx, const char* name, const char* value);

typedef struct golap_device_info {
    char     name[128];
    char     gcn_arch[64];
    uint32_t compute_units;
    uint32_t lds_bytes_per_block;
    uint32_t wavefront_size;
    uint32_t clock_mhz;
    uint32_t memory_clock_mhz;
    uint32_t memory_bus_bits;
    uint64_t total_mem_bytes;
    uint64_t free_mem_bytes;
    uint32_t l2_bytes;
    /* ABI 4: which form of the five hand-scheduled kernels this library launches by default.  The forms with hidden (inline-asm) memory
     * operations and counted waits were audited -- statically on the generated code and on a GPU against the plain forms -- for ONE
     * hipcc (audited_with); a library built by any other hipcc (built_with) launches the plain forms: hidden_memory_forms = 0. */
    uint32_t hidden_memory_forms;
    char     built_with[32];     /* "hip 7.2.26015" */
    char     audited_with[32];
} golap_device_info;
int32_t golap_device_info_get(golap_ctx* ctx, golap_device_info* out);
int32_t golap_ctx_synchronize(golap_ctx* ctx, void* stream);

/* ---- SlabAllocator -------------------------------------------------------
 * gpu-memory-manager/src/slab_allocator.rs: new :25-47, allocate :50-68,
 * free :71-93, find_slab_class :95-107.  Size classes 1/4/16/64/256 MiB
 * (README.md:142) extended with 1/4/16/64 GiB classes so an 8 GB column fits
 * (the reference's top class would reject it).  Unlike the reference
 * (which leaks, :146) slabs beyond the pool limit are really hipFree'd. */
int32_t golap_slab_alloc(golap_ctx* ctx, uint64_t size, uint64_t* out_ptr, uint32_t* out_class_idx);
int32_t golap_slab_free(golap_ctx* ctx, uint64_t ptr, uint32_t class_idx);
typedef struct golap_slab_stats {
    uint32_t n_classes;
    uint64_t class_size[16];
    uint32_t allocated[16];   /* live + pooled, slab_allocator.rs:21 allocated_count */
    uint32_t pooled[16];      /* free_slabs.len()                                     */
    uint64_t bytes_reserved;
} golap_slab_stats;
int32_t golap_slab_get_stats(golap_ctx* ctx, golap_slab_stats* out);

/* ---- TransferQueue -------------------------------------------------------
 * gpu-memory-manager/src/transfer_queue.rs: TransferDirection :10-13,
 * enqueue_transfer :55-79 (stream_id % num_streams, :72), synchronize :133-139.
 * Really copies (hipMemcpyAsync); pageable host memory is staged through two
 * pinned bounce buffers per stream (double buffering). */
enum { GOLAP_HOST_TO_DEVICE = 0, GOLAP_DEVICE_TO_HOST = 1 };
int32_t golap_xfer_enqueue(golap_ctx* ctx, int32_t direction, uint64_t src, uint64_t dst, uint64_t size, uint32_t stream_id);
int32_t golap_xfer_synchronize(golap_ctx* ctx);

/* ---- GpuFilter -----------------------------------------------------------
 * CmpOp / FilterPredicate: gpu-kernel-compiler/src/codegen.rs:11-28.
 * FilterParams: filter_kernel.rs:17-28.  The NVRTC/PTX string of codegen.rs:36-47
 * is replaced by ahead-of-time gfx950 kernels with `op` as a runtime argument. */
typedef enum golap_cmp_op {
    GOLAP_CMP_EQ = 0, GOLAP_CMP_NE = 1, GOLAP_CMP_LT = 2,
    GOLAP_CMP_LE = 3, GOLAP_CMP_GT = 4, GOLAP_CMP_GE = 5,
    GOLAP_CMP_NEVER = 6     /* executor.rs:464: unknown operator selects no row */
} golap_cmp_op;
/* "=", "==", "!=", "<>", ">", ">=", "<", "<=" as executor.rs:457-465; else NEVER. */
golap_cmp_op golap_parse_cmp_op(const char* op);

typedef struct golap_filter_predicate {   /* codegen.rs:24-28 */
    uint64_t col_index;
    int32_t  op;          /* golap_cmp_op */
    int64_t  literal;
} golap_filter_predicate;

typedef struct golap_filter_params {      /* filter_kernel.rs:17-28 */
    uint64_t column_ptr;    /* device, int64[n_rows] */
    uint64_t n_rows;
    uint64_t out_mask_ptr;  /* device, u8[n_rows], 1 = pass */
    uint32_t grid_dim;      /* accepted for API parity; 0 = library chooses */
    uint32_t block_dim;
    uint64_t validity_ptr;  /* extension: device u8[n_rows], 0 = NULL (ColumnBuffer.validity, column_buffer.rs:59-60); 0 pointer = no NULLs.
                               A comparison with NULL is not true: the row does not pass (the reference's ingest turns NULL into 0,
                               record_batch_convert.rs:102-110, which would let it pass `col < 10`). */
} golap_filter_params;

/* filter_kernel.rs:75-79 (and aggregate_kernel.rs:127-131): block 256, grid ceil(n/256) >= 1 */
void    golap_compute_launch_params(uint64_t n_rows, uint32_t* grid, uint32_t* block);
/* FilterKernel::launch, filter_kernel.rs:42-72: writes the u8 mask. */
int32_t golap_filter_launch(golap_ctx* ctx, const golap_filter_predicate* pred, const golap_filter_params* params, void* stream);
/* The "second compaction pass (prefix-sum + scatter)" of filter_kernel.rs:3-5,
 * fused with the predicate: writes the ascending row indices of passing rows
 * (original order, executor.rs:479-484) and their count (u64, device). */
int32_t golap_filter_compact(golap_ctx* ctx, const golap_filter_predicate* pred, uint64_t column_ptr, uint64_t n_rows,
                             uint64_t out_indices_ptr /*u32[n_rows]*/, uint64_t out_count_ptr /*u64*/, void* stream);
/* The same with a validity byte per row (0 = NULL; 0 pointer = no NULLs): NULL rows never pass. */
int32_t golap_filter_compact_nullable(golap_ctx* ctx, const golap_filter_predicate* pred, uint64_t column_ptr, uint64_t validity_ptr, uint64_t n_rows,
                                      uint64_t out_indices_ptr, uint64_t out_count_ptr, void* stream);
/* Conjunction of up to GOLAP_FILTER_AND_MAX comparisons, each on its own int64 column (the same column may appear twice: a range), in
 * ONE pass: a row passes when every comparison is true (a NULL operand makes a comparison not true).  optimizer.rs:155-159 merges
 * stacked filters into `a AND b`; the reference's CPU mask has no AND (executor.rs:457-465 -> no row): SQL semantics here are an
 * extension.  preds[k].col_index is informational; column_ptrs[k] / validity_ptrs[k] (0 = no NULLs; the array may be NULL) are
 * host arrays of device pointers. */
#define GOLAP_FILTER_AND_MAX 4
int32_t golap_filter_compact_and(golap_ctx* ctx, const golap_filter_predicate* preds, const uint64_t* column_ptrs, const uint64_t* validity_ptrs,
                                 uint32_t n_preds, uint64_t n_rows, uint64_t out_indices_ptr /*u32[n_rows]*/, uint64_t out_count_ptr /*u64*/, void* stream);
/* take_int64, executor.rs:496-507: out[i] = col[indices[i]] for 8-byte columns (index 0xFFFFFFFF = NULL row -> 0). */
int32_t golap_take_8byte(golap_ctx* ctx, uint64_t column_ptr, uint64_t indices_ptr, uint64_t count_ptr /*u64 device, or 0*/,
                         uint64_t n_indices_max, uint64_t out_ptr, void* stream);
/* BASELINE config 1 shape: execute_filter (executor.rs:161-178) feeding
 * execute_aggregate Sum (executor.rs:297-355) in one pass:
 * out[0] = f64 sum of v[i] (value_is_f64 ? f64 : (f64)i64) over rows with k[i] <op> literal,
 * out[1] = (f64) number of kept rows. */
int32_t golap_filter_sum(golap_ctx* ctx, const golap_filter_predicate* pred, uint64_t key_column_ptr, uint64_t value_column_ptr,
                         int32_t value_is_f64, uint64_t n_rows, uint64_t out_ptr /*f64[2] device*/, void* stream);

/* NULL-aware form: rows whose key is NULL do not pass, NULL values are skipped by the sum and the count (SQL aggregates). */
int32_t golap_filter_sum_nullable(golap_ctx* ctx, const golap_filter_predicate* pred, uint64_t key_column_ptr, uint64_t key_validity_ptr,
                                  uint64_t value_column_ptr, uint64_t value_validity_ptr, int32_t value_is_f64, uint64_t n_rows,
                                  uint64_t out_ptr /*f64[2] device*/, void* stream);

/* ---- GpuHashJoin ---------------------------------------------------------
 * JoinStrategy join_kernel.rs:14-18, JoinParams :20-27, JoinKernel::join :39-45,
 * choose_strategy :71-77.  Device arithmetic: join_kernels.cuh:26-41 (hash),
 * :45-76 (radix partition), :80-111 (build), :115-166 (probe). */
typedef enum golap_join_strategy {
    GOLAP_JOIN_RADIX_HASH = 0,
    GOLAP_JOIN_SORT_MERGE = 1,      /* returns GOLAP_ERR_UNSUPPORTED (out of scope) */
    GOLAP_JOIN_BROADCAST_HASH = 2
} golap_join_strategy;
golap_join_strategy golap_choose_strategy(uint64_t n_build);   /* join_kernel.rs:71-77 */

typedef struct golap_join_params {
    /* --- join_kernel.rs:20-27, same names and meaning --- */
    uint64_t build_key_ptr;   /* device int64[n_build]  (left side, executor.rs:241) */
    uint64_t probe_key_ptr;   /* device int64[n_probe]                               */
    uint64_t n_build;
    uint64_t n_probe;
    uint64_t out_pairs_ptr;   /* device u32[2*max_pairs]: (build_row_id, probe_row_id) interleaved,
                                 = left_matches/right_matches of join_kernels.cuh:125-126 */
    uint64_t out_count_ptr;   /* device u64: TRUE match count; keeps counting past max_pairs
                                 (join_kernels.cuh:155-160); zeroed by the call            */
    /* --- extensions --- */
    uint64_t max_pairs;       /* capacity of out_pairs in pairs (join_kernels.cuh:127 max_matches) */
    uint32_t build_rid_base;  /* global row id of build row 0 (executor.rs:249 running offset)     */
    uint32_t probe_rid_base;
    int32_t  join_type;       /* JoinType, logical_plan.rs:53-58: 0 Inner, 1 Left (every build row kept),
                                 2 Right (every probe row kept), 3 Full; the missing side of an unmatched
                                 row is row id 0xFFFFFFFF.  The reference's kernels are inner-only.  */
    /* --- composite keys and NULLs (ABI 2) --- */
    uint32_t n_key_cols;      /* 0 / 1: one key column per side (build_key_ptr / probe_key_ptr).  2..4: composite key -- physical_plan.rs:35-36
                                 declares left_keys / right_keys as Vec<PhysicalExpr>, the reference executor only ever reads [0]
                                 (executor.rs:245) -- the further columns are *_key_extra[0..n_key_cols-2]; rows match when EVERY column is equal */
    uint64_t build_key_extra[3];   /* device int64[n_build] */
    uint64_t probe_key_extra[3];   /* device int64[n_probe] */
    uint64_t build_validity[4];    /* per key column: device u8[n_build], 0 = NULL -- the byte-per-row validity of ColumnBuffer
                                      (column_buffer.rs:59-60); a 0 pointer = that column has no NULLs.  SQL semantics (README.md:411
                                      "null handling" is on the reference's roadmap): a row with a NULL in any key column matches nothing,
                                      and on the preserved side of an outer join it survives as an unmatched row */
    uint64_t probe_validity[4];
    /* (Composite and NULL-aware joins run on 64-bit surrogate keys and verify every pair against the real columns; up to 64
     * colliding candidate pairs -- expected 0.05 in a 1B x 1B join -- are repaired on the device.  Beyond that the count word is
     * set to 2^64-1: an impossible count, never observed outside the test hook that narrows the surrogates.) */
} golap_join_params;

/* Join on int64 keys; row ids are 0-based positions + *_rid_base.  Pair
 * order is unspecified (as in the reference, join_kernels.cuh:155); compare
 * after sorting by (probe_row_id, build_row_id).  Any int64 key is accepted;
 * when every key of a large join lies in [0, 2^32) the partitioned rows travel
 * as 8-byte (key32, row id) tuples -- decided on the device, invisible here. */
int32_t golap_join(golap_ctx* ctx, golap_join_strategy strategy, const golap_join_params* params, void* stream);

/* Scratch bytes golap_join(RadixHashJoin) will take from the slab allocator. */
uint64_t golap_join_scratch_bytes(golap_ctx* ctx, uint64_t n_build, uint64_t n_probe);

/* radix_partition_kernel, join_kernels.cuh:45-76, as a stand-alone step (pass 1
 * of the radix join; also what each GPU runs before the xGMI partition exchange).
 * Partition id = (hash_int64(key) >> 0) & 255 (:69).  Writes keys/row-ids grouped
 * by partition and the 257 exclusive offsets.  With n_owners > 1 the partitions
 * are laid out owner-major: partition p (owner p % n_owners) sits at slot
 * (p % n_owners) * (256 / n_owners) + p / n_owners, so the rows bound for each
 * peer GPU are one contiguous range.  out_offsets_ptr[s] is indexed by slot. */
typedef struct golap_partition_params {
    uint64_t key_ptr;          /* device int64[n_rows] */
    uint64_t row_id_ptr;       /* device u32[n_rows] or 0 -> row id = rid_base + i (join_kernels.cuh:75) */
    uint64_t n_rows;
    uint32_t rid_base;
    uint32_t n_owners;         /* 1, 2, 4, 8, ... divides 256 */
    uint32_t extra_bits;       /* 0..2: refine every partition by this many further hash bits (bucket = slot << extra_bits | bits);
                                  out_offsets then has (256 << extra_bits) + 1 entries */
    uint64_t out_key_ptr;      /* device int64[n_rows] (narrow: the (key32, row id) tuples, 8 bytes per row) */
    uint64_t out_row_id_ptr;   /* device u32[n_rows]   (narrow: unused, may be 0) */
    uint64_t out_offsets_ptr;  /* device u32[(256 << extra_bits) + 1] */
    /* narrow rows for the partition exchange: when `narrow` != 0 the partitioned rows are written as ONE array of 8-byte
     * (key32, row id) tuples (x = low 32 key bits, y = row id) -- a third less to send over xGMI.  Only valid when every key
     * lies in [0, 2^32): the call sets *wide_flag_ptr (device u32, zeroed by the caller) to 1 otherwise, and the caller must
     * then discard the output and partition again with narrow = 0.  Needs row_id_ptr == 0. */
    uint32_t narrow;
    uint32_t reserved0;
    uint64_t wide_flag_ptr;
} golap_partition_params;
int32_t golap_radix_partition(golap_ctx* ctx, const golap_partition_params* params, void* stream);

/* Join of already radix-partitioned inputs (after the partition exchange).
 * Each side is described by n_slots * n_sources segments: segment (slot s,
 * source r) holds the rows of local partition slot s that came from source r,
 * seg_off/seg_len in rows, row-major [slot][source] HOST arrays. */
typedef struct golap_partitioned_side {
    uint64_t key_ptr;      /* device int64[] */
    uint64_t row_id_ptr;   /* device u32[]   */
    uint64_t n_rows;       /* total rows in the buffers */
    const uint64_t* seg_off;   /* host [n_slots*n_sources] */
    const uint64_t* seg_len;   /* host [n_slots*n_sources] */
} golap_partitioned_side;
typedef struct golap_join_partitioned_params {
    golap_partitioned_side build, probe;
    uint32_t n_slots;      /* local groups: (256 / n_owners) << extra_bits */
    uint32_t n_sources;
    uint32_t pre_bits;     /* hash bits the exchange-time partition consumed: 8 + extra_bits (0 = 8) */
    uint64_t total_build_rows_hint;  /* global build rows / n_owners: sizes the sub-partition fan-out */
    uint64_t out_pairs_ptr, out_count_ptr, max_pairs;
} golap_join_partitioned_params;
int32_t golap_join_partitioned(golap_ctx* ctx, const golap_join_partitioned_params* params, void* stream);

/* The same join in three steps, so that the partition exchange of one side can overlap the device work of
 * the other: begin (plan + workspace), side (local sub-partitioning of the build (0) or probe (1) rows, may be
 * called as soon as that side's rows have arrived), finish (LDS build/probe, pair output).
 * Between begin and finish the context's scratch workspace belongs to this join: every other compute call on the
 * same context (golap_join, golap_radix_partition, golap_aggregate, golap_filter_*) fails with GOLAP_ERR_INVALID_ARG
 * instead of overwriting the partitioned rows -- partition both sides BEFORE begin, or use a second context.
 * golap_join_partitioned_abort releases the workspace without joining. */
typedef struct golap_join_partitioned_plan {
    uint32_t n_groups;     /* (256 / n_owners) << extra_bits */
    uint32_t n_sources;
    uint32_t pre_bits;     /* 8 + extra_bits */
    uint64_t total_build_rows_hint;
    uint64_t build_rows, probe_rows;   /* rows this rank holds after the exchange */
    uint32_t narrow;       /* != 0: both sides arrive as (key32, row id) tuples in key_ptr (golap_partition_params.narrow); row_id_ptr unused */
    uint32_t reserved0;
} golap_join_partitioned_plan;
/* 1 when the local passes this plan needs exist in tuple form (host-only; ask before partitioning with narrow = 1). */
int32_t golap_join_partitioned_narrow_ok(const golap_join_partitioned_plan* plan);
int32_t golap_join_partitioned_begin(golap_ctx* ctx, const golap_join_partitioned_plan* plan, void* stream);
int32_t golap_join_partitioned_side(golap_ctx* ctx, int32_t which, const golap_partitioned_side* side, void* stream);
int32_t golap_join_partitioned_finish(golap_ctx* ctx, uint64_t out_pairs_ptr, uint64_t out_count_ptr, uint64_t max_pairs, void* stream);
int32_t golap_join_partitioned_abort(golap_ctx* ctx);

/* ---- multi-GPU: one process per GPU, partition exchange over xGMI ----------------------------
 * The reference has no distributed layer (README.md:414); north_star asks for the 256 radix partitions sharded over the
 * GPUs of one node with an RCCL all-to-all.  Rows of both tables start range-sharded with GLOBAL u32 row ids; every rank
 * radix-partitions its rows (join_kernels.cuh:69, partition p -> rank p % P, owner-major so each peer's rows are one
 * slice), the slices are exchanged, every rank joins its own partitions; output pairs stay sharded.
 *
 * golap_comm binds a context to a communicator: the built-in RCCL one (librccl is dlopen'ed; the host distributes the
 * 128-byte unique id any way it likes) or host-supplied transport callbacks (other fabrics; the tests' gloo rehearsal).
 * golap_sharded_join is collective: every rank of the communicator calls it with the same totals.  Fast path (large
 * joins whose keys fit 8-byte (hash, row id) tuples: all in [0, 2^32), or -- agreed on across the ranks from samples that ride in the
 * agreement round -- one common upper half, a window of consecutive upper halves as wide as the partition's fan-out, or up to four
 * far-apart ones): no histogram and no size exchange -- every bucket owns a fixed, provisioned
 * range, messages have sizes both ends can compute, the per-bucket row counts travel device-to-device (all-gather) and
 * only reach the kernels; the host synchronises twice per join: on the agreement word (did every rank get through its argument
 * checks and allocations? -- read while the first partition kernel runs) and on the status words at the end.  A rank's rows are
 * partitioned and shipped in chunks, so the exchange runs beside the partition pass from the first chunk on.  Anything else (upper key
 * halves beyond that, a skewed bucket, small inputs) takes the exact path: histogram partition, size all-gather read back by the
 * host, variable-size exchange, segment-fed local passes.  All ranks take the same path (the status word is gathered).
 * Failures: whatever can fail on the host before the exchange (arguments, plan, workspace) is agreed on first -- if any rank
 * failed, the failing rank returns its error and every other rank GOLAP_ERR_PEER, nobody waits in a collective.  A HIP error in
 * the middle of the exchange is as fatal as it is for any RCCL program (peers block until the communicator is destroyed). */
typedef struct golap_comm golap_comm;
typedef struct golap_p2p_op {
    int32_t  peer;       /* rank in the communicator */
    int32_t  is_send;    /* 1 = send, 0 = receive */
    uint64_t ptr;        /* device address */
    uint64_t bytes;
} golap_p2p_op;
typedef struct golap_transport {
    void* user;
    /* every rank contributes bytes_per_rank from send_ptr; recv_ptr receives world * bytes_per_rank, rank-major (device memory) */
    int32_t (*all_gather)(void* user, uint64_t send_ptr, uint64_t recv_ptr, uint64_t bytes_per_rank, void* stream);
    /* one group of point-to-point transfers; between two ranks sends and receives match in list order */
    int32_t (*exchange)(void* user, const golap_p2p_op* ops, uint32_t n_ops, void* stream);
} golap_transport;
#define GOLAP_COMM_ID_BYTES 128
int32_t golap_comm_unique_id(uint8_t out_id[GOLAP_COMM_ID_BYTES]);                     /* ncclGetUniqueId (rank 0, then broadcast by the host) */
int32_t golap_comm_create_rccl(golap_ctx* ctx, const uint8_t id[GOLAP_COMM_ID_BYTES], int32_t world, int32_t rank, golap_comm** out);
int32_t golap_comm_create_custom(golap_ctx* ctx, const golap_transport* transport, int32_t world, int32_t rank, golap_comm** out);
void    golap_comm_destroy(golap_comm* comm);   /* before golap_ctx_destroy of the context it was created on */

typedef struct golap_sharded_join_params {
    uint64_t build_key_ptr, probe_key_ptr;       /* device int64: this rank's rows */
    uint64_t n_build_local, n_probe_local;
    uint32_t build_rid_base, probe_rid_base;     /* global row id of local row 0 */
    uint64_t n_build_total, n_probe_total;       /* over all ranks */
    const uint64_t* build_rows_per_rank;         /* host [world] or NULL = range sharding (golap_shard_range) */
    const uint64_t* probe_rows_per_rank;
    uint64_t out_pairs_ptr, out_count_ptr, max_pairs;   /* as golap_join_params; pairs carry global row ids */
    int32_t  join_type;                          /* as golap_join_params */
    int32_t  force_exact;                        /* != 0: skip the fast path */
    uint64_t max_msg_bytes;                      /* 0 = 512 MiB: larger messages are split (both ends compute the same pieces); the built-in RCCL transport caps it at 1 GiB */
    /* Composite keys and NULLs, as golap_join_params (physical_plan.rs:35-36, column_buffer.rs:59-60): up to 4 key columns per side
     * (column 0 = build_key_ptr / probe_key_ptr) and one validity byte array per column (0 pointer = no NULLs; a row with a NULL in any
     * key column matches nothing, an outer join keeps it as an unmatched row).  The ranks agree on every column's value range
     * (one small all-gather, read by the host) and join on the exact mixed-radix number of the columns' digits -- no hashing, no
     * collisions.  A tuple wider than 62 bits (hash-like ids in several columns; ONE full-range column with NULLs) cannot be packed:
     * the ranks then join on a 64-bit surrogate of the tuple and ship the real key columns with the rows; the owner of a partition
     * checks every pair against them before it emits global row ids (exact path; ABI 4 -- ABI 3 refused such keys).
     * Validity arrays may be given on some ranks only (a shard without NULLs has none): the call's first agreement round settles
     * the form for all ranks. */
    uint32_t n_key_cols;                         /* 0 or 1: one key column */
    uint32_t reserved0;
    uint64_t build_key_extra[3], probe_key_extra[3];
    uint64_t build_validity[4], probe_validity[4];
} golap_sharded_join_params;
/* Stages of the fast path, timed on the device when the context has profiling on (begin and end in ms since the call's first
 * kernel; events on the compute stream and on the communicator's stream): what a scaling line needs to say where the time went. */
enum { GOLAP_SHARD_ST_PART_BUILD = 0, GOLAP_SHARD_ST_PART_PROBE = 1, GOLAP_SHARD_ST_XCHG_BUILD = 2, GOLAP_SHARD_ST_XCHG_PROBE = 3,
       GOLAP_SHARD_ST_RING_BUILD = 4, GOLAP_SHARD_ST_RING_PROBE = 5, GOLAP_SHARD_ST_JOIN = 6, GOLAP_SHARD_ST_STATUS = 7, GOLAP_SHARD_STAGES = 8 };
typedef struct golap_sharded_join_info {          /* what the last golap_sharded_join on this communicator did */
    uint32_t fast_path;          /* 1: the no-histogram tuple path produced the result */
    uint32_t fast_path_abandoned;/* 1: it was tried and its status word sent every rank to the exact path */
    uint32_t extra_bits, n_groups, local_bits;
    uint32_t host_syncs;         /* host synchronisations inside the call (fast path: the agreement word, hidden behind the first partition
                                  * kernel, and the status words at the end) */
    uint64_t rows_received[2];   /* build, probe (exact path; fast path: provisioned rows) */
    uint64_t bytes_sent[2];
    uint32_t chunks;             /* fast path: row chunks per side; chunk k's messages travel while chunk k + 1 is partitioned */
    uint32_t sub_ranges;         /* fast path: per-XCD sub-ranges per bucket */
    float    stage_ms[GOLAP_SHARD_STAGES][2];   /* [stage][begin, end]; zeros without profiling or on the exact path */
    float    total_ms;           /* first kernel -> status words on the host's side of the last copy */
    uint32_t halves;             /* fast path: the probe side's rows travel in this many group halves (1 or 2); ring pass and join of a half run
                                  * while the next one is on the wire */
} golap_sharded_join_info;
int32_t golap_sharded_join(golap_ctx* ctx, golap_comm* comm, const golap_sharded_join_params* params, void* stream);
int32_t golap_sharded_join_info_get(golap_comm* comm, golap_sharded_join_info* out);
/* Diagnostic: ONE unsplit message of `bytes` bytes from this rank to itself through the transport, verified on the device:
 * *out_bad_words = 4-byte words that arrived wrong (first one in *out_first_bad_word).  golap_sharded_join never sends more
 * than max_msg_bytes in one message; this pins where larger ones stop working on a given RCCL build. */
int32_t golap_comm_selftest(golap_ctx* ctx, golap_comm* comm, uint64_t bytes, uint64_t* out_bad_words, uint64_t* out_first_bad_word);
/* rank g of `world` holds rows [lo, hi) of n_total (last rank takes the remainder) */
void    golap_shard_range(uint64_t n_total, uint32_t rank, uint32_t world, uint64_t* lo, uint64_t* hi);
/* Host-only description of the fast path's messages (every rank computes the same numbers; no device needed):
 * rows provisioned per bucket by each source rank and the pieces of every (source, destination) message. */
typedef struct golap_sharded_plan {
    uint32_t eligible;           /* 1: the fast path applies to these sizes */
    uint32_t extra_bits, n_buckets, buckets_per_rank, local_bits;
    uint32_t sub_ranges;         /* every bucket is sub_ranges ranges of cap_rows rows (one per XCD of the partitioning GPU) */
    uint32_t chunks;             /* row chunks per side: a rank partitions and ships its rows in `chunks` pieces (all but the last equally long) */
    uint64_t cap_rows[2][64];    /* [side][source rank] rows provisioned per bucket sub-range of chunks 0 .. chunks - 2 */
    uint64_t cap_rows_last[2][64];   /* the same for the last chunk (the remainder) */
    uint64_t msg_bytes[2][64];   /* [side][source rank] bytes every peer receives from that rank (all chunks) */
    uint32_t msg_pieces[2][64];  /* messages that makes (chunks, each split at max_msg_bytes) */
} golap_sharded_plan;
int32_t golap_sharded_plan_get(uint32_t world, uint64_t n_build_total, uint64_t n_probe_total, const uint64_t* build_rows_per_rank,
                               const uint64_t* probe_rows_per_rank, uint64_t max_msg_bytes, golap_sharded_plan* out);

/* ---- GpuAggregate --------------------------------------------------------
 * AggFunc + identity_value aggregate_kernel.rs:17-34, AggregateSpec :38-44,
 * AggregateParams :47-58, AggregateKernel::aggregate :80-125. */
typedef enum golap_agg_func {
    GOLAP_AGG_SUM = 0, GOLAP_AGG_COUNT = 1, GOLAP_AGG_MIN = 2, GOLAP_AGG_MAX = 3, GOLAP_AGG_AVG = 4
} golap_agg_func;
int64_t golap_agg_identity_value(int32_t func);     /* aggregate_kernel.rs:27-33 */

typedef struct golap_aggregate_spec {               /* aggregate_kernel.rs:38-44 */
    int32_t  func;               /* golap_agg_func */
    uint64_t value_col_index;    /* index into value_ptrs */
    uint64_t output_col_index;   /* which [agg] plane of out_ptr */
} golap_aggregate_spec;

enum { GOLAP_DT_INT64 = 0, GOLAP_DT_FLOAT64 = 1, GOLAP_DT_INT32 = 2 };

typedef struct golap_aggregate_params {             /* aggregate_kernel.rs:47-58 */
    uint64_t group_key_ptr;      /* device; int64 (reference layout, :48) or int32 (BASELINE config 4); 0 = ungrouped */
    const uint64_t* value_ptrs;  /* host array of device pointers, one per value column */
    uint32_t n_value_cols;
    uint64_t n_rows;
    uint32_t n_groups;           /* upper bound on distinct groups (:54); keys must lie in [0, n_groups) (dense) */
    uint64_t out_ptr;            /* device f64[n_specs][n_groups] : one 8-byte slot per group per agg (:56, :98) */
    /* --- extensions --- */
    int32_t  key_dtype;          /* GOLAP_DT_INT64 | GOLAP_DT_INT32 */
    const int32_t* value_dtypes; /* host, per value column: GOLAP_DT_INT64 | GOLAP_DT_FLOAT64 */
    uint64_t out_count_ptr;      /* device u64[n_groups] rows per group (0 = not wanted); groups with 0 rows are absent */
    int32_t  key_mode;           /* GOLAP_KEYS_DENSE (0): slot = key % n_groups exactly as join_kernels.cuh:230 -- a correct GROUP BY when the keys
                                    lie in [0, n_groups); distinct keys congruent mod n_groups are merged (the reference kernel's behaviour).
                                    GOLAP_KEYS_HASHED (1): ANY int64 / int32 keys: an open-addressing key -> slot table in device memory
                                    (aggregate_kernel.rs:76-77 "global-memory hash table with open addressing") gives every distinct key a dense
                                    slot, then the dense path runs on the slots.  n_groups = upper bound on the number of distinct keys. */
    int32_t  reserved0;
    uint64_t out_keys_ptr;       /* HASHED: device int64[n_groups]: the key that owns each slot (one slot per distinct key; which slot is unspecified, like the row order of GROUP BY) */
    uint64_t out_n_groups_ptr;   /* HASHED: device u32: distinct keys found.  A value above n_groups means the bound was too small: the outputs
                                    are then meaningless (memory-safe) and the call must be repeated with a larger n_groups. */
    const uint64_t* value_validity_ptrs;   /* host array (n_value_cols) of device u8[n_rows], 0 = NULL (column_buffer.rs:59-60); a null array or a
                                    0 entry = that column has no NULLs.  SQL aggregates: SUM / MIN / MAX / AVG skip NULL values, COUNT counts
                                    the non-NULL ones; a group whose values are all NULL reports COUNT 0 and the reference's identities
                                    (executor.rs:308-311, :339-345).  out_count_ptr then counts the non-NULL rows of the LAST aggregated column. */
    uint64_t out_slot_ids_ptr;   /* ABI 4, HASHED only, optional: device u32[n_rows], the slot (row of the outputs) every input row was aggregated
                                    into -- what a caller needs to check rows against their group (golap_group_keys_verify) */
} golap_aggregate_params;
enum { GOLAP_KEYS_DENSE = 0, GOLAP_KEYS_HASHED = 1 };

/* Results are f64 like the CPU path (executor.rs:330-349): SUM, COUNT (as f64),
 * MIN, MAX, AVG (= sum/count, 0.0 for an empty group, executor.rs:339-345).
 * Int64 inputs are converted `as f64` (executor.rs:521).  Ungrouped
 * (group_key_ptr == 0, n_groups == 1) reproduces execute_aggregate. */
int32_t golap_aggregate(golap_ctx* ctx, const golap_aggregate_spec* specs, uint32_t n_specs,
                        const golap_aggregate_params* params, void* stream);

/* GROUP BY over row-sharded inputs, one process per GPU (collective; same communicator as golap_sharded_join).  Same parameter
 * structs as golap_aggregate, describing THIS rank's rows; every rank passes the same specs, key_mode and n_groups.
 *   GOLAP_KEYS_DENSE (and ungrouped): local pre-aggregation, all-gather of the [statistic][group] tables, fold in rank order (a
 *     fixed order: identical on every rank, reproducible) -- the result (out_ptr, out_count_ptr) is REPLICATED on every rank.
 *   GOLAP_KEYS_HASHED: local pre-aggregation by key, partial groups exchanged by key hash (the join's partition function), merged by
 *     the owner -- the result stays SHARDED: a key is reported by exactly one rank (out_keys_ptr / out_n_groups_ptr as
 *     golap_aggregate; n_groups bounds the distinct keys of any one rank's input AND of any one rank's share of the result).
 * Host-side failures before the exchange are agreed on like golap_sharded_join's (GOLAP_ERR_PEER).  Synchronises the stream. */
int32_t golap_sharded_aggregate(golap_ctx* ctx, golap_comm* comm, const golap_aggregate_spec* specs, uint32_t n_specs,
                                const golap_aggregate_params* params, void* stream);

/* ---- arrow-interop -------------------------------------------------------
 * record_batch_convert.rs:35-110: numeric columns widened to 8-byte LE, nulls -> 0
 * (validity = one byte per row, column_buffer.rs:59-60; 0 pointer = all valid). */
int32_t golap_widen_to_8byte(golap_ctx* ctx, uint64_t in_ptr, int32_t in_dtype /*GOLAP_WIDEN_* */, uint64_t validity_ptr,
                             uint64_t n_rows, uint64_t out_ptr, void* stream);
/* Utf8 columns, record_batch_convert.rs:93-97 + :112-130: out[i] = FNV-1a 64 of bytes [offsets[i], offsets[i + 1]) as int64 (offset
 * basis 14695981039346656037, prime 1099511628211), NULL rows (validity byte 0) -> 0.  offsets: device int64[n_rows + 1] (Arrow's
 * int32 offsets widened by the caller), data: device bytes, validity: device u8[n_rows] or 0. */
int32_t golap_hash_utf8(golap_ctx* ctx, uint64_t offsets_ptr, uint64_t data_ptr, uint64_t validity_ptr, uint64_t n_rows, uint64_t out_ptr, void* stream);
enum { GOLAP_WIDEN_I8 = 0, GOLAP_WIDEN_I16, GOLAP_WIDEN_I32, GOLAP_WIDEN_I64, GOLAP_WIDEN_U8, GOLAP_WIDEN_U16,
       GOLAP_WIDEN_U32, GOLAP_WIDEN_U64, GOLAP_WIDEN_F32, GOLAP_WIDEN_F64 };

/* ---- device-side ingest and materialisation (ABI 4) ------------------------
 * What record_batch_convert.rs:35-110 (Arrow -> GPU buffers) and :146-178 (GPU buffers -> Arrow) do in per-value host loops, on the
 * device: the raw Arrow buffers are uploaded as they are (values at their own width, validity as the BITMAP) and expanded, widened,
 * gathered and packed where they live.  The operator layer (golap_host.h) is built on these; so is any host that keeps its columns
 * resident in HBM between operators. */
/* validity bitmap -> one byte per row: out[i] = bit (bit_offset + i) of the bitmap, LSB first (array.is_valid(i), :36-40) */
int32_t golap_expand_validity(golap_ctx* ctx, uint64_t bitmap_ptr /*device*/, uint64_t bit_offset, uint64_t n_rows, uint64_t out_ptr /*u8[n_rows]*/, void* stream);
/* one byte per row -> Arrow validity bitmap (ceil(n_rows / 8) bytes, unused bits 0) and the number of NULL rows (u64, device) */
int32_t golap_pack_validity(golap_ctx* ctx, uint64_t validity_ptr, uint64_t n_rows, uint64_t out_bitmap_ptr, uint64_t out_null_count_ptr, void* stream);
/* validity of gathered rows: out[i] = 0 when indices[i] is the NULL row id 0xFFFFFFFF (the missing side of an outer-join row), else
 * src[indices[i]] != 0; src_ptr = 0: the source column has no NULLs (out[i] = 1).  count_ptr as golap_take_8byte. */
int32_t golap_take_1byte(golap_ctx* ctx, uint64_t src_ptr, uint64_t indices_ptr, uint64_t count_ptr /*u64 device, or 0*/, uint64_t n_indices_max,
                         uint64_t out_ptr /*u8*/, void* stream);
/* (build_rid, probe_rid) pairs -> one u32 index vector per side (left_matches / right_matches, join_kernels.cuh:125-126) */
int32_t golap_pairs_unzip(golap_ctx* ctx, uint64_t pairs_ptr, uint64_t n_pairs, uint64_t out_build_idx_ptr, uint64_t out_probe_idx_ptr, void* stream);
/* ascending in-place sort of n 64-bit words (n < 2^32).  A pair list sorted as 64-bit words is in the canonical (probe_rid, build_rid)
 * order of the parity tests, NULL row ids last: the kernels emit pairs in atomic order (join_kernels.cuh:155). */
int32_t golap_sort_u64(golap_ctx* ctx, uint64_t ptr, uint64_t n, void* stream);
/* out[0] = min, out[1] = max (int64, exact), out[2] = number of the valid rows of an int64 column; no valid row: INT64_MAX, INT64_MIN, 0 */
int32_t golap_key_minmax(golap_ctx* ctx, uint64_t column_ptr, uint64_t validity_ptr /*u8[n] or 0*/, uint64_t n_rows, uint64_t out_ptr /*int64[3] device*/, void* stream);
/* Composite GROUP BY key of up to 4 int64 columns as ONE exact int64 (physical_plan.rs:52-54 carries a Vec of key columns):
 * out[i] = sum_c digit_c(i) * weight[c], digit_c = value - min_value[c] for a valid row, null_digit[c] for a NULL one (the caller picks
 * weights so that the sum is unique per key tuple: a mixed-radix number). */
typedef struct golap_pack_keys_params {
    uint64_t col_ptrs[4];        /* device int64[n_rows] */
    uint64_t validity_ptrs[4];   /* device u8[n_rows] or 0 */
    int64_t  min_value[4];
    uint64_t weight[4];
    uint64_t null_digit[4];
    uint32_t n_cols;
    uint32_t reserved0;
    uint64_t n_rows;
    uint64_t out_ptr;            /* device int64[n_rows] */
} golap_pack_keys_params;
int32_t golap_pack_group_keys(golap_ctx* ctx, const golap_pack_keys_params* params, void* stream);
/* Composite keys that do NOT fit one int64 (physical_plan.rs:52-54 puts no bound on the key columns): GROUP BY runs on a 64-bit
 * SURROGATE -- out[i] = a seeded 64-bit mix of the row's key tuple, a NULL in a column mixed in as its own symbol (NULLs group together,
 * (NULL, 1) != (1, NULL)); params: col_ptrs / validity_ptrs / n_cols / n_rows / out_ptr (min_value, weight, null_digit unused) -- and
 * is then CHECKED: golap_group_keys_verify names a representative row per slot (rep_rows[s] = the smallest row index aggregated into
 * slot s) and counts the rows whose key tuple differs from their slot's representative.  0: every group holds exactly one tuple and
 * the result is exact; otherwise two tuples shared a surrogate (probability ~ groups^2 / 2^65) and the caller repeats with another seed. */
int32_t golap_hash_key_columns(golap_ctx* ctx, const golap_pack_keys_params* params, uint64_t seed, void* stream);
int32_t golap_group_keys_verify(golap_ctx* ctx, const golap_pack_keys_params* params, uint64_t slot_ids_ptr /*u32[n_rows]*/, uint64_t n_slots,
                                uint64_t out_rep_rows_ptr /*u32[n_slots]*/, uint64_t out_mismatch_ptr /*u64 device*/, void* stream);
/* device -> device copy and byte fill in stream order (concatenating the batches of an operator's input) */
int32_t golap_device_copy(golap_ctx* ctx, uint64_t src_ptr, uint64_t dst_ptr, uint64_t bytes, void* stream);
int32_t golap_device_fill_u8(golap_ctx* ctx, uint64_t ptr, uint32_t value, uint64_t bytes, void* stream);

/* ---- synthetic generators --------------------
 * Counter-based, so each GPU produces its own shard.  kind selects the formula. */
enum {
    GOLAP_GEN_BUILD_KEYS = 0,   /* int64: (2654435761*i + 12345) mod modulus              */
    GOLAP_GEN_SM64_MOD_I64 = 1, /* int64: sm64(seed, i) mod modulus  (probe keys: seed 43) */
    GOLAP_GEN_SM64_MOD_I32 = 2, /* int32                                                   */
    GOLAP_GEN_SM64_MOD_F64 = 3, /* f64 (integer-valued)                                    */
    GOLAP_GEN_SHIFT32_I64 = 4   /* int64: i << 32  (fold-collision keys)       */
};
int32_t golap_generate(golap_ctx* ctx, int32_t kind, uint64_t seed, uint64_t first, uint64_t count, uint64_t modulus,
                       uint64_t out_ptr, void* stream);

/* Order-independent verification of a pair list on the device (size-independent
 * property check for the 1B-row runs): out[0] = number of pairs i < n_pairs whose
 * keys are equal (build_keys[b - build_rid_base] == probe_keys[p - probe_rid_base]),
 * out[1] = sum of pair_mix(b, p) mod 2^64, out[2] = sum of probe ids, out[3] = xor of mix(probe id). */
int32_t golap_verify_pairs(golap_ctx* ctx, uint64_t pairs_ptr, uint64_t n_pairs, uint64_t build_key_ptr, uint64_t probe_key_ptr,
                           uint32_t build_rid_base, uint32_t probe_rid_base, uint64_t out_ptr /*u64[4] device*/, void* stream);

/* ---- stats ---------------------------------------------------------------
 * Per-kernel accumulated device time (hipEvent pairs on the launch stream),
 * collected when enable_profiling = 1.  golap_stats_get resolves pending
 * events (synchronises them). */
#define GOLAP_MAX_KERNEL_STATS 64
typedef struct golap_kernel_stat {
    char     name[48];
    uint64_t launches;
    double   total_ms;
    uint64_t bytes;        /* algorithmic bytes credited to these launches */
} golap_kernel_stat;
typedef struct golap_stats {
    uint32_t n;
    golap_kernel_stat k[GOLAP_MAX_KERNEL_STATS];
} golap_stats;
int32_t golap_stats_get(golap_ctx* ctx, golap_stats* out);
int32_t golap_stats_reset(golap_ctx* ctx);

/* Plan the last join used (for DESIGN/bench reporting). */
typedef struct golap_join_plan {
    uint32_t bits1, bits2;        /* fan-out bits of pass 1 (>= radix_bits) and of the later passes together */
    uint32_t tile_rows;           /* rows per tile of the first scatter pass */
    uint32_t chunks1;             /* work-groups (chunks of tiles) of the first pass */
    uint32_t slots2;              /* chunks per group in pass 2 (1 = one work-group per group, the no-histogram form) */
    uint32_t table_capacity;      /* LDS hash-table rows per sub-partition */
    uint64_t n_subpartitions;
    uint32_t no_histogram_last_pass;   /* 1 when pass 2 writes into pre-provisioned slots */
    uint32_t no_histogram_first_pass;  /* 1 when pass 1 reserves its bucket ranges by atomics instead of a histogram */
    uint32_t optimistic_narrow_keys;   /* 1 when the join first tries 8-byte (key32, row id) tuples; a key outside [0, 2^32)
                                          hands over to the 64-bit pipeline on the device */
    uint32_t reserved0;
} golap_join_plan;
int32_t golap_join_plan_get(golap_ctx* ctx, uint64_t n_build, uint32_t n_owners, golap_join_plan* out);

#ifdef __cplusplus
}
#endif
#endif /* GOLAP_H */
