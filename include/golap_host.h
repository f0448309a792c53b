/* golap_host.h -- operator-level C ABI: the reference's PhysicalPlan operator API
 * (gpu-olap-core/src/physical_plan.rs:11-84) and GpuExecutor::execute
 * (gpu-olap-core/src/executor.rs:15-24, :37-104) over Arrow RecordBatches, implemented in
 * C++ (gpu_olap_engine_amd/host/golap_executor.cpp) purely on top of include/golap.h.
 *
 * RecordBatches cross the boundary through the Arrow C Data Interface (struct arrays:
 * ArrowSchema format "+s", children "l" Int64 / "g" Float64 / "i" Int32), the ABI arrow-rs,
 * pyarrow and Arrow C++ all export.  A plan is built bottom-up with the constructors below
 * (each mirrors one PhysicalPlan variant) and is consumed by golap_execute.
 */
#ifndef GOLAP_HOST_H
#define GOLAP_HOST_H

#include <stdint.h>

#include "golap.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Arrow C Data Interface (https://arrow.apache.org/docs/format/CDataInterface.html) ---- */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
    const char* format;
    const char* name;
    const char* metadata;
    int64_t flags;
    int64_t n_children;
    struct ArrowSchema** children;
    struct ArrowSchema* dictionary;
    void (*release)(struct ArrowSchema*);
    void* private_data;
};
struct ArrowArray {
    int64_t length;
    int64_t null_count;
    int64_t offset;
    int64_t n_buffers;
    int64_t n_children;
    const void** buffers;
    struct ArrowArray** children;
    struct ArrowArray* dictionary;
    void (*release)(struct ArrowArray*);
    void* private_data;
};
#endif

typedef struct golap_plan golap_plan;    /* PhysicalPlan, physical_plan.rs:11-64 */
typedef struct golap_expr golap_expr;    /* PhysicalExpr, physical_plan.rs:66-75 */

/* JoinType, logical_plan.rs:53-58 */
typedef enum golap_join_type { GOLAP_JOIN_INNER = 0, GOLAP_JOIN_LEFT = 1, GOLAP_JOIN_RIGHT = 2, GOLAP_JOIN_FULL = 3 } golap_join_type;

/* ---- PhysicalExpr constructors (ownership of children moves into the parent) ---- */
golap_expr* golap_expr_column(uint64_t index, const char* name);                       /* Column { index, name }  :68 */
golap_expr* golap_expr_literal(const char* value);                                     /* Literal { value }       :69 */
golap_expr* golap_expr_binary(golap_expr* left, const char* op, golap_expr* right);     /* BinaryExpr {..}         :70-74 */
void        golap_expr_free(golap_expr* e);

/* ---- PhysicalPlan constructors ---- */
/* A materialised child: the Vec<RecordBatch> an already-executed subtree returned (executor.rs:49,66-67,94).
 * The arrays are imported (moved: their release callbacks are called by the library). */
golap_plan* golap_plan_batches(struct ArrowArray** arrays, struct ArrowSchema** schemas, uint32_t n_batches);
/* GpuTableScan: like the reference (executor.rs:118-154) it yields one EMPTY batch of the schema. */
golap_plan* golap_plan_table_scan(const char* table_name, struct ArrowSchema* schema);
golap_plan* golap_plan_filter(golap_plan* input, golap_expr* predicate);                                           /* GpuFilter     :20-23 */
golap_plan* golap_plan_projection(golap_plan* input, golap_expr** exprs, uint32_t n_exprs);                         /* GpuProjection :26-29 */
golap_plan* golap_plan_hash_join(golap_plan* left, golap_plan* right, golap_expr** left_keys, uint32_t n_left_keys,
                                 golap_expr** right_keys, uint32_t n_right_keys, golap_join_type join_type);        /* GpuHashJoin   :32-39 */
/* aggr_funcs[i] is a golap_agg_func (AggregateExpr::{Sum,Count,Min,Max,Avg}, :78-84) over aggr_inputs[i]. */
golap_plan* golap_plan_aggregate(golap_plan* input, golap_expr** group_by, uint32_t n_group_by, const int32_t* aggr_funcs,
                                 golap_expr** aggr_inputs, uint32_t n_aggr);                                        /* GpuAggregate  :52-57 */
golap_plan* golap_plan_sort(golap_plan* input, golap_expr** exprs, uint32_t n_exprs);                               /* GpuSort       :60-63 (pass-through, executor.rs:361-370) */
void        golap_plan_free(golap_plan* p);

/* ---- execute (executor.rs:15-24) ----
 * Consumes the plan.  On success *out_n batches are available through golap_result_export; the result
 * handle must be freed with golap_result_free.  Errors: status != 0 and golap_last_error(ctx). */
typedef struct golap_result golap_result;
int32_t golap_execute(golap_ctx* ctx, golap_plan* plan, golap_result** out_result, uint32_t* out_n_batches);
/* Where the wall-clock time of the golap_execute call behind this result went.  Columns are device-resident between operators: an
 * operator uploads what is not in HBM yet (every transfer of that phase queued on the TransferQueue's streams, one wait), runs its
 * kernels, and reads back counts; the final result is downloaded once.  ("H2D excluded and reported separately".) */
typedef struct golap_exec_profile {
    uint64_t h2d_bytes, d2h_bytes;
    double   h2d_ms;           /* wall time of the upload phases (enqueue from several host threads + the one wait) */
    double   d2h_ms;           /* wall time of the download phases (counts, group tables, the final result) */
    double   kernel_wait_ms;   /* wall time the host spent waiting for the device's compute stream */
    double   total_ms;         /* the whole call */
} golap_exec_profile;
int32_t golap_result_profile(const golap_result* r, golap_exec_profile* out);
/* Host threads that feed the TransferQueue in a transfer phase (thread t -> stream t % num_streams; default 8 = EngineConfig's
 * num_streams, lib.rs:37).  Process-wide; more than the context's streams only makes threads share a stream. */
void golap_host_set_transfer_threads(uint32_t n);
/* 1: GpuFilter over several batches uploads batch k + 1 (a background thread feeding the TransferQueue) while the device filters batch
 * k, and reads all kept-row counts back once, behind the last batch.  0 (the default): one transfer phase for all batches, then the
 * kernels -- measured faster on the PCIe-bound plans this layer sees (54 against 58-62 ms).  Process-wide; results
 * do not depend on it. */
void golap_host_set_pipelining(uint32_t on);
/* Moves batch i out as an Arrow struct array (the caller releases it through the Arrow callbacks). */
int32_t golap_result_export(golap_result* r, uint32_t i, struct ArrowArray* out_array, struct ArrowSchema* out_schema);
void    golap_result_free(golap_result* r);
/* Message of the last host-layer failure on this thread (plan construction, execute, export). */
const char* golap_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* GOLAP_HOST_H */
