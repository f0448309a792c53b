// golap_executor.cpp -- GpuExecutor: the reference's operator API over Arrow RecordBatches,
// written in C++ on top of the C ABI (include/golap.h) only.  No HIP calls here: device
// memory comes from the SlabAllocator entry points, copies go through the TransferQueue.
//
// Mirrors gpu-olap-core/src/executor.rs:
//   execute            :15-24, dispatch :37-104
//   execute_table_scan :110-155   (empty batch, as the reference)
//   execute_filter     :161-178   -> golap_filter_compact + golap_take_8byte / golap_take_1byte
//   execute_projection :184-214   (column references and literals)
//   execute_hash_join  :220-265   -> golap_join; unlike the reference (which returns an empty
//                                    batch at :255-264) the pairs are materialised
//   execute_aggregate  :297-355   -> golap_aggregate (ungrouped = reference; grouped = dense or hashed keys)
//   execute_sort       :361-370   (pass-through, as the reference)
//   eval_expr          :377-409   (column by name first, then by index)
//
// Round 4: columns are DEVICE-RESIDENT between operators.  A batch is imported without a copy (the Arrow arrays are held, not
// memcpy'd), uploaded once -- values at their Arrow width, validity as the Arrow BITMAP, every transfer of an operator queued on the
// TransferQueue's streams from several host threads and waited for once -- and expanded / widened / hashed where it lands
// (record_batch_convert.rs:35-110 on the device: golap_expand_validity, golap_widen_to_8byte, golap_hash_utf8).  Filter and join outputs
// stay in HBM (row-id lists never visit the host: validity is gathered by golap_take_1byte, pairs are ordered by golap_sort_u64 and
// split by golap_pairs_unzip; composite GROUP BY keys are ranged by golap_key_minmax and packed by golap_pack_group_keys); only counts
// (8 bytes), group tables and the final result come back.  No per-row host loop is left in execute_filter / _hash_join / _aggregate.
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/golap_host.h"

namespace {

std::atomic<uint32_t> g_xfer_threads{8};   // host threads (= transfer streams used) of a transfer phase: golap_host_set_transfer_threads
std::atomic<uint32_t> g_pipeline{0};       // GpuFilter over several batches: batch k is filtered while batch k + 1 is uploaded (golap_host_set_pipelining; measured: loses, off)
thread_local std::string g_err;
int32_t herr(int32_t code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define H_TRY(expr)                      \
    do {                                 \
        int32_t _s = (expr);             \
        if (_s != GOLAP_OK) return _s;   \
    } while (0)

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---------------------------------------------------------------------------
// columns
// ---------------------------------------------------------------------------
// logical type (what the Arrow field said); on the device every numeric column is 8 bytes wide (record_batch_convert.rs:19:
// "All numeric types are widened to i64 / f64"), a Utf8 column is its FNV-1a 64 hash, i.e. Int64 (:93-97)
enum class DType { Int64, Float64, Int32, Float32, IntNarrow, Utf8, Other };
bool is_float(DType d) { return d == DType::Float64 || d == DType::Float32; }
bool is_numeric(DType d) { return d != DType::Other; }

struct ArrowFmt { DType dtype; int width; int widen; };   // widen: GOLAP_WIDEN_* code of the raw values
ArrowFmt fmt_of(const char* f) {
    if (!f || !f[0] || f[1]) return {DType::Other, 0, -1};
    switch (f[0]) {
        case 'l': return {DType::Int64, 8, GOLAP_WIDEN_I64};
        case 'g': return {DType::Float64, 8, GOLAP_WIDEN_F64};
        case 'i': return {DType::Int32, 4, GOLAP_WIDEN_I32};
        case 'f': return {DType::Float32, 4, GOLAP_WIDEN_F32};
        case 'c': return {DType::IntNarrow, 1, GOLAP_WIDEN_I8};
        case 'C': return {DType::IntNarrow, 1, GOLAP_WIDEN_U8};
        case 's': return {DType::IntNarrow, 2, GOLAP_WIDEN_I16};
        case 'S': return {DType::IntNarrow, 2, GOLAP_WIDEN_U16};
        case 'I': return {DType::IntNarrow, 4, GOLAP_WIDEN_U32};
        case 'L': return {DType::IntNarrow, 8, GOLAP_WIDEN_U64};
        case 'u': case 'U': return {DType::Utf8, 0, -1};
        default: return {DType::Other, 0, -1};
    }
}

struct DevBuf {
    golap_ctx* ctx = nullptr;
    uint64_t ptr = 0;
    uint32_t cls = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (ptr) golap_slab_free(ctx, ptr, cls); }
    int32_t alloc(golap_ctx* c, uint64_t bytes) {
        ctx = c;
        return golap_slab_alloc(c, std::max<uint64_t>(bytes, 8), &ptr, &cls);
    }
};
using DevPtr = std::shared_ptr<DevBuf>;

// the moved-in Arrow arrays of one imported batch, alive while any column still points into them
struct ArrowHold {
    ArrowArray arr{};
    ArrowSchema sch{};
    ~ArrowHold() {
        if (arr.release) arr.release(&arr);
        if (sch.release) sch.release(&sch);
    }
};

struct Column {
    std::string name;
    DType dtype = DType::Other;
    uint64_t rows = 0;
    enum Where { Arrow, Device, Host } where = Host;
    // Arrow: as imported (no copy).  values / bitmap / offsets point into `hold`'s buffers; `off` is the array offset in rows
    std::shared_ptr<ArrowHold> hold;
    const uint8_t* a_values = nullptr;
    const uint8_t* a_bitmap = nullptr;    // null: no NULLs
    const uint8_t* a_offsets = nullptr;   // Utf8: int32 / int64 offsets, a_values = the bytes
    int64_t off = 0;
    char fmt = 0;
    // Device: 8 bytes per row + one validity byte per row (column_buffer.rs:59-60; null pointer = no NULLs)
    DevPtr d_data, d_valid;
    // Host: small results (aggregates, literals) and what export hands out; 8 bytes per row unless `fmt` says otherwise (a passed-through column)
    std::vector<uint8_t> data;
    std::vector<uint8_t> in_valid;        // one byte per row, empty = no NULLs
    std::vector<uint8_t> valid;           // Arrow bitmap (export form), with null_count
    uint64_t null_count = 0;
    bool has_validity() const { return where == Arrow ? a_bitmap != nullptr : where == Device ? (bool)d_valid : !in_valid.empty(); }
};

struct RecordBatch {
    std::vector<Column> cols;
    uint64_t num_rows = 0;
    int index_of(const std::string& name) const {
        for (size_t i = 0; i < cols.size(); ++i)
            if (cols[i].name == name) return (int)i;
        return -1;
    }
};

bool import_batch(ArrowArray* arr, ArrowSchema* sch, RecordBatch& out, std::string& err) {
    if (!arr || !sch || !sch->format || strcmp(sch->format, "+s")) { err = "expected an Arrow struct array (format +s)"; return false; }
    if (arr->n_children != sch->n_children) { err = "array/schema children mismatch"; return false; }
    // the arrays are MOVED in (Arrow C Data Interface: copy the structs, mark the sources released); nothing is copied
    auto hold = std::make_shared<ArrowHold>();
    hold->arr = *arr;
    hold->sch = *sch;
    arr->release = nullptr;
    sch->release = nullptr;
    const ArrowArray& A = hold->arr;
    const ArrowSchema& S = hold->sch;
    out.num_rows = (uint64_t)A.length;
    for (int64_t c = 0; c < S.n_children; ++c) {
        const ArrowSchema* cs = S.children[c];
        const ArrowArray* ca = A.children[c];
        Column col;
        col.name = cs->name ? cs->name : "";
        const ArrowFmt f = fmt_of(cs->format);
        col.dtype = f.dtype;
        col.fmt = cs->format && cs->format[0] ? cs->format[0] : 0;
        col.rows = out.num_rows;
        col.where = Column::Arrow;
        col.hold = hold;
        col.off = ca->offset + A.offset;
        const uint8_t* bitmap = ca->n_buffers >= 1 ? (const uint8_t*)ca->buffers[0] : nullptr;
        col.a_bitmap = (bitmap && ca->null_count != 0) ? bitmap : nullptr;
        if (f.dtype == DType::Utf8) {
            if (out.num_rows && (ca->n_buffers < 3 || !ca->buffers[1] || !ca->buffers[2])) { err = "string column '" + col.name + "' has no offsets / data buffer"; return false; }
            col.a_offsets = ca->n_buffers >= 2 ? (const uint8_t*)ca->buffers[1] : nullptr;
            col.a_values = ca->n_buffers >= 3 ? (const uint8_t*)ca->buffers[2] : nullptr;
        } else if (f.dtype != DType::Other) {
            col.a_values = ca->n_buffers >= 2 ? (const uint8_t*)ca->buffers[1] : nullptr;
            if (out.num_rows && !col.a_values) { err = "column '" + col.name + "' has no data buffer"; return false; }
        }
        out.cols.push_back(std::move(col));
    }
    return true;
}

// ---- export (the exported arrays own their buffers) ----
struct ExportPriv {
    std::vector<std::vector<uint8_t>> data, valid;
    std::vector<const void*> bufs;   // 2 per child + 1 for the struct
    std::vector<ArrowArray> child_arrays;
    std::vector<ArrowArray*> child_ptrs;
    std::vector<ArrowSchema> child_schemas;
    std::vector<ArrowSchema*> child_sptrs;
    std::vector<std::string> names, formats;
};
void release_child_array(ArrowArray* a) { a->release = nullptr; }
void release_child_schema(ArrowSchema* s) { s->release = nullptr; }
void release_array(ArrowArray* a) {
    if (!a || !a->release) return;
    for (int64_t i = 0; i < a->n_children; ++i)
        if (a->children[i]->release) a->children[i]->release(a->children[i]);
    delete (std::shared_ptr<ExportPriv>*)a->private_data;
    a->release = nullptr;
}
void release_schema(ArrowSchema* s) {
    if (!s || !s->release) return;
    for (int64_t i = 0; i < s->n_children; ++i)
        if (s->children[i]->release) s->children[i]->release(s->children[i]);
    delete (std::shared_ptr<ExportPriv>*)s->private_data;
    s->release = nullptr;
}

// the Arrow format a Host-state column leaves as: a passed-through column keeps its own (`fmt` set), everything that has been on the
// device is Int64 / Float64 (gpu_buffers_to_record_batch, record_batch_convert.rs:146-178, builds nothing else)
std::string export_format(const Column& c) {
    if (c.dtype == DType::Other) return "n";
    if (c.fmt && c.dtype != DType::Utf8) return std::string(1, c.fmt);
    return is_float(c.dtype) ? "g" : "l";
}

// Host-state batch -> Arrow (byte-per-row validity is packed here: result-sized host work on columns that never were on the device)
void export_batch(RecordBatch&& b, ArrowArray* oa, ArrowSchema* os) {
    auto priv = std::make_shared<ExportPriv>();
    const size_t nc = b.cols.size();
    priv->data.resize(nc);
    priv->valid.resize(nc);
    priv->bufs.resize(2 * nc + 1, nullptr);
    priv->child_arrays.resize(nc);
    priv->child_schemas.resize(nc);
    priv->names.resize(nc);
    priv->formats.resize(nc);
    for (size_t c = 0; c < nc; ++c) {
        Column& col = b.cols[c];
        if (col.valid.empty() && !col.in_valid.empty()) {
            uint64_t nulls = 0;
            col.valid.assign((size_t)(b.num_rows + 7) / 8, 0);
            for (uint64_t i = 0; i < b.num_rows; ++i) {
                if (col.in_valid[i]) col.valid[i >> 3] |= (uint8_t)(1u << (i & 7));
                else ++nulls;
            }
            col.null_count = nulls;
            if (!nulls) col.valid.clear();
        }
        priv->formats[c] = export_format(col);
        priv->data[c] = std::move(col.data);
        priv->valid[c] = std::move(col.valid);
        priv->names[c] = col.name;
        priv->bufs[2 * c] = priv->valid[c].empty() ? nullptr : (const void*)priv->valid[c].data();
        priv->bufs[2 * c + 1] = priv->data[c].empty() ? (const void*)"" : (const void*)priv->data[c].data();
        ArrowArray& ca = priv->child_arrays[c];
        memset(&ca, 0, sizeof ca);
        ca.length = (int64_t)b.num_rows;
        ca.null_count = (int64_t)col.null_count;
        ca.n_buffers = 2;
        ca.buffers = &priv->bufs[2 * c];
        ca.release = release_child_array;
        ArrowSchema& cs = priv->child_schemas[c];
        memset(&cs, 0, sizeof cs);
        cs.format = priv->formats[c].c_str();
        cs.name = priv->names[c].c_str();
        cs.flags = col.null_count ? 2 /* ARROW_FLAG_NULLABLE */ : 0;
        cs.release = release_child_schema;
        priv->child_ptrs.push_back(&ca);
        priv->child_sptrs.push_back(&cs);
    }
    memset(oa, 0, sizeof *oa);
    oa->length = (int64_t)b.num_rows;
    oa->n_buffers = 1;
    oa->buffers = &priv->bufs[2 * nc];
    oa->n_children = (int64_t)nc;
    oa->children = priv->child_ptrs.data();
    oa->release = release_array;
    oa->private_data = new std::shared_ptr<ExportPriv>(priv);
    memset(os, 0, sizeof *os);
    os->format = "+s";
    os->name = "";
    os->n_children = (int64_t)nc;
    os->children = priv->child_sptrs.data();
    os->release = release_schema;
    os->private_data = new std::shared_ptr<ExportPriv>(priv);
}

}  // namespace

// ---------------------------------------------------------------------------
// plan / expr nodes
// ---------------------------------------------------------------------------
struct golap_expr {
    enum Kind { Column, Literal, Binary } kind;
    uint64_t index = 0;
    std::string name, value, op;
    std::unique_ptr<golap_expr> left, right;
};

struct golap_plan {
    enum Kind { Batches, TableScan, Filter, Projection, HashJoin, Aggregate, Sort } kind;
    std::vector<RecordBatch> batches;              // Batches / TableScan (one empty batch)
    std::unique_ptr<golap_plan> input, right;      // input = left for joins
    std::unique_ptr<golap_expr> predicate;
    std::vector<std::unique_ptr<golap_expr>> exprs, left_keys, right_keys, group_by, aggr_inputs;
    std::vector<int32_t> aggr_funcs;
    golap_join_type join_type = GOLAP_JOIN_INNER;
};

struct golap_result {
    std::vector<RecordBatch> batches;
    golap_exec_profile profile{};
};

namespace {

// ---------------------------------------------------------------------------
// the executor
// ---------------------------------------------------------------------------
struct Xfer { int32_t dir; uint64_t src, dst, bytes; };

struct Exec {
    golap_ctx* ctx = nullptr;
    golap_exec_profile prof{};
    std::vector<Xfer> pending;          // transfers of the current phase, run together by flush()
    std::vector<DevPtr> keep;           // device temporaries that kernels still in flight may read: released by end_op() behind a ksync

    int32_t need_ctx() const { return ctx ? GOLAP_OK : herr(GOLAP_ERR_INVALID_ARG, "this plan needs the device: golap_execute was called without a context (there is no CPU fallback)"); }
    int32_t dev_alloc(DevPtr& out, uint64_t bytes) {
        out = std::make_shared<DevBuf>();
        return out->alloc(ctx, bytes);
    }
    int32_t temp(DevPtr& out, uint64_t bytes) {
        H_TRY(dev_alloc(out, bytes));
        keep.push_back(out);
        return GOLAP_OK;
    }
    void drop(DevPtr& p) {   // a temporary no kernel in flight uses any more (the caller has just synchronised): back to the pool now
        keep.erase(std::remove(keep.begin(), keep.end(), p), keep.end());
        p.reset();
    }
    void up(const void* src, uint64_t dst, uint64_t bytes) { if (bytes) pending.push_back({GOLAP_HOST_TO_DEVICE, (uint64_t)(uintptr_t)src, dst, bytes}); }
    void down(uint64_t src, void* dst, uint64_t bytes) { if (bytes) pending.push_back({GOLAP_DEVICE_TO_HOST, src, (uint64_t)(uintptr_t)dst, bytes}); }
    int32_t flush();
    // the transfers of `list`, cut into pieces and dealt to the host threads, ONE wait; touches nothing of this object but `ctx`
    // (flush() runs it on the calling thread, the filter's Uploader on its own)
    int32_t run_transfers(const std::vector<Xfer>& list, uint64_t& up_bytes, uint64_t& down_bytes) const;
    int32_t ksync() {
        const double t0 = now_ms();
        const int32_t s = golap_ctx_synchronize(ctx, nullptr);
        prof.kernel_wait_ms += now_ms() - t0;
        return s;
    }
    // end of an operator: the device is drained ONCE, then its temporaries go back to the slab pool
    int32_t end_op() {
        int32_t s = GOLAP_OK;
        if (ctx && !keep.empty()) s = ksync();
        keep.clear();
        return s;
    }
    int32_t read_u64(uint64_t dev_ptr, uint64_t* out, uint32_t n = 1) {   // a count the host needs to size what comes next
        H_TRY(ksync());
        down(dev_ptr, out, 8ull * n);
        return flush();
    }

    // eval_expr, executor.rs:377-409: column by NAME first, then by index
    static int resolve_column(const golap_expr& e, const RecordBatch& b) {
        int i = b.index_of(e.name);
        if (i >= 0) return i;
        if (e.index < b.cols.size()) return (int)e.index;
        return -1;
    }

    // record_batch_to_gpu_buffers in three steps: queue (allocations + the raw uploads into `pending`), the transfer, finish (the kernels
    // that expand / widen / hash what has landed) -- to_device() is the three in a row, the pipelined filter runs them per batch
    struct IngestJob { Column* c = nullptr; DevPtr raw, bitmap, offs, offs32; uint64_t bit_off = 0; int64_t str_first = 0; int widen = -1; bool large = false; };
    int32_t ingest_queue(const std::vector<RecordBatch*>& bs, std::vector<IngestJob>& jobs);
    int32_t ingest_finish(std::vector<IngestJob>& jobs);
    int32_t to_device(const std::vector<RecordBatch*>& bs);
    int32_t to_host(std::vector<RecordBatch>& bs);
    int32_t run(golap_plan& p, std::vector<RecordBatch>& out);
    int32_t filter(std::vector<RecordBatch>& in, const golap_expr& pred, std::vector<RecordBatch>& out);
    struct FilterTodo { bool pass = true; std::vector<golap_filter_predicate> fps; std::vector<int> fcols; };
    int32_t filter_pipelined(std::vector<RecordBatch>& in, const std::vector<FilterTodo>& todo, std::vector<RecordBatch>& out);
    int32_t projection(std::vector<RecordBatch>& in, const std::vector<std::unique_ptr<golap_expr>>& exprs, std::vector<RecordBatch>& out);
    int32_t hash_join(std::vector<RecordBatch>& left, std::vector<RecordBatch>& right, golap_plan& p, std::vector<RecordBatch>& out);
    int32_t aggregate(std::vector<RecordBatch>& in, golap_plan& p, std::vector<RecordBatch>& out);

    struct Cat { uint64_t ptr = 0, vptr = 0; };   // one column of an operator's whole input as ONE device array (+ validity bytes, 0 = no NULLs)
    int32_t concat(const std::vector<RecordBatch*>& bs, const std::vector<int>& ci, uint64_t n, Cat& out);
};

// Runs the queued transfers: large ones are cut into 32 MiB pieces and dealt to up to 8 host threads, thread t on transfer stream t
// (transfer_queue.rs:72: stream_id % num_streams) -- pageable Arrow buffers are staged through that stream's two pinned bounce buffers,
// so the staging memcpy of one stream runs beside the DMA of the others; the caller waits ONCE, for all of them.
int32_t Exec::run_transfers(const std::vector<Xfer>& list, uint64_t& up_bytes, uint64_t& down_bytes) const {
    std::vector<Xfer> pieces;
    up_bytes = down_bytes = 0;
    const uint64_t piece = 32ull << 20;
    for (const Xfer& x : list) {
        (x.dir == GOLAP_HOST_TO_DEVICE ? up_bytes : down_bytes) += x.bytes;
        for (uint64_t o = 0; o < x.bytes; o += piece) pieces.push_back({x.dir, x.src + o, x.dst + o, std::min(piece, x.bytes - o)});
    }
    if (pieces.empty()) return GOLAP_OK;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t T = (up_bytes + down_bytes) < (8ull << 20) ? 1 : std::min<size_t>({(size_t)g_xfer_threads.load(), hw, pieces.size()});
    std::vector<int32_t> status(T, GOLAP_OK);
    golap_ctx* c = ctx;
    auto work = [&](size_t t) {
        for (size_t i = t; i < pieces.size(); i += T) {
            const int32_t s = golap_xfer_enqueue(c, pieces[i].dir, pieces[i].src, pieces[i].dst, pieces[i].bytes, (uint32_t)t);
            if (s != GOLAP_OK) { status[t] = s; return; }
        }
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (size_t t = 1; t < T; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
    }
    const int32_t sync = golap_xfer_synchronize(c);
    for (int32_t s : status) H_TRY(s);
    return sync;
}

int32_t Exec::flush() {
    if (pending.empty()) return GOLAP_OK;
    H_TRY(need_ctx());
    const double t0 = now_ms();
    std::vector<Xfer> list;
    list.swap(pending);
    uint64_t up_bytes = 0, down_bytes = 0;
    const int32_t s = run_transfers(list, up_bytes, down_bytes);
    const double dt = now_ms() - t0;
    // (a phase moves one way: uploads at an operator's start, downloads of counts / results at its end)
    if (up_bytes >= down_bytes) { prof.h2d_bytes += up_bytes; prof.h2d_ms += dt; prof.d2h_bytes += down_bytes; }
    else { prof.d2h_bytes += down_bytes; prof.d2h_ms += dt; prof.h2d_bytes += up_bytes; }
    return s;
}

// ---------------------------------------------------------------------------
// record_batch_to_gpu_buffers (record_batch_convert.rs:22-33, column_to_buffer :35-110) on the device
// ---------------------------------------------------------------------------
// Every column of `bs` that is not in HBM yet goes there: phase 1 queues the raw uploads of ALL of them (values at their Arrow width
// from the array offset on, the validity bitmap from the byte that holds the first row's bit, string offsets and bytes), one flush;
// phase 2 expands the bitmaps to a byte per row, widens to 8 bytes with NULL -> 0 (:102-110) and hashes strings (:112-130) in place.
int32_t Exec::ingest_queue(const std::vector<RecordBatch*>& bs, std::vector<IngestJob>& jobs) {
    using Job = IngestJob;
    for (RecordBatch* b : bs)
        for (Column& c : b->cols) {
            if (c.where == Column::Device) continue;
            if (!is_numeric(c.dtype)) return herr(GOLAP_ERR_UNSUPPORTED, "column '" + c.name + "': unsupported column type for GPU conversion (record_batch_convert.rs:98)");
            H_TRY(need_ctx());
            Job j;
            j.c = &c;
            const uint64_t n = c.rows;
            H_TRY(dev_alloc(c.d_data, n * 8));
            if (c.where == Column::Host) {                         // a small result of an earlier operator: already 8 bytes wide, validity as bytes
                up(c.data.data(), c.d_data->ptr, n * 8);
                if (!c.in_valid.empty()) { H_TRY(dev_alloc(c.d_valid, n)); up(c.in_valid.data(), c.d_valid->ptr, n); }
                jobs.push_back(std::move(j));
                continue;
            }
            if (n && c.a_bitmap) {
                const uint64_t first_byte = (uint64_t)c.off >> 3;
                j.bit_off = (uint64_t)c.off & 7;
                const uint64_t nbytes = (j.bit_off + n + 7) / 8;
                H_TRY(temp(j.bitmap, nbytes));
                H_TRY(dev_alloc(c.d_valid, n));
                up(c.a_bitmap + first_byte, j.bitmap->ptr, nbytes);
            }
            if (c.dtype == DType::Utf8) {
                j.large = c.fmt == 'U';
                const size_t ow = j.large ? 8 : 4;
                if (n) {
                    const int64_t first = j.large ? ((const int64_t*)c.a_offsets)[c.off] : (int64_t)((const int32_t*)c.a_offsets)[c.off];
                    const int64_t last = j.large ? ((const int64_t*)c.a_offsets)[c.off + (int64_t)n] : (int64_t)((const int32_t*)c.a_offsets)[c.off + (int64_t)n];
                    j.str_first = first;
                    H_TRY(temp(j.offs, (n + 1) * 8));
                    if (!j.large) H_TRY(temp(j.offs32, (n + 1) * 4));   // (int32 offsets are widened on the device)
                    H_TRY(temp(j.raw, (uint64_t)std::max<int64_t>(last - first, 8)));
                    up(c.a_offsets + (size_t)c.off * ow, j.large ? j.offs->ptr : j.offs32->ptr, (n + 1) * ow);
                    if (last > first) up(c.a_values + first, j.raw->ptr, (uint64_t)(last - first));
                }
            } else {
                const ArrowFmt f = fmt_of(std::string(1, c.fmt).c_str());
                j.widen = f.widen;
                if (n) {
                    if (f.width == 8) up(c.a_values + (size_t)c.off * 8, c.d_data->ptr, n * 8);          // straight into the column
                    else { H_TRY(temp(j.raw, n * (uint64_t)f.width)); up(c.a_values + (size_t)c.off * (size_t)f.width, j.raw->ptr, n * (uint64_t)f.width); }
                }
            }
            jobs.push_back(std::move(j));
        }
    return GOLAP_OK;
}

int32_t Exec::ingest_finish(std::vector<IngestJob>& jobs) {
    for (IngestJob& j : jobs) {
        Column& c = *j.c;
        const uint64_t n = c.rows;
        if (c.where == Column::Arrow && n) {
            const uint64_t vptr = c.d_valid ? c.d_valid->ptr : 0;
            if (j.bitmap) H_TRY(golap_expand_validity(ctx, j.bitmap->ptr, j.bit_off, n, vptr, nullptr));
            if (c.dtype == DType::Utf8) {
                if (!j.large) H_TRY(golap_widen_to_8byte(ctx, j.offs32->ptr, GOLAP_WIDEN_I32, 0, n + 1, j.offs->ptr, nullptr));
                // (offsets are absolute positions in the Arrow data buffer; only [first, last) of it was uploaded)
                H_TRY(golap_hash_utf8(ctx, j.offs->ptr, j.raw->ptr - (uint64_t)j.str_first, vptr, n, c.d_data->ptr, nullptr));
            } else if (j.raw) {
                H_TRY(golap_widen_to_8byte(ctx, j.raw->ptr, j.widen, vptr, n, c.d_data->ptr, nullptr));
            } else if (vptr) {
                H_TRY(golap_widen_to_8byte(ctx, c.d_data->ptr, j.widen, vptr, n, c.d_data->ptr, nullptr));   // 8-byte values: only NULL -> 0, in place
            }
        }
        if (c.dtype == DType::Utf8) c.dtype = DType::Int64;      // from here on: the hash (DictEncodedString as i64, :93-97)
        c.where = Column::Device;
        c.hold.reset();
        c.a_values = c.a_bitmap = c.a_offsets = nullptr;
        c.fmt = 0;
        c.data.clear(); c.data.shrink_to_fit();
        c.in_valid.clear(); c.in_valid.shrink_to_fit();
    }
    return GOLAP_OK;
}

int32_t Exec::to_device(const std::vector<RecordBatch*>& bs) {
    std::vector<IngestJob> jobs;
    H_TRY(ingest_queue(bs, jobs));
    if (jobs.empty()) return GOLAP_OK;
    H_TRY(flush());
    return ingest_finish(jobs);
}

// gpu_buffers_to_record_batch (record_batch_convert.rs:136-178): the result leaves the device once -- values as they are, validity
// packed to the Arrow bitmap (and counted) on the device first; passed-through Arrow columns are copied out of their arrays
int32_t Exec::to_host(std::vector<RecordBatch>& bs) {
    // a string column that no operator touched still leaves as its hash (what the GPU path holds): it has to visit the device
    std::vector<RecordBatch*> strs;
    for (RecordBatch& b : bs) {
        bool any = false;
        for (Column& c : b.cols) any = any || (c.where == Column::Arrow && c.dtype == DType::Utf8);
        if (any) strs.push_back(&b);
    }
    if (!strs.empty()) H_TRY(to_device(strs));
    struct Job { Column* c; DevPtr bitmap, nulls; uint64_t null_count = 0; };
    std::vector<Job> jobs;
    for (RecordBatch& b : bs)
        for (Column& c : b.cols) {
            const uint64_t n = c.rows;
            if (c.where == Column::Arrow) {
                const ArrowFmt f = fmt_of(std::string(1, c.fmt).c_str());
                c.data.resize((size_t)n * (size_t)f.width);
                if (n) memcpy(c.data.data(), c.a_values + (size_t)c.off * (size_t)f.width, (size_t)n * (size_t)f.width);
                if (c.a_bitmap && n) {   // (a passed-through column never was on the device: its bitmap is re-based and its NULL slots zeroed here)
                    c.in_valid.resize((size_t)n);
                    for (uint64_t i = 0; i < n; ++i) {
                        const int64_t k = c.off + (int64_t)i;
                        const uint8_t ok = (c.a_bitmap[k >> 3] >> (k & 7)) & 1;
                        c.in_valid[i] = ok;
                        if (!ok) memset(c.data.data() + i * (size_t)f.width, 0, (size_t)f.width);   // null -> 0 (:102-110)
                    }
                }
                c.where = Column::Host;
                c.hold.reset();
                c.a_values = c.a_bitmap = c.a_offsets = nullptr;
            } else if (c.where == Column::Device) {
                Job j;
                j.c = &c;
                if (c.d_valid && n) {
                    H_TRY(temp(j.bitmap, (n + 7) / 8));
                    H_TRY(temp(j.nulls, 8));
                    H_TRY(golap_pack_validity(ctx, c.d_valid->ptr, n, j.bitmap->ptr, j.nulls->ptr, nullptr));
                }
                jobs.push_back(std::move(j));
            }
        }
    if (jobs.empty()) return GOLAP_OK;
    H_TRY(ksync());
    for (Job& j : jobs) {
        Column& c = *j.c;
        c.data.resize((size_t)c.rows * 8);
        down(c.d_data->ptr, c.data.data(), c.rows * 8);
        if (j.bitmap) {
            c.valid.resize((size_t)(c.rows + 7) / 8);
            down(j.bitmap->ptr, c.valid.data(), (c.rows + 7) / 8);
            down(j.nulls->ptr, &j.null_count, 8);
        }
    }
    H_TRY(flush());
    for (Job& j : jobs) {
        Column& c = *j.c;
        c.null_count = j.null_count;
        if (!c.null_count) c.valid.clear();
        c.where = Column::Host;
        c.d_data.reset();
        c.d_valid.reset();
    }
    return end_op();
}

// column ci[k] of batch bs[k], all batches, as one device array of n rows (values; validity bytes when any batch carries NULLs).
// One batch -- the usual case -- is used where it lies; several are copied together on the device.
int32_t Exec::concat(const std::vector<RecordBatch*>& bs, const std::vector<int>& ci, uint64_t n, Cat& out) {
    out = Cat();
    std::vector<size_t> live;
    bool any_valid = false;
    for (size_t k = 0; k < bs.size(); ++k)
        if (bs[k]->num_rows) { live.push_back(k); any_valid = any_valid || (bool)bs[k]->cols[ci[k]].d_valid; }
    if (live.empty()) return GOLAP_OK;
    if (live.size() == 1) {
        const Column& c = bs[live[0]]->cols[ci[live[0]]];
        out.ptr = c.d_data->ptr;
        out.vptr = c.d_valid ? c.d_valid->ptr : 0;
        return GOLAP_OK;
    }
    DevPtr d, v;
    H_TRY(temp(d, n * 8));
    if (any_valid) H_TRY(temp(v, n));
    uint64_t o = 0;
    for (size_t k : live) {
        const Column& c = bs[k]->cols[ci[k]];
        H_TRY(golap_device_copy(ctx, c.d_data->ptr, d->ptr + o * 8, c.rows * 8, nullptr));
        if (any_valid) {
            if (c.d_valid) H_TRY(golap_device_copy(ctx, c.d_valid->ptr, v->ptr + o, c.rows, nullptr));
            else H_TRY(golap_device_fill_u8(ctx, v->ptr + o, 1, c.rows, nullptr));
        }
        o += c.rows;
    }
    out.ptr = d->ptr;
    out.vptr = any_valid ? v->ptr : 0;
    return GOLAP_OK;
}

bool parse_i64(const std::string& s, int64_t& v) {
    if (s.empty()) return false;
    char* end = nullptr;
    errno = 0;
    long long x = strtoll(s.c_str(), &end, 10);
    if (errno || *end) return false;
    v = x;
    return true;
}

golap_cmp_op flip(golap_cmp_op op) {
    switch (op) {
        case GOLAP_CMP_LT: return GOLAP_CMP_GT;
        case GOLAP_CMP_LE: return GOLAP_CMP_GE;
        case GOLAP_CMP_GT: return GOLAP_CMP_LT;
        case GOLAP_CMP_GE: return GOLAP_CMP_LE;
        default: return op;
    }
}

// ---------------------------------------------------------------------------
// execute_filter, executor.rs:161-178 / eval_filter_mask :443-476 / filter_record_batch :478-494
// ---------------------------------------------------------------------------
// leaves of a conjunction: `a AND b AND ...` of <column> <op> <literal> comparisons (optimizer.rs:155-159 merges stacked filters with
// AND).  The reference's CPU mask knows no AND (eval_filter_mask, executor.rs:457-465: unknown operator -> no row); evaluating it
// with SQL semantics on the device is an extension -- one pass over the columns involved, every leaf with the reference's own edge
// cases (non-Int64 operands: that leaf passes every row; unknown comparison: no row).
// (only a node whose operands are themselves comparisons / conjunctions: `a AND 3` keeps the reference's reading -- an unknown
// operator between two Int64 operands, no row)
static bool is_and(const golap_expr& e) {
    return e.kind == golap_expr::Binary && (e.op == "AND" || e.op == "and" || e.op == "&&") && e.left && e.right && e.left->kind == golap_expr::Binary &&
           e.right->kind == golap_expr::Binary;
}
static void and_leaves(const golap_expr& e, std::vector<const golap_expr*>& out) {
    if (is_and(e) && e.left && e.right) { and_leaves(*e.left, out); and_leaves(*e.right, out); }
    else out.push_back(&e);
}

int32_t Exec::filter(std::vector<RecordBatch>& in, const golap_expr& pred, std::vector<RecordBatch>& out) {
    // what each batch needs: nothing (every row passes: :474, :449-452), or a conjunction of <Int64 column> <op> <literal> comparisons
    using Todo = FilterTodo;
    std::vector<Todo> todo(in.size());
    std::vector<RecordBatch*> upload;
    for (size_t bi = 0; bi < in.size(); ++bi) {
        RecordBatch& b = in[bi];
        Todo& td = todo[bi];
        if (pred.kind != golap_expr::Binary) continue;   // non-binary predicate => every row passes (:474)
        std::vector<const golap_expr*> leaves;
        and_leaves(pred, leaves);
        if (leaves.size() > GOLAP_FILTER_AND_MAX) return herr(GOLAP_ERR_UNSUPPORTED, "GpuFilter: at most " + std::to_string(GOLAP_FILTER_AND_MAX) + " comparisons in one conjunction");
        for (const golap_expr* lf : leaves) {
            if (lf->kind != golap_expr::Binary) continue;   // a non-binary leaf passes every row (:474)
            const golap_expr *l = lf->left.get(), *r = lf->right.get();
            golap_cmp_op op = golap_parse_cmp_op(lf->op.c_str());
            const golap_expr *colx = nullptr, *litx = nullptr;
            if (l->kind == golap_expr::Column && r->kind == golap_expr::Literal) { colx = l; litx = r; }
            else if (l->kind == golap_expr::Literal && r->kind == golap_expr::Column) { colx = r; litx = l; op = flip(op); }
            else return herr(GOLAP_ERR_UNSUPPORTED, "GpuFilter: only <column> <op> <literal> predicates (and conjunctions of them) run on the device");
            const int ci = resolve_column(*colx, b);
            if (ci < 0) return herr(GOLAP_ERR_INVALID_ARG, "Column '" + colx->name + "' not found in batch");   // :386
            int64_t lit;
            // operands that are not both Int64 => every row passes (:449-452, :471-472); a Utf8 column is its Int64 hash on this path (:93-97)
            if ((b.cols[ci].dtype != DType::Int64 && b.cols[ci].dtype != DType::Utf8) || !parse_i64(litx->value, lit)) continue;
            td.fps.push_back(golap_filter_predicate{(uint64_t)ci, (int32_t)op, lit});
            td.fcols.push_back(ci);
        }
        td.pass = td.fps.empty();
        if (!td.pass && b.num_rows) upload.push_back(&b);
    }
    // several batches, on request: batch k is filtered while batch k + 1 travels (filter_pipelined below; measured slower, see there)
    if (upload.size() >= 2 && g_pipeline.load()) return filter_pipelined(in, todo, out);
    // ONE transfer phase for every batch the predicate runs on (a phase per batch left the link idle while the previous batch was filtered)
    if (!upload.empty()) {
        H_TRY(need_ctx());
        H_TRY(to_device(upload));
    }
    for (size_t bi = 0; bi < in.size(); ++bi) {
        RecordBatch& b = in[bi];
        const Todo& td = todo[bi];
        if (td.pass) { out.push_back(std::move(b)); continue; }
        const std::vector<golap_filter_predicate>& fps = td.fps;
        const std::vector<int>& fcols = td.fcols;
        const uint64_t n = b.num_rows;
        if (n == 0) continue;   // empty batches are dropped (:173)
        DevPtr didx, dcnt;
        H_TRY(temp(didx, n * 4));
        H_TRY(temp(dcnt, 8));
        std::vector<uint64_t> kptrs(fps.size(), 0), vptrs(fps.size(), 0);
        for (size_t k = 0; k < fps.size(); ++k) {
            const Column& c = b.cols[fcols[k]];
            kptrs[k] = c.d_data->ptr;                           // (a range lo <= x AND x < hi names the same column twice: read once by the kernel)
            vptrs[k] = c.d_valid ? c.d_valid->ptr : 0;          // a comparison with NULL is not true: such rows do not pass
        }
        if (fps.size() == 1) H_TRY(golap_filter_compact_nullable(ctx, &fps[0], kptrs[0], vptrs[0], n, didx->ptr, dcnt->ptr, nullptr));
        else H_TRY(golap_filter_compact_and(ctx, fps.data(), kptrs.data(), vptrs.data(), (uint32_t)fps.size(), n, didx->ptr, dcnt->ptr, nullptr));
        uint64_t kept = 0;
        H_TRY(read_u64(dcnt->ptr, &kept));
        if (kept == 0) { H_TRY(end_op()); continue; }
        RecordBatch ob;
        ob.num_rows = kept;
        for (Column& c : b.cols) {      // take_int64 (:496-507) for every column -- values and, where the column has them, validity bytes: the row ids stay on the device
            Column oc;
            oc.name = c.name; oc.dtype = c.dtype; oc.rows = kept; oc.where = Column::Device;
            H_TRY(dev_alloc(oc.d_data, kept * 8));
            H_TRY(golap_take_8byte(ctx, c.d_data->ptr, didx->ptr, 0, kept, oc.d_data->ptr, nullptr));
            if (c.d_valid) {
                H_TRY(dev_alloc(oc.d_valid, kept));
                H_TRY(golap_take_1byte(ctx, c.d_valid->ptr, didx->ptr, 0, kept, oc.d_valid->ptr, nullptr));
            }
            ob.cols.push_back(std::move(oc));
        }
        out.push_back(std::move(ob));
        H_TRY(end_op());
        for (Column& c : b.cols) { c.d_data.reset(); c.d_valid.reset(); }   // the input batch's columns are not needed any more
    }
    return GOLAP_OK;
}

// GpuFilter over several batches, pipelined (golap_host_set_pipelining(1); OFF by default): the TransferQueue's double-buffered streams
// carry batch k + 1 (fed by a background thread that runs the batches' transfer lists in order, each with its own wait) while the compute
// stream expands / widens / filters / gathers batch k.  Nothing is read back per batch: the kept-row count stays in HBM
// (golap_take_8byte / golap_take_1byte take it as count_ptr), the output columns are provisioned for the batch's row count, and ALL counts
// come back in one download behind the last batch.
// MEASURED (bench.py --config OP [--pipelining], 100M orders in 4 batches, interleaved runs on one box, profiles/r04_final/op_pipelining_ab.log):
// 58.3 / 62.1 ms pipelined against 53.9 / 53.8 ms with one transfer phase for all batches -- the upload falls from 55 to 47-51 GB/s.
// Knowing that batch k has landed takes golap_xfer_synchronize, which drains EVERY stream: at each batch boundary the link waits for the
// last bounce buffers (8 MiB per stream) and then for the first staging memcpy of the next batch, ~2 ms per boundary, to hide a filter
// that is 0.3 ms of kernels per 25M-row batch on this chip.  The plan is bound by the PCIe link (46 of 54 ms); what the device does
// behind it -- the join, the pair sort, the gathers, the GROUP BY -- needs every batch anyway.  Kept as a tested option (both settings
// give the same rows: test_filter_batches_pipelined_and_in_one_phase, the operator fuzzer flips it per round), not as the default.
struct Uploader {
    const Exec& ex;
    std::vector<std::vector<Xfer>> lists;
    std::vector<int32_t> status;
    size_t done = 0;
    uint64_t up_bytes = 0;
    double busy_ms = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<bool> stop{false};
    std::thread th;
    explicit Uploader(const Exec& e) : ex(e) {}
    void start() {
        status.assign(lists.size(), GOLAP_OK);
        th = std::thread([this] {
            for (size_t i = 0; i < lists.size(); ++i) {
                int32_t s = GOLAP_OK;
                uint64_t up = 0, down = 0;
                const double t0 = now_ms();
                if (!stop.load()) s = ex.run_transfers(lists[i], up, down);
                const double dt = now_ms() - t0;
                std::lock_guard<std::mutex> lk(mu);
                status[i] = s;
                up_bytes += up;
                busy_ms += dt;
                done = i + 1;
                cv.notify_all();
            }
        });
    }
    int32_t wait(size_t i) {   // the transfers of list i have landed
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done > i; });
        return status[i];
    }
    void join() { if (th.joinable()) th.join(); }
    ~Uploader() { stop.store(true); join(); }   // (an operator that fails half-way: nothing is enqueued any more, what is in flight is waited for)
};

int32_t Exec::filter_pipelined(std::vector<RecordBatch>& in, const std::vector<FilterTodo>& todo, std::vector<RecordBatch>& out) {
    H_TRY(need_ctx());
    struct Slot { int list = -1; std::vector<IngestJob> jobs; DevPtr didx, dcnt; RecordBatch ob; uint64_t kept = 0; bool ran = false; };
    std::vector<Slot> slots(in.size());
    Uploader upl(*this);
    // allocations and transfer lists of every batch first (what one transfer phase for all of them allocated up front as well)
    for (size_t bi = 0; bi < in.size(); ++bi) {
        if (todo[bi].pass || in[bi].num_rows == 0) continue;
        std::vector<RecordBatch*> one{&in[bi]};
        H_TRY(ingest_queue(one, slots[bi].jobs));
        slots[bi].list = (int)upl.lists.size();
        upl.lists.emplace_back();
        upl.lists.back().swap(pending);
    }
    const double t_start = now_ms();
    upl.start();
    for (size_t bi = 0; bi < in.size(); ++bi) {
        RecordBatch& b = in[bi];
        const FilterTodo& td = todo[bi];
        Slot& sl = slots[bi];
        if (td.pass || b.num_rows == 0) continue;
        const uint64_t n = b.num_rows;
        H_TRY(upl.wait((size_t)sl.list));
        H_TRY(ingest_finish(sl.jobs));
        H_TRY(temp(sl.didx, n * 4));
        H_TRY(temp(sl.dcnt, 8));
        const size_t nf = td.fps.size();
        std::vector<uint64_t> kptrs(nf, 0), vptrs(nf, 0);
        for (size_t k = 0; k < nf; ++k) {
            const Column& c = b.cols[td.fcols[k]];
            kptrs[k] = c.d_data->ptr;
            vptrs[k] = c.d_valid ? c.d_valid->ptr : 0;          // a comparison with NULL is not true: such rows do not pass
        }
        if (nf == 1) H_TRY(golap_filter_compact_nullable(ctx, &td.fps[0], kptrs[0], vptrs[0], n, sl.didx->ptr, sl.dcnt->ptr, nullptr));
        else H_TRY(golap_filter_compact_and(ctx, td.fps.data(), kptrs.data(), vptrs.data(), (uint32_t)nf, n, sl.didx->ptr, sl.dcnt->ptr, nullptr));
        for (Column& c : b.cols) {      // take_int64 (:496-507) of at most n rows: the kernels read the count where the compaction left it
            Column oc;
            oc.name = c.name; oc.dtype = c.dtype; oc.rows = 0; oc.where = Column::Device;
            H_TRY(dev_alloc(oc.d_data, n * 8));
            H_TRY(golap_take_8byte(ctx, c.d_data->ptr, sl.didx->ptr, sl.dcnt->ptr, n, oc.d_data->ptr, nullptr));
            if (c.d_valid) {
                H_TRY(dev_alloc(oc.d_valid, n));
                H_TRY(golap_take_1byte(ctx, c.d_valid->ptr, sl.didx->ptr, sl.dcnt->ptr, n, oc.d_valid->ptr, nullptr));
            }
            sl.ob.cols.push_back(std::move(oc));
        }
        sl.ran = true;
    }
    upl.join();
    prof.h2d_bytes += upl.up_bytes;
    prof.h2d_ms += now_ms() - t_start;   // the upload phase's wall time; the kernels of all batches but the last ran inside it
    H_TRY(ksync());
    for (Slot& sl : slots)
        if (sl.ran) down(sl.dcnt->ptr, &sl.kept, 8);
    H_TRY(flush());
    for (size_t bi = 0; bi < in.size(); ++bi) {
        Slot& sl = slots[bi];
        if (todo[bi].pass) { out.push_back(std::move(in[bi])); continue; }
        if (!sl.ran || sl.kept == 0) continue;                  // empty batches are dropped (:173)
        sl.ob.num_rows = sl.kept;
        for (Column& c : sl.ob.cols) c.rows = sl.kept;
        out.push_back(std::move(sl.ob));
    }
    H_TRY(end_op());
    for (size_t bi = 0; bi < in.size(); ++bi)
        if (slots[bi].ran)
            for (Column& c : in[bi].cols) { c.d_data.reset(); c.d_valid.reset(); }   // the input batches' columns are not needed any more
    return GOLAP_OK;
}

// execute_projection, executor.rs:184-214 (column references and literals; arithmetic stays on the planner's CPU side)
int32_t Exec::projection(std::vector<RecordBatch>& in, const std::vector<std::unique_ptr<golap_expr>>& exprs, std::vector<RecordBatch>& out) {
    for (RecordBatch& b : in) {
        RecordBatch ob;
        ob.num_rows = b.num_rows;
        for (const auto& e : exprs) {
            if (e->kind == golap_expr::Column) {
                const int ci = resolve_column(*e, b);
                if (ci < 0) return herr(GOLAP_ERR_INVALID_ARG, "Column '" + e->name + "' not found in batch");
                Column c = b.cols[ci];                 // (device and Arrow buffers are shared, not copied)
                c.name = e->name;                      // expr_name, :531
                ob.cols.push_back(std::move(c));
            } else if (e->kind == golap_expr::Literal) {
                int64_t v;
                if (!parse_i64(e->value, v)) return herr(GOLAP_ERR_UNSUPPORTED, "Cannot parse literal '" + e->value + "'");   // :400
                Column c;
                c.name = "lit(" + e->value + ")";      // :532
                c.dtype = DType::Int64; c.rows = b.num_rows; c.where = Column::Host;
                c.data.resize(b.num_rows * 8);
                for (uint64_t i = 0; i < b.num_rows; ++i) memcpy(c.data.data() + i * 8, &v, 8);
                ob.cols.push_back(std::move(c));
            } else {
                return herr(GOLAP_ERR_UNSUPPORTED, "GpuProjection: arithmetic expressions are outside the accelerated path");
            }
        }
        out.push_back(std::move(ob));
    }
    return GOLAP_OK;
}

// ---------------------------------------------------------------------------
// execute_hash_join, executor.rs:220-265
// ---------------------------------------------------------------------------
// JoinKernel::choose_strategy (join_kernel.rs:71-77) looks at the build side only: <= 1M rows -> BroadcastHashJoin.  The operator knows the
// probe side too, and the broadcast join's global table only probes fast while it stays in an XCD's L2 (18 bytes per build row): with 1M
// build rows it takes 3.26 ms for 50M probe rows, the radix join 0.75 ms; with 30 000 build rows the two are level at 50M probe rows and
// broadcast wins below; with 1000 build rows broadcast wins everywhere (profiles/r04_final/strategy_ab.log).  So: the reference's rule,
// except that a build side beyond 65 536 rows goes through the radix join.  Same pairs either way (both strategies are tested against the
// oracle); the C ABI's golap_choose_strategy stays the reference's rule.
static golap_join_strategy pick_strategy(uint64_t n_build) {
    const golap_join_strategy s = golap_choose_strategy(n_build);
    return (s == GOLAP_JOIN_BROADCAST_HASH && n_build > 65536) ? GOLAP_JOIN_RADIX_HASH : s;
}

int32_t Exec::hash_join(std::vector<RecordBatch>& left, std::vector<RecordBatch>& right, golap_plan& p, std::vector<RecordBatch>& out) {
    if (left.empty() || right.empty()) return GOLAP_OK;                 // :231-233 -> vec![]
    if (p.left_keys.empty() || p.right_keys.empty()) return herr(GOLAP_ERR_INVALID_ARG, "GpuHashJoin needs one key per side");
    // The reference executor reads the FIRST key expression only (:245) although the plan carries vectors (physical_plan.rs:35-36);
    // here every key column takes part (up to 4), and NULL keys match nothing (SQL; the reference has turned them into 0 by now).
    const size_t nk = p.left_keys.size();
    if (p.right_keys.size() != nk) return herr(GOLAP_ERR_INVALID_ARG, "GpuHashJoin: left_keys and right_keys differ in length");
    if (nk > 4) return herr(GOLAP_ERR_UNSUPPORTED, "GpuHashJoin: at most 4 key columns");
    H_TRY(need_ctx());
    std::vector<RecordBatch*> lb, rb, all;
    for (RecordBatch& b : left) { lb.push_back(&b); all.push_back(&b); }
    for (RecordBatch& b : right) { rb.push_back(&b); all.push_back(&b); }
    // key columns, resolved per batch
    std::vector<std::vector<int>> lci(nk, std::vector<int>(lb.size())), rci(nk, std::vector<int>(rb.size()));
    for (size_t k = 0; k < nk; ++k) {
        const golap_expr &lk = *p.left_keys[k], &rk = *p.right_keys[k];
        if (lk.kind != golap_expr::Column || rk.kind != golap_expr::Column) return herr(GOLAP_ERR_UNSUPPORTED, "GpuHashJoin: key expressions must be column references");
        for (int side = 0; side < 2; ++side) {
            const std::vector<RecordBatch*>& bs = side ? rb : lb;
            const golap_expr& key = side ? rk : lk;
            for (size_t i = 0; i < bs.size(); ++i) {
                const int ci = resolve_column(key, *bs[i]);
                if (ci < 0) return herr(GOLAP_ERR_INVALID_ARG, "Column '" + key.name + "' not found in batch");
                const DType d = bs[i]->cols[ci].dtype;
                if (d != DType::Int64 && d != DType::Utf8) return herr(GOLAP_ERR_INVALID_ARG, "join key '" + key.name + "' must be Int64 (executor.rs:246)");
                (side ? rci : lci)[k][i] = ci;
            }
        }
    }
    // output schema: left fields then right fields
    RecordBatch ob;
    for (const Column& c : left[0].cols) { Column oc; oc.name = c.name; oc.dtype = c.dtype == DType::Utf8 ? DType::Int64 : c.dtype; ob.cols.push_back(oc); }
    const size_t n_left_cols = ob.cols.size();
    for (const Column& c : right[0].cols) { Column oc; oc.name = c.name; oc.dtype = c.dtype == DType::Utf8 ? DType::Int64 : c.dtype; ob.cols.push_back(oc); }
    for (const Column& c : ob.cols)
        if (!is_numeric(c.dtype)) return herr(GOLAP_ERR_UNSUPPORTED, "GpuHashJoin: payload column '" + c.name + "' has no device representation");
    for (int side = 0; side < 2; ++side)
        for (const RecordBatch* b : side ? rb : lb) {
            const RecordBatch& first = side ? right[0] : left[0];
            if (b->cols.size() != first.cols.size()) return herr(GOLAP_ERR_INVALID_ARG, "batches of one side differ in schema");
            for (size_t c = 0; c < b->cols.size(); ++c)
                if (is_float(b->cols[c].dtype) != is_float(first.cols[c].dtype)) return herr(GOLAP_ERR_INVALID_ARG, "batches of one side differ in schema");
        }
    uint64_t nb = 0, np = 0;   // global row index = running offset over batches (:249, :252)
    for (const RecordBatch* b : lb) nb += b->num_rows;
    for (const RecordBatch* b : rb) np += b->num_rows;
    H_TRY(to_device(all));
    uint64_t m = 0;
    DevPtr dpairs;
    const bool outer_keeps_something = (nb && (p.join_type == GOLAP_JOIN_LEFT || p.join_type == GOLAP_JOIN_FULL)) ||
                                       (np && (p.join_type == GOLAP_JOIN_RIGHT || p.join_type == GOLAP_JOIN_FULL));
    if ((nb && np) || outer_keeps_something) {
        std::vector<Cat> bk(nk), pk(nk);
        DevPtr dummy;
        H_TRY(temp(dummy, 8));
        for (size_t k = 0; k < nk; ++k) {
            H_TRY(concat(lb, lci[k], nb, bk[k]));
            H_TRY(concat(rb, rci[k], np, pk[k]));
            if (!nb) bk[k].ptr = dummy->ptr;     // (an empty side of an outer join: no row is read, the pointer just must not be null)
            if (!np) pk[k].ptr = dummy->ptr;
        }
        DevPtr dcnt;
        H_TRY(temp(dcnt, 8));
        uint64_t cap = std::max<uint64_t>(np + ((p.join_type & 1) ? nb : 0), 1);   // first guess; the count is the TRUE total (join_kernels.cuh:155-160)
        for (int attempt = 0; attempt < 2; ++attempt) {
            H_TRY(temp(dpairs, cap * 8));
            golap_join_params jp;
            memset(&jp, 0, sizeof jp);
            jp.build_key_ptr = bk[0].ptr; jp.probe_key_ptr = pk[0].ptr; jp.n_build = nb; jp.n_probe = np;
            jp.out_pairs_ptr = dpairs->ptr; jp.out_count_ptr = dcnt->ptr; jp.max_pairs = cap;
            jp.join_type = (int32_t)p.join_type;
            jp.n_key_cols = (uint32_t)nk;
            for (size_t k = 0; k < nk; ++k) {
                if (k) { jp.build_key_extra[k - 1] = bk[k].ptr; jp.probe_key_extra[k - 1] = pk[k].ptr; }
                jp.build_validity[k] = bk[k].vptr;
                jp.probe_validity[k] = pk[k].vptr;
            }
            H_TRY(golap_join(ctx, pick_strategy(nb), &jp, nullptr));   // JoinKernel::choose_strategy, join_kernel.rs:71-77, refined above
            H_TRY(read_u64(dcnt->ptr, &m));
            // golap.h (golap_join_params): 2^64 - 1 is no count -- the verification of a surrogate-key join met more false pairs than it
            // can repair (probability < 1e-60 with 64-bit surrogates; the test hook that narrows them gets here)
            if (m == ~0ull) return herr(GOLAP_ERR_UNSUPPORTED, "GpuHashJoin: more colliding surrogate keys than the pair verification can repair");
            if (m <= cap) break;
            cap = m;   // too small: retry with the exact size (the first buffer goes back to the pool at the end of the operator)
        }
    }
    ob.num_rows = m;
    for (Column& c : ob.cols) { c.rows = m; c.where = m ? Column::Device : Column::Host; }
    if (m) {
        // canonical order (probe_rid, build_rid): the device order is unspecified (join_kernels.cuh:155) -- sorted, split into one index
        // vector per side and used for the gathers without leaving the device.  (Leaving the sort out where a GROUP BY hides the order
        // gains nothing: the radix join emits pairs in sub-partition order, and the five payload gathers of the OP plan then take 4.3
        // instead of 2.2 ms -- exactly the sort's 2.2 ms; profiles/r04_final/op_breakdown_unsorted_pairs.log.)
        H_TRY(golap_sort_u64(ctx, dpairs->ptr, m, nullptr));
        DevPtr dl, dr;
        H_TRY(temp(dl, m * 4));
        H_TRY(temp(dr, m * 4));
        H_TRY(golap_pairs_unzip(ctx, dpairs->ptr, m, dl->ptr, dr->ptr, nullptr));
        for (size_t c = 0; c < ob.cols.size(); ++c) {
            const bool is_left = c < n_left_cols;
            const std::vector<RecordBatch*>& side = is_left ? lb : rb;
            const int sc = (int)(is_left ? c : c - n_left_cols);
            const uint64_t rows = is_left ? nb : np;
            Cat src;
            H_TRY(concat(side, std::vector<int>(side.size(), sc), rows, src));
            const uint64_t idx = is_left ? dl->ptr : dr->ptr;
            H_TRY(dev_alloc(ob.cols[c].d_data, m * 8));
            if (rows) H_TRY(golap_take_8byte(ctx, src.ptr, idx, 0, m, ob.cols[c].d_data->ptr, nullptr));
            else H_TRY(golap_device_fill_u8(ctx, ob.cols[c].d_data->ptr, 0, m * 8, nullptr));
            // rows whose source side is missing (NULL row id of an outer join) or whose source row is NULL become Arrow nulls
            const bool missing_side = is_left ? (p.join_type & 2) != 0 : (p.join_type & 1) != 0;
            if (src.vptr || missing_side) {
                H_TRY(dev_alloc(ob.cols[c].d_valid, m));
                H_TRY(golap_take_1byte(ctx, src.vptr, idx, 0, m, ob.cols[c].d_valid->ptr, nullptr));
            }
        }
    }
    out.push_back(std::move(ob));   // one batch, like the reference's single output batch (:262-264)
    return end_op();
}

static const char* agg_name(int32_t f) {
    switch (f) { case GOLAP_AGG_SUM: return "sum"; case GOLAP_AGG_COUNT: return "count"; case GOLAP_AGG_MIN: return "min"; case GOLAP_AGG_MAX: return "max"; default: return "avg"; }
}

// ---------------------------------------------------------------------------
// execute_aggregate, executor.rs:297-355 (+ eval_agg_input :509-527)
// ---------------------------------------------------------------------------
// GROUP BY (physical_plan.rs:52-54: group_by is a Vec<PhysicalExpr>; lowering :157-172 keeps every column; the reference's executor
// ignores it, executor.rs:307).  Key columns are Int64 / Int32 (and string hashes), so a row's key -- one column or several, with or
// without NULLs -- is a mixed-radix number: digit_c = value - min_c (NULL: one past the column's largest digit -- NULLs sort last, and
// (NULL, 1) is a group of its own), weight_c = the product of the later columns' digit counts.  While that number fits 62 bits --
// dictionary-like ids, dates, small codes: the usual GROUP BY columns -- it IS a single Int64 key without NULLs: exact, no collisions.
// Ranges come from golap_key_minmax (exact int64, NULL-aware), the packing is golap_pack_group_keys, both on the device; a single
// key column without NULLs is used as it is.  The group table is unpacked into the key columns again on the host (result-sized).
int32_t Exec::aggregate(std::vector<RecordBatch>& in, golap_plan& p, std::vector<RecordBatch>& out) {
    const uint32_t na = (uint32_t)p.aggr_inputs.size();
    const size_t K = p.group_by.size();
    if (K > 4) return herr(GOLAP_ERR_UNSUPPORTED, "GpuAggregate: at most 4 group-by columns");
    H_TRY(need_ctx());
    std::vector<RecordBatch*> bs;
    for (RecordBatch& b : in) bs.push_back(&b);
    uint64_t n = 0;
    for (const RecordBatch* b : bs) n += b->num_rows;
    // the distinct value columns the aggregates read, and the key columns, resolved per batch
    std::vector<std::string> val_names;
    std::vector<const golap_expr*> val_expr;
    std::vector<uint64_t> spec_col(na);
    for (uint32_t i = 0; i < na; ++i) {
        const golap_expr& e = *p.aggr_inputs[i];
        if (e.kind != golap_expr::Column) return herr(GOLAP_ERR_UNSUPPORTED, "GpuAggregate: aggregate inputs must be column references");
        auto it = std::find(val_names.begin(), val_names.end(), e.name);
        if (it == val_names.end()) { val_names.push_back(e.name); val_expr.push_back(&e); spec_col[i] = val_names.size() - 1; }
        else spec_col[i] = (uint64_t)(it - val_names.begin());
    }
    const uint32_t nv = (uint32_t)val_names.size();
    std::vector<std::vector<int>> vci(nv, std::vector<int>(bs.size())), kci(K, std::vector<int>(bs.size()));
    std::vector<int32_t> vdt(std::max<uint32_t>(nv, 1), GOLAP_DT_INT64);
    for (uint32_t v = 0; v < nv; ++v)
        for (size_t i = 0; i < bs.size(); ++i) {
            const int ci = resolve_column(*val_expr[v], *bs[i]);
            if (ci < 0) return herr(GOLAP_ERR_INVALID_ARG, "Column '" + val_names[v] + "' not found in batch");
            const Column& c = bs[i]->cols[ci];
            if (!is_numeric(c.dtype) || c.dtype == DType::Utf8)
                return herr(GOLAP_ERR_UNSUPPORTED, "GpuAggregate: input '" + c.name + "' must be Int64 or Float64 (executor.rs:520-526)");
            vdt[v] = is_float(c.dtype) ? GOLAP_DT_FLOAT64 : GOLAP_DT_INT64;
            vci[v][i] = ci;
        }
    for (size_t k = 0; k < K; ++k) {
        if (p.group_by[k]->kind != golap_expr::Column) return herr(GOLAP_ERR_UNSUPPORTED, "GpuAggregate: group-by expressions must be column references");
        for (size_t i = 0; i < bs.size(); ++i) {
            const int ci = resolve_column(*p.group_by[k], *bs[i]);
            if (ci < 0) return herr(GOLAP_ERR_INVALID_ARG, "Column '" + p.group_by[k]->name + "' not found in batch");
            const DType d = bs[i]->cols[ci].dtype;
            if (is_float(d) || !is_numeric(d)) return herr(GOLAP_ERR_UNSUPPORTED, "GpuAggregate: group key must be Int64 or Int32");
            kci[k][i] = ci;
        }
    }
    H_TRY(to_device(bs));
    std::vector<Cat> vals(nv);
    std::vector<uint64_t> vptrs(std::max<uint32_t>(nv, 1), 0), vvptrs(std::max<uint32_t>(nv, 1), 0);   // vvptrs: validity of the value columns (0 = that column has no NULLs)
    bool any_value_nulls = false;
    for (uint32_t v = 0; v < nv; ++v) {
        H_TRY(concat(bs, vci[v], n, vals[v]));
        vptrs[v] = vals[v].ptr;
        vvptrs[v] = vals[v].vptr;
        any_value_nulls = any_value_nulls || vals[v].vptr;
    }
    std::vector<golap_aggregate_spec> specs(na);
    for (uint32_t i = 0; i < na; ++i) specs[i] = golap_aggregate_spec{p.aggr_funcs[i], spec_col[i], i};
    auto agg_col = [&](uint32_t i) {
        Column c;
        c.name = std::string(agg_name(p.aggr_funcs[i])) + "(" + p.aggr_inputs[i]->name + ")";
        c.dtype = DType::Float64; c.where = Column::Host;
        return c;
    };
    RecordBatch ob;
    golap_aggregate_params ap;
    if (K == 0) {
        // the reference's single-group result: one Float64 scalar per aggregate (:330-349)
        DevPtr dout;
        H_TRY(temp(dout, (uint64_t)std::max<uint32_t>(na, 1) * 8));
        memset(&ap, 0, sizeof ap);
        ap.value_ptrs = vptrs.data(); ap.n_value_cols = nv; ap.n_rows = n; ap.n_groups = 1; ap.out_ptr = dout->ptr;
        ap.value_validity_ptrs = any_value_nulls ? vvptrs.data() : nullptr;   // SQL aggregates skip NULL values
        ap.key_dtype = GOLAP_DT_INT64; ap.value_dtypes = vdt.data();
        H_TRY(golap_aggregate(ctx, specs.data(), na, &ap, nullptr));
        std::vector<double> res(std::max<uint32_t>(na, 1));
        H_TRY(ksync());
        down(dout->ptr, res.data(), (uint64_t)na * 8);
        H_TRY(flush());
        ob.num_rows = 1;
        for (uint32_t i = 0; i < na; ++i) {
            Column c = agg_col(i);
            c.rows = 1;
            c.data.resize(8);
            memcpy(c.data.data(), &res[i], 8);
            ob.cols.push_back(std::move(c));
        }
        out.push_back(std::move(ob));
        return end_op();
    }
    // ---- the key: ranges, then (unless it is one plain column) the packed column ----
    std::vector<Column> kcs(K);
    for (size_t k = 0; k < K; ++k) { kcs[k].name = p.group_by[k]->name; kcs[k].dtype = DType::Int64; kcs[k].where = Column::Host; }
    std::vector<Column> acs;
    for (uint32_t i = 0; i < na; ++i) acs.push_back(agg_col(i));
    auto finish = [&]() {
        for (auto& c : kcs) { c.rows = ob.num_rows; ob.cols.push_back(std::move(c)); }
        for (auto& c : acs) { c.rows = ob.num_rows; ob.cols.push_back(std::move(c)); }
        out.push_back(std::move(ob));
        return end_op();
    };
    ob.num_rows = 0;
    if (n == 0) return finish();
    std::vector<Cat> keys(K);
    DevPtr dmm;
    H_TRY(temp(dmm, K * 24));
    for (size_t k = 0; k < K; ++k) {
        H_TRY(concat(bs, kci[k], n, keys[k]));
        H_TRY(golap_key_minmax(ctx, keys[k].ptr, keys[k].vptr, n, dmm->ptr + k * 24, nullptr));
    }
    std::vector<int64_t> mm(K * 3);
    H_TRY(read_u64(dmm->ptr, (uint64_t*)mm.data(), (uint32_t)(K * 3)));
    std::vector<int64_t> mn(K), mx(K);
    std::vector<bool> has_null(K);
    bool any_key_null = false;
    for (size_t k = 0; k < K; ++k) {
        has_null[k] = (uint64_t)mm[3 * k + 2] < n;
        any_key_null = any_key_null || has_null[k];
        mn[k] = mm[3 * k]; mx[k] = mm[3 * k + 1];
        if (mm[3 * k + 2] == 0) mn[k] = mx[k] = 0;   // no valid value at all
    }
    const bool packed = K > 1 || any_key_null;
    // digit counts and weights (first column most significant: rows come out sorted by (k0, k1, ...), NULLS LAST per column)
    std::vector<unsigned __int128> digits(K);
    std::vector<uint64_t> weight(K, 1);
    unsigned __int128 total = 1;
    bool wide = false;   // the key tuple does not fit one exact int64: group on a 64-bit surrogate, then verify (below)
    if (packed)
        for (size_t c = K; c-- > 0;) {
            digits[c] = (unsigned __int128)((__int128)mx[c] - (__int128)mn[c]) + 1 + (has_null[c] ? 1 : 0);
            weight[c] = (uint64_t)total;
            total *= digits[c];
            if (total > ((unsigned __int128)1 << 62)) { wide = true; break; }
        }
    if (getenv("GOLAP_DEBUG_ENV"))   // (test hook, honoured like the library's knobs only in a debugging environment: the surrogate route for keys that would pack)
        if (const char* e = getenv("GOLAP_HOST_WIDE_KEYS")) wide = wide || (packed && e[0] == '1');
    // the hashed GROUP BY (GOLAP_KEYS_HASHED, aggregate_kernel.rs:76-77) on `hkey`; the bound on the distinct keys starts modest and
    // grows only when the data asks for it: every buffer is sized from it.  Leaves the group count in nd, the tables in dout / dkeys.
    struct Hashed { DevPtr dout, dcnt, dkeys, dn; uint64_t bound = 0; uint32_t nd = 0; };
    auto run_hashed = [&](uint64_t hkey, uint64_t slot_ids_ptr, Hashed& h) -> int32_t {
        h.bound = std::min<uint64_t>(n, 1ull << 22);
        const uint64_t bound_max = std::min<uint64_t>(n, 1ull << 28);   // distinct keys <= rows; the key table holds at most 2^28 groups
        int32_t kd64 = GOLAP_DT_INT64;
        for (;;) {
            H_TRY(temp(h.dout, (uint64_t)std::max<uint32_t>(na, 1) * h.bound * 8));
            H_TRY(temp(h.dcnt, h.bound * 8));
            H_TRY(temp(h.dkeys, h.bound * 8));
            H_TRY(temp(h.dn, 8));
            golap_aggregate_params hp;
            memset(&hp, 0, sizeof hp);
            hp.group_key_ptr = hkey; hp.value_ptrs = vptrs.data(); hp.n_value_cols = nv; hp.n_rows = n; hp.n_groups = (uint32_t)h.bound;
            hp.value_validity_ptrs = any_value_nulls ? vvptrs.data() : nullptr;
            hp.out_ptr = h.dout->ptr; hp.key_dtype = GOLAP_DT_INT64; hp.value_dtypes = vdt.data(); hp.out_count_ptr = h.dcnt->ptr;
            hp.key_mode = GOLAP_KEYS_HASHED; hp.out_keys_ptr = h.dkeys->ptr; hp.out_n_groups_ptr = h.dn->ptr;
            hp.out_slot_ids_ptr = slot_ids_ptr;
            if (na == 0) {
                golap_aggregate_spec cs{GOLAP_AGG_COUNT, 0, 0};
                hp.value_ptrs = &hkey; hp.n_value_cols = 1; hp.value_dtypes = &kd64; hp.value_validity_ptrs = nullptr;
                H_TRY(golap_aggregate(ctx, &cs, 1, &hp, nullptr));
            } else {
                H_TRY(golap_aggregate(ctx, specs.data(), na, &hp, nullptr));
            }
            uint64_t nd64 = 0;
            H_TRY(read_u64(h.dn->ptr, &nd64));
            h.nd = (uint32_t)nd64;
            if (h.nd <= h.bound) return GOLAP_OK;
            // (0xFFFFFFFF: hopelessly small -- the insert pass gave up early)
            if (h.bound >= bound_max) return herr(GOLAP_ERR_UNSUPPORTED, "GpuAggregate: more than 2^28 distinct group keys");
            h.bound = std::min<uint64_t>(bound_max, h.nd == 0xFFFFFFFFu ? h.bound * 16 : std::max<uint64_t>(h.bound * 4, (uint64_t)h.nd + h.nd / 8));
            for (DevPtr* d : {&h.dout, &h.dcnt, &h.dkeys, &h.dn}) drop(*d);   // (the device is idle: the count was just read)
        }
    };
    if (wide) {
        // ---- key tuples beyond one int64 (hash-like ids in several columns; a full-range column with NULLs): GROUP BY on a 64-bit
        //      surrogate of the tuple, every row then CHECKED against its group's representative row on the device; a surrogate shared
        //      by two tuples (probability ~ groups^2 / 2^65) shows up as a mismatch and the grouping is repeated under another seed.
        //      The result is exact or an error -- never a silently merged group. ----
        golap_pack_keys_params kp;
        memset(&kp, 0, sizeof kp);
        kp.n_cols = (uint32_t)K; kp.n_rows = n;
        for (size_t c = 0; c < K; ++c) { kp.col_ptrs[c] = keys[c].ptr; kp.validity_ptrs[c] = keys[c].vptr; }
        DevPtr dsur, dslots, drep, dbad;
        H_TRY(temp(dsur, n * 8));
        H_TRY(temp(dslots, n * 4));
        H_TRY(temp(dbad, 8));
        kp.out_ptr = dsur->ptr;
        Hashed h;
        const uint64_t seeds[3] = {0x6A09E667F3BCC908ull, 0xBB67AE8584CAA73Bull, 0x3C6EF372FE94F82Bull};
        uint64_t bad = 1;
        for (int attempt = 0; attempt < 3 && bad; ++attempt) {
            if (attempt) { for (DevPtr* d : {&h.dout, &h.dcnt, &h.dkeys, &h.dn, &drep}) drop(*d); }
            H_TRY(golap_hash_key_columns(ctx, &kp, seeds[attempt], nullptr));
            H_TRY(run_hashed(dsur->ptr, dslots->ptr, h));
            H_TRY(temp(drep, (uint64_t)std::max<uint32_t>(h.nd, 1) * 4));
            H_TRY(golap_group_keys_verify(ctx, &kp, dslots->ptr, h.nd, drep->ptr, dbad->ptr, nullptr));
            H_TRY(read_u64(dbad->ptr, &bad));
        }
        if (bad) return herr(GOLAP_ERR_UNSUPPORTED, "GpuAggregate: three different 64-bit surrogates of the composite group key all collided (" + std::to_string(bad) + " rows)");
        const uint32_t nd = h.nd;
        // the groups' key tuples = the representative rows' (gathered on the device), the aggregates = the slots'
        std::vector<std::vector<int64_t>> gk(K, std::vector<int64_t>(nd));
        std::vector<std::vector<uint8_t>> gv(K);
        std::vector<std::vector<double>> res(std::max<uint32_t>(na, 1), std::vector<double>(nd));
        std::vector<DevPtr> dk(K), dv(K);
        for (size_t c = 0; c < K && nd; ++c) {
            H_TRY(temp(dk[c], (uint64_t)nd * 8));
            H_TRY(golap_take_8byte(ctx, keys[c].ptr, drep->ptr, 0, nd, dk[c]->ptr, nullptr));
            if (keys[c].vptr) {
                H_TRY(temp(dv[c], nd));
                H_TRY(golap_take_1byte(ctx, keys[c].vptr, drep->ptr, 0, nd, dv[c]->ptr, nullptr));
                gv[c].resize(nd);
            }
        }
        H_TRY(ksync());
        for (size_t c = 0; c < K && nd; ++c) {
            down(dk[c]->ptr, gk[c].data(), (uint64_t)nd * 8);
            if (dv[c]) down(dv[c]->ptr, gv[c].data(), nd);
        }
        for (uint32_t i = 0; i < na; ++i) down(h.dout->ptr + (uint64_t)i * h.bound * 8, res[i].data(), (uint64_t)nd * 8);
        H_TRY(flush());
        std::vector<uint32_t> order(nd);
        for (uint32_t i = 0; i < nd; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {     // rows sorted by (k0, k1, ...), NULLS LAST per column
            for (size_t c = 0; c < K; ++c) {
                const bool na_ = !gv[c].empty() && !gv[c][a], nb_ = !gv[c].empty() && !gv[c][b];
                if (na_ != nb_) return nb_;
                if (!na_ && gk[c][a] != gk[c][b]) return gk[c][a] < gk[c][b];
            }
            return false;
        });
        for (uint32_t slot : order) {
            for (size_t c = 0; c < K; ++c) {
                const bool is_null = !gv[c].empty() && !gv[c][slot];
                const int64_t v = is_null ? 0 : gk[c][slot];
                kcs[c].data.insert(kcs[c].data.end(), (const uint8_t*)&v, (const uint8_t*)&v + 8);
                if (has_null[c]) kcs[c].in_valid.push_back(is_null ? 0 : 1);
            }
            for (uint32_t i = 0; i < na; ++i) acs[i].data.insert(acs[i].data.end(), (const uint8_t*)&res[i][slot], (const uint8_t*)&res[i][slot] + 8);
            ob.num_rows++;
        }
        return finish();
    }
    uint64_t key_ptr = keys[0].ptr;
    int64_t kmin = mn[0], kmax = mx[0];
    if (packed) {
        DevPtr dpk;
        H_TRY(temp(dpk, n * 8));
        golap_pack_keys_params pk;
        memset(&pk, 0, sizeof pk);
        pk.n_cols = (uint32_t)K; pk.n_rows = n; pk.out_ptr = dpk->ptr;
        for (size_t c = 0; c < K; ++c) {
            pk.col_ptrs[c] = keys[c].ptr; pk.validity_ptrs[c] = keys[c].vptr; pk.min_value[c] = mn[c]; pk.weight[c] = weight[c];
            pk.null_digit[c] = (uint64_t)(digits[c] - 1);
        }
        H_TRY(golap_pack_group_keys(ctx, &pk, nullptr));
        key_ptr = dpk->ptr;
        kmin = 0;
        kmax = (int64_t)(uint64_t)(total - 1);
    }
    // one group key (packed or plain) -> key columns of the result, appended in the order the caller walks the groups
    auto emit_key = [&](int64_t key) {
        for (size_t c = 0; c < K; ++c) {
            int64_t v = key;
            uint8_t ok = 1;
            if (packed) {
                const uint64_t dg = (uint64_t)(((unsigned __int128)(uint64_t)key / weight[c]) % digits[c]);
                if (has_null[c] && dg == (uint64_t)(digits[c] - 1)) { v = 0; ok = 0; }
                else v = mn[c] + (int64_t)dg;
            }
            kcs[c].data.insert(kcs[c].data.end(), (const uint8_t*)&v, (const uint8_t*)&v + 8);
            if (has_null[c]) kcs[c].in_valid.push_back(ok);
        }
    };
    int32_t kd = GOLAP_DT_INT64;
    // dense keys (non-negative, range not much wider than the row count) index the group table directly (slot = key % G,
    // join_kernels.cuh:230); anything else -- negative, sparse, hash-like keys -- goes through the key -> slot hash table
    // (GOLAP_KEYS_HASHED, aggregate_kernel.rs:76-77)
    const unsigned __int128 span = (unsigned __int128)((__int128)kmax - (__int128)kmin);
    const bool dense_keys = kmin >= 0 && span < (unsigned __int128)(64ull << 20) && span <= (unsigned __int128)(4 * n + 1024);
    if (!dense_keys) {
        Hashed h;
        H_TRY(run_hashed(key_ptr, 0, h));
        const uint32_t nd = h.nd;
        std::vector<int64_t> gk(nd);
        std::vector<std::vector<double>> res(std::max<uint32_t>(na, 1), std::vector<double>(nd));   // sized from the groups found, not from the bound
        down(h.dkeys->ptr, gk.data(), (uint64_t)nd * 8);
        for (uint32_t i = 0; i < na; ++i) down(h.dout->ptr + (uint64_t)i * h.bound * 8, res[i].data(), (uint64_t)nd * 8);
        H_TRY(flush());
        std::vector<uint32_t> order(nd);
        for (uint32_t i = 0; i < nd; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return gk[a] < gk[b]; });   // rows sorted by key
        for (uint32_t slot : order) {
            emit_key(gk[slot]);
            for (uint32_t i = 0; i < na; ++i) acs[i].data.insert(acs[i].data.end(), (const uint8_t*)&res[i][slot], (const uint8_t*)&res[i][slot] + 8);
            ob.num_rows++;
        }
        return finish();
    }
    const uint64_t G = (uint64_t)(kmax - kmin) + 1;
    DevPtr dout, dcnt;
    H_TRY(temp(dout, (uint64_t)std::max<uint32_t>(na, 1) * G * 8));
    H_TRY(temp(dcnt, G * 8));
    memset(&ap, 0, sizeof ap);
    ap.group_key_ptr = key_ptr; ap.value_ptrs = vptrs.data(); ap.n_value_cols = nv; ap.n_rows = n; ap.n_groups = (uint32_t)G;
    ap.value_validity_ptrs = any_value_nulls ? vvptrs.data() : nullptr;
    ap.out_ptr = dout->ptr; ap.key_dtype = GOLAP_DT_INT64; ap.value_dtypes = vdt.data(); ap.out_count_ptr = dcnt->ptr;
    if (na == 0) {   // GROUP BY without aggregates: count only, to find the present groups
        golap_aggregate_spec cs{GOLAP_AGG_COUNT, 0, 0};
        ap.value_ptrs = &key_ptr; ap.n_value_cols = 1; ap.value_dtypes = &kd; ap.value_validity_ptrs = nullptr;
        H_TRY(golap_aggregate(ctx, &cs, 1, &ap, nullptr));
    } else {
        H_TRY(golap_aggregate(ctx, specs.data(), na, &ap, nullptr));
        if (any_value_nulls) {
            // out_count counted the non-NULL values of the last column; which groups EXIST is a question about rows: count the
            // key column (no NULLs) once more into dcnt (a group whose values are all NULL is still a group)
            DevPtr dtmp;
            H_TRY(temp(dtmp, G * 8));
            golap_aggregate_spec cs{GOLAP_AGG_COUNT, 0, 0};
            golap_aggregate_params cp = ap;
            cp.value_ptrs = &key_ptr; cp.n_value_cols = 1; cp.value_dtypes = &kd; cp.value_validity_ptrs = nullptr; cp.out_ptr = dtmp->ptr;
            H_TRY(golap_aggregate(ctx, &cs, 1, &cp, nullptr));
        }
    }
    std::vector<double> res((size_t)std::max<uint32_t>(na, 1) * G);
    std::vector<uint64_t> cnt(G);
    H_TRY(ksync());
    down(dout->ptr, res.data(), res.size() * 8);
    down(dcnt->ptr, cnt.data(), G * 8);
    H_TRY(flush());
    for (int64_t k = kmin; k <= kmax; ++k) {           // rows sorted by key
        const uint64_t slot = (uint64_t)k % G;
        if (!cnt[slot]) continue;
        emit_key(k);
        for (uint32_t i = 0; i < na; ++i) {
            const double v = res[(size_t)i * G + slot];
            acs[i].data.insert(acs[i].data.end(), (const uint8_t*)&v, (const uint8_t*)&v + 8);
        }
        ob.num_rows++;
    }
    return finish();
}

// GpuExecutor::execute, executor.rs:37-104: children fully materialised before the parent runs
int32_t Exec::run(golap_plan& p, std::vector<RecordBatch>& out) {
    switch (p.kind) {
        case golap_plan::Batches:
        case golap_plan::TableScan:
            out = std::move(p.batches);
            return GOLAP_OK;
        case golap_plan::Filter: {
            std::vector<RecordBatch> in;
            H_TRY(run(*p.input, in));
            return filter(in, *p.predicate, out);
        }
        case golap_plan::Projection: {
            std::vector<RecordBatch> in;
            H_TRY(run(*p.input, in));
            return projection(in, p.exprs, out);
        }
        case golap_plan::HashJoin: {
            std::vector<RecordBatch> l, r;
            H_TRY(run(*p.input, l));
            H_TRY(run(*p.right, r));
            return hash_join(l, r, p, out);
        }
        case golap_plan::Aggregate: {
            std::vector<RecordBatch> in;
            H_TRY(run(*p.input, in));
            return aggregate(in, p, out);
        }
        case golap_plan::Sort:
            return run(*p.input, out);   // executor.rs:361-370: pass-through
    }
    return herr(GOLAP_ERR_INVALID_ARG, "unknown plan node");
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

golap_expr* golap_expr_column(uint64_t index, const char* name) {
    auto* e = new golap_expr();
    e->kind = golap_expr::Column; e->index = index; e->name = name ? name : "";
    return e;
}
golap_expr* golap_expr_literal(const char* value) {
    auto* e = new golap_expr();
    e->kind = golap_expr::Literal; e->value = value ? value : "";
    return e;
}
golap_expr* golap_expr_binary(golap_expr* left, const char* op, golap_expr* right) {
    auto* e = new golap_expr();
    e->kind = golap_expr::Binary; e->op = op ? op : "";
    e->left.reset(left); e->right.reset(right);
    return e;
}
void golap_expr_free(golap_expr* e) { delete e; }

golap_plan* golap_plan_batches(ArrowArray** arrays, ArrowSchema** schemas, uint32_t n_batches) {
    auto p = std::make_unique<golap_plan>();
    p->kind = golap_plan::Batches;
    bool ok = true;
    std::string err;
    for (uint32_t i = 0; i < n_batches; ++i) {
        RecordBatch b;
        if (ok && !import_batch(arrays[i], schemas[i], b, err)) ok = false;
        if (ok) p->batches.push_back(std::move(b));
        if (arrays[i] && arrays[i]->release) arrays[i]->release(arrays[i]);      // (not imported: released here -- the arrays are consumed either way)
        if (schemas[i] && schemas[i]->release) schemas[i]->release(schemas[i]);
    }
    if (!ok) { g_err = err; return nullptr; }
    return p.release();
}

golap_plan* golap_plan_table_scan(const char* table_name, ArrowSchema* schema) {
    (void)table_name;
    auto p = std::make_unique<golap_plan>();
    p->kind = golap_plan::TableScan;
    RecordBatch b;   // executor.rs:138-154: one empty batch; Float32/Float64 fields -> Float64, everything else -> Int64
    if (schema && schema->format && !strcmp(schema->format, "+s")) {
        for (int64_t c = 0; c < schema->n_children; ++c) {
            Column col;
            col.name = schema->children[c]->name ? schema->children[c]->name : "";
            const char* f = schema->children[c]->format;
            col.dtype = (f && (!strcmp(f, "g") || !strcmp(f, "f"))) ? DType::Float64 : DType::Int64;
            col.where = Column::Host;
            b.cols.push_back(col);
        }
    }
    if (schema && schema->release) schema->release(schema);
    p->batches.push_back(std::move(b));
    return p.release();
}

golap_plan* golap_plan_filter(golap_plan* input, golap_expr* predicate) {
    auto* p = new golap_plan();
    p->kind = golap_plan::Filter; p->input.reset(input); p->predicate.reset(predicate);
    return p;
}
golap_plan* golap_plan_projection(golap_plan* input, golap_expr** exprs, uint32_t n) {
    auto* p = new golap_plan();
    p->kind = golap_plan::Projection; p->input.reset(input);
    for (uint32_t i = 0; i < n; ++i) p->exprs.emplace_back(exprs[i]);
    return p;
}
golap_plan* golap_plan_hash_join(golap_plan* left, golap_plan* right, golap_expr** lk, uint32_t nl, golap_expr** rk, uint32_t nr, golap_join_type jt) {
    auto* p = new golap_plan();
    p->kind = golap_plan::HashJoin; p->input.reset(left); p->right.reset(right); p->join_type = jt;
    for (uint32_t i = 0; i < nl; ++i) p->left_keys.emplace_back(lk[i]);
    for (uint32_t i = 0; i < nr; ++i) p->right_keys.emplace_back(rk[i]);
    return p;
}
golap_plan* golap_plan_aggregate(golap_plan* input, golap_expr** group_by, uint32_t ng, const int32_t* funcs, golap_expr** inputs, uint32_t na) {
    auto* p = new golap_plan();
    p->kind = golap_plan::Aggregate; p->input.reset(input);
    for (uint32_t i = 0; i < ng; ++i) p->group_by.emplace_back(group_by[i]);
    for (uint32_t i = 0; i < na; ++i) { p->aggr_inputs.emplace_back(inputs[i]); p->aggr_funcs.push_back(funcs[i]); }
    return p;
}
golap_plan* golap_plan_sort(golap_plan* input, golap_expr** exprs, uint32_t n) {
    auto* p = new golap_plan();
    p->kind = golap_plan::Sort; p->input.reset(input);
    for (uint32_t i = 0; i < n; ++i) p->exprs.emplace_back(exprs[i]);
    return p;
}
void golap_plan_free(golap_plan* p) { delete p; }

// golap_last_error(ctx) reads ctx->err first; host-layer errors are reported through this helper instead
const char* golap_host_last_error(void) { return g_err.c_str(); }

int32_t golap_execute(golap_ctx* ctx, golap_plan* plan, golap_result** out_result, uint32_t* out_n) {
    std::unique_ptr<golap_plan> owned(plan);
    // ctx may be NULL for plans that never touch the device (Batches / TableScan / Sort / Projection of numeric columns);
    // any device step then fails with a "needs the device" error: there is no CPU fallback.
    if (!plan || !out_result || !out_n) return herr(GOLAP_ERR_INVALID_ARG, "null argument");
    g_err.clear();
    Exec ex;
    ex.ctx = ctx;
    const double t0 = now_ms();
    auto res = std::make_unique<golap_result>();
    int32_t s = ex.run(*plan, res->batches);
    if (s == GOLAP_OK) s = ex.to_host(res->batches);
    if (s != GOLAP_OK) {
        const std::string why = !g_err.empty() ? g_err : std::string(golap_last_error(ctx));
        if (ctx) {   // nothing of this plan may still be running or in flight when its buffers go back to the pool
            (void)golap_ctx_synchronize(ctx, nullptr);
            (void)golap_xfer_synchronize(ctx);
        }
        ex.pending.clear();
        ex.keep.clear();
        g_err = !why.empty() ? why : std::string(golap_last_error(nullptr));
        return s;
    }
    ex.keep.clear();
    ex.prof.total_ms = now_ms() - t0;
    res->profile = ex.prof;
    *out_n = (uint32_t)res->batches.size();
    *out_result = res.release();
    return GOLAP_OK;
}

void golap_host_set_transfer_threads(uint32_t n) { g_xfer_threads.store(std::max<uint32_t>(1, std::min<uint32_t>(n, 64))); }
void golap_host_set_pipelining(uint32_t on) { g_pipeline.store(on ? 1u : 0u); }

int32_t golap_result_profile(const golap_result* r, golap_exec_profile* out) {
    if (!r || !out) return herr(GOLAP_ERR_INVALID_ARG, "null argument");
    *out = r->profile;
    return GOLAP_OK;
}

int32_t golap_result_export(golap_result* r, uint32_t i, ArrowArray* oa, ArrowSchema* os) {
    if (!r || !oa || !os || i >= r->batches.size()) return herr(GOLAP_ERR_INVALID_ARG, "bad result index");
    export_batch(std::move(r->batches[i]), oa, os);
    return GOLAP_OK;
}
void golap_result_free(golap_result* r) { delete r; }

}  // extern "C"
