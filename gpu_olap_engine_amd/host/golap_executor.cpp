// golap_executor.cpp -- GpuExecutor: the reference's operator API over Arrow RecordBatches,
// written in C++ on top of the C ABI (include/golap.h) only.  No HIP calls here: device
// memory comes from the SlabAllocator entry points, copies go through the TransferQueue.
//
// Mirrors gpu-olap-core/src/executor.rs:
//   execute            :15-24, dispatch :37-104
//   execute_table_scan :110-155   (empty batch, as the reference)
//   execute_filter     :161-178   -> golap_filter_compact + golap_take_8byte
//   execute_projection :184-214   (column references and literals)
//   execute_hash_join  :220-265   -> golap_join; unlike the reference (which returns an empty
//                                    batch at :255-264) the pairs are materialised
//   execute_aggregate  :297-355   -> golap_aggregate (ungrouped = reference; grouped = dense keys)
//   execute_sort       :361-370   (pass-through, as the reference)
//   eval_expr          :377-409   (column by name first, then by index)
#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/golap_host.h"

namespace {

// ---------------------------------------------------------------------------
// host RecordBatch
// ---------------------------------------------------------------------------
enum class DType { Int64, Float64, Int32, Utf8Raw, Other };

struct Column {
    std::string name;
    DType dtype = DType::Other;
    std::vector<uint8_t> data;   // packed little-endian values, nulls -> 0 (record_batch_convert.rs:102-110)
    std::vector<uint8_t> valid;  // Arrow validity bitmap of an exported column (empty = all valid)
    std::vector<uint8_t> in_valid;   // byte per row, 0 = NULL (ColumnBuffer.validity, column_buffer.rs:59-60); empty = no NULLs.  Travels with
                                     // the column through the operators; GpuFilter and GpuHashJoin hand it to the kernels (NULL keys
                                     // match nothing), GpuAggregate skips NULL values and puts NULL group keys into a group of their own
    uint64_t null_count = 0;
    uint64_t rows = 0;
    // Utf8 / LargeUtf8 as imported: value i = str_bytes[str_off[i] .. str_off[i + 1]).  golap_execute turns such a column into its
    // FNV-1a 64 hashes on the device before any operator runs (record_batch_convert.rs:93-97, :112-130: the reference's GPU columns
    // hold the hash, null -> 0): from then on it is an Int64 column.
    std::vector<int64_t> str_off;
    std::vector<uint8_t> str_bytes;
    size_t width() const { return dtype == DType::Int32 ? 4 : 8; }
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

thread_local std::string g_err;
int32_t herr(int32_t code, const std::string& msg) {
    g_err = msg;
    return code;
}

DType dtype_of_format(const char* f) {
    if (!f) return DType::Other;
    if (!strcmp(f, "l")) return DType::Int64;
    if (!strcmp(f, "g")) return DType::Float64;
    if (!strcmp(f, "i")) return DType::Int32;
    return DType::Other;   // ("u" / "U" strings become DType::Utf8Raw in import_batch)
}
const char* format_of_dtype(DType d) { return d == DType::Int64 ? "l" : d == DType::Float64 ? "g" : d == DType::Int32 ? "i" : "n"; }

bool import_batch(ArrowArray* arr, ArrowSchema* sch, RecordBatch& out, std::string& err) {
    if (!arr || !sch || !sch->format || strcmp(sch->format, "+s")) { err = "expected an Arrow struct array (format +s)"; return false; }
    if (arr->n_children != sch->n_children) { err = "array/schema children mismatch"; return false; }
    out.num_rows = (uint64_t)arr->length;
    for (int64_t c = 0; c < sch->n_children; ++c) {
        ArrowSchema* cs = sch->children[c];
        ArrowArray* ca = arr->children[c];
        Column col;
        col.name = cs->name ? cs->name : "";
        col.dtype = dtype_of_format(cs->format);
        col.rows = out.num_rows;
        if (cs->format && (!strcmp(cs->format, "u") || !strcmp(cs->format, "U"))) {
            const bool large = cs->format[0] == 'U';
            const int64_t off = ca->offset + arr->offset;
            const uint8_t* valid = ca->n_buffers >= 1 ? (const uint8_t*)ca->buffers[0] : nullptr;
            if (ca->n_buffers < 3 || (out.num_rows && (!ca->buffers[1] || !ca->buffers[2]))) {
                if (out.num_rows) { err = "string column '" + col.name + "' has no offsets / data buffer"; return false; }
            }
            col.dtype = DType::Utf8Raw;
            col.str_off.resize((size_t)out.num_rows + 1, 0);
            int64_t first = 0, last = 0;
            for (uint64_t i = 0; i <= out.num_rows && out.num_rows; ++i) {
                const int64_t o = large ? ((const int64_t*)ca->buffers[1])[off + (int64_t)i] : (int64_t)((const int32_t*)ca->buffers[1])[off + (int64_t)i];
                if (i == 0) first = o;
                col.str_off[i] = o - first;
                last = o;
            }
            if (out.num_rows) col.str_bytes.assign((const uint8_t*)ca->buffers[2] + first, (const uint8_t*)ca->buffers[2] + last);
            if (valid && ca->null_count != 0) {
                col.in_valid.assign((size_t)out.num_rows, 1);
                for (uint64_t i = 0; i < out.num_rows; ++i) {
                    const int64_t j = off + (int64_t)i;
                    if (!((valid[j >> 3] >> (j & 7)) & 1)) col.in_valid[i] = 0;
                }
            }
        } else if (col.dtype != DType::Other) {
            const size_t w = col.width();
            col.data.resize((size_t)out.num_rows * w);
            const uint8_t* src = ca->n_buffers >= 2 ? (const uint8_t*)ca->buffers[1] : nullptr;
            const uint8_t* valid = ca->n_buffers >= 1 ? (const uint8_t*)ca->buffers[0] : nullptr;
            const int64_t off = ca->offset + arr->offset;
            if (out.num_rows && !src) { err = "column '" + col.name + "' has no data buffer"; return false; }
            if (out.num_rows) memcpy(col.data.data(), src + (size_t)off * w, (size_t)out.num_rows * w);
            if (valid && ca->null_count != 0) {
                col.in_valid.assign((size_t)out.num_rows, 1);
                for (uint64_t i = 0; i < out.num_rows; ++i) {
                    const int64_t j = off + (int64_t)i;
                    if (!((valid[j >> 3] >> (j & 7)) & 1)) {
                        memset(col.data.data() + i * w, 0, w);   // null -> 0
                        col.in_valid[i] = 0;
                    }
                }
            }
        }
        out.cols.push_back(std::move(col));
    }
    return true;
}

// ---- export (the exported arrays own copies of the column data) ----
struct ExportPriv {
    std::vector<std::vector<uint8_t>> data, valid;
    std::vector<const void*> bufs;   // 2 per child + 1 for the struct
    std::vector<ArrowArray> child_arrays;
    std::vector<ArrowArray*> child_ptrs;
    std::vector<ArrowSchema> child_schemas;
    std::vector<ArrowSchema*> child_sptrs;
    std::vector<std::string> names;
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

void export_batch(RecordBatch&& b, ArrowArray* oa, ArrowSchema* os) {
    auto priv = std::make_shared<ExportPriv>();
    const size_t nc = b.cols.size();
    priv->data.resize(nc);
    priv->valid.resize(nc);
    priv->bufs.resize(2 * nc + 1, nullptr);
    priv->child_arrays.resize(nc);
    priv->child_schemas.resize(nc);
    priv->names.resize(nc);
    for (size_t c = 0; c < nc; ++c) {
        Column& col = b.cols[c];
        if (col.valid.empty() && !col.in_valid.empty()) {   // byte-per-row validity -> Arrow bitmap
            uint64_t nulls = 0;
            col.valid.assign((size_t)(b.num_rows + 7) / 8, 0);
            for (uint64_t i = 0; i < b.num_rows; ++i) {
                if (col.in_valid[i]) col.valid[i >> 3] |= (uint8_t)(1u << (i & 7));
                else ++nulls;
            }
            col.null_count = nulls;
            if (!nulls) col.valid.clear();
        }
        priv->data[c] = std::move(b.cols[c].data);
        priv->valid[c] = std::move(b.cols[c].valid);
        priv->names[c] = b.cols[c].name;
        priv->bufs[2 * c] = priv->valid[c].empty() ? nullptr : (const void*)priv->valid[c].data();
        priv->bufs[2 * c + 1] = priv->data[c].empty() ? (const void*)"" : (const void*)priv->data[c].data();
        ArrowArray& ca = priv->child_arrays[c];
        memset(&ca, 0, sizeof ca);
        ca.length = (int64_t)b.num_rows;
        ca.null_count = (int64_t)b.cols[c].null_count;
        ca.n_buffers = 2;
        ca.buffers = &priv->bufs[2 * c];
        ca.release = release_child_array;
        ArrowSchema& cs = priv->child_schemas[c];
        memset(&cs, 0, sizeof cs);
        cs.format = format_of_dtype(b.cols[c].dtype);
        cs.name = priv->names[c].c_str();
        cs.flags = b.cols[c].null_count ? 2 /* ARROW_FLAG_NULLABLE */ : 0;
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
};

namespace {

// ---------------------------------------------------------------------------
// device helpers (slabs + transfer queue)
// ---------------------------------------------------------------------------
struct DevBuf {
    golap_ctx* ctx = nullptr;
    uint64_t ptr = 0;
    uint32_t cls = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : ctx(o.ctx), ptr(o.ptr), cls(o.cls) { o.ptr = 0; }
    ~DevBuf() { if (ptr) golap_slab_free(ctx, ptr, cls); }
    int32_t alloc(golap_ctx* c, uint64_t bytes) {
        ctx = c;
        return golap_slab_alloc(c, std::max<uint64_t>(bytes, 8), &ptr, &cls);
    }
};

#define H_TRY(expr)                      \
    do {                                 \
        int32_t _s = (expr);             \
        if (_s != GOLAP_OK) return _s;   \
    } while (0)

struct Exec {
    golap_ctx* ctx;
    uint32_t next_stream = 0;

    int32_t h2d(const void* src, uint64_t dst, uint64_t bytes) {
        // split large columns over the transfer streams (transfer_queue.rs:72 round-robin)
        const uint64_t piece = 64ull << 20;
        for (uint64_t o = 0; o < bytes; o += piece)
            H_TRY(golap_xfer_enqueue(ctx, GOLAP_HOST_TO_DEVICE, (uint64_t)(uintptr_t)src + o, dst + o, std::min(piece, bytes - o), next_stream++));
        return GOLAP_OK;
    }
    int32_t d2h(uint64_t src, void* dst, uint64_t bytes) {
        const uint64_t piece = 64ull << 20;
        for (uint64_t o = 0; o < bytes; o += piece)
            H_TRY(golap_xfer_enqueue(ctx, GOLAP_DEVICE_TO_HOST, src + o, (uint64_t)(uintptr_t)dst + o, std::min(piece, bytes - o), next_stream++));
        return GOLAP_OK;
    }
    int32_t xsync() { return golap_xfer_synchronize(ctx); }
    int32_t ksync() { return golap_ctx_synchronize(ctx, nullptr); }

    // eval_expr, executor.rs:377-409: column by NAME first, then by index
    static int resolve_column(const golap_expr& e, const RecordBatch& b) {
        int i = b.index_of(e.name);
        if (i >= 0) return i;
        if (e.index < b.cols.size()) return (int)e.index;
        return -1;
    }

    int32_t run(golap_plan& p, std::vector<RecordBatch>& out);
    int32_t filter(std::vector<RecordBatch>& in, const golap_expr& pred, std::vector<RecordBatch>& out);
    int32_t projection(std::vector<RecordBatch>& in, const std::vector<std::unique_ptr<golap_expr>>& exprs, std::vector<RecordBatch>& out);
    int32_t hash_join(std::vector<RecordBatch>& left, std::vector<RecordBatch>& right, golap_plan& p, std::vector<RecordBatch>& out);
    int32_t aggregate(std::vector<RecordBatch>& in, golap_plan& p, std::vector<RecordBatch>& out);
    int32_t aggregate_multi_key(std::vector<RecordBatch>& in, golap_plan& p, std::vector<RecordBatch>& out);
    int32_t hash_strings(golap_plan& p);
};

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
    for (RecordBatch& b : in) {
        // non-binary predicate => every row passes (:474)
        if (pred.kind != golap_expr::Binary) { out.push_back(std::move(b)); continue; }
        std::vector<const golap_expr*> leaves;
        and_leaves(pred, leaves);
        if (leaves.size() > GOLAP_FILTER_AND_MAX) return herr(GOLAP_ERR_UNSUPPORTED, "GpuFilter: at most " + std::to_string(GOLAP_FILTER_AND_MAX) + " comparisons in one conjunction");
        std::vector<golap_filter_predicate> fps;
        std::vector<int> fcols;
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
            // operands that are not both Int64 => every row passes (:449-452, :471-472)
            if (b.cols[ci].dtype != DType::Int64 || !parse_i64(litx->value, lit)) continue;
            fps.push_back(golap_filter_predicate{(uint64_t)ci, (int32_t)op, lit});
            fcols.push_back(ci);
        }
        if (fps.empty()) { out.push_back(std::move(b)); continue; }
        const uint64_t n = b.num_rows;
        if (n == 0) continue;   // empty batches are dropped (:173)
        DevBuf didx, dcnt;
        std::vector<DevBuf> dkeys(fps.size()), dvals(fps.size());
        std::vector<uint64_t> kptrs(fps.size(), 0), vptrs(fps.size(), 0);
        H_TRY(didx.alloc(ctx, n * 4));
        H_TRY(dcnt.alloc(ctx, 8));
        for (size_t k = 0; k < fps.size(); ++k) {
            const Column& c = b.cols[fcols[k]];
            size_t same = k;
            for (size_t j = 0; j < k; ++j) if (fcols[j] == fcols[k]) { same = j; break; }
            if (same != k) { kptrs[k] = kptrs[same]; vptrs[k] = vptrs[same]; continue; }   // (a range: lo <= x AND x < hi reads the column once)
            H_TRY(dkeys[k].alloc(ctx, n * 8));
            H_TRY(h2d(c.data.data(), dkeys[k].ptr, n * 8));
            kptrs[k] = dkeys[k].ptr;
            if (!c.in_valid.empty()) {   // a comparison with NULL is not true: such rows do not pass
                H_TRY(dvals[k].alloc(ctx, n));
                H_TRY(h2d(c.in_valid.data(), dvals[k].ptr, n));
                vptrs[k] = dvals[k].ptr;
            }
        }
        H_TRY(xsync());
        if (fps.size() == 1) H_TRY(golap_filter_compact_nullable(ctx, &fps[0], kptrs[0], vptrs[0], n, didx.ptr, dcnt.ptr, nullptr));
        else H_TRY(golap_filter_compact_and(ctx, fps.data(), kptrs.data(), vptrs.data(), (uint32_t)fps.size(), n, didx.ptr, dcnt.ptr, nullptr));
        H_TRY(ksync());
        uint64_t kept = 0;
        H_TRY(d2h(dcnt.ptr, &kept, 8));
        H_TRY(xsync());
        if (kept == 0) continue;
        std::vector<uint32_t> hidx;   // the kept row ids on the host too, when some column carries NULLs (their validity is gathered here)
        bool any_nulls = false;
        for (const Column& c : b.cols) any_nulls = any_nulls || !c.in_valid.empty();
        if (any_nulls) {
            hidx.resize(kept);
            H_TRY(d2h(didx.ptr, hidx.data(), kept * 4));
            H_TRY(xsync());
        }
        RecordBatch ob;
        ob.num_rows = kept;
        for (Column& c : b.cols) {
            if (c.dtype != DType::Int64 && c.dtype != DType::Float64)
                return herr(GOLAP_ERR_UNSUPPORTED, "GpuFilter: column '" + c.name + "' is neither Int64 nor Float64 (take_int64, executor.rs:496-507)");
            DevBuf dcol, dout;
            H_TRY(dcol.alloc(ctx, n * 8));
            H_TRY(dout.alloc(ctx, kept * 8));
            H_TRY(h2d(c.data.data(), dcol.ptr, n * 8));
            H_TRY(xsync());
            H_TRY(golap_take_8byte(ctx, dcol.ptr, didx.ptr, 0, kept, dout.ptr, nullptr));
            H_TRY(ksync());
            Column oc;
            oc.name = c.name; oc.dtype = c.dtype; oc.rows = kept;
            oc.data.resize(kept * 8);
            H_TRY(d2h(dout.ptr, oc.data.data(), kept * 8));
            H_TRY(xsync());
            if (!c.in_valid.empty()) {
                oc.in_valid.resize(kept);
                for (uint64_t i = 0; i < kept; ++i) oc.in_valid[i] = c.in_valid[hidx[i]];
            }
            ob.cols.push_back(std::move(oc));
        }
        out.push_back(std::move(ob));
    }
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
                Column c = b.cols[ci];
                c.name = e->name;                      // expr_name, :531
                ob.cols.push_back(std::move(c));
            } else if (e->kind == golap_expr::Literal) {
                int64_t v;
                if (!parse_i64(e->value, v)) return herr(GOLAP_ERR_UNSUPPORTED, "Cannot parse literal '" + e->value + "'");   // :400
                Column c;
                c.name = "lit(" + e->value + ")";      // :532
                c.dtype = DType::Int64; c.rows = b.num_rows;
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

// concatenates column `ci(b)` of every batch (+ its validity bytes: `valid` stays empty when no batch has NULLs in it);
// returns false if any batch's column is not Int64
static bool concat_i64(const std::vector<RecordBatch>& bs, const golap_expr& key, std::vector<int64_t>& out, std::vector<uint8_t>& valid, std::string& err) {
    bool any_nulls = false;
    for (const RecordBatch& b : bs) {
        const int ci = Exec::resolve_column(key, b);
        if (ci < 0) { err = "Column '" + key.name + "' not found in batch"; return false; }
        if (b.cols[ci].dtype != DType::Int64) { err = "join key '" + key.name + "' must be Int64 (executor.rs:246)"; return false; }
        any_nulls = any_nulls || !b.cols[ci].in_valid.empty();
    }
    for (const RecordBatch& b : bs) {
        const int ci = Exec::resolve_column(key, b);
        const size_t o = out.size();
        out.resize(o + b.num_rows);
        if (b.num_rows) memcpy(out.data() + o, b.cols[ci].data.data(), b.num_rows * 8);
        if (any_nulls) {
            valid.resize(o + b.num_rows, 1);
            if (!b.cols[ci].in_valid.empty() && b.num_rows) memcpy(valid.data() + o, b.cols[ci].in_valid.data(), b.num_rows);
        }
    }
    return true;
}

// ---------------------------------------------------------------------------
// execute_hash_join, executor.rs:220-265
// ---------------------------------------------------------------------------
int32_t Exec::hash_join(std::vector<RecordBatch>& left, std::vector<RecordBatch>& right, golap_plan& p, std::vector<RecordBatch>& out) {
    if (left.empty() || right.empty()) return GOLAP_OK;                 // :231-233 -> vec![]
    if (p.left_keys.empty() || p.right_keys.empty()) return herr(GOLAP_ERR_INVALID_ARG, "GpuHashJoin needs one key per side");
    // The reference executor reads the FIRST key expression only (:245) although the plan carries vectors (physical_plan.rs:35-36);
    // here every key column takes part (up to 4), and NULL keys match nothing (SQL; the reference has turned them into 0 by now).
    const size_t nk = p.left_keys.size();
    if (p.right_keys.size() != nk) return herr(GOLAP_ERR_INVALID_ARG, "GpuHashJoin: left_keys and right_keys differ in length");
    if (nk > 4) return herr(GOLAP_ERR_UNSUPPORTED, "GpuHashJoin: at most 4 key columns");
    std::vector<std::vector<int64_t>> bkc(nk), pkc(nk);
    std::vector<std::vector<uint8_t>> bvc(nk), pvc(nk);
    std::string err;
    for (size_t k = 0; k < nk; ++k) {
        const golap_expr &lk = *p.left_keys[k], &rk = *p.right_keys[k];
        if (lk.kind != golap_expr::Column || rk.kind != golap_expr::Column) return herr(GOLAP_ERR_UNSUPPORTED, "GpuHashJoin: key expressions must be column references");
        if (!concat_i64(left, lk, bkc[k], bvc[k], err) || !concat_i64(right, rk, pkc[k], pvc[k], err)) return herr(GOLAP_ERR_INVALID_ARG, err);
    }
    const uint64_t nb = bkc[0].size(), np = pkc[0].size();   // global row index = running offset over batches (:249, :252)
    RecordBatch ob;
    // output schema: left fields then right fields
    for (const Column& c : left[0].cols) { Column oc; oc.name = c.name; oc.dtype = c.dtype; ob.cols.push_back(oc); }
    const size_t n_left_cols = ob.cols.size();
    for (const Column& c : right[0].cols) { Column oc; oc.name = c.name; oc.dtype = c.dtype; ob.cols.push_back(oc); }
    for (const Column& c : ob.cols)
        if (c.dtype != DType::Int64 && c.dtype != DType::Float64) return herr(GOLAP_ERR_UNSUPPORTED, "GpuHashJoin: payload column '" + c.name + "' is neither Int64 nor Float64");
    uint64_t m = 0;
    DevBuf dpairs;
    const bool outer_keeps_something = (nb && (p.join_type == GOLAP_JOIN_LEFT || p.join_type == GOLAP_JOIN_FULL)) ||
                                       (np && (p.join_type == GOLAP_JOIN_RIGHT || p.join_type == GOLAP_JOIN_FULL));
    if ((nb && np) || outer_keeps_something) {
        DevBuf dcnt;
        std::vector<DevBuf> db(nk), dp(nk), dbv(nk), dpv(nk);
        H_TRY(dcnt.alloc(ctx, 8));
        for (size_t k = 0; k < nk; ++k) {
            H_TRY(db[k].alloc(ctx, nb * 8));
            H_TRY(dp[k].alloc(ctx, np * 8));
            H_TRY(h2d(bkc[k].data(), db[k].ptr, nb * 8));
            H_TRY(h2d(pkc[k].data(), dp[k].ptr, np * 8));
            if (!bvc[k].empty()) { H_TRY(dbv[k].alloc(ctx, nb)); H_TRY(h2d(bvc[k].data(), dbv[k].ptr, nb)); }
            if (!pvc[k].empty()) { H_TRY(dpv[k].alloc(ctx, np)); H_TRY(h2d(pvc[k].data(), dpv[k].ptr, np)); }
        }
        H_TRY(xsync());
        uint64_t cap = std::max<uint64_t>(np + ((p.join_type & 1) ? nb : 0), 1);   // first guess; the count is the TRUE total (join_kernels.cuh:155-160)
        for (int attempt = 0; attempt < 2; ++attempt) {
            H_TRY(dpairs.alloc(ctx, cap * 8));
            golap_join_params jp;
            memset(&jp, 0, sizeof jp);
            jp.build_key_ptr = db[0].ptr; jp.probe_key_ptr = dp[0].ptr; jp.n_build = nb; jp.n_probe = np;
            jp.out_pairs_ptr = dpairs.ptr; jp.out_count_ptr = dcnt.ptr; jp.max_pairs = cap;
            jp.join_type = (int32_t)p.join_type;
            jp.n_key_cols = (uint32_t)nk;
            for (size_t k = 0; k < nk; ++k) {
                if (k) { jp.build_key_extra[k - 1] = db[k].ptr; jp.probe_key_extra[k - 1] = dp[k].ptr; }
                jp.build_validity[k] = bvc[k].empty() ? 0 : dbv[k].ptr;
                jp.probe_validity[k] = pvc[k].empty() ? 0 : dpv[k].ptr;
            }
            H_TRY(golap_join(ctx, golap_choose_strategy(nb), &jp, nullptr));   // JoinKernel::choose_strategy, join_kernel.rs:71-77
            H_TRY(ksync());
            H_TRY(d2h(dcnt.ptr, &m, 8));
            H_TRY(xsync());
            if (m <= cap) break;
            DevBuf old = std::move(dpairs);   // too small: release, then retry with the exact size
            cap = m;
        }
    }
    ob.num_rows = m;
    if (m) {
        // split the interleaved pairs into two index vectors on the host, then gather every payload column on the device
        std::vector<uint32_t> pairs(2 * m), lidx(m), ridx(m);
        H_TRY(d2h(dpairs.ptr, pairs.data(), m * 8));
        H_TRY(xsync());
        // canonical order (probe_rid, build_rid): the device order is unspecified (join_kernels.cuh:155)
        std::vector<uint64_t> keyed(m);
        for (uint64_t i = 0; i < m; ++i) keyed[i] = ((uint64_t)pairs[2 * i + 1] << 32) | pairs[2 * i];
        std::sort(keyed.begin(), keyed.end());
        for (uint64_t i = 0; i < m; ++i) { lidx[i] = (uint32_t)keyed[i]; ridx[i] = (uint32_t)(keyed[i] >> 32); }
        DevBuf dl, dr;
        H_TRY(dl.alloc(ctx, m * 4));
        H_TRY(dr.alloc(ctx, m * 4));
        H_TRY(h2d(lidx.data(), dl.ptr, m * 4));
        H_TRY(h2d(ridx.data(), dr.ptr, m * 4));
        H_TRY(xsync());
        for (size_t c = 0; c < ob.cols.size(); ++c) {
            const bool is_left = c < n_left_cols;
            const std::vector<RecordBatch>& side = is_left ? left : right;
            const size_t sc = is_left ? c : c - n_left_cols;
            const uint64_t rows = is_left ? nb : np;
            std::vector<uint8_t> whole(rows * 8), src_valid;
            size_t o = 0;
            bool src_nulls = false;
            for (const RecordBatch& b : side) src_nulls = src_nulls || (sc < b.cols.size() && !b.cols[sc].in_valid.empty());
            if (src_nulls) src_valid.assign(rows, 1);
            for (const RecordBatch& b : side) {
                if (sc >= b.cols.size() || b.cols[sc].dtype != ob.cols[c].dtype) return herr(GOLAP_ERR_INVALID_ARG, "batches of one side differ in schema");
                if (b.num_rows) memcpy(whole.data() + o, b.cols[sc].data.data(), b.num_rows * 8);
                if (src_nulls && !b.cols[sc].in_valid.empty() && b.num_rows) memcpy(src_valid.data() + o / 8, b.cols[sc].in_valid.data(), b.num_rows);
                o += b.num_rows * 8;
            }
            DevBuf dcol, dout;
            H_TRY(dcol.alloc(ctx, rows * 8));
            H_TRY(dout.alloc(ctx, m * 8));
            H_TRY(h2d(whole.data(), dcol.ptr, rows * 8));
            H_TRY(xsync());
            H_TRY(golap_take_8byte(ctx, dcol.ptr, is_left ? dl.ptr : dr.ptr, 0, m, dout.ptr, nullptr));
            H_TRY(ksync());
            ob.cols[c].rows = m;
            {   // rows whose source side is missing (NULL row id of an outer join) become Arrow nulls
                const std::vector<uint32_t>& ix = is_left ? lidx : ridx;
                uint64_t nulls = 0;
                auto is_null = [&](uint64_t i) { return ix[i] == 0xFFFFFFFFu || (src_nulls && !src_valid[ix[i]]); };   // missing side, or a NULL in the source row
                for (uint64_t i = 0; i < m; ++i) nulls += is_null(i);
                if (nulls) {
                    ob.cols[c].null_count = nulls;
                    ob.cols[c].valid.assign((m + 7) / 8, 0);
                    for (uint64_t i = 0; i < m; ++i)
                        if (!is_null(i)) ob.cols[c].valid[i >> 3] |= (uint8_t)(1u << (i & 7));
                }
            }
            ob.cols[c].data.resize(m * 8);
            H_TRY(d2h(dout.ptr, ob.cols[c].data.data(), m * 8));
            H_TRY(xsync());
        }
    }
    out.push_back(std::move(ob));   // one batch, like the reference's single output batch (:262-264)
    return GOLAP_OK;
}

static const char* agg_name(int32_t f) {
    switch (f) { case GOLAP_AGG_SUM: return "sum"; case GOLAP_AGG_COUNT: return "count"; case GOLAP_AGG_MIN: return "min"; case GOLAP_AGG_MAX: return "max"; default: return "avg"; }
}

// ---------------------------------------------------------------------------
// execute_aggregate, executor.rs:297-355 (+ eval_agg_input :509-527)
// ---------------------------------------------------------------------------
// rows `idx` of a batch (host side; values, widths and validity bytes travel together)
static RecordBatch take_rows(const RecordBatch& b, const std::vector<uint64_t>& idx) {
    RecordBatch o;
    o.num_rows = idx.size();
    for (const Column& c : b.cols) {
        Column oc;
        oc.name = c.name; oc.dtype = c.dtype; oc.rows = idx.size();
        const size_t w = c.width();
        oc.data.resize(idx.size() * w);
        for (size_t i = 0; i < idx.size(); ++i) memcpy(oc.data.data() + i * w, c.data.data() + idx[i] * w, w);
        if (!c.in_valid.empty()) {
            oc.in_valid.resize(idx.size());
            for (size_t i = 0; i < idx.size(); ++i) oc.in_valid[i] = c.in_valid[idx[i]];
        }
        o.cols.push_back(std::move(oc));
    }
    return o;
}

// GROUP BY over several columns (physical_plan.rs:52-54: group_by is a Vec<PhysicalExpr>; lowering :157-172 keeps every column; the
// reference's executor ignores it, executor.rs:307).  The columns are Int64 / Int32, so a row's composite key is a mixed-radix number:
// digit_c = value - min_c (NULL: one past the column's largest digit -- NULLs sort last, and (NULL, 1) is a group of its own), weight_c
// = the product of the later columns' digit counts.  While that number fits 62 bits -- dictionary-like ids, dates, small codes: the
// usual GROUP BY columns -- it IS a single Int64 key: exact, no collisions, and the one-column path below (dense or hashed) does the
// rest; the result's packed key column is unpacked into the key columns again.  Wider composites are refused with their sizes.
int32_t Exec::aggregate_multi_key(std::vector<RecordBatch>& in, golap_plan& p, std::vector<RecordBatch>& out) {
    const size_t K = p.group_by.size();
    const uint32_t na = (uint32_t)p.aggr_inputs.size();
    uint64_t n = 0;
    for (const RecordBatch& b : in) n += b.num_rows;
    std::vector<std::vector<int64_t>> vals(K);
    std::vector<std::vector<uint8_t>> valid(K);
    std::vector<int64_t> mn(K, INT64_MAX), mx(K, INT64_MIN);
    std::vector<bool> has_null(K, false);
    for (size_t c = 0; c < K; ++c) {
        if (p.group_by[c]->kind != golap_expr::Column) return herr(GOLAP_ERR_UNSUPPORTED, "GpuAggregate: group-by expressions must be column references");
        vals[c].reserve(n);
        valid[c].reserve(n);
        for (const RecordBatch& b : in) {
            const int ci = resolve_column(*p.group_by[c], b);
            if (ci < 0) return herr(GOLAP_ERR_INVALID_ARG, "Column '" + p.group_by[c]->name + "' not found in batch");
            const Column& col = b.cols[ci];
            if (col.dtype != DType::Int64 && col.dtype != DType::Int32) return herr(GOLAP_ERR_UNSUPPORTED, "GpuAggregate: group key must be Int64 or Int32");
            for (uint64_t i = 0; i < b.num_rows; ++i) {
                const int64_t v = col.dtype == DType::Int64 ? ((const int64_t*)col.data.data())[i] : (int64_t)((const int32_t*)col.data.data())[i];
                const uint8_t ok = col.in_valid.empty() ? 1 : col.in_valid[i];
                vals[c].push_back(v);
                valid[c].push_back(ok);
                if (ok) { mn[c] = std::min(mn[c], v); mx[c] = std::max(mx[c], v); } else has_null[c] = true;
            }
        }
        if (mn[c] > mx[c]) mn[c] = mx[c] = 0;   // no valid value at all
    }
    // digit counts and weights (first column most significant: rows come out sorted by (k0, k1, ...), NULLS LAST per column)
    std::vector<unsigned __int128> digits(K);
    std::vector<uint64_t> weight(K, 1);
    unsigned __int128 total = 1;
    for (size_t c = K; c-- > 0;) {
        digits[c] = (unsigned __int128)((__int128)mx[c] - (__int128)mn[c]) + 1 + (has_null[c] ? 1 : 0);
        weight[c] = (uint64_t)total;
        total *= digits[c];
        if (total > ((unsigned __int128)1 << 62)) {
            std::string sizes;
            for (size_t k = 0; k < K; ++k) sizes += (k ? " x " : "") + std::to_string((double)((__int128)mx[k] - (__int128)mn[k]) + 1.0);
            return herr(GOLAP_ERR_UNSUPPORTED, "GpuAggregate: the composite group key does not fit 62 bits (value ranges " + sizes + ")");
        }
    }
    // one packed Int64 key column per batch, under a name no input column has
    const std::string pk = "__golap_group_key";
    uint64_t o = 0;
    for (RecordBatch& b : in) {
        Column kc;
        kc.name = pk; kc.dtype = DType::Int64; kc.rows = b.num_rows;
        kc.data.resize(b.num_rows * 8);
        int64_t* d = (int64_t*)kc.data.data();
        for (uint64_t i = 0; i < b.num_rows; ++i) {
            uint64_t key = 0;
            for (size_t c = 0; c < K; ++c) {
                const uint64_t dg = valid[c][o + i] ? (uint64_t)(vals[c][o + i] - mn[c]) : (uint64_t)(digits[c] - 1);
                key += dg * weight[c];
            }
            d[i] = (int64_t)key;
        }
        o += b.num_rows;
        b.cols.push_back(std::move(kc));
    }
    auto gb = std::move(p.group_by);
    p.group_by.clear();
    p.group_by.emplace_back(new golap_expr());
    p.group_by[0]->kind = golap_expr::Column;
    p.group_by[0]->name = pk;
    std::vector<RecordBatch> g;
    const int32_t rc = aggregate(in, p, g);
    p.group_by = std::move(gb);
    for (RecordBatch& b : in) b.cols.pop_back();
    H_TRY(rc);
    if (g.empty() || g[0].cols.size() != 1 + (size_t)na) return herr(GOLAP_ERR_INVALID_ARG, "GpuAggregate: unexpected grouped result");
    RecordBatch ob;
    ob.num_rows = g[0].num_rows;
    const int64_t* packed = (const int64_t*)g[0].cols[0].data.data();
    for (size_t c = 0; c < K; ++c) {
        Column kc;
        kc.name = p.group_by[c]->name; kc.dtype = DType::Int64; kc.rows = ob.num_rows;
        kc.data.resize(ob.num_rows * 8);
        if (has_null[c]) kc.in_valid.assign(ob.num_rows, 1);
        int64_t* d = (int64_t*)kc.data.data();
        for (uint64_t i = 0; i < ob.num_rows; ++i) {
            const uint64_t dg = (uint64_t)(((unsigned __int128)(uint64_t)packed[i] / weight[c]) % digits[c]);
            if (has_null[c] && dg == (uint64_t)(digits[c] - 1)) { d[i] = 0; kc.in_valid[i] = 0; }
            else d[i] = mn[c] + (int64_t)dg;
        }
        ob.cols.push_back(std::move(kc));
    }
    for (uint32_t i = 0; i < na; ++i) ob.cols.push_back(std::move(g[0].cols[1 + i]));
    out.push_back(std::move(ob));
    return GOLAP_OK;
}

int32_t Exec::aggregate(std::vector<RecordBatch>& in, golap_plan& p, std::vector<RecordBatch>& out) {
    const uint32_t na = (uint32_t)p.aggr_inputs.size();
    if (p.group_by.size() > 1) return aggregate_multi_key(in, p, out);
    // NULL group keys: SQL puts them into ONE group of their own.  Only when the key column really carries NULLs: the rows are
    // split on the host, the rows with a key take the ordinary path below, the others an ungrouped aggregate, and the result
    // gets one more row whose key is NULL (last, like NULLS LAST).
    if (p.group_by.size() == 1 && p.group_by[0]->kind == golap_expr::Column) {
        bool key_nulls = false;
        for (const RecordBatch& b : in) {
            const int ci = resolve_column(*p.group_by[0], b);
            if (ci >= 0 && !b.cols[ci].in_valid.empty())
                for (uint8_t v : b.cols[ci].in_valid) if (!v) { key_nulls = true; break; }
            if (key_nulls) break;
        }
        if (key_nulls) {
            std::vector<RecordBatch> with_key, without_key;
            for (const RecordBatch& b : in) {
                const int ci = resolve_column(*p.group_by[0], b);
                if (ci < 0) return herr(GOLAP_ERR_INVALID_ARG, "Column '" + p.group_by[0]->name + "' not found in batch");
                std::vector<uint64_t> yes, no;
                const std::vector<uint8_t>& kv = b.cols[ci].in_valid;
                for (uint64_t i = 0; i < b.num_rows; ++i) (kv.empty() || kv[i] ? yes : no).push_back(i);
                RecordBatch y = take_rows(b, yes), n = take_rows(b, no);
                y.cols[ci].in_valid.clear();              // no NULL keys left on this side
                with_key.push_back(std::move(y));
                without_key.push_back(std::move(n));
            }
            std::vector<RecordBatch> g, u;
            H_TRY(aggregate(with_key, p, g));
            auto gb = std::move(p.group_by);
            p.group_by.clear();
            const int32_t rc = aggregate(without_key, p, u);   // one row: the aggregates over the rows whose key is NULL
            p.group_by = std::move(gb);
            H_TRY(rc);
            RecordBatch ob = g.empty() ? RecordBatch() : std::move(g[0]);
            if (ob.cols.empty()) return herr(GOLAP_ERR_INVALID_ARG, "GpuAggregate: empty grouped result");
            Column& kc = ob.cols[0];
            kc.in_valid.assign(ob.num_rows, 1);
            kc.in_valid.push_back(0);
            kc.data.resize(kc.data.size() + 8, 0);
            for (uint32_t i = 0; i < na; ++i) {
                Column& ac = ob.cols[1 + i];
                const Column& uc = u[0].cols[i];
                ac.data.insert(ac.data.end(), uc.data.begin(), uc.data.begin() + 8);
                if (!ac.in_valid.empty()) ac.in_valid.push_back(1);
                ac.rows = ob.num_rows + 1;
            }
            ob.num_rows += 1;
            kc.rows = ob.num_rows;
            out.push_back(std::move(ob));
            return GOLAP_OK;
        }
    }
    // gather the distinct value columns the aggregates read
    std::vector<std::string> val_names;
    std::vector<uint64_t> spec_col(na);
    for (uint32_t i = 0; i < na; ++i) {
        const golap_expr& e = *p.aggr_inputs[i];
        if (e.kind != golap_expr::Column) return herr(GOLAP_ERR_UNSUPPORTED, "GpuAggregate: aggregate inputs must be column references");
        auto it = std::find(val_names.begin(), val_names.end(), e.name);
        if (it == val_names.end()) { val_names.push_back(e.name); spec_col[i] = val_names.size() - 1; }
        else spec_col[i] = (uint64_t)(it - val_names.begin());
    }
    uint64_t n = 0;
    for (const RecordBatch& b : in) n += b.num_rows;
    const uint32_t nv = (uint32_t)val_names.size();
    std::vector<DevBuf> dvals(nv), dvalid(nv);
    std::vector<uint64_t> vptrs(nv), vvptrs(nv, 0);   // vvptrs: validity of the value columns (0 = that column has no NULLs)
    std::vector<int32_t> vdt(nv, GOLAP_DT_INT64);
    bool any_value_nulls = false;
    for (uint32_t v = 0; v < nv; ++v) {
        H_TRY(dvals[v].alloc(ctx, n * 8));
        vptrs[v] = dvals[v].ptr;
        uint64_t o = 0;
        std::vector<uint8_t> valid;   // byte per row over all batches, only when some batch carries NULLs in this column
        {
            bool nulls = false;
            for (const RecordBatch& b : in) {
                golap_expr e; e.kind = golap_expr::Column; e.name = val_names[v];
                for (uint32_t i = 0; i < na; ++i) if (p.aggr_inputs[i]->name == val_names[v]) { e.index = p.aggr_inputs[i]->index; break; }
                const int ci = resolve_column(e, b);
                if (ci >= 0 && !b.cols[ci].in_valid.empty()) nulls = true;
            }
            if (nulls && n) {
                valid.assign(n, 1);
                uint64_t ro = 0;
                for (const RecordBatch& b : in) {
                    golap_expr e; e.kind = golap_expr::Column; e.name = val_names[v];
                    for (uint32_t i = 0; i < na; ++i) if (p.aggr_inputs[i]->name == val_names[v]) { e.index = p.aggr_inputs[i]->index; break; }
                    const int ci = resolve_column(e, b);
                    if (ci >= 0 && !b.cols[ci].in_valid.empty() && b.num_rows) memcpy(valid.data() + ro, b.cols[ci].in_valid.data(), b.num_rows);
                    ro += b.num_rows;
                }
                H_TRY(dvalid[v].alloc(ctx, n));
                H_TRY(h2d(valid.data(), dvalid[v].ptr, n));
                H_TRY(xsync());   // (the host vector goes out of scope below)
                vvptrs[v] = dvalid[v].ptr;
                any_value_nulls = true;
            }
        }
        for (const RecordBatch& b : in) {
            golap_expr e; e.kind = golap_expr::Column; e.name = val_names[v];
            for (uint32_t i = 0; i < na; ++i) if (p.aggr_inputs[i]->name == val_names[v]) { e.index = p.aggr_inputs[i]->index; break; }
            const int ci = resolve_column(e, b);
            if (ci < 0) return herr(GOLAP_ERR_INVALID_ARG, "Column '" + val_names[v] + "' not found in batch");
            const Column& c = b.cols[ci];
            if (c.dtype != DType::Int64 && c.dtype != DType::Float64)
                return herr(GOLAP_ERR_UNSUPPORTED, "GpuAggregate: input '" + c.name + "' must be Int64 or Float64 (executor.rs:520-526)");
            vdt[v] = c.dtype == DType::Float64 ? GOLAP_DT_FLOAT64 : GOLAP_DT_INT64;
            if (b.num_rows) H_TRY(h2d(c.data.data(), dvals[v].ptr + o, b.num_rows * 8));
            o += b.num_rows * 8;
        }
    }
    std::vector<golap_aggregate_spec> specs(na);
    for (uint32_t i = 0; i < na; ++i) specs[i] = golap_aggregate_spec{p.aggr_funcs[i], spec_col[i], i};
    RecordBatch ob;
    if (p.group_by.empty()) {
        // the reference's single-group result: one Float64 scalar per aggregate (:330-349)
        DevBuf dout;
        H_TRY(dout.alloc(ctx, (uint64_t)std::max<uint32_t>(na, 1) * 8));
        H_TRY(xsync());
        golap_aggregate_params ap;
        memset(&ap, 0, sizeof ap);
        ap.value_ptrs = vptrs.data(); ap.n_value_cols = nv; ap.n_rows = n; ap.n_groups = 1; ap.out_ptr = dout.ptr;
        ap.value_validity_ptrs = any_value_nulls ? vvptrs.data() : nullptr;   // SQL aggregates skip NULL values
        ap.key_dtype = GOLAP_DT_INT64; ap.value_dtypes = vdt.data();
        H_TRY(golap_aggregate(ctx, specs.data(), na, &ap, nullptr));
        H_TRY(ksync());
        std::vector<double> res(std::max<uint32_t>(na, 1));
        H_TRY(d2h(dout.ptr, res.data(), (uint64_t)na * 8));
        H_TRY(xsync());
        ob.num_rows = 1;
        for (uint32_t i = 0; i < na; ++i) {
            Column c;
            c.name = std::string(agg_name(p.aggr_funcs[i])) + "(" + p.aggr_inputs[i]->name + ")";
            c.dtype = DType::Float64; c.rows = 1;
            c.data.resize(8);
            memcpy(c.data.data(), &res[i], 8);
            ob.cols.push_back(std::move(c));
        }
        out.push_back(std::move(ob));
        return GOLAP_OK;
    }
    // GROUP BY one Int64/Int32 key with dense non-negative values: slot = key % n_groups (join_kernels.cuh:230)
    if (p.group_by.size() != 1 || p.group_by[0]->kind != golap_expr::Column) return herr(GOLAP_ERR_UNSUPPORTED, "GpuAggregate: one group-by column is supported");
    std::vector<int64_t> keys;
    {
        keys.reserve(n);
        for (const RecordBatch& b : in) {
            const int ci = resolve_column(*p.group_by[0], b);
            if (ci < 0) return herr(GOLAP_ERR_INVALID_ARG, "Column '" + p.group_by[0]->name + "' not found in batch");
            const Column& c = b.cols[ci];
            if (c.dtype == DType::Int64) { const int64_t* k = (const int64_t*)c.data.data(); keys.insert(keys.end(), k, k + b.num_rows); }
            else if (c.dtype == DType::Int32) { const int32_t* k = (const int32_t*)c.data.data(); for (uint64_t i = 0; i < b.num_rows; ++i) keys.push_back(k[i]); }
            else return herr(GOLAP_ERR_UNSUPPORTED, "GpuAggregate: group key must be Int64 or Int32");
        }
    }
    ob.num_rows = 0;
    Column kc; kc.name = p.group_by[0]->name; kc.dtype = DType::Int64;
    std::vector<Column> acs(na);
    for (uint32_t i = 0; i < na; ++i) { acs[i].name = std::string(agg_name(p.aggr_funcs[i])) + "(" + p.aggr_inputs[i]->name + ")"; acs[i].dtype = DType::Float64; }
    if (n) {
        DevBuf dk, dmm;
        H_TRY(dk.alloc(ctx, n * 8));
        H_TRY(dmm.alloc(ctx, 16));
        H_TRY(h2d(keys.data(), dk.ptr, n * 8));
        H_TRY(xsync());
        // key range on the device: ungrouped MIN / MAX of the key column
        golap_aggregate_spec mm[2] = {{GOLAP_AGG_MIN, 0, 0}, {GOLAP_AGG_MAX, 0, 1}};
        uint64_t kp = dk.ptr; int32_t kd = GOLAP_DT_INT64;
        golap_aggregate_params ap;
        memset(&ap, 0, sizeof ap);
        ap.value_ptrs = &kp; ap.n_value_cols = 1; ap.n_rows = n; ap.n_groups = 1; ap.out_ptr = dmm.ptr; ap.value_dtypes = &kd;
        H_TRY(golap_aggregate(ctx, mm, 2, &ap, nullptr));
        H_TRY(ksync());
        double mmv[2];
        H_TRY(d2h(dmm.ptr, mmv, 16));
        H_TRY(xsync());
        // dense keys (non-negative, range not much wider than the row count) index the group table directly (slot = key % G,
        // join_kernels.cuh:230); anything else -- negative, sparse, hash-like keys -- goes through the key -> slot hash table
        // (GOLAP_KEYS_HASHED, aggregate_kernel.rs:76-77)
        const bool dense_keys = mmv[0] >= 0 && mmv[1] <= 9e15 && (mmv[1] - mmv[0]) < (double)(64ull << 20) && (mmv[1] - mmv[0]) <= 4.0 * (double)n + 1024.0;
        if (!dense_keys) {
            const uint64_t bound = std::min<uint64_t>(n, 1ull << 28);   // distinct keys <= rows; the key table holds at most 2^28 groups
            DevBuf dout, dcnt, dkeys, dn;
            H_TRY(dout.alloc(ctx, (uint64_t)std::max<uint32_t>(na, 1) * bound * 8));
            H_TRY(dcnt.alloc(ctx, bound * 8));
            H_TRY(dkeys.alloc(ctx, bound * 8));
            H_TRY(dn.alloc(ctx, 8));
            memset(&ap, 0, sizeof ap);
            ap.group_key_ptr = dk.ptr; ap.value_ptrs = vptrs.data(); ap.n_value_cols = nv; ap.n_rows = n; ap.n_groups = (uint32_t)bound;
            ap.value_validity_ptrs = any_value_nulls ? vvptrs.data() : nullptr;
            ap.out_ptr = dout.ptr; ap.key_dtype = GOLAP_DT_INT64; ap.value_dtypes = vdt.data(); ap.out_count_ptr = dcnt.ptr;
            ap.key_mode = GOLAP_KEYS_HASHED; ap.out_keys_ptr = dkeys.ptr; ap.out_n_groups_ptr = dn.ptr;
            if (na == 0) {
                golap_aggregate_spec cs{GOLAP_AGG_COUNT, 0, 0};
                ap.value_ptrs = &kp; ap.n_value_cols = 1; ap.value_dtypes = &kd; ap.value_validity_ptrs = nullptr;
                H_TRY(golap_aggregate(ctx, &cs, 1, &ap, nullptr));
            } else {
                H_TRY(golap_aggregate(ctx, specs.data(), na, &ap, nullptr));
            }
            H_TRY(ksync());
            uint32_t nd = 0;
            H_TRY(d2h(dn.ptr, &nd, 4));
            H_TRY(xsync());
            if (nd > bound) return herr(GOLAP_ERR_UNSUPPORTED, "GpuAggregate: more than 2^28 distinct group keys");
            std::vector<int64_t> gk(nd);
            std::vector<double> res((size_t)std::max<uint32_t>(na, 1) * bound);
            if (nd) H_TRY(d2h(dkeys.ptr, gk.data(), (uint64_t)nd * 8));
            H_TRY(d2h(dout.ptr, res.data(), res.size() * 8));
            H_TRY(xsync());
            std::vector<uint32_t> order(nd);
            for (uint32_t i = 0; i < nd; ++i) order[i] = i;
            std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return gk[a] < gk[b]; });   // rows sorted by key
            for (uint32_t slot : order) {
                kc.data.insert(kc.data.end(), (const uint8_t*)&gk[slot], (const uint8_t*)&gk[slot] + 8);
                for (uint32_t i = 0; i < na; ++i) {
                    const double v = res[(size_t)i * bound + slot];
                    acs[i].data.insert(acs[i].data.end(), (const uint8_t*)&v, (const uint8_t*)&v + 8);
                }
                ob.num_rows++;
            }
            kc.rows = ob.num_rows;
            ob.cols.push_back(std::move(kc));
            for (auto& c : acs) { c.rows = ob.num_rows; ob.cols.push_back(std::move(c)); }
            out.push_back(std::move(ob));
            return GOLAP_OK;
        }
        const int64_t kmin = (int64_t)mmv[0], kmax = (int64_t)mmv[1];
        const uint64_t G = (uint64_t)(kmax - kmin) + 1;
        DevBuf dout, dcnt;
        H_TRY(dout.alloc(ctx, (uint64_t)std::max<uint32_t>(na, 1) * G * 8));
        H_TRY(dcnt.alloc(ctx, G * 8));
        memset(&ap, 0, sizeof ap);
        ap.group_key_ptr = dk.ptr; ap.value_ptrs = vptrs.data(); ap.n_value_cols = nv; ap.n_rows = n; ap.n_groups = (uint32_t)G;
        ap.value_validity_ptrs = any_value_nulls ? vvptrs.data() : nullptr;
        ap.out_ptr = dout.ptr; ap.key_dtype = GOLAP_DT_INT64; ap.value_dtypes = vdt.data(); ap.out_count_ptr = dcnt.ptr;
        if (na == 0) {   // GROUP BY without aggregates: count only, to find the present groups
            golap_aggregate_spec cs{GOLAP_AGG_COUNT, 0, 0};
            ap.value_ptrs = &kp; ap.n_value_cols = 1; ap.value_dtypes = &kd; ap.value_validity_ptrs = nullptr;
            H_TRY(golap_aggregate(ctx, &cs, 1, &ap, nullptr));
        } else {
            H_TRY(golap_aggregate(ctx, specs.data(), na, &ap, nullptr));
            if (any_value_nulls) {
                // out_count counted the non-NULL values of the last column; which groups EXIST is a question about rows: count the
                // key column (no NULLs) once more into dcnt (a group whose values are all NULL is still a group)
                DevBuf dtmp;
                H_TRY(dtmp.alloc(ctx, G * 8));
                golap_aggregate_spec cs{GOLAP_AGG_COUNT, 0, 0};
                golap_aggregate_params cp = ap;
                cp.value_ptrs = &kp; cp.n_value_cols = 1; cp.value_dtypes = &kd; cp.value_validity_ptrs = nullptr; cp.out_ptr = dtmp.ptr;
                H_TRY(golap_aggregate(ctx, &cs, 1, &cp, nullptr));
                H_TRY(ksync());
            }
        }
        H_TRY(ksync());
        std::vector<double> res((size_t)std::max<uint32_t>(na, 1) * G);
        std::vector<uint64_t> cnt(G);
        H_TRY(d2h(dout.ptr, res.data(), res.size() * 8));
        H_TRY(d2h(dcnt.ptr, cnt.data(), G * 8));
        H_TRY(xsync());
        for (int64_t k = kmin; k <= kmax; ++k) {           // rows sorted by key
            const uint64_t slot = (uint64_t)k % G;
            if (!cnt[slot]) continue;
            kc.data.insert(kc.data.end(), (const uint8_t*)&k, (const uint8_t*)&k + 8);
            for (uint32_t i = 0; i < na; ++i) {
                const double v = res[(size_t)i * G + slot];
                acs[i].data.insert(acs[i].data.end(), (const uint8_t*)&v, (const uint8_t*)&v + 8);
            }
            ob.num_rows++;
        }
    }
    kc.rows = ob.num_rows;
    ob.cols.push_back(std::move(kc));
    for (auto& c : acs) { c.rows = ob.num_rows; ob.cols.push_back(std::move(c)); }
    out.push_back(std::move(ob));
    return GOLAP_OK;
}

// record_batch_to_gpu_buffers for Utf8 columns (record_batch_convert.rs:93-97, :112-130): every string column of every input batch
// becomes its FNV-1a 64 hashes (as Int64, null -> 0) -- on the device: offsets and bytes go up, one thread hashes one value.  From
// here on the column is what the reference's GPU path sees: an 8-byte key that joins, groups and filters like any Int64 (two different
// strings with the same 64-bit hash are one key, as in the reference; the column comes back as its hash).
int32_t Exec::hash_strings(golap_plan& p) {
    if (p.input) H_TRY(hash_strings(*p.input));
    if (p.right) H_TRY(hash_strings(*p.right));
    for (RecordBatch& b : p.batches)
        for (Column& c : b.cols) {
            if (c.dtype != DType::Utf8Raw) continue;
            const uint64_t n = b.num_rows;
            c.data.assign(n * 8, 0);
            if (n) {
                DevBuf doff, dbytes, dvalid, dout;
                H_TRY(doff.alloc(ctx, (n + 1) * 8));
                H_TRY(dbytes.alloc(ctx, std::max<size_t>(c.str_bytes.size(), 8)));
                H_TRY(dout.alloc(ctx, n * 8));
                H_TRY(h2d(c.str_off.data(), doff.ptr, (n + 1) * 8));
                if (!c.str_bytes.empty()) H_TRY(h2d(c.str_bytes.data(), dbytes.ptr, c.str_bytes.size()));
                if (!c.in_valid.empty()) {
                    H_TRY(dvalid.alloc(ctx, n));
                    H_TRY(h2d(c.in_valid.data(), dvalid.ptr, n));
                }
                H_TRY(xsync());
                H_TRY(golap_hash_utf8(ctx, doff.ptr, dbytes.ptr, c.in_valid.empty() ? 0 : dvalid.ptr, n, dout.ptr, nullptr));
                H_TRY(ksync());
                H_TRY(d2h(dout.ptr, c.data.data(), n * 8));
                H_TRY(xsync());
            }
            c.dtype = DType::Int64;
            c.str_off.clear(); c.str_off.shrink_to_fit();
            c.str_bytes.clear(); c.str_bytes.shrink_to_fit();
        }
    return GOLAP_OK;
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
        if (arrays[i] && arrays[i]->release) arrays[i]->release(arrays[i]);      // moved in
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
    // ctx may be NULL for plans that never touch the device (Batches / TableScan / Sort / Projection);
    // any device step then fails inside the C ABI with its own "null ctx" error.
    if (!plan || !out_result || !out_n) return herr(GOLAP_ERR_INVALID_ARG, "null argument");
    g_err.clear();
    Exec ex{ctx};
    auto res = std::make_unique<golap_result>();
    int32_t s = ex.hash_strings(*plan);
    if (s == GOLAP_OK) s = ex.run(*plan, res->batches);
    if (s != GOLAP_OK) {
        if (g_err.empty()) g_err = golap_last_error(ctx);
        if (g_err.empty()) g_err = golap_last_error(nullptr);
        return s;
    }
    *out_n = (uint32_t)res->batches.size();
    *out_result = res.release();
    return GOLAP_OK;
}

int32_t golap_result_export(golap_result* r, uint32_t i, ArrowArray* oa, ArrowSchema* os) {
    if (!r || !oa || !os || i >= r->batches.size()) return herr(GOLAP_ERR_INVALID_ARG, "bad result index");
    export_batch(std::move(r->batches[i]), oa, os);
    return GOLAP_OK;
}
void golap_result_free(golap_result* r) { delete r; }

}  // extern "C"
