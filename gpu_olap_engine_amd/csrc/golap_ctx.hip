// golap_ctx.hip -- context, errors, profiler, workspace, SlabAllocator, TransferQueue.
#include "golap_internal.h"

#include <algorithm>

namespace golap {

static thread_local std::string g_thread_err;
thread_local const Knobs* g_knobs = nullptr;
thread_local const golap_engine_config* g_scope_cfg = nullptr;

void set_thread_error(const char* msg) { g_thread_err = msg ? msg : ""; }

int32_t fail(golap_ctx* ctx, int32_t code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_thread_err = buf;
    if (ctx) {
        std::lock_guard<std::mutex> lk(ctx->err_mu);
        ctx->err = buf;
    }
    return code;
}

// ---------------------------------------------------------------------------
// Profiler
// ---------------------------------------------------------------------------
uint32_t Profiler::stat_index(const char* name) {
    for (uint32_t i = 0; i < stats.size(); ++i)
        if (!strncmp(stats[i].name, name, sizeof(stats[i].name) - 1)) return i;
    golap_kernel_stat s;
    memset(&s, 0, sizeof s);
    strncpy(s.name, name, sizeof(s.name) - 1);
    stats.push_back(s);
    return (uint32_t)stats.size() - 1;
}

hipEvent_t Profiler::get_event() {
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

uint32_t* Profiler::take_gate_slot() {
    const uint32_t blk = gate_used / GATE_BLOCK, idx = gate_used % GATE_BLOCK;
    if (blk >= gate_blocks.size()) {
        uint32_t* b = nullptr;
        if (hipHostMalloc((void**)&b, GATE_BLOCK * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        gate_blocks.push_back(b);
    }
    ++gate_used;
    uint32_t* s = gate_blocks[blk] + idx;
    *s = GATE_PENDING;
    return s;
}

void Profiler::resolve() {
    for (auto& p : pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.stop) == hipSuccess && hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess)
            stats[p.stat].total_ms += (double)ms;
        // (the stop event of a call's last kernel precedes the copy of its gate word on the stream: wait for the stream's
        // tail through the LAST pending event of the list -- events are recorded in order -- before reading any slot)
        pool.push_back(p.start);
        pool.push_back(p.stop);
    }
    if (!pending.empty()) {
        bool any = false;
        for (auto& p : pending) any = any || p.gate_host;
        if (any) (void)hipDeviceSynchronize();   // the slot copies sit behind the events just waited for; profiling runs only
        for (auto& p : pending)
            if (p.gate_host && *p.gate_host != GATE_PENDING && ((*p.gate_host != 0u) == p.credit_when_set)) stats[p.stat].bytes += p.cond_bytes;
    }
    pending.clear();
    (void)recent_gate();      // (what has landed is remembered; the slots themselves are handed out again from here on)
    noted_gates.clear();
    gate_used = 0;
}

void Profiler::destroy() {
    resolve();
    for (auto e : pool) (void)hipEventDestroy(e);
    pool.clear();
    for (auto b : gate_blocks) (void)hipHostFree(b);
    gate_blocks.clear();
}

// ---------------------------------------------------------------------------
// SlabAllocator -- gpu-memory-manager/src/slab_allocator.rs
// ---------------------------------------------------------------------------
void SlabAllocator::init(uint64_t max_b, uint32_t max_slabs_per_class) {
    static const uint64_t MiB = 1ull << 20, GiB = 1ull << 30;
    // README.md:142 classes 1/4/16/64/256 MiB, extended upwards so whole columns fit.
    const uint64_t sizes[] = {1 * MiB,  4 * MiB,  16 * MiB, 64 * MiB, 256 * MiB, 1 * GiB,  2 * GiB,  4 * GiB,
                              8 * GiB,  16 * GiB, 24 * GiB, 32 * GiB, 48 * GiB,  64 * GiB, 96 * GiB, 128 * GiB};
    classes.clear();
    for (uint64_t s : sizes) {
        SlabClass c;
        c.size = s;
        c.max_slabs = max_slabs_per_class;
        classes.push_back(c);
    }
    max_bytes = max_b;
    bytes_reserved = 0;
}

int32_t SlabAllocator::allocate(golap_ctx* ctx, uint64_t size, uint64_t* ptr, uint32_t* cls) {
    // find_slab_class, slab_allocator.rs:95-107: first class >= size, else error.
    uint32_t idx = (uint32_t)classes.size();
    for (uint32_t i = 0; i < classes.size(); ++i)
        if (classes[i].size >= size) { idx = i; break; }
    if (idx == classes.size())
        return fail(ctx, GOLAP_ERR_OOM, "Requested size %llu exceeds largest slab size %llu", (unsigned long long)size,
                    (unsigned long long)classes.back().size);
    std::lock_guard<std::mutex> lk(mu);
    SlabClass& c = classes[idx];
    if (!c.free_slabs.empty()) {                       // :57-60 reuse
        *ptr = c.free_slabs.front();
        c.free_slabs.pop_front();
        *cls = idx;
        return GOLAP_OK;
    }
    if (c.allocated_count >= c.max_slabs)              // :112-118
        return fail(ctx, GOLAP_ERR_OOM, "Maximum number of slabs (%u) reached for class %u", c.max_slabs, idx);
    if (max_bytes && bytes_reserved + c.size > max_bytes) {
        // give pooled slabs of other classes back to the device before refusing
        for (auto& oc : classes) {
            while (!oc.free_slabs.empty()) {
                (void)hipFree((void*)oc.free_slabs.front());
                oc.free_slabs.pop_front();
                oc.allocated_count -= 1;
                bytes_reserved -= oc.size;
            }
        }
        if (bytes_reserved + c.size > max_bytes)
            return fail(ctx, GOLAP_ERR_OOM, "max_gpu_memory (%llu) would be exceeded by a %llu-byte slab",
                        (unsigned long long)max_bytes, (unsigned long long)c.size);
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, c.size);              // :132-149 (really allocates; nothing is leaked)
    if (e != hipSuccess) {   // the device is full: pooled slabs of every class go back to it before giving up
        (void)hipGetLastError();
        for (auto& oc : classes)
            while (!oc.free_slabs.empty()) {
                (void)hipFree((void*)oc.free_slabs.front());
                oc.free_slabs.pop_front();
                oc.allocated_count -= 1;
                bytes_reserved -= oc.size;
            }
        e = hipMalloc(&p, c.size);
    }
    if (e != hipSuccess) return fail(ctx, GOLAP_ERR_OOM, "hipMalloc(%llu) failed: %s", (unsigned long long)c.size, hipGetErrorString(e));
    c.allocated_count += 1;
    bytes_reserved += c.size;
    *ptr = (uint64_t)p;
    *cls = idx;
    return GOLAP_OK;
}

int32_t SlabAllocator::release(golap_ctx* ctx, uint64_t ptr, uint32_t cls) {
    if (cls >= classes.size()) return fail(ctx, GOLAP_ERR_INVALID_ARG, "Invalid slab class index: %u", cls);   // :72-74
    std::lock_guard<std::mutex> lk(mu);
    SlabClass& c = classes[cls];
    if (c.free_slabs.size() >= std::min<uint32_t>(c.max_slabs, 16u)) {   // :80-86: really free instead of only forgetting (the pool keeps <= 16 per class)
        (void)hipFree((void*)ptr);
        c.allocated_count -= 1;
        bytes_reserved -= c.size;
    } else {
        c.free_slabs.push_back(ptr);                   // :88-89
    }
    return GOLAP_OK;
}

void SlabAllocator::destroy() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& c : classes) {
        for (uint64_t p : c.free_slabs) (void)hipFree((void*)p);   // the free loop slab_allocator.rs:161-164 leaves commented out
        c.free_slabs.clear();
    }
}

// ---------------------------------------------------------------------------
// TransferQueue -- gpu-memory-manager/src/transfer_queue.rs
// ---------------------------------------------------------------------------
int32_t TransferQueue::init(golap_ctx* ctx, uint32_t num_streams) {
    bounce_bytes = 8ull << 20;
    for (uint32_t i = 0; i < num_streams; ++i) {
        XferStream* s = new XferStream();
        GOLAP_HIP_TRY(ctx, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        streams.push_back(s);
    }
    return GOLAP_OK;
}

static bool is_pinned_or_device(const void* p) {
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost || a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

int32_t TransferQueue::enqueue(golap_ctx* ctx, int32_t direction, uint64_t src, uint64_t dst, uint64_t size, uint32_t stream_id) {
    if (streams.empty()) return fail(ctx, GOLAP_ERR_INVALID_ARG, "TransferQueue has no streams");
    if (direction != GOLAP_HOST_TO_DEVICE && direction != GOLAP_DEVICE_TO_HOST)
        return fail(ctx, GOLAP_ERR_INVALID_ARG, "bad transfer direction %d", direction);
    if (size == 0) return GOLAP_OK;
    XferStream* s = streams[stream_id % streams.size()];          // transfer_queue.rs:72
    std::lock_guard<std::mutex> lk(s->mu);                        // :82-83
    const hipMemcpyKind kind = direction == GOLAP_HOST_TO_DEVICE ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
    const void* host = direction == GOLAP_HOST_TO_DEVICE ? (const void*)src : (const void*)dst;
    if (is_pinned_or_device(host)) {
        hipEvent_t permit = nullptr;
        {   // :49, :63-64: a permit per transfer in flight, 4N in all
            std::lock_guard<std::mutex> pk(permit_mu);
            const size_t slots = 4 * streams.size();
            if (permits.size() < slots) { permits.resize(slots, nullptr); permit_used.resize(slots, 0); }
            const size_t k = (size_t)(permit_next++ % slots);
            if (!permits[k]) GOLAP_HIP_TRY(ctx, hipEventCreateWithFlags(&permits[k], hipEventDisableTiming));
            if (permit_used[k]) {
                if (hipEventQuery(permits[k]) != hipSuccess) { (void)hipGetLastError(); ++permit_waits; }
                GOLAP_HIP_TRY(ctx, hipEventSynchronize(permits[k]));
            }
            permit_used[k] = 1;
            permit = permits[k];
            GOLAP_HIP_TRY(ctx, hipMemcpyAsync((void*)dst, (const void*)src, size, kind, s->stream));   // :93 / :115
            GOLAP_HIP_TRY(ctx, hipEventRecord(permit, s->stream));
        }
        return GOLAP_OK;
    }
    // Pageable host memory: stage through two pinned bounce buffers so the copy
    // of chunk k overlaps the host memcpy of chunk k+1 (double buffering).
    for (int b = 0; b < 2; ++b)
        if (!s->bounce[b]) {
            GOLAP_HIP_TRY(ctx, hipHostMalloc(&s->bounce[b], bounce_bytes, hipHostMallocDefault));
            GOLAP_HIP_TRY(ctx, hipEventCreateWithFlags(&s->bounce_free[b], hipEventDisableTiming));
            GOLAP_HIP_TRY(ctx, hipEventRecord(s->bounce_free[b], s->stream));
        }
    uint64_t done = 0;
    while (done < size) {
        const uint64_t n = std::min(bounce_bytes, size - done);
        const uint32_t b = s->next_bounce;
        s->next_bounce ^= 1u;
        GOLAP_HIP_TRY(ctx, hipEventSynchronize(s->bounce_free[b]));
        if (direction == GOLAP_HOST_TO_DEVICE) {
            memcpy(s->bounce[b], (const char*)src + done, n);
            GOLAP_HIP_TRY(ctx, hipMemcpyAsync((char*)dst + done, s->bounce[b], n, hipMemcpyHostToDevice, s->stream));
            GOLAP_HIP_TRY(ctx, hipEventRecord(s->bounce_free[b], s->stream));
        } else {
            GOLAP_HIP_TRY(ctx, hipMemcpyAsync(s->bounce[b], (const char*)src + done, n, hipMemcpyDeviceToHost, s->stream));
            GOLAP_HIP_TRY(ctx, hipEventRecord(s->bounce_free[b], s->stream));
            GOLAP_HIP_TRY(ctx, hipEventSynchronize(s->bounce_free[b]));
            memcpy((char*)dst + done, s->bounce[b], n);
        }
        done += n;
    }
    return GOLAP_OK;
}

int32_t TransferQueue::synchronize(golap_ctx* ctx) {
    for (XferStream* s : streams) {                               // transfer_queue.rs:133-139
        std::lock_guard<std::mutex> lk(s->mu);
        GOLAP_HIP_TRY(ctx, hipStreamSynchronize(s->stream));
    }
    return GOLAP_OK;
}

void TransferQueue::destroy() {
    for (XferStream* s : streams) {
        (void)hipStreamSynchronize(s->stream);
        for (int b = 0; b < 2; ++b) {
            if (s->bounce[b]) (void)hipHostFree(s->bounce[b]);
            if (s->bounce_free[b]) (void)hipEventDestroy(s->bounce_free[b]);
        }
        (void)hipStreamDestroy(s->stream);
        delete s;
    }
    streams.clear();
    for (hipEvent_t e : permits)
        if (e) (void)hipEventDestroy(e);
    permits.clear();
    permit_used.clear();
}

// ---------------------------------------------------------------------------
// zero_async
// ---------------------------------------------------------------------------
__global__ void k_zero_u32(uint32_t* __restrict__ a, unsigned long long n) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) a[i] = 0u;
}

int32_t zero_async(golap_ctx* ctx, hipStream_t st, void* ptr, uint64_t bytes) {
    if (!bytes) return GOLAP_OK;
    if ((uint64_t)(uintptr_t)ptr & 3) return fail(ctx, GOLAP_ERR_INVALID_ARG, "zero_async: pointer must be 4-byte aligned");
    const unsigned long long n = (bytes + 3) / 4;
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)ctx->prop.multiProcessorCount * 8, (n + 255) / 256));
    hipLaunchKernelGGL(k_zero_u32, dim3(grid), dim3(256), 0, st, (uint32_t*)ptr, n);
    GOLAP_HIP_TRY(ctx, hipGetLastError());
    return GOLAP_OK;
}

// ---------------------------------------------------------------------------
// Workspace
// ---------------------------------------------------------------------------
int32_t ws_begin(golap_ctx* ctx, uint64_t bytes, hipStream_t st) {
    if (ctx->ws_owned)
        return fail(ctx, GOLAP_ERR_INVALID_ARG, "the context's workspace belongs to a join opened with golap_join_partitioned_begin: finish or abort it first");
    ctx->err.clear();   // a new compute call starts: golap_last_error reports its failures, not an earlier call's
    bytes = round_up(bytes + 4096, 4096);
    if (ctx->ws.size < bytes) {
        if (ctx->ws.base) {
            // the old slab may still be in use by queued kernels
            if (ctx->ws_done) (void)hipStreamSynchronize(ctx->ws_stream);
            GOLAP_TRY(ctx->slabs.release(ctx, ctx->ws.base, ctx->ws.cls));
            ctx->ws = Workspace();
        }
        uint64_t p = 0;
        uint32_t cls = 0;
        GOLAP_TRY(ctx->slabs.allocate(ctx, bytes, &p, &cls));
        ctx->ws.base = p;
        ctx->ws.cls = cls;
        ctx->ws.size = ctx->slabs.classes[cls].size;
    }
    if (ctx->ws_done && ctx->ws_stream != st)
        GOLAP_HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ws_done, 0));
    ctx->ws.used = 0;
    return GOLAP_OK;
}

uint64_t ws_take(golap_ctx* ctx, uint64_t bytes) {
    const uint64_t o = round_up(ctx->ws.used, 256);
    if (o + bytes > ctx->ws.size) return 0;
    ctx->ws.used = o + bytes;
    return ctx->ws.base + o;
}

void ws_end(golap_ctx* ctx, hipStream_t st) {
    if (!ctx->ws_done) (void)hipEventCreateWithFlags(&ctx->ws_done, hipEventDisableTiming);
    (void)hipEventRecord(ctx->ws_done, st);
    ctx->ws_stream = st;
}

int32_t aux_begin(golap_ctx* ctx, uint64_t bytes, hipStream_t st) {
    bytes = round_up(bytes + 4096, 4096);
    if (ctx->aux.size < bytes) {
        if (ctx->aux.base) {
            if (ctx->aux_done) (void)hipStreamSynchronize(ctx->aux_stream);   // the old slab may still be in use by queued kernels
            GOLAP_TRY(ctx->slabs.release(ctx, ctx->aux.base, ctx->aux.cls));
            ctx->aux = Workspace();
        }
        uint64_t p = 0;
        uint32_t cls = 0;
        GOLAP_TRY(ctx->slabs.allocate(ctx, bytes, &p, &cls));
        ctx->aux.base = p;
        ctx->aux.cls = cls;
        ctx->aux.size = ctx->slabs.classes[cls].size;
    }
    if (ctx->aux_done && ctx->aux_stream != st) GOLAP_HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->aux_done, 0));
    ctx->aux.used = 0;
    return GOLAP_OK;
}

uint64_t aux_take(golap_ctx* ctx, uint64_t bytes) {
    const uint64_t o = round_up(ctx->aux.used, 256);
    if (o + bytes > ctx->aux.size) return 0;
    ctx->aux.used = o + bytes;
    return ctx->aux.base + o;
}

void aux_end(golap_ctx* ctx, hipStream_t st) {
    if (!ctx->aux_done) (void)hipEventCreateWithFlags(&ctx->aux_done, hipEventDisableTiming);
    (void)hipEventRecord(ctx->aux_done, st);
    ctx->aux_stream = st;
}

}  // namespace golap

using namespace golap;

// ---------------------------------------------------------------------------
// C ABI: context
// ---------------------------------------------------------------------------
extern "C" {

uint32_t golap_abi_version(void) { return GOLAP_ABI_VERSION; }

void golap_engine_config_default(golap_engine_config* cfg) {
    if (!cfg) return;
    memset(cfg, 0, sizeof *cfg);
    cfg->max_gpu_memory = 8ull * 1024 * 1024 * 1024;   // lib.rs:36
    cfg->num_streams = 8;                              // lib.rs:37
    cfg->use_unified_memory = 1;                       // lib.rs:38
    cfg->batch_size = 1024 * 1024;                     // lib.rs:39
    cfg->enable_cache = 1;                             // lib.rs:40
    cfg->device_id = -1;
    cfg->radix_bits = 8;                               // join_kernels.cuh:22
    cfg->enable_profiling = 0;
    cfg->max_slabs_per_class = 4096;
    cfg->provision_sigmas = 0;
}

const char* golap_last_error(const golap_ctx* ctx) {
    if (ctx && !ctx->err.empty()) return ctx->err.c_str();
    return g_thread_err.c_str();
}

int32_t golap_ctx_create(const golap_engine_config* cfg, golap_ctx** out_ctx) {
    if (!out_ctx) return fail(nullptr, GOLAP_ERR_INVALID_ARG, "out_ctx is null");
    *out_ctx = nullptr;
    golap_engine_config c;
    if (cfg) c = *cfg; else golap_engine_config_default(&c);
    if (c.radix_bits != 8) return fail(nullptr, GOLAP_ERR_INVALID_ARG, "radix_bits must be 8 (join_kernels.cuh:22), got %u", c.radix_bits);
    if (c.num_streams == 0 || c.num_streams > 64) return fail(nullptr, GOLAP_ERR_INVALID_ARG, "num_streams must be in 1..64");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, GOLAP_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    }
    golap_ctx* ctx = new golap_ctx();
    ctx->cfg = c;
    if (c.device_id >= 0) {
        if (c.device_id >= ndev) { delete ctx; return fail(nullptr, GOLAP_ERR_INVALID_ARG, "device_id %d out of range (%d devices)", c.device_id, ndev); }
        ctx->device = c.device_id;
        e = hipSetDevice(ctx->device);
    } else {
        e = hipGetDevice(&ctx->device);
    }
    if (e != hipSuccess) { delete ctx; return fail(nullptr, GOLAP_ERR_HIP, "selecting device failed: %s", hipGetErrorString(e)); }
    e = hipGetDeviceProperties(&ctx->prop, ctx->device);
    if (e != hipSuccess) { delete ctx; return fail(nullptr, GOLAP_ERR_HIP, "hipGetDeviceProperties failed: %s", hipGetErrorString(e)); }
    ctx->slabs.init(c.max_gpu_memory, c.max_slabs_per_class ? c.max_slabs_per_class : 4096);
    int32_t s = ctx->xfer.init(ctx, c.num_streams);
    if (s != GOLAP_OK) { golap_ctx_destroy(ctx); return s; }
    ctx->prof.enabled = c.enable_profiling != 0;
    ctx->prof.light = c.enable_profiling >= 2;
    if (getenv("GOLAP_DEBUG_ENV")) {   // developer convenience: import GOLAP_* variables ONCE as knobs (scripts/, multi-process tests)
        extern char** environ;
        for (char** e = environ; e && *e; ++e)
            if (!strncmp(*e, "GOLAP_", 6)) {
                const char* eq = strchr(*e, '=');
                if (eq) ctx->knobs.set(std::string(*e, eq - *e).c_str(), eq + 1);
            }
    }
    *out_ctx = ctx;
    return GOLAP_OK;
}

void golap_ctx_destroy(golap_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    ctx->prof.destroy();
    ctx->xfer.destroy();
    if (ctx->ws.base) (void)ctx->slabs.release(ctx, ctx->ws.base, ctx->ws.cls);
    if (ctx->aux.base) (void)ctx->slabs.release(ctx, ctx->aux.base, ctx->aux.cls);
    if (ctx->aux_done) (void)hipEventDestroy(ctx->aux_done);
    ctx->slabs.destroy();
    if (ctx->ws_done) (void)hipEventDestroy(ctx->ws_done);
    delete ctx;
}

int32_t golap_debug_set(golap_ctx* ctx, const char* name, const char* value) {
    if (!ctx || !name) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->knobs.set(name, value);
    return GOLAP_OK;
}

int32_t golap_device_info_get(golap_ctx* ctx, golap_device_info* out) {
    if (!ctx || !out) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    memset(out, 0, sizeof *out);
    strncpy(out->name, ctx->prop.name, sizeof(out->name) - 1);
    strncpy(out->gcn_arch, ctx->prop.gcnArchName, sizeof(out->gcn_arch) - 1);
    out->compute_units = (uint32_t)ctx->prop.multiProcessorCount;
    out->lds_bytes_per_block = (uint32_t)ctx->prop.sharedMemPerBlock;
    out->wavefront_size = (uint32_t)ctx->prop.warpSize;
    out->clock_mhz = (uint32_t)(ctx->prop.clockRate / 1000);
    out->memory_clock_mhz = (uint32_t)(ctx->prop.memoryClockRate / 1000);
    out->memory_bus_bits = (uint32_t)ctx->prop.memoryBusWidth;
    out->l2_bytes = (uint32_t)ctx->prop.l2CacheSize;
    size_t fr = 0, tot = 0;
    GOLAP_HIP_TRY(ctx, hipMemGetInfo(&fr, &tot));
    out->total_mem_bytes = tot;
    out->free_mem_bytes = fr;
    {
        KnobScope knobs_(ctx);
        out->hidden_memory_forms = kernel_forms_now().hidden ? 1u : 0u;
    }
    snprintf(out->built_with, sizeof out->built_with, "hip %d.%d.%d", (int)GOLAP_BUILD_HIP_MAJOR, (int)GOLAP_BUILD_HIP_MINOR, (int)GOLAP_BUILD_HIP_PATCH);
    snprintf(out->audited_with, sizeof out->audited_with, "hip %d.%d.%d", GOLAP_HIDDEN_AUDITED_HIP_MAJOR, GOLAP_HIDDEN_AUDITED_HIP_MINOR, GOLAP_HIDDEN_AUDITED_HIP_PATCH);
    return GOLAP_OK;
}

int32_t golap_ctx_synchronize(golap_ctx* ctx, void* stream) {
    if (!ctx) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null ctx");
    GOLAP_HIP_TRY(ctx, hipStreamSynchronize(pick_stream(ctx, stream)));
    return GOLAP_OK;
}

// ---- SlabAllocator ----------------------------------------------------------
int32_t golap_slab_alloc(golap_ctx* ctx, uint64_t size, uint64_t* out_ptr, uint32_t* out_class_idx) {
    if (!ctx || !out_ptr || !out_class_idx) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    if (size == 0) return fail(ctx, GOLAP_ERR_INVALID_ARG, "size must be > 0");
    return ctx->slabs.allocate(ctx, size, out_ptr, out_class_idx);
}

int32_t golap_slab_free(golap_ctx* ctx, uint64_t ptr, uint32_t class_idx) {
    if (!ctx) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null ctx");
    return ctx->slabs.release(ctx, ptr, class_idx);
}

int32_t golap_slab_get_stats(golap_ctx* ctx, golap_slab_stats* out) {
    if (!ctx || !out) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    memset(out, 0, sizeof *out);
    std::lock_guard<std::mutex> lk(ctx->slabs.mu);
    out->n_classes = (uint32_t)std::min<size_t>(16, ctx->slabs.classes.size());
    for (uint32_t i = 0; i < out->n_classes; ++i) {
        out->class_size[i] = ctx->slabs.classes[i].size;
        out->allocated[i] = ctx->slabs.classes[i].allocated_count;
        out->pooled[i] = (uint32_t)ctx->slabs.classes[i].free_slabs.size();
    }
    out->bytes_reserved = ctx->slabs.bytes_reserved;
    return GOLAP_OK;
}

// ---- TransferQueue ----------------------------------------------------------
int32_t golap_xfer_enqueue(golap_ctx* ctx, int32_t direction, uint64_t src, uint64_t dst, uint64_t size, uint32_t stream_id) {
    if (!ctx) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null ctx");
    if (size && (!src || !dst)) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null src/dst");
    return ctx->xfer.enqueue(ctx, direction, src, dst, size, stream_id);
}

int32_t golap_xfer_synchronize(golap_ctx* ctx) {
    if (!ctx) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null ctx");
    return ctx->xfer.synchronize(ctx);
}

// ---- stats ------------------------------------------------------------------
int32_t golap_stats_get(golap_ctx* ctx, golap_stats* out) {
    if (!ctx || !out) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->prof.resolve();
    memset(out, 0, sizeof *out);
    out->n = (uint32_t)std::min<size_t>(GOLAP_MAX_KERNEL_STATS, ctx->prof.stats.size());
    for (uint32_t i = 0; i < out->n; ++i) out->k[i] = ctx->prof.stats[i];
    return GOLAP_OK;
}

int32_t golap_stats_reset(golap_ctx* ctx) {
    if (!ctx) return fail(ctx, GOLAP_ERR_INVALID_ARG, "null ctx");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->prof.resolve();
    ctx->prof.stats.clear();
    return GOLAP_OK;
}

// ---- API constants the reference pins with unit tests --------------------------
golap_join_strategy golap_choose_strategy(uint64_t n_build) {        // join_kernel.rs:71-77
    return n_build <= 1000000ull ? GOLAP_JOIN_BROADCAST_HASH : GOLAP_JOIN_RADIX_HASH;
}

void golap_compute_launch_params(uint64_t n_rows, uint32_t* grid, uint32_t* block) {   // filter_kernel.rs:75-79
    const uint32_t B = 256;
    uint32_t g = ((uint32_t)n_rows + B - 1) / B;
    if (grid) *grid = g > 1 ? g : 1;
    if (block) *block = B;
}

int64_t golap_agg_identity_value(int32_t func) {                     // aggregate_kernel.rs:27-33
    switch (func) {
        case GOLAP_AGG_MIN: return INT64_MAX;
        case GOLAP_AGG_MAX: return INT64_MIN;
        default: return 0;
    }
}

golap_cmp_op golap_parse_cmp_op(const char* op) {                    // executor.rs:457-465
    if (!op) return GOLAP_CMP_NEVER;
    if (!strcmp(op, "=") || !strcmp(op, "==")) return GOLAP_CMP_EQ;
    if (!strcmp(op, "!=") || !strcmp(op, "<>")) return GOLAP_CMP_NE;
    if (!strcmp(op, ">")) return GOLAP_CMP_GT;
    if (!strcmp(op, ">=")) return GOLAP_CMP_GE;
    if (!strcmp(op, "<")) return GOLAP_CMP_LT;
    if (!strcmp(op, "<=")) return GOLAP_CMP_LE;
    return GOLAP_CMP_NEVER;
}

}  // extern "C"
