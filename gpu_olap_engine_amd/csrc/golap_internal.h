// golap_internal.h -- host-side internals shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/golap.h"
#include "golap_variants.h"

namespace golap {

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
void set_thread_error(const char* msg);
int32_t fail(golap_ctx* ctx, int32_t code, const char* fmt, ...);

#define GOLAP_HIP_TRY(ctx, expr)                                                                 \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return ::golap::fail((ctx), GOLAP_ERR_HIP, "%s failed: %s (%s:%d)", #expr,            \
                                 hipGetErrorString(_e), __FILE__, __LINE__);                     \
    } while (0)

#define GOLAP_TRY(expr)                          \
    do {                                         \
        int32_t _s = (expr);                     \
        if (_s != GOLAP_OK) return _s;           \
    } while (0)

// ---------------------------------------------------------------------------
// SlabAllocator (gpu-memory-manager/src/slab_allocator.rs)
// ---------------------------------------------------------------------------
struct SlabClass {
    uint64_t size = 0;
    std::deque<uint64_t> free_slabs;   // slab_allocator.rs:19
    uint32_t max_slabs = 0;            // :20
    uint32_t allocated_count = 0;      // :21
};

struct SlabAllocator {
    std::mutex mu;
    std::vector<SlabClass> classes;
    uint64_t bytes_reserved = 0;
    uint64_t max_bytes = 0;            // EngineConfig::max_gpu_memory (0 = uncapped)
    void init(uint64_t max_bytes, uint32_t max_slabs_per_class);
    int32_t allocate(golap_ctx* ctx, uint64_t size, uint64_t* ptr, uint32_t* cls);
    int32_t release(golap_ctx* ctx, uint64_t ptr, uint32_t cls);
    void destroy();
};

// ---------------------------------------------------------------------------
// TransferQueue (gpu-memory-manager/src/transfer_queue.rs)
// ---------------------------------------------------------------------------
struct XferStream {
    hipStream_t stream = nullptr;
    void* bounce[2] = {nullptr, nullptr};   // pinned double buffer for pageable host memory
    hipEvent_t bounce_free[2] = {nullptr, nullptr};
    uint32_t next_bounce = 0;
    std::mutex mu;                           // transfer_queue.rs:29 (Mutex per stream)
};

struct TransferQueue {
    std::vector<XferStream*> streams;
    uint64_t bounce_bytes = 0;
    // back-pressure, transfer_queue.rs:49 (`Semaphore::new(num_streams * 4)`): at most 4N asynchronous transfers are in flight -- the
    // caller of enqueue number k + 4N waits until transfer k has completed (a ring of 4N events; the reference's permit only spans
    // the enqueue call itself, its copies being no-ops)
    std::mutex permit_mu;
    std::vector<hipEvent_t> permits;
    std::vector<uint8_t> permit_used;
    uint64_t permit_next = 0, permit_waits = 0;
    int32_t init(golap_ctx* ctx, uint32_t num_streams);
    int32_t enqueue(golap_ctx* ctx, int32_t direction, uint64_t src, uint64_t dst, uint64_t size, uint32_t stream_id);
    int32_t synchronize(golap_ctx* ctx);
    void destroy();
};

// ---------------------------------------------------------------------------
// profiling: hipEvent pairs on the launch stream, resolved lazily
// ---------------------------------------------------------------------------
struct PendingEvent {
    uint32_t stat;
    hipEvent_t start, stop;
    // bytes credited only once the call's gate word is known (the optimistic tuple pipeline and the 64-bit pipeline standing by
    // behind it are both enqueued; exactly one of them does the work): `gate_host` is a pinned copy of the device word, written
    // by an async copy behind the call's last kernel; credited when (*gate_host != 0) == credit_when_set
    uint64_t cond_bytes = 0;
    const uint32_t* gate_host = nullptr;
    bool credit_when_set = true;
};

struct Profiler {
    bool enabled = false;
    std::vector<golap_kernel_stat> stats;
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> pool;
    std::vector<uint32_t*> gate_blocks;      // pinned blocks of GATE_BLOCK words
    uint32_t gate_used = 0;                  // words handed out since the last resolve()
    const uint32_t* cur_gate_host = nullptr; // the running call's slot (nullptr: no conditional credits, e.g. during stream capture)
    static constexpr uint32_t GATE_BLOCK = 256;
    static constexpr uint32_t GATE_PENDING = 0xFFFFFFFFu;   // a slot whose copy has not landed yet (gate words are 0 / 1)
    // Timing events are launches of their own on the stream (~4 us each): a join that times its ~17 stand-by launches pays more for
    // measuring them than they cost.  `mute` switches KernelTimer off for a scope; radix_hash_join mutes the stand-by pipeline when the
    // most recent join whose gate word has already landed did not need it (recent_gates: the last calls' pinned slots).  A call whose
    // gate turns out raised after all ran its 64-bit pipeline untimed -- once: the next call sees that word.
    bool mute = false;
    bool light = false;                        // golap_engine_config.enable_profiling == 2: muting is allowed
    std::deque<const uint32_t*> noted_gates;   // slots of the joins since the last resolve(), oldest first (their copies land in order)
    int last_known_gate = -1;                  // 0 / 1: the gate word of the most recent join whose copy has landed; -1: none yet
    void note_gate_slot(const uint32_t* s) { noted_gates.push_back(s); }
    int recent_gate() {
        while (!noted_gates.empty()) {
            const uint32_t v = *(const volatile uint32_t*)noted_gates.front();
            if (v == GATE_PENDING) break;
            last_known_gate = v ? 1 : 0;
            noted_gates.pop_front();
        }
        return last_known_gate;
    }
    uint32_t* take_gate_slot();
    uint32_t stat_index(const char* name);
    hipEvent_t get_event();
    void resolve();
    void destroy();
};

// Developer / test knobs (golap_debug_set): named strings kept in the context.  The library never reads the process
// environment on a compute path; at context creation GOLAP_* variables are imported once, and only when GOLAP_DEBUG_ENV
// is set.  Planning helpers that have no context argument find the calling context's knobs through a thread-local pointer
// installed by KnobScope at every C-ABI entry point.
struct Knobs {
    std::vector<std::pair<std::string, std::string>> kv;
    const char* get(const char* name) const {
        for (const auto& e : kv)
            if (e.first == name) return e.second.c_str();
        return nullptr;
    }
    void set(const char* name, const char* value) {
        for (size_t i = 0; i < kv.size(); ++i)
            if (kv[i].first == name) {
                if (value) kv[i].second = value; else kv.erase(kv.begin() + i);
                return;
            }
        if (value) kv.push_back({name, value});
    }
};
extern thread_local const Knobs* g_knobs;
extern thread_local const golap_engine_config* g_scope_cfg;
inline const char* knob(const char* name) { return g_knobs ? g_knobs->get(name) : nullptr; }
// hidden-memory or plain kernel forms by default: by the compiler that built the library (golap_variants.h), unless GOLAP_HIDDEN_FORMS says otherwise
inline KernelForms kernel_forms_now() { return default_kernel_forms(knob("GOLAP_HIDDEN_FORMS")); }

// Workspace: one slab, bump-allocated per call, cached in the context.
struct Workspace {
    uint64_t base = 0, size = 0, used = 0;
    uint32_t cls = 0;
};

}  // namespace golap

struct golap_ctx {
    golap_engine_config cfg;
    int device = 0;
    hipDeviceProp_t prop;
    std::mutex mu;                    // serialises compute calls on one context (SURVEY 8(b) threading)
    std::mutex err_mu;                // fail() may be reached from several host threads at once (golap_xfer_enqueue is called per stream from many)
    std::string err;
    golap::SlabAllocator slabs;
    golap::TransferQueue xfer;
    golap::Profiler prof;
    golap::Workspace ws;
    hipEvent_t ws_done = nullptr;     // last use of the workspace (cross-stream ordering)
    hipStream_t ws_stream = nullptr;
    golap::Workspace aux;             // second cached slab: scratch of an operator that itself calls another one (which re-begins `ws`)
    hipEvent_t aux_done = nullptr;
    hipStream_t aux_stream = nullptr;
    std::shared_ptr<void> pjs;        // golap::PartJoinState (golap_join.hip)
    golap::Knobs knobs;               // golap_debug_set
    bool ws_owned = false;            // between golap_join_partitioned_begin and _finish / _abort: ws_begin refuses other callers
    std::vector<bool> attr_done;      // per kernel family: hipFuncSetAttribute(MaxDynamicSharedMemorySize) applied on THIS context's device
};

namespace golap {

// RAII kernel timer: records start now, stop at scope exit (when profiling is on).
struct KernelTimer {
    golap_ctx* ctx;
    hipStream_t st;
    PendingEvent ev;
    bool on;
    // cond_bytes: credited at resolve time if the call's gate word turned out (credit_when_set ? non-zero : zero)
    KernelTimer(golap_ctx* c, const char* name, hipStream_t s, uint64_t bytes = 0, uint64_t cond_bytes = 0, bool credit_when_set = true)
        : ctx(c), st(s), on(c->prof.enabled && !c->prof.mute) {
        if (on) {   // timing events have no meaning inside a stream capture (and must not become graph nodes)
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cs) != hipSuccess) (void)hipGetLastError();
            if (cs != hipStreamCaptureStatusNone) on = false;
        }
        if (!on) return;
        ev.stat = c->prof.stat_index(name);
        c->prof.stats[ev.stat].launches += 1;
        c->prof.stats[ev.stat].bytes += bytes;
        if (cond_bytes && c->prof.cur_gate_host) {
            ev.cond_bytes = cond_bytes;
            ev.gate_host = c->prof.cur_gate_host;
            ev.credit_when_set = credit_when_set;
        } else if (cond_bytes && !credit_when_set) {
            c->prof.stats[ev.stat].bytes += cond_bytes;   // no read-back in this call: the optimistic pipeline is credited as launched
        }
        ev.start = c->prof.get_event();
        ev.stop = c->prof.get_event();
        (void)hipEventRecord(ev.start, st);
    }
    ~KernelTimer() {
        if (on) {
            (void)hipEventRecord(ev.stop, st);
            ctx->prof.pending.push_back(ev);
        }
        static const bool dbg = getenv("GOLAP_DEBUG_SYNC") != nullptr;   // debugging aid: serialise + name every phase
        if (dbg) {
            hipError_t e = hipStreamSynchronize(st);
            fprintf(stderr, "[golap] %s finished: %s\n", on ? ctx->prof.stats[ev.stat].name : "kernel", hipGetErrorString(e));
            fflush(stderr);
        }
    }
};

struct KnobScope {
    const Knobs* prev;
    const golap_engine_config* prev_cfg;
    explicit KnobScope(golap_ctx* c) : prev(g_knobs), prev_cfg(g_scope_cfg) { g_knobs = c ? &c->knobs : nullptr; g_scope_cfg = c ? &c->cfg : nullptr; }
    ~KnobScope() { g_knobs = prev; g_scope_cfg = prev_cfg; }
};

// head-room of the provisioned (no-histogram) layouts: standard deviations + pad rows (golap_engine_config::provision_sigmas;
// the per-pass test knobs GOLAP_RESERVE_SIGMAS / GOLAP_SLOT_SIGMAS override it)
constexpr uint32_t PROVISION_MAX_SIGMAS = 12;
inline void provision_of(const char* knob_name, uint32_t& sigmas, uint32_t& pad) {
    sigmas = 6;
    pad = 48;
    if (g_scope_cfg && g_scope_cfg->provision_sigmas) { sigmas = g_scope_cfg->provision_sigmas == GOLAP_PROVISION_NONE ? 0u : g_scope_cfg->provision_sigmas; pad = sigmas ? 48 : 0; }
    if (const char* v = knob(knob_name)) { sigmas = (uint32_t)atoi(v); pad = sigmas ? 48 : 0; }
    if (sigmas > PROVISION_MAX_SIGMAS) sigmas = PROVISION_MAX_SIGMAS;   // (6 sigmas already mean < 1e-8 per slot; more only buys buffer)
}

// NULL is the HIP null (legacy default) stream, exactly as in every HIP API -- it is also what
// torch.cuda.current_stream().cuda_stream returns for torch's default stream.
inline hipStream_t pick_stream(golap_ctx* ctx, void* stream) { (void)ctx; return (hipStream_t)stream; }

// workspace helpers (call with ctx->mu held)
int32_t ws_begin(golap_ctx* ctx, uint64_t bytes, hipStream_t st);
uint64_t ws_take(golap_ctx* ctx, uint64_t bytes);   // 256-B aligned bump; returns 0 when exhausted
void ws_end(golap_ctx* ctx, hipStream_t st);
int32_t aux_begin(golap_ctx* ctx, uint64_t bytes, hipStream_t st);   // same three calls on the second slab (ctx->aux)
uint64_t aux_take(golap_ctx* ctx, uint64_t bytes);
void aux_end(golap_ctx* ctx, hipStream_t st);

// in-place exclusive scan of n u32 values; block_sums needs flat_scan_scratch_words(n) u32 (= ceil(n/4096)+1 while n <= 16.7M) (golap_join.hip)
int32_t flat_scan(golap_ctx* ctx, hipStream_t st, uint32_t* a, uint32_t n, uint32_t* block_sums, const uint32_t* run_if = nullptr);
uint64_t flat_scan_scratch_words(uint64_t n);
constexpr uint32_t FLAT_SCAN_BLOCK = 4096;

// Zero-fill by a kernel, not hipMemsetAsync: a captured memset node was observed to write garbage on its first
// graph launch when a pageable D2H copy ran between capture and launch (ROCm 7.0 runtime); kernel nodes are immune.
// `bytes` is rounded up to a multiple of 4 (callers own at least that much).
int32_t zero_async(golap_ctx* ctx, hipStream_t st, void* ptr, uint64_t bytes);

static inline uint64_t round_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }
static inline uint64_t ceil_div(uint64_t x, uint64_t a) { return (x + a - 1) / a; }

}  // namespace golap
