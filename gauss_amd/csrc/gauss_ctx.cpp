// libgauss_hip.so -- contexts: streams and the per-device hardware-queue registry, block caches, destroy hooks.
#include "gauss_job.h"

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

bool trace_on(const char* what)
{
    const char* e = getenv("GAUSS_TRACE");
    if (!e || !*e) return false;
    return strstr(e, "all") != nullptr || strstr(e, what) != nullptr;
}

// Lifetime rule of the C ABI (include/gauss_hip.h): a context may be destroyed while jobs and row stores made on it
// are still alive.  gauss_hip_destroy waits for their queued work, releases everything they hold on the device and
// leaves the job handles as empty shells ("orphans": ctx == nullptr) that only gauss_job_destroy accepts; destroy hooks
// let the host layer drop what it cached per context.  An Rcpp driver whose objects unwind in any order
// (Rcpp::stop between create and destroy) therefore never touches freed memory.
static std::mutex g_hook_mu;
static std::vector<std::pair<void (*)(gauss_ctx*, uint64_t, void*), void*>> g_destroy_hooks;
static std::atomic<uint64_t> g_next_ctx_id{1};

// ------------------------------------------------------------------------------------------
// Streams and hardware queues (DESIGN.md section 5 "the queue rule"; docs/HISTORY.md section 4 "Streams and hardware queues"; profiles/r05_queue_map.txt is the runtime's own
// log of the mapping on MI355X, ROCm 7.2): a stream created with a priority takes a hardware queue of its own while its
// priority class holds fewer than GPU_MAX_HW_QUEUES (default 4); from then on "Selected queue refCount": the least used
// queue of the class is shared, and kernels of the streams that share it run in submission order.  One context makes three
// high-priority streams at once (main, chain, upload), a fourth on its first streamed window (aux) and one low-priority
// stream: alone on its device each owns a queue.  A second context's streams start sharing -- in the logged run its chain
// stream landed on ITS OWN main stream's queue -- and a kernel that spins for another stream's progress may then sit in
// front of the kernels it is waiting for.  So: the library counts its priority streams per device, and a run is queued in
// the merged form (spinning consumers, k_gram.hip: wait_count_kernel) only while both classes fit their pools.
// ------------------------------------------------------------------------------------------
namespace {
struct QueueRegistry {
    std::shared_mutex mu;             // shared: a run is being queued; exclusive: the stream population of a device changes
    std::mutex count_mu;              // guards the counts and the context list
    int hi[64] = {0}, lo[64] = {0};
    std::vector<gauss_ctx*> ctxs;
};
QueueRegistry& registry() { static QueueRegistry* r = new QueueRegistry(); return *r; }

// hardware queues per priority class the runtime will make; 0: unknown mapping (never assume exclusive queues)
int hw_queue_cap()
{
    if (const char* d = getenv("DEBUG_HIP_DYNAMIC_QUEUES")) { if (atoi(d) != 0) return 0; }
    if (const char* e = getenv("GPU_MAX_HW_QUEUES")) { const int v = atoi(e); return v > 0 ? v : 0; }
    return 4;
}
}  // namespace

std::shared_mutex& queue_registry_mutex() { return registry().mu; }

bool queues_exclusive(int device)
{
    QueueRegistry& r = registry();
    const int cap = hw_queue_cap();
    std::lock_guard<std::mutex> lock(r.count_mu);
    const int d = device & 63;
    return r.hi[d] <= cap && r.lo[d] <= cap;
}

// Do kernels of `waiter_q` and `setter_q` run side by side?  A one-wave kernel on waiter_q waits (a few milliseconds at most) for
// a counter that a kernel queued AFTERWARDS on setter_q advances.  On different hardware queues the second kernel runs at once and the
// first returns; on a shared queue the second sits behind the first, which gives up at its bound.  This is the situation a merged
// Gram launch must never be in (its waiting kernels are queued on the chain and low-priority queues, what they wait for on the main
// queue), seen directly -- whatever the runtime's mapping rule, and whoever else holds priority streams in the process.
static bool queues_side_by_side(gauss_ctx* c, hipStream_t waiter_q, hipStream_t setter_q)
{
    void* d = nullptr;
    if (hipMalloc(&d, 256) != hipSuccess) { (void)hipGetLastError(); return false; }
    unsigned long long* cnt = (unsigned long long*)d;
    int* st = (int*)((char*)d + 128);
    bool ok = hipMemset(d, 0, 256) == hipSuccess;
    if (ok) {
        // first launches load the code object and make the runtime set up both queues: not part of what is timed
        launch_count_up(cnt, waiter_q);
        launch_count_up(cnt, setter_q);
        ok = hipStreamSynchronize(waiter_q) == hipSuccess && hipStreamSynchronize(setter_q) == hipSuccess;
    }
    if (ok) {
        launch_wait_count_for(cnt, 3, st, waiter_q, 5000.0);       // counter is 2 now; 5 ms bound
        launch_count_up(cnt, setter_q);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(waiter_q) == hipSuccess && hipStreamSynchronize(setter_q) == hipSuccess;
    }
    int gave_up = 1;
    if (ok) ok = hipMemcpy(&gave_up, st, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    (void)hipGetLastError();
    return ok && gave_up == 0;
}

int ctx_stream_create(gauss_ctx* c, hipStream_t* out, StreamClass cls)
{
    if (cls == STREAM_NORMAL) { HIPCHK(hipStreamCreateWithFlags(out, hipStreamNonBlocking)); return GAUSS_OK; }
    QueueRegistry& r = registry();
    const int d = c->device & 63;
    bool overflow;
    {
        // the slot is taken first: from here on a run that looks at the count (job_queue_run) sees the newcomer
        std::lock_guard<std::mutex> lock(r.count_mu);
        int& n = cls == STREAM_HIGH ? r.hi[d] : r.lo[d];
        overflow = n == hw_queue_cap();
        n++;
        (cls == STREAM_HIGH ? c->n_hi : c->n_lo)++;
    }
    // The stream that overflows its class's pool will share a hardware queue.  Runs queued from now on see the count and
    // take the two-launch form; the ones already queued in the merged form are let finish first (no run is being queued
    // meanwhile: job_queue_run holds the registry's mutex shared), so that no spinning kernel of theirs is ever joined
    // in its hardware queue by a newcomer's work.
    if (overflow) {
        std::unique_lock<std::shared_mutex> excl(r.mu);
        std::vector<gauss_ctx*> others;
        { std::lock_guard<std::mutex> lock(r.count_mu); others = r.ctxs; }
        for (gauss_ctx* o : others)
            if (o->device == c->device)
                for (hipStream_t q : {o->chain, o->side}) if (q) (void)hipStreamSynchronize(q);
    }
    int lo = 0, hi = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(out, hipStreamNonBlocking, cls == STREAM_HIGH ? hi : lo);
    if (e != hipSuccess) {
        std::lock_guard<std::mutex> lock(r.count_mu);
        (cls == STREAM_HIGH ? r.hi[d] : r.lo[d])--;
        (cls == STREAM_HIGH ? c->n_hi : c->n_lo)--;
        return fail(GAUSS_E_DEVICE, "hipStreamCreateWithPriority failed: %s", hipGetErrorString(e));
    }
    return GAUSS_OK;
}

void ctx_stream_destroy(gauss_ctx* c, hipStream_t* s, StreamClass cls)
{
    if (!*s) return;
    (void)hipStreamSynchronize(*s);
    (void)hipStreamDestroy(*s);
    *s = nullptr;
    if (cls == STREAM_NORMAL) return;
    QueueRegistry& r = registry();
    std::lock_guard<std::mutex> lock(r.count_mu);
    const int d = c->device & 63;
    if (cls == STREAM_HIGH) { r.hi[d]--; c->n_hi--; } else { r.lo[d]--; c->n_lo--; }
}

void ctx_join_prepin(gauss_ctx* c)
{
    std::lock_guard<std::mutex> lock(c->prepin_mu);
    if (c->prepin.joinable()) c->prepin.join();
}

// Freed job workspaces (device) and staging blocks (pinned host) are kept per context and handed to the next job
// that fits (BlockCache, gauss_job.h).
static void ctx_flush_dev_cache_locked(gauss_ctx* c)
{
    for (auto& kv : c->dev_cache.free_blocks) { (void)hipFree(kv.second); c->block_size.erase(kv.second); }
    c->dev_cache.free_blocks.clear(); c->dev_cache.held = 0;
    (void)hipGetLastError();
}
// hipMalloc that gives the context's cached workspaces back to the device before it reports failure (a row store of
// tens of GB, a scratch buffer or another context on the same device may need the room the cache is sitting on)
hipError_t ctx_malloc_retry(gauss_ctx* c, void** out, size_t bytes)
{
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lock(c->mu);
        if (c->dev_cache.free_blocks.empty()) return e;
        ctx_flush_dev_cache_locked(c);
    }
    return hipMalloc(out, bytes);
}

hipError_t ctx_dev_alloc(gauss_ctx* c, size_t bytes, void** out)
{
    std::lock_guard<std::mutex> lock(c->mu);
    if (void* p = c->dev_cache.take(bytes)) { *out = p; return hipSuccess; }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess && !c->dev_cache.free_blocks.empty()) {       // make room and retry once
        ctx_flush_dev_cache_locked(c);
        e = hipMalloc(out, bytes);
    }
    if (e == hipSuccess) c->block_size[*out] = bytes;
    return e;
}
void ctx_dev_release(gauss_ctx* c, void* p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lock(c->mu);
    auto it = c->block_size.find(p);
    const size_t bytes = it == c->block_size.end() ? 0 : it->second;
    if (bytes && c->dev_cache.held + bytes <= c->dev_cache_limit) { c->dev_cache.free_blocks.emplace(bytes, p); c->dev_cache.held += bytes; return; }
    if (it != c->block_size.end()) c->block_size.erase(it);
    (void)hipFree(p);
}
hipError_t ctx_pin_alloc(gauss_ctx* c, size_t bytes, void** out)
{
    std::lock_guard<std::mutex> lock(c->mu);
    if (void* p = c->pin_cache.take(bytes)) { *out = p; return hipSuccess; }
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e == hipSuccess) c->block_size[*out] = bytes;
    return e;
}
void ctx_pin_release(gauss_ctx* c, void* p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lock(c->mu);
    auto it = c->block_size.find(p);
    const size_t bytes = it == c->block_size.end() ? 0 : it->second;
    if (bytes && c->pin_cache.held + bytes <= PIN_CACHE_LIMIT) { c->pin_cache.free_blocks.emplace(bytes, p); c->pin_cache.held += bytes; return; }
    if (it != c->block_size.end()) c->block_size.erase(it);
    (void)hipHostFree(p);
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char* gauss_last_error(void) { return g_err.c_str(); }
const char* gauss_hip_version(void) { return "gauss_hip 0.2 (gfx950)"; }

int gauss_hip_init(int device, gauss_ctx** out_ctx)
{
    if (!out_ctx) return fail(GAUSS_E_INVALID, "out_ctx is NULL");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(GAUSS_E_INVALID, "device %d out of range (have %d)", device, n);
    HIPCHK(hipSetDevice(device));
    std::unique_ptr<gauss_ctx> guard(new gauss_ctx());
    gauss_ctx* c = guard.get();
    c->device = device;
    c->id = g_next_ctx_id.fetch_add(1);
    {
        // freed job workspaces are kept for reuse up to a third of the device's memory (96 GB of an MI355X's 288 GB)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { total_b = (size_t)288 << 30; (void)hipGetLastError(); }
        c->dev_cache_limit = total_b / 3;
    }
    const char* e = getenv("GAUSS_GRAM_DTYPE");
    c->gram_i8 = (e && (strcmp(e, "i8") == 0 || strcmp(e, "int8") == 0)) ? 1 : 0;
    const char* one = getenv("GAUSS_SIDE_STREAM");
    c->main_cls = (one && atoi(one) == 0) ? STREAM_NORMAL : STREAM_HIGH;
    int rc = ctx_stream_create(c, &c->stream, c->main_cls);
    if (!rc && c->main_cls == STREAM_HIGH) {
        // main: high priority (a workgroup of the chain's next launch must win the CU an epilogue workgroup frees when the chain
        // runs on it); side: low priority (B21's epilogue tiles beside the chain; the early tiles of a merged launch, which the
        // hardware hands out only when the Gram grid has no workgroup left to dispatch); chain: the factorisation chain beside
        // the Gram kernel (job_queue_run, k_solve_lite.hip).  GAUSS_SIDE_STREAM=0: every kernel of a job on one queue.
        rc = ctx_stream_create(c, &c->side, STREAM_LOW);
        if (!rc) rc = ctx_stream_create(c, &c->chain, STREAM_HIGH);
    }
    // What a session's FIRST row-store upload would otherwise pay in front of its first byte (measured on MI355X, round 4): the
    // queue of its own that piecewise / background uploads travel on (creating a stream: ~15 ms the first time) and the two
    // pinned staging buffers (hipHostMalloc of 2 x 32 MB: 4-14 ms).  The queue is made here; the buffers are made in the
    // background and parked in the context's pinned-block cache.
    if (!rc) rc = ctx_stream_create(c, &c->upload, STREAM_HIGH);       // (at least the main queue's priority: see gauss_store_fill)
    if (rc) {
        ctx_stream_destroy(c, &c->upload, STREAM_HIGH);
        ctx_stream_destroy(c, &c->chain, STREAM_HIGH);
        ctx_stream_destroy(c, &c->side, STREAM_LOW);
        ctx_stream_destroy(c, &c->stream, c->main_cls);
        return rc;
    }
    if (c->chain && c->side) c->queues_probed_distinct = queues_side_by_side(c, c->chain, c->stream) && queues_side_by_side(c, c->side, c->stream);
    c->prepin = std::thread([c]() {
        (void)hipSetDevice(c->device);
        {
            // the upload queue's first KERNEL (job_build zeroes new workspaces there; background fills copy by kernel) makes
            // the runtime set up its compute queue: 100 ms when it happened in the middle of a chromosome's first call
            // (one job creation of 103 ms, round 4) -- done here, off everybody's path
            void* d = nullptr;
            if (hipMalloc(&d, 256) == hipSuccess) {
                (void)hipMemsetAsync(d, 0, 256, c->upload);
                (void)hipStreamSynchronize(c->upload);
                (void)hipFree(d);
            } else (void)hipGetLastError();
        }
        for (int b = 0; b < 2; b++) {
            void* p = nullptr;                             // (outside c->mu: the main thread may be building its first job meanwhile)
            if (hipHostMalloc(&p, UPLOAD_CHUNK, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); break; }
            { std::lock_guard<std::mutex> lock(c->mu); c->block_size[p] = UPLOAD_CHUNK; }
            ctx_pin_release(c, p);                         // parked in the pinned-block cache
        }
    });
    { QueueRegistry& r = registry(); std::lock_guard<std::mutex> lock(r.count_mu); r.ctxs.push_back(c); }
    *out_ctx = guard.release();
    return GAUSS_OK;
}

int gauss_hip_device_count(int* out_n)
{
    if (!out_n) return fail(GAUSS_E_INVALID, "out_n is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice) { n = 0; e = hipSuccess; (void)hipGetLastError(); }
    if (e != hipSuccess) return fail(GAUSS_E_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    *out_n = n;
    return GAUSS_OK;
}

int gauss_hip_device_of(const gauss_ctx* ctx) { return ctx ? ctx->device : fail(GAUSS_E_INVALID, "ctx is NULL"); }


void gauss_hip_destroy(gauss_ctx* ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    ctx_join_prepin(ctx);
    // 1. whoever cached something per context (the host layer's resident panels) lets go of it
    std::vector<std::pair<void (*)(gauss_ctx*, uint64_t, void*), void*>> hooks;
    { std::lock_guard<std::mutex> lock(g_hook_mu); hooks = g_destroy_hooks; }
    for (auto& h : hooks) h.first(ctx, ctx->id, h.second);
    // 2. jobs that outlive the context: wait for their work, release what they hold, leave empty shells behind
    std::vector<gauss_job*> live;
    { std::lock_guard<std::mutex> lock(ctx->mu); live.assign(ctx->jobs.begin(), ctx->jobs.end()); }
    for (gauss_job* j : live) job_release(j);
    hipStreamSynchronize(ctx->stream);
    {
        // (another context's gauss_hip_init may be waiting for this one's spinning kernels: it reads the list under the mutex)
        QueueRegistry& r = registry();
        std::unique_lock<std::shared_mutex> excl(r.mu);
        { std::lock_guard<std::mutex> lock(r.count_mu); r.ctxs.erase(std::remove(r.ctxs.begin(), r.ctxs.end(), ctx), r.ctxs.end()); }
        ctx_stream_destroy(ctx, &ctx->side, STREAM_LOW);
        delete ctx->worker;
        ctx_stream_destroy(ctx, &ctx->copy, STREAM_NORMAL);
        ctx_stream_destroy(ctx, &ctx->aux, STREAM_HIGH);
        ctx_stream_destroy(ctx, &ctx->chain, STREAM_HIGH);
    }
    for (hipEvent_t e : ctx->ev_pool) hipEventDestroy(e);
    if (ctx->landing) (void)hipFree(ctx->landing);
    // 3. row stores nobody freed (uploads still running are finished first)
    for (auto& kv : ctx->uploads) kv.second->finish();
    ctx->uploads.clear();
    ctx_stream_destroy(ctx, &ctx->upload, STREAM_HIGH);
    for (auto& kv : ctx->stores) (void)hipFree(const_cast<void*>(kv.first));
    ctx->stores.clear();
    for (auto& kv : ctx->dev_cache.free_blocks) (void)hipFree(kv.second);
    for (auto& kv : ctx->pin_cache.free_blocks) (void)hipHostFree(kv.second);
    ctx_stream_destroy(ctx, &ctx->stream, ctx->main_cls);
    delete ctx;
}

uint64_t gauss_hip_context_id(const gauss_ctx* ctx) { return ctx ? ctx->id : 0; }

int gauss_hip_add_destroy_hook(void (*fn)(gauss_ctx*, uint64_t, void*), void* user)
{
    if (!fn) return fail(GAUSS_E_INVALID, "hook is NULL");
    std::lock_guard<std::mutex> lock(g_hook_mu);
    for (auto& h : g_destroy_hooks) if (h.first == fn && h.second == user) return GAUSS_OK;
    g_destroy_hooks.emplace_back(fn, user);
    return GAUSS_OK;
}

int gauss_hip_trim_cache(gauss_ctx* ctx, int64_t* out_bytes_freed)
{
    if (!ctx) return fail(GAUSS_E_INVALID, "ctx is NULL");
    HIPCHK(hipSetDevice(ctx->device));
    // blocks in the cache may still be read by work queued on the streams (a retired job's last launches)
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->side) HIPCHK(hipStreamSynchronize(ctx->side));
    if (ctx->chain) HIPCHK(hipStreamSynchronize(ctx->chain));
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (out_bytes_freed) *out_bytes_freed = (int64_t)ctx->dev_cache.held;
    ctx_flush_dev_cache_locked(ctx);
    return GAUSS_OK;
}

int gauss_hip_counters(gauss_ctx* ctx, int64_t* out4)
{
    if (!ctx || !out4) return fail(GAUSS_E_INVALID, "bad arguments to gauss_hip_counters");
    out4[0] = ctx->n_runs_merged.load();
    out4[1] = ctx->n_runs_demoted.load();
    out4[2] = ctx->n_merged_giveups.load();
    out4[3] = ctx->n_rerun_failed.load();
    return GAUSS_OK;
}

int gauss_hip_queues(gauss_ctx* ctx, int32_t* out4)
{
    if (!ctx || !out4) return fail(GAUSS_E_INVALID, "bad arguments to gauss_hip_queues");
    QueueRegistry& r = registry();
    std::lock_guard<std::mutex> lock(r.count_mu);
    out4[0] = r.hi[ctx->device & 63];
    out4[1] = r.lo[ctx->device & 63];
    out4[2] = hw_queue_cap();
    out4[3] = ctx->queues_probed_distinct ? 1 : 0;
    return GAUSS_OK;
}

int gauss_pinned_alloc(gauss_ctx* ctx, int64_t bytes, void** out_host_ptr)
{
    if (!ctx || bytes <= 0 || !out_host_ptr) return fail(GAUSS_E_INVALID, "bad arguments to gauss_pinned_alloc");
    HIPCHK(hipSetDevice(ctx->device));
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(GAUSS_E_NOMEM, "hipHostMalloc(%lld bytes) failed: %s", (long long)bytes, hipGetErrorString(e));
    *out_host_ptr = p;
    return GAUSS_OK;
}

int gauss_pinned_free(gauss_ctx* ctx, void* host_ptr)
{
    if (!ctx) return fail(GAUSS_E_INVALID, "ctx is NULL");
    HIPCHK(hipSetDevice(ctx->device));
    if (host_ptr) HIPCHK(hipHostFree(host_ptr));
    return GAUSS_OK;
}

int gauss_hip_set_gram_dtype(gauss_ctx* ctx, int dtype)
{
    if (!ctx || (dtype != GAUSS_GRAM_F32 && dtype != GAUSS_GRAM_I8)) return fail(GAUSS_E_INVALID, "bad gram dtype %d", dtype);
    ctx->gram_i8 = (dtype == GAUSS_GRAM_I8);
    return GAUSS_OK;
}

}  // extern "C"
