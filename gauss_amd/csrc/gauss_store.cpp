// libgauss_hip.so -- resident row stores: genotype rows from host memory or a file section into HBM.
#include "gauss_job.h"

extern "C" {

// Host rows -> HBM.  Large stores (a packed panel's genotype section is ~0.8 GB per chromosome) go through two
// pinned staging buffers: host threads copy chunk k+1 out of the caller's (pageable, typically mmap'd) memory while
// chunk k travels by hipMemcpyAsync -- the staged copy the runtime would do by itself for pageable memory, made
// parallel and overlapped with the DMA.  Small stores take one plain copy.
// Where the rows of an upload come from: host memory, or a section of a file that is read with pread straight into the
// pinned staging buffers.  A memcpy out of a fresh mmap takes a page fault per 4 KB on the process's address space --
// 206 000 of them for a chromosome -- and every other thread of the process that faults or allocates (the data layer
// running beside the upload) queues behind them; pread touches no page tables.
struct RowSource2 {
    const uint8_t* ptr = nullptr;
    int fd = -1;
    int64_t file_off = 0;
    bool copy(uint8_t* dst, size_t off, size_t len) const
    {
        if (ptr) { memcpy(dst, ptr + off, len); return true; }
        while (len > 0) {
            const ssize_t n = pread(fd, dst, len, (off_t)(file_off + (int64_t)off));
            if (n <= 0) return false;
            dst += n; off += (size_t)n; len -= (size_t)n;
        }
        return true;
    }
};

// by_kernel: the staged chunks cross PCIe through launch_h2d_copy (a small-footprint kernel that reads the pinned staging
// buffer itself) instead of hipMemcpyAsync, which stalls behind any kernel that holds every CU (k_misc.hip): the form for
// uploads that are meant to run BESIDE compute.
static int upload_rows(gauss_ctx* ctx, void* d, const RowSource2& src, size_t bytes, hipStream_t stream = nullptr,
                       const std::function<void(size_t)>& chunk_queued = nullptr, bool by_kernel = false)
{
    const size_t CH = UPLOAD_CHUNK;
    if (!stream) stream = ctx->stream;
    ctx_join_prepin(ctx);                                  // the staging buffers made at init are in the pinned cache (or about to be)

    if (bytes < 2 * CH && !chunk_queued && src.ptr && stream == ctx->stream) { HIPCHK(hipMemcpy(d, src.ptr, bytes, hipMemcpyHostToDevice)); return GAUSS_OK; }
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    int rc = GAUSS_OK;
    auto cleanup = [&]() {
        for (int b = 0; b < 2; b++) { if (ev[b]) hipEventDestroy(ev[b]); ctx_pin_release(ctx, pin[b]); }
    };
    const bool trace = trace_on("upload");
    const auto t_up0 = std::chrono::steady_clock::now();
    for (int b = 0; b < 2; b++) {
        if (ctx_pin_alloc(ctx, CH, &pin[b]) != hipSuccess || hipEventCreateWithFlags(&ev[b], hipEventDisableTiming) != hipSuccess) {
            cleanup();
            return fail(GAUSS_E_NOMEM, "pinned staging buffers for the row store upload could not be allocated");
        }
    }
    const auto t_up1 = std::chrono::steady_clock::now();
    double t_stage = 0, t_wait = 0;
    const unsigned hw = std::thread::hardware_concurrency();
    // a background upload (chunk_queued set) shares the host with the data layer it runs beside: fewer copy threads
    // one process per GPU (torchrun exports LOCAL_WORLD_SIZE): the ranks of a node share its cores
    const unsigned ranks = (unsigned)std::max(1, env_int("LOCAL_WORLD_SIZE", 1));
    const int nt = (int)std::max(1u, std::min(chunk_queued ? 4u : 8u, (hw ? hw / 2 : 2u) / ranks));
    std::atomic<bool> read_ok{true};
    size_t k = 0;
    for (size_t off = 0; off < bytes && rc == GAUSS_OK; off += CH, k++) {
        const int b = (int)(k & 1);
        const size_t len = std::min(CH, bytes - off);
        const auto tw0 = std::chrono::steady_clock::now();
        if (k >= 2 && hipEventSynchronize(ev[b]) != hipSuccess) { rc = fail(GAUSS_E_DEVICE, "row store upload: event wait failed"); break; }
        const auto tw1 = std::chrono::steady_clock::now();
        t_wait += std::chrono::duration<double, std::milli>(tw1 - tw0).count();
        std::vector<std::thread> th;
        const size_t per = (len + nt - 1) / nt;
        for (int t = 1; t < nt; t++) {
            const size_t o = per * t;
            if (o < len) th.emplace_back([&, o]() { if (!src.copy((uint8_t*)pin[b] + o, off + o, std::min(per, len - o))) read_ok = false; });
        }
        if (!src.copy((uint8_t*)pin[b], off, std::min(per, len))) read_ok = false;
        for (std::thread& x : th) x.join();
        t_stage += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw1).count();
        if (!read_ok) { rc = fail(GAUSS_E_INVALID, "row store upload: reading the source failed (short file?)"); break; }
        hipError_t ce = hipSuccess;
        const size_t body = (by_kernel && ((uintptr_t)((uint8_t*)d + off) & 15) == 0) ? len / 16 * 16 : 0;
        if (body) { launch_h2d_copy((uint8_t*)d + off, pin[b], body, stream); ce = hipGetLastError(); }
        if (ce == hipSuccess && body < len) ce = hipMemcpyAsync((uint8_t*)d + off + body, (uint8_t*)pin[b] + body, len - body, hipMemcpyHostToDevice, stream);
        if (ce != hipSuccess ||
            hipEventRecord(ev[b], stream) != hipSuccess)
            rc = fail(GAUSS_E_DEVICE, "row store upload: hipMemcpyAsync failed");
        else if (chunk_queued) chunk_queued(off + len);
    }
    if (hipStreamSynchronize(stream) != hipSuccess && rc == GAUSS_OK) rc = fail(GAUSS_E_DEVICE, "row store upload failed");
    if (trace)
        fprintf(stderr, "[upload] %.1f MB: pinned buffers %.2f ms, staging %.2f ms (%d threads), waiting for the copies %.2f ms, total %.2f ms\n", bytes / 1e6,
                std::chrono::duration<double, std::milli>(t_up1 - t_up0).count(), t_stage, nt, t_wait,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up0).count());
    cleanup();
    return rc;
}

int gauss_store_upload(gauss_ctx* ctx, const void* host_rows, int64_t bytes, void** out_device_ptr)
{
    if (!ctx || !host_rows || bytes <= 0 || !out_device_ptr) return fail(GAUSS_E_INVALID, "bad arguments to gauss_store_upload");
    HIPCHK(hipSetDevice(ctx->device));
    void* d = nullptr;
    hipError_t e = ctx_malloc_retry(ctx, &d, (size_t)bytes + 64);      // slack: a row's last dword load may end on the last byte
    if (e != hipSuccess) return fail(GAUSS_E_NOMEM, "hipMalloc(%lld bytes row store) failed: %s", (long long)bytes, hipGetErrorString(e));
    RowSource2 src;
    src.ptr = (const uint8_t*)host_rows;
    const int rc = upload_rows(ctx, d, src, (size_t)bytes);
    if (rc) { hipFree(d); return rc; }
    { std::lock_guard<std::mutex> lock(ctx->mu); ctx->stores[d] = (size_t)bytes; }
    *out_device_ptr = d;
    return GAUSS_OK;
}

// The rows are a section of an open FILE (a packed panel's genotype section), read with pread straight into the pinned
// staging buffers: a memcpy out of a fresh mapping of the file takes a page fault per 4 KB on the process's address space
// (206 000 for a chromosome) -- pread touches no page table of the caller.
int gauss_store_upload_fd(gauss_ctx* ctx, int fd, int64_t file_offset, int64_t bytes, void** out_device_ptr)
{
    if (!ctx || fd < 0 || file_offset < 0 || bytes <= 0 || !out_device_ptr) return fail(GAUSS_E_INVALID, "bad arguments to gauss_store_upload_fd");
    HIPCHK(hipSetDevice(ctx->device));
    void* d = nullptr;
    hipError_t e = ctx_malloc_retry(ctx, &d, (size_t)bytes + 64);
    if (e != hipSuccess) return fail(GAUSS_E_NOMEM, "hipMalloc(%lld bytes row store) failed: %s", (long long)bytes, hipGetErrorString(e));
    RowSource2 src;
    src.fd = fd; src.file_off = file_offset;
    const int rc = upload_rows(ctx, d, src, (size_t)bytes);
    if (rc) { hipFree(d); return rc; }
    { std::lock_guard<std::mutex> lock(ctx->mu); ctx->stores[d] = (size_t)bytes; }
    *out_device_ptr = d;
    return GAUSS_OK;
}

int gauss_store_alloc(gauss_ctx* ctx, int64_t bytes, void** out_device_ptr)
{
    if (!ctx || bytes <= 0 || !out_device_ptr) return fail(GAUSS_E_INVALID, "bad arguments to gauss_store_alloc");
    HIPCHK(hipSetDevice(ctx->device));
    void* d = nullptr;
    hipError_t e = ctx_malloc_retry(ctx, &d, (size_t)bytes + 64);      // slack: a row's last dword load may end on the last byte
    if (e != hipSuccess) return fail(GAUSS_E_NOMEM, "hipMalloc(%lld bytes row store) failed: %s", (long long)bytes, hipGetErrorString(e));
    { std::lock_guard<std::mutex> lock(ctx->mu); ctx->stores[d] = (size_t)bytes; }
    *out_device_ptr = d;
    return GAUSS_OK;
}

// Bytes [offset, offset + len) of a store made by gauss_store_alloc, from the same offsets of host_rows; returns when they have
// landed.  The copy travels on the context's upload queue, so whatever the main queue is computing keeps running.
static int store_fill(gauss_ctx* ctx, void* device_ptr, const RowSource2& src0, int64_t offset, int64_t len);

int gauss_store_fill(gauss_ctx* ctx, void* device_ptr, const void* host_rows, int64_t offset, int64_t len)
{
    if (!ctx || !device_ptr || !host_rows || offset < 0 || len < 0) return fail(GAUSS_E_INVALID, "bad arguments to gauss_store_fill");
    RowSource2 src;
    src.ptr = (const uint8_t*)host_rows;
    return store_fill(ctx, device_ptr, src, offset, len);
}

// the same piece from a file: bytes [file_offset + offset, + len) of fd (see gauss_store_upload_fd)
int gauss_store_fill_fd(gauss_ctx* ctx, void* device_ptr, int fd, int64_t file_offset, int64_t offset, int64_t len)
{
    if (!ctx || !device_ptr || fd < 0 || file_offset < 0 || offset < 0 || len < 0) return fail(GAUSS_E_INVALID, "bad arguments to gauss_store_fill_fd");
    RowSource2 src;
    src.fd = fd; src.file_off = file_offset;
    return store_fill(ctx, device_ptr, src, offset, len);
}

static int store_fill(gauss_ctx* ctx, void* device_ptr, const RowSource2& src0, int64_t offset, int64_t len)
{
    if (len == 0) return GAUSS_OK;
    HIPCHK(hipSetDevice(ctx->device));
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        auto it = ctx->stores.find(device_ptr);
        if (it == ctx->stores.end()) return fail(GAUSS_E_INVALID, "gauss_store_fill: not a row store of this context");
        if ((size_t)(offset + len) > it->second) return fail(GAUSS_E_INVALID, "gauss_store_fill: bytes [%lld, %lld) lie outside the store (%zu bytes)",
                                                             (long long)offset, (long long)(offset + len), it->second);
    }
    RowSource2 src = src0;
    if (src.ptr) src.ptr += offset; else src.file_off += offset;
    // by kernel: a piece is meant to travel beside whatever the main queue computes, and hipMemcpyAsync stalls behind a kernel
    // that holds every CU (k_misc.hip: h2d_copy_kernel); the upload queue has the main queue's priority -- a lower one is not
    // dispatched while the Gram grid has workgroups left
    return upload_rows(ctx, (uint8_t*)device_ptr + offset, src, (size_t)len, ctx->upload, nullptr, true);
}

static int store_upload_async(gauss_ctx* ctx, const RowSource2& src, int64_t bytes, void** out_device_ptr);

int gauss_store_upload_async(gauss_ctx* ctx, const void* host_rows, int64_t bytes, void** out_device_ptr)
{
    if (!ctx || !host_rows || bytes <= 0 || !out_device_ptr) return fail(GAUSS_E_INVALID, "bad arguments to gauss_store_upload_async");
    RowSource2 src;
    src.ptr = (const uint8_t*)host_rows;
    return store_upload_async(ctx, src, bytes, out_device_ptr);
}

int gauss_store_upload_fd_async(gauss_ctx* ctx, int fd, int64_t file_offset, int64_t bytes, void** out_device_ptr)
{
    if (!ctx || fd < 0 || file_offset < 0 || bytes <= 0 || !out_device_ptr) return fail(GAUSS_E_INVALID, "bad arguments to gauss_store_upload_fd_async");
    RowSource2 src;
    src.fd = fd; src.file_off = file_offset;
    return store_upload_async(ctx, src, bytes, out_device_ptr);
}

static int store_upload_async(gauss_ctx* ctx, const RowSource2& src, int64_t bytes, void** out_device_ptr)
{
    HIPCHK(hipSetDevice(ctx->device));
    void* d = nullptr;
    hipError_t e = ctx_malloc_retry(ctx, &d, (size_t)bytes + 64);
    if (e != hipSuccess) return fail(GAUSS_E_NOMEM, "hipMalloc(%lld bytes row store) failed: %s", (long long)bytes, hipGetErrorString(e));
    std::shared_ptr<StoreUpload> up(new StoreUpload());
    StoreUpload* u = up.get();
    u->d = d; u->bytes = (size_t)bytes;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        ctx->stores[d] = (size_t)bytes;
        ctx->uploads[d] = std::move(up);
    }
    const int device = ctx->device;
    hipStream_t us = ctx->upload;
    u->th = std::thread([ctx, u, src, device, us]() {
        (void)hipSetDevice(device);
        // A background upload runs beside whatever the context computes.  Up to 4 GB (a chromosome's rows: the upload is over
        // after 20 ms, most of it before the first large batch) its chunks travel by hipMemcpyAsync: measured on a chromosome's
        // first call (round 4, tools/cold_trace.sh), with the copy kernel the first batch's 4.9 ms of GPU work ended with the
        // upload, 13.6 ms after it began -- a stream of copy kernels at raised wave priority holds the Gram kernel back -- and
        // beside the DMA engines it takes 5.0 ms.  Above (a whole-genome panel: seconds of PCIe traffic beside full-size Gram
        // launches, where hipMemcpyAsync was measured to stall, tools/h2d_under_load_probe.py) by kernel.  GAUSS_UPLOAD_BY_KERNEL=0 / 1.
        const int bk = env_int("GAUSS_UPLOAD_BY_KERNEL", -1);
        const bool by_kernel = bk >= 0 ? bk != 0 : u->bytes > ((size_t)4 << 30);
        const int rc = upload_rows(ctx, u->d, src, u->bytes, us, [u, us](size_t upto) {
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, us) != hipSuccess) return;
            std::lock_guard<std::mutex> lock(u->mu);
            u->marks.emplace_back(upto, ev);
            u->cv.notify_all();
        }, by_kernel);
        std::lock_guard<std::mutex> lock(u->mu);
        u->rc = rc;
        if (rc) u->err = g_err;
        u->done = true;
        u->cv.notify_all();
    });
    *out_device_ptr = d;
    return GAUSS_OK;
}

int gauss_store_wait(gauss_ctx* ctx, const void* device_ptr, int64_t bytes_needed)
{
    if (!ctx || !device_ptr) return fail(GAUSS_E_INVALID, "bad arguments to gauss_store_wait");
    // the waiter holds the bookkeeping alive: another waiter (or gauss_store_free) may retire the entry meanwhile
    std::shared_ptr<StoreUpload> u;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        auto it = ctx->uploads.find(device_ptr);
        if (it == ctx->uploads.end()) {
            // not an asynchronous store, or complete and retired -- but it must still BE a store of this context: a caller that
            // kept the pointer of a store another call has freed since (two chromosome calls in flight on one resident panel whose
            // background upload failed: the first to notice evicts it) is told so, instead of queuing a job over freed memory
            if (!ctx->stores.count(device_ptr))
                return fail(GAUSS_E_INVALID, "gauss_store_wait: not a row store of this context (freed by another call after a failed upload?)");
            return GAUSS_OK;
        }
        u = it->second;
    }
    const bool all = bytes_needed <= 0 || (size_t)bytes_needed >= u->bytes;
    hipEvent_t ev = nullptr;
    bool done = false;
    {
        std::unique_lock<std::mutex> lock(u->mu);
        const size_t need = all ? u->bytes : (size_t)bytes_needed;
        // the whole store: until the upload thread has synchronised its queue (`done`), as the header promises -- the
        // host may read the rows' consequences right after; a prefix: until the mark that covers it has been queued
        u->cv.wait(lock, [&] { return u->done || (!all && !u->marks.empty() && u->marks.back().first >= need); });
        if (u->rc) return fail(u->rc, "%s", u->err.c_str());
        done = u->done;
        if (!done)
            for (auto& m : u->marks) if (m.first >= need) { ev = m.second; break; }
    }
    if (done) {
        // complete (upload_rows synchronised its stream): nothing to wait for, and the bookkeeping can go
        std::shared_ptr<StoreUpload> dead;
        {
            std::lock_guard<std::mutex> lock(ctx->mu);
            auto it = ctx->uploads.find(device_ptr);
            if (it != ctx->uploads.end() && it->second == u) { dead = std::move(it->second); ctx->uploads.erase(it); }
        }
        return GAUSS_OK;                                          // (`dead` and `u` let go of the entry here, outside ctx->mu)
    }
    HIPCHK(hipSetDevice(ctx->device));
    // (the event belongs to `u`, which this call holds; marks are only destroyed with it)
    if (ev) HIPCHK(hipStreamWaitEvent(ctx->stream, ev, 0));
    return GAUSS_OK;
}

int gauss_store_free(gauss_ctx* ctx, void* device_ptr)
{
    if (!ctx) return fail(GAUSS_E_INVALID, "ctx is NULL");
    HIPCHK(hipSetDevice(ctx->device));
    if (device_ptr) {
        {
            // an upload that is still running is finished first (its destructor joins the thread)
            std::shared_ptr<StoreUpload> dead;
            { std::lock_guard<std::mutex> lock(ctx->mu); auto it = ctx->uploads.find(device_ptr); if (it != ctx->uploads.end()) { dead = std::move(it->second); ctx->uploads.erase(it); } }
            if (dead) dead->finish();             // (a concurrent gauss_store_wait may hold the entry a little longer: the thread has ended either way)
        }
        { std::lock_guard<std::mutex> lock(ctx->mu); ctx->stores.erase(device_ptr); }
        HIPCHK(hipFree(device_ptr));
    }
    return GAUSS_OK;
}

}  // extern "C"
