// libgauss_hip.so -- C ABI, planner and job orchestration (see include/gauss_hip.h).
//
// A job is a batch of independent windows that share every launch: one pack, one Gram, one
// epilogue, nblk factor steps and one solve launch serve all windows of the job, so the chip is
// filled by work items of many windows at once (the per-window matrices are too small to fill
// 256 CUs on their own) and the latency-bound factor steps are amortised over the batch.
#include "gauss_internal.h"
#include "../../include/gauss_hip.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <thread>
#include <string>
#include <vector>

#include <unistd.h>

using namespace gauss;

namespace gauss {
void launch_jacobi_clamp(const Prob* d_probs, int prob, const Prob& hp, double* d_work, bool apply, hipStream_t s);
}

// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(GAUSS_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                  \
    } while (0)

// Freed job workspaces (device) and staging blocks (pinned host) are kept per context and handed to the next job
// that fits: hipFree / hipHostFree wait for the device to go idle, which would stall a pipeline that retires job
// k while job k+1 is running, and a 15 GB hipMalloc per chromosome is not free either.
struct BlockCache {
    std::multimap<size_t, void*> free_blocks;
    size_t held = 0;
    void* take(size_t bytes)
    {
        auto it = free_blocks.lower_bound(bytes);
        if (it == free_blocks.end() || it->first > bytes + bytes / 2 + (1u << 20)) return nullptr;
        void* p = it->second;
        held -= it->first;
        free_blocks.erase(it);
        return p;
    }
};

struct gauss_job;

// One long-lived thread per context that runs the copy loop of a streamed window (job_run_streamed): starting a thread
// per call cost ~170 us before the first byte moved.
struct CopyWorker {
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> task;
    bool busy = false, stop = false;
    std::thread th;
    explicit CopyWorker(int device)
    {
        th = std::thread([this, device]() {
            (void)hipSetDevice(device);
            std::unique_lock<std::mutex> lock(mu);
            for (;;) {
                cv.wait(lock, [&] { return stop || task; });
                if (stop) return;
                std::function<void()> t;
                t.swap(task);
                lock.unlock();
                t();
                lock.lock();
                busy = false;
                cv.notify_all();
            }
        });
    }
    void submit(std::function<void()> t)
    {
        std::lock_guard<std::mutex> lock(mu);
        task = std::move(t); busy = true;
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return !busy; });
    }
    ~CopyWorker()
    {
        { std::lock_guard<std::mutex> lock(mu); stop = true; cv.notify_all(); }
        if (th.joinable()) th.join();
    }
};

// A row store whose bytes are still on their way (gauss_store_upload_async): a library thread streams them through the
// pinned double buffers in order and leaves a mark (bytes landed so far, event on the upload stream) after every chunk;
// gauss_store_wait makes the main stream wait for the mark that covers what a job is about to read.
struct StoreUpload {
    void* d = nullptr;
    size_t bytes = 0;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::pair<size_t, hipEvent_t>> marks;      // (bytes issued up to here, event recorded behind that copy)
    bool done = false;
    int rc = 0;
    std::string err;
    std::mutex join_mu;
    void finish() { std::lock_guard<std::mutex> lock(join_mu); if (th.joinable()) th.join(); }      // the upload thread has ended
    ~StoreUpload()
    {
        finish();
        for (auto& m : marks) if (m.second) hipEventDestroy(m.second);
    }
};

struct gauss_ctx {
    int device;
    uint64_t id = 0;                         // unique per process, never reused (a new context at a freed context's address is a new id)
    hipStream_t stream;
    // B21's half of the LD epilogue runs here, beside the factorisation chain on `stream` (which only needs B11): the
    // chain's launches are few, short and dependent and leave most of the chip idle (GAUSS_SIDE_STREAM=0: one stream)
    hipStream_t side = nullptr;
    // host -> HBM copies of a streamed window (gauss_impute_window on host bytes) run here, chunk by chunk, while the
    // main stream already packs and multiplies the rows that have landed; created on first use
    hipStream_t copy = nullptr;
    hipStream_t aux = nullptr;               // streamed window: pack + row tables of the chunk that has just landed
    hipStream_t chain = nullptr;             // streamed window: B11's epilogue tiles + the factorisation chain (high priority)
    struct CopyWorker* worker = nullptr;     // the thread that issues a streamed window's copies (created on first use)
    // where a streamed window's raw rows land: owned by the context and kept between calls, so that the first copy can
    // be issued before the window has even been planned (its destination is known at once)
    uint8_t* landing = nullptr;
    size_t landing_bytes = 0;
    std::vector<hipEvent_t> ev_pool;         // "chunk g has landed" events, reused by every streamed call
    std::mutex stream_mu;                    // one streamed call at a time per context (they share landing buffer and worker)
    hipStream_t upload = nullptr;            // asynchronous row-store uploads (gauss_store_upload_async)
    std::map<const void*, std::shared_ptr<StoreUpload>> uploads;      // by device pointer (guarded by mu); shared: a waiter keeps its entry alive
    int gram_i8 = 0;
    std::map<const void*, size_t> stores;    // row stores made by gauss_store_upload: base pointer -> bytes
    std::mutex mu;
    BlockCache dev_cache, pin_cache;
    std::map<void*, size_t> block_size;      // every live block handed out by ctx_dev_alloc / ctx_pin_alloc
    std::set<gauss_job*> jobs;               // live jobs of this context (guarded by mu): gauss_hip_destroy orphans them
    size_t dev_cache_limit = 0;              // bytes of freed workspace kept for reuse (a third of the device's memory)
    std::thread prepin;                      // makes the upload staging buffers in the background (gauss_hip_init)
    std::mutex prepin_mu;
};

// Lifetime rule of the C ABI (include/gauss_hip.h): a context may be destroyed while jobs and row stores made on it
// are still alive.  gauss_hip_destroy waits for their queued work, releases everything they hold on the device and
// leaves the job handles as empty shells ("orphans": ctx == nullptr) that only gauss_job_destroy accepts; destroy hooks
// let the host layer drop what it cached per context.  An Rcpp driver whose objects unwind in any order
// (Rcpp::stop between create and destroy) therefore never touches freed memory.
static std::mutex g_hook_mu;
static std::vector<std::pair<void (*)(gauss_ctx*, uint64_t, void*), void*>> g_destroy_hooks;
static std::atomic<uint64_t> g_next_ctx_id{1};

static const size_t PIN_CACHE_LIMIT = (size_t)1 << 30;
static const size_t UPLOAD_CHUNK = (size_t)32 << 20;       // staging buffer of a row-store upload (upload_rows)

static void ctx_flush_dev_cache_locked(gauss_ctx* c)
{
    for (auto& kv : c->dev_cache.free_blocks) { (void)hipFree(kv.second); c->block_size.erase(kv.second); }
    c->dev_cache.free_blocks.clear(); c->dev_cache.held = 0;
    (void)hipGetLastError();
}
// hipMalloc that gives the context's cached workspaces back to the device before it reports failure (a row store of
// tens of GB, a scratch buffer or another context on the same device may need the room the cache is sitting on)
static hipError_t ctx_malloc_retry(gauss_ctx* c, void** out, size_t bytes)
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

static hipError_t ctx_dev_alloc(gauss_ctx* c, size_t bytes, void** out)
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
static void ctx_dev_release(gauss_ctx* c, void* p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lock(c->mu);
    auto it = c->block_size.find(p);
    const size_t bytes = it == c->block_size.end() ? 0 : it->second;
    if (bytes && c->dev_cache.held + bytes <= c->dev_cache_limit) { c->dev_cache.free_blocks.emplace(bytes, p); c->dev_cache.held += bytes; return; }
    if (it != c->block_size.end()) c->block_size.erase(it);
    (void)hipFree(p);
}
static hipError_t ctx_pin_alloc(gauss_ctx* c, size_t bytes, void** out)
{
    std::lock_guard<std::mutex> lock(c->mu);
    if (void* p = c->pin_cache.take(bytes)) { *out = p; return hipSuccess; }
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e == hipSuccess) c->block_size[*out] = bytes;
    return e;
}
static void ctx_pin_release(gauss_ctx* c, void* p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lock(c->mu);
    auto it = c->block_size.find(p);
    const size_t bytes = it == c->block_size.end() ? 0 : it->second;
    if (bytes && c->pin_cache.held + bytes <= PIN_CACHE_LIMIT) { c->pin_cache.free_blocks.emplace(bytes, p); c->pin_cache.held += bytes; return; }
    if (it != c->block_size.end()) c->block_size.erase(it);
    (void)hipHostFree(p);
}

static inline size_t rup(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Device scratch released on every exit path (an early HIPCHK return included).
struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    hipError_t alloc(gauss_ctx* c, size_t bytes) { return ctx_malloc_retry(c, &p, bytes ? bytes : 1); }
    template <typename T> T* as() const { return (T*)p; }
};

// Host-side plan of one problem
struct Plan {
    Prob p;                                  // device descriptor (pointers filled at layout time)
    std::vector<int> pop_raw_off, pop_pk_off, seg_pop, seg_k0, seg_k1, pop_seg0;
    std::vector<int> pair_ti, pair_tj, pair_lut;
    std::vector<double> pop_w, pop_wf, pop_md, z1;
    std::vector<uint8_t> word_pop, word_run;
    std::vector<uint32_t> chunk_live;      // nibble per K chunk (Item::chunk_live)
    std::vector<int> run_pk_off, run_src, run_len;     // 2-bit source blocks: packed column range, byte offset in a source row, live samples
    bool run_len_known(size_t q) const { return q < run_len.size(); }
    std::vector<int32_t> rows_m, rows_u;     // store rows; empty = contiguous
    size_t row_bytes = 0;                    // bytes of a source row that the kernels read
    std::vector<int> gene_off;
    std::vector<long long> gene_out_off;
    std::vector<std::pair<int, int>> groups;   // runs of consecutive segments handled by one work item
    std::vector<std::pair<int, int>> fine;     // one run per segment: the short work items that fill the end of the launch
    std::vector<int> tile_live;                // job-wide measured rows: live rows of every row tile (clusters end in padding rows)
    // user pointers
    const uint8_t* h_geno_m = nullptr;
    const uint8_t* h_geno_u = nullptr;
    long long user_ld = 0;
    double* out_z = nullptr;
    double* out_info = nullptr;
    int32_t* out_status = nullptr;
    double* out_b11 = nullptr;
    double* out_b21 = nullptr;
    double* out_r = nullptr;
    int32_t* out_num_eig = nullptr;
    double* out_ld_user = nullptr;           // ld_only / gene outputs
    int U_user = 0;                          // geno_u rows as passed by the caller (before codings)
    size_t out_ld_count = 0;
    double* d_b11_copy = nullptr;
    size_t res_off = 0;                      // offset (in doubles) of this problem's z in the result block
};

struct ProfSlot { hipEvent_t a, b; int kernel; unsigned run; int launches = 1; };      // run: gauss_job::run_seq of the run that recorded it;
                                                                                        // launches: kernel launches the slot spans

struct gauss_job {
    gauss_ctx* ctx = nullptr;
    int n = 0;
    int on_device = 0;
    std::vector<Plan> plans;
    char* d_tab = nullptr;      size_t tab_bytes = 0;     // tables (host mirrored)
    char* d_ws = nullptr;       size_t ws_bytes = 0;      // workspace
    std::vector<char> h_tab;                               // host image of the tables while they are being built
    char* h_pin = nullptr;                                 // pinned block: [table image | results | status]
    hipEvent_t begin = nullptr;                            // recorded when gauss_job_run starts queuing
    hipEvent_t zeroed = nullptr;                           // the workspace has been zeroed (on the upload queue: job_build)
    hipEvent_t done = nullptr;                             // recorded after the result copies of gauss_job_run
    // Cross-queue events of a run, one set per run parity (two runs of a job may be in flight, see run_seq below):
    //   gram  B11's Gram launch (chain-aside) / the Gram kernel (otherwise) has been queued behind on the main queue
    //   side  the chain queue / side queue has finished what the closing product reads
    //   pack  operands packed (main)            rows  row tables + certificate done (GAUSS_ROWS_ASIDE: side queue)
    // Every one is armed by a plain hipEventRecord in stream order and consumed by a hipStreamWaitEvent issued LATER IN
    // HOST ORDER by the same call of job_run: the wait captures the record that precedes it, so re-arming the event for the
    // next run could never redirect an earlier wait.  Each parity still owns its set: no event object is ever re-armed
    // while a wait on its previous record may be pending in another queue (tools/experiments/README.md, "any-order launch").
    struct RunEvents { hipEvent_t gram = nullptr, side = nullptr, pack = nullptr, rows = nullptr, epi = nullptr; };     // epi: the early windows' epilogue tiles done (low-priority queue)
    RunEvents rev[2];
    Prob* d_probs = nullptr;
    Item* d_items = nullptr;    int n_items = 0;
    int2* d_rowmap = nullptr;   int n_rows = 0;
    // Shared measured rows (job_build): the windows of a chromosome name their measured SNPs as rows of one resident store,
    // each window a contiguous run of the chromosome's measured SNPs.  Those rows are then packed once into job-wide
    // arrays ("problem" n behind the windows' descriptors), B11's tile pairs are formed on job-wide row tiles and
    // multiplied once for all the windows they lie in (consecutive 2 Mb windows share half their measured SNPs: the
    // reference recomputes every pair per call, dist.cpp:171-179); every window's epilogue reads the shared slabs.
    std::unique_ptr<Plan> gplan;                           // the job-wide measured rows; null: nothing is shared
    std::vector<int> g0;                                   // per window: job-wide index of its first measured SNP
    std::vector<std::vector<int2>> win_tiles;              // per window: its epilogue tiles (tilemap entries)
    int2* d_tilemap = nullptr;  int n_tiles = 0;            // LD epilogue tiles: B11's first (n_tiles_b11 of them), then B21's
    int n_tiles_b11 = 0;
    int n_items_b11 = 0;                                   // chain_aside: B11's work items come first and are a launch of their own
    bool chain_aside = false;                              // the factorisation chain runs beside the Gram launch of B21's items (job_run)
    int2* d_panelmap = nullptr; int n_panels = 0;          // fused path: (window, panel of [I | z1])
    int2* d_dpanelmap = nullptr; int n_dpanels = 0;        // stand-alone solve: (window, panel of right-hand sides)
    int2* d_gemmmap = nullptr;  int n_gemm = 0;            // (window, rhs panel of gemm_ut << 8 | k block of 128), longest first
    int gemm_ut = 128;                                     // right-hand sides per tile of the product: 128, small jobs 64
    int2* d_finmap = nullptr;   int n_fin = 0;             // (window, chunk of 256 right-hand sides)
    int max_nblk = 0;
    int max_npanel = 0;                                    // most solve panels of any one window
    int solve_split = 0;                                   // rows of the inverse with at least this many products are cut (0: none)
    int own_panel = 0;                                     // 1: small job, the update launches form their own panel tiles (no panel launches)
    // Streamed single window (job_run_streamed): the measured rows first, then the unmeasured rows in chunks of whole
    // row tiles; a group's work items need the rows of groups <= it only
    struct StreamGroup { int item0, n_items, row0, n_rows, tile0, n_tiles; };      // tile0 / n_tiles: the chunk's B21 epilogue tiles
    std::vector<StreamGroup> sgroups;                      // empty: not a streamed job
    std::vector<hipEvent_t> sevp;                          // "group g's rows are packed, their tables made" (aux stream)
    int max_pop = 1;
    int gram_i8 = 0;
    int* d_status = nullptr;                               // [n][4] + 4 job-wide ints
    unsigned long long* d_b11_done = nullptr;              // merged Gram launch: B11's items that have finished, over all runs so far
    bool merged = false;                                   // chain_aside as ONE Gram launch (B11's items first, counted; job_run)
    double wait_bound_us = 0.0;                            // give-up bound of the merged launch's waits: 50 x the launch's estimated time (>= 2 s)
    unsigned long long merged_runs = 0;                    // merged Gram launches queued so far (the counter's target is this x n_items_b11)
    // Early epilogue (merged launches): B21's items of the "early" windows come before those of the "late" ones and count
    // themselves off in a second counter; their epilogue tiles wait for that count on the LOW-priority queue, whose workgroups
    // the hardware dispatches when the Gram grid has none left to hand out -- i.e. they fill the Gram launch's last round.
    int n_items_b21_early = 0;                             // B21 items of the early windows (right behind B11's items)
    int n_tiles_b21_early = 0;                             // their epilogue tiles (first among B21's tiles)
    double* d_results = nullptr; size_t n_results = 0;     // z then info per problem
    // Two runs may be in flight: run k + 1 can be queued before run k has been fetched, so that the host's share of a
    // step (waking up, copying results out, queuing the next run) overlaps GPU work.  Result mirrors and completion
    // events alternate by run parity; gauss_job_fetch collects the oldest run that has not been fetched.
    double* h_res2[2] = {nullptr, nullptr};                // inside h_pin
    int* h_st2[2] = {nullptr, nullptr};                    // inside h_pin
    hipEvent_t done2[2] = {nullptr, nullptr};
    unsigned run_seq = 0, fetch_seq = 0;
    double* h_results = nullptr;                           // mirror of the run being fetched
    int* h_status = nullptr;
    bool prof = false;
    std::vector<ProfSlot> slots;
    double prof_ms[5] = {0, 0, 0, 0, 0};
    long long prof_n[5] = {0, 0, 0, 0, 0};
    bool ran = false;
};

// What gauss_impute_window fixes BEFORE the window is planned, so that the copy worker can start at once: where the raw
// rows land, how the unmeasured rows are cut into chunks, and the events that say "chunk g has landed".
struct StreamSetup {
    uint8_t* d_m = nullptr;                  // measured rows  [M][ldraw]   (inside the context's landing buffer)
    uint8_t* d_u = nullptr;                  // unmeasured rows [U][ldraw]
    long long ldraw = 0;
    size_t row_bytes = 0;
    int M = 0, U = 0;
    std::vector<int> first_tile;             // first_tile[g], g >= 1: first unmeasured row tile of chunk g; back() = all tiles
    std::vector<int> tile_group;             // unmeasured row tile -> chunk (>= 1)
    std::vector<hipEvent_t> ev;              // [0] measured rows, [g] chunk g
    std::mutex mu;                           // progress of the copy worker
    std::condition_variable cv;
    int recorded = 0, rc = 0;
    std::string err;
    int n_groups() const { return (int)first_tile.size() - 1; }
};

// ------------------------------------------------------------------------------------------
// planning
// ------------------------------------------------------------------------------------------
struct WinSpec {
    int mode, n_pop;
    const int32_t* pop_off;
    const double* pop_wgt;
    int M, U;
    const uint8_t* geno_m;
    const uint8_t* geno_u;
    long long ld;
    const double* z1;
    double lambda, eps, diag;
    int ld_only;
    const int32_t* gene_off;
    int n_gene;
    int kind = 0, n_head = 0, n_predm = 0;   // QCAT windows (qcat.cpp:134-262)
    int u_codings = 0;                       // GAUSS_CODE_* mask for the geno_u rows (0 = additive)
    int geno_fmt = 0;                        // GAUSS_GENO_*
    const int32_t* rows_m = nullptr;         // store rows (host arrays) or NULL = contiguous matrices
    const int32_t* rows_u = nullptr;
    const int32_t* pop_src_off = nullptr;    // 2-bit: byte offset of each population block in a row
    double eig_cutoff = 0.01;
    const int32_t* pair_i = nullptr;         // LD-only: only these SNP pairs are wanted (tile pairs they touch), else all
    const int32_t* pair_j = nullptr;
    int64_t n_pairs = 0;
};

// K segment length: short segments give a single window enough work items to fill 256 CUs;
// a batch of windows has plenty of items already, and longer segments mean fewer partial slabs
// for the epilogue to read back.  Either way a partial sum stays an exact f32 integer
// (15 * 15 * 8192 < 2^24).
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static int seg_max_for(size_t n_windows)
{
    static const int ov = env_int("GAUSS_SEG_MAX", 0);          // experiment override (multiple of 64)
    if (ov > 0) return ov / KC * KC;
    // 4096 for batched jobs (8192 until the chain moved beside the Gram kernel: with B11's and B21's items in launches of their
    // own the shorter items balance each launch's last round better -- 5 / 9 / 18 / 36 windows: step 4.86 -> 4.75, 8.97 -> 8.93,
    // 19.47 -> 19.29, 41.00 -> 40.96 ms -- for 19 % more work items)
    return n_windows >= 4 ? 4096 : SEG_MAX;
}
// Consecutive segments are chained into one work item until the run reaches this many samples
// (a fresh item costs a pipeline fill: descriptor, first operand tiles, barrier).  0 = no chaining.
static int group_target_for(size_t n_windows)
{
    static const int ov = env_int("GAUSS_GROUP_TARGET", -1);    // experiment override
    if (ov >= 0) return ov;
    // 2048 rather than 4096: same kernel time on the bench workload, 13 % less fabric read traffic (the tiles
    // co-resident items share stay in the XCD's L2 more often; tools/group_pmc.sh)
    return n_windows >= 4 ? 2048 : 0;
}

static int plan_problem(const WinSpec& w, Plan& pl, int seg_max, int group_target)
{
    if (w.mode != GAUSS_MODE_POOLED && w.mode != GAUSS_MODE_WEIGHTED) return fail(GAUSS_E_INVALID, "bad mode %d", w.mode);
    if (w.n_pop < 1 || w.n_pop > 64) return fail(GAUSS_E_INVALID, "n_pop must be in 1..64 (got %d)", w.n_pop);
    if (!w.pop_off) return fail(GAUSS_E_INVALID, "pop_off is NULL");
    if (w.M < 1) return fail(GAUSS_E_INVALID, "need at least one measured SNP row (got %d)", w.M);
    if (w.U < 0) return fail(GAUSS_E_INVALID, "negative n_unmeasured");
    if (w.mode == GAUSS_MODE_WEIGHTED && !w.pop_wgt) return fail(GAUSS_E_INVALID, "pop_wgt is NULL in weighted mode");
    if (!w.geno_m || (w.U > 0 && !w.geno_u)) return fail(GAUSS_E_INVALID, "genotype pointer is NULL");
    for (int p = 0; p < w.n_pop; p++)
        if (w.pop_off[p + 1] < w.pop_off[p]) return fail(GAUSS_E_INVALID, "pop_off must be non-decreasing");
    if (w.pop_off[0] != 0) return fail(GAUSS_E_INVALID, "pop_off[0] must be 0");
    const int N = w.pop_off[w.n_pop];
    if (N < 1) return fail(GAUSS_E_INVALID, "no samples");
    // sums of code products are kept as exact integers: 15 * 15 * n must stay below 2^31
    if ((long long)N * 225 >= (1LL << 31)) return fail(GAUSS_E_RANGE, "%d samples exceed the exact-integer range (9.5 M)", N);
    if (w.geno_fmt != GAUSS_GENO_U8 && w.geno_fmt != GAUSS_GENO_2BIT) return fail(GAUSS_E_INVALID, "bad geno_format %d", w.geno_fmt);
    if (w.geno_fmt == GAUSS_GENO_U8 && w.ld < N) return fail(GAUSS_E_INVALID, "ld (%lld) < n_samples (%d)", w.ld, N);
    if (w.geno_fmt == GAUSS_GENO_2BIT) {
        if (w.ld % 16) return fail(GAUSS_E_INVALID, "2-bit rows need a stride that is a multiple of 16 bytes (got %lld)", w.ld);
        long long end = 0;
        for (int q = 0; q < w.n_pop; q++) {
            const long long blk = (long long)rup((size_t)(w.pop_off[q + 1] - w.pop_off[q]), 64) / 4;
            const long long off = w.pop_src_off ? w.pop_src_off[q] : end;
            if (off < 0 || off % 16) return fail(GAUSS_E_INVALID, "pop_src_off[%d] = %lld is not a multiple of 16", q, off);
            if (off + blk > w.ld) return fail(GAUSS_E_INVALID, "population block %d ends past the row stride", q);
            if (!w.pop_src_off) end = off + blk;
        }
    }
    if (w.kind != GAUSS_WIN_IMPUTE && w.kind != GAUSS_WIN_QCAT && w.kind != GAUSS_WIN_LD)
        return fail(GAUSS_E_INVALID, "bad window kind %d", w.kind);
    if (!w.ld_only && w.kind != GAUSS_WIN_LD && (w.U > 0 || w.kind == GAUSS_WIN_QCAT) && !w.z1) return fail(GAUSS_E_INVALID, "z1 is NULL");
    if (w.u_codings & ~(GAUSS_CODE_ADDITIVE | GAUSS_CODE_DOMINANT | GAUSS_CODE_RECESSIVE))
        return fail(GAUSS_E_INVALID, "bad u_codings mask %d", w.u_codings);
    if (w.kind == GAUSS_WIN_QCAT && (w.n_head < 0 || w.n_predm < 0 || w.n_head + w.n_predm > w.M))
        return fail(GAUSS_E_INVALID, "QCAT: n_head_measured + n_pred_measured exceeds n_measured");

    Prob& p = pl.p;
    memset(&p, 0, sizeof(p));
    p.mode = w.mode;
    p.M = w.M; p.N = N;
    {
        int nc = 0;
        for (int c = 0; c < 3; c++) if (w.u_codings & (1 << c)) p.code_blk[nc++] = c;
        if (nc == 0) { p.code_blk[0] = 0; nc = 1; }
        p.U_raw = std::max(w.U, 1);
        p.U = w.U * nc;                          // one block of U rows per coding
    }
    p.lambda = w.lambda; p.diag = w.diag;
    p.ld_only = w.ld_only;
    p.kind = w.kind; p.n_head = w.n_head; p.n_predm = w.n_predm;
    // the shifted factorisation tests lambda_min against MakePosDef's floor (imputation, util.cpp:310)
    // or against CountPC's cutoff (QCAT, util.cpp:379)
    p.eps = (w.kind == GAUSS_WIN_QCAT) ? w.eig_cutoff : w.eps;
    p.n_rhs = (w.kind == GAUSS_WIN_QCAT) ? w.n_predm + p.U : p.U;
    if (w.mode == GAUSS_MODE_POOLED) {
        // CalCor pools every selected population (util.cpp:53-64): one pseudo-population
        p.P = 1;
        pl.pop_raw_off = {0, N};
        pl.pop_w = {1.0};
    } else {
        p.P = w.n_pop;
        pl.pop_raw_off.assign(w.pop_off, w.pop_off + w.n_pop + 1);
        pl.pop_w.assign(w.pop_wgt, w.pop_wgt + w.n_pop);
    }
    const int P = p.P;
    for (int q = 0; q < P; q++) {
        const int m = pl.pop_raw_off[q + 1] - pl.pop_raw_off[q];
        const double factor = ((double)m) / (m - 1);           // util.cpp:117 (inf for m == 1, like the reference)
        pl.pop_wf.push_back(pl.pop_w[q] * factor);             // util.cpp:118: wgt_val*factor*(...) groups left to right
        pl.pop_md.push_back((double)m);
    }
    pl.pop_pk_off.assign(P + 1, 0);
    for (int q = 0; q < P; q++) {
        const int m = pl.pop_raw_off[q + 1] - pl.pop_raw_off[q];
        pl.pop_pk_off[q + 1] = pl.pop_pk_off[q] + (int)rup((size_t)m, KC);
    }
    p.geno_fmt = w.geno_fmt;
    pl.row_bytes = (size_t)N;
    if (w.geno_fmt == GAUSS_GENO_2BIT) {
        // every selected population is one source block ("run") padded to 64 samples, in the source row and in
        // the packed operand row alike; pooled statistics still see one pseudo-population spanning all runs
        pl.run_pk_off.assign(w.n_pop + 1, 0);
        long long end = 0;
        pl.row_bytes = 0;
        for (int q = 0; q < w.n_pop; q++) {
            const int blk = (int)rup((size_t)(w.pop_off[q + 1] - w.pop_off[q]), 64);
            pl.run_pk_off[q + 1] = pl.run_pk_off[q] + blk;
            const long long off = w.pop_src_off ? w.pop_src_off[q] : end;
            pl.run_src.push_back((int)off);
            pl.run_len.push_back(w.pop_off[q + 1] - w.pop_off[q]);
            if (!w.pop_src_off) end = off + blk / 4;
            pl.row_bytes = std::max(pl.row_bytes, (size_t)(off + blk / 4));
        }
        if (P == 1) pl.pop_pk_off[1] = pl.run_pk_off[w.n_pop];
        p.n_run = w.n_pop;
        pl.word_run.assign(pl.run_pk_off[w.n_pop] / 16, 0);
        for (int q = 0; q < w.n_pop; q++)
            for (int b = pl.run_pk_off[q] / 16; b < pl.run_pk_off[q + 1] / 16; b++) pl.word_run[b] = (uint8_t)q;
    }
    if (w.rows_m) pl.rows_m.assign(w.rows_m, w.rows_m + w.M);
    if (w.rows_u && w.U > 0) pl.rows_u.assign(w.rows_u, w.rows_u + w.U);
    p.Kp = pl.pop_pk_off[P];
    // 16-bit partial slabs: 2-bit sources carry codes 0..3 (recoding only lowers them), so a segment of at most 7168
    // samples sums to <= 9 * 7168 < 2^16; the fast epilogue reads them (windows with LDS-resident population tables);
    // LD-only calls and gene batches keep f32 / int32 slabs
    static const bool no16 = getenv("GAUSS_NO_SLAB16") != nullptr;
    p.slab16 = (!no16 && w.geno_fmt == GAUSS_GENO_2BIT && !w.ld_only && !w.gene_off && P <= 32) ? 1 : 0;
    if (p.slab16) seg_max = std::min(seg_max, 7168);
    pl.word_pop.assign(p.Kp / 16, 0);
    // K chunks that end a zero-padded block (a population, or a 2-bit source block) with fewer than 64 live samples: how many
    // units of 8 samples are live (Item::chunk_live; 0 = the whole chunk)
    pl.chunk_live.assign((p.Kp / KC + 7) / 8 + 1, 0u);
    auto set_live = [&](int chunk, int samples) {
        const int units = (samples + 7) / 8;
        if (units >= 1 && units < 8) pl.chunk_live[chunk >> 3] |= (uint32_t)units << (4 * (chunk & 7));
    };
    if (w.geno_fmt == GAUSS_GENO_2BIT) {
        for (int q = 0; q < w.n_pop; q++) {
            const int m = w.pop_off[q + 1] - w.pop_off[q], rem = m % KC;
            if (m > 0 && rem) set_live(pl.run_pk_off[q + 1] / KC - 1, rem);
        }
    } else {
        for (int q = 0; q < P; q++) {
            const int m = pl.pop_raw_off[q + 1] - pl.pop_raw_off[q], rem = m % KC;
            if (m > 0 && rem) set_live(pl.pop_pk_off[q + 1] / KC - 1, rem);
        }
    }
    pl.pop_seg0.assign(P + 1, 0);
    for (int q = 0; q < P; q++) {
        for (int b = pl.pop_pk_off[q] / 16; b < pl.pop_pk_off[q + 1] / 16; b++) pl.word_pop[b] = (uint8_t)q;
        const int chunks = (pl.pop_pk_off[q + 1] - pl.pop_pk_off[q]) / KC;
        pl.pop_seg0[q] = (int)pl.seg_pop.size();
        if (chunks > 0) {
            const int max_chunks = seg_max / KC;
            const int ns = (chunks + max_chunks - 1) / max_chunks;
            const int per = (chunks + ns - 1) / ns;
            for (int c = 0; c < chunks; c += per) {
                const int c1 = std::min(chunks, c + per);
                pl.seg_pop.push_back(q);
                pl.seg_k0.push_back(pl.pop_pk_off[q] + c * KC);
                pl.seg_k1.push_back(pl.pop_pk_off[q] + c1 * KC);
            }
        }
    }
    pl.pop_seg0[P] = (int)pl.seg_pop.size();
    p.nseg = (int)pl.seg_pop.size();
    for (int s0 = 0; s0 < p.nseg;) {
        int s1 = s0 + 1;
        int len = pl.seg_k1[s0] - pl.seg_k0[s0];
        while (s1 < p.nseg && len < group_target) { len += pl.seg_k1[s1] - pl.seg_k0[s1]; s1++; }
        pl.groups.push_back(std::make_pair(s0, s1));
        s0 = s1;
    }
    for (int s0 = 0; s0 < p.nseg; s0++) pl.fine.push_back(std::make_pair(s0, s0 + 1));

    p.Mp = (int)rup((size_t)w.M, TILE);
    p.Up = (int)rup((size_t)p.U, TILE);
    p.Sp = p.Mp + p.Up;
    p.nT = p.Sp / TILE;
    const int mt = p.Mp / TILE;
    pl.pair_lut.assign((size_t)p.nT * p.nT, -1);
    auto add_pair = [&](int ti, int tj) {
        if (pl.pair_lut[(size_t)ti * p.nT + tj] >= 0) return;
        const int id = (int)pl.pair_ti.size();
        pl.pair_ti.push_back(ti); pl.pair_tj.push_back(tj);
        pl.pair_lut[(size_t)ti * p.nT + tj] = id;
        pl.pair_lut[(size_t)tj * p.nT + ti] = id;
    };
    if (w.gene_off) {
        // LD is only needed inside genes (gene.cpp:305-315): tile pairs touched by some gene
        pl.gene_off.assign(w.gene_off, w.gene_off + w.n_gene + 1);
        long long off = 0;
        for (int g = 0; g < w.n_gene; g++) {
            const int r0 = pl.gene_off[g], r1 = pl.gene_off[g + 1];
            if (r0 < 0 || r1 < r0 || r1 > w.M) return fail(GAUSS_E_INVALID, "gene_off out of range at gene %d", g);
            pl.gene_out_off.push_back(off);
            off += (long long)(r1 - r0) * (r1 - r0);
            if (r1 > r0)
                for (int ti = r0 / TILE; ti <= (r1 - 1) / TILE; ti++)
                    for (int tj = ti; tj <= (r1 - 1) / TILE; tj++) add_pair(ti, tj);
        }
        pl.out_ld_count = (size_t)off;
        p.n_gene = w.n_gene;
    } else if (w.pair_i) {
        // listed pairs (prep_zmix selectors): the tile pairs they touch
        for (int64_t k = 0; k < w.n_pairs; k++) {
            const int i = w.pair_i[k], j = w.pair_j[k];
            if (i < 0 || j <= i || j >= w.M) return fail(GAUSS_E_INVALID, "pair %lld = (%d, %d) is not i < j < n_snp", (long long)k, i, j);
            add_pair(i / TILE, j / TILE);
        }
        pl.out_ld_count = 0;
    } else {
        for (int ti = 0; ti < mt; ti++)
            for (int tj = ti; tj < mt; tj++) add_pair(ti, tj);          // B11 (upper tiles)
        for (int tu = mt; tu < p.nT; tu++)
            for (int tj = 0; tj < mt; tj++) add_pair(tu, tj);           // B21
        if (w.ld_only) pl.out_ld_count = (size_t)w.M * w.M;
    }
    p.npair = (int)pl.pair_ti.size();
    p.Mld = (int)rup((size_t)w.M, NB);
    p.nblk = p.Mld / NB;
    p.npanel = (w.ld_only || w.kind == GAUSS_WIN_LD) ? 0 : (p.n_rhs + NRU - 1) / NRU;
    p.npi = p.npanel > 0 ? (w.M + 1 + NR - 1) / NR : 0;
    p.Up128 = (int)rup((size_t)std::max(p.n_rhs, 1), 128);
    pl.U_user = w.U;
    if (w.z1) pl.z1.assign(w.z1, w.z1 + w.M);
    pl.h_geno_m = w.geno_m; pl.h_geno_u = w.geno_u; pl.user_ld = w.ld;
    return GAUSS_OK;
}

// Arena layout helper
struct Arena {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off = rup(off + bytes, 256); return o; }
};

template <typename T>
static size_t put(std::vector<char>& blob, Arena& a, const std::vector<T>& v)
{
    const size_t bytes = std::max<size_t>(v.size() * sizeof(T), 1);
    const size_t o = a.take(bytes);
    if (blob.size() < a.off) blob.resize(a.off);
    if (!v.empty()) memcpy(blob.data() + o, v.data(), v.size() * sizeof(T));
    return o;
}

static void job_free(gauss_job* job);

// streamed: one window on contiguous host matrices whose upload is left to job_run_streamed (chunk by chunk on the
// copy stream, overlapped with the pack / Gram launches of the rows that have landed)
static int job_build(gauss_ctx* ctx, const std::vector<WinSpec>& specs, int on_device, gauss_job** out, const StreamSetup* stream = nullptr)
{
    const bool streamed = stream != nullptr;
    const auto tb0 = std::chrono::steady_clock::now();
    gauss_job* job = new gauss_job();
    // every early return below (bad arguments, a failed HIP call) releases the job and what it owns
    std::unique_ptr<gauss_job, void (*)(gauss_job*)> guard(job, job_free);
    job->ctx = ctx;
    job->n = (int)specs.size();
    job->on_device = on_device;
    job->gram_i8 = ctx->gram_i8;
    job->plans.resize(job->n);
    for (int i = 0; i < job->n; i++) {
        int rc = plan_problem(specs[i], job->plans[i], seg_max_for(specs.size()), group_target_for(specs.size()));
        if (rc) return rc;
        job->plans[i].p.gram_i8 = job->gram_i8;
    }
    // Shared measured rows: every window reads its measured SNPs from the same resident store under the same populations.
    // The job keeps ONE list of measured rows, made of CLUSTERS: a window whose rows continue a run of the current cluster
    // (the next window of a chromosome: half its measured SNPs are the previous window's) joins it at that offset -- when
    // that costs no more B11 tile pairs than tiles of its own would -- and every other window starts a new cluster on a
    // tile boundary (padding rows in between: never packed, all zero), where it costs exactly what its own tiles would.
    // So sharing can only save work: the scattered windows of a multi-GPU share keep their own tiles, two neighbours that
    // landed on the same rank share theirs (round 3 had one all-or-nothing list whose tiles started where the LIST started:
    // a share's scattered windows paid an extra row tile each and sharing was switched off for them).
    if (on_device && !streamed && job->n >= 2 && env_int("GAUSS_SHARE_MEASURED", 1) != 0) {
        const Plan& a = job->plans[0];
        bool ok = true;
        std::vector<int32_t> gl;
        std::vector<int> g0((size_t)job->n, 0);
        std::set<std::pair<int, int>> have;                     // job-wide B11 tile pairs so far
        size_t cl0 = 0, own_pairs = 0;                          // start of the current cluster in gl
        for (int i = 0; i < job->n && ok; i++) {
            const Plan& b = job->plans[i];
            ok = !b.rows_m.empty() && !b.p.ld_only && !b.p.n_gene && b.p.P <= 32 && b.h_geno_m == a.h_geno_m && b.user_ld == a.user_ld &&
                 b.p.mode == a.p.mode && b.p.P == a.p.P && b.p.geno_fmt == a.p.geno_fmt && b.p.slab16 == a.p.slab16 && b.p.Kp == a.p.Kp &&
                 b.pop_raw_off == a.pop_raw_off && b.pop_pk_off == a.pop_pk_off && b.pop_w == a.pop_w && b.seg_k0 == a.seg_k0 &&
                 b.seg_k1 == a.seg_k1 && b.seg_pop == a.seg_pop && b.run_src == a.run_src && b.run_pk_off == a.run_pk_off &&
                 b.groups == a.groups;
            if (!ok) break;
            const size_t M = b.rows_m.size(), mt = (M + TILE - 1) / TILE;
            own_pairs += mt * (mt + 1) / 2;
            // does the window continue a run of the current cluster?  (the cluster is ascending where that matters: a window
            // only joins through a binary search for its first row and an element-wise comparison of the overlap)
            size_t pos = gl.size();
            bool join = false;
            if (gl.size() > cl0 && std::is_sorted(gl.begin() + (ptrdiff_t)cl0, gl.end())) {
                const int32_t first = b.rows_m[0];
                const size_t p0 = (size_t)(std::lower_bound(gl.begin() + (ptrdiff_t)cl0, gl.end(), first) - gl.begin());
                if (p0 < gl.size() && gl[p0] == first) {
                    join = true;
                    for (size_t k = 0; k < M && join; k++) {
                        if (k > 0 && b.rows_m[k] <= b.rows_m[k - 1]) join = false;
                        else if (p0 + k < gl.size() && gl[p0 + k] != b.rows_m[k]) join = false;
                    }
                    if (join) {
                        // tile pairs the window would ADD as part of the cluster, against tiles of its own
                        const int lo = (int)(p0 / TILE), hi = (int)((p0 + M - 1) / TILE);
                        size_t add = 0;
                        for (int ti = lo; ti <= hi; ti++)
                            for (int tj = ti; tj <= hi; tj++) add += have.count(std::make_pair(ti, tj)) ? 0 : 1;
                        join = add <= mt * (mt + 1) / 2;
                        pos = p0;
                    }
                }
            }
            if (!join) {
                gl.resize(rup(gl.size(), TILE), -1);               // a new cluster on a tile boundary (padding rows: never packed)
                cl0 = pos = gl.size();
            }
            g0[(size_t)i] = (int)pos;
            for (size_t k = 0; k < M; k++)
                if (pos + k >= gl.size()) gl.push_back(b.rows_m[k]);
            const int lo = (int)(pos / TILE), hi = (int)((pos + M - 1) / TILE);
            for (int ti = lo; ti <= hi; ti++)
                for (int tj = ti; tj <= hi; tj++) have.insert(std::make_pair(ti, tj));
        }
        // worth the extra descriptor only if something IS shared (GAUSS_SHARE_MEASURED=2: always, for the tests)
        if (ok && (long long)gl.size() < (1 << 24) && (have.size() < own_pairs || env_int("GAUSS_SHARE_MEASURED", 1) == 2)) {
            job->gplan.reset(new Plan(a));
            job->g0 = g0;
            Plan& g = *job->gplan;
            g.rows_m = gl;
            g.rows_u.clear(); g.z1.clear(); g.gene_off.clear(); g.gene_out_off.clear();
            Prob& q = g.p;
            q.M = (int)gl.size(); q.U = 0; q.U_raw = 1; q.n_rhs = 0;
            q.Mp = (int)rup((size_t)q.M, TILE); q.Up = 0; q.Sp = q.Mp; q.nT = q.Mp / TILE;
            q.Mld = 0; q.nblk = 0; q.npanel = 0; q.npi = 0; q.kind = 0; q.ld_only = 0; q.n_head = q.n_predm = 0;
            g.U_user = 0; g.h_geno_u = nullptr;
            // job-wide B11 pairs: the tile pairs some window lies in
            g.pair_ti.clear(); g.pair_tj.clear(); g.pair_lut.assign((size_t)q.nT * q.nT, -1);
            for (int i = 0; i < job->n; i++) {
                const int lo = g0[(size_t)i] / TILE, hi = (g0[(size_t)i] + job->plans[i].p.M - 1) / TILE;
                for (int ti = lo; ti <= hi; ti++)
                    for (int tj = ti; tj <= hi; tj++)
                        if (g.pair_lut[(size_t)ti * q.nT + tj] < 0) {
                            g.pair_lut[(size_t)ti * q.nT + tj] = g.pair_lut[(size_t)tj * q.nT + ti] = (int)g.pair_ti.size();
                            g.pair_ti.push_back(ti); g.pair_tj.push_back(tj);
                        }
            }
            q.npair = (int)g.pair_ti.size();
            // live rows per job-wide row tile: up to the last real row in it (a cluster's last tile ends in padding rows, whose
            // dead 32-row halves the Gram kernel skips exactly as it does in a window's own last tile)
            g.tile_live.assign((size_t)q.nT, 0);
            for (size_t r = 0; r < gl.size(); r++)
                if (gl[r] >= 0) g.tile_live[r / TILE] = (int)(r % TILE) + 1;
        }
    }
    const bool shm = job->gplan != nullptr;
    const int n_prob = job->n + (shm ? 1 : 0);             // descriptors on the device: the windows, then the job-wide rows
    auto plan_of = [&](int i) -> Plan& { return i < job->n ? job->plans[i] : *job->gplan; };

    // Row lists are resolved by the pack kernel: an index beyond the store would be an out-of-bounds read on the
    // GPU.  Host stores cannot be checked (only a pointer is known), stores made by gauss_store_upload can.
    std::map<const void*, size_t> stores;
    { std::lock_guard<std::mutex> lock(ctx->mu); stores = ctx->stores; }
    for (int i = 0; i < job->n; i++) {
        const Plan& pl = job->plans[i];
        for (int side = 0; side < 2; side++) {
            const std::vector<int32_t>& rows = side ? pl.rows_u : pl.rows_m;
            const uint8_t* base = side ? pl.h_geno_u : pl.h_geno_m;
            if (rows.empty()) continue;
            long long mx = -1;
            for (int32_t r : rows) {
                if (r < 0) { return fail(GAUSS_E_INVALID, "window %d: negative row index %d", i, (int)r); }
                mx = std::max<long long>(mx, r);
            }
            if (!on_device) continue;
            // the store that contains `base` (a window may point into the middle of an uploaded store)
            auto it = stores.upper_bound(base);
            if (it == stores.begin()) continue;                      // not one of ours: caller's responsibility
            --it;
            const uint8_t* s0 = (const uint8_t*)it->first;
            if (base >= s0 + it->second) continue;
            const size_t need = (size_t)(base - s0) + (size_t)mx * (size_t)pl.user_ld + pl.row_bytes;
            if (need > it->second) {
                return fail(GAUSS_E_INVALID, "window %d: row index %lld reaches past the end of the row store (%zu bytes)", i, mx, it->second);
            }
        }
    }
    HIPCHK(hipSetDevice(ctx->device));

    // ---- table arena (host mirrored) ----
    Arena ta;
    std::vector<char>& blob = job->h_tab;
    struct TabOff { size_t raw_off, pk_off, w, wf, md, seg_pop, k0, k1, seg0, ti, tj, lut, wp, z1, goff, gout, wr, rpk, rsrc, rm, ru, ch; };
    std::vector<TabOff> to((size_t)n_prob);
    for (int i = 0; i < n_prob; i++) {
        Plan& pl = plan_of(i);
        to[i].raw_off = put(blob, ta, pl.pop_raw_off);
        to[i].pk_off = put(blob, ta, pl.pop_pk_off);
        to[i].w = put(blob, ta, pl.pop_w);
        to[i].wf = put(blob, ta, pl.pop_wf);
        to[i].md = put(blob, ta, pl.pop_md);
        to[i].seg_pop = put(blob, ta, pl.seg_pop);
        to[i].k0 = put(blob, ta, pl.seg_k0);
        to[i].k1 = put(blob, ta, pl.seg_k1);
        to[i].seg0 = put(blob, ta, pl.pop_seg0);
        to[i].ti = put(blob, ta, pl.pair_ti);
        to[i].tj = put(blob, ta, pl.pair_tj);
        to[i].lut = put(blob, ta, pl.pair_lut);
        to[i].wp = put(blob, ta, pl.word_pop);
        to[i].z1 = put(blob, ta, pl.z1);
        to[i].goff = put(blob, ta, pl.gene_off);
        to[i].gout = put(blob, ta, pl.gene_out_off);
        to[i].wr = put(blob, ta, pl.word_run);
        to[i].rpk = put(blob, ta, pl.run_pk_off);
        to[i].rsrc = put(blob, ta, pl.run_src);
        to[i].rm = put(blob, ta, pl.rows_m);
        to[i].ru = put(blob, ta, pl.rows_u);
        to[i].ch = put(blob, ta, pl.chunk_live);
    }
    // work lists
    struct ItemH { int prob, pair, group, len; };
    std::vector<ItemH> items;
    std::vector<char> late_window;                         // early epilogue: windows whose B21 items end the merged launch
    std::vector<int2> rowmap, tilemap, tilemap_b21, panelmap, dpanelmap, gemmmap, finmap;
    job->max_nblk = 0;
    {
        // tiles of the closing product at 128 right-hand sides each: a small job (an 8-rank share: ~570) cannot fill the
        // chip's 512 workgroup slots with them and is bound by the tiles' K loops, so it takes 64 (k_solve.hip)
        size_t t128 = 0;
        for (int i = 0; i < job->n; i++) {
            const Prob& p = job->plans[i].p;
            if (p.npanel > 0) t128 += (size_t)(p.Up128 / 128) * ((p.Mld + 127) / 128);
        }
        const int small = env_int("GAUSS_GEMM_SMALL_TILES", 1600);         // read per job: tests drive both tile widths
        job->gemm_ut = t128 < (size_t)small ? 64 : 128;
    }
    job->win_tiles.assign((size_t)job->n, std::vector<int2>());
    if (shm) {
        // the job-wide measured rows are packed once, and B11's job-wide tile pairs multiplied once
        const Plan& g = *job->gplan;
        for (int pr = 0; pr < g.p.npair; pr++)
            for (size_t k = 0; k < g.groups.size(); k++)
                items.push_back(ItemH{job->n, pr, (int)k, g.seg_k1[g.groups[k].second - 1] - g.seg_k0[g.groups[k].first]});
        for (int r = 0; r < g.p.M; r++)
            if (g.rows_m[(size_t)r] >= 0) rowmap.push_back(make_int2(job->n, r));      // (padding rows between clusters stay zero)
    }
    // Every `fine_every`-th tile pair is cut into one work item per K segment (a population: 64 ... 3 600 samples)
    // instead of runs of >= 2048 samples: sorted by length they end up last and fill the launch's final round, in which
    // the 1 024 workgroup slots otherwise finish up to one 0.75 ms item apart.  Same segments, same slabs: same bits.
    // Measured on the bench job (Gram kernel): none 37.43 ms; every 16th / 8th / 4th / 2nd pair 37.11 / 37.15 / 37.11 /
    // 37.10; every pair 37.08 ms with twice the work items (GAUSS_FINE_PAIRS, 0 = none).
    const int fine_every = job->n >= 4 ? env_int("GAUSS_FINE_PAIRS", 16) : 0;
    int pair_no = 0;
    for (int i = 0; i < job->n; i++) {
        const Prob& p = job->plans[i].p;
        const int mt_i = p.Mp / TILE;
        for (int pr = 0; pr < p.npair; pr++) {
            if (shm && job->plans[i].pair_ti[pr] < mt_i) continue;       // a B11 pair: done on the job-wide tiles
            if (fine_every > 0 && ++pair_no % fine_every == 0 && job->plans[i].groups.size() < job->plans[i].fine.size()) {
                for (size_t g = 0; g < job->plans[i].fine.size(); g++) {
                    const std::pair<int, int>& gr = job->plans[i].fine[g];
                    items.push_back(ItemH{i, pr, -1 - (int)g, job->plans[i].seg_k1[gr.second - 1] - job->plans[i].seg_k0[gr.first]});
                }
                continue;
            }
            for (size_t g = 0; g < job->plans[i].groups.size(); g++) {
                const std::pair<int, int>& gr = job->plans[i].groups[g];
                items.push_back(ItemH{i, pr, (int)g, job->plans[i].seg_k1[gr.second - 1] - job->plans[i].seg_k0[gr.first]});
            }
        }
        for (int r = shm ? p.M : 0; r < p.M + p.U; r++) rowmap.push_back(make_int2(i, r));
        if (shm) {
            // this window's view of the job-wide B11 pairs it lies in
            const Plan& g = *job->gplan;
            const int lo = job->g0[(size_t)i] / TILE, hi = (job->g0[(size_t)i] + p.M - 1) / TILE;
            for (int ti = lo; ti <= hi; ti++)
                for (int tj = ti; tj <= hi; tj++) {
                    const int2 e = make_int2(i, g.pair_lut[(size_t)ti * g.p.nT + tj] | TILE_GB11);
                    tilemap.push_back(e); job->win_tiles[(size_t)i].push_back(e);
                }
        }
        if (!p.n_gene)
            for (int pr = 0; pr < p.npair; pr++) {
                const bool b21 = !p.ld_only && job->plans[i].pair_ti[pr] >= p.Mp / TILE;      // a tile of U rows x M columns
                if (shm && !b21) continue;
                (b21 ? tilemap_b21 : tilemap).push_back(make_int2(i, pr));
                job->win_tiles[(size_t)i].push_back(make_int2(i, pr));
            }
        for (int pn = 0; pn < p.npi; pn++) panelmap.push_back(make_int2(i, pn));
        for (int pn = 0; pn < p.npanel; pn++) dpanelmap.push_back(make_int2(i, pn));
        if (p.npanel > 0) {
            job->max_nblk = std::max(job->max_nblk, p.nblk); job->max_npanel = std::max(job->max_npanel, p.npi);
            for (int up = 0; up < p.Up128 / job->gemm_ut; up++)
                for (int kb = 0; kb < (p.Mld + 127) / 128; kb++) gemmmap.push_back(make_int2(i, (up << 8) | kb));
            for (int c = 0; c < (p.n_rhs + 255) / 256; c++) finmap.push_back(make_int2(i, c));
        }
        if (p.mode != 0) job->max_pop = std::max(job->max_pop, p.P);
    }
    // longest segments first: the tail of the launch is then made of short items
    std::stable_sort(items.begin(), items.end(), [](const ItemH& a, const ItemH& b) { return a.len > b.len; });
    // B11's items of the job (job-wide pairs, or the windows' own measured x measured pairs)
    auto is_b11 = [&](const ItemH& h) {
        return h.prob == job->n || job->plans[(size_t)h.prob].pair_ti[(size_t)h.pair] < job->plans[(size_t)h.prob].p.Mp / TILE;
    };
    {
        // Chain beside the Gram kernel (k_solve_lite.hip): B11's items become a launch of their own, B11's epilogue tiles and
        // the factorisation chain follow it on the chain queue, and the chain's latency hides under the Gram launch of
        // B21's items.  Worth it when that launch is long enough to cover B11's small-footprint epilogue tiles (~0.5 ms) and
        // the chain, which runs ~3 x slower beside the Gram kernel than alone (~100 us per block step: two launches);
        // GAUSS_CHAIN_ASIDE = 0 never, 2 always (tests), 1 (default) by this estimate.
        const int mode = env_int("GAUSS_CHAIN_ASIDE", 1);
        bool genes = false;
        double b21_len = 0.0;
        for (const ItemH& h : items) if (!is_b11(h)) b21_len += (double)h.len;
        for (int i = 0; i < job->n; i++) genes = genes || job->plans[i].p.n_gene > 0;
        // (the int8 Gram kernel is ~8 x faster: an 8-rank share's B21 launch, 0.5 ms, no longer covers its chain)
        const double t_b21 = b21_len * 2.0 * TILE * TILE / (job->gram_i8 ? 960e12 : 120e12), t_chain = 0.5e-3 + 100e-6 * job->max_nblk;
        {
            double all_len = 0.0;
            for (const ItemH& h : items) all_len += (double)h.len;
            job->wait_bound_us = 50.0 * 1e6 * all_len * 2.0 * TILE * TILE / (job->gram_i8 ? 960e12 : 120e12);      // a whole-genome job must not trip a fixed bound
        }
        job->chain_aside = !streamed && mode != 0 && !panelmap.empty() && !tilemap_b21.empty() && !genes && job->ctx->chain &&
                           job->ctx->side && (mode == 2 || t_b21 >= 1.2 * t_chain);
        // (GAUSS_GRAM_SPLIT=1: the two-launch form of the Gram kernel without the chain beside it -- bench.py's one-stream pass,
        // which times the fp64 tails stand-alone, launches the Gram kernel the way the headline run does, so that a profile of the
        // whole command holds one kind of gram_kernel launch)
        const bool split_only = !streamed && mode != 0 && env_int("GAUSS_GRAM_SPLIT", 0) != 0 && !panelmap.empty() && !tilemap_b21.empty() && !genes;
        if (job->chain_aside || split_only) {
            std::stable_sort(items.begin(), items.end(), [&](const ItemH& a, const ItemH& b) { return is_b11(a) && !is_b11(b); });
            for (const ItemH& h : items) job->n_items_b11 += is_b11(h) ? 1 : 0;
        }
        // Merged launch (round 4): B11's items and B21's items are ONE launch, B11's first; they count themselves off and the
        // chain queue starts when the count is complete (k_gram.hip: wait_count_kernel) -- the chip is never drained between
        // the two halves (two launches: 37.1 ms, one: 36.6 on the 36-window job).  GAUSS_CHAIN_MERGED=0: two launches + event.
        // (f32 only: the int8 Gram kernel is operand-delivery bound and the chain's memory traffic beside ALL of it costs more than
        // the second launch's start-up -- measured 8.70 ms per step merged against 8.35 ms as two launches; =2 forces it for both)
        {
            const int mm = env_int("GAUSS_CHAIN_MERGED", 1);
            job->merged = job->chain_aside && mm != 0 && (!job->gram_i8 || mm == 2);
        }
        // Early epilogue: the smallest windows of the job that together hold about a third of B21's Gram work are "late" -- their
        // items end the launch -- and every other window is "early" (GAUSS_EPI_EARLY=0: off; GAUSS_EPI_LATE_PCT: the share.  Measured,
        // 36-window step / slowest 8-rank share: off 40.04 / 5.45 ms; 6 % 39.89 / 5.51; 12 % 39.76 / 5.51; 25 % 39.75 / 5.45; 35 %
        // 39.67 / 5.40; 50 % 39.64 / 5.47 -- a late part that is too small opens the gate only when the launch is all but over and
        // the two epilogue launches cost their event hops).  One window: nothing to split.
        late_window.assign((size_t)job->n + 1, 0);
        if (job->merged && job->n >= 2 && job->ctx->side && env_int("GAUSS_EPI_EARLY", 1) != 0) {
            std::vector<double> w21((size_t)job->n, 0.0);
            double tot = 0;
            for (const ItemH& h : items) if (!is_b11(h)) { w21[(size_t)h.prob] += h.len; tot += h.len; }
            const double want = tot * env_int("GAUSS_EPI_LATE_PCT", 35) / 100.0;
            double acc = 0;
            int n_late = 0;
            // (the windows with the least B21 work: a share of four windows gives up its smallest one, not whichever comes last)
            std::vector<int> by_work((size_t)job->n);
            for (int i = 0; i < job->n; i++) by_work[(size_t)i] = i;
            std::stable_sort(by_work.begin(), by_work.end(), [&](int a, int b) { return w21[(size_t)a] < w21[(size_t)b]; });
            for (int k = 0; k + 1 < job->n && acc < want; k++) { late_window[(size_t)by_work[(size_t)k]] = 1; acc += w21[(size_t)by_work[(size_t)k]]; n_late++; }
            if (n_late > 0 && acc < tot) {
                auto is_late = [&](const ItemH& h) { return !is_b11(h) && late_window[(size_t)h.prob] != 0; };
                std::stable_sort(items.begin() + job->n_items_b11, items.end(), [&](const ItemH& a, const ItemH& b) { return !is_late(a) && is_late(b); });
                for (size_t n = (size_t)job->n_items_b11; n < items.size(); n++) job->n_items_b21_early += is_late(items[n]) ? 0 : 1;
            } else late_window.assign((size_t)job->n + 1, 0);
        }
    }
    std::vector<int> sgroup_of_item;
    if (streamed) {
        // streamed window: B11's items (measured rows only) first, then B21's items by chunk of `ct` unmeasured row
        // tiles; each group is one Gram launch that starts as soon as its rows have landed
        const Plan& pl0 = job->plans[0];
        const int mt = pl0.p.Mp / TILE;
        const std::vector<int>& tile_group = stream->tile_group;
        const std::vector<int>& first_tile = stream->first_tile;
        const int ngrp = (int)first_tile.size() - 1;
        auto grp = [&](const ItemH& h) { const int ti = pl0.pair_ti[h.pair]; return ti < mt ? 0 : tile_group[(size_t)(ti - mt)]; };
        std::stable_sort(items.begin(), items.end(), [&](const ItemH& a, const ItemH& b) { return grp(a) < grp(b); });
        job->sgroups.assign((size_t)ngrp, gauss_job::StreamGroup{0, 0, 0, 0, 0, 0});
        for (size_t n = 0; n < items.size(); n++) {
            gauss_job::StreamGroup& g = job->sgroups[(size_t)grp(items[n])];
            if (g.n_items == 0) g.item0 = (int)n;
            g.n_items++;
        }
        job->sgroups[0].row0 = 0; job->sgroups[0].n_rows = pl0.p.M;
        for (int g = 1; g < ngrp; g++) {
            const int u0 = first_tile[(size_t)g] * TILE, u1 = std::min(pl0.p.U, first_tile[(size_t)g + 1] * TILE);
            job->sgroups[g].row0 = pl0.p.M + u0; job->sgroups[g].n_rows = std::max(0, u1 - u0);
            // B21's epilogue tiles are listed in pair order (row tile, then column tile): a chunk's tiles are contiguous
            job->sgroups[g].tile0 = first_tile[(size_t)g] * mt;
            job->sgroups[g].n_tiles = (first_tile[(size_t)g + 1] - first_tile[(size_t)g]) * mt;
        }
    }
    // XCD-aware launch order.  Workgroup b runs on XCD b % 8 (each XCD has its own 4 MiB L2).  Neighbours in
    // the sorted list share operand tiles (same window, same K range, adjacent tile pairs), so the list is
    // cut into super-blocks of xcd_block items and super-block j is queued on XCD j % 8: the items that are
    // resident together on one XCD then read the same tiles at about the same K position.  Measured on the
    // bench workload (rocprofv3 FETCH_SIZE): 22.0 GB -> 16.0 GB per launch at 36, same kernel time; larger
    // blocks start to cost time (load balance).  GAUSS_XCD_BLOCK overrides (0 = plain order).
    {
        // Small jobs (the 4-5 windows an 8-rank run leaves per GPU, ~7 000 items) take blocks of 8: a block of 36 is
        // 4 % of an XCD's share there, and the XCD that gets one more than the others finishes last (Gram kernel of
        // the 8-rank shares 5.10 -> 5.03 ms; 36 windows: 38.5 ms either way, but 36 reads 27 % less from the fabric).
        static const int xcd_env = [] { const char* e = getenv("GAUSS_XCD_BLOCK"); return e ? atoi(e) : -1; }();
        const int xcd_block = xcd_env >= 0 ? xcd_env : (items.size() >= 20000 ? 36 : 8);
        // (a job whose B11 items are a launch of their own interleaves each launch's list by itself)
        auto interleave = [&](size_t i0, size_t i1) {
            if (xcd_block <= 0 || i1 - i0 <= (size_t)8 * xcd_block) return;
            std::vector<std::vector<ItemH>> q(8);
            for (size_t i = i0; i < i1; i++) q[((i - i0) / xcd_block) % 8].push_back(items[i]);
            size_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n = i0;
            while (n < i1)
                for (int x = 0; x < 8; x++)
                    if (pos[x] < q[x].size()) items[n++] = q[x][pos[x]++];
        };
        if (!streamed) {
            interleave(0, (size_t)job->n_items_b11);
            if (job->n_items_b21_early > 0) {
                interleave((size_t)job->n_items_b11, (size_t)(job->n_items_b11 + job->n_items_b21_early));
                interleave((size_t)(job->n_items_b11 + job->n_items_b21_early), items.size());
            } else interleave((size_t)job->n_items_b11, items.size());
        }
    }
    const size_t o_items = ta.take(sizeof(Item) * std::max<size_t>(items.size(), 1));
    const size_t o_rowmap = put(blob, ta, rowmap);
    job->n_tiles_b11 = (int)tilemap.size();
    if (job->n_items_b21_early > 0) {
        std::stable_sort(tilemap_b21.begin(), tilemap_b21.end(), [&](const int2& a, const int2& b) { return !late_window[(size_t)a.x] && late_window[(size_t)b.x]; });
        for (const int2& t : tilemap_b21) job->n_tiles_b21_early += late_window[(size_t)t.x] ? 0 : 1;
    }
    tilemap.insert(tilemap.end(), tilemap_b21.begin(), tilemap_b21.end());
    const size_t o_tilemap = put(blob, ta, tilemap);
    // the product's tiles with the longest K loop (highest k block) first
    std::stable_sort(gemmmap.begin(), gemmmap.end(), [](const int2& a, const int2& b) { return (a.y & 255) > (b.y & 255); });
    const size_t o_panelmap = put(blob, ta, panelmap);
    const size_t o_dpanelmap = put(blob, ta, dpanelmap);
    const size_t o_gemmmap = put(blob, ta, gemmmap);
    const size_t o_finmap = put(blob, ta, finmap);
    const size_t o_probs = ta.take(sizeof(Prob) * (size_t)n_prob);
    blob.resize(ta.off);
    job->n_items = (int)items.size();
    job->n_rows = (int)rowmap.size();
    job->n_tiles = (int)tilemap.size();
    job->n_panels = (int)panelmap.size();
    job->n_dpanels = (int)dpanelmap.size();
    job->n_gemm = (int)gemmmap.size();
    job->n_fin = (int)finmap.size();

    // ---- workspace arena ----
    Arena wa;         // zeroed once per job: operand padding, B21 padding and the solve matrices rely on it
    Arena wslab;      // partial slabs: every entry a reader keeps is written by the Gram kernel first, so no zeroing
    struct WsOff { size_t raw_m, raw_u, packed, sx, sxx, slab, sd, wm, mu, wmu, A, Linv, B21, V, ld, b11c, gsum, part; long long ldraw; };
    std::vector<WsOff> wo(job->n);
    size_t res = 0;
    {
        // The early products of a row of the inverse (k_solve.hip, ride_pre) are a chain of dependent tile products:
        // rows with at least `solve_split` of them give one workgroup to each of their SOLVE_SPLIT classes, shorter rows
        // run the classes in one workgroup; 0 = never cut.  Either form sums in the same order.  Cutting from two
        // products on is what keeps every riding workgroup shorter than the diagonal tile's (36 windows, factorisation
        // with riding rows: never 1.72 ms, >= 8 1.70, >= 4 1.49, >= 2 1.46 before the pre / fin form, 1.31 with it; factorisation alone 1.10).
        const int thr = env_int("GAUSS_SOLVE_SPLIT_MIN", 2);               // read per job: tests drive both forms
        // few windows: every launch of the factorisation is a latency-bound link of a chain -- drop the panel launches
        // (k_solve.hip, factor_update_kernel own_panel); same bits either way.  Factorisation + riding rows, with /
        // without panel launches: 5 windows 0.65 / 0.58 ms, 9 windows 0.82 / 0.78, 18 windows 0.87 / 0.84, 36 windows
        // 1.31 / 1.42 (the repeated panel products start to cost workgroup slots)
        int n_solve = 0;
        for (int i = 0; i < job->n; i++) n_solve += job->plans[i].p.npanel > 0 ? 1 : 0;
        job->own_panel = (n_solve > 0 && n_solve <= env_int("GAUSS_OWN_PANEL_MAX_WINDOWS", 20)) ? 1 : 0;
        job->solve_split = job->n_panels > 0 ? thr : 0;
    }
    for (int i = 0; i < job->n; i++) {
        Plan& pl = job->plans[i];
        Prob& p = pl.p;
        WsOff& w = wo[i];
        if (streamed) {
            w.ldraw = stream->ldraw;                               // the rows land in the context's landing buffer
        } else if (!on_device) {
            // contiguous host matrices whose stride is close to the row length keep their stride on the device:
            // the upload is then ONE linear copy (a pitched copy of 3 000 rows runs at a fraction of that rate)
            const bool linear = pl.rows_m.empty() && pl.rows_u.empty() && (size_t)pl.user_ld <= pl.row_bytes + pl.row_bytes / 8 + 64;
            w.ldraw = linear ? pl.user_ld : (long long)rup(pl.row_bytes, 16);
            w.raw_m = wa.take((size_t)p.M * w.ldraw + 64);
            w.raw_u = wa.take((size_t)std::max(pl.U_user, 1) * w.ldraw + 64);
        } else {
            w.ldraw = pl.user_ld;
        }
        w.packed = wa.take((size_t)p.Sp * p.Kp);
        w.sx = wa.take((size_t)p.Sp * p.P * sizeof(int));
        w.sxx = wa.take((size_t)p.Sp * p.P * sizeof(int));
        w.slab = wslab.take((size_t)p.npair * p.nseg * TILE * TILE * (p.slab16 ? sizeof(uint16_t) : sizeof(float)));
        w.sd = wa.take((size_t)p.Sp * sizeof(double));
        w.wm = wa.take((size_t)p.Sp * sizeof(double));
        w.mu = wa.take((size_t)p.Sp * p.P * sizeof(double));
        w.wmu = wa.take((size_t)p.Sp * p.P * sizeof(double));
        if (!p.ld_only) {
            // the LD epilogue writes B11 (and its shifted twin) and B21 for every window that is not a plain
            // gauss_ld / gene batch; the factor and solve scratch only exists when there is something to solve
            w.A = wa.take((size_t)(p.npanel > 0 ? 5 : 2) * p.Mld * p.Mld * sizeof(double));   // A0 A1 [L0 L1 W0]
            w.B21 = wa.take((size_t)std::max(p.U, 1) * p.Mld * sizeof(double));
        }
        if (p.npanel > 0) {
            w.Linv = wa.take((size_t)2 * p.nblk * NB * NB * sizeof(double));
            w.V = wa.take((size_t)std::max(p.npanel, p.npi) * p.Mld * NR * sizeof(double));
            w.gsum = wa.take((size_t)((p.Mld + 127) / 128) * p.Up128 * 3 * sizeof(double));
            w.part = wa.take((size_t)2 * p.npi * 4 * NB * NR * sizeof(double));      // double buffered by row parity
            w.b11c = wa.take((size_t)p.Mld * p.Mld * sizeof(double));
        }
        w.ld = wa.take(std::max<size_t>(pl.out_ld_count, 1) * sizeof(double));
        pl.res_off = res;
        res += 2 * (size_t)p.n_rhs;
    }
    // job-wide measured rows (shared measured rows): one more tile of rows than Mp, because a window's last row tile
    // starts wherever the window starts and may reach past the chromosome's last measured SNP (zero rows there)
    struct GOff { size_t packed, sx, sxx, sd, wm, mu, wmu, slab; } go = {0, 0, 0, 0, 0, 0, 0, 0};
    if (shm) {
        const Prob& q = job->gplan->p;
        const size_t rows = (size_t)q.Mp + TILE;
        go.packed = wa.take(rows * q.Kp);
        go.sx = wa.take(rows * q.P * sizeof(int));
        go.sxx = wa.take(rows * q.P * sizeof(int));
        go.sd = wa.take(rows * sizeof(double));
        go.wm = wa.take(rows * sizeof(double));
        go.mu = wa.take(rows * q.P * sizeof(double));
        go.wmu = wa.take(rows * q.P * sizeof(double));
        go.slab = wslab.take((size_t)q.npair * q.nseg * TILE * TILE * (q.slab16 ? sizeof(uint16_t) : sizeof(float)));
    }
    const size_t o_status = wa.take(sizeof(int) * (4 * job->n + 4));    // [n][4], then 4 job-wide ints ([4 n]: the chain queue timed out)
    const size_t o_count = wa.take(128);                                 // counters of the merged Gram launch, a cache line each: [0] B11's items, [8] the early windows' B21 items (zeroed once; they only grow)
    const size_t o_results = wa.take(sizeof(double) * std::max<size_t>(res, 1));
    job->n_results = res;
    const size_t slab_base = rup(wa.off, 4096);
    for (WsOff& w : wo) w.slab += slab_base;
    go.slab += slab_base;
    job->ws_bytes = slab_base + wslab.off;
    job->tab_bytes = rup(blob.size(), 16);                               // (copied in 16-byte words, below)

    static const bool job_trace = getenv("GAUSS_JOB_TRACE") != nullptr;
    const auto tj0 = std::chrono::steady_clock::now();
    hipError_t e = ctx_dev_alloc(ctx, job->ws_bytes, (void**)&job->d_ws);
    if (e != hipSuccess) { job->d_ws = nullptr; return fail(GAUSS_E_NOMEM, "hipMalloc(%zu bytes workspace) failed: %s", wa.off, hipGetErrorString(e)); }
    e = ctx_dev_alloc(ctx, job->tab_bytes, (void**)&job->d_tab);
    if (e != hipSuccess) { job->d_tab = nullptr; return fail(GAUSS_E_NOMEM, "hipMalloc(%zu bytes tables) failed", blob.size()); }
    // one pinned block for the table image and the result mirrors: the table upload is then a true asynchronous
    // DMA and a job over resident rows is created without waiting for the stream (another job may be running on it)
    const size_t pin_tab = rup(job->tab_bytes, 256), pin_res = rup(sizeof(double) * std::max<size_t>(res, 1), 256);
    const size_t pin_st = rup(sizeof(int) * (4 * job->n + 4), 256);
    e = ctx_pin_alloc(ctx, pin_tab + 2 * (pin_res + pin_st), (void**)&job->h_pin);
    if (e != hipSuccess) { job->h_pin = nullptr; return fail(GAUSS_E_NOMEM, "hipHostMalloc(%zu bytes) failed", pin_tab + 2 * pin_res); }
    for (int k = 0; k < 2; k++) {
        job->h_res2[k] = (double*)(job->h_pin + pin_tab + k * (pin_res + pin_st));
        job->h_st2[k] = (int*)(job->h_pin + pin_tab + k * (pin_res + pin_st) + pin_res);
        HIPCHK(hipEventCreate(&job->done2[k]));
    }
    job->h_results = job->h_res2[0];
    job->h_status = job->h_st2[0];
    job->done = job->done2[0];
    const auto tj1 = std::chrono::steady_clock::now();
    HIPCHK(hipEventCreate(&job->begin));
    for (int k = 0; k < 2; k++)
        for (hipEvent_t* e : {&job->rev[k].gram, &job->rev[k].side, &job->rev[k].pack, &job->rev[k].rows, &job->rev[k].epi})
            HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));

    hipStream_t st = ctx->stream;
    // zero once: operand padding, B21 padding and the solve matrices rely on it.  On the (otherwise idle) upload queue, not on the
    // main queue: a pipeline creates the job of batch b + 1 while batch b computes, and 1-2 GB of zeroes at the head of the next
    // batch were a 0.3-0.4 ms gap between the batches of a chromosome (gauss_host_impute_chromosome: 36.4 ms of GPU span for
    // 35.0 ms of batches); beside the previous batch's Gram launch they cost nothing.  The main queue waits for the event IN ORDER,
    // i.e. behind whatever it is computing now.  While a background upload is using that queue the zeroes stay where they were.
    hipStream_t zs = st;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        if (ctx->upload && ctx->uploads.empty() && env_int("GAUSS_ZERO_ASIDE", 1) != 0) zs = ctx->upload;
    }
    HIPCHK(hipMemsetAsync(job->d_ws, 0, slab_base, zs));
    if (zs != st) {
        HIPCHK(hipEventCreateWithFlags(&job->zeroed, hipEventDisableTiming));
        HIPCHK(hipEventRecord(job->zeroed, zs));
        HIPCHK(hipStreamWaitEvent(st, job->zeroed, 0));
    }
    if (streamed) {
        job->sevp.resize(job->sgroups.size(), nullptr);
        for (hipEvent_t& e : job->sevp) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }

    job->d_status = (int*)(job->d_ws + o_status);
    job->d_b11_done = (unsigned long long*)(job->d_ws + o_count);
    job->d_results = (double*)(job->d_ws + o_results);
    std::deque<std::vector<uint8_t>> stage;      // host gather buffers, alive until the copies have drained
    for (int i = 0; i < job->n; i++) {
        Plan& pl = job->plans[i];
        Prob& p = pl.p;
        const WsOff& w = wo[i];
        char* T = job->d_tab;
        char* W = job->d_ws;
        p.ld_raw = w.ldraw;
        if (streamed) { p.raw_m = stream->d_m; p.raw_u = stream->d_u; }
        else if (on_device) { p.raw_m = pl.h_geno_m; p.raw_u = pl.h_geno_u; }
        else { p.raw_m = (const uint8_t*)(W + w.raw_m); p.raw_u = (const uint8_t*)(W + w.raw_u); }
        p.packed = (uint8_t*)(W + w.packed);
        p.sx = (int*)(W + w.sx); p.sxx = (int*)(W + w.sxx);
        p.pop_raw_off = (const int*)(T + to[i].raw_off);
        p.pop_pk_off = (const int*)(T + to[i].pk_off);
        p.pop_w = (const double*)(T + to[i].w);
        p.pop_wf = (const double*)(T + to[i].wf);
        p.pop_md = (const double*)(T + to[i].md);
        p.seg_pop = (const int*)(T + to[i].seg_pop);
        p.seg_k0 = (const int*)(T + to[i].k0);
        p.seg_k1 = (const int*)(T + to[i].k1);
        p.pop_seg0 = (const int*)(T + to[i].seg0);
        p.pair_ti = (const int*)(T + to[i].ti);
        p.pair_tj = (const int*)(T + to[i].tj);
        p.pair_lut = (const int*)(T + to[i].lut);
        p.word_pop = (const uint8_t*)(T + to[i].wp);
        p.word_run = (const uint8_t*)(T + to[i].wr);
        // row lists are resolved on the device only for a resident store; host rows are gathered while staging
        p.rows_m = (on_device && !pl.rows_m.empty()) ? (const int*)(T + to[i].rm) : nullptr;
        p.rows_u = (on_device && !pl.rows_u.empty()) ? (const int*)(T + to[i].ru) : nullptr;
        p.run_pk_off = (const int*)(T + to[i].rpk);
        p.run_src = (const int*)(T + to[i].rsrc);
        p.slab = (float*)(W + w.slab);
        p.rt_sd = (double*)(W + w.sd); p.rt_wm = (double*)(W + w.wm);
        p.rt_mu = (double*)(W + w.mu); p.rt_wmu = (double*)(W + w.wmu);
        p.z1 = (const double*)(T + to[i].z1);
        if (!p.ld_only) { p.A = (double*)(W + w.A); p.B21 = (double*)(W + w.B21); }
        if (p.npanel > 0) {
            p.Linv = (double*)(W + w.Linv); p.V = (double*)(W + w.V); p.Gsum = (double*)(W + w.gsum);
            p.Part = (double*)(W + w.part);
            pl.d_b11_copy = (double*)(W + w.b11c);
        }
        p.out_z = job->d_results + pl.res_off;
        p.out_info = job->d_results + pl.res_off + p.n_rhs;
        p.status = job->d_status + 4 * i;
        p.out_ld = (double*)(W + w.ld);
        p.gene_off = p.n_gene ? (const int*)(T + to[i].goff) : nullptr;
        p.gene_out_off = p.n_gene ? (long long*)(T + to[i].gout) : nullptr;
        // the unmeasured part of every row array follows the measured part ...
        p.packed_u = p.packed + (size_t)p.Mp * p.Kp;
        p.sx_u = p.sx + (size_t)p.Mp * p.P; p.sxx_u = p.sxx + (size_t)p.Mp * p.P;
        p.rt_sd_u = p.rt_sd + p.Mp; p.rt_wm_u = p.rt_wm + p.Mp;
        p.rt_mu_u = p.rt_mu + (size_t)p.Mp * p.P; p.rt_wmu_u = p.rt_wmu + (size_t)p.Mp * p.P;
        p.g0 = 0; p.n_gpair = 0; p.gpair_ti = p.pair_ti; p.gpair_tj = p.pair_tj; p.slab_g = p.slab;
        if (shm) {
            // ... unless the measured rows are the job-wide ones: this window's run starts at g0
            const Prob& q = job->gplan->p;
            const size_t g0 = (size_t)job->g0[(size_t)i];
            p.g0 = (int)g0; p.n_gpair = q.npair;
            p.packed = (uint8_t*)(W + go.packed) + g0 * q.Kp;
            p.sx = (int*)(W + go.sx) + g0 * q.P; p.sxx = (int*)(W + go.sxx) + g0 * q.P;
            p.rt_sd = (double*)(W + go.sd) + g0; p.rt_wm = (double*)(W + go.wm) + g0;
            p.rt_mu = (double*)(W + go.mu) + g0 * q.P; p.rt_wmu = (double*)(W + go.wmu) + g0 * q.P;
            p.gpair_ti = (const int*)(T + to[(size_t)job->n].ti); p.gpair_tj = (const int*)(T + to[(size_t)job->n].tj);
            p.slab_g = (float*)(W + go.slab);
        }
        memcpy(blob.data() + o_probs + sizeof(Prob) * i, &p, sizeof(Prob));
        if (!on_device && !streamed) {
            auto upload = [&](size_t dst_off, const uint8_t* src, const std::vector<int32_t>& rows, int nrows) -> int {
                if (nrows <= 0) return GAUSS_OK;
                if (rows.empty()) {
                    if (w.ldraw == pl.user_ld)      // same stride: one linear copy (the last row stops at its data)
                        HIPCHK(hipMemcpyAsync(W + dst_off, src, (size_t)(nrows - 1) * pl.user_ld + pl.row_bytes,
                                              hipMemcpyHostToDevice, st));
                    else
                        HIPCHK(hipMemcpy2DAsync(W + dst_off, (size_t)w.ldraw, src, (size_t)pl.user_ld, pl.row_bytes,
                                                (size_t)nrows, hipMemcpyHostToDevice, st));
                    return GAUSS_OK;
                }
                stage.emplace_back((size_t)nrows * w.ldraw);               // gather the listed store rows
                std::vector<uint8_t>& buf = stage.back();
                for (int r = 0; r < nrows; r++)
                    memcpy(buf.data() + (size_t)r * w.ldraw, src + (size_t)rows[r] * pl.user_ld, pl.row_bytes);
                HIPCHK(hipMemcpyAsync(W + dst_off, buf.data(), buf.size(), hipMemcpyHostToDevice, st));
                return GAUSS_OK;
            };
            int rc = upload(w.raw_m, pl.h_geno_m, pl.rows_m, p.M);
            if (!rc) rc = upload(w.raw_u, pl.h_geno_u, pl.rows_u, pl.U_user);
            if (rc) return rc;
        }
    }
    if (shm) {
        // descriptor n: the job-wide measured rows (pack_stats / row_stats work on it; nothing else is launched for it)
        Plan& g = *job->gplan;
        Prob& q = g.p;
        const size_t n = (size_t)job->n;
        char* T = job->d_tab;
        char* W = job->d_ws;
        q.ld_raw = g.user_ld;
        q.raw_m = g.h_geno_m; q.raw_u = nullptr;
        q.packed = (uint8_t*)(W + go.packed); q.packed_u = q.packed;
        q.sx = (int*)(W + go.sx); q.sxx = (int*)(W + go.sxx); q.sx_u = q.sx; q.sxx_u = q.sxx;
        q.rt_sd = (double*)(W + go.sd); q.rt_wm = (double*)(W + go.wm); q.rt_sd_u = q.rt_sd; q.rt_wm_u = q.rt_wm;
        q.rt_mu = (double*)(W + go.mu); q.rt_wmu = (double*)(W + go.wmu); q.rt_mu_u = q.rt_mu; q.rt_wmu_u = q.rt_wmu;
        q.pop_raw_off = (const int*)(T + to[n].raw_off); q.pop_pk_off = (const int*)(T + to[n].pk_off);
        q.pop_w = (const double*)(T + to[n].w); q.pop_wf = (const double*)(T + to[n].wf); q.pop_md = (const double*)(T + to[n].md);
        q.seg_pop = (const int*)(T + to[n].seg_pop); q.seg_k0 = (const int*)(T + to[n].k0); q.seg_k1 = (const int*)(T + to[n].k1);
        q.pop_seg0 = (const int*)(T + to[n].seg0);
        q.pair_ti = (const int*)(T + to[n].ti); q.pair_tj = (const int*)(T + to[n].tj); q.pair_lut = (const int*)(T + to[n].lut);
        q.word_pop = (const uint8_t*)(T + to[n].wp); q.word_run = (const uint8_t*)(T + to[n].wr);
        q.rows_m = (const int*)(T + to[n].rm); q.rows_u = nullptr;
        q.run_pk_off = (const int*)(T + to[n].rpk); q.run_src = (const int*)(T + to[n].rsrc);
        q.slab = (float*)(W + go.slab); q.slab_g = q.slab; q.gpair_ti = q.pair_ti; q.gpair_tj = q.pair_tj; q.g0 = 0; q.n_gpair = q.npair;
        q.z1 = nullptr; q.A = nullptr; q.B21 = nullptr; q.Linv = nullptr; q.V = nullptr; q.Gsum = nullptr; q.Part = nullptr;
        q.out_z = q.out_info = nullptr; q.out_ld = nullptr; q.status = job->d_status;      // never written for this descriptor
        q.gene_off = nullptr; q.gene_out_off = nullptr; q.n_gene = 0;
        memcpy(blob.data() + o_probs + sizeof(Prob) * n, &q, sizeof(Prob));
    }
    // device work items: every pointer is resolved here so the kernel starts loading operands at once
    for (size_t n = 0; n < items.size(); n++) {
        const ItemH& h = items[n];
        const Plan& pl = plan_of(h.prob);
        const Prob& p = pl.p;
        const std::pair<int, int>& gr = h.group >= 0 ? pl.groups[h.group] : pl.fine[(size_t)(-1 - h.group)];
        const int ti = pl.pair_ti[h.pair], tj = pl.pair_tj[h.pair];
        const int mt = p.Mp / TILE;
        auto rows = [&](int t) {
            if (!pl.tile_live.empty()) return pl.tile_live[(size_t)t];
            int left = (t < mt) ? p.M - t * TILE : p.U - (t - mt) * TILE; return left > TILE ? TILE : left;
        };
        // a row tile of the measured part (for a window that shares its measured rows: inside the job-wide array, from
        // wherever the window starts) or of the unmeasured part
        auto tile_rows = [&](int t) { return t < mt ? p.packed + (size_t)t * TILE * p.Kp : p.packed_u + (size_t)(t - mt) * TILE * p.Kp; };
        Item it;
        it.a = tile_rows(ti);
        it.b = tile_rows(tj);
        it.slab = p.slab + ((size_t)h.pair * p.nseg + gr.first) * (p.slab16 ? TILE * TILE / 2 : TILE * TILE);
        it.seg_k1 = p.seg_k1 + gr.first;
        it.chunk_live = (const uint32_t*)(job->d_tab + to[h.prob].ch);
        it.Kp = p.Kp; it.k0 = pl.seg_k0[gr.first]; it.nseg = gr.second - gr.first;
        static const int no_edge16 = env_int("GAUSS_GRAM_EDGE16", 1) == 0 ? 8 : 0;
        it.rows_a = rows(ti); it.rows_b = rows(tj); it.flags = (ti == tj ? 1 : 0) | (p.slab16 ? 2 : 0) | no_edge16;
        if (job->merged && (int)n < job->n_items_b11) it.flags |= 16;         // counts itself off in b11_done[0]
        else if (job->merged && (int)n < job->n_items_b11 + job->n_items_b21_early) it.flags |= 32;      // ... in b11_done[8] (the early windows' B21 items)
        memcpy(blob.data() + o_items + sizeof(Item) * n, &it, sizeof(Item));
    }
    const auto tj2 = std::chrono::steady_clock::now();
    memcpy(job->h_pin, blob.data(), blob.size());
    // The table image crosses PCIe by kernel, not by hipMemcpyAsync: while a background upload keeps the DMA engines busy
    // (gauss_store_upload_async: a chromosome's first call) the runtime made the CALLER wait for an engine -- one job creation in
    // five stalled for 11-16 ms on these few MB (GAUSS_JOB_TRACE=1, round 4), the GPU idle meanwhile.
    launch_h2d_copy(job->d_tab, job->h_pin, job->tab_bytes, st);
    HIPCHK(hipGetLastError());
    if (job_trace) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[job] %d windows: plan %.2f ms, allocations %.2f ms (workspace %.1f MB, tables %.2f MB, pinned %.2f MB), events + zeroing + tables %.2f ms, table copy queued in %.2f ms\n",
                job->n, ms(tb0, tj0), ms(tj0, tj1), job->ws_bytes / 1e6, job->tab_bytes / 1e6, (pin_tab + 2 * (pin_res + pin_st)) / 1e6, ms(tj1, tj2),
                ms(tj2, std::chrono::steady_clock::now()));
    }
    job->d_probs = (Prob*)(job->d_tab + o_probs);
    job->d_items = (Item*)(job->d_tab + o_items);
    job->d_rowmap = (int2*)(job->d_tab + o_rowmap);
    job->d_tilemap = (int2*)(job->d_tab + o_tilemap);
    job->d_panelmap = (int2*)(job->d_tab + o_panelmap);
    job->d_dpanelmap = (int2*)(job->d_tab + o_dpanelmap);
    job->d_gemmmap = (int2*)(job->d_tab + o_gemmmap);
    job->d_finmap = (int2*)(job->d_tab + o_finmap);
    if (!on_device && !streamed) HIPCHK(hipStreamSynchronize(st));   // uploads from pageable user memory are complete
    std::vector<char>().swap(job->h_tab);
    { std::lock_guard<std::mutex> lock(ctx->mu); ctx->jobs.insert(job); }
    *out = guard.release();
    return GAUSS_OK;
}

// ------------------------------------------------------------------------------------------
// profiling helpers
// ------------------------------------------------------------------------------------------
static void prof_begin(gauss_job* job, int kernel, hipStream_t st, int launches = 1)
{
    if (!job->prof) return;
    ProfSlot s;
    s.kernel = kernel;
    s.launches = launches;
    s.run = job->run_seq;
    hipEventCreate(&s.a);
    hipEventCreate(&s.b);
    hipEventRecord(s.a, st);
    job->slots.push_back(s);
}
static void prof_end(gauss_job* job, hipStream_t st)
{
    if (!job->prof) return;
    hipEventRecord(job->slots.back().b, st);
}
// Collects the stage timers of the runs before `run_end` (default: all).  gauss_job_fetch passes the run it has just
// fetched: with two runs in flight the later run's events are still pending, and waiting for them here would make
// the fetch of run k block until run k + 1 has finished -- the host's share of a step would no longer overlap GPU work.
static void prof_collect(gauss_job* job, unsigned run_end = ~0u)
{
    std::vector<ProfSlot> keep;
    for (ProfSlot& s : job->slots) {
        if (run_end != ~0u && (int)(s.run - run_end) >= 0) { keep.push_back(s); continue; }
        hipEventSynchronize(s.b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, s.a, s.b);
        job->prof_ms[s.kernel] += ms;
        job->prof_n[s.kernel] += s.launches;
        hipEventDestroy(s.a);
        hipEventDestroy(s.b);
    }
    job->slots.swap(keep);
}

// ------------------------------------------------------------------------------------------
static int job_run_finish(gauss_job* job, hipStream_t st)
{
    HIPCHK(hipGetLastError());
    // the result mirrors travel with the run, so that gauss_job_fetch waits for THIS job only (an event), not for
    // whatever else has been queued on the stream since (the next job of a pipeline)
    const int par = (int)(job->run_seq & 1u);
    static const bool job_trace = getenv("GAUSS_JOB_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    // By kernel into the pinned mirrors, not by hipMemcpyAsync: beside a background upload (a chromosome's first call) one run in
    // eight made the caller wait 10-18 ms for a DMA engine here, and every run of a 36-window job 5 ms (GAUSS_JOB_TRACE=1, round 4).
    // Both mirrors have room for the 16-byte word the copy rounds up to (pin_res in job_build; the status block is 16 (n + 1) bytes);
    // the event below is a system-scope release, so the host reads what the kernel wrote.
    if (job->n_results) launch_h2d_copy(job->h_res2[par], job->d_results, rup(sizeof(double) * job->n_results, 16), st);
    launch_h2d_copy(job->h_st2[par], job->d_status, sizeof(int) * (4 * job->n + 4), st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(job->done2[par], st));
    if (job_trace) fprintf(stderr, "[job] run: result copies queued in %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    job->done = job->done2[par];
    job->run_seq++;
    job->ran = true;
    return GAUSS_OK;
}

static int job_run(gauss_job* job, bool solve)
{
    hipStream_t st = job->ctx->stream;
    HIPCHK(hipSetDevice(job->ctx->device));
    if (job->run_seq - job->fetch_seq >= 2u)
        return fail(GAUSS_E_INVALID, "gauss_job_run: two runs of this job are in flight already; fetch one first");
    const gauss_job::RunEvents& ev = job->rev[job->run_seq & 1u];      // this run's cross-queue events (the parity's own set)
    const auto t_run0 = std::chrono::steady_clock::now();
    HIPCHK(hipEventRecord(job->begin, st));
    HIPCHK(hipMemsetAsync(job->d_status, 0, sizeof(int) * (4 * job->n + 4), st));
    if (getenv("GAUSS_JOB_TRACE")) fprintf(stderr, "[job] run: begin mark + status zeroing queued in %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run0).count());
    // fused tail: the factorisation chain needs B11 only and the closing product is the first reader of B21, so B21's
    // tiles of the epilogue (85 % of them) go to the side stream and run beside the chain; the row tables and the
    // certificate, which only the epilogue and the factorisation read, go there too and slip in while the Gram kernel
    // starts up
    const bool fused = env_int("GAUSS_FUSED_SOLVE", 1) != 0;           // read per run: the tests drive both forms
    // (off by default: worth 0.04 ms on an 8-rank share, nothing on 36 windows -- the Gram kernel gets that much slower --
    // and one profiled launch in a hundred saw the two small kernels and the Gram kernel's start tangle for 28 ms)
    const bool rows_aside = env_int("GAUSS_ROWS_ASIDE", 0) != 0;        // read per run: the tests drive both forms
    hipStream_t side = (solve && fused && job->n_panels > 0 && job->n_tiles > job->n_tiles_b11) ? job->ctx->side : nullptr;
    prof_begin(job, 1, st);
    launch_pack_stats(job->d_probs, job->d_rowmap, job->n_rows, st);
    hipStream_t rs = (side && rows_aside) ? side : st;
    if (rs != st) {
        HIPCHK(hipEventRecord(ev.pack, st));
        HIPCHK(hipStreamWaitEvent(rs, ev.pack, 0));
    }
    // the certificate needs the row tables only and is read by B11's epilogue tiles and the chain: in a merged launch it
    // moves to the head of the chain queue, beside the Gram kernel's start (23 us off the main queue's critical path)
    const bool cert_on_chain = solve && job->chain_aside && job->merged && rs == st;
    if (solve && job->n_panels > 0 && !cert_on_chain) launch_shift_cert(job->d_probs, job->n, rs);
    if (rs != st) HIPCHK(hipEventRecord(ev.rows, rs));
    prof_end(job, st);
    if (solve && job->chain_aside) {
        // Chain beside the Gram kernel.
        //   main:   Gram(B11's items) -> Gram(B21's items) -> B21's epilogue tiles -> closing product.  (The second launch queued
        //           "any order" through hipExtLaunchKernelGGL, with the first launch's own completion event for the chain queue,
        //           starts 5 us earlier and gains 0.08 ms of 41 -- the first launch's last workgroups are dispatched at its very
        //           end -- not worth a launch path of its own);
        //   chain:  from the moment the FIRST launch has finished: B11's epilogue tiles, then the whole
        //           factorisation with the riding rows of the inverse, all in small-footprint form (k_pack_epilogue.hip
        //           epilogue_b11_lite_kernel, k_solve_lite.hip): their workgroups fit into what the Gram kernel's four workgroups
        //           per CU leave free, so the ~19 dependent block steps run UNDER the second Gram launch instead of behind it.
        // Same arithmetic, same bits as the path below.
        hipStream_t ch = job->ctx->chain;
        if (job->merged) {
            // ONE launch: B11's items first (they count themselves off in d_b11_done), B21's items behind them in the same grid.
            // The chain queue joins the main queue right BEFORE the launch (operands, row tables and certificate are complete)
            // and then waits for the count of this run: the counter only grows, run r is complete at (r + 1) x n_items_b11.
            if (rs != st) HIPCHK(hipStreamWaitEvent(st, ev.rows, 0));
            HIPCHK(hipEventRecord(ev.gram, st));
            prof_begin(job, 0, st, 1);
            launch_gram(job->d_items, job->n_items, job->gram_i8, st, job->d_b11_done);
            job->merged_runs++;                            // counted per LAUNCH, not per completed call: a later error must not shift the target
            prof_end(job, st);
            HIPCHK(hipStreamWaitEvent(ch, ev.gram, 0));
            if (cert_on_chain) launch_shift_cert(job->d_probs, job->n, ch);
            launch_wait_count(job->d_b11_done, job->merged_runs * (unsigned long long)job->n_items_b11, job->d_status + 4 * job->n, 1, ch, job->wait_bound_us);
        } else {
        prof_begin(job, 0, st, 2);
        launch_gram(job->d_items, job->n_items_b11, job->gram_i8, st);
        HIPCHK(hipEventRecord(ev.gram, st));
        launch_gram(job->d_items + job->n_items_b11, job->n_items - job->n_items_b11, job->gram_i8, st);
        prof_end(job, st);
        if (rs != st) {
            // GAUSS_ROWS_ASIDE: the row tables and the certificate were made on the side queue; everything that reads them
            // joins it here -- B11's epilogue tiles and the chain (status[3]) on the chain queue, B21's epilogue tiles on the
            // main queue (behind both Gram launches, which read neither)
            HIPCHK(hipStreamWaitEvent(ch, ev.rows, 0));
            HIPCHK(hipStreamWaitEvent(st, ev.rows, 0));
        }
        HIPCHK(hipStreamWaitEvent(ch, ev.gram, 0));
        }
        prof_begin(job, 2, ch);
        launch_epilogue_b11_lite(job->d_probs, job->d_tilemap, job->n_tiles_b11, job->gram_i8, ch);
        prof_end(job, ch);
        for (int i = 0; i < job->n; i++) {
            Plan& pl = job->plans[i];
            if (pl.out_b11 && pl.p.npanel > 0)
                HIPCHK(hipMemcpyAsync(pl.d_b11_copy, pl.p.A, sizeof(double) * pl.p.Mld * pl.p.Mld, hipMemcpyDeviceToDevice, ch));
        }
        prof_begin(job, 3, ch);
        for (int s = 0; s < job->max_nblk; s++)
            launch_factor_step_lite(job->d_probs, job->n, s, job->max_nblk, job->max_npanel, job->solve_split, ch);
        launch_solve_last_lite(job->d_probs, job->d_panelmap, job->n_panels, job->max_nblk, job->solve_split, ch);
        prof_end(job, ch);
        HIPCHK(hipEventRecord(ev.side, ch));
        // (B21's tiles read neither B11 nor the certificate -- epilogue_tile looks at status[3] for B11's tiles only -- so they
        // need not wait for the chain queue; the closing product does)
        if (job->merged && job->n_tiles_b21_early > 0) {
            // Early epilogue: the tiles of the windows whose B21 items are done before the launch's last round go to the LOW-priority
            // queue behind a wait for their count.  The hardware hands a lower-priority queue's workgroups out when the Gram grid
            // has none left to dispatch: they run in the slots the launch's last round leaves idle (measured: 0.16 ms of the
            // 36-window step, 0.09 ms of an 8-rank share's).  The late windows' tiles follow the launch on the main queue.
            hipStream_t lo = job->ctx->side;
            HIPCHK(hipStreamWaitEvent(lo, ev.gram, 0));
            launch_wait_count(job->d_b11_done + 8, job->merged_runs * (unsigned long long)job->n_items_b21_early, job->d_status + 4 * job->n, 1, lo, job->wait_bound_us);
            launch_epilogue(job->d_probs, job->d_tilemap + job->n_tiles_b11, job->n_tiles_b21_early, job->max_pop, job->gram_i8, lo);
            HIPCHK(hipEventRecord(ev.epi, lo));
            prof_begin(job, 2, st);
            launch_epilogue(job->d_probs, job->d_tilemap + job->n_tiles_b11 + job->n_tiles_b21_early,
                            job->n_tiles - job->n_tiles_b11 - job->n_tiles_b21_early, job->max_pop, job->gram_i8, st);
            prof_end(job, st);
            HIPCHK(hipStreamWaitEvent(st, ev.epi, 0));
        } else {
        prof_begin(job, 2, st);
        launch_epilogue(job->d_probs, job->d_tilemap + job->n_tiles_b11, job->n_tiles - job->n_tiles_b11, job->max_pop, job->gram_i8, st);
        prof_end(job, st);
        }
        HIPCHK(hipStreamWaitEvent(st, ev.side, 0));
        prof_begin(job, 4, st);
        launch_impute_gemm(job->d_probs, job->d_gemmmap, job->n_gemm, job->gemm_ut, job->d_finmap, job->n_fin, st);
        prof_end(job, st);
        return job_run_finish(job, st);
    }
    prof_begin(job, 0, st, job->n_items_b11 > 0 ? 2 : 1);
    launch_gram(job->d_items, job->n_items_b11, job->gram_i8, st);             // (none unless GAUSS_GRAM_SPLIT: B11's items first)
    launch_gram(job->d_items + job->n_items_b11, job->n_items - job->n_items_b11, job->gram_i8, st);
    prof_end(job, st);
    if (rs != st) HIPCHK(hipStreamWaitEvent(st, ev.rows, 0));
    if (side) {
        HIPCHK(hipEventRecord(ev.gram, st));
        HIPCHK(hipStreamWaitEvent(side, ev.gram, 0));
        launch_epilogue(job->d_probs, job->d_tilemap + job->n_tiles_b11, job->n_tiles - job->n_tiles_b11, job->max_pop, job->gram_i8, side);
        HIPCHK(hipEventRecord(ev.side, side));
    }
    prof_begin(job, 2, st);
    launch_epilogue(job->d_probs, job->d_tilemap, side ? job->n_tiles_b11 : job->n_tiles, job->max_pop, job->gram_i8, st);
    for (int i = 0; i < job->n; i++)
        if (job->plans[i].p.n_gene) launch_gene_epilogue(job->d_probs, i, job->plans[i].p.n_gene, st);
    prof_end(job, st);
    if (solve && job->n_panels > 0) {
        for (int i = 0; i < job->n; i++) {
            Plan& pl = job->plans[i];
            if (pl.out_b11 && pl.p.npanel > 0)
                HIPCHK(hipMemcpyAsync(pl.d_b11_copy, pl.p.A, sizeof(double) * pl.p.Mld * pl.p.Mld, hipMemcpyDeviceToDevice, st));
        }
        // fused (default): the rows of [X | y] = L^-1 [I | z1] ride in the factorisation's update launches and the
        // closing product forms z / info (k_solve.hip); the stage timers read "factor" = factorisation + riding rows,
        // "solve" = closing row + product + finish
        prof_begin(job, 3, st);
        for (int s = 0; s < job->max_nblk; s++)
            launch_factor_step(job->d_probs, job->n, s, job->max_nblk, fused ? job->max_npanel : 0, job->solve_split,
                               job->own_panel, st);
        prof_end(job, st);
        prof_begin(job, 4, st);
        if (fused) {
            launch_solve_last(job->d_probs, job->d_panelmap, job->n_panels, job->max_nblk, job->solve_split, st);
            if (side) HIPCHK(hipStreamWaitEvent(st, ev.side, 0));
            launch_impute_gemm(job->d_probs, job->d_gemmmap, job->n_gemm, job->gemm_ut, job->d_finmap, job->n_fin, st);
        } else launch_solve(job->d_probs, job->d_dpanelmap, job->n_dpanels, st);
        prof_end(job, st);
    }
    return job_run_finish(job, st);
}

// One window whose genotype rows are still in HOST memory (the blocking call the Rcpp drivers bind).  Four queues:
//   copy   the rows travel chunk by chunk -- measured rows first, then the unmeasured rows a few row tiles at a time --
//          into the context's landing buffer; issued by the context's copy worker (stream_start_copies), which starts
//          BEFORE the window is planned: a copy from pageable memory (an Rcpp driver's std::vector) returns only when
//          the runtime has staged the bytes, and the PCIe link must wait neither for the planner nor for launches
//   aux    pack + row tables of a chunk as soon as it has landed (and the certificate after the measured rows)
//   main   the Gram launches, one per chunk, back to back: they are what bounds the compute side
//   chain  B11's epilogue tiles and the whole factorisation chain (it needs B11 only, i.e. the first Gram launch):
//          latency-bound launches that slip in between the chunks' Gram launches
// then B21's epilogue tiles, the closing product and the results on the main stream.  Same kernels on the same data as
// job_run: the same bits.
static int ctx_stream_init(gauss_ctx* ctx)
{
    if (ctx->copy) return GAUSS_OK;
    int lo = 0, hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithPriority(&ctx->aux, hipStreamNonBlocking, hi));
    if (!ctx->chain) HIPCHK(hipStreamCreateWithPriority(&ctx->chain, hipStreamNonBlocking, hi));
    ctx->worker = new CopyWorker(ctx->device);
    return GAUSS_OK;
}

// Chunk table, landing buffer and events of a streamed window; then the copy worker is set going.
static int stream_start_copies(gauss_ctx* ctx, const gauss_window_desc& win, size_t row_bytes, StreamSetup& su)
{
    int rc = ctx_stream_init(ctx);
    if (rc) return rc;
    su.M = win.n_measured; su.U = win.n_unmeasured; su.row_bytes = row_bytes;
    const bool linear = (size_t)win.ld <= row_bytes + row_bytes / 8 + 64;        // same rule as job_build: one linear copy per chunk
    su.ldraw = linear ? win.ld : (long long)rup(row_bytes, 16);
    // Chunks of `ct` row tiles (the last `lt` tiles may form a closing chunk of their own).  Measured on a mean chr22
    // window (M = 736, U = 2526, 105 MB of genotype bytes, 20 row tiles; tools/window_trace.py, medians of interleaved
    // calls): 4 to 8 tiles per chunk 2.77-2.83 ms, 3 tiles 3.4 ms (every chunk pays under-filled launches and two event
    // hops of ~50 us), a closing chunk of 1 or 2 tiles +0.06-0.1 ms; upload-then-run 3.83 ms
    const int ct = std::max(1, env_int("GAUSS_STREAM_CHUNK_TILES", 6));
    const int n_ut = (su.U + TILE - 1) / TILE;
    const int lt = std::max(0, std::min(env_int("GAUSS_STREAM_LAST_TILES", 0), n_ut - 1));
    su.tile_group.assign((size_t)std::max(n_ut, 1), 1);
    su.first_tile = {0, 0};
    for (int t = 0, g = 1; t < n_ut; g++) {
        const int body = n_ut - lt - t;
        const int sz = body > 0 ? std::min(ct, body) : n_ut - t;
        for (int k = 0; k < sz; k++) su.tile_group[(size_t)(t + k)] = g;
        t += sz;
        su.first_tile.push_back(t);
    }
    const int ng = su.n_groups();
    const size_t need = (size_t)(su.M + su.U) * (size_t)su.ldraw + 256;
    if (ctx->landing_bytes < need) {
        if (ctx->landing) { HIPCHK(hipStreamSynchronize(ctx->copy)); HIPCHK(hipFree(ctx->landing)); ctx->landing = nullptr; ctx->landing_bytes = 0; }
        const size_t want = need + need / 4;
        void* d = nullptr;
        hipError_t e = ctx_malloc_retry(ctx, &d, want);
        if (e != hipSuccess) return fail(GAUSS_E_NOMEM, "hipMalloc(%zu bytes landing buffer) failed: %s", want, hipGetErrorString(e));
        ctx->landing = (uint8_t*)d; ctx->landing_bytes = want;
    }
    su.d_m = ctx->landing;
    su.d_u = ctx->landing + rup((size_t)su.M * (size_t)su.ldraw + 64, 256);
    while ((int)ctx->ev_pool.size() < ng) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->ev_pool.push_back(e);
    }
    su.ev.assign(ctx->ev_pool.begin(), ctx->ev_pool.begin() + ng);
    hipStream_t cs = ctx->copy;
    const uint8_t* hm = win.geno_m;
    const uint8_t* hu = win.geno_u;
    const long long user_ld = win.ld;
    StreamSetup* sp = &su;
    ctx->worker->submit([sp, cs, hm, hu, user_ld, ng]() {
        StreamSetup& su = *sp;
        for (int g = 0; g < ng; g++) {
            const int r0 = g == 0 ? 0 : su.first_tile[(size_t)g] * TILE;
            const int r1 = g == 0 ? su.M : std::min(su.U, su.first_tile[(size_t)g + 1] * TILE);
            const int nrows = r1 - r0;
            uint8_t* dst = (g == 0 ? su.d_m : su.d_u) + (size_t)r0 * su.ldraw;
            const uint8_t* src = (g == 0 ? hm : hu) + (size_t)r0 * user_ld;
            hipError_t e = hipSuccess;
            if (nrows > 0) {
                if (su.ldraw == user_ld)
                    e = hipMemcpyAsync(dst, src, (size_t)(nrows - 1) * user_ld + su.row_bytes, hipMemcpyHostToDevice, cs);
                else
                    e = hipMemcpy2DAsync(dst, (size_t)su.ldraw, src, (size_t)user_ld, su.row_bytes, (size_t)nrows, hipMemcpyHostToDevice, cs);
            }
            if (e == hipSuccess) e = hipEventRecord(su.ev[(size_t)g], cs);
            std::lock_guard<std::mutex> lock(su.mu);
            if (e != hipSuccess) {
                su.rc = GAUSS_E_DEVICE;
                su.err = std::string("streamed window: copy of a chunk failed: ") + hipGetErrorString(e);
                su.recorded = ng;
                su.cv.notify_all();
                return;
            }
            su.recorded = g + 1;
            su.cv.notify_all();
        }
    });
    return GAUSS_OK;
}

static int job_run_streamed(gauss_job* job, StreamSetup& su)
{
    gauss_ctx* ctx = job->ctx;
    hipStream_t st = ctx->stream;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t ax = ctx->aux;
    const bool chain_aside = env_int("GAUSS_STREAM_CHAIN_ASIDE", 1) != 0;      // read per call: tools/window_trace.py A/B
    const bool pack_aside = env_int("GAUSS_STREAM_PACK_ASIDE", 1) != 0;
    hipStream_t ch = chain_aside ? ctx->chain : st;
    Plan& pl = job->plans[0];
    const Prob& p = pl.p;
    const size_t ng = job->sgroups.size();
    const gauss_job::RunEvents& ev = job->rev[job->run_seq & 1u];
    HIPCHK(hipEventRecord(job->begin, st));
    HIPCHK(hipMemsetAsync(job->d_status, 0, sizeof(int) * (4 * job->n + 4), st));
    // what job_build queued on the main stream (zeroing, tables) comes before anything on the other queues
    HIPCHK(hipEventRecord(ev.pack, st));
    HIPCHK(hipStreamWaitEvent(ax, ev.pack, 0));
    for (size_t g = 0; g < ng; g++) {
        const gauss_job::StreamGroup& sg = job->sgroups[g];
        {
            // a stream can only wait for an event that HAS been recorded: take chunk g up once the worker has queued
            // "chunk g has landed" behind its copy
            std::unique_lock<std::mutex> lock(su.mu);
            su.cv.wait(lock, [&] { return su.recorded > (int)g; });
            if (su.rc) return fail(su.rc, "%s", su.err.c_str());
        }
        // pack + row tables on a stream of their own: the pack kernel shares the chip with the previous chunk's Gram
        // launch (measured: 2.86 ms per call against 3.07 with pack in front of the Gram launch on the main stream)
        hipStream_t ps = pack_aside ? ax : st;
        HIPCHK(hipStreamWaitEvent(ps, su.ev[g], 0));
        launch_pack_stats(job->d_probs, job->d_rowmap + sg.row0, sg.n_rows, ps);
        if (g == 0) launch_shift_cert(job->d_probs, job->n, ps);
        if (ps != st) {
            HIPCHK(hipEventRecord(job->sevp[g], ps));
            HIPCHK(hipStreamWaitEvent(st, job->sevp[g], 0));
        }
        launch_gram(job->d_items + sg.item0, sg.n_items, job->gram_i8, st);
        if (g == 0) {
            if (ch != st) {
                HIPCHK(hipEventRecord(ev.gram, st));
                HIPCHK(hipStreamWaitEvent(ch, ev.gram, 0));
            }
            launch_epilogue(job->d_probs, job->d_tilemap, job->n_tiles_b11, job->max_pop, job->gram_i8, ch);
            if (pl.out_b11)
                HIPCHK(hipMemcpyAsync(pl.d_b11_copy, p.A, sizeof(double) * p.Mld * p.Mld, hipMemcpyDeviceToDevice, ch));
            for (int s = 0; s < job->max_nblk; s++)
                launch_factor_step(job->d_probs, job->n, s, job->max_nblk, job->max_npanel, job->solve_split, job->own_panel, ch);
            launch_solve_last(job->d_probs, job->d_panelmap, job->n_panels, job->max_nblk, job->solve_split, ch);
            if (ch != st) HIPCHK(hipEventRecord(ev.side, ch));
        }
    }
    launch_epilogue(job->d_probs, job->d_tilemap + job->n_tiles_b11, job->n_tiles - job->n_tiles_b11, job->max_pop, job->gram_i8, st);
    if (ch != st) HIPCHK(hipStreamWaitEvent(st, ev.side, 0));
    launch_impute_gemm(job->d_probs, job->d_gemmmap, job->n_gemm, job->gemm_ut, job->d_finmap, job->n_fin, st);
    HIPCHK(hipGetLastError());
    const int par = (int)(job->run_seq & 1u);
    static const bool job_trace = getenv("GAUSS_JOB_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    // By kernel into the pinned mirrors, not by hipMemcpyAsync: beside a background upload (a chromosome's first call) one run in
    // eight made the caller wait 10-18 ms for a DMA engine here, and every run of a 36-window job 5 ms (GAUSS_JOB_TRACE=1, round 4).
    // Both mirrors have room for the 16-byte word the copy rounds up to (pin_res in job_build; the status block is 16 (n + 1) bytes);
    // the event below is a system-scope release, so the host reads what the kernel wrote.
    if (job->n_results) launch_h2d_copy(job->h_res2[par], job->d_results, rup(sizeof(double) * job->n_results, 16), st);
    launch_h2d_copy(job->h_st2[par], job->d_status, sizeof(int) * (4 * job->n + 4), st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(job->done2[par], st));
    if (job_trace) fprintf(stderr, "[job] run: result copies queued in %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    job->done = job->done2[par];
    job->run_seq++;
    job->ran = true;
    return GAUSS_OK;
}

// Rare path: MakePosDef would have modified B11 (util.cpp:310-317).  Rebuild B11 from the
// epilogue, clamp its spectrum on the device (Jacobi), refactor and re-solve this window alone.
static int job_clamp_window(gauss_job* job, int i, int* status_bits)
{
    hipStream_t st = job->ctx->stream;
    Plan& pl = job->plans[i];
    Prob& p = pl.p;
    // Re-run the epilogue for this problem only to restore A[0] (the factorisation overwrote it)
    std::vector<int2> tm;
    tm = job->win_tiles[(size_t)i];                   // this window's epilogue tiles (job-wide B11 pairs included)
    DevBuf d_tm, d_work, d_pm;
    HIPCHK(d_tm.alloc(job->ctx, sizeof(int2) * tm.size()));
    HIPCHK(hipMemcpyAsync(d_tm.p, tm.data(), sizeof(int2) * tm.size(), hipMemcpyHostToDevice, st));
    launch_epilogue(job->d_probs, d_tm.as<int2>(), (int)tm.size(), job->max_pop, job->gram_i8, st);
    HIPCHK(hipGetLastError());
    const size_t n = (size_t)p.Mld;
    HIPCHK(d_work.alloc(job->ctx, sizeof(double) * (2 * n * n + 4 * n)));
    HIPCHK(hipMemsetAsync(p.status, 0, sizeof(int) * 4, st));
    launch_jacobi_clamp(job->d_probs, i, p, d_work.as<double>(), true, st);
    HIPCHK(hipGetLastError());
    // refactor (both matrices are factored again; only matrix 0 is used) and solve this window
    std::vector<int2> pm;
    for (int pn = 0; pn < p.npanel; pn++) pm.push_back(make_int2(i, pn));
    HIPCHK(d_pm.alloc(job->ctx, sizeof(int2) * pm.size()));
    HIPCHK(hipMemcpyAsync(d_pm.p, pm.data(), sizeof(int2) * pm.size(), hipMemcpyHostToDevice, st));
    if (pl.out_b11) HIPCHK(hipMemcpyAsync(pl.d_b11_copy, p.A, sizeof(double) * n * n, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(p.A + 4 * n * n, p.A, sizeof(double) * n * n, hipMemcpyDeviceToDevice, st));   // W0 = clamped B11
    for (int s = 0; s < p.nblk; s++) {
        // launch over all problems would redo the others; use a single-problem launch instead
        launch_factor_step(job->d_probs + i, 1, s, p.nblk, 0, 0, 0, st);
    }
    launch_solve(job->d_probs, d_pm.as<int2>(), (int)pm.size(), st);
    HIPCHK(hipGetLastError());
    int h_status[4];
    HIPCHK(hipMemcpyAsync(h_status, p.status, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(job->h_results + pl.res_off, job->d_results + pl.res_off, sizeof(double) * 2 * p.n_rhs, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *status_bits = (h_status[2] || h_status[0]) ? GAUSS_ST_NONFINITE : GAUSS_ST_CLAMPED;
    return GAUSS_OK;
}

// Device matrix [rows x pitch] -> host [rows x width] doubles.  One linear copy into a staging buffer and a
// row-wise compaction on the host: a pitched device-to-host copy of a few thousand rows is many times slower.
static int fetch_matrix(double* dst, const double* d_src, int rows, int width, int pitch)
{
    if (rows <= 0 || width <= 0) return GAUSS_OK;
    if (pitch == width) { HIPCHK(hipMemcpy(dst, d_src, sizeof(double) * (size_t)rows * width, hipMemcpyDeviceToHost)); return GAUSS_OK; }
    std::vector<double> tmp((size_t)(rows - 1) * pitch + width);
    HIPCHK(hipMemcpy(tmp.data(), d_src, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
    for (int r = 0; r < rows; r++) memcpy(dst + (size_t)r * width, tmp.data() + (size_t)r * pitch, sizeof(double) * width);
    return GAUSS_OK;
}

// CountPC (util.cpp:355-388) when the smallest eigenvalue of B11 is below the cutoff: eigenvalues by the
// device Jacobi sweep, counted on the host (the matrix itself is left alone).
static int job_count_small_eigs(gauss_job* job, int i, int* num_eig)
{
    hipStream_t st = job->ctx->stream;
    Plan& pl = job->plans[i];
    Prob& p = pl.p;
    const size_t n = (size_t)p.Mld;
    DevBuf d_work;
    HIPCHK(d_work.alloc(job->ctx, sizeof(double) * (2 * n * n + 4 * n)));
    launch_jacobi_clamp(job->d_probs, i, p, d_work.as<double>(), false, st);
    HIPCHK(hipGetLastError());
    std::vector<double> delta(n);
    HIPCHK(hipMemcpyAsync(delta.data(), d_work.as<double>() + 2 * n * n, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int small = 0;
    for (size_t k = 0; k < n; k++) if (delta[k] > 0.0) small++;
    *num_eig = p.M - small;
    return GAUSS_OK;
}

static int job_fetch(gauss_job* job)
{
    if (!job->ran || job->fetch_seq == job->run_seq) return fail(GAUSS_E_INVALID, "gauss_job_fetch: no run of this job is waiting to be fetched");
    hipStream_t st = job->ctx->stream;
    HIPCHK(hipSetDevice(job->ctx->device));
    const int par = (int)(job->fetch_seq & 1u);
    job->h_results = job->h_res2[par];
    job->h_status = job->h_st2[par];
    HIPCHK(hipEventSynchronize(job->done2[par]));
    if (job->h_status[4 * job->n] != 0) {
        job->fetch_seq++;
        return fail(GAUSS_E_DEVICE, "the chain queue gave up waiting for B11's tile pairs of this run (merged Gram launch): results are not valid");
    }
    // With a later run of the job already queued, anything that reads the job's DEVICE buffers (matrix exports, the
    // clamp path, the eigenvalue count) first lets that run finish: the job's inputs do not change between runs, so
    // what it leaves on the device is what the fetched run left.
    if (job->run_seq - job->fetch_seq > 1u) {
        bool device_reads = false;
        for (int i = 0; i < job->n && !device_reads; i++) {
            const Plan& pl = job->plans[i];
            device_reads = pl.out_b11 || pl.out_b21 || (pl.out_ld_user && pl.out_ld_count) ||
                           job->h_status[4 * i + 0] || job->h_status[4 * i + 1];
        }
        if (device_reads) HIPCHK(hipEventSynchronize(job->done));
    }
    job->fetch_seq++;
    for (int i = 0; i < job->n; i++) {
        Plan& pl = job->plans[i];
        const Prob& p = pl.p;
        int bits = 0;
        if (p.kind == GAUSS_WIN_LD) {
            // raw LD export: B11 sits unfactored in A[0] (diagonal 1 + lambda), B21 in its buffer
            if (pl.out_b11)
                { int rc2 = fetch_matrix(pl.out_b11, p.A, p.M, p.M, p.Mld); if (rc2) return rc2; }
            if (pl.out_b21 && p.U > 0)
                { int rc2 = fetch_matrix(pl.out_b21, p.B21, p.U, p.M, p.Mld); if (rc2) return rc2; }
            if (pl.out_status) *pl.out_status = 0;
            continue;
        }
        if (p.kind == GAUSS_WIN_QCAT) {
            // QCAT never repairs B11 (MakePosDef is commented out, qcat.cpp:206); CountPC only counts
            int num_eig = p.M;
            if (job->h_status[4 * i + 0]) bits = GAUSS_ST_NONFINITE;          // B11 has no Cholesky factor
            // (no factor: B11 is indefinite -- weights summing far above 1 -- or not finite.  The reference still counts: CountPC runs
            // before the factorisation, qcat.cpp:203, and an eigenvalue below the cutoff, negative ones included, is not counted)
            if (job->h_status[4 * i + 0] || job->h_status[4 * i + 1]) { int rc = job_count_small_eigs(job, i, &num_eig); if (rc) return rc; }
            if (bits & GAUSS_ST_NONFINITE)
                for (int u = 0; u < 2 * p.n_rhs; u++) job->h_results[pl.res_off + u] = NAN;
            if (pl.out_r) memcpy(pl.out_r, job->h_results + pl.res_off, sizeof(double) * p.n_rhs);
            if (pl.out_num_eig) *pl.out_num_eig = num_eig;
            if (pl.out_status) *pl.out_status = bits;
            if (pl.out_b11)
                { int rc2 = fetch_matrix(pl.out_b11, pl.d_b11_copy, p.M, p.M, p.Mld); if (rc2) return rc2; }
            if (pl.out_b21 && p.U > 0)
                { int rc2 = fetch_matrix(pl.out_b21, p.B21, p.U, p.M, p.Mld); if (rc2) return rc2; }
            continue;
        }
        if (p.npanel > 0 && (job->h_status[4 * i + 0] || job->h_status[4 * i + 1])) {
            int rc = job_clamp_window(job, i, &bits);
            if (rc) return rc;
        }
        if (p.npanel > 0) {
            if (bits & GAUSS_ST_NONFINITE) {
                // the reference's eigen-solver / LU propagate non-finite values to every output
                for (int u = 0; u < 2 * p.U; u++) job->h_results[pl.res_off + u] = NAN;
            }
            if (pl.out_z) memcpy(pl.out_z, job->h_results + pl.res_off, sizeof(double) * p.U);
            if (pl.out_info) memcpy(pl.out_info, job->h_results + pl.res_off + p.U, sizeof(double) * p.U);
            if (pl.out_b11)
                { int rc2 = fetch_matrix(pl.out_b11, pl.d_b11_copy, p.M, p.M, p.Mld); if (rc2) return rc2; }
            if (pl.out_b21 && p.U > 0)
                { int rc2 = fetch_matrix(pl.out_b21, p.B21, p.U, p.M, p.Mld); if (rc2) return rc2; }
        }
        if (pl.out_status) *pl.out_status = bits;
        if (pl.out_ld_user && pl.out_ld_count)
            HIPCHK(hipMemcpy(pl.out_ld_user, p.out_ld, sizeof(double) * pl.out_ld_count, hipMemcpyDeviceToHost));
    }
    if (job->prof) prof_collect(job, job->fetch_seq);      // the slots of the run just fetched (fetch_seq already counts it)
    return GAUSS_OK;
}

// Everything a job holds on its context: waits for its queued runs, then gives the blocks back and destroys the events.
// Called by job_free, and by gauss_hip_destroy for the jobs that outlive their context (the context is still whole then).
static void job_release(gauss_job* job)
{
    gauss_ctx* ctx = job->ctx;
    if (!ctx) return;
    hipSetDevice(ctx->device);
    // runs that were queued and never fetched: their result copies target this job's pinned block
    if (job->run_seq != job->fetch_seq && job->done) (void)hipEventSynchronize(job->done);
    // a streamed window that failed half way may still have row copies in flight towards its workspace
    if (!job->sgroups.empty())
        for (hipStream_t q : {ctx->aux, ctx->chain}) if (q) (void)hipStreamSynchronize(q);
    for (ProfSlot& s : job->slots) { hipEventDestroy(s.a); hipEventDestroy(s.b); }
    job->slots.clear();
    ctx_dev_release(ctx, job->d_ws);
    ctx_dev_release(ctx, job->d_tab);
    ctx_pin_release(ctx, job->h_pin);
    job->d_ws = nullptr; job->d_tab = nullptr; job->h_pin = nullptr;
    if (job->begin) hipEventDestroy(job->begin);
    if (job->zeroed) { hipEventDestroy(job->zeroed); job->zeroed = nullptr; }
    for (int k = 0; k < 2; k++) if (job->done2[k]) hipEventDestroy(job->done2[k]);
    for (int k = 0; k < 2; k++)
        for (hipEvent_t* e : {&job->rev[k].gram, &job->rev[k].side, &job->rev[k].pack, &job->rev[k].rows, &job->rev[k].epi})
            if (*e) { hipEventDestroy(*e); *e = nullptr; }
    for (hipEvent_t e : job->sevp) if (e) hipEventDestroy(e);
    job->sevp.clear();
    job->begin = job->done = nullptr;
    job->done2[0] = job->done2[1] = nullptr;
    { std::lock_guard<std::mutex> lock(ctx->mu); ctx->jobs.erase(job); }
    job->ctx = nullptr;                       // from here on the handle is an orphan: only gauss_job_destroy accepts it
}

static void job_free(gauss_job* job)
{
    if (!job) return;
    job_release(job);
    delete job;
}

// entry points that need the job's context
#define JOB_ALIVE(job)                                                                                          \
    do {                                                                                                        \
        if (!(job)) return fail(GAUSS_E_INVALID, "job is NULL");                                                \
        if (!(job)->ctx) return fail(GAUSS_E_INVALID, "the job's context has been destroyed (gauss_hip_destroy)"); \
    } while (0)

static WinSpec spec_from_desc(const gauss_window_desc& d)
{
    WinSpec w;
    w.mode = d.mode; w.n_pop = d.n_pop; w.pop_off = d.pop_off; w.pop_wgt = d.pop_wgt;
    w.M = d.n_measured; w.U = d.n_unmeasured; w.geno_m = d.geno_m; w.geno_u = d.geno_u; w.ld = d.ld;
    w.z1 = d.z1; w.lambda = d.lambda; w.eps = d.min_abs_eig; w.diag = 1.0; w.ld_only = 0;
    w.gene_off = nullptr; w.n_gene = 0;
    w.kind = d.kind; w.n_head = d.n_head_measured; w.n_predm = d.n_pred_measured; w.eig_cutoff = d.eig_cutoff;
    w.u_codings = d.u_codings;
    w.geno_fmt = d.geno_format; w.rows_m = d.rows_m; w.rows_u = d.rows_u; w.pop_src_off = d.pop_src_off;
    return w;
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char* gauss_last_error(void) { return g_err.c_str(); }
const char* gauss_hip_version(void) { return "gauss_hip 0.1 (gfx950)"; }

int gauss_hip_init(int device, gauss_ctx** out_ctx)
{
    if (!out_ctx) return fail(GAUSS_E_INVALID, "out_ctx is NULL");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(GAUSS_E_INVALID, "device %d out of range (have %d)", device, n);
    HIPCHK(hipSetDevice(device));
    gauss_ctx* c = new gauss_ctx();
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
    const int cu_main = env_int("GAUSS_CU_MASK_MAIN", 0);
    if (cu_main > 0) {
        // experiment (tools/experiments/README.md, round 3): every launch of the context confined to the first `cu_main`
        // CUs (hipExtStreamCreateWithCUMask) -- what the Gram kernel loses when CUs are set aside for a tail stream,
        // and what the factorisation chain takes on a handful of CUs.  One stream; never the default.
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        const int total = prop.multiProcessorCount;
        std::vector<uint32_t> mask((size_t)(total + 31) / 32, 0u);
        // mask bit i is CU i / 8 of XCD i % 8 (measured: clearing bits 0, 32, 64, ... slows the Gram kernel by a third --
        // one XCD short of 8 CUs): the first n bits are n / 8 CUs of every XCD
        for (int cu = 0; cu < std::min(cu_main, total); cu++) mask[(size_t)cu / 32] |= 1u << (cu % 32);
        HIPCHK(hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)mask.size(), mask.data()));
    } else if (env_int("GAUSS_SIDE_STREAM", 1)) {
        // the chain gets the higher priority: a workgroup of its next launch must win the CU an epilogue workgroup frees
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
        HIPCHK(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, lo));
        // the queue of the factorisation chain when it runs beside the Gram kernel (job_run, k_solve_lite.hip)
        HIPCHK(hipStreamCreateWithPriority(&c->chain, hipStreamNonBlocking, hi));
    } else HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    // What a session's FIRST row-store upload would otherwise pay in front of its first byte (measured on MI355X, round 4): the
    // queue of its own that piecewise / background uploads travel on (creating a stream: ~15 ms the first time) and the two
    // pinned staging buffers (hipHostMalloc of 2 x 32 MB: 4-14 ms).  The queue is made here; the buffers are made in the
    // background and parked in the context's pinned-block cache.  GAUSS_PREPIN=0: both on first use.
    if (env_int("GAUSS_PREPIN", 1) != 0) {
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&c->upload, hipStreamNonBlocking, hi));       // see gauss_store_fill
        c->prepin = std::thread([c]() {
            (void)hipSetDevice(c->device);
            {
                // the upload queue's first KERNEL (job_build zeroes new workspaces there; background fills copy by kernel) makes
                // the runtime set up its compute queue: 100 ms when it happened in the middle of a chromosome's first call
                // (GAUSS_CHROM_TRACE: one job creation of 103 ms, round 4) -- done here, off everybody's path
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
    }
    *out_ctx = c;
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
    if (ctx->prepin.joinable()) ctx->prepin.join();
    // 1. whoever cached something per context (the host layer's resident panels) lets go of it
    std::vector<std::pair<void (*)(gauss_ctx*, uint64_t, void*), void*>> hooks;
    { std::lock_guard<std::mutex> lock(g_hook_mu); hooks = g_destroy_hooks; }
    for (auto& h : hooks) h.first(ctx, ctx->id, h.second);
    // 2. jobs that outlive the context: wait for their work, release what they hold, leave empty shells behind
    std::vector<gauss_job*> live;
    { std::lock_guard<std::mutex> lock(ctx->mu); live.assign(ctx->jobs.begin(), ctx->jobs.end()); }
    for (gauss_job* j : live) job_release(j);
    hipStreamSynchronize(ctx->stream);
    if (ctx->side) { hipStreamSynchronize(ctx->side); hipStreamDestroy(ctx->side); }
    delete ctx->worker;
    for (hipStream_t* q : {&ctx->copy, &ctx->aux, &ctx->chain})
        if (*q) { hipStreamSynchronize(*q); hipStreamDestroy(*q); }
    for (hipEvent_t e : ctx->ev_pool) hipEventDestroy(e);
    if (ctx->landing) (void)hipFree(ctx->landing);
    // 3. row stores nobody freed (uploads still running are finished first)
    for (auto& kv : ctx->uploads) kv.second->finish();
    ctx->uploads.clear();
    if (ctx->upload) { hipStreamSynchronize(ctx->upload); hipStreamDestroy(ctx->upload); }
    for (auto& kv : ctx->stores) (void)hipFree(const_cast<void*>(kv.first));
    ctx->stores.clear();
    for (auto& kv : ctx->dev_cache.free_blocks) (void)hipFree(kv.second);
    for (auto& kv : ctx->pin_cache.free_blocks) (void)hipHostFree(kv.second);
    hipStreamDestroy(ctx->stream);
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
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (out_bytes_freed) *out_bytes_freed = (int64_t)ctx->dev_cache.held;
    ctx_flush_dev_cache_locked(ctx);
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
    if (ctx->prepin.joinable()) {                          // the staging buffers made at init are in the pinned cache (or about to be)
        std::lock_guard<std::mutex> lock(ctx->prepin_mu);
        if (ctx->prepin.joinable()) ctx->prepin.join();
    }

    if (bytes < 2 * CH && !chunk_queued && src.ptr && stream == ctx->stream) { HIPCHK(hipMemcpy(d, src.ptr, bytes, hipMemcpyHostToDevice)); return GAUSS_OK; }
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    int rc = GAUSS_OK;
    auto cleanup = [&]() {
        for (int b = 0; b < 2; b++) { if (ev[b]) hipEventDestroy(ev[b]); ctx_pin_release(ctx, pin[b]); }
    };
    const bool trace = getenv("GAUSS_UPLOAD_TRACE") != nullptr;
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
    const int nt_env = env_int("GAUSS_UPLOAD_THREADS", 0);
    // one process per GPU (torchrun exports LOCAL_WORLD_SIZE): the ranks of a node share its cores
    const unsigned ranks = (unsigned)std::max(1, env_int("LOCAL_WORLD_SIZE", 1));
    const int nt = nt_env > 0 ? nt_env : (int)std::max(1u, std::min(chunk_queued ? 4u : 8u, (hw ? hw / 2 : 2u) / ranks));
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
        if (!ctx->upload) {
            // at least the main queue's priority: a lower one is not dispatched while the Gram grid has workgroups left
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            hipError_t e = hipStreamCreateWithPriority(&ctx->upload, hipStreamNonBlocking, hi);
            if (e != hipSuccess) return fail(GAUSS_E_DEVICE, "gauss_store_fill: %s", hipGetErrorString(e));
        }
    }
    RowSource2 src = src0;
    if (src.ptr) src.ptr += offset; else src.file_off += offset;
    static const bool by_kernel = env_int("GAUSS_FILL_BY_KERNEL", 1) != 0;
    return upload_rows(ctx, (uint8_t*)device_ptr + offset, src, (size_t)len, ctx->upload, nullptr, by_kernel);
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
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        if (!ctx->upload) {
            int lo = 0, hi = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            HIPCHK(hipStreamCreateWithPriority(&ctx->upload, hipStreamNonBlocking, hi));       // see gauss_store_fill
        }
    }
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
        if (it == ctx->uploads.end()) return GAUSS_OK;            // not an asynchronous store, or complete and retired
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

int gauss_hip_set_gram_dtype(gauss_ctx* ctx, int dtype)
{
    if (!ctx || (dtype != GAUSS_GRAM_F32 && dtype != GAUSS_GRAM_I8)) return fail(GAUSS_E_INVALID, "bad gram dtype %d", dtype);
    ctx->gram_i8 = (dtype == GAUSS_GRAM_I8);
    return GAUSS_OK;
}

int gauss_job_create(gauss_ctx* ctx, const gauss_window_desc* wins, int n_win, int on_device, gauss_job** out_job)
{
    if (!ctx || !wins || n_win < 1 || !out_job) return fail(GAUSS_E_INVALID, "bad arguments to gauss_job_create");
    std::vector<WinSpec> specs;
    for (int i = 0; i < n_win; i++) {
        if (wins[i].n_unmeasured < 1 && wins[i].kind != GAUSS_WIN_LD &&
            !(wins[i].kind == GAUSS_WIN_QCAT && wins[i].n_pred_measured > 0))
            return fail(GAUSS_E_INVALID, "window %d has no unmeasured SNPs", i);
        specs.push_back(spec_from_desc(wins[i]));
    }
    gauss_job* job = nullptr;
    int rc = job_build(ctx, specs, on_device, &job);
    if (rc) return rc;
    for (int i = 0; i < n_win; i++) {
        Plan& pl = job->plans[i];
        pl.out_z = wins[i].out_z; pl.out_info = wins[i].out_info; pl.out_status = wins[i].out_status;
        pl.out_b11 = wins[i].out_b11; pl.out_b21 = wins[i].out_b21;
        pl.out_r = wins[i].out_r; pl.out_num_eig = wins[i].out_num_eig;
    }
    *out_job = job;
    return GAUSS_OK;
}

int gauss_job_run(gauss_job* job) { JOB_ALIVE(job); return job_run(job, true); }
int gauss_job_fetch(gauss_job* job) { JOB_ALIVE(job); return job_fetch(job); }
void gauss_job_destroy(gauss_job* job) { job_free(job); }

int gauss_job_span_ms(gauss_job* first, gauss_job* last, double* out_ms)
{
    JOB_ALIVE(first); JOB_ALIVE(last);
    if (!out_ms || !first->ran || !last->ran) return fail(GAUSS_E_INVALID, "gauss_job_span_ms: both jobs must have run");
    HIPCHK(hipEventSynchronize(last->done));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, first->begin, last->done));
    *out_ms = (double)ms;
    return GAUSS_OK;
}

int gauss_job_profile(gauss_job* job, int enable)
{
    JOB_ALIVE(job);
    if (job->prof && !enable) prof_collect(job);
    job->prof = enable != 0;
    if (enable) { for (int k = 0; k < 5; k++) { job->prof_ms[k] = 0; job->prof_n[k] = 0; } }
    return GAUSS_OK;
}

int gauss_job_profile_get(gauss_job* job, int kernel, double* out_ms, int64_t* out_launches)
{
    JOB_ALIVE(job);
    if (kernel < 0 || kernel > 4) return fail(GAUSS_E_INVALID, "bad arguments");
    hipStreamSynchronize(job->ctx->stream);
    prof_collect(job);
    if (out_ms) *out_ms = job->prof_ms[kernel];
    if (out_launches) *out_launches = job->prof_n[kernel];
    return GAUSS_OK;
}

int gauss_job_work(gauss_job* job, double* out_ld_flops, double* out_solve_flops, double* out_bytes, int64_t* out_imputed)
{
    if (!job) return fail(GAUSS_E_INVALID, "job is NULL");
    double ldf = 0, sf = 0, by = 0;
    int64_t imp = 0;
    for (const Plan& pl : job->plans) {
        const double M = pl.p.M, U = pl.p.U, N = pl.p.N;
        ldf += N * M * (M + 1) + 2.0 * N * U * M;            // symmetric half of B11 + B21
        sf += M * M * M / 3.0 + 2.0 * U * M * M + 4.0 * U * M;
        by += (M + U) * N + (M * M + U * M) * 8.0;
        imp += pl.p.U;
    }
    if (out_ld_flops) *out_ld_flops = ldf;
    if (out_solve_flops) *out_solve_flops = sf;
    if (out_bytes) *out_bytes = by;
    if (out_imputed) *out_imputed = imp;
    return GAUSS_OK;
}

int gauss_job_stats(gauss_job* job, double* out4)
{
    if (!job || !out4) return fail(GAUSS_E_INVALID, "bad arguments");
    double flops = 0, slab = 0;
    const bool shm = job->gplan != nullptr;
    const bool edge16 = env_int("GAUSS_GRAM_EDGE16", 1) != 0;
    auto add = [&](const Plan& pl, bool skip_b11) {
        const Prob& p = pl.p;
        const int mt = p.Mp / TILE;
        auto rows = [&](int t) {
            if (!pl.tile_live.empty()) return pl.tile_live[(size_t)t];
            int left = (t < mt) ? p.M - t * TILE : p.U - (t - mt) * TILE; return left > TILE ? TILE : left;
        };
        auto halves = [](int r, int w) { int n = (r - w * 64 + 31) / 32; return n < 0 ? 0 : (n > 2 ? 2 : n); };
        // samples per row the kernel multiplies: every zero-padded block (population, or 2-bit source block) rounded up to
        // the units the K loop can skip -- 8 samples on the f32 path (Item::chunk_live), 32 on the int8 path (whole groups);
        // the 16-column edge routine takes every chunk whole
        double k_main = 0;
        {
            const int gran = job->gram_i8 ? 32 : 8;
            const std::vector<int>& blk = (p.geno_fmt == GAUSS_GENO_2BIT && !pl.run_pk_off.empty()) ? pl.run_pk_off : pl.pop_pk_off;
            const bool runs = &blk == &pl.run_pk_off;
            for (size_t q = 0; q + 1 < blk.size(); q++) {
                // live samples of the block: its real size where known (populations), else its padded size
                int live = blk[q + 1] - blk[q];
                if (!runs && q + 1 < pl.pop_raw_off.size()) live = pl.pop_raw_off[q + 1] - pl.pop_raw_off[q];
                else if (runs && pl.run_len_known(q)) live = pl.run_len[q];
                k_main += (double)((live + gran - 1) / gran * gran);
            }
        }
        int pairs = 0;
        for (int pr = 0; pr < p.npair; pr++) {
            const int ti = pl.pair_ti[pr], tj = pl.pair_tj[pr];
            if (skip_b11 && ti < mt) continue;               // multiplied once, on the job-wide tiles
            pairs++;
            double tiles32 = 0, tiles_edge = 0;
            for (int wr = 0; wr < 2; wr++)
                for (int wc = 0; wc < 2; wc++) {
                    if (ti == tj && wr == 1 && wc == 0) continue;
                    const int na = halves(rows(ti), wr);
                    // f32 path: a wave whose last live 32-column half holds at most 16 live columns multiplies 16-column groups
                    // (k_gram.hip, chunk_mfma_edge): 1 or 3 of them
                    int nb16 = (rows(tj) - wc * 64 + 15) / 16;
                    nb16 = nb16 < 0 ? 0 : (nb16 > 4 ? 4 : nb16);
                    if (edge16 && !job->gram_i8 && na > 0 && (nb16 & 1)) { tiles_edge += na * nb16 * 0.5; continue; }
                    double t32 = na * halves(rows(tj), wc);
                    if (ti == tj && wr == wc && t32 == 4) t32 = 3;      // mirrored 32 x 32 sub-block of a diagonal quadrant
                    tiles32 += t32;
                }
            flops += 32.0 * 32.0 * 2.0 * (tiles32 * k_main + tiles_edge * p.Kp);
        }
        slab += (double)pairs * p.nseg * TILE * TILE * (p.slab16 ? sizeof(uint16_t) : sizeof(float));
    };
    for (const Plan& pl : job->plans) add(pl, shm);
    if (shm) add(*job->gplan, false);
    out4[0] = job->n_items; out4[1] = flops; out4[2] = slab; out4[3] = (double)job->ws_bytes;
    return GAUSS_OK;
}

int gauss_impute_window(gauss_ctx* ctx, const gauss_window_desc* win)
{
    if (!ctx || !win) return fail(GAUSS_E_INVALID, "bad arguments to gauss_impute_window");
    // Streamed form (default): upload and compute overlap (job_run_streamed).  It covers the windows the drivers make --
    // contiguous host matrices, additive coding, something to solve; the clamp path re-reads the job's buffers and works
    // on either form.  GAUSS_STREAM_WINDOW=0 (or GAUSS_FUSED_SOLVE=0): upload everything, then run.
    const bool fused = env_int("GAUSS_FUSED_SOLVE", 1) != 0;           // read per run: the tests drive both forms
    bool streamed = env_int("GAUSS_STREAM_WINDOW", 1) != 0 && fused && !win->rows_m && !win->rows_u && win->n_unmeasured >= 1 &&
                    win->n_measured >= 1 && win->kind != GAUSS_WIN_LD && (win->u_codings & ~GAUSS_CODE_ADDITIVE) == 0 &&
                    win->geno_m && win->geno_u && win->pop_off && win->n_pop >= 1 && win->n_pop <= 64;
    // bytes of a source row (what plan_problem will find; anything odd is left to the unstreamed path and its messages)
    size_t row_bytes = 0;
    if (streamed) {
        const int N = win->pop_off[win->n_pop];
        if (win->geno_format == GAUSS_GENO_U8) {
            streamed = N >= 1 && win->ld >= N;
            row_bytes = (size_t)std::max(N, 0);
        } else if (win->geno_format == GAUSS_GENO_2BIT && win->ld % 16 == 0) {
            long long end = 0;
            for (int q = 0; q < win->n_pop && streamed; q++) {
                const long long blk = (long long)rup((size_t)std::max(win->pop_off[q + 1] - win->pop_off[q], 0), 64) / 4;
                const long long off = win->pop_src_off ? win->pop_src_off[q] : end;
                if (off < 0 || off % 16 || off + blk > win->ld) streamed = false;
                if (!win->pop_src_off) end = off + blk;
                row_bytes = std::max(row_bytes, (size_t)(off + blk));
            }
        } else streamed = false;
    }
    gauss_job* job = nullptr;
    int rc;
    const bool trace = env_int("GAUSS_STREAM_TRACE", 0) != 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (trace) fprintf(stderr, "[stream] %s at %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    if (streamed) {
        std::lock_guard<std::mutex> one(ctx->stream_mu);
        HIPCHK(hipSetDevice(ctx->device));
        StreamSetup su;
        rc = stream_start_copies(ctx, *win, row_bytes, su);
        if (rc) return rc;
        // On every path out of here the worker has finished its task (`su` lives on this stack) AND the copies it queued
        // have left the caller's matrices: from pinned memory they are true asynchronous DMAs, and the caller may free
        // the matrices the moment this call returns.  A successful fetch has waited for them already (`landed`).
        struct Waiter {
            gauss_ctx* c; bool landed = false;
            ~Waiter() { c->worker->wait(); if (!landed) (void)hipStreamSynchronize(c->copy); }
        } waiter{ctx};
        lap("copies started");
        std::vector<WinSpec> specs{spec_from_desc(*win)};
        rc = job_build(ctx, specs, 0, &job, &su);
        lap("job built");
        if (rc) return rc;
        Plan& pl = job->plans[0];
        pl.out_z = win->out_z; pl.out_info = win->out_info; pl.out_status = win->out_status;
        pl.out_b11 = win->out_b11; pl.out_b21 = win->out_b21; pl.out_r = win->out_r; pl.out_num_eig = win->out_num_eig;
        rc = job_run_streamed(job, su);
        lap("run queued");
        if (!rc) rc = job_fetch(job);
        lap("fetched");
        if (!rc) waiter.landed = true;             // the results were computed from every chunk: all copies are complete
        else {
            // nothing may still be writing into the landing buffer or reading it when the next call reuses it
            ctx->worker->wait();
            for (hipStream_t q : {ctx->copy, ctx->aux, ctx->chain, ctx->stream}) (void)hipStreamSynchronize(q);
        }
        job_free(job);
        lap("freed");
        return rc;
    }
    rc = gauss_job_create(ctx, win, 1, 0, &job);
    if (rc) return rc;
    rc = job_run(job, true);
    if (!rc) rc = job_fetch(job);
    job_free(job);
    return rc;
}

struct RowSource {                 // where the rows of an LD-only call come from (default: a contiguous host byte matrix)
    int geno_fmt = GAUSS_GENO_U8;
    const int32_t* rows = nullptr;
    const int32_t* pop_src_off = nullptr;
    int on_device = 0;
};

static int ld_common(gauss_ctx* ctx, int mode, const uint8_t* geno, int n_snp, int64_t ld,
                     const int32_t* pop_off, const double* pop_wgt, int n_pop, double diag,
                     const int32_t* gene_off, int n_gene, double* out, int64_t* out_counts, int n_samples,
                     const RowSource& src = RowSource())
{
    if (!ctx || !geno || (!out && !out_counts)) return fail(GAUSS_E_INVALID, "bad arguments");
    WinSpec w;
    int32_t off1[2] = {0, n_samples};
    w.mode = mode; w.n_pop = out_counts ? 1 : n_pop; w.pop_off = out_counts ? off1 : pop_off; w.pop_wgt = pop_wgt;
    w.M = n_snp; w.U = 0; w.geno_m = geno; w.geno_u = nullptr; w.ld = ld; w.z1 = nullptr;
    w.lambda = 0; w.eps = 0; w.diag = diag; w.ld_only = 1; w.gene_off = gene_off; w.n_gene = n_gene;
    w.geno_fmt = src.geno_fmt; w.rows_m = src.rows; w.pop_src_off = src.pop_src_off;
    gauss_job* job = nullptr;
    std::vector<WinSpec> specs{w};
    int rc = job_build(ctx, specs, src.on_device, &job);
    if (rc) return rc;
    std::unique_ptr<gauss_job, void (*)(gauss_job*)> guard(job, job_free);
    if (!out_counts) job->plans[0].out_ld_user = out;
    rc = job_run(job, false);
    if (rc) return rc;
    if (out_counts) {
        DevBuf d_cnt;
        const size_t bytes = sizeof(long long) * (size_t)n_snp * n_snp;
        if (d_cnt.alloc(ctx, bytes) != hipSuccess) return fail(GAUSS_E_NOMEM, "hipMalloc(%zu bytes of counts) failed", bytes);
        launch_counts(job->d_probs, 0, job->plans[0].p.npair, d_cnt.as<long long>(), ctx->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipMemcpy(out_counts, d_cnt.p, bytes, hipMemcpyDeviceToHost));
    }
    return job_fetch(job);
}

int gauss_ld(gauss_ctx* ctx, int mode, const uint8_t* geno, int n_snp, int64_t ld, const int32_t* pop_off,
             const double* pop_wgt, int n_pop, double diag, double* out_cor)
{
    return ld_common(ctx, mode, geno, n_snp, ld, pop_off, pop_wgt, n_pop, diag, nullptr, 0, out_cor, nullptr, 0);
}

int gauss_gene_ld_batch(gauss_ctx* ctx, int mode, const uint8_t* geno, int n_snp, int64_t ld,
                        const int32_t* pop_off, const double* pop_wgt, int n_pop,
                        const int32_t* gene_off, int n_gene, double diag, double* out_blocks)
{
    if (!gene_off || n_gene < 1) return fail(GAUSS_E_INVALID, "gene_off is NULL or n_gene < 1");
    return ld_common(ctx, mode, geno, n_snp, ld, pop_off, pop_wgt, n_pop, diag, gene_off, n_gene, out_blocks, nullptr, 0);
}

int gauss_ld_rows(gauss_ctx* ctx, int mode, const uint8_t* store, int64_t ld, int geno_format, const int32_t* rows, int n_snp,
                  const int32_t* pop_off, const int32_t* pop_src_off, const double* pop_wgt, int n_pop, double diag,
                  int on_device, double* out_cor)
{
    RowSource src;
    src.geno_fmt = geno_format; src.rows = rows; src.pop_src_off = pop_src_off; src.on_device = on_device;
    return ld_common(ctx, mode, store, n_snp, ld, pop_off, pop_wgt, n_pop, diag, nullptr, 0, out_cor, nullptr, 0, src);
}

int gauss_gene_ld_batch_rows(gauss_ctx* ctx, int mode, const uint8_t* store, int64_t ld, int geno_format, const int32_t* rows,
                             int n_snp, const int32_t* pop_off, const int32_t* pop_src_off, const double* pop_wgt, int n_pop,
                             const int32_t* gene_off, int n_gene, double diag, int on_device, double* out_blocks)
{
    if (!gene_off || n_gene < 1) return fail(GAUSS_E_INVALID, "gene_off is NULL or n_gene < 1");
    RowSource src;
    src.geno_fmt = geno_format; src.rows = rows; src.pop_src_off = pop_src_off; src.on_device = on_device;
    return ld_common(ctx, mode, store, n_snp, ld, pop_off, pop_wgt, n_pop, diag, gene_off, n_gene, out_blocks, nullptr, 0, src);
}

int gauss_ld_per_pop(gauss_ctx* ctx, const uint8_t* geno, int n_snp, int64_t ld, const int32_t* pop_off, int n_pop,
                     double* out)
{
    if (!ctx || !geno || !pop_off || !out) return fail(GAUSS_E_INVALID, "bad arguments to gauss_ld_per_pop");
    if (n_snp < 2) return fail(GAUSS_E_INVALID, "need at least two SNPs");
    // the weighted layout keeps one exact Gram partial per population: all that is needed here
    std::vector<double> ones((size_t)std::max(n_pop, 1), 1.0);
    WinSpec w;
    w.mode = GAUSS_MODE_WEIGHTED; w.n_pop = n_pop; w.pop_off = pop_off; w.pop_wgt = ones.data();
    w.M = n_snp; w.U = 0; w.geno_m = geno; w.geno_u = nullptr; w.ld = ld; w.z1 = nullptr;
    w.lambda = 0; w.eps = 0; w.diag = 1.0; w.ld_only = 1; w.gene_off = nullptr; w.n_gene = 0;
    gauss_job* job = nullptr;
    std::vector<WinSpec> specs{w};
    int rc = job_build(ctx, specs, 0, &job);
    if (rc) return rc;
    std::unique_ptr<gauss_job, void (*)(gauss_job*)> guard(job, job_free);
    rc = job_run(job, false);
    if (rc) return rc;
    const size_t npairs = (size_t)n_snp * (n_snp - 1) / 2;
    const size_t bytes = sizeof(double) * npairs * (size_t)n_pop;
    DevBuf d_out;
    if (d_out.alloc(ctx, bytes) != hipSuccess) return fail(GAUSS_E_NOMEM, "hipMalloc(%zu bytes per-population LD) failed", bytes);
    launch_pop_cor(job->d_probs, 0, job->plans[0].p.npair, d_out.as<double>(), ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(out, d_out.p, bytes, hipMemcpyDeviceToHost));
    return GAUSS_OK;
}

int gauss_ld_per_pop_pairs(gauss_ctx* ctx, const uint8_t* geno, int n_snp, int64_t ld, const int32_t* pop_off, int n_pop,
                           const int32_t* pop_group, int n_group, const int32_t* pair_i, const int32_t* pair_j, int64_t n_pairs,
                           double* out)
{
    if (!ctx || !geno || !pop_off || !out || !pair_i || !pair_j) return fail(GAUSS_E_INVALID, "bad arguments to gauss_ld_per_pop_pairs");
    if (n_snp < 2 || n_pairs < 1) return fail(GAUSS_E_INVALID, "need at least two SNPs and one pair");
    if (!pop_group) n_group = n_pop;
    if (n_group < 1) return fail(GAUSS_E_INVALID, "n_group < 1");
    if (pop_group)
        for (int p = 0; p < n_pop; p++)
            if (pop_group[p] < 0 || pop_group[p] >= n_group) return fail(GAUSS_E_INVALID, "pop_group[%d] = %d is outside 0..%d", p, pop_group[p], n_group - 1);
    std::vector<double> ones((size_t)std::max(n_pop, 1), 1.0);
    WinSpec w;
    w.mode = GAUSS_MODE_WEIGHTED; w.n_pop = n_pop; w.pop_off = pop_off; w.pop_wgt = ones.data();
    w.M = n_snp; w.U = 0; w.geno_m = geno; w.geno_u = nullptr; w.ld = ld; w.z1 = nullptr;
    w.lambda = 0; w.eps = 0; w.diag = 1.0; w.ld_only = 1; w.gene_off = nullptr; w.n_gene = 0;
    w.pair_i = pair_i; w.pair_j = pair_j; w.n_pairs = n_pairs;
    gauss_job* job = nullptr;
    std::vector<WinSpec> specs{w};
    int rc = job_build(ctx, specs, 0, &job);
    if (rc) return rc;
    std::unique_ptr<gauss_job, void (*)(gauss_job*)> guard(job, job_free);
    // pack + Gram only: the LD epilogue has nothing to write for a pair list
    hipStream_t st = ctx->stream;
    launch_pack_stats(job->d_probs, job->d_rowmap, job->n_rows, st);
    launch_gram(job->d_items, job->n_items, job->gram_i8, st);
    HIPCHK(hipGetLastError());
    std::vector<int2> pairs((size_t)n_pairs);
    for (int64_t k = 0; k < n_pairs; k++) pairs[(size_t)k] = make_int2(pair_i[k], pair_j[k]);
    DevBuf d_pairs, d_grp, d_out;
    const size_t out_bytes = sizeof(double) * (size_t)n_pairs * (size_t)n_group;
    if (d_pairs.alloc(ctx, sizeof(int2) * pairs.size()) != hipSuccess || d_grp.alloc(ctx, sizeof(int) * (size_t)std::max(n_pop, 1)) != hipSuccess ||
        d_out.alloc(ctx, out_bytes) != hipSuccess)
        return fail(GAUSS_E_NOMEM, "hipMalloc(%zu bytes of pair correlations) failed", out_bytes);
    HIPCHK(hipMemcpyAsync(d_pairs.p, pairs.data(), sizeof(int2) * pairs.size(), hipMemcpyHostToDevice, st));
    if (pop_group) HIPCHK(hipMemcpyAsync(d_grp.p, pop_group, sizeof(int) * (size_t)n_pop, hipMemcpyHostToDevice, st));
    launch_pair_cor(job->d_probs, 0, d_pairs.as<int2>(), n_pairs, pop_group ? d_grp.as<int>() : nullptr, n_group, d_out.as<double>(), st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(out, d_out.p, out_bytes, hipMemcpyDeviceToHost));
    return GAUSS_OK;
}

int gauss_gram_counts(gauss_ctx* ctx, const uint8_t* geno, int n_snp, int n_samples, int64_t ld, int64_t* out_counts)
{
    if (!out_counts) return fail(GAUSS_E_INVALID, "out_counts is NULL");
    return ld_common(ctx, GAUSS_MODE_POOLED, geno, n_snp, ld, nullptr, nullptr, 1, 1.0, nullptr, 0, nullptr,
                     out_counts, n_samples);
}

int gauss_pack2bit_device(gauss_ctx* ctx, const uint8_t* d_in, int64_t ld_in, uint8_t* d_out, int64_t ld_out,
                          int n_snp, const int32_t* pop_off, int n_pop)
{
    if (!ctx || !d_in || !d_out || !pop_off || n_snp < 1 || n_pop < 1) return fail(GAUSS_E_INVALID, "bad arguments");
    std::vector<int> blk(n_pop + 1, 0);
    for (int q = 0; q < n_pop; q++) blk[q + 1] = blk[q] + (int)rup((size_t)(pop_off[q + 1] - pop_off[q]), 64) / 4;
    if (ld_out % 16 || ld_out < blk[n_pop]) return fail(GAUSS_E_INVALID, "ld_out must be a multiple of 16 and >= %d", blk[n_pop]);
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf tab;
    HIPCHK(tab.alloc(sizeof(int) * 2 * (n_pop + 1)));
    int* d_tab = tab.as<int>();
    HIPCHK(hipMemcpy(d_tab, pop_off, sizeof(int) * (n_pop + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_tab + n_pop + 1, blk.data(), sizeof(int) * (n_pop + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemsetAsync(d_out, 0, (size_t)n_snp * ld_out, ctx->stream));
    launch_pack2bit(d_in, ld_in, d_out, ld_out, n_snp, d_tab, d_tab + n_pop + 1, n_pop, ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GAUSS_OK;
}

int gauss_synth_device(gauss_ctx* ctx, uint8_t* d_out, int n_snp, int64_t ld, const int32_t* pop_off, int n_pop,
                       const float* thr, const float* rho, uint64_t seed)
{
    if (!ctx || !d_out || !pop_off || !thr || !rho || n_snp < 1 || n_pop < 1) return fail(GAUSS_E_INVALID, "bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    const int N = pop_off[n_pop];
    DevBuf b_off, b_thr, b_rho;
    HIPCHK(b_off.alloc(sizeof(int) * (n_pop + 1)));
    HIPCHK(b_thr.alloc(sizeof(float) * (size_t)n_snp * n_pop));
    HIPCHK(b_rho.alloc(sizeof(float) * n_snp));
    HIPCHK(hipMemcpy(b_off.p, pop_off, sizeof(int) * (n_pop + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b_thr.p, thr, sizeof(float) * (size_t)n_snp * n_pop, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b_rho.p, rho, sizeof(float) * n_snp, hipMemcpyHostToDevice));
    launch_synth(d_out, n_snp, ld, b_off.as<int>(), n_pop, N, b_thr.as<float>(), b_rho.as<float>(), seed, ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GAUSS_OK;
}

}  // extern "C"
