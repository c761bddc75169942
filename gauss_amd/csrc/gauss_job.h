// Host-side declarations shared by the translation units of libgauss_hip.so that are not kernels:
//   gauss_ctx.cpp    contexts, the per-device stream / hardware-queue registry, block caches, destroy hooks
//   gauss_plan.cpp   the planner: one window -> Plan, a batch of windows -> a job (tables, work items, workspace)
//   gauss_run.cpp    queuing a run of a job on the context's queues, fetching its results, the rare repair paths
//   gauss_store.cpp  resident row stores (host / file -> HBM through pinned double buffers)
//   gauss_abi.cpp    the C ABI of include/gauss_hip.h on top of the above
#pragma once
#include "gauss_internal.h"
// the library is built with -fvisibility=hidden: only the C ABI of include/gauss_hip.h is exported (gauss_amd/build.py)
#pragma GCC visibility push(default)
#include "../../include/gauss_hip.h"
#pragma GCC visibility pop

#include <algorithm>
#include <atomic>
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
#include <shared_mutex>
#include <thread>
#include <string>
#include <vector>

#include <unistd.h>

using namespace gauss;

namespace gauss {
void launch_jacobi_clamp(const Prob* d_probs, int prob, const Prob& hp, double* d_work, bool apply, hipStream_t s);
}

// ------------------------------------------------------------------------------------------
extern thread_local std::string g_err;          // gauss_last_error() (gauss_ctx.cpp)
int fail(int code, const char* fmt, ...);

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

enum StreamClass { STREAM_NORMAL = 0, STREAM_HIGH = 1, STREAM_LOW = 2 };      // priority class of a stream (queue registry, below)

struct gauss_ctx {
    int device;
    uint64_t id = 0;                         // unique per process, never reused (a new context at a freed context's address is a new id)
    hipStream_t stream = nullptr;
    StreamClass main_cls = STREAM_HIGH;      // GAUSS_SIDE_STREAM=0: one normal-priority queue and nothing beside it
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
    // One run is queued at a time per context: a run's kernels go onto several queues, and a merged run's waiting kernels are only
    // sound if every queue sees the runs in the SAME order -- two host threads that interleaved their launches (the genome driver
    // keeps two chromosome calls in flight on one context) could put run B's wait for its Gram items on the chain queue in front of
    // run A's chain, which run A's product -- in front of run B's Gram launch on the main queue -- waits for: a cycle.
    std::mutex run_mu;
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
    std::mutex prepin_mu;                    // guards `prepin` (joinable / join): ctx_join_prepin
    int n_hi = 0, n_lo = 0;                  // streams of this context in the device's high / low priority pools (queue registry)
    bool queues_probed_distinct = false;     // gauss_hip_init SAW a kernel on the chain queue and one on the low-priority queue run beside a
                                             // later kernel of the main queue (queues_side_by_side): the three are different hardware queues
    // how the runs of this context's jobs were queued (gauss_hip_counters)
    std::atomic<long long> n_runs_merged{0}, n_runs_demoted{0}, n_merged_giveups{0}, n_rerun_failed{0};
};

// ---- streams and hardware queues (gauss_ctx.cpp; DESIGN.md section 5 "the queue rule"; docs/HISTORY.md section 4) ----------------------
// The HIP runtime multiplexes streams onto a pool of hardware (HSA) queues PER PRIORITY CLASS: a new stream gets a queue of
// its own until the class holds GPU_MAX_HW_QUEUES of them (default 4), after that the least used queue of the class is handed
// out again.  Kernels of streams that share a hardware queue run in submission order -- so a kernel that SPINS for the
// progress of another stream (k_gram.hip: wait_count_kernel at the head of the chain queue / the low-priority queue of a
// merged Gram launch) is only safe while no stream it may depend on, directly or through another context's events, can sit
// behind it in the same hardware queue.  The registry counts the priority streams this library has alive per device;
// while they fit the pools every one of them owns its hardware queue, and only then is a run queued in the merged form.
int ctx_stream_create(gauss_ctx* c, hipStream_t* out, StreamClass cls);       // registers the stream in the device's pool count
void ctx_stream_destroy(gauss_ctx* c, hipStream_t* s, StreamClass cls);
// Held (shared) by job_queue_run from its decision until the run is queued; a context whose streams overflow a pool takes it
// exclusively and lets the other contexts' spinning kernels drain before its streams exist.
std::shared_mutex& queue_registry_mutex();
bool queues_exclusive(int device);                    // every priority stream of the library on `device` owns a hardware queue
void ctx_join_prepin(gauss_ctx* c);

hipError_t ctx_malloc_retry(gauss_ctx* c, void** out, size_t bytes);
hipError_t ctx_dev_alloc(gauss_ctx* c, size_t bytes, void** out);
void ctx_dev_release(gauss_ctx* c, void* p);
hipError_t ctx_pin_alloc(gauss_ctx* c, size_t bytes, void** out);
void ctx_pin_release(gauss_ctx* c, void* p);

// Diagnostics on stderr: GAUSS_TRACE=job,upload,stream (any subset; "all")
bool trace_on(const char* what);

static inline int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

// Plan thresholds (measured on MI355X).  Both sides of each are ordinary production forms that a job
// takes by its size, and the tests reach them by size.
constexpr int GEMM_SMALL_TILES = 1600;        // jobs with fewer 128-wide tiles of the closing product take 64-wide ones
constexpr int OWN_PANEL_MAX_WINDOWS = 20;     // jobs of at most this many windows factor without panel launches
constexpr int SOLVE_SPLIT_MIN = 2;            // rows of the inverse with at least this many early products are cut into class sums

static const size_t PIN_CACHE_LIMIT = (size_t)1 << 30;
static const size_t UPLOAD_CHUNK = (size_t)32 << 20;       // staging buffer of a row-store upload (upload_rows)
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
    bool merged = false;                                   // chain_aside as ONE Gram launch (B11's items first, counted; job_queue_run)
    bool force_merged = false;                             // GAUSS_CHAIN_MERGED=2: merged even when the context's streams may share hardware queues
    bool queue_touched = false;                            // something of this job may still be queued that no completed fetch covers (job_release)
    hipStream_t zero_queue = nullptr;                      // where job_build zeroed the workspace (the upload queue, or the main queue)
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
    // Matrix exports (out_b11 / out_b21 of LD-export, QCAT and want_mats windows): the run writes them compact into a pinned mirror
    // by kernel, chunk by chunk with an event behind each chunk, and gauss_job_fetch copies a chunk to the caller's memory as soon
    // as its event has completed (a few host threads for large jobs) -- the link, the host copies and the rest of the run overlap.
    // Empty when the job exports nothing or the mirror would not fit the pinned budget (then fetch_matrix per matrix, as before).
    struct Export { int plan, which; size_t off; int rows, width, pitch; double* user; };     // which: 0 = B11, 1 = B21; off in doubles
    std::vector<Export> exports;
    std::vector<std::pair<int, int>> exp_chunks;           // [first, last) exports of every chunk
    std::vector<hipEvent_t> exp_ev;                        // chunk c is in the mirror
    double* h_export = nullptr;                            // inside h_pin
    ExportD* d_exports = nullptr;
    bool prof = false;
    std::vector<ProfSlot> slots;
    double prof_ms[5] = {0, 0, 0, 0, 0};
    long long prof_n[5] = {0, 0, 0, 0, 0};
    bool ran = false;
    bool ran_solve = true;                                 // what the last gauss_job_run asked for (the re-run after a give-up repeats it)
    unsigned prof_run = 0;                                 // run the stage timers being recorded belong to
    long long n_merged = 0, n_demoted = 0, n_giveups = 0, n_rerun_failed = 0;      // this job's share of gauss_hip_counters (gauss_job_counters)
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
    double* out_b11 = nullptr;               // matrices the caller wants back (the job plans their export at build time)
    double* out_b21 = nullptr;
};

// ---- gauss_plan.cpp ----
int plan_problem(const WinSpec& w, Plan& pl, int seg_max, int group_target);
// streamed: one window on contiguous host matrices whose upload is left to job_run_streamed (chunk by chunk on the
// copy stream, overlapped with the pack / Gram launches of the rows that have landed)
int job_build(gauss_ctx* ctx, const std::vector<WinSpec>& specs, int on_device, gauss_job** out, const StreamSetup* stream = nullptr);

// ---- gauss_run.cpp ----
int job_run(gauss_job* job, bool solve);
int job_fetch(gauss_job* job);
int stream_start_copies(gauss_ctx* ctx, const gauss_window_desc& win, size_t row_bytes, StreamSetup& su);
int job_run_streamed(gauss_job* job, StreamSetup& su);
void prof_collect(gauss_job* job, unsigned run_end = ~0u);
// Everything a job holds on its context: waits for its queued runs, then gives the blocks back and destroys the events.
// Called by job_free, and by gauss_hip_destroy for the jobs that outlive their context (the context is still whole then).
void job_release(gauss_job* job);
void job_free(gauss_job* job);

// ---- gauss_store.cpp ----
// (the C ABI's gauss_store_* functions; nothing else is shared)
