"""C-ABI lifetime rule (include/gauss_hip.h): a context may be destroyed before the jobs and row stores made on it.

Round 2 had no such rule and its one process abort came from exactly that order: a failing test kept a Job alive in its
traceback, the session fixture closed the context, and at interpreter exit gauss_job_destroy wrote into the freed
context (mutex, block maps) -- heap corruption, "dumped core".  The scenarios run in a child process so that a
regression shows up as that child's exit status, not as the death of the test runner."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %(root)r)
from gauss_amd import hotpath, synth, _lib

pops = synth.pop_table(scale=0.02, min_size=30)[:6]
rng = np.random.default_rng(5)
bp = np.sort(rng.choice(np.arange(1, 200_000), size=260, replace=False))
G, _ = synth.synth_genotypes(bp, pops, seed=3)
G = G[G.min(1) != G.max(1)]
off = synth.pop_offsets([p[1] for p in pops])
w = rng.uniform(0.05, 0.3, len(pops))
wins = []
for m, u in ((70, 90), (64, 40)):
    idx = rng.permutation(G.shape[0])
    wins.append(dict(mode=1, geno_m=np.ascontiguousarray(G[np.sort(idx[:m])]), geno_u=np.ascontiguousarray(G[np.sort(idx[m:m + u])]),
                     pop_off=off, pop_wgt=w, z1=rng.standard_normal(m)))

ctx = hotpath.Context(0)
ref_job = hotpath.Job(wins, ctx=ctx)
ref_job.run()
ref = ref_job.fetch()

job = hotpath.Job(wins, ctx=ctx)          # two runs queued, none fetched
job.run()
job.run()
store = hotpath.RowStore(G, ctx=ctx)      # a row store nobody frees
lib = ctx.lib
handle = job.handle
ctx.close()                               # context first: waits for the runs, releases the job's blocks and the store
rc = lib.gauss_job_run(handle)            # the handle is an empty shell now
assert rc != 0 and b"context has been destroyed" in lib.gauss_last_error(), (rc, lib.gauss_last_error())
assert lib.gauss_job_fetch(handle) != 0
job.close()                               # frees the shell
ref_job.close()
store.ptr = None                          # its memory went with the context

# a new context (possibly at the old one's address) starts clean and gives the same bits
ctx2 = hotpath.Context(0)
assert ctx2.id != 0
j2 = hotpath.Job(wins, ctx=ctx2)
j2.run()
got = j2.fetch()
for a, b in zip(got, ref):
    assert np.array_equal(a["z"], b["z"]) and np.array_equal(a["info"], b["info"])
freed = ctx2.trim_cache()
assert freed >= 0
j2.close()
assert ctx2.trim_cache() > 0              # the job's workspace had gone to the cache
ctx2.close()
print("lifetime ok")
"""


@pytest.mark.gpu
def test_context_destroyed_before_its_jobs_and_stores():
    out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-4000:])
    assert "lifetime ok" in out.stdout


@pytest.mark.gpu
def test_resident_panel_does_not_survive_its_context(tmp_path):
    """the host layer's resident-panel cache was keyed by the context's address and never invalidated -- a
    context created at a destroyed context's address inherited a device pointer that was not its own.  Keys are context
    ids now and a destroy hook drops them: the second context uploads the panel again and returns the same table."""
    from gauss_amd import api, hotpath
    from test_farm import make_study, WGT
    st = make_study(tmp_path)
    p = st["paths"]
    gpk = str(tmp_path / "panel.gpk")
    assert api.pack_panel(p["index.gz"], p["data.gz"], p["desc.txt"], gpk) > 0
    kw = dict(kind=api.KIND_DISTMIX, pop_wgt_df=WGT, window_size=500_000, chr=22, start_bp=1_000_001, end_bp=4_000_000,
              wing_size=200_000, input_file=p["gwas.txt"], reference_data_file=gpk, reference_pop_desc_file=p["desc.txt"])
    tables = []
    ids = []
    for _ in range(3):
        c = hotpath.Context(0)
        ids.append(c.id)
        res = api.impute_chromosome(ctx=c, **kw)
        assert res.stats["panel_bytes_uploaded"] > 0              # never inherited from a dead context
        again = api.impute_chromosome(ctx=c, **kw)
        assert again.stats["panel_bytes_uploaded"] == 0           # resident on this one
        tables.append(res.columns["z"])
        c.close()                                                 # no panel_evict: the destroy hook lets go of it
    assert len(set(ids)) == 3
    assert all(np.array_equal(t, tables[0]) for t in tables[1:])


@pytest.mark.gpu
def test_asynchronous_row_store_upload(ctx, tmp_path):
    """gauss_store_upload_async + gauss_store_wait: a job over the first rows of a store may be queued while the rest is
    still on its way (the library's stream waits for the mark that covers the rows named), a job over all rows after
    wait(all), and both give the bits of a synchronous store.  Large enough for several 32 MB chunks."""
    from gauss_amd import hotpath, synth, panel
    pops = synth.pop_table(scale=0.25, min_size=64)[:8]
    sizes = [p[1] for p in pops]
    off = synth.pop_offsets(sizes)
    rng = np.random.default_rng(11)
    n_snp = 120000
    bp = np.sort(rng.choice(np.arange(1, 3_000_000), size=n_snp, replace=False))
    G, _ = synth.synth_genotypes(bp[:600], pops, seed=4)
    G = G[G.min(1) != G.max(1)][:512]
    packed, _ = panel.pack2bit(G, off)
    rows = np.ascontiguousarray(np.tile(packed, (n_snp // 512 + 1, 1))[:n_snp])     # 100+ MB of 2-bit rows
    assert rows.nbytes > 3 * (32 << 20)
    w = rng.uniform(0.05, 0.3, len(pops))

    def job_over(store, lo):
        rm = np.arange(lo, lo + 200, dtype=np.int32)
        ru = np.arange(lo + 200, lo + 500, dtype=np.int32)
        d = dict(mode=1, pop_off=off, pop_wgt=w, z1=np.linspace(-2, 2, 200), dev=(store.ptr, store.ptr, 200, 300, store.ld),
                 packed=dict(fmt=1, rows_m=rm, rows_u=ru))
        j = hotpath.Job([d], ctx=ctx, on_device=True)
        j.run()
        r = j.fetch()[0]
        j.close()
        return r

    sync = hotpath.RowStore(rows, ctx=ctx)
    want_lo, want_hi = job_over(sync, 0), job_over(sync, n_snp - 512)
    sync.close()
    st = hotpath.RowStore(rows, ctx=ctx, asynchronous=True)
    st.wait(512)                                   # the first rows only: the job below starts while the upload goes on
    got_lo = job_over(st, 0)
    st.wait(0)
    got_hi = job_over(st, n_snp - 512)
    st.wait(0)                                     # a second full wait is a no-op
    st.close()
    for a, b in ((got_lo, want_lo), (got_hi, want_hi)):
        assert np.array_equal(a["z"], b["z"]) and np.array_equal(a["info"], b["info"])
    # the same store reserved and filled piece by piece (gauss_store_alloc / gauss_store_fill: the chromosome driver's first
    # call on a panel brings the rows of batch k + 1 up while batch k computes): a job over the first rows runs while a later
    # piece is still to come; pieces of less than a staging chunk, of several chunks, and an empty one
    pw = hotpath.RowStore(rows, ctx=ctx, reserve_only=True)
    pw.fill(0, 700)
    job_lo = job_over(pw, 0)
    pw.fill(700, 700)
    pw.fill(700, n_snp - 9000)
    pw.fill(n_snp - 9000, n_snp)
    job_hi = job_over(pw, n_snp - 512)
    with pytest.raises(Exception) as ei:
        pw.fill(n_snp - 10, n_snp + 1)
    assert "outside the store" in str(ei.value)
    pw.close()
    for a, b in ((job_lo, want_lo), (job_hi, want_hi)):
        assert np.array_equal(a["z"], b["z"]) and np.array_equal(a["info"], b["info"])
    busy = hotpath.RowStore(rows, ctx=ctx, asynchronous=True)
    busy.close()                                   # freed while the upload may still be running: it is finished first
    # the rows as a section of a FILE (what the chromosome driver uploads: a packed panel's genotype section, read with pread
    # into the pinned staging buffers): in one go (gauss_store_upload_fd) and piece by piece (gauss_store_fill_fd), at an offset
    # that is not page aligned
    path = str(tmp_path / "rows.bin")
    with open(path, "wb") as fh:
        fh.write(b"x" * 1000)
        rows.tofile(fh)
    ff = hotpath.RowStore.from_file(path, 1000, n_snp, rows.shape[1], ctx=ctx)
    f_lo, f_hi = job_over(ff, 0), job_over(ff, n_snp - 512)
    ff.close()
    fp_ = hotpath.RowStore.from_file(path, 1000, n_snp, rows.shape[1], ctx=ctx, reserve_only=True)
    fp_.fill(0, 900)
    p_lo = job_over(fp_, 0)
    fp_.fill(900, n_snp)
    p_hi = job_over(fp_, n_snp - 512)
    fp_.close()
    with pytest.raises(Exception):
        hotpath.RowStore.from_file(path, 1000, n_snp + 5, rows.shape[1], ctx=ctx)       # the file is shorter than that
    for a, b in ((f_lo, want_lo), (f_hi, want_hi), (p_lo, want_lo), (p_hi, want_hi)):
        assert np.array_equal(a["z"], b["z"]) and np.array_equal(a["info"], b["info"])


@pytest.mark.gpu
def test_streamed_window_call_strided_rows_and_error_paths(ctx, monkeypatch):
    """The blocking call on host bytes is streamed (copy worker + landing buffer; docs/HISTORY.md section 9f): same bits as
    upload-then-run, also for matrices whose row stride is far from the row length (pitched chunk copies), for a 2-bit
    host store, and a call that fails after its copies have started leaves the context usable."""
    import ctypes as C
    from gauss_amd import hotpath, synth, _lib, panel
    pops = synth.pop_table(scale=0.03, min_size=40)[:9]
    off = synth.pop_offsets([p[1] for p in pops])
    N = int(off[-1])
    rng = np.random.default_rng(21)
    bp = np.sort(rng.choice(np.arange(1, 900_000), size=1100, replace=False))
    G, _ = synth.synth_genotypes(bp, pops, seed=6)
    G = G[G.min(1) != G.max(1)]
    idx = rng.permutation(G.shape[0])
    gm = np.ascontiguousarray(G[np.sort(idx[:300])])
    gu = np.ascontiguousarray(G[np.sort(idx[300:300 + 700])])          # six row tiles: several chunks
    w = rng.uniform(0.05, 0.3, len(pops))
    z1 = rng.standard_normal(gm.shape[0])

    def call(stream, a=gm, b=gu, ld=None, mode=1):
        monkeypatch.setenv("GAUSS_STREAM_WINDOW", "1" if stream else "0")
        desc = _lib.WindowDesc()
        win = hotpath._Win(desc, mode, gm, gu, off, w, z1, 0.1, 1e-5, False)
        if ld is not None:
            desc.geno_m, desc.geno_u, desc.ld = a.ctypes.data, b.ctypes.data, ld
        _lib.check(ctx.lib.gauss_impute_window(ctx.handle, C.byref(desc)))
        return win.result()

    want = call(False)
    got = call(True)
    assert np.array_equal(got["z"], want["z"]) and np.array_equal(got["info"], want["info"])
    # rows inside much wider host matrices: pitched copies, chunk by chunk
    wide = N + N // 4 + 512
    wm = np.full((gm.shape[0], wide), 7, dtype=np.uint8)
    wu = np.full((gu.shape[0], wide), 7, dtype=np.uint8)
    wm[:, :N] = gm
    wu[:, :N] = gu
    for stream in (False, True):
        r = call(stream, wm, wu, wide)
        assert np.array_equal(r["z"], want["z"]) and np.array_equal(r["info"], want["info"])
    # a call that fails in the planner, after the copy worker has been set going
    monkeypatch.setenv("GAUSS_STREAM_WINDOW", "1")
    with pytest.raises(Exception) as ei:
        call(True, mode=7)
    assert "bad mode" in str(ei.value)
    again = call(True)
    assert np.array_equal(again["z"], want["z"]) and np.array_equal(again["info"], want["info"])


def _merged_job_windows(store, src_off, p, rng, n_win=3):
    wins = []
    for k in range(n_win):
        mi = np.arange(100 * k, 100 * k + 700, 2, dtype=np.int32)
        ui = np.arange(100 * k + 1, 100 * k + 601, 2, dtype=np.int32)
        wins.append(dict(mode=1, pop_off=p["off"], pop_wgt=p["w"], z1=rng.standard_normal(len(mi)), dev=(store.ptr, store.ptr, len(mi), len(ui), store.ld),
                         packed=dict(fmt=1, rows_m=mi, rows_u=ui, pop_src_off=src_off)))
    return wins


def _check_against_oracle(p, wins, res):
    import oracle
    for w, r in zip(wins, res):
        gm, gu = p["G"][w["packed"]["rows_m"]], p["G"][w["packed"]["rows_u"]]
        want = oracle.run_impute(1, np.ascontiguousarray(gm), np.ascontiguousarray(gu), p["off"], p["w"], w["z1"])
        assert r["status"] == 0
        assert np.max(np.abs(r["z"] - want["z"]) / np.maximum(1.0, np.abs(want["z"]))) <= 1e-8
        assert np.max(np.abs(r["info"] - want["info"]) / want["info"]) <= 1e-8


@pytest.mark.gpu
def test_merged_launch_that_gives_up_is_run_again_in_the_two_launch_form(ctx, monkeypatch):
    """The chain queue of a merged Gram launch waits for a COUNT of finished B11 items (k_gram.hip: wait_count_kernel), and
    that wait is bounded.  Made to wait for a count that never comes (the test hook), it gives up after its bound and raises the
    run's failure flag; gauss_job_fetch then runs the job once more in the two-launch form -- no kernel of which waits for
    another queue -- inside the same call: the caller gets its windows (one call returns its window or an error,
    dist.cpp:30-126), the context counts the give-up, and the same job runs merged again afterwards (the counter's target is per
    launch, so a failed run does not shift the next one's).  With two runs in flight both are repaired."""
    from gauss_amd import hotpath, panel
    from helpers import small_panel
    p = small_panel(n_snp=1500, scale=0.05, seed=17)
    rows2, src_off = panel.pack2bit(p["G"], p["off"])
    store = hotpath.RowStore(rows2, ctx=ctx)
    wins = _merged_job_windows(store, src_off, p, np.random.default_rng(3))
    monkeypatch.setenv("GAUSS_CHAIN_ASIDE", "2")              # the merged form on a job this small ...
    monkeypatch.setenv("GAUSS_CHAIN_MERGED", "2")             # ... whatever the session's environment or the queue registry says
    job = hotpath.Job(wins, ctx=ctx, on_device=True)
    other = hotpath.Job(wins, ctx=ctx, on_device=True)        # a second job on the context: its own counters stay at zero
    c0 = ctx.counters()
    monkeypatch.setenv("GAUSS_WAIT_COUNT_TIMEOUT_US", "-2000")     # wait 2 ms for a count that never comes
    job.profile(True)
    job.run()
    res = job.fetch()
    c1 = ctx.counters()
    assert c1["giveups"] == c0["giveups"] + 1 and c1["rerun_failed"] == c0["rerun_failed"], (c0, c1)
    assert job.counters() == dict(merged=1, demoted=1, giveups=1, rerun_failed=0)      # (the re-run is a run in the two-launch form)
    assert other.counters() == dict(merged=0, demoted=0, giveups=0, rerun_failed=0)
    # the stage timers describe the run that delivered (the two-launch re-run), not the sum of it and the void merged run
    gram_ms, gram_launches = job.profile_get(0)
    assert gram_launches == 2 and 0 < gram_ms < 50.0, (gram_ms, gram_launches)
    job.profile(False)
    other.close()
    _check_against_oracle(p, wins, res)
    job.run()
    job.run()                                                # two failing runs in flight
    a, b = job.fetch(), job.fetch()
    c2 = ctx.counters()
    assert c2["giveups"] == c1["giveups"] + 2 and c2["rerun_failed"] == c0["rerun_failed"], (c1, c2)
    monkeypatch.delenv("GAUSS_WAIT_COUNT_TIMEOUT_US")
    job.run()
    again = job.fetch()
    c3 = ctx.counters()
    assert c3["giveups"] == c2["giveups"] and c3["merged"] == c2["merged"] + 1, (c2, c3)
    job.close()
    for r in (a, b, again):
        for x, y in zip(res, r):
            assert np.array_equal(x["z"], y["z"]) and np.array_equal(x["info"], y["info"])
    store.close()


@pytest.mark.gpu
@pytest.mark.parametrize("force", [False, True])
def test_two_contexts_run_merged_jobs_at_the_same_time(ctx, monkeypatch, request, force):
    """Two contexts on device 0, two host threads, each running a 3-window job built for the merged Gram launch, at the same
    time.  The runtime gives a priority stream a hardware queue of its own only while its class holds at most
    GPU_MAX_HW_QUEUES streams (profiles/r05_queue_map.txt): with two contexts the streams share queues, and a kernel that
    spins for another queue's progress could sit in front of the work it waits for.  So (force = False, the default) both
    contexts queue their runs as two launches joined by an event while the other one is alive -- counted as `demoted` -- and
    the first context goes back to merged runs once it is alone again.  With the merged form forced on both (=2) whatever
    happens on the shared queues must end in the right bits: a run whose waiting kernel gives up is repaired inside its fetch.
    Either way: the bits of the single-context run, and no run takes anywhere near the two seconds of a give-up unless one
    was counted."""
    import threading
    import time
    from gauss_amd import hotpath, panel
    from helpers import small_panel
    p = small_panel(n_snp=1500, scale=0.05, seed=19)
    rows2, src_off = panel.pack2bit(p["G"], p["off"])
    monkeypatch.setenv("GAUSS_CHAIN_ASIDE", "2")
    monkeypatch.setenv("GAUSS_CHAIN_MERGED", "2" if force else "1")
    # (the merged launch is the f32 kernel's default form: under a suite-wide GAUSS_GRAM_DTYPE=i8 this test still runs it, and hands the
    # session's context back as it found it)
    ctx.set_gram_dtype("f32")
    request.addfinalizer(lambda: ctx.set_gram_dtype(os.environ.get("GAUSS_GRAM_DTYPE", "f32")))
    rng = np.random.default_rng(4)
    store = hotpath.RowStore(rows2, ctx=ctx)
    wins = _merged_job_windows(store, src_off, p, rng)
    solo = hotpath.Job(wins, ctx=ctx, on_device=True)
    c_before = ctx.counters()
    solo.run()
    want = solo.fetch()
    c_solo = ctx.counters()
    assert c_solo["merged"] == c_before["merged"] + 1, (c_before, c_solo)      # alone on the device: the merged form
    _check_against_oracle(p, wins, want)

    q0 = ctx.queues()
    assert q0["probed_distinct"] and q0["hi_streams"] <= q0["hw_queues_per_class"], q0      # the session's context was made alone on the device
    other = hotpath.Context(0)
    other.set_gram_dtype("f32")          # (the merged launch is the f32 kernel's form: a suite run under GAUSS_GRAM_DTYPE=i8 still tests it here)
    q1 = other.queues()
    assert q1["hi_streams"] > q1["hw_queues_per_class"], q1                               # two contexts: more priority streams than hardware queues
    store2 = hotpath.RowStore(rows2, ctx=other)
    wins2 = [dict(w, dev=(store2.ptr, store2.ptr, w["dev"][2], w["dev"][3], store2.ld)) for w in wins]
    jobs = [solo, hotpath.Job(wins2, ctx=other, on_device=True)]
    n_runs = 12
    slowest = [0.0, 0.0]
    results = [[], []]
    errors = []
    gate = threading.Barrier(2)

    def worker(k):
        try:
            gate.wait()
            for _ in range(n_runs):
                t0 = time.perf_counter()
                jobs[k].run()
                results[k].append(jobs[k].fetch())
                slowest[k] = max(slowest[k], time.perf_counter() - t0)
        except Exception as ex:      # pragma: no cover
            errors.append((k, repr(ex)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    ca, cb = ctx.counters(), other.counters()
    for k in range(2):
        for r in results[k]:
            for x, y in zip(want, r):
                assert np.array_equal(x["z"], y["z"]) and np.array_equal(x["info"], y["info"]), k
    gave_up = (ca["giveups"] - c_solo["giveups"]) + cb["giveups"]
    assert ca["rerun_failed"] == c_solo["rerun_failed"] and cb["rerun_failed"] == 0
    if not force:
        assert ca["demoted"] - c_solo["demoted"] == n_runs and cb["demoted"] == n_runs, (ca, cb)
        assert ca["merged"] == c_solo["merged"] and cb["merged"] == 0 and gave_up == 0, (ca, cb)
    else:
        assert ca["merged"] - c_solo["merged"] == n_runs and cb["merged"] == n_runs, (ca, cb)
    if gave_up == 0:
        assert max(slowest) < 1.0, slowest                       # nothing stalled
    jobs[1].close()
    store2.close()
    other.close()
    if not force:
        solo.run()
        back = solo.fetch()
        assert ctx.counters()["merged"] == ca["merged"] + 1      # alone again: merged again
        for x, y in zip(want, back):
            assert np.array_equal(x["z"], y["z"]) and np.array_equal(x["info"], y["info"])
    solo.close()
    store.close()


@pytest.mark.gpu
def test_threads_share_one_context_with_merged_runs(ctx, monkeypatch):
    """Several host threads queue merged runs on ONE context at the same time (what gauss_host_impute_genome does with its two
    chromosome calls in flight).  A run's kernels go onto three queues and its waiting kernels rely on every queue seeing the
    context's runs in the same order; the library queues one run at a time per context (gauss_ctx::run_mu).  Three threads, their
    own job each, ten runs each: every result the single-threaded one, no give-up, nothing near a stall."""
    import threading
    import time
    from gauss_amd import hotpath, panel
    from helpers import small_panel
    p = small_panel(n_snp=1500, scale=0.05, seed=23)
    rows2, src_off = panel.pack2bit(p["G"], p["off"])
    monkeypatch.setenv("GAUSS_CHAIN_ASIDE", "2")
    monkeypatch.setenv("GAUSS_CHAIN_MERGED", "2")
    store = hotpath.RowStore(rows2, ctx=ctx)
    wins = _merged_job_windows(store, src_off, p, np.random.default_rng(6))
    jobs = [hotpath.Job(wins, ctx=ctx, on_device=True) for _ in range(3)]
    jobs[0].run()
    want = jobs[0].fetch()
    c0 = ctx.counters()
    results, errors, slowest = [[], [], []], [], [0.0] * 3
    gate = threading.Barrier(3)

    def worker(k):
        try:
            gate.wait()
            for _ in range(10):
                t0 = time.perf_counter()
                jobs[k].run()
                results[k].append(jobs[k].fetch())
                slowest[k] = max(slowest[k], time.perf_counter() - t0)
        except Exception as ex:      # pragma: no cover
            errors.append((k, repr(ex)))
    th = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    c1 = ctx.counters()
    assert not errors, errors
    assert c1["merged"] - c0["merged"] == 30 and c1["giveups"] == c0["giveups"], (c0, c1)
    assert max(slowest) < 1.0, slowest
    for k in range(3):
        for r in results[k]:
            for x, y in zip(want, r):
                assert np.array_equal(x["z"], y["z"]) and np.array_equal(x["info"], y["info"])
    for j in jobs:
        j.close()
    store.close()
