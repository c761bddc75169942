"""bench.py modes beside the headline: e2e (files -> table), computeLD (configs[1]), jepegmix (configs[4]).

Everything here is harness code: it builds synthetic inputs in the reference's on-disk formats (or the packed
panel), calls the product through its public entry points and times it; the CPU checker under oracle/ is never touched here.
"""
import ctypes as C
import os
import shutil
import tempfile
import time

import numpy as np

from . import _lib, api, hotpath, panel, synth, workload

HBM_PEAK_GBS = 8000.0
FP32_MFMA_PEAK_TFLOPS = 157.3


# ------------------------------------------------------------------------------------------------------------
# study files on disk
# ------------------------------------------------------------------------------------------------------------
def write_study_files(rig, ch, outdir, seed=5, snp_mask=None):
    """The chromosome of `workload.make_chromosome` as files: a packed panel with ALL 29 populations of the 33KG
    table (N = 32 953; the selected populations' genotypes are the ones bench.py's headline uses, the others are
    filled in), the population description, and the GWAS summary file (measured SNPs only).  Genotypes are
    generated and 2-bit packed on the GPU, then written with panel.write_packed_panel.  Returns a dict of paths."""
    torch, ctx = rig.torch, rig.ctx
    pops_all = synth.pop_table()
    names_sel = [p[0] for p in ch["pops"]]
    sel = [k for k, q in enumerate(pops_all) if q[0] in names_sel]
    rest = [k for k in range(len(pops_all)) if k not in sel]
    rng = np.random.default_rng(seed)
    S = len(ch["bp"])
    thr_all = np.zeros((S, len(pops_all)), dtype=np.float32)
    thr_all[:, sel] = ch["thr"]
    if rest:
        thr_all[:, rest] = ch["thr"][:, rng.integers(0, len(sel), len(rest))]
    off_all = synth.pop_offsets([q[1] for q in pops_all])
    N = int(off_all[-1])
    ld = (N + 63) // 64 * 64
    g = torch.empty((S, ld), dtype=torch.uint8, device="cuda")
    _lib.check(ctx.lib.gauss_synth_device(ctx.handle, g.data_ptr(), S, ld, off_all.ctypes.data_as(C.POINTER(C.c_int32)),
                                          len(pops_all), np.ascontiguousarray(thr_all).ctypes.data_as(C.POINTER(C.c_float)),
                                          ch["rho"].ctypes.data_as(C.POINTER(C.c_float)), C.c_uint64(20260216)))
    sizes = np.diff(off_all)
    ld2 = int(sum((int(m) + 63) // 64 * 16 for m in sizes))
    store = torch.empty((S, ld2), dtype=torch.uint8, device="cuda")
    _lib.check(ctx.lib.gauss_pack2bit_device(ctx.handle, g.data_ptr(), ld, store.data_ptr(), ld2, S,
                                             off_all.ctypes.data_as(C.POINTER(C.c_int32)), len(pops_all)))
    cnt = torch.stack([g[:, off_all[k]:off_all[k + 1]].sum(1, dtype=torch.int32) for k in range(len(pops_all))], 1).cpu().numpy()
    af = cnt / (2.0 * sizes)[None, :]
    rows = store.cpu().numpy()
    del g, store
    torch.cuda.empty_cache()
    # measured SNPs keep the study file's identity (rsid, alleles); the others get synthetic names
    rs, mbp, ma1, ma2, _ = workload.read_study(os.path.join(workload.ROOT, ch["study"]))
    rsid = np.array([f"snp{i}" for i in range(S)], dtype=object)
    alle = np.array(list("ACGT"))
    a1 = alle[rng.integers(0, 4, S)].astype(object)
    a2 = alle[(np.searchsorted(alle, a1.astype(str)) + rng.integers(1, 4, S)) % 4].astype(object)
    m = np.nonzero(ch["measured"])[0]
    if len(m) == len(mbp) and np.array_equal(ch["bp"][m], mbp):
        rsid[m], a1[m], a2[m] = rs, ma1, ma2
    os.makedirs(outdir, exist_ok=True)
    gpk = os.path.join(outdir, "chr22.gpk")
    nbytes = panel.write_packed_panel(gpk, pops_all, rsid, np.full(S, 22), ch["bp"], a1, a2, rows, af, cnt)
    desc = os.path.join(outdir, "desc.txt")
    panel.write_pop_desc(desc, pops_all)
    gwas = os.path.join(outdir, "gwas.txt")
    panel.write_gwas(gwas, rsid[m], np.full(len(m), 22), ch["bp"][m], a1[m], a2[m], ch["z"][m])
    return dict(panel=gpk, desc=desc, gwas=gwas, panel_bytes=int(nbytes), samples_in_file=N, rows2bit=rows,
                rsid=rsid, a1=a1, a2=a2, pops_all=pops_all, af=af)


def study_args(ch, files):
    if ch["mode"] == "dist":
        return dict(kind=api.KIND_DIST, study_pop="EUR", pop_wgt_df=None)
    return dict(kind=api.KIND_DISTMIX, study_pop=None, pop_wgt_df=(list(synth.PGC2_WEIGHTS.keys()), list(synth.PGC2_WEIGHTS.values())))


def chromosome_span(ch):
    lo = (int(ch["bp"][0]) // workload.WINDOW_BP) * workload.WINDOW_BP + 1
    return lo, int(ch["bp"][-1])


def e2e_measure(rig, ch, files, steps, wing, n_batches=0):
    """files -> table through gauss_host_impute_chromosome: one cold call (panel upload included) and `steps` warm
    ones (the panel resident, the way a session imputes study after study).  With several ranks every rank takes
    its LPT share of the windows and rank 0 merges the tables in window order."""
    sa = study_args(ch, files)
    lo, hi = chromosome_span(ch)
    n_batches = n_batches or int(os.environ.get("GAUSS_E2E_BATCHES", "0"))     # experiment override
    kw = dict(chr=22, start_bp=lo, end_bp=hi, wing_size=wing, input_file=files["gwas"], reference_data_file=files["panel"],
              reference_pop_desc_file=files["desc"], rank=rig.rank, world=rig.world, n_batches=n_batches, ctx=rig.ctx, **sa)

    def once():
        rig.barrier()
        t0 = time.perf_counter()
        res = api.impute_chromosome(**kw)
        if rig.world > 1:
            parts = rig.gather(res)
            res = api.ChromResult.merge(parts) if rig.rank == 0 else None
        rig.barrier()
        return time.perf_counter() - t0, res

    api.panel_evict(ctx=rig.ctx)
    cold_s, cold = once()
    warm, res = [], cold
    for _ in range(max(1, steps)):
        t, res = once()
        warm.append(t)
    return cold_s, cold, warm, res


def emulate_world_e2e(rig, ch, files, wing, warm_one_s, world=8, calls=7):
    """The files -> table path as rank r of `world` would run it (gauss_host_impute_chromosome(rank, world): the same plan on every
    rank, this rank's windows through its own data layer, jobs and tables), every rank timed alone on the ONE GPU, warm (the panel
    resident), `calls` calls per rank, median.  predicted_efficiency = one-rank warm time / (world x the slowest rank's);
    host_ms_not_overlapped = a call's time outside its GPU span (plan, the first batch's data layer and job, the last batch's
    tables, the Python wrapper): what does not shrink with the rank's share.  An emulation, not a multi-GPU measurement."""
    sa = study_args(ch, files)
    lo, hi = chromosome_span(ch)

    def sweep(local_world):
        """local_world: what LOCAL_WORLD_SIZE says while the rank runs -- the driver gives a rank its share of the host's cores"""
        old = os.environ.get("LOCAL_WORLD_SIZE")
        if local_world:
            os.environ["LOCAL_WORLD_SIZE"] = str(local_world)
        per = []
        try:
            for r in range(world):
                kw = dict(chr=22, start_bp=lo, end_bp=hi, wing_size=wing, input_file=files["gwas"], reference_data_file=files["panel"],
                          reference_pop_desc_file=files["desc"], rank=r, world=world, n_batches=0, ctx=rig.ctx, **sa)
                ts, res = [], None
                for _ in range(2):                   # this rank's windows into the window cache, its job shapes into the block caches: warm, as defined
                    res = api.impute_chromosome(**kw)
                for _ in range(calls):
                    res = None                       # the previous call's table is dropped BEFORE the clock starts (freeing it is not this call's work)
                    t0 = time.perf_counter()
                    res = api.impute_chromosome(**kw)
                    ts.append(time.perf_counter() - t0)
                st = res.stats
                py = st.get("py_ms", {})
                k = int(np.argsort(ts)[len(ts) // 2])
                per.append({"rank": r, "windows": int(st["n_windows_mine"]), "batches": int(st["n_batches"]), "imputed": int(st["imputed"]),
                            "warm_ms": ts[k] * 1e3, "warm_ms_all": [t * 1e3 for t in ts], "gpu_span_ms": float(st["gpu_span_ms"]),
                            "host_ms_not_overlapped": ts[-1] * 1e3 - float(st["gpu_span_ms"]),
                            "t_plan_ms": st["t_plan"] * 1e3, "t_feeder_wait_ms": st["t_feeder_wait"] * 1e3, "t_job_create_ms": st["t_job_create"] * 1e3,
                            "t_gpu_wait_ms": st["t_gpu_wait"] * 1e3, "t_tables_ms": st["t_tables"] * 1e3,
                            "t_tables_tail_ms": st.get("t_tables_tail", 0.0) * 1e3, "t_native_ms": st["t_total"] * 1e3,
                            "py_wrapper_ms": ts[-1] * 1e3 - st["t_total"] * 1e3, "py_columns_ms": py.get("columns")})
        finally:
            if old is None:
                os.environ.pop("LOCAL_WORLD_SIZE", None)
            else:
                os.environ["LOCAL_WORLD_SIZE"] = old
        return per
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    # (Python's cyclic collector off while these few-millisecond calls are timed, as for the jepegmix leg: the bench process holds
    # millions of objects by now and a collection inside a 5 ms call is the harness's time, not the call's)
    import gc
    gc_was = gc.isenabled()
    gc.disable()
    try:
        per = sweep(world)               # the ranks share THIS box's cores (16 for the one-GPU lease: two host threads a rank)
        ample = sweep(0)                 # a rank with eight host threads of its own (an 8-GPU node has >= 16 cores per GPU)
    finally:
        if gc_was:
            gc.enable()

    def genome(r, w, n_chrom, local_world):
        """gauss_host_impute_genome: n_chrom chromosomes (the chr22 files stand in for each), two calls in flight; ms per chromosome"""
        old = os.environ.get("LOCAL_WORLD_SIZE")
        if local_world:
            os.environ["LOCAL_WORLD_SIZE"] = str(local_world)
        try:
            kw = dict(chromosomes=[(22, lo, hi)] * n_chrom, wing_size=wing, input_file=files["gwas"], reference_data_file=files["panel"],
                      reference_pop_desc_file=files["desc"], rank=r, world=w, depth=2, ctx=rig.ctx, **sa)
            api.impute_genome(**dict(kw, chromosomes=kw["chromosomes"][:3]))       # warm-up
            t0 = time.perf_counter()
            res = api.impute_genome(**kw)
            dt = (time.perf_counter() - t0) / n_chrom
        finally:
            if old is None:
                os.environ.pop("LOCAL_WORLD_SIZE", None)
            else:
                os.environ["LOCAL_WORLD_SIZE"] = old
        return dt * 1e3, float(np.mean([q.stats["gpu_span_ms"] for q in res])), int(res[0].stats["imputed"])
    n_genome = 22                        # a genome's worth of chromosome-sized calls: the pipeline's start-up and tail are paid once
    one_ms, one_span, one_imputed = genome(0, 1, n_genome, 0)
    g_per = [genome(r, world, n_genome, world) for r in range(world)]
    g_slow = max(q[0] for q in g_per)
    slow = max(q["warm_ms"] for q in per)
    slow_a = max(q["warm_ms"] for q in ample)
    med = lambda k: float(np.median([q[k] for q in per if q.get(k) is not None]))
    phases = {
        # one rank's call, medians over the eight ranks (ms), in the order they happen; what the GPU does not overlap is 1 + 2 + 3 + 5 + 6
        "1_plan": med("t_plan_ms"), "2_data_layer_wait": med("t_feeder_wait_ms"), "3_job_tables_and_queue": med("t_job_create_ms"),
        "4_gpu_span": med("gpu_span_ms"), "5_tables_after_last_result": med("t_tables_tail_ms"), "6_python_wrapper": med("py_wrapper_ms"),
        "why": "1: windows + owners, every rank derives the plan of all 36 windows; 2: the windows' merges (0 on a repeat call: window cache) "
               "+ starting the host threads; 3: planner + work-item tables of the rank's ONE job (a rank's 4-5 windows: ~12 000 items) and "
               "queuing ~45 launches -- the GPU starts at the first launch, so ~0.15 ms of it runs beside the GPU; 5: z / info / pval of the "
               "unmeasured SNPs (pnorm on the host, two threads a rank on this box); 6: ctypes + ten numpy column copies"}
    return {"world": world, "calls_per_rank": calls, "per_rank_warm_ms": [q["warm_ms"] for q in per], "slowest": slow, "phase_ms": phases,
            "one_rank_warm_ms": warm_one_s * 1e3, "predicted_efficiency": warm_one_s * 1e3 / (world * slow),
            "host_ms_not_overlapped": [q["host_ms_not_overlapped"] for q in per],
            "host_threads_per_rank": max(1, min(8, cores // world)), "usable_cores": cores,
            "with_eight_host_threads_per_rank": {"per_rank_warm_ms": [q["warm_ms"] for q in ample], "slowest": slow_a,
                                                 "predicted_efficiency": warm_one_s * 1e3 / (world * slow_a),
                                                 "host_ms_not_overlapped": [q["host_ms_not_overlapped"] for q in ample]},
            "genome_pipeline": {
                "what": "gauss_host_impute_genome: chromosome after chromosome with TWO calls in flight on the rank's context (the chr22 files "
                        "stand in for every one of %d chromosomes), a share of <= 8 windows as one batch: one call's host part runs under the "
                        "other's GPU work" % n_genome,
                "chromosomes": n_genome,
                "per_rank_ms_per_chromosome": [q[0] for q in g_per], "per_rank_gpu_span_ms": [q[1] for q in g_per], "slowest": g_slow,
                "one_rank_ms_per_chromosome": one_ms, "one_rank_gpu_span_ms": one_span,
                "predicted_efficiency": one_ms / (world * g_slow),
                "imputed_per_chromosome": one_imputed, "imputed_snps_per_s_one_rank": one_imputed / (one_ms * 1e-3),
                "imputed_snps_per_s_predicted_world": one_imputed / (g_slow * 1e-3)},
            "imputed_total": int(sum(q["imputed"] for q in per)), "per_rank": per,
            "note": "every rank's share of gauss_host_impute_chromosome(rank, world) timed alone on one GPU, panel resident; the headline "
                    "figures give a rank 1/world of this box's cores (LOCAL_WORLD_SIZE = world, as under torchrun on one node), "
                    "with_eight_host_threads_per_rank what a rank gets on a node with >= 16 cores per GPU; host_ms_not_overlapped is the "
                    "LAST call's wall time minus its GPU span"}


def other_configs_block(args, rig, ch, files, tmp):
    """BASELINE.json configs[1], [2] and [4] in the default line, at most five timed steps each (the headline stays configs[3]):
    computeLD (the 3 Mb window, 32 copies batched), dist (EUR, N = 20 281) and jepegmix (350 genes from files), each with its
    value, roofline, a bounded cpu_baseline sample and a parity_spot against the CPU oracle."""
    import argparse
    import bench
    out = {}
    t_all = time.perf_counter()
    # configs[1]
    t0 = time.perf_counter()
    a = argparse.Namespace(**dict(vars(args), steps=5, warmup=2, mode="computeLD"))
    line, sample = run_computeld(a, rig)
    line["cpu_baseline"] = bench.cpu_baseline_computeld(sample)
    d = float(np.max(np.abs(sample["gpu_ld"] - sample["oracle_ld"])))
    line["parity_spot"] = {"snps": int(sample["gpu_ld"].shape[0]), "max_abs_ld_diff": d, "tolerance": 1e-12, "ok": bool(d <= 1e-12),
                           "what": "the first 160 SNPs' block of the GPU's LD matrix (blocking gauss_ld on host bytes; the resident and "
                                   "batched forms return the same bits: config.results_identical_across_forms) vs oracle.compute_ld"}
    line["seconds"] = time.perf_counter() - t0
    out["computeLD"] = {k: line[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "config", "roofline", "forms", "cpu_baseline",
                                             "parity_spot", "seconds")}
    # configs[2]
    t0 = time.perf_counter()
    a = argparse.Namespace(**dict(vars(args), steps=5, warmup=2, mode="dist", no_i8_variant=True, no_e2e=True, emulate_world=0,
                                  no_tails_alone=True, no_other_configs=True, streams=1))
    line = bench.run_impute(a, rig, quiet=True, light=True)
    line["seconds"] = time.perf_counter() - t0
    out["dist"] = {k: line[k] for k in ("value", "unit", "ms_per_step", "steps", "config", "roofline", "launch_form", "cpu_baseline", "parity_spot",
                                        "seconds") if k in line}
    out["dist"]["metric"] = "dist(study_pop=EUR) imputed SNPs/s, chr22, one GPU (BASELINE.json configs[2])"
    # configs[4]
    t0 = time.perf_counter()
    line = jepegmix_measure(rig, ch, files, tmp, 5, checks=bench.jepegmix_checks)
    line["seconds"] = time.perf_counter() - t0
    out["jepegmix"] = {k: line[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "config", "breakdown", "roofline", "cpu_baseline",
                                            "parity_spot", "seconds", "timed_with_gc_off", "emulated_world8") if k in line}
    out["seconds_total"] = time.perf_counter() - t_all
    out["all_parity_ok"] = all(bool(out[k].get("parity_spot", {}).get("ok", False)) for k in ("computeLD", "dist", "jepegmix"))
    return out


def from_text_block(rig, ch, files, tmp, wing, n_snp=100_000, warm_runs=3):
    """The reference's OWN on-disk format at chromosome scale (gauss.cpp:293-399, 720-785: BGZF text index + data, one ~33 kB
    line per SNP): the first `n_snp` SNPs of the study's panel written as a BGZF text panel (all 29 populations, N = 32 953),
    then text -> packed panel in the cache (the feeder: inflate + parse + 2-bit pack) -> upload -> distmix over its windows,
    cold and warm.  Harness only: the genotype rows are the ones write_study_files made."""
    import zlib
    n = int(min(n_snp, len(ch["bp"])))
    pops_all = files["pops_all"]
    sizes = [q[1] for q in pops_all]
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    idx, dat = os.path.join(tmp, "text_index.gz"), os.path.join(tmp, "text_data.gz")
    rows2 = files["rows2bit"]
    inflated = panel.write_panel_fast(idx, dat, files["rsid"][:n], np.full(n, 22), ch["bp"][:n], files["a1"][:n], files["a2"][:n],
                                      lambda a, b: panel.unpack2bit(rows2[a:b], sizes), files["af"][:n], sizes, threads=min(16, cores))
    m = np.nonzero(ch["measured"][:n])[0]
    gwas = os.path.join(tmp, "text_gwas.txt")
    panel.write_gwas(gwas, files["rsid"][m], np.full(len(m), 22), ch["bp"][m], files["a1"][m], files["a2"][m], ch["z"][m])
    make_s = time.perf_counter() - t0
    # what plain zlib inflates per core (the feeder's inflate is this plus parsing and packing): the first 400 members, on one
    # thread with the box to itself, and on as many threads at once as the feeder uses (Python's zlib releases the GIL): the
    # feeder's per-thread rate is measured under that load, the box's 16 logical cores are not 16 times one of them
    raw = open(dat, "rb").read(64 << 20)
    parts, pos = [], 0
    while pos + 18 <= len(raw) and len(parts) < 400:
        bsize = int.from_bytes(raw[pos + 16:pos + 18], "little") + 1
        if pos + bsize > len(raw):
            break
        parts.append(raw[pos + 18:pos + bsize - 8])
        pos += bsize

    def inflate_all(_=None):
        return sum(len(zlib.decompress(m, -15)) for m in parts)
    t0 = time.perf_counter()
    got = inflate_all()
    zlib_rate = got / max(time.perf_counter() - t0, 1e-9)
    from concurrent.futures import ThreadPoolExecutor
    nthr = min(16, cores)
    with ThreadPoolExecutor(max_workers=nthr) as pool:
        t0 = time.perf_counter()
        got_n = sum(pool.map(inflate_all, range(nthr)))
        zlib_rate_loaded = got_n / nthr / max(time.perf_counter() - t0, 1e-9)
    old_cache = os.environ.get("GAUSS_PANEL_CACHE")
    os.environ["GAUSS_PANEL_CACHE"] = os.path.join(tmp, "panel_cache")
    try:
        sa = study_args(ch, files)
        lo, _ = chromosome_span(ch)
        kw = dict(chr=22, start_bp=lo, end_bp=int(ch["bp"][n - 1]), wing_size=wing, input_file=gwas, reference_index_file=idx,
                  reference_data_file=dat, reference_pop_desc_file=files["desc"], ctx=rig.ctx, **sa)

        def once():
            t = time.perf_counter()
            r = api.impute_chromosome(**kw)
            return time.perf_counter() - t, r
        api.panel_evict(ctx=rig.ctx)
        t0 = time.perf_counter()
        packed_path, n_packed = api.panel_cache(idx, dat, files["desc"])          # the feeder: text -> packed panel (first use)
        pack_s = time.perf_counter() - t0
        cold_s, cold = once()                                                   # cached, not resident: upload + impute
        warm = []
        res = cold
        for _ in range(warm_runs):
            t, res = once()
            warm.append(t)
        api.panel_evict(ctx=rig.ctx)
    finally:
        if old_cache is None:
            os.environ.pop("GAUSS_PANEL_CACHE", None)
        else:
            os.environ["GAUSS_PANEL_CACHE"] = old_cache
    threads = min(16, cores)
    imputed = int(res.stats["imputed"])
    z = res.columns["z"]
    return {
        "what": f"BGZF text panel in the reference's format ({n} SNPs x {int(sum(sizes))} samples, 29 populations: "
                f"{inflated / 1e6:.0f} MB of text in {(os.path.getsize(dat) + os.path.getsize(idx)) / 1e6:.0f} MB of BGZF) + GWAS text file -> "
                "gauss_host_impute_chromosome: packed into the panel cache on first use, uploaded, distmix over every 1 Mb window",
        "snps": n, "samples": int(sum(sizes)), "inflated_text_bytes": int(inflated), "packed_snps": int(n_packed),
        "pack_s": pack_s, "pack_threads": threads,
        "feeder_inflated_MB_per_s": inflated / pack_s / 1e6, "feeder_inflated_MB_per_s_per_core": inflated / pack_s / 1e6 / threads,
        "zlib_inflate_alone_MB_per_s_one_core": zlib_rate / 1e6,
        "zlib_inflate_MB_per_s_per_thread_all_threads_busy": zlib_rate_loaded / 1e6,
        "feeder_over_zlib_per_core": (inflated / pack_s / threads) / zlib_rate_loaded if zlib_rate_loaded > 0 else None,
        "feeder_over_zlib_alone_one_core": (inflated / pack_s / threads) / zlib_rate if zlib_rate > 0 else None,
        "ms_per_snp_line": pack_s / n * 1e3,
        "cold_from_text_s": pack_s + cold_s, "cold_cached_not_resident_s": cold_s, "warm_s_median": float(np.median(warm)), "warm_s_all": warm,
        "imputed_snps": imputed, "table_rows": int(len(z)), "all_finite": bool(np.all(np.isfinite(z))),
        "same_table_cold_and_warm": bool(np.array_equal(cold.columns["z"], res.columns["z"])),
        "imputed_snps_per_s_cold_from_text": imputed / (pack_s + cold_s), "imputed_snps_per_s_warm": imputed / float(np.median(warm)),
        "make_files_s": make_s,
        "note": "pack_s is paid once per panel (the cache is keyed by the three files' identity); zlib_inflate_* is Python's zlib on the "
                "same members, no parsing: one thread with the box to itself, and per thread with as many threads inflating at once as the "
                "feeder runs (feeder_over_zlib_per_core compares the feeder's per-thread rate with THAT figure: both under the same load)",
    }


def e2e_block(args, rig, ch=None, steps=5):
    """The end_to_end object of the bench line: a chr22-sized packed panel FILE (100 000 SNPs x 32 953 samples, 29
    populations) and the GWAS text file on disk -> distmix over all 1 Mb windows -> one result table.  Never the
    headline `value`: it is reported beside it."""
    if ch is None:
        ch = workload.make_chromosome(args.snps, "distmix", seed=20260216, sample_scale=args.sample_scale)
    tmp = tempfile.mkdtemp(prefix="gauss_e2e_")
    blk = None
    try:
        t0 = time.perf_counter()
        files = write_study_files(rig, ch, tmp) if rig.rank == 0 else None
        if rig.world > 1:
            paths = rig.gather({k: v for k, v in files.items() if isinstance(v, (str, int))} if files else None)[0]
            files = files or paths
        make_s = time.perf_counter() - t0
        cold_s, cold, warm, res = e2e_measure(rig, ch, files, steps, args.wing)
        if rig.rank == 0:
            warm_s = float(np.median(warm))
            st = res.stats
            tot = lambda k: (float(np.sum(st[k])) if isinstance(st[k], list) else float(st[k]))
            span = (max(st["gpu_span_ms"]) if isinstance(st["gpu_span_ms"], list) else st["gpu_span_ms"])
            z = res.columns["z"]
            blk = {
                "what": f"packed panel file ({files['panel_bytes'] / 1e6:.0f} MB: {len(ch['bp'])} SNPs x {files['samples_in_file']} samples, "
                        f"29 populations) + GWAS text file ({int(ch['measured'].sum())} SNPs) on disk -> distmix over every 1 Mb window -> "
                        "one result table (gauss_host_impute_chromosome: host data layer, upload, GPU pipeline, tables)",
                "imputed_snps": int(tot("imputed")), "table_rows": int(len(z)), "all_finite": bool(np.all(np.isfinite(z))),
                "windows": int(np.max(st["n_windows"]) if isinstance(st["n_windows"], list) else st["n_windows"]),
                "warm_s_median": warm_s, "warm_s_all": warm, "imputed_snps_per_s_warm": tot("imputed") / warm_s,
                "cold_s": cold_s, "imputed_snps_per_s_cold": tot("imputed") / cold_s, "cold_over_warm": cold_s / warm_s,
                "gpu_span_ms": span, "warm_over_gpu_span": warm_s * 1e3 / span if span else None,
                "warm_definition": "panel rows already resident in HBM (a session imputing study after study); cold = first call, "
                                   "panel upload through pinned double buffers included (the rows travel while the first batches compute on what has landed)",
                "stats_last_warm_run": st, "stats_cold_run": cold.stats if cold is not None else None,
                "make_files_s": make_s,
            }
            if rig.world == 1:
                blk["emulated_world8"] = emulate_world_e2e(rig, ch, files, args.wing, warm_s, world=8)
            if rig.world == 1 and not getattr(args, "no_from_text", False):
                blk["from_text"] = from_text_block(rig, ch, files, tmp, args.wing, n_snp=getattr(args, "text_snps", 100_000))
            if rig.world == 1 and not getattr(args, "no_other_configs", False) and getattr(args, "mode", "") == "distmix":
                blk["_other_configs"] = other_configs_block(args, rig, ch, files, tmp)
    finally:
        rig.barrier()
        if rig.rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)
    return blk


def run_e2e(args, rig):
    """--mode e2e: the end_to_end block as a bench line of its own (also with --gpus N: every rank takes its LPT
    share of the windows, rank 0 merges the tables)."""
    ch = workload.make_chromosome(args.snps, "distmix", seed=20260216, sample_scale=args.sample_scale)
    blk = e2e_block(args, rig, ch, steps=args.steps if args.steps < 50 else 10)
    out = None
    if rig.rank == 0:
        out = {
            "metric": "imputed SNPs/sec end to end: packed panel file + GWAS file on disk -> distmix result table",
            "value": blk["imputed_snps_per_s_warm"], "unit": "imputed SNPs/s", "n_gpus": rig.world, "steps": len(blk["warm_s_all"]),
            "warmup": 1, "ms_per_step": blk["warm_s_median"] * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "distmix() chr22 end to end (BASELINE.json configs[3] from files): " + blk["what"]},
            "end_to_end": blk,
        }
        import bench
        bench.emit_line(out, headline=False)
    return out


def _time_job(rig, job, steps, warmup):
    for _ in range(max(1, warmup)):
        job.run(); res = job.fetch()
    job.profile(True)
    rig.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        job.run(); res = job.fetch()
    rig.barrier()
    dt = (time.perf_counter() - t0) / steps
    stage = {k: job.profile_get(i) for i, k in enumerate(("gram", "pack_stats", "ld_epilogue", "factor", "solve"))}
    job.profile(False)
    return dt, stage, res


def run_computeld(args, rig):
    """--mode computeLD (BASELINE.json configs[1]): the LD matrix of the chr10 104-107 Mb window of the reference's
    3 Mb study file (tests/golden/PGC2_3Mb.txt: 529 of its 721 SNPs) against a 33KG-shaped panel, PGC2 weights,
    N = 32 147.  Three figures: the blocking gauss_ld call on host bytes (what the Rcpp driver binds; PCIe inclusive),
    one window over resident 2-bit rows, and 32 such windows batched into one job (the form that fills the chip).
    value = LD-GEMM TFLOP/s of the batched form (algorithmic N M (M+1) per window / Gram kernel time)."""
    torch, ctx = rig.torch, rig.ctx
    study = os.path.join(workload.ROOT, "tests", "golden", "PGC2_3Mb.txt")
    _, bp, _, _, _ = workload.read_study(study)
    ch = workload.make_chromosome(100_000, "distmix", seed=20260214, sample_scale=args.sample_scale, study=study)
    # only the study's SNPs matter for computeLD (measured SNPs of the window, computeLD.cpp:80-86)
    keep = np.nonzero(ch["measured"])[0]
    sub = dict(ch, bp=ch["bp"][keep], thr=np.ascontiguousarray(ch["thr"][keep]), z=ch["z"][keep],
               measured=ch["measured"][keep])
    rho = np.ones(len(keep), dtype=np.float32)
    rho[1:] = np.exp(-np.diff(sub["bp"]) / 50e3)
    sub["rho"] = rho
    import bench
    panel_u8, ld = bench.synth_panel(rig, sub, 20260214)
    store, ld2 = bench.pack_store(rig, sub, panel_u8, ld)
    N = int(ch["off"][-1])
    rows = np.nonzero((sub["bp"] >= 104_000_001) & (sub["bp"] <= 107_000_000))[0].astype(np.int32)
    M = len(rows)
    host = np.ascontiguousarray(panel_u8.index_select(0, torch.from_numpy(rows.astype(np.int64)).cuda())[:, :N].cpu().numpy())
    del panel_u8

    def desc():
        return dict(mode=hotpath.MODE_WEIGHTED, pop_off=ch["off"], pop_wgt=ch["w"], z1=np.zeros(M), lam=0.0,
                    dev=(store.data_ptr(), store.data_ptr(), M, 0, ld2), ld_codings=_lib.CODE_ADDITIVE,
                    packed=dict(fmt=1, rows_m=rows, rows_u=np.zeros(0, np.int32)))
    flops = float(N) * M * (M + 1)
    steps = min(args.steps, 50)
    # (a) blocking host-pointer call
    hotpath.ld_matrix(host, ch["off"], ch["w"], ctx=ctx)
    t0 = time.perf_counter()
    for _ in range(5):
        ref = hotpath.ld_matrix(host, ch["off"], ch["w"], ctx=ctx)
    t_block = (time.perf_counter() - t0) / 5
    # (b) one resident window, (c) 32 resident windows in one job
    one = hotpath.Job([desc()], ctx=ctx, on_device=True)
    dt1, st1, r1 = _time_job(rig, one, steps, args.warmup)
    B = 32
    # (the 32 windows are 32 copies of the config's one window: a job would recognise their common measured rows and multiply the
    # B11 tile pairs ONCE for all of them -- shared measured rows, DESIGN.md section 3 -- which is not what 32 different
    # computeLD() windows cost; built with sharing off, every copy packs and multiplies its own rows)
    old_share = os.environ.get("GAUSS_SHARE_MEASURED")
    os.environ["GAUSS_SHARE_MEASURED"] = "0"
    try:
        many = hotpath.Job([desc() for _ in range(B)], ctx=ctx, on_device=True)
    finally:
        if old_share is None:
            del os.environ["GAUSS_SHARE_MEASURED"]
        else:
            os.environ["GAUSS_SHARE_MEASURED"] = old_share
    dtb, stb, rb = _time_job(rig, many, steps, args.warmup)
    ok = bool(np.array_equal(r1[0]["b11"], ref) and all(np.array_equal(x["b11"], ref) for x in rb) and
              np.allclose(np.diag(ref), 1.0) and np.array_equal(ref, ref.T))
    g1 = st1["gram"][0] / max(1, st1["gram"][1]) * 1e-3
    gb = stb["gram"][0] / max(1, stb["gram"][1]) * 1e-3
    out = None
    if rig.rank == 0:
        ach = B * flops / gb / 1e12
        out = {
            "metric": "computeLD() LD-GEMM MFMA TFLOP/s vs peak (BASELINE.json configs[1])",
            "value": ach, "unit": "TFLOP/s", "n_gpus": 1, "steps": steps, "warmup": args.warmup, "ms_per_step": dtb * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"computeLD() chr10:104-107 Mb, {M} of the 721 SNPs of tests/golden/PGC2_3Mb.txt x {N} samples "
                                   f"(21 populations, PGC2 weights), weighted LD (computeLD.cpp:95-116); batched form = {B} such windows in one job",
                       "M": M, "samples": N, "batch": B, "results_identical_across_forms": ok},
            "roofline": {"kernel": "gram_kernel (LD GEMM, v_mfma_f32_32x32x2_f32), batched form", "bound": "mfma", "achieved": ach,
                         "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                         "algorithmic_flops_per_launch": B * flops, "avg_launch_ms": gb * 1e3},
            "forms": {
                "blocking_gauss_ld_host_bytes": {"ms_per_call": t_block * 1e3, "ld_matrices_per_s": 1.0 / t_block,
                                                 "note": f"{host.nbytes / 1e6:.1f} MB of genotype bytes over PCIe per call, {M * M * 8 / 1e6:.1f} MB back"},
                "one_resident_window": {"ms_per_step": dt1 * 1e3, "gram_ms": g1 * 1e3, "gram_tflops": flops / g1 / 1e12,
                                        "frac": flops / g1 / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                        "stage_ms": {k: v[0] / steps for k, v in st1.items()}},
                "batched_resident_windows": {"windows": B, "ms_per_step": dtb * 1e3, "gram_ms": gb * 1e3, "gram_tflops": ach,
                                             "ld_matrices_per_s": B / dtb, "stage_ms": {k: v[0] / steps for k, v in stb.items()}},
            },
        }
    one.close(); many.close()
    sample = dict(kind="computeLD", geno=host[:160], off=ch["off"], w=ch["w"], M=M, N=N, batch=B, gpu_ld=ref[:160, :160].copy())
    return out, sample


def make_annotation(ch, files, outdir, n_genes=350, seed=9):
    """A JEPEG annotation in the documented 8-column format (gauss.cpp:1308,1319-1330): n_genes genes, 1-20 measured SNPs
    each (consecutive measured SNPs around a random locus), 1-2 of the 6 functional categories per SNP, random weights
    -- the stand-in for the missing JEPEG_SNP_Annotation.v1.0.txt."""
    rng = np.random.default_rng(seed)
    m = np.nonzero(ch["measured"])[0]
    rows = []
    starts = np.sort(rng.choice(len(m) - 20, size=n_genes, replace=False))
    n_snp = 0
    for g, s0 in enumerate(starts):
        k = int(rng.integers(1, 21))
        for s in m[s0:s0 + k]:
            for c in rng.choice(6, size=int(rng.integers(1, 3)), replace=False):
                rows.append((files["rsid"][s], 22, int(ch["bp"][s]), files["a1"][s], files["a2"][s], f"GENE{g:04d}", panel.CATEGS[c],
                             float(np.round(rng.uniform(0.2, 2.0), 3))))
        n_snp += k
    path = os.path.join(outdir, "annot.txt")
    panel.write_annotation(path, rows)
    return path, n_genes, n_snp


def run_jepegmix(args, rig, checks=None):
    """--mode jepegmix (BASELINE.json configs[4]): jepegmix() over ~350 synthetic genes (1-20 SNPs each) of the chr22
    study, PGC2 weights, N = 32 147, from files: packed panel + GWAS + annotation -> gene table.  The GPU part is the
    batched gene LD (pack -> Gram on the tile pairs genes touch -> gene epilogue); the k x k tail (k <= 6) runs on the
    host.  value = genes/s of the whole call (host bound, as SURVEY.md section 8d expects); the roofline entry is the
    gene epilogue's byte roofline with the algorithmic bytes sum n_g N (2-bit: / 4) in + sum n_g^2 8 out."""
    ch = workload.make_chromosome(args.snps, "distmix", seed=20260216, sample_scale=args.sample_scale)
    tmp = tempfile.mkdtemp(prefix="gauss_jepeg_")
    out = None
    try:
        files = write_study_files(rig, ch, tmp)
        api.panel_evict(ctx=rig.ctx)                               # breakdown.cold_call_s includes the panel's upload
        out = jepegmix_measure(rig, ch, files, tmp, min(args.steps, 10), checks=checks)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out, None


def jepegmix_measure(rig, ch, files, tmp, steps, checks=None, spot_genes=24):
    """jepegmix() on the study files `files` (write_study_files) with a synthetic annotation written into `tmp`.
    checks: bench.py's jepegmix_checks (the CPU checker lives outside the package): given the prepared object and the GPU's gene
    LD blocks it returns the `cpu_baseline` and `parity_spot` entries of the line."""
    out = None
    annot, n_genes, _ = make_annotation(ch, files, tmp)
    wgt = (list(synth.PGC2_WEIGHTS.keys()), list(synth.PGC2_WEIGHTS.values()))
    kw = dict(pop_wgt_df=wgt, input_file=files["gwas"], annotation_file=annot, reference_index_file="(packed)",
              reference_data_file=files["panel"], reference_pop_desc_file=files["desc"], ctx=rig.ctx)
    t0 = time.perf_counter()
    tab = api.jepegmix(**kw)
    cold = time.perf_counter() - t0
    # (the cyclic collector off while the calls are timed, as `timeit` does: at the end of the default run the process holds a few
    # million Python objects -- the text leg's SNP names, the CPU baseline's samples -- and a generation-2 pass over them inside a
    # 3 ms call is not the call's time: 5.2 ms against 3.3 measured with it on)
    import gc
    gc_was = gc.isenabled()
    gc.disable()
    try:
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            tab = api.jepegmix(**kw)
            ts.append(time.perf_counter() - t0)
    finally:
        if gc_was:
            gc.enable()
    warm = float(np.median(ts))
    emu = jepeg_emulate_world(rig, kw, tab, warm, steps)
    # the host data layer alone (no GPU call): what bounds the run
    t0 = time.perf_counter()
    pr = api.Prepared(api.KIND_JEPEGMIX, **{k: v for k, v in kw.items() if k != "ctx"})
    t_host = time.perf_counter() - t0
    go = pr.gene_off()
    sizes = np.diff(go).astype(np.int64)
    S, N = pr.M, pr.N
    # the GPU part alone, resident rows, through the C ABI entry point the driver uses
    gpu = gene_batch_gpu_time(rig, files, pr, ch, steps)
    extra = {}
    if checks is not None and rig.rank == 0:
        extra = checks(pr, gpu["blocks"], go, sizes, spot_genes, len(tab), warm)
    pr.close()
    if rig.rank == 0:
        bytes_in = float(sizes.sum()) * N / 4.0
        bytes_out = float((sizes * sizes).sum()) * 8.0
        out = {
            "metric": "jepegmix() genes/sec from files (BASELINE.json configs[4])",
            "value": len(tab) / warm, "unit": "genes/s", "n_gpus": 1, "steps": steps, "warmup": 1, "ms_per_step": warm * 1e3,
            "timed_with_gc_off": True,          # Python's cyclic collector is off while the calls are timed (as `timeit` does; see above)
            "emulated_world8": emu,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"jepegmix() chr22: {n_genes} synthetic genes (1-20 SNPs, {int(sizes.sum())} gene SNPs after the AF filter) from "
                                   f"{ch['study']}, PGC2 weights, N = {N}; packed panel ({files['panel_bytes'] / 1e6:.0f} MB) + GWAS + annotation files -> gene table",
                       "genes_in_table": int(len(tab)), "gene_snps": int(sizes.sum()), "samples": N,
                       "all_finite_pvals": bool(np.all(np.isfinite(tab["jepeg_pval"].to_numpy()[tab["df"].to_numpy() > 0])))},
            "breakdown": {"cold_call_s": cold, "warm_call_s_median": warm, "host_data_layer_s": t_host,
                          "gpu_gene_ld_batch_ms": gpu["ms_per_call"], "note": "host bound: the SNP map and the annotation merge dominate; host_data_layer_s is gauss_host_prepare "
                          "on the WHOLE study (what the reference builds); the driver itself enters only the study SNPs at positions the annotation "
                          "names (host_calls.cpp:run_jepeg, ~1.0 ms of the call); the GPU batch is pack + Gram on the tile pairs that genes touch"},
            "roofline": {"kernel": "gauss_gene_ld_batch_rows (pack_stats + gram on gene tile pairs + gene_epilogue_kernel)", "bound": "hbm",
                         "achieved": (bytes_in + bytes_out) / (gpu["ms_per_call"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (bytes_in + bytes_out) / (gpu["ms_per_call"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes": bytes_in + bytes_out,
                         "note": "algorithmic bytes = sum n_g N / 4 (2-bit rows) in + sum n_g^2 8 out; the call is host and latency bound "
                                 "(a handful of tile pairs; the whole GPU batch is a tenth of the call), not bandwidth bound: the fraction "
                                 "says how little of the chip a 350-gene chromosome can use, not how good the kernels are"},
        }
        out.update(extra)
    return out


def jepeg_emulate_world(rig, kw, tab_one, warm_one_s, steps, world=8, n_calls=22):
    """configs[4] names 8 GPUs.  Two splits exist: (a) the GENES of one call over the ranks
    (gauss_host_jepeg_rank: every rank repeats the host data layer, which is most of the call, and computes CorG + tails of its
    contiguous gene range) and (b) WHOLE calls dealt to the ranks (gauss_host_jepeg_genome: one call per chromosome file set).  Both
    emulated on the one GPU, every rank's share timed alone, warm, collector off, median of `steps` calls: predicted efficiency =
    one-rank time / (world x the slowest rank's).  Stated as measured -- (a) is far below 0.5 because the call is host-bound."""
    import gc
    import pandas as pd
    args = {k: v for k, v in kw.items() if k not in ("pop_wgt_df", "ctx")}
    gc_was = gc.isenabled()
    gc.disable()
    try:
        per, parts = [], []
        for r in range(world):
            ts = []
            for _ in range(max(3, steps)):
                t0 = time.perf_counter()
                t, rng = api.jepeg_rank(api.KIND_JEPEGMIX, pop_wgt_df=kw["pop_wgt_df"], rank=r, world=world, ctx=rig.ctx, **args)
                ts.append(time.perf_counter() - t0)
            per.append(float(np.median(ts)) * 1e3)
            parts.append(t)
        cat = pd.concat(parts, ignore_index=True)
        same = bool(list(cat.columns) == list(tab_one.columns) and len(cat) == len(tab_one) and
                    all((np.array_equal(cat[c].to_numpy().view(np.uint64), tab_one[c].to_numpy().view(np.uint64))
                         if cat[c].dtype.kind == "f" else list(cat[c]) == list(tab_one[c])) for c in cat.columns))
        # (b) n_calls chromosome-sized calls (the chr22 files stand in for each), whole calls per rank
        calls = [(kw["input_file"], kw["annotation_file"], kw["reference_index_file"], kw["reference_data_file"])] * n_calls
        def genome(r, w):
            t0 = time.perf_counter()
            tabs, owner = api.jepeg_genome(api.KIND_JEPEGMIX, calls, kw["reference_pop_desc_file"], pop_wgt_df=kw["pop_wgt_df"], rank=r, world=w,
                                           ctx=rig.ctx)
            return (time.perf_counter() - t0) * 1e3, sum(1 for t in tabs if t is not None)
        genome(0, world)                                                            # warm-up
        g_one, _ = genome(0, 1)
        g_per = [genome(r, world) for r in range(world)]
    finally:
        if gc_was:
            gc.enable()
    slow = max(per)
    g_slow = max(q[0] for q in g_per)
    return {"world": world, "split": "contiguous gene ranges of ONE call (gauss_host_jepeg_rank); every rank repeats the host data layer",
            "per_rank_ms": per, "slowest_ms": slow, "one_rank_ms": warm_one_s * 1e3, "predicted_efficiency": warm_one_s * 1e3 / (world * slow),
            "tables_identical_to_one_rank": same,
            "whole_calls": {"split": "whole calls dealt to the ranks (gauss_host_jepeg_genome), %d chromosome-sized calls" % n_calls,
                            "calls": n_calls, "calls_per_rank": [q[1] for q in g_per], "per_rank_ms": [q[0] for q in g_per], "slowest_ms": g_slow,
                            "one_rank_ms": g_one, "predicted_efficiency": g_one / (world * g_slow)},
            "note": "every rank's share timed alone on ONE GPU, warm, Python's collector off: an emulation, not a multi-GPU measurement"}


def gene_batch_gpu_time(rig, files, pr, ch, steps):
    """gauss_gene_ld_batch_rows alone on the resident panel (what run_jepeg calls between the data layer and the tail)."""
    ctx = rig.ctx
    api.panel_resident(files["panel"], ctx=ctx)
    d = pr.window_rows(files["panel"], ctx)
    go = np.ascontiguousarray(pr.gene_off(), dtype=np.int32)
    sizes = np.diff(go).astype(np.int64)
    outb = np.zeros(int((sizes * sizes).sum()))
    ip = C.POINTER(C.c_int32)
    dp = C.POINTER(C.c_double)
    po, w = np.ascontiguousarray(pr.pop_off(), np.int32), np.ascontiguousarray(pr.pop_wgt(), np.float64)

    def call():
        _lib.check(ctx.lib.gauss_gene_ld_batch_rows(ctx.handle, hotpath.MODE_WEIGHTED, d["store"], d["ld"], _lib.GENO_2BIT,
                                                    d["rows_m"].ctypes.data_as(ip), pr.M, po.ctypes.data_as(ip),
                                                    d["pop_src_off"].ctypes.data_as(ip), w.ctypes.data_as(dp), len(w),
                                                    go.ctypes.data_as(ip), len(go) - 1, 1.1, 1, outb.ctypes.data_as(dp)))
    call()
    t0 = time.perf_counter()
    for _ in range(max(1, steps)):
        call()
    dt = (time.perf_counter() - t0) / max(1, steps)
    blocks, o = [], 0
    for n in sizes:
        blocks.append(outb[o:o + n * n].reshape(n, n))
        o += n * n
    return {"ms_per_call": dt * 1e3, "blocks": blocks}


# ------------------------------------------------------------------------------------------------------------
# --mode window: the call the Rcpp drop-in makes -- one window, genotype bytes in host memory
# ------------------------------------------------------------------------------------------------------------
PCIE_PEAK_GBS = 63.0          # PCIe 5.0 x16, one direction (64 GT/s x 16 lanes, 128b/130b)


def run_window(args, rig):
    """--mode window: blocking gauss_impute_window on HOST genotype bytes (what src/dist.cpp / distmix.cpp bind after
    ReadGenotype, gauss.cpp:720-785): job build, host -> HBM, every kernel, z / info back.  Measured on the largest
    and on a mean-sized window of the chr22 study, with the genotype matrices in pageable and in pinned host memory,
    next to (a) the same window over rows already resident in HBM (compute only) and (b) a bare host -> device copy of
    the same bytes (transfer only), which bound the call from below: t_call >= max(a, b), and a perfect pipeline
    reaches it.  value = imputed SNPs / s of the pageable call on the mean-sized window (never the headline)."""
    import bench
    torch, ctx = rig.torch, rig.ctx
    ch = workload.make_chromosome(args.snps, "distmix", seed=20260216, sample_scale=args.sample_scale)
    wins = workload.windows_of(ch, args.wing, 0)
    N = int(ch["off"][-1])
    panel_u8, ld = bench.synth_panel(rig, ch, 20260216)
    store, ld2 = bench.pack_store(rig, ch, panel_u8, ld)
    cost = [workload.window_flops(N, len(mi), len(ui)) for _, mi, ui in wins]
    k_max = int(np.argmax([len(mi) for _, mi, _ in wins]))
    k_mean = int(np.argmin(np.abs(np.array(cost) - np.mean(cost))))
    reps = 7
    forms = {}
    for label, k in (("largest_window", k_max), ("mean_window", k_mean)):
        _, mi, ui = wins[k]
        M, U = len(mi), len(ui)
        gm = np.ascontiguousarray(panel_u8.index_select(0, torch.from_numpy(mi).cuda())[:, :N].cpu().numpy())
        gu = np.ascontiguousarray(panel_u8.index_select(0, torch.from_numpy(ui).cuda())[:, :N].cpu().numpy())
        z1 = ch["z"][mi]
        nbytes = gm.nbytes + gu.nbytes

        def call(a, b):
            hotpath.impute_window(1, a, b, ch["off"], ch["w"], z1, ctx=ctx)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                r = hotpath.impute_window(1, a, b, ch["off"], ch["w"], z1, ctx=ctx)
                ts.append(time.perf_counter() - t0)
            return float(np.median(ts)), r
        t_page, r_page = call(gm, gu)
        pm, pu = hotpath.PinnedArray(gm.shape, ctx=ctx), hotpath.PinnedArray(gu.shape, ctx=ctx)
        pm.array[:] = gm
        pu.array[:] = gu
        t_pin, r_pin = call(pm.array, pu.array)
        # (a) compute only: the same window over the resident 2-bit store
        job = hotpath.Job(bench.window_descs(ch, [wins[k]], store, ld2, "distmix"), ctx=ctx, on_device=True)
        dt_res, st_res, r_res = _time_job(rig, job, 20, 3)
        job.close()
        # (b) transfer only: a bare copy of the same bytes, pinned and pageable
        dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        tp = torch.from_numpy(np.concatenate([pm.array.reshape(-1), pu.array.reshape(-1)])).pin_memory()
        tg = torch.from_numpy(np.concatenate([gm.reshape(-1), gu.reshape(-1)]))

        def copy(src):
            ts = []
            for _ in range(reps + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                dev.copy_(src, non_blocking=False)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            return float(np.median(ts[1:]))
        t_cp_pin, t_cp_page = copy(tp), copy(tg)
        del dev, tp, tg
        pm.close(); pu.close()
        same = bool(np.array_equal(r_page["z"], r_pin["z"]) and np.array_equal(r_page["info"], r_pin["info"]) and
                    np.array_equal(r_page["z"], r_res[0]["z"]) and np.array_equal(r_page["info"], r_res[0]["info"]))

        def overlap(t_call, t_copy):
            lo = max(t_copy, dt_res)
            return float(np.clip((t_copy + dt_res - t_call) / max(min(t_copy, dt_res), 1e-9), 0.0, 1.0)), lo
        ov_pin, lo_pin = overlap(t_pin, t_cp_pin)
        ov_page, lo_page = overlap(t_page, t_cp_page)
        forms[label] = {
            "window": {"index": k, "M": M, "U": U, "N": N, "genotype_bytes": int(nbytes)},
            "pageable_call_ms": t_page * 1e3, "pinned_call_ms": t_pin * 1e3,
            "imputed_snps_per_s_pageable": U / t_page, "imputed_snps_per_s_pinned": U / t_pin,
            "compute_only_resident_ms": dt_res * 1e3, "compute_stage_ms": {kk: v[0] / 20 for kk, v in st_res.items()},
            "copy_only_ms": {"pinned": t_cp_pin * 1e3, "pageable": t_cp_page * 1e3},
            "h2d_GBs_of_copy_only": {"pinned": nbytes / t_cp_pin / 1e9, "pageable": nbytes / t_cp_page / 1e9},
            "effective_GBs_of_call": {"pinned": nbytes / t_pin / 1e9, "pageable": nbytes / t_page / 1e9},
            "lower_bound_ms": {"pinned": lo_pin * 1e3, "pageable": lo_page * 1e3},
            "overlap_fraction": {"pinned": ov_pin, "pageable": ov_page},
            "bits_equal_resident_job": same,
        }
    del panel_u8, store
    out = None
    if rig.rank == 0:
        f = forms["mean_window"]
        out = {
            "metric": "imputed SNPs/s of ONE blocking gauss_impute_window call on host genotype bytes (PCIe inclusive; never the headline)",
            "value": f["imputed_snps_per_s_pageable"], "unit": "imputed SNPs/s", "n_gpus": 1, "steps": reps, "warmup": 1,
            "ms_per_step": f["pageable_call_ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "distmix() one window per call, genotype bytes (one byte per genotype) in host memory, as the Rcpp "
                                   "driver holds them after ReadGenotype (gauss.cpp:720-785): the largest and a mean-cost window of the "
                                   f"chr22 study ({len(wins)} windows, N = {N})", "windows": 1, "samples": N},
            "roofline": {"kernel": "host -> HBM copy of the window's genotype rows (the call's floor)", "bound": "pcie",
                         "achieved": f["effective_GBs_of_call"]["pinned"], "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                         "frac": f["effective_GBs_of_call"]["pinned"] / PCIE_PEAK_GBS, "traffic": None,
                         "note": "achieved = genotype bytes / whole pinned call (job build, copy, kernels, results back); the bare copy "
                                 "reaches h2d_GBs_of_copy_only"},
            "forms": forms,
        }
    return out, None
