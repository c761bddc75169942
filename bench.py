#!/usr/bin/env python3
"""bench.py -- imputed SNPs/sec of the DISTMIX hot path on synthetic chr22-scale input.

One "step" = one full pass of the hot path (pack/stats -> fp32-MFMA LD Gram -> fp64 LD epilogue
-> Cholesky -> solve) over every 1 Mb window of one synthetic chromosome that is already
resident in HBM (BASELINE.json configs[3]: distmix, PGC2 weights, ~100k SNPs x 32 147 samples,
500 kb wings).  With N GPUs every rank owns one such chromosome (independent windows, no
data-path collective): weak scaling, value = all ranks' imputed SNPs / max-over-ranks time.

Prints ONE JSON line (rank 0).  Extra keys: "roofline" for the LD GEMM kernel (HIP events
recorded by the library on its own stream around every launch) and "cpu_baseline" (the CPU
oracle timed on a bounded sample and scaled to the workload's pair count).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def make_chromosome(args, seed):
    """Positions, measured mask, thresholds: host-side description of one synthetic chromosome."""
    from scipy.stats import norm
    from gauss_amd import synth
    rng = np.random.default_rng(seed)
    if getattr(args, "mode", "distmix") == "dist":
        pops = [p for p in synth.POPS_33KG if p[2] == "EUR"]                   # dist(study_pop="EUR"): N = 20 281
    else:
        pops = [p for p in synth.POPS_33KG if p[0] in synth.PGC2_WEIGHTS]      # 21 populations, N = 32 147
    if args.sample_scale != 1.0:
        pops = [(a, max(30, int(n * args.sample_scale)), s) for a, n, s in pops]
    w = np.array([synth.PGC2_WEIGHTS.get(p[0], 1.0) for p in pops])
    off = synth.pop_offsets([p[1] for p in pops])
    lo, hi = 16_050_000, 51_210_000                                        # chr22 span of the PGC2 file
    bp = np.sort(rng.choice(np.arange(lo, hi), size=args.snps, replace=False))
    measured = np.zeros(args.snps, dtype=bool)
    measured[rng.choice(args.snps, size=int(round(args.snps * 0.13362)), replace=False)] = True
    # Balding-Nichols frequencies (as gauss_amd/synth.py), kept inside (0.02, 0.98)
    p0 = rng.uniform(0.03, 0.5, args.snps)
    p0 = np.where(rng.random(args.snps) < 0.5, 1 - p0, p0)
    sups = sorted(set(p[2] for p in pops))

    def bn(p, f):
        return np.clip(rng.beta(p * (1 - f) / f, (1 - p) * (1 - f) / f), 0.02, 0.98)
    psup = {s: bn(p0, 0.15) for s in sups}
    ppop = np.stack([bn(psup[p[2]], 0.05) for p in pops], axis=1)
    thr = norm.ppf(ppop).astype(np.float32)
    rho = np.ones(args.snps, dtype=np.float32)
    rho[1:] = np.exp(-np.diff(bp) / 50e3)
    z = rng.standard_normal(args.snps) * 1.5
    return dict(pops=pops, w=w, off=off, bp=bp, measured=measured, thr=thr, rho=rho, z=z)


def windows_of(ch, args):
    """1 Mb prediction windows with 500 kb wings (dist.cpp:135-139 membership rules)."""
    bp, meas = ch["bp"], ch["measured"]
    out = []
    start = (int(bp[0]) // 1_000_000) * 1_000_000 + 1
    while start <= bp[-1]:
        end = start + 1_000_000 - 1
        ext = (bp >= start - args.wing) & (bp <= end + args.wing)
        pred = (bp >= start) & (bp <= end)
        mi = np.nonzero(ext & meas)[0]
        ui = np.nonzero(pred & ~meas)[0]
        if len(mi) > 10 and len(ui) > 10:                                  # dist.cpp:145-146
            out.append((mi, ui))
        start += 1_000_000
    if args.windows:
        out = out[: args.windows]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--snps", type=int, default=100_000)
    ap.add_argument("--windows", type=int, default=0, help="limit the number of windows (0 = all)")
    ap.add_argument("--wing", type=int, default=500_000)
    ap.add_argument("--sample-scale", type=float, default=1.0, help="shrink every population (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gram-dtype", choices=["f32", "i8"], default="f32",
                    help="LD Gram arithmetic of the headline run (f32 MFMA = north star; i8 MFMA = exact fast variant)")
    ap.add_argument("--no-i8-variant", action="store_true", help="skip the extra timing of the exact int8 variant")
    ap.add_argument("--mode", choices=["distmix", "dist"], default="distmix",
                    help="distmix (BASELINE configs[3], the headline) or dist on the EUR super-population (configs[2])")
    ap.add_argument("--panel-format", choices=["u8", "2bit"], default="2bit",
                    help="how the chromosome sits in HBM: per-window byte matrices, or one 2-bit packed row store "
                         "(the packed panel's resident form) that windows index by row")
    ap.add_argument("--streams", type=int, default=1, help="split the windows over this many jobs/streams")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP hot path has no CPU fallback")
    # rehearsal hook: several ranks on ONE card (gloo instead of RCCL, which refuses duplicate devices)
    rehearsal = os.environ.get("GAUSS_BENCH_SHARED_DEVICE") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # RCCL ("nccl" on ROCm); used for the timing barriers and two scalar reductions only
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)
    red_dev = "cpu" if rehearsal else "cuda"

    from gauss_amd import _lib, hotpath
    ctx = hotpath.Context(local)
    lib = ctx.lib

    # ---- synthetic chromosome, generated straight into HBM --------------------------------
    ch = make_chromosome(args, seed=20260213 + 3 + rank)
    N = int(ch["off"][-1])
    ld = (N + 63) // 64 * 64
    panel = torch.empty((args.snps, ld), dtype=torch.uint8, device="cuda")
    thr = np.ascontiguousarray(ch["thr"])
    import ctypes as C
    _lib.check(lib.gauss_synth_device(ctx.handle, panel.data_ptr(), args.snps, ld,
                                      ch["off"].ctypes.data_as(C.POINTER(C.c_int32)), len(ch["pops"]),
                                      thr.ctypes.data_as(C.POINTER(C.c_float)),
                                      ch["rho"].ctypes.data_as(C.POINTER(C.c_float)),
                                      C.c_uint64(20260213 + rank)))
    wins = windows_of(ch, args)
    win_mode = hotpath.MODE_POOLED if args.mode == "dist" else hotpath.MODE_WEIGHTED
    keep, descs = [], []
    store = None
    if args.panel_format == "2bit":
        # the whole chromosome as ONE resident 2-bit row store (288 GB of HBM hold an entire panel); windows
        # name their rows, the pack kernel gathers and unpacks them
        sizes = np.diff(ch["off"])
        ld2 = int(sum((int(m) + 63) // 64 * 16 for m in sizes))
        store = torch.empty((args.snps, ld2), dtype=torch.uint8, device="cuda")
        _lib.check(lib.gauss_pack2bit_device(ctx.handle, panel.data_ptr(), ld, store.data_ptr(), ld2, args.snps,
                                             ch["off"].ctypes.data_as(C.POINTER(C.c_int32)), len(ch["pops"])))
    for k, (mi, ui) in enumerate(wins):
        if store is None or k == 0:
            gm = panel.index_select(0, torch.from_numpy(mi).cuda())
            gu = panel.index_select(0, torch.from_numpy(ui).cuda())
            keep.append((gm, gu))                       # window 0 also feeds the CPU baseline sample
        if store is None:
            descs.append(dict(mode=win_mode, pop_off=ch["off"], pop_wgt=ch["w"], z1=ch["z"][mi],
                              dev=(gm.data_ptr(), gu.data_ptr(), len(mi), len(ui), ld)))
        else:
            descs.append(dict(mode=win_mode, pop_off=ch["off"], pop_wgt=ch["w"], z1=ch["z"][mi],
                              dev=(store.data_ptr(), store.data_ptr(), len(mi), len(ui), ld2),
                              packed=dict(fmt=1, rows_m=mi.astype(np.int32), rows_u=ui.astype(np.int32))))
    if store is not None:
        del panel
    torch.cuda.synchronize()
    ctx.set_gram_dtype(args.gram_dtype)
    if args.streams > 1:
        jobs = [hotpath.Job(descs[i::args.streams], ctx=(ctx if i == 0 else hotpath.Context(local)), on_device=True)
                for i in range(args.streams)]
    else:
        jobs = [hotpath.Job(descs, ctx=ctx, on_device=True)]
    job = jobs[0]
    work = {k: sum(j.work()[k] for j in jobs) for k in job.work()}
    stats = {k: sum(j.stats()[k] for j in jobs) for k in job.stats()}

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        for j in jobs:
            j.run()
        out = []
        for j in jobs:
            out += j.fetch()
        return out

    for _ in range(args.warmup):
        step()
    job.profile(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    gram_ms, gram_n = job.profile_get(0)
    stage_ms = {k: job.profile_get(i)[0] / max(1, args.steps) for i, k in
                enumerate(["gram", "pack_stats", "ld_epilogue", "factor", "solve"])}
    job.profile(False)

    # the same job with the LD Gram on the int8 matrix cores (identical integers, identical outputs):
    # reported next to the headline, never as `value`
    i8_variant = None
    if args.gram_dtype == "f32" and not args.no_i8_variant and args.streams == 1 and world == 1:
        ctx.set_gram_dtype("i8")
        j8 = hotpath.Job(descs, ctx=ctx, on_device=True)
        ctx.set_gram_dtype("f32")
        for _ in range(max(1, args.warmup)):
            j8.run(); r8 = j8.fetch()
        j8.profile(True)
        barrier()
        t8 = time.perf_counter()
        for _ in range(args.steps):
            j8.run(); r8 = j8.fetch()
        torch.cuda.synchronize()
        dt8 = time.perf_counter() - t8
        g8, n8 = j8.profile_get(0)
        same = all(np.array_equal(a["z"], b["z"]) and np.array_equal(a["info"], b["info"]) for a, b in zip(res, r8))
        i8_variant = {"ms_per_step": dt8 / args.steps * 1e3, "imputed_snps_per_s_this_rank": work["imputed_snps"] / (dt8 / args.steps),
                      "gram_ms": g8 / max(1, n8), "gram_tops_algorithmic": work["ld_flops"] / (g8 / max(1, n8) * 1e-3) / 1e12,
                      "kernel": "gram_kernel<i32x16> (v_mfma_i32_32x32x32_i8)", "bit_identical_to_f32_path": bool(same)}
        j8.close()

    bad = sum(int(r["status"] != 0) for r in res)
    finite = all(np.all(np.isfinite(r["z"])) and np.all(np.isfinite(r["info"])) for r in res)

    tmax, snps = dt, float(work["imputed_snps"])
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        s = torch.tensor([snps], dtype=torch.float64, device=red_dev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        tmax, snps = float(t.item()), float(s.item())

    out = None
    if rank == 0:
        ms_per_step = tmax / args.steps * 1e3
        avg_gram_s = gram_ms / max(1, gram_n) * 1e-3
        achieved = work["ld_flops"] / avg_gram_s / 1e12 if avg_gram_s > 0 else 0.0
        out = {
            "metric": "imputed SNPs/sec (whole node) at 1/2/4/8 GPUs; LD-GEMM MFMA TFLOP/s vs peak",
            "value": snps / (tmax / args.steps),
            "unit": "imputed SNPs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.gram_dtype == "f32" else "i8",
            "dtype_detail": ("LD GEMM on the %s matrix cores with exact integer partial sums; correlation tails, Cholesky and "
                             "solve in f64" % ("fp32" if args.gram_dtype == "f32" else "int8")),
            "data": "synthetic",
            "config": {
                "workload": ("distmix() synthetic chr22-scale (BASELINE.json configs[3]): " if args.mode == "distmix"
                             else "dist(study_pop=EUR) synthetic chr22-scale (BASELINE.json configs[2]): ") +
                            f"{args.snps} SNPs x {N} samples ({len(ch['pops'])} populations), {len(wins)} windows of 1 Mb, "
                            f"{args.wing // 1000} kb wings, one chromosome per GPU",
                "windows_per_gpu": len(wins), "snps": args.snps, "samples": N,
                "imputed_snps_per_gpu": int(work["imputed_snps"]),
                "mean_measured": float(np.mean([len(m) for m, _ in wins])),
                "mean_unmeasured": float(np.mean([len(u) for _, u in wins])),
                "windows_flagged": bad, "all_finite": bool(finite),
                "panel_in_hbm": ("one 2-bit packed row store, windows index it by row" if args.panel_format == "2bit"
                                 else "per-window one-byte genotype matrices"),
            },
            "roofline": {
                "kernel": "gram_kernel (LD GEMM, v_mfma_f32_32x32x2_f32)",
                "bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                "traffic": pmc_traffic() if (len(wins) == 36 and args.snps == 100_000) else None,
                "algorithmic_flops_per_launch": work["ld_flops"],
                "avg_launch_ms": gram_ms / max(1, gram_n), "launches": int(gram_n),
                "issued_flops_per_launch": stats["executed_flops"],
                "issued_tflops": stats["executed_flops"] / avg_gram_s / 1e12 if avg_gram_s > 0 else 0.0,
                "work_items": stats["items"], "partial_slab_bytes": stats["slab_bytes"],
            },
            "stage_ms_per_step": stage_ms,
        }
        if i8_variant is not None:
            out["int8_exact_variant"] = i8_variant
        if not args.no_cpu_baseline and world == 1:        # the CPU baseline is timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(ch, wins, keep, work, 0 if args.mode == "dist" else 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    job.close()
    return out


def pmc_traffic(kernel="gauss::gram_kernel<float>"):
    """HBM-side bytes per launch of the Gram kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc_traffic.csv: separate FETCH_SIZE / WRITE_SIZE runs of this same command,
    gfx950 correction applied).  PMC counters cannot be collected from inside the timed run."""
    import csv
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.csv"))):
        for r in csv.DictReader(open(f)):
            if r["kernel"] == kernel or r["kernel"] == kernel.split("<")[0]:
                best = float(r["hbm_bytes_per_launch_corrected"])
    return best


def cpu_baseline(ch, wins, keep, work, mode=1):
    """The loop-literal CPU oracle (1 thread, like the reference) on a bounded sample, scaled to
    the workload by its pair count: the reference's cost is N inner iterations per SNP pair
    (util.cpp:103-124), M(M+1)/2 + U + U*M pairs per window (distmix.cpp:180-217)."""
    import oracle
    ms_, us_ = 600, 600
    gm, gu = keep[0]
    mi, ui = wins[0]
    N = int(ch["off"][-1])
    gm_h = np.ascontiguousarray(gm[:ms_, :N].cpu().numpy())
    gu_h = np.ascontiguousarray(gu[:us_, :N].cpu().numpy())
    z1 = ch["z"][mi][:ms_]
    t0 = time.perf_counter()
    oracle.run_impute(mode, gm_h, gu_h, ch["off"], ch["w"], z1)
    t = time.perf_counter() - t0
    m, u = gm_h.shape[0], gu_h.shape[0]
    pairs_sample = m * (m + 1) / 2 + u + u * m
    pairs_total = sum(len(a) * (len(a) + 1) / 2 + len(b) + len(a) * len(b) for a, b in wins)
    est = t * pairs_total / pairs_sample
    # what an R user could do with one process per window: the same sample on several cores at once
    # (threads here: the oracle is plain C behind ctypes, which releases the GIL)
    from concurrent.futures import ThreadPoolExecutor
    par = max(1, min(16, (os.cpu_count() or 1)))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=par) as pool:
        list(pool.map(lambda _: oracle.run_impute(mode, gm_h, gu_h, ch["off"], ch["w"], z1), range(par)))
    tp = time.perf_counter() - t0
    est_par = tp / par * pairs_total / pairs_sample
    return {
        "value": work["imputed_snps"] / est, "unit": "imputed SNPs/s", "cores": 1, "kind": "port",
        "host_cores": os.cpu_count(),
        "windows_in_parallel": {"value": work["imputed_snps"] / est_par, "unit": "imputed SNPs/s", "cores": par,
                                "sample": f"{par} concurrent copies of the same sample in {tp:.2f} s"},
        "sample": f"oracle run_{'distmix' if mode else 'dist'} on a sub-window of window 0 (M={m}, U={u}, N={N}): {t:.2f} s for "
                  f"{pairs_sample:.0f} SNP pairs; scaled by the workload's {pairs_total:.3g} pairs "
                  f"(estimated {est:.0f} s per chromosome, dense tail of the full-size windows not included)",
    }


if __name__ == "__main__":
    main()
