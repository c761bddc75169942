#!/usr/bin/env python3
"""bench.py -- imputed SNPs/sec of the DISTMIX hot path on the chr22 study (BASELINE.json configs[3]).

One "step" = one full pass of the hot path (pack/stats -> fp32-MFMA LD Gram -> fp64 LD epilogue ->
Cholesky + inverse factor -> product with the z / info sums) over every 1 Mb window of ONE chromosome whose panel is already resident in HBM.
Measured SNPs (positions, z) are the reference's own chr22 study file; unmeasured SNPs and genotypes
are synthetic (gauss_amd/workload.py).

Multi-GPU (`--gpus N`, one process per GPU, no data-path collective):
  --scaling strong (default)  the windows of the ONE chromosome are sharded over the ranks: whole windows by
                              LPT on their cost, then levelled by cutting a few windows' unmeasured SNPs
                              between two ranks (farm.level_windows; --shard lpt = whole windows only);
                              every rank keeps only the panel rows its windows touch; value = the chromosome's imputed SNPs / max-over-ranks
                              time.  This is configs[3]: "windows sharded across 8xMI355X".
  --scaling weak              every rank imputes its own whole chromosome (round-1 behaviour).
With N > 1 the strong line carries a "weak_scaling" block measured in the same launch.
`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (child process,
before anything touches the GPU).

Other modes (never the headline): --mode dist (configs[2]), --mode computeLD (configs[1]),
--mode jepegmix (configs[4]), --mode e2e (packed panel file + GWAS file on disk -> result table).

Prints ONE JSON line (rank 0).  "roofline" describes the LD GEMM kernel from HIP events recorded by
the library on its own stream around every launch; "cpu_baseline" is the CPU oracle timed on a
bounded sample (N = 1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
FP64_MFMA_PEAK_TFLOPS = 78.6    # v_mfma_f64_16x16x4_f64: 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz
I8_MFMA_PEAK_TOPS = 5033.6      # v_mfma_i32_32x32x32_i8: 2 x the BF16 rate per clock (MI355X_MICROARCH.md, matrix cores) = 32 x the fp32 matrix peak
HBM_PEAK_GBS = 8000.0
METRIC = "imputed SNPs/sec (whole node) at 1/2/4/8 GPUs; LD-GEMM MFMA TFLOP/s vs peak"
STAGES = ["gram", "pack_stats", "ld_epilogue", "factor", "solve"]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--snps", type=int, default=100_000)
    ap.add_argument("--windows", type=int, default=0, help="limit the number of windows (0 = all)")
    ap.add_argument("--wing", type=int, default=500_000)
    ap.add_argument("--sample-scale", type=float, default=1.0, help="shrink every population (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gram-dtype", choices=["f32", "i8"], default="f32",
                    help="LD Gram arithmetic of the headline run (f32 MFMA = north star; i8 MFMA = exact fast variant)")
    ap.add_argument("--no-i8-variant", action="store_true", help="skip the extra timing of the exact int8 variant")
    ap.add_argument("--mode", choices=["distmix", "dist", "computeLD", "jepegmix", "e2e", "window"], default="distmix",
                    help="distmix = BASELINE configs[3], the headline; the others are the remaining configs / the file-to-table run")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--shard", choices=["auto", "leveled", "contiguous", "lpt"], default="auto",
                    help="strong scaling: lpt = whole windows, longest first (farm.assign_windows); leveled = LPT, then the most "
                         "loaded ranks hand slices of a window's unmeasured SNPs to the least loaded ones (farm.level_windows); "
                         "contiguous = equal-cost stretches of the chromosome, boundary windows cut (farm.balance_windows)")
    ap.add_argument("--no-weak-line", action="store_true", help="N > 1, strong: skip the extra weak-scaling pass")
    ap.add_argument("--verify-shards", action="store_true",
                    help="N > 1, strong: rank 0 also runs every window itself and checks the ranks' z / info bit for bit")
    ap.add_argument("--no-from-text", action="store_true", help="end_to_end: skip the leg that starts from a BGZF TEXT panel")
    ap.add_argument("--text-snps", type=int, default=100_000, help="end_to_end.from_text: SNPs of the text panel (default: the whole chromosome)")
    ap.add_argument("--no-tails-alone", action="store_true", help="skip the one-stream pass that times the fp64 tails and the HBM-bound kernels stand-alone (probes)")
    ap.add_argument("--no-e2e", action="store_true", help="N = 1: skip the end_to_end block (files on disk -> result table)")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1: skip the other_configs block (configs[1], [2], [4] at <= 5 timed steps each)")
    ap.add_argument("--emulate-world", type=int, default=-1,
                    help="single GPU: time every rank's share of an N-rank strong-scaling run one after the other "
                         "(what one rank of N would do per step); printed as `emulated_strong_scaling`, never as `value`.  "
                         "Default (-1): 8 on the whole-chromosome N = 1 run, with at most --emulate-steps steps per share; 0 = off")
    ap.add_argument("--emulate-rank", type=int, default=-1, help="probe: emulate only this rank's share (no efficiency is derived)")
    ap.add_argument("--emulate-steps", type=int, default=10, help="timed steps per emulated share (each share also gets 3 warm-up steps)")
    ap.add_argument("--no-parity-spot", action="store_true", help="skip the oracle spot check of the timed job's own results")
    ap.add_argument("--streams", type=int, default=1, help="split a rank's windows over this many contexts (one stream pair each)")
    return ap.parse_args(argv)


def launch_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a child (torch.distributed.run),
    relay its output and exit code.  Nothing in this process has touched the GPU."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


class Rig:
    """Process-level plumbing: rank, device, torch.distributed (timing barriers + scalar reductions only)."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP hot path has no CPU fallback")
        from gauss_amd import hotpath
        # rehearsal hook: several ranks on ONE card (gloo instead of RCCL, which refuses duplicate devices)
        self.rehearsal = os.environ.get("GAUSS_BENCH_SHARED_DEVICE") == "1"
        self.local = hotpath.rank_device()
        torch.cuda.set_device(self.local)
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            # RCCL ("nccl" on ROCm); used for the timing barriers and a few scalar reductions only
            dist.init_process_group("gloo" if self.rehearsal else "nccl", rank=self.rank, world_size=self.world)
            self.dist = dist
            # python objects (window lists, digests, result tables) travel over a gloo group: host data, host transport
            self.obj_group = None if self.rehearsal else dist.new_group(backend="gloo")
        self.red_dev = "cpu" if (self.rehearsal or self.world == 1) else "cuda"
        self.ctx = hotpath.Context(self.local)

    def _dist_barrier(self):
        # RCCL: name the device (otherwise every rank prints two warnings about guessing it -- 4 KB of stderr at 8 ranks,
        # and the driver's record is the tail of stdout + stderr); gloo takes no device_ids
        if self.rehearsal:
            self.dist.barrier()
        else:
            self.dist.barrier(device_ids=[self.local])

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self._dist_barrier()
        self.torch.cuda.synchronize()

    def reduce(self, x, op):
        if self.dist is None:
            return float(x)
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.red_dev)
        self.dist.all_reduce(t, op={"max": self.dist.ReduceOp.MAX, "sum": self.dist.ReduceOp.SUM}[op])
        return float(t.item())

    def gather(self, obj):
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj, group=self.obj_group)
        return out

    def close(self):
        if self.dist is not None:
            self._dist_barrier()
            self.dist.destroy_process_group()


def synth_panel(rig, ch, seed):
    """The chromosome's genotypes, generated in HBM: (u8 panel tensor, row stride)."""
    import ctypes as C
    from gauss_amd import _lib
    torch, ctx = rig.torch, rig.ctx
    N = int(ch["off"][-1])
    ld = (N + 63) // 64 * 64
    S = len(ch["bp"])
    panel = torch.empty((S, ld), dtype=torch.uint8, device="cuda")
    thr = np.ascontiguousarray(ch["thr"])
    _lib.check(ctx.lib.gauss_synth_device(ctx.handle, panel.data_ptr(), S, ld,
                                          ch["off"].ctypes.data_as(C.POINTER(C.c_int32)), len(ch["pops"]),
                                          thr.ctypes.data_as(C.POINTER(C.c_float)),
                                          ch["rho"].ctypes.data_as(C.POINTER(C.c_float)), C.c_uint64(seed)))
    return panel, ld


def pack_store(rig, ch, panel, ld):
    """The whole chromosome as ONE 2-bit row store (the packed panel's resident form)."""
    import ctypes as C
    from gauss_amd import _lib
    torch, ctx = rig.torch, rig.ctx
    sizes = np.diff(ch["off"])
    ld2 = int(sum((int(m) + 63) // 64 * 16 for m in sizes))
    store = torch.empty((panel.shape[0], ld2), dtype=torch.uint8, device="cuda")
    _lib.check(ctx.lib.gauss_pack2bit_device(ctx.handle, panel.data_ptr(), ld, store.data_ptr(), ld2, panel.shape[0],
                                             ch["off"].ctypes.data_as(C.POINTER(C.c_int32)), len(ch["pops"])))
    return store, ld2


def window_descs(ch, wins, store, ld2, mode, rows_of=None):
    """Window descriptors naming their rows in a resident 2-bit store (rows_of maps chromosome rows to store rows)."""
    from gauss_amd import hotpath
    win_mode = hotpath.MODE_POOLED if mode == "dist" else hotpath.MODE_WEIGHTED
    descs = []
    for _, mi, ui in wins:
        rm = mi if rows_of is None else np.searchsorted(rows_of, mi)
        ru = ui if rows_of is None else np.searchsorted(rows_of, ui)
        descs.append(dict(mode=win_mode, pop_off=ch["off"], pop_wgt=ch["w"], z1=ch["z"][mi],
                          dev=(store.data_ptr(), store.data_ptr(), len(mi), len(ui), ld2),
                          packed=dict(fmt=1, rows_m=rm.astype(np.int32), rows_u=ru.astype(np.int32))))
    return descs


class Runner:
    """A rank's windows as one job per context (stream).  step() queues the next run of every job and THEN collects the
    previous one (the library keeps two result mirrors per job): the host's share of a step -- waking up, copying the
    results out, building the result arrays, queuing the next run -- overlaps GPU work instead of leaving the GPU
    idle for ~70 us per step.  drain() collects the last run.  Every run is fetched exactly once."""

    def __init__(self, rig, descs, streams=1):
        from gauss_amd import hotpath
        self.rig = rig
        self.ctxs = [rig.ctx] + [hotpath.Context(rig.local) for _ in range(max(1, streams) - 1)]
        self.jobs, self.order = [], []
        if descs:
            n = min(len(self.ctxs), len(descs))
            self.jobs = [hotpath.Job(descs[i::n], ctx=self.ctxs[i], on_device=True) for i in range(n)]
            self.order = [k for i in range(n) for k in range(i, len(descs), n)]
        self.inflight = False
        self.done_at = []               # host time at which each fetched run was complete (timed(): the longest step of the region)
        self.work = {k: sum(j.work()[k] for j in self.jobs) for k in ("ld_flops", "solve_flops", "bytes", "imputed_snps")}
        self.stats = {k: sum(j.stats()[k] for j in self.jobs) for k in ("items", "executed_flops", "slab_bytes", "workspace_bytes")}

    def step(self):
        """Queue one more run; returns the results of the PREVIOUS one (None on the first call after a drain)."""
        for j in self.jobs:
            j.run()
        out = None
        if self.inflight:
            out = []
            for j in self.jobs:
                out += j.fetch()
            self.done_at.append(time.perf_counter())
        self.inflight = True
        return out

    def drain(self):
        out = []
        if self.inflight:
            for j in self.jobs:
                out += j.fetch()
            self.done_at.append(time.perf_counter())
            self.inflight = False
        return out

    def results_in_order(self, res):
        out = [None] * len(res)
        for k, r in zip(self.order, res):
            out[k] = r
        return out

    def profile(self, on):
        for j in self.jobs:
            j.profile(on)

    def stage_ms(self):
        """(ms, launches) per stage summed over this rank's jobs since profiling was enabled."""
        tot = {}
        for i, k in enumerate(STAGES):
            ms = n = 0
            for j in self.jobs:
                a, b = j.profile_get(i)
                ms += a
                n += b
            tot[k] = (ms, n)
        return tot

    def timed(self, steps, warmup, events_in_region=True):
        """events_in_region: the library's per-stage HIP events are recorded during the timed steps (N = 1: the roofline
        is measured live over the timed region).  With several ranks the events would only slow the rank that
        records them (~0.04 ms per step, and the slowest rank is the result), so the stages are timed on a few
        extra steps after the region instead."""
        for _ in range(warmup):
            self.step()
        self.drain()
        self.profile(events_in_region)
        self.rig.barrier()
        c0 = [c.counters() for c in self.ctxs]
        t0 = time.perf_counter()
        self.done_at = []
        for _ in range(steps):
            self.step()
        res = self.drain()
        self.rig.barrier()
        dt = time.perf_counter() - t0
        # how the library queued the runs of the TIMED region only (the contexts' counters are cumulative: warm-up and earlier legs excluded)
        c1 = [c.counters() for c in self.ctxs]
        self.launch_form = {k: sum(b[k] - a[k] for a, b in zip(c0, c1)) for k in c1[0]}
        # completion to completion of consecutive runs (the first one from the start of the region): the longest step
        marks = [t0] + self.done_at
        self.longest_step_ms = max((b - a) for a, b in zip(marks[:-1], marks[1:])) * 1e3 if len(marks) > 1 else 0.0
        if events_in_region:
            st = self.stage_ms()
        else:
            extra = 3
            self.profile(True)
            for _ in range(extra):
                self.step()
            self.drain()
            st = {k: (ms * steps / extra, n * steps // extra) for k, (ms, n) in self.stage_ms().items()}
        self.profile(False)
        return dt, st, res

    def close(self):
        for j in self.jobs:
            j.close()
        for c in self.ctxs[1:]:
            c.close()
        self.jobs = []


def result_digest(res):
    """Order-independent check values of a list of window results (exact: sums of the raw bit patterns)."""
    z = np.concatenate([r["z"] for r in res]) if res else np.zeros(0)
    info = np.concatenate([r["info"] for r in res]) if res else np.zeros(0)
    return [int(z.view(np.uint64).astype(object).sum() % (1 << 61)), int(info.view(np.uint64).astype(object).sum() % (1 << 61)),
            int(sum(int(r["status"] != 0) for r in res)), bool(np.all(np.isfinite(z)) and np.all(np.isfinite(info)))]


SPOT_TOL = 1e-8


def spot_inputs(torch, panel, wins, n_unmeasured=16, mid_m=330):
    """Host copies of the genotype rows the parity spot check needs: ALL measured rows and `n_unmeasured` evenly spaced
    unmeasured rows of the smallest window and of a mid-size one (the window whose measured count is nearest `mid_m`:
    the oracle's pair loops over it stay under ~2 s at N = 32 147)."""
    ms = [len(w[1]) for w in wins]
    k_small = int(np.argmin(ms))
    k_mid = int(np.argmin([abs(m - mid_m) + (10 ** 9 if k == k_small else 0) for k, m in enumerate(ms)]))
    out = []
    for k in sorted({k_small, k_mid}):
        _, mi, ui = wins[k]
        sel = np.unique(np.linspace(0, len(ui) - 1, num=min(n_unmeasured, len(ui))).round().astype(np.int64))
        out.append((k, sel, panel.index_select(0, torch.from_numpy(mi).cuda()).cpu().numpy(),
                    panel.index_select(0, torch.from_numpy(ui[sel]).cuda()).cpu().numpy()))
    return out


def parity_spot(ch, wins, spot, res, mode):
    """The numbers the bench has just timed, against the CPU oracle (outside the timed region): z / info of the selected
    unmeasured SNPs of the spot windows as the HEADLINE job returned them (resident 2-bit store, shared measured rows, two
    Gram launches, chain beside the second, wide product tiles) vs oracle.run_impute on the same genotypes at full N
    (distmix.cpp:165-228 / dist.cpp:156-202; an unmeasured SNP's z / info depend on the window's measured set and on its
    own row only, so a subset of them is imputed exactly as inside the whole window)."""
    import oracle
    N = int(ch["off"][-1])
    t0 = time.perf_counter()
    worst_z = worst_i = 0.0
    snps = 0
    detail = []
    for k, sel, gm, gu in spot:
        _, mi, ui = wins[k]
        want = oracle.run_impute(mode, np.ascontiguousarray(gm[:, :N]), np.ascontiguousarray(gu[:, :N]), ch["off"], ch["w"], ch["z"][mi])
        got_z, got_i = res[k]["z"][sel], res[k]["info"][sel]
        dz = float(np.max(np.abs(got_z - want["z"]) / np.maximum(1.0, np.abs(want["z"]))))
        di = float(np.max(np.abs(got_i - want["info"]) / np.abs(want["info"])))
        worst_z, worst_i = max(worst_z, dz), max(worst_i, di)
        snps += len(sel)
        detail.append({"window": int(k), "measured": int(len(mi)), "unmeasured_checked": int(len(sel)), "max_rel_z": dz, "max_rel_info": di})
    return {"windows": len(spot), "snps": snps, "max_rel_z": worst_z, "max_rel_info": worst_i, "tolerance": SPOT_TOL,
            "ok": bool(worst_z <= SPOT_TOL and worst_i <= SPOT_TOL), "oracle_seconds": round(time.perf_counter() - t0, 2),
            "per_window": detail,
            "what": "z / info of the timed job's own results vs the CPU oracle on the same rows at full N (all measured SNPs of the "
                    "window, a spread of its unmeasured ones); |dz| / max(1, |z|) and |dinfo| / info"}


def run_impute(args, rig, quiet=False, light=False):
    """--mode distmix / dist: the headline.  quiet: return the line instead of printing it; light: a smaller CPU-baseline sample
    without the all-cores pass (the `other_configs` leg of the default line, which has seconds, not half a minute)."""
    from gauss_amd import workload
    torch = rig.torch
    strong = args.scaling == "strong"
    seed = 20260216 + (0 if strong else rig.rank)
    ch = workload.make_chromosome(args.snps, args.mode, seed=seed, sample_scale=args.sample_scale)
    N = int(ch["off"][-1])
    wins = workload.windows_of(ch, args.wing, args.windows)
    shares, load = shares_of(args, wins, N, rig.world if strong else 1)
    mine = shares[rank_of(rig) if strong else 0]              # [(window, u0, u1)]: a cut window appears on two ranks
    my_wins = workload.pieces_of(wins, mine)

    panel, ld = synth_panel(rig, ch, seed)
    store, ld2 = pack_store(rig, ch, panel, ld)
    keep0 = None
    if rig.rank == 0 and not args.no_cpu_baseline and rig.world == 1:
        n0 = 250 if light else 600
        k0 = next((k for k, w in enumerate(wins) if len(w[1]) >= n0 and len(w[2]) >= n0), 0)
        _, mi, ui = wins[k0]                                  # this window also feeds the CPU baseline sample
        keep0 = (k0, panel.index_select(0, torch.from_numpy(mi[:n0]).cuda()).cpu().numpy(),
                 panel.index_select(0, torch.from_numpy(ui[:n0]).cuda()).cpu().numpy())
    spot = None
    if rig.rank == 0 and rig.world == 1 and not args.no_parity_spot and wins:
        spot = spot_inputs(torch, panel, wins)
    del panel
    full_store = store
    rows_of = None
    if strong and rig.world > 1:
        # this rank keeps only the panel rows its windows touch (the slice a farm rank uploads)
        rows_of = np.unique(np.concatenate([np.concatenate([w[1], w[2]]) for w in my_wins])) if my_wins else np.zeros(0, np.int64)
        store = full_store.index_select(0, torch.from_numpy(rows_of).cuda()) if len(rows_of) else full_store[:1]
    torch.cuda.synchronize()
    rig.ctx.set_gram_dtype(args.gram_dtype)
    runner = Runner(rig, window_descs(ch, my_wins, store, ld2, args.mode, rows_of), args.streams)
    work, stats = runner.work, runner.stats

    dt, st, res = runner.timed(args.steps, args.warmup, events_in_region=rig.world == 1)
    # how the library queued the timed runs (gauss_hip_counters): merged = ONE Gram launch with the chain beside it; demoted = two
    # launches because the context shared its device with another one; giveups = merged runs repaired inside their fetch
    launch_form = runner.launch_form
    res = runner.results_in_order(res) if runner.jobs else []
    gram_ms, gram_n = st["gram"]
    tmax = rig.reduce(dt, "max")
    snps = rig.reduce(work["imputed_snps"], "sum")
    digests = rig.gather(dict(rank=rig.rank, windows=mine, digest=result_digest(res), dt=dt,
                              resident_rows=int(store.shape[0]), imputed=int(work["imputed_snps"])))

    shard_check = None
    if args.verify_shards and strong and rig.world > 1:
        # every rank hands its windows' z / info to rank 0, which imputes the whole chromosome itself
        parts = rig.gather({piece: (r["z"], r["info"]) for piece, r in zip(mine, res)})
        if rig.rank == 0:
            allr = Runner(rig, window_descs(ch, wins, full_store, ld2, args.mode), 1)
            allr.step()
            ref = allr.drain()
            allr.close()
            shard_check = pieces_equal_whole(parts, ref, wins)

    # the same job with the LD Gram on the int8 matrix cores (identical integers, identical outputs):
    # reported next to the headline, never as `value`
    i8_variant = None
    if args.gram_dtype == "f32" and not args.no_i8_variant and args.streams == 1 and rig.world == 1:
        rig.ctx.set_gram_dtype("i8")
        r8 = Runner(rig, window_descs(ch, wins, store, ld2, args.mode), 1)
        rig.ctx.set_gram_dtype("f32")
        dt8, st8, res8 = r8.timed(args.steps, max(1, args.warmup))
        same = all(np.array_equal(a["z"], b["z"]) and np.array_equal(a["info"], b["info"]) for a, b in zip(res, res8))
        g8, n8 = st8["gram"]
        i8_variant = {"ms_per_step": dt8 / args.steps * 1e3, "imputed_snps_per_s_this_rank": work["imputed_snps"] / (dt8 / args.steps),
                      "gram_ms": g8 / args.steps, "gram_launches_per_step": max(1, round(n8 / args.steps)),
                      "gram_tops_algorithmic": work["ld_flops"] / (g8 / args.steps * 1e-3) / 1e12,
                      "kernel": "gram_kernel<i32x16> (v_mfma_i32_32x32x32_i8)", "bit_identical_to_f32_path": bool(same)}
        r8.close()

    # the fp64 tails on their own: in the headline run B21's half of the LD epilogue co-runs with the factorisation
    # chain on the library's side stream, so their stage timers overlap; a short pass on a one-stream context gives
    # the stand-alone figures `roofline_solve` quotes (never `value`)
    tails_alone = None
    if rig.world == 1 and rig.rank == 0 and runner.jobs and not args.no_tails_alone:
        from gauss_amd import hotpath
        old_env = os.environ.get("GAUSS_SIDE_STREAM")
        os.environ["GAUSS_SIDE_STREAM"] = "0"
        main_ctx, rig.ctx = rig.ctx, hotpath.Context(rig.local)
        if old_env is None:
            del os.environ["GAUSS_SIDE_STREAM"]
        else:
            os.environ["GAUSS_SIDE_STREAM"] = old_env
        rig.ctx.set_gram_dtype(args.gram_dtype)
        r1 = Runner(rig, window_descs(ch, my_wins, store, ld2, args.mode, rows_of), 1)
        n1 = max(3, min(10, args.steps))
        dt1, st1, _ = r1.timed(n1, 2)
        tails_alone = {"steps": n1, "ms_per_step": dt1 / n1 * 1e3, "stage_ms_per_step": {k: v[0] / n1 for k, v in st1.items()}}
        r1.close()
        rig.ctx.close()
        rig.ctx = main_ctx

    weak = None
    if strong and rig.world > 1 and not args.no_weak_line:
        # second line: every rank imputes the WHOLE chromosome (per-GPU work fixed as N grows)
        runner.close()
        wr = Runner(rig, window_descs(ch, wins, full_store, ld2, args.mode), 1)
        wdt, _, _ = wr.timed(args.steps, max(1, args.warmup))
        wt = rig.reduce(wdt, "max")
        ws = rig.reduce(wr.work["imputed_snps"], "sum")
        weak = {"scaling": "weak", "value": ws / (wt / args.steps), "unit": "imputed SNPs/s", "ms_per_step": wt / args.steps * 1e3,
                "workload": "the whole chromosome on every rank"}
        wr.close()

    emu = None
    emu_world = args.emulate_world
    if emu_world < 0:           # default: the driver's N = 1 line records the 8-rank prediction (whole chromosome only)
        emu_world = 8 if (rig.world == 1 and strong and not args.windows and args.streams == 1 and len(wins) >= 16) else 0
    if rig.world == 1 and emu_world > 1:
        e_steps = max(1, min(args.steps, args.emulate_steps)) if args.emulate_world < 0 else args.steps
        e_warm = 3 if args.emulate_world < 0 else max(1, args.warmup)
        shares_e, load_e = shares_of(args, wins, N, emu_world)
        per_rank, parts_e = [], []
        for r in range(emu_world):
            if args.emulate_rank >= 0 and r != args.emulate_rank:
                continue
            wr = workload.pieces_of(wins, shares_e[r])
            rr = Runner(rig, window_descs(ch, wr, store, ld2, args.mode), 1)
            # as a rank of a real N > 1 run is timed: no per-stage events inside the timed steps (Runner.timed), stages from extra steps
            dtr, str_, res_r = rr.timed(e_steps, e_warm, events_in_region=False)
            parts_e.append({piece: (q["z"], q["info"]) for piece, q in zip(shares_e[r], rr.results_in_order(res_r))})
            per_rank.append({"rank": r, "windows": len(wr), "ms_per_step": dtr / e_steps * 1e3,
                             "executed_flops": rr.stats["executed_flops"], "work_items": rr.stats["items"], "algorithmic_flops": rr.work["ld_flops"],
                             "mu": [(int(len(w[1])), int(len(w[2]))) for w in wr],
                             "stage_ms": {k: v[0] / e_steps for k, v in str_.items()},
                             "gram_frac_of_peak": (rr.work["ld_flops"] * e_steps / (str_["gram"][0] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS)
                             if str_["gram"][0] > 0 else 0.0})
            rr.close()
        slow = max(q["ms_per_step"] for q in per_rank)
        emu = {"world": emu_world, "steps_per_share": e_steps, "per_rank": per_rank, "slowest_rank_ms": slow,
               "one_gpu_ms": dt / args.steps * 1e3, "predicted_speedup": dt / args.steps * 1e3 / slow,
               "predicted_efficiency": dt / args.steps * 1e3 / slow / emu_world,
               "load_imbalance": max(load_e) / (sum(load_e) / len(load_e)), "shard": shard_mode(args, emu_world),
               "cut_windows": sum(1 for sh in shares_e for _, u0, _ in sh if u0 > 0),
               "pieces_bit_identical_to_one_job": pieces_equal_whole(parts_e, res, wins) if args.emulate_rank < 0 else None,
               "note": "each rank's share timed alone on ONE GPU, one after the other, the way a rank of an N > 1 run is timed (no "
                       "per-stage events inside the timed steps; stage_ms from three extra steps): an emulation of the per-rank step "
                       "time, not a multi-GPU measurement (no 8-GPU node is available to the builder)"}

    e2e = None
    if rig.world == 1 and args.mode == "distmix" and not args.no_e2e and not args.windows and args.streams == 1:
        runner.close()
        del store, full_store
        torch.cuda.empty_cache()
        from gauss_amd import benchmodes
        e2e = benchmodes.e2e_block(args, rig, ch, steps=5)

    out = None
    if rig.rank == 0:
        nsteps = max(1, args.steps)
        # the Gram kernel may take two launches per step (B11's tile pairs, then B21's with the factorisation chain beside
        # it): per-launch figures are averages over the launches, flops per launch = flops per step / launches per step
        gram_lps = max(1, round(gram_n / nsteps))
        avg_gram_s = gram_ms / max(1, gram_n) * 1e-3
        achieved = work["ld_flops"] / gram_lps / avg_gram_s / 1e12 if avg_gram_s > 0 else 0.0
        per_step = {k: v[0] / nsteps for k, v in st.items()}
        alone = tails_alone["stage_ms_per_step"] if tails_alone else per_step
        tail_s = (alone["factor"] + alone["solve"]) * 1e-3
        solve_ach = work["solve_flops"] / tail_s / 1e12 if tail_s > 0 else 0.0
        bad = sum(d["digest"][2] for d in digests)
        finite = all(d["digest"][3] for d in digests)
        m_all = [len(mi) for _, mi, _ in wins]
        u_all = [len(ui) for _, _, ui in wins]
        out = {
            "metric": METRIC,
            "value": snps / (tmax / args.steps),
            "unit": "imputed SNPs/s",
            "n_gpus": rig.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": tmax / args.steps * 1e3,
            "ms_longest_step": runner.longest_step_ms,        # completion to completion of consecutive steps in the timed region (this rank)
            # exact check values of the timed job's z / info (sums of the raw bit patterns, summed over the ranks): two builds that
            # return the same bits for the same seeded panel print the same pair
            "launch_form": launch_form,
            "result_digest": [sum(d["digest"][0] for d in digests) % (1 << 61), sum(d["digest"][1] for d in digests) % (1 << 61)],
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32" if args.gram_dtype == "f32" else "i8",
            "dtype_detail": ("LD GEMM on the %s matrix cores with exact integer partial sums; correlation tails, Cholesky and "
                             "solve in f64" % ("fp32" if args.gram_dtype == "f32" else "int8")),
            "data": "synthetic",
            "config": {
                "workload": (("distmix() chr22 (BASELINE.json configs[3]): " if args.mode == "distmix"
                              else "dist(study_pop=EUR) chr22 (BASELINE.json configs[2]): ") +
                             f"measured SNPs (positions, z) from {ch['study']}, synthetic unmeasured SNPs and genotypes; "
                             f"{len(ch['bp'])} SNPs x {N} samples ({len(ch['pops'])} populations), {len(wins)} windows of 1 Mb, "
                             f"{args.wing // 1000} kb wings; " +
                             (("the windows of ONE chromosome sharded over the ranks " +
                               {"leveled": "(LPT on modelled cost, then levelled: a few windows are cut between two ranks, which both "
                                           "factor the window's B11)",
                                "contiguous": "(contiguous shares of equal cost; a window on a boundary is cut between two ranks)",
                                "lpt": "(whole windows, LPT on modelled cost)"}[shard_mode(args, rig.world)]) if strong
                              else "one whole chromosome per rank")),
                "windows": len(wins), "snps": int(len(ch["bp"])), "samples": N,
                "imputed_snps_per_step": int(snps),
                "measured_per_window": {"min": int(min(m_all)), "mean": float(np.mean(m_all)), "max": int(max(m_all))},
                "unmeasured_per_window": {"min": int(min(u_all)), "mean": float(np.mean(u_all)), "max": int(max(u_all))},
                "windows_per_rank": [len(d["windows"]) for d in digests],
                "resident_panel_rows_per_rank": [d["resident_rows"] for d in digests],
                "shard": shard_mode(args, rig.world) if strong else None,
                "cut_windows": sum(1 for d in digests for _, u0, _ in d["windows"] if u0 > 0),
                "load_imbalance": (max(load) / (sum(load) / len(load))) if strong else 1.0,
                "rank_seconds": [d["dt"] for d in digests],
                "windows_flagged": bad, "all_finite": bool(finite),
                "panel_in_hbm": "one 2-bit packed row store per rank, windows index it by row",
            },
            "roofline": {
                "kernel": "gram_kernel (LD GEMM, v_mfma_f32_32x32x2_f32)" + (" on rank 0" if rig.world > 1 else ""),
                "bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                "algorithmic_flops_per_launch": work["ld_flops"] / gram_lps, "algorithmic_flops_per_step": work["ld_flops"],
                "avg_launch_ms": gram_ms / max(1, gram_n), "launches": int(gram_n), "launches_per_step": gram_lps,
                "kernel_ms_per_step": gram_ms / nsteps,
                "issued_flops_per_launch": stats["executed_flops"] / gram_lps,
                "issued_tflops": stats["executed_flops"] / gram_lps / avg_gram_s / 1e12 if avg_gram_s > 0 else 0.0,
                "work_items": stats["items"], "partial_slab_bytes": stats["slab_bytes"],
                "measured_on": "HIP events of every launch in the timed region" if rig.world == 1 else
                               "HIP events of 3 extra steps on rank 0 after the timed region (no events inside it: they would slow rank 0 only)",
            },
            "roofline_solve": {
                "kernel": "factor_*_kernel (Cholesky + rows of L^-1) + impute_gemm_kernel (v_mfma_f64_16x16x4_f64)", "bound": "mfma",
                "achieved": solve_ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": solve_ach / FP64_MFMA_PEAK_TFLOPS,
                "algorithmic_flops_per_step": work["solve_flops"], "ms_per_step": tail_s * 1e3,
                "definition": "sum over windows of M^3/3 + 2 U M^2 + 4 U M (SURVEY.md 8d) / (factor + solve time)",
                "measured_on": ("a one-stream pass of the same job (GAUSS_SIDE_STREAM=0, %d steps, %.2f ms/step): in the headline run "
                                "B21's half of the LD epilogue co-runs with the factorisation chain and the stage timers overlap"
                                % (tails_alone["steps"], tails_alone["ms_per_step"])) if tails_alone else "the headline run's stage timers",
            },
            "stage_ms_per_step": per_step,
            "stage_note": "HIP-event time per stage on the queue it runs on.  With the chain beside the Gram kernel (default for jobs whose "
                          "B21 items can cover it): gram = the ONE launch per step (B11's items first, counted off for the chain queue); "
                          "ld_epilogue = B11's tiles (chain queue, under the Gram launch) + the LATE windows' B21 tiles on the main queue -- "
                          "the early windows' B21 tiles run on the low-priority queue in the Gram launch's last round and are not on this "
                          "timer; `factor` = the small-footprint chain on the chain queue UNDER the Gram launch (hidden: not part of the "
                          "step's critical path); otherwise ld_epilogue = B11's tiles and B21's run on the side queue beside `factor`",
        }
        # the HBM-bound kernels (K1 pack, K4 LD epilogue), timed stand-alone in the one-stream pass (in the
        # headline run B11's epilogue tiles run beside the Gram kernel): algorithmic bytes = the 2-bit source rows in
        # ((M + U) N / 4 per window) for pack, the fp64 LD entries out ((M^2 + U M) 8 per window) for the epilogue
        if tails_alone is not None and tails_alone["stage_ms_per_step"].get("gram", 0) > 0:
            # `frac` above is the launch as it runs in the timed step, with the factorisation chain, B11's epilogue tiles and the
            # early windows' B21 tiles running beside it on purpose (their vector instructions cost it matrix-pipe time; the
            # step is shorter for it).  The one-stream pass runs the same launch with nothing beside it.
            g1 = tails_alone["stage_ms_per_step"]["gram"] / gram_lps
            out["roofline"]["alone_launch_ms"] = g1
            out["roofline"]["frac_alone"] = work["ld_flops"] / gram_lps / (g1 * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS
            out["roofline"]["alone_note"] = ("the same launch in the one-stream pass (GAUSS_SIDE_STREAM=0: nothing runs beside it; the step "
                                             "is %.2f ms that way against %.2f)" % (tails_alone["ms_per_step"], tmax / args.steps * 1e3))
        if tails_alone is not None:
            n_m = np.array([len(mi) for _, mi, _ in my_wins], dtype=np.float64)
            n_u = np.array([len(ui) for _, _, ui in my_wins], dtype=np.float64)
            for key, kern, alg, what in (
                    ("roofline_pack", "gauss::pack_stats_kernel", float(np.sum((n_m + n_u) * N * 0.25)),
                     "sum over windows of (M + U) N / 4 bytes of 2-bit source rows in (the kernel also writes one operand byte per genotype)"),
                    ("roofline_epilogue", "gauss::epilogue_kernel<false>", float(np.sum((n_m * n_m + n_u * n_m) * 8.0)),
                     "sum over windows of (M^2 + U M) 8 bytes of fp64 LD entries out (the kernel also reads the exact partial slabs)")):
                ms = alone["pack_stats" if key == "roofline_pack" else "ld_epilogue"]
                # (the epilogue is launched on tile lists of different lengths; the one-stream pass timed here is its all-tiles launch)
                tr = pmc_traffic(len(wins) == 36 and len(ch["bp"]) == 100_000, kernel=kern, largest=key == "roofline_epilogue")
                ach = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                out[key] = {"kernel": kern.replace("gauss::", ""), "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg, "launch_ms": ms, "definition": what,
                            "traffic": tr.get("traffic"),
                            "moved_gbs": (tr["traffic"] / (ms * 1e-3) / 1e9) if tr.get("traffic") and ms > 0 else None,
                            "measured_on": "the one-stream pass of the same job (one launch per step for all windows)"}
                if tr.get("traffic_stale"):
                    out[key]["traffic_stale"] = True
        if tails_alone is None and gram_lps > 1:
            # no stand-alone pass (N > 1), and in the headline run the chain is timed UNDER the Gram kernel: its stage timer says
            # nothing about the fp64 kernels' own speed
            out["roofline_solve"] = None
            out["roofline_solve_note"] = "measured at N = 1 only (a one-stream pass of the same job)"
        # the launch the headline runs: ONE launch of the job's full grid (work items x 256 threads).  Counters exist for that grid from
        # the one-stream pass (counter collection serialises the queues: the merged form itself cannot run under it, its evidence is
        # the kernel trace, profiles/README.md); in the two-launch form the two halves' rows add up to the step's traffic
        out["roofline"].update(pmc_traffic(len(wins) == 36 and len(ch["bp"]) == 100_000 and rig.world == 1, largest=True,
                                           grid=stats["items"] * 256 if gram_lps == 1 else None))
        if shard_check is not None:
            out["config"]["shards_bit_identical_to_one_rank"] = shard_check
        if weak is not None:
            out["weak_scaling"] = weak
        if i8_variant is not None:
            out["int8_exact_variant"] = i8_variant
        if emu is not None:
            out["emulated_strong_scaling"] = emu
        if e2e is not None:
            out["end_to_end"] = e2e
        if spot is not None and res:
            out["parity_spot"] = parity_spot(ch, wins, spot, res, 0 if args.mode == "dist" else 1)
        if keep0 is not None:
            out["cpu_baseline"] = cpu_baseline(ch, wins, keep0, work, 0 if args.mode == "dist" else 1, light=light)
        if e2e is not None and "_other_configs" in e2e:
            out["other_configs"] = e2e.pop("_other_configs")
        if i8_variant is not None and isinstance(out.get("other_configs"), dict):
            # the exact int8 Gram path as a mode of its own (gauss_hip_set_gram_dtype / GAUSS_GRAM_DTYPE=i8): the same job, the same
            # bits, the LD GEMM on v_mfma_i32_32x32x32_i8; never the headline (`dtype` stays f32: the north star names the fp32 matrix cores)
            out["other_configs"]["int8_exact"] = {
                "metric": "imputed SNPs/s, the headline job with the LD GEMM on the int8 matrix cores (exact: identical bits)",
                "value": i8_variant["imputed_snps_per_s_this_rank"], "unit": "imputed SNPs/s", "ms_per_step": i8_variant["ms_per_step"],
                "steps": args.steps, "gram_ms": i8_variant["gram_ms"], "kernel": i8_variant["kernel"],
                "bit_identical_to_f32_path": i8_variant["bit_identical_to_f32_path"],
                "roofline": {"kernel": i8_variant["kernel"], "bound": "mfma", "achieved": i8_variant["gram_tops_algorithmic"],
                             "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s", "frac": i8_variant["gram_tops_algorithmic"] / I8_MFMA_PEAK_TOPS, "traffic": None,
                             "note": "algorithmic LD ops (same count as the f32 flops) / Gram kernel time; the kernel is bound by operand "
                                     "delivery (1 KB of LDS fragment reads per MFMA with 64 x 64 wave tiles), not by the int8 pipe"},
                "parity_spot": {"ok": bool(i8_variant["bit_identical_to_f32_path"]),
                                "what": "z / info of every window bit-identical to the f32 job's, whose parity_spot against the oracle is the headline's"}}
            out["other_configs"]["all_parity_ok"] = bool(out["other_configs"].get("all_parity_ok", True) and i8_variant["bit_identical_to_f32_path"])
        if not quiet:
            emit_line(out, headline=True)
    runner.close()
    return out


def rank_of(rig):
    return rig.rank


LINE_OUT = None          # the process's real stdout once claim_stdout() has run


def claim_stdout():
    """From here on file descriptor 1 is stderr for everybody except emit_line: RCCL prints its version banner and gloo one
    "[Gloo] Rank r is connected ..." line per rank on STDOUT from native code, and stdout is to carry ONE line."""
    global LINE_OUT
    if LINE_OUT is not None:
        return
    sys.stdout.flush()
    LINE_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)


def emit_line(out, headline):
    """The detail goes to bench_detail.json beside this file and into gpurun_out/ (the directory that travels back from a GPU
    box); stderr gets one short line saying so and stdout the ONE final line, which gauss_amd/benchline.py keeps under 6 KB
    (round 5's 28 KB line was more than the driver's reader took; stdout + stderr together now stay under 7 KB)."""
    from gauss_amd import benchline
    return benchline.emit(out, headline=headline, detail_path=os.path.join(ROOT, "bench_detail.json"),
                          also_dirs=(os.path.join(ROOT, "gpurun_out"),), stdout=LINE_OUT)


def shard_mode(args, world):
    """--shard auto: two ranks take contiguous halves of the chromosome -- neighbouring windows share half their measured SNPs
    and a rank's job multiplies the shared B11 tile pairs once (emulated efficiency 0.971 against 0.955 for scattered
    windows); from four ranks on the levelled LPT shares win (at 8: 0.864 against 0.837: a rank's few windows share
    little and every cut window's B11 is factored twice)."""
    if args.shard != "auto":
        return args.shard
    return "contiguous" if world <= 2 else "leveled"


def shares_of(args, wins, n_samples, world):
    """Per-rank shares [(window, u0, u1)] and per-rank modelled cost."""
    from gauss_amd import workload
    mode = shard_mode(args, world)
    if mode != "lpt":
        return workload.shard_balanced(wins, n_samples, world, contiguous=mode == "contiguous")
    owner, load = workload.shard(wins, n_samples, world)
    return [[(k, 0, len(wins[k][2])) for k in range(len(wins)) if owner[k] == r] for r in range(world)], load


def pieces_equal_whole(parts, ref, wins):
    """Every window's z / info, put together from the ranks' pieces, equal bit for bit what ONE job over the whole
    chromosome computed (parts: one {(window, u0, u1): (z, info)} per rank; ref: that job's results)."""
    got = {}
    for p in parts:
        got.update(p)
    for k in range(len(wins)):
        pieces = sorted(q for q in got if q[0] == k)
        if not pieces or pieces[0][1] != 0 or pieces[-1][2] != len(wins[k][2]) or any(a[2] != b[1] for a, b in zip(pieces, pieces[1:])):
            return False
        z = np.concatenate([got[q][0] for q in pieces])
        info = np.concatenate([got[q][1] for q in pieces])
        if not (np.array_equal(z, ref[k]["z"]) and np.array_equal(info, ref[k]["info"])):
            return False
    return True


def pmc_traffic(applicable, kernel="gauss::gram_kernel<float>", largest=False, grid=None):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic*.csv: separate
    FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 correction applied).  PMC counters cannot be collected from inside
    the timed run, so the number is only quoted while it describes the code that is running: the newest profile's recorded
    source hash (profiles/<tag>_provenance.json, tools/summarize_profiles.py) must equal the hash compiled into the loaded
    library (gauss_hip_source_hash); otherwise traffic is null and traffic_stale says why.  Null too when the run is not the
    profiled workload.
    grid (threads): the row of <tag>_pmc_traffic_by_grid.csv whose launch has exactly this grid -- the Gram kernel is launched on
    several grids by the profiled command (the full grid, the halves of the two-launch form the runs are demoted to under counter
    collection, the emulated shares) and an average over them describes none; traffic_form says which launch the row is."""
    import csv
    import glob
    if not applicable:
        return {"traffic": None, "traffic_source": None}
    from gauss_amd import _lib
    running = _lib.load().gauss_hip_source_hash().decode()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.csv")))
    if not files:
        return {"traffic": None, "traffic_source": None}
    f = files[-1]
    src = os.path.relpath(f, ROOT)
    profiled = None
    try:
        with open(f.replace("_pmc_traffic.csv", "_provenance.json")) as fh:
            prov = json.load(fh)
            profiled = prov.get("csrc_hash_of_loaded_library") or prov.get("csrc_hash")
    except Exception:
        pass
    if profiled != running:
        return {"traffic": None, "traffic_source": src, "traffic_stale": True,
                "traffic_note": f"{src} was collected on sources {profiled or 'unknown (no provenance file)'}, this library is {running}: "
                                "not quoted; re-collect with tools/collect_profiles.sh"}
    fg = f.replace("_pmc_traffic.csv", "_pmc_traffic_by_grid.csv")
    if grid is not None and os.path.exists(fg):
        for r in csv.DictReader(open(fg)):
            if r["kernel"] == kernel and int(r["grid_threads"]) == int(grid):
                return {"traffic": float(r["hbm_bytes_per_launch_corrected"]), "traffic_source": os.path.relpath(fg, ROOT), "traffic_stale": False,
                        "traffic_sources_hash": profiled, "traffic_form": r.get("launch_form") or None, "traffic_grid_threads": int(grid)}
        return {"traffic": None, "traffic_source": os.path.relpath(fg, ROOT), "traffic_stale": False,
                "traffic_note": f"no launch of {kernel} with a grid of {int(grid)} threads in the profiled run"}
    best = None
    for r in csv.DictReader(open(f)):
        if r["kernel"] == kernel or r["kernel"] == kernel.split("<")[0]:
            best = float(r["hbm_bytes_largest_launch_corrected"] if largest and r.get("hbm_bytes_largest_launch_corrected") else r["hbm_bytes_per_launch_corrected"])
    return {"traffic": best, "traffic_source": src, "traffic_stale": False, "traffic_sources_hash": profiled,
            "traffic_form": "largest launch of the kernel in the profiled run" if largest else "average over the kernel's launches in the profiled run"}


def usable_cores():
    """Cores this process may really use: its affinity mask, cut by a cgroup CPU quota if one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                t = fh.read().split()
            if path.endswith("cpu.max"):
                if t[0] != "max":
                    n = min(n, max(1, int(int(t[0]) / int(t[1]))))
            else:
                q = int(t[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                        n = min(n, max(1, q // int(fh.read())))
            break
        except Exception:
            continue
    return max(1, n)


def cpu_baseline(ch, wins, keep0, work, mode=1, light=False):
    """The loop-literal CPU oracle (1 thread, like the reference) on a bounded sample, scaled to the workload in two
    parts.  (1) The pair loops: the reference's cost is N inner iterations per SNP pair (util.cpp:103-124),
    M(M+1)/2 + U + U*M pairs per window (distmix.cpp:180-217) -- the sample run's time, less its own dense tail, scaled
    by the workload's pair count.  (2) The dense tail, MakePosDef's eigen-decomposition + InvMat's full-pivot LU
    (util.cpp:298-318, cubic in M): timed at the sample's, the mean and the largest window's M, and summed over the
    windows as c * M^3."""
    import oracle
    k0, gm, gu = keep0
    _, mi, ui = wins[k0]
    N = int(ch["off"][-1])
    ms_ = min(600, gm.shape[0], 150 if N < 5000 else 600)
    gm_h = np.ascontiguousarray(gm[:ms_, :N])
    gu_h = np.ascontiguousarray(gu[:600, :N])
    z1 = ch["z"][mi][:ms_]
    t0 = time.perf_counter()
    oracle.run_impute(mode, gm_h, gu_h, ch["off"], ch["w"], z1)
    t = time.perf_counter() - t0
    m, u = gm_h.shape[0], gu_h.shape[0]

    def dense_tail_s(size):
        rng = np.random.default_rng(size)
        x = rng.standard_normal((size, 2 * size))
        a = np.corrcoef(x) + 0.1 * np.eye(size)                 # B11-shaped: a correlation matrix + lambda I (dist.cpp:172)
        t1 = time.perf_counter()
        a2, _ = oracle.make_pos_def(a)                          # util.cpp:302-318 (decomposes whether or not it clamps)
        oracle.inv_mat(a2)                                      # util.cpp:298-300
        return time.perf_counter() - t1

    ms_all = np.array([len(a) for _, a, _ in wins], dtype=np.float64)
    sizes = sorted({int(m)} if light else {int(m), int(round(ms_all.mean())), int(ms_all.max())})
    tails = {s: dense_tail_s(s) for s in sizes}
    c3 = float(np.mean([tails[s] / s ** 3 for s in sizes]))
    tail_total = float(c3 * np.sum(ms_all ** 3))
    pairs_sample = m * (m + 1) / 2 + u + u * m
    pairs_total = sum(len(a) * (len(a) + 1) / 2 + len(b) + len(a) * len(b) for _, a, b in wins)
    t_pairs = max(t - c3 * m ** 3, 0.5 * t)
    est = t_pairs * pairs_total / pairs_sample + tail_total
    # what an R user could do with one process per window: the same sample on several cores at once
    # (threads here: the oracle is plain C behind ctypes, which releases the GIL)
    from concurrent.futures import ThreadPoolExecutor
    # one process per window on every core the host gives us ("a windows-in-parallel run on all cores"): as
    # many concurrent copies as there are windows to hand out, at most one per usable core
    par = max(1, min(usable_cores(), len(wins)))
    if light:
        return {"value": work["imputed_snps"] / est, "unit": "imputed SNPs/s", "cores": 1, "kind": "port", "host_cores": os.cpu_count(),
                "sample": f"oracle run_{'distmix' if mode else 'dist'} on a sub-window of window {k0} (M={m}, U={u}, N={N}): {t:.2f} s for "
                          f"{pairs_sample:.0f} SNP pairs, pair loops scaled by the workload's {pairs_total:.3g} pairs, dense tail c M^3 from "
                          f"M = {m}: estimated {est:.0f} s per chromosome (bounded sample of the other_configs leg)"}
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=par) as pool:
        list(pool.map(lambda _: oracle.run_impute(mode, gm_h, gu_h, ch["off"], ch["w"], z1), range(par)))
    tp = time.perf_counter() - t0
    est_par = (tp / par) * (est / t)                            # same scaling, per-copy time under memory contention
    return {
        "value": work["imputed_snps"] / est, "unit": "imputed SNPs/s", "cores": 1, "kind": "port",
        "host_cores": os.cpu_count(), "usable_cores": usable_cores(),
        "windows_in_parallel": {"value": work["imputed_snps"] / est_par, "unit": "imputed SNPs/s", "cores": par,
                                "speedup_over_one_core": est / est_par,
                                "sample": f"{par} concurrent copies of the same sample (min(usable cores, windows)) in {tp:.2f} s"},
        "dense_tail": {"what": "orc_make_pos_def + orc_inv_mat (MakePosDef eigen-decomposition + InvMat full-pivot LU, util.cpp:298-318)",
                       "seconds_at_M": {str(s): round(tails[s], 3) for s in sizes}, "seconds_per_M3": c3,
                       "seconds_per_chromosome": round(tail_total, 1), "share_of_estimate": round(tail_total / est, 4)},
        "sample": f"oracle run_{'distmix' if mode else 'dist'} on a sub-window of window {k0} (M={m}, U={u}, N={N}): {t:.2f} s for "
                  f"{pairs_sample:.0f} SNP pairs, pair loops scaled by the workload's {pairs_total:.3g} pairs, plus the dense tail "
                  f"timed at M = {', '.join(str(s) for s in sizes)} and summed over the {len(wins)} windows as c M^3 "
                  f"({tail_total:.0f} s): estimated {est:.0f} s per chromosome",
    }


def cpu_baseline_computeld(sample):
    """computeLD's pair loops (computeLD.cpp:95-116) in the CPU oracle on a bounded sample of the window's rows,
    scaled by the pair count M (M + 1) / 2 + M to the whole window: LD matrices per second on one core.  The sample's LD matrix is
    left in sample["oracle_ld"] (the parity spot of the other_configs leg compares the GPU's entries with it)."""
    import oracle
    g = np.ascontiguousarray(sample["geno"])
    t0 = time.perf_counter()
    sample["oracle_ld"] = oracle.compute_ld(g, sample["off"], sample["w"])
    t = time.perf_counter() - t0
    m, M = g.shape[0], sample["M"]
    est = t * (M * (M + 1) / 2 + M) / (m * (m + 1) / 2 + m)
    return {"value": 1.0 / est, "unit": "LD matrices/s", "cores": 1, "kind": "port", "host_cores": os.cpu_count(),
            "sample": f"oracle compute_ld on {m} of the window's {M} SNPs (N={sample['N']}): {t:.2f} s, scaled by pair count to "
                      f"{est:.1f} s per window"}


def jepegmix_checks(pr, blocks, go, sizes, spot_genes, n_table, warm_s):
    """cpu_baseline + parity_spot of the jepegmix leg: CorG's pair loops (gene.cpp:571-586: CalWgtCov per SNP pair of a gene) of the
    first `spot_genes` genes in the CPU oracle, scaled by pair count to all genes, one core; and the GPU's gene LD blocks of the
    same genes against those oracle matrices (the k x k tail is host code in both, tests/test_oracle.py).  The oracle is the
    checker here, never the thing measured."""
    import oracle
    G = pr.geno_m()                                               # one byte per genotype, the rows the GPU batch multiplied
    off, w = np.ascontiguousarray(pr.pop_off(), np.int32), np.ascontiguousarray(pr.pop_wgt(), np.float64)
    pick = [g for g in range(len(sizes)) if sizes[g] >= 2][:spot_genes]
    worst, t_or, pairs_s = 0.0, 0.0, 0
    for g in pick:
        rows = np.ascontiguousarray(G[go[g]:go[g + 1]])
        t0 = time.perf_counter()
        want = oracle.compute_ld(rows, off, w)
        t_or += time.perf_counter() - t0
        got = blocks[g]
        n = rows.shape[0]
        iu = np.triu_indices(n, 1)
        worst = max(worst, float(np.max(np.abs(got[iu] - want[iu]))) if n > 1 else 0.0)
        pairs_s += n * (n + 1) // 2
    pairs_all = int((sizes * (sizes + 1) // 2).sum())
    est = t_or * pairs_all / max(pairs_s, 1)
    return {
        "cpu_baseline": {"value": n_table / est if est > 0 else None, "unit": "genes/s", "cores": 1, "kind": "port", "host_cores": os.cpu_count(),
                         "sample": f"oracle CalWgtCov pair loops (gene.cpp:571-586) of {len(pick)} genes ({pairs_s} SNP pairs, N = {G.shape[1]}): {t_or:.2f} s, "
                                   f"scaled by pair count to all {len(sizes)} genes ({pairs_all} pairs): {est:.1f} s per call; the k x k tails (microseconds) "
                                   "and the reference's text feeder are not included"},
        "parity_spot": {"genes": len(pick), "max_abs_ld_diff": worst, "tolerance": 1e-12, "ok": bool(worst <= 1e-12),
                        "what": "off-diagonal entries of the GPU's gene LD blocks (gauss_gene_ld_batch_rows on the resident 2-bit panel) "
                                "vs oracle.compute_ld on the same genotype rows at full N"},
    }


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    if argv is None:
        claim_stdout()
    rig = Rig(args)
    if args.mode in ("distmix", "dist"):
        out = run_impute(args, rig)
    else:
        from gauss_amd import benchmodes
        if args.gpus > 1 and args.mode != "e2e":
            raise SystemExit(f"--mode {args.mode} is a single-GPU measurement")
        if args.mode == "e2e":
            out = benchmodes.run_e2e(args, rig)
        else:
            if args.mode == "jepegmix":
                out, sample = benchmodes.run_jepegmix(args, rig, checks=None if args.no_cpu_baseline else jepegmix_checks)
            else:
                out, sample = {"computeLD": benchmodes.run_computeld, "window": benchmodes.run_window}[args.mode](args, rig)
            if out is not None and sample is not None and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline_computeld(sample)
            if out is not None:
                emit_line(out, headline=False)
    rig.close()           # not in a `finally`: a rank that failed must not wait in a barrier for the others
    if isinstance(out, dict) and isinstance(out.get("other_configs"), dict) and not out["other_configs"].get("all_parity_ok", True):
        print("bench.py: a parity_spot of the other_configs block failed", file=sys.stderr)
        sys.exit(4)
    if isinstance(out, dict) and isinstance(out.get("parity_spot"), dict) and not out["parity_spot"]["ok"]:
        print("bench.py: the timed job's results differ from the CPU oracle beyond %g (parity_spot)" % SPOT_TOL, file=sys.stderr)
        sys.exit(4)
    return out


if __name__ == "__main__":
    main()
