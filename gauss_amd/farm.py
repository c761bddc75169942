"""Window farm: tile a chromosome into prediction windows and spread them over the GPUs of a node.

The reference imputes one window per R call (dist.cpp:30-126 builds and destroys everything per
call); the caller-level loop over windows is not part of it.  Windows are independent units
(SURVEY.md section 8e), so the farm shards them across ranks with no data-path collective: one
process per GPU, every rank runs its windows as one batched job, and only the finished tables are
gathered (torch.distributed, RCCL/gloo) and concatenated in window order on rank 0.
"""
import os

import numpy as np

from . import api, hotpath


def make_windows(start_bp, end_bp, window_size=1_000_000):
    """Prediction windows [s, s + window_size - 1] covering [start_bp, end_bp]."""
    out = []
    s = int(start_bp)
    while s <= end_bp:
        out.append((s, min(int(end_bp), s + int(window_size) - 1)))
        s += int(window_size)
    return out


def assign_windows(costs, world_size):
    """Longest-processing-time assignment of windows to ranks; deterministic on every rank.
    Returns a list (len = n windows) of owning ranks."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world_size
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += costs[i]
    return owner


def _units32(rows, edge16=False):
    """32-row units the Gram kernel ISSUES for `rows` live rows of ONE operand side, tile by tile (k_gram.hip): every 64-row
    wave half rounds its live rows up to 32; on the column side a half whose last 32 hold at most 16 live columns works in
    groups of 16 (the 16-column edge routine) -- the rule gauss_job_stats applies to a built job."""
    total = 0.0
    for t0 in range(0, rows, 128):
        r = min(128, rows - t0)
        for w in (0, 1):
            left = r - 64 * w
            if left <= 0:
                continue
            n16 = min(4, (left + 15) // 16)
            total += n16 * 0.5 if (edge16 and n16 % 2) else min(2, (left + 31) // 32)
    return total


def _b11_units(m):
    """32 x 32 MFMA tiles the Gram kernel issues for a window's own B11 (upper tile triangle; a diagonal tile skips its mirrored
    quadrant and the mirrored sub-block of its two diagonal quadrants)."""
    nt = (m + 127) // 128
    rows = [min(128, m - 128 * t) for t in range(nt)]
    total = 0.0
    for ti in range(nt):
        a = [min(2, max(0, (rows[ti] - 64 * w + 31) // 32)) for w in (0, 1)]
        for tj in range(ti, nt):
            for wr in (0, 1):
                for wc in (0, 1):
                    if ti == tj and wr == 1 and wc == 0:
                        continue
                    left = rows[tj] - 64 * wc
                    if left <= 0 or a[wr] == 0:
                        continue
                    n16 = min(4, (left + 15) // 16)
                    if n16 % 2:
                        total += a[wr] * n16 * 0.5
                        continue
                    t32 = a[wr] * min(2, (left + 31) // 32)
                    total += 3 if (ti == tj and wr == wc and t32 == 4) else t32
    return total


# What the ranks' step times follow (round 4, 8-rank emulation on MI355X: Gram time per ISSUED flop agrees to +-2 % between
# shares whose algorithmic flops differ by 4.5 %; everything that is not the Gram kernel is 0.58-0.61 ms on every rank): the
# flops the Gram kernel issues for the rank's pieces -- 128-row tiles with the kernel's 32 / 16 granular edges, B11's tile
# triangle -- plus 8 % on B21's share for what follows it per entry (B21's epilogue tiles, the closing product).  The
# factorisation chain runs UNDER the Gram kernel (k_solve_lite.hip), so B11's fp64 work no longer costs a rank time.
TAIL_PER_B21 = 0.08


def piece_cost(n_samples, m, u):
    """Cost of imputing `u` unmeasured SNPs of a window with `m` measured ones, in issued fp32-matrix-flop units:
    (setup, per unmeasured SNP).  The setup -- B11's tile pairs -- is paid by every rank that holds a piece of the window."""
    setup = 2048.0 * n_samples * _b11_units(m)
    per_u = 2.0 * n_samples * 32.0 * _units32(m, edge16=True) * (1.0 + TAIL_PER_B21)
    return setup, per_u


def balance_windows(mu, n_samples, world_size, granule=64):
    """Contiguous, balanced shares of a chromosome's windows with window SPLITTING.

    A window's imputed SNPs are independent given its measured set (dist.cpp:181-198: one row of B21, one solve,
    per unmeasured SNP), so a window may be cut into pieces that share the measured SNPs and divide the unmeasured
    ones; each piece repeats the B11 work.  Walking the windows in chromosome order, every rank is filled up to a
    common load T and the window that straddles the boundary is cut (at a multiple of `granule` unmeasured SNPs);
    the smallest feasible T is found by bisection.  At most world_size - 1 windows are cut, the ranks' loads agree
    to within one granule, and every rank's windows cover one contiguous stretch of the chromosome (so its slice of
    the panel is about 1 / world_size of it, plus the wings).

    mu = [(M, U)] per window.  Returns (shares, loads): shares[r] = [(window, u0, u1)], deterministic on every rank."""
    costs = [piece_cost(n_samples, m, u) for m, u in mu]
    total = sum(b + u * r for (b, r), (_, u) in zip(costs, mu))
    if world_size <= 1 or not mu:
        return [[(k, 0, u) for k, (_, u) in enumerate(mu)]] + [[] for _ in range(world_size - 1)], [total] + [0.0] * (world_size - 1)

    def fill(T):
        shares, loads, cap = [[]], [0.0], T
        for k, (_, U) in enumerate(mu):
            b, r = costs[k]
            u0 = 0
            while u0 < U:
                need = b + (U - u0) * r
                if need <= cap:
                    shares[-1].append((k, u0, U)); loads[-1] += need; cap -= need; u0 = U
                    continue
                room = int(max(0.0, cap - b) // r) // granule * granule
                if room >= granule and U - u0 - room >= granule:
                    shares[-1].append((k, u0, u0 + room)); loads[-1] += b + room * r; u0 += room
                elif not shares[-1]:
                    return None                      # T does not even hold this piece on an empty rank
                if len(shares) == world_size:
                    return None
                shares.append([]); loads.append(0.0); cap = T
        while len(shares) < world_size:
            shares.append([]); loads.append(0.0)
        return shares, loads

    lo, hi = total / world_size, total + 1.0
    best = fill(hi)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        got = fill(mid)
        if got is None:
            lo = mid
        else:
            best, hi = got, mid
    return best


# One block step (64 measured SNPs) of a job's factorisation chain in piece_cost units: the chain is as long as the
# job's tallest window.  It used to be latency on the critical path of a small job (19 us per step against 8.06 ms per 1e12
# cost units: 2.4e9); since round 3 it runs UNDER the Gram kernel (k_solve_lite.hip) and costs the rank only the vector
# instructions it takes from the matrix pipe.  At 2.4e9 the planner kept Gram work away from the ranks that hold the tall windows --
# 7 % of a rank's load that no longer exists: their Gram kernels finished 4 % early (round 4 emulation) -- so the term is
# down to a tie-breaker that still steers tall windows onto the same ranks.
CHAIN_STEP_COST = 2.0e8


def adjacency_saving(n_samples, m_a, m_b, shared):
    """What a rank saves when it holds BOTH of two neighbouring windows (m_a, m_b measured SNPs, the last `shared` of a's being the
    first of b's): the job then keeps one list of measured rows for the two (gauss_plan.cpp: shared measured rows, clusters) and
    multiplies the B11 tile pairs that lie in both windows once.  Counted the way the job builder decides it: window b joins
    a's cluster at row offset m_a - shared if that adds no more tile pairs than tiles of its own would; the saving is the
    difference, priced like B11's other pairs (piece_cost's setup per pair of full tiles, less the average edge skipping)."""
    if shared <= 0 or m_b <= 0:
        return 0.0
    pos = m_a - shared
    lo, hi = pos // 128, (pos + m_b - 1) // 128
    a_hi = (m_a - 1) // 128
    add = sum(1 for ti in range(lo, hi + 1) for tj in range(ti, hi + 1) if not (tj <= a_hi))
    mt = (m_b + 127) // 128
    own = mt * (mt + 1) // 2
    if add > own:
        return 0.0
    per_pair = 2048.0 * n_samples * _b11_units(m_b) / own          # B11's issued units of window b, per tile pair of its own
    return (own - add) * per_pair


def level_windows(mu, n_samples, world_size, granule=64, shared=None):
    """Whole windows by LPT, a local search, then LEVELLING with window cuts.

    A rank's load is the cost of its pieces plus the factorisation chain of its tallest window (CHAIN_STEP_COST per
    64 measured SNPs).  (1) LPT on the windows' costs; (2) while it lowers the largest load, one window of the most
    loaded rank moves to, or trades places with a window of, another rank -- tall windows end up sharing ranks, the
    other ranks' chains get short; (3) while it lowers the largest load, the most loaded rank hands a slice of one
    window's unmeasured SNPs to the least loaded rank, which factors that window's B11 itself (piece_cost's setup):
    in practice a window with few measured SNPs, whose B11 is cheap to repeat.  On the chr22 study at 8 ranks the
    largest modelled load goes 1.07 -> 1.01 of the mean for < 1 % of repeated work; contiguous shares
    (balance_windows) repeat 3.2 %.

    mu = [(M, U)] per window.  Returns (shares, loads): shares[r] = [(window, u0, u1)], deterministic on every rank."""
    costs = [piece_cost(n_samples, m, u) for m, u in mu]
    nblk = [(m + 63) // 64 for m, _ in mu]

    # shared[k]: measured SNPs windows k and k + 1 have in common (None: unknown, no adjacency term).  A rank that holds two
    # neighbouring windows multiplies their common B11 tile pairs once (adjacency_saving); pieces of a cut window count too --
    # every piece carries the window's whole B11.
    adj = [0.0] * len(mu)
    if shared is not None:
        for k in range(len(mu) - 1):
            adj[k] = adjacency_saving(n_samples, mu[k][0], mu[k + 1][0], int(shared[k]))

    def load_of(pieces):
        held = {k for k, _, _ in pieces}
        return (sum(costs[k][0] + (u1 - u0) * costs[k][1] for k, u0, u1 in pieces)
                - sum(adj[k] for k in held if k + 1 in held)
                + CHAIN_STEP_COST * max((nblk[k] for k, _, _ in pieces), default=0))

    owner = assign_windows([b + u * r for (b, r), (_, u) in zip(costs, mu)], world_size)
    shares = [[(k, 0, mu[k][1]) for k in range(len(mu)) if owner[k] == r] for r in range(world_size)]
    if world_size > 1:
        for _ in range(4 * len(mu)):                                     # (2) moves and swaps of whole windows
            loads = [load_of(sh) for sh in shares]
            hi = max(range(world_size), key=lambda r: (loads[r], -r))
            best = None
            for a in shares[hi]:
                rest = [p for p in shares[hi] if p != a]
                for r in range(world_size):
                    if r == hi:
                        continue
                    for b in [None] + shares[r]:
                        g_hi = rest + ([b] if b else [])
                        g_r = [p for p in shares[r] if p != b] + [a]
                        m = max(load_of(g_hi), load_of(g_r))
                        if m < loads[hi] * (1 - 1e-9) and (best is None or m < best[0]):
                            best = (m, r, g_hi, g_r)
            if best is None:
                break
            _, r, shares[hi], shares[r] = best
        if shared is not None:
            # (2b) neighbours together: any two ranks trade or move a window when that lowers the LARGER of their two loads (the
            # largest load of all never grows); with the adjacency term this is what collects neighbouring windows on one rank
            for _ in range(8 * len(mu)):
                loads = [load_of(sh) for sh in shares]
                best = None
                for ra in range(world_size):
                    for rb in range(ra + 1, world_size):
                        cur = max(loads[ra], loads[rb])
                        for a in [None] + shares[ra]:
                            for b in [None] + shares[rb]:
                                if a is None and b is None:
                                    continue
                                g_a = [p for p in shares[ra] if p != a] + ([b] if b else [])
                                g_b = [p for p in shares[rb] if p != b] + ([a] if a else [])
                                m = max(load_of(g_a), load_of(g_b))
                                gain = cur - m
                                if gain > 1e-9 * cur and (best is None or gain > best[0]):
                                    best = (gain, ra, rb, g_a, g_b)
                if best is None:
                    break
                _, ra, rb, shares[ra], shares[rb] = best
        for _ in range(2 * world_size):                                  # (3) cuts
            loads = [load_of(sh) for sh in shares]
            hi = max(range(world_size), key=lambda r: (loads[r], -r))
            lo = min(range(world_size), key=lambda r: (loads[r], r))
            best = None
            for idx, (k, u0, u1) in enumerate(shares[hi]):
                b, r = costs[k]
                extra = load_of(shares[lo] + [(k, 0, 0)]) - loads[lo]     # B11 again, and maybe a longer chain
                d = min(int((loads[hi] - loads[lo] - extra) / (2.0 * r)) // granule * granule, (u1 - u0) - granule)
                if d >= granule:
                    g_hi = shares[hi][:idx] + [(k, u0, u1 - d)] + shares[hi][idx + 1:]
                    g_lo = shares[lo] + [(k, u1 - d, u1)]
                    m = max(load_of(g_hi), load_of(g_lo))
                    if m < loads[hi] * (1 - 1e-9) and (best is None or m < best[0]):
                        best = (m, g_hi, g_lo)
            if best is None:
                break
            _, shares[hi], shares[lo] = best
    shares = [sorted(sh) for sh in shares]
    return shares, [load_of(sh) for sh in shares]


def window_cost(n_samples, m, u):
    """Pair-loop cost of one window: N * (M(M+1)/2 + U*M)."""
    return float(n_samples) * (m * (m + 1) / 2.0 + float(u) * m)


def gpu_compute(prepared_list, ctx=None, resident=True, timings=None):
    """Run the prepared windows as ONE batched job on this rank's GPU; returns one table per window.

    Windows that read a packed panel name their genotype rows by index.  With `resident` the slice of the
    panel that this rank's windows touch is uploaded once (gauss_store_upload) and the pack kernel gathers the
    rows from HBM; overlapping windows (the 500 kb wings) then share one copy instead of uploading twice."""
    if not prepared_list:
        return []
    ctx = ctx or hotpath.default_context()
    lib = ctx.lib
    import ctypes as C
    descs = (hotpath.WindowDesc * len(prepared_list))()
    for i, p in enumerate(prepared_list):
        d = p.window_desc()
        C.memmove(C.byref(descs[i]), C.byref(d), C.sizeof(d))
    import time
    t0 = time.perf_counter()
    stores = [p.packed_store() for p in prepared_list]
    store, keep, on_device = None, [], 0
    if resident and all(s is not None and s == stores[0] for s in stores):
        base, nbytes, rb = stores[0]
        rows = []
        for d in descs:
            rm = np.ctypeslib.as_array(d.rows_m, shape=(d.n_measured,))
            ru = np.ctypeslib.as_array(d.rows_u, shape=(d.n_unmeasured,)) if d.n_unmeasured else np.zeros(0, np.int32)
            rows.append((rm, ru))
        lo = min(int(min(rm.min(), ru.min() if len(ru) else rm.min())) for rm, ru in rows)
        hi = max(int(max(rm.max(), ru.max() if len(ru) else rm.max())) for rm, ru in rows) + 1
        dev = C.c_void_p()
        hotpath.check(lib.gauss_store_upload(ctx.handle, C.c_void_p(base + lo * rb), (hi - lo) * rb, C.byref(dev)))
        store, on_device = dev, 1
        for d, (rm, ru) in zip(descs, rows):
            a, b = np.ascontiguousarray(rm - lo, dtype=np.int32), np.ascontiguousarray(ru - lo, dtype=np.int32)
            keep += [a, b]
            d.rows_m = a.ctypes.data_as(C.POINTER(C.c_int32))
            d.rows_u = b.ctypes.data_as(C.POINTER(C.c_int32))
            d.geno_m = d.geno_u = dev.value
    h = C.c_void_p()
    t1 = time.perf_counter()
    try:
        hotpath.check(lib.gauss_job_create(ctx.handle, descs, len(prepared_list), on_device, C.byref(h)))
        t2 = time.perf_counter()
        try:
            hotpath.check(lib.gauss_job_run(h))
            hotpath.check(lib.gauss_job_fetch(h))
            if timings is not None:
                timings.update(store_upload_s=t1 - t0, job_create_s=t2 - t1, run_fetch_s=time.perf_counter() - t2)
        finally:
            lib.gauss_job_destroy(h)
    finally:
        if store is not None:
            lib.gauss_store_free(ctx.handle, store)
    return [p.finish() for p in prepared_list]


def impute_chromosome_native(kind, chr, start_bp, end_bp, wing_size, input_file, reference_data_file, reference_pop_desc_file,
                             study_pop=None, pop_wgt_df=None, af1_cutoff=None, window_size=1_000_000, group=None, ctx=None,
                             n_batches=0):
    """The farm on a PACKED panel, natively: every rank makes ONE call into libgauss_host
    (gauss_host_impute_chromosome: the same LPT plan on every rank, this rank's windows pipelined through its GPU in
    batches against the resident panel), then the ranks' column arrays are gathered on rank 0 and merged in window
    order.  Returns an api.ChromResult on rank 0 (or the single process), None elsewhere."""
    dist, rank, world = None, 0, 1
    try:
        import torch.distributed as dist_mod
        if dist_mod.is_available() and dist_mod.is_initialized():
            dist = dist_mod
            rank, world = dist.get_rank(group), dist.get_world_size(group)
    except ImportError:     # pragma: no cover
        pass
    ctx = ctx or hotpath.default_context()          # this rank's own device (LOCAL_RANK)
    res = api.impute_chromosome(kind, chr, start_bp, end_bp, wing_size, input_file, reference_data_file, reference_pop_desc_file,
                                study_pop=study_pop, pop_wgt_df=pop_wgt_df, af1_cutoff=af1_cutoff, window_size=window_size,
                                rank=rank, world=world, n_batches=n_batches, ctx=ctx)
    if dist is None or world == 1:
        return res
    parts = [None] * world if rank == 0 else None
    dist.gather_object(res, parts, dst=0, group=group)
    return api.ChromResult.merge(parts) if rank == 0 else None


def impute_chromosome(kind, chr, start_bp, end_bp, wing_size, input_file, reference_index_file, reference_data_file,
                      reference_pop_desc_file, study_pop=None, pop_wgt_df=None, af1_cutoff=None,
                      window_size=1_000_000, compute=gpu_compute, group=None, threads=None, timings=None):
    """dist()/distmix() over every window of [start_bp, end_bp], sharded across the ranks of `group`.

    kind: api.KIND_DIST / KIND_DISTMIX (imputation) or api.KIND_QCAT / KIND_QCATMIX (QC test).  Returns on rank 0 (or the single process) a dict
    {"table": DataFrame of all windows in window order, "skipped": [(window, reason), ...]};
    other ranks return None.  Windows that fail the reference's ">10 measured / >10 unmeasured"
    guard (dist.cpp:145-151) are reported in "skipped" instead of aborting the run.
    """
    import pandas as pd
    dist = None
    rank, world = 0, 1
    try:
        import torch.distributed as dist_mod
        if dist_mod.is_available() and dist_mod.is_initialized():
            dist = dist_mod
            rank, world = dist.get_rank(group), dist.get_world_size(group)
    except ImportError:     # pragma: no cover
        pass

    windows = make_windows(start_bp, end_bp, window_size)
    # cheap, rank-independent cost estimate from the GWAS file alone (measured SNP density)
    bp = _gwas_positions(input_file, chr)
    costs = []
    for s, e in windows:
        m = int(np.count_nonzero((bp >= s - wing_size) & (bp <= e + wing_size)))
        costs.append(m * m + 1.0)
    owner = assign_windows(costs, world)

    import time
    from concurrent.futures import ThreadPoolExecutor
    mine = [i for i in range(len(windows)) if owner[i] == rank]
    prepared, ok_idx, skipped = [], [], []

    def prep(i):
        # the host data layer of one window (BGZF inflate + parse + AF filter + partition) is plain C++
        # behind ctypes, which releases the GIL: windows decode in parallel on the host cores
        s, e = windows[i]
        try:
            return i, api.Prepared(kind, chr=chr, start_bp=s, end_bp=e, wing_size=wing_size, study_pop=study_pop,
                                   pop_wgt_df=pop_wgt_df, input_file=input_file,
                                   reference_index_file=reference_index_file, reference_data_file=reference_data_file,
                                   reference_pop_desc_file=reference_pop_desc_file, af1_cutoff=af1_cutoff), None
        except api.GaussError as ex:
            return i, None, str(ex)

    t0 = time.perf_counter()
    cores = max(1, (os.cpu_count() or 2) // max(1, world))
    nthreads = threads or min(16, cores, max(1, len(mine)))
    api.set_host_threads(max(1, min(16, cores) // nthreads))      # windows x lines-per-window parallelism
    with ThreadPoolExecutor(max_workers=nthreads) as pool:
        results = list(pool.map(prep, mine))
    for i, p, err in results:
        if err is not None:
            skipped.append((i, err))
        elif p.M <= 10 or (p.U <= 10 and kind != api.KIND_QCAT):       # qcat.cpp:157 guards on measured SNPs only
            skipped.append((i, f"Not enough number of SNPs loaded (measured {p.M}, unmeasured {p.U})"))
            p.close()
        else:
            prepared.append(p)
            ok_idx.append(i)
    t1 = time.perf_counter()
    tables = compute(prepared)
    t2 = time.perf_counter()
    if timings is not None:
        timings.update(feeder_s=t1 - t0, compute_s=t2 - t1, windows=len(prepared), threads=nthreads,
                       imputed=int(sum(p.U for p in prepared)))
    for p in prepared:
        p.close()
    local = {"tables": dict(zip(ok_idx, tables)), "skipped": skipped}

    if dist is not None and world > 1:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(local, gathered, dst=0, group=group)
        if rank != 0:
            return None
        parts = gathered
    else:
        parts = [local]
    by_window, skipped_all = {}, []
    for part in parts:
        by_window.update(part["tables"])
        skipped_all += part["skipped"]
    frames = [by_window[i] for i in sorted(by_window)]
    table = pd.concat(frames, ignore_index=True) if frames else pd.DataFrame()
    return {"table": table, "skipped": sorted((windows[i], why) for i, why in skipped_all),
            "windows": windows, "owner": owner}


def _dist_state(group):
    try:
        import torch.distributed as dist_mod
        if dist_mod.is_available() and dist_mod.is_initialized():
            return dist_mod, dist_mod.get_rank(group), dist_mod.get_world_size(group)
    except ImportError:     # pragma: no cover
        pass
    return None, 0, 1


def jepeg(kind, input_file, annotation_file, reference_index_file, reference_data_file, reference_pop_desc_file, study_pop=None,
          pop_wgt_df=None, af1_cutoff=None, compute=None, group=None, ctx=None):
    """jepeg() / jepegmix() (kind = api.KIND_JEPEG / KIND_JEPEGMIX) with the GENES of the one call split over the ranks of `group`
    (BASELINE.json configs[4]).  Genes are independent (jepeg.cpp:114-131, gauss.cpp:1383-1439): every rank
    runs the same host data layer on the same files, derives the same plan (contiguous gene ranges of equal cost,
    gauss_prepared_jepeg_plan) and computes CorG and the k x k tails of its range only -- no data-path collective; the ranks'
    tables are gathered on rank 0 and concatenated in rank order, which is gene order.  Returns on rank 0 (or the single process)
    {"table": DataFrame, "ranges": [(g0, g1)] per rank}; None on the other ranks.

    compute = None: the native call on this rank's GPU (gauss_host_jepeg_rank).  compute = fn(prepared, g0, g1) -> list of CorG
    blocks (n_g x n_g, diagonal 1 + lambda): the Python form of the same split, for harnesses that compute CorG themselves (the
    CPU tests put the oracle there); plan and tails are the library's (gauss_prepared_jepeg_plan / _finish)."""
    import pandas as pd
    dist, rank, world = _dist_state(group)
    if compute is None:
        tab, (g0, g1, ng) = api.jepeg_rank(kind, input_file, annotation_file, reference_index_file, reference_data_file,
                                           reference_pop_desc_file, study_pop=study_pop, pop_wgt_df=pop_wgt_df, af1_cutoff=af1_cutoff,
                                           rank=rank, world=world, ctx=ctx or hotpath.default_context())
    else:
        pr = api.Prepared(kind, study_pop=study_pop, pop_wgt_df=pop_wgt_df, input_file=input_file, annotation_file=annotation_file,
                          reference_index_file=reference_index_file, reference_data_file=reference_data_file,
                          reference_pop_desc_file=reference_pop_desc_file, af1_cutoff=af1_cutoff)
        try:
            first = pr.jepeg_plan(world)
            g0, g1, ng = first[rank], first[rank + 1], first[-1]
            tab = pr.jepeg_finish(g0, g1, compute(pr, g0, g1) if g1 > g0 else [])
        finally:
            pr.close()
    local = {"table": tab, "range": (g0, g1), "genes": ng}
    if dist is None or world == 1:
        return {"table": tab, "ranges": [(g0, g1)]}
    parts = [None] * world if rank == 0 else None
    dist.gather_object(local, parts, dst=0, group=group)
    if rank != 0:
        return None
    ranges = [p["range"] for p in parts]
    # every rank derived the same plan: the ranges tile [0, genes) in rank order
    assert ranges[0][0] == 0 and ranges[-1][1] == parts[0]["genes"] and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])), ranges
    return {"table": pd.concat([p["table"] for p in parts], ignore_index=True), "ranges": ranges}


def jepeg_genome(kind, calls, reference_pop_desc_file, study_pop=None, pop_wgt_df=None, af1_cutoff=None, group=None, ctx=None):
    """One jepeg() / jepegmix() call per entry of `calls` = [(input_file, annotation_file, reference_index_file,
    reference_data_file)] -- the reference's user runs one per chromosome -- dealt WHOLE to the ranks (gauss_host_jepeg_genome: longest
    annotation first onto the least loaded rank; the same deal on every rank, no communication).  A call is host-bound, so this is
    the split that scales.  Returns on rank 0 {"tables": [DataFrame per call], "owner": [rank per call]}; None elsewhere."""
    dist, rank, world = _dist_state(group)
    tabs, owner = api.jepeg_genome(kind, calls, reference_pop_desc_file, study_pop=study_pop, pop_wgt_df=pop_wgt_df,
                                   af1_cutoff=af1_cutoff, rank=rank, world=world, ctx=ctx or hotpath.default_context())
    if dist is None or world == 1:
        return {"tables": tabs, "owner": owner}
    parts = [None] * world if rank == 0 else None
    dist.gather_object(tabs, parts, dst=0, group=group)
    if rank != 0:
        return None
    return {"tables": [parts[owner[c]][c] for c in range(len(calls))], "owner": owner}


def _gwas_positions(path, chr):
    bp = []
    with open(path) as f:
        f.readline()
        for line in f:
            t = line.split()
            if len(t) >= 3 and (chr <= 0 or int(t[1]) == chr):
                bp.append(int(t[2]))
    return np.array(bp, dtype=np.int64)
