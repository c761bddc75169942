"""Window farm: sharding plan, and the N > 1 path under gloo on CPU (world_size 2).

The CPU ranks cannot run the HIP library, so the farm's `compute` hook is replaced by the CPU
oracle here (allowed: tests may use the oracle as the checker); what is under test is the
sharding, the per-rank batching and the gather/concatenate step -- the part that differs between
1 and N ranks.  The GPU compute itself is covered by test_gpu_farm_single_rank (-m gpu)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from gauss_amd import api, farm, panel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POPS = [("AAA", 120, "EUR"), ("BBB", 95, "EUR"), ("CCC", 110, "ASN"), ("DDD", 83, "AFR")]
WGT = (["AAA", "CCC", "DDD"], [0.5, 0.3, 0.261])


def oracle_compute(prepared_list):
    import oracle
    out = []
    for p in prepared_list:
        d = p.window_desc()
        mode = d.mode
        res = oracle.run_impute(mode, p.geno_m(), p.geno_u(), p.pop_off(), p.pop_wgt(), p.z1())
        C.memmove(d.out_z, res["z"].ctypes.data, 8 * p.U)
        C.memmove(d.out_info, res["info"].ctypes.data, 8 * p.U)
        out.append(p.finish())
    return out


def make_study(d):
    return panel.make_synthetic_study(str(d), POPS, n_snp=900, bp_lo=1_000_000, bp_hi=4_000_000, frac_measured=0.3, seed=23)


def test_assign_windows_is_balanced_and_deterministic():
    costs = [9.0, 1.0, 8.0, 2.0, 7.0, 3.0, 6.0, 4.0]
    own = farm.assign_windows(costs, 2)
    assert own == farm.assign_windows(list(costs), 2)
    load = [sum(c for c, o in zip(costs, own) if o == r) for r in (0, 1)]
    assert abs(load[0] - load[1]) <= 2.0
    assert farm.make_windows(1, 2_500_000, 1_000_000) == [(1, 1_000_000), (1_000_001, 2_000_000), (2_000_001, 2_500_000)]


def _check_shares(shares, mu, world):
    assert len(shares) == world
    cover = {}
    for sh in shares:
        for k, u0, u1 in sh:
            assert 0 <= u0 < u1 <= mu[k][1]
            cover.setdefault(k, []).append((u0, u1))
    assert sorted(cover) == list(range(len(mu)))                       # every window, and ...
    for k, segs in cover.items():
        segs.sort()
        assert segs[0][0] == 0 and segs[-1][1] == mu[k][1]             # ... all of its unmeasured SNPs, once
        assert all(a[1] == b[0] for a, b in zip(segs, segs[1:]))


@pytest.mark.parametrize("splitter", ["level", "balance"])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shares_with_window_cuts_cover_every_snp_once_and_are_balanced(splitter, world):
    """farm.level_windows / farm.balance_windows: pieces (window, u0, u1) tile every window's unmeasured SNPs exactly
    once, the plan is deterministic, and the modelled loads are closer than whole-window LPT leaves them."""
    rng = np.random.default_rng(5)
    mu = [(int(m), int(u)) for m, u in zip(rng.integers(150, 1250, 36), rng.integers(500, 2600, 36))]
    N = 32147
    fn = farm.level_windows if splitter == "level" else farm.balance_windows
    shares, loads = fn(mu, N, world)
    assert (shares, loads) == fn(list(mu), N, world)
    _check_shares(shares, mu, world)
    if world > 1:
        full = [b + u * r for (b, r), (_, u) in zip((farm.piece_cost(N, m, u) for m, u in mu), mu)]
        own = farm.assign_windows(full, world)
        chain = (lambda r: farm.CHAIN_STEP_COST * max(((mu[k][0] + 63) // 64 for k in range(len(mu)) if own[k] == r), default=0)) \
            if splitter == "level" else (lambda r: 0.0)
        lpt = [sum(c for c, o in zip(full, own) if o == r) + chain(r) for r in range(world)]
        assert max(loads) / (sum(loads) / world) <= (1.03 if splitter == "level" else 1.10)
        assert max(loads) <= max(lpt) * (1.0 if splitter == "level" else 1.10)     # contiguity costs: balance_windows repeats a B11 per boundary
    cuts = sum(1 for sh in shares for _, u0, _ in sh if u0 > 0)
    assert cuts <= 2 * world
    # degenerate inputs
    assert fn([], N, 4)[0] == [[], [], [], []]
    one = fn([(300, 900)], N, 4)[0]
    _check_shares(one, [(300, 900)], 4)


def test_farm_single_process_matches_per_window_calls(tmp_path):
    st = make_study(tmp_path)
    p = st["paths"]
    files = dict(input_file=p["gwas.txt"], reference_index_file=p["index.gz"], reference_data_file=p["data.gz"],
                 reference_pop_desc_file=p["desc.txt"])
    res = farm.impute_chromosome(api.KIND_DISTMIX, 22, 1_000_001, 4_000_000, 200_000, pop_wgt_df=WGT,
                                 compute=oracle_compute, **files)
    tab = res["table"]
    assert len(res["windows"]) == 3 and not res["skipped"]
    assert list(tab["bp"]) == sorted(tab["bp"])                   # window order = position order
    from oracle import feeder_py as fp
    want_bp = []
    for s, e in res["windows"]:
        w = fp.distmix(22, s, e, 200_000, WGT, p["gwas.txt"], p["index.gz"], p["data.gz"], p["desc.txt"])
        want_bp += w["bp"]
    assert list(tab["bp"]) == want_bp


WORKER = r"""
import os, sys, pickle
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
from gauss_amd import api, farm
from test_farm import oracle_compute, WGT
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size={world})
files = pickle.load(open({files!r}, "rb"))
res = farm.impute_chromosome(api.KIND_DISTMIX, 22, 1_000_001, 4_000_000, 200_000, pop_wgt_df=WGT,
                             compute=oracle_compute, window_size={window}, **files)
if dist.get_rank() == 0:
    pickle.dump(dict(bp=list(res["table"]["bp"]), z=list(res["table"]["z"]), owner=res["owner"],
                     skipped=res["skipped"]), open({out!r}, "wb"))
else:
    assert res is None
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,window", [(2, 500_000), (8, 250_000)])
def test_farm_ranks_gloo_equal_one_rank(tmp_path, world, window):
    """One process per rank over gloo, windows sharded with no data-path collective: the gathered table equals the one-rank
    table bit for bit.  World 2, and world 8 -- the node size of BASELINE.json configs[3] (eight real processes, twelve
    windows: the rendezvous, the plan every rank derives for itself and the gather at the size the driver's SCALE run uses;
    the GPU box admits at most 6 processes on its card, so the 8-rank case can only be rehearsed on the CPU)."""
    import pickle
    st = make_study(tmp_path)
    p = st["paths"]
    files = dict(input_file=p["gwas.txt"], reference_index_file=p["index.gz"], reference_data_file=p["data.gz"],
                 reference_pop_desc_file=p["desc.txt"])
    one = farm.impute_chromosome(api.KIND_DISTMIX, 22, 1_000_001, 4_000_000, 200_000, pop_wgt_df=WGT,
                                 compute=oracle_compute, window_size=window, **files)
    fpath, opath = str(tmp_path / "files.pkl"), str(tmp_path / "out.pkl")
    pickle.dump(files, open(fpath, "wb"))
    port = 29500 + (os.getpid() % 2000) + world
    script = str(tmp_path / "worker.py")
    open(script, "w").write(WORKER.format(root=ROOT, port=port, files=fpath, out=opath, world=world, window=window))
    procs = [subprocess.Popen([sys.executable, script, str(r)]) for r in range(world)]
    for pr in procs:
        assert pr.wait(timeout=300) == 0
    two = pickle.load(open(opath, "rb"))
    assert sorted(set(two["owner"])) == list(range(world))        # every rank really had windows
    assert two["bp"] == list(one["table"]["bp"])
    assert np.array_equal(np.array(two["z"]), one["table"]["z"].to_numpy())
    assert two["skipped"] == one["skipped"]


@pytest.mark.gpu
def test_gpu_farm_single_rank(ctx, tmp_path):
    st = make_study(tmp_path)
    p = st["paths"]
    files = dict(input_file=p["gwas.txt"], reference_index_file=p["index.gz"], reference_data_file=p["data.gz"],
                 reference_pop_desc_file=p["desc.txt"])
    res = farm.impute_chromosome(api.KIND_DISTMIX, 22, 1_000_001, 4_000_000, 200_000, pop_wgt_df=WGT,
                                 window_size=500_000, compute=lambda pl: farm.gpu_compute(pl, ctx), **files)
    want = farm.impute_chromosome(api.KIND_DISTMIX, 22, 1_000_001, 4_000_000, 200_000, pop_wgt_df=WGT,
                                  window_size=500_000, compute=oracle_compute, **files)
    a, b = res["table"], want["table"]
    assert list(a["rsid"]) == list(b["rsid"])
    assert np.max(np.abs(a["z"].to_numpy() - b["z"].to_numpy()) / np.maximum(1, np.abs(b["z"].to_numpy()))) <= 1e-8
    assert np.max(np.abs(a["info"].to_numpy() - b["info"].to_numpy())) <= 1e-8


@pytest.mark.gpu
def test_gpu_farm_packed_panel_resident_store(ctx, tmp_path):
    """A packed panel is uploaded once per rank (the slice its windows touch) and the windows gather their
    rows from HBM: same table as the BGZF text path, bit for bit; and the same again without residency."""
    st = make_study(tmp_path)
    p = st["paths"]
    gpk = str(tmp_path / "panel.gpk")
    assert api.pack_panel(p["index.gz"], p["data.gz"], p["desc.txt"], gpk) > 0
    common = dict(input_file=p["gwas.txt"], reference_index_file=p["index.gz"], reference_pop_desc_file=p["desc.txt"])
    run = lambda data, **kw: farm.impute_chromosome(
        api.KIND_DISTMIX, 22, 1_000_001, 4_000_000, 200_000, pop_wgt_df=WGT, window_size=500_000,
        compute=lambda pl: farm.gpu_compute(pl, ctx, **kw), reference_data_file=data, **common)["table"]
    text = run(p["data.gz"])
    for kw in (dict(resident=True), dict(resident=False)):
        got = run(gpk, **kw)
        assert list(got["rsid"]) == list(text["rsid"])
        assert np.array_equal(got["z"].to_numpy(), text["z"].to_numpy())
        assert np.array_equal(got["info"].to_numpy(), text["info"].to_numpy())


class _StubLib:
    """Stands in for libgauss_hip.so: records the device gauss_hip_init is asked for (no GPU needed)."""

    def __init__(self, n_devices):
        self.n_devices = n_devices
        self.init_devices = []

    def gauss_hip_device_count(self, out):
        out._obj.value = self.n_devices
        return 0

    def gauss_hip_init(self, device, out):
        self.init_devices.append(device)
        out._obj.value = 0xBEEF
        return 0

    def gauss_hip_destroy(self, handle):
        pass


@pytest.mark.parametrize("local_rank,n_dev,want", [(0, 8, 0), (3, 8, 3), (7, 8, 7), (9, 8, 1), (2, 1, 0)])
def test_farm_rank_binds_its_own_device(monkeypatch, local_rank, n_dev, want):
    """One process per GPU: the default context of rank r is created on device LOCAL_RANK % n_devices
    (round 1 bound every rank to device 0)."""
    from gauss_amd import _lib, hotpath
    stub = _StubLib(n_dev)
    monkeypatch.setattr(_lib, "load", lambda: stub)
    monkeypatch.setattr(hotpath, "_default_ctx", None)
    monkeypatch.setenv("LOCAL_RANK", str(local_rank))
    monkeypatch.delenv("GAUSS_SHARED_DEVICE", raising=False)
    monkeypatch.delenv("GAUSS_BENCH_SHARED_DEVICE", raising=False)
    ctx = hotpath.default_context()
    assert stub.init_devices == [want] and ctx.device == want
    monkeypatch.setattr(hotpath, "_default_ctx", None)
    monkeypatch.setenv("GAUSS_SHARED_DEVICE", "1")          # rehearsal: several ranks on one card
    assert hotpath.default_context().device == 0
    monkeypatch.setattr(hotpath, "_default_ctx", None)


@pytest.mark.gpu
def test_gpu_context_reports_its_device(ctx):
    from gauss_amd import hotpath
    assert ctx.lib.gauss_hip_device_of(ctx.handle) == ctx.device
    assert hotpath.device_count() >= 1
    assert hotpath.rank_device(local_rank=hotpath.device_count() + 2) == 2 % hotpath.device_count()


@pytest.mark.gpu
def test_gpu_native_chromosome_run_equals_the_python_farm(ctx, tmp_path):
    """gauss_host_impute_chromosome (native windows loop: LPT shard, batched pipeline against the resident panel,
    per-job events) must hand back, bit for bit, what the Python farm assembles window by window -- for one rank,
    for two ranks merged, for every batch count -- and report guarded windows instead of failing."""
    st = make_study(tmp_path)
    p = st["paths"]
    gpk = str(tmp_path / "panel.gpk")
    assert api.pack_panel(p["index.gz"], p["data.gz"], p["desc.txt"], gpk) > 0
    span = dict(chr=22, start_bp=1_000_001, end_bp=4_000_000, wing_size=200_000)
    want = farm.impute_chromosome(api.KIND_DISTMIX, pop_wgt_df=WGT, window_size=250_000, input_file=p["gwas.txt"],
                                  reference_index_file=p["index.gz"], reference_data_file=gpk, reference_pop_desc_file=p["desc.txt"],
                                  compute=lambda pl: farm.gpu_compute(pl, ctx), **span)
    wt = want["table"]
    kw = dict(kind=api.KIND_DISTMIX, pop_wgt_df=WGT, window_size=250_000, input_file=p["gwas.txt"], reference_data_file=gpk,
              reference_pop_desc_file=p["desc.txt"], ctx=ctx, **span)

    def same(res):
        f = res.frame()
        assert list(f.columns) == list(wt.columns)
        assert list(f["rsid"]) == list(wt["rsid"]) and list(f["bp"]) == list(wt["bp"])
        for c in ("z", "info", "pval", "af1mix"):
            assert np.array_equal(f[c].to_numpy(), wt[c].to_numpy()), c
        assert list(f["type"]) == list(wt["type"])

    for nb in (1, 2, 5):
        one = api.impute_chromosome(n_batches=nb, **kw)
        same(one)
        assert one.stats["n_windows"] == 12 and one.stats["n_failed"] == 0
        assert one.stats["n_skipped"] == len(want["skipped"])
        assert sorted(np.nonzero(one.windows[:, 3] == 1)[0]) == sorted(want["windows"].index(w) for w, _ in want["skipped"])
    assert one.stats["panel_bytes_uploaded"] == 0                     # resident since the first call
    parts = [api.impute_chromosome(rank=r, world=2, **kw) for r in range(2)]
    assert all(q.stats["n_windows_mine"] > 0 for q in parts)
    assert not (set(parts[0].columns["window"]) & set(parts[1].columns["window"]))
    same(api.ChromResult.merge(parts))
    api.panel_evict(ctx=ctx)
    again = api.impute_chromosome(**kw)
    assert again.stats["panel_bytes_uploaded"] > 0
    same(again)
    api.panel_evict(ctx=ctx)


@pytest.mark.gpu
def test_gpu_native_chromosome_first_use_of_a_panel(ctx, tmp_path, monkeypatch):
    """First use of a panel on a context: its rows travel to HBM WHILE the batches compute (background upload, a batch waits for
    the rows it names, gauss_store_wait), in six graded batches when the rank holds twenty-eight windows or more; a later call finds
    the panel resident and runs four.  Either way, and for every form of the upload (DMA copies, the copy kernel, in one go
    before the first batch), the table is the same, bit for bit."""
    st = make_study(tmp_path)
    p = st["paths"]
    gpk = str(tmp_path / "panel.gpk")
    assert api.pack_panel(p["index.gz"], p["data.gz"], p["desc.txt"], gpk) > 0
    kw = dict(kind=api.KIND_DISTMIX, pop_wgt_df=WGT, window_size=100_000, input_file=p["gwas.txt"], reference_data_file=gpk,
              reference_pop_desc_file=p["desc.txt"], ctx=ctx, chr=22, start_bp=1_000_001, end_bp=4_000_000, wing_size=200_000)
    api.panel_evict(ctx=ctx)
    first = api.impute_chromosome(**kw)
    assert first.stats["n_windows_mine"] == 30 and first.stats["n_failed"] == 0
    assert first.stats["panel_bytes_uploaded"] > 0 and first.stats["n_batches"] == 6
    warm = api.impute_chromosome(**kw)
    assert warm.stats["panel_bytes_uploaded"] == 0 and warm.stats["n_batches"] == 4

    def same(a, b):
        assert list(a.columns) == list(b.columns)
        for c in a.columns:
            x, y = a.columns[c], b.columns[c]
            assert np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y), c

    same(first, warm)
    assert first.stats["imputed"] == warm.stats["imputed"] > 0
    for env in (dict(GAUSS_UPLOAD_BY_KERNEL="1"), dict(GAUSS_CHROM_ASYNC_UPLOAD="0")):
        api.panel_evict(ctx=ctx)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        again = api.impute_chromosome(**kw)
        for k in env:
            monkeypatch.delenv(k)
        assert again.stats["panel_bytes_uploaded"] > 0, env
        assert again.stats["n_batches"] == (4 if "GAUSS_CHROM_ASYNC_UPLOAD" in env else 6), env      # in one go: nothing to grade the batches for
        assert again.stats["n_merged_giveups"] == 0
        same(again, warm)
    api.panel_evict(ctx=ctx)


@pytest.mark.gpu
def test_gpu_native_chromosome_failed_first_call_leaves_no_half_uploaded_panel(ctx, tmp_path):
    """the chromosome driver starts the panel's upload in the background BEFORE it parses the study file.  A first call
    that fails right after (a study file that does not exist) used to return with the upload still running and the panel listed
    as resident: an LD call on that panel a moment later could read rows that had not landed.  Every exit of the driver now waits
    for the upload it started (or drops the store if the upload failed), and whoever asks for a resident panel's pointer waits
    for its rows: computeLD right after the failed call returns the matrix of a fresh context."""
    st = make_study(tmp_path)
    p = st["paths"]
    gpk = str(tmp_path / "panel.gpk")
    assert api.pack_panel(p["index.gz"], p["data.gz"], p["desc.txt"], gpk) > 0
    ld_kw = dict(chr=22, start_bp=1_000_001, end_bp=1_600_000, pop_wgt_df=WGT, input_file=p["gwas.txt"], reference_index_file="(packed)",
                 reference_data_file=gpk, reference_pop_desc_file=p["desc.txt"])
    api.panel_evict(ctx=ctx)
    want = api.computeLD(ctx=ctx, **ld_kw)
    api.panel_evict(ctx=ctx)
    with pytest.raises(Exception):
        api.impute_chromosome(kind=api.KIND_DISTMIX, pop_wgt_df=WGT, window_size=125_000, input_file=str(tmp_path / "no_such_study.txt"),
                              reference_data_file=gpk, reference_pop_desc_file=p["desc.txt"], ctx=ctx, chr=22, start_bp=1_000_001,
                              end_bp=4_000_000, wing_size=200_000)
    got = api.computeLD(ctx=ctx, **ld_kw)            # the panel is resident (its upload was started): every row must have landed
    assert np.array_equal(got["cormat"], want["cormat"])
    assert api.impute_chromosome(kind=api.KIND_DISTMIX, pop_wgt_df=WGT, window_size=125_000, input_file=p["gwas.txt"], reference_data_file=gpk,
                                 reference_pop_desc_file=p["desc.txt"], ctx=ctx, chr=22, start_bp=1_000_001, end_bp=4_000_000,
                                 wing_size=200_000).stats["panel_bytes_uploaded"] == 0
    api.panel_evict(ctx=ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_gpu_native_genome_driver_equals_the_chromosome_calls(ctx, tmp_path, world):
    """gauss_host_impute_genome: the loop over chromosomes with two calls in flight on one context (host threads of the library:
    one call's data layer and tables run under the other call's GPU work).  Every chromosome's table and window matrix are what
    gauss_host_impute_chromosome returns for it alone, bit for bit, for one rank and for the ranks of a two-rank run; a
    chromosome whose arguments are bad fails alone."""
    st = make_study(tmp_path)
    p = st["paths"]
    gpk = str(tmp_path / "panel.gpk")
    assert api.pack_panel(p["index.gz"], p["data.gz"], p["desc.txt"], gpk) > 0
    kw = dict(kind=api.KIND_DISTMIX, pop_wgt_df=WGT, window_size=250_000, input_file=p["gwas.txt"], reference_data_file=gpk,
              reference_pop_desc_file=p["desc.txt"], ctx=ctx, wing_size=200_000)
    chroms = [(22, 1_000_001, 2_500_000), (22, 2_500_001, 4_000_000), (22, 1_000_001, 4_000_000), (22, 1_750_001, 3_250_000),
              (21, 1_000_001, 2_000_000), (22, 1_000_001, 1_500_000)]
    for rank in range(world):
        want = [api.impute_chromosome(chr=c, start_bp=a, end_bp=b, rank=rank, world=world, **kw) for c, a, b in chroms]
        for depth in (1, 2, 3):
            got = api.impute_genome(chromosomes=chroms, rank=rank, world=world, depth=depth, **kw)
            assert len(got) == len(want)
            for g, w in zip(got, want):
                assert list(g.columns) == list(w.columns)
                for c in g.columns:
                    x, y = g.columns[c], w.columns[c]
                    assert np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y), (depth, c)
                assert np.array_equal(g.windows, w.windows)
                assert g.stats["imputed"] == w.stats["imputed"] and g.stats["n_failed"] == 0
    bad = chroms[:2] + [(22, 3_000_000, 2_000_000)] + chroms[2:4]
    got, err = api.impute_genome(chromosomes=bad, depth=2, raise_on_error=False, **kw)
    assert got[2] is None and "call 2" in err and all(g is not None for k, g in enumerate(got) if k != 2)
    assert np.array_equal(got[3].columns["z"], want[2].columns["z"]) if world == 1 else True
    api.panel_evict(ctx=ctx)


@pytest.mark.gpu
def test_gpu_native_chromosome_isolates_a_failing_window(ctx, tmp_path):
    """A window whose data layer fails (here: the GWAS file lists one SNP with both allele orders inside that
    window, the reference's "duplicates" error, gauss.cpp:388-391) is reported, the other windows still come back."""
    st = make_study(tmp_path)
    p = st["paths"]
    gpk = str(tmp_path / "panel.gpk")
    api.pack_panel(p["index.gz"], p["data.gz"], p["desc.txt"], gpk)
    lines = open(p["gwas.txt"]).read().splitlines()
    rows = [l.split() for l in lines[1:]]
    victim = next(r for r in rows if 1_600_000 < int(r[2]) < 1_900_000)
    bad = tmp_path / "gwas_dup.txt"
    bad.write_text("\n".join(lines + [" ".join([victim[0], victim[1], victim[2], victim[4], victim[3], victim[5]])]) + "\n")
    kw = dict(kind=api.KIND_DISTMIX, pop_wgt_df=WGT, window_size=500_000, chr=22, start_bp=1_000_001, end_bp=4_000_000,
              wing_size=100_000, reference_data_file=gpk, reference_pop_desc_file=p["desc.txt"], ctx=ctx)
    good = api.impute_chromosome(input_file=p["gwas.txt"], **kw)
    res = api.impute_chromosome(input_file=str(bad), **kw)
    assert res.stats["n_failed"] >= 1 and any("duplicate" in m for m in res.messages)
    failed = set(np.nonzero(res.windows[:, 3] == 2)[0])
    keep = ~np.isin(good.columns["window"], list(failed))
    assert np.array_equal(res.columns["z"], good.columns["z"][keep])
    assert np.array_equal(res.columns["rsid"], good.columns["rsid"][keep])
    api.panel_evict(ctx=ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("kind_name", ["DIST", "QCAT", "QCATMIX"])
def test_gpu_native_chromosome_run_other_kinds_equal_per_window_calls(ctx, tmp_path, kind_name):
    """dist / qcat / qcatmix through the native chromosome driver against the reference-style entry point called window
    by window on the same packed panel: same rows, same bits."""
    st = make_study(tmp_path)
    p = st["paths"]
    gpk = str(tmp_path / "panel.gpk")
    api.pack_panel(p["index.gz"], p["data.gz"], p["desc.txt"], gpk)
    kind = getattr(api, "KIND_" + kind_name)
    mix = kind_name.endswith("MIX")
    sel = dict(pop_wgt_df=WGT) if mix else dict(study_pop="EUR")
    res = api.impute_chromosome(kind, 22, 1_000_001, 4_000_000, 200_000, input_file=p["gwas.txt"], reference_data_file=gpk,
                                reference_pop_desc_file=p["desc.txt"], window_size=750_000, n_batches=2, ctx=ctx, **sel)
    assert res.stats["n_failed"] == 0 and res.stats["n_windows"] == 4
    fn = {"DIST": api.dist, "QCAT": api.qcat, "QCATMIX": api.qcatmix}[kind_name]
    frames = []
    for k, (s, e, owner, status, m, u) in enumerate(res.windows):
        if status != 0:
            continue
        a = (22, int(s), int(e), 200_000, WGT if mix else "EUR", p["gwas.txt"], p["index.gz"], gpk, p["desc.txt"])
        frames.append(fn(*a, ctx=ctx))
    import pandas as pd
    want = pd.concat(frames, ignore_index=True)
    got = res.frame()
    assert list(got.columns) == list(want.columns) and len(got) == len(want)
    for c in want.columns:
        if want[c].dtype.kind == "f":
            assert np.array_equal(got[c].to_numpy(), want[c].to_numpy(), equal_nan=True), c
        else:
            assert list(got[c]) == list(want[c]), c
    api.panel_evict(ctx=ctx)


@pytest.mark.gpu
def test_gpu_native_chromosome_window_cache_returns_the_same_table(ctx, tmp_path, monkeypatch):
    """The window cache (host_chrom.cpp: a built window is kept across calls, keyed by panel, study file and arguments): a repeat
    call -- every window a cache hit, a COPY handed out -- returns the table of the first call bit for bit, and that table equals the
    one built with the cache off; another kind, other weights or another wing on the same files miss (the key holds them) and equal
    their own uncached results; a study file rewritten in place (same path, other content) is a different study."""
    st = make_study(tmp_path)
    p = st["paths"]
    gpk = str(tmp_path / "panel.gpk")
    api.pack_panel(p["index.gz"], p["data.gz"], p["desc.txt"], gpk)
    base = dict(chr=22, start_bp=1_000_001, end_bp=4_000_000, input_file=p["gwas.txt"], reference_data_file=gpk,
                reference_pop_desc_file=p["desc.txt"], window_size=500_000, ctx=ctx)

    def same(a, b, owners=True):
        assert list(a.columns) == list(b.columns)
        for k in a.columns:
            assert np.array_equal(a.columns[k], b.columns[k], equal_nan=(a.columns[k].dtype.kind == "f")), k
        cols = [0, 1, 2, 3, 4, 5] if owners else [0, 1, 3, 4, 5]
        assert np.array_equal(a.windows[:, cols], b.windows[:, cols])

    variants = [dict(kind=api.KIND_DISTMIX, pop_wgt_df=WGT, wing_size=200_000),
                dict(kind=api.KIND_DISTMIX, pop_wgt_df=(WGT[0], [0.2, 0.6, 0.261]), wing_size=200_000),
                dict(kind=api.KIND_DISTMIX, pop_wgt_df=WGT, wing_size=100_000),
                dict(kind=api.KIND_QCATMIX, pop_wgt_df=WGT, wing_size=200_000),
                dict(kind=api.KIND_DIST, study_pop="EUR", wing_size=200_000)]
    monkeypatch.setenv("GAUSS_WINDOW_CACHE", "0")
    plain = [api.impute_chromosome(**base, **v) for v in variants]
    monkeypatch.delenv("GAUSS_WINDOW_CACHE")
    for v, want in zip(variants, plain):
        first = api.impute_chromosome(**base, **v)              # builds and stores
        again = api.impute_chromosome(**base, **v)              # every window a hit
        same(first, want)
        same(again, want)
    # two ranks' shares come out of the same cache entries
    halves = [api.impute_chromosome(**base, **variants[0], rank=r, world=2) for r in (0, 1)]
    same(api.ChromResult.merge(halves), plain[0], owners=False)
    # the same path with another study behind it
    lines = open(p["gwas.txt"]).read().splitlines()
    flipped = [lines[0]] + [" ".join(t[:5] + [repr(-float(t[5]))]) for t in (l.split() for l in lines[1:])]
    open(p["gwas.txt"], "w").write("\n".join(flipped) + "\n")
    neg = api.impute_chromosome(**base, **variants[0])
    meas = plain[0].columns["type"] == 1
    assert np.array_equal(neg.columns["z"][meas], -plain[0].columns["z"][meas])
    api.panel_evict(ctx=ctx)


def test_bench_pieces_are_put_together_per_window():
    """bench.py's shard check (pure Python): pieces (window, u0, u1) from several ranks must tile every window and
    reproduce the one-rank z / info bit for bit; a gap, an overlap or a changed bit is reported."""
    sys.path.insert(0, ROOT)
    import bench
    from gauss_amd import workload
    rng = np.random.default_rng(1)
    wins = [(1 + 1_000_000 * k, np.arange(5 + k), np.arange(100 * k, 100 * k + 30 + 7 * k)) for k in range(4)]
    ref = [dict(z=rng.standard_normal(len(w[2])), info=rng.random(len(w[2]))) for w in wins]
    shares = [[(0, 0, 30), (2, 0, 20)], [(1, 0, 37), (2, 20, 44)], [(3, 0, 51)]]
    assert [len(w[2]) for w in workload.pieces_of(wins, shares[0])] == [30, 20]
    assert np.array_equal(workload.pieces_of(wins, shares[1])[1][2], wins[2][2][20:44])
    parts = [{pc: (ref[pc[0]]["z"][pc[1]:pc[2]].copy(), ref[pc[0]]["info"][pc[1]:pc[2]].copy()) for pc in sh} for sh in shares]
    assert bench.pieces_equal_whole(parts, ref, wins)
    bad = [dict(p) for p in parts]
    z, info = bad[1][(2, 20, 44)]
    z = z.copy(); z[3] = np.nextafter(z[3], 1.0)
    bad[1][(2, 20, 44)] = (z, info)
    assert not bench.pieces_equal_whole(bad, ref, wins)                 # one bit off
    gap = [dict(p) for p in parts]
    del gap[1][(2, 20, 44)]
    assert not bench.pieces_equal_whole(gap, ref, wins)                 # window 2 is not covered
    lap = [dict(p) for p in parts]
    lap[2][(2, 10, 44)] = (ref[2]["z"][10:44], ref[2]["info"][10:44])
    assert not bench.pieces_equal_whole(lap, ref, wins)                 # pieces overlap


@pytest.mark.gpu
def test_gpu_native_chromosome_accepts_the_text_panel(ctx, tmp_path, monkeypatch):
    """The chromosome driver takes the reference's own panel format (BGZF index + data, gauss.cpp:293-399, 720-785):
    the packed form is made on first use in the panel cache, found again afterwards, and the table equals the one from
    a panel packed by hand; GAUSS_AUTO_PACK=0 restores the refusal."""
    st = make_study(tmp_path)
    p = st["paths"]
    monkeypatch.setenv("GAUSS_PANEL_CACHE", str(tmp_path / "cache"))
    monkeypatch.delenv("GAUSS_AUTO_PACK", raising=False)
    gpk = str(tmp_path / "panel.gpk")
    api.pack_panel(p["index.gz"], p["data.gz"], p["desc.txt"], gpk)
    kw = dict(kind=api.KIND_DISTMIX, pop_wgt_df=WGT, window_size=500_000, chr=22, start_bp=1_000_001, end_bp=4_000_000,
              wing_size=200_000, input_file=p["gwas.txt"], reference_pop_desc_file=p["desc.txt"], ctx=ctx)
    want = api.impute_chromosome(reference_data_file=gpk, **kw)
    assert api.panel_cache(p["index.gz"], p["data.gz"], p["desc.txt"], create=False) == (None, 0)
    got = api.impute_chromosome(reference_index_file=p["index.gz"], reference_data_file=p["data.gz"], **kw)
    cached, made_now = api.panel_cache(p["index.gz"], p["data.gz"], p["desc.txt"], create=False)
    assert cached is not None and made_now == 0
    again = api.impute_chromosome(reference_index_file=p["index.gz"], reference_data_file=p["data.gz"], **kw)
    assert again.stats["panel_bytes_uploaded"] == 0              # the cached panel is resident since the first call
    for r in (got, again):
        assert list(r.columns) == list(want.columns)
        for c in want.columns:
            assert np.array_equal(r.columns[c], want.columns[c]), c
    monkeypatch.setenv("GAUSS_AUTO_PACK", "0")
    with pytest.raises(Exception) as ei:
        api.impute_chromosome(reference_index_file=p["index.gz"], reference_data_file=p["data.gz"], **kw)
    assert "packed panel" in str(ei.value)
    api.panel_evict(ctx=ctx)


def test_native_planner_cost_equals_the_python_planner():
    """gauss_host_impute_chromosome balances its ranks on gauss_host_plan_cost; bench.py's strong-scaling shares and the
    Python farm on farm.piece_cost: the same model (flops the Gram kernel issues: 128-row tiles, 32 / 16 granular edges, B11's
    tile triangle; + 8 % on B21's share), two statements -- they must agree on every shape, tile edges included."""
    import ctypes as C
    h = api.load_host()
    h.gauss_host_plan_cost.restype = C.c_double
    h.gauss_host_plan_cost.argtypes = [C.c_int, C.c_int]
    N = 32147
    for m in (1, 15, 16, 17, 31, 33, 64, 65, 100, 127, 128, 129, 156, 200, 300, 423, 594, 742, 941, 1213, 2500, 4096):
        for u in (0, 1, 64, 513, 2407):
            b, r = farm.piece_cost(N, m, u)
            want = (b + u * r) / N
            got = h.gauss_host_plan_cost(m, u)
            assert abs(got - want) <= 1e-9 * max(1.0, want), (m, u, got, want)


def test_planner_adjacency_term_collects_neighbours_and_keeps_the_tiling():
    """farm.level_windows with the measured SNPs consecutive windows share: neighbours on one rank are priced at what the job
    builder saves for them (farm.adjacency_saving), the second phase of the local search collects them, and the shares still
    tile every window exactly; without the list nothing changes."""
    rng = np.random.default_rng(5)
    N = 32147
    mu = [(int(rng.integers(150, 1200)), int(rng.integers(500, 2600))) for _ in range(36)]
    shared = [min(mu[k][0], mu[k + 1][0]) // 2 for k in range(35)]
    plain, _ = farm.level_windows(mu, N, 8)
    adj, loads = farm.level_windows(mu, N, 8, shared=shared)
    for shares in (plain, adj):
        _check_shares(shares, mu, 8)
    pairs = lambda shares: sum(1 for s in shares for k, _, _ in s if any(q[0] == k + 1 for q in s))
    assert pairs(adj) > pairs(plain)
    assert max(loads) / (sum(loads) / len(loads)) < 1.02
    assert farm.adjacency_saving(N, 700, 700, 0) == 0.0 and farm.adjacency_saving(N, 700, 700, 350) > 0.0
    assert farm.adjacency_saving(N, 700, 640, 1) == 0.0             # one common SNP: joining at row 699 would cost 640 rows a sixth tile, the builder declines
