"""jepeg() / jepegmix() split over ranks (BASELINE.json configs[4]): the gene plan, the N > 1 path under gloo on
the CPU (world 2, real processes) and, -m gpu, the native per-rank call and the whole-call deal of the genome driver.

On the CPU the CorG blocks come from the oracle (the farm's `compute` hook: tests may use the oracle as the checker); what is under
test is the plan every rank derives for itself, the per-range tails and the gather -- the part that differs between 1 and N ranks.
The reference for the split is the independence of genes: jepeg.cpp:114-131, jepegmix.cpp:119-140, gauss.cpp:1383-1439."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

from gauss_amd import api, farm, panel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POPS = [("AAA", 120, "EUR"), ("BBB", 95, "EUR"), ("CCC", 110, "ASN"), ("DDD", 83, "AFR")]
WGT = (["AAA", "CCC", "DDD"], [0.5, 0.3, 0.261])
LAMBDA = 0.1


def oracle_gene_ld(kind):
    """compute hook: CorG of genes [g0, g1) by the CPU oracle (gene.cpp:306-315 pooled / 576-586 weighted), diagonal 1 + lambda."""
    def compute(pr, g0, g1):
        import oracle
        go, G, off, w = pr.gene_off(), pr.geno_m(), pr.pop_off(), pr.pop_wgt()
        out = []
        for g in range(g0, g1):
            rows = np.ascontiguousarray(G[go[g]:go[g + 1]])
            if kind == api.KIND_JEPEGMIX:
                c = oracle.compute_ld(rows, off, w)
                np.fill_diagonal(c, 1.0 + LAMBDA)
            else:
                c = oracle.ld_pooled(rows, off, 1.0 + LAMBDA)
            out.append(c)
        return out
    return compute


def make_study(d, n_genes=37, seed=31):
    return panel.make_synthetic_study(str(d), POPS, n_snp=800, bp_lo=1_000_000, bp_hi=3_000_000, frac_measured=0.4, n_genes=n_genes, seed=seed)


def files_of(st):
    p = st["paths"]
    return dict(input_file=p["gwas.txt"], annotation_file=p["annot.txt"], reference_index_file=p["index.gz"],
                reference_data_file=p["data.gz"], reference_pop_desc_file=p["desc.txt"])


def who(kind):
    return dict(pop_wgt_df=WGT) if kind == api.KIND_JEPEGMIX else dict(study_pop="EUR")


def same_table(a, b):
    assert list(a.columns) == list(b.columns) and len(a) == len(b)
    for c in a.columns:
        x, y = a[c].to_numpy(), b[c].to_numpy()
        if x.dtype.kind == "f":
            assert np.array_equal(x.view(np.uint64), y.view(np.uint64)), c           # bit for bit
        else:
            assert list(x) == list(y), c


@pytest.mark.parametrize("world", [1, 2, 3, 8, 64])
def test_gene_plan_is_contiguous_balanced_and_the_same_everywhere(tmp_path, world):
    st = make_study(tmp_path)
    pr = api.Prepared(api.KIND_JEPEGMIX, **files_of(st), **who(api.KIND_JEPEGMIX))
    first = pr.jepeg_plan(world)
    assert first == pr.jepeg_plan(world)                                     # deterministic
    go = pr.gene_off()
    ng = len(go) - 1
    assert ng >= 30 and first[0] == 0 and first[-1] == ng and len(first) == world + 1
    assert all(a <= b for a, b in zip(first, first[1:]))                     # contiguous ranges that tile [0, genes)
    n = np.diff(go).astype(np.int64)
    cost = n * (n + 1) + 64
    load = [int(cost[a:b].sum()) for a, b in zip(first, first[1:])]
    assert sum(load) == int(cost.sum())
    if world <= 8:
        assert max(load) <= cost.sum() / world + cost.max()                  # no rank more than one gene over its share
    pr.close()


@pytest.mark.parametrize("kind", [api.KIND_JEPEG, api.KIND_JEPEGMIX])
def test_ranges_concatenate_to_the_one_rank_table_in_process(tmp_path, kind):
    """Every rank's range through the Python form of the split, one after the other in this process: the concatenation equals the
    world-1 table bit for bit, and that table equals the Python restatement of the reference driver (oracle/feeder_py.py)."""
    import pandas as pd
    from oracle import feeder_py as fp
    st = make_study(tmp_path)
    f = files_of(st)
    pr = api.Prepared(kind, **f, **who(kind))
    comp = oracle_gene_ld(kind)
    ng = pr.jepeg_plan(1)[-1]
    one = pr.jepeg_finish(0, ng, comp(pr, 0, ng))
    for world in (2, 5, 8):
        first = pr.jepeg_plan(world)
        parts = [pr.jepeg_finish(a, b, comp(pr, a, b)) for a, b in zip(first, first[1:])]
        assert [len(t) for t in parts] == [b - a for a, b in zip(first, first[1:])]
        same_table(pd.concat(parts, ignore_index=True), one)
    pr.close()
    args = (f["input_file"], f["annotation_file"], f["reference_index_file"], f["reference_data_file"], f["reference_pop_desc_file"])
    want = fp.jepegmix(WGT, *args) if kind == api.KIND_JEPEGMIX else fp.jepeg("EUR", *args)
    assert len(one) == len(want) and len(one) == ng
    for (_, r), w in zip(one.iterrows(), want):
        assert r["df"] == w["df"] and r["num_snp"] == w["num_snp"]
        if w["df"] > 0:
            assert abs(r["chisq"] - w["chisq"]) <= 1e-9 * max(1.0, abs(w["chisq"]))


WORKER = r"""
import os, sys, pickle
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
from gauss_amd import api, farm
from test_farm_jepeg import oracle_gene_ld, who
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size={world})
files = pickle.load(open({files!r}, "rb"))
res = farm.jepeg({kind}, compute=oracle_gene_ld({kind}), **files, **who({kind}))
if dist.get_rank() == 0:
    pickle.dump(res, open({out!r}, "wb"))
else:
    assert res is None
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,kind", [(2, api.KIND_JEPEGMIX), (2, api.KIND_JEPEG), (4, api.KIND_JEPEGMIX)])
def test_jepeg_ranks_gloo_equal_one_rank(tmp_path, world, kind):
    """One process per rank over gloo, the genes of ONE call split with no data-path collective: the gathered table equals the
    one-rank table row for row, bit for bit."""
    st = make_study(tmp_path)
    f = files_of(st)
    one = farm.jepeg(kind, compute=oracle_gene_ld(kind), **f, **who(kind))
    assert one["ranges"] == [(0, len(one["table"]))]
    fpath, opath = str(tmp_path / "files.pkl"), str(tmp_path / "out.pkl")
    pickle.dump(f, open(fpath, "wb"))
    port = 31500 + (os.getpid() % 2000) + 7 * world + kind
    script = str(tmp_path / "worker.py")
    open(script, "w").write(WORKER.format(root=ROOT, port=port, files=fpath, out=opath, world=world, kind=kind))
    procs = [subprocess.Popen([sys.executable, script, str(r)]) for r in range(world)]
    for pr in procs:
        assert pr.wait(timeout=300) == 0
    got = pickle.load(open(opath, "rb"))
    assert len(got["ranges"]) == world and all(b > a for a, b in got["ranges"])        # every rank really had genes
    same_table(got["table"], one["table"])


# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("kind", [api.KIND_JEPEG, api.KIND_JEPEGMIX])
@pytest.mark.parametrize("packed", [False, True])
def test_gpu_native_rank_tables_concatenate_to_the_one_rank_table(ctx, tmp_path, kind, packed):
    """gauss_host_jepeg_rank on the GPU: ranks 0 and 1 of 2 (and 0 .. 7 of 8) concatenate to gauss_host_jepeg(mix)'s table bit for
    bit, on the text panel and on the packed panel."""
    import pandas as pd
    st = make_study(tmp_path)
    f = files_of(st)
    if packed:
        gpk = str(tmp_path / "p.gpk")
        api.pack_panel(f["reference_index_file"], f["reference_data_file"], f["reference_pop_desc_file"], gpk)
        f = dict(f, reference_data_file=gpk)
    args = (f["input_file"], f["annotation_file"], f["reference_index_file"], f["reference_data_file"], f["reference_pop_desc_file"])
    one = api.jepegmix(WGT, *args, ctx=ctx) if kind == api.KIND_JEPEGMIX else api.jepeg("EUR", *args, ctx=ctx)
    for world in (2, 8):
        parts, ranges = [], []
        for r in range(world):
            t, (g0, g1, ng) = api.jepeg_rank(kind, *args, rank=r, world=world, ctx=ctx, **who(kind))
            assert ng == len(one) and len(t) == g1 - g0
            parts.append(t)
            ranges.append((g0, g1))
        assert ranges[0][0] == 0 and ranges[-1][1] == len(one) and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        same_table(pd.concat(parts, ignore_index=True), one)
    # the Python form of the split with the oracle's CorG agrees with the GPU's table to rounding
    pr = api.Prepared(kind, **f, **who(kind))
    ref = pr.jepeg_finish(0, len(one), oracle_gene_ld(kind)(pr, 0, len(one)))
    pr.close()
    assert list(ref["geneid"]) == list(one["geneid"]) and list(ref["df"]) == list(one["df"])
    ok = one["df"].to_numpy() > 0
    assert np.allclose(ref["chisq"].to_numpy()[ok], one["chisq"].to_numpy()[ok], rtol=1e-8, atol=0)
    with pytest.raises(api.GaussError):
        api.jepeg_rank(kind, *args, rank=2, world=2, ctx=ctx, **who(kind))


@pytest.mark.gpu
def test_gpu_genome_driver_deals_whole_calls(ctx, tmp_path):
    """gauss_host_jepeg_genome: five calls (five studies with annotations of different sizes) dealt whole to 1, 2 and 3 ranks; every
    call is run by exactly one rank, the owner's table equals the stand-alone call's, the deal puts the longest annotation first and
    is the same on every rank; a call whose file is missing fails alone."""
    sts = []
    for k, ng in enumerate((12, 40, 25, 8, 31)):
        d = tmp_path / f"s{k}"
        d.mkdir()
        sts.append(files_of(make_study(d, n_genes=ng, seed=40 + k)))
    desc = sts[0]["reference_pop_desc_file"]
    calls = [(f["input_file"], f["annotation_file"], f["reference_index_file"], f["reference_data_file"]) for f in sts]
    alone = [api.jepegmix(WGT, c[0], c[1], c[2], c[3], desc, ctx=ctx) for c in calls]
    for world in (1, 2, 3):
        seen = [None] * len(calls)
        owners = None
        for r in range(world):
            tabs, owner = api.jepeg_genome(api.KIND_JEPEGMIX, calls, desc, pop_wgt_df=WGT, rank=r, world=world, ctx=ctx)
            assert owners is None or owners == owner
            owners = owner
            for c, t in enumerate(tabs):
                assert (t is not None) == (owner[c] == r)
                if t is not None:
                    assert seen[c] is None
                    seen[c] = t
        for t, a in zip(seen, alone):
            same_table(t, a)
        if world > 1:
            sizes = [os.path.getsize(c[1]) for c in calls]
            assert owners[int(np.argmax(sizes))] == 0 and len(set(owners)) == world
    bad = list(calls)
    bad[2] = (calls[2][0], str(tmp_path / "missing.txt"), calls[2][2], calls[2][3])
    tabs, owner = api.jepeg_genome(api.KIND_JEPEGMIX, bad, desc, pop_wgt_df=WGT, ctx=ctx, raise_on_error=False)
    assert tabs[2] is None and all(t is not None for k, t in enumerate(tabs) if k != 2)
    with pytest.raises(api.GaussError):
        api.jepeg_genome(api.KIND_JEPEGMIX, bad, desc, pop_wgt_df=WGT, ctx=ctx)
