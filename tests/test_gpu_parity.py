import os
"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): integer Gram counts bit-exact; LD values, z and info within
1e-5 relative of the fp64 reference path.  The tolerances asserted here are far tighter because
the design keeps the integer part exact and does every floating-point tail in fp64.
"""
import numpy as np
import pytest

import oracle
from gauss_amd import hotpath
from gauss_amd import panel as panel_mod
from helpers import relerr, small_panel, split_window

pytestmark = pytest.mark.gpu

LD_TOL = 1e-12      # LD entries: reference operation order in fp64 -> expected bit-exact
Z_TOL = 1e-8        # z / info: Cholesky vs the reference's LU inverse, both fp64


def test_gram_counts_bit_exact(ctx):
    p = small_panel(n_snp=150, scale=0.03)
    G = p["G"]
    got = hotpath.gram_counts(G, ctx=ctx)
    want = oracle.gram_counts(G)
    assert got.dtype == np.int64
    assert np.array_equal(got, want)


def test_gram_counts_ascii_and_ragged_stride(ctx):
    p = small_panel(n_snp=40, scale=0.01, n_pops=5)
    G = p["G"]
    S, N = G.shape
    buf = np.zeros((S, N + 13), dtype=np.uint8)       # row stride != N, N not a multiple of 64
    buf[:, :N] = G + ord("0")                           # ASCII digits as the reference stores them
    got = hotpath.gram_counts(buf[:, :N], ctx=ctx)
    assert np.array_equal(got, oracle.gram_counts(G))


def test_compute_ld_weighted_matches_oracle(ctx):
    p = small_panel(n_snp=140, scale=0.02)
    got = hotpath.ld_matrix(p["G"], p["off"], p["w"], mode=hotpath.MODE_WEIGHTED, diag=1.0, ctx=ctx)
    want = oracle.compute_ld(p["G"], p["off"], p["w"])
    assert np.allclose(got, got.T, rtol=0, atol=0)
    assert np.all(np.diag(got) == 1.0)
    assert np.max(np.abs(got - want)) <= LD_TOL
    # the fp64 tails follow the reference's operation order: expect identical bits
    assert np.mean(got == want) > 0.999


def test_ld_pooled_matches_oracle(ctx):
    p = small_panel(n_snp=90, scale=0.02, n_pops=6)
    got = hotpath.ld_matrix(p["G"], p["off"], None, mode=hotpath.MODE_POOLED, diag=1.1, ctx=ctx)
    want = oracle.ld_pooled(p["G"], p["off"], 1.1)
    assert np.max(np.abs(got - want)) <= LD_TOL
    assert np.mean(got == want) > 0.999


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("shape", [(11, 11), (60, 90), (150, 170)])
def test_impute_window_matches_oracle(ctx, mode, shape):
    M, U = shape
    p = small_panel(n_snp=M + U + 30, scale=0.02, seed=11 + M)
    G = p["G"][: M + U]
    gm, gu, z1 = split_window(dict(G=G), M)
    got = hotpath.impute_window(mode, gm, gu, p["off"], p["w"], z1, want_mats=True, ctx=ctx)
    want = oracle.run_impute(mode, gm, gu, p["off"], p["w"], z1, want_mats=True)
    assert want["mpd"] == 0 and got["status"] == 0
    assert np.max(np.abs(got["b11"] - want["b11"])) <= LD_TOL
    assert np.max(np.abs(got["b21"] - want["b21"])) <= LD_TOL
    assert relerr(got["info"], want["info"]) <= Z_TOL
    assert np.max(np.abs(got["z"] - want["z"]) / np.maximum(1.0, np.abs(want["z"]))) <= Z_TOL


def test_segmented_population_over_2048_samples(ctx):
    # one population larger than SEG_MAX forces several K segments per population
    from gauss_amd import synth
    pops = [("AAA", 2500, "X"), ("BBB", 70, "X"), ("CCC", 1, "Y"), ("DDD", 333, "Y")]
    rng = np.random.default_rng(5)
    bp = np.sort(rng.choice(np.arange(1, 200000), size=48, replace=False))
    G, _ = synth.synth_genotypes(bp, pops, seed=9)
    G = G[G.min(1) != G.max(1)]
    off = synth.pop_offsets([q[1] for q in pops])
    assert np.array_equal(hotpath.gram_counts(G, ctx=ctx), oracle.gram_counts(G))
    # a population of one sample makes the reference's m/(m-1) infinite: reproduce, do not "fix"
    w = np.array([0.5, 0.2, 0.0, 0.3])
    got = hotpath.ld_matrix(G, off, w, ctx=ctx)
    want = oracle.compute_ld(G, off, w)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    pops3 = [pops[0], pops[1], pops[3]]
    keep = np.r_[0:2570, 2571:2904]
    G3 = np.ascontiguousarray(G[:, keep])
    off3 = synth.pop_offsets([q[1] for q in pops3])
    w3 = np.array([0.5, 0.2, 0.3])
    got = hotpath.ld_matrix(G3, off3, w3, ctx=ctx)
    want = oracle.compute_ld(G3, off3, w3)
    assert np.max(np.abs(got - want)) <= LD_TOL


def test_heterozygous_only_snp_gives_nan_like_reference(ctx):
    # a SNP whose genotypes are all '1' passes the AF filter (af = 0.5) but has zero variance:
    # the reference's CalCor returns 0/0 and every output of the window becomes NaN.
    p = small_panel(n_snp=60, scale=0.01, n_pops=4)
    gm, gu, z1 = split_window(p, 25)
    gm = gm.copy()
    gm[3, :] = 1
    got = hotpath.impute_window(0, gm, gu, p["off"], None, z1, ctx=ctx)
    want = oracle.run_impute(0, gm, gu, p["off"], None, z1)
    assert np.all(np.isnan(want["z"]))
    assert np.all(np.isnan(got["z"])) and np.all(np.isnan(got["info"]))
    assert got["status"] & 2


def test_makeposdef_clamp_path(ctx):
    # duplicated measured SNPs with lambda = 0 make B11 singular: MakePosDef (util.cpp:310-317)
    # lifts the zero eigenvalues to 1e-5.  The GPU detects this through the shifted factorisation
    # and clamps the spectrum on the device.
    p = small_panel(n_snp=70, scale=0.02, n_pops=6, seed=21)
    gm, gu, z1 = split_window(p, 30)
    gm = np.ascontiguousarray(np.vstack([gm, gm[:3]]))
    z1 = np.concatenate([z1, z1[:3]])
    got = hotpath.impute_window(0, gm, gu, p["off"], None, z1, lam=0.0, want_mats=True, ctx=ctx)
    want = oracle.run_impute(0, gm, gu, p["off"], None, z1, lam=0.0, want_mats=True)
    assert want["mpd"] == 1
    assert got["status"] & 1
    assert np.max(np.abs(got["b11"] - want["b11"])) <= 1e-9
    assert relerr(got["info"], want["info"]) <= 1e-5
    assert np.max(np.abs(got["z"] - want["z"]) / np.maximum(1.0, np.abs(want["z"]))) <= 1e-5


def test_gene_ld_batch(ctx):
    p = small_panel(n_snp=330, scale=0.01, n_pops=8, seed=4)
    G = p["G"]
    S = G.shape[0]
    rng = np.random.default_rng(8)
    cuts = np.sort(rng.choice(np.arange(1, S), size=40, replace=False))
    gene_off = np.r_[0, cuts, S].astype(np.int32)          # includes genes straddling tile edges
    for mode, w, fn in ((0, None, lambda g: oracle.ld_pooled(g, p["off"], 1.1)),
                        (1, p["w"], None)):
        blocks = hotpath.gene_ld_batch(G, p["off"], gene_off, pop_wgt=w, mode=mode, diag=1.1, ctx=ctx)
        assert len(blocks) == len(gene_off) - 1
        for g, blk in enumerate(blocks):
            rows = G[gene_off[g]:gene_off[g + 1]]
            if mode == 0:
                want = fn(rows)
            else:
                want = oracle.compute_ld(rows, p["off"], p["w"])
                np.fill_diagonal(want, 1.1)
            assert blk.shape == want.shape
            assert np.max(np.abs(blk - want)) <= LD_TOL


def test_job_batches_heterogeneous_windows(ctx):
    p = small_panel(n_snp=260, scale=0.02, seed=31)
    wins, wants = [], []
    for k, (mode, M, U) in enumerate([(0, 40, 55), (1, 130, 20), (1, 12, 140), (0, 75, 75)]):
        G = p["G"][k * 5: k * 5 + M + U]
        gm, gu, z1 = split_window(dict(G=G), M, seed=k)
        wins.append(dict(mode=mode, geno_m=gm, geno_u=gu, pop_off=p["off"], pop_wgt=p["w"], z1=z1))
        wants.append(oracle.run_impute(mode, gm, gu, p["off"], p["w"], z1))
    job = hotpath.Job(wins, ctx=ctx)
    for _ in range(2):                                     # a job can be re-run (bench does)
        job.run()
        res = job.fetch()
    for got, want in zip(res, wants):
        assert got["status"] == 0
        assert relerr(got["info"], want["info"]) <= Z_TOL
        assert np.max(np.abs(got["z"] - want["z"]) / np.maximum(1.0, np.abs(want["z"]))) <= Z_TOL
    work = job.work()
    assert work["imputed_snps"] == 55 + 20 + 140 + 75
    job.close()


def test_invalid_arguments_are_reported(ctx):
    p = small_panel(n_snp=30, scale=0.01, n_pops=3)
    with pytest.raises(Exception) as ei:
        hotpath.ld_matrix(p["G"], p["off"], None, mode=hotpath.MODE_WEIGHTED, ctx=ctx)
    assert "pop_wgt" in str(ei.value)
    bad = p["off"].copy()
    bad[1] = bad[2] + 5
    with pytest.raises(Exception):
        hotpath.ld_matrix(p["G"], bad, p["w"], ctx=ctx)


def test_int8_gram_path_is_bit_identical_to_f32_path(ctx):
    """The optional i8-MFMA Gram kernel accumulates the same exact integers, so LD, z and info must be
    bit-identical to the default f32-MFMA path (and the counts bit-exact against the oracle)."""
    p = small_panel(n_snp=420, scale=0.03, seed=41)
    G = p["G"]
    gm, gu, z1 = split_window(dict(G=G[:330]), 140)
    base = {}
    try:
        for dt in ("f32", "i8"):
            ctx.set_gram_dtype(dt)
            cnt = hotpath.gram_counts(G[:200], ctx=ctx)
            assert np.array_equal(cnt, oracle.gram_counts(G[:200]))
            ld = hotpath.ld_matrix(G[:300], p["off"], p["w"], ctx=ctx)
            r0 = hotpath.impute_window(0, gm, gu, p["off"], None, z1, want_mats=True, ctx=ctx)
            r1 = hotpath.impute_window(1, gm, gu, p["off"], p["w"], z1, want_mats=True, ctx=ctx)
            base[dt] = (ld, r0, r1)
    finally:
        ctx.set_gram_dtype(os.environ.get("GAUSS_GRAM_DTYPE", "f32"))      # (back to the session's form: a suite run under GAUSS_GRAM_DTYPE=i8 stays on it)
    (lda, a0, a1), (ldb, b0, b1) = base["f32"], base["i8"]
    assert np.array_equal(lda, ldb)
    for a, b in ((a0, b0), (a1, b1)):
        for k in ("b11", "b21", "z", "info"):
            assert np.array_equal(a[k], b[k]), k
    want = oracle.run_impute(1, gm, gu, p["off"], p["w"], z1)
    assert relerr(b1["info"], want["info"]) <= Z_TOL


R_TOL = 1e-9        # QCAT r: one-pass moments after the MFMA triangular solve vs the reference's two-pass CalCor


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("shape", [(11, 0, 2, 9), (70, 90, 10, 50), (150, 130, 0, 150), (130, 64, 60, 63)])
def test_qcat_window_matches_oracle(ctx, mode, shape):
    """run_qcat / run_qcatmix core (qcat.cpp:166-245, qcatmix.cpp:179-277)."""
    M, U, n_head, n_pred = shape
    p = small_panel(n_snp=M + U + 30, scale=0.02, seed=17 + M)
    G = p["G"][: M + U]
    gm, gu, z1 = split_window(dict(G=G), M)
    got = hotpath.qcat_window(mode, gm, gu if U else None, p["off"], p["w"], z1, n_head, n_pred,
                              want_mats=True, ctx=ctx)
    want = oracle.run_qcat(mode, gm, gu if U else None, p["off"], p["w"], z1, n_head, n_pred, want_mats=True)
    assert got["status"] == 0
    assert got["num_eig"] == want["num_eig"] == M
    assert np.max(np.abs(got["b11"] - want["b11"])) <= LD_TOL
    if U:
        assert np.max(np.abs(got["b21"] - want["b21"])) <= LD_TOL
    assert got["r"].shape == want["r"].shape == (n_pred + U,)
    assert np.max(np.abs(got["r"] - want["r"])) <= R_TOL


def test_qcat_counts_eigenvalues_below_cutoff(ctx):
    # duplicated measured SNPs with a ridge below eig_cutoff: CountPC (util.cpp:355-388) drops them from
    # num_eig while the Cholesky factor still exists (lambda > 0)
    p = small_panel(n_snp=80, scale=0.02, n_pops=6, seed=23)
    gm, gu, z1 = split_window(p, 36)
    gm = np.ascontiguousarray(np.vstack([gm, gm[:5]]))
    z1 = np.concatenate([z1, z1[:5]])
    got = hotpath.qcat_window(0, gm, gu, p["off"], None, z1, 4, 20, lam=0.004, ctx=ctx)
    want = oracle.run_qcat(0, gm, gu, p["off"], None, z1, 4, 20, lam=0.004)
    assert want["num_eig"] <= len(gm) - 5
    assert got["num_eig"] == want["num_eig"]
    assert np.max(np.abs(got["r"] - want["r"])) <= 1e-7


def test_qcat_and_impute_windows_share_a_job(ctx):
    p = small_panel(n_snp=260, scale=0.02, seed=33)
    gm, gu, z1 = split_window(dict(G=p["G"][:200]), 120)
    wins = [dict(mode=1, geno_m=gm, geno_u=gu, pop_off=p["off"], pop_wgt=p["w"], z1=z1),
            dict(mode=1, geno_m=gm, geno_u=gu, pop_off=p["off"], pop_wgt=p["w"], z1=z1, qcat=(20, 70, 0.01)),
            dict(mode=0, geno_m=gm[:40], geno_u=gu[:30], pop_off=p["off"], pop_wgt=None, z1=z1[:40], qcat=(0, 40, 0.01))]
    job = hotpath.Job(wins, ctx=ctx)
    job.run()
    res = job.fetch()
    job.close()
    a = oracle.run_impute(1, gm, gu, p["off"], p["w"], z1)
    b = oracle.run_qcat(1, gm, gu, p["off"], p["w"], z1, 20, 70)
    c = oracle.run_qcat(0, gm[:40], gu[:30], p["off"], None, z1[:40], 0, 40)
    assert relerr(res[0]["info"], a["info"]) <= Z_TOL
    assert np.max(np.abs(res[1]["r"] - b["r"])) <= R_TOL and res[1]["num_eig"] == b["num_eig"]
    assert np.max(np.abs(res[2]["r"] - c["r"])) <= R_TOL and res[2]["num_eig"] == c["num_eig"]


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("codings", [(0,), (0, 1, 2), (2,)])
def test_ld_export_with_recoded_rows_matches_oracle(ctx, mode, codings):
    """Raw LD export for prep_qcat (prep_qcat.cpp:104-132) and prep_recessive_impute
    (prep_qcatmix.cpp:136-221): B11 with unit diagonal, B21 per coding of the prediction-window SNPs."""
    p = small_panel(n_snp=330, scale=0.02, seed=51)
    gm, gu = p["G"][:140], p["G"][100:300]                       # overlapping sets, as in prep_qcat (pred_all includes measured)
    mask = sum(1 << c for c in codings)
    got = hotpath.ld_window(mode, gm, gu, p["off"], p["w"], lam=0.0, codings=mask, ctx=ctx)
    want = oracle.ld_blocks(mode, gm, gu, p["off"], p["w"], diag=1.0, codings=codings)
    assert got["b21"].shape == want["b21"].shape == (len(codings) * 200, 140)
    assert np.all(np.diag(got["b11"]) == 1.0)
    assert np.max(np.abs(got["b11"] - want["b11"])) <= LD_TOL
    # a rare SNP without homozygotes recodes to an all-zero recessive row: 0/0 = NaN in the reference too
    nan = np.isnan(want["b21"])
    assert np.array_equal(np.isnan(got["b21"]), nan)
    assert np.max(np.abs(got["b21"][~nan] - want["b21"][~nan])) <= LD_TOL
    assert np.mean(got["b21"][~nan] == want["b21"][~nan]) > 0.999


def test_ld_export_recoding_leaves_codes_above_two_alone(ctx):
    # gauss.cpp:1209-1215: characters outside '0'..'2' are copied through unchanged
    p = small_panel(n_snp=60, scale=0.01, n_pops=4, seed=52)
    gm = p["G"][:30]
    gu = p["G"][30:50].copy()
    rng = np.random.default_rng(3)
    gu[rng.random(gu.shape) < 0.05] = 3
    gu[rng.random(gu.shape) < 0.02] = 7
    got = hotpath.ld_window(0, gm, gu, p["off"], None, codings=6, ctx=ctx)
    want = oracle.ld_blocks(0, gm, gu, p["off"], None, codings=(1, 2))
    nan = np.isnan(want["b21"])
    assert np.array_equal(np.isnan(got["b21"]), nan) and not nan.all()
    assert np.max(np.abs(got["b21"][~nan] - want["b21"][~nan])) <= LD_TOL


def _same(a, b, keys=("z", "info", "b11", "b21")):
    return all(np.array_equal(a[k], b[k]) for k in keys)


@pytest.mark.parametrize("mode", [0, 1])
def test_packed_2bit_rows_give_bit_identical_results(ctx, mode):
    """GAUSS_GENO_2BIT sources (packed panel rows) feed the same exact Gram: LD, z and info must equal the
    one-byte-per-genotype path bit for bit, and hence the oracle."""
    from gauss_amd import panel
    p = small_panel(n_snp=300, scale=0.03, seed=61)
    G, off = p["G"], p["off"]
    gm, gu = np.ascontiguousarray(G[:110]), np.ascontiguousarray(G[110:260])
    z1 = np.random.default_rng(7).normal(size=110) * 2.0
    base = hotpath.impute_window(mode, gm, gu, off, p["w"], z1, want_mats=True, ctx=ctx)
    rows, src_off = panel.pack2bit(G[:260], off)
    assert np.array_equal(panel.unpack2bit(rows, np.diff(off)), G[:260])
    # (a) contiguous packed matrices from host memory
    job = hotpath.Job([dict(mode=mode, geno_m=rows[:110], geno_u=rows[110:], pop_off=off, pop_wgt=p["w"], z1=z1,
                            packed=dict(fmt=1))], ctx=ctx, want_mats=True)
    job.run()
    a = job.fetch()[0]
    job.close()
    assert _same(a, base)
    # (b) row lists into a host row store (shuffled store order)
    perm = np.random.default_rng(1).permutation(260)
    store = np.ascontiguousarray(rows[perm])
    inv = np.argsort(perm).astype(np.int32)
    job = hotpath.Job([dict(mode=mode, geno_m=store, geno_u=store, pop_off=off, pop_wgt=p["w"], z1=z1,
                            packed=dict(fmt=1, rows_m=inv[:110], rows_u=inv[110:]))], ctx=ctx, want_mats=True)
    job.run()
    b = job.fetch()[0]
    job.close()
    assert _same(b, base)
    # (c) the same store resident in HBM, rows gathered by the pack kernel
    rs = hotpath.RowStore(store, ctx=ctx)
    job = hotpath.Job([dict(mode=mode, pop_off=off, pop_wgt=p["w"], z1=z1, dev=(rs.ptr, rs.ptr, 110, 150, rs.ld),
                            packed=dict(fmt=1, rows_m=inv[:110], rows_u=inv[110:]))], ctx=ctx, on_device=True,
                      want_mats=True)
    job.run()
    c = job.fetch()[0]
    job.close()
    rs.close()
    assert _same(c, base)
    want = oracle.run_impute(mode, gm, gu, off, p["w"], z1)
    assert relerr(c["info"], want["info"]) <= Z_TOL


def test_packed_store_with_population_subset(ctx):
    # the store holds every population of the panel; a window selects some of them through pop_src_off
    from gauss_amd import panel
    p = small_panel(n_snp=200, scale=0.03, seed=62, n_pops=7)
    G, off = p["G"], p["off"]
    rows, src_off = panel.pack2bit(G, off)
    sel = [1, 3, 4, 6]
    cols = np.concatenate([np.arange(off[q], off[q + 1]) for q in sel])
    Gs = np.ascontiguousarray(G[:, cols])
    offs = np.concatenate([[0], np.cumsum([off[q + 1] - off[q] for q in sel])]).astype(np.int32)
    w = p["w"][sel]
    keep = np.flatnonzero(Gs.min(1) != Gs.max(1))[:160].astype(np.int32)
    M = 70
    z1 = np.random.default_rng(2).normal(size=M)
    rs = hotpath.RowStore(rows, ctx=ctx)
    for mode in (0, 1):
        base = hotpath.impute_window(mode, Gs[keep[:M]], Gs[keep[M:]], offs, w, z1, want_mats=True, ctx=ctx)
        job = hotpath.Job([dict(mode=mode, pop_off=offs, pop_wgt=w, z1=z1, dev=(rs.ptr, rs.ptr, M, len(keep) - M, rs.ld),
                                packed=dict(fmt=1, rows_m=keep[:M], rows_u=keep[M:], pop_src_off=src_off[sel]))],
                          ctx=ctx, on_device=True, want_mats=True)
        job.run()
        got = job.fetch()[0]
        job.close()
        assert _same(got, base)
    rs.close()


def test_u8_row_lists_from_a_resident_store(ctx):
    p = small_panel(n_snp=150, scale=0.02, seed=63)
    G, off = p["G"], p["off"]
    S, N = G.shape
    store = np.zeros((S, (N + 15) // 16 * 16), dtype=np.uint8)
    store[:, :N] = G
    idx = np.random.default_rng(4).permutation(S).astype(np.int32)
    M = 60
    z1 = np.random.default_rng(5).normal(size=M)
    base = hotpath.impute_window(1, G[idx[:M]], G[idx[M:]], off, p["w"], z1, want_mats=True, ctx=ctx)
    rs = hotpath.RowStore(store, ctx=ctx)
    job = hotpath.Job([dict(mode=1, pop_off=off, pop_wgt=p["w"], z1=z1, dev=(rs.ptr, rs.ptr, M, S - M, rs.ld),
                            packed=dict(fmt=0, rows_m=idx[:M], rows_u=idx[M:]))], ctx=ctx, on_device=True, want_mats=True)
    job.run()
    got = job.fetch()[0]
    job.close()
    rs.close()
    assert _same(got, base)


def test_per_population_ld_matches_oracle(ctx):
    """prep_zmix5's pair table (zmix.cpp:158-176): CalCor on each population's strings for every SNP pair."""
    p = small_panel(n_snp=170, scale=0.03, seed=71, n_pops=9)
    G, off = p["G"][:150], p["off"]
    got = hotpath.ld_per_pop(G, off, ctx=ctx)
    want = oracle.ld_per_pop(G, off)
    assert got.shape == want.shape == (9, 150 * 149 // 2)
    nan = np.isnan(want)                                   # SNP monomorphic in a small population: 0/0 in both
    assert np.array_equal(np.isnan(got), nan) and nan.mean() < 0.5
    assert np.max(np.abs(got[~nan] - want[~nan])) <= LD_TOL
    assert np.mean(got[~nan] == want[~nan]) > 0.999


@pytest.mark.parametrize("wscale,lam", [(1.0, 0.1), (3.0, 0.1), (3.0, 2e-5), (8.0, 1e-3), (40.0, 1e-3), (1.0, 0.0)])
def test_shift_certificate_never_changes_the_answer(ctx, wscale, lam, monkeypatch):
    """The factorisation of B11 - eps I is skipped when lambda - (sum w - 1)+ * sum_p w_p |mu_p / sd|^2 certifies
    lambda_min(B11) > eps (k_solve.hip:shift_cert_kernel).  Whatever the certificate decides -- weights far above
    1, lambda next to eps, lambda = 0 -- z, info and the MakePosDef decision must equal the oracle's, which runs
    the reference's eigen-decomposition test (util.cpp:302-318)."""
    p = small_panel(n_snp=90, scale=0.02, n_pops=7, seed=31)
    gm, gu, z1 = split_window(p, 40)
    w = p["w"] * wscale
    got = hotpath.impute_window(1, gm, gu, p["off"], w, z1, lam=lam, ctx=ctx)
    # GAUSS_NO_SHIFT_CERT=1 (README: the supported fallback that never trusts the bound and always factors B11 - eps I too): the
    # same decision and the same bits, window by window
    monkeypatch.setenv("GAUSS_NO_SHIFT_CERT", "1")
    exact = hotpath.impute_window(1, gm, gu, p["off"], w, z1, lam=lam, ctx=ctx)
    monkeypatch.delenv("GAUSS_NO_SHIFT_CERT")
    assert exact["status"] == got["status"]
    assert np.array_equal(exact["z"], got["z"], equal_nan=True) and np.array_equal(exact["info"], got["info"], equal_nan=True)
    want = oracle.run_impute(1, gm, gu, p["off"], w, z1, lam=lam, want_mats=True)
    if want["mpd"] < 0:       # weights far above 1 can turn a self-covariance negative: NaN everywhere, like the reference
        assert got["status"] & 2 and np.all(np.isnan(got["z"])) and np.all(np.isnan(want["z"]))
        return
    assert bool(got["status"] & 1) == bool(want["mpd"])
    tol = 1e-5 if want["mpd"] else 1e-8
    assert relerr(got["info"], want["info"]) <= tol
    assert np.max(np.abs(got["z"] - want["z"]) / np.maximum(1.0, np.abs(want["z"]))) <= tol


@pytest.mark.parametrize("mode", [0, 1])
def test_ld_and_gene_batches_over_row_stores_equal_the_byte_calls(ctx, mode):
    """gauss_ld_rows / gauss_gene_ld_batch_rows (what computeLD / jepeg use on a packed panel: rows named in a host
    or resident store, 2-bit or one byte per genotype) must return the bits of gauss_ld / gauss_gene_ld_batch."""
    from gauss_amd import panel
    p = small_panel(n_snp=260, scale=0.03, seed=77)
    G, off, w = p["G"], p["off"], p["w"]
    S = G.shape[0]
    rng = np.random.default_rng(3)
    pick = np.sort(rng.choice(S, size=150, replace=False)).astype(np.int32)
    cuts = np.concatenate([[0], np.sort(rng.choice(np.arange(1, 150), size=30, replace=False)), [150]]).astype(np.int32)
    diag = 1.1 if mode == 0 else 1.0
    want_ld = hotpath.ld_matrix(np.ascontiguousarray(G[pick]), off, w, mode=mode, diag=diag, ctx=ctx)
    want_blocks = hotpath.gene_ld_batch(np.ascontiguousarray(G[pick]), off, cuts, pop_wgt=w, mode=mode, diag=diag, ctx=ctx)
    rows2, src_off = panel.pack2bit(G, off)
    stores = [("2-bit host", np.ascontiguousarray(rows2), 1, src_off), ("u8 host", np.ascontiguousarray(G), 0, None)]
    res = hotpath.RowStore(rows2, ctx=ctx)
    stores.append(("2-bit resident", res, 1, src_off))
    for name, st, fmt, so in stores:
        got = hotpath.ld_matrix_rows(st, pick, off, w, mode=mode, diag=diag, fmt=fmt, pop_src_off=so, ctx=ctx)
        assert np.array_equal(got, want_ld), name
        blocks = hotpath.gene_ld_batch_rows(st, pick, off, cuts, pop_wgt=w, mode=mode, diag=diag, fmt=fmt, pop_src_off=so, ctx=ctx)
        assert len(blocks) == len(want_blocks) and all(np.array_equal(a, b) for a, b in zip(blocks, want_blocks)), name
    res.close()
    want = oracle.compute_ld(G[pick], off, w) if mode == 1 else oracle.ld_pooled(G[pick], off, diag)
    assert np.max(np.abs(want_ld - want)) <= 1e-12


@pytest.mark.parametrize("mode", [0, 1])
def test_solve_forms_agree(ctx, mode):
    """The fused path (k_solve.hip) has job-size dependent forms of the same arithmetic: the factorisation with or without
    panel launches (jobs of up to 20 windows form their panel tiles inside the update launches, gauss_job.h:
    OWN_PANEL_MAX_WINDOWS) and the closing product in tiles of 64 or 128 right-hand sides (jobs with fewer than 1 600 tiles
    of 128 take 64, GEMM_SMALL_TILES).  The same four windows alone (small-job forms) and among 27 filler windows that push the
    job over both thresholds (large-job forms): each within 1e-8 of the oracle and bit-identical to the other (every form
    sums in the same order).  Rows of the inverse with one early product (uncut) and with several (cut into class sums) occur
    in both: the windows have 1 to 5 factor blocks."""
    p = small_panel(n_snp=420, scale=0.03, seed=5)
    G, off = p["G"], p["off"]
    rng = np.random.default_rng(2)
    wins = []
    for (m, u) in [(300, 100), (130, 200), (65, 70), (64, 10)]:          # 5, 3, 2 and 1 factor blocks
        idx = rng.permutation(G.shape[0])
        gm, gu = np.ascontiguousarray(G[np.sort(idx[:m])]), np.ascontiguousarray(G[np.sort(idx[m:m + u])])
        wins.append(dict(mode=mode, geno_m=gm, geno_u=gu, pop_off=off, pop_wgt=p["w"], z1=rng.standard_normal(m) * 2))
    # filler: 27 windows of 300 measured SNPs and 2 560 unmeasured rows (rows may repeat among the unmeasured: each is imputed
    # on its own) = 27 x 20 x 3 = 1 620 tiles of the closing product at 128 right-hand sides, 31 windows in the job
    filler = []
    for k in range(27):
        idx = rng.permutation(G.shape[0])
        gm = np.ascontiguousarray(G[np.sort(idx[:300])])
        gu = np.ascontiguousarray(G[rng.integers(0, G.shape[0], size=2560)])
        filler.append(dict(mode=mode, geno_m=gm, geno_u=gu, pop_off=off, pop_wgt=p["w"], z1=rng.standard_normal(300)))
    out = {}
    for name, batch in (("small", wins), ("large", wins + filler)):
        job = hotpath.Job(batch, ctx=ctx)
        job.run()
        out[name] = job.fetch()[:len(wins)]
        job.close()
    for w, a, b in zip(wins, out["small"], out["large"]):
        want = oracle.run_impute(mode, w["geno_m"], w["geno_u"], off, p["w"], w["z1"])
        for r in (a, b):
            assert r["status"] == 0
            assert relerr(r["info"], want["info"]) <= Z_TOL
            assert np.max(np.abs(r["z"] - want["z"]) / np.maximum(1.0, np.abs(want["z"]))) <= Z_TOL
        assert np.array_equal(a["info"], b["info"]) and np.array_equal(a["z"], b["z"])


@pytest.mark.parametrize("mode", [0, 1])
def test_pieces_of_a_cut_window_equal_the_whole_window(ctx, mode):
    """A window's unmeasured SNPs are imputed independently given its measured set (dist.cpp:181-198), so the
    multi-GPU planner may cut a window between two ranks (farm.level_windows): the pieces -- same measured SNPs, a
    slice of the unmeasured ones, each in a job of its own with other windows around it -- must return the whole
    window's z / info bit for bit."""
    p = small_panel(n_snp=520, scale=0.03, seed=9)
    G, off = p["G"], p["off"]
    rng = np.random.default_rng(7)
    idx = rng.permutation(G.shape[0])
    m, u = 200, 300
    gm, gu = np.ascontiguousarray(G[np.sort(idx[:m])]), np.ascontiguousarray(G[np.sort(idx[m:m + u])])
    z1 = rng.standard_normal(m) * 2
    other = dict(mode=mode, geno_m=np.ascontiguousarray(G[:90]), geno_u=np.ascontiguousarray(G[90:160]), pop_off=off,
                 pop_wgt=p["w"], z1=rng.standard_normal(90))

    def run(wins):
        job = hotpath.Job(wins, ctx=ctx)
        job.run()
        out = job.fetch()
        job.close()
        return out
    whole = run([dict(mode=mode, geno_m=gm, geno_u=gu, pop_off=off, pop_wgt=p["w"], z1=z1)])[0]
    want = oracle.run_impute(mode, gm, gu, off, p["w"], z1)
    assert relerr(whole["info"], want["info"]) <= Z_TOL
    cuts = [0, 64, 193, u]                                     # a tile-aligned and a ragged cut
    z, info = [], []
    for a, b in zip(cuts, cuts[1:]):
        piece = dict(mode=mode, geno_m=gm, geno_u=np.ascontiguousarray(gu[a:b]), pop_off=off, pop_wgt=p["w"], z1=z1)
        res = run([other, piece] if a else [piece, other])
        r = res[1] if a else res[0]
        assert r["status"] == 0
        z.append(r["z"]); info.append(r["info"])
    assert np.array_equal(np.concatenate(z), whole["z"]) and np.array_equal(np.concatenate(info), whole["info"])


def test_jobs_queued_back_to_back_keep_their_results_apart(ctx):
    """Pipelines (the chromosome driver): three jobs of one context queued before any fetch, twice over, must each
    return what they return alone -- a job's result mirrors travel with its own run and its fetch waits for its
    own event only."""
    p = small_panel(n_snp=400, scale=0.03, seed=13)
    rng = np.random.default_rng(3)
    jobs, alone = [], []
    for k, (m, u) in enumerate([(150, 120), (70, 200), (260, 90)]):
        idx = rng.permutation(p["G"].shape[0])
        w = dict(mode=k % 2, geno_m=np.ascontiguousarray(p["G"][np.sort(idx[:m])]),
                 geno_u=np.ascontiguousarray(p["G"][np.sort(idx[m:m + u])]), pop_off=p["off"], pop_wgt=p["w"],
                 z1=rng.standard_normal(m))
        j = hotpath.Job([w], ctx=ctx)
        j.run()
        alone.append(j.fetch()[0])
        jobs.append(j)
    for _ in range(2):
        for j in jobs:
            j.run()
        for j, want in zip(jobs, alone):
            got = j.fetch()[0]
            assert np.array_equal(got["z"], want["z"]) and np.array_equal(got["info"], want["info"])
    for j in jobs:
        j.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_tail_block_edges(ctx, mode):
    """The fused tail works in 64-blocks of the measured SNPs (factor, inverse panels [I | z1]) and 128-blocks of the
    product's tiles: shapes on every side of those edges -- M = 64 k puts the z1 column in a panel of its own, M < 64
    is a single block, U = 1 a single right-hand side -- for dist/distmix and for QCAT (whose first right-hand sides
    are columns of B11), all windows of one job, against the oracle."""
    shapes = [(11, 1), (63, 5), (64, 64), (65, 63), (127, 130), (128, 1), (129, 70), (192, 129), (200, 257)]
    p = small_panel(n_snp=480, scale=0.03, seed=77)
    G, off = p["G"], p["off"]
    rng = np.random.default_rng(11)
    wins, wants = [], []
    for m, u in shapes:
        idx = rng.permutation(G.shape[0])
        gm, gu = np.ascontiguousarray(G[np.sort(idx[:m])]), np.ascontiguousarray(G[np.sort(idx[m:m + u])])
        z1 = rng.standard_normal(m) * 2
        wins.append(dict(mode=mode, geno_m=gm, geno_u=gu, pop_off=off, pop_wgt=p["w"], z1=z1))
        wants.append(("impute", oracle.run_impute(mode, gm, gu, off, p["w"], z1)))
        n_head, n_pred = m // 5, m - m // 5 - m // 7
        wins.append(dict(mode=mode, geno_m=gm, geno_u=gu, pop_off=off, pop_wgt=p["w"], z1=z1, qcat=(n_head, n_pred, 0.01)))
        wants.append(("qcat", oracle.run_qcat(mode, gm, gu, off, p["w"], z1, n_head, n_pred)))
    job = hotpath.Job(wins, ctx=ctx)
    job.run()
    res = job.fetch()
    job.close()
    for (kind, want), got, w in zip(wants, res, wins):
        shape = (w["geno_m"].shape[0], w["geno_u"].shape[0])
        assert got["status"] == 0, shape
        if kind == "impute":
            assert relerr(got["info"], want["info"]) <= Z_TOL, shape
            assert np.max(np.abs(got["z"] - want["z"]) / np.maximum(1.0, np.abs(want["z"]))) <= Z_TOL, shape
        else:
            assert got["num_eig"] == want["num_eig"], shape
            assert np.max(np.abs(got["r"] - want["r"])) <= R_TOL, shape


def test_side_stream_does_not_change_results(monkeypatch):
    """A context created with GAUSS_SIDE_STREAM=0 runs every kernel of a job on one stream; the default context runs
    B21's epilogue tiles, the row tables and the certificate on its side stream beside the main one.  Same windows,
    same bits -- including a re-run of the same job, whose side-stream work must not overtake the previous run."""
    p = small_panel(n_snp=460, scale=0.03, seed=21)
    rng = np.random.default_rng(5)
    wins = []
    for k, (m, u) in enumerate([(210, 150), (90, 260), (140, 40)]):
        idx = rng.permutation(p["G"].shape[0])
        wins.append(dict(mode=k % 2, geno_m=np.ascontiguousarray(p["G"][np.sort(idx[:m])]),
                         geno_u=np.ascontiguousarray(p["G"][np.sort(idx[m:m + u])]), pop_off=p["off"], pop_wgt=p["w"],
                         z1=rng.standard_normal(m)))
    out = {}
    for name, val in (("two", "1"), ("one", "0")):
        monkeypatch.setenv("GAUSS_SIDE_STREAM", val)
        c = hotpath.Context(0)
        job = hotpath.Job(wins, ctx=c)
        runs = []
        for _ in range(3):
            job.run()
            runs.append(job.fetch())
        job.close()
        c.close()
        for r in runs[1:]:
            for a, b in zip(runs[0], r):
                assert np.array_equal(a["z"], b["z"]) and np.array_equal(a["info"], b["info"])
        out[name] = runs[0]
    for a, b in zip(out["two"], out["one"]):
        assert a["status"] == b["status"] == 0
        assert np.array_equal(a["z"], b["z"]) and np.array_equal(a["info"], b["info"])


def test_two_runs_of_a_job_in_flight(ctx):
    """gauss_job_run may be called again before the previous run has been fetched (two result mirrors per job): the
    fetches then return the runs in order, with the bits a run-fetch-run-fetch sequence gives -- also for a job whose
    fetch reads device buffers (matrix export) -- and a third run without a fetch is refused."""
    p = small_panel(n_snp=380, scale=0.03, seed=29)
    rng = np.random.default_rng(8)
    wins = []
    for k, (m, u) in enumerate([(170, 120), (66, 190)]):
        idx = rng.permutation(p["G"].shape[0])
        wins.append(dict(mode=k % 2, geno_m=np.ascontiguousarray(p["G"][np.sort(idx[:m])]),
                         geno_u=np.ascontiguousarray(p["G"][np.sort(idx[m:m + u])]), pop_off=p["off"], pop_wgt=p["w"],
                         z1=rng.standard_normal(m)))
    for want_mats in (False, True):
        job = hotpath.Job(wins, ctx=ctx, want_mats=want_mats)
        job.run()
        ref = job.fetch()
        job.run()
        job.run()
        with pytest.raises(Exception) as ei:
            job.run()
        assert "in flight" in str(ei.value)
        a = job.fetch()
        job.run()
        b = job.fetch()
        c = job.fetch()
        with pytest.raises(Exception):
            job.fetch()
        for got in (a, b, c):
            for r, w in zip(got, ref):
                assert np.array_equal(r["z"], w["z"]) and np.array_equal(r["info"], w["info"])
                if want_mats:
                    assert np.array_equal(r["b11"], w["b11"]) and np.array_equal(r["b21"], w["b21"])
        job.close()


@pytest.mark.gpu
def test_shared_measured_rows_give_the_bits_of_separate_windows(ctx, monkeypatch):
    """Windows of a chromosome name their measured SNPs as runs of one ascending list of store rows: the job packs those
    rows once and multiplies B11's tile pairs once on job-wide row tiles (gauss_plan.cpp, shared measured rows) -- LD
    entries depend on the SNP pair only (distmix.cpp:190-200), so z, info, B11 and B21 must equal, bit for bit, what the
    same windows give when every window packs and multiplies its own copy.  Window starts fall inside row tiles (not
    multiples of 128), windows overlap by half, one window ends the list, one QCAT window shares the job."""
    p = small_panel(n_snp=1500, scale=0.05, seed=31)
    G = p["G"]
    rows2, src_off = panel_mod.pack2bit(G, p["off"])
    store = hotpath.RowStore(rows2, ctx=ctx)
    rng = np.random.default_rng(12)
    n = G.shape[0]
    measured = np.sort(rng.choice(n, size=n // 3, replace=False))
    unmeasured = np.setdiff1d(np.arange(n), measured)
    z = rng.standard_normal(n)
    wins = []
    for k, (a, b) in enumerate([(0, 230), (97, 340), (211, 455), (330, len(measured))]):
        mi = measured[a:b]
        lo, hi = mi[len(mi) // 4], mi[3 * len(mi) // 4]
        ui = unmeasured[(unmeasured > lo) & (unmeasured < hi)]
        d = dict(mode=k % 2 if k < 3 else 1, pop_off=p["off"], pop_wgt=p["w"], z1=z[mi], dev=(store.ptr, store.ptr, len(mi), len(ui), store.ld),
                 packed=dict(fmt=1, rows_m=mi.astype(np.int32), rows_u=ui.astype(np.int32), pop_src_off=src_off))
        wins.append(d)
    # the same populations and mode are a condition for sharing: use one mode for the job proper
    for d in wins:
        d["mode"] = 1
    wins[1] = dict(wins[1], qcat=(20, 60, 0.01))

    def run(share):
        monkeypatch.setenv("GAUSS_SHARE_MEASURED", "1" if share else "0")
        job = hotpath.Job(wins, ctx=ctx, on_device=True, want_mats=True)
        st = job.stats()
        job.run()
        a = job.fetch()
        job.run()                                   # a second run of the same job: nothing stale is left behind
        b = job.fetch()
        job.close()
        for x, y in zip(a, b):
            for key in x:
                if isinstance(x[key], np.ndarray):
                    assert np.array_equal(x[key], y[key], equal_nan=True), key
        return a, st

    own, st_own = run(False)
    shared, st_sh = run(True)
    assert all("b11" in r and "b21" in r for r in own)
    assert st_sh["executed_flops"] < st_own["executed_flops"] and st_sh["items"] < st_own["items"]
    for x, y in zip(own, shared):
        for key in x:
            if isinstance(x[key], np.ndarray):
                assert np.array_equal(x[key], y[key], equal_nan=True), key
            else:
                assert x[key] == y[key], key
    # Clusters (round 4): the windows of a multi-GPU share are scattered -- a window that does not continue the previous one's
    # rows starts a cluster of its own on a tile boundary, two neighbours share theirs.  Here: windows 0 and 1 (neighbours), a
    # far one, then windows 2 and 3 out of chromosome order (3 before 2: not a run of the cluster, a cluster of its own) and a
    # window whose measured rows are not even ascending.  Same bits as separate windows, and never more issued work.
    far = measured[len(measured) - 140:]
    ui_far = unmeasured[unmeasured > far[70]][:90]
    w_far = dict(mode=1, pop_off=p["off"], pop_wgt=p["w"], z1=z[far], dev=(store.ptr, store.ptr, len(far), len(ui_far), store.ld),
                 packed=dict(fmt=1, rows_m=far.astype(np.int32), rows_u=ui_far.astype(np.int32), pop_src_off=src_off))
    perm = rng.permutation(150)
    mi_p = measured[250:400][perm]
    w_perm = dict(mode=1, pop_off=p["off"], pop_wgt=p["w"], z1=z[mi_p], dev=(store.ptr, store.ptr, len(mi_p), len(ui_far), store.ld),
                  packed=dict(fmt=1, rows_m=mi_p.astype(np.int32), rows_u=ui_far.astype(np.int32), pop_src_off=src_off))
    keep = wins
    wins = [keep[0], keep[1], w_far, keep[3], keep[2], w_perm]
    own2, st_own2 = run(False)
    shared2, st_sh2 = run(True)
    assert st_sh2["executed_flops"] <= st_own2["executed_flops"]         # neighbours share where it saves tile pairs; nobody pays for it
    for x, y in zip(own2, shared2):
        for key in x:
            if isinstance(x[key], np.ndarray):
                assert np.array_equal(x[key], y[key], equal_nan=True), key
            else:
                assert x[key] == y[key], key
    wins = [keep[0], w_far, w_perm]                                          # nothing to share at all: the job keeps the windows' own tiles
    own3, st_own3 = run(False)
    shared3, st_sh3 = run(True)
    assert st_sh3["executed_flops"] == st_own3["executed_flops"] and st_sh3["items"] == st_own3["items"]
    for x, y in zip(own3, shared3):
        assert np.array_equal(x["z"], y["z"]) and np.array_equal(x["info"], y["info"])
    store.close()


@pytest.mark.gpu
def test_chain_beside_the_gram_kernel_gives_the_same_bits(ctx, monkeypatch):
    """The factorisation chain in its small-footprint form (k_solve_lite.hip), queued beside the Gram launch of B21's items
    (GAUSS_CHAIN_ASIDE=2 forces it on a job this small), against the chain behind one Gram launch (=0): z, info, status, B11 and
    B21 bit for bit.  Windows of 2 to 9 factor blocks (M not a multiple of 64), shared and unshared measured rows, a QCAT window,
    a window whose lambda is too small for the certificate (the shifted matrix is factored too), and one whose B11 is not
    positive definite at all (lambda = 0 on duplicated rows: the clamp path reruns it).  The chain beside the Gram kernel comes
    in two launch forms -- ONE Gram launch whose B11 items count themselves off for the chain queue (k_gram.hip:
    wait_count_kernel; GAUSS_CHAIN_MERGED=2 asks for it whatever the queue registry says, and the context's counters must show
    that the runs were queued that way), or two launches joined by an event (GAUSS_CHAIN_MERGED=0) -- with the early windows'
    epilogue tiles on the low-priority queue or all behind the launch (GAUSS_EPI_EARLY=0); all are driven."""
    p = small_panel(n_snp=2600, scale=0.05, seed=41)
    G = p["G"].copy()
    G[7] = G[3]                                    # two identical SNPs: singular B11 at lambda = 0
    rows2, src_off = panel_mod.pack2bit(G, p["off"])
    store = hotpath.RowStore(rows2, ctx=ctx)
    rng = np.random.default_rng(5)
    n = G.shape[0]
    measured = np.sort(np.concatenate([np.arange(12), 12 + rng.choice(n - 12, size=n // 3, replace=False)]))
    unmeasured = np.setdiff1d(np.arange(n), measured)
    z = rng.standard_normal(n)
    wins = []
    for k, (a, b) in enumerate([(0, 131), (97, 340), (211, 760), (330, len(measured)), (500, 640)]):
        mi = measured[a:b]
        lo, hi = mi[len(mi) // 4], mi[3 * len(mi) // 4]
        ui = unmeasured[(unmeasured > lo) & (unmeasured < hi)]
        wins.append(dict(mode=1, pop_off=p["off"], pop_wgt=p["w"], z1=z[mi], dev=(store.ptr, store.ptr, len(mi), len(ui), store.ld),
                         packed=dict(fmt=1, rows_m=mi.astype(np.int32), rows_u=ui.astype(np.int32), pop_src_off=src_off)))
    wins[1] = dict(wins[1], qcat=(20, 60, 0.01))
    wins[4] = dict(wins[4], lam=1e-7)              # no certificate: the exact test factors B11 - eps I as well
    wins[0] = dict(wins[0], lam=0.0)               # singular: status says clamp, the host reruns the window

    def run(aside, share, merged=True, early=True):
        monkeypatch.setenv("GAUSS_CHAIN_MERGED", "2" if merged else "0")     # B11's and B21's items as ONE launch (counted items) or two
        monkeypatch.setenv("GAUSS_EPI_EARLY", "1" if early else "0")
        monkeypatch.setenv("GAUSS_CHAIN_ASIDE", "2" if aside else "0")
        monkeypatch.setenv("GAUSS_SHARE_MEASURED", "1" if share else "0")
        job = hotpath.Job(wins, ctx=ctx, on_device=True, want_mats=True)
        job.run()
        job.run()                                   # two runs in flight
        a = job.fetch()
        b = job.fetch()
        job.close()
        for x, y in zip(a, b):
            for key in x:
                if isinstance(x[key], np.ndarray):
                    assert np.array_equal(x[key], y[key], equal_nan=True), key
        return a

    for share in (False, True):
        behind = run(False, share)
        assert any(r["status"] != 0 for r in behind)
        c0 = ctx.counters()
        forms = [run(True, share), run(True, share, early=False)]
        c1 = ctx.counters()
        assert c1["merged"] - c0["merged"] == 4 and c1["giveups"] == c0["giveups"], (c0, c1)      # two runs each, queued merged
        forms.append(run(True, share, merged=False))
        assert ctx.counters()["merged"] == c1["merged"]
        for other in forms:
            for k, (x, y) in enumerate(zip(behind, other)):
                for key in x:
                    if isinstance(x[key], np.ndarray):
                        assert np.array_equal(x[key], y[key], equal_nan=True), (share, k, key)
                    else:
                        assert x[key] == y[key], (share, k, key)
    store.close()
