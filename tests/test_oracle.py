"""CPU suite: the C oracle against the independent numpy/scipy implementation + invariants.

The reference ships no tests or golden vectors for this path, so the oracle is
cross-validated here and then frozen into tests/golden/ (see tests/golden/make_golden.py).
"""
import numpy as np
import pytest
from scipy import stats

import oracle
from oracle import oracle_np as onp
from helpers import relerr, small_panel, split_window


def test_calcor_and_calwgtcov_match_gram_form():
    p = small_panel(n_snp=50, scale=0.01, n_pops=7)
    G, off, w = p["G"], p["off"], p["w"]
    r = onp.pooled_cor(G)
    c = onp.weighted_cov(G, None, off, w)
    for i, j in [(0, 1), (3, 17), (20, 20), (41, 2)]:
        assert abs(oracle.calcor(G[i], G[j], off) - r[i, j]) <= 1e-14
        assert abs(oracle.calwgtcov(G[i], G[j], off, w) - c[i, j]) <= 1e-12 * max(1.0, abs(c[i, j]))


def test_compute_ld_invariants_and_cross_check():
    p = small_panel(n_snp=70, scale=0.02)
    ld = oracle.compute_ld(p["G"], p["off"], p["w"])
    assert np.array_equal(ld, ld.T)
    assert np.all(np.diag(ld) == 1.0)
    assert np.max(np.abs(ld - onp.compute_ld(p["G"], p["off"], p["w"]))) <= 1e-13
    assert np.linalg.eigvalsh(ld).min() > -1e-8          # a correlation matrix


def test_population_permutation_invariance():
    # CalWgtCov sums over populations: permuting populations (with their weights) changes nothing
    p = small_panel(n_snp=30, scale=0.02, n_pops=6)
    G, off, w = p["G"], p["off"], p["w"]
    perm = [3, 0, 5, 1, 4, 2]
    cols = np.concatenate([np.arange(off[k], off[k + 1]) for k in perm])
    sizes = [off[k + 1] - off[k] for k in perm]
    off2 = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    a = oracle.compute_ld(G, off, w)
    b = oracle.compute_ld(np.ascontiguousarray(G[:, cols]), off2, w[perm])
    assert np.max(np.abs(a - b)) <= 1e-13


@pytest.mark.parametrize("mode", [0, 1])
def test_run_impute_c_vs_numpy(mode):
    p = small_panel(n_snp=110, scale=0.02, seed=5)
    gm, gu, z1 = split_window(p, 45)
    a = oracle.run_impute(mode, gm, gu, p["off"], p["w"], z1, want_mats=True)
    b = onp.run_impute(mode, gm, gu, p["off"], p["w"], z1)
    assert a["mpd"] == b["mpd"] == 0
    assert np.max(np.abs(a["b11"] - b["b11"])) <= 1e-13
    assert relerr(a["info"], b["info"]) <= 1e-10
    assert np.max(np.abs(a["z"] - b["z"])) <= 1e-10
    assert np.all(a["info"] > 0) and np.all(a["info"] < 1.0 + 1e-9)


@pytest.mark.parametrize("mode", [0, 1])
def test_run_qcat_c_vs_numpy(mode):
    # qcat.cpp:166-245 / qcatmix.cpp:179-277: restatement (LLT + full-pivot LU inverse) against an
    # independent numpy Cholesky solve
    p = small_panel(n_snp=120, scale=0.02, seed=6)
    gm, gu, z1 = split_window(p, 50)
    a = oracle.run_qcat(mode, gm, gu, p["off"], p["w"], z1, n_head=7, n_pred=30, want_mats=True)
    b = onp.run_qcat(mode, gm, gu, p["off"], p["w"], z1, 7, 30)
    assert a["num_eig"] == b["num_eig"] == 50
    assert np.max(np.abs(a["b11"] - b["b11"])) <= 1e-13
    assert a["r"].shape == (30 + len(gu),)
    assert np.max(np.abs(a["r"] - b["r"])) <= 1e-10
    assert np.all(np.abs(a["r"]) <= 1.0)
    # no unmeasured SNPs (qcat only guards on the measured count, qcat.cpp:157)
    c = oracle.run_qcat(mode, gm, None, p["off"], p["w"], z1, n_head=7, n_pred=30)
    assert np.array_equal(c["r"], a["r"][:30])


def test_count_pc_counts_small_eigenvalues():
    # CountPC (util.cpp:355-388): duplicated SNPs with lambda = 0 give exact zero eigenvalues
    p = small_panel(n_snp=60, scale=0.02, n_pops=5, seed=3)
    gm, gu, z1 = split_window(p, 24)
    gm = np.ascontiguousarray(np.vstack([gm, gm[:4]]))
    z1 = np.concatenate([z1, z1[:4]])
    b11 = onp.pooled_cor(gm)
    assert oracle.count_pc(b11) == len(gm) - int(np.sum(np.linalg.eigvalsh(b11) < 0.01))
    assert oracle.count_pc(b11) <= len(gm) - 4
    assert oracle.count_pc(b11 + 0.1 * np.eye(len(gm))) == len(gm)


def test_allele_flip_flips_imputed_z():
    # recoding an unmeasured SNP 0<->2 negates its correlations, hence its imputed z; info unchanged
    p = small_panel(n_snp=60, scale=0.02, n_pops=5)
    gm, gu, z1 = split_window(p, 25)
    a = oracle.run_impute(0, gm, gu, p["off"], None, z1)
    gu2 = gu.copy()
    gu2[4] = 2 - gu2[4]
    b = oracle.run_impute(0, gm, gu2, p["off"], None, z1)
    assert abs(a["z"][4] + b["z"][4]) <= 1e-10 and abs(a["info"][4] - b["info"][4]) <= 1e-12
    assert np.max(np.abs(np.delete(a["z"], 4) - np.delete(b["z"], 4))) == 0.0


def test_make_pos_def_and_inverse():
    rng = np.random.default_rng(0)
    for n in (1, 2, 6, 37):
        a = rng.standard_normal((n, n))
        a = a @ a.T / n - 0.4 * np.eye(n)
        got, rc = oracle.make_pos_def(a)
        want, rc2 = onp.make_pos_def(a)
        assert rc == rc2
        assert np.max(np.abs(got - want)) <= 1e-12
        b = want + 0.1 * np.eye(n)
        assert np.max(np.abs(oracle.inv_mat(b) - np.linalg.inv(b))) <= 1e-10
    a = np.eye(5) * 2.0
    got, rc = oracle.make_pos_def(a)
    assert rc == 0 and np.array_equal(got, a)            # untouched when already positive definite


def test_tails_match_scipy():
    for x in (0.0, 0.3, 1.96, 3.7785313, 6.5, 12.0):
        assert abs(oracle.pnorm_upper(x) / stats.norm.sf(x) - 1) <= 1e-12
        for df in range(1, 7):
            want = stats.chi2.sf(x, df)
            assert abs(oracle.pchisq_upper(x, df) - want) <= 1e-13 * max(want, 1e-300) + 1e-300
    assert abs(oracle.pchisq_upper(38.41841, 1) / stats.chi2.sf(38.41841, 1) - 1) <= 1e-10


def _random_gene(rng, p, n):
    rows = p["G"][rng.choice(p["G"].shape[0], n, replace=False)]
    corg = oracle.ld_pooled(rows, p["off"], 1.1)
    z = rng.standard_normal(n) * 2
    info = np.where(rng.random(n) < 0.3, rng.uniform(0.3, 1.0, n), 1.0)     # imputed SNPs carry info < 1 (gene.cpp:871)
    has = (rng.random((n, 6)) < 0.35).astype(np.int32)
    has[rng.integers(0, n), rng.integers(0, 6)] = 1
    wgt = np.where(has, rng.uniform(0.1, 2.0, (n, 6)), 0.0)
    return corg, z, info, has, wgt


def _same_tail(a, b):
    assert a["df"] == b["df"] and a["num_snp"] == b["num_snp"]
    if b["df"]:
        assert abs(a["chisq"] - b["chisq"]) <= 1e-9 * max(1.0, abs(b["chisq"]))
        assert abs(a["jepeg_pval"] - b["jepeg_pval"]) <= 1e-9 * b["jepeg_pval"] + 1e-300
        assert a["top_categ"] == b["top_categ"] and a["top_snp"] == b["top_snp"]
        assert abs(a["top_categ_pval"] - b["top_categ_pval"]) <= 1e-9 * b["top_categ_pval"] + 1e-300
        assert abs(a["top_snp_pval"] - b["top_snp_pval"]) <= 1e-9 * b["top_snp_pval"] + 1e-300
    else:
        assert a["chisq"] == -1.0 and a["jepeg_pval"] == -1.0 and a["top_categ"] == -1 and a["top_snp"] == -1


def test_jepeg_gene_tail_c_oracle_and_product_host_tail_vs_independent_numpy():
    """Three statements of gene.cpp:317-550 must agree: the C oracle (loop-literal), the product's host tail
    (host_tables.cpp:jepeg_tail, reached through gauss_host_jepeg_gene_tail, no GPU) and the independent numpy /
    LAPACK / scipy statement in oracle_np.py that was written from the reference alone.  Cases include collinear
    categories (|r| > 0.8 pruning from the last category down), low-variance categories, imputed SNPs (info < 1),
    genes whose every category is pruned (df = 0) and single-SNP genes."""
    from gauss_amd import api
    rng = np.random.default_rng(3)
    p = small_panel(n_snp=40, scale=0.02, n_pops=5)
    seen_df0 = seen_pruned = 0
    for trial in range(60):
        n = int(rng.integers(1, 14))
        corg, z, info, has, wgt = _random_gene(rng, p, n)
        if trial % 5 == 1 and n >= 2:          # make two categories collinear: identical membership and weights
            has[:, 3], wgt[:, 3] = has[:, 1], wgt[:, 1]
            if not has[:, 1].any():
                has[0, 1] = has[0, 3] = 1
                wgt[0, 1] = wgt[0, 3] = 0.7
        if trial % 7 == 2:                      # a category carried by one weakly informative SNP: low variance
            has[:, 5] = 0
            has[0, 5] = 1
            wgt[:, 5] = 0.0
            wgt[0, 5] = 1.0
            info[0] = 0.05
        want = onp.jepeg_gene_tail(corg, z, info, has, wgt)
        _same_tail(oracle.jepeg_gene_tail(corg, z, info, has, wgt), want)
        _same_tail(api.jepeg_gene_tail(corg, z, info, has, wgt), want)
        k = int(has.any(0).sum())
        seen_df0 += want["df"] == 0
        seen_pruned += 0 < want["df"] < k
    assert seen_pruned >= 5                     # the pruning branches were really exercised


def test_count_pc_c_vs_numpy():
    p = small_panel(n_snp=50, scale=0.02, n_pops=4, seed=12)
    gm = np.vstack([p["G"][:30], p["G"][:6]])                 # duplicated SNPs: eigenvalues at lambda exactly
    for lam in (0.0, 0.005, 0.1):
        b11 = onp.pooled_cor(gm)
        np.fill_diagonal(b11, 1.0 + lam)
        assert oracle.count_pc(b11) == onp.count_pc(b11)
    assert onp.count_pc(b11) == len(gm)


def test_gram_counts_match_numpy():
    p = small_panel(n_snp=25, scale=0.01, n_pops=4)
    G = p["G"].astype(np.int64)
    assert np.array_equal(oracle.gram_counts(p["G"]), G @ G.T)


def test_recode_and_ld_blocks_against_numpy():
    # dominant / recessive recoding (gauss.cpp:1196-1250) then CalCor against additive measured rows
    p = small_panel(n_snp=70, scale=0.02, n_pops=5, seed=8)
    gm, gu = p["G"][:30], p["G"][25:60]
    dom, rec = oracle.recode(gu, 1), oracle.recode(gu, 2)
    assert np.array_equal(dom, (gu >= 1).astype(np.uint8)) and np.array_equal(rec, (gu == 2).astype(np.uint8))
    asc = gu + np.uint8(48)
    assert np.array_equal(oracle.recode(asc, 1), dom + 48) and np.array_equal(oracle.recode(asc, 2), rec + 48)
    got = oracle.ld_blocks(0, gm, gu, p["off"], None, codings=(0, 1, 2))
    with np.errstate(invalid="ignore", divide="ignore"):
        want = np.vstack([onp.pooled_cor(x, gm) for x in (gu, dom, rec)])
    ok = np.isfinite(want)
    assert np.array_equal(ok, np.isfinite(got["b21"]))
    assert np.max(np.abs(got["b21"][ok] - want[ok])) <= 1e-12
    gw = oracle.ld_blocks(1, gm, gu, p["off"], p["w"], codings=(1,))
    with np.errstate(invalid="ignore", divide="ignore"):
        ww = onp.weighted_cor(dom, gm, p["off"], p["w"])
    ok = np.isfinite(ww)
    assert np.max(np.abs(gw["b21"][ok] - ww[ok])) <= 1e-12


def test_ld_per_pop_against_numpy():
    p = small_panel(n_snp=40, scale=0.03, n_pops=4, seed=9)
    G, off = p["G"][:25], p["off"]
    got = oracle.ld_per_pop(G, off)
    iu = np.triu_indices(25, 1)
    for k in range(4):
        with np.errstate(invalid="ignore", divide="ignore"):
            r = np.corrcoef(G[:, off[k]:off[k + 1]].astype(float))[iu]
        ok = np.isfinite(r) & np.isfinite(got[k])
        assert ok.sum() > 50 and np.max(np.abs(got[k][ok] - r[ok])) <= 1e-12


def test_r_quantile_type7_matches_numpy_linear():
    from oracle import feeder_py as fp
    rng = np.random.default_rng(12)
    for n in (1, 2, 7, 100):
        x = rng.random(n)
        for q in (0.0, 0.3, 0.5, 0.99, 1.0):
            assert abs(fp.r_quantile7(x, q) - np.quantile(x, q)) <= 1e-15
