"""GPU: the five reference entry points end to end (files -> C++ host layer -> HIP -> table) against
the Python restatement of the reference drivers running the CPU oracle on the same files."""
import numpy as np
import pytest

from gauss_amd import api, panel
from oracle import feeder_py as fp

pytestmark = pytest.mark.gpu

POPS = [("AAA", 160, "EUR"), ("BBB", 145, "EUR"), ("CCC", 170, "ASN"), ("DDD", 133, "AFR"), ("EEE", 152, "EUR"),
        ("FFF", 90, "ASN")]
WGT = (["aaa", "CCC", "eee", "FFF", "zzz"], [0.45, 0.2, 0.25, 0.161, 0.3])


@pytest.fixture(scope="module")
def study(tmp_path_factory):
    d = tmp_path_factory.mktemp("study")
    return panel.make_synthetic_study(str(d), POPS, n_snp=700, bp_lo=1_000_000, bp_hi=2_400_000, n_genes=40,
                                      frac_measured=0.3, seed=17)


def _files(st):
    p = st["paths"]
    return p["gwas.txt"], p["index.gz"], p["data.gz"], p["desc.txt"]


def _cmp_impute(df, want, afcol):
    assert list(df.columns) == ["rsid", "chr", "bp", "a1", "a2", afcol, "z", "pval", "info", "type"]
    assert list(df["rsid"]) == want["rsid"] and list(df["bp"]) == want["bp"]
    assert list(df["a1"]) == want["a1"] and list(df["a2"]) == want["a2"] and list(df["type"]) == want["type"]
    assert np.array_equal(df[afcol].to_numpy(), np.array(want["af"]))
    z, wz = df["z"].to_numpy(), np.array(want["z"])
    assert np.max(np.abs(z - wz) / np.maximum(1.0, np.abs(wz))) <= 1e-8
    assert np.max(np.abs(df["info"].to_numpy() - np.array(want["info"])) / np.array(want["info"])) <= 1e-8
    wp = np.array(want["pval"])
    assert np.max(np.abs(df["pval"].to_numpy() - wp) / wp) <= 1e-6
    meas = df["type"].to_numpy() == 1
    assert np.all(df["info"].to_numpy()[meas] == 1.0)              # measured SNPs keep info 1 (gauss.cpp:142)
    assert (df["type"].to_numpy() == 0).sum() == want["n_unmeasured"]


def test_dist_end_to_end(ctx, study):
    args = (22, 1_500_000, 2_000_000, 300_000, "EUR") + _files(study)
    df = api.dist(*args, ctx=ctx)
    _cmp_impute(df, fp.dist(*args), "af1ref")


def test_distmix_end_to_end(ctx, study):
    args = (22, 1_500_000, 2_000_000, 300_000, WGT) + _files(study)
    df = api.distmix(*args, af1_cutoff=0.02, ctx=ctx)
    _cmp_impute(df, fp.distmix(*args, af1_cutoff=0.02), "af1mix")


def _cmp_qcat(df, want, afcol):
    assert list(df.columns) == ["rsid", "chr", "bp", "a1", "a2", afcol, "z", "qcat_m", "qcat_t", "qcat_chisq",
                                "qcat_pval", "type"]
    assert list(df["rsid"]) == want["rsid"] and list(df["bp"]) == want["bp"] and list(df["type"]) == want["type"]
    assert np.array_equal(df[afcol].to_numpy(), np.array(want["af"]))
    assert np.array_equal(df["z"].to_numpy(), np.array(want["z"]))
    assert list(df["qcat_m"]) == want["qcat_m"]
    t, wt = df["qcat_t"].to_numpy(), np.array(want["qcat_t"])
    assert np.max(np.abs(t - wt)) <= 1e-8
    c, wc = df["qcat_chisq"].to_numpy(), np.array(want["qcat_chisq"])
    assert np.max(np.abs(c - wc) / np.maximum(1.0, wc)) <= 1e-8
    pv, wp = df["qcat_pval"].to_numpy(), np.array(want["qcat_pval"])
    assert np.max(np.abs(pv - wp) / wp) <= 1e-6
    # GWAS-only SNPs (type 2) are never tested and keep the constructor's zeros (snp.cpp:26-28)
    t2 = df["type"].to_numpy() == 2
    assert np.all(df["qcat_m"].to_numpy()[t2] == 0) and np.all(pv[t2] == 1.0)
    assert want["n_pred"] > 0 and (df["qcat_m"].to_numpy() > 0).sum() == want["n_pred"] + want["n_unmeasured"]


def test_qcat_end_to_end(ctx, study):
    args = (22, 1_500_000, 2_000_000, 300_000, "EUR") + _files(study)
    _cmp_qcat(api.qcat(*args, ctx=ctx), fp.qcat(*args), "af1ref")


def test_qcatmix_end_to_end(ctx, study):
    args = (22, 1_500_000, 2_000_000, 300_000, WGT) + _files(study)
    _cmp_qcat(api.qcatmix(*args, af1_cutoff=0.02, ctx=ctx), fp.qcatmix(*args, af1_cutoff=0.02), "af1mix")


def _close(a, b, tol=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    nan = np.isnan(b)
    return a.shape == b.shape and np.array_equal(np.isnan(a), nan) and (nan.all() or np.max(np.abs(a[~nan] - b[~nan])) <= tol)


def test_prep_qcat_end_to_end(ctx, study):
    args = (22, 1_500_000, 2_000_000, 300_000, "EUR") + _files(study)
    got, want = api.prep_qcat(*args, ctx=ctx), fp.prep_qcat(*args)
    sl = got["snplist"]
    assert list(sl.columns) == ["rsid", "chr", "bp", "a1", "a2", "af1ref", "z", "type"]
    assert list(sl["rsid"]) == want["rsid"] and list(sl["type"]) == want["type"]      # whole extended window
    assert sl["bp"].min() < 1_500_000 and sl["bp"].max() > 2_000_000
    assert np.array_equal(got["z_vec"], want["z_vec"])
    assert np.all(np.diag(got["cor_mat1"]) == 1.0)
    assert _close(got["cor_mat1"], want["cor_mat1"]) and _close(got["cor_mat2"], want["cor_mat2"])
    # cor_mat2 covers measured AND unmeasured SNPs of the prediction window: a measured one correlates 1 with itself
    assert got["cor_mat2"].shape[0] > (sl["type"] == 0).sum() * 0 + 11 and np.nanmax(got["cor_mat2"]) == pytest.approx(1.0, abs=1e-12)


def test_prep_recessive_impute_end_to_end(ctx, study):
    args = (22, 1_500_000, 2_000_000, 300_000, WGT) + _files(study)
    got, want = api.prep_recessive_impute(*args, ctx=ctx), fp.prep_recessive_impute(*args)
    sl = got["snplist"]
    assert list(sl.columns) == ["rsid", "chr", "bp", "a1", "a2", "af1mix", "z", "type"]
    assert list(sl["rsid"]) == want["rsid"] and list(sl["a1"]) == want["a1"] and list(sl["a2"]) == want["a2"]
    assert np.array_equal(sl["af1mix"].to_numpy(), np.array(want["af"])) and sl["af1mix"].max() <= 0.5   # minor allele
    assert np.array_equal(sl["z"].to_numpy(), np.array(want["z"]))
    assert np.array_equal(got["zvec"], want["zvec"])
    for k in ("cormat", "cormat_add", "cormat_dom", "cormat_rec"):
        assert _close(got[k], want[k]), k
    assert got["cormat_dom"].shape == got["cormat_add"].shape == (len(sl), len(got["zvec"]))
    assert not _close(got["cormat_dom"], got["cormat_add"], 1e-3)


def test_computeLD_end_to_end(ctx, study):
    args = (22, 1_200_000, 2_300_000, WGT) + _files(study)
    got = api.computeLD(*args, ctx=ctx)
    want = fp.computeLD(*args)
    sl = got["snplist"]
    assert list(sl.columns) == ["rsid", "chr", "bp", "a1", "a2", "af1mix"]
    assert list(sl["rsid"]) == want["rsid"] and np.array_equal(sl["af1mix"].to_numpy(), np.array(want["af1mix"]))
    assert got["cormat"].shape == want["cormat"].shape
    assert np.max(np.abs(got["cormat"] - want["cormat"])) <= 1e-12
    assert np.all(np.diag(got["cormat"]) == 1.0)


@pytest.mark.parametrize("mix", [False, True])
def test_jepeg_end_to_end(ctx, study, mix):
    inp, idx, dat, desc = _files(study)
    ann = study["paths"]["annot.txt"]
    if mix:
        df = api.jepegmix(WGT, inp, ann, idx, dat, desc, ctx=ctx)
        want = fp.jepegmix(WGT, inp, ann, idx, dat, desc)
    else:
        df = api.jepeg("EUR", inp, ann, idx, dat, desc, ctx=ctx)
        want = fp.jepeg("EUR", inp, ann, idx, dat, desc)
    assert list(df.columns) == ["geneid", "chisq", "df", "jepeg_pval", "num_snp", "top_categ", "top_categ_pval",
                                "top_snp", "top_snp_pval"]
    assert len(df) == len(want) and len(df) > 10
    names = ["PFS", "TFB", "STR", "TAR", "CIS", "TRN"]
    for i, w in enumerate(want):
        r = df.iloc[i]
        assert r["num_snp"] == w["num_snp"] and r["df"] == w["df"] and r["geneid"] == w["geneid"]
        if w["df"]:
            assert abs(r["chisq"] - w["chisq"]) <= 1e-8 * max(1.0, abs(w["chisq"]))
            assert abs(r["jepeg_pval"] - w["jepeg_pval"]) <= 1e-7 * w["jepeg_pval"] + 1e-300
            assert r["top_categ"] == names[w["top_categ"]]
            assert abs(r["top_categ_pval"] - w["top_categ_pval"]) <= 1e-7 * w["top_categ_pval"] + 1e-300
            assert r["top_snp"] == w["top_snp_id"]
        else:
            assert r["chisq"] == -1.0 and r["jepeg_pval"] == -1.0 and r["top_categ"] == "." and r["top_snp"] == "."


def test_not_enough_snps_is_an_error(ctx, study):
    with pytest.raises(api.GaussError, match="Not enough number of SNPs loaded - DISTMIX not performed"):
        api.distmix(22, 1_000_000, 1_005_000, 0, WGT, *_files(study), ctx=ctx)
    with pytest.raises(api.GaussError, match="Not enough number of SNPs loaded - computeLD not performed"):
        api.computeLD(22, 1_000_000, 1_005_000, WGT, *_files(study), ctx=ctx)


# ---- packed panel: same entry points, reference_data_file = packed file ------------------------------
@pytest.fixture(scope="module")
def packed(study):
    import os
    _, idx, dat, desc = _files(study)
    out = os.path.join(os.path.dirname(dat), "panel.gpk")
    assert api.pack_panel(idx, dat, desc, out) > 0
    return out


def _same_frame(a, b):
    assert list(a.columns) == list(b.columns) and len(a) == len(b)
    for c in a.columns:
        if a[c].dtype.kind == "f":
            assert np.array_equal(a[c].to_numpy(), b[c].to_numpy(), equal_nan=True), c
        else:
            assert list(a[c]) == list(b[c]), c


def test_packed_panel_gives_identical_tables(ctx, study, packed):
    """The packed feeder changes how bytes reach the GPU (2-bit rows gathered by index), not one output bit."""
    inp, idx, dat, desc = _files(study)
    win = (22, 1_500_000, 2_000_000, 300_000)
    for fn, who in ((api.dist, "EUR"), (api.distmix, WGT), (api.qcat, "EUR"), (api.qcatmix, WGT)):
        t = fn(*win, who, inp, idx, dat, desc, ctx=ctx)
        p = fn(*win, who, inp, "(unused)", packed, desc, ctx=ctx)
        _same_frame(t, p)
    t = api.prep_qcat(*win, "EUR", inp, idx, dat, desc, ctx=ctx)
    p = api.prep_qcat(*win, "EUR", inp, idx, packed, desc, ctx=ctx)
    _same_frame(t["snplist"], p["snplist"])
    for k in ("z_vec", "cor_mat1", "cor_mat2"):
        assert np.array_equal(t[k], p[k], equal_nan=True), k
    t = api.prep_recessive_impute(*win, WGT, inp, idx, dat, desc, ctx=ctx)
    p = api.prep_recessive_impute(*win, WGT, inp, idx, packed, desc, ctx=ctx)
    for k in ("zvec", "cormat", "cormat_add", "cormat_dom", "cormat_rec"):
        assert np.array_equal(t[k], p[k], equal_nan=True), k
    t = api.computeLD(22, 1_200_000, 2_300_000, WGT, inp, idx, dat, desc, ctx=ctx)
    p = api.computeLD(22, 1_200_000, 2_300_000, WGT, inp, idx, packed, desc, ctx=ctx)
    _same_frame(t["snplist"], p["snplist"])
    assert np.array_equal(t["cormat"], p["cormat"])
    ann = study["paths"]["annot.txt"]
    _same_frame(api.jepeg("EUR", inp, ann, idx, dat, desc, ctx=ctx), api.jepeg("EUR", inp, ann, idx, packed, desc, ctx=ctx))


@pytest.mark.parametrize("use_packed", [False, True])
def test_prep_zmix5_end_to_end(ctx, study, packed, use_packed):
    """prep_zmix5 (zmix.cpp:44-190): ancestry-informative SNPs by AF variance, then per-population LD of all pairs."""
    inp, idx, dat, desc = _files(study)
    want = fp.prep_zmix5(inp, idx, dat, desc, percentile=0.8, interval=2)
    got, snps = api.prep_zmix5(inp, idx, packed if use_packed else dat, desc, percentile=0.8, interval=2, ctx=ctx,
                               with_snps=True)
    assert list(snps["rsid"]) == want["rsid"] and len(want["rsid"]) > 10
    assert np.array_equal(snps["norm_var"].to_numpy(), np.array(want["norm_var"]))
    assert got.shape == want["data_mat"].shape == (len(want["rsid"]) * (len(want["rsid"]) - 1) // 2, 1 + len(POPS))
    assert np.array_equal(got[:, 0], want["data_mat"][:, 0])
    nan = np.isnan(want["data_mat"])
    assert np.array_equal(np.isnan(got), nan)
    assert np.max(np.abs(got[~nan] - want["data_mat"][~nan])) <= 1e-12


@pytest.mark.parametrize("use_packed", [False, True])
@pytest.mark.parametrize("variant,kw", [("zmix", dict(interval=7)), ("zmix2", dict(interval=5, p2=3)), ("zmix2", dict(interval=2, p2=4)),
                                        ("zmix3", dict(interval=3, p2=4)), ("zmix4", dict(interval=6, p2=2)),
                                        ("zmix5_sup", dict(percentile=0.7, interval=2))])
def test_prep_zmix_selectors_end_to_end(ctx, study, packed, use_packed, variant, kw):
    """prep_zmix / prep_zmix2 / prep_zmix3 / prep_zmix4 / prep_zmix5_sup (zmix.cpp:201-1076): the pairs each selector lists,
    z_i * z_j, and the pair's genotype correlation per population (per super-population for _sup) against the loop-literal
    restatement; only the tile pairs the listed pairs touch are multiplied on the GPU (gauss_ld_per_pop_pairs)."""
    inp, idx, dat, desc = _files(study)
    want = fp.prep_zmix_variant(variant, inp, idx, dat, desc, **kw)
    got = api.prep_zmix_variant(variant, inp, idx, packed if use_packed else dat, desc, ctx=ctx, **kw)
    assert len(want["pairs"]) > 10
    rs = list(got["snps"]["rsid"])
    assert [(rs[i], rs[j]) for i, j in got["pairs"]] == want["pairs"]
    assert got["groups"] == want["groups"]
    g, w = got["data_mat"], want["data_mat"]
    assert g.shape == w.shape
    lead = 2 if variant == "zmix4" else 1
    assert np.array_equal(g[:, :lead], w[:, :lead])                  # h (zmix4) and z_i * z_j: exact
    nan = np.isnan(w)
    assert np.array_equal(np.isnan(g), nan)
    assert np.max(np.abs(g[~nan] - w[~nan])) <= 1e-12
    if variant == "zmix5_sup":
        assert len(want["groups"]) < len(POPS) and "norm_var" in got["snps"]


@pytest.mark.parametrize("seed", range(6))
def test_gene_drivers_annotated_positions_only_equal_the_whole_study(ctx, tmp_path, monkeypatch, seed):
    """jepeg() / jepegmix() enter only the study SNPs at positions the annotation names (plus the positions the study lists more than
    once or under equal alleles) into their SNP map (host_calls.cpp:run_jepeg); GAUSS_HOST_FULL_MAP=1 enters the whole study, as
    the reference does (gauss.cpp:121-190).  Same gene table, bit for bit, text panel and packed -- on a plain synthetic study and
    on studies made of odd sites (tests/test_feeder.py:_odd_study: repeated, swapped, multi-allelic sites; a study that trips the
    reference's duplicate check must trip it either way, also at a position no gene names)."""
    from test_feeder import _odd_study
    rng = np.random.default_rng(700 + seed)
    if seed < 2:
        pops = [("P00", 60, "EUR"), ("P01", 45, "ASN"), ("P02", 52, "EUR")]
        st = panel.make_synthetic_study(str(tmp_path), pops, n_snp=500, bp_lo=1_000_000, bp_hi=2_000_000, frac_measured=0.5,
                                        frac_swapped=0.3, frac_not_in_panel=0.05, n_genes=30, seed=40 + seed)
        p = st["paths"]
        inp, idx, dat, desc, ann = p["gwas.txt"], p["index.gz"], p["data.gz"], p["desc.txt"], p["annot.txt"]
        gpk = str(tmp_path / "f.gpk")
        assert api.pack_panel(idx, dat, desc, gpk) > 0
        wgt = (["p00", "P01", "p02"], [0.5, 0.2, 0.3])
    else:
        st = _odd_study(str(tmp_path), 900 + seed, n_sites=300 if seed % 2 else None)
        inp, idx, dat, desc, gpk = st["gwas"], st["idx"], st["dat"], st["desc"], st["gpk"]
        # an annotation over the study's own rows: some under the study's allele order, some under the other one, some sites twice,
        # a few positions the study does not have; about half of the study's positions stay unnamed
        rows = [l.split() for l in open(inp).read().splitlines()[1:]]
        ann_rows = []
        cats = ["PROTEIN", "TFBS", "WTH_HAIR", "WTH_TARGET", "CIS_EQTL", "TRANS_EQTL", "NO_SUCH"]
        for r in rows:
            if r[1] != "22" or rng.random() < 0.5:
                continue
            a1, a2 = (r[3], r[4]) if rng.random() < 0.7 else (r[4], r[3])
            for _ in range(int(rng.integers(1, 3))):
                ann_rows.append(("x", 22, int(r[2]), a1, a2, f"GENE{int(r[2]) // 4000:03d}", str(rng.choice(cats)), float(np.round(rng.uniform(0.2, 2.0), 3))))
        ann_rows += [("x", 22, 999_999, "A", "C", "GENE999", "TFBS", 1.0), ("x", 21, 5000, "A", "C", "GENE998", "TFBS", 1.0)]
        ann = str(tmp_path / "annot.txt")
        panel.write_annotation(ann, ann_rows)
        wgt = (["AAA", "BBB", "ccc"], [0.5, 0.3, 0.2])
    for data in (dat, gpk):
        for call, who in ((api.jepeg, "EUR"), (api.jepegmix, wgt)):
            res = []
            for full in ("0", "1"):
                monkeypatch.setenv("GAUSS_HOST_FULL_MAP", full)
                try:
                    res.append(call(who, inp, ann, idx, data, desc, af1_cutoff=0.0001, ctx=ctx))
                except api.GaussError as e:
                    res.append(str(e))
            if isinstance(res[1], str):
                assert res[0] == res[1] and "duplicates" in res[1]
                print("seed %d %s %s: both fail (%s)" % (seed, call.__name__, "packed" if data == gpk else "text", res[1][:40]))
            else:
                assert not isinstance(res[0], str), res[0]
                _same_frame(res[0], res[1])
                assert len(res[0]) >= 1
                print("seed %d %s %s: %d genes, %d with df > 0" % (seed, call.__name__, "packed" if data == gpk else "text", len(res[0]), int((res[0]["df"] > 0).sum())))


@pytest.mark.parametrize("seed", range(6))
def test_window_calls_and_chromosome_driver_equal_the_literal_data_layer(ctx, tmp_path, monkeypatch, seed):
    """dist() / distmix() / qcat() / qcatmix() on a sorted packed panel build their window as a merge of the study's rows and the
    panel's SNP table and read its genotype rows from the panel's resident copy (host_calls.cpp:run_impute), and so does the
    chromosome driver; GAUSS_HOST_FULL_MAP=1 takes the literal path -- the reference's per-SNP objects in a map (gauss_host_prepare),
    the window's rows gathered on the host.  Same tables bit for bit and the same errors, on a plain study and on studies made of
    odd sites (repeated, swapped, multi-allelic, duplicate-error sites)."""
    from test_feeder import _odd_study
    rng = np.random.default_rng(1700 + seed)
    if seed < 2:
        pops = [("P00", 60, "EUR"), ("P01", 45, "ASN"), ("P02", 52, "EUR")]
        st = panel.make_synthetic_study(str(tmp_path), pops, n_snp=900, bp_lo=1_000_000, bp_hi=2_000_000, frac_measured=0.4,
                                        frac_swapped=0.3, frac_not_in_panel=0.05, seed=60 + seed)
        p = st["paths"]
        inp, idx, dat, desc = p["gwas.txt"], p["index.gz"], p["data.gz"], p["desc.txt"]
        gpk = str(tmp_path / "f.gpk")
        assert api.pack_panel(idx, dat, desc, gpk) > 0
        wgt, span, wsize = (["p00", "P01", "p02"], [0.5, 0.2, 0.3]), (1_000_001, 2_000_000), 250_000
    else:
        st = _odd_study(str(tmp_path), 1900 + seed, n_sites=400)
        inp, idx, desc, gpk = st["gwas"], st["idx"], st["desc"], st["gpk"]
        wgt, span, wsize = (["AAA", "BBB", "ccc"], [0.5, 0.3, 0.2]), (1_000, 60_000), 15_000
    wing = int(rng.choice([0, wsize // 3]))
    calls = {"dist": (api.dist, "EUR"), "distmix": (api.distmix, wgt), "qcat": (api.qcat, "EUR"), "qcatmix": (api.qcatmix, wgt)}
    name = list(calls)[seed % 4]
    fn, who = calls[name]
    n_ok = 0
    for s0 in range(span[0], span[1], wsize):
        res = []
        for full in ("0", "1"):
            monkeypatch.setenv("GAUSS_HOST_FULL_MAP", full)
            try:
                res.append(fn(22, s0, min(span[1], s0 + wsize - 1), wing, who, inp, idx, gpk, desc, af1_cutoff=0.01, ctx=ctx))
            except api.GaussError as e:
                res.append(str(e))
        if isinstance(res[1], str):
            assert res[0] == res[1], res
        else:
            assert not isinstance(res[0], str), res[0]
            _same_frame(res[0], res[1])
            n_ok += 1
    assert n_ok >= 1
    # computeLD() on the same windows (measured SNPs only: computeLD.cpp:80-86)
    n_ld = 0
    for s0 in range(span[0], span[1], 2 * wsize):
        res = []
        for full in ("0", "1"):
            monkeypatch.setenv("GAUSS_HOST_FULL_MAP", full)
            try:
                res.append(api.computeLD(22, s0, min(span[1], s0 + 2 * wsize - 1), wgt, inp, idx, gpk, desc, af1_cutoff=0.01, ctx=ctx))
            except api.GaussError as e:
                res.append(str(e))
        if isinstance(res[1], str):
            assert res[0] == res[1], res
        else:
            assert not isinstance(res[0], str), res[0]
            _same_frame(res[0]["snplist"], res[1]["snplist"])
            assert np.array_equal(res[0]["cormat"], res[1]["cormat"])
            n_ld += 1
    assert n_ld >= 1
    kind = getattr(api, "KIND_" + name.upper())
    sel = dict(pop_wgt_df=who) if name.endswith("mix") else dict(study_pop="EUR")
    tabs = []
    for full in ("0", "1"):
        monkeypatch.setenv("GAUSS_HOST_FULL_MAP", full)
        tabs.append(api.impute_chromosome(kind, 22, span[0], span[1], wing, input_file=inp, reference_data_file=gpk, reference_pop_desc_file=desc,
                                          window_size=wsize, af1_cutoff=0.01, ctx=ctx, **sel))
    assert np.array_equal(tabs[0].windows, tabs[1].windows) and tabs[0].messages == tabs[1].messages
    for c in tabs[0].columns:
        x, y = tabs[0].columns[c], tabs[1].columns[c]
        assert np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y), c
