"""Known answers printed by the reference's vignettes (tests/golden/vignette_known_answers.json).

The reference holds no tests; its rendered vignettes are the only numbers it ever published for this path.  Most
of them need the external 33KG panel (replayed only where GAUSS_33KG_DIR points at it), but the single-SNP genes of
the jepeg() / jepegmix() tables do not: for one SNP in one category

    chisq = (w z sqrt(info))^2 / (w^2 info (1 + lambda)) = z^2 / 1.1

whatever genotypes the panel holds, and z is in the reference's own study file.  Those rows therefore pin -- against
numbers the reference itself printed -- lambda = 0.1 on CorG's diagonal (gene.cpp:307), the W scaling (gene.cpp:871),
the chi-square and normal tails (R::pchisq / R::pnorm5, gene.cpp:376,509,522), df, top-SNP selection, and, on the
GPU, the whole product path for 1 x 1 gene blocks (feeder -> gauss_gene_ld_batch -> host tail).
"""
import json
import os

import numpy as np
import pytest

import oracle
from oracle import oracle_np as onp
from gauss_amd import api, panel, synth

HERE = os.path.dirname(__file__)
KA = json.load(open(os.path.join(HERE, "golden", "vignette_known_answers.json")))
STUDY = os.path.join(HERE, "golden", "PGC2_Chr22_ilmn1M_Z.txt")
CATEG = {"PFS": ("PROTEIN", 0), "TFB": ("TFBS", 1), "STR": ("WTH_HAIR", 2), "TAR": ("WTH_TARGET", 3), "CIS": ("CIS_EQTL", 4),
         "TRN": ("TRANS_EQTL", 5)}


def study_rows():
    rows = {}
    with open(STUDY) as f:
        f.readline()
        for line in f:
            t = line.split()
            rows[t[0]] = t
    return rows


def printed(value, shown, decimals):
    """`shown` is `value` as the vignette printed it (kable: rounded to `decimals` places)."""
    return abs(value - shown) <= 0.5 * 10.0 ** (-decimals) * 1.02


def single_snp_rows():
    out = []
    for fn in ("jepeg", "jepegmix"):
        for row in KA[fn]["head"]:
            if row["num_snp"] == 1:
                out.append((fn, row))
    return out


@pytest.mark.parametrize("fn,row", single_snp_rows(), ids=lambda v: v if isinstance(v, str) else v["geneid"])
def test_single_snp_genes_of_the_vignette_pin_the_jepeg_tail(fn, row):
    """No GPU: the three statements of the k x k tail (C oracle, independent numpy oracle, the product's host tail)
    reproduce the reference's printed chisq / df / p-values from the reference's own z-score."""
    z = float(study_rows()[row["top_snp"]][5])
    cat = CATEG[row["top_categ"]][1]
    corg = np.array([[1.1]])                         # CorG(i, i) = 1.0 + lambda_ (gene.cpp:307)
    has = np.zeros((1, 6), dtype=np.int32)
    has[0, cat] = 1
    for w in (1.0, 0.37):                            # the category weight cancels
        wgt = np.zeros((1, 6))
        wgt[0, cat] = w
        for tail in (oracle.jepeg_gene_tail, onp.jepeg_gene_tail, api.jepeg_gene_tail):
            r = tail(corg, [z], [1.0], has, wgt)
            assert r["df"] == row["df"] == 1 and r["num_snp"] == 1
            assert printed(r["chisq"], row["chisq"], 5), (tail.__module__, r["chisq"], row["chisq"])
            assert printed(r["jepeg_pval"], row["jepeg_pval"], 7)
            assert printed(r["top_categ_pval"], row["top_categ_pval"], 7)
            assert printed(r["top_snp_pval"], row["top_snp_pval"], 7)
            assert r["top_categ"] == cat and r["top_snp"] == 0


def test_printed_z_pval_pairs_of_the_dist_vignettes_pin_the_normal_tail():
    """dist() / distmix() print z and pval = 2 * pnorm(|z|, lower = FALSE) side by side (dist.cpp:101,
    distmix.cpp:110; dist_example.md:163-170,267-274): twelve (z, pval) pairs that need no panel.  The C oracle's
    tail, scipy's, and the product's host tail (reached through a one-SNP gene, whose top_snp_pval is that very
    expression, gene.cpp:522) must all print as the reference did."""
    from scipy.stats import norm
    n = 0
    for fn in ("dist", "distmix"):
        for row in KA[fn]["head"]:
            z, shown = row["z"], row["pval"]
            got = {"oracle": 2 * oracle.pnorm_upper(abs(z)), "scipy": 2 * norm.sf(abs(z)),
                   "product": api.jepeg_gene_tail(np.array([[1.1]]), [z], [1.0], np.array([[1, 0, 0, 0, 0, 0]], dtype=np.int32),
                                                  np.array([[1.0, 0, 0, 0, 0, 0]]))["top_snp_pval"]}
            # both columns are rounded to 7 decimals: |dp| <= 0.5e-7 (pval) + 2 phi(z) * 0.5e-7 (z)
            tol = 0.51e-7 * (1.0 + 2.0 * norm.pdf(z))
            for who, p in got.items():
                assert abs(p - shown) <= tol, (fn, row["rsid"], who, p, shown)
            n += 1
    assert n == 12


@pytest.mark.gpu
@pytest.mark.parametrize("fn", ["jepeg", "jepegmix"])
def test_gpu_product_reproduces_the_vignette_rows_of_single_snp_genes(ctx, tmp_path, fn):
    """The whole product call -- files -> feeder -> gene LD on the GPU -> host tail -> table -- on a synthetic panel
    that carries the vignette's SNPs: the rows of the single-SNP genes must read as the reference printed them
    (they do not depend on the genotypes), next to multi-SNP genes that exercise the batched LD."""
    rows = study_rows()
    want = [r for r in KA[fn]["head"] if r["num_snp"] == 1]
    rng = np.random.default_rng(4)
    keep = {r["top_snp"] for r in want}
    others = [k for k in rows if k not in keep]
    extra = [others[i] for i in rng.choice(len(others), size=180, replace=False)]
    gwas = tmp_path / "gwas.txt"
    gwas.write_text("rsid chr bp a1 a2 z\n" + "\n".join(" ".join(rows[k]) for k in list(keep) + extra) + "\n")
    pops = synth.pop_table(scale=0.04, min_size=40)           # all 29 populations, ~1 500 samples
    cfg = panel.make_panel_for_gwas(str(tmp_path), pops, str(gwas), n_extra=100, frac_swapped=0.3, seed=11)
    p = cfg["paths"]
    ann = []
    for r in want:
        t = rows[r["top_snp"]]
        ann.append((t[0], int(t[1]), int(t[2]), t[3], t[4], r["geneid"], CATEG[r["top_categ"]][0], 0.8))
    for g in range(12):                                       # multi-SNP genes around them
        for k in rng.choice(len(extra), size=int(rng.integers(2, 7)), replace=False):
            t = rows[extra[k]]
            ann.append((t[0], int(t[1]), int(t[2]), t[3], t[4], f"MULTI{g:02d}", panel.CATEGS[int(rng.integers(0, 6))],
                        float(np.round(rng.uniform(0.2, 2.0), 3))))
    annot = tmp_path / "annot.txt"
    panel.write_annotation(str(annot), ann)
    files = dict(input_file=str(gwas), annotation_file=str(annot), reference_index_file=p["index.gz"],
                 reference_data_file=p["data.gz"], reference_pop_desc_file=p["desc.txt"])
    if fn == "jepeg":
        tab = api.jepeg("GBR", af1_cutoff=0.0, ctx=ctx, **files)
    else:
        wgt = (list(synth.PGC2_WEIGHTS.keys()), list(synth.PGC2_WEIGHTS.values()))
        tab = api.jepegmix(wgt, af1_cutoff=0.0, ctx=ctx, **files)
    got = {r["geneid"]: r for r in tab.to_dict("records")}
    assert sum(g.startswith("MULTI") for g in got) >= 8
    for r in want:
        g = got[r["geneid"]]
        assert g["df"] == 1 and g["num_snp"] == 1 and g["top_snp"] == r["top_snp"] and g["top_categ"] == r["top_categ"]
        assert printed(g["chisq"], r["chisq"], 5), (g["chisq"], r["chisq"])
        assert printed(g["jepeg_pval"], r["jepeg_pval"], 7) and printed(g["top_snp_pval"], r["top_snp_pval"], 7)
        assert printed(g["top_categ_pval"], r["top_categ_pval"], 7)


# ---- deferred: the tables that need the real 33KG panel ---------------------------------------------------------
KG = os.environ.get("GAUSS_33KG_DIR")
needs_33kg = pytest.mark.skipif(not KG, reason="set GAUSS_33KG_DIR to the 33KG panel (33kg_index.gz, 33kg_geno.gz, "
                                               "33kg_pop_desc.txt; docs/articles/ref_33KG.md:7) to replay the vignette tables")


def _kg_files(study):
    return dict(input_file=study, reference_index_file=os.path.join(KG, "33kg_index.gz"),
                reference_data_file=os.path.join(KG, "33kg_geno.gz"), reference_pop_desc_file=os.path.join(KG, "33kg_pop_desc.txt"))


@pytest.mark.gpu
@needs_33kg
def test_gpu_vignette_computeLD_on_33KG(ctx):
    wgt = (list(synth.PGC2_WEIGHTS.keys()), list(synth.PGC2_WEIGHTS.values()))
    c = KA["computeLD"]["call"]
    res = api.computeLD(c["chr"], c["start_bp"], c["end_bp"], wgt, af1_cutoff=c["af1_cutoff"], ctx=ctx,
                        **_kg_files(os.path.join(HERE, "golden", "PGC2_3Mb.txt")))
    sl = res["snplist"]
    for k, row in enumerate(KA["computeLD"]["snplist_head"]):
        assert sl["rsid"][k] == row["rsid"] and sl["bp"][k] == row["bp"] and printed(sl["af1mix"][k], row["af1mix"], 7)
    assert np.max(np.abs(res["cormat"][:3, :3] - np.array(KA["computeLD"]["cormat_3x3"]))) <= 6e-8


@pytest.mark.gpu
@needs_33kg
@pytest.mark.parametrize("fn", ["dist", "distmix"])
def test_gpu_vignette_dist_and_distmix_on_33KG(ctx, fn):
    c = KA[fn]["call"]
    files = _kg_files(os.path.join(HERE, "golden", "PGC2_3Mb.txt"))
    if fn == "dist":
        tab = api.dist(c["chr"], c["start_bp"], c["end_bp"], c["wing_size"], c["study_pop"], af1_cutoff=c["af1_cutoff"], ctx=ctx, **files)
        afcol = "af1ref"
    else:
        wgt = (list(synth.PGC2_WEIGHTS.keys()), list(synth.PGC2_WEIGHTS.values()))
        tab = api.distmix(c["chr"], c["start_bp"], c["end_bp"], c["wing_size"], wgt, af1_cutoff=c["af1_cutoff"], ctx=ctx, **files)
        afcol = "af1mix"
    for k, row in enumerate(KA[fn]["head"]):
        assert tab["rsid"][k] == row["rsid"] and tab["bp"][k] == row["bp"] and tab["type"][k] == row["type"]
        assert printed(tab[afcol][k], row[afcol], 7 if fn == "distmix" else 5)
        # the north star's bar: imputed z within 1e-5 relative of the reference's Eigen path (as printed: 7 decimals)
        assert abs(tab["z"][k] - row["z"]) <= 1e-5 * max(1.0, abs(row["z"])) + 5.1e-8
        assert abs(tab["info"][k] - row["info"]) <= 1e-5 + 5.1e-8
