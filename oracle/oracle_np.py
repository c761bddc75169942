"""Independent numpy/scipy (LAPACK) cross-implementation of the oracle -- test infrastructure only.

Written in *Gram form* (matrix products, ``eigh``, ``inv``, ``scipy.stats``), i.e. deliberately
NOT loop-literal, so that agreement with ``gauss_oracle.c`` (loop-literal, hand-written eigen/LU)
checks both the restatement and its dense helpers.  Reference lines: util.cpp:49-70, 103-124,
298-318; dist.cpp:156-202; distmix.cpp:165-228; computeLD.cpp:95-116; gene.cpp:288-550.
"""
import numpy as np
from scipy import stats


def _num(g):
    g = np.asarray(g)
    if g.dtype == np.uint8 and g.size and g.max() >= 48:
        g = g - 48
    return g.astype(np.float64)


def pooled_cor(ga, gb=None):
    """Pearson r over all columns (CalCor util.cpp:49-70), Gram form. Returns (Sa, Sb)."""
    a = _num(ga)
    b = a if gb is None else _num(gb)
    n = a.shape[1]
    sxy = a @ b.T
    sa, sb = a.sum(1), b.sum(1)
    saa, sbb = (a * a).sum(1), (b * b).sum(1)
    numer = n * sxy - np.outer(sa, sb)
    with np.errstate(invalid="ignore", divide="ignore"):
        denor = np.outer(np.sqrt(n * saa - sa * sa), np.sqrt(n * sbb - sb * sb))
        return numer / denor


def weighted_cov(ga, gb, pop_off, w):
    """CalWgtCov (util.cpp:103-124) for all pairs, Gram form."""
    a = _num(ga)
    b = a if gb is None else _num(gb)
    cov = np.zeros((a.shape[0], b.shape[0]))
    wmi = np.zeros(a.shape[0])
    wmj = np.zeros(b.shape[0])
    for p in range(len(pop_off) - 1):
        c0, c1 = pop_off[p], pop_off[p + 1]
        m = c1 - c0
        ap, bp = a[:, c0:c1], b[:, c0:c1]
        sx, sy = ap.sum(1), bp.sum(1)
        sxy = ap @ bp.T
        cov += w[p] * (m / (m - 1.0)) * (m * sxy - np.outer(sx, sy))
        cov += w[p] * np.outer(sx / m, sy / m)
        wmi += w[p] * sx / m
        wmj += w[p] * sy / m
    return cov - np.outer(wmi, wmj)


def weighted_cor(ga, gb, pop_off, w):
    cov = weighted_cov(ga, gb, pop_off, w)
    a = _num(ga)
    b = a if gb is None else _num(gb)
    va = np.array([weighted_cov(a[i:i + 1], None, pop_off, w)[0, 0] for i in range(a.shape[0])])
    vb = va if gb is None else np.array(
        [weighted_cov(b[i:i + 1], None, pop_off, w)[0, 0] for i in range(b.shape[0])])
    with np.errstate(invalid="ignore", divide="ignore"):
        return cov / np.outer(np.sqrt(va), np.sqrt(vb))


def compute_ld(g, pop_off, w):
    r = weighted_cor(g, None, pop_off, w)
    np.fill_diagonal(r, 1.0)
    return r


def make_pos_def(a, min_abs_eig=1e-5):
    vals, vecs = np.linalg.eigh(a)
    if vals.min() < min_abs_eig:
        vals = np.maximum(vals, min_abs_eig)
        return (vecs * vals) @ vecs.T, 1
    return a, 0


def run_impute(mode, gm, gu, pop_off, w, z1, lam=0.1, min_abs_eig=1e-5):
    if mode == 0:
        b11 = pooled_cor(gm)
        b21 = pooled_cor(gu, gm)
    else:
        b11 = weighted_cor(gm, None, pop_off, w)
        b21 = weighted_cor(gu, gm, pop_off, w)
    np.fill_diagonal(b11, 1.0 + lam)
    b11, mpd = make_pos_def(b11, min_abs_eig)
    inv = np.linalg.inv(b11)
    y = b21 @ inv
    z = y @ np.asarray(z1, dtype=np.float64)
    info = np.abs(np.einsum("ij,ij->i", y, b21))
    return dict(z=z / np.sqrt(info), info=info, mpd=mpd, b11=b11, b21=b21)


def run_qcat(mode, gm, gu, pop_off, w, z1, n_head, n_pred, lam=0.1, eig_cutoff=0.01):
    """Independent statement of run_qcat / run_qcatmix (qcat.cpp:166-245): numpy Cholesky + solve."""
    if mode == 0:
        b11 = pooled_cor(gm)
        b21 = pooled_cor(gu, gm) if len(gu) else np.zeros((0, len(gm)))
    else:
        b11 = weighted_cor(gm, None, pop_off, w)
        b21 = weighted_cor(gu, gm, pop_off, w) if len(gu) else np.zeros((0, len(gm)))
    np.fill_diagonal(b11, 1.0 + lam)
    vals = np.linalg.eigvalsh(b11)
    num_eig = len(vals) - (int(np.sum(vals < eig_cutoff)) if vals[0] < eig_cutoff else 0)
    L = np.linalg.cholesky(b11)
    rhs = np.vstack([b11[n_head:n_head + n_pred], b21]).T          # M x (n_pred + U)
    wz = np.linalg.solve(L, np.asarray(z1, dtype=np.float64))
    wb = np.linalg.solve(L, rhs)
    wz = wz - wz.mean()
    wb = wb - wb.mean(0)
    r = (wz @ wb) / np.sqrt((wz @ wz) * np.einsum("ij,ij->j", wb, wb))
    return dict(r=r, num_eig=num_eig, b11=b11, b21=b21)


def count_pc(b11, eig_cutoff=0.01):
    """CountPC (util.cpp:355-388): M minus the number of eigenvalues below the cutoff, counted only when the
    smallest one is below it (LAPACK eigvalsh instead of the C oracle's Jacobi sweep)."""
    vals = np.linalg.eigvalsh(np.asarray(b11, dtype=np.float64))
    return len(vals) - (int(np.sum(vals < eig_cutoff)) if vals[0] < eig_cutoff else 0)


def jepeg_gene_tail(corg, z, info, has, wgt, min_abs_eig=1e-5, categ_cor_cutoff=0.8, denorm_norm_w=3):
    """The k x k tail of one gene, stated from the reference alone (Gene::RunJepeg gene.cpp:88-185 for the category
    bookkeeping, CalJepegPval gene.cpp:317-550, GetW :859-877, GetTopCateg :880-891, GetTopSNP :894-904,
    CnvrtCovToCor util.cpp:284-296), in matrix form with LAPACK / scipy -- written without consulting
    gauss_oracle.c or the product's host tail, so that a shared misreading cannot pass silently.

    corg [n x n] is CorG including the 1 + lambda diagonal; has / wgt [n x 6] say which of the six functional
    categories a SNP carries (Snp::categ_map_) and with what weight.  Returns the fields of the reference's output
    row; for df = 0 the reference leaves chisq / p-values at their constructor value -1 and never sets the names.
    """
    corg = np.asarray(corg, dtype=np.float64)
    z = np.asarray(z, dtype=np.float64)
    info = np.asarray(info, dtype=np.float64)
    has = np.asarray(has).astype(bool)
    wgt = np.asarray(wgt, dtype=np.float64)
    n = len(z)
    # categories present in the gene, in category-number order (categ_count_vec_, categ_vec_: gene.cpp:152-166)
    cats = [c for c in range(6) if has[:, c].any()]
    k = len(cats)
    out = dict(num_snp=n, df=0, chisq=-1.0, jepeg_pval=-1.0, top_categ=-1, top_categ_pval=-1.0, top_snp=-1, top_snp_pval=-1.0)
    if k == 0 or n == 0:
        return out
    # W[c][s] = GetCategWgt(s, c) * sqrt(info_s); a SNP without the category contributes 0 (snp.cpp:37-43)
    W = np.where(has[:, cats], wgt[:, cats], 0.0).T * np.sqrt(info)[None, :]
    WWt = W @ W.T
    CovU = W @ corg @ W.T
    sd = np.sqrt(np.diag(CovU))
    with np.errstate(invalid="ignore", divide="ignore"):
        CorU = CovU / (sd[:, None] * sd[None, :])
        U = W @ z
        pv = 2.0 * stats.norm.sf(np.abs(U / sd))
    rmv = np.zeros(k, dtype=bool)
    for j in range(k - 1, 0, -1):                      # from the last category down to the second (gene.cpp:391-399)
        if np.any(np.abs(CorU[:j, j]) > categ_cor_cutoff):
            rmv[j] = True
    rmv |= np.diag(CovU) < np.diag(WWt) / denorm_norm_w  # low variance (gene.cpp:408-414)
    df = int(k - rmv.sum())
    out["df"] = df
    if df == 0:
        return out
    keep = ~rmv
    X = U[keep]
    CovX = CovU[np.ix_(keep, keep)]
    vals, vecs = np.linalg.eigh(CovX)                   # MakePosDef (util.cpp:302-318)
    if vals.min() < min_abs_eig:
        CovX = (vecs * np.maximum(vals, min_abs_eig)) @ vecs.T
    out["chisq"] = float(X @ np.linalg.inv(CovX) @ X)
    out["jepeg_pval"] = float(stats.chi2.sf(out["chisq"], df))
    top = 0                                             # GetTopCateg starts at index 0 even if that category was removed
    for i in range(k):
        if (pv[top] > pv[i]) and not rmv[i]:
            top = i
    out["top_categ"], out["top_categ_pval"] = cats[top], float(pv[top])
    ts = 0                                              # GetTopSNP: first SNP with the largest |z|
    for i in range(n):
        if abs(z[ts]) < abs(z[i]):
            ts = i
    out["top_snp"], out["top_snp_pval"] = ts, float(2.0 * stats.norm.sf(abs(z[ts])))
    return out


def pnorm_upper(x):
    return stats.norm.sf(x)


def pchisq_upper(x, df):
    return stats.chi2.sf(x, df)
