"""GPU edge cases: tile / block / panel boundaries, tiny and ragged inputs, many populations, large
genotype codes, error reporting, and size-independent properties at the BASELINE sizes."""
import os

import numpy as np
import pytest

import oracle
from gauss_amd import hotpath, synth
from gauss_amd import panel as panel_mod
from helpers import relerr, small_panel

pytestmark = pytest.mark.gpu


def rand_geno(rng, rows, n, lo=0.05, hi=0.95):
    af = rng.uniform(lo, hi, size=(rows, 1))
    g = (rng.random((rows, n)) < af).astype(np.uint8) + (rng.random((rows, n)) < af).astype(np.uint8)
    for r in range(rows):                      # no monomorphic rows
        if g[r].min() == g[r].max():
            g[r, 0], g[r, 1] = 0, 2
    return g


def check_window(ctx, mode, gm, gu, off, w, z1, tol=1e-8):
    got = hotpath.impute_window(mode, gm, gu, off, w, z1, want_mats=True, ctx=ctx)
    want = oracle.run_impute(mode, gm, gu, off, w, z1, want_mats=True)
    assert got["status"] == 0 and want["mpd"] == 0
    assert np.max(np.abs(got["b11"] - want["b11"])) <= 1e-12
    assert np.max(np.abs(got["b21"] - want["b21"])) <= 1e-12
    assert relerr(got["info"], want["info"]) <= tol
    assert np.max(np.abs(got["z"] - want["z"]) / np.maximum(1.0, np.abs(want["z"]))) <= tol


@pytest.mark.parametrize("M,U", [(1, 1), (2, 3), (63, 62), (64, 63), (65, 64), (127, 126), (128, 127), (129, 128),
                                 (192, 130), (257, 5)])
def test_tile_block_and_panel_boundaries(ctx, M, U):
    rng = np.random.default_rng(M * 1000 + U)
    N = 333                                     # not a multiple of 16 or 64
    off = np.array([0, 100, 101 + 30, 333], dtype=np.int32)
    w = np.array([0.3, 0.5, 0.261])
    G = rand_geno(rng, M + U, N)
    z1 = rng.standard_normal(M)
    for mode in (0, 1):
        check_window(ctx, mode, G[:M], G[M:], off, w, z1)


def test_tiny_populations_and_row_stride(ctx):
    rng = np.random.default_rng(3)
    sizes = [2, 3, 2, 17, 2]
    off = synth.pop_offsets(sizes)
    N = int(off[-1])
    buf = np.zeros((40, N + 7), dtype=np.uint8)          # ld > N
    buf[:, :N] = rand_geno(rng, 40, N, 0.2, 0.8)
    gm, gu = buf[:15, :N], buf[15:, :N]                   # non-contiguous views keep the parent stride
    w = np.array([0.2, 0.2, 0.2, 0.3, 0.161])
    z1 = rng.standard_normal(15)
    got = hotpath.impute_window(1, np.ascontiguousarray(gm), np.ascontiguousarray(gu), off, w, z1, want_mats=True, ctx=ctx)
    want = oracle.run_impute(1, gm, gu, off, w, z1, want_mats=True)
    ok = np.isfinite(want["b11"])
    assert np.array_equal(np.isfinite(got["b11"]), ok)
    assert np.max(np.abs(got["b11"][ok] - want["b11"][ok])) <= 1e-12


@pytest.mark.parametrize("P", [33, 64])
def test_many_populations_use_the_global_table_path(ctx, P):
    rng = np.random.default_rng(P)
    sizes = rng.integers(20, 60, size=P)
    off = synth.pop_offsets(sizes)
    w = rng.uniform(0.5, 1.5, P)
    w *= 1.061 / w.sum()
    G = rand_geno(rng, 70, int(off[-1]))
    got = hotpath.ld_matrix(G, off, w, ctx=ctx)
    want = oracle.compute_ld(G, off, w)
    assert np.max(np.abs(got - want)) <= 1e-12
    check_window(ctx, 1, G[:30], G[30:], off, w, rng.standard_normal(30))


def test_codes_up_to_15_and_ascii_digits(ctx):
    rng = np.random.default_rng(9)
    G = rng.integers(0, 16, size=(37, 500)).astype(np.uint8)
    want = G.astype(np.int64) @ G.astype(np.int64).T
    assert np.array_equal(hotpath.gram_counts(G, ctx=ctx), want)
    D = rng.integers(0, 10, size=(20, 300)).astype(np.uint8)
    assert np.array_equal(hotpath.gram_counts(D + ord("0"), ctx=ctx), D.astype(np.int64) @ D.astype(np.int64).T)


def test_ld_only_shapes(ctx):
    rng = np.random.default_rng(4)
    off = np.array([0, 150, 400], dtype=np.int32)
    w = np.array([0.4, 0.661])
    for S in (1, 2, 127, 128, 129, 300):
        G = rand_geno(rng, S, 400)
        got = hotpath.ld_matrix(G, off, w, ctx=ctx)
        want = oracle.compute_ld(G, off, w)
        assert got.shape == (S, S) and np.max(np.abs(got - want)) <= 1e-12


def test_gene_batch_sizes_one_to_many_tiles(ctx):
    rng = np.random.default_rng(6)
    off = np.array([0, 200, 520], dtype=np.int32)
    sizes = [1, 0, 3, 140, 1, 1, 260, 2]                   # empty gene, genes larger than one / two tiles
    gene_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    G = rand_geno(rng, int(gene_off[-1]), 520)
    blocks = hotpath.gene_ld_batch(G, off, gene_off, mode=0, diag=1.1, ctx=ctx)
    for g, n in enumerate(sizes):
        assert blocks[g].shape == (n, n)
        if n:
            want = oracle.ld_pooled(G[gene_off[g]:gene_off[g + 1]], off, 1.1)
            assert np.max(np.abs(blocks[g] - want)) <= 1e-12


def test_clamped_window_inside_a_batch(ctx):
    rng = np.random.default_rng(12)
    off = np.array([0, 300], dtype=np.int32)
    wins, wants = [], []
    for k in range(3):
        G = rand_geno(rng, 60, 300)
        gm, gu = G[:25], G[25:]
        lam = 0.1
        if k == 1:                                         # duplicated SNPs + lambda 0: singular B11 -> MakePosDef acts
            gm = np.vstack([gm, gm[:2]])
            lam = 0.0
        z1 = rng.standard_normal(gm.shape[0])
        wins.append(dict(mode=0, geno_m=np.ascontiguousarray(gm), geno_u=np.ascontiguousarray(gu), pop_off=off, z1=z1, lam=lam))
        wants.append(oracle.run_impute(0, gm, gu, off, None, z1, lam=lam, want_mats=True))
    job = hotpath.Job(wins, ctx=ctx)
    job.run()
    res = job.fetch()
    job.close()
    assert [r["status"] for r in res] == [0, 1, 0] and [w["mpd"] for w in wants] == [0, 1, 0]
    for r, w in zip(res, wants):
        assert relerr(r["info"], w["info"]) <= 1e-5
        assert np.max(np.abs(r["z"] - w["z"]) / np.maximum(1.0, np.abs(w["z"]))) <= 1e-5
    # the same batch with B11 / B21 wanted back: the matrices travel through the job's export mirror (gauss_run.cpp: queue_exports),
    # the clamped window's B11 -- rewritten by MakePosDef's repair inside the fetch -- is fetched again after the repair; and with a
    # second run of the job already queued the fetch of the first still returns the same matrices
    job = hotpath.Job(wins, ctx=ctx, want_mats=True)
    job.run()
    job.run()
    first = [dict(b11=r["b11"].copy(), b21=r["b21"].copy(), z=r["z"].copy()) for r in job.fetch()]
    second = job.fetch()
    job.close()
    for a, b, r, w in zip(first, second, res, wants):
        assert np.array_equal(a["b11"], b["b11"]) and np.array_equal(a["b21"], b["b21"]) and np.array_equal(a["z"], r["z"])
        assert np.max(np.abs(a["b11"] - w["b11"])) <= (1e-9 if w["mpd"] else 1e-12)
        assert np.max(np.abs(a["b21"] - w["b21"])) <= 1e-12


def test_bad_arguments(ctx):
    rng = np.random.default_rng(1)
    G = rand_geno(rng, 12, 100)
    with pytest.raises(Exception, match="n_pop"):
        hotpath.ld_matrix(G, np.arange(0, 66, dtype=np.int32), np.ones(65), ctx=ctx)
    with pytest.raises(Exception, match="pop_off"):
        hotpath.ld_matrix(G, np.array([5, 100], dtype=np.int32), np.ones(1), ctx=ctx)
    with pytest.raises(Exception, match="ld"):
        hotpath.ld_matrix(G, np.array([0, 200], dtype=np.int32), np.ones(1), ctx=ctx)
    with pytest.raises(Exception, match="unmeasured"):
        hotpath.Job([dict(mode=0, geno_m=G, geno_u=G[:0], pop_off=[0, 100], z1=np.zeros(12))], ctx=ctx)


def test_full_size_properties_at_baseline_shape(ctx):
    """DISTMIX at the BASELINE shape (21 PGC2 populations, N = 32 147, M ~ 740, U ~ 2 400): properties that
    do not need the oracle at full size, plus an oracle spot check of individual LD entries."""
    pops = [p for p in synth.POPS_33KG if p[0] in synth.PGC2_WEIGHTS]
    off = synth.pop_offsets([p[1] for p in pops])
    w = np.array([synth.PGC2_WEIGHTS[p[0]] for p in pops])
    N = int(off[-1])
    assert N == 32147
    rng = np.random.default_rng(77)
    M, U = 737, 2407
    base = rand_geno(rng, 160, N)                          # LD structure: rows are noisy copies of a few bases
    src = rng.integers(0, 160, size=M + U)
    G = base[src].copy()
    noise = rng.random(G.shape) < 0.35
    G[noise] = rand_geno(rng, 1, N)[0][np.nonzero(noise)[1]]
    gm, gu = np.ascontiguousarray(G[:M]), np.ascontiguousarray(G[M:])
    z1 = rng.standard_normal(M) * 2
    r = hotpath.impute_window(1, gm, gu, off, w, z1, want_mats=True, ctx=ctx)
    assert r["status"] == 0
    b11, b21 = r["b11"], r["b21"]
    assert np.array_equal(b11, b11.T) and np.all(np.diag(b11) == 1.1)
    assert np.all(np.abs(b21) <= 1 + 1e-9) and np.all(np.isfinite(r["z"]))
    assert np.all(r["info"] > 0) and np.all(r["info"] < 1 + 1e-9)
    assert np.linalg.eigvalsh(b11).min() > 0.09            # R is PSD, B11 = R + 0.1 I
    # integer Gram diagonal = sum x^2, exactly
    cnt = hotpath.gram_counts(gm[:200], ctx=ctx)
    assert np.array_equal(np.diag(cnt), (gm[:200].astype(np.int64) ** 2).sum(1))
    # spot-check entries against the loop-literal oracle pair function
    def std(row):
        return np.sqrt(oracle.calwgtcov(row, row, off, w))
    for (i, j) in [(0, 1), (5, 700), (300, 301), (736, 2)]:
        want = oracle.calwgtcov(gm[i], gm[j], off, w) / (std(gm[i]) * std(gm[j]))
        assert abs(b11[i, j] - want) <= 1e-13
    for (u, j) in [(0, 0), (2406, 736), (1200, 17)]:
        want = oracle.calwgtcov(gu[u], gm[j], off, w) / (std(gu[u]) * std(gm[j]))
        assert abs(b21[u, j] - want) <= 1e-13
    # the solve agrees with numpy's inverse on the GPU's own B11 / B21
    y = b21 @ np.linalg.inv(b11)
    info = np.abs(np.einsum("ij,ij->i", y, b21))
    assert relerr(r["info"], info) <= 1e-9
    assert np.max(np.abs(r["z"] - (y @ z1) / np.sqrt(info))) <= 1e-8
    # recoding an unmeasured SNP (0 <-> 2) flips its z and keeps its info.  Exact for the pooled Pearson
    # estimator only: CalWgtCov with weights that do not sum to 1 (PGC2: 1.061, quirk Q2) is not
    # antisymmetric under x -> 2 - x, in the reference as here.
    p0 = hotpath.impute_window(0, gm, gu, off, None, z1, ctx=ctx)
    gu2 = gu.copy()
    gu2[10] = 2 - gu2[10]
    p1 = hotpath.impute_window(0, gm, gu2, off, None, z1, ctx=ctx)
    assert abs(p1["z"][10] + p0["z"][10]) <= 1e-9 and abs(p1["info"][10] - p0["info"][10]) <= 1e-12
    assert np.max(np.abs(np.delete(p1["z"], 10) - np.delete(p0["z"], 10))) == 0.0
    # permuting populations (with their weights) changes nothing beyond rounding
    perm = rng.permutation(len(pops))
    cols = np.concatenate([np.arange(off[k], off[k + 1]) for k in perm])
    off2 = synth.pop_offsets([pops[k][1] for k in perm])
    r3 = hotpath.impute_window(1, np.ascontiguousarray(gm[:, cols]), np.ascontiguousarray(gu[:, cols]), off2, w[perm], z1, ctx=ctx)
    assert np.max(np.abs(r3["z"] - r["z"])) <= 1e-8 and relerr(r3["info"], r["info"]) <= 1e-9


def test_randomized_shapes_against_oracle(ctx):
    """Fuzz: random (M, U, population layout, mode) batches through one job each, against the oracle."""
    rng = np.random.default_rng(2026)
    for trial in range(6):
        P = int(rng.integers(1, 9))
        sizes = rng.integers(2, 400, size=P)
        if rng.random() < 0.3:
            sizes[rng.integers(0, P)] = int(rng.integers(2100, 2600))      # a population split into several segments
        off = synth.pop_offsets(sizes)
        N = int(off[-1])
        w = rng.uniform(0.05, 1.0, P)
        w *= (1.0 if rng.random() < 0.5 else 1.061) / w.sum()
        wins, wants = [], []
        for k in range(int(rng.integers(1, 6))):
            M, U = int(rng.integers(1, 200)), int(rng.integers(1, 220))
            mode = int(rng.integers(0, 2))
            G = rand_geno(rng, M + U, N, 0.1, 0.9)
            z1 = rng.standard_normal(M)
            gm, gu = np.ascontiguousarray(G[:M]), np.ascontiguousarray(G[M:])
            wins.append(dict(mode=mode, geno_m=gm, geno_u=gu, pop_off=off, pop_wgt=w, z1=z1))
            wants.append(oracle.run_impute(mode, gm, gu, off, w, z1, want_mats=True))
        job = hotpath.Job(wins, ctx=ctx, want_mats=True)
        job.run()
        res = job.fetch()
        job.close()
        for got, want in zip(res, wants):
            if want["mpd"] != 0 or not np.all(np.isfinite(want["z"])):
                continue
            assert got["status"] == 0
            assert np.max(np.abs(got["b11"] - want["b11"])) <= 1e-12
            assert np.max(np.abs(got["b21"] - want["b21"])) <= 1e-12
            assert relerr(got["info"], want["info"]) <= 1e-7
            assert np.max(np.abs(got["z"] - want["z"]) / np.maximum(1.0, np.abs(want["z"]))) <= 1e-7


def test_row_index_past_a_resident_store_is_refused(ctx):
    """Row lists are resolved on the GPU; an index beyond a store made by gauss_store_upload must be caught on the
    host (an out-of-bounds read there could fault the device), and so must a negative one."""
    from gauss_amd import panel
    p = small_panel(n_snp=60, scale=0.02, seed=81)
    rows, _ = panel.pack2bit(p["G"], p["off"])
    rs = hotpath.RowStore(rows, ctx=ctx)
    z1 = np.zeros(20)
    good = np.arange(20, dtype=np.int32)
    for bad_u in (np.array([25, 26, len(rows)], dtype=np.int32), np.array([25, -1, 27], dtype=np.int32)):
        with pytest.raises(Exception) as ei:
            hotpath.Job([dict(mode=0, pop_off=p["off"], pop_wgt=None, z1=z1, dev=(rs.ptr, rs.ptr, 20, 3, rs.ld),
                              packed=dict(fmt=1, rows_m=good, rows_u=bad_u))], ctx=ctx, on_device=True)
        assert "row" in str(ei.value)
    job = hotpath.Job([dict(mode=0, pop_off=p["off"], pop_wgt=None, z1=z1, dev=(rs.ptr, rs.ptr, 20, 3, rs.ld),
                            packed=dict(fmt=1, rows_m=good, rows_u=np.array([25, 26, len(rows) - 1], dtype=np.int32)))],
                      ctx=ctx, on_device=True)
    job.run()
    assert job.fetch()[0]["status"] == 0
    job.close()
    rs.close()


def test_full_size_properties_of_the_widened_rows(ctx):
    """QCAT, LD export with recoded rows, per-population LD and the 2-bit source at the BASELINE shape
    (N = 32 147, 21 populations): properties that need no oracle at full size + numpy spot checks."""
    from gauss_amd import panel
    pops = [p for p in synth.POPS_33KG if p[0] in synth.PGC2_WEIGHTS]
    off = synth.pop_offsets([p[1] for p in pops])
    w = np.array([synth.PGC2_WEIGHTS[p[0]] for p in pops])
    N = int(off[-1])
    rng = np.random.default_rng(78)
    M, U = 700, 900
    base = rand_geno(rng, 120, N)
    G = base[rng.integers(0, 120, size=M + U)].copy()
    noise = rng.random(G.shape) < 0.4
    G[noise] = rand_geno(rng, 1, N)[0][np.nonzero(noise)[1]]
    gm, gu = np.ascontiguousarray(G[:M]), np.ascontiguousarray(G[M:])
    z1 = rng.standard_normal(M) * 2
    # QCAT: |r| <= 1, num_eig = M (ridge 0.1 keeps every eigenvalue above the 0.01 cutoff), and r equals the
    # numpy statement on the GPU's own B11 / B21
    n_head, n_pred = 150, 400
    q = hotpath.qcat_window(1, gm, gu, off, w, z1, n_head, n_pred, want_mats=True, ctx=ctx)
    assert q["status"] == 0 and q["num_eig"] == M and q["r"].shape == (n_pred + U,)
    assert np.all(np.abs(q["r"]) <= 1 + 1e-12)
    L = np.linalg.cholesky(q["b11"])
    rhs = np.vstack([q["b11"][n_head:n_head + n_pred], q["b21"]]).T
    wz = np.linalg.solve(L, z1); wb = np.linalg.solve(L, rhs)
    wz = wz - wz.mean(); wb = wb - wb.mean(0)
    want = (wz @ wb) / np.sqrt((wz @ wz) * np.einsum("ij,ij->j", wb, wb))
    assert np.max(np.abs(q["r"] - want)) <= 1e-9
    # the same window from 2-bit rows in a resident store, rows in scrambled order: bit-identical
    rows, src_off = panel.pack2bit(G, off)
    perm = rng.permutation(M + U)
    store = hotpath.RowStore(np.ascontiguousarray(rows[perm]), ctx=ctx)
    inv = np.argsort(perm).astype(np.int32)
    job = hotpath.Job([dict(mode=1, pop_off=off, pop_wgt=w, z1=z1, dev=(store.ptr, store.ptr, M, U, store.ld),
                            qcat=(n_head, n_pred, 0.01), packed=dict(fmt=1, rows_m=inv[:M], rows_u=inv[M:]))],
                      ctx=ctx, on_device=True)
    job.run()
    q2 = job.fetch()[0]
    job.close()
    store.close()
    assert np.array_equal(q2["r"], q["r"]) and q2["num_eig"] == M
    # LD export with recoded prediction rows: the dominant / recessive blocks equal the additive LD of the
    # recoded genotypes; unit diagonal
    ld = hotpath.ld_window(1, gm[:300], gu[:200], off, w, lam=0.0, codings=7, ctx=ctx)
    dom = hotpath.ld_window(1, gm[:300], (gu[:200] >= 1).astype(np.uint8), off, w, lam=0.0, codings=1, ctx=ctx)
    rec = hotpath.ld_window(1, gm[:300], (gu[:200] == 2).astype(np.uint8), off, w, lam=0.0, codings=1, ctx=ctx)
    assert np.all(np.diag(ld["b11"]) == 1.0)
    assert np.array_equal(ld["b21"][200:400], dom["b21"], equal_nan=True)
    assert np.array_equal(ld["b21"][400:600], rec["b21"], equal_nan=True)
    # per-population LD: population k's column equals numpy's corrcoef on that population's samples
    pp = hotpath.ld_per_pop(gm[:120], off, ctx=ctx)
    iu = np.triu_indices(120, 1)
    for k in (0, 7, 20):
        with np.errstate(invalid="ignore", divide="ignore"):
            r = np.corrcoef(gm[:120, off[k]:off[k + 1]].astype(float))[iu]
        ok = np.isfinite(r)
        assert np.array_equal(np.isfinite(pp[k]), ok) and np.max(np.abs(pp[k][ok] - r[ok])) <= 1e-12


def test_randomized_kinds_and_sources_against_oracle(ctx):
    """Fuzz over the widened rows: each job mixes imputation, QCAT and LD-export windows; genotype sources are
    byte matrices, 2-bit matrices or row lists into a resident 2-bit store, at random."""
    from gauss_amd import panel
    rng = np.random.default_rng(2027)
    for trial in range(5):
        P = int(rng.integers(1, 8))
        sizes = rng.integers(20, 300, size=P)
        off = synth.pop_offsets(sizes)
        N = int(off[-1])
        w = rng.uniform(0.05, 1.0, P)
        w /= w.sum()
        S = int(rng.integers(150, 420))
        G = rand_geno(rng, S, N, 0.15, 0.85)
        rows2, _ = panel.pack2bit(G, off)
        store = hotpath.RowStore(rows2, ctx=ctx)
        wins, checks = [], []
        on_device = bool(rng.integers(0, 2))
        for k in range(int(rng.integers(2, 6))):
            M, U = int(rng.integers(12, 120)), int(rng.integers(1, 130))
            idx = rng.permutation(S)[: M + U].astype(np.int32)
            mi, ui = np.sort(idx[:M]), np.sort(idx[M:])
            gm, gu = np.ascontiguousarray(G[mi]), np.ascontiguousarray(G[ui])
            mode = int(rng.integers(0, 2))
            z1 = rng.standard_normal(M)
            d = dict(mode=mode, pop_off=off, pop_wgt=w, z1=z1)
            if on_device:
                d.update(dev=(store.ptr, store.ptr, M, U, store.ld), packed=dict(fmt=1, rows_m=mi, rows_u=ui))
            elif rng.random() < 0.5:
                d.update(geno_m=np.ascontiguousarray(rows2[mi]), geno_u=np.ascontiguousarray(rows2[ui]), packed=dict(fmt=1))
            else:
                d.update(geno_m=gm, geno_u=gu)
            kind = int(rng.integers(0, 3))
            if kind == 1:
                n_head = int(rng.integers(0, M // 2)); n_pred = int(rng.integers(1, M - n_head + 1))
                d["qcat"] = (n_head, n_pred, 0.01)
                checks.append(("qcat", oracle.run_qcat(mode, gm, gu, off, w, z1, n_head, n_pred)))
            elif kind == 2:
                codes = [c for c in (0, 1, 2) if rng.random() < 0.6] or [0]
                d["ld_codings"] = sum(1 << c for c in codes)
                d["lam"] = 0.0
                checks.append(("ld", oracle.ld_blocks(mode, gm, gu, off, w, 1.0, tuple(codes))))
            else:
                checks.append(("imp", oracle.run_impute(mode, gm, gu, off, w, z1)))
            wins.append(d)
        job = hotpath.Job(wins, ctx=ctx, on_device=on_device)
        job.run()
        res = job.fetch()
        job.close()
        store.close()
        for got, (kind, want) in zip(res, checks):
            if kind == "imp":
                if want["mpd"] != 0 or not np.all(np.isfinite(want["z"])):
                    continue
                assert relerr(got["info"], want["info"]) <= 1e-7
                assert np.max(np.abs(got["z"] - want["z"]) / np.maximum(1.0, np.abs(want["z"]))) <= 1e-7
            elif kind == "qcat":
                if not np.all(np.isfinite(want["r"])):
                    continue
                assert got["num_eig"] == want["num_eig"] and np.max(np.abs(got["r"] - want["r"])) <= 1e-8
            else:
                nan = np.isnan(want["b21"])
                assert np.array_equal(np.isnan(got["b21"]), nan)
                assert np.max(np.abs(got["b11"] - want["b11"])) <= 1e-12
                assert nan.all() or np.max(np.abs(got["b21"][~nan] - want["b21"][~nan])) <= 1e-12


@pytest.mark.parametrize("mode,pop_names", [(1, None), (0, ["GBR"])])
def test_full_size_gene_ld_batch(ctx, mode, pop_names):
    """BASELINE.json configs[4] at its real size: ~350 genes with 1-20 SNPs each (3 700 gene SNPs), jepegmix on the 21
    PGC2 populations (N = 32 147, weighted LD) and jepeg on GBR (N = 2 020, pooled Pearson): gene blocks against the
    whole-matrix LD of the same rows (bit for bit), structural properties, and oracle spot checks of single entries."""
    if pop_names is None:
        pops = [p for p in synth.POPS_33KG if p[0] in synth.PGC2_WEIGHTS]
        w = np.array([synth.PGC2_WEIGHTS[p[0]] for p in pops])
    else:
        pops = [p for p in synth.POPS_33KG if p[0] in pop_names]
        w = None
    off = synth.pop_offsets([p[1] for p in pops])
    N = int(off[-1])
    assert N == (32147 if mode else 2020)
    rng = np.random.default_rng(5 + mode)
    sizes = rng.integers(1, 21, size=350)
    S = int(sizes.sum())
    gene_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    base = rand_geno(rng, 200, N)
    G = base[rng.integers(0, 200, size=S)].copy()
    noise = rng.random(G.shape) < 0.3
    G[noise] = rand_geno(rng, 1, N)[0][np.nonzero(noise)[1]]
    G = np.ascontiguousarray(G)
    diag = 1.1
    blocks = hotpath.gene_ld_batch(G, off, gene_off, pop_wgt=w, mode=mode, diag=diag, ctx=ctx)
    assert len(blocks) == 350
    for g, b in enumerate(blocks):
        n = sizes[g]
        assert b.shape == (n, n) and np.array_equal(b, b.T) and np.all(np.diag(b) == diag)
        assert np.all(np.isfinite(b)) and np.all(np.abs(b - np.diag(np.diag(b))) <= 1 + 1e-12)
    # the same entries as the full LD matrix of a stretch of rows (one launch, all tile pairs)
    r0, r1 = int(gene_off[40]), int(gene_off[75])
    full = hotpath.ld_matrix(G[r0:r1], off, w, mode=mode, diag=diag, ctx=ctx)
    for g in range(40, 75):
        a, b = int(gene_off[g]) - r0, int(gene_off[g + 1]) - r0
        assert np.array_equal(blocks[g], full[a:b, a:b])
    # oracle spot checks (loop-literal pair functions)
    def entry(i, j):
        if mode == 0:
            return oracle.calcor(G[i], G[j], off)
        sd = lambda r: np.sqrt(oracle.calwgtcov(r, r, off, w))
        return oracle.calwgtcov(G[i], G[j], off, w) / (sd(G[i]) * sd(G[j]))
    checked = 0
    for g in rng.choice(350, size=12, replace=False):
        n = sizes[g]
        if n < 2:
            continue
        i, j = sorted(rng.choice(n, size=2, replace=False))
        assert abs(blocks[g][i, j] - entry(gene_off[g] + i, gene_off[g] + j)) <= 1e-13
        checked += 1
    assert checked >= 6


def test_full_size_dist_window_at_the_largest_real_window(ctx):
    """BASELINE.json configs[2] at its real size: dist(study_pop = "EUR"), N = 20 281 (6 populations pooled), with the
    shape of the LARGEST window of the chr22 study (M = 1 213 measured = 19 factor blocks, U = 2 583): properties that
    need no oracle at full size, oracle spot checks of single LD entries (loop-literal CalCor), and the solve against
    numpy on the GPU's own B11 / B21."""
    pops = [p for p in synth.POPS_33KG if p[2] == "EUR"]
    off = synth.pop_offsets([p[1] for p in pops])
    N = int(off[-1])
    assert N == 20281
    rng = np.random.default_rng(99)
    M, U = 1213, 2583
    base = rand_geno(rng, 220, N)
    G = base[rng.integers(0, 220, size=M + U)].copy()
    noise = rng.random(G.shape) < 0.4
    G[noise] = rand_geno(rng, 1, N)[0][np.nonzero(noise)[1]]
    gm, gu = np.ascontiguousarray(G[:M]), np.ascontiguousarray(G[M:])
    z1 = rng.standard_normal(M) * 2
    r = hotpath.impute_window(0, gm, gu, off, None, z1, want_mats=True, ctx=ctx)
    assert r["status"] == 0
    b11, b21 = r["b11"], r["b21"]
    assert np.array_equal(b11, b11.T) and np.all(np.diag(b11) == 1.1)
    assert np.all(np.abs(b21) <= 1 + 1e-9) and np.all(np.isfinite(r["z"]))
    assert np.all(r["info"] > 0) and np.all(r["info"] < 1 + 1e-9)
    assert np.linalg.eigvalsh(b11).min() > 0.09
    for (i, j) in [(0, 1), (5, 1200), (640, 641), (1212, 3)]:
        assert abs(b11[i, j] - oracle.calcor(gm[i], gm[j], off)) <= 1e-13
    for (u, j) in [(0, 0), (2582, 1212), (1300, 17)]:
        assert abs(b21[u, j] - oracle.calcor(gu[u], gm[j], off)) <= 1e-13
    y = b21 @ np.linalg.inv(b11)
    info = np.abs(np.einsum("ij,ij->i", y, b21))
    assert relerr(r["info"], info) <= 1e-9
    assert np.max(np.abs(r["z"] - (y @ z1) / np.sqrt(info))) <= 1e-8
    # the same window inside a batch of others (shared launches, riding rows of the inverse, shared tiles of the closing product) gives the same bits
    small = dict(mode=0, geno_m=gm[:200], geno_u=gu[:300], pop_off=off, pop_wgt=None, z1=z1[:200])
    job = hotpath.Job([small, dict(mode=0, geno_m=gm, geno_u=gu, pop_off=off, pop_wgt=None, z1=z1), small], ctx=ctx)
    job.run()
    res = job.fetch()
    job.close()
    assert np.array_equal(res[1]["z"], r["z"]) and np.array_equal(res[1]["info"], r["info"])
    assert np.array_equal(res[0]["z"], res[2]["z"])


@pytest.mark.parametrize("mode", [0, 1])
def test_tail_accuracy_level_against_the_oracle(ctx, mode):
    """The north star allows 1e-5 relative, the parity tests hold 1e-8; the level actually reached by the fused tail
    (Cholesky, inverse factor, fp64-MFMA product) against the oracle's full-pivot-LU path in the reference's operation
    order is ~1e-13 on windows of up to 1 200 measured SNPs: assert 1e-12, so that a loss of accuracy shows long
    before it matters."""
    p = small_panel(n_snp=1700, scale=0.1, seed=3, span_bp=3_000_000)
    from helpers import split_window
    for M in (200, 640, 1200):
        gm, gu, z1 = split_window(dict(G=p["G"][: M + 400]), M)
        got = hotpath.impute_window(mode, gm, gu, p["off"], p["w"], z1, ctx=ctx)
        want = oracle.run_impute(mode, gm, gu, p["off"], p["w"], z1)
        assert np.max(np.abs(got["z"] - want["z"]) / np.maximum(1.0, np.abs(want["z"]))) <= 1e-12, M
        assert relerr(got["info"], want["info"]) <= 1e-12, M


@pytest.mark.gpu
def test_sixteen_column_edge_of_the_gram_kernel(ctx):
    """A wave of the f32 Gram kernel whose last live 32-column half holds at most 16 live columns multiplies its columns with
    v_mfma_f32_16x16x4_f32 (k_gram.hip, chunk_mfma_edge): row counts on both sides of every 16 / 32 / 64 / 128 boundary give the
    oracle's integers (f32 slabs: one-byte codes up to 15), and windows over a 2-bit store (16-bit slabs) with such M give the
    bits of the int8 path, which has no edge routine."""
    from gauss_amd import panel as panel_mod
    rng = np.random.default_rng(77)
    N = 700
    for S in (1, 9, 16, 17, 31, 33, 40, 48, 49, 64, 65, 77, 80, 81, 97, 112, 113, 129, 140, 144, 145, 161, 176, 177, 200, 209):
        G = rng.integers(0, 16 if S % 2 else 3, size=(S, N)).astype(np.uint8)
        want = G.astype(np.int64) @ G.astype(np.int64).T
        assert np.array_equal(hotpath.gram_counts(G, ctx=ctx), want), S
    p = small_panel(n_snp=900, scale=0.04, seed=13)
    rows2, src_off = panel_mod.pack2bit(p["G"], p["off"])
    store = hotpath.RowStore(rows2, ctx=ctx)
    wins = []
    for k, M in enumerate((7, 16, 17, 40, 47, 73, 80, 100, 139, 145, 176, 205)):
        mi = np.arange(k, k + 2 * M, 2, dtype=np.int32)
        ui = np.arange(k + 1, k + 1 + 2 * 150, 2, dtype=np.int32)
        wins.append(dict(mode=k % 2, pop_off=p["off"], pop_wgt=p["w"], z1=rng.standard_normal(M), dev=(store.ptr, store.ptr, M, len(ui), store.ld),
                         packed=dict(fmt=1, rows_m=mi, rows_u=ui, pop_src_off=src_off)))
    out = {}
    try:
        for dt in ("f32", "i8"):
            ctx.set_gram_dtype(dt)
            res = []
            for grp in (wins[0::2], wins[1::2]):                 # one mode per job
                job = hotpath.Job(grp, ctx=ctx, on_device=True, want_mats=True)
                job.run()
                res += job.fetch()
                job.close()
            out[dt] = res
    finally:
        ctx.set_gram_dtype(os.environ.get("GAUSS_GRAM_DTYPE", "f32"))      # (back to the session's form: a suite run under GAUSS_GRAM_DTYPE=i8 stays on it)
    for a, b in zip(out["f32"], out["i8"]):
        for key in ("z", "info", "b11", "b21"):
            assert np.array_equal(a[key], b[key], equal_nan=True), key
    store.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_more_samples_than_the_pack_kernels_lds_word_table(ctx, mode):
    """N = 140 000 samples in 3 populations: a packed row has more than 8 192 words of 16 samples, so pack_stats_kernel
    looks its word -> block table up in global memory (`tab_lds == false`, k_pack_epilogue.hip) -- a branch no panel of
    33KG size reaches.  One-byte and 2-bit sources against the whole oracle (util.cpp:49-70 / 103-124 at N = 140 000;
    dist.cpp:181-202), the two sources bit for bit against each other."""
    from gauss_amd import panel as panel_mod
    rng = np.random.default_rng(140)
    sizes = [70_001, 40_000, 29_999]
    off = synth.pop_offsets(sizes)
    N = int(off[-1])
    assert N == 140_000 and (N + 63) // 64 * 64 // 16 > 8192
    M, U = 40, 40
    G = rand_geno(rng, M + U, N)
    gm, gu = np.ascontiguousarray(G[:M]), np.ascontiguousarray(G[M:])
    w = np.array([0.5, 0.361, 0.2])
    z1 = rng.standard_normal(M) * 2
    got = hotpath.impute_window(mode, gm, gu, off, w, z1, want_mats=True, ctx=ctx)
    want = oracle.run_impute(mode, gm, gu, off, w, z1, want_mats=True)
    assert got["status"] == 0 and want["mpd"] == 0
    assert np.max(np.abs(got["b11"] - want["b11"])) <= 1e-12 and np.max(np.abs(got["b21"] - want["b21"])) <= 1e-12
    assert relerr(got["info"], want["info"]) <= 1e-8
    assert np.max(np.abs(got["z"] - want["z"]) / np.maximum(1.0, np.abs(want["z"]))) <= 1e-8
    cnt = hotpath.gram_counts(gm[:20], ctx=ctx)                      # exact integers over 140 000 samples
    assert np.array_equal(cnt, gm[:20].astype(np.int64) @ gm[:20].astype(np.int64).T)
    rows2, src_off = panel_mod.pack2bit(G, off)
    job = hotpath.Job([dict(mode=mode, geno_m=rows2[:M], geno_u=rows2[M:], pop_off=off, pop_wgt=w, z1=z1, packed=dict(fmt=1))],
                      ctx=ctx, want_mats=True)
    job.run()
    two = job.fetch()[0]
    job.close()
    for key in ("z", "info", "b11", "b21"):
        assert np.array_equal(two[key], got[key]), key


@pytest.mark.gpu
def test_window_of_sixty_four_factor_blocks(ctx):
    """M = 4 096 measured SNPs (a 2 Mb window of a dense array): 64 factor blocks, 65 panels of [I | z1], 32 k blocks of the
    closing product and Mld-sized workspaces no chr22 window reaches (M <= 1 213 there).  N = 2 000, U = 1 024.  The oracle's
    eigen-decomposition of a 4 096 x 4 096 matrix is out of a test's reach, so: LD entries spot-checked against the oracle's pair
    function (util.cpp:103-124), structure of B11, and the solve against numpy on the GPU's own B11 / B21 (dist.cpp:181-202)."""
    rng = np.random.default_rng(4096)
    sizes = [700, 650, 650]
    off = synth.pop_offsets(sizes)
    N = int(off[-1])
    M, U = 4096, 1024
    G = rand_geno(rng, M + U, N)
    gm, gu = np.ascontiguousarray(G[:M]), np.ascontiguousarray(G[M:])
    w = np.array([0.4, 0.4, 0.261])
    z1 = rng.standard_normal(M) * 2
    r = hotpath.impute_window(1, gm, gu, off, w, z1, want_mats=True, ctx=ctx)
    assert r["status"] == 0
    b11, b21 = r["b11"], r["b21"]
    assert b11.shape == (M, M) and b21.shape == (U, M)
    assert np.array_equal(b11, b11.T) and np.all(np.diag(b11) == 1.1)

    def std(row):
        return np.sqrt(oracle.calwgtcov(row, row, off, w))
    for (i, j) in [(0, 1), (63, 64), (4095, 0), (2048, 4094), (1000, 3001)]:
        want = oracle.calwgtcov(gm[i], gm[j], off, w) / (std(gm[i]) * std(gm[j]))
        assert abs(b11[i, j] - want) <= 1e-13, (i, j)
    for (u, j) in [(0, 0), (1023, 4095), (512, 64), (77, 4032)]:
        want = oracle.calwgtcov(gu[u], gm[j], off, w) / (std(gu[u]) * std(gm[j]))
        assert abs(b21[u, j] - want) <= 1e-13, (u, j)
    y = np.linalg.solve(b11, b21.T).T                              # b21 B11^-1  (B11 symmetric)
    info = np.abs(np.einsum("ij,ij->i", y, b21))
    assert relerr(r["info"], info) <= 1e-9
    assert np.max(np.abs(r["z"] - (y @ z1) / np.sqrt(info))) <= 1e-8
    # the same window in a job of its own kind of company: a small window beside it must not change its bits
    small = dict(mode=1, geno_m=gm[:70], geno_u=gu[:30], pop_off=off, pop_wgt=w, z1=z1[:70])
    job = hotpath.Job([small, dict(mode=1, geno_m=gm, geno_u=gu, pop_off=off, pop_wgt=w, z1=z1)], ctx=ctx)
    job.run()
    res = job.fetch()
    job.close()
    assert np.array_equal(res[1]["z"], r["z"]) and np.array_equal(res[1]["info"], r["info"])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_standalone_solver_against_the_oracle(ctx, mode, monkeypatch):
    """GAUSS_FUSED_SOLVE=0 (README: a supported fallback) factors B11 and pushes the window's own right-hand sides through the
    substitution (solve_kernel, the clamp path's solver) instead of inverse rows + closing product: the shapes of
    test_tail_block_edges -- every side of the 64- and 128-block edges -- as one job and as single-window calls (which then
    upload, then run: no streamed form), against the oracle (dist.cpp:181-202, distmix.cpp:165-228)."""
    monkeypatch.setenv("GAUSS_FUSED_SOLVE", "0")
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
        wants.append(oracle.run_impute(mode, gm, gu, off, p["w"], z1))
    job = hotpath.Job(wins, ctx=ctx)
    job.run()
    res = job.fetch()
    job.close()
    for want, got, wn in zip(wants, res, wins):
        shape = (wn["geno_m"].shape[0], wn["geno_u"].shape[0])
        assert got["status"] == 0, shape
        assert relerr(got["info"], want["info"]) <= 1e-8, shape
        assert np.max(np.abs(got["z"] - want["z"]) / np.maximum(1.0, np.abs(want["z"]))) <= 1e-8, shape
    for k in (0, 3, 8):
        wn = wins[k]
        one = hotpath.impute_window(mode, wn["geno_m"], wn["geno_u"], off, p["w"], wn["z1"], ctx=ctx)
        assert np.array_equal(one["z"], res[k]["z"]) and np.array_equal(one["info"], res[k]["info"]), k


@pytest.mark.gpu
@pytest.mark.parametrize("asynchronous", [False, True])
def test_row_store_beyond_four_gigabytes(ctx, asynchronous):
    """A whole-genome panel is tens of GB in one row store (the 33KG panel: 82 GB): byte offsets of rows past 4 GB must be
    64-bit everywhere -- the staged upload's chunk offsets, the background upload's copy kernel (stores above 4 GB travel by
    kernel) and its per-chunk marks, the pack kernel's row gather.  A 5.4 GB store of 8 KB rows holds the same 2-bit rows at
    row 0 and at row 600 000 (4.9 GB in): a window that names the high rows must give the bits of the one that names the low
    rows, which is checked against the oracle."""
    p = small_panel(n_snp=230, scale=0.02, seed=77)
    G, off = p["G"], p["off"]
    rows2, src_off = panel_mod.pack2bit(G, off)
    S, rb = rows2.shape
    ld, n_rows, hi0 = 8192, 660_000, 600_000
    assert n_rows * ld > (5 << 30) and hi0 * ld > (4 << 30) and rb <= ld
    big = np.zeros((n_rows, ld), dtype=np.uint8)                    # (untouched pages cost nothing on the host)
    big[:S, :rb] = rows2
    big[hi0:hi0 + S, :rb] = rows2
    store = hotpath.RowStore(big, ctx=ctx, asynchronous=asynchronous)
    try:
        rng = np.random.default_rng(5)
        idx = rng.permutation(S)
        mi, ui = np.sort(idx[:90]), np.sort(idx[90:200])
        z1 = rng.standard_normal(len(mi))
        if asynchronous:
            store.wait(hi0 + S)                                     # the rows the job names have landed (the rest may still travel)
        wins = []
        for base in (0, hi0):
            wins.append(dict(mode=1, pop_off=off, pop_wgt=p["w"], z1=z1, dev=(store.ptr, store.ptr, len(mi), len(ui), store.ld),
                             packed=dict(fmt=1, rows_m=(mi + base).astype(np.int32), rows_u=(ui + base).astype(np.int32), pop_src_off=src_off)))
        job = hotpath.Job(wins, ctx=ctx, on_device=True, want_mats=True)
        job.run()
        lo, hi = job.fetch()
        job.close()
        if asynchronous:
            store.wait(0)
        for key in ("z", "info", "b11", "b21"):
            assert np.array_equal(lo[key], hi[key], equal_nan=True), key
        want = oracle.run_impute(1, G[mi], G[ui], off, p["w"], z1, want_mats=True)
        assert np.max(np.abs(lo["b21"] - want["b21"])) <= 1e-12
        assert np.max(np.abs(lo["z"] - want["z"]) / np.maximum(1.0, np.abs(want["z"]))) <= 1e-8
        assert np.all(np.isfinite(hi["z"])) and np.any(hi["z"] != 0)
    finally:
        store.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_window_with_thirty_thousand_unmeasured_snps(ctx, mode):
    """A sparse array imputed against a dense panel: few measured SNPs, tens of thousands of unmeasured ones in one window (235
    row tiles of B21, one column tile) -- every z / info against the oracle."""
    from gauss_amd import synth
    pops = [("AAA", 170, "EUR"), ("BBB", 230, "EUR"), ("CCC", 200, "ASN")]
    rng = np.random.default_rng(99)
    S = 30_060
    bp = np.sort(rng.choice(np.arange(1, 3_000_000), size=S, replace=False))
    G, _ = synth.synth_genotypes(bp, pops, seed=31)
    G = np.ascontiguousarray(G[G.min(1) != G.max(1)])
    off = synth.pop_offsets([q[1] for q in pops])
    mi = np.sort(rng.choice(G.shape[0], size=44, replace=False))
    ui = np.setdiff1d(np.arange(G.shape[0]), mi)
    assert len(ui) > 29_000
    gm, gu = np.ascontiguousarray(G[mi]), np.ascontiguousarray(G[ui])
    z1 = rng.standard_normal(len(mi)) * 2.0
    w = np.array([0.4, 0.35, 0.3]) if mode else None
    got = hotpath.impute_window(mode, gm, gu, off, w, z1, ctx=ctx)
    want = oracle.run_impute(mode, gm, gu, off, w, z1)
    assert got["status"] == 0 and got["z"].shape == (len(ui),)
    assert np.max(np.abs(got["z"] - want["z"]) / np.maximum(1.0, np.abs(want["z"]))) <= 1e-8
    assert relerr(got["info"], want["info"]) <= 1e-8


@pytest.mark.gpu
def test_sixty_four_populations_and_one_too_many(ctx):
    """The population table's limit: 64 populations (sizes 2 ... 60, weighted and pooled) against the oracle, 65 refused."""
    from gauss_amd import synth
    rng = np.random.default_rng(64)
    pops = [(f"P{k:02d}", int(rng.integers(2, 61)), "EUR") for k in range(64)]
    bp = np.sort(rng.choice(np.arange(1, 500_000), size=260, replace=False))
    G, _ = synth.synth_genotypes(bp, pops, seed=3)
    G = np.ascontiguousarray(G[G.min(1) != G.max(1)])
    off = synth.pop_offsets([q[1] for q in pops])
    w = rng.uniform(0.0, 0.03, size=64)
    gm, gu = np.ascontiguousarray(G[:120]), np.ascontiguousarray(G[120:240])
    z1 = rng.standard_normal(120)
    for mode, ww in ((1, w), (0, None)):
        got = hotpath.impute_window(mode, gm, gu, off, ww, z1, want_mats=True, ctx=ctx)
        want = oracle.run_impute(mode, gm, gu, off, ww, z1, want_mats=True)
        nan = np.isnan(want["b21"])
        assert np.array_equal(np.isnan(got["b21"]), nan)
        assert np.max(np.abs(got["b21"][~nan] - want["b21"][~nan]), initial=0.0) <= 1e-12
        ok = ~np.isnan(want["z"])
        assert np.array_equal(np.isnan(got["z"]), ~ok)
        assert np.max(np.abs(got["z"][ok] - want["z"][ok]) / np.maximum(1.0, np.abs(want["z"][ok])), initial=0.0) <= 1e-8
    pops65 = pops + [("P64", 10, "EUR")]
    off65 = synth.pop_offsets([q[1] for q in pops65])
    G65 = np.ascontiguousarray(np.hstack([G, G[:, :10]]))
    from gauss_amd import _lib
    with pytest.raises(_lib.GaussHipError, match="n_pop must be in 1..64"):
        hotpath.impute_window(0, G65[:120], G65[120:240], off65, None, z1, ctx=ctx)


def test_matrix_exports_through_the_mirror_and_past_its_budget(ctx):
    """A job's B11 / B21 leave through a pinned export mirror (gauss_run.cpp: queue_exports) while they fit 256 MB of it, and matrix
    by matrix (one pitched copy each, compacted on the host) when they do not.  Five LD-export windows of 1 900 + 1 900 SNPs want
    36 M doubles back (289 MB: past the budget); the same windows one job each go through the mirror; both forms return the same
    bits, and the first window's matrices equal the oracle's pair loops on a corner."""
    rng = np.random.default_rng(77)
    off = np.array([0, 150, 330], dtype=np.int32)
    w = np.array([0.6, 0.5])
    wins = []
    for k in range(5):
        G = rand_geno(rng, 3800, 330, 0.1, 0.9)
        wins.append(dict(mode=1, geno_m=np.ascontiguousarray(G[:1900]), geno_u=np.ascontiguousarray(G[1900:]), pop_off=off, pop_wgt=w,
                         z1=np.zeros(1900), lam=0.0, ld_codings=1))
    job = hotpath.Job(wins, ctx=ctx)
    job.run()
    big = [dict(b11=r["b11"].copy(), b21=r["b21"].copy()) for r in job.fetch()]
    job.close()
    for k, win in enumerate(wins):
        one = hotpath.Job([win], ctx=ctx)
        one.run()
        r = one.fetch()[0]
        one.close()
        assert np.array_equal(r["b11"], big[k]["b11"]) and np.array_equal(r["b21"], big[k]["b21"]), k
    gm, gu = wins[0]["geno_m"], wins[0]["geno_u"]
    want = oracle.compute_ld(np.ascontiguousarray(np.vstack([gm[:40], gu[:40]])), off, w)
    assert np.max(np.abs(big[0]["b11"][:40, :40] - want[:40, :40])) <= 1e-12
    assert np.max(np.abs(big[0]["b21"][:40, :40] - want[40:, :40])) <= 1e-12
    assert np.all(np.diag(big[0]["b11"]) == 1.0) and np.array_equal(big[0]["b11"], big[0]["b11"].T)
