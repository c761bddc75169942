"""Golden vectors (tests/golden/window_small.npz, made by tests/golden/make_golden.py).
CPU: the oracle still reproduces them.  GPU: the HIP path matches them."""
import os

import numpy as np
import pytest

import oracle

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "window_small.npz"))


def test_oracle_reproduces_golden():
    for mode in (0, 1):
        r = oracle.run_impute(mode, G["gm"], G["gu"], G["off"], G["w"], G["zin"], want_mats=True)
        assert np.array_equal(r["b11"], G[f"b11_{mode}"]) and np.array_equal(r["b21"], G[f"b21_{mode}"])
        assert np.max(np.abs(r["z"] - G[f"z{mode}"])) <= 1e-12 and np.max(np.abs(r["info"] - G[f"info{mode}"])) <= 1e-13
    assert np.array_equal(oracle.compute_ld(G["gm"], G["off"], G["w"]), G["ld_weighted"])
    assert np.array_equal(oracle.gram_counts(G["gm"]), G["counts"])


@pytest.mark.gpu
def test_hip_matches_golden(ctx):
    from gauss_amd import hotpath
    assert np.array_equal(hotpath.gram_counts(G["gm"], ctx=ctx), G["counts"])
    assert np.max(np.abs(hotpath.ld_matrix(G["gm"], G["off"], G["w"], ctx=ctx) - G["ld_weighted"])) <= 1e-12
    assert np.max(np.abs(hotpath.ld_matrix(G["gm"], G["off"], None, mode=0, diag=1.1, ctx=ctx) - G["ld_pooled"])) <= 1e-12
    for mode in (0, 1):
        r = hotpath.impute_window(mode, G["gm"], G["gu"], G["off"], G["w"], G["zin"], want_mats=True, ctx=ctx)
        assert np.max(np.abs(r["b11"] - G[f"b11_{mode}"])) <= 1e-12
        assert np.max(np.abs(r["z"] - G[f"z{mode}"]) / np.maximum(1, np.abs(G[f"z{mode}"]))) <= 1e-8
        assert np.max(np.abs(r["info"] - G[f"info{mode}"]) / G[f"info{mode}"]) <= 1e-8


X = np.load(os.path.join(os.path.dirname(__file__), "golden", "window_small_ext.npz"))


def _eq(a, b, tol):
    nan = np.isnan(b)
    return a.shape == b.shape and np.array_equal(np.isnan(a), nan) and (nan.all() or np.max(np.abs(a[~nan] - b[~nan])) <= tol)


def test_oracle_reproduces_extended_golden():
    nh, npred = int(X["n_head"]), int(X["n_pred"])
    for mode in (0, 1):
        q = oracle.run_qcat(mode, G["gm"], G["gu"], G["off"], G["w"], G["zin"], nh, npred)
        assert q["num_eig"] == int(X[f"qcat_num_eig{mode}"]) and _eq(q["r"], X[f"qcat_r{mode}"], 1e-13)
        b = oracle.ld_blocks(mode, G["gm"], G["gu"], G["off"], G["w"], 1.0, (0, 1, 2))
        assert _eq(b["b11"], X[f"ldx_b11_{mode}"], 0.0) and _eq(b["b21"], X[f"ldx_b21_{mode}"], 0.0)
    assert _eq(oracle.ld_per_pop(G["gm"][:24], G["off"]), X["ld_per_pop"], 0.0)


@pytest.mark.gpu
def test_hip_matches_extended_golden(ctx):
    from gauss_amd import hotpath
    nh, npred = int(X["n_head"]), int(X["n_pred"])
    for mode in (0, 1):
        q = hotpath.qcat_window(mode, G["gm"], G["gu"], G["off"], G["w"], G["zin"], nh, npred, ctx=ctx)
        assert q["num_eig"] == int(X[f"qcat_num_eig{mode}"]) and _eq(q["r"], X[f"qcat_r{mode}"], 1e-9)
        b = hotpath.ld_window(mode, G["gm"], G["gu"], G["off"], G["w"], lam=0.0, codings=7, ctx=ctx)
        assert _eq(b["b11"], X[f"ldx_b11_{mode}"], 1e-12) and _eq(b["b21"], X[f"ldx_b21_{mode}"], 1e-12)
    assert _eq(hotpath.ld_per_pop(G["gm"][:24], G["off"], ctx=ctx), X["ld_per_pop"], 1e-12)
