#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz: seeded inputs + the CPU oracle's outputs, frozen.

The reference ships no golden vectors for this path and cannot be run here (no R/Rcpp/RcppEigen), so
these vectors pin the *oracle* (cross-validated against numpy/scipy in tests/test_oracle.py), not the
reference binary: "parity unpinned" in the sense of the task statement.  They guard against silent
drift of either the oracle or the HIP path between rounds.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from gauss_amd import synth  # noqa: E402


def main():
    pops = [("CEU", 151, "EUR"), ("FIN", 97, "EUR"), ("CCE", 130, "ASN"), ("ASW", 41, "AFR"), ("GBR", 88, "EUR"),
            ("JPT", 64, "ASN"), ("YRI", 29, "AFR")]
    rng = np.random.default_rng(20260213)
    bp = np.sort(rng.choice(np.arange(1, 250_000), size=110, replace=False))
    G, _ = synth.synth_genotypes(bp, pops, seed=20260213)
    G = G[G.min(1) != G.max(1)][:96]
    off = synth.pop_offsets([p[1] for p in pops])
    w = np.array([0.165, 0.138, 0.008, 0.036, 0.165, 0.011, 0.538])        # sums to 1.061 like PGC2
    idx = rng.permutation(G.shape[0])
    gm = np.ascontiguousarray(G[np.sort(idx[:42])])
    gu = np.ascontiguousarray(G[np.sort(idx[42:])])
    z1 = np.round(rng.standard_normal(gm.shape[0]) * 2, 6)
    out = dict(gm=gm, gu=gu, off=off, w=w, zin=z1)
    for mode in (0, 1):
        r = oracle.run_impute(mode, gm, gu, off, w, z1, want_mats=True)
        out[f"z{mode}"], out[f"info{mode}"], out[f"b11_{mode}"], out[f"b21_{mode}"] = r["z"], r["info"], r["b11"], r["b21"]
    out["ld_weighted"] = oracle.compute_ld(gm, off, w)
    out["ld_pooled"] = oracle.ld_pooled(gm, off, 1.1)
    out["counts"] = oracle.gram_counts(gm)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "window_small.npz"), **out)
    print("wrote tests/golden/window_small.npz", {k: v.shape for k, v in out.items()})

    # the widened rows: QCAT, raw LD export with recoded rows, per-population LD
    ext = dict(n_head=np.int32(6), n_pred=np.int32(25))
    for mode in (0, 1):
        q = oracle.run_qcat(mode, gm, gu, off, w, z1, 6, 25)
        ext[f"qcat_r{mode}"], ext[f"qcat_num_eig{mode}"] = q["r"], np.int32(q["num_eig"])
        b = oracle.ld_blocks(mode, gm, gu, off, w, 1.0, (0, 1, 2))
        ext[f"ldx_b11_{mode}"], ext[f"ldx_b21_{mode}"] = b["b11"], b["b21"]
    ext["ld_per_pop"] = oracle.ld_per_pop(gm[:24], off)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "window_small_ext.npz"), **ext)
    print("wrote tests/golden/window_small_ext.npz", {k: np.shape(v) for k, v in ext.items()})


if __name__ == "__main__":
    main()
