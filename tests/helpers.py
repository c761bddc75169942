"""Shared synthetic inputs for the tests (seeded, small enough for the loop-literal oracle)."""
import numpy as np

from gauss_amd import synth


def small_panel(n_snp=120, scale=0.02, n_pops=None, seed=7, span_bp=400_000, min_size=30):
    pops = synth.pop_table(scale=scale, min_size=min_size)
    if n_pops:
        pops = pops[:n_pops]
    rng = np.random.default_rng(seed)
    bp = np.sort(rng.choice(np.arange(1, span_bp), size=n_snp, replace=False))
    G, af = synth.synth_genotypes(bp, pops, seed=seed)
    poly = G.min(1) != G.max(1)
    G, af, bp = G[poly], af[poly], bp[poly]
    off = synth.pop_offsets([p[1] for p in pops])
    w = rng.uniform(0.01, 0.3, size=len(pops))
    w *= 1.061 / w.sum()                    # un-normalised like the PGC2 weights (sum = 1.061)
    return dict(G=np.ascontiguousarray(G), af=af, bp=bp, pops=pops, off=off, w=w, rng=rng)


def split_window(panel, n_measured, seed=3):
    rng = np.random.default_rng(seed)
    S = panel["G"].shape[0]
    idx = rng.permutation(S)
    m = np.sort(idx[:n_measured])
    u = np.sort(idx[n_measured:])
    z1 = rng.standard_normal(len(m)) * 2.0
    return np.ascontiguousarray(panel["G"][m]), np.ascontiguousarray(panel["G"][u]), z1


def relerr(a, b, floor=1e-300):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0
