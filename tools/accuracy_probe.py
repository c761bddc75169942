"""How far are the GPU's z / info from the CPU oracle's (full-pivot LU inverse, reference operation order)?
   python tools/accuracy_probe.py        (on the GPU box)
Windows of M = 200 ... 1 200 measured SNPs on a 29-population panel of ~3 300 samples, dist and distmix."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle                                     # noqa: E402
from gauss_amd import hotpath                     # noqa: E402
from helpers import small_panel, split_window    # noqa: E402

ctx = hotpath.Context(0)
p = small_panel(n_snp=1700, scale=0.1, seed=3, span_bp=3_000_000)
for mode in (0, 1):
    for M in (200, 640, 1200):
        gm, gu, z1 = split_window(dict(G=p["G"][: M + 400]), M)
        got = hotpath.impute_window(mode, gm, gu, p["off"], p["w"], z1, ctx=ctx)
        want = oracle.run_impute(mode, gm, gu, p["off"], p["w"], z1)
        ez = np.max(np.abs(got["z"] - want["z"]) / np.maximum(1.0, np.abs(want["z"])))
        ei = np.max(np.abs(got["info"] - want["info"]) / np.maximum(1e-300, np.abs(want["info"])))
        print(f"mode {mode}  M {M:5d}  U {gu.shape[0]}  N {gm.shape[1]}:  max rel |dz| {ez:.2e}   max rel |dinfo| {ei:.2e}")
