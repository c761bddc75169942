#!/usr/bin/env python3
"""PCIe-inclusive rate of the blocking host-pointer call (gauss_impute_window): host genotype
buffers in, z/info out, one full-size DISTMIX window.  Never reported as bench `value`."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gauss_amd import hotpath, synth  # noqa: E402

pops = [p for p in synth.POPS_33KG if p[0] in synth.PGC2_WEIGHTS]
off = synth.pop_offsets([p[1] for p in pops])
w = np.array([synth.PGC2_WEIGHTS[p[0]] for p in pops])
N = int(off[-1])
M, U = 737, 2407
rng = np.random.default_rng(0)
af = rng.uniform(0.05, 0.95, size=(M + U, 1))
G = (rng.random((M + U, N)) < af).astype(np.uint8) + (rng.random((M + U, N)) < af).astype(np.uint8)
gm, gu = np.ascontiguousarray(G[:M]), np.ascontiguousarray(G[M:])
z1 = rng.standard_normal(M)
ctx = hotpath.Context(0)
hotpath.impute_window(1, gm, gu, off, w, z1, ctx=ctx)
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    r = hotpath.impute_window(1, gm, gu, off, w, z1, ctx=ctx)
    ts.append(time.perf_counter() - t0)
t = float(np.median(ts))
# the same call with the genotype matrices in pinned host memory (gauss_pinned_alloc)
pm, pu = hotpath.PinnedArray(gm.shape, ctx=ctx), hotpath.PinnedArray(gu.shape, ctx=ctx)
pm.array[:] = gm
pu.array[:] = gu
hotpath.impute_window(1, pm.array, pu.array, off, w, z1, ctx=ctx)
tp = []
for _ in range(5):
    t0 = time.perf_counter()
    r2 = hotpath.impute_window(1, pm.array, pu.array, off, w, z1, ctx=ctx)
    tp.append(time.perf_counter() - t0)
t2 = float(np.median(tp))
assert np.array_equal(r["z"], r2["z"])
print(json.dumps({"window": {"M": M, "U": U, "N": N}, "blocking_call_ms": t * 1e3, "imputed_snps_per_s": U / t,
                  "pinned_call_ms": t2 * 1e3, "pinned_imputed_snps_per_s": U / t2,
                  "bytes_uploaded": int((M + U) * N), "note": "host memory in, z/info out: job build, H2D, all kernels, D2H"}))
