#!/usr/bin/env python3
"""Gram-kernel efficiency probe: perfectly tile-aligned windows (no padding, no skipped halves)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gauss_amd import hotpath

M, U, N, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
mode = int(sys.argv[5]) if len(sys.argv) > 5 else 0
rng = np.random.default_rng(0)
base = (rng.random((M + U, N)) < 0.3).astype(np.uint8) + (rng.random((M + U, N)) < 0.3).astype(np.uint8)
off = np.array([0, N], dtype=np.int32) if mode == 0 else np.arange(0, N + 1, N // 16, dtype=np.int32)
w = np.full(len(off) - 1, 1.0 / (len(off) - 1))
wins = [dict(mode=mode, geno_m=base[:M], geno_u=base[M:], pop_off=off, pop_wgt=w, z1=rng.standard_normal(M)) for _ in range(W)]
ctx = hotpath.Context(0)
job = hotpath.Job(wins, ctx=ctx)
for _ in range(2):
    job.run(); job.fetch()
job.profile(True)
for _ in range(5):
    job.run(); job.fetch()
ms, n = job.profile_get(0)
st, wk = job.stats(), job.work()
print(json.dumps(dict(M=M, U=U, N=N, W=W, mode=mode, gram_ms=ms / n, issued_tflops=st["executed_flops"] / (ms / n * 1e-3) / 1e12,
                      alg_tflops=wk["ld_flops"] / (ms / n * 1e-3) / 1e12, items=st["items"])))
