#!/usr/bin/env python3
"""A/B harness for the streamed blocking call (gauss_impute_window on host bytes): the configurations given on the command
line ("NAME=VALUE,NAME=VALUE" each; "-" = defaults) are run interleaved, REPS times each, in ONE process on one mean-sized
chr22 window (M = 736, U = 2526, N = 32 147: 105 MB of genotype bytes); prints min / median per configuration."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from gauss_amd import hotpath, synth  # noqa: E402

KNOBS = ("GAUSS_STREAM_WINDOW", "GAUSS_TRACE")      # (round 5: the chunking knobs of the round-3 A/Bs are constants now)
configs = sys.argv[1:] or ["-"]
reps = int(os.environ.get("REPS", "15"))
pops = [p for p in synth.POPS_33KG if p[0] in synth.PGC2_WEIGHTS]
off = synth.pop_offsets([p[1] for p in pops])
w = np.array([synth.PGC2_WEIGHTS[p[0]] for p in pops])
N = int(off[-1])
M, U = 736, 2526
rng = np.random.default_rng(0)
af = rng.uniform(0.05, 0.95, size=(M + U, 1))
G = (rng.random((M + U, N)) < af).astype(np.uint8) + (rng.random((M + U, N)) < af).astype(np.uint8)
gm, gu = np.ascontiguousarray(G[:M]), np.ascontiguousarray(G[M:])
z1 = rng.standard_normal(M)
ctx = hotpath.Context(0)


def setenv(cfg):
    for k in KNOBS:
        os.environ.pop(k, None)
    if cfg != "-":
        for kv in cfg.split(","):
            k, v = kv.split("=")
            os.environ[k] = v


ref = None
times = {c: [] for c in configs}
for r in range(reps + 2):
    for c in configs:
        setenv(c)
        t0 = time.perf_counter()
        res = hotpath.impute_window(1, gm, gu, off, w, z1, ctx=ctx)
        dt = (time.perf_counter() - t0) * 1e3
        if r >= 2:
            times[c].append(dt)
        if ref is None:
            ref = res
        assert np.array_equal(res["z"], ref["z"]) and np.array_equal(res["info"], ref["info"]), c
for c in configs:
    t = np.array(times[c])
    print("%-70s min %.3f  median %.3f  max %.3f ms" % (c, t.min(), np.median(t), t.max()))
