#!/usr/bin/env python3
"""Stress of the merged Gram launch's hand-offs (B11's items -> chain queue, early windows' B21 items -> low-priority queue):
the 36-window headline job, N steps with two runs in flight, every step's z / info hashed and compared with the first step's.
A hand-off that let a consumer read slabs before they were visible would show as a differing step.  (through gpurun)"""
import hashlib
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gauss_amd import workload

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
args = bench.parse_args(["--no-cpu-baseline", "--no-e2e"])
rig = bench.Rig(args)
ch = workload.make_chromosome(100_000, "distmix", seed=20260216)
wins = workload.windows_of(ch, 500_000, 0)
panel, ld = bench.synth_panel(rig, ch, 20260216)
store, ld2 = bench.pack_store(rig, ch, panel, ld)
del panel
runner = bench.Runner(rig, bench.window_descs(ch, wins, store, ld2, "distmix"), 1)


def digest(res):
    h = hashlib.sha256()
    for r in res:
        h.update(r["z"].tobytes()); h.update(r["info"].tobytes())
    return h.hexdigest()


# second phase below: two DIFFERENT jobs (the chromosome's halves) stepped alternately on the same context -- their queues interleave
first, bad = None, 0
for s in range(steps):
    out = runner.step()
    if out is not None:
        d = digest(out)
        first = first or d
        bad += d != first
d = digest(runner.drain())
bad += d != first
print(f"{steps} steps of the 36-window job, two runs in flight: {bad} steps differ from the first (digest {first[:16]})")
runner.close()
half = len(wins) // 2
ra = bench.Runner(rig, bench.window_descs(ch, wins[:half], store, ld2, "distmix"), 1)
rb = bench.Runner(rig, bench.window_descs(ch, wins[half:], store, ld2, "distmix"), 1)
fa = fb = None
bad2 = 0
for s in range(steps // 2):
    oa = ra.step()
    ob = rb.step()
    if oa is not None:
        da, db = digest(oa), digest(ob)
        fa, fb = fa or da, fb or db
        bad2 += (da != fa) + (db != fb)
bad2 += (digest(ra.drain()) != fa) + (digest(rb.drain()) != fb)
print(f"{steps // 2} alternating steps of two jobs (the chromosome's halves) on one context: {bad2} differ")
ra.close(); rb.close()
sys.exit(1 if (bad or bad2) else 0)
