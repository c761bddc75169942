import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from gauss_amd import hotpath, synth
pops = [p for p in synth.POPS_33KG if p[0] in synth.PGC2_WEIGHTS]
off = synth.pop_offsets([p[1] for p in pops]); w = np.array([synth.PGC2_WEIGHTS[p[0]] for p in pops]); N = int(off[-1])
M, U = 737, 2407
rng = np.random.default_rng(0)
G = rng.integers(0, 3, size=(M + U, N), dtype=np.uint8)
gm, gu = np.ascontiguousarray(G[:M]), np.ascontiguousarray(G[M:]); z1 = rng.standard_normal(M)
ctx = hotpath.Context(0)
hotpath.impute_window(1, gm, gu, off, w, z1, ctx=ctx)
for _ in range(3):
    t0 = time.perf_counter(); job = hotpath.Job([dict(mode=1, geno_m=gm, geno_u=gu, pop_off=off, pop_wgt=w, z1=z1)], ctx=ctx)
    t1 = time.perf_counter(); job.run(); r = job.fetch()
    t2 = time.perf_counter(); job.close(); t3 = time.perf_counter()
    print("create %.2f ms  run+fetch %.2f ms  destroy %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))
# raw LD export (prep_qcat-style): B11 + B21 fetched to the host
for _ in range(3):
    t0 = time.perf_counter()
    r = hotpath.ld_window(1, gm, gu, off, w, lam=0.0, codings=1, ctx=ctx)
    print("ld_window (B11 %dx%d + B21 %dx%d to host) %.2f ms" % (M, M, U, M, (time.perf_counter() - t0) * 1e3))
