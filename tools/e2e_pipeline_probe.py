"""Can one rank hide its host work by running chromosome c + 1's call beside chromosome c's?  Two host threads call
gauss_host_impute_chromosome(rank, world) on the SAME context, alternating, K calls each (the same chr22 files stand in for
consecutive chromosomes); compared with the same 2 K calls one after the other: wall time per call, and every table against the
sequential one.
    python tools/e2e_pipeline_probe.py [rank] [world] [K] > gpurun_out/e2e_pipeline_probe.txt 2>&1"""
import os
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    import bench
    from gauss_amd import api, benchmodes, workload
    args = bench.parse_args(["--no-cpu-baseline"])
    rig = bench.Rig(args)
    ch = workload.make_chromosome(args.snps, "distmix", seed=20260216, sample_scale=1.0)
    tmp = tempfile.mkdtemp(prefix="gauss_pipe_")
    files = benchmodes.write_study_files(rig, ch, tmp)
    sa = benchmodes.study_args(ch, files)
    lo, hi = benchmodes.chromosome_span(ch)
    kw = dict(chr=22, start_bp=lo, end_bp=hi, wing_size=args.wing, input_file=files["gwas"], reference_data_file=files["panel"],
              reference_pop_desc_file=files["desc"], rank=rank, world=world, n_batches=0, ctx=rig.ctx, **sa)
    for _ in range(3):
        ref = api.impute_chromosome(**kw)
    t0 = time.perf_counter()
    for _ in range(2 * K):
        api.impute_chromosome(**kw)
    seq = (time.perf_counter() - t0) / (2 * K)
    out, errs = [[], []], []

    def worker(k):
        try:
            for _ in range(K):
                out[k].append(api.impute_chromosome(**kw))
        except Exception as ex:
            errs.append(repr(ex))
    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    par = (time.perf_counter() - t0) / (2 * K)
    same = all(np.array_equal(r.columns["z"], ref.columns["z"], equal_nan=True) and np.array_equal(r.columns["info"], ref.columns["info"], equal_nan=True)
               for o in out for r in o)
    print("rank %d of %d: sequential %.3f ms per call, two threads %.3f ms per call, gpu span of one call %.3f ms; tables equal: %s; errors: %s; counters %s"
          % (rank, world, seq * 1e3, par * 1e3, ref.stats["gpu_span_ms"], same, errs, rig.ctx.counters()))
    rig.close()


if __name__ == "__main__":
    main()
