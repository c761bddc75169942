"""The reference's own usage pattern, call by call: distmix() for each 1 Mb window of the chr22-sized study from the packed panel file
(gauss_host_distmix: data layer -> one window on the GPU -> table), warm.  Prints ms per call and the tables' digest.
    python tools/window_calls_probe.py [passes] > gpurun_out/window_calls.txt 2>&1"""
import os
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    import bench
    from gauss_amd import api, benchmodes, workload
    args = bench.parse_args(["--no-cpu-baseline"])
    rig = bench.Rig(args)
    ch = workload.make_chromosome(args.snps, "distmix", seed=20260216, sample_scale=1.0)
    tmp = tempfile.mkdtemp(prefix="gauss_wincalls_")
    files = benchmodes.write_study_files(rig, ch, tmp)
    sa = benchmodes.study_args(ch, files)
    lo, hi = benchmodes.chromosome_span(ch)
    wins = [(s, min(hi, s + 999_999)) for s in range(lo, hi + 1, 1_000_000)]
    for k in range(passes):
        ts, rows, crc = [], 0, 0
        for s, e in wins:
            t0 = time.perf_counter()
            try:
                df = api.distmix(22, s, e, args.wing, sa["pop_wgt_df"], files["gwas"], "(packed)", files["panel"], files["desc"], ctx=rig.ctx)
            except api.GaussError as ex:
                ts.append((time.perf_counter() - t0) * 1e3)
                continue
            ts.append((time.perf_counter() - t0) * 1e3)
            rows += len(df)
            crc = zlib.crc32(np.ascontiguousarray(df["z"].to_numpy()).tobytes(), crc)
        print("pass %d: %d calls, %.2f ms per call (median %.2f, max %.2f), %d rows, z crc %08x" % (k, len(ts), np.mean(ts), np.median(ts), max(ts), rows, crc))
    rig.close()


if __name__ == "__main__":
    main()
