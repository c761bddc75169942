"""Where one rank's share of the files -> table path spends its time: gauss_host_impute_chromosome(rank, world) on the chr22-sized
packed panel with GAUSS_TRACE=chrom, warm, as `bench.py`'s end_to_end.emulated_world8 times it.
    python tools/e2e_rank_trace.py [rank] [world] [n_batches] > gpurun_out/e2e_rank_trace.txt 2>&1"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    n_batches = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    import bench
    from gauss_amd import api, benchmodes, workload
    args = bench.parse_args(["--no-cpu-baseline"])
    rig = bench.Rig(args)
    ch = workload.make_chromosome(args.snps, "distmix", seed=20260216, sample_scale=1.0)
    tmp = tempfile.mkdtemp(prefix="gauss_trace_")
    files = benchmodes.write_study_files(rig, ch, tmp)
    sa = benchmodes.study_args(ch, files)
    lo, hi = benchmodes.chromosome_span(ch)
    kw = dict(chr=22, start_bp=lo, end_bp=hi, wing_size=args.wing, input_file=files["gwas"], reference_data_file=files["panel"],
              reference_pop_desc_file=files["desc"], rank=rank, world=world, n_batches=n_batches, ctx=rig.ctx, **sa)
    for k in range(3):
        api.impute_chromosome(**kw)
    os.environ["GAUSS_TRACE"] = "chrom,job"
    for k in range(2):
        t0 = time.perf_counter()
        res = api.impute_chromosome(**kw)
        print("call %d: %.3f ms wall, stats %s" % (k, (time.perf_counter() - t0) * 1e3, {a: (round(b * 1e3, 3) if a.startswith("t_") else b) for a, b in res.stats.items()}), file=sys.stderr)
    rig.close()


if __name__ == "__main__":
    main()
