"""The files -> table call as rank r of 8, every rank timed alone (bench.py's end_to_end.emulated_world8, without the rest of the bench):
per-rank wall, GPU span, the phases.   python tools/e2e_w8_probe.py [calls]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    import bench
    from gauss_amd import api, benchmodes, workload
    args = bench.parse_args(["--no-cpu-baseline"])
    rig = bench.Rig(args)
    ch = workload.make_chromosome(args.snps, "distmix", seed=20260216, sample_scale=1.0)
    tmp = tempfile.mkdtemp(prefix="gauss_w8_")
    files = benchmodes.write_study_files(rig, ch, tmp)
    sa = benchmodes.study_args(ch, files)
    lo, hi = benchmodes.chromosome_span(ch)
    base = dict(chr=22, start_bp=lo, end_bp=hi, wing_size=args.wing, input_file=files["gwas"], reference_data_file=files["panel"],
                reference_pop_desc_file=files["desc"], ctx=rig.ctx, **sa)
    one = []
    for _ in range(calls + 2):
        t0 = time.perf_counter(); r1 = api.impute_chromosome(rank=0, world=1, **base); one.append((time.perf_counter() - t0) * 1e3)
    t1 = float(np.median(one[2:]))
    os.environ["LOCAL_WORLD_SIZE"] = "8"
    slow = 0
    res = None
    for r in range(8):
        ts = []
        for _ in range(calls):
            res = None
            t0 = time.perf_counter(); res = api.impute_chromosome(rank=r, world=8, **base); ts.append((time.perf_counter() - t0) * 1e3)
        st = res.stats
        med = float(np.median(ts)); slow = max(slow, med)
        print("rank %d: %.3f ms (all %s) span %.3f | plan %.3f feeder %.3f create %.3f gpu_wait %.3f tables %.3f total %.3f py %s" % (
            r, med, [round(t, 2) for t in ts], st["gpu_span_ms"], st["t_plan"] * 1e3, st["t_feeder_wait"] * 1e3, st["t_job_create"] * 1e3,
            st["t_gpu_wait"] * 1e3, st["t_tables"] * 1e3, st["t_total"] * 1e3, {k: round(v, 3) for k, v in st["py_ms"].items()}))
    print("one rank %.3f ms (span %.2f); slowest of 8: %.3f ms; predicted efficiency %.4f" % (t1, r1.stats["gpu_span_ms"], slow, t1 / (8 * slow)))
    rig.close()


if __name__ == "__main__":
    main()
