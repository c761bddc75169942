"""Driver of tools/coresident_probe.hip: a chain of small-footprint dependent launches alone and beside the bench job.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/libcoprobe.so tools/coresident_probe.hip
    python3 tools/coresident_probe.py [--windows 5]

Prints the Gram kernel's time per step and the chain's duration for: job alone, chain alone, both together (the chain queued right
after a step's kernels, so it runs under the job's Gram kernel), for a few chain shapes and LDS sizes."""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=0)
    ap.add_argument("--lib", default="/tmp/libcoprobe.so")
    a = ap.parse_args()
    args = bench.parse_args(["--no-cpu-baseline", "--no-i8-variant", "--no-e2e", "--windows", str(a.windows)])
    from gauss_amd import workload
    rig = bench.Rig(args)
    ch = workload.make_chromosome(args.snps, args.mode, seed=20260216, sample_scale=1.0)
    wins = workload.windows_of(ch, args.wing, args.windows)
    panel, ld = bench.synth_panel(rig, ch, 20260216)
    store, ld2 = bench.pack_store(rig, ch, panel, ld)
    del panel
    rig.torch.cuda.synchronize()
    rig.ctx.set_gram_dtype("f32")
    runner = bench.Runner(rig, bench.window_descs(ch, wins, store, ld2, args.mode), 1)
    lib = C.CDLL(a.lib)
    lib.coprobe_create.restype = C.c_void_p
    lib.coprobe_create.argtypes = [C.c_int, C.c_int]
    lib.coprobe_chain.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.coprobe_ms.restype = C.c_float
    lib.coprobe_ms.argtypes = [C.c_void_p, C.c_int]
    print("lite_step VGPRs:", lib.coprobe_vgprs(), flush=True)

    def job_alone(steps=10):
        dt, st, _ = runner.timed(steps, 3)
        return dt / steps * 1e3, st["gram"][0] / max(1, st["gram"][1]), {k: v[0] / steps for k, v in st.items()}

    ms, gram, st = job_alone()
    print("job alone: step %.3f ms, gram %.3f ms" % (ms, gram), {k: round(v, 3) for k, v in st.items()}, flush=True)
    for prio in (1,):
        pr = lib.coprobe_create(1024, prio)
        for (n_launch, n_wg, serial, wprio) in ((40, 40, 64, 0), (40, 40, 64, 1), (40, 400, 64, 1), (40, 40, 0, 1), (1, 1, 0, 1), (4, 40, 64, 1), (200, 40, 64, 1)):
            idx = [lib.coprobe_chain(pr, n_launch, n_wg, serial, wprio) for _ in range(3)]
            alone = min(lib.coprobe_ms(pr, i) for i in idx)
            # together: queue the chain right after each step's kernels
            for _ in range(3):
                runner.step()
            runner.drain()
            runner.profile(True)
            steps = 8
            t0 = time.perf_counter()
            idx = []
            for _ in range(steps):
                runner.step()
                idx.append(lib.coprobe_chain(pr, n_launch, n_wg, serial, wprio))
            runner.drain()
            co = [lib.coprobe_ms(pr, i) for i in idx]
            dt = (time.perf_counter() - t0) / steps * 1e3
            stg = runner.stage_ms()
            runner.profile(False)
            g = stg["gram"][0] / max(1, stg["gram"][1])
            print("stream priority %s, s_setprio %d, chain %d launches x %d workgroups, serial %d: alone %.3f ms (%.1f us per launch); beside the job %s ms; "
                  "step %.3f ms, gram %.3f ms" % ("high" if prio else "low", 3 * wprio, n_launch, n_wg, serial, alone, alone / n_launch * 1e3,
                                                  " ".join("%.2f" % c for c in co), dt, g), flush=True)
    ms, gram, st = job_alone()
    print("job alone again: step %.3f ms, gram %.3f ms" % (ms, gram), flush=True)
    runner.close()
    rig.close()


if __name__ == "__main__":
    main()
