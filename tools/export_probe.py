"""Where a batched LD-export step spends its time: run() / fetch() split, and the export chunks' waits and copies (GAUSS_TRACE=job)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gauss_amd import benchmodes

args = bench.parse_args(["--mode", "computeLD", "--steps", "5"])
rig = bench.Rig(args)
t0 = time.perf_counter()
out, sample = benchmodes.run_computeld(args, rig)
print("forms", {k: (round(v.get("ms_per_step", v.get("ms_per_call")), 3)) for k, v in out["forms"].items()})
