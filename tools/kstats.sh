#!/bin/bash
# per-kernel durations of a short bench run (rocprofv3 --kernel-trace --stats), gauss kernels only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pf
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-i8-variant "$@" > gpurun_out/pf.json 2> gpurun_out/pf.log
python3 - <<PY
import glob,os,csv,json
f=max(glob.glob("gpurun_out/pf/**/*kernel_stats.csv",recursive=True),key=os.path.getsize)
for r in csv.DictReader(open(f)):
    if "gauss" in r["Name"]: print(r["Name"][:60].ljust(60), r["Calls"].rjust(4), str(round(float(r["AverageNs"])/1e3,1)).rjust(9), "us  min", r["MinNs"], "max", r["MaxNs"])
d=json.loads(open("gpurun_out/pf.json").readlines()[-1]); print("step ms", round(d["ms_per_step"],3), d["stage_ms_per_step"])
PY
