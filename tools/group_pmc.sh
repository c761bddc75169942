#!/bin/bash
# experiment: fabric read traffic (FETCH_SIZE) of the f32 Gram kernel vs work-item K length and launch-order block
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
for gt in 4096 2048 1024; do
for xb in 36 144; do
  rm -rf gpurun_out/gpmc
  GAUSS_GROUP_TARGET=$gt GAUSS_XCD_BLOCK=$xb rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/gpmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-i8-variant > gpurun_out/gpmc.json 2> gpurun_out/gpmc.log
  python3 - <<PY
import csv,glob,json
f=glob.glob("gpurun_out/gpmc/**/*counter_collection.csv",recursive=True)[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"]=="FETCH_SIZE" and "gram_kernel" in r["Kernel_Name"]]
d=json.loads(open("gpurun_out/gpmc.json").readlines()[-1])
print("group $gt xcd $xb: gram FETCH_SIZE GB", round(sum(v)/len(v)/1e6,2), "gram ms (under pmc)", round(d["stage_ms_per_step"]["gram"],2))
PY
done; done
