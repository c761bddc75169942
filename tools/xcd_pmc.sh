#!/bin/bash
# experiment: fabric read traffic of the Gram kernel vs launch order (FETCH_SIZE pass only)
set -e
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
for b in 36 6; do
  rm -rf gpurun_out/xcdpmc_$b
  GAUSS_XCD_BLOCK=$b rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/xcdpmc_$b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-i8-variant > gpurun_out/xcdpmc_$b.json 2> gpurun_out/xcdpmc_$b.log
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/xcdpmc_$b/**/*counter_collection.csv",recursive=True)[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"]=="FETCH_SIZE" and "gram_kernel" in r["Kernel_Name"]]
print("block $b: gram FETCH_SIZE KB avg", sum(v)/len(v), "launches", len(v))
PY
done
