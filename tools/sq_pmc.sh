#!/bin/bash
# where do the waves of each kernel spend their cycles?  (SQ counters, one PMC pass; MI355X_MICROARCH.md:
# WAIT_ANY = parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing)
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
rm -rf gpurun_out/sqpmc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d gpurun_out/sqpmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-i8-variant > gpurun_out/sqpmc.json 2> gpurun_out/sqpmc.log
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/sqpmc/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")[:40]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items():
    if "gauss" not in k: continue
    w=v.get("SQ_WAVE_CYCLES",0) or 1
    print(k.ljust(42), " ".join("%s=%.2f"%(c.replace("SQ_",""), v.get(c,0)/w) for c in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_WAIT_INST_LDS")))
PY
