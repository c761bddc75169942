#!/bin/bash
# SQ counters of the Gram kernel(s) of a short bench run: where do the waves wait, how busy are the matrix pipes?
# usage (through gpurun): bash tools/gram_pmc.sh <tag> [ENV=VALUE ...]   -> gpurun_out/<tag>.txt
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
rm -rf gpurun_out/gpmc_$TAG
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/gpmc_$TAG -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-i8-variant --no-e2e > gpurun_out/gpmc_$TAG.json 2> gpurun_out/gpmc_$TAG.log
python3 - "$TAG" <<'PY' > gpurun_out/$TAG.txt
import csv,glob,collections,sys
tag=sys.argv[1]
f=glob.glob("gpurun_out/gpmc_%s/**/*counter_collection.csv"%tag,recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter(); seen=set()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")[:48]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); n[k]+=1
for k,v in sorted(acc.items()):
    if "gauss" not in k: continue
    w=v.get("SQ_WAVE_CYCLES",0) or 1
    b=v.get("SQ_BUSY_CYCLES",0) or 1
    print(k.ljust(50), "n=%d"%n[k], "parked=%.3f stall=%.3f issuing=%.3f lds_wait=%.3f mfma_util=%.3f valu_insts/launch=%.3e" % (
        v["SQ_WAIT_ANY"]/w, v["SQ_WAIT_INST_ANY"]/w, v["SQ_ACTIVE_INST_ANY"]/w, v["SQ_WAIT_INST_LDS"]/w, v["SQ_VALU_MFMA_BUSY_CYCLES"]/(32*b), v["SQ_INSTS_VALU"]/max(n[k],1)))
PY
cat gpurun_out/$TAG.txt
