#!/bin/bash
# Kernel-by-kernel timeline of the LAST step of a short bench run (rocprofv3 --kernel-trace).
#   tools/trace_step.sh OUTNAME [bench.py args...]      (through gpurun; writes gpurun_out/OUTNAME.txt)
set -e
NAME=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/trace_$NAME; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -- python3 bench.py --steps 2 --warmup 1 --no-e2e --no-cpu-baseline --no-i8-variant "$@" > "$OUT/bench.json" 2> "$OUT/log.txt"
python3 - <<PY
import csv, glob, os
f = max(glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "pack_stats_kernel" in r["Kernel_Name"])
seq = rows[last:]
t0 = int(seq[0]["Start_Timestamp"])
with open("$ROOT/gpurun_out/$NAME.txt", "w") as out:
    prev_end = t0
    for r in seq:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gauss::", "")[:34]
        out.write(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {name}  grid {r.get('Grid_Size_X', r.get('Grid_Size',''))}x{r.get('Grid_Size_Y','')}\n")
        prev_end = e
    out.write(f"total {(prev_end - t0) / 1e3:.1f} us\n")
PY
rm -rf "$OUT/kt"
