#!/bin/bash
# Kernel timeline of one step of ONE emulated 8-rank share (rank $1, default 0): rocprofv3 --kernel-trace of a bench run that
# emulates only that share, then tools/step_timeline.py on one of the share's steps (the trace's last steps are the share's).
R=${1:-0}
OUT=gpurun_out/share_trace_$R
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --emulate-world 8 --emulate-rank $R --emulate-steps 12 --no-cpu-baseline --no-i8-variant --no-e2e --no-parity-spot --no-tails-alone --no-from-text > $OUT/bench.json 2> $OUT/log.txt
python3 - <<PY
import csv, glob, os
f = max(glob.glob(os.path.join("$OUT", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getsize)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(f)))
starts = [i for i, r in enumerate(rows) if "pack_stats_kernel" in r[2]]
k = len(starts) - 4                      # one of the share's last steps
a, b = starts[k], starts[k + 1]
t0 = rows[a][0]
print("share step span %.3f ms" % ((rows[b][0] - t0) / 1e6))
for s, e, name, q in rows[a:b]:
    short = name.split("(")[0].replace("void gauss::", "").replace("gauss::", "")[:44]
    print("%-44s queue %-3s start %7.3f end %7.3f  %7.3f ms" % (short, q, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
PY
