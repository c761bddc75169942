#!/bin/bash
# kernel + copy timeline of the last blocking gauss_impute_window call of tools/window_trace.py (streamed form)
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
rm -rf gpurun_out/wtl
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/wtl -- python3 tools/window_trace.py > gpurun_out/wtl.log 2>&1
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob("gpurun_out/wtl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void gauss::", "")[:38]))
for f in glob.glob("gpurun_out/wtl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:24]))
ev.sort()
# last call: events after the last gap > 1 ms preceding the final pack_stats burst
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[max(0, i - 50):i]) > 300_000:
        cut = i
sel = ev[cut:]
t0 = sel[0][0]
for s, e, n in sel:
    print("%9.1f us  +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
