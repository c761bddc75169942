"""Timeline of one bench step from a rocprofv3 kernel trace (csv): per kernel launch start / end relative to the step's first
launch, for one step of the headline run (the trace also holds bench.py's later one-stream pass).
usage: step_timeline.py <dir with *kernel_trace.csv> [step index, default 3]"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getsize)
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    # steps start at pack_stats_kernel
    starts = [i for i, r in enumerate(rows) if "pack_stats_kernel" in r[2]]
    if len(starts) < 3:
        print("too few steps in the trace")
        return
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    a, b = starts[k], starts[k + 1]
    t0 = rows[a][0]
    agg = {}
    for s, e, name, q in rows[a:b]:
        short = name.split("(")[0].replace("void gauss::", "")
        k = (short, q)
        if k not in agg:
            agg[k] = [s, e, 0, 0]
        agg[k][0] = min(agg[k][0], s); agg[k][1] = max(agg[k][1], e); agg[k][2] += 1; agg[k][3] += e - s
    print("step span %.3f ms" % ((rows[b][0] - t0) / 1e6))
    for (short, q), (s, e, n, busy) in sorted(agg.items(), key=lambda kv: kv[1][0]):
        print("%-46s queue %-3s launches %4d  first start %8.3f  last end %8.3f  sum of durations %8.3f ms" % (short[:46], q, n, (s - t0) / 1e6, (e - t0) / 1e6, busy / 1e6))
    # individual gram launches
    for s, e, name, q in rows[a:b]:
        if "gram_kernel" in name:
            print("   gram launch: start %.3f end %.3f (%.3f ms) queue %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q))


if __name__ == "__main__":
    main()
