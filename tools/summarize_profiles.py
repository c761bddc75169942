"""Boil the rocprofv3 output of tools/collect_profiles.sh down to the small CSVs kept under profiles/.

usage: summarize_profiles.py <tag> <dir>     (dir = gpurun_out/prof_<tag>)
Writes <dir>/summary/{tag}_kernel_stats.csv, {tag}_pmc_traffic.csv, {tag}_bench_under_rocprof.json;
copy those into profiles/ (tracked)."""
import csv
import glob
import os
import re
import shutil
import sys
from collections import defaultdict


def find(d, pat):
    """The largest match: a run leaves one file per traced process, the bench process has by far the most rows."""
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return max(hits, key=os.path.getsize) if hits else None


def short(name):
    """'void gauss::gram_kernel<float __vector(16)>(gauss::Item const*)' -> 'gauss::gram_kernel<float>'"""
    name = name.replace("void ", "").strip()
    m = re.match(r"([\w:]+)(<(\w+))?", name)
    if not m:
        return name
    return m.group(1) + (f"<{m.group(3)}>" if m.group(3) else "")


def pmc(dirname, counter):
    f = find(dirname, "*counter_collection.csv")
    per = defaultdict(list)
    if not f:
        return per
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") == counter:
                per[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return per


def main():
    tag, d = sys.argv[1], sys.argv[2]
    out = os.path.join(d, "summary")
    os.makedirs(out, exist_ok=True)
    ks = find(os.path.join(d, "stats"), "*kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(out, f"{tag}_kernel_stats.csv"))
    bj = os.path.join(d, "bench_under_rocprof.json")
    if os.path.exists(bj):
        with open(bj) as fh:
            lines = [l for l in fh if l.startswith("{")]
        with open(os.path.join(out, f"{tag}_bench_under_rocprof.json"), "w") as fh:
            fh.write(lines[-1] if lines else "")
    fetch = pmc(os.path.join(d, "pmc_fetch"), "FETCH_SIZE")
    write = pmc(os.path.join(d, "pmc_write"), "WRITE_SIZE")
    with open(os.path.join(out, f"{tag}_pmc_traffic.csv"), "w") as fh:
        fh.write("kernel,launches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_bytes_per_launch_corrected\n")
        for k in sorted(set(fetch) | set(write)):
            fa = sum(fetch[k]) / max(1, len(fetch[k]))
            wa = sum(write[k]) / max(1, len(write[k]))
            # gfx950: FETCH_SIZE under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section)
            fh.write(f"{k},{max(len(fetch[k]), len(write[k]))},{fa:.1f},{wa:.1f},{int((2 * fa + wa) * 1024)}\n")
    print("summaries in", out, os.listdir(out))


if __name__ == "__main__":
    main()
