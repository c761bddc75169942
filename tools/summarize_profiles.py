"""Boil the rocprofv3 output of tools/collect_profiles.sh down to the small CSVs kept under profiles/.

usage: summarize_profiles.py <tag> <dir>     (dir = gpurun_out/prof_<tag>_<utc>: collect_profiles.sh never reuses or deletes one)
Writes <dir>/summary/{tag}_kernel_stats.csv, {tag}_pmc_traffic.csv, {tag}_sq_mfma.csv, {tag}_step_timeline.txt, {tag}_bench_under_rocprof.json and
{tag}_provenance.json (source hash + git head of the profiled code);
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


def pmc_by_grid(dirname, counter):
    """(kernel, grid size in threads) -> [counter value per dispatch]: a kernel launched on grids of different sizes (the Gram kernel:
    the headline job's full grid, the two halves of the two-launch form, the emulated shares) has no meaningful average over them."""
    f = find(dirname, "*counter_collection.csv")
    per = defaultdict(list)
    if not f:
        return per
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") == counter:
                per[(short(row["Kernel_Name"]), int(row.get("Grid_Size") or 0))].append(float(row["Counter_Value"]))
    return per


def gram_forms(grids):
    """Names for the Gram kernel's grids in a PMC pass of the bench command.  Counter collection serialises the hardware queues, the
    library's queue probe (gauss_hip_init) therefore fails and the headline runs are queued in the TWO-launch form (B11's items, then
    B21's); the one-stream pass launches the full grid in one piece -- the grid the merged form launches when nothing is profiling."""
    out = {}
    gs = sorted(grids)
    if not gs:
        return out
    full = gs[-1]
    out[full] = "full grid in ONE launch (the grid of the headline's merged launch; here from the one-stream pass, nothing beside it)"
    for a in gs:
        for b in gs:
            if a < b and a + b == full:
                out[a] = "two-launch form, first launch (B11's items): the headline runs are demoted to this form under counter collection"
                out[b] = "two-launch form, second launch (B21's items)"
    for g in gs:
        out.setdefault(g, "other job of the same command")
    return out


def pmc_all(dirname):
    """kernel -> counter -> (sum over dispatches, dispatches).  The f32 Gram kernel is ALSO listed per grid size, as
    `gauss::gram_kernel<float>@<grid threads>`: the profiled command launches it on the full grid, on the two halves of the
    two-launch form (the form the headline runs are demoted to under counter collection) and on the emulated shares, and per-launch
    counts averaged over those describe none of them (ratios such as mfma_util are fine either way)."""
    f = find(dirname, "*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    if f:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                keys = [k] + ([f"{k}@{int(row.get('Grid_Size') or 0)}"] if k.startswith("gauss::gram_kernel<float>") else [])
                for kk in keys:
                    a = acc[kk][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    return acc


def sq_mfma(tag, d, out):
    """Wave-state fractions and matrix-core utilisation per kernel (MI355X_MICROARCH.md, rocprofv3 PMC slots):
    parked = SQ_WAIT_ANY, issue stall = SQ_WAIT_INST_ANY, issuing = SQ_ACTIVE_INST_ANY, each / SQ_WAVE_CYCLES
    (all three in quad-cycles).  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES): the busy cycles of the
    1024 matrix pipes (256 CUs x 4 SIMDs, summed) over 1024 x the kernel's duration in cycles; SQ_BUSY_CYCLES is
    reported summed over the 32 shader engines, so duration = SQ_BUSY_CYCLES / 32 (cross-check on this chip:
    GRBM_GUI_ACTIVE, summed over the 8 XCDs, / 8 gives the same duration within 3 %).  For the f32 Gram kernel the
    figure equals issued MFMA flops / peak (one v_mfma_f32_32x32x2_f32 holds a pipe for 64 cycles)."""
    a = pmc_all(os.path.join(d, "pmc_sq"))
    b = pmc_all(os.path.join(d, "pmc_sq2"))
    if not a:
        return
    cols = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
            "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_F64"]
    cols2 = ["GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
             "SQ_VALU_MFMA_COEXEC_CYCLES"]
    with open(os.path.join(out, f"{tag}_sq_mfma.csv"), "w") as fh:
        fh.write("kernel,launches,parked_frac,issue_stall_frac,issuing_frac,mfma_util," +
                 ",".join(c + "_per_launch" for c in cols + cols2) + ",launch_form\n")
        forms = gram_forms({int(k.split("@")[1]) for k in a if "@" in k})
        for k in sorted(a):
            if "gauss" not in k:
                continue
            v = a[k]
            n = max(1, v["SQ_WAVE_CYCLES"][1])
            w = v["SQ_WAVE_CYCLES"][0] or 1.0
            busy = v["SQ_BUSY_CYCLES"][0] or 1.0
            row = [k, str(n), "%.3f" % (v["SQ_WAIT_ANY"][0] / w), "%.3f" % (v["SQ_WAIT_INST_ANY"][0] / w),
                   "%.3f" % (v["SQ_ACTIVE_INST_ANY"][0] / w), "%.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (32.0 * busy))]
            row += ["%.0f" % (v[c][0] / n) for c in cols]
            vb = b.get(k, {})
            row += ["%.0f" % (vb[c][0] / max(1, vb[c][1])) if c in vb else "" for c in cols2]
            row.append('"%s"' % (forms.get(int(k.split("@")[1]), "") if "@" in k else ("all launches of the kernel in the profiled command" if k.startswith("gauss::gram_kernel") else "")))
            fh.write(",".join(row) + "\n")


def by_grid(tag, d, out):
    """Launch statistics per (kernel, grid size) from the kernel trace: the default bench command launches the same kernel on the
    36-window headline job, on the eight emulated 8-rank shares and in the one-stream pass -- rocprofv3's own --stats table
    averages a kernel's launches over all of them; here the headline job's launches are a line of their own."""
    f = find(os.path.join(d, "stats"), "*kernel_trace.csv")
    if not f:
        return
    acc = defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "gauss" not in r["Kernel_Name"]:
                continue
            grid = r.get("Grid_Size_X") or r.get("Grid_Size") or ""
            acc[(short(r["Kernel_Name"]), grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(os.path.join(out, f"{tag}_kernel_stats_by_grid.csv"), "w") as fh:
        fh.write("kernel,grid_size_x_threads,calls,avg_us,min_us,max_us,total_ms\n")
        for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            if sum(v) < 200.0 and len(v) < 4:        # (the long tail of one-off launches)
                continue
            fh.write(f"{k},{g},{len(v)},{sum(v) / len(v):.1f},{min(v):.1f},{max(v):.1f},{sum(v) / 1e3:.3f}\n")


def outliers(tag, d, out):
    """A Gram launch that takes more than 1.2 x the median of its grid's launches (profiles/README.md: one launch of 57 ms among 37
    under --kernel-trace in rounds 3 and 4, never outside the profiler) gets a report of its own, so that the evidence outlives the
    collection: every dispatch of every queue that overlaps the launch (1 ms of margin either side) with start / end relative to
    the launch's start, per queue the longest interval without a running dispatch inside the launch, and the clocks rocm-smi
    showed before and after the traced run.  Returns the number of outliers (0: the report says so)."""
    f = find(os.path.join(d, "stats"), "*kernel_trace.csv")
    path = os.path.join(out, f"{tag}_outlier.txt")
    if not f:
        return 0
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", ""),
                         r.get("Grid_Size_X") or r.get("Grid_Size") or "", r.get("Dispatch_Id", "")))
    rows.sort()
    by = defaultdict(list)
    for r in rows:
        if "gram_kernel" in r[2]:
            by[(r[2], r[4])].append(r)
    found = []
    for key, v in by.items():
        if len(v) < 5:
            continue
        dur = sorted(e - s for s, e, *_ in v)
        med = dur[len(dur) // 2]
        found += [(r, med, len(v)) for r in v if (r[1] - r[0]) > 1.2 * med]
    with open(path, "w") as fh:
        fh.write("# Gram launches longer than 1.2 x the median of their grid in the --kernel-trace run of this collection (tools/summarize_profiles.py)\n")
        if not found:
            fh.write("none: %s\n" % "; ".join("%s grid %s: %d launches, median %.3f ms, longest %.3f ms" % (k[0], k[1], len(v), sorted(e - s for s, e, *_ in v)[len(v) // 2] / 1e6,
                                                                                                            max(e - s for s, e, *_ in v) / 1e6) for k, v in by.items() if len(v) >= 5))
        for (s0, e0, name, q0, grid, did), med, n in found:
            fh.write("\n== %s grid %s dispatch %s on queue %s: %.3f ms (median of %d launches %.3f ms)\n" % (name, grid, did, q0, (e0 - s0) / 1e6, n, med / 1e6))
            near = [r for r in rows if r[1] >= s0 - 1_000_000 and r[0] <= e0 + 1_000_000]
            fh.write("dispatches overlapping it (ms from its start):\n")
            for s, e, nm, q, g, di in near:
                fh.write("  queue %-3s %-44s grid %-10s start %9.3f end %9.3f (%8.3f ms)\n" % (q, nm[:44], g, (s - s0) / 1e6, (e - s0) / 1e6, (e - s) / 1e6))
            fh.write("per queue, the longest stretch INSIDE the launch with no dispatch of that queue running:\n")
            for q in sorted({r[3] for r in near}):
                iv = sorted((max(s, s0), min(e, e0)) for s, e, _, qq, _, _ in near if qq == q and e > s0 and s < e0)
                t, gap, at = s0, 0, s0
                for s, e in iv:
                    if s - t > gap:
                        gap, at = s - t, t
                    t = max(t, e)
                if e0 - t > gap:
                    gap, at = e0 - t, t
                fh.write("  queue %-3s idle %.3f ms from %.3f ms\n" % (q, gap / 1e6, (at - s0) / 1e6))
        for nm in ("smi_before.txt", "smi_after_trace.txt"):
            sp = os.path.join(d, nm)
            if os.path.exists(sp):
                fh.write("\n-- %s\n" % nm)
                with open(sp) as sf:
                    fh.write("".join(l for l in sf if any(w in l.lower() for w in ("sclk", "mclk", "fclk", "perf", "temp")))[:4000])
    return len(found)


def provenance(tag, out):
    """Which code the profiles were taken on: the source hash compiled into the profiled library (gauss_hip_source_hash)
    and the git head the development container recorded for those sources (gauss_amd/lib/build_stamp.json; the GPU box
    has no .git).  bench.py quotes roofline.traffic from <tag>_pmc_traffic.csv only while this hash equals the hash of
    the library it is running."""
    import datetime
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stamp = {}
    try:
        with open(os.path.join(root, "gauss_amd", "lib", "build_stamp.json")) as fh:
            stamp = json.load(fh)
    except Exception:
        pass
    try:
        sys.path.insert(0, root)
        from gauss_amd import _lib
        stamp["csrc_hash_of_loaded_library"] = _lib.load().gauss_hip_source_hash().decode()
    except Exception as ex:      # the summary can be rebuilt off the GPU box
        stamp["csrc_hash_of_loaded_library"] = None
    # the end_to_end figures also depend on libgauss_host.so: its identity is the hash of its sources (it has no compiled-in one)
    try:
        import hashlib
        h = hashlib.sha256()
        hdir = os.path.join(root, "gauss_amd", "csrc", "host")
        for f in sorted(os.listdir(hdir)) + ["../../../include/gauss_host.h"]:
            with open(os.path.join(hdir, f), "rb") as fh:
                h.update(os.path.basename(f).encode() + b"\0" + fh.read())
        stamp["host_src_hash"] = h.hexdigest()[:16]
    except Exception:
        stamp["host_src_hash"] = None
    stamp["tag"] = tag
    stamp["collected_utc"] = datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ")
    stamp["commands"] = "tools/collect_profiles.sh " + tag
    with open(os.path.join(out, f"{tag}_provenance.json"), "w") as fh:
        json.dump(stamp, fh, indent=1)
    return stamp


def main():
    tag, d = sys.argv[1], sys.argv[2]
    out = os.path.join(d, "summary")
    os.makedirs(out, exist_ok=True)
    stamp = provenance(tag, out)
    ks = find(os.path.join(d, "stats"), "*kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(out, f"{tag}_kernel_stats.csv"))
    by_grid(tag, d, out)
    n_out = outliers(tag, d, out)
    if n_out:
        print("NOTE: %d Gram launch(es) over 1.2 x the median: see %s_outlier.txt (keep %s)" % (n_out, tag, d))
    bj = os.path.join(d, "bench_under_rocprof.json")
    if os.path.exists(bj):
        with open(bj) as fh:
            lines = [l for l in fh if l.startswith("{")]
        with open(os.path.join(out, f"{tag}_bench_under_rocprof.json"), "w") as fh:
            fh.write(lines[-1] if lines else "")
    fetch = pmc(os.path.join(d, "pmc_fetch"), "FETCH_SIZE")
    write = pmc(os.path.join(d, "pmc_write"), "WRITE_SIZE")
    with open(os.path.join(out, f"{tag}_pmc_traffic.csv"), "w") as fh:
        fh.write("kernel,launches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_bytes_per_launch_corrected,csrc_hash,hbm_bytes_largest_launch_corrected\n")
        for k in sorted(set(fetch) | set(write)):
            fa = sum(fetch[k]) / max(1, len(fetch[k]))
            wa = sum(write[k]) / max(1, len(write[k]))
            # the largest launch of the kernel (a kernel launched on tile lists of different lengths -- the epilogue's early / late /
            # all-tiles launches -- has no meaningful average): largest fetch + largest write (the same launch in practice)
            fm = max(fetch[k]) if fetch[k] else 0.0
            wm = max(write[k]) if write[k] else 0.0
            # gfx950: FETCH_SIZE under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section)
            fh.write(f"{k},{max(len(fetch[k]), len(write[k]))},{fa:.1f},{wa:.1f},{int((2 * fa + wa) * 1024)},{stamp.get('csrc_hash') or ''},{int((2 * fm + wm) * 1024)}\n")
    # the same, per grid size: the row bench.py quotes is the one whose grid equals the running job's (roofline.traffic_form names it)
    fg, wg = pmc_by_grid(os.path.join(d, "pmc_fetch"), "FETCH_SIZE"), pmc_by_grid(os.path.join(d, "pmc_write"), "WRITE_SIZE")
    forms = gram_forms({g for (k, g) in set(fg) | set(wg) if k.startswith("gauss::gram_kernel<float>")})
    with open(os.path.join(out, f"{tag}_pmc_traffic_by_grid.csv"), "w") as fh:
        fh.write("kernel,grid_threads,launches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_bytes_per_launch_corrected,csrc_hash,launch_form\n")
        for k, g in sorted(set(fg) | set(wg)):
            if "gauss" not in k:
                continue
            f_, w_ = fg.get((k, g), []), wg.get((k, g), [])
            fa = sum(f_) / max(1, len(f_))
            wa = sum(w_) / max(1, len(w_))
            form = forms.get(g, "") if k.startswith("gauss::gram_kernel<float>") else ""
            fh.write(f"{k},{g},{max(len(f_), len(w_))},{fa:.1f},{wa:.1f},{int((2 * fa + wa) * 1024)},{stamp.get('csrc_hash') or ''},\"{form}\"\n")
    sq_mfma(tag, d, out)
    # one step of the headline run, kernel by kernel and queue by queue (what runs under the second Gram launch)
    try:
        import subprocess
        tl = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_timeline.py"),
                             os.path.join(d, "stats"), "8"], capture_output=True, text=True, timeout=120)
        if tl.returncode == 0 and tl.stdout.strip():
            with open(os.path.join(out, f"{tag}_step_timeline.txt"), "w") as fh:
                fh.write("# one step of the --kernel-trace run (tools/step_timeline.py): start / end per kernel and hardware queue, ms from the step's first launch\n")
                fh.write(tl.stdout)
    except Exception:
        pass
    print("summaries in", out, os.listdir(out))


if __name__ == "__main__":
    main()
