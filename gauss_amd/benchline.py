"""The ONE JSON line bench.py prints last, kept small enough for any reader of its stdout.

bench.py measures far more than the line's contract asks for (per-rank tables of the emulated 8-rank runs, the phase
statistics of the files -> table calls, the definitions of every figure).  All of that is the DETAIL: it is written to
`bench_detail.json` beside bench.py (and into `gpurun_out/`, which is what travels back from a GPU box); stderr only says where.  The final stdout line holds the
contract's keys and the headline numbers of every block only, and `final_line` refuses to return more than LIMIT bytes
(round 5's 28 KB line could not be parsed by the driver; tests/test_benchline.py builds the line from a committed sample
of the detail and checks its size and keys).

Nothing here touches the GPU or the oracle: plain dict work, testable anywhere.
"""
import json
import math

LIMIT = 6000          # bytes of the final line (the driver keeps the last 8 KB of stdout)
HARD_LIMIT = 8000     # final_line() raises above this

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def sig(x, n=6):
    """Floats to n significant digits (ints, bools, None, strings untouched)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if not math.isfinite(x) or x == 0.0:
        return x
    return float(f"{x:.{n}g}")


def _round(o, n=6):
    if isinstance(o, dict):
        return {k: _round(v, n) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_round(v, n) for v in o]
    return sig(o, n)


def pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def clip(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 1].rstrip() + "~"


def _roofline(r, extra=()):
    if not isinstance(r, dict):
        return None
    out = pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_form") + tuple(extra))
    if "traffic_form" in out:
        out["traffic_form"] = clip(out["traffic_form"], 64)
    if "kernel" in r:
        out["kernel"] = clip(r["kernel"], 48)
    if "traffic" not in out:
        out["traffic"] = None
    return out


def _cpu(c, n_sample=160):
    if not isinstance(c, dict):
        return None
    out = pick(c, ("value", "unit", "cores", "kind", "host_cores"))
    if "sample" in c:
        out["sample"] = clip(c["sample"], n_sample)
    wp = c.get("windows_in_parallel")
    if isinstance(wp, dict):
        out["all_cores"] = pick(wp, ("value", "cores"))
    return out


def _parity(p):
    if not isinstance(p, dict):
        return None
    return pick(p, ("max_rel_z", "max_rel_info", "max_abs_ld_diff", "tolerance", "ok"))


def _other(name, c):
    """One of BASELINE.json's other configs: value, unit, ms_per_step, roofline.frac, cpu_baseline.value, parity ok."""
    if not isinstance(c, dict):
        return None
    out = pick(c, ("value", "unit", "ms_per_step", "steps"))
    r = c.get("roofline") or {}
    out["roofline"] = pick(r, ("bound", "achieved", "peak", "unit", "frac"))
    cb = c.get("cpu_baseline") or {}
    if cb:
        out["cpu_baseline"] = pick(cb, ("value", "unit", "cores", "kind"))
    ps = c.get("parity_spot") or {}
    out["parity_ok"] = bool(ps.get("ok", False))
    if name == "computeLD":
        f = c.get("forms") or {}
        one, bat, blk = f.get("one_resident_window") or {}, f.get("batched_resident_windows") or {}, f.get("blocking_gauss_ld_host_bytes") or {}
        out["one_resident_window"] = pick(one, ("ms_per_step", "gram_ms", "gram_tflops", "frac"))
        out["batched"] = pick(bat, ("windows", "ms_per_step", "gram_ms", "gram_tflops", "ld_matrices_per_s"))
        out["blocking_call_ms"] = blk.get("ms_per_call")
        out["results_identical_across_forms"] = (c.get("config") or {}).get("results_identical_across_forms")
    if name == "dist":
        out["launch_form"] = c.get("launch_form")
    if name == "jepegmix":
        b = c.get("breakdown") or {}
        out["breakdown"] = pick(b, ("cold_call_s", "warm_call_s_median", "host_data_layer_s", "gpu_gene_ld_batch_ms"))
        out["timed_with_gc_off"] = c.get("timed_with_gc_off", True)
        e = c.get("emulated_world8")
        if isinstance(e, dict):
            out["emulated_world8"] = pick(e, ("world", "slowest_ms", "one_rank_ms", "predicted_efficiency", "tables_identical_to_one_rank"))
            wc = e.get("whole_calls")
            if isinstance(wc, dict):
                out["emulated_world8"]["whole_calls"] = pick(wc, ("calls", "slowest_ms", "one_rank_ms", "predicted_efficiency"))
    if name == "int8_exact":
        out.update(pick(c, ("bit_identical_to_f32_path", "gram_ms")))
    return out


def compact(d):
    """The final line from the detail dict of a headline (distmix / dist) run."""
    out = pick(d, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_longest_step", "higher_is_better", "scaling",
                   "dtype", "data", "launch_form", "result_digest"))
    out["vs_baseline"] = d.get("vs_baseline")
    cfg = d.get("config") or {}
    out["config"] = pick(cfg, ("windows", "snps", "samples", "imputed_snps_per_step", "windows_per_rank", "shard", "cut_windows", "load_imbalance",
                               "windows_flagged", "all_finite", "shards_bit_identical_to_one_rank"))
    out["config"]["workload"] = clip(cfg.get("workload", ""), 200)
    out["roofline"] = _roofline(d.get("roofline"), ("algorithmic_flops_per_launch", "avg_launch_ms", "launches", "launches_per_step", "frac_alone"))
    for k, extra in (("roofline_pack", ("algorithmic_bytes_per_launch", "launch_ms", "moved_gbs")),
                     ("roofline_epilogue", ("algorithmic_bytes_per_launch", "launch_ms", "moved_gbs")),
                     ("roofline_solve", ("algorithmic_flops_per_step", "ms_per_step"))):
        if isinstance(d.get(k), dict):
            r = _roofline(d[k], extra)
            r.pop("traffic_source", None)
            out[k] = r
    if "stage_ms_per_step" in d:
        out["stage_ms_per_step"] = d["stage_ms_per_step"]
    if "cpu_baseline" in d:
        out["cpu_baseline"] = _cpu(d["cpu_baseline"])
    if "parity_spot" in d:
        out["parity_spot"] = _parity(d["parity_spot"])
    if isinstance(d.get("weak_scaling"), dict):
        out["weak_scaling"] = pick(d["weak_scaling"], ("scaling", "value", "unit", "ms_per_step"))
    e = d.get("emulated_strong_scaling")
    if isinstance(e, dict):
        out["emulated_strong_scaling"] = pick(e, ("world", "slowest_rank_ms", "one_gpu_ms", "predicted_efficiency", "load_imbalance", "shard",
                                                 "pieces_bit_identical_to_one_job"))
    e = d.get("end_to_end")
    if isinstance(e, dict):
        blk = pick(e, ("imputed_snps", "windows", "warm_s_median", "cold_s", "imputed_snps_per_s_warm", "gpu_span_ms", "all_finite"))
        w8 = e.get("emulated_world8")
        if isinstance(w8, dict):
            blk["emulated_world8"] = pick(w8, ("world", "slowest", "one_rank_warm_ms", "predicted_efficiency", "result_identical_to_one_rank"))
            ph = w8.get("phase_ms")
            if isinstance(ph, dict):
                blk["emulated_world8"]["phase_ms"] = {k: v for k, v in ph.items() if k != "why"}
            h = w8.get("host_ms_not_overlapped")
            if isinstance(h, list) and h:
                blk["emulated_world8"]["host_ms_not_overlapped_max"] = max(h)
            a = w8.get("with_eight_host_threads_per_rank")
            if isinstance(a, dict):
                blk["emulated_world8"]["predicted_efficiency_8_host_threads"] = a.get("predicted_efficiency")
            g = w8.get("genome_pipeline")
            if isinstance(g, dict):
                blk["genome_pipeline"] = pick(g, ("predicted_efficiency", "slowest", "one_rank_ms_per_chromosome"))
        ft = e.get("from_text")
        if isinstance(ft, dict):
            blk["from_text"] = pick(ft, ("pack_s", "cold_from_text_s", "warm_s_median", "feeder_inflated_MB_per_s", "same_table_cold_and_warm"))
        out["end_to_end"] = blk
    oc = d.get("other_configs")
    if isinstance(oc, dict):
        blk = {k: _other(k, oc[k]) for k in ("computeLD", "dist", "jepegmix", "int8_exact") if k in oc}
        blk["all_parity_ok"] = oc.get("all_parity_ok")
        out["other_configs"] = blk
    i8 = d.get("int8_exact_variant")
    if isinstance(i8, dict) and not (isinstance(oc, dict) and "int8_exact" in oc):
        out["int8_exact_variant"] = pick(i8, ("ms_per_step", "gram_ms", "gram_tops_algorithmic", "bit_identical_to_f32_path"))
    out["detail"] = d.get("detail_file", "bench_detail.json")
    out = _round(out)
    for k in ("value", "ms_per_step"):          # the contract's two figures at full precision
        if k in d:
            out[k] = d[k]
    return out


def shrink(o, max_str=160, max_list=8):
    """Generic reduction for the lines of the other modes: long strings clipped, long lists summarised, keys that are prose dropped."""
    if isinstance(o, dict):
        return {k: shrink(v, max_str, max_list) for k, v in o.items()
                if not (k in ("note", "what", "definition", "stage_note", "warm_definition", "alone_note", "per_rank", "per_window") or k.startswith("stats_"))}
    if isinstance(o, (list, tuple)):
        if len(o) > max_list and all(isinstance(v, (int, float)) and not isinstance(v, bool) for v in o):
            return {"n": len(o), "min": sig(float(min(o))), "max": sig(float(max(o)))}
        return [shrink(v, max_str, max_list) for v in o[:max_list]]
    if isinstance(o, str):
        return clip(o, max_str)
    return sig(o)


def final_line(detail, headline=True):
    """The string to print last.  headline: a distmix / dist line (compact()); otherwise the generic shrink, and if even that is too
    long only the contract's keys survive."""
    line = compact(detail) if headline else shrink(detail)
    s = json.dumps(line)
    if len(s) > LIMIT and not headline:
        line = {k: line[k] for k in REQUIRED + ("cpu_baseline",) if k in line}
        line["detail"] = detail.get("detail_file", "bench_detail.json")
        s = json.dumps(line)
    if len(s) > HARD_LIMIT:
        raise ValueError(f"bench line is {len(s)} bytes (> {HARD_LIMIT}): move the new block into the detail")
    return s


def emit(detail, headline=True, detail_path=None, stdout=None, stderr=None, also_dirs=()):
    """Write the detail to detail_path (and a copy into every directory of also_dirs that exists or can be made), say so in ONE short
    stderr line, then print the final line on stdout.  Returns the final line's dict.

    The detail itself is NOT printed: a reader that keeps the last few KB of stdout followed by stderr (the driver's record does:
    `stdout ... ---- stderr ---- ...`) would find a 30 KB stderr line where the final line should be.  GAUSS_BENCH_DETAIL_STDERR=1
    prints it on stderr, tagged {"detail": ...}, for a terminal user who wants it inline."""
    import os
    import sys
    stdout = stdout or sys.stdout
    stderr = stderr or sys.stderr
    written = []
    if detail_path:
        name = detail_path.rsplit("/", 1)[-1]
        detail = dict(detail, detail_file=name)
        blob = json.dumps(detail)
        paths = [detail_path]
        for d in also_dirs:
            try:
                os.makedirs(d, exist_ok=True)
                paths.append(os.path.join(d, name))
            except OSError:
                pass
        for p in paths:
            try:
                with open(p, "w") as fh:
                    fh.write(blob)
                written.append(p)
            except OSError:
                pass
        if not written:
            detail = {k: v for k, v in detail.items() if k != "detail_file"}
    if os.environ.get("GAUSS_BENCH_DETAIL_STDERR") == "1":
        print(json.dumps({"detail": detail}), file=stderr, flush=True)
    elif written:
        print("bench.py: detail in %s" % ", ".join(written), file=stderr, flush=True)
    s = final_line(detail, headline)
    print(s, file=stdout, flush=True)
    return json.loads(s)
