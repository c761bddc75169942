#!/usr/bin/env python3
"""The reference's own panel format in, one table out, through the chromosome driver:
BGZF text panel (index + data + description, gauss.cpp:293-399, 720-785) + GWAS text file -> gauss_host_impute_chromosome.
First call: the panel is packed into the cache, uploaded, imputed ("cold"); second call, same process: cached and
resident ("warm"); a fresh context: cached but not resident.  Compared with the text feeder window by window (Python
farm + gauss_host_prepare, GAUSS_AUTO_PACK=0 -- the round-1 path) and checked for identical tables."""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gauss_amd import api, farm, hotpath, panel, synth  # noqa: E402

n_snp = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
d = tempfile.mkdtemp(prefix="gauss_text_e2e_")
os.environ["GAUSS_PANEL_CACHE"] = os.path.join(d, "cache")
pops = synth.pop_table()                                   # all 29 populations, N = 32 953
st = panel.make_synthetic_study(d, pops, n_snp=n_snp, bp_lo=20_000_000, bp_hi=23_000_000, frac_measured=0.13362, seed=3)
wgt = (list(synth.PGC2_WEIGHTS.keys()), list(synth.PGC2_WEIGHTS.values()))
p = st["paths"]
span = dict(chr=22, start_bp=20_000_001, end_bp=23_000_000, wing_size=500_000)
kw = dict(kind=api.KIND_DISTMIX, pop_wgt_df=wgt, window_size=1_000_000, input_file=p["gwas.txt"], reference_index_file=p["index.gz"],
          reference_data_file=p["data.gz"], reference_pop_desc_file=p["desc.txt"], **span)
out = {"n_snp": n_snp, "samples": int(sum(q[1] for q in pops)), "text_panel_bytes": os.path.getsize(p["data.gz"]) + os.path.getsize(p["index.gz"])}
ctx = hotpath.Context(0)
hotpath.impute_window  # noqa: B018  (library loaded)


def timed(**extra):
    t0 = time.perf_counter()
    r = api.impute_chromosome(ctx=extra.pop("ctx", ctx), **dict(kw, **extra))
    return time.perf_counter() - t0, r


t_cold, r_cold = timed()
t_warm, r_warm = min((timed() for _ in range(3)), key=lambda q: q[0])
ctx2 = hotpath.Context(0)
t_cached, r_cached = timed(ctx=ctx2)                       # a new session: packed panel in the cache, nothing resident
ctx2.close()
imputed = int(r_cold.stats["imputed"])
out["chromosome_driver"] = {
    "cold_pack_upload_impute_s": t_cold, "warm_s": t_warm, "cached_not_resident_s": t_cached, "imputed_snps": imputed,
    "imputed_per_s_cold": imputed / t_cold, "imputed_per_s_warm": imputed / t_warm, "imputed_per_s_cached": imputed / t_cached,
    "packed_bytes": sum(os.path.getsize(os.path.join(os.environ["GAUSS_PANEL_CACHE"], f)) for f in os.listdir(os.environ["GAUSS_PANEL_CACHE"])),
}
# the text feeder, window by window (what round 1 measured at 59 k SNPs/s)
os.environ["GAUSS_AUTO_PACK"] = "0"
tm = {}
t0 = time.perf_counter()
res = farm.impute_chromosome(api.KIND_DISTMIX, pop_wgt_df=wgt, window_size=1_000_000, threads=3, timings=tm, input_file=p["gwas.txt"],
                             reference_index_file=p["index.gz"], reference_data_file=p["data.gz"], reference_pop_desc_file=p["desc.txt"],
                             compute=lambda pl: farm.gpu_compute(pl, ctx, timings=tm), **span)
t_text = time.perf_counter() - t0
del os.environ["GAUSS_AUTO_PACK"]
f = r_warm.frame()
same = bool(np.array_equal(f["z"].to_numpy(), res["table"]["z"].to_numpy()) and list(f["rsid"]) == list(res["table"]["rsid"]) and
            np.array_equal(r_cold.columns["z"], r_warm.columns["z"]) and np.array_equal(r_cached.columns["z"], r_warm.columns["z"]))
out["text_feeder_window_by_window"] = {"s": t_text, "imputed_per_s": imputed / t_text, "feeder_s": tm.get("feeder_s"), "compute_s": tm.get("compute_s")}
out["tables_identical"] = same
print(json.dumps(out))
shutil.rmtree(d, ignore_errors=True)
