#!/usr/bin/env python3
"""End-to-end probe: files on disk (reference formats, BGZF text panel) -> host data layer (C++) -> HIP ->
tables, with the feeder and the GPU part timed separately (BASELINE.md measurement plan)."""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gauss_amd import api, farm, hotpath, panel, synth  # noqa: E402

n_snp = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
d = tempfile.mkdtemp(prefix="gauss_e2e_")
pops = synth.pop_table()                                   # all 29 populations, N = 32 953
t0 = time.perf_counter()
st = panel.make_synthetic_study(d, pops, n_snp=n_snp, bp_lo=20_000_000, bp_hi=23_000_000, frac_measured=0.13362, seed=3)
t_gen = time.perf_counter() - t0
wgt = (list(synth.PGC2_WEIGHTS.keys()), list(synth.PGC2_WEIGHTS.values()))
p = st["paths"]
files = dict(input_file=p["gwas.txt"], reference_index_file=p["index.gz"], reference_data_file=p["data.gz"],
             reference_pop_desc_file=p["desc.txt"])
ctx = hotpath.Context(0)
out = {"n_snp": n_snp, "panel_bytes": os.path.getsize(p["data.gz"]), "generate_s": t_gen}
for threads in (1, 3):
    tm = {}
    res = farm.impute_chromosome(api.KIND_DISTMIX, 22, 20_000_001, 23_000_000, 500_000, pop_wgt_df=wgt, threads=threads,
                                 timings=tm, compute=lambda pl: farm.gpu_compute(pl, ctx, timings=tm), **files)
    tm["rows"] = len(res["table"])
    tm["imputed_per_s_end_to_end"] = tm["imputed"] / (tm["feeder_s"] + tm["compute_s"])
    out[f"threads_{threads}"] = tm
# the same study through the packed panel: no inflate / text parse, rows resident in HBM
gpk = os.path.join(d, "panel.gpk")
t0 = time.perf_counter()
api.pack_panel(p["index.gz"], p["data.gz"], p["desc.txt"], gpk)
out["pack_panel_s"] = time.perf_counter() - t0
out["packed_bytes"] = os.path.getsize(gpk)
pfiles = dict(files, reference_data_file=gpk)
for threads in (1, 3):
    for resident in (False, True):
        tm = {}
        res2 = farm.impute_chromosome(api.KIND_DISTMIX, 22, 20_000_001, 23_000_000, 500_000, pop_wgt_df=wgt, threads=threads,
                                      timings=tm, compute=lambda pl: farm.gpu_compute(pl, ctx, resident=resident, timings=tm), **pfiles)
        tm["rows"] = len(res2["table"])
        tm["identical_to_text_path"] = bool(np.array_equal(res2["table"]["z"].to_numpy(), res["table"]["z"].to_numpy()))
        tm["imputed_per_s_end_to_end"] = tm["imputed"] / (tm["feeder_s"] + tm["compute_s"])
        out[f"packed_threads_{threads}_{'resident' if resident else 'staged'}"] = tm
print(json.dumps(out))
