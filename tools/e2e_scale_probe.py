#!/usr/bin/env python3
"""End to end at the BASELINE scale: a chr22-sized packed panel FILE (100 000 SNPs x 32 953 samples, 29 populations)
and a GWAS text file on disk -> distmix over all 1 Mb windows -> result table, timed cold (panel slice uploaded)
and warm (a second study against the panel that is already mapped).  Genotypes are synthesised on the GPU
(bench.py's generator), packed there and written with panel.write_packed_panel."""
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gauss_amd import _lib, api, farm, hotpath, panel, synth  # noqa: E402
import bench  # noqa: E402


class A:
    snps = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    sample_scale = 1.0
    wing = 500_000
    windows = 0


def main():
    torch.cuda.set_device(0)
    ctx = hotpath.Context(0)
    lib = ctx.lib
    ch = bench.make_chromosome(A, seed=20260213)
    # bench's chromosome has the 21 PGC2 populations; the panel FILE carries all 29, the weights select 21
    pops_all = synth.pop_table()
    sel = [k for k, q in enumerate(pops_all) if q[0] in synth.PGC2_WEIGHTS]
    rng = np.random.default_rng(5)
    thr_all = np.zeros((A.snps, len(pops_all)), dtype=np.float32)
    thr_all[:, sel] = ch["thr"]
    rest = [k for k in range(len(pops_all)) if k not in sel]
    thr_all[:, rest] = ch["thr"][:, rng.integers(0, len(sel), len(rest))]
    off_all = synth.pop_offsets([q[1] for q in pops_all])
    N = int(off_all[-1])
    ld = (N + 63) // 64 * 64
    t0 = time.perf_counter()
    g = torch.empty((A.snps, ld), dtype=torch.uint8, device="cuda")
    _lib.check(lib.gauss_synth_device(ctx.handle, g.data_ptr(), A.snps, ld, off_all.ctypes.data_as(C.POINTER(C.c_int32)),
                                      len(pops_all), np.ascontiguousarray(thr_all).ctypes.data_as(C.POINTER(C.c_float)),
                                      ch["rho"].ctypes.data_as(C.POINTER(C.c_float)), C.c_uint64(20260213)))
    sizes = np.diff(off_all)
    ld2 = int(sum((int(m) + 63) // 64 * 16 for m in sizes))
    store = torch.empty((A.snps, ld2), dtype=torch.uint8, device="cuda")
    _lib.check(lib.gauss_pack2bit_device(ctx.handle, g.data_ptr(), ld, store.data_ptr(), ld2, A.snps,
                                         off_all.ctypes.data_as(C.POINTER(C.c_int32)), len(pops_all)))
    cnt = torch.stack([g[:, off_all[k]:off_all[k + 1]].sum(1, dtype=torch.int32) for k in range(len(pops_all))], 1).cpu().numpy()
    af = cnt / (2.0 * sizes)[None, :]
    rows = store.cpu().numpy()
    del g, store
    torch.cuda.empty_cache()
    d = tempfile.mkdtemp(prefix="gauss_scale_")
    rsid = np.array([f"rs{i}" for i in range(A.snps)])
    alle = np.array(list("ACGT"))
    a1 = alle[rng.integers(0, 4, A.snps)]
    a2 = alle[(np.searchsorted(alle, a1) + rng.integers(1, 4, A.snps)) % 4]
    gpk = os.path.join(d, "chr22.gpk")
    nbytes = panel.write_packed_panel(gpk, pops_all, rsid, np.full(A.snps, 22), ch["bp"], a1, a2, rows, af, cnt)
    desc = os.path.join(d, "desc.txt")
    panel.write_pop_desc(desc, pops_all)
    m = np.nonzero(ch["measured"])[0]
    gwas = os.path.join(d, "gwas.txt")
    panel.write_gwas(gwas, rsid[m], np.full(len(m), 22), ch["bp"][m], a1[m], a2[m], ch["z"][m])
    t_make = time.perf_counter() - t0
    wgt = (list(synth.PGC2_WEIGHTS.keys()), list(synth.PGC2_WEIGHTS.values()))
    files = dict(input_file=gwas, reference_index_file="(unused)", reference_data_file=gpk, reference_pop_desc_file=desc)
    lo = (int(ch["bp"][0]) // 1_000_000) * 1_000_000 + 1
    hi = int(ch["bp"][-1])
    out = {"snps": A.snps, "samples_in_file": N, "packed_file_bytes": nbytes, "make_s": t_make}
    for label in ("cold", "warm"):
        tm = {}
        t0 = time.perf_counter()
        res = farm.impute_chromosome(api.KIND_DISTMIX, 22, lo, hi, 500_000, pop_wgt_df=wgt, timings=tm,
                                     compute=lambda pl: farm.gpu_compute(pl, ctx, timings=tm), **files)
        tm["total_s"] = time.perf_counter() - t0
        tm["rows"] = len(res["table"])
        tm["imputed_per_s_end_to_end"] = tm["imputed"] / tm["total_s"]
        tm["all_finite"] = bool(np.all(np.isfinite(res["table"]["z"].to_numpy())))
        out[label] = tm
    print(json.dumps(out))


if __name__ == "__main__":
    main()
