"""CPU probe of the host data layer at chromosome scale: the 36 windows of the chr22 study against a 100 000-SNP packed panel with
the 33KG population table shrunk to a few hundred samples (the data layer touches no genotype byte: its cost does not depend
on N).  Prints ms per window of gauss_host_prepare + window descriptor, and with GAUSS_TRACE=prep the phases of each.
    python tools/datalayer_probe.py [repeats]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    from gauss_amd import api, panel, synth, workload
    ch = workload.make_chromosome(100_000, "distmix", seed=20260216, sample_scale=0.01)
    pops_all = [(a, max(8, int(n * 0.01)), s) for a, n, s in synth.pop_table()]
    sizes = [q[1] for q in pops_all]
    S = len(ch["bp"])
    rng = np.random.default_rng(1)
    G = rng.integers(0, 3, size=(S, int(sum(sizes))), dtype=np.uint8)
    off = np.concatenate([[0], np.cumsum(sizes)])
    cnt = np.stack([G[:, off[k]:off[k + 1]].sum(1) for k in range(len(sizes))], 1).astype(np.int32)
    af = cnt / (2.0 * np.array(sizes))[None, :]
    af = np.clip(af, 0.02, 0.98)                      # keep every SNP through the AF filter: the full-size lists
    rows, _ = panel.pack2bit(G, off.astype(np.int32))
    rs, mbp, ma1, ma2, _ = workload.read_study(os.path.join(workload.ROOT, ch["study"]))
    rsid = np.array([f"snp{i}" for i in range(S)], dtype=object)
    alle = np.array(list("ACGT"))
    a1 = alle[rng.integers(0, 4, S)].astype(object)
    a2 = alle[(np.searchsorted(alle, a1.astype(str)) + rng.integers(1, 4, S)) % 4].astype(object)
    m = np.nonzero(ch["measured"])[0]
    rsid[m], a1[m], a2[m] = rs, ma1, ma2
    tmp = tempfile.mkdtemp(prefix="gauss_dl_")
    gpk, desc, gwas = os.path.join(tmp, "p.gpk"), os.path.join(tmp, "desc.txt"), os.path.join(tmp, "gwas.txt")
    panel.write_packed_panel(gpk, pops_all, rsid, np.full(S, 22), ch["bp"], a1, a2, rows, af, cnt)
    panel.write_pop_desc(desc, pops_all)
    panel.write_gwas(gwas, rsid[m], np.full(len(m), 22), ch["bp"][m], a1[m], a2[m], ch["z"][m])
    wgt = (list(synth.PGC2_WEIGHTS.keys()), list(synth.PGC2_WEIGHTS.values()))
    lo = (int(ch["bp"][0]) // 1_000_000) * 1_000_000 + 1
    wins = [(s, s + 999_999) for s in range(lo, int(ch["bp"][-1]) + 1, 1_000_000)]
    for r in range(reps):
        ts = []
        for s, e in wins:
            t0 = time.perf_counter()
            pr = api.Prepared(api.KIND_DISTMIX, 22, s, e, 500_000, pop_wgt_df=wgt, input_file=gwas, reference_index_file="(packed)",
                              reference_data_file=gpk, reference_pop_desc_file=desc)
            d = pr.window_desc() if hasattr(pr, "window_desc") else None
            ts.append((time.perf_counter() - t0) * 1e3)
            mu = (pr.M, pr.U)
            pr.close()
        print("pass %d: %d windows, %.2f ms total, mean %.3f ms, max %.3f ms (last window M, U = %s)" % (r, len(wins), sum(ts), np.mean(ts), max(ts), mu))
    # the chromosome driver's own window (a merge of the two sorted tables), through its test view: setup + build + the view's table
    h = api.load_host()
    import ctypes as C
    names, w, n = api._pop_wgt(wgt)
    for r in range(reps):
        ts = []
        for s, e in wins:
            out = C.c_void_p()
            t0 = time.perf_counter()
            rc = h.gauss_host_chrom_window_view(api.KIND_DISTMIX, 22, s, e, 500_000, None, names, w.ctypes.data_as(C.POINTER(C.c_double)), n,
                                                gwas.encode(), gpk.encode(), desc.encode(), float("nan"), C.byref(out))
            ts.append((time.perf_counter() - t0) * 1e3)
            assert rc == 0
            h.gauss_table_free(out)
        print("driver's window, pass %d: %.2f ms total, mean %.3f ms, max %.3f ms" % (r, sum(ts), np.mean(ts), max(ts)))


if __name__ == "__main__":
    main()
