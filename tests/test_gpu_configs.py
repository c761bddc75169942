"""BASELINE.json configs[0] / configs[1]: computeLD() on the reference's own GWAS file data/PGC2_3Mb.txt
(721 SNPs, chr10 103.0-106.0 Mb; committed as the input fixture tests/golden/PGC2_3Mb.txt) against a
33KG-shaped panel: all 29 populations with their real sizes (N = 32 953), the PGC2 ancestry weights
(21 populations, N = 32 147).  The real 33KG genotypes are an external download, so the panel is synthetic
(seeded); what is real is the SNP list, the z-scores, the population structure and hence the problem size
of the config.  configs[0] (the reference's CPU path) is played by the CPU oracle, configs[1] is the HIP path;
the bar is the north star's: LD within 1e-5 of the CPU path (asserted far tighter)."""
import os

import numpy as np
import pytest

from gauss_amd import api, panel, synth
from oracle import feeder_py as fp

pytestmark = pytest.mark.gpu
GWAS = os.path.join(os.path.dirname(__file__), "golden", "PGC2_3Mb.txt")
WGT = (list(synth.PGC2_WEIGHTS.keys()), list(synth.PGC2_WEIGHTS.values()))


@pytest.fixture(scope="module")
def cfg(tmp_path_factory):
    d = tmp_path_factory.mktemp("cfg1")
    return panel.make_panel_for_gwas(str(d), synth.pop_table(), GWAS, n_extra=600, seed=20260213)


def test_config1_computeLD_3Mb_window_matches_cpu_path(ctx, cfg):
    p = cfg["paths"]
    args = (10, 104_000_001, 107_000_000, WGT, p["gwas.txt"], p["index.gz"], p["data.gz"], p["desc.txt"])
    got = api.computeLD(*args, ctx=ctx)
    want = fp.computeLD(*args)                               # loop-literal CalWgtCov pair loops (util.cpp:103-124)
    sl, cm = got["snplist"], got["cormat"]
    assert list(sl["rsid"]) == want["rsid"] and list(sl["bp"]) == want["bp"]
    assert 400 < len(sl) <= 529                              # the config's window holds 529 of the file's 721 SNPs
    assert cm.shape == want["cormat"].shape == (len(sl), len(sl))
    assert np.all(np.diag(cm) == 1.0)
    assert np.max(np.abs(cm - want["cormat"])) <= 1e-12     # north star: 1e-5
    assert np.mean(cm == want["cormat"]) > 0.999
    assert cfg["n_swapped"] > 0


def test_config1_through_the_packed_panel(ctx, cfg, tmp_path):
    p = cfg["paths"]
    gpk = str(tmp_path / "cfg.gpk")
    assert api.pack_panel(p["index.gz"], p["data.gz"], p["desc.txt"], gpk) == len(cfg["rsid"])
    a = api.computeLD(10, 104_000_001, 107_000_000, WGT, p["gwas.txt"], p["index.gz"], p["data.gz"], p["desc.txt"], ctx=ctx)
    b = api.computeLD(10, 104_000_001, 107_000_000, WGT, p["gwas.txt"], "(unused)", gpk, p["desc.txt"], ctx=ctx)
    assert list(a["snplist"]["rsid"]) == list(b["snplist"]["rsid"]) and np.array_equal(a["cormat"], b["cormat"])
    # dist() on the European super-population over the middle megabase, text vs packed
    d1 = api.dist(10, 104_000_001, 105_000_000, 500_000, "EUR", p["gwas.txt"], p["index.gz"], p["data.gz"], p["desc.txt"], ctx=ctx)
    d2 = api.dist(10, 104_000_001, 105_000_000, 500_000, "EUR", p["gwas.txt"], "(unused)", gpk, p["desc.txt"], ctx=ctx)
    assert np.array_equal(d1["z"].to_numpy(), d2["z"].to_numpy()) and (d1["type"] == 0).sum() > 10
