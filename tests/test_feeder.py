"""CPU suite: the C++ host data layer (libgauss_host.so) against the Python restatement of the
reference feeder (oracle/feeder_py.py), and the BGZF codec against the reference's own bgzf.c
(oracle/_ref/libref_bgzf.so, compiled from /root/reference/src/bgzf.c where that tree exists)."""
import ctypes as C
import os
import sys
import subprocess

import numpy as np
import pytest

from gauss_amd import api, panel, synth
from oracle import feeder_py as fp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POPS = [("AAA", 60, "EUR"), ("BBB", 45, "EUR"), ("CCC", 70, "ASN"), ("DDD", 33, "AFR"), ("EEE", 52, "EUR")]
WGT = (["aaa", "CCC", "eee", "ZZZ"], [0.5, 0.3, 0.25, 0.4])       # lower case + a population the panel lacks


@pytest.fixture(scope="module")
def study(tmp_path_factory):
    d = tmp_path_factory.mktemp("study")
    return panel.make_synthetic_study(str(d), POPS, n_snp=420, bp_lo=1_000_000, bp_hi=2_600_000, n_genes=25, seed=5)


def _files(st):
    p = st["paths"]
    return p["gwas.txt"], p["index.gz"], p["data.gz"], p["desc.txt"]


def test_bgzf_against_reference_codec(study, tmp_path):
    ref = os.path.join(ROOT, "oracle", "_ref", "libref_bgzf.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    lib = C.CDLL(ref)
    lib.bgzf_open.restype = C.c_void_p
    lib.bgzf_open.argtypes = [C.c_char_p, C.c_char_p]
    lib.bgzf_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.bgzf_seek.restype = C.c_int64
    lib.bgzf_seek.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    lib.bgzf_close.argtypes = [C.c_void_p]

    def ref_read_all(path):
        h = lib.bgzf_open(path.encode(), b"r")
        assert h
        out, buf = b"", C.create_string_buffer(1 << 16)
        while True:
            n = lib.bgzf_read(h, buf, 1 << 16)
            assert n >= 0
            if n == 0:
                break
            out += buf.raw[:n]
        lib.bgzf_close(h)
        return out

    data = study["paths"]["data.gz"]
    want = ref_read_all(data)                                   # the reference codec reads our writer's file
    ours = "\n".join(fp.Bgzf(data).lines()) + "\n"
    assert ours.encode() == want
    # every fpos in the index is a valid virtual offset for the reference's bgzf_seek
    h = lib.bgzf_open(data.encode(), b"r")
    idx = [l.split() for l in fp.Bgzf(study["paths"]["index.gz"]).lines()]
    buf = C.create_string_buffer(64)
    for t in idx[::37]:
        assert lib.bgzf_seek(h, int(t[6]), 0) == 0
        assert lib.bgzf_read(h, buf, 32) == 32
        line = fp.Bgzf(data).line_at(int(t[6]))
        assert buf.raw[:32] == line.encode()[:32]
    lib.bgzf_close(h)
    # C++ reader + writer round trip, then read back by the reference codec
    out = str(tmp_path / "copy.gz")
    n = api.load_host().gauss_host_bgzf_copy(data.encode(), out.encode())
    assert n == len(idx)
    assert ref_read_all(out) == want


@pytest.mark.parametrize("env", [{"GAUSS_NO_LIBDEFLATE": "1"}, {"GAUSS_BGZF_NO_MMAP": "1"}, {"GAUSS_NO_LIBDEFLATE": "1", "GAUSS_BGZF_NO_MMAP": "1"}])
def test_bgzf_reader_fallbacks_read_the_same_bytes(study, tmp_path, env):
    """README: GAUSS_NO_LIBDEFLATE=1 (inflate with zlib when libdeflate cannot be loaded) and GAUSS_BGZF_NO_MMAP=1 (FILE* reads
    when the file cannot be mapped) are the reader's fall-backs.  Each in a process of its own (the inflater is chosen once per
    process): the C++ reader + writer round trip of the panel's data file, and one window of the text feeder, give the bytes and
    the table the default reader gives."""
    data = study["paths"]["data.gz"]
    inp, idx, dat, desc = _files(study)
    code = (
        "import sys, hashlib; sys.path.insert(0, %r)\n"
        "from gauss_amd import api\n"
        "n = api.load_host().gauss_host_bgzf_copy(%r.encode(), sys.argv[1].encode())\n"
        "pr = api.Prepared(api.KIND_DIST, 22, 1_200_001, 1_800_000, 200_000, study_pop='EUR', input_file=%r, reference_index_file=%r,\n"
        "                  reference_data_file=%r, reference_pop_desc_file=%r)\n"
        "h = hashlib.sha256(pr.geno_m().tobytes() + pr.geno_u().tobytes()).hexdigest()\n"
        "print(n, pr.M, pr.U, h)\n" % (ROOT, data, inp, idx, dat, desc))
    outs = []
    for k, e in enumerate(({}, env)):
        out = str(tmp_path / f"copy{k}.gz")
        run = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, GAUSS_AUTO_PACK="0", **e), capture_output=True, text=True, timeout=300)
        assert run.returncode == 0, run.stderr[-2000:]
        outs.append((run.stdout.strip(), "\n".join(fp.Bgzf(out).lines())))
    assert outs[0] == outs[1]
    assert int(outs[0][0].split()[1]) > 10


def _check_prepared(pr, exp_vec, exp_meas, exp_unme, mix):
    df = pr.snps()
    assert list(df["rsid"]) == [s.rsid for s in exp_vec]
    assert list(df["bp"]) == [s.bp for s in exp_vec]
    assert list(df["a1"]) == [s.a1 for s in exp_vec] and list(df["a2"]) == [s.a2 for s in exp_vec]
    assert list(df["type"]) == [s.type for s in exp_vec]
    assert np.array_equal(df["z"].to_numpy(), np.array([s.z for s in exp_vec]))
    af = df["af1mix" if mix else "af1ref"].to_numpy()
    assert np.array_equal(af, np.array([(s.af1mix if mix else s.af1ref) for s in exp_vec]))
    assert pr.M == len(exp_meas) and pr.U == len(exp_unme)
    rs = list(df["rsid"])
    assert [rs[i] for i in pr.measured_rows()] == [s.rsid for s in exp_meas]
    assert [rs[i] for i in pr.unmeasured_rows()] == [s.rsid for s in exp_unme]
    assert np.array_equal(pr.geno_m(), fp._matrix(exp_meas))
    if exp_unme:
        assert np.array_equal(pr.geno_u(), fp._matrix(exp_unme))
    assert np.array_equal(pr.z1(), np.array([s.z for s in exp_meas]))


@pytest.mark.parametrize("mix", [False, True])
def test_prepare_window_matches_python_feeder(study, mix):
    inp, idx, dat, desc = _files(study)
    chr_, lo, hi, wing = 22, 1_400_000, 2_000_000, 250_000
    pops = fp.read_ref_desc(desc)
    if mix:
        flags, w = fp.pop_flags_wgt(pops, *WGT)
    else:
        flags, w = fp.pop_flags(pops, "EUR"), None
    m = fp.read_input_z(inp, chr_, lo - wing, hi + wing, False)
    fp.read_reference_index(m, idx, chr_, lo - wing, hi + wing, False)
    vec = fp.make_snp_vec(m, dat, flags, 0.01, w)
    meas = [s for s in vec if s.type == 1]
    unme = [s for s in vec if s.type == 0 and lo <= s.bp <= hi]
    assert any(s.type == 2 for s in m.values())                  # GWAS-only SNPs exist and drop out (quirk Q6)
    pr = api.Prepared(api.KIND_DISTMIX if mix else api.KIND_DIST, chr=chr_, start_bp=lo, end_bp=hi, wing_size=wing,
                      study_pop=None if mix else "EUR", pop_wgt_df=WGT if mix else None, input_file=inp,
                      reference_index_file=idx, reference_data_file=dat, reference_pop_desc_file=desc)
    _check_prepared(pr, vec, meas, unme, mix)
    assert np.array_equal(pr.pop_off(), fp._selected_off(pops, flags))
    if mix:
        assert np.array_equal(pr.pop_wgt(), np.array(w))          # panel order, unknown "ZZZ" ignored
        assert pr.P == 3
    # allele-swapped GWAS SNPs adopt the panel's alleles and flip z (gauss.cpp:358-370)
    flipped = [s for s in meas if s.rsid in set(study["rsid"])]
    assert flipped
    pr.close()


def test_prepare_qcat_counts_head_and_prediction_snps(study):
    # qcat.cpp:140-152: same partition as dist plus the counts of measured SNPs left of / inside the
    # prediction window; qcat's default af1_cutoff is 0.05 (qcat.cpp:53-57), qcatmix's 0.01
    inp, idx, dat, desc = _files(study)
    chr_, lo, hi, wing = 22, 1_400_000, 2_000_000, 250_000
    pops = fp.read_ref_desc(desc)
    flags = fp.pop_flags(pops, "EUR")
    m = fp.read_input_z(inp, chr_, lo - wing, hi + wing, False)
    fp.read_reference_index(m, idx, chr_, lo - wing, hi + wing, False)
    vec = fp.make_snp_vec(m, dat, flags, 0.05, None)
    meas = [s for s in vec if s.type == 1]
    unme = [s for s in vec if s.type == 0 and lo <= s.bp <= hi]
    pr = api.Prepared(api.KIND_QCAT, chr=chr_, start_bp=lo, end_bp=hi, wing_size=wing, study_pop="EUR", input_file=inp,
                      reference_index_file=idx, reference_data_file=dat, reference_pop_desc_file=desc)
    _check_prepared(pr, vec, meas, unme, False)
    assert pr.n_head == sum(1 for s in meas if s.bp < lo) > 0
    assert pr.n_pred == sum(1 for s in meas if lo <= s.bp <= hi) > 0
    d = pr.window_desc()
    assert d.kind == 1 and d.n_head_measured == pr.n_head and d.n_pred_measured == pr.n_pred and d.eig_cutoff == 0.01
    pr.close()
    pr = api.Prepared(api.KIND_QCATMIX, chr=chr_, start_bp=lo, end_bp=hi, wing_size=wing, pop_wgt_df=WGT, input_file=inp,
                      reference_index_file=idx, reference_data_file=dat, reference_pop_desc_file=desc)
    flags, w = fp.pop_flags_wgt(pops, *WGT)
    vec = fp.make_snp_vec(m2 := _reload(fp, inp, idx, chr_, lo - wing, hi + wing), dat, flags, 0.01, w)
    assert pr.M == sum(1 for s in vec if s.type == 1)
    assert pr.window_desc().mode == 1
    pr.close()


def test_prepare_prep_recessive_flips_to_minor_allele(study):
    # prep_qcatmix.cpp:96-117 + UpdateSnpToMinorAllele (gauss.cpp:1137-1184): SNPs with af1mix > 0.5 get
    # 1 - af, -z, swapped alleles and genotypes 2 - d; the "unmeasured" block is the whole prediction window
    inp, idx, dat, desc = _files(study)
    chr_, lo, hi, wing = 22, 1_400_000, 2_000_000, 250_000
    pops = fp.read_ref_desc(desc)
    flags, w = fp.pop_flags_wgt(pops, *WGT)
    m = _reload(fp, inp, idx, chr_, lo - wing, hi + wing)
    vec = fp.make_snp_vec(m, dat, flags, 0.01, w)
    n_flip = sum(1 for s in vec if s.af1mix > 0.5)
    assert 0 < n_flip < len(vec)
    tr = str.maketrans("012", "210")
    for s in vec:
        if s.af1mix > 0.5:
            s.af1mix, s.z, s.a1, s.a2 = 1 - s.af1mix, -s.z, s.a2, s.a1
            s.geno = [g.translate(tr) for g in s.geno]
    meas = [s for s in vec if s.type == 1]
    pred = [s for s in vec if s.type != 2 and lo <= s.bp <= hi]
    pr = api.Prepared(api.KIND_PREP_RECESSIVE, chr=chr_, start_bp=lo, end_bp=hi, wing_size=wing, pop_wgt_df=WGT,
                      input_file=inp, reference_index_file=idx, reference_data_file=dat, reference_pop_desc_file=desc)
    assert pr.M == len(meas) and pr.U == len(pred) and any(s.type == 1 for s in pred)
    assert np.array_equal(pr.geno_m(), fp._matrix(meas)) and np.array_equal(pr.geno_u(), fp._matrix(pred))
    assert np.array_equal(pr.z1(), np.array([s.z for s in meas]))
    df = pr.snps()
    assert list(df["a1"]) == [s.a1 for s in vec] and np.array_equal(df["af1mix"].to_numpy(), np.array([s.af1mix for s in vec]))
    d = pr.window_desc()
    assert d.kind == 2 and d.u_codings == 7 and d.lambda_ == 0.0
    pr.close()


def _reload(fp, inp, idx, chr_, lo, hi):
    m = fp.read_input_z(inp, chr_, lo, hi, False)
    fp.read_reference_index(m, idx, chr_, lo, hi, False)
    return m


def test_prepare_jepeg_matches_python_feeder(study):
    inp, idx, dat, desc = _files(study)
    ann = study["paths"]["annot.txt"]
    pops = fp.read_ref_desc(desc)
    flags = fp.pop_flags(pops, "EUR")
    m = fp.read_input_z(inp, 0, 0, 0, True)
    fp.read_reference_index(m, idx, 0, 0, 0, True)
    fp.read_annotation(m, ann)
    vec = fp.make_snp_vec(m, dat, flags, 0.01, None)
    gs = sorted([s for s in vec if s.geneid != "." and s.type == 1], key=lambda s: s.geneid)
    pr = api.Prepared(api.KIND_JEPEG, study_pop="EUR", input_file=inp, annotation_file=ann, reference_index_file=idx,
                      reference_data_file=dat, reference_pop_desc_file=desc)
    df = pr.snps()
    assert list(df["rsid"]) == [s.rsid for s in vec]
    assert list(df["geneid"]) == [s.geneid for s in vec]
    rs, gid = list(df["rsid"]), list(df["geneid"])
    rows = pr.measured_rows()
    go = pr.gene_off()
    genes = {}
    for g in range(pr.n_gene):
        ids = {gid[i] for i in rows[go[g]:go[g + 1]]}
        assert len(ids) == 1
        genes[ids.pop()] = sorted(rs[i] for i in rows[go[g]:go[g + 1]])
    want = {}
    for s in gs:
        want.setdefault(s.geneid, []).append(s.rsid)
    assert genes == {k: sorted(v) for k, v in want.items()}
    assert list(np.diff(go) > 0) == [True] * pr.n_gene
    pr.close()


def test_error_texts_follow_the_reference(study):
    inp, idx, dat, desc = _files(study)
    with pytest.raises(api.GaussError, match="invalid population name 'XYZ'"):
        api.Prepared(api.KIND_DIST, chr=22, start_bp=1, end_bp=2, study_pop="XYZ", input_file=inp,
                     reference_index_file=idx, reference_data_file=dat, reference_pop_desc_file=desc)
    with pytest.raises(api.GaussError, match="can't open input file"):
        api.Prepared(api.KIND_DIST, chr=22, start_bp=1, end_bp=2, study_pop="EUR", input_file=inp + ".nope",
                     reference_index_file=idx, reference_data_file=dat, reference_pop_desc_file=desc)
    pr = api.Prepared(api.KIND_DIST, chr=22, start_bp=1_000_000, end_bp=1_010_000, wing_size=0, study_pop="EUR",
                      input_file=inp, reference_index_file=idx, reference_data_file=dat, reference_pop_desc_file=desc)
    with pytest.raises(api.GaussError, match="Not enough number of SNPs loaded - DIST not performed"):
        pr.window_desc()                                          # dist.cpp:145-151
    pr.close()


def test_host_header_symbols_exported():
    import re
    src = open(os.path.join(ROOT, "include", "gauss_host.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    declared = sorted(set(re.findall(r"\b(gauss_[a-z_A-Z0-9]+)\s*\(", src)))
    assert sorted(api.HOST_SYMBOLS) == declared
    h = api.load_host()
    for s in declared:
        assert hasattr(h, s)


# ---- packed panel ------------------------------------------------
@pytest.fixture(scope="module")
def packed(study):
    out = os.path.join(os.path.dirname(study["paths"]["data.gz"]), "panel.gpk")
    _, idx, dat, desc = _files(study)
    n = api.pack_panel(idx, dat, desc, out)
    assert n == len(study["panel_rsid"]) if "panel_rsid" in study else n > 0
    return out


def _same_prepared(a, b, window=None):
    """window = (start_bp, end_bp): `b` is the packed feeder of dist / distmix, which does not enter panel SNPs that no
    study SNP shares a position with and that lie in a wing (type 0 outside the prediction window: the partition drops
    them, dist.cpp:132-140, and the output is cut to the window, dist.cpp:91-93) -- `a`'s SNP list is compared without
    them and its row numbers are renumbered accordingly."""
    da, db = a.snps(), b.snps()
    assert list(da.columns) == list(db.columns)
    rows_a = np.arange(len(da))
    if window is not None:
        keep = (da["type"].to_numpy() != 0) | ((da["bp"].to_numpy() >= window[0]) & (da["bp"].to_numpy() <= window[1]))
        assert keep.sum() < len(da)                    # the wings do hold such SNPs
        rows_a = np.cumsum(keep) - 1                   # old row -> row among the kept ones
        da = da[keep].reset_index(drop=True)
    for c in da.columns:
        if c == "fpos":
            continue                                   # virtual file offset vs row number
        if da[c].dtype.kind == "f":
            assert np.array_equal(da[c].to_numpy(), db[c].to_numpy(), equal_nan=True), c
        else:
            assert list(da[c]) == list(db[c]), c
    assert (a.M, a.U, a.N, a.P, a.n_gene) == (b.M, b.U, b.N, b.P, b.n_gene)
    assert np.array_equal(rows_a[a.measured_rows()], b.measured_rows()) and np.array_equal(rows_a[a.unmeasured_rows()], b.unmeasured_rows())
    assert np.array_equal(a.geno_m(), b.geno_m()) and np.array_equal(a.geno_u(), b.geno_u())
    assert np.array_equal(a.z1(), b.z1()) and np.array_equal(a.pop_off(), b.pop_off())
    assert np.array_equal(a.pop_wgt(), b.pop_wgt()) and np.array_equal(a.gene_off(), b.gene_off())


@pytest.mark.parametrize("kind", ["DIST", "DISTMIX", "QCAT", "QCATMIX", "PREP_QCAT", "PREP_RECESSIVE", "COMPUTELD"])
def test_packed_panel_feeder_equals_text_feeder(study, packed, kind):
    """The packed feeder (binary-searched SNP table, tabulated AF / allele counts, 2-bit rows) must hand the
    numeric step exactly what the BGZF text feeder does."""
    inp, idx, dat, desc = _files(study)
    k = getattr(api, "KIND_" + kind)
    mix = kind in ("DISTMIX", "QCATMIX", "PREP_RECESSIVE", "COMPUTELD")
    kw = dict(chr=22, start_bp=1_400_000, end_bp=2_000_000, wing_size=250_000, study_pop=None if mix else "EUR",
              pop_wgt_df=WGT if mix else None, input_file=inp, reference_index_file=idx, reference_pop_desc_file=desc)
    a = api.Prepared(k, reference_data_file=dat, **kw)
    b = api.Prepared(k, reference_data_file=packed, **dict(kw, reference_index_file="(ignored)"))
    _same_prepared(a, b, window=(kw["start_bp"], kw["end_bp"]) if kind in ("DIST", "DISTMIX") else None)
    if kind in ("DIST", "DISTMIX", "QCAT", "QCATMIX", "PREP_QCAT"):
        d = b.window_desc()
        assert d.geno_format == 1 and bool(d.rows_m) and bool(d.pop_src_off) and d.ld % 16 == 0
        base, nbytes, rb = b.packed_store()
        assert rb == d.ld and nbytes % rb == 0
        # the rows the descriptor names unpack to the same genotypes
        rows = np.ctypeslib.as_array(C.cast(base, C.POINTER(C.c_uint8)), shape=(nbytes // rb, rb))
        rm = np.ctypeslib.as_array(d.rows_m, shape=(b.M,))
        so = np.ctypeslib.as_array(d.pop_src_off, shape=(b.P,))
        sizes = np.diff(b.pop_off())
        assert np.array_equal(panel.unpack2bit(rows[rm], sizes, so) + 48, b.geno_m())
    elif kind == "COMPUTELD":
        # LD-only kinds leave the rows in the panel as well (gauss_ld_rows names them by index)
        assert b.packed_store() is not None
    else:
        assert b.packed_store() is None        # the minor-allele flip needs bytes on the host
    a.close()
    b.close()


def test_packed_panel_jepeg_and_errors(study, packed, tmp_path):
    inp, idx, dat, desc = _files(study)
    ann = study["paths"]["annot.txt"]
    kw = dict(study_pop="ASN", input_file=inp, annotation_file=ann, reference_index_file=idx, reference_pop_desc_file=desc)
    a = api.Prepared(api.KIND_JEPEG, reference_data_file=dat, **kw)
    b = api.Prepared(api.KIND_JEPEG, reference_data_file=packed, **kw)
    _same_prepared(a, b)
    a.close()
    b.close()
    # a description file that disagrees with the packed panel is refused
    bad = tmp_path / "desc_bad.txt"
    lines = open(desc).read().splitlines()
    bad.write_text("\n".join(lines[:-1]) + "\n")
    with pytest.raises(api.GaussError) as ei:
        api.Prepared(api.KIND_DIST, chr=22, start_bp=1_400_000, end_bp=2_000_000, wing_size=250_000, study_pop="EUR",
                     input_file=inp, reference_index_file=idx, reference_data_file=packed, reference_pop_desc_file=str(bad))
    assert "populations" in str(ei.value)


def test_gene_drivers_index_merge_on_odd_positions(tmp_path):
    """jepeg / jepegmix merge the genome-wide index into the study's SNP map (ReadReferenceIndexAll, gauss.cpp:431-518).  On a
    packed panel that merge walks the map's positions (host_feeder.cpp:ReadReferenceIndex) instead of the panel: positions with
    two panel entries, with two study SNPs, with swapped alleles, study-only and panel-only positions, and an annotation that
    names the swapped order must come out exactly as the text feeder's panel-order scan leaves them -- and a site listed
    under both allele orders in the study is the reference's duplicate error in both."""
    pops = [("AAA", 40, "EUR"), ("BBB", 36, "ASN")]
    sizes = [q[1] for q in pops]
    rng = np.random.default_rng(5)
    #        bp       a1   a2
    pan = [(1000, "A", "C"), (2000, "A", "C"), (2000, "A", "G"), (3000, "A", "G"), (4000, "T", "C"), (5000, "G", "A"),
           (6000, "A", "C"), (7000, "C", "T"), (7000, "C", "G"), (8000, "A", "T"), (9000, "G", "C"),
           (9100, "ACGTACGTACGTACGTACGTA", "A"), (9200, "T", "TGGGGGGGGGGGGGGGGGGGGGGGG")]         # indels: alleles longer than a small string
    S = len(pan)
    G = rng.integers(0, 3, size=(S, sum(sizes)), dtype=np.uint8)
    off = np.concatenate([[0], np.cumsum(sizes)])
    af = np.stack([G[:, off[k]:off[k + 1]].sum(1) / (2.0 * sizes[k]) for k in range(len(pops))], 1)
    rsid = np.array([f"rs{i}" for i in range(S)])
    d = str(tmp_path)
    idx, dat, desc, gpk = d + "/index.gz", d + "/data.gz", d + "/desc.txt", d + "/p.gpk"
    panel.write_pop_desc(desc, pops)
    panel.write_panel(idx, dat, rsid, np.full(S, 22), np.array([q[0] for q in pan]), np.array([q[1] for q in pan]),
                      np.array([q[2] for q in pan]), G, af, sizes)
    assert api.pack_panel(idx, dat, desc, gpk) == S
    #          bp     a1   a2    (1000 same; 2000 one study SNP, two panel entries; 3000 two study SNPs, one panel entry; 4000 swapped;
    #                             4500 study only; 6000 different alleles; 7000 two and two; 8000 swapped; 9000 same)
    gw = [(1000, "A", "C"), (2000, "A", "G"), (3000, "A", "G"), (3000, "A", "T"), (4000, "C", "T"), (4500, "A", "C"),
          (6000, "A", "G"), (7000, "C", "T"), (7000, "G", "C"), (8000, "T", "A"), (9000, "G", "C"),
          (9100, "A", "ACGTACGTACGTACGTACGTA"), (9200, "T", "TGGGGGGGGGGGGGGGGGGGGGGGG")]
    z = rng.standard_normal(len(gw))
    gwas = d + "/gwas.txt"
    panel.write_gwas(gwas, [f"g{i}" for i in range(len(gw))], [22] * len(gw), [q[0] for q in gw], [q[1] for q in gw], [q[2] for q in gw], z)
    ann = d + "/annot.txt"
    panel.write_annotation(ann, [("x", 22, 1000, "A", "C", "GENE1", "PROTEIN", 1.5), ("x", 22, 2000, "G", "A", "GENE1", "TFBS", 0.5),
                                 ("x", 22, 4000, "C", "T", "GENE1", "CIS_EQTL", 1.0), ("x", 22, 7000, "C", "T", "GENE2", "PROTEIN", 2.0),
                                 ("x", 22, 7000, "C", "G", "GENE2", "NO_SUCH", 0.7), ("x", 22, 8000, "A", "T", "GENE2", "WTH_HAIR", 0.3),
                                 ("x", 22, 9000, "G", "C", "GENE2", "TRANS_EQTL", 0.9), ("x", 22, 9500, "G", "C", "GENE3", "TFBS", 0.9),
                                 ("x", 22, 9100, "ACGTACGTACGTACGTACGTA", "A", "GENE3", "PROTEIN", 1.1),
                                 ("x", 22, 9200, "TGGGGGGGGGGGGGGGGGGGGGGGG", "T", "GENE3", "CIS_EQTL", 0.4)])
    for kind, kw in ((api.KIND_JEPEG, dict(study_pop="EUR")), (api.KIND_JEPEGMIX, dict(pop_wgt_df=(["AAA", "BBB"], [0.6, 0.4])))):
        base = dict(input_file=gwas, annotation_file=ann, reference_index_file=idx, reference_pop_desc_file=desc, af1_cutoff=0.0001, **kw)
        a = api.Prepared(kind, reference_data_file=dat, **base)
        b = api.Prepared(kind, reference_data_file=gpk, **base)
        _same_prepared(a, b)
        da = b.snps()
        assert len(da) >= 10 and set(da["type"]) == {1}                   # (study-only SNPs have no panel line: the AF filter drops them)
        assert "ACGTACGTACGTACGTACGTA" in set(da["a1"]) and "TGGGGGGGGGGGGGGGGGGGGGGGG" in (set(da["a1"]) | set(da["a2"]))
        assert b.n_gene >= 1
        a.close(); b.close()
        b2 = api.Prepared(kind, reference_data_file=gpk, **base)         # the annotation comes from the cache now
        assert list(b2.snps()["geneid"]) == list(da["geneid"])
        b2.close()
    # the window feeders on the same odd sites (ReadReferenceIndex, gauss.cpp:293-399: panel SNPs the study lacks are entered too)
    for kind, kw in ((api.KIND_DIST, dict(study_pop="EUR")), (api.KIND_DISTMIX, dict(pop_wgt_df=(["AAA", "BBB"], [0.6, 0.4])))):
        base = dict(chr=22, start_bp=500, end_bp=9500, wing_size=0, input_file=gwas, reference_index_file=idx, reference_pop_desc_file=desc,
                    af1_cutoff=0.0001, **kw)
        a = api.Prepared(kind, reference_data_file=dat, **base)
        b = api.Prepared(kind, reference_data_file=gpk, **base)
        _same_prepared(a, b)                                      # (no wing: the packed feeder leaves nothing out)
        assert {0, 1} <= set(b.snps()["type"])
        a.close(); b.close()
    # a study that lists one site under both allele orders: "duplicates" from both feeders
    bad = d + "/gwas_dup.txt"
    panel.write_gwas(bad, ["g0", "g1"], [22, 22], [1000, 1000], ["A", "C"], ["C", "A"], [0.5, -0.5])
    for data in (dat, gpk):
        with pytest.raises(api.GaussError) as ei:
            api.Prepared(api.KIND_JEPEG, study_pop="EUR", input_file=bad, annotation_file=ann, reference_index_file=idx,
                         reference_data_file=data, reference_pop_desc_file=desc)
        assert "duplicates" in str(ei.value)


def test_python_packed_writer_is_byte_identical_to_the_converter(study, packed, tmp_path):
    """panel.write_packed_panel (arrays -> GAUSSPK1) and api.pack_panel (BGZF text -> GAUSSPK1) must agree byte
    for byte: two independent writers of the same format."""
    sizes = [q[1] for q in POPS]
    off = np.concatenate([[0], np.cumsum(sizes)])
    G = study["G"]
    rows, _ = panel.pack2bit(G, off)
    cnt = np.stack([G[:, off[k]:off[k + 1]].sum(1) for k in range(len(POPS))], axis=1)
    out = str(tmp_path / "py.gpk")
    panel.write_packed_panel(out, POPS, study["rsid"], np.full(len(G), 22), study["bp"], study["a1"], study["a2"], rows,
                             study["af"], cnt)
    assert open(out, "rb").read() == open(packed, "rb").read()


def test_fast_text_panel_writer_equals_the_line_by_line_writer(study, packed, tmp_path):
    """panel.write_panel_fast (bench.py's chromosome-sized text panels: array-built lines, members deflated on a thread pool)
    writes the same text as panel.write_panel, and the converter makes the same packed panel from it, byte for byte -- so
    its virtual offsets (other members sizes: another deflate level) name the same lines (gauss.cpp:328-330, 755-763)."""
    import gzip
    sizes = [q[1] for q in POPS]
    idx, dat = str(tmp_path / "fast_index.gz"), str(tmp_path / "fast_data.gz")
    n = panel.write_panel_fast(idx, dat, study["rsid"], np.full(len(study["G"]), 22), study["bp"], study["a1"], study["a2"], study["G"],
                               study["af"], sizes, threads=3)
    text = gzip.open(study["paths"]["data.gz"]).read()
    assert gzip.open(dat).read() == text and n == len(text)
    out = str(tmp_path / "fast.gpk")
    assert api.pack_panel(idx, dat, study["paths"]["desc.txt"], out) == len(study["G"])
    assert open(out, "rb").read() == open(packed, "rb").read()
    # slab by slab (a chromosome's 3.4 GB of text is never held whole): rows through a callable, every slab a fresh member
    idx2, dat2 = str(tmp_path / "slab_index.gz"), str(tmp_path / "slab_data.gz")
    G = study["G"]
    n2 = panel.write_panel_fast(idx2, dat2, study["rsid"], np.full(len(G), 22), study["bp"], study["a1"], study["a2"], lambda a, b: G[a:b],
                                study["af"], sizes, threads=3, slab=97)
    assert gzip.open(dat2).read() == text and n2 == len(text)
    out2 = str(tmp_path / "slab.gpk")
    assert api.pack_panel(idx2, dat2, study["paths"]["desc.txt"], out2) == len(G)
    assert open(out2, "rb").read() == open(packed, "rb").read()


def test_window_snp_maps_live_in_pooled_blocks(study, packed):
    """A window's SNP objects and map nodes are carved out of pooled 2 MB blocks (host_internal.h: BlockPool): many windows
    open at once on several threads, closed in another order, opened again out of the returned blocks -- the SNP lists are
    the same every time, and with no block kept between windows and blocks so small that a window spans several
    (GAUSS_HOST_ARENA_KEEP_MB=0, GAUSS_HOST_ARENA_BLOCK_KB=64: a child process) as well."""
    from concurrent.futures import ThreadPoolExecutor
    inp, idx, dat, desc = _files(study)
    spans = [(1_000_001 + 150_000 * k, 1_000_000 + 150_000 * (k + 1)) for k in range(12)]

    def prep(span):
        return api.Prepared(api.KIND_DISTMIX, chr=22, start_bp=span[0], end_bp=span[1], wing_size=300_000, pop_wgt_df=WGT,
                            input_file=inp, reference_index_file="(packed)", reference_data_file=packed, reference_pop_desc_file=desc)

    def image(pr):
        d = pr.snps()
        return (list(d["rsid"]), list(d["bp"]), d["z"].to_numpy().copy(), pr.measured_rows().copy(), pr.unmeasured_rows().copy())

    first = []
    for rep in range(3):
        with ThreadPoolExecutor(max_workers=4) as pool:
            prs = list(pool.map(prep, spans))
        got = [image(pr) for pr in prs]
        for pr in (prs[::2] + prs[1::2]):                  # not the order they were opened in
            pr.close()
        if rep == 0:
            first = got
            assert sum(len(g[0]) for g in got) > 500
        for a, b in zip(first, got):
            assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from gauss_amd import api\n"
            "n = 0\n"
            "for k in range(3):\n"
            "    pr = api.Prepared(api.KIND_DISTMIX, chr=22, start_bp=1_000_001, end_bp=4_000_000, wing_size=250_000, pop_wgt_df=%r,\n"
            "                      input_file=%r, reference_index_file='(packed)', reference_data_file=%r, reference_pop_desc_file=%r)\n"
            "    n += len(pr.snps()); pr.close()\n"
            "print(n)\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), WGT, inp, packed, desc)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GAUSS_HOST_ARENA_KEEP_MB="0", GAUSS_HOST_ARENA_BLOCK_KB="64"), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    pr = api.Prepared(api.KIND_DISTMIX, chr=22, start_bp=1_000_001, end_bp=4_000_000, wing_size=250_000, pop_wgt_df=WGT, input_file=inp,
                      reference_index_file="(packed)", reference_data_file=packed, reference_pop_desc_file=desc)
    assert int(out.stdout.strip()) == 3 * len(pr.snps()) and len(pr.snps()) * 400 > 2 * 65536       # (several 64 KB blocks' worth of SNP objects)
    pr.close()


# ---- feeder edge cases the reference's drivers hit (no GPU involved) --------------------------------------
def _prep(kind, inp, idx, dat, desc, **kw):
    base = dict(chr=22, start_bp=1_400_000, end_bp=2_000_000, wing_size=250_000, study_pop="EUR", input_file=inp,
                reference_index_file=idx, reference_data_file=dat, reference_pop_desc_file=desc)
    base.update(kw)
    return api.Prepared(kind, **base)


def test_duplicate_gwas_alleles_are_an_error(study, packed, tmp_path):
    """A GWAS file that lists a panel SNP under both allele orders trips the reference's duplicate check
    (gauss.cpp:386-392) -- through the text index and through the packed SNP table alike."""
    inp, idx, dat, desc = _files(study)
    lines = open(inp).read().splitlines()
    # find a GWAS SNP that the panel has, and add its allele-swapped twin
    panel_keys = {(int(b), a, c) for b, a, c in zip(study["bp"], study["a1"], study["a2"])}
    for l in lines[1:]:
        t = l.split()
        if (int(t[2]), t[3], t[4]) in panel_keys or (int(t[2]), t[4], t[3]) in panel_keys:
            twin = " ".join([t[0] + "_dup", t[1], t[2], t[4], t[3], t[5]])
            break
    bad = tmp_path / "dup.txt"
    bad.write_text("\n".join(lines + [twin]) + "\n")
    for data in (dat, packed):
        with pytest.raises(api.GaussError, match="input file contains duplicates"):
            _prep(api.KIND_DIST, str(bad), idx, data, desc, start_bp=1_000_000, end_bp=2_600_000, wing_size=0)


def test_af_cutoffs_are_strict_and_other_chromosomes_are_ignored(study, packed):
    inp, idx, dat, desc = _files(study)
    a = _prep(api.KIND_DIST, inp, idx, dat, desc, af1_cutoff=0.01)
    af = a.snps()["af1ref"].to_numpy()
    assert np.all((af > 0.01) & (af < 0.99))                    # strict on both sides (gauss.cpp:593)
    # a cutoff exactly equal to some SNP's (ceil-rounded) frequency excludes that SNP
    cut = float(np.sort(af)[3])
    b = _prep(api.KIND_DIST, inp, idx, dat, desc, af1_cutoff=cut)
    afb = b.snps()["af1ref"].to_numpy()
    assert cut not in afb and np.all(afb > cut) and len(afb) < len(af)
    a.close()
    b.close()
    # chr filter: nothing of chr 22 is in a chr-21 window; the driver then refuses the window
    for data in (dat, packed):
        c = _prep(api.KIND_DIST, inp, idx, data, desc, chr=21)
        assert c.M == 0 and c.U == 0
        with pytest.raises(api.GaussError, match="Not enough number of SNPs loaded"):
            c.window_desc()
        c.close()


def test_more_than_ten_guard_is_exact(study):
    """dist.cpp:145-151: measured <= 10 or unmeasured <= 10 stops; 11 of each passes."""
    inp, idx, dat, desc = _files(study)
    # grow the window from its left edge until it holds 11 measured and 11 unmeasured SNPs
    lo = 1_000_000
    last = None
    for hi in range(1_020_000, 2_600_000, 10_000):
        p = _prep(api.KIND_DIST, inp, idx, dat, desc, start_bp=lo, end_bp=hi, wing_size=0)
        ok = p.M > 10 and p.U > 10
        if ok:
            p.window_desc()                                      # accepted
            assert last is not None and (last[0] <= 10 or last[1] <= 10)
            p.close()
            break
        with pytest.raises(api.GaussError, match="DIST not performed"):
            p.window_desc()
        last = (p.M, p.U)
        p.close()
    else:
        pytest.fail("window never reached 11 + 11 SNPs")


@pytest.mark.parametrize("seed", [101, 102, 103, 104])
def test_feeder_fuzz_against_python_feeder(tmp_path, seed):
    """Random small studies (population tables, SNP densities, allele swaps, GWAS-only SNPs, window placement,
    cutoffs): the C++ host layer -- text and packed -- must hand over exactly what the Python restatement of the
    reference feeder produces."""
    rng = np.random.default_rng(seed)
    npop = int(rng.integers(2, 7))
    sups = ["EUR", "ASN", "AFR"]
    pops = [(f"P{k:02d}", int(rng.integers(25, 90)), sups[int(rng.integers(0, 3))]) for k in range(npop)]
    pops[0] = (pops[0][0], pops[0][1], "EUR")
    st = panel.make_synthetic_study(str(tmp_path), pops, n_snp=int(rng.integers(120, 260)), bp_lo=500_000, bp_hi=1_500_000,
                                    frac_measured=float(rng.uniform(0.2, 0.6)), frac_swapped=float(rng.uniform(0, 0.4)),
                                    frac_not_in_panel=float(rng.uniform(0, 0.1)), seed=seed)
    p = st["paths"]
    inp, idx, dat, desc = p["gwas.txt"], p["index.gz"], p["data.gz"], p["desc.txt"]
    gpk = str(tmp_path / "f.gpk")
    api.pack_panel(idx, dat, desc, gpk)
    lo = int(rng.integers(500_000, 1_000_000))
    hi = lo + int(rng.integers(150_000, 450_000))
    wing = int(rng.integers(0, 200_000))
    mix = bool(rng.integers(0, 2))
    cutoff = float(rng.choice([0.01, 0.05, 0.1]))
    ref_pops = fp.read_ref_desc(desc)
    if mix:
        names = [q[0].lower() for q in pops if rng.random() < 0.7] or [pops[0][0]]
        wgt = (names, [float(x) for x in rng.uniform(0.05, 0.6, len(names))])
        flags, w = fp.pop_flags_wgt(ref_pops, *wgt)
    else:
        flags, w, wgt = fp.pop_flags(ref_pops, "EUR"), None, None
    m = fp.read_input_z(inp, 22, lo - wing, hi + wing, False)
    fp.read_reference_index(m, idx, 22, lo - wing, hi + wing, False)
    vec = fp.make_snp_vec(m, dat, flags, cutoff, w)
    meas = [s for s in vec if s.type == 1]
    unme = [s for s in vec if s.type == 0 and lo <= s.bp <= hi]
    for data in (dat, gpk):
        pr = api.Prepared(api.KIND_DISTMIX if mix else api.KIND_DIST, chr=22, start_bp=lo, end_bp=hi, wing_size=wing,
                          study_pop=None if mix else "EUR", pop_wgt_df=wgt, input_file=inp, reference_index_file=idx,
                          reference_data_file=data, reference_pop_desc_file=desc, af1_cutoff=cutoff)
        # the packed feeder does not enter wing SNPs that nothing reads (type 0 outside the prediction window)
        seen = vec if data == dat else [s for s in vec if s.type != 0 or lo <= s.bp <= hi]
        _check_prepared(pr, seen, meas, unme, mix)
        assert np.array_equal(pr.pop_off(), fp._selected_off(ref_pops, flags))
        pr.close()


def test_corrupt_packed_panels_are_refused_not_read(study, packed, tmp_path):
    """A packed panel is untrusted input (it is mmap'd and its offsets end up on the GPU): truncated files, counts
    that overflow, string offsets outside the string table and population blocks outside the row must all be
    reported as errors at open time."""
    import struct
    inp, idx, dat, desc = _files(study)
    good = open(packed, "rb").read()
    hdr = struct.unpack("<8sIIQQQQQQQQQI36x", good[:128])
    (magic, ver, n_pop, n_snp, row_bytes, off_pops, off_snps, off_str, off_af, off_cnt, off_geno, total, srt) = hdr

    def repack(**kw):
        f = dict(magic=magic, ver=ver, n_pop=n_pop, n_snp=n_snp, row_bytes=row_bytes, off_pops=off_pops, off_snps=off_snps,
                 off_str=off_str, off_af=off_af, off_cnt=off_cnt, off_geno=off_geno, total=total, srt=srt)
        f.update(kw)
        return struct.pack("<8sIIQQQQQQQQQI36x", f["magic"], f["ver"], f["n_pop"], f["n_snp"], f["row_bytes"], f["off_pops"],
                           f["off_snps"], f["off_str"], f["off_af"], f["off_cnt"], f["off_geno"], f["total"], f["srt"])

    cases = {}
    cases["truncated"] = good[: len(good) // 2]
    cases["n_snp_overflow"] = repack(n_snp=(1 << 62) + 5) + good[128:]
    cases["n_snp_too_many"] = repack(n_snp=n_snp + 1000) + good[128:]
    cases["row_bytes_unaligned"] = repack(row_bytes=row_bytes + 4) + good[128:]
    cases["geno_offset_past_end"] = repack(off_geno=total - 16) + good[128:]
    b = bytearray(good)                                   # first SNP's rsid offset -> far outside the string table
    struct.pack_into("<I", b, off_snps + 4, 0x7FFFFFF0)
    cases["string_offset"] = bytes(b)
    b = bytearray(good)                                   # last population block pushed past the end of a row
    struct.pack_into("<I", b, off_pops + 56 * (n_pop - 1) + 52, (row_bytes + 64) // 16 * 16)
    cases["pop_block"] = bytes(b)
    b = bytearray(good)
    struct.pack_into("<I", b, off_pops + 52, 8)           # population block not 16-byte aligned
    cases["pop_alignment"] = bytes(b)
    for name, blob in cases.items():
        path = tmp_path / f"{name}.gpk"
        path.write_bytes(blob)
        with pytest.raises(api.GaussError):
            api.Prepared(api.KIND_DIST, chr=22, start_bp=1_400_000, end_bp=2_000_000, wing_size=250_000, study_pop="EUR",
                         input_file=inp, reference_index_file=idx, reference_data_file=str(path), reference_pop_desc_file=desc)


def test_panel_cache_packs_once_and_keys_on_file_identity(tmp_path, monkeypatch):
    """Auto-pack on first use (gauss_host_panel_cache): the text panel's packed form is made once, found again, made
    again when a panel file changes, never picked up for a different panel; a one-window entry point uses the cached
    panel when one exists (GAUSS_AUTO_PACK unset) and hands on exactly what the text feeder hands on."""
    import shutil
    import time as _t
    (tmp_path / "s").mkdir()
    st = panel.make_synthetic_study(str(tmp_path / "s"), [("AAA", 60, "EUR"), ("BBB", 45, "EUR"), ("CCC", 50, "ASN")], n_snp=260,
                                    bp_lo=1_000_000, bp_hi=2_000_000, frac_measured=0.3, seed=4)
    p = st["paths"]
    cache = tmp_path / "cache"
    monkeypatch.setenv("GAUSS_PANEL_CACHE", str(cache))
    monkeypatch.delenv("GAUSS_AUTO_PACK", raising=False)
    assert api.panel_cache(p["index.gz"], p["data.gz"], p["desc.txt"], create=False) == (None, 0)
    kw = dict(chr=22, start_bp=1_200_000, end_bp=1_800_000, wing_size=150_000, study_pop="EUR", input_file=p["gwas.txt"],
              reference_index_file=p["index.gz"], reference_data_file=p["data.gz"], reference_pop_desc_file=p["desc.txt"])
    text = api.Prepared(api.KIND_DIST, **kw)                      # no cache entry yet: the text feeder
    assert text.packed_store() is None
    path, n = api.panel_cache(p["index.gz"], p["data.gz"], p["desc.txt"])
    assert n == 260 and os.path.dirname(path) == str(cache) and os.path.basename(path).startswith(os.path.basename(p["data.gz"]) + ".")
    assert api.panel_cache(p["index.gz"], p["data.gz"], p["desc.txt"]) == (path, 0)          # found, not made again
    assert api.panel_cache("(unused)", path, p["desc.txt"]) == (path, 0)                      # a packed panel is its own cache entry
    cached = api.Prepared(api.KIND_DIST, **kw)                    # same arguments: now through the cached packed panel
    assert cached.packed_store() is not None
    _same_prepared(text, cached, window=(kw["start_bp"], kw["end_bp"]))
    monkeypatch.setenv("GAUSS_AUTO_PACK", "0")
    again = api.Prepared(api.KIND_DIST, **kw)                     # the cache is ignored on request
    assert again.packed_store() is None
    monkeypatch.delenv("GAUSS_AUTO_PACK")
    for q in (text, cached, again):
        q.close()
    # another panel (a copy with another mtime is another identity) gets its own entry
    other = tmp_path / "other"
    other.mkdir()
    for f in ("index.gz", "data.gz", "desc.txt"):
        shutil.copy(p[f], other / os.path.basename(p[f]))
    _t.sleep(0.01)
    os.utime(other / os.path.basename(p["data.gz"]), None)
    path2, n2 = api.panel_cache(*(str(other / os.path.basename(p[f])) for f in ("index.gz", "data.gz", "desc.txt")))
    assert n2 == 260 and path2 != path
    assert len([f for f in os.listdir(cache) if f.endswith(".gpk")]) == 2 and not [f for f in os.listdir(cache) if ".tmp." in f or f.endswith(".lock")]


def test_panel_cache_concurrent_callers_pack_once(tmp_path):
    """One rank per GPU asks for the same panel at the same moment: one of them packs (under the lock file), the
    others wait and find it."""
    import subprocess
    import sys
    (tmp_path / "s").mkdir()
    st = panel.make_synthetic_study(str(tmp_path / "s"), [("AAA", 60, "EUR"), ("BBB", 45, "ASN")], n_snp=400, bp_lo=1_000_000,
                                    bp_hi=2_000_000, frac_measured=0.3, seed=5)
    p = st["paths"]
    code = ("import sys; sys.path.insert(0, %r); from gauss_amd import api; "
            "print(*api.panel_cache(%r, %r, %r))" % (ROOT, p["index.gz"], p["data.gz"], p["desc.txt"]))
    env = dict(os.environ, GAUSS_PANEL_CACHE=str(tmp_path / "cache"))
    env.pop("LD_PRELOAD", None)                 # under tools/tsan_host.sh / asan_host.sh the children run the plain library
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True, env=env) for _ in range(4)]
    outs = [q.communicate(timeout=120)[0].split() for q in procs]
    assert all(q.returncode == 0 for q in procs)
    assert len({o[0] for o in outs}) == 1
    assert sorted(int(o[1]) for o in outs) == [0, 0, 0, 400]                  # exactly one of them did the packing


# ---- the chromosome driver's own window (host_chrom.cpp:LeanWindow): a merge of two sorted tables ---------------------
def _view_equals_prepared(kind, mix, kw, gpk):
    """gauss_host_chrom_window_view against gauss_host_prepare on the same packed panel: the SNP list (without the wings' type-0
    SNPs, which the driver's window does not enter and nothing reads), the panel rows of the measured / unmeasured SNPs, z1,
    the QCAT counts and the guard's text -- or the same error from both."""
    args = dict(kw, reference_data_file=gpk, reference_index_file="(packed)")
    vkw = dict(chr=kw["chr"], start_bp=kw["start_bp"], end_bp=kw["end_bp"], wing_size=kw["wing_size"], input_file=kw["input_file"],
               packed_file=gpk, reference_pop_desc_file=kw["reference_pop_desc_file"], study_pop=kw.get("study_pop"),
               pop_wgt_df=kw.get("pop_wgt_df"), af1_cutoff=kw.get("af1_cutoff"))
    try:
        pr = api.Prepared(kind, **args)
    except api.GaussError as e:
        with pytest.raises(api.GaussError) as ei:
            api.chrom_window_view(kind, **vkw)
        assert str(ei.value) == str(e)
        return "error"
    df, named, msg = api.chrom_window_view(kind, **vkw)
    dp = pr.snps()
    lo, hi = kw["start_bp"], kw["end_bp"]
    keep = (dp["type"].to_numpy() != 0) | ((dp["bp"].to_numpy() >= lo) & (dp["bp"].to_numpy() <= hi))
    renum = np.cumsum(keep) - 1
    dpk = dp[keep].reset_index(drop=True)
    assert len(dpk) == len(df)
    for c in df.columns:
        if df[c].dtype.kind == "f":
            assert np.array_equal(df[c].to_numpy(), dpk[c].to_numpy(), equal_nan=True), c
            assert np.array_equal(np.signbit(df[c].to_numpy()), np.signbit(dpk[c].to_numpy())), c       # (-0.0 of a flipped z = 0)
        else:
            assert list(df[c]) == list(dpk[c]), c
    fpos = dp["fpos"].to_numpy()
    assert np.array_equal(named["rows_m"], fpos[pr.measured_rows()]) and np.array_equal(named["rows_u"], fpos[pr.unmeasured_rows()])
    assert np.all(keep[pr.measured_rows()]) and np.all(keep[pr.unmeasured_rows()])
    assert np.array_equal(named["z1"], pr.z1())
    assert list(named["counts"]) == [pr.M, pr.U, pr.n_head, pr.n_pred]
    try:
        pr.window_desc()
        assert msg is None
    except api.GaussError as e:
        assert msg == str(e)
    pr.close()
    return "ok"


@pytest.mark.parametrize("kind", ["DIST", "DISTMIX", "QCAT", "QCATMIX"])
def test_chrom_window_view_equals_prepare(study, packed, kind):
    inp, idx, dat, desc = _files(study)
    mix = kind in ("DISTMIX", "QCATMIX")
    for lo, hi, wing in ((1_400_000, 2_000_000, 250_000), (1_000_000, 2_600_000, 0), (1_000_000, 1_030_000, 5_000), (5_000_000, 6_000_000, 1000)):
        kw = dict(chr=22, start_bp=lo, end_bp=hi, wing_size=wing, study_pop=None if mix else "EUR", pop_wgt_df=WGT if mix else None,
                  input_file=inp, reference_pop_desc_file=desc)
        assert _view_equals_prepared(getattr(api, "KIND_" + kind), mix, kw, packed) == "ok"
    # arguments prepare() refuses are refused with its words
    kw = dict(chr=22, start_bp=1_400_000, end_bp=2_000_000, wing_size=0, study_pop=None if mix else "NOPE",
              pop_wgt_df=None, input_file=inp, reference_pop_desc_file=desc)
    assert _view_equals_prepared(getattr(api, "KIND_" + kind), mix, kw, packed) == "error"


def _odd_study(d, seed, n_sites=None):
    """A small panel and study full of the sites the merge has to hand to the map code: positions the panel lists two or three times
    (other alleles, the same alleles, swapped alleles), equal alleles, study rows listed twice, under other alleles, under both
    orders, study-only positions, rows of other chromosomes on both sides."""
    rng = np.random.default_rng(seed)
    pops = [("AAA", 40, "EUR"), ("BBB", 36, "ASN"), ("CCC", 28, "EUR")]
    sizes = [q[1] for q in pops]
    alle = list("ACGT") + ["AT", "ACGTACGTACGTACGTACGTACG"]
    sites = np.sort(rng.choice(np.arange(1000, 60_000, 100), size=int(n_sites or rng.integers(60, 140)), replace=False))
    pan = []
    for bp in sites:
        k = int(rng.choice([1, 1, 1, 1, 2, 2, 3]))
        first = None
        for j in range(k):
            a1, a2 = rng.choice(alle, 2, replace=bool(rng.random() < (0.08 if n_sites is None else 0.01)))   # (equal alleles + a study row of them: "duplicates")
            if first is not None and rng.random() < (0.3 if n_sites is None else 0.04):   # (a large study: rarely, or every window trips the duplicate check)
                a1, a2 = first if rng.random() < 0.5 else first[::-1]        # the same site again, or under the other order
            first = first or (a1, a2)
            pan.append((22, int(bp), str(a1), str(a2)))
    pan = [(21, 5000, "A", "C"), (21, 7000, "G", "T")] + pan + [(23, 100, "A", "C")]
    S = len(pan)
    G = rng.integers(0, 3, size=(S, sum(sizes)), dtype=np.uint8)
    off = np.concatenate([[0], np.cumsum(sizes)])
    af = np.stack([G[:, off[k]:off[k + 1]].sum(1) / (2.0 * sizes[k]) for k in range(len(pops))], 1)
    idx, dat, desc, gpk, gwas = (os.path.join(d, n) for n in ("index.gz", "data.gz", "desc.txt", "p.gpk", "gwas.txt"))
    panel.write_pop_desc(desc, pops)
    panel.write_panel(idx, dat, np.array([f"rs{i}" for i in range(S)]), np.array([q[0] for q in pan]), np.array([q[1] for q in pan]),
                      np.array([q[2] for q in pan]), np.array([q[3] for q in pan]), G, af, sizes)
    assert api.pack_panel(idx, dat, desc, gpk) == S
    gw = []
    both_orders = bool(rng.random() < (0.25 if n_sites is None else 0.5))      # the reference's "duplicates" error, somewhere in the study
    for c, bp, a1, a2 in pan:
        u = rng.random()
        if u < 0.35:
            continue
        if u < 0.6: gw.append((c, bp, a1, a2))
        elif u < 0.75: gw.append((c, bp, a2, a1))
        elif u < 0.85: gw.append((c, bp, a1, str(rng.choice(alle))))
        elif u < 0.93: gw += [(c, bp, a1, a2), (c, bp, a1, a2)]     # listed twice: the later row counts
        else: gw += [(c, bp, a1, a2), (c, bp, str(rng.choice(alle)), str(rng.choice(alle)))]
    if n_sites is not None:
        # a large study: one window at most may trip the duplicate check (the injected site below) -- a row whose swapped twin is also
        # listed goes, and so does a row of equal alleles that the panel lists too
        have = set(gw)
        pset = set(pan)
        gw = [r for r in gw if not ((r[2] != r[3] and (r[0], r[1], r[3], r[2]) in have) or (r[2] == r[3] and r in pset))]
    if both_orders:
        c, bp, a1, a2 = pan[int(rng.integers(2, S - 1))]
        if a1 != a2:
            gw += [(c, bp, a1, a2), (c, bp, a2, a1)]
    gw += [(22, int(b), "A", "G") for b in rng.choice(np.arange(1050, 60_000, 100), size=6, replace=False)]        # study only
    gw += [(21, 5000, "A", "C"), (20, 30_000, "A", "C")]
    order = rng.permutation(len(gw))
    z = np.round(rng.standard_normal(len(gw)) * 2, 4)
    z[rng.random(len(gw)) < 0.05] = 0.0
    panel.write_gwas(gwas, [f"g{i}" for i in range(len(gw))], [gw[i][0] for i in order], [gw[i][1] for i in order],
                     [gw[i][2] for i in order], [gw[i][3] for i in order], z)
    return dict(idx=idx, dat=dat, desc=desc, gpk=gpk, gwas=gwas, sites=sites)


@pytest.mark.parametrize("seed", range(12))
def test_chrom_window_on_odd_sites_equals_prepare_and_the_text_feeder(tmp_path, seed):
    """Random panels / studies made of the odd sites (above): the driver's window, the packed feeder and the text feeder agree on every
    one of them, for the four window kinds and several window placements -- and fail with the same words where the reference does."""
    st = _odd_study(str(tmp_path), seed)
    rng = np.random.default_rng(1000 + seed)
    outcomes = set()
    for kind in ("DIST", "DISTMIX", "QCAT", "QCATMIX"):
        mix = kind in ("DISTMIX", "QCATMIX")
        for _ in range(3):
            lo = int(rng.integers(500, 30_000))
            hi = lo + int(rng.integers(5_000, 30_000))
            wing = int(rng.choice([0, 3_000, 20_000]))
            kw = dict(chr=22, start_bp=lo, end_bp=hi, wing_size=wing, study_pop=None if mix else "EUR",
                      pop_wgt_df=(["AAA", "BBB", "ccc"], [0.5, 0.3, 0.2]) if mix else None, input_file=st["gwas"],
                      reference_pop_desc_file=st["desc"], af1_cutoff=float(rng.choice([0.0001, 0.05, 0.2])))
            k = getattr(api, "KIND_" + kind)
            outcomes.add(_view_equals_prepared(k, mix, kw, st["gpk"]))
            # the packed feeder against the text feeder (a site the panel lists twice in a WING is a measured SNP in both)
            try:
                a = api.Prepared(k, **dict(kw, reference_index_file=st["idx"], reference_data_file=st["dat"]))
            except api.GaussError as e:
                with pytest.raises(api.GaussError) as ei:
                    api.Prepared(k, **dict(kw, reference_index_file=st["idx"], reference_data_file=st["gpk"]))
                assert str(ei.value) == str(e)
                continue
            b = api.Prepared(k, **dict(kw, reference_index_file=st["idx"], reference_data_file=st["gpk"]))
            da, db = a.snps(), b.snps()
            if kind in ("DIST", "DISTMIX"):
                # (the packed feeder leaves out a wing's type-0 SNP that is its position's only panel entry; the ones it does enter --
                # nothing reads them -- must be the text feeder's, in its order)
                keep = (da["type"].to_numpy() != 0) | ((da["bp"].to_numpy() >= lo) & (da["bp"].to_numpy() <= hi))
                db_keys = set(zip(db["bp"], db["a1"], db["a2"]))
                keep |= np.array([(x, y, w) in db_keys for x, y, w in zip(da["bp"], da["a1"], da["a2"])], dtype=bool)
                da = da[keep].reset_index(drop=True)
            for c in da.columns:
                if c == "fpos":
                    continue
                if da[c].dtype.kind == "f":
                    assert np.array_equal(da[c].to_numpy(), db[c].to_numpy(), equal_nan=True), (kind, c)
                else:
                    assert list(da[c]) == list(db[c]), (kind, c)
            assert (a.M, a.U) == (b.M, b.U) and np.array_equal(a.z1(), b.z1())
            assert np.array_equal(a.geno_m(), b.geno_m()) and np.array_equal(a.geno_u(), b.geno_u())
            a.close(); b.close()
    assert outcomes <= {"ok", "error"}


def test_a_site_the_panel_lists_twice_in_a_wing_is_measured_everywhere(tmp_path):
    """gauss.cpp:356-361: the second of two identical panel entries FINDS the first one in the SNP map and turns it into a type-1
    SNP (z = 0, info = -1) -- also in a wing, where the packed feeders leave a lone unmeasured SNP out.  Text feeder, packed feeder
    and the driver's window agree on it."""
    pops = [("AAA", 40, "EUR"), ("BBB", 36, "ASN")]
    sizes = [q[1] for q in pops]
    rng = np.random.default_rng(3)
    pan = [(1000, "A", "C"), (1000, "A", "C"), (1500, "G", "T")] + [(2000 + 100 * i, "A", "G") for i in range(30)]
    S = len(pan)
    G = rng.integers(0, 3, size=(S, sum(sizes)), dtype=np.uint8)
    off = np.concatenate([[0], np.cumsum(sizes)])
    af = np.stack([G[:, off[k]:off[k + 1]].sum(1) / (2.0 * sizes[k]) for k in range(len(pops))], 1)
    d = str(tmp_path)
    idx, dat, desc, gpk, gwas = d + "/index.gz", d + "/data.gz", d + "/desc.txt", d + "/p.gpk", d + "/gwas.txt"
    panel.write_pop_desc(desc, pops)
    panel.write_panel(idx, dat, np.array([f"rs{i}" for i in range(S)]), np.full(S, 22), np.array([q[0] for q in pan]),
                      np.array([q[1] for q in pan]), np.array([q[2] for q in pan]), G, af, sizes)
    assert api.pack_panel(idx, dat, desc, gpk) == S
    meas = list(range(3, S, 2))
    panel.write_gwas(gwas, [f"g{i}" for i in meas], [22] * len(meas), [pan[i][0] for i in meas], [pan[i][1] for i in meas],
                     [pan[i][2] for i in meas], rng.standard_normal(len(meas)))
    kw = dict(chr=22, start_bp=2000, end_bp=9000, wing_size=1500, study_pop="EUR", input_file=gwas, reference_pop_desc_file=desc, af1_cutoff=0.0001)
    a = api.Prepared(api.KIND_DIST, reference_index_file=idx, reference_data_file=dat, **kw)
    b = api.Prepared(api.KIND_DIST, reference_index_file=idx, reference_data_file=gpk, **kw)
    df, named, msg = api.chrom_window_view(api.KIND_DIST, 22, 2000, 9000, 1500, gwas, gpk, desc, study_pop="EUR", af1_cutoff=0.0001)
    for t in (a.snps(), b.snps(), df):
        row = t[t["bp"] == 1000]
        assert len(row) == 1 and int(row["type"].iloc[0]) == 1 and row["rsid"].iloc[0] == "rs1" and float(row["z"].iloc[0]) == 0.0 and float(row["info"].iloc[0]) == -1.0
    assert 1500 in set(a.snps()["bp"]) and 1500 not in set(b.snps()["bp"]) and 1500 not in set(df["bp"])      # the lone wing SNP
    assert a.M == b.M == int(named["counts"][0]) == len(meas) + 1 and msg is None
    assert np.array_equal(a.z1(), b.z1()) and np.array_equal(a.z1(), named["z1"]) and a.z1()[0] == 0.0
    a.close(); b.close()
