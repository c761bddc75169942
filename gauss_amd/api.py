"""The reference's R-level entry points, same names and argument lists (R/RcppExports.R:28-104).

    computeLD(chr, start_bp, end_bp, pop_wgt_df, input_file, reference_index_file,
              reference_data_file, reference_pop_desc_file, af1_cutoff=None)
    dist(chr, start_bp, end_bp, wing_size, study_pop, input_file, ...)
    distmix(chr, start_bp, end_bp, wing_size, pop_wgt_df, input_file, ...)
    jepeg(study_pop, input_file, annotation_file, ...)
    jepegmix(pop_wgt_df, input_file, annotation_file, ...)

Each is a thin ctypes call into libgauss_host.so (C++ host data layer) which delegates the numeric
hot path to libgauss_hip.so (HIP).  A ``pop_wgt_df`` is anything with two columns (population
names, weights): a pandas DataFrame, a dict, or a (names, weights) pair.  Results come back as
pandas DataFrames with the reference's column names; computeLD returns
``{"snplist": DataFrame, "cormat": ndarray}`` like the reference's R list.  Errors the reference
raises with Rcpp::stop surface as ``GaussError`` with the same text.
"""
import ctypes as C
import os

import numpy as np

from . import _lib, hotpath

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libgauss_host.so")

KIND_COMPUTELD, KIND_DIST, KIND_DISTMIX, KIND_JEPEG, KIND_JEPEGMIX, KIND_QCAT, KIND_QCATMIX, \
    KIND_PREP_QCAT, KIND_PREP_RECESSIVE = range(9)

HOST_SYMBOLS = [
    "gauss_host_last_error", "gauss_table_nrow", "gauss_table_ncol", "gauss_table_colname",
    "gauss_table_coltype", "gauss_table_str", "gauss_table_int", "gauss_table_dbl", "gauss_table_matrix",
    "gauss_table_free", "gauss_table_strcol", "gauss_host_computeLD", "gauss_host_dist", "gauss_host_distmix", "gauss_host_jepeg",
    "gauss_host_jepegmix", "gauss_host_qcat", "gauss_host_qcatmix", "gauss_prepared_qcat_counts",
    "gauss_host_prep_qcat", "gauss_host_prep_recessive_impute", "gauss_host_pack_panel", "gauss_prepared_packed_store", "gauss_host_prep_zmix5", "gauss_host_prep_zmix", "gauss_host_prep_zmix2", "gauss_host_prep_zmix3", "gauss_host_prep_zmix4", "gauss_host_prep_zmix5_sup", "gauss_table_n_named", "gauss_table_named_name",
    "gauss_table_named", "gauss_host_prepare", "gauss_prepared_snps", "gauss_prepared_counts",
    "gauss_prepared_measured_rows", "gauss_prepared_unmeasured_rows", "gauss_prepared_geno_m",
    "gauss_prepared_geno_u", "gauss_prepared_pop_off", "gauss_prepared_pop_wgt", "gauss_prepared_z1",
    "gauss_prepared_gene_off", "gauss_prepared_window_desc", "gauss_prepared_finish", "gauss_prepared_free",
    "gauss_host_bgzf_copy", "gauss_host_set_threads",
    "gauss_host_panel_resident", "gauss_host_panel_evict", "gauss_host_impute_chromosome", "gauss_host_impute_genome", "gauss_host_chrom_window_view", "gauss_host_panel_cache", "gauss_table_n_messages",
    "gauss_table_message", "gauss_table_strcol_fixed", "gauss_host_panel_device_rows", "gauss_prepared_store_rows",
    "gauss_host_jepeg_gene_tail", "gauss_host_plan_cost",
    "gauss_host_jepeg_rank", "gauss_host_jepeg_genome", "gauss_prepared_jepeg_plan", "gauss_prepared_jepeg_finish",
]


class ChromStats(C.Structure):
    """gauss_chrom_stats (include/gauss_host.h)"""
    _fields_ = [("n_windows", C.c_int32), ("n_windows_mine", C.c_int32), ("n_skipped", C.c_int32), ("n_failed", C.c_int32),
                ("n_batches", C.c_int32), ("n_merged_giveups", C.c_int32), ("imputed", C.c_int64), ("panel_bytes_uploaded", C.c_int64),
                ("t_total", C.c_double), ("t_plan", C.c_double), ("t_panel_upload", C.c_double), ("t_feeder_wait", C.c_double),
                ("t_job_create", C.c_double), ("t_gpu_wait", C.c_double), ("t_tables", C.c_double), ("gpu_span_ms", C.c_double),
                ("t_tables_tail", C.c_double)]


class GaussError(RuntimeError):
    pass


_host = None
_vp, _cp, _i64, _dbl = C.c_void_p, C.c_char_p, C.c_int64, C.c_double
_strs = C.POINTER(C.c_char_p)
_dp = C.POINTER(C.c_double)


def load_host():
    global _host
    if _host is not None:
        return _host
    _lib.load()       # libgauss_host.so links against libgauss_hip.so: fail loudly if that is missing
    if not os.path.exists(HOST_LIB_PATH):
        raise GaussError(f"{HOST_LIB_PATH} not found: build it with `python -m gauss_amd.build`")
    h = C.CDLL(HOST_LIB_PATH)
    for s in HOST_SYMBOLS:
        if not hasattr(h, s):
            raise GaussError(f"libgauss_host.so does not export {s}")
    h.gauss_host_last_error.restype = _cp
    h.gauss_table_colname.restype = _cp
    h.gauss_table_colname.argtypes = [_vp, C.c_int]
    h.gauss_table_str.restype = _cp
    h.gauss_table_str.argtypes = [_vp, C.c_int, C.c_int]
    h.gauss_table_strcol.restype = C.c_void_p
    h.gauss_table_strcol.argtypes = [_vp, C.c_int, C.POINTER(_i64)]
    h.gauss_table_int.restype = C.POINTER(C.c_int32)
    h.gauss_table_int.argtypes = [_vp, C.c_int]
    h.gauss_table_dbl.restype = _dp
    h.gauss_table_dbl.argtypes = [_vp, C.c_int]
    h.gauss_table_matrix.restype = _dp
    h.gauss_table_matrix.argtypes = [_vp, C.POINTER(C.c_int)]
    for f in ("gauss_table_nrow", "gauss_table_ncol"):
        getattr(h, f).argtypes = [_vp]
    h.gauss_table_coltype.argtypes = [_vp, C.c_int]
    h.gauss_table_free.argtypes = [_vp]
    h.gauss_table_free.restype = None
    files4 = [_cp, _cp, _cp, _cp]
    h.gauss_host_computeLD.argtypes = [_vp, C.c_int, _i64, _i64, _strs, _dp, C.c_int] + files4 + [_dbl, C.POINTER(_vp)]
    h.gauss_host_dist.argtypes = [_vp, C.c_int, _i64, _i64, _i64, _cp] + files4 + [_dbl, C.POINTER(_vp)]
    h.gauss_host_distmix.argtypes = [_vp, C.c_int, _i64, _i64, _i64, _strs, _dp, C.c_int] + files4 + [_dbl, C.POINTER(_vp)]
    h.gauss_host_jepeg.argtypes = [_vp, _cp, _cp] + files4 + [_dbl, C.POINTER(_vp)]
    h.gauss_host_jepegmix.argtypes = [_vp, _strs, _dp, C.c_int, _cp] + files4 + [_dbl, C.POINTER(_vp)]
    h.gauss_host_jepeg_rank.argtypes = [_vp, C.c_int, _cp, _strs, _dp, C.c_int, _cp] + files4 + [_dbl, C.c_int, C.c_int, C.POINTER(_vp)]
    h.gauss_host_jepeg_genome.argtypes = [_vp, C.c_int, C.c_int, _cp, _strs, _dp, C.c_int, _strs, _strs, _strs, _strs, _cp, _dbl, C.c_int, C.c_int,
                                          C.POINTER(_vp), C.POINTER(C.c_int32)]
    h.gauss_prepared_jepeg_plan.argtypes = [_vp, C.c_int, C.POINTER(C.c_int32)]
    h.gauss_prepared_jepeg_finish.argtypes = [_vp, C.c_int, C.c_int, _dp, C.POINTER(_vp)]
    h.gauss_host_prep_zmix5.argtypes = [_vp, _cp, _cp, _cp, _cp, _dbl, C.c_int, C.POINTER(_vp)]
    h.gauss_host_prep_zmix5_sup.argtypes = [_vp, _cp, _cp, _cp, _cp, _dbl, C.c_int, C.POINTER(_vp)]
    h.gauss_host_prep_zmix.argtypes = [_vp, _cp, _cp, _cp, _cp, C.c_int, C.POINTER(_vp)]
    for f in (h.gauss_host_prep_zmix2, h.gauss_host_prep_zmix3, h.gauss_host_prep_zmix4):
        f.argtypes = [_vp, _cp, _cp, _cp, _cp, C.c_int, C.c_int, C.POINTER(_vp)]
    h.gauss_host_pack_panel.restype = _i64
    h.gauss_host_pack_panel.argtypes = [_cp, _cp, _cp, _cp]
    h.gauss_prepared_packed_store.argtypes = [_vp, C.POINTER(C.c_void_p), C.POINTER(_i64), C.POINTER(_i64)]
    h.gauss_host_prep_qcat.argtypes = h.gauss_host_dist.argtypes
    h.gauss_host_prep_recessive_impute.argtypes = h.gauss_host_distmix.argtypes
    h.gauss_table_n_named.argtypes = [_vp]
    h.gauss_table_named_name.restype = _cp
    h.gauss_table_named_name.argtypes = [_vp, C.c_int]
    h.gauss_table_named.restype = _dp
    h.gauss_table_named.argtypes = [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    h.gauss_host_qcat.argtypes = h.gauss_host_dist.argtypes
    h.gauss_host_qcatmix.argtypes = h.gauss_host_distmix.argtypes
    h.gauss_prepared_qcat_counts.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    h.gauss_host_prepare.argtypes = [C.c_int, C.c_int, _i64, _i64, _i64, _cp, _strs, _dp, C.c_int, _cp, _cp, _cp, _cp, _cp,
                                     _dbl, C.POINTER(_vp)]
    h.gauss_prepared_snps.restype = _vp
    h.gauss_prepared_snps.argtypes = [_vp]
    h.gauss_prepared_counts.argtypes = [_vp] + [C.POINTER(C.c_int)] * 5
    for f in ("gauss_prepared_measured_rows", "gauss_prepared_unmeasured_rows", "gauss_prepared_pop_off",
              "gauss_prepared_gene_off"):
        getattr(h, f).restype = C.POINTER(C.c_int32)
        getattr(h, f).argtypes = [_vp]
    for f in ("gauss_prepared_pop_wgt", "gauss_prepared_z1"):
        getattr(h, f).restype = _dp
        getattr(h, f).argtypes = [_vp]
    for f in ("gauss_prepared_geno_m", "gauss_prepared_geno_u"):
        getattr(h, f).restype = C.POINTER(C.c_uint8)
        getattr(h, f).argtypes = [_vp, C.POINTER(_i64)]
    h.gauss_prepared_window_desc.argtypes = [_vp, C.POINTER(_lib.WindowDesc)]
    h.gauss_prepared_finish.argtypes = [_vp, C.POINTER(_vp)]
    h.gauss_prepared_free.argtypes = [_vp]
    h.gauss_prepared_free.restype = None
    h.gauss_host_set_threads.argtypes = [C.c_int]
    h.gauss_host_set_threads.restype = None
    h.gauss_host_bgzf_copy.restype = _i64
    h.gauss_host_bgzf_copy.argtypes = [_cp, _cp]
    h.gauss_host_panel_resident.argtypes = [_vp, _cp, C.POINTER(_i64)]
    h.gauss_host_panel_evict.argtypes = [_vp, _cp]
    h.gauss_host_panel_device_rows.argtypes = [_vp, _cp, C.POINTER(C.c_void_p)]
    ipp = C.POINTER(C.POINTER(C.c_int32))
    h.gauss_prepared_store_rows.argtypes = [_vp, ipp, ipp, ipp, C.POINTER(C.c_int)]
    h.gauss_host_impute_chromosome.argtypes = [_vp, C.c_int, C.c_int, _i64, _i64, _i64, _i64, _cp, _strs, _dp, C.c_int, _cp, _cp, _cp, _cp,
                                               _dbl, C.c_int, C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(ChromStats)]
    h.gauss_host_impute_genome.argtypes = [_vp, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(_i64), C.POINTER(_i64), _i64, _i64, _cp, _strs, _dp,
                                           C.c_int, _cp, _cp, _cp, _cp, _dbl, C.c_int, C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(ChromStats)]
    h.gauss_host_chrom_window_view.argtypes = [C.c_int, C.c_int, _i64, _i64, _i64, _cp, _strs, _dp, C.c_int, _cp, _cp, _cp, _dbl, C.POINTER(_vp)]
    h.gauss_host_panel_cache.argtypes = [_cp, _cp, _cp, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64)]
    h.gauss_table_n_messages.argtypes = [_vp]
    h.gauss_table_message.restype = _cp
    h.gauss_table_message.argtypes = [_vp, C.c_int]
    h.gauss_table_strcol_fixed.restype = C.c_void_p
    h.gauss_table_strcol_fixed.argtypes = [_vp, C.c_int, C.POINTER(C.c_int)]
    _host = h
    return h


def pack_panel(reference_index_file, reference_data_file, reference_pop_desc_file, out_file):
    """BGZF text panel -> packed panel (gauss_host_pack_panel); returns the number of SNPs.  The packed file is
    then passed as reference_data_file to any entry point."""
    n = load_host().gauss_host_pack_panel(_enc(reference_index_file), _enc(reference_data_file),
                                          _enc(reference_pop_desc_file), _enc(out_file))
    if n < 0:
        raise GaussError(load_host().gauss_host_last_error().decode())
    return int(n)


def set_host_threads(n):
    """Threads used inside one prepare call to inflate/split panel lines (gauss_host_set_threads)."""
    load_host().gauss_host_set_threads(int(n))


def _hcheck(rc):
    if rc != 0:
        raise GaussError(load_host().gauss_host_last_error().decode())


def _enc(s):
    return None if s is None else os.fspath(s).encode()


def _pop_wgt(pop_wgt_df):
    """(names, weights) from a DataFrame / dict / pair; first column names, second weights."""
    if hasattr(pop_wgt_df, "iloc"):
        names, w = list(pop_wgt_df.iloc[:, 0]), list(pop_wgt_df.iloc[:, 1])
    elif isinstance(pop_wgt_df, dict):
        names, w = list(pop_wgt_df.keys()), list(pop_wgt_df.values())
    else:
        names, w = list(pop_wgt_df[0]), list(pop_wgt_df[1])
    arr = (C.c_char_p * len(names))(*[str(n).encode() for n in names])
    wv = np.ascontiguousarray(w, dtype=np.float64)
    return arr, wv, len(names)


def _table(h, t, free=True):
    """gauss_table -> pandas DataFrame (or dict of columns when pandas is unavailable)."""
    cols = {}
    n = h.gauss_table_nrow(t)
    for c in range(h.gauss_table_ncol(t)):
        name = h.gauss_table_colname(t, c).decode()
        ty = h.gauss_table_coltype(t, c)
        if ty == 0:
            nb = _i64()
            buf = h.gauss_table_strcol(t, c, C.byref(nb))       # one call per column, not one per cell
            cols[name] = C.string_at(buf, nb.value).decode().split("\0")[:n] if (n and buf) else []
        elif ty == 1:
            cols[name] = np.ctypeslib.as_array(h.gauss_table_int(t, c), shape=(n,)).copy() if n else np.zeros(0, np.int32)
        else:
            cols[name] = np.ctypeslib.as_array(h.gauss_table_dbl(t, c), shape=(n,)).copy() if n else np.zeros(0)
    nn = C.c_int()
    mp = h.gauss_table_matrix(t, C.byref(nn))
    mat = np.ctypeslib.as_array(mp, shape=(nn.value, nn.value)).copy() if mp else None
    if free:
        h.gauss_table_free(t)
    try:
        import pandas as pd
        df = pd.DataFrame(cols)
    except ImportError:       # pragma: no cover
        df = cols
    return df, mat


def _af(af1_cutoff):
    return float("nan") if af1_cutoff is None else float(af1_cutoff)


def _ctx(ctx):
    return (ctx or hotpath.default_context()).handle


def computeLD(chr, start_bp, end_bp, pop_wgt_df, input_file, reference_index_file, reference_data_file,
              reference_pop_desc_file, af1_cutoff=None, ctx=None):
    h = load_host()
    names, w, n = _pop_wgt(pop_wgt_df)
    out = _vp()
    _hcheck(h.gauss_host_computeLD(_ctx(ctx), int(chr), int(start_bp), int(end_bp), names, w.ctypes.data_as(_dp), n,
                                   _enc(input_file), _enc(reference_index_file), _enc(reference_data_file),
                                   _enc(reference_pop_desc_file), _af(af1_cutoff), C.byref(out)))
    df, mat = _table(h, out)
    return {"snplist": df, "cormat": mat}


def dist(chr, start_bp, end_bp, wing_size, study_pop, input_file, reference_index_file, reference_data_file,
         reference_pop_desc_file, af1_cutoff=None, ctx=None):
    h = load_host()
    out = _vp()
    _hcheck(h.gauss_host_dist(_ctx(ctx), int(chr), int(start_bp), int(end_bp), int(wing_size), _enc(study_pop),
                              _enc(input_file), _enc(reference_index_file), _enc(reference_data_file),
                              _enc(reference_pop_desc_file), _af(af1_cutoff), C.byref(out)))
    return _table(h, out)[0]


def distmix(chr, start_bp, end_bp, wing_size, pop_wgt_df, input_file, reference_index_file, reference_data_file,
            reference_pop_desc_file, af1_cutoff=None, ctx=None):
    h = load_host()
    names, w, n = _pop_wgt(pop_wgt_df)
    out = _vp()
    _hcheck(h.gauss_host_distmix(_ctx(ctx), int(chr), int(start_bp), int(end_bp), int(wing_size), names,
                                 w.ctypes.data_as(_dp), n, _enc(input_file), _enc(reference_index_file),
                                 _enc(reference_data_file), _enc(reference_pop_desc_file), _af(af1_cutoff), C.byref(out)))
    return _table(h, out)[0]


def qcat(chr, start_bp, end_bp, wing_size, study_pop, input_file, reference_index_file, reference_data_file,
         reference_pop_desc_file, af1_cutoff=None, ctx=None):
    """qcat() of the reference (qcat.cpp:30-132); af1_cutoff None -> 0.05."""
    h = load_host()
    out = _vp()
    _hcheck(h.gauss_host_qcat(_ctx(ctx), int(chr), int(start_bp), int(end_bp), int(wing_size), _enc(study_pop),
                              _enc(input_file), _enc(reference_index_file), _enc(reference_data_file),
                              _enc(reference_pop_desc_file), _af(af1_cutoff), C.byref(out)))
    return _table(h, out)[0]


def qcatmix(chr, start_bp, end_bp, wing_size, pop_wgt_df, input_file, reference_index_file, reference_data_file,
            reference_pop_desc_file, af1_cutoff=None, ctx=None):
    """qcatmix() of the reference (qcatmix.cpp:30-140); af1_cutoff None -> 0.01."""
    h = load_host()
    names, w, n = _pop_wgt(pop_wgt_df)
    out = _vp()
    _hcheck(h.gauss_host_qcatmix(_ctx(ctx), int(chr), int(start_bp), int(end_bp), int(wing_size), names,
                                 w.ctypes.data_as(_dp), n, _enc(input_file), _enc(reference_index_file),
                                 _enc(reference_data_file), _enc(reference_pop_desc_file), _af(af1_cutoff), C.byref(out)))
    return _table(h, out)[0]


def _named(h, t):
    """Named numeric members of a result List (column-major in C, returned as (nrow, ncol) arrays)."""
    out = {}
    for k in range(h.gauss_table_n_named(t)):
        nr, nc = C.c_int(), C.c_int()
        p = h.gauss_table_named(t, k, C.byref(nr), C.byref(nc))
        name = h.gauss_table_named_name(t, k).decode()
        n = nr.value * nc.value
        a = np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0)
        out[name] = a if nc.value == 1 else a.reshape(nc.value, nr.value).T
    return out


def prep_qcat(chr, start_bp, end_bp, wing_size, study_pop, input_file, reference_index_file, reference_data_file,
              reference_pop_desc_file, af1_cutoff=None, ctx=None):
    """prep_qcat() of the reference (prep_qcat.cpp:16-205): list(snplist, z_vec, cor_mat1, cor_mat2)."""
    h = load_host()
    out = _vp()
    _hcheck(h.gauss_host_prep_qcat(_ctx(ctx), int(chr), int(start_bp), int(end_bp), int(wing_size), _enc(study_pop),
                                   _enc(input_file), _enc(reference_index_file), _enc(reference_data_file),
                                   _enc(reference_pop_desc_file), _af(af1_cutoff), C.byref(out)))
    named = _named(h, out)
    return dict(snplist=_table(h, out)[0], **named)


def prep_recessive_impute(chr, start_bp, end_bp, wing_size, pop_wgt_df, input_file, reference_index_file,
                          reference_data_file, reference_pop_desc_file, af1_cutoff=None, ctx=None):
    """prep_recessive_impute() of the reference (prep_qcatmix.cpp:36-316): list(snplist, zvec, cormat,
    cormat_add, cormat_dom, cormat_rec)."""
    h = load_host()
    names, w, n = _pop_wgt(pop_wgt_df)
    out = _vp()
    _hcheck(h.gauss_host_prep_recessive_impute(_ctx(ctx), int(chr), int(start_bp), int(end_bp), int(wing_size), names,
                                               w.ctypes.data_as(_dp), n, _enc(input_file), _enc(reference_index_file),
                                               _enc(reference_data_file), _enc(reference_pop_desc_file),
                                               _af(af1_cutoff), C.byref(out)))
    named = _named(h, out)
    return dict(snplist=_table(h, out)[0], **named)


def prep_zmix5(input_file, reference_index_file, reference_data_file, reference_pop_desc_file, percentile=None,
               interval=None, ctx=None, with_snps=False):
    """prep_zmix5() of the reference (zmix.cpp:44-190): matrix [n_pairs x (1 + n_pop)], column 0 = z_i*z_j,
    column k+1 = genotype correlation of the pair inside population k."""
    h = load_host()
    out = _vp()
    _hcheck(h.gauss_host_prep_zmix5(_ctx(ctx), _enc(input_file), _enc(reference_index_file), _enc(reference_data_file),
                                    _enc(reference_pop_desc_file), _af(percentile), int(interval or 0), C.byref(out)))
    named = _named(h, out)
    df = _table(h, out)[0]
    return (named["data_mat"], df) if with_snps else named["data_mat"]


def prep_zmix_variant(variant, input_file, reference_index_file, reference_data_file, reference_pop_desc_file, percentile=None,
                      interval=None, p2=None, ctx=None):
    """prep_zmix() / prep_zmix2() / prep_zmix3() / prep_zmix4() / prep_zmix5_sup() of the reference (zmix.cpp:201-1076;
    variant "zmix", "zmix2", "zmix3", "zmix4", "zmix5_sup"; p2 = offset or steps).  Returns dict(data_mat, snps (DataFrame),
    pairs [n_pairs x 2] rows of snps, groups: the names of the correlation columns)."""
    h = load_host()
    out = _vp()
    files = (_enc(input_file), _enc(reference_index_file), _enc(reference_data_file), _enc(reference_pop_desc_file))
    if variant == "zmix":
        _hcheck(h.gauss_host_prep_zmix(_ctx(ctx), *files, int(interval or 0), C.byref(out)))
    elif variant == "zmix5_sup":
        _hcheck(h.gauss_host_prep_zmix5_sup(_ctx(ctx), *files, _af(percentile), int(interval or 0), C.byref(out)))
    else:
        fn = {"zmix2": h.gauss_host_prep_zmix2, "zmix3": h.gauss_host_prep_zmix3, "zmix4": h.gauss_host_prep_zmix4}[variant]
        _hcheck(fn(_ctx(ctx), *files, int(interval or 0), int(p2 or 0), C.byref(out)))
    named = _named(h, out)
    groups = [h.gauss_table_message(out, k).decode() for k in range(h.gauss_table_n_messages(out))]
    df = _table(h, out)[0]
    return dict(data_mat=named["data_mat"], pairs=named["pairs"].astype(np.int64), snps=df, groups=groups)


def jepeg(study_pop, input_file, annotation_file, reference_index_file, reference_data_file, reference_pop_desc_file,
          af1_cutoff=None, ctx=None):
    h = load_host()
    out = _vp()
    _hcheck(h.gauss_host_jepeg(_ctx(ctx), _enc(study_pop), _enc(input_file), _enc(annotation_file),
                               _enc(reference_index_file), _enc(reference_data_file), _enc(reference_pop_desc_file),
                               _af(af1_cutoff), C.byref(out)))
    return _table(h, out)[0]


def jepegmix(pop_wgt_df, input_file, annotation_file, reference_index_file, reference_data_file,
             reference_pop_desc_file, af1_cutoff=None, ctx=None):
    h = load_host()
    names, w, n = _pop_wgt(pop_wgt_df)
    out = _vp()
    _hcheck(h.gauss_host_jepegmix(_ctx(ctx), names, w.ctypes.data_as(_dp), n, _enc(input_file), _enc(annotation_file),
                                  _enc(reference_index_file), _enc(reference_data_file), _enc(reference_pop_desc_file),
                                  _af(af1_cutoff), C.byref(out)))
    return _table(h, out)[0]


def jepeg_rank(kind, input_file, annotation_file, reference_index_file, reference_data_file, reference_pop_desc_file,
               study_pop=None, pop_wgt_df=None, af1_cutoff=None, rank=0, world=1, ctx=None):
    """gauss_host_jepeg_rank: rank `rank` of `world`'s contiguous gene range of a jepeg() (KIND_JEPEG, study_pop) or jepegmix()
    (KIND_JEPEGMIX, pop_wgt_df) call.  Returns (table of the range's genes, (first gene, one past the last, genes in all)); the
    ranks' tables concatenated in rank order are the one-rank table."""
    h = load_host()
    names, w, n = (None, None, 0) if pop_wgt_df is None else _pop_wgt(pop_wgt_df)
    out = _vp()
    _hcheck(h.gauss_host_jepeg_rank(_ctx(ctx), int(kind), _enc(study_pop), names, None if w is None else w.ctypes.data_as(_dp), n,
                                    _enc(input_file), _enc(annotation_file), _enc(reference_index_file), _enc(reference_data_file),
                                    _enc(reference_pop_desc_file), _af(af1_cutoff), int(rank), int(world), C.byref(out)))
    rng = tuple(int(v) for v in _named(h, out)["gene_range"].reshape(-1))
    return _table(h, out)[0], rng


def jepeg_genome(kind, calls, reference_pop_desc_file, study_pop=None, pop_wgt_df=None, af1_cutoff=None, rank=0, world=1, ctx=None,
                 raise_on_error=True):
    """gauss_host_jepeg_genome: `calls` = [(input_file, annotation_file, reference_index_file, reference_data_file)], one jepeg() /
    jepegmix() call each (the reference's user: one per chromosome), dealt whole to the ranks.  Returns (tables, owner): tables[c] is
    call c's gene table on the rank that ran it and None elsewhere; owner[c] that rank."""
    h = load_host()
    names, w, n = (None, None, 0) if pop_wgt_df is None else _pop_wgt(pop_wgt_df)
    nc = len(calls)
    arr = lambda k: (C.c_char_p * nc)(*[_enc(c[k]) for c in calls])
    outs = (_vp * nc)()
    owner = (C.c_int32 * nc)()
    rc = h.gauss_host_jepeg_genome(_ctx(ctx), int(kind), nc, _enc(study_pop), names, None if w is None else w.ctypes.data_as(_dp), n,
                                   arr(0), arr(1), arr(2), arr(3), _enc(reference_pop_desc_file), _af(af1_cutoff), int(rank), int(world),
                                   outs, owner)
    tabs = [(_table(h, _vp(outs[c]))[0] if outs[c] else None) for c in range(nc)]
    if rc != 0 and raise_on_error:
        _hcheck(rc)
    return tabs, [int(o) for o in owner]


class _TableOwner:
    """Keeps a gauss_table alive for the numpy views handed out over its columns; frees it when the last view is gone."""

    def __init__(self, h, t):
        self.h, self.t = h, t

    def __del__(self):
        try:
            if self.t:
                self.h.gauss_table_free(self.t)
                self.t = None
        except Exception:
            pass


def _columns(h, t, owner=None):
    """gauss_table -> {name: numpy array}; string columns come across as ONE fixed-width bytes array each
    (dtype "S<w>"), not as Python strings: a chromosome's table has ~10^5 rows.
    owner = None: the arrays are copies (the caller frees the table).  owner = a _TableOwner of `t`: the arrays are VIEWS of the
    library's columns -- no copy; every array keeps the owner, and with it the table, alive."""
    cols = {}
    n = h.gauss_table_nrow(t)

    def view(ctype, addr, count, dtype):
        buf = (ctype * count).from_address(addr)
        if owner is None:
            return np.frombuffer(buf, dtype=dtype).copy()
        buf._gauss_owner = owner                     # numpy array -> .base (this ctypes view) -> the table's owner
        return np.frombuffer(buf, dtype=dtype)

    for c in range(h.gauss_table_ncol(t)):
        name = h.gauss_table_colname(t, c).decode()
        ty = h.gauss_table_coltype(t, c)
        if ty == 0:
            w = C.c_int()
            buf = h.gauss_table_strcol_fixed(t, c, C.byref(w))
            cols[name] = view(C.c_char, buf, n * w.value, f"S{w.value}") if (n and buf) else np.zeros(0, dtype="S1")
        elif ty == 1:
            cols[name] = view(C.c_int32, C.addressof(h.gauss_table_int(t, c).contents), n, np.int32) if n else np.zeros(0, np.int32)
        else:
            cols[name] = view(C.c_double, C.addressof(h.gauss_table_dbl(t, c).contents), n, np.float64) if n else np.zeros(0)
    return cols


class ChromResult:
    """What gauss_host_impute_chromosome hands back for one rank: `columns` (the reference's output columns as numpy
    arrays plus "window"), `windows` [n_windows x 6: start_bp end_bp owner status measured unmeasured], `stats`
    (dict of gauss_chrom_stats) and `messages` (one text per failed window)."""

    def __init__(self, columns, windows, stats, messages):
        self.columns, self.windows, self.stats, self.messages = columns, windows, stats, messages

    def frame(self):
        """pandas DataFrame with the reference's column names and types (strings decoded)."""
        import pandas as pd
        return pd.DataFrame({k: (v.astype(str) if v.dtype.kind == "S" else v) for k, v in self.columns.items() if k != "window"})

    @staticmethod
    def merge(parts):
        """Tables of several ranks -> one, in window order (prediction windows are disjoint: no de-duplication)."""
        parts = [p for p in parts if p is not None]
        names = list(parts[0].columns)
        cat = {}
        for k in names:
            arrs = [p.columns[k] for p in parts]
            if arrs[0].dtype.kind == "S":
                w = max(a.dtype.itemsize for a in arrs)
                arrs = [a.astype(f"S{w}") for a in arrs]
            cat[k] = np.concatenate(arrs)
        order = np.argsort(cat["window"], kind="stable")
        cat = {k: v[order] for k, v in cat.items()}
        windows = parts[0].windows.copy()
        for p in parts[1:]:
            mine = p.windows[:, 3] >= 0
            windows[mine] = p.windows[mine]
        stats = {k: ([p.stats[k] for p in parts]) for k in parts[0].stats}
        return ChromResult(cat, windows, stats, [m for p in parts for m in p.messages])


def jepeg_gene_tail(corg, z, info, has, wgt):
    """gauss_host_jepeg_gene_tail: the host k x k tail of jepeg()/jepegmix() for one gene (no GPU involved)."""
    h = load_host()
    corg = np.ascontiguousarray(corg, dtype=np.float64)
    n = corg.shape[0]
    z, info = np.ascontiguousarray(z, dtype=np.float64), np.ascontiguousarray(info, dtype=np.float64)
    has = np.ascontiguousarray(has, dtype=np.int32).reshape(n, 6)
    wgt = np.ascontiguousarray(wgt, dtype=np.float64).reshape(n, 6)
    chisq, jp, tcp, tsp = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    df, tc, ts = C.c_int32(), C.c_int32(), C.c_int32()
    ip = C.POINTER(C.c_int32)
    h.gauss_host_jepeg_gene_tail.argtypes = [C.c_int, _dp, _dp, _dp, ip, _dp, _dp, ip, _dp, ip, _dp, ip, _dp]
    _hcheck(h.gauss_host_jepeg_gene_tail(n, corg.ctypes.data_as(_dp), z.ctypes.data_as(_dp), info.ctypes.data_as(_dp),
                                         has.ctypes.data_as(ip), wgt.ctypes.data_as(_dp), C.byref(chisq), C.byref(df), C.byref(jp),
                                         C.byref(tc), C.byref(tcp), C.byref(ts), C.byref(tsp)))
    return dict(chisq=chisq.value, df=df.value, jepeg_pval=jp.value, top_categ=tc.value, top_categ_pval=tcp.value,
                top_snp=ts.value, top_snp_pval=tsp.value, num_snp=n)


def panel_resident(packed_file, ctx=None):
    """Upload a packed panel's genotype rows to the GPU once (gauss_host_panel_resident); returns the bytes moved now."""
    n = _i64()
    _hcheck(load_host().gauss_host_panel_resident(_ctx(ctx), _enc(packed_file), C.byref(n)))
    return n.value


def panel_evict(packed_file=None, ctx=None):
    _hcheck(load_host().gauss_host_panel_evict(_ctx(ctx), _enc(packed_file)))


def panel_cache(reference_index_file, reference_data_file, reference_pop_desc_file, create=True):
    """The packed form of a text panel in the panel cache (gauss_host_panel_cache): (path, SNPs packed by this call),
    or (None, 0) when there is none and create is False."""
    buf = C.create_string_buffer(4096)
    n = C.c_int64()
    rc = load_host().gauss_host_panel_cache(_enc(reference_index_file), _enc(reference_data_file), _enc(reference_pop_desc_file),
                                            1 if create else 0, buf, len(buf), C.byref(n))
    if rc < 0:
        _hcheck(rc)
    return (buf.value.decode(), n.value) if rc == 0 else (None, 0)


def impute_chromosome(kind, chr, start_bp, end_bp, wing_size, input_file, reference_data_file, reference_pop_desc_file,
                      study_pop=None, pop_wgt_df=None, af1_cutoff=None, window_size=1_000_000, rank=0, world=1, n_batches=0,
                      ctx=None, reference_index_file=None):
    """dist / distmix / qcat / qcatmix over every window of [start_bp, end_bp] as ONE native call
    (gauss_host_impute_chromosome): windows sharded over `world` ranks, this rank's windows pipelined through the
    GPU in batches against the resident packed panel.  reference_data_file: a packed panel, or the reference's BGZF
    text panel together with reference_index_file (packed on first use into the panel cache).  Returns a ChromResult."""
    import time
    t0 = time.perf_counter()
    h = load_host()
    names, w, n = (None, None, 0) if pop_wgt_df is None else _pop_wgt(pop_wgt_df)
    out, st = _vp(), ChromStats()
    t1 = time.perf_counter()
    _hcheck(h.gauss_host_impute_chromosome(_ctx(ctx), int(kind), int(chr), int(start_bp), int(end_bp), int(wing_size),
                                           int(window_size), _enc(study_pop), names, None if w is None else w.ctypes.data_as(_dp), n,
                                           _enc(input_file), _enc(reference_index_file), _enc(reference_data_file),
                                           _enc(reference_pop_desc_file),
                                           _af(af1_cutoff), int(rank), int(world), int(n_batches), C.byref(out), C.byref(st)))
    t2 = time.perf_counter()
    owner = _TableOwner(h, out)                                  # frees the table when the last reference to it is gone
    cols = _columns(h, out, owner=owner)                         # views: the table lives as long as any of its columns
    t3 = time.perf_counter()
    windows = _named(h, out)["windows"]
    msgs = [h.gauss_table_message(out, k).decode() for k in range(h.gauss_table_n_messages(out))]
    stats = {k: getattr(st, k) for k, _ in ChromStats._fields_}
    # the wrapper's own time around the native call (ms): arguments in, columns out, the rest (named matrix, messages, free, stats)
    stats["py_ms"] = dict(args=(t1 - t0) * 1e3, native=(t2 - t1) * 1e3, columns=(t3 - t2) * 1e3, rest=(time.perf_counter() - t3) * 1e3)
    res = ChromResult(cols, windows, stats, msgs)
    res._owner = owner                                           # (a table without rows has no view to keep it)
    return res


def chrom_window_view(kind, chr, start_bp, end_bp, wing_size, input_file, packed_file, reference_pop_desc_file, study_pop=None,
                      pop_wgt_df=None, af1_cutoff=None):
    """gauss_host_chrom_window_view (no GPU): one window as the chromosome driver builds it on a sorted packed panel.
    Returns (snps DataFrame, dict of named vectors rows_m / rows_u / z1 / counts, guard message or None)."""
    h = load_host()
    names, w, n = (None, None, 0) if pop_wgt_df is None else _pop_wgt(pop_wgt_df)
    out = _vp()
    _hcheck(h.gauss_host_chrom_window_view(int(kind), int(chr), int(start_bp), int(end_bp), int(wing_size), _enc(study_pop), names,
                                           None if w is None else w.ctypes.data_as(_dp), n, _enc(input_file), _enc(packed_file),
                                           _enc(reference_pop_desc_file), _af(af1_cutoff), C.byref(out)))
    named = {k: v.reshape(-1) for k, v in _named(h, out).items()}
    msgs = [h.gauss_table_message(out, k).decode() for k in range(h.gauss_table_n_messages(out))]
    df = _table(h, out)[0]
    return df, named, (msgs[0] if msgs else None)


def impute_genome(kind, chromosomes, wing_size, input_file, reference_data_file, reference_pop_desc_file, study_pop=None,
                  pop_wgt_df=None, af1_cutoff=None, window_size=1_000_000, rank=0, world=1, depth=2, ctx=None, reference_index_file=None,
                  raise_on_error=True):
    """gauss_host_impute_genome: gauss_host_impute_chromosome for every (chr, start_bp, end_bp) of `chromosomes`, `depth` calls in
    flight on the context at a time (host threads of the library), so that one call's host part runs under the other's GPU work.
    Returns one ChromResult per chromosome, identical to what impute_chromosome returns for it (raise_on_error=False: None in
    the place of a chromosome that failed, and the first failure's message as a second return value)."""
    h = load_host()
    names, w, n = (None, None, 0) if pop_wgt_df is None else _pop_wgt(pop_wgt_df)
    nc = len(chromosomes)
    chrs = np.ascontiguousarray([c[0] for c in chromosomes], dtype=np.int32)
    lo = np.ascontiguousarray([c[1] for c in chromosomes], dtype=np.int64)
    hi = np.ascontiguousarray([c[2] for c in chromosomes], dtype=np.int64)
    outs = (_vp * nc)()
    sts = (ChromStats * nc)()
    rc = h.gauss_host_impute_genome(_ctx(ctx), int(kind), nc, chrs.ctypes.data_as(C.POINTER(C.c_int32)), lo.ctypes.data_as(C.POINTER(_i64)),
                                    hi.ctypes.data_as(C.POINTER(_i64)), int(wing_size), int(window_size), _enc(study_pop), names,
                                    None if w is None else w.ctypes.data_as(_dp), n, _enc(input_file), _enc(reference_index_file),
                                    _enc(reference_data_file), _enc(reference_pop_desc_file), _af(af1_cutoff), int(rank), int(world), int(depth),
                                    outs, sts)
    res = []
    for c in range(nc):
        if not outs[c]:
            res.append(None)
            continue
        out = _vp(outs[c])
        cols = _columns(h, out)
        windows = _named(h, out)["windows"]
        msgs = [h.gauss_table_message(out, k).decode() for k in range(h.gauss_table_n_messages(out))]
        h.gauss_table_free(out)
        res.append(ChromResult(cols, windows, {k: getattr(sts[c], k) for k, _ in ChromStats._fields_}, msgs))
    if not raise_on_error:
        return res, (h.gauss_host_last_error().decode() if rc != 0 else None)
    _hcheck(rc)
    return res


class Prepared:
    """Host data layer output for one window / gene set (no GPU involved): gauss_host_prepare."""

    def __init__(self, kind, chr=0, start_bp=0, end_bp=0, wing_size=0, study_pop=None, pop_wgt_df=None, input_file=None,
                 annotation_file=None, reference_index_file=None, reference_data_file=None, reference_pop_desc_file=None,
                 af1_cutoff=None):
        self.h = load_host()
        names, w, n = (None, None, 0) if pop_wgt_df is None else _pop_wgt(pop_wgt_df)
        self._keep = (names, w)
        out = _vp()
        _hcheck(self.h.gauss_host_prepare(int(kind), int(chr), int(start_bp), int(end_bp), int(wing_size), _enc(study_pop),
                                          names, None if w is None else w.ctypes.data_as(_dp), n, _enc(input_file),
                                          _enc(annotation_file), _enc(reference_index_file), _enc(reference_data_file),
                                          _enc(reference_pop_desc_file), _af(af1_cutoff), C.byref(out)))
        self.handle = out
        c = [C.c_int() for _ in range(5)]
        self.h.gauss_prepared_counts(out, *[C.byref(x) for x in c])
        self.M, self.U, self.N, self.P, self.n_gene = [x.value for x in c]
        a, b = C.c_int(), C.c_int()
        self.h.gauss_prepared_qcat_counts(out, C.byref(a), C.byref(b))
        self.n_head, self.n_pred = a.value, b.value

    def snps(self):
        return _table(self.h, self.h.gauss_prepared_snps(self.handle), free=False)[0]

    def _arr(self, fn, n, dtype):
        p = fn(self.handle)
        return np.ctypeslib.as_array(p, shape=(n,)).astype(dtype).copy() if (n and p) else np.zeros(0, dtype)

    def measured_rows(self):
        return self._arr(self.h.gauss_prepared_measured_rows, self.M, np.int32)

    def unmeasured_rows(self):
        return self._arr(self.h.gauss_prepared_unmeasured_rows, self.U, np.int32)

    def pop_off(self):
        return self._arr(self.h.gauss_prepared_pop_off, self.P + 1, np.int32)

    def pop_wgt(self):
        return self._arr(self.h.gauss_prepared_pop_wgt, self.P, np.float64)

    def z1(self):
        return self._arr(self.h.gauss_prepared_z1, self.M, np.float64)

    def gene_off(self):
        return self._arr(self.h.gauss_prepared_gene_off, self.n_gene + 1 if self.n_gene else 0, np.int32)

    def _geno(self, fn, rows):
        ld = _i64()
        p = fn(self.handle, C.byref(ld))
        if not rows:
            return np.zeros((0, self.N), dtype=np.uint8)
        a = np.ctypeslib.as_array(p, shape=(rows, ld.value))
        return a[:, : self.N]          # a view: keep `self` alive while using it

    def geno_m(self):
        return self._geno(self.h.gauss_prepared_geno_m, self.M)

    def geno_u(self):
        return self._geno(self.h.gauss_prepared_geno_u, self.U)

    def packed_store(self):
        """(host pointer, bytes, row stride) of the packed panel's genotype section, or None for byte matrices."""
        base, nb, rb = C.c_void_p(), _i64(), _i64()
        _hcheck(self.h.gauss_prepared_packed_store(self.handle, C.byref(base), C.byref(nb), C.byref(rb)))
        return (base.value, nb.value, rb.value) if base.value else None

    def window_rows(self, packed_file=None, ctx=None):
        """For a packed-panel object: dict(rows_m, rows_u, pop_src_off, ld[, store]) -- the panel rows of its SNPs and,
        when `packed_file` is resident on `ctx`, the device pointer of the panel's row 0."""
        rm, ru, so = (C.POINTER(C.c_int32)() for _ in range(3))
        n = C.c_int()
        _hcheck(self.h.gauss_prepared_store_rows(self.handle, C.byref(rm), C.byref(ru), C.byref(so), C.byref(n)))
        arr = lambda p, k: (np.ctypeslib.as_array(p, shape=(k,)).astype(np.int32).copy() if k else np.zeros(0, np.int32))
        out = dict(rows_m=arr(rm, self.M), rows_u=arr(ru, self.U), pop_src_off=arr(so, n.value), ld=self.packed_store()[2])
        if packed_file is not None:
            dev = C.c_void_p()
            _hcheck(self.h.gauss_host_panel_device_rows(_ctx(ctx), _enc(packed_file), C.byref(dev)))
            out["store"] = dev.value
        return out

    def window_desc(self):
        d = _lib.WindowDesc()
        _hcheck(self.h.gauss_prepared_window_desc(self.handle, C.byref(d)))
        return d

    def finish(self):
        out = _vp()
        _hcheck(self.h.gauss_prepared_finish(self.handle, C.byref(out)))
        return _table(self.h, out)[0]

    def jepeg_plan(self, world):
        """gauss_prepared_jepeg_plan: first[r] = rank r's first gene (contiguous ranges of equal cost), first[world] = genes in all."""
        first = (C.c_int32 * (int(world) + 1))()
        _hcheck(self.h.gauss_prepared_jepeg_plan(self.handle, int(world), first))
        return [int(v) for v in first]

    def jepeg_finish(self, g0, g1, blocks):
        """gauss_prepared_jepeg_finish: the gene table of genes [g0, g1) from their CorG blocks (a list of n_g x n_g arrays, or
        their concatenation; diagonal 1 + lambda)."""
        flat = np.ascontiguousarray(np.concatenate([np.asarray(b, dtype=np.float64).reshape(-1) for b in blocks]) if len(blocks)
                                    else np.zeros(1), dtype=np.float64)
        out = _vp()
        _hcheck(self.h.gauss_prepared_jepeg_finish(self.handle, int(g0), int(g1), flat.ctypes.data_as(_dp), C.byref(out)))
        return _table(self.h, out)[0]

    def close(self):
        if self.handle:
            self.h.gauss_prepared_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
