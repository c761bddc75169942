"""ctypes front-end of oracle/gauss_oracle.c (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    """Compile gauss_oracle.c (and, where /root/reference exists, oracle/_ref)."""
    src = os.path.join(_HERE, "gauss_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _SO


def load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_calcor.restype = C.c_double
        _lib.orc_calwgtcov.restype = C.c_double
        _lib.orc_pnorm_upper.restype = C.c_double
        _lib.orc_pnorm_upper.argtypes = [C.c_double]
        _lib.orc_pchisq_upper.restype = C.c_double
        _lib.orc_pchisq_upper.argtypes = [C.c_double, C.c_int]
    return _lib


def _geno(g):
    """Accept uint8 {0,1,2} or ASCII matrices; return a C-contiguous ASCII (S, N) array."""
    g = np.ascontiguousarray(g)
    if g.dtype != np.uint8:
        g = g.astype(np.uint8)
    if g.size and g.max() < 16:
        g = g + np.uint8(ord("0"))
    return np.ascontiguousarray(g)


def _off(pop_off):
    return np.ascontiguousarray(np.asarray(pop_off, dtype=np.int32))


def _cp(a, t):
    return a.ctypes.data_as(t)


def calcor(x, y, pop_off):
    lib = load()
    g = _geno(np.stack([x, y]))
    po = _off(pop_off)
    return lib.orc_calcor(g[0].ctypes.data_as(C.c_char_p), g[1].ctypes.data_as(C.c_char_p),
                          _cp(po, _ip), C.c_int(len(po) - 1))


def calwgtcov(x, y, pop_off, pop_wgt):
    lib = load()
    g = _geno(np.stack([x, y]))
    po = _off(pop_off)
    w = np.ascontiguousarray(pop_wgt, dtype=np.float64)
    return lib.orc_calwgtcov(g[0].ctypes.data_as(C.c_char_p), g[1].ctypes.data_as(C.c_char_p),
                             _cp(po, _ip), C.c_int(len(po) - 1), _cp(w, _dp))


def compute_ld(geno, pop_off, pop_wgt):
    """computeLD core (computeLD.cpp:95-116): weighted correlation, unit diagonal. Returns (S,S)."""
    lib = load()
    g = _geno(geno)
    S, N = g.shape
    po = _off(pop_off)
    w = np.ascontiguousarray(pop_wgt, dtype=np.float64)
    cor = np.zeros((S, S), dtype=np.float64)
    lib.orc_compute_ld(g.ctypes.data_as(C.c_char_p), C.c_long(N), C.c_int(S), _cp(po, _ip),
                       C.c_int(len(po) - 1), _cp(w, _dp), _cp(cor, _dp))
    return cor  # symmetric, so col-major == row-major


def ld_pooled(geno, pop_off, diag):
    lib = load()
    g = _geno(geno)
    S, N = g.shape
    po = _off(pop_off)
    cor = np.zeros((S, S), dtype=np.float64)
    lib.orc_ld_pooled(g.ctypes.data_as(C.c_char_p), C.c_long(N), C.c_int(S), _cp(po, _ip),
                      C.c_int(len(po) - 1), C.c_double(diag), _cp(cor, _dp))
    return cor


def run_impute(mode, geno_m, geno_u, pop_off, pop_wgt, z1, lam=0.1, min_abs_eig=1e-5,
               want_mats=False):
    """run_dist (mode 0) / run_distmix (mode 1). Returns dict(z, info, mpd[, b11, b21])."""
    lib = load()
    gm, gu = _geno(geno_m), _geno(geno_u)
    M, N = gm.shape
    U = gu.shape[0]
    assert gu.shape[1] == N or U == 0
    po = _off(pop_off)
    P = len(po) - 1
    w = np.ascontiguousarray(pop_wgt if pop_wgt is not None else np.ones(P), dtype=np.float64)
    z1 = np.ascontiguousarray(z1, dtype=np.float64)
    z = np.zeros(U)
    info = np.zeros(U)
    b11 = np.zeros((M, M)) if want_mats else None
    b21 = np.zeros((U, M)) if want_mats else None
    mpd = lib.orc_run_impute(C.c_int(mode), gm.ctypes.data_as(C.c_char_p), C.c_int(M),
                             gu.ctypes.data_as(C.c_char_p), C.c_int(U), C.c_long(N),
                             _cp(po, _ip), C.c_int(P), _cp(w, _dp), _cp(z1, _dp),
                             C.c_double(lam), C.c_double(min_abs_eig), _cp(z, _dp), _cp(info, _dp),
                             _cp(b11, _dp) if want_mats else None,
                             _cp(b21, _dp) if want_mats else None)
    out = dict(z=z, info=info, mpd=mpd)
    if want_mats:
        out["b11"] = b11
        out["b21"] = b21
    return out


def run_qcat(mode, geno_m, geno_u, pop_off, pop_wgt, z1, n_head, n_pred, lam=0.1, eig_cutoff=0.01,
             want_mats=False):
    """run_qcat (mode 0, qcat.cpp:166-245) / run_qcatmix (mode 1). Returns dict(r, num_eig[, b11, b21]);
    r lists the n_pred tested measured SNPs first, then the unmeasured ones."""
    lib = load()
    gm = _geno(geno_m)
    M, N = gm.shape
    gu = _geno(geno_u) if geno_u is not None and len(geno_u) else np.zeros((0, N), dtype=gm.dtype)
    U = gu.shape[0]
    po = _off(pop_off)
    P = len(po) - 1
    w = np.ascontiguousarray(pop_wgt if pop_wgt is not None else np.ones(P), dtype=np.float64)
    z1 = np.ascontiguousarray(z1, dtype=np.float64)
    r = np.zeros(n_pred + U)
    num_eig = C.c_int(0)
    b11 = np.zeros((M, M)) if want_mats else None
    b21 = np.zeros((max(U, 1), M)) if want_mats else None
    lib.orc_run_qcat(C.c_int(mode), gm.ctypes.data_as(C.c_char_p), C.c_int(M),
                     gu.ctypes.data_as(C.c_char_p), C.c_int(U), C.c_long(N),
                     _cp(po, _ip), C.c_int(P), _cp(w, _dp), _cp(z1, _dp),
                     C.c_double(lam), C.c_double(eig_cutoff), C.c_int(n_head), C.c_int(n_pred),
                     _cp(r, _dp), C.byref(num_eig),
                     _cp(b11, _dp) if want_mats else None, _cp(b21, _dp) if want_mats else None)
    out = dict(r=r, num_eig=num_eig.value)
    if want_mats:
        out["b11"], out["b21"] = b11, b21[:U]
    return out


def recode(geno, coding):
    """ConvertGenotypesToDominant (coding 1) / ToRecessive (coding 2), gauss.cpp:1196-1250: only the
    codes 0..2 are mapped, in the input's own alphabet (ASCII digits or small integers)."""
    g = np.array(geno, dtype=np.uint8)
    base = np.where(g >= 48, 48, 0).astype(np.uint8)
    v = g - base
    ok = v <= 2
    new = (v >= 1) if coding == 1 else (v == 2)
    return np.where(ok, new.astype(np.uint8) + base, g).astype(np.uint8) if coding else g


def ld_blocks(mode, geno_m, geno_u, pop_off, pop_wgt, diag=1.0, codings=(0,)):
    """B11 (diagonal `diag`) and the stacked B21 blocks, one per coding (0 additive, 1 dominant,
    2 recessive) of the geno_u rows: prep_qcat.cpp:104-132, prep_qcatmix.cpp:136-221."""
    lib = load()
    gm = _geno(geno_m)
    M, N = gm.shape
    po = _off(pop_off)
    P = len(po) - 1
    w = np.ascontiguousarray(pop_wgt if pop_wgt is not None else np.ones(P), dtype=np.float64)
    b11 = np.zeros((M, M))
    blocks = []
    for k, c in enumerate(codings):
        gu = _geno(recode(geno_u, c)) if len(geno_u) else np.zeros((0, N), dtype=gm.dtype)
        U = gu.shape[0]
        b21 = np.zeros((max(U, 1), M))
        lib.orc_ld_blocks(C.c_int(mode), gm.ctypes.data_as(C.c_char_p), C.c_int(M),
                          gu.ctypes.data_as(C.c_char_p), C.c_int(U), C.c_long(N),
                          _cp(po, _ip), C.c_int(P), _cp(w, _dp), C.c_double(diag),
                          _cp(b11, _dp) if k == 0 else None, _cp(b21, _dp))
        blocks.append(b21[:U])
    return dict(b11=b11, b21=np.vstack(blocks))


def ld_per_pop(geno, pop_off):
    """(P, S(S-1)/2) per-population Pearson r of every pair i < j (zmix.cpp:158-176)."""
    lib = load()
    g = _geno(geno)
    S, N = g.shape
    po = _off(pop_off)
    P = len(po) - 1
    out = np.zeros((P, S * (S - 1) // 2))
    lib.orc_ld_per_pop.restype = None
    lib.orc_ld_per_pop(g.ctypes.data_as(C.c_char_p), C.c_long(N), C.c_int(S), _cp(po, _ip), C.c_int(P), _cp(out, _dp))
    return out


def count_pc(a, eig_cutoff=0.01):
    lib = load()
    a = np.array(a, dtype=np.float64, order="F")
    return lib.orc_count_pc(_cp(a, _dp), C.c_int(a.shape[0]), C.c_double(eig_cutoff))


def make_pos_def(a, min_abs_eig=1e-5):
    lib = load()
    a = np.array(a, dtype=np.float64, order="F")
    n = a.shape[0]
    rc = lib.orc_make_pos_def(_cp(a, _dp), C.c_int(n), C.c_double(min_abs_eig))
    return np.ascontiguousarray(a), rc


def inv_mat(a):
    lib = load()
    a = np.array(a, dtype=np.float64, order="F")
    n = a.shape[0]
    inv = np.zeros((n, n), dtype=np.float64, order="F")
    lib.orc_inv_mat(_cp(inv, _dp), _cp(a, _dp), C.c_int(n))
    return np.ascontiguousarray(inv)


def pnorm_upper(x):
    return load().orc_pnorm_upper(float(x))


def pchisq_upper(x, df):
    return load().orc_pchisq_upper(float(x), int(df))


def jepeg_gene_tail(corg, z, info, has, wgt, min_abs_eig=1e-5, categ_cor_cutoff=0.8,
                    denorm_norm_w=3):
    lib = load()
    corg = np.array(corg, dtype=np.float64, order="F")
    n = corg.shape[0]
    z = np.ascontiguousarray(z, dtype=np.float64)
    info = np.ascontiguousarray(info, dtype=np.float64)
    has = np.ascontiguousarray(has, dtype=np.int32).reshape(n, 6)
    wgt = np.ascontiguousarray(wgt, dtype=np.float64).reshape(n, 6)
    chisq, jp, tcp, tsp = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    df, tc, ts = C.c_int(), C.c_int(), C.c_int()
    lib.orc_jepeg_gene_tail(C.c_int(n), _cp(corg, _dp), _cp(z, _dp), _cp(info, _dp),
                            _cp(has, _ip), _cp(wgt, _dp), C.c_double(min_abs_eig),
                            C.c_double(categ_cor_cutoff), C.c_int(denorm_norm_w),
                            C.byref(chisq), C.byref(df), C.byref(jp), C.byref(tc), C.byref(tcp),
                            C.byref(ts), C.byref(tsp))
    return dict(chisq=chisq.value, df=df.value, jepeg_pval=jp.value, top_categ=tc.value,
                top_categ_pval=tcp.value, top_snp=ts.value, top_snp_pval=tsp.value, num_snp=n)


def gram_counts(geno, c0=0, c1=None):
    lib = load()
    g = _geno(geno)
    S, N = g.shape
    if c1 is None:
        c1 = N
    out = np.zeros((S, S), dtype=np.int64)
    lib.orc_gram_counts(g.ctypes.data_as(C.c_char_p), C.c_long(N), C.c_int(S), C.c_int(c0),
                        C.c_int(c1), out.ctypes.data_as(C.POINTER(C.c_longlong)))
    return out
