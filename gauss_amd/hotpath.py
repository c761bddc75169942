"""numpy-level front end of the C ABI (include/gauss_hip.h) -- the numeric hot path on the GPU.

Each function is a thin marshalling layer over one C entry point; all arithmetic happens in
libgauss_hip.so on the device.  Names follow the reference functions they replace.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MODE_POOLED, MODE_WEIGHTED, WindowDesc, check  # noqa: F401

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class Context:
    """One HIP device context (gauss_hip_init / gauss_hip_destroy)."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.gauss_hip_init(int(device), C.byref(h)))
        self.handle = h
        self.device = device

    def set_gram_dtype(self, dtype):
        """'f32' (default, v_mfma_f32_32x32x2_f32) or 'i8' (v_mfma_i32_32x32x32_i8); same exact integers."""
        code = {"f32": _lib.GRAM_F32, "i8": _lib.GRAM_I8}[dtype] if isinstance(dtype, str) else int(dtype)
        check(self.lib.gauss_hip_set_gram_dtype(self.handle, code))

    @property
    def id(self):
        """Process-unique id of the context (gauss_hip_context_id); 0 once closed."""
        return int(self.lib.gauss_hip_context_id(self.handle)) if self.handle else 0

    def trim_cache(self):
        """Give the context's cached job workspaces back to the device; returns the bytes freed."""
        n = C.c_int64()
        check(self.lib.gauss_hip_trim_cache(self.handle, C.byref(n)))
        return n.value

    def counters(self):
        """gauss_hip_counters: runs queued merged / demoted to two launches, merged runs that gave up (re-run inside the
        fetch), re-runs that failed too."""
        out = (C.c_int64 * 4)()
        check(self.lib.gauss_hip_counters(self.handle, out))
        return dict(merged=int(out[0]), demoted=int(out[1]), giveups=int(out[2]), rerun_failed=int(out[3]))

    def queues(self):
        """gauss_hip_queues: the library's priority streams on this device, the runtime's hardware queues per class, and whether
        this context's chain / low-priority / main queues were SEEN to run side by side when it was made."""
        out = (C.c_int32 * 4)()
        check(self.lib.gauss_hip_queues(self.handle, out))
        return dict(hi_streams=int(out[0]), lo_streams=int(out[1]), hw_queues_per_class=int(out[2]), probed_distinct=bool(out[3]))

    def close(self):
        if self.handle:
            # jobs and row stores that are still alive are released here (gauss_hip.h, lifetime rule)
            self.lib.gauss_hip_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def device_count():
    """HIP devices visible to this process (gauss_hip_device_count; does not create a context)."""
    n = C.c_int()
    check(_lib.load().gauss_hip_device_count(C.byref(n)))
    return n.value


def rank_device(local_rank=None, n_devices=None):
    """The device a rank of a one-process-per-GPU job owns: LOCAL_RANK modulo the visible devices.
    GAUSS_SHARED_DEVICE=1 (a rehearsal of several ranks on one card) maps every rank to device 0."""
    import os
    if os.environ.get("GAUSS_SHARED_DEVICE") == "1" or os.environ.get("GAUSS_BENCH_SHARED_DEVICE") == "1":
        return 0
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if n_devices is None:
        n_devices = device_count()
    if n_devices < 1:
        raise _lib.GaussHipError("no HIP device is visible: the hot path has no CPU fallback")
    return int(local_rank) % int(n_devices)


def default_context():
    """One context per process, on the device this rank owns (LOCAL_RANK under torchrun, else 0)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(rank_device())
    return _default_ctx


def _pops(pop_off, pop_wgt):
    po = np.ascontiguousarray(pop_off, dtype=np.int32)
    w = None if pop_wgt is None else np.ascontiguousarray(pop_wgt, dtype=np.float64)
    return po, w


def gram_counts(geno, ctx=None):
    """Exact integer sum_n x_i[n] x_j[n] (util.cpp:62,114 `sumxy`), int64 (S, S)."""
    ctx = ctx or default_context()
    g = _lib.as_u8(geno)
    S, N = g.shape
    out = np.zeros((S, S), dtype=np.int64)
    check(ctx.lib.gauss_gram_counts(ctx.handle, g.ctypes.data, S, N, g.strides[0],
                                    out.ctypes.data_as(C.POINTER(C.c_int64))))
    return out


def ld_matrix(geno, pop_off, pop_wgt=None, mode=MODE_WEIGHTED, diag=1.0, ctx=None):
    """LD matrix among the rows of `geno`.

    mode=MODE_WEIGHTED, diag=1.0  -> computeLD core (computeLD.cpp:95-116)
    mode=MODE_POOLED, diag=1+lambda -> CorG of jepeg (gene.cpp:305-315)
    """
    ctx = ctx or default_context()
    g = _lib.as_u8(geno)
    S = g.shape[0]
    po, w = _pops(pop_off, pop_wgt)
    out = np.zeros((S, S), dtype=np.float64)
    check(ctx.lib.gauss_ld(ctx.handle, int(mode), g.ctypes.data, S, g.strides[0],
                           po.ctypes.data_as(_ip), _lib.ptr(w, _dp), len(po) - 1, float(diag),
                           out.ctypes.data_as(_dp)))
    return out


def ld_per_pop(geno, pop_off, ctx=None):
    """Per-population Pearson r of every SNP pair i < j (prep_zmix5, zmix.cpp:158-176): (P, S(S-1)/2)."""
    ctx = ctx or default_context()
    g = _lib.as_u8(geno)
    po, _ = _pops(pop_off, None)
    S, P = g.shape[0], len(po) - 1
    out = np.zeros((P, S * (S - 1) // 2))
    check(ctx.lib.gauss_ld_per_pop(ctx.handle, g.ctypes.data, S, g.strides[0], po.ctypes.data_as(_ip), P,
                                   out.ctypes.data_as(_dp)))
    return out


def gene_ld_batch(geno, pop_off, gene_off, pop_wgt=None, mode=MODE_POOLED, diag=1.1, ctx=None):
    """LD blocks of all genes in one launch; returns a list of (n_g, n_g) arrays."""
    ctx = ctx or default_context()
    g = _lib.as_u8(geno)
    S = g.shape[0]
    po, w = _pops(pop_off, pop_wgt)
    go = np.ascontiguousarray(gene_off, dtype=np.int32)
    sizes = np.diff(go).astype(np.int64)
    out = np.zeros(int((sizes * sizes).sum()), dtype=np.float64)
    check(ctx.lib.gauss_gene_ld_batch(ctx.handle, int(mode), g.ctypes.data, S, g.strides[0],
                                      po.ctypes.data_as(_ip), _lib.ptr(w, _dp), len(po) - 1,
                                      go.ctypes.data_as(_ip), len(go) - 1, float(diag),
                                      out.ctypes.data_as(_dp)))
    blocks, o = [], 0
    for n in sizes:
        blocks.append(out[o:o + n * n].reshape(n, n))
        o += n * n
    return blocks


def _row_args(store, rows, pop_src_off):
    """(pointer, int32 row array or None, int32 offsets or None) of a row-store call; `store` is a numpy matrix
    (host rows), a RowStore (resident rows) or a raw device pointer."""
    r = None if rows is None else np.ascontiguousarray(rows, dtype=np.int32)
    so = None if pop_src_off is None else np.ascontiguousarray(pop_src_off, dtype=np.int32)
    if isinstance(store, RowStore):
        return store.ptr, store.ld, 1, r, so
    if isinstance(store, np.ndarray):
        return store.ctypes.data, store.strides[0], 0, r, so
    ptr, ld = store
    return ptr, ld, 1, r, so


def ld_matrix_rows(store, rows, pop_off, pop_wgt=None, mode=MODE_WEIGHTED, diag=1.0, fmt=_lib.GENO_2BIT, pop_src_off=None,
                   ctx=None):
    """gauss_ld_rows: the LD matrix of the store rows `rows` (computeLD on a packed / resident panel)."""
    ctx = ctx or default_context()
    ptr, ld, on_dev, r, so = _row_args(store, rows, pop_src_off)
    po, w = _pops(pop_off, pop_wgt)
    S = len(r)
    out = np.zeros((S, S))
    check(ctx.lib.gauss_ld_rows(ctx.handle, int(mode), ptr, ld, int(fmt), _lib.ptr(r, _ip), S, po.ctypes.data_as(_ip),
                                _lib.ptr(so, _ip), _lib.ptr(w, _dp), len(po) - 1, float(diag), on_dev, out.ctypes.data_as(_dp)))
    return out


def gene_ld_batch_rows(store, rows, pop_off, gene_off, pop_wgt=None, mode=MODE_POOLED, diag=1.1, fmt=_lib.GENO_2BIT,
                       pop_src_off=None, ctx=None):
    """gauss_gene_ld_batch_rows: LD blocks of all genes whose SNPs are the store rows `rows`."""
    ctx = ctx or default_context()
    ptr, ld, on_dev, r, so = _row_args(store, rows, pop_src_off)
    po, w = _pops(pop_off, pop_wgt)
    go = np.ascontiguousarray(gene_off, dtype=np.int32)
    sizes = np.diff(go).astype(np.int64)
    out = np.zeros(int((sizes * sizes).sum()))
    check(ctx.lib.gauss_gene_ld_batch_rows(ctx.handle, int(mode), ptr, ld, int(fmt), _lib.ptr(r, _ip), len(r), po.ctypes.data_as(_ip),
                                           _lib.ptr(so, _ip), _lib.ptr(w, _dp), len(po) - 1, go.ctypes.data_as(_ip), len(go) - 1,
                                           float(diag), on_dev, out.ctypes.data_as(_dp)))
    blocks, o = [], 0
    for n in sizes:
        blocks.append(out[o:o + n * n].reshape(n, n))
        o += n * n
    return blocks


class _Win:
    """Keeps the numpy buffers of one window alive next to its C descriptor."""

    def __init__(self, desc, mode, geno_m, geno_u, pop_off, pop_wgt, z1, lam, min_abs_eig,
                 want_mats, dev_ptrs=None, qcat=None, ld_codings=None, packed=None):
        self.po, self.w = _pops(pop_off, pop_wgt)
        self.z1 = np.ascontiguousarray(z1, dtype=np.float64)
        if dev_ptrs is None:
            self.gm = _lib.as_u8(geno_m)
            self.gu = _lib.as_u8(geno_u) if geno_u is not None and len(geno_u) else np.zeros((0, 1), np.uint8)
            M, U = self.gm.shape[0], self.gu.shape[0]
            pm, pu, ld = self.gm.ctypes.data, self.gu.ctypes.data, self.gm.strides[0]
            if U and self.gu.strides[0] != ld:
                raise ValueError("geno_m and geno_u must share a row stride")
        else:
            pm, pu, M, U, ld = dev_ptrs
        if packed is not None and packed.get("rows_m") is not None:
            M = len(packed["rows_m"])                       # matrices are row lists into a store
            U = len(packed["rows_u"]) if packed.get("rows_u") is not None else 0
        self.M, self.U = M, U
        self.ld_codings = ld_codings
        if ld_codings is not None:
            ncode = max(1, bin(int(ld_codings)).count("1"))
            want_mats = True
            self.b21_ld = np.zeros((ncode * U, M))
        self.z = np.zeros(U)
        self.info = np.zeros(U)
        self.status = np.zeros(1, dtype=np.int32)
        self.b11 = np.zeros((M, M)) if want_mats else None
        self.b21 = np.zeros((U, M)) if want_mats else None
        desc.mode = int(mode)
        desc.n_pop = len(self.po) - 1
        desc.pop_off = self.po.ctypes.data_as(_ip)
        desc.pop_wgt = _lib.ptr(self.w, _dp)
        desc.n_measured, desc.n_unmeasured = M, U
        desc.geno_m, desc.geno_u, desc.ld = pm, pu, ld
        desc.z1 = self.z1.ctypes.data_as(_dp)
        desc.lambda_, desc.min_abs_eig = float(lam), float(min_abs_eig)
        desc.out_z = self.z.ctypes.data_as(_dp)
        desc.out_info = self.info.ctypes.data_as(_dp)
        desc.out_status = self.status.ctypes.data_as(_ip)
        desc.out_b11 = _lib.ptr(self.b11, _dp)
        desc.out_b21 = _lib.ptr(self.b21, _dp)
        if packed is not None:
            # packed=dict(fmt=GENO_*, rows_m=, rows_u=, pop_src_off=): rows taken from a row store (dev_ptrs or
            # geno_m/geno_u give its base pointer and stride), optionally 2-bit packed (include/gauss_hip.h)
            desc.geno_format = int(packed.get("fmt", _lib.GENO_U8))
            self._rows = []
            for key in ("rows_m", "rows_u", "pop_src_off"):
                a = packed.get(key)
                if a is not None:
                    a = np.ascontiguousarray(a, dtype=np.int32)
                    self._rows.append(a)
                    setattr(desc, key, a.ctypes.data_as(_ip))
        if ld_codings is not None:
            desc.kind = _lib.WIN_LD
            desc.u_codings = int(ld_codings)
            desc.out_b21 = _lib.ptr(self.b21_ld, _dp)
            desc.out_z = desc.out_info = None
            desc.z1 = None
        self.qcat = qcat
        if qcat is not None:
            n_head, n_pred, eig_cutoff = qcat
            self.r = np.zeros(n_pred + U)
            self.num_eig = np.zeros(1, dtype=np.int32)
            desc.kind = _lib.WIN_QCAT
            desc.n_head_measured, desc.n_pred_measured = int(n_head), int(n_pred)
            desc.eig_cutoff = float(eig_cutoff)
            desc.out_r = self.r.ctypes.data_as(_dp)
            desc.out_num_eig = self.num_eig.ctypes.data_as(_ip)
            desc.out_z = desc.out_info = None

    def result(self):
        if self.ld_codings is not None:
            return dict(b11=self.b11, b21=self.b21_ld, status=int(self.status[0]))
        if self.qcat is not None:
            out = dict(r=self.r, num_eig=int(self.num_eig[0]), status=int(self.status[0]))
            if self.b11 is not None:
                out["b11"], out["b21"] = self.b11, self.b21
            return out
        out = dict(z=self.z, info=self.info, status=int(self.status[0]))
        if self.b11 is not None:
            out["b11"], out["b21"] = self.b11, self.b21
        return out


def impute_window(mode, geno_m, geno_u, pop_off, pop_wgt, z1, lam=0.1, min_abs_eig=1e-5,
                  want_mats=False, ctx=None):
    """run_dist (mode 0, dist.cpp:129-227) / run_distmix (mode 1, distmix.cpp:138-253)."""
    ctx = ctx or default_context()
    desc = WindowDesc()
    win = _Win(desc, mode, geno_m, geno_u, pop_off, pop_wgt, z1, lam, min_abs_eig, want_mats)
    check(ctx.lib.gauss_impute_window(ctx.handle, C.byref(desc)))
    return win.result()


def qcat_window(mode, geno_m, geno_u, pop_off, pop_wgt, z1, n_head, n_pred, lam=0.1, eig_cutoff=0.01,
                want_mats=False, ctx=None):
    """run_qcat (mode 0, qcat.cpp:134-262) / run_qcatmix (mode 1, qcatmix.cpp): correlation r between the
    whitened measured Z-scores and the whitened LD column of every tested SNP -- first the n_pred measured
    SNPs that follow the n_head left-wing ones, then the unmeasured SNPs -- plus CountPC's num_eig."""
    ctx = ctx or default_context()
    desc = WindowDesc()
    win = _Win(desc, mode, geno_m, geno_u, pop_off, pop_wgt, z1, lam, 1e-5, want_mats,
               qcat=(n_head, n_pred, eig_cutoff))
    check(ctx.lib.gauss_impute_window(ctx.handle, C.byref(desc)))
    return win.result()


def ld_window(mode, geno_m, geno_u, pop_off, pop_wgt, lam=0.0, codings=_lib.CODE_ADDITIVE, ctx=None):
    """Raw LD export (prep_qcat.cpp:104-132, prep_qcatmix.cpp:136-221): B11 among the measured rows
    (diagonal 1 + lam) and B21 of the geno_u rows against them, one block of rows per coding in
    `codings` (additive, dominant, recessive)."""
    ctx = ctx or default_context()
    desc = WindowDesc()
    M = len(geno_m)
    win = _Win(desc, mode, geno_m, geno_u, pop_off, pop_wgt, np.zeros(M), lam, 1e-5, True, ld_codings=codings)
    check(ctx.lib.gauss_impute_window(ctx.handle, C.byref(desc)))
    return win.result()


class PinnedArray:
    """A numpy uint8 matrix in page-locked host memory (gauss_pinned_alloc): uploads from it run at PCIe speed."""

    def __init__(self, shape, ctx=None):
        self.ctx = ctx or default_context()
        n = int(np.prod(shape))
        p = C.c_void_p()
        check(self.ctx.lib.gauss_pinned_alloc(self.ctx.handle, max(n, 1), C.byref(p)))
        self.ptr = p.value
        self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(n, 1),))[:n].reshape(shape)

    def close(self):
        if self.ptr:
            self.array = None
            self.ctx.lib.gauss_pinned_free(self.ctx.handle, C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RowStore:
    """Genotype rows resident in HBM (gauss_store_upload): a whole packed chromosome is uploaded once and
    windows name their rows by index."""

    @classmethod
    def from_file(cls, path, file_offset, n_rows, ld, ctx=None, reserve_only=False):
        """The rows are a section of a FILE (a packed panel's genotype section): gauss_store_upload_fd, or with reserve_only
        gauss_store_alloc + fill() through gauss_store_fill_fd -- pread straight into the pinned staging buffers."""
        import os
        self = cls.__new__(cls)
        self.ctx = ctx or default_context()
        self.n_rows, self.ld = int(n_rows), int(ld)
        self._host = None
        self._fd, self._file_off = os.open(path, os.O_RDONLY), int(file_offset)
        p = C.c_void_p()
        try:
            if reserve_only:
                check(self.ctx.lib.gauss_store_alloc(self.ctx.handle, self.n_rows * self.ld, C.byref(p)))
            else:
                check(self.ctx.lib.gauss_store_upload_fd(self.ctx.handle, self._fd, C.c_int64(self._file_off), C.c_int64(self.n_rows * self.ld), C.byref(p)))
        except Exception:
            os.close(self._fd)
            raise
        self.ptr = p.value
        return self

    def __init__(self, rows, ctx=None, asynchronous=False, reserve_only=False):
        """asynchronous: gauss_store_upload_async -- the rows travel in the background (keep `rows` alive until
        wait() has returned); wait(n_rows) makes the context's stream wait for the first n_rows rows.
        reserve_only: gauss_store_alloc -- the store is only reserved; fill(r0, r1) brings rows [r0, r1) up (gauss_store_fill)."""
        self.ctx = ctx or default_context()
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        self.n_rows, self.ld = rows.shape
        self._fd = None
        p = C.c_void_p()
        if reserve_only:
            self._host = rows
            check(self.ctx.lib.gauss_store_alloc(self.ctx.handle, rows.nbytes, C.byref(p)))
        elif asynchronous:
            self._host = rows
            check(self.ctx.lib.gauss_store_upload_async(self.ctx.handle, rows.ctypes.data_as(C.c_void_p), rows.nbytes, C.byref(p)))
        else:
            check(self.ctx.lib.gauss_store_upload(self.ctx.handle, rows.ctypes.data_as(C.c_void_p), rows.nbytes, C.byref(p)))
        self.ptr = p.value

    def fill(self, r0, r1):
        """Rows [r0, r1) of a reserve_only store travel now (on a queue of their own); returns when they have landed."""
        if self._fd is not None:
            check(self.ctx.lib.gauss_store_fill_fd(self.ctx.handle, C.c_void_p(self.ptr), self._fd, C.c_int64(self._file_off),
                                                   C.c_int64(int(r0) * self.ld), C.c_int64((int(r1) - int(r0)) * self.ld)))
            return
        check(self.ctx.lib.gauss_store_fill(self.ctx.handle, C.c_void_p(self.ptr), self._host.ctypes.data_as(C.c_void_p),
                                            int(r0) * self.ld, (int(r1) - int(r0)) * self.ld))

    def wait(self, n_rows=0):
        """Rows [0, n_rows) have landed for everything queued on the context afterwards (0: all rows, host waits)."""
        check(self.ctx.lib.gauss_store_wait(self.ctx.handle, C.c_void_p(self.ptr), int(n_rows) * self.ld))

    def close(self):
        if self.ptr:
            self.ctx.lib.gauss_store_free(self.ctx.handle, C.c_void_p(self.ptr))
            self.ptr = None
        if getattr(self, "_fd", None) is not None:
            import os
            os.close(self._fd)
            self._fd = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Job:
    """A batch of windows sharing every launch (gauss_job_*)."""

    def __init__(self, windows, ctx=None, on_device=False, want_mats=False):
        """windows: list of dicts(mode, geno_m, geno_u, pop_off, pop_wgt, z1[, lam, min_abs_eig])
        or, with on_device=True, dicts carrying dev=(ptr_m, ptr_u, M, U, ld) instead of arrays."""
        self.ctx = ctx or default_context()
        n = len(windows)
        self.descs = (WindowDesc * n)()
        self.wins = []
        for i, w in enumerate(windows):
            self.wins.append(_Win(self.descs[i], w["mode"], w.get("geno_m"), w.get("geno_u"),
                                  w["pop_off"], w.get("pop_wgt"), w["z1"], w.get("lam", 0.1),
                                  w.get("min_abs_eig", 1e-5), want_mats, w.get("dev"), w.get("qcat"), w.get("ld_codings"), w.get("packed")))
        h = C.c_void_p()
        check(self.ctx.lib.gauss_job_create(self.ctx.handle, self.descs, n, 1 if on_device else 0,
                                            C.byref(h)))
        self.handle = h

    def run(self):
        check(self.ctx.lib.gauss_job_run(self.handle))

    def fetch(self):
        check(self.ctx.lib.gauss_job_fetch(self.handle))
        return [w.result() for w in self.wins]

    def counters(self):
        """gauss_job_counters: this job's own runs -- queued merged / demoted, give-ups repaired inside a fetch, re-runs that failed."""
        out = (C.c_int64 * 4)()
        check(self.ctx.lib.gauss_job_counters(self.handle, out))
        return dict(merged=int(out[0]), demoted=int(out[1]), giveups=int(out[2]), rerun_failed=int(out[3]))

    def profile(self, enable=True):
        check(self.ctx.lib.gauss_job_profile(self.handle, 1 if enable else 0))

    def profile_get(self, kernel=0):
        ms, n = C.c_double(), C.c_int64()
        check(self.ctx.lib.gauss_job_profile_get(self.handle, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def work(self):
        a, b, c, d = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        check(self.ctx.lib.gauss_job_work(self.handle, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(ld_flops=a.value, solve_flops=b.value, bytes=c.value, imputed_snps=d.value)

    def stats(self):
        out = (C.c_double * 4)()
        check(self.ctx.lib.gauss_job_stats(self.handle, out))
        return dict(items=int(out[0]), executed_flops=out[1], slab_bytes=out[2], workspace_bytes=out[3])

    def close(self):
        if self.handle:
            # safe in any order with the context (interpreter shutdown destroys objects in no particular order):
            # gauss_hip_destroy leaves the jobs that outlive it as empty shells, which gauss_job_destroy frees
            self.ctx.lib.gauss_job_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
