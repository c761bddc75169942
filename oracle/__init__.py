"""CPU oracle for the GAUSS LD + DIST/DISTMIX hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product (``gauss_amd``) never does; it fails loudly when its HIP
library is missing instead of falling back to anything in here.

Parity status: **parity unpinned** for the numeric functions (the reference ships no tests
or golden vectors for this path and cannot be built here: no R/Rcpp/RcppEigen/Rmath).  The
C restatement (``gauss_oracle.c``) is cross-validated against the independent numpy/scipy
implementation in ``oracle_np.py``; the BGZF codec is pinned against the reference's own
``src/bgzf.c`` compiled into ``oracle/_ref/``.
"""
from .oracle_c import (  # noqa: F401
    load, calcor, calwgtcov, compute_ld, ld_pooled, run_impute, run_qcat, count_pc, ld_blocks, recode, ld_per_pop, make_pos_def, inv_mat,
    pnorm_upper, pchisq_upper, jepeg_gene_tail, gram_counts, build,
)
