/*
 * gauss_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * A plain-C, fp64, loop-literal restatement of the numeric hot path of
 * statsleelab/gauss (LD build + DIST/DISTMIX conditional-mean solve + the JEPEG
 * k x k tail).  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load this library; the shipped library
 * (libgauss_hip.so) never links, loads or calls it.
 *
 * PARITY STATUS: "parity unpinned" for the numeric functions.  The reference
 * ships no tests, golden vectors or fixtures for this path, and it cannot be
 * built or run in this image (no R / Rcpp / RcppEigen / Rmath).  The only known
 * answers (vignette print-outs) need the external 33KG panel.  The restatement
 * is cross-validated against an independent numpy/scipy (LAPACK) implementation
 * (oracle/oracle_np.py, tests/test_oracle.py), and the BGZF codec used by the
 * feeder is pinned against the reference's own src/bgzf.c compiled into
 * oracle/_ref/ (see oracle/Makefile).
 *
 * Third-party arithmetic the reference takes from libraries that are not in
 * /root/reference (restated here from their published algorithms):
 *   - Eigen (via CRAN RcppEigen, unpinned; DESCRIPTION:12-16)
 *       SelfAdjointEigenSolver  -> symmetric tridiagonalisation + implicit QL
 *       FullPivLU::inverse      -> Gaussian elimination with complete pivoting
 *   - Rmath (R itself)  R::pnorm5(x,0,1,0,0), R::pchisq(x,df,0,0)
 *
 * Genotype layout used by every function below: one SNP = one row of `n`
 * ASCII characters '0'/'1'/'2' -- the concatenation, in panel order, of the
 * per-population strings the reference keeps in Snp::genotype_vec_
 * (src/snp.h:109).  pop_off[0..P] gives the column range of each population.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* a1. CalCor(vector<string>&, vector<string>&)      src/util.cpp:49-70      */
/* ------------------------------------------------------------------------- */
ORC_API double orc_calcor(const char* x, const char* y, const int* pop_off, int P)
{
    int num_samples = 0;
    double xij = 0, yij = 0, sumx = 0, sumy = 0, sumxsq = 0, sumysq = 0, sumxy = 0;
    for (int i = 0; i < P; i++) {                    /* util.cpp:53  loop over pop strings */
        int m = pop_off[i + 1] - pop_off[i];         /* util.cpp:54  x[i].length()         */
        const char* xs = x + pop_off[i];
        const char* ys = y + pop_off[i];
        for (int j = 0; j < m; j++) {                /* util.cpp:55-63 */
            xij = (double)(xs[j] - '0');
            yij = (double)(ys[j] - '0');
            sumx += xij;
            sumy += yij;
            sumxsq += xij * xij;
            sumysq += yij * yij;
            sumxy += xij * yij;
        }
        num_samples += m;
    }
    double numer = num_samples * sumxy - sumx * sumy;                       /* util.cpp:66 */
    double denor = sqrt((num_samples) * sumxsq - sumx * sumx) *
                   sqrt((num_samples) * sumysq - sumy * sumy);              /* util.cpp:67 */
    double r = numer / denor;                                               /* util.cpp:68 */
    return r;
}

/* ------------------------------------------------------------------------- */
/* a2. CalWgtCov(x, y, pop_wgt_vec)                  src/util.cpp:103-124    */
/* ------------------------------------------------------------------------- */
ORC_API double orc_calwgtcov(const char* x, const char* y, const int* pop_off, int P,
                             const double* pop_wgt)
{
    double wsumcov = 0, wsum_mi_mj = 0, wsum_mi = 0, wsum_mj = 0;
    for (int i = 0; i < P; i++) {
        int m = pop_off[i + 1] - pop_off[i];
        const char* xs = x + pop_off[i];
        const char* ys = y + pop_off[i];
        double xij = 0, yij = 0, sumx = 0, sumy = 0, sumxy = 0;
        double wgt_val = pop_wgt[i];
        for (int j = 0; j < m; j++) {
            xij = (double)(xs[j] - '0');
            yij = (double)(ys[j] - '0');
            sumx += xij;
            sumy += yij;
            sumxy += xij * yij;
        }
        double factor = ((double)m) / (m - 1);                              /* util.cpp:117 */
        wsumcov += wgt_val * factor * (m * sumxy - sumx * sumy);            /* util.cpp:118 */
        wsum_mi_mj += wgt_val * (sumx / m) * (sumy / m);                    /* util.cpp:119 */
        wsum_mi += wgt_val * (sumx / m);                                    /* util.cpp:120 */
        wsum_mj += wgt_val * (sumy / m);                                    /* util.cpp:121 */
    }
    return (wsumcov + wsum_mi_mj - wsum_mi * wsum_mj);                      /* util.cpp:123 */
}

/* ------------------------------------------------------------------------- */
/* Dense helpers (column-major n x n, like Eigen::MatrixXd)                   */
/* ------------------------------------------------------------------------- */

/* Symmetric eigendecomposition A = V diag(d) V^T.
 * Householder tridiagonalisation followed by implicit-shift QL -- the same
 * two-stage scheme Eigen::SelfAdjointEigenSolver uses (util.cpp:304).
 * V is column-major; on return column k is the eigenvector of d[k], d ascending.
 * Returns 0 on success, 1 if the QL iteration does not converge (the analogue
 * of solver.info() != Eigen::Success, util.cpp:305). */
static int sym_eig(int n, double* V, double* d, double* e)
{
#define VV(i, j) V[(size_t)(j) * n + (i)]
    /* --- tridiagonalise (Householder, accumulate transforms) --- */
    for (int j = 0; j < n; j++) d[j] = VV(n - 1, j);
    for (int i = n - 1; i > 0; i--) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; k++) scale += fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; j++) {
                d[j] = VV(i - 1, j);
                VV(i, j) = 0.0;
                VV(j, i) = 0.0;
            }
        } else {
            for (int k = 0; k < i; k++) { d[k] /= scale; h += d[k] * d[k]; }
            double f = d[i - 1];
            double g = sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h = h - f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; j++) e[j] = 0.0;
            for (int j = 0; j < i; j++) {
                f = d[j];
                VV(j, i) = f;
                g = e[j] + VV(j, j) * f;
                for (int k = j + 1; k <= i - 1; k++) {
                    g += VV(k, j) * d[k];
                    e[k] += VV(k, j) * f;
                }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; j++) { e[j] /= h; f += e[j] * d[j]; }
            double hh = f / (h + h);
            for (int j = 0; j < i; j++) e[j] -= hh * d[j];
            for (int j = 0; j < i; j++) {
                f = d[j];
                g = e[j];
                for (int k = j; k <= i - 1; k++) VV(k, j) -= (f * e[k] + g * d[k]);
                d[j] = VV(i - 1, j);
                VV(i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < n - 1; i++) {
        VV(n - 1, i) = VV(i, i);
        VV(i, i) = 1.0;
        double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; k++) d[k] = VV(k, i + 1) / h;
            for (int j = 0; j <= i; j++) {
                double g = 0.0;
                for (int k = 0; k <= i; k++) g += VV(k, i + 1) * VV(k, j);
                for (int k = 0; k <= i; k++) VV(k, j) -= g * d[k];
            }
        }
        for (int k = 0; k <= i; k++) VV(k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; j++) { d[j] = VV(n - 1, j); VV(n - 1, j) = 0.0; }
    VV(n - 1, n - 1) = 1.0;
    e[0] = 0.0;

    /* --- implicit QL on the tridiagonal --- */
    for (int i = 1; i < n; i++) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < n; l++) {
        double t = fabs(d[l]) + fabs(e[l]);
        if (t > tst1) tst1 = t;
        int m = l;
        while (m < n) {
            if (fabs(e[m]) <= eps * tst1) break;
            m++;
        }
        if (m >= n) m = n - 1;
        if (m > l) {
            int iter = 0;
            do {
                if (++iter > 300) return 1;       /* no convergence */
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; i++) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c;
                double el1 = e[l + 1];
                double s = 0.0, s2 = 0.0;
                for (int i = m - 1; i >= l; i--) {
                    c3 = c2;
                    c2 = c;
                    s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < n; k++) {
                        h = VV(k, i + 1);
                        VV(k, i + 1) = s * VV(k, i) + c * h;
                        VV(k, i) = c * VV(k, i) - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (fabs(e[l]) > eps * tst1);
        }
        d[l] = d[l] + f;
        e[l] = 0.0;
    }
    /* non-finite input never converges meaningfully: report failure */
    for (int i = 0; i < n; i++)
        if (!isfinite(d[i])) return 1;
    /* sort ascending (Eigen returns eigenvalues in increasing order) */
    for (int i = 0; i < n - 1; i++) {
        int k = i;
        double p = d[i];
        for (int j = i + 1; j < n; j++)
            if (d[j] < p) { k = j; p = d[j]; }
        if (k != i) {
            d[k] = d[i];
            d[i] = p;
            for (int j = 0; j < n; j++) {
                double t = VV(j, i);
                VV(j, i) = VV(j, k);
                VV(j, k) = t;
            }
        }
    }
    return 0;
#undef VV
}

/* a6. MakePosDef(m1, min_abs_eig)                   src/util.cpp:302-318
 * Returns 1 if the matrix was rebuilt (some eigenvalue < min_abs_eig), 0 if it
 * was left untouched, -1 if the eigen-solver failed (matrix left untouched,
 * util.cpp:305-307). */
ORC_API int orc_make_pos_def(double* A, int n, double min_abs_eig)
{
    if (n <= 0) return 0;
    double* V = (double*)malloc(sizeof(double) * (size_t)n * n);
    double* d = (double*)malloc(sizeof(double) * n);
    double* e = (double*)malloc(sizeof(double) * n);
    memcpy(V, A, sizeof(double) * (size_t)n * n);
    int rc = sym_eig(n, V, d, e);
    int changed = 0;
    if (rc != 0) {
        changed = -1;
    } else {
        double mn = d[0];
        for (int i = 1; i < n; i++) if (d[i] < mn) mn = d[i];
        if (mn < min_abs_eig) {                                              /* util.cpp:310 */
            for (int i = 0; i < n; i++) if (d[i] < min_abs_eig) d[i] = min_abs_eig;
            /* m1 = V * diag * V^T                                            util.cpp:316 */
            for (int j = 0; j < n; j++)
                for (int i = 0; i < n; i++) {
                    double s = 0.0;
                    for (int k = 0; k < n; k++) s += V[(size_t)k * n + i] * d[k] * V[(size_t)k * n + j];
                    A[(size_t)j * n + i] = s;
                }
            changed = 1;
        }
    }
    free(V); free(d); free(e);
    return changed;
}

/* a7. InvMat: m1.fullPivLu().inverse()               src/util.cpp:298-300
 * Gaussian elimination with complete pivoting, P A Q = L U, then
 * inverse = Q U^-1 L^-1 P (Eigen FullPivLU).  Column-major. */
ORC_API void orc_inv_mat(double* inv, const double* A_in, int n)
{
    size_t nn = (size_t)n * n;
    double* A = (double*)malloc(sizeof(double) * nn);
    int* rp = (int*)malloc(sizeof(int) * n);   /* row transpositions    */
    int* cp = (int*)malloc(sizeof(int) * n);   /* column transpositions */
    memcpy(A, A_in, sizeof(double) * nn);
#define AA(i, j) A[(size_t)(j) * n + (i)]
    for (int k = 0; k < n; k++) {
        int pi = k, pj = k;
        double best = -1.0;
        for (int j = k; j < n; j++)
            for (int i = k; i < n; i++) {
                double v = fabs(AA(i, j));
                if (v > best) { best = v; pi = i; pj = j; }
            }
        rp[k] = pi; cp[k] = pj;
        if (pi != k) for (int j = 0; j < n; j++) { double t = AA(k, j); AA(k, j) = AA(pi, j); AA(pi, j) = t; }
        if (pj != k) for (int i = 0; i < n; i++) { double t = AA(i, k); AA(i, k) = AA(i, pj); AA(i, pj) = t; }
        double piv = AA(k, k);
        for (int i = k + 1; i < n; i++) AA(i, k) /= piv;
        for (int j = k + 1; j < n; j++) {
            double u = AA(k, j);
            if (u != 0.0)
                for (int i = k + 1; i < n; i++) AA(i, j) -= AA(i, k) * u;
        }
    }
    /* Solve A X = I column by column:  X = Q U^-1 L^-1 P */
    double* col = (double*)malloc(sizeof(double) * n);
    for (int c = 0; c < n; c++) {
        for (int i = 0; i < n; i++) col[i] = (i == c) ? 1.0 : 0.0;
        for (int k = 0; k < n; k++)                      /* apply P */
            if (rp[k] != k) { double t = col[k]; col[k] = col[rp[k]]; col[rp[k]] = t; }
        for (int k = 0; k < n; k++)                      /* L^-1 (unit lower) */
            for (int i = k + 1; i < n; i++) col[i] -= AA(i, k) * col[k];
        for (int k = n - 1; k >= 0; k--) {               /* U^-1 */
            col[k] /= AA(k, k);
            for (int i = 0; i < k; i++) col[i] -= AA(i, k) * col[k];
        }
        for (int k = n - 1; k >= 0; k--)                 /* apply Q */
            if (cp[k] != k) { double t = col[k]; col[k] = col[cp[k]]; col[cp[k]] = t; }
        memcpy(inv + (size_t)c * n, col, sizeof(double) * n);
    }
#undef AA
    free(col); free(A); free(rp); free(cp);
}

/* a13. R::pnorm5(x, 0, 1, lower=0, log=0) = upper normal tail (dist.cpp:101) */
ORC_API double orc_pnorm_upper(double x) { return 0.5 * erfc(x / 1.4142135623730951); }

/* a13. R::pchisq(x, df, lower=0, log=0) for integer df >= 1 (gene.cpp:509).
 * Closed-form regularised upper incomplete gamma Q(df/2, x/2). */
ORC_API double orc_pchisq_upper(double x, int df)
{
    if (df <= 0) return NAN;
    if (!(x > 0.0)) return (x != x) ? NAN : 1.0;
    double h = 0.5 * x;
    if ((df & 1) == 0) {
        double term = 1.0, sum = 1.0;
        for (int k = 1; k < df / 2; k++) { term *= h / k; sum += term; }
        return exp(-h) * sum;
    } else {
        double q = erfc(sqrt(h));
        if (df > 1) {
            /* sum_{k=1}^{(df-1)/2} h^{k-1/2} / Gamma(k+1/2) */
            double term = sqrt(h) / 0.886226925452758;   /* h^{1/2}/Gamma(3/2) */
            double sum = term;
            for (int k = 2; k <= (df - 1) / 2; k++) { term *= h / (k - 0.5); sum += term; }
            q += exp(-h) * sum;
        }
        return q;
    }
}

/* ------------------------------------------------------------------------- */
/* a3. computeLD core                                src/computeLD.cpp:95-116 */
/*     cor is S x S column-major (Rcpp NumericMatrix).                        */
/* ------------------------------------------------------------------------- */
ORC_API int orc_compute_ld(const char* geno, long ld, int S, const int* pop_off, int P,
                           const double* pop_wgt, double* cor)
{
    double* sd = (double*)malloc(sizeof(double) * (S > 0 ? S : 1));
    for (int i = 0; i < S; i++) {                                            /* :100-103 */
        double v = orc_calwgtcov(geno + (size_t)i * ld, geno + (size_t)i * ld, pop_off, P, pop_wgt);
        sd[i] = sqrt(v);
    }
    for (int i = 0; i < S; i++) {                                            /* :106-116 */
        cor[(size_t)i * S + i] = 1.0;
        double stdi = sd[i];
        for (int j = i + 1; j < S; j++) {
            double stdj = sd[j];
            double cov = orc_calwgtcov(geno + (size_t)i * ld, geno + (size_t)j * ld, pop_off, P, pop_wgt);
            double c = cov / (stdi * stdj);
            cor[(size_t)j * S + i] = c;
            cor[(size_t)i * S + j] = c;
        }
    }
    free(sd);
    return 0;
}

/* Pooled-Pearson LD with a chosen diagonal (CorG of gene.cpp:305-315 with
 * diag = 1 + lambda, or B11 of dist.cpp:171-179). Column-major S x S. */
ORC_API int orc_ld_pooled(const char* geno, long ld, int S, const int* pop_off, int P,
                          double diag, double* cor)
{
    for (int i = 0; i < S; i++) {
        cor[(size_t)i * S + i] = diag;
        for (int j = i + 1; j < S; j++) {
            double v = orc_calcor(geno + (size_t)i * ld, geno + (size_t)j * ld, pop_off, P);
            cor[(size_t)j * S + i] = v;
            cor[(size_t)i * S + j] = v;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* a4/a5. run_dist (dist.cpp:156-202) and run_distmix (distmix.cpp:165-228)   */
/*   mode 0 = pooled Pearson (CalCor), mode 1 = weighted (CalWgtCov).         */
/*   geno_m: M measured rows, geno_u: U unmeasured rows, z1[M].               */
/*   Optional outputs b11 (M x M col-major, after MakePosDef) and b21         */
/*   (U x M row-major) may be NULL.                                           */
/*   Returns the MakePosDef code (0 untouched / 1 rebuilt / -1 solver fail).  */
/* ------------------------------------------------------------------------- */
ORC_API int orc_run_impute(int mode, const char* geno_m, int M, const char* geno_u, int U, long ld,
                           const int* pop_off, int P, const double* pop_wgt, const double* z1,
                           double lambda, double min_abs_eig, double* out_z, double* out_info,
                           double* b11_out, double* b21_out)
{
    size_t MM = (size_t)M * M;
    double* B11 = (double*)calloc(MM ? MM : 1, sizeof(double));
    double* B11Inv = (double*)calloc(MM ? MM : 1, sizeof(double));
    double* b21 = (double*)calloc(M ? M : 1, sizeof(double));
    double* b21B11Inv = (double*)calloc(M ? M : 1, sizeof(double));
    double* sd = (double*)calloc((size_t)(M + U) ? (size_t)(M + U) : 1, sizeof(double));

    if (mode == 1) {                                                        /* distmix.cpp:180-187 */
        for (int i = 0; i < M; i++)
            sd[i] = sqrt(orc_calwgtcov(geno_m + (size_t)i * ld, geno_m + (size_t)i * ld, pop_off, P, pop_wgt));
        for (int i = 0; i < U; i++)
            sd[M + i] = sqrt(orc_calwgtcov(geno_u + (size_t)i * ld, geno_u + (size_t)i * ld, pop_off, P, pop_wgt));
    }
    for (int i = 0; i < M; i++) {                                           /* dist.cpp:171-179 / distmix.cpp:190-200 */
        B11[(size_t)i * M + i] = 1.0 + lambda;
        for (int j = i + 1; j < M; j++) {
            double v;
            if (mode == 0) {
                v = orc_calcor(geno_m + (size_t)i * ld, geno_m + (size_t)j * ld, pop_off, P);
            } else {
                double cov = orc_calwgtcov(geno_m + (size_t)i * ld, geno_m + (size_t)j * ld, pop_off, P, pop_wgt);
                v = cov / (sd[i] * sd[j]);
            }
            B11[(size_t)j * M + i] = v;
            B11[(size_t)i * M + j] = v;
        }
    }
    int mpd = orc_make_pos_def(B11, M, min_abs_eig);                        /* dist.cpp:181 */
    orc_inv_mat(B11Inv, B11, M);                                            /* dist.cpp:182 */
    if (b11_out) memcpy(b11_out, B11, sizeof(double) * MM);

    for (int i = 0; i < U; i++) {                                           /* dist.cpp:187-202 */
        for (int j = 0; j < M; j++) {
            double v;
            if (mode == 0) {
                v = orc_calcor(geno_u + (size_t)i * ld, geno_m + (size_t)j * ld, pop_off, P);
            } else {
                double cov = orc_calwgtcov(geno_u + (size_t)i * ld, geno_m + (size_t)j * ld, pop_off, P, pop_wgt);
                v = cov / (sd[M + i] * sd[j]);
            }
            b21[j] = v;
        }
        if (b21_out) memcpy(b21_out + (size_t)i * M, b21, sizeof(double) * M);
        for (int c = 0; c < M; c++) {                /* b21B11Inv = b21 * B11Inv   (1xM)(MxM) */
            double s = 0.0;
            for (int k = 0; k < M; k++) s += b21[k] * B11Inv[(size_t)c * M + k];
            b21B11Inv[c] = s;
        }
        double z = 0.0, q = 0.0;
        for (int k = 0; k < M; k++) z += b21B11Inv[k] * z1[k];              /* dist.cpp:194 */
        for (int k = 0; k < M; k++) q += b21B11Inv[k] * b21[k];             /* dist.cpp:197 */
        double info = fabs(q);                                              /* dist.cpp:198 */
        out_z[i] = z / sqrt(info);                                          /* dist.cpp:200 */
        out_info[i] = info;                                                 /* dist.cpp:202 */
    }
    free(B11); free(B11Inv); free(b21); free(b21B11Inv); free(sd);
    return mpd;
}

/* ------------------------------------------------------------------------- */
/* n4. per-population LD of every SNP pair   src/zmix.cpp:158-176              */
/*   data_mat(row, k+1) = CalCor(snpi_geno_vec[k], snpj_geno_vec[k])           */
/*   (CalCor(std::string&, std::string&), util.cpp:153-169 == a1 on one pop).  */
/*   out: [P][S(S-1)/2], population-major, rows in the reference's (i, j>i)    */
/*   order.                                                                    */
/* ------------------------------------------------------------------------- */
ORC_API void orc_ld_per_pop(const char* geno, long ld, int S, const int* pop_off, int P, double* out)
{
    size_t npairs = (size_t)S * (S - 1) / 2;
    size_t row = 0;
    for (int i = 0; i < S; i++)
        for (int j = i + 1; j < S; j++) {
            for (int k = 0; k < P; k++)
                out[(size_t)k * npairs + row] = orc_calcor(geno + (size_t)i * ld, geno + (size_t)j * ld, pop_off + k, 1);
            row++;
        }
}

/* ------------------------------------------------------------------------- */
/* n2. raw LD export   src/prep_qcat.cpp:104-132 (mode 0, CalCor)              */
/*                     src/prep_qcatmix.cpp:136-158, 187-197 (mode 1)          */
/*   b11: M x M, diagonal `diag`; b21: U x M row-major, x = geno_u row (first  */
/*   argument of CalCor / CalWgtCov), y = measured row.  Recoded (dominant /    */
/*   recessive) rows are passed in already recoded (gauss.cpp:1196-1250).       */
/* ------------------------------------------------------------------------- */
ORC_API int orc_ld_blocks(int mode, const char* geno_m, int M, const char* geno_u, int U, long ld,
                          const int* pop_off, int P, const double* pop_wgt, double diag,
                          double* b11, double* b21)
{
    double* sd = (double*)calloc((size_t)(M + U) ? (size_t)(M + U) : 1, sizeof(double));
    if (mode == 1) {
        for (int i = 0; i < M; i++)
            sd[i] = sqrt(orc_calwgtcov(geno_m + (size_t)i * ld, geno_m + (size_t)i * ld, pop_off, P, pop_wgt));
        for (int i = 0; i < U; i++)
            sd[M + i] = sqrt(orc_calwgtcov(geno_u + (size_t)i * ld, geno_u + (size_t)i * ld, pop_off, P, pop_wgt));
    }
    if (b11)
        for (int i = 0; i < M; i++) {
            b11[(size_t)i * M + i] = diag;
            for (int j = i + 1; j < M; j++) {
                double v;
                if (mode == 0) v = orc_calcor(geno_m + (size_t)i * ld, geno_m + (size_t)j * ld, pop_off, P);
                else v = orc_calwgtcov(geno_m + (size_t)i * ld, geno_m + (size_t)j * ld, pop_off, P, pop_wgt) / (sd[i] * sd[j]);
                b11[(size_t)i * M + j] = v;
                b11[(size_t)j * M + i] = v;
            }
        }
    for (int i = 0; i < U; i++)
        for (int j = 0; j < M; j++) {
            double v;
            if (mode == 0) v = orc_calcor(geno_u + (size_t)i * ld, geno_m + (size_t)j * ld, pop_off, P);
            else v = orc_calwgtcov(geno_u + (size_t)i * ld, geno_m + (size_t)j * ld, pop_off, P, pop_wgt) / (sd[M + i] * sd[j]);
            b21[(size_t)i * M + j] = v;
        }
    free(sd);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* n1. run_qcat / run_qcatmix core          src/qcat.cpp:166-245 (mode 0)      */
/*                                          src/qcatmix.cpp:179-277 (mode 1)   */
/*   B11 (+lambda on the diagonal) and B21 as in DIST / DISTMIX; then          */
/*     num_eig = CountPC(B11, eig_cutoff)               util.cpp:355-388       */
/*     L = LLT(B11).matrixL()                           util.cpp:271-274       */
/*     LInv = L.fullPivLu().inverse()                   util.cpp:298-300       */
/*     r_i  = CalCor(LInv Z1, LInv b_i)                 util.cpp:194-203       */
/*   b_i: rows n_head .. n_head+n_predm-1 of B11 (tested measured SNPs), then  */
/*   the U rows of B21.  out_r has n_predm + U entries.  The t score, chi^2    */
/*   and p-value are scalar tails of (num_eig, r) formed by the caller.        */
/*   MakePosDef is commented out in the reference (qcat.cpp:206) -- not run.   */
/* ------------------------------------------------------------------------- */
static double vec_cor(const double* x, const double* y, int n)              /* util.cpp:194-203 */
{
    double mx = 0.0, my = 0.0;
    for (int i = 0; i < n; i++) { mx += x[i]; my += y[i]; }
    mx /= n; my /= n;
    double sxx = 0.0, syy = 0.0, sxy = 0.0;
    for (int i = 0; i < n; i++) {
        double dx = x[i] - mx, dy = y[i] - my;
        sxx += dx * dx; syy += dy * dy; sxy += dx * dy;
    }
    return sxy / sqrt(sxx * syy);
}

ORC_API int orc_count_pc(const double* A, int n, double eig_cutoff)         /* util.cpp:355-388 */
{
    if (n <= 0) return 0;
    double* V = (double*)malloc(sizeof(double) * (size_t)n * n);
    double* d = (double*)malloc(sizeof(double) * n);
    double* e = (double*)malloc(sizeof(double) * n);
    memcpy(V, A, sizeof(double) * (size_t)n * n);
    int num_eig = n;
    if (sym_eig(n, V, d, e) == 0) {
        double mn = d[0];
        for (int i = 1; i < n; i++) if (d[i] < mn) mn = d[i];
        if (mn < eig_cutoff)
            for (int i = 0; i < n; i++) if (d[i] < eig_cutoff) num_eig--;
    }
    free(V); free(d); free(e);
    return num_eig;
}

ORC_API int orc_run_qcat(int mode, const char* geno_m, int M, const char* geno_u, int U, long ld,
                         const int* pop_off, int P, const double* pop_wgt, const double* z1,
                         double lambda, double eig_cutoff, int n_head, int n_predm,
                         double* out_r, int* out_num_eig, double* b11_out, double* b21_out)
{
    size_t MM = (size_t)M * M;
    double* B11 = (double*)calloc(MM ? MM : 1, sizeof(double));
    double* L = (double*)calloc(MM ? MM : 1, sizeof(double));
    double* LInv = (double*)calloc(MM ? MM : 1, sizeof(double));
    double* b = (double*)calloc(M ? M : 1, sizeof(double));
    double* wz = (double*)calloc(M ? M : 1, sizeof(double));
    double* wb = (double*)calloc(M ? M : 1, sizeof(double));
    double* sd = (double*)calloc((size_t)(M + U) ? (size_t)(M + U) : 1, sizeof(double));

    if (mode == 1) {                                                        /* qcatmix.cpp:197-204 */
        for (int i = 0; i < M; i++)
            sd[i] = sqrt(orc_calwgtcov(geno_m + (size_t)i * ld, geno_m + (size_t)i * ld, pop_off, P, pop_wgt));
        for (int i = 0; i < U; i++)
            sd[M + i] = sqrt(orc_calwgtcov(geno_u + (size_t)i * ld, geno_u + (size_t)i * ld, pop_off, P, pop_wgt));
    }
    for (int i = 0; i < M; i++) {                                           /* qcat.cpp:185-192 / qcatmix.cpp:207-217 */
        B11[(size_t)i * M + i] = 1.0 + lambda;
        for (int j = i + 1; j < M; j++) {
            double v;
            if (mode == 0) v = orc_calcor(geno_m + (size_t)i * ld, geno_m + (size_t)j * ld, pop_off, P);
            else v = orc_calwgtcov(geno_m + (size_t)i * ld, geno_m + (size_t)j * ld, pop_off, P, pop_wgt) / (sd[i] * sd[j]);
            B11[(size_t)j * M + i] = v;
            B11[(size_t)i * M + j] = v;
        }
    }
    if (b11_out) memcpy(b11_out, B11, sizeof(double) * MM);
    *out_num_eig = orc_count_pc(B11, M, eig_cutoff);                        /* qcat.cpp:203 */

    /* Eigen::LLT lower factor, column-major L(i,j) = L[j*M+i]              util.cpp:271-274 */
#define LL(i, j) L[(size_t)(j) * M + (i)]
    for (int j = 0; j < M; j++) {
        double s = B11[(size_t)j * M + j];
        for (int k = 0; k < j; k++) s -= LL(j, k) * LL(j, k);
        double dj = sqrt(s);
        LL(j, j) = dj;
        for (int i = j + 1; i < M; i++) {
            double t = B11[(size_t)j * M + i];
            for (int k = 0; k < j; k++) t -= LL(i, k) * LL(j, k);
            LL(i, j) = t / dj;
        }
    }
#undef LL
    orc_inv_mat(LInv, L, M);                                                /* qcat.cpp:207 */
    for (int i = 0; i < M; i++) {                                           /* qcat.cpp:208 */
        double s = 0.0;
        for (int k = 0; k < M; k++) s += LInv[(size_t)k * M + i] * z1[k];
        wz[i] = s;
    }
    for (int t = 0; t < n_predm + U; t++) {
        if (t < n_predm) {                                                  /* qcat.cpp:217-221 */
            for (int j = 0; j < M; j++) b[j] = B11[(size_t)j * M + (t + n_head)];
        } else {                                                            /* qcat.cpp:176-184, 235-239 */
            int i = t - n_predm;
            for (int j = 0; j < M; j++) {
                if (mode == 0) b[j] = orc_calcor(geno_u + (size_t)i * ld, geno_m + (size_t)j * ld, pop_off, P);
                else b[j] = orc_calwgtcov(geno_u + (size_t)i * ld, geno_m + (size_t)j * ld, pop_off, P, pop_wgt) / (sd[M + i] * sd[j]);
            }
            if (b21_out) memcpy(b21_out + (size_t)i * M, b, sizeof(double) * M);
        }
        for (int i = 0; i < M; i++) {
            double s = 0.0;
            for (int k = 0; k < M; k++) s += LInv[(size_t)k * M + i] * b[k];
            wb[i] = s;
        }
        out_r[t] = vec_cor(wz, wb, M);
    }
    free(B11); free(L); free(LInv); free(b); free(wz); free(wb); free(sd);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* a9. Gene::CalJepegPval / CalJepegmixPval tail      src/gene.cpp:317-550    */
/*  Inputs for one gene with n SNPs:                                          */
/*    CorG     n x n col-major LD among the gene SNPs, diag = 1+lambda        */
/*    z[n], info[n]                                                            */
/*    has[n*6] 1 if SNP s carries category c (Snp::categ_map_ membership),    */
/*    wgt[n*6] its weight (0 when absent, Snp::GetCategWgt snp.cpp)           */
/*  Outputs: chisq, df, jepeg_pval, top_categ (0..5, -1 = "."), top_categ_pval,*/
/*           top_snp (index), top_snp_pval.  Defaults follow Gene::Gene       */
/*           (gene.cpp:46-70): chisq=-1, df=0, pvals=-1.                      */
/* ------------------------------------------------------------------------- */
ORC_API int orc_jepeg_gene_tail(int n, const double* CorG, const double* z, const double* info,
                                const int* has, const double* wgt,
                                double min_abs_eig, double categ_cor_cutoff, int denorm_norm_w,
                                double* chisq, int* df_out, double* jepeg_pval,
                                int* top_categ, double* top_categ_pval,
                                int* top_snp, double* top_snp_pval)
{
    /* RunJepeg: category census                                   gene.cpp:97-160 */
    int count[6] = {0, 0, 0, 0, 0, 0};
    for (int s = 0; s < n; s++)
        for (int c = 0; c < 6; c++)
            if (has[s * 6 + c]) count[c]++;
    int cat_num[6], k = 0;
    for (int c = 0; c < 6; c++) if (count[c]) cat_num[k++] = c;     /* categ_vec_ gene.cpp:146-151 */
    double cat_pval[6]; int cat_rmv[6];
    for (int i = 0; i < k; i++) { cat_pval[i] = 0.0; cat_rmv[i] = 0; }

    *chisq = -1.0; *df_out = 0; *jepeg_pval = -1.0;
    *top_categ = -1; *top_categ_pval = -1.0; *top_snp = -1; *top_snp_pval = -1.0;
    if (n <= 0) return 0;

    double* W = (double*)calloc((size_t)(k ? k : 1) * n, sizeof(double));      /* k x n, W[i + k*s] col-major */
    double* WC = (double*)calloc((size_t)(k ? k : 1) * n, sizeof(double));
    double CovU[36], CorU[36], WWt[36], Uv[6];
    /* GetW                                                          gene.cpp:859-877 */
    for (int s = 0; s < n; s++) {
        int kk = 0;
        for (int c = 0; c < 6; c++)
            if (count[c] != 0) { W[kk + (size_t)k * s] = wgt[s * 6 + c] * sqrt(info[s]); kk++; }
    }
    for (int i = 0; i < k; i++)                                                  /* WWt gene.cpp:337 */
        for (int j = 0; j < k; j++) {
            double sum = 0.0;
            for (int s = 0; s < n; s++) sum += W[i + (size_t)k * s] * W[j + (size_t)k * s];
            WWt[i + 6 * j] = sum;
        }
    for (int i = 0; i < k; i++)                                                  /* W_CorG gene.cpp:343 */
        for (int s = 0; s < n; s++) {
            double sum = 0.0;
            for (int t = 0; t < n; t++) sum += W[i + (size_t)k * t] * CorG[(size_t)s * n + t];
            WC[i + (size_t)k * s] = sum;
        }
    for (int i = 0; i < k; i++)                                                  /* CovU gene.cpp:344 */
        for (int j = 0; j < k; j++) {
            double sum = 0.0;
            for (int s = 0; s < n; s++) sum += WC[i + (size_t)k * s] * W[j + (size_t)k * s];
            CovU[i + 6 * j] = sum;
        }
    for (int i = 0; i < k; i++)                                                  /* CnvrtCovToCor util.cpp:284-296 */
        for (int j = i; j < k; j++) {
            double std1 = sqrt(CovU[i + 6 * i]);
            double std2 = sqrt(CovU[j + 6 * j]);
            double cor = CovU[i + 6 * j] / (std1 * std2);
            CorU[i + 6 * j] = cor;
            if (i != j) CorU[j + 6 * i] = cor;
        }
    for (int i = 0; i < k; i++) {                                                /* U = W Z gene.cpp:365 */
        double sum = 0.0;
        for (int s = 0; s < n; s++) sum += W[i + (size_t)k * s] * z[s];
        Uv[i] = sum;
    }
    for (int i = 0; i < k; i++) {                                                /* gene.cpp:372-377 */
        double var = CovU[i + 6 * i];
        double u = Uv[i] / sqrt(var);
        cat_pval[i] = 2 * orc_pnorm_upper(fabs(u));
    }
    for (int j = k - 1; j > 0; j--)                                              /* gene.cpp:391-399 */
        for (int i = 0; i < j; i++) {
            double cor = CorU[i + 6 * j];
            if (fabs(cor) > categ_cor_cutoff) { cat_rmv[j] = 1; break; }
        }
    for (int i = 0; i < k; i++) {                                                /* gene.cpp:408-414 */
        double varU = CovU[i + 6 * i];
        double normW = WWt[i + 6 * i] / denorm_norm_w;
        if (varU < normW) cat_rmv[i] = 1;
    }
    int nrmv = 0;
    for (int i = 0; i < k; i++) if (cat_rmv[i]) nrmv++;
    int df = k - nrmv;                                                           /* gene.cpp:430 */
    *df_out = df;
    if (df) {                                                                    /* gene.cpp:443-549 */
        double X[6], CovX[36], CovXInv[36];
        int ii = 0;
        for (int i = 0; i < k; i++) if (!cat_rmv[i]) X[ii++] = Uv[i];
        int nn = 0;
        for (int i = 0; i < k; i++) {
            if (cat_rmv[i]) continue;
            int mm = 0;
            for (int j = 0; j < k; j++) {
                if (cat_rmv[j]) continue;
                CovX[nn + df * mm] = CovU[i + 6 * j];
                mm++;
            }
            nn++;
        }
        orc_make_pos_def(CovX, df, min_abs_eig);                                 /* gene.cpp:493 */
        orc_inv_mat(CovXInv, CovX, df);                                          /* gene.cpp:494 */
        double t[6];
        for (int c = 0; c < df; c++) {                                           /* Xt*CovXInv gene.cpp:501 */
            double sum = 0.0;
            for (int r = 0; r < df; r++) sum += X[r] * CovXInv[r + df * c];
            t[c] = sum;
        }
        double cs = 0.0;
        for (int c = 0; c < df; c++) cs += t[c] * X[c];                          /* gene.cpp:502 */
        *chisq = cs;
        *jepeg_pval = orc_pchisq_upper(cs, df);                                  /* gene.cpp:509 */
        /* GetTopCateg                                                gene.cpp:880-891 */
        int top_index = 0;
        for (int i = 0; i < k; i++) {
            double top_p = cat_pval[top_index];
            double p = cat_pval[i];
            if ((top_p > p) & !cat_rmv[i]) top_index = i;
        }
        *top_categ = cat_num[top_index];
        *top_categ_pval = cat_pval[top_index];
        /* GetTopSNP                                                  gene.cpp:894-904 */
        int ts = 0;
        for (int i = 0; i < n; i++)
            if (fabs(z[ts]) < fabs(z[i])) ts = i;
        *top_snp = ts;
        *top_snp_pval = 2 * orc_pnorm_upper(fabs(z[ts]));                        /* gene.cpp:522 */
    }
    free(W); free(WC);
    return 0;
}

/* Raw co-occurrence counts: sxy[i][j] = sum_n x_i[n]*x_j[n] over columns
 * [c0,c1) -- the integer the reference accumulates as `sumxy` (util.cpp:62,114).
 * Row-major S x S int64.  Used for the bit-exact integer parity tests. */
ORC_API void orc_gram_counts(const char* geno, long ld, int S, int c0, int c1, long long* out)
{
    for (int i = 0; i < S; i++)
        for (int j = i; j < S; j++) {
            const char* x = geno + (size_t)i * ld;
            const char* y = geno + (size_t)j * ld;
            long long s = 0;
            for (int n = c0; n < c1; n++) s += (long long)(x[n] - '0') * (y[n] - '0');
            out[(size_t)i * S + j] = s;
            out[(size_t)j * S + i] = s;
        }
}
