/*
 * totsu_oracle.c -- CPU f64 restatement of the Totsu conic solver hot path.
 * TEST INFRASTRUCTURE ONLY (see totsu_oracle.h).  Citations are relative to
 * /root/reference/solver_rust_conic/.
 *
 * Build: see oracle/Makefile (-ffp-contract=off so that a*b+c is two roundings, like the
 * Rust reference, which never contracts to FMA).
 */
#include "totsu_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* test / bench convenience (no counterpart in the reference): small problems run faster on a few threads than on
 * every core of a large host, where each parallel region costs more than the loop it splits */
void oc_set_num_threads(int k)
{
#ifdef _OPENMP
    if (k > 0) omp_set_num_threads(k);
#else
    (void)k;
#endif
}

int oc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ======================================================================================
 * LinAlg primitives -- FloatGeneric semantics (totsu_core/src/floatgeneric.rs:16-84).
 * Summation orders are the reference's sequential orders; the OpenMP variants below keep
 * the per-element order wherever a result element is a sequential sum (transform_ge).
 * ==================================================================================== */

/* floatgeneric.rs:21-28: sqrt of sequential sum of squares (NOT dnrm2's scaled form) */
double oc_norm(size_t n, const double *x)
{
    double sum = 0.0;
    for (size_t i = 0; i < n; ++i) sum = sum + x[i] * x[i];
    return sqrt(sum);
}

/* floatgeneric.rs:30-37 */
void oc_copy(size_t n, const double *x, double *y)
{
    for (size_t i = 0; i < n; ++i) y[i] = x[i];
}

/* floatgeneric.rs:39-44 (0*NaN stays NaN, like the reference) */
void oc_scale(size_t n, double alpha, double *x)
{
    for (size_t i = 0; i < n; ++i) x[i] = alpha * x[i];
}

/* floatgeneric.rs:46-53 */
void oc_add(size_t n, double alpha, const double *x, double *y)
{
    for (size_t i = 0; i < n; ++i) y[i] = y[i] + alpha * x[i];
}

/* floatgeneric.rs:55-60 */
void oc_adds(size_t n, double s, double *y)
{
    for (size_t i = 0; i < n; ++i) y[i] = y[i] + s;
}

/* floatgeneric.rs:62-74: sum |x[0]|, |x[incx]|, ... over chunks(incx) of a slice of length len;
 * incx == 0 -> 0.  (F64LAPACK: dasum with n = ceil(len/incx), f64lapack.rs:51-59 -- same set.) */
double oc_abssum(size_t len, const double *x, size_t incx)
{
    if (incx == 0) return 0.0;
    double sum = 0.0;
    for (size_t i = 0; i < len; i += incx) sum = sum + fabs(x[i]);
    return sum;
}

/* floatgeneric.rs:76-84: y = alpha*d*x + beta*y, evaluated as (alpha*d[i])*x[i] + beta*y[i] */
void oc_transform_di(size_t n, double alpha, const double *d, const double *x, double beta, double *y)
{
    for (size_t i = 0; i < n; ++i) y[i] = alpha * d[i] * x[i] + beta * y[i];
}

/* ======================================================================================
 * LinAlgEx
 * ==================================================================================== */

/* floatgeneric.rs:331-353 (MatIdx :97-105: element (r,c) at c*n_row + r, swapped when transposed).
 * y[r] = alpha * (sum_c mat(r,c) x[c], c ascending, start 0) + beta*y[r].
 * The loop nest is re-ordered per direction for memory locality, and rows / columns are split
 * over OpenMP threads, but every y element is still the same left-to-right sequential sum. */
void oc_transform_ge(int transpose, size_t n_row, size_t n_col, double alpha,
                     const double *mat, const double *x, double beta, double *y)
{
    if (transpose) {
        /* y has n_col entries, y[c] = sum_r mat[c*n_row + r] * x[r] */
        #pragma omp parallel for schedule(static) if (n_row * n_col > 65536)
        for (ptrdiff_t c = 0; c < (ptrdiff_t)n_col; ++c) {
            const double *col = mat + (size_t)c * n_row;
            double s = 0.0;
            for (size_t r = 0; r < n_row; ++r) s = s + col[r] * x[r];
            y[c] = alpha * s + beta * y[c];
        }
    } else {
        /* y has n_row entries; accumulate column by column into row blocks kept in cache */
        const size_t RB = 256;
        const ptrdiff_t nblk = (ptrdiff_t)((n_row + RB - 1) / RB);
        #pragma omp parallel for schedule(static) if (n_row * n_col > 65536)
        for (ptrdiff_t b = 0; b < nblk; ++b) {
            const size_t r0 = (size_t)b * RB;
            const size_t r1 = (r0 + RB < n_row) ? r0 + RB : n_row;
            double acc[256];
            for (size_t r = r0; r < r1; ++r) acc[r - r0] = 0.0;
            for (size_t c = 0; c < n_col; ++c) {
                const double xc = x[c];
                const double *col = mat + c * n_row;
                for (size_t r = r0; r < r1; ++r) acc[r - r0] = acc[r - r0] + col[r] * xc;
            }
            for (size_t r = r0; r < r1; ++r) y[r] = alpha * acc[r - r0] + beta * y[r];
        }
    }
}

/* packed-upper index, floatgeneric.rs:206-215 */
static inline size_t sp_idx(size_t r, size_t c)
{
    if (r > c) { size_t t = r; r = c; c = t; }
    return c * (c + 1) / 2 + r;
}

/* floatgeneric.rs:356-376 */
void oc_transform_sp(size_t n, double alpha, const double *mat, const double *x, double beta, double *y)
{
    for (size_t r = 0; r < n; ++r) {
        double s = 0.0;
        for (size_t c = 0; c < n; ++c) s = s + mat[sp_idx(r, c)] * x[c];
        y[r] = alpha * s + beta * y[r];
    }
}

/* floatgeneric.rs:378-384 */
size_t oc_map_eig_worklen(size_t n) { return n + n * n; }

/* cyclic Jacobi on a packed symmetric matrix, eigenvectors accumulated in z (col-major n x n).
 * floatgeneric.rs:273-324: same sweep order, same threshold test, same rotation formulas, same
 * in-place aliasing through the symmetric index (the k==i / k==j writes are overwritten by the
 * explicit diagonal / off-diagonal assignments at the end of the rotation). */
static void jacobi_eig(size_t n, double *xp, double *z, double eps)
{
    const double tol = eps * eps;
    int conv = 0;
    while (!conv) {
        conv = 1;
        for (size_t i = 0; i < n; ++i) {
            for (size_t j = i + 1; j < n; ++j) {
                const double a = xp[sp_idx(i, i)];
                const double b = xp[sp_idx(j, j)];
                const double d = xp[sp_idx(i, j)];
                if ((d * d > tol * a * b) && (d * d > tol)) {
                    conv = 0;
                    const double zeta = (b - a) / (2.0 * d);
                    double t;
                    if (zeta > 0.0) t = 1.0 / (zeta + sqrt(1.0 + zeta * zeta));
                    else            t = -1.0 / (-zeta + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t);
                    const double s = c * t;
                    for (size_t k = 0; k < n; ++k) {
                        const double xi = xp[sp_idx(k, i)];
                        const double xj = xp[sp_idx(k, j)];
                        xp[sp_idx(k, i)] = c * xi - s * xj;
                        xp[sp_idx(k, j)] = s * xi + c * xj;
                        const double zi = z[i * n + k];
                        const double zj = z[j * n + k];
                        z[i * n + k] = c * zi - s * zj;
                        z[j * n + k] = s * zi + c * zj;
                    }
                    xp[sp_idx(i, i)] = c * c * a + s * s * b - 2.0 * c * s * d;
                    xp[sp_idx(j, j)] = s * s * a + c * c * b + 2.0 * c * s * d;
                    xp[sp_idx(i, j)] = 0.0;
                }
            }
        }
    }
}

static size_t tri_order(size_t sn)
{
    /* floatgeneric.rs:389-391: n = (sqrt(8 sn + 1) - 1) / 2, checked */
    size_t n = ((size_t)sqrt((double)(8 * sn + 1)) - 1) / 2;
    assert(n * (n + 1) / 2 == sn);
    return n;
}

static int map_apply(int kind, double e, double *out)
{
    if (e > 0.0) { *out = (kind == 1) ? sqrt(e) : e; return 1; }
    return 0;
}

/* floatgeneric.rs:386-439 */
void oc_map_eig(size_t sn, double *mat, int has_scale, double scale_diag, double eps_zero,
                double *work, int map_kind)
{
    const size_t n = tri_order(sn);
    double *w = work;
    double *z = work + n;

    if (has_scale)
        for (size_t i = 0; i < n; ++i) mat[sp_idx(i, i)] = mat[sp_idx(i, i)] * scale_diag;

    for (size_t i = 0; i < n * n; ++i) z[i] = 0.0;
    for (size_t i = 0; i < n; ++i) z[i * n + i] = 1.0;

    jacobi_eig(n, mat, z, eps_zero);

    for (size_t i = 0; i < n; ++i) w[i] = mat[sp_idx(i, i)];

    for (size_t i = 0; i < sn; ++i) mat[i] = 0.0;
    for (size_t i = 0; i < n; ++i) {
        double e;
        if (map_apply(map_kind, w[i], &e)) {
            const double *zc = z + i * n;
            /* rank1op, floatgeneric.rs:240-249: self[(r,c)] = alpha*x[r]*x[c] + self[(r,c)] */
            for (size_t c = 0; c < n; ++c)
                for (size_t r = 0; r <= c; ++r)
                    mat[sp_idx(r, c)] = e * zc[r] * zc[c] + mat[sp_idx(r, c)];
        }
    }

    if (has_scale) {
        const double inv = 1.0 / scale_diag;
        for (size_t i = 0; i < n; ++i) mat[sp_idx(i, i)] = mat[sp_idx(i, i)] * inv;
    }
}

/* f64lapack.rs:195-224: packed upper (by columns) -> full col-major upper triangle, diag *= scale.
 * The strictly lower part of m is left untouched (LAPACK 'U' never reads it). */
void oc_vec_to_mat(size_t n, const double *v, double *m, int has_scale, double scale)
{
    size_t off = 0;
    for (size_t c = 0; c < n; ++c) {
        for (size_t r = 0; r <= c; ++r) m[c * n + r] = v[off + r];
        off += c + 1;
    }
    if (has_scale)
        for (size_t i = 0; i < n; ++i) m[i * (n + 1)] = scale * m[i * (n + 1)];
}

/* f64lapack.rs:226-255: diag *= 1/scale, then copy the upper triangle back to packed */
void oc_mat_to_vec(size_t n, double *m, double *v, int has_scale, double scale)
{
    if (has_scale) {
        const double inv = 1.0 / scale;
        for (size_t i = 0; i < n; ++i) m[i * (n + 1)] = inv * m[i * (n + 1)];
    }
    size_t off = 0;
    for (size_t c = 0; c < n; ++c) {
        for (size_t r = 0; r <= c; ++r) v[off + r] = m[c * n + r];
        off += c + 1;
    }
}

/* ---- symmetric eigensolver: Householder tridiagonalisation + implicit-shift QL ----------
 * Stand-in for LAPACK dsyevr as called at f64lapack.rs:86-91 (jobz V, range V (0,inf], uplo U).
 * a: full col-major n x n, only the upper triangle is read; on return a holds the eigenvectors
 * in columns, d the eigenvalues (ascending not guaranteed before the sort), e is scratch. */
static void sym_tridiag(size_t n, double *a, double *d, double *e)
{
    /* symmetrise: work on the lower triangle of a copy-in-place (a[j*n+i], i>=j mirrors upper) */
    for (size_t c = 0; c < n; ++c)
        for (size_t r = c + 1; r < n; ++r) a[c * n + r] = a[r * n + c];
    /* classical Householder reduction, accumulating the orthogonal transform in a.
       Indexing A(i,j) = a[j*n+i]; since the matrix is symmetric we are free to use rows. */
#define A_(i, j) a[(size_t)(j) * n + (size_t)(i)]
    for (size_t ii = n; ii-- > 1;) {
        const size_t i = ii, l = i - 1;
        double h = 0.0, scale = 0.0;
        if (l > 0) {
            for (size_t k = 0; k <= l; ++k) scale += fabs(A_(i, k));
            if (scale == 0.0) {
                e[i] = A_(i, l);
            } else {
                for (size_t k = 0; k <= l; ++k) { A_(i, k) /= scale; h += A_(i, k) * A_(i, k); }
                double f = A_(i, l);
                double g = (f >= 0.0) ? -sqrt(h) : sqrt(h);
                e[i] = scale * g;
                h -= f * g;
                A_(i, l) = f - g;
                f = 0.0;
                for (size_t j = 0; j <= l; ++j) {
                    A_(j, i) = A_(i, j) / h;
                    g = 0.0;
                    for (size_t k = 0; k <= j; ++k) g += A_(j, k) * A_(i, k);
                    for (size_t k = j + 1; k <= l; ++k) g += A_(k, j) * A_(i, k);
                    e[j] = g / h;
                    f += e[j] * A_(i, j);
                }
                const double hh = f / (h + h);
                for (size_t j = 0; j <= l; ++j) {
                    f = A_(i, j);
                    e[j] = g = e[j] - hh * f;
                    for (size_t k = 0; k <= j; ++k) A_(j, k) -= (f * e[k] + g * A_(i, k));
                }
            }
        } else {
            e[i] = A_(i, l);
        }
        d[i] = h;
    }
    d[0] = 0.0;
    e[0] = 0.0;
    for (size_t i = 0; i < n; ++i) {
        if (d[i] != 0.0) {
            for (size_t j = 0; j < i; ++j) {
                double g = 0.0;
                for (size_t k = 0; k < i; ++k) g += A_(i, k) * A_(k, j);
                for (size_t k = 0; k < i; ++k) A_(k, j) -= g * A_(k, i);
            }
        }
        d[i] = A_(i, i);
        A_(i, i) = 1.0;
        for (size_t j = 0; j < i; ++j) A_(j, i) = A_(i, j) = 0.0;
    }
}

static int tridiag_ql(size_t n, double *d, double *e, double *a)
{
    for (size_t i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    for (size_t l = 0; l < n; ++l) {
        int iter = 0;
        size_t m;
        do {
            for (m = l; m + 1 < n; ++m) {
                const double dd = fabs(d[m]) + fabs(d[m + 1]);
                if (fabs(e[m]) <= 2.220446049250313e-16 * dd) break;
            }
            if (m != l) {
                if (iter++ == 200) return -1;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
                double s = 1.0, c = 1.0, p = 0.0;
                size_t i;
                int under = 0;
                for (i = m; i-- > l;) {
                    double f = s * e[i];
                    const double b = c * e[i];
                    e[i + 1] = (r = hypot(f, g));
                    if (r == 0.0) {
                        d[i + 1] -= p;
                        e[m] = 0.0;
                        under = 1;
                        break;
                    }
                    s = f / r;
                    c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    d[i + 1] = g + (p = s * r);
                    g = c * r - b;
                    for (size_t k = 0; k < n; ++k) {
                        f = A_(k, i + 1);
                        A_(k, i + 1) = s * A_(k, i) + c * f;
                        A_(k, i) = c * A_(k, i) - s * f;
                    }
                }
                if (under) continue;
                d[l] -= p;
                e[l] = g;
                e[m] = 0.0;
            }
        } while (m != l);
    }
    return 0;
#undef A_
}

size_t oc_map_eig_worklen_ql(size_t n) { return n * n + n + n * n; }   /* f64lapack.rs:110-116,165-170 */

/* f64lapack.rs:172-190 (map_eig) + :78-108 (eig_func): vec_to_mat, eigen-decompose, a <- 0,
 * one upper rank-1 update e*z*z^T per kept eigenpair in ascending-eigenvalue order (dsyevr
 * returns the selected eigenvalues ascending), mat_to_vec.  Only eigenvalues in (0, inf] reach
 * the map closure (range 'V', vl = 0), exactly like dsyevr's half-open interval. */
void oc_map_eig_ql(size_t sn, double *mat, int has_scale, double scale_diag, double eps_zero,
                   double *work, int map_kind)
{
    (void)eps_zero;   /* dsyevr's abstol; the QL iteration runs to machine precision */
    const size_t n = tri_order(sn);
    double *a = work;
    double *w = work + n * n;
    double *z = w + n;

    oc_vec_to_mat(n, mat, a, has_scale, scale_diag);

    /* eigen-decomposition of the upper triangle of a; vectors -> z */
    for (size_t c = 0; c < n; ++c)
        for (size_t r = 0; r <= c; ++r) z[c * n + r] = a[c * n + r];
    double *e = (double *)malloc(sizeof(double) * (n ? n : 1));
    if (n > 0) {
        sym_tridiag(n, z, w, e);
        int rc = tridiag_ql(n, w, e, z);
        assert(rc == 0);
        (void)rc;
    }
    free(e);
    /* ascending order of eigenvalues (selection sort on indices; n is small) */
    size_t *ord = (size_t *)malloc(sizeof(size_t) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) ord[i] = i;
    for (size_t i = 0; i + 1 < n; ++i) {
        size_t k = i;
        for (size_t j = i + 1; j < n; ++j) if (w[ord[j]] < w[ord[k]]) k = j;
        size_t t = ord[i]; ord[i] = ord[k]; ord[k] = t;
    }

    for (size_t i = 0; i < n * n; ++i) a[i] = 0.0;      /* F64LAPACK::scale(0., a) */
    for (size_t q = 0; q < n; ++q) {
        const size_t i = ord[q];
        if (!(w[i] > 0.0)) continue;                    /* outside (vl, vu] */
        double ev;
        if (!map_apply(map_kind, w[i], &ev)) continue;
        const double *zc = z + i * n;
        /* dsyr upper: a(r,c) += ev * z[r] * z[c], r <= c; per reference BLAS: temp = alpha*x[c] */
        #pragma omp parallel for schedule(static) if (n > 128)
        for (ptrdiff_t c = 0; c < (ptrdiff_t)n; ++c) {
            const double temp = ev * zc[c];
            double *ac = a + (size_t)c * n;
            for (size_t r = 0; r <= (size_t)c; ++r) ac[r] = ac[r] + zc[r] * temp;
        }
    }
    free(ord);

    oc_mat_to_vec(n, a, mat, has_scale, scale_diag);
}

/* ======================================================================================
 * Cones
 * ==================================================================================== */

/* cone_zero.rs:38-44 */
void oc_proj_zero(int dual_cone, size_t n, double *x)
{
    if (!dual_cone) oc_scale(n, 0.0, x);
}

/* cone_rpos.rs:38-45: e = e.max(0)  (Rust f64::max: NaN.max(0) = 0) */
void oc_proj_rpos(size_t n, double *x)
{
    for (size_t i = 0; i < n; ++i) x[i] = fmax(x[i], 0.0);
}

/* cone_soc.rs:38-65 */
void oc_proj_soc(size_t n, double *x)
{
    if (n == 0) return;
    double *s = x;
    double *v = x + 1;
    const size_t nv = n - 1;
    const double val_s = s[0];
    const double norm_v = oc_norm(nv, v);
    if (norm_v <= -val_s) {
        oc_scale(nv, 0.0, v);
        s[0] = 0.0;
    } else if (norm_v <= val_s) {
        /* as they are */
    } else {
        const double alpha = (1.0 + val_s / norm_v) / 2.0;
        oc_scale(nv, alpha, v);
        s[0] = (norm_v + val_s) / 2.0;
    }
}

/* cone_rotsoc.rs:38-65 */
void oc_proj_rotsoc(size_t n, double *x)
{
    if (n == 0) return;
    if (n == 1) { x[0] = fmax(x[0], 0.0); return; }
    const double fsqrt2 = sqrt(2.0);
    double r = x[0], s = x[1];
    x[0] = (r + s) / fsqrt2;
    x[1] = (r - s) / fsqrt2;
    oc_proj_soc(n, x);
    r = x[0]; s = x[1];
    x[0] = (r + s) / fsqrt2;
    x[1] = (r - s) / fsqrt2;
}

/* cone_psd.rs:56-79 (+ query_worklen :32-38) */
int oc_proj_psd(size_t sn, double *x, double eps_zero, double *work, size_t worklen, int use_ql)
{
    const size_t n = tri_order(sn);
    const size_t need = use_ql ? oc_map_eig_worklen_ql(n) : oc_map_eig_worklen(n);
    if (worklen < need) return -1;
    const double fsqrt2 = sqrt(1.0 + 1.0);
    if (use_ql) oc_map_eig_ql(sn, x, 1, fsqrt2, eps_zero, work, 0);
    else        oc_map_eig(sn, x, 1, fsqrt2, eps_zero, work, 0);
    return 0;
}

/* ======================================================================================
 * MatOp (totsu_core/src/matop.rs)
 * ==================================================================================== */

/* matop.rs:76-96 */
void oc_matop_op(const oc_matop *m, int transpose, double alpha, const double *x, double beta, double *y)
{
    if (m->typ == OC_MAT_GENERAL) {
        if (m->nr > 0 && m->nc > 0) {
            oc_transform_ge(transpose, m->nr, m->nc, alpha, m->array, x, beta, y);
        } else {
            /* L::scale(beta, y) over y's own length */
            oc_scale(transpose ? m->nc : m->nr, beta, y);
        }
    } else {
        if (m->nr > 0) oc_transform_sp(m->nr, alpha, m->array, x, beta, y);
        else           oc_scale(0, beta, y);
    }
}

/* matop.rs:98-138 */
void oc_matop_absadd(const oc_matop *m, int colwise, double *y)
{
    if (m->typ == OC_MAT_GENERAL) {
        const size_t nr = m->nr, nc = m->nc;
        if (colwise) {
            #pragma omp parallel for schedule(static) if (nr * nc > 65536)
            for (ptrdiff_t i = 0; i < (ptrdiff_t)nc; ++i)
                y[i] = oc_abssum(nr, m->array + (size_t)i * nr, 1) + y[i];
        } else {
            #pragma omp parallel for schedule(static) if (nr * nc > 65536)
            for (ptrdiff_t i = 0; i < (ptrdiff_t)nr; ++i)
                y[i] = oc_abssum(nr * nc > (size_t)i ? nr * nc - (size_t)i : 0, m->array + i, nr) + y[i];
        }
    } else {
        const size_t n = m->nr;
        size_t sum = 0;
        for (size_t c = 0; c < n; ++c) {
            const double *col = m->array + sum;
            sum += c + 1;
            y[c] = oc_abssum(c + 1, col, 1) + y[c];
            for (size_t i = 0; i < c; ++i) y[i] = y[i] + fabs(col[i]);
        }
    }
}

static void matop_size(void *ctx, size_t *nr, size_t *nc) { oc_matop *m = ctx; *nr = m->nr; *nc = m->nc; }
static void matop_op_(void *ctx, double a, const double *x, double b, double *y) { oc_matop_op(ctx, 0, a, x, b, y); }
static void matop_top_(void *ctx, double a, const double *x, double b, double *y) { oc_matop_op(ctx, 1, a, x, b, y); }
static void matop_ac_(void *ctx, double *t) { oc_matop_absadd(ctx, 1, t); }
static void matop_ar_(void *ctx, double *s) { oc_matop_absadd(ctx, 0, s); }

oc_operator oc_matop_as_operator(oc_matop *m)
{
    oc_operator o = { m, matop_size, matop_op_, matop_top_, matop_ac_, matop_ar_ };
    return o;
}

/* ======================================================================================
 * Solver (totsu_core/src/solver/solver.rs)
 * ==================================================================================== */

/* solver.rs:27-41 */
void oc_param_default(oc_param *p)
{
    p->max_iter = -1;
    p->eps_acc = pow(10.0, -6);
    p->eps_inf = pow(10.0, -6);
    p->eps_zero = pow(10.0, -12);
    p->log_period = 10000;
}

/* solver.rs:231-249 */
size_t oc_query_worklen(size_t m, size_t n)
{
    return (n + m + m + 1) + (n + m + 1) + (n + m + m + 1) + (n + m + 1) + (n + m + m + 1) + (n + m + m + 1);
}

typedef struct {
    oc_operator *c, *a, *b;
    size_t m, n;
} sde;

/* solver.rs:85-107 */
static double fr_norm(oc_operator *op, double *work_v, double *work_t)
{
    size_t nr, nc;
    op->size(op->ctx, &nr, &nc);
    oc_scale(nc, 0.0, work_v);
    double sq_norm = 0.0;
    for (size_t row = 0; row < nc; ++row) {
        work_v[row] = 1.0;
        op->op(op->ctx, 1.0, work_v, 0.0, work_t);
        const double nn = oc_norm(nr, work_t);
        sq_norm = sq_norm + nn * nn;
        work_v[row] = 0.0;
    }
    return sqrt(sq_norm);
}

/* solver.rs:109-131 */
static void sde_op(const sde *k, double alpha, const double *x, double beta, double *y)
{
    const size_t m = k->m, n = k->n;
    const double *x_x = x, *x_y = x + n, *x_s = x + n + m, *x_tau = x + n + m + m;
    double *y_n = y, *y_m = y + n, *y_1 = y + n + m;

    k->a->trans_op(k->a->ctx, alpha, x_y, beta, y_n);
    k->c->op(k->c->ctx, alpha, x_tau, 1.0, y_n);

    k->a->op(k->a->ctx, -alpha, x_x, beta, y_m);
    oc_add(m, -alpha, x_s, y_m);
    k->b->op(k->b->ctx, alpha, x_tau, 1.0, y_m);

    k->c->trans_op(k->c->ctx, -alpha, x_x, beta, y_1);
    k->b->trans_op(k->b->ctx, -alpha, x_y, 1.0, y_1);
}

/* solver.rs:133-157 */
static void sde_trans_op(const sde *k, double alpha, const double *x, double beta, double *y)
{
    const size_t m = k->m, n = k->n;
    const double *x_n = x, *x_m = x + n, *x_1 = x + n + m;
    double *y_x = y, *y_y = y + n, *y_s = y + n + m, *y_tau = y + n + m + m;

    k->a->trans_op(k->a->ctx, -alpha, x_m, beta, y_x);
    k->c->op(k->c->ctx, -alpha, x_1, 1.0, y_x);

    k->a->op(k->a->ctx, alpha, x_n, beta, y_y);
    k->b->op(k->b->ctx, -alpha, x_1, 1.0, y_y);

    oc_scale(m, beta, y_s);
    oc_add(m, -alpha, x_m, y_s);

    k->c->trans_op(k->c->ctx, alpha, x_n, beta, y_tau);
    k->b->trans_op(k->b->ctx, alpha, x_m, 1.0, y_tau);
}

/* solver.rs:159-183 */
static void sde_abssum(const sde *k, double *tau, double *sigma)
{
    const size_t m = k->m, n = k->n;
    oc_scale(n + m + m + 1, 0.0, tau);
    double *tau_x = tau, *tau_y = tau + n, *tau_s = tau + n + m, *tau_tau = tau + n + m + m;

    k->a->absadd_cols(k->a->ctx, tau_x);
    k->c->absadd_rows(k->c->ctx, tau_x);
    k->a->absadd_rows(k->a->ctx, tau_y);
    k->b->absadd_rows(k->b->ctx, tau_y);
    oc_adds(m, 1.0, tau_s);
    k->c->absadd_cols(k->c->ctx, tau_tau);
    k->b->absadd_cols(k->b->ctx, tau_tau);

    double *sigma_n = sigma, *sigma_m = sigma + n, *sigma_1 = sigma + n + m;
    oc_copy(n, tau_x, sigma_n);
    oc_copy(m, tau_y, sigma_m);
    oc_add(m, 1.0, tau_s, sigma_m);
    oc_copy(1, tau_tau, sigma_1);
}

/* solver.rs:509-520, the `group` closure */
static void group_min(double *tau_group, size_t len)
{
    if (len > 0) {
        double min_t = tau_group[0];
        for (size_t i = 0; i < len; ++i) min_t = fmin(min_t, tau_group[i]);
        for (size_t i = 0; i < len; ++i) tau_group[i] = min_t;
    }
}

typedef struct {
    const oc_param *par;
    sde k;
    oc_cone *cone;
} core_t;

/* solver.rs:496-524 */
static void calc_precond(core_t *s, double *dp_tau, double *dp_sigma)
{
    const size_t m = s->k.m, n = s->k.n;
    sde_abssum(&s->k, dp_tau, dp_sigma);
    for (size_t i = 0; i < n + m + m + 1; ++i) dp_tau[i] = 1.0 / fmax(dp_tau[i], s->par->eps_zero);
    for (size_t i = 0; i < n + m + 1; ++i)     dp_sigma[i] = 1.0 / fmax(dp_sigma[i], s->par->eps_zero);
    s->cone->product_group(s->cone->ctx, dp_tau + n, m, group_min);
    s->cone->product_group(s->cone->ctx, dp_tau + n + m, m, group_min);
}

/* solver.rs:526-571; returns 0 ok / -1 cone failure */
static int update_vecs(core_t *s, double *x, double *y, const double *dp_tau, const double *dp_sigma,
                       double *tmpw, double *val_tau_out)
{
    const size_t m = s->k.m, n = s->k.n;
    const size_t N = n + m + m + 1, M = n + m + 1;
    double *rx = tmpw, *tx = tmpw + N;

    oc_copy(N, x, rx);

    sde_trans_op(&s->k, -1.0, y, 0.0, tx);
    oc_transform_di(N, 1.0, dp_tau, tx, 1.0, x);

    if (s->cone->proj(s->cone->ctx, 1, x + n, m) != 0) return -1;
    if (s->cone->proj(s->cone->ctx, 0, x + n + m, m) != 0) return -1;
    const double val_tau = fmax(x[n + m + m], 0.0);
    x[n + m + m] = val_tau;

    oc_add(N, -1.0 - 1.0, x, rx);

    double *ty = tx;
    sde_op(&s->k, -1.0, rx, 0.0, ty);
    oc_transform_di(M, 1.0, dp_sigma, ty, 1.0, y);

    const double kappa = fmin(y[n + m], 0.0);
    y[n + m] = kappa;

    *val_tau_out = val_tau;
    return 0;
}

/* solver.rs:573-612 */
static void criteria_conv(core_t *s, const double *x, double norm_c, double norm_b, double *tmpw,
                          double *cri_pri, double *cri_dual, double *cri_gap)
{
    const size_t m = s->k.m, n = s->k.n;
    const double *x_x = x, *x_y = x + n, *x_s = x + n + m;
    double *p = tmpw, *d = tmpw + m;
    const double val_tau = x[n + m + m];
    assert(val_tau > 0.0);
    double work_one[1] = { 1.0 };
    const double rtau = 1.0 / val_tau;

    oc_copy(m, x_s, p);
    s->k.b->op(s->k.b->ctx, -1.0, work_one, rtau, p);
    s->k.a->op(s->k.a->ctx, rtau, x_x, 1.0, p);

    s->k.c->op(s->k.c->ctx, 1.0, work_one, 0.0, d);
    s->k.a->trans_op(s->k.a->ctx, rtau, x_y, 1.0, d);

    s->k.c->trans_op(s->k.c->ctx, rtau, x_x, 0.0, work_one);
    const double g_x = work_one[0];
    s->k.b->trans_op(s->k.b->ctx, rtau, x_y, 0.0, work_one);
    const double g_y = work_one[0];
    const double g = g_x + g_y;

    *cri_pri = oc_norm(m, p) / (1.0 + norm_b);
    *cri_dual = oc_norm(n, d) / (1.0 + norm_c);
    *cri_gap = fabs(g) / (1.0 + fabs(g_x) + fabs(g_y));
}

/* solver.rs:614-656 */
static void criteria_inf(core_t *s, const double *x, double norm_c, double norm_b, double *tmpw,
                         double *cri_unbdd, double *cri_infeas)
{
    const size_t m = s->k.m, n = s->k.n;
    const double *x_x = x, *x_y = x + n, *x_s = x + n + m;
    double *p = tmpw, *d = tmpw + m;
    double work_one[1] = { 0.0 };

    oc_copy(m, x_s, p);
    s->k.a->op(s->k.a->ctx, 1.0, x_x, 1.0, p);
    s->k.a->trans_op(s->k.a->ctx, 1.0, x_y, 0.0, d);

    s->k.c->trans_op(s->k.c->ctx, -1.0, x_x, 0.0, work_one);
    const double m_cx = work_one[0];
    s->k.b->trans_op(s->k.b->ctx, -1.0, x_y, 0.0, work_one);
    const double m_by = work_one[0];

    *cri_unbdd = (m_cx > s->par->eps_zero) ? oc_norm(m, p) * norm_c / m_cx : INFINITY;
    *cri_infeas = (m_by > s->par->eps_zero) ? oc_norm(n, d) * norm_b / m_by : INFINITY;
}

static void trace_push(oc_trace *t, int64_t iter, int kind, double v0, double v1, double v2)
{
    if (!t) return;
    if (t->rec && t->len < t->cap) {
        oc_trace_rec *r = &t->rec[t->len];
        r->iter = iter; r->kind = kind; r->v0 = v0; r->v1 = v1; r->v2 = v2;
    }
    t->len++;
}

/* solver.rs:340-458 */
static int core_solve(core_t *s, double *work, oc_trace *trace)
{
    const size_t m = s->k.m, n = s->k.n;
    const size_t N = n + m + m + 1, M = n + m + 1;

    /* calc_norms, solver.rs:460-481: both use the head of work as scratch */
    double work1[1] = { 0.0 };
    const double norm_b = fr_norm(s->k.b, work1, work);
    const double norm_c = fr_norm(s->k.c, work1, work);
    if (trace) { trace->norm_b = norm_b; trace->norm_c = norm_c; trace->len = 0; trace->iters = -1; }

    double *x = work, *y = x + N, *dp_tau = y + M, *dp_sigma = dp_tau + N, *tmpw = dp_sigma + M;

    /* init_vecs, solver.rs:483-494 */
    oc_scale(N, 0.0, x);
    oc_scale(M, 0.0, y);
    x[n + m + m] = 1.0;

    calc_precond(s, dp_tau, dp_sigma);
    if (trace && trace->precond_out) memcpy(trace->precond_out, dp_tau, (N + M) * sizeof(double));

    int64_t i = 0;
    for (;;) {
        const int excess_iter = (s->par->max_iter >= 0) ? (i + 1 >= s->par->max_iter) : 0;

        double val_tau;
        if (update_vecs(s, x, y, dp_tau, dp_sigma, tmpw, &val_tau) != 0) return OC_CONE_FAILURE;

        if (trace && trace->snap_out) {
            for (size_t q = 0; q < trace->n_snap; ++q)
                if (trace->snap_iters[q] == i) {
                    memcpy(trace->snap_out + q * (N + M), x, sizeof(double) * N);
                    memcpy(trace->snap_out + q * (N + M) + N, y, sizeof(double) * M);
                }
        }
        if (trace) trace->iters = i;

        if (val_tau > s->par->eps_zero) {
            double cri_pri, cri_dual, cri_gap;
            criteria_conv(s, x, norm_c, norm_b, tmpw, &cri_pri, &cri_dual, &cri_gap);
            trace_push(trace, i, 0, cri_pri, cri_dual, cri_gap);
            const int term_conv = (cri_pri <= s->par->eps_acc) && (cri_dual <= s->par->eps_acc)
                                  && (cri_gap <= s->par->eps_acc);
            if (excess_iter || term_conv) {
                oc_scale(n, 1.0 / val_tau, x);
                oc_scale(m, 1.0 / val_tau, x + n);
                return term_conv ? OC_OK : OC_EXCESS_ITER;
            }
        } else {
            double cri_unbdd, cri_infeas;
            criteria_inf(s, x, norm_c, norm_b, tmpw, &cri_unbdd, &cri_infeas);
            trace_push(trace, i, 1, cri_unbdd, cri_infeas, 0.0);
            const int term_unbdd = cri_unbdd <= s->par->eps_inf;
            const int term_infeas = cri_infeas <= s->par->eps_inf;
            if (excess_iter || term_unbdd || term_infeas) {
                if (term_unbdd) return OC_UNBOUNDED;
                if (term_infeas) return OC_INFEASIBLE;
                return OC_EXCESS_ITER;
            }
        }
        i += 1;
    }
}

/* solver.rs:285-321 */
int oc_solve(const oc_param *par, oc_operator *op_c, oc_operator *op_a, oc_operator *op_b,
             oc_cone *cone, double *work, size_t worklen, oc_trace *trace)
{
    size_t m, n, r, c;
    op_a->size(op_a->ctx, &m, &n);
    op_c->size(op_c->ctx, &r, &c);
    if (r != n || c != 1) return OC_INVALID_OP;
    op_b->size(op_b->ctx, &r, &c);
    if (r != m || c != 1) return OC_INVALID_OP;
    if (oc_query_worklen(m, n) > worklen) return OC_WORK_SHORTAGE;

    core_t s;
    s.par = par;
    s.k.c = op_c; s.k.a = op_a; s.k.b = op_b; s.k.m = m; s.k.n = n;
    s.cone = cone;
    return core_solve(&s, work, trace);
}

/* ======================================================================================
 * Product cone over consecutive segments -- the shape of ProbLPCone (lp.rs:198-217),
 * ProbSOCPCone (socp.rs:294-331), ProbSDPCone (sdp.rs:196-217).
 * ==================================================================================== */
typedef struct {
    size_t n_seg;
    const int32_t *type;
    const int64_t *len;
    double eps_zero;
    int use_ql;
    double *psd_work; size_t psd_worklen;
} seg_cone;

static int seg_proj(void *ctx, int dual_cone, double *x, size_t len)
{
    seg_cone *sc = ctx;
    size_t done = 0;
    for (size_t i = 0; i < sc->n_seg; ++i) {
        const size_t l = (size_t)sc->len[i];
        double *xs = x + done;
        switch (sc->type[i]) {
        case OC_CONE_ZERO:   oc_proj_zero(dual_cone, l, xs); break;
        case OC_CONE_RPOS:   oc_proj_rpos(l, xs); break;
        case OC_CONE_SOC:    oc_proj_soc(l, xs); break;
        case OC_CONE_ROTSOC: oc_proj_rotsoc(l, xs); break;
        case OC_CONE_PSD:
            if (oc_proj_psd(l, xs, sc->eps_zero, sc->psd_work, sc->psd_worklen, sc->use_ql) != 0) return -1;
            break;
        default: return -1;
        }
        done += l;
    }
    assert(done == len);
    (void)len;
    return 0;
}

static void seg_group(void *ctx, double *dp_tau, size_t len, oc_group_fn group)
{
    seg_cone *sc = ctx;
    size_t done = 0;
    for (size_t i = 0; i < sc->n_seg; ++i) {
        const size_t l = (size_t)sc->len[i];
        /* zero / rpos: do nothing (cone_zero.rs:46-49, cone_rpos.rs:47-50); others: group(dp_tau) */
        if (sc->type[i] != OC_CONE_ZERO && sc->type[i] != OC_CONE_RPOS) group(dp_tau + done, l);
        done += l;
    }
    (void)len;
}

static void seg_cone_init(seg_cone *sc, size_t n_seg, const int32_t *type, const int64_t *len,
                          double eps_zero, int use_ql)
{
    sc->n_seg = n_seg; sc->type = type; sc->len = len; sc->eps_zero = eps_zero; sc->use_ql = use_ql;
    size_t need = 0;
    for (size_t i = 0; i < n_seg; ++i)
        if (type[i] == OC_CONE_PSD) {
            const size_t k = tri_order((size_t)len[i]);
            const size_t w = use_ql ? oc_map_eig_worklen_ql(k) : oc_map_eig_worklen(k);
            if (w > need) need = w;
        }
    sc->psd_worklen = need;
    sc->psd_work = need ? (double *)calloc(need, sizeof(double)) : NULL;
}

static int run_and_extract(const oc_param *par, oc_operator *oc, oc_operator *oa, oc_operator *ob,
                           oc_cone *cone, size_t n, size_t m, double *out_x, double *out_y, oc_trace *trace)
{
    const size_t wl = oc_query_worklen(m, n);
    double *work = (double *)calloc(wl ? wl : 1, sizeof(double));
    const int rc = oc_solve(par, oc, oa, ob, cone, work, wl, trace);
    if (out_x) memcpy(out_x, work, sizeof(double) * n);          /* solver.rs:317-318 */
    if (out_y) memcpy(out_y, work + n, sizeof(double) * m);
    free(work);
    return rc;
}

/* plain MatOp operators + a product cone: the shape of totsu_core/tests/solver.rs:14-53 and
 * examples/nostd_cortex-m/src/main.rs:57-99 */
int oc_solve_matop_cones(const oc_param *par, size_t n, size_t m,
                         const double *vec_c, const double *mat_a, const double *vec_b,
                         size_t n_seg, const int32_t *seg_type, const int64_t *seg_len,
                         int use_ql, double *out_x, double *out_y, oc_trace *trace)
{
    oc_matop mc = { OC_MAT_GENERAL, n, 1, vec_c };
    oc_matop ma = { OC_MAT_GENERAL, m, n, mat_a };
    oc_matop mb = { OC_MAT_GENERAL, m, 1, vec_b };
    oc_operator oc = oc_matop_as_operator(&mc), oa = oc_matop_as_operator(&ma), ob = oc_matop_as_operator(&mb);
    seg_cone sc;
    seg_cone_init(&sc, n_seg, seg_type, seg_len, par->eps_zero, use_ql);
    oc_cone cone = { &sc, seg_proj, seg_group };
    const int rc = run_and_extract(par, &oc, &oa, &ob, &cone, n, m, out_x, out_y, trace);
    free(sc.psd_work);
    return rc;
}

/* ======================================================================================
 * A sparse user-defined Operator (the trait: totsu_core/src/solver/operator.rs:11-156; the pattern of a caller's own
 * operator: examples/imgnr_udef/src/prob_op_a.rs:33-120).  The matrix comes in compressed-sparse-column form; op walks a
 * row-major mirror built here, trans_op the columns -- both are gathers, so both parallelise over their outputs.  This is the
 * CPU baseline of bench.py's sparse workloads and the oracle of their parity tests; tests/test_oracle_golden.py pins it
 * to the dense MatOp path (same iterates on the dense-ified matrix).
 * ==================================================================================== */
typedef struct {
    size_t m, n;
    const int64_t *cp; const int32_t *ri; const double *cv;      /* by columns (the caller's) */
    int64_t *rp; int32_t *ci; double *rv;                        /* by rows (built here) */
} csc_op;

static void csc_size(void *ctx, size_t *nr, size_t *nc) { csc_op *o = ctx; *nr = o->m; *nc = o->n; }
/* y = alpha A x + beta y (operator.rs:40-57) */
static void csc_opf(void *ctx, double alpha, const double *x, double beta, double *y)
{
    csc_op *o = ctx;
    #pragma omp parallel for schedule(dynamic, 256)
    for (size_t r = 0; r < o->m; ++r) {
        double s = 0.0;
        for (int64_t k = o->rp[r]; k < o->rp[r + 1]; ++k) s += o->rv[k] * x[o->ci[k]];
        y[r] = alpha * s + beta * y[r];
    }
}
/* y = alpha A^T x + beta y (operator.rs:59-75) */
static void csc_top(void *ctx, double alpha, const double *x, double beta, double *y)
{
    csc_op *o = ctx;
    #pragma omp parallel for schedule(dynamic, 256)
    for (size_t c = 0; c < o->n; ++c) {
        double s = 0.0;
        for (int64_t k = o->cp[c]; k < o->cp[c + 1]; ++k) s += o->cv[k] * x[o->ri[k]];
        y[c] = alpha * s + beta * y[c];
    }
}
/* tau[c] += sum_r |A(r,c)| (operator.rs:82-113) ; sigma[r] += sum_c |A(r,c)| (operator.rs:123-154) */
static void csc_ac(void *ctx, double *tau)
{
    csc_op *o = ctx;
    for (size_t c = 0; c < o->n; ++c) { double s = 0.0; for (int64_t k = o->cp[c]; k < o->cp[c + 1]; ++k) s += fabs(o->cv[k]); tau[c] += s; }
}
static void csc_ar(void *ctx, double *sigma)
{
    csc_op *o = ctx;
    for (size_t r = 0; r < o->m; ++r) { double s = 0.0; for (int64_t k = o->rp[r]; k < o->rp[r + 1]; ++k) s += fabs(o->rv[k]); sigma[r] += s; }
}

int oc_solve_csc_cones(const oc_param *par, size_t n, size_t m, const double *vec_c,
                       const int64_t *colptr, const int32_t *rowidx, const double *vals, const double *vec_b,
                       size_t n_seg, const int32_t *seg_type, const int64_t *seg_len,
                       int use_ql, double *out_x, double *out_y, oc_trace *trace)
{
    csc_op o = { m, n, colptr, rowidx, vals, NULL, NULL, NULL };
    const int64_t nnz = n ? colptr[n] : 0;
    o.rp = (int64_t *)calloc(m + 2, sizeof(int64_t));
    o.ci = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
    o.rv = (double *)malloc(sizeof(double) * (size_t)(nnz ? nnz : 1));
    for (int64_t k = 0; k < nnz; ++k) o.rp[rowidx[k] + 2]++;
    for (size_t r = 0; r < m; ++r) o.rp[r + 2] += o.rp[r + 1];            /* rp[r + 1] = start of row r while filling */
    for (size_t c = 0; c < n; ++c)
        for (int64_t k = colptr[c]; k < colptr[c + 1]; ++k) {
            const int64_t d = o.rp[rowidx[k] + 1]++;
            o.ci[d] = (int32_t)c; o.rv[d] = vals[k];
        }
    oc_matop mc = { OC_MAT_GENERAL, n, 1, vec_c };
    oc_matop mb = { OC_MAT_GENERAL, m, 1, vec_b };
    oc_operator oc = oc_matop_as_operator(&mc), ob = oc_matop_as_operator(&mb);
    oc_operator oa = { &o, csc_size, csc_opf, csc_top, csc_ac, csc_ar };
    seg_cone sc;
    seg_cone_init(&sc, n_seg, seg_type, seg_len, par->eps_zero, use_ql);
    oc_cone cone = { &sc, seg_proj, seg_group };
    const int rc = run_and_extract(par, &oc, &oa, &ob, &cone, n, m, out_x, out_y, trace);
    free(sc.psd_work);
    free(o.rp); free(o.ci); free(o.rv);
    return rc;
}

/* ======================================================================================
 * ProbLP (totsu/src/problem/lp.rs)
 * ==================================================================================== */
typedef struct { oc_matop g, a; } lp_opa;      /* lp.rs:50-54 */
typedef struct { oc_matop h, b; } lp_opb;      /* lp.rs:119-123 */

static void lp_a_size(void *ctx, size_t *nr, size_t *nc) { lp_opa *o = ctx; *nr = o->g.nr + o->a.nr; *nc = o->g.nc; }
/* lp.rs:76-87 */
static void lp_a_op(void *ctx, double alpha, const double *x, double beta, double *y)
{
    lp_opa *o = ctx;
    oc_matop_op(&o->g, 0, alpha, x, beta, y);
    oc_matop_op(&o->a, 0, alpha, x, beta, y + o->g.nr);
}
/* lp.rs:89-98 */
static void lp_a_top(void *ctx, double alpha, const double *x, double beta, double *y)
{
    lp_opa *o = ctx;
    oc_matop_op(&o->g, 1, alpha, x, beta, y);
    oc_matop_op(&o->a, 1, alpha, x + o->g.nr, 1.0, y);
}
/* lp.rs:100-104 */
static void lp_a_ac(void *ctx, double *tau) { lp_opa *o = ctx; oc_matop_absadd(&o->g, 1, tau); oc_matop_absadd(&o->a, 1, tau); }
/* lp.rs:106-114 */
static void lp_a_ar(void *ctx, double *sigma) { lp_opa *o = ctx; oc_matop_absadd(&o->g, 0, sigma); oc_matop_absadd(&o->a, 0, sigma + o->g.nr); }

static void lp_b_size(void *ctx, size_t *nr, size_t *nc) { lp_opb *o = ctx; *nr = o->h.nr + o->b.nr; *nc = 1; }
/* lp.rs:147-158 */
static void lp_b_op(void *ctx, double alpha, const double *x, double beta, double *y)
{
    lp_opb *o = ctx;
    oc_matop_op(&o->h, 0, alpha, x, beta, y);
    oc_matop_op(&o->b, 0, alpha, x, beta, y + o->h.nr);
}
/* lp.rs:160-169 */
static void lp_b_top(void *ctx, double alpha, const double *x, double beta, double *y)
{
    lp_opb *o = ctx;
    oc_matop_op(&o->h, 1, alpha, x, beta, y);
    oc_matop_op(&o->b, 1, alpha, x + o->h.nr, 1.0, y);
}
static void lp_b_ac(void *ctx, double *tau) { lp_opb *o = ctx; oc_matop_absadd(&o->h, 1, tau); oc_matop_absadd(&o->b, 1, tau); }
static void lp_b_ar(void *ctx, double *sigma) { lp_opb *o = ctx; oc_matop_absadd(&o->h, 0, sigma); oc_matop_absadd(&o->b, 0, sigma + o->h.nr); }

/* lp.rs:309-337 (problem()) */
int oc_solve_lp(const oc_param *par, size_t n, size_t m, size_t p,
                const double *vec_c, const double *mat_g, const double *vec_h,
                const double *mat_a, const double *vec_b,
                double *out_x, double *out_y, oc_trace *trace)
{
    oc_matop mc = { OC_MAT_GENERAL, n, 1, vec_c };
    lp_opa oa_ = { { OC_MAT_GENERAL, m, n, mat_g }, { OC_MAT_GENERAL, p, n, mat_a } };
    lp_opb ob_ = { { OC_MAT_GENERAL, m, 1, vec_h }, { OC_MAT_GENERAL, p, 1, vec_b } };
    oc_operator oc = oc_matop_as_operator(&mc);
    oc_operator oa = { &oa_, lp_a_size, lp_a_op, lp_a_top, lp_a_ac, lp_a_ar };
    oc_operator ob = { &ob_, lp_b_size, lp_b_op, lp_b_top, lp_b_ac, lp_b_ar };
    const int32_t types[2] = { OC_CONE_RPOS, OC_CONE_ZERO };
    const int64_t lens[2] = { (int64_t)m, (int64_t)p };
    seg_cone sc;
    seg_cone_init(&sc, 2, types, lens, par->eps_zero, 0);
    oc_cone cone = { &sc, seg_proj, seg_group };
    return run_and_extract(par, &oc, &oa, &ob, &cone, n, m + p, out_x, out_y, trace);
}

/* ======================================================================================
 * ProbSOCP (totsu/src/problem/socp.rs)
 * ==================================================================================== */
typedef struct {
    size_t n, n_cones, p;
    oc_matop *mats_g, *vecs_c, *vecs_h;
    const double *scls_d;
    double abssum_scls_d;
    oc_matop mat_a, vec_b;
} socp_t;

static size_t socp_rows(const socp_t *s)
{
    size_t sum = 0;
    for (size_t i = 0; i < s->n_cones; ++i) sum += 1 + s->mats_g[i].nr;
    return sum + s->p;
}
static void socp_a_size(void *ctx, size_t *nr, size_t *nc) { socp_t *s = ctx; *nr = socp_rows(s); *nc = s->n; }
/* socp.rs:77-101 */
static void socp_a_op(void *ctx, double alpha, const double *x, double beta, double *y)
{
    socp_t *s = ctx;
    size_t done = 0;
    for (size_t i = 0; i < s->n_cones; ++i) {
        const size_t ni = s->mats_g[i].nr;
        oc_matop_op(&s->vecs_c[i], 1, -alpha, x, beta, y + done);
        oc_matop_op(&s->mats_g[i], 0, -alpha, x, beta, y + done + 1);
        done += 1 + ni;
    }
    oc_matop_op(&s->mat_a, 0, alpha, x, beta, y + done);
}
/* socp.rs:103-130 */
static void socp_a_top(void *ctx, double alpha, const double *x, double beta, double *y)
{
    socp_t *s = ctx;
    oc_scale(s->n, beta, y);
    size_t done = 0;
    for (size_t i = 0; i < s->n_cones; ++i) {
        const size_t ni = s->mats_g[i].nr;
        oc_matop_op(&s->vecs_c[i], 0, -alpha, x + done, 1.0, y);
        oc_matop_op(&s->mats_g[i], 1, -alpha, x + done + 1, 1.0, y);
        done += 1 + ni;
    }
    oc_matop_op(&s->mat_a, 1, alpha, x + done, 1.0, y);
}
/* socp.rs:132-141 */
static void socp_a_ac(void *ctx, double *tau)
{
    socp_t *s = ctx;
    for (size_t i = 0; i < s->n_cones; ++i) oc_matop_absadd(&s->vecs_c[i], 0, tau);
    for (size_t i = 0; i < s->n_cones; ++i) oc_matop_absadd(&s->mats_g[i], 1, tau);
    oc_matop_absadd(&s->mat_a, 1, tau);
}
/* socp.rs:143-162 */
static void socp_a_ar(void *ctx, double *sigma)
{
    socp_t *s = ctx;
    size_t done = 0;
    for (size_t i = 0; i < s->n_cones; ++i) {
        const size_t ni = s->mats_g[i].nr;
        oc_matop_absadd(&s->vecs_c[i], 1, sigma + done);
        oc_matop_absadd(&s->mats_g[i], 0, sigma + done + 1);
        done += 1 + ni;
    }
    oc_matop_absadd(&s->mat_a, 0, sigma + done);
}

static void socp_b_size(void *ctx, size_t *nr, size_t *nc) { socp_t *s = ctx; *nr = socp_rows(s); *nc = 1; }
/* socp.rs:194-217 */
static void socp_b_op(void *ctx, double alpha, const double *x, double beta, double *y)
{
    socp_t *s = ctx;
    size_t done = 0;
    for (size_t i = 0; i < s->n_cones; ++i) {
        const size_t ni = s->vecs_h[i].nr;
        oc_scale(1, beta, y + done);
        oc_add(1, alpha * s->scls_d[i], x, y + done);
        oc_matop_op(&s->vecs_h[i], 0, alpha, x, beta, y + done + 1);
        done += 1 + ni;
    }
    oc_matop_op(&s->vec_b, 0, alpha, x, beta, y + done);
}
/* socp.rs:219-246 */
static void socp_b_top(void *ctx, double alpha, const double *x, double beta, double *y)
{
    socp_t *s = ctx;
    oc_scale(1, beta, y);
    size_t done = 0;
    for (size_t i = 0; i < s->n_cones; ++i) {
        const size_t ni = s->vecs_h[i].nr;
        oc_add(1, alpha * s->scls_d[i], x + done, y);
        oc_matop_op(&s->vecs_h[i], 1, alpha, x + done + 1, 1.0, y);
        done += 1 + ni;
    }
    oc_matop_op(&s->vec_b, 1, alpha, x + done, 1.0, y);
}
/* socp.rs:248-257 */
static void socp_b_ac(void *ctx, double *tau)
{
    socp_t *s = ctx;
    tau[0] = tau[0] + s->abssum_scls_d;
    for (size_t i = 0; i < s->n_cones; ++i) oc_matop_absadd(&s->vecs_h[i], 1, tau);
    oc_matop_absadd(&s->vec_b, 1, tau);
}
/* socp.rs:259-279 -- note: adds scl_d itself, not |scl_d| (reference behaviour, kept) */
static void socp_b_ar(void *ctx, double *sigma)
{
    socp_t *s = ctx;
    size_t done = 0;
    for (size_t i = 0; i < s->n_cones; ++i) {
        const size_t ni = s->vecs_h[i].nr;
        sigma[done] = sigma[done] + s->scls_d[i];
        oc_matop_absadd(&s->vecs_h[i], 0, sigma + done + 1);
        done += 1 + ni;
    }
    oc_matop_absadd(&s->vec_b, 0, sigma + done);
}

/* socp.rs:430-473 (problem()) */
int oc_solve_socp(const oc_param *par, size_t n, size_t n_cones, const int64_t *ni, size_t p,
                  const double *vec_f, const double *mats_g, const double *vecs_h,
                  const double *vecs_c, const double *scls_d,
                  const double *mat_a, const double *vec_b,
                  double *out_x, double *out_y, oc_trace *trace)
{
    socp_t s;
    s.n = n; s.n_cones = n_cones; s.p = p;
    s.mats_g = (oc_matop *)malloc(sizeof(oc_matop) * (n_cones ? n_cones : 1));
    s.vecs_c = (oc_matop *)malloc(sizeof(oc_matop) * (n_cones ? n_cones : 1));
    s.vecs_h = (oc_matop *)malloc(sizeof(oc_matop) * (n_cones ? n_cones : 1));
    int32_t *types = (int32_t *)malloc(sizeof(int32_t) * (n_cones + 1));
    int64_t *lens = (int64_t *)malloc(sizeof(int64_t) * (n_cones + 1));
    size_t goff = 0, hoff = 0;
    for (size_t i = 0; i < n_cones; ++i) {
        const size_t nii = (size_t)ni[i];
        oc_matop g = { OC_MAT_GENERAL, nii, n, mats_g + goff };
        oc_matop c = { OC_MAT_GENERAL, n, 1, vecs_c + i * n };
        oc_matop h = { OC_MAT_GENERAL, nii, 1, vecs_h + hoff };
        s.mats_g[i] = g; s.vecs_c[i] = c; s.vecs_h[i] = h;
        goff += nii * n; hoff += nii;
        types[i] = OC_CONE_SOC; lens[i] = (int64_t)(1 + nii);
    }
    types[n_cones] = OC_CONE_ZERO; lens[n_cones] = (int64_t)p;
    s.scls_d = scls_d;
    s.abssum_scls_d = oc_abssum(n_cones, scls_d, 1);        /* socp.rs:456 */
    oc_matop ma = { OC_MAT_GENERAL, p, n, mat_a }, mb = { OC_MAT_GENERAL, p, 1, vec_b };
    s.mat_a = ma; s.vec_b = mb;

    oc_matop mf = { OC_MAT_GENERAL, n, 1, vec_f };
    oc_operator oc = oc_matop_as_operator(&mf);
    oc_operator oa = { &s, socp_a_size, socp_a_op, socp_a_top, socp_a_ac, socp_a_ar };
    oc_operator ob = { &s, socp_b_size, socp_b_op, socp_b_top, socp_b_ac, socp_b_ar };
    seg_cone sc;
    seg_cone_init(&sc, n_cones + 1, types, lens, par->eps_zero, 0);
    oc_cone cone = { &sc, seg_proj, seg_group };
    const int rc = run_and_extract(par, &oc, &oa, &ob, &cone, n, socp_rows(&s), out_x, out_y, trace);
    free(s.mats_g); free(s.vecs_c); free(s.vecs_h); free(types); free(lens);
    return rc;
}

/* ======================================================================================
 * ProbSDP (totsu/src/problem/sdp.rs)
 * ==================================================================================== */

/* matbuild/mod.rs:147-156 (SymPack arm): scale every strictly-upper entry of a packed matrix.
 * The reference loop runs c in 0..n-1 and scales the (c+1)-th column's off-diagonal part. */
void oc_matbuild_scale_nondiag_sympack(size_t n, double *packed, double alpha)
{
    if (n == 0) return;
    for (size_t c = 0; c + 1 < n; ++c) {
        const size_t i = sp_idx(c, c);
        const size_t ii = sp_idx(c + 1, c + 1);
        oc_scale(ii - i - 1, alpha, packed + i + 1);
    }
}

typedef struct { oc_matop f, a; } sdp_opa;     /* sdp.rs:49-53 */
typedef struct { oc_matop fn, b; } sdp_opb;    /* sdp.rs:118-122 */

static void sdp_a_size(void *ctx, size_t *nr, size_t *nc) { sdp_opa *o = ctx; *nr = o->f.nr + o->a.nr; *nc = o->f.nc; }
static void sdp_a_op(void *ctx, double alpha, const double *x, double beta, double *y)      /* sdp.rs:75-86 */
{ sdp_opa *o = ctx; oc_matop_op(&o->f, 0, alpha, x, beta, y); oc_matop_op(&o->a, 0, alpha, x, beta, y + o->f.nr); }
static void sdp_a_top(void *ctx, double alpha, const double *x, double beta, double *y)     /* sdp.rs:88-97 */
{ sdp_opa *o = ctx; oc_matop_op(&o->f, 1, alpha, x, beta, y); oc_matop_op(&o->a, 1, alpha, x + o->f.nr, 1.0, y); }
static void sdp_a_ac(void *ctx, double *tau) { sdp_opa *o = ctx; oc_matop_absadd(&o->f, 1, tau); oc_matop_absadd(&o->a, 1, tau); }
static void sdp_a_ar(void *ctx, double *sg) { sdp_opa *o = ctx; oc_matop_absadd(&o->f, 0, sg); oc_matop_absadd(&o->a, 0, sg + o->f.nr); }

static void sdp_b_size(void *ctx, size_t *nr, size_t *nc) { sdp_opb *o = ctx; *nr = o->fn.nr + o->b.nr; *nc = 1; }
static void sdp_b_op(void *ctx, double alpha, const double *x, double beta, double *y)      /* sdp.rs:147-158: -alpha on F_n */
{ sdp_opb *o = ctx; oc_matop_op(&o->fn, 0, -alpha, x, beta, y); oc_matop_op(&o->b, 0, alpha, x, beta, y + o->fn.nr); }
static void sdp_b_top(void *ctx, double alpha, const double *x, double beta, double *y)     /* sdp.rs:160-169 */
{ sdp_opb *o = ctx; oc_matop_op(&o->fn, 1, -alpha, x, beta, y); oc_matop_op(&o->b, 1, alpha, x + o->fn.nr, 1.0, y); }
static void sdp_b_ac(void *ctx, double *tau) { sdp_opb *o = ctx; oc_matop_absadd(&o->fn, 1, tau); oc_matop_absadd(&o->b, 1, tau); }
static void sdp_b_ar(void *ctx, double *sg) { sdp_opb *o = ctx; oc_matop_absadd(&o->fn, 0, sg); oc_matop_absadd(&o->b, 0, sg + o->fn.nr); }

/* sdp.rs:250-297 (new: scale_nondiag(sqrt 2) + reshape + stack columns) and :299-331 (problem()) */
int oc_solve_sdp(const oc_param *par, size_t n, size_t k, size_t p,
                 const double *vec_c, const double *syms_f,
                 const double *mat_a, const double *vec_b, double eps_zero, int use_ql,
                 double *out_x, double *out_y, oc_trace *trace)
{
    const size_t sk = k * (k + 1) / 2;
    const double fsqrt2 = sqrt(1.0 + 1.0);
    double *f = (double *)malloc(sizeof(double) * sk * (n + 1));
    memcpy(f, syms_f, sizeof(double) * sk * (n + 1));
    for (size_t i = 0; i <= n; ++i) oc_matbuild_scale_nondiag_sympack(k, f + i * sk, fsqrt2);
    /* symmat_f (sk x n, col-major) = columns 0..n-1; symvec_f_n = column n */
    sdp_opa oa_ = { { OC_MAT_GENERAL, sk, n, f }, { OC_MAT_GENERAL, p, n, mat_a } };
    sdp_opb ob_ = { { OC_MAT_GENERAL, sk, 1, f + n * sk }, { OC_MAT_GENERAL, p, 1, vec_b } };
    oc_matop mc = { OC_MAT_GENERAL, n, 1, vec_c };
    oc_operator oc = oc_matop_as_operator(&mc);
    oc_operator oa = { &oa_, sdp_a_size, sdp_a_op, sdp_a_top, sdp_a_ac, sdp_a_ar };
    oc_operator ob = { &ob_, sdp_b_size, sdp_b_op, sdp_b_top, sdp_b_ac, sdp_b_ar };
    const int32_t types[2] = { OC_CONE_PSD, OC_CONE_ZERO };
    const int64_t lens[2] = { (int64_t)sk, (int64_t)p };
    seg_cone sc;
    seg_cone_init(&sc, 2, types, lens, eps_zero, use_ql);
    oc_cone cone = { &sc, seg_proj, seg_group };
    const int rc = run_and_extract(par, &oc, &oa, &ob, &cone, n, sk + p, out_x, out_y, trace);
    free(sc.psd_work);
    free(f);
    return rc;
}

/* ======================================================================================
 * Counter-based synthetic data.  The same integer function is restated in
 * totsu_amd/csrc/thip_gen.hip so that the CPU baseline and every GPU shard generate
 * bit-identical f32 entries from (seed, stream, index) without shipping the matrix.
 * ==================================================================================== */
uint64_t oc_rng_hash(uint64_t seed, uint64_t stream, uint64_t idx)
{
    /* splitmix64 finaliser over a mixed counter */
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + stream * 0xD1B54A32D192ED03ull + idx * 0xBF58476D1CE4E5B9ull
                 + 0x94D049BB133111EBull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return z;
}

float oc_rng_uniform(uint64_t seed, uint64_t stream, uint64_t idx)
{
    return (float)(oc_rng_hash(seed, stream, idx) >> 40) * (1.0f / 16777216.0f);
}

float oc_rng_normal(uint64_t seed, uint64_t stream, uint64_t idx)
{
    /* Irwin-Hall: sum of four 16-bit uniforms, exact in integers; variance 4/12 -> * sqrt(3) */
    const uint64_t h = oc_rng_hash(seed, stream, idx);
    const uint32_t s = (uint32_t)(h & 0xFFFF) + (uint32_t)((h >> 16) & 0xFFFF)
                     + (uint32_t)((h >> 32) & 0xFFFF) + (uint32_t)((h >> 48) & 0xFFFF);
    /* mean of the sum = 4 * 32767.5 = 131070 */
    return ((float)((int32_t)s - 131070) * (1.0f / 65536.0f)) * 1.7320508f;
}

/* column-major block of the synthetic matrix: out(r,c) = scale * g(idx) + shift with
 * idx = (row0 + r) + (col0 + c) * ld_index, evaluated in f32 like thip_gen_matrix, stored as f64 */
void oc_gen_matrix(double *out, size_t n_row, size_t n_col, size_t lda, uint64_t seed, uint64_t stream,
                   uint64_t row0, uint64_t col0, uint64_t ld_index, int kind, float scale, float shift)
{
    #pragma omp parallel for schedule(static)
    for (ptrdiff_t c = 0; c < (ptrdiff_t)n_col; ++c) {
        const uint64_t cbase = (col0 + (uint64_t)c) * ld_index + row0;
        for (size_t r = 0; r < n_row; ++r) {
            const float g = kind ? oc_rng_normal(seed, stream, cbase + r) : oc_rng_uniform(seed, stream, cbase + r);
            out[(size_t)c * lda + r] = (double)(scale * g + shift);
        }
    }
}

void oc_gen_vector(double *out, size_t n, uint64_t seed, uint64_t stream, uint64_t idx0, int kind, float scale, float shift)
{
    for (size_t i = 0; i < n; ++i) {
        const float g = kind ? oc_rng_normal(seed, stream, idx0 + i) : oc_rng_uniform(seed, stream, idx0 + i);
        out[i] = (double)(scale * g + shift);
    }
}
