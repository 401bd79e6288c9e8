/*
 * totsu_oracle.h -- CPU f64 restatement of the Totsu first-order conic solver hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / the timed CPU
 * baseline.  The product path (totsu_amd + libtotsu_f32hip.so) never links, imports
 * or calls it.
 *
 * Parity status: PINNED.  The restatement reproduces
 *   - examples/nostd_cortex-m/log_qemu.txt:1-26 (17 residual triples, iteration 159,
 *     x to 16 digits) -- tests/test_oracle_golden.py
 *   - the known-answer tests of totsu_core/tests/solver.rs, totsu/tests/{lp,socp,sdp,qp,qcqp}.rs,
 *     cone_psd.rs:90-110, matop.rs:180-212, f64lapack.rs:262-287, matbuild/mod.rs:305-333.
 * The reference itself (Rust) cannot be built here (no rustc/cargo), see DESIGN.md.
 *
 * Every function cites the reference file:line (relative to
 * /root/reference/solver_rust_conic/) it follows.  No reference source is copied: the
 * reference is Rust generics over traits, this is plain C over (pointer,length) pairs.
 */
#ifndef TOTSU_ORACLE_H
#define TOTSU_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- LinAlg primitives: semantics of FloatGeneric (totsu_core/src/floatgeneric.rs:16-84) ---- */
double oc_norm(size_t n, const double *x);                                   /* :21-28 */
void   oc_copy(size_t n, const double *x, double *y);                        /* :30-37 */
void   oc_scale(size_t n, double alpha, double *x);                          /* :39-44 */
void   oc_add(size_t n, double alpha, const double *x, double *y);           /* :46-53 */
void   oc_adds(size_t n, double s, double *y);                               /* :55-60 */
double oc_abssum(size_t len, const double *x, size_t incx);                  /* :62-74 */
void   oc_transform_di(size_t n, double alpha, const double *d, const double *x,
                       double beta, double *y);                              /* :76-84 */
/* ---- LinAlgEx (floatgeneric.rs:328-439) ---- */
void   oc_transform_ge(int transpose, size_t n_row, size_t n_col, double alpha,
                       const double *mat, const double *x, double beta, double *y); /* :331-353 */
void   oc_transform_sp(size_t n, double alpha, const double *mat, const double *x,
                       double beta, double *y);                              /* :356-376 */
size_t oc_map_eig_worklen(size_t n);                                         /* :378-384 */
/* map kinds: 0 = keep e>0 (cone_psd.rs:69-76), 1 = sqrt of e>0 (matbuild/mod.rs:231-238) */
void   oc_map_eig(size_t sn, double *mat, int has_scale, double scale_diag, double eps_zero,
                  double *work, int map_kind);                               /* :386-439, Jacobi :273-324 */
/* same contract, eigen-decomposition by Householder tridiagonalisation + implicit QL
 * (stand-in for F64LAPACK's dsyevr path, totsu_f64lapack/src/f64lapack.rs:78-108,172-190);
 * work length n*n + n + n*n like f64lapack.rs:165-170 */
size_t oc_map_eig_worklen_ql(size_t n);
void   oc_map_eig_ql(size_t sn, double *mat, int has_scale, double scale_diag, double eps_zero,
                     double *work, int map_kind);
void   oc_vec_to_mat(size_t n, const double *v, double *m, int has_scale, double scale);  /* f64lapack.rs:195-224 */
void   oc_mat_to_vec(size_t n, double *m, double *v, int has_scale, double scale);        /* f64lapack.rs:226-255 */

/* ---- Cones (totsu_core/src/cone_*.rs) ---- */
enum { OC_CONE_ZERO = 0, OC_CONE_RPOS = 1, OC_CONE_SOC = 2, OC_CONE_ROTSOC = 3, OC_CONE_PSD = 4 };
void oc_proj_zero(int dual_cone, size_t n, double *x);      /* cone_zero.rs:38-44 */
void oc_proj_rpos(size_t n, double *x);                     /* cone_rpos.rs:38-45 */
void oc_proj_soc(size_t n, double *x);                      /* cone_soc.rs:38-65 */
void oc_proj_rotsoc(size_t n, double *x);                   /* cone_rotsoc.rs:38-65 */
/* returns 0 ok, -1 work shortage (cone_psd.rs:58-61) */
int  oc_proj_psd(size_t sn, double *x, double eps_zero, double *work, size_t worklen, int use_ql); /* cone_psd.rs:56-79 */

/* ---- Operator vtable (totsu_core/src/solver/operator.rs:11-156) ---- */
typedef struct oc_operator {
    void  *ctx;
    void (*size)(void *ctx, size_t *n_row, size_t *n_col);
    void (*op)(void *ctx, double alpha, const double *x, double beta, double *y);
    void (*trans_op)(void *ctx, double alpha, const double *x, double beta, double *y);
    void (*absadd_cols)(void *ctx, double *tau);
    void (*absadd_rows)(void *ctx, double *sigma);
} oc_operator;

/* ---- Cone vtable (totsu_core/src/solver/cone.rs:9-30) ---- */
typedef void (*oc_group_fn)(double *tau_group, size_t len);
typedef struct oc_cone {
    void *ctx;
    int  (*proj)(void *ctx, int dual_cone, double *x, size_t len);          /* 0 ok, -1 Err(()) */
    void (*product_group)(void *ctx, double *dp_tau, size_t len, oc_group_fn group);
} oc_cone;

/* ---- MatOp (totsu_core/src/matop.rs) ---- */
enum { OC_MAT_GENERAL = 0, OC_MAT_SYMPACK = 1 };
typedef struct oc_matop {
    int typ; size_t nr, nc;            /* SymPack(n): nr = nc = n */
    const double *array;               /* column-major / packed upper by columns */
} oc_matop;
oc_operator oc_matop_as_operator(oc_matop *m);
void oc_matop_op(const oc_matop *m, int transpose, double alpha, const double *x, double beta, double *y); /* matop.rs:76-96 */
void oc_matop_absadd(const oc_matop *m, int colwise, double *y);                                          /* matop.rs:98-138 */

/* ---- Solver (totsu_core/src/solver/solver.rs) ---- */
typedef struct oc_param {
    int64_t max_iter;       /* <0 : None */
    double  eps_acc, eps_inf, eps_zero;
    int64_t log_period;
} oc_param;
void oc_param_default(oc_param *p);                        /* solver.rs:27-41 */

enum { OC_OK = 0, OC_UNBOUNDED = 1, OC_INFEASIBLE = 2, OC_EXCESS_ITER = 3,
       OC_INVALID_OP = 4, OC_WORK_SHORTAGE = 5, OC_CONE_FAILURE = 6 };   /* solver_error.rs:3-17 */

/* one record per iteration (the values solver.rs:390-395 / 424-429 log) */
typedef struct oc_trace_rec {
    int64_t iter;
    int32_t kind;            /* 0: pri_dual_gap, 1: unbdd_infeas */
    double  v0, v1, v2;
} oc_trace_rec;

typedef struct oc_trace {
    oc_trace_rec *rec; size_t cap; size_t len;   /* filled up to cap */
    int64_t iters;                               /* index i of the final iteration */
    double  norm_b, norm_c;
    /* optional state snapshots: x then y (N+M doubles) at the iterations listed */
    const int64_t *snap_iters; size_t n_snap; double *snap_out;
    /* optional: dp_tau (N) then dp_sigma (M) as calc_precond leaves them (solver.rs:496-524) */
    double *precond_out;
} oc_trace;

size_t oc_query_worklen(size_t m, size_t n);               /* solver.rs:231-249 */
int oc_solve(const oc_param *par, oc_operator *op_c, oc_operator *op_a, oc_operator *op_b,
             oc_cone *cone, double *work, size_t worklen, oc_trace *trace);   /* solver.rs:285-321, 340-458 */

/* ---- Problem builders: flat-array entry points used by tests / cpu baseline ---- */
/* product cone over consecutive segments; PSD segments use eps_zero, private work */
int oc_solve_matop_cones(const oc_param *par, size_t n, size_t m,
                         const double *vec_c, const double *mat_a, const double *vec_b,
                         size_t n_seg, const int32_t *seg_type, const int64_t *seg_len,
                         int use_ql, double *out_x, double *out_y, oc_trace *trace);

/* the same with A given sparse, by columns (int64 column pointers, int32 row indices, f64 values): a user-defined
 * Operator in the pattern of examples/imgnr_udef/src/prob_op_a.rs:33-120 (operator.rs:11-156) */
int oc_solve_csc_cones(const oc_param *par, size_t n, size_t m, const double *vec_c,
                       const int64_t *colptr, const int32_t *rowidx, const double *vals, const double *vec_b,
                       size_t n_seg, const int32_t *seg_type, const int64_t *seg_len,
                       int use_ql, double *out_x, double *out_y, oc_trace *trace);

/* ProbLP (totsu/src/problem/lp.rs) */
int oc_solve_lp(const oc_param *par, size_t n, size_t m, size_t p,
                const double *vec_c, const double *mat_g, const double *vec_h,
                const double *mat_a, const double *vec_b,
                double *out_x, double *out_y, oc_trace *trace);
/* ProbSOCP (totsu/src/problem/socp.rs): mats_g concatenated (each ni x n col-major), vecs_h concatenated,
 * vecs_c concatenated (each n) */
int oc_solve_socp(const oc_param *par, size_t n, size_t n_cones, const int64_t *ni, size_t p,
                  const double *vec_f, const double *mats_g, const double *vecs_h,
                  const double *vecs_c, const double *scls_d,
                  const double *mat_a, const double *vec_b,
                  double *out_x, double *out_y, oc_trace *trace);
/* ProbSDP (totsu/src/problem/sdp.rs): syms_f = n+1 packed k(k+1)/2 arrays, concatenated, NOT yet scaled */
int oc_solve_sdp(const oc_param *par, size_t n, size_t k, size_t p,
                 const double *vec_c, const double *syms_f,
                 const double *mat_a, const double *vec_b, double eps_zero, int use_ql,
                 double *out_x, double *out_y, oc_trace *trace);
/* MatBuild helpers (totsu/src/matbuild/mod.rs) */
void oc_matbuild_scale_nondiag_sympack(size_t n, double *packed, double alpha);  /* :147-156 */

/* ---- counter-based synthetic data (shared definition with totsu_amd/csrc) ---- */
uint64_t oc_rng_hash(uint64_t seed, uint64_t stream, uint64_t idx);
float    oc_rng_uniform(uint64_t seed, uint64_t stream, uint64_t idx);   /* [0,1) on a 2^-24 grid */
float    oc_rng_normal(uint64_t seed, uint64_t stream, uint64_t idx);    /* Irwin-Hall(4), unit variance */

void oc_gen_matrix(double *out, size_t n_row, size_t n_col, size_t lda, uint64_t seed, uint64_t stream,
                   uint64_t row0, uint64_t col0, uint64_t ld_index, int kind, float scale, float shift);
void oc_gen_vector(double *out, size_t n, uint64_t seed, uint64_t stream, uint64_t idx0, int kind, float scale, float shift);

int oc_num_threads(void);
void oc_set_num_threads(int k);

#ifdef __cplusplus
}
#endif
#endif
