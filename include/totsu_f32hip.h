/*
 * totsu_f32hip.h -- C ABI of libtotsu_f32hip.so, the MI355X (gfx950) linear-algebra backend for
 * Totsu's first-order conic solver.  This is the drop-in boundary: a Rust `totsu_f32hip` crate
 * (`F32HIP: LinAlgEx`, `F32HIPSlice: SliceLike`) binds exactly these entry points,
 * the way `totsu_f32cuda` binds cuBLAS / cuSOLVER.
 *
 * Conventions
 *   - every `float *` / `const float *` is a DEVICE pointer unless the name starts with `host_`;
 *   - every call enqueues on the context stream (thip_set_stream) and returns without
 *     synchronising, except the ones that return a host scalar (marked SYNC).  This holds as stated by DEFAULT and
 *     always while a caller-provided stream is installed.  A host that drives everything through this API may opt into
 *     deferred execution of small calls with thip_set_lazy_gemv(1) (see there): those calls are then enqueued at the
 *     latest when the next non-deferred entry point -- including thip_get_stream / thip_sync -- is entered, so call
 *     order is still what every thip_* caller observes, but work the HOST enqueues by itself on the stream must be
 *     preceded by thip_get_stream();
 *   - return value: 0 = ok, otherwise a hipError_t (or THIP_E_* below); thip_last_error() gives text.
 *     The reference backends assert on library status (f32cuda.rs:38 etc.): a binding should
 *     `assert_eq!(rc, 0)`;
 *   - zero-length vectors and zero-sized matrices are legal everywhere (matop.rs:83-85);
 *   - one context per process (one process per GPU); like the reference's backends (thread_local state, !Send types)
 *     the API is meant to be driven from one host thread -- uploads and the shared scratch are mutex-protected, and
 *     separate thip_solver objects may be driven from separate threads (they own their scratch);
 *   - citations are relative to /root/reference/solver_rust_conic/.
 */
#ifndef TOTSU_F32HIP_H
#define TOTSU_F32HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define THIP_E_INVALID   10001   /* bad argument */
#define THIP_E_NOTINIT   10002
#define THIP_E_NOGPU     10003   /* no HIP device: the product path fails loudly, there is no CPU fallback */
#define THIP_E_WORK      10004   /* work buffer too short */
#define THIP_E_NOCONV    10005   /* eigen iteration did not converge */
#define THIP_E_TIMEOUT   10006   /* a bounded device-side wait ran out (a peer rank or its host stalled, or the one-pass kernel
                                  * gave up on some rank of a column-sharded run): the solve cannot go on consistently */

/* ---------------------------------------------------------------------------------------------
 * Context.  Replaces totsu_f32cuda/src/cuda_mgr.rs:24-111 (context + cuBLAS/cuSOLVER handles,
 * device 0 only there; any device here).
 * ------------------------------------------------------------------------------------------- */
int  thip_init(int device);                        /* cuda_mgr.rs:30-60 */
int  thip_shutdown(void);
int  thip_device_count(int *host_count);
int  thip_set_stream(void *hip_stream);            /* NULL = the context's own stream */
void *thip_get_stream(void);                       /* runs what is deferred first; NULL = that failed (thip_last_error) */
int  thip_sync(void);                              /* SYNC */
const char *thip_last_error(void);
const char *thip_version(void);

/* device memory: replaces cuda_mgr.rs:119-138 (buf_from_slice / buf_zeroes) and
 * f32cuda_slice.rs:343-355 (sync_from_dev / sync_from_host) */
int thip_alloc(size_t n, float **out);             /* n floats, 256-byte aligned, NOT zeroed */
int thip_alloc_zeroed(size_t n, float **out);
int thip_free(float *p);
int thip_h2d(float *dst, const float *host_src, size_t n);     /* async on the stream (pageable src: staged) */
int thip_d2h(float *host_dst, const float *src, size_t n);     /* SYNC */
int thip_get(const float *x, size_t idx, float *host_out);     /* SYNC; SliceLike::get, slicelike.rs:54-59 */
int thip_set(float *x, size_t idx, float val);                 /*       SliceLike::set, slicelike.rs:62-69 */

/* ---------------------------------------------------------------------------------------------
 * LinAlg (totsu_core/src/solver/linalg.rs:10-68); CUDA counterparts f32cuda.rs:27-136
 * ------------------------------------------------------------------------------------------- */
int thip_norm(size_t n, const float *x, float *host_out);                       /* SYNC  linalg.rs:22  (cublasSnrm2, f32cuda.rs:27-42) */
int thip_copy(size_t n, const float *x, float *y);                              /*       linalg.rs:29  (f32cuda.rs:44-57) */
int thip_scale(size_t n, float alpha, float *x);                                /*       linalg.rs:35  (f32cuda.rs:59-69) */
int thip_add(size_t n, float alpha, const float *x, float *y);                  /*       linalg.rs:42  (f32cuda.rs:71-84) */
int thip_adds(size_t n, float s, float *y);                                     /*       linalg.rs:49  (f32cuda.rs:86-99) */
int thip_abssum(size_t len, const float *x, size_t incx, float *host_out);      /* SYNC  linalg.rs:56  (f32cuda.rs:101-121); sums |x[0]|,|x[incx]|,.. < len */
int thip_transform_di(size_t n, float alpha, const float *d, const float *x,
                      float beta, float *y);                                    /*       linalg.rs:67  (cublasSsbmv k=0, f32cuda.rs:123-136) */

/* ---------------------------------------------------------------------------------------------
 * LinAlgEx (totsu_core/src/linalg_ex.rs:7-66); CUDA counterparts f32cuda.rs:144-370
 * ------------------------------------------------------------------------------------------- */
/* y = alpha * G^(T) x + beta * y, G column-major n_row x n_col, lda = n_row.
 * linalg_ex.rs:23 (cublasSgemv, f32cuda.rs:144-171) */
int thip_transform_ge(int transpose, size_t n_row, size_t n_col, float alpha, const float *mat,
                      const float *x, float beta, float *y);
/* OPT-IN deferred, batched execution of small calls (off by default; off while a caller stream is installed with
 * thip_set_stream): thip_transform_ge on matrices <= 64 MB, thip_scale / thip_add on vectors <= 1024 long and the
 * single-cone projections thip_proj_soc / _rotsoc / _rpos / _zero are RECORDED instead of launched.  A composite operator
 * issues them per block (ProbSOCPOpA / ProbSOCPOpB, socp.rs:77-130,194-246: 5000 per product at BASELINE configs[2];
 * ProbSOCPCone::proj, socp.rs:296-313: 1000 per projection); the library runs the record -- one grouped launch per kind
 * of product plus one finishing launch, one launch for a run of projections -- as soon as ANY other entry point is
 * called or a new call would read or overwrite a pending result, so call order is still what every caller of this API
 * observes.  Contributions to the same y are summed in a fixed order of their own (last-bit differences from sequential
 * accumulation).  Errors of a deferred call surface at the entry point that runs the record.  The trait-level hosts
 * (include/totsu_f32hip.hpp Solver, totsu_amd.Solver, the Rust crate's init()) switch it on for their own calls;
 * THIP_LAZY_GEMV=1 switches it on from the environment. */
int thip_set_lazy_gemv(int on);
int thip_get_lazy_gemv(int *host_on);
int thip_lazy_gemv_stats(int64_t *host_deferred, int64_t *host_flushes);
/* a flushed segment is kept as a plan (call list, device tables, launch geometry); the loop repeats its call sequence
 * every iteration, so the next pass re-uses the plan (hit) instead of analysing and building again (miss) */
int thip_lazy_plan_stats(int64_t *host_hits, int64_t *host_misses);
/* read-ahead of SYNC scalars (deferred mode only): the thip_get / thip_norm calls of a host loop that asks for the same
 * addresses every pass -- the reference's ConeSOC::proj, cone_soc.rs:44-47, 2 reads per cone -- are learnt, and on the next
 * pass the first read fetches all of them with one kernel and one transfer (a fetch); the others are served from the host
 * copy as long as each request is the predicted one and nothing issued in between writes a range still to be read */
int thip_lazy_read_stats(int64_t *host_served, int64_t *host_fetches);
/* y = alpha * S x + beta * y, S symmetric, packed upper by columns.  linalg_ex.rs:37 (cublasSspmv, f32cuda.rs:174-187) */
int thip_transform_sp(size_t n, float alpha, const float *mat, const float *x, float beta, float *y);
/* Reduced-precision STORAGE of a dense operator (not part of the reference's trait surface):
 * bf16 elements (round to nearest even), f32 accumulation.  mat16 is column-major with leading dimension ld16 >= n_row
 * (a multiple of 8 and a 16-byte aligned base give the 16-byte-load kernel); rows n_row..ld16 are written as zeros.
 * thip_transform_ge_bf16 has the semantics of thip_transform_ge on the rounded matrix. */
enum { THIP_A_F32 = 0, THIP_A_BF16 = 1, THIP_A_F16 = 2 };
int thip_to_bf16(size_t n_row, size_t n_col, const float *mat, uint16_t *mat16, size_t ld16);
int thip_transform_ge_bf16(int transpose, size_t n_row, size_t n_col, float alpha, const uint16_t *mat16, size_t ld16,
                           const float *x, float beta, float *y);
/* The same with IEEE f16 elements and one power-of-two scale per column (8x finer rounding than bf16 at the same
 * bytes; the scale keeps any f32 column inside f16's range): stored(r, c) = f16(a(r, c) * s_c), s_c the power of two
 * that brings the column's largest magnitude into [2^13, 2^14); inv_scale[c] = 1 / s_c (n_col floats, written by
 * thip_to_f16, read by the products). */
int thip_to_f16(size_t n_row, size_t n_col, const float *mat, uint16_t *mat16, size_t ld16, float *inv_scale);
int thip_transform_ge_f16(int transpose, size_t n_row, size_t n_col, float alpha, const uint16_t *mat16, size_t ld16,
                          const float *inv_scale, const float *x, float beta, float *y);
/* linalg_ex.rs:44 (f32cuda.rs:243-251) */
size_t thip_map_eig_worklen(size_t n);
/* linalg_ex.rs:64-65 with the two closures that exist in the reference, evaluated on the device:
 *   map_kind 0: e > 0 -> e            (cone_psd.rs:69-76)
 *   map_kind 1: e > 0 -> sqrt(e)      (totsu/src/matbuild/mod.rs:231-238)
 * mat: packed upper by columns, length n(n+1)/2, in/out; scale_diag applied as f64lapack.rs:195-255. */
int thip_map_eig(size_t n, float *mat, int has_scale, float scale_diag, float eps_zero,
                 float *work, size_t worklen, int map_kind);
/* arbitrary host closure, two phases (keeps the `M: Fn(F)->Option<F>` contract of linalg_ex.rs:64-65):
 *   1) decompose: eigenvalues -> host_w[n] (SYNC), eigenvectors stay in work;
 *   2) the host applies its closure: host_e[i] = mapped value, host_keep[i] = 1 for Some, 0 for None;
 *   3) rebuild: mat <- sum_i keep_i * e_i z_i z_i^T, diag scaled back, repacked. */
int thip_eig_decompose(size_t n, float *mat, int has_scale, float scale_diag, float eps_zero,
                       float *work, size_t worklen, float *host_w);
int thip_eig_rebuild(size_t n, float *mat, int has_scale, float scale_diag,
                     float *work, size_t worklen, const float *host_e, const uint8_t *host_keep);
/* which engine served the last decomposition of order > 32 on this context (dsyevr / syevdx in the reference,
 * f64lapack.rs:78-108, f32cuda.rs:253-263): *host_engine = 1 host QL + rotation replay, 2 = device multisection + twisted
 * factorisation (certified), 3 = 2 failed its certificate and 1 took over; *host_polish = Newton-Schulz polish steps
 * applied; host_cert[0] = ||Z Z^T - I||_F, host_cert[1] = largest relative residual of the tridiagonal stage,
 * host_cert[2] = how the Householder reduction ran: 1 / 2 one persistent launch (device / one XCD), 0 one launch per reflector, -1 the
 * persistent launch gave up (a bounded spin ran out) and the launches took over.  host_cert holds 3 floats. */
int thip_eig_engine_info(int *host_engine, int *host_polish, float *host_cert);

/* Sparse operators: y = alpha * A x + beta * y, A in CSR (int64 row pointers, int32 column indices,
 * all on the device); abs_mode != 0 uses |A| and x = 1 (MatOp::absadd_*, matop.rs:98-117).  The transposed product
 * is the same call on the CSR of A^T.  Backs a user-level `Operator` (operator.rs:11-156) on the trait-level path. */
int thip_spmv_csr(size_t n_row, size_t n_col, size_t nnz, const int64_t *dev_rowptr, const int32_t *dev_colidx,
                  const float *vals, float alpha, const float *x, float beta, float *y, int abs_mode);

/* A sparse operator held ONCE on the device and serving both products (round 6): built from the caller's HOST arrays in
 * compressed-sparse-column form (int64 column pointers, int32 row indices, f32 values -- what a MatBuild<General> with its zeros
 * dropped holds, matbuild/mod.rs:22-41) into 4096 x 4096 tiles of {value, local row | local column << 16} entries; both A x and
 * A^T y stream the same 8 bytes per entry with 16-byte loads and scatter into LDS accumulators -- no second copy of the values,
 * no global atomics (thip_sptile.hip has the measurements behind the format).  The accumulators are 64-bit fixed-point words fed by
 * integer LDS adds: the sums are bitwise reproducible from run to run.  Rows inside a column may come in any order; where they ascend,
 * a tile of full height whose every column holds all 4096 rows (a dense block of the matrix) is stored WITHOUT its index words -- 4 bytes
 * per entry on the device and per product (thip_sptile_layout reports how many).
 * thip_sptile_mv is `Operator::op / trans_op` (operator.rs:40-75) for such an operator: y = alpha A x + beta y,
 * transpose != 0: A^T; abs_mode != 0: |A| and x = 1 (absadd_rows / absadd_cols, operator.rs:82-154).  x, y on the device. */
typedef struct thip_sptile thip_sptile;
int thip_sptile_create(size_t n_row, size_t n_col, size_t nnz, const int64_t *host_colptr, const int32_t *host_rowidx,
                       const float *host_vals, thip_sptile **out);
int thip_sptile_destroy(thip_sptile *mat);
int thip_sptile_mv(thip_sptile *mat, int transpose, float alpha, const float *x, float beta, float *y, int abs_mode);
/* stored entries (tiles padded to whole 16-byte quads), tiles, work items and slices (partial sums per block) of the two
 * products, bytes on the device */
int thip_sptile_info(const thip_sptile *mat, size_t *host_nnz_stored, int *host_tiles, int *host_items_n, int *host_items_t,
                     int *host_slices_n, int *host_slices_t, size_t *host_bytes);
/* how the entries are stored: tiles held without indices (full 4096-row tiles whose every column is full: 4 bytes per entry),
 * entries that carry an index (8 bytes per entry), and the bytes of entries ONE product streams */
int thip_sptile_layout(const thip_sptile *mat, int *host_dense_tiles, size_t *host_indexed_entries, size_t *host_bytes_per_product);

/* ---------------------------------------------------------------------------------------------
 * Device-resident variants used by the fused path (no host round trip).  They replace host loops
 * in totsu_core that a generic backend cannot intercept ("hard parts").
 * ------------------------------------------------------------------------------------------- */
int thip_norm_dev(size_t n, const float *x, float *dev_out);                    /* ||x||_2 -> *dev_out */
int thip_dot_dev(size_t n, const float *x, const float *y, float *dev_out);     /* x.y -> *dev_out */
int thip_abssum_dev(size_t len, const float *x, size_t incx, float *dev_out);

/* MatOp::absadd_impl for General (matop.rs:98-117) in one pass each: tau[c] += sum_r |G(r,c)|,
 * sigma[r] += sum_c |G(r,c)| -- replaces n + m blocking cublasSasum calls */
int thip_absadd_cols(size_t n_row, size_t n_col, const float *mat, float *tau);
int thip_absadd_rows(size_t n_row, size_t n_col, const float *mat, float *sigma);
/* SymPack arm, matop.rs:119-136 */
int thip_absadd_sympack(size_t n, const float *mat, float *y);
/* calc_precond host loops, solver.rs:501-506: x[i] = 1 / max(x[i], eps_zero) */
int thip_recip_max(size_t n, float eps_zero, float *x);

/* Stacking the blocks of a composite operator into ONE column-major matrix on the device (what
 * ProbLPOpA / ProbSOCPOpA / ProbSDPOpA are, lp.rs:76-98, socp.rs:77-130, sdp.rs:75-97, seen as a single MatOp):
 * dst(r, c) = sign * src(r, c) for an n_row x n_col column-major block (lda = n_row), dst pointing at the block's first
 * row inside a matrix of leading dimension ld_dst; transposed != 0: src is an n_col-vector written as ONE row (the
 * -c_i^T rows of socp.rs:88-93). */
int thip_copy_block(int transposed, size_t n_row, size_t n_col, float sign, const float *src, float *dst, size_t ld_dst);

/* Cone projections (totsu_core/src/cone_*.rs) on device-resident vectors */
enum { THIP_CONE_ZERO = 0, THIP_CONE_RPOS = 1, THIP_CONE_SOC = 2, THIP_CONE_ROTSOC = 3, THIP_CONE_PSD = 4 };
int thip_proj_zero(int dual_cone, size_t n, float *x);          /* cone_zero.rs:38-44 */
int thip_proj_rpos(size_t n, float *x);                         /* cone_rpos.rs:38-45 (host loop in the reference) */
int thip_proj_soc(size_t n, float *x);                          /* cone_soc.rs:38-65, one cone */
int thip_proj_rotsoc(size_t n, float *x);                       /* cone_rotsoc.rs:38-65, one cone */
/* many cones in one launch: cone i occupies x[host_offs[i] .. host_offs[i+1]); rotated != 0 -> ConeRotSOC.
 * Replaces the per-cone loop of ProbSOCPCone::proj (totsu/src/problem/socp.rs:296-313). */
int thip_proj_soc_batched(float *x, const int64_t *dev_offs, size_t n_cones, int rotated, size_t max_len);
/* ConePSD::proj, cone_psd.rs:56-79: returns THIP_E_WORK on work shortage (-> Err(())) */
int thip_proj_psd(size_t sn, float *x, float eps_zero, float *work, size_t worklen);
/* the `group` closure of solver.rs:509-520 applied to many blocks at once (product_group) */
int thip_group_min_batched(float *dp_tau, const int64_t *dev_offs, size_t n_groups, size_t max_len);

/* ---------------------------------------------------------------------------------------------
 * Fused conic iteration on the device.  Native restatement of
 * totsu_core/src/solver/solver.rs:340-657 (SolverCore) for operators that are dense matrices:
 * all per-iteration arithmetic, the projections and the termination test run on the GPU; the host
 * only polls a status word.
 * ------------------------------------------------------------------------------------------- */
typedef struct thip_solver thip_solver;

/* Arithmetic of the iterate updates x += T o tx, y += S o ty (solver.rs:542,560) of the fused loop.  Storage and
 * every product stay f32 either way.
 *   THIP_STATE_COMPENSATED: each of the five iterate vectors (x_x, x_y, x_s, u, v) carries a Kahan term, O(n + m)
 *     floats next to an m x n matrix.  Without it the f32 iterate stops moving once an update is below half an ulp of
 *     the entry it is added to, and the dual criterion floors (1.6e-4 on the n = 50 000 SOCP) -- a floor
 *     the reference's own f32 backend shares, which is why it is run at eps_acc = 1e-3 (benchmark_lp/src/main.rs:62-65);
 *   THIP_STATE_PLAIN: x + inc in plain f32, the reference's literal arithmetic. */
enum { THIP_STATE_COMPENSATED = 0, THIP_STATE_PLAIN = 1 };

/* ABI 3 (thip_version): thip_param grew by state_arith + reserved in round 2 (32 -> 40 bytes): callers built against the
 * 32-byte struct must be rebuilt.  Its zero value selects THIP_STATE_COMPENSATED, i.e. the DEFAULT fused loop does not
 * perform the reference's literal f32 iterate additions (solver.rs:542,560); set THIP_STATE_PLAIN for those. */
typedef struct thip_param {           /* solver.rs:13-41 */
    int64_t max_iter;                 /* < 0: None */
    float   eps_acc, eps_inf, eps_zero;
    int64_t log_period;               /* 0: no periodic log */
    int32_t state_arith;              /* THIP_STATE_* (not in the reference; zero-initialised = the default) */
    int32_t reserved;                 /* 0 */
} thip_param;

enum { THIP_ST_RUNNING = -1,
       THIP_ST_OK = 0, THIP_ST_UNBOUNDED = 1, THIP_ST_INFEASIBLE = 2, THIP_ST_EXCESS_ITER = 3,
       THIP_ST_INVALID_OP = 4, THIP_ST_WORK_SHORTAGE = 5, THIP_ST_CONE_FAILURE = 6 };   /* solver_error.rs:3-17 */

enum { THIP_SCHED_REFERENCE = 0,   /* 6 single GEMVs per iteration, the reference's op sequence (solver.rs:122-597) */
       THIP_SCHED_FUSED     = 1,   /* 3 passes over A: N and T products of one stage share a tile read */
       THIP_SCHED_CARRIED   = 2,   /* 2 passes: K*rx obtained by linearity from the criteria products */
       THIP_SCHED_SWEEP     = 3 }; /* 1 pass: per column, both dots, the x_x / u updates and both axpys while the column is in
                                    * registers (thip_sweep.hip; dense f32 A on one GPU, else the carried schedule runs) */

typedef struct thip_problem {
    size_t n, m;                      /* A is m x n (local rows if sharded) */
    const float *mat_a;               /* device, column-major, lda = m */
    const float *vec_b;               /* device, m */
    const float *vec_c;               /* device, n */
    const float *vec_b_rowabs;        /* device, m, or NULL: what op_b.absadd_rows adds per row (default |b|).
                                         ProbSOCPOpB adds scl_d, not |scl_d| (socp.rs:259-279) */
    size_t n_seg;                     /* product cone over consecutive segments of the m rows */
    const int32_t *host_seg_type;     /* THIP_CONE_* */
    const int64_t *host_seg_len;
} thip_problem;

typedef struct thip_status {
    int32_t state;                    /* THIP_ST_* */
    int64_t iter;                     /* index i of the last executed iteration */
    int32_t kind;                     /* 0: pri_dual_gap valid, 1: unbdd_infeas valid */
    float   cri[3];
    float   tau, kappa;
    float   norm_b, norm_c;
} thip_status;

/* collective hook for a row-sharded A: sum-all-reduce `n` floats in place on the
 * given stream.  NULL = single GPU. */
typedef int (*thip_allreduce_fn)(void *ctx, float *dev_buf, size_t n, void *hip_stream);

/* Native RCCL (xGMI) communicator, one per process: rank 0 creates the 128-byte id, the launcher distributes it
 * (e.g. one torch.distributed broadcast), every rank calls thip_comm_init.  thip_solver_use_rccl installs an
 * all-reduce that is enqueued on the library's own stream (no event hand-off to a communication stream). */
int thip_comm_unique_id(uint8_t *host_id128);
int thip_comm_init(int rank, int world, const uint8_t *host_id128);
int thip_comm_allreduce(float *dev_buf, size_t n);       /* in-place float sum over the ranks, on the context stream */
int thip_comm_count(int *host_ranks);                    /* ranks of the live communicator, read back from RCCL (ncclCommCount); 0 = none */
int thip_comm_destroy(void);
int thip_solver_use_rccl(thip_solver *s);                /* before thip_solver_init */

/* One-shot all-reduce over peer-mapped buffers (third transport behind thip_solver_set_allreduce), for the latency-bound
 * messages of the loop (n + 1024 floats: 204 KB at BASELINE configs[2]): every rank publishes its contribution in a slot
 * the peers have mapped (hipIpcGetMemHandle / hipIpcOpenMemHandle), reads its N - 1 peers directly -- one xGMI link
 * each -- and sums in rank order (identical bits on every rank); ONE launch per call, a 4-byte flag handshake per
 * chunk of 2048 floats, no trailing barrier (slots alternate by call parity).  Up to 16 ranks, messages up to 2 MB;
 * works between processes sharing one GPU as well (that is how a 1-GPU box tests it).
 *   every rank: thip_oneshot_init(rank, world, max_floats, handle)    -> its 64-byte IPC handle
 *   the launcher all-gathers the handles (rank order)                  -> thip_oneshot_connect(handles[world * 64])
 *   thip_solver_use_oneshot(s) before thip_solver_init; thip_oneshot_destroy() after a barrier at the end.
 * A rank that waits more than ~4 s for a peer gives up and raises an error word instead of hanging the GPU:
 * thip_oneshot_error (SYNC) reads it.  The collective runs on whatever stream the overlap mode hands it. */
int thip_oneshot_init(int rank, int world, size_t max_floats, uint8_t *host_handle64);
int thip_oneshot_connect(const uint8_t *host_handles);
int thip_oneshot_allreduce(float *dev_buf, size_t n);    /* in-place float sum over the ranks, on the context stream */
int thip_oneshot_error(int *host_err);                   /* SYNC */
int thip_oneshot_destroy(void);
int thip_solver_use_oneshot(thip_solver *s);             /* before thip_solver_init */

int thip_solver_create(const thip_problem *prob, const thip_param *par, int schedule, thip_solver **out);
/* Sparse A for the fused loop (before thip_solver_init; prob->mat_a may then be NULL): CSR of A (m x n) and CSR of
 * A^T (n x m), int64 row pointers / int32 column indices / f32 values, all on the device.  Each stage then costs one
 * gather over A and one over A^T instead of a pass over a dense matrix. */
int thip_solver_set_csr(thip_solver *s, size_t nnz,
                        const int64_t *dev_rowptr, const int32_t *dev_colidx, const float *dev_vals,
                        const int64_t *dev_t_rowptr, const int32_t *dev_t_colidx, const float *dev_t_vals);
/* Sparse A for the fused loop, ONE copy (before thip_solver_init; prob->mat_a may be NULL; `mat` must outlive the solver).
 * Every schedule runs on it; THIP_SCHED_SWEEP is the one-pass recurrence in three launches -- A^T [v x_y], the per-column
 * updates, A [u x_x'] -- 16 bytes per stored entry and iteration (thip_solver_set_csr's two copies under the carried schedule:
 * 32).  Replaces SolverCore's op / trans_op calls on a user's sparse Operator (solver.rs:109-157). */
int thip_solver_set_sptile(thip_solver *s, thip_sptile *mat);
int thip_solver_set_allreduce(thip_solver *s, thip_allreduce_fn fn, void *ctx);
/* Row-sharded runs: where the all-reduce of a stage's A^T y runs relative to the other work (solver.rs:146 vs 149,
 * 122 vs 125 are the independences used).  May be switched between thip_solver_run calls.
 *   THIP_OVERLAP_OFF (0, default)  in order on the launch stream;
 *   THIP_OVERLAP_LOCAL_ROWS (1)    on a side HIP stream of the solver (event in / event out) while the launch stream goes
 *       on with the stage's work on the LOCAL rows (the x_y / x_s update and the cone projections in the x-stage, the v
 *       update in the y-stage); the x_x / u / tau / kappa updates wait for it.  Results are bitwise those of mode 0.
 *       Hides ~10 us of work and costs two launches + two event hand-offs per stage (+27 us at world size 1);
 *   THIP_OVERLAP_COLUMN_PIPELINE (2)  carried schedule, dense A: every stage's products run as two launches over the
 *       column halves [0, n1) and [n1, n) (n1 on a boundary of the GEMV plan's column chunks), and the all-reduce of a
 *       half's A^T y travels on the side stream under the NEXT half-launch, which only needs the entries of the other
 *       half; the sharded block partials ride with the second half; the termination test of iteration k is enqueued
 *       after the first half-launch of iteration k + 1 (which writes scratch only).  Every collective has a whole
 *       half-pass over the local A to complete in.  Costs six more launches per iteration;
 *   THIP_OVERLAP_COLUMN_INORDER (3)   the kernels of mode 2 with the collectives in order on the launch stream: the
 *       bitwise reference of mode 2 (same arithmetic, no concurrency).
 * Modes 2 / 3 fall back to 1 / 0 where the pipeline does not apply (thip_solver_overlap_info tells).  In modes 1 and 2
 * the hook receives the side stream; a hook that ignores its stream argument stays correct (and un-overlapped). */
enum { THIP_OVERLAP_OFF = 0, THIP_OVERLAP_LOCAL_ROWS = 1, THIP_OVERLAP_COLUMN_PIPELINE = 2, THIP_OVERLAP_COLUMN_INORDER = 3 };
int thip_solver_set_overlap(thip_solver *s, int mode);
/* GEMV plan autotune of this solver (thip_solver_init times nine tilings on the actual matrix and keeps the fastest).
 * For a given plan every result is bitwise reproducible run to run; two solves that autotune to different plans agree
 * to f32 round-off only.  on = 0: the shape heuristic, whatever the timings -- bit-reproducible across runs and hosts.
 * Before thip_solver_init (or between runs: takes effect at the next init / storage switch). */
int thip_solver_set_gemv_autotune(thip_solver *s, int on);
/* Leading-dimension padding of the library-owned f32 copy of A (made when m is not a multiple of `floats`, default 16 =
 * 64 bytes, and the copy fits a third of the free HBM): 0 = never copy (stream the caller's matrix as
 * it is).  Before thip_solver_init. */
int thip_solver_set_lda_pad(thip_solver *s, int floats);
/* the mode the next thip_solver_run will actually use, GEMV launches per pass over A (2 when column-split) and the split
 * column n1 (0 = none) */
int thip_solver_overlap_info(thip_solver *s, int *host_mode, int *host_launches_per_pass, size_t *host_split_col);
/* Storage of the dense A the iteration streams: THIP_A_F32 (default: prob->mat_a as given), or THIP_A_BF16 /
 * THIP_A_F16 (a library-owned 16-bit copy, made on the first request; f16 is column-scaled and rounds 8x finer than
 * bf16: half the bytes per pass, the problem solved is the one with the ROUNDED matrix).  Before thip_solver_init the preconditioners are computed from the stored form; between
 * thip_solver_run calls it switches the operator of the running iteration (e.g. bf16 passes first, f32 passes to
 * finish on the exact matrix) -- the iteration is a fixed-point method, so the iterate carries over.
 * prob->mat_a must stay valid while THIP_A_F32 may still be selected. */
int thip_solver_set_a_storage(thip_solver *s, int a_kind);
/* A caller-built bf16 matrix (thip_to_bf16 on column blocks, or any producer of bf16 bit patterns), column-major with
 * leading dimension ld16 >= m; before thip_solver_init; prob->mat_a may then be NULL (and THIP_A_F32 cannot be
 * selected).  The f32 matrix never has to exist as a whole: a 16-bit A is half the HBM footprint, e.g. BASELINE.json's
 * 320 GB LP (configs[4]) is 160 GB and fits one 288 GB MI355X.  The matrix stays caller-owned. */
int thip_solver_set_a_bf16(thip_solver *s, const uint16_t *mat16, size_t ld16);
/* the same for a caller-built f16 matrix with its per-column inverse scales (thip_to_f16) */
int thip_solver_set_a_f16(thip_solver *s, const uint16_t *mat16, size_t ld16, const float *inv_scale);
int thip_solver_init(thip_solver *s);                                 /* calc_norms + init_vecs + calc_precond, solver.rs:460-524 */
/* enqueue up to max_steps iterations (the device stops by itself on termination), poll every
 * `poll_every` iterations; returns when terminated or after max_steps.  SYNC. */
int thip_solver_run(thip_solver *s, int64_t max_steps, int64_t poll_every, thip_status *host_status);
int thip_solver_status(thip_solver *s, thip_status *host_status);     /* SYNC */
/* Continue a solve that ended THIP_ST_OK / THIP_ST_EXCESS_ITER (criteria kind 0): undoes the final 1/tau scaling of
 * solver.rs:397-400 and clears the termination, so that thip_solver_run goes on from the same iterate -- after
 * thip_solver_set_a_storage (finish on the exact matrix) or thip_solver_set_param (a tighter eps_acc, a larger
 * max_iter).  A no-op on a running solve. */
int thip_solver_resume(thip_solver *s);
int thip_solver_set_param(thip_solver *s, const thip_param *par);
/* solver.rs:317-320: x = work[0..n], y = work[n..n+m] */
int thip_solver_solution(thip_solver *s, float *host_x, float *host_y);
/* raw iterate (x: n+2m+1, y: n+m+1, reference layout solver.rs:349-355), for parity tests */
int thip_solver_iterate(thip_solver *s, float *host_x, float *host_y);
int thip_solver_precond(thip_solver *s, float *host_dp_tau, float *host_dp_sigma);
int thip_solver_destroy(thip_solver *s);
/* physical passes over A per iteration of the schedule in use, and bytes one pass reads */
int thip_solver_passes(const thip_solver *s, int *host_passes, size_t *host_bytes_per_pass);
/* the schedule the next thip_solver_run will execute: the one given to thip_solver_create, or THIP_SCHED_CARRIED when
 * THIP_SCHED_SWEEP was asked for and the one-pass kernel cannot take this problem (sharded rows, sparse or 16-bit A,
 * m or lda not a multiple of 4, fewer than 40 column panels per group, a device that is not 8 XCDs x 32 CUs, a failed
 * placement census) */
int thip_solver_schedule_in_use(thip_solver *s, int *host_schedule);
/* THIP_SCHED_SWEEP is considered for matrices of at least this many bytes (default 128 MiB); above it thip_solver_init
 * times the kernel's geometries on the matrix and keeps the carried schedule if none beats an estimate of its two passes
 * (measured on square-ish SOCPs: the carried schedule stays at 190 MB, the sweep is 1.37x faster at 274 MB, 1.58x at
 * 512 MB, 1.7x at 2 GB, 2.0x at 20 GB; a 1 200 x 100 000 matrix keeps the carried schedule, 100 000 x 1 200 gains 1.53x);
 * 0 = whenever the kernel can take the shape, unconditionally.  Before thip_solver_init. */
int thip_solver_set_sweep_min_bytes(thip_solver *s, size_t bytes);
/* the geometry of the one-pass kernel chosen for this solver (thip_solver_init times the candidates on the actual matrix
 * unless thip_solver_set_gemv_autotune(s, 0)): workgroups per column group, columns per panel, 16-byte slots per thread,
 * and its measured time per sweep in ms (0 = not timed); all 0 when the one-pass schedule is not in use */
int thip_solver_sweep_plan(thip_solver *s, int *host_members, int *host_cols_per_panel, int *host_slots, float *host_ms);
/* N > 1 with THIP_SCHED_SWEEP: the problem given to thip_solver_create is this rank's block of COLUMNS -- mat_a is
 * m x n_local (all m rows), vec_c its n_local entries, vec_b and the cone segments the whole problem's -- the n-vectors are
 * sharded, the m-vectors replicated and updated redundantly, and the hook of thip_solver_set_allreduce is called ONCE per
 * iteration on 2 * roundup(m, 64) + 2048 floats (the two N products and the block partials of the sums over n; one call
 * more in thip_solver_init).  Every rank must get bitwise the same sums back (RCCL, the one-shot transport and a host sum
 * all do).  thip_solver_solution / _iterate return this rank's block of x and the whole y.  Before thip_solver_init. */
int thip_solver_set_column_shard(thip_solver *s, int on);
/* Can the one-pass kernel run on this device for an m x n_local block (geometry + one placement census, no collective)?
 * A multi-rank host asks every rank and takes the minimum BEFORE building column-sharded solvers: a rank that found out
 * inside thip_solver_init would leave the others waiting in their first all-reduce.  elem = the stored form of A the run
 * will stream (THIP_A_F32 / THIP_A_BF16 / THIP_A_F16: a 16-bit plan has eight rows per slot and caps of its own).
 * A rank whose plan still fails later (a re-plan inside thip_solver_run after thip_solver_set_a_storage /
 * _set_sweep_min_bytes) takes part in the peers' collectives with its fault flag raised: every rank of the run returns
 * THIP_E_TIMEOUT at the same batch. */
int thip_sweep_probe(size_t m, size_t n_local, size_t lda, int elem, int *host_ok);

/* What THIS device streams: a bare non-temporal read of `bytes` at dev_ptr (device memory, 16-byte aligned -- e.g. the
 * solver's own A), best and average of `reps` timed launches per grid (HIP events).  bench.py prints it beside the
 * sweep's rate: the boxes of one pool differ by several percent, and a roofline fraction means little without it. */
int thip_stream_probe(const void *dev_ptr, size_t bytes, int reps, float *host_best_ms, float *host_avg_ms);

/* The one-pass kernel is persistent and every wait in it is bounded.  When a workgroup gives up (placement changed under
 * it, a peer workgroup was withheld), thip_solver_run restores the iterate of the last completed batch from a device
 * snapshot, re-arms the kernel's census and tries the batch once more; a second failure in a row hands the rest of the
 * solve to the 2-pass carried schedule (thip_solver_schedule_in_use then says THIP_SCHED_CARRIED); a
 * column-sharded run, which has no 2-pass form, raises the fault through its all-reduce so that every rank restores the
 * same iterate and retries together, and fails with THIP_E_TIMEOUT on every rank at once after two retries.
 * host_faults = recoveries so far in this solve, host_last_word = the kernel's error word of the last one (1, 2 census,
 * 3 a gather ran out of spins), host_restored_iter = the iteration the restored snapshot held (-1: none). */
int thip_solver_sweep_faults(thip_solver *s, int *host_faults, int *host_last_word, int64_t *host_restored_iter);
/* partial dots published with agent-scope (sc1) stores (1), with plain stores that stay in the group's L2 (0), or by the
 * verdict of the process's publish-scope self-test (-1, the default); between thip_solver_run calls */
int thip_solver_set_sweep_publish(thip_solver *s, int agent_scope);
/* The publish-scope self-test: once per process, before the first plan of the one-pass kernel, 200 sweeps of a scratch matrix
 * with 32 members per group, a short polling bound and the kernel's poll counters decide whether plain-store publishing is
 * visible to the gatherers on this driver / firmware (0) or every solver publishes at agent scope (1).  mode 0: the cached
 * verdict (run now if need be); 1: run again; 2: run again and count it as failed (test hook).  host_info (4 ints): the
 * kernel's error word, polls summed over all gathers, the most polls any one gather needed, sweeps run.
 * THIP_SWEEP_PUBLISH=0 / 1 in the environment pins the verdict without a test. */
int thip_sweep_publish_selftest(int mode, int *host_agent_scope, int *host_info);


/* the GEMV tiling chosen by the create-time autotune (rows groups per lane, grid size, its measured ms); 0 = heuristic */
int thip_solver_gemv_plan(const thip_solver *s, int *host_nj, int *host_blocks, float *host_ms);

/* per-launch timing of the pass over A in the fused loop (HIP events on the launch stream): enable, run, then read the
 * number of timed launches and their summed duration.  Used by bench.py's roofline.  on = 0: off; on = N >= 1: every N-th
 * launch is timed (an event pair costs the stream 3-5 us: with N = 1 that is 9 % of a 0.14 ms iteration). */
int thip_prof_enable(int on);
/* the same for the PSD cones of the fused loop: one span per iteration around the projection chains of all its PSD blocks
 * (x_y and x_s together): number of spans timed and their summed duration -- bench.py's roofline_eig */
int thip_prof_read_psd(int64_t *host_spans, double *host_total_ms);
int thip_prof_read(int64_t *host_launches, double *host_total_ms);     /* SYNC */

/* ---------------------------------------------------------------------------------------------
 * Synthetic data on the device (bench / tests): counter-based generator keyed by
 * (seed, stream, index), bit-identical to oracle/totsu_oracle.c:oc_rng_*.
 * kind 0: U[0,1) ; kind 1: approx N(0,1) (Irwin-Hall 4).  out[i] = scale * g(idx0 + i) + shift.
 * For a column-major block: index of element (r,c) is (row0 + r) + (col0 + c) * ld_index.
 * ------------------------------------------------------------------------------------------- */
int thip_gen_vector(float *out, size_t n, uint64_t seed, uint64_t stream, uint64_t idx0,
                    int kind, float scale, float shift);
int thip_gen_matrix(float *out, size_t n_row, size_t n_col, size_t lda, uint64_t seed, uint64_t stream,
                    uint64_t row0, uint64_t col0, uint64_t ld_index, int kind, float scale, float shift);
/* out(r, c) = value if row0 + r == c else 0 over an n_row x n_col block (the -I rows of the benchmark_lp matrix,
 * experimental/benchmark_lp/src/main.rs:27-40) */
int thip_gen_identity(float *out, size_t n_row, size_t n_col, size_t lda, uint64_t row0, float value);

/* Test hooks and timing probes (thip_test_*: fault injection, engine switches, the kernels alone) are declared in
 * totsu_f32hip_test.h: exported by the same library, used by tests/ and tools/, not part of the interface a binding wraps. */

#ifdef __cplusplus
}
#endif
#endif /* TOTSU_F32HIP_H */
