// thip_common.h -- internal helpers shared by the gfx950 kernels of libtotsu_f32hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "totsu_f32hip.h"
#include "totsu_f32hip_test.h"

namespace thip {

struct Ctx {
    bool         inited = false;
    int          device = 0;
    hipStream_t  own_stream = nullptr;
    hipStream_t  stream = nullptr;       // the stream every entry point enqueues on
    float       *scratch = nullptr;      // partial sums of the two-stage reductions / GEMV partials
    size_t       scratch_n = 0;
    float       *dev_scalar = nullptr;   // 64 floats of device scalars for SYNC calls
    float       *pinned = nullptr;       // 64 floats, host-pinned
    void        *stage = nullptr;        // pinned staging for pageable h2d: two halves of stage_bytes each
    size_t       stage_bytes = 0;
    hipEvent_t   stage_ev[2] = { nullptr, nullptr };   // "the DMA out of half k has finished"
    int          num_cu = 256;
    int         *never_stop = nullptr;  // a device int that stays 0: the stop flag of launches outside a solver loop
    float       *eig_pin = nullptr;     // pinned staging of the QL rotation record (thip_eig.hip), grown on demand
    size_t       eig_pin_floats = 0;
    hipStream_t  eig_side = nullptr;    // Q is formed here while the tridiagonal eigenproblem runs on `stream` (thip_eig.hip)
    hipEvent_t   eig_ev[2] = { nullptr, nullptr };
};

// *host_out = *dev_src, in stream order (SYNC)
int  fetch_scalar(const float *dev_src, float *host_out);

Ctx &ctx();
// deferred small GEMVs (thip_lazy.hip)
bool lazy_pending();
int  lazy_flush();
int  lazy_push(int transpose, size_t n_row, size_t n_col, float alpha, const float *mat, const float *x, float beta,
               float *y, int *deferred);
int  lazy_push_scale(size_t n, float alpha, float *x, int *deferred);
int  lazy_push_add(size_t n, float alpha, const float *x, float *y, int *deferred);
int  lazy_push_proj(int kind, size_t n, float *x, int *deferred);      // single-cone projections (THIP_CONE_*)
int  lazy_push_set(float *x, float value, int *deferred);              // SliceLike::set
// SYNC scalar reads: *served = 1 and the value when the read-ahead has it; else the record has been run and the caller reads
int  lazy_read(int is_norm, const float *p, size_t n, float *value, int *served);
void lazy_release();      // frees the queue's device memory (thip_shutdown)
void lazy_forget(uintptr_t lo, uintptr_t hi);      // drops the learnt plans that hold an address in [lo, hi) (thip_free)
int  fail(int code, const char *what, const char *file, int line);
int  need_init();
// returns a scratch buffer of at least n floats (grows with hipMalloc; not inside graph capture)
int  scratch(size_t n, float **out);

#define THIP_TRY(expr)                                                            \
    do {                                                                          \
        hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess) return ::thip::fail((int)e__, #expr, __FILE__, __LINE__); \
    } while (0)

#define THIP_RC(expr)                                                             \
    do {                                                                          \
        int rc__ = (expr);                                                        \
        if (rc__ != 0) return rc__;                                               \
    } while (0)

// every entry point first runs what thip_transform_ge has deferred (thip_lazy.hip), so that stream order == call order
#define THIP_NEED_INIT()                                                          \
    do {                                                                          \
        if (!::thip::ctx().inited) return ::thip::need_init();                    \
        if (::thip::lazy_pending()) { int rcl__ = ::thip::lazy_flush(); if (rcl__ != 0) return rcl__; } \
    } while (0)

#define THIP_NEED_INIT_NOFLUSH()                                                  \
    do {                                                                          \
        if (!::thip::ctx().inited) return ::thip::need_init();                    \
    } while (0)

#define THIP_LAUNCH_CHECK() THIP_TRY(hipGetLastError())

// ---- device helpers ---------------------------------------------------------------------

// sum over the 64 lanes of a wave; result valid in every lane
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// the same sum through the DPP network (row shifts 1, 2, 4, 8, then row_bcast 15 and 31: six VALU operations instead of six
// LDS-crossbar permutes), result read from lane 63 and broadcast.  The order of the additions differs from wave_sum's
// butterfly, so a kernel uses one or the other throughout; this one is for the latency chains of the eigen engine.
__device__ __forceinline__ float wave_sum_dpp(float v)
{
#define THIP_DPP_ADD(ctrl, rows) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rows, 0xf, true))
    THIP_DPP_ADD(0x111, 0xf);
    THIP_DPP_ADD(0x112, 0xf);
    THIP_DPP_ADD(0x114, 0xf);
    THIP_DPP_ADD(0x118, 0xf);
    THIP_DPP_ADD(0x142, 0xa);
    THIP_DPP_ADD(0x143, 0xc);
#undef THIP_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); `sh` holds >= 16 floats; result in all threads
__device__ __forceinline__ float block_sum(float v, float *sh)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float t = (lane < nw) ? sh[lane] : 0.0f;
    t = wave_sum(t);
    return t;
}

// block_sum on the DPP network (see wave_sum_dpp: another order of additions)
__device__ __forceinline__ float block_sum_dpp(float v, float *sh)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum_dpp(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return wave_sum_dpp((lane < nw) ? sh[lane] : 0.0f);
}

__device__ __forceinline__ double block_sum_d(double v, double *sh)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum_d(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double t = (lane < nw) ? sh[lane] : 0.0;
    t = wave_sum_d(t);
    return t;
}

__device__ __forceinline__ float block_min(float v, float *sh)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_min(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float t = (lane < nw) ? sh[lane] : __builtin_inff();
    t = wave_min(t);
    return t;
}

static inline unsigned grid_for(size_t n, unsigned block, unsigned max_blocks)
{
    size_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (unsigned)g;
}

// ---- internal entry points shared between translation units (device pointers, stream given) ----
// dual GEMV: optional N product (xn: n_col -> hN: n_row) and T product (xt: n_row -> gT: n_col) from
// one pass over the column-major matrix.  Results are written as finished vectors:
//   outN[r] = alphaN * (A xn)[r] + betaN * outN[r]      (betaN == 0: outN not read)
//   outT[c] = alphaT * (A^T xt)[c] + betaT * outT[c]
// abs_mode: use |A| (for absadd_*), x vectors ignored (taken as all-ones).
// a_kind: THIP_A_F32 (mat = const float *), THIP_A_BF16 or THIP_A_F16 (mat = const uint16_t *, lda in elements;
// F16 also needs inv_s, the per-column 1 / scale)
int dual_gemv(hipStream_t st, size_t n_row, size_t n_col, const void *mat, size_t lda,
              const float *xn, float alphaN, float betaN, float *outN,
              const float *xt, float alphaT, float betaT, float *outT,
              bool abs_mode, const int *stop_flag, int a_kind = 0, const float *inv_s = nullptr);
int to_bf16(hipStream_t st, size_t n_row, size_t n_col, const float *src, uint16_t *dst, size_t ld16);
// f16 with one power-of-two scale per column: inv_s[c] = 1 / s_c is written (n_col floats)
int to_f16(hipStream_t st, size_t n_row, size_t n_col, const float *src, uint16_t *dst, size_t ld16, float *inv_s);
// raw form: leaves per-chunk / per-tile partial sums in scratch and reports their geometry so that a
// consumer kernel can fold the second reduction stage into its own pass
struct GemvPartials {
    const float *partN; int nN; size_t strideN;   // outN[r] = sum_{k<nN} partN[k*strideN + r]
    const float *partT; int nT; size_t strideT;   // outT[c] = sum_{k<nT} partT[k*strideT + c]
};
// descriptor of one matrix of a grouped launch: column-major nr x nc, lda = nr; xn (nc long) / xt (nr long) = the input
// vectors of the N / T product; partN / partT = where their partial sums go; cpc = columns per chunk
struct GroupDesc { const float *A; const float *xn; const float *xt; float *partN; float *partT; int nr, nc, cpc, pad; };
// mode 0: N products, 1: T products, 2: both products of every descriptor from one read of its matrix
int grouped_gemv(hipStream_t st, const GroupDesc *dev_tab, int n_desc, int max_tiles, int max_chunks, int mode);
struct GemvHint { int nj; int target_blocks; };        // tiling override: row groups per lane, grid size
const GemvHint *gemv_candidates(int *count);           // plans worth timing on a given matrix
int dual_gemv_partials(hipStream_t st, size_t n_row, size_t n_col, const void *mat, size_t lda,
                       const float *xn, const float *xt, bool do_n, bool do_t, bool abs_mode,
                       float *scratch_base, size_t scratch_floats, GemvPartials *out, const int *stop_flag,
                       const GemvHint *hint = nullptr, int a_kind = 0, const float *inv_s = nullptr,
                       bool pad_zero = false);     // pad_zero: rows n_row .. lda - 1 of mat are zeros (the library's own copy)
// the product over columns [col0, col1) as a launch of its own, partial sums left where a consumer of the whole product
// expects them (thip_gemv.hip)
int dual_gemv_partials_cols(hipStream_t st, size_t n_row, size_t n_col, const void *mat, size_t lda,
                            const float *xn, const float *xt, bool do_n, bool do_t,
                            float *scratch_base, size_t scratch_floats, GemvPartials *out, const int *stop_flag,
                            const GemvHint *hint, int a_kind, const float *inv_s, bool pad_zero,
                            size_t col0, size_t col1, int chunk_row0, int max_chunk_rows, int *chunks_used);
int dual_gemv_chunk_rows(size_t n_row, size_t cols, bool vec_ok, int a_kind, const GemvHint *hint, int *tiles);
int dual_gemv_cols_per_chunk(size_t n_row, size_t n_col, const void *mat, size_t lda, const GemvHint *hint, int a_kind,
                             int *chunks);
int dual_gemv_partials_geometry(size_t n_row, size_t n_col, const void *mat, size_t lda, bool do_n, bool do_t,
                                float *scratch_base, GemvPartials *out);
size_t dual_gemv_scratch_floats(size_t n_row, size_t n_col);
// y[i] = alpha * sum_k part[k*stride + i] + beta * y[i]
int finalize_partials(hipStream_t st, size_t n, const float *part, int np, size_t stride, float alpha, float beta,
                      float *y, const int *stop);

// thip_sweep.hip: one pass over A per iteration (THIP_SCHED_SWEEP)
constexpr int SW_SPIN_MAX = 2000000;       // polls of a gather before a workgroup gives up (~2-4 s)
struct SweepGeom { int G, ngroups, rows_per_member, cols_per_group, nslot, npan, w, variant, m_eff; size_t mpad; int elem; };
struct SweepArgs {
    const float *A; size_t lda; int m, n;  // A: the matrix as stored (f32, or 16-bit: elem = THIP_A_BF16 / THIP_A_F16), lda in elements
    const float *inv_s;                     // f16 storage: 1 / scale per column (else NULL)
    int G, rows_per_member, cols_per_group;
    const float *v, *xy;                    // the m-vectors every column is multiplied with
    const float *c, *Su, *Tx;               // per column
    float *u, *ku;                          // updated in place (ku == NULL: plain additions)
    const float *xx_in, *kx_in;             // x_x_k (and its Kahan term) ...
    float *xx_out, *kx_out;                 // ... x_x_{k+1}
    float *gP;                              // A^T x_y of the previous sweep in, of this one out
    float *partH; size_t mpad;              // [group][2][mpad]: the groups' shares of A u_k and A x_x_{k+1}
    unsigned long long *gran;               // [group][SW_RING][G][2 W] granules
    unsigned *census;                       // [0..7] workgroups per XCD, [8] total, [9] error word
    unsigned seq, tagbase;
    int first;                              // 1: u is current (a (re)start): no u update
    int dbg;                                // experiments (THIP_SWEEP_DBG): 1 no polling, 2 no wave reduction of the dots
    const int *stop; const float *kappa_p, *rtau_p;
    // the sums over n of the criteria and of the scalar updates, accumulated by the workgroup that writes a column:
    // pn[q * pn_stride + workgroup], q = 0 ||d||^2 (d = c + A^T x_y / tau, or A^T x_y when tau <= eps_zero), 1 c.x_x_k,
    // 2 c.u_k, 3 c.(x_x_k - 2 x_x_{k+1}); every one of the 256 workgroups writes its four
    float *pn; int pn_stride; const float *tau_p; float eps_zero;
    // where the kappa update reads c.rx_x of the PREVIOUS sweep: never the buffer this launch writes (two buffers by launch
    // parity; column-sharded: the all-reduced tail) -- a fast workgroup's exit must not overtake a slow one's entry
    const float *pn_in; int pn_in_stride;
    int spin_max;                           // bound of a gather's polling loop (SW_SPIN_MAX; tests shorten it)
    int fault;                              // TEST HOOK: != 0: one workgroup withholds its publishes from the middle of the sweep on
    int pub_agent;                          // != 0: partial dots published with agent-scope (sc1) stores
    // kappa_out != NULL and first == 0: the sweep opens with the kappa update (solver.rs:566-567) -- every workgroup forms
    // kappa_k = min(*kappa_p + *skappa_p (sum pn[3][0 .. pn_count) + sum pm_brx[0 .. np_m)), 0) for itself (same inputs, same
    // order, same value) and one of them stores it; *kappa_p is then the copy sw_vm_k left of kappa_{k-1}
    float *kappa_out; const float *skappa_p; const float *pm_brx; int np_m, pn_count;
};
// how the groups' partial dots are published by default in this process: 0 plain stores (the publish-scope self-test of
// thip_sweep.hip passed), 1 agent scope.  Runs the self-test on first use (allocates and synchronises: plan time only)
int sweep_publish_default();
int sweep_plan(size_t m, size_t n, size_t lda, const void *mat, SweepGeom *g, int elem = 0);
int sweep_candidates(size_t m, size_t n, size_t lda, const void *mat, SweepGeom *out, int max_out, int elem = 0);
int sweep_launch16(hipStream_t st, const SweepGeom &g, const SweepArgs &a);       // thip_sweep16.hip
size_t sweep_gran_words(const SweepGeom &g);
int sweep_census_dry_run(hipStream_t st, unsigned *census, unsigned seq);
int sweep_launch(hipStream_t st, const SweepGeom &g, const SweepArgs &a);

// thip_sptile.hip: a sparse operator stored once in 4096 x 4096 tiles; both products scatter into LDS accumulators and leave the
// slices' shares part[slice][2][pad] (pad = the out-vector's length rounded to 64)
size_t sptile_part_floats(const thip_sptile *M, bool tphase);
int sptile_slices(const thip_sptile *M, bool tphase);
size_t sptile_pad(const thip_sptile *M, bool tphase);
size_t sptile_bytes_per_pass(const thip_sptile *M);
void sptile_dims(const thip_sptile *M, size_t *m, size_t *n, size_t *nnz);
int sptile_product(hipStream_t st, const thip_sptile *M, bool tphase, const float *in0, const float *in1, float *part,
                   int abs_mode, const int *stop, bool xmax_ready = false);
int sptile_colupdate(hipStream_t st, const thip_sptile *M, const SweepArgs &a, const float *partT);

// thip_oneshot.hip: the hook thip_solver_use_oneshot installs and the device address of its error word (NULL: not set up)
thip_allreduce_fn oneshot_hook();
const unsigned *oneshot_error_word();

void prof_release();      // destroys the HIP events of thip_prof_* (thip_shutdown)

int reduce_to_dev(hipStream_t st, int op, size_t n, const float *x, const float *y, size_t incx, float *dev_out);
enum { RED_SUMSQ_SQRT = 0, RED_ABSSUM = 1, RED_DOT = 2 };

int soc_batched(hipStream_t st, float *x, const int64_t *dev_begs, const int64_t *dev_ends, size_t n_cones,
                int rotated, size_t max_len, const int *stop);
int soc_batched2(hipStream_t st, float *x0, float *x1, float *rx0, float *rx1, const int64_t *dev_begs,
                 const int64_t *dev_ends, size_t n_cones, int rotated, size_t max_len, const int *stop);
int group_min_batched(hipStream_t st, float *t, const int64_t *dev_begs, const int64_t *dev_ends, size_t n_groups,
                      size_t max_len);

// nbatch matrices per call: packed + z * pstride each, work holds nbatch * thip_map_eig_worklen(n) floats
int eig_psd_project(hipStream_t st, size_t n, float *packed, int has_scale, float scale_diag, float eps_zero,
                    float *work, size_t worklen, int map_kind, const int *stop, int nbatch = 1, ptrdiff_t pstride = 0,
                    float *rx = nullptr, ptrdiff_t rx_stride = 0);
// rx != nullptr (only where psd_project_takes_rx(n)): also rx <- rx - 2 x on the projected entries
bool psd_project_takes_rx(size_t n);
// PSD projection of `count` matrices of the SAME order n <= thip_psd_small_max() in one launch: matrix (i, z) is at
// base + dev_offs[i] + z * pstride, z < nbatch (the x_y and x_s blocks of cone i)
// rx != nullptr: also rx <- rx - 2 x on the projected entries (rx + dev_offs[i] + z * rx_stride)
int eig_psd_project_small(hipStream_t st, size_t n, float *base, const int64_t *dev_offs, int count, int has_scale,
                          float scale_diag, const int *stop, int nbatch, ptrdiff_t pstride, float *rx = nullptr,
                          ptrdiff_t rx_stride = 0);
size_t psd_small_max();

// thip_trieig.hip: eigenvalues (multisection) and eigenvectors (twisted factorisation) of a symmetric tridiagonal matrix
// on the device, f64 inside; the caller certifies orthogonality / residuals (thip_eig.hip decompose_tridiag)
size_t tri_eigen_scratch_floats(int n);
int tri_eigen(hipStream_t st, int n, int ld, const float *d, const float *e, float *w32, float *V0, unsigned *cert_bits,
              float *scr);
int tri_orth_partials(hipStream_t st, int n, int ld, const float *P, float *part, int nblocks);
int tri_map(hipStream_t st, int n, int ld, int map_kind, const float *w, float *e);

// counter-based generator, identical integer function to oracle/totsu_oracle.c:oc_rng_hash
__host__ __device__ __forceinline__ uint64_t rng_hash(uint64_t seed, uint64_t stream, uint64_t idx)
{
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + stream * 0xD1B54A32D192ED03ull + idx * 0xBF58476D1CE4E5B9ull
                 + 0x94D049BB133111EBull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return z;
}
__host__ __device__ __forceinline__ float rng_uniform(uint64_t seed, uint64_t stream, uint64_t idx)
{
    return (float)(rng_hash(seed, stream, idx) >> 40) * (1.0f / 16777216.0f);
}
__host__ __device__ __forceinline__ float rng_normal(uint64_t seed, uint64_t stream, uint64_t idx)
{
    const uint64_t h = rng_hash(seed, stream, idx);
    const uint32_t s = (uint32_t)(h & 0xFFFF) + (uint32_t)((h >> 16) & 0xFFFF)
                     + (uint32_t)((h >> 32) & 0xFFFF) + (uint32_t)((h >> 48) & 0xFFFF);
    return ((float)((int32_t)s - 131070) * (1.0f / 65536.0f)) * 1.7320508f;
}

}  // namespace thip
