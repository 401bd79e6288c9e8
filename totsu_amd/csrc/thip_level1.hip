// thip_level1.hip -- LinAlg vector primitives (totsu_core/src/solver/linalg.rs:10-68) as gfx950 kernels.
// Semantic spec: totsu_core/src/floatgeneric.rs:16-84; CUDA call sites: totsu_f32cuda/src/f32cuda.rs:27-136.
//
// All of these are O(len) over vectors that are <0.2 % of an iteration's bytes: they are
// written for correctness at any alignment (the solver hands out sub-slices at offsets n, n+m, n+2m,
// solver.rs:116-118) and low launch count, not for bandwidth records.
#include "thip_common.h"

using namespace thip;

namespace {

constexpr int BLK = 256;
constexpr unsigned MAXB = 2048;

__global__ void copy_k(size_t n, const float *__restrict__ x, float *__restrict__ y)
{
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK) y[i] = x[i];
}
__global__ void scale_k(size_t n, float a, float *__restrict__ x)
{
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK) x[i] = a * x[i];
}
__global__ void add_k(size_t n, float a, const float *__restrict__ x, float *__restrict__ y)
{
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK) y[i] = y[i] + a * x[i];
}
__global__ void adds_k(size_t n, float s, float *__restrict__ y)
{
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK) y[i] = y[i] + s;
}
__global__ void transform_di_k(size_t n, float a, const float *__restrict__ d, const float *__restrict__ x,
                               float b, float *__restrict__ y)
{
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK) {
        const float t = a * d[i] * x[i];
        y[i] = (b == 0.0f) ? t : t + b * y[i];
    }
}
__global__ void recip_max_k(size_t n, float eps, float *__restrict__ x)
{
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK)
        x[i] = 1.0f / fmaxf(x[i], eps);
}

// stage 1 of the deterministic two-stage reductions: one partial per block
template <int OP>
__global__ void reduce1_k(size_t n, const float *__restrict__ x, const float *__restrict__ y, size_t incx,
                          float *__restrict__ part)
{
    if constexpr (OP == RED_SUMSQ_SQRT) {
        // LinAlg::norm is scale invariant in the reference (nrm2 of BLAS / cuBLAS, f64lapack.rs, f32cuda.rs): the squares are
        // accumulated in f64 and a block hands on its 2-NORM, which is an f32 number whenever the entries are (the squares
        // of a vector of 1e-25s are not)
        __shared__ double shd[16];
        double acc = 0.0;
        for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK) {
            const double v = (double)x[i];
            acc += v * v;
        }
        acc = block_sum_d(acc, shd);
        if (threadIdx.x == 0) part[blockIdx.x] = (float)sqrt(acc);
    } else {
        __shared__ float sh[16];
        float acc = 0.0f;
        for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK) {
            if (OP == RED_ABSSUM) { acc += fabsf(x[i * incx]); }
            else { acc += x[i] * y[i]; }
        }
        acc = block_sum(acc, sh);
        if (threadIdx.x == 0) part[blockIdx.x] = acc;
    }
}

// stage 2: one block, double accumulation of the (<= MAXB) partials
template <int OP>
__global__ void reduce2_k(int np, const float *__restrict__ part, float *__restrict__ out)
{
    __shared__ double shd[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < np; i += BLK) acc += (OP == RED_SUMSQ_SQRT) ? (double)part[i] * (double)part[i] : (double)part[i];
    acc = block_sum_d(acc, shd);
    if (threadIdx.x == 0) out[0] = (OP == RED_SUMSQ_SQRT) ? (float)sqrt(acc) : (float)acc;
}

__global__ void gen_vec_k(float *out, size_t n, uint64_t seed, uint64_t stream, uint64_t idx0, int kind,
                          float scale, float shift)
{
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK) {
        const float g = kind ? rng_normal(seed, stream, idx0 + i) : rng_uniform(seed, stream, idx0 + i);
        out[i] = scale * g + shift;
    }
}

__global__ void gen_mat_k(float *out, size_t n_row, size_t n_col, size_t lda, uint64_t seed, uint64_t stream,
                          uint64_t row0, uint64_t col0, uint64_t ld_index, int kind, float scale, float shift)
{
    // one block column-strip at a time: consecutive threads -> consecutive rows (coalesced stores)
    for (size_t c = blockIdx.y; c < n_col; c += gridDim.y) {
        const uint64_t cbase = (col0 + c) * ld_index + row0;
        for (size_t r = blockIdx.x * (size_t)BLK + threadIdx.x; r < n_row; r += (size_t)gridDim.x * BLK) {
            const float g = kind ? rng_normal(seed, stream, cbase + r) : rng_uniform(seed, stream, cbase + r);
            out[c * lda + r] = scale * g + shift;
        }
    }
}

__global__ void gen_ident_k(float *out, size_t n_row, size_t n_col, size_t lda, uint64_t row0, float value)
{
    for (size_t c = blockIdx.y; c < n_col; c += gridDim.y)
        for (size_t r = blockIdx.x * (size_t)BLK + threadIdx.x; r < n_row; r += (size_t)gridDim.x * BLK)
            out[c * lda + r] = (row0 + r == c) ? value : 0.0f;
}

}  // namespace

namespace thip {

int reduce_to_dev(hipStream_t st, int op, size_t n, const float *x, const float *y, size_t incx, float *dev_out)
{
    if (n == 0) {
        THIP_TRY(hipMemsetAsync(dev_out, 0, sizeof(float), st));
        return 0;
    }
    const unsigned g = grid_for(n, BLK * 4, 1024);
    float *part = nullptr;
    THIP_RC(scratch(1024, &part));
    switch (op) {
    case RED_SUMSQ_SQRT:
        hipLaunchKernelGGL(reduce1_k<RED_SUMSQ_SQRT>, dim3(g), dim3(BLK), 0, st, n, x, y, incx, part);
        hipLaunchKernelGGL(reduce2_k<RED_SUMSQ_SQRT>, dim3(1), dim3(BLK), 0, st, (int)g, part, dev_out);
        break;
    case RED_ABSSUM:
        hipLaunchKernelGGL(reduce1_k<RED_ABSSUM>, dim3(g), dim3(BLK), 0, st, n, x, y, incx, part);
        hipLaunchKernelGGL(reduce2_k<RED_ABSSUM>, dim3(1), dim3(BLK), 0, st, (int)g, part, dev_out);
        break;
    default:
        hipLaunchKernelGGL(reduce1_k<RED_DOT>, dim3(g), dim3(BLK), 0, st, n, x, y, incx, part);
        hipLaunchKernelGGL(reduce2_k<RED_DOT>, dim3(1), dim3(BLK), 0, st, (int)g, part, dev_out);
        break;
    }
    THIP_LAUNCH_CHECK();
    return 0;
}

}  // namespace thip

static int host_scalar(float *host_out)
{
    return fetch_scalar(ctx().dev_scalar, host_out);
}

extern "C" {

int thip_copy(size_t n, const float *x, float *y)
{
    THIP_NEED_INIT();
    if (n == 0) return 0;
    hipLaunchKernelGGL(copy_k, dim3(grid_for(n, BLK, MAXB)), dim3(BLK), 0, ctx().stream, n, x, y);
    THIP_LAUNCH_CHECK();
    return 0;
}

int thip_scale(size_t n, float alpha, float *x)
{
    THIP_NEED_INIT_NOFLUSH();
    if (n == 0) return 0;
    // short vectors join the deferred record (thip_lazy.hip: ProbSOCPOpB scales and adds one number per cone,
    // socp.rs:194-246); anything else runs now, after what is pending
    int deferred = 0;
    THIP_RC(lazy_push_scale(n, alpha, x, &deferred));
    if (deferred) return 0;
    if (alpha == 0.0f) {
        // exact zero fill (also clears NaN/Inf of an uninitialised work buffer; solver.rs:490-491)
        THIP_TRY(hipMemsetAsync(x, 0, n * sizeof(float), ctx().stream));
        return 0;
    }
    if (alpha == 1.0f) return 0;
    hipLaunchKernelGGL(scale_k, dim3(grid_for(n, BLK, MAXB)), dim3(BLK), 0, ctx().stream, n, alpha, x);
    THIP_LAUNCH_CHECK();
    return 0;
}

int thip_add(size_t n, float alpha, const float *x, float *y)
{
    THIP_NEED_INIT_NOFLUSH();
    if (n == 0) return 0;
    int deferred = 0;
    THIP_RC(lazy_push_add(n, alpha, x, y, &deferred));
    if (deferred) return 0;
    hipLaunchKernelGGL(add_k, dim3(grid_for(n, BLK, MAXB)), dim3(BLK), 0, ctx().stream, n, alpha, x, y);
    THIP_LAUNCH_CHECK();
    return 0;
}

int thip_adds(size_t n, float s, float *y)
{
    THIP_NEED_INIT();
    if (n == 0) return 0;
    hipLaunchKernelGGL(adds_k, dim3(grid_for(n, BLK, MAXB)), dim3(BLK), 0, ctx().stream, n, s, y);
    THIP_LAUNCH_CHECK();
    return 0;
}

int thip_transform_di(size_t n, float alpha, const float *d, const float *x, float beta, float *y)
{
    THIP_NEED_INIT();
    if (n == 0) return 0;
    hipLaunchKernelGGL(transform_di_k, dim3(grid_for(n, BLK, MAXB)), dim3(BLK), 0, ctx().stream, n, alpha, d, x, beta, y);
    THIP_LAUNCH_CHECK();
    return 0;
}

int thip_recip_max(size_t n, float eps_zero, float *x)
{
    THIP_NEED_INIT();
    if (n == 0) return 0;
    hipLaunchKernelGGL(recip_max_k, dim3(grid_for(n, BLK, MAXB)), dim3(BLK), 0, ctx().stream, n, eps_zero, x);
    THIP_LAUNCH_CHECK();
    return 0;
}

namespace {
// dst(r0 + r, c) = sign * src(r, c) for an n_row x n_col column-major block (lda = n_row) written into the rows of a
// taller column-major matrix (leading dimension ld_dst); transposed: src is an n_col-vector used as ONE row
__global__ void copy_block_k(int transposed, size_t n_row, size_t n_col, float sign, const float *__restrict__ src,
                             float *__restrict__ dst, size_t ld_dst)
{
    const size_t tot = n_row * n_col;
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < tot; i += (size_t)gridDim.x * BLK) {
        const size_t c = i / n_row, r = i - c * n_row;
        dst[c * ld_dst + r] = sign * (transposed ? src[c] : src[i]);
    }
}
}  // namespace

int thip_copy_block(int transposed, size_t n_row, size_t n_col, float sign, const float *src, float *dst, size_t ld_dst)
{
    THIP_NEED_INIT();
    if (transposed) n_row = 1;
    if (n_row == 0 || n_col == 0) return 0;
    if (ld_dst < n_row) return fail(THIP_E_INVALID, "ld_dst < n_row", __FILE__, __LINE__);
    hipLaunchKernelGGL(copy_block_k, dim3(grid_for(n_row * n_col, BLK, 16384)), dim3(BLK), 0, ctx().stream, transposed,
                       n_row, n_col, sign, src, dst, ld_dst);
    THIP_LAUNCH_CHECK();
    return 0;
}

int thip_norm_dev(size_t n, const float *x, float *dev_out)
{
    THIP_NEED_INIT();
    return reduce_to_dev(ctx().stream, RED_SUMSQ_SQRT, n, x, nullptr, 1, dev_out);
}

int thip_dot_dev(size_t n, const float *x, const float *y, float *dev_out)
{
    THIP_NEED_INIT();
    return reduce_to_dev(ctx().stream, RED_DOT, n, x, y, 1, dev_out);
}

int thip_abssum_dev(size_t len, const float *x, size_t incx, float *dev_out)
{
    THIP_NEED_INIT();
    // floatgeneric.rs:62-74: chunks(incx) -> ceil(len/incx) terms, incx == 0 -> 0
    const size_t cnt = incx ? (len + incx - 1) / incx : 0;
    return reduce_to_dev(ctx().stream, RED_ABSSUM, cnt, x, nullptr, incx, dev_out);
}

int thip_norm(size_t n, const float *x, float *host_out)
{
    THIP_NEED_INIT_NOFLUSH();
    if (n == 0) { *host_out = 0.0f; return 0; }
    int served = 0;
    THIP_RC(lazy_read(1, x, n, host_out, &served));       // runs what is recorded; may have the value already
    if (served) return 0;
    THIP_RC(reduce_to_dev(ctx().stream, RED_SUMSQ_SQRT, n, x, nullptr, 1, ctx().dev_scalar));
    return host_scalar(host_out);
}

int thip_abssum(size_t len, const float *x, size_t incx, float *host_out)
{
    THIP_NEED_INIT();
    THIP_RC(thip_abssum_dev(len, x, incx, ctx().dev_scalar));
    return host_scalar(host_out);
}

int thip_gen_vector(float *out, size_t n, uint64_t seed, uint64_t stream, uint64_t idx0, int kind,
                    float scale, float shift)
{
    THIP_NEED_INIT();
    if (n == 0) return 0;
    hipLaunchKernelGGL(gen_vec_k, dim3(grid_for(n, BLK, MAXB)), dim3(BLK), 0, ctx().stream, out, n, seed, stream,
                       idx0, kind, scale, shift);
    THIP_LAUNCH_CHECK();
    return 0;
}

int thip_gen_identity(float *out, size_t n_row, size_t n_col, size_t lda, uint64_t row0, float value)
{
    THIP_NEED_INIT();
    if (n_row == 0 || n_col == 0) return 0;
    dim3 g(grid_for(n_row, BLK, 64), (unsigned)(n_col < 4096 ? n_col : 4096));
    hipLaunchKernelGGL(gen_ident_k, g, dim3(BLK), 0, ctx().stream, out, n_row, n_col, lda, row0, value);
    THIP_LAUNCH_CHECK();
    return 0;
}

int thip_gen_matrix(float *out, size_t n_row, size_t n_col, size_t lda, uint64_t seed, uint64_t stream,
                    uint64_t row0, uint64_t col0, uint64_t ld_index, int kind, float scale, float shift)
{
    THIP_NEED_INIT();
    if (n_row == 0 || n_col == 0) return 0;
    dim3 g(grid_for(n_row, BLK, 64), (unsigned)(n_col < 4096 ? n_col : 4096));
    hipLaunchKernelGGL(gen_mat_k, g, dim3(BLK), 0, ctx().stream, out, n_row, n_col, lda, seed, stream, row0, col0,
                       ld_index, kind, scale, shift);
    THIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
