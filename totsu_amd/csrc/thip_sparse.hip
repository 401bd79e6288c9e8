// thip_sparse.hip -- CSR matrix-vector product for sparse `Operator`s (the reference's example
// matrices -- l1reg_lp, partitioning_sdp, toruscompl_socp -- are mostly zeros; user-defined operators follow the
// pattern of examples/imgnr_udef/src/prob_op_a.rs).  y = alpha * A x + beta * y with A in CSR; the transposed product
// is the same kernel on the CSR of A^T (kept by the host-side SparseMatOp), so both directions are deterministic
// gathers -- no float atomics.
// Row-adaptive "CSR-vector": a group of G = 2^k lanes (k chosen from the mean row length) owns one row, its non-zeros
// are read coalesced, the group is reduced with a shuffle tree.
#include "thip_common.h"

using namespace thip;

namespace {

constexpr int BLK = 256;

template <int G>
__global__ __launch_bounds__(BLK) void spmv_csr_k(int64_t n_row, const int64_t *__restrict__ rowptr,
                                                  const int32_t *__restrict__ colidx, const float *__restrict__ vals,
                                                  float alpha, const float *__restrict__ x, float beta,
                                                  float *__restrict__ y, int abs_mode)
{
    const int lane = threadIdx.x & (G - 1);
    const int64_t groups_per_block = BLK / G;
    for (int64_t r = blockIdx.x * groups_per_block + threadIdx.x / G; r < n_row; r += (int64_t)gridDim.x * groups_per_block) {
        const int64_t b = rowptr[r], e = rowptr[r + 1];
        float s = 0.0f;
        for (int64_t k = b + lane; k < e; k += G) {
            const float a = abs_mode ? fabsf(vals[k]) : vals[k];
            s = fmaf(a, abs_mode ? 1.0f : x[colidx[k]], s);
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) y[r] = (beta == 0.0f) ? alpha * s : alpha * s + beta * y[r];
    }
}

}  // namespace

extern "C" {

int thip_spmv_csr(size_t n_row, size_t n_col, size_t nnz, const int64_t *dev_rowptr, const int32_t *dev_colidx,
                  const float *vals, float alpha, const float *x, float beta, float *y, int abs_mode)
{
    THIP_NEED_INIT();
    (void)n_col;
    if (n_row == 0) return 0;
    hipStream_t st = ctx().stream;
    const double mean = (double)nnz / (double)n_row;
#define THIP_SPMV(G)                                                                                              \
    hipLaunchKernelGGL(spmv_csr_k<G>, dim3(grid_for(n_row, BLK / G, 8192)), dim3(BLK), 0, st, (int64_t)n_row,      \
                       dev_rowptr, dev_colidx, vals, alpha, x, beta, y, abs_mode)
    if (mean <= 2.0) THIP_SPMV(2);
    else if (mean <= 6.0) THIP_SPMV(4);
    else if (mean <= 12.0) THIP_SPMV(8);
    else if (mean <= 24.0) THIP_SPMV(16);
    else if (mean <= 48.0) THIP_SPMV(32);
    else THIP_SPMV(64);
#undef THIP_SPMV
    THIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
