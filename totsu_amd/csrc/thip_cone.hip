// thip_cone.hip -- Cone::proj / Cone::product_group on device-resident vectors.
// Reference: totsu_core/src/cone_zero.rs:38-44, cone_rpos.rs:38-45 (a HOST loop over get_mut() in the
// reference: the CUDA backend copies the block D2H, clamps on the host and copies back, twice per
// iteration), cone_soc.rs:38-65, cone_rotsoc.rs:38-65, and the `group` closure of solver.rs:509-520.
//
// Second-order cones are batched: ProbSOCPCone::proj (totsu/src/problem/socp.rs:296-313) loops over the
// cones issuing get + nrm2 + scal + set per cone (>= 4 host syncs each on CUDA); here one launch projects
// all cones, one wavefront per cone (width-64 shuffle tree for ||v||^2), or one workgroup per cone when
// cones are long.
#include "thip_common.h"

using namespace thip;

namespace {

constexpr int BLK = 256;

__global__ void rpos_k(size_t n, float *__restrict__ x)
{
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK)
        x[i] = fmaxf(x[i], 0.0f);
}

// BLOCKWISE = false: 4 cones per 256-thread block, one wave each; true: one cone per block.
// Optional second vector x1 (the fused loop projects x_y and x_s in one launch: grid covers 2 * n_cones) and
// optional rx0 / rx1: after the projection rx <- rx - 2 x over the cone's rows (solver.rs:555 folded in).
template <bool BLOCKWISE>
__global__ __launch_bounds__(BLK) void soc_k(float *__restrict__ x0, float *__restrict__ x1, float *__restrict__ rx0,
                                             float *__restrict__ rx1, const int64_t *__restrict__ begs,
                                             const int64_t *__restrict__ ends, int64_t n_cones,
                                             int rotated, int64_t single_len, const int *__restrict__ stop)
{
    if (stop != nullptr && *stop != 0) return;
    __shared__ double shd[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t total = x1 ? 2 * n_cones : n_cones;
    int64_t cone = BLOCKWISE ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 4 + wave;
    if (!BLOCKWISE && cone >= total) return;     // whole wave leaves together
    float *x = x0, *rx = rx0;
    if (cone >= n_cones) { cone -= n_cones; x = x1; rx = rx1; }
    const int64_t beg = begs ? begs[cone] : 0;
    const int64_t end = ends ? ends[cone] : single_len;
    const int64_t len = end - beg;
    if (len <= 0) return;                  // uniform per group
    const int gid = BLOCKWISE ? (int)threadIdx.x : lane;
    const int gsz = BLOCKWISE ? BLK : 64;
    const float fsqrt2 = sqrtf(2.0f);

    if (rotated && len == 1) {             // cone_rotsoc.rs:46-49
        if (gid == 0) {
            const float v = fmaxf(x[beg], 0.0f);
            x[beg] = v;
            if (rx) rx[beg] = rx[beg] - 2.0f * v;
        }
        return;
    }

    float s0, v1 = 0.0f;
    if (rotated) {                         // cone_rotsoc.rs:51-54
        const float r = x[beg], s = x[beg + 1];
        s0 = (r + s) / fsqrt2;
        v1 = (r - s) / fsqrt2;
    } else {
        s0 = x[beg];
    }

    // ||v|| over x[beg+1 .. end): the reference takes LinAlg::norm (cone_soc.rs:47), an nrm2 that neither underflows nor
    // overflows on the squares -- here the squares are f64
    double acc = 0.0;
    for (int64_t i = beg + 1 + gid; i < end; i += gsz) {
        const double t = (double)((rotated && i == beg + 1) ? v1 : x[i]);
        acc += t * t;
    }
    const double sumsq = BLOCKWISE ? block_sum_d(acc, shd) : wave_sum_d(acc);
    const float norm_v = (float)sqrt(sumsq);

    // cone_soc.rs:49-61
    float f, s_new;
    if (norm_v <= -s0) { f = 0.0f; s_new = 0.0f; }
    else if (norm_v <= s0) { f = 1.0f; s_new = s0; }
    else { f = (1.0f + s0 / norm_v) / 2.0f; s_new = (norm_v + s0) / 2.0f; }

    if (!rotated) {
        if (f == 1.0f && rx == nullptr) return;
        if (gid == 0) {
            x[beg] = s_new;
            if (rx) rx[beg] = rx[beg] - 2.0f * s_new;
        }
        for (int64_t i = beg + 1 + gid; i < end; i += gsz) {
            const float nv = (f == 1.0f) ? x[i] : f * x[i];
            x[i] = nv;
            if (rx) rx[i] = rx[i] - 2.0f * nv;
        }
    } else {
        const float v1n = f * v1;
        for (int64_t i = beg + 2 + gid; i < end; i += gsz) {
            const float nv = (f == 1.0f) ? x[i] : f * x[i];
            x[i] = nv;
            if (rx) rx[i] = rx[i] - 2.0f * nv;
        }
        if (gid == 0) {                    // cone_rotsoc.rs:58-61
            const float a = (s_new + v1n) / fsqrt2, b = (s_new - v1n) / fsqrt2;
            x[beg] = a;
            x[beg + 1] = b;
            if (rx) { rx[beg] = rx[beg] - 2.0f * a; rx[beg + 1] = rx[beg + 1] - 2.0f * b; }
        }
    }
}

template <bool BLOCKWISE>
__global__ __launch_bounds__(BLK) void group_min_k(float *__restrict__ t, const int64_t *__restrict__ begs,
                                                   const int64_t *__restrict__ ends, int64_t n_groups)
{
    __shared__ float sh[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t g = BLOCKWISE ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 4 + wave;
    if (!BLOCKWISE && g >= n_groups) return;
    const int64_t beg = begs[g], end = ends[g];
    if (end <= beg) return;
    const int gid = BLOCKWISE ? (int)threadIdx.x : lane;
    const int gsz = BLOCKWISE ? BLK : 64;
    float mn = __builtin_inff();
    for (int64_t i = beg + gid; i < end; i += gsz) mn = fminf(mn, t[i]);
    mn = BLOCKWISE ? block_min(mn, sh) : wave_min(mn);
    for (int64_t i = beg + gid; i < end; i += gsz) t[i] = mn;
}

}  // namespace

namespace thip {
// internal: batched SOC with a stop flag (fused iteration)
int soc_batched2(hipStream_t st, float *x0, float *x1, float *rx0, float *rx1, const int64_t *dev_begs,
                 const int64_t *dev_ends, size_t n_cones, int rotated, size_t max_len, const int *stop)
{
    if (n_cones == 0) return 0;
    const size_t total = x1 ? 2 * n_cones : n_cones;
    if (max_len > 2048)
        hipLaunchKernelGGL(soc_k<true>, dim3((unsigned)total), dim3(BLK), 0, st, x0, x1, rx0, rx1, dev_begs, dev_ends,
                           (int64_t)n_cones, rotated, (int64_t)0, stop);
    else
        hipLaunchKernelGGL(soc_k<false>, dim3((unsigned)((total + 3) / 4)), dim3(BLK), 0, st, x0, x1, rx0, rx1, dev_begs,
                           dev_ends, (int64_t)n_cones, rotated, (int64_t)0, stop);
    THIP_LAUNCH_CHECK();
    return 0;
}

int soc_batched(hipStream_t st, float *x, const int64_t *dev_begs, const int64_t *dev_ends, size_t n_cones,
                int rotated, size_t max_len, const int *stop)
{
    return soc_batched2(st, x, nullptr, nullptr, nullptr, dev_begs, dev_ends, n_cones, rotated, max_len, stop);
}

int group_min_batched(hipStream_t st, float *t, const int64_t *dev_begs, const int64_t *dev_ends, size_t n_groups,
                      size_t max_len)
{
    if (n_groups == 0) return 0;
    if (max_len > 2048)
        hipLaunchKernelGGL(group_min_k<true>, dim3((unsigned)n_groups), dim3(BLK), 0, st, t, dev_begs, dev_ends,
                           (int64_t)n_groups);
    else
        hipLaunchKernelGGL(group_min_k<false>, dim3((unsigned)((n_groups + 3) / 4)), dim3(BLK), 0, st, t, dev_begs,
                           dev_ends, (int64_t)n_groups);
    THIP_LAUNCH_CHECK();
    return 0;
}
}  // namespace thip

extern "C" {

// The single-cone projections join the deferred record when a host has opted in (thip_set_lazy_gemv): consecutive
// projections of one kind on disjoint slices -- the per-cone loop of ProbSOCPCone::proj, socp.rs:296-313 -- run as one
// launch (thip_lazy.hip).
int thip_proj_zero(int dual_cone, size_t n, float *x)
{
    THIP_NEED_INIT_NOFLUSH();
    if (dual_cone || n == 0) return 0;               // cone_zero.rs:38-44: the dual cone is everything
    int deferred = 0;
    THIP_RC(lazy_push_proj(THIP_CONE_ZERO, n, x, &deferred));
    if (deferred) return 0;
    return thip_scale(n, 0.0f, x);
}

int thip_proj_rpos(size_t n, float *x)
{
    THIP_NEED_INIT_NOFLUSH();
    if (n == 0) return 0;
    int deferred = 0;
    THIP_RC(lazy_push_proj(THIP_CONE_RPOS, n, x, &deferred));
    if (deferred) return 0;
    hipLaunchKernelGGL(rpos_k, dim3(grid_for(n, BLK, 2048)), dim3(BLK), 0, ctx().stream, n, x);
    THIP_LAUNCH_CHECK();
    return 0;
}

static int soc_single(size_t n, float *x, int rotated)
{
    THIP_NEED_INIT_NOFLUSH();
    if (n == 0) return 0;
    int deferred = 0;
    THIP_RC(lazy_push_proj(rotated ? THIP_CONE_ROTSOC : THIP_CONE_SOC, n, x, &deferred));
    if (deferred) return 0;
    if (n > 2048)
        hipLaunchKernelGGL(soc_k<true>, dim3(1), dim3(BLK), 0, ctx().stream, x, (float *)nullptr, (float *)nullptr,
                           (float *)nullptr, (const int64_t *)nullptr, (const int64_t *)nullptr, (int64_t)1, rotated,
                           (int64_t)n, (const int *)nullptr);
    else
        hipLaunchKernelGGL(soc_k<false>, dim3(1), dim3(BLK), 0, ctx().stream, x, (float *)nullptr, (float *)nullptr,
                           (float *)nullptr, (const int64_t *)nullptr, (const int64_t *)nullptr, (int64_t)1, rotated,
                           (int64_t)n, (const int *)nullptr);
    THIP_LAUNCH_CHECK();
    return 0;
}

int thip_proj_soc(size_t n, float *x) { return soc_single(n, x, 0); }
int thip_proj_rotsoc(size_t n, float *x) { return soc_single(n, x, 1); }

int thip_proj_soc_batched(float *x, const int64_t *dev_offs, size_t n_cones, int rotated, size_t max_len)
{
    THIP_NEED_INIT();
    return soc_batched(ctx().stream, x, dev_offs, dev_offs + 1, n_cones, rotated, max_len, nullptr);
}

int thip_group_min_batched(float *dp_tau, const int64_t *dev_offs, size_t n_groups, size_t max_len)
{
    THIP_NEED_INIT();
    return group_min_batched(ctx().stream, dp_tau, dev_offs, dev_offs + 1, n_groups, max_len);
}

}  // extern "C"
