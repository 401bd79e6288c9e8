// thip_solver.hip -- the conic iteration resident on the GPU.
//
// Native restatement of totsu_core/src/solver/solver.rs:340-657 (SolverCore::{solve, calc_norms, init_vecs,
// calc_precond, update_vecs, criteria_conv, criteria_inf}) and of SelfDualEmbed::{op, trans_op, abssum}
// (solver.rs:109-183) for operators that are dense column-major matrices (MatOp, matop.rs) and a product
// cone of zero / nonneg / second-order / rotated second-order / PSD blocks (cone_*.rs).
//
// Differences from the reference are of SCHEDULE only:
//   * state vectors, tau, kappa, the dots and norms and the termination test stay on the device; the host
//     enqueues iterations back to back and polls a status word (the reference reads >= 8 scalars per
//     iteration through SliceLike::get, solver.rs:551-567,599-608).  Once the device has decided to stop,
//     every later kernel returns immediately, so the result is that of stopping at exactly that iteration;
//   * THIP_SCHED_FUSED: the A x and A^T y products of one stage come from one read of A (dual GEMV);
//   * THIP_SCHED_CARRIED: K*rx is obtained from the criteria products by linearity, rx = x_k - 2 x_{k+1}
//     (solver.rs:555) => A rx_x = (A x_k) - 2 (A x_{k+1}); both right-hand products are recomputed from the
//     iterate every iteration (no recursion, no drift);
//   * THIP_SCHED_REFERENCE issues the reference's six single GEMVs.
// Row-sharded A (one process per GPU): every m-length vector is sharded like the rows, every n-length
// vector is replicated; the only exchange is a sum-all-reduce of the n-vector A_g^T y_g (with the sharded
// scalars riding in its tail) per transposed product.
#include "thip_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

using namespace thip;

namespace {

constexpr int BLK = 256;
constexpr unsigned NPS = 256;    // sharded run: post_k's grid = number of block partials per sharded sum
constexpr int TAIL = 4 * NPS;    // block partials of up to 3 sharded sums riding behind an n-vector through the all-reduce
constexpr unsigned EG = 512;     // max blocks of the elementwise kernels (block partials per quantity)
constexpr unsigned PG = 4096;    // max blocks of post_k (64 elements per block)

// device status block (copied whole to the host when polling)
struct DevStatus {
    int       stop;              // != 0: every kernel returns at entry
    int       state;             // THIP_ST_*
    int       kind;
    int       xbuf;              // sweep schedule: which of the two x_x buffers holds the iterate the device stopped at
    long long iter;              // index of the iteration being / last executed
    float     cri[3];
    float     tau, kappa;
    float     norm_b, norm_c;
    float     t_tau, s_kappa;    // preconditioner entries of tau / kappa
    float     r_tau;             // rx_tau
    float     kappa_in;          // sweep schedule: kappa_{k-1} as sw_vm_k left it for the sweep that forms kappa_k
    float     tau_next, r_tau_next;   // sweep schedule: tau_{k+1} and rx_tau as the termination test of iterate k left them for the
                                 // next step's m-kernel (which commits them): no block of that kernel reads what another writes
    float     tau_r[2]; long long iter_r[2];   // sweep schedule with the termination test folded into the next step's m-kernel: tau
                                 // and the iteration index in two copies by step parity -- every block of that kernel reads one
                                 // copy, its block 0 writes the other (and tau / iter themselves, which no block of it reads)
    int       fault;             // column-sharded sweep: some rank's one-pass kernel gave up (seen by every rank in the same
                                 // all-reduce, thip_solver_run restores the snapshot on all of them together)
};

// ---------------------------------------------------------------------------------------------------
// kernels.  One iteration of the carried schedule is 7 launches (8 with block cones):
//   gemv, post, [all-reduce], xupdate, soc | gemv, post, [all-reduce], ycrit, status
// The stage's dots are left as block partials by post_k and summed by their consumers (block 0 / the status block).
// In a row-sharded run the block partials of the SHARDED sums (b.v, ||p||^2, b.x_y, b.rx_y) are written straight into
// the tail of the n-vector that is all-reduced (the sum over ranks of block partials is the block partials of the
// global sum), so the sharded path has the single-GPU launch count.  With overlap on, xupdate / ycrit run as two
// launches: the m-part (local rows: needs no collective) while the all-reduce is in flight on the side stream, the
// n-part after it.  The final 1/tau scaling is not a per-iteration launch: the host applies it once when it sees the
// terminated state.
// ---------------------------------------------------------------------------------------------------

// sum over k = k0, k0 + 4, k0 + 8, .. < np of p[k * stride] in f64 with eight independent loads in flight per lane:
// the second reduction stage is a latency chain over ~100-200 partials per element, not a bandwidth problem
__device__ __forceinline__ double sum_partials4(const float *__restrict__ p, size_t stride, int k0, int np)
{
    double s[8] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
    int k = k0;
    for (; k + 28 < np; k += 32) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = p[(size_t)(k + 4 * u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += (double)a[u];
    }
    for (; k < np; k += 4) s[0] += (double)p[(size_t)k * stride];
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// After a dual GEMV: second reduction stage of both products + the stage's sharded / replicated reductions.
//   g[i] = sum_k partT[k][i] (n) ; h[i] = sum_k partN[k][i] (m)
//   q0 = dn_a . dn_b over n (optional) ; q1 = dm_a . dm_b over m (optional)
//   crit != 0: q2 = ||p||^2, q3 = b . x_y with
//      tau > eps_zero: p = x_s/tau - b + h/tau (solver.rs:592-594) ; else p = x_s + h (solver.rs:631-632)
// block partials: q0 (a sum over the replicated n-vectors) -> part_rep[blockIdx.x];
//   q1..q3 (sums over the local rows) -> part_sh[q * gridDim.x + blockIdx.x] (part_sh = the all-reduce tail when sharded)
__global__ __launch_bounds__(BLK) void post_k(int n, int m,
                                             const float *__restrict__ partT, int nT, size_t strideT, float *__restrict__ g,
                                             const float *__restrict__ partN, int nN, size_t strideN, float *__restrict__ h,
                                             const float *__restrict__ dn_a, const float *__restrict__ dn_b,
                                             const float *__restrict__ dm_a, const float *__restrict__ dm_b,
                                             int crit, const float *__restrict__ xs, const float *__restrict__ xy,
                                             const float *__restrict__ b, float eps_zero,
                                             float *__restrict__ part_rep, float *__restrict__ part_sh,
                                             const DevStatus *st, int do_m)
{
    if (st->stop != 0) return;
    // do_m == 0 (column-split runs, first half): the n-part only -- g over the given column range and its q0 partial;
    // the m-part and q1 .. q3 belong to the launch that follows the last column range
    // 256 threads = 64 elements x 4 partial-index lanes: the sum over the ~100 partials of one element is split
    // four ways (4x the loads in flight), combined through LDS, and lane 0 of each element does the epilogue
    __shared__ float sh[16];
    __shared__ double comb[3][64];       // the second reduction stage accumulates in f64 (free here; see DESIGN.md 5, f32 floor)
    const int e = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const size_t gstride = (size_t)gridDim.x * 64;
    float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f, q3 = 0.0f;
    const float tau = st->tau;
    const bool conv = tau > eps_zero;
    const float rt = conv ? 1.0f / tau : 1.0f;

    for (size_t i0 = blockIdx.x * (size_t)64; i0 < (size_t)n; i0 += gstride) {
        const size_t i = i0 + e;
        double sd = 0.0;
        if (i < (size_t)n) sd = sum_partials4(partT + i, strideT, kq, nT);
        if (kq > 0) comb[kq - 1][e] = sd;
        __syncthreads();
        if (kq == 0 && i < (size_t)n) {
            float s;
            if (nT >= 0) { s = (float)((sd + comb[0][e]) + (comb[1][e] + comb[2][e])); g[i] = s; }
            else s = g[i];                    // nT < 0: g already holds the finished product (sparse path)
            if (dn_a) q0 = fmaf(dn_a[i], dn_b[i], q0);
        }
        __syncthreads();
    }
    if (do_m)
    for (size_t i0 = blockIdx.x * (size_t)64; i0 < (size_t)m; i0 += gstride) {
        const size_t i = i0 + e;
        double sd = 0.0;
        if (i < (size_t)m) sd = sum_partials4(partN + i, strideN, kq, nN);
        if (kq > 0) comb[kq - 1][e] = sd;
        __syncthreads();
        if (kq == 0 && i < (size_t)m) {
            float s;
            if (nN >= 0) { s = (float)((sd + comb[0][e]) + (comb[1][e] + comb[2][e])); h[i] = s; }
            else s = h[i];
            if (dm_a) q1 = fmaf(dm_a[i], dm_b[i], q1);
            if (crit) {
                const float bi = b[i];
                float p;
                if (conv) { p = xs[i] * rt - bi; p = fmaf(rt, s, p); }
                else p = xs[i] + s;
                q2 = fmaf(p, p, q2);
                q3 = fmaf(bi, xy[i], q3);
            }
        }
        __syncthreads();
    }
    q0 = block_sum(q0, sh);
    if (threadIdx.x == 0) part_rep[blockIdx.x] = q0;
    if (!do_m) return;
    q1 = block_sum(q1, sh);
    if (threadIdx.x == 0) part_sh[gridDim.x + blockIdx.x] = q1;
    if (crit) {
        q2 = block_sum(q2, sh); q3 = block_sum(q3, sh);
        if (threadIdx.x == 0) { part_sh[2 * gridDim.x + blockIdx.x] = q2; part_sh[3 * gridDim.x + blockIdx.x] = q3; }
    }
}

// sum of np block partials by one whole block (f64 accumulation); the result is valid in every thread.
// shd: 16 doubles of LDS.
__device__ __forceinline__ float block_sum_of_partials(const float *part, int np, double *shd)
{
    double acc = 0.0;
    for (int k = threadIdx.x; k < np; k += BLK) acc += (double)part[k];
    acc = block_sum_d(acc, shd);
    __syncthreads();
    return (float)acc;
}

// x + inc, optionally compensated: k[i] carries the rounding error of the previous additions into this entry (Kahan).
// The f32 iterate otherwise stops moving once an update is below half an ulp of the entry while the dual residual
// is still ~1e-5..1e-4 (a floor the reference's f32 arithmetic has too; numpy emulation: 6.3e-6 -> 1e-7 at n = 200).
__device__ __forceinline__ float comp_add(float x, float inc, float *__restrict__ k, size_t i)
{
    if (k == nullptr) return x + inc;
    const float y = inc - k[i];
    const float t = x + y;
    k[i] = (t - x) - y;
    return t;
}

// x-update, solver.rs:538-555 (everything except the block cones):
//   x += T o tx with tx = -K^T y (SelfDualEmbed::trans_op, solver.rs:133-157):
//     x_x += Tx o ( gT + c kappa)        gT = A^T v   (after the all-reduce)
//     x_y += Ty o (-hN + b kappa)        hN = A u
//     x_s += Ts o ( v )
//     tau += Ttau (-c.u - b.v) ; tau <- max(tau, 0)                       (solver.rs:551-552)
//   element-wise cones folded in: cls 0 = zero cone (dual: identity, primal: 0; cone_zero.rs:38-44),
//   cls 1 = nonneg (max(.,0) both; cone_rpos.rs:38-45), cls >= 2 = member of a block cone (projected by the
//   next launch, which also finishes rx for those rows);
//   rx = x_k - 2 x_{k+1} (solver.rs:538,555) for x_x, tau and the cls 0/1 rows; rx = x_k for block-cone rows.
__global__ __launch_bounds__(BLK) void xupdate_k(int n, int m, const float *__restrict__ gT, const float *__restrict__ hN,
                                                const float *__restrict__ c, const float *__restrict__ b,
                                                const float *__restrict__ v, const float *__restrict__ Tx,
                                                const float *__restrict__ Ty, const float *__restrict__ Ts,
                                                const unsigned char *__restrict__ cls,
                                                float *__restrict__ xx, float *__restrict__ xy, float *__restrict__ xs,
                                                float *__restrict__ rxx, float *__restrict__ rxy, float *__restrict__ rxs,
                                                DevStatus *st, const float *ps_c, int np_c, const float *ps_b, int np_b,
                                                int do_n, int do_m, int do_tau,
                                                float *__restrict__ kx, float *__restrict__ ky, float *__restrict__ ks)
{
    if (st->stop != 0) return;
    // do_tau: the tau update (with do_n in one launch; column-split runs give the x_x rows as two ranges -- n and the
    // n-pointers then describe a range -- and update tau with the second)
    // kx / ky / ks != NULL: compensated (Kahan) accumulation of the iterate -- see comp_add
    // do_n: the x_x rows and tau (need the all-reduced gT and b.v); do_m: the x_y / x_s rows (local).  Both in one
    // launch, or the m-part first while the all-reduce is in flight.
    // block 0 sums post_k's block partials of c.u (ps_c) and b.v (ps_b: all-reduced block partials when sharded)
    float dc = 0.0f, db = 0.0f;
    if (do_tau && blockIdx.x == 0) {
        __shared__ double shd[16];
        dc = block_sum_of_partials(ps_c, np_c, shd);
        db = block_sum_of_partials(ps_b, np_b, shd);
    }
    const float kappa = st->kappa;
    const size_t gstride = (size_t)gridDim.x * BLK;
    if (do_n)
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < (size_t)n; i += gstride) {
        const float old = xx[i];
        const float nw = comp_add(old, Tx[i] * (gT[i] + c[i] * kappa), kx, i);
        xx[i] = nw;
        rxx[i] = old - 2.0f * nw;
    }
    if (do_m)
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < (size_t)m; i += gstride) {
        const unsigned char k = cls[i];
        const float oy = xy[i], os = xs[i];
        float ny = comp_add(oy, Ty[i] * (b[i] * kappa - hN[i]), ky, i);
        float ns = comp_add(os, Ts[i] * v[i], ks, i);
        if (k == 1) { ny = fmaxf(ny, 0.0f); ns = fmaxf(ns, 0.0f); }
        else if (k == 0) { ns = 0.0f; }
        xy[i] = ny;
        xs[i] = ns;
        rxy[i] = (k < 2) ? oy - 2.0f * ny : oy;
        rxs[i] = (k < 2) ? os - 2.0f * ns : os;
    }
    if (do_tau && blockIdx.x == 0 && threadIdx.x == 0) {
        // every other thread only reads st->kappa / st->stop; tau is written by this thread alone
        const float old = st->tau;
        float t = old + st->t_tau * (-dc - db);
        t = fmaxf(t, 0.0f);
        st->tau = t;
        st->r_tau = old - 2.0f * t;
    }
}

// rx <- rx - 2 x on the rows of PSD blocks (cls 3), after their projection
__global__ void rx_psd_k(int m, const unsigned char *__restrict__ cls, const float *__restrict__ xy,
                         const float *__restrict__ xs, float *__restrict__ rxy, float *__restrict__ rxs,
                         const DevStatus *st)
{
    if (st->stop != 0) return;
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < (size_t)m; i += (size_t)gridDim.x * BLK)
        if (cls[i] == 3) { rxy[i] = rxy[i] - 2.0f * xy[i]; rxs[i] = rxs[i] - 2.0f * xs[i]; }
}

// y-update, solver.rs:557-567 with ty = -K rx (SelfDualEmbed::op solver.rs:109-131):
//   u += Su o (-g2 - c rtau)          g2 = A^T rx_y
//   v += Sv o ( h2 + rx_s - b rtau)   h2 = A rx_x
//   kappa += Skappa (c.rx_x + b.rx_y) ; kappa <- min(kappa, 0)
// carried != 0: g2 = gP - 2 g3, h2 = hP - 2 h3 from the criteria products of x_k (gP, hP) and x_{k+1} (g3, h3),
//   which then become the previous ones;  carried == 0: g2 / h2 are given.
// docrit != 0: also the n-part of the criteria: block partials of ||d||^2 and c.x_x,
//   d = c + (A^T x_y)/tau (solver.rs:596-597) or A^T x_y (solver.rs:634)
__global__ __launch_bounds__(BLK) void ycrit_k(int n, int m, int doy, int carried, int docrit,
                                              const float *__restrict__ g3, const float *__restrict__ h3,
                                              float *__restrict__ gP, float *__restrict__ hP,
                                              const float *__restrict__ g2in, const float *__restrict__ h2in,
                                              const float *__restrict__ c, const float *__restrict__ b,
                                              const float *__restrict__ rxs, const float *__restrict__ Su,
                                              const float *__restrict__ Sv, float *__restrict__ u, float *__restrict__ v,
                                              const float *__restrict__ xx, float eps_zero, float *__restrict__ part,
                                              DevStatus *st, const float *ps_c, int np_c, const float *ps_b, int np_b,
                                              int do_n, int do_m, int do_kappa, int pbase, int pstride,
                                              float *__restrict__ ku, float *__restrict__ kv)
{
    if (st->stop != 0) return;
    // do_kappa: the kappa update (with do_n in one launch; column-split runs give the u rows as two ranges and update
    // kappa with the second).  The criteria partials of this launch go to part[pbase + block] (||d||^2) and
    // part[pstride + pbase + block] (c.x_x): one launch pbase = 0, pstride = gridDim.x
    // do_n: the u rows, kappa and the n-part of the criteria (need the all-reduced products); do_m: the v rows (local)
    __shared__ float sh[16];
    float dc = 0.0f, db = 0.0f;      // c.rx_x and b.rx_y: block partials of post_k summed here by block 0
    if (doy && do_kappa && blockIdx.x == 0) {
        __shared__ double shd[16];
        dc = block_sum_of_partials(ps_c, np_c, shd);
        db = block_sum_of_partials(ps_b, np_b, shd);
    }
    const float rtau = st->r_tau;
    const float tau = st->tau;
    const bool conv = tau > eps_zero;
    const float rt = conv ? 1.0f / tau : 1.0f;
    const size_t gstride = (size_t)gridDim.x * BLK;
    float dd = 0.0f, cx = 0.0f;
    if (do_n)
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < (size_t)n; i += gstride) {
        const float ci = c[i];
        if (doy) {
            float g2;
            if (carried) { const float nw = g3[i]; g2 = gP[i] - 2.0f * nw; gP[i] = nw; }
            else g2 = g2in[i];
            u[i] = comp_add(u[i], Su[i] * (-g2 - ci * rtau), ku, i);
        }
        if (docrit) {
            const float d = conv ? fmaf(rt, g3[i], ci) : g3[i];
            dd = fmaf(d, d, dd);
            cx = fmaf(ci, xx[i], cx);
        }
    }
    if (doy) {
        if (do_m)
        for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < (size_t)m; i += gstride) {
            float h2;
            if (carried) { const float nw = h3[i]; h2 = hP[i] - 2.0f * nw; hP[i] = nw; }
            else h2 = h2in[i];
            v[i] = comp_add(v[i], Sv[i] * (h2 + rxs[i] - b[i] * rtau), kv, i);
        }
        if (do_kappa && blockIdx.x == 0 && threadIdx.x == 0) {
            const float k = st->kappa + st->s_kappa * (dc + db);
            st->kappa = fminf(k, 0.0f);       // solver.rs:566-567
        }
    }
    if (docrit && do_n) {
        dd = block_sum(dd, sh);
        cx = block_sum(cx, sh);
        if (threadIdx.x == 0) { part[pbase + blockIdx.x] = dd; part[pstride + pbase + blockIdx.x] = cx; }
    }
}

// the termination test, solver.rs:381-451 + the tails of criteria_conv / criteria_inf (solver.rs:599-611,
// 636-655).  status_eval is run by a whole block of 256 threads: six sums of block partials (f64 accumulation), each by ONE
// wave, two per wave, all loads in flight together and one barrier; every thread then holds the verdict.  status_k -- one
// block -- commits it; the merged m-kernels of the one-pass schedule evaluate it at their head in EVERY block (same inputs,
// same arithmetic, same verdict) so that the test of iterate k needs no launch of its own between sweep k and step k + 1.
struct StatArgs {
    int np; const float *part;                       // sums over n: ||d||^2 = part[0 .. np), c.x_x = part[np .. 2 np)
    const float *ps_pp, *ps_by; int npsum;           // sums over m: ||p||^2, b.x_y (post_k's / the m-kernel's block partials)
    const float *ps_cu; int np_cu; const float *ps_bv; int np_bv;      // sweep schedule: c.u, b.v -> the next tau; else NULL
    const float *fault_flag;                         // column-sharded sweep: the all-reduced "a kernel gave up" flag, else NULL
    float eps_acc, eps_inf, eps_zero; long long max_iter; int xbuf;
};
struct StatOut { int state, kind; float cri[3]; float tau_next, r_tau_next; };      // state: THIP_ST_*, or -2: a peer's fault

__device__ __forceinline__ StatOut status_eval(const StatArgs &a, float tau, long long i, float norm_b, float norm_c, float t_tau,
                                               double *sums /* 8 doubles of LDS */)
{
    StatOut o;
    o.state = THIP_ST_RUNNING; o.kind = 0; o.cri[0] = o.cri[1] = o.cri[2] = 0.0f; o.tau_next = tau; o.r_tau_next = 0.0f;
    if (a.fault_flag != nullptr && *a.fault_flag > 0.0f) { o.state = -2; return o; }      // (uniform: every thread reads the same word)
    {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const float *src[2] = { nullptr, nullptr };
        int cnt[2] = { 0, 0 };
        switch (w) {
        case 0: src[0] = a.ps_pp; cnt[0] = a.npsum; src[1] = a.ps_cu; cnt[1] = a.ps_cu ? a.np_cu : 0; break;
        case 1: src[0] = a.ps_by; cnt[0] = a.npsum; src[1] = a.ps_bv; cnt[1] = a.ps_bv ? a.np_bv : 0; break;
        case 2: src[0] = a.part; cnt[0] = a.np; break;
        default: src[0] = a.part + a.np; cnt[0] = a.np; break;
        }
        double acc[2] = { 0.0, 0.0 };
#pragma unroll
        for (int q = 0; q < 2; ++q)
            for (int k = lane; k < cnt[q]; k += 64) acc[q] += (double)src[q][k];
        acc[0] = wave_sum_d(acc[0]);
        acc[1] = wave_sum_d(acc[1]);
        if (lane == 0) { sums[w] = acc[0]; sums[4 + w] = acc[1]; }
    }
    __syncthreads();
    const float pp = (float)sums[0], by = (float)sums[1];
    const float dd = (float)sums[2], cx = (float)sums[3];
    const float dcu = (float)sums[4], dbv = (float)sums[5];
    const bool excess_iter = (a.max_iter >= 0) ? (i + 1 >= a.max_iter) : false;
    const float norm_p = sqrtf(pp), norm_d = sqrtf(dd);
    int state = THIP_ST_RUNNING;
    if (tau > a.eps_zero) {
        const float rt = 1.0f / tau;
        const float g_x = rt * cx;
        const float g_y = rt * by;
        const float g = g_x + g_y;
        const float cri_pri = norm_p / (1.0f + norm_b);
        const float cri_dual = norm_d / (1.0f + norm_c);
        const float cri_gap = fabsf(g) / (1.0f + fabsf(g_x) + fabsf(g_y));
        o.kind = 0; o.cri[0] = cri_pri; o.cri[1] = cri_dual; o.cri[2] = cri_gap;
        const bool term_conv = (cri_pri <= a.eps_acc) && (cri_dual <= a.eps_acc) && (cri_gap <= a.eps_acc);
        if (term_conv) state = THIP_ST_OK;
        else if (excess_iter) state = THIP_ST_EXCESS_ITER;
    } else {
        const float m_cx = -cx;
        const float m_by = -by;
        const float cri_unbdd = (m_cx > a.eps_zero) ? norm_p * norm_c / m_cx : __builtin_inff();
        const float cri_infeas = (m_by > a.eps_zero) ? norm_d * norm_b / m_by : __builtin_inff();
        o.kind = 1; o.cri[0] = cri_unbdd; o.cri[1] = cri_infeas; o.cri[2] = 0.0f;
        const bool term_unbdd = cri_unbdd <= a.eps_inf, term_infeas = cri_infeas <= a.eps_inf;
        if (term_unbdd) state = THIP_ST_UNBOUNDED;
        else if (term_infeas) state = THIP_ST_INFEASIBLE;
        else if (excess_iter) state = THIP_ST_EXCESS_ITER;
    }
    o.state = state;
    if (state == THIP_ST_RUNNING && a.ps_cu != nullptr) {
        // sweep schedule: c.u_k and b.v_k are complete here too -- the tau update of the NEXT step (solver.rs:551-552)
        const float t = fmaxf(tau + t_tau * (-dcu - dbv), 0.0f);
        o.tau_next = t;
        o.r_tau_next = tau - 2.0f * t;
    }
    return o;
}

// what ONE thread writes for a verdict that ends the loop (or for a peer's fault); a RUNNING verdict is committed by its
// caller (status_k here; the m-kernels at their end)
__device__ __forceinline__ void status_commit_stop(DevStatus *st, const StatOut &o, int xbuf)
{
    if (o.state == -2) { st->fault = 1; st->stop = 1; return; }
    st->kind = o.kind; st->cri[0] = o.cri[0]; st->cri[1] = o.cri[1]; st->cri[2] = o.cri[2];
    st->state = o.state; st->xbuf = xbuf; st->stop = 1;      // later launches are no-ops
}

__global__ __launch_bounds__(BLK) void status_k(const StatArgs a, DevStatus *st)
{
    if (st->stop != 0) return;
    __shared__ double sums[8];
    const long long i = st->iter;
    const StatOut o = status_eval(a, st->tau, i, st->norm_b, st->norm_c, st->t_tau, sums);
    if (threadIdx.x != 0) return;
    if (o.state != THIP_ST_RUNNING) { status_commit_stop(st, o, a.xbuf); return; }
    st->kind = o.kind; st->cri[0] = o.cri[0]; st->cri[1] = o.cri[1]; st->cri[2] = o.cri[2];
    st->iter = i + 1;
    if (a.ps_cu != nullptr) { st->tau_next = o.tau_next; st->r_tau_next = o.r_tau_next; }
}

// ---------------------------------------------------------------------------------------------------
// THIP_SCHED_SWEEP: the O(n + m) work between two sweeps over A (thip_sweep.hip).  Step k (k >= 1) is
//   sw_xm_k   tau_k ; x_y_k, x_s_k from hN = A u_{k-1} (the groups' shares summed here), element-wise cones, rx
//   [block cones]
//   sw_vm_k   v_k from h2 = hP - 2 h3 (h3 = A x_x_k, carried form) ; partial sums of b.v_k, b.rx_y, ||p_k||^2, b.x_y_k
//   SWEEP     kappa_k (every workgroup for itself, at entry) ; u_k, x_x_{k+1}, gP = A^T x_y_k, the shares of A u_k and A x_x_{k+1}, and -- per workgroup, over the columns it
//             writes -- the partial sums over n: ||d_k||^2, c.x_x_k, c.u_k, c.rx_x_k
//   status_k  the termination test of iterate k (solver.rs:381-451)
// The arithmetic of every update is xupdate_k's / ycrit_k's / post_k's.
// ---------------------------------------------------------------------------------------------------
// MERGE (no block cones: every row's projection is element-wise): the same launch also does sw_vm_k's part of the row -- v_k,
// the sums over m -- with tau_k / rx_tau as the previous termination test left them (DevStatus::tau_next), and leaves its
// four block partials in `part` (gridDim.x each)
template <bool MERGE>
__global__ __launch_bounds__(BLK) void sw_xm_k(int m, int ngroups, size_t mpad, const float *__restrict__ partH,
                                              float *__restrict__ h3, const float *__restrict__ b,
                                              float *__restrict__ v, const float *__restrict__ Ty,
                                              const float *__restrict__ Ts, const unsigned char *__restrict__ cls,
                                              float *__restrict__ xy, float *__restrict__ xs, float *__restrict__ rxy,
                                              float *__restrict__ rxs, DevStatus *st, float *__restrict__ ky, float *__restrict__ ks,
                                              float *__restrict__ hP, const float *__restrict__ Sv, float *__restrict__ kv,
                                              float eps_zero, float *__restrict__ part, const StatArgs sa, int fold, int par)
{
    if (st->stop != 0) return;
    // fold != 0: the termination test of the PREVIOUS iterate has had no launch of its own -- every block evaluates it here,
    // from the same sums, before anything is written; a verdict that ends the loop leaves the iterate alone (the launches
    // behind this one -- block cones, sw_vm_k, the sweep -- return at entry on the stop flag block 0 raises)
    float tau, rtau;
    StatOut so;
    long long it0 = 0;
    if (fold) {
        __shared__ double ssum[8];
        it0 = st->iter_r[par];
        so = status_eval(sa, st->tau_r[par], it0, st->norm_b, st->norm_c, st->t_tau, ssum);
        if (so.state != THIP_ST_RUNNING) {
            if (blockIdx.x == 0 && threadIdx.x == 0) status_commit_stop(st, so, sa.xbuf);
            return;
        }
        tau = so.tau_next; rtau = so.r_tau_next;
    } else {
        tau = st->tau_next; rtau = st->r_tau_next;      // tau_k, rx_tau as status_k / sw_tau_k left them (read-only here)
    }
    const float kappa = st->kappa;
    const bool conv = tau > eps_zero;
    const float rt = conv ? 1.0f / tau : 1.0f;
    float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f, q3 = 0.0f;
    // 256 threads = 64 rows x 4 group lanes (post_k's shape): the sum over the groups' shares of one row -- up to 256 of
    // them, 128 at the 10 000-variable LP -- is split four ways, combined through LDS, and lane 0 of a row does the update
    __shared__ float comb[2][3][64];
    const int e = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const size_t gstride = (size_t)gridDim.x * 64;
    for (size_t i0 = blockIdx.x * (size_t)64; i0 < (size_t)m; i0 += gstride) {
        const size_t i = i0 + e;
        float sa0 = 0.0f, sa1 = 0.0f, sb0 = 0.0f, sb1 = 0.0f;
        if (i < (size_t)m) {
            int g = kq;
            // (sixteen loads in flight, added in the order of the plain loop below: the shares were written by workgroups on
            // other XCDs, so every one of them is a trip to the Infinity Cache)
            for (; g + 28 < ngroups; g += 32) {
                float t[16];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float *q = partH + (size_t)(g + 8 * j) * 2 * mpad + i;
                    t[4 * j] = q[0]; t[4 * j + 1] = q[mpad]; t[4 * j + 2] = q[8 * mpad]; t[4 * j + 3] = q[9 * mpad];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { sa0 += t[4 * j]; sb0 += t[4 * j + 1]; sa1 += t[4 * j + 2]; sb1 += t[4 * j + 3]; }
            }
            for (; g + 4 < ngroups; g += 8) {
                const float *q = partH + (size_t)g * 2 * mpad + i;
                sa0 += q[0]; sb0 += q[mpad]; sa1 += q[8 * mpad]; sb1 += q[9 * mpad];
            }
            if (g < ngroups) { sa0 += partH[((size_t)g * 2 + 0) * mpad + i]; sb0 += partH[((size_t)g * 2 + 1) * mpad + i]; }
        }
        const float sa = sa0 + sa1, sb = sb0 + sb1;
        if (kq > 0) { comb[0][kq - 1][e] = sa; comb[1][kq - 1][e] = sb; }
        __syncthreads();
        if (kq == 0 && i < (size_t)m) {
            const float hN = (sa + comb[0][0][e]) + (comb[0][1][e] + comb[0][2][e]);
            const float hx = (sb + comb[1][0][e]) + (comb[1][1][e] + comb[1][2][e]);
            const unsigned char k = cls[i];
            const float oy = xy[i], os = xs[i], bi = b[i], vi = v[i];
            float ny = comp_add(oy, Ty[i] * (bi * kappa - hN), ky, i);
            float ns = comp_add(os, Ts[i] * vi, ks, i);
            if (k == 1) { ny = fmaxf(ny, 0.0f); ns = fmaxf(ns, 0.0f); }
            else if (k == 0) { ns = 0.0f; }
            xy[i] = ny;
            xs[i] = ns;
            const float ry = (k < 2) ? oy - 2.0f * ny : oy, rs = (k < 2) ? os - 2.0f * ns : os;
            rxy[i] = ry;
            rxs[i] = rs;
            if constexpr (MERGE) {
                // sw_vm_k's row: v_k from h2 = hP - 2 h3 (h3 = A x_x_k) ; the criteria sums over m
                const float h2 = hP[i] - 2.0f * hx;
                hP[i] = hx;
                const float vn = comp_add(vi, Sv[i] * (h2 + rs - bi * rtau), kv, i);
                v[i] = vn;
                q0 = fmaf(bi, vn, q0);
                q1 = fmaf(bi, ry, q1);
                float p;
                if (conv) { p = ns * rt - bi; p = fmaf(rt, hx, p); }
                else p = ns + hx;
                q2 = fmaf(p, p, q2);
                q3 = fmaf(bi, ny, q3);
            } else {
                h3[i] = hx;
            }
        }
        __syncthreads();
    }
    if constexpr (MERGE) {
        __shared__ float sh[16];
        q0 = block_sum(q0, sh); q1 = block_sum(q1, sh); q2 = block_sum(q2, sh); q3 = block_sum(q3, sh);
        if (threadIdx.x == 0) {
            part[blockIdx.x] = q0; part[gridDim.x + blockIdx.x] = q1;
            part[2 * gridDim.x + blockIdx.x] = q2; part[3 * gridDim.x + blockIdx.x] = q3;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->tau = tau;
        st->r_tau = rtau;
        if (MERGE) st->kappa_in = kappa;          // nobody writes kappa between here and the sweep
        st->tau_r[par ^ 1] = tau;
        if (fold) {
            st->kind = so.kind; st->cri[0] = so.cri[0]; st->cri[1] = so.cri[1]; st->cri[2] = so.cri[2];
            st->iter = it0 + 1; st->iter_r[par ^ 1] = it0 + 1;
        } else {
            st->iter_r[par ^ 1] = st->iter;
        }
    }
}

// Every row in a second-order cone of at most 129 rows (BASELINE configs[2]: 1000 cones of 100): the WHOLE m-tail of a step
// as one launch, one workgroup per cone (looping when there are more than EG cones) -- sw_xm_k's row: the groups' shares
// summed by four lanes per row exactly as there, x_y / x_s; then on the workgroup's first wave soc_k's projection of both
// blocks (cone_soc.rs:38-65; the same lane <-> row mapping and f64 sum of squares) and sw_vm_k's row (v, the sums over m).
// Lane l of that wave holds rows beg + 1 + l and beg + 65 + l, lane 0 also the cone's first row.  Every per-row value has
// the arithmetic of the three-launch form; only the block partials of the four sums over m are grouped differently.
__global__ __launch_bounds__(BLK) void sw_cone_k(int n_cones, const int64_t *__restrict__ begs, const int64_t *__restrict__ ends,
                                                int ngroups, size_t mpad, const float *__restrict__ partH,
                                                const float *__restrict__ b, float *__restrict__ v, const float *__restrict__ Ty,
                                                const float *__restrict__ Ts, float *__restrict__ xy, float *__restrict__ xs,
                                                float *__restrict__ rxy, float *__restrict__ rxs, DevStatus *st,
                                                float *__restrict__ ky, float *__restrict__ ks, float *__restrict__ hP,
                                                const float *__restrict__ Sv, float *__restrict__ kv, float eps_zero,
                                                float *__restrict__ part, const StatArgs sa, int fold, int par)
{
    if (st->stop != 0) return;
    const int e = threadIdx.x & 63, kq = threadIdx.x >> 6;
    float tau, rtau;                               // (the folded termination test: see sw_xm_k)
    StatOut so;
    long long it0 = 0;
    if (fold) {
        __shared__ double ssum[8];
        it0 = st->iter_r[par];
        so = status_eval(sa, st->tau_r[par], it0, st->norm_b, st->norm_c, st->t_tau, ssum);
        if (so.state != THIP_ST_RUNNING) {
            if (blockIdx.x == 0 && threadIdx.x == 0) status_commit_stop(st, so, sa.xbuf);
            return;
        }
        tau = so.tau_next; rtau = so.r_tau_next;
    } else {
        tau = st->tau_next; rtau = st->r_tau_next;
    }
    const float kappa = st->kappa;
    const bool conv = tau > eps_zero;
    const float rt = conv ? 1.0f / tau : 1.0f;
    __shared__ float comb[3][2][3][64];           // [row slot][product][lane quarter - 1][row lane]
    float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f, q3 = 0.0f;
    for (int cone = blockIdx.x; cone < n_cones; cone += gridDim.x) {
        const int64_t beg = begs[cone], end = ends[cone];
        if (end <= beg) continue;                  // (uniform over the workgroup)
        // slot 0 / 1: rows beg + 1 + e (+ 64); slot 2: the first row (lane 0)
        const size_t idx[3] = { (size_t)(beg + 1 + e), (size_t)(beg + 65 + e), (size_t)beg };
        const bool ok[3] = { beg + 1 + e < end, beg + 65 + e < end, e == 0 };
        // the first wave's row data: requested before the shares, used after the barrier (one round trip for everything)
        // (the Kahan terms, hP and Sv too: the cone phase below is then arithmetic and stores only -- with them fetched where
        // they are used it was three dependent round trips per cone, 20 us per launch at configs[2])
        float oy[3], os[3], bi[3], vi[3], tyv[3], tsv[3], kyv[3], ksv[3], kvv[3], hPv[3], svv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            oy[k] = os[k] = bi[k] = vi[k] = tyv[k] = tsv[k] = kyv[k] = ksv[k] = kvv[k] = hPv[k] = svv[k] = 0.0f;
            if (kq == 0 && ok[k]) {
                const size_t i = idx[k];
                oy[k] = xy[i]; os[k] = xs[i]; bi[k] = b[i]; vi[k] = v[i]; tyv[k] = Ty[i]; tsv[k] = Ts[i];
                hPv[k] = hP[i]; svv[k] = Sv[i];
                if (ky) kyv[k] = ky[i];
                if (ks) ksv[k] = ks[i];
                if (kv) kvv[k] = kv[i];
            }
        }
        // comp_add with the Kahan term already in a register
        auto cadd = [](float x, float inc, float *__restrict__ kp, size_t i, float kval) -> float {
            if (kp == nullptr) return x + inc;
            const float y = inc - kval;
            const float t = x + y;
            kp[i] = (t - x) - y;
            return t;
        };
        float sas[3], sbs[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            // the shares of row idx[k]: this thread's quarter of the groups, in sw_xm_k's order
            float sa0 = 0.0f, sa1 = 0.0f, sb0 = 0.0f, sb1 = 0.0f;
            if (ok[k]) {
                const size_t i = idx[k];
                int g = kq;
                for (; g + 28 < ngroups; g += 32) {
                    float t[16];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float *q = partH + (size_t)(g + 8 * j) * 2 * mpad + i;
                        t[4 * j] = q[0]; t[4 * j + 1] = q[mpad]; t[4 * j + 2] = q[8 * mpad]; t[4 * j + 3] = q[9 * mpad];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { sa0 += t[4 * j]; sb0 += t[4 * j + 1]; sa1 += t[4 * j + 2]; sb1 += t[4 * j + 3]; }
                }
                for (; g + 4 < ngroups; g += 8) {
                    const float *q = partH + (size_t)g * 2 * mpad + i;
                    sa0 += q[0]; sb0 += q[mpad]; sa1 += q[8 * mpad]; sb1 += q[9 * mpad];
                }
                if (g < ngroups) { sa0 += partH[((size_t)g * 2 + 0) * mpad + i]; sb0 += partH[((size_t)g * 2 + 1) * mpad + i]; }
            }
            sas[k] = sa0 + sa1; sbs[k] = sb0 + sb1;
            if (kq > 0) { comb[k][0][kq - 1][e] = sas[k]; comb[k][1][kq - 1][e] = sbs[k]; }
        }
        __syncthreads();
        float hNs[3] = { 0.0f, 0.0f, 0.0f }, hxs[3] = { 0.0f, 0.0f, 0.0f };
        if (kq == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                hNs[k] = (sas[k] + comb[k][0][0][e]) + (comb[k][0][1][e] + comb[k][0][2][e]);
                hxs[k] = (sbs[k] + comb[k][1][0][e]) + (comb[k][1][1][e] + comb[k][1][2][e]);
            }
        }
        __syncthreads();                           // (the next cone's shares may overwrite comb)
        if (kq != 0) continue;                     // the cone itself is the first wave's (no barrier below)
        float ny[3], ns[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ny[k] = ns[k] = 0.0f;
            if (!ok[k]) continue;
            const size_t i = idx[k];
            ny[k] = cadd(oy[k], tyv[k] * (bi[k] * kappa - hNs[k]), ky, i, kyv[k]);
            ns[k] = cadd(os[k], tsv[k] * vi[k], ks, i, ksv[k]);
        }
        // the projection of the x_y block and of the x_s block (soc_k, not rotated)
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            float *x = which ? ns : ny;
            const float s0 = __shfl(x[2], 0, 64);
            double acc = 0.0;
            if (ok[0]) acc += (double)x[0] * (double)x[0];
            if (ok[1]) acc += (double)x[1] * (double)x[1];
            const float norm_v = (float)sqrt(wave_sum_d(acc));
            float f, s_new;
            if (norm_v <= -s0) { f = 0.0f; s_new = 0.0f; }
            else if (norm_v <= s0) { f = 1.0f; s_new = s0; }
            else { f = (1.0f + s0 / norm_v) / 2.0f; s_new = (norm_v + s0) / 2.0f; }
            x[2] = s_new;
            if (f != 1.0f) { x[0] = f * x[0]; x[1] = f * x[1]; }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!ok[k]) continue;
            const size_t i = idx[k];
            xy[i] = ny[k];
            xs[i] = ns[k];
            const float ry = oy[k] - 2.0f * ny[k], rs = os[k] - 2.0f * ns[k];
            rxy[i] = ry;
            rxs[i] = rs;
            const float hx = hxs[k];
            const float h2 = hPv[k] - 2.0f * hx;
            hP[i] = hx;
            const float vn = cadd(vi[k], svv[k] * (h2 + rs - bi[k] * rtau), kv, i, kvv[k]);
            v[i] = vn;
            q0 = fmaf(bi[k], vn, q0);
            q1 = fmaf(bi[k], ry, q1);
            float p;
            if (conv) { p = ns[k] * rt - bi[k]; p = fmaf(rt, hx, p); }
            else p = ns[k] + hx;
            q2 = fmaf(p, p, q2);
            q3 = fmaf(bi[k], ny[k], q3);
        }
    }
    if (kq == 0) {
        q0 = wave_sum(q0); q1 = wave_sum(q1); q2 = wave_sum(q2); q3 = wave_sum(q3);
        if (e == 0) {
            part[blockIdx.x] = q0; part[gridDim.x + blockIdx.x] = q1;
            part[2 * gridDim.x + blockIdx.x] = q2; part[3 * gridDim.x + blockIdx.x] = q3;
            if (blockIdx.x == 0) {
                st->tau = tau;
                st->r_tau = rtau;
                st->kappa_in = kappa;
                st->tau_r[par ^ 1] = tau;
                if (fold) {
                    st->kind = so.kind; st->cri[0] = so.cri[0]; st->cri[1] = so.cri[1]; st->cri[2] = so.cri[2];
                    st->iter = it0 + 1; st->iter_r[par ^ 1] = it0 + 1;
                } else {
                    st->iter_r[par ^ 1] = st->iter;
                }
            }
        }
    }
}

// (re)start of the one-pass schedule: the tau update a regular step takes from the previous termination test
__global__ __launch_bounds__(BLK) void sw_tau_k(DevStatus *st, const float *ps_cu, int np_cu, const float *ps_bv, int np_bv)
{
    if (st->stop != 0) return;
    // (summed as status_k sums them: one wave per quantity)
    __shared__ double sums[2];
    {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (w < 2) {
            const float *src = w == 0 ? ps_cu : ps_bv;
            const int cnt = w == 0 ? np_cu : np_bv;
            double acc = 0.0;
            for (int k = lane; k < cnt; k += 64) acc += (double)src[k];
            acc = wave_sum_d(acc);
            if (lane == 0) sums[w] = acc;
        }
    }
    __syncthreads();
    const float dc = (float)sums[0], db = (float)sums[1];
    if (threadIdx.x == 0) {
        const float old = st->tau;
        const float t = fmaxf(old + st->t_tau * (-dc - db), 0.0f);
        st->tau_next = t;
        st->r_tau_next = old - 2.0f * t;
    }
}

// part: [0] b.v_k, [1] b.rx_y, [2] ||p_k||^2, [3] b.x_y_k, gridDim.x block partials each
__global__ __launch_bounds__(BLK) void sw_vm_k(int m, const float *__restrict__ h3, float *__restrict__ hP,
                                              const float *__restrict__ b, const float *__restrict__ rxs,
                                              const float *__restrict__ rxy, const float *__restrict__ Sv,
                                              float *__restrict__ v, float *__restrict__ kv, const float *__restrict__ xs,
                                              const float *__restrict__ xy, float eps_zero, DevStatus *st,
                                              float *__restrict__ part)
{
    if (st->stop != 0) return;
    __shared__ float sh[16];
    if (blockIdx.x == 0 && threadIdx.x == 0) st->kappa_in = st->kappa;      // nobody writes kappa between here and the sweep
    const float rtau = st->r_tau, tau = st->tau;
    const bool conv = tau > eps_zero;
    const float rt = conv ? 1.0f / tau : 1.0f;
    float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f, q3 = 0.0f;
    const size_t gstride = (size_t)gridDim.x * BLK;
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < (size_t)m; i += gstride) {
        const float nw = h3[i], bi = b[i];
        const float h2 = hP[i] - 2.0f * nw;
        hP[i] = nw;
        const float vn = comp_add(v[i], Sv[i] * (h2 + rxs[i] - bi * rtau), kv, i);
        v[i] = vn;
        q0 = fmaf(bi, vn, q0);
        q1 = fmaf(bi, rxy[i], q1);
        float p;
        if (conv) { p = xs[i] * rt - bi; p = fmaf(rt, nw, p); }
        else p = xs[i] + nw;
        q2 = fmaf(p, p, q2);
        q3 = fmaf(bi, xy[i], q3);
    }
    q0 = block_sum(q0, sh); q1 = block_sum(q1, sh); q2 = block_sum(q2, sh); q3 = block_sum(q3, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = q0; part[gridDim.x + blockIdx.x] = q1;
        part[2 * gridDim.x + blockIdx.x] = q2; part[3 * gridDim.x + blockIdx.x] = q3;
    }
}

// (re)start of the one-pass schedule from a consistent iterate: block partials of b.v (a step leaves them for the next
// one's tau update; whatever ran before this -- nothing, or the carried schedule -- did not)
__global__ __launch_bounds__(BLK) void sw_bv_k(int m, const float *__restrict__ b, const float *__restrict__ v,
                                              float *__restrict__ part, const DevStatus *st)
{
    if (st->stop != 0) return;
    __shared__ float sh[16];
    float q = 0.0f;
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < (size_t)m; i += (size_t)gridDim.x * BLK) q = fmaf(b[i], v[i], q);
    q = block_sum(q, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = q;
}

// column-sharded runs: the groups' shares of the two N products summed into the buffer that is all-reduced
// + this rank's 4 x 256 sums over n (the sweep wrote them to its own buffer) into the 4 x EG slots of the tail, and this
// rank's "my kernel gave up" flag behind them
__global__ __launch_bounds__(BLK) void sw_gsum_k(int m, int ngroups, size_t mpad, const float *__restrict__ partH,
                                                float *__restrict__ out, const DevStatus *st, const float *__restrict__ pn_loc,
                                                const unsigned *__restrict__ errw)
{
    if (st->stop != 0) return;
    if (blockIdx.x == 0) {
        float *tail = out + 2 * mpad;
        for (int q = 0; q < 4; ++q) tail[q * (int)EG + threadIdx.x] = pn_loc[q * 256 + threadIdx.x];      // BLK == 256 workgroups of the sweep
        if (threadIdx.x == 0) tail[4 * EG] = *errw != 0u ? 1.0f : 0.0f;
    }
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < (size_t)m; i += (size_t)gridDim.x * BLK) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f, b3 = 0.0f;      // as in sw_xm_k
        int g = 0;
        for (; g + 4 <= ngroups; g += 4) {
            const float *q = partH + (size_t)g * 2 * mpad + i;
            a0 += q[0]; b0 += q[mpad]; a1 += q[2 * mpad]; b1 += q[3 * mpad];
            a2 += q[4 * mpad]; b2 += q[5 * mpad]; a3 += q[6 * mpad]; b3 += q[7 * mpad];
        }
        for (; g < ngroups; ++g) { a0 += partH[((size_t)g * 2 + 0) * mpad + i]; b0 += partH[((size_t)g * 2 + 1) * mpad + i]; }
        out[i] = (a0 + a1) + (a2 + a3); out[mpad + i] = (b0 + b1) + (b2 + b3);
    }
}

// solver.rs:397-400: on Converged / ExcessIter in the tau > eps_zero branch, x_x and x_y are scaled by 1/tau.  Launched
// once, by the host, when it first sees the terminated state (poll()): status_k itself raises the stop flag, so
// nothing else touches the iterate in between and no per-iteration launch is spent on a no-op.
__global__ void finalize_k(int n, int m, float *__restrict__ xx, float *__restrict__ xy, const DevStatus *st)
{
    const bool scale = (st->kind == 0) && (st->state == THIP_ST_OK || st->state == THIP_ST_EXCESS_ITER);
    if (!scale) return;
    const float rt = 1.0f / st->tau;
    const size_t gstride = (size_t)gridDim.x * BLK;
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < (size_t)n; i += gstride) xx[i] = rt * xx[i];
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < (size_t)m; i += gstride) xy[i] = rt * xy[i];
}

// thip_solver_resume: undo finalize_k (x_x, x_y back to the homogeneous iterate) and clear the termination
__global__ void resume_k(int n, int m, float *__restrict__ xx, float *__restrict__ xy, const DevStatus *st)
{
    const float tau = st->tau;
    const size_t gstride = (size_t)gridDim.x * BLK;
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < (size_t)n; i += gstride) xx[i] = tau * xx[i];
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < (size_t)m; i += gstride) xy[i] = tau * xy[i];
}
__global__ void resume_flags_k(DevStatus *st)
{
    st->state = THIP_ST_RUNNING;
    st->iter = st->iter + 1;
    st->stop = 0;
}

// sum of `np` block partials of `nq` quantities (part[q*np + k]) -> out[q]; one block (init only)
__global__ void sum_partials_k(int nq, int np, const float *__restrict__ part, float *__restrict__ out,
                               const int *__restrict__ stop)
{
    if (stop != nullptr && *stop != 0) return;
    __shared__ double shd[16];
    for (int q = 0; q < nq; ++q) {
        double acc = 0.0;
        for (int k = threadIdx.x; k < np; k += blockDim.x) acc += (double)part[(size_t)q * np + k];
        acc = block_sum_d(acc, shd);
        if (threadIdx.x == 0) out[q] = (float)acc;
        __syncthreads();
    }
}

__global__ void init_status_k(DevStatus *st, float norm_b_sq_dummy)
{
    (void)norm_b_sq_dummy;
    st->stop = 0; st->state = THIP_ST_RUNNING; st->kind = 0; st->iter = 0; st->fault = 0;
    st->cri[0] = st->cri[1] = st->cri[2] = 0.0f;
    st->tau = 1.0f; st->kappa = 0.0f; st->r_tau = 0.0f;
}

// norms + scalar preconditioner entries (solver.rs:460-481, 159-183, 501-506)
//   sums[0] = sum b^2 (all-reduced), sums[1] = sum |b| (all-reduced), loc[0] = sum c^2, loc[1] = sum |c|
__global__ void init_scalars_k(const float *__restrict__ sums, const float *__restrict__ loc, float eps_zero,
                               DevStatus *st)
{
    // fr_norm (solver.rs:85-107): n = norm(col); sq_norm += n*n; sqrt(sq_norm)
    const float nb = sqrtf(sums[0]), nc = sqrtf(loc[0]);
    st->norm_b = sqrtf(nb * nb);
    st->norm_c = sqrtf(nc * nc);
    const float tau_tau = loc[1] + sums[1];            // c.absadd_cols + b.absadd_cols (solver.rs:171-172)
    st->t_tau = 1.0f / fmaxf(tau_tau, eps_zero);
    st->s_kappa = 1.0f / fmaxf(tau_tau, eps_zero);     // sigma_1 = tau_tau (solver.rs:182)
}

// block partials of sum b^2, sum |b| (q0,q1 over m) and sum c^2, sum |c| (q2,q3 over n)
__global__ void init_sums_k(int m, const float *__restrict__ b, int n, const float *__restrict__ c,
                            float *__restrict__ part)
{
    __shared__ float sh[16];
    float b2 = 0.0f, b1 = 0.0f, c2 = 0.0f, c1 = 0.0f;
    const size_t gstride = (size_t)gridDim.x * BLK;
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < (size_t)m; i += gstride) { const float t = b[i]; b2 = fmaf(t, t, b2); b1 += fabsf(t); }
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < (size_t)n; i += gstride) { const float t = c[i]; c2 = fmaf(t, t, c2); c1 += fabsf(t); }
    b2 = block_sum(b2, sh); b1 = block_sum(b1, sh); c2 = block_sum(c2, sh); c1 = block_sum(c1, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = b2; part[gridDim.x + blockIdx.x] = b1;
        part[2 * gridDim.x + blockIdx.x] = c2; part[3 * gridDim.x + blockIdx.x] = c1;
    }
}

// vector preconditioners (solver.rs:159-183 then 501-506):
//   tau_x = colabs(A) + |c| ; tau_y = rowabs(A) + rowabs(b) ; tau_s = 1
//   sigma_n = tau_x ; sigma_m = tau_y + tau_s
__global__ void precond_k(int n, int m, const float *__restrict__ colabs, const float *__restrict__ rowabs,
                          const float *__restrict__ c, const float *__restrict__ b, const float *__restrict__ b_rowabs,
                          float eps_zero, float *__restrict__ Tx, float *__restrict__ Ty, float *__restrict__ Ts,
                          float *__restrict__ Su, float *__restrict__ Sv)
{
    const size_t gstride = (size_t)gridDim.x * BLK;
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < (size_t)n; i += gstride) {
        const float t = colabs[i] + fabsf(c[i]);
        const float r = 1.0f / fmaxf(t, eps_zero);
        Tx[i] = r; Su[i] = r;
    }
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < (size_t)m; i += gstride) {
        const float t = rowabs[i] + (b_rowabs ? b_rowabs[i] : fabsf(b[i]));
        Ty[i] = 1.0f / fmaxf(t, eps_zero);
        Ts[i] = 1.0f / fmaxf(1.0f, eps_zero);
        Sv[i] = 1.0f / fmaxf(t + 1.0f, eps_zero);
    }
}

// Cross-stream hand-offs of the column-split pipeline without events: the producer stream runs signal_k after its last
// kernel (in stream order), the consumer stream runs gate_k, which spins until the flag has reached the value.  A
// hipEventRecord / hipStreamWaitEvent pair leaves ~13 us of idle stream behind the record on this part (rocprof timeline,
// DESIGN.md 6.1); a one-thread kernel boundary costs 2-3 us.  Every gate is enqueued (host order) after its signal, so
// the pair is satisfiable on any queue mapping; a gate gives up after ~2 s and raises err instead of hanging the GPU.
__global__ void signal_k(unsigned *flag, unsigned val)
{
    __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void gate_k(const unsigned *flag, unsigned val, long long timeout_ticks, unsigned *err, int *stop)
{
    const long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - val) < 0) {
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > timeout_ticks) {
            // the sums this gate waits for never came: nothing enqueued behind it may touch the iterate (every kernel of the
            // loop returns at entry on the stop flag); the host reads the error word after the batch (thip_solver_run)
            atomicExch(err, 1u);
            __hip_atomic_store(stop, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

// thip_test_spin_allreduce: a stand-in collective that only takes time -- one thread spinning on the constant-rate
// device clock for `ticks`, on whatever stream the hook is given
__global__ void spin_k(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// the snapshot of the consistent iterate the one-pass schedule keeps per batch (and its restore): seven contiguous pieces
// of the arena and the status block in ONE launch
struct SnapArgs { const float *src[7]; float *dst[7]; size_t len[7]; const DevStatus *st_src; DevStatus *st_dst; };
__global__ __launch_bounds__(BLK) void snap_copy_k(const SnapArgs a)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const size_t stride = (size_t)gridDim.x * BLK;
    for (int q = 0; q < 7; ++q) {
        const f4 *sp = reinterpret_cast<const f4 *>(a.src[q]);
        f4 *dp = reinterpret_cast<f4 *>(a.dst[q]);
        for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < a.len[q] / 4; i += stride) dp[i] = sp[i];      // lengths are multiples of 64
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.st_dst = *a.st_src;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------

struct thip_solver {
    size_t n = 0, m = 0;
    const float *A = nullptr, *b = nullptr, *c = nullptr, *b_rowabs = nullptr;
    thip_param par{};
    int schedule = THIP_SCHED_FUSED;

    thip_allreduce_fn allreduce = nullptr;
    void *allreduce_ctx = nullptr;
    // overlap (thip_solver_set_overlap): 0 in order on the launch stream; 1 the stage's all-reduce on `side` (event in /
    // event out) under the stage's local-row work; 2 column-split pipeline (one_iteration_split); 3 the kernels of 2
    // with the collectives in order (its bitwise reference)
    int overlap = 0;
    hipStream_t side = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    hipEvent_t sev_in[4] = { nullptr, nullptr, nullptr, nullptr }, sev_out[4] = { nullptr, nullptr, nullptr, nullptr };
    size_t n1 = 0;                // split column of modes 2 / 3 (the same on every rank: a function of n); 0: no split
    int rows1 = 0, rows2 = 0;     // chunk rows of partial sums the two half-launches fill under the plan in use
    bool tail_pending = false;    // mode 2: the last column half of the y update + the termination test of the previous
                                  // iteration are still to be enqueued (they wait for its last all-reduce)
    long long spin_ticks = 0;     // thip_test_spin_allreduce
    // hand-offs of mode 2 through device flags instead of events (signal_k / gate_k): [0..3] "producer done", [4..7]
    // "collective done", [8] error word; values = per-slot call counters.  (Folding the signal into the producer's last
    // block and the gate into the consumer's entry was built and measured SLOWER, 848 vs 839 us: an agent-scope acquire
    // in every consumer block invalidates its XCD's L2.)
    bool use_gates = false;
    unsigned *gflags = nullptr;
    unsigned gseq[4] = { 0, 0, 0, 0 };
    long long gate_ticks = 0;

    // optional sparse A (CSR of A and of A^T)
    bool sparse = false; size_t nnz = 0;
    const int64_t *rp = nullptr, *trp = nullptr;
    const int32_t *ci = nullptr, *tci = nullptr;
    const float *sv = nullptr, *tsv = nullptr;
    // ... or ONE tiled copy serving both products (thip_sptile.hip; thip_solver_set_sptile): the slices' shares of the N
    // products go to sw_partH (the buffer the one-pass schedule's m-tail reads), those of the T products to sw_partT
    thip_sptile *spt = nullptr; float *sw_partT = nullptr;

    // cone structure
    std::vector<int32_t> seg_type;
    std::vector<int64_t> seg_len;
    unsigned char *cls = nullptr;                 // per-row class for the element-wise cones
    int64_t *soc_beg = nullptr, *soc_end = nullptr; size_t n_soc = 0, soc_max = 0;
    int64_t *rot_beg = nullptr, *rot_end = nullptr; size_t n_rot = 0, rot_max = 0;
    int64_t *grp_beg = nullptr, *grp_end = nullptr; size_t n_grp = 0, grp_max = 0;
    std::vector<std::pair<int64_t, int64_t>> psd;  // (offset, packed length)
    // PSD cones of order <= 64, grouped by order: one launch projects the x_y and x_s blocks of every cone of a group
    struct PsdGroup { size_t k; int count; int64_t *dev_offs; };
    std::vector<PsdGroup> psd_groups;
    float *psd_work = nullptr; size_t psd_worklen = 0;

    // device vectors (one arena)
    float *arena = nullptr; size_t arena_n = 0;
    float *xx, *xy, *xs, *u, *v, *Tx, *Ty, *Ts, *Su, *Sv, *rxx, *rxy, *rxs;
    float *g1, *h1, *g2, *h2, *g3, *h3, *gP, *hP;
    // Kahan terms of the five iterate vectors (always allocated: O(n + m) floats); passed to the kernels when
    // par.state_arith == THIP_STATE_COMPENSATED, NULL (plain f32 additions) otherwise
    float *kx = nullptr, *ky = nullptr, *ks = nullptr, *ku = nullptr, *kv = nullptr;
    bool comp() const { return par.state_arith == THIP_STATE_COMPENSATED; }
    bool carried_like() const { return schedule == THIP_SCHED_CARRIED || schedule == THIP_SCHED_SWEEP; }
    size_t kahan_n = 0;                              // kx .. kv are contiguous: kahan_n floats from kx
    float *part = nullptr;                           // block partials (4 * EG)
    float *dotc = nullptr;                           // local scalars: [0] c.u, [1] c.rx_x, [2..3] dd,cx
    float *gemv_scr = nullptr; size_t gemv_scr_n = 0;
    GemvHint hint{0, 0}; bool tuned = false; float tuned_ms = 0.0f;
    // storage of A streamed by the iteration: the caller's f32 matrix, or an owned bf16 copy (ld16 = m rounded to 8)
    // (one 16-bit copy at a time: a16_kind says whether A16 holds bf16 or scaled f16; inv_s = the f16 column scales)
    int a_kind = THIP_A_F32; uint16_t *A16 = nullptr; size_t ld16 = 0; bool A16_owned = false; int a16_kind = 0;
    float *inv_s = nullptr; bool inv_s_owned = false;
    GemvHint hint16{0, 0}; bool tuned16 = false; float tuned16_ms = 0.0f;
    // the column-split form (overlap modes 2 / 3) has its own tuned plans: a half-launch has half the workgroups of the
    // whole-matrix launch, so the best tiling differs (the 12 500 x 50 000 shard: ~1k tall tiles unsplit)
    bool split_plan = false;      // the next run streams A as two column-half launches per pass
    GemvHint hint_sp{0, 0}; bool tuned_sp = false; float tuned_sp_ms = 0.0f;
    GemvHint hint16_sp{0, 0}; bool tuned16_sp = false; float tuned16_sp_ms = 0.0f;
    // f32 with m % 16 != 0 (e.g. the k = 500 SDP: m = 125 250; a 12 500-row shard): a library-owned copy with the leading
    // dimension padded to a multiple of 16 floats, made by ensure_apad() when an f32 pass is about to run and the copy
    // fits a third of the free HBM
    float *Apad = nullptr; size_t ldpad = 0;
    int lda_pad = -1;             // thip_solver_set_lda_pad: -1 = default (16 floats, or THIP_LDA_PAD), 0 = never copy
    int autotune = -1;            // thip_solver_set_gemv_autotune: -1 = default (on, or THIP_GEMV_AUTOTUNE), 0 / 1
    bool is16() const { return a_kind != THIP_A_F32; }
    const void *amat() const { return is16() ? (const void *)A16 : (Apad ? (const void *)Apad : (const void *)A); }
    size_t alda() const { return is16() ? ld16 : (Apad ? ldpad : m); }
    // rows m .. alda() - 1 of the matrix in use are zeros written by this library (its padded f32 copy, or a 16-bit copy it made)
    bool apadz() const { return is16() ? A16_owned : Apad != nullptr; }
    const float *ainv() const { return a_kind == THIP_A_F16 ? inv_s : nullptr; }
    const GemvHint *ahint() const
    {
        if (split_plan) return is16() ? (tuned16_sp ? &hint16_sp : nullptr) : (tuned_sp ? &hint_sp : nullptr);
        return is16() ? (tuned16 ? &hint16 : nullptr) : (tuned ? &hint : nullptr);
    }
    // THIP_SCHED_SWEEP (thip_sweep.hip): the second x_x buffer (x_x_{k+1} is formed while x_x_k is still the iterate), its
    // Kahan term, the groups' shares of the N products, the granule ring, the census words, block partials
    float *xx2 = nullptr, *kx2 = nullptr, *xx_home = nullptr, *kx_home = nullptr;
    int xbuf = 0;                 // which of the two buffers s->xx points at (0: the arena's own)
    SweepGeom sgeom{};
    int sweep_state = 0;          // 0 not examined, 1 usable, -1 not usable for this problem / device
    bool sw_first = true;         // the next sweep step starts from a consistent iterate (no u update, no test)
    float *sw_partH = nullptr; unsigned long long *sw_gran = nullptr; unsigned *sw_census = nullptr;
    float *sw_part = nullptr;     // 8 * EG block partials
    unsigned sw_seq = 0, sw_tag = 0;
    float sw_plan_ms = 0.0f;      // the chosen geometry's time per sweep as measured by the plan autotune (0: not tuned)
    // column-sharded sweep (thip_solver_set_column_shard): this rank holds a block of COLUMNS of A (all m rows), the
    // n-vectors are its block, the m-vectors are replicated; one all-reduce per iteration of cs_buf = [A u (mpad) ;
    // A x_x (mpad) ; 4 x EG block partials of the sums over n]
    bool col_shard = false;
    float *cs_buf = nullptr; size_t cs_n = 0;
    size_t sweep_min_bytes = (size_t)128 << 20;     // thip_solver_set_sweep_min_bytes: smaller matrices run the carried schedule
    bool all_soc_short = false;   // every row belongs to a plain second-order cone of <= 129 rows (sw_cone_k)
    bool no_fold = false;         // thip_test_sweep_fault(kind 5): the termination test as a launch of its own in every iteration
    bool status_pending = false;  // the last enqueued iteration's termination test has not been evaluated yet (the next m-kernel's head does)
    int step_par = 0;             // parity of the tau / iter copies the next m-kernel reads
    int pm_par = 0;               // which of the two buffers of sums over m (sw_part + (4 + 4 par) EG) holds the latest
    bool no_merge = false;        // thip_test_sweep_fault(kind 3): the two m-kernels of a step as two launches also without block cones
    int pn_par = 0;               // which of the two buffers of sums over n (sw_part + par * 2 EG) the LAST sweep wrote
    int pub_agent = -1;           // thip_solver_set_sweep_publish: 0 plain stores, 1 agent scope, -1 what the process's self-test said
    // recovery when the persistent kernel gives up (thip_solver_run): a device copy of the consistent iterate of the last
    // completed batch -- x_x, u, (x_y x_s v), their Kahan terms, the status block
    float *snap = nullptr; DevStatus *snap_st = nullptr; long long snap_iter = -1;
    size_t pn_len = 0, pm_len = 0;                  // padded lengths of an n- / m-vector of the arena
    unsigned *hflags = nullptr;                     // pinned: [0] sweep error word [1] gate error [2] one-shot error [3] DevStatus.fault
    int sweep_faults = 0; unsigned sweep_fault_word = 0; long long sweep_fault_iter = -1;
    int fault_kind = 0; long long fault_after = -1; int spin_max = 0;      // thip_test_sweep_fault
    DevStatus *dst = nullptr;
    DevStatus *hst = nullptr;                        // pinned
    bool inited = false;
    bool finalized = false;      // finalize_k has been applied to the terminated iterate
    bool carried_stale = false;  // the stored form of A changed under a running carried schedule: gP / hP must be rebuilt
};

namespace {

size_t pad64(size_t v) { return (v + 63) / 64 * 64; }

// The protocol of a column-sharded run, in ONE place -- thip_solver_run, sweep_prepare, the termination test and col_shard_abort
// (a rank that cannot plan its kernel and has to mirror its peers' collectives) must agree on it, or ranks end up in all-reduces
// of different lengths: the all-reduced buffer [A u (mpad) ; A x_x (mpad) ; 4 x EG sums over n ; fault flag + padding (64)] and
// the number of attempts every rank makes from the same snapshot before all of them return THIP_E_TIMEOUT.
constexpr int COL_SHARD_ATTEMPTS = 3;
inline size_t cs_flag_slot(size_t mpad) { return 2 * mpad + 4 * EG; }
inline size_t cs_floats(size_t mpad) { return cs_flag_slot(mpad) + 64; }

// optional per-launch timing of the dominant kernel (bench.py roofline): HIP event pairs recorded on the
// launch stream around every GEMV kernel of products()
struct Prof {
    bool on = false;
    int period = 1;                 // the spans of every period-th ITERATION are timed (an event pair costs the stream 3-5 us: 9 % of
                                    // a 0.14 ms iteration).  By iteration, not by span: a schedule with two kinds of span per iteration
                                    // (A^T pass then A pass; the two half-launches of the column split; the two products of the tiled
                                    // sparse copy) would otherwise, with an even period, only ever time one kind
    long long seen = 0;             // iterations begun since thip_prof_enable
    bool iter_open = true;          // the spans of the current iteration are timed
    bool open = false;              // the current span is one of the timed ones
    std::vector<hipEvent_t> ev;     // pairs
    size_t used = 0;
    double total_ms = 0.0;
    long long launches = 0;
} g_prof, g_prof_psd;      // the pass over A; the PSD cones' projection chains of an iteration (bench.py's roofline_eig)

void prof_begin(hipStream_t st, Prof &p = g_prof)
{
    if (!p.on) return;
    p.open = p.iter_open;
    if (!p.open) return;
    if (p.used + 2 > p.ev.size()) {
        for (int i = 0; i < 64; ++i) { hipEvent_t e; hipEventCreate(&e); p.ev.push_back(e); }
    }
    hipEventRecord(p.ev[p.used], st);
}
// once per iteration of thip_solver_run
void prof_tick()
{
    for (Prof *p : { &g_prof, &g_prof_psd })
        if (p->on) p->iter_open = (p->seen++ % p->period) == 0;
}
void prof_end(hipStream_t st, Prof &p = g_prof)
{
    if (!p.on || !p.open) return;
    hipEventRecord(p.ev[p.used + 1], st);
    p.used += 2;
    p.open = false;
}

}  // namespace
namespace thip {
void prof_release()
{
    for (Prof *p : { &g_prof, &g_prof_psd }) {
        for (hipEvent_t e : p->ev) hipEventDestroy(e);
        p->ev.clear();
        p->used = 0; p->on = false;
    }
}
}  // namespace thip
namespace {

int do_allreduce(thip_solver *s, float *buf, size_t count)
{
    if (!s->allreduce) return 0;
    const int rc = s->allreduce(s->allreduce_ctx, buf, count, (void *)ctx().stream);
    if (rc != 0) return fail(rc, "all-reduce callback failed", __FILE__, __LINE__);
    return 0;
}

// The all-reduce of a stage, optionally on the solver's side stream (overlap): begin = "the producer kernel has been
// enqueued on the launch stream": the side stream waits for it (event in) and runs the collective; end = the launch
// stream waits for the collective (event out) before the first consumer.  Whatever is enqueued on the launch stream
// between begin and end -- the stage's work on the local rows -- overlaps the collective.  Without overlap the
// collective is enqueued in order on the launch stream (begin) and end is a no-op.
int allreduce_begin(thip_solver *s, float *buf, size_t count)
{
    if (!s->allreduce) return 0;
    if (!(s->overlap == 1 || s->overlap == 2)) return do_allreduce(s, buf, count);
    hipStream_t st = ctx().stream;
    THIP_TRY(hipEventRecord(s->ev_in, st));
    THIP_TRY(hipStreamWaitEvent(s->side, s->ev_in, 0));
    const int rc = s->allreduce(s->allreduce_ctx, buf, count, (void *)s->side);
    if (rc != 0) return fail(rc, "all-reduce callback failed", __FILE__, __LINE__);
    THIP_TRY(hipEventRecord(s->ev_out, s->side));
    return 0;
}
int allreduce_end(thip_solver *s)
{
    if (!s->allreduce || !(s->overlap == 1 || s->overlap == 2)) return 0;
    THIP_TRY(hipStreamWaitEvent(ctx().stream, s->ev_out, 0));
    return 0;
}

unsigned egrid(size_t n) { return grid_for(n, BLK, EG); }

int ensure_gemv_scratch(thip_solver *s)
{
    if (s->gemv_scr || s->sparse || s->m == 0 || s->n == 0) return 0;
    s->gemv_scr_n = 2 * dual_gemv_scratch_floats(s->m, s->n);
    THIP_TRY(hipMalloc((void **)&s->gemv_scr, s->gemv_scr_n * sizeof(float)));
    return 0;
}

// one stage's products as partial sums: N partials of A xn (m), T partials of A^T xt (n).  The fused and carried
// schedules read A once (dual launch); the reference schedule issues the reference's two single GEMVs.
int products(thip_solver *s, const float *xn, const float *xt, GemvPartials *gp, float *hN, float *gT)
{
    hipStream_t st = ctx().stream;
    const int *stop = &s->dst->stop;
    gp->partN = gp->partT = nullptr; gp->nN = gp->nT = 0; gp->strideN = gp->strideT = 0;
    if (s->m == 0 || s->n == 0) return 0;     // zero-sized operator: products are 0 (matop.rs:83-85)
    if (s->spt) {
        // the tiled copy: each product is one pass over the stored entries, its slices added up into a finished vector
        const size_t mp = sptile_pad(s->spt, false), np_ = sptile_pad(s->spt, true);
        prof_begin(st);
        THIP_RC(sptile_product(st, s->spt, false, xn, nullptr, s->sw_partH, 0, stop));
        prof_end(st);
        THIP_RC(finalize_partials(st, s->m, s->sw_partH, sptile_slices(s->spt, false), 2 * mp, 1.0f, 0.0f, hN, nullptr));
        prof_begin(st);
        THIP_RC(sptile_product(st, s->spt, true, xt, nullptr, s->sw_partT, 0, stop));
        prof_end(st);
        THIP_RC(finalize_partials(st, s->n, s->sw_partT, sptile_slices(s->spt, true), 2 * np_, 1.0f, 0.0f, gT, nullptr));
        gp->nN = gp->nT = -1;
        return 0;
    }
    if (s->sparse) {
        // hN = A xn and gT = A^T xt as finished vectors (gathers over the CSR of A and of A^T); nN = nT = -1 tells
        // post_k to take them as they are.  The stop flag is honoured by the consumers (a stray product is harmless).
        (void)stop;
        prof_begin(st);
        THIP_RC(thip_spmv_csr(s->m, s->n, s->nnz, s->rp, s->ci, s->sv, 1.0f, xn, 0.0f, hN, 0));
        THIP_RC(thip_spmv_csr(s->n, s->m, s->nnz, s->trp, s->tci, s->tsv, 1.0f, xt, 0.0f, gT, 0));
        prof_end(st);
        gp->nN = gp->nT = -1;
        return 0;
    }
    if (s->schedule == THIP_SCHED_REFERENCE) {
        GemvPartials a, b;
        const size_t half = s->gemv_scr_n / 2;
        prof_begin(st);
        THIP_RC(dual_gemv_partials(st, s->m, s->n, s->amat(), s->alda(), nullptr, xt, false, true, false, s->gemv_scr, half, &a, stop, s->ahint(), s->a_kind, s->ainv(), s->apadz()));
        prof_end(st);
        prof_begin(st);
        THIP_RC(dual_gemv_partials(st, s->m, s->n, s->amat(), s->alda(), xn, nullptr, true, false, false, s->gemv_scr + half, half, &b, stop, s->ahint(), s->a_kind, s->ainv(), s->apadz()));
        prof_end(st);
        gp->partT = a.partT; gp->nT = a.nT; gp->strideT = a.strideT;
        gp->partN = b.partN; gp->nN = b.nN; gp->strideN = b.strideN;
    } else {
        prof_begin(st);
        THIP_RC(dual_gemv_partials(st, s->m, s->n, s->amat(), s->alda(), xn, xt, true, true, false, s->gemv_scr, s->gemv_scr_n, gp, stop, s->ahint(), s->a_kind, s->ainv(), s->apadz()));
        prof_end(st);
    }
    return 0;
}

int project_blocks(thip_solver *s)
{
    hipStream_t st = ctx().stream;
    const int *stop = &s->dst->stop;
    // SOC / RotSOC / PSD are self-dual (cone_soc.rs:38, cone_psd.rs:56): x_y and x_s get the same projection
    THIP_RC(soc_batched2(st, s->xy, s->xs, s->rxy, s->rxs, s->soc_beg, s->soc_end, s->n_soc, 0, s->soc_max, stop));
    THIP_RC(soc_batched2(st, s->xy, s->xs, s->rxy, s->rxs, s->rot_beg, s->rot_end, s->n_rot, 1, s->rot_max, stop));
    if (!s->psd.empty()) {
        // the x_y and x_s blocks of a cone go through the projection chain together (2 items per launch)
        // the reflection rx <- rx - 2 x of the projected rows rides in the projection kernels' pack when every cone's
        // engine takes it (the polar kernels do)
        bool fold = true;
        for (auto &pr : s->psd) {
            const size_t k = (size_t)((std::sqrt((double)(8 * pr.second + 1)) - 1.0) / 2.0 + 0.5);
            fold = fold && psd_project_takes_rx(k);
        }
        prof_begin(st, g_prof_psd);
        for (auto &g : s->psd_groups)
            THIP_RC(eig_psd_project_small(st, g.k, s->xy, g.dev_offs, g.count, 1, std::sqrt(2.0f), stop, 2, s->xs - s->xy,
                                          fold ? s->rxy : nullptr, s->rxs - s->rxy));
        for (auto &pr : s->psd) {
            const size_t sn = (size_t)pr.second;
            const size_t k = (size_t)((std::sqrt((double)(8 * sn + 1)) - 1.0) / 2.0 + 0.5);
            if (k <= psd_small_max()) continue;           // went with its group
            THIP_RC(eig_psd_project(st, k, s->xy + pr.first, 1, std::sqrt(2.0f), s->par.eps_zero, s->psd_work,
                                    s->psd_worklen, 0, stop, 2, s->xs - s->xy, fold ? s->rxy + pr.first : nullptr,
                                    s->rxs - s->rxy));
        }
        prof_end(st, g_prof_psd);
        if (fold) return 0;
        hipLaunchKernelGGL(rx_psd_k, dim3(egrid(s->m)), dim3(BLK), 0, st, (int)s->m, s->cls, s->xy, s->xs, s->rxy, s->rxs, s->dst);
    }
    return 0;
}

int one_iteration(thip_solver *s)
{
    hipStream_t st = ctx().stream;
    const int n = (int)s->n, m = (int)s->m;
    const unsigned g = egrid(s->n > s->m ? s->n : s->m);
    float *const part = s->part;
    const bool carried = s->carried_like();
    const float ez = s->par.eps_zero;
    GemvPartials gp;
    // post_k leaves its sums as block partials: q0 (over the replicated n-vectors) in `part`, q1..q3 (over the local
    // rows) in `part` too on a single GPU, or -- row-sharded -- in the tail of the n-vector that is all-reduced next,
    // with a fixed grid of NPS blocks so that every rank fills the same NPS slots per sum.
    const bool local = s->allreduce == nullptr;
    const unsigned gq = local ? grid_for(s->n > s->m ? s->n : s->m, 64, PG) : NPS;
    auto shp = [&](float *nvec) { return local ? part : nvec + s->n; };
    const size_t arcount = s->n + TAIL;
    float *const part_y = s->part + 4 * PG;          // ycrit_k's own partials (read by status_k)
    const bool split = !local && (s->overlap == 1 || s->overlap == 2);   // m-part under the all-reduce, n-part after it
    float *const kx = s->comp() ? s->kx : nullptr, *const ky = s->comp() ? s->ky : nullptr;
    float *const ks = s->comp() ? s->ks : nullptr, *const ku = s->comp() ? s->ku : nullptr;
    float *const kv = s->comp() ? s->kv : nullptr;

    auto xupdate = [&](int do_n, int do_m) {
        hipLaunchKernelGGL(xupdate_k, dim3(g), dim3(BLK), 0, st, n, m, s->g1, s->h1, s->c, s->b, s->v, s->Tx, s->Ty, s->Ts,
                           s->cls, s->xx, s->xy, s->xs, s->rxx, s->rxy, s->rxs, s->dst, part, (int)gq, shp(s->g1) + gq, (int)gq,
                           do_n, do_m, do_n, kx, ky, ks);
    };
    // ---- stage X: x update (solver.rs:538-555) ----------------------------------------------------
    THIP_RC(products(s, s->u, s->v, &gp, s->h1, s->g1));
    hipLaunchKernelGGL(post_k, dim3(gq), dim3(BLK), 0, st, n, m, gp.partT, gp.nT, gp.strideT, s->g1, gp.partN, gp.nN,
                       gp.strideN, s->h1, s->c, s->u, s->b, s->v, 0, (const float *)nullptr, (const float *)nullptr,
                       (const float *)nullptr, ez, part, shp(s->g1), s->dst, 1);
    THIP_RC(allreduce_begin(s, s->g1, arcount));
    if (split) {
        xupdate(0, 1);
        THIP_RC(project_blocks(s));      // the block cones live on the local rows
        THIP_RC(allreduce_end(s));
        xupdate(1, 0);
    } else {
        THIP_RC(allreduce_end(s));
        xupdate(1, 1);
        THIP_RC(project_blocks(s));
    }

    // ---- stage Y: y update from K rx (solver.rs:557-567), own products unless carried ---------------
    if (!carried) {
        THIP_RC(products(s, s->rxx, s->rxy, &gp, s->h2, s->g2));
        hipLaunchKernelGGL(post_k, dim3(gq), dim3(BLK), 0, st, n, m, gp.partT, gp.nT, gp.strideT, s->g2, gp.partN, gp.nN,
                           gp.strideN, s->h2, s->c, s->rxx, s->b, s->rxy, 0, (const float *)nullptr,
                           (const float *)nullptr, (const float *)nullptr, ez, part, shp(s->g2), s->dst, 1);
        auto yupdate = [&](int do_n, int do_m) {
            hipLaunchKernelGGL(ycrit_k, dim3(g), dim3(BLK), 0, st, n, m, 1, 0, 0, (const float *)nullptr,
                               (const float *)nullptr, (float *)nullptr, (float *)nullptr, s->g2, s->h2, s->c, s->b, s->rxs,
                               s->Su, s->Sv, s->u, s->v, s->xx, ez, part_y, s->dst, part, (int)gq, shp(s->g2) + gq, (int)gq,
                               do_n, do_m, do_n, 0, (int)g, ku, kv);
        };
        THIP_RC(allreduce_begin(s, s->g2, arcount));
        if (split) { yupdate(0, 1); THIP_RC(allreduce_end(s)); yupdate(1, 0); }
        else       { THIP_RC(allreduce_end(s)); yupdate(1, 1); }
    }

    // ---- stage C: criteria products of the new iterate (solver.rs:573-656) --------------------------
    THIP_RC(products(s, s->xx, s->xy, &gp, s->h3, s->g3));
    // block partials: q0 = c.rx_x, q1 = b.rx_y (carried), q2 = ||p||^2, q3 = b.x_y
    hipLaunchKernelGGL(post_k, dim3(gq), dim3(BLK), 0, st, n, m, gp.partT, gp.nT, gp.strideT, s->g3, gp.partN, gp.nN,
                       gp.strideN, s->h3, carried ? s->c : (const float *)nullptr, s->rxx,
                       carried ? s->b : (const float *)nullptr, s->rxy, 1, s->xs, s->xy, s->b, ez, part, shp(s->g3), s->dst, 1);
    auto ycrit = [&](int do_n, int do_m) {
        hipLaunchKernelGGL(ycrit_k, dim3(g), dim3(BLK), 0, st, n, m, carried ? 1 : 0, 1, 1, s->g3, s->h3, s->gP, s->hP,
                           (const float *)nullptr, (const float *)nullptr, s->c, s->b, s->rxs, s->Su, s->Sv, s->u, s->v, s->xx,
                           ez, part_y, s->dst, part, (int)gq, shp(s->g3) + gq, (int)gq, do_n, do_m, do_n, 0, (int)g, ku, kv);
    };
    THIP_RC(allreduce_begin(s, s->g3, arcount));
    if (split && carried) { ycrit(0, 1); THIP_RC(allreduce_end(s)); ycrit(1, 0); }
    else                  { THIP_RC(allreduce_end(s)); ycrit(1, 1); }
    {
        const StatArgs sa{ (int)g, part_y, shp(s->g3) + 2 * gq, shp(s->g3) + 3 * gq, (int)gq, nullptr, 0, nullptr, 0, nullptr,
                           s->par.eps_acc, s->par.eps_inf, ez, (long long)s->par.max_iter, s->xbuf };
        hipLaunchKernelGGL(status_k, dim3(1), dim3(BLK), 0, st, sa, s->dst);
    }
    THIP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Column-split pipeline for row-sharded runs (thip_solver_set_overlap(s, 2); carried schedule, dense A).
//
// A stage's products only need, column range by column range, the all-reduced entries of the n-vector they multiply
// (solver.rs:146 vs 149, 122 vs 125: the N product of columns [a, b) reads x[a .. b) only).  So every stage runs as two
// launches over the column halves H1 = [0, n1) and H2 = [n1, n), and the all-reduce of a half's A^T y travels on the
// side stream while the NEXT half-launch streams its part of A:
//
//   X.1  gemv(H1; u, v)  post -> g1[H1]                         AR(g1[H1])   ---------.
//        [tail of the previous iteration: wait AR(g3[H2]) ; u[H2], kappa, criteria ; termination test]
//   X.2  gemv(H2; u, v)  post -> g1[H2], h1, b.v partials       AR(g1[H2] + tail) ----|--.
//        wait AR(g1[H1]) <----------------------------------------------------------'  |
//        x_x[H1], x_y, x_s, block cones                                                 |
//   C.1  gemv(H1; x_x, x_y)  post -> g3[H1]                     AR(g3[H1])   ---------. |
//        wait AR(g1[H2]) <------------------------------------------------------------|-'
//        x_x[H2], tau                                                                  |
//   C.2  gemv(H2; x_x, x_y)  post -> g3[H2], h3, criteria sums  AR(g3[H2] + tail) ----|--.  (consumed by the tail above)
//        wait AR(g3[H1]) <------------------------------------------------------------'
//        u[H1], v
//
// Every collective has a whole half-launch (0.19 ms on a 1/8 shard of BASELINE configs[2]) to complete in.  The sharded
// block partials ride with the second half.  The termination test of iteration k is enqueued after the first
// half-launch of iteration k + 1; that launch and its reduction only write scratch (partial sums, g1[H1]), so the
// iterate is still exactly the one of stopping at iteration k.  The split column n1 is a function of n alone -- every
// rank must issue collectives of the same lengths, whatever plan its own autotune picked; each half-launch has its own
// column chunks and fills consecutive chunk rows of the partial sums (dual_gemv_partials_cols).  Mode 3 enqueues the same
// kernels with the collectives in order and no skew: the iterates of modes 2 and 3 are bitwise equal
// (tests/test_gpu_sharded.py).
// ---------------------------------------------------------------------------------------------------
// The split column is the SAME on every rank (it sets the lengths of the collectives): a function of n alone, never of a
// rank's own tuned plan.  A multiple of 8 floats keeps every buffer offset 32-byte aligned.
size_t split_column(const thip_solver *s)
{
    if (s->sparse || s->m == 0 || s->n < 16) return 0;
    return (s->n / 2) & ~(size_t)7;
}

// chunk rows the two half-launches fill under the plan in use
void split_rows(thip_solver *s)
{
    const bool h16 = s->is16();
    const bool vec_ok = (((uintptr_t)s->amat() & 15u) == 0) && (s->alda() % (h16 ? 8 : 4) == 0);
    s->rows1 = dual_gemv_chunk_rows(s->m, s->n1, vec_ok, s->a_kind, s->ahint(), nullptr);
    s->rows2 = dual_gemv_chunk_rows(s->m, s->n - s->n1, vec_ok, s->a_kind, s->ahint(), nullptr);
}

bool split_active(const thip_solver *s)
{
    return s->split_plan && s->n1 > 0 && s->n1 < s->n;
}

int autotune_gemv(thip_solver *s);
bool sweep_active(const thip_solver *s);

// decides the form of the next run: column-split (its own tuned plan, its split column) or one launch per pass
int prepare_split(thip_solver *s)
{
    s->split_plan = s->overlap >= 2 && s->allreduce != nullptr && !s->sparse && s->carried_like()
                    && s->m > 0 && s->n > 0;
    s->n1 = 0;
    if (s->inited && !sweep_active(s)) THIP_RC(autotune_gemv(s));        // once per stored form and launch form (a no-op afterwards)
    if (!s->split_plan) return 0;
    s->n1 = split_column(s);
    if (s->n1 == 0 || s->n1 >= s->n) { s->split_plan = false; s->n1 = 0; return 0; }
    split_rows(s);
    return 0;
}

int ar_begin(thip_solver *s, int slot, float *buf, size_t count)
{
    if (s->overlap != 2) return do_allreduce(s, buf, count);
    hipStream_t st = ctx().stream;
    if (s->use_gates) {
        const unsigned v = ++s->gseq[slot];
        hipLaunchKernelGGL(signal_k, dim3(1), dim3(1), 0, st, s->gflags + slot, v);
        hipLaunchKernelGGL(gate_k, dim3(1), dim3(1), 0, s->side, s->gflags + slot, v, s->gate_ticks, s->gflags + 8, &s->dst->stop);
        const int rc = s->allreduce(s->allreduce_ctx, buf, count, (void *)s->side);
        if (rc != 0) return fail(rc, "all-reduce callback failed", __FILE__, __LINE__);
        hipLaunchKernelGGL(signal_k, dim3(1), dim3(1), 0, s->side, s->gflags + 4 + slot, v);
        THIP_LAUNCH_CHECK();
        return 0;
    }
    THIP_TRY(hipEventRecord(s->sev_in[slot], st));
    THIP_TRY(hipStreamWaitEvent(s->side, s->sev_in[slot], 0));
    const int rc = s->allreduce(s->allreduce_ctx, buf, count, (void *)s->side);
    if (rc != 0) return fail(rc, "all-reduce callback failed", __FILE__, __LINE__);
    THIP_TRY(hipEventRecord(s->sev_out[slot], s->side));
    return 0;
}

int ar_wait(thip_solver *s, int slot)
{
    if (s->overlap != 2) return 0;
    if (s->use_gates) {
        hipLaunchKernelGGL(gate_k, dim3(1), dim3(1), 0, ctx().stream, s->gflags + 4 + slot, s->gseq[slot], s->gate_ticks,
                           s->gflags + 8, &s->dst->stop);
        THIP_LAUNCH_CHECK();
        return 0;
    }
    THIP_TRY(hipStreamWaitEvent(ctx().stream, s->sev_out[slot], 0));
    return 0;
}

int products_cols(thip_solver *s, const float *xn, const float *xt, GemvPartials *gp, int half)
{
    hipStream_t st = ctx().stream;
    prof_begin(st);
    THIP_RC(dual_gemv_partials_cols(st, s->m, s->n, s->amat(), s->alda(), xn, xt, true, true, s->gemv_scr, s->gemv_scr_n, gp,
                                    &s->dst->stop, s->ahint(), s->a_kind, s->ainv(), s->apadz(), half ? s->n1 : 0,
                                    half ? s->n : s->n1, half ? s->rows1 : 0, s->rows1 + s->rows2, nullptr));
    prof_end(st);
    return 0;
}

struct SplitCtx {
    thip_solver *s; hipStream_t st; int m; unsigned g; size_t n1, n2;
    float *partX, *partC, *part_y, *kx, *ky, *ks, *ku, *kv; float ez;
};

SplitCtx split_ctx(thip_solver *s)
{
    SplitCtx c;
    c.s = s; c.st = ctx().stream; c.m = (int)s->m; c.g = egrid(s->n > s->m ? s->n : s->m);
    c.n1 = s->n1; c.n2 = s->n - s->n1;
    c.partX = s->part; c.partC = s->part + 2 * NPS; c.part_y = s->part + 4 * PG;
    const bool k = s->comp();
    c.kx = k ? s->kx : nullptr; c.ky = k ? s->ky : nullptr; c.ks = k ? s->ks : nullptr;
    c.ku = k ? s->ku : nullptr; c.kv = k ? s->kv : nullptr;
    c.ez = s->par.eps_zero;
    return c;
}

// y update of one column range of u (+ the v rows, + kappa) and that range's share of the criteria sums
void split_ycrit(const SplitCtx &c, int half, int do_m, int do_kappa)
{
    thip_solver *s = c.s;
    const size_t c0 = half ? c.n1 : 0, len = half ? c.n2 : c.n1;
    hipLaunchKernelGGL(ycrit_k, dim3(c.g), dim3(BLK), 0, c.st, (int)len, c.m, 1, 1, 1, s->g3 + c0, s->h3, s->gP + c0, s->hP,
                       (const float *)nullptr, (const float *)nullptr, s->c + c0, s->b, s->rxs, s->Su + c0, s->Sv, s->u + c0,
                       s->v, s->xx + c0, c.ez, c.part_y, s->dst, c.partC, (int)(2 * NPS), s->g3 + s->n + NPS, (int)NPS,
                       1, do_m, do_kappa, half * (int)c.g, 2 * (int)c.g, c.ku ? c.ku + c0 : (float *)nullptr, c.kv);
}

// what iteration k leaves for after its last all-reduce: u[H2], kappa, the H2 share of the criteria, the termination test
int split_tail(thip_solver *s)
{
    const SplitCtx c = split_ctx(s);
    THIP_RC(ar_wait(s, 3));
    split_ycrit(c, 1, 0, 1);
    {
        const StatArgs sa{ 2 * (int)c.g, c.part_y, s->g3 + s->n + 2 * NPS, s->g3 + s->n + 3 * NPS, (int)NPS, nullptr, 0, nullptr, 0, nullptr,
                           s->par.eps_acc, s->par.eps_inf, c.ez, (long long)s->par.max_iter, s->xbuf };
        hipLaunchKernelGGL(status_k, dim3(1), dim3(BLK), 0, c.st, sa, s->dst);
    }
    THIP_LAUNCH_CHECK();
    s->tail_pending = false;
    return 0;
}

int one_iteration_split(thip_solver *s)
{
    const SplitCtx c = split_ctx(s);
    hipStream_t st = c.st;
    const int m = c.m;
    const size_t n = s->n, n1 = c.n1, n2 = c.n2;
    GemvPartials gp;
    // second reduction stage of one column range (+ the m-part and the sharded sums with the last range)
    auto post = [&](float *gvec, float *hvec, int half, const float *dn_b, const float *dm_b, int crit, float *prep) {
        const size_t c0 = half ? n1 : 0, len = half ? n2 : n1;
        hipLaunchKernelGGL(post_k, dim3(NPS), dim3(BLK), 0, st, (int)len, m, gp.partT + c0, gp.nT, gp.strideT, gvec + c0,
                           gp.partN, gp.nN, gp.strideN, hvec, s->c + c0, dn_b + c0, s->b, dm_b, crit, s->xs, s->xy, s->b,
                           c.ez, prep + (size_t)half * NPS, gvec + n, s->dst, half);
    };
    auto xupd = [&](int half, int do_m, int do_tau) {
        const size_t c0 = half ? n1 : 0, len = half ? n2 : n1;
        hipLaunchKernelGGL(xupdate_k, dim3(c.g), dim3(BLK), 0, st, (int)len, m, s->g1 + c0, s->h1, s->c + c0, s->b, s->v,
                           s->Tx + c0, s->Ty, s->Ts, s->cls, s->xx + c0, s->xy, s->xs, s->rxx + c0, s->rxy, s->rxs, s->dst,
                           c.partX, (int)(2 * NPS), s->g1 + n + NPS, (int)NPS, 1, do_m, do_tau,
                           c.kx ? c.kx + c0 : (float *)nullptr, c.ky, c.ks);
    };

    // ---- stage X, first half ----
    THIP_RC(products_cols(s, s->u, s->v, &gp, 0));
    post(s->g1, s->h1, 0, s->u, s->v, 0, c.partX);
    THIP_RC(ar_begin(s, 0, s->g1, n1));
    if (s->tail_pending) THIP_RC(split_tail(s));           // the previous iteration ends here
    // ---- stage X, second half ----
    THIP_RC(products_cols(s, s->u, s->v, &gp, 1));
    post(s->g1, s->h1, 1, s->u, s->v, 0, c.partX);
    THIP_RC(ar_begin(s, 1, s->g1 + n1, n2 + TAIL));
    THIP_RC(ar_wait(s, 0));
    xupd(0, 1, 0);
    THIP_RC(project_blocks(s));
    // ---- stage C, first half ----
    THIP_RC(products_cols(s, s->xx, s->xy, &gp, 0));
    post(s->g3, s->h3, 0, s->rxx, s->rxy, 0, c.partC);
    THIP_RC(ar_begin(s, 2, s->g3, n1));
    THIP_RC(ar_wait(s, 1));
    xupd(1, 0, 1);
    // ---- stage C, second half ----
    THIP_RC(products_cols(s, s->xx, s->xy, &gp, 1));
    post(s->g3, s->h3, 1, s->rxx, s->rxy, 1, c.partC);
    THIP_RC(ar_begin(s, 3, s->g3 + n1, n2 + TAIL));
    THIP_RC(ar_wait(s, 2));
    split_ycrit(c, 0, 1, 0);
    THIP_LAUNCH_CHECK();
    s->tail_pending = true;
    if (s->overlap != 2) THIP_RC(split_tail(s));           // in order: no skew
    return 0;
}

// Times the candidate tilings of the dual GEMV on THIS matrix (two launches each, the second one timed with HIP
// events) and keeps the fastest: a handful of passes over A, once per solve.  THIP_GEMV_AUTOTUNE=0 disables it.
int autotune_gemv(thip_solver *s)
{
    const char *env = getenv("THIP_GEMV_AUTOTUNE");
    if (s->autotune == 0 || (s->autotune < 0 && env && atoi(env) == 0) || getenv("THIP_GEMV_NJ") || getenv("THIP_GEMV_BLOCKS")) return 0;
    if (s->sparse || s->m * s->n < (size_t)1 << 22) return 0;   // sparse, or tiny: nothing to tune
    const bool b16 = s->is16(), sp = s->split_plan;
    if (sp ? (b16 ? s->tuned16_sp : s->tuned_sp) : (b16 ? s->tuned16 : s->tuned)) return 0;
    hipStream_t st = ctx().stream;
    hipEvent_t e0, e1;
    THIP_TRY(hipEventCreate(&e0));
    THIP_TRY(hipEventCreate(&e1));
    int nc = 0;
    const GemvHint *c = gemv_candidates(&nc);
    float best = 1e30f;
    GemvPartials gp;
    GemvHint pick{0, 0};
    // one pass in the form the iteration will use: one launch, or (split) two launches over the column halves of the
    // candidate's own chunking
    auto one_pass = [&](const GemvHint *h) -> int {
        if (!sp)
            return dual_gemv_partials(st, s->m, s->n, s->amat(), s->alda(), s->u, s->v, true, true, false, s->gemv_scr,
                                      s->gemv_scr_n, &gp, nullptr, h, s->a_kind, s->ainv(), s->apadz());
        const bool h16 = s->is16();
        const bool vec_ok = (((uintptr_t)s->amat() & 15u) == 0) && (s->alda() % (h16 ? 8 : 4) == 0);
        const size_t n1 = split_column(s);
        if (n1 == 0 || n1 >= s->n)
            return dual_gemv_partials(st, s->m, s->n, s->amat(), s->alda(), s->u, s->v, true, true, false, s->gemv_scr,
                                      s->gemv_scr_n, &gp, nullptr, h, s->a_kind, s->ainv(), s->apadz());
        const int r1 = dual_gemv_chunk_rows(s->m, n1, vec_ok, s->a_kind, h, nullptr);
        const int r2 = dual_gemv_chunk_rows(s->m, s->n - n1, vec_ok, s->a_kind, h, nullptr);
        THIP_RC(dual_gemv_partials_cols(st, s->m, s->n, s->amat(), s->alda(), s->u, s->v, true, true, s->gemv_scr, s->gemv_scr_n,
                                        &gp, nullptr, h, s->a_kind, s->ainv(), s->apadz(), 0, n1, 0, r1 + r2, nullptr));
        THIP_RC(dual_gemv_partials_cols(st, s->m, s->n, s->amat(), s->alda(), s->u, s->v, true, true, s->gemv_scr, s->gemv_scr_n,
                                        &gp, nullptr, h, s->a_kind, s->ainv(), s->apadz(), n1, s->n, r1, r1 + r2, nullptr));
        return 0;
    };
    for (int w = 0; w < 3; ++w)         // clocks and caches settle before anything is timed
        THIP_RC(one_pass(nullptr));
    // inputs: the iterate if the loop is already running (storage switch), else zeros -- timing does not depend on them
    for (int i = 0; i < nc; ++i) {
        float ms = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            THIP_TRY(hipEventRecord(e0, st));
            THIP_RC(one_pass(&c[i]));
            // the second reduction stage is part of the price of a plan (finer grids leave more partials to post_k):
            // time it too, into g2 / h2, which every schedule rewrites before reading
            THIP_RC(finalize_partials(st, s->m, gp.partN, gp.nN, gp.strideN, 1.0f, 0.0f, s->h2, nullptr));
            THIP_RC(finalize_partials(st, s->n, gp.partT, gp.nT, gp.strideT, 1.0f, 0.0f, s->g2, nullptr));
            THIP_TRY(hipEventRecord(e1, st));
            THIP_TRY(hipEventSynchronize(e1));
            float t = 0.0f;
            THIP_TRY(hipEventElapsedTime(&t, e0, e1));
            if (rep > 0 && t < ms) ms = t;
        }
        if (ms < best) { best = ms; pick = c[i]; }
    }
    if (sp) {
        if (b16) { s->hint16_sp = pick; s->tuned16_sp = true; s->tuned16_sp_ms = best; }
        else     { s->hint_sp = pick; s->tuned_sp = true; s->tuned_sp_ms = best; }
    } else {
        if (b16) { s->hint16 = pick; s->tuned16 = true; s->tuned16_ms = best; }
        else     { s->hint = pick; s->tuned = true; s->tuned_ms = best; }
    }
    THIP_TRY(hipEventDestroy(e0));
    THIP_TRY(hipEventDestroy(e1));
    return 0;
}

// The carried schedule keeps gP = A^T x_y and hP = A x_x of the current iterate (ycrit_k).  After the stored form of A
// has been switched inside a solve (16-bit passes first, f32 passes to finish) they still hold the products with the
// OLD matrix: the first y-update would mix A_old and A_new, a one-off error of (A_old - A)^T x ~ 2^-9 .. 2^-12 relative,
// far above the step size near convergence.  Recompute them with the matrix now in use: one pass over A.
int rebuild_carried(thip_solver *s)
{
    s->carried_stale = false;
    if (!s->carried_like() || s->m == 0 || s->n == 0) return 0;
    hipStream_t st = ctx().stream;
    GemvPartials gp;
    THIP_RC(products(s, s->xx, s->xy, &gp, s->hP, s->gP));
    if (gp.nN >= 0) {      // dense: finish the partial sums (the sparse products are finished vectors already)
        THIP_RC(finalize_partials(st, s->m, gp.partN, gp.nN, gp.strideN, 1.0f, 0.0f, s->hP, nullptr));
        THIP_RC(finalize_partials(st, s->n, gp.partT, gp.nT, gp.strideT, 1.0f, 0.0f, s->gP, nullptr));
    }
    THIP_RC(do_allreduce(s, s->gP, s->n));
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// THIP_SCHED_SWEEP on the host
// ---------------------------------------------------------------------------------------------------
// Can the next run use the one-pass kernel?  Dense f32 A on one GPU in a shape sweep_plan() takes, on a device whose
// placement census came out as 8 x 32 in a dry run.  Examined once per (re)initialisation.
int sweep_pass(thip_solver *s, int first, int np_m = 0, bool timing = false);

// the partial-sum buffers of a solver on the tiled sparse copy (zeroed once: a block without entries is never written) and the
// words the sweep schedule's host side reads (error word, block partials)
int spt_buffers(thip_solver *s)
{
    hipStream_t st = ctx().stream;
    if (!s->sw_partT) {
        const size_t fh = std::max<size_t>(sptile_part_floats(s->spt, false), 64), ft = std::max<size_t>(sptile_part_floats(s->spt, true), 64);
        if (s->sw_partH) { THIP_TRY(hipFree(s->sw_partH)); s->sw_partH = nullptr; }
        THIP_TRY(hipMalloc((void **)&s->sw_partH, fh * sizeof(float)));
        THIP_TRY(hipMalloc((void **)&s->sw_partT, ft * sizeof(float)));
        THIP_TRY(hipMemsetAsync(s->sw_partH, 0, fh * sizeof(float), st));
        THIP_TRY(hipMemsetAsync(s->sw_partT, 0, ft * sizeof(float), st));
    }
    if (!s->sw_census) {
        THIP_TRY(hipMalloc((void **)&s->sw_census, 64 * sizeof(unsigned)));
        THIP_TRY(hipMalloc((void **)&s->sw_part, 12 * EG * sizeof(float)));
        THIP_TRY(hipMemsetAsync(s->sw_census, 0, 64 * sizeof(unsigned), st));
        THIP_TRY(hipMemsetAsync(s->sw_part, 0, 12 * EG * sizeof(float), st));
    }
    return 0;
}

int sweep_prepare(thip_solver *s)
{
    if (s->spt) {
        // the tiled sparse copy: no persistent kernel, no placement census, no geometry to time -- the one-pass recurrence
        // in three launches (sweep_pass) whenever it was asked for on one GPU
        THIP_RC(spt_buffers(s));
        if (s->schedule != THIP_SCHED_SWEEP || s->allreduce != nullptr || s->col_shard || s->m == 0 || s->n == 0) return 0;
        if (s->sweep_state != 0) return 0;
        s->sw_first = true;
        s->sgeom = SweepGeom{};
        s->sgeom.ngroups = sptile_slices(s->spt, false);
        s->sgeom.mpad = sptile_pad(s->spt, false);
        s->sgeom.m_eff = (int)s->m;
        s->sw_plan_ms = 0.0f;
        s->sweep_state = 1;
        return 0;
    }
    if (s->schedule != THIP_SCHED_SWEEP) return 0;
    if ((s->allreduce != nullptr) != s->col_shard) return 0;      // row shards run the carried schedule; column shards need the hook
    if (s->sparse || s->m == 0 || s->n == 0) return 0;      // not now (may change)
    if (s->sweep_state != 0) return 0;
    s->sweep_state = -1;
    // a (re-)plan restarts the schedule from the consistent iterate: the timing sweeps below rewrite the groups' shares and
    // the granule ring, so whatever a previous run left of them is gone (a first = 1 sweep rebuilds all of it)
    s->sw_first = true;
    static const int env_off = getenv("THIP_SWEEP_OFF") ? atoi(getenv("THIP_SWEEP_OFF")) : 0;
    if (env_off) return 0;
    size_t m_eff = s->m;
    const int elem = s->a_kind;                   // THIP_A_F32, or the 16-bit form the iteration streams now
    const size_t esize = elem ? 2 : 4;
    // a library-owned padded copy has zero rows behind row m, and every m-vector of the arena has zeros behind entry m
    if (!elem && m_eff % 4 != 0 && s->Apad != nullptr && s->ldpad >= (m_eff + 3) / 4 * 4) m_eff = (m_eff + 3) / 4 * 4;
    if (elem && m_eff % 8 != 0 && s->A16_owned && s->ld16 >= (m_eff + 7) / 8 * 8) m_eff = (m_eff + 7) / 8 * 8;
    if (!s->col_shard && s->m * s->n * esize < s->sweep_min_bytes) return 0;
    // how the partial dots are published in this process: decided HERE, at plan time (the self-test allocates 64 MB and
    // synchronises; left to the first sweep_pass it ran in the middle of the first batch when the autotune is off)
    (void)sweep_publish_default();
    // the geometries the kernel offers for this matrix (group size, columns per panel); THIP_SWEEP_CLASS pins one
    SweepGeom cand[6];
    int nc = 0;
    if (getenv("THIP_SWEEP_CLASS")) { if (sweep_plan(m_eff, s->n, s->alda(), s->amat(), &cand[0], elem) == 0) nc = 1; }
    else nc = sweep_candidates(m_eff, s->n, s->alda(), s->amat(), cand, 6, elem);
    if (nc == 0) return 0;
    hipStream_t st = ctx().stream;
    if (!s->sw_census) {
        THIP_TRY(hipMalloc((void **)&s->sw_census, 64 * sizeof(unsigned)));
        THIP_TRY(hipMalloc((void **)&s->sw_part, 12 * EG * sizeof(float)));
    }
    THIP_TRY(hipMemsetAsync(s->sw_census, 0, 64 * sizeof(unsigned), st));
    s->sw_seq = 0;
    THIP_RC(sweep_census_dry_run(st, s->sw_census, s->sw_seq++));
    unsigned hc[10];
    THIP_TRY(hipMemcpyAsync(hc, s->sw_census, sizeof(hc), hipMemcpyDeviceToHost, st));
    THIP_TRY(hipStreamSynchronize(st));
    if (s->fault_kind == 1) { hc[9] = 2u; s->fault_kind = 0; }      // TEST HOOK: "the placement is not 8 x 32"
    if (hc[9] != 0u) return 0;                      // not 32 workgroups per XCD: the carried schedule runs
    size_t maxH = 0, maxG = 0;
    for (int c = 0; c < nc; ++c) {
        cand[c].m_eff = (int)m_eff;
        maxH = std::max(maxH, (size_t)cand[c].ngroups * 2 * cand[c].mpad);
        maxG = std::max(maxG, sweep_gran_words(cand[c]));
    }
    if (s->sw_partH) { THIP_TRY(hipFree(s->sw_partH)); s->sw_partH = nullptr; }
    if (s->sw_gran) { THIP_TRY(hipFree(s->sw_gran)); s->sw_gran = nullptr; }
    THIP_TRY(hipMalloc((void **)&s->sw_partH, maxH * sizeof(float)));
    THIP_TRY(hipMalloc((void **)&s->sw_gran, maxG * sizeof(unsigned long long)));
    THIP_TRY(hipMemsetAsync(s->sw_partH, 0, maxH * sizeof(float), st));
    THIP_TRY(hipMemsetAsync(s->sw_gran, 0, maxG * sizeof(unsigned long long), st));
    SweepGeom g = cand[0];
    s->sw_plan_ms = 0.0f;
    const char *env_at = getenv("THIP_GEMV_AUTOTUNE");
    const bool tune = !(s->autotune == 0 || (s->autotune < 0 && env_at && atoi(env_at) == 0));
    if (tune) {
        // time every geometry on the actual matrix, like the GEMV plans: idempotent sweeps (first = 1: u stays, x_x goes
        // to the buffer that is not the iterate, gP is rewritten with what it has to hold anyway); one warm-up, five timed
        // one by one
        hipEvent_t e0, e1;
        THIP_TRY(hipEventCreate(&e0));
        THIP_TRY(hipEventCreate(&e1));
        constexpr int REP = 7;
        float med[6], spread[6], cost[6];
        for (int c = 0; c < nc; ++c) {
            s->sgeom = cand[c];
            THIP_TRY(hipMemsetAsync(s->sw_gran, 0, maxG * sizeof(unsigned long long), st));
            THIP_RC(sweep_pass(s, 1, 0, true));
            float t[REP];
            for (int r = 0; r < REP; ++r) {
                THIP_TRY(hipEventRecord(e0, st));
                THIP_RC(sweep_pass(s, 1, 0, true));
                THIP_TRY(hipEventRecord(e1, st));
                THIP_TRY(hipEventSynchronize(e1));
                THIP_TRY(hipEventElapsedTime(&t[r], e0, e1));
            }
            unsigned err = 0;
            THIP_TRY(hipMemcpy(&err, s->sw_census + 9, sizeof(err), hipMemcpyDeviceToHost));
            if (err != 0u) {
                // a ticket was off or a spin ran out: the carried schedule runs.  Inside a running solve (a re-plan after
                // thip_solver_set_sweep_min_bytes) the timing sweeps have rewritten gP: rebuilt before the next step
                hipEventDestroy(e0); hipEventDestroy(e1);
                s->carried_stale = true;
                return 0;
            }
            std::sort(t, t + REP);
            med[c] = t[REP / 2];
            spread[c] = (t[REP - 1] - t[0]) / t[REP / 2];
            // what a geometry costs OUTSIDE the kernel: the next step's m-tail reads every group's share of the two N
            // products (2 m floats per group; 20 MB at the 10 000-variable LP with 128 groups) at a few TB/s
            cost[c] = med[c] + (float)((double)cand[c].ngroups * 2.0 * (double)s->m * sizeof(float) / 3.0e12 * 1e3);
        }
        hipEventDestroy(e0); hipEventDestroy(e1);
        // the default geometry (candidate 0) keeps its place unless another one wins by more than repeated sweeps of either
        // differ among themselves (at least 2 %): a geometry that flips with the noise flips the order of the sums -- the
        // bits of every later iterate -- with it
        int bi = 0;
        for (int c = 1; c < nc; ++c) {
            const float margin = std::max(0.02f, std::max(spread[bi], spread[c]));
            if (cost[c] < cost[bi] * (1.0f - margin)) bi = c;
        }
        g = cand[bi];
        s->sw_plan_ms = med[bi];
        THIP_TRY(hipMemsetAsync(s->sw_gran, 0, maxG * sizeof(unsigned long long), st));
        // a safety net for shapes the kernel takes badly (e.g. very few rows per workgroup): a sweep that is not faster
        // than the two passes of the carried schedule gives way to it -- unless the caller asked for the one-pass schedule
        // "whenever the kernel can take the shape" (sweep_min_bytes = 0).  The two passes are priced at the dual GEMV's
        // usual 6.2 TB/s first, and when that is anywhere near, with the carried plan's own MEASURED pass
        if (!s->col_shard && s->sweep_min_bytes != 0) {
            double carried_ms = 2.0 * (double)s->m * (double)s->n * esize / 6.2e12 * 1e3 + 0.03;
            if ((double)s->sw_plan_ms > 0.75 * carried_ms) {
                THIP_RC(autotune_gemv(s));
                const float pass_ms = elem ? (s->tuned16 ? s->tuned16_ms : 0.0f) : (s->tuned ? s->tuned_ms : 0.0f);
                if (pass_ms > 0.0f) carried_ms = 2.0 * (double)pass_ms + 0.03;
            }
            if ((double)s->sw_plan_ms > carried_ms) return 0;
        }
    }
    if (s->col_shard) {
        const size_t need = cs_floats(g.mpad);            // [A u ; A x_x ; 4 x EG sums over n ; "my kernel gave up" flag]
        if (s->cs_n != need) {
            if (s->cs_buf) { THIP_TRY(hipFree(s->cs_buf)); s->cs_buf = nullptr; }
            THIP_TRY(hipMalloc((void **)&s->cs_buf, need * sizeof(float)));
            s->cs_n = need;
        }
        THIP_TRY(hipMemsetAsync(s->cs_buf, 0, need * sizeof(float), st));
    }
    s->sgeom = g;
    s->sgeom.m_eff = (int)m_eff;
    s->sweep_state = 1;
    return 0;
}

bool sweep_active(const thip_solver *s)
{
    if (s->spt) return s->schedule == THIP_SCHED_SWEEP && s->sweep_state == 1 && s->allreduce == nullptr && !s->col_shard;
    if (!(s->schedule == THIP_SCHED_SWEEP && s->sweep_state == 1 && (s->allreduce != nullptr) == s->col_shard && !s->sparse
          && s->sgeom.elem == s->a_kind)) return false;       // (planned for the stored form of A in use now)
    // planned on the padded copy (m not a multiple of the rows per slot): only while that copy is the matrix in use
    if ((size_t)s->sgeom.m_eff == s->m) return true;
    return s->is16() ? (s->A16_owned && s->ld16 >= (size_t)s->sgeom.m_eff) : (s->Apad != nullptr && s->ldpad >= (size_t)s->sgeom.m_eff);
}

int sweep_pass(thip_solver *s, int first, int np_m, bool timing)
{
    hipStream_t st = ctx().stream;
    const SweepGeom &g = s->sgeom;
    SweepArgs a;
    a.A = reinterpret_cast<const float *>(s->amat()); a.lda = s->alda(); a.m = g.m_eff; a.n = (int)s->n;
    a.inv_s = s->ainv();
    a.G = g.G; a.rows_per_member = g.rows_per_member; a.cols_per_group = g.cols_per_group;
    a.v = s->v; a.xy = s->xy; a.c = s->c; a.Su = s->Su; a.Tx = s->Tx;
    a.u = s->u; a.ku = s->comp() ? s->ku : nullptr;
    const bool b0 = s->xbuf == 0;        // s->xx is the arena's own buffer
    a.xx_in = s->xx; a.xx_out = b0 ? s->xx2 : s->xx_home;
    a.kx_in = s->comp() ? s->kx : nullptr; a.kx_out = s->comp() ? (b0 ? s->kx2 : s->kx_home) : nullptr;
    a.gP = s->gP;
    a.partH = s->sw_partH; a.mpad = g.mpad; a.gran = s->sw_gran; a.census = s->sw_census;
    a.seq = s->sw_seq++; a.tagbase = s->sw_tag; s->sw_tag += (unsigned)g.npan + 1u;
    a.first = first; a.dbg = 0;
    a.stop = &s->dst->stop; a.kappa_p = &s->dst->kappa; a.rtau_p = &s->dst->r_tau;
    a.tau_p = &s->dst->tau; a.eps_zero = s->par.eps_zero;
    a.kappa_out = nullptr; a.skappa_p = &s->dst->s_kappa; a.pm_brx = nullptr; a.np_m = 0; a.pn_count = 0;
    // the sums over n: two buffers by launch parity -- this launch writes one, its kappa update reads what the previous
    // sweep left in the other (column-sharded: what came back from the all-reduce, which no sweep writes)
    const int par = s->pn_par ^ 1;
    a.pn = s->sw_part + (size_t)par * 2 * EG; a.pn_stride = 256;
    a.pn_in = s->sw_part + (size_t)s->pn_par * 2 * EG; a.pn_in_stride = 256;
    if (timing) {
        // a REGULAR sweep's instruction stream and memory traffic (the u update, its Kahan term, five column stores per turn)
        // that changes nothing of the iterate: u / ku go to scratch n-vectors, kappa stays.  The geometries are timed like this:
        // timed as first = 1 sweeps (no u update) the plan autotune picked the 16-bit geometry that is 4 % slower in the loop
        // about every other run (two columns per panel: 1.56 ms as timed, 1.63 in the loop; four: 1.58 / 1.55)
        a.first = 0;
        a.u = s->g1; a.ku = s->comp() ? s->g2 : nullptr;
    } else if (!first) {
        // the sweep of a regular step opens with the kappa update: c.rx_x from the previous sweep's partials, b.rx_y from sw_vm_k
        const unsigned gm_ = np_m > 0 ? (unsigned)np_m : egrid(s->m);      // block partials per sum over m (the m-kernel's grid)
        a.kappa_p = &s->dst->kappa_in; a.kappa_out = &s->dst->kappa;
        a.pm_brx = s->sw_part + (size_t)(4 + 4 * s->pm_par) * EG + gm_; a.np_m = (int)gm_;
        a.pn_count = 256;
        if (s->col_shard) { a.pn_in = s->cs_buf + 2 * g.mpad; a.pn_in_stride = (int)EG; a.pn_count = (int)EG; }
    }
    s->pn_par = par;
    a.spin_max = s->spin_max > 0 ? s->spin_max : SW_SPIN_MAX;
    a.pub_agent = s->pub_agent >= 0 ? s->pub_agent : sweep_publish_default();
    a.fault = 0;
    if (!first && s->fault_kind == 2 && s->fault_after >= 0 && s->fault_after-- == 0) { a.fault = 1; s->fault_kind = 0; }
    if (!first && s->fault_kind == 7 && (s->fault_after < 0 || s->fault_after-- <= 0)) { a.fault = 1; s->fault_after = -1; }      // every sweep from then on
    if (s->spt) {
        // A^T [v x_y] -> per column: u_k[j], x_x_{k+1}[j], kappa, the sums over n -> A [u_k x_x_{k+1}] as the slices' shares
        prof_begin(st);
        THIP_RC(sptile_product(st, s->spt, true, a.v, a.xy, s->sw_partT, 0, a.stop));
        prof_end(st);
        THIP_RC(sptile_colupdate(st, s->spt, a, s->sw_partT));
        prof_begin(st);
        THIP_RC(sptile_product(st, s->spt, false, a.u, a.xx_out, s->sw_partH, 0, a.stop));
        prof_end(st);
        return 0;
    }
    prof_begin(st);
    THIP_RC(sweep_launch(st, g, a));
    prof_end(st);
    return 0;
}

void sweep_swap(thip_solver *s)
{
    const bool b0 = s->xbuf == 0;
    s->xx = b0 ? s->xx2 : s->xx_home;
    s->kx = b0 ? s->kx2 : s->kx_home;
    s->xbuf ^= 1;
}

// the x_x buffer that is NOT the iterate: x_x_{k+1} after a sweep
float *sweep_next(thip_solver *s) { return s->xbuf == 0 ? s->xx2 : s->xx_home; }

// last: the host looks at the status block after this iteration (end of a polling batch / of the run)
int one_iteration_sweep(thip_solver *s, bool last)
{
    hipStream_t st = ctx().stream;
    const int m = (int)s->m;
    const unsigned gm = egrid(s->m);
    const float ez = s->par.eps_zero;
    const bool cols = s->col_shard;
    // sums over n: [0] ||d||^2 [1] c.x_x [2] c.u [3] c.rx_x -- one partial per workgroup of the sweep; column-sharded: EG
    // slots each (the same on every rank, the unused ones stay zero) behind the two N products in the buffer that is
    // all-reduced
    // (the sweep itself writes 4 x 256 to one of two buffers by launch parity: pn_now(); sw_gsum_k moves them into the tail)
    auto pn_now = [&]() -> float * { return cols ? s->cs_buf + 2 * s->sgeom.mpad : s->sw_part + (size_t)s->pn_par * 2 * EG; };
    const int pns = cols ? (int)EG : 256;         // one slot per workgroup of the sweep (256), EG in the all-reduced buffer
    // sums over m: [0] b.v [1] b.rx_y [2] ||p||^2 [3] b.x_y, gmm block partials each -- two buffers: a merged m-kernel that
    // evaluates the previous iterate's termination test at its head reads one (the previous step's) and writes the other
    auto pm_cur = [&]() -> float * { return s->sw_part + (size_t)(4 + 4 * s->pm_par) * EG; };
    float *const ky = s->comp() ? s->ky : nullptr, *const ks = s->comp() ? s->ks : nullptr;
    float *const kv = s->comp() ? s->kv : nullptr;
    auto post = [&]() -> int {
        // (the sums over n -- ||d||^2, c.x_x, c.u, c.rx_x -- come out of the sweep itself: SweepArgs::pn)
        if (cols) {
            hipLaunchKernelGGL(sw_gsum_k, dim3(gm), dim3(BLK), 0, st, m, s->sgeom.ngroups, s->sgeom.mpad, s->sw_partH, s->cs_buf, s->dst,
                               s->sw_part + (size_t)s->pn_par * 2 * EG, s->sw_census + 9);
            THIP_RC(do_allreduce(s, s->cs_buf, s->cs_n));
        }
        return 0;
    };
    // no block cones (an LP: zero / nonneg rows only): the two m-kernels of a step are one launch
    const bool merge = s->n_soc == 0 && s->n_rot == 0 && s->psd.empty() && !s->no_merge;
    // every row in a (plain) second-order cone of at most 129 rows: the three m-launches of a step are one, a wave per cone
    const bool cone_merge = !merge && !s->no_merge && s->all_soc_short;
    const unsigned gx = merge ? grid_for(s->m, 64, EG) : grid_for(s->m, 64, 4096);      // (merged: its block partials fill gm slots)
    const unsigned gc = grid_for(s->n_soc, 1, EG);       // a workgroup per cone
    const unsigned gmm = merge ? gx : (cone_merge ? gc : gm);           // block partials per sum over m
    if (s->sw_first) {
        // from a consistent iterate (x_0, or wherever a run stopped): u is current, so the sweep leaves it alone
        THIP_RC(sweep_pass(s, 1, (int)gmm));
        THIP_RC(post());
        hipLaunchKernelGGL(sw_bv_k, dim3(gmm), dim3(BLK), 0, st, m, s->b, s->v, pm_cur(), s->dst);
        hipLaunchKernelGGL(sw_tau_k, dim3(1), dim3(BLK), 0, st, s->dst, pn_now() + 2 * pns, pns, pm_cur(), (int)gmm);
        s->sw_first = false;
        s->status_pending = false;
    }
    // the termination test of iterate k: by status_k after the sweep when the host is about to look, else by every block of
    // the NEXT step's (first) m-kernel at its head
    auto stat_args = [&](float *pmb) -> StatArgs {
        return StatArgs{ pns, pn_now(), pmb + 2 * gmm, pmb + 3 * gmm, (int)gmm, pn_now() + 2 * pns, pns, pmb, (int)gmm,
                         cols ? (const float *)(s->cs_buf + cs_flag_slot(s->sgeom.mpad)) : (const float *)nullptr,
                         s->par.eps_acc, s->par.eps_inf, ez, (long long)s->par.max_iter, s->xbuf };
    };
    const bool foldable = !s->no_fold;
    const int fold = (foldable && s->status_pending) ? 1 : 0;
    const StatArgs sa_prev = stat_args(pm_cur());          // (read by the head only when fold != 0)
    // a merged kernel writes its sums over m while other blocks still read the previous step's at their head: the other buffer
    // (the three-launch form writes them in sw_vm_k, a later launch: one buffer)
    if (foldable && (merge || cone_merge)) s->pm_par ^= 1;
    float *const pm = pm_cur();
    if (merge) {
        hipLaunchKernelGGL(sw_xm_k<true>, dim3(gx), dim3(BLK), 0, st, m, cols ? 1 : s->sgeom.ngroups, s->sgeom.mpad,
                           cols ? s->cs_buf : s->sw_partH, s->h3, s->b, s->v, s->Ty, s->Ts, s->cls, s->xy, s->xs, s->rxy, s->rxs,
                           s->dst, ky, ks, s->hP, s->Sv, kv, ez, pm, sa_prev, fold, s->step_par);
        s->step_par ^= 1;
    } else if (cone_merge) {
        hipLaunchKernelGGL(sw_cone_k, dim3(gc), dim3(BLK), 0, st, (int)s->n_soc, s->soc_beg, s->soc_end, cols ? 1 : s->sgeom.ngroups,
                           s->sgeom.mpad, cols ? s->cs_buf : s->sw_partH, s->b, s->v, s->Ty, s->Ts, s->xy, s->xs, s->rxy, s->rxs,
                           s->dst, ky, ks, s->hP, s->Sv, kv, ez, pm, sa_prev, fold, s->step_par);
        s->step_par ^= 1;
    } else {
        hipLaunchKernelGGL(sw_xm_k<false>, dim3(gx), dim3(BLK), 0, st, m, cols ? 1 : s->sgeom.ngroups, s->sgeom.mpad,
                           cols ? s->cs_buf : s->sw_partH, s->h3, s->b, s->v, s->Ty, s->Ts, s->cls, s->xy, s->xs, s->rxy, s->rxs,
                           s->dst, ky, ks, s->hP, s->Sv, kv, ez, pm, sa_prev, fold, s->step_par);
        s->step_par ^= 1;
        THIP_RC(project_blocks(s));
        hipLaunchKernelGGL(sw_vm_k, dim3(gm), dim3(BLK), 0, st, m, s->h3, s->hP, s->b, s->rxs, s->rxy, s->Sv, s->v, kv, s->xs,
                           s->xy, ez, s->dst, pm);
    }
    // (kappa_k is formed by the sweep's workgroups at entry: SweepArgs::kappa_out)
    sweep_swap(s);                                // x_x_k (formed by the previous sweep) is now the iterate
    THIP_RC(sweep_pass(s, 0, (int)gmm));
    THIP_RC(post());
    if (last || !foldable) {
        hipLaunchKernelGGL(status_k, dim3(1), dim3(BLK), 0, st, stat_args(pm), s->dst);
        s->status_pending = false;
    } else {
        s->status_pending = true;
    }
    THIP_LAUNCH_CHECK();
    return 0;
}

// The f32 passes stream a library-owned copy of A whose leading dimension is padded to a multiple of 16 floats (64 bytes)
// when m is not one already: aligned columns (measured on row shards of the 50 000-column SOCP: columns of 50 000 B,
// m = 12 500, 6.32 -> 6.79 TB/s; of 100 000 B 6.3 -> 6.7 TB/s; a 128-byte pitch gains nothing more) and a partial last
// row tile that may run the unguarded kernel body.  Made only when an f32 pass is about to run (never
// while the solve streams a 16-bit copy) and when it leaves two thirds of the free HBM untouched (a 40 GB shard of
// BASELINE configs[4] on a 288 GB part: yes); refreshed by every thip_solver_init, so a caller that rewrites mat_a in
// place between solves is seen.  THIP_LDA_PAD = the multiple in floats (default 16; 0 = never copy); the API form is
// thip_solver_set_lda_pad.
int ensure_apad(thip_solver *s, bool refresh)
{
    if (s->sparse || !s->A || s->m == 0 || s->n == 0) return 0;
    hipStream_t st = ctx().stream;
    const size_t m = s->m, n = s->n;
    size_t padto = 16;
    if (s->lda_pad >= 0) padto = (size_t)s->lda_pad;
    else if (getenv("THIP_LDA_PAD")) padto = (size_t)atoi(getenv("THIP_LDA_PAD"));
    if (padto == 0 || m % padto == 0) {
        if (s->Apad) { THIP_TRY(hipStreamSynchronize(st)); THIP_TRY(hipFree(s->Apad)); s->Apad = nullptr; s->ldpad = 0; s->tuned = s->tuned_sp = false; }
        return 0;
    }
    const size_t ld = (m + padto - 1) / padto * padto;
    if (s->Apad && s->ldpad != ld) { THIP_TRY(hipStreamSynchronize(st)); THIP_TRY(hipFree(s->Apad)); s->Apad = nullptr; s->tuned = s->tuned_sp = false; }
    bool fresh = false;
    if (!s->Apad) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        if (ld * n * sizeof(float) >= free_b / 3) return 0;
        s->ldpad = ld;
        THIP_TRY(hipMalloc((void **)&s->Apad, ld * n * sizeof(float)));
        THIP_TRY(hipMemsetAsync(s->Apad, 0, ld * n * sizeof(float), st));
        fresh = true;
        s->tuned = s->tuned_sp = false;             // the plans were timed on the other pitch
    }
    if (fresh || refresh)
        THIP_TRY(hipMemcpy2DAsync(s->Apad, ld * sizeof(float), s->A, m * sizeof(float), m * sizeof(float), n,
                                  hipMemcpyDeviceToDevice, st));
    return 0;
}

int poll(thip_solver *s, thip_status *out);

// x_x, u, (x_y x_s v), their Kahan terms, hP and the status block <-> the snapshot; restore = the other direction, into
// whichever x_x buffer is the iterate's now
int snapshot(thip_solver *s, bool restore)
{
    hipStream_t st = ctx().stream;
    const size_t pn = s->pn_len, pm = s->pm_len;
    if (!s->snap) {
        THIP_TRY(hipMalloc((void **)&s->snap, (4 * pn + 7 * pm) * sizeof(float)));
        THIP_TRY(hipMalloc((void **)&s->snap_st, sizeof(DevStatus)));
    }
    // xy xs v and ky ks kv are contiguous in the arena; hP = A x_x of the iterate is carried state too (the v update takes
    // A (x_k - 2 x_{k+1}) from it), and unlike gP no (re)start of the schedule recomputes it
    float *live[7] = { s->xx, s->u, s->xy, s->kx, s->ku, s->ky, s->hP };
    const size_t len[7] = { pn, pn, 3 * pm, pn, pn, 3 * pm, pm };
    SnapArgs a;
    float *p = s->snap;
    for (int q = 0; q < 7; ++q) {
        a.src[q] = restore ? p : live[q]; a.dst[q] = restore ? live[q] : p; a.len[q] = len[q];
        p += len[q];
    }
    a.st_src = restore ? s->snap_st : s->dst; a.st_dst = restore ? s->dst : s->snap_st;
    hipLaunchKernelGGL(snap_copy_k, dim3(256), dim3(BLK), 0, st, a);
    THIP_LAUNCH_CHECK();
    if (restore) {
        s->finalized = false;
        s->status_pending = false;
        s->sw_first = true;             // the restored iterate is a consistent one: the next sweep step starts from it
        THIP_RC(poll(s, nullptr));      // host copy of the status block (state RUNNING again)
    } else {
        s->snap_iter = s->hst->iter;
    }
    return 0;
}

// after a failed batch of a column-sharded run: clean census words, granule ring and error word for the retry
int sweep_rearm(thip_solver *s)
{
    hipStream_t st = ctx().stream;
    THIP_TRY(hipMemsetAsync(s->sw_census, 0, 64 * sizeof(unsigned), st));
    THIP_TRY(hipMemsetAsync(s->sw_gran, 0, sweep_gran_words(s->sgeom) * sizeof(unsigned long long), st));
    s->sw_seq = 0;
    THIP_RC(sweep_census_dry_run(st, s->sw_census, s->sw_seq++));
    return 0;
}

// the error words of every bounded device-side wait, read once per batch (one synchronisation): the one-pass kernel's
// (recoverable: *sw_err, and in a column-sharded run the all-reduced *peer_fault), the gates of the column-split pipeline and
// the one-shot all-reduce (a peer rank stalled: the ranks' states have diverged -- THIP_E_TIMEOUT)
int batch_faults(thip_solver *s, bool sweep, unsigned *sw_err, unsigned *peer_fault)
{
    hipStream_t st = ctx().stream;
    unsigned *h = s->hflags;
    h[0] = h[1] = h[2] = h[3] = 0u;
    const unsigned *os_err = s->allreduce != nullptr && s->allreduce == oneshot_hook() ? oneshot_error_word() : nullptr;
    if (!sweep && !s->use_gates && !os_err) return 0;
    if (sweep) THIP_TRY(hipMemcpyAsync(h + 0, s->sw_census + 9, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    if (sweep && s->col_shard) THIP_TRY(hipMemcpyAsync(h + 3, &s->dst->fault, sizeof(int), hipMemcpyDeviceToHost, st));
    if (s->use_gates && s->gflags) THIP_TRY(hipMemcpyAsync(h + 1, s->gflags + 8, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    if (os_err) THIP_TRY(hipMemcpyAsync(h + 2, os_err, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    THIP_TRY(hipStreamSynchronize(st));
    if (h[1] != 0u)
        return fail(THIP_E_TIMEOUT, "a hand-off of the column-split pipeline waited > 2 s for a collective (a peer rank stalled): the ranks have diverged", __FILE__, __LINE__);
    if (h[2] != 0u)
        return fail(THIP_E_TIMEOUT, "the one-shot all-reduce waited > 4 s for a peer: the sums of this batch are partial", __FILE__, __LINE__);
    *sw_err = h[0]; *peer_fault = h[3];
    return 0;
}

int poll(thip_solver *s, thip_status *out)
{
    hipStream_t st = ctx().stream;
    THIP_TRY(hipMemcpyAsync(s->hst, s->dst, sizeof(DevStatus), hipMemcpyDeviceToHost, st));
    THIP_TRY(hipStreamSynchronize(st));
    if (s->hst->state != THIP_ST_RUNNING && s->schedule == THIP_SCHED_SWEEP) {
        // the device stopped at an iterate of its own choosing: the host kept swapping the two x_x buffers for the
        // launches that then returned at entry -- point s->xx at the buffer the termination test recorded
        if (s->sweep_state == 1 && !s->sw_first && s->hst->xbuf != s->xbuf) sweep_swap(s);
        s->sw_first = true;
    }
    if (s->hst->state != THIP_ST_RUNNING && !s->finalized) {
        // the device has stopped by itself: apply the final 1/tau scaling once (solver.rs:397-400)
        hipLaunchKernelGGL(finalize_k, dim3(egrid(s->n > s->m ? s->n : s->m)), dim3(BLK), 0, st, (int)s->n, (int)s->m,
                           s->xx, s->xy, s->dst);
        THIP_LAUNCH_CHECK();
        s->finalized = true;
    }
    if (out) {
        out->state = s->hst->state; out->iter = s->hst->iter; out->kind = s->hst->kind;
        out->cri[0] = s->hst->cri[0]; out->cri[1] = s->hst->cri[1]; out->cri[2] = s->hst->cri[2];
        out->tau = s->hst->tau; out->kappa = s->hst->kappa;
        out->norm_b = s->hst->norm_b; out->norm_c = s->hst->norm_c;
    }
    return 0;
}

}  // namespace

extern "C" {

static int solver_create_impl(const thip_problem *prob, const thip_param *par, int schedule, thip_solver **out)
{
    THIP_NEED_INIT();
    if (!prob || !par || !out) return fail(THIP_E_INVALID, "null argument", __FILE__, __LINE__);
    if (schedule < 0 || schedule > THIP_SCHED_SWEEP) return fail(THIP_E_INVALID, "bad schedule", __FILE__, __LINE__);
    int64_t tot = 0;
    for (size_t i = 0; i < prob->n_seg; ++i) {
        if (prob->host_seg_len[i] < 0 || prob->host_seg_type[i] < 0 || prob->host_seg_type[i] > THIP_CONE_PSD)
            return fail(THIP_E_INVALID, "bad cone segment", __FILE__, __LINE__);
        tot += prob->host_seg_len[i];
    }
    if ((size_t)tot != prob->m) return fail(THIP_E_INVALID, "cone segments do not cover m rows", __FILE__, __LINE__);

    thip_solver *s = new thip_solver();
    *out = s;                      // the caller releases it if anything below fails
    s->n = prob->n; s->m = prob->m;
    s->A = prob->mat_a; s->b = prob->vec_b; s->c = prob->vec_c; s->b_rowabs = prob->vec_b_rowabs;
    s->par = *par; s->schedule = schedule;
    s->seg_type.assign(prob->host_seg_type, prob->host_seg_type + prob->n_seg);
    s->seg_len.assign(prob->host_seg_len, prob->host_seg_len + prob->n_seg);
    hipStream_t st = ctx().stream;
    const size_t n = s->n, m = s->m;

    // ---- cone tables ----
    std::vector<unsigned char> cls(m ? m : 1, 2);
    std::vector<int64_t> sb, se, rb, re, gb, ge;
    int64_t off = 0;
    size_t psd_kmax = 0;
    for (size_t i = 0; i < s->seg_type.size(); ++i) {
        const int64_t l = s->seg_len[i];
        switch (s->seg_type[i]) {
        case THIP_CONE_ZERO: for (int64_t r = 0; r < l; ++r) cls[off + r] = 0; break;
        case THIP_CONE_RPOS: for (int64_t r = 0; r < l; ++r) cls[off + r] = 1; break;
        case THIP_CONE_SOC:
            sb.push_back(off); se.push_back(off + l); gb.push_back(off); ge.push_back(off + l);
            if ((size_t)l > s->soc_max) s->soc_max = (size_t)l;
            break;
        case THIP_CONE_ROTSOC:
            rb.push_back(off); re.push_back(off + l); gb.push_back(off); ge.push_back(off + l);
            if ((size_t)l > s->rot_max) s->rot_max = (size_t)l;
            break;
        case THIP_CONE_PSD: {
            const size_t k = (size_t)((std::sqrt((double)(8 * l + 1)) - 1.0) / 2.0 + 0.5);
            if ((int64_t)(k * (k + 1) / 2) != l) { return fail(THIP_E_INVALID, "PSD segment is not triangular", __FILE__, __LINE__); }
            s->psd.push_back({off, l});
            for (int64_t r = 0; r < l; ++r) cls[off + r] = 3;
            gb.push_back(off); ge.push_back(off + l);
            if (k > psd_kmax) psd_kmax = k;
            break; }
        }
        if ((size_t)l > s->grp_max && s->seg_type[i] >= THIP_CONE_SOC) s->grp_max = (size_t)l;
        off += l;
    }
    s->n_soc = sb.size(); s->n_rot = rb.size(); s->n_grp = gb.size();
    {
        int64_t soc_rows = 0;
        for (size_t i = 0; i < sb.size(); ++i) soc_rows += se[i] - sb[i];
        s->all_soc_short = m > 0 && !sb.empty() && rb.empty() && (size_t)soc_rows == m && s->soc_max <= 129;
    }
    auto up64 = [&](const std::vector<int64_t> &h, int64_t **d) -> int {
        *d = nullptr;
        if (h.empty()) return 0;
        THIP_TRY(hipMalloc((void **)d, h.size() * sizeof(int64_t)));
        THIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        return 0;
    };
    THIP_RC(up64(sb, &s->soc_beg)); THIP_RC(up64(se, &s->soc_end));
    THIP_RC(up64(rb, &s->rot_beg)); THIP_RC(up64(re, &s->rot_end));
    THIP_RC(up64(gb, &s->grp_beg)); THIP_RC(up64(ge, &s->grp_end));
    THIP_TRY(hipMalloc((void **)&s->cls, cls.size()));
    THIP_TRY(hipMemcpy(s->cls, cls.data(), cls.size(), hipMemcpyHostToDevice));
    {
        std::map<size_t, std::vector<int64_t>> by_order;
        for (auto &pr : s->psd) {
            const size_t k = (size_t)((std::sqrt((double)(8 * pr.second + 1)) - 1.0) / 2.0 + 0.5);
            if (k <= psd_small_max()) by_order[k].push_back(pr.first);
        }
        for (auto &kv : by_order) {
            thip_solver::PsdGroup g{ kv.first, (int)kv.second.size(), nullptr };
            THIP_RC(up64(kv.second, &g.dev_offs));
            s->psd_groups.push_back(g);
        }
    }
    if (psd_kmax) {
        s->psd_worklen = 2 * thip_map_eig_worklen(psd_kmax);
        THIP_TRY(hipMalloc((void **)&s->psd_work, s->psd_worklen * sizeof(float)));
    }

    // ---- vectors ----
    const size_t pn = pad64(n + TAIL), pm = pad64(m + 1);
    if (par->state_arith != THIP_STATE_COMPENSATED && par->state_arith != THIP_STATE_PLAIN)
        return fail(THIP_E_INVALID, "bad thip_param.state_arith", __FILE__, __LINE__);
    const size_t total = 10 * pn /* xx u Tx Su rxx g1 g2 g3 gP xx2 */ + 13 * pm + 64 + 3 * pn + 3 * pm /* Kahan terms */;
    THIP_TRY(hipMalloc((void **)&s->arena, total * sizeof(float)));
    THIP_TRY(hipMemsetAsync(s->arena, 0, total * sizeof(float), st));
    s->arena_n = total;
    float *p = s->arena;
    auto take = [&](size_t k) { float *r = p; p += k; return r; };
    s->xx = take(pn); s->u = take(pn); s->Tx = take(pn); s->Su = take(pn); s->rxx = take(pn);
    s->g1 = take(pn); s->g2 = take(pn); s->g3 = take(pn); s->gP = take(pn); s->xx2 = take(pn);
    s->xy = take(pm); s->xs = take(pm); s->v = take(pm); s->Ty = take(pm); s->Ts = take(pm); s->Sv = take(pm);
    s->rxy = take(pm); s->rxs = take(pm); s->h1 = take(pm); s->h2 = take(pm); s->h3 = take(pm); s->hP = take(pm);
    (void)take(pm);
    s->dotc = take(64);
    s->kx = take(pn); s->ku = take(pn); s->ky = take(pm); s->ks = take(pm); s->kv = take(pm); s->kx2 = take(pn);
    s->kahan_n = 3 * pn + 3 * pm;
    s->xx_home = s->xx; s->kx_home = s->kx;
    s->pn_len = pn; s->pm_len = pm;
    THIP_TRY(hipHostMalloc((void **)&s->hflags, 8 * sizeof(unsigned), hipHostMallocDefault));

    THIP_TRY(hipMalloc((void **)&s->part, (4 * PG + 4 * EG) * sizeof(float)));
    // the dense GEMV partial-sum scratch (~ m n / 256 floats) is allocated by thip_solver_init, and only for a dense A
    // (thip_solver_set_csr comes between create and init: a sparse 1e6 x 1e6 operator must not pay 15 GB for it)
    THIP_TRY(hipMalloc((void **)&s->dst, sizeof(DevStatus)));
    THIP_TRY(hipMemsetAsync(s->dst, 0, sizeof(DevStatus), st));
    THIP_TRY(hipHostMalloc((void **)&s->hst, sizeof(DevStatus), hipHostMallocDefault));
    return 0;
}

int thip_solver_create(const thip_problem *prob, const thip_param *par, int schedule, thip_solver **out)
{
    if (!out) return fail(THIP_E_INVALID, "null argument", __FILE__, __LINE__);
    *out = nullptr;
    thip_solver *s = nullptr;
    const int rc = solver_create_impl(prob, par, schedule, &s);
    if (rc != 0) {
        if (s) thip_solver_destroy(s);      // partial device allocations (e.g. out of memory half-way)
        return rc;
    }
    *out = s;
    return 0;
}

int thip_solver_set_csr(thip_solver *s, size_t nnz, const int64_t *dev_rowptr, const int32_t *dev_colidx,
                        const float *dev_vals, const int64_t *dev_t_rowptr, const int32_t *dev_t_colidx,
                        const float *dev_t_vals)
{
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    if (!dev_rowptr || !dev_t_rowptr) return fail(THIP_E_INVALID, "null CSR arrays", __FILE__, __LINE__);
    s->sparse = true; s->nnz = nnz;
    s->rp = dev_rowptr; s->ci = dev_colidx; s->sv = dev_vals;
    s->trp = dev_t_rowptr; s->tci = dev_t_colidx; s->tsv = dev_t_vals;
    return 0;
}

int thip_solver_set_sptile(thip_solver *s, thip_sptile *mat)
{
    if (!s || !mat) return fail(THIP_E_INVALID, "null argument", __FILE__, __LINE__);
    if (s->inited) return fail(THIP_E_INVALID, "thip_solver_set_sptile comes before thip_solver_init", __FILE__, __LINE__);
    size_t m = 0, n = 0, nnz = 0;
    sptile_dims(mat, &m, &n, &nnz);
    if (m != s->m || n != s->n) return fail(THIP_E_INVALID, "the sparse operator's shape is not the problem's m x n", __FILE__, __LINE__);
    s->sparse = true; s->nnz = nnz; s->spt = mat;
    return 0;
}

int thip_solver_set_allreduce(thip_solver *s, thip_allreduce_fn fn, void *c)
{
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    s->allreduce = fn; s->allreduce_ctx = c;
    return 0;
}

int thip_solver_set_overlap(thip_solver *s, int on)
{
    THIP_NEED_INIT();
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    if (on < 0 || on > 3) return fail(THIP_E_INVALID, "overlap mode is 0 .. 3", __FILE__, __LINE__);
    if (on && !s->side) {
        THIP_TRY(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
        THIP_TRY(hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming));
        THIP_TRY(hipEventCreateWithFlags(&s->ev_out, hipEventDisableTiming));
        for (int k = 0; k < 4; ++k) {
            THIP_TRY(hipEventCreateWithFlags(&s->sev_in[k], hipEventDisableTiming));
            THIP_TRY(hipEventCreateWithFlags(&s->sev_out[k], hipEventDisableTiming));
        }
    }
    if (s->side && ctx().inited) THIP_TRY(hipStreamSynchronize(s->side));
    if (on == 2 && !s->gflags) {
        // THIP_PIPE_GATES=0: cross-stream events instead of the device-flag hand-offs
        static const int gates_on = getenv("THIP_PIPE_GATES") ? atoi(getenv("THIP_PIPE_GATES")) : 1;
        THIP_TRY(hipExtMallocWithFlags((void **)&s->gflags, 16 * sizeof(unsigned), hipDeviceMallocUncached));
        THIP_TRY(hipMemset(s->gflags, 0, 16 * sizeof(unsigned)));
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx().device) != hipSuccess || khz <= 0) khz = 100000;
        s->gate_ticks = (long long)khz * 2000ll;
        s->use_gates = gates_on != 0;
    }
    s->overlap = on;
    return 0;
}

int thip_solver_init(thip_solver *s)
{
    THIP_NEED_INIT();
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    hipStream_t st = ctx().stream;
    const size_t n = s->n, m = s->m;
    const unsigned g = egrid(n > m ? n : m);

    // per-solve host state: a solver may be initialised again after a solve that terminated
    s->finalized = false;
    s->carried_stale = false;
    s->hst->state = THIP_ST_RUNNING;
    THIP_RC(ensure_gemv_scratch(s));
    if (!s->is16()) THIP_RC(ensure_apad(s, true));      // a fresh solve re-reads the caller's A (it may have changed in place)
    s->xx = s->xx_home; s->kx = s->kx_home; s->xbuf = 0;
    s->sw_first = true; s->sweep_state = 0; s->pn_par = 0; s->status_pending = false; s->step_par = 0; s->pm_par = 0;
    s->sweep_faults = 0; s->sweep_fault_word = 0; s->sweep_fault_iter = -1; s->snap_iter = -1;
    // init_vecs (solver.rs:483-494): x = 0, y = 0, tau = 1
    THIP_TRY(hipMemsetAsync(s->arena, 0, s->arena_n * sizeof(float), st));
    hipLaunchKernelGGL(init_status_k, dim3(1), dim3(1), 0, st, s->dst, 0.0f);
    THIP_RC(sweep_prepare(s));        // (its plan autotune runs idempotent sweeps: after the stop flag has been cleared)
    if (s->sw_part) THIP_TRY(hipMemsetAsync(s->sw_part, 0, 12 * EG * sizeof(float), st));
    THIP_TRY(hipMemsetAsync(s->arena, 0, s->arena_n * sizeof(float), st));

    // calc_norms (solver.rs:460-481) + scalar parts of abssum (solver.rs:171-172)
    hipLaunchKernelGGL(init_sums_k, dim3(g), dim3(BLK), 0, st, (int)m, s->b, (int)n, s->c, s->part);
    float *sums = s->g1 + n;        // [0] sum b^2, [1] sum |b|  (sharded -> all-reduce)
    float *loc = s->dotc + 8;       // [0] sum c^2, [1] sum |c|
    hipLaunchKernelGGL(sum_partials_k, dim3(1), dim3(BLK), 0, st, 2, (int)g, s->part, sums, (const int *)nullptr);
    hipLaunchKernelGGL(sum_partials_k, dim3(1), dim3(BLK), 0, st, 2, (int)g, s->part + 2 * g, loc, (const int *)nullptr);

    // |A| column sums (sharded partial -> all-reduce with the two scalars in the tail) and row sums
    float *colabs = s->g1, *rowabs = s->h1;
    if (n && m && s->spt) {
        THIP_RC(sptile_product(st, s->spt, false, s->c, nullptr, s->sw_partH, 1, nullptr));
        THIP_RC(finalize_partials(st, m, s->sw_partH, sptile_slices(s->spt, false), 2 * sptile_pad(s->spt, false), 1.0f, 0.0f, rowabs, nullptr));
        THIP_RC(sptile_product(st, s->spt, true, s->c, nullptr, s->sw_partT, 1, nullptr));
        THIP_RC(finalize_partials(st, n, s->sw_partT, sptile_slices(s->spt, true), 2 * sptile_pad(s->spt, true), 1.0f, 0.0f, colabs, nullptr));
    } else if (n && m && s->sparse) {
        THIP_RC(thip_spmv_csr(m, n, s->nnz, s->rp, s->ci, s->sv, 1.0f, s->sv, 0.0f, rowabs, 1));
        THIP_RC(thip_spmv_csr(n, m, s->nnz, s->trp, s->tci, s->tsv, 1.0f, s->tsv, 0.0f, colabs, 1));
    } else if (n && m) {
        // solver-owned scratch (several solvers may share the context, e.g. one per thread)
        GemvPartials gp;
        THIP_RC(dual_gemv_partials(st, m, n, s->amat(), s->alda(), nullptr, nullptr, true, true, true, s->gemv_scr,
                                   s->gemv_scr_n, &gp, nullptr, nullptr, s->a_kind, s->ainv(), s->apadz()));
        THIP_RC(finalize_partials(st, m, gp.partN, gp.nN, gp.strideN, 1.0f, 0.0f, rowabs, nullptr));
        THIP_RC(finalize_partials(st, n, gp.partT, gp.nT, gp.strideT, 1.0f, 0.0f, colabs, nullptr));
    }
    if (s->col_shard) {
        // this rank holds a block of columns: b and the m-vectors are replicated, c is its block.  What the ranks have to
        // add up is the |A| row sums and sum c^2, sum |c| (the b sums and the column sums are complete as they are)
        if (!sweep_active(s))
            return fail(THIP_E_INVALID, "a column-sharded run needs THIP_SCHED_SWEEP and a shape its kernel takes", __FILE__, __LINE__);
        if (s->cs_n < m + 2) return fail(THIP_E_INVALID, "column-shard buffer too small", __FILE__, __LINE__);
        THIP_TRY(hipMemcpyAsync(s->cs_buf, rowabs, m * sizeof(float), hipMemcpyDeviceToDevice, st));
        THIP_TRY(hipMemcpyAsync(s->cs_buf + m, loc, 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
        THIP_RC(do_allreduce(s, s->cs_buf, s->cs_n));
        THIP_TRY(hipMemcpyAsync(rowabs, s->cs_buf, m * sizeof(float), hipMemcpyDeviceToDevice, st));
        THIP_TRY(hipMemcpyAsync(loc, s->cs_buf + m, 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
        THIP_TRY(hipMemsetAsync(s->cs_buf, 0, s->cs_n * sizeof(float), st));
    } else {
        THIP_RC(do_allreduce(s, s->g1, n + 2));
    }
    hipLaunchKernelGGL(init_scalars_k, dim3(1), dim3(1), 0, st, sums, loc, s->par.eps_zero, s->dst);
    hipLaunchKernelGGL(precond_k, dim3(g), dim3(BLK), 0, st, (int)n, (int)m, colabs, rowabs, s->c, s->b, s->b_rowabs,
                       s->par.eps_zero, s->Tx, s->Ty, s->Ts, s->Su, s->Sv);
    // product_group (solver.rs:521-523): per block cone, dp_tau's x_y and x_s parts <- their minimum
    THIP_RC(group_min_batched(st, s->Ty, s->grp_beg, s->grp_end, s->n_grp, s->grp_max));
    THIP_RC(group_min_batched(st, s->Ts, s->grp_beg, s->grp_end, s->n_grp, s->grp_max));
    // scratch vectors used above must read as zero again for the carried products (A x_0 = 0)
    THIP_TRY(hipMemsetAsync(s->g1, 0, (n + TAIL) * sizeof(float), st));
    THIP_TRY(hipMemsetAsync(s->h1, 0, (m ? m : 1) * sizeof(float), st));
    THIP_LAUNCH_CHECK();
    s->split_plan = false;           // the one-launch form here; the column-split form is tuned by the first run that uses it
    if (!sweep_active(s)) THIP_RC(autotune_gemv(s));      // (a run that falls back to the dual GEMV tunes it then: prepare_split)
    s->inited = true;
    return 0;
}

// A column shard whose (re-)plan of the one-pass kernel failed on THIS rank (thip_solver_set_a_storage / _set_sweep_min_bytes
// re-plan inside run(); the 16-bit plan rejecting the shape, a timing sweep raising the error word, a placement census that
// changed): there is no 2-pass form over a column block, and the peers are already sweeping -- their next batch holds one
// all-reduce per iteration (+ one when it starts from a consistent iterate).  This rank takes part in exactly those
// collectives with its fault flag raised (the tail slot every rank's termination test reads), for the three attempts the
// peers make from their snapshot, so that every rank of the run returns THIP_E_TIMEOUT at the same batch instead of one
// rank iterating a row-sharded schedule on a column block while the others wait in a collective of another size.
static int col_shard_abort(thip_solver *s, int64_t max_steps, int64_t poll_every)
{
    hipStream_t st = ctx().stream;
    const size_t mpad = pad64(s->m);                  // SweepGeom::mpad of every plan of this m (rows round to 4 or 8, mpad to 64)
    const size_t need = cs_floats(mpad);
    if (s->cs_n != need) {
        if (s->cs_buf) { THIP_TRY(hipStreamSynchronize(st)); THIP_TRY(hipFree(s->cs_buf)); s->cs_buf = nullptr; }
        THIP_TRY(hipMalloc((void **)&s->cs_buf, need * sizeof(float)));
        s->cs_n = need;
    }
    int64_t batch = poll_every;
    if (max_steps >= 0 && batch > max_steps) batch = max_steps;
    const float one = 1.0f;
    bool first = s->sw_first;
    for (int attempt = 0; attempt < COL_SHARD_ATTEMPTS; ++attempt) {
        const int64_t calls = batch + (first ? 1 : 0);
        for (int64_t k = 0; k < calls; ++k) {
            THIP_TRY(hipMemsetAsync(s->cs_buf, 0, need * sizeof(float), st));
            THIP_TRY(hipMemcpyAsync(s->cs_buf + cs_flag_slot(mpad), &one, sizeof(float), hipMemcpyHostToDevice, st));
            THIP_RC(do_allreduce(s, s->cs_buf, s->cs_n));
        }
        THIP_TRY(hipStreamSynchronize(st));
        first = true;                                 // the peers restore their snapshot: a consistent iterate
    }
    s->sweep_faults += 1;
    s->sweep_fault_word = 5u;                         // 5: this rank could not plan the kernel at all
    return fail(THIP_E_TIMEOUT, "the one-pass kernel could not be planned on this rank of a column-sharded run: every rank stops", __FILE__, __LINE__);
}

int thip_solver_run(thip_solver *s, int64_t max_steps, int64_t poll_every, thip_status *host_status)
{
    THIP_NEED_INIT();
    if (!s || !s->inited) return fail(THIP_E_INVALID, "solver not initialised", __FILE__, __LINE__);
    if (poll_every <= 0) poll_every = 16;
    int64_t done = 0;
    THIP_RC(poll(s, host_status));
    if (s->carried_stale && s->hst->state == THIP_ST_RUNNING) THIP_RC(rebuild_carried(s));
    THIP_RC(sweep_prepare(s));
    bool sweep = sweep_active(s);
    if (s->col_shard && !sweep) {
        // a column shard that was never going to sweep is a caller's mistake, not a fault to be raised through collectives
        // (without a hook do_allreduce is a no-op: the "abort" would spin through dummy calls and report a time-out)
        if (s->allreduce == nullptr) return fail(THIP_E_INVALID, "a column-sharded solver needs an all-reduce (thip_solver_set_allreduce / _use_rccl / _use_oneshot)", __FILE__, __LINE__);
        if (s->schedule != THIP_SCHED_SWEEP) return fail(THIP_E_INVALID, "a column-sharded solver runs THIP_SCHED_SWEEP only", __FILE__, __LINE__);
        if (s->sparse) return fail(THIP_E_INVALID, "column shards are for a dense A", __FILE__, __LINE__);
        if (s->hst->state != THIP_ST_RUNNING || max_steps == 0) return 0;      // nothing would run on any rank
        return col_shard_abort(s, max_steps, poll_every);
    }
    THIP_RC(prepare_split(s));
    bool split = split_active(s);
    if (!sweep) s->sw_first = true;         // whatever runs instead leaves a consistent iterate and gP / hP of it
    if (sweep && s->hst->state == THIP_ST_RUNNING) THIP_RC(snapshot(s, false));      // the iterate this run starts from
    int retries = 0;
    while (s->hst->state == THIP_ST_RUNNING && (max_steps < 0 || done < max_steps)) {
        int64_t batch = poll_every;
        if (max_steps >= 0 && done + batch > max_steps) batch = max_steps - done;
        for (int64_t k = 0; k < batch; ++k) {
            prof_tick();
            THIP_RC(sweep ? one_iteration_sweep(s, k + 1 == batch) : (split ? one_iteration_split(s) : one_iteration(s)));
        }
        if (s->tail_pending) THIP_RC(split_tail(s));       // drain the pipeline before the host looks
        // every bounded device-side wait of the batch: did one run out?
        unsigned sw_err = 0, peer_fault = 0;
        THIP_RC(batch_faults(s, sweep, &sw_err, &peer_fault));
        // column-sharded: the verdict is the all-reduced one, so that every rank takes the same branch at the same batch
        if (sweep && (s->col_shard ? peer_fault != 0u : sw_err != 0u)) {
            s->sweep_faults += 1;
            s->sweep_fault_word = sw_err != 0u ? sw_err : 4u;       // 4: a peer rank's kernel
            s->sweep_fault_iter = s->snap_iter;
            THIP_RC(snapshot(s, true));                            // back to the last batch that completed
            if (s->col_shard) {
                // a column shard has no 2-pass form to fall back to: every rank restores and retries together (a transient --
                // another process on the GPU for a moment -- passes; a placement that stays wrong fails cleanly everywhere)
                if (++retries > COL_SHARD_ATTEMPTS - 1)
                    return fail(THIP_E_TIMEOUT, "the one-pass kernel gave up on some rank of a column-sharded run (3 attempts from the same iterate)", __FILE__, __LINE__);
                THIP_RC(sweep_rearm(s));
            } else if (++retries <= 1 && s->sweep_faults < 3) {
                // (at most three faults per solve: a disturbance that keeps coming back costs a ~2 s spin-out and a re-run batch
                // each time -- from the third on the rest of the solve runs the 2-pass schedule)
                // one GPU: a transient (another process on the device for a moment) should not halve the rate of the 100 000
                // iterations that may follow -- clean census, ring and error word and give the one-pass schedule ONE more batch
                // from the restored iterate before giving it up
                THIP_RC(sweep_rearm(s));
            } else {
                s->sweep_state = -1;                               // for the rest of this solve: the 2-pass schedule
                THIP_RC(prepare_split(s));                         // (tunes the GEMV plan if that has not happened yet)
                split = split_active(s);
                THIP_RC(rebuild_carried(s));
                sweep = false;
            }
            continue;                                              // the batch is run again (done has not moved)
        }
        retries = 0;
        done += batch;
        THIP_RC(poll(s, host_status));
        if (sweep && s->hst->state == THIP_ST_RUNNING) THIP_RC(snapshot(s, false));
    }
    return 0;
}

int thip_solver_status(thip_solver *s, thip_status *host_status)
{
    THIP_NEED_INIT();
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    return poll(s, host_status);
}

int thip_solver_solution(thip_solver *s, float *host_x, float *host_y)
{
    THIP_NEED_INIT();
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    THIP_RC(poll(s, nullptr));           // a terminated iterate gets its final scaling before it is read
    if (host_x) THIP_RC(thip_d2h(host_x, s->xx, s->n));
    if (host_y) THIP_RC(thip_d2h(host_y, s->xy, s->m));
    return 0;
}

int thip_solver_iterate(thip_solver *s, float *host_x, float *host_y)
{
    THIP_NEED_INIT();
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    const size_t n = s->n, m = s->m;
    THIP_RC(poll(s, nullptr));
    if (host_x) {
        THIP_RC(thip_d2h(host_x, s->xx, n));
        THIP_RC(thip_d2h(host_x + n, s->xy, m));
        THIP_RC(thip_d2h(host_x + n + m, s->xs, m));
        host_x[n + m + m] = s->hst->tau;
    }
    if (host_y) {
        THIP_RC(thip_d2h(host_y, s->u, n));
        THIP_RC(thip_d2h(host_y + n, s->v, m));
        host_y[n + m] = s->hst->kappa;
    }
    return 0;
}

int thip_solver_precond(thip_solver *s, float *host_dp_tau, float *host_dp_sigma)
{
    THIP_NEED_INIT();
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    const size_t n = s->n, m = s->m;
    THIP_RC(poll(s, nullptr));
    if (host_dp_tau) {
        THIP_RC(thip_d2h(host_dp_tau, s->Tx, n));
        THIP_RC(thip_d2h(host_dp_tau + n, s->Ty, m));
        THIP_RC(thip_d2h(host_dp_tau + n + m, s->Ts, m));
        host_dp_tau[n + m + m] = s->hst->t_tau;
    }
    if (host_dp_sigma) {
        THIP_RC(thip_d2h(host_dp_sigma, s->Su, n));
        THIP_RC(thip_d2h(host_dp_sigma + n, s->Sv, m));
        host_dp_sigma[n + m] = s->hst->s_kappa;
    }
    return 0;
}

static int set_a16_external(thip_solver *s, const uint16_t *mat16, size_t ld16, int kind, const float *inv_scale)
{
    THIP_NEED_INIT();
    if (!s || !mat16) return fail(THIP_E_INVALID, "null argument", __FILE__, __LINE__);
    if (s->inited) return fail(THIP_E_INVALID, "a caller-built 16-bit matrix must precede thip_solver_init", __FILE__, __LINE__);
    if (s->sparse) return fail(THIP_E_INVALID, "storage kinds apply to a dense A", __FILE__, __LINE__);
    if (ld16 < s->m) return fail(THIP_E_INVALID, "ld16 < m", __FILE__, __LINE__);
    if (kind == THIP_A_F16 && !inv_scale) return fail(THIP_E_INVALID, "f16 storage needs the per-column scales", __FILE__, __LINE__);
    if (s->A16_owned) { THIP_TRY(hipFree(s->A16)); s->A16_owned = false; }
    if (s->inv_s_owned) { THIP_TRY(hipFree(s->inv_s)); s->inv_s_owned = false; }
    s->A16 = const_cast<uint16_t *>(mat16);      // caller-owned, only ever read
    s->inv_s = const_cast<float *>(inv_scale);
    s->ld16 = ld16;
    s->a_kind = s->a16_kind = kind;
    return 0;
}

int thip_solver_set_a_bf16(thip_solver *s, const uint16_t *mat16, size_t ld16)
{
    return set_a16_external(s, mat16, ld16, THIP_A_BF16, nullptr);
}

int thip_solver_set_a_f16(thip_solver *s, const uint16_t *mat16, size_t ld16, const float *inv_scale)
{
    return set_a16_external(s, mat16, ld16, THIP_A_F16, inv_scale);
}

int thip_solver_set_a_storage(thip_solver *s, int a_kind)
{
    THIP_NEED_INIT();
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    if (a_kind != THIP_A_F32 && a_kind != THIP_A_BF16 && a_kind != THIP_A_F16)
        return fail(THIP_E_INVALID, "bad storage kind", __FILE__, __LINE__);
    if (s->sparse) return fail(THIP_E_INVALID, "storage kinds apply to a dense A", __FILE__, __LINE__);
    if (a_kind == THIP_A_F32 && !s->A && s->m && s->n) return fail(THIP_E_INVALID, "no f32 matrix was given", __FILE__, __LINE__);
    if (a_kind != THIP_A_F32 && s->a16_kind != a_kind && s->m && s->n) {
        // (re)build the library-owned 16-bit copy in the requested format
        if (!s->A) return fail(THIP_E_INVALID, "no f32 matrix to convert", __FILE__, __LINE__);
        if (s->A16 && !s->A16_owned) return fail(THIP_E_INVALID, "the 16-bit matrix is caller-built", __FILE__, __LINE__);
        hipStream_t st = ctx().stream;
        if (!s->A16) {
            s->ld16 = (s->m + 7) / 8 * 8;
            THIP_TRY(hipMalloc((void **)&s->A16, s->ld16 * s->n * sizeof(uint16_t)));
            s->A16_owned = true;
        }
        if (a_kind == THIP_A_F16) {
            if (!s->inv_s) { THIP_TRY(hipMalloc((void **)&s->inv_s, s->n * sizeof(float))); s->inv_s_owned = true; }
            THIP_RC(to_f16(st, s->m, s->n, s->A, s->A16, s->ld16, s->inv_s));
        } else {
            THIP_RC(to_bf16(st, s->m, s->n, s->A, s->A16, s->ld16));
        }
        s->a16_kind = a_kind;
        s->tuned16 = s->tuned16_sp = false;
    }
    const bool changed = s->a_kind != a_kind;
    s->a_kind = a_kind;
    if (changed) s->sweep_state = 0;           // the one-pass schedule is planned per stored form (another kernel instance)
    if (s->inited && a_kind == THIP_A_F32) THIP_RC(ensure_apad(s, false));      // first f32 pass of this solve
    if (s->inited) {
        THIP_RC(autotune_gemv(s));      // a switch inside a running solve: tune the other kernel once
        if (changed) s->carried_stale = true;      // rebuilt by the next thip_solver_run (after thip_solver_resume)
    }
    return 0;
}

int thip_solver_set_param(thip_solver *s, const thip_param *par)
{
    if (!s || !par) return fail(THIP_E_INVALID, "null argument", __FILE__, __LINE__);
    if (par->state_arith != THIP_STATE_COMPENSATED && par->state_arith != THIP_STATE_PLAIN)
        return fail(THIP_E_INVALID, "bad thip_param.state_arith", __FILE__, __LINE__);
    // compensation switched on inside a solve starts from clean Kahan terms
    if (s->par.state_arith != par->state_arith && par->state_arith == THIP_STATE_COMPENSATED && s->kx && ctx().inited)
        THIP_TRY(hipMemsetAsync(s->kx_home, 0, s->kahan_n * sizeof(float), ctx().stream));
    s->par = *par;
    return 0;
}

int thip_solver_resume(thip_solver *s)
{
    THIP_NEED_INIT();
    if (!s || !s->inited) return fail(THIP_E_INVALID, "solver not initialised", __FILE__, __LINE__);
    THIP_RC(poll(s, nullptr));
    const int state = s->hst->state;
    if (state == THIP_ST_RUNNING) return 0;
    if (!(s->hst->kind == 0 && (state == THIP_ST_OK || state == THIP_ST_EXCESS_ITER)))
        return fail(THIP_E_INVALID, "only a solve that ended Converged / ExcessIter (tau > eps_zero) can be resumed",
                    __FILE__, __LINE__);
    hipStream_t st = ctx().stream;
    const unsigned g = egrid(s->n > s->m ? s->n : s->m);
    if (s->finalized) hipLaunchKernelGGL(resume_k, dim3(g), dim3(BLK), 0, st, (int)s->n, (int)s->m, s->xx, s->xy, s->dst);
    s->finalized = false;
    hipLaunchKernelGGL(resume_flags_k, dim3(1), dim3(1), 0, st, s->dst);
    THIP_LAUNCH_CHECK();
    return 0;
}

int thip_solver_passes(const thip_solver *s, int *host_passes, size_t *host_bytes_per_pass)
{
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    if (s->spt) {
        // every product is one pass over the stored entries (8 bytes each): two per stage
        if (host_passes) *host_passes = (s->schedule == THIP_SCHED_REFERENCE || s->schedule == THIP_SCHED_FUSED) ? 6 : (sweep_active(s) ? 2 : 4);
        if (host_bytes_per_pass) *host_bytes_per_pass = sptile_bytes_per_pass(s->spt);
        return 0;
    }
    if (host_passes) *host_passes = s->schedule == THIP_SCHED_REFERENCE ? 6 : (s->schedule == THIP_SCHED_FUSED ? 3 : (sweep_active(s) ? 1 : 2));
    // the algorithmic bytes of a pass (4 m n, or 2 m n for a 16-bit A); the padding rows of a library-owned
    // copy (at most 15 per column) are zeros that the kernel never loads
    if (host_bytes_per_pass) *host_bytes_per_pass = s->sparse ? 2 * s->nnz * (sizeof(float) + sizeof(int32_t))
                                                              : s->m * s->n * (s->is16() ? 2 : sizeof(float));
    return 0;
}

int thip_solver_schedule_in_use(thip_solver *s, int *host_schedule)
{
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    THIP_NEED_INIT();
    if (s->inited) THIP_RC(sweep_prepare(s));
    if (host_schedule) *host_schedule = (s->schedule == THIP_SCHED_SWEEP && !sweep_active(s)) ? THIP_SCHED_CARRIED : s->schedule;
    return 0;
}

int thip_solver_set_column_shard(thip_solver *s, int on)
{
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    if (s->inited && (on != 0) != s->col_shard)
        return fail(THIP_E_INVALID, "thip_solver_set_column_shard comes before thip_solver_init (the norms and preconditioners depend on it)", __FILE__, __LINE__);
    s->col_shard = on != 0;
    s->sweep_state = 0;
    return 0;
}

int thip_solver_set_sweep_min_bytes(thip_solver *s, size_t bytes)
{
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    s->sweep_min_bytes = bytes;
    s->sweep_state = 0;           // re-planned by the next run / query; sweep_prepare restarts the schedule (sw_first) when it does
    return 0;
}

int thip_solver_sweep_faults(thip_solver *s, int *host_faults, int *host_last_word, int64_t *host_restored_iter)
{
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    if (host_faults) *host_faults = s->sweep_faults;
    if (host_last_word) *host_last_word = (int)s->sweep_fault_word;
    if (host_restored_iter) *host_restored_iter = s->sweep_faults ? (int64_t)s->sweep_fault_iter : -1;
    return 0;
}

int thip_test_sweep_fault(thip_solver *s, int kind, int64_t after_sweeps, int spin_max)
{
    if (!s || kind < 0 || kind > 7) return fail(THIP_E_INVALID, "bad argument", __FILE__, __LINE__);
    if (kind == 5 || kind == 6) { s->no_fold = kind == 5; return 0; }      // 5 / 6: the termination test as its own launch / folded again
    if (kind >= 3 && kind <= 4) { s->no_merge = kind == 3; return 0; }       // 3 / 4: the step's m-kernels as two launches / merged again
    s->fault_kind = kind; s->fault_after = (kind == 2 || kind == 7) ? (long long)after_sweeps : -1;
    s->spin_max = spin_max > 0 ? spin_max : 0;
    return 0;
}

int thip_solver_set_sweep_publish(thip_solver *s, int agent_scope)
{
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    s->pub_agent = agent_scope < 0 ? -1 : (agent_scope != 0);
    return 0;
}

int thip_solver_set_gemv_autotune(thip_solver *s, int on)
{
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    s->autotune = on != 0;
    if (!on) { s->tuned = s->tuned16 = s->tuned_sp = s->tuned16_sp = false; }        // back to the shape heuristic: the plan no longer depends on timings
    return 0;
}

int thip_solver_set_lda_pad(thip_solver *s, int floats)
{
    if (!s || floats < 0) return fail(THIP_E_INVALID, "bad argument", __FILE__, __LINE__);
    if (s->inited) return fail(THIP_E_INVALID, "thip_solver_set_lda_pad must precede thip_solver_init", __FILE__, __LINE__);
    s->lda_pad = floats;
    return 0;
}

int thip_solver_overlap_info(thip_solver *s, int *host_mode, int *host_launches_per_pass, size_t *host_split_col)
{
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    THIP_RC(prepare_split(s));
    const bool split = split_active(s);
    // what the next thip_solver_run will do: modes 2 / 3 fall back to 1 / 0 where the pipeline does not apply (no
    // collective installed, sparse A, a schedule other than carried, fewer than two column chunks)
    if (host_mode) *host_mode = split ? s->overlap : (s->overlap == 2 ? 1 : (s->overlap == 3 ? 0 : s->overlap));
    if (host_launches_per_pass) *host_launches_per_pass = split ? 2 : 1;
    if (host_split_col) *host_split_col = split ? s->n1 : 0;
    return 0;
}

static int spin_allreduce(void *c, float *, size_t, void *stream)
{
    const thip_solver *s = static_cast<const thip_solver *>(c);
    if (s->spin_ticks > 0) hipLaunchKernelGGL(spin_k, dim3(1), dim3(1), 0, (hipStream_t)stream, s->spin_ticks);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int thip_test_spin_allreduce(thip_solver *s, int latency_us)
{
    THIP_NEED_INIT();
    if (!s || latency_us < 0) return fail(THIP_E_INVALID, "bad argument", __FILE__, __LINE__);
    int khz = 0;
    THIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx().device));
    if (khz <= 0) khz = 100000;
    s->spin_ticks = (long long)latency_us * khz / 1000;
    return thip_solver_set_allreduce(s, spin_allreduce, s);
}

int thip_solver_gemv_plan(const thip_solver *s, int *host_nj, int *host_blocks, float *host_ms)
{
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    const GemvHint *h = s->ahint();
    if (host_nj) *host_nj = h ? h->nj : 0;
    if (host_blocks) *host_blocks = h ? h->target_blocks : 0;
    if (host_ms) *host_ms = s->split_plan ? (s->is16() ? s->tuned16_sp_ms : s->tuned_sp_ms) : (s->is16() ? s->tuned16_ms : s->tuned_ms);
    return 0;
}

int thip_solver_sweep_plan(thip_solver *s, int *host_members, int *host_cols_per_panel, int *host_slots, float *host_ms)
{
    THIP_NEED_INIT();
    if (!s) return fail(THIP_E_INVALID, "null solver", __FILE__, __LINE__);
    if (s->inited) THIP_RC(sweep_prepare(s));
    const bool on = sweep_active(s);
    if (host_members) *host_members = on ? s->sgeom.G : 0;
    if (host_cols_per_panel) *host_cols_per_panel = on ? s->sgeom.w : 0;
    if (host_slots) *host_slots = on ? s->sgeom.nslot : 0;
    if (host_ms) *host_ms = on ? s->sw_plan_ms : 0.0f;
    return 0;
}

int thip_prof_enable(int on)
{
    THIP_NEED_INIT();
    // on = 0: off; on = N >= 1: every N-th span of each kind is timed (1: all of them)
    for (Prof *p : { &g_prof, &g_prof_psd }) {
        p->on = on > 0;
        p->period = on > 0 ? on : 1;
        p->seen = 0; p->open = false; p->iter_open = true;
        p->used = 0; p->total_ms = 0.0; p->launches = 0;
    }
    return 0;
}

static int prof_collect(Prof &p, int64_t *host_count, double *host_total_ms)
{
    THIP_TRY(hipStreamSynchronize(ctx().stream));
    for (size_t i = 0; i + 1 < p.used; i += 2) {
        float ms = 0.0f;
        THIP_TRY(hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]));
        p.total_ms += ms;
        p.launches += 1;
    }
    p.used = 0;
    if (host_count) *host_count = p.launches;
    if (host_total_ms) *host_total_ms = p.total_ms;
    return 0;
}

int thip_prof_read_psd(int64_t *host_spans, double *host_total_ms)
{
    THIP_NEED_INIT();
    return prof_collect(g_prof_psd, host_spans, host_total_ms);
}

int thip_prof_read(int64_t *host_launches, double *host_total_ms)
{
    THIP_NEED_INIT();
    return prof_collect(g_prof, host_launches, host_total_ms);
}

int thip_solver_destroy(thip_solver *s)
{
    if (!s) return 0;
    if (ctx().inited) hipStreamSynchronize(ctx().stream);
    if (s->side) { hipStreamSynchronize(s->side); hipStreamDestroy(s->side); }
    if (s->ev_in) hipEventDestroy(s->ev_in);
    if (s->ev_out) hipEventDestroy(s->ev_out);
    for (int k = 0; k < 4; ++k) { if (s->sev_in[k]) hipEventDestroy(s->sev_in[k]); if (s->sev_out[k]) hipEventDestroy(s->sev_out[k]); }
    if (s->gflags) hipFree(s->gflags);
    hipFree(s->cls); hipFree(s->soc_beg); hipFree(s->soc_end); hipFree(s->rot_beg); hipFree(s->rot_end);
    for (auto &g : s->psd_groups) hipFree(g.dev_offs);
    hipFree(s->grp_beg); hipFree(s->grp_end); hipFree(s->psd_work); hipFree(s->arena); hipFree(s->part);
    hipFree(s->gemv_scr); hipFree(s->dst); hipFree(s->Apad); if (s->A16_owned) hipFree(s->A16); if (s->inv_s_owned) hipFree(s->inv_s);
    hipFree(s->sw_partT);
    hipFree(s->sw_partH); hipFree(s->sw_gran); hipFree(s->sw_census); hipFree(s->sw_part); hipFree(s->cs_buf);
    if (s->hst) hipHostFree(s->hst);
    if (s->hflags) hipHostFree(s->hflags);
    hipFree(s->snap); hipFree(s->snap_st);
    delete s;
    return 0;
}

}  // extern "C"
