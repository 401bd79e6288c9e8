// thip_trieig.hip -- eigenvalues and eigenvectors of the symmetric tridiagonal T = Q^T M Q that the Householder
// reduction of thip_eig.hip leaves behind, ON THE DEVICE (the closure path of LinAlgEx::map_eig, linalg_ex.rs:64-65; the
// routines the reference calls here are dsyevr / syevdx, f64lapack.rs:78-108, f32cuda.rs:253-263).
//
// Round 2 sent (d, e) to the host, ran the implicit QL recurrence there and replayed its 236 k Givens rotations on the
// device: 4.5 ms of a scalar chain at k = 500 that nothing parallel can shorten.  This file replaces the chain by two
// embarrassingly parallel stages in f64 (T carries the f32 round-off of the reduction, ~1e-7 ||T||; f64 makes every
// cluster of that noise a set of well separated eigenvalues):
//   1. te_bisect_k: eigenvalue k of its unreduced block by Sturm-count MULTIsection -- 64 shifts per round (6 bits),
//      9 rounds to ulp ||T||; the count is the inertia of the twisted factorisation, formed from both ends by two waves
//      (dstebz's pivmin-guarded recurrence, half the chain each);
//   2. te_vec_k: ONE eigenvector per eigenvalue from the twisted factorisation of T - lambda I (the dlar1v step of
//      MRRR without the representation tree): forward and backward pivots, the twist index r = argmin |gamma_r|,
//      z from the two multiplier chains; one LANE per eigenvector, lanes in lockstep over the rows.
// Nothing orthogonalises the vectors explicitly: eigenvalues further apart than ~1e-10 ||T|| give orthogonality to
// f32 level by themselves, and what they do not give is measured, not assumed -- thip_eig.hip forms
// P = 3/2 I - 1/2 Z Z^T (the Newton-Schulz polish operator, an MFMA GEMM) from the back-transformed vectors and reads
// ||Z Z^T - I||_F and the largest residual |gamma_r| / ||z|| back with the eigenvalues: small -> done, moderate -> Z <- P Z
// and measure again, anything else (glued Wilkinson-type pairs that agree to 1e-15) -> the QL engine of round 2.
// n orthonormal vectors with small residuals ARE the decomposition, so the certificate is complete.
// tools/tridiag_vec_probe.py is the numpy restatement this was designed on (12 spectra, n up to 500).
#include "thip_common.h"

#include <cfloat>

namespace thip {

namespace {

constexpr int TE_MAXN = 2048;
constexpr double TE_EPS = 2.220446049250313e-16;

__device__ __forceinline__ double fix_pivot(double x, double piv)
{
    if (fabs(x) < piv) return x < 0.0 ? -piv : piv;
    return x;
}

// 1 / x: v_rcp_f64 and two Newton steps.  The compiler's IEEE division is ~25 dependent instructions (scaling, fix-ups);
// the recurrences below are latency chains of one reciprocal per row, and their arguments are normal numbers by
// construction (|x| >= pivmin), so the plain form is enough -- 6 dependent instructions.
__device__ __forceinline__ double rcp2(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}

// (d, e) f32 -> f64 copies with negligible couplings set to zero, the unreduced block [lo, hi) of every index, the
// Gershgorin interval and norm of that block; head[0] = pivmin, head[1] = max |bound|, cert words zeroed.
__global__ __launch_bounds__(1024) void te_prep_k(int n, const float *__restrict__ d32, const float *__restrict__ e32,
                                                  double *__restrict__ dd, double *__restrict__ ee, int2 *__restrict__ blk,
                                                  double *__restrict__ bnd, double *__restrict__ head,
                                                  unsigned *__restrict__ cert_bits)
{
    __shared__ int slo[TE_MAXN], shi[TE_MAXN];
    __shared__ double sd[TE_MAXN], se[TE_MAXN];
    __shared__ double red[16];
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += 1024) sd[i] = (double)d32[i];
    __syncthreads();
    double emax = 0.0;
    for (int i = tid; i < n; i += 1024) {
        double e = (i + 1 < n) ? (double)e32[i] : 0.0;
        if (i + 1 < n && !(fabs(e) > TE_EPS * (fabs(sd[i]) + fabs(sd[i + 1])))) e = 0.0;      // also catches NaN
        se[i] = e;
        emax = fmax(emax, fabs(e));
    }
    // block max of |e|
    {
        const int lane = tid & 63, w = tid >> 6;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) emax = fmax(emax, __shfl_xor(emax, o, 64));
        __syncthreads();
        if (lane == 0) red[w] = emax;
        __syncthreads();
        emax = 0.0;
        for (int k = 0; k < 16; ++k) emax = fmax(emax, red[k]);
    }
    const double pivmin = DBL_MIN * fmax(1.0, emax * emax);
    for (int i = tid; i < n; i += 1024) {
        slo[i] = (i == 0 || se[i - 1] == 0.0) ? i : 0;
        shi[i] = (se[i] == 0.0) ? i + 1 : n;
    }
    __syncthreads();
    // inclusive prefix max of slo, suffix min of shi (Hillis-Steele; n <= 2048 = two elements per thread)
    for (int off = 1; off < n; off <<= 1) {
        int a[2], b[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 1024;
            if (i < n) {
                a[u] = slo[i]; b[u] = shi[i];
                if (i >= off) a[u] = max(a[u], slo[i - off]);
                if (i + off < n) b[u] = min(b[u], shi[i + off]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 1024;
            if (i < n) { slo[i] = a[u]; shi[i] = b[u]; }
        }
        __syncthreads();
    }
    // Gershgorin bounds of every block: segmented min / max scans over [lo, i], read back at the block's last row.
    // (d, e) go out to global memory first; the scans then run in the LDS arrays that held them.
    double *smin = sd, *smax = se;
    {
        double vmin[2], vmax[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 1024;
            if (i < n) {
                const double r = (i > 0 ? fabs(se[i - 1]) : 0.0) + fabs(se[i]);      // a cut coupling is 0
                vmin[u] = sd[i] - r;
                vmax[u] = sd[i] + r;
                dd[i] = sd[i];
                ee[i] = se[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 1024;
            if (i < n) { smin[i] = vmin[u]; smax[i] = vmax[u]; }
        }
        __syncthreads();
    }
    for (int off = 1; off < n; off <<= 1) {
        double a[2], b[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 1024;
            if (i < n) {
                a[u] = smin[i]; b[u] = smax[i];
                if (i - off >= slo[i]) { a[u] = fmin(a[u], smin[i - off]); b[u] = fmax(b[u], smax[i - off]); }
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 1024;
            if (i < n) { smin[i] = a[u]; smax[i] = b[u]; }
        }
        __syncthreads();
    }
    double tmax = 0.0;
    for (int i = tid; i < n; i += 1024) {
        const int lo = slo[i], hi = shi[i];
        const double gl = smin[hi - 1], gu = smax[hi - 1];
        const double tn = fmax(fabs(gl), fabs(gu));
        const double margin = 2.1 * tn * TE_EPS * (double)(hi - lo) + 2.1 * pivmin;
        blk[i] = make_int2(lo, hi);
        bnd[3 * (size_t)i + 0] = gl - margin;
        bnd[3 * (size_t)i + 1] = gu + margin;
        bnd[3 * (size_t)i + 2] = tn;
        tmax = fmax(tmax, tn);
    }
    {
        const int lane = tid & 63, w = tid >> 6;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tmax = fmax(tmax, __shfl_xor(tmax, o, 64));
        __syncthreads();
        if (lane == 0) red[w] = tmax;
        __syncthreads();
        tmax = 0.0;
        for (int k = 0; k < 16; ++k) tmax = fmax(tmax, red[k]);
    }
    if (tid == 0) { head[0] = pivmin; head[1] = tmax; cert_bits[0] = 0u; cert_bits[1] = 0u; }
}

// TWO waves per eigenvalue (index t is eigenvalue number t - lo of its block [lo, hi)): the number of eigenvalues below a
// shift is the inertia of T - sigma I, and the twisted factorisation gives it from BOTH ends at once -- the negative pivots
// D+ of rows [lo, mid), the negative pivots D- of rows (mid, hi), and the sign of gamma_mid = (d_mid - sigma) - e^2/D+ -
// e^2/D- (Sylvester: the factorisation is a congruence).  The forward and the backward recurrences are independent chains
// of half the length; the waves exchange (count, e^2 / last pivot) through LDS once per round.  Four eigenvalues (eight
// waves) per workgroup; a round is two barriers, whether an eigenvalue has converged or not.
__global__ __launch_bounds__(512) void te_bisect_k(int n, const double *__restrict__ dd, const double *__restrict__ ee,
                                                   const int2 *__restrict__ blk, const double *__restrict__ bnd,
                                                   const double *__restrict__ head, double *__restrict__ lam,
                                                   float *__restrict__ w32)
{
    extern __shared__ double te_sh[];
    double2 *sde = reinterpret_cast<double2 *>(te_sh);            // (d_i, e_{i-1}^2): one 16-byte LDS read per row
    __shared__ double xterm[4][64];
    __shared__ int xcnt[4][64];
    __shared__ double xlohi[4][2];
    __shared__ int xdone[4];
    for (int i = threadIdx.x; i < n; i += 512) { const double e = i > 0 ? ee[i - 1] : 0.0; sde[i] = make_double2(dd[i], e * e); }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q4 = wave >> 1, dir = wave & 1;
    const int t = blockIdx.x * 4 + q4;
    const bool live = t < n;
    int2 b = make_int2(0, 1);
    if (live) b = blk[t];
    const int k = t - b.x, m = b.y - b.x, mid = b.x + m / 2;
    const double pivmin = head[0];
    double lo = 0.0, hi = 0.0, tol = 0.0;
    bool done = !live || m == 1;
    if (live && m > 1) {
        lo = bnd[3 * (size_t)t]; hi = bnd[3 * (size_t)t + 1];
        // absolute tolerance, relative to the block's norm (dstebz's default abstol = ulp ||T||): T itself is only known to
        // eps ||T||, and a tolerance relative to |lambda| would spend four more rounds on every eigenvalue near zero
        tol = TE_EPS * bnd[3 * (size_t)t + 2] + 2.0 * pivmin;
    }
    for (int round = 0; round < 13; ++round) {
        const double w = hi - lo;
        if (!done && w <= tol) done = true;
        const double sig = lo + w * ((double)(lane + 1) * (1.0 / 65.0));
        int cnt = 0;
        double term = 0.0;
        if (!done) {
            if (dir == 0) {
                // rows b.x .. mid - 1 (at least one): D+ and its negative pivots; term = e_{mid-1}^2 / D+[mid - 1]
                double q = sde[b.x].x - sig;
                if (fabs(q) < pivmin) q = -pivmin;
                cnt = q < 0.0 ? 1 : 0;
#pragma unroll 4
                for (int i = b.x + 1; i < mid; ++i) {
                    const double2 de = sde[i];
                    q = fma(-de.y, rcp2(q), de.x - sig);
                    if (fabs(q) < pivmin) q = -pivmin;
                    cnt += q < 0.0 ? 1 : 0;
                }
                term = sde[mid].y * rcp2(q);
            } else if (mid + 1 < b.y) {
                // rows b.y - 1 .. mid + 1: D- and its negative pivots; term = e_mid^2 / D-[mid + 1]
                double q = sde[b.y - 1].x - sig;
                if (fabs(q) < pivmin) q = -pivmin;
                cnt = q < 0.0 ? 1 : 0;
#pragma unroll 4
                for (int i = b.y - 2; i > mid; --i) {
                    const double e2 = sde[i + 1].y;
                    q = fma(-e2, rcp2(q), sde[i].x - sig);
                    if (fabs(q) < pivmin) q = -pivmin;
                    cnt += q < 0.0 ? 1 : 0;
                }
                term = sde[mid + 1].y * rcp2(q);
            }
            if (dir == 1) { xterm[q4][lane] = term; xcnt[q4][lane] = cnt; }
        }
        __syncthreads();
        if (dir == 0) {
            if (!done) {
                double g = (sde[mid].x - sig) - term - xterm[q4][lane];
                if (fabs(g) < pivmin) g = -pivmin;
                cnt += xcnt[q4][lane] + (g < 0.0 ? 1 : 0);
                const unsigned long long mask = __ballot(cnt <= k);          // lanes whose shift is still a lower bound
                const int idx = (mask == ~0ull) ? 64 : __ffsll((long long)~mask) - 1;
                const double below = __shfl(sig, idx > 0 ? idx - 1 : 0, 64);
                const double above = __shfl(sig, idx < 64 ? idx : 63, 64);
                if (idx > 0) lo = below;
                if (idx < 64) hi = above;
            }
            if (lane == 0) { xlohi[q4][0] = lo; xlohi[q4][1] = hi; xdone[q4] = done ? 1 : 0; }
        }
        __syncthreads();
        lo = xlohi[q4][0]; hi = xlohi[q4][1];
        done = xdone[q4] != 0;
        // (all four eigenvalues done: the remaining rounds are two barriers each; not worth a vote)
    }
    if (live && dir == 0 && lane == 0) {
        const double l = m == 1 ? sde[b.x].x : 0.5 * (lo + hi);
        lam[t] = l; w32[t] = (float)l;
    }
}

constexpr int TE_B = 24;          // rows fetched per batch by te_vec_k's unchained passes (one memory round trip per batch)
// 64 eigenvectors per workgroup, one LANE per eigenvector in each of two waves: wave 0 runs the forward pivots D+ and the
// part of z above the twist, wave 1 the backward pivots D- and the part below -- the four recurrences of the twisted
// factorisation are two pairs of independent chains.  Dp and Dm are n x n doubles, element (row i, vector t) at
// [i * n + t] (coalesced across the lanes).  On return column t of Dp holds the unnormalised vector on rows [lo, r],
// column t of Dm on rows (r, hi); twist[t] = r, nrm[t] = its 2-norm; cert_bits[0] = max over t of
// |gamma_r| / (||z|| ||T||) as float bits.  Loads run TE_B rows ahead of the chains (the multipliers L+ = e / D+ and
// U- = e / D- of the z chains are formed off-chain from them).
__global__ __launch_bounds__(128) void te_vec_k(int n, const double *__restrict__ dd, const double *__restrict__ ee,
                                                const int2 *__restrict__ blk, const double *__restrict__ bnd,
                                                const double *__restrict__ head, const double *__restrict__ lam,
                                                double *__restrict__ Dp, double *__restrict__ Dm, double *__restrict__ nrm,
                                                int *__restrict__ twist, unsigned *__restrict__ cert_bits)
{
    extern __shared__ double te_sh[];
    double *sd = te_sh, *se = te_sh + n;
    __shared__ double xbest[2][64], xg[2][64], xss[64];
    __shared__ int xr[2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < n; i += 128) { sd[i] = dd[i]; se[i] = ee[i]; }
    __syncthreads();
    const int t = blockIdx.x * 64 + lane;
    const bool act = t < n;
    const int tt = act ? t : n - 1;                  // address of an inactive lane's (discarded) loads
    int lo = n, hi = 0;
    double l = 0.0, piv = 1.0;
    if (act) {
        const int2 b = blk[t];
        lo = b.x; hi = b.y;
        l = lam[t];
        piv = TE_EPS * bnd[3 * (size_t)t + 2] * 1.0e-3 + DBL_MIN;
    }
    int wlo = lo, whi = hi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { wlo = min(wlo, __shfl_xor(wlo, o, 64)); whi = max(whi, __shfl_xor(whi, o, 64)); }
    // (the LDS operands of four rows are read ahead of the chain: a read inside the predicated body would add its latency
    // to every link)
    if (wave == 0) {
        // D+[lo] = d - l; D+[i + 1] = (d[i + 1] - l) - e_i^2 / D+[i]
        double D = 0.0;
        for (int i0 = wlo; i0 < whi; i0 += 4) {
            double dv[4], ev[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u, whi - 1);
                dv[u] = sd[i];
                ev[u] = i > 0 ? se[i - 1] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                if (i < whi && act && i >= lo && i < hi) {
                    const double e2 = i > lo ? ev[u] * ev[u] : 0.0;
                    D = fma(-e2, i > lo ? rcp2(fix_pivot(D, piv)) : 0.0, dv[u] - l);
                    Dp[(size_t)i * n + t] = D;
                }
            }
        }
    } else {
        // D-[hi - 1] = d - l; D-[i] = (d[i] - l) - e_i^2 / D-[i + 1]
        double D = 0.0;
        for (int i0 = whi - 1; i0 >= wlo; i0 -= 4) {
            double dv[4], ev[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = max(i0 - u, wlo);
                dv[u] = sd[i];
                ev[u] = se[i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 - u;
                if (i >= wlo && act && i >= lo && i < hi) {
                    const double e2 = i < hi - 1 ? ev[u] * ev[u] : 0.0;
                    D = fma(-e2, i < hi - 1 ? rcp2(fix_pivot(D, piv)) : 0.0, dv[u] - l);
                    Dm[(size_t)i * n + t] = D;
                }
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    // gamma_i = D+[i] + D-[i] - (d_i - l): no chain; the waves take half of the rows each
    {
        double best = DBL_MAX, gbest = 0.0;
        int r = lo;
        const int mid = (wlo + whi) >> 1;
        const int i_beg = wave == 0 ? wlo : mid, i_end = wave == 0 ? mid : whi;
        for (int i0 = i_beg; i0 < i_end; i0 += TE_B) {
            double a[TE_B], b[TE_B];
#pragma unroll
            for (int u = 0; u < TE_B; ++u) {
                const int i = min(i0 + u, i_end - 1);
                a[u] = __builtin_nontemporal_load(&Dp[(size_t)i * n + tt]) - sd[i];
                b[u] = __builtin_nontemporal_load(&Dm[(size_t)i * n + tt]);
            }
#pragma unroll
            for (int u = 0; u < TE_B; ++u) {
                const int i = i0 + u;
                if (i < i_end && act && i >= lo && i < hi) {
                    const double g = a[u] + b[u] + l;
                    const double ag = fabs(g);
                    if (ag < best) { best = ag; gbest = g; r = i; }      // NaN never wins
                }
            }
        }
        xbest[wave][lane] = best; xg[wave][lane] = gbest; xr[wave][lane] = r;
    }
    __syncthreads();
    int r;
    double gbest;
    {
        const bool first = xbest[0][lane] <= xbest[1][lane];
        r = first ? xr[0][lane] : xr[1][lane];
        gbest = first ? xg[0][lane] : xg[1][lane];
    }
    __syncthreads();
    double z = 1.0, ss = 0.0;
    if (wave == 0) {
        // upwards: z_i = -(e_i / D+[i]) z_{i+1}, i = r - 1 .. lo; z_r = 1 is stored in Dp as well
        ss = 1.0;
        if (act && hi > lo) Dp[(size_t)r * n + t] = 1.0;
        for (int i0 = whi - 2; i0 >= wlo; i0 -= TE_B) {
            double dv[TE_B];
#pragma unroll
            for (int u = 0; u < TE_B; ++u) {
                const int i = max(i0 - u, wlo);
                dv[u] = -(se[i] * rcp2(fix_pivot(__builtin_nontemporal_load(&Dp[(size_t)i * n + tt]), piv)));
            }
#pragma unroll
            for (int u = 0; u < TE_B; ++u) {
                const int i = i0 - u;
                if (i >= wlo && act && i >= lo && i < r) {
                    z = dv[u] * z;
                    Dp[(size_t)i * n + t] = z;
                    ss = fma(z, z, ss);
                }
            }
        }
    } else {
        // downwards: z_{i+1} = -(e_i / D-[i + 1]) z_i, i = r .. hi - 2
        for (int i0 = wlo; i0 < whi - 1; i0 += TE_B) {
            double dv[TE_B];
#pragma unroll
            for (int u = 0; u < TE_B; ++u) {
                const int i = min(i0 + u, whi - 2);
                dv[u] = -(se[i] * rcp2(fix_pivot(__builtin_nontemporal_load(&Dm[(size_t)(i + 1) * n + tt]), piv)));
            }
#pragma unroll
            for (int u = 0; u < TE_B; ++u) {
                const int i = i0 + u;
                if (i < whi - 1 && act && i >= r && i < hi - 1) {
                    z = dv[u] * z;
                    Dm[(size_t)(i + 1) * n + t] = z;
                    ss = fma(z, z, ss);
                }
            }
        }
        xss[lane] = ss;
    }
    __syncthreads();
    if (wave == 0 && act) {
        const double nz = sqrt(ss + xss[lane]);
        nrm[t] = nz;
        twist[t] = r;
        const double tn = head[1];
        float res = tn > 0.0 ? (float)(fabs(gbest) / (nz * tn)) : 0.0f;
        if (!(res >= 0.0f) || !(nz > 0.0) || nz > DBL_MAX) res = __builtin_inff();          // NaN / overflow -> fails the certificate
        atomicMax(cert_bits, __float_as_uint(res));
    }
}

// V0(i, t) = z_t[i] / ||z_t|| on the rows of t's block, 0 elsewhere; column-major ld x ld f32, zero padded.
// 32 x 32 tiles through LDS: reads run along t, writes along i.
__global__ __launch_bounds__(256) void te_pack_k(int n, int ld, const double *__restrict__ Dp, const double *__restrict__ Dm,
                                                 const double *__restrict__ nrm, const int *__restrict__ twist,
                                                 const int2 *__restrict__ blk, float *__restrict__ V0)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int i0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    {
        const int t = t0 + tx;
        int2 b = make_int2(0, 0);
        double inv = 0.0;
        int r = 0;
        if (t < n) { b = blk[t]; inv = 1.0 / nrm[t]; r = twist[t]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + ty + 8 * u;
            float v = 0.0f;
            if (t < n && i >= b.x && i < b.y) v = (float)((i <= r ? Dp : Dm)[(size_t)i * n + t] * inv);
            tile[ty + 8 * u][tx] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = t0 + ty + 8 * u, i = i0 + tx;
        V0[(size_t)t * ld + i] = tile[tx][ty + 8 * u];
    }
}

// partial sums of ||P - I_n||_F^2 (P = 3/2 I - 1/2 Z Z^T, so this is ||Z Z^T - I||_F^2 / 4): part[blockIdx.x]
__global__ __launch_bounds__(256) void te_orth_k(int n, int ld, const float *__restrict__ P, float *__restrict__ part)
{
    __shared__ double shd[16];
    double acc = 0.0;
    const size_t tot = (size_t)ld * ld;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i % ld), c = (int)(i / ld);
        const double v = (double)P[i] - ((r == c && r < n) ? 1.0 : 0.0);
        acc += v * v;
    }
    acc = block_sum_d(acc, shd);
    if (threadIdx.x == 0) part[blockIdx.x] = (float)acc;
}

// mapped eigenvalues for the two closures that exist (cone_psd.rs:69-76, matbuild/mod.rs:231-238)
__global__ void te_map_k(int n, int ld, int map_kind, const float *__restrict__ w, float *__restrict__ e)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ld) return;
    float v = 0.0f;
    if (i < n) {
        const float lam = w[i];
        if (map_kind == 0) v = lam > 0.0f ? lam : 0.0f;
        else if (map_kind == 1) v = lam > 0.0f ? sqrtf(lam) : 0.0f;
    }
    e[i] = v;
}

}  // namespace

size_t tri_eigen_scratch_floats(int n)
{
    // doubles: dd, ee, lam, nrm (n each), bnd (3 n), head (4), Dp and Dm (n^2 each); int2 blk (n) = n doubles; twist (n ints)
    return 2 * ((size_t)9 * n + 8 + 2 * (size_t)n * n) + 16;
}

// T = (d, e) -> eigenvalues w32[0 .. n), eigenvectors as the columns of V0 (ld x ld, zero padded); cert_bits[0] receives the
// largest relative residual (float bits), cert_bits[1] is zeroed for the caller.  scr: tri_eigen_scratch_floats(n) floats.
int tri_eigen(hipStream_t st, int n, int ld, const float *d, const float *e, float *w32, float *V0, unsigned *cert_bits,
              float *scr)
{
    if (n > TE_MAXN) return fail(THIP_E_INVALID, "tri_eigen: order above 2048", __FILE__, __LINE__);
    double *base = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(scr) + 15) & ~(uintptr_t)15);
    double *dd = base, *ee = dd + n, *lam = ee + n, *nrm = lam + n, *bnd = nrm + n, *head = bnd + 3 * (size_t)n;
    int2 *blk = reinterpret_cast<int2 *>(head + 4);
    int *twist = reinterpret_cast<int *>(head + 4 + n);
    double *Dp = head + 4 + 2 * (size_t)n, *Dm = Dp + (size_t)n * n;
    hipLaunchKernelGGL(te_prep_k, dim3(1), dim3(1024), 0, st, n, d, e, dd, ee, blk, bnd, head, cert_bits);
    const size_t lds = 2 * (size_t)n * sizeof(double);
    hipLaunchKernelGGL(te_bisect_k, dim3((unsigned)((n + 3) / 4)), dim3(512), lds, st, n, dd, ee, blk, bnd, head, lam, w32);
    hipLaunchKernelGGL(te_vec_k, dim3((unsigned)((n + 63) / 64)), dim3(128), lds, st, n, dd, ee, blk, bnd, head, lam, Dp, Dm,
                       nrm, twist, cert_bits);
    hipLaunchKernelGGL(te_pack_k, dim3((unsigned)(ld / 32), (unsigned)(ld / 32)), dim3(256), 0, st, n, ld, Dp, Dm, nrm, twist,
                       blk, V0);
    THIP_LAUNCH_CHECK();
    return 0;
}

int tri_orth_partials(hipStream_t st, int n, int ld, const float *P, float *part, int nblocks)
{
    hipLaunchKernelGGL(te_orth_k, dim3((unsigned)nblocks), dim3(256), 0, st, n, ld, P, part);
    THIP_LAUNCH_CHECK();
    return 0;
}

int tri_map(hipStream_t st, int n, int ld, int map_kind, const float *w, float *e)
{
    hipLaunchKernelGGL(te_map_k, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, st, n, ld, map_kind, w, e);
    THIP_LAUNCH_CHECK();
    return 0;
}

}  // namespace thip
