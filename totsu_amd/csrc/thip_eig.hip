// thip_eig.hip -- LinAlgEx::map_eig (totsu_core/src/linalg_ex.rs:44-65) and ConePSD::proj
// (totsu_core/src/cone_psd.rs:56-79) on the device.
//
// Contract (F64LAPACK recipe, totsu_f64lapack/src/f64lapack.rs:172-255; CUDA: f32cuda.rs:196-370):
//   packed upper (by columns) --vec_to_mat, diag*scale--> M ; M = Z diag(w) Z^T ; R = sum_i map(w_i) z_i z_i^T ;
//   R --diag/scale, pack upper--> packed.
//
// Two engines:
//  (1) eigen-decomposition by parallel one-sided Jacobi (Hestenes) on the definite shift M + sigma I,
//      sigma > ||M||_F: for a positive definite matrix the right singular vectors ARE the eigenvectors and no
//      +/-lambda pair can mix.  Round-robin ordering gives n/2 independent column-pair rotations per step, one
//      wavefront per pair (3 width-64 shuffle-tree dots, then a rotation of two G and two V columns).  n <= 64
//      runs as one workgroup with G and V in LDS; larger n as one launch per step.  This engine serves the
//      arbitrary-closure contract (thip_eig_decompose / thip_eig_rebuild) and map_kind 1 (sqrt).
//  (2) PSD projection without eigenvectors: P = (M + M sign(M)) / 2, sign(M) by an odd matrix polynomial
//      iteration -- a chain of n x n x n f32 GEMMs on v_mfma_f32_32x32x2_f32.  This is where the matrix cores
//      are a real dense contraction; a Householder/QR chain at k = 500 is O(k) dependent latency-bound
//      steps while this is 50 GEMMs.  Used by ConePSD::proj for n > 20 (measured cross-over with (1)).
#include "thip_common.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

using namespace thip;

namespace {

constexpr int BLK = 256;
constexpr int SMALL_N = 64;          // largest order the single-workgroup Jacobi kernel holds in LDS
constexpr int POLAR_MIN_N = 20;      // from here on the MFMA polar chain (~0.2 ms flat) beats it for PSD projections
constexpr int POLAR_SMALL_MIN_N = 1; // orders 1 .. 64: the polar chain inside one workgroup, 29-75 us (Jacobi: 16-21 us up to order 4, 31 at 5,
                                     // 200 at 16 -- and not scale invariant: its column products underflow for entries of 1e-18)
constexpr int MAX_SWEEPS = 18;

// ld of an order-n operand: n rounded up to 64; above 512 the chain's kernels walk K in nch = ceil(ld / 512) chunks of
// 4 waves x KW (KW a multiple of 16, <= 128), so ld / 64 must be a multiple of nch (704 -> 768, 1088 -> 1152, 1600 -> 1792)
__host__ __device__ inline size_t np_of(size_t n)
{
    size_t ld = (n + 63) / 64 * 64;
    while (ld > 512 && (ld / 64) % ((ld + 511) / 512) != 0) ld += 64;
    return ld;
}
// row pitch of the ld x ld operands of the polar chain.  A power-of-two pitch (2 KB at ld = 512) looked 6 % slower than
// 544 or 576 floats in the one-tile probe of round 2 (tools/gemm_phases.hip); built in round 3 and measured in the
// production chain (32 x 64 blocks, dwordx2 loads of operand b): pads of 0 .. 160 floats all give 0.350-0.355 ms per k = 500
// projection and 1270-1286 iter/s on the SDP -- neutral, so the default stays 0.  THIP_PSD_PITCH_PAD = floats (experiments).
inline size_t pitch_of(size_t ld)
{
    static const size_t pad = getenv("THIP_PSD_PITCH_PAD") ? (size_t)atoi(getenv("THIP_PSD_PITCH_PAD")) : 0;
    return (ld == 256 || ld == 512) ? ld + pad : ld;
}

// round-robin (circle method) pairing on n_even players: step in [0, n_even-1), k in [0, n_even/2)
__device__ __forceinline__ void rr_pair(int n_even, int step, int k, int &p, int &q)
{
    const int n1 = n_even - 1;
    if (k == 0) { p = step; q = n1; }
    else { p = (step + k) % n1; q = (step - k + n1) % n1; }
    if (p > q) { const int t = p; p = q; q = t; }
}

// one wave orthogonalises columns p, q of G (and applies the same rotation to V); returns 1 if it rotated
__device__ __forceinline__ int rotate_pair(float *G, float *V, int ld, int n, int p, int q, int lane)
{
    float *gp = G + (size_t)p * ld, *gq = G + (size_t)q * ld;
    float a = 0.0f, b = 0.0f, g = 0.0f;
    for (int r = lane; r < n; r += 64) {
        const float x = gp[r], y = gq[r];
        a = fmaf(x, x, a); b = fmaf(y, y, b); g = fmaf(x, y, g);
    }
    a = wave_sum(a); b = wave_sum(b); g = wave_sum(g);
    // rotate when the columns are not orthogonal to working precision
    const float thr = 1.0e-7f;
    if (!(g * g > thr * thr * a * b) || !(g * g > 1.0e-37f)) return 0;
    const float zeta = (b - a) / (2.0f * g);
    const float t = (zeta > 0.0f ? 1.0f : -1.0f) / (fabsf(zeta) + sqrtf(1.0f + zeta * zeta));
    const float c = 1.0f / sqrtf(1.0f + t * t);
    const float s = c * t;
    float *vp = V + (size_t)p * ld, *vq = V + (size_t)q * ld;
    for (int r = lane; r < n; r += 64) {
        const float x = gp[r], y = gq[r];
        gp[r] = c * x - s * y;
        gq[r] = s * x + c * y;
        const float u = vp[r], w = vq[r];
        vp[r] = c * u - s * w;
        vq[r] = s * u + c * w;
    }
    return 1;
}

// packed upper (by columns) -> full symmetric G (ld x ld, zero padded), diag * scale ; V = I ; block partials of
// the squared Frobenius norm
// Batched launches (blockIdx.z = item): item z works on packed + z * ps and on work pointers + z * ws.
__global__ void unpack_k(int n, int ld, const float *__restrict__ packed, int has_scale, float scale,
                         float *__restrict__ G, float *__restrict__ V, float *__restrict__ part,
                         const int *__restrict__ stop, size_t ws = 0, ptrdiff_t ps = 0, int pitch = 0)
{
    if (pitch == 0) pitch = ld;
    if (stop != nullptr && *stop != 0) return;
    packed += (ptrdiff_t)blockIdx.z * ps; G += blockIdx.z * ws; part += blockIdx.z * ws;
    if (V) V += blockIdx.z * ws;
    __shared__ double shd[16];
    double acc = 0.0;
    const size_t tot = (size_t)ld * ld;
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < tot; i += (size_t)gridDim.x * BLK) {
        const int r = (int)(i % ld), c = (int)(i / ld);
        float v = 0.0f;
        if (r < n && c < n) {
            const int lo = r < c ? r : c, hi = r < c ? c : r;
            v = packed[(size_t)hi * (hi + 1) / 2 + lo];
            if (r == c && has_scale) v *= scale;
            acc += (double)v * (double)v;
        }
        G[(size_t)c * pitch + r] = v;
        if (V) V[(size_t)c * pitch + r] = (r == c && r < n) ? 1.0f : 0.0f;
    }
    // the block's 2-norm, not its sum of squares: representable in f32 whenever the entries are (the squares of a
    // block of 1e-25s are not, and a norm of 0 for a nonzero matrix makes the projection return M / 2)
    acc = block_sum_d(acc, shd);
    if (threadIdx.x == 0) part[blockIdx.x] = (float)sqrt(acc);
}

// sc[0] = ||M||_F, sc[1] = sigma = 1.01 ||M||_F + tiny ; G += sigma I (when shift != 0)
__global__ void shift_k(int n, int ld, int np, const float *__restrict__ part, float *__restrict__ G,
                        float *__restrict__ sc, int shift, const int *__restrict__ stop, size_t ws = 0)
{
    if (stop != nullptr && *stop != 0) return;
    part += blockIdx.z * ws; G += blockIdx.z * ws; sc += blockIdx.z * ws;
    __shared__ double shd[16];
    double acc = 0.0;
    for (int k = threadIdx.x; k < np; k += blockDim.x) acc += (double)part[k] * (double)part[k];
    acc = block_sum_d(acc, shd);
    const float fro = (float)sqrt(acc);
    const float sigma = 1.01f * fro + 1.0e-30f;
    if (threadIdx.x == 0) { sc[0] = fro; sc[1] = sigma; sc[2] = 0.0f; }
    if (shift)
        for (int i = threadIdx.x; i < n; i += blockDim.x) G[(size_t)i * ld + i] += sigma;
}

// G <- G / sigma = M / sigma + I: the one-sided Jacobi works on products of columns, which underflow for a matrix of
// 1e-18s and overflow for one of 1e+20s; normalised, every entry is O(1) whatever the scale of M (eigvals_k undoes it)
__global__ void normalise_k(size_t tot, float *__restrict__ G, const float *__restrict__ sc, const int *__restrict__ stop)
{
    if (stop != nullptr && *stop != 0) return;
    const float sigma = sc[1];
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < tot; i += (size_t)gridDim.x * BLK) G[i] = G[i] / sigma;
}

// n <= 64: whole decomposition in one workgroup, G and V staged in LDS
__global__ __launch_bounds__(BLK) void jacobi_small_k(int n, int ld, float *__restrict__ Gg, float *__restrict__ Vg,
                                                     const int *__restrict__ stop)
{
    if (stop != nullptr && *stop != 0) return;
    __shared__ float G[SMALL_N * SMALL_N];
    __shared__ float V[SMALL_N * SMALL_N];
    __shared__ int rotated;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n * n; i += BLK) {
        const int r = i % n, c = i / n;
        G[c * n + r] = Gg[(size_t)c * ld + r];
        V[c * n + r] = (r == c) ? 1.0f : 0.0f;
    }
    __syncthreads();
    const int n_even = (n + 1) & ~1;
    for (int sweep = 0; sweep < MAX_SWEEPS; ++sweep) {
        if (tid == 0) rotated = 0;
        __syncthreads();
        for (int step = 0; step < n_even - 1; ++step) {
            int cnt = 0;
            for (int k = wave; k < n_even / 2; k += 4) {
                int p, q;
                rr_pair(n_even, step, k, p, q);
                if (q < n) cnt += rotate_pair(G, V, n, n, p, q, lane);
            }
            if (lane == 0 && cnt) atomicAdd(&rotated, cnt);
            __syncthreads();
        }
        const int r = rotated;
        __syncthreads();
        if (r == 0) break;
    }
    for (int i = tid; i < n * n; i += BLK) {
        const int r = i % n, c = i / n;
        Gg[(size_t)c * ld + r] = G[c * n + r];
        Vg[(size_t)c * ld + r] = V[c * n + r];
    }
}

// n > 64: one launch per round-robin step, one wave per pair
__global__ __launch_bounds__(BLK) void jacobi_step_k(int n, int ld, float *__restrict__ G, float *__restrict__ V,
                                                    int step, int *__restrict__ counters, int sweep,
                                                    const int *__restrict__ stop)
{
    if (stop != nullptr && *stop != 0) return;
    // counters[s] = rotations of sweep s ; a sweep without rotations ends the decomposition
    if (sweep > 0 && counters[sweep - 1] == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_even = (n + 1) & ~1;
    const int k = blockIdx.x * 4 + wave;
    if (k >= n_even / 2) return;
    int p, q;
    rr_pair(n_even, step, k, p, q);
    if (q >= n) return;
    const int c = rotate_pair(G, V, ld, n, p, q, lane);
    if (lane == 0 && c) atomicAdd(&counters[sweep], 1);
}

__global__ void zero_counters_k(int *counters, int n) { for (int i = threadIdx.x; i < n; i += blockDim.x) counters[i] = 0; }

// w[i] = v_i . g_i - sigma (g_i = (M + sigma I) v_i) ; e[i], keep[i] from the built-in maps
__global__ __launch_bounds__(BLK) void eigvals_k(int n, int ld, const float *__restrict__ G, const float *__restrict__ V,
                                                const float *__restrict__ sc, int map_kind, float *__restrict__ w,
                                                float *__restrict__ e, const int *__restrict__ stop)
{
    if (stop != nullptr && *stop != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i >= n) return;
    float acc = 0.0f, nv = 0.0f;
    for (int r = lane; r < n; r += 64) {
        const float v = V[(size_t)i * ld + r];
        acc = fmaf(v, G[(size_t)i * ld + r], acc);
        nv = fmaf(v, v, nv);
    }
    acc = wave_sum(acc); nv = wave_sum(nv);
    if (lane == 0) {
        const float lam = (acc / nv - 1.0f) * sc[1];           // G = M / sigma + I (normalise_k)
        w[i] = lam;
        // map_kind 0: e > 0 -> e (cone_psd.rs:69-76); 1: e > 0 -> sqrt(e) (matbuild/mod.rs:231-238); < 0: host-supplied
        if (map_kind == 0) e[i] = lam > 0.0f ? lam : 0.0f;
        else if (map_kind == 1) e[i] = lam > 0.0f ? sqrtf(lam) : 0.0f;
    }
}

// packed(r,c) = sum_i e_i V(r,i) V(c,i), r <= c ; diag / scale (f64lapack.rs:96-107, 226-255).  One thread per
// output entry, consecutive threads -> consecutive r of one column (coalesced V reads).
__global__ __launch_bounds__(BLK) void rebuild_k(int n, int ld, const float *__restrict__ V, const float *__restrict__ e,
                                                int has_scale, float scale, float *__restrict__ packed,
                                                const int *__restrict__ stop)
{
    if (stop != nullptr && *stop != 0) return;
    const int c = blockIdx.y;
    for (int r = blockIdx.x * BLK + threadIdx.x; r <= c; r += gridDim.x * BLK) {
        float s = 0.0f;
        for (int i = 0; i < n; ++i) {
            const float ei = e[i];
            if (ei != 0.0f) s = fmaf(ei * V[(size_t)i * ld + c], V[(size_t)i * ld + r], s);
        }
        if (r == c && has_scale) s = s / scale;
        packed[(size_t)c * (c + 1) / 2 + r] = s;
    }
}

// ---------------------------------------------------------------------------------------------------
// engine (2): f32 GEMM on the matrix cores for the sign-function chain.  All operands are ld x ld with
// ld a multiple of 64 (zero padded: a block-diagonal [S 0; 0 0] stays block-diagonal under products).
//   C = alpha * A * B + beta * D      (A, B, C, D column-major ld x ld; D may alias C or be null)
// Workgroup = 256 threads = 4 waves, tile 64 x 64, each wave one 32 x 32 accumulator on
// v_mfma_f32_32x32x2_f32 (A operand: lane l holds A[i = l & 31][k = l >> 5]; B: B[k = l >> 5][j = l & 31];
// C/D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)).  K is staged through LDS in slabs of 16.
// ---------------------------------------------------------------------------------------------------
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int GT = 32;

// f32 GEMMs of the sign-function chain on v_mfma_f32_32x32x2_f32, two shapes (ld x ld, ld % 64 == 0, zero padded):
//   GEN = false:  C = alpha * X * Y^T + beta * D + gamma * I_n   with C symmetric by construction (X = Y, or X, Y
//                 bitwise-symmetric polynomials of one another): the Gram matrix S S^T and the step polynomial;
//   GEN = true :  C = alpha * Sym * Gen + beta * D + gamma * I_n  with Sym bitwise symmetric, Gen arbitrary:
//                 the update S <- q(S S^T) S.  This LEFT-multiplied form is the Newton-Schulz polar iteration, which
//                 damps antisymmetric round-off; S <- q(S S^T) S^T or S q(S S^T) doubles it every step.
// Workgroup = 8 waves, one 32 x 32 output tile; the waves split K eight ways and are combined through LDS, so a
// 512^3 product runs on 256 workgroups (every CU) with 8 short dependent load->MFMA chains each.  The matrices are L2-resident (1 MiB each at k = 500) and the
// operands go straight from global memory to the MFMA registers, coalesced:
//   operand b (lane l: k = l >> 5, column index l & 31) = Mem[k * ld + j0 + (l & 31)]   (a row of a symmetric matrix)
//   operand a: GEN = false the same from X;  GEN = true needs Gen(k, c) = GenMem[c * ld + k], contiguous in k, so each
//   wave stages a 32 x 8 slab through LDS (one float4 load along k per lane) and reads it back transposed.
// acc reg r of lane l is tile element (ti = (r & 3) + 8 (r >> 2) + 4 (l >> 5), tj = l & 31); it is stored at
// Cmem[(i0 + ti) * ld + j0 + tj] -- coalesced -- which is C(row j0 + tj, col i0 + ti): the true position for GEN
// (a indexes columns of C there), the mirrored one for the symmetric case.
// NW waves per workgroup split K NW ways (NW = 8: 512 threads; each wave walks ld / 8 of K in slabs of 8 = 4 MFMAs,
// the next slab's loads are issued before the current slab's MFMAs).
constexpr int GNW = 8;
constexpr int GSL = 8;              // K slab per wave per step

template <bool GEN>
__global__ __launch_bounds__(GNW * 64) void gemm_k(int n, int ld, float alpha, const float *__restrict__ X,
                                                   const float *__restrict__ Y, float beta, const float *D, float gamma,
                                                   float *C, const int *__restrict__ stop, size_t ws)
{
    if (stop != nullptr && *stop != 0) return;
    X += blockIdx.z * ws; Y += blockIdx.z * ws; C += blockIdx.z * ws;
    if (D) D += blockIdx.z * ws;
    __shared__ float red[GNW - 1][16][64];
    __shared__ float stg[GEN ? GNW : 1][GSL][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * GT, j0 = blockIdx.y * GT;
    const int kw = ld / GNW, kb = wave * kw;
    // GEN: X is the symmetric factor (b operand), Y the general one (a operand).  !GEN: a from X, b from Y.
    const float *pb = (GEN ? X : Y) + (size_t)(kb + (lane >> 5)) * ld + j0 + (lane & 31);
    // GEN staging: lane -> (column i' = lane >> 1, 4 consecutive k starting at (lane & 1) * 4): one float4 per slab
    const float *pa = GEN ? (Y + (size_t)(i0 + (lane >> 1)) * ld + kb + (lane & 1) * 4)
                          : (X + (size_t)(kb + (lane >> 5)) * ld + i0 + (lane & 31));
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    float bv[GSL / 2], av[GSL / 2];
    f32x4_t q;
    // prologue: slab 0
#pragma unroll
    for (int u = 0; u < GSL / 2; ++u) bv[u] = pb[(size_t)(2 * u) * ld];
    if constexpr (GEN) q = *reinterpret_cast<const f32x4_t *>(pa);
    else {
#pragma unroll
        for (int u = 0; u < GSL / 2; ++u) av[u] = pa[(size_t)(2 * u) * ld];
    }
    for (int k = 0; k < kw; k += GSL) {
        float bn[GSL / 2], an[GSL / 2];
        f32x4_t qn;
        const bool more = k + GSL < kw;
        if (more) {                                       // next slab's loads in flight during this slab's MFMAs
#pragma unroll
            for (int u = 0; u < GSL / 2; ++u) bn[u] = pb[(size_t)(k + GSL + 2 * u) * ld];
            if constexpr (GEN) qn = *reinterpret_cast<const f32x4_t *>(pa + k + GSL);
            else {
#pragma unroll
                for (int u = 0; u < GSL / 2; ++u) an[u] = pa[(size_t)(k + GSL + 2 * u) * ld];
            }
        }
        if constexpr (GEN) {
            const int ii = lane >> 1, kq = (lane & 1) * 4;
#pragma unroll
            for (int t = 0; t < 4; ++t) stg[wave][kq + t][ii] = q[t];
            __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS writes have landed
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < GSL / 2; ++u) av[u] = stg[wave][2 * u + (lane >> 5)][lane & 31];
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int u = 0; u < GSL / 2; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        if (more) {
#pragma unroll
            for (int u = 0; u < GSL / 2; ++u) bv[u] = bn[u];
            if constexpr (GEN) q = qn;
            else {
#pragma unroll
                for (int u = 0; u < GSL / 2; ++u) av[u] = an[u];
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
        const int tj = j0 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ti = i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float v = acc[r];
#pragma unroll
            for (int w = 0; w < GNW - 1; ++w) v += red[w][r][lane];
            v *= alpha;
            const size_t o = (size_t)ti * ld + tj;
            if (beta != 0.0f) v = fmaf(beta, D[o], v);
            if (ti == tj && ti < n) v += gamma;
            C[o] = v;
        }
    }
}

// The same two products with every operand load of a wave issued BEFORE its first MFMA (ld <= 512: a wave's K range is
// ld / 8 <= 64).  gemm_k above prefetches one slab of 8 ahead, i.e. a wave walks its K range in ld / 64 dependent L2
// round trips (~0.7 us each: the 6-7 us of an 8.7 us launch that are neither MFMA issue nor launch ramp).  Here:
//   * GEN: operand a is read ALONG k, as the general factor is stored: a(i, k) = Gen[(i0 + i) * ld + k], one dwordx4 per
//     lane per 8 k: lane (h = l >> 5, i = l & 31) gets k = 8 q + 4 h + t, t < 4.  An MFMA may pair ANY two k values as
//     long as a and b agree, so MFMA (q, t) takes k = 8 q + t on lanes 0-31 and k = 8 q + 4 + t on lanes 32-63: no LDS
//     transposition.  (Not for the X X^T products: X = S is symmetric only to round-off, and S^T S^T instead of the Gram
//     matrix S S^T loses the damping of antisymmetric round-off that makes the left-multiplied iteration stable.)
//   * every other operand: row k = 8 q + 4 h + t, coalesced along the tile (one dword per lane per MFMA);
//   * a wave's loads are in flight 8 slabs of 8 k (GEN: 12) ahead of its KW / 2 MFMAs.
// The summation order over k differs from gemm_k's (both are fixed, so results stay bitwise reproducible and the
// symmetric products stay bitwise symmetric: the mirrored tile swaps a and b, and the products commute).
// NW waves per workgroup split K NW ways.  tools/gemm_floor.hip (a dependent chain of launches that add one phase at a
// time, 512^3): empty launch 2.4 us, + flag and operand loads 2.9, + the MFMAs 7.4 with 8 waves (2 per SIMD) but 6.1
// with 4 waves (1 per SIMD, 64 MFMAs each), + the LDS reduction +0.1-0.3: the launch is bounded by MFMA ISSUE (about
// twice the 1.7 us that 64 instructions of 64 cycles per SIMD take at 2.4 GHz), not by operand latency.  Hence NW = 4,
// and SYM: a product whose result is symmetric (X Y^T with X = Y, or both symmetric polynomials of one matrix) computes
// only the 136 tiles on and below the diagonal of the 16 x 16 tile grid and writes each off-diagonal tile twice, the
// mirror image through an LDS transposition -- half the MFMAs, and bitwise symmetry by construction.
template <bool GEN, int KW, int NW, bool SYM>
__global__ __launch_bounds__(NW * 64) void gemm_pre_k(int n, int ld, float alpha, const float *__restrict__ X,
                                                      const float *__restrict__ Y, float beta, const float *D, float gamma,
                                                      float *C, const int *__restrict__ stop, size_t ws, int pitch, int dsym)
{
    static_assert(!(GEN && SYM), "the symmetric shortcut is for X Y^T products");
    // the stop flag is fetched now and looked at just before the first store: tested here, every launch of the chain
    // would start with an L2 round trip on which all of its operand loads wait
    const int halted = *stop;                   // never null: gemm() passes ctx().never_stop
    int bi = blockIdx.x, bj = blockIdx.y;
    X += blockIdx.z * ws; Y += blockIdx.z * ws; C += blockIdx.z * ws;
    if (D) D += blockIdx.z * ws;
    __shared__ float red[NW - 1][16][64];
    __shared__ float tr[SYM ? 32 : 1][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (SYM) {
        // blockIdx.x = t runs over the lower triangle: t = bi (bi + 1) / 2 + bj, bj <= bi
        const int t = blockIdx.x;
        bi = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
        while (bi * (bi + 1) / 2 > t) --bi;
        bj = t - bi * (bi + 1) / 2;
    }
    const int i0 = bi * GT, j0 = bj * GT;
    const int kb = wave * KW, h = lane >> 5, li = lane & 31;
    // GEN: X symmetric (b), Y general (a, along k).  !GEN: a(i, k) = X(i, k) = Xmem[k * ld + i], b(k, j) = Y(j, k).
    const float *pa = GEN ? Y + (size_t)(i0 + li) * pitch + kb + 4 * h : X + (size_t)(kb + 4 * h) * pitch + i0 + li;
    const float *pb = (GEN ? X : Y) + (size_t)(kb + 4 * h) * pitch + j0 + li;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    constexpr int NQ = KW / 8;
    // the loads run DEP slabs of 8 k ahead of the MFMAs: as many as fit in the 64 slots a wave has for loads in flight
    // (all of them up to ld = 256).  Issued all up front, the 65th stalls the wave until the first returns, and the
    // first MFMA waits behind the last load's ISSUE (tools/gemm_phases.hip).
    constexpr int DEP = (GEN ? 12 : 8) < NQ ? (GEN ? 12 : 8) : NQ;
    f32x4_t av[NQ];
    float bv[NQ][4];
    auto load = [&](const int q) {
        if constexpr (GEN) av[q] = *reinterpret_cast<const f32x4_t *>(pa + 8 * q);
        else {
#pragma unroll
            for (int t = 0; t < 4; ++t) av[q][t] = pa[(size_t)(8 * q + t) * pitch];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[q][t] = pb[(size_t)(8 * q + t) * pitch];
    };
#pragma unroll
    for (int q = 0; q < DEP; ++q) load(q);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // left alone the scheduler sinks the loads between the MFMAs to save registers (38 VGPRs, one or two loads in
    // flight): nothing may cross these points
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][t], bv[q][t], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (q + DEP < NQ) load(q + DEP);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (halted != 0) return;
    if (wave == 0) {
        const int tj = j0 + (lane & 31);
        float vv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ti = i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float v = acc[r];
#pragma unroll
            for (int w = 0; w < NW - 1; ++w) v += red[w][r][lane];
            v *= alpha;
            const size_t o = (size_t)ti * pitch + tj;
            if (beta != 0.0f) v = fmaf(beta, D[o], v);
            if (ti == tj && ti < n) v += gamma;
            if (!(SYM && dsym != 0 && bi == bj)) C[o] = v;
            vv[r] = v;
        }
        if constexpr (SYM) {
            if (bi == bj && dsym != 0) {
                // X != Y: the diagonal tile is stored as the average of itself and its transpose (see gemm_pre2_k)
#pragma unroll
                for (int r = 0; r < 16; ++r) tr[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = vv[r];
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    C[(size_t)(i0 + tl) * pitch + j0 + (lane & 31)] = 0.5f * (vv[r] + tr[lane & 31][tl]);
                }
            }
            if (bi != bj) {
                // the mirror tile: element (ti, tj) of this tile goes to Cmem[(j0 + tj) * ld + i0 + ti]; through LDS so
                // that the 32 lanes of a store walk along ti
#pragma unroll
                for (int r = 0; r < 16; ++r) tr[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = vv[r];
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tjl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);      // row of the mirror tile
                    C[(size_t)(j0 + tjl) * pitch + i0 + (lane & 31)] = tr[lane & 31][tjl];
                }
            }
        }
    }
}

// gemm_pre_k with a 32 x 64 block of the result per workgroup, for launches in which one 32 x 32 tile per workgroup would
// put more than one workgroup on a CU (the batched pair of 512^3 products: 512 tiles, or 272 for the symmetric shape, on
// 256 CUs).  tools/gemm_phases.hip (device-clock stamps inside the kernel) shows what bounds such a launch:
//   * operand loads: a wave may have 64 vector loads in flight, an L2 round trip under this load is ~1 us, so four
//     waves of dword loads (256 B per instruction) pull 64 KB per us into a CU, and the 128 KB of a tile take 1.4 us;
//   * the SECOND workgroup of a CU gets its operands at 4-6 us, not 2.8; its 1.8 us of MFMAs (64 x 64 cycles, issue
//     bound) and its store follow: 8.4 us for the two against 4.8 for a single tile.
// Here the two accumulators of a wave are the EVEN and the ODD columns of the 64-wide strip, so operand b is one dwordx2
// per lane (512 B per instruction, half as many instructions in flight for the same bytes) and the result leaves as
// dwordx2; operand a is shared (192 KB per workgroup instead of 2 x 128); the loads run DEP slabs of 8 k ahead of the
// MFMAs (as many as fit in the 64 slots) instead of all up front, so the first MFMA issues after one round trip and not
// after the wave has been allowed to issue its last load; and no CU holds more waves than SIMDs.  All four waves take
// part in the epilogue (four accumulator registers each).  Per element the order of every sum is that of gemm_pre_k:
// the results are bitwise the same.  SYM: the block covers tiles (bi, 2p) and (bi, 2p + 1) of the lower triangle; in
// the last block of an even row the second tile lies above the diagonal and is computed but not stored.
// MC (round 6, orders above 512): K = ld is walked in ld / (4 KW) chunks of 4 waves x KW, each chunk the one-chunk kernel's body
// (its own slab registers, its own first DEP slabs up front).  MC = false is the kernel as it was: the same bits.
template <bool GEN, int KW, bool SYM, bool MC = false>
__global__ __launch_bounds__(256) void gemm_pre2_k(int n, int ld, float alpha, const float *__restrict__ X,
                                                   const float *__restrict__ Y, float beta, const float *D, float gamma,
                                                   float *C, const int *__restrict__ stop, size_t ws, int pitch, int dsym)
{
    static_assert(!(GEN && SYM), "the symmetric shortcut is for X Y^T products");
    // the stop flag is fetched now and looked at just before the first store: tested here, every launch of the chain
    // would start with an L2 round trip on which all of its operand loads wait
    const int halted = *stop;                   // never null: gemm() passes ctx().never_stop
    constexpr int NW = 4;
    int bi, bj;
    const int item = blockIdx.z;
    if constexpr (SYM) {
        // blockIdx.x runs over the blocks of the lower triangle: tile row bi holds bi / 2 + 1 of them
        // (dealing the items out to the XCDs as polar_dual_k does measured 9.25 against 9.09 us here: not taken)
        int t = blockIdx.x;
        bi = 0;
        while (t >= bi / 2 + 1) { t -= bi / 2 + 1; ++bi; }
        bj = 2 * t;
    } else {
        bi = blockIdx.x;
        bj = 2 * blockIdx.y;
    }
    X += item * ws; Y += item * ws; C += item * ws;
    if (D) D += item * ws;
    __shared__ float red[NW][2][16][64];
    __shared__ float tr[SYM ? 32 : 1][65];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = bi * GT, j0 = bj * GT;
    const int kb = wave * KW, h = lane >> 5, li = lane & 31;
    const float *pa = GEN ? Y + (size_t)(i0 + li) * pitch + kb + 4 * h : X + (size_t)(kb + 4 * h) * pitch + i0 + li;
    const float *pb = (GEN ? X : Y) + (size_t)(kb + 4 * h) * pitch + j0 + 2 * li;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    constexpr int NQ = KW / 8;
    constexpr int DEP = (GEN ? 12 : 8) < NQ ? (GEN ? 12 : 8) : NQ;
    static_assert(!MC || !GEN, "the chunked form is the symmetric one");
    f32x16 acce, acco;
#pragma unroll
    for (int r = 0; r < 16; ++r) acce[r] = acco[r] = 0.0f;
    const int nch = MC ? ld / (4 * KW) : 1;
    for (int ch = 0; ch < nch; ++ch) {
        // (the slabs are this scope's: a chunk is the one-chunk kernel's body, its first DEP slabs a round trip that the
        // CU's other workgroup covers with its MFMAs -- carried across chunks the slab arrays went to scratch memory)
        f32x4_t av[NQ];
        f32x2_t bv[NQ][4];
        auto load = [&](const int q) {
            if constexpr (GEN) av[q] = *reinterpret_cast<const f32x4_t *>(pa + 8 * q);
            else {
#pragma unroll
                for (int t = 0; t < 4; ++t) av[q][t] = pa[(size_t)(8 * q + t) * pitch];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) bv[q][t] = *reinterpret_cast<const f32x2_t *>(pb + (size_t)(8 * q + t) * pitch);
        };
#pragma unroll
        for (int q = 0; q < DEP; ++q) load(q);
        // nothing may cross these points: left alone the scheduler sinks every load to its first use
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acce = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][t], bv[q][t][0], acce, 0, 0, 0);
                acco = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][t], bv[q][t][1], acco, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (q + DEP < NQ) load(q + DEP);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MC) { pa += (size_t)(4 * KW) * pitch; pb += (size_t)(4 * KW) * pitch; }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { red[wave][0][r][lane] = acce[r]; red[wave][1][r][lane] = acco[r]; }
    __syncthreads();
    if (halted != 0) return;
    // wave w finishes accumulator registers 4 w .. 4 w + 3 of both column sets: rows (r & 3) + 8 w + 4 h
    const int tj = j0 + 2 * li;                               // this lane's columns tj, tj + 1
    const bool colok = !SYM || tj < (bi + 1) * GT;
    // dsym (SYM, X != Y: the product of two commuting symmetric matrices): inside the DIAGONAL tile element (r, c) and
    // (c, r) are different sums -- the tile is stored as the average of itself and its transpose (a + b == b + a: bitwise
    // symmetric), after the block has been staged in tr
    const bool indiag = SYM && dsym != 0 && tj >= bi * GT && tj < (bi + 1) * GT;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = 4 * wave + rr;
        const int tl = rr + 8 * wave + 4 * h;
        const int ti = i0 + tl;
        float ve = red[0][0][r][lane], vo = red[0][1][r][lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) { ve += red[w][0][r][lane]; vo += red[w][1][r][lane]; }
        ve *= alpha; vo *= alpha;
        const size_t o = (size_t)ti * pitch + tj;
        if (beta != 0.0f) {
            const f32x2_t d = *reinterpret_cast<const f32x2_t *>(D + o);
            ve = fmaf(beta, d[0], ve); vo = fmaf(beta, d[1], vo);
        }
        if (ti < n) {
            if (ti == tj) ve += gamma;
            if (ti == tj + 1) vo += gamma;
        }
        if (colok && !indiag) {
            f32x2_t v2;
            v2[0] = ve; v2[1] = vo;
            *reinterpret_cast<f32x2_t *>(C + o) = v2;
        }
        if constexpr (SYM) { tr[tl][2 * li] = ve; tr[tl][2 * li + 1] = vo; }
    }
    if constexpr (SYM) {
        // the mirror image of the tiles strictly below the diagonal: element (ti, tj) also goes to Cmem[tj * ld + ti];
        // through LDS so that the 32 lanes of a store walk along ti.  Wave w: rows j0 + 16 w .. + 15 of the image.
        __syncthreads();
        if (indiag) {
            const int cd = bi * GT - j0;                      // the diagonal tile's first column inside the strip
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int tl = rr + 8 * wave + 4 * h;
                const int c0 = 2 * li - cd;                   // tile columns c0, c0 + 1
                f32x2_t v2;
                v2[0] = 0.5f * (tr[tl][2 * li] + tr[c0][cd + tl]);
                v2[1] = 0.5f * (tr[tl][2 * li + 1] + tr[c0 + 1][cd + tl]);
                *reinterpret_cast<f32x2_t *>(C + (size_t)(i0 + tl) * pitch + tj) = v2;
            }
        }
#pragma unroll
        for (int sI = 0; sI < 8; ++sI) {
            const int jj = 16 * wave + 2 * sI + h;
            if (bj + (jj >> 5) < bi) C[(size_t)(j0 + jj) * pitch + i0 + li] = tr[li][jj];
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Two symmetric products that SHARE their first factor, in one launch (round 5: the degree-7 polar step and the
// merged Newton-Schulz / M sign(M) pair):  O_p = alpha_p * A * B_p + beta_p * B_p + gamma_p * I_n,  p < nprod <= 2,
// A and B_p bitwise symmetric and commuting (polynomials of one symmetric matrix).  Only the lower triangle of tiles
// is computed, every tile is stored with its mirror image, diagonal tiles of a product with dsym_p != 0 (A != B_p)
// as the average of the tile and its transpose -- every result is bitwise symmetric again.
// A workgroup owns tile row bi's panel of A and up to NT of that row's tile-jobs (tile column j, product p), taken
// NT at a time from the list (j = 0, p = 0), (0, 1), (1, 0), ...: at ld = 512 the batch of two has 544 tile-jobs,
// NT = 3 puts them on 192 workgroups -- one per CU, which is what the chain's kernels need (a second workgroup on a CU
// halves both) -- of 3 x 64 MFMAs per wave.  Four waves split K four ways (KW each) and are combined
// through LDS; operand loads run DEP slabs of 8 k ahead of the MFMAs, as in gemm_pre_k, whose order of sums per
// element this kernel keeps.
struct DualArgs {
    int          n, ld, pitch, nprod;
    const float *A;
    const float *B[2];
    float       *O[2];
    float        alpha[2], beta[2], gamma[2];
    int          dsym[2];
    const int   *stop;
    size_t       ws;
    // pack != nullptr (nprod == 1): instead of O_0, the lower triangle of P = (M + O_0) / 2 leaves packed (diagonal /
    // scale) -- the last product of the projection -- and rx <- rx - 2 P rides along when rx != nullptr
    float       *pack;
    const float *M;
    float       *rx;
    ptrdiff_t    ps, rps;
    int          has_scale;
    float        scale;
    // workgroup -> (item, group of tile-jobs).  xpi > 0: a 1-D grid of 8 * spx blocks; block b runs on XCD b % 8 (observed,
    // not promised: only the speed depends on it); the XCDs are dealt out xpi per item and each takes a CONTIGUOUS chunk of
    // spx of its item's gp groups, i.e. a range of tile rows -- an XCD then pulls ONE item's operands through the fabric, and
    // of those mostly the panels of its rows (every operand of a launch was written by the previous launch on other XCDs:
    // the chain is bound by that traffic, NOTEBOOK 9.1b).  xpi == 0: blockIdx.x = group, blockIdx.z = item.
    // half > 0 (with xpi == 4): the item's four XCDs take two-dimensional blocks of the tile triangle instead (polar_dual_k)
    int          xpi, gp, spx, half;
};

__host__ __device__ inline int dual_groups(int nt, int nprod, int NT)
{
    int g = 0;
    for (int i = 0; i < nt; ++i) g += (nprod * (i + 1) + NT - 1) / NT;
    return g;
}

template <int KW, int NT, bool MC = false>
__global__ __launch_bounds__(256) void polar_dual_k(const DualArgs a)
{
    const int halted = *a.stop;                 // looked at before the first store (see gemm_pre_k)
    constexpr int NW = 4;
    // blockIdx.x -> (tile row bi, group gl of NT jobs inside it)
    int bi = 0, gl = blockIdx.x, item = blockIdx.z;
    int c0 = 0, njobs;                                  // first tile column of the row's jobs here, and how many jobs it has
    if (a.xpi == 4 && a.half > 0) {
        // TWO-DIMENSIONAL blocks (batch of two: four XCDs per item).  The lower triangle of the nt x nt tile grid, hf = nt / 2:
        //   block 0: rows [0, hf), columns <= row            block 3: rows [hf, nt), columns [hf, row]
        //   block 1: rows [hf, hf + hf / 2), columns [0, hf)    block 2: rows [hf + hf / 2, nt), columns [0, hf)
        // -- 36 / 32 / 32 / 36 tiles at nt = 16, and an XCD needs the panels of its rows and of its columns only: 8 or 12 of
        // the 16 of each operand instead of all of them on the XCD that holds the last rows of a row-wise deal
        const int xcd = blockIdx.x & 7;
        int slot = blockIdx.x >> 3;
        item = xcd >> 2;
        const int blk = xcd & 3, hf = a.half, nt = 2 * hf;
        const bool tri = blk == 0 || blk == 3;
        const int r0 = blk == 0 ? 0 : (blk == 2 ? hf + hf / 2 : hf), r1 = blk == 0 ? hf : (blk == 1 ? hf + hf / 2 : nt);
        c0 = blk == 3 ? hf : 0;
        bi = r0;
        for (;;) {
            if (bi >= r1) return;                           // (a block with fewer groups than the largest one)
            njobs = a.nprod * (tri ? bi - c0 + 1 : hf);
            const int gi = (njobs + NT - 1) / NT;
            if (slot < gi) break;
            slot -= gi; ++bi;
        }
        gl = slot;
    } else {
        if (a.xpi > 0) {
            const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
            item = xcd / a.xpi;
            gl = (xcd % a.xpi) * a.spx + slot;
            if (gl >= a.gp) return;                         // (the last chunk of an item may be short)
        }
        for (;;) {
            const int gi = (a.nprod * (bi + 1) + NT - 1) / NT;
            if (gl < gi) break;
            gl -= gi; ++bi;
        }
        njobs = a.nprod * (bi + 1);
    }
    const size_t zo = item * a.ws;
    __shared__ float red[NW][NT][16][64];
    __shared__ float tr[NT][32][33];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // uniform: the panel bases below stay in SGPRs
    const int h = lane >> 5, li = lane & 31;
    const int i0 = bi * GT, kb = wave * KW;
    const int pitch = a.pitch;
    int bj[NT], pr[NT];
    bool live[NT];
    // every load is (uniform base of the panel row) + (this lane's 32-bit byte offset): one global_load with an SGPR base and
    // no address arithmetic on the vector ALU, so that a load can issue in the shadow of the MFMA before it
    const char *pb[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int t = NT * gl + u;
        live[u] = t < njobs;
        const int tt = live[u] ? t : NT * gl;                      // a job past the row's end repeats the group's first (not stored)
        bj[u] = c0 + (a.nprod == 2 ? tt >> 1 : tt);
        pr[u] = a.nprod == 2 ? tt & 1 : 0;
        pb[u] = reinterpret_cast<const char *>(a.B[pr[u]] + zo + (size_t)kb * pitch + bj[u] * GT);
    }
    const char *pa = reinterpret_cast<const char *>(a.A + zo + (size_t)kb * pitch + i0);
    const unsigned lo = (unsigned)((4 * h * pitch + li) * (int)sizeof(float));
    const unsigned rowb = (unsigned)pitch * (unsigned)sizeof(float);
    constexpr int NQ = KW / 8;
    constexpr int PER = 4 * (NT + 1);                              // load instructions per slab of 8 k
    constexpr int DEP0 = 64 / PER < 1 ? 1 : 64 / PER;              // as many slabs ahead as fit the 64 loads a wave may have in flight
    constexpr int DEP = DEP0 < NQ ? DEP0 : NQ;
    float dv[NT][4];
    f32x16 acc[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
    }
    const int nch = MC ? a.ld / (4 * KW) : 1;
    for (int ch = 0; ch < nch; ++ch) {
        // (MC: a chunk of 4 KW rows of K is the one-chunk kernel's body with slab registers of its own -- see gemm_pre2_k)
        float av[NQ][4], bv[NQ][NT][4];
        // load number x of slab q: x < 4: A row 8 q + x; else panel (x - 4) / 4, row 8 q + (x - 4) % 4
        auto load1 = [&](const int q, const int x) {
            if (x < 4) av[q][x] = *reinterpret_cast<const float *>(pa + (size_t)((8 * q + x) * rowb) + lo);
            else {
                const int u = (x - 4) >> 2, t = (x - 4) & 3;
                bv[q][u][t] = *reinterpret_cast<const float *>(pb[u] + (size_t)((8 * q + t) * rowb) + lo);
            }
        };
#pragma unroll
        for (int q = 0; q < DEP; ++q) {
#pragma unroll
            for (int x = 0; x < PER; ++x) load1(q, x);
        }
        if (ch == 0) {
            // the beta * B_p term of the epilogue: this lane's four elements of every tile, fetched now instead of after the sums
#pragma unroll
            for (int u = 0; u < NT; ++u) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    dv[u][rr] = a.B[pr[u]][zo + (size_t)(i0 + rr + 8 * wave + 4 * h) * pitch + bj[u] * GT + li];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // slab q's 4 NT MFMAs with slab q + DEP's PER loads dealt out between them (an MFMA holds the pipe for 64 cycles: the
        // loads issue in its shadow; in blocks after the MFMAs they cost the wave a quarter of the MFMAs' time again)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            int x = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][t], bv[q][u][t], acc[u], 0, 0, 0);
                    if (q + DEP < NQ) {
                        const int upto = ((t * NT + u + 1) * PER + 4 * NT - 1) / (4 * NT);
                        for (; x < upto && x < PER; ++x) load1(q + DEP, x);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if constexpr (MC) {
            pa += (size_t)(4 * KW) * rowb;
#pragma unroll
            for (int u = 0; u < NT; ++u) pb[u] += (size_t)(4 * KW) * rowb;
        }
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][u][r][lane] = acc[u][r];
    }
    __syncthreads();
    if (halted != 0) return;
    // wave w finishes accumulator registers 4 w .. 4 w + 3 of every tile: rows tl = rr + 8 w + 4 h, column li
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int p = pr[u], j0 = bj[u] * GT;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = 4 * wave + rr, tl = rr + 8 * wave + 4 * h;
            float v = red[0][u][r][lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) v += red[w][u][r][lane];
            v *= a.alpha[p];
            const int ti = i0 + tl, tj = j0 + li;
            v = fmaf(a.beta[p], dv[u][rr], v);
            if (ti == tj && ti < a.n) v += a.gamma[p];
            tr[u][tl][li] = v;
        }
    }
    __syncthreads();
    if (a.pack == nullptr) {
        // the tile and its mirror image leave as 16-byte stores: thread t owns floats 4 (t % 8) .. of row t / 8 of either
        // (8 store instructions per tile per thread as dwords cost 1.5 us per tile of the launch: the stores' issue)
        typedef float f32x4s __attribute__((ext_vector_type(4)));
        const int er = tid >> 3, ec = (tid & 7) * 4;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            if (!live[u]) continue;
            const int p = pr[u], j0 = bj[u] * GT;
            const bool diag = bj[u] == bi;
            float *Om = a.O[p] + zo;
            f32x4s v, w;
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = tr[u][er][ec + k]; w[k] = tr[u][ec + k][er]; }
            if (diag && a.dsym[p] != 0) v = 0.5f * (v + w);
            *reinterpret_cast<f32x4s *>(Om + (size_t)(i0 + er) * pitch + j0 + ec) = v;
            if (!diag) *reinterpret_cast<f32x4s *>(Om + (size_t)(j0 + er) * pitch + i0 + ec) = w;
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        if (!live[u]) continue;
        const int p = pr[u], j0 = bj[u] * GT;
        const bool diag = bj[u] == bi;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int tl = rr + 8 * wave + 4 * h;
            float v = tr[u][tl][li];
            if (diag && a.dsym[p] != 0) v = 0.5f * (v + tr[u][li][tl]);
            {
                // element (row r = j0 + li, column c = i0 + tl) of the upper triangle by columns: r <= c
                const int rr_ = j0 + li, cc = i0 + tl;
                if (rr_ <= cc && cc < a.n) {
                    float pv = 0.5f * (a.M[zo + (size_t)cc * pitch + rr_] + v);
                    if (rr_ == cc && a.has_scale) pv = pv / a.scale;
                    const size_t o = (size_t)cc * (cc + 1) / 2 + rr_;
                    a.pack[(ptrdiff_t)item * a.ps + o] = pv;
                    if (a.rx != nullptr) {
                        float *rx = a.rx + (ptrdiff_t)item * a.rps;
                        rx[o] = rx[o] - 2.0f * pv;
                    }
                }
            }
        }
    }
}

// the grid of a launch and its workgroup -> (item, group) mapping (DualArgs::xpi)
static dim3 dual_grid(DualArgs &a, int nt, int NT, int nb)
{
    static const int xcd_map = getenv("THIP_PSD_XCD_MAP") ? atoi(getenv("THIP_PSD_XCD_MAP")) : 2;
    const int gp = dual_groups(nt, a.nprod, NT);
    a.gp = gp; a.xpi = 0; a.spx = 0; a.half = 0;
    if (xcd_map >= 2 && nb == 2 && nt % 4 == 0) {
        // two-dimensional blocks: slots per XCD = the groups of the largest of the four blocks
        const int hf = nt / 2;
        int gtri = 0;
        for (int r = 0; r < hf; ++r) gtri += (a.nprod * (r + 1) + NT - 1) / NT;
        const int grect = (hf / 2) * ((a.nprod * hf + NT - 1) / NT);
        a.xpi = 4; a.half = hf; a.spx = gtri > grect ? gtri : grect;
        return dim3((unsigned)(8 * a.spx), 1, 1);
    }
    if (xcd_map && (nb == 1 || nb == 2 || nb == 4 || nb == 8)) {
        a.xpi = 8 / nb;
        a.spx = (gp + a.xpi - 1) / a.xpi;
        return dim3((unsigned)(8 * a.spx), 1, 1);
    }
    return dim3((unsigned)gp, 1, (unsigned)nb);
}

// K chunks of an ld above 512 and the K range of a wave in each (np_of guarantees that the division is exact)
static inline int chunks_of(int ld) { return (ld + 511) / 512; }
static inline int kw_of(int ld) { return ld / (4 * chunks_of(ld)); }

template <int NT>
static int launch_dual(hipStream_t st, const DualArgs &a_in, int nb)
{
    const int nt = a_in.ld / GT;
    DualArgs a = a_in;
    const dim3 g = dual_grid(a, nt, NT, nb);
    if (a.ld > 512) {
        if constexpr (NT == 3) {
#define THIP_DUAL_MC(KW) hipLaunchKernelGGL((polar_dual_k<KW, 3, true>), g, dim3(256), 0, st, a)
            switch (kw_of(a.ld)) {
            case 80: THIP_DUAL_MC(80); break;
            case 96: THIP_DUAL_MC(96); break;
            case 112: THIP_DUAL_MC(112); break;
            case 128: THIP_DUAL_MC(128); break;
            default: return fail(THIP_E_INVALID, "dual: ld is not one np_of() gives", __FILE__, __LINE__);
            }
#undef THIP_DUAL_MC
            THIP_LAUNCH_CHECK();
            return 0;
        } else {
            return fail(THIP_E_INVALID, "dual: orders above 512 run three tile-jobs per workgroup", __FILE__, __LINE__);
        }
    }
#define THIP_DUAL(KW) hipLaunchKernelGGL((polar_dual_k<KW, NT>), g, dim3(256), 0, st, a)
    switch (a.ld / 4) {
    case 16: THIP_DUAL(16); break;
    case 32: THIP_DUAL(32); break;
    case 48: THIP_DUAL(48); break;
    case 64: THIP_DUAL(64); break;
    case 80: THIP_DUAL(80); break;
    case 96: THIP_DUAL(96); break;
    case 112: THIP_DUAL(112); break;
    default: THIP_DUAL(128); break;
    }
#undef THIP_DUAL
    THIP_LAUNCH_CHECK();
    return 0;
}


// (a variant of this kernel with the operands staged through LDS by global_load_lds_dwordx4 -- a ring of 8 slabs per wave, counted
// vmcnt waits, no barrier in the loop -- was built, passed the same bitwise tests and measured the same: 12.4 vs 12.3 us for
// three tile-jobs, 9.8 vs 9.5 for two.  Compile-time variants of it located the time: operand traffic 1.1 us, MFMAs 4.3 us,
// and 3-4 us that disappear when the launch stores nothing -- because then the NEXT launch finds its operands in its XCD's
// L2 instead of pulling them through the fabric.  Removed; NOTEBOOK 9.1b has the table.)
// the smallest NT whose grid fits one workgroup per CU
static int dual(hipStream_t st, DualArgs a, int nb)
{
    if (a.ld % 64 != 0 || (a.ld > 512 && (a.ld / 64) % chunks_of(a.ld) != 0)) return fail(THIP_E_INVALID, "dual: ld is not one np_of() gives", __FILE__, __LINE__);
    if (a.stop == nullptr) a.stop = ctx().never_stop;
    const int nt = a.ld / GT;
    static const int force_nt = getenv("THIP_PSD_DUAL_NT") ? atoi(getenv("THIP_PSD_DUAL_NT")) : 0;
    int NT = 1;
    // (one workgroup per CU: in all, and -- with the items dealt out to the XCDs, dual_grid -- on every XCD)
    auto too_many = [&](int NT_) {
        const int gp = dual_groups(nt, a.nprod, NT_);
        if (gp * nb > ctx().num_cu) return true;
        if (nb == 2 && nt % 4 == 0) {                  // (the two-dimensional blocks of dual_grid: the largest block's groups)
            const int hf = nt / 2;
            int gtri = 0;
            for (int r = 0; r < hf; ++r) gtri += (a.nprod * (r + 1) + NT_ - 1) / NT_;
            const int grect = (hf / 2) * ((a.nprod * hf + NT_ - 1) / NT_);
            return (gtri > grect ? gtri : grect) > ctx().num_cu / 8;
        }
        if (nb == 1 || nb == 2 || nb == 4 || nb == 8) return (gp + 8 / nb - 1) / (8 / nb) > ctx().num_cu / 8;
        return false;
    };
    while (NT < 3 && too_many(NT)) ++NT;
    if (force_nt >= 1 && force_nt <= 3) NT = force_nt;
    if (a.ld > 512) NT = 3;
    return NT == 1 ? launch_dual<1>(st, a, nb) : NT == 2 ? launch_dual<2>(st, a, nb) : launch_dual<3>(st, a, nb);
}

// S = M / ||M||_F; the exact zero matrix stays zero.  ||M||_F from unpack_k's block 2-norms (part, np of them), summed by
// EVERY workgroup for itself (np <= 512 floats from L2: cheaper than a launch that does it once)
__global__ __launch_bounds__(BLK) void scale_by_fro_k(size_t tot, const float *__restrict__ M, const float *__restrict__ part, int np,
                                                     float *__restrict__ S, const int *__restrict__ stop, size_t ws, float mul)
{
    if (stop != nullptr && *stop != 0) return;
    M += blockIdx.z * ws; part += blockIdx.z * ws; S += blockIdx.z * ws;
    __shared__ double shd[16];
    double acc = 0.0;
    for (int k = threadIdx.x; k < np; k += BLK) acc += (double)part[k] * (double)part[k];
    acc = block_sum_d(acc, shd);
    // a division per element: 1 / f overflows for a subnormal norm, and 0 * inf would poison the iterate
    const float f = (float)sqrt(acc);
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < tot; i += (size_t)gridDim.x * BLK) S[i] = f > 0.0f ? mul * (M[i] / f) : 0.0f;
}

// packed(r,c) = (M + MS)(r,c) / 2 symmetrised, diag / scale
__global__ void pack_half_k(int n, int ld, const float *__restrict__ M, const float *__restrict__ MS, int has_scale,
                            float scale, float *__restrict__ packed, const int *__restrict__ stop, size_t ws, ptrdiff_t ps,
                            float *__restrict__ rx, ptrdiff_t rps, int pitch)
{
    if (stop != nullptr && *stop != 0) return;
    M += blockIdx.z * ws; MS += blockIdx.z * ws; packed += (ptrdiff_t)blockIdx.z * ps;
    if (rx != nullptr) rx += (ptrdiff_t)blockIdx.z * rps;       // the fused loop's reflection rx <- rx - 2 x rides along
    const int c = blockIdx.y;
    for (int r = blockIdx.x * BLK + threadIdx.x; r <= c; r += gridDim.x * BLK) {
        const size_t o1 = (size_t)c * pitch + r, o2 = (size_t)r * pitch + c;
        float v = 0.5f * (M[o1] + 0.5f * (MS[o1] + MS[o2]));
        if (r == c && has_scale) v = v / scale;
        const size_t o = (size_t)c * (c + 1) / 2 + r;
        packed[o] = v;
        if (rx != nullptr) rx[o] = rx[o] - 2.0f * v;
    }
}

// The whole PSD projection of a matrix of order n <= 64 in ONE workgroup, one launch for a batch (blockIdx.x = item):
// unpack, Frobenius norm, a 45-product quintic polar chain (the schedule rounds 1-4 ran as launches) and the symmetrised pack, all operands in
// LDS.  At these orders the chain of launches is nothing but launch boundaries (45 x 4.4 us = 0.2 ms at any n <= 128;
// the reference's own SDP example, partitioning_sdp, has order 48), while the products themselves are (n / 2) MFMAs per
// wave: four waves, one 32 x 32 quadrant of the result each (one MFMA wave per SIMD), K runs over the
// n columns that are not padding.  Matrices are row-major with a pitch of 65 words: the a operand (a row per lane) and
// the b operand of either shape (a row per lane for X X^T, consecutive words for Sym * Gen) are conflict-free.
// Y = S S^T and T = c Y Y + b Y + a I come out bitwise symmetric (the mirrored element sums the same products in the
// same order), Z = T S is the left-multiplied update.
constexpr int PSN = 64, PSP = 65;
typedef float ps_mat[PSN][PSP];
// five operands in LDS: 83 200 bytes static -- more than the 64 KB of older parts.  This library is built for gfx950 only
// (160 KB of LDS per CU; Makefile: --offload-arch=gfx950), which is what makes the one-workgroup form the default engine.
static_assert(5 * sizeof(ps_mat) <= 160 * 1024, "polar_small_k keeps five 64 x 65 operands in LDS");

// C = alpha * (FORM 0: A B^T, FORM 1: A B) + beta * D + gamma * I_n on this wave's quadrant; nk = number of MFMA steps
template <int FORM, int NK>
__device__ __forceinline__ void ps_gemm(ps_mat &C, const ps_mat &A, const ps_mat &B, const ps_mat *D, float alpha, float beta,
                                        float gamma, int n, int qi, int qj, int h, int li, bool live)
{
    if (live) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const float *pa = &A[32 * qi + li][h];
        const float *pb = FORM == 0 ? &B[32 * qj + li][h] : &B[h][32 * qj + li];
        constexpr int SB = FORM == 0 ? 2 : 2 * PSP;
        // NK MFMA steps, a multiple of 4 (columns n .. 63 are zero padding inside the arrays), unrolled: the LDS reads
        // of later steps are in flight under the MFMAs of earlier ones
        float av[NK], bv[NK];
#pragma unroll
        for (int u = 0; u < NK; ++u) { av[u] = pa[2 * u]; bv[u] = pb[SB * u]; }
#pragma unroll
        for (int u = 0; u < NK; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 32 * qi + (r & 3) + 8 * (r >> 2) + 4 * h, j = 32 * qj + li;
            float v = alpha * acc[r];
            if (D != nullptr) v = fmaf(beta, (*D)[i][j], v);
            if (i == j && i < n) v += gamma;
            C[i][j] = v;
        }
    }
    __syncthreads();
}

template <int NK>
__global__ __launch_bounds__(256) void polar_small_k(int n, float *__restrict__ packed, int has_scale, float scale,
                                                     const int *__restrict__ stop, ptrdiff_t ps,
                                                     const int64_t *__restrict__ offs, float *__restrict__ rx, ptrdiff_t rps)
{
    if (stop != nullptr && *stop != 0) return;
    // blockIdx.x: which matrix of a table of offsets (cones of one order), blockIdx.y: which item of its batch
    packed += (offs != nullptr ? (ptrdiff_t)offs[blockIdx.x] : 0) + (ptrdiff_t)blockIdx.y * ps;
    // rx != nullptr (the fused loop): the reflection rx <- rx - 2 x of the projected rows rides in the pack
    if (rx != nullptr) rx += (offs != nullptr ? (ptrdiff_t)offs[blockIdx.x] : 0) + (ptrdiff_t)blockIdx.y * rps;
    __shared__ ps_mat M, S0, S1, Y, T;
    __shared__ double shd[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = wave >> 1, qj = wave & 1, h = lane >> 5, li = lane & 31;
    // a quadrant that is all padding is neither computed nor read (its operands' padding stays zero from here on)
    const bool live = 32 * qi < n && 32 * qj < n;
    double acc = 0.0;
    for (int e = tid; e < PSN * PSN; e += 256) {
        const int r = e & 63, c = e >> 6;
        float v = 0.0f;
        if (r < n && c < n) {
            const int lo = r < c ? r : c, hi = r < c ? c : r;
            v = packed[(size_t)hi * (hi + 1) / 2 + lo];
            if (r == c && has_scale) v *= scale;
            acc += (double)v * (double)v;
        }
        M[r][c] = v;
        S1[r][c] = 0.0f; Y[r][c] = 0.0f; T[r][c] = 0.0f;
    }
    acc = block_sum_d(acc, shd);
    const float fro = (float)sqrt(acc);
    // a DIVISION per element, not a multiplication by 1 / fro: the reciprocal of a subnormal norm is infinite (a slack
    // block on its way to zero gets there), and 0 * inf poisons the iterate.  The exact zero matrix stays zero.
    for (int e = tid; e < PSN * PSN; e += 256) S0[e >> 6][e & 63] = fro > 0.0f ? M[e >> 6][e & 63] / fro : 0.0f;
    __syncthreads();
    ps_mat *S = &S0, *Z = &S1;
    // 11 lifting quintics (the first on 1.7 x; band [0.3, 1.7], gain 3.94), 3 minimax quintics, 1 Newton-Schulz
    const float LIFT[3] = { 4.02942496f, -3.82532605f, 0.95951948f };
    const float TAILC[3][3] = {
        { 2.647997920f, -1.945904487f, 0.440483961f },
        { 1.967564378f, -1.351306898f, 0.386705679f },
        { 1.884943743f, -1.269148602f, 0.384197480f },
    };
    for (int it = 0; it < 14; ++it) {
        float a, b, c;
        if (it == 0) { a = LIFT[0] * 1.7f; b = LIFT[1] * 4.913f; c = LIFT[2] * 14.19857f; }
        else if (it < 11) { a = LIFT[0]; b = LIFT[1]; c = LIFT[2]; }
        else { a = TAILC[it - 11][0]; b = TAILC[it - 11][1]; c = TAILC[it - 11][2]; }
        ps_gemm<0, NK>(Y, *S, *S, nullptr, 1.0f, 0.0f, 0.0f, n, qi, qj, h, li, live);
        ps_gemm<1, NK>(T, Y, Y, &Y, c, b, a, n, qi, qj, h, li, live);
        ps_gemm<1, NK>(*Z, T, *S, nullptr, 1.0f, 0.0f, 0.0f, n, qi, qj, h, li, live);
        ps_mat *t = S; S = Z; Z = t;
    }
    ps_gemm<0, NK>(T, *S, *S, nullptr, -0.5f, 0.0f, 1.5f, n, qi, qj, h, li, live);
    ps_gemm<1, NK>(*Z, T, *S, nullptr, 1.0f, 0.0f, 0.0f, n, qi, qj, h, li, live);
    { ps_mat *t = S; S = Z; Z = t; }
    ps_gemm<1, NK>(*Z, M, *S, nullptr, 1.0f, 0.0f, 0.0f, n, qi, qj, h, li, live);       // M sign(M)
    for (int e = tid; e < PSN * PSN; e += 256) {
        const int r = e & 63, c = e >> 6;
        if (r <= c && c < n) {
            float v = 0.5f * (M[r][c] + 0.5f * ((*Z)[r][c] + (*Z)[c][r]));
            if (r == c && has_scale) v = v / scale;
            const size_t o = (size_t)c * (c + 1) / 2 + r;
            packed[o] = v;
            if (rx != nullptr) rx[o] = rx[o] - 2.0f * v;
        }
    }
}

static int g_force_kernel = 0;     // thip_test_gemm_chain: 1 = one tile per workgroup, 2 = 32 x 64 blocks, 0 = by tile count
// gen == false: C = alpha X Y^T + beta D + gamma I (symmetric result); gen == true: C = alpha X Y + ... (X symmetric)
// dsym (gen == false only): X != Y, both bitwise symmetric and commuting -- diagonal tiles are stored averaged with
// their transpose, so that the result is bitwise symmetric like an X X^T product's
int gemm(hipStream_t st, bool gen, int n, int ld, float alpha, const float *X, const float *Y, float beta, const float *D,
         float gamma, float *C, const int *stop, int nb = 1, size_t ws = 0, int pitch = 0, int dsym = 0)
{
    if (pitch == 0) pitch = ld;                 // rows of the operands are `pitch` floats apart; ld = the extent of every index
    dim3 g(ld / GT, ld / GT, nb);
    if (stop == nullptr) stop = ctx().never_stop;
    // ld <= 512: gemm_pre_k (loads up front, 4 waves, symmetric results from the lower triangle of tiles), or its
    // two-tile form when there are more tiles than CUs; larger orders, or THIP_GEMM_MODE=0: gemm_k (slab prefetch, 8
    // waves, any ld).  THIP_GEMM_MODE=2: one tile per workgroup whatever the count.
    static const int mode = getenv("THIP_GEMM_MODE") ? atoi(getenv("THIP_GEMM_MODE")) : 3;
    const int nt = ld / GT;
    const int tiles = (gen ? nt * nt : nt * (nt + 1) / 2) * nb;
    const bool pairs = g_force_kernel != 0 ? g_force_kernel == 2 && nt % 2 == 0 : mode >= 3 && tiles > ctx().num_cu && nt % 2 == 0;
    int npair = 0;
    for (int bi = 0; bi < nt; ++bi) npair += bi / 2 + 1;
#define THIP_GEMM_PRE4(KW)                                                                                                  \
    do {                                                                                                                    \
        if (pairs && gen) hipLaunchKernelGGL((gemm_pre2_k<true, KW, false>), dim3(nt, nt / 2, nb), dim3(256), 0, st, n, ld, alpha, X, Y, beta, D, gamma, C, stop, ws, pitch, dsym); \
        else if (pairs)   hipLaunchKernelGGL((gemm_pre2_k<false, KW, true>), dim3(npair, 1, nb), dim3(256), 0, st, n, ld, alpha, X, Y, beta, D, gamma, C, stop, ws, pitch, dsym); \
        else if (gen) hipLaunchKernelGGL((gemm_pre_k<true, KW, 4, false>), g, dim3(256), 0, st, n, ld, alpha, X, Y, beta, D, gamma, C, stop, ws, pitch, dsym);  \
        else     hipLaunchKernelGGL((gemm_pre_k<false, KW, 4, true>), dim3(nt * (nt + 1) / 2, 1, nb), dim3(256), 0, st, n, ld, alpha, X, Y, beta, D, gamma, C, stop, ws, pitch, dsym); \
    } while (0)
    // (a dsym product exists in the block kernels only: THIP_GEMM_MODE < 2 -- a debug setting -- does not apply to it)
    if ((mode >= 2 || dsym != 0) && ld <= 512) {
        switch (ld / 4) {
        case 16: THIP_GEMM_PRE4(16); break;
        case 32: THIP_GEMM_PRE4(32); break;
        case 48: THIP_GEMM_PRE4(48); break;
        case 64: THIP_GEMM_PRE4(64); break;
        case 80: THIP_GEMM_PRE4(80); break;
        case 96: THIP_GEMM_PRE4(96); break;
        case 112: THIP_GEMM_PRE4(112); break;
        default: THIP_GEMM_PRE4(128); break;
        }
    }
    else if ((mode >= 2 || dsym != 0) && !gen && nt % 2 == 0 && (ld / 64) % chunks_of(ld) == 0 && kw_of(ld) >= 80 && kw_of(ld) % 16 == 0) {
        // orders above 512, symmetric result: the 32 x 64 block kernel walking K in chunks
#define THIP_GEMM_MC(KW) hipLaunchKernelGGL((gemm_pre2_k<false, KW, true, true>), dim3(npair, 1, nb), dim3(256), 0, st, n, ld, alpha, X, Y, beta, D, gamma, C, stop, ws, pitch, dsym)
        switch (kw_of(ld)) {
        case 80: THIP_GEMM_MC(80); break;
        case 96: THIP_GEMM_MC(96); break;
        case 112: THIP_GEMM_MC(112); break;
        default: THIP_GEMM_MC(128); break;
        }
#undef THIP_GEMM_MC
    }
    else if (pitch != ld || dsym != 0) return fail(THIP_E_INVALID, "gemm: a padded pitch / dsym needs the block kernels (THIP_GEMM_MODE >= 2, a symmetric product)", __FILE__, __LINE__);
    else if (gen) hipLaunchKernelGGL(gemm_k<true>, g, dim3(GNW * 64), 0, st, n, ld, alpha, X, Y, beta, D, gamma, C, stop, ws);
    else     hipLaunchKernelGGL(gemm_k<false>, g, dim3(GNW * 64), 0, st, n, ld, alpha, X, Y, beta, D, gamma, C, stop, ws);
#undef THIP_GEMM_PRE4
    THIP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// engine (3): Householder tridiagonalisation + implicit QL -- the general eigen-decomposition for n > 32
// (BASELINE.json north_star: "symmetric eigendecomposition (Householder tridiag + QR)"; the routine the reference
// calls is dsyevr / cusolver syevdx, f64lapack.rs:78-108, f32cuda.rs:253-263).  O(n) dependent steps instead of the
// O(n * sweeps) of the Jacobi engine:
//   1. Q^T M Q = T, n - 2 reflectors, ONE launch per reflector (tri_step_k: applies the symmetric rank-2 update of the
//      previous reflector, A -= v w^T + w v^T on the full trailing square so that a "row" stays a contiguous column, to the
//      columns it owns while it forms the next reflector v_j and p = tau A v from them, one wave per column);
//   2. Z = Q formed column by column (form_q_k: one wave per column applies all reflectors, no global step);
//   3. d, e visit the host: implicit-shift QL in f64 (the O(n^2) scalar recurrence of tql2) which only RECORDS its
//      Givens rotations, one (c, s) pair each, grouped in sweeps of adjacent pairs;
//   4. rot_apply_k replays the record on Z: rows are independent, one lane per row, the 64 rows of a workgroup live
//      in LDS ([column][row], conflict-free), the entry carried from one rotation of a sweep to the next in a register;
//   5. the rebuild V diag(e) V^T is a GEMM on the matrix cores (gemm(false): X Y^T with X = V diag(e)).
// ---------------------------------------------------------------------------------------------------
constexpr int TRI_MAXN = 2048;

__global__ __launch_bounds__(BLK) void tri_upd_k(int n, int ld, int j, float *__restrict__ G, const float *__restrict__ Vh,
                                                const float *__restrict__ p, const float *__restrict__ tau)
{
    __shared__ float vsh[TRI_MAXN];
    __shared__ float wsh[TRI_MAXN];
    __shared__ float red[16];
    const int tid = threadIdx.x;
    const int L = n - j - 1;
    const float t = tau[j];
    if (t == 0.0f) return;                           // H = I
    float acc = 0.0f;
    for (int i = tid; i < L; i += BLK) {
        const float vi = Vh[(size_t)j * ld + j + 1 + i], pi = p[i];
        vsh[i] = vi; wsh[i] = pi;
        acc = fmaf(pi, vi, acc);
    }
    acc = block_sum(acc, red);
    const float k = -0.5f * t * acc;
    for (int i = tid; i < L; i += BLK) wsh[i] = fmaf(k, vsh[i], wsh[i]);
    __syncthreads();
    for (int cc = 0; cc < 4; ++cc) {
        const int c = blockIdx.x * 4 + cc;
        if (c >= L) break;
        float *col = G + (size_t)(j + 1 + c) * ld + j + 1;
        const float wc = wsh[c], vc = vsh[c];
        for (int r = tid; r < L; r += BLK) col[r] = col[r] - (vsh[r] * wc + wsh[r] * vc);
    }
}

// One launch per reflector: step j applies the rank-2 update of step j - 1 to the columns it owns WHILE it forms
// p_j = tau_j A v_j from them.  Every workgroup first rebuilds, redundantly, what it needs of step j - 1 (w_{j-1} from
// p_{j-1} and v_{j-1}) and the updated column j (-> d_j, the reflector v_j, tau_j, e_j): O(n) work per workgroup against
// a launch boundary saved per step.  Column c's update is local to the workgroup that owns c, and p_j[c] = column c . v_j
// by symmetry, so nothing crosses workgroups inside the launch.  Step 0 has no pending update (first != 0).
// Round 3: everything a workgroup reads from global memory -- p_{j-1}, v_{j-1}, column j and the wave's own column -- is
// requested at ENTRY, into registers (NQ = ceil(n / 256) resp. ceil(n / 64) values per thread); the kernel used to walk
// three dependent L2 round trips (p and v -> w; column j -> v_j; own column -> p_j) and is one round trip plus three block
// reductions now.
template <int NQ>       // n <= 256 * NQ
__global__ __launch_bounds__(BLK) void tri_step_k(int n, int ld, int j, int first, float *__restrict__ G, float *__restrict__ Vh,
                                                 const float *__restrict__ p_prev, float *__restrict__ p_out,
                                                 float *__restrict__ d, float *__restrict__ e, float *__restrict__ tau)
{
    __shared__ float vp[256 * NQ];      // v_{j-1}, indices j .. n-1  (local 0 .. Lp-1)
    __shared__ float wp[256 * NQ];      // w_{j-1}
    __shared__ float vsh[256 * NQ];     // v_j, indices j+1 .. n-1   (local 0 .. L-1)
    __shared__ float red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Lp = n - j, L = n - j - 1;
    // ---- all global reads up front
    const float *cj = G + (size_t)j * ld + j;
    float r_v[NQ], r_p[NQ], r_x[NQ], r_a[4 * NQ];
    float tprev = 0.0f;
    if (!first) tprev = tau[j - 1];
#pragma unroll
    for (int m = 0; m < NQ; ++m) {
        const int i = tid + BLK * m;
        r_v[m] = (!first && i < Lp) ? Vh[(size_t)(j - 1) * ld + j + i] : 0.0f;
        r_p[m] = (!first && i < Lp) ? p_prev[i] : 0.0f;
        r_x[m] = i < Lp ? cj[i] : 0.0f;
    }
    const int c = blockIdx.x * 4 + wave;
    float *col = G + (size_t)(j + 1 + c) * ld + j + 1;
    if (c < L) {
#pragma unroll
        for (int m = 0; m < 4 * NQ; ++m) {
            const int r = lane + 64 * m;
            r_a[m] = r < L ? col[r] : 0.0f;
        }
    }
    // ---- w_{j-1}
    if (!first) {
        float acc = 0.0f;
#pragma unroll
        for (int m = 0; m < NQ; ++m) {
            const int i = tid + BLK * m;
            if (i < Lp) { vp[i] = r_v[m]; wp[i] = r_p[m]; acc = fmaf(r_p[m], r_v[m], acc); }
        }
        acc = block_sum_dpp(acc, red);
        const float kk = -0.5f * tprev * acc;
#pragma unroll
        for (int m = 0; m < NQ; ++m) {
            const int i = tid + BLK * m;
            if (i < Lp) wp[i] = fmaf(kk, r_v[m], r_p[m]);
        }
        __syncthreads();
    }
    const bool upd = !first && tprev != 0.0f;
    // column j after the pending update (local row r <-> global row j + r); x = its rows below the diagonal
    const float w0 = upd ? wp[0] : 0.0f, v0 = upd ? vp[0] : 0.0f;
    float ss = 0.0f;
#pragma unroll
    for (int m = 0; m < NQ; ++m) {
        const int i = tid + BLK * m;
        if (i < Lp) {
            float x = r_x[m];
            if (upd) x -= vp[i] * w0 + wp[i] * v0;
            if (i >= 1) { vsh[i - 1] = x; if (i >= 2) ss = fmaf(x, x, ss); }
            else if (blockIdx.x == 0) d[j] = x;
        }
    }
    ss = block_sum_dpp(ss, red);
    __syncthreads();
    const float alpha = vsh[0];
    const float xnorm = sqrtf(ss);
    float t = 0.0f, beta = alpha, scale = 0.0f;
    if (xnorm != 0.0f) {
        beta = -copysignf(hypotf(alpha, xnorm), alpha);
        t = (beta - alpha) / beta;
        scale = 1.0f / (alpha - beta);
    }
    __syncthreads();
    for (int i = tid; i < L; i += BLK) vsh[i] = i == 0 ? 1.0f : vsh[i] * scale;
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int i = tid; i < L; i += BLK) Vh[(size_t)j * ld + j + 1 + i] = vsh[i];
        if (tid == 0) { e[j] = beta; tau[j] = t; }
    }
    // own columns: one wave per column c (trailing block of step j: global column j + 1 + c)
    if (c < L) {
        const float wc = upd ? wp[c + 1] : 0.0f, vc = upd ? vp[c + 1] : 0.0f;
        float sacc = 0.0f;
#pragma unroll
        for (int m = 0; m < 4 * NQ; ++m) {
            const int r = lane + 64 * m;
            if (r < L) {
                float a = r_a[m];
                if (upd) { a -= vp[r + 1] * wc + wp[r + 1] * vc; col[r] = a; }
                sacc = fmaf(a, vsh[r], sacc);
            }
        }
        sacc = wave_sum_dpp(sacc);
        if (lane == 0) p_out[c] = t * sacc;
    }
}

// ---------------------------------------------------------------------------------------------------
// The same reduction as ONE persistent launch (round 3).  A launch per reflector costs 4.6 us at k = 500 (the kernel is
// three dependent L2 round trips and three block reductions; 2.3 ms for 498 of them).  Here W workgroups keep the columns
// they own (column c belongs to workgroup c mod W) in LDS for the whole reduction, and the only thing that crosses
// workgroups per reflector is ONE all-gather: the entries of p_j = tau_j A v_j a workgroup formed from its columns, and
// column j + 1 from its owner -- "stale" by the rank-2 update of step j, which every workgroup applies to its copy itself
// (it has v_j and forms w_j like everybody else), exactly as tri_step_k does.  Transport: 8-byte {value, tag} granules
// written with one write-through (sc1) store each and polled with sc1 loads (MI355X_MICROARCH.md, "allgather" row of the
// price list); tag = step + 1, two buffers alternate by step parity (a workgroup can be at most one step ahead of the
// slowest: to publish step j + 1 it must have gathered all of step j).  Every spin is bounded: a workgroup that gives up
// raises *errflag and leaves, the others follow, and the host redoes the reduction with one launch per reflector.
// ---------------------------------------------------------------------------------------------------
constexpr int TP_THREADS = 1024;     // 16 waves: one column per wave in the pass over the workgroup's columns (n <= 512)
constexpr int TP_SPIN_MAX = 300000;
constexpr unsigned TP_DONE = 0xD0E5u;             // *errflag after a completed persistent reduction
constexpr int TP_GM = 2048 / TP_THREADS;          // granules of one array a thread may have to fetch (n <= 2048)

__device__ __forceinline__ unsigned long long tp_pack(float v, unsigned tag)
{
    return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}

// gathers rows [r0, n) of one or two granule arrays into LDS; false if a granule never arrived
__device__ __forceinline__ bool tp_gather(const unsigned long long *__restrict__ g0, float *__restrict__ dst0,
                                          const unsigned long long *__restrict__ g1, float *__restrict__ dst1, int r0, int n,
                                          unsigned tag, unsigned *errflag)
{
    const int tid = threadIdx.x;
    unsigned pend = 0;
#pragma unroll
    for (int m = 0; m < TP_GM; ++m) {
        const int r = r0 + tid + TP_THREADS * m;
        if (r < n) pend |= (1u << m) | (g1 != nullptr ? (1u << (m + TP_GM)) : 0u);
    }
    int spins = 0;
    bool ok = true;
    while (pend) {
        unsigned long long v[2 * TP_GM];
#pragma unroll
        for (int m = 0; m < 2 * TP_GM; ++m) {
            const int r = r0 + tid + TP_THREADS * (m % TP_GM);
            if ((pend >> m) & 1u) v[m] = __hip_atomic_load((m < TP_GM ? g0 : g1) + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int m = 0; m < 2 * TP_GM; ++m) {
            const int r = r0 + tid + TP_THREADS * (m % TP_GM);
            if (((pend >> m) & 1u) && (unsigned)(v[m] >> 32) == tag) {
                (m < TP_GM ? dst0 : dst1)[r] = __uint_as_float((unsigned)v[m]);
                pend &= ~(1u << m);
            }
        }
        if (pend) {
            ++spins;
            if (spins > TP_SPIN_MAX || ((spins & 1023) == 0 && __hip_atomic_load(errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return ok;
}

// XCD_LOCAL: all W workgroups run on ONE XCD (the launch has 8 W + 64 of them; those that find HW_REG_XCC_ID == 0 draw a
// ticket, the first W tickets are the roles, everybody else leaves at once), so that the granules are exchanged through the
// L2 they share: plain 8-byte stores (the line stays in that L2) and sc1 loads (served by it), no trip over the fabric.
template <bool XCD_LOCAL>
__device__ __forceinline__ void tp_store(unsigned long long *p, unsigned long long v)
{
    if constexpr (XCD_LOCAL) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool XCD_LOCAL>
__global__ __launch_bounds__(TP_THREADS) void tri_persist_k(int n, int ld, int W, int slots, float *__restrict__ G,
                                                           float *__restrict__ Vh, float *__restrict__ d, float *__restrict__ e,
                                                           float *__restrict__ tau, unsigned long long *__restrict__ gran,
                                                           unsigned *__restrict__ errflag, unsigned *__restrict__ ticket,
                                                           unsigned long long *__restrict__ stamps)
{
    extern __shared__ float tp_sh[];
    __shared__ int role;
    if constexpr (XCD_LOCAL) {
        if (threadIdx.x == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;        // HW_REG_XCC_ID, bits 3:0
            role = xcc == 0u ? (int)atomicAdd(ticket, 1u) : -1;
        }
        __syncthreads();
        if (role < 0 || role >= W) return;
    }
    float *cols = tp_sh;                                  // [slots][n]: column s * W + wg, all rows
    float *va = cols + (size_t)slots * n, *vb = va + n;   // v_{j-1} and v_j (they swap), indexed by global row, zero above
    float *wp = vb + n;                                   // w_{j-1}
    float *ps = wp + n;                                   // p_{n-3} as gathered for the 2 x 2 tail
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = XCD_LOCAL ? role : (int)blockIdx.x;
    for (int sl = 0; sl < slots; ++sl) {
        const int c = sl * W + wg;
        if (c < n)
            for (int r = tid; r < n; r += TP_THREADS) cols[(size_t)sl * n + r] = G[(size_t)c * ld + r];
    }
    for (int r = tid; r < n; r += TP_THREADS) { va[r] = 0.0f; vb[r] = 0.0f; wp[r] = 0.0f; }
    __syncthreads();
    float *vprev = va, *vcur = vb;
    float tprev = 0.0f;
    // Between two exchanges a thread works on the rows r = tid + 1024 m it owns, in REGISTERS (what it gathered never
    // visits LDS; only w and v_j do, for the pass over the columns), and a block sum costs ONE barrier: the waves' partial
    // sums and the broadcast value of a sum (p_{j-1}[j], then x[j + 1]) go through a scratch that alternates between two
    // copies, so the write of a sum can never overtake the reads of the sum before last.
    __shared__ float rsh[2][20];
    __shared__ int bail;                    // a thread whose granule never came says so here, ahead of the step's first barrier
    if (tid == 0) bail = 0;
    int phase = 0;
    auto sum1 = [&](float &a, float bc_val, bool bc_mine, float &bc_out) {
        a = wave_sum_dpp(a);
        if (lane == 0) rsh[phase][wave] = a;
        if (bc_mine) rsh[phase][16] = bc_val;
        __syncthreads();
        constexpr int nw = TP_THREADS / 64;
        a = wave_sum_dpp(lane < nw ? rsh[phase][lane] : 0.0f);
        bc_out = rsh[phase][16];
        phase ^= 1;
    };
#ifdef THIP_TP_PROFILE
    unsigned long long tacc[5] = { 0, 0, 0, 0, 0 }, tlast = __builtin_amdgcn_s_memrealtime();
#define TP_STAMP(i) do { const unsigned long long tn_ = __builtin_amdgcn_s_memrealtime(); tacc[i] += tn_ - tlast; tlast = tn_; } while (0)
#else
#define TP_STAMP(i) do { } while (0)
#endif
    for (int j = 0; j + 2 < n; ++j) {
        const int par = j & 1;
        const bool upd = j > 0 && tprev != 0.0f;
        float pr[TP_GM], xr[TP_GM], vpr[TP_GM], wpr[TP_GM];
        bool failed = false;
        if (j > 0) {
            // S1: p_{j-1} (rows >= j) and column j as its owner had it before the update of step j - 1
            const unsigned long long *gp = gran + (size_t)(par ^ 1) * 2 * n, *gc = gp + n;
            unsigned pend = 0;
#pragma unroll
            for (int m = 0; m < TP_GM; ++m) {
                const int r = tid + TP_THREADS * m;
                pr[m] = 0.0f; xr[m] = 0.0f;
                if (r >= j && r < n) pend |= (1u << m) | (1u << (m + TP_GM));
            }
            int spins = 0;
            while (pend) {
                unsigned long long v[2 * TP_GM];
#pragma unroll
                for (int m = 0; m < 2 * TP_GM; ++m)
                    if ((pend >> m) & 1u)
                        v[m] = __hip_atomic_load((m < TP_GM ? gp : gc) + tid + TP_THREADS * (m % TP_GM), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int m = 0; m < 2 * TP_GM; ++m) {
                    if (((pend >> m) & 1u) && (unsigned)(v[m] >> 32) == (unsigned)j) {
                        const float val = __uint_as_float((unsigned)v[m]);
                        if (m < TP_GM) pr[m] = val; else xr[m - TP_GM] = val;
                        pend &= ~(1u << m);
                    }
                }
                if (pend) {
                    ++spins;
                    if (spins > TP_SPIN_MAX || ((spins & 1023) == 0 && __hip_atomic_load(errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        failed = true;
                        break;
                    }
                }
            }
            if (failed) bail = 1;
        } else {
#pragma unroll
            for (int m = 0; m < TP_GM; ++m) {
                const int r = tid + TP_THREADS * m;
                pr[m] = 0.0f;
                xr[m] = r < n ? G[r] : 0.0f;
            }
        }
        TP_STAMP(0);
        // w_{j-1} = p - (tau / 2)(p . v) v
        float acc = 0.0f, pj = 0.0f;
        bool mine = false;
#pragma unroll
        for (int m = 0; m < TP_GM; ++m) {
            const int r = tid + TP_THREADS * m;
            vpr[m] = (r >= j && r < n) ? vprev[r] : 0.0f;
            acc = fmaf(pr[m], vpr[m], acc);
            if (r == j) { mine = true; pj = pr[m]; }
        }
        float pj_all;
        sum1(acc, pj, mine, pj_all);
        TP_STAMP(1);
        if (bail != 0) {
            if (tid == 0) __hip_atomic_store(errflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        const float kk = -0.5f * tprev * acc;
        // column j after the pending update (v_{j-1}[j] = 1); d_j, the reflector v_j, tau_j, e_j -- every workgroup for itself
        const float wj = upd ? fmaf(kk, 1.0f, pj_all) : 0.0f, vj = upd ? 1.0f : 0.0f;
        float ss = 0.0f, alpha_mine = 0.0f;
        mine = false;
#pragma unroll
        for (int m = 0; m < TP_GM; ++m) {
            const int r = tid + TP_THREADS * m;
            wpr[m] = fmaf(kk, vpr[m], pr[m]);
            if (r >= j && r < n) {
                wp[r] = wpr[m];
                float x = xr[m];
                if (upd) x -= vpr[m] * wj + wpr[m] * vj;
                xr[m] = x;
                if (r >= j + 2) ss = fmaf(x, x, ss);
                if (r == j + 1) { mine = true; alpha_mine = x; }
                if (r == j && wg == 0) d[j] = x;
            }
        }
        float alpha;
        sum1(ss, alpha_mine, mine, alpha);
        TP_STAMP(2);
        // beta = -sign(alpha) sqrt(alpha^2 + ss): ss IS the sum of squares already, so hypot's overflow care buys nothing here;
        // the two quotients through v_rcp_f32 (1 ulp): these scalars sit on the critical path of every reflector
        float t = 0.0f, beta = alpha, scale = 0.0f;
        if (ss != 0.0f) {
            beta = -copysignf(sqrtf(fmaf(alpha, alpha, ss)), alpha);
            t = (beta - alpha) * __builtin_amdgcn_rcpf(beta);
            scale = __builtin_amdgcn_rcpf(alpha - beta);
        }
#pragma unroll
        for (int m = 0; m < TP_GM; ++m) {
            const int r = tid + TP_THREADS * m;
            if (r < n) {
                const float v = r <= j ? 0.0f : (r == j + 1 ? 1.0f : xr[m] * scale);
                vcur[r] = v;
                if (wg == 0 && r > j) Vh[(size_t)j * ld + r] = v;
            }
        }
        if (wg == 0 && tid == 0) { e[j] = beta; tau[j] = t; }
        __syncthreads();
        TP_STAMP(3);
        // S4: own columns c >= j + 1, rows >= j + 1: the update of step j - 1, then p_j[c] = tau_j column . v_j; one wave per column
        unsigned long long *gp = gran + (size_t)par * 2 * n, *gc = gp + n;
        const unsigned tag = (unsigned)(j + 1);
        for (int sl = wave; sl < slots; sl += TP_THREADS / 64) {
            const int c = sl * W + wg;
            if (c < j + 1 || c >= n) continue;
            float *col = cols + (size_t)sl * n;
            const float wc = upd ? wp[c] : 0.0f, vc = upd ? vprev[c] : 0.0f;
            const bool next = c == j + 1;
            float acc = 0.0f;
            for (int r0 = j + 1; r0 < n; r0 += 256) {
                float a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = r0 + lane + 64 * u;
                    a[u] = r < n ? col[r] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = r0 + lane + 64 * u;
                    if (r < n) {
                        if (upd) { a[u] -= vprev[r] * wc + wp[r] * vc; col[r] = a[u]; }
                        acc = fmaf(a[u], vcur[r], acc);
                        if (next) tp_store<XCD_LOCAL>(gc + r, tp_pack(a[u], tag));
                    }
                }
            }
            acc = wave_sum_dpp(acc);
            if (lane == 0) tp_store<XCD_LOCAL>(gp + c, tp_pack(t * acc, tag));
        }
        float *sw = vprev; vprev = vcur; vcur = sw;
        tprev = t;
        TP_STAMP(4);
    }
#ifdef THIP_TP_PROFILE
    if (wg == 0 && tid == 0) for (int i = 0; i < 5; ++i) stamps[i] = tacc[i];
#endif
    // the update of the last reflector on the 2 x 2 tail (tri_fin_k reads it from G)
    if (n >= 3) {
        const int j = n - 2;
        const unsigned long long *gp = gran + (size_t)((j & 1) ^ 1) * 2 * n;
        const bool ok = tp_gather(gp, ps, nullptr, nullptr, j, n, (unsigned)j, errflag);
        if (__syncthreads_or(ok ? 0 : 1)) {
            if (tid == 0) __hip_atomic_store(errflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (tid == 0) {
            const float acc = ps[j] * vprev[j] + ps[j + 1] * vprev[j + 1];
            const float kk = -0.5f * tprev * acc;
            wp[j] = fmaf(kk, vprev[j], ps[j]);
            wp[j + 1] = fmaf(kk, vprev[j + 1], ps[j + 1]);
        }
        __syncthreads();
        if (tid < 4) {
            const int c = j + (tid >> 1), r = j + (tid & 1);
            if (c % W == wg) {
                float a = cols[(size_t)(c / W) * n + r];
                if (tprev != 0.0f) a -= vprev[r] * wp[c] + wp[r] * vprev[c];
                G[(size_t)c * ld + r] = a;
            }
        }
    }
    // workgroup 0 has gathered every p and written every d, e, tau and reflector: the positive "done" (a launch in which
    // nobody took a role, or one that bailed, leaves 0 or 1 and the host redoes the reduction with launches)
    if (wg == 0 && tid == 0) atomicCAS(errflag, 0u, TP_DONE);
}

// the 2 x 2 tail (and the whole of n <= 2)
__global__ void tri_fin_k(int n, int ld, const float *__restrict__ G, float *__restrict__ d, float *__restrict__ e)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (n == 1) { d[0] = G[0]; e[0] = 0.0f; return; }
    d[n - 2] = G[(size_t)(n - 2) * ld + n - 2];
    d[n - 1] = G[(size_t)(n - 1) * ld + n - 1];
    e[n - 2] = G[(size_t)(n - 2) * ld + n - 1];
    e[n - 1] = 0.0f;
}

// Z(:, c) = H_0 H_1 .. H_{n-3} e_c, one wave per column (zero padded to ld x ld), four columns per workgroup.
// A column is a chain of n - 2 reflector applications (a dot product across the wave, then an update).  Round 2 kept the
// column in LDS and fetched each reflector from global memory when its turn came: every link of the chain was a string of
// dependent LDS / L2 round trips (1.1 us per reflector, 0.53 ms at n = 500).  Now the column lives in REGISTERS (lane l
// holds rows l, l + 64, ..: NQ values), the workgroup streams the reflectors through LDS FQ_G at a time, stored at their
// global row positions with zeros above (so that the register file and the LDS slot are indexed alike and every read
// of a link is issued at once), and the next group's loads are in flight while the current group is applied.
constexpr int FQ_G = 8;              // reflectors per group
template <int NQ>                    // rows per lane: n <= 64 NQ
__global__ __launch_bounds__(BLK) void form_q_k(int n, int ld, const float *__restrict__ Vh, const float *__restrict__ tau,
                                               float *__restrict__ Z)
{
    constexpr int NR = 64 * NQ;                 // slot length
    constexpr int PT = FQ_G * NR / BLK;         // staged elements per thread per group
    __shared__ float ring[2][FQ_G][NR];
    __shared__ float taus[NR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x * 4 + wave;
    float q[NQ];
#pragma unroll
    for (int m = 0; m < NQ; ++m) q[m] = (lane + 64 * m == c) ? 1.0f : 0.0f;
    for (int i = tid; i < NR; i += BLK) taus[i] = i + 2 < n ? tau[i] : 0.0f;
    const int nref = n - 2;                                    // reflectors j = nref - 1 .. 0; slot u of group g: j = jtop - u
    const int ngroups = nref > 0 ? (nref + FQ_G - 1) / FQ_G : 0;
    float nxt[PT];
    auto fetch = [&](int g) {
        const int jtop = nref - 1 - g * FQ_G;
#pragma unroll
        for (int w = 0; w < PT; ++w) {
            const int e = tid + w * BLK, u = e / NR, row = e % NR, j = jtop - u;
            nxt[w] = (j >= 0 && row > j && row < n) ? Vh[(size_t)j * ld + row] : 0.0f;
        }
    };
    auto land = [&](int half) {
#pragma unroll
        for (int w = 0; w < PT; ++w) {
            const int e = tid + w * BLK;
            (&ring[half][0][0])[e] = nxt[w];
        }
    };
    if (ngroups > 0) { fetch(0); land(0); }
    __syncthreads();
    for (int g = 0; g < ngroups; ++g) {
        const int half = g & 1;
        const int jtop = nref - 1 - g * FQ_G;
        const bool more = g + 1 < ngroups;
        if (more) fetch(g + 1);
        if (c < n) {
#pragma unroll
            for (int u = 0; u < FQ_G; ++u) {
                const int j = jtop - u;
                if (j < 0) break;
                const float t = taus[j];
                if (t == 0.0f) continue;
                const int m0 = (j + 1) / 64;                  // chunks below hold only zeros of this reflector
                float v[NQ];
#pragma unroll
                for (int m = 0; m < NQ; ++m) v[m] = m >= m0 ? ring[half][u][lane + 64 * m] : 0.0f;
                float sacc = 0.0f;
#pragma unroll
                for (int m = 0; m < NQ; ++m) sacc = fmaf(v[m], q[m], sacc);
                sacc = wave_sum_dpp(sacc) * t;
#pragma unroll
                for (int m = 0; m < NQ; ++m) q[m] = fmaf(-sacc, v[m], q[m]);
            }
        }
        if (more) land(half ^ 1);
        __syncthreads();
    }
    if (c < ld) {
#pragma unroll
        for (int m = 0; m < NQ; ++m) {
            const int r = lane + 64 * m;
            if (r < ld) Z[(size_t)c * ld + r] = (r < n && c < n) ? q[m] : 0.0f;
        }
    }
}

static int launch_form_q(hipStream_t st, int n, int ld, const float *Vh, const float *tau, float *Z)
{
    const dim3 g((unsigned)((ld + 3) / 4)), b(BLK);
    if (n <= 256) hipLaunchKernelGGL(form_q_k<4>, g, b, 0, st, n, ld, Vh, tau, Z);
    else if (n <= 512) hipLaunchKernelGGL(form_q_k<8>, g, b, 0, st, n, ld, Vh, tau, Z);
    else if (n <= 1024) hipLaunchKernelGGL(form_q_k<16>, g, b, 0, st, n, ld, Vh, tau, Z);
    else hipLaunchKernelGGL(form_q_k<32>, g, b, 0, st, n, ld, Vh, tau, Z);
    THIP_LAUNCH_CHECK();
    return 0;
}

struct RotSweep { int start, count, off, pad; };     // rotations on columns (i, i + 1), i = start, start - 1, ..

// replays the QL rotations on the rows of Z: RB rows per workgroup in LDS as zs[column][row], one lane per row.
// The replay is a chain: every rotation needs the entry its predecessor left.  Rows are independent but LDS holds
// only ~64 of them per CU, and ONE wave issues ~8 instructions per rotation at one instruction per ~8 cycles.  So the
// RW waves of a workgroup work on the SAME rows and take the sweeps round-robin (wave w: sweeps w, w + RW, ..), as a
// software pipeline: sweep q + 1 follows sweep q down the columns and may touch column c only after sweep q has
// moved below it.  Progress of sweep q lives in LDS slot q & 7 as the key (q << 12) | (4095 - (lowest index done + 1)),
// published with atomicMax: keys only grow -- within a sweep as it descends, and from sweep q to the sweep q + 8 that
// reuses the slot -- so a late writer can never hide a newer sweep, and a reader that finds a LARGER sweep number in the
// slot knows its predecessor finished long ago.
constexpr int RW = 4;
template <int RB>
__global__ __launch_bounds__(64 * RW) void rot_apply_k(int n, int ld, float *__restrict__ Z, const float2 *__restrict__ rot,
                                                      const RotSweep *__restrict__ sw, int nsw)
{
    extern __shared__ float zs[];
    __shared__ int prog[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool act = lane < RB;
    const int row = blockIdx.x * RB + lane;
    const bool ok = act && row < n;
    if (threadIdx.x < 8) prog[threadIdx.x] = -1;
    for (int c = wave; c < n; c += RW) if (act) zs[c * RB + lane] = ok ? Z[(size_t)c * ld + row] : 0.0f;
    __syncthreads();
    volatile int *vprog = prog;
    // The (c, s) pairs: each lane fetches ONE pair of the next 64 rotations (a coalesced load: one memory round trip per
    // 64 rotations) and the pairs are broadcast from lane u with v_readlane; the block after the one being replayed is
    // already in flight.
    for (int q = wave; q < nsw; q += RW) {
        const RotSweep w = sw[q];
        float2 mine = lane < min(64, w.count) ? rot[w.off + lane] : make_float2(1.0f, 0.0f);
        if (lane == 0) atomicMax(&prog[q & 7], q << 12);         // sweep q: nothing done yet (done + 1 = 4095)
        int i = w.start;
        bool first = true;
        float hi = 0.0f;
        for (int t0 = 0; t0 < w.count; t0 += 64) {
            const int nb = min(64, w.count - t0);
            float2 next = make_float2(1.0f, 0.0f);
            if (t0 + 64 < w.count && lane < min(64, w.count - t0 - 64)) next = rot[w.off + t0 + 64 + lane];
            for (int u = 0; u < nb; u += 8) {
                const int m8 = min(8, nb - u);
                // columns i - m8 + 1 .. i + 1 are touched: the previous sweep must be below them
                if (q > 0) {
                    const int need = i - m8 + 1;
                    for (;;) {
                        const int v = vprog[(q - 1) & 7];
                        const int tag = v >> 12;
                        if (tag > q - 1 || (tag == q - 1 && (4095 - (v & 4095)) - 1 < need)) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                if (first) { hi = act ? zs[(i + 1) * RB + lane] : 0.0f; first = false; }
                if (m8 == 8) {
                    float lo[8], nh[8];
#pragma unroll
                    for (int v = 0; v < 8; ++v) lo[v] = act ? zs[(i - v) * RB + lane] : 0.0f;
#pragma unroll
                    for (int v = 0; v < 8; ++v) {
                        const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.x), u + v));
                        const float sn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.y), u + v));
                        // the only loop-carried value is hi: ONE dependent FMA per rotation on that chain
                        const float cl = c * lo[v], sl = sn * lo[v];
                        nh[v] = fmaf(c, hi, sl);                       // s z_i + c z_{i+1}
                        hi = fmaf(-sn, hi, cl);                        // c z_i - s z_{i+1}
                    }
                    if (act) {
#pragma unroll
                        for (int v = 0; v < 8; ++v) zs[(i - v + 1) * RB + lane] = nh[v];
                    }
                    i -= 8;
                } else {
                    for (int v = 0; v < m8; ++v, --i) {
                        const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.x), u + v));
                        const float sn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.y), u + v));
                        const float lo = act ? zs[i * RB + lane] : 0.0f;
                        if (act) zs[(i + 1) * RB + lane] = fmaf(c, hi, sn * lo);
                        hi = fmaf(-sn, hi, c * lo);
                    }
                }
                // column i + 1 is still in a register (hi): everything above it is final for this sweep
                const bool last = t0 + u + m8 >= w.count;
                if (last && act) zs[(i + 1) * RB + lane] = hi;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) atomicMax(&prog[q & 7], (q << 12) | (4095 - (last ? 0 : (i + 2))));   // done = i + 1 (last: -1)
            }
            mine = next;
        }
    }
    __syncthreads();
    if (ok)
        for (int c = wave; c < n; c += RW) Z[(size_t)c * ld + row] = zs[c * RB + lane];
}

// X(:, i) = e_i V(:, i)   (ld x ld, padding stays zero)
__global__ void scale_cols_k(int ld, const float *__restrict__ V, const float *__restrict__ e, int n, float *__restrict__ X)
{
    const size_t tot = (size_t)ld * ld;
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < tot; i += (size_t)gridDim.x * BLK) {
        const int c = (int)(i / ld);
        X[i] = c < n ? e[c] * V[i] : 0.0f;
    }
}

// packed(r, c) = C(r, c), r <= c, diag / scale
__global__ void pack_sym_k(int n, int ld, const float *__restrict__ Cm, int has_scale, float scale, float *__restrict__ packed)
{
    const int c = blockIdx.y;
    for (int r = blockIdx.x * BLK + threadIdx.x; r <= c; r += gridDim.x * BLK) {
        float v = 0.5f * (Cm[(size_t)c * ld + r] + Cm[(size_t)r * ld + c]);
        if (r == c && has_scale) v = v / scale;
        packed[(size_t)c * (c + 1) / 2 + r] = v;
    }
}

// implicit-shift QL on the tridiagonal (d, e) in f64 (EISPACK tql2 / NR tqli recurrence); every Givens rotation is
// recorded as (c, s), grouped in sweeps.  Returns false if an eigenvalue needs more than 60 iterations.
template <typename Emit>
bool ql_record(int n, std::vector<double> &d, std::vector<double> &e, std::vector<float2> &rot, std::vector<RotSweep> &sweeps,
               size_t chunk_rotations, Emit emit)
{
    const double eps = 1.1102230246251565e-16;
    for (int l = 0; l < n; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; ++m) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= eps * dd) break;
            }
            if (m != l) {
                if (iter++ == 60) return false;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::sqrt(g * g + 1.0);
                g = d[m] - d[l] + e[l] / (g + std::copysign(r, g));
                double s = 1.0, c = 1.0, p = 0.0;
                RotSweep sw{ m - 1, 0, (int)rot.size(), 0 };
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i];
                    const double b = c * e[i];
                    e[i + 1] = r = std::sqrt(f * f + g * g);     // |f|, |g| <= ||T||: no overflow guard needed in f64
                    if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
                    s = f / r; c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    d[i + 1] = g + (p = s * r);
                    g = c * r - b;
                    rot.push_back(make_float2((float)c, (float)s));
                    sw.count += 1;
                }
                if (sw.count) sweeps.push_back(sw);
                // hand the record to the device in chunks: the replay of chunk k runs while the host computes chunk k + 1
                if (rot.size() >= chunk_rotations) { if (!emit()) return false; rot.clear(); sweeps.clear(); }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p; e[l] = g; e[m] = 0.0;
            }
        } while (m != l);
    }
    return true;
}

struct Work {
    float *G, *V, *S, *Y, *Z;      // ld x ld each
    float *w, *e;                  // ld each
    float *sc;                     // 16 scalars
    float *part;                   // 512 block partials
    int   *counters;               // MAX_SWEEPS + 2 ints
};

Work carve(float *work, size_t n)
{
    const size_t ld = np_of(n), sq = pitch_of(ld) * ld;
    Work k;
    k.G = work; k.V = k.G + sq; k.S = k.V + sq; k.Y = k.S + sq; k.Z = k.Y + sq;
    k.w = k.Z + sq; k.e = k.w + ld; k.sc = k.e + ld; k.part = k.sc + 16;
    k.counters = reinterpret_cast<int *>(k.part + 512);
    return k;
}

int decompose_tridiag(hipStream_t st, size_t n, const float *packed, int has_scale, float scale, const Work &k, int map_kind);
constexpr int TRI_MIN_N = 32;        // above: Householder + QL (host-facing calls); up to here the one-workgroup Jacobi

int decompose(hipStream_t st, size_t n, const float *packed, int has_scale, float scale, const Work &k,
              int map_kind, const int *stop)
{
    if (n > (size_t)TRI_MIN_N && stop == nullptr && !getenv("THIP_EIG_JACOBI"))
        return decompose_tridiag(st, n, packed, has_scale, scale, k, map_kind);
    const int ni = (int)n, ld = (int)np_of(n);
    const unsigned g = grid_for((size_t)ld * ld, BLK, 512);
    hipLaunchKernelGGL(unpack_k, dim3(g), dim3(BLK), 0, st, ni, ld, packed, has_scale, scale, k.G, k.V, k.part, stop);
    hipLaunchKernelGGL(shift_k, dim3(1), dim3(BLK), 0, st, ni, ld, (int)g, k.part, k.G, k.sc, 1, stop);
    hipLaunchKernelGGL(normalise_k, dim3(g), dim3(BLK), 0, st, (size_t)ld * ld, k.G, k.sc, stop);
    if (n <= SMALL_N) {
        hipLaunchKernelGGL(jacobi_small_k, dim3(1), dim3(BLK), 0, st, ni, ld, k.G, k.V, stop);
    } else {
        hipLaunchKernelGGL(zero_counters_k, dim3(1), dim3(64), 0, st, k.counters, MAX_SWEEPS + 2);
        const int n_even = (ni + 1) & ~1;
        const unsigned blocks = (unsigned)((n_even / 2 + 3) / 4);
        for (int sweep = 0; sweep < MAX_SWEEPS; ++sweep) {
            for (int step = 0; step < n_even - 1; ++step)
                hipLaunchKernelGGL(jacobi_step_k, dim3(blocks), dim3(BLK), 0, st, ni, ld, k.G, k.V, step, k.counters,
                                   sweep, stop);
            if (stop == nullptr) {
                // host-facing call (thip_map_eig / thip_eig_decompose): stop enqueuing once a sweep rotated nothing
                int rotated = 1;
                THIP_TRY(hipMemcpyAsync(&rotated, k.counters + sweep, sizeof(int), hipMemcpyDeviceToHost, st));
                THIP_TRY(hipStreamSynchronize(st));
                if (rotated == 0) break;
            }
        }
    }
    hipLaunchKernelGGL(eigvals_k, dim3((unsigned)((n + 3) / 4)), dim3(BLK), 0, st, ni, ld, k.G, k.V, k.sc, map_kind, k.w,
                       k.e, stop);
    THIP_LAUNCH_CHECK();
    return 0;
}

template <int RB>
int launch_rot(hipStream_t st, int n, int ld, float *Z, const float2 *rot, const RotSweep *sw, int nsw)
{
    const size_t lds = (size_t)n * RB * sizeof(float);
    THIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&rot_apply_k<RB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(rot_apply_k<RB>, dim3((n + RB - 1) / RB), dim3(64 * RW), lds, st, n, ld, Z, rot, sw, nsw);
    THIP_LAUNCH_CHECK();
    return 0;
}

// which engine served the last decomposition of order > 32 (thip_eig_engine_info), and the test switch
int   g_eig_force = 0;          // thip_test_eig_force: 0 = default, 1 = the QL engine, 2 = the device engine with a failing certificate
int   g_eig_engine = 0;         // 1 = host QL + rotation replay, 2 = multisection + twisted factorisation, 3 = 2 failed its certificate -> 1
int   g_eig_polish = 0;
int   g_tri_force = 0;          // thip_test_eig_force bits 2-3: 1 (+ 4) the persistent reduction over the whole device, 2 (+ 8) on one XCD, 3 (+ 12) launches
int   g_tri_sabotage = 0;       // thip_test_eig_force + 16: the one-XCD persistent launch is started with a role missing (time-out path)
int   g_persist_broken = 0;     // a persistent launch gave up once: do not pay its time-out again (thip_test_eig_force resets it)
int   g_tri_persist = 0;        // how the last reduction ran: 1 = persistent launch, 0 = one launch per reflector, -1 = persistent gave up -> 0
float g_eig_orth = 0.0f, g_eig_resid = 0.0f;

// Q^T M Q = T: d -> k.Y[0 .. ld), e -> k.Y[ld .. 2 ld), tau -> k.Y[2 ld .. 3 ld), reflectors -> k.S (below the diagonal)
int tridiagonalise(hipStream_t st, int ni, int ld, const float *packed, int has_scale, float scale, const Work &k,
                   int persist)         // 0: one launch per reflector, 1: persistent over the whole device, 2: persistent on one XCD
{
    if (ni > TRI_MAXN) return fail(THIP_E_INVALID, "map_eig: order above 2048", __FILE__, __LINE__);
    const unsigned g = grid_for((size_t)ld * ld, BLK, 512);
    hipLaunchKernelGGL(unpack_k, dim3(g), dim3(BLK), 0, st, ni, ld, packed, has_scale, scale, k.G, (float *)nullptr, k.part,
                       (const int *)nullptr);
    float *d = k.Y, *e = k.Y + ld, *tau = k.Y + 2 * (size_t)ld, *p = k.Y + 3 * (size_t)ld, *Vh = k.S;
    unsigned *errflag = reinterpret_cast<unsigned *>(k.sc + 6);
    if (persist) {
        // k.Z is free until the eigenvector stage: 4 n granules of 8 bytes, zeroed (tag 0 is never waited for), then the ticket
        const bool local = persist == 2;
        const int W = local ? std::min(32, (ni + 3) / 4) : (ni + (ni <= 1536 ? 16 : 12) - 1) / (ni <= 1536 ? 16 : 12);
        const int slots = local ? (ni + W - 1) / W : (ni <= 1536 ? 16 : 12);
        unsigned long long *gran = reinterpret_cast<unsigned long long *>(k.Z);
        unsigned *ticket = reinterpret_cast<unsigned *>(gran + 4 * (size_t)ni);
        unsigned long long *stamps = reinterpret_cast<unsigned long long *>(k.part + 128);      // -DTHIP_TP_PROFILE: phase times of workgroup 0
        THIP_TRY(hipMemsetAsync(gran, 0, (4 * (size_t)ni + 1) * sizeof(unsigned long long), st));
        if (g_tri_sabotage && local) {           // test: role 0 is never taken, so every workgroup runs into its spin bound
            const unsigned one = 1u;
            THIP_TRY(hipMemcpyAsync(ticket, &one, sizeof(one), hipMemcpyHostToDevice, st));
        }
        THIP_TRY(hipMemsetAsync(errflag, 0, sizeof(unsigned), st));
        const size_t lds = ((size_t)slots + 4) * ni * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            THIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&tri_persist_k<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         140 * 1024));
            THIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&tri_persist_k<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         156 * 1024));
            attr_set = true;
        }
        if (local) hipLaunchKernelGGL(tri_persist_k<true>, dim3((unsigned)(8 * W + 64)), dim3(TP_THREADS), lds, st, ni, ld, W, slots,
                                      k.G, Vh, d, e, tau, gran, errflag, ticket, stamps);
        else hipLaunchKernelGGL(tri_persist_k<false>, dim3((unsigned)W), dim3(TP_THREADS), lds, st, ni, ld, W, slots, k.G, Vh, d, e,
                                tau, gran, errflag, ticket, stamps);
    } else {
        float *pbuf[2] = { p, p + ld };            // p_{j-1} is read while p_j is written
        for (int j = 0; j + 2 < ni; ++j) {
            const unsigned blocks = (unsigned)((ni - j - 1 + 3) / 4);
#define THIP_TRI_STEP(NQ) hipLaunchKernelGGL(tri_step_k<NQ>, dim3(blocks), dim3(BLK), 0, st, ni, ld, j, j == 0 ? 1 : 0, k.G, Vh, \
                                             pbuf[(j + 1) & 1], pbuf[j & 1], d, e, tau)
            const int Lp = ni - j;             // the kernel is instantiated by the length that is left, not by n
            if (Lp <= 256) THIP_TRI_STEP(1);
            else if (Lp <= 512) THIP_TRI_STEP(2);
            else if (Lp <= 1024) THIP_TRI_STEP(4);
            else THIP_TRI_STEP(8);
#undef THIP_TRI_STEP
        }
        if (ni >= 3) {              // the last reflector's update of the 2 x 2 tail is still pending
            const int j = ni - 3;
            hipLaunchKernelGGL(tri_upd_k, dim3((unsigned)((ni - j - 1 + 3) / 4)), dim3(BLK), 0, st, ni, ld, j, k.G, Vh, pbuf[j & 1], tau);
        }
    }
    hipLaunchKernelGGL(tri_fin_k, dim3(1), dim3(64), 0, st, ni, ld, k.G, d, e);
    THIP_LAUNCH_CHECK();
    return 0;
}

int eig_pin_floats(size_t want, float **out)
{
    Ctx &cx = ctx();
    if (cx.eig_pin_floats < want) {
        if (cx.eig_pin) THIP_TRY(hipHostFree(cx.eig_pin));
        cx.eig_pin = nullptr; cx.eig_pin_floats = 0;
        THIP_TRY(hipHostMalloc((void **)&cx.eig_pin, want * sizeof(float), hipHostMallocDefault));
        cx.eig_pin_floats = want;
    }
    *out = cx.eig_pin;
    return 0;
}

// The tridiagonal eigenproblem on the device (thip_trieig.hip), the back-transform Z = Q V0 and the certificate
// ||Z Z^T - I||_F, max residual -- three products on the matrix cores (this engine's dense contractions).
//   k.Z <- Q (form_q_k), k.G <- V0 (T's eigenvectors), k.V <- Q V0, k.G <- P = 3/2 I - 1/2 Z Z^T, [k.Z <- P Z -> k.V]
// k.S (reflectors) and the head of k.Y (d, e, tau) stay intact, so a failed certificate can hand over to the QL engine.
// SYNC: one read-back of the eigenvalues and the two certificate numbers per measurement.
int decompose_device(hipStream_t st, size_t n, const Work &k, int map_kind, int *ok, int *tri_failed)
{
    static std::mutex eig_mu;
    std::lock_guard<std::mutex> eig_lock(eig_mu);
    *ok = 0;
    const int ni = (int)n, ld = (int)np_of(n);
    float *d = k.Y, *e = k.Y + ld, *tau = k.Y + 2 * (size_t)ld, *Vh = k.S;
    // Q is formed on a side stream under the tridiagonal eigenproblem: both need only what the reduction left
    Ctx &cx = ctx();
    static const int side_on = getenv("THIP_EIG_SIDE") ? atoi(getenv("THIP_EIG_SIDE")) : 1;
    const bool side = side_on != 0 && ni >= 128;
    if (side) {
        if (cx.eig_side == nullptr) {
            THIP_TRY(hipStreamCreateWithFlags(&cx.eig_side, hipStreamNonBlocking));
            for (int b = 0; b < 2; ++b) THIP_TRY(hipEventCreateWithFlags(&cx.eig_ev[b], hipEventDisableTiming));
        }
        THIP_TRY(hipEventRecord(cx.eig_ev[0], st));
        THIP_TRY(hipStreamWaitEvent(cx.eig_side, cx.eig_ev[0], 0));
        THIP_RC(launch_form_q(cx.eig_side, ni, ld, Vh, tau, k.Z));
        THIP_TRY(hipEventRecord(cx.eig_ev[1], cx.eig_side));
    } else THIP_RC(launch_form_q(st, ni, ld, Vh, tau, k.Z));
    float *scr = nullptr;
    THIP_RC(scratch(tri_eigen_scratch_floats(ni), &scr));
    unsigned *cert = reinterpret_cast<unsigned *>(k.sc + 4);
    THIP_RC(tri_eigen(st, ni, ld, d, e, k.w, k.G, cert, scr));
    if (map_kind >= 0) THIP_RC(tri_map(st, ni, ld, map_kind, k.w, k.e));
    if (side) THIP_TRY(hipStreamWaitEvent(st, cx.eig_ev[1], 0));
    THIP_RC(gemm(st, true, ni, ld, 1.0f, k.Z, k.G, 0.0f, nullptr, 0.0f, k.V, nullptr));
    const int nparts = 64;
    const size_t back = 2 * (size_t)ld + 16 + nparts;          // k.w, k.e, k.sc, k.part are adjacent (carve)
    float *pin = nullptr;
    THIP_RC(eig_pin_floats(back, &pin));
    const float thr = std::fmax(1.5e-7f * (float)ni, 1.0e-5f);
    g_eig_polish = 0;
    for (int round = 0;; ++round) {
        THIP_RC(gemm(st, false, ni, ld, -0.5f, k.V, k.V, 0.0f, nullptr, 1.5f, k.G, nullptr));
        THIP_RC(tri_orth_partials(st, ni, ld, k.G, k.part, nparts));
        THIP_TRY(hipMemcpyAsync(pin, k.w, back * sizeof(float), hipMemcpyDeviceToHost, st));
        THIP_TRY(hipStreamSynchronize(st));
        double acc = 0.0;
        for (int i = 0; i < nparts; ++i) acc += (double)pin[2 * (size_t)ld + 16 + i];
        unsigned bits;
        memcpy(&bits, pin + 2 * (size_t)ld + 4, sizeof(bits));
        float resid;
        memcpy(&resid, &bits, sizeof(resid));
        const float orth = 2.0f * (float)std::sqrt(acc);
        g_eig_orth = orth; g_eig_resid = resid;
        if (tri_failed != nullptr && round == 0) {
            unsigned fl;
            memcpy(&fl, pin + 2 * (size_t)ld + 6, sizeof(fl));
            if (fl != TP_DONE) { *tri_failed = 1; return 0; }
        }
        if (g_eig_force == 2 || !(resid <= 1.0e-9f) || !(orth < 0.5f)) return 0;          // -> the QL engine
        if (orth <= (round == 0 ? thr : 2.0f * thr)) break;
        if (round == 3) return 0;
        // Newton-Schulz polish: Z <- (3/2 I - 1/2 Z Z^T) Z, quadratic in ||Z Z^T - I||
        THIP_RC(gemm(st, true, ni, ld, 1.0f, k.G, k.V, 0.0f, nullptr, 0.0f, k.Z, nullptr));
        THIP_TRY(hipMemcpyAsync(k.V, k.Z, (size_t)ld * ld * sizeof(float), hipMemcpyDeviceToDevice, st));
        g_eig_polish = round + 1;
    }
    *ok = 1;
    return 0;
}

int decompose_ql(hipStream_t st, size_t n, const Work &k, int map_kind);

// M = Z diag(w) Z^T: Householder tridiagonalisation, then T's eigenproblem on the device (certified) or, failing that, by
// QL on the host; eigenvectors -> k.V (columns), eigenvalues -> k.w, and for map_kind 0 / 1 the mapped values -> k.e.  SYNC.
int decompose_tridiag(hipStream_t st, size_t n, const float *packed, int has_scale, float scale, const Work &k, int map_kind)
{
    const int ni = (int)n, ld = (int)np_of(n);
    static const int env_ql = getenv("THIP_EIG_QL") ? atoi(getenv("THIP_EIG_QL")) : 0;
    static const int env_persist = getenv("THIP_TRI_PERSIST") ? atoi(getenv("THIP_TRI_PERSIST")) : 2;       // 0 launches, 1 whole device, 2 one XCD (orders <= 1024)
    for (int attempt = 0; attempt < 2; ++attempt) {
        int persist = 0;
        if (attempt == 0 && !g_persist_broken) {
            persist = g_tri_force != 0 ? (g_tri_force == 3 ? 0 : g_tri_force) : env_persist;
            if (persist == 2 && (ni > 1024 || (((size_t)(ni + 31) / 32) + 5) * ni * sizeof(float) > 150 * 1024)) persist = 0;
        }
        g_tri_persist = attempt == 0 ? persist : -1;
        THIP_RC(tridiagonalise(st, ni, ld, packed, has_scale, scale, k, persist));
        int tri_failed = 0;
        if (!env_ql && g_eig_force != 1) {
            int ok = 0;
            THIP_RC(decompose_device(st, n, k, map_kind, &ok, persist ? &tri_failed : nullptr));
            g_eig_engine = ok ? 2 : 3;
            if (ok) return 0;
        } else {
            g_eig_engine = 1;
            if (persist) {
                float f = 0.0f;
                THIP_RC(fetch_scalar(k.sc + 6, &f));
                unsigned fl;
                memcpy(&fl, &f, sizeof(fl));
                tri_failed = fl != TP_DONE;
            }
        }
        if (tri_failed) { g_persist_broken = 1; continue; }           // the persistent launch gave up: one launch per reflector
        return decompose_ql(st, n, k, map_kind);
    }
    return fail(THIP_E_NOCONV, "map_eig: tridiagonalisation failed twice", __FILE__, __LINE__);
}

// the QL engine (round 2): d, e visit the host, the rotations are replayed on Q.  SYNC.
int decompose_ql(hipStream_t st, size_t n, const Work &k, int map_kind)
{
    const int ni = (int)n, ld = (int)np_of(n);
    float *d = k.Y, *tau = k.Y + 2 * (size_t)ld, *Vh = k.S;
    std::vector<float> hde(2 * (size_t)ld);
    THIP_TRY(hipMemcpyAsync(hde.data(), d, 2 * (size_t)ld * sizeof(float), hipMemcpyDeviceToHost, st));
    THIP_RC(launch_form_q(st, ni, ld, Vh, tau, k.V));      // runs under the host QL
    THIP_TRY(hipStreamSynchronize(st));
    std::vector<double> dd(ni), ee(ni, 0.0);
    for (int i = 0; i < ni; ++i) { dd[i] = hde[i]; if (i + 1 < ni) ee[i] = hde[ld + i]; }
    // The record goes to the device in chunks of >= 16k rotations (whole sweeps): two device buffers alternate, an event
    // each says "the replay that read this buffer has finished".
    const size_t chunk = 16384, cap_rot = chunk + (size_t)ni + 8, cap_sw = chunk / 2 + 16;
    const size_t per = 2 * cap_rot + 4 * cap_sw;             // floats per buffer (float2 pairs, then 16-byte sweep records)
    float *scr = nullptr;
    THIP_RC(scratch(2 * per + 64, &scr));
    // pinned staging for the uploads, one half per device buffer (the same event guards both)
    // (context-owned: released by thip_shutdown; one decomposition at a time per context -- the mutex covers the whole
    // replay, like the reference's one-thread-per-backend contract, linalg_ex.rs / cuda_mgr.rs thread_local state)
    static std::mutex eig_mu;
    std::lock_guard<std::mutex> eig_lock(eig_mu);
    float *pin_base = nullptr;
    THIP_RC(eig_pin_floats(2 * per, &pin_base));
    float *const pin = pin_base;
    hipEvent_t ev[2] = { nullptr, nullptr };
    int which = 0, rc = 0;
    std::vector<float2> rot;
    std::vector<RotSweep> sweeps;
    rot.reserve(cap_rot);
    auto emit = [&]() -> bool {
        if (sweeps.empty()) return true;
        if (sweeps.size() > cap_sw) { rc = fail(THIP_E_NOCONV, "QL: too many short sweeps", __FILE__, __LINE__); return false; }
        float *buf = scr + (size_t)which * per;
        float2 *drot = reinterpret_cast<float2 *>(buf);
        RotSweep *dsw = reinterpret_cast<RotSweep *>(buf + 2 * cap_rot);
        if (ev[which] == nullptr) { if (hipEventCreateWithFlags(&ev[which], hipEventDisableTiming) != hipSuccess) { rc = -1; return false; } }
        else if (hipEventSynchronize(ev[which]) != hipSuccess) { rc = -1; return false; }
        const int off0 = sweeps[0].off;
        for (RotSweep &w : sweeps) w.off -= off0;            // offsets relative to the chunk
        float *hbuf = pin + (size_t)which * per;
        memcpy(hbuf, rot.data(), rot.size() * sizeof(float2));
        memcpy(hbuf + 2 * cap_rot, sweeps.data(), sweeps.size() * sizeof(RotSweep));
        if (hipMemcpyAsync(drot, hbuf, rot.size() * sizeof(float2), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(dsw, hbuf + 2 * cap_rot, sweeps.size() * sizeof(RotSweep), hipMemcpyHostToDevice, st) != hipSuccess) { rc = -1; return false; }
        int r2;
        if ((size_t)ni * 64 * sizeof(float) <= 150 * 1024) r2 = launch_rot<64>(st, ni, ld, k.V, drot, dsw, (int)sweeps.size());
        else if ((size_t)ni * 32 * sizeof(float) <= 150 * 1024) r2 = launch_rot<32>(st, ni, ld, k.V, drot, dsw, (int)sweeps.size());
        else r2 = launch_rot<16>(st, ni, ld, k.V, drot, dsw, (int)sweeps.size());
        if (r2 != 0) { rc = r2; return false; }
        hipEventRecord(ev[which], st);
        which ^= 1;
        return true;
    };
    const bool conv = ql_record(ni, dd, ee, rot, sweeps, chunk, emit);
    if (conv) emit();
    THIP_TRY(hipStreamSynchronize(st));
    for (int b = 0; b < 2; ++b) if (ev[b]) hipEventDestroy(ev[b]);
    if (rc != 0) return rc > 0 ? rc : fail(THIP_E_INVALID, "QL replay: HIP call failed", __FILE__, __LINE__);
    if (!conv) return fail(THIP_E_NOCONV, "QL iteration did not converge", __FILE__, __LINE__);
    std::vector<float> hw(2 * (size_t)ld, 0.0f);
    for (int i = 0; i < ni; ++i) {
        const float lam = (float)dd[i];
        hw[i] = lam;
        if (map_kind == 0) hw[ld + i] = lam > 0.0f ? lam : 0.0f;
        else if (map_kind == 1) hw[ld + i] = lam > 0.0f ? std::sqrt(lam) : 0.0f;
    }
    // k.w and k.e are adjacent (carve): one upload
    THIP_TRY(hipMemcpyAsync(k.w, hw.data(), 2 * (size_t)ld * sizeof(float), hipMemcpyHostToDevice, st));
    THIP_TRY(hipStreamSynchronize(st));
    return 0;
}

// packed <- V diag(e) V^T on the matrix cores: X = V diag(e), C = X V^T (gemm(false)), then the symmetric pack
int rebuild_mfma(hipStream_t st, size_t n, float *packed, int has_scale, float scale, const Work &k)
{
    const int ni = (int)n, ld = (int)np_of(n);
    const unsigned g = grid_for((size_t)ld * ld, BLK, 512);
    hipLaunchKernelGGL(scale_cols_k, dim3(g), dim3(BLK), 0, st, ld, k.V, k.e, ni, k.S);
    THIP_RC(gemm(st, false, ni, ld, 1.0f, k.S, k.V, 0.0f, nullptr, 0.0f, k.Z, nullptr));
    dim3 gp((unsigned)((n + BLK - 1) / BLK), (unsigned)n);
    hipLaunchKernelGGL(pack_sym_k, gp, dim3(BLK), 0, st, ni, ld, k.Z, has_scale, scale, packed);
    THIP_LAUNCH_CHECK();
    return 0;
}

int rebuild(hipStream_t st, size_t n, float *packed, int has_scale, float scale, const Work &k, const int *stop)
{
    if (n > (size_t)TRI_MIN_N && stop == nullptr) return rebuild_mfma(st, n, packed, has_scale, scale, k);
    const int ni = (int)n, ld = (int)np_of(n);
    dim3 g((unsigned)((n + BLK - 1) / BLK), (unsigned)n);
    hipLaunchKernelGGL(rebuild_k, g, dim3(BLK), 0, st, ni, ld, k.V, k.e, has_scale, scale, packed, stop);
    THIP_LAUNCH_CHECK();
    return 0;
}

// P = (M + M sign(M)) / 2 through the matrix cores; nb items per launch (work regions ws floats apart, packed vectors ps floats
// apart): the chain is launch-bound at k = 500, so the x_y and x_s projections of one iteration share its launches.
// sign(M) = the polar factor of S = M / ||M||_F by the polar iteration S <- p(S^2) S with odd polynomials: only GEMMs.
//   Lifting phase: ONE polynomial, the LP solution of "maximise the gain s subject to p(x) >= s x below the band, lo <= p(x) <= hi
//   on the band" (tools/polar_coeffs.py): every singular value below the band grows by s per step and a value inside the band STAYS
//   inside.  The polynomial ACCEPTS (0, hi] and RETURNS slightly inside the band: with "returns <= hi" the LP solution has
//   p(hi) = hi, a fixed point with p' >> 1 -- a singular value that reaches the interior maximum lands on it and round-off decides
//   which way it leaves; upwards is an overflow within a few steps (it happened in round 2: a 20 x 20 iterate of
//   test_synth_sdp_converges_to_oracle_objective; tests/golden/psd_k20_band_edge_iterate.npy).  The unconstrained minimax
//   composition ("Polar Express", Amsel et al. 2025) is two steps shorter, but its early polynomials equioscillate between ~0 and
//   2: an already-large eigenvalue can be thrown back to 1e-6, below the absolute round-off of the evaluation (measured: 3e-5 |X|
//   error on rank-deficient inputs instead of 2e-8).
//   Then minimax polynomials of 1 on the band, and one Newton-Schulz step x (3 - x^2) / 2 that squares the remaining error.
// (Rounds 1-4 ran quintic steps with general T S products, 48 launches; removed in round 6 when the degree-7 chain below learnt
// orders above 512.)
// Round 5: the same projection with every product SYMMETRIC and degree-7 steps -- 37 launches instead of 48.
//   * S is kept bitwise symmetric: T S is computed like S S^T (lower triangle of tiles, mirrored; diagonal tiles averaged
//     with their transpose, `dsym`), which turns the antisymmetric round-off of a step into a symmetric perturbation of
//     the same size instead of carrying it along (measured, tools/psd_err_sweep.py and the numpy restatement in DESIGN.md:
//     <= 3e-7 |X| on the spectra of tests/test_gpu_eig.py, 2-5e-8 at k = 500).  With S symmetric every operand of every
//     product can be read as rows, and half the tiles of the 16 general products go away.
//   * an odd polynomial of degree 7 is x q(x^2) with q cubic, and a real cubic always has a real root:
//     p(S) = U V with U = q0 Y^2 + q1 Y + q2 I and V = (Y - r0 I) S, Y = S^2.  Y Y and Y S share their first factor and
//     run as ONE launch (polar_dual_k), so a degree-7 step is three dependent launches like the quintic's -- and the
//     chain is bound by its launches, not its flops.  Gain 5.64 per step instead of 3.94 on the band
//     [0.15, 1.85] (tools/polar_coeffs.py: the polynomial accepts (0, 1.85] and returns [0.1525, 1.828]): 8 lifting
//     steps bring relative eigenvalues >= 1e-7 into the band (5.644^8 = 1.03e6; S_0 = 1.85 M / ||M||_F), 3 minimax
//     steps take the band to 1 +- 8e-7, and the Newton-Schulz step shares a launch with M S:
//       {T = 1.5 I - 0.5 S S, R = M S}  ->  M sign(M) = T R, packed by the last launch itself.
// Orders above 512 (round 6): the same chain, its kernels walking K in chunks of 512 (gemm_pre2_k / polar_dual_k, MC = true).
int polar_project7(hipStream_t st, size_t n, float *packed, int has_scale, float scale, const Work &k, const int *stop,
                   int nb, size_t ws, ptrdiff_t ps, float *rx, ptrdiff_t rps)
{
    const int ni = (int)n, ld = (int)np_of(n), pitch = (int)pitch_of((size_t)ld);
    const size_t tot = (size_t)pitch * ld;
    const unsigned g = grid_for(tot, BLK, 512);
    float *M = k.G, *S = k.S, *Y = k.Y, *U = k.Z, *V = k.V;
    hipLaunchKernelGGL(unpack_k, dim3(g, 1, nb), dim3(BLK), 0, st, ni, ld, packed, has_scale, scale, M, (float *)nullptr,
                       k.part, stop, ws, ps, pitch);
    hipLaunchKernelGGL(scale_by_fro_k, dim3(g, 1, nb), dim3(BLK), 0, st, tot, M, k.part, (int)g, S, stop, ws, 1.85f);
    // r0, q0, q1, q2 of q(y) = (y - r0)(q0 y^2 + q1 y + q2)      (tools/polar_coeffs.py)
    static const float LIFT7[4] = { 3.446387665f, -0.824204593f, 2.299589116f, -1.666968041f };
    static const float TAIL7[3][4] = {
        { 3.563184240f, -0.454986096f, 1.283810123f, -1.135084021f },   // -> [0.58562779, 1.41437225]
        { 2.790799048f, -0.331962123f, 0.657669174f, -0.874710814f },   // -> [0.98052265, 1.01947735]
        { 2.503539308f, -0.323241073f, 0.536661220f, -0.878517037f },   // -> [0.99999922, 1.00000079]
    };
    DualArgs d;
    memset(&d, 0, sizeof(d));
    d.n = ni; d.ld = ld; d.pitch = pitch; d.stop = stop; d.ws = ws;
    // one symmetric product O = A B (dsym: A != B): the 32 x 64 block kernel, whose second operand is one dwordx2 per lane
    // (8.8-9.0 us per batched launch at k = 500; the two-product kernel's one-product form: 9.5)
    auto prod1 = [&](const float *A_, const float *B_, float *O_, int dsym) -> int {
        return gemm(st, false, ni, ld, 1.0f, A_, B_, 0.0f, nullptr, 0.0f, O_, stop, nb, ws, pitch, dsym);
    };
    for (int it = 0; it < 11; ++it) {
        const float *c = it < 8 ? LIFT7 : TAIL7[it - 8];
        THIP_RC(prod1(S, S, Y, 0));                                                                        // Y = S S
        d.nprod = 2; d.A = Y;
        d.B[0] = Y; d.O[0] = U; d.alpha[0] = c[1]; d.beta[0] = c[2]; d.gamma[0] = c[3]; d.dsym[0] = 0;    // U = q0 Y Y + q1 Y + q2 I
        d.B[1] = S; d.O[1] = V; d.alpha[1] = 1.0f; d.beta[1] = -c[0]; d.gamma[1] = 0.0f; d.dsym[1] = 1;   // V = Y S - r0 S
        THIP_RC(dual(st, d, nb));
        THIP_RC(prod1(U, V, S, 1));                                                                        // S <- U V
    }
    // Newton-Schulz x (3 - x^2) / 2 and M sign(M), merged:  T = 1.5 I - 0.5 S S (into Y),  R = S M (into U);  T R -> packed
    d.nprod = 2; d.A = S;
    d.B[0] = S; d.O[0] = Y; d.alpha[0] = -0.5f; d.beta[0] = 0.0f; d.gamma[0] = 1.5f; d.dsym[0] = 0;
    d.B[1] = M; d.O[1] = U; d.alpha[1] = 1.0f; d.beta[1] = 0.0f; d.gamma[1] = 0.0f; d.dsym[1] = 1;
    THIP_RC(dual(st, d, nb));
    d.nprod = 1; d.A = Y;
    d.B[0] = U; d.O[0] = nullptr; d.alpha[0] = 1.0f; d.beta[0] = 0.0f; d.gamma[0] = 0.0f; d.dsym[0] = 1;
    d.B[1] = nullptr; d.O[1] = nullptr;
    d.pack = packed; d.M = M; d.rx = rx; d.ps = ps; d.rps = rps; d.has_scale = has_scale; d.scale = scale;
    THIP_RC(dual(st, d, nb));
    return 0;
}

}  // namespace

namespace thip {

size_t psd_small_max()
{
    static const int small_on = getenv("THIP_POLAR_SMALL") ? atoi(getenv("THIP_POLAR_SMALL")) : 1;
    return small_on ? (size_t)PSN : 0;
}

int eig_psd_project_small(hipStream_t st, size_t n, float *base, const int64_t *dev_offs, int count, int has_scale,
                          float scale_diag, const int *stop, int nbatch, ptrdiff_t pstride, float *rx, ptrdiff_t rx_stride)
{
    if (n == 0 || count <= 0 || nbatch <= 0) return 0;
    if (n > (size_t)PSN) return fail(THIP_E_INVALID, "eig_psd_project_small: order above 64", __FILE__, __LINE__);
    // NK = MFMA steps per product: ceil(n / 2) rounded up to a multiple of 4
#define THIP_PS(NK) hipLaunchKernelGGL(polar_small_k<NK>, dim3(count, nbatch), dim3(256), 0, st, (int)n, base, has_scale, scale_diag, stop, pstride, dev_offs, rx, rx_stride)
    switch (((int)n + 7) / 8) {
    case 0: case 1: THIP_PS(4); break;
    case 2: THIP_PS(8); break;
    case 3: THIP_PS(12); break;
    case 4: THIP_PS(16); break;
    case 5: THIP_PS(20); break;
    case 6: THIP_PS(24); break;
    case 7: THIP_PS(28); break;
    default: THIP_PS(32); break;
    }
#undef THIP_PS
    THIP_LAUNCH_CHECK();
    return 0;
}

bool psd_project_takes_rx(size_t n)
{
    return n <= psd_small_max() || n > (size_t)POLAR_MIN_N;
}

int eig_psd_project(hipStream_t st, size_t n, float *packed, int has_scale, float scale_diag, float eps_zero,
                    float *work, size_t worklen, int map_kind, const int *stop, int nbatch, ptrdiff_t pstride,
                    float *rx, ptrdiff_t rx_stride)
{
    (void)eps_zero;   // F32CUDA ignores eps_zero as well (f32cuda.rs:196); f32 round-off is the floor
    if (n == 0 || nbatch <= 0) return 0;
    const size_t ws = thip_map_eig_worklen(n);
    if (worklen < ws * (size_t)nbatch) return fail(THIP_E_WORK, "map_eig work too short", __FILE__, __LINE__);
    const Work k = carve(work, n);
    // THIP_POLAR_SMALL=0: the chain of launches at every order; THIP_POLAR_SMALL_MIN=n: lowest order of the one-workgroup form
    static const int small_on = getenv("THIP_POLAR_SMALL") ? atoi(getenv("THIP_POLAR_SMALL")) : 1;
    static const int small_min = getenv("THIP_POLAR_SMALL_MIN") ? atoi(getenv("THIP_POLAR_SMALL_MIN")) : POLAR_SMALL_MIN_N;
    if (map_kind == 0 && small_on && n <= PSN && (int)n >= small_min) {
        return eig_psd_project_small(st, n, packed, nullptr, 1, has_scale, scale_diag, stop, nbatch, pstride, rx, rx_stride);
    }
    if (map_kind == 0 && n > POLAR_MIN_N) {
        if (pitch_of(np_of(n)) % 4 != 0) return fail(THIP_E_INVALID, "THIP_PSD_PITCH_PAD must be a multiple of 4 floats", __FILE__, __LINE__);
        return polar_project7(st, n, packed, has_scale, scale_diag, k, stop, nbatch, ws, pstride, rx, rx_stride);
    }
    for (int z = 0; z < nbatch; ++z) {          // the Jacobi engine works on one matrix at a time
        THIP_RC(decompose(st, n, packed + z * pstride, has_scale, scale_diag, k, map_kind, stop));
        THIP_RC(rebuild(st, n, packed + z * pstride, has_scale, scale_diag, k, stop));
    }
    return 0;
}

}  // namespace thip

extern "C" {

// test entry point: C = alpha A B + beta D + gamma I_n, A symmetric, B arbitrary, ld x ld (ld % 64 == 0)
int thip_test_gemm_sym(int n, int ld, float alpha, const float *A, const float *B, float beta, const float *D, float gamma, float *C)
{
    THIP_NEED_INIT();
    // gen form: C = alpha A B + beta D + gamma I with A symmetric, B arbitrary
    return gemm(ctx().stream, true, n, ld, alpha, A, B, beta, D, gamma, C, nullptr);
}

int thip_test_gemm_chain(int shape, int kernel, int n, int ld, int nb, float alpha, const float *X, const float *Y,
                         float beta, const float *D, float gamma, float *C)
{
    THIP_NEED_INIT();
    if (ld <= 0 || ld % 64 != 0 || ld > 512 || n < 0 || n > ld || nb < 1 || kernel < 0 || kernel > 2 || (shape != 0 && shape != 1))
        return fail(THIP_E_INVALID, "thip_test_gemm_chain: ld a multiple of 64 up to 512, n <= ld, nb >= 1", __FILE__, __LINE__);
    g_force_kernel = kernel;
    const int rc = gemm(ctx().stream, shape == 1, n, ld, alpha, X, Y, beta, D, gamma, C, nullptr, nb, (size_t)ld * ld);
    g_force_kernel = 0;
    return rc;
}


// timing probe (tools/psd_chain_probe.py): `reps` DEPENDENT launches of one product shape, microseconds per launch.
//   mode 0 / 1: the 32 x 64 block kernel on a batch of two (symmetric / general result) -- the round-4 chain's launches;
//   mode 2: one tile per workgroup, symmetric, batch of two (272 workgroups at ld = 512); mode 3: the same, one item;
//   mode 4: mode 3 on TWO streams at once (one item each; us per launch PAIR); mode 5: one tile, general, one item
// test entry point for the round-5 kernels of the chain: O_p = alpha_p A B_p + beta_p B_p + gamma_p I_n from the lower triangle
// of tiles (mirrored; diagonal tiles of a product with dsym_p averaged with their transpose), nb items ld * ld apart.
// coef = { alpha0, beta0, gamma0, dsym0, alpha1, beta1, gamma1, dsym1 }; B1 == nullptr: one product.  kernel 0: polar_dual_k with
// the library's NT; 1 .. 3: NT forced; 4 / 5: the one-tile / 32 x 64 block kernel of gemm() (one product, beta = 0)
int thip_test_gemm_dual(int kernel, int n, int ld, int nb, const float *A, const float *B0, const float *B1, const float *coef,
                        float *O0, float *O1)
{
    THIP_NEED_INIT();
    if (ld <= 0 || ld % 64 != 0 || ld > 512 || n < 0 || n > ld || nb < 1 || kernel < 0 || kernel > 5 || !A || !B0 || !coef || !O0)
        return fail(THIP_E_INVALID, "thip_test_gemm_dual", __FILE__, __LINE__);
    const size_t ws = (size_t)ld * ld;
    if (kernel == 4 || kernel == 5) {
        if (B1 != nullptr || coef[1] != 0.0f) return fail(THIP_E_INVALID, "thip_test_gemm_dual: kernels 4, 5 take one product, beta = 0", __FILE__, __LINE__);
        g_force_kernel = kernel == 4 ? 1 : 2;
        const int rc = gemm(ctx().stream, false, n, ld, coef[0], A, B0, 0.0f, nullptr, coef[2], O0, nullptr, nb, ws, ld, coef[3] != 0.0f);
        g_force_kernel = 0;
        return rc;
    }
    DualArgs d;
    memset(&d, 0, sizeof(d));
    d.n = n; d.ld = ld; d.pitch = ld; d.ws = ws; d.A = A; d.stop = ctx().never_stop;
    d.nprod = B1 ? 2 : 1;
    d.B[0] = B0; d.O[0] = O0; d.alpha[0] = coef[0]; d.beta[0] = coef[1]; d.gamma[0] = coef[2]; d.dsym[0] = coef[3] != 0.0f;
    if (B1) { d.B[1] = B1; d.O[1] = O1; d.alpha[1] = coef[4]; d.beta[1] = coef[5]; d.gamma[1] = coef[6]; d.dsym[1] = coef[7] != 0.0f; }
    if (kernel == 0) return dual(ctx().stream, d, nb);
    return kernel == 1 ? launch_dual<1>(ctx().stream, d, nb) : kernel == 2 ? launch_dual<2>(ctx().stream, d, nb) : launch_dual<3>(ctx().stream, d, nb);
}

__global__ void probe_delay_k(long long cycles)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
}

int thip_test_chain_probe(int mode, int ld, int reps, float *host_us)
{
    THIP_NEED_INIT();
    if (ld <= 0 || ld % 64 != 0 || ld > 512 || reps < 1 || mode < 0 || mode > 9)
        return fail(THIP_E_INVALID, "thip_test_chain_probe", __FILE__, __LINE__);
    Ctx &c = ctx();
    const size_t ws = (size_t)ld * ld;
    float *buf = nullptr;
    THIP_TRY(hipMalloc(&buf, sizeof(float) * ws * 6));
    THIP_TRY(hipMemsetAsync(buf, 0, sizeof(float) * ws * 6, c.stream));
    if (c.eig_side == nullptr) THIP_TRY(hipStreamCreateWithFlags(&c.eig_side, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k)
        if (c.eig_ev[k] == nullptr) THIP_TRY(hipEventCreateWithFlags(&c.eig_ev[k], hipEventDisableTiming));
    hipEvent_t e0, e1;
    THIP_TRY(hipEventCreate(&e0));
    THIP_TRY(hipEventCreate(&e1));
    const bool gen = mode == 1 || mode == 5;
    const int nb = (mode <= 2 || mode >= 6) ? 2 : 1;
    float *pk = nullptr;
    if (mode == 8) { THIP_TRY(hipMalloc(&pk, sizeof(float) * ws * 4)); THIP_TRY(hipMemsetAsync(pk, 0, sizeof(float) * ws * 4, c.stream)); }
    int rc = 0;
    for (int pass = 0; pass < 2 && rc == 0; ++pass) {          // pass 0 warms up
        // the host enqueues behind a 100 MHz-clock delay so that the chain's time is the device's, not the launch rate's
        hipLaunchKernelGGL(probe_delay_k, dim3(1), dim3(64), 0, c.stream, (long long)(100 * (12 * reps + 500)));
        hipEventRecord(e0, c.stream);
        if (mode == 4) { hipEventRecord(c.eig_ev[0], c.stream); hipStreamWaitEvent(c.eig_side, c.eig_ev[0], 0); }
        for (int r = 0; r < reps && rc == 0; ++r) {
            // items 0 and 1 live 3 ws apart; a launch reads buffer r % 3 and writes (r + 1) % 3 of its item(s)
            float *X = buf + (size_t)(r % 3) * ws, *C = buf + (size_t)((r + 1) % 3) * ws;
            if (mode >= 6) {
                // 6: the degree-7 step's middle launch (two products sharing A); 7: one product, diagonal tiles averaged;
                // 8: the last launch (one product, packed output + rx); 9: mode 7 through the 32 x 64 block kernel
                DualArgs d;
                memset(&d, 0, sizeof(d));
                d.n = ld; d.ld = ld; d.pitch = ld; d.ws = 3 * ws; d.A = X;
                d.B[0] = X; d.O[0] = C; d.alpha[0] = 0.5f; d.beta[0] = mode == 6 ? 0.1f : 0.0f; d.gamma[0] = 0.25f;
                d.nprod = mode == 6 ? 2 : 1;
                if (mode == 6) { d.B[1] = buf + (size_t)((r + 2) % 3) * ws; d.O[1] = d.B[1] == X ? C : const_cast<float *>(d.B[1]); d.alpha[1] = 0.5f; d.beta[1] = 0.1f; d.dsym[1] = 1; }
                if (mode == 7) d.dsym[0] = 1;
                if (mode == 8) { d.dsym[0] = 1; d.pack = pk; d.M = X; d.rx = pk + 2 * ws; d.ps = (ptrdiff_t)ws; d.rps = (ptrdiff_t)ws; d.has_scale = 1; d.scale = 1.414f; }
                if (mode == 9) rc = gemm(c.stream, false, ld, ld, 0.5f, X, X, 0.0f, nullptr, 0.25f, C, nullptr, nb, 3 * ws, ld, 1);
                else rc = dual(c.stream, d, nb);
                continue;
            }
            g_force_kernel = mode <= 1 ? 2 : 1;
            rc = gemm(c.stream, gen, ld, ld, 0.5f, X, X, 0.0f, nullptr, 0.25f, C, nullptr, nb, 3 * ws);
            if (mode == 4 && rc == 0)
                rc = gemm(c.eig_side, gen, ld, ld, 0.5f, X + 3 * ws, X + 3 * ws, 0.0f, nullptr, 0.25f, C + 3 * ws, nullptr, 1, 3 * ws);
            g_force_kernel = 0;
        }
        if (mode == 4) { hipEventRecord(c.eig_ev[1], c.eig_side); hipStreamWaitEvent(c.stream, c.eig_ev[1], 0); }
        hipEventRecord(e1, c.stream);
        THIP_TRY(hipEventSynchronize(e1));
    }
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    *host_us = 1e3f * ms / (float)reps;
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(buf);
    if (pk) hipFree(pk);
    return rc;
}

size_t thip_map_eig_worklen(size_t n)
{
    const size_t ld = np_of(n);
    return 5 * pitch_of(ld) * ld + 2 * ld + 16 + 512 + 64;
}

int thip_map_eig(size_t n, float *mat, int has_scale, float scale_diag, float eps_zero, float *work, size_t worklen,
                 int map_kind)
{
    THIP_NEED_INIT();
    if (map_kind != 0 && map_kind != 1) return fail(THIP_E_INVALID, "map_kind", __FILE__, __LINE__);
    return eig_psd_project(ctx().stream, n, mat, has_scale, scale_diag, eps_zero, work, worklen, map_kind, nullptr);
}

int thip_eig_decompose(size_t n, float *mat, int has_scale, float scale_diag, float eps_zero, float *work,
                       size_t worklen, float *host_w)
{
    THIP_NEED_INIT();
    (void)eps_zero;
    if (n == 0) return 0;
    if (worklen < thip_map_eig_worklen(n)) return fail(THIP_E_WORK, "map_eig work too short", __FILE__, __LINE__);
    const Work k = carve(work, n);
    THIP_RC(decompose(ctx().stream, n, mat, has_scale, scale_diag, k, -1, nullptr));
    return thip_d2h(host_w, k.w, n);
}

int thip_eig_rebuild(size_t n, float *mat, int has_scale, float scale_diag, float *work, size_t worklen,
                     const float *host_e, const uint8_t *host_keep)
{
    THIP_NEED_INIT();
    if (n == 0) return 0;
    if (worklen < thip_map_eig_worklen(n)) return fail(THIP_E_WORK, "map_eig work too short", __FILE__, __LINE__);
    const Work k = carve(work, n);
    // None -> contributes nothing; a kept value of exactly 0 contributes nothing either
    float *tmp = (float *)malloc(sizeof(float) * n);
    for (size_t i = 0; i < n; ++i) tmp[i] = host_keep[i] ? host_e[i] : 0.0f;
    const int rc = thip_h2d(k.e, tmp, n);
    free(tmp);
    THIP_RC(rc);
    return rebuild(ctx().stream, n, mat, has_scale, scale_diag, k, nullptr);
}

int thip_eig_engine_info(int *host_engine, int *host_polish, float *host_cert)
{
    THIP_NEED_INIT_NOFLUSH();
    if (host_engine) *host_engine = g_eig_engine;
    if (host_polish) *host_polish = g_eig_polish;
    if (host_cert) { host_cert[0] = g_eig_orth; host_cert[1] = g_eig_resid; host_cert[2] = (float)g_tri_persist; }
    return 0;
}

int thip_test_eig_force(int engine)
{
    THIP_NEED_INIT_NOFLUSH();
    if (engine < 0 || (engine & 3) > 2 || engine > 31) return fail(THIP_E_INVALID, "thip_test_eig_force: 0, 1 or 2, + 4, + 8 or + 12, + 16", __FILE__, __LINE__);
    g_eig_force = engine & 3;
    g_tri_force = (engine >> 2) & 3;
    g_tri_sabotage = (engine >> 4) & 1;
    g_persist_broken = 0;
    return 0;
}

int thip_proj_psd(size_t sn, float *x, float eps_zero, float *work, size_t worklen)
{
    THIP_NEED_INIT();
    const size_t n = (size_t)((std::sqrt((double)(8 * sn + 1)) - 1.0) / 2.0 + 0.5);
    if (n * (n + 1) / 2 != sn) return fail(THIP_E_INVALID, "not a triangular number", __FILE__, __LINE__);
    if (worklen < thip_map_eig_worklen(n)) return fail(THIP_E_WORK, "ConePSD work shortage", __FILE__, __LINE__);
    return eig_psd_project(ctx().stream, n, x, 1, std::sqrt(2.0f), eps_zero, work, worklen, 0, nullptr);
}

}  // extern "C"
