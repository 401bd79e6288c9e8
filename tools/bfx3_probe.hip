// bfx3_probe.hip -- would the PSD chain's f32 products run faster on the bf16 matrix cores?  (NOTEBOOK.md 9.4)
//
// An f32 operand is split EXACTLY into three bf16 terms (hi + mid + lo: 3 x 8 significand bits), stored as three planes in
// the layout v_mfma_f32_32x32x16_bf16 reads (8 consecutive k per lane: plane[(k >> 3) * ld + r][k & 7], one coalesced 16-byte
// load per lane and 16 k), and a product is six of the nine cross products (the three dropped are below 2^-24 of |a||b|):
// 6 x 32 cycles per 16 k against 8 x 64 for v_mfma_f32_32x32x2_f32.  This probe times a DEPENDENT chain of symmetric
// ld x ld products of that form (lower triangle of 32 x 32 tiles, NT tile-jobs per workgroup sharing their first operand,
// four waves splitting K, result written back as three planes -- direct and mirrored -- by the epilogue) for a batch of two,
// and checks one product against f64.  Build: hipcc -O3 --offload-arch=gfx950 tools/bfx3_probe.hip -o tools/bfx3_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef unsigned short bfraw;

__host__ __device__ inline bfraw bf_rne(float x)
{
    unsigned u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bfraw)((u >> 16) | 0x40);      // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bfraw)(u >> 16);
}
__host__ __device__ inline float bf_f32(bfraw b)
{
    unsigned u = (unsigned)b << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}
// x == hi + mid + lo exactly (for x whose low terms do not underflow)
__host__ __device__ inline void split3(float x, bfraw &h, bfraw &m, bfraw &l)
{
    h = bf_rne(x);
    const float r1 = x - bf_f32(h);
    m = bf_rne(r1);
    const float r2 = r1 - bf_f32(m);
    l = bf_rne(r2);
}

// plane element (row r, k c) of an ld x ld matrix
__host__ __device__ inline size_t pidx(int ld, int r, int c) { return ((size_t)(c >> 3) * ld + r) * 8 + (c & 7); }

__global__ void to_planes_k(int ld, const float *__restrict__ X, bfraw *__restrict__ P, size_t plane)
{
    const size_t tot = (size_t)ld * ld;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i % ld), c = (int)(i / ld);
        bfraw h, m, l;
        split3(X[i], h, m, l);
        const size_t o = pidx(ld, r, c);
        P[o] = h; P[plane + o] = m; P[2 * plane + o] = l;
    }
}

__global__ void delay_k(long long cycles)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
}

struct Args {
    int ld, nprod;
    const bfraw *A;          // 3 planes, `plane` elements apart; items `item` elements apart
    const bfraw *B[2];
    bfraw *O[2];
    float *F;                // optional f32 output of product 0 (ld x ld per item), for the check
    size_t plane, item;
    float alpha;
    int gp, spx, xpi;
};

template <int KW, int NT>
__global__ __launch_bounds__(256) void bfx3_k(const Args a)
{
    constexpr int NW = 4, NQ = KW / 16;
    int bi = 0, gl, item;
    {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        item = xcd / a.xpi;
        gl = (xcd % a.xpi) * a.spx + slot;
        if (gl >= a.gp) return;
    }
    for (;;) {
        const int gi = (a.nprod * (bi + 1) + NT - 1) / NT;
        if (gl < gi) break;
        gl -= gi; ++bi;
    }
    const int njobs = a.nprod * (bi + 1);
    __shared__ float red[NW][NT][16][64];
    __shared__ float tr[NT][32][33];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, li = lane & 31;
    const int ld = a.ld, i0 = bi * 32, kb = wave * KW;
    int bj[NT], pr[NT];
    bool live[NT];
    const u32x4 *pb[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int t = NT * gl + u;
        live[u] = t < njobs;
        const int tt = live[u] ? t : NT * gl;
        bj[u] = a.nprod == 2 ? tt >> 1 : tt;
        pr[u] = a.nprod == 2 ? tt & 1 : 0;
        pb[u] = reinterpret_cast<const u32x4 *>(a.B[pr[u]] + item * a.item) + ((size_t)((kb >> 3) + h) * ld + bj[u] * 32 + li);
    }
    const u32x4 *pa = reinterpret_cast<const u32x4 *>(a.A + item * a.item) + ((size_t)((kb >> 3) + h) * ld + i0 + li);
    const size_t pl4 = a.plane / 8;                  // plane stride in 16-byte units
    const size_t kstep = (size_t)2 * ld;             // 16 k = two 8-blocks of ld rows
    // loads run DEP blocks of 16 k ahead of the MFMAs ((1 + NT) x 3 planes dwordx4 per block: 36-48 of the 64 a wave may have in flight)
    constexpr int DEP = NQ < 4 ? NQ : 4;
    u32x4 av[NQ][3], bv[NQ][NT][3];
    auto load = [&](const int q) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            av[q][p] = pa[q * kstep + p * pl4];
#pragma unroll
            for (int u = 0; u < NT; ++u) bv[q][u][p] = pb[u][q * kstep + p * pl4];
        }
    };
#pragma unroll
    for (int q = 0; q < DEP; ++q) load(q);
    f32x16 acc[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, av[q][0]), am = __builtin_bit_cast(bf16x8, av[q][1]), al = __builtin_bit_cast(bf16x8, av[q][2]);
            const bf16x8 bh = __builtin_bit_cast(bf16x8, bv[q][u][0]), bm = __builtin_bit_cast(bf16x8, bv[q][u][1]), bl = __builtin_bit_cast(bf16x8, bv[q][u][2]);
            // the small terms first
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[u], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (q + DEP < NQ) load(q + DEP);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][u][r][lane] = acc[u][r];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NT; ++u) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = 4 * wave + rr, tl = rr + 8 * wave + 4 * h;
            float v = red[0][u][r][lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) v += red[w][u][r][lane];
            tr[u][tl][li] = v * a.alpha;
        }
    }
    __syncthreads();
    // thread t: floats 4 (t % 8) .. of row t / 8 of the tile and of its mirror image: 8 bytes per plane each
    const int er = tid >> 3, ec = (tid & 7) * 4;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        if (!live[u]) continue;
        const int j0 = bj[u] * 32;
        const bool diag = bj[u] == bi;
        bfraw *O = a.O[pr[u]] + item * a.item;
        float v[4], w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = tr[u][er][ec + k]; w[k] = tr[u][ec + k][er]; }
        if (diag) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = 0.5f * (v[k] + w[k]);
        }
        if (a.F != nullptr && pr[u] == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a.F[item * (size_t)ld * ld + (size_t)(i0 + er) * ld + j0 + ec + k] = v[k];
                if (!diag) a.F[item * (size_t)ld * ld + (size_t)(j0 + er) * ld + i0 + ec + k] = w[k];
            }
        }
        // element (row r, k c): the tile's (i0 + er, j0 + ec + k) and the mirror's (j0 + er, i0 + ec + k)
        unsigned long long ph = 0, pm = 0, pl = 0, qh = 0, qm = 0, ql = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bfraw x, y, z;
            split3(v[k], x, y, z);
            ph |= (unsigned long long)x << (16 * k); pm |= (unsigned long long)y << (16 * k); pl |= (unsigned long long)z << (16 * k);
            split3(w[k], x, y, z);
            qh |= (unsigned long long)x << (16 * k); qm |= (unsigned long long)y << (16 * k); ql |= (unsigned long long)z << (16 * k);
        }
        const size_t o1 = pidx(ld, i0 + er, j0 + ec);
        *reinterpret_cast<unsigned long long *>(O + o1) = ph;
        *reinterpret_cast<unsigned long long *>(O + a.plane + o1) = pm;
        *reinterpret_cast<unsigned long long *>(O + 2 * a.plane + o1) = pl;
        if (!diag) {
            const size_t o2 = pidx(ld, j0 + er, i0 + ec);
            *reinterpret_cast<unsigned long long *>(O + o2) = qh;
            *reinterpret_cast<unsigned long long *>(O + a.plane + o2) = qm;
            *reinterpret_cast<unsigned long long *>(O + 2 * a.plane + o2) = ql;
        }
    }
}

static int groups(int nt, int nprod, int NT)
{
    int g = 0;
    for (int i = 0; i < nt; ++i) g += (nprod * (i + 1) + NT - 1) / NT;
    return g;
}

template <int NT>
static void launch(hipStream_t st, Args a, int nb)
{
    const int nt = a.ld / 32;
    a.gp = groups(nt, a.nprod, NT);
    a.xpi = 8 / nb;
    a.spx = (a.gp + a.xpi - 1) / a.xpi;
    hipLaunchKernelGGL((bfx3_k<128, NT>), dim3(8 * a.spx), dim3(256), 0, st, a);
}

int main(int argc, char **argv)
{
    const int ld = 512, nb = 2, reps = argc > 1 ? atoi(argv[1]) : 200;
    const size_t n2 = (size_t)ld * ld, plane = n2, item = 3 * plane;
    std::vector<float> X(n2 * nb), Y(n2 * nb);
    srand(1);
    for (int it = 0; it < nb; ++it)
        for (int c = 0; c < ld; ++c)
            for (int r = 0; r <= c; ++r) {
                const float x = ((float)rand() / RAND_MAX - 0.5f) * 0.09f, y = ((float)rand() / RAND_MAX - 0.5f) * 0.09f;
                X[it * n2 + (size_t)c * ld + r] = X[it * n2 + (size_t)r * ld + c] = x;
                Y[it * n2 + (size_t)c * ld + r] = Y[it * n2 + (size_t)r * ld + c] = y;
            }
    float *dX, *dY, *dF;
    bfraw *P[4];
    CK(hipMalloc(&dX, n2 * nb * 4)); CK(hipMalloc(&dY, n2 * nb * 4)); CK(hipMalloc(&dF, n2 * nb * 4));
    for (int k = 0; k < 4; ++k) CK(hipMalloc(&P[k], item * nb * sizeof(bfraw)));
    CK(hipMemcpy(dX, X.data(), n2 * nb * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dY, Y.data(), n2 * nb * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int it = 0; it < nb; ++it) {
        hipLaunchKernelGGL(to_planes_k, dim3(512), dim3(256), 0, st, ld, dX + it * n2, P[0] + it * item, plane);
        hipLaunchKernelGGL(to_planes_k, dim3(512), dim3(256), 0, st, ld, dY + it * n2, P[1] + it * item, plane);
    }
    // ---- one product X * Y checked against f64 (and the f32 result's planes read back: the split is exact)
    Args a;
    memset(&a, 0, sizeof(a));
    a.ld = ld; a.nprod = 1; a.A = P[0]; a.B[0] = P[1]; a.O[0] = P[2]; a.F = dF; a.plane = plane; a.item = item; a.alpha = 1.0f;
    launch<2>(st, a, nb);
    CK(hipStreamSynchronize(st));
    std::vector<float> F(n2 * nb);
    std::vector<bfraw> Pl(item * nb);
    CK(hipMemcpy(F.data(), dF, n2 * nb * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(Pl.data(), P[2], item * nb * sizeof(bfraw), hipMemcpyDeviceToHost));
    double worst = 0.0, worst32 = 0.0, worst_split = 0.0;
    for (int it = 0; it < nb; ++it)
        for (int i = 0; i < ld; i += 3)
            for (int j = 0; j <= i; j += 5) {
                double ref = 0.0, sc = 0.0;
                float f32acc = 0.0f;
                for (int k = 0; k < ld; ++k) {
                    const double x = X[it * n2 + (size_t)k * ld + i], y = Y[it * n2 + (size_t)k * ld + j];
                    ref += x * y; sc += fabs(x * y);
                    f32acc = fmaf((float)x, (float)y, f32acc);
                }
                double want = ref;
                if ((i >> 5) == (j >> 5)) {                 // diagonal tile: averaged with its transpose
                    double r2 = 0.0;
                    for (int k = 0; k < ld; ++k) r2 += (double)X[it * n2 + (size_t)k * ld + j] * Y[it * n2 + (size_t)k * ld + i];
                    want = 0.5 * (ref + r2);
                }
                const float got = F[it * n2 + (size_t)i * ld + j];
                worst = fmax(worst, fabs(got - want) / sc);
                if ((i >> 5) != (j >> 5)) worst32 = fmax(worst32, fabs((double)f32acc - ref) / sc);
                const size_t o = pidx(ld, i, j);
                const float back = bf_f32(Pl[it * item + o]) + bf_f32(Pl[it * item + plane + o]) + bf_f32(Pl[it * item + 2 * plane + o]);
                worst_split = fmax(worst_split, fabs((double)back - (double)got) / (fabs((double)got) + 1e-30));
            }
    printf("one product, 6 of 9 bf16 cross products: max |err| / sum|a||b| = %.2e   (a scalar f32 fma chain: %.2e);  planes vs f32 result: %.1e\n",
           worst, worst32, worst_split);
    // ---- dependent chains: NT = 2 one product (Y = S S) and NT = 3 two products sharing A
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    a.F = nullptr; a.alpha = 0.5f;
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int pass = 0; pass < 4; ++pass) {
            hipLaunchKernelGGL(delay_k, dim3(1), dim3(64), 0, st, (long long)(100 * (14 * reps + 500)));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) {
                a.A = P[r % 3]; a.O[0] = P[(r + 1) % 3];
                if (mode == 0) { a.nprod = 1; a.B[0] = a.A; launch<2>(st, a, nb); }
                else { a.nprod = 2; a.B[0] = a.A; a.B[1] = P[(r + 2) % 3]; a.O[1] = P[3]; launch<3>(st, a, nb); }
            }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (pass > 0) best = fminf(best, 1e3f * ms / reps);
        }
        printf("ld = 512, batch of two, %s: %.2f us per dependent launch (f32 kernels: %s)\n",
               mode == 0 ? "one symmetric product, 2 tile-jobs per workgroup" : "two products sharing A, 3 tile-jobs per workgroup", best,
               mode == 0 ? "8.8-9.0 gemm_pre2_k, 9.3 polar_dual_k" : "12.1 polar_dual_k");
    }
    return 0;
}
