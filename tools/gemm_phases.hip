// gemm_phases.hip -- timestamps inside the 512^3 f32 GEMM of the PSD chain (4 waves per workgroup, all loads up front):
// per workgroup the times of kernel entry, "all operand loads have landed", "MFMAs done" and "stored", taken with
// s_memrealtime (100 MHz, one clock for the whole device) by wave 0 of ONE launch in the middle of a dependent chain of such launches.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/gemm_phases.hip -o /tmp/gemm_phases && /tmp/gemm_phases
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int LD = 512, NW = 4, KW = LD / NW, NQ = KW / 8;

template <int CHAINS>
__global__ __launch_bounds__(NW * 64) void k(const float *__restrict__ X, const float *__restrict__ Y, float *C, unsigned long long *ts, const int ld)
{
    const unsigned long long t0 = wall_clock64();
    X += (size_t)blockIdx.z * LD * ld; Y += (size_t)blockIdx.z * LD * ld; C += (size_t)blockIdx.z * LD * ld;
    __shared__ float red[NW - 1][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int kb = wave * KW, h = lane >> 5, li = lane & 31;
    const float *pa = X + (size_t)(kb + 4 * h) * ld + i0 + li;
    const float *pb = Y + (size_t)(kb + 4 * h) * ld + j0 + li;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    f32x4_t av[NQ], bv[NQ];
    if (CHAINS >= 4) {
        // both operands ALONG k (the stored transposes): one dwordx4 per lane per 8 k, 2 x NQ loads per wave
        const float *qa = X + (size_t)(i0 + li) * ld + kb + 4 * h, *qb = Y + (size_t)(j0 + li) * ld + kb + 4 * h;
#pragma unroll
        for (int q = 0; q < NQ; ++q) { av[q] = *reinterpret_cast<const f32x4_t *>(qa + 8 * q); bv[q] = *reinterpret_cast<const f32x4_t *>(qb + 8 * q); }
    } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) { av[q][t] = pa[(size_t)(8 * q + t) * ld]; bv[q][t] = pb[(size_t)(8 * q + t) * ld]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (CHAINS != 3 && CHAINS != 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // CHAINS == 3: the MFMAs start as their operands land
    const unsigned long long t1 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.0f;
    if (CHAINS != 2) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][t], bv[q][t], acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int t = 0; t < 4; t += 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][t], bv[q][t], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][t + 1], bv[q][t + 1], acc2, 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float keep = acc[0];
    asm volatile("" : "+v"(keep));
    const unsigned long long t2 = wall_clock64();
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r];
#pragma unroll
            for (int w = 0; w < NW - 1; ++w) v += red[w][r][lane];
            C[(size_t)(i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * ld + j0 + (lane & 31)] = v * 1e-3f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t3 = wall_clock64();
        if (ts != nullptr && lane == 0) {
            const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            ts[4 * b] = t0; ts[4 * b + 1] = t1; ts[4 * b + 2] = t2; ts[4 * b + 3] = t3;
        }
    }
}

// two tiles per workgroup sharing operand a (gemm_pre2_k of thip_eig.hip): stamps t0 entry, t1 first tile's MFMAs done,
// t2 second tile's MFMAs done, t3 stored.  grid (16, 8, nz)
template <int WAITALL>
__global__ __launch_bounds__(256) void kpair(const float *__restrict__ X, const float *__restrict__ Y, float *C, unsigned long long *ts, const int ld)
{
    const unsigned long long t0 = wall_clock64();
    X += (size_t)blockIdx.z * LD * ld; Y += (size_t)blockIdx.z * LD * ld; C += (size_t)blockIdx.z * LD * ld;
    __shared__ float red[NW][2][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 64;
    const int kb = wave * KW, h = lane >> 5, li = lane & 31;
    const float *pa = X + (size_t)(kb + 4 * h) * ld + i0 + li;
    const float *pb = Y + (size_t)(kb + 4 * h) * ld + j0 + li;
    float av[NQ][4], b0[NQ][4], b1[NQ][4];
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
    if (WAITALL == 2) {
        // software pipeline: DEP slabs of 8 k (64 loads, what a wave may have in flight) ahead of the MFMAs
        constexpr int DEP = 8;
#pragma unroll
        for (int q = 0; q < DEP; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) { av[q][t] = pa[(size_t)(8 * q + t) * ld]; b0[q][t] = pb[(size_t)(8 * q + t) * ld]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][t], b0[q][t], acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q + DEP < NQ) {
#pragma unroll
                for (int t = 0; t < 4; ++t) { av[q + DEP][t] = pa[(size_t)(8 * (q + DEP) + t) * ld]; b0[q + DEP][t] = pb[(size_t)(8 * (q + DEP) + t) * ld]; }
            } else {
                const int q1 = 2 * (q + DEP - NQ);
#pragma unroll
                for (int t = 0; t < 4; ++t) { b1[q1][t] = pb[(size_t)(8 * q1 + t) * ld + 32]; b1[q1 + 1][t] = pb[(size_t)(8 * (q1 + 1) + t) * ld + 32]; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) { av[q][t] = pa[(size_t)(8 * q + t) * ld]; b0[q][t] = pb[(size_t)(8 * q + t) * ld]; }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) b1[q][t] = pb[(size_t)(8 * q + t) * ld + 32];
    __builtin_amdgcn_sched_barrier(0);
    if (WAITALL == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][t], b0[q][t], acc0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][0][r][lane] = acc0[r];
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][t], b1[q][t], acc1, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][1][r][lane] = acc1[r];
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t2 = wall_clock64();
    __syncthreads();
    const int tt = wave & 1, hh = wave >> 1;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int r = 8 * hh + rr;
        float v = red[0][tt][r][lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[w][tt][r][lane];
        C[(size_t)(i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * ld + j0 + 32 * tt + (lane & 31)] = v * 1e-3f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = wall_clock64();
    if (ts != nullptr && tid == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        ts[4 * b] = t0; ts[4 * b + 1] = t1; ts[4 * b + 2] = t2; ts[4 * b + 3] = t3;
    }
}

// the 32 x 64 block kernel as shipped (gemm_pre2_k of thip_eig.hip, symmetric-operand shape, full grid): even / odd column
// accumulators, operand b as dwordx2, loads DEP slabs ahead.  Stamps: t0 entry, t1 MFMAs done, t2 partial sums in LDS and
// workgroup barrier passed, t3 stored.  grid (16, 8, nz)
template <int DEP>
__global__ __launch_bounds__(256) void kblock(const float *__restrict__ X, const float *__restrict__ Y, float *C, unsigned long long *ts, const int ld)
{
    const unsigned long long t0 = wall_clock64();
    X += (size_t)blockIdx.z * LD * ld; Y += (size_t)blockIdx.z * LD * ld; C += (size_t)blockIdx.z * LD * ld;
    __shared__ float red[NW][2][16][64];
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 64;
    const int kb = wave * KW, h = lane >> 5, li = lane & 31;
    const float *pa = X + (size_t)(kb + 4 * h) * ld + i0 + li;
    const float *pb = Y + (size_t)(kb + 4 * h) * ld + j0 + 2 * li;
    float av[NQ][4];
    f32x2_t bv[NQ][4];
    auto load = [&](const int q) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { av[q][t] = pa[(size_t)(8 * q + t) * ld]; bv[q][t] = *reinterpret_cast<const f32x2_t *>(pb + (size_t)(8 * q + t) * ld); }
    };
#pragma unroll
    for (int q = 0; q < DEP; ++q) load(q);
    f32x16 acce, acco;
#pragma unroll
    for (int r = 0; r < 16; ++r) acce[r] = acco[r] = 0.0f;
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
#pragma unroll
    for (int r = 0; r < 16; ++r) { red[wave][0][r][lane] = acce[r]; red[wave][1][r][lane] = acco[r]; }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = wall_clock64();
    __syncthreads();
    const unsigned long long t2 = wall_clock64();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = 4 * wave + rr;
        float ve = red[0][0][r][lane], vo = red[0][1][r][lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) { ve += red[w][0][r][lane]; vo += red[w][1][r][lane]; }
        f32x2_t v2;
        v2[0] = ve * 1e-3f; v2[1] = vo * 1e-3f;
        *reinterpret_cast<f32x2_t *>(C + (size_t)(i0 + rr + 8 * wave + 4 * h) * ld + j0 + 2 * li) = v2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = wall_clock64();
    if (ts != nullptr && tid == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        ts[4 * b] = t0; ts[4 * b + 1] = t1; ts[4 * b + 2] = t2; ts[4 * b + 3] = t3;
    }
}

// the same with NWV waves splitting K (8: two waves per SIMD)
// the 32 x 64 block kernel as shipped (gemm_pre2_k of thip_eig.hip, symmetric-operand shape, full grid): even / odd column
// accumulators, operand b as dwordx2, loads DEP slabs ahead.  Stamps: t0 entry, t1 MFMAs done, t2 partial sums in LDS and
// workgroup barrier passed, t3 stored.  grid (16, 8, nz)
template <int DEP, int NWV>
__global__ __launch_bounds__(NWV * 64) void kblockw(const float *__restrict__ X, const float *__restrict__ Y, float *C, unsigned long long *ts, const int ld)
{
    const unsigned long long t0 = wall_clock64();
    X += (size_t)blockIdx.z * LD * ld; Y += (size_t)blockIdx.z * LD * ld; C += (size_t)blockIdx.z * LD * ld;
    constexpr int KWv = LD / NWV, NQv = KWv / 8;
    __shared__ float red[NWV][2][16][64];
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 64;
    const int kb = wave * KWv, h = lane >> 5, li = lane & 31;
    const float *pa = X + (size_t)(kb + 4 * h) * ld + i0 + li;
    const float *pb = Y + (size_t)(kb + 4 * h) * ld + j0 + 2 * li;
    float av[NQv][4];
    f32x2_t bv[NQv][4];
    auto load = [&](const int q) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { av[q][t] = pa[(size_t)(8 * q + t) * ld]; bv[q][t] = *reinterpret_cast<const f32x2_t *>(pb + (size_t)(8 * q + t) * ld); }
    };
#pragma unroll
    for (int q = 0; q < DEP; ++q) load(q);
    f32x16 acce, acco;
#pragma unroll
    for (int r = 0; r < 16; ++r) acce[r] = acco[r] = 0.0f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQv; ++q) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acce = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][t], bv[q][t][0], acce, 0, 0, 0);
            acco = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][t], bv[q][t][1], acco, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (q + DEP < NQv) load(q + DEP);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { red[wave][0][r][lane] = acce[r]; red[wave][1][r][lane] = acco[r]; }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = wall_clock64();
    __syncthreads();
    const unsigned long long t2 = wall_clock64();
#pragma unroll
    for (int rr = 0; rr < 16 / NWV; ++rr) {
        const int r = (16 / NWV) * wave + rr;
        float ve = red[0][0][r][lane], vo = red[0][1][r][lane];
#pragma unroll
        for (int w = 1; w < NWV; ++w) { ve += red[w][0][r][lane]; vo += red[w][1][r][lane]; }
        f32x2_t v2;
        v2[0] = ve * 1e-3f; v2[1] = vo * 1e-3f;
        *reinterpret_cast<f32x2_t *>(C + (size_t)(i0 + (r & 3) + 8 * (r >> 2) + 4 * h) * ld + j0 + 2 * li) = v2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = wall_clock64();
    if (ts != nullptr && tid == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        ts[4 * b] = t0; ts[4 * b + 1] = t1; ts[4 * b + 2] = t2; ts[4 * b + 3] = t3;
    }
}

int main()
{
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const size_t sq = (size_t)LD * 1024;
    float *A, *B, *Cc; unsigned long long *ts;
    hipMalloc(&A, 2 * sq * 4); hipMalloc(&B, 2 * sq * 4); hipMalloc(&Cc, 2 * sq * 4);
    std::vector<float> h(2 * sq, 1e-3f);
    hipMemcpy(A, h.data(), 2 * sq * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), 2 * sq * 4, hipMemcpyHostToDevice);
    hipMemcpy(Cc, h.data(), 2 * sq * 4, hipMemcpyHostToDevice);
    for (int ld : { 512 })
    for (int place = 0; place <= 0; ++place)        // 1: dynamic LDS sized so that exactly nz workgroups fit a CU
    for (int chains : { 10, 12, 13 })
    for (int nz = 2; nz <= 2; ++nz) {
        auto kern = chains == 1 ? k<1> : chains == 6 ? kpair<0> : chains == 7 ? kpair<1> : chains == 8 ? kpair<2> : chains == 9 ? kblock<8> : chains == 10 ? kblock<6> : chains == 11 ? kblock<16> : chains == 12 ? kblockw<8, 8> : kblockw<4, 8>;
        const bool pr = chains >= 6;
        const int nwg = (pr ? 128 : 256) * nz;
        const size_t dyn = place ? (nz == 1 ? 100 : 48) * 1024 : 0;
        hipMalloc(&ts, (size_t)3 * nwg * 4 * 8);
        float *bufs[3] = { A, B, Cc };
        hipEvent_t e0, ev1; hipEventCreate(&e0); hipEventCreate(&ev1);
        const int reps = 100;
        for (int warm = 0; warm < 200; ++warm)
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(16, pr ? 8 : 16, nz), dim3((chains >= 12 ? 8 : NW) * 64), dyn, st, bufs[i % 3], bufs[(i + 1) % 3], bufs[(i + 2) % 3], nullptr, ld);
        hipEventRecord(e0, st);
        for (int i = 0; i < reps; ++i) {
            unsigned long long *t = (i >= 50 && i < 53) ? ts + (size_t)(i - 50) * nwg * 4 : nullptr;
            hipLaunchKernelGGL(kern, dim3(16, pr ? 8 : 16, nz), dim3((chains >= 12 ? 8 : NW) * 64), dyn, st, bufs[i % 3], bufs[(i + 1) % 3], bufs[(i + 2) % 3], t, ld);
        }
        hipEventRecord(ev1, st); hipEventSynchronize(ev1);
        float ms; hipEventElapsedTime(&ms, e0, ev1);
        std::vector<unsigned long long> hts((size_t)3 * nwg * 4);
        hipMemcpy(hts.data(), ts, hts.size() * 8, hipMemcpyDeviceToHost);
        const double us_per_launch = 1e3 * ms / reps, tpu = 100.0;
        auto T = [&](int L, int b, int w) { return hts[((size_t)L * nwg + b) * 4 + w]; };
        unsigned long long s1 = ~0ull, smax = 0, e1 = 0, s2 = ~0ull, s0 = ~0ull;
        double d_load = 0, d_mfma = 0, d_store = 0;
        for (int b = 0; b < nwg; ++b) {
            s0 = std::min(s0, T(0, b, 0));
            s1 = std::min(s1, T(1, b, 0)); smax = std::max(smax, T(1, b, 0)); e1 = std::max(e1, T(1, b, 3)); s2 = std::min(s2, T(2, b, 0));
            d_load += (double)(T(1, b, 1) - T(1, b, 0)); d_mfma += (double)(T(1, b, 2) - T(1, b, 1)); d_store += (double)(T(1, b, 3) - T(1, b, 2));
        }
        printf("ld %d, placed %d, %d chain(s), z = %d: %.2f us per launch (events), %.2f by the device clock | per workgroup avg: loads %.2f us, mfma %.2f us, "
               "reduce+store %.2f us | first start -> last start %.2f us, first start -> last store %.2f us, "
               "last store -> next launch's first start %.2f us\n",
               ld, place, chains, nz, us_per_launch, (double)(s2 - s0) / 2 / tpu, d_load / nwg / tpu, d_mfma / nwg / tpu, d_store / nwg / tpu,
               (double)(smax - s1) / tpu, (double)(e1 - s1) / tpu, (double)(long long)(s2 - e1) / tpu);
        {
            const char *names[5] = { "start", "phase 1", "phase 2", "phase 3", "end" };
            for (int w = 0; w < 5; ++w) {
                std::vector<double> v;
                for (int b = 0; b < nwg; ++b)
                    v.push_back(w == 0 ? (double)(T(1, b, 0) - s1) / tpu : w == 4 ? (double)(T(1, b, 3) - s1) / tpu : (double)(T(1, b, w) - T(1, b, w - 1)) / tpu);
                std::sort(v.begin(), v.end());
                printf("    %-13s min %.2f  p10 %.2f  p25 %.2f  p50 %.2f  p75 %.2f  p90 %.2f  max %.2f\n", names[w], v[0], v[nwg / 10], v[nwg / 4],
                       v[nwg / 2], v[3 * nwg / 4], v[9 * nwg / 10], v[nwg - 1]);
            }
        }
        hipFree(ts);
    }
    return 0;
}
