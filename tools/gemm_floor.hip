// gemm_floor.hip -- where do the 8.4 us of one 512^3 f32 GEMM launch of the PSD chain go?  A chain of dependent launches
// (same stream, each reads what the previous wrote) of kernels that add one phase at a time:
//   K0 empty                                  K1 + the stop-flag load and one store per workgroup
//   K2 + all operand loads (no MFMA)          K3 + the MFMAs (every wave stores its own partial tile)
//   K4 + the 8-way split-K reduction through LDS (= gemm_pre_k<false, 64>)
// and the same with 4 waves / 256 threads per workgroup (K split 4) and with 128 workgroups of 64 x 32 tiles.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/gemm_floor.hip -o /tmp/gemm_floor && /tmp/gemm_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int LD = 512;

template <int PHASE, int NW>
__global__ __launch_bounds__(NW * 64) void k(const float *__restrict__ X, const float *__restrict__ Y, float *C, const int *stop)
{
    if (PHASE == 0) return;
    if (*stop != 0) return;
    X += (size_t)blockIdx.z * LD * LD; Y += (size_t)blockIdx.z * LD * LD; C += (size_t)blockIdx.z * LD * LD;
    __shared__ float red[NW - 1][16][64];
    constexpr int KW = LD / NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    if (PHASE == 1) { if (tid == 0) C[(size_t)i0 * LD + j0] = 1.0f; return; }
    if (PHASE == 6) {                    // the MFMAs alone: operands from registers, no operand loads
        f32x16 acc6;
        for (int r = 0; r < 16; ++r) acc6[r] = 0.0f;
        float a6 = 1.0f + tid * 1e-6f, b6 = 0.5f;
#pragma unroll
        for (int q = 0; q < LD / NW / 2; ++q) acc6 = __builtin_amdgcn_mfma_f32_32x32x2f32(a6, b6, acc6, 0, 0, 0);
        if (acc6[0] == 12345.678f) C[0] = acc6[1];
        return;
    }
    const int kb = wave * KW, h = lane >> 5, li = lane & 31;
    const float *pa = X + (size_t)(kb + 4 * h) * LD + i0 + li;
    const float *pb = Y + (size_t)(kb + 4 * h) * LD + j0 + li;
    constexpr int NQ = KW / 8;
    float av[NQ][4], bv[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) { av[q][t] = pa[(size_t)(8 * q + t) * LD]; bv[q][t] = pb[(size_t)(8 * q + t) * LD]; }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    if (PHASE == 5) {                    // all operand loads of every wave, kept alive by a store per wave; no MFMA
        float s5 = 0.0f;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) s5 += av[q][t] * bv[q][t];
        if (s5 == 12345.678f) C[(size_t)(i0 + li) * LD + j0 + wave] = s5;
        return;
    }
    if (PHASE == 2) {
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) s += av[q][t] * bv[q][t];
        if (wave == 0) C[(size_t)(i0 + (lane >> 1)) * LD + j0 + (lane & 1)] = s;
        return;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][t], bv[q][t], acc, 0, 0, 0);
    if (PHASE == 3) {
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) C[(size_t)(i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LD + j0 + (lane & 31)] = acc[r];
        } else if (acc[0] == 12345.678f) C[0] = acc[1];
        return;
    }
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
            C[(size_t)(i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LD + j0 + (lane & 31)] = v * 1e-3f;
        }
    }
}

template <int PHASE, int NW>
float chain(hipStream_t st, float *A, float *B, float *Cc, int *stop, int nz)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    float *bufs[3] = { A, B, Cc };
    for (int w = 0; w < 2; ++w) {
        if (w == 1) hipEventRecord(e0, st);
        for (int i = 0; i < reps; ++i)
            hipLaunchKernelGGL((k<PHASE, NW>), dim3(16, 16, nz), dim3(NW * 64), 0, st, bufs[i % 3], bufs[(i + 1) % 3], bufs[(i + 2) % 3], stop);
    }
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return 1e3f * ms / reps;
}

int main()
{
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    float *A, *B, *Cc; int *stop;
    const size_t sq = (size_t)LD * LD;
    hipMalloc(&A, 2 * sq * 4); hipMalloc(&B, 2 * sq * 4); hipMalloc(&Cc, 2 * sq * 4); hipMalloc(&stop, 4);
    hipMemset(stop, 0, 4);
    std::vector<float> h(2 * sq, 1e-3f);
    hipMemcpy(A, h.data(), 2 * sq * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), 2 * sq * 4, hipMemcpyHostToDevice);
    hipMemcpy(Cc, h.data(), 2 * sq * 4, hipMemcpyHostToDevice);
    printf("us per launch in a dependent chain, 16 x 16 tiles of 32 x 32, ld = 512 (grid z = 1: 256 workgroups)\n");
    printf("8 waves/WG:  empty %.2f  flag+store %.2f  +loads %.2f  +mfma %.2f  +lds-reduce %.2f\n",
           chain<0, 8>(st, A, B, Cc, stop, 1), chain<1, 8>(st, A, B, Cc, stop, 1), chain<2, 8>(st, A, B, Cc, stop, 1),
           chain<3, 8>(st, A, B, Cc, stop, 1), chain<4, 8>(st, A, B, Cc, stop, 1));
    printf("4 waves/WG:  empty %.2f  flag+store %.2f  +loads %.2f  +mfma %.2f  +lds-reduce %.2f\n",
           chain<0, 4>(st, A, B, Cc, stop, 1), chain<1, 4>(st, A, B, Cc, stop, 1), chain<2, 4>(st, A, B, Cc, stop, 1),
           chain<3, 4>(st, A, B, Cc, stop, 1), chain<4, 4>(st, A, B, Cc, stop, 1));
    printf("4 waves/WG, z = 1: loads only (all waves) %.2f  mfma only %.2f ;  z = 2: loads only %.2f  mfma only %.2f  full %.2f\n",
           chain<5, 4>(st, A, B, Cc, stop, 1), chain<6, 4>(st, A, B, Cc, stop, 1), chain<5, 4>(st, A, B, Cc, stop, 2),
           chain<6, 4>(st, A, B, Cc, stop, 2), chain<4, 4>(st, A, B, Cc, stop, 2));
    printf("8 waves/WG, z = 2: loads only %.2f  mfma only %.2f\n", chain<5, 8>(st, A, B, Cc, stop, 2), chain<6, 8>(st, A, B, Cc, stop, 2));
    printf("grid z = 2 (the batched pair), 8 waves: +loads %.2f  +mfma %.2f  full %.2f ; 4 waves: full %.2f\n",
           chain<2, 8>(st, A, B, Cc, stop, 2), chain<3, 8>(st, A, B, Cc, stop, 2), chain<4, 8>(st, A, B, Cc, stop, 2),
           chain<4, 4>(st, A, B, Cc, stop, 2));
    return 0;
}
