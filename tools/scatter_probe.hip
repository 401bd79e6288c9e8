// scatter_probe.hip -- what a sparse one-pass sweep may cost on gfx950: per stored entry (4 B value + 4 B row index, streamed
// with 16-byte loads) either TWO gathers (the dots with v and x_y) or TWO float atomic adds (the axpys into A u and A x_x), for
// row indices that are contiguous within a wave (a dense column block: the scaled l1reg_lp matrix) or random (a general
// sparse matrix), accumulators of 256 KB (m = 65 536) or 8 MB (m = 2 M), atomics at agent scope into ONE accumulator pair or at
// workgroup scope into one pair per XCD (performed in that XCD's L2; correct only if all adders of a copy sit on one XCD --
// checked here against the expected sums).
//   hipcc -O3 --offload-arch=gfx950 tools/scatter_probe.hip -o tools/scatter_probe && ./tools/scatter_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void fill_k(float *vals, int *idx, size_t nnz, int m, int random)
{
    for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < nnz; k += (size_t)gridDim.x * blockDim.x) {
        vals[k] = 1.0f;
        unsigned long long z = k * 0x9E3779B97F4A7C15ull + 0x94D049BB133111EBull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        idx[k] = random ? (int)(z % (unsigned long long)m) : (int)(k % (size_t)m);
    }
}

// MODE 0: stream only; 1: two gathers per entry; 2: two agent-scope atomics per entry; 3: two workgroup-scope atomics per entry
// into the XCD's own copy; 4: gathers AND agent atomics (the one-pass sweep); 5: gathers and per-XCD atomics
template <int MODE>
__global__ __launch_bounds__(256) void probe_k(const f32x4 *__restrict__ vals, const i32x4 *__restrict__ idx, size_t n4,
                                               const float *__restrict__ v, const float *__restrict__ xy, float *acc1,
                                               float *acc2, size_t acc_stride, float *out)
{
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;
    if (MODE == 3 || MODE == 5) { acc1 += xcc * acc_stride; acc2 += xcc * acc_stride; }
    const size_t stride = (size_t)gridDim.x * 256;
    float s = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 2 * stride) {
        f32x4 a[2]; i32x4 r[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t k = i + u * stride < n4 ? i + u * stride : i;
            a[u] = __builtin_nontemporal_load(vals + k);
            r[u] = __builtin_nontemporal_load(idx + k);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (i + u * stride >= n4) continue;
            float d1 = 0.0f, d2 = 0.0f;
            if (MODE == 1 || MODE >= 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { d1 = fmaf(a[u][e], v[r[u][e]], d1); d2 = fmaf(a[u][e], xy[r[u][e]], d2); }
            } else {
                d1 = a[u][0] + a[u][1]; d2 = a[u][2] + a[u][3] + (float)(r[u][0] ^ r[u][3]);
            }
            s += d1 + d2;
            if (MODE == 2 || MODE == 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __hip_atomic_fetch_add(acc1 + r[u][e], a[u][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(acc2 + r[u][e], a[u][e] * 2.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (MODE == 3 || MODE == 5) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __hip_atomic_fetch_add(acc1 + r[u][e], a[u][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(acc2 + r[u][e], a[u][e] * 2.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    if (s == 123.456f) out[0] = s;
}

// entries of a lane are 4 CONSECUTIVE rows in the contiguous pattern (a dwordx4 of a dense column): the same instruction's
// lanes then cover 256 consecutive rows; the e-th atomic of the wave touches rows e, e + 4, ... (stride 4)

int main()
{
    const size_t nnz = (size_t)1 << 28;        // 2 GB of entries (value + index)
    const size_t n4 = nnz / 4;
    float *vals, *v, *xy, *acc, *out; int *idx;
    hipMalloc((void **)&vals, nnz * 4); hipMalloc((void **)&idx, nnz * 4);
    const size_t mmax = (size_t)1 << 21;
    hipMalloc((void **)&v, mmax * 4); hipMalloc((void **)&xy, mmax * 4);
    hipMalloc((void **)&acc, 2 * 8 * mmax * 4); hipMalloc((void **)&out, 4);
    hipMemset(v, 0, mmax * 4); hipMemset(xy, 0, mmax * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    static const char *names[6] = { "stream only", "2 gathers", "2 agent atomics", "2 per-XCD atomics", "gathers + agent atomics", "gathers + per-XCD atomics" };
    for (int m : { 65536, 1 << 21 })
        for (int random = 0; random < 2; ++random) {
            hipLaunchKernelGGL(fill_k, dim3(4096), dim3(256), 0, 0, vals, idx, nnz, m, random);
            for (int mode = 0; mode < 6; ++mode)
                for (int blocks : { 2048, 8192 }) {
                    float best = 1e30f;
                    for (int rep = 0; rep < 3; ++rep) {
                        hipMemsetAsync(acc, 0, 2 * 8 * mmax * 4, 0);
                        hipEventRecord(e0, 0);
#define GO(M) hipLaunchKernelGGL(probe_k<M>, dim3(blocks), dim3(256), 0, 0, (const f32x4 *)vals, (const i32x4 *)idx, n4, v, xy, acc, acc + 8 * mmax, mmax, out)
                        switch (mode) { case 0: GO(0); break; case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break; default: GO(5); }
                        hipEventRecord(e1, 0); hipEventSynchronize(e1);
                        float ms; hipEventElapsedTime(&ms, e0, e1);
                        if (rep > 0 && ms < best) best = ms;
                    }
                    // check the sums: every entry added 1 to acc1[row] (over all copies)
                    const char *verdict = "";
                    if (mode >= 2 && blocks == 2048) {
                        std::vector<float> h(8 * mmax);
                        hipMemcpy(h.data(), acc, 8 * mmax * 4, hipMemcpyDeviceToHost);
                        double tot = 0.0; const int copies = (mode == 3 || mode == 5) ? 8 : 1;
                        for (int c = 0; c < copies; ++c) for (int r = 0; r < m; ++r) tot += h[c * mmax + r];
                        verdict = tot == (double)nnz ? "  sums OK" : "  SUMS WRONG";
                        if (tot != (double)nnz) printf("   (total %.0f of %.0f)\n", tot, (double)nnz);
                    }
                    printf("m=%-8d %s  %-28s blocks=%-5d %8.3f ms  %7.1f GB/s of entries%s\n", m, random ? "random    " : "contiguous", names[mode], blocks, best, nnz * 8.0 / (best * 1e-3) / 1e9, verdict);
                }
        }
    return 0;
}
