// Does a second pass over a matrix that does not fit the 256 MB Infinity Cache run faster when it walks the memory in the
// OPPOSITE direction (its head is the first pass's tail, possibly still cached)?  Two back-to-back streaming reads of the same
// buffer, forward / forward against forward / backward, non-temporal and plain loads.
//   hipcc -O3 --offload-arch=gfx950 tools/mall_reverse_probe.hip -o tools/mall_probe_bin && ./tools/mall_probe_bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void read_k(const f32x4_t *__restrict__ p, size_t n4, int reverse, float *out)
{
    // block b reads a contiguous segment (like a column chunk of the GEMV); segments are taken in dispatch order or reversed
    const size_t nb = gridDim.x, b = reverse ? nb - 1 - blockIdx.x : blockIdx.x;
    const size_t per = (n4 + nb - 1) / nb, beg = b * per, end = beg + per < n4 ? beg + per : n4;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    size_t i = beg + threadIdx.x;
    for (; i + 7 * 256 < end; i += 8 * 256) {
        f32x4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; i < end; i += 256) acc += NT ? __builtin_nontemporal_load(p + i) : p[i];
    const float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 123.456f) out[0] = s;
}

int main()
{
    for (size_t mb : {400, 800, 1000, 2000, 20000}) {
        const size_t bytes = mb * 1000000ull, n4 = bytes / 16;
        f32x4_t *p; float *out;
        if (hipMalloc((void **)&p, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
        (void)hipMalloc((void **)&out, 4);
        (void)hipMemset(p, 0x3c, bytes);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int nt = 0; nt < 2; ++nt)
            for (int alt = 0; alt < 2; ++alt) {
                const int blocks = 8192, reps = 20;
                for (int w = 0; w < 4; ++w) {
                    if (nt) hipLaunchKernelGGL(read_k<true>, dim3(blocks), dim3(256), 0, 0, p, n4, alt ? (w & 1) : 0, out);
                    else hipLaunchKernelGGL(read_k<false>, dim3(blocks), dim3(256), 0, 0, p, n4, alt ? (w & 1) : 0, out);
                }
                (void)hipEventRecord(e0, 0);
                for (int r = 0; r < reps; ++r) {
                    if (nt) hipLaunchKernelGGL(read_k<true>, dim3(blocks), dim3(256), 0, 0, p, n4, alt ? (r & 1) : 0, out);
                    else hipLaunchKernelGGL(read_k<false>, dim3(blocks), dim3(256), 0, 0, p, n4, alt ? (r & 1) : 0, out);
                }
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                printf("%6zu MB  nt=%d  %s  %.3f ms per pass  %.0f GB/s\n", mb, nt, alt ? "alternating" : "forward    ", ms / reps,
                       bytes / (ms / reps * 1e-3) / 1e9);
            }
        (void)hipFree(p); (void)hipFree(out);
    }
    return 0;
}
