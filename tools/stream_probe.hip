// stream_probe.hip -- bare HBM read stream on gfx950: each lane sums non-temporal float4 loads, 8 in flight; the
// practical ceiling the dual-GEMV kernel is compared with in DESIGN.md.
//   hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o tools/stream_probe && ./tools/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void read_k(const f32x4_t *__restrict__ p, size_t n4, float *out)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + 7 * stride < n4; i += 8 * stride) {
        f32x4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; i < n4; i += stride) acc += NT ? __builtin_nontemporal_load(p + i) : p[i];
    const float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 123.456f) out[0] = s;      // keeps the loads alive
}

int main()
{
    const size_t bytes = (size_t)20e9;
    const size_t n4 = bytes / 16;
    f32x4_t *p; float *out;
    if (hipMalloc((void **)&p, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc((void **)&out, 4);
    hipMemset(p, 0x3c, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nt = 0; nt < 2; ++nt)
        for (int blocks : {2048, 4096, 8192, 16384, 65536}) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0, 0);
                if (nt) hipLaunchKernelGGL(read_k<true>, dim3(blocks), dim3(256), 0, 0, p, n4, out);
                else    hipLaunchKernelGGL(read_k<false>, dim3(blocks), dim3(256), 0, 0, p, n4, out);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("nt=%d blocks=%-6d  %.3f ms  %.1f GB/s\n", nt, blocks, best, bytes / (best * 1e-3) / 1e9);
        }
    return 0;
}
