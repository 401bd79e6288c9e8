// lds_atomic_probe.hip -- the rate of LDS atomic adds on gfx950 by type: what bounds the scattered-row side of the tiled sparse
// products (thip_sptile.hip: two ds_add_f32 per entry ran at 0.8 TB/s of entries).  256 workgroups x 512 threads, each lane adds to a
// pseudo-random word of an 8192-word LDS array, `reps` times; reported: lane-adds per second over the chip and per clock per CU.
//   hipcc -O3 --offload-arch=gfx950 tools/lds_atomic_probe.hip -o tools/lds_atomic_probe && ./tools/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(512) void probe_k(int reps, int stride_mode, float *out)
{
    __shared__ float f[8192];
    __shared__ unsigned long long u8[4096];
    unsigned *u = reinterpret_cast<unsigned *>(f);
    for (int i = threadIdx.x; i < 8192; i += 512) f[i] = 0.0f;
    for (int i = threadIdx.x; i < 4096; i += 512) u8[i] = 0ull;
    __syncthreads();
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x;
    for (int r = 0; r < reps; ++r) {
        h = h * 1664525u + 1013904223u;
        const int idx = stride_mode ? ((threadIdx.x + r * 64) & 8191) : (int)(h >> 19);       // consecutive lanes / random words
        if (MODE == 0) atomicAdd(&f[idx], 1.0f);
        else if (MODE == 1) atomicAdd(&u[idx], 1u);
        else if (MODE == 2) atomicAdd(&u8[idx & 4095], 1ull);
        else if (MODE == 3) f[idx] += 1.0f;                      // plain read-modify-write (not atomic: the rate of ds_read + ds_write)
        else if (MODE == 4) { float v = f[idx]; asm volatile("" : "+v"(v)); if (v == 12345.f) out[1] = v; }   // one ds_read
    }
    __syncthreads();
    if (threadIdx.x == 0 && (f[0] == -1.0f || u8[0] == 7ull)) out[0] = 1.0f;
}

int main()
{
    float *out; hipMalloc((void **)&out, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20000;
    static const char *names[5] = { "ds_add_f32", "ds_add_u32", "ds_add_u64", "ds_read + add + ds_write (not atomic)", "ds_read_b32" };
    for (int sm = 0; sm < 2; ++sm)
        for (int mode = 0; mode < 5; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                switch (mode) {
                case 0: hipLaunchKernelGGL(probe_k<0>, dim3(256), dim3(512), 0, 0, reps, sm, out); break;
                case 1: hipLaunchKernelGGL(probe_k<1>, dim3(256), dim3(512), 0, 0, reps, sm, out); break;
                case 2: hipLaunchKernelGGL(probe_k<2>, dim3(256), dim3(512), 0, 0, reps, sm, out); break;
                case 3: hipLaunchKernelGGL(probe_k<3>, dim3(256), dim3(512), 0, 0, reps, sm, out); break;
                default: hipLaunchKernelGGL(probe_k<4>, dim3(256), dim3(512), 0, 0, reps, sm, out); break;
                }
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double adds = 256.0 * 512.0 * reps;
            printf("%-10s %-40s %8.3f ms  %7.1f G lane-ops/s  %.2f per clock per CU (2.4 GHz)\n", sm ? "consecutive" : "random", names[mode], best,
                   adds / (best * 1e-3) / 1e9, adds / (best * 1e-3) / 256.0 / 2.4e9);
        }
    return 0;
}
