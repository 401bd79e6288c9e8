// accuracy of v_rcp_f64 on gfx950 (how many Newton steps the f64 recurrences of thip_trieig.hip need)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const double *x, double *r0, double *r1, double *r2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    r = fma(r, fma(-v, r, 1.0), r);
    r1[i] = r;
    r = fma(r, fma(-v, r, 1.0), r);
    r2[i] = r;
}
int main()
{
    const int n = 1 << 20;
    double *hx = new double[n], *h0 = new double[n], *h1 = new double[n], *h2 = new double[n];
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double m = 1.0 + (double)(s >> 11) / 9007199254740992.0;
        int e = (int)((s >> 3) % 600) - 300;
        hx[i] = ldexp(m, e) * ((s & 1) ? 1 : -1);
    }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    hipMemcpy(h0, d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)hx[i];
        e0 = fmax(e0, (double)fabsl(((long double)h0[i] - t) / t));
        e1 = fmax(e1, (double)fabsl(((long double)h1[i] - t) / t));
        e2 = fmax(e2, (double)fabsl(((long double)h2[i] - t) / t));
    }
    printf("max relative error of v_rcp_f64: %.3e (2^%.1f); after one Newton step %.3e; after two %.3e\n", e0, log2(e0), e1, e2);
    return 0;
}
