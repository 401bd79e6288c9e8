// thip_context.hip -- context, stream, device memory and host<->device transfers.
// Replaces totsu_f32cuda/src/cuda_mgr.rs (context + library handles) and the host/device mirroring
// primitives of totsu_f32cuda/src/f32cuda_slice.rs:343-355.
#include "thip_common.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>

namespace thip {

static Ctx g_ctx;
static char g_err[512] = "";
static std::mutex g_stage_mutex;     // the pinned staging buffer and the shared scratch are per-context resources

Ctx &ctx() { return g_ctx; }

int fail(int code, const char *what, const char *file, int line)
{
    const char *hs = (code > 0 && code < 10000) ? hipGetErrorString((hipError_t)code) : "thip error";
    snprintf(g_err, sizeof(g_err), "%s:%d: %s -> %d (%s)", file, line, what, code, hs);
    return code;
}

int need_init()
{
    return fail(THIP_E_NOTINIT, "thip_init() has not been called (or found no GPU)", __FILE__, __LINE__);
}

int scratch(size_t n, float **out)
{
    std::lock_guard<std::mutex> lock(g_stage_mutex);
    Ctx &c = ctx();
    if (n > c.scratch_n) {
        // grow geometrically; callers size the scratch before entering a hot loop
        size_t want = n + n / 4 + 1024;
        float *p = nullptr;
        THIP_TRY(hipStreamSynchronize(c.stream));
        THIP_TRY(hipMalloc((void **)&p, want * sizeof(float)));
        if (c.scratch) THIP_TRY(hipFree(c.scratch));
        c.scratch = p;
        c.scratch_n = want;
    }
    *out = c.scratch;
    return 0;
}

// *host_out = *dev_src in stream order (SYNC).  (A pinned "mailbox" written by a one-thread kernel and polled by the host
// instead of copy + stream wait was measured: +1.4 % on the trait-level LP rate -- not kept.)
int fetch_scalar(const float *dev_src, float *host_out)
{
    Ctx &c = ctx();
    THIP_TRY(hipMemcpyAsync(c.pinned, dev_src, sizeof(float), hipMemcpyDeviceToHost, c.stream));
    THIP_TRY(hipStreamSynchronize(c.stream));
    *host_out = c.pinned[0];
    return 0;
}

}  // namespace thip

using namespace thip;

extern "C" {

const char *thip_last_error(void) { return g_err; }
const char *thip_version(void) { return "totsu_f32hip 0.3 (gfx950; ABI 3: thip_param is 40 bytes)"; }

int thip_device_count(int *host_count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    if (host_count) *host_count = n;
    return 0;
}

int thip_init(int device)
{
    Ctx &c = ctx();
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(THIP_E_NOGPU, "no HIP device visible: totsu_f32hip has no CPU fallback", __FILE__, __LINE__);
    if (device < 0 || device >= n) return fail(THIP_E_INVALID, "device index out of range", __FILE__, __LINE__);
    if (c.inited && c.device == device) return 0;
    if (c.inited) thip_shutdown();
    THIP_TRY(hipSetDevice(device));
    c.device = device;
    THIP_TRY(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
    c.stream = c.own_stream;
    THIP_TRY(hipMalloc((void **)&c.dev_scalar, 64 * sizeof(float)));
    THIP_TRY(hipMalloc((void **)&c.never_stop, sizeof(int)));
    THIP_TRY(hipMemset(c.never_stop, 0, sizeof(int)));
    THIP_TRY(hipHostMalloc((void **)&c.pinned, 64 * sizeof(float), hipHostMallocDefault));
    c.stage_bytes = 8u << 20;
    THIP_TRY(hipHostMalloc(&c.stage, 2 * c.stage_bytes, hipHostMallocDefault));
    for (int k = 0; k < 2; ++k) THIP_TRY(hipEventCreateWithFlags(&c.stage_ev[k], hipEventDisableTiming));
    hipDeviceProp_t prop;
    THIP_TRY(hipGetDeviceProperties(&prop, device));
    c.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c.inited = true;
    return 0;
}

int thip_shutdown(void)
{
    Ctx &c = ctx();
    if (!c.inited) return 0;
    hipSetDevice(c.device);
    hipStreamSynchronize(c.stream);
    prof_release();
    lazy_release();
    if (c.scratch) hipFree(c.scratch);
    if (c.dev_scalar) hipFree(c.dev_scalar);
    if (c.never_stop) hipFree(c.never_stop);
    c.never_stop = nullptr;
    if (c.pinned) hipHostFree(c.pinned);
    if (c.eig_pin) hipHostFree(c.eig_pin);
    for (int k = 0; k < 2; ++k) if (c.eig_ev[k]) hipEventDestroy(c.eig_ev[k]);
    if (c.eig_side) hipStreamDestroy(c.eig_side);
    if (c.stage) hipHostFree(c.stage);
    for (int k = 0; k < 2; ++k) if (c.stage_ev[k]) hipEventDestroy(c.stage_ev[k]);
    if (c.own_stream) hipStreamDestroy(c.own_stream);
    c = Ctx();
    return 0;
}

int thip_set_stream(void *hip_stream)
{
    THIP_NEED_INIT();
    Ctx &c = ctx();
    c.stream = hip_stream ? (hipStream_t)hip_stream : c.own_stream;
    return 0;
}

void *thip_get_stream(void)
{
    // the caller may enqueue its own work behind ours: what has been deferred is launched first; if that fails the
    // stream is NOT handed out (NULL; thip_last_error() has the reason)
    if (ctx().inited && lazy_pending() && lazy_flush() != 0) return nullptr;
    return (void *)ctx().stream;
}

int thip_sync(void)
{
    THIP_NEED_INIT();
    THIP_TRY(hipStreamSynchronize(ctx().stream));
    return 0;
}

int thip_alloc(size_t n, float **out)
{
    THIP_NEED_INIT();
    if (!out) return fail(THIP_E_INVALID, "out == NULL", __FILE__, __LINE__);
    *out = nullptr;
    // never hand out a NULL for n == 0: zero-length slices are legal and may be offset
    THIP_TRY(hipMalloc((void **)out, (n ? n : 1) * sizeof(float)));
    return 0;
}

int thip_alloc_zeroed(size_t n, float **out)
{
    THIP_RC(thip_alloc(n, out));
    THIP_TRY(hipMemsetAsync(*out, 0, (n ? n : 1) * sizeof(float), ctx().stream));
    return 0;
}

int thip_free(float *p)
{
    THIP_NEED_INIT();
    if (!p) return 0;
    THIP_TRY(hipStreamSynchronize(ctx().stream));
    {
        // learnt call plans / read-ahead plans hold raw device addresses (thip_lazy.hip): those inside this allocation go
        hipDeviceptr_t base = nullptr; size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) == hipSuccess && size > 0)
            lazy_forget((uintptr_t)base, (uintptr_t)base + size);
        else
            lazy_forget(0, ~(uintptr_t)0);
    }
    THIP_TRY(hipFree(p));
    return 0;
}

int thip_h2d(float *dst, const float *host_src, size_t n)
{
    THIP_NEED_INIT();
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lock(g_stage_mutex);
    Ctx &c = ctx();
    // pageable source: staged through two pinned halves, so the host copy of chunk i + 1 overlaps the DMA of chunk i;
    // the call returns only when everything has landed (the caller may reuse host_src)
    const char *src = (const char *)host_src;
    char *d = (char *)dst;
    size_t bytes = n * sizeof(float);
    for (size_t i = 0; bytes; ++i) {
        const size_t b = bytes < c.stage_bytes ? bytes : c.stage_bytes;
        const int k = (int)(i & 1);
        char *half = (char *)c.stage + (size_t)k * c.stage_bytes;
        if (i >= 2) THIP_TRY(hipEventSynchronize(c.stage_ev[k]));     // the DMA that last read this half is done
        memcpy(half, src, b);
        THIP_TRY(hipMemcpyAsync(d, half, b, hipMemcpyHostToDevice, c.stream));
        THIP_TRY(hipEventRecord(c.stage_ev[k], c.stream));
        src += b; d += b; bytes -= b;
    }
    THIP_TRY(hipStreamSynchronize(c.stream));
    return 0;
}

int thip_d2h(float *host_dst, const float *src, size_t n)
{
    THIP_NEED_INIT();
    if (n == 0) return 0;
    Ctx &c = ctx();
    THIP_TRY(hipMemcpyAsync(host_dst, src, n * sizeof(float), hipMemcpyDeviceToHost, c.stream));
    THIP_TRY(hipStreamSynchronize(c.stream));
    return 0;
}

int thip_get(const float *x, size_t idx, float *host_out)
{
    THIP_NEED_INIT_NOFLUSH();
    // with deferred execution on, the reads of a host loop are fetched ahead in one transfer (thip_lazy.hip)
    int served = 0;
    THIP_RC(lazy_read(0, x + idx, 1, host_out, &served));
    if (served) return 0;
    return fetch_scalar(x + idx, host_out);
}

__global__ void set_kernel(float *x, float v) { x[0] = v; }

int thip_set(float *x, size_t idx, float val)
{
    THIP_NEED_INIT_NOFLUSH();
    int deferred = 0;
    THIP_RC(lazy_push_set(x + idx, val, &deferred));
    if (deferred) return 0;
    hipLaunchKernelGGL(set_kernel, dim3(1), dim3(1), 0, ctx().stream, x + idx, val);
    THIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
