// thip_oneshot.hip -- one-shot all-reduce over peer-mapped buffers for the latency-bound messages of the row-sharded
// solver (n + 1024 floats = 204 KB at BASELINE configs[2], two per iteration).
//
// xGMI is point to point: every GPU of a node reaches each of its 7 peers over its own link.  A ring all-reduce spends
// 2 (N - 1) dependent hops on a message that small; here every rank
//   1. copies its contribution into its own communication slot (device memory that the peers have mapped through
//      hipIpcOpenMemHandle) and publishes "chunk c of call q is there" by writing q into the flag word [c][rank] of
//      EVERY peer (a 4-byte remote store per peer and chunk),
//   2. waits on its LOCAL flag words until all N contributions of the chunk have been announced, reads the N slots (its
//      own and N - 1 remote reads, one per link) and sums them IN RANK ORDER into the caller's buffer.
// One launch, no trailing barrier: slots alternate by the parity of the call number, and a rank can only START call
// q + 2 after every peer has announced call q + 1, i.e. after every peer has finished reading call q's slot.  The sum
// has the same order on every rank, so replicated vectors stay bitwise identical across ranks (the RCCL ring gives
// that too; a tree would not).  The unit of work is a chunk of 2048 floats per workgroup -- chunks progress
// independently, so nothing in the kernel needs a grid-wide barrier; the grid (<= 256 workgroups of 256 threads) is
// co-resident on any gfx950 part.
// The slots are allocated uncached (hipDeviceMallocUncached, what RCCL uses for its own peer buffers) so that a peer's
// read observes the data the flag announces.  A rank that waits longer than ~4 s gives up and raises the
// communicator's error word instead of hanging the GPU (thip_oneshot_error).
// The reference has no collectives (cuda_mgr.rs:37-39 hard-codes device 0).
#include "thip_common.h"

#include <cstring>
#include <vector>

using namespace thip;

namespace {

constexpr int CH = 2048;                 // floats per chunk = per workgroup
constexpr int OBLK = 256;
constexpr int MAXW = 16;                 // ranks

struct Peers { float *slot[MAXW]; unsigned *flag[MAXW]; };

struct OneShot {
    int rank = 0, world = 0;
    size_t cap = 0;                      // floats per slot (a multiple of CH)
    size_t nchunk = 0;
    char *mem = nullptr;                 // local region: [flags: nchunk * MAXW words][error word ...][slot 0][slot 1]
    size_t flag_bytes = 0, bytes = 0;
    void *peer_mem[MAXW] = {};           // mapped regions of the peers (own entry = mem)
    bool connected = false;
    unsigned seq = 0;                    // calls so far
} g;

__device__ __forceinline__ unsigned ld_flag(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// buf[0 .. n) <- sum over ranks of buf, in rank order.  slot(q, r) = peer r's slot of parity q & 1.
__global__ __launch_bounds__(OBLK) void oneshot_k(Peers pe, float *__restrict__ buf, size_t n, size_t cap, unsigned seq,
                                                 int rank, int world, unsigned *__restrict__ err, long long timeout_ticks)
{
    const size_t c = blockIdx.x, i0 = c * CH;
    const size_t par = (size_t)(seq & 1u) * cap;
    const int tid = threadIdx.x;
    float *mine = pe.slot[rank] + par + i0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const size_t len = n - i0 < (size_t)CH ? n - i0 : (size_t)CH;
    // 1. contribution -> own slot (peers read it from there), then announce the chunk to every rank (own flags included)
    for (size_t i = tid; i < len; i += OBLK) __builtin_nontemporal_store(buf[i0 + i], mine + i);
    __threadfence_system();
    __syncthreads();
    if (tid < world) __hip_atomic_store(pe.flag[tid] + c * MAXW + rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // 2. wait for the N announcements of this chunk (local words), bounded
    __shared__ int bad;
    if (tid == 0) bad = 0;
    __syncthreads();
    if (tid < world) {
        const unsigned *f = pe.flag[rank] + c * MAXW + tid;
        const long long t0 = wall_clock64();
        // sequence numbers only grow; (int) difference handles the wrap
        while ((int)(ld_flag(f) - seq) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > timeout_ticks) { bad = 1; break; }
        }
    }
    __syncthreads();
    if (bad) { if (tid == 0) atomicExch(err, 1u); return; }
    // the N slots, summed in rank order (identical on every rank)
    for (size_t i = (size_t)tid * 4; i < len; i += (size_t)OBLK * 4) {
        if (i + 4 <= len) {
            f4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
            for (int r = 0; r < world; ++r) {
                const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(pe.slot[r] + par + i0 + i));
                acc += v;
            }
            *reinterpret_cast<f4 *>(buf + i0 + i) = acc;
        } else {
            for (size_t k = i; k < len; ++k) {
                float acc = 0.0f;
                for (int r = 0; r < world; ++r) acc += __builtin_nontemporal_load(pe.slot[r] + par + i0 + k);
                buf[i0 + k] = acc;
            }
        }
    }
}

unsigned *flags_of(void *region) { return reinterpret_cast<unsigned *>(region); }
float *slots_of(void *region) { return reinterpret_cast<float *>(static_cast<char *>(region) + g.flag_bytes); }
unsigned *err_word() { return flags_of(g.mem) + g.nchunk * MAXW; }

int oneshot_allreduce(void *, float *buf, size_t n, void *stream)
{
    if (!g.connected) return fail(THIP_E_NOTINIT, "thip_oneshot_connect() has not been called", __FILE__, __LINE__);
    if (n == 0) return 0;
    if (n > g.cap) return fail(THIP_E_INVALID, "one-shot all-reduce: message longer than the slots", __FILE__, __LINE__);
    if (((uintptr_t)buf & 15u) != 0) return fail(THIP_E_INVALID, "one-shot all-reduce: buffer not 16-byte aligned", __FILE__, __LINE__);
    g.seq += 1;
    Peers pe;
    for (int r = 0; r < g.world; ++r) { pe.slot[r] = slots_of(g.peer_mem[r]); pe.flag[r] = flags_of(g.peer_mem[r]); }
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx().device) != hipSuccess || khz <= 0) khz = 100000;
    const unsigned grid = (unsigned)((n + CH - 1) / CH);
    hipLaunchKernelGGL(oneshot_k, dim3(grid), dim3(OBLK), 0, (hipStream_t)stream, pe, buf, n, g.cap, g.seq, g.rank, g.world,
                       err_word(), (long long)khz * 4000ll);
    THIP_LAUNCH_CHECK();
    return 0;
}

}  // namespace

namespace thip {
thip_allreduce_fn oneshot_hook() { return oneshot_allreduce; }
const unsigned *oneshot_error_word() { return g.mem ? err_word() : nullptr; }
}  // namespace thip

extern "C" {

int thip_oneshot_init(int rank, int world, size_t max_floats, uint8_t *host_handle64)
{
    THIP_NEED_INIT();
    if (g.mem) return fail(THIP_E_INVALID, "one-shot communicator already initialised", __FILE__, __LINE__);
    if (world < 1 || world > MAXW || rank < 0 || rank >= world || !host_handle64 || max_floats == 0)
        return fail(THIP_E_INVALID, "thip_oneshot_init: bad argument (at most 16 ranks)", __FILE__, __LINE__);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    g.rank = rank; g.world = world;
    g.nchunk = (max_floats + CH - 1) / CH;
    if (g.nchunk > 256) return fail(THIP_E_INVALID, "one-shot all-reduce is for messages up to 2 MB", __FILE__, __LINE__);
    g.cap = g.nchunk * CH;
    g.flag_bytes = (g.nchunk * MAXW * sizeof(unsigned) + 256 + 255) / 256 * 256;
    g.bytes = g.flag_bytes + 2 * g.cap * sizeof(float);
    THIP_TRY(hipSetDevice(ctx().device));
    THIP_TRY(hipExtMallocWithFlags((void **)&g.mem, g.bytes, hipDeviceMallocUncached));
    THIP_TRY(hipMemset(g.mem, 0, g.bytes));
    THIP_TRY(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    THIP_TRY(hipIpcGetMemHandle(&h, g.mem));
    memcpy(host_handle64, &h, 64);
    g.seq = 0; g.connected = false;
    return 0;
}

int thip_oneshot_connect(const uint8_t *host_handles)
{
    THIP_NEED_INIT();
    if (!g.mem || !host_handles) return fail(THIP_E_NOTINIT, "thip_oneshot_init() first", __FILE__, __LINE__);
    for (int r = 0; r < g.world; ++r) {
        if (r == g.rank) { g.peer_mem[r] = g.mem; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, host_handles + (size_t)r * 64, 64);
        THIP_TRY(hipIpcOpenMemHandle(&g.peer_mem[r], h, hipIpcMemLazyEnablePeerAccess));
    }
    g.connected = true;
    return 0;
}

int thip_oneshot_allreduce(float *dev_buf, size_t n)
{
    THIP_NEED_INIT();
    return oneshot_allreduce(nullptr, dev_buf, n, (void *)ctx().stream);
}

int thip_oneshot_error(int *host_err)
{
    THIP_NEED_INIT();
    if (!host_err) return fail(THIP_E_INVALID, "null argument", __FILE__, __LINE__);
    *host_err = 0;
    if (!g.mem) return 0;
    unsigned e = 0;
    THIP_TRY(hipStreamSynchronize(ctx().stream));
    THIP_TRY(hipMemcpy(&e, err_word(), sizeof(unsigned), hipMemcpyDeviceToHost));
    *host_err = (int)e;
    return 0;
}

int thip_solver_use_oneshot(thip_solver *s)
{
    if (!g.connected) return fail(THIP_E_NOTINIT, "thip_oneshot_connect() has not been called", __FILE__, __LINE__);
    return thip_solver_set_allreduce(s, oneshot_allreduce, nullptr);
}

int thip_oneshot_destroy(void)
{
    if (!g.mem) return 0;
    if (ctx().inited) hipDeviceSynchronize();
    for (int r = 0; r < g.world; ++r)
        if (r != g.rank && g.peer_mem[r]) hipIpcCloseMemHandle(g.peer_mem[r]);
    hipFree(g.mem);
    g = OneShot();
    return 0;
}

}  // extern "C"
