// thip_comm.hip -- RCCL all-reduce over xGMI for the row-sharded solver, called natively from the
// library: the collective is enqueued on the library's own launch stream, in order with the kernels that produce
// and consume the buffer -- no second stream, no event hand-offs.  librccl is dlopen()ed on first use so that the
// single-GPU product has no dependency on it; one communicator per process (one process per GPU).
// The reference has no collectives (cuda_mgr.rs:37-39 hard-codes device 0).
#include "thip_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

using namespace thip;

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
} g;

int load_rccl()
{
    if (g.handle) return 0;
    // a copy already loaded by the host process (e.g. torch's) is reused: lookup is by soname
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char *nm : names) {
        g.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (g.handle) break;
    }
    if (!g.handle) return fail(THIP_E_INVALID, "cannot dlopen librccl", __FILE__, __LINE__);
    g.GetUniqueId = (decltype(g.GetUniqueId))dlsym(g.handle, "ncclGetUniqueId");
    g.CommInitRank = (decltype(g.CommInitRank))dlsym(g.handle, "ncclCommInitRank");
    g.AllReduce = (decltype(g.AllReduce))dlsym(g.handle, "ncclAllReduce");
    g.CommDestroy = (decltype(g.CommDestroy))dlsym(g.handle, "ncclCommDestroy");
    g.GetErrorString = (decltype(g.GetErrorString))dlsym(g.handle, "ncclGetErrorString");
    g.CommCount = (decltype(g.CommCount))dlsym(g.handle, "ncclCommCount");
    if (!g.GetUniqueId || !g.CommInitRank || !g.AllReduce || !g.CommDestroy)
        return fail(THIP_E_INVALID, "librccl lacks a required symbol", __FILE__, __LINE__);
    return 0;
}

int nccl_fail(ncclResult_t r, const char *what)
{
    return fail(20000 + (int)r, g.GetErrorString ? g.GetErrorString(r) : what, __FILE__, __LINE__);
}

// thip_allreduce_fn: in-place float sum on the given stream
int rccl_allreduce(void *, float *buf, size_t n, void *stream)
{
    if (!g.comm) return fail(THIP_E_NOTINIT, "thip_comm_init() has not been called", __FILE__, __LINE__);
    const ncclResult_t r = g.AllReduce(buf, buf, n, ncclFloat, ncclSum, g.comm, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : nccl_fail(r, "ncclAllReduce");
}

}  // namespace

extern "C" {

int thip_comm_unique_id(uint8_t *host_id128)
{
    THIP_RC(load_rccl());
    ncclUniqueId id;
    const ncclResult_t r = g.GetUniqueId(&id);
    if (r != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    memcpy(host_id128, &id, 128);
    return 0;
}

int thip_comm_init(int rank, int world, const uint8_t *host_id128)
{
    THIP_NEED_INIT();
    THIP_RC(load_rccl());
    if (g.comm) return fail(THIP_E_INVALID, "communicator already initialised", __FILE__, __LINE__);
    if (world < 1 || rank < 0 || rank >= world) return fail(THIP_E_INVALID, "bad rank / world", __FILE__, __LINE__);
    ncclUniqueId id;
    memcpy(&id, host_id128, 128);
    THIP_TRY(hipSetDevice(ctx().device));
    const ncclResult_t r = g.CommInitRank(&g.comm, world, id, rank);
    if (r != ncclSuccess) { g.comm = nullptr; return nccl_fail(r, "ncclCommInitRank"); }
    g.rank = rank; g.world = world;
    return 0;
}

int thip_comm_destroy(void)
{
    if (!g.comm) return 0;
    if (ctx().inited) hipStreamSynchronize(ctx().stream);
    const ncclResult_t r = g.CommDestroy(g.comm);
    g.comm = nullptr;
    return r == ncclSuccess ? 0 : nccl_fail(r, "ncclCommDestroy");
}

int thip_comm_count(int *host_ranks)
{
    if (!host_ranks) return fail(THIP_E_INVALID, "null argument", __FILE__, __LINE__);
    *host_ranks = 0;
    if (!g.comm) return 0;                 // no communicator: 0 ranks, not an error
    if (!g.CommCount) return fail(THIP_E_INVALID, "librccl lacks ncclCommCount", __FILE__, __LINE__);
    const ncclResult_t r = g.CommCount(g.comm, host_ranks);
    return r == ncclSuccess ? 0 : nccl_fail(r, "ncclCommCount");
}

int thip_comm_allreduce(float *dev_buf, size_t n)
{
    THIP_NEED_INIT();
    return rccl_allreduce(nullptr, dev_buf, n, (void *)ctx().stream);
}

int thip_solver_use_rccl(thip_solver *s)
{
    if (!g.comm) return fail(THIP_E_NOTINIT, "thip_comm_init() has not been called", __FILE__, __LINE__);
    // in order on the launch stream by default; thip_solver_set_overlap(s, 1) moves the collective to the solver's side
    // stream, under the stage's local-row work (it costs two extra launches and two cross-stream event hand-offs per
    // stage -- measured +27 us per iteration at world size 1 -- so it pays only where the collective's latency exceeds
    // that: callers time both, as bench.py does)
    return thip_solver_set_allreduce(s, rccl_allreduce, nullptr);
}

}  // extern "C"
