// thip_lazy.hip -- deferred, batched execution of the small calls of a composite operator and of a product cone: the
// "grouped GEMV over a descriptor table" for the TRAIT-LEVEL path (hard parts).  OPT-IN (thip_set_lazy_gemv).
//
// An unchanged totsu drives a composite operator block by block: ProbSOCPOpA::op / ::trans_op (totsu/src/problem/
// socp.rs:77-130) issue one `LinAlgEx::transform_ge` per c_i and per G_i, ProbSOCPOpB (socp.rs:194-246) a `scale`, an
// `add` and a `transform_ge` per cone, ProbSOCPCone::proj (socp.rs:296-313) one projection per cone -- 2000 + 3000 calls
// per K product and 2 x 1000 projections per iteration at BASELINE configs[2], tens of thousands per iteration.  One
// launch each is launch-bound (0.75 iter/s measured).  The trait surface gives no handle on the loops, but nothing
// OBSERVES a result until some other call reads it.  So thip_transform_ge, thip_scale / thip_add (vectors <= 1024 long)
// and the single-cone projections only RECORD their call; the record is run when any other entry point is called
// (THIP_NEED_INIT), when a new call would read or overwrite what a pending one writes, or when it is full.  Stream
// order therefore still equals call order as far as any caller of the API can tell.
//
// A record is a SEGMENT of one of two kinds:
//  (a) products: every call has the form  y <- beta y + (a contribution)  on one output vector y:
//        N     alpha A x               matrix, nr, nc > 1                 (partial sums over column chunks)
//        T     alpha A^T x                                                (partial sums over row tiles)
//        AXPY  alpha v x[0]            op of a column vector / trans_op of a row vector
//        DOT   alpha v . x             trans_op of a column vector / op of a row vector  (y is one number)
//        ADDV  alpha x                 LinAlg::add
//        SCALE (nothing)               LinAlg::scale: beta only
//      Calls on the SAME y compose into one group  y <- B y + sum_k a_k c_k : a later call with factor beta multiplies B
//      and every a_k recorded so far.  ProbSOCPOpA::trans_op (one scale + 2000 contributions into one n-vector) and
//      ProbSOCPOpB::trans_op (one scale + 2000 into one number) each become ONE group, summed in a fixed order
//      (deterministic; the order differs from the reference's sequential additions in the last bits only).
//      An N and a T product of the SAME block in one segment -- SelfDualEmbed::op / trans_op issue a.trans_op and a.op
//      back to back (solver.rs:122-125, 146-149) -- are PAIRED: one read of the block serves both (a 6-pass iteration
//      becomes a 3-pass one without the caller changing a line).  Blocks above 64 MB (the single G of a ProbLP) join
//      the record too and run through the dual GEMV of the fused loop, paired the same way.
//  (b) projections: a run of thip_proj_soc / _rotsoc / _rpos / _zero calls of one kind on pairwise disjoint slices
//      (the x_y and then the x_s blocks of ProbSOCPCone::proj): ONE launch over a table of slices.
//
// The loop issues the SAME call sequence every iteration, so a flushed segment is kept as a PLAN (its call list, its
// device tables, its launch geometry) and the plan that followed it last time is PREDICTED to follow it again: while
// the incoming calls match the prediction call for call (same entry point, shapes and addresses; alpha and beta may
// differ -- criteria_conv passes 1 / tau, solver.rs:594-597) nothing is analysed or built -- a compare and two stores per
// call -- and the flush is the plan's launches (plus an upload of the factors if they changed).  The first mismatch
// falls back to the analysing path with nothing lost.
#include "thip_common.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

using namespace thip;

namespace {

constexpr int BLK = 256;
constexpr size_t LAZY_MAX_ELEMS = (size_t)16 << 20;     // blocks above 64 MB take the dual GEMV of the fused loop at the flush
constexpr size_t LAZY_MAX_VEC = 1024;                   // scale / add on longer vectors run at once
constexpr size_t LAZY_MAX_OPS = 32768;
constexpr int SHORT_LEN = 2048;                         // groups up to this long are finished by one workgroup each
constexpr int MAX_PLANS = 64;

enum { K_N = 0, K_T = 1, K_AXPY = 2, K_DOT = 3, K_ADDV = 4, K_SCALE = 5, K_CONST = 6 };
enum { OP_GE = 0, OP_SCALE = 1, OP_ADD = 2, OP_PROJ = 3, OP_SET = 4 };

// one API call as issued
struct Call {
    uint8_t op, transpose, bclass, pkind;   // bclass: beta == 0 -> 0, == 1 -> 1, else 2
    int grp, mem;                           // (products) group it went to / its own index if it left a member, else -1
    float alpha, beta;
    size_t nr, nc;                          // GE: the matrix; SCALE / ADD / PROJ: nr = length
    const float *A, *x;
    float *y;
};
bool same_shape(const Call &a, const Call &b)
{
    return a.op == b.op && a.transpose == b.transpose && a.bclass == b.bclass && a.pkind == b.pkind && a.nr == b.nr
           && a.nc == b.nc && a.A == b.A && a.x == b.x && a.y == b.y;
}
uint8_t beta_class(float b) { return b == 0.0f ? 0 : (b == 1.0f ? 1 : 2); }

struct Member {
    int kind; size_t nr, nc, inlen;
    float alpha;
    const float *A, *x;            // A: matrix / the vector v ; x: the input vector (ADDV: x only)
    int call;                      // index of the call that made it
};
struct Group {
    float *y; size_t len; float beta;
    std::vector<Member> mem;
};

// device-side tables
struct DotD { const float *v, *x; float *out; int len; int pad; };
enum { M_PART = 0, M_AXPY = 1, M_ADDV = 2, M_CONST = 3 };
struct FinMember { const float *src; const float *xs; int count; int type; size_t stride; };
struct FinGroup { float *y; int len; int first, count; };
struct BigMat { const float *A, *xn, *xt; size_t nr, nc; size_t scr_off, scr_floats; };   // run by dual_gemv_partials at flush

struct Plan {
    int type = 0;                            // 0: products, 1: projections
    std::vector<Call> calls;
    uint64_t key = 0;
    bool replayable = true;
    unsigned generation = 0;                 // of the shared partial-sum buffer its tables point into
    char *dev = nullptr; size_t dev_bytes = 0;
    // products
    int nN = 0, nT = 0, nD = 0, nDot = 0, nLong = 0, nShort = 0, maxlen = 0;
    int long_members = 0;                    // most members of a long group (picks the form of fin_k<true>)
    int maxt[3] = { 0, 0, 0 }, maxc[3] = { 0, 0, 0 };
    size_t off_dot = 0, off_grp = 0, off_mem = 0, off_alpha = 0, off_beta = 0;
    size_t n_members = 0, n_groups = 0;
    std::vector<float> alphas, betas;        // what the device tables hold
    std::vector<int> mem_slot, grp_slot;     // call index -> member slot; recording-time group index -> group slot
    std::vector<BigMat> big;
    // projections
    int pkind = 0; float *base = nullptr; size_t ncones = 0, max_len = 0;
    Plan *next = nullptr;                    // the segment that followed this one last time
    uint64_t last_use = 0;
};

// ---- read-ahead of the SYNC scalars of a host loop ------------------------------------------------------------------------
// The reference's ConeSOC::proj reads two host scalars per cone -- SliceLike::get(0) and LinAlg::norm (cone_soc.rs:44-47) --
// and ProbSOCPCone::proj visits 1000 cones twice per iteration (socp.rs:296-313): 4000 blocking round trips of ~25 us.
// The reads of a pass are independent of each other (each cone only WRITES its own slice, after its reads), and the loop
// asks for the same addresses every iteration.  So the sequence of reads of a pass is learnt, and on the next pass the
// first read fetches ALL of them with one kernel and one transfer; the later ones are served from the host copy -- as
// long as each request is the predicted one and no call issued in between writes into a range still to be read (checked
// against every recorded write; any other entry point drops the cache).  The writes between the reads (scale, set) are
// recorded, so a pass of the reference's literal cone code costs one fetch and one flush.
enum { RD_GET = 0, RD_NORM = 1 };
struct ReadReq { int kind; const float *p; size_t n; };
struct ReadD { const float *p; unsigned n; int kind; };
struct ReadPlan {
    std::vector<ReadReq> reqs;
    std::vector<uintptr_t> suf_lo, suf_hi;      // hull of the ranges of reqs[i ..]
    char *dev = nullptr;                        // table of ReadD, then the results
    uint64_t last_use = 0;
};

struct Queue {
    // ---- read-ahead
    std::vector<ReadReq> rd_learn; bool rd_open = false;
    std::vector<ReadPlan *> rd_plans;
    ReadPlan *rd_cur = nullptr; size_t rd_pos = 0;
    std::vector<float> rd_vals;
    float *rd_pin = nullptr; size_t rd_pin_n = 0;
    long long rd_served = 0, rd_fetches = 0;
    // ---- the segment being recorded by the analysing path
    std::vector<Group> groups;
    std::unordered_map<const float *, int> target;         // y -> group
    size_t n_members = 0;
    uintptr_t xlo = ~(uintptr_t)0, xhi = 0, ylo = ~(uintptr_t)0, yhi = 0;
    std::vector<Call> calls;                               // of the segment being recorded (either kind)
    bool seg_replayable = true;
    int proj_kind = -1;                                    // >= 0: the segment is a projection run of this kind
    uintptr_t plo = ~(uintptr_t)0, phi = 0;
    // ---- replay
    Plan *pred = nullptr;                                  // predicted plan
    size_t pos = 0;                                        // calls of pred matched so far
    bool replaying = false;
    std::vector<float> cur_alpha, cur_beta;                // factors of the matched calls (by call index)
    Plan *last_plan = nullptr;                             // the plan flushed last (its `next` is learnt)
    std::vector<Plan *> plans;
    uint64_t tick = 0;
    // ---- shared
    std::mutex mu;
    bool enabled = false, env_read = false;     // OFF unless a host asks for it (thip_set_lazy_gemv) or THIP_LAZY_GEMV=1
    float *part = nullptr; size_t part_floats = 0; unsigned generation = 1;      // partial sums (shared by the plans)
    // the plans' device tables live in one arena, handed out front to back; when it (or the plan list) is full ALL plans
    // are dropped and the arena starts over -- no allocation per plan (a host loop that never repeats itself, e.g. a
    // known-answer test driven call by call, would otherwise pay a hipMalloc and a hipFree per flushed segment)
    char *tab = nullptr; size_t tab_bytes = 0, tab_used = 0;
    // pinned staging of tables / factors: two halves, an event each ("the upload out of this half has finished")
    char *pin[2] = { nullptr, nullptr }; size_t pin_bytes[2] = { 0, 0 }; hipEvent_t pin_ev[2] = { nullptr, nullptr };
    int pin_next = 0;
    long long flushes = 0, deferred = 0, hits = 0, misses = 0;
    std::atomic<bool> pending{ false };                     // read without the lock by every entry point
} Q;

bool overlap(uintptr_t a0, uintptr_t a1, uintptr_t b0, uintptr_t b1) { return a0 < b1 && b0 < a1; }

// THIP_LAZY_TRACE=1: one line on stderr per flushed segment with what made it end (experiments only)
bool tracing()
{
    static const int on = getenv("THIP_LAZY_TRACE") ? atoi(getenv("THIP_LAZY_TRACE")) : 0;
    return on != 0;
}
const char *g_why = "entry point";

// out[0] = v . x   (one block per product; the factor is applied by the finishing kernel).  1024 threads, four loads in
// flight each: c . x over the 10 000 .. 50 000 entries of a ProbLP / ProbSOCP objective row is a latency chain otherwise
constexpr int DBLK = 1024;
__global__ __launch_bounds__(DBLK) void dot_k(const DotD *__restrict__ tab)
{
    __shared__ double shd[16];
    const DotD d = tab[blockIdx.x];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = threadIdx.x;
    for (; i + 3 * DBLK < d.len; i += 4 * DBLK) {
        const float a0 = d.v[i], a1 = d.v[i + DBLK], a2 = d.v[i + 2 * DBLK], a3 = d.v[i + 3 * DBLK];
        const float b0 = d.x[i], b1 = d.x[i + DBLK], b2 = d.x[i + 2 * DBLK], b3 = d.x[i + 3 * DBLK];
        s0 += (double)a0 * (double)b0; s1 += (double)a1 * (double)b1; s2 += (double)a2 * (double)b2; s3 += (double)a3 * (double)b3;
    }
    for (; i < d.len; i += DBLK) s0 += (double)d.v[i] * (double)d.x[i];
    const double s = block_sum_d((s0 + s1) + (s2 + s3), shd);
    if (threadIdx.x == 0) d.out[0] = (float)s;
}

// one member's contribution to element c (alpha applied)
__device__ __forceinline__ double one_member(const FinMember &m, float al, int c)
{
    if (m.type == M_PART) {
        // partial sums are read SIXTEEN at a time: the sum over the ~100 column-chunk partials of an N product is a
        // latency chain, not a bandwidth problem (four at a time: 40 us for the 14 MB of an LP's block, of which 30 are waits)
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        const float *p = m.src + c;
        int t = 0;
        for (; t + 16 <= m.count; t += 16) {
            float a[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) a[u] = p[(size_t)(t + u) * m.stride];
#pragma unroll
            for (int u = 0; u < 16; u += 4) { s0 += (double)a[u]; s1 += (double)a[u + 1]; s2 += (double)a[u + 2]; s3 += (double)a[u + 3]; }
        }
        for (; t + 4 <= m.count; t += 4) {
            const float a0 = p[(size_t)t * m.stride], a1 = p[(size_t)(t + 1) * m.stride];
            const float a2 = p[(size_t)(t + 2) * m.stride], a3 = p[(size_t)(t + 3) * m.stride];
            s0 += (double)a0; s1 += (double)a1; s2 += (double)a2; s3 += (double)a3;
        }
        for (; t < m.count; ++t) s0 += (double)p[(size_t)t * m.stride];
        return (double)al * ((s0 + s1) + (s2 + s3));
    }
    if (m.type == M_AXPY) return (double)(al * m.xs[0] * m.src[c]);
    if (m.type == M_CONST) return (double)al;                     // SliceLike::set: y <- value
    return (double)(al * m.src[c]);
}

// contributions of members k0, k0 + kstep, .. of a group to element c.  Eight members at a time: their descriptors
// (wave-uniform: scalar loads), then their eight data loads, then the sums -- a group with 2000 members (the G_i^T x_i and
// c_i x_i of ProbSOCPOpA::trans_op) is otherwise 2000 dependent descriptor -> data round trips per element.
__device__ __forceinline__ double contributions(const FinGroup &g, const FinMember *__restrict__ mem,
                                                const float *__restrict__ alphas, int c, int k0, int kstep)
{
    constexpr int U = 8;
    double acc = 0.0;
    int k = k0;
    for (; k + (U - 1) * kstep < g.count; k += U * kstep) {
        FinMember m[U];
        float al[U];
        bool simple = true;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            m[u] = mem[g.first + k + u * kstep];
            al[u] = alphas[g.first + k + u * kstep];
            simple = simple && (m[u].type == M_ADDV || (m[u].type == M_PART && m[u].count == 1) || m[u].type == M_AXPY);
        }
        if (simple) {
            float v[U], x0[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { v[u] = m[u].src[c]; x0[u] = m[u].type == M_AXPY ? m[u].xs[0] : 1.0f; }
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc += m[u].type == M_PART ? (double)al[u] * (double)v[u] : (double)(al[u] * x0[u] * v[u]);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) acc += one_member(m[u], al[u], c);
        }
    }
    for (; k < g.count; k += kstep) acc += one_member(mem[g.first + k], alphas[g.first + k], c);
    return acc;
}

// the same for E elements c0, c0 + 64, .. of one thread: a member's descriptor (a 32-byte scalar-path load per wave) is
// read once for E data loads.  A group of 2000 members is 72 KB of descriptors and factors per wave; with one element per
// thread the 12 500 waves of a 50 000-long output pulled 900 MB through the scalar caches (fin_k<true>: 0.34 ms for 200 MB
// of data, whatever the number of member lanes).
template <int E>
__device__ __forceinline__ void contributions_multi(const FinGroup &g, const FinMember *__restrict__ mem,
                                                    const float *__restrict__ alphas, int c0, int k0, int kstep, double *acc)
{
    constexpr int U = 8;
    int k = k0;
    for (; k + (U - 1) * kstep < g.count; k += U * kstep) {
        FinMember m[U];
        float al[U];
        bool simple = true;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            m[u] = mem[g.first + k + u * kstep];
            al[u] = alphas[g.first + k + u * kstep];
            simple = simple && (m[u].type == M_ADDV || (m[u].type == M_PART && m[u].count == 1) || m[u].type == M_AXPY);
        }
        if (simple) {
            float v[U][E], x0[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                x0[u] = m[u].type == M_AXPY ? m[u].xs[0] : 1.0f;
#pragma unroll
                for (int i = 0; i < E; ++i) v[u][i] = c0 + 64 * i < g.len ? m[u].src[c0 + 64 * i] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < E; ++i)
                    acc[i] += m[u].type == M_PART ? (double)al[u] * (double)v[u][i] : (double)(al[u] * x0[u] * v[u][i]);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < E; ++i)
                    if (c0 + 64 * i < g.len) acc[i] += one_member(m[u], al[u], c0 + 64 * i);
        }
    }
    for (; k < g.count; k += kstep) {
        const FinMember m = mem[g.first + k];
        const float al = alphas[g.first + k];
#pragma unroll
        for (int i = 0; i < E; ++i)
            if (c0 + 64 * i < g.len) acc[i] += one_member(m, al, c0 + 64 * i);
    }
}

// y[c] = beta y[c] + sum of the group's contributions.
// LONG: grid (len / 64, groups): 64 elements x 16 member lanes per workgroup -- a group with thousands of members (the
// 1000 G_i^T x_i + 1000 c_i x_i of ProbSOCPOpA::trans_op, socp.rs:104-130) is summed eight members at a time per lane
// and combined through LDS.  Else one workgroup per group (the 2000 short outputs of ProbSOCPOpA::op).
// FIN_E elements per thread of the LONG form.  Thousands of members (the SOCP's T partials): 4 x 16 lanes -- measured
// 1 -> 0.34 ms, 4 -> 0.14 ms, 8 -> 0.35 ms (too few waves).  A handful of members with many partial slabs each (the one big
// block of an LP): the elements are the only parallelism, so one element per thread and 4 lanes (4 x 16 there: 40 -> 61 us).
template <bool LONG, int FIN_LONG_LANES, int FIN_E = 1>          // member lanes of the LONG form: 64 elements x 16 (4) lanes = 1024 (256) threads
__global__ __launch_bounds__(LONG ? 64 * FIN_LONG_LANES : BLK) void fin_k(const FinGroup *__restrict__ groups, const FinMember *__restrict__ mem,
                                             const float *__restrict__ alphas, const float *__restrict__ betas)
{
    const int gi = LONG ? blockIdx.y : blockIdx.x;
    const FinGroup g = groups[gi];
    const float beta = betas[gi];
    if (LONG) {
        // round 3: 16 member lanes instead of 4 -- a group of 2000 members was 62 dependent descriptor -> data round trips per
        // thread (0.34 ms for the 200 MB of T partials of a pass over the SOCP blocks: 590 GB/s); now 16
        constexpr int E = FIN_E;             // elements per thread: c0 + 64 i; a workgroup covers 64 E consecutive elements
        __shared__ double comb[FIN_LONG_LANES - 1][E][64];
        const int e = threadIdx.x & 63, kq = threadIdx.x >> 6;
        const int c0 = blockIdx.x * (64 * E) + e;
        double a[E];
#pragma unroll
        for (int i = 0; i < E; ++i) a[i] = 0.0;
        if (c0 < g.len) contributions_multi<E>(g, mem, alphas, c0, kq, FIN_LONG_LANES, a);
        if (kq > 0) {
#pragma unroll
            for (int i = 0; i < E; ++i) comb[kq - 1][i][e] = a[i];
        }
        __syncthreads();
        if (kq == 0) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int c = c0 + 64 * i;
                if (c < g.len) {
                    double sacc = a[i];
#pragma unroll
                    for (int q = 0; q < FIN_LONG_LANES - 1; ++q) sacc += comb[q][i][e];
                    const float v = (float)sacc;
                    g.y[c] = beta == 0.0f ? v : fmaf(beta, g.y[c], v);
                }
            }
        }
    } else if (g.len <= 4 && g.count > 32) {
        // a few numbers with MANY contributions (ProbSOCPOpB::trans_op, socp.rs:219-246: 1000 d_i x_i + 1000 h_i . x_i
        // into ONE number): the members are the parallel dimension, one block-wide sum per element
        __shared__ double shd[16];
        for (int c = 0; c < g.len; ++c) {
            const double a = block_sum_d(contributions(g, mem, alphas, c, threadIdx.x, BLK), shd);
            if (threadIdx.x == 0) { const float v = (float)a; g.y[c] = beta == 0.0f ? v : fmaf(beta, g.y[c], v); }
            __syncthreads();
        }
    } else {
        for (int c = threadIdx.x; c < g.len; c += BLK) {
            const float a = (float)contributions(g, mem, alphas, c, 0, 1);
            g.y[c] = beta == 0.0f ? a : fmaf(beta, g.y[c], a);
        }
    }
}

// element-wise cones over a table of slices: x <- max(x, 0) (cone_rpos.rs:38-45) or x <- 0 (cone_zero.rs:38-44, primal)
__global__ __launch_bounds__(BLK) void ewise_table_k(float *__restrict__ base, const int64_t *__restrict__ begs,
                                                     const int64_t *__restrict__ ends, int zero)
{
    const int64_t b = begs[blockIdx.x], e = ends[blockIdx.x];
    for (int64_t i = b + (int64_t)blockIdx.y * BLK + threadIdx.x; i < e; i += (int64_t)gridDim.y * BLK)
        base[i] = zero ? 0.0f : fmaxf(base[i], 0.0f);
}

// all the scalars of a pass in one launch: request i -> out[i] (one workgroup each): x[0], or ||x||_2 with f64 squares
__global__ __launch_bounds__(BLK) void read_batch_k(const ReadD *__restrict__ tab, float *__restrict__ out)
{
    __shared__ double shd[16];
    const ReadD d = tab[blockIdx.x];
    if (d.kind == RD_GET) { if (threadIdx.x == 0) out[blockIdx.x] = d.p[0]; return; }
    double acc = 0.0;
    for (unsigned i = threadIdx.x; i < d.n; i += BLK) { const double t = (double)d.p[i]; acc += t * t; }
    acc = block_sum_d(acc, shd);
    if (threadIdx.x == 0) out[blockIdx.x] = (float)sqrt(acc);
}

size_t up256(size_t v) { return (v + 255) / 256 * 256; }

void free_plan(Plan *p) { delete p; }        // its tables live in the arena

void drop_all_plans()
{
    for (Plan *p : Q.plans) free_plan(p);
    Q.plans.clear();
    Q.pred = Q.last_plan = nullptr;
    Q.replaying = false; Q.pos = 0;
    Q.tab_used = 0;
}

// `bytes` of table space for a new plan (256-byte aligned)
int alloc_tab(size_t bytes, char **out)
{
    bytes = (bytes + 255) / 256 * 256;
    if (Q.tab_used + bytes > Q.tab_bytes || (int)Q.plans.size() >= MAX_PLANS) {
        // start over: launches still in flight may be reading the old tables
        THIP_TRY(hipStreamSynchronize(ctx().stream));
        drop_all_plans();
        if (bytes > Q.tab_bytes) {
            if (Q.tab) { THIP_TRY(hipFree(Q.tab)); Q.tab = nullptr; Q.tab_bytes = 0; }
            const size_t want = std::max<size_t>((size_t)8 << 20, 2 * bytes);
            THIP_TRY(hipMalloc((void **)&Q.tab, want));
            Q.tab_bytes = want;
        }
    }
    *out = Q.tab + Q.tab_used;
    Q.tab_used += bytes;
    return 0;
}

// the shared partial-sum buffer; growing it invalidates every plan (their tables hold pointers into it)
int ensure_part(size_t floats)
{
    if (floats <= Q.part_floats) return 0;
    THIP_TRY(hipStreamSynchronize(ctx().stream));
    if (Q.part) { THIP_TRY(hipFree(Q.part)); Q.part = nullptr; Q.part_floats = 0; }
    const size_t want = floats + floats / 4 + (1 << 18);
    THIP_TRY(hipMalloc((void **)&Q.part, want * sizeof(float)));
    Q.part_floats = want;
    Q.generation += 1;
    return 0;
}

// a pinned staging half that no upload is reading any more
int staging(size_t bytes, char **out, int *half_out)
{
    const int half = Q.pin_next;
    Q.pin_next ^= 1;
    if (Q.pin_ev[half] == nullptr) THIP_TRY(hipEventCreateWithFlags(&Q.pin_ev[half], hipEventDisableTiming));
    else THIP_TRY(hipEventSynchronize(Q.pin_ev[half]));
    if (Q.pin_bytes[half] < bytes) {
        if (Q.pin[half]) THIP_TRY(hipHostFree(Q.pin[half]));
        Q.pin[half] = nullptr; Q.pin_bytes[half] = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        THIP_TRY(hipHostMalloc((void **)&Q.pin[half], want, hipHostMallocDefault));
        Q.pin_bytes[half] = want;
    }
    *out = Q.pin[half];
    *half_out = half;
    return 0;
}

void reset_segment()
{
    Q.groups.clear(); Q.target.clear(); Q.n_members = 0;
    Q.xlo = Q.ylo = ~(uintptr_t)0; Q.xhi = Q.yhi = 0;
    Q.calls.clear();
    Q.seg_replayable = true;
    Q.proj_kind = -1; Q.plo = ~(uintptr_t)0; Q.phi = 0;
}

void update_pending()
{
    // a live read cache / an open learning pass count as pending: ANY other entry point must pass through lazy_flush so
    // that it drops them (it may write what a cached read stands for)
    Q.pending.store(!Q.calls.empty() || (Q.replaying && Q.pos > 0) || Q.rd_cur != nullptr || Q.rd_open, std::memory_order_relaxed);
}

uint64_t mix(uint64_t h, uint64_t v)
{
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    return h * 0xBF58476D1CE4E5B9ull;
}
uint64_t key_of(const std::vector<Call> &calls)
{
    uint64_t h = 0x94D049BB133111EBull ^ calls.size();
    for (const Call &c : calls) {
        h = mix(h, (uint64_t)c.op | ((uint64_t)c.transpose << 8) | ((uint64_t)c.bclass << 16) | ((uint64_t)c.pkind << 24));
        h = mix(h, c.nr); h = mix(h, c.nc);
        h = mix(h, (uint64_t)(uintptr_t)c.A); h = mix(h, (uint64_t)(uintptr_t)c.x); h = mix(h, (uint64_t)(uintptr_t)c.y);
    }
    return h;
}

// ---- running a plan ------------------------------------------------------------------------------------------------------

int launch_plan(Plan *p)
{
    hipStream_t st = ctx().stream;
    Q.flushes += 1;
    p->last_use = ++Q.tick;
    if (p->type == 1) {
        const int64_t *begs = reinterpret_cast<const int64_t *>(p->dev), *ends = begs + p->ncones;
        if (p->pkind == THIP_CONE_SOC || p->pkind == THIP_CONE_ROTSOC)
            THIP_RC(soc_batched(st, p->base, begs, ends, p->ncones, p->pkind == THIP_CONE_ROTSOC, p->max_len, nullptr));
        else {
            const unsigned gy = (unsigned)std::min<size_t>(64, (p->max_len + BLK - 1) / BLK);
            hipLaunchKernelGGL(ewise_table_k, dim3((unsigned)p->ncones, gy ? gy : 1), dim3(BLK), 0, st, p->base, begs, ends,
                               p->pkind == THIP_CONE_ZERO ? 1 : 0);
            THIP_LAUNCH_CHECK();
        }
        return 0;
    }
    const GroupDesc *dgd = reinterpret_cast<const GroupDesc *>(p->dev);
    THIP_RC(grouped_gemv(st, dgd, p->nD, p->maxt[2], (p->maxc[2] + 3) / 4, 2));
    THIP_RC(grouped_gemv(st, dgd + p->nD, p->nN, p->maxt[0], (p->maxc[0] + 3) / 4, 0));
    THIP_RC(grouped_gemv(st, dgd + p->nD + p->nN, p->nT, p->maxt[1], (p->maxc[1] + 3) / 4, 1));
    for (const BigMat &b : p->big) {
        GemvPartials gp;
        THIP_RC(dual_gemv_partials(st, b.nr, b.nc, b.A, b.nr, b.xn, b.xt, b.xn != nullptr, b.xt != nullptr, false,
                                   Q.part + b.scr_off, b.scr_floats, &gp, nullptr));
    }
    if (p->nDot) hipLaunchKernelGGL(dot_k, dim3(p->nDot), dim3(DBLK), 0, st, reinterpret_cast<const DotD *>(p->dev + p->off_dot));
    const FinGroup *dg = reinterpret_cast<const FinGroup *>(p->dev + p->off_grp);
    const FinMember *dm = reinterpret_cast<const FinMember *>(p->dev + p->off_mem);
    const float *da = reinterpret_cast<const float *>(p->dev + p->off_alpha);
    const float *db = reinterpret_cast<const float *>(p->dev + p->off_beta);
    if (p->nLong && p->long_members < 256)
        hipLaunchKernelGGL((fin_k<true, 4, 1>), dim3((p->maxlen + 63) / 64, (unsigned)p->nLong), dim3(256), 0, st, dg, dm, da, db);
    else if (p->nLong)
        hipLaunchKernelGGL((fin_k<true, 16, 4>), dim3((p->maxlen + 255) / 256, (unsigned)p->nLong), dim3(1024), 0, st, dg, dm, da, db);
    if (p->nShort) hipLaunchKernelGGL((fin_k<false, 4>), dim3((unsigned)p->nShort), dim3(BLK), 0, st, dg + p->nLong, dm, da, db + p->nLong);
    THIP_LAUNCH_CHECK();
    return 0;
}

// the factors of this pass differ from what the plan's device tables hold: upload them
int upload_factors(Plan *p, const std::vector<float> &alphas, const std::vector<float> &betas)
{
    const bool da = alphas != p->alphas, db = betas != p->betas;
    if (!da && !db) return 0;
    hipStream_t st = ctx().stream;
    const size_t ba = up256(alphas.size() * sizeof(float)), bb = up256(betas.size() * sizeof(float));
    char *host; int half;
    THIP_RC(staging(ba + bb, &host, &half));
    if (da) {
        memcpy(host, alphas.data(), alphas.size() * sizeof(float));
        THIP_TRY(hipMemcpyAsync(p->dev + p->off_alpha, host, alphas.size() * sizeof(float), hipMemcpyHostToDevice, st));
        p->alphas = alphas;
    }
    if (db) {
        memcpy(host + ba, betas.data(), betas.size() * sizeof(float));
        THIP_TRY(hipMemcpyAsync(p->dev + p->off_beta, host + ba, betas.size() * sizeof(float), hipMemcpyHostToDevice, st));
        p->betas = betas;
    }
    THIP_TRY(hipEventRecord(Q.pin_ev[half], st));
    return 0;
}

// factors of a replayable plan from the factors the calls carry now (by call index): in a replayable segment every
// factor beta != 1 arrives while its group is still empty, so a member's factor is its own call's alpha and a group's
// is the running product push_products forms
void factors_from_calls(const Plan *p, const float *alpha, const float *beta, std::vector<float> &alphas, std::vector<float> &betas)
{
    alphas.assign(p->n_members, 0.0f);
    betas.assign(p->n_groups, 1.0f);
    for (size_t i = 0; i < p->calls.size(); ++i) {
        const Call &c = p->calls[i];
        const int gs = p->grp_slot[c.grp];
        if (c.op != OP_ADD && c.bclass != 1) betas[gs] = c.bclass == 0 ? 0.0f : betas[gs] * beta[i];
        if (c.mem >= 0 && p->mem_slot[i] >= 0) alphas[p->mem_slot[i]] = alpha[i];
    }
}

void plan_done(Plan *p)
{
    if (Q.last_plan && Q.last_plan != p) Q.last_plan->next = p;
    Q.last_plan = p;
    Q.pred = p->next;
    Q.pos = 0;
    Q.replaying = Q.pred != nullptr;
}

void remember(Plan *p) { Q.plans.push_back(p); }       // alloc_tab() has made room

Plan *find_plan(uint64_t key, const std::vector<Call> &calls, int type)
{
    for (Plan *p : Q.plans) {
        if (p->key != key || p->type != type || p->calls.size() != calls.size() || p->generation != Q.generation) continue;
        bool same = true;
        for (size_t i = 0; i < calls.size() && same; ++i) same = same_shape(p->calls[i], calls[i]);
        if (same) return p;
    }
    return nullptr;
}

// ---- the analysing path ----------------------------------------------------------------------------------------------------

int flush_projections()
{
    std::vector<Call> calls;
    calls.swap(Q.calls);
    const int kind = Q.proj_kind;
    reset_segment();
    update_pending();
    const uint64_t key = key_of(calls);
    Plan *p = find_plan(key, calls, 1);
    if (p) { Q.hits += 1; }
    else {
        Q.misses += 1;
        p = new Plan();
        p->type = 1; p->key = key; p->pkind = kind; p->ncones = calls.size(); p->generation = Q.generation;
        uintptr_t lo = ~(uintptr_t)0;
        for (const Call &c : calls) { lo = std::min(lo, (uintptr_t)c.y); p->max_len = std::max(p->max_len, c.nr); }
        p->base = reinterpret_cast<float *>(lo);
        const size_t bytes = up256(2 * calls.size() * sizeof(int64_t));
        hipStream_t st = ctx().stream;
        char *host; int half;
        int rc = staging(bytes, &host, &half);
        if (rc == 0) rc = alloc_tab(bytes, &p->dev);
        if (rc != 0) { delete p; return rc; }
        p->dev_bytes = bytes;
        int64_t *begs = reinterpret_cast<int64_t *>(host), *ends = begs + calls.size();
        for (size_t i = 0; i < calls.size(); ++i) {
            begs[i] = (int64_t)(((uintptr_t)calls[i].y - lo) / sizeof(float));
            ends[i] = begs[i] + (int64_t)calls[i].nr;
        }
        THIP_TRY(hipMemcpyAsync(p->dev, host, 2 * calls.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
        THIP_TRY(hipEventRecord(Q.pin_ev[half], st));
        p->calls.swap(calls);
        remember(p);
    }
    THIP_RC(launch_plan(p));
    plan_done(p);
    return 0;
}

int flush_products()
{
    hipStream_t st = ctx().stream;
    std::vector<Group> groups;
    groups.swap(Q.groups);
    std::vector<Call> calls;
    calls.swap(Q.calls);
    const size_t n_members = Q.n_members;
    const bool replayable = Q.seg_replayable;
    reset_segment();
    update_pending();

    // table order: long groups first, then the short ones (each class is one launch over a contiguous range)
    std::vector<int> order;
    order.reserve(groups.size());
    for (size_t i = 0; i < groups.size(); ++i) if ((int)groups[i].len > SHORT_LEN) order.push_back((int)i);
    const int nLong = (int)order.size();
    for (size_t i = 0; i < groups.size(); ++i) if ((int)groups[i].len <= SHORT_LEN) order.push_back((int)i);
    // the factors of this pass in that order: what either a cached or a new plan needs
    std::vector<float> alphas, betas;
    alphas.reserve(n_members); betas.reserve(groups.size());
    for (int gi : order) { betas.push_back(groups[gi].beta); for (const Member &m : groups[gi].mem) alphas.push_back(m.alpha); }

    const uint64_t key = key_of(calls);
    Plan *p = find_plan(key, calls, 0);
    if (p) {
        Q.hits += 1;
        THIP_RC(upload_factors(p, alphas, betas));
        THIP_RC(launch_plan(p));
        plan_done(p);
        return 0;
    }
    Q.misses += 1;
    p = new Plan();
    p->type = 0; p->key = key; p->replayable = replayable;
    p->n_members = n_members; p->n_groups = groups.size();
    p->nLong = nLong; p->nShort = (int)groups.size() - nLong;

    // ---- pairing: an N and a T product of the same block share one read of it ----
    struct MatRef { int g, m; };
    std::unordered_map<const float *, std::vector<MatRef>> byA;
    for (size_t gi = 0; gi < groups.size(); ++gi)
        for (size_t mi = 0; mi < groups[gi].mem.size(); ++mi) {
            const Member &m = groups[gi].mem[mi];
            if (m.kind == K_N || m.kind == K_T) byA[m.A].push_back(MatRef{ (int)gi, (int)mi });
        }
    std::vector<std::vector<MatRef>> partner(groups.size());
    for (size_t gi = 0; gi < groups.size(); ++gi) partner[gi].assign(groups[gi].mem.size(), MatRef{ -1, -1 });
    for (auto &kv : byA) {
        std::vector<MatRef> ns, ts;
        for (const MatRef &r : kv.second) (groups[r.g].mem[r.m].kind == K_N ? ns : ts).push_back(r);
        const size_t np = std::min(ns.size(), ts.size());
        for (size_t k = 0; k < np; ++k) {
            const Member &a = groups[ns[k].g].mem[ns[k].m], &b = groups[ts[k].g].mem[ts[k].m];
            if (a.nr != b.nr || a.nc != b.nc) continue;
            partner[ns[k].g][ns[k].m] = ts[k];
            partner[ts[k].g][ts[k].m] = ns[k];
        }
    }

    // ---- layout of the partial sums (shared buffer), per matrix member ----
    struct MatPlan { size_t offN, offT; int cpc, tiles, chunks; int big; };       // big: index into p->big, or -1
    std::vector<std::vector<MatPlan>> mp(groups.size());
    for (size_t gi = 0; gi < groups.size(); ++gi) mp[gi].assign(groups[gi].mem.size(), MatPlan{ 0, 0, 0, 0, 0, -1 });
    size_t floats = 0;
    int nDot = 0;
    for (size_t gi = 0; gi < groups.size(); ++gi)
        for (size_t mi = 0; mi < groups[gi].mem.size(); ++mi) {
            const Member &m = groups[gi].mem[mi];
            if (m.kind == K_DOT) { nDot += 1; continue; }
            if (m.kind != K_N && m.kind != K_T) continue;
            const MatRef pr = partner[gi][mi];
            const bool paired = pr.g >= 0;
            if (paired && m.kind == K_T) continue;                 // laid out with its N partner
            const Member *tm = paired ? &groups[pr.g].mem[pr.m] : nullptr;
            const bool doN = m.kind == K_N, doT = m.kind == K_T || paired;
            MatPlan q{ 0, 0, 0, 0, 0, -1 };
            if (m.nr * m.nc > LAZY_MAX_ELEMS) {
                // a big block: the dual GEMV of the fused loop into its own scratch region
                BigMat b{ m.A, doN ? m.x : nullptr, m.kind == K_T ? m.x : (tm ? tm->x : nullptr), m.nr, m.nc, floats,
                          dual_gemv_scratch_floats(m.nr, m.nc) };
                q.big = (int)p->big.size();
                p->big.push_back(b);
                floats += b.scr_floats; floats = (floats + 63) / 64 * 64;
            } else {
                // the unit of the grouped kernel is a wave: 128 rows x cpc columns, ~0.1-0.2 MB of the matrix each
                // (8 .. 1024 columns per chunk); a workgroup = four consecutive chunks
                static const size_t wave_elems = getenv("THIP_LAZY_WAVE_ELEMS") ? (size_t)atol(getenv("THIP_LAZY_WAVE_ELEMS")) : 40000;
                size_t c = wave_elems / (m.nr < 128 ? m.nr : 128);
                c = std::max<size_t>(8, std::min<size_t>(1024, c / 8 * 8));
                q.cpc = (int)c; q.tiles = (int)((m.nr + 127) / 128); q.chunks = (int)((m.nc + c - 1) / c);
                if (doN) { q.offN = floats; floats += (size_t)q.chunks * m.nr; floats = (floats + 63) / 64 * 64; }
                if (doT) { q.offT = floats; floats += (size_t)q.tiles * m.nc; floats = (floats + 63) / 64 * 64; }
            }
            mp[gi][mi] = q;
            if (paired) mp[pr.g][pr.m] = q;
        }
    const size_t dot_off = floats;
    floats += (size_t)nDot;
    int rc = ensure_part(floats);
    if (rc != 0) { delete p; return rc; }
    p->generation = Q.generation;
    float *dpart = Q.part;
    // where dual_gemv_partials will leave the partial sums of the big blocks (a pure function of the shape)
    std::vector<GemvPartials> biggp(p->big.size());
    for (size_t k = 0; k < p->big.size(); ++k) {
        const BigMat &b = p->big[k];
        dual_gemv_partials_geometry(b.nr, b.nc, b.A, b.nr, b.xn != nullptr, b.xt != nullptr, dpart + b.scr_off, &biggp[k]);
    }

    // ---- tables ----
    int nN = 0, nT = 0, nD = 0;
    for (size_t gi = 0; gi < groups.size(); ++gi)
        for (size_t mi = 0; mi < groups[gi].mem.size(); ++mi) {
            const Member &m = groups[gi].mem[mi];
            if ((m.kind != K_N && m.kind != K_T) || mp[gi][mi].big >= 0) continue;
            if (partner[gi][mi].g >= 0) { if (m.kind == K_N) nD += 1; }
            else if (m.kind == K_N) nN += 1; else nT += 1;
        }
    p->nN = nN; p->nT = nT; p->nD = nD; p->nDot = nDot;
    const size_t b_gd = up256((size_t)(nN + nT + nD) * sizeof(GroupDesc));
    const size_t b_dot = up256((size_t)nDot * sizeof(DotD));
    const size_t b_grp = up256(groups.size() * sizeof(FinGroup));
    const size_t b_mem = up256(n_members * sizeof(FinMember));
    const size_t b_al = up256(n_members * sizeof(float)), b_be = up256(groups.size() * sizeof(float));
    p->off_dot = b_gd; p->off_grp = b_gd + b_dot; p->off_mem = p->off_grp + b_grp; p->off_alpha = p->off_mem + b_mem;
    p->off_beta = p->off_alpha + b_al;
    const size_t b_tab = p->off_beta + b_be;
    char *host; int half;
    rc = staging(b_tab, &host, &half);
    if (rc == 0) rc = alloc_tab(b_tab, &p->dev);
    if (rc != 0) { delete p; return rc; }
    p->dev_bytes = b_tab;
    memset(host, 0, b_tab);
    GroupDesc *gdD = reinterpret_cast<GroupDesc *>(host), *gdN = gdD + nD, *gdT = gdN + nN;
    DotD *dd = reinterpret_cast<DotD *>(host + p->off_dot);
    FinGroup *hg = reinterpret_cast<FinGroup *>(host + p->off_grp);
    FinMember *hm = reinterpret_cast<FinMember *>(host + p->off_mem);
    p->mem_slot.assign(calls.size(), -1);
    p->grp_slot.assign(groups.size(), -1);
    int iD = 0, iN = 0, iT = 0, iDot = 0;
    size_t mpos = 0;
    for (size_t oi = 0; oi < order.size(); ++oi) {
        const int gi = order[oi];
        const Group &g = groups[gi];
        p->grp_slot[gi] = (int)oi;
        if ((int)g.len > SHORT_LEN) { p->maxlen = std::max(p->maxlen, (int)g.len); p->long_members = std::max(p->long_members, (int)g.mem.size()); }
        hg[oi] = FinGroup{ g.y, (int)g.len, (int)mpos, (int)g.mem.size() };
        for (size_t mi = 0; mi < g.mem.size(); ++mi) {
            const Member &m = g.mem[mi];
            FinMember fm{ nullptr, nullptr, 0, M_ADDV, 0 };
            if (m.kind == K_N || m.kind == K_T) {
                const MatPlan &q = mp[gi][mi];
                const MatRef pr = partner[gi][mi];
                fm.type = M_PART;
                if (q.big >= 0) {
                    const GemvPartials &gp = biggp[q.big];
                    if (m.kind == K_N) { fm.src = gp.partN; fm.count = gp.nN; fm.stride = gp.strideN; }
                    else               { fm.src = gp.partT; fm.count = gp.nT; fm.stride = gp.strideT; }
                } else if (m.kind == K_N) {
                    GroupDesc gd{ m.A, m.x, nullptr, dpart + q.offN, nullptr, (int)m.nr, (int)m.nc, q.cpc, 0 };
                    if (pr.g >= 0) {
                        gd.xt = groups[pr.g].mem[pr.m].x; gd.partT = dpart + q.offT;
                        gdD[iD++] = gd; p->maxt[2] = std::max(p->maxt[2], q.tiles); p->maxc[2] = std::max(p->maxc[2], q.chunks);
                    } else { gdN[iN++] = gd; p->maxt[0] = std::max(p->maxt[0], q.tiles); p->maxc[0] = std::max(p->maxc[0], q.chunks); }
                    fm.src = dpart + q.offN; fm.count = q.chunks; fm.stride = m.nr;
                } else {
                    if (pr.g < 0) {
                        GroupDesc gd{ m.A, nullptr, m.x, nullptr, dpart + q.offT, (int)m.nr, (int)m.nc, q.cpc, 0 };
                        gdT[iT++] = gd; p->maxt[1] = std::max(p->maxt[1], q.tiles); p->maxc[1] = std::max(p->maxc[1], q.chunks);
                    }
                    fm.src = dpart + q.offT; fm.count = q.tiles; fm.stride = m.nc;
                }
            } else if (m.kind == K_DOT) {
                dd[iDot] = DotD{ m.A, m.x, dpart + dot_off + iDot, (int)m.inlen, 0 };
                fm.src = dpart + dot_off + iDot; fm.type = M_ADDV;
                ++iDot;
            } else if (m.kind == K_AXPY) { fm.src = m.A; fm.xs = m.x; fm.type = M_AXPY; }
            else if (m.kind == K_CONST) { fm.type = M_CONST; }
            else { fm.src = m.x; fm.type = M_ADDV; }                                          // K_ADDV
            if (m.call >= 0 && (size_t)m.call < calls.size()) p->mem_slot[m.call] = (int)mpos;
            hm[mpos++] = fm;
        }
    }
    memcpy(host + p->off_alpha, alphas.data(), alphas.size() * sizeof(float));
    memcpy(host + p->off_beta, betas.data(), betas.size() * sizeof(float));
    THIP_TRY(hipMemcpyAsync(p->dev, host, b_tab, hipMemcpyHostToDevice, st));
    THIP_TRY(hipEventRecord(Q.pin_ev[half], st));
    p->alphas = alphas; p->betas = betas;
    p->calls.swap(calls);
    if (tracing()) fprintf(stderr, "lazy: new product plan of %zu calls: dual %d, N %d, T %d, big %zu, dots %d, replayable %d\n",
                           p->calls.size(), p->nD, p->nN, p->nT, p->big.size(), p->nDot, (int)p->replayable);
    remember(p);
    THIP_RC(launch_plan(p));
    plan_done(p);
    return 0;
}

int flush_segment()
{
    if (Q.calls.empty()) return 0;
    if (tracing()) fprintf(stderr, "lazy: flush %s segment of %zu calls (%zu groups, %zu members): %s\n",
                           Q.proj_kind >= 0 ? "projection" : "product", Q.calls.size(), Q.groups.size(), Q.n_members, g_why);
    g_why = "entry point";
    return Q.proj_kind >= 0 ? flush_projections() : flush_products();
}

// records  y(len) <- beta y + [member]  (member.kind == K_SCALE: no contribution)
int push_products(float *y, size_t len, float beta, Member m, Call c, int *deferred)
{
    if (Q.proj_kind >= 0) { g_why = "a product follows projections"; THIP_RC(flush_segment()); }   // a projection run is pending: it comes first
    const bool has_in = m.kind != K_SCALE && m.kind != K_CONST;
    const uintptr_t y0 = (uintptr_t)y, y1 = y0 + len * sizeof(float);
    const uintptr_t x0 = (uintptr_t)m.x, x1 = has_in ? x0 + m.inlen * sizeof(float) : x0;
    const bool has_a = m.kind == K_N || m.kind == K_T || m.kind == K_AXPY || m.kind == K_DOT;
    const uintptr_t a0 = (uintptr_t)m.A, a1 = has_a ? a0 + m.nr * m.nc * sizeof(float) : a0;
    bool must_flush = Q.n_members + Q.groups.size() >= LAZY_MAX_OPS;
    int join = -1;
    if (!must_flush && !Q.groups.empty()) {
        // hulls first (O(1)): the blocks of a composite operator read one vector and write disjoint pieces of another
        const bool raw = (has_in && overlap(x0, x1, Q.ylo, Q.yhi)) || (has_a && overlap(a0, a1, Q.ylo, Q.yhi));
        const bool war = overlap(y0, y1, Q.xlo, Q.xhi);
        const bool waw = overlap(y0, y1, Q.ylo, Q.yhi);
        if (waw) {
            // pending outputs are pairwise disjoint (anything else was flushed when it was pushed): an exact hit in the
            // map settles the write-write side without a scan
            auto it = Q.target.find(y);
            if (it != Q.target.end() && Q.groups[it->second].len == len) join = it->second;
        }
        if (raw || war || (waw && join < 0)) {
            // exact test against everything pending
            for (size_t gi = 0; gi < Q.groups.size() && !must_flush; ++gi) {
                const Group &g = Q.groups[gi];
                const uintptr_t py0 = (uintptr_t)g.y, py1 = py0 + g.len * sizeof(float);
                if ((has_in && overlap(x0, x1, py0, py1)) || (has_a && overlap(a0, a1, py0, py1))) { must_flush = true; break; }
                if ((int)gi != join && overlap(y0, y1, py0, py1)) { must_flush = true; break; }
                if (war)
                    for (const Member &p : g.mem) {
                        if (p.kind == K_CONST) continue;
                        const uintptr_t px0 = (uintptr_t)p.x, px1 = px0 + p.inlen * sizeof(float);
                        if (overlap(y0, y1, px0, px1)) { must_flush = true; break; }
                        if (p.kind != K_ADDV) {
                            const uintptr_t pa0 = (uintptr_t)p.A, pa1 = pa0 + p.nr * p.nc * sizeof(float);
                            if (overlap(y0, y1, pa0, pa1)) { must_flush = true; break; }
                        }
                    }
            }
        }
    }
    if (must_flush) { g_why = "hazard / full (products)"; THIP_RC(flush_segment()); join = -1; }
    if (join < 0) {
        Q.target[y] = (int)Q.groups.size();
        Q.groups.push_back(Group{ y, len, 1.0f, {} });
        join = (int)Q.groups.size() - 1;
        Q.ylo = std::min(Q.ylo, y0); Q.yhi = std::max(Q.yhi, y1);
    }
    Group &g = Q.groups[join];
    if (beta != 1.0f) {                 // y <- beta (B y + sum a_k c_k) + ...
        if (!g.mem.empty()) Q.seg_replayable = false;          // a factor that rescales recorded members: analysed every time
        if (beta == 0.0f) {
            Q.n_members -= g.mem.size();
            for (Member &pm : g.mem) Q.calls[pm.call].mem = -1;
            g.mem.clear(); g.beta = 0.0f;
        } else { g.beta *= beta; for (Member &p : g.mem) p.alpha *= beta; }
    }
    c.grp = join; c.mem = -1;
    if (m.kind != K_SCALE) {
        m.call = (int)Q.calls.size();
        c.mem = m.call;
        g.mem.push_back(m);
        Q.n_members += 1;
        if (has_in) { Q.xlo = std::min(Q.xlo, x0); Q.xhi = std::max(Q.xhi, x1); }
        if (has_a) { Q.xlo = std::min(Q.xlo, a0); Q.xhi = std::max(Q.xhi, a1); }
    }
    Q.calls.push_back(c);
    Q.deferred += 1;
    *deferred = 1;
    update_pending();
    return 0;
}

int push_projection(const Call &c, int *deferred)
{
    if (Q.proj_kind < 0 && !Q.calls.empty()) { g_why = "a projection follows products"; THIP_RC(flush_segment()); }   // pending products come first
    const uintptr_t p0 = (uintptr_t)c.y, p1 = p0 + c.nr * sizeof(float);
    if (Q.proj_kind >= 0) {
        bool clash = Q.proj_kind != (int)c.pkind || Q.calls.size() >= LAZY_MAX_OPS;
        if (!clash && overlap(p0, p1, Q.plo, Q.phi))
            for (const Call &q : Q.calls) {
                const uintptr_t q0 = (uintptr_t)q.y;
                if (overlap(p0, p1, q0, q0 + q.nr * sizeof(float))) { clash = true; break; }
            }
        if (clash) { g_why = "projection kind / overlap"; THIP_RC(flush_segment()); }
    }
    Q.proj_kind = (int)c.pkind;
    Q.plo = std::min(Q.plo, p0); Q.phi = std::max(Q.phi, p1);
    Q.calls.push_back(c);
    Q.deferred += 1;
    *deferred = 1;
    update_pending();
    return 0;
}

int push_slow(const Call &c, int *deferred)
{
    if (c.op == OP_PROJ) return push_projection(c, deferred);
    Member m{};
    size_t outlen = c.nr;
    float beta = 1.0f;
    if (c.op == OP_GE) {
        m.nr = c.nr; m.nc = c.nc; m.alpha = c.alpha; m.A = c.A; m.x = c.x;
        const bool tr = c.transpose != 0;
        if (c.nc == 1)      { m.kind = tr ? K_DOT : K_AXPY; m.inlen = tr ? c.nr : 1; outlen = tr ? 1 : c.nr; }
        else if (c.nr == 1) { m.kind = tr ? K_AXPY : K_DOT; m.inlen = tr ? 1 : c.nc; outlen = tr ? c.nc : 1; }
        else                { m.kind = tr ? K_T : K_N; m.inlen = tr ? c.nr : c.nc; outlen = tr ? c.nc : c.nr; }
        beta = c.beta;
    } else if (c.op == OP_SCALE) { m.kind = K_SCALE; beta = c.beta; }
    else if (c.op == OP_SET) { m.kind = K_CONST; m.alpha = c.alpha; beta = 0.0f; outlen = 1; }
    else { m.kind = K_ADDV; m.alpha = c.alpha; m.x = c.x; m.inlen = c.nr; m.nr = m.nc = 0; }
    return push_products(c.y, outlen, beta, m, c, deferred);
}

// the prediction failed after `pos` matched calls: hand them to the analysing path with the factors they carried
int abandon_replay()
{
    const size_t n = Q.pos;
    if (tracing()) fprintf(stderr, "lazy: prediction of %zu calls abandoned after %zu\n", Q.pred->calls.size(), n);
    // a copy: the analysing path may create plans, and creating one may evict the plan these calls belong to
    const std::vector<Call> matched(Q.pred->calls.begin(), Q.pred->calls.begin() + n);
    Q.replaying = false; Q.pred = nullptr; Q.pos = 0;
    for (size_t i = 0; i < n; ++i) {
        Call c = matched[i];
        c.alpha = Q.cur_alpha[i]; c.beta = Q.cur_beta[i];
        int d = 0;
        Q.deferred -= 1;                       // counted when it was matched
        THIP_RC(push_slow(c, &d));
    }
    update_pending();
    return 0;
}

// the predicted segment is complete: its launches, with this pass's factors
int finish_replay()
{
    Plan *p = Q.pred;
    if (p->type == 0) {
        std::vector<float> alphas, betas;
        factors_from_calls(p, Q.cur_alpha.data(), Q.cur_beta.data(), alphas, betas);
        THIP_RC(upload_factors(p, alphas, betas));
    }
    Q.hits += 1;
    Q.pos = 0;
    if (tracing()) fprintf(stderr, "lazy: replayed %s plan of %zu calls (dual %d, N %d, T %d, big %zu)\n",
                           p->type ? "projection" : "product", p->calls.size(), p->nD, p->nN, p->nT, p->big.size());
    THIP_RC(launch_plan(p));
    plan_done(p);
    update_pending();
    return 0;
}

// ---- read-ahead ----------------------------------------------------------------------------------------------------------------

void drop_read_cache() { Q.rd_cur = nullptr; Q.rd_pos = 0; }

void free_read_plan(ReadPlan *p) { if (p->dev) hipFree(p->dev); delete p; }

// the learning pass ends: keep what it saw as the plan for the next pass (passes shorter than 16 reads are not worth one)
void close_read_group()
{
    if (!Q.rd_open) return;
    Q.rd_open = false;
    if (Q.rd_learn.size() < 16 || Q.rd_learn.size() > 65536) { Q.rd_learn.clear(); return; }
    for (size_t i = 0; i < Q.rd_plans.size(); ++i) {
        const ReadReq &f = Q.rd_plans[i]->reqs[0];
        if (f.kind == Q.rd_learn[0].kind && f.p == Q.rd_learn[0].p && f.n == Q.rd_learn[0].n) {
            hipStreamSynchronize(ctx().stream);
            free_read_plan(Q.rd_plans[i]);
            Q.rd_plans.erase(Q.rd_plans.begin() + i);
            break;
        }
    }
    if (Q.rd_plans.size() >= 4) {
        size_t v = 0;
        for (size_t i = 1; i < Q.rd_plans.size(); ++i) if (Q.rd_plans[i]->last_use < Q.rd_plans[v]->last_use) v = i;
        hipStreamSynchronize(ctx().stream);
        free_read_plan(Q.rd_plans[v]);
        Q.rd_plans.erase(Q.rd_plans.begin() + v);
    }
    ReadPlan *p = new ReadPlan();
    p->reqs.swap(Q.rd_learn);
    const size_t n = p->reqs.size();
    p->suf_lo.assign(n + 1, ~(uintptr_t)0); p->suf_hi.assign(n + 1, 0);
    for (size_t i = n; i-- > 0;) {
        const uintptr_t a = (uintptr_t)p->reqs[i].p, b = a + p->reqs[i].n * sizeof(float);
        p->suf_lo[i] = std::min(p->suf_lo[i + 1], a); p->suf_hi[i] = std::max(p->suf_hi[i + 1], b);
    }
    const size_t b_tab = up256(n * sizeof(ReadD)), b_out = up256(n * sizeof(float));
    if (hipMalloc((void **)&p->dev, b_tab + b_out) != hipSuccess) { (void)hipGetLastError(); delete p; return; }
    std::vector<ReadD> tab(n);
    for (size_t i = 0; i < n; ++i) tab[i] = ReadD{ p->reqs[i].p, (unsigned)p->reqs[i].n, p->reqs[i].kind };
    if (hipMemcpy(p->dev, tab.data(), n * sizeof(ReadD), hipMemcpyHostToDevice) != hipSuccess) { free_read_plan(p); return; }
    Q.rd_plans.push_back(p);
}

int flush_work();      // runs what is recorded (below)

// a recorded write into [w0, w1): any cached read it could change is dropped
void reads_vs_write(uintptr_t w0, uintptr_t w1)
{
    if (Q.rd_cur == nullptr) return;
    const ReadPlan *p = Q.rd_cur;
    if (!overlap(w0, w1, p->suf_lo[Q.rd_pos], p->suf_hi[Q.rd_pos])) return;
    for (size_t i = Q.rd_pos; i < p->reqs.size(); ++i) {
        const uintptr_t a = (uintptr_t)p->reqs[i].p;
        if (overlap(w0, w1, a, a + p->reqs[i].n * sizeof(float))) { drop_read_cache(); return; }
    }
}

// thip_get / thip_norm: *served = 1 and the value if the read-ahead has it; else everything recorded has been run and the
// caller performs the read itself
int read_request(int kind, const float *p, size_t n, float *value, int *served)
{
    *served = 0;
    if (Q.rd_cur) {
        const ReadPlan *pl = Q.rd_cur;
        const ReadReq &r = pl->reqs[Q.rd_pos];
        if (r.kind == kind && r.p == p && r.n == n) {
            *value = Q.rd_vals[Q.rd_pos];
            Q.rd_pos += 1; Q.rd_served += 1;
            if (Q.rd_pos == pl->reqs.size()) drop_read_cache();
            *served = 1;
            update_pending();
            return 0;
        }
        drop_read_cache();                 // not the predicted request
    }
    ReadPlan *pl = nullptr;
    if (!Q.rd_open)
        for (ReadPlan *q : Q.rd_plans) if (q->reqs[0].kind == kind && q->reqs[0].p == p && q->reqs[0].n == n) { pl = q; break; }
    THIP_RC(flush_work());                 // the read (ours or the caller's) comes after everything recorded so far
    if (pl) {
        hipStream_t st = ctx().stream;
        const size_t nr = pl->reqs.size();
        if (Q.rd_pin_n < nr) {
            if (Q.rd_pin) THIP_TRY(hipHostFree(Q.rd_pin));
            Q.rd_pin = nullptr; Q.rd_pin_n = 0;
            THIP_TRY(hipHostMalloc((void **)&Q.rd_pin, (nr + 1024) * sizeof(float), hipHostMallocDefault));
            Q.rd_pin_n = nr + 1024;
        }
        float *dout = reinterpret_cast<float *>(pl->dev + up256(nr * sizeof(ReadD)));
        hipLaunchKernelGGL(read_batch_k, dim3((unsigned)nr), dim3(BLK), 0, st, reinterpret_cast<const ReadD *>(pl->dev), dout);
        THIP_LAUNCH_CHECK();
        THIP_TRY(hipMemcpyAsync(Q.rd_pin, dout, nr * sizeof(float), hipMemcpyDeviceToHost, st));
        THIP_TRY(hipStreamSynchronize(st));
        Q.rd_vals.assign(Q.rd_pin, Q.rd_pin + nr);
        pl->last_use = ++Q.tick;
        Q.rd_fetches += 1; Q.rd_served += 1;
        Q.rd_cur = pl; Q.rd_pos = 1;
        if (nr == 1) drop_read_cache();
        *value = Q.rd_vals[0];
        *served = 1;
        update_pending();
        return 0;
    }
    // uncached: the caller reads; this pass is being learnt
    if (!Q.rd_open) { Q.rd_learn.clear(); Q.rd_open = true; }
    Q.rd_learn.push_back(ReadReq{ kind, p, n });
    update_pending();
    return 0;
}

uintptr_t write_len(const Call &c)
{
    if (c.op == OP_GE) return (c.transpose ? c.nc : c.nr) * sizeof(float);
    if (c.op == OP_SET) return sizeof(float);
    return c.nr * sizeof(float);
}

int push_call(const Call &c, int *deferred)
{
    *deferred = 0;
    if (Q.rd_cur) reads_vs_write((uintptr_t)c.y, (uintptr_t)c.y + write_len(c));
    for (int guard = 0; guard < 4 && Q.replaying; ++guard) {
        Plan *p = Q.pred;
        if (p->generation != Q.generation || !p->replayable) { THIP_RC(abandon_replay()); break; }
        if (Q.pos == p->calls.size()) { THIP_RC(finish_replay()); continue; }      // next prediction, same call
        if (same_shape(c, p->calls[Q.pos])) {
            if (Q.cur_alpha.size() < p->calls.size()) { Q.cur_alpha.resize(p->calls.size()); Q.cur_beta.resize(p->calls.size()); }
            Q.cur_alpha[Q.pos] = c.alpha; Q.cur_beta[Q.pos] = c.beta;
            Q.pos += 1;
            Q.deferred += 1;
            *deferred = 1;
            Q.pending.store(true, std::memory_order_relaxed);
            return 0;
        }
        THIP_RC(abandon_replay());
        break;
    }
    if (Q.replaying) THIP_RC(abandon_replay());          // (only after four complete predictions in a row on one call)
    return push_slow(c, deferred);
}

// every entry point that is not a recorded call or a served read: what it does may change what a cached read stands for
int flush_locked()
{
    drop_read_cache();
    close_read_group();
    THIP_RC(flush_work());
    update_pending();
    return 0;
}

int flush_work()
{
    if (Q.replaying && Q.pos > 0) {
        if (Q.pos == Q.pred->calls.size() && Q.pred->generation == Q.generation) return finish_replay();
        THIP_RC(abandon_replay());
    }
    THIP_RC(flush_segment());
    update_pending();
    return 0;
}

bool lazy_on()
{
    if (!Q.env_read) {
        const char *e = getenv("THIP_LAZY_GEMV");
        if (e) Q.enabled = atoi(e) != 0;
        Q.env_read = true;
    }
    // a caller that installed its own stream (thip_set_stream) may interleave its own work with ours on it: every call
    // must then have been ENQUEUED when it returns, so nothing is deferred while a foreign stream is installed
    return Q.enabled && ctx().stream == ctx().own_stream;
}

}  // namespace

namespace thip {

bool lazy_pending() { return Q.pending.load(std::memory_order_relaxed); }

int lazy_flush()
{
    std::lock_guard<std::mutex> lock(Q.mu);
    return flush_locked();
}

// thip_free: every learnt plan holds raw device addresses (the scalar reads of a pass; the operands of a replayed segment).
// Once a buffer has been released the addresses may be unmapped -- or, worse, mapped to something else -- and a first read
// that happens to match a plan's head would launch read_batch_k over all of them.  What touches the released range
// [lo, hi) is forgotten: the read-ahead plans whose address hull meets it, and -- when any recorded call of any product /
// projection plan has an operand in it -- the call plans (all of them: they are chained by prediction).  Plans on other
// buffers survive, so a host that solves again on the same vectors does not learn its 4000 scalar reads a second time.
void lazy_forget(uintptr_t lo, uintptr_t hi)
{
    std::lock_guard<std::mutex> lock(Q.mu);
    if (Q.rd_plans.empty() && !Q.rd_open && Q.rd_learn.empty() && Q.plans.empty()) return;
    auto in = [&](const void *p) { const uintptr_t a = (uintptr_t)p; return p != nullptr && a >= lo && a < hi; };
    bool hit = false;
    for (const Plan *p : Q.plans) {
        if (in(p->base)) hit = true;
        for (const Call &c : p->calls) if (in(c.A) || in(c.x) || in(c.y)) { hit = true; break; }
        if (hit) break;
    }
    if (hit) drop_all_plans();
    // read-ahead: the group being learnt and the cached one may hold the range too -- cheap to learn again
    bool rd_hit = false;
    for (const ReadReq &r : Q.rd_learn) if (overlap((uintptr_t)r.p, (uintptr_t)(r.p + (r.n ? r.n : 1)), lo, hi)) rd_hit = true;
    if (rd_hit) { Q.rd_open = false; Q.rd_learn.clear(); }
    for (size_t i = 0; i < Q.rd_plans.size();) {
        ReadPlan *p = Q.rd_plans[i];
        const bool meets = !p->suf_lo.empty() && overlap(p->suf_lo[0], p->suf_hi[0], lo, hi);
        if (meets) {
            if (Q.rd_cur == p) drop_read_cache();
            free_read_plan(p);
            Q.rd_plans.erase(Q.rd_plans.begin() + i);
        } else ++i;
    }
}

void lazy_release()
{
    std::lock_guard<std::mutex> lock(Q.mu);
    reset_segment();
    drop_all_plans();
    drop_read_cache();
    Q.rd_open = false; Q.rd_learn.clear();
    for (ReadPlan *p : Q.rd_plans) free_read_plan(p);
    Q.rd_plans.clear();
    if (Q.rd_pin) { hipHostFree(Q.rd_pin); Q.rd_pin = nullptr; Q.rd_pin_n = 0; }
    Q.pending.store(false, std::memory_order_relaxed);
    if (Q.part) { hipFree(Q.part); Q.part = nullptr; Q.part_floats = 0; Q.generation += 1; }
    if (Q.tab) { hipFree(Q.tab); Q.tab = nullptr; Q.tab_bytes = 0; Q.tab_used = 0; }
    for (int k = 0; k < 2; ++k) {
        if (Q.pin[k]) { hipHostFree(Q.pin[k]); Q.pin[k] = nullptr; Q.pin_bytes[k] = 0; }
        if (Q.pin_ev[k]) { hipEventDestroy(Q.pin_ev[k]); Q.pin_ev[k] = nullptr; }
    }
}

int lazy_push(int transpose, size_t n_row, size_t n_col, float alpha, const float *mat, const float *x, float beta,
              float *y, int *deferred)
{
    std::lock_guard<std::mutex> lock(Q.mu);
    *deferred = 0;
    if (!lazy_on() || n_row > 0x7fffffffull || n_col > 0x7fffffffull) return flush_locked();     // runs now, after the record
    Call c{};
    c.op = OP_GE; c.transpose = transpose ? 1 : 0; c.bclass = beta_class(beta); c.alpha = alpha; c.beta = beta;
    c.nr = n_row; c.nc = n_col; c.A = mat; c.x = x; c.y = y;
    return push_call(c, deferred);
}

// LinAlg::scale (x <- alpha x) and LinAlg::add (y <- alpha x + y) on short vectors join the record
int lazy_push_scale(size_t n, float alpha, float *x, int *deferred)
{
    std::lock_guard<std::mutex> lock(Q.mu);
    *deferred = 0;
    if (alpha == 1.0f) { *deferred = 1; return 0; }          // x <- 1 x: nothing to do, nothing to order (any length)
    if (!lazy_on() || n > LAZY_MAX_VEC || n == 0) return flush_locked();
    Call c{};
    c.op = OP_SCALE; c.bclass = beta_class(alpha); c.beta = alpha; c.nr = n; c.y = x;
    return push_call(c, deferred);
}

int lazy_push_add(size_t n, float alpha, const float *x, float *y, int *deferred)
{
    std::lock_guard<std::mutex> lock(Q.mu);
    *deferred = 0;
    if (!lazy_on() || n > LAZY_MAX_VEC || n == 0) return flush_locked();
    Call c{};
    c.op = OP_ADD; c.bclass = 1; c.alpha = alpha; c.beta = 1.0f; c.nr = n; c.x = x; c.y = y;
    return push_call(c, deferred);
}

// SliceLike::set(idx, value) joins the record (y <- value)
int lazy_push_set(float *x, float value, int *deferred)
{
    std::lock_guard<std::mutex> lock(Q.mu);
    *deferred = 0;
    if (!lazy_on()) return flush_locked();
    Call c{};
    c.op = OP_SET; c.bclass = 0; c.alpha = value; c.beta = 0.0f; c.nr = 1; c.y = x;
    return push_call(c, deferred);
}

// SYNC scalar reads (SliceLike::get, LinAlg::norm): served from the read-ahead when it has them
int lazy_read(int is_norm, const float *p, size_t n, float *value, int *served)
{
    std::lock_guard<std::mutex> lock(Q.mu);
    *served = 0;
    if (!lazy_on() || n == 0 || n > 0xffffffffull) return flush_locked();
    return read_request(is_norm ? RD_NORM : RD_GET, p, n, value, served);
}

// single-cone projections (thip_proj_soc / _rotsoc / _rpos / _zero on x[0 .. n)): consecutive calls of one kind on
// disjoint slices become one launch
int lazy_push_proj(int kind, size_t n, float *x, int *deferred)
{
    std::lock_guard<std::mutex> lock(Q.mu);
    *deferred = 0;
    if (!lazy_on() || n == 0 || n > 0x7fffffffull) return flush_locked();
    Call c{};
    c.op = OP_PROJ; c.pkind = (uint8_t)kind; c.bclass = 1; c.nr = n; c.y = x;
    return push_call(c, deferred);
}

}  // namespace thip

extern "C" {

int thip_set_lazy_gemv(int on)
{
    THIP_NEED_INIT();                // runs what is pending
    std::lock_guard<std::mutex> lock(Q.mu);
    Q.enabled = on != 0;
    Q.env_read = true;
    return 0;
}

int thip_get_lazy_gemv(int *host_on)
{
    if (!host_on) return fail(THIP_E_INVALID, "null argument", __FILE__, __LINE__);
    std::lock_guard<std::mutex> lock(Q.mu);
    (void)lazy_on();                 // reads THIP_LAZY_GEMV once
    *host_on = Q.enabled ? 1 : 0;
    return 0;
}

int thip_lazy_gemv_stats(int64_t *host_deferred, int64_t *host_flushes)
{
    if (host_deferred) *host_deferred = Q.deferred;
    if (host_flushes) *host_flushes = Q.flushes;
    return 0;
}

int thip_lazy_read_stats(int64_t *host_served, int64_t *host_fetches)
{
    if (host_served) *host_served = Q.rd_served;
    if (host_fetches) *host_fetches = Q.rd_fetches;
    return 0;
}

int thip_lazy_plan_stats(int64_t *host_hits, int64_t *host_misses)
{
    if (host_hits) *host_hits = Q.hits;
    if (host_misses) *host_misses = Q.misses;
    return 0;
}

}  // extern "C"
