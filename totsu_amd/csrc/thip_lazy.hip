// thip_lazy.hip -- deferred, batched execution of the small calls of a composite operator: the "grouped GEMV over a
// descriptor table" for the TRAIT-LEVEL path (hard parts).
//
// An unchanged totsu drives a composite operator block by block: ProbSOCPOpA::op / ::trans_op (totsu/src/problem/
// socp.rs:77-130) issue one `LinAlgEx::transform_ge` per c_i and per G_i, ProbSOCPOpB (socp.rs:194-246) a `scale`, an
// `add` and a `transform_ge` per cone -- 2000 + 3000 calls per K product at BASELINE configs[2], tens of thousands per
// iteration.  One launch each is launch-bound (0.75 iter/s measured).  The trait surface gives no handle on the loops,
// but nothing OBSERVES a result until some other call reads it.  So thip_transform_ge (matrices <= 64 MB) and
// thip_scale / thip_add (vectors <= 1024 long) only RECORD their call; the record is run -- one grouped launch per
// kind of product plus one finishing launch per class of output -- when any other entry point is called
// (THIP_NEED_INIT), when a new call would read or overwrite what a pending one writes, or when it is full.  Stream order
// therefore still equals call order as far as any caller can tell.
//
// Every recorded call has the form  y <- beta y + (a contribution)  on one output vector y:
//   N     alpha A x               matrix, nr, nc > 1                 (partial sums over column chunks)
//   T     alpha A^T x                                                (partial sums over row tiles)
//   AXPY  alpha v x[0]            op of a column vector / trans_op of a row vector
//   DOT   alpha v . x             trans_op of a column vector / op of a row vector  (y is one number)
//   ADDV  alpha x                 LinAlg::add
//   SCALE (nothing)               LinAlg::scale: beta only
// Calls on the SAME y compose into one group  y <- B y + sum_k a_k c_k : a later call with factor beta multiplies B and
// every a_k recorded so far.  ProbSOCPOpA::trans_op (one scale + 2000 contributions into one n-vector) and
// ProbSOCPOpB::trans_op (one scale + 2000 into one number) each become ONE group, summed in a fixed order (deterministic;
// the order differs from the reference's sequential additions in the last bits only).
#include "thip_common.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

using namespace thip;

namespace {

constexpr int BLK = 256;
constexpr size_t LAZY_MAX_ELEMS = (size_t)16 << 20;     // products of matrices above 64 MB run at once (their own launch pays)
constexpr size_t LAZY_MAX_VEC = 1024;                   // scale / add on longer vectors run at once
constexpr size_t LAZY_MAX_OPS = 32768;
constexpr int SHORT_LEN = 2048;                         // groups up to this long are finished by one workgroup each

enum { K_N = 0, K_T = 1, K_AXPY = 2, K_DOT = 3, K_ADDV = 4, K_SCALE = 5 };

struct Member {
    int kind; size_t nr, nc, inlen;
    float alpha;
    const float *A, *x;            // A: matrix / the vector v ; x: the input vector (ADDV: x only)
};
struct Group {
    float *y; size_t len; float beta;
    std::vector<Member> mem;
};

// device-side tables
struct DotD { const float *v, *x; float *out; int len; int pad; };
enum { M_PART = 0, M_AXPY = 1, M_ADDV = 2 };
struct FinMember { const float *src; const float *xs; int count; int type; float alpha; int pad; };
struct FinGroup { float *y; int len; int first, count; float beta; };

struct Queue {
    std::vector<Group> groups;
    std::unordered_map<const float *, int> target;         // y -> group
    size_t n_members = 0;
    uintptr_t xlo = ~(uintptr_t)0, xhi = 0, ylo = ~(uintptr_t)0, yhi = 0;
    std::mutex mu;
    bool enabled = false, env_read = false;     // OFF unless a host asks for it (thip_set_lazy_gemv) or THIP_LAZY_GEMV=1
    char *dev = nullptr; size_t dev_bytes = 0;             // partial sums and tables
    // pinned staging of the tables: two halves, an event each ("the upload out of this half has finished")
    char *pin[2] = { nullptr, nullptr }; size_t pin_bytes[2] = { 0, 0 }; hipEvent_t pin_ev[2] = { nullptr, nullptr };
    int pin_next = 0;
    long long flushes = 0, deferred = 0;
    std::atomic<bool> pending{ false };                     // read without the lock by every entry point
} Q;

bool overlap(uintptr_t a0, uintptr_t a1, uintptr_t b0, uintptr_t b1) { return a0 < b1 && b0 < a1; }

// out[0] = v . x   (one block per product; the factor is applied by the finishing kernel)
__global__ __launch_bounds__(BLK) void dot_k(const DotD *__restrict__ tab)
{
    __shared__ double shd[16];
    const DotD d = tab[blockIdx.x];
    double s = 0.0;
    for (int i = threadIdx.x; i < d.len; i += BLK) s += (double)d.v[i] * (double)d.x[i];
    s = block_sum_d(s, shd);
    if (threadIdx.x == 0) d.out[0] = (float)s;
}

__device__ __forceinline__ double contributions(const FinGroup &g, const FinMember *__restrict__ mem, int c)
{
    double acc = 0.0;
    for (int k = 0; k < g.count; ++k) {
        const FinMember m = mem[g.first + k];
        if (m.type == M_PART) {
            double s = 0.0;
            for (int t = 0; t < m.count; ++t) s += (double)m.src[(size_t)t * g.len + c];
            acc += (double)m.alpha * s;
        } else if (m.type == M_AXPY) acc += (double)(m.alpha * m.xs[0] * m.src[c]);
        else acc += (double)(m.alpha * m.src[c]);
    }
    return acc;
}

// y[c] = beta y[c] + sum of the group's contributions.  LONG: grid (len / 256, groups); else one workgroup per group
template <bool LONG>
__global__ __launch_bounds__(BLK) void fin_k(const FinGroup *__restrict__ groups, const FinMember *__restrict__ mem)
{
    const FinGroup g = groups[LONG ? blockIdx.y : blockIdx.x];
    if (LONG) {
        const int c = blockIdx.x * BLK + threadIdx.x;
        if (c >= g.len) return;
        const float a = (float)contributions(g, mem, c);
        g.y[c] = g.beta == 0.0f ? a : fmaf(g.beta, g.y[c], a);
    } else {
        for (int c = threadIdx.x; c < g.len; c += BLK) {
            const float a = (float)contributions(g, mem, c);
            g.y[c] = g.beta == 0.0f ? a : fmaf(g.beta, g.y[c], a);
        }
    }
}

int ensure_dev(size_t bytes)
{
    if (bytes <= Q.dev_bytes) return 0;
    if (Q.dev) { THIP_TRY(hipStreamSynchronize(ctx().stream)); THIP_TRY(hipFree(Q.dev)); Q.dev = nullptr; Q.dev_bytes = 0; }
    const size_t want = bytes + bytes / 4 + (1 << 20);
    THIP_TRY(hipMalloc((void **)&Q.dev, want));
    Q.dev_bytes = want;
    return 0;
}

size_t up256(size_t v) { return (v + 255) / 256 * 256; }

void reset_queue()
{
    Q.pending.store(false, std::memory_order_relaxed);
    Q.groups.clear(); Q.target.clear(); Q.n_members = 0;
    Q.xlo = Q.ylo = ~(uintptr_t)0; Q.xhi = Q.yhi = 0;
}

int flush_locked()
{
    if (Q.groups.empty()) return 0;
    hipStream_t st = ctx().stream;
    std::vector<Group> groups;
    groups.swap(Q.groups);
    const size_t n_members = Q.n_members;
    reset_queue();
    Q.flushes += 1;

    // ---- layout of the device buffer: partial sums | dot results | tables ----
    size_t floats = 0;
    int nN = 0, nT = 0, nDot = 0, nLong = 0, nShort = 0, maxlen = 0;
    for (const Group &g : groups) {
        ((int)g.len > SHORT_LEN ? nLong : nShort) += 1;
        if ((int)g.len > SHORT_LEN) maxlen = std::max(maxlen, (int)g.len);
        for (const Member &m : g.mem) {
            if (m.kind == K_N) nN += 1; else if (m.kind == K_T) nT += 1; else if (m.kind == K_DOT) nDot += 1;
        }
    }
    struct MatPlan { size_t off; int cpc, tiles, chunks; };
    std::vector<MatPlan> plan;
    plan.reserve(nN + nT);
    std::vector<size_t> plan_base(groups.size() + 1, 0);
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        for (const Member &m : groups[gi].mem) {
            if (m.kind != K_N && m.kind != K_T) continue;
            // the unit of the grouped kernel is a wave: 128 rows x cpc columns, ~0.1-0.2 MB of the matrix each
            // (8 .. 1024 columns per chunk); a workgroup = four consecutive chunks
            size_t c = 40000 / (m.nr < 128 ? m.nr : 128);
            c = std::max<size_t>(8, std::min<size_t>(1024, c / 8 * 8));
            MatPlan p{ floats, (int)c, (int)((m.nr + 127) / 128), (int)((m.nc + c - 1) / c) };
            floats += m.kind == K_N ? (size_t)p.chunks * m.nr : (size_t)p.tiles * m.nc;
            floats = (floats + 63) / 64 * 64;
            plan.push_back(p);
        }
        plan_base[gi + 1] = plan.size();
    }
    const size_t dot_off = floats;
    floats += (size_t)nDot;
    const size_t b_part = up256(floats * sizeof(float));
    const size_t b_gd = up256((size_t)(nN + nT) * sizeof(GroupDesc));
    const size_t b_dot = up256((size_t)nDot * sizeof(DotD));
    const size_t b_grp = up256(groups.size() * sizeof(FinGroup));
    const size_t b_mem = up256(n_members * sizeof(FinMember));
    const size_t b_tab = b_gd + b_dot + b_grp + b_mem;
    THIP_RC(ensure_dev(b_part + b_tab));
    float *dpart = reinterpret_cast<float *>(Q.dev);
    char *dtab = Q.dev + b_part;

    const int half = Q.pin_next;
    Q.pin_next ^= 1;
    if (Q.pin_ev[half] == nullptr) THIP_TRY(hipEventCreateWithFlags(&Q.pin_ev[half], hipEventDisableTiming));
    else THIP_TRY(hipEventSynchronize(Q.pin_ev[half]));          // the previous upload out of this half is done
    if (Q.pin_bytes[half] < b_tab) {
        if (Q.pin[half]) THIP_TRY(hipHostFree(Q.pin[half]));
        Q.pin_bytes[half] = b_tab + b_tab / 4 + 4096;
        THIP_TRY(hipHostMalloc((void **)&Q.pin[half], Q.pin_bytes[half], hipHostMallocDefault));
    }
    char *host = Q.pin[half];
    memset(host, 0, b_tab);
    GroupDesc *gdN = reinterpret_cast<GroupDesc *>(host);
    GroupDesc *gdT = gdN + nN;
    DotD *dd = reinterpret_cast<DotD *>(host + b_gd);
    FinGroup *hg = reinterpret_cast<FinGroup *>(host + b_gd + b_dot);
    FinMember *hm = reinterpret_cast<FinMember *>(host + b_gd + b_dot + b_grp);

    // long groups first, then the short ones (each class is one launch over a contiguous range of the table)
    std::vector<int> order;
    order.reserve(groups.size());
    for (size_t i = 0; i < groups.size(); ++i) if ((int)groups[i].len > SHORT_LEN) order.push_back((int)i);
    for (size_t i = 0; i < groups.size(); ++i) if ((int)groups[i].len <= SHORT_LEN) order.push_back((int)i);
    int iN = 0, iT = 0, iD = 0, maxtN = 0, maxcN = 0, maxtT = 0, maxcT = 0;
    size_t mpos = 0;
    for (size_t oi = 0; oi < order.size(); ++oi) {
        const Group &g = groups[order[oi]];
        size_t pk = plan_base[order[oi]];
        hg[oi] = FinGroup{ g.y, (int)g.len, (int)mpos, (int)g.mem.size(), g.beta };
        for (const Member &m : g.mem) {
            FinMember fm{ nullptr, nullptr, 0, M_ADDV, m.alpha, 0 };
            if (m.kind == K_N || m.kind == K_T) {
                const MatPlan &p = plan[pk++];
                GroupDesc gd{ m.A, m.x, dpart + p.off, (int)m.nr, (int)m.nc, p.cpc, 0 };
                if (m.kind == K_N) { gdN[iN++] = gd; maxtN = std::max(maxtN, p.tiles); maxcN = std::max(maxcN, p.chunks); }
                else               { gdT[iT++] = gd; maxtT = std::max(maxtT, p.tiles); maxcT = std::max(maxcT, p.chunks); }
                fm.src = dpart + p.off; fm.count = m.kind == K_N ? p.chunks : p.tiles; fm.type = M_PART;
            } else if (m.kind == K_DOT) {
                dd[iD] = DotD{ m.A, m.x, dpart + dot_off + iD, (int)m.inlen, 0 };
                fm.src = dpart + dot_off + iD; fm.type = M_ADDV;
                ++iD;
            } else if (m.kind == K_AXPY) { fm.src = m.A; fm.xs = m.x; fm.type = M_AXPY; }
            else { fm.src = m.x; fm.type = M_ADDV; }                                          // K_ADDV
            hm[mpos++] = fm;
        }
    }
    THIP_TRY(hipMemcpyAsync(dtab, host, b_tab, hipMemcpyHostToDevice, st));
    THIP_TRY(hipEventRecord(Q.pin_ev[half], st));

    const GroupDesc *dgd = reinterpret_cast<const GroupDesc *>(dtab);
    THIP_RC(grouped_gemv(st, dgd, nN, maxtN, (maxcN + 3) / 4, false));
    THIP_RC(grouped_gemv(st, dgd + nN, nT, maxtT, (maxcT + 3) / 4, true));
    if (nDot) hipLaunchKernelGGL(dot_k, dim3(nDot), dim3(BLK), 0, st, reinterpret_cast<const DotD *>(dtab + b_gd));
    const FinGroup *dg = reinterpret_cast<const FinGroup *>(dtab + b_gd + b_dot);
    const FinMember *dm = reinterpret_cast<const FinMember *>(dtab + b_gd + b_dot + b_grp);
    if (nLong) hipLaunchKernelGGL(fin_k<true>, dim3((maxlen + BLK - 1) / BLK, (unsigned)nLong), dim3(BLK), 0, st, dg, dm);
    if (nShort) hipLaunchKernelGGL(fin_k<false>, dim3((unsigned)nShort), dim3(BLK), 0, st, dg + nLong, dm);
    THIP_LAUNCH_CHECK();
    return 0;
}

// records  y(len) <- beta y + [member]  (member.kind == K_SCALE: no contribution)
int push_locked(float *y, size_t len, float beta, const Member &m, int *deferred)
{
    const bool has_in = m.kind != K_SCALE;
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
    if (must_flush) { THIP_RC(flush_locked()); join = -1; }
    if (join < 0) {
        Q.target[y] = (int)Q.groups.size();
        Q.groups.push_back(Group{ y, len, 1.0f, {} });
        join = (int)Q.groups.size() - 1;
        Q.ylo = std::min(Q.ylo, y0); Q.yhi = std::max(Q.yhi, y1);
    }
    Group &g = Q.groups[join];
    if (beta != 1.0f) {                 // y <- beta (B y + sum a_k c_k) + ...
        if (beta == 0.0f) { Q.n_members -= g.mem.size(); g.mem.clear(); g.beta = 0.0f; }
        else { g.beta *= beta; for (Member &p : g.mem) p.alpha *= beta; }
    }
    if (m.kind != K_SCALE) {
        g.mem.push_back(m);
        Q.n_members += 1;
        Q.xlo = std::min(Q.xlo, x0); Q.xhi = std::max(Q.xhi, x1);
        if (has_a) { Q.xlo = std::min(Q.xlo, a0); Q.xhi = std::max(Q.xhi, a1); }
    }
    Q.deferred += 1;
    Q.pending.store(true, std::memory_order_relaxed);
    *deferred = 1;
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

void lazy_release()
{
    std::lock_guard<std::mutex> lock(Q.mu);
    reset_queue();
    if (Q.dev) { hipFree(Q.dev); Q.dev = nullptr; Q.dev_bytes = 0; }
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
    if (!lazy_on() || n_row * n_col > LAZY_MAX_ELEMS || n_row > 0x7fffffffull || n_col > 0x7fffffffull)
        return flush_locked();                  // runs now, after everything recorded so far
    Member m;
    m.nr = n_row; m.nc = n_col; m.alpha = alpha; m.A = mat; m.x = x;
    size_t outlen;
    if (n_col == 1)      { m.kind = transpose ? K_DOT : K_AXPY; m.inlen = transpose ? n_row : 1; outlen = transpose ? 1 : n_row; }
    else if (n_row == 1) { m.kind = transpose ? K_AXPY : K_DOT; m.inlen = transpose ? 1 : n_col; outlen = transpose ? n_col : 1; }
    else                 { m.kind = transpose ? K_T : K_N; m.inlen = transpose ? n_row : n_col; outlen = transpose ? n_col : n_row; }
    return push_locked(y, outlen, beta, m, deferred);
}

// LinAlg::scale (x <- alpha x) and LinAlg::add (y <- alpha x + y) on short vectors join the record
int lazy_push_scale(size_t n, float alpha, float *x, int *deferred)
{
    std::lock_guard<std::mutex> lock(Q.mu);
    *deferred = 0;
    if (alpha == 1.0f) { *deferred = 1; return 0; }          // x <- 1 x: nothing to do, nothing to order (any length)
    if (!lazy_on() || n > LAZY_MAX_VEC || n == 0) return flush_locked();
    Member m{};
    m.kind = K_SCALE;
    return push_locked(x, n, alpha, m, deferred);
}

int lazy_push_add(size_t n, float alpha, const float *x, float *y, int *deferred)
{
    std::lock_guard<std::mutex> lock(Q.mu);
    *deferred = 0;
    if (!lazy_on() || n > LAZY_MAX_VEC || n == 0) return flush_locked();
    Member m{};
    m.kind = K_ADDV; m.alpha = alpha; m.x = x; m.inlen = n; m.nr = m.nc = 0;
    return push_locked(y, n, 1.0f, m, deferred);
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

}  // extern "C"
