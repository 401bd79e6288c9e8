// totsu_f32hip.hpp -- header-only C++ host mirror of totsu_core's trait structure, layered ONLY on the C ABI
// (include/totsu_f32hip.h).  It exists so that a compiled-language host exercises the boundary the way the Rust crate
// would: `F32HIP` = LinAlg + LinAlgEx (solver/linalg.rs:10-68, linalg_ex.rs:7-66), `Slice` =
// SliceLike over device memory (solver/slicelike.rs:9-70), `Operator` / `MatOp` (solver/operator.rs:11-156,
// matop.rs:43-175), `Cone*` (cone_*.rs), `Solver` (solver/solver.rs:219-657).  Every arithmetic step is one ABI call.
// Citations are relative to /root/reference/solver_rust_conic/totsu_core/src/.
#pragma once

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "totsu_f32hip.h"

namespace totsu {

inline void chk(int rc)
{
    if (rc != 0) throw std::runtime_error(std::string("totsu_f32hip: ") + thip_last_error());   // backends assert on status
}

// ---- SliceLike: a window of device memory (children are pointer offsets) ----------------------------------------
struct Slice {
    float *p = nullptr;
    size_t n = 0;
    size_t len() const { return n; }
    std::pair<Slice, Slice> split(size_t mid) const { return { Slice{p, mid}, Slice{p + mid, n - mid} }; }
    Slice sub(size_t off, size_t cnt) const { return Slice{p + off, cnt}; }
    float get(size_t i) const { float v; chk(thip_get(p, i, &v)); return v; }           // slicelike.rs:54-59
    void set(size_t i, float v) const { chk(thip_set(p, i, v)); }                      // slicelike.rs:62-69
};

// owning device mirror of a host buffer: new_ref / new_mut + drop (download for mutable roots)
class DeviceVec {
public:
    explicit DeviceVec(size_t n) : n_(n) { chk(thip_alloc_zeroed(n, &d_)); }
    explicit DeviceVec(const std::vector<float> &h) : n_(h.size()) { chk(thip_alloc(n_, &d_)); chk(thip_h2d(d_, h.data(), n_)); }
    ~DeviceVec() { if (d_) thip_free(d_); }
    DeviceVec(const DeviceVec &) = delete;
    DeviceVec &operator=(const DeviceVec &) = delete;
    Slice slice() const { return Slice{d_, n_}; }
    std::vector<float> to_host() const { std::vector<float> h(n_); if (n_) chk(thip_d2h(h.data(), d_, n_)); return h; }
private:
    float *d_ = nullptr;
    size_t n_ = 0;
};

// ---- LinAlg + LinAlgEx -------------------------------------------------------------------------------------------
struct F32HIP {
    static float norm(Slice x) { float r; chk(thip_norm(x.n, x.p, &r)); return r; }
    static void copy(Slice x, Slice y) { chk(thip_copy(x.n, x.p, y.p)); }
    static void scale(float a, Slice x) { chk(thip_scale(x.n, a, x.p)); }
    static void add(float a, Slice x, Slice y) { chk(thip_add(x.n, a, x.p, y.p)); }
    static void adds(float s, Slice y) { chk(thip_adds(y.n, s, y.p)); }
    static float abssum(Slice x, size_t incx) { float r; chk(thip_abssum(x.n, x.p, incx, &r)); return r; }
    static void transform_di(float a, Slice d, Slice x, float b, Slice y) { chk(thip_transform_di(x.n, a, d.p, x.p, b, y.p)); }
    static void transform_ge(bool tr, size_t nr, size_t nc, float a, Slice mat, Slice x, float b, Slice y)
    { chk(thip_transform_ge(tr ? 1 : 0, nr, nc, a, mat.p, x.p, b, y.p)); }
    static void transform_sp(size_t n, float a, Slice mat, Slice x, float b, Slice y) { chk(thip_transform_sp(n, a, mat.p, x.p, b, y.p)); }
    static size_t map_eig_worklen(size_t n) { return thip_map_eig_worklen(n); }
};

// ---- Operator / MatOp ----------------------------------------------------------------------------------------------
struct Operator {
    virtual ~Operator() {}
    virtual std::pair<size_t, size_t> size() const = 0;
    virtual void op(float alpha, Slice x, float beta, Slice y) const = 0;
    virtual void trans_op(float alpha, Slice x, float beta, Slice y) const = 0;
    virtual void absadd_cols(Slice tau) const = 0;
    virtual void absadd_rows(Slice sigma) const = 0;
};

struct MatOp : Operator {                      // MatType::General, column-major (matop.rs:43-175)
    size_t nr, nc;
    Slice array;
    MatOp(size_t r, size_t c, Slice a) : nr(r), nc(c), array(a) {}
    std::pair<size_t, size_t> size() const override { return { nr, nc }; }
    void impl(bool tr, float alpha, Slice x, float beta, Slice y) const
    {
        if (nr > 0 && nc > 0) F32HIP::transform_ge(tr, nr, nc, alpha, array, x, beta, y);
        else F32HIP::scale(beta, y);                                                    // matop.rs:83-85
    }
    void op(float a, Slice x, float b, Slice y) const override { impl(false, a, x, b, y); }
    void trans_op(float a, Slice x, float b, Slice y) const override { impl(true, a, x, b, y); }
    void absadd_cols(Slice tau) const override { if (nr && nc) chk(thip_absadd_cols(nr, nc, array.p, tau.p)); }
    void absadd_rows(Slice sigma) const override { if (nr && nc) chk(thip_absadd_rows(nr, nc, array.p, sigma.p)); }
};

// A sparse operator held ONCE on the device (thip_sptile_*): a caller's own Operator in the pattern of
// examples/imgnr_udef/src/prob_op_a.rs:33-120, given by columns (host arrays: int64 column pointers, int32 row indices, f32 values)
struct SparseOp : Operator {
    size_t nr, nc;
    thip_sptile *mat = nullptr;
    SparseOp(size_t r, size_t c, const std::vector<int64_t> &colptr, const std::vector<int32_t> &rowidx, const std::vector<float> &vals)
        : nr(r), nc(c)
    {
        chk(thip_sptile_create(r, c, vals.size(), colptr.data(), rowidx.data(), vals.data(), &mat));
    }
    SparseOp(const SparseOp &) = delete;
    SparseOp &operator=(const SparseOp &) = delete;
    ~SparseOp() override { thip_sptile_destroy(mat); }
    std::pair<size_t, size_t> size() const override { return { nr, nc }; }
    void impl(bool tr, float alpha, Slice x, float beta, Slice y) const
    {
        if (nr > 0 && nc > 0) chk(thip_sptile_mv(mat, tr ? 1 : 0, alpha, x.p, beta, y.p, 0));
        else F32HIP::scale(beta, y);
    }
    void op(float a, Slice x, float b, Slice y) const override { impl(false, a, x, b, y); }                 // operator.rs:40-57
    void trans_op(float a, Slice x, float b, Slice y) const override { impl(true, a, x, b, y); }            // operator.rs:59-75
    void absadd_cols(Slice tau) const override { if (nr && nc) chk(thip_sptile_mv(mat, 1, 1.0f, tau.p, 1.0f, tau.p, 1)); }      // operator.rs:82-113
    void absadd_rows(Slice sigma) const override { if (nr && nc) chk(thip_sptile_mv(mat, 0, 1.0f, sigma.p, 1.0f, sigma.p, 1)); } // operator.rs:123-154
};

// ---- Cones -----------------------------------------------------------------------------------------------------------
struct Cone {
    virtual ~Cone() {}
    virtual bool proj(bool dual_cone, Slice x) = 0;                                     // false == Err(())
    virtual void product_group(Slice dp_tau) const = 0;                                 // applies the min-group closure
};
inline void group_min(Slice t)                                                          // solver.rs:509-520
{
    if (t.n == 0) return;
    const int64_t offs[2] = { 0, (int64_t)t.n };
    float *d = nullptr;
    chk(thip_alloc(4, &d));
    chk(thip_h2d(d, reinterpret_cast<const float *>(offs), 4));
    chk(thip_group_min_batched(t.p, reinterpret_cast<const int64_t *>(d), 1, t.n));
    chk(thip_free(d));
}
struct ConeZero : Cone {
    bool proj(bool dual, Slice x) override { chk(thip_proj_zero(dual ? 1 : 0, x.n, x.p)); return true; }
    void product_group(Slice) const override {}
};
struct ConeRPos : Cone {
    bool proj(bool, Slice x) override { chk(thip_proj_rpos(x.n, x.p)); return true; }
    void product_group(Slice) const override {}
};
struct ConeSOC : Cone {
    bool proj(bool, Slice x) override { chk(thip_proj_soc(x.n, x.p)); return true; }
    void product_group(Slice t) const override { group_min(t); }
};
struct ConeRotSOC : Cone {
    bool proj(bool, Slice x) override { chk(thip_proj_rotsoc(x.n, x.p)); return true; }
    void product_group(Slice t) const override { group_min(t); }
};
struct ConePSD : Cone {                                                                 // cone_psd.rs:22-85
    Slice work; float eps_zero;
    ConePSD(Slice w, float e) : work(w), eps_zero(e) {}
    static size_t query_worklen(size_t nvars)
    {
        const size_t n = (size_t)((std::sqrt((double)(8 * nvars + 1)) - 1.0) / 2.0 + 0.5);
        return thip_map_eig_worklen(n);
    }
    bool proj(bool, Slice x) override
    {
        if (work.n < query_worklen(x.n)) return false;
        chk(thip_proj_psd(x.n, x.p, eps_zero, work.p, work.n));
        return true;
    }
    void product_group(Slice t) const override { group_min(t); }
};
// consecutive blocks (ProbLPCone / ProbSOCPCone / ProbSDPCone shape)
struct ConeProduct : Cone {
    std::vector<std::pair<Cone *, size_t>> blocks;
    bool proj(bool dual, Slice x) override
    {
        size_t done = 0;
        for (auto &b : blocks) { if (!b.first->proj(dual, x.sub(done, b.second))) return false; done += b.second; }
        return true;
    }
    void product_group(Slice t) const override
    {
        size_t done = 0;
        for (auto &b : blocks) { b.first->product_group(t.sub(done, b.second)); done += b.second; }
    }
};

// ---- Solver (solver.rs) -------------------------------------------------------------------------------------------
struct SolverParam {                                                                    // solver.rs:13-41
    int64_t max_iter = -1;                 // < 0: None
    float eps_acc = 1e-6f, eps_inf = 1e-6f, eps_zero = 1e-12f;
    int64_t log_period = 10000;
    int32_t state_arith = THIP_STATE_COMPENSATED;   // fused loop only (not in the reference): THIP_STATE_*
};
enum class SolverError { Ok = 0, Unbounded, Infeasible, ExcessIter, InvalidOp, WorkShortage, ConeFailure };   // solver_error.rs:3-17

// The trait-level loop issues one C-ABI call per reference call; it opts into the library's deferred, batched execution
// of the small ones (thip_set_lazy_gemv) for the duration of a solve and restores the caller's setting afterwards.
struct LazyCalls {
    int prev = 0;
    LazyCalls() { chk(thip_get_lazy_gemv(&prev)); chk(thip_set_lazy_gemv(1)); }
    ~LazyCalls() { thip_set_lazy_gemv(prev); }
    LazyCalls(const LazyCalls &) = delete;
    LazyCalls &operator=(const LazyCalls &) = delete;
};

class Solver {
public:
    SolverParam par;
    int64_t iters = -1;
    static size_t query_worklen(size_t m, size_t n)                                     // solver.rs:231-249
    { return (n + 2 * m + 1) * 4 + (n + m + 1) * 2; }

    // work: device slice of query_worklen floats; on Ok the answers are work[0..n) and work[n..n+m) (solver.rs:317-320)
    SolverError solve(const Operator &c, const Operator &a, const Operator &b, Cone &cone, Slice work)
    {
        const size_t m = a.size().first, n = a.size().second;
        if (c.size() != std::make_pair(n, (size_t)1) || b.size() != std::make_pair(m, (size_t)1)) return SolverError::InvalidOp;
        if (query_worklen(m, n) > work.n) return SolverError::WorkShortage;
        const size_t N = n + 2 * m + 1, M = n + m + 1;
        LazyCalls lazy_scope;
        DeviceVec one(1);
        // calc_norms, solver.rs:460-481 (fr_norm of a single column)
        const float norm_b = fr_norm(b, one.slice(), work.sub(0, m));
        const float norm_c = fr_norm(c, one.slice(), work.sub(0, n));
        Slice x = work.sub(0, N), y = work.sub(N, M), dpt = work.sub(N + M, N), dps = work.sub(2 * N + M, M), tmpw = work.sub(2 * N + 2 * M, 2 * N);
        F32HIP::scale(0.f, x); F32HIP::scale(0.f, y); x.set(n + 2 * m, 1.f);            // init_vecs, solver.rs:483-494
        // calc_precond, solver.rs:496-524
        abssum(c, a, b, m, n, dpt, dps);
        chk(thip_recip_max(N, par.eps_zero, dpt.p)); chk(thip_recip_max(M, par.eps_zero, dps.p));
        cone.product_group(dpt.sub(n, m)); cone.product_group(dpt.sub(n + m, m));
        for (int64_t i = 0;; ++i) {
            const bool excess = par.max_iter >= 0 ? (i + 1 >= par.max_iter) : false;
            // update_vecs, solver.rs:526-571
            Slice rx = tmpw.sub(0, N), tx = tmpw.sub(N, N);
            F32HIP::copy(x, rx);
            k_trans_op(c, a, b, m, n, -1.f, y, tx);
            F32HIP::transform_di(1.f, dpt, tx, 1.f, x);
            if (!cone.proj(true, x.sub(n, m)) || !cone.proj(false, x.sub(n + m, m))) return SolverError::ConeFailure;
            const float tau = std::fmax(x.get(n + 2 * m), 0.f);
            x.set(n + 2 * m, tau);
            F32HIP::add(-2.f, x, rx);
            Slice ty = tx.sub(0, M);
            k_op(c, a, b, m, n, -1.f, rx, ty);
            F32HIP::transform_di(1.f, dps, ty, 1.f, y);
            y.set(n + m, std::fmin(y.get(n + m), 0.f));
            iters = i;
            Slice p = tmpw.sub(0, m), d = tmpw.sub(m, n);
            if (tau > par.eps_zero) {                                                   // criteria_conv, solver.rs:573-612
                const float rt = 1.f / tau;
                one.slice().set(0, 1.f);
                F32HIP::copy(x.sub(n + m, m), p);
                b.op(-1.f, one.slice(), rt, p);
                a.op(rt, x.sub(0, n), 1.f, p);
                c.op(1.f, one.slice(), 0.f, d);
                a.trans_op(rt, x.sub(n, m), 1.f, d);
                c.trans_op(rt, x.sub(0, n), 0.f, one.slice()); const float gx = one.slice().get(0);
                b.trans_op(rt, x.sub(n, m), 0.f, one.slice()); const float gy = one.slice().get(0);
                const float pri = F32HIP::norm(p) / (1.f + norm_b), dual = F32HIP::norm(d) / (1.f + norm_c);
                const float gap = std::fabs(gx + gy) / (1.f + std::fabs(gx) + std::fabs(gy));
                const bool conv = pri <= par.eps_acc && dual <= par.eps_acc && gap <= par.eps_acc;
                if (excess || conv) {
                    F32HIP::scale(rt, x.sub(0, n)); F32HIP::scale(rt, x.sub(n, m));
                    return conv ? SolverError::Ok : SolverError::ExcessIter;
                }
            } else {                                                                    // criteria_inf, solver.rs:614-656
                one.slice().set(0, 0.f);
                F32HIP::copy(x.sub(n + m, m), p);
                a.op(1.f, x.sub(0, n), 1.f, p);
                a.trans_op(1.f, x.sub(n, m), 0.f, d);
                c.trans_op(-1.f, x.sub(0, n), 0.f, one.slice()); const float mcx = one.slice().get(0);
                b.trans_op(-1.f, x.sub(n, m), 0.f, one.slice()); const float mby = one.slice().get(0);
                const float unbdd = mcx > par.eps_zero ? F32HIP::norm(p) * norm_c / mcx : INFINITY;
                const float infeas = mby > par.eps_zero ? F32HIP::norm(d) * norm_b / mby : INFINITY;
                const bool tu = unbdd <= par.eps_inf, ti = infeas <= par.eps_inf;
                if (excess || tu || ti) return tu ? SolverError::Unbounded : (ti ? SolverError::Infeasible : SolverError::ExcessIter);
            }
        }
    }

private:
    static float fr_norm(const Operator &o, Slice v1, Slice t)                          // solver.rs:85-107, one column
    {
        v1.set(0, 1.f);
        o.op(1.f, v1, 0.f, t);
        const float nn = F32HIP::norm(t);
        v1.set(0, 0.f);
        return std::sqrt(nn * nn);
    }
    static void k_op(const Operator &c, const Operator &a, const Operator &b, size_t m, size_t n, float alpha, Slice x, Slice y)
    {                                                                                   // SelfDualEmbed::op, solver.rs:109-131 (beta = 0)
        Slice xx = x.sub(0, n), xy = x.sub(n, m), xs = x.sub(n + m, m), xt = x.sub(n + 2 * m, 1);
        Slice yn = y.sub(0, n), ym = y.sub(n, m), y1 = y.sub(n + m, 1);
        a.trans_op(alpha, xy, 0.f, yn); c.op(alpha, xt, 1.f, yn);
        a.op(-alpha, xx, 0.f, ym); F32HIP::add(-alpha, xs, ym); b.op(alpha, xt, 1.f, ym);
        c.trans_op(-alpha, xx, 0.f, y1); b.trans_op(-alpha, xy, 1.f, y1);
    }
    static void k_trans_op(const Operator &c, const Operator &a, const Operator &b, size_t m, size_t n, float alpha, Slice x, Slice y)
    {                                                                                   // SelfDualEmbed::trans_op, solver.rs:133-157 (beta = 0)
        Slice xn = x.sub(0, n), xm = x.sub(n, m), x1 = x.sub(n + m, 1);
        Slice yx = y.sub(0, n), yy = y.sub(n, m), ys = y.sub(n + m, m), yt = y.sub(n + 2 * m, 1);
        a.trans_op(-alpha, xm, 0.f, yx); c.op(-alpha, x1, 1.f, yx);
        a.op(alpha, xn, 0.f, yy); b.op(-alpha, x1, 1.f, yy);
        F32HIP::scale(0.f, ys); F32HIP::add(-alpha, xm, ys);
        c.trans_op(alpha, xn, 0.f, yt); b.trans_op(alpha, xm, 1.f, yt);
    }
    static void abssum(const Operator &c, const Operator &a, const Operator &b, size_t m, size_t n, Slice tau, Slice sigma)
    {                                                                                   // solver.rs:159-183
        F32HIP::scale(0.f, tau);
        Slice tx = tau.sub(0, n), ty = tau.sub(n, m), ts = tau.sub(n + m, m), tt = tau.sub(n + 2 * m, 1);
        a.absadd_cols(tx); c.absadd_rows(tx); a.absadd_rows(ty); b.absadd_rows(ty);
        F32HIP::adds(1.f, ts); c.absadd_cols(tt); b.absadd_cols(tt);
        F32HIP::copy(tx, sigma.sub(0, n)); F32HIP::copy(ty, sigma.sub(n, m));
        F32HIP::add(1.f, ts, sigma.sub(n, m)); F32HIP::copy(tt, sigma.sub(n + m, 1));
    }
};

// ---- FusedSolver: RAII over the device-resident loop (thip_solver_*), the drop-in for Solver::solve when A is a dense
// MatOp and the cone a product of Zero / RPos / SOC / RotSOC / PSD segments over consecutive rows ----------------------
class FusedSolver {
public:
    // a, b, c: device slices (A column-major m x n); seg_type: THIP_CONE_*, seg_len: rows of each segment
    FusedSolver(size_t n, size_t m, Slice a, Slice b, Slice c, const std::vector<int32_t> &seg_type,
                const std::vector<int64_t> &seg_len, const SolverParam &par, int schedule = THIP_SCHED_SWEEP)
        : n_(n), m_(m), seg_type_(seg_type), seg_len_(seg_len)
    {
        thip_problem prob{};
        prob.n = n; prob.m = m; prob.mat_a = a.p; prob.vec_b = b.p; prob.vec_c = c.p; prob.vec_b_rowabs = nullptr;
        prob.n_seg = seg_type_.size(); prob.host_seg_type = seg_type_.data(); prob.host_seg_len = seg_len_.data();
        const thip_param p = to_c(par);
        chk(thip_solver_create(&prob, &p, schedule, &h_));
    }
    FusedSolver(const FusedSolver &) = delete;
    FusedSolver &operator=(const FusedSolver &) = delete;
    ~FusedSolver() { thip_solver_destroy(h_); }

    // THIP_A_F32 / THIP_A_BF16 / THIP_A_F16; before init(), or between run() calls (then resume() if it had finished)
    void set_a_storage(int kind) { chk(thip_solver_set_a_storage(h_, kind)); }
    // THIP_SCHED_SWEEP (the default schedule) runs for matrices of at least this many bytes; before init()
    void set_sweep_min_bytes(size_t bytes) { chk(thip_solver_set_sweep_min_bytes(h_, bytes)); }
    // N > 1: this solver holds a block of COLUMNS (a, c: the block; b, segments: the whole problem's); needs an all-reduce
    // hook (thip_solver_use_rccl / _use_oneshot / _set_allreduce on handle()); before init()
    void set_column_shard(bool on) { chk(thip_solver_set_column_shard(h_, on ? 1 : 0)); }
    // the schedule the next run() executes: THIP_SCHED_CARRIED when THIP_SCHED_SWEEP cannot take the problem
    int schedule_in_use() { int v = 0; chk(thip_solver_schedule_in_use(h_, &v)); return v; }
    // recoveries from a one-pass kernel that gave up (the run restored its snapshot and went on with the 2-pass schedule)
    int sweep_faults(int *last_word = nullptr, int64_t *restored_iter = nullptr)
    {
        int k = 0, w = 0; int64_t it = -1;
        chk(thip_solver_sweep_faults(h_, &k, &w, &it));
        if (last_word) *last_word = w;
        if (restored_iter) *restored_iter = it;
        return k;
    }
    thip_solver *handle() { return h_; }
    void init() { chk(thip_solver_init(h_)); inited_ = true; }
    // runs until termination (max_steps < 0) or for max_steps iterations; returns the reference's SolverError
    SolverError run(int64_t max_steps = -1, int64_t poll_every = 64)
    {
        if (!inited_) init();
        chk(thip_solver_run(h_, max_steps, poll_every, &st_));
        return st_.state <= 0 ? SolverError::Ok : (SolverError)st_.state;     // THIP_ST_RUNNING (-1) reads as "not failed yet"
    }
    bool running() const { return st_.state == THIP_ST_RUNNING; }
    void resume(const SolverParam *par = nullptr)
    {
        if (par) { const thip_param p = to_c(*par); chk(thip_solver_set_param(h_, &p)); }
        chk(thip_solver_resume(h_));
    }
    int64_t iters() const { return st_.iter; }
    const thip_status &status() const { return st_; }
    void solution(std::vector<float> &x, std::vector<float> &y)               // solver.rs:317-320
    {
        x.resize(n_); y.resize(m_);
        chk(thip_solver_solution(h_, x.data(), y.data()));
    }

private:
    static thip_param to_c(const SolverParam &par)
    {
        thip_param p{};
        p.max_iter = par.max_iter; p.eps_acc = par.eps_acc; p.eps_inf = par.eps_inf; p.eps_zero = par.eps_zero;
        p.log_period = 0;
        p.state_arith = par.state_arith;
        return p;
    }
    size_t n_, m_;
    std::vector<int32_t> seg_type_;
    std::vector<int64_t> seg_len_;
    thip_solver *h_ = nullptr;
    thip_status st_{};
    bool inited_ = false;
};

}  // namespace totsu
