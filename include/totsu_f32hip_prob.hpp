// totsu_f32hip_prob.hpp -- C++ host mirror of totsu's problem builders for the MI355X backend, layered on
// totsu_f32hip.hpp (itself layered only on the C ABI): `MatBuild` (totsu/src/matbuild/mod.rs:11-40, General only),
// `ProbLP` (totsu/src/problem/lp.rs:222-338) and `ProbSOCP` (socp.rs:336-474) with their composite operators and cones,
// restated block for block, and the two things SURVEY.md 8(f1) asks of the binding crate:
//   * `HipProbLP` / `HipProbSOCP` / `HipProbSDP`: the same builders (same constructor arguments, same `problem()`), whose `problem()`
//     additionally carries a DENSE DESCRIPTION of itself -- so that `solve(Solver &, Problem &)` can hand the problem
//     to the device-resident loop (FusedSolver) instead of driving it call by call: an unchanged caller gets the fused
//     rate by changing one type name;
//   * pointer-keyed de-duplication of device mirrors: `MatBuild::as_op()` is called twice per G_i by ProbSOCP::problem
//     (socp.rs:450,463) and once per solve by every builder; the mirror of a host array is uploaded once and shared
//     (`MirrorCache`), where totsu_f32cuda uploads on every `new_ref` (f32cuda_slice.rs:89-113).
// Citations are relative to /root/reference/solver_rust_conic/.
#pragma once

#include <map>
#include <memory>

#include "totsu_f32hip.hpp"

namespace totsu {

// ---- MatBuild (General, column-major) ---------------------------------------------------------------------------------
struct MatBuild {
    size_t nr = 0, nc = 0;
    std::vector<float> array;                      // column-major, lda = nr (linalg_ex.rs:15-16)
    MatBuild() {}
    MatBuild(size_t r, size_t c) : nr(r), nc(c), array(r * c, 0.f) {}
    std::pair<size_t, size_t> size() const { return { nr, nc }; }
    float &at(size_t r, size_t c) { return array[c * nr + r]; }
    float at(size_t r, size_t c) const { return array[c * nr + r]; }
    MatBuild &iter_colmaj(const std::vector<float> &v) { array.assign(v.begin(), v.end()); array.resize(nr * nc, 0.f); return *this; }
};

// device mirrors of host arrays, keyed by the host pointer: one upload per array however many operators view it
class MirrorCache {
public:
    Slice get(const MatBuild &mb)
    {
        if (mb.array.empty()) return Slice{ nullptr, 0 };
        auto it = map_.find(mb.array.data());
        if (it == map_.end()) {
            it = map_.emplace(mb.array.data(), std::unique_ptr<DeviceVec>(new DeviceVec(mb.array))).first;
            ++uploads;
        }
        return it->second->slice();
    }
    size_t uploads = 0;
private:
    std::map<const float *, std::unique_ptr<DeviceVec>> map_;
};

// ---- composite operators ----------------------------------------------------------------------------------------------
// a column vector as Operator: ProbLPOpC / ProbSOCPOpC (lp.rs:11-46)
struct OpVec : Operator {
    MatOp vec;
    explicit OpVec(MatOp v) : vec(v) {}
    std::pair<size_t, size_t> size() const override { return vec.size(); }
    void op(float a, Slice x, float b, Slice y) const override { vec.op(a, x, b, y); }
    void trans_op(float a, Slice x, float b, Slice y) const override { vec.trans_op(a, x, b, y); }
    void absadd_cols(Slice t) const override { vec.absadd_cols(t); }
    void absadd_rows(Slice s) const override { vec.absadd_rows(s); }
};

// two row-stacked blocks sharing the columns: ProbLPOpA (lp.rs:50-115), ProbLPOpB (lp.rs:119-191), ProbSDPOpA
// (sdp.rs:49-114), ProbSDPOpB (sdp.rs:118-190: the first block enters with -alpha, sf = -1)
struct OpStack2 : Operator {
    MatOp first, second;
    float sf;
    OpStack2(MatOp f, MatOp s, float sign_first = 1.f) : first(f), second(s), sf(sign_first) {}
    std::pair<size_t, size_t> size() const override { return { first.nr + second.nr, first.nc }; }
    void op(float a, Slice x, float b, Slice y) const override
    {
        first.op(sf * a, x, b, y.sub(0, first.nr));
        second.op(a, x, b, y.sub(first.nr, second.nr));
    }
    void trans_op(float a, Slice x, float b, Slice y) const override
    {
        first.trans_op(sf * a, x.sub(0, first.nr), b, y);
        second.trans_op(a, x.sub(first.nr, second.nr), 1.f, y);
    }
    void absadd_cols(Slice t) const override { first.absadd_cols(t); second.absadd_cols(t); }
    void absadd_rows(Slice s) const override { first.absadd_rows(s.sub(0, first.nr)); second.absadd_rows(s.sub(first.nr, second.nr)); }
};

// ProbSOCPOpA (socp.rs:49-163): per cone i the rows [-c_i^T ; -G_i], then the equality block A
struct ProbSOCPOpA : Operator {
    std::vector<MatOp> mats_g, vecs_c;
    MatOp mat_a;
    ProbSOCPOpA(std::vector<MatOp> g, std::vector<MatOp> c, MatOp a) : mats_g(std::move(g)), vecs_c(std::move(c)), mat_a(a) {}
    std::pair<size_t, size_t> size() const override
    {
        size_t s = 0;
        for (auto &g : mats_g) s += 1 + g.nr;
        return { s + mat_a.nr, mat_a.nc };
    }
    void op(float alpha, Slice x, float beta, Slice y) const override                   // socp.rs:77-101
    {
        size_t done = 0;
        for (size_t i = 0; i < mats_g.size(); ++i) {
            const size_t ni = mats_g[i].nr;
            vecs_c[i].trans_op(-alpha, x, beta, y.sub(done, 1));
            mats_g[i].op(-alpha, x, beta, y.sub(done + 1, ni));
            done += 1 + ni;
        }
        mat_a.op(alpha, x, beta, y.sub(done, mat_a.nr));
    }
    void trans_op(float alpha, Slice x, float beta, Slice y) const override             // socp.rs:103-130
    {
        F32HIP::scale(beta, y);
        size_t done = 0;
        for (size_t i = 0; i < mats_g.size(); ++i) {
            const size_t ni = mats_g[i].nr;
            vecs_c[i].op(-alpha, x.sub(done, 1), 1.f, y);
            mats_g[i].trans_op(-alpha, x.sub(done + 1, ni), 1.f, y);
            done += 1 + ni;
        }
        mat_a.trans_op(alpha, x.sub(done, mat_a.nr), 1.f, y);
    }
    void absadd_cols(Slice tau) const override                                          // socp.rs:132-141
    {
        for (auto &c : vecs_c) c.absadd_rows(tau);
        for (auto &g : mats_g) g.absadd_cols(tau);
        mat_a.absadd_cols(tau);
    }
    void absadd_rows(Slice sigma) const override                                        // socp.rs:143-162
    {
        size_t done = 0;
        for (size_t i = 0; i < mats_g.size(); ++i) {
            const size_t ni = mats_g[i].nr;
            vecs_c[i].absadd_cols(sigma.sub(done, 1));
            mats_g[i].absadd_rows(sigma.sub(done + 1, ni));
            done += 1 + ni;
        }
        mat_a.absadd_rows(sigma.sub(done, mat_a.nr));
    }
};

// ProbSOCPOpB (socp.rs:166-280): b = [d_i ; h_i] per cone, then the equality right-hand side
struct ProbSOCPOpB : Operator {
    std::vector<MatOp> vecs_h;
    std::vector<float> scls_d;
    float abssum_d;
    MatOp vec_b;
    ProbSOCPOpB(std::vector<MatOp> h, std::vector<float> d, MatOp b) : vecs_h(std::move(h)), scls_d(std::move(d)), vec_b(b)
    {
        abssum_d = 0.f;
        for (float v : scls_d) abssum_d += std::fabs(v);
    }
    std::pair<size_t, size_t> size() const override
    {
        size_t s = 0;
        for (auto &h : vecs_h) s += 1 + h.nr;
        return { s + vec_b.nr, 1 };
    }
    void op(float alpha, Slice x, float beta, Slice y) const override                   // socp.rs:194-217
    {
        size_t done = 0;
        for (size_t i = 0; i < vecs_h.size(); ++i) {
            const size_t ni = vecs_h[i].nr;
            Slice y1 = y.sub(done, 1);
            F32HIP::scale(beta, y1);
            F32HIP::add(alpha * scls_d[i], x, y1);
            vecs_h[i].op(alpha, x, beta, y.sub(done + 1, ni));
            done += 1 + ni;
        }
        vec_b.op(alpha, x, beta, y.sub(done, vec_b.nr));
    }
    void trans_op(float alpha, Slice x, float beta, Slice y) const override             // socp.rs:219-246
    {
        F32HIP::scale(beta, y);
        size_t done = 0;
        for (size_t i = 0; i < vecs_h.size(); ++i) {
            const size_t ni = vecs_h[i].nr;
            F32HIP::add(alpha * scls_d[i], x.sub(done, 1), y);
            vecs_h[i].trans_op(alpha, x.sub(done + 1, ni), 1.f, y);
            done += 1 + ni;
        }
        vec_b.trans_op(alpha, x.sub(done, vec_b.nr), 1.f, y);
    }
    void absadd_cols(Slice tau) const override                                          // socp.rs:248-257
    {
        tau.set(0, tau.get(0) + abssum_d);
        for (auto &h : vecs_h) h.absadd_cols(tau);
        vec_b.absadd_cols(tau);
    }
    void absadd_rows(Slice sigma) const override                                        // socp.rs:259-279 (adds d_i, not |d_i|)
    {
        size_t done = 0;
        for (size_t i = 0; i < vecs_h.size(); ++i) {
            const size_t ni = vecs_h[i].nr;
            Slice s1 = sigma.sub(done, 1);
            s1.set(0, s1.get(0) + scls_d[i]);
            vecs_h[i].absadd_rows(sigma.sub(done + 1, ni));
            done += 1 + ni;
        }
        vec_b.absadd_rows(sigma.sub(done, vec_b.nr));
    }
};

// ---- the reference's cones, LITERALLY: only SliceLike / LinAlg calls (what an unchanged totsu_core executes) -----------
// ConeRPos::proj is a host loop over get_mut() (cone_rpos.rs:38-45): with a device-resident slice that is a download,
// the loop, and an upload (f32cuda_slice.rs:343-355 does the same through its host / device dirty flags)
struct ConeRPosRef : Cone {
    std::vector<float> host;
    bool proj(bool, Slice x) override
    {
        if (x.n == 0) return true;
        host.resize(x.n);
        chk(thip_d2h(host.data(), x.p, x.n));
        for (float &e : host) e = e > 0.f ? e : 0.f;
        chk(thip_h2d(x.p, host.data(), x.n));
        return true;
    }
    void product_group(Slice) const override {}
};
struct ConeSOCRef : Cone {                                                              // cone_soc.rs:38-65
    bool proj(bool, Slice x) override
    {
        if (x.n == 0) return true;
        Slice s = x.sub(0, 1), v = x.sub(1, x.n - 1);
        const float val_s = s.get(0);
        const float norm_v = F32HIP::norm(v);
        if (norm_v <= -val_s) { F32HIP::scale(0.f, v); s.set(0, 0.f); }
        else if (norm_v <= val_s) { }
        else {
            const float alpha = (1.f + val_s / norm_v) / 2.f;
            F32HIP::scale(alpha, v);
            s.set(0, (norm_v + val_s) / 2.f);
        }
        return true;
    }
    void product_group(Slice t) const override { group_min(t); }
};
// ConePSD::proj with the closure evaluated on the HOST (cone_psd.rs:69-76 through linalg_ex.rs:64-65): two-phase map_eig
struct ConePSDRef : Cone {
    Slice work; float eps_zero;
    std::vector<float> w, e; std::vector<uint8_t> keep;
    ConePSDRef(Slice wk, float ez) : work(wk), eps_zero(ez) {}
    bool proj(bool, Slice x) override
    {
        if (work.n < ConePSD::query_worklen(x.n)) return false;
        const size_t n = (size_t)((std::sqrt((double)(8 * x.n + 1)) - 1.0) / 2.0 + 0.5);
        w.resize(n); e.resize(n); keep.resize(n);
        const float sq2 = std::sqrt(2.f);
        chk(thip_eig_decompose(n, x.p, 1, sq2, eps_zero, work.p, work.n, w.data()));
        for (size_t i = 0; i < n; ++i) { keep[i] = w[i] > 0.f; e[i] = w[i]; }           // |e| if e > 0 { Some(e) } else { None }
        chk(thip_eig_rebuild(n, x.p, 1, sq2, work.p, work.n, e.data(), keep.data()));
        return true;
    }
    void product_group(Slice t) const override { group_min(t); }
};

// ---- dense description for the device-resident loop ------------------------------------------------------------------------
struct DenseDesc {
    size_t n = 0, m = 0;
    // row blocks of the stacked A, top to bottom: (device matrix nr x n col-major, sign); a block with transposed = true
    // is an n x 1 column vector used as ONE row (the -c_i^T rows of socp.rs:88-93)
    struct Block { Slice mat; size_t nr; float sign; bool transposed; };
    std::vector<Block> blocks;
    std::vector<float> b, b_rowabs;               // host: right-hand side and what op_b.absadd_rows adds per row
    Slice c;                                      // device, n
    std::vector<int32_t> seg_type;
    std::vector<int64_t> seg_len;
};

struct Problem {
    std::unique_ptr<Operator> op_c, op_a, op_b;
    std::unique_ptr<Cone> cone;
    std::unique_ptr<DeviceVec> work;              // Solver::query_worklen floats; x = work[0..n), y = work[n..n+m) on Ok
    std::unique_ptr<DenseDesc> dense;             // non-null for the Hip* builders: solve() may take the fused loop
};

// which cones `problem()` instantiates: the device fast paths (thip_proj_*), or the reference's literal trait-level code
enum class ConeImpl { Device, Reference };

// ---- ProbLP (lp.rs:222-338) -------------------------------------------------------------------------------------------
class ProbLP {
public:
    ProbLP(const MatBuild &vec_c, const MatBuild &mat_g, const MatBuild &vec_h, const MatBuild &mat_a, const MatBuild &vec_b)
        : c_(vec_c), g_(mat_g), h_(vec_h), a_(mat_a), b_(vec_b)
    {
        const size_t n = c_.nr, m = h_.nr, p = b_.nr;
        if (c_.nc != 1 || g_.size() != std::make_pair(m, n) || h_.nc != 1 || a_.size() != std::make_pair(p, n) || b_.nc != 1)
            throw std::invalid_argument("ProbLP: inconsistent sizes");                    // lp.rs:283-296
    }
    virtual ~ProbLP() {}
    ConeImpl cones = ConeImpl::Device;
    MirrorCache cache;

    virtual Problem problem()                                                           // lp.rs:309-337
    {
        const size_t n = c_.nr, m = h_.nr, p = b_.nr;
        Problem pr;
        pr.op_c.reset(new OpVec(MatOp(n, 1, cache.get(c_))));
        pr.op_a.reset(new OpStack2(MatOp(m, n, cache.get(g_)), MatOp(p, n, cache.get(a_))));
        pr.op_b.reset(new OpStack2(MatOp(m, 1, cache.get(h_)), MatOp(p, 1, cache.get(b_))));
        auto *cone = new ConeProduct();
        if (cones == ConeImpl::Device) rp_.reset(new ConeRPos()); else rp_.reset(new ConeRPosRef());
        cone->blocks.push_back({ rp_.get(), m });
        cone->blocks.push_back({ &zero_, p });
        pr.cone.reset(cone);
        pr.work.reset(new DeviceVec(Solver::query_worklen(m + p, n)));
        return pr;
    }

protected:
    const MatBuild &c_, &g_, &h_, &a_, &b_;
    std::unique_ptr<Cone> rp_;
    ConeZero zero_;
};

// ---- ProbSOCP (socp.rs:336-474) --------------------------------------------------------------------------------------
class ProbSOCP {
public:
    ProbSOCP(const MatBuild &vec_f, const std::vector<MatBuild> &mats_g, const std::vector<MatBuild> &vecs_h,
             const std::vector<MatBuild> &vecs_c, const std::vector<float> &scls_d, const MatBuild &mat_a, const MatBuild &vec_b)
        : f_(vec_f), g_(mats_g), h_(vecs_h), cc_(vecs_c), d_(scls_d), a_(mat_a), b_(vec_b)
    {
        const size_t n = f_.nr, mc = g_.size(), p = b_.nr;
        if (h_.size() != mc || cc_.size() != mc || d_.size() != mc || f_.nc != 1) throw std::invalid_argument("ProbSOCP: sizes");
        for (size_t i = 0; i < mc; ++i)
            if (g_[i].nc != n || h_[i].size() != std::make_pair(g_[i].nr, (size_t)1) || cc_[i].size() != std::make_pair(n, (size_t)1))
                throw std::invalid_argument("ProbSOCP: block sizes");                     // socp.rs:405-417
        if (a_.size() != std::make_pair(p, n) || b_.nc != 1) throw std::invalid_argument("ProbSOCP: equality block");
    }
    virtual ~ProbSOCP() {}
    ConeImpl cones = ConeImpl::Device;
    MirrorCache cache;

    virtual Problem problem()                                                           // socp.rs:430-473
    {
        const size_t n = f_.nr, p = b_.nr;
        Problem pr;
        std::vector<MatOp> og, oc, oh;
        size_t m = 0;
        for (size_t i = 0; i < g_.size(); ++i) {
            og.push_back(MatOp(g_[i].nr, n, cache.get(g_[i])));
            oc.push_back(MatOp(n, 1, cache.get(cc_[i])));
            oh.push_back(MatOp(h_[i].nr, 1, cache.get(h_[i])));
            m += 1 + g_[i].nr;
        }
        pr.op_c.reset(new OpVec(MatOp(n, 1, cache.get(f_))));
        pr.op_a.reset(new ProbSOCPOpA(og, oc, MatOp(p, n, cache.get(a_))));
        pr.op_b.reset(new ProbSOCPOpB(oh, d_, MatOp(p, 1, cache.get(b_))));
        auto *cone = new ConeProduct();                                                 // ProbSOCPCone, socp.rs:284-332
        if (cones == ConeImpl::Device) soc_.reset(new ConeSOC()); else soc_.reset(new ConeSOCRef());
        for (size_t i = 0; i < g_.size(); ++i) cone->blocks.push_back({ soc_.get(), 1 + g_[i].nr });
        cone->blocks.push_back({ &zero_, p });
        pr.cone.reset(cone);
        pr.work.reset(new DeviceVec(Solver::query_worklen(m + p, n)));
        return pr;
    }

protected:
    const MatBuild &f_;
    const std::vector<MatBuild> &g_, &h_, &cc_;
    std::vector<float> d_;
    const MatBuild &a_, &b_;
    std::unique_ptr<Cone> soc_;
    ConeZero zero_;
};

// ---- ProbSDP (sdp.rs:222-332) ----------------------------------------------------------------------------------------
// MatType::SymPack(k): the upper triangle by columns (matbuild/mod.rs:254-279)
struct SymPack {
    size_t k = 0;
    std::vector<float> array;
    SymPack() {}
    explicit SymPack(size_t k_) : k(k_), array(k_ * (k_ + 1) / 2, 0.f) {}
    void set(size_t r, size_t c, float v) { if (r > c) std::swap(r, c); array[c * (c + 1) / 2 + r] = v; }
    SymPack &iter_rowmaj(const std::vector<float> &v)             // set_iter_rowmaj on the full k x k matrix: later entries win
    {
        size_t i = 0;
        for (size_t r = 0; r < k; ++r) for (size_t c = 0; c < k && i < v.size(); ++c) set(r, c, v[i++]);
        return *this;
    }
    // scale_nondiag(sqrt 2) + reshape_colvec (sdp.rs:271-274): svec
    std::vector<float> svec() const
    {
        std::vector<float> o(array);
        const float s2 = std::sqrt(2.f);
        for (size_t c = 0; c < k; ++c) for (size_t r = 0; r < c; ++r) o[c * (c + 1) / 2 + r] *= s2;
        return o;
    }
};

class ProbSDP {
public:
    ProbSDP(const MatBuild &vec_c, const std::vector<SymPack> &syms_f, const MatBuild &mat_a, const MatBuild &vec_b, float eps_zero)
        : c_(vec_c), a_(mat_a), b_(vec_b), eps_zero_(eps_zero)
    {
        const size_t n = c_.nr, p = b_.nr;
        if (c_.nc != 1 || syms_f.size() != n + 1 || a_.size() != std::make_pair(p, n) || b_.nc != 1) throw std::invalid_argument("ProbSDP: sizes");
        const size_t k = syms_f[0].k;
        for (auto &f : syms_f) if (f.k != k) throw std::invalid_argument("ProbSDP: orders differ");
        sk_ = k * (k + 1) / 2;
        symmat_f_ = MatBuild(sk_, n);                                                     // sdp.rs:276-281
        for (size_t c = 0; c < n; ++c) { const std::vector<float> v = syms_f[c].svec(); std::copy(v.begin(), v.end(), symmat_f_.array.begin() + c * sk_); }
        symvec_f_n_ = MatBuild(sk_, 1);
        symvec_f_n_.array = syms_f[n].svec();
    }
    virtual ~ProbSDP() {}
    ConeImpl cones = ConeImpl::Device;
    MirrorCache cache;

    virtual Problem problem()                                                           // sdp.rs:299-331
    {
        const size_t n = c_.nr, p = b_.nr;
        Problem pr;
        pr.op_c.reset(new OpVec(MatOp(n, 1, cache.get(c_))));
        pr.op_a.reset(new OpStack2(MatOp(sk_, n, cache.get(symmat_f_)), MatOp(p, n, cache.get(a_))));
        pr.op_b.reset(new OpStack2(MatOp(sk_, 1, cache.get(symvec_f_n_)), MatOp(p, 1, cache.get(b_)), -1.f));
        w_cone_.reset(new DeviceVec(ConePSD::query_worklen(sk_)));
        if (cones == ConeImpl::Device) psd_.reset(new ConePSD(w_cone_->slice(), eps_zero_));
        else psd_.reset(new ConePSDRef(w_cone_->slice(), eps_zero_));
        auto *cone = new ConeProduct();
        cone->blocks.push_back({ psd_.get(), sk_ });
        cone->blocks.push_back({ &zero_, p });
        pr.cone.reset(cone);
        pr.work.reset(new DeviceVec(Solver::query_worklen(sk_ + p, n)));
        return pr;
    }

protected:
    const MatBuild &c_, &a_, &b_;
    MatBuild symmat_f_, symvec_f_n_;
    size_t sk_ = 0;
    float eps_zero_;
    std::unique_ptr<DeviceVec> w_cone_;
    std::unique_ptr<Cone> psd_;
    ConeZero zero_;
};

// ---- Hip* aliases: the same builders + a dense description of themselves -------------------------------------------------
class HipProbLP : public ProbLP {
public:
    using ProbLP::ProbLP;
    Problem problem() override
    {
        Problem pr = ProbLP::problem();
        const size_t n = c_.nr, m = h_.nr, p = b_.nr;
        std::unique_ptr<DenseDesc> d(new DenseDesc());
        d->n = n; d->m = m + p;
        d->blocks.push_back({ cache.get(g_), m, 1.f, false });
        d->blocks.push_back({ cache.get(a_), p, 1.f, false });
        d->b.insert(d->b.end(), h_.array.begin(), h_.array.end());
        d->b.insert(d->b.end(), b_.array.begin(), b_.array.end());
        for (float v : d->b) d->b_rowabs.push_back(std::fabs(v));
        d->c = cache.get(c_);
        d->seg_type = { THIP_CONE_RPOS, THIP_CONE_ZERO };
        d->seg_len = { (int64_t)m, (int64_t)p };
        pr.dense = std::move(d);
        return pr;
    }
};

class HipProbSOCP : public ProbSOCP {
public:
    using ProbSOCP::ProbSOCP;
    Problem problem() override
    {
        Problem pr = ProbSOCP::problem();
        const size_t n = f_.nr, p = b_.nr;
        std::unique_ptr<DenseDesc> d(new DenseDesc());
        d->n = n;
        for (size_t i = 0; i < g_.size(); ++i) {                                        // rows [-c_i^T ; -G_i], b = [d_i ; h_i]
            d->blocks.push_back({ cache.get(cc_[i]), 1, -1.f, true });
            d->blocks.push_back({ cache.get(g_[i]), g_[i].nr, -1.f, false });
            d->b.push_back(d_[i]); d->b_rowabs.push_back(d_[i]);                          // socp.rs:259-279 adds d_i, not |d_i|
            for (float v : h_[i].array) { d->b.push_back(v); d->b_rowabs.push_back(std::fabs(v)); }
            d->seg_type.push_back(THIP_CONE_SOC); d->seg_len.push_back((int64_t)(1 + g_[i].nr));
            d->m += 1 + g_[i].nr;
        }
        d->blocks.push_back({ cache.get(a_), p, 1.f, false });
        for (float v : b_.array) { d->b.push_back(v); d->b_rowabs.push_back(std::fabs(v)); }
        d->seg_type.push_back(THIP_CONE_ZERO); d->seg_len.push_back((int64_t)p);
        d->m += p;
        d->c = cache.get(f_);
        pr.dense = std::move(d);
        return pr;
    }
};

class HipProbSDP : public ProbSDP {
public:
    using ProbSDP::ProbSDP;
    Problem problem() override
    {
        Problem pr = ProbSDP::problem();
        const size_t n = c_.nr, p = b_.nr;
        std::unique_ptr<DenseDesc> d(new DenseDesc());
        d->n = n; d->m = sk_ + p;
        d->blocks.push_back({ cache.get(symmat_f_), sk_, 1.f, false });
        d->blocks.push_back({ cache.get(a_), p, 1.f, false });
        for (float v : symvec_f_n_.array) { d->b.push_back(-v); d->b_rowabs.push_back(std::fabs(v)); }   // b = [-svec(F_n) ; b], sdp.rs:152
        for (float v : b_.array) { d->b.push_back(v); d->b_rowabs.push_back(std::fabs(v)); }
        d->c = cache.get(c_);
        d->seg_type = { THIP_CONE_PSD, THIP_CONE_ZERO };
        d->seg_len = { (int64_t)sk_, (int64_t)p };
        pr.dense = std::move(d);
        return pr;
    }
};

// ---- Solver::solve on a Problem: the fused loop when the problem describes itself densely, else call by call --------------
struct SolveInfo { bool fused = false; int64_t iters = -1; double stack_seconds = 0.0; };

inline SolverError solve(Solver &s, Problem &pr, SolveInfo *info = nullptr, bool allow_fused = true, int64_t max_steps = -1)
{
    const size_t m = pr.op_a->size().first, n = pr.op_a->size().second;
    if (allow_fused && pr.dense) {
        const DenseDesc &d = *pr.dense;
        // one stacked matrix: a single block that covers every row is used in place; otherwise the blocks are copied
        // into a library-side m x n array (strided device copies; a transposed block is one row)
        std::unique_ptr<DeviceVec> stacked;
        Slice A{ nullptr, 0 };
        size_t nonempty = 0;
        for (auto &b : d.blocks) nonempty += b.nr > 0;
        if (nonempty == 1 && d.blocks[0].nr == d.m && d.blocks[0].sign == 1.f && !d.blocks[0].transposed) A = d.blocks[0].mat;
        else if (d.m && d.n) {
            stacked.reset(new DeviceVec(d.m * d.n));
            A = stacked->slice();
            size_t r0 = 0;
            for (auto &b : d.blocks) {
                if (b.nr == 0) continue;
                chk(thip_copy_block(b.transposed ? 1 : 0, b.nr, d.n, b.sign, b.mat.p, A.p + r0, d.m));
                r0 += b.nr;
            }
        }
        DeviceVec db(d.b), dabs(d.b_rowabs);
        thip_problem prob{};
        prob.n = d.n; prob.m = d.m; prob.mat_a = A.p; prob.vec_b = db.slice().p; prob.vec_c = d.c.p;
        prob.vec_b_rowabs = dabs.slice().p;
        prob.n_seg = d.seg_type.size(); prob.host_seg_type = d.seg_type.data(); prob.host_seg_len = d.seg_len.data();
        thip_param p{};
        p.max_iter = s.par.max_iter; p.eps_acc = s.par.eps_acc; p.eps_inf = s.par.eps_inf; p.eps_zero = s.par.eps_zero;
        p.state_arith = s.par.state_arith;
        thip_solver *h = nullptr;
        chk(thip_solver_create(&prob, &p, THIP_SCHED_SWEEP, &h));      // falls back to the carried schedule by itself
        thip_status st{};
        int rc = thip_solver_init(h);
        if (rc == 0) rc = thip_solver_run(h, max_steps, 64, &st);
        if (rc == 0) {
            // solver.rs:317-320: the answers are work[0..n) and work[n..n+m)
            std::vector<float> x(n), y(m);
            rc = thip_solver_solution(h, x.data(), y.data());
            if (rc == 0 && n) rc = thip_h2d(pr.work->slice().p, x.data(), n);
            if (rc == 0 && m) rc = thip_h2d(pr.work->slice().p + n, y.data(), m);
        }
        thip_solver_destroy(h);
        chk(rc);
        s.iters = st.iter;
        if (info) { info->fused = true; info->iters = st.iter; }
        return st.state <= 0 ? SolverError::Ok : (SolverError)st.state;
    }
    const SolverError e = s.solve(*pr.op_c, *pr.op_a, *pr.op_b, *pr.cone, pr.work->slice());
    if (info) { info->fused = false; info->iters = s.iters; }
    return e;
}

}  // namespace totsu
