// cpp_trait_demo.cpp -- a compiled host over include/totsu_f32hip.hpp (which is layered only on the C ABI):
// the reference's trait-level calling sequence `Solver::solve((op_c, op_a, op_b, cone, work))` in C++, on three
// known-answer problems: the nostd_cortex-m LP (x = [2, 2]), a second-order-cone problem (x = [-1, -1], the data of
// totsu/tests/socp.rs test_socp1 in stacked form) and the 2x2 PSD problem of totsu_core/tests/solver.rs (x = -2); then
// the LP with op_a as a sparse user Operator (SparseOp), and once more through FusedSolver (the device-resident loop) with f16
// storage of A and an f32 finish.
#include <cstdio>
#include "totsu_f32hip.hpp"

using namespace totsu;

static int run(const char *name, size_t n, size_t m, const std::vector<float> &c, const std::vector<float> &a,
               const std::vector<float> &b, Cone &cone, const std::vector<float> &want, float tol)
{
    DeviceVec dc(c), da(a), db(b);
    MatOp op_c(n, 1, dc.slice()), op_a(m, n, da.slice()), op_b(m, 1, db.slice());
    DeviceVec work(Solver::query_worklen(m, n));
    Solver s;
    s.par.max_iter = 100000;
    s.par.eps_acc = 1e-5f;
    const SolverError e = s.solve(op_c, op_a, op_b, cone, work.slice());
    const std::vector<float> w = work.to_host();
    bool ok = e == SolverError::Ok;
    printf("%-10s status %d after %lld iterations: x = [", name, (int)e, (long long)s.iters);
    for (size_t i = 0; i < n; ++i) { printf("%s%.5f", i ? ", " : "", w[i]); ok = ok && std::fabs(w[i] - want[i]) <= tol; }
    printf("]  %s\n", ok ? "OK" : "MISMATCH");
    return ok ? 0 : 1;
}

int main()
{
    int ndev = 0;
    thip_device_count(&ndev);
    if (ndev == 0) { fprintf(stderr, "no GPU: no CPU fallback\n"); return 3; }
    chk(thip_init(0));
    int bad = 0;
    {
        ConeRPos cone;
        bad += run("lp", 2, 3, { -1, 0 }, { 4, -1, -1, -1, 4, -1 }, { 6, 6, 1 }, cone, { 2, 2 }, 1e-3f);
    }
    {   // min x0 + x1  s.t.  ||x|| <= sqrt 2 : rows [-c^T ; -G] = [0 0 ; -1 0 ; 0 -1], b = [sqrt 2 ; 0 ; 0]
        ConeSOC cone;
        bad += run("socp", 2, 3, { 1, 1 }, { 0, -1, 0, 0, 0, -1 }, { 1.41421356f, 0, 0 }, cone, { -1, -1 }, 1e-3f);
    }
    {
        DeviceVec w(ConePSD::query_worklen(3));
        ConePSD cone(w.slice(), 1e-12f);
        bad += run("psd", 1, 3, { 1 }, { 0, -1.41421356f, -3 }, { 1, 0, 10 }, cone, { -2 }, 1e-3f);
    }
    {   // the LP with op_a as a caller's sparse Operator (SparseOp over thip_sptile_*: the matrix by columns, one copy on the device)
        ConeRPos cone;
        DeviceVec dc(std::vector<float>{ -1, 0 }), db(std::vector<float>{ 6, 6, 1 });
        MatOp op_c(2, 1, dc.slice()), op_b(3, 1, db.slice());
        SparseOp op_a(3, 2, { 0, 3, 6 }, { 0, 1, 2, 0, 1, 2 }, { 4, -1, -1, -1, 4, -1 });
        DeviceVec work(Solver::query_worklen(3, 2));
        Solver s;
        s.par.max_iter = 100000;
        s.par.eps_acc = 1e-5f;
        const SolverError e = s.solve(op_c, op_a, op_b, cone, work.slice());
        const std::vector<float> w = work.to_host();
        const bool ok = e == SolverError::Ok && std::fabs(w[0] - 2.f) <= 1e-3f && std::fabs(w[1] - 2.f) <= 1e-3f;
        printf("%-10s status %d after %lld iterations: x = [%.5f, %.5f]  %s\n", "sparse-lp", (int)e, (long long)s.iters, w[0], w[1], ok ? "OK" : "MISMATCH");
        bad += ok ? 0 : 1;
    }
    {   // the same LP through the device-resident loop, with the 16-bit storage of A and a resumed finish on f32:
        // f16 passes to eps 1e-4, then thip_solver_set_a_storage(F32) + resume with eps 1e-5 from the same iterate
        DeviceVec dc(std::vector<float>{ -1, 0 }), da(std::vector<float>{ 4, -1, -1, -1, 4, -1 }), db(std::vector<float>{ 6, 6, 1 });
        SolverParam par;
        par.max_iter = 100000; par.eps_acc = 1e-4f;
        FusedSolver fs(2, 3, da.slice(), db.slice(), dc.slice(), { THIP_CONE_RPOS }, { 3 }, par);
        fs.set_a_storage(THIP_A_F16);
        SolverError e = fs.run();
        const long long it16 = (long long)fs.iters();
        par.eps_acc = 1e-5f;
        fs.set_a_storage(THIP_A_F32);
        fs.resume(&par);
        e = fs.run();
        std::vector<float> x, y;
        fs.solution(x, y);
        const bool ok = e == SolverError::Ok && std::fabs(x[0] - 2.f) <= 1e-3f && std::fabs(x[1] - 2.f) <= 1e-3f && fs.iters() >= it16;
        printf("%-10s status %d after %lld (+%lld on f32) iterations: x = [%.5f, %.5f]  %s\n", "fused-lp", (int)e, it16,
               (long long)fs.iters() - it16, x[0], x[1], ok ? "OK" : "MISMATCH");
        bad += ok ? 0 : 1;
    }
    chk(thip_shutdown());
    return bad;
}
