// hip_prob_demo.cpp -- the drop-in by ALIAS: a caller written against totsu's builders (`ProbLP::new(vec_c, mat_g, vec_h,
// mat_a, vec_b)` -> `problem()` -> `Solver::solve`, totsu/src/problem/lp.rs:222-338) switches the type name to HipProbLP /
// HipProbSOCP (include/totsu_f32hip_prob.hpp) and `solve` hands the whole problem to the device-resident loop.
//   1. known-answer problems through BOTH routes -- the reference's composite operators call by call with the
//      reference's literal cones, and the alias route -- must agree (totsu/tests/socp.rs test_socp1: x = [-1, -1];
//      the nostd_cortex-m LP: x = [2, 2]);
//   2. the benchmark_lp construction (experimental/benchmark_lp/src/main.rs:14-57) at size n (default 10 000 = BASELINE
//      configs[1]) built on the HOST as MatBuild arrays, uploaded through the mirror cache, solved through HipProbLP:
//      iterations/sec next to a FusedSolver driven directly on the same device data.  Prints one JSON line.
// usage: hip_prob_demo [n] [iters]
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "totsu_f32hip_prob.hpp"

using namespace totsu;

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// xorshift64*: any uniform generator will do for the construction
static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static float uni()
{
    g_state ^= g_state >> 12; g_state ^= g_state << 25; g_state ^= g_state >> 27;
    return (float)((g_state * 0x2545F4914F6CDD1Dull) >> 40) * (1.0f / 16777216.0f);
}

static int kats()
{
    int bad = 0;
    {   // nostd_cortex-m LP (examples/nostd_cortex-m/src/main.rs): x = [2, 2]
        MatBuild c(2, 1), g(3, 2), h(3, 1), a(0, 2), b(0, 1);
        c.iter_colmaj({ -1, 0 }); g.iter_colmaj({ 4, -1, -1, -1, 4, -1 }); h.iter_colmaj({ 6, 6, 1 });
        for (int route = 0; route < 2; ++route) {
            HipProbLP lp(c, g, h, a, b);
            lp.cones = route == 0 ? ConeImpl::Reference : ConeImpl::Device;
            Problem pr = lp.problem();
            Solver s;
            s.par.max_iter = 100000; s.par.eps_acc = 1e-5f;
            SolveInfo info;
            const SolverError e = solve(s, pr, &info, route == 1);
            const std::vector<float> w = pr.work->to_host();
            const bool ok = e == SolverError::Ok && std::fabs(w[0] - 2.f) <= 1e-3f && std::fabs(w[1] - 2.f) <= 1e-3f && info.fused == (route == 1);
            printf("kat lp   %-28s status %d iters %lld x = [%.5f, %.5f] uploads %zu  %s\n",
                   route ? "HipProbLP -> fused loop" : "composite ops, reference cones", (int)e, (long long)info.iters, w[0], w[1],
                   lp.cache.uploads, ok ? "OK" : "MISMATCH");
            bad += !ok;
        }
    }
    {   // totsu/tests/socp.rs test_socp1: min x0 + x1 s.t. ||x|| <= sqrt 2 -> x = [-1, -1]
        MatBuild f(2, 1), a(0, 2), b(0, 1);
        f.iter_colmaj({ 1, 1 });
        std::vector<MatBuild> gs(1, MatBuild(2, 2)), hs(1, MatBuild(2, 1)), cs(1, MatBuild(2, 1));
        gs[0].iter_colmaj({ 1, 0, 0, 1 });
        std::vector<float> d{ 1.41421356f };
        for (int route = 0; route < 2; ++route) {
            HipProbSOCP socp(f, gs, hs, cs, d, a, b);
            socp.cones = route == 0 ? ConeImpl::Reference : ConeImpl::Device;
            Problem pr = socp.problem();
            Solver s;
            s.par.max_iter = 100000; s.par.eps_acc = 1e-5f;
            SolveInfo info;
            const SolverError e = solve(s, pr, &info, route == 1);
            const std::vector<float> w = pr.work->to_host();
            const bool ok = e == SolverError::Ok && std::fabs(w[0] + 1.f) <= 1e-3f && std::fabs(w[1] + 1.f) <= 1e-3f && info.fused == (route == 1);
            // as_op() is taken twice per G_i (socp.rs:450,463): the mirror cache uploads each host array once
            printf("kat socp %-28s status %d iters %lld x = [%.5f, %.5f] uploads %zu  %s\n",
                   route ? "HipProbSOCP -> fused loop" : "composite ops, reference cones", (int)e, (long long)info.iters, w[0], w[1],
                   socp.cache.uploads, ok ? "OK" : "MISMATCH");
            bad += !ok;
        }
    }
    {   // totsu/tests/sdp.rs test_sdp1: min x0 + x1 s.t. x0 F0 + x1 F1 + F2 <= 0 -> x = [3, 4]
        MatBuild c(2, 1), a(0, 2), b(0, 1);
        c.iter_colmaj({ 1, 1 });
        std::vector<SymPack> syms(3, SymPack(2));
        syms[0].iter_rowmaj({ -1, 0, 0, 0 });
        syms[1].iter_rowmaj({ 0, 0, 0, -1 });
        syms[2].iter_rowmaj({ 3, 0, 0, 4 });
        for (int route = 0; route < 2; ++route) {
            HipProbSDP sdp(c, syms, a, b, 1e-12f);
            sdp.cones = route == 0 ? ConeImpl::Reference : ConeImpl::Device;
            Problem pr = sdp.problem();
            Solver s;
            s.par.max_iter = 100000; s.par.eps_acc = 1e-5f;
            SolveInfo info;
            const SolverError e = solve(s, pr, &info, route == 1);
            const std::vector<float> w = pr.work->to_host();
            const bool ok = e == SolverError::Ok && std::fabs(w[0] - 3.f) <= 1e-3f && std::fabs(w[1] - 4.f) <= 1e-3f && info.fused == (route == 1);
            printf("kat sdp  %-28s status %d iters %lld x = [%.5f, %.5f] uploads %zu  %s\n",
                   route ? "HipProbSDP -> fused loop" : "composite ops, reference cones", (int)e, (long long)info.iters, w[0], w[1],
                   sdp.cache.uploads, ok ? "OK" : "MISMATCH");
            bad += !ok;
        }
    }
    return bad;
}

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 10000;
    const int64_t iters = argc > 2 ? atol(argv[2]) : 400;
    int ndev = 0;
    thip_device_count(&ndev);
    if (ndev == 0) { fprintf(stderr, "no GPU: no CPU fallback\n"); return 3; }
    chk(thip_init(0));
    int bad = kats();

    // benchmark_lp: c = -U(0,1), G = [-I ; U(0,1)], h = [0 ; U(0,1)], no equalities
    const size_t m = 2 * n;
    MatBuild c(n, 1), g(m, n), h(m, 1), a(0, n), b(0, 1);
    for (size_t i = 0; i < n; ++i) c.array[i] = -uni();
    for (size_t col = 0; col < n; ++col) {
        g.at(col, col) = -1.f;
        for (size_t r = n; r < m; ++r) g.at(r, col) = uni();
    }
    for (size_t r = n; r < m; ++r) h.array[r] = uni();

    HipProbLP lp(c, g, h, a, b);
    const double tu0 = now();
    Problem pr = lp.problem();
    chk(thip_sync());
    const double upload_s = now() - tu0;
    Solver s;
    s.par.eps_acc = 0.f; s.par.eps_inf = 0.f;
    SolveInfo info;
    auto run_alias = [&](int64_t k) {
        s.par.max_iter = k;
        chk(thip_sync());
        const double t0 = now();
        const SolverError e = solve(s, pr, &info);
        chk(thip_sync());
        if (e != SolverError::ExcessIter || !info.fused) bad += 1;
        return now() - t0;
    };
    const int64_t k1 = iters / 4 > 0 ? iters / 4 : 1;
    run_alias(k1);                                      // first call: library warm-up
    const double t1 = run_alias(k1), t2 = run_alias(k1 + iters);
    const double alias_rate = (double)iters / (t2 - t1);

    // the fused loop driven directly on the same device data
    double fused_rate = 0.0;
    {
        DeviceVec db(h.array);
        SolverParam par;
        par.eps_acc = 0.f; par.eps_inf = 0.f;
        FusedSolver fs(n, m, lp.cache.get(g), db.slice(), lp.cache.get(c), { THIP_CONE_RPOS }, { (int64_t)m }, par);
        fs.run(k1, k1);
        chk(thip_sync());
        const double t0 = now();
        fs.run(iters, iters);
        chk(thip_sync());
        fused_rate = (double)iters / (now() - t0);
    }
    printf("{\"what\": \"benchmark_lp n=%zu m=%zu through HipProbLP (alias of ProbLP, lp.rs:222-338) vs FusedSolver driven directly\", "
           "\"alias_iter_per_s\": %.1f, \"fused_iter_per_s\": %.1f, \"ratio\": %.3f, \"upload_s\": %.3f, \"uploads\": %zu, "
           "\"alias_solve_overhead_s\": %.4f}\n",
           n, m, alias_rate, fused_rate, alias_rate / fused_rate, upload_s, lp.cache.uploads, t1 - (double)k1 / alias_rate);
    if (alias_rate < 0.9 * fused_rate) bad += 1;
    chk(thip_shutdown());
    return bad;
}
