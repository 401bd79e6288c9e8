"""GPU parity tests, solver level: the reference's known-answer tests through the trait-level `Solver(F32HIP)`
and through the fused device loop, iterate-level agreement with the CPU oracle on seeded LP / SOCP / SDP
instances, agreement of the three schedules, and size-independent properties at BASELINE.json's LP size."""
import numpy as np
import pytest

import oracle as O
from problems import benchmark_lp, random_sdp, random_socp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import totsu_amd
    from totsu_amd import _lib
    _lib.init()
    return totsu_amd


def _mb(T, typ):
    return T.MatBuild(T.F32HIP, typ)


def _par(s, **kw):
    """trait-level Solver: the fused drop-in dispatch is switched off so that every L call is exercised"""
    s.fused = None
    for k, v in kw.items():
        setattr(s.param, k, v)
    return s


# ---- known-answer tests (f32: eps_acc 1e-4 like-for-like with the reference's own f32 tolerance 1e-3,
# experimental/benchmark_lp/src/main.rs:62-65; answers asserted to abs 1e-3 like totsu/tests/*.rs) -----------

def test_kat_nostd_lp_trait_level(T):
    # examples/nostd_cortex-m/src/main.rs:57-99 on F32HIP
    L = T.F32HIP
    op_c = T.MatOp(L, T.MatType.General(2, 1), np.array([-1., 0.], np.float32))
    op_a = T.MatOp(L, T.MatType.General(3, 2), np.array([4., -1., -1., -1., 4., -1.], np.float32))
    op_b = T.MatOp(L, T.MatType.General(3, 1), np.array([6., 6., 1.], np.float32))
    s = _par(T.Solver(L), max_iter=100_000, eps_acc=1e-5)
    s.trace = []
    work = np.zeros(48, dtype=np.float32)
    x, y = s.solve((op_c, op_a, op_b, T.ConeRPos(L), work))
    assert np.allclose(x, [2.0, 2.0], atol=1e-3)
    # the f64 golden trace converges at iteration 159 with eps 1e-6; f32 at 1e-5 must be in the same regime
    assert 100 < s.trace[-1][0] < 200
    ro = O.solve_matop_cones(O.param(max_iter=100000, eps_acc=1e-5), [-1., 0.], [4., -1., -1., -1., 4., -1.],
                             [6., 6., 1.], [O.CONE_RPOS], [3], trace_cap=400)
    assert abs(s.trace[-1][0] - ro.iters) <= 2
    for a, b in zip(s.trace[:100], ro.trace[:100]):
        assert np.allclose(a[2:], b[2:], rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize("schedule", ["reference", "fused", "carried"])
def test_kat_nostd_lp_fused(T, schedule):
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 100_000, 1e-5
    fs = T.FusedSolver(2, 3, [4., -1., -1., -1., 4., -1.], [6., 6., 1.], [-1., 0.], [1], [3], p, schedule)
    x, y = fs.solve()
    r = fs.status()
    assert np.allclose(x, [2.0, 2.0], atol=1e-3)
    ro = O.solve_matop_cones(O.param(max_iter=100000, eps_acc=1e-5), [-1., 0.], [4., -1., -1., -1., 4., -1.],
                             [6., 6., 1.], [O.CONE_RPOS], [3])
    assert abs(r.iters - ro.iters) <= 2
    assert np.allclose(y, ro.y, atol=1e-3)
    fs.destroy()


def _lp(T, c, g_rowmaj, h, n, m):
    return T.ProbLP(_mb(T, T.MatType.General(n, 1)).iter_colmaj(c), _mb(T, T.MatType.General(m, n)).iter_rowmaj(g_rowmaj),
                    _mb(T, T.MatType.General(m, 1)).iter_colmaj(h), _mb(T, T.MatType.General(0, n)),
                    _mb(T, T.MatType.General(0, 1)))


@pytest.mark.parametrize("path", ["trait", "reference", "fused", "carried"])
def test_kat_lp_infeasible_unbounded(T, path):
    # totsu/tests/lp.rs:13-82
    for g, h, want in (([1., -1.], [-5., -10.], T.SolverError.Infeasible), ([1., 1.], [5., 10.], T.SolverError.Unbounded)):
        lp = _lp(T, [1.], g, h, 1, 2)
        with pytest.raises(T.SolverError) as e:
            if path == "trait":
                _par(T.Solver(T.F32HIP), max_iter=100_000).solve(lp.problem())
            else:
                p = T.SolverParam()
                p.max_iter = 100_000
                T.FusedSolver.from_dense(lp.dense(), p, path).solve()
        assert e.value.kind == want
        lp.drop()


def _socp_kats(T):
    n = 2
    vec_f = _mb(T, T.MatType.General(n, 1)).by_fn(lambda r, c: 1.0)
    g = _mb(T, T.MatType.General(2, n))
    g[(0, 0)] = 1.0
    g[(1, 1)] = 1.0
    s1 = T.ProbSOCP(vec_f, [g], [_mb(T, T.MatType.General(2, 1))], [_mb(T, T.MatType.General(n, 1))], [np.sqrt(2.0)],
                    _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)))
    vec_f = _mb(T, T.MatType.General(n, 1)).iter_colmaj([0., 1.])
    mats_g = [_mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(1, n)).iter_rowmaj([-1., 0.])]
    vecs_h = [_mb(T, T.MatType.General(0, 1)), _mb(T, T.MatType.General(1, 1)).iter_colmaj([2.])]
    vecs_c = [_mb(T, T.MatType.General(2, 1)).iter_colmaj([0., -1.]), _mb(T, T.MatType.General(2, 1)).iter_colmaj([0., 1.])]
    s2 = T.ProbSOCP(vec_f, mats_g, vecs_h, vecs_c, [50., 0.], _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)))
    return (s1, [-1., -1.]), (s2, [2., 0.])


@pytest.mark.parametrize("path", ["trait", "reference", "fused", "carried"])
def test_kat_socp(T, path):
    # totsu/tests/socp.rs:14-93 (second one has a zero-row cone)
    for socp, want in _socp_kats(T):
        if path == "trait":
            x, _ = _par(T.Solver(T.F32HIP), max_iter=100_000, eps_acc=1e-5).solve(socp.problem())
        else:
            p = T.SolverParam()
            p.max_iter, p.eps_acc = 100_000, 1e-5
            x, _ = T.FusedSolver.from_dense(socp.dense(), p, path).solve()
        assert np.allclose(x, want, atol=1e-3), (path, x)
        socp.drop()


@pytest.mark.parametrize("path", ["trait", "fused", "carried"])
def test_kat_psd(T, path):
    # totsu_core/tests/solver.rs:14-53 == totsu_f32cuda/tests/solver.rs:16-54, and totsu/tests/sdp.rs:14-50
    L = T.F32HIP
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 100_000, 1e-5
    if path == "trait":
        op_c = T.MatOp(L, T.MatType.General(1, 1), np.array([1.], np.float32))
        op_a = T.MatOp(L, T.MatType.General(3, 1), np.array([0., -1. * 1.41421356, -3.], np.float32))
        op_b = T.MatOp(L, T.MatType.General(3, 1), np.array([1., 0., 10.], np.float32))
        s = _par(T.Solver(L), max_iter=100_000, eps_acc=1e-5)
        cone_w = np.zeros(T.ConePSD.query_worklen(L, 3), dtype=np.float32)
        cone = T.ConePSD(L, cone_w, s.param.eps_zero)
        work = np.zeros(T.Solver.query_worklen(op_a.size()), dtype=np.float32)
        x, _ = s.solve((op_c, op_a, op_b, cone, work))
    else:
        x, _ = T.FusedSolver(1, 3, [0., -1. * 1.41421356, -3.], [1., 0., 10.], [1.], [4], [3], p, path).solve()
    assert abs(x[0] + 2.0) <= 1e-3

    n, k = 2, 2
    vec_c = _mb(T, T.MatType.General(n, 1)).iter_colmaj([1., 1.])
    syms = [_mb(T, T.MatType.SymPack(k)) for _ in range(n + 1)]
    syms[0].set_iter_rowmaj([-1., 0., 0., 0.])
    syms[1].set_iter_rowmaj([0., 0., 0., -1.])
    syms[2].set_iter_rowmaj([3., 0., 0., 4.])
    sdp = T.ProbSDP(vec_c, syms, _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)), 1e-12)
    if path == "trait":
        x, _ = _par(T.Solver(L), max_iter=100_000, eps_acc=1e-5).solve(sdp.problem())
    else:
        x, _ = T.FusedSolver.from_dense(sdp.dense(), p, path).solve()
    assert np.allclose(x, [3., 4.], atol=1e-3)
    sdp.drop()


# ---- iterate-level parity with the oracle on seeded instances ------------------------------------------

def _oracle_snaps(dense, iters):
    par = O.param(max_iter=max(iters) + 2, eps_acc=1e-30)
    return O.solve_matop_cones(par, dense.vec_c, dense.mat_a, dense.vec_b, dense.seg_type, dense.seg_len,
                               snap_iters=iters, trace_cap=max(iters) + 3, use_ql=True)


def _check_iterates(T, dense, schedule, iters, tols):
    ro = _oracle_snaps(dense, iters)
    p = T.SolverParam()
    p.eps_acc = 1e-30
    fs = T.FusedSolver.from_dense(dense, p, schedule)
    t, s = fs.precond()
    N = dense.n + 2 * dense.m + 1
    # calc_precond (solver.rs:496-524) = 1 / max(abssum, eps_zero) with the block-cone minima of
    # product_group, against the oracle's dp_tau / dp_sigma
    assert np.allclose(t, ro.precond[:N], rtol=2e-5, atol=0), np.abs(t / ro.precond[:N] - 1).max()
    assert np.allclose(s, ro.precond[N:], rtol=2e-5, atol=0), np.abs(s / ro.precond[N:] - 1).max()
    done = 0
    for q, (it, tol) in enumerate(zip(iters, tols)):
        fs.run(it + 1 - done, poll_every=64)
        done = it + 1
        x, y = fs.iterate()
        rx, ry = ro.snaps[q][:N], ro.snaps[q][N:]
        sx = max(np.abs(rx).max(), 1e-6)
        sy = max(np.abs(ry).max(), 1e-6)
        assert np.abs(x - rx).max() <= tol * sx, (schedule, it, np.abs(x - rx).max() / sx)
        assert np.abs(y - ry).max() <= tol * sy, (schedule, it, np.abs(y - ry).max() / sy)
        st = fs.status()
        assert st.iters == it + 1 or st.iters == it       # running: next index
        tr = ro.trace[it]
        assert np.allclose(st.cri, tr[2:], rtol=max(50 * tol, 1e-3), atol=1e-5), (it, st.cri, tr)
    fs.destroy()


@pytest.mark.parametrize("schedule", ["reference", "fused", "carried"])
def test_iterates_lp(T, schedule):
    c, G, h = benchmark_lp(40, seed=1)
    lp = T.ProbLP(_mb(T, T.MatType.General(40, 1)).set_array(c.reshape(-1, 1)), _mb(T, T.MatType.General(80, 40)).set_array(G),
                  _mb(T, T.MatType.General(80, 1)).set_array(h.reshape(-1, 1)), _mb(T, T.MatType.General(0, 40)),
                  _mb(T, T.MatType.General(0, 1)))
    # f32 round-off accumulates over iterations: 1e-5 after 1-2, 1e-4 after 10, 2e-3 after 100
    _check_iterates(T, lp.dense(), schedule, [0, 1, 9, 99], [2e-5, 2e-5, 1e-4, 2e-3])


@pytest.mark.parametrize("schedule", ["reference", "fused", "carried"])
def test_iterates_socp(T, schedule):
    n, cones = 30, [5, 1, 0, 17, 99, 3]
    f, Gs, hs, cs, d = random_socp(n, cones, seed=2)
    socp = T.ProbSOCP(_mb(T, T.MatType.General(n, 1)).set_array(f.reshape(-1, 1)),
                      [_mb(T, T.MatType.General(G.shape[0], n)).set_array(G) for G in Gs],
                      [_mb(T, T.MatType.General(len(h_), 1)).set_array(h_.reshape(-1, 1)) for h_ in hs],
                      [_mb(T, T.MatType.General(n, 1)).set_array(c_.reshape(-1, 1)) for c_ in cs], d,
                      _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)))
    _check_iterates(T, socp.dense(), schedule, [0, 1, 9, 99], [2e-5, 2e-5, 1e-4, 2e-3])


@pytest.mark.parametrize("k", [9, 24, 70])     # 9, 24: the one-workgroup polar kernel (x_y and x_s in one launch); 70: the chain of launches
@pytest.mark.parametrize("schedule", ["fused", "carried"])
def test_iterates_sdp(T, schedule, k):
    n = 6
    c, syms = random_sdp(n, k, seed=3)
    sdp = T.ProbSDP(_mb(T, T.MatType.General(n, 1)).set_array(c.reshape(-1, 1)),
                    [_mb(T, T.MatType.SymPack(k)).set_array(s) for s in syms],
                    _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)), 1e-12)
    _check_iterates(T, sdp.dense(), schedule, [0, 1, 9, 49], [5e-5, 5e-5, 3e-4, 3e-3])


@pytest.mark.parametrize("schedule", ["fused", "carried"])
def test_iterates_many_psd_cones(T, schedule):
    """several PSD cones in one problem: orders <= 64 are projected group by group (all cones of one order, x_y and x_s,
    in ONE launch over a table of offsets), the order-70 cone by the chain of launches; a nonnegative segment in between
    moves the offsets off any regular stride"""
    from totsu_amd.problem import _Dense
    from totsu_amd import _lib
    n = 5
    orders = [3, 6, 6, 12, 6, 33, 70, 12, 1]
    blocks, bs, st, sl = [], [], [], []
    c0 = None
    for q, k in enumerate(orders):
        c, syms = random_sdp(n, k, seed=10 + q)
        sdp = T.ProbSDP(_mb(T, T.MatType.General(n, 1)).set_array(c.reshape(-1, 1)),
                        [_mb(T, T.MatType.SymPack(k)).set_array(sy) for sy in syms],
                        _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)), 1e-12)
        d = sdp.dense()
        sk = k * (k + 1) // 2
        blocks.append(np.asarray(d.mat_a, dtype=np.float32).reshape((n, d.m)).T[:sk])
        bs.append(np.asarray(d.vec_b, dtype=np.float32)[:sk])
        st.append(_lib.CONE_PSD)
        sl.append(sk)
        c0 = np.asarray(d.vec_c, dtype=np.float32) if c0 is None else c0 + np.asarray(d.vec_c, dtype=np.float32)
        sdp.drop()
        if q == 2:                                             # 7 nonnegative rows after the third cone
            rng = np.random.default_rng(99)
            blocks.append(rng.standard_normal((7, n)).astype(np.float32))
            bs.append(np.abs(rng.standard_normal(7)).astype(np.float32) + 1.0)
            st.append(_lib.CONE_RPOS)
            sl.append(7)
    a = np.asfortranarray(np.vstack(blocks))
    dense = _Dense(n, a.shape[0], a.ravel(order="F"), np.concatenate(bs), c0 / len(orders), st, sl)
    _check_iterates(T, dense, schedule, [0, 1, 9, 49], [5e-5, 5e-5, 3e-4, 3e-3])


def test_trait_level_equals_fused_on_lp(T):
    c, G, h = benchmark_lp(25, seed=4)
    mk = lambda: T.ProbLP(_mb(T, T.MatType.General(25, 1)).set_array(c.reshape(-1, 1)), _mb(T, T.MatType.General(50, 25)).set_array(G),
                          _mb(T, T.MatType.General(50, 1)).set_array(h.reshape(-1, 1)), _mb(T, T.MatType.General(0, 25)),
                          _mb(T, T.MatType.General(0, 1)))
    lp = mk()
    s = _par(T.Solver(T.F32HIP), max_iter=100_000, eps_acc=1e-4)
    s.trace = []
    x1, y1 = s.solve(lp.problem())
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 100_000, 1e-4
    fs = T.FusedSolver.from_dense(lp.dense(), p, "reference")
    x2, y2 = fs.solve()
    ro = O.solve_lp(O.param(max_iter=100000, eps_acc=1e-4), c, G, h, np.zeros((0, 25)), [])
    assert ro.status == O.OK
    # same objective as the f64 CPU path within 1e-4 relative (BASELINE.json north_star)
    pobj = float(c.astype(np.float64) @ ro.x)
    for x in (x1, x2):
        assert abs(float(c.astype(np.float64) @ x) - pobj) <= 1e-3 * (1 + abs(pobj))
    assert abs(s.trace[-1][0] - ro.iters) <= max(5, 0.02 * ro.iters)
    assert abs(fs.status().iters - ro.iters) <= max(5, 0.02 * ro.iters)
    lp.drop()


# ---- size-independent properties at BASELINE.json's LP and SOCP sizes (800 MB and 20 GB of A) ------------

@pytest.mark.parametrize("n,m", [(10_000, 20_000), (50_000, 100_000)])      # configs[1] (0.8 GB) and configs[2] (20 GB)
def test_gemv_properties_full_lp_size(T, n, m):
    import ctypes as C
    from totsu_amd._lib import lib
    D = T.DeviceBuffer
    A = D(n * m)
    lib.thip_gen_matrix(A.ptr, m, n, m, 0, 1, 0, 0, m, 0, 1.0, 0.0)
    x, y = D(n), D(m)
    lib.thip_gen_vector(x.ptr, n, 0, 2, 0, 1, 1.0, 0.0)
    lib.thip_gen_vector(y.ptr, m, 0, 3, 0, 1, 1.0, 0.0)
    Ax, Aty = D(m), D(n)
    lib.thip_transform_ge(0, m, n, 1.0, A.ptr, x.ptr, 0.0, Ax.ptr)
    lib.thip_transform_ge(1, m, n, 1.0, A.ptr, y.ptr, 0.0, Aty.ptr)
    hx, hy, hAx, hAty = x.to_host().astype(np.float64), y.to_host().astype(np.float64), Ax.to_host().astype(np.float64), Aty.to_host().astype(np.float64)
    # adjointness <A x, y> == <x, A^T y>, relative to the size of the terms (f32 sums of 2e8 products)
    lhs, rhs = hAx @ hy, hx @ hAty
    assert abs(lhs - rhs) <= 1e-5 * (np.abs(hAx) @ np.abs(hy))
    # spot rows / columns against f64 sums of the regenerated entries (counter-based generator)
    for r in (0, 1, 7777, m - 1):
        row = np.array([O.rng_uniform(0, 1, r + cc * m) for cc in range(n)], dtype=np.float64)
        assert abs(hAx[r] - row @ hx) <= 2e-5 * (np.abs(row) @ np.abs(hx))
    for cc in (0, 5, n - 1):
        col = np.array([O.rng_uniform(0, 1, r + cc * m) for r in range(m)], dtype=np.float64)
        assert abs(hAty[cc] - col @ hy) <= 2e-5 * (np.abs(col) @ np.abs(hy))
    # linearity: A(2x) == 2 A x bitwise-close, and beta accumulation
    lib.thip_transform_ge(0, m, n, 2.0, A.ptr, x.ptr, -1.0, Ax.ptr)       # 2Ax - Ax = Ax
    assert np.allclose(Ax.to_host(), hAx, rtol=1e-5, atol=1e-3)
    for d in (A, x, y, Ax, Aty):
        d.free()


# ---- convergence to the f64 CPU reference's objective (BASELINE.json north_star: 1e-4 relative) ----------

def _synth_dense_to_host(inst):
    a = inst.mat_a.to_host()[:inst.m * inst.n].astype(np.float64)
    return a, inst.vec_b_host.astype(np.float64), inst.vec_c_host.astype(np.float64)


_ORACLE_CACHE = {}


def _oracle_few_threads(key, fn):
    """the oracle's answer for a small instance, computed once per module and on 8 OpenMP threads: on the GPU box's
    256-CPU host every parallel region costs more than the 1000 x 500 loop it splits (2 x 110 s of the suite before)"""
    if key not in _ORACLE_CACHE:
        k = O.num_threads()
        O.set_num_threads(min(k, 8))
        try:
            _ORACLE_CACHE[key] = fn()
        finally:
            O.set_num_threads(k)
    return _ORACLE_CACHE[key]


@pytest.mark.parametrize("schedule", ["fused", "carried", "sweep"])
def test_synth_socp_converges_to_oracle_objective(T, schedule):
    """north_star's objective gate -- the f64 CPU reference's primal / dual objective within 1e-4 relative -- through every
    schedule the bench can report, the one-pass schedule (bench.py's default) included"""
    from totsu_amd import synth
    inst = synth.SocpInstance(500, 10, 99, seed=3)
    a, b, c = _synth_dense_to_host(inst)
    ro = _oracle_few_threads("synth_socp_500_10", lambda: O.solve_matop_cones(
        O.param(max_iter=400000, eps_acc=1e-5), c, a, b, [O.CONE_SOC] * 10, [100] * 10))
    assert ro.status == O.OK
    pobj, dobj = float(c @ ro.x), -float(b @ ro.y)
    assert abs(pobj - dobj) <= 1e-4 * (1 + abs(pobj))
    # f32 on the GPU: eps_acc 1e-4 (the reference itself runs f32 at 1e-3, benchmark_lp/src/main.rs:62-65);
    # the f64 oracle at 1e-5 is the objective being matched, within 1e-4 relative (BASELINE.json north_star)
    p = T.SolverParam()
    p.eps_acc, p.max_iter = 1e-4, 400_000
    fs = T.FusedSolver(inst.n, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, schedule,
                       sweep_min_bytes=0 if schedule == "sweep" else None)
    assert fs.schedule_in_use() == schedule
    x, y = fs.solve(poll_every=256)
    gp, gd = float(c @ x.astype(np.float64)), -float(b @ y.astype(np.float64))
    assert abs(gp - pobj) <= 1e-4 * (1 + abs(pobj)), (gp, pobj)
    assert abs(gd - dobj) <= 1e-4 * (1 + abs(dobj)), (gd, dobj)
    r4 = O.solve_matop_cones(O.param(max_iter=400000, eps_acc=1e-4), c, a, b, [O.CONE_SOC] * 10, [100] * 10)
    assert abs(fs.status().iters - r4.iters) <= 0.05 * r4.iters + 10
    fs.destroy()
    inst.free()


@pytest.mark.parametrize("nk,schedule", [((12, 20), "carried"), ((10, 33), "carried"), ((48, 20), "sweep"), ((60, 33), "sweep")])
def test_synth_sdp_converges_to_oracle_objective(T, nk, schedule):
    # (the one-pass kernel wants >= 40 columns: its two instances have n = 48 and 60)
    from totsu_amd import synth
    inst = synth.SdpInstance(nk[0], nk[1], seed=4)
    a, b, c = _synth_dense_to_host(inst)
    ro = O.solve_matop_cones(O.param(max_iter=200000, eps_acc=1e-5), c, a, b, [O.CONE_PSD], [inst.m], use_ql=True)
    assert ro.status == O.OK
    pobj = float(c @ ro.x)
    p = T.SolverParam()
    p.eps_acc, p.max_iter = 1e-4, 200_000
    fs = T.FusedSolver(inst.n, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, schedule,
                       sweep_min_bytes=0 if schedule == "sweep" else None)
    assert fs.schedule_in_use() == schedule
    x, y = fs.solve(poll_every=64)
    gobj = float(c @ x.astype(np.float64))
    assert abs(gobj - pobj) <= 3e-4 * (1 + abs(pobj))            # eps 1e-4 stop vs the eps 1e-5 answer
    r4 = O.solve_matop_cones(O.param(max_iter=200000, eps_acc=1e-4), c, a, b, [O.CONE_PSD], [inst.m], use_ql=True)
    assert abs(fs.status().iters - r4.iters) <= 0.05 * r4.iters + 10
    assert abs(gobj - float(c @ r4.x)) <= 1e-5 * (1 + abs(pobj))  # like for like: the oracle stopped at the same eps
    fs.destroy()
    inst.free()


# ---- QP / QCQP ("next" rows of SURVEY.md 8f): known-answer tests on the device ---------------------------

@pytest.mark.parametrize("path", ["trait", "fused", "carried"])
def test_kat_qp_qcqp(T, path):
    # totsu/tests/qp.rs:14-48 (== the crate doc-tests, totsu_f32cuda/src/lib.rs:31-75) and qcqp.rs:14-47
    L = T.F32HIP
    n = 2

    def qp():
        sym_p = _mb(T, T.MatType.SymPack(n)); sym_p[(0, 0)] = 1.0; sym_p[(1, 1)] = 1.0
        vec_q = _mb(T, T.MatType.General(n, 1)); vec_q[(0, 0)] = 1.0; vec_q[(1, 0)] = 2.0
        mat_g = _mb(T, T.MatType.General(1, n)); mat_g[(0, 0)] = -0.5; mat_g[(0, 1)] = -1.0 / 3.0
        vec_h = _mb(T, T.MatType.General(1, 1)); vec_h[(0, 0)] = -1.0
        return T.ProbQP(sym_p, vec_q, mat_g, vec_h, _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)), 1e-12)

    def qcqp():
        sp = [_mb(T, T.MatType.SymPack(n)), _mb(T, T.MatType.SymPack(n))]
        sp[0][(0, 0)] = 1.0; sp[0][(1, 1)] = 1.0
        vq = [_mb(T, T.MatType.General(n, 1)), _mb(T, T.MatType.General(n, 1))]
        vq[0][(0, 0)] = -5.0; vq[0][(1, 0)] = -4.0; vq[1][(0, 0)] = -0.5; vq[1][(1, 0)] = -1.0 / 3.0
        return T.ProbQCQP(sp, vq, [0.0, 1.0], _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)), 1e-12)

    for prob, want in ((qp(), [2.0, 0.0]), (qcqp(), [5.0, 4.0])):
        if path == "trait":
            x, _ = _par(T.Solver(L), max_iter=100_000, eps_acc=1e-5).solve(prob.problem())
        else:
            p = T.SolverParam()
            p.max_iter, p.eps_acc = 100_000, 1e-5
            x, _ = T.FusedSolver.from_dense(prob.dense(), p, path).solve()
        assert np.allclose(x[:2], want, atol=1e-3), (path, x)
        prob.drop()


def test_synth_lp_instance_matches_benchmark_lp_shape(T):
    # experimental/benchmark_lp/src/main.rs:14-57: G = [-I ; U(0,1)], h = [0 ; U(0,1)], c = -U(0,1); sharded rows
    from totsu_amd import synth
    n = 24
    full = synth.LpInstance(n, seed=1)
    A = full.mat_a.to_host()[:2 * n * n].reshape((n, 2 * n)).T
    assert np.array_equal(A[:n], -np.eye(n, dtype=np.float32))
    assert (A[n:] >= 0).all() and (A[n:] < 1).all() and A[n:].std() > 0.2
    assert not full.vec_b_host[:n].any() and (full.vec_c_host <= 0).all()
    parts = [synth.LpInstance(n, seed=1, rank=r, world=3) for r in range(3)]
    A3 = np.vstack([p.mat_a.to_host()[:p.m * n].reshape((n, p.m)).T for p in parts])
    assert np.array_equal(A3, A)
    assert np.array_equal(np.concatenate([p.vec_b_host for p in parts]), full.vec_b_host)
    ro = O.solve_lp(O.param(max_iter=200000, eps_acc=1e-4), full.vec_c_host, A, full.vec_b_host, np.zeros((0, n)), [])
    p = T.SolverParam()
    p.eps_acc, p.max_iter = 1e-4, 200_000
    fs = T.FusedSolver(n, full.m, full.mat_a, full.vec_b, full.vec_c, full.seg_type, full.seg_len, p, "carried")
    x, _ = fs.solve()
    pobj = float(full.vec_c_host.astype(np.float64) @ ro.x)
    assert ro.status == O.OK and abs(float(full.vec_c_host.astype(np.float64) @ x) - pobj) <= 1e-3 * (1 + abs(pobj))


def test_unchanged_caller_takes_the_fused_path(T):
    # `Solver::<F32HIP>::new().par(..).solve(lp.problem())` -- the reference's calling sequence, unchanged
    # (experimental/benchmark_lp/src/main.rs:60-69) -- runs the device-resident loop and fills `work` like solver.rs:317-320
    c, G, h = benchmark_lp(30, seed=8)
    lp = T.ProbLP(_mb(T, T.MatType.General(30, 1)).set_array(c.reshape(-1, 1)), _mb(T, T.MatType.General(60, 30)).set_array(G),
                  _mb(T, T.MatType.General(60, 1)).set_array(h.reshape(-1, 1)), _mb(T, T.MatType.General(0, 30)),
                  _mb(T, T.MatType.General(0, 1)))
    s = T.Solver(T.F32HIP)
    s.param.eps_acc = 1e-3
    x, y = s.solve(lp.problem())
    assert s.fused == "sweep" and s.iters > 10          # the asked-for schedule; a 60 x 30 problem runs "carried" (thip_solver_schedule_in_use)
    assert np.array_equal(lp.w_solver[:30], x) and np.array_equal(lp.w_solver[30:90], y)
    ro = O.solve_lp(O.param(eps_acc=1e-3), c, G, h, np.zeros((0, 30)), [])
    assert abs(s.iters - ro.iters) <= max(3, 0.02 * ro.iters)
    pobj = float(c.astype(np.float64) @ ro.x)
    assert abs(float(c.astype(np.float64) @ x) - pobj) <= 1e-3 * (1 + abs(pobj))
    lp.drop()


def test_config0_l1reg_lp_gpu_vs_oracle(T):
    # BASELINE.json configs[0] (examples/l1reg_lp) through the unchanged-caller path on F32HIP vs the f64 oracle
    from problems import l1reg_lp
    c, G, h = l1reg_lp(20, seed=0)
    n, m = c.size, h.size
    ro = O.solve_lp(O.param(eps_acc=1e-3), c, G, h, np.zeros((0, n)), [])
    lp = T.ProbLP(_mb(T, T.MatType.General(n, 1)).set_array(c.reshape(-1, 1)), _mb(T, T.MatType.General(m, n)).set_array(G),
                  _mb(T, T.MatType.General(m, 1)).set_array(h.reshape(-1, 1)), _mb(T, T.MatType.General(0, n)),
                  _mb(T, T.MatType.General(0, 1)))
    for fused in ("carried", None):
        s = T.Solver(T.F32HIP)
        s.fused = fused
        s.param.eps_acc = 1e-3
        x, y = s.solve(lp.problem())
        assert abs(s.iters - ro.iters) <= max(3, 0.03 * ro.iters), (fused, s.iters, ro.iters)
        pobj = float(c @ ro.x)
        assert abs(float(c @ x.astype(np.float64)) - pobj) <= 1e-3 * (1 + abs(pobj))
    lp.drop()


def test_plain_c_host_over_the_abi():
    # examples/c_api_demo.c: a C program linked against libtotsu_f32hip.so solves the nostd_cortex-m LP (no Python)
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "c_api_demo")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "x = [2.0000" in r.stdout or "x = [1.9999" in r.stdout
    assert "sparse: ||A c|| = 4.2426" in r.stdout          # the same LP through thip_sptile_* / thip_solver_set_sptile from plain C


@pytest.mark.parametrize("path", ["trait", "reference", "fused", "carried"])
def test_lp_with_equality_rows_zero_cone(T, path):
    # ProbLP with p > 0: the ConeZero block (cone_zero.rs:38-44: primal -> 0, dual -> identity) next to ConeRPos
    rng = np.random.default_rng(21)
    n, m, p_ = 12, 20, 4
    x0 = rng.uniform(0.1, 1.0, n)
    G = np.vstack([-np.eye(n), rng.uniform(0, 1, (m - n, n))])
    h = np.concatenate([np.zeros(n), G[n:] @ x0 + rng.uniform(0.1, 1, m - n)])
    A = rng.standard_normal((p_, n))
    b = A @ x0
    c = rng.uniform(0.1, 1, n)                      # c > 0, x >= 0: bounded
    ro = O.solve_lp(O.param(max_iter=400000, eps_acc=1e-5), c, G, h, A, b)
    assert ro.status == O.OK
    pobj = float(c @ ro.x)
    lp = T.ProbLP(_mb(T, T.MatType.General(n, 1)).set_array(c.reshape(-1, 1)), _mb(T, T.MatType.General(m, n)).set_array(G),
                  _mb(T, T.MatType.General(m, 1)).set_array(h.reshape(-1, 1)), _mb(T, T.MatType.General(p_, n)).set_array(A),
                  _mb(T, T.MatType.General(p_, 1)).set_array(b.reshape(-1, 1)))
    if path == "trait":
        x, y = _par(T.Solver(T.F32HIP), max_iter=400_000, eps_acc=1e-4).solve(lp.problem())
    else:
        pr = T.SolverParam()
        pr.max_iter, pr.eps_acc = 400_000, 1e-4
        x, y = T.FusedSolver.from_dense(lp.dense(), pr, path).solve()
    assert abs(float(c @ x.astype(np.float64)) - pobj) <= 1e-3 * (1 + abs(pobj)), (path, float(c @ x), pobj)
    assert np.abs(A @ x.astype(np.float64) - b).max() <= 5e-3 * (1 + np.abs(b).max())
    lp.drop()


def test_box_lp_vertex(T):
    # the known-answer LP of rust/totsu_f32hip/tests/kat_box_lp.rs, here through the Python mirror on the GPU
    rows = np.array([[-1., 0.], [0., -1.], [1., 0.], [0., 1.], [1., 1.]])
    rhs = np.array([0., 0., 3., 1.5, 4.])
    lp = T.ProbLP(_mb(T, T.MatType.General(2, 1)).iter_colmaj([-1., -2.]), _mb(T, T.MatType.General(5, 2)).set_array(rows),
                  _mb(T, T.MatType.General(5, 1)).iter_colmaj(rhs), _mb(T, T.MatType.General(0, 2)),
                  _mb(T, T.MatType.General(0, 1)))
    for fused in ("carried", None):
        s = T.Solver(T.F32HIP)
        s.fused = fused
        s.param.eps_acc, s.param.max_iter = 1e-4, 200_000
        x, _ = s.solve(lp.problem())
        assert np.allclose(x, [2.5, 1.5], atol=2e-3), (fused, x)
    lp.drop()


def test_cpp_trait_mirror_host():
    # examples/cpp_trait_demo.cpp over include/totsu_f32hip.hpp: the trait-level loop in C++ on three KATs, then the LP
    # through the C++ FusedSolver wrapper (f16 storage of A, switched to f32 and resumed with a tighter eps)
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "cpp_trait_demo")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("OK") == 5 and "fused-lp" in r.stdout and "sparse-lp" in r.stdout


@pytest.mark.parametrize("grid", [(2, 3), (3, 4), (8, 6)])       # (8, 6): the size of the reference's example, PSD order 48
def test_partitioning_sdp_gpu_vs_oracle(T, grid):
    # BASELINE.json configs[3] template (examples/partitioning_sdp): PSD cone + equality rows through ProbSDP
    from problems import partitioning_sdp
    w, syms_f, mat_a, vec_b = partitioning_sdp(*grid, seed=1)
    l = grid[0] * grid[1]
    n = w.size
    ro = O.solve_sdp(O.param(max_iter=400000, eps_acc=1e-5), w, syms_f, mat_a, vec_b, 1e-12, use_ql=True)
    assert ro.status == O.OK
    pobj = float(w @ ro.x)
    sdp = T.ProbSDP(_mb(T, T.MatType.General(n, 1)).set_array(w.reshape(-1, 1)),
                    [_mb(T, T.MatType.SymPack(l)).set_array(s) for s in syms_f],
                    _mb(T, T.MatType.General(l, n)).set_array(mat_a), _mb(T, T.MatType.General(l, 1)).set_array(vec_b.reshape(-1, 1)),
                    1e-12)
    for fused in (("carried", None) if l <= 6 else ("carried",)):
        s = T.Solver(T.F32HIP)
        s.fused = fused
        s.param.eps_acc, s.param.max_iter = 1e-4, 400_000
        x, _ = s.solve(sdp.problem())
        assert abs(float(w @ x.astype(np.float64)) - pobj) <= 1e-3 * (1 + abs(pobj)), (fused, float(w @ x), pobj)
        diag = [x[c * (c + 1) // 2 + c] for c in range(l)]
        assert np.allclose(diag, 1.0, atol=5e-3)
    sdp.drop()


def test_abi_rejects_bad_arguments(T):
    # the boundary fails loudly: bad cone segments, bad schedule, PSD segment that is not triangular, short work
    from totsu_amd._lib import ThipError, E_INVALID, E_WORK, lib
    a = np.zeros(6, np.float32)
    with pytest.raises(ThipError) as e:
        T.FusedSolver(2, 3, a, np.zeros(3, np.float32), np.zeros(2, np.float32), [1], [2])        # segments cover 2 of 3 rows
    assert e.value.code == E_INVALID
    with pytest.raises(ThipError) as e:
        T.FusedSolver(2, 4, np.zeros(8, np.float32), np.zeros(4, np.float32), np.zeros(2, np.float32), [4], [4])   # 4 is not k(k+1)/2
    assert e.value.code == E_INVALID
    with pytest.raises(KeyError):
        T.FusedSolver(2, 3, a, np.zeros(3, np.float32), np.zeros(2, np.float32), [1], [3], schedule="nope")
    x = T.DeviceBuffer(6)
    w = T.DeviceBuffer(4)
    with pytest.raises(ThipError) as e:
        lib.thip_proj_psd(6, x.ptr, 1e-12, w.ptr, 4)
    assert e.value.code == E_WORK
    x.free(); w.free()


def test_gemv_bandwidth_floor_at_lp_size(T):
    # regression guard, deliberately loose: at BASELINE.json's LP size (800 MB of A) the dual-GEMV kernel has measured
    # 5.8-6.1 TB/s on MI355X; anything below 4 TB/s means a tiling / codegen regression (e.g. serialised loads)
    import ctypes as C
    from totsu_amd import synth
    from totsu_amd._lib import lib
    inst = synth.LpInstance(10_000, seed=0)
    p = T.SolverParam()
    p.eps_acc = 0.0
    fs = T.FusedSolver(inst.n, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, "carried")
    fs.run(10, poll_every=10)
    lib.thip_prof_enable(1)
    fs.run(50, poll_every=50)
    nl, ms = C.c_int64(), C.c_double()
    lib.thip_prof_read(C.byref(nl), C.byref(ms))
    lib.thip_prof_enable(0)
    passes, nbytes = fs.passes()
    assert nl.value == 50 * passes
    gbps = nbytes / (ms.value / nl.value * 1e-3) / 1e9
    assert gbps >= 4000.0, gbps
    fs.destroy()
    inst.free()


@pytest.mark.parametrize("schedule", ["reference", "fused", "carried"])
def test_two_solves_of_the_same_problem_are_bitwise_identical(T, schedule):
    # every reduction is two-stage with a fixed order (no float atomics): same input, same bits -- iterates, criteria
    # and iteration count -- which is also what keeps the replicated vectors of a sharded solve in lockstep
    from totsu_amd import synth
    inst = synth.SocpInstance(700, 12, 99, seed=11)
    p = T.SolverParam()
    p.eps_acc, p.max_iter = 1e-4, 50_000
    out = []
    for rep in range(2):
        fs = T.FusedSolver(inst.n, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, schedule)
        fs.run(37, poll_every=37)
        mid = fs.iterate()
        r = fs.run(-1, poll_every=64)
        x, y = fs.solution()
        out.append((mid[0].copy(), mid[1].copy(), r.iters, r.cri, x.copy(), y.copy()))
        fs.destroy()
    a, b = out
    assert a[2] == b[2] and a[3] == b[3]
    for u, v in ((a[0], b[0]), (a[1], b[1]), (a[4], b[4]), (a[5], b[5])):
        assert np.array_equal(u, v)
    inst.free()


def test_bitwise_reproducible_with_the_autotune_switched_off_through_the_api(T):
    # at sizes where thip_solver_init times GEMV plans, two solves may pick different plans (timings) and then agree to
    # f32 round-off only; thip_solver_set_gemv_autotune(s, 0) (FusedSolver(gemv_autotune=False)) pins the shape
    # heuristic: same bits run to run, no environment variable involved
    from totsu_amd import synth
    inst = synth.SocpInstance(3000, 30, 99, seed=3)          # 3000 x 3000 = 9e6 entries: above the autotune threshold
    p = T.SolverParam()
    p.eps_acc = 0.0
    out = []
    for rep in range(2):
        fs = T.FusedSolver(inst.n, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, "carried",
                           gemv_autotune=False)
        assert fs.gemv_plan()["target_workgroups"] == 0      # no timed plan: the heuristic
        fs.run(200, poll_every=50)
        out.append(fs.iterate())
        fs.destroy()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    fs = T.FusedSolver(inst.n, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, "carried")
    assert fs.gemv_plan()["target_workgroups"] > 0           # default: autotuned
    fs.run(200, poll_every=50)
    x, y = fs.iterate()
    fs.destroy()
    assert np.abs(x - out[0][0]).max() <= 1e-4 * np.abs(x).max()
    inst.free()


def test_lda_pad_switch_through_the_api(T):
    # m = 12 x 101 = 1212 rows (not a multiple of 16): by default the f32 passes stream a padded library-owned copy;
    # thip_solver_set_lda_pad(s, 0) streams the caller's matrix as it is.  Same iterates to round-off either way, and a
    # re-init after the caller rewrote A in place must see the new matrix (the copy was never refreshed)
    from totsu_amd import synth
    from totsu_amd._lib import lib
    inst = synth.SocpInstance(4000, 12, 100, seed=5)
    assert inst.m % 16 != 0
    p = T.SolverParam()
    p.eps_acc = 0.0
    its = []
    for pad in (None, 0):
        fs = T.FusedSolver(inst.n, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, "carried",
                           lda_pad=pad)
        fs.run(50, poll_every=50)
        its.append(fs.iterate())
        if pad is None:
            # rewrite A in place (scale by 2) and start again: the solve must be the one of the NEW matrix
            lib.thip_scale(inst.m * inst.n, 2.0, inst.mat_a.ptr)
            fs.reinit()
            fs.run(50, poll_every=50)
            new_pad = fs.iterate()
            fs.destroy()
            fs = T.FusedSolver(inst.n, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p,
                               "carried", lda_pad=0)
            fs.run(50, poll_every=50)
            new_nopad = fs.iterate()
            lib.thip_scale(inst.m * inst.n, 0.5, inst.mat_a.ptr)
            assert np.abs(new_pad[0] - new_nopad[0]).max() <= 1e-4 * np.abs(new_nopad[0]).max()
            assert np.abs(new_pad[0] - its[0][0]).max() > 1e-3 * np.abs(its[0][0]).max()     # it really is another problem
        fs.destroy()
    assert np.abs(its[0][0] - its[1][0]).max() <= 1e-4 * np.abs(its[1][0]).max()
    assert np.abs(its[0][1] - its[1][1]).max() <= 1e-4 * np.abs(its[1][1]).max()
    inst.free()


# ---- the reference's QP / QCQP examples as parity cases (constructions in tests/problems.py) ------------------

def test_svm_qp_example_gpu_vs_oracle(T):
    # examples/svm_qp/src/main.rs:47-106 through ProbQP (sqrt of the kernel matrix on the device eigen engine,
    # rotated cone + nonneg + zero cone) and the fused loop, eps_acc 1e-3 like the example (main.rs:98-101)
    from problems import svm_qp, sym_pack
    l = 30                                  # the example uses 50 samples; 30 keeps the GPU suite short
    q = svm_qp(l, seed=0)
    ro = O.solve_qp(O.param(max_iter=2_000_000, eps_acc=1e-3), sym_pack(q["sym_p"]), q["vec_q"], q["mat_g"], q["vec_h"],
                    q["mat_a"], q["vec_b"])
    assert ro.status == O.OK
    sym_p = _mb(T, T.MatType.SymPack(l)).set_by_fn(lambda r, c: q["sym_p"][r, c])
    qp = T.ProbQP(sym_p, _mb(T, T.MatType.General(l, 1)).set_array(q["vec_q"].reshape(-1, 1)),
                  _mb(T, T.MatType.General(l, l)).set_array(q["mat_g"]),
                  _mb(T, T.MatType.General(l, 1)).set_array(q["vec_h"].reshape(-1, 1)),
                  _mb(T, T.MatType.General(1, l)).set_array(q["mat_a"]),
                  _mb(T, T.MatType.General(1, 1)).set_array(q["vec_b"].reshape(-1, 1)), 1e-12)
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 2_000_000, 1e-3
    fs = T.FusedSolver.from_dense(qp.dense(), p, "carried")
    x, _ = fs.solve(poll_every=256)
    st = fs.status()
    fs.destroy()
    assert st.state == 0 and abs(st.iters - ro.iters) <= 0.1 * ro.iters, (st.iters, ro.iters)
    a, ar = x[:l].astype(np.float64), ro.x[:l]
    obj = lambda v: 0.5 * v @ q["sym_p"] @ v - v.sum()
    assert abs(obj(a) - obj(ar)) <= 2e-3 * (1 + abs(obj(ar))), (obj(a), obj(ar))
    assert np.abs(a - ar).max() <= 2e-2 * np.abs(ar).max()
    qp.drop()


def test_trajplan_qcqp_example_gpu_vs_oracle(T):
    # examples/trajplan_qcqp/src/main.rs:19-151 through ProbQCQP: 29 matrix square roots, 29 rotated cones, 12
    # equality rows; eps_acc 1e-3 like the example
    from problems import sym_pack, trajplan_qcqp
    t_cap, a_cap = 15, 70.0                 # the example: 30 grids, a_cap 90; this size keeps the bound active too
    n = 2 * t_cap
    c = trajplan_qcqp(t_cap, a_cap)
    ro = O.solve_qcqp(O.param(max_iter=2_000_000, eps_acc=1e-3), [sym_pack(s) for s in c["syms_p"]], c["vecs_q"],
                      c["scls_r"], c["mat_a"], c["vec_b"])
    assert ro.status == O.OK
    syms = [_mb(T, T.MatType.SymPack(n)).set_by_fn(lambda r, cc, S=S: S[r, cc]) for S in c["syms_p"]]
    vqs = [_mb(T, T.MatType.General(n, 1)).set_array(v.reshape(-1, 1)) for v in c["vecs_q"]]
    prob = T.ProbQCQP(syms, vqs, list(c["scls_r"]), _mb(T, T.MatType.General(12, n)).set_array(c["mat_a"]),
                      _mb(T, T.MatType.General(12, 1)).set_array(c["vec_b"].reshape(-1, 1)), 1e-12)
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 2_000_000, 1e-3
    fs = T.FusedSolver.from_dense(prob.dense(), p, "carried")
    x, _ = fs.solve(poll_every=256)
    st = fs.status()
    fs.destroy()
    assert st.state == 0 and abs(st.iters - ro.iters) <= 0.3 * ro.iters, (st.iters, ro.iters)   # f32 square roots of 1/dt^4-scaled matrices
    xg, xr = x[:n].astype(np.float64), ro.x[:n]
    obj = lambda v: 0.5 * v @ c["syms_p"][0] @ v
    assert abs(obj(xg) - obj(xr)) <= 3e-3 * (1 + abs(obj(xr))), (obj(xg), obj(xr))
    # the primal criterion is relative to 1 + ||b|| (solver.rs:605), and b carries the 0.5 a_cap^2 = 2450 of every
    # constraint (||b|| ~ 9000): eps_acc 1e-3 leaves the equality rows satisfied to ~0.1 in absolute terms
    assert np.abs(c["mat_a"] @ xg - c["vec_b"]).max() < 0.15
    assert np.abs(xg - xr).max() <= 0.1 * np.abs(xr).max()
    prob.drop()


def test_toruscompl_socp_example_gpu_vs_oracle(T):
    # examples/toruscompl_socp/src/main.rs:43-268 at the example's size through ProbSOCP and the fused loop: 158 cones
    # of 1 + 2 rows and 317 cones of 1 + 0 rows in one batched launch, 112 zero-cone rows; eps_acc 1e-3 like the example
    from problems import toruscompl_socp
    q = toruscompl_socp(9, 7, 0.2)
    n = q["vec_f"].size
    ro = O.solve_socp(O.param(max_iter=1_000_000, eps_acc=1e-3), q["vec_f"], q["mats_g"], q["vecs_h"], q["vecs_c"],
                      q["scls_d"], q["mat_a"], q["vec_b"])
    assert ro.status == O.OK
    col = lambda v: np.asarray(v, np.float32).reshape(-1, 1)
    socp = T.ProbSOCP(_mb(T, T.MatType.General(n, 1)).set_array(col(q["vec_f"])),
                      [_mb(T, T.MatType.General(G.shape[0], n)).set_array(G) for G in q["mats_g"]],
                      [_mb(T, T.MatType.General(len(h_), 1)).set_array(col(h_)) for h_ in q["vecs_h"]],
                      [_mb(T, T.MatType.General(n, 1)).set_array(col(c_)) for c_ in q["vecs_c"]], list(q["scls_d"]),
                      _mb(T, T.MatType.General(q["vec_b"].size, n)).set_array(q["mat_a"]),
                      _mb(T, T.MatType.General(q["vec_b"].size, 1)).set_array(col(q["vec_b"])))
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 1_000_000, 1e-3
    for sched in ("reference", "carried"):
        fs = T.FusedSolver.from_dense(socp.dense(), p, sched)
        x, _ = fs.solve(poll_every=64)
        st = fs.status()
        fs.destroy()
        assert st.state == 0 and abs(st.iters - ro.iters) <= 0.05 * ro.iters + 5, (sched, st.iters, ro.iters)
        obj, obj_r = float(q["vec_f"] @ x.astype(np.float64)), float(q["vec_f"] @ ro.x)
        assert abs(obj - obj_r) <= 1e-3 * (1 + abs(obj_r)), (sched, obj, obj_r)
        assert np.abs(x - ro.x).max() <= 1e-2 * np.abs(ro.x).max(), sched


def test_schedules_agree_at_the_full_socp_size(T):
    # BASELINE.json configs[2] (n = 50 000, 1000 cones, A = 20 GB): no CPU oracle finishes here, so parity is carried by
    # a size-independent property -- the reference's own op sequence (6 GEMVs, `reference`), the 2-pass `carried`
    # schedule and the one-pass `sweep` (the headline's) are the same iteration, whose small-size parity with the oracle is
    # pinned by test_iterates_socp / test_sweep_iterates_socp (the oracle itself at THIS size:
    # test_gpu_configs.py::test_c3_full_size_sweep_vs_oracle)
    from totsu_amd import synth
    inst = synth.SocpInstance(50_000, 1000, 99, seed=0)
    p = T.SolverParam()
    p.eps_acc = 0.0
    out = {}
    for sched in ("reference", "carried", "sweep"):
        fs = T.FusedSolver(inst.n, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, sched)
        assert fs.schedule_in_use() == sched
        r = fs.run(15, poll_every=15)
        out[sched] = (fs.iterate(), r)
        fs.destroy()
    (xr, yr), rr = out["reference"]
    for sched in ("carried", "sweep"):
        (xc, yc), rc = out[sched]
        assert rr.iters == rc.iters == 15
        assert np.abs(xr - xc).max() <= 2e-4 * np.abs(xr).max() and np.abs(yr - yc).max() <= 2e-4 * np.abs(yr).max(), sched
        assert np.allclose(rr.cri, rc.cri, rtol=2e-3, atol=1e-6), sched
    inst.free()


def test_compensated_state_lowers_the_f32_floor(T):
    # In plain f32 the iterate stops moving once an update is below half an ulp of the entry it is added to: the dual
    # criterion of this n = 200 SOCP freezes at 6.3e-6 (the same number in a numpy f32 emulation of the reference's
    # loop, 1.2e-14 in f64).  thip_param.state_arith = THIP_STATE_COMPENSATED (Kahan terms for x_x, x_y, x_s, u, v)
    # takes it to ~1e-7 with the same f32 storage and arithmetic; THIP_STATE_PLAIN is the reference's literal f32.
    f, Gs, hs, cs, d = random_socp(200, [99] * 6, seed=1)
    n = 200
    socp = T.ProbSOCP(_mb(T, T.MatType.General(n, 1)).set_array(f.reshape(-1, 1)),
                      [_mb(T, T.MatType.General(G.shape[0], n)).set_array(G) for G in Gs],
                      [_mb(T, T.MatType.General(len(h_), 1)).set_array(h_.reshape(-1, 1)) for h_ in hs],
                      [_mb(T, T.MatType.General(n, 1)).set_array(c_.reshape(-1, 1)) for c_ in cs], d,
                      _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)))
    dense = socp.dense()
    p = T.SolverParam()
    p.eps_acc = 0.0
    floors = {}
    for flag in ("0", "1"):
        p.state_arith = "compensated" if flag == "1" else "plain"
        fs = T.FusedSolver.from_dense(dense, p, "carried")
        r = fs.run(15000, poll_every=64)
        floors[flag] = r.cri[1]
        assert r.cri[0] < 1e-6 and r.cri[2] < 1e-6
        fs.destroy()
    assert 2e-6 < floors["0"] < 2e-5, floors
    assert floors["1"] < 5e-7, floors


def test_solver_can_be_initialised_again_after_a_finished_solve(T):
    # thip_solver_init after a terminated solve starts a fresh solve: same iteration count, same (1/tau-scaled) answer
    c, G, h = benchmark_lp(48, seed=5)
    lp = T.ProbLP(_mb(T, T.MatType.General(48, 1)).set_array(c.reshape(-1, 1)), _mb(T, T.MatType.General(96, 48)).set_array(G),
                  _mb(T, T.MatType.General(96, 1)).set_array(h.reshape(-1, 1)), _mb(T, T.MatType.General(0, 48)),
                  _mb(T, T.MatType.General(0, 1)))
    p = T.SolverParam()
    p.eps_acc, p.max_iter = 1e-4, 200_000
    for sched in ("fused", "carried"):
        fs = T.FusedSolver.from_dense(lp.dense(), p, sched)
        x1, y1 = fs.solve()
        it1 = fs.status().iters
        fs.reinit()
        assert fs.status().state == -1 and fs.status().iters == 0
        x2, y2 = fs.solve()
        assert fs.status().iters == it1
        assert np.array_equal(x1, x2) and np.array_equal(y1, y2)
        fs.destroy()
    lp.drop()


def test_storage_switch_inside_a_carried_solve_rebuilds_the_carried_products(T):
    # carried keeps A^T x_y and A x_x of the current iterate; after bf16 -> f32 they must be recomputed with the f32
    # matrix, or the first y-update mixes the two matrices (a 2^-9 relative one-off error).  The fused schedule carries
    # nothing, so it is the reference: same switch, same iterates.
    f, Gs, hs, cs, d = random_socp(60, [9, 30, 17], seed=4)
    n = 60
    socp = T.ProbSOCP(_mb(T, T.MatType.General(n, 1)).set_array(f.reshape(-1, 1)),
                      [_mb(T, T.MatType.General(G.shape[0], n)).set_array(G) for G in Gs],
                      [_mb(T, T.MatType.General(len(h_), 1)).set_array(h_.reshape(-1, 1)) for h_ in hs],
                      [_mb(T, T.MatType.General(n, 1)).set_array(c_.reshape(-1, 1)) for c_ in cs], d,
                      _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)))
    p = T.SolverParam()
    p.eps_acc = 0.0
    its = {}
    for sched in ("fused", "carried"):
        fs = T.FusedSolver.from_dense(socp.dense(), p, sched, a_storage="bf16")
        fs.run(200, poll_every=50)
        fs.set_a_storage("f32")
        fs.run(3, poll_every=3)
        its[sched] = fs.iterate()
        fs.destroy()
    for a, b in zip(its["fused"], its["carried"]):
        sc = np.abs(a).max()
        assert np.abs(a - b).max() <= 2e-5 * sc, np.abs(a - b).max() / sc
    socp.drop()


def test_literal_reference_cones_over_the_hip_backend(T):
    """the trait-level loop with the reference's literal cone code (host loop over get_mut, get + norm + scale + set,
    map_eig with a host closure) over F32HIP -- no device fast path -- reproduces the known answers"""
    import totsu_amd.cone as cone_mod
    L = T.F32HIP
    cone_mod.DEVICE_FAST_PATHS = False
    try:
        # nostd_cortex-m LP (ConeRPos through get_mut): x = [2, 2]
        lp = T.ProbLP(_mb(T, T.MatType.General(2, 1)).iter_colmaj([-1., 0.]), _mb(T, T.MatType.General(3, 2)).iter_colmaj([4., -1., -1., -1., 4., -1.]),
                      _mb(T, T.MatType.General(3, 1)).iter_colmaj([6., 6., 1.]), _mb(T, T.MatType.General(0, 2)), _mb(T, T.MatType.General(0, 1)))
        x, _ = _par(T.Solver(L), max_iter=100_000, eps_acc=1e-5).solve(lp.problem())
        assert np.allclose(x, [2., 2.], atol=1e-3)
        lp.drop()
        # totsu/tests/socp.rs test_socp1 (ConeSOC through get / norm / scale / set): x = [-1, -1]
        n = 2
        socp = T.ProbSOCP(_mb(T, T.MatType.General(n, 1)).iter_colmaj([1., 1.]), [_mb(T, T.MatType.General(2, 2)).iter_colmaj([1., 0., 0., 1.])],
                          [_mb(T, T.MatType.General(2, 1))], [_mb(T, T.MatType.General(n, 1))], [2. ** 0.5],
                          _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)))
        x, _ = _par(T.Solver(L), max_iter=100_000, eps_acc=1e-5).solve(socp.problem())
        assert np.allclose(x, [-1., -1.], atol=1e-3)
        socp.drop()
        # totsu_core/tests/solver.rs (ConePSD through map_eig with the host closure e > 0 -> Some(e)): x = -2
        op_c = T.MatOp(L, T.MatType.General(1, 1), np.array([1.], np.float32))
        op_a = T.MatOp(L, T.MatType.General(3, 1), np.array([0., -1. * 1.41421356, -3.], np.float32))
        op_b = T.MatOp(L, T.MatType.General(3, 1), np.array([1., 0., 10.], np.float32))
        sv = _par(T.Solver(L), max_iter=100_000, eps_acc=1e-5)
        cone_w = np.zeros(T.ConePSD.query_worklen(L, 3), dtype=np.float32)
        cone = T.ConePSD(L, cone_w, sv.param.eps_zero)
        work = np.zeros(T.Solver.query_worklen(op_a.size()), dtype=np.float32)
        x, _ = sv.solve((op_c, op_a, op_b, cone, work))
        assert abs(x[0] + 2.0) <= 1e-3
    finally:
        cone_mod.DEVICE_FAST_PATHS = True
