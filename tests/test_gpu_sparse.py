"""GPU: sparse (CSR) Operator on the trait-level path.  The operator contract is the
reference's: trans_op is the adjoint of op and absadd_* are the |.| column / row sums (operator.rs:40-154; the
adjointness-test pattern of examples/imgnr_udef/src/prob_op_a.rs:137-203); the solve must agree with the dense
oracle on the same matrix."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O
from problems import l1reg_lp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import totsu_amd
    from totsu_amd import _lib
    _lib.init()
    return totsu_amd


def _sl(L, a):
    return L.Sl.new_mut(np.ascontiguousarray(a, dtype=np.float32))


@pytest.mark.parametrize("two_copies", [False, True])
@pytest.mark.parametrize("shape,density", [((1, 1), 1.0), ((50, 30), 0.1), ((300, 1000), 0.01), ((5000, 4000), 0.002),
                                           ((64, 64), 0.9), ((7, 2000), 0.5), ((2000, 3), 0.4),
                                           # several 4096 x 4096 tiles; tall dense columns (a wave's entries share a column);
                                           # an empty row block in the middle; a visit below the LDS-staging threshold
                                           ((9000, 5000), 0.01), ((13000, 70), 0.9), ((300, 20000), 0.02), ((12500, 4097), 0.0008),
                                           # one item of 74 small tiles: the flat walk of the LITE instance refills its table of 64
                                           ((64, 300000), 0.0015), ((300000, 64), 0.0015)])
def test_sparse_operator_contract(T, shape, density, two_copies):
    from totsu_amd.sparse import SparseMatOp
    L = T.F32HIP
    rng = np.random.default_rng(shape[0] + shape[1])
    a = sp.random(shape[0], shape[1], density=density, format="csr", random_state=rng, dtype=np.float64)
    a.data = rng.standard_normal(a.nnz)
    if shape == (12500, 4097):
        a = a.tolil(); a[4096:8192, :] = 0.0; a = a.tocsr(); a.eliminate_zeros()
    op = SparseMatOp(L, a, two_copies=two_copies)
    d = a.toarray()
    x = rng.standard_normal(shape[1]).astype(np.float32)
    y0 = rng.standard_normal(shape[0]).astype(np.float32)
    sx, sy = _sl(L, x), _sl(L, y0.copy())
    op.op(0.7, sx, -0.3, sy)
    ref = 0.7 * d @ x - 0.3 * y0
    scale = 0.7 * np.abs(d) @ np.abs(x) + 0.3 * np.abs(y0) + 1e-6
    assert np.all(np.abs(sy.get_ref() - ref) <= 1e-5 * scale)
    sy2, sx2 = _sl(L, y0), _sl(L, x.copy())
    op.trans_op(-1.5, sy2, 0.5, sx2)
    ref = -1.5 * d.T @ y0 + 0.5 * x
    scale = 1.5 * np.abs(d.T) @ np.abs(y0) + 0.5 * np.abs(x) + 1e-6
    assert np.all(np.abs(sx2.get_ref() - ref) <= 1e-5 * scale)
    # adjointness <A x, y> == <x, A^T y>
    ax, aty = _sl(L, np.zeros(shape[0])), _sl(L, np.zeros(shape[1]))
    op.op(1.0, sx, 0.0, ax)
    op.trans_op(1.0, sy2, 0.0, aty)
    lhs = float(ax.get_ref().astype(np.float64) @ y0)
    rhs = float(x.astype(np.float64) @ aty.get_ref().astype(np.float64))
    assert abs(lhs - rhs) <= 1e-5 * (np.abs(d) @ np.abs(x)) @ np.abs(y0) + 1e-6
    t0 = rng.uniform(0, 1, shape[1]).astype(np.float32)
    s0 = rng.uniform(0, 1, shape[0]).astype(np.float32)
    st, ss = _sl(L, t0.copy()), _sl(L, s0.copy())
    op.absadd_cols(st)
    op.absadd_rows(ss)
    assert np.allclose(st.get_ref(), t0 + np.abs(d).sum(axis=0), rtol=1e-5, atol=1e-6)
    assert np.allclose(ss.get_ref(), s0 + np.abs(d).sum(axis=1), rtol=1e-5, atol=1e-6)
    op.drop()


def test_sparse_l1reg_lp_solve_matches_dense_oracle(T):
    # the l1reg_lp matrix (examples/l1reg_lp) is ~80 % zeros: solve it with the sparse operator on the trait-level path
    from totsu_amd.sparse import SparseMatOp
    L = T.F32HIP
    c, G, h = l1reg_lp(20, seed=0)
    n, m = c.size, h.size
    op_c = T.MatOp(L, T.MatType.General(n, 1), c.astype(np.float32))
    op_a = SparseMatOp(L, sp.csr_matrix(G))
    op_b = T.MatOp(L, T.MatType.General(m, 1), h.astype(np.float32))
    s = T.Solver(L)
    s.param.eps_acc = 1e-3
    work = np.zeros(T.Solver.query_worklen((m, n)), dtype=np.float32)
    x, y = s.solve((op_c, op_a, op_b, T.ConeRPos(L), work))
    ro = O.solve_lp(O.param(eps_acc=1e-3), c, G, h, np.zeros((0, n)), [])
    assert abs(s.iters - ro.iters) <= max(3, 0.03 * ro.iters)
    pobj = float(c @ ro.x)
    assert abs(float(c @ x.astype(np.float64)) - pobj) <= 1e-3 * (1 + abs(pobj))
    op_a.drop()


@pytest.mark.parametrize("schedule,two_copies", [("reference", True), ("fused", True), ("carried", True), ("reference", False),
                                                 ("fused", False), ("carried", False), ("sweep", False)])
def test_fused_loop_on_csr_matches_dense(T, schedule, two_copies):
    # the device-resident loop with A given as scipy.sparse (one tiled copy, or round 5's two CSR copies): same iterates as
    # with the dense matrix
    c, G, h = l1reg_lp(20, seed=1)
    n, m = c.size, h.size
    p = T.SolverParam()
    p.eps_acc = 1e-3
    dense = T.FusedSolver(n, m, np.asfortranarray(G).ravel(order="F").astype(np.float32), h, c, [1], [m], p,
                          "carried" if schedule == "sweep" else schedule)
    xd, yd = dense.solve()
    sparse = T.FusedSolver(n, m, sp.csr_matrix(G.astype(np.float32)), h, c, [1], [m], p, schedule, sparse_two_copies=two_copies)
    assert sparse.schedule_in_use() == schedule
    xs, ys = sparse.solve()
    assert abs(dense.status().iters - sparse.status().iters) <= max(3, 0.02 * dense.status().iters)
    ro = O.solve_lp(O.param(eps_acc=1e-3), c, G, h, np.zeros((0, n)), [])
    pobj = float(c @ ro.x)
    for x in (xd, xs):
        assert abs(float(c @ x.astype(np.float64)) - pobj) <= 1e-3 * (1 + abs(pobj))
    td, sd = dense.precond()
    ts, ss = sparse.precond()
    assert np.allclose(td, ts, rtol=1e-5) and np.allclose(sd, ss, rtol=1e-5)
    dense.destroy()
    sparse.destroy()


@pytest.mark.parametrize("schedule,two_copies", [("carried", True), ("carried", False), ("sweep", False), ("fused", False)])
def test_fused_loop_sparse_socp_iterates_vs_oracle(T, schedule, two_copies):
    # sparse SOCP blocks (90 % zeros): iterates of the sparse fused loop against the dense f64 oracle
    from problems import random_socp
    n, cones = 40, [6, 25, 0, 11]
    f, Gs, hs, cs, d = random_socp(n, cones, seed=7)
    rng = np.random.default_rng(0)
    Gs = [g * (rng.uniform(0, 1, g.shape) < 0.15) for g in Gs]
    rows = [np.vstack([-c_.reshape(1, n), -g]) for g, c_ in zip(Gs, cs)]
    A = np.vstack(rows).astype(np.float32)
    b = np.concatenate([np.concatenate([[dd], h_]) for dd, h_ in zip(d, hs)]).astype(np.float32)
    seg_t, seg_l = [2] * len(cones), [1 + k for k in cones]
    m = A.shape[0]
    ro = O.solve_matop_cones(O.param(max_iter=60, eps_acc=1e-300), f, np.asfortranarray(A).ravel(order="F"), b, seg_t, seg_l,
                             snap_iters=[0, 9, 49], trace_cap=64)
    p = T.SolverParam()
    p.eps_acc = 0.0
    fs = T.FusedSolver(n, m, sp.csr_matrix(A), b, f, seg_t, seg_l, p, schedule, sparse_two_copies=two_copies)
    N = n + 2 * m + 1
    done = 0
    for q, (it, tol) in enumerate(zip([0, 9, 49], [3e-5, 2e-4, 2e-3])):
        fs.run(it + 1 - done, poll_every=64)
        done = it + 1
        x, y = fs.iterate()
        rx, ry = ro.snaps[q][:N], ro.snaps[q][N:]
        assert np.abs(x - rx).max() <= tol * max(np.abs(rx).max(), 1e-6)
        assert np.abs(y - ry).max() <= tol * max(np.abs(ry).max(), 1e-6)
    fs.destroy()


def _iterates_vs_oracle(T, A, b, c, seg_t, seg_l, iters, tols, schedule="sweep"):
    """iterates of the fused loop on the tiled sparse copy of A against the f64 oracle on the dense-ified A"""
    m, n = A.shape
    Ad = np.asfortranarray(A.toarray().astype(np.float32)).ravel(order="F")
    ro = O.solve_matop_cones(O.param(max_iter=iters[-1] + 2, eps_acc=1e-300), c, Ad, b, seg_t, seg_l, snap_iters=list(iters),
                             trace_cap=64)
    p = T.SolverParam()
    p.eps_acc = 0.0
    fs = T.FusedSolver(n, m, A, b, c, seg_t, seg_l, p, schedule)
    assert fs.schedule_in_use() == schedule
    N = n + 2 * m + 1
    done = 0
    for q, (it, tol) in enumerate(zip(iters, tols)):
        fs.run(it + 1 - done, poll_every=64)
        done = it + 1
        x, y = fs.iterate()
        rx, ry = ro.snaps[q][:N], ro.snaps[q][N:]
        assert np.abs(x - rx).max() <= tol * max(np.abs(rx).max(), 1e-6), (it, np.abs(x - rx).max(), np.abs(rx).max())
        assert np.abs(y - ry).max() <= tol * max(np.abs(ry).max(), 1e-6), (it, np.abs(y - ry).max(), np.abs(ry).max())
    fs.destroy()


@pytest.mark.parametrize("schedule", ["sweep", "carried"])
def test_sparse_lp_workload_iterates_vs_oracle(T, schedule):
    # bench.py --workload sparse-lp at a size the oracle holds dense: the l1reg_lp construction (examples/l1reg_lp/src/main.rs:50-116)
    # with l = 1500 samples -- m = 6000 (two row blocks), n = 4501 (two column blocks), the 3000 x 1500 kernel block dense --
    # iterates 0, 1, 2, 9 against the f64 oracle on the dense-ified matrix
    c, G, h = l1reg_lp(1500, seed=3)
    A = sp.csc_matrix(G.astype(np.float32))
    assert A.nnz < 0.2 * G.size
    _iterates_vs_oracle(T, A, h.astype(np.float32), c.astype(np.float32), [1], [h.size], [0, 1, 2, 9], [3e-5, 6e-5, 1e-4, 3e-4], schedule)


@pytest.mark.parametrize("schedule", ["sweep", "carried"])
def test_sparse_sdp_workload_iterates_vs_oracle(T, schedule):
    # bench.py --workload sparse-sdp at a size the oracle holds dense: the partitioning_sdp construction
    # (examples/partitioning_sdp/src/main.rs:45-78) on a 6 x 5 grid -- PSD order 30, n = sk = 465, one -1 (or -sqrt 2) per column of
    # the PSD rows and one 1 per equality row -- through ProbSDP's stacking, iterates 0, 1, 2, 9 against the oracle
    from problems import partitioning_sdp
    from test_gpu_solver import _mb
    w, syms_f, mat_a, vec_b = partitioning_sdp(6, 5, seed=2)
    l, n = 30, w.size
    sdp = T.ProbSDP(_mb(T, T.MatType.General(n, 1)).set_array(w.reshape(-1, 1)),
                    [_mb(T, T.MatType.SymPack(l)).set_array(s_) for s_ in syms_f],
                    _mb(T, T.MatType.General(l, n)).set_array(mat_a), _mb(T, T.MatType.General(l, 1)).set_array(vec_b.reshape(-1, 1)),
                    1e-12)
    d = sdp.dense()
    A = sp.csc_matrix(np.asarray(d.mat_a).reshape((d.m, d.n), order="F"))
    assert A.nnz <= d.n + l + 1
    _iterates_vs_oracle(T, A, np.asarray(d.vec_b, np.float32), np.asarray(d.vec_c, np.float32), d.seg_type, d.seg_len,
                        [0, 1, 2, 9], [3e-5, 6e-5, 1e-4, 3e-4], schedule)
    sdp.drop()


def test_toruscompl_socp_example_on_the_sparse_copy(T):
    # examples/toruscompl_socp/src/main.rs:43-268 (the third mostly-zero example matrix SURVEY.md 8f names: 99 % zeros) stacked by ProbSOCP
    # and held sparse: 158 cones of 1 + 2 rows, 317 cones of 1 + 0 rows, 112 zero-cone rows -- through the one-pass recurrence and the
    # carried schedule on the tiled copy, against the f64 oracle at the example's eps_acc
    from problems import toruscompl_socp
    from test_gpu_solver import _mb
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
    d = socp.dense()
    A = sp.csc_matrix(np.asarray(d.mat_a).reshape((d.m, d.n), order="F"))
    assert A.nnz < 0.02 * d.m * d.n
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 1_000_000, 1e-3
    for sched in ("sweep", "carried"):
        fs = T.FusedSolver(d.n, d.m, A, np.asarray(d.vec_b, np.float32), np.asarray(d.vec_c, np.float32), d.seg_type, d.seg_len, p, sched,
                           vec_b_rowabs=d.vec_b_rowabs)
        assert fs.schedule_in_use() == sched
        x, _ = fs.solve(poll_every=64)
        st = fs.status()
        fs.destroy()
        assert st.state == 0 and abs(st.iters - ro.iters) <= 0.05 * ro.iters + 5, (sched, st.iters, ro.iters)
        obj, obj_r = float(q["vec_f"] @ x.astype(np.float64)), float(q["vec_f"] @ ro.x)
        assert abs(obj - obj_r) <= 1e-3 * (1 + abs(obj_r)), (sched, obj, obj_r)
    socp.drop()


def test_sparse_sweep_converges_to_the_dense_answer_multi_tile(T):
    # a sparse LP spanning 3 x 2 tiles (benchmark_lp's construction with 97 % of the random block dropped): the one-pass recurrence
    # on the tiled copy stops within a few iterations of the dense one-pass / carried solve, at the same objective
    rng = np.random.default_rng(5)
    nn = 4200
    R = sp.random(2 * nn, nn, density=0.03, format="csc", random_state=rng, dtype=np.float64)
    R.data = rng.uniform(0, 1, R.nnz)
    G = sp.vstack([-sp.identity(nn), R]).tocsc().astype(np.float32)
    c = -rng.uniform(0, 1, nn).astype(np.float32)
    h = np.concatenate([np.zeros(nn), rng.uniform(0, 1, 2 * nn)]).astype(np.float32)
    m, n = G.shape
    p = T.SolverParam()
    p.eps_acc = 1e-3
    dense = T.FusedSolver(n, m, np.asfortranarray(G.toarray()).ravel(order="F"), h, c, [1], [m], p, "carried")
    xd, _ = dense.solve()
    it_d = dense.status().iters
    dense.destroy()
    for sched in ("sweep", "carried"):
        fs = T.FusedSolver(n, m, G, h, c, [1], [m], p, sched)
        xs, _ = fs.solve()
        assert abs(fs.status().iters - it_d) <= max(5, 0.02 * it_d), (sched, fs.status().iters, it_d)
        od, os_ = float(c.astype(np.float64) @ xd), float(c.astype(np.float64) @ xs)
        assert abs(od - os_) <= 1e-3 * (1 + abs(od))
        passes, bpp = fs.passes()
        assert passes == (2 if sched == "sweep" else 4) and bpp >= 8 * G.nnz
        fs.destroy()


def test_sptile_edge_cases_and_errors_through_the_c_abi(T):
    # what the boundary must accept (operator.rs: zero-sized operators, a matrix without entries, entries only in the padding of a
    # quad) and what it must refuse loudly (row index out of range, column pointers that do not span nnz)
    import ctypes as C
    from totsu_amd import _lib
    from totsu_amd._lib import ThipError, E_INVALID, lib
    from totsu_amd.sparse import SpTile, SparseMatOp
    L = T.F32HIP
    # no entries at all: y = beta y, absadd adds nothing
    z = SparseMatOp(L, sp.csc_matrix((5, 3), dtype=np.float32))
    x, y = _sl(L, np.ones(3)), _sl(L, np.arange(5.0))
    z.op(2.0, x, 0.5, y)
    assert np.allclose(y.get_ref(), 0.5 * np.arange(5.0))
    t = _sl(L, np.full(3, 7.0))
    z.absadd_cols(t)
    assert np.allclose(t.get_ref(), 7.0)
    z.drop()
    # zero rows / zero columns (matop.rs:83-85: the product of an empty operator is beta * y)
    for shape in ((0, 4), (4, 0)):
        e = SparseMatOp(L, sp.csc_matrix(shape, dtype=np.float32))
        xs, ys = _sl(L, np.ones(shape[1])), _sl(L, np.full(shape[0], 3.0))
        e.op(1.0, xs, 2.0, ys)
        assert np.allclose(ys.get_ref(), 6.0)
        xt, yt = _sl(L, np.ones(shape[0])), _sl(L, np.full(shape[1], 3.0))
        e.trans_op(1.0, xt, -1.0, yt)
        assert np.allclose(yt.get_ref(), -3.0)
        e.drop()
    # one entry in the last row and column of a matrix that spans three tiles each way (quads of 1 + 3 padding entries)
    big = sp.csc_matrix(([2.5], ([9999], [8500])), shape=(10000, 8501), dtype=np.float32)
    o = SparseMatOp(L, big)
    assert o.t.info()["tiles"] == 1 and o.t.info()["nnz_stored"] == 4
    x, y = _sl(L, np.arange(8501.0)), _sl(L, np.zeros(10000))
    o.op(1.0, x, 0.0, y)
    yy = y.get_ref()
    assert yy[9999] == 2.5 * 8500 and not yy[:9999].any()
    o.drop()
    # a non-finite entry in the in-vector is answered with NaN (the fixed-point accumulators cannot carry it: it must not vanish),
    # and magnitudes far from 1 are as good as near it (the fixed-point scale follows the in-vector's maximum)
    rng = np.random.default_rng(3)
    mat = sp.random(300, 200, density=0.05, format="csc", random_state=rng, dtype=np.float64)
    mat.data = rng.standard_normal(mat.nnz)
    o = SparseMatOp(L, mat)
    bad = np.ones(200)
    bad[17] = np.nan
    y = _sl(L, np.zeros(300))
    o.op(1.0, _sl(L, bad), 0.0, y)
    assert np.isnan(y.get_ref()).all()
    for scale in (1e-30, 1e25):
        xv = (rng.standard_normal(200) * scale).astype(np.float32)
        y = _sl(L, np.zeros(300))
        o.op(1.0, _sl(L, xv), 0.0, y)
        d = mat.toarray()
        assert np.all(np.abs(y.get_ref() - d @ xv.astype(np.float64)) <= 1e-5 * (np.abs(d) @ np.abs(xv.astype(np.float64))) + 1e-44)
    o.drop()
    # refused: a row index beyond n_row, column pointers that do not end at nnz
    cp = np.array([0, 1, 2], np.int64)
    ri = np.array([0, 5], np.int32)
    va = np.array([1.0, 2.0], np.float32)
    h = C.c_void_p()
    with pytest.raises(ThipError) as ei:
        lib.thip_sptile_create(3, 2, 2, cp.ctypes.data, ri.ctypes.data, va.ctypes.data, C.byref(h))
    assert ei.value.code == E_INVALID
    cp2 = np.array([0, 1, 1], np.int64)
    with pytest.raises(ThipError) as ei:
        lib.thip_sptile_create(3, 2, 2, cp2.ctypes.data, ri.ctypes.data, va.ctypes.data, C.byref(h))
    assert ei.value.code == E_INVALID
    # a sparse operator whose shape is not the problem's is refused by the solver
    st = SpTile(sp.csc_matrix(np.eye(4, dtype=np.float32)))
    with pytest.raises(AssertionError):                      # (the Python layer's own check)
        T.FusedSolver(3, 4, st, np.zeros(4, np.float32), np.zeros(3, np.float32), [1], [4])
    db, dc = T.DeviceBuffer(4, zero=True), T.DeviceBuffer(3, zero=True)
    seg_t, seg_l = (C.c_int32 * 1)(1), (C.c_int64 * 1)(4)
    prob = _lib.Problem(3, 4, None, db.ptr, dc.ptr, None, 1, C.cast(seg_t, C.POINTER(C.c_int32)), C.cast(seg_l, C.POINTER(C.c_int64)))
    par = T.fused._c_param(T.SolverParam())
    hs = C.c_void_p()
    lib.thip_solver_create(C.byref(prob), C.byref(par), _lib.SCHED_SWEEP, C.byref(hs))
    with pytest.raises(ThipError) as ei:                     # (and the library's)
        lib.thip_solver_set_sptile(hs, st.h)
    assert ei.value.code == E_INVALID
    lib.thip_solver_destroy(hs)
    st.free()


def test_tiled_sparse_products_are_bitwise_reproducible(T):
    # the LDS accumulators of the tiled products are 64-bit fixed-point words fed by INTEGER adds: whatever order the waves reach
    # them in, the sums are the same -- two operators built from the same matrix, and two runs of the loop, agree bit for bit
    from totsu_amd.sparse import SparseMatOp
    L = T.F32HIP
    rng = np.random.default_rng(11)
    a = sp.random(9000, 6000, density=0.01, format="csc", random_state=rng, dtype=np.float64)
    a.data = rng.standard_normal(a.nnz) * np.exp(rng.uniform(-8, 8, a.nnz))          # 7 decades of magnitudes
    x = rng.standard_normal(6000).astype(np.float32)
    yv = rng.standard_normal(9000).astype(np.float32)
    outs = []
    for _ in range(3):
        op = SparseMatOp(L, a)
        sy, sx = _sl(L, np.zeros(9000)), _sl(L, np.zeros(6000))
        op.op(1.0, _sl(L, x), 0.0, sy)
        op.trans_op(1.0, _sl(L, yv), 0.0, sx)
        outs.append((sy.get_ref().copy(), sx.get_ref().copy()))
        op.drop()
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])
    d = a.toarray()
    assert np.all(np.abs(outs[0][0] - d @ x) <= 1e-5 * (np.abs(d) @ np.abs(x)) + 1e-30)
    assert np.all(np.abs(outs[0][1] - d.T @ yv) <= 1e-5 * (np.abs(d.T) @ np.abs(yv)) + 1e-30)
    # the loop: two solvers on the same sparse LP stop at the same iteration with the same bits
    c, G, h = l1reg_lp(60, seed=4)
    A = sp.csc_matrix(G.astype(np.float32))
    p = T.SolverParam()
    p.eps_acc = 1e-3
    res = []
    for _ in range(2):
        fs = T.FusedSolver(c.size, h.size, A, h.astype(np.float32), c.astype(np.float32), [1], [h.size], p, "sweep")
        xs, ys = fs.solve()
        res.append((fs.status().iters, xs, ys))
        fs.destroy()
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


def _with_full_tiles(rng):
    """8242 x 4106: tile (0, 0) full, (1, 0) 30 % random, (2, 0) = the last 50 rows, full but not of full height, (0, 1) = 4096 x 10
    full, (1, 1) empty, (2, 1) one entry"""
    d = np.zeros((8242, 4106))
    d[:4096, :4096] = rng.standard_normal((4096, 4096))
    blk = rng.standard_normal((4096, 4096))
    blk[rng.uniform(size=blk.shape) > 0.3] = 0.0
    d[4096:8192, :4096] = blk
    d[8192:, :4096] = rng.standard_normal((50, 4096))
    d[:4096, 4096:] = rng.standard_normal((4096, 10))
    d[8200, 4100] = 2.5
    return d


def test_full_tiles_are_held_without_indices(T):
    # a tile of full height whose every column holds all 4096 rows stores values only (4 bytes per entry) and runs the dense
    # routines of sp_tile_k; same contract, same tolerances; a caller whose rows do not ascend gets the indexed form and the same answer
    from totsu_amd.sparse import SparseMatOp, SpTile
    L = T.F32HIP
    rng = np.random.default_rng(21)
    d = _with_full_tiles(rng)
    a = sp.csc_matrix(d)
    a.sort_indices()
    x = rng.standard_normal(d.shape[1]).astype(np.float32)
    yv = rng.standard_normal(d.shape[0]).astype(np.float32)

    def products(op):
        sy, sx = _sl(L, yv.copy()), _sl(L, x.copy())
        op.op(0.7, _sl(L, x), -0.3, sy)
        op.trans_op(-1.5, _sl(L, yv), 0.5, sx)
        st, ss = _sl(L, np.zeros(d.shape[1])), _sl(L, np.zeros(d.shape[0]))
        op.absadd_cols(st)
        op.absadd_rows(ss)
        return sy.get_ref().copy(), sx.get_ref().copy(), st.get_ref().copy(), ss.get_ref().copy()

    op = SparseMatOp(L, a)
    info = op.t.info()
    assert info["dense_tiles"] == 2 and info["tiles"] == 5
    assert info["indexed_entries"] == info["nnz_stored"] - 4096 * 4096 - 4096 * 10
    assert info["bytes_per_product"] == 4 * info["nnz_stored"] + 4 * info["indexed_entries"]
    o1 = products(op)
    op.drop()
    op = SparseMatOp(L, a)
    o2 = products(op)
    op.drop()
    for u, v in zip(o1, o2):
        assert np.array_equal(u, v)                                     # bitwise reproducible, as the indexed form
    ref_n = 0.7 * d @ x - 0.3 * yv
    ref_t = -1.5 * d.T @ yv + 0.5 * x
    assert np.all(np.abs(o1[0] - ref_n) <= 1e-5 * (0.7 * np.abs(d) @ np.abs(x) + 0.3 * np.abs(yv) + 1e-6))
    assert np.all(np.abs(o1[1] - ref_t) <= 1e-5 * (1.5 * np.abs(d.T) @ np.abs(yv) + 0.5 * np.abs(x) + 1e-6))
    assert np.allclose(o1[2], np.abs(d).sum(axis=0), rtol=1e-5) and np.allclose(o1[3], np.abs(d).sum(axis=1), rtol=1e-5)
    # rows handed over in DESCENDING order within every column: nothing may be taken for dense
    cp = a.indptr.astype(np.int64)
    ri, va = a.indices.copy(), a.data.astype(np.float32)
    for j in range(d.shape[1]):
        ri[cp[j]:cp[j + 1]] = ri[cp[j]:cp[j + 1]][::-1]
        va[cp[j]:cp[j + 1]] = va[cp[j]:cp[j + 1]][::-1]
    t = SpTile.from_csc_arrays(d.shape[0], d.shape[1], cp, ri.astype(np.int32), va)
    assert t.info()["dense_tiles"] == 0 and t.info()["indexed_entries"] == t.info()["nnz_stored"]
    sy, sx = _sl(L, np.zeros(d.shape[0])), _sl(L, np.zeros(d.shape[1]))
    t.mv(False, 0.7, _sl(L, x), 0.0, sy)
    t.mv(True, -1.5, _sl(L, yv), 0.0, sx)
    assert np.all(np.abs(sy.get_ref() - 0.7 * d @ x) <= 1e-5 * (0.7 * np.abs(d) @ np.abs(x) + 1e-6))
    assert np.all(np.abs(sx.get_ref() + 1.5 * d.T @ yv) <= 1e-5 * (1.5 * np.abs(d.T) @ np.abs(yv) + 1e-6))
    t.free()


def test_sparse_lp_with_full_tiles_iterates_vs_oracle(T):
    # the l1reg_lp construction at l = 4096: the kernel block [K ; -K] is two full tiles (held without indices), the diagonals
    # beside it indexed ones -- iterates 0, 1, 2, 9 of the one-pass loop against the f64 oracle on the dense-ified matrix
    from totsu_amd.sparse import SpTile
    c, G, h = l1reg_lp(4096, seed=5)
    A = sp.csc_matrix(G.astype(np.float32))
    A.sort_indices()
    t = SpTile(A)
    assert t.info()["dense_tiles"] == 2
    t.free()
    _iterates_vs_oracle(T, A, h.astype(np.float32), c.astype(np.float32), [1], [h.size], [0, 1, 2, 9], [3e-5, 6e-5, 1e-4, 3e-4], "sweep")


def test_stencil_lp_with_long_vectors_iterates_vs_sparse_oracle(T):
    # a 5-point Laplacian on an 800 x 800 grid as the inequality matrix of an LP: m = n = 640 000 -- long enough for the wide-block
    # forms of the O(m + n) kernels of the sparse one-pass loop (sw_xm_k<., FLAT>, sp_col_k, sp_absmax_k with 1024 threads) --
    # iterates 0, 1, 2, 9 against the f64 oracle running the same matrix through its sparse user-operator
    g = 800
    e = np.ones(g, np.float32)
    L1 = sp.diags([-e[:-1], 2 * e, -e[:-1]], [-1, 0, 1], format="csc")
    A = (sp.kron(sp.identity(g, dtype=np.float32), L1) + sp.kron(L1, sp.identity(g, dtype=np.float32))).tocsc().astype(np.float32)
    A.sort_indices()
    m, n = A.shape
    rng = np.random.default_rng(8)
    b = rng.uniform(0.5, 1.5, m).astype(np.float32)
    c = rng.standard_normal(n).astype(np.float32)
    iters, tols = [0, 1, 2, 9], [3e-5, 6e-5, 1e-4, 3e-4]
    ro = O.solve_csc_cones(O.param(max_iter=iters[-1] + 2, eps_acc=1e-300), c, A.indptr, A.indices, A.data, b, [1], [m],
                           snap_iters=list(iters), trace_cap=64)
    p = T.SolverParam()
    p.eps_acc = 0.0
    fs = T.FusedSolver(n, m, A, b, c, [1], [m], p, "sweep")
    assert fs.schedule_in_use() == "sweep"
    N = n + 2 * m + 1
    done = 0
    for q, (it, tol) in enumerate(zip(iters, tols)):
        fs.run(it + 1 - done, poll_every=64)
        done = it + 1
        x, y = fs.iterate()
        rx, ry = ro.snaps[q][:N], ro.snaps[q][N:]
        assert np.abs(x - rx).max() <= tol * max(np.abs(rx).max(), 1e-6), (it, np.abs(x - rx).max(), np.abs(rx).max())
        assert np.abs(y - ry).max() <= tol * max(np.abs(ry).max(), 1e-6), (it, np.abs(y - ry).max(), np.abs(ry).max())
    fs.destroy()


class _DiffOp:
    """A user-defined matrix-free Operator built only from LinAlg primitives, in the pattern of
    examples/imgnr_udef/src/prob_op_a.rs: the (n-1) x n forward-difference matrix D (D x)_i = x_{i+1} - x_i, never
    materialised.  op / trans_op / absadd_* are hand-written with split + add + scale."""

    def __init__(self, L, n):
        self.L, self.n = L, n

    def size(self):
        return (self.n - 1, self.n)

    def op(self, alpha, x, beta, y):                      # y = alpha (x[1:] - x[:-1]) + beta y
        L = self.L
        _, hi = x.split(1)
        lo, _ = x.split(self.n - 1)
        L.scale(beta, y)
        L.add(alpha, hi, y)
        L.add(-alpha, lo, y)

    def trans_op(self, alpha, x, beta, y):                # y = alpha D^T x + beta y: y[1:] += x, y[:-1] -= x
        L = self.L
        L.scale(beta, y)
        _, hi = y.split(1)
        lo, _ = y.split(self.n - 1)
        L.add(alpha, x, hi)
        L.add(-alpha, x, lo)

    def absadd_cols(self, tau):                           # column abs sums of D: 1, 2, ..., 2, 1
        L = self.L
        _, hi = tau.split(1)
        lo, _ = tau.split(self.n - 1)
        L.adds(1.0, hi)
        L.adds(1.0, lo)

    def absadd_rows(self, sigma):                         # every row has |-1| + |1|
        self.L.adds(2.0, sigma)


def test_user_defined_matrix_free_operator(T):
    # adjointness / absadd checks against the dense reference (examples/utils2/src/operator_ref.rs:5-69 pattern), then
    # a solve: total-variation-like LP  min 1^T t  s.t.  -t <= D z - d <= t  written with the custom operator inside
    L = T.F32HIP
    n = 40
    D = np.zeros((n - 1, n))
    D[np.arange(n - 1), np.arange(n - 1)] = -1.0
    D[np.arange(n - 1), np.arange(1, n)] = 1.0
    op = _DiffOp(L, n)
    rng = np.random.default_rng(2)
    x = rng.standard_normal(n).astype(np.float32)
    y = rng.standard_normal(n - 1).astype(np.float32)
    sx, sy = _sl(L, x), _sl(L, y.copy())
    op.op(0.5, sx, -2.0, sy)
    assert np.allclose(sy.get_ref(), 0.5 * D @ x - 2.0 * y, atol=1e-5)
    sy2, sx2 = _sl(L, y), _sl(L, x.copy())
    op.trans_op(1.5, sy2, 0.25, sx2)
    assert np.allclose(sx2.get_ref(), 1.5 * D.T @ y + 0.25 * x, atol=1e-5)
    t, s = _sl(L, np.zeros(n)), _sl(L, np.zeros(n - 1))
    op.absadd_cols(t)
    op.absadd_rows(s)
    assert np.allclose(t.get_ref(), np.abs(D).sum(axis=0)) and np.allclose(s.get_ref(), np.abs(D).sum(axis=1))

    # an LP whose A stacks the matrix-free D with identity blocks:  variables (z in R^n, t in R^{n-1})
    #   min 1^T t   s.t.   D z - t <= d,  -D z - t <= -d,  z_0 = 0 handled by the objective's null space being harmless
    d = np.sign(np.sin(np.arange(n - 1) / 3.0)).astype(np.float64)

    class _OpA:
        def __init__(self):
            self.n, self.k = n, n - 1

        def size(self):
            return (2 * self.k, self.n + self.k)

        def op(self, alpha, xv, beta, yv):
            z, tt = xv.split(self.n)
            y1, y2 = yv.split(self.k)
            op.op(alpha, z, beta, y1)
            L.add(-alpha, tt, y1)
            op.op(-alpha, z, beta, y2)
            L.add(-alpha, tt, y2)

        def trans_op(self, alpha, xv, beta, yv):
            x1, x2 = xv.split(self.k)
            z, tt = yv.split(self.n)
            op.trans_op(alpha, x1, beta, z)
            op.trans_op(-alpha, x2, 1.0, z)
            L.scale(beta, tt)
            L.add(-alpha, x1, tt)
            L.add(-alpha, x2, tt)

        def absadd_cols(self, tau):
            z, tt = tau.split(self.n)
            op.absadd_cols(z)
            op.absadd_cols(z)
            L.adds(2.0, tt)

        def absadd_rows(self, sigma):
            s1, s2 = sigma.split(self.k)
            op.absadd_rows(s1)
            L.adds(1.0, s1)
            op.absadd_rows(s2)
            L.adds(1.0, s2)

    k = n - 1
    A = np.block([[D, -np.eye(k)], [-D, -np.eye(k)]])
    c = np.concatenate([np.zeros(n), np.ones(k)])
    b = np.concatenate([d, -d])
    ro = O.solve_matop_cones(O.param(max_iter=400000, eps_acc=1e-4), c, np.asfortranarray(A).ravel(order="F"), b,
                             [O.CONE_RPOS], [2 * k])
    assert ro.status == O.OK
    op_c = T.MatOp(L, T.MatType.General(n + k, 1), c.astype(np.float32))
    op_b = T.MatOp(L, T.MatType.General(2 * k, 1), b.astype(np.float32))
    s_ = T.Solver(L)
    s_.param.eps_acc, s_.param.max_iter = 1e-3, 400_000
    work = np.zeros(T.Solver.query_worklen((2 * k, n + k)), dtype=np.float32)
    xs, ys = s_.solve((op_c, _OpA(), op_b, T.ConeRPos(L), work))
    assert abs(float(c @ xs.astype(np.float64)) - float(c @ ro.x)) <= 5e-3 * (1 + abs(float(c @ ro.x)))
