"""Pins the CPU oracle (oracle/) to every golden vector / known-answer test the reference holds
for the hot path.  CPU only.  Citations relative to /root/reference/."""
import json
import os

import numpy as np
import pytest

import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def _rust_exp(v):
    """format like Rust's {:.2e} (no zero padding / sign in the exponent)"""
    s = "%.2e" % v
    mant, ex = s.split("e")
    return "%se%d" % (mant, int(ex))


def test_log_qemu_golden_trace():
    # examples/nostd_cortex-m/log_qemu.txt:1-26 -- every logged residual triple, the iteration
    # count and the solution to all 16 printed digits
    g = json.load(open(os.path.join(HERE, "golden", "log_qemu.json")))
    pb = g["problem"]
    assert O.lib().oc_query_worklen(3, 2) == g["query_worklen"] == 48
    par = O.param(max_iter=pb["max_iter"], log_period=pb["log_period"])
    r = O.solve_matop_cones(par, pb["vec_c"], pb["mat_a_colmajor"], pb["vec_b"], [O.CONE_RPOS], [pb["m"]],
                            trace_cap=4096)
    assert r.status == O.OK
    assert r.iters == g["trace"][-1]["iter"] == 159
    by_iter = {t[0]: t for t in r.trace}
    for rec in g["trace"]:
        t = by_iter[rec["iter"]]
        assert t[1] == 0
        assert [_rust_exp(v) for v in t[2:5]] == rec["text"], rec
    assert [repr(float(v)) for v in r.x] == g["x_text"]


def test_core_solver_psd_kat():
    # totsu_core/tests/solver.rs:14-53 (and totsu_f64lapack/tests/solver.rs:17-55 for the QL flavour)
    for use_ql in (False, True):
        par = O.param(max_iter=100000)
        r = O.solve_matop_cones(par, [1.0], [0.0, -1.0 * 1.41421356, -3.0], [1.0, 0.0 * 1.41421356, 10.0],
                                [O.CONE_PSD], [3], use_ql=use_ql)
        assert r.status == O.OK
        assert abs(r.x[0] - (-2.0)) <= 1e-3


def test_lp1_infeasible():
    # totsu/tests/lp.rs:13-45
    par = O.param(max_iter=100000)
    r = O.solve_lp(par, [1.0], np.array([[1.0], [-1.0]]), [-5.0, -10.0], np.zeros((0, 1)), [])
    assert r.status == O.INFEASIBLE


def test_lp2_unbounded():
    # totsu/tests/lp.rs:50-82
    par = O.param(max_iter=100000)
    r = O.solve_lp(par, [1.0], np.array([[1.0], [1.0]]), [5.0, 10.0], np.zeros((0, 1)), [])
    assert r.status == O.UNBOUNDED


def test_socp1():
    # totsu/tests/socp.rs:14-46 (default params: max_iter None)
    par = O.param()
    G = np.eye(2)
    r = O.solve_socp(par, [1.0, 1.0], [G], [[0.0, 0.0]], [[0.0, 0.0]], [np.sqrt(2.0)], np.zeros((0, 2)), [])
    assert r.status == O.OK
    assert np.allclose(r.x, [-1.0, -1.0], atol=1e-3)


def test_socp2_zero_row_cone():
    # totsu/tests/socp.rs:52-93 -- includes a General(0, n) block
    par = O.param(max_iter=100000)
    r = O.solve_socp(par, [0.0, 1.0],
                     [np.zeros((0, 2)), np.array([[-1.0, 0.0]])],
                     [[], [2.0]],
                     [[0.0, -1.0], [0.0, 1.0]],
                     [50.0, 0.0], np.zeros((0, 2)), [])
    assert r.status == O.OK
    assert np.allclose(r.x, [2.0, 0.0], atol=1e-3)


def test_sdp1():
    # totsu/tests/sdp.rs:14-50 ; SymPack set_iter_rowmaj of a symmetric 2x2 -> packed [a00, a01, a11]
    for use_ql in (False, True):
        par = O.param(max_iter=100000)
        syms = [[-1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [3.0, 0.0, 4.0]]
        r = O.solve_sdp(par, [1.0, 1.0], syms, np.zeros((0, 2)), [], par.eps_zero, use_ql=use_ql)
        assert r.status == O.OK
        assert np.allclose(r.x, [3.0, 4.0], atol=1e-3)


def test_cone_psd_unit_kat():
    # totsu_core/src/cone_psd.rs:90-110
    for use_ql in (False, True):
        x = O.proj(O.CONE_PSD, [5.0, 0.0, -5.0], eps_zero=1e-12, use_ql=use_ql)
        assert np.allclose(x, [5.0, 0.0, 0.0], atol=1e-6)


def test_matop_sympack_kat():
    # totsu_core/src/matop.rs:180-212
    array = [1., 2., 3., 4., 5., 6., 7., 8., 9., 10., 11., 12., 13., 14., 15.]
    ref = np.array([[1., 2., 4., 7., 11.], [2., 3., 5., 8., 12.], [4., 5., 6., 9., 13.],
                    [7., 8., 9., 10., 14.], [11., 12., 13., 14., 15.]])
    for i in range(5):
        x = np.zeros(5)
        x[i] = 1.0
        y = O.transform_sp(5, 1.0, array, x, 0.0, np.zeros(5))
        assert np.allclose(y, ref[i], atol=1e-3)


def test_vec_to_mat_kat():
    # totsu_f64lapack/src/f64lapack.rs:262-287
    ref_v = np.array([1. * 0.7, 2., 3. * 0.7, 4., 5., 6. * 0.7, 7., 8., 9., 10. * 0.7,
                      11., 12., 13., 14., 15. * 0.7])
    ref_m = np.array([1., 0, 0, 0, 0, 2., 3., 0, 0, 0, 4., 5., 6., 0, 0, 7., 8., 9., 10., 0,
                      11., 12., 13., 14., 15.])
    m = O.vec_to_mat(ref_v, scale=np.sqrt(2.0))
    assert np.all(np.abs(m - ref_m) <= 0.5)
    v, _ = O.mat_to_vec(m, scale=np.sqrt(2.0))
    assert np.allclose(v, ref_v, atol=1e-6)


def test_scale_nondiag_kat():
    # totsu/src/matbuild/mod.rs:305-333
    packed = [1., 2., 3., 4., 5., 6., 7., 8., 9., 10., 11., 12., 13., 14., 15.]
    ref = [1., 2. * 1.4, 3., 4. * 1.4, 5. * 1.4, 6., 7. * 1.4, 8. * 1.4, 9. * 1.4, 10.,
           11. * 1.4, 12. * 1.4, 13. * 1.4, 14. * 1.4, 15.]
    assert np.allclose(O.scale_nondiag_sympack(packed, 1.4), ref, atol=1e-3)


def test_abssum_striding():
    # floatgeneric.rs:62-74 / f64lapack.rs:51-59: chunks(incx) semantics incl. ragged tail and incx 0
    x = np.array([1., -2., 3., -4., 5., -6., 7.])
    assert O.abssum(x, 1) == 28.0
    assert O.abssum(x, 2) == 1 + 3 + 5 + 7
    assert O.abssum(x, 3) == 1 + 4 + 7
    assert O.abssum(x, 7) == 1
    assert O.abssum(x, 100) == 1
    assert O.abssum(x, 0) == 0.0
    assert O.abssum(np.zeros(0), 1) == 0.0


def test_transform_ge_matches_numpy():
    rng = np.random.default_rng(0)
    for (nr, nc) in [(1, 1), (3, 2), (7, 13), (64, 65), (300, 17)]:
        a = rng.standard_normal((nr, nc))
        for tr in (False, True):
            x = rng.standard_normal(nr if tr else nc)
            y0 = rng.standard_normal(nc if tr else nr)
            y = O.transform_ge(tr, nr, nc, 0.7, a.ravel(order="F"), x, -0.3, y0)
            ref = 0.7 * ((a.T if tr else a) @ x) - 0.3 * y0
            assert np.allclose(y, ref, rtol=1e-12, atol=1e-12)


def test_map_eig_jacobi_vs_ql_vs_numpy():
    rng = np.random.default_rng(1)
    for k in (1, 2, 3, 8, 25):
        b = rng.standard_normal((k, k))
        s = (b + b.T) / 2
        iu = [(r, c) for c in range(k) for r in range(c + 1)]
        packed = np.array([s[r, c] * (np.sqrt(2.0) if r != c else 1.0) for (r, c) in iu])
        w, v = np.linalg.eigh(s)
        ref = (v * np.maximum(w, 0)) @ v.T
        ref_p = np.array([ref[r, c] * (np.sqrt(2.0) if r != c else 1.0) for (r, c) in iu])
        pj = O.proj(O.CONE_PSD, packed, use_ql=False)
        pq = O.proj(O.CONE_PSD, packed, use_ql=True)
        assert np.allclose(pj, ref_p, atol=1e-9)
        assert np.allclose(pq, ref_p, atol=1e-9)


def test_soc_rotsoc_branches():
    # cone_soc.rs:38-65: the three branches + empty + length 1
    assert O.proj(O.CONE_SOC, []).size == 0
    assert np.array_equal(O.proj(O.CONE_SOC, [-1.0]), [0.0])            # |v|=0 <= 1 -> zero
    assert np.array_equal(O.proj(O.CONE_SOC, [2.0]), [2.0])
    assert np.array_equal(O.proj(O.CONE_SOC, [-5.0, 3.0, 4.0]), [0.0, 0.0, 0.0])
    assert np.array_equal(O.proj(O.CONE_SOC, [5.0, 3.0, 4.0]), [5.0, 3.0, 4.0])
    y = O.proj(O.CONE_SOC, [1.0, 3.0, 4.0])
    assert np.allclose(y, [3.0, 3.0 * 0.6, 4.0 * 0.6])
    # cone_rotsoc.rs:38-65
    assert np.array_equal(O.proj(O.CONE_ROTSOC, [-3.0]), [0.0])
    z = O.proj(O.CONE_ROTSOC, [1.0, 1.0, 0.5])          # 2rs >= |v|^2 : inside
    assert np.allclose(z, [1.0, 1.0, 0.5])


def test_rng_is_deterministic_and_sane():
    u = np.array([O.rng_uniform(0, 1, i) for i in range(4000)])
    g = np.array([O.rng_normal(0, 2, i) for i in range(4000)])
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.03
    assert abs(g.mean()) < 0.06 and abs(g.std() - 1.0) < 0.05
    assert O.rng_uniform(0, 1, 17) == O.rng_uniform(0, 1, 17)
    assert O.rng_uniform(0, 1, 17) != O.rng_uniform(1, 1, 17)


def test_qp1_kat():
    # totsu/tests/qp.rs:14-48
    par = O.param(max_iter=100000)
    r = O.solve_qp(par, [1.0, 0.0, 1.0], [1.0, 2.0], np.array([[-0.5, -1.0 / 3.0]]), [-1.0], np.zeros((0, 2)), [])
    assert r.status == O.OK
    assert np.allclose(r.x[:2], [2.0, 0.0], atol=1e-3)


def test_qcqp1_kat():
    # totsu/tests/qcqp.rs:14-47
    par = O.param(max_iter=100000)
    r = O.solve_qcqp(par, [[1.0, 0.0, 1.0], [0.0, 0.0, 0.0]], [[-5.0, -4.0], [-0.5, -1.0 / 3.0]], [0.0, 1.0],
                     np.zeros((0, 2)), [])
    assert r.status == O.OK
    assert np.allclose(r.x[:2], [5.0, 4.0], atol=1e-3)


def test_partitioning_sdp_construction_on_oracle():
    # examples/partitioning_sdp/src/main.rs:45-79 at 2 x 3 nodes: the relaxation's optimum has unit diagonal, is PSD,
    # and its value is a lower bound of every cut sampled from it (main.rs:82-140)
    from problems import partitioning_sdp
    w, syms_f, mat_a, vec_b = partitioning_sdp(2, 3, seed=1)
    par = O.param(max_iter=200000, eps_acc=1e-6)
    for use_ql in (False, True):
        r = O.solve_sdp(par, w, syms_f, mat_a, vec_b, 1e-12, use_ql=use_ql)
        assert r.status == O.OK
        l = 6
        X = np.zeros((l, l))
        kk = 0
        for j in range(l):
            for i in range(j + 1):
                X[i, j] = X[j, i] = r.x[kk]
                kk += 1
        assert np.allclose(np.diag(X), 1.0, atol=1e-4)
        assert np.linalg.eigvalsh(X).min() >= -1e-4
        Wm = np.zeros((l, l))
        kk = 0
        for j in range(l):
            for i in range(j + 1):
                Wm[i, j] = Wm[j, i] = w[kk] / (1.0 if i == j else 2.0)   # sum_k w_k X_k counts each edge once
                kk += 1
        relax = float(w @ r.x)
        best = min(float(s @ Wm @ s) for s in (np.array([1 if (b >> t) & 1 else -1 for t in range(l)]) for b in range(64)))
        assert relax <= best + 1e-3


def test_svm_qp_construction_on_oracle():
    # examples/svm_qp/src/main.rs:47-106 (ProbQP with a SymPack kernel matrix, one equality row): KKT-level checks
    from problems import svm_qp, sym_pack
    q = svm_qp(50, seed=0)
    r = O.solve_qp(O.param(max_iter=2_000_000, eps_acc=1e-4), sym_pack(q["sym_p"]), q["vec_q"], q["mat_g"], q["vec_h"],
                   q["mat_a"], q["vec_b"])
    assert r.status == O.OK
    a = r.x[:50]
    assert a.min() > -1e-3 * a.max() and abs(float(q["mat_a"][0] @ a)) < 1e-3 * np.abs(a).sum()
    # the support vectors (a_i > 0) sit on the margin: y_i (w.x_i + bias) = 1 for a common bias (main.rs:112-121)
    y = q["mat_a"][0]
    f = (q["sym_p"] * y[None, :] * y[:, None]) @ (a * y)         # sum_j a_j y_j k(x_i, x_j)
    sv = a > 1e-2 * a.max()
    bias = (y[sv] - f[sv]).mean()
    assert np.abs(y[sv] * (f[sv] + bias) - 1.0).max() < 5e-2


def test_trajplan_qcqp_construction_on_oracle():
    # examples/trajplan_qcqp/src/main.rs:19-151 (ProbQCQP: 28 rotated-cone constraints + 12 equality rows)
    from problems import sym_pack, trajplan_qcqp
    t_cap, a_cap = 30, 90.0
    c = trajplan_qcqp(t_cap, a_cap)
    r = O.solve_qcqp(O.param(max_iter=2_000_000, eps_acc=1e-3), [sym_pack(s) for s in c["syms_p"]], c["vecs_q"],
                     c["scls_r"], c["mat_a"], c["vec_b"])
    assert r.status == O.OK
    x = r.x[:2 * t_cap]
    assert np.abs(c["mat_a"] @ x - c["vec_b"]).max() < 2e-3
    px, py = x[:t_cap], x[t_cap:]
    acc = np.hypot(np.diff(px, 2), np.diff(py, 2)) * t_cap * t_cap
    assert acc.max() <= a_cap * 1.01
    assert acc.max() >= a_cap * 0.9                       # the bound is active somewhere at a_cap = 90 (the example's point)


def test_toruscompl_socp_construction_on_oracle():
    # examples/toruscompl_socp/src/main.rs:43-268 at the example's size (9 x 7 nodes): 158 three-row cones, 317 cones
    # WITHOUT rows (bounds and the volume constraint as one-dimensional cones), 112 equality rows
    from problems import toruscompl_socp
    q = toruscompl_socp(9, 7, 0.2)
    l = len(q["members"])
    assert (l, q["vec_b"].size) == (158, 112)
    r = O.solve_socp(O.param(max_iter=1_000_000, eps_acc=1e-3), q["vec_f"], q["mats_g"], q["vecs_h"], q["vecs_c"],
                     q["scls_d"], q["mat_a"], q["vec_b"])
    assert r.status == O.OK
    x, qf, w = r.x[:l], r.x[l:2 * l], r.x[2 * l:3 * l]
    assert x.min() > -1e-3 and x.max() < 1 + 1e-3
    assert abs(float(q["length"] @ x) / (0.2 * q["length"].sum()) - 1.0) < 5e-3          # the volume bound is active
    assert np.abs(q["mat_a"] @ r.x - q["vec_b"]).max() < 1e-2                               # force balance at every free node
    assert np.all(2.0 * w * x + 1e-2 >= q["length"] * qf * qf)                               # w_i x_i >= v_i q_i^2 / 2


def test_sparse_operator_oracle_is_the_dense_oracle():
    # the CSC user-operator of the oracle (the CPU baseline and checker of the sparse workloads; pattern:
    # examples/imgnr_udef/src/prob_op_a.rs:33-120) against the MatOp path that the golden vectors pin: same iterates, same trace,
    # same preconditioner on the dense-ified matrix -- an LP (l1reg_lp construction) and an SOCP with sparse blocks
    import scipy.sparse as sp
    from problems import l1reg_lp, random_socp
    c, G, h = l1reg_lp(40, seed=2)
    A = sp.csc_matrix(G)
    par = O.param(max_iter=60, eps_acc=1e-300)
    rd = O.solve_matop_cones(par, c, G, h, [O.CONE_RPOS], [h.size], snap_iters=[0, 9, 59], trace_cap=64)
    rs = O.solve_csc_cones(par, c, A.indptr, A.indices, A.data, h, [O.CONE_RPOS], [h.size], snap_iters=[0, 9, 59], trace_cap=64)
    for a, b in zip(rd.snaps, rs.snaps):
        assert np.allclose(a, b, rtol=1e-12, atol=1e-13)
    n, cones = 30, [5, 12, 0, 7]
    f, Gs, hs, cs, d = random_socp(n, cones, seed=4)
    rng = np.random.default_rng(1)
    Gs = [g * (rng.uniform(0, 1, g.shape) < 0.2) for g in Gs]
    Ad = np.vstack([np.vstack([-c_.reshape(1, n), -g]) for g, c_ in zip(Gs, cs)]).astype(np.float64)
    b = np.concatenate([np.concatenate([[dd], h_]) for dd, h_ in zip(d, hs)]).astype(np.float64)
    seg_t, seg_l = [O.CONE_SOC] * len(cones), [1 + k for k in cones]
    As = sp.csc_matrix(Ad)
    rd = O.solve_matop_cones(O.param(eps_acc=1e-6, max_iter=200000), f, Ad, b, seg_t, seg_l)
    rs = O.solve_csc_cones(O.param(eps_acc=1e-6, max_iter=200000), f, As.indptr, As.indices, As.data, b, seg_t, seg_l)
    # (dropping entries of G after f was formed loses the construction's boundedness: both paths must say so at the same iteration)
    assert rd.status == rs.status and abs(rd.iters - rs.iters) <= 1
    assert np.allclose(rd.x, rs.x, rtol=1e-9, atol=1e-11)
    rd = O.solve_matop_cones(O.param(eps_acc=1e-5, max_iter=400000), c, G, h, [O.CONE_RPOS], [h.size])
    rs = O.solve_csc_cones(O.param(eps_acc=1e-5, max_iter=400000), c, A.indptr, A.indices, A.data, h, [O.CONE_RPOS], [h.size])
    assert rd.status == rs.status == O.OK and abs(rd.iters - rs.iters) <= 1
    assert np.allclose(rd.x, rs.x, rtol=1e-9, atol=1e-11)
