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


@pytest.mark.parametrize("shape,density", [((1, 1), 1.0), ((50, 30), 0.1), ((300, 1000), 0.01), ((5000, 4000), 0.002),
                                           ((64, 64), 0.9), ((7, 2000), 0.5), ((2000, 3), 0.4)])
def test_sparse_operator_contract(T, shape, density):
    from totsu_amd.sparse import SparseMatOp
    L = T.F32HIP
    rng = np.random.default_rng(shape[0] + shape[1])
    a = sp.random(shape[0], shape[1], density=density, format="csr", random_state=rng, dtype=np.float64)
    a.data = rng.standard_normal(a.nnz)
    op = SparseMatOp(L, a)
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


@pytest.mark.parametrize("schedule", ["reference", "fused", "carried"])
def test_fused_loop_on_csr_matches_dense(T, schedule):
    # the device-resident loop with A given as scipy.sparse: same iterates as with the dense matrix
    c, G, h = l1reg_lp(20, seed=1)
    n, m = c.size, h.size
    p = T.SolverParam()
    p.eps_acc = 1e-3
    dense = T.FusedSolver(n, m, np.asfortranarray(G).ravel(order="F").astype(np.float32), h, c, [1], [m], p, schedule)
    xd, yd = dense.solve()
    sparse = T.FusedSolver(n, m, sp.csr_matrix(G.astype(np.float32)), h, c, [1], [m], p, schedule)
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


def test_fused_loop_sparse_socp_iterates_vs_oracle(T):
    # sparse SOCP blocks (90 % zeros): iterates of the CSR fused loop against the dense f64 oracle
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
    fs = T.FusedSolver(n, m, sp.csr_matrix(A), b, f, seg_t, seg_l, p, "carried")
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
