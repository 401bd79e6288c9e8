"""Problem builders ProbLP / ProbSOCP / ProbSDP: compose Operators and Cones out of MatOps and the core cones.
Mirror of totsu/src/problem/{lp,socp,sdp}.rs (same struct and method names).  `problem()` returns the tuple
`Solver.solve` takes; `dense()` (this repo's addition) returns the stacked dense description the fused
device loop takes (totsu_amd.fused), built from the very same MatBuild arrays."""
import math

import numpy as np

from . import _lib
from .cone import ConePSD, ConeRPos, ConeSOC, ConeZero
from .linalg import splitm
from .matbuild import MatBuild
from .matop import MatType
from .solver import Solver


class _OpVec:
    """ProbLPOpC / ProbSOCPOpC / ProbSDPOpC: a column vector as Operator (lp.rs:11-46)"""

    def __init__(self, vec):
        self.vec = vec

    def size(self):
        n, one = self.vec.size()
        assert one == 1
        return (n, 1)

    def op(self, alpha, x, beta, y):
        self.vec.op(alpha, x, beta, y)

    def trans_op(self, alpha, x, beta, y):
        self.vec.trans_op(alpha, x, beta, y)

    def absadd_cols(self, tau):
        self.vec.absadd_cols(tau)

    def absadd_rows(self, sigma):
        self.vec.absadd_rows(sigma)


class _OpStack2:
    """two row-stacked blocks sharing the columns: ProbLPOpA (lp.rs:50-115), ProbLPOpB (lp.rs:119-191),
    ProbSDPOpA (sdp.rs:49-114), ProbSDPOpB (sdp.rs:118-190; sign_first = -1)"""

    def __init__(self, first, second, sign_first=1.0):
        self.first, self.second, self.sf = first, second, sign_first

    def size(self):
        m, n = self.first.size()
        p, n_ = self.second.size()
        assert n == n_
        return (m + p, n)

    def op(self, alpha, x, beta, y):
        m, p = self.first.size()[0], self.second.size()[0]
        y_m, y_p = splitm(y, m, p)
        self.first.op(self.sf * alpha, x, beta, y_m)
        self.second.op(alpha, x, beta, y_p)

    def trans_op(self, alpha, x, beta, y):
        m, p = self.first.size()[0], self.second.size()[0]
        x_m, x_p = splitm(x, m, p)
        self.first.trans_op(self.sf * alpha, x_m, beta, y)
        self.second.trans_op(alpha, x_p, 1.0, y)

    def absadd_cols(self, tau):
        self.first.absadd_cols(tau)
        self.second.absadd_cols(tau)

    def absadd_rows(self, sigma):
        m, p = self.first.size()[0], self.second.size()[0]
        s_m, s_p = splitm(sigma, m, p)
        self.first.absadd_rows(s_m)
        self.second.absadd_rows(s_p)


class _ConeStack2:
    """ProbLPCone (lp.rs:195-218), ProbSDPCone (sdp.rs:194-218)"""

    def __init__(self, m, p, cone_m, cone_p):
        self.m, self.p, self.cone_m, self.cone_p = m, p, cone_m, cone_p

    def proj(self, dual_cone, x):
        x_m, x_p = splitm(x, self.m, self.p)
        return self.cone_m.proj(dual_cone, x_m) and self.cone_p.proj(dual_cone, x_p)

    def product_group(self, dp_tau, group):
        t_m, t_p = splitm(dp_tau, self.m, self.p)
        self.cone_m.product_group(t_m, group)
        self.cone_p.product_group(t_p, group)


class _Dense:
    """stacked dense description for the fused loop: A (m x n col-major), b, c, cone segments"""

    def __init__(self, n, m, mat_a, vec_b, vec_c, seg_type, seg_len, vec_b_rowabs=None):
        self.n, self.m = n, m
        self.mat_a, self.vec_b, self.vec_c = mat_a, vec_b, vec_c
        self.seg_type, self.seg_len = list(seg_type), list(seg_len)
        self.vec_b_rowabs = vec_b_rowabs


class ProbLP:
    """lp.rs:222-338"""

    def __init__(self, vec_c, mat_g, vec_h, mat_a, vec_b):
        n, m, p = vec_c.size()[0], vec_h.size()[0], vec_b.size()[0]
        assert vec_c.size() == (n, 1) and mat_g.size() == (m, n) and vec_h.size() == (m, 1)
        assert mat_a.size() == (p, n) and vec_b.size() == (p, 1)
        self.L = vec_c.L
        self.vec_c, self.mat_g, self.vec_h, self.mat_a, self.vec_b = vec_c, mat_g, vec_h, mat_a, vec_b
        self.w_solver = np.zeros(0, dtype=self.L.F)
        self._ops = []

    def problem(self):                                             # lp.rs:309-337
        L = self.L
        m, p = self.vec_h.size()[0], self.vec_b.size()[0]
        ops = [self.vec_c.as_op(), self.mat_g.as_op(), self.mat_a.as_op(), self.vec_h.as_op(), self.vec_b.as_op()]
        self._ops = ops
        op_c = _OpVec(ops[0])
        op_a = _OpStack2(ops[1], ops[2])
        op_a.dense_src = self
        op_b = _OpStack2(ops[3], ops[4])
        cone = _ConeStack2(m, p, ConeRPos(L), ConeZero(L))
        self.w_solver = np.zeros(Solver.query_worklen(op_a.size()), dtype=L.F)
        return (op_c, op_a, op_b, cone, self.w_solver)

    def drop(self):
        for o in self._ops:
            o.drop()
        self._ops = []

    def dense(self):
        n, m, p = self.vec_c.size()[0], self.vec_h.size()[0], self.vec_b.size()[0]
        F = self.L.F
        G = self.mat_g.array.reshape((n, m)).T if m else np.zeros((0, n), F)      # col-major -> (m, n)
        A = self.mat_a.array.reshape((n, p)).T if p else np.zeros((0, n), F)
        a = np.asfortranarray(np.vstack([G, A]).astype(F))
        b = np.concatenate([self.vec_h.array, self.vec_b.array]).astype(F)
        return _Dense(n, m + p, a.ravel(order="F"), b, self.vec_c.array.astype(F),
                      [_lib.CONE_RPOS, _lib.CONE_ZERO], [m, p])


class _ProbSOCPOpA:
    """socp.rs:49-163"""

    def __init__(self, L, mats_g, vecs_c, mat_a):
        self.L, self.mats_g, self.vecs_c, self.mat_a = L, mats_g, vecs_c, mat_a

    def size(self):
        p, n = self.mat_a.size()
        s = 0
        for g, c in zip(self.mats_g, self.vecs_c):
            assert c.size() == (n, 1) and g.size()[1] == n
            s += 1 + g.size()[0]
        return (s + p, n)

    def op(self, alpha, x, beta, y):                               # socp.rs:77-101
        p = self.mat_a.size()[0]
        done = 0
        for g, c in zip(self.mats_g, self.vecs_c):
            ni = g.size()[0]
            _, y_1, y_ni = splitm(y, done, 1, ni)
            done += 1 + ni
            c.trans_op(-alpha, x, beta, y_1)
            g.op(-alpha, x, beta, y_ni)
        _, y_p = splitm(y, done, p)
        self.mat_a.op(alpha, x, beta, y_p)

    def trans_op(self, alpha, x, beta, y):                         # socp.rs:103-130
        p = self.mat_a.size()[0]
        self.L.scale(beta, y)
        done = 0
        for g, c in zip(self.mats_g, self.vecs_c):
            ni = g.size()[0]
            _, x_1, x_ni = splitm(x, done, 1, ni)
            done += 1 + ni
            c.op(-alpha, x_1, 1.0, y)
            g.trans_op(-alpha, x_ni, 1.0, y)
        _, x_p = splitm(x, done, p)
        self.mat_a.trans_op(alpha, x_p, 1.0, y)

    def absadd_cols(self, tau):                                    # socp.rs:132-141
        for c in self.vecs_c:
            c.absadd_rows(tau)
        for g in self.mats_g:
            g.absadd_cols(tau)
        self.mat_a.absadd_cols(tau)

    def absadd_rows(self, sigma):                                  # socp.rs:143-162
        p = self.mat_a.size()[0]
        done = 0
        for g, c in zip(self.mats_g, self.vecs_c):
            ni = g.size()[0]
            _, s_1, s_ni = splitm(sigma, done, 1, ni)
            done += 1 + ni
            c.absadd_cols(s_1)
            g.absadd_rows(s_ni)
        _, s_p = splitm(sigma, done, p)
        self.mat_a.absadd_rows(s_p)


class _ProbSOCPOpB:
    """socp.rs:166-280"""

    def __init__(self, L, vecs_h, scls_d, abssum_scls_d, vec_b):
        self.L, self.vecs_h, self.scls_d, self.abssum_scls_d, self.vec_b = L, vecs_h, scls_d, abssum_scls_d, vec_b

    def size(self):
        s = sum(1 + h.size()[0] for h in self.vecs_h)
        return (s + self.vec_b.size()[0], 1)

    def op(self, alpha, x, beta, y):                               # socp.rs:194-217
        L = self.L
        p = self.vec_b.size()[0]
        done = 0
        for h, d in zip(self.vecs_h, self.scls_d):
            ni = h.size()[0]
            _, y_1, y_ni = splitm(y, done, 1, ni)
            done += 1 + ni
            L.scale(beta, y_1)
            L.add(alpha * d, x, y_1)
            h.op(alpha, x, beta, y_ni)
        _, y_p = splitm(y, done, p)
        self.vec_b.op(alpha, x, beta, y_p)

    def trans_op(self, alpha, x, beta, y):                         # socp.rs:219-246
        L = self.L
        p = self.vec_b.size()[0]
        L.scale(beta, y)
        done = 0
        for h, d in zip(self.vecs_h, self.scls_d):
            ni = h.size()[0]
            _, x_1, x_ni = splitm(x, done, 1, ni)
            done += 1 + ni
            L.add(alpha * d, x_1, y)
            h.trans_op(alpha, x_ni, 1.0, y)
        _, x_p = splitm(x, done, p)
        self.vec_b.trans_op(alpha, x_p, 1.0, y)

    def absadd_cols(self, tau):                                    # socp.rs:248-257
        tau.set(0, tau.get(0) + self.abssum_scls_d)
        for h in self.vecs_h:
            h.absadd_cols(tau)
        self.vec_b.absadd_cols(tau)

    def absadd_rows(self, sigma):                                  # socp.rs:259-279 (adds scl_d, not |scl_d|)
        p = self.vec_b.size()[0]
        done = 0
        for h, d in zip(self.vecs_h, self.scls_d):
            ni = h.size()[0]
            _, s_1, s_ni = splitm(sigma, done, 1, ni)
            done += 1 + ni
            s_1.set(0, s_1.get(0) + d)
            h.absadd_rows(s_ni)
        _, s_p = splitm(sigma, done, p)
        self.vec_b.absadd_rows(s_p)


class _ProbSOCPCone:
    """socp.rs:284-332"""

    def __init__(self, L, nis, p):
        self.nis, self.p = nis, p
        self.cone_soc, self.cone_zero = ConeSOC(L), ConeZero(L)

    def proj(self, dual_cone, x):
        done = 0
        for ni in self.nis:
            _, x_ni1 = splitm(x, done, 1 + ni)
            done += 1 + ni
            if not self.cone_soc.proj(dual_cone, x_ni1):
                return False
        _, x_p = splitm(x, done, self.p)
        return self.cone_zero.proj(dual_cone, x_p)

    def product_group(self, dp_tau, group):
        done = 0
        for ni in self.nis:
            _, t = splitm(dp_tau, done, 1 + ni)
            done += 1 + ni
            self.cone_soc.product_group(t, group)
        _, t_p = splitm(dp_tau, done, self.p)
        self.cone_zero.product_group(t_p, group)


class ProbSOCP:
    """socp.rs:336-474"""

    def __init__(self, vec_f, mats_g, vecs_h, vecs_c, scls_d, mat_a, vec_b):
        n, m, p = vec_f.size()[0], len(mats_g), vec_b.size()[0]
        assert len(vecs_h) == m and len(vecs_c) == m and len(scls_d) == m and vec_f.size() == (n, 1)
        for i in range(m):
            ni = mats_g[i].size()[0]
            assert mats_g[i].size() == (ni, n) and vecs_h[i].size() == (ni, 1) and vecs_c[i].size() == (n, 1)
        assert mat_a.size() == (p, n) and vec_b.size() == (p, 1)
        self.L = vec_f.L
        self.vec_f, self.mats_g, self.vecs_h, self.vecs_c = vec_f, mats_g, vecs_h, vecs_c
        self.scls_d = [float(d) for d in scls_d]
        self.mat_a, self.vec_b = mat_a, vec_b
        self.w_solver = np.zeros(0, dtype=self.L.F)
        self._ops = []

    def problem(self):                                             # socp.rs:430-473
        L = self.L
        p = self.vec_b.size()[0]
        of = self.vec_f.as_op()
        og = [g.as_op() for g in self.mats_g]
        ocs = [c.as_op() for c in self.vecs_c]
        oa, ob = self.mat_a.as_op(), self.vec_b.as_op()
        oh = [h.as_op() for h in self.vecs_h]
        self._ops = [of, oa, ob] + og + ocs + oh
        d_sl = L.Sl.new_ref(np.asarray(self.scls_d, dtype=L.F))
        abssum_d = L.abssum(d_sl, 1)
        d_sl.drop()
        op_c = _OpVec(of)
        op_a = _ProbSOCPOpA(L, og, ocs, oa)
        op_a.dense_src = self
        op_b = _ProbSOCPOpB(L, oh, self.scls_d, abssum_d, ob)
        cone = _ProbSOCPCone(L, [g.size()[0] for g in self.mats_g], p)
        self.w_solver = np.zeros(Solver.query_worklen(op_a.size()), dtype=L.F)
        return (op_c, op_a, op_b, cone, self.w_solver)

    def drop(self):
        for o in self._ops:
            o.drop()
        self._ops = []

    def dense(self):
        """one stacked operator: rows of cone i are [-c_i^T ; -G_i] (socp.rs:88-93), b = [d_i ; h_i]"""
        F = self.L.F
        n, p = self.vec_f.size()[0], self.vec_b.size()[0]
        rows, bs, babs, st, sl = [], [], [], [], []
        for g, h, c, d in zip(self.mats_g, self.vecs_h, self.vecs_c, self.scls_d):
            ni = g.size()[0]
            rows.append(-c.array.reshape(1, n))
            rows.append(-(g.array.reshape((n, ni)).T))
            bs += [np.array([d], F), h.array]
            babs += [np.array([d], F), np.abs(h.array)]
            st.append(_lib.CONE_SOC)
            sl.append(1 + ni)
        rows.append(self.mat_a.array.reshape((n, p)).T if p else np.zeros((0, n), F))
        bs.append(self.vec_b.array)
        babs.append(np.abs(self.vec_b.array))
        st.append(_lib.CONE_ZERO)
        sl.append(p)
        a = np.asfortranarray(np.vstack(rows).astype(F))
        m = a.shape[0]
        return _Dense(n, m, a.ravel(order="F"), np.concatenate(bs).astype(F), self.vec_f.array.astype(F), st, sl,
                      np.concatenate(babs).astype(F))


class ProbSDP:
    """sdp.rs:222-332"""

    def __init__(self, vec_c, syms_f, mat_a, vec_b, eps_zero):
        n, p = vec_c.size()[0], vec_b.size()[0]
        assert vec_c.size() == (n, 1) and len(syms_f) == n + 1
        k = syms_f[0].size()[0]
        for s in syms_f:
            assert s.is_sympack() and s.size() == (k, k)
        assert mat_a.size() == (p, n) and vec_b.size() == (p, 1)
        L = vec_c.L
        self.L = L
        fsqrt2 = math.sqrt(2.0)
        syms_f = [s.clone() for s in syms_f]
        for s in syms_f:                                           # sdp.rs:271-274
            s.set_scale_nondiag(fsqrt2)
            s.set_reshape_colvec()
        self.symvec_f_n = syms_f.pop()
        sk = self.symvec_f_n.size()[0]
        self.symmat_f = MatBuild(L, MatType.General(sk, n))
        for c in range(n):                                         # sdp.rs:279-280
            self.symmat_f.array[c * sk:(c + 1) * sk] = syms_f[c].array
        self.vec_c, self.mat_a, self.vec_b, self.eps_zero = vec_c, mat_a, vec_b, eps_zero
        self.w_cone_psd = np.zeros(0, dtype=L.F)
        self.w_solver = np.zeros(0, dtype=L.F)
        self._ops = []
        self._cone = None

    def problem(self):                                             # sdp.rs:299-331
        L = self.L
        p, sk = self.vec_b.size()[0], self.symvec_f_n.size()[0]
        ops = [self.vec_c.as_op(), self.symmat_f.as_op(), self.mat_a.as_op(), self.symvec_f_n.as_op(), self.vec_b.as_op()]
        self._ops = ops
        op_c = _OpVec(ops[0])
        op_a = _OpStack2(ops[1], ops[2])
        op_a.dense_src = self
        op_b = _OpStack2(ops[3], ops[4], sign_first=-1.0)
        self.w_cone_psd = np.zeros(ConePSD.query_worklen(L, sk), dtype=L.F)
        self._cone = ConePSD(L, self.w_cone_psd, self.eps_zero)
        cone = _ConeStack2(sk, p, self._cone, ConeZero(L))
        self.w_solver = np.zeros(Solver.query_worklen(op_a.size()), dtype=L.F)
        return (op_c, op_a, op_b, cone, self.w_solver)

    def drop(self):
        for o in self._ops:
            o.drop()
        self._ops = []
        if self._cone is not None:
            self._cone.drop()
            self._cone = None

    def dense(self):
        F = self.L.F
        n, p, sk = self.vec_c.size()[0], self.vec_b.size()[0], self.symvec_f_n.size()[0]
        Fm = self.symmat_f.array.reshape((n, sk)).T
        A = self.mat_a.array.reshape((n, p)).T if p else np.zeros((0, n), F)
        a = np.asfortranarray(np.vstack([Fm, A]).astype(F))
        b = np.concatenate([-self.symvec_f_n.array, self.vec_b.array]).astype(F)   # sdp.rs:152: -alpha on F_n
        return _Dense(n, sk + p, a.ravel(order="F"), b, self.vec_c.array.astype(F),
                      [_lib.CONE_PSD, _lib.CONE_ZERO], [sk, p])


# ------------------------------------------------------------------------------------------------------
# ProbQP (totsu/src/problem/qp.rs) and ProbQCQP (qcqp.rs): variables (x, t), minimise t, the quadratic
# objective / constraints as rotated second-order cones on (r, s, P^{1/2} x).
# ------------------------------------------------------------------------------------------------------

class _ProbQPOpC:
    """qp.rs:9-62 (also qcqp.rs:9-58): c = [0_n ; 1]"""

    def __init__(self, L, n):
        self.L, self.n = L, n

    def size(self):
        return (self.n + 1, 1)

    def op(self, alpha, x, beta, y):
        y_n, y_t = splitm(y, self.n, 1)
        self.L.scale(beta, y_n)
        self.L.scale(beta, y_t)
        self.L.add(alpha, x, y_t)

    def trans_op(self, alpha, x, beta, y):
        _, x_t = splitm(x, self.n, 1)
        self.L.scale(beta, y)
        self.L.add(alpha, x_t, y)

    def absadd_cols(self, tau):
        tau.set(0, tau.get(0) + 1.0)

    def absadd_rows(self, sigma):
        _, s_t = splitm(sigma, self.n, 1)
        s_t.set(0, s_t.get(0) + 1.0)


class _ProbQCQPOpA:
    """qcqp.rs:62-183; ProbQPOpA (qp.rs:66-172) is the m1 == 1 case with the extra [G ; A] rows"""

    def __init__(self, L, syms_p_sqrt, vecs_q, mats_tail):
        self.L, self.ps, self.qs, self.tail = L, syms_p_sqrt, vecs_q, mats_tail    # tail: list of MatOp (rows x n)

    def _dims(self):
        n = self.qs[0].size()[0]
        return n, len(self.ps)

    def size(self):
        n, m1 = self._dims()
        return (m1 * (2 + n) + sum(t.size()[0] for t in self.tail), n + 1)

    def op(self, alpha, x, beta, y):
        L = self.L
        n, m1 = self._dims()
        x_n, x_t = splitm(x, n, 1)
        for i, (ps, q) in enumerate(zip(self.ps, self.qs)):
            _, y_r, y_s, y_n = splitm(y, i * (2 + n), 1, 1, n)
            L.scale(beta, y_r)
            q.trans_op(alpha, x_n, beta, y_s)
            if i == 0:
                L.add(-alpha, x_t, y_s)
            ps.op(-alpha, x_n, beta, y_n)
        done = m1 * (2 + n)
        for t in self.tail:
            r = t.size()[0]
            _, y_p = splitm(y, done, r)
            t.op(alpha, x_n, beta, y_p)
            done += r

    def trans_op(self, alpha, x, beta, y):
        L = self.L
        n, m1 = self._dims()
        y_n, y_t = splitm(y, n, 1)
        L.scale(beta, y_n)
        L.scale(beta, y_t)
        for i, (ps, q) in enumerate(zip(self.ps, self.qs)):
            _, _, x_s, x_n = splitm(x, i * (2 + n), 1, 1, n)
            q.op(alpha, x_s, 1.0, y_n)
            ps.op(-alpha, x_n, 1.0, y_n)
            if i == 0:
                L.add(-alpha, x_s, y_t)
        done = m1 * (2 + n)
        for t in self.tail:
            r = t.size()[0]
            _, x_p = splitm(x, done, r)
            t.trans_op(alpha, x_p, 1.0, y_n)
            done += r

    def absadd_cols(self, tau):
        n, _ = self._dims()
        tau_n, tau_t = splitm(tau, n, 1)
        for q in self.qs:
            q.absadd_rows(tau_n)
        for ps in self.ps:
            ps.absadd_cols(tau_n)
        for t in self.tail:
            t.absadd_cols(tau_n)
        tau_t.set(0, tau_t.get(0) + 1.0)

    def absadd_rows(self, sigma):
        n, m1 = self._dims()
        for i, (ps, q) in enumerate(zip(self.ps, self.qs)):
            _, _, s_s, s_n = splitm(sigma, i * (2 + n), 1, 1, n)
            q.absadd_cols(s_s)
            if i == 0:
                s_s.set(0, s_s.get(0) + 1.0)
            ps.absadd_rows(s_n)
        done = m1 * (2 + n)
        for t in self.tail:
            r = t.size()[0]
            _, s_p = splitm(sigma, done, r)
            t.absadd_rows(s_p)
            done += r


class _ProbQCQPOpB:
    """qcqp.rs:187-294; ProbQPOpB (qp.rs:176-260) is scls_r == [0] with the extra [h ; b] rows"""

    def __init__(self, L, n, scls_r, vecs_tail):
        self.L, self.n, self.rs, self.tail = L, n, scls_r, vecs_tail

    def size(self):
        return (len(self.rs) * (2 + self.n) + sum(t.size()[0] for t in self.tail), 1)

    def op(self, alpha, x, beta, y):
        L, n = self.L, self.n
        for i, r in enumerate(self.rs):
            _, y_r, y_s, y_n = splitm(y, i * (2 + n), 1, 1, n)
            L.scale(beta, y_r)
            L.add(alpha, x, y_r)
            L.scale(beta, y_s)
            L.add(-alpha * r, x, y_s)
            L.scale(beta, y_n)
        done = len(self.rs) * (2 + n)
        for t in self.tail:
            k = t.size()[0]
            _, y_p = splitm(y, done, k)
            t.op(alpha, x, beta, y_p)
            done += k

    def trans_op(self, alpha, x, beta, y):
        L, n = self.L, self.n
        L.scale(beta, y)
        for i, r in enumerate(self.rs):
            _, x_r, x_s, _ = splitm(x, i * (2 + n), 1, 1, n)
            L.add(alpha, x_r, y)
            L.add(-alpha * r, x_s, y)
        done = len(self.rs) * (2 + n)
        for t in self.tail:
            k = t.size()[0]
            _, x_p = splitm(x, done, k)
            t.trans_op(alpha, x_p, 1.0, y)
            done += k

    def absadd_cols(self, tau):
        tau.set(0, tau.get(0) + len(self.rs) + sum(abs(r) for r in self.rs))
        for t in self.tail:
            t.absadd_cols(tau)

    def absadd_rows(self, sigma):
        n = self.n
        for i, r in enumerate(self.rs):
            _, s_r, s_s, _ = splitm(sigma, i * (2 + n), 1, 1, n)
            s_r.set(0, s_r.get(0) + 1.0)
            s_s.set(0, s_s.get(0) + abs(r))
        done = len(self.rs) * (2 + n)
        for t in self.tail:
            k = t.size()[0]
            _, s_p = splitm(sigma, done, k)
            t.absadd_rows(s_p)
            done += k


class _ConeList:
    """consecutive (cone, length) blocks: ProbQPCone (qp.rs:264-300), ProbQCQPCone (qcqp.rs:298-345)"""

    def __init__(self, blocks):
        self.blocks = blocks

    def proj(self, dual_cone, x):
        done = 0
        for cone, ln in self.blocks:
            _, xb = splitm(x, done, ln)
            done += ln
            if not cone.proj(dual_cone, xb):
                return False
        return True

    def product_group(self, dp_tau, group):
        done = 0
        for cone, ln in self.blocks:
            _, tb = splitm(dp_tau, done, ln)
            done += ln
            cone.product_group(tb, group)


def _dense_quadratic(L, n, ps_sqrt, qs, rs, tails_a, tails_b, tail_types):
    """stacked rows of ProbQP / ProbQCQP: block i = [0 ; q_i^T, -(i==0) ; -P_i^{1/2}, 0], b = [1 ; -r_i ; 0]"""
    F = L.F
    rows, bs, st, sl = [], [], [], []
    for i, (ps, q, r) in enumerate(zip(ps_sqrt, qs, rs)):
        full = np.zeros((n, n), dtype=F)
        for c in range(n):
            for rr in range(c + 1):
                full[rr, c] = full[c, rr] = ps.array[c * (c + 1) // 2 + rr]
        blk = np.zeros((2 + n, n + 1), dtype=F)
        blk[1, :n] = q.array
        blk[1, n] = -1.0 if i == 0 else 0.0
        blk[2:, :n] = -full
        rows.append(blk)
        bb = np.zeros(2 + n, dtype=F)
        bb[0] = 1.0
        bb[1] = -r
        bs.append(bb)
        st.append(_lib.CONE_ROTSOC)
        sl.append(2 + n)
    for ta, tb, tt in zip(tails_a, tails_b, tail_types):
        k = ta.size()[0]
        blk = np.zeros((k, n + 1), dtype=F)
        if k:
            blk[:, :n] = ta.array.reshape((n, k)).T
        rows.append(blk)
        bs.append(tb.array.astype(F))
        st.append(tt)
        sl.append(k)
    a = np.asfortranarray(np.vstack(rows).astype(F))
    c = np.zeros(n + 1, dtype=F)
    c[n] = 1.0
    return _Dense(n + 1, a.shape[0], a.ravel(order="F"), np.concatenate(bs).astype(F), c, st, sl)


class ProbQP:
    """qp.rs:304-440: min (1/2) x^T P x + q^T x  s.t. G x <= h, A x = b"""

    def __init__(self, sym_p, vec_q, mat_g, vec_h, mat_a, vec_b, eps_zero):
        n, m, p = vec_q.size()[0], vec_h.size()[0], vec_b.size()[0]
        assert sym_p.is_sympack() and sym_p.size() == (n, n) and vec_q.size() == (n, 1)
        assert mat_g.size() == (m, n) and vec_h.size() == (m, 1) and mat_a.size() == (p, n) and vec_b.size() == (p, 1)
        self.L = vec_q.L
        self.sym_p_sqrt = sym_p.clone().sqrt(eps_zero)            # qp.rs:386
        self.vec_q, self.mat_g, self.vec_h, self.mat_a, self.vec_b = vec_q, mat_g, vec_h, mat_a, vec_b
        self.w_solver = np.zeros(0, dtype=self.L.F)
        self._ops = []

    def problem(self):                                             # qp.rs:400-439
        L = self.L
        n, m, p = self.vec_q.size()[0], self.vec_h.size()[0], self.vec_b.size()[0]
        ops = [self.sym_p_sqrt.as_op(), self.vec_q.as_op(), self.mat_g.as_op(), self.mat_a.as_op(),
               self.vec_h.as_op(), self.vec_b.as_op()]
        self._ops = ops
        op_c = _ProbQPOpC(L, n)
        op_a = _ProbQCQPOpA(L, [ops[0]], [ops[1]], [ops[2], ops[3]])
        op_a.dense_src = self
        op_b = _ProbQCQPOpB(L, n, [0.0], [ops[4], ops[5]])
        from .cone import ConeRotSOC
        cone = _ConeList([(ConeRotSOC(L), 2 + n), (ConeRPos(L), m), (ConeZero(L), p)])
        self.w_solver = np.zeros(Solver.query_worklen(op_a.size()), dtype=L.F)
        return (op_c, op_a, op_b, cone, self.w_solver)

    def drop(self):
        for o in self._ops:
            o.drop()
        self._ops = []

    def dense(self):
        n = self.vec_q.size()[0]
        return _dense_quadratic(self.L, n, [self.sym_p_sqrt], [self.vec_q], [0.0], [self.mat_g, self.mat_a],
                                [self.vec_h, self.vec_b], [_lib.CONE_RPOS, _lib.CONE_ZERO])


class ProbQCQP:
    """qcqp.rs:349-505: min (1/2) x^T P_0 x + q_0^T x + r_0  s.t. (1/2) x^T P_i x + q_i^T x + r_i <= 0, A x = b"""

    def __init__(self, syms_p, vecs_q, scls_r, mat_a, vec_b, eps_zero):
        p, n = mat_a.size()
        m1 = len(syms_p)
        assert len(vecs_q) == m1 and len(scls_r) == m1
        for sp, q in zip(syms_p, vecs_q):
            assert sp.is_sympack() and sp.size() == (n, n) and q.size() == (n, 1)
        assert vec_b.size() == (p, 1)
        self.L = mat_a.L
        self.syms_p_sqrt = [sp.clone().sqrt(eps_zero) for sp in syms_p]      # qcqp.rs:443-448
        self.vecs_q, self.scls_r, self.mat_a, self.vec_b = vecs_q, [float(r) for r in scls_r], mat_a, vec_b
        self.w_solver = np.zeros(0, dtype=self.L.F)
        self._ops = []

    def problem(self):                                             # qcqp.rs:463-504
        L = self.L
        p, n = self.mat_a.size()
        ps = [s.as_op() for s in self.syms_p_sqrt]
        qs = [q.as_op() for q in self.vecs_q]
        oa, ob = self.mat_a.as_op(), self.vec_b.as_op()
        self._ops = ps + qs + [oa, ob]
        op_c = _ProbQPOpC(L, n)
        op_a = _ProbQCQPOpA(L, ps, qs, [oa])
        op_a.dense_src = self
        op_b = _ProbQCQPOpB(L, n, self.scls_r, [ob])
        from .cone import ConeRotSOC
        cone = _ConeList([(ConeRotSOC(L), 2 + n) for _ in ps] + [(ConeZero(L), p)])
        self.w_solver = np.zeros(Solver.query_worklen(op_a.size()), dtype=L.F)
        return (op_c, op_a, op_b, cone, self.w_solver)

    def drop(self):
        for o in self._ops:
            o.drop()
        self._ops = []

    def dense(self):
        n = self.mat_a.size()[1]
        return _dense_quadratic(self.L, n, self.syms_p_sqrt, self.vecs_q, self.scls_r, [self.mat_a], [self.vec_b],
                                [_lib.CONE_ZERO])
