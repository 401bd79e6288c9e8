"""`Solver<L>`: the first-order conic solver, generic over a LinAlg backend `L`, an `Operator` triple and a
`Cone` -- the trait-level drop-in.  Mirror of totsu_core/src/solver/solver.rs (same names, same argument
meaning, same error behaviour); every arithmetic step is a call on `L`, so with `L = F32HIP` each one is a
gfx950 kernel.  The fused device-resident loop for dense operators is `totsu_amd.fused.FusedSolver`.
"""
import logging
import math

import numpy as np

log = logging.getLogger("totsu_amd")     # the reference logs through the `log` crate facade (solver.rs:342-446)

from .linalg import splitm


class SolverError(Exception):
    """solver_error.rs:3-17"""
    Unbounded, Infeasible, ExcessIter, InvalidOp, WorkShortage, ConeFailure = 1, 2, 3, 4, 5, 6
    _TEXT = {1: "Unbounded: found an unbounded certificate",
             2: "Infeasible: found an infeasibile certificate",
             3: "ExcessIter: exceed max iterations",
             4: "InvalidOp: invalid Operator",
             5: "WorkShortage: shortage of work slice length",
             6: "ConeFailure: failure caused by Cone"}

    def __init__(self, kind):
        super().__init__(SolverError._TEXT[kind])
        self.kind = kind


class SolverParam:
    """solver.rs:13-41"""

    def __init__(self):
        self.max_iter = None
        self.eps_acc = 1e-6
        self.eps_inf = 1e-6
        self.eps_zero = 1e-12
        self.log_period = 10_000
        # not in the reference: arithmetic of the iterate updates of the fused device loop, "compensated" | "plain"
        # (thip_param.state_arith, include/totsu_f32hip.h); the trait-level loop ignores it
        self.state_arith = "compensated"

    def __repr__(self):
        return ("SolverParam { max_iter: %r, eps_acc: %r, eps_inf: %r, eps_zero: %r, log_period: %r }"
                % (self.max_iter, self.eps_acc, self.eps_inf, self.eps_zero, self.log_period))


class _SelfDualEmbed:
    """solver.rs:45-184"""

    def __init__(self, L, c, a, b):
        self.L, self.c, self.a, self.b = L, c, a, b

    def fr_norm(self, op, work_v, work_t):                          # solver.rs:85-107
        L = self.L
        assert work_v.len() == op.size()[1] and work_t.len() == op.size()[0]
        L.scale(0.0, work_v)
        sq_norm = 0.0
        for row in range(op.size()[1]):
            work_v.set(row, 1.0)
            op.op(1.0, work_v, 0.0, work_t)
            nn = L.norm(work_t)
            sq_norm = sq_norm + nn * nn
            work_v.set(row, 0.0)
        return math.sqrt(sq_norm)

    def op(self, alpha, x, beta, y):                                # solver.rs:109-131
        L = self.L
        m, n = self.a.size()
        assert x.len() == n + m + m + 1 and y.len() == n + m + 1
        x_x, x_y, x_s, x_tau = splitm(x, n, m, m, 1)
        y_n, y_m, y_1 = splitm(y, n, m, 1)
        self.a.trans_op(alpha, x_y, beta, y_n)
        self.c.op(alpha, x_tau, 1.0, y_n)
        self.a.op(-alpha, x_x, beta, y_m)
        L.add(-alpha, x_s, y_m)
        self.b.op(alpha, x_tau, 1.0, y_m)
        self.c.trans_op(-alpha, x_x, beta, y_1)
        self.b.trans_op(-alpha, x_y, 1.0, y_1)

    def trans_op(self, alpha, x, beta, y):                          # solver.rs:133-157
        L = self.L
        m, n = self.a.size()
        assert x.len() == n + m + 1 and y.len() == n + m + m + 1
        x_n, x_m, x_1 = splitm(x, n, m, 1)
        y_x, y_y, y_s, y_tau = splitm(y, n, m, m, 1)
        self.a.trans_op(-alpha, x_m, beta, y_x)
        self.c.op(-alpha, x_1, 1.0, y_x)
        self.a.op(alpha, x_n, beta, y_y)
        self.b.op(-alpha, x_1, 1.0, y_y)
        L.scale(beta, y_s)
        L.add(-alpha, x_m, y_s)
        self.c.trans_op(alpha, x_n, beta, y_tau)
        self.b.trans_op(alpha, x_m, 1.0, y_tau)

    def abssum(self, tau, sigma):                                   # solver.rs:159-183
        L = self.L
        m, n = self.a.size()
        L.scale(0.0, tau)
        tau_x, tau_y, tau_s, tau_tau = splitm(tau, n, m, m, 1)
        self.a.absadd_cols(tau_x)
        self.c.absadd_rows(tau_x)
        self.a.absadd_rows(tau_y)
        self.b.absadd_rows(tau_y)
        L.adds(1.0, tau_s)
        self.c.absadd_cols(tau_tau)
        self.b.absadd_cols(tau_tau)
        sigma_n, sigma_m, sigma_1 = splitm(sigma, n, m, 1)
        L.copy(tau_x, sigma_n)
        L.copy(tau_y, sigma_m)
        L.add(1.0, tau_s, sigma_m)
        L.copy(tau_tau, sigma_1)


class Solver:
    """solver.rs:219-322.  `Solver(L).par(lambda p: ...).solve((op_c, op_a, op_b, cone, work))`."""

    def __init__(self, L):
        self.L = L
        self.param = SolverParam()
        self.trace = None           # optional list collecting (i, kind, v0, v1, v2) like the debug log
        # drop-in fast path: when L is F32HIP and the operators come from one of this package's problem builders
        # (dense MatOps), solve() runs the device-resident fused loop with this schedule; None = always take the
        # trait-level loop below (one L call per reference call)
        self.fused = "sweep"            # one pass over A per iteration where the kernel takes the problem, else "carried"
        self.iters = -1

    @staticmethod
    def query_worklen(op_a_size):                                   # solver.rs:231-249
        m, n = op_a_size
        return ((n + m + m + 1) + (n + m + 1) + (n + m + m + 1) + (n + m + 1)
                + (n + m + m + 1) + (n + m + m + 1))

    def par(self, f):                                               # solver.rs:265-270
        f(self.param)
        return self

    def solve(self, prob):                                          # solver.rs:285-321
        op_c, op_a, op_b, cone, work = prob
        L = self.L
        m, n = op_a.size()
        if op_c.size() != (n, 1) or op_b.size() != (m, 1):
            raise SolverError(SolverError.InvalidOp)
        if Solver.query_worklen((m, n)) > len(work):
            raise SolverError(SolverError.WorkShortage)
        src = getattr(op_a, "dense_src", None)
        if self.fused and src is not None and getattr(L, "name", "") == "F32HIP" and self.trace is None:
            from .fused import FusedSolver
            log.info("----- Initializing")
            fs = FusedSolver.from_dense(src.dense(), self.param, self.fused)
            log.info("----- Started")
            try:
                period = int(self.param.log_period)
                if period > 0 and log.isEnabledFor(logging.DEBUG):
                    # solver.rs:371-395: the residual triple every log_period iterations
                    while True:
                        r = fs.run(period, min(64, period))
                        _log_status(r)
                        if r.state != -1:
                            break
                else:
                    r = fs.run(-1, 64)
                    _log_status(r)
                x, y = fs.solution()
            finally:
                fs.destroy()
            _log_end(r.state)
            self.iters = r.iters
            work[:n] = x                  # solver.rs:317-320: the answers are the head of the caller's work slice
            work[n:n + m] = y
            if r.state != 0:
                raise SolverError(r.state)
            return work[:n], work[n:n + m]
        core = _SolverCore(L, self.param, _SelfDualEmbed(L, op_c, op_a, op_b), cone, self.trace)
        w = L.Sl.new_mut(work)
        try:
            if getattr(L, "name", "") == "F32HIP":
                # one C-ABI call per reference call: opt into the library's deferred, batched small calls for this solve
                from ._lib import lazy_calls
                with lazy_calls():
                    err = core.solve(w)
            else:
                err = core.solve(w)
        finally:
            w.drop()                 # the host `work` is up to date again (f32cuda_slice.rs:203-207)
        self.iters = core.iters
        if err is not None:
            raise SolverError(err)
        return work[:n], work[n:n + m]


def _log_status(r):
    if r.kind == 0:
        log.debug("%d: pri_dual_gap %.2e %.2e %.2e", r.iters, r.cri[0], r.cri[1], r.cri[2])
    else:
        log.debug("%d: unbdd_infeas %.2e %.2e", r.iters, r.cri[0], r.cri[1])


def _log_end(state):
    if state == 0:
        log.info("----- Converged")
    else:
        log.warning("----- %s", {1: "Unbounded", 2: "Infeasible", 3: "ExcessIter"}.get(state, "Error %d" % state))


class _SolverCore:
    """solver.rs:326-657"""

    def __init__(self, L, par, op_k, cone, trace):
        self.L, self.par, self.op_k, self.cone, self.trace = L, par, op_k, cone, trace
        self.iters = -1

    def solve(self, work):                                          # solver.rs:340-458
        L, par = self.L, self.par
        m, n = self.op_k.a.size()
        log.info("----- Initializing")
        norm_b, norm_c = self.calc_norms(work)
        N, M = n + m + m + 1, n + m + 1
        x, y, dp_tau, dp_sigma, tmpw = splitm(work, N, M, N, M, 2 * N)
        self.init_vecs(x, y)
        self.calc_precond(dp_tau, dp_sigma)
        log.info("----- Started")
        i = 0
        while True:
            excess_iter = (i + 1 >= par.max_iter) if par.max_iter is not None else False
            log_trig = (i % par.log_period == 0) if par.log_period > 0 else False
            val_tau = self.update_vecs(x, y, dp_tau, dp_sigma, tmpw)
            if val_tau is None:
                return SolverError.ConeFailure
            self.iters = i
            if val_tau > par.eps_zero:
                cri_pri, cri_dual, cri_gap = self.criteria_conv(x, norm_c, norm_b, tmpw)
                if self.trace is not None:
                    self.trace.append((i, 0, cri_pri, cri_dual, cri_gap))
                term_conv = cri_pri <= par.eps_acc and cri_dual <= par.eps_acc and cri_gap <= par.eps_acc
                if log_trig or excess_iter or term_conv:
                    log.debug("%d: pri_dual_gap %.2e %.2e %.2e", i, cri_pri, cri_dual, cri_gap)
                if excess_iter or term_conv:
                    x_x_ast, x_y_ast = splitm(x, n, m)
                    L.scale(1.0 / val_tau, x_x_ast)
                    L.scale(1.0 / val_tau, x_y_ast)
                    _log_end(0 if term_conv else 3)
                    return None if term_conv else SolverError.ExcessIter
            else:
                cri_unbdd, cri_infeas = self.criteria_inf(x, norm_c, norm_b, tmpw)
                if self.trace is not None:
                    self.trace.append((i, 1, cri_unbdd, cri_infeas, 0.0))
                term_unbdd = cri_unbdd <= par.eps_inf
                term_infeas = cri_infeas <= par.eps_inf
                if log_trig or excess_iter or term_unbdd or term_infeas:
                    log.debug("%d: unbdd_infeas %.2e %.2e", i, cri_unbdd, cri_infeas)
                if excess_iter or term_unbdd or term_infeas:
                    _log_end(1 if term_unbdd else (2 if term_infeas else 3))
                    if term_unbdd:
                        return SolverError.Unbounded
                    if term_infeas:
                        return SolverError.Infeasible
                    return SolverError.ExcessIter
            i += 1

    def calc_norms(self, work):                                     # solver.rs:460-481
        L = self.L
        work1 = np.zeros(1, dtype=L.F)
        work_one = L.Sl.new_mut(work1)
        try:
            mb = self.op_k.b.size()[0]
            (t,) = splitm(work, mb)
            norm_b = self.op_k.fr_norm(self.op_k.b, work_one, t)
            nc = self.op_k.c.size()[0]
            (t,) = splitm(work, nc)
            norm_c = self.op_k.fr_norm(self.op_k.c, work_one, t)
        finally:
            work_one.drop()
        return norm_b, norm_c

    def init_vecs(self, x, y):                                      # solver.rs:483-494
        L = self.L
        m, n = self.op_k.a.size()
        L.scale(0.0, x)
        L.scale(0.0, y)
        x.set(n + m + m, 1.0)

    def calc_precond(self, dp_tau, dp_sigma):                       # solver.rs:496-524
        L, par = self.L, self.par
        m, n = self.op_k.a.size()
        self.op_k.abssum(dp_tau, dp_sigma)
        if hasattr(L, "recip_max"):
            L.recip_max(par.eps_zero, dp_tau)
            L.recip_max(par.eps_zero, dp_sigma)
        else:
            F = L.F
            t = dp_tau.get_mut()
            t[:] = F(1.0) / np.maximum(t, F(par.eps_zero))
            s = dp_sigma.get_mut()
            s[:] = F(1.0) / np.maximum(s, F(par.eps_zero))

        def group(tau_group):
            if tau_group.len() > 0:
                g = tau_group.get_mut()
                g[:] = g.min()
        _, dpt_dual_cone, dpt_cone, _ = splitm(dp_tau, n, m, m, 1)
        self.cone.product_group(dpt_dual_cone, group)
        self.cone.product_group(dpt_cone, group)

    def update_vecs(self, x, y, dp_tau, dp_sigma, tmpw):            # solver.rs:526-571
        L = self.L
        m, n = self.op_k.a.size()
        rx, tx = splitm(tmpw, x.len(), x.len())
        L.copy(x, rx)
        self.op_k.trans_op(-1.0, y, 0.0, tx)
        L.transform_di(1.0, dp_tau, tx, 1.0, x)
        _, x_y, x_s, x_tau = splitm(x, n, m, m, 1)
        if not self.cone.proj(True, x_y):
            return None
        if not self.cone.proj(False, x_s):
            return None
        val_tau = max(x_tau.get(0), 0.0)
        x_tau.set(0, val_tau)
        L.add(-2.0, x, rx)
        (ty,) = splitm(tx, y.len())
        self.op_k.op(-1.0, rx, 0.0, ty)
        L.transform_di(1.0, dp_sigma, ty, 1.0, y)
        _, y_1 = splitm(y, n + m, 1)
        kappa = min(y_1.get(0), 0.0)
        y_1.set(0, kappa)
        return val_tau

    def criteria_conv(self, x, norm_c, norm_b, tmpw):               # solver.rs:573-612
        L = self.L
        m, n = self.op_k.a.size()
        x_x, x_y, x_s, x_tau = splitm(x, n, m, m, 1)
        p, d = splitm(tmpw, m, n)
        val_tau = x_tau.get(0)
        assert val_tau > 0.0
        work1 = np.ones(1, dtype=L.F)
        work_one = L.Sl.new_mut(work1)
        try:
            L.copy(x_s, p)
            self.op_k.b.op(-1.0, work_one, 1.0 / val_tau, p)
            self.op_k.a.op(1.0 / val_tau, x_x, 1.0, p)
            self.op_k.c.op(1.0, work_one, 0.0, d)
            self.op_k.a.trans_op(1.0 / val_tau, x_y, 1.0, d)
            self.op_k.c.trans_op(1.0 / val_tau, x_x, 0.0, work_one)
            g_x = work_one.get(0)
            self.op_k.b.trans_op(1.0 / val_tau, x_y, 0.0, work_one)
            g_y = work_one.get(0)
        finally:
            work_one.drop()
        g = g_x + g_y
        cri_pri = L.norm(p) / (1.0 + norm_b)
        cri_dual = L.norm(d) / (1.0 + norm_c)
        cri_gap = abs(g) / (1.0 + abs(g_x) + abs(g_y))
        return cri_pri, cri_dual, cri_gap

    def criteria_inf(self, x, norm_c, norm_b, tmpw):                # solver.rs:614-656
        L, par = self.L, self.par
        m, n = self.op_k.a.size()
        x_x, x_y, x_s, _ = splitm(x, n, m, m, 1)
        p, d = splitm(tmpw, m, n)
        work1 = np.zeros(1, dtype=L.F)
        work_one = L.Sl.new_mut(work1)
        try:
            L.copy(x_s, p)
            self.op_k.a.op(1.0, x_x, 1.0, p)
            self.op_k.a.trans_op(1.0, x_y, 0.0, d)
            self.op_k.c.trans_op(-1.0, x_x, 0.0, work_one)
            m_cx = work_one.get(0)
            self.op_k.b.trans_op(-1.0, x_y, 0.0, work_one)
            m_by = work_one.get(0)
        finally:
            work_one.drop()
        cri_unbdd = L.norm(p) * norm_c / m_cx if m_cx > par.eps_zero else math.inf
        cri_infeas = L.norm(d) * norm_b / m_by if m_by > par.eps_zero else math.inf
        return cri_unbdd, cri_infeas
