"""numpy (f64) restatement of the ONE-PASS schedule with column shards (totsu_amd/csrc/thip_sweep.hip + the sw_* kernels of
thip_solver.hip), for the CPU tests of the N > 1 path: the same recurrences as solver.rs:525-570 in the
skewed order the device evaluates them in, a rank holding a block of COLUMNS of A, the n-vectors sharded, the m-vectors
replicated, ONE all-reduce per iteration.  Test infrastructure (host logic), not a fallback of the product."""
import numpy as np

CONE_ZERO, CONE_RPOS, CONE_SOC = 0, 1, 2


def _proj_soc(z):                                   # cone_soc.rs:38-65
    t, nrm = z[0], np.linalg.norm(z[1:])
    if nrm <= -t:
        z[:] = 0.0
    elif nrm > t:
        a = (t + nrm) / 2
        z[1:] *= a / nrm
        z[0] = a


class SweepCols:
    def __init__(self, A_loc, b, c_loc, seg_type, seg_len, allreduce, eps_zero=1e-12):
        self.A, self.b, self.c, self.ar = np.asarray(A_loc, float), np.asarray(b, float), np.asarray(c_loc, float), allreduce
        self.seg = list(zip(seg_type, seg_len))
        m, nl = self.A.shape
        self.m, self.nl, self.ez = m, nl, eps_zero
        # calc_norms / calc_precond (solver.rs:460-524): the |A| row sums and the sums over c are completed by an all-reduce
        red = self.ar(np.concatenate([np.abs(self.A).sum(axis=1), [np.sum(self.c ** 2), np.abs(self.c).sum()]]))
        rowabs, c2, c1 = red[:m], red[m], red[m + 1]
        self.norm_b, self.norm_c = np.sqrt(np.sum(self.b ** 2)), np.sqrt(c2)
        inv = lambda t: 1.0 / np.maximum(t, eps_zero)
        self.Tx = self.Su = inv(np.abs(self.A).sum(axis=0) + np.abs(self.c))
        self.Ty, self.Ts, self.Sv = inv(rowabs + np.abs(self.b)), np.full(m, inv(1.0)), inv(rowabs + np.abs(self.b) + 1.0)
        self.t_tau = self.s_kappa = inv(c1 + np.abs(self.b).sum())
        off = 0
        for ty, ln in self.seg:                      # product_group (solver.rs:509-523)
            if ty >= CONE_SOC and ln:
                self.Ty[off:off + ln] = self.Ty[off:off + ln].min()
                self.Ts[off:off + ln] = self.Ts[off:off + ln].min()
            off += ln
        self.xx, self.u = np.zeros(nl), np.zeros(nl)
        self.xy, self.xs, self.v = np.zeros(m), np.zeros(m), np.zeros(m)
        self.tau, self.kappa, self.rtau = 1.0, 0.0, 0.0
        self.gP, self.hP = np.zeros(nl), np.zeros(m)
        self.first, self.iters, self.cri, self.collectives = True, 0, None, 1

    def _project(self, z, dual):
        off = 0
        for ty, ln in self.seg:
            s = z[off:off + ln]
            if ty == CONE_ZERO:
                if not dual:
                    s[:] = 0.0
            elif ty == CONE_RPOS:
                np.maximum(s, 0.0, out=s)
            elif ln:
                _proj_soc(s)
            off += ln

    def _sweep(self, first):
        gT, g3 = self.A.T @ self.v, self.A.T @ self.xy            # both dots of every column
        if not first:
            self.u = self.u + self.Su * (-(self.gP - 2 * g3) - self.c * self.rtau)
        self.gP = g3
        self.xx_next = self.xx + self.Tx * (gT + self.c * self.kappa)
        conv = self.tau > self.ez
        d = self.c + g3 / self.tau if conv else g3
        loc = np.concatenate([self.A @ self.u, self.A @ self.xx_next,
                              [d @ d, self.c @ self.xx, self.c @ self.u, self.c @ (self.xx - 2 * self.xx_next)]])
        red = self.ar(loc)                                      # THE all-reduce of the iteration
        self.collectives += 1
        m = self.m
        self.hN, self.h3 = red[:m], red[m:2 * m]
        self.dd, self.cx, self.cu, self.crx = red[2 * m:]

    def step(self):
        if self.first:
            self.bv = self.b @ self.v
            self._sweep(True)
            self.first = False
        b = self.b
        t_new = max(self.tau + self.t_tau * (-self.cu - self.bv), 0.0)           # solver.rs:551-552
        self.rtau, self.tau = self.tau - 2 * t_new, t_new
        oy, os_ = self.xy.copy(), self.xs.copy()
        self.xy = self.xy + self.Ty * (b * self.kappa - self.hN)
        self.xs = self.xs + self.Ts * self.v
        self._project(self.xy, True)
        self._project(self.xs, False)
        rxy, rxs = oy - 2 * self.xy, os_ - 2 * self.xs
        h2, self.hP = self.hP - 2 * self.h3, self.h3
        self.v = self.v + self.Sv * (h2 + rxs - b * self.rtau)
        self.bv = b @ self.v
        conv = self.tau > self.ez
        p = self.xs / self.tau - b + self.h3 / self.tau if conv else self.xs + self.h3
        pp, by = p @ p, b @ self.xy
        self.kappa = min(self.kappa + self.s_kappa * (self.crx + b @ rxy), 0.0)  # solver.rs:566-567
        self.xx = self.xx_next
        self._sweep(False)
        if conv:                                                                 # criteria_conv, solver.rs:573-612
            gx, gy = self.cx / self.tau, by / self.tau
            self.cri = (np.sqrt(pp) / (1 + self.norm_b), np.sqrt(self.dd) / (1 + self.norm_c),
                        abs(gx + gy) / (1 + abs(gx) + abs(gy)))
        self.iters += 1

    def iterate(self):
        """(x_x block, [x_y, x_s, tau]), (u block, [v, kappa]) of the consistent iterate"""
        return (self.xx.copy(), np.concatenate([self.xy, self.xs, [self.tau]])), (self.u.copy(), np.concatenate([self.v, [self.kappa]]))
