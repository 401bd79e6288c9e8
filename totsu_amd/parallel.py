"""Row-sharded solve at the trait level: one process per GPU, `A` split into contiguous
cone-aligned row blocks, every m-length vector sharded like the rows, every n-length vector replicated, and ONE
exchange per transposed product: a sum-all-reduce of the n-vector A_g^T y_g (plus the sharded scalars).

`ShardedSolver(L, comm)` is `Solver<L>` (totsu_core/src/solver/solver.rs) with the reductions that cross the
row split made collective; the arithmetic per rank is the reference's sequence of `L` calls on the local block.
`comm` wraps torch.distributed (backend nccl = RCCL on GPUs, gloo in the CPU tests).  The device-resident
counterpart is FusedSolver(..., allreduce=hook) (totsu_amd/fused.py)."""
import math

import numpy as np

from .linalg import splitm
from .solver import Solver, SolverError, _SelfDualEmbed, _SolverCore


def agree_on_column_shards(ok_local, allreduce_sum, world):
    """Every rank of a column-sharded one-pass run must be able to run the persistent kernel (thip_sweep_probe: 8 XCDs x 32
    CUs, a shape the kernel takes).  The ranks agree BEFORE anything column-sharded is built: `allreduce_sum` sums a float
    over the ranks; True only when every one of the `world` ranks said yes -- otherwise ALL of them take row shards and
    the 2-pass schedule (a rank that found out alone, inside thip_solver_init, would leave the others in their first
    all-reduce)."""
    total = float(allreduce_sum(np.array([1.0 if ok_local else 0.0], dtype=np.float32))[0])
    return int(round(total)) == int(world)


def shard_segments(seg_len, world, rank):
    """contiguous split of the cone segments into `world` blocks with balanced row counts; a cone never
    straddles a boundary.  Returns (first_segment, last_segment_exclusive, first_row, last_row_exclusive)."""
    tot = int(sum(seg_len))
    cum = np.concatenate([[0], np.cumsum(seg_len)])
    cuts = [0]
    for r in range(1, world):
        target = tot * r / world
        k = int(np.argmin(np.abs(cum - target)))
        cuts.append(max(k, cuts[-1]))
    cuts.append(len(seg_len))
    s0, s1 = cuts[rank], cuts[rank + 1]
    return s0, s1, int(cum[s0]), int(cum[s1])


class TorchComm:
    """sum-all-reduce over torch.distributed for slices of a LinAlg backend"""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_collectives = 0

    def allreduce_slice(self, L, sl):
        if sl.len() == 0:
            return
        if getattr(L, "name", "") == "F32HIP":
            class _CAI:
                pass
            o = _CAI()
            o.__cuda_array_interface__ = {"shape": (sl.len(),), "typestr": "<f4", "data": (sl.dev(), False), "version": 2}
            # trait-level path: order the collective by full synchronisation on both sides (the library launches on
            # its own stream; this path is host-driven and synchronises per scalar anyway)
            L.sync()
            t = self.torch.as_tensor(o, device="cuda")
            self.dist.all_reduce(t, group=self.group)
            self.torch.cuda.synchronize()
        else:
            t = self.torch.from_numpy(sl.get_mut())
            self.dist.all_reduce(t, group=self.group)
        self.n_collectives += 1

    def allreduce_scalar(self, v):
        t = self.torch.tensor([float(v)], dtype=self.torch.float64)
        if self.dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t, group=self.group)
        self.n_collectives += 1
        return float(t.item())


class _ShardedEmbed(_SelfDualEmbed):
    """SelfDualEmbed (solver.rs:45-184) over a local row block of A and b"""

    def __init__(self, L, c, a, b, comm):
        super().__init__(L, c, a, b)
        self.comm = comm
        self._one = np.zeros(1, dtype=L.F)

    def _b_dot(self, alpha, x_m):
        """alpha * b^T x_m summed over the row shards"""
        L = self.L
        t = L.Sl.new_mut(self._one)
        try:
            self.b.trans_op(alpha, x_m, 0.0, t)
            v = t.get(0)
        finally:
            t.drop()
        return self.comm.allreduce_scalar(v)

    def op(self, alpha, x, beta, y):                                # solver.rs:109-131
        assert beta == 0.0
        L = self.L
        m, n = self.a.size()
        x_x, x_y, x_s, x_tau = splitm(x, n, m, m, 1)
        y_n, y_m, y_1 = splitm(y, n, m, 1)
        self.a.trans_op(alpha, x_y, 0.0, y_n)
        self.comm.allreduce_slice(L, y_n)                           # sum_g A_g^T x_y
        self.c.op(alpha, x_tau, 1.0, y_n)
        self.a.op(-alpha, x_x, 0.0, y_m)
        L.add(-alpha, x_s, y_m)
        self.b.op(alpha, x_tau, 1.0, y_m)
        self.c.trans_op(-alpha, x_x, 0.0, y_1)
        y_1.set(0, y_1.get(0) + self._b_dot(-alpha, x_y))

    def trans_op(self, alpha, x, beta, y):                          # solver.rs:133-157
        assert beta == 0.0
        L = self.L
        m, n = self.a.size()
        x_n, x_m, x_1 = splitm(x, n, m, 1)
        y_x, y_y, y_s, y_tau = splitm(y, n, m, m, 1)
        self.a.trans_op(-alpha, x_m, 0.0, y_x)
        self.comm.allreduce_slice(L, y_x)
        self.c.op(-alpha, x_1, 1.0, y_x)
        self.a.op(alpha, x_n, 0.0, y_y)
        self.b.op(-alpha, x_1, 1.0, y_y)
        L.scale(0.0, y_s)
        L.add(-alpha, x_m, y_s)
        self.c.trans_op(alpha, x_n, 0.0, y_tau)
        y_tau.set(0, y_tau.get(0) + self._b_dot(alpha, x_m))

    def abssum(self, tau, sigma):                                   # solver.rs:159-183
        L = self.L
        m, n = self.a.size()
        L.scale(0.0, tau)
        tau_x, tau_y, tau_s, tau_tau = splitm(tau, n, m, m, 1)
        self.a.absadd_cols(tau_x)
        self.comm.allreduce_slice(L, tau_x)                         # column sums of |A| over all row blocks
        self.c.absadd_rows(tau_x)
        self.a.absadd_rows(tau_y)
        self.b.absadd_rows(tau_y)
        L.adds(1.0, tau_s)
        self.c.absadd_cols(tau_tau)
        loc = L.Sl.new_mut(self._one)
        try:
            L.scale(0.0, loc)
            self.b.absadd_cols(loc)
            bsum = loc.get(0)
        finally:
            loc.drop()
        tau_tau.set(0, tau_tau.get(0) + self.comm.allreduce_scalar(bsum))
        sigma_n, sigma_m, sigma_1 = splitm(sigma, n, m, 1)
        L.copy(tau_x, sigma_n)
        L.copy(tau_y, sigma_m)
        L.add(1.0, tau_s, sigma_m)
        L.copy(tau_tau, sigma_1)


class _ShardedCore(_SolverCore):
    def __init__(self, L, par, op_k, cone, trace, comm):
        super().__init__(L, par, op_k, cone, trace)
        self.comm = comm

    def _gnorm(self, sl):
        v = self.L.norm(sl)
        return math.sqrt(self.comm.allreduce_scalar(v * v))

    def calc_norms(self, work):                                     # solver.rs:460-481
        L = self.L
        work1 = np.zeros(1, dtype=L.F)
        work_one = L.Sl.new_mut(work1)
        try:
            mb = self.op_k.b.size()[0]
            (t,) = splitm(work, mb)
            nb = self.op_k.fr_norm(self.op_k.b, work_one, t)       # local block of b
            norm_b = math.sqrt(self.comm.allreduce_scalar(nb * nb))
            nc = self.op_k.c.size()[0]
            (t,) = splitm(work, nc)
            norm_c = self.op_k.fr_norm(self.op_k.c, work_one, t)
        finally:
            work_one.drop()
        return norm_b, norm_c

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
            self.op_k.a.trans_op(1.0 / val_tau, x_y, 0.0, d)
            self.comm.allreduce_slice(L, d)
            self.op_k.c.op(1.0, work_one, 1.0, d)
            self.op_k.c.trans_op(1.0 / val_tau, x_x, 0.0, work_one)
            g_x = work_one.get(0)
        finally:
            work_one.drop()
        g_y = self.op_k._b_dot(1.0 / val_tau, x_y)
        g = g_x + g_y
        cri_pri = self._gnorm(p) / (1.0 + norm_b)
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
            self.comm.allreduce_slice(L, d)
            self.op_k.c.trans_op(-1.0, x_x, 0.0, work_one)
            m_cx = work_one.get(0)
        finally:
            work_one.drop()
        m_by = self.op_k._b_dot(-1.0, x_y)
        cri_unbdd = self._gnorm(p) * norm_c / m_cx if m_cx > par.eps_zero else math.inf
        cri_infeas = L.norm(d) * norm_b / m_by if m_by > par.eps_zero else math.inf
        return cri_unbdd, cri_infeas


class ShardedSolver(Solver):
    """`Solver<L>` for a row-sharded problem: op_a / op_b / cone describe THIS rank's row block, op_c and all
    n-vectors are replicated.  Returns (x, y_local)."""

    def __init__(self, L, comm):
        super().__init__(L)
        self.comm = comm

    def solve(self, prob):
        op_c, op_a, op_b, cone, work = prob
        L = self.L
        m, n = op_a.size()
        if op_c.size() != (n, 1) or op_b.size() != (m, 1):
            raise SolverError(SolverError.InvalidOp)
        if Solver.query_worklen((m, n)) > len(work):
            raise SolverError(SolverError.WorkShortage)
        core = _ShardedCore(L, self.param, _ShardedEmbed(L, op_c, op_a, op_b, self.comm), cone, self.trace, self.comm)
        w = L.Sl.new_mut(work)
        try:
            err = core.solve(w)
        finally:
            w.drop()
        self.iters = core.iters
        if err is not None:
            raise SolverError(err)
        return work[:n], work[n:n + m]
