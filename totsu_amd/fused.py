"""`FusedSolver`: the device-resident conic iteration (thip_solver_* in include/totsu_f32hip.h) for problems whose
operators are dense matrices -- what `Solver::solve` (solver.rs:285-321) does, with every per-iteration step,
the projections and the termination test on the GPU.  Takes the stacked dense description produced by
`ProbLP/ProbSOCP/ProbSDP.dense()` or device buffers generated in place (bench)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib
from .solver import SolverError, SolverParam

SCHEDULES = {"reference": _lib.SCHED_REFERENCE, "fused": _lib.SCHED_FUSED, "carried": _lib.SCHED_CARRIED,
             "sweep": _lib.SCHED_SWEEP}


class DeviceBuffer:
    """float32 device array owned through the C ABI"""

    def __init__(self, n, zero=False):
        _lib.ensure_init()
        self.n = int(n)
        p = C.c_void_p()
        (lib.thip_alloc_zeroed if zero else lib.thip_alloc)(self.n, C.byref(p))
        self.ptr = p.value

    @staticmethod
    def from_host(a):
        a = np.ascontiguousarray(a, dtype=np.float32).ravel()
        d = DeviceBuffer(a.size)
        if a.size:
            lib.thip_h2d(d.ptr, a.ctypes.data, a.size)
        return d

    def to_host(self):
        out = np.empty(self.n, dtype=np.float32)
        if self.n:
            lib.thip_d2h(out.ctypes.data, self.ptr, self.n)
        return out

    def free(self):
        if self.ptr is not None:
            lib.thip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


A_STORAGE = {"f32": 0, "bf16": 1, "f16": 2}


class Bf16Matrix:
    """A dense m x n matrix held on the device in 16 bits per entry only (column-major, leading dimension ld16 = m
    rounded up to 8): built from f32 column blocks that are converted in place of being kept (thip_to_bf16 /
    thip_to_f16), so the f32 matrix never exists as a whole.  kind = "bf16", or "f16" (one power-of-two scale per
    column, 8x finer rounding).  Accepted by FusedSolver as `mat_a` (thip_solver_set_a_bf16 / _f16)."""

    def __init__(self, m, n, kind="bf16"):
        _lib.ensure_init()
        assert kind in ("bf16", "f16")
        self.m, self.n, self.kind = int(m), int(n), kind
        self.ld16 = (self.m + 7) // 8 * 8
        self._buf = DeviceBuffer((self.ld16 * self.n + 1) // 2 + 4)      # 2 elements per float slot
        self.ptr = self._buf.ptr
        self._inv = DeviceBuffer(max(self.n, 1)) if kind == "f16" else None

    @property
    def inv_ptr(self):
        return None if self._inv is None else self._inv.ptr

    def set_columns(self, c0, block, ncols):
        """columns c0 .. c0 + ncols of the matrix <- the f32 block (DeviceBuffer, m x ncols, lda = m)"""
        assert 0 <= c0 and c0 + ncols <= self.n and block.n >= self.m * ncols
        dst = self.ptr + 2 * self.ld16 * c0
        if self.kind == "f16":
            lib.thip_to_f16(self.m, ncols, block.ptr, dst, self.ld16, self._inv.ptr + 4 * c0)
        else:
            lib.thip_to_bf16(self.m, ncols, block.ptr, dst, self.ld16)

    @staticmethod
    def from_f32(mat, m, n, kind="bf16"):
        """mat: DeviceBuffer or host array, column-major m x n"""
        d = mat if isinstance(mat, DeviceBuffer) else DeviceBuffer.from_host(mat)
        out = Bf16Matrix(m, n, kind)
        out.set_columns(0, d, n)
        if d is not mat:
            d.free()
        return out

    def free(self):
        self._buf.free()
        if self._inv is not None:
            self._inv.free()
        self.ptr = None


STATE_ARITH = {"compensated": _lib.STATE_COMPENSATED, "plain": _lib.STATE_PLAIN}


def _c_param(p):
    """SolverParam -> thip_param.  `state_arith` is an attribute of the fused loop only (not in the reference's
    SolverParam, solver.rs:13-41): "compensated" (default) or "plain" f32 iterate updates."""
    return _lib.Param(-1 if p.max_iter is None else int(p.max_iter), p.eps_acc, p.eps_inf, p.eps_zero,
                      int(p.log_period), STATE_ARITH[getattr(p, "state_arith", "compensated")], 0)


class FusedResult:
    def __init__(self, st):
        self.state = st.state
        self.iters = st.iter
        self.kind = st.kind
        self.cri = (st.cri[0], st.cri[1], st.cri[2])
        self.tau, self.kappa = st.tau, st.kappa
        self.norm_b, self.norm_c = st.norm_b, st.norm_c


class FusedSolver:
    def __init__(self, n, m, mat_a, vec_b, vec_c, seg_type, seg_len, param=None, schedule="fused",
                 vec_b_rowabs=None, allreduce=None, a_storage="f32", overlap=None, gemv_autotune=None, lda_pad=None,
                 sweep_min_bytes=None, col_shard=False, sparse_two_copies=False):
        """mat_a / vec_b / vec_c / vec_b_rowabs: DeviceBuffer or host arrays (uploaded).
        a_storage: "f32" (the matrix as given), "bf16" or "f16" (a rounded 16-bit copy streamed at half the bytes; f16
        is column-scaled and rounds 8x finer; see set_a_storage / include/totsu_f32hip.h).
        allreduce: None (single GPU), "rccl" (native communicator set up with comm_init), or a Python callable
        (ctx, dev_ptr, count, stream) -> 0."""
        _lib.ensure_init()
        self.n, self.m = int(n), int(m)
        self._owned = []
        self._csr = None
        self._spt = None
        self._spt_owned = False
        from .sparse import SpTile
        if isinstance(mat_a, SpTile):                     # a tiled sparse operator built by the caller (kept by the caller)
            assert mat_a.shape == (self.m, self.n)
            self._spt = mat_a
            mat_a = DeviceBuffer(1)
            self._owned.append(mat_a)
        elif hasattr(mat_a, "tocsr"):
            # scipy.sparse matrix: ONE tiled copy serving both products (thip_sptile.hip); sparse_two_copies=True: the round-5
            # form, CSR of A and of A^T under the 2-pass schedules
            assert mat_a.shape == (self.m, self.n)
            if sparse_two_copies:
                from .sparse import _Csr
                self._csr = (_Csr(mat_a), _Csr(mat_a.T))
            else:
                self._spt = SpTile(mat_a)
                self._spt_owned = True
            mat_a = DeviceBuffer(1)
            self._owned.append(mat_a)
        self._a16 = mat_a if isinstance(mat_a, Bf16Matrix) else None
        if self._a16 is not None:
            assert (mat_a.m, mat_a.n) == (self.m, self.n)
            mat_a = DeviceBuffer(1)                       # no f32 matrix: thip_problem.mat_a is not read
            self._owned.append(mat_a)
        self.mat_a = self._dev(mat_a, 1 if (self._csr or self._spt or self._a16 is not None) else self.n * self.m)
        self.vec_b = self._dev(vec_b, self.m)
        self.vec_c = self._dev(vec_c, self.n)
        self.vec_b_rowabs = None if vec_b_rowabs is None else self._dev(vec_b_rowabs, self.m)
        self.param = param or SolverParam()
        self._st = np.ascontiguousarray(seg_type, dtype=np.int32)
        self._sl = np.ascontiguousarray(seg_len, dtype=np.int64)
        prob = _lib.Problem(self.n, self.m, None if self._a16 is not None else self.mat_a.ptr, self.vec_b.ptr, self.vec_c.ptr,
                            None if self.vec_b_rowabs is None else self.vec_b_rowabs.ptr, len(self._st),
                            self._st.ctypes.data_as(C.POINTER(C.c_int32)),
                            self._sl.ctypes.data_as(C.POINTER(C.c_int64)))
        par = _c_param(self.param)
        h = C.c_void_p()
        lib.thip_solver_create(C.byref(prob), C.byref(par), SCHEDULES[schedule], C.byref(h))
        self.h = h
        self.schedule = schedule
        if self._csr:
            a, at = self._csr
            lib.thip_solver_set_csr(self.h, a.nnz, a.rowptr.ptr, a.colidx.ptr, a.vals.ptr, at.rowptr.ptr, at.colidx.ptr,
                                    at.vals.ptr)
        if self._spt is not None:
            lib.thip_solver_set_sptile(self.h, self._spt.h)
        self._cb = None
        if allreduce == "rccl":
            lib.thip_solver_use_rccl(self.h)            # native RCCL on the library's stream (thip_comm_init first)
        elif allreduce == "oneshot":
            lib.thip_solver_use_oneshot(self.h)         # one-launch all-reduce over peer-mapped buffers (thip_oneshot_*)
        elif allreduce is not None and allreduce != "oneshot" and not (allreduce == "spin" or (isinstance(allreduce, tuple) and allreduce[0] == "spin")):
            self._cb = _lib.ALLREDUCE_FN(allreduce)
            lib.thip_solver_set_allreduce(self.h, self._cb, None)
        elif allreduce == "spin" or (isinstance(allreduce, tuple) and allreduce[0] == "spin"):
            # test hook: a stand-in collective that only takes time (thip_test_spin_allreduce), ("spin", microseconds)
            lib.thip_test_spin_allreduce(self.h, int(allreduce[1]) if isinstance(allreduce, tuple) else 0)
        if overlap is not None and allreduce is not None:
            # OFF by default (the library's default).  True / 1: all-reduce on the solver's side stream under the
            # local-row work; 2 / "pipeline": column-split pipeline; 3 / "pipeline-inorder": its in-order reference
            self.set_overlap(overlap)
        if gemv_autotune is not None:
            lib.thip_solver_set_gemv_autotune(self.h, 1 if gemv_autotune else 0)     # False: bit-reproducible across runs
        if lda_pad is not None:
            lib.thip_solver_set_lda_pad(self.h, int(lda_pad))
        if col_shard:
            # this rank's block of COLUMNS (n = its column count, vec_c its block, vec_b / cones the whole problem's); "sweep" only
            lib.thip_solver_set_column_shard(self.h, 1)
        if sweep_min_bytes is not None:
            lib.thip_solver_set_sweep_min_bytes(self.h, int(sweep_min_bytes))   # 0: "sweep" whenever the kernel takes the shape
        self.a_storage = "f32"
        if self._a16 is not None:
            if self._a16.kind == "f16":
                lib.thip_solver_set_a_f16(self.h, self._a16.ptr, self._a16.ld16, self._a16.inv_ptr)
            else:
                lib.thip_solver_set_a_bf16(self.h, self._a16.ptr, self._a16.ld16)
            self.a_storage = self._a16.kind
        elif a_storage != "f32":
            self.set_a_storage(a_storage)
        lib.thip_solver_init(self.h)

    OVERLAP = {False: 0, True: 1, 0: 0, 1: 1, 2: 2, 3: 3, "off": 0, "on": 1, "local-rows": 1, "pipeline": 2,
               "pipeline-inorder": 3}

    def set_overlap(self, mode):
        """thip_solver_set_overlap: 0 / "off", 1 / "on" (local-row work under the collective), 2 / "pipeline"
        (column-split pipeline), 3 / "pipeline-inorder" (bitwise reference of 2); allowed between run() calls"""
        lib.thip_solver_set_overlap(self.h, self.OVERLAP[mode])

    def overlap_info(self):
        mode, lpp, col = C.c_int(), C.c_int(), C.c_size_t()
        lib.thip_solver_overlap_info(self.h, C.byref(mode), C.byref(lpp), C.byref(col))
        return {"mode": mode.value, "launches_per_pass": lpp.value, "split_col": col.value}

    def set_spin_latency(self, us):
        """test hook (thip_test_spin_allreduce): the stand-in collective's latency in microseconds"""
        lib.thip_test_spin_allreduce(self.h, int(us))

    def reinit(self):
        """thip_solver_init again: a fresh solve of the same problem (x = 0, tau = 1)"""
        lib.thip_solver_init(self.h)

    def resume(self, param=None):
        """continue a solve that ended Converged / ExcessIter, optionally with new parameters"""
        if param is not None:
            self.param = param
            par = _c_param(param)
            lib.thip_solver_set_param(self.h, C.byref(par))
        lib.thip_solver_resume(self.h)

    def set_a_storage(self, kind):
        """Switch the stored form of the dense A ("f32" | "bf16" | "f16"); allowed between run() calls."""
        lib.thip_solver_set_a_storage(self.h, A_STORAGE[kind])
        self.a_storage = kind

    @staticmethod
    def from_dense(d, param=None, schedule="fused", a_storage="f32", **kw):
        return FusedSolver(d.n, d.m, d.mat_a, d.vec_b, d.vec_c, d.seg_type, d.seg_len, param, schedule,
                           d.vec_b_rowabs, a_storage=a_storage, **kw)

    def _dev(self, a, n):
        if isinstance(a, DeviceBuffer):
            assert a.n >= n
            return a
        d = DeviceBuffer.from_host(a)
        assert d.n == n, (d.n, n)
        self._owned.append(d)
        return d

    def run(self, max_steps=-1, poll_every=16):
        st = _lib.Status()
        lib.thip_solver_run(self.h, int(max_steps), int(poll_every), C.byref(st))
        return FusedResult(st)

    def status(self):
        st = _lib.Status()
        lib.thip_solver_status(self.h, C.byref(st))
        return FusedResult(st)

    def solution(self):
        x = np.empty(self.n, dtype=np.float32)
        y = np.empty(self.m, dtype=np.float32)
        lib.thip_solver_solution(self.h, x.ctypes.data, y.ctypes.data)
        return x, y

    def iterate(self):
        x = np.empty(self.n + 2 * self.m + 1, dtype=np.float32)
        y = np.empty(self.n + self.m + 1, dtype=np.float32)
        lib.thip_solver_iterate(self.h, x.ctypes.data, y.ctypes.data)
        return x, y

    def precond(self):
        t = np.empty(self.n + 2 * self.m + 1, dtype=np.float32)
        s = np.empty(self.n + self.m + 1, dtype=np.float32)
        lib.thip_solver_precond(self.h, t.ctypes.data, s.ctypes.data)
        return t, s

    def gemv_plan(self):
        nj, bl, ms = C.c_int(), C.c_int(), C.c_float()
        lib.thip_solver_gemv_plan(self.h, C.byref(nj), C.byref(bl), C.byref(ms))
        return {"rows_groups_per_lane": nj.value, "target_workgroups": bl.value, "autotune_ms": ms.value}

    def schedule_in_use(self):
        """the schedule the next run executes: "sweep" falls back to "carried" when the one-pass kernel cannot take the
        problem (thip_solver_schedule_in_use)"""
        v = C.c_int()
        lib.thip_solver_schedule_in_use(self.h, C.byref(v))
        return {v_: k for k, v_ in SCHEDULES.items()}[v.value]

    def sweep_plan(self):
        g, w, sl, ms = C.c_int(), C.c_int(), C.c_int(), C.c_float()
        lib.thip_solver_sweep_plan(self.h, C.byref(g), C.byref(w), C.byref(sl), C.byref(ms))
        return {"workgroups_per_column_group": g.value, "columns_per_panel": w.value, "slots_per_thread": sl.value,
                "autotune_ms_per_sweep": ms.value}

    def set_sweep_min_bytes(self, nbytes):
        lib.thip_solver_set_sweep_min_bytes(self.h, int(nbytes))

    def sweep_faults(self):
        """recoveries from a one-pass kernel that gave up (thip_solver_sweep_faults): how many in this solve, the kernel's
        error word of the last one, the iteration of the snapshot it went back to (-1: none)"""
        k, w, it = C.c_int(), C.c_int(), C.c_int64()
        lib.thip_solver_sweep_faults(self.h, C.byref(k), C.byref(w), C.byref(it))
        return {"faults": k.value, "last_word": w.value, "restored_iter": it.value}

    def inject_sweep_fault(self, kind, after_sweeps=0, spin_max=0):
        """TEST HOOK (thip_test_sweep_fault): kind 1 = the next plan's placement census fails; 7 = kind 2 in every sweep from then on;
        2 = one workgroup of the
        after_sweeps-th regular sweep from now withholds its partial dots; spin_max shortens the polling bound"""
        lib.thip_test_sweep_fault(self.h, int(kind), int(after_sweeps), int(spin_max))

    def set_sweep_publish(self, agent_scope):
        lib.thip_solver_set_sweep_publish(self.h, 1 if agent_scope else 0)

    def passes(self):
        p, b = C.c_int(), C.c_size_t()
        lib.thip_solver_passes(self.h, C.byref(p), C.byref(b))
        return p.value, b.value

    def solve(self, poll_every=16):
        """Solver::solve semantics: returns (x, y) or raises SolverError (solver.rs:285-321)."""
        r = self.run(-1, poll_every)
        if r.state != _lib.ST_OK:
            raise SolverError(r.state)
        return self.solution()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def destroy(self):
        if getattr(self, "h", None) is not None:
            lib.thip_solver_destroy(self.h)
            self.h = None
        for d in getattr(self, "_owned", []):
            d.free()
        self._owned = []
        if getattr(self, "_csr", None):
            for c in self._csr:
                c.free()
            self._csr = None
        if getattr(self, "_spt", None) is not None and self._spt_owned:
            self._spt.free()
        self._spt = None


def comm_init(rank, world, broadcast_bytes):
    """Creates the process's native RCCL communicator.  `broadcast_bytes(b: bytes | None) -> bytes` must return rank
    0's 128 bytes on every rank (e.g. through one torch.distributed broadcast)."""
    _lib.ensure_init()
    buf = (C.c_uint8 * 128)()
    if rank == 0:
        lib.thip_comm_unique_id(buf)
    data = broadcast_bytes(bytes(buf) if rank == 0 else None)
    buf2 = (C.c_uint8 * 128).from_buffer_copy(data)
    lib.thip_comm_init(rank, world, buf2)


def comm_destroy():
    lib.thip_comm_destroy()
