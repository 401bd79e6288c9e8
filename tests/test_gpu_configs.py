"""GPU: BASELINE.json's configs[3] and configs[4] at their FULL sizes.

configs[4] = dense LP n = 200 000, m = 400 000 row-partitioned over 8 GPUs: one rank's shard is 50 000 x 200 000 f32
(40 GB).  A 1-GPU box holds any ONE of the eight shards, so the per-rank product path is checked shard by shard through
size-independent properties (exact answers on the -I rows, regenerated rows / columns of the counter-based generator,
adjointness, linearity, agreement of the schedules), and the 8-way sharded solve is checked against the oracle at
reduced n with the shards keeping C5's 1 : 4 aspect ratio.

configs[3] = SDP with one PSD cone of order 500 (n = 2000, A 125 250 x 2000): iterates of the full-size instance
against the oracle's snapshots (Householder + QL at k = 500 on the CPU), and the unaligned-lda GEMV shape."""
import numpy as np
import pytest

import oracle as O
from problems import benchmark_lp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import totsu_amd
    from totsu_amd import _lib
    _lib.init()
    return totsu_amd


N5, WORLD5 = 200_000, 8


@pytest.mark.parametrize("rank", [0, 7])
def test_c5_shard_products(T, rank):
    """rank 0 holds rows 0 .. 50 000 of G = [-I ; U(0,1)] (all inside the -I block: the products are known EXACTLY);
    rank 7 holds the last 50 000 dense rows (entries regenerated from the counter-based generator in f64)."""
    from totsu_amd import synth
    from totsu_amd._lib import lib
    D = T.DeviceBuffer
    inst = synth.LpInstance(N5, seed=0, rank=rank, world=WORLD5)
    m, n, m_total = inst.m, inst.n, inst.m_total
    assert (m, n) == (50_000, 200_000) and inst.r0 == rank * 50_000
    x, y = D(n), D(m)
    lib.thip_gen_vector(x.ptr, n, 0, 2, 0, 1, 1.0, 0.0)
    lib.thip_gen_vector(y.ptr, m, 0, 3, 0, 1, 1.0, 0.0)
    Ax, Aty = D(m), D(n)
    lib.thip_transform_ge(0, m, n, 1.0, inst.mat_a.ptr, x.ptr, 0.0, Ax.ptr)
    lib.thip_transform_ge(1, m, n, 1.0, inst.mat_a.ptr, y.ptr, 0.0, Aty.ptr)
    hx, hy = x.to_host(), y.to_host()
    hAx, hAty = Ax.to_host(), Aty.to_host()
    if rank == 0:
        # A = [-I_50000 | 0]: A x = -x[:m], A^T y = [-y ; 0] -- bit-exact (every other term is an exact zero)
        assert np.array_equal(hAx, -hx[:m])
        assert np.array_equal(hAty[:m], -hy) and not hAty[m:].any()
    else:
        hx64, hy64 = hx.astype(np.float64), hy.astype(np.float64)
        for r in (0, 1, 31_337, m - 1):
            row = np.array([O.rng_uniform(0, synth.STREAM_A, (inst.r0 + r) + cc * m_total) for cc in range(n)])
            assert abs(hAx[r] - row @ hx64) <= 2e-5 * (np.abs(row) @ np.abs(hx64))
        for cc in (0, 77_777, n - 1):
            col = np.array([O.rng_uniform(0, synth.STREAM_A, (inst.r0 + r) + cc * m_total) for r in range(m)])
            assert abs(hAty[cc] - col @ hy64) <= 2e-5 * (np.abs(col) @ np.abs(hy64))
    # adjointness <A x, y> == <x, A^T y> in f64 of the f32 results
    lhs = hAx.astype(np.float64) @ hy.astype(np.float64)
    rhs = hx.astype(np.float64) @ hAty.astype(np.float64)
    assert abs(lhs - rhs) <= 1e-5 * (np.abs(hAx).astype(np.float64) @ np.abs(hy).astype(np.float64))
    # linearity and beta accumulation: 2 A x - A x == A x
    lib.thip_transform_ge(0, m, n, 2.0, inst.mat_a.ptr, x.ptr, -1.0, Ax.ptr)
    assert np.allclose(Ax.to_host(), hAx, rtol=1e-5, atol=1e-3)
    for d in (x, y, Ax, Aty):
        d.free()
    inst.free()


def test_c5_shard_iteration_schedules_agree(T):
    """five iterations of the device loop on rank 7's 40 GB shard with the all-reduce hook installed (one rank: the sum
    over ranks is the identity, the hook placement and tail scalars are the sharded code path): the 2-pass carried
    schedule and the reference's 6-GEMV schedule must produce the same iterate"""
    from totsu_amd import synth
    inst = synth.LpInstance(N5, seed=0, rank=7, world=WORLD5)
    p = T.SolverParam()
    p.eps_acc, p.eps_inf, p.max_iter = 0.0, 0.0, None
    calls = []

    def hook(ctx, ptr, cnt, stream):
        calls.append(cnt)
        return 0

    its = {}
    for sched in ("reference", "carried"):
        fs = T.FusedSolver(inst.n, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, sched,
                           allreduce=hook)
        r = fs.run(5, poll_every=5)
        assert r.state == -1 and r.iters == 5 and np.isfinite(r.cri[0]) and np.isfinite(r.tau)
        its[sched] = fs.iterate()
        fs.destroy()
    # the A^T partial with the block partials of the sharded sums in its tail (n + 4 * 256), and the one at init
    # (|A| column sums + 2 scalars): nothing else is exchanged
    assert set(calls) == {inst.n + 2, inst.n + 1024}, set(calls)
    for a, b in zip(its["reference"], its["carried"]):
        sc = max(np.abs(a).max(), 1e-6)
        assert np.abs(a - b).max() <= 5e-5 * sc, np.abs(a - b).max() / sc
    inst.free()


def test_c5_column_shard_through_the_full_loop(T):
    """rank 0's COLUMN shard of configs[4] (400 000 x 25 000 f32 = 40 GB; LP n = 200 000 over 8 GPUs) through the WHOLE device
    loop -- plan, preconditioner, 55 iterations of sweep + m-tail + termination test -- with a stand-in collective of 30 us
    (bench.py --emulate-world 8: the sum over one rank is the identity; the hook's placement, the message and the fault flag
    in its tail are the sharded code path).  Twice: the rate is the shard's, not an accident of one run.  The line goes to
    gpurun_out/ for profiles/r05_c5_column_shard_full_loop.json"""
    import json
    import os
    import subprocess
    import sys
    import psutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = []
    for rep in range(2):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "lp", "--size", str(N5), "--emulate-world",
                            str(WORLD5), "--emulate-latency", "30", "--steps", "50", "--warmup", "5"], capture_output=True, text=True,
                           timeout=1200, cwd=root)
        assert r.returncode == 0, r.stderr[-3000:]
        out = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(out) == 1
        lines.append(json.loads(out[0]))
    d = lines[0]
    cfg = d["config"]
    assert cfg["schedule"] == "sweep" and cfg["passes_over_A_per_iter"] == 1 and cfg["emulated_world"] == WORLD5
    assert cfg["rows_per_gpu"] == 2 * N5 and cfg["cols_per_gpu"] == N5 // WORLD5
    assert cfg["hbm_plan"]["A_bytes"] == 4 * 2 * N5 * (N5 // WORLD5) and cfg["hbm_plan"]["fits"]
    assert d["sweep_faults"]["faults"] == 0
    assert d["roofline"]["passes_over_A_per_iter"] == 1 and d["roofline"]["frac"] > 0.70, d["roofline"]
    assert d["roofline"]["launches_timed"] == 50          # (50 steps: every launch is timed)
    # one iteration = one sweep of 40 GB + the m-tail + the 30 us stand-in: well under 8 ms, well over the bare sweep
    assert 5.0 < d["ms_per_step"] < 8.0, d["ms_per_step"]
    assert abs(lines[1]["ms_per_step"] / d["ms_per_step"] - 1) < 0.05
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    json.dump(d, open(os.path.join(root, "gpurun_out", "c5_column_shard_full_loop.json"), "w"))


def test_c5_aspect_eight_emulated_ranks_vs_oracle(T):
    """the benchmark_lp construction at n = 96 split over EIGHT emulated ranks (shards of 24 x 96: C5's 1 : 4 aspect
    ratio; ranks 0-3 hold -I rows, ranks 4-7 dense rows, as at full size) against the oracle's solve"""
    from test_gpu_sharded import _mb, _run_sharded
    n = 96
    c, G, h = benchmark_lp(n, seed=8)
    lp = T.ProbLP(_mb(T, T.MatType.General(n, 1)).set_array(c.reshape(-1, 1)), _mb(T, T.MatType.General(2 * n, n)).set_array(G),
                  _mb(T, T.MatType.General(2 * n, 1)).set_array(h.reshape(-1, 1)), _mb(T, T.MatType.General(0, n)),
                  _mb(T, T.MatType.General(0, 1)))
    dense = lp.dense()
    m = dense.m
    A = dense.mat_a.reshape((n, m)).T
    rows = m // 8
    parts = [dict(n=n, m=rows, mat_a=np.asfortranarray(A[k * rows:(k + 1) * rows]).ravel(order="F"),
                  vec_b=dense.vec_b[k * rows:(k + 1) * rows], vec_c=dense.vec_c, seg_type=[1], seg_len=[rows], rowabs=None)
             for k in range(8)]
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 400_000, 1e-4
    out = _run_sharded(T, parts, p, "carried")
    ro = O.solve_matop_cones(O.param(max_iter=400000, eps_acc=1e-4), dense.vec_c, dense.mat_a, dense.vec_b,
                             dense.seg_type, dense.seg_len)
    assert ro.status == O.OK
    assert all(o[0].state == 0 for o in out)
    assert len({o[0].iters for o in out}) == 1                     # every rank stops at the same iteration
    assert abs(out[0][0].iters - ro.iters) <= 0.02 * ro.iters + 5
    for o in out[1:]:
        assert np.array_equal(o[1], out[0][1])                     # replicated x is bitwise identical on all ranks
    x = out[0][1].astype(np.float64)
    y = np.concatenate([o[2] for o in out]).astype(np.float64)
    pobj = float(dense.vec_c.astype(np.float64) @ ro.x)
    assert abs(float(dense.vec_c.astype(np.float64) @ x) - pobj) <= 1e-4 * (1 + abs(pobj))
    assert np.allclose(x, ro.x, rtol=0, atol=2e-3 * max(1.0, np.abs(ro.x).max()))
    assert np.allclose(y, ro.y, rtol=0, atol=2e-3 * max(1.0, np.abs(ro.y).max()))
    lp.drop()


# ---- configs[3] at full size --------------------------------------------------------------------------------------

def test_c4_gemv_unaligned_lda_properties(T):
    """A of the k = 500 SDP: 125 250 x 2000, lda = 125 250 (not a multiple of 4)"""
    from totsu_amd._lib import lib
    D = T.DeviceBuffer
    m, n = 125_250, 2000
    A = D(n * m)
    lib.thip_gen_matrix(A.ptr, m, n, m, 0, 1, 0, 0, m, 0, 1.0, 0.0)
    x, y = D(n), D(m)
    lib.thip_gen_vector(x.ptr, n, 0, 2, 0, 1, 1.0, 0.0)
    lib.thip_gen_vector(y.ptr, m, 0, 3, 0, 1, 1.0, 0.0)
    Ax, Aty = D(m), D(n)
    lib.thip_transform_ge(0, m, n, 1.0, A.ptr, x.ptr, 0.0, Ax.ptr)
    lib.thip_transform_ge(1, m, n, 1.0, A.ptr, y.ptr, 0.0, Aty.ptr)
    hx, hy = x.to_host().astype(np.float64), y.to_host().astype(np.float64)
    hAx, hAty = Ax.to_host().astype(np.float64), Aty.to_host().astype(np.float64)
    assert abs(hAx @ hy - hx @ hAty) <= 1e-5 * (np.abs(hAx) @ np.abs(hy))
    for r in (0, 1, 2, 3, 62_501, m - 2, m - 1):
        row = np.array([O.rng_uniform(0, 1, r + cc * m) for cc in range(n)])
        assert abs(hAx[r] - row @ hx) <= 2e-5 * (np.abs(row) @ np.abs(hx))
    for cc in (0, 1, 999, n - 1):
        col = np.array([O.rng_uniform(0, 1, r + cc * m) for r in range(m)])
        assert abs(hAty[cc] - col @ hy) <= 2e-5 * (np.abs(col) @ np.abs(hy))
    for d in (A, x, y, Ax, Aty):
        d.free()


@pytest.mark.parametrize("schedule", ["fused", "carried", "sweep"])
def test_c4_full_size_sdp_iterates_vs_oracle(T, schedule):
    """the bench's configs[3] instance (k = 500, n = 2000, A 1 GB) for three iterations: preconditioner and iterates
    against the oracle (f64; its PSD projection is Householder + QL at k = 500)"""
    from totsu_amd import synth
    inst = synth.SdpInstance(2000, 500, seed=0)
    n, m = inst.n, inst.m
    a = inst.mat_a.to_host()[:m * n].astype(np.float64)
    b, c = inst.vec_b_host.astype(np.float64), inst.vec_c_host.astype(np.float64)
    iters = [0, 1, 2]
    k = O.num_threads()
    O.set_num_threads(max(k, min(64, (__import__("os").cpu_count() or 8))))      # 2 GB products: use the host's cores
    try:
        ro = O.solve_matop_cones(O.param(max_iter=5, eps_acc=1e-30), c, a, b, [O.CONE_PSD], [m], use_ql=True,
                                 snap_iters=iters, trace_cap=8)
    finally:
        O.set_num_threads(k)
    p = T.SolverParam()
    p.eps_acc = 1e-30
    fs = T.FusedSolver(n, m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, schedule)
    assert fs.schedule_in_use() == schedule          # "sweep": the one-pass kernel takes this size by itself
    t, s = fs.precond()
    N = n + 2 * m + 1
    assert np.allclose(t, ro.precond[:N], rtol=5e-5, atol=0)
    assert np.allclose(s, ro.precond[N:], rtol=5e-5, atol=0)
    done = 0
    for q, it in enumerate(iters):
        fs.run(it + 1 - done, poll_every=8)
        done = it + 1
        x, y = fs.iterate()
        rx, ry = ro.snaps[q][:N], ro.snaps[q][N:]
        sx, sy = max(np.abs(rx).max(), 1e-6), max(np.abs(ry).max(), 1e-6)
        assert np.abs(x - rx).max() <= 1e-4 * sx, (it, np.abs(x - rx).max() / sx)
        assert np.abs(y - ry).max() <= 1e-4 * sy, (it, np.abs(y - ry).max() / sy)
        tr = ro.trace[it]
        assert np.allclose(fs.status().cri, tr[2:], rtol=5e-3, atol=1e-5), (it, fs.status().cri, tr)
    fs.destroy()
    inst.free()


# ---- configs[2] at the full n ---------------------------------------------------------------------------------------

def _c3_vs_oracle(T, schedule, sub, want_members=None, iters=(0, 1, 2)):
    """the standalone problem made of the first `sub` of the 1000 cones of BASELINE configs[2] at its full n = 50 000 (sub =
    1000: the headline instance itself): preconditioner (solver.rs:496-524), iterates after iterations `iters`
    (solver.rs:526-571) and the criteria triple (solver.rs:573-612) against the f64 oracle on the SAME inputs.  The
    tolerance grows with the iteration like the toy-size snapshots' (tests/test_gpu_sweep.py::_check_sweep_iterates):
    f32 round-off of the iterate accumulates over the steps, the oracle's f64 does not"""
    import math
    import os
    from totsu_amd import synth
    from totsu_amd._lib import lib
    n, cones_full, ni = 50_000, 1000, 99
    rows = 1 + ni
    inst = synth.SocpInstance(n, cones_full, ni, seed=0, first_cones=sub)
    m = inst.m
    assert m == sub * rows and inst.m_total == cones_full * rows
    k = O.num_threads()
    O.set_num_threads(max(k, min(128, (os.cpu_count() or 8))))
    try:
        a = O.gen_matrix(m, n, 0, synth.STREAM_A, 0, 0, inst.m_total, 1, -1.0 / math.sqrt(n))
        # the device holds the same entries: whole columns 0, 17 and n - 1, bit for bit
        col = T.DeviceBuffer(m)
        for cc in (0, 17, n - 1):
            lib.thip_copy(m, inst.mat_a.ptr + 4 * cc * m, col.ptr)
            assert np.array_equal(col.to_host()[:m].astype(np.float64), np.asarray(a)[cc * m:(cc + 1) * m]), cc
        col.free()
        b, c = inst.vec_b_host.astype(np.float64), inst.vec_c_host.astype(np.float64)
        iters = list(iters)
        ro = O.solve_matop_cones(O.param(max_iter=iters[-1] + 3, eps_acc=1e-30), c, a, b, [O.CONE_SOC] * sub, [rows] * sub,
                                 snap_iters=iters, trace_cap=iters[-1] + 4)
    finally:
        O.set_num_threads(k)
    del a
    p = T.SolverParam()
    p.eps_acc = 1e-30
    # (the default geometry, no timing: the test pins WHICH kernel instance it checks)
    fs = T.FusedSolver(n, m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, schedule,
                       gemv_autotune=False if want_members else None)
    assert fs.schedule_in_use() == schedule          # "sweep": the one-pass kernel takes this size by itself
    if want_members:
        pl = fs.sweep_plan()
        assert (pl["workgroups_per_column_group"], pl["columns_per_panel"], pl["slots_per_thread"]) == want_members, pl
    t, s = fs.precond()
    N = n + 2 * m + 1
    assert np.allclose(t, ro.precond[:N], rtol=5e-5, atol=0), np.abs(t / ro.precond[:N] - 1).max()
    assert np.allclose(s, ro.precond[N:], rtol=5e-5, atol=0), np.abs(s / ro.precond[N:] - 1).max()
    done = 0
    for q, it in enumerate(iters):
        fs.run(it + 1 - done, poll_every=8)
        done = it + 1
        x, y = fs.iterate()
        rx, ry = ro.snaps[q][:N], ro.snaps[q][N:]
        sx, sy = max(np.abs(rx).max(), 1e-6), max(np.abs(ry).max(), 1e-6)
        tol = 1e-4 if it < 9 else (3e-4 if it < 99 else 2e-3)
        assert np.abs(x - rx).max() <= tol * sx, (it, np.abs(x - rx).max() / sx)
        assert np.abs(y - ry).max() <= tol * sy, (it, np.abs(y - ry).max() / sy)
        tr = ro.trace[it]
        assert np.allclose(fs.status().cri, tr[2:], rtol=max(5e-3, 50 * tol), atol=1e-5), (it, fs.status().cri, tr)
    fs.destroy()
    inst.free()


def test_c3_first_328_cones_at_full_n_through_sweep_to_iteration_99(T):
    """the chain of evidence of the headline schedule at the headline's column count: the one-pass schedule's iterates 0, 1, 2,
    9 and 99 of the 328-cone sub-instance (A_sub 32 800 x 50 000) against the f64 oracle's -- the toy-size snapshots of
    tests/test_gpu_sweep.py at n = 50 000 (the oracle needs ~ 100 iterations of 6 products over 13 GB of f64: minutes on
    the GPU box's cores)"""
    _c3_vs_oracle(T, "sweep", 328, iters=(0, 1, 2, 9, 99))


@pytest.mark.parametrize("schedule", ["fused", "carried", "sweep"])
def test_c3_first_328_cones_at_full_n_iterates_vs_oracle(T, schedule):
    """A_sub 32 800 x 50 000, 6.6 GB f32 on the GPU, 13 GB f64 in the oracle -- the sub-instance bench.py's cpu_baseline leg
    times: the HIP path compared with the restated reference at the headline's column count, not with itself."""
    _c3_vs_oracle(T, schedule, 328)


def test_c3_full_size_sweep_vs_oracle(T):
    """BASELINE configs[2] ITSELF -- 1000 cones, m = 100 000, n = 50 000, A 20 GB f32 on the GPU -- through the one-pass
    schedule in the geometry of the headline line (8 workgroups per column group x 12 500 rows, 7 slots per thread,
    sweep_k<7,1,2,1,3>) against the f64 oracle, whose A is 40 GB: needs a host with that much memory to spare."""
    import psutil
    free = psutil.virtual_memory().available
    if free < 64 * 2 ** 30:
        pytest.skip("the oracle's f64 copy of A is 40 GB; only %.0f GiB of host memory available" % (free / 2 ** 30))
    _c3_vs_oracle(T, "sweep", 1000, want_members=(8, 1, 7), iters=(0, 1, 2, 9))


def test_c5_column_shard_through_the_one_pass_kernel(T):
    """rank 7's COLUMN shard of configs[4] (LP n = 200 000, m = 400 000 over 8 GPUs: 400 000 x 25 000 f32 = 40 GB) through
    sweep_k alone (thip_test_sweep): the geometry with 32 workgroups per column x 7 slots per thread that only this
    height reaches.  Checked against entries regenerated from the counter-based generator in f64: whole columns (both
    dots -> u, x_x, gP) and whole rows (both axpys -> A u, A x_x)."""
    import ctypes as C
    from totsu_amd import _lib, synth
    lib = _lib.lib
    D = T.DeviceBuffer
    inst = synth.LpInstanceCols(N5, seed=0, rank=7, world=WORLD5)
    m, nl, col0 = inst.m, inst.n_local, inst.col0
    assert (m, nl, col0) == (400_000, 25_000, 175_000)
    vec = {}
    for k, (ln, stream, kind, sc, sh) in dict(v=(m, 21, 1, 1.0, 0.0), xy=(m, 22, 1, 1.0, 0.0), c=(nl, 23, 1, 1.0, 0.0),
                                              su=(nl, 24, 0, 1e-5, 1e-6), tx=(nl, 25, 0, 1e-5, 1e-6), u=(nl, 26, 1, 1.0, 0.0),
                                              xx=(nl, 27, 1, 1.0, 0.0), gp=(nl, 28, 1, 1.0, 0.0)).items():
        vec[k] = D(ln)
        lib.thip_gen_vector(vec[k].ptr, ln, 5, stream, 0, kind, sc, sh)
    host = {k: d.to_host().astype(np.float64) for k, d in vec.items()}
    outs = {k: D(ln, zero=True) for k, ln in dict(xx_out=nl, hn=m, h3=m).items()}
    t = _lib.SweepTest()
    t.m, t.n, t.lda = m, nl, m
    t.mat_a, t.v, t.xy, t.c, t.su, t.tx = inst.mat_a.ptr, vec["v"].ptr, vec["xy"].ptr, vec["c"].ptr, vec["su"].ptr, vec["tx"].ptr
    t.u, t.ku, t.xx_in, t.kx_in, t.xx_out, t.kx_out = vec["u"].ptr, None, vec["xx"].ptr, None, outs["xx_out"].ptr, None
    t.gp, t.hn, t.h3 = vec["gp"].ptr, outs["hn"].ptr, outs["h3"].ptr
    kappa, rtau = -0.37, 0.81
    t.kappa, t.rtau, t.first, t.reps = kappa, rtau, 0, 1
    ms, info = (C.c_float * 2)(), (C.c_int * 8)()
    lib.thip_test_sweep(C.byref(t), ms, info)
    assert info[0] == 0, "the kernel raised its error word: %d" % info[0]
    assert info[1] == 32 and info[2] == 8, list(info)          # 32 workgroups per column, 8 column groups
    u_new, x_new, g3 = vec["u"].to_host().astype(np.float64), outs["xx_out"].to_host().astype(np.float64), vec["gp"].to_host().astype(np.float64)
    hn, h3 = outs["hn"].to_host().astype(np.float64), outs["h3"].to_host().astype(np.float64)
    n_glob = N5

    def entry_col(cl):
        col = np.array([O.rng_uniform(0, synth.STREAM_A, r + (col0 + cl) * m) for r in range(n_glob, m)])
        full = np.zeros(m)
        full[n_glob:] = col
        full[col0 + cl] = -1.0           # the -I block: row == global column
        return full

    # columns from different groups (8 groups of 3125 columns), first / last / interior
    for cl in (0, 3124, 3125, 12_345, nl - 1):
        a = entry_col(cl)
        dT, d3 = a @ host["v"], a @ host["xy"]
        sc = np.abs(a) @ np.abs(host["v"])
        assert abs(g3[cl] - d3) <= 2e-5 * (np.abs(a) @ np.abs(host["xy"])), cl
        u_ref = host["u"][cl] + host["su"][cl] * (-(host["gp"][cl] - 2 * d3) - host["c"][cl] * rtau)
        x_ref = host["xx"][cl] + host["tx"][cl] * (dT + host["c"][cl] * kappa)
        assert abs(u_new[cl] - u_ref) <= 1e-6 * abs(u_ref) + 2e-5 * host["su"][cl] * 2 * (np.abs(a) @ np.abs(host["xy"])), cl
        assert abs(x_new[cl] - x_ref) <= 1e-6 * abs(x_ref) + 2e-5 * host["tx"][cl] * sc, cl
    # rows owned by different members (32 members of 12 500 rows): a -I row (exact), dense rows
    for r in (col0 + 17, n_glob, n_glob + 12_499, n_glob + 12_500, 333_333, m - 1):
        if r < n_glob:
            cl = r - col0
            assert abs(hn[r] + u_new[cl]) <= 1e-6 * abs(u_new[cl]) and abs(h3[r] + x_new[cl]) <= 1e-6 * abs(x_new[cl])
            continue
        row = np.array([O.rng_uniform(0, synth.STREAM_A, r + (col0 + cl) * m) for cl in range(nl)])
        assert abs(hn[r] - row @ u_new) <= 2e-5 * (np.abs(row) @ np.abs(u_new)), r
        assert abs(h3[r] - row @ x_new) <= 2e-5 * (np.abs(row) @ np.abs(x_new)), r
    # -I rows that belong to other ranks' columns are zero rows of this shard
    assert not hn[:col0].any() and not h3[:col0].any()
    for d in list(vec.values()) + list(outs.values()):
        d.free()
    inst.free()


# ---- configs[1] at full size ----------------------------------------------------------------------------------------

@pytest.mark.parametrize("schedule", ["reference", "carried", "sweep"])
def test_c2_full_size_lp_iterates_vs_oracle(T, schedule):
    """BASELINE configs[1]: the benchmark_lp construction at n = 10 000, m = 20 000 (A 0.8 GB f32 on the GPU, 1.6 GB f64 in
    the oracle, the device's own entries widened): preconditioner, iterates after iterations 0, 1, 2, 9 and the criteria
    against the f64 oracle -- the reference's six-GEMV sequence and the 2-pass carried schedule"""
    import os
    from totsu_amd import synth
    inst = synth.LpInstance(10_000, seed=0)
    n, m = inst.n, inst.m
    a = inst.mat_a.to_host()[:m * n].astype(np.float64)
    b, c = inst.vec_b_host.astype(np.float64), inst.vec_c_host.astype(np.float64)
    iters = [0, 1, 2, 9]
    k = O.num_threads()
    O.set_num_threads(max(k, min(64, (os.cpu_count() or 8))))
    try:
        ro = O.solve_matop_cones(O.param(max_iter=12, eps_acc=1e-30), c, a, b, [O.CONE_RPOS], [m], snap_iters=iters, trace_cap=16)
    finally:
        O.set_num_threads(k)
    del a
    p = T.SolverParam()
    p.eps_acc = 1e-30
    fs = T.FusedSolver(n, m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, schedule)
    assert fs.schedule_in_use() == schedule          # "sweep": the one-pass kernel takes this size by itself
    t, s = fs.precond()
    N = n + 2 * m + 1
    assert np.allclose(t, ro.precond[:N], rtol=5e-5, atol=0), np.abs(t / ro.precond[:N] - 1).max()
    assert np.allclose(s, ro.precond[N:], rtol=5e-5, atol=0), np.abs(s / ro.precond[N:] - 1).max()
    done = 0
    for q, it in enumerate(iters):
        fs.run(it + 1 - done, poll_every=8)
        done = it + 1
        x, y = fs.iterate()
        rx, ry = ro.snaps[q][:N], ro.snaps[q][N:]
        sx, sy = max(np.abs(rx).max(), 1e-6), max(np.abs(ry).max(), 1e-6)
        assert np.abs(x - rx).max() <= 1e-4 * sx, (it, np.abs(x - rx).max() / sx)
        assert np.abs(y - ry).max() <= 1e-4 * sy, (it, np.abs(y - ry).max() / sy)
        tr = ro.trace[it]
        assert np.allclose(fs.status().cri, tr[2:], rtol=5e-3, atol=1e-5), (it, fs.status().cri, tr)
    fs.destroy()
    inst.free()


def test_bench_lines_of_the_lp_and_sdp_workloads_carry_their_gates_and_the_eig_record():
    """round 5: `bench.py --workload lp|sdp --to-eps …` lines carry objective_gate.this_run (the answer of the run re-evaluated
    in f64 against the regenerated A: kkt_f64_lp / kkt_f64_sdp), the SDP line the second roofline record `roofline_eig` (the PSD
    chain timed by HIP events, its flops, the fraction of the f32 matrix peak), and a `--mixed-leg` run its second time-to-eps
    record -- at sizes that take seconds"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def line(args):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu", "--steps", "20", "--warmup", "5"] + args,
                           capture_output=True, text=True, timeout=900, cwd=root)
        assert r.returncode == 0, r.stderr[-3000:]
        out = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(out) == 1
        return json.loads(out[0])
    d = line(["--workload", "sdp", "--k", "96", "--size", "300", "--to-eps", "1e-3"])
    e = d["roofline_eig"]
    assert e["bound"] == "mfma" and e["order"] == 96 and e["launches_per_projection_pair"] == 37 and e["spans_timed"] == 20
    assert 0.05 < e["ms_per_projection_pair"] < 1.0 and 0 < e["frac"] < 1 and 0 < e["share_of_iteration"] < 1
    g = d["objective_gate"]["this_run"]
    assert d["time_to_eps"]["state"] == 0 and "error" not in g, g
    assert g["dual_residual_rel_f64"] <= 1.1e-3 and g["gap_rel"] <= 1.1e-3 and g["primal_cone_violation_rel_to_norm_b"] <= 1e-4, g
    assert g["dual_cone_violation"] <= 1e-4
    assert d["config"]["hbm_plan"]["fits"] and d["config"]["hbm_plan"]["A_bytes"] == 4 * (96 * 97 // 2) * 300
    d = line(["--workload", "lp", "--size", "1500", "--to-eps", "1e-3", "--mixed-leg"])
    g = d["objective_gate"]["this_run"]
    assert d["time_to_eps"]["state"] == 0 and "error" not in g, g
    assert g["dual_residual_rel_f64"] <= 1.1e-3 and g["gap_rel"] <= 1.1e-3 and g["dual_cone_violation"] <= 1e-5, g
    assert d["roofline_eig"] is None
    mx = d["time_to_eps_mixed"]
    assert mx["state"] == 0 and mx["a_storage"] == "mixed" and "f16_phase" in mx
    assert mx["vs_f32_leg"]["primal_obj_rel_diff"] <= 1e-3
    assert "error" not in mx["objective_gate_this_run"]
