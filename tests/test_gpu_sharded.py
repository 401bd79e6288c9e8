"""GPU: the row-sharded fused loop.  Two ranks are emulated inside one process -- one thread per
rank, each driving its own FusedSolver over its cone-aligned row block, with an all-reduce hook that meets at a
barrier and sums the two device buffers on the shared stream.  This exercises exactly the product code that runs
under torch.distributed (hook placement, tail scalars, replicated / sharded vectors) and must reproduce the
unsharded solve: same status, same iteration count (+-1), same iterates to f32 round-off."""
import threading

import numpy as np
import pytest

import oracle as O
from problems import benchmark_lp, random_socp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import totsu_amd
    from totsu_amd import _lib
    _lib.init()
    return totsu_amd


def _split_rows(dense, T, cut_seg):
    """split the stacked problem after `cut_seg` cone segments"""
    n, m = dense.n, dense.m
    A = dense.mat_a.reshape((n, m)).T
    r = int(sum(dense.seg_len[:cut_seg]))
    parts = []
    for lo, hi, st, sl in ((0, r, dense.seg_type[:cut_seg], dense.seg_len[:cut_seg]),
                           (r, m, dense.seg_type[cut_seg:], dense.seg_len[cut_seg:])):
        parts.append(dict(n=n, m=hi - lo, mat_a=np.asfortranarray(A[lo:hi]).ravel(order="F"), vec_b=dense.vec_b[lo:hi],
                          vec_c=dense.vec_c, seg_type=list(st), seg_len=list(sl),
                          rowabs=None if dense.vec_b_rowabs is None else dense.vec_b_rowabs[lo:hi]))
    return parts


def _run_sharded(T, parts, param, schedule, max_steps=-1, overlap=None):
    from totsu_amd._lib import lib
    barrier = threading.Barrier(len(parts))
    bufs = [None] * len(parts)
    out = [None] * len(parts)
    errs = []

    def make_hook(rank):
        def hook(ctx, ptr, cnt, stream):
            try:
                bufs[rank] = ptr
                barrier.wait(timeout=60)
                if rank == 0:
                    for r in range(1, len(parts)):
                        lib.thip_add(cnt, 1.0, bufs[r], bufs[0])
                    for r in range(1, len(parts)):
                        lib.thip_copy(cnt, bufs[0], bufs[r])
                barrier.wait(timeout=60)
                return 0
            except Exception as e:      # noqa
                errs.append(e)
                return 1
        return hook

    def worker(rank):
        try:
            p = parts[rank]
            fs = T.FusedSolver(p["n"], p["m"], p["mat_a"], p["vec_b"], p["vec_c"], p["seg_type"], p["seg_len"], param,
                               schedule, vec_b_rowabs=p["rowabs"], allreduce=make_hook(rank), overlap=overlap)
            r = fs.run(max_steps, poll_every=32)
            x, y = fs.solution()
            out[rank] = (r, x, y, fs.iterate())
            fs.destroy()
        except Exception as e:          # noqa
            errs.append(e)
            barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(len(parts))]
    [t_.start() for t_ in th]
    [t_.join() for t_ in th]
    assert not errs, errs
    return out


def _mb(T, typ):
    return T.MatBuild(T.F32HIP, typ)


@pytest.mark.parametrize("schedule", ["reference", "fused", "carried"])
def test_sharded_socp_matches_unsharded(T, schedule):
    n, cones = 40, [9, 30, 0, 5, 64, 12]
    f, Gs, hs, cs, d = random_socp(n, cones, seed=5)
    socp = T.ProbSOCP(_mb(T, T.MatType.General(n, 1)).set_array(f.reshape(-1, 1)),
                      [_mb(T, T.MatType.General(g.shape[0], n)).set_array(g) for g in Gs],
                      [_mb(T, T.MatType.General(len(v), 1)).set_array(v.reshape(-1, 1)) for v in hs],
                      [_mb(T, T.MatType.General(n, 1)).set_array(v.reshape(-1, 1)) for v in cs], d,
                      _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)))
    dense = socp.dense()
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 200_000, 1e-4
    fs = T.FusedSolver.from_dense(dense, p, schedule)
    x1, y1 = fs.solve()
    r1 = fs.status()
    fs.destroy()
    parts = _split_rows(dense, T, 3)
    out = _run_sharded(T, parts, p, schedule)
    (ra, xa, ya, _), (rb, xb, yb, _) = out
    assert ra.state == rb.state == 0
    assert ra.iters == rb.iters and abs(ra.iters - r1.iters) <= max(2, 0.01 * r1.iters)
    assert np.array_equal(xa, xb)                         # replicated n-vectors stay bitwise identical across ranks
    assert np.allclose(xa, x1, rtol=2e-3, atol=2e-4)
    assert np.allclose(np.concatenate([ya, yb]), y1, rtol=2e-3, atol=2e-4)
    ro = O.solve_matop_cones(O.param(max_iter=200000, eps_acc=1e-4), dense.vec_c, dense.mat_a, dense.vec_b,
                             dense.seg_type, dense.seg_len)
    pobj = float(dense.vec_c.astype(np.float64) @ ro.x)
    assert abs(float(dense.vec_c.astype(np.float64) @ xa) - pobj) <= 1e-3 * (1 + abs(pobj))


@pytest.mark.parametrize("schedule", ["fused", "carried"])
def test_sharded_sparse_lp_matches_unsharded(T, schedule):
    """row shards of a SPARSE operator (each rank holds the tiled copy of its row block, thip_solver_set_sptile): the products come
    back as finished vectors, the all-reduce of A_g^T y_g and the sharded sums are the dense path's -- two emulated ranks reproduce
    the unsharded sparse solve and the oracle's objective"""
    import scipy.sparse as sp
    from problems import l1reg_lp
    c, G, h = l1reg_lp(30, seed=7)
    n, m = c.size, h.size
    A = sp.csr_matrix(G.astype(np.float32))
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 400_000, 1e-3
    fs = T.FusedSolver(n, m, A, h.astype(np.float32), c.astype(np.float32), [1], [m], p, schedule)
    x1, y1 = fs.solve()
    r1 = fs.status()
    fs.destroy()
    cut = 70
    parts = [dict(n=n, m=hi - lo, mat_a=A[lo:hi], vec_b=h[lo:hi].astype(np.float32), vec_c=c.astype(np.float32), seg_type=[1],
                  seg_len=[hi - lo], rowabs=None) for lo, hi in ((0, cut), (cut, m))]
    (ra, xa, ya, _), (rb, xb, yb, _) = _run_sharded(T, parts, p, schedule)
    assert ra.state == rb.state == 0 and ra.iters == rb.iters and abs(ra.iters - r1.iters) <= max(3, 0.02 * r1.iters)
    assert np.array_equal(xa, xb)
    ro = O.solve_lp(O.param(eps_acc=1e-3), c, G, h, np.zeros((0, n)), [])
    pobj = float(c @ ro.x)
    for x in (x1, xa):
        assert abs(float(c @ x.astype(np.float64)) - pobj) <= 1e-3 * (1 + abs(pobj))


def test_sharded_lp_first_iterates(T):
    c, G, h = benchmark_lp(48, seed=6)
    lp = T.ProbLP(_mb(T, T.MatType.General(48, 1)).set_array(c.reshape(-1, 1)), _mb(T, T.MatType.General(96, 48)).set_array(G),
                  _mb(T, T.MatType.General(96, 1)).set_array(h.reshape(-1, 1)), _mb(T, T.MatType.General(0, 48)),
                  _mb(T, T.MatType.General(0, 1)))
    dense = lp.dense()
    # nonneg cone is separable: split its rows 60 / 36 (+ the empty zero cone on the second shard)
    n, m = dense.n, dense.m
    A = dense.mat_a.reshape((n, m)).T
    parts = [dict(n=n, m=60, mat_a=np.asfortranarray(A[:60]).ravel(order="F"), vec_b=dense.vec_b[:60], vec_c=dense.vec_c,
                  seg_type=[1], seg_len=[60], rowabs=None),
             dict(n=n, m=36, mat_a=np.asfortranarray(A[60:]).ravel(order="F"), vec_b=dense.vec_b[60:], vec_c=dense.vec_c,
                  seg_type=[1, 0], seg_len=[36, 0], rowabs=None)]
    p = T.SolverParam()
    p.eps_acc = 0.0
    ro = O.solve_matop_cones(O.param(max_iter=52, eps_acc=1e-300), dense.vec_c, dense.mat_a, dense.vec_b, dense.seg_type,
                             dense.seg_len, snap_iters=[49], trace_cap=60)
    out = _run_sharded(T, parts, p, "carried", max_steps=50)
    N = n + 2 * m + 1
    rx, ry = ro.snaps[0][:N], ro.snaps[0][N:]
    (_, _, _, (xa, ya)), (_, _, _, (xb, yb)) = out
    x = np.concatenate([xa[:n], xa[n:n + 60], xb[n:n + 36], xa[n + 60:n + 120], xb[n + 36:n + 72], xa[-1:]])
    y = np.concatenate([ya[:n], ya[n:n + 60], yb[n:n + 36], ya[-1:]])
    assert np.abs(x - rx).max() <= 1e-3 * max(np.abs(rx).max(), 1e-6)
    assert np.abs(y - ry).max() <= 1e-3 * max(np.abs(ry).max(), 1e-6)
    assert xa[-1] == xb[-1] and ya[-1] == yb[-1]          # tau, kappa replicated


def test_native_rccl_single_rank(T):
    # the native communicator at world size 1 (all a 1-GPU box can run): id -> init -> in-place sum -> solver hook
    import ctypes as C
    from totsu_amd._lib import lib
    from totsu_amd.fused import comm_destroy, comm_init
    comm_init(0, 1, lambda b: b)
    try:
        x = np.arange(1000, dtype=np.float32)
        d = T.DeviceBuffer.from_host(x)
        lib.thip_comm_allreduce(d.ptr, 1000)
        assert np.array_equal(d.to_host(), x)
        d.free()
        c, G, h = benchmark_lp(20, seed=3)
        lp = T.ProbLP(_mb(T, T.MatType.General(20, 1)).set_array(c.reshape(-1, 1)), _mb(T, T.MatType.General(40, 20)).set_array(G),
                      _mb(T, T.MatType.General(40, 1)).set_array(h.reshape(-1, 1)), _mb(T, T.MatType.General(0, 20)),
                      _mb(T, T.MatType.General(0, 1)))
        dn = lp.dense()
        p = T.SolverParam()
        p.eps_acc, p.max_iter = 1e-4, 100_000
        a = T.FusedSolver.from_dense(dn, p, "carried")
        xa, _ = a.solve()
        b = T.FusedSolver(dn.n, dn.m, dn.mat_a, dn.vec_b, dn.vec_c, dn.seg_type, dn.seg_len, p, "carried", allreduce="rccl")
        xb, _ = b.solve()
        assert a.status().iters == b.status().iters and np.array_equal(xa, xb)
        a.destroy()
        b.destroy()
    finally:
        comm_destroy()


def test_bench_two_processes_share_one_gpu_and_reproduce_single_process():
    # bench.py under torch.distributed.run with 2 ranks on ONE GPU (all-reduce staged through gloo, since RCCL refuses
    # two ranks per device): the row-sharded run must reproduce the single-process run -- same iteration count to
    # eps (+-1 %), same primal / dual objective -- and print exactly one JSON line from rank 0.
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--size", "1500", "--cones", "30", "--steps", "5", "--warmup", "1", "--no-cpu", "--to-eps", "1e-3"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True,
                        timeout=900, env=env, cwd=root)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", "29577", os.path.join(root, "bench.py"),
                         "--gpus", "2", "--collective", "gloo"] + common, capture_output=True, text=True, timeout=900,
                        env=env, cwd=root)
    assert r2.returncode == 0, r2.stderr[-2000:]
    lines1 = [l for l in r1.stdout.splitlines() if l.strip()]
    lines2 = [l for l in r2.stdout.splitlines() if l.strip()]
    assert len(lines1) == 1 and len(lines2) == 1, (lines1, lines2)
    d1, d2 = json.loads(lines1[0]), json.loads(lines2[0])
    assert d2["n_gpus"] == 2 and d2["config"]["rows_per_gpu"] == 1500
    t1, t2 = d1["time_to_eps"], d2["time_to_eps"]
    assert t1["state"] == t2["state"] == 0
    assert abs(t1["iterations"] - t2["iterations"]) <= max(3, 0.01 * t1["iterations"]), (t1, t2)
    assert abs(t1["primal_obj"] - t2["primal_obj"]) <= 1e-4 * (1 + abs(t1["primal_obj"]))
    assert abs(t1["dual_obj"] - t2["dual_obj"]) <= 1e-4 * (1 + abs(t1["dual_obj"]))


def test_bench_gpus_flag_spawns_the_ranks_itself():
    # the driver's own command: `python bench.py --gpus N ...` with no launcher and no WORLD_SIZE must start the N ranks
    # itself (the flag was parsed and ignored).  On this 1-GPU box the two ranks share the GPU, so the
    # all-reduce falls back to the gloo-staged transport by itself; the line must say n_gpus == 2 and reproduce the
    # 1-rank objectives, and the f64 re-evaluation of THIS run's answer must be in the line for both.
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--size", "2000", "--cones", "40", "--steps", "5", "--warmup", "1", "--no-cpu", "--to-eps", "1e-3"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, capture_output=True,
                        text=True, timeout=900, env=env, cwd=root)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--collective", "gloo"] + common,
                        capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r2.returncode == 0, r2.stderr[-2000:]
    lines1 = [l for l in r1.stdout.splitlines() if l.strip()]
    lines2 = [l for l in r2.stdout.splitlines() if l.strip()]
    assert len(lines1) == 1 and len(lines2) == 1, (lines1, lines2)
    d1, d2 = json.loads(lines1[0]), json.loads(lines2[0])
    assert d1["n_gpus"] == 1 and d2["n_gpus"] == 2 and d2["physical_gpus"] == 1
    assert d2["config"]["rows_per_gpu"] == 2000 and "gloo" in d2["config"]["collective"]
    t1, t2 = d1["time_to_eps"], d2["time_to_eps"]
    assert t1["state"] == t2["state"] == 0
    assert abs(t1["iterations"] - t2["iterations"]) <= max(3, 0.01 * t1["iterations"]), (t1, t2)
    assert abs(t1["primal_obj"] - t2["primal_obj"]) <= 1e-4 * (1 + abs(t1["primal_obj"]))
    assert abs(t1["dual_obj"] - t2["dual_obj"]) <= 1e-4 * (1 + abs(t1["dual_obj"]))
    for d in (d1, d2):
        g = d["objective_gate"]["this_run"]
        assert g is not None and "error" not in g, g
        # the f64 re-evaluation agrees with the solver's own f32 criteria: dual residual at eps, primal inside the cone
        assert g["dual_residual_rel_f64"] <= 1.05e-3 and g["primal_cone_violation_rel_to_norm_b"] <= 1e-5, g
        assert g["gap_rel"] <= 1e-3, g
        assert abs(g["primal_obj_f64"] - d["time_to_eps"]["primal_obj"]) <= 1e-5 * (1 + abs(g["primal_obj_f64"]))
    g1, g2 = d1["objective_gate"]["this_run"], d2["objective_gate"]["this_run"]
    assert abs(g1["primal_obj_f64"] - g2["primal_obj_f64"]) <= 1e-4 * (1 + abs(g1["primal_obj_f64"]))
    # without --collective the shared GPU is detected and the transport switched (the driver passes no such flag)
    r3 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--size", "1500", "--cones", "30",
                         "--steps", "3", "--warmup", "1", "--no-cpu", "--no-to-eps"], capture_output=True, text=True,
                        timeout=900, env=env, cwd=root)
    assert r3.returncode == 0, r3.stderr[-2000:]
    d3 = json.loads([l for l in r3.stdout.splitlines() if l.strip()][0])
    assert d3["n_gpus"] == 2 and "gloo" in d3["config"]["collective"] and d3["rccl_ranks"] is None


def test_bench_prints_one_json_line_with_native_rccl_in_the_loop():
    # RCCL writes a version banner through C stdio when a communicator is created; it must not land on stdout next
    # to the ONE JSON line of the bench contract (it used to: flushed from libc's buffer at exit)
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29579")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--size", "1500", "--cones", "30", "--steps", "5",
                        "--warmup", "1", "--no-cpu", "--force-collective"], capture_output=True, text=True, timeout=900,
                       env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert "RCCL" in d["config"]["collective"] and d["n_gpus"] == 1
    assert d["rccl_ranks"] == 1          # read back from the communicator (ncclCommCount)


@pytest.mark.parametrize("schedule", ["fused", "carried"])
def test_overlapped_allreduce_is_bitwise_the_in_order_run(T, schedule):
    # thip_solver_set_overlap: the x / y updates run as an m-part (local rows, under the collective) and an n-part
    # (after it); the arithmetic per element is unchanged, so the iterates must be bitwise those of the in-order run
    n, cones = 36, [7, 20, 33, 4]
    f, Gs, hs, cs, d = random_socp(n, cones, seed=11)
    socp = T.ProbSOCP(_mb(T, T.MatType.General(n, 1)).set_array(f.reshape(-1, 1)),
                      [_mb(T, T.MatType.General(g.shape[0], n)).set_array(g) for g in Gs],
                      [_mb(T, T.MatType.General(len(v), 1)).set_array(v.reshape(-1, 1)) for v in hs],
                      [_mb(T, T.MatType.General(n, 1)).set_array(v.reshape(-1, 1)) for v in cs], d,
                      _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)))
    parts = _split_rows(socp.dense(), T, 2)
    p = T.SolverParam()
    p.eps_acc = 0.0
    a = _run_sharded(T, parts, p, schedule, max_steps=300, overlap=False)
    b = _run_sharded(T, parts, p, schedule, max_steps=300, overlap=True)
    for (ra, _, _, (xa, ya)), (rb, _, _, (xb, yb)) in zip(a, b):
        assert ra.iters == rb.iters == 300
        assert np.array_equal(xa, xb) and np.array_equal(ya, yb)
        assert ra.cri == rb.cri
    socp.drop()


# ---- column-split pipeline (thip_solver_set_overlap modes 2 / 3) ------------------------------------------------------

def _pipeline_socp(T, n, cones, seed):
    f, Gs, hs, cs, d = random_socp(n, cones, seed=seed)
    return T.ProbSOCP(_mb(T, T.MatType.General(n, 1)).set_array(f.reshape(-1, 1)),
                      [_mb(T, T.MatType.General(g.shape[0], n)).set_array(g) for g in Gs],
                      [_mb(T, T.MatType.General(len(v), 1)).set_array(v.reshape(-1, 1)) for v in hs],
                      [_mb(T, T.MatType.General(n, 1)).set_array(v.reshape(-1, 1)) for v in cs], d,
                      _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)))


@pytest.mark.parametrize("n,cones", [(36, [7, 20, 33, 4]), (700, [40, 3, 250, 99, 0, 130])])
def test_column_pipeline_is_bitwise_its_in_order_form(T, n, cones):
    # mode 2 (all-reduce of a column half under the next half-launch, termination test one half-launch late) against
    # mode 3 (the same kernels, collectives in order, no skew): same arithmetic, so the iterates after 300 iterations
    # must be equal bit for bit; and against mode 0 (one launch per stage) to f32 round-off
    socp = _pipeline_socp(T, n, cones, 11)
    parts = _split_rows(socp.dense(), T, 2)
    p = T.SolverParam()
    p.eps_acc = 0.0
    a = _run_sharded(T, parts, p, "carried", max_steps=300, overlap=3)
    b = _run_sharded(T, parts, p, "carried", max_steps=300, overlap=2)
    c = _run_sharded(T, parts, p, "carried", max_steps=300, overlap=0)
    for (ra, _, _, (xa, ya)), (rb, _, _, (xb, yb)), (rc, _, _, (xc, yc)) in zip(a, b, c):
        assert ra.iters == rb.iters == rc.iters == 300
        assert np.array_equal(xa, xb) and np.array_equal(ya, yb)
        assert ra.cri == rb.cri
        assert np.abs(xa - xc).max() <= 2e-4 * max(np.abs(xc).max(), 1e-6)
        assert np.abs(ya - yc).max() <= 2e-4 * max(np.abs(yc).max(), 1e-6)
    socp.drop()


def test_column_pipeline_stops_at_the_iteration_the_in_order_run_stops_at(T):
    # the termination test of iteration k is enqueued after the first half-launch of iteration k + 1: the answer must
    # still be the iterate of stopping at exactly iteration k -- same iteration count, bitwise the same x, y as mode 3,
    # whatever the polling period; and the converged answer agrees with the unsharded solve and the oracle
    socp = _pipeline_socp(T, 40, [9, 30, 0, 5, 64, 12], 5)
    dense = socp.dense()
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 200_000, 1e-4
    parts = _split_rows(dense, T, 3)
    a = _run_sharded(T, parts, p, "carried", overlap=3)
    b = _run_sharded(T, parts, p, "carried", overlap=2)
    for (ra, xa, ya, _), (rb, xb, yb, _) in zip(a, b):
        assert ra.state == rb.state == 0 and ra.iters == rb.iters
        assert np.array_equal(xa, xb) and np.array_equal(ya, yb)
    fs = T.FusedSolver.from_dense(dense, p, "carried")
    x1, y1 = fs.solve()
    r1 = fs.status()
    fs.destroy()
    (rb0, xb0, yb0, _), (rb1, _, yb1, _) = b
    assert abs(rb0.iters - r1.iters) <= max(2, 0.01 * r1.iters)
    assert np.allclose(xb0, x1, rtol=2e-3, atol=2e-4)
    assert np.allclose(np.concatenate([yb0, yb1]), y1, rtol=2e-3, atol=2e-4)
    ro = O.solve_matop_cones(O.param(max_iter=200000, eps_acc=1e-4), dense.vec_c, dense.mat_a, dense.vec_b,
                             dense.seg_type, dense.seg_len)
    pobj = float(dense.vec_c.astype(np.float64) @ ro.x)
    assert abs(float(dense.vec_c.astype(np.float64) @ xb0) - pobj) <= 1e-3 * (1 + abs(pobj))
    # max_iter inside the run: ExcessIter at the same iteration in both forms
    p2 = T.SolverParam()
    p2.max_iter, p2.eps_acc = 37, 0.0
    a = _run_sharded(T, parts, p2, "carried", overlap=3)
    b = _run_sharded(T, parts, p2, "carried", overlap=2)
    for (ra, xa, ya, _), (rb, xb, yb, _) in zip(a, b):
        assert ra.state == rb.state == 3 and ra.iters == rb.iters == 36
        assert np.array_equal(xa, xb) and np.array_equal(ya, yb)
    socp.drop()


def test_column_pipeline_falls_back_where_it_does_not_apply(T):
    # modes 2 / 3 need a collective, a dense A and the carried schedule; elsewhere the library says which mode runs
    socp = _pipeline_socp(T, 36, [7, 20, 33, 4], 3)
    p = T.SolverParam()
    p.eps_acc = 0.0
    d = socp.dense()
    fs = T.FusedSolver(d.n, d.m, d.mat_a, d.vec_b, d.vec_c, d.seg_type, d.seg_len, p, "carried", allreduce=("spin", 0),
                       overlap=2)
    info = fs.overlap_info()
    assert info["mode"] == 2 and info["launches_per_pass"] == 2 and 0 < info["split_col"] < d.n
    fs.run(20)
    fs.destroy()
    fs = T.FusedSolver(d.n, d.m, d.mat_a, d.vec_b, d.vec_c, d.seg_type, d.seg_len, p, "fused", allreduce=("spin", 0),
                       overlap=2)
    assert fs.overlap_info() == {"mode": 1, "launches_per_pass": 1, "split_col": 0}
    fs.run(20)
    fs.destroy()
    fs = T.FusedSolver(d.n, d.m, d.mat_a, d.vec_b, d.vec_c, d.seg_type, d.seg_len, p, "carried")
    fs.set_overlap(3)
    assert fs.overlap_info()["mode"] == 0        # no collective installed
    fs.destroy()
    socp.drop()


def test_column_pipeline_hides_an_injected_collective_latency(T):
    # The per-GPU work of BASELINE configs[2] at N = 8: a 12 500 x 50 000 row shard (2.5 GB).  The "collective" is a
    # device spin of L microseconds on the stream the hook is given (thip_test_spin_allreduce: the sum over one rank).
    # In order (mode 0) an iteration grows by 2 L (two collectives per iteration of the carried schedule); with the
    # column-split pipeline (mode 2) every collective has a half-pass over the shard (~0.19 ms) to hide in.
    import json
    import os
    import time
    from totsu_amd import synth
    from totsu_amd._lib import lib
    inst = synth.SocpInstance(50_000, 1000, 99, seed=0, first_cones=125)
    p = T.SolverParam()
    p.eps_acc = 0.0
    K = 300

    def rate(fs):
        fs.run(40, poll_every=40)
        best = 1e30
        for _ in range(3):
            lib.thip_sync()
            t0 = time.perf_counter()
            fs.run(K, poll_every=K)
            lib.thip_sync()
            best = min(best, (time.perf_counter() - t0) / K)
        return best * 1e6          # us per iteration

    fs = T.FusedSolver(inst.n, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, "carried",
                       allreduce=("spin", 0), overlap=0)
    t = {}
    plans = {}
    for mode in (0, 1, 3, 2):
        fs.set_overlap(mode)
        for L in (0, 30, 60):
            fs.set_spin_latency(L)
            t[mode, L] = rate(fs)
        plans[mode] = fs.gemv_plan()
    info = fs.overlap_info()
    assert info["mode"] == 2 and info["launches_per_pass"] == 2
    # bitwise: the pipeline against its in-order form, from the same fresh start
    its = {}
    for mode in (3, 2):
        fs.set_overlap(mode)
        fs.set_spin_latency(40)
        fs.reinit()
        fs.run(25, poll_every=7)
        its[mode] = fs.iterate()
    fs.destroy()
    inst.free()
    rec = {"what": "us per iteration of the carried schedule on a 12 500 x 50 000 row shard (BASELINE configs[2] at N = 8) with "
                   "a stand-in collective of L us (device spin on the stream the hook is given); overlap modes 0 in order, "
                   "1 local rows under the collective, 3 column-split in order, 2 column-split pipeline",
           "us_per_iteration": {"mode%d_L%d" % k: round(v, 1) for k, v in t.items()},
           "gemv_plan": {str(k): v for k, v in plans.items()}, "split_col": info["split_col"]}
    print(json.dumps(rec))
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                           "pipeline_latency_injection.json"), "w") as f:
        json.dump(rec, f)
    assert np.array_equal(its[2][0], its[3][0]) and np.array_equal(its[2][1], its[3][1])
    # in order: + 2 L (allow 25 % slack for launch jitter), split or not
    assert t[0, 60] - t[0, 0] >= 1.5 * 60, t
    assert t[0, 30] - t[0, 0] >= 1.5 * 30, t
    assert t[3, 60] - t[3, 0] >= 1.5 * 60, t
    # pipelined: the latency disappears under the half-launches
    assert t[2, 60] - t[2, 0] <= 10.0, t
    assert t[2, 30] - t[2, 0] <= 10.0, t
    # and it pays: at L = 60 the pipeline beats every other form
    assert t[2, 60] < min(t[0, 60], t[1, 60], t[3, 60]), t


# ---- one-shot all-reduce over peer-mapped buffers (thip_oneshot_*) --------------------------------------------------

@pytest.mark.parametrize("world", [2, 4, 8])
def test_oneshot_allreduce_between_processes_sharing_the_gpu(world):
    # hipIpc* works between processes on the SAME device, so a 1-GPU box can run the real protocol: N processes, the
    # handshake flags, the alternating slots, sums in rank order -- checked bit for bit by every rank (tests/oneshot_worker.py)
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29590 + world),
                        os.path.join(root, "tests", "oneshot_worker.py")], capture_output=True, text=True, timeout=600,
                       env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "ONESHOT_OK world=%d" % world in r.stdout


def test_bench_oneshot_transport_is_bitwise_the_gloo_staged_run():
    # two ranks sharing the GPU, the solver's own collectives through the one-shot transport: a + b == b + a in f32, so
    # the run must reproduce the gloo-staged one EXACTLY -- same iteration count, same objective digits (the per-rank
    # matrix is below the GEMV autotune threshold, so both runs use the same tiling)
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--gpus", "2", "--size", "2000", "--cones", "40", "--steps", "5", "--warmup", "1", "--no-cpu", "--no-gate",
              "--to-eps", "1e-3", "--overlap", "off"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = {}
    for coll in ("gloo", "oneshot"):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--collective", coll] + common,
                           capture_output=True, text=True, timeout=900, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, lines
        out[coll] = json.loads(lines[0])
    g, o = out["gloo"], out["oneshot"]
    assert o["n_gpus"] == 2 and "one-shot" in o["config"]["collective"]
    assert g["time_to_eps"]["state"] == o["time_to_eps"]["state"] == 0
    assert g["time_to_eps"]["iterations"] == o["time_to_eps"]["iterations"]
    assert g["time_to_eps"]["primal_obj"] == o["time_to_eps"]["primal_obj"]
    assert g["time_to_eps"]["dual_obj"] == o["time_to_eps"]["dual_obj"]
    assert g["time_to_eps"]["cri"] == o["time_to_eps"]["cri"]
    # and with the column-split pipeline on top (collectives on the side stream, four per iteration)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--collective", "oneshot"] + common[:-1] + ["pipeline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    p = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert p["config"]["overlap_mode_run"] == 2
    assert p["time_to_eps"]["state"] == 0
    assert abs(p["time_to_eps"]["iterations"] - o["time_to_eps"]["iterations"]) <= max(3, 0.01 * o["time_to_eps"]["iterations"])
    assert abs(p["time_to_eps"]["primal_obj"] - o["time_to_eps"]["primal_obj"]) <= 1e-4 * (1 + abs(o["time_to_eps"]["primal_obj"]))
