#!/usr/bin/env python
"""bench.py -- solver iterations/sec of the device-resident conic loop on the dense random SOCP of
BASELINE.json (configs[2]: n = 50 000, 1000 second-order cones of 1 + 99 rows => A is 100 000 x 50 000 f32,
20 GB), row-sharded over N GPUs (one process per GPU; the A^T y partial sums are all-reduced over RCCL).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one solver iteration (SolverCore::solve loop body, totsu_core/src/solver/solver.rs:364-457:
update_vecs + criteria).  Inputs are generated on the device before the timed region.  Prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="socp", choices=["socp", "lp", "sdp", "sparse-lp", "sparse-sdp"],
                    help="sparse-lp: the l1reg_lp construction (examples/l1reg_lp/src/main.rs:50-116) with --n samples (default "
                         "16384: 2.1 GB of non-zeros), A held sparse ONCE on the device (thip_sptile); sparse-sdp: the partitioning_sdp "
                         "construction (examples/partitioning_sdp/src/main.rs:45-78) with a PSD cone of order --k, n = k (k + 1) / 2")
    ap.add_argument("--no-two-copy", action="store_true",
                    help="sparse workloads: skip the A/B leg that iterates the same instance on round 5's two CSR copies")
    ap.add_argument("--k", type=int, default=500, help="PSD order of the sdp workload")
    ap.add_argument("--n", "--size", dest="n", type=int, default=None, help="number of variables (use --size under torchrun: --n is an ambiguous prefix there)")
    ap.add_argument("--cones", type=int, default=1000)
    ap.add_argument("--schedule", default="sweep", choices=["reference", "fused", "carried", "sweep"],
                    help="sweep = one pass over A per iteration (one GPU, dense f32 A); where it cannot run the library "
                         "executes the carried schedule (2 passes) and the line says so")
    ap.add_argument("--shard", default="auto", choices=["auto", "rows", "cols"],
                    help="N > 1: rows = cone-aligned row blocks, 2 passes per iteration (carried schedule) and an all-reduce "
                         "per A^T product; cols = column blocks, ONE pass per iteration (sweep schedule) and one all-reduce "
                         "of the two N products; auto = cols for --schedule sweep on the f32 fused path, else rows")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="one GPU: build RANK 0's shard of a run over this many GPUs and iterate it alone (the collective is a "
                         "stand-in that takes --emulate-latency microseconds): the per-GPU rate of that run, not a solve")
    ap.add_argument("--emulate-latency", type=int, default=0)
    ap.add_argument("--a-storage", default="f32", choices=["f32", "bf16", "f16", "mixed", "mixed-bf16"],
                    help="stored form of A streamed by the iteration (default f32 = the reference's data). bf16: a rounded copy, "
                         "half the bytes per pass, f32 accumulation -- solves the ROUNDED problem, not the headline metric. "
                         "mixed (with --to-eps): f16 passes to eps, then f32 passes to eps on the exact matrix (mixed-bf16: bf16 first)")
    ap.add_argument("--bf16-direct", action="store_true",
                    help="lp workload: build A as bf16 from f32 column blocks, never holding the f32 matrix (a 16-bit A is half "
                         "the HBM: configs[4], n = 200000, fits ONE GPU this way); implies --a-storage bf16")
    ap.add_argument("--f16-direct", action="store_true", help="like --bf16-direct with column-scaled f16 entries")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-gate", action="store_true",
                    help="skip the f64 re-evaluation of the time-to-eps answer (objective_gate.this_run; ~5 s of host f64 at "
                         "the full size, outside every timed region)")
    ap.add_argument("--force-collective", action="store_true", help="install the all-reduce hook even at N = 1")
    ap.add_argument("--collective", default="rccl", choices=["rccl", "torch", "gloo", "oneshot"],
                    help="rccl: native RCCL call on the library's stream; torch: torch.distributed.all_reduce hook; "
                         "oneshot: the library's one-launch all-reduce over peer-mapped buffers (hipIpc*; every rank reads its "
                         "N - 1 peers directly -- for the latency-bound <= 2 MB messages; also works with ranks sharing a GPU); "
                         "gloo: all-reduce staged through the host (lets several ranks share ONE GPU: plumbing tests)")
    ap.add_argument("--to-eps", type=float, default=None,
                    help="also solve to this eps_acc and report time-to-eps (default: 1e-3 for the socp workload at its "
                         "full size -- the eps_acc the reference runs its f32 backend at, benchmark_lp/src/main.rs:62-65)")
    ap.add_argument("--overlap", default="auto", choices=["auto", "on", "off", "pipeline", "pipeline-inorder"],
                    help="N > 1 (thip_solver_set_overlap): off = collectives in order on the launch stream; on = on the "
                         "solver's side stream under the local-row work; pipeline = column-split pipeline (the all-reduce of "
                         "a column half under the next half-launch); auto = time off / on / pipeline on the real communicator "
                         "during the warm-up and keep the fastest")
    ap.add_argument("--path", default="fused", choices=["fused", "trait"],
                    help="fused: the device-resident loop (the product's hot path).  trait: the compiled trait-level host "
                         "(examples/trait_host.cpp over include/totsu_f32hip.hpp): Solver::solve call by call through the "
                         "reference's composite operators, one L:: call per reference call -- what an UNCHANGED caller "
                         "gets without the Hip* alias types; N = 1, lp / socp workloads")
    ap.add_argument("--trait-cones", default="reference", choices=["reference", "device"],
                    help="--path trait: the reference's literal cone code (host loop over get_mut / get + norm + scale per "
                         "cone) or the device-side projections (thip_proj_*)")
    ap.add_argument("--no-to-eps", action="store_true", help="skip the time-to-eps leg (iterations/sec only)")
    ap.add_argument("--no-mixed-leg", action="store_true",
                    help="the default socp line also solves to eps with the first phase streamed from an f16-stored copy of A "
                         "(time_to_eps_mixed, ~ +3 minutes); this skips it")
    ap.add_argument("--mixed-leg", action="store_true", help="add the time_to_eps_mixed leg to any single-GPU --to-eps run")
    ap.add_argument("--no-row-leg", action="store_true",
                    help="column-sharded runs also time a short row-sharded (carried schedule) leg on the same ranks and report it as "
                         "`row_sharded`; this skips it")
    ap.add_argument("--to-eps-budget", type=float, default=1200.0,
                    help="stop the time-to-eps leg after this many seconds and report the criteria reached (state -1)")
    ap.add_argument("--state", default="compensated", choices=["compensated", "plain"],
                    help="thip_param.state_arith: compensated (Kahan) or plain f32 iterate updates")
    ap.add_argument("--watchdog", type=float, default=1500.0,
                    help="N > 1: seconds after which rank 0 prints the line as far as it has got and every rank leaves (a leg behind "
                         "the timed region that hangs in a collective must not cost the line); 0 = off")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU, no HIP library: run the control path only -- rank spawn, process group (gloo), the column-shard "
                         "agreement, the shard plan with its HBM budget, the barrier / max-over-ranks timing bracket, the ONE JSON line")
    ap.add_argument("--cpu-cones", type=int, default=328,
                    help="cones in the CPU sample: the first 328 of the instance by default (A_sub 13 GB of f64), a FIXED "
                         "count so that the baseline reproduces from host to host")
    return ap.parse_args()


class _CAI:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class GlooAllreduce:
    """all-reduce through host memory over the gloo backend: slow, but lets N processes share one GPU, which RCCL
    refuses -- used to test the multi-process plumbing (sharding, hook placement, rank-0 output) on a 1-GPU box"""

    def __init__(self, torch, dist, lib):
        self.torch, self.dist, self.lib, self.calls = torch, dist, lib, 0

    def __call__(self, ctx, ptr, n, stream):
        try:
            h = np.empty(n, dtype=np.float32)
            self.lib.thip_d2h(h.ctypes.data, ptr, n)          # synchronises the library's stream
            t = self.torch.from_numpy(h)
            self.dist.all_reduce(t)
            self.lib.thip_h2d(ptr, h.ctypes.data, n)
            self.calls += 1
            return 0
        except Exception as e:
            sys.stderr.write("gloo all-reduce hook failed: %r\n" % (e,))
            return 1


class TorchAllreduce:
    """thip_allreduce_fn over torch.distributed (backend nccl == RCCL on ROCm): sums `n` floats in place.
    The collective is issued with the LIBRARY's launch stream as torch's current stream (ExternalStream over the
    handle the hook receives), so RCCL is ordered after the kernels that produced the buffer and before the
    kernels that consume it, whatever stream the library runs on."""

    def __init__(self, torch, dist):
        self.torch, self.dist, self.cache, self.calls = torch, dist, {}, 0
        self.ext = {}

    def __call__(self, ctx, ptr, n, stream):
        try:
            key = (ptr, n)
            t = self.cache.get(key)
            if t is None:
                t = self.torch.as_tensor(_CAI(ptr, n), device="cuda")
                self.cache[key] = t
            ext = self.ext.get(stream)
            if ext is None:
                ext = self.torch.cuda.ExternalStream(stream) if stream else self.torch.cuda.default_stream()
                self.ext[stream] = ext
            with self.torch.cuda.stream(ext):
                self.dist.all_reduce(t)
            self.calls += 1
            return 0
        except Exception as e:      # an exception must not unwind through the C frame
            sys.stderr.write("all-reduce hook failed: %r\n" % (e,))
            return 1


def oracle_sub_instance(n, n_cones_full, ni, seed, cones_sub):
    """The first `cones_sub` cones of the synthetic SOCP as a standalone problem, built on the HOST in f64 through the
    oracle's generator (identical matrix entries to the device's; synth.SocpInstance(first_cones=cones_sub) is the
    device-side twin).  Returns (f, A column-major flat, b, seg_types, seg_lens) for oracle.solve_matop_cones."""
    import oracle as O
    from totsu_amd import synth as S
    rows = 1 + ni
    m = cones_sub * rows
    m_total = n_cones_full * rows
    A = O.gen_matrix(m, n, seed, S.STREAM_A, 0, 0, m_total, 1, -1.0 / math.sqrt(n))
    x0 = O.gen_vector(n, seed, S.STREAM_X0, 0, 1)
    w = O.transform_ge(False, m, n, 1.0, A, x0, 0.0, np.zeros(m)).reshape(cones_sub, rows)
    h = O.gen_vector(m, seed, S.STREAM_H, 0, 1).reshape(cones_sub, rows)
    margin = O.gen_vector(cones_sub, seed, S.STREAM_D, 0, 0, 1.0, 0.1)
    b = h.copy()
    b[:, 0] = np.linalg.norm(h[:, 1:] - w[:, 1:], axis=1) + w[:, 0] + margin
    t = O.gen_vector(cones_sub, seed, S.STREAM_T, 0, 0, 1.0, 0.5)
    wd = O.gen_vector(m, seed, S.STREAM_W, 0, 1).reshape(cones_sub, rows)[:, 1:]
    ws = O.gen_vector(cones_sub, seed, S.STREAM_WS, 0, 0)
    wd = wd * (0.9 * t * ws / np.maximum(np.linalg.norm(wd, axis=1), 1e-9))[:, None]
    z = np.concatenate([t[:, None], wd], axis=1).reshape(-1)
    f = O.transform_ge(True, m, n, -1.0, A, z, 0.0, np.zeros(n))
    return f, A, b.reshape(-1), [O.CONE_SOC] * cones_sub, [rows] * cones_sub


def cpu_baseline(n, n_cones_full, ni, seed, cones_sub, budget_s=25.0):
    """Times the CPU oracle (f64, OpenMP) on a BOUNDED sample: the first `cones_sub` cones of the same
    instance (identical matrix entries), a few iterations; iteration cost is linear in the number of rows, so
    the full-size rate is the measured rate * cones_sub / n_cones_full."""
    import oracle as O
    rows = 1 + ni
    m = cones_sub * rows
    f, A, b, seg_t, seg_l = oracle_sub_instance(n, n_cones_full, ni, seed, cones_sub)

    def run(k):
        par = O.param(max_iter=k, eps_acc=1e-300)
        t0 = time.perf_counter()
        r = O.solve_matop_cones(par, f, A, b, seg_t, seg_l)
        return time.perf_counter() - t0, r

    t1, _ = run(2)                      # init (norms, preconditioner) + 2 iterations
    per_iter_guess = max(t1 / 4.0, 1e-3)
    # three independent timings of k2 iterations each (every one pays the init again, subtracted as t1): the median is
    # the value, min / max the spread
    k2 = int(max(3, min(70, budget_s / 3.0 / per_iter_guess)))
    rates = []
    for _ in range(3):
        t2, r2 = run(2 + k2)
        rates.append(k2 / (t2 - t1) if t2 > 1.05 * t1 else (2 + k2) / max(t2, 1e-9))   # tiny samples: init time is noise
    rates.sort()
    rate_sub = rates[1]
    # The reference's own f64 backend spends its iteration in 3 + 3 dgemv calls (f64lapack.rs:123-146, MKL there):
    # the same six products through numpy's BLAS on the same sub-matrix bound its iteration rate from above.
    blas = None
    At = None
    try:
        At = np.asarray(A).reshape(n, m)            # column-major (m x n) seen row-major is A^T
        xv, yv = np.ones(n), np.ones(m)
        reps, tb = 0, 0.0
        (At.T @ xv, At @ yv)                        # warm-up
        while reps < 3 or (tb < 2.0 and reps < 50):
            t0 = time.perf_counter()
            for _ in range(3):
                yv2 = At.T @ xv
                xv2 = At @ yv
            tb += time.perf_counter() - t0
            reps += 1
        nthr = None
        try:
            from threadpoolctl import threadpool_info
            nthr = max([i.get("num_threads", 0) for i in threadpool_info() if i.get("user_api") == "blas"] or [0]) or None
        except Exception:
            pass
        rate_blas = reps / tb
        blas = {"value": rate_blas * cones_sub / n_cones_full, "unit": "iter/s (GEMV share only: an upper bound)",
                "threads": nthr, "measured_sub_instance_iter_per_s": rate_blas,
                "GBps": 6.0 * m * n * 8.0 * rate_blas / 1e9,
                "what": "3 x (A x) + 3 x (A^T y) in f64 through numpy's BLAS on the same sub-matrix, scaled by rows"}
    except Exception as e:                          # never let the optional leg break the bench line
        blas = {"error": repr(e)}
    sub = {
        "value": rate_sub * cones_sub / n_cones_full,
        "unit": "iter/s",
        "cores": O.num_threads(),
        "kind": "port",
        "sample": ("oracle (C, f64, OpenMP %d threads) on the first %d of %d cones of the same instance "
                   "(A_sub %d x %d f64), median of 3 timings of %d iterations: %.3f iter/s (min %.3f, max %.3f), "
                   "scaled by rows %d/%d"
                   % (O.num_threads(), cones_sub, n_cones_full, m, n, k2, rate_sub, rates[0], rates[2], cones_sub,
                      n_cones_full)),
        "measured_sub_instance_iter_per_s": rate_sub,
        "spread_iter_per_s": [rates[0] * cones_sub / n_cones_full, rates[2] * cones_sub / n_cones_full],
        "host_cpu_count": os.cpu_count(),
        "blas_gemv_bound": blas,
    }
    A = At = None                        # (At: the BLAS leg's view of the same 13 GB)
    # The CONFIG itself, not a sample: all n_cones_full cones (A 40 GB of f64 at the headline size), 3 + 3 iterations -- when the
    # host can hold it; the sub-instance figure above stays beside it as the cross-check
    full = None
    try:
        import psutil
        need = 8.0 * n * n_cones_full * rows
        if cones_sub < n_cones_full and psutil.virtual_memory().available > 1.5 * need:
            t0 = time.perf_counter()
            f2, A2, b2, st2, sl2 = oracle_sub_instance(n, n_cones_full, ni, seed, n_cones_full)
            t_gen = time.perf_counter() - t0

            def run_full(k):
                t0 = time.perf_counter()
                O.solve_matop_cones(O.param(max_iter=k, eps_acc=1e-300), f2, A2, b2, st2, sl2)
                return time.perf_counter() - t0
            ta, tb = run_full(3), run_full(6)
            rate_full = 3.0 / (tb - ta)
            full = {"value": rate_full, "unit": "iter/s", "iterations": "3 + 3 (two solves from x = 0, the second three iterations longer)",
                    "seconds": [ta, tb], "gen_seconds": t_gen,
                    "what": "oracle (C, f64, OpenMP %d threads) on the WHOLE instance: %d cones, A %d x %d f64 = %.0f GB"
                            % (O.num_threads(), n_cones_full, n_cones_full * rows, n, need / 1e9)}
            del A2
    except Exception as e:              # the optional leg never costs the line
        full = {"error": repr(e)}
    if full and "value" in full:
        out = dict(sub)
        out.update({"value": full["value"], "sample": full["what"] + ", 3 + 3 iterations: %.3f iter/s (sub-instance of %d cones scaled by rows: %.3f)"
                    % (full["value"], cones_sub, sub["value"]), "full_instance": full,
                    "sub_instance": {k: sub[k] for k in ("value", "sample", "measured_sub_instance_iter_per_s", "spread_iter_per_s")}})
        return out
    sub["full_instance"] = full
    return sub


def kkt_f64(inst, x, y_local, allreduce_host, block_cones=50):
    """The answer of THIS run evaluated in f64 on the host (post-solve leg, outside every timed region): A is regenerated
    block by block, widened, from the counter-based generator (the oracle's oc_gen_matrix: the checker, bit-identical to
    the entries the device holds), and the reference's stopping quantities (solver.rs:573-612) are recomputed from x, y:
        s = b - A x   (distance of s from K: max(0, ||s_1|| - s_0) per cone),   r = c + A^T y,   gap = c.x + b.y.
    An approximate-KKT evaluation, not a bracket: with a dual residual of eps_acc the primal objective may sit below the
    dual one.  Row-sharded runs evaluate their own cones and sum / max over the ranks."""
    import oracle as O
    from totsu_amd import synth as S
    n, rows = inst.n, 1 + inst.ni
    nloc = inst.c1 - inst.c0
    x64, y64 = x.astype(np.float64), y_local.astype(np.float64)
    b64, c64 = inst.vec_b_host.astype(np.float64), inst.vec_c_host.astype(np.float64)
    r = np.zeros(n)
    viol_p = viol_d = 0.0
    t0 = time.perf_counter()
    for k0 in range(0, nloc, block_cones):
        nc = min(block_cones, nloc - k0)
        mb = nc * rows
        A = np.asarray(O.gen_matrix(mb, n, inst.seed, S.STREAM_A, (inst.c0 + k0) * rows, 0, inst.m_total, 1,
                                    -1.0 / math.sqrt(n)))
        At = A.reshape(n, mb)                     # column-major (mb x n) seen row-major is A^T
        yb = y64[k0 * rows:(k0 + nc) * rows]
        sl = (b64[k0 * rows:(k0 + nc) * rows] - At.T @ x64).reshape(nc, rows)
        r += At @ yb
        viol_p = max(viol_p, float(np.max(np.maximum(0.0, np.linalg.norm(sl[:, 1:], axis=1) - sl[:, 0]))))
        yy = yb.reshape(nc, rows)
        viol_d = max(viol_d, float(np.max(np.maximum(0.0, np.linalg.norm(yy[:, 1:], axis=1) - yy[:, 0]))))
    sums = allreduce_host(np.concatenate([r, [float(b64 @ y64), float(b64 @ b64)]]))
    mx = allreduce_host(np.array([viol_p, viol_d]), op="max")
    r = sums[:n] + c64
    by, bb = float(sums[n]), float(sums[n + 1])
    pobj, dobj = float(c64 @ x64), -by
    return {
        "primal_obj_f64": pobj, "dual_obj_f64": dobj,
        "gap_rel": abs(pobj - dobj) / (1.0 + abs(pobj) + abs(dobj)),
        "dual_residual_rel_f64": float(np.linalg.norm(r)) / (1.0 + float(np.linalg.norm(c64))),
        "primal_cone_violation": float(mx[0]),
        "primal_cone_violation_rel_to_norm_b": float(mx[0]) / (1.0 + math.sqrt(bb)),
        "dual_cone_violation": float(mx[1]),
        "f64_evaluation_seconds": time.perf_counter() - t0,
        "what": "x, y of THIS run's time_to_eps solve re-evaluated in f64 against the regenerated A (solver.rs:573-612's "
                "quantities): approximate KKT, not a bracket",
    }


def kkt_f64_lp(inst, x, y_local, allreduce_host, block_rows=4000):
    """kkt_f64 for the benchmark_lp construction (rows [r0, r1) of [-I ; U(0, 1)] on this rank, nonneg cone): the same
    quantities, the cone tests being min(s) >= 0 and min(y) >= 0."""
    import oracle as O
    from totsu_amd import synth as S
    n = inst.n
    x64, y64 = x.astype(np.float64), y_local.astype(np.float64)
    b64, c64 = inst.vec_b_host.astype(np.float64), inst.vec_c_host.astype(np.float64)
    r = np.zeros(n)
    viol_p = viol_d = 0.0
    t0 = time.perf_counter()
    for k0 in range(0, inst.m, block_rows):
        mb = min(block_rows, inst.m - k0)
        g0 = inst.r0 + k0                          # global index of the block's first row
        At = np.asarray(O.gen_matrix(mb, n, inst.seed, S.STREAM_A, g0, 0, inst.m_total, 0, 1.0)).reshape(n, mb)
        for i in range(max(0, min(mb, n - g0))):   # rows below n of the full matrix are -I
            At[:, i] = 0.0
            At[g0 + i, i] = -1.0
        yb = y64[k0:k0 + mb]
        sl = b64[k0:k0 + mb] - At.T @ x64
        r += At @ yb
        viol_p = max(viol_p, float(np.max(np.maximum(0.0, -sl))))
        viol_d = max(viol_d, float(np.max(np.maximum(0.0, -yb))))
    sums = allreduce_host(np.concatenate([r, [float(b64 @ y64), float(b64 @ b64)]]))
    mx = allreduce_host(np.array([viol_p, viol_d]), op="max")
    r = sums[:n] + c64
    pobj, dobj = float(c64 @ x64), -float(sums[n])
    return {
        "primal_obj_f64": pobj, "dual_obj_f64": dobj,
        "gap_rel": abs(pobj - dobj) / (1.0 + abs(pobj) + abs(dobj)),
        "dual_residual_rel_f64": float(np.linalg.norm(r)) / (1.0 + float(np.linalg.norm(c64))),
        "primal_cone_violation": float(mx[0]),
        "primal_cone_violation_rel_to_norm_b": float(mx[0]) / (1.0 + math.sqrt(float(sums[n + 1]))),
        "dual_cone_violation": float(mx[1]),
        "f64_evaluation_seconds": time.perf_counter() - t0,
        "what": "x, y of THIS run's time_to_eps solve re-evaluated in f64 against the regenerated A = [-I ; U] (solver.rs:573-612's "
                "quantities; cone tests: s >= 0, y >= 0): approximate KKT, not a bracket",
    }


def kkt_f64_sdp(inst, x, y):
    """kkt_f64 for the synthetic SDP (one PSD cone of order k, A = [svec(F_i)]): the cone tests are the most negative
    eigenvalue of mat(s) and of mat(y) (numpy eigh in f64; svec scales the off-diagonal entries by sqrt 2)."""
    import oracle as O
    from totsu_amd import synth as S
    n, k, sk = inst.n, inst.k, inst.m
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    b64, c64 = inst.vec_b_host.astype(np.float64), inst.vec_c_host.astype(np.float64)
    t0 = time.perf_counter()
    ax = np.zeros(sk)
    r = np.zeros(n)
    for j0 in range(0, n, 250):
        nc = min(250, n - j0)
        A = np.asarray(O.gen_matrix(sk, nc, inst.seed, S.STREAM_A, 0, j0, sk, 1, 1.0 / math.sqrt(k))).reshape(nc, sk)
        ax += x64[j0:j0 + nc] @ A                 # row j of the reshaped block is column j of A
        r[j0:j0 + nc] = A @ y64
    r += c64

    def lam_min(v):
        mat = np.zeros((k, k))
        iu = np.triu_indices(k)                   # packed upper by columns: entry c (c + 1) / 2 + r is (r, c), r <= c
        order = np.lexsort((iu[0], iu[1]))
        rr, cc = iu[0][order], iu[1][order]
        mat[rr, cc] = v / np.where(rr == cc, 1.0, math.sqrt(2.0))
        mat = mat + np.triu(mat, 1).T
        return float(np.linalg.eigvalsh(mat)[0])
    sl = b64 - ax
    viol_p, viol_d = max(0.0, -lam_min(sl)), max(0.0, -lam_min(y64))
    pobj, dobj = float(c64 @ x64), -float(b64 @ y64)
    return {
        "primal_obj_f64": pobj, "dual_obj_f64": dobj,
        "gap_rel": abs(pobj - dobj) / (1.0 + abs(pobj) + abs(dobj)),
        "dual_residual_rel_f64": float(np.linalg.norm(r)) / (1.0 + float(np.linalg.norm(c64))),
        "primal_cone_violation": viol_p,
        "primal_cone_violation_rel_to_norm_b": viol_p / (1.0 + float(np.linalg.norm(b64))),
        "dual_cone_violation": viol_d,
        "f64_evaluation_seconds": time.perf_counter() - t0,
        "what": "x, y of THIS run's time_to_eps solve re-evaluated in f64 against the regenerated A (solver.rs:573-612's "
                "quantities; cone tests: the most negative eigenvalue of mat(s) and of mat(y)): approximate KKT, not a bracket",
    }


def kkt_f64_cols(inst, x_local, y, allreduce_host, block_cols=500):
    """kkt_f64 for a COLUMN-sharded answer: this rank holds x over its columns [col0, col1) and the whole of y (replicated).
    A x adds up over the ranks (one all-reduce of an m-vector), the dual residual r = c + A^T y is this rank's block (its
    squared norm adds up), the cone tests are the whole problem's on every rank."""
    import oracle as O
    from totsu_amd import synth as S
    n, rows, m = inst.n, 1 + inst.ni, inst.m
    nl = inst.col1 - inst.col0
    x64, y64 = x_local.astype(np.float64), y.astype(np.float64)
    b64, c64 = inst.vec_b_host.astype(np.float64), inst.vec_c_host.astype(np.float64)
    ax = np.zeros(m)
    r = np.zeros(nl)
    t0 = time.perf_counter()
    for j0 in range(0, nl, block_cols):
        nc = min(block_cols, nl - j0)
        A = np.asarray(O.gen_matrix(m, nc, inst.seed, S.STREAM_A, 0, inst.col0 + j0, m, 1, -1.0 / math.sqrt(n))).reshape(nc, m)
        ax += x64[j0:j0 + nc] @ A                 # row j of the reshaped block is column j of A
        r[j0:j0 + nc] = A @ y64
    ax = allreduce_host(ax)
    r += c64
    sums = allreduce_host(np.array([float(r @ r), float(c64 @ c64), float(c64 @ x64)]))
    sl = (b64 - ax).reshape(-1, rows)
    yy = y64.reshape(-1, rows)
    viol_p = float(np.max(np.maximum(0.0, np.linalg.norm(sl[:, 1:], axis=1) - sl[:, 0])))
    viol_d = float(np.max(np.maximum(0.0, np.linalg.norm(yy[:, 1:], axis=1) - yy[:, 0])))
    pobj, dobj = float(sums[2]), -float(b64 @ y64)
    return {
        "primal_obj_f64": pobj, "dual_obj_f64": dobj,
        "gap_rel": abs(pobj - dobj) / (1.0 + abs(pobj) + abs(dobj)),
        "dual_residual_rel_f64": math.sqrt(float(sums[0])) / (1.0 + math.sqrt(float(sums[1]))),
        "primal_cone_violation": viol_p,
        "primal_cone_violation_rel_to_norm_b": viol_p / (1.0 + float(np.linalg.norm(b64))),
        "dual_cone_violation": viol_d,
        "f64_evaluation_seconds": time.perf_counter() - t0,
        "what": "x (column blocks over the ranks), y of THIS run's time_to_eps solve re-evaluated in f64 against the regenerated "
                "A (solver.rs:573-612's quantities): approximate KKT, not a bracket",
    }


def stored_objective_evidence(schedule="sweep"):
    """what earlier runs measured about the 1e-4 objective gate, loaded from the committed files WITH their provenance
    (never re-typed into this file); None for a file that is absent.  The converged-objective-vs-f64-oracle files are the
    ones whose run used THIS line's schedule (tests/measure_objective_gap.py --schedule)."""
    ev = {}
    files = {"sweep": [("vs_f64_oracle_at_n2000", "r05_objective_gap_sweep_n2000.json"),
                       ("vs_f64_oracle_at_n8000", "r05_objective_gap_sweep_n8000.json")]}.get(
        schedule, [("vs_f64_oracle_at_n2000", "r01_objective_gap_socp_n2000.json")])
    for key, fn in files:
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", fn)))
            ev[key] = {"source": "profiles/%s (stored run of schedule %r, not this one)" % (fn, d.get("schedule", "carried")),
                       "values": d}
        except Exception:
            pass
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r02_c3_f64_certificate.json")))
        pts = {pt["eps_acc"]: pt for pt in d["points"]}
        a_, b_ = pts[1e-3]["primal_obj_f64"], pts[1e-4]["primal_obj_f64"]
        ev["objective_drift_eps1e-3_to_1e-4_full_size"] = {
            "source": "profiles/r02_c3_f64_certificate.json (stored run, not this one)",
            "primal_obj_eps1e-3": a_, "primal_obj_eps1e-4": b_, "relative_drift": abs(a_ - b_) / abs(b_)}
    except Exception:
        pass
    return ev or None


MFMA_F32_PEAK_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md: dense f32 matrix peak (256 CUs x 4 SIMDs x 64 flop/clk x 2.4 GHz)


def eig_record(k, ms_per_pair, spans, ms_per_iter):
    """north_star: 'MFMA utilisation for the eig step'.  The PSD cone's projection (ConePSD::proj of x_y and of x_s,
    cone_psd.rs:56-79; replaces the syevdx + <= k syr of f32cuda.rs:196-303) timed by HIP events around the chain of BOTH
    blocks, with the flops the chain executes (thip_eig.hip polar_project7: 32 x 32 x ld tiles of the lower triangle)."""
    ld = (k + 63) // 64 * 64
    nt = ld // 32
    tri = nt * (nt + 1) // 2
    while ld > 512 and (ld // 64) % ((ld + 511) // 512) != 0:    # thip_eig.hip np_of: whole K chunks above order 512
        ld += 64
    nt = ld // 32
    tri = nt * (nt + 1) // 2
    tiles = 11 * (tri + 2 * tri + tri) + 2 * tri + tri      # 11 degree-7 steps (S S; {Y Y, Y S}; U V), {NS, M S}, T R
    launches, products, chain = 11 * 3 + 2 + 2, 11 * 4 + 3, "all-symmetric degree-7 polar steps (round 5; orders above 512: round 6)"
    flops_pair = 2.0 * tiles * (2.0 * 32 * 32 * ld)             # both blocks
    tf = flops_pair / (ms_per_pair * 1e-3) / 1e12
    rec = {"bound": "mfma", "kernel": "gemm_pre2_k / polar_dual_k (v_mfma_f32_32x32x2_f32): the PSD projection chain of x_y and x_s",
           "chain": chain, "order": k, "ld": ld, "ms_per_projection_pair": ms_per_pair, "launches_per_projection_pair": launches,
           "products_per_projection": products, "flops_per_projection_pair": flops_pair,
           "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
           "share_of_iteration": ms_per_pair / ms_per_iter, "spans_timed": spans,
           "timer": "hip_events on the launch stream around the chain of every iteration of the timed region (thip_prof_read_psd)",
           "MfmaUtil_stored": None}
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r05_sdp_k500_mfma_util.json")))
        if d.get("order") == k:
            rec["MfmaUtil_stored"] = d
    except Exception:
        pass
    return rec


# ---------------------------------------------------------------------------------------------------
# sparse workloads: the reference's own example constructions, A held sparse once on the device
# ---------------------------------------------------------------------------------------------------
def sparse_lp_instance(l, seed=0, lam=0.2):
    """examples/l1reg_lp/src/main.rs:50-116 with l samples (the example: 20): n = 3 l + 1, m = 4 l, G = the dense 2 l x l Gaussian-kernel
    block [K ; -K] (sigma^2 = 1 / 8) between +-1 diagonals and one dense column -- nnz = 2 l^2 + 8 l of 12 l^2 + 4 l entries (1 / 6).
    Sample points from numpy's generator (the reference's Xoshiro stream is not reproducible here).  CSC arrays on the host."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 1, (2, l))
    y = np.cos(5.0 * x[0]) * np.cos(7.0 * x[1])
    n, m = 3 * l + 1, 4 * l
    nnz = 2 * l * l + 8 * l
    vals = np.empty(nnz, np.float32)
    rows = np.empty(nnz, np.int32)
    colptr = np.empty(n + 1, np.int64)
    ar = np.arange(l, dtype=np.int32)
    # columns 0 .. l - 1 (z): -1 at rows i and l + i
    vals[:2 * l] = -1.0
    rows[0:2 * l:2] = ar
    rows[1:2 * l:2] = l + ar
    colptr[:l] = 2 * np.arange(l, dtype=np.int64)
    pos = 2 * l
    # columns l .. 2 l - 1 (alpha): K[:, c], -K[:, c], +1 at 2 l + c, -1 at 3 l + c
    w = 2 * l + 2
    bv = vals[pos:pos + l * w].reshape(l, w)
    br = rows[pos:pos + l * w].reshape(l, w)
    x0, x1 = x[0].astype(np.float32), x[1].astype(np.float32)
    for c0 in range(0, l, 2048):
        c1 = min(l, c0 + 2048)
        d2 = (x0[c0:c1, None] - x0[None, :]) ** 2 + (x1[c0:c1, None] - x1[None, :]) ** 2
        k = np.exp(-8.0 * d2, dtype=np.float32)
        bv[c0:c1, :l] = k
        bv[c0:c1, l:2 * l] = -k
    bv[:, 2 * l] = 1.0
    bv[:, 2 * l + 1] = -1.0
    br[:, :2 * l] = np.arange(2 * l, dtype=np.int32)[None, :]
    br[:, 2 * l] = 2 * l + ar
    br[:, 2 * l + 1] = 3 * l + ar
    colptr[l:2 * l] = pos + w * np.arange(l, dtype=np.int64)
    pos += l * w
    # columns 2 l .. 3 l - 1 (beta): -1 at rows 2 l + c and 3 l + c
    vals[pos:pos + 2 * l] = -1.0
    rows[pos:pos + 2 * l:2] = 2 * l + ar
    rows[pos + 1:pos + 2 * l:2] = 3 * l + ar
    colptr[2 * l:3 * l] = pos + 2 * np.arange(l, dtype=np.int64)
    pos += 2 * l
    # column 3 l (bias): +1 at rows 0 .. l - 1, -1 at rows l .. 2 l - 1
    vals[pos:pos + l] = 1.0
    vals[pos + l:pos + 2 * l] = -1.0
    rows[pos:pos + 2 * l] = np.arange(2 * l, dtype=np.int32)
    colptr[3 * l] = pos
    pos += 2 * l
    colptr[n] = pos
    assert pos == nnz
    c = np.zeros(n, np.float32)
    c[:l] = 1.0
    c[2 * l:3 * l] = lam
    h = np.zeros(m, np.float32)
    h[:l] = y
    h[l:2 * l] = -y
    return {"n": n, "m": m, "colptr": colptr, "rowidx": rows, "vals": vals, "b": h, "c": c, "seg_type": [1], "seg_len": [m],
            "what": "l1reg_lp construction (examples/l1reg_lp/src/main.rs:50-116) with l = %d samples: sparse LP n=%d m=%d, nnz=%d "
                    "(%.1f %% of m n; the 2l x l kernel block dense), f32" % (l, n, m, nnz, 100.0 * nnz / (float(m) * n))}


def sparse_sdp_instance(k, seed=0):
    """examples/partitioning_sdp/src/main.rs:21-79 on a grid graph of k nodes: n = sk = k (k + 1) / 2 packed entries of X, minimise
    sum W_ij X_ij, X >= 0 (F_kk = -E_ij: one entry per column of the PSD rows, -1 on the diagonal, -sqrt 2 off it after ProbSDP's
    scale_nondiag, sdp.rs:271-274), diag X = 1 (one 1 per equality row).  The example's dense symmat_f is sk x sk (62.7 GB at
    k = 500); held sparse it is sk + k entries."""
    x_num = max(d for d in range(1, int(math.isqrt(k)) + 1) if k % d == 0)
    y_num = k // x_num
    rng = np.random.default_rng(seed)
    sk = k * (k + 1) // 2
    n, m = sk, sk + k
    jj = np.repeat(np.arange(k), np.arange(1, k + 1))                  # column j of the packed entry
    ii = np.arange(sk) - jj * (jj + 1) // 2                            # its row i <= j
    diag = ii == jj
    cnt = 1 + diag.astype(np.int64)
    colptr = np.zeros(n + 1, np.int64)
    np.cumsum(cnt, out=colptr[1:])
    nnz = int(colptr[n])
    rows = np.empty(nnz, np.int32)
    vals = np.empty(nnz, np.float32)
    first = colptr[:-1]
    rows[first] = np.arange(sk, dtype=np.int32)
    vals[first] = np.where(diag, -1.0, -math.sqrt(2.0)).astype(np.float32)
    dcols = np.nonzero(diag)[0]
    rows[first[dcols] + 1] = sk + jj[dcols]
    vals[first[dcols] + 1] = 1.0
    w = np.zeros(n, np.float32)
    for i in range(k):
        gx, gy = divmod(i, y_num)
        if gx < x_num - 1:
            j = i + y_num
            w[j * (j + 1) // 2 + i] = rng.standard_normal()
        if gy < y_num - 1:
            j = i + 1
            w[j * (j + 1) // 2 + i] = rng.standard_normal()
    b = np.concatenate([np.zeros(sk, np.float32), np.ones(k, np.float32)])
    return {"n": n, "m": m, "colptr": colptr, "rowidx": rows, "vals": vals, "b": b, "c": w, "seg_type": [4, 0], "seg_len": [sk, k],
            "what": "partitioning_sdp construction (examples/partitioning_sdp/src/main.rs:45-78) on a %d x %d grid: sparse SDP, one PSD "
                    "cone of order %d (sk=%d), n=%d m=%d, nnz=%d, f32" % (x_num, y_num, k, sk, n, m, nnz)}


def cpu_baseline_sparse(inst, budget_s=25.0):
    """the f64 oracle (C, OpenMP) through its sparse user-operator (oracle/totsu_oracle.c oc_solve_csc_cones: the pattern of
    examples/imgnr_udef/src/prob_op_a.rs) on the SAME instance, a few iterations"""
    import oracle as O
    v64 = inst["vals"].astype(np.float64)

    def run(k):
        par = O.param(max_iter=k, eps_acc=1e-300)
        t0 = time.perf_counter()
        O.solve_csc_cones(par, inst["c"], inst["colptr"], inst["rowidx"], v64, inst["b"], inst["seg_type"], inst["seg_len"], use_ql=True)
        return time.perf_counter() - t0

    t1 = run(2)
    k2 = int(max(3, min(200, budget_s / 3.0 / max(t1 / 4.0, 1e-3))))
    rates = []
    for _ in range(3):
        t2 = run(2 + k2)
        rates.append(k2 / (t2 - t1) if t2 > 1.05 * t1 else (2 + k2) / max(t2, 1e-9))
    rates.sort()
    return {"value": rates[1], "unit": "iter/s", "cores": O.num_threads(), "kind": "port",
            "sample": "oracle (C, f64, OpenMP %d threads) with A as a sparse user-defined Operator (CSC + its row-major mirror) on the "
                      "WHOLE instance, median of 3 timings of %d iterations (min %.3f, max %.3f iter/s); init %.2f s subtracted"
                      % (O.num_threads(), k2, rates[0], rates[2], t1),
            "spread_iter_per_s": [rates[0], rates[2]], "host_cpu_count": os.cpu_count()}


def run_sparse(a, rank, T, lib, _lib):
    import ctypes as C
    from totsu_amd.sparse import SpTile
    assert int(os.environ.get("WORLD_SIZE", "1")) == 1, "the sparse workloads run on one GPU"
    t_gen0 = time.perf_counter()
    if a.workload == "sparse-lp":
        inst = sparse_lp_instance(a.n or 16384)
    else:
        inst = sparse_sdp_instance(a.k)
    n, m = inst["n"], inst["m"]
    t_host = time.perf_counter() - t_gen0
    t0 = time.perf_counter()
    spt = SpTile.from_csc_arrays(m, n, inst["colptr"], inst["rowidx"], inst["vals"])
    lib.thip_sync()
    t_build = time.perf_counter() - t0
    info = spt.info()
    p = T.SolverParam()
    p.max_iter = None
    p.eps_acc = 0.0
    p.eps_inf = 0.0
    p.state_arith = a.state
    fs = T.FusedSolver(n, m, spt, inst["b"], inst["c"], inst["seg_type"], inst["seg_len"], p, a.schedule)
    # what THIS box streams: a bare read of as many bytes as one product reads
    box_read = None
    nbytes = min(info["bytes_per_product"], 20_000_000_000) // 16 * 16
    if nbytes >= 1 << 20:
        scratch = T.DeviceBuffer(nbytes // 4)
        pb, pa = C.c_float(), C.c_float()
        lib.thip_stream_probe(scratch.ptr, nbytes, 5, C.byref(pb), C.byref(pa))
        scratch.free()
        box_read = {"bytes": nbytes, "best_GBps": nbytes / (pb.value * 1e-3) / 1e9, "avg_GBps": nbytes / (pa.value * 1e-3) / 1e9}
    import torch
    fs.run(a.warmup, poll_every=max(a.warmup, 1))
    torch.cuda.synchronize()
    prof_period = max(1, a.steps // 32)
    lib.thip_prof_enable(0 if os.environ.get("THIP_BENCH_NO_PROF") else prof_period)
    t0 = time.perf_counter()
    r = fs.run(a.steps, poll_every=a.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    nl, tot_ms = C.c_int64(), C.c_double()
    lib.thip_prof_read(C.byref(nl), C.byref(tot_ms))
    npsd, psd_ms = C.c_int64(), C.c_double()
    lib.thip_prof_read_psd(C.byref(npsd), C.byref(psd_ms))
    lib.thip_prof_enable(0)
    assert r.state == _lib.ST_RUNNING and r.iters == a.warmup + a.steps, (r.state, r.iters)
    # (tau may sit at its clamp 0 for a while -- this LP's does from iteration 5 on, on every schedule and on the dense matrix
    # alike -- and the convergence criteria are then reported as inf: solver.rs:573-656 evaluates the infeasibility pair instead)
    xi, yi = fs.iterate()
    assert math.isfinite(r.tau) and np.isfinite(xi).all() and np.isfinite(yi).all(), "iterate blew up"
    passes, bytes_per_pass = fs.passes()
    # one product = one launch over the stored entries (8 B each; 4 B in a tile held dense, without indices) + its in-vectors
    # (2 per product) and the slices' partial sums
    vec_T = 4 * (2 * m + 2 * info["slices_t"] * n)
    vec_N = 4 * (2 * n + 2 * info["slices_n"] * m)
    bytes_per_launch = bytes_per_pass + 0.5 * (vec_T + vec_N)
    avg_ms = tot_ms.value / max(nl.value, 1)
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if nl.value else 0.0
    iters_per_s = a.steps / elapsed
    roofline = {
        "bound": "hbm",
        "kernel": "sp_tile_k<T> / sp_tile_k<N> (thip_sptile.hip): A^T [v x_y] and A [u x_x'] from the ONE tiled copy, 16-byte loads of "
                  "{value, local row | local column} entries (values alone in full tiles), LDS accumulators; averaged over both launches of an iteration",
        "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
        "traffic": None, "traffic_source": None,
        "timer": "hip_events on the launch stream around both product launches of every %s iteration of the timed region (thip_prof_*)"
                 % ("" if prof_period == 1 else "%d-th" % prof_period),
        "bytes_per_launch": bytes_per_launch, "bytes_of_entries_per_launch": bytes_per_pass,
        "vector_bytes_per_launch": 0.5 * (vec_T + vec_N),
        "avg_launch_ms": avg_ms, "launches_timed": nl.value, "passes_over_A_per_iter": passes,
        "bytes_per_stored_entry_and_iteration": bytes_per_pass * passes / max(info["nnz_stored"], 1),
        "dense_tiles": info["dense_tiles"], "tiles": info["tiles"], "indexed_entries": info["indexed_entries"],
        "stored_entries": info["nnz_stored"],
        "box_read_GBps": box_read["best_GBps"] if box_read else None,
        "frac_of_box_read": (achieved / box_read["best_GBps"]) if (box_read and nl.value) else None,
    }
    prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(prof):
        try:
            tr = json.load(open(prof))
            key = "%s_n%d_m%d_%s" % (a.workload, n, m, fs.schedule_in_use())
            if key in tr:
                roofline["traffic"] = tr[key]["hbm_bytes_per_launch"]
                roofline["traffic_source"] = ("stored PMC run (profiles/hbm_traffic.json: %s), not a counter of this run"
                                              % tr[key].get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE"))
        except Exception:
            pass
    roofline_eig = None
    if a.workload == "sparse-sdp" and npsd.value:
        roofline_eig = eig_record(a.k, psd_ms.value / npsd.value, npsd.value, 1e3 * elapsed / a.steps)
    out = {
        "metric": "solver iterations/sec, %s" % ("sparse LP (l1reg_lp construction)" if a.workload == "sparse-lp"
                                                 else "sparse SDP (partitioning_sdp construction)"),
        "value": iters_per_s, "unit": "iter/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (the reference example's construction, sample points / edge weights from numpy's generator, seed 0)",
        "state_arith": a.state,
        "config": {"workload": inst["what"], "schedule": fs.schedule_in_use(), "schedule_asked": a.schedule,
                   "passes_over_A_per_iter": passes, "storage": "one tiled copy (thip_sptile): 4096 x 4096 tiles, 8 B per entry",
                   "sptile": info, "host_gen_seconds": round(t_host, 2), "tile_build_and_upload_seconds": round(t_build, 2),
                   "parallelism": "none: one GPU"},
        "roofline": roofline, "roofline_eig": roofline_eig,
        "objective_gate": {"tolerance": 1e-4, "this_run": None,
                           "asserted_in_tests": "tests/test_gpu_sparse.py::test_sparse_%s_workload_iterates_vs_oracle (iterates 0, 1, 2, 9 "
                                                "of this construction against the f64 oracle on the dense-ified matrix), "
                                                "::test_sparse_sweep_converges_to_the_dense_answer_multi_tile"
                                                % ("lp" if a.workload == "sparse-lp" else "sdp")},
    }
    fs.destroy()
    # ---- time to eps (default 1e-3: the eps_acc the reference runs its f32 backend at, benchmark_lp/src/main.rs:62-65) ----
    to_eps = 1e-3 if (a.to_eps is None and not a.no_to_eps) else a.to_eps
    if to_eps is not None and not a.no_to_eps:
        p2 = T.SolverParam()
        p2.eps_acc = to_eps
        p2.state_arith = a.state
        budget = a.to_eps_budget if a.to_eps is not None else min(a.to_eps_budget, 240.0)      # (the default leg stays short)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fs3 = T.FusedSolver(n, m, spt, inst["b"], inst["c"], inst["seg_type"], inst["seg_len"], p2, a.schedule)
        while True:
            r2 = fs3.run(5000, poll_every=100)
            sys.stderr.write("to-eps[%s]: iter %d state %d cri %.3e %.3e %.3e t %.1f s\n"
                             % (a.workload, r2.iters + 1, r2.state, r2.cri[0], r2.cri[1], r2.cri[2], time.perf_counter() - t0))
            if r2.state != _lib.ST_RUNNING or time.perf_counter() - t0 > budget:
                break
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        done = r2.state != _lib.ST_RUNNING
        rec = {"eps_acc": to_eps, "seconds": dt, "iterations": int(r2.iters) + 1, "state": int(r2.state) if done else -1,
               "budget_seconds": budget, "criteria": [float(v) for v in r2.cri], "iters_per_s": (int(r2.iters) + 1) / dt}
        if r2.state == _lib.ST_OK:
            xs, ys = fs3.solution()
            pobj = float(inst["c"].astype(np.float64) @ xs.astype(np.float64))
            dobj = -float(inst["b"].astype(np.float64) @ ys.astype(np.float64))
            rec["primal_objective"], rec["dual_objective"] = pobj, dobj
            out["objective_gate"]["this_run"] = {"what": "primal vs dual objective of THIS run's answer (f64 dot products on the host)",
                                                 "rel_gap": abs(pobj - dobj) / max(1.0, abs(pobj))}
            # north_star's gate -- the f64 CPU reference's objective within 1e-4 -- where the oracle's answer for this very
            # instance is on file (tools/sparse_sdp_oracle_objective.py: 10 minutes of 8 CPU threads for k = 500)
            try:
                ev = json.load(open(os.path.join(ROOT, "profiles", "r06_sparse_sdp_oracle_objective.json")))
                if a.workload == "sparse-sdp" and ev["workload"] == inst["what"] and ev["eps_acc"] == to_eps:
                    out["objective_gate"]["this_run"].update({
                        "oracle_primal_objective": ev["primal_objective"], "oracle_iterations": ev["iterations"],
                        "rel_diff_vs_f64_oracle": abs(pobj - ev["primal_objective"]) / max(1.0, abs(ev["primal_objective"])),
                        "oracle_source": "profiles/r06_sparse_sdp_oracle_objective.json (oracle through its sparse user-operator, same instance, same eps_acc)"})
            except Exception:
                pass
        out["time_to_eps"] = rec
        fs3.destroy()
    # ---- A/B: the same instance on round 5's two CSR copies (carried schedule: 2 dual gathers per iteration) ----
    if not a.no_two_copy:
        try:
            import scipy.sparse as sp
            t0 = time.perf_counter()
            A = sp.csc_matrix((inst["vals"], inst["rowidx"], inst["colptr"]), shape=(m, n))
            fs2 = T.FusedSolver(n, m, A, inst["b"], inst["c"], inst["seg_type"], inst["seg_len"], p, "carried", sparse_two_copies=True)
            t_conv = time.perf_counter() - t0
            del A
            fs2.run(a.warmup, poll_every=max(a.warmup, 1))
            e2 = 1e30
            for _ in range(3):              # (best of three regions: this leg has come out 4x slow on its first region now and then)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fs2.run(a.steps, poll_every=a.steps)
                torch.cuda.synchronize()
                e2 = min(e2, time.perf_counter() - t0)
            out["two_copy_csr"] = {"value": a.steps / e2, "unit": "iter/s", "ms_per_step": 1e3 * e2 / a.steps,
                                   "what": "round 5's form on the same instance: CSR of A and CSR of A^T (two copies of the values), carried "
                                           "schedule, 32 B per stored entry and iteration (thip_spmv_csr); host conversion + upload %.1f s" % t_conv,
                                   "speedup_of_the_tiled_copy": iters_per_s / (a.steps / e2)}
            fs2.destroy()
        except Exception as e:              # the optional leg never costs the line
            out["two_copy_csr"] = {"error": repr(e)}
    spt.free()
    out["cpu_baseline"] = None if a.no_cpu else cpu_baseline_sparse(inst)
    return out, rank, (lambda: None)


def run_trait(a, inst, n, wl, t_gen, rank):
    """--path trait: iterations/sec of the trait-level drop-in (no fused loop, no alias types) on the same instance"""
    import ctypes as C
    assert int(os.environ.get("WORLD_SIZE", "1")) == 1 and a.workload in ("lp", "socp")
    so = os.path.join(ROOT, "totsu_amd", "lib", "libtotsu_trait_host.so")
    if not os.path.exists(so):
        raise ImportError("%s is missing -- run __graft_entry__.build()" % so)
    host = C.CDLL(so)
    out = (C.c_double * 2)()
    ref = 1 if a.trait_cones == "reference" else 0
    m = inst.m
    if a.workload == "lp":
        host.thost_lp.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        rc = host.thost_lp(n, m, inst.mat_a.ptr, inst.vec_b.ptr, inst.vec_c.ptr, a.steps, ref, out)
    else:
        host.thost_socp.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                    C.c_void_p]
        rc = host.thost_socp(n, a.cones, 99, inst.mat_a.ptr, inst.vec_b.ptr, inst.vec_c.ptr, a.steps, ref, out)
    assert rc == 0, "trait host failed (%d)" % rc
    sec = out[0]
    phys = 6 * 4.0 * m * n            # the reference's op sequence: 6 GEMV passes over A per iteration
    res = {
        "metric": "solver iters/sec, trait-level drop-in path (Solver::solve call by call; NOT the headline)",
        "path": "trait", "trait_cones": a.trait_cones,
        "value": 1.0 / sec, "unit": "iter/s", "n_gpus": 1, "steps": a.steps, "warmup": max(a.steps // 4, 1),
        "ms_per_step": 1e3 * sec, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (counter-based generator on device, seed 0)",
        "config": {"workload": wl, "schedule": "reference op sequence, one C-ABI call per LinAlg call",
                   "host": "examples/trait_host.cpp (compiled C++ over include/totsu_f32hip.hpp)",
                   "init_seconds": out[1], "gen_seconds": round(t_gen, 3)},
        "roofline": {"bound": "hbm", "achieved": phys / sec / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": phys / sec / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                     "what": "6 passes x 4 m n bytes per iteration / wall time per iteration (whole loop, not one kernel)"},
        "cpu_baseline": None,
    }
    inst.free()
    return res, rank, (lambda: None)


HBM_BYTES = 288e9               # MI355X: 288 GB of HBM3E per GPU (the dry run's stand-in for torch.cuda.mem_get_info)


def shard_plan(a, world, rank, cols, row_leg, mixed_leg, hbm_total=None):
    """What THIS rank will hold in HBM, computed BEFORE anything is allocated (the shapes are the real ones: the same
    synth.shard_cols / shard_cones / LpInstance row split the instances use), and asserted against the device's memory so
    that a run cannot die in hipMalloc after ten minutes of generation.  Terms: the shard of A; the row-sharded extra
    leg's shard (column-sharded runs time both partitionings); the library's padded f32 copies (thip_solver.hip ensure_apad:
    made when the leading dimension is not a multiple of 16 floats AND the copy is under a third of the free memory, so it
    is counted at that cap); the 16-bit copy of a bf16 / f16 / mixed run; partial sums and vectors (< 1 % of A)."""
    from totsu_amd import synth
    n = a.n or ({"socp": 50_000, "lp": 10_000, "sdp": 2000}[a.workload])
    if a.workload == "socp":
        m_total = a.cones * 100
    elif a.workload == "lp":
        m_total = 2 * n
    else:
        m_total = a.k * (a.k + 1) // 2
    if cols:
        c0, c1 = synth.shard_cols(n, world, rank)
        m_loc, n_loc = m_total, c1 - c0
    elif a.workload == "socp":
        c0, c1 = synth.shard_cones(a.cones, world, rank)
        m_loc, n_loc = (c1 - c0) * 100, n
    elif a.workload == "lp":
        base, rem = divmod(m_total, world)
        m_loc, n_loc = base + (1 if rank < rem else 0), n
    else:
        m_loc, n_loc = m_total, n
    esz = 2 if (a.bf16_direct or a.f16_direct) else 4
    plan = {"rank": rank, "world": world, "partition": "columns" if cols else "rows", "rows": m_loc, "cols": n_loc,
            "A_bytes": esz * m_loc * n_loc}
    solvers = 1 + (1 if a.to_eps is not None else 0)          # the timed solver and the time-to-eps one share A, not their copies

    def pad_copy(m_, n_):
        return 0 if m_ % 16 == 0 or esz == 2 else 4 * ((m_ + 15) // 16 * 16) * n_
    extra = solvers * pad_copy(m_loc, n_loc)
    if a.a_storage != "f32" and esz == 4:
        extra += solvers * 2 * ((m_loc + 7) // 8 * 8) * n_loc      # the library's 16-bit copy of an f32 A
    if mixed_leg:
        extra += 2 * ((m_loc + 7) // 8 * 8) * n_loc
    leg = 0
    if row_leg and cols:
        if a.workload == "socp":
            r0, r1 = synth.shard_cones(a.cones, world, rank)
            mr = (r1 - r0) * 100
        else:
            base, rem = divmod(m_total, world)
            mr = base + (1 if rank < rem else 0)
        leg = 4 * mr * n + pad_copy(mr, n)
        plan["row_leg_rows"] = mr
    small = int(0.01 * plan["A_bytes"]) + 64 * 4 * (m_loc + n_loc) + (64 << 20)
    plan.update({"library_copies_bytes": extra, "row_leg_bytes": leg, "vectors_and_partials_bytes": small})
    plan["total_bytes"] = plan["A_bytes"] + extra + leg + small
    total = hbm_total if hbm_total else HBM_BYTES
    plan["hbm_bytes"] = total
    plan["fits"] = plan["total_bytes"] <= 0.94 * total
    return plan


def dry_run(a, rank, world, dist, allreduce_host, cols, emu, mixed_leg):
    """--dry-run: the control path of a run WITHOUT a GPU or the HIP library -- the ranks are real processes (gloo), the
    column-shard agreement, the shard plan and its memory assertion are the real code, the timed region is the contract's
    barrier / max-over-ranks bracket around a sleep, and rank 0 prints ONE line with the keys of a real one (value null).
    What a first 8-GPU run can get wrong before its first kernel is exactly this part (tests/test_dist_cpu.py)."""
    import torch
    plan = shard_plan(a, emu or world, rank, cols, bool(cols and world > 1 and not a.no_row_leg), mixed_leg)
    fits = float(allreduce_host(np.array([0.0 if plan["fits"] else 1.0]))[0]) == 0.0
    plans = None
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, plan)
        plans = gathered
    else:
        plans = [plan]
    if not fits:
        raise SystemExit("bench.py: the shard plan does not fit the device on some rank: %s"
                         % json.dumps([p for p in plans if not p["fits"]][:2]))

    def barrier():
        if world > 1:
            dist.barrier()
    if os.environ.get("THIP_DRY_HANG_RANK") == str(rank):      # (test hook: this rank never reaches the barrier)
        time.sleep(3600)
    _PARTIAL["out"] = {"metric": "DRY RUN (no GPU, no kernels): control path only", "dry_run": True, "value": None, "n_gpus": world,
                       "config": {"hbm_plan": plans}}
    barrier()
    t0 = time.perf_counter()
    time.sleep(0.001 * a.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    n = a.n or ({"socp": 50_000, "lp": 10_000, "sdp": 2000}[a.workload])
    out = {"metric": "DRY RUN (no GPU, no kernels): control path only", "dry_run": True, "value": None, "unit": "iter/s",
           "n_gpus": world, "physical_gpus": 0, "rccl_ranks": None, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": 1e3 * elapsed / max(a.steps, 1), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f32", "data": "none",
           "config": {"workload": a.workload, "n": n, "schedule": "sweep" if cols else ("carried" if a.schedule == "sweep" and world > 1 else a.schedule),
                      "schedule_asked": a.schedule, "rows_per_gpu": plan["rows"], "cols_per_gpu": plan["cols"],
                      "parallelism": ("column-sharded A x%d" % world) if cols else ("row-sharded A x%d" % world if world > 1 else "none"),
                      "collective": "gloo (dry run)", "hbm_plan": plans},
           "value_partitioning": ("column-sharded (one-pass schedule, one all-reduce of 2 m floats per iteration)" if cols
                                  else ("none (one GPU)" if world == 1 else "row-sharded (north_star's form: cone-aligned row blocks, "
                                        "2-pass schedule, all-reduce of A^T y per transposed product)")),
           "value_column_sharded": None, "value_row_sharded": None,
           "roofline": None, "cpu_baseline": None, "objective_gate": None,
           "row_sharded": ({"rows_per_gpu": plan.get("row_leg_rows"), "value": None} if plan["row_leg_bytes"] else None),
           "time_to_eps": None}
    return out


def spawn_ranks(a):
    """`python bench.py --gpus N` outside a launcher: start the N ranks here, the way the driver's multi-GPU command does
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...).  Rank 0's JSON line goes to
    this process's stdout unchanged; the exit code is the launcher's."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without a launcher: running %s\n" % (a.gpus, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


# the line as far as it has got (rank 0): what the watchdog prints when a later leg of a multi-rank run never comes back
_PARTIAL = {"out": None, "fd": None, "timer": None}


def _watchdog():
    """N > 1 only: the legs behind the timed region (the row-sharded leg, time-to-eps) each hold collectives, and a first run on
    a node nobody has had may hang in one.  After --watchdog seconds rank 0 prints the line with whatever it holds (the timed
    region's value is complete by then or the line says it is not) and leaves; the launcher then takes the other ranks down.
    A timer THREAD, not SIGALRM: a Python signal handler does not run while the main thread sits inside a collective's C call."""
    out = _PARTIAL["out"]
    rank = int(os.environ.get("RANK", "0"))
    sys.stderr.write("bench.py: watchdog on rank %d: a leg did not come back; %s\n"
                     % (rank, "printing the line as far as it got" if (rank == 0 and out) else "leaving"))
    sys.stderr.flush()
    if rank == 0 and out is not None and _PARTIAL["fd"] is not None:
        out["watchdog"] = "a leg behind the timed region did not return within the watchdog's time: this line is partial"
        try:
            os.write(_PARTIAL["fd"], (json.dumps(out, default=repr) + "\n").encode())
        except Exception:
            pass
    os._exit(3 if out is None else 0)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    # RCCL prints a version banner on stdout when a communicator is created: keep fd 1 clean for the ONE JSON line
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and a.watchdog > 0:
        import threading
        _PARTIAL["fd"] = saved_stdout
        _PARTIAL["timer"] = threading.Timer(a.watchdog, _watchdog)
        _PARTIAL["timer"].daemon = True
        _PARTIAL["timer"].start()
    try:
        out, rank, cleanup = run(a)
    finally:
        sys.stdout.flush()
        try:
            # RCCL's banner goes through C stdio: it sits in libc's buffer (stdout is a pipe, so fully buffered)
            # until exit -- flush it to stderr NOW, while fd 1 still points there
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        if _PARTIAL["timer"] is not None:
            _PARTIAL["timer"].cancel()
            _PARTIAL["fd"] = None
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)
    cleanup()


def run(a):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist
    if a.gpus != world and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s): WORLD_SIZE wins\n" % (a.gpus, world))
    n_dev = world if a.dry_run else max(torch.cuda.device_count(), 1)
    if a.dry_run:
        a.collective, a.no_cpu, a.no_gate = "gloo", True, True
    shared_gpu = world > n_dev
    if shared_gpu and a.collective not in ("gloo", "oneshot"):
        # RCCL refuses two ranks on one device: the ranks share the GPU(s) and the all-reduce is staged through the host
        if rank == 0:
            sys.stderr.write("bench.py: %d ranks on %d GPU(s): --collective %s -> gloo (staged through host memory; "
                             "plumbing mode, not a performance configuration)\n" % (world, n_dev, a.collective))
        a.collective = "gloo"
    if shared_gpu and (a.schedule == "sweep" or a.shard == "cols"):
        # the one-pass kernel is persistent and needs every CU of its GPU: two ranks on one device would wait for each
        # other's workgroups (bounded, then an error).  Ranks sharing a GPU are a plumbing mode: carried schedule, row shards
        if rank == 0:
            sys.stderr.write("bench.py: ranks share a GPU: --schedule %s / --shard %s -> carried / rows\n" % (a.schedule, a.shard))
        a.schedule = "carried" if a.schedule == "sweep" else a.schedule
        a.shard = "rows"
    # the process group (bootstrap, host-side reductions, barriers): RCCL needs one device per rank
    pg_gloo = a.collective == "gloo" or shared_gpu
    if pg_gloo:
        local_rank = local_rank % n_dev      # ranks may share a GPU
    if not a.dry_run:
        torch.cuda.set_device(local_rank)
    use_dist = world > 1 or a.force_collective
    emu = a.emulate_world if (a.emulate_world > 1 and world == 1 and not a.force_collective) else 0
    if emu:
        a.no_to_eps, a.no_gate, a.no_cpu = True, True, True
    if use_dist and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if pg_gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import totsu_amd as T
    from totsu_amd import _lib, synth
    from totsu_amd._lib import lib
    if not a.dry_run:
        _lib.init(local_rank)      # the library launches on its own non-blocking stream (thip_get_stream)

    def allreduce_host(v, op="sum"):
        if not use_dist:
            return v
        t = torch.from_numpy(np.ascontiguousarray(v).copy())
        if not pg_gloo:
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        return t.cpu().numpy()

    if a.workload in ("sparse-lp", "sparse-sdp"):
        return run_sparse(a, rank, T, lib, _lib)
    t_gen0 = time.perf_counter()
    cols = (use_dist or emu) and a.path == "fused" and a.workload in ("socp", "lp") and a.a_storage == "f32" \
        and (a.shard == "cols" or (a.shard == "auto" and a.schedule == "sweep")) and not (a.bf16_direct or a.f16_direct)
    if cols:
        # every rank must be able to run the one-pass kernel on its block (256 CUs in 8 XCDs, a shape the kernel takes):
        # agree BEFORE anything column-sharded is built -- a rank that found out in its solver's init would leave the
        # others waiting in their first all-reduce
        import ctypes as C_
        nn = a.n or (50_000 if a.workload == "socp" else 10_000)
        mm = a.cones * 100 if a.workload == "socp" else 2 * nn
        c0_, c1_ = synth.shard_cols(nn, emu or world, rank)
        ok_ = C_.c_int(0)
        if a.dry_run:               # (the probe's verdict is a stub here: yes, unless the test names this rank)
            ok_.value = 0 if os.environ.get("THIP_DRY_PROBE_NO") == str(rank) else int(c1_ - c0_ >= 40)
        else:
            lib.thip_sweep_probe(mm, c1_ - c0_, mm, {"f32": 0, "bf16": 1, "f16": 2}.get(a.a_storage, 0), C_.byref(ok_))
        from totsu_amd.parallel import agree_on_column_shards
        all_ok = agree_on_column_shards(ok_.value != 0, allreduce_host, world) if use_dist else ok_.value
        if not all_ok:
            if rank == 0:
                sys.stderr.write("bench.py: the one-pass kernel cannot run on every rank: row shards, carried schedule\n")
            cols = False
            a.schedule = "carried" if a.schedule == "sweep" else a.schedule
    # the default line's legs (decided before anything is built: the memory plan below counts their copies)
    mixed_leg = False
    n_ = a.n or ({"socp": 50_000, "lp": 10_000, "sdp": 2000}[a.workload])
    if a.to_eps is None and not a.no_to_eps and not emu and a.workload == "socp" and n_ == 50_000 and a.cones == 1000 \
            and a.a_storage == "f32" and a.path == "fused":
        a.to_eps = 1e-3
        # the default line (the driver's command) also times the mixed f16 -> f32 solve to the same stopping test
        mixed_leg = not a.no_mixed_leg and world == 1
    if a.mixed_leg and a.to_eps is not None and a.a_storage == "f32" and world == 1:
        mixed_leg = True
    if a.dry_run:
        out = dry_run(a, rank, world, dist, allreduce_host, cols, emu, mixed_leg)

        def cleanup_dry():
            if use_dist:
                dist.barrier()
                dist.destroy_process_group()
        return out, rank, cleanup_dry
    # what this rank is about to allocate, against what its device has: every rank or none goes on
    plan = shard_plan(a, emu or world, rank, cols, bool(cols and use_dist and not a.no_row_leg), mixed_leg,
                      hbm_total=float(torch.cuda.mem_get_info()[1]))
    if float(allreduce_host(np.array([0.0 if plan["fits"] else 1.0]))[0]) != 0.0:
        raise SystemExit("bench.py: rank %d: the shard plan does not fit every rank's device (this rank: %s)" % (rank, json.dumps(plan)))
    if a.workload == "socp" and cols:
        n = a.n or 50_000
        inst = synth.SocpInstanceCols(n, a.cones, 99, seed=0, rank=rank, world=emu or world, allreduce_host=allreduce_host)
        wl = "random dense SOCP n=%d, %d second-order cones of 1+99 rows (m=%d), f32" % (n, a.cones, inst.m_total)
    elif a.workload == "lp" and cols:
        n = a.n or 10_000
        inst = synth.LpInstanceCols(n, seed=0, rank=rank, world=emu or world)
        wl = "benchmark_lp dense LP n=%d m=%d, f32" % (n, inst.m_total)
    elif a.workload == "socp":
        n = a.n or 50_000
        inst = synth.SocpInstance(n, a.cones, 99, seed=0, rank=rank, world=emu or world, allreduce_host=allreduce_host)
        wl = "random dense SOCP n=%d, %d second-order cones of 1+99 rows (m=%d), f32" % (n, a.cones, inst.m_total)
    elif a.workload == "sdp":
        assert world == 1, "one PSD cone does not shard"
        n = a.n or 2000
        inst = synth.SdpInstance(n, a.k, seed=0)
        wl = "dense SDP n=%d, one PSD cone of order %d (sk=%d), f32" % (n, a.k, inst.m_total)
    else:
        n = a.n or 10_000
        if a.bf16_direct or a.f16_direct:
            a.a_storage = "f16" if a.f16_direct else "bf16"
        inst = synth.LpInstance(n, seed=0, rank=rank, world=emu or world,
                                bf16_direct="f16" if a.f16_direct else a.bf16_direct)
        wl = "benchmark_lp dense LP n=%d m=%d, f32" % (n, inst.m_total)
    lib.thip_sync()
    t_gen = time.perf_counter() - t_gen0

    if a.path == "trait":
        return run_trait(a, inst, n, wl, t_gen, rank)
    p = T.SolverParam()
    p.max_iter = None
    p.eps_acc = 0.0            # never terminates inside the timed region: every step does full work
    p.eps_inf = 0.0
    p.state_arith = a.state
    hook, coll = None, "none"
    if emu:
        hook, coll = ("spin", a.emulate_latency), "stand-in collective of %d us (rank 0's shard of a %d-GPU run iterated alone)" % (a.emulate_latency, emu)
    dev_pg = "cpu" if pg_gloo else "cuda"
    if use_dist and a.collective == "gloo":
        hook, coll = GlooAllreduce(torch, dist, lib), "gloo through host memory (plumbing test mode)"
    elif use_dist and a.collective == "oneshot":
        # slots for the longest message of the loop (n + the 1024 block partials), handles exchanged through the group
        import ctypes as C
        hb = (C.c_uint8 * 64)()
        lib.thip_oneshot_init(rank, world, max(n + 2048, 2 * ((inst.m + 63) // 64 * 64) + 2048 + 256), hb)
        mine = torch.frombuffer(bytearray(bytes(hb)), dtype=torch.uint8).to(dev_pg)
        allh = [torch.zeros(64, dtype=torch.uint8, device=dev_pg) for _ in range(world)]
        dist.all_gather(allh, mine)
        blob = b"".join(bytes(t.cpu().numpy().tobytes()) for t in allh)
        lib.thip_oneshot_connect((C.c_uint8 * (64 * world)).from_buffer_copy(blob))
        dist.barrier()
        hook, coll = "oneshot", ("one-shot all-reduce over peer-mapped buffers (hipIpc), %d rank(s)%s"
                                 % (world, " sharing %d GPU(s)" % n_dev if shared_gpu else ""))
    elif use_dist:
        if a.collective == "rccl":
            # every rank takes part in every collective below, whatever fails locally, so that no rank is left waiting
            import ctypes as C
            idbuf = (C.c_uint8 * 128)()
            ok = 1
            if rank == 0:
                try:
                    lib.thip_comm_unique_id(idbuf)
                except Exception as e:
                    ok = 0
                    sys.stderr.write("native RCCL unavailable (%r)\n" % (e,))
            t = torch.zeros(129, dtype=torch.uint8, device="cuda")
            if rank == 0:
                t[0] = ok
                t[1:] = torch.frombuffer(bytearray(bytes(idbuf)), dtype=torch.uint8).cuda()
            dist.broadcast(t, src=0)
            th = t.cpu().numpy()
            mine = 0
            if int(th[0]) == 1:
                try:
                    lib.thip_comm_init(rank, world, (C.c_uint8 * 128).from_buffer_copy(th[1:].tobytes()))
                    mine = 1
                except Exception as e:
                    sys.stderr.write("thip_comm_init failed on rank %d (%r)\n" % (rank, e))
            flag = torch.tensor([mine], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                hook, coll = "rccl", "native RCCL all-reduce on the compute stream"
            else:
                if mine:
                    lib.thip_comm_destroy()
                sys.stderr.write("falling back to the torch.distributed all-reduce hook\n")
        if hook is None:
            hook, coll = TorchAllreduce(torch, dist), "torch.distributed.all_reduce hook (nccl)"
    n_loc = inst.n_local if cols else n          # the solver's n: this rank's columns when column-sharded
    fs = T.FusedSolver(n_loc, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, a.schedule,
                       allreduce=hook, a_storage={"f32": "f32", "f16": "f16", "mixed": "f16"}.get(a.a_storage, "bf16"),
                       overlap=None, col_shard=cols)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    OVM = {"off": 0, "on": 1, "pipeline": 2, "pipeline-inorder": 3}

    def tune_overlap(fs_):
        """row-sharded runs: where the all-reduce goes relative to the other work (thip_solver_set_overlap).  --overlap auto
        times 3 x 20 iterations per mode on the real communicator (untimed warm-up work, max over ranks) and keeps the fastest
        -- like the GEMV plan autotune.  Returns (mode, ms per iteration per mode or None, iterations spent)."""
        if a.overlap != "auto":
            lib.thip_solver_set_overlap(fs_.h, OVM[a.overlap])
            return a.overlap, None, 0
        best = {}
        cand = ("off", "on", "pipeline")
        for mode in cand * 3:
            lib.thip_solver_set_overlap(fs_.h, OVM[mode])
            barrier()
            t0_ = time.perf_counter()
            fs_.run(20, poll_every=20)
            barrier()
            best[mode] = min(best.get(mode, 1e30), time.perf_counter() - t0_)
        tt_ = torch.tensor([best[c] for c in cand], dtype=torch.float64, device=dev_pg)
        if use_dist:
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
        tl = [float(v) for v in tt_]
        pick = cand[tl.index(min(tl))]
        lib.thip_solver_set_overlap(fs_.h, OVM[pick])
        return pick, {c: 1e3 * v / 20 for c, v in zip(cand, tl)}, 20 * 3 * len(cand)

    overlap_pick, overlap_times = None, None
    if hook is not None and a.collective != "gloo" and not cols:
        overlap_pick, overlap_times, a.warmup_extra = tune_overlap(fs)
    # what THIS box streams: a bare non-temporal read of the solver's own A (the whole of it up to 20 GB), outside every
    # timed region -- the boxes of one pool differ by several percent, so the line carries its own yardstick
    box_read = None
    if not (a.bf16_direct or a.f16_direct) and inst.m * n_loc > 0:
        import ctypes as C
        pb, pa = C.c_float(), C.c_float()
        nbytes = min(4 * inst.m * n_loc, 20_000_000_000) // 16 * 16
        lib.thip_stream_probe(inst.mat_a.ptr, nbytes, 5, C.byref(pb), C.byref(pa))
        box_read = {"bytes": nbytes, "best_ms": pb.value, "avg_ms": pa.value,
                    "best_GBps": nbytes / (pb.value * 1e-3) / 1e9, "avg_GBps": nbytes / (pa.value * 1e-3) / 1e9}
    fs.run(a.warmup, poll_every=max(a.warmup, 1))
    barrier()
    # HIP events around the dominant kernel of the timed region.  An event pair costs the stream 3-5 us -- nothing in a 2.8 ms
    # iteration, 9 % of a 0.14 ms one (measured: LP 6 490 -> 7 085 iter/s without them) -- so a sample is timed: every launch
    # up to 32 steps (the default line), about 32 launches spread over a longer region
    prof_period = max(1, a.steps // 32)
    lib.thip_prof_enable(0 if os.environ.get("THIP_BENCH_NO_PROF") else prof_period)
    t0 = time.perf_counter()
    r = fs.run(a.steps, poll_every=a.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    import ctypes as C
    nl, tot_ms = C.c_int64(), C.c_double()
    lib.thip_prof_read(C.byref(nl), C.byref(tot_ms))
    npsd, psd_ms = C.c_int64(), C.c_double()
    lib.thip_prof_read_psd(C.byref(npsd), C.byref(psd_ms))
    lib.thip_prof_enable(0)
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev_pg)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert r.state == _lib.ST_RUNNING and r.iters == a.warmup + a.steps + getattr(a, "warmup_extra", 0), (r.state, r.iters)
    assert math.isfinite(r.tau) and math.isfinite(r.cri[0]), "iterate blew up"

    rccl_ranks = None
    if hook == "rccl":
        cnt = C.c_int(0)
        lib.thip_comm_count(C.byref(cnt))          # read back from the communicator (ncclCommCount), not from the env
        rccl_ranks = cnt.value
        assert rccl_ranks == world, (rccl_ranks, world)
    passes, bytes_per_pass = fs.passes()
    ovi = fs.overlap_info()
    lpp = ovi["launches_per_pass"]         # 2 when the column-split pipeline runs: a pass is two half-launches
    iters_per_s = a.steps / elapsed
    avg_ms = tot_ms.value / max(nl.value, 1)
    achieved = bytes_per_pass / lpp / (avg_ms * 1e-3) / 1e9 if nl.value else 0.0
    m_total = inst.m_total
    b_iter = 24.0 * m_total * n                              # 6 GEMVs x 4 m n bytes
    roofline = {
        "bound": "hbm",
        "kernel": ("sweep_k (one pass over A per iteration: per column both dots, the x_x / u updates and both axpys)"
                   if passes == 1 else "dual_gemv_k (one pass over the local A: y_N = A x_N and y_T = A^T x_T)"),
        "achieved": achieved,                                # physical bytes of one pass / avg launch duration
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS,
        "traffic": None,
        "traffic_source": None,
        "timer": ("hip_events on the launch stream around every launch of that kernel in the timed region (thip_prof_*)" if prof_period == 1
                  else "hip_events on the launch stream around every %d-th launch of that kernel in the timed region (thip_prof_*)" % prof_period),
        "bytes_per_launch": bytes_per_pass / lpp,
        "launches_per_pass": lpp,
        "avg_launch_ms": avg_ms,
        "launches_timed": nl.value,
        "passes_over_A_per_iter": passes,
        # reference op sequence = 6 GEMVs/iter (B_iter = 24 m n): rate the whole job sustains in those terms
        "algorithmic_GBps_per_gpu": b_iter * iters_per_s / 1e9 / world,
        "algorithmic_frac": b_iter * iters_per_s / 1e9 / world / HBM_PEAK_GBPS,
        # the same box's bare read of the same buffer (thip_stream_probe: non-temporal 16-byte loads, 8 in flight, nothing
        # else; best of 3 grids x 5 launches, HIP events): what "achievable" means on the box this line was measured on
        "box_read_GBps": box_read["best_GBps"] if box_read else None,
        "box_read_avg_GBps": box_read["avg_GBps"] if box_read else None,
        "box_read_bytes": box_read["bytes"] if box_read else None,
        "frac_of_box_read": (achieved / box_read["best_GBps"]) if (box_read and nl.value) else None,
    }
    prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(prof) and a.a_storage in ("f32", "bf16", "f16"):
        try:
            tr = json.load(open(prof))
            key = "%s_n%d_m%d_%s%s" % (a.workload, n, inst.m, fs.schedule_in_use(), "" if a.a_storage == "f32" else "_" + a.a_storage)
            if key in tr:
                roofline["traffic"] = tr[key]["hbm_bytes_per_launch"]
                roofline["traffic_source"] = ("stored PMC run (profiles/hbm_traffic.json: %s), not a counter of this run"
                                              % tr[key].get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE"))
                if "rocprof_avg_launch_ms" in tr[key]:
                    # the same kernel under rocprofv3 --kernel-trace --stats (includes the autotune / warm-up launches)
                    roofline["rocprof_avg_launch_ms"] = tr[key]["rocprof_avg_launch_ms"]
                    roofline["rocprof_frac"] = bytes_per_pass / (tr[key]["rocprof_avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
        except Exception:
            pass

    roofline_eig = None
    if a.workload == "sdp" and npsd.value:
        roofline_eig = eig_record(a.k, psd_ms.value / npsd.value, npsd.value, 1e3 * elapsed / a.steps)

    out = {
        "metric": "solver iters/sec + time-to-eps, dense SOCP n=50k (value = iters/sec; time_to_eps beside it)" if a.workload == "socp"
                  else "solver iterations/sec, dense %s" % a.workload.upper(),
        "a_storage": a.a_storage,
        "value": iters_per_s,
        "unit": "iter/s",
        "n_gpus": world,
        "physical_gpus": min(world, n_dev),
        "rccl_ranks": rccl_ranks,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32" if a.a_storage == "f32" else "f32 arithmetic on a 16-bit-STORED A (%s; rounded problem; not the headline)" % a.a_storage,
        "data": "synthetic (counter-based generator on device, seed 0; 'normal' entries are Irwin-Hall(4) sums scaled to "
                "unit variance, not exact Gaussians)",
        "state_arith": a.state,
        "config": {"workload": wl, "schedule": fs.schedule_in_use(), "schedule_asked": a.schedule, "passes_over_A_per_iter": passes,
                   "rows_per_gpu": inst.m, "cols_per_gpu": n_loc, "emulated_world": emu or None,
                   "parallelism": ("column-sharded A x%d, one all-reduce of the two N products (2 m floats) per iteration" % (emu or world)) if cols
                                  else ("none: one GPU holds the whole A" if hook is None
                                        else "row-sharded A x%d, all-reduce of A^T y" % (emu or world)), "collective": coll, "overlap": overlap_pick,
                   "overlap_mode_run": ovi["mode"], "overlap_split_col": ovi["split_col"], "overlap_autotune_ms_per_iter": overlap_times,
                   "gen_seconds": round(t_gen, 3), "gemv_plan": fs.gemv_plan(), "sweep_plan": fs.sweep_plan(), "a_storage": a.a_storage,
                   "hbm_plan": plan},
        # N > 1: which partitioning produced `value`, and BOTH rates at top level, so that a SCALE record cannot be read as
        # north_star's row form when it is the column form (the other one is filled by the row_sharded leg below)
        "value_partitioning": ("column-sharded (one-pass schedule, one all-reduce of 2 m floats per iteration)" if cols
                               else ("none (one GPU)" if hook is None else "row-sharded (north_star's form: cone-aligned row blocks, "
                                     "2-pass schedule, all-reduce of A^T y per transposed product)")),
        "value_column_sharded": iters_per_s if cols else None,
        "value_row_sharded": iters_per_s if (hook is not None and not cols) else None,
        "roofline": roofline,
        "roofline_eig": roofline_eig,
        # north_star: "same primal/dual objective as the f64 CPU reference within 1e-4 relative".  The f64 oracle runs the
        # full-size instance at ~0.24 iter/s (1e5 iterations = 5 days), so at this size the gate is (a) THIS run's answer
        # re-evaluated in f64 (`this_run`, filled after the time_to_eps leg) and (b) stored evidence, by reference
        "objective_gate": {"tolerance": 1e-4, "this_run": None, "stored_evidence": stored_objective_evidence(fs.schedule_in_use()),
                           "asserted_in_tests": "tests/test_gpu_solver.py::test_synth_socp_converges_to_oracle_objective[%s] (n = 500, "
                                                "this schedule vs the f64 oracle's objective); at the full size: "
                                                "tests/test_gpu_configs.py::test_c3_full_size_sweep_vs_oracle (iterates 0-2 of the "
                                                "1000-cone instance, 0-99 of the 328-cone sub-instance vs the oracle)"
                                                % fs.schedule_in_use()},
        "sweep_faults": fs.sweep_faults(),
    }
    _PARTIAL["out"] = out          # (the legs below add to it in place)

    if cols and use_dist and a.workload in ("socp", "lp") and not a.no_row_leg:
        # north_star / configs[4] name ROW blocks with the A^T y all-reduce overlapped on a side stream; the default at N > 1
        # is column blocks (one pass, one collective).  A short leg of the row-sharded carried run on the same ranks, same
        # transport, so that every multi-GPU line carries both figures
        def all_ranks_ok(ok_here):
            # a rank that failed locally (no memory for inst_r, a solver error) must not leave the others in a collective:
            # every stage of the leg that ends in one is entered only when EVERY rank got there
            return float(allreduce_host(np.array([0.0 if ok_here else 1.0]))[0]) == 0.0

        inst_r = fs_r = None
        stage, err_r = "build", None
        try:
            try:
                # (SocpInstance sums f over the ranks while it builds: the host sum below is one every rank reaches)
                if a.workload == "socp":
                    inst_r = synth.SocpInstance(n, a.cones, 99, seed=0, rank=rank, world=world, allreduce_host=None)
                    fh = allreduce_host(inst_r.vec_c_host)
                    lib.thip_h2d(inst_r.vec_c.ptr, np.ascontiguousarray(fh, dtype=np.float32).ctypes.data, n)
                    inst_r.vec_c_host = np.asarray(fh, dtype=np.float32)
                else:
                    inst_r = synth.LpInstance(n, seed=0, rank=rank, world=world)
                lib.thip_sync()
            except Exception as e:
                err_r = e
                if a.workload == "socp" and inst_r is None:
                    allreduce_host(np.zeros(n, dtype=np.float32))      # the sum the healthy ranks are in
            # (the solver's init holds a collective of its own -- the |A| column sums: entered only when every rank has its shard)
            if not all_ranks_ok(err_r is None):
                raise RuntimeError("the row-sharded leg's shard could not be built on every rank (this rank: %r)" % (err_r,))
            stage = "init"
            try:
                fs_r = T.FusedSolver(n, inst_r.m, inst_r.mat_a, inst_r.vec_b, inst_r.vec_c, inst_r.seg_type, inst_r.seg_len, p,
                                     "carried", allreduce=hook, overlap=None)
            except Exception as e:
                err_r = e
            if not all_ranks_ok(err_r is None):
                raise RuntimeError("the row-sharded leg's solver could not be set up on every rank (this rank: %r)" % (err_r,))
            stage = "run"
            pick_r, times_r, _ = tune_overlap(fs_r) if a.collective != "gloo" else (None, None, 0)
            fs_r.run(a.warmup, poll_every=max(a.warmup, 1))
            barrier()
            t0 = time.perf_counter()
            rr = fs_r.run(a.steps, poll_every=a.steps)
            barrier()
            el_r = time.perf_counter() - t0
            if use_dist:
                tt = torch.tensor([el_r], dtype=torch.float64, device=dev_pg)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el_r = float(tt.item())
            assert rr.state == _lib.ST_RUNNING and math.isfinite(rr.cri[0])
            out["row_sharded"] = {"value": a.steps / el_r, "unit": "iter/s", "ms_per_step": 1e3 * el_r / a.steps, "steps": a.steps,
                                  "schedule": fs_r.schedule_in_use(), "passes_over_A_per_iter": fs_r.passes()[0],
                                  "rows_per_gpu": inst_r.m, "overlap": pick_r, "overlap_mode_run": fs_r.overlap_info()["mode"],
                                  "overlap_autotune_ms_per_iter": times_r,
                                  "parallelism": "row-sharded A x%d (cone-aligned row blocks), all-reduce of A^T y per transposed "
                                                 "product: the partitioning north_star and configs[4] name" % world}
            out["value_row_sharded"] = out["row_sharded"]["value"]
        except Exception as e:          # the extra leg must never cost the line
            out["row_sharded"] = {"error": repr(e), "stage": stage}
        finally:
            if fs_r is not None:
                fs_r.destroy()
            if inst_r is not None:
                inst_r.free()

    def to_eps_leg(storage):
        """solve from x = 0 to the reference's stopping test at --to-eps with A streamed as `storage` (mixed: 16-bit passes to
        the same test on the ROUNDED matrix, then thip_solver_resume on the exact f32 matrix); returns (record, f64 gate)"""
        p2 = T.SolverParam()
        p2.eps_acc = a.to_eps
        p2.state_arith = a.state
        budget = {"hit": False}

        def run_to_end(fs):
            # in chunks, with a progress line on stderr: a run that hits an outer time limit still leaves its trail
            while True:
                r = fs.run(5000, poll_every=100)
                if rank == 0:
                    sys.stderr.write("to-eps[%s]: iter %d state %d cri %.3e %.3e %.3e t %.1f s\n"
                                     % (storage, r.iters + 1, r.state, r.cri[0], r.cri[1], r.cri[2], time.perf_counter() - t0))
                    sys.stderr.flush()
                if r.state != _lib.ST_RUNNING:
                    return r
                # every rank sees the same elapsed-time decision (rank 0's clock)
                over = time.perf_counter() - t0 > a.to_eps_budget
                if use_dist:
                    tt = torch.tensor([1 if over else 0], dtype=torch.int32, device=dev_pg)
                    dist.broadcast(tt, src=0)
                    over = bool(int(tt.item()))
                if over:
                    budget["hit"] = True
                    return r

        barrier()
        t0 = time.perf_counter()
        fs2 = T.FusedSolver(n_loc, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p2,
                            a.schedule, allreduce=hook, a_storage={"f32": "f32", "f16": "f16", "mixed": "f16"}.get(storage, "bf16"),
                            overlap=overlap_pick, col_shard=cols)
        r2 = run_to_end(fs2)
        barrier()
        phase1 = None
        if storage in ("mixed", "mixed-bf16") and r2.state == _lib.ST_OK:
            # the 16-bit passes have converged on the rounded matrix: finish on the exact one
            phase1 = {"seconds": time.perf_counter() - t0, "iterations": r2.iters + 1, "cri": list(r2.cri),
                      "schedule": fs2.schedule_in_use()}
            # the answer of the ROUNDED problem, for the record; its downloads are excluded from the reported seconds
            t_skip = time.perf_counter()
            x1, y1 = fs2.solution()
            phase1["primal_obj"] = float(inst.vec_c_host.astype(np.float64) @ x1.astype(np.float64))
            d1 = -float(inst.vec_b_host.astype(np.float64) @ y1.astype(np.float64))
            phase1["dual_obj"] = float(allreduce_host(np.array([d1], dtype=np.float32))[0]) if use_dist else d1
            t0 += time.perf_counter() - t_skip
            fs2.set_a_storage("f32")
            fs2.resume()
            r2 = run_to_end(fs2)
            barrier()
        rec = {"eps_acc": a.to_eps, "seconds": time.perf_counter() - t0, "iterations": r2.iters + 1,
               "state": r2.state, "cri": list(r2.cri), "a_storage": storage, "state_arith": a.state,
               "budget_s": a.to_eps_budget, "budget_hit": budget["hit"],
               "schedule": fs2.schedule_in_use(), "sweep_plan": fs2.sweep_plan(), "sweep_faults": fs2.sweep_faults(),
               "what": "wall time of the solve from the initial iterate (x = 0, tau = 1) to the reference's "
                       "stopping test at eps_acc (solver.rs:381-400), A resident in HBM, init (norms, "
                       "preconditioner, plan autotune%s) included"
                       % (", the 16-bit copy of A" if storage != "f32" else "")}
        if phase1:
            rec["f16_phase" if storage == "mixed" else "bf16_phase"] = phase1
        x, y = fs2.solution()
        if r2.state == _lib.ST_RUNNING and r2.tau > 0:
            # budget hit before the stopping test: the iterate is still the homogeneous one (solver.rs:397-400 scales by
            # 1/tau only on termination) -- scale it here so that the objectives and the f64 evaluation mean something
            x, y = x / np.float32(r2.tau), y / np.float32(r2.tau)
        pobj = float(inst.vec_c_host.astype(np.float64) @ x.astype(np.float64))
        dloc = -float(inst.vec_b_host.astype(np.float64) @ y.astype(np.float64))
        if cols:          # x is this rank's block of columns (c.x adds up over ranks), y is the whole dual vector everywhere
            pobj = float(allreduce_host(np.array([pobj], dtype=np.float32))[0]) if use_dist else pobj
            dobj = dloc
        else:
            dobj = float(allreduce_host(np.array([dloc], dtype=np.float32))[0]) if use_dist else dloc
        rec.update({"primal_obj": pobj, "dual_obj": dobj})
        gate = None
        if not a.no_gate and not (cols and a.workload != "socp") and not (a.bf16_direct or a.f16_direct):
            try:
                if a.workload == "lp":
                    gate = kkt_f64_lp(inst, x, y, allreduce_host)
                elif a.workload == "sdp":
                    gate = kkt_f64_sdp(inst, x, y)
                else:
                    gate = kkt_f64_cols(inst, x, y, allreduce_host) if cols else kkt_f64(inst, x, y, allreduce_host)
                gate.update({"eps_acc": a.to_eps, "gpu_criteria_f32": list(r2.cri), "state": r2.state})
            except Exception as e:                      # the checker must never break the bench line
                gate = {"error": repr(e)}
        fs2.destroy()
        return rec, gate

    if a.to_eps is not None:
        out["time_to_eps"], out["objective_gate"]["this_run"] = to_eps_leg(a.a_storage)
        if mixed_leg and out["time_to_eps"]["state"] == _lib.ST_OK:
            # BASELINE's metric has two halves; the second one has a lever the f32 headline does not use: the same solve with
            # the first phase streamed from an f16-stored copy of A (half the bytes per iteration), finished on the exact f32
            # matrix to the SAME stopping test.  Reported beside the f32 leg, never instead of it.
            try:
                rec, gate = to_eps_leg("mixed")
                rec["objective_gate_this_run"] = gate
                rec["vs_f32_leg"] = {"seconds_ratio": rec["seconds"] / out["time_to_eps"]["seconds"],
                                     "primal_obj_rel_diff": abs(rec["primal_obj"] - out["time_to_eps"]["primal_obj"])
                                     / (1.0 + abs(out["time_to_eps"]["primal_obj"]))}
                out["time_to_eps_mixed"] = rec
            except Exception as e:          # the extra leg must never cost the line (single GPU: no peer is left waiting)
                out["time_to_eps_mixed"] = {"error": repr(e)}

    if rank == 0 and world == 1 and not a.no_cpu and a.workload == "socp":
        import oracle as O
        # a fixed sample (default: the first 328 cones, A_sub 13 GB of f64) so that the figure reproduces; shrunk only
        # when the host's free memory cannot hold it
        cc = a.cpu_cones
        try:
            import psutil
            cc = max(8, min(cc, int(0.4 * psutil.virtual_memory().available / (8.0 * n * 100))))
        except Exception:
            pass
        out["cpu_baseline"] = cpu_baseline(n, a.cones, 99, 0, min(cc, a.cones))
    elif rank == 0:
        out["cpu_baseline"] = None

    fs.destroy()
    inst.free()
    if hook == "rccl":
        from totsu_amd.fused import comm_destroy
        comm_destroy()
    if hook == "oneshot":
        err = C.c_int(0)
        lib.thip_oneshot_error(C.byref(err))
        assert err.value == 0, "one-shot all-reduce timed out waiting for a peer"
        if use_dist:
            dist.barrier()               # nobody unmaps while a peer may still be reading
        lib.thip_oneshot_destroy()
    def cleanup():
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
    return out, rank, cleanup


if __name__ == "__main__":
    main()
