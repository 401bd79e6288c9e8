"""One-off measurement for SURVEY.md 8d ("final objective within 1e-4 relative of the f64 CPU result on the same
instance") at a size the CPU oracle still finishes: the synthetic SOCP of bench.py at n = 5000, 100 cones, solved to
the same eps_acc by the GPU (f32, and f16-stored passes then f32, through --schedule) and by the C oracle (f64, OpenMP) on the
host.  Prints one JSON line.
Usage: python tests/measure_objective_gap.py [n] [cones] [eps] [--schedule carried|sweep|fused] [--force-sweep]
  --schedule     the GPU schedule of both legs (round 5: the line bench.py cites is the one whose schedule matches the run's)
  --force-sweep  sweep_min_bytes = 0: the one-pass kernel also below the size where it is the default (128 MiB of A)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O                     # noqa: E402  (lives under tests/: the oracle is the checker here)


def host_problem(n, cones):
    """--host-bc: the instance with b and c formed on the HOST in f64 (bench.py's oracle_sub_instance over all cones) and
    rounded to f32 -- the matrix entries are the counter-based generator's on both sides, so the GPU legs (--gpu-only, on
    the GPU box) and the oracle leg (--cpu-only, anywhere: it needs no GPU) solve the same problem in two processes"""
    import bench
    f, A, b, _, _ = bench.oracle_sub_instance(n, cones, 99, 0, cones)
    return A, np.asarray(b, dtype=np.float32), np.asarray(f, dtype=np.float32)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("n", nargs="?", type=int, default=2000)
    ap.add_argument("cones", nargs="?", type=int, default=40)
    ap.add_argument("eps", nargs="?", type=float, default=1e-3)
    ap.add_argument("--schedule", default="carried", choices=["carried", "sweep", "fused"])
    ap.add_argument("--force-sweep", action="store_true")
    ap.add_argument("--host-bc", action="store_true", help="b and c formed on the host (see host_problem)")
    ap.add_argument("--cpu-only", action="store_true", help="with --host-bc: the f64 oracle leg alone (no GPU needed)")
    ap.add_argument("--gpu-only", action="store_true", help="with --host-bc: the GPU legs alone")
    ap.add_argument("--threads", type=int, default=0)
    ar = ap.parse_args()
    n, cones, eps = ar.n, ar.cones, ar.eps
    if ar.threads:
        O.set_num_threads(ar.threads)
    if ar.cpu_only:
        assert ar.host_bc
        a, b32, c32 = host_problem(n, cones)
        b, c = b32.astype(np.float64), c32.astype(np.float64)
        t0 = time.perf_counter()
        ro = O.solve_matop_cones(O.param(max_iter=10_000_000, eps_acc=eps), c, a, b, [O.CONE_SOC] * cones, [100] * cones)
        print(json.dumps({"instance": "synthetic SOCP n=%d, %d cones of 1+99 rows (m=%d), seed 0; b, c formed on the host in f64 and "
                                      "rounded to f32" % (n, cones, 100 * cones), "eps_acc": eps,
                          "cpu_oracle_f64": {"status": ro.status, "iterations": ro.iters + 1, "seconds": time.perf_counter() - t0,
                                             "threads": O.num_threads(), "host_cpu_count": os.cpu_count(),
                                             "primal_obj": float(c @ ro.x), "dual_obj": -float(b @ ro.y)}}))
        return
    import totsu_amd as T
    from totsu_amd import _lib, synth
    from totsu_amd.fused import DeviceBuffer
    _lib.init()
    inst = synth.SocpInstance(n, cones, 99, seed=0)
    if ar.host_bc:
        a, b32, c32 = host_problem(n, cones)
        assert np.array_equal(inst.mat_a.to_host()[:inst.m * inst.n].astype(np.float64), np.asarray(a))      # the same matrix, bit for bit
        inst.vec_b.free()
        inst.vec_c.free()
        inst.vec_b_host, inst.vec_c_host = b32, c32
        inst.vec_b, inst.vec_c = DeviceBuffer.from_host(b32), DeviceBuffer.from_host(c32)
    else:
        a = inst.mat_a.to_host()[:inst.m * inst.n].astype(np.float64)
    b, c = inst.vec_b_host.astype(np.float64), inst.vec_c_host.astype(np.float64)
    out = {"instance": "synthetic SOCP n=%d, %d cones of 1+99 rows (m=%d), seed 0" % (n, cones, inst.m), "eps_acc": eps,
           "schedule": ar.schedule, "sweep_forced_below_its_default_size": bool(ar.force_sweep)}
    kw = {"sweep_min_bytes": 0} if ar.force_sweep else {}
    p = T.SolverParam()
    p.eps_acc, p.max_iter = eps, None
    for mode in ("f32", "mixed"):
        fs = T.FusedSolver(inst.n, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, ar.schedule,
                           a_storage="f32" if mode == "f32" else "f16", **kw)
        sched1 = fs.schedule_in_use()
        t0 = time.perf_counter()
        r = fs.run(-1, poll_every=64)
        if mode == "mixed":
            fs.set_a_storage("f32")
            fs.resume()
            r = fs.run(-1, poll_every=64)
        dt = time.perf_counter() - t0
        x, y = fs.solution()
        out["gpu_" + mode] = {"state": r.state, "iterations": r.iters + 1, "seconds": dt, "cri": list(r.cri),
                              "schedule_in_use": [sched1, fs.schedule_in_use()], "sweep_faults": fs.sweep_faults(),
                              "primal_obj": float(c @ x.astype(np.float64)), "dual_obj": -float(b @ y.astype(np.float64))}
        fs.destroy()
        sys.stderr.write(json.dumps(out) + "\n")       # the CPU leg below can take very long: keep what is known
        sys.stderr.flush()
    if ar.gpu_only:
        print(json.dumps(out))
        return
    t0 = time.perf_counter()
    ro = O.solve_matop_cones(O.param(max_iter=10_000_000, eps_acc=eps), c, a, b, [O.CONE_SOC] * cones, [100] * cones)
    out["cpu_oracle_f64"] = {"status": ro.status, "iterations": ro.iters + 1, "seconds": time.perf_counter() - t0,
                             "threads": O.num_threads(), "primal_obj": float(c @ ro.x), "dual_obj": -float(b @ ro.y)}
    ref = out["cpu_oracle_f64"]["primal_obj"]
    for mode in ("f32", "mixed"):
        g = out["gpu_" + mode]
        g["rel_gap_to_cpu_primal"] = abs(g["primal_obj"] - ref) / abs(ref)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
