"""A few iterations of the carried schedule on the 1/8 row shard of BASELINE configs[2] (12 500 x 50 000) with the stand-in
collective (thip_test_spin_allreduce) in a given overlap mode -- meant to be run under
    rocprofv3 --kernel-trace --stats -d <dir> -- python tools/pipeline_probe.py --mode 2 --latency 60
whose kernel trace shows spin_k (the "collective", on the side stream) running under dual_gemv_k (the next half-launch).
Prints the iteration time."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import totsu_amd as T                   # noqa: E402
from totsu_amd import _lib, synth       # noqa: E402
from totsu_amd._lib import lib          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", type=int, default=2)
    ap.add_argument("--latency", type=int, default=60)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--cones", type=int, default=125)
    a = ap.parse_args()
    _lib.init(0)
    inst = synth.SocpInstance(50_000, 1000, 99, seed=0, first_cones=a.cones)
    p = T.SolverParam()
    p.eps_acc = 0.0
    fs = T.FusedSolver(inst.n, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, "carried",
                       allreduce=("spin", a.latency), overlap=a.mode)
    fs.run(20, poll_every=20)
    lib.thip_sync()
    t0 = time.perf_counter()
    fs.run(a.iters, poll_every=a.iters)
    lib.thip_sync()
    dt = (time.perf_counter() - t0) / a.iters
    print(json.dumps({"mode": a.mode, "latency_us": a.latency, "us_per_iteration": 1e6 * dt, "overlap": fs.overlap_info(),
                      "gemv_plan": fs.gemv_plan()}))
    fs.destroy()
    inst.free()


if __name__ == "__main__":
    main()
