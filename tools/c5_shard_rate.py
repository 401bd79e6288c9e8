"""One-off: the per-GPU rate of BASELINE.json's C5 (LP n = 200 000, m = 400 000 over 8 GPUs) measured on ONE GPU by
running rank 0's row shard (50 000 x 200 000 f32 = 40 GB, generated on the device) with the native RCCL all-reduce
in the loop at world size 1.  What is missing relative to the 8-GPU run is only the xGMI latency of the two 800 KB
all-reduces per iteration.  Usage: python tools/c5_shard_rate.py [steps] [overlap: off | on | pipeline]
                                  [spin latency_us: a stand-in collective of that latency instead of RCCL at world 1]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import totsu_amd as T                   # noqa: E402
from totsu_amd import _lib, synth       # noqa: E402
from totsu_amd._lib import lib          # noqa: E402
from totsu_amd.fused import comm_destroy, comm_init    # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    overlap = sys.argv[2] if len(sys.argv) > 2 else "off"
    spin = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[3] == "spin" else None
    _lib.init(0)
    if spin is None:
        comm_init(0, 1, lambda b: b)            # world size 1: rank 0's id needs no broadcast
    inst = synth.LpInstance(200_000, seed=0, rank=0, world=8)
    p = T.SolverParam()
    p.eps_acc, p.eps_inf, p.max_iter = 0.0, 0.0, None
    fs = T.FusedSolver(inst.n, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, "carried",
                       allreduce="rccl" if spin is None else ("spin", spin), overlap=overlap)
    fs.run(5, poll_every=5)
    lib.thip_sync()
    t0 = time.perf_counter()
    r = fs.run(steps, poll_every=steps)
    lib.thip_sync()
    dt = time.perf_counter() - t0
    passes, nbytes = fs.passes()
    print(json.dumps({"shard": "rank 0 of 8 of LP n=200000 m=400000: %d x %d f32 (%.1f GB)" % (inst.m, inst.n, nbytes / 1e9),
                      "iter_per_s": steps / dt, "ms_per_iter": 1e3 * dt / steps, "passes_per_iter": passes,
                      "GBps_over_the_iteration": passes * nbytes * steps / dt / 1e9, "gemv_plan": fs.gemv_plan(),
                      "overlap": fs.overlap_info(), "collective": "native RCCL, world 1" if spin is None else "stand-in, %d us" % spin,
                      "state": r.state, "tau": r.tau}))
    fs.destroy()
    inst.free()
    if spin is None:
        comm_destroy()


if __name__ == "__main__":
    main()
