"""An f64 certificate for the GPU's answer at the FULL size of BASELINE.json configs[2] (n = 50 000, 1000 cones, A 20 GB),
where the f64 CPU solver itself cannot be run to convergence (0.23 iter/s x 1e5 iterations).

The GPU solves to eps_acc (f32); x (n) and y (m) are downloaded; the host regenerates A block by block in f64 from the
counter-based generator (the same entries the device holds, widened) and evaluates, in f64,
    primal:  s = b - A x,  violation of s in K (distance-like: max(0, ||s_1|| - s_0) per cone)
    dual:    r = c + A^T y, ||r|| / (1 + ||c||),  violation of y in K* = K
    gap:     |c.x + b.y| / (1 + |c.x| + |b.y|)
For a primal-dual pair that is feasible to these tolerances the duality gap bounds the distance of c.x from the optimal
value: this is the size-independent form of "same objective as the f64 reference within 1e-4".
Usage: python tools/c3_f64_certificate.py [eps ...]      (default: 1e-3 1e-4; the solve is resumed from eps to eps)"""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O                      # noqa: E402  (checker only: regenerates A in f64)
import totsu_amd as T                   # noqa: E402
from totsu_amd import _lib, synth       # noqa: E402


def certificate(inst, x, y, block_cones=50):
    n, ni, rows = inst.n, inst.ni, 1 + inst.ni
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    b64, c64 = inst.vec_b_host.astype(np.float64), inst.vec_c_host.astype(np.float64)
    r = c64.copy()
    viol_p = viol_d = 0.0
    s_norm = 0.0
    for c0 in range(0, inst.n_cones, block_cones):
        nc = min(block_cones, inst.n_cones - c0)
        mb = nc * rows
        A = np.asarray(O.gen_matrix(mb, n, inst.seed, synth.STREAM_A, c0 * rows, 0, inst.m_total, 1, -1.0 / math.sqrt(n)))
        At = A.reshape(n, mb)                     # column-major (mb x n) seen row-major is A^T
        yb = y64[c0 * rows:(c0 + nc) * rows]
        s = (b64[c0 * rows:(c0 + nc) * rows] - At.T @ x64).reshape(nc, rows)
        r += At @ yb
        viol_p = max(viol_p, float(np.max(np.maximum(0.0, np.linalg.norm(s[:, 1:], axis=1) - s[:, 0]))))
        yy = yb.reshape(nc, rows)
        viol_d = max(viol_d, float(np.max(np.maximum(0.0, np.linalg.norm(yy[:, 1:], axis=1) - yy[:, 0]))))
        s_norm = max(s_norm, float(np.abs(s).max()))
    pobj, dobj = float(c64 @ x64), -float(b64 @ y64)
    return {
        "primal_obj_f64": pobj, "dual_obj_f64": dobj,
        "gap_rel": abs(pobj - dobj) / (1.0 + abs(pobj) + abs(dobj)),
        "objective_gap_rel_to_dual": abs(pobj - dobj) / max(abs(dobj), 1e-300),
        "dual_residual_rel": float(np.linalg.norm(r)) / (1.0 + float(np.linalg.norm(c64))),
        "primal_cone_violation_max": viol_p, "primal_cone_violation_rel_to_norm_b": viol_p / (1.0 + float(np.linalg.norm(b64))),
        "dual_cone_violation_max": viol_d, "slack_abs_max": s_norm,
    }


def main():
    eps_list = [float(a) for a in sys.argv[1:]] or [1e-3, 1e-4]
    _lib.init(0)
    if O.num_threads() < 32:
        O.set_num_threads(min(64, os.cpu_count() or 8))
    inst = synth.SocpInstance(int(os.environ.get("C3_N", "50000")), int(os.environ.get("C3_CONES", "1000")), 99, seed=0)
    p = T.SolverParam()
    p.eps_acc = eps_list[0]
    fs = T.FusedSolver(inst.n, inst.m, inst.mat_a, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, "carried")
    out = {"workload": "random dense SOCP n=%d, %d second-order cones of 1+99 rows (m=%d), f32 on the GPU; " % (inst.n, inst.n_cones, inst.m_total) +
                       "certificate evaluated in f64 on the host with A regenerated from the counter-based generator",
           "points": []}
    t0 = time.perf_counter()
    for k, eps in enumerate(eps_list):
        if k > 0:
            p.eps_acc = eps
            fs.resume(p)
        while True:
            r = fs.run(5000, poll_every=100)
            if r.state != _lib.ST_RUNNING:
                break
        t_gpu = time.perf_counter() - t0
        x, y = fs.solution()
        tc = time.perf_counter()
        cert = certificate(inst, x, y)
        t0 += time.perf_counter() - tc                # the host's f64 evaluation is not solver time
        cert.update({"eps_acc": eps, "state": r.state, "iterations": r.iters + 1, "gpu_criteria_f32": list(r.cri),
                     "gpu_seconds_cumulative": t_gpu, "f64_evaluation_seconds": time.perf_counter() - tc})
        out["points"].append(cert)
        sys.stderr.write(json.dumps(cert) + "\n")
        sys.stderr.flush()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
