"""The f32 floor of the dual criterion and what compensated iterate updates (thip_param.state_arith) do to it:
a small SOCP (n = 200, 6 cones of 1 + 99 rows) run for a fixed number of iterations with eps_acc = 0; prints the
criteria along the way.  A numpy emulation of the same iteration gives 6.3e-6 (plain f32) vs 1e-7 (Kahan).
Usage: [STATE=plain] python tools/state_floor.py [n] [cones] [iters]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import totsu_amd as T                       # noqa: E402
from problems import random_socp            # noqa: E402
from totsu_amd import _lib                  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    nc = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30000
    _lib.init()
    f, Gs, hs, cs, d = random_socp(n, [99] * nc, seed=1)
    mb = lambda typ: T.MatBuild(T.F32HIP, typ)
    socp = T.ProbSOCP(mb(T.MatType.General(n, 1)).set_array(f.reshape(-1, 1)),
                      [mb(T.MatType.General(G.shape[0], n)).set_array(G) for G in Gs],
                      [mb(T.MatType.General(len(h_), 1)).set_array(h_.reshape(-1, 1)) for h_ in hs],
                      [mb(T.MatType.General(n, 1)).set_array(c_.reshape(-1, 1)) for c_ in cs], d,
                      mb(T.MatType.General(0, n)), mb(T.MatType.General(0, 1)))
    p = T.SolverParam()
    p.eps_acc, p.max_iter = 0.0, None
    p.state_arith = os.environ.get("STATE", "compensated")
    for sched in ("reference", "carried"):
        fs = T.FusedSolver.from_dense(socp.dense(), p, sched)
        for k in range(6):
            r = fs.run(iters // 6, poll_every=64)
            print("compensated=%s %-9s iter %6d  pri %.3e dual %.3e gap %.3e"
                  % (p.state_arith, sched, r.iters, r.cri[0], r.cri[1], r.cri[2]), flush=True)
        fs.destroy()


if __name__ == "__main__":
    main()
