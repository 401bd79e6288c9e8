"""The conic loop on a STENCIL operator (5-point Laplacian on a g x g grid as the inequality matrix of an LP: the matrix-free pattern
of examples/imgnr_udef) through the tiled copy under the one-pass schedule and through round 5's two CSR copies under the carried
one: iterations per second side by side.  Per product the CSR gathers are the faster form on this pattern (profiles/
r06_sparse_product_rates.txt); per ITERATION the tiled copy serves two right-hand sides per pass.
    python tools/stencil_loop_compare.py [g = 3000]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import totsu_amd as T  # noqa: E402
from totsu_amd import _lib  # noqa: E402


def main():
    _lib.init()
    import torch
    g = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    e = np.ones(g, np.float32)
    L1 = sp.diags([-e[:-1], 2 * e, -e[:-1]], [-1, 0, 1], format="csc")
    A = (sp.kron(sp.identity(g, dtype=np.float32), L1) + sp.kron(L1, sp.identity(g, dtype=np.float32))).tocsc().astype(np.float32)
    A.sort_indices()
    m, n = A.shape
    rng = np.random.default_rng(0)
    b = rng.uniform(0.5, 1.5, m).astype(np.float32)
    c = rng.standard_normal(n).astype(np.float32)
    p = T.SolverParam()
    p.eps_acc = 0.0
    p.eps_inf = 0.0
    p.max_iter = None
    print("5-point Laplacian, %d x %d grid: m = n = %d, %d entries" % (g, g, n, A.nnz))
    for name, kw, sched in (("tiled copy, one-pass schedule", {}, "sweep"), ("tiled copy, carried schedule", {}, "carried"),
                            ("two CSR copies, carried schedule", {"sparse_two_copies": True}, "carried")):
        fs = T.FusedSolver(n, m, A, b, c, [1], [m], p, sched, **kw)
        fs.run(20, poll_every=20)
        torch.cuda.synchronize()
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            fs.run(200, poll_every=200)
            torch.cuda.synchronize()
            best = max(best, 200 / (time.perf_counter() - t0))
        print("    %-36s %8.1f iter/s" % (name, best))
        fs.destroy()


if __name__ == "__main__":
    main()
