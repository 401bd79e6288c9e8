"""Worker of tests/test_dist_cpu.py: one rank of a world_size-2 gloo job (CPU).  Runs the trait-level
ShardedSolver on a numpy backend over this rank's cone-aligned row block and writes its result for rank 0's
comparison with the single-process oracle."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    import torch.distributed as dist
    out_dir, case = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from np_backend import F64NP as La
    from problems import benchmark_lp, random_socp
    from totsu_amd.cone import ConeRPos, ConeSOC, ConeZero
    from totsu_amd.matop import MatOp, MatType
    from totsu_amd.parallel import ShardedSolver, TorchComm, shard_segments
    from totsu_amd.problem import _ConeStack2, _ProbSOCPCone
    from totsu_amd.solver import Solver, SolverError

    if case == "socp":
        n, cones = 16, [4, 9, 0, 2, 7, 5]
        f, Gs, hs, cs, d = random_socp(n, cones, seed=11)
        rows = [np.vstack([-c.reshape(1, n), -G]) for G, c in zip(Gs, cs)]
        bs = [np.concatenate([[dd], h]) for dd, h in zip(d, hs)]
        seg_len = [1 + k for k in cones]
        s0, s1, r0, r1 = shard_segments(seg_len, world, rank)
        A = np.vstack(rows).astype(np.float64)[r0:r1]
        b = np.concatenate(bs).astype(np.float64)[r0:r1]
        c = f.astype(np.float64)
        cone = _ProbSOCPCone(La, [k - 1 for k in seg_len[s0:s1]], 0)
        max_iter, eps = 20000, 1e-7
    else:
        c32, G, h = benchmark_lp(20, seed=12)
        seg_len = [1] * 40                         # nonneg rows are separable: any row split is cone-aligned
        s0, s1, r0, r1 = shard_segments(seg_len, world, rank)
        A = G.astype(np.float64)[r0:r1]
        b = h.astype(np.float64)[r0:r1]
        c = c32.astype(np.float64)
        cone = _ConeStack2(r1 - r0, 0, ConeRPos(La), ConeZero(La))
        max_iter, eps = 60000, 1e-6
    m = r1 - r0
    op_c = MatOp(La, MatType.General(c.size, 1), c)
    op_a = MatOp(La, MatType.General(m, c.size), np.asfortranarray(A).ravel(order="F"))
    op_b = MatOp(La, MatType.General(m, 1), b)
    comm = TorchComm()
    s = ShardedSolver(La, comm)
    s.param.max_iter, s.param.eps_acc = max_iter, eps
    s.trace = []
    work = np.zeros(Solver.query_worklen((m, c.size)))
    status = 0
    try:
        x, y = s.solve((op_c, op_a, op_b, cone, work))
    except SolverError as e:
        status = e.kind
        x, y = work[:c.size], work[c.size:c.size + m]
    json.dump({"rank": rank, "status": status, "iters": s.trace[-1][0], "x": list(map(float, x)), "y": list(map(float, y)),
               "rows": [r0, r1], "trace_head": [list(t) for t in s.trace[:30]], "collectives": comm.n_collectives},
              open(os.path.join(out_dir, "rank%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
