"""Worker of tests/test_dist_cpu.py::test_column_sharded_sweep_gloo: one rank of a world_size-2 gloo job (CPU) running
tests/sweep_numpy.py over its block of columns; writes its iterates for the comparison with the single-process oracle."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    import torch
    import torch.distributed as dist
    out_dir, case = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from problems import benchmark_lp, random_socp
    from sweep_numpy import CONE_RPOS, CONE_SOC, SweepCols

    def allreduce(v):
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64).copy())
        dist.all_reduce(t)
        return t.numpy()

    if case == "socp":
        n, cones = 16, [4, 9, 0, 2, 7, 5]
        f, Gs, hs, cs, d = random_socp(n, cones, seed=11)
        A = np.vstack([np.vstack([-c.reshape(1, n), -G]) for G, c in zip(Gs, cs)]).astype(np.float64)
        b = np.concatenate([np.concatenate([[dd], h]) for dd, h in zip(d, hs)]).astype(np.float64)
        c = f.astype(np.float64)
        seg_type, seg_len = [CONE_SOC] * len(cones), [1 + k for k in cones]
    else:
        c32, G, h = benchmark_lp(20, seed=12)
        A, b, c = G.astype(np.float64), h.astype(np.float64), c32.astype(np.float64)
        seg_type, seg_len = [CONE_RPOS], [40]
    n = c.size
    lo, hi = (0, 7) if rank == 0 else (7, n)          # an uneven split
    s = SweepCols(A[:, lo:hi], b, c[lo:hi], seg_type, seg_len, allreduce)
    snaps, cri = {}, []
    for k in range(1, 61):
        s.step()
        cri.append(list(map(float, s.cri)))
        if k in (1, 2, 10, 60):
            (xx, xm), (u, ym) = s.iterate()
            snaps[k] = dict(xx=list(xx), xm=list(xm), u=list(u), ym=list(ym))
    json.dump({"rank": rank, "cols": [lo, hi], "snaps": snaps, "cri": cri, "collectives": s.collectives},
              open(os.path.join(out_dir, "rank%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
