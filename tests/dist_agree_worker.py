"""Worker of tests/test_dist_cpu.py::test_ranks_agree_on_column_shards_gloo: two gloo ranks decide together whether the
column-sharded one-pass run may be built (totsu_amd.parallel.agree_on_column_shards, what bench.py does with the answers of
thip_sweep_probe)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    import torch
    import torch.distributed as dist
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from totsu_amd.parallel import agree_on_column_shards

    def allreduce(v):
        t = torch.from_numpy(np.ascontiguousarray(v).copy())
        dist.all_reduce(t)
        return t.numpy()

    res = {
        "all_yes": agree_on_column_shards(True, allreduce, world),
        "rank1_no": agree_on_column_shards(rank != 1, allreduce, world),      # this rank's probe said "not 8 x 32 CUs"
        "rank0_no": agree_on_column_shards(rank != 0, allreduce, world),
    }
    json.dump(res, open(os.path.join(out_dir, "rank%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
