"""Worker of tests/test_gpu_sharded.py::test_oneshot_allreduce_*: N processes (torch.distributed.run, gloo for the
bootstrap) sharing the one GPU of the box; every rank calls the library's one-shot all-reduce over peer-mapped buffers on
a sequence of messages and checks the result against the sum IN RANK ORDER computed on the host -- bit for bit."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from totsu_amd import _lib                     # noqa: E402
from totsu_amd._lib import lib                 # noqa: E402
from totsu_amd.fused import DeviceBuffer       # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _lib.init(0)
    cap = 60_000
    hb = (C.c_uint8 * 64)()
    lib.thip_oneshot_init(rank, world, cap, hb)
    mine = torch.frombuffer(bytearray(bytes(hb)), dtype=torch.uint8)
    allh = [torch.zeros(64, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(allh, mine)
    lib.thip_oneshot_connect((C.c_uint8 * (64 * world)).from_buffer_copy(b"".join(t.numpy().tobytes() for t in allh)))
    dist.barrier()
    # lengths: one element, not a multiple of 4, exactly a chunk, several chunks + a ragged tail, the loop's own size
    lengths = [1, 7, 2048, 2049, 5000, 51_024, 4, 51_024, 51_024, 333]
    buf = DeviceBuffer(cap)
    for call, n in enumerate(lengths * 3):
        # every rank can regenerate every rank's contribution: no second channel needed for the expected value
        contrib = [np.random.default_rng(1000 * call + r).standard_normal(n).astype(np.float32) * (10.0 ** (r - 1)) for r in range(world)]
        lib.thip_h2d(buf.ptr, contrib[rank].ctypes.data, n)
        lib.thip_oneshot_allreduce(buf.ptr, n)
        got = np.empty(n, dtype=np.float32)
        lib.thip_d2h(got.ctypes.data, buf.ptr, n)
        want = np.zeros(n, dtype=np.float32)
        for r in range(world):
            want = (want + contrib[r]).astype(np.float32)           # rank order, f32 at every step
        assert np.array_equal(got, want), (rank, call, n, np.abs(got - want).max())
    err = C.c_int(0)
    lib.thip_oneshot_error(C.byref(err))
    assert err.value == 0
    dist.barrier()
    lib.thip_oneshot_destroy()
    buf.free()
    dist.barrier()
    if rank == 0:
        print("ONESHOT_OK world=%d calls=%d" % (world, 3 * len(lengths)))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
