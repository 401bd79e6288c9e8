import sys, time, numpy as np
sys.path.insert(0, ".")
from totsu_amd import _lib
from totsu_amd.fused import DeviceBuffer
_lib.init()
a = np.random.default_rng(0).standard_normal(200_000_000).astype(np.float32)   # 800 MB pageable
for rep in range(2):
    t0 = time.perf_counter(); d = DeviceBuffer.from_host(a); t1 = time.perf_counter()
    b = d.to_host(); t2 = time.perf_counter()
    print("h2d %.2f GB/s   d2h %.2f GB/s   roundtrip ok %s" % (0.8 / (t1 - t0), 0.8 / (t2 - t1), np.array_equal(a, b)))
    d.free()
