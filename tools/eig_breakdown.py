"""k = 500 general eigen-decomposition (Householder + QL engine): wall time of map_eig(sqrt) x 5, for rocprofv3"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from totsu_amd import F32HIP as L, _lib
_lib.init()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(0)
b = rng.standard_normal((k, k)); s = b @ b.T / k + 0.05 * np.eye(k)
packed = np.array([s[r, c] for c in range(k) for r in range(c + 1)], dtype=np.float32)
work = L.Sl.new_mut(np.zeros(L.map_eig_worklen(k), dtype=np.float32))
for i in range(6):
    sl = L.Sl.new_mut(packed.copy()); sl.dev(); L.sync()
    t0 = time.perf_counter(); L.map_eig(sl, None, 1e-12, work, "sqrt_pos"); L.sync(); t1 = time.perf_counter()
    print("k=%d map_eig(sqrt): %.2f ms" % (k, 1e3 * (t1 - t0)))
    sl.drop()
