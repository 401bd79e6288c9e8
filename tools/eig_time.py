import sys, time, numpy as np
sys.path.insert(0, ".")
from totsu_amd import F32HIP as L, _lib
_lib.init()
rng = np.random.default_rng(0)
for k in (4, 8, 16, 24, 32, 48, 64, 128, 256, 500):
    b = rng.standard_normal((k, k)); s = b @ b.T / k + 0.05 * np.eye(k)
    packed = np.array([s[r, c] for c in range(k) for r in range(c + 1)], dtype=np.float32)
    work = L.Sl.new_mut(np.zeros(L.map_eig_worklen(k), dtype=np.float32))
    for name in ("sqrt_pos", "pos"):
        sl = L.Sl.new_mut(packed.copy()); L.map_eig(sl, None, 1e-12, work, name); L.sync()
        t0 = time.perf_counter(); L.map_eig(sl, None, 1e-12, work, name); L.sync(); t1 = time.perf_counter()
        print("k=%d map_eig(%s): %.2f ms" % (k, name, 1e3 * (t1 - t0)))
        sl.drop()
    work.drop()
