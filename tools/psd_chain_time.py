"""Time of ConePSD::proj through the matrix-core polar chain (thip_proj_psd) at k = 128 / 256 / 500, and its error
against numpy's eigh.  THIP_GEMM_MODE=0 selects the slab-prefetch GEMM (round 1), default = loads-up-front GEMM.
Usage: [THIP_GEMM_MODE=0] [THIP_POLAR_SMALL=0] python tools/psd_chain_time.py [k ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from totsu_amd import F32HIP as L, _lib     # noqa: E402

_lib.init()
rng = np.random.default_rng(0)
KS = [int(a) for a in sys.argv[1:]] or [128, 256, 500]
for k in KS:
    b = rng.standard_normal((k, k))
    s = (b + b.T) / 2
    w, z = np.linalg.eigh(s)
    ref = (z * np.maximum(w, 0)) @ z.T
    sq2 = np.sqrt(2.0)
    pk = lambda a: np.array([a[r, c] * (sq2 if r != c else 1.0) for c in range(k) for r in range(c + 1)], dtype=np.float32)
    packed, pref = pk(s), pk(ref)
    work = L.Sl.new_mut(np.zeros(L.map_eig_worklen(k), dtype=np.float32))
    sl = L.Sl.new_mut(packed.copy())
    L.map_eig(sl, sq2, 1e-12, work, "pos")
    err = np.abs(sl.get_ref() - pref).max() / np.linalg.norm(packed)
    reps, dt = 50, float("inf")
    for _ in range(5):                      # best of 5: a host hiccup in one batch of launches is not the chain's time
        L.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            L.map_eig(sl, sq2, 1e-12, work, "pos")
        L.sync()
        dt = min(dt, (time.perf_counter() - t0) / reps)
    print("k=%d PSD projection: %.3f ms  err/|x| %.2e  (THIP_GEMM_MODE=%s)" % (k, 1e3 * dt, err, os.environ.get("THIP_GEMM_MODE", "default")))
    sl.drop()
    work.drop()
