"""The closure path of map_eig (thip_eig_decompose -> host closure -> thip_eig_rebuild) on both tridiagonal engines:
which engine served, the certificate, eigenvalue / reconstruction error against numpy (f64) and the wall time.
Run on a GPU box: python tools/eig_engine_probe.py [k ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from totsu_amd import F32HIP as L, _lib  # noqa: E402
from tools.tridiag_vec_probe import cases  # noqa: E402

_lib.init()
lib = _lib.load()


def info():
    e, p = C.c_int(), C.c_int()
    c = (C.c_float * 3)()
    lib.thip_eig_engine_info(C.byref(e), C.byref(p), c)
    return e.value, p.value, c[0], c[1], c[2]


def run(s, force, reps=3):
    k = s.shape[0]
    packed = np.array([s[r, c] for c in range(k) for r in range(c + 1)], dtype=np.float32)
    work = L.Sl.new_mut(np.zeros(L.map_eig_worklen(k), dtype=np.float32))
    lib.thip_test_eig_force(force)
    seen = []
    sl = L.Sl.new_mut(packed.copy())
    L.map_eig(sl, None, 1e-12, work, lambda e: (seen.append(e), e)[1])
    got = sl.get_ref().copy()
    sl.drop()
    eng = info()
    ts = []
    for _ in range(reps):
        sl = L.Sl.new_mut(packed.copy())
        sl.dev()
        L.sync()
        t0 = time.perf_counter()
        L.map_eig(sl, None, 1e-12, work, "sqrt_pos")
        L.sync()
        ts.append(time.perf_counter() - t0)
        sl.drop()
    work.drop()
    lib.thip_test_eig_force(0)
    s32 = np.zeros((k, k))
    for c in range(k):
        for r in range(c + 1):
            s32[r, c] = s32[c, r] = packed[c * (c + 1) // 2 + r]
    w = np.linalg.eigvalsh(s32)
    nrm = max(np.abs(w).max(), 1e-300)
    return dict(engine=eng[0], polish=eng[1], orth=eng[2], resid=eng[3], tri=eng[4],
                eig_err=np.abs(np.sort(np.array(seen, dtype=np.float64)) - w).max() / nrm,
                recon_err=np.abs(got.astype(np.float64) - packed).max() / nrm, ms=1e3 * min(ts))


if __name__ == "__main__":
    ks = [int(a) for a in sys.argv[1:]] or [33, 100, 500]
    for k in ks:
        rng = np.random.default_rng(1)
        for name, a in cases(k, rng):
            for force in (0, 12, 4, 8, 1):
                r = run(a, force)
                print("k=%-4d %-34s %s  engine=%d tri=%d polish=%d orth=%.1e resid=%.1e eig=%.1e recon=%.1e  %.2f ms" % (
                    k, name, {0: "default  ", 12: "launches ", 4: "persist  ", 8: "persistX ", 1: "forced-QL"}[force], r["engine"], r["tri"], r["polish"], r["orth"], r["resid"],
                    r["eig_err"], r["recon_err"], r["ms"]), flush=True)
