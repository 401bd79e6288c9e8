"""Where a short sweep's time goes: thip_test_sweep over n at fixed m (time = fixed + per-column), for the planner's group
size and forced ones.   python tools/sweep_scaling.py [m ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from totsu_amd import _lib
    if os.environ.get("SWEEP_SCALING_SO"):        # e.g. a -DSW_PROFILE build of thip_sweep.hip (phase stamps on stderr)
        _lib.SO_PATH = os.environ["SWEEP_SCALING_SO"]
    from totsu_amd.fused import DeviceBuffer
    _lib.init(0)
    lib = _lib.lib
    ms_list = [int(v) for v in sys.argv[1:]] or [20000, 125252, 100000]
    reps = int(os.environ.get("SWEEP_SCALING_REPS", "8"))
    gs = [int(v) for v in os.environ.get("SWEEP_SCALING_G", "0,1,2,4,8,16,32").split(",")]
    for m in ms_list:
        nmax = min(40000, int(6e9 / (4 * m)))
        A = DeviceBuffer(m * nmax)
        lib.thip_gen_matrix(A.ptr, m, nmax, m, 0, 1, 0, 0, m, 1, 0.01, 0.0)
        vecs = {k: DeviceBuffer(max(m, nmax), zero=True) for k in ("v", "xy", "c", "su", "tx", "u", "xx", "gp", "xo", "hn", "h3")}
        for G in gs:
            pts = []
            for n in (nmax // 16, nmax // 8, nmax // 4, nmax // 2, nmax):
                t = _lib.SweepTest()
                t.m, t.n, t.lda = m, n, m
                t.mat_a, t.v, t.xy, t.c, t.su, t.tx = A.ptr, vecs["v"].ptr, vecs["xy"].ptr, vecs["c"].ptr, vecs["su"].ptr, vecs["tx"].ptr
                t.u, t.ku, t.xx_in, t.kx_in, t.xx_out, t.kx_out = vecs["u"].ptr, None, vecs["xx"].ptr, None, vecs["xo"].ptr, None
                t.gp, t.hn, t.h3 = vecs["gp"].ptr, vecs["hn"].ptr, vecs["h3"].ptr
                t.variant = int(os.environ.get('SWEEP_SCALING_VARIANT', '0'))
                t.kappa, t.rtau, t.first, t.reps, t.force_members, t.pub_agent = 0.0, 0.0, 1, reps, G, int(os.environ.get('SWEEP_SCALING_PUB', '0'))
                ms, info = (C.c_float * 2)(), (C.c_int * 8)()
                try:
                    lib.thip_test_sweep(C.byref(t), ms, info)
                except Exception:
                    pts = None
                    break
                pts.append((n, ms[0], info[1], info[4], info[3]))
            if not pts:
                continue
            ns = np.array([p[0] for p in pts], float)
            ts = np.array([p[1] for p in pts], float)
            b, a = np.polyfit(ns, ts, 1)
            print("m %6d G %2d (asked %2d) slots %d: fixed %.1f us, per column %.3f us = %.0f GB/s asymptotic; points %s"
                  % (m, pts[-1][2], G, pts[-1][3], 1e3 * a, 1e3 * b, 4.0 * m / (b * 1e-3) / 1e9,
                     " ".join("n=%d:%.1fus(%.0fGB/s)" % (p[0], 1e3 * p[1], 4.0 * m * p[0] / (p[1] * 1e-3) / 1e9) for p in pts)), flush=True)
        A.free()
        for d in vecs.values():
            d.free()


if __name__ == "__main__":
    main()
