"""The one-pass kernel alone at a fixed shape (default BASELINE configs[2]) for f32 / bf16 / f16 storage and a list of
(workgroups per column group, columns per panel) geometries: time of a sweep and the bytes of A it streams per second.

    python tools/sweep16_probe.py [--m M --n N --reps R --geoms 0:0,16:2,16:4 --elems 0,1,2]

`0:0` = the planner's default geometry.  With SWEEP_SCALING_SO=<a -DSW_PROFILE build> the service wave's time split and the
streaming wave's phase stamps come out on stderr after every case.  Timing only: the matrix holds arbitrary bit patterns."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=100_000)
    ap.add_argument("--n", type=int, default=50_000)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--geoms", default="0:0,8:1,16:1,16:2,16:4,32:4")
    ap.add_argument("--elems", default="0,1,2")
    a = ap.parse_args()
    from totsu_amd import _lib
    if os.environ.get("SWEEP_SCALING_SO"):
        _lib.SO_PATH = os.environ["SWEEP_SCALING_SO"]
    from totsu_amd.fused import DeviceBuffer
    _lib.init(0)
    lib = _lib.lib
    m, n = a.m, a.n
    A = DeviceBuffer(m * n)
    lib.thip_gen_matrix(A.ptr, m, n, m, 0, 1, 0, 0, m, 1, 0.01, 0.0)
    vecs = {k: DeviceBuffer(max(m, n) + 64, zero=True) for k in ("v", "xy", "c", "su", "tx", "u", "xx", "gp", "xo", "hn", "h3")}
    import numpy as np
    ones = DeviceBuffer.from_host(np.ones(n, np.float32))
    for elem in [int(v) for v in a.elems.split(",")]:
        esize = 4 if elem == 0 else 2
        for geom in a.geoms.split(","):
            G, W = [int(v) for v in geom.split(":")]
            t = _lib.SweepTest()
            t.m, t.n, t.lda = m, n, m                     # (16-bit: lda counts 16-bit elements; the same buffer, half used)
            t.mat_a, t.v, t.xy, t.c, t.su, t.tx = A.ptr, vecs["v"].ptr, vecs["xy"].ptr, vecs["c"].ptr, vecs["su"].ptr, vecs["tx"].ptr
            t.u, t.ku, t.xx_in, t.kx_in, t.xx_out, t.kx_out = vecs["u"].ptr, None, vecs["xx"].ptr, None, vecs["xo"].ptr, None
            t.gp, t.hn, t.h3 = vecs["gp"].ptr, vecs["hn"].ptr, vecs["h3"].ptr
            t.kappa, t.rtau, t.first, t.reps, t.force_members = 0.0, 0.0, 1, a.reps, G
            t.elem, t.inv_s = elem, (ones.ptr if elem == 2 else None)
            t.variant = (12 if W == 2 else 0) if elem == 0 else W
            ms, info = (C.c_float * 2)(), (C.c_int * 8)()
            try:
                lib.thip_test_sweep(C.byref(t), ms, info)
            except Exception as e:
                print("elem %d G %2d W %d: refused (%s)" % (elem, G, W, str(e).splitlines()[0][:90]), flush=True)
                continue
            print("elem %d asked G %2d W %d -> G %2d slots %d panels %5d err %d: best %.1f us avg %.1f us = %.0f GB/s of A"
                  % (elem, G, W, info[1], info[4], info[3], info[0], 1e3 * ms[0], 1e3 * ms[1],
                     esize * m * n / (ms[0] * 1e-3) / 1e9), flush=True)
            sys.stderr.flush()
    A.free(); ones.free()
    for d in vecs.values():
        d.free()


if __name__ == "__main__":
    main()
