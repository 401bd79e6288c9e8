"""diagnostic: the one-pass kernel on a 16-bit matrix, error of every output against numpy on the rounded matrix"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import totsu_amd as T
    from totsu_amd import _lib
    from test_gpu_bf16 import bf16_round, f16_quantize
    _lib.init(0)
    D = T.DeviceBuffer
    for kind in ("bf16", "f16"):
        for (m, n, w) in ((4096, 3000, 2), (3584, 2400, 1)):
            rng = np.random.default_rng(1)
            A = (rng.standard_normal((m, n)) * np.exp(rng.uniform(-3, 3, n))[None, :] / np.sqrt(n)).astype(np.float32)
            mat = T.Bf16Matrix.from_f32(np.asfortranarray(A).ravel(order="F"), m, n, kind)
            Ar = (bf16_round(A) if kind == "bf16" else f16_quantize(A)[2]).astype(np.float64)
            host = dict(v=rng.standard_normal(m), xy=rng.standard_normal(m), c=rng.standard_normal(n), su=rng.random(n) + 0.5,
                        tx=rng.random(n) + 0.5, u=rng.standard_normal(n), xx=rng.standard_normal(n), gp=rng.standard_normal(n))
            host = {k: np.asarray(a, np.float32) for k, a in host.items()}
            bufs = {k: D.from_host(a) for k, a in host.items()}
            outs = {k: D(sz, zero=True) for k, sz in dict(xx_out=n, hn=m + 8, h3=m + 8).items()}
            t = _lib.SweepTest()
            t.m, t.n, t.lda = m, n, mat.ld16
            t.mat_a, t.v, t.xy, t.c, t.su, t.tx = mat.ptr, bufs["v"].ptr, bufs["xy"].ptr, bufs["c"].ptr, bufs["su"].ptr, bufs["tx"].ptr
            t.u, t.ku, t.xx_in, t.kx_in, t.xx_out, t.kx_out = bufs["u"].ptr, None, bufs["xx"].ptr, None, outs["xx_out"].ptr, None
            t.gp, t.hn, t.h3 = bufs["gp"].ptr, outs["hn"].ptr, outs["h3"].ptr
            t.kappa, t.rtau, t.first, t.reps, t.force_members = -0.37, 0.81, 1, 1, 0
            t.elem, t.inv_s, t.variant = (1 if kind == "bf16" else 2), mat.inv_ptr, w
            ms, info = (C.c_float * 2)(), (C.c_int * 8)()
            _lib.lib.thip_test_sweep(C.byref(t), ms, info)
            g3 = Ar.T @ host["xy"].astype(np.float64)
            gT = Ar.T @ host["v"].astype(np.float64)
            x_ref = host["xx"] + host["tx"] * (gT + host["c"] * (-0.37))
            u_ref = host["u"].astype(np.float64)
            got = {"x": outs["xx_out"].to_host(), "gp": bufs["gp"].to_host(), "hn": outs["hn"].to_host()[:m], "h3": outs["h3"].to_host()[:m]}
            ref = {"x": x_ref, "gp": g3, "hn": Ar @ u_ref, "h3": Ar @ x_ref}
            errs = {k: float(np.abs(got[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-30)) for k in ref}
            inv = mat._inv.to_host() if kind == "f16" else None
            print(kind, m, n, w, list(info), errs)
            if kind == "f16":
                r = got["gp"] / g3
                print("  gp ratio got/ref: median %.6g, min %.6g max %.6g; inv_s min %.3g max %.3g; ratio*1/inv median %.6g" %
                      (np.median(r), r.min(), r.max(), inv.min(), inv.max(), np.median(r / inv)))
                print("  first columns: got", got["gp"][:4], "ref", g3[:4], "inv", inv[:4])


if __name__ == "__main__":
    main()
