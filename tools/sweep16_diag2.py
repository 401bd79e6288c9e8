import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import totsu_amd as T
from totsu_amd import _lib
_lib.init(0)
D = T.DeviceBuffer
m, n = 4096, 3000
for kind in ("bf16", "f16"):
    A = np.tile(((np.arange(m) % 64) + 1).astype(np.float32)[:, None], (1, n))
    mat = T.Bf16Matrix.from_f32(np.asfortranarray(A).ravel(order="F"), m, n, kind)
    for r0 in (0, 1, 2, 3, 9, 700):
        host = dict(v=np.zeros(m), xy=np.zeros(m), c=np.zeros(n), su=np.ones(n), tx=np.ones(n), u=np.zeros(n), xx=np.zeros(n), gp=np.zeros(n))
        host["xy"][r0] = 1.0
        host["v"][:] = 1.0
        bufs = {k: D.from_host(np.asarray(a, np.float32)) for k, a in host.items()}
        outs = {k: D(sz, zero=True) for k, sz in dict(xx_out=n, hn=m + 8, h3=m + 8).items()}
        t = _lib.SweepTest()
        t.m, t.n, t.lda = m, n, mat.ld16
        t.mat_a, t.v, t.xy, t.c, t.su, t.tx = mat.ptr, bufs["v"].ptr, bufs["xy"].ptr, bufs["c"].ptr, bufs["su"].ptr, bufs["tx"].ptr
        t.u, t.ku, t.xx_in, t.kx_in, t.xx_out, t.kx_out = bufs["u"].ptr, None, bufs["xx"].ptr, None, outs["xx_out"].ptr, None
        t.gp, t.hn, t.h3 = bufs["gp"].ptr, outs["hn"].ptr, outs["h3"].ptr
        t.kappa, t.rtau, t.first, t.reps, t.force_members = 0.0, 0.0, 1, 1, 0
        t.elem, t.inv_s, t.variant = (1 if kind == "bf16" else 2), mat.inv_ptr, 2
        ms, info = (C.c_float * 2)(), (C.c_int * 8)()
        _lib.lib.thip_test_sweep(C.byref(t), ms, info)
        gp = bufs["gp"].to_host(); x = outs["xx_out"].to_host()
        print(kind, "r0", r0, "expect gp", (r0 % 64) + 1, "got", gp[:3], "uniq", np.unique(gp)[:5], "| x (sum of column = %g) got" % A[:, 0].sum(), x[:3])
    if kind == "f16":
        print("inv", mat._inv.to_host()[:3], "raw", mat._buf.to_host().view(np.uint16)[:8])
