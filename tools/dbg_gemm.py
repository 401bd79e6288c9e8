import sys, ctypes as C, numpy as np
sys.path.insert(0, ".")
from totsu_amd import _lib
from totsu_amd.fused import DeviceBuffer
_lib.init()
raw = _lib.load()
f = raw.thip_dbg_gemm_sym
f.argtypes = [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p]
rng = np.random.default_rng(0)
for n, ld in ((64, 64), (65, 128), (500, 512)):
    def sym():
        b = rng.standard_normal((ld, ld)).astype(np.float32); b = (b + b.T) / 2
        b[n:, :] = 0; b[:, n:] = 0
        return b
    A, B, D = sym(), sym(), sym()
    dA, dB, dD, dC = [DeviceBuffer.from_host(x.ravel(order="F")) for x in (A, B, D)] + [DeviceBuffer(ld * ld)]
    rc = f(n, ld, 0.5, dA.ptr, dB.ptr, -2.0, dD.ptr, 3.0, dC.ptr)
    got = dC.to_host().reshape((ld, ld)).T      # col-major -> [r, c]
    I = np.zeros((ld, ld)); I[:n, :n] = np.eye(n)
    ref = 0.5 * A.astype(np.float64) @ B.astype(np.float64) - 2.0 * D + 3.0 * I
    # the kernel stores the transposed element: compare with ref^T
    e1 = np.abs(got - ref).max(); e2 = np.abs(got - ref.T).max()
    print(n, ld, rc, "GEN: A sym, B general: err vs ref %.3e  vs ref^T %.3e  scale %.2f" % (e1, e2, np.abs(ref).max()))
    Bg = rng.standard_normal((ld, ld)).astype(np.float32); Bg[n:, :] = 0; Bg[:, n:] = 0
    dG = DeviceBuffer.from_host(Bg.ravel(order="F"))
    rc = f(n, ld, 1.0, dA.ptr, dG.ptr, 0.0, None, 0.0, dC.ptr)
    got = dC.to_host().reshape((ld, ld)).T
    print("   A*Bgeneral err %.3e" % np.abs(got - A.astype(np.float64) @ Bg.astype(np.float64)).max())
    # A B with A == B
    rc = f(n, ld, 1.0, dA.ptr, dA.ptr, 0.0, None, 0.0, dC.ptr)
    got = dC.to_host().reshape((ld, ld)).T
    ref = A.astype(np.float64) @ A.astype(np.float64)
    print("   A*A err %.3e" % np.abs(got - ref).max())
