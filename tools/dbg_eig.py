import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle as O
from totsu_amd import F32HIP as L, _lib, ConePSD
_lib.init()
from test_gpu_eig import _rand_sym, _packed
for k in (65, 100, 128, 200, 500):
    for rd in (False, True):
        s = _rand_sym(k, k + 17 * rd, rd)
        x = _packed(s)
        ref = O.proj(O.CONE_PSD, x.astype(np.float64), use_ql=True)
        w = np.zeros(ConePSD.query_worklen(L, x.size), dtype=np.float32)
        cone = ConePSD(L, w, 1e-12)
        sl = L.Sl.new_mut(x.copy()); cone.proj(False, sl); got = sl.get_ref().copy(); sl.drop(); cone.drop()
        print(k, rd, "err/|x| = %.3e" % (np.abs(got - ref).max() / np.linalg.norm(x)), "nan" if np.isnan(got).any() else "")
