import sys, os, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle as O
from totsu_amd import F32HIP as L, _lib, ConePSD
_lib.init()
from test_gpu_eig import _rand_sym, _packed
for k in (21, 33, 64, 65, 100, 128, 200, 256, 500):
    for rd in (False, True):
        s = _rand_sym(k, k + 17 * rd, rd)
        x = _packed(s)
        ref = O.proj(O.CONE_PSD, x.astype(np.float64), use_ql=True)
        w = np.zeros(ConePSD.query_worklen(L, x.size), dtype=np.float32)
        cone = ConePSD(L, w, 1e-12)
        sl = L.Sl.new_mut(x.copy())
        cone.proj(False, sl)
        got = sl.get_ref().copy()
        sl.drop(); cone.drop()
        print("k=%d rank_def=%s err/|x| = %.2e" % (k, rd, np.abs(got - ref).max() / np.linalg.norm(x)), flush=True)
