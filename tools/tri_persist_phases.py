"""Phase times inside the persistent Householder reduction (tri_persist_k), from s_memrealtime stamps taken by workgroup 0.
Needs a library built with -DTHIP_TP_PROFILE:  make -C totsu_amd/csrc clean && make -C totsu_amd/csrc CXXEXTRA=-DTHIP_TP_PROFILE
    python tools/tri_persist_phases.py [k]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from totsu_amd import F32HIP as L, _lib
_lib.init()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(0)
b = rng.standard_normal((k, k)); s = (b + b.T) / 2
packed = np.array([s[r, c] for c in range(k) for r in range(c + 1)], dtype=np.float32)
wl = L.map_eig_worklen(k)
work = L.Sl.new_mut(np.zeros(wl, dtype=np.float32))
for _ in range(3):
    sl = L.Sl.new_mut(packed.copy()); L.map_eig(sl, None, 1e-12, work, "sqrt_pos"); L.sync(); sl.drop()
ld = (k + 63) // 64 * 64
off = 5 * ld * ld + 2 * ld + 16 + 128
w = work.get_ref()
st = np.frombuffer(np.ascontiguousarray(w[off:off + 10]).tobytes(), dtype=np.uint64)
names = ["gather (exchange)", "sum 1 (p . v, vote)", "sum 2 (norm)", "v_j + barrier", "pass over own columns + publish"]
tot = st.sum()
for n_, v in zip(names, st):
    print("%-34s %8.3f us per reflector  (%4.1f %%)" % (n_, v * 0.01 / (k - 2), 100.0 * v / max(tot, 1)))
print("%-34s %8.3f us per reflector" % ("total", tot * 0.01 / (k - 2)))
