"""Run bench.py (or any script) against ANOTHER build of libtotsu_f32hip.so on the same box: A/B of two library builds.
    python tools/ab_lib.py <path to .so> bench.py --a-storage bf16 --no-cpu --no-to-eps
Symbols the other build lacks are dropped from the binding table (an older build under the current Python layer)."""
import ctypes as C
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from totsu_amd import _lib  # noqa: E402

so = os.path.abspath(sys.argv[1])
try:
    import torch  # noqa: F401  (one HIP runtime in the process: as _lib.load does)
except Exception:
    pass
_lib.SO_PATH = so
probe = C.CDLL(so, mode=C.RTLD_LOCAL)
for name in list(_lib.PROTOTYPES):
    if not hasattr(probe, name):
        del _lib.PROTOTYPES[name]
        setattr(_lib.lib, name, (lambda *a: 0))         # a call the other build does not know does nothing
sys.argv = sys.argv[2:]
runpy.run_path(os.path.join(ROOT, sys.argv[0]) if not os.path.isabs(sys.argv[0]) else sys.argv[0], run_name="__main__")
