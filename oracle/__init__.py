"""ctypes binding of the CPU oracle (oracle/libtotsu_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under totsu_amd/ may import this package.
"""
from .binding import *  # noqa: F401,F403
