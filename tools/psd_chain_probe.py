"""What a launch of the PSD chain costs by shape (thip_test_chain_probe): dependent launches of ld x ld f32 products.
Usage: python tools/psd_chain_probe.py [ld ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from totsu_amd import _lib     # noqa: E402

_lib.init()
lib = _lib.lib
NAMES = ["32x64 blocks, symmetric, batch 2", "32x64 blocks, general, batch 2", "one tile, symmetric, batch 2",
         "one tile, symmetric, one item", "one tile, symmetric, one item on each of 2 streams (per pair)",
         "one tile, general, one item", "polar_dual_k: two products sharing A, batch 2", "polar_dual_k: one product, dsym, batch 2",
         "polar_dual_k: one product, packed output + rx, batch 2", "32x64 blocks, symmetric, dsym, batch 2"]
for ld in [int(a) for a in sys.argv[1:]] or [512]:
    for mode, name in enumerate(NAMES):
        best = 1e9
        for _ in range(3):
            us = C.c_float()
            lib.thip_test_chain_probe(mode, ld, 200, C.byref(us))
            best = min(best, us.value)
        print("ld=%d mode %d %-62s %.2f us per launch" % (ld, mode, name, best), flush=True)
